"""HIP-graph replay of the per-clip forward (fixed clip geometry / token count).

An eager forward is ~1100 kernel launches; on an MI355X the GPU needs ~16 ms for them while the
Python/launch path needs longer, so the whole forward + query selection is captured once into a
hipGraph (torch.cuda.CUDAGraph) and replayed per clip: one launch, no host work in the loop.
The hand-written kernels are launched through the C ABI on torch's current stream, so they are
captured like any other node.  Shapes are static per runner; build one runner per geometry.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import clip_parallel as CP
from . import hot_ops
from . import postprocessing as P
from .nested_tensor import NestedTensor


# Stream-capture error mode.  The drivers decode and upload the NEXT clip on a background thread (clip_io.VideoClipCache)
# while this thread captures a graph: in HIP's default "global" mode a hipMalloc / hipHostMalloc issued by ANY thread during
# the capture invalidates it (hipErrorStreamCaptureInvalidated -- seen as a rare failure of the Ref-YouTube-VOS driver test in
# the first session on a fresh box, when the allocator pools are still cold).  Only this thread's calls matter for the
# capture, so the check is thread-local.
CAPTURE_MODE = "thread_local"


class ClipGraph:
    def __init__(self, model, T: int, H: int, W: int, L: int, device, warmup: int = 2):
        self.model, self.T, self.H, self.W, self.L = model, T, H, W, L
        self.device = torch.device(device)
        self.clip = torch.zeros(T, 1, 3, H, W, device=self.device)
        self.pad = torch.zeros(T, 1, H, W, dtype=torch.bool, device=self.device)
        self.ids = torch.ones(1, L, dtype=torch.long, device=self.device)
        self.attn = torch.ones(1, L, dtype=torch.long, device=self.device)
        self.targets = [[{"size": (H, W)}] for _ in range(T)]
        Q = model.num_queries
        hm, wm = -(-H // 4), -(-W // 4)
        self.record = torch.zeros(CP.record_size(T, Q, hm, wm), device=self.device)
        self.out: Optional[Dict[str, torch.Tensor]] = None
        self.graph = torch.cuda.CUDAGraph()
        assert hot_ops._prof is None, "do not capture while kernel profiling is on"
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # warm-up off the capture: lazy inits, algorithm finds
            for _ in range(warmup):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            self.out = self._forward()

    def _forward(self):
        samples = NestedTensor(self.clip, self.pad, unpadded=True)  # static all-False pad mask
        out = self.model(samples, None, {"input_ids": self.ids, "attention_mask": self.attn}, self.targets)
        idx, masks = P.select_trajectory(out)
        CP.pack_record(self.record, idx, out["pred_cls"][:, 0, :, 0], masks)
        return out

    def run(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None,
            attn: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """clip [T,3,H,W] or [T,1,3,H,W] on the device; ids / attn [1,L] token ids and attention mask (ones
        when omitted); returns the static output dict (valid until the next run) -- `self.record` holds the
        packed (query, scores, selected masks) result."""
        self.stage_inputs(clip, ids, attn)
        return self.replay()

    def stage_inputs(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None,
                     attn: Optional[torch.Tensor] = None) -> None:
        """Copy the inputs into the graph's static buffers (on the current stream).  After this call the caller's `clip`
        may be overwritten by work ordered behind it -- a streaming feeder releases its device slot here, not after the
        replay (clip_io.DoubleBufferedH2D)."""
        self.clip.copy_(clip.view(self.clip.shape), non_blocking=True)
        if ids is not None:
            self.ids.copy_(ids.view(self.ids.shape), non_blocking=True)
            if attn is None:
                if not getattr(self, "_attn_ones", True):   # the static mask is all ones until a caller stages another one
                    self.attn.fill_(1)
                    self._attn_ones = True
            else:
                self.attn.copy_(attn.view(self.attn.shape), non_blocking=True)
                self._attn_ones = False

    def replay(self) -> Dict[str, torch.Tensor]:
        self.graph.replay()
        return self.out


class PipelinedClipGraph:
    """Two clips in flight inside ONE stream of graph replays (software pipeline across clips).

    Graph k runs the TAIL of the previous clip (FPN, query decoder, VOC, heads, mask head, selection -- ~120 short,
    latency-bound launches that leave most of the GPU idle) on a side branch while the HEAD of the next clip
    (text ‖ Video-Swin, fusion, deformable encoder -- chip-filling kernels) runs on the main branch.  The head hands
    over through a double-buffered static state, so graphs 0 / 1 alternate and are never replayed concurrently
    (concurrent replays hang on this stack, tools/experiments/README.md, which also records the three-stage variant
    that was tried and dropped); the side branch forks directly from the capture stream.  Per clip the kernels and
    their order are those of ClipGraph; only which clips' kernels share the GPU changes.

        for clip in clips:
            rec = g.run(clip, ids)        # packed record of the clip submitted one call earlier, or None
        rest = g.flush()                  # list with the record (a clone) of the clip still in flight
    """

    DEPTH = 2

    def __init__(self, model, T: int, H: int, W: int, L: int, device, warmup: int = 2):
        self.model, self.T, self.H, self.W, self.L = model, T, H, W, L
        dev = self.device = torch.device(device)
        self.clip = torch.zeros(T, 1, 3, H, W, device=dev)
        self.pad = torch.zeros(T, 1, H, W, dtype=torch.bool, device=dev)
        self.ids = torch.ones(1, L, dtype=torch.long, device=dev)
        self.attn = torch.ones(1, L, dtype=torch.long, device=dev)
        self.targets = [[{"size": (H, W)}] for _ in range(T)]
        hm, wm = -(-H // 4), -(-W // 4)
        self.record = torch.zeros(CP.record_size(T, model.num_queries, hm, wm), device=dev)
        assert hot_ops._prof is None, "do not capture while kernel profiling is on"
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                sb = self._head(fork=False)
                self._tail(sb, fork=False)
            # static, double-buffered hand-over state (only what changes from clip to clip is copied; geometry
            # constants are shared)
            self.sb = [self._clone(sb), self._clone(sb)]
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # The tail branch is captured from a HIGH-priority stream: its short launches are taken first whenever a CU frees up
        # beside the head's chip-filling kernels (measured on one box, alternating: 6.31 -> 6.24 ms per clip;
        # SOC_TAIL_PRIORITY=0 restores the default-priority branch)
        import os
        self._pc = torch.cuda.Stream(device=dev, priority=0 if os.environ.get("SOC_TAIL_PRIORITY", "1") == "0" else -1)

        def capture(body):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                body()
            return g

        def tail_beside_head(k):
            cur = torch.cuda.current_stream(dev)
            self._pc.wait_stream(cur)
            with torch.cuda.stream(self._pc):
                self._tail(self.sb[1 - k], fork=False)          # clip i-1
            self._store(self._placed_head(self.sb[k]), self.sb[k])      # clip i
            cur.wait_stream(self._pc)

        self.steady = [capture(lambda k=k: tail_beside_head(k)) for k in (0, 1)]
        self.drain = [capture(lambda k=k: self._tail(self.sb[k], fork=True)) for k in (0, 1)]
        self._n = 0            # clips submitted since the last flush

    # -- stages ------------------------------------------------------------------------------------------
    def _head(self, fork: bool):
        samples = NestedTensor(self.clip, self.pad, unpadded=True)
        sa = self.model.forward_backbone(samples, None, {"input_ids": self.ids, "attention_mask": self.attn})
        return self.model.forward_fuse_encode(sa, fork=fork)

    def _placed_head(self, dst):
        """The head with its two largest hand-over tensors -- the stage-0 token map (44 MB at the BASELINE size) and the encoder
        memory (40 MB) -- produced IN the static state `dst` (hot_ops.place_output) instead of copied there behind the head."""
        f0, mem = dst["feats0"], dst["ctx"][0]
        n, c, h, w = f0.shape
        tok = f0.permute(0, 2, 3, 1)                     # '(b t) h w c': the layout the stage writes
        import os
        if os.environ.get("SOC_NO_PLACE", "0") == "1":  # diagnostic: the copies of round 3
            return self._head(fork=True)
        try:
            if tok.is_contiguous():
                hot_ops.place_output("swin0", tok.view(n // self.T, self.T, h, w, c))
            if mem.is_contiguous():
                hot_ops.place_output("encoder_memory", mem)
            return self._head(fork=True)
        finally:
            hot_ops.place_output("swin0", None)
            hot_ops.place_output("encoder_memory", None)

    def _tail(self, sb, fork: bool):
        # the tail runs beside another clip's head, which pays for the tail's CU time and not for its launch count: the
        # query chain keeps K7's small workgroups here (hot_ops.row_chain_fusion)
        import os
        prev, hot_ops.row_chain_fusion = hot_ops.row_chain_fusion, os.environ.get("SOC_TAIL_ROW_FUSION", "0") == "1"
        if os.environ.get("SOC_TAIL_NO_FORK", "0") == "1":
            fork = False
        try:
            out = self.model.forward_tail(sb, self.targets, fork=fork)
        finally:
            hot_ops.row_chain_fusion = prev
        idx, masks = P.select_trajectory(out)
        CP.pack_record(self.record, idx, out["pred_cls"][:, 0, :, 0], masks)

    _VARY = ("ctx", "feats0", "lang_last", "word_pad", "sentence")    # what the head hands to the tail per clip

    @staticmethod
    def _like(t):
        new = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
        new.copy_(t)
        return new

    def _clone(self, st):
        new = dict(st)
        for k in self._VARY:
            v = st[k]
            if k == "ctx":
                new[k] = (self._like(v[0]),) + tuple(v[1:])
            elif isinstance(v, (list, tuple)):
                new[k] = [self._like(t) for t in v]
            else:
                new[k] = self._like(v)
        return new

    def _store(self, st, dst):
        pairs = []

        def put(d, t):
            if d.data_ptr() != t.data_ptr():             # else: already produced in place (_placed_head)
                pairs.append((d, t))

        for k in self._VARY:
            v = st[k]
            if k == "ctx":
                put(dst[k][0], v[0])
            elif isinstance(v, (list, tuple)):
                for d, t in zip(dst[k], v):
                    put(d, t)
            else:
                put(dst[k], v)
        # what is left are a few small tensors (the words' features, their padding mask, the sentence feature): one multi-tensor
        # launch per dtype instead of a copy launch each at the very end of the head
        groups = {}
        for d, t in pairs:
            if d.dtype == t.dtype and d.shape == t.shape:
                groups.setdefault(d.dtype, []).append((d, t))
            else:
                d.copy_(t)
        for group in groups.values():
            torch._foreach_copy_([d for d, _ in group], [t for _, t in group])

    # -- driving -----------------------------------------------------------------------------------------
    def run(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None, attn: Optional[torch.Tensor] = None):
        """Submit `clip`.  Returns self.record if this replay finished the clip submitted one call earlier
        (copy it out before the next call), else None."""
        self.stage_inputs(clip, ids, attn)
        return self.replay()

    stage_inputs = ClipGraph.stage_inputs

    def replay(self):
        self.steady[self._n % 2].replay()
        self._n += 1
        return self.record if self._n >= self.DEPTH else None

    def flush(self):
        """Drain the pipeline: [clone of the record of the clip still in flight] (empty if none)."""
        if self._n == 0:
            return []
        self.drain[(self._n - 1) % 2].replay()       # the last head wrote sb[(n-1) % 2]
        self._n = 0
        return [self.record.clone()]
