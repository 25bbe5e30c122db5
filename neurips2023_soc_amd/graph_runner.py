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


# Diagnostic switches read from the environment (gathered here; bench.py records every one that is set).  The
# first four are read while a graph is CAPTURED and change what is captured; none of them changes results beyond f32 rounding.
SWITCHES = {
    "SOC_PIPELINE": "one-graph | two-stream: which software pipeline pipeline_class() returns (default one-graph)",
    "SOC_NO_PLACE": "1: the head's two large hand-over tensors are copied into the static state instead of produced there",
    "SOC_TAIL_ROW_FUSION": "1: the pipelined tail keeps K16's fused row chains (default: K7's small workgroups beside a head)",
    "SOC_TAIL_NO_FORK": "1: the tail never forks FPN || query chain, not even in the drain graph",
    "SOC_TAIL_PRIORITY": "0: the tail branch of the one-graph pipeline is captured at default stream priority (default: high)",
    "SOC_GROUP_SEQ_FIRST": "1: a launch group keeps the reference's '(t h w) b c' / [T,B,...] layouts (a permute copy of every level each way)",
    "SOC_K24_SPANS": "0: K24 plans one column range per workgroup (the cut of round 4) instead of spans of ranges",
    "SOC_MATMUL": "split | f32: arithmetic a model is built with (hot_ops.DEFAULT_MATMUL_MODE; SOC.matmul_mode overrides per model)",
    "SOC_SPLIT_OFF": "comma list of call sites / kernels forced back to the f32 path (k1, k13, k24, mlp, swin, gelu, ...)",
    "SOC_PNG": "pillow: the drivers write PNGs through Pillow instead of libsoc_host.so (SOC_PNG_LEVEL: its compress_level)",
}


def switches_set() -> Dict[str, str]:
    """The diagnostic switches that are set in this process's environment (empty in a default run)."""
    import os
    return {k: os.environ[k] for k in list(SWITCHES) + ["SOC_PNG_LEVEL"] if os.environ.get(k) not in (None, "")}


# Stream-capture error mode.  The drivers decode and upload the NEXT clip on a background thread (clip_io.VideoClipCache)
# while this thread captures a graph: in HIP's default "global" mode a hipMalloc / hipHostMalloc issued by ANY thread during
# the capture invalidates it (hipErrorStreamCaptureInvalidated -- seen as a rare failure of the Ref-YouTube-VOS driver test in
# the first session on a fresh box, when the allocator pools are still cold).  Only this thread's calls matter for the
# capture, so the check is thread-local.
CAPTURE_MODE = "thread_local"


class ClipGraph:
    CLIPS = 1              # clips per replay

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
        assert not hot_ops.op_profile.active(), "do not capture while kernel profiling is on"
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
        hot_ops.select_pack(out["pred_cls"], out["pred_masks"], self.record[None])      # K26 (was select_trajectory + pack_record)
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
    CLIPS = 1              # clips per replay (PairPipelinedClipGraph: 2)

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
        assert not hot_ops.op_profile.active(), "do not capture while kernel profiling is on"
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
        # the first replay after a flush has no tail to run: head only, into sb[0] (a steady graph there would run the tail of
        # stale state beside it)
        self.first = capture(lambda: self._store(self._placed_head(self.sb[0]), self.sb[0]))
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
        hot_ops.select_pack(out["pred_cls"], out["pred_masks"], self.record[None])      # K26

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
        (self.first if self._n == 0 else self.steady[self._n % 2]).replay()
        self._n += 1
        return self.record if self._n >= self.DEPTH else None

    def flush(self):
        """Drain the pipeline: [clone of the record of the clip still in flight] (empty if none)."""
        if self._n == 0:
            return []
        self.drain[(self._n - 1) % 2].replay()       # the last head wrote sb[(n-1) % 2]
        self._n = 0
        return [self.record.clone()]


class TwoStreamClipGraph(PipelinedClipGraph):
    """The software pipeline across clips as SEPARATE graphs on TWO streams (round 5).

        main stream:  Video-Swin(i) ------------------------> fusion + deformable encoder(i)  | Video-Swin(i+1) ...
        aux stream:   text encoder(i) -> tail(i-1: FPN, query decoder, VOC, heads, mask head)  | text encoder(i+1) ...

    PipelinedClipGraph keeps both clips inside one graph: the tail on a side branch, the text encoder on another one beside
    Video-Swin -- two sources of short launches, each of which can take a CU away from a one-workgroup-per-CU kernel of the head
    for its whole duration.  Here every short, latency-bound launch of a period (text encoder: ~150, tail: ~150) sits on ONE
    auxiliary stream, one behind the other, so at most one of them is beside the head at any time, and the head itself is a
    single-branch graph.  Measured (tools/experiments/partition_probe.py, pipeline_ab.py; always inside one process, the boxes
    differ by 6 %): the schedule above 6.07-6.10 ms per clip on the box where the pieces alone take 3.37 + 2.19 (main) and 0.81 +
    0.88 (aux); 6.17 with the fusion levels forked, 6.7 with the text encoder on a third stream, 6.96 with the tail in front of
    the text encoder.  Against the one-graph pipeline in the same process it is a tie (6.43-6.51 both), so the idea that the
    head loses 0.6-0.7 ms to COLLISIONS with short launches is wrong: what the head loses is the CU time the tail and the text
    encoder really need.  Giving the auxiliary stream its own CUs (hipExtStreamCreateWithCUMask; the kernels size their grids
    from the stream's mask, soc_stream_cus) was measured too and is slower at every split tried (8 / 16 / 24 / 32 CUs: 14.3 /
    7.8 / 8.9 / 7.0 ms): Video-Swin alone takes 5.2 ms on 248 CUs against 3.4 on 256 -- the dispatcher deals workgroups to the
    shader engines evenly whatever the mask leaves of each.  Kept as an option (SOC_PIPELINE=two-stream, bench.py --pipeline):
    its four single-branch launches cost the host 0.25 ms per clip where the one multi-branch launch blocks it for 1.2-4.8 ms.

    Hand-over: Video-Swin(i) and the encoder write the double-buffered state sb[i % 2] (as in the parent); the text encoder
    writes tx[i % 2].  Cross-stream order is carried by events recorded / awaited around the graph launches (never inside a
    capture).  Same driving interface and the same records as the parent: run() returns the record of the clip submitted one
    call earlier, flush() drains.
    """

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
        assert not hot_ops.op_profile.active(), "do not capture while kernel profiling is on"
        self.main, self.aux = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        caller = torch.cuda.current_stream(dev)
        self.main.wait_stream(caller)
        with torch.cuda.stream(self.main):      # warm-up off the capture: lazy inits, algorithm finds, packed weight images
            for _ in range(max(warmup, 1)):
                tx = self._text()
                sb = self._fuse(self._video(), tx)
                self._tail(sb, fork=False)
            self.tx = [self._clone_text(tx), self._clone_text(tx)]
            self.sb = [self._clone(sb), self._clone(sb)]
        self.aux.wait_stream(self.main)
        caller.wait_stream(self.main)
        torch.cuda.synchronize(dev)

        def capture(body, stream):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode=CAPTURE_MODE):
                body()
            return g

        self._vs = [None, None]

        def video(k):       # Video-Swin of clip i, its stage-0 token map produced IN sb[k]
            self._vs[k] = self._placed(lambda: self._video(), self.sb[k], "swin0")

        def fuse(k):        # fusion + encoder of clip i: reads Video-Swin's static outputs and tx[k], fills sb[k]
            st = self._placed(lambda: self._fuse(self._vs[k], self.tx[k]), self.sb[k], "encoder_memory")
            self._store(st, self.sb[k])

        def text(k):
            self._store_text(self._text(), self.tx[k])

        self.g_video = [capture(lambda k=k: video(k), self.main) for k in (0, 1)]
        self.g_fuse = [capture(lambda k=k: fuse(k), self.main) for k in (0, 1)]
        self.g_text = [capture(lambda k=k: text(k), self.aux) for k in (0, 1)]
        self.g_tail = [capture(lambda k=k: self._tail(self.sb[k], fork=False), self.aux) for k in (0, 1)]
        torch.cuda.synchronize(dev)
        E = torch.cuda.Event
        self._ev_staged, self._ev_clip_free, self._ev_text = E(), E(), E()
        self._ev_head, self._ev_tail = [E(), E()], [E(), E()]
        self._tail_pending = [False, False]      # a tail reading sb[k] has been enqueued and not yet been waited for by the head
        self._inputs_busy = False
        self._n = 0

    # -- stages ------------------------------------------------------------------------------------------
    _TEXT = ("words", "word_pad", "text_pos", "sentence")

    def _text(self):
        return self.model.forward_text_state({"input_ids": self.ids, "attention_mask": self.attn}, self.device)

    def _video(self):
        return self.model.forward_video(NestedTensor(self.clip, self.pad, unpadded=True))

    def _fuse(self, vs, tx):
        return self.model.forward_fuse_encode({**vs, **tx}, fork=False)

    def _placed(self, body, dst, tag):
        """`body()` with one of its large outputs produced IN the static state (hot_ops.place_output) instead of copied there."""
        f0, mem = dst["feats0"], dst["ctx"][0]
        try:
            if tag == "swin0":
                n, c, h, w = f0.shape
                tok = f0.permute(0, 2, 3, 1)                 # '(b t) h w c': the layout the stage writes
                if tok.is_contiguous():
                    hot_ops.place_output("swin0", tok.view(n // self.T, self.T, h, w, c))
            elif mem.is_contiguous():
                hot_ops.place_output("encoder_memory", mem)
            return body()
        finally:
            hot_ops.place_output(tag, None)

    def _clone_text(self, tx):
        return {k: self._like(tx[k]) for k in self._TEXT}

    def _store_text(self, tx, dst):
        torch._foreach_copy_([dst[k] for k in self._TEXT if dst[k].dtype == torch.float32],
                             [tx[k] for k in self._TEXT if dst[k].dtype == torch.float32])
        for k in self._TEXT:
            if dst[k].dtype != torch.float32:
                dst[k].copy_(tx[k])

    # -- driving -----------------------------------------------------------------------------------------
    def stage_inputs(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None,
                     attn: Optional[torch.Tensor] = None) -> None:
        """As ClipGraph.stage_inputs, on the caller's stream -- which first waits until Video-Swin and the text encoder of the
        previous clip have read the static inputs."""
        if self._inputs_busy:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._ev_clip_free)
            cur.wait_event(self._ev_text)
        ClipGraph.stage_inputs(self, clip, ids, attn)

    def replay(self):
        dev, k = self.device, self._n % 2
        cur = torch.cuda.current_stream(dev)
        self._ev_staged.record(cur)
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(self._ev_staged)
            self.g_text[k].replay()                      # clip i: ids -> tx[k]  (tx[k] was last read by tail(i-2): this stream)
            self._ev_text.record(self.aux)
            if self._n >= 1:
                self.aux.wait_event(self._ev_head[1 - k])
                self.g_tail[1 - k].replay()              # clip i-1
                self._ev_tail[1 - k].record(self.aux)
                self._tail_pending[1 - k] = True
        with torch.cuda.stream(self.main):
            self.main.wait_event(self._ev_staged)
            if self._tail_pending[k]:                    # sb[k] is still being read by tail(i-2) until then
                self.main.wait_event(self._ev_tail[k])
                self._tail_pending[k] = False
            self.g_video[k].replay()
            self._ev_clip_free.record(self.main)
            self.main.wait_event(self._ev_text)
            self.g_fuse[k].replay()
            self._ev_head[k].record(self.main)
        self._inputs_busy = True
        self._n += 1
        if self._n >= self.DEPTH:
            cur.wait_event(self._ev_tail[1 - k])         # the caller reads self.record on its own stream
            return self.record
        return None

    def flush(self):
        """Drain the pipeline: [clone of the record of the clip still in flight] (empty if none)."""
        if self._n == 0:
            return []
        dev, k = self.device, (self._n - 1) % 2          # the last head wrote sb[k]
        cur = torch.cuda.current_stream(dev)
        self._ev_staged.record(cur)                      # the caller has copied the previous record out by now
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(self._ev_staged)
            self.aux.wait_event(self._ev_head[k])
            self.g_tail[k].replay()
            self._ev_tail[k].record(self.aux)
        cur.wait_event(self._ev_tail[k])
        cur.wait_event(self._ev_clip_free)
        self._tail_pending = [False, False]
        self._inputs_busy = False
        self._n = 0
        return [self.record.clone()]


def group_tail(model, state, targets_one, fork: bool, records) -> None:
    """The tail of a group of INDEPENDENT clips whose head ran as one batch: one batched pass over FPN, query decoder, heads and
    mask head with the VOC module per clip (SOC.forward_tail(voc_per_clip=True): VOC is the only place where the reference
    couples the clips of a batch), then selection + record packing per clip into records[b]."""
    B = state["B"]
    targets = [[frame[0]] * B for frame in targets_one]
    out = model.forward_tail(state, targets, fork=fork, voc_per_clip=True)
    hot_ops.select_pack(out["pred_cls"], out["pred_masks"], records[:B])        # K26: selection + packing of every clip, one launch


class PairPipelinedClipGraph(PipelinedClipGraph):
    """The one-graph software pipeline with a GROUP of independent clips per launch (round 5; CLIPS = 2 here, 4 in
    QuadPipelinedClipGraph).

    The head of SOC's forward -- Video-Swin, vision-language fusion, deformable encoder -- treats the clips of a batch
    independently (per token, per window, per frame), and at B = 1 its later stages have too few rows for 256 CUs (stage 2: 460
    row tiles, stage 3: 120; K23 / K24 / K1 run at 0.2-0.35 of their ceilings there, every launch pays its fixed ~10 us).  Over
    two clips the same launches do twice the work: head 5.59 -> 5.10 ms per clip (tools/experiments/batch2_probe.py).  The TAIL
    needs one exception: the reference's own B = 2 forward gives a clip other results than its B = 1 forward (0.32 on a logit
    scale of 6.6, run on the reference itself), and the inference drivers' results are the B = 1 ones (infer_refytb.py:206-227).
    The ONE place where the reference couples the clips of a batch is its VOC module (tools/experiments/batch2_voc_probe.py:
    with VOC run per clip a batched forward gives every clip its single-clip outputs to 5e-5), so the tail of a group is one
    batched pass -- FPN, query decoder, heads, mask head -- with VOC once per clip (group_tail, SOC.forward_tail(voc_per_clip=
    True)).  Graph k runs  head(group i)  beside  tail(group i-1).  Per clip the results equal the single-clip pipeline's to f32
    rounding (5e-5 on logits of 37, the run-to-run noise of the library kernels), whichever slot a clip sits in.

        g.stage_inputs(clip_a, ids_a, slot=0); g.stage_inputs(clip_b, ids_b, slot=1)
        recs = g.replay()             # [2, R] records of the pair submitted one call earlier, or None
        rest = g.flush()              # [clone of the records of the pair still in flight]
    """

    CLIPS = 2

    def __init__(self, model, T: int, H: int, W: int, L: int, device, warmup: int = 2):
        self.model, self.T, self.H, self.W, self.L = model, T, H, W, L
        dev = self.device = torch.device(device)
        n = self.CLIPS
        # stored clip-major: the backbone's '(b t)' flatten of the [T,B,...] samples is then a view, not an 88 MB copy per replay
        import os
        if os.environ.get("SOC_GROUP_SEQ_FIRST") == "1":
            self.clip, self.pad = torch.zeros(T, n, 3, H, W, device=dev), torch.zeros(T, n, H, W, dtype=torch.bool, device=dev)
        else:
            self.clip = torch.zeros(n, T, 3, H, W, device=dev).transpose(0, 1)
            self.pad = torch.zeros(n, T, H, W, dtype=torch.bool, device=dev).transpose(0, 1)
        self.ids = torch.ones(n, L, dtype=torch.long, device=dev)
        self.attn = torch.ones(n, L, dtype=torch.long, device=dev)
        self.targets = [[{"size": (H, W)}] for _ in range(T)]              # of ONE clip: the tail runs per clip
        hm, wm = -(-H // 4), -(-W // 4)
        self.record = torch.zeros(n, CP.record_size(T, model.num_queries, hm, wm), device=dev)
        self._finish_init(warmup)

    def _finish_init(self, warmup):
        """PipelinedClipGraph.__init__ behind the static buffers: warm-up, double-buffered state, the four captures."""
        dev = self.device
        assert not hot_ops.op_profile.active(), "do not capture while kernel profiling is on"
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                sb = self._head(fork=False)
                self._tail(sb, fork=False)
            self.sb = [self._clone(sb), self._clone(sb)]
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
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
                self._tail(self.sb[1 - k], fork=False)
            self._store(self._placed_head(self.sb[k]), self.sb[k])
            cur.wait_stream(self._pc)

        self.steady = [capture(lambda k=k: tail_beside_head(k)) for k in (0, 1)]
        self.drain = [capture(lambda k=k: self._tail(self.sb[k], fork=True)) for k in (0, 1)]
        self.first = capture(lambda: self._store(self._placed_head(self.sb[0]), self.sb[0]))      # head only: see the parent
        self._n = 0

    def _tail(self, sb, fork: bool):
        import os
        prev, hot_ops.row_chain_fusion = hot_ops.row_chain_fusion, os.environ.get("SOC_TAIL_ROW_FUSION", "0") == "1"
        if os.environ.get("SOC_TAIL_NO_FORK", "0") == "1":
            fork = False
        try:
            group_tail(self.model, sb, self.targets, fork, self.record)
        finally:
            hot_ops.row_chain_fusion = prev

    def stage_inputs(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None, attn: Optional[torch.Tensor] = None,
                     slot: int = 0) -> None:
        """Copy ONE clip [T,3,H,W] (+ its token ids [1,L] / [L]) into slot `slot` (0 .. CLIPS - 1) of the group, on the current
        stream."""
        self.clip[:, slot].copy_(clip.view(self.clip.shape[0], *self.clip.shape[2:]), non_blocking=True)
        if ids is not None:
            self.ids[slot].copy_(ids.view(-1), non_blocking=True)
            masked = self.__dict__.setdefault("_attn_masked", set())       # slots whose static mask is not all ones
            if attn is None:
                if slot in masked:
                    self.attn[slot].fill_(1)
                    masked.discard(slot)
            else:
                self.attn[slot].copy_(attn.view(-1), non_blocking=True)
                masked.add(slot)

    def run(self, clips, ids=None, attn=None):
        """clips: two [T,3,H,W] tensors; ids / attn: two [1,L] tensors each (or None)."""
        for b in range(self.CLIPS):
            self.stage_inputs(clips[b], None if ids is None else ids[b], None if attn is None else attn[b], slot=b)
        return self.replay()


class QuadPipelinedClipGraph(PairPipelinedClipGraph):
    """Four clips per launch group (bench.py takes the largest group that divides its clip count: default_pipeline()).  Same box, bench.py, 20 / 200 steps: 5.74-5.80 / 5.59 ms per clip against
    5.89-5.93 / 5.76 for pairs and 6.25-6.39 / 6.24 for one clip per launch; three clips per group lose a slot whenever the clip
    count is not a multiple of three (6.21 at 20 steps)."""
    CLIPS = 4


class OctPipelinedClipGraph(PairPipelinedClipGraph):
    """Eight clips per launch group: what bench.py times up to 360x640.  Same box, bench.py, 48-64 steps: Swin-T 4.84-4.85 ms per
    clip against 5.02 for fours; Swin-B 360p 10.25-10.27 against 10.50-10.55 (every clip still within 5e-5 of its single-clip
    logits, no flips).  A result arrives one replay (~40 ms) behind its group."""
    CLIPS = 8


PIPELINES = {"two-stream": TwoStreamClipGraph, "one-graph": PipelinedClipGraph, "pairs": PairPipelinedClipGraph,
             "quads": QuadPipelinedClipGraph, "octs": OctPipelinedClipGraph}


def group_pipeline_class(clips: int):
    """The launch-group pipeline for `clips` clips per replay (any count >= 2; 2 / 4 / 8 are the named classes)."""
    named = {c.CLIPS: c for c in (PairPipelinedClipGraph, QuadPipelinedClipGraph, OctPipelinedClipGraph)}
    if clips in named:
        return named[clips]
    if clips < 2:
        raise ValueError("a launch group has at least two clips")
    return type(f"Group{clips}PipelinedClipGraph", (PairPipelinedClipGraph,), {"CLIPS": int(clips)})


def pipeline_class(name: Optional[str] = None):
    """The software pipeline across clips the drivers and bench.py stream through: "one-graph" (default; SOC_PIPELINE
    overrides) or "two-stream".  Same GPU time per clip within the box-to-box spread (tools/experiments/pipeline_ab.py in
    one process: 6.44-6.51 ms both; bench.py on one of three boxes: two-stream 0.2 ms slower); the two-stream form needs
    0.25 ms of host time per clip for its four single-branch launches where the one multi-branch launch blocks 1.2-4.8 ms."""
    import os
    name = name or os.environ.get("SOC_PIPELINE", "one-graph")
    if name.startswith("group") and name[5:].isdigit():          # "group10": ten clips per launch group
        return group_pipeline_class(int(name[5:]))
    return PIPELINES[name]
