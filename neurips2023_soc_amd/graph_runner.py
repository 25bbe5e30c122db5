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
        with torch.cuda.graph(self.graph):
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
        self.clip.copy_(clip.view(self.clip.shape), non_blocking=True)
        if ids is not None:
            self.ids.copy_(ids.view(self.ids.shape), non_blocking=True)
            if attn is None:
                self.attn.fill_(1)
            else:
                self.attn.copy_(attn.view(self.attn.shape), non_blocking=True)
        self.graph.replay()
        return self.out


class PipelinedClipGraph:
    """Two clips in flight inside ONE stream of graph replays: graph k runs the TAIL of the previous clip (FPN,
    query decoder, VOC, heads, mask head, selection -- ~150 short, latency-bound launches that leave most of the GPU
    idle) on a side branch while the HEAD of the next clip (text ‖ Video-Swin, fusion, deformable encoder -- chip-
    filling kernels) runs on the main branch.  The head hands over through a double-buffered static state (encoder
    memory, stride-4 backbone map, text features), so graphs 0 / 1 alternate and are never replayed concurrently
    (concurrent replays hang on this stack, tools/experiments/README.md).  Results are those of ClipGraph: same
    kernels, same order inside each clip; only which clip's kernels share the GPU changes.

        for clip in clips:  prev = g.run(clip, ids)      # -> packed record of the PREVIOUS clip (None at first)
        last = g.flush()                                  # -> record of the last clip
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
        assert hot_ops._prof is None, "do not capture while kernel profiling is on"
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                st = self._head()
                self._tail(st)
            # static, double-buffered hand-over state (only what changes from clip to clip; geometry constants
            # inside ctx are shared)
            self.state = [self._clone_state(st), self._clone_state(st)]
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._pipe = torch.cuda.Stream(device=dev, priority=-1)   # the short tail kernels jump the queue
        self.graphs, self.tails = [], []
        for k in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cur = torch.cuda.current_stream(dev)
                self._pipe.wait_stream(cur)
                with torch.cuda.stream(self._pipe):          # previous clip's tail, beside ...
                    self._tail(self.state[1 - k], fork=False)
                self._store_state(self._head(), self.state[k])   # ... this clip's head
                cur.wait_stream(self._pipe)
            self.graphs.append(g)
        for k in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._tail(self.state[k])
            self.tails.append(g)
        self._k = 0            # graph of the next run()
        self._pending = False  # a head has run whose tail has not

    # -- pieces ------------------------------------------------------------------------------------------
    def _head(self):
        samples = NestedTensor(self.clip, self.pad, unpadded=True)
        return self.model.forward_head(samples, None, {"input_ids": self.ids, "attention_mask": self.attn})

    def _tail(self, state, fork: bool = True):
        out = self.model.forward_tail(state, self.targets, fork=fork)
        idx, masks = P.select_trajectory(out)
        CP.pack_record(self.record, idx, out["pred_cls"][:, 0, :, 0], masks)

    _VARYING = ("feats0", "lang_last", "word_pad", "sentence")

    def _clone_state(self, st):
        new = dict(st)
        for k in self._VARYING:
            new[k] = torch.empty_strided(st[k].shape, st[k].stride(), dtype=st[k].dtype, device=st[k].device)
            new[k].copy_(st[k])
        new["ctx"] = (st["ctx"][0].clone(),) + tuple(st["ctx"][1:])
        return new

    def _store_state(self, st, dst):
        for k in self._VARYING:
            dst[k].copy_(st[k])
        dst["ctx"][0].copy_(st["ctx"][0])

    # -- driving -----------------------------------------------------------------------------------------
    def run(self, clip: torch.Tensor, ids: Optional[torch.Tensor] = None, attn: Optional[torch.Tensor] = None):
        """Enqueue `clip`; returns self.record holding the PREVIOUS clip's result once this replay has run
        (None for the very first clip).  Copy the record out before the next run()."""
        self.clip.copy_(clip.view(self.clip.shape), non_blocking=True)
        if ids is not None:
            self.ids.copy_(ids.view(self.ids.shape), non_blocking=True)
            if attn is None:
                self.attn.fill_(1)
            else:
                self.attn.copy_(attn.view(self.attn.shape), non_blocking=True)
        had = self._pending
        self.graphs[self._k].replay()       # tail(state[1-k]) ‖ head -> state[k]
        self._k ^= 1
        self._pending = True
        return self.record if had else None

    def flush(self):
        """Run the tail of the last enqueued clip; returns self.record (or None if nothing is pending)."""
        if not self._pending:
            return None
        self.tails[self._k ^ 1].replay()
        self._pending = False
        return self.record
