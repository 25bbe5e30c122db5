"""Clip-parallel inference over the GPUs of one node.

Each (video, expression) clip is an independent forward, so the stream is sharded over ranks with
no data-path collective -- the reference does the same with one mp.Process per GPU and a static
split of the video list (infer_refytb.py:92-106).  The only exchange is ONE all_gather of
fixed-size per-clip result records at the end (RCCL over xGMI when the backend is "nccl";
"gloo" for the CPU tests): [query index, pred_cls scores (T*Q), selected mask logits (T*h*w)].
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_clips(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: rank r takes clips i with i % world == r."""
    return list(range(rank, n_clips, world))


def record_size(T: int, Q: int, h: int, w: int) -> int:
    return 1 + T * Q + T * h * w


def pack_record(dst: torch.Tensor, query_idx: torch.Tensor, pred_cls: torch.Tensor,
                mask_logits: torch.Tensor) -> None:
    """Write one clip's result into the 1-D float32 record ``dst`` (no host sync)."""
    n_cls = pred_cls.numel()
    dst[0] = query_idx.to(torch.float32)
    dst[1:1 + n_cls] = pred_cls.reshape(-1)
    dst[1 + n_cls:] = mask_logits.reshape(-1)


def unpack_record(rec: torch.Tensor, T: int, Q: int, h: int, w: int) -> Tuple[int, torch.Tensor, torch.Tensor]:
    q = int(rec[0].item())
    return q, rec[1:1 + T * Q].view(T, Q), rec[1 + T * Q:].view(T, h, w)


def gather_results(local: torch.Tensor) -> torch.Tensor:
    """local [n_local, R] float32 -> [world, n_local, R] on every rank (one collective).
    Every rank must pass the same n_local (pad the last shard)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local[None]
    world = dist.get_world_size()
    out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))  # concat layout
    dist.all_gather_into_tensor(out, local.contiguous())
    return out.view((world,) + tuple(local.shape))


def interleave(gathered: torch.Tensor, n_clips: int) -> torch.Tensor:
    """[world, n_local, R] -> [n_clips, R] in original clip order (inverse of shard_clips)."""
    world, n_local, R = gathered.shape
    return gathered.transpose(0, 1).reshape(world * n_local, R)[:n_clips]
