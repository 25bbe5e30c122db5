"""Clip-parallel inference over the GPUs of one node.

Each (video, expression) clip is an independent forward, so the stream is sharded over ranks with
no data-path collective -- the reference does the same with one mp.Process per GPU and a static
split of the video list (infer_refytb.py:92-106).  The only exchange is ONE all_gather of
fixed-size per-clip result records at the end (RCCL over xGMI when the backend is "nccl";
"gloo" for the CPU tests): [query index, pred_cls scores (T*Q), selected mask logits (T*h*w)].
"""
from __future__ import annotations

import os
import socket
import subprocess
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

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


# ---------------------------------------------------------------------------------------------------
# process fan-out and the measured rank loop (shared by bench.py, infer.py and the gloo tests)
# ---------------------------------------------------------------------------------------------------

def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment() -> None:
    """Process environment every rank needs, whichever way it was started (spawn_ranks or torch.distributed.run): the host
    driver of this pool only supports dmabuf IPC, and RCCL's / torch's cross-process GPU buffer sharing fails with
    `hipIpcGetMemHandle: invalid argument` without HSA_ENABLE_IPC_MODE_LEGACY=0.  The HIP runtime reads the variable when it
    initialises, so this runs first thing in main(), before any GPU call; an explicit setting by the caller wins."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def granted_cpus(cgroup_root: str = "/sys/fs/cgroup") -> int:
    """CPUs this process may actually use: the cgroup CPU quota (v2 cpu.max, v1 cfs quota / period) rounded up, capped by
    the affinity mask."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open(os.path.join(cgroup_root, "cpu.max")) as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = -(-int(q) // int(per))
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as f:
                q = int(f.read())
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as f:
                per = int(f.read())
            if q > 0:
                quota = -(-q // per)
        except (OSError, ValueError):
            pass
    return max(1, min(avail, quota) if quota else avail)


def rank_cpu_share(rank: int, world: int, allowed: Sequence[int], granted: int) -> List[int]:
    """The CPUs rank `rank` of `world` keeps to: `granted // world` of them (at least one), consecutive in the sorted
    affinity mask, disjoint between ranks while the mask is long enough.  Eight ranks behind a 16-CPU quota get two CPUs
    each instead of eight capture threads and eight intra-op pools competing for the sixteen."""
    cpus = sorted(allowed)
    k = max(1, min(granted, len(cpus)) // max(world, 1))
    start = (rank * k) % len(cpus)
    return [cpus[(start + i) % len(cpus)] for i in range(min(k, len(cpus)))]


def pin_rank_cpus(rank: int, world: int) -> List[int]:
    """Pin this process to its share (world > 1 only) and size torch's intra-op pool to it.  Returns the CPUs kept.
    The share is a share of THIS NODE's CPUs: under a multi-node launch the node's ranks are LOCAL_RANK of LOCAL_WORLD_SIZE
    (torch.distributed.run sets both), not RANK of WORLD_SIZE.  The native library is loaded -- and, if stale, rebuilt with
    the node's full CPU set -- before the pin, not after it with a two-CPU compiler pool while the other ranks wait."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(local_world, 1)))
    if not (0 < local_world <= world and 0 <= local_rank < local_world):
        local_rank, local_world = rank, world
    try:
        from . import _lib
        _lib.load()
    except Exception:           # noqa: BLE001 -- a CPU-only stub run has no library to load; the product path fails later, loudly
        pass
    share = rank_cpu_share(local_rank, local_world, os.sched_getaffinity(0), granted_cpus())
    try:
        os.sched_setaffinity(0, share)
        torch.set_num_threads(len(share))
    except OSError:
        pass
    return share


def launched_as_rank() -> bool:
    """True when an outer launcher (torch.distributed.run, spawn_ranks) already made this process a rank."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(n: int, argv: Sequence[str], extra_env: Optional[Dict[str, str]] = None,
                timeout: Optional[float] = None) -> int:
    """Start ``n`` fresh processes running ``argv``, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set -- the reference's ``mp.Process`` fan-out from ``-ng N``
    (infer_refytb.py:84-109).  The caller must not have touched the GPU yet and is never replaced (no exec):
    it only waits.  stdout / stderr are inherited, so rank 0's JSON line is the parent's.  If a rank dies the
    others are terminated (by PID) instead of hanging in a collective.  Returns the worst exit code."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # as rank_environment(): set for the child before it starts
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(argv), env=env))
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                worst = worst or code
                for q in live:             # a dead rank would leave the others waiting in the all_gather
                    q.terminate()
        if live and deadline is not None and time.monotonic() > deadline:
            for q in live:
                q.kill()
            worst = worst or 124
        if live:
            time.sleep(0.05)
    return worst


def init_rank(device_type: str = "cuda", expect_world: Optional[int] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the launcher's environment; joins the process group when there is one
    ("nccl" = RCCL over xGMI for GPU ranks, "gloo" for the CPU tests).  ``expect_world`` (the ``--gpus`` /
    ``-ng`` value) must match WORLD_SIZE: a mismatch fails instead of silently measuring fewer ranks."""
    rank_environment()      # no-op when main() already did it; a caller that skipped it gets it before the first GPU call here
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if expect_world is not None and expect_world != world:
        raise RuntimeError(f"--gpus {expect_world} but WORLD_SIZE={world}: launch one rank per requested GPU")
    if world > 1 or "MASTER_PORT" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    elif device_type == "cuda":
        torch.cuda.set_device(local)
    return rank, local, world


def _fence(device) -> None:
    cuda = device is not None and torch.device(device).type == "cuda"
    if cuda:
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if cuda:
        torch.cuda.synchronize(device)


def timed_sharded_run(run_local: Callable[[torch.Tensor], None], results: torch.Tensor, device=None) -> Dict:
    """The measured body of a clip-parallel job: barrier + sync, this rank's clips (``run_local(results)`` fills
    the [n_local, R] record buffer), the ONE result all_gather, barrier + sync; then the max over ranks of the
    elapsed time.  Returns {"gathered": [world, n_local, R], "seconds": max over ranks, "ranks_seen": [...]}.
    ``ranks_seen`` comes out of the collective itself (each rank contributes its id next to its time), so a job
    that lost a rank cannot report the full world."""
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    _fence(device)
    t0 = time.perf_counter()
    run_local(results)
    gathered = gather_results(results)
    _fence(device)
    dt = time.perf_counter() - t0
    mine = torch.tensor([[float(rank), dt]], dtype=torch.float64, device=results.device)
    per_rank = gather_results(mine)[:, 0]                    # [world, 2]
    return {"gathered": gathered, "seconds": float(per_rank[:, 1].max()),
            "ranks_seen": sorted(int(r) for r in per_rank[:, 0].tolist()),
            "seconds_per_rank": [float(v) for v in per_rank[:, 1].tolist()]}


def _selftest(argv=None) -> int:
    """``python -m neurips2023_soc_amd.clip_parallel --gpus N [--device cpu]``: the fan-out + rank loop with a stub
    clip step (record i = f(clip id)), no model.  Used by the CPU tests (gloo) and as a quick RCCL check."""
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--clips", type=int, default=5)
    ap.add_argument("--device", default="cpu", choices=["cpu", "cuda"])
    ap.add_argument("--fail-rank", type=int, default=-1, help="this rank exits 3 before the collective (test)")
    a = ap.parse_args(argv)
    if a.gpus > 1 and not launched_as_rank():
        import sys
        return spawn_ranks(a.gpus, [sys.executable, "-m", "neurips2023_soc_amd.clip_parallel", *(argv or sys.argv[1:])])
    rank, local, world = init_rank(a.device, expect_world=a.gpus)
    dev = torch.device("cuda", local) if a.device == "cuda" else torch.device("cpu")
    if rank == a.fail_rank:
        os._exit(3)
    mine = shard_clips(a.clips, rank, world)
    results = torch.zeros(-(-a.clips // world), 4, device=dev)

    def run_local(out):
        for slot, cid in enumerate(mine):
            out[slot] = torch.tensor([cid, cid * cid, rank, 1.0], device=dev)

    res = timed_sharded_run(run_local, results, dev)
    allr = interleave(res["gathered"], a.clips).cpu()
    ok = all(allr[i].tolist() == [i, i * i, i % world, 1.0] for i in range(a.clips))
    if rank == 0:
        print(json.dumps({"n_gpus": world, "ranks_seen": res["ranks_seen"], "ok": bool(ok), "clips": a.clips}),
              flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(_selftest())
