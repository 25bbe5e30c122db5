"""N>1 path on CPU: two gloo ranks shard a clip stream, all_gather fixed-size records once, and
every rank reassembles the stream in the original order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neurips2023_soc_amd import clip_parallel as CP

T, Q, H, W = 2, 3, 4, 5


def fake_result(clip_id: int):
    g = torch.Generator().manual_seed(clip_id)
    return (torch.tensor(clip_id % Q), torch.randn(T, Q, generator=g), torch.randn(T, H, W, generator=g))


def _worker(rank, world, port, n_clips, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = CP.shard_clips(n_clips, rank, world)
    n_local = -(-n_clips // world)
    local = torch.zeros(n_local, CP.record_size(T, Q, H, W))
    for slot, cid in enumerate(mine):
        CP.pack_record(local[slot], *fake_result(cid))
    allr = CP.interleave(CP.gather_results(local), n_clips)
    ok = True
    for cid in range(n_clips):
        q, cls, m = CP.unpack_record(allr[cid], T, Q, H, W)
        eq, ecls, em = fake_result(cid)
        ok &= q == int(eq) and torch.equal(cls, ecls) and torch.equal(m, em)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    n_clips = 5  # ragged: rank 0 gets 3 clips, rank 1 gets 2 (+1 padded slot)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def test_shard_is_a_partition():
    for n, w in ((0, 2), (1, 2), (7, 3), (8, 8), (202, 8)):
        parts = [CP.shard_clips(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_gather_is_identity():
    x = torch.arange(6.0).view(2, 3)
    assert torch.equal(CP.gather_results(x), x[None])
    assert torch.equal(CP.interleave(x[None], 2), x)
