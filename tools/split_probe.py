#!/usr/bin/env python
"""K20 (soc_linear_split_f32) against the f32 library GEMM on the layer shapes of the BASELINE config:
time per launch (HIP events, back-to-back replays) and error against an f64 reference on the GPU.

    python tools/split_probe.py [--tiles 0,1,2,3,4] [--shapes swin|enc|all] [--reps 20]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurips2023_soc_amd import gemm_tuning, hot_ops  # noqa: E402

SWIN = [  # (name, M, K, N, ln, act, residual)
    ("s0.qkv", 115200, 96, 288, True, "none", False), ("s0.proj", 115200, 96, 96, False, "none", True),
    ("s0.fc1", 115200, 96, 384, True, "gelu", False), ("s0.fc2", 115200, 384, 96, False, "none", True),
    ("s1.qkv", 28800, 192, 576, True, "none", False), ("s1.proj", 28800, 192, 192, False, "none", True),
    ("s1.fc1", 28800, 192, 768, True, "gelu", False), ("s1.fc2", 28800, 768, 192, False, "none", True),
    ("s2.qkv", 7360, 384, 1152, True, "none", False), ("s2.proj", 7360, 384, 384, False, "none", True),
    ("s2.fc1", 7360, 384, 1536, True, "gelu", False), ("s2.fc2", 7360, 1536, 384, False, "none", True),
    ("s3.qkv", 1920, 768, 2304, True, "none", False), ("s3.proj", 1920, 768, 768, False, "none", True),
    ("s3.fc1", 1920, 768, 3072, True, "gelu", False), ("s3.fc2", 1920, 3072, 768, False, "none", True),
    ("merge1", 28800, 384, 192, False, "none", False), ("merge2", 7360, 768, 384, False, "none", False),
    ("merge3", 1920, 1536, 768, False, "none", False),
]
ENC = [
    ("enc.value", 38560, 256, 256, False, "none", False), ("enc.offw", 38560, 256, 384, False, "none", False),
    ("enc.out", 38560, 256, 256, False, "none", True), ("enc.ffn1", 38560, 256, 2048, False, "relu", False),
    ("enc.ffn2", 38560, 2048, 256, False, "none", True), ("proj0", 28800, 192, 256, False, "none", False),
    ("vlf.q", 28800, 256, 256, False, "none", False),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(20_000_000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


def stamps(shapes, tiles):
    """-DSOC_K20_STAMPS build (never shipped): every wave sums s_memtime differences per phase of the pipeline loop.
    Shares, not times: the stamps themselves serialise what the product build overlaps."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = "/tmp/libk20_stamps.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DSOC_K20_STAMPS",
                    "-I", os.path.join(root, "include"), "-I", os.path.join(root, "neurips2023_soc_amd/csrc"),
                    "-o", so, os.path.join(root, "neurips2023_soc_amd/csrc/linear_split.hip")], check=True)
    lib = C.CDLL(so)
    lib.soc_linear_split_packed_bytes.restype = C.c_size_t
    names = ["prologue", "commit", "issue", "compute", "tile_barrier", "epilogue", "step_barrier", "lifetime"]
    dev = torch.device("cuda")
    for name, M, K, N, ln, act, res in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        st = hot_ops.row_stats(x, 1e-5) if ln else None
        cs = w.double().sum(1).float().contiguous() if ln else None
        packed = torch.empty(lib.soc_linear_split_packed_bytes(N, K), dtype=torch.uint8, device=dev)
        assert lib.soc_linear_split_pack_f32(C.c_void_p(w.data_ptr()), C.c_void_p(packed.data_ptr()), N, K, None) == 0
        out = torch.empty(M, N, device=dev)
        for tile in tiles:
            dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
            lib.soc_debug_set_buffer_k20(C.c_void_p(dbg.data_ptr()))
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
            for _ in range(3):
                rc = lib.soc_linear_split_f32(p(x), None, p(st), p(cs), p(packed), p(b), p(r), None, p(out), None, 0, C.c_long(M), N, K,
                                              {"none": 0, "relu": 1, "gelu": 2}[act], tile, None)
                assert rc == 0, rc
            torch.cuda.synchronize()
            d = dbg.view(256, 8, 8).double()
            live = d[:, :, 7] > 0
            tot = d[:, :, 7][live].mean()
            shares = {names[i]: round(float(d[:, :, i][live].mean() / tot), 3) for i in range(7)}
            print(json.dumps({"layer": name, "tile": tile, "lifetime_cycles": round(float(tot)), **shares}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated layer names")
    ap.add_argument("--tiles", default="auto,0,1,2,3,4")
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--plain", action="store_true", help="no LayerNorm / activation / residual: the bare GEMM")
    a = ap.parse_args()
    gemm_tuning.enable_tuned_gemms()
    dev = torch.device("cuda")
    shapes = {"swin": SWIN, "enc": ENC, "all": SWIN + ENC}[a.shapes]
    if a.only:
        shapes = [sh for sh in shapes if sh[0] in a.only.split(",")]
    if a.stamps:
        return stamps(shapes, [int(t) for t in a.tiles.split(",") if t != "auto"])
    rows = []
    for name, M, K, N, ln, act, res in shapes:
        if a.plain:
            ln, act, res = False, "none", False
        g = torch.Generator(device="cpu").manual_seed(M + K + N)
        x = (torch.randn(M, K, generator=g) * 1.3 + 0.1).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        gam, bet = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.1).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if res else None
        lnp = (gam, bet, 1e-5) if ln else None

        def lib():
            h = F.layer_norm(x, (K,), gam, bet, 1e-5) if ln else x
            y = F.linear(h, w, b)
            y = F.gelu(y) if act == "gelu" else (F.relu(y) if act == "relu" else y)
            return y + r if res else y

        # f64 reference on a row sample (the f64 GEMM of the whole layer is the slow part)
        idx = torch.arange(0, M, max(1, M // 4096), device=dev)
        xd = x[idx].double()
        hd = F.layer_norm(xd, (K,), gam.double(), bet.double(), 1e-5) if ln else xd
        yd = F.linear(hd, w.double(), b.double())
        yd = F.gelu(yd) if act == "gelu" else (F.relu(yd) if act == "relu" else yd)
        if res:
            yd = yd + r[idx].double()
        scale = float(yd.abs().max())
        e_lib = float((lib()[idx].double() - yd).abs().max())
        t_lib = timeit(lib, a.reps)
        entry = {"layer": name, "M": M, "K": K, "N": N, "ln": ln, "act": act, "res": res, "lib_us": round(t_lib, 1),
                 "lib_tflops": round(2e-6 * M * N * K / t_lib, 1), "lib_err": e_lib, "scale": scale, "split": {}}
        for tl in a.tiles.split(","):
            tile = None if tl == "auto" else int(tl)
            try:
                fn = lambda: hot_ops.linear_split(x, w, b, lnp, r, act, tile=tile)   # noqa: E731
                y = fn()
                torch.cuda.synchronize()
            except Exception as exc:   # noqa: BLE001
                entry["split"][tl] = {"error": str(exc)[:80]}
                continue
            err = float((y[idx].double() - yd).abs().max())
            t = timeit(fn, a.reps)
            entry["split"][tl] = {"us": round(t, 1), "tflops_f32_equiv": round(2e-6 * M * N * K / t, 1), "err": err,
                                  "cfg": hot_ops.split_tile_for(M, N, K) if tile is None else tile}
        best = min((v["us"], k) for k, v in entry["split"].items() if "us" in v)
        entry["best"] = {"tile": best[1], "us": best[0], "speedup_vs_lib": round(t_lib / best[0], 2)}
        rows.append(entry)
        print(json.dumps(entry), flush=True)
    tot_lib = sum(r["lib_us"] for r in rows)
    tot_best = sum(r["best"]["us"] for r in rows)
    print(json.dumps({"sum_lib_us": tot_lib, "sum_best_us": tot_best}))


if __name__ == "__main__":
    main()
