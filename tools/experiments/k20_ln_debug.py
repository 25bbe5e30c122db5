#!/usr/bin/env python
"""Debug: K20 with the LayerNorm prologue is wrong on the 128x256 / 256x128 tiles only.  Where are the wrong elements,
and does it depend on the LDS footprint (gamma / beta sit above 144 KB there)?"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import hot_ops  # noqa: E402


def build(flags, tag):
    so = f"/tmp/libk20_{tag}.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags,
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "neurips2023_soc_amd/csrc"),
                    "-o", so, os.path.join(ROOT, "neurips2023_soc_amd/csrc/linear_split.hip")], check=True)
    lib = C.CDLL(so)
    lib.soc_linear_split_packed_bytes.restype = C.c_size_t
    return lib


def run(lib, x, w, b, stats, gam, bet, tile):
    M, K = x.shape
    N = w.shape[0]
    packed = torch.empty(lib.soc_linear_split_packed_bytes(N, K), dtype=torch.uint8, device=x.device)
    assert lib.soc_linear_split_pack_f32(C.c_void_p(w.data_ptr()), C.c_void_p(packed.data_ptr()), N, K, None) == 0
    out = torch.full((M, N), 777.0, device=x.device)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    if gam is not None:      # fold the affine part (as hot_ops.split_pack does)
        b = (b.double() + w.double() @ bet.double()).float()
        w = (w * gam[None]).contiguous()
        assert lib.soc_linear_split_pack_f32(C.c_void_p(w.data_ptr()), C.c_void_p(packed.data_ptr()), N, K, None) == 0
    cs = w.double().sum(1).float().contiguous() if stats is not None else None
    rc = lib.soc_linear_split_f32(p(x), None, p(stats), p(cs), p(packed), p(b), None, None, p(out), None, 0, C.c_long(M), N, K,
                                  0, tile, None)
    torch.cuda.synchronize()
    return rc, out


dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
M, K, N = 512, 96, 512
x = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
b = torch.zeros(N, device=dev)
gam = (torch.rand(K, generator=g) + 0.5).to(dev)
bet = (torch.randn(K, generator=g) * 0.1).to(dev)
stats = hot_ops.row_stats(x, 1e-5)
ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (K,), gam, bet, 1e-5), w, b)
ident = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], 1).contiguous()
ref_ident = torch.nn.functional.linear(x * gam + bet, w, b)
plain = torch.nn.functional.linear(x, w, b)
for tag, flags in (("std", []),):
    lib = build(flags, tag)
    for tile in (0, 1, 2, 3):
        e0 = []
        for rep in range(200):
            rc0, out0 = run(lib, x, w, b, None, None, None, tile)
            e0.append(round(float((out0 - plain).abs().max()), 3))
        print(tag, "tile", tile, "no-LN kernel on the same data, 200 runs: bad", sum(e > 1e-3 for e in e0), "max", max(e0))
        eln = []
        for rep in range(200):
            rc, out = run(lib, x, w, b, stats, gam, bet, tile)
            eln.append(round(float((out - ref).abs().max()), 3))
        print(tag, "tile", tile, "LN kernel, 200 runs: bad", sum(e > 1e-3 for e in eln), "max", max(eln))
        rc, out = run(lib, x, w, b, stats, gam, bet, tile)
        err = (out - ref).abs()
        bad = err > 1e-3
        rc2, out2 = run(lib, x, w, b, ident, gam, bet, tile)
        err2 = (out2 - ref_ident).abs()
        rc3, out3 = run(lib, x, w, b, ident, torch.ones_like(gam), torch.zeros_like(bet), tile)
        err3 = (out3 - torch.nn.functional.linear(x, w, b)).abs()
        print(tag, "tile", tile, "rc", rc, "max err", float(err.max()), "bad frac", float(bad.float().mean()),
              "| untouched(777)", int((out == 777.0).sum()),
              "| ident-stats err", float(err2.max()), "| ident-stats+unit-affine err", float(err3.max()))
        if bad.any():
            rows = bad.any(1).nonzero().flatten().tolist()
            cols = bad.any(0).nonzero().flatten().tolist()
            print("   bad rows", rows[:12], "...", len(rows), " bad cols", cols[:12], "...", len(cols))
            r0 = rows[0]
            print("   row", r0, "got", out[r0, :6].tolist(), "want", ref[r0, :6].tolist())
            # what would the output be without beta / without gamma / with x un-normalised?
            for name, alt in (("no-LN", torch.nn.functional.linear(x, w, b)),
                              ("gamma-only", torch.nn.functional.linear((x - stats[:, :1]) * stats[:, 1:] * gam, w, b)),
                              ("beta=gamma=0 (bias only)", b[None].expand(M, N))):
                print("   vs", name, float((out - alt).abs()[bad].max()))
