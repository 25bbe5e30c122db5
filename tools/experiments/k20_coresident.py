#!/usr/bin/env python
"""Is K20 correct when TWO of its workgroups share a CU?  The product launches one persistent workgroup per CU; the
128 x 64 tile needs 72 KB of LDS, so a build that launches 2 workgroups per CU (-DSOC_K20_DBG_BLOCKS_PER_CU=2) puts two
on a CU.  (Question behind it: are LDS-DMA destinations relocated by the workgroup's LDS base?)"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(flags, tag):
    so = f"/tmp/libk20_{tag}.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags,
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "neurips2023_soc_amd/csrc"),
                    "-o", so, os.path.join(ROOT, "neurips2023_soc_amd/csrc/linear_split.hip")], check=True)
    lib = C.CDLL(so)
    lib.soc_linear_split_packed_bytes.restype = C.c_size_t
    return lib


dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
M, K, N = 38560, 256, 512
x = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(N, K, generator=g) / 16).to(dev)
b = torch.randn(N, generator=g).to(dev)
ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
for tag, flags in (("one", []), ("two", ["-DSOC_K20_DBG_BLOCKS_PER_CU=2"]), ("four", ["-DSOC_K20_DBG_BLOCKS_PER_CU=4"])):
    lib = build(flags, tag)
    packed = torch.empty(lib.soc_linear_split_packed_bytes(N, K), dtype=torch.uint8, device=dev)
    assert lib.soc_linear_split_pack_f32(p(w), p(packed), N, K, None) == 0
    for tile in (4, 2, 0):
        errs = []
        for rep in range(20):
            out = torch.full((M, N), 777.0, device=dev)
            rc = lib.soc_linear_split_f32(p(x), None, None, None, p(packed), p(b), None, None, p(out), None, 0, C.c_long(M), N, K, 0,
                                          tile, None)
            assert rc == 0
            errs.append(float((out.double() - ref).abs().max()))
        print(tag, "workgroups per CU, tile", tile, ": max err over 20 launches", max(errs), "bad launches", sum(e > 1e-3 for e in errs))
