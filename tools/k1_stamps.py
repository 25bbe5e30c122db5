"""Diagnostic: per-phase s_memtime stamps of K1 (separate -DSOC_K1_STAMPS build; never shipped).
usage: python tools/k1_stamps.py [stage]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "gpurun_out", "libsoc_k1_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DSOC_K1_STAMPS",
                "-I", os.path.join(ROOT, "include"), "-o", so,
                os.path.join(ROOT, "neurips2023_soc_amd/csrc/win_attn3d.hip")], check=True)
lib = C.CDLL(so)
st = int(sys.argv[1]) if len(sys.argv) > 1 else 0
H, W, nH = [(90, 160, 3), (45, 80, 6), (23, 40, 12), (12, 20, 24)][st]
Cc = nH * 32
g = torch.Generator().manual_seed(0)
qkv = torch.randn(1, 8, H, W, 3 * Cc, generator=g).cuda()
bias = torch.randn(3 * Cc, generator=g).cuda()
table = torch.randn(2535, nH, generator=g).cuda()
out = torch.empty(1, 8, H, W, Cc).cuda()
nblk = 4096
dbg = torch.zeros(nblk * 8 * 32, dtype=torch.int64).cuda()
lib.soc_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
args = [C.c_void_p(t.data_ptr()) for t in (qkv, bias, table, out)] + [C.c_int(v) for v in
        (1, 8, H, W, Cc, nH, 8, 7, 7, 0, 3, 3, 8, 7, 7)] + [C.c_void_p(0)]
for _ in range(2):
    rc = lib.soc_win_attn3d_f32(*args)
    assert rc == 0, rc
torch.cuda.synchronize()
d = dbg.cpu().view(nblk, 8, 32)
for blk in (0, 300):
    for w in (0, 4, 1):
        s = d[blk, w]
        n = int((s > 0).sum())
        rel = [(int(s[i]) - int(s[0])) for i in range(n)]
        deltas = [rel[i] - rel[i - 1] for i in range(1, n)]
        print(f"blk {blk} wave {w}: start@{int(s[0]) - int(d[0, 0, 0])} deltas {deltas}")
