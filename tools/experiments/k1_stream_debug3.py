"""Round-6 debugging aid: diagnostic builds of the streaming K1 form (-DSOC_K1_DBG=1: never the two-pass tile, =2: always)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import soc_oracle as O  # noqa: E402


def build(flags, tag):
    so = f"/tmp/libk1_{tag}.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags,
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "neurips2023_soc_amd/csrc"),
                    "-o", so, os.path.join(ROOT, "neurips2023_soc_amd/csrc/win_attn3d.hip"),
                    os.path.join(ROOT, "neurips2023_soc_amd/csrc/soc_capi.hip")], check=True)
    return C.CDLL(so)


def call(lib, qkv, bias, table, nH, shift, form=1):
    B, D, H, W, C3 = qkv.shape
    out = torch.empty(B, D, H, W, C3 // 3, device="cuda")
    sh = (0 if D <= 8 else shift[0], shift[1] if H > 7 else 0, shift[2] if W > 7 else 0)
    args = [C.c_void_p(x.data_ptr()) for x in (qkv, bias, table, out)] + [C.c_int(v) for v in
            (B, D, H, W, C3 // 3, nH, 8, 7, 7, *sh, 8, 7, 7, form)] + [C.c_void_p(torch.cuda.current_stream().cuda_stream)]
    rc = lib.soc_win_attn3d_f32(*args)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu()


g = torch.Generator().manual_seed(3)
nH, D, H, W = 1, 8, 14, 14
Cc = 32
qkv = torch.randn(1, D, H, W, 3 * Cc, generator=g)
bias = torch.randn(3 * Cc, generator=g) * 0.5
table = torch.randn(15 * 13 * 13, nH, generator=g) * 0.5
libs = {"normal": build([], "n"), "never two-pass": build(["-DSOC_K1_DBG=1"], "d1"), "always two-pass": build(["-DSOC_K1_DBG=2"], "d2")}
for shift in ((0, 0, 0), (4, 3, 3)):
    ref = O.window_attention_core(qkv, bias, table, nH, O.WINDOW, shift)
    for name, lib in libs.items():
        out = call(lib, qkv.cuda(), bias.cuda(), table.cuda(), nH, shift)
        e = (out - ref).abs()[0]
        print(f"shift {shift} {name:16s}: frames 0-3 {float(e[:4].max()):.2e}  frames 4-7 {float(e[4:].max()):.2e}")
