"""K24: the XCD-aware workgroup mapping (the column ranges of a row group on ONE XCD) against the round-4 mapping, same
process.  Builds a second copy of the K24 translation units with -DSOC_K24_NO_XCD_ROWS and calls both through the C ABI on the
same packed image: equality bit for bit, time per launch (HIP events around back-to-back launches).
usage: python tools/experiments/k24_xcd_rows.py [reps]        (under rocprofv3 --pmc FETCH_SIZE for the traffic)"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import _lib, hot_ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
csrc = os.path.join(ROOT, "neurips2023_soc_amd", "csrc")
so = os.path.join(ROOT, "tools", "experiments", "_build", "libk24_base.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or any(os.path.getmtime(os.path.join(csrc, f)) > os.path.getmtime(so)
                                 for f in ("xs_linear_split.h", "xs_linear_split.hip", "xs_linear_split_wide.hip")):
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DSOC_K24_NO_XCD_ROWS",
                    "-I", os.path.join(ROOT, "include"), "-I", csrc, "-o", so, os.path.join(csrc, "xs_linear_split.hip"),
                    os.path.join(csrc, "xs_linear_split_wide.hip"), os.path.join(csrc, "soc_capi.hip")], check=True)
if not torch.cuda.is_available():
    print("built", so)
    raise SystemExit(0)
base = C.CDLL(so)
new = _lib.load()
p, i, f = C.c_void_p, C.c_int, C.c_float
for lib in (base, new):
    lib.soc_xs_linear_f32.restype = i
    lib.soc_xs_linear_f32.argtypes = [p, p, p, p, p, f, p, p, C.c_long, i, i, i, i, i, p]
g = torch.Generator().manual_seed(0)
for name, M, N, K, res in (("swin-t s2 qkv", 7360, 1152, 384, False), ("swin-t s2 proj", 7360, 384, 384, True),
                           ("merge into s2", 7360, 384, 768, False), ("swin-t s3 qkv", 1920, 2304, 768, False),
                           ("encoder value", 38560, 256, 256, False), ("swin-b s2 proj", 7360, 512, 512, True),
                           ("swin-b s2 qkv", 7360, 1536, 512, False), ("merge into s1", 28800, 192, 384, False)):
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    packed = hot_ops._xs_packed(w)
    outs = {}
    times = {}
    st = torch.cuda.current_stream().cuda_stream
    for tag, lib in (("round-4 mapping", base), ("xcd rows", new), ("round-4 mapping again", base), ("xcd rows again", new)):
        out = torch.empty(M, N, device="cuda")

        def call():
            rc = lib.soc_xs_linear_f32(x.data_ptr(), packed.data_ptr(), b.data_ptr(), None, None, 0.0,
                                       r.data_ptr() if r is not None else None, out.data_ptr(), M, N, K, 0, 0, 0, st)
            assert rc == 0, rc
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            call()
        e.record()
        torch.cuda.synchronize()
        times[tag] = 1e3 * s.elapsed_time(e) / reps
        outs[tag] = out
    nrg, ncr, nct = hot_ops.xs_linear_plan(M, N, K)
    print(f"{name:16s} [{M},{N},{K}] plan nrg {nrg} ncr {ncr}: round-4 mapping {times['round-4 mapping']:6.1f} / "
          f"{times['round-4 mapping again']:6.1f} us, xcd rows {times['xcd rows']:6.1f} / {times['xcd rows again']:6.1f} us, "
          f"equal {bool(torch.equal(outs['round-4 mapping'], outs['xcd rows']))}")
