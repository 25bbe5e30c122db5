"""Where a K24 launch spends its time: a -DSOC_K24_STAMPS build (never shipped) whose waves stamp s_memtime at the phase
boundaries of a pass -- start | row loads + first LDS-DMA issued | LayerNorm + split done | first ring piece landed |
MFMA loop done | stores issued | block retired -- for the K <= 384 shapes of the model.
    python tools/experiments/k24_stamps.py --build ;  python tools/experiments/k24_stamps.py   (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "experiments", "_build")
LIB = os.path.join(BUILD, "libk24_stamps.so")
CSRC = os.path.join(ROOT, "neurips2023_soc_amd", "csrc")
if "--build" in sys.argv:
    os.makedirs(BUILD, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DSOC_K24_STAMPS",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB, os.path.join(CSRC, "xs_linear_split.hip"),
           os.path.join(CSRC, "xs_linear_split_wide.hip"), os.path.join(CSRC, "soc_capi.hip")]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    sys.exit(0)

import torch  # noqa: E402

lib = C.CDLL(LIB)
p, i, f = C.c_void_p, C.c_int, C.c_float
lib.soc_xs_linear_packed_bytes.restype = C.c_size_t
lib.soc_xs_linear_packed_bytes.argtypes = [i, i]
lib.soc_xs_linear_pack_f32.argtypes = [p, p, i, i, p]
lib.soc_xs_linear_plan.argtypes = [C.c_long, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.c_void_p]
lib.soc_xs_linear_f32.argtypes = [p, p, p, p, p, f, p, p, C.c_long, i, i, i, i, i, p]
lib.soc_xs_debug_set_buffer.argtypes = [p]
g = torch.Generator().manual_seed(0)
names = ["rows + DMA issued", "LN + split", "piece 0 landed", "MFMA loop", "epilogue + stores", "final barrier"]
for name, M, N, K in (("s2.qkv", 7360, 1152, 384), ("s2.proj", 7360, 384, 384), ("enc.value", 38560, 256, 256), ("vlf.q", 28800, 256, 256)):
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    packed = torch.empty(lib.soc_xs_linear_packed_bytes(N, K), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.soc_xs_linear_pack_f32(w.data_ptr(), packed.data_ptr(), N, K, st) == 0
    nrg, ncr, nct = C.c_int(), C.c_int(), C.c_int()
    lib.soc_xs_linear_plan(M, N, K, C.byref(nrg), C.byref(ncr), C.byref(nct), None)
    dbg = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device="cuda")

    def run():
        return lib.soc_xs_linear_f32(x.data_ptr(), packed.data_ptr(), b.data_ptr(), None, None, 0.0, None, out.data_ptr(), M, N, K, 0,
                                     0, 0, st)
    lib.soc_xs_debug_set_buffer(None)
    for _ in range(20):
        assert run() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    lib.soc_xs_debug_set_buffer(dbg.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.soc_xs_debug_set_buffer(None)
    d = dbg.cpu().view(4096, 8, 8)
    nblk = nrg.value * ncr.value
    d = d[:min(nblk, 4096)]
    print(f"\n{name} {M}x{N}x{K}: plan ({nrg.value}, {ncr.value}, {nct.value}) = {nblk} workgroups, {us:.1f} us per launch (last pass of a wave stamped)")
    for wv in (0, 4):
        t = d[:, wv, :7].double()
        ok = (t[:, 0] > 0) & (t[:, 6] > 0)
        t = t[ok]
        deltas = (t[:, 1:] - t[:, :-1])
        med = deltas.median(0)[0]
        print(f"   wave {wv}: " + ", ".join(f"{n} {int(v)}" for n, v in zip(names, med.tolist())) + f"  | total {int((t[:, 6] - t[:, 0]).median())} clk")
