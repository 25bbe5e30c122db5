"""K23 (mlp_split.hip) probe: numerics against f64 and the time of every diagnostic variant (ring depth, compute-only and
stream-only ceilings) on the model's shapes.  Builds its own library with -DSOC_K23_VARIANTS:

    python tools/experiments/k23_time.py --build        (here: hipcc cross-compiles)
    python tools/experiments/k23_time.py [--quick]      (GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "experiments", "_build")
LIB = os.path.join(BUILD, "libk23_probe%s.so" % os.environ.get("K23_LIB_TAG", ""))
CSRC = os.path.join(ROOT, "neurips2023_soc_amd", "csrc")


def build():
    os.makedirs(BUILD, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DSOC_K23_VARIANTS",
           *os.environ.get("K23_EXTRA_FLAGS", "").split(),
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB, os.path.join(CSRC, "mlp_split.hip"),
           os.path.join(CSRC, "soc_capi.hip")]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


if "--build" in sys.argv:
    build()
    sys.exit(0)

import torch  # noqa: E402

from neurips2023_soc_amd import fused, hot_ops  # noqa: E402

lib = C.CDLL(LIB)
p, i, f = C.c_void_p, C.c_int, C.c_float
lib.soc_mlp_split_packed_bytes.restype = C.c_size_t
lib.soc_mlp_split_packed_bytes.argtypes = [i, i]
lib.soc_mlp_split_pack_f32.argtypes = [p, p, p, i, i, p]
lib.soc_mlp_split_plan.argtypes = [C.c_long, i, i, C.POINTER(i), C.POINTER(i), C.c_void_p]
lib.soc_mlp_split_variant_f32.argtypes = [p, p, p, p, p, p, f, p, p, p, f, p, p, p, C.c_long, i, i, i, i, i, i, i, p]
g = torch.Generator().manual_seed(0)
quick = "--quick" in sys.argv


def t(fn, reps=20):
    for _ in range(3):
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


def case(name, M, Cw, F, act, ln, res, cuts, variants):
    x = torch.randn(M, Cw, generator=g).cuda()
    w1 = (torch.randn(F, Cw, generator=g) / Cw ** 0.5).cuda()
    b1 = torch.randn(F, generator=g).cuda()
    w2 = (torch.randn(Cw, F, generator=g) / F ** 0.5).cuda()
    b2 = torch.randn(Cw, generator=g).cuda()
    gam = (torch.rand(Cw, generator=g) + 0.5).cuda() if ln else None
    bet = (torch.randn(Cw, generator=g) * 0.1).cuda() if ln else None
    r = torch.randn(M, Cw, generator=g).cuda() if res else None
    packed = torch.empty(lib.soc_mlp_split_packed_bytes(Cw, F), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.soc_mlp_split_pack_f32(w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), Cw, F, st) == 0
    xd = x.double()
    if ln:
        xd = torch.nn.functional.layer_norm(xd, (Cw,), gam.double(), bet.double(), 1e-5)
    h = xd @ w1.double().t() + b1.double()
    h = torch.relu(h) if act == "relu" else torch.nn.functional.gelu(h)
    ref = h @ w2.double().t() + b2.double()
    if res:
        ref = ref + r.double()
    del h, xd
    scale = ref.abs().max().item()
    out = torch.empty_like(x)
    ptr = lambda v: v.data_ptr() if v is not None else None   # noqa: E731
    fl = 4.0 * M * F * Cw
    for cut in cuts:
        if cut is None:
            nrg, nfs = C.c_int(0), C.c_int(0)
            lib.soc_mlp_split_plan(M, Cw, F, C.byref(nrg), C.byref(nfs), None)
            nrg, nfs = nrg.value, nfs.value
        else:
            nrg, nfs = cut
        ws = torch.empty(nfs * M * Cw, dtype=torch.float32, device="cuda") if nfs > 1 else None
        for v in variants:
            def run():
                return lib.soc_mlp_split_variant_f32(x.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), ptr(gam),
                                                     ptr(bet), 1e-5, ptr(r), None, None, 0.0, out.data_ptr(), None, ptr(ws), M, Cw, F,
                                                     1 if act == "relu" else 2, 0, nrg, nfs, v, st)
            out.zero_()
            rc = run()
            if rc != 0:
                print(f"{name} cut ({nrg},{nfs}) variant {v}: rc {rc}")
                continue
            torch.cuda.synchronize()
            err = (out.double() - ref).abs().max().item() / scale
            us = t(run, 10 if quick else 20)
            dbg = (v >> 5) & 15
            tag = f"  NS={v & 7} SB={1 << ((v >> 3) & 3)} PF={((v >> 9) & 1) + 1} STAG={v >> 10}" + ("" if not dbg else "  [wrong on purpose:" + "".join(
                n for b, n in ((1, " no DMA in loop"), (2, " no MFMA"), (4, " no fragment reads"), (8, " no activation")) if dbg & b) + "]")
            print(f"{name} M={M} C={Cw} F={F} cut ({nrg},{nfs}) variant {v:3d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  "
                  f"max err / max|ref| {err:.2e}{tag}", flush=True)
    return x, w1, b1, w2, b2, gam, bet, r, ref, scale


def vr(ns=3, lsb=0, stag=0, pf=1, dbg=0):
    return ns + 8 * lsb + 32 * dbg + 512 * (pf - 1) + 1024 * stag


V = [0, vr(3), vr(3, stag=4), vr(2, stag=4)]
if "--enc-only" in sys.argv:         # round 6: the encoder form with and without its hand-off barrier (K23_EXTRA_FLAGS=-DSOC_K23_NO_BARRIER)
    case("enc", 32768, 256, 2048, "relu", False, True, [(256, 1)], [0, vr(3, dbg=13), vr(3, dbg=1), vr(3, dbg=4)])
    case("s0", 115200, 96, 384, "gelu", True, True, [(256, 1)], [0])
    case("s2", 7360, 384, 1536, "gelu", True, True, [None], [0])
    sys.exit(0)
case("enc", 32768, 256, 2048, "relu", False, True, [(256, 1)], V)
case("s0", 115200, 96, 384, "gelu", True, True, [(256, 1)], [0, vr(3), vr(3, stag=4), vr(4, stag=4)])
case("s1", 28800, 192, 768, "gelu", True, True, [None], V)
case("s2", 7360, 384, 1536, "gelu", True, True, [None], [0, vr(4, 1), vr(4, 1, stag=4)])
# Swin-B stage 2 (C = 512): the shipped form (three half-block slots), its ceilings, quarter-block pieces through 4 / 5 / 6 slots
case("s2b", 7360, 512, 2048, "gelu", True, True, [None, (115, 1), (115, 4)],
     [0, vr(3, 1), vr(3, 1, dbg=1), vr(3, 1, dbg=2), vr(3, 1, dbg=4), vr(3, 1, dbg=8), vr(3, 1, dbg=13), vr(4, 2), vr(5, 2), vr(6, 2)])
