"""Times K4 (dynamic mask head) and the experimental variants of tools/experiments/k4_variants.hip on an MI355X.

    python tools/k4_probe.py            # shipped kernel + variants, BASELINE geometry (T=8, Q=20, 90x160)
"""
import ctypes as C
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurips2023_soc_amd import hot_ops  # noqa: E402


def time_us(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "experiments", "k4_variants.hip")
    so = "/tmp/k4_variants.so"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-o", so, src])
    lib = C.CDLL(so)
    p, i, f = C.c_void_p, C.c_int, C.c_float
    lib.k4_variant.argtypes = [i, i, p, p, p, p, i, i, i, i, f, f, i, p]
    T, Q, h, w = 8, 20, 90, 160
    g = torch.Generator(device="cuda").manual_seed(0)
    feats = torch.randn(T, 8, h, w, device="cuda", generator=g)
    params = torch.randn(T * Q, 169, device="cuda", generator=g) * 0.3
    refs = torch.rand(T * Q, 2, device="cuda", generator=g)
    want = hot_ops.dynamic_mask(feats, params, refs, (360.0, 640.0), 4)
    base = time_us(lambda: hot_ops.dynamic_mask(feats, params, refs, (360.0, 640.0), 4))
    print(f"shipped (through hot_ops): {base:7.2f} us")
    out = torch.empty(T * Q, h, w, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for P in (1, 2, 4):
        for qpb in (1, 2, 3, 4, 5, 7, 10, 20):
            def run():
                rc = lib.k4_variant(P, qpb, feats.data_ptr(), params.data_ptr(), refs.data_ptr(), out.data_ptr(), T, Q, h, w,
                                    360.0, 640.0, 4, st)
                assert rc == 0
            out.zero_()
            run()
            err = (out.view_as(want) - want).abs().max().item()
            print(f"P={P} q_per_block={qpb:2d}: {time_us(run):7.2f} us   max|d| vs shipped {err:.2e}")


if __name__ == "__main__":
    main()
