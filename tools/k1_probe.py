"""K1 diagnostics on an MI355X: (A) HIP-event time per Video-Swin stage geometry through the shipped library,
(B) a -DSOC_K1_STAMPS build (never shipped) whose workgroups stamp s_memtime at phase boundaries and
s_memrealtime at start / end: in-kernel clock under load, per-phase cycles, workgroup rounds per CU.
(C) --sweep: forced schedules.  (D) --insitu: forced schedules timed inside the 12-launch sequence of a forward (every
launch on its own qkv tensor, 634 MB in all, so that K/V staging sees the cache state it has inside the model).
usage: python tools/k1_probe.py [--stamps|--sweep|--insitu] [--flags=-DX,-DY] [stages...]"""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GEO = [(90, 160, 3), (45, 80, 6), (23, 40, 12), (12, 20, 24)]      # Swin-T, T=8, 360x640: (H, W, heads) per stage
CLIPS = int(os.environ.get("K1_CLIPS", "1"))                       # clips per launch (10 = the launch group of the bench)
FLOP = lambda H, W, nH: CLIPS * 4.0 * 392 * 392 * 32 * nH * (-(-H // 7)) * (-(-W // 7))   # noqa: E731


def inputs(st):
    H, W, nH = GEO[st]
    Cc = nH * 32
    g = torch.Generator().manual_seed(st)
    return (torch.randn(CLIPS, 8, H, W, 3 * Cc, generator=g).cuda(), torch.randn(3 * Cc, generator=g).cuda(),
            (torch.randn(2535, nH, generator=g) * 0.2).cuda(), torch.empty(CLIPS, 8, H, W, Cc).cuda())


def build(flags, tag):
    so = f"/tmp/libk1_{tag}.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", *flags,
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "neurips2023_soc_amd/csrc"),
                    "-o", so, os.path.join(ROOT, "neurips2023_soc_amd/csrc", os.environ.get("K1_SRC", "win_attn3d.hip")),
                    os.path.join(ROOT, "neurips2023_soc_amd/csrc/soc_capi.hip")], check=True)
    return C.CDLL(so)


def call(lib, t, st, shift):
    H, W, nH = GEO[st]
    qkv, bias, table, out = t
    args = [C.c_void_p(x.data_ptr()) for x in (qkv, bias, table, out)] + [C.c_int(v) for v in
            (CLIPS, 8, H, W, nH * 32, nH, 8, 7, 7, *shift, 8, 7, 7, int(os.environ.get('K1_SPLIT', '1')))] + [C.c_void_p(torch.cuda.current_stream().cuda_stream)]
    rc = lib.soc_win_attn3d_f32(*args)
    assert rc == 0, rc


def time_lib(lib, stages, reps=200):
    res = {}
    for st in stages:
        t = inputs(st)
        for shift in ((0, 0, 0), (4, 3, 3)):
            for _ in range(20):
                call(lib, t, st, shift)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                call(lib, t, st, shift)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / reps * 1e3
            res[(st, shift != (0, 0, 0))] = us
            print(f"stage {st} shift {shift}: {us:7.1f} us  {FLOP(*GEO[st]) / us / 1e6:6.1f} TFLOP/s "
                  f"({FLOP(*GEO[st]) / us / 1e6 / 157.3:.3f} of f32 MFMA peak)", flush=True)
    return res


def stamps(stages, flags):
    lib = build(["-DSOC_K1_STAMPS", *flags], "stamps")
    nblk = 8192 if CLIPS == 1 else 16384
    dbg = torch.zeros(nblk * 8 * 32, dtype=torch.int64).cuda()
    lib.soc_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
    for st in stages:
        t = inputs(st)
        for shift in ((0, 0, 0), (4, 3, 3)):
            t0 = time.time()
            while time.time() - t0 < 1.5:                  # hold the load so the clock settles
                for _ in range(50):
                    call(lib, t, st, shift)
                torch.cuda.synchronize()
            dbg.zero_()
            for _ in range(3):
                call(lib, t, st, shift)
            torch.cuda.synchronize()
            d = dbg.cpu().view(nblk, 8, 32)
            used = d[:, 0, 0] > 0
            d = d[used]
            nb = d.shape[0]
            rt0, rt1 = d[:, :, 30], d[:, :, 31]             # 100 MHz
            span_us = (int(rt1.max()) - int(rt0.min())) / 100.0
            # in-kernel clock: cycles between first and last s_memtime stamp over the realtime between 30 and 31
            last = torch.zeros(nb, 8, dtype=torch.int64)
            for b in range(nb):
                for w in range(8):
                    row = d[b, w, :29]
                    nz = row[row > 0]
                    last[b, w] = nz[-1] if len(nz) else 0
            cyc = (last - d[:, :, 0]).double()
            rt = (rt1 - rt0).double() * 10.0               # ns
            ghz = (cyc / rt)[rt > 0]
            blk_cyc = (last.max(1)[0] - d[:, :, 0].min(1)[0]).double()
            print(f"\n== stage {st} shift {shift}: {nb} workgroups, kernel span {span_us:.1f} us, in-kernel clock "
                  f"median {float(ghz.median()):.3f} GHz (p10 {float(ghz.quantile(0.1)):.3f}, p90 {float(ghz.quantile(0.9)):.3f})")
            print(f"   workgroup lifetime cycles: median {float(blk_cyc.median()):.0f}, p10 {float(blk_cyc.quantile(0.1)):.0f}, "
                  f"p90 {float(blk_cyc.quantile(0.9)):.0f}")
            # phase deltas of wave 0 / wave 4 (SIMD partners) / wave 1 for the median-lifetime workgroup
            order = torch.argsort(blk_cyc)
            for b in (int(order[nb // 2]),):
                for w in (0, 4, 1, 5):
                    row = d[b, w, :29]
                    n = int((row > 0).sum())
                    rel = [int(row[i]) - int(row[0]) for i in range(n)]
                    print(f"   blk {b} wave {w}: deltas {[rel[i] - rel[i - 1] for i in range(1, n)]}")
            # rounds: workgroups per (xcc, cu) and their start times
            hw = d[:, 0, 29]
            xcc = (hw >> 32) & 0xF
            cu = (hw >> 8) & 0xF
            se = (hw >> 13) & 0x7
            key = (xcc * 64 + se * 16 + cu).tolist()
            per = {}
            for b, k in enumerate(key):
                per.setdefault(k, []).append((int(rt0[b, 0]) - int(rt0.min())) / 100.0)
            counts = sorted(len(v) for v in per.values())
            print(f"   distinct CUs {len(per)}; workgroups per CU min {counts[0]} median {counts[len(counts) // 2]} max {counts[-1]}")
            starts = sorted(per.values(), key=len)[-1]
            print(f"   start times (us) on the busiest CU: {[round(x, 1) for x in sorted(starts)]}")
            ends = ((rt1.max(1)[0] - int(rt0.min())).double() / 100.0)
            hist = torch.histc(ends.float(), bins=12, min=0, max=span_us)
            print(f"   workgroup end-time histogram over the span: {[int(x) for x in hist.tolist()]}")


def sweep(stages, flags):
    """time forced (n_main, qsplit) schedules (diagnostic build -DSOC_K1_TUNE) to calibrate the planner's cost model"""
    lib = build(["-DSOC_K1_TUNE", *flags], "tune")
    for st in stages:
        H, W, nH = GEO[st]
        pairs = nH * (-(-H // 7)) * (-(-W // 7))
        t = inputs(st)
        cands = [(pairs, 1)]
        for q in (2, 3, 4, 5, 7):
            for nm in sorted({0, pairs - pairs % 256, pairs - pairs % 256 - 256, pairs - pairs % 256 - 64,
                              pairs - pairs % 256 - 32, pairs - pairs % 256 + 32, pairs - pairs % 256 - 128}):
                if 0 <= nm < pairs and (pairs - nm) * q <= 4096:
                    cands.append((nm, q))
        rows = []
        for nm, q in cands:
            lib.soc_debug_force_k1_plan(nm, q)
            for _ in range(10):
                call(lib, t, st, (4, 3, 3))
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(60):
                call(lib, t, st, (4, 3, 3))
            e.record()
            torch.cuda.synchronize()
            rows.append((s.elapsed_time(e) / 60 * 1e3, nm, q))
        lib.soc_debug_force_k1_plan(0, 0)
        for _ in range(10):
            call(lib, t, st, (4, 3, 3))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(60):
            call(lib, t, st, (4, 3, 3))
        e.record()
        torch.cuda.synchronize()
        print(f"stage {st}: {pairs} pairs; planner's own choice {s.elapsed_time(e) / 60 * 1e3:.1f} us")
        for us, nm, q in sorted(rows)[:8] + [r for r in rows if r[2] == 1]:
            print(f"   n_main {nm:5d} qsplit {q}: {us:7.1f} us  blocks {nm + (pairs - nm) * q}")


def insitu(stages, flags, reps=12, own_only=False):
    lib = build(["-DSOC_K1_TUNE", *flags], "tune" + "".join(f.replace("-D", "_") for f in flags))
    seq = [0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3]                      # stage of the 12 K1 launches of a Swin-T forward
    tens = [inputs(st) for st in seq]
    shifts = [(0, 0, 0), (4, 3, 3)] * 6

    def run(target, plan):
        """mean time (us) of one launch of stage `target` under `plan` = (n_main, qsplit) or None (planner)"""
        tot, n = 0.0, 0
        for rep in range(reps + 2):
            evs = []
            for i, st in enumerate(seq):
                forced = plan is not None and st == target
                lib.soc_debug_force_k1_plan(*(plan if forced else (0, 0)))
                if st == target:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    call(lib, tens[i], st, shifts[i])
                    b.record()
                    evs.append((a, b))
                else:
                    call(lib, tens[i], st, shifts[i])
            torch.cuda.synchronize()
            if rep >= 2:
                tot += sum(a.elapsed_time(b) for a, b in evs) * 1e3
                n += len(evs)
        lib.soc_debug_force_k1_plan(0, 0)
        return tot / n

    for st in stages:
        H, W, nH = GEO[st]
        pairs = nH * (-(-H // 7)) * (-(-W // 7))
        base = pairs - pairs % 256
        cands = {(pairs, 1)}
        for q in (2, 3, 4, 5, 6, 7, 8):
            for nm in (0, base - 256, base - 192, base - 128, base - 96, base - 64, base - 32, base, base + 32, base + 64):
                if 0 <= nm < pairs and (pairs - nm) * q <= 4096:
                    cands.add((nm // 8 * 8, q))
        run(st, None)                                            # warm-up: clocks, caches
        own = run(st, None)
        if own_only:
            print(f"stage {st}: planner's own choice {own:.1f} / {run(st, None):.1f} us  flags {flags}")
            continue
        rows = sorted((run(st, c), c) for c in sorted(cands))
        own2 = run(st, None)
        print(f"stage {st}: {pairs} pairs; planner's own choice {own:.1f} / {own2:.1f} us before / after the sweep "
              f"(event pairs, in sequence)")
        for us, (nm, q) in rows[:8]:
            print(f"   n_main {nm:5d} qsplit {q}: {us:7.1f} us  blocks {nm + (pairs - nm) * q}")


if __name__ == "__main__":
    av = sys.argv[1:]
    flags = []
    for a in list(av):
        if a.startswith("--flags="):
            flags = a.split("=", 1)[1].split(",")
            av.remove(a)
    do_stamps = "--stamps" in av
    stages = [int(a) for a in av if a.isdigit()] or [0, 1, 2, 3]
    if "--insitu" in av:
        insitu(stages, flags, own_only="--own" in av)
    elif "--sweep" in av:
        sweep(stages, flags)
    elif do_stamps:
        stamps(stages, flags)
    else:
        lib = build(flags, "var") if flags else None
        if lib is None:
            from neurips2023_soc_amd import _lib
            lib = _lib.load()
        time_lib(lib, stages)
