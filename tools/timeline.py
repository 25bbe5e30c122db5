"""Steady-state timeline of bench.py's graph replays from a rocprofv3 --kernel-trace CSV: per clip (delimited by the
single dyn_mask launch each forward ends with) the wall time, the time with 0 / 1 / >= 2 kernels in flight and the
kernel-time sum, plus the longest idle gaps and what ran around them.
usage: python tools/timeline.py <kernel_trace.csv> [groups_to_skip]
SOC_TRACE_CLIPS_PER_GROUP=2 for the pair pipeline (a replay ends with two dyn_mask launches: every second one delimits)."""
import csv
import os
import sys

PER = int(os.environ.get("SOC_TRACE_CLIPS_PER_GROUP", "1"))

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spin_kernel" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "dyn_mask" in r["Kernel_Name"]][PER - 1::PER]
ends = [int(rows[i]["End_Timestamp"]) for i in idx]
SPAN = 10 if PER == 1 else max(2, 14 // PER)       # launch groups in the window
if len(ends) < SPAN + 2:
    sys.exit("too few clips in the trace")
if len(sys.argv) > 2:
    skip = int(sys.argv[2])
else:
    # the SPAN consecutive whole replays that took the least wall time: a steady-state stretch of one timed pass (the passes of
    # bench.py are separated by host work; a pass begins with a head-only replay and ends with a tail-only one, whose segments
    # hold fewer kernels than a whole replay's)
    counts = [idx[j] - idx[j - 1] for j in range(1, len(idx))]             # kernels of segment j - 1 -> j
    med = sorted(counts)[len(counts) // 2]
    whole = [c >= 0.9 * med for c in counts]
    cands = [s_ for s_ in range(0, len(ends) - SPAN) if all(whole[s_:s_ + SPAN])]
    skip = min(cands or range(1, len(ends) - SPAN), key=lambda s_: ends[s_ + SPAN] - ends[s_])
t_lo, t_hi = ends[skip], ends[skip + SPAN]
n_clips = SPAN * PER
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= t_lo or s >= t_hi:
        continue
    ev.append((max(s, t_lo), 1, r["Kernel_Name"]))
    ev.append((min(e, t_hi), -1, r["Kernel_Name"]))
ev.sort(key=lambda x: (x[0], x[1]))
depth, last = 0, t_lo
hist = {0: 0, 1: 0, 2: 0}
gaps, prev_name = [], ""
for t, d, name in ev:
    hist[min(depth, 2)] += t - last
    if depth == 0 and t > last:
        gaps.append((t - last, prev_name, name))
    depth += d
    last = t
    if d < 0:
        prev_name = name
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows
           if t_lo <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t_hi)
wall = t_hi - t_lo
print(f"{n_clips} steady-state clips: wall {wall / n_clips / 1e6:.3f} ms/clip, kernel-time sum {busy / n_clips / 1e6:.3f} ms/clip")
print(f"   idle (no kernel)   {hist[0] / n_clips / 1e6:.3f} ms/clip")
print(f"   exactly one kernel {hist[1] / n_clips / 1e6:.3f} ms/clip")
print(f"   two or more        {hist[2] / n_clips / 1e6:.3f} ms/clip")
nk = sum(1 for r in rows if t_lo <= int(r["Start_Timestamp"]) < t_hi)
print(f"   kernels per clip {nk / n_clips:.0f}; idle gaps: {len(gaps) / n_clips:.0f} per clip, mean {hist[0] / max(len(gaps), 1) / 1e3:.2f} us")
print("   longest idle gaps (us): after -> before")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"      {g / 1e3:7.1f}  {a[:60]} -> {b[:60]}")
