"""Ordered kernel launches of one steady-state forward in a rocprofv3 --kernel-trace CSV (one line per launch).
usage: python tools/launch_sequence.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spin_kernel" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os  # noqa: E402
PER = int(os.environ.get("SOC_TRACE_CLIPS_PER_GROUP", "1"))      # pair pipeline: a replay (2 clips) ends with two dyn_mask launches
idx = [i for i, r in enumerate(rows) if "dyn_mask" in r["Kernel_Name"]][PER - 1::PER]
# a steady-state replay: among the segments between two delimiters, the ones with the most common kernel count are whole
# replays (head of one group beside the tail of the previous one; the passes of bench.py begin with a head-only replay, end
# with a tail-only one and are separated by host work) -- of those, the one that took the least wall time
_segs = [(idx[j] - idx[j - 1], int(rows[idx[j]]["End_Timestamp"]) - int(rows[idx[j - 1]]["End_Timestamp"]), j) for j in range(1, len(idx))]
_med = sorted(n for n, _, _ in _segs)[len(_segs) // 2]
k = min((w, j) for n, w, j in _segs if n >= 0.9 * _med)[1]
seg = rows[idx[k - 1] + 1: idx[k] + 1]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
for i, r in enumerate(seg):
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", n)
    n = m.group(1) if m else n.split("(")[0][:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {r['Grid_Size_X']:>8s}  {n}")
    prev_end = max(prev_end, e)
