import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last reps replays: take kernels after the last capture: simply the last N*k rows where k = count per replay
names = [r["Kernel_Name"] for r in rows]
# find period: number of kernels between consecutive first-dyn_mask occurrences at the end
idx = [i for i, n in enumerate(names) if "dyn_mask" in n]
per = idx[-1] - idx[-1 - 4] if len(idx) > 8 else 0
tail = rows[-per * reps:] if per else rows
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:60]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"kernels per replay {per}, kernel time per replay {tot / reps:.1f} us")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{us / reps:8.1f} us  {c / reps:6.1f} x  {n}")
