"""Aggregate rocprofv3 --pmc passes: mean counter value per launch, per kernel-name pattern.
usage: python tools/pmc_agg.py --kernels name1=substr1,substr2 name2=substr ... -- DIR [DIR ...] > json
Every *counter_collection.csv under the DIRs is read (one pass = one DIR; counters that did not fit one pass were
collected in separate runs, as MI355X_MICROARCH.md prescribes).  `skip` leading launches per kernel are dropped
(warm-up)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    av = sys.argv[1:]
    sep = av.index("--")
    opts, dirs = av[:sep], av[sep + 1:]
    kernels, skip = collections.OrderedDict(), 3
    for o in opts:
        if o == "--kernels":
            continue
        if o.startswith("--skip="):
            skip = int(o.split("=")[1])
            continue
        name, pats = o.split("=", 1)
        kernels[name] = pats.split(",")
    out = {k: {} for k in kernels}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(path)):
                for k, pats in kernels.items():
                    if all(p in r["Kernel_Name"] for p in pats):
                        per[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for k, counters in per.items():
                for c, vals in counters.items():
                    vals.sort()
                    vals = [v for _, v in vals[skip:]] or [v for _, v in vals]
                    out[k][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
        # rocprofv3 >= 7 writes a rocpd SQLite database unless --output-format csv is given
        for path in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            import sqlite3
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            durs = collections.defaultdict(dict)
            con = sqlite3.connect(path)
            for name, did, cname, val, dur in con.execute(
                    "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
                for k, pats in kernels.items():
                    if all(p in name for p in pats):
                        per[k][cname].append((did, float(val)))
                        durs[k][did] = dur
            for k, counters in per.items():
                for c, vals in counters.items():
                    vals.sort()
                    vals = [v for _, v in vals[skip:]] or [v for _, v in vals]
                    out[k][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
                dd = [v for _, v in sorted(durs[k].items())][skip:]
                if dd:
                    out[k].setdefault("duration_ns_under_profiler", {})[os.path.basename(d.rstrip("/"))] = sum(dd) / len(dd)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
