#!/usr/bin/env python3
"""tools/pmc_summary.py DIR [substr] [--runs] — per-dispatch counter values from a rocprofv3 --pmc CSV directory, in
dispatch order, for kernels whose name contains `substr` (default: csr_).  --runs: one line per run of consecutive
dispatches of the same kernel (a workload of bench.py: warm-up + timed launches): count, median ms, counters of the last."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "csr_"
rows = collections.OrderedDict()
for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(anonymous namespace)::", 1)[-1].split("(")[0][-48:])
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
if "--runs" in sys.argv:
    import statistics

    run = []

    def flush():
        if run:
            (d0, name), last = run[0][0], run[-1][1]
            print(f"{d0}-{run[-1][0][0]} {name}: {len(run)} dispatches, median ms={statistics.median(v['ms'] for _, v in run):.4g}; last: "
                  + " ".join(f"{k}={last[k]:.4g}" for k in sorted(last) if k != "ms"))
        run.clear()

    prev = None
    for key, v in sorted(rows.items()):
        if prev is not None and (key[1] != prev[1] or key[0] != prev[0] + 1):
            flush()
        run.append((key, v))
        prev = key
    flush()
else:
    for (did, name), v in sorted(rows.items()):
        print(did, name, " ".join(f"{k}={v[k]:.4g}" for k in sorted(v)))
