#!/usr/bin/env python3
"""tools/pmc_summary.py DIR [substr] — per-dispatch counter values from a rocprofv3 --pmc CSV directory, in dispatch
order, for kernels whose name contains `substr` (default: csr_)."""
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
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][-48:])
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (did, name), v in sorted(rows.items()):
    print(did, name, " ".join(f"{k}={v[k]:.4g}" for k in sorted(v)))
