#!/usr/bin/env python3
"""Steady-state per-launch counters of the first query tier from rocprofv3 counter_collection.csv files: tools/pmc_variant.py <dir> [last N dispatches, default 4]"""
import csv, glob, os, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_query_pool_fast" in r["Kernel_Name"]:
            rows.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            rows[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for c in sorted(rows):
    v = [rows[c][k] for k in sorted(rows[c])][-last:]
    print(f"{c:32s} mean of the last {len(v)} dispatches: {sum(v) / len(v):16.1f}")
