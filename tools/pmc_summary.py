#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files (one counter per pass) into per-kernel means.
usage: python tools/pmc_summary.py gpurun_out/pmc_kb_FETCH_SIZE gpurun_out/pmc_kb_WRITE_SIZE ... > profiles/xxx.json"""
import collections
import csv
import glob
import json
import sys

out = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out[k][c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
