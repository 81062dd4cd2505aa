#!/usr/bin/env python
"""Summarise rocprofv3 counter_collection CSVs for one kernel: mean per dispatch of each counter.

usage: python profiles/summarize_pmc.py gpurun_out/pmc_r01 [kernel-substring] > profiles/r01_pmc_summary.txt
"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "nsff_field_kernel"
for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    sums, disp = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern not in row["Kernel_Name"]:
                continue
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Counter_Name"]].add(row["Dispatch_Id"])
    print(f"# {os.path.basename(path)}  kernel~'{kern}'")
    for name in sorted(sums):
        n = len(disp[name])
        print(f"{name:32s} dispatches={n:3d}  mean/dispatch={sums[name] / n:.6g}  total={sums[name]:.6g}")
