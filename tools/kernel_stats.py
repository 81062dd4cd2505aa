#!/usr/bin/env python
"""Per-step summary of a rocprofv3 --kernel-trace --stats run:  python tools/kernel_stats.py <dir-or-csv> <steps> [top]"""
import csv, glob, os, sys
src, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.1f} launches/step")
own = sum(float(r["TotalDurationNs"]) for r in rows if "nsff" in r["Name"] or "anonymous namespace)::" in r["Name"] and "at::native" not in r["Name"])
print(f"  of which this library's kernels {own / steps / 1e6:.3f} ms/step")
for r in rows[:top]:
    print(f"{int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
