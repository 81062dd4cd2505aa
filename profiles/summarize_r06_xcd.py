#!/usr/bin/env python
"""Summarise tools/gpu/r06_xcd.sh: HBM traffic and L2 hit rate of the field launches of the README configuration (a
view-direction static trunk beside the dynamic one: unequal trunks), this tree against the round-5 tree.

    python profiles/summarize_r06_xcd.py gpurun_out/r06_<tag>

Per dispatch of nsff_field_kernel_h3a (both-trunk launches have the larger grid): FETCH_SIZE (KiB; doubled as
MI355X_MICROARCH.md prescribes for gfx950: it tallies 128-B requests at 64 B), WRITE_SIZE (KiB), TCC hits / requests."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]


def per_dispatch(name):
    out = collections.defaultdict(dict)
    for path in glob.glob(os.path.join(root, f"readme_pmc_{name}", "**", "*_counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "nsff_field_kernel_h3a" not in row["Kernel_Name"]:
                    continue
                d = out[row["Dispatch_Id"]]
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                d["grid"] = int(row.get("Grid_Size", 0) or 0)
    return out


def table(label, fetch, write=None, tcc=None):
    by_grid = collections.defaultdict(list)
    for did, d in fetch.items():
        by_grid[d["grid"]].append((did, d))
    print(f"## {label}")
    for grid in sorted(by_grid):
        rows = by_grid[grid]
        f = [2.0 * d["FETCH_SIZE"] * 1024 for _, d in rows]
        line = f"  grid {grid:8d} threads ({grid // 256:5d} workgroups)  launches {len(rows):3d}  FETCH x2 mean {sum(f) / len(f) / 1e6:9.1f} MB  max {max(f) / 1e6:9.1f} MB"
        if write:
            w = [write[did]["WRITE_SIZE"] * 1024 for did, _ in rows if did in write]
            if w:
                line += f"  WRITE mean {sum(w) / len(w) / 1e6:8.1f} MB"
        if tcc:
            h = [(tcc[did]["TCC_HIT_sum"], tcc[did]["TCC_REQ_sum"]) for did, _ in rows if did in tcc and tcc[did].get("TCC_REQ_sum")]
            if h:
                line += f"  L2 hit {100.0 * sum(a for a, _ in h) / sum(b for _, b in h):6.2f} %"
        print(line)


table("this tree (trunk by XCD, the longer trunk's tail as a second round of workgroups)", per_dispatch("fetch"), per_dispatch("write"), per_dispatch("tcc"))
table("round-5 tree (trunks split by workgroup index: both trunks on every XCD)", per_dispatch("fetch_base"))
