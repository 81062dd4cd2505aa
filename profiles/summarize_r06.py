#!/usr/bin/env python
"""Summarise a profiles/collect_r06.sh run: per kernel the average duration (from the kernel trace of the SAME
counter pass), the counters per dispatch and the derived rates.

    python profiles/summarize_r06.py gpurun_out/r06_<tag> > profiles/r06_<tag>_summary.txt

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950 (it tallies 128-B requests at 64 B).  Durations of counter passes are inflated by the profiler; the un-profiled
averages are the ones in the *_stats directories (rocprofv3 --stats) and in bench.py's own HIP-event timing."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
KERNELS = ("nsff_field_kernel_h3a_save", "nsff_field_kernel_h3a", "nsff_field_kernel_h3", "nsff_field_kernel(", "mpi_composite_kernel", "composite_bwd_kernel", "composite_kernel",
           "fine_samples_kernel", "coarse_samples_kernel", "warp_points_kernel", "splat_tiles_kernel", "splat_far_kernel",
           "splat_scan_kernel", "splat_bin_kernel", "splat_gather_kernel", "frustum_visibility_kernel",
           "nsff_field_bwd_kernel_h3b", "nsff_field_bwd_kernel", "nsff_wgrad_kernel", "nsff_wgrad_head_kernel", "nsff_wgrad_accumulate_kernel", "nsff_wgrad_reduce_kernel",
           "fold_grads_kernel", "fold_dense_head_kernel", "fold_dense_final_kernel", "nsff_time_bias_kernel", "nsff_side_bias_kernel", "field_input_bwd_kernel", "loss_rays_kernel",
           "loss_select_kernel", "loss_stats2_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            extra = ""
            if k == "nsff_field_kernel_h3" and "<" in name:
                extra = name[name.index("<"):name.index(">") + 1]
            if k == "nsff_wgrad_kernel":
                extra = name[name.index("<"):name.index(">") + 1]
            return k.rstrip("(") + extra
    return None


def load(dirname):
    """{kernel: {counter: [sum, dispatches]}} and {kernel: [duration sum ns, n]} per pass"""
    out = {}
    for path in sorted(glob.glob(os.path.join(root, dirname, "*_counter_collection.csv"))):
        tag = os.path.basename(path).split("_counter_collection")[0]
        cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                c = cnt[k][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"])
                c[1].add(row["Dispatch_Id"])
        dur = collections.defaultdict(lambda: [0.0, 0])
        tr = path.replace("_counter_collection.csv", "_kernel_trace.csv")
        if os.path.exists(tr):
            with open(tr) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if k is not None:
                        dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        dur[k][1] += 1
        out[tag] = (cnt, dur)
    return out


def stats_table(dirname):
    for path in glob.glob(os.path.join(root, dirname, "**", "*kernel_stats.csv"), recursive=True):
        print(f"## {dirname}: rocprofv3 --kernel-trace --stats (un-counted run)")
        with open(path) as f:
            for row in list(csv.DictReader(f))[:18]:
                print(f"  {int(row['Calls']):5d} x  avg {float(row['AverageNs']) / 1e3:9.1f} us  {float(row['Percentage']):6.2f} %  {row['Name'][:110]}")


for d in ("bench_stats", "eval_stats", "train_stats", "readme_stats", "readme_train_stats", "interp_stats"):
    stats_table(d)
for d in ("bench_pmc", "interp_pmc", "eval_pmc", "train_pmc", "readme_pmc"):
    passes = load(d)
    if not passes:
        continue
    print(f"\n## {d}: counters per dispatch (mean over the dispatches of each kernel)")
    kernels = sorted({k for cnt, _ in passes.values() for k in cnt})
    for k in kernels:
        print(f"### {k}")
        vals = {}
        for tag, (cnt, dur) in passes.items():
            if k not in cnt:
                continue
            d_ns = dur[k][0] / max(dur[k][1], 1)
            line = [f"  [{tag}] avg duration in this pass {d_ns / 1e3:.1f} us;"]
            for name, (tot, disp) in sorted(cnt[k].items()):
                m = tot / max(len(disp), 1)
                vals[name] = m
                vals[name + "@dur"] = d_ns
                line.append(f"{name}={m:.6g}")
            print(" ".join(line))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
            dur_s = 0.5 * (vals["FETCH_SIZE@dur"] + vals["WRITE_SIZE@dur"]) * 1e-9
            print(f"  => HBM bytes per launch: read {rd / 1e6:.2f} MB (2 x FETCH_SIZE) + write {wr / 1e6:.2f} MB = {(rd + wr) / 1e6:.2f} MB;"
                  f" {(rd + wr) / dur_s / 1e12:.3f} TB/s over the counted passes' average duration = {(rd + wr) / dur_s / 8e12 * 100:.1f} % of the 8 TB/s HBM peak")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
            busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024)
            clk = vals["GRBM_GUI_ACTIVE"] / 8 / (vals["GRBM_GUI_ACTIVE@dur"] * 1e-9) / 1e9
            print(f"  => matrix pipe busy {busy * 100:.1f} % of the SIMD-cycles of the launch (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); shader clock in this pass {clk:.2f} GHz")
        if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            print(f"  => LDS bank-conflict cycles / LDS active cycles = {vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE'] * 100:.1f} %")
        if "TCC_HIT_sum" in vals:
            print(f"  => L2 hit rate {vals['TCC_HIT_sum'] / max(vals['TCC_HIT_sum'] + vals['TCC_MISS_sum'], 1) * 100:.2f} %")
