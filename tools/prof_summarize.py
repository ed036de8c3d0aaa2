#!/usr/bin/env python3
"""Condenses the CSVs of tools/profile.sh into one text summary: per kernel the
dispatch count and average duration (kernel-trace stats) and the per-dispatch
average of every PMC counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("lfa_stream_kernel", "dcn_fused_kernel", "cts_gemm_split_kernel", "cts_gemm_ov_kernel", "cts_gemm_kernel", "pack_weight_split_kernel", "cts_sample_kernel",
                "decode_mask_kernel", "decode_group_kernel", "decode_pack_kernel",
                "nchw_to_nhwc_kernel", "nhwc_repack_kernel", "pack_weight_kernel"):
        if key in name:
            tail = ""
            if "dcn_fused" in name or "cts_" in name or "lfa_stream" in name:
                tail = "<" + ("bf16" if "bf16" in name else "f16" if "DF16" in name else "f32") + ">"
            return key + tail
    return name.split("<")[0][:60]


def main(root):
    print("# profile summary of", os.path.basename(root))
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("\n## rocprofv3 --kernel-trace --stats (per kernel)")
        print("%-42s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in csv.DictReader(open(f)):
            print("%-42s %8s %14s %12.1f %8s" % (
                short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                float(r["AverageNs"]), r["Percentage"]))
    print("\n## PMC counters, average per dispatch (sum over dispatches / dispatches)")
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k].add(r["Dispatch_Id"])
        print("\n### pass", os.path.basename(d))
        for k in sorted(acc):
            n = max(len(cnt[k]), 1)
            print("  %s  (dispatches %d)" % (k, n))
            for c in sorted(acc[k]):
                print("      %-36s %18.1f" % (c, acc[k][c] / n))


if __name__ == "__main__":
    main(sys.argv[1])
