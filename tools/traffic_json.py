#!/usr/bin/env python3
"""Writes one workload's entry of profiles/pmc_traffic.json from the PMC passes of
tools/profile.sh (gpurun_out/prof_<tag>/pmc_fetch, pmc_write):

    python tools/traffic_json.py <tag> <key> [steps_per_run]

bytes per LFA call = sum over the LFA's kernels of (2 * FETCH_SIZE + WRITE_SIZE) KiB, the
counters summed over the run's dispatches and divided by the number of LFA calls in the run
(= dispatches of the sampling / fused / persistent kernel divided by its launches per call).
The entry records where it came from: the summary file, `git rev-parse HEAD`, and a hash of
the kernel sources, which bench.py compares with the sources it runs on."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LFA_KERNELS = ("dcn_fused_kernel", "dcn_taps_kernel", "cts_gemm", "cts_sample_kernel",
               "lfa_stream_kernel", "lfa_ring_kernel")
KERNEL_SOURCES = ("deform_conv.hip", "deform_conv_cts.hip", "deform_conv_stream.hip",
                  "deform_conv_ring.hip", "dcn_device.h")


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "ganet_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def counter_sums(d, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in LFA_KERNELS:
                if k in r["Kernel_Name"]:
                    name = r["Kernel_Name"].split("<")[0].split("::")[-1]
                    tot[name] += float(r["Counter_Value"])
                    disp[name].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


def main():
    tag, key = sys.argv[1], sys.argv[2]
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    root = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    fetch, nf = counter_sums(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write, nw = counter_sums(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    if not fetch or not write:
        raise SystemExit("no LFA kernels in the PMC passes of " + root)
    if not calls:
        raise SystemExit("give the number of LFA calls of the profiled run (spin-up + warm-up + steps)")
    kib = sum(2.0 * fetch[k] / calls for k in fetch) + sum(write[k] / calls for k in write)
    entry = {
        "bytes": int(kib * 1024),
        "per_kernel_kib_per_call": {k: {"fetch_x2": 2.0 * fetch[k] / calls,
                                        "write": write.get(k, 0.0) / calls,
                                        "dispatches_per_call": nf[k] / calls} for k in sorted(fetch)},
        "formula": "sum over kernels of (2*FETCH_SIZE + WRITE_SIZE) KiB per call",
        "summary": "profiles/%s_rocprofv3_summary.txt" % tag,
        "head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True,
                               text=True).stdout.strip(),
        "kernel_sources": kernel_source_hash(),
    }
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = entry
    json.dump(doc, open(path, "w"), indent=1)
    print(key, json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
