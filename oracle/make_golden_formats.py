"""TEST INFRASTRUCTURE ONLY - writes tests/golden/formats.json from the REAL reference.

Run in the build container (needs /root/reference):
    python oracle/make_golden_formats.py

For a set of the decode golden cases it runs the reference chain
    ktdet_decode_fast -> PostProcessor -> adjust_result -> tusimple_convert_formal / out_result
(ganet_head.py:203-238, post_process.py:197-262, tusimple/test_dataset.py:49-84,
culane/test_dataset.py:79-148, common.py:118-147), imported where it lies by
oracle/ref_import.py, and stores the parameters and the expected outputs as JSON.
Inputs are the planes of tests/golden/decode_<case>.npz.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT
sys.path.insert(1, os.path.join(ROOT, "tests"))

from oracle import ref_import  # noqa: E402
import golden_util  # noqa: E402

# dataset conventions of the reference's test scripts
TUSIMPLE = dict(crop_bbox=(0, 160, 1280, 720), img_shape=(320, 800, 3), points_thr=4,
                h_samples=list(range(160, 720, 10)), im_width=1280)
CULANE = dict(crop_bbox=(0, 270, 1640, 590), img_shape=(320, 800, 3), points_thr=4)

CASES = [
    ("s8_tusimple_seed0", "tusimple", 8), ("s8_tusimple_seed1", "tusimple", 8),
    ("s8_tusimple_seed2", "tusimple", 8), ("s4_tusimple_seed1", "tusimple", 4),
    ("s8_culane_seed3", "culane", 8), ("s4_culane_seed0", "culane", 4),
    ("s8_tusimple_seed5_nocenter", "tusimple", 8),
    ("random_0", "culane", 8), ("random_3_nocenter", "tusimple", 8),
    ("edge_triangle_single_groups", "culane", 8), ("edge_empty", "tusimple", 8),
]


def main():
    F = ref_import.load_reference_formats()
    out = []
    for name, dataset, ds in CASES:
        d = golden_util.load_decode_case(name)
        cpt, kpt = ref_import.reference_decode(
            torch.from_numpy(d["heat"]), torch.from_numpy(d["offset"]),
            torch.from_numpy(d["error"]), thr=float(d["thr"]), root_thr=float(d["root_thr"]))
        lanes, centers = ref_import.reference_post_process(
            cpt, kpt, ds, float(d["T"]), d["by_center_thr"], group_fast=True)
        cfg = TUSIMPLE if dataset == "tusimple" else CULANE
        adjust = F["adjust_result_" + dataset]
        result, vcs, ccs = adjust(lanes=lanes, centers=centers, crop_bbox=cfg["crop_bbox"],
                                  img_shape=cfg["img_shape"], points_thr=cfg["points_thr"])
        rec = dict(case=name, dataset=dataset, downscale=ds,
                   crop_bbox=list(cfg["crop_bbox"]), img_shape=list(cfg["img_shape"]),
                   points_thr=cfg["points_thr"],
                   result=[[[int(v) for v in p] for p in lane] for lane in result],
                   virtual_centers=[[[int(v) for v in p] for p in lane] for lane in vcs],
                   cluster_centers=[[int(v) for v in c] for c in ccs])
        for lane in result:
            for p in lane:
                assert all(type(v) is int for v in p)
        if dataset == "tusimple":
            rec["h_samples"] = cfg["h_samples"]
            rec["im_width"] = cfg["im_width"]
            rec["tusimple_lanes"] = F["tusimple_convert_formal"](
                result, cfg["h_samples"], cfg["im_width"])
        else:
            with tempfile.TemporaryDirectory() as tmp:
                dst = os.path.join(tmp, "x.lines.txt")
                F["out_result"](result, dst=dst)
                rec["lines_txt"] = open(dst).read()
        out.append(rec)
        print(name, dataset, "lanes", len(result), "centres", len(ccs))
    # hand-made lines for get_line_intersection (end points, horizontal and reversed
    # segments, rounding)
    lines = [
        [(10, 100), (20, 110), (35, 120), (35, 130)],
        [(5.5, 300), (5.5, 300), (9.25, 290), (100.75, 200)],
        [(0, 10), (7, 10), (14, 20)],
        [(-3.2, 50), (4.9, 40)],
    ]
    ys = [90, 100, 105, 110, 115, 120, 125, 130, 131, 10, 15, 20, 40, 45, 50, 200, 250, 290, 295, 300]
    inter = [[F["get_line_intersection"](y, ln, 1280) for y in ys] for ln in lines]
    doc = dict(numpy=np.__version__, cases=out,
               intersections=dict(lines=lines, ys=ys, expected=inter))
    with open(os.path.join(ROOT, "tests", "golden", "formats.json"), "w") as f:
        json.dump(doc, f)
    print("wrote formats.json")


if __name__ == "__main__":
    main()
