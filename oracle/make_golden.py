"""TEST INFRASTRUCTURE ONLY - writes tests/golden/*.npz from the REAL reference.

Run in the build container (needs /root/reference):
    python oracle/make_golden.py

For every case it runs the reference's own `ktdet_decode_fast`
(mmdet/models/dense_heads/ganet_head.py:203-238) and `PostProcessor`
(tools/ganet/post_process.py:197-262), imported where they lie by
oracle/ref_import.py, and stores inputs + expected outputs as plain arrays
(no pickles).  Large synthetic inputs are stored as (generator, seed, sha256)
instead of raw planes; tests regenerate them with ganet_amd/synth.py and check
the hash.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT

from ganet_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
STORE_INPUT_LIMIT = 40 * 100  # planes larger than this are stored by seed+hash


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_reference(heat, offset, error, thr, root_thr, T, by_center, ds):
    """heat [1,1,H,W], offset/error [1,2,H,W] float32 numpy."""
    cpt, kpt = ref_import.reference_decode(
        torch.from_numpy(heat), torch.from_numpy(offset),
        torch.from_numpy(error), thr=thr, root_thr=root_thr)
    n = len(kpt)
    align = np.zeros((n, 2), np.float32)
    root = np.zeros((n, 2), np.float64)
    for i, (a, r) in enumerate(kpt):
        assert a.dtype == np.float32 and r.dtype == np.float64
        align[i], root[i] = a, r
    groups, means = ref_import.reference_group_points_fast(
        kpt, cpt, T, T if by_center else None)
    gid = np.full(n, -1, np.int32)
    for g, members in enumerate(groups):
        last = -1
        for m in members:
            assert m[0] > last  # members come in keypoint order
            last = m[0]
            gid[m[0]] = g
    lanes, centers = ref_import.reference_post_process(
        cpt, kpt, ds, T, T if by_center else None, group_fast=True)
    lane_ids = np.array([l["id_class"] for l in lanes], np.int64)
    lane_sizes = np.array([len(l["points"]) for l in lanes], np.int64)
    pts = [p for l in lanes for p in l["points"]]
    ctr = [c for l in lanes for c in l["centers"]]
    for p in pts:
        assert all(isinstance(v, np.float32) for v in p), [type(v) for v in p]
    for c in ctr:
        assert all(isinstance(v, np.float64) for v in c)
    out = dict(
        root_center_arr=np.asarray(cpt, np.int64).reshape(-1, 2),
        align=align, root=root, kpt_gid=gid,
        n_groups=np.int64(len(groups)),
        group_means=np.asarray(means, np.int64).reshape(-1, 2),
        lane_ids=lane_ids, lane_sizes=lane_sizes,
        lane_points=np.asarray(pts, np.float32).reshape(-1, 2),
        lane_centers=np.asarray(ctr, np.float64).reshape(-1, 2),
        cluster_centers=np.asarray([c["center"] for c in centers],
                                   np.int64).reshape(-1, 2),
    )
    return out


def save(name, heat, offset, error, thr, root_thr, T, by_center, ds,
         gen=None, note=""):
    out = run_reference(heat, offset, error, thr, root_thr, T, by_center, ds)
    H, W = heat.shape[-2:]
    rec = dict(out, H=np.int64(H), W=np.int64(W), thr=np.float64(thr),
               root_thr=np.float64(root_thr), T=np.float64(T),
               by_center=np.int64(1 if by_center else 0), ds=np.int64(ds),
               input_sha256=np.array(sha(heat, offset, error)),
               note=np.array(note))
    if gen is None or H * W <= STORE_INPUT_LIMIT:
        rec.update(heat=heat, offset=offset, error=error)
    if gen is not None:
        rec.update(gen_lanes=np.int64(gen["lanes"]), gen_seed=np.int64(gen["seed"]))
    path = os.path.join(GOLDEN, "decode_%s.npz" % name)
    np.savez_compressed(path, **rec)
    print("%-28s HxW=%dx%d kpts=%d centres=%d groups=%d lanes=%d  %s" % (
        name, H, W, len(out["align"]), len(out["root_center_arr"]),
        int(out["n_groups"]), len(out["lane_ids"]), note))


def blank(H, W, base=0.01):
    heat = np.full((1, 1, H, W), base, np.float32)
    offset = np.full((1, 2, H, W), 50.0, np.float32)  # far-away roots, no centres
    error = np.zeros((1, 2, H, W), np.float32)
    return heat, offset, error


def edge_cases():
    f32 = np.float32
    T = 4
    # --- plateau + exact threshold values + root_thr boundary -----------------
    heat, off, err = blank(8, 16)
    heat[0, 0, 1, 3:6] = 0.8                       # 3-wide plateau: all kept
    heat[0, 0, 2, 2] = f32(0.3)                    # == f32(thr): NOT a keypoint
    heat[0, 0, 2, 6] = np.nextafter(f32(0.3), f32(1))  # one ulp above: keypoint
    heat[0, 0, 3, 0] = 0.9                         # left border
    heat[0, 0, 3, 15] = 0.9                        # right border
    heat[0, 0, 0, 8] = 0.7                         # top border
    heat[0, 0, 7, 8] = 0.7                         # bottom border
    heat[0, 0, 5, 4:6] = [0.6, 0.9]                # neighbour suppressed
    heat[0, 0, 6, 10:13] = [0.5, 0.5, 0.6]         # plateau broken on the right
    off[0, 1, 3, 0] = 1.0                          # == root_thr: NOT a centre
    off[0, 1, 3, 15] = np.nextafter(f32(1.0), f32(0))  # just below: centre
    off[0, 0, 3, 15] = 0.0
    off[0, 1, 7, 8] = 0.25
    off[0, 0, 7, 8] = -0.5
    err[0, 0] = 0.25
    err[0, 1] = 0.75
    save("edge_plateau_thresholds", heat, off, err, 0.3, 1, T, True, 8,
         note="plateaus, ==thr, ==root_thr, borders")

    # --- distances exactly T, bridging centre, truncating mean ---------------
    heat, off, err = blank(16, 24)
    # three centres in x-major order A(2,4) B(3,10) C(4,7): C joins A's group
    for (x, y) in [(2, 4), (3, 10), (4, 7)]:
        heat[0, 0, y, x] = 0.9
        off[0, 0, y, x] = 0.0
        off[0, 1, y, x] = 0.0
    # an isolated centre group at (15,12),(15,13),(16,13): mean truncates
    for (x, y) in [(15, 12), (15, 13), (16, 13)]:
        heat[0, 0, y, x] = 0.9 - 0.01 * (x - 15)
        off[0, 0, y, x] = 0.0
        off[0, 1, y, x] = 0.0
    heat[0, 0, 13, 15] = 0.95
    # keypoints whose roots sit at distance exactly T / just beyond T from
    # the mean of the second cluster, and a 3-4-5 triangle with T = 5 is in
    # the next fixture
    kp = [(8, 2), (9, 3), (10, 5), (20, 1), (21, 2)]
    save_pts = []
    for i, (x, y) in enumerate(kp):
        heat[0, 0, y, x] = 0.85
        off[0, 1, y, x] = 5.0  # not a centre
        save_pts.append((x, y))
    # roots: filled after we know the means; the clusters' means are
    # (3,5) [A,C], (3,10) [B], (15,12) [trunc of 15.33, 12.67]
    def set_root(x, y, rx, ry):
        off[0, 0, y, x] = f32(rx - x)
        off[0, 1, y, x] = f32(ry - y)
    set_root(8, 2, 3 + 4, 5)          # distance exactly 4 from (3,5): grouped
    set_root(9, 3, 3, 10 + 4)         # exactly 4 from (3,10)
    set_root(10, 5, 15 + 4.000004, 12)   # just beyond 4 (survives f32 rounding)
    set_root(20, 1, 15, 12 - 4)       # exactly 4 from (15,12)
    set_root(21, 2, 100, 100)         # far away: dropped
    save("edge_exact_distance_bridge", heat, off, err, 0.3, 1, 4, True, 8,
         note="dist==T, bridging centre, truncated mean, dropped kpt")

    # --- 3-4-5 triangle with T=5, single-point groups -------------------------
    heat, off, err = blank(16, 24)
    heat[0, 0, 14, 10] = 0.9
    off[0, :, 14, 10] = 0.0           # centre (10,14)
    for i, (x, y, rx, ry) in enumerate([(3, 2, 13, 10), (4, 4, 7, 18),
                                        (6, 6, 10, 9), (8, 8, 10, 8.99)]):
        heat[0, 0, y, x] = 0.8
        off[0, 0, y, x] = f32(rx - x)
        off[0, 1, y, x] = f32(ry - y) if ry - y >= 1 else f32(ry - y)
    heat[0, 0, 1, 20] = 0.9           # a second centre with no other member
    off[0, :, 1, 20] = 0.0
    save("edge_triangle_single_groups", heat, off, err, 0.4, 1, 5, True, 8,
         note="3-4-5 distance == T, single-member group dropped from lanes")

    # --- nothing detected ------------------------------------------------------
    heat, off, err = blank(8, 16)
    save("edge_empty", heat, off, err, 0.3, 1, 4, True, 8, note="no detections")
    save("edge_empty_nocenter", heat, off, err, 0.3, 1, 4, False, 8,
         note="no detections, cluster_by_center_thr=None")

    # --- keypoints but no centres ---------------------------------------------
    heat, off, err = blank(8, 16)
    heat[0, 0, 2, 3] = 0.9
    heat[0, 0, 4, 9] = 0.9
    save("edge_kpts_no_centres", heat, off, err, 0.3, 1, 4, True, 8,
         note="keypoints without any centre: all dropped")

    # --- NaN / inf in the heat-map ----------------------------------------------
    heat, off, err = blank(8, 16)
    heat[0, 0, 1, 4] = np.nan
    heat[0, 0, 1, 5] = 0.9            # neighbour of NaN: pooled max is NaN
    heat[0, 0, 3, 7] = np.inf
    heat[0, 0, 3, 8] = 0.9            # suppressed by +inf
    heat[0, 0, 5, 2] = -np.inf
    heat[0, 0, 5, 3] = 0.9
    heat[0, 0, 6, 12] = 0.9
    off[0, :, 6, 12] = 0.0
    off[0, :, 3, 7] = 0.0
    save("edge_nan_inf", heat, off, err, 0.3, 1, 4, True, 8, note="NaN, +inf, -inf")

    # --- negative threshold: every suppressed pixel (0 > thr) is a keypoint -----
    rng = np.random.default_rng(7)
    heat = rng.uniform(0.0, 1.0, (1, 1, 6, 12)).astype(f32)
    off = rng.normal(0, 3, (1, 2, 6, 12)).astype(f32)
    err = rng.uniform(0, 1, (1, 2, 6, 12)).astype(f32)
    save("edge_negative_thr", heat, off, err, -0.5, 1, 3, True, 4,
         note="thr<0: all pixels are keypoints")
    save("edge_negative_thr_nocenter", heat, off, err, -0.5, 1, 3, False, 4,
         note="thr<0, cluster_by_center_thr=None")


def random_cases():
    f32 = np.float32
    for seed, (H, W, thr, T) in enumerate([(24, 40, 0.5, 4), (24, 40, 0.6, 3),
                                           (32, 64, 0.7, 5), (17, 33, 0.4, 6),
                                           (40, 100, 0.8, 4), (9, 130, 0.5, 2)]):
        rng = np.random.default_rng(100 + seed)
        heat = rng.uniform(0, 1, (1, 1, H, W)).astype(f32)
        off = rng.normal(0, 2.0, (1, 2, H, W)).astype(f32)
        err = rng.uniform(0, 1, (1, 2, H, W)).astype(f32)
        save("random_%d" % seed, heat, off, err, thr, 1, T, True, 8,
             note="uniform noise heat, N(0,2) offsets")
        if seed < 4:
            save("random_%d_nocenter" % seed, heat, off, err, thr, 1, T, False, 8,
                 note="same, cluster_by_center_thr=None")
    # integer-valued offsets: many distances land exactly on integers
    rng = np.random.default_rng(200)
    H, W = 24, 48
    heat = rng.uniform(0, 1, (1, 1, H, W)).astype(f32)
    off = rng.integers(-6, 7, (1, 2, H, W)).astype(f32)
    err = rng.integers(0, 2, (1, 2, H, W)).astype(f32)
    save("random_integer_offsets", heat, off, err, 0.6, 1, 5, True, 8,
         note="integer offsets: exact-boundary distances (3-4-5 etc.)")
    save("random_integer_offsets_nocenter", heat, off, err, 0.6, 1, 5, False, 8,
         note="integer offsets, cluster_by_center_thr=None")


def synthetic_cases():
    for name, seeds in [("s8_tusimple", (0, 1, 2)), ("s8_culane", (3,)),
                        ("s4_culane", (0,)), ("s4_tusimple", (1,)),
                        ("dense_s8", (0,))]:
        cfg = synth.SHAPES[name]
        for seed in seeds:
            heat, off, err = synth.decode_inputs(1, cfg["H"], cfg["W"],
                                                 lanes=cfg["lanes"], seed=seed)
            save("%s_seed%d" % (name, seed), heat, off, err, cfg["thr"],
                 cfg["root_thr"], cfg["T"], True, cfg["ds"],
                 gen=dict(lanes=cfg["lanes"], seed=seed),
                 note="synthetic lanes (ganet_amd/synth.py)")
    cfg = synth.SHAPES["s8_tusimple"]
    heat, off, err = synth.decode_inputs(1, cfg["H"], cfg["W"], lanes=4, seed=5)
    save("s8_tusimple_seed5_nocenter", heat, off, err, cfg["thr"], 1.0, 4, False, 8,
         gen=dict(lanes=4, seed=5), note="speed_test.py style (no centre clustering)")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    edge_cases()
    random_cases()
    synthetic_cases()
