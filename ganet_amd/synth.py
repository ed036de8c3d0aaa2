"""Seeded synthetic inputs of the shapes the hot path sees.

Pure numpy, no dependency on the HIP library or on the oracle: tests, the
golden-fixture script and bench.py all draw their inputs from here so that
every party runs on identical tensors.

Channel conventions follow the reference:
  * deform-conv offsets: channel 2k = y (h) offset, 2k+1 = x (w) offset of
    tap k (mmdet/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu:222-225);
  * decode: offset[0] = dx, offset[1] = dy to the lane's start point, and
    error[0], error[1] = sub-pixel x, y
    (mmdet/datasets/pipelines/lane_formating.py:515-516).
"""
import math

import numpy as np

# name -> (H, W, K, kpt_thr, root_thr, cluster_thr T, downscale, lanes)
SHAPES = {
    # GANet-S/M TuSimple 320x800 at stride 8 (configs/tusimple/final_exp_res18_s8.py)
    "s8_tusimple": dict(H=40, W=100, K=7, thr=0.3, root_thr=1.0, T=4, ds=8, lanes=4),
    # GANet-S/M CULane 320x800 at stride 8 (configs/culane/test_common_s8.py)
    "s8_culane": dict(H=40, W=100, K=7, thr=0.4, root_thr=1.0, T=5, ds=8, lanes=4),
    # GANet-L at stride 4 (configs/culane/final_exp_res101_s4.py)
    "s4_culane": dict(H=80, W=200, K=9, thr=0.3, root_thr=1.0, T=10, ds=4, lanes=4),
    "s4_tusimple": dict(H=80, W=200, K=9, thr=0.3, root_thr=1.0, T=6, ds=4, lanes=4),
    # dense stress: 1280x3200 frames
    "dense_s8": dict(H=160, W=400, K=7, thr=0.05, root_thr=1.0, T=4, ds=8, lanes=16),
    "dense_s4": dict(H=320, W=800, K=9, thr=0.05, root_thr=1.0, T=6, ds=4, lanes=16),
}


def lfa_points(B, H, W, K, seed=0):
    """pts_out_init [B,2K,H,W] (y,x pairs, pixels relative to the output
    location) and dcn_offset = pts - base with base = [0,-p, 0,-p+1, ...]
    (mmdet/models/dense_heads/lanepoints_conv.py:63-74,170)."""
    p = (K - 1) // 2
    base = np.zeros((2 * K, 1, 1), dtype=np.float32)
    base[1::2, 0, 0] = np.arange(-p, p + 1, dtype=np.float32)
    pts = np.empty((B, 2 * K, H, W), dtype=np.float32)
    for b in range(B):
        rng = np.random.default_rng([seed + b, 1])
        for k in range(K):
            pts[b, 2 * k] = rng.normal(0.0, 1.5, (H, W))
            pts[b, 2 * k + 1] = ((k - p) * rng.uniform(0.5, 1.5, (H, W))
                                 + rng.normal(0.0, 0.5, (H, W)))
    return pts, pts - base[None]


def lfa_weight(Cin, K, Cout=64, seed=0):
    """weight [Cout,Cin,1,K] ~ U(+-1/sqrt(Cin*K))  (mmdet/ops/dcn/deform_conv.py:294-299)."""
    rng = np.random.default_rng([seed, 2])
    bound = 1.0 / math.sqrt(Cin * K)
    return rng.uniform(-bound, bound, (Cout, Cin, 1, K)).astype(np.float32)


def lfa_inputs(B, Cin, H, W, K, Cout=64, seed=0, dtype=np.float32):
    """x [B,Cin,H,W], dcn_offset [B,2K,H,W], weight [Cout,Cin,1,K], pts [B,2K,H,W]."""
    x = np.empty((B, Cin, H, W), dtype=np.float32)
    for b in range(B):
        x[b] = np.random.default_rng([seed + b, 0]).standard_normal(
            (Cin, H, W), dtype=np.float32)
    pts, off = lfa_points(B, H, W, K, seed)
    weight = lfa_weight(Cin, K, Cout, seed)
    return x.astype(dtype), off, weight.astype(dtype), pts


def decode_inputs(B, H, W, lanes=4, seed=0, noise=0.05):
    """heat [B,1,H,W], offset [B,2,H,W], error [B,2,H,W], all float32.

    Lane-like heat-maps: each lane is a line from a start point on the bottom
    rows with slope U(-0.4, 0.4); the start-point offsets are exact within
    +-1 px of a lane and N(0,5) elsewhere.
    """
    heats, offs, errs = [], [], []
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32),
                         np.arange(W, dtype=np.float32), indexing="ij")
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        heat = np.zeros((H, W), dtype=np.float32)
        off = rng.normal(0.0, 5.0, (2, H, W)).astype(np.float32)
        # keep the background from accidentally looking like a start row
        off[1] = np.abs(off[1]) + 2.0
        for l in range(lanes):
            x_start = rng.uniform(0.1 * W, 0.9 * W)
            y_start = H - 1 - rng.integers(0, 3)
            slope = rng.uniform(-0.4, 0.4)
            length = rng.integers(max(2, H // 2), H)
            x_l = x_start + slope * (y_start - yy)
            on = (yy <= y_start) & (yy > y_start - length)
            g = 0.95 * np.exp(-0.5 * (xx - x_l) ** 2) * on
            heat = np.maximum(heat, g.astype(np.float32))
            near = on & (np.abs(xx - x_l) <= 1.0)
            sx = np.float32(round(float(x_start)))
            off[0][near] = (sx - xx)[near]
            off[1][near] = (np.float32(y_start) - yy)[near]
        heat = heat + rng.uniform(0.0, noise, (H, W)).astype(np.float32)
        heat = np.clip(heat, 1e-4, 1 - 1e-4).astype(np.float32)
        err = rng.uniform(0.0, 1.0, (2, H, W)).astype(np.float32)
        heats.append(heat[None])
        offs.append(off)
        errs.append(err)
    return (np.stack(heats).astype(np.float32), np.stack(offs).astype(np.float32),
            np.stack(errs).astype(np.float32))
