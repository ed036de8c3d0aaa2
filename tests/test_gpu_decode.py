"""GPU parity: the HIP decode + grouping (through the C-ABI) against the golden
vectors of the real reference and against the CPU oracle.  Bit-exact."""
import numpy as np
import pytest
import torch

from ganet_amd import decode as D
from ganet_amd import synth
from ganet_amd.post_process import PostProcessor, lanes_from_image
from oracle import cpu_oracle as O
from tests.golden_util import (assert_bits_equal, golden_names, lanes_to_arrays,
                               load_decode_case)

pytestmark = pytest.mark.gpu


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check_image_against(img, centers, align, root, gid, means, ng):
    assert_bits_equal(img["centers"].astype(np.int64), centers, "centres")
    assert_bits_equal(np.array(img["align"]), align, "align")
    assert_bits_equal(np.array(img["root"]), root, "root")
    assert img["n_groups"] == ng
    assert_bits_equal(np.array(img["kpt_gid"]), gid, "kpt_gid")
    assert_bits_equal(img["group_center"].astype(np.int64).reshape(-1, 2), means,
                      "group means")
    counts = np.bincount(gid[gid >= 0], minlength=ng)
    assert np.array_equal(img["group_count"], counts)


@pytest.mark.parametrize("name", golden_names())
def test_decode_batch_matches_reference_golden(name):
    d = load_decode_case(name)
    res = D.decode_batch(cuda(d["heat"]), cuda(d["offset"]), cuda(d["error"]),
                         thr=float(d["thr"]), root_thr=float(d["root_thr"]),
                         cluster_thr=float(d["T"]),
                         cluster_by_center_thr=d["by_center_thr"])
    img = res.image(0)
    check_image_against(img, d["root_center_arr"], d["align"],
                        d["root"].astype(np.float32), d["kpt_gid"], d["group_means"],
                        int(d["n_groups"]))
    assert np.array_equal(np.array(img["root"]).astype(np.float64), d["root"])
    lanes, cc = lanes_from_image(img, int(d["ds"]))
    ids, sizes, pts, ctr, ccs = lanes_to_arrays(lanes, cc)
    assert_bits_equal(ids, d["lane_ids"], "lane ids")
    assert_bits_equal(sizes, d["lane_sizes"], "lane sizes")
    assert_bits_equal(pts, d["lane_points"], "lane points")
    assert_bits_equal(ctr, d["lane_centers"], "lane centres")
    assert_bits_equal(ccs, d["cluster_centers"], "cluster centres")


@pytest.mark.parametrize("name", ["s8_tusimple_seed0", "random_1", "random_1_nocenter",
                                  "edge_exact_distance_bridge", "edge_empty"])
def test_reference_call_sequence(name):
    """ktdet_decode_fast -> PostProcessor, as tools/ganet/*/test_dataset.py call
    them; structures and dtypes as the reference returns them."""
    d = load_decode_case(name)
    cpt, kpt = D.ktdet_decode_fast(cuda(d["heat"]), cuda(d["offset"]), cuda(d["error"]),
                                   thr=float(d["thr"]), root_thr=float(d["root_thr"]))
    assert isinstance(cpt, np.ndarray) and cpt.dtype == np.int64
    assert_bits_equal(cpt, d["root_center_arr"], "root_center_arr")
    assert isinstance(kpt, list) and len(kpt) == len(d["align"])
    for (a, r), ga, gr in zip(kpt, d["align"], d["root"]):
        assert a.dtype == np.float32 and r.dtype == np.float64
        assert a.tobytes() == ga.tobytes() and r.tobytes() == gr.tobytes()
    pp = PostProcessor(cluster_thr=float(d["T"]),
                       cluster_by_center_thr=d["by_center_thr"], group_fast=True)
    lanes, cc = pp((cpt, kpt), int(d["ds"]))
    ids, sizes, pts, ctr, ccs = lanes_to_arrays(lanes, cc)
    assert_bits_equal(ids, d["lane_ids"], "lane ids")
    assert_bits_equal(pts, d["lane_points"], "lane points")
    assert_bits_equal(ctr, d["lane_centers"], "lane centres")
    assert_bits_equal(ccs, d["cluster_centers"], "cluster centres")
    # seeds that only exist on the host (plain lists) are grouped on the GPU too
    lanes2, cc2 = pp((np.array(cpt), [(a, r) for a, r in kpt]), int(d["ds"]))
    ids2, sizes2, pts2, ctr2, ccs2 = lanes_to_arrays(lanes2, cc2)
    assert_bits_equal(ids2, d["lane_ids"], "lane ids (host seeds)")
    assert_bits_equal(pts2, d["lane_points"], "lane points (host seeds)")
    assert_bits_equal(ccs2, d["cluster_centers"], "cluster centres (host seeds)")


@pytest.mark.parametrize("shape,B,by_center", [("s8_culane", 128, True),
                                               ("s4_culane", 32, True),
                                               ("s4_culane", 128, True),   # C4's per-GPU share at 8 GPUs
                                               ("dense_s4", 4, True),
                                               ("dense_s8", 8, True),
                                               ("s8_tusimple", 16, False)])
def test_batch_against_oracle(shape, B, by_center):
    """BASELINE.json shapes at batch size, every image bit-exact vs the oracle."""
    cfg = synth.SHAPES[shape]
    heat, off, err = synth.decode_inputs(B, cfg["H"], cfg["W"], lanes=cfg["lanes"], seed=11)
    T = float(cfg["T"])
    res = D.decode_batch(cuda(heat), cuda(off), cuda(err), thr=cfg["thr"],
                         root_thr=cfg["root_thr"], cluster_thr=T,
                         cluster_by_center_thr=T if by_center else None)
    assert res.B == B
    for b in range(B):
        c, kxy, al, rt = O.decode(heat[b], off[b], err[b], cfg["thr"], cfg["root_thr"])
        gid, means, cgid, ng = O.group_fast(c, rt, T, T if by_center else None)
        img = res.image(b)
        check_image_against(img, c, al, rt, gid, means, ng)
        assert np.array_equal(img["kpt_xy"], kxy)
        if by_center:
            assert np.array_equal(img["center_gid"], cgid)


@pytest.mark.parametrize("H,W,thr", [(48, 96, 0.3), (96, 160, 0.85)])
def test_capacity_growth_and_noise_stress(H, W, thr):
    """Uniform-noise heat-maps with a low threshold: thousands of keypoints and
    hundreds of centres per image; the default capacities overflow and regrow.  48x96 takes
    the one-workgroup-per-image kernel, 96x160 (more than 8192 pixels) the gridded
    keypoint gather / assignment kernels around it."""
    rng = np.random.default_rng(5)
    B = 6
    heat = rng.uniform(0, 1, (B, 1, H, W)).astype(np.float32)
    off = rng.normal(0, 2.5, (B, 2, H, W)).astype(np.float32)
    err = rng.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    for by_center in (4.0, None):
        res = D.decode_batch(cuda(heat), cuda(off), cuda(err), thr=thr, root_thr=1,
                             cluster_thr=3.0, cluster_by_center_thr=by_center,
                             max_kpts=64, max_centers=16)
        for b in range(B):
            c, kxy, al, rt = O.decode(heat[b], off[b], err[b], thr, 1.0)
            gid, means, cgid, ng = O.group_fast(c, rt, 3.0, by_center)
            assert len(al) > 500
            check_image_against(res.image(b), c, al, rt, gid, means, ng)


def test_size_independent_properties_dense_s4():
    """Full-size dense frame (320x800, 16 lanes, thr 0.05): properties that hold
    whatever the size - ordering, idempotence, consistency of counts."""
    cfg = synth.SHAPES["dense_s4"]
    heat, off, err = synth.decode_inputs(2, cfg["H"], cfg["W"], lanes=16, seed=3)
    T = float(cfg["T"])
    args = dict(thr=cfg["thr"], root_thr=1.0, cluster_thr=T, cluster_by_center_thr=T)
    r1 = D.decode_batch(cuda(heat), cuda(off), cuda(err), **args)
    r2 = D.decode_batch(cuda(heat), cuda(off), cuda(err), **args)
    for b in range(2):
        a, c = r1.image(b), r2.image(b)
        for k in ("centers", "kpt_xy", "align", "root", "kpt_gid", "group_center"):
            assert np.array_equal(a[k], c[k])                       # deterministic
        kxy = a["kpt_xy"].astype(np.int64)
        lin = kxy[:, 1] * cfg["W"] + kxy[:, 0]
        assert np.all(np.diff(lin) > 0)                              # row-major, unique
        cx = a["centers"].astype(np.int64)
        assert np.all(np.diff(cx[:, 0] * cfg["H"] + cx[:, 1]) > 0)  # x-major, unique
        assert a["group_count"].sum() == (a["kpt_gid"] >= 0).sum()
        assert a["kpt_gid"].max() < a["n_groups"]
        # every keypoint is a strict 1x3 local max above the threshold
        h = heat[b, 0]
        v = h[kxy[:, 1], kxy[:, 0]]
        assert np.all(v > np.float32(cfg["thr"]))
    # and the oracle agrees on one full-size image
    c, kxy, al, rt = O.decode(heat[0], off[0], err[0], cfg["thr"], 1.0)
    gid, means, cgid, ng = O.group_fast(c, rt, T, T)
    check_image_against(r1.image(0), c, al, rt, gid, means, ng)


@pytest.mark.parametrize("shape,B", [("s8_culane", 16), ("dense_s8", 4)])
def test_logits_prologue_matches_torch_sigmoid_clamp(shape, B):
    """flags bit3: the decode reads the head's logits and applies
    clamp(sigmoid(.), 1e-4, 1 - 1e-4) (ganet_head.py:291) itself.  Checked bit-exact
    against the reference's own order on this GPU - torch's sigmoid and clamp, then the
    decode on probabilities - and against the oracle on those probabilities.  Small maps
    take the one-kernel form (plane staged in LDS), large ones decode_mask_kernel."""
    cfg = synth.SHAPES[shape]
    heat, off, err = synth.decode_inputs(B, cfg["H"], cfg["W"], lanes=cfg["lanes"], seed=5)
    logit = np.log(heat.astype(np.float64) / (1.0 - heat.astype(np.float64))).astype(np.float32)
    # saturated, clamped and non-finite logits in image 0
    logit[0, 0, 0, :8] = [100.0, -100.0, 20.0, -20.0, np.inf, -np.inf, np.nan, 9.3]
    logit[0, 0, 3, 5:9] = 12.0     # a plateau that the clamp flattens to 1 - 1e-4
    lg = cuda(logit)
    prob = torch.clamp(lg.sigmoid(), min=1e-4, max=1 - 1e-4)
    T = float(cfg["T"])
    kw = dict(thr=cfg["thr"], root_thr=cfg["root_thr"], cluster_thr=T, cluster_by_center_thr=T)
    got = D.decode_batch(lg, cuda(off), cuda(err), logits=True, **kw)
    ref = D.decode_batch(prob, cuda(off), cuda(err), **kw)
    ph = prob.cpu().numpy()
    total = 0
    for b in range(B):
        g, r = got.image(b), ref.image(b)
        for k in ("centers", "kpt_xy", "align", "root", "kpt_gid", "group_center", "group_count"):
            assert np.asarray(g[k]).tobytes() == np.asarray(r[k]).tobytes(), (b, k)
        c, kxy, al, rt = O.decode(ph[b], off[b], err[b], cfg["thr"], cfg["root_thr"])
        gid, means, cgid, ng = O.group_fast(c, rt, T, T)
        check_image_against(g, c, al, rt, gid, means, ng)
        total += len(kxy)
    assert total > 50 * B


@pytest.mark.parametrize("H,W", [(40, 100), (7, 9), (1, 5), (5, 1), (3, 16), (9, 17), (64, 128), (2, 33)])
@pytest.mark.parametrize("thr", [0.3, -0.5, 0.0])
def test_nms_special_values_match_oracle(H, W, thr):
    """The 1x3 NMS of the one-workgroup decode (16 pixels per thread, two comparisons per
    pixel) on the values where ATen's pool is particular: NaN, +-inf, +-0, plateaus, row ends,
    rows shorter than a thread's run, a threshold below zero (h * 0 = 0 passes it).  The oracle's
    NMS is pinned to torch.max_pool2d (tests/test_decode_variants.py)."""
    rng = np.random.default_rng(H * 1000 + W)
    B = 3
    heat = np.round(rng.uniform(-0.2, 1, (B, 1, H, W)) * 8).astype(np.float32) / 8
    flat = heat.reshape(B, -1)
    n = flat.shape[1]
    for b in range(B):
        for val in (np.nan, np.inf, -np.inf, -0.0, 0.0):
            flat[b, rng.integers(0, n, max(1, n // 40))] = val
        flat[b, 0] = (np.nan, np.inf, 0.5)[b]                  # first pixel of the image
        flat[b, n - 1] = (0.5, np.nan, -np.inf)[b]              # last pixel
        if H > 1:
            flat[b, W - 1] = np.nan                             # a row end beside a row start
            flat[b, W] = 1.0
    off = rng.normal(0, 3, (B, 2, H, W)).astype(np.float32)
    err = rng.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    res = D.decode_batch(cuda(heat), cuda(off), cuda(err), thr=thr, root_thr=-5.0, cluster_thr=4,
                         cluster_by_center_thr=4, group=False, max_kpts=H * W, max_centers=min(H * W, 2048))
    for b in range(B):
        c, kxy, al, rt = O.decode(heat[b, 0], off[b], err[b], thr, -5.0)
        img = res.image(b)
        assert np.array_equal(img["kpt_xy"], kxy)
        assert img["align"].tobytes() == al.tobytes() and img["root"].tobytes() == rt.tobytes()
        assert np.array_equal(img["centers"], c)
