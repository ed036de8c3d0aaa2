"""Result formatting after grouping: ganet_amd/formats.py against
tests/golden/formats.json, which oracle/make_golden_formats.py wrote by running the
reference's own adjust_result / tusimple_convert_formal / out_result on the decode golden
cases.  CPU tests feed the mirror from the C oracle (itself pinned bit-exact to the
reference by test_oracle_golden.py); the GPU test feeds it from the HIP decode."""
import copy
import json
import os

import numpy as np
import pytest

from ganet_amd import formats
from oracle import cpu_oracle as O

import golden_util as G

DOC = json.load(open(os.path.join(G.GOLDEN_DIR, "formats.json")))
CASES = {c["case"]: c for c in DOC["cases"]}


def as_lists(lanes):
    return [[[int(v) for v in p] for p in lane] for lane in lanes]


def oracle_image(d):
    """Decode + group one golden case with the CPU oracle; returns the dict layout of
    DecodeResult.image() and the PostProcessor-shaped (lanes, cluster_centers)."""
    T = float(d["T"])
    centers, kxy, align, root = O.decode(d["heat"][0], d["offset"][0], d["error"][0],
                                         float(d["thr"]), float(d["root_thr"]))
    gid, means, cgid, ng = O.group_fast(centers, root, T, d["by_center_thr"])
    img = dict(kpt_gid=gid, n_groups=ng, align=align, root=root, group_center=means)
    return img


def check_against_golden(c, result, vcs, ccs):
    assert as_lists(result) == c["result"]
    assert as_lists(vcs) == c["virtual_centers"]
    assert [[int(v) for v in p] for p in ccs] == c["cluster_centers"]
    for lane in result:
        for p in lane:
            assert isinstance(p, tuple) and all(type(v) is int for v in p)
    if c["dataset"] == "tusimple":
        got = formats.tusimple_convert_formal(result, c["h_samples"], c["im_width"])
        assert got == c["tusimple_lanes"]
        line = json.loads(formats.tusimple_json_line(result, c["h_samples"], "a/b.jpg",
                                                     c["im_width"]))
        assert line == dict(lanes=c["tusimple_lanes"], h_samples=c["h_samples"],
                            raw_file="a/b.jpg", run_time=20)
    else:
        assert formats.culane_lines_text(result) == c["lines_txt"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_adjust_result_chain_matches_reference(name, tmp_path):
    c = CASES[name]
    d = G.load_decode_case(name)
    img = oracle_image(d)
    lanes, centers = O.lanes_from_groups(img["kpt_gid"], img["n_groups"], img["align"],
                                         img["root"], img["group_center"], c["downscale"])
    keep = copy.deepcopy(lanes)
    result, vcs, ccs = formats.adjust_result(
        lanes=lanes, centers=centers, crop_bbox=tuple(c["crop_bbox"]),
        img_shape=tuple(c["img_shape"]), points_thr=c["points_thr"])
    check_against_golden(c, result, vcs, ccs)
    # like the reference, the call overwrites the coordinates inside its inputs
    if keep and keep[0]["points"]:
        assert type(lanes[0]["points"][0][0]) is int
    # array form: same answer without the per-point Python objects
    r2, v2, c2 = formats.image_lanes(img, c["downscale"], tuple(c["crop_bbox"]),
                                     tuple(c["img_shape"]), c["points_thr"])
    check_against_golden(c, r2, v2, c2)
    if c["dataset"] == "culane":
        dst = tmp_path / "out.lines.txt"
        formats.out_result(result, dst=str(dst))
        assert dst.read_text() == c["lines_txt"]
        formats.out_result(result, dst=None)  # no file, no error


def test_get_line_intersection_matches_reference():
    t = DOC["intersections"]
    for line, exp in zip(t["lines"], t["expected"]):
        line = [tuple(p) for p in line]
        assert [formats.get_line_intersection(y, line, 1280) for y in t["ys"]] == exp
    assert formats.get_line_intersection(5, [], 1280) == -2
    assert formats.get_line_intersection(5, [(1, 5)], 1280, reg_x=-7) == -7


def test_image_lanes_rejects_non_finite_like_int():
    img = dict(kpt_gid=np.array([0, 0], np.int32), n_groups=1,
               align=np.array([[1, 2], [np.nan, 3]], np.float32),
               root=np.zeros((2, 2), np.float32), group_center=np.zeros((0, 2), np.int32))
    with pytest.raises(ValueError):
        formats.image_lanes(img, 8, (0, 160, 1280, 720), (320, 800, 3), 0)
    img["align"][1, 0] = np.inf
    with pytest.raises(OverflowError):
        formats.image_lanes(img, 8, (0, 160, 1280, 720), (320, 800, 3), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_decode_to_result_files(name):
    import torch
    from ganet_amd import decode as D
    from ganet_amd.post_process import PostProcessor
    c = CASES[name]
    d = G.load_decode_case(name)
    dev = torch.device("cuda", 0)
    heat, off, err = (torch.from_numpy(d[k]).to(dev) for k in ("heat", "offset", "error"))
    res = D.decode_batch(heat, off, err, thr=float(d["thr"]), root_thr=float(d["root_thr"]),
                         cluster_thr=float(d["T"]), cluster_by_center_thr=d["by_center_thr"])
    img = res.image(0)
    r2, v2, c2 = formats.image_lanes(img, c["downscale"], tuple(c["crop_bbox"]),
                                     tuple(c["img_shape"]), c["points_thr"])
    check_against_golden(c, r2, v2, c2)
    pp = PostProcessor(cluster_thr=float(d["T"]), cluster_by_center_thr=d["by_center_thr"],
                       group_fast=True)
    lanes, centers = pp(res, c["downscale"])
    result, vcs, ccs = formats.adjust_result(lanes, centers, tuple(c["crop_bbox"]),
                                             tuple(c["img_shape"]), c["points_thr"])
    check_against_golden(c, result, vcs, ccs)


# ---- whole batches through the library's host routines (ganet_lanes_from_packed & co) ----

def _batch_from_golden(names):
    imgs = [oracle_image(G.load_decode_case(n)) for n in names]
    return imgs


@pytest.mark.parametrize("dataset", ["tusimple", "culane"])
def test_format_batch_matches_reference_golden(dataset):
    """Every golden case of one dataset as ONE batch: the C++ routines give the reference's
    adjust_result / tusimple_convert_formal / out_result answers image by image."""
    names = [n for n in sorted(CASES) if CASES[n]["dataset"] == dataset]
    groups = {}
    for n in names:      # one call per (downscale, crop, shape, thr, grouping mode) setting
        c, d = CASES[n], G.load_decode_case(n)
        key = (c["downscale"], tuple(c["crop_bbox"]), tuple(c["img_shape"]), c["points_thr"],
               d["by_center_thr"] is not None)
        groups.setdefault(key, []).append(n)
    assert groups
    for (ds, crop, shape, thr, by_center), members in groups.items():
        imgs = _batch_from_golden(members)
        packed = formats.packed_from_images(imgs, by_center=by_center)
        for threads in (1, 3):
            bl = formats.format_batch(packed, len(imgs), by_center, ds, crop, shape, thr,
                                      threads=threads)
            texts = bl.culane_texts()
            for b, n in enumerate(members):
                c = CASES[n]
                r, v, cc = bl.image(b)
                check_against_golden(c, r, v, cc)
                if dataset == "culane":
                    assert texts[b] == c["lines_txt"]
                else:
                    xs, start = bl.tusimple(c["h_samples"])
                    assert xs[start[b]:start[b + 1]].tolist() == c["tusimple_lanes"]
                    line = json.loads(bl.tusimple_json_lines(c["h_samples"],
                                                             ["a/b.jpg"] * len(members))[b])
                    assert line["lanes"] == c["tusimple_lanes"]


def test_format_batch_equals_python_mirror_on_random_images():
    """Random grouped images (negative and large coordinates, empty images, dropped
    keypoints, groups at the points_thr boundary): batch routine == image_lanes."""
    rng = np.random.default_rng(0)
    imgs = []
    for b in range(37):
        nk = int(rng.integers(0, 60)) if b % 5 else 0
        ng = int(rng.integers(0, 7)) if nk else 0
        gid = rng.integers(-1, max(ng, 1), nk).astype(np.int32) if ng else -np.ones(nk, np.int32)
        imgs.append(dict(kpt_gid=gid, n_groups=ng,
                         align=(rng.standard_normal((nk, 2)) * 300).astype(np.float32),
                         root=(rng.standard_normal((nk, 2)) * 300).astype(np.float32),
                         group_center=rng.integers(-50, 400, (ng, 2)).astype(np.int32)))
    crop, shape = (0, 270, 1640, 590), (320, 800, 3)
    for thr in (0, 3):
        bl = formats.format_batch(formats.packed_from_images(imgs), len(imgs), True, 8, crop,
                                  shape, thr, threads=4)
        for b, im in enumerate(imgs):
            assert bl.image(b) == formats.image_lanes(im, 8, crop, shape, thr)
        txt = bl.culane_texts()
        for b, im in enumerate(imgs):
            assert txt[b] == formats.culane_lines_text(formats.image_lanes(im, 8, crop, shape, thr)[0])


def test_format_batch_rejects_non_finite_like_int():
    img = dict(kpt_gid=np.array([0, 0], np.int32), n_groups=1,
               align=np.array([[1, 2], [np.nan, 3]], np.float32),
               root=np.zeros((2, 2), np.float32), group_center=np.zeros((1, 2), np.int32))
    with pytest.raises(ValueError):
        formats.format_batch(formats.packed_from_images([img]), 1, True, 8,
                             (0, 160, 1280, 720), (320, 800, 3), 0)


def test_lanes_from_packed_legacy_float64_scalars():
    """float32_scalars = 0: NumPy 1.x (the reference's own environment) promotes
    `np.float32 * python_float` to float64; checked against that formula written out."""
    import ctypes
    from ganet_amd import _lib
    rng = np.random.default_rng(3)
    nk = 40
    img = dict(kpt_gid=np.zeros(nk, np.int32), n_groups=1,
               align=(rng.standard_normal((nk, 2)) * 77).astype(np.float32),
               root=(rng.standard_normal((nk, 2)) * 77).astype(np.float32),
               group_center=np.array([[5, 9]], np.int32))
    packed = formats.packed_from_images([img])
    f = _lib.LaneFormat()
    f.B, f.by_center, f.downscale, f.points_thr, f.float32_scalars, f.threads = 1, 1, 8, 0, 0, 1
    f.img_h, f.img_w = 320, 800
    f.crop_x0, f.crop_y0, f.crop_x1, f.crop_y1 = 0.0, 270.0, 1640.0, 590.0
    f.packed_words = packed.size
    need = ctypes.c_size_t(0)
    L = _lib.lib()
    L.ganet_lanes_from_packed(packed.ctypes.data, ctypes.byref(f), None, 0, ctypes.byref(need))
    out = np.empty(need.value, np.int32)
    assert L.ganet_lanes_from_packed(packed.ctypes.data, ctypes.byref(f), out.ctypes.data,
                                     out.size, ctypes.byref(need)) == 0
    pts = out[4 + 2:4 + 2 + 2 * nk].reshape(nk, 2)
    rx, ry = 1640 / 800, (590 - 270) / 320
    a = (img["align"] * np.float32(8)).astype(np.float64)
    exp = np.stack([np.trunc(a[:, 0] * rx + 0), np.trunc(a[:, 1] * ry + 270)], 1).astype(np.int64)
    assert np.array_equal(pts, exp)


def test_lanes_from_packed_refuses_records_outside_the_buffer():
    """A record stream shorter than its headers claim (or with negative counts) is refused
    with GANET_ERR_BAD_SHAPE instead of being read past its end."""
    import ctypes
    from ganet_amd import _lib
    nk = 12
    img = dict(kpt_gid=np.zeros(nk, np.int32), n_groups=1,
               align=np.ones((nk, 2), np.float32), root=np.ones((nk, 2), np.float32),
               group_center=np.array([[5, 9]], np.int32))
    packed = formats.packed_from_images([img, img])
    L = _lib.lib()

    def call(buf, words):
        f = _lib.LaneFormat()
        f.B, f.by_center, f.downscale, f.points_thr, f.float32_scalars, f.threads = 2, 1, 8, 0, 1, 1
        f.img_h, f.img_w = 320, 800
        f.crop_x0, f.crop_y0, f.crop_x1, f.crop_y1 = 0.0, 270.0, 1640.0, 590.0
        f.packed_words = words
        need = ctypes.c_size_t(0)
        return L.ganet_lanes_from_packed(buf.ctypes.data, ctypes.byref(f), None, 0,
                                         ctypes.byref(need))

    assert call(packed, packed.size) == _lib.GANET_ERR_WORKSPACE      # the size query
    assert call(packed, packed.size - 1) == _lib.GANET_ERR_BAD_SHAPE  # last record cut short
    assert call(packed, 8) == _lib.GANET_ERR_BAD_SHAPE                # not even the headers
    neg = packed.copy()
    neg[0] = -5
    assert call(neg, neg.size) == _lib.GANET_ERR_BAD_SHAPE
    far = packed.copy()
    far[8 + 4] = 1 << 30                                               # record offset far outside
    assert call(far, far.size) == _lib.GANET_ERR_BAD_SHAPE


def test_format_batch_with_reused_buffers():
    """scratch=dict: the lane and text buffers are kept between calls (no size query, no
    allocation while they are large enough); same results as the allocating path, through
    growing and shrinking batches."""
    rng = np.random.default_rng(0)

    def img(nk, ng=3):
        return dict(kpt_gid=rng.integers(-1, ng, nk).astype(np.int32), n_groups=ng,
                    align=(rng.standard_normal((nk, 2)) * 50 + 100).astype(np.float32),
                    root=(rng.standard_normal((nk, 2)) * 50 + 100).astype(np.float32),
                    group_center=rng.integers(0, 100, (ng, 2)).astype(np.int32))

    scratch = {}
    args = (4, True, 8, (0, 270, 1640, 590), (320, 800, 3), 2)
    for nk in (30, 80, 10, 200, 5):
        packed = formats.packed_from_images([img(nk) for _ in range(4)])
        a = formats.format_batch(packed, *args)
        b = formats.format_batch(packed, *args, scratch=scratch)
        assert np.array_equal(a.stream, b.stream)
        texts = a.culane_texts()
        buf, start, length = b.culane_text_buffer(scratch=scratch)
        for i in range(4):
            assert bytes(buf[start[i]:start[i] + length[i]]).decode("ascii") == texts[i]
        assert [b.image(i) for i in range(4)] == [a.image(i) for i in range(4)]
    assert set(scratch) == {"lanes", "text"}
