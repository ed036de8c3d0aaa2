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
