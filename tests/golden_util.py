"""Loader for tests/golden/decode_*.npz (written by oracle/make_golden.py from
the real reference).  Pure numpy; never touches /root/reference."""
import glob
import hashlib
import os

import numpy as np

from ganet_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix="decode_"):
    return sorted(os.path.basename(p)[len(prefix):-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_decode_case(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, "decode_%s.npz" % name),
                     allow_pickle=False))
    H, W = int(d["H"]), int(d["W"])
    if "heat" not in d:
        heat, off, err = synth.decode_inputs(1, H, W, lanes=int(d["gen_lanes"]),
                                             seed=int(d["gen_seed"]))
        d.update(heat=heat, offset=off, error=err)
    got = _sha(d["heat"], d["offset"], d["error"])
    assert got == str(d["input_sha256"]), (
        "synthetic generator no longer reproduces the inputs of golden case %s" % name)
    d["by_center_thr"] = float(d["T"]) if int(d["by_center"]) else None
    return d


def lanes_to_arrays(lanes, cluster_centers):
    """Flatten PostProcessor output for exact comparison (dtype-checked)."""
    ids = np.array([l["id_class"] for l in lanes], np.int64)
    sizes = np.array([len(l["points"]) for l in lanes], np.int64)
    pts, ctr = [], []
    for l in lanes:
        for p in l["points"]:
            assert all(isinstance(v, np.float32) for v in p), "points must be np.float32"
            pts.append(p)
        for c in l["centers"]:
            assert all(isinstance(v, np.float64) for v in c), "centers must be np.float64"
            ctr.append(c)
    cc = []
    for i, c in enumerate(cluster_centers):
        assert c["id_class"] == i
        assert all(isinstance(v, np.integer) for v in c["center"])
        cc.append(c["center"])
    return (ids, sizes, np.asarray(pts, np.float32).reshape(-1, 2),
            np.asarray(ctr, np.float64).reshape(-1, 2),
            np.asarray(cc, np.int64).reshape(-1, 2))


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s shape %s != %s" % (what, a.shape, b.shape)
    assert a.dtype == b.dtype, "%s dtype %s != %s" % (what, a.dtype, b.dtype)
    assert a.tobytes() == b.tobytes(), "%s differs bitwise" % what
