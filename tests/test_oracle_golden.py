"""The C restatement (oracle/) against the golden vectors the REAL reference
produced (oracle/make_golden.py): bit-exact decode, grouping and lanes."""
import numpy as np
import pytest

from oracle import cpu_oracle as O
from tests.golden_util import (assert_bits_equal, golden_names, lanes_to_arrays,
                               load_decode_case)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_decode_matches_reference(name):
    d = load_decode_case(name)
    centers, kxy, align, root = O.decode(d["heat"], d["offset"][0], d["error"][0],
                                         float(d["thr"]), float(d["root_thr"]))
    assert_bits_equal(centers, d["root_center_arr"], "root_center_arr")
    assert_bits_equal(align, d["align"], "align")
    assert_bits_equal(root.astype(np.float64), d["root"], "root")
    gid, means, cgid, ng = O.group_fast(centers, root, float(d["T"]), d["by_center_thr"])
    assert ng == int(d["n_groups"])
    assert_bits_equal(gid, d["kpt_gid"], "kpt_gid")
    assert_bits_equal(means, d["group_means"], "group_means")
    lanes, cc = O.lanes_from_groups(gid, ng, align, root, means, int(d["ds"]))
    ids, sizes, pts, ctr, ccs = lanes_to_arrays(lanes, cc)
    assert_bits_equal(ids, d["lane_ids"], "lane ids")
    assert_bits_equal(sizes, d["lane_sizes"], "lane sizes")
    assert_bits_equal(pts, d["lane_points"], "lane points")
    assert_bits_equal(ctr, d["lane_centers"], "lane centers")
    assert_bits_equal(ccs, d["cluster_centers"], "cluster centers")


def test_square_vs_libm_pow():
    """The reference squares float64 differences with numpy's scalar `**`,
    i.e. libm pow(x, 2.0), which is NOT correctly rounded: it is 1 ulp off
    x*x for a fraction of a percent of operands whose square is inexact.  The
    oracle calls pow() like the reference; the HIP kernel multiplies.  A
    grouping decision can only differ when sqrt(dx^2+dy^2) lies within an ulp
    of the threshold without the squares being exact.  This test documents
    the rate and checks that (a) exactly representable squares agree and
    (b) no `<= T` decision flips on a large sample of the operand domain."""
    import math
    rng = np.random.default_rng(0)
    n = 100000
    root = (rng.integers(0, 800, (n, 2)).astype(np.float32)
            + rng.normal(0, 3, (n, 2)).astype(np.float32))
    c = root.astype(np.float64).round() + rng.integers(-6, 7, (n, 2))
    d = root.astype(np.float64) - c
    off = 0
    for T in (4.0, 5.0, 6.0, 10.0):
        for (dx, dy) in d[:25000]:
            a = math.sqrt(math.pow(dx, 2.0) + math.pow(dy, 2.0))
            b = math.sqrt(dx * dx + dy * dy)
            off += (a != b)
            assert (a <= T) == (b <= T)
    assert off < 0.02 * 4 * 25000
    # exactly representable squares (<= 26 significant bits) always agree
    xi = rng.integers(-(1 << 20), 1 << 20, 20000).astype(np.float64) / 1024.0
    for v in xi:
        assert math.pow(v, 2.0) == v * v
