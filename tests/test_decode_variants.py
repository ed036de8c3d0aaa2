"""The two decode variants that BASELINE.json's north star names and the reference's code
does not have (3x3 heat-map NMS; nearest instead of first group centre): non-default flags of
ganet_decode_batch.  PARITY UNPINNED by the reference - they are pinned by torch's own
max_pool2d (NMS) and by a NumPy restatement (assignment) on the oracle, and by the oracle
on the GPU."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as O


def rand_planes(rng, H, W, plateau=True):
    heat = rng.uniform(0, 1, (H, W)).astype(np.float32)
    if plateau:
        heat = np.round(heat * 12) / 12          # many exact ties -> plateaus
    off = rng.normal(0, 3, (2, H, W)).astype(np.float32)
    err = rng.uniform(0, 1, (2, H, W)).astype(np.float32)
    return heat.astype(np.float32), off, err


@pytest.mark.parametrize("H,W", [(7, 9), (40, 100), (1, 5), (5, 1)])
def test_oracle_nms_3x3_is_torch_max_pool(H, W):
    rng = np.random.default_rng(H * 100 + W)
    heat, off, err = rand_planes(rng, H, W)
    if H > 2 and W > 2:
        heat[1, 1] = np.nan
        heat[H - 1, W - 1] = np.inf
    t = torch.from_numpy(heat)[None, None]
    for rows, kernel, pad in ((1, (1, 3), (0, 1)), (3, (3, 3), (1, 1))):
        hmax = torch.nn.functional.max_pool2d(t, kernel, stride=(1, 1), padding=pad)
        nms = (t * (hmax == t).float())[0, 0]
        thr = 0.3
        want = torch.nonzero(nms > thr).numpy()          # row-major (y, x)
        c, kxy, al, rt = O.decode(heat, off, err, thr, 1.0, nms_rows=rows)
        assert np.array_equal(kxy[:, ::-1], want)
    # 3x3 keeps a subset of what 1x3 keeps
    k1 = {tuple(v) for v in O.decode(heat, off, err, 0.3, 1.0, nms_rows=1)[1]}
    k3 = {tuple(v) for v in O.decode(heat, off, err, 0.3, 1.0, nms_rows=3)[1]}
    assert k3 <= k1


def test_oracle_nearest_assignment():
    rng = np.random.default_rng(4)
    centers = np.array([[10, 10], [18, 10], [40, 10], [12, 30]], np.int64)  # balls of 0 and 1 overlap
    root = np.concatenate([rng.uniform(5, 23, (60, 2)), rng.uniform(0, 50, (40, 2)),
                           [[15, 10], [16, 11.5], [14.5, 9]]]).astype(np.float32)  # in both balls
    T = 6.0
    gid_first, means, cg, ng = O.group_fast(centers, root, T, T)
    gid_near, means2, cg2, ng2 = O.group_fast(centers, root, T, T, nearest=True)
    assert ng == ng2 and np.array_equal(means, means2) and np.array_equal(cg, cg2)
    d = np.sqrt(((root[:, None, :].astype(np.float64) - means[None].astype(np.float64)) ** 2).sum(-1))
    inside = d <= T
    want_first = np.where(inside.any(1), inside.argmax(1), -1)
    dn = np.where(inside, d, np.inf)
    want_near = np.where(inside.any(1), dn.argmin(1), -1)
    assert np.array_equal(gid_first, want_first)
    assert np.array_equal(gid_near, want_near)
    assert (gid_first != gid_near).any()                 # the variants really differ here
    assert np.array_equal(gid_first >= 0, gid_near >= 0)  # same keypoints kept


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(40, 100), (96, 130), (1, 7)])   # LDS bitmap / mask kernel
@pytest.mark.parametrize("nms3,nearest", [(True, False), (False, True), (True, True)])
def test_gpu_variants_match_oracle(H, W, nms3, nearest):
    from ganet_amd import decode
    rng = np.random.default_rng(H + W)
    B = 3
    planes = [rand_planes(rng, H, W) for _ in range(B)]
    heat = np.stack([p[0] for p in planes])[:, None]
    off = np.stack([p[1] for p in planes])
    err = np.stack([p[2] for p in planes])
    thr, T = 0.85, 5.0
    res = decode.decode_batch(torch.from_numpy(heat).cuda(), torch.from_numpy(off).cuda(),
                              torch.from_numpy(err).cuda(), thr=thr, root_thr=1,
                              cluster_thr=T, cluster_by_center_thr=T, nms_3x3=nms3,
                              assign_nearest=nearest)
    for b in range(B):
        c, kxy, al, rt = O.decode(heat[b], off[b], err[b], thr, 1.0, nms_rows=3 if nms3 else 1)
        gid, means, cgid, ng = O.group_fast(c, rt, T, T, nearest=nearest)
        img = res.image(b)
        assert np.array_equal(img["centers"], c)
        assert img["align"].tobytes() == al.tobytes()
        assert img["root"].tobytes() == rt.tobytes()
        assert np.array_equal(img["kpt_gid"], gid)
        assert np.array_equal(img["group_center"], means)
