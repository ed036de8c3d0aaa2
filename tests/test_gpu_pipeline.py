"""HotPath (ganet_amd/pipeline.py): the allocation-free LFA + decode step bench.py times,
checked against the CPU oracle; the pinned-host pack, the device pack + copy and a HIP
graph replay of the step must all give the same result."""
import numpy as np
import pytest
import torch

from ganet_amd import synth
from ganet_amd.pipeline import HotPath
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def build(B, dtype, **kw):
    cfg = synth.SHAPES["s8_culane"]
    H, W, K = cfg["H"], cfg["W"], cfg["K"]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu")
    g.manual_seed(3)
    x = torch.randn((B, 256, H, W), generator=g).to(dev).to(dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    pts, _ = synth.lfa_points(B, H, W, K, seed=3)
    w = synth.lfa_weight(256, K, 64, seed=3)
    heat, off, err = synth.decode_inputs(B, H, W, lanes=cfg["lanes"], seed=3)
    T = float(cfg["T"])
    hp = HotPath(B, H, W, Cin=256, Cout=64, K=K, dtype=dtype, device=dev, thr=cfg["thr"],
                 root_thr=cfg["root_thr"], cluster_thr=T, by_center_thr=T, max_kpts=1024,
                 max_centers=256, **kw)
    hp.set_weight(torch.from_numpy(w).to(dev))
    d = dict(x=x, pts=torch.from_numpy(pts).to(dev), heat=torch.from_numpy(heat).to(dev),
             off=torch.from_numpy(off).to(dev), err=torch.from_numpy(err).to(dev),
             heat_h=heat, off_h=off, err_h=err, pts_h=pts, w_h=w, cfg=cfg, T=T)
    return hp, d


def check_decode(res, d, B):
    cfg, T = d["cfg"], d["T"]
    for b in range(B):
        c, kxy, al, rt = O.decode(d["heat_h"][b], d["off_h"][b], d["err_h"][b], cfg["thr"],
                                  cfg["root_thr"])
        gid, means, cgid, ng = O.group_fast(c, rt, T, T)
        img = res.image(b)
        assert np.array_equal(img["centers"], c)
        assert img["align"].tobytes() == al.tobytes()
        assert img["root"].tobytes() == rt.tobytes()
        assert np.array_equal(img["kpt_gid"], gid)
        assert np.array_equal(img["group_center"], means)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 1e-4)])
def test_step_matches_oracle(dtype, tol):
    B = 6
    hp, d = build(B, dtype)
    hp.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    res = hp.result()
    check_decode(res, d, B)
    K = d["cfg"]["K"]
    base = np.zeros((2 * K, 1, 1), np.float32)
    base[1::2, 0, 0] = np.arange(-(K - 1) // 2, (K - 1) // 2 + 1)
    xr = d["x"].float().cpu().numpy()
    wr = torch.from_numpy(d["w_h"]).to(dtype).float().numpy()
    ref = np.maximum(O.deform_conv_forward(xr, d["pts_h"] - base[None], wr, (1, 1),
                                           (0, (K - 1) // 2), (1, 1), nthreads=8), 0)
    got = hp.feature.float().cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() <= tol


def test_pack_paths_and_graph_replay_agree():
    B = 5
    hp, d = build(B, torch.bfloat16)
    hp.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    r_host = hp.result()
    feat = hp.feature.clone()
    hp2, _ = build(B, torch.bfloat16, pack_to_host=False, expect_kpts=8)  # forces the top-up copy
    hp2.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    r_dev = hp2.result()
    assert np.array_equal(r_host.header, r_dev.header)
    assert np.array_equal(r_host.records, r_dev.records)
    # graph replay, twice, with the inputs changed in place between the replays
    graph = hp.capture(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    hp.feature.zero_()
    graph.replay()
    r_g = hp.result()
    assert np.array_equal(r_host.header, r_g.header)
    assert np.array_equal(r_host.records, r_g.records)
    assert torch.equal(hp.feature, feat)
    # the same step recorded as two branches (LFA | decode) of one graph
    hp3, _ = build(B, torch.bfloat16)
    forked = hp3.capture(d["x"], d["pts"], d["heat"], d["off"], d["err"], fork=True)
    hp3.feature.zero_()
    forked.replay()
    r_f = hp3.result()
    assert np.array_equal(r_host.header, r_f.header) and np.array_equal(r_host.records, r_f.records)
    assert torch.equal(hp3.feature, feat)
    d["heat"].mul_(0.45)    # the weaker lane points drop below the threshold
    graph.replay()
    r_g2 = hp.result()
    forked.replay()
    r_f2 = hp3.result()
    assert np.array_equal(r_g2.header, r_f2.header) and np.array_equal(r_g2.records, r_f2.records)
    hp2.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    r_d2 = hp2.result()
    assert np.array_equal(r_g2.header, r_d2.header)
    assert np.array_equal(r_g2.records, r_d2.records)
    assert int(r_g2.header[:, 0].sum()) < int(r_host.header[:, 0].sum())


def test_decode_beside_the_lfa_on_a_second_stream():
    """side_decode: decode/grouping/pack run on a stream of their own; several steps back to
    back (the streams drift apart) must still give the one-stream result, also when the
    inputs of a later step were produced on the current stream just before it."""
    B = 7
    hp, d = build(B, torch.bfloat16)
    hp.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    r1 = hp.result()
    feat = hp.feature.clone()
    hs, _ = build(B, torch.bfloat16, side_decode=True)
    for _ in range(5):
        hs.step(d["x"], d["pts"], d["heat"], d["off"], d["err"])
    rs = hs.result()
    assert np.array_equal(r1.header, rs.header)
    assert np.array_equal(r1.records, rs.records)
    torch.cuda.synchronize()
    assert torch.equal(hs.feature, feat)
    check_decode(rs, d, B)
    heat2 = d["heat"] * 0.45          # produced on the current stream, consumed on the side one
    hs.step(d["x"], d["pts"], heat2, d["off"], d["err"])
    r2s = hs.result()
    hp.step(d["x"], d["pts"], heat2, d["off"], d["err"])
    r2 = hp.result()
    assert np.array_equal(r2.header, r2s.header)
    assert np.array_equal(r2.records, r2s.records)
    assert int(r2.header[:, 0].sum()) < int(r1.header[:, 0].sum())


def test_step_on_logits_matches_step_on_probabilities():
    """heat_is_logits: the step takes the head's raw keypoint map; the records must equal
    those of the step fed with torch's own clamp(sigmoid(.)) of the same logits
    (ganet_head.py:291)."""
    B = 6
    hp, d = build(B, torch.bfloat16)
    hl, _ = build(B, torch.bfloat16, heat_is_logits=True)
    logit = torch.logit(d["heat"].double()).float()
    prob = torch.clamp(logit.sigmoid(), min=1e-4, max=1 - 1e-4)
    hp.step(d["x"], d["pts"], prob, d["off"], d["err"])
    rp = hp.result()
    hl.step(d["x"], d["pts"], logit, d["off"], d["err"])
    rl = hl.result()
    assert np.array_equal(rp.header, rl.header)
    assert np.array_equal(rp.records, rl.records)
    assert int(rl.header[:, 0].sum()) > 50 * B
