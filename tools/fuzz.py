#!/usr/bin/env python3
"""Differential fuzzing of the HIP library against the CPU oracle on random small problems
(tools/ only; the oracle is the checker).  Usage, on the GPU box:

    PYTHONPATH=. python tools/fuzz.py --seed 0 --forward 120 --backward 60 --decode 150

Every case prints one line; a mismatch prints FAIL with the parameters and the exit code is 1.
A configuration the library refuses with GANET_ERR_UNSUPPORTED is counted, not failed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ganet_amd import _lib, dcn  # noqa: E402
from ganet_amd import decode as D  # noqa: E402
from ganet_amd import formats as F  # noqa: E402
from oracle import cpu_oracle as O  # noqa: E402

DEV = torch.device("cuda", 0)
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}
# half an ulp of the blended value plus half an ulp of the output, against the sum of |terms|
EPS = {torch.float32: 1e-5, torch.bfloat16: 4e-3, torch.float16: 5e-4}


def rnd_conv(rng):
    kH, kW = (1, int(rng.choice([1, 3, 5, 7, 9]))) if rng.random() < 0.5 else \
        (int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 3])))
    dg = int(rng.choice([1, 1, 1, 2, 4]))
    groups = int(rng.choice([1, 1, 1, 2]))
    unit = dg * groups
    mode = rng.random()
    if mode < 0.3:
        Cin = unit * int(rng.integers(1, 9))
    elif mode < 0.6:
        Cin = int(rng.choice([32, 64, 128, 192, 256]))
        Cin = max(unit, Cin // unit * unit)
    else:
        Cin = unit * int(rng.integers(1, 70))
    Cout = groups * int(rng.choice([1, 3, 8, 16, 32, 35, 64])) if rng.random() < 0.7 else \
        groups * int(rng.integers(1, 70))
    st = (int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2])))
    dil = (int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2])))
    pad = (int(rng.integers(0, 3)), int(rng.integers(0, 5)))
    H = int(rng.integers(1, 20)) + dil[0] * (kH - 1)
    W = int(rng.integers(1, 28)) + dil[1] * (kW - 1)
    B = int(rng.integers(1, 5))
    return dict(B=B, Cin=Cin, H=H, W=W, Cout=Cout, kH=kH, kW=kW, st=st, pad=pad, dil=dil, dg=dg,
                groups=groups)


def tensors(rng, c, modulated):
    Ho = O.dcn_out_size(c["H"], c["pad"][0], c["dil"][0], c["kH"], c["st"][0])
    Wo = O.dcn_out_size(c["W"], c["pad"][1], c["dil"][1], c["kW"], c["st"][1])
    x = rng.standard_normal((c["B"], c["Cin"], c["H"], c["W"])).astype(np.float32)
    w = (rng.standard_normal((c["Cout"], c["Cin"] // c["groups"], c["kH"], c["kW"])) * 0.3).astype(np.float32)
    scale = float(rng.choice([0.0, 0.5, 2.0, 8.0, 40.0]))
    off = (rng.standard_normal((c["B"], c["dg"] * 2 * c["kH"] * c["kW"], Ho, Wo)) * scale).astype(np.float32)
    if rng.random() < 0.3:      # integer and half-integer positions, the strict bounds
        off = np.round(off * 2) / 2
    mask = rng.uniform(0, 1, (c["B"], c["dg"] * c["kH"] * c["kW"], Ho, Wo)).astype(np.float32) \
        if modulated else None
    bias = rng.standard_normal(c["Cout"]).astype(np.float32) if modulated and rng.random() < 0.7 else None
    return x, off, w, mask, bias, Ho, Wo


def fuzz_forward(rng, n):
    bad = unsupported = 0
    why = {}
    for i in range(n):
        c = rnd_conv(rng)
        modulated = rng.random() < 0.4
        dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
        nhwc = rng.random() < 0.5
        relu = rng.random() < 0.3
        algo = int(rng.choice([0, 0, 0, 1, 2, 5]))
        x, off, w, mask, bias, Ho, Wo = tensors(rng, c, modulated)
        if Ho < 1 or Wo < 1:
            continue
        tx = torch.from_numpy(x).to(DEV).to(dtype)
        tw = torch.from_numpy(w).to(DEV).to(dtype)
        tb = None if bias is None else torch.from_numpy(bias).to(DEV).to(dtype)
        if nhwc:
            tx = tx.contiguous(memory_format=torch.channels_last)
        tag = "fwd %3d %s dtype=%s nhwc=%d mod=%d relu=%d algo=%d" % (
            i, c, str(dtype).split(".")[1], nhwc, modulated, relu, algo)
        try:
            out = dcn.deform_conv_forward(
                tx, torch.from_numpy(off).to(DEV), tw, c["st"], c["pad"], c["dil"], c["groups"], c["dg"],
                mask=None if mask is None else torch.from_numpy(mask).to(DEV), bias=tb,
                fuse_relu=relu, algo=algo)
        except _lib.GanetError as e:
            if e.status == _lib.GANET_ERR_UNSUPPORTED and (algo == 2 or c["dg"] > 1):
                unsupported += 1
                k = "algo 2 forced" if algo == 2 else "deformable groups %d%s" % (
                    c["dg"], " (algo 5 forced)" if algo == 5 else "")
                why[k] = why.get(k, 0) + 1
                continue
            print("FAIL (status %d) %s" % (e.status, tag))
            bad += 1
            continue
        ref = O.deform_conv_forward(tx.float().cpu().numpy(), off, tw.float().cpu().numpy(), c["st"],
                                    c["pad"], c["dil"], c["groups"], c["dg"], mask=mask,
                                    bias=None if tb is None else tb.float().cpu().numpy(), nthreads=8)
        # the tolerance is relative to the scale of the sums BEFORE the ReLU: with a bias that
        # pushes most outputs below zero the surviving values are small against the rounding
        # of the terms they are made of
        scale = max(float(np.abs(ref).max()), 1e-6)
        if relu:
            ref = np.maximum(ref, 0)
        got = out.float().cpu().numpy()
        ok = got.shape == ref.shape and np.isfinite(got).all() and \
            float(np.abs(got - ref).max()) <= TOL[dtype] * scale
        if not ok and got.shape == ref.shape and np.isfinite(got).all():
            # a handful of outputs can all be small by cancellation; the rounding of the
            # 16-bit blend is bounded by the sum of the magnitudes of the terms instead
            mag = O.deform_conv_forward(np.abs(tx.float().cpu().numpy()), off,
                                        np.abs(tw.float().cpu().numpy()), c["st"], c["pad"], c["dil"],
                                        c["groups"], c["dg"], mask=mask,
                                        bias=None if tb is None else np.abs(tb.float().cpu().numpy()),
                                        nthreads=8)
            ok = float(np.abs(got - ref).max()) <= EPS[dtype] * float(np.abs(mag).max())
        if not ok:
            err = float(np.abs(got - ref).max()) / scale if got.shape == ref.shape else -1
            print("FAIL (err %.3g) %s" % (err, tag))
            bad += 1
    print("forward: %d cases, %d unsupported %s, %d failed" % (n, unsupported, why, bad), flush=True)
    return bad


def fuzz_backward(rng, n):
    bad = unsupported = 0
    for i in range(n):
        c = rnd_conv(rng)
        modulated = rng.random() < 0.4
        nhwc = rng.random() < 0.5
        algo = int(rng.choice([0, 0, 1, 2]))
        x, off, w, mask, bias, Ho, Wo = tensors(rng, c, modulated)
        if Ho < 1 or Wo < 1 or c["H"] < c["kH"] or c["W"] < c["kW"]:
            continue
        go = rng.standard_normal((c["B"], c["Cout"], Ho, Wo)).astype(np.float32)
        tx, toff, tw, tgo = (torch.from_numpy(a).to(DEV) for a in (x, off, w, go))
        tm = None if mask is None else torch.from_numpy(mask).to(DEV)
        if nhwc:
            tx = tx.contiguous(memory_format=torch.channels_last)
            tgo = tgo.contiguous(memory_format=torch.channels_last)
        tag = "bwd %3d %s nhwc=%d mod=%d algo=%d" % (i, c, nhwc, modulated, algo)
        keep = {}
        try:
            if rng.random() < 0.5:
                dcn.deform_conv_forward(tx, toff, tw, c["st"], c["pad"], c["dil"], c["groups"], c["dg"],
                                        mask=tm, keep_y=keep)
            g = dcn.deform_conv_backward(tx, toff, tw, tgo, c["st"], c["pad"], c["dil"], c["groups"],
                                         c["dg"], mask=tm, need_bias=modulated, algo=algo,
                                         y_saved=keep.get("y"), overwrite=bool(rng.random() < 0.5))
        except _lib.GanetError as e:
            if e.status == _lib.GANET_ERR_UNSUPPORTED:
                unsupported += 1
                continue
            print("FAIL (status %d) %s" % (e.status, tag))
            bad += 1
            continue
        ref = O.deform_conv_backward(x, off, w, go, c["st"], c["pad"], c["dil"], c["groups"], c["dg"],
                                     mask=mask)
        names = ["grad_x", "grad_offset", "grad_mask", "grad_weight", "grad_bias"]
        for t, nm in zip(g, names):
            if t is None:
                continue
            r = ref[nm]
            a = t.float().cpu().numpy()
            scale = max(float(np.abs(r).max()), 1e-6)
            if a.shape != r.shape or not np.isfinite(a).all() or \
                    float(np.abs(a - r).max()) > 2e-4 * scale:
                err = float(np.abs(a - r).max()) / scale if a.shape == r.shape else -1
                print("FAIL %s (err %.3g) %s" % (nm, err, tag))
                bad += 1
                break
    print("backward: %d cases, %d unsupported, %d failed" % (n, unsupported, bad), flush=True)
    return bad


def fuzz_decode(rng, n):
    bad = 0
    for i in range(n):
        B = int(rng.integers(1, 6))
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 130))
        if rng.random() < 0.15:
            H, W = int(rng.integers(90, 170)), int(rng.integers(100, 410))   # the large-image kernels
        dens = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
        heat = rng.uniform(0, 1, (B, 1, H, W)).astype(np.float32)
        heat *= (rng.random((B, 1, H, W)) < dens)
        if rng.random() < 0.3:     # plateaus: equal neighbours
            heat = np.round(heat * 4) / 4
        off = (rng.standard_normal((B, 2, H, W)) * float(rng.choice([0.5, 3.0, 30.0]))).astype(np.float32)
        err = (rng.standard_normal((B, 2, H, W)) * 0.5).astype(np.float32)
        thr = float(rng.choice([0.1, 0.3, 0.5, 0.25]))
        root_thr = float(rng.choice([1.0, 0.5, 2.0]))
        T = float(rng.choice([2.0, 4.0, 5.0, 10.0]))
        by_center = rng.random() < 0.7
        nms3, nearest = bool(rng.random() < 0.2), bool(rng.random() < 0.2)
        tag = "dec %3d B=%d %dx%d dens=%.2f thr=%.2f root=%.1f T=%.0f by_center=%d nms3=%d nearest=%d" % (
            i, B, H, W, dens, thr, root_thr, T, by_center, nms3, nearest)
        try:
            res = D.decode_batch(torch.from_numpy(heat).to(DEV), torch.from_numpy(off).to(DEV),
                                 torch.from_numpy(err).to(DEV), thr=thr, root_thr=root_thr,
                                 cluster_thr=T, cluster_by_center_thr=T if by_center else None,
                                 nms_3x3=nms3, assign_nearest=nearest)
        except RuntimeError as e:
            if "exceed the device limit" in str(e):
                continue
            print("FAIL (%r) %s" % (e, tag))
            bad += 1
            continue
        # the same decode fed with logits (flags bit 3) against torch's own sigmoid + clamp
        if rng.random() < 0.3:
            lg = torch.from_numpy((rng.standard_normal((B, 1, H, W)) * 3).astype(np.float32)).to(DEV)
            pr = torch.clamp(lg.sigmoid(), min=1e-4, max=1 - 1e-4)
            kw = dict(thr=thr, root_thr=root_thr, cluster_thr=T,
                      cluster_by_center_thr=T if by_center else None)
            try:
                r1 = D.decode_batch(lg, torch.from_numpy(off).to(DEV), torch.from_numpy(err).to(DEV),
                                    logits=True, **kw)
                r2 = D.decode_batch(pr, torch.from_numpy(off).to(DEV), torch.from_numpy(err).to(DEV), **kw)
                if not (np.array_equal(r1.header[:, :3], r2.header[:, :3])
                        and np.array_equal(r1.records, r2.records)):
                    print("FAIL (logits) %s" % tag)
                    bad += 1
            except RuntimeError as e:
                if "exceed the device limit" not in str(e):
                    print("FAIL (logits %r) %s" % (e, tag))
                    bad += 1
        # the C++ host formatting against its per-image Python mirror
        if by_center and rng.random() < 0.5:
            crop = (0, int(rng.integers(0, 300)), int(rng.integers(800, 1700)), int(rng.integers(400, 800)))
            shape, pthr, ds = (H * 8, W * 8, 3), int(rng.integers(0, 5)), int(rng.choice([4, 8]))
            try:
                bl = F.format_batch(res.packed(), B, True, ds, crop, shape, pthr, threads=int(rng.integers(1, 5)))
                txt = bl.culane_texts()
                for b in range(B):
                    ref_l = F.image_lanes(res.image(b), ds, crop, shape, pthr)
                    if bl.image(b) != ref_l or txt[b] != F.culane_lines_text(ref_l[0]):
                        print("FAIL (formats, image %d) %s" % (b, tag))
                        bad += 1
                        break
            except (ValueError, OverflowError):
                pass     # a non-finite coordinate: both sides refuse (checked in tests/test_formats.py)
        for b in range(B):
            c, kxy, al, rt = O.decode(heat[b], off[b], err[b], thr, root_thr, nms_rows=3 if nms3 else 1)
            gid, means, cgid, ng = O.group_fast(c, rt, T, T if by_center else None, nearest=nearest)
            img = res.image(b)
            ok = (np.array_equal(img["centers"], c) and np.array_equal(img["kpt_xy"], kxy)
                  and np.asarray(img["align"]).tobytes() == al.tobytes()
                  and np.asarray(img["root"]).tobytes() == rt.tobytes()
                  and np.array_equal(img["kpt_gid"], gid) and img["n_groups"] == ng
                  and np.array_equal(img["group_center"].reshape(-1, 2), means))
            if not ok:
                print("FAIL (image %d) %s" % (b, tag))
                bad += 1
                break
    print("decode: %d cases, %d failed" % (n, bad), flush=True)
    return bad


def fuzz_lfa(rng, n):
    """LFA-shaped problems at sizes that walk the batched kernels: several chunks, ragged
    tiles, every algorithm that applies, the fused offset subtraction."""
    bad = 0
    for i in range(n):
        Cin = int(rng.choice([64, 128, 256]))
        Cout = int(rng.choice([64, 64, 128])) if Cin == 256 else 64
        K = int(rng.choice([5, 7, 9]))
        H, W, B = int(rng.integers(1, 48)), int(rng.integers(K, 110)), int(rng.integers(1, 40))   # W >= K: the reference refuses narrower maps
        dtype = [torch.float32, torch.bfloat16, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
        algos = [0, 1, 5] + ([2] if 2 * Cout <= Cin else []) + \
            ([3] if 2 * Cout <= Cin and Cout == 64 and dtype != torch.float32 else []) + \
            ([4] if 2 * Cout <= Cin and Cout == 64 and dtype != torch.float32 and H * W <= 4096 else [])
        algo = int(rng.choice(algos))
        nhwc, relu, points = rng.random() < 0.7, rng.random() < 0.5, rng.random() < 0.5
        x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 1, K)) / np.sqrt(Cin * K)).astype(np.float32)
        off = (rng.standard_normal((B, 2 * K, H, W)) * float(rng.choice([0.5, 1.5, 6.0]))).astype(np.float32)
        base = np.zeros((1, 2 * K, 1, 1), np.float32)
        base[0, 1::2, 0, 0] = np.arange(-(K - 1) // 2, (K - 1) // 2 + 1)
        tx = torch.from_numpy(x).to(DEV).to(dtype)
        tw = torch.from_numpy(w).to(DEV).to(dtype)
        if nhwc:
            tx = tx.contiguous(memory_format=torch.channels_last)
        tag = "lfa %3d B=%d Cin=%d Cout=%d K=%d %dx%d dtype=%s nhwc=%d relu=%d points=%d algo=%d" % (
            i, B, Cin, Cout, K, H, W, str(dtype).split(".")[1], nhwc, relu, points, algo)
        given = off + base if points else off
        try:
            out = dcn.deform_conv_forward(tx, torch.from_numpy(given).to(DEV), tw, (1, 1),
                                          (0, (K - 1) // 2), (1, 1), fuse_relu=relu,
                                          offset_is_points=points, algo=algo, check=True)
        except Exception as e:  # noqa: BLE001
            print("FAIL (%r) %s" % (e, tag))
            bad += 1
            continue
        ref = O.deform_conv_forward(tx.float().cpu().numpy(), off, tw.float().cpu().numpy(), (1, 1),
                                    (0, (K - 1) // 2), (1, 1), nthreads=16)
        scale = max(float(np.abs(ref).max()), 1e-6)     # before the ReLU, as above
        if relu:
            ref = np.maximum(ref, 0)
        got = out.float().cpu().numpy()
        err = float(np.abs(got - ref).max()) / scale
        if not np.isfinite(got).all() or err > TOL[dtype]:
            print("FAIL (err %.3g) %s" % (err, tag))
            bad += 1
    print("lfa: %d cases, %d failed" % (n, bad), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--forward", type=int, default=100)
    ap.add_argument("--backward", type=int, default=50)
    ap.add_argument("--decode", type=int, default=100)
    ap.add_argument("--lfa", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    bad = fuzz_forward(rng, args.forward) + fuzz_backward(rng, args.backward) + \
        fuzz_decode(rng, args.decode) + fuzz_lfa(rng, args.lfa)
    print("fuzz seed %d: %d failures in %.0f s" % (args.seed, bad, time.time() - t0))
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
