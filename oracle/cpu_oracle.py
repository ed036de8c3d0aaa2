"""TEST INFRASTRUCTURE ONLY - ctypes front end of oracle/libganet_oracle.so.

The C file it loads (oracle/ganet_oracle.c) is the CPU restatement of the
reference's hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libganet_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ganet_oracle.c")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_dcn_out_size.restype = ctypes.c_int
        L.oracle_deform_conv_forward.restype = ctypes.c_int
        L.oracle_deform_conv_forward.argtypes = (
            [_f32p] * 6 + [ctypes.c_int] * 17)
        L.oracle_decode.restype = ctypes.c_int
        L.oracle_decode.argtypes = [
            _f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
            ctypes.c_float, _i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
            _i32p, _f32p, _f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.oracle_decode_opt.restype = ctypes.c_int
        L.oracle_decode_opt.argtypes = [
            _f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
            ctypes.c_float, ctypes.c_int, _i64p, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int), _i32p, _f32p, _f32p, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int)]
        L.oracle_group_fast_opt.restype = ctypes.c_int
        L.oracle_group_fast_opt.argtypes = [
            _i64p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_double,
            ctypes.c_double, ctypes.c_int, _i32p, _i64p, _i32p]
        L.oracle_group_fast.restype = ctypes.c_int
        L.oracle_group_fast.argtypes = [
            _i64p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_double,
            ctypes.c_double, _i32p, _i64p, _i32p]
        L.oracle_decode_group_batch.restype = ctypes.c_long
        L.oracle_decode_group_batch.argtypes = [
            _f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_double,
            ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else ctypes.cast(None, t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def dcn_out_size(size, pad, dil, k, stride):
    return lib().oracle_dcn_out_size(size, pad, dil, k, stride)


def deform_conv_forward(x, offset, weight, stride=(1, 1), padding=(0, 0),
                        dilation=(1, 1), groups=1, deformable_groups=1,
                        mask=None, bias=None, acc64=True, nthreads=1):
    """NCHW float32 deformable convolution (DCNv1; DCNv2 when mask is given)."""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    mask = _f32(mask) if mask is not None else None
    bias = _f32(bias) if bias is not None else None
    B, Cin, H, W = x.shape
    Cout, _, kH, kW = weight.shape
    Ho = dcn_out_size(H, padding[0], dilation[0], kH, stride[0])
    Wo = dcn_out_size(W, padding[1], dilation[1], kW, stride[1])
    if Ho < 1 or Wo < 1:
        raise ValueError("output too small")
    assert offset.shape == (B, deformable_groups * 2 * kH * kW, Ho, Wo), offset.shape
    out = np.empty((B, Cout, Ho, Wo), dtype=np.float32)
    rc = lib().oracle_deform_conv_forward(
        _p(x, _f32p), _p(offset, _f32p), _p(mask, _f32p), _p(weight, _f32p),
        _p(bias, _f32p), _p(out, _f32p), B, Cin, H, W, Cout, kH, kW,
        stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
        groups, deformable_groups, 1 if acc64 else 0, nthreads)
    if rc != 0:
        raise RuntimeError("oracle_deform_conv_forward failed: %d" % rc)
    return out


def deform_conv_backward(x, offset, weight, grad_out, stride=(1, 1), padding=(0, 0),
                         dilation=(1, 1), groups=1, deformable_groups=1, mask=None):
    """All gradients of the deformable convolution (NCHW float32); dict of arrays.
    grad_mask / grad_bias only with a mask (DCNv2)."""
    x, offset, weight, grad_out = _f32(x), _f32(offset), _f32(weight), _f32(grad_out)
    mask = _f32(mask) if mask is not None else None
    B, Cin, H, W = x.shape
    Cout, _, kH, kW = weight.shape
    gx = np.empty_like(x)
    goff = np.empty_like(offset)
    gw = np.empty_like(weight)
    gm = np.empty_like(mask) if mask is not None else None
    gb = np.empty(Cout, np.float32) if mask is not None else None
    L = lib()
    L.oracle_deform_conv_backward.restype = ctypes.c_int
    L.oracle_deform_conv_backward.argtypes = [_f32p] * 10 + [ctypes.c_int] * 15
    rc = L.oracle_deform_conv_backward(
        _p(x, _f32p), _p(offset, _f32p), _p(mask, _f32p), _p(weight, _f32p),
        _p(grad_out, _f32p), _p(gx, _f32p), _p(goff, _f32p), _p(gm, _f32p),
        _p(gw, _f32p), _p(gb, _f32p), B, Cin, H, W, Cout, kH, kW,
        stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
        groups, deformable_groups)
    if rc != 0:
        raise RuntimeError("oracle_deform_conv_backward failed: %d" % rc)
    out = dict(grad_x=gx, grad_offset=goff, grad_weight=gw)
    if mask is not None:
        out.update(grad_mask=gm, grad_bias=gb)
    return out


def decode(heat, offset, error, thr, root_thr=1.0, nms_rows=1):
    """One image.  heat [H,W] (or [1,1,H,W]), offset/error [2,H,W].
    nms_rows = 3: the north star's 3x3 NMS variant (the reference runs 1x3).
    Returns centers int64 [Nc,2] (x,y; x-major), kpt_xy int32 [N,2],
    align f32 [N,2], root f32 [N,2] (row-major keypoint order)."""
    heat = _f32(heat).reshape(heat.shape[-2], heat.shape[-1])
    H, W = heat.shape
    offset = _f32(offset).reshape(2, H, W)
    error = _f32(error).reshape(2, H, W)
    cap = H * W
    centers = np.zeros((cap, 2), np.int64)
    kxy = np.zeros((cap, 2), np.int32)
    align = np.zeros((cap, 2), np.float32)
    root = np.zeros((cap, 2), np.float32)
    nc, nk = ctypes.c_int(0), ctypes.c_int(0)
    lib().oracle_decode_opt(_p(heat, _f32p), _p(offset, _f32p), _p(error, _f32p),
                            H, W, float(thr), float(root_thr), int(nms_rows),
                            _p(centers, _i64p), cap, ctypes.byref(nc), _p(kxy, _i32p),
                            _p(align, _f32p), _p(root, _f32p), cap, ctypes.byref(nk))
    nc, nk = nc.value, nk.value
    return (centers[:nc].copy(), kxy[:nk].copy(), align[:nk].copy(),
            root[:nk].copy())


def group_fast(centers, root, thr, by_center_thr, nearest=False):
    """group_points_fast.  Returns (kpt_gid int32[N], group_mean int64[G,2],
    center_gid int32[Nc], n_groups).  nearest: the north star's nearest-centre variant
    (the reference takes the first centre within the threshold)."""
    centers = np.ascontiguousarray(centers, dtype=np.int64).reshape(-1, 2)
    root = _f32(root).reshape(-1, 2)
    nc, nk = len(centers), len(root)
    cg = np.full(max(nc, 1), -1, np.int32)
    kg = np.full(max(nk, 1), -1, np.int32)
    gm = np.zeros((max(nc, nk, 1), 2), np.int64)
    ng = lib().oracle_group_fast_opt(
        _p(centers, _i64p), nc, _p(root, _f32p), nk, float(thr),
        -1.0 if by_center_thr is None else float(by_center_thr), 1 if nearest else 0,
        _p(cg, _i32p), _p(gm, _i64p), _p(kg, _i32p))
    n_means = ng if by_center_thr is not None else 0
    return kg[:nk].copy(), gm[:n_means].copy(), cg[:nc].copy(), ng


def decode_group_batch(heat, offset, error, thr, root_thr, cluster_thr,
                       by_center_thr, nthreads=1):
    """Batch decode + grouping (CPU baseline leg). Returns #grouped keypoints."""
    heat, offset, error = _f32(heat), _f32(offset), _f32(error)
    B, _, H, W = heat.shape
    return lib().oracle_decode_group_batch(
        _p(heat, _f32p), _p(offset, _f32p), _p(error, _f32p), B, H, W,
        float(thr), float(root_thr), float(cluster_thr),
        -1.0 if by_center_thr is None else float(by_center_thr), H * W,
        nthreads)


# ---------------------------------------------------------------------------
# Reference-shaped Python structures from the oracle's arrays (used by tests to
# compare the oracle with the golden vectors written from the real reference).
# ---------------------------------------------------------------------------

def seeds_from_arrays(centers, align, root):
    """(root_center_arr, kpt_seeds) as ganet_head.py:219,234-238 builds them."""
    kpt_seeds = [(a, np.array(r, dtype=float)) for a, r in zip(align, root)]
    return np.asarray(centers, dtype=int).reshape(-1, 2), kpt_seeds


def lanes_from_groups(kpt_gid, n_groups, align, root, group_mean, downscale):
    """lane_post_process (post_process.py:216-242) from per-keypoint group ids."""
    lanes = []
    root64 = np.asarray(root, dtype=np.float64)
    for g in range(n_groups):
        idx = np.flatnonzero(kpt_gid == g)
        if len(idx) > 1:
            lanes.append(dict(
                id_class=g,
                points=[[align[i][0] * downscale, align[i][1] * downscale] for i in idx],
                centers=[[root64[i][0] * downscale, root64[i][1] * downscale] for i in idx]))
    cluster_centers = [dict(id_class=i, center=[c[0] * downscale, c[1] * downscale])
                       for i, c in enumerate(group_mean)]
    return lanes, cluster_centers
