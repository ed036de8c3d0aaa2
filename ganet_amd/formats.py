"""Host-side mirror of the steps between PostProcessor and the result files
(SURVEY.md 8 f2): the per-image Python the reference runs after grouping.

    adjust_result(lanes, centers, crop_bbox, img_shape, points_thr)
        tools/ganet/tusimple/test_dataset.py:49-84 == tools/ganet/culane/test_dataset.py:79-113
        heat-map-scaled points -> original image pixels: `int(v * ratio + offset)`
        (truncation towards zero), lanes with `len(pts) > points_thr` kept.
    get_line_intersection / tusimple_convert_formal     tools/ganet/common.py:118-147
        x of every lane at the benchmark's h_samples rows, -2 where a lane has none.
    tusimple_json_line                                   tusimple/test_dataset.py:203-213
    out_result / culane_lines_text                       culane/test_dataset.py:138-148
        `%.2f %.2f ...` one lane per line (CULane .lines.txt).

Same names, argument meaning, return structures and in-place side effects as the
reference (adjust_result overwrites the coordinates inside `lanes` / `centers`).
The arithmetic is written on the same numpy / Python scalar types the reference
handles, so numpy's promotion rules of the installed version apply identically
(np.float32 * float is float64 before NEP 50 and float32 after it).

`image_lanes` is the array form of one image: decoded image in, final integer lanes out,
equal to PostProcessor -> adjust_result by construction and by test.

`format_batch` is the production path: a whole decoded batch (the record stream of
ganet_decode_pack) through the library's host routines ganet_lanes_from_packed /
ganet_tusimple_from_lanes / ganet_culane_text - no per-point interpreter work; the Python
structures of the reference are only built for the images a caller asks for.
"""
import ctypes
import json
import os

import numpy as np

from . import _lib


class _CropMap(object):
    """Heat-map-scaled (x, y) -> pixel of the original frame: v * ratio + offset per axis, then
    Python's int() (truncation towards zero, errors on NaN / infinity kept)."""

    def __init__(self, crop_bbox, img_shape):
        x0, y0, x1, y1 = crop_bbox[:4]
        height, width = img_shape[:2]
        self.scale = ((x1 - x0) / width, (y1 - y0) / height)
        self.shift = (x0, y0)

    def move(self, xy):
        """Rewrites the mutable pair `xy` in place, as the reference does, and returns it as a
        tuple."""
        for axis in (0, 1):
            xy[axis] = int(xy[axis] * self.scale[axis] + self.shift[axis])
        return tuple(xy)


def adjust_result(lanes, centers, crop_bbox, img_shape, points_thr):
    """tools/ganet/culane/test_dataset.py:79-113 (== tusimple/test_dataset.py:49-83): same
    arguments, same three result lists, same in-place rewrite of the coordinates inside `lanes`
    and `centers`; lanes of more than `points_thr` points are kept, with their start points."""
    to_frame = _CropMap(crop_bbox, img_shape)
    kept_points, kept_starts = [], []
    for lane in (lanes if lanes is not None else ()):
        moved = [to_frame.move(p) for p in lane['points']]
        starts = [to_frame.move(c) for c in lane['centers']]
        if len(moved) > points_thr:
            kept_points.append(moved)
            kept_starts.append(starts)
    cluster = [to_frame.move(entry['center']) for entry in (centers if centers is not None else ())]
    return kept_points, kept_starts, cluster


def _scalar_dtype(sample, ratio):
    """dtype numpy gives `sample_scalar * python_float` in this installation."""
    return np.asarray(sample * ratio).dtype


def image_lanes(img, downscale, crop_bbox, img_shape, points_thr):
    """PostProcessor.lane_post_process + adjust_result on the arrays of one decoded and
    grouped image (DecodeResult.image(b)); returns (results, virtual_centers,
    cluster_centers) as adjust_result does."""
    h_img, w_img = img_shape[:2]
    ratio = ((crop_bbox[2] - crop_bbox[0]) / w_img, (crop_bbox[3] - crop_bbox[1]) / h_img)
    offset = (crop_bbox[0], crop_bbox[1])
    gid = np.asarray(img["kpt_gid"])
    align = np.asarray(img["align"], dtype=np.float32) * downscale        # float32
    root = np.asarray(img["root"], dtype=np.float32).astype(np.float64) * downscale
    results, virtual_centers = [], []

    def to_int(v, axis_ratio, axis_offset):
        # int(scalar * ratio + offset) for a whole column: product and sum in the dtype
        # numpy gives the SCALAR expression in this installation, then truncation
        # towards zero; int() errors on NaN / inf are kept
        dt = _scalar_dtype(v.dtype.type(1), axis_ratio)
        t = v.astype(dt) * dt.type(axis_ratio) + dt.type(axis_offset)
        if np.isnan(t).any():
            raise ValueError("cannot convert float NaN to integer")
        if np.isinf(t).any():
            raise OverflowError("cannot convert float infinity to integer")
        return np.trunc(t).astype(np.int64)

    n_groups = int(img["n_groups"])
    if len(gid):
        order = np.argsort(gid, kind="stable")
        sg = gid[order]
        starts = np.searchsorted(sg, np.arange(n_groups), side="left")
        ends = np.searchsorted(sg, np.arange(n_groups), side="right")
        px = to_int(align[:, 0], ratio[0], offset[0])
        py = to_int(align[:, 1], ratio[1], offset[1])
        cx = to_int(root[:, 0], ratio[0], offset[0])
        cy = to_int(root[:, 1], ratio[1], offset[1])
        for g in range(n_groups):
            idx = order[starts[g]:ends[g]]
            # lane_post_process keeps groups of more than one point, adjust_result
            # those with more than points_thr
            if len(idx) > 1 and len(idx) > points_thr:
                results.append([(int(px[i]), int(py[i])) for i in idx])
                virtual_centers.append([(int(cx[i]), int(cy[i])) for i in idx])
    gc = np.asarray(img["group_center"], dtype=np.int64).reshape(-1, 2) * downscale
    cluster_centers = []
    if len(gc):
        gx = to_int(gc[:, 0], ratio[0], offset[0])
        gy = to_int(gc[:, 1], ratio[1], offset[1])
        cluster_centers = [(int(a), int(b)) for a, b in zip(gx, gy)]
    return results, virtual_centers, cluster_centers


def get_line_intersection(y, line, im_width, reg_x=-2):
    """x of the first segment of `line` whose y-range holds `y` (end points included,
    horizontal segments skipped), rounded by `int(x + 0.49999)`; `reg_x` if none."""
    for i in range(len(line) - 1):
        (x0, y0), (x1, y1) = line[i], line[i + 1]
        lo, hi = min(y0, y1), max(y0, y1)
        if lo <= y <= hi and lo != hi:
            k = (x1 - x0) / (y1 - y0)
            return int(k * (y - y0) + x0 + 0.49999)
    return reg_x


def tusimple_convert_formal(lanes, h_samples, im_width, reg_x=-2):
    return [[get_line_intersection(h, lane, im_width, reg_x) for h in h_samples]
            for lane in lanes]


def tusimple_json_line(result, h_samples, raw_file, im_width, run_time=20):
    """One line of the TuSimple submission file (tusimple/test_dataset.py:203-213)."""
    sample = dict(lanes=tusimple_convert_formal(result, h_samples, im_width),
                  h_samples=h_samples, raw_file=raw_file, run_time=run_time)
    return json.dumps(sample)


def culane_lines_text(lanes):
    """Text of a CULane `.lines.txt` file: one lane per line, `x y` pairs with two
    decimals separated by single spaces."""
    out = []
    for lane in lanes:
        out.append(" ".join("{:.2f} {:.2f}".format(p[0], p[1]) for p in lane) + "\n")
    return "".join(out)


def out_result(lanes, dst=None):
    if dst is not None:
        with open(dst, 'w') as f:
            f.write(culane_lines_text(lanes))


def float32_scalars():
    """1 if this NumPy keeps `np.float32 * python_float` in float32 (NumPy >= 2), else 0."""
    return int(_scalar_dtype(np.float32(1), 0.5) == np.float32)


def packed_from_images(images, by_center=True):
    """The record stream of ganet_decode_pack (include/ganet_hip.h) built on the host from
    per-image arrays (dicts as DecodeResult.image() returns them: kpt_gid, n_groups, align,
    root, group_center; kpt_xy / centers / center_gid / group_count optional).  For callers
    that hold decoded images as arrays (tests, CPU pipelines)."""
    B = len(images)
    header = np.zeros((B, 8), np.int32)
    recs, off = [], 0
    for b, im in enumerate(images):
        gid = np.asarray(im["kpt_gid"], np.int32).reshape(-1)
        nk, ng = len(gid), int(im["n_groups"])
        cen = np.asarray(im.get("centers", np.zeros((0, 2))), np.int32).reshape(-1, 2)
        nc = len(cen)
        gc = np.asarray(im["group_center"], np.int32).reshape(-1, 2) if by_center \
            else np.zeros((0, 2), np.int32)
        assert len(gc) == (ng if by_center else 0)
        parts = [np.asarray(im.get("kpt_xy", np.zeros((nk, 2))), np.int32).reshape(-1),
                 np.ascontiguousarray(im["align"], np.float32).reshape(-1).view(np.int32),
                 np.ascontiguousarray(im["root"], np.float32).reshape(-1).view(np.int32),
                 gid, cen.reshape(-1),
                 np.asarray(im.get("center_gid", np.zeros(nc)), np.int32).reshape(-1),
                 gc.reshape(-1),
                 np.asarray(im.get("group_count", np.zeros(ng)), np.int32).reshape(-1)]
        rec = np.concatenate(parts).astype(np.int32)
        header[b, :6] = (nk, nc, ng, 0, off, len(rec))
        recs.append(rec)
        off += len(rec)
    header[0, 6] = off
    return np.concatenate([header.reshape(-1)] + recs).astype(np.int32)


class BatchLanes(object):
    """Final integer lanes of a batch (ganet_lanes_from_packed).  image(b) rebuilds the
    reference's Python structures for one image; tusimple() / culane_texts() give the wire
    formats for the whole batch."""

    def __init__(self, stream, B, threads):
        self.stream, self.B, self.threads = stream, B, threads
        self.header = stream[:4 * B].reshape(B, 4)

    def n_lanes(self):
        return self.header[:, 0]

    def image(self, b):
        nl, ncc, off, words = (int(v) for v in self.header[b])
        rec = self.stream[4 * self.B + off:4 * self.B + off + words]
        results, vcs, pos = [], [], 0
        for _ in range(nl):
            n = int(rec[pos + 1])
            pts = rec[pos + 2:pos + 2 + 2 * n].reshape(n, 2).tolist()
            cts = rec[pos + 2 + 2 * n:pos + 2 + 4 * n].reshape(n, 2).tolist()
            results.append([tuple(p) for p in pts])
            vcs.append([tuple(c) for c in cts])
            pos += 2 + 4 * n
        ccs = [tuple(c) for c in rec[pos:pos + 2 * ncc].reshape(ncc, 2).tolist()]
        return results, vcs, ccs

    def tusimple(self, h_samples, reg_x=-2):
        """tusimple_convert_formal for every image: (xs [total lanes, len(h_samples)] int32,
        lane_start [B+1])."""
        L = _lib.lib()
        hs = np.ascontiguousarray(h_samples, np.int32)
        start = np.zeros(self.B + 1, np.int32)
        xs = np.empty((int(self.header[:, 0].sum()), len(hs)), np.int32)
        rc = L.ganet_tusimple_from_lanes(self.stream.ctypes.data, self.B, hs.ctypes.data, len(hs),
                                         reg_x, self.threads, xs.ctypes.data, xs.size,
                                         start.ctypes.data)
        _lib.check(rc, "ganet_tusimple_from_lanes")
        return xs, start

    def tusimple_json_lines(self, h_samples, raw_files, run_time=20):
        """One submission line per image (tusimple/test_dataset.py:203-213)."""
        xs, start = self.tusimple(h_samples)
        hs = [int(h) for h in h_samples]
        return [json.dumps(dict(lanes=xs[start[b]:start[b + 1]].tolist(), h_samples=hs,
                                raw_file=raw_files[b], run_time=run_time))
                for b in range(self.B)]

    def culane_text_buffer(self, scratch=None):
        """The `.lines.txt` texts of the batch in one buffer: (bytes-like uint8 array,
        start [B], length [B]); image b's text is buf[start[b]:start[b] + length[b]].
        scratch: a dict the caller keeps between calls (one per thread); the text buffer in it
        is reused while it is large enough, which saves the size query and the allocation -
        the returned buffer is then only valid until the next call with the same dict."""
        L = _lib.lib()
        start = np.zeros(2 * self.B + 1, np.uint64)
        buf = scratch.get("text") if scratch is not None else None
        rc = _lib.GANET_ERR_WORKSPACE
        if buf is not None:
            rc = L.ganet_culane_text(self.stream.ctypes.data, self.B, self.threads,
                                     buf.ctypes.data, buf.size, start.ctypes.data)
        if rc == _lib.GANET_ERR_WORKSPACE:
            L.ganet_culane_text(self.stream.ctypes.data, self.B, self.threads, None, 0,
                                start.ctypes.data)
            need = max(int(start[self.B]), 1)
            buf = np.empty(need if scratch is None else need + need // 4, np.uint8)
            if scratch is not None:
                scratch["text"] = buf
            rc = L.ganet_culane_text(self.stream.ctypes.data, self.B, self.threads,
                                     buf.ctypes.data, buf.size, start.ctypes.data)
        _lib.check(rc, "ganet_culane_text")
        return buf, start[:self.B].astype(np.int64), start[self.B + 1:].astype(np.int64)

    def culane_texts(self):
        """The `.lines.txt` text of every image (culane/test_dataset.py:138-148)."""
        buf, start, length = self.culane_text_buffer()
        raw = buf.tobytes()
        return [raw[s:s + n].decode("ascii") for s, n in zip(start.tolist(), length.tolist())]


def format_batch(packed, B, by_center, downscale, crop_bbox, img_shape, points_thr, threads=0,
                 scratch=None):
    """lane_post_process + adjust_result for a whole batch.  `packed`: the int32 record stream
    of ganet_decode_pack on the host (DecodeResult.packed(), HotPath's pinned block, or
    packed_from_images).  Raises ValueError where the reference's int() would
    (NaN / infinite coordinates).  scratch: a dict kept by the caller between calls (one per
    thread): the lane buffer in it is reused while large enough (no size query, no
    allocation); the returned BatchLanes then lives until the next call with that dict."""
    L = _lib.lib()
    packed = np.ascontiguousarray(packed, np.int32)
    f = _lib.LaneFormat()
    f.B, f.by_center, f.downscale, f.points_thr = B, int(bool(by_center)), int(downscale), int(points_thr)
    f.float32_scalars = float32_scalars()
    f.threads = threads or min(16, len(os.sched_getaffinity(0)))
    f.img_h, f.img_w = int(img_shape[0]), int(img_shape[1])
    f.crop_x0, f.crop_y0, f.crop_x1, f.crop_y1 = (float(v) for v in crop_bbox)
    f.packed_words = packed.size
    need = ctypes.c_size_t(0)
    out = scratch.get("lanes") if scratch is not None else None
    rc = _lib.GANET_ERR_WORKSPACE
    if out is not None:
        rc = L.ganet_lanes_from_packed(packed.ctypes.data, ctypes.byref(f), out.ctypes.data,
                                       out.size, ctypes.byref(need))
    if rc == _lib.GANET_ERR_WORKSPACE:
        rc = L.ganet_lanes_from_packed(packed.ctypes.data, ctypes.byref(f), None, 0,
                                       ctypes.byref(need))
        if rc not in (_lib.GANET_OK, _lib.GANET_ERR_WORKSPACE):
            _lib.check(rc, "ganet_lanes_from_packed")
        out = np.empty(need.value if scratch is None else need.value + need.value // 4 + 64,
                       np.int32)
        if scratch is not None:
            scratch["lanes"] = out
        rc = L.ganet_lanes_from_packed(packed.ctypes.data, ctypes.byref(f), out.ctypes.data,
                                       out.size, ctypes.byref(need))
    if rc == _lib.GANET_ERR_NOT_FINITE:
        raise ValueError("cannot convert a non-finite coordinate to integer")
    _lib.check(rc, "ganet_lanes_from_packed")
    return BatchLanes(out[:need.value], B, f.threads)
