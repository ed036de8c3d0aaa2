"""Host-side mirror of the reference's keypoint decode on top of the gfx950 C-ABI.

  decode_batch(...)        whole batch in one launch sequence + one D2H copy
  ktdet_decode_fast(...)   same signature / return structures as
                           GANetHeadFast.ktdet_decode_fast
                           (mmdet/models/dense_heads/ganet_head.py:203-238)

The device does NMS, thresholding, ordered compaction, start-point regression
and (optionally) the lane grouping of tools/ganet/post_process.py:141-194; this
file only moves pointers and rebuilds the reference's Python structures
(int64 centre array, list of (float32[2], float64[2]) seeds).  No CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_FIELDS = (  # name, per-image int32 words as a function of the capacities
    ("counts", lambda k, c, g: _lib.DECODE_COUNT_STRIDE),
    ("centers_xy", lambda k, c, g: 2 * c),
    ("center_gid", lambda k, c, g: c),
    ("kpt_xy", lambda k, c, g: 2 * k),
    ("kpt_align", lambda k, c, g: 2 * k),
    ("kpt_root", lambda k, c, g: 2 * k),
    ("kpt_gid", lambda k, c, g: k),
    ("group_center", lambda k, c, g: 2 * g),
    ("group_count", lambda k, c, g: g),
)
_FLOAT_FIELDS = ("kpt_align", "kpt_root")


class DecodeBuffers(object):
    """One int32 device allocation carved into the arrays of ganet_decode_out."""

    def __init__(self, B, max_kpts, max_centers, max_groups, device):
        self.B, self.max_kpts = B, max_kpts
        self.max_centers, self.max_groups = max_centers, max_groups
        self.offsets, total = {}, 0
        for name, words in _FIELDS:
            n = B * words(max_kpts, max_centers, max_groups)
            self.offsets[name] = (total, n)
            total += (n + 63) // 64 * 64
        self.total = total
        self.dev = torch.empty(total, dtype=torch.int32, device=device)
        self.out = _lib.DecodeOut()
        base = self.dev.data_ptr()
        for name, _ in _FIELDS:
            setattr(self.out, name, base + 4 * self.offsets[name][0])
        # dense record stream written by ganet_decode_pack (what the host copies)
        self.header_words = B * _lib.DECODE_COUNT_STRIDE
        self.packed_words = self.header_words + B * (
            7 * max_kpts + 3 * max_centers + 3 * max_groups)
        self.packed = torch.empty(self.packed_words, dtype=torch.int32, device=device)

    def pack(self, L, p, stream):
        rc = L.ganet_decode_pack(ctypes.byref(p), ctypes.byref(self.out),
                                 self.packed.data_ptr(), self.packed_words, stream)
        _lib.check(rc, "ganet_decode_pack")

    def fetch(self):
        """Synchronous two-step D2H: the header, then exactly the records found."""
        hdr = self.packed[:self.header_words].cpu().numpy()
        total = int(hdr[6])
        rec = self.packed[self.header_words:self.header_words + total].cpu().numpy()
        return hdr.reshape(self.B, _lib.DECODE_COUNT_STRIDE), rec

    def host_views(self, flat):
        """numpy views of a host copy `flat` (int32 [total])."""
        v = {}
        for name, words in _FIELDS:
            off, n = self.offsets[name]
            a = flat[off:off + n]
            if name in _FLOAT_FIELDS:
                a = a.view(np.float32)
            v[name] = a.reshape(self.B, -1)
        v["centers_xy"] = v["centers_xy"].reshape(self.B, self.max_centers, 2)
        v["kpt_xy"] = v["kpt_xy"].reshape(self.B, self.max_kpts, 2)
        v["kpt_align"] = v["kpt_align"].reshape(self.B, self.max_kpts, 2)
        v["kpt_root"] = v["kpt_root"].reshape(self.B, self.max_kpts, 2)
        v["group_center"] = v["group_center"].reshape(self.B, self.max_groups, 2)
        return v


def records_in_image_order(header, records):
    """(header, records) with the records one after the other in image order, as
    ganet_decode_pack lays them out.  ganet_decode_batch_packed's one-launch form places a record
    where its image got to first; the words of every record are the same either way, so this
    only moves them (and rewrites header[:, 4]).  A stream with records that did not fit
    (overflow bit 3) is returned as it is."""
    off, words = header[:, 4].astype(np.int64), header[:, 5].astype(np.int64)
    if (header[:, 3] & 8).any():
        return header, records
    dense = np.cumsum(words) - words
    if np.array_equal(dense, off):
        return header, records
    idx = np.repeat(off - dense, words) + np.arange(int(words.sum()), dtype=np.int64)
    header = header.copy()
    header[:, 4] = dense
    return header, records[idx]


class DecodeResult(object):
    """Host copy of a decoded batch: the header [B,8] and the dense record
    stream of ganet_decode_pack.  image(b) slices one image's arrays out."""

    def __init__(self, header, records, params, grouped, buffers=None):
        self.header = header
        self.records = records
        self.params = params          # dict(thr, root_thr, cluster_thr, by_center_thr)
        self.grouped = grouped
        self.buffers = buffers        # device buffers (kept for a later grouping pass)
        self.B = header.shape[0]

    @property
    def counts(self):
        return self.header[:, :4]

    def n_kpts(self, b=0):
        return int(self.header[b, 0])

    def packed(self):
        """header + records as one int32 stream (the layout of ganet_decode_pack), the input
        of formats.format_batch."""
        return np.concatenate([np.asarray(self.header, np.int32).reshape(-1),
                               np.asarray(self.records, np.int32).reshape(-1)])

    def image(self, b=0):
        h = self.header[b]
        if h[3] & 7:
            raise RuntimeError("decode capacity overflow in image %d (bits %d)" % (b, h[3]))
        by_center = self.params["by_center_thr"] is not None
        off, words = int(h[4]), int(h[5])
        r = self.records[off:off + words]
        nk, nc, ng = int(h[0]), int(h[1]), int(h[2])
        ngc = ng if by_center else 0
        assert words == 7 * nk + 3 * nc + 2 * ngc + ng, "corrupt decode record"
        pos = [0]

        def take(n, shape=None, f32=False):
            a = r[pos[0]:pos[0] + n]
            pos[0] += n
            if f32:
                a = a.view(np.float32)
            return a.reshape(shape) if shape else a

        d = dict(kpt_xy=take(2 * nk, (nk, 2)), align=take(2 * nk, (nk, 2), True),
                 root=take(2 * nk, (nk, 2), True))
        kgid = take(nk)
        d["centers"] = take(2 * nc, (nc, 2))
        cgid = take(nc)
        gcen = take(2 * ngc, (ngc, 2))
        gcnt = take(ng)
        if self.grouped:
            d.update(kpt_gid=kgid, n_groups=ng, group_center=gcen, group_count=gcnt,
                     center_gid=cgid if by_center else cgid[:0])
        return d


def _default_caps(H, W, by_center, max_kpts, max_centers, max_groups):
    HW = H * W
    if max_kpts is None:
        max_kpts = min(HW, max(1024, HW // 8))
    if max_centers is None:
        max_centers = min(HW, 512)
    max_centers = min(max_centers, _lib.DECODE_MAX_CENTERS)
    if max_groups is None:
        max_groups = max_centers if by_center else max_kpts
    return int(max_kpts), int(max_centers), int(max_groups)


def _params(B, H, W, thr, root_thr, cluster_thr, by_center_thr, caps, flags=0):
    p = _lib.DecodeParams()
    p.B, p.H, p.W = B, H, W
    p.kpt_thr, p.root_thr = float(thr), float(root_thr)
    p.cluster_thr = float(cluster_thr)
    p.by_center_thr = -1.0 if by_center_thr is None else float(by_center_thr)
    p.max_kpts, p.max_centers, p.max_groups = caps
    p.flags = flags
    return p


FLAG_NO_GROUP, FLAG_NMS_3X3, FLAG_NEAREST, FLAG_LOGITS = 1, 2, 4, 8


def decode_batch(heat, offset, error, thr=0.1, root_thr=1, cluster_thr=4,
                 cluster_by_center_thr=None, group=True, max_kpts=None,
                 max_centers=None, max_groups=None, keep_device=False,
                 nms_3x3=False, assign_nearest=False, logits=False):
    """Decode (and group) a batch on the GPU.

    logits: `heat` is the head's raw keypoint map; clamp(sigmoid(.), 1e-4, 1 - 1e-4)
    (ganet_head.py:291) is applied on the device as the plane is read.

    nms_3x3 / assign_nearest: non-default variants with no counterpart in the
    reference's code (3x3 heat-map NMS instead of its 1x3, nearest centre within the
    threshold instead of the first one) - the forms BASELINE.json's north star names.

    heat [B,1,H,W], offset [B,2,H,W], error [B,2,H,W]: float32 CUDA tensors.
    Returns a DecodeResult (host arrays after one D2H copy).  Capacities grow
    automatically when an image overflows them; more than 2048 centres in one
    image is a hard error (the reference has no such limit).
    """
    if not (heat.is_cuda and offset.is_cuda and error.is_cuda):
        raise NotImplementedError("ganet_amd.decode: CUDA tensors required")
    L = _lib.lib()
    heat = heat.contiguous().float()
    offset = offset.contiguous().float()
    error = error.contiguous().float()
    B, _, H, W = heat.shape
    assert offset.shape == (B, 2, H, W) and error.shape == (B, 2, H, W)
    by_center = cluster_by_center_thr is not None
    caps = _default_caps(H, W, by_center, max_kpts, max_centers, max_groups)
    while True:
        p = _params(B, H, W, thr, root_thr, cluster_thr, cluster_by_center_thr,
                    caps, flags=(0 if group else FLAG_NO_GROUP) |
                    (FLAG_NMS_3X3 if nms_3x3 else 0) |
                    (FLAG_NEAREST if assign_nearest else 0) |
                    (FLAG_LOGITS if logits else 0))
        with torch.cuda.device(heat.device):
            bufs = DecodeBuffers(B, caps[0], caps[1], caps[2], heat.device)
            # decode, grouping and the record stream in one call (one launch where a
            # workgroup per image does it all); its counters start at zero
            ws_bytes = L.ganet_decode_packed_workspace_bytes(ctypes.byref(p))
            ws = torch.zeros(max(ws_bytes, 256), dtype=torch.uint8, device=heat.device)
            stream = torch.cuda.current_stream(heat.device).cuda_stream
            rc = L.ganet_decode_batch_packed(heat.data_ptr(), offset.data_ptr(),
                                             error.data_ptr(), ctypes.byref(p),
                                             ctypes.byref(bufs.out), bufs.packed.data_ptr(),
                                             bufs.packed_words, ws.data_ptr(), ws_bytes, stream)
            _lib.check(rc, "ganet_decode_batch_packed")
            header, records = records_in_image_order(*bufs.fetch())
        counts = header[:, :4]
        over = int(np.bitwise_or.reduce(counts[:, 3])) & 7
        if over == 0:
            break
        nk, nc, ng = (int(counts[:, i].max()) for i in range(3))
        if (over & 2) and nc > _lib.DECODE_MAX_CENTERS:
            raise RuntimeError(
                "ganet_amd.decode: %d centre points in one image exceed the "
                "device limit of %d" % (nc, _lib.DECODE_MAX_CENTERS))
        caps = (max(caps[0], nk), max(caps[1], min(nc, _lib.DECODE_MAX_CENTERS)),
                max(caps[2], ng, (nc if by_center else nk)))
    params = dict(thr=thr, root_thr=root_thr, cluster_thr=cluster_thr,
                  by_center_thr=cluster_by_center_thr)
    return DecodeResult(header, records, params, grouped=group,
                        buffers=bufs if keep_device else None)


class KptSeeds(list):
    """The reference's kpt_seeds list, plus the device-side result it came from
    so that PostProcessor can group on the GPU without re-uploading."""
    ganet_result = None


def seeds_from_image(img):
    """(root_center_arr, kpt_seeds) exactly as ganet_head.py:219,234-238 builds
    them: int64 [Nc,2] and a list of (float32[2], float64[2]) tuples."""
    centers = np.ascontiguousarray(img["centers"]).astype(np.int64).reshape(-1, 2)
    align = np.array(img["align"], dtype=np.float32)
    root = np.array(img["root"], dtype=np.float32)
    seeds = KptSeeds()
    for a, r in zip(align, root):
        seeds.append((a, np.array(r, dtype=float)))
    return centers, seeds


def ktdet_decode_fast(heat, offset, error, thr=0.1, root_thr=1):
    """Drop-in for GANetHeadFast.ktdet_decode_fast (ganet_head.py:203-238).
    Batch must be 1, as in the reference (`squeeze(0)`, `[0, 0]`)."""
    if heat.shape[0] != 1:
        raise ValueError("ktdet_decode_fast handles one image; use decode_batch")
    res = decode_batch(heat, offset, error, thr=thr, root_thr=root_thr,
                       group=False, keep_device=True)
    centers, seeds = seeds_from_image(res.image(0))
    seeds.ganet_result = res
    return centers, seeds


def group_on_device(res, cluster_thr, cluster_by_center_thr):
    """Run the grouping kernel on a decoded batch still resident on the GPU."""
    L = _lib.lib()
    bufs = res.buffers
    if bufs is None:
        raise RuntimeError("decode result no longer holds its device buffers")
    by_center = cluster_by_center_thr is not None
    counts = res.counts
    need_groups = int(counts[:, 1].max() if by_center else counts[:, 0].max())
    if need_groups > bufs.max_groups:
        raise RuntimeError("group capacity too small")  # caller re-decodes
    p = _params(bufs.B, 1, 1, 0.0, 0.0, cluster_thr, cluster_by_center_thr,
                (bufs.max_kpts, bufs.max_centers, bufs.max_groups))
    with torch.cuda.device(bufs.dev.device):
        stream = torch.cuda.current_stream(bufs.dev.device).cuda_stream
        rc = L.ganet_group_batch(ctypes.byref(p), ctypes.byref(bufs.out), stream)
        _lib.check(rc, "ganet_group_batch")
        bufs.pack(L, p, stream)
        header, records = bufs.fetch()
    params = dict(res.params, cluster_thr=cluster_thr,
                  by_center_thr=cluster_by_center_thr)
    return DecodeResult(header, records, params, grouped=True, buffers=bufs)


def group_seeds_on_device(cpt_seeds, kpt_seeds, cluster_thr, cluster_by_center_thr,
                          device=None):
    """Grouping for seeds that live on the host (lists/arrays in the reference's
    format): upload, group on the GPU, download."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    centers = np.asarray(cpt_seeds, dtype=np.int64).reshape(-1, 2)
    n = len(kpt_seeds)
    align = np.zeros((n, 2), np.float32)
    root64 = np.zeros((n, 2), np.float64)
    for i, (a, r) in enumerate(kpt_seeds):
        align[i], root64[i] = a, r
    root = root64.astype(np.float32)
    if not np.array_equal(root.astype(np.float64), root64):
        raise ValueError("ganet_amd: seed roots must be float32-representable "
                         "(they are in the reference's decode)")
    nc = len(centers)
    if nc > _lib.DECODE_MAX_CENTERS:
        raise RuntimeError("too many centre points for the device (%d)" % nc)
    by_center = cluster_by_center_thr is not None
    caps = (max(n, 1), max(nc, 1), max(nc if by_center else n, 1))
    bufs = DecodeBuffers(1, caps[0], caps[1], caps[2], device)
    flat = np.zeros(bufs.total, np.int32)
    v = bufs.host_views(flat)
    v["counts"][0, 0], v["counts"][0, 1] = n, nc
    v["centers_xy"][0, :nc] = centers
    v["kpt_root"][0, :n] = root
    v["kpt_align"][0, :n] = align
    bufs.dev.copy_(torch.from_numpy(flat))
    res = DecodeResult(v["counts"].copy(), np.zeros(0, np.int32),
                       dict(thr=None, root_thr=None, cluster_thr=cluster_thr,
                            by_center_thr=cluster_by_center_thr),
                       grouped=False, buffers=bufs)
    return group_on_device(res, cluster_thr, cluster_by_center_thr)
