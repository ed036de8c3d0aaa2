"""ganet_decode_batch_packed (decode + grouping + the record stream in one call; a single
launch where one workgroup per image does the whole decode) against ganet_decode_batch followed
by ganet_decode_pack: the output arrays, every header and every record bit for bit.  The two
differ only in where a record sits in the stream (readers go by the header's offsets)."""
import ctypes

import numpy as np
import pytest
import torch

from ganet_amd import _lib, synth
from ganet_amd import decode as D

pytestmark = pytest.mark.gpu


def _inputs(B, H, W, lanes, seed):
    return [torch.from_numpy(a).cuda() for a in synth.decode_inputs(B, H, W, lanes=lanes, seed=seed)]


def _two_step(L, p, tensors, bufs):
    heat, off, err = tensors
    n = L.ganet_decode_workspace_bytes(ctypes.byref(p))
    ws = torch.empty(max(n, 256), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ganet_decode_batch(heat.data_ptr(), off.data_ptr(), err.data_ptr(), ctypes.byref(p),
                                    ctypes.byref(bufs.out), ws.data_ptr(), n, st), "decode")
    _lib.check(L.ganet_decode_pack(ctypes.byref(p), ctypes.byref(bufs.out), bufs.packed.data_ptr(),
                                   bufs.packed_words, st), "pack")
    torch.cuda.synchronize()


def _one_call(L, p, tensors, bufs, ws, packed=None, packed_words=None):
    heat, off, err = tensors
    st = torch.cuda.current_stream().cuda_stream
    packed = bufs.packed if packed is None else packed
    rc = L.ganet_decode_batch_packed(heat.data_ptr(), off.data_ptr(), err.data_ptr(), ctypes.byref(p),
                                     ctypes.byref(bufs.out), packed.data_ptr(),
                                     packed.numel() if packed_words is None else packed_words,
                                     ws.data_ptr(), ws.numel(), st)
    _lib.check(rc, "decode_packed")
    torch.cuda.synchronize()


def _records(packed, B):
    flat = packed.cpu().numpy()
    hdr = flat[:8 * B].reshape(B, 8).copy()
    recs = [flat[8 * B + hdr[b, 4]:8 * B + hdr[b, 4] + hdr[b, 5]].copy() for b in range(B)]
    return hdr, recs


def _compare(B, a, b, image_order):
    ha, ra = _records(a.packed, B)
    hb, rb = _records(b.packed, B)
    assert np.array_equal(ha[:, :4], hb[:, :4]) and np.array_equal(ha[:, 5], hb[:, 5])
    assert int(ha[0, 6]) == int(hb[0, 6]) == int(ha[:, 5].sum())
    assert np.array_equal(ha[1:, 6:], hb[1:, 6:]) and ha[0, 7] == hb[0, 7] == 0
    for x, y in zip(ra, rb):
        assert x.tobytes() == y.tobytes()
    # the records tile the stream without gaps or overlap, in whatever order
    order = np.argsort(hb[:, 4], kind="stable")
    assert np.array_equal(np.cumsum(hb[order, 5]) - hb[order, 5], hb[order, 4])
    if image_order:
        assert np.array_equal(ha[:, 4], hb[:, 4])
    assert torch.equal(a.dev, b.dev)


CASES = [  # key, B, lanes, by-centre, flags
    ("s8_tusimple", 1, 4, True, 0),
    ("s8_culane", 128, 4, True, 0),
    ("s8_culane", 37, 6, False, 0),
    ("s8_culane", 16, 4, True, D.FLAG_NO_GROUP),
    ("s8_culane", 16, 4, True, D.FLAG_NEAREST | D.FLAG_NMS_3X3),
    ("s4_culane", 9, 4, True, 0),            # 32000 pixels: grid-wide kernels + the pack kernel
    ("s4_culane", 5, 4, False, 0),           # no-centre mode at that size: still one workgroup per image
]


@pytest.mark.parametrize("key,B,lanes,by_center,flags", CASES)
def test_one_call_matches_decode_then_pack(key, B, lanes, by_center, flags):
    L = _lib.lib()
    cfg = synth.SHAPES[key]
    H, W = cfg["H"], cfg["W"]
    T = float(cfg["T"])
    caps = (min(H * W, 2048), 256, 256 if by_center else min(H * W, 2048))
    p = D._params(B, H, W, cfg["thr"], cfg["root_thr"], T, T if by_center else None, caps, flags=flags)
    n = L.ganet_decode_packed_workspace_bytes(ctypes.byref(p))
    assert n == L.ganet_decode_workspace_bytes(ctypes.byref(p)) + 256
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    one_launch = H * W <= 8192 or not by_center and not (flags & D.FLAG_NO_GROUP)
    # two calls in a row on ONE workspace (the counters must be back at zero), the second on
    # other images
    for seed in (0, 1000):
        tensors = _inputs(B, H, W, lanes, seed)
        a = D.DecodeBuffers(B, *caps, "cuda")
        b = D.DecodeBuffers(B, *caps, "cuda")
        for bufs in (a, b):
            bufs.dev.zero_()
            bufs.packed.fill_(-7)
        _two_step(L, p, tensors, a)
        _one_call(L, p, tensors, b, ws)
        _compare(B, a, b, image_order=not one_launch)
        assert int(ws[-256:].view(torch.int64)[:2].abs().sum()) == 0


def test_records_that_do_not_fit_are_flagged_not_written():
    """A stream too short for every record: the records that got their words inside it are
    whole, the others carry bit 3 and nothing is written past the end."""
    L = _lib.lib()
    cfg = synth.SHAPES["s8_culane"]
    B, H, W, T = 24, cfg["H"], cfg["W"], float(cfg["T"])
    caps = (1024, 256, 256)
    p = D._params(B, H, W, cfg["thr"], cfg["root_thr"], T, T, caps)
    tensors = _inputs(B, H, W, 4, 5)
    a = D.DecodeBuffers(B, *caps, "cuda")
    b = D.DecodeBuffers(B, *caps, "cuda")
    _two_step(L, p, tensors, a)
    ha, ra = _records(a.packed, B)
    need = int(ha[0, 6])
    short = 8 * B + need // 2
    ws = torch.zeros(L.ganet_decode_packed_workspace_bytes(ctypes.byref(p)), dtype=torch.uint8, device="cuda")
    b.packed.fill_(-7)
    _one_call(L, p, tensors, b, ws, packed_words=short)
    flat = b.packed.cpu().numpy()
    assert (flat[short:] == -7).all()
    hb = flat[:8 * B].reshape(B, 8)
    fitted = (hb[:, 3] & 8) == 0
    assert 0 < fitted.sum() < B
    assert int(hb[0, 6]) == need
    for i in range(B):
        assert hb[i, 5] == ha[i, 5]
        if fitted[i]:
            assert 8 * B + hb[i, 4] + hb[i, 5] <= short
            assert flat[8 * B + hb[i, 4]:8 * B + hb[i, 4] + hb[i, 5]].tobytes() == ra[i].tobytes()
        else:
            assert 8 * B + hb[i, 4] + hb[i, 5] > short
    # the next call on the same workspace is unaffected
    b.packed.fill_(-7)
    b.dev.zero_()
    a.dev.zero_()
    _two_step(L, p, tensors, a)
    _one_call(L, p, tensors, b, ws)
    _compare(B, a, b, image_order=False)


def test_argument_checks():
    L = _lib.lib()
    p = D._params(2, 40, 100, 0.3, 1.0, 4.0, 4.0, (64, 16, 16))
    bufs = D.DecodeBuffers(2, 64, 16, 16, "cuda")
    t = torch.zeros(2 * 2 * 4000, device="cuda")
    ws = torch.zeros(L.ganet_decode_packed_workspace_bytes(ctypes.byref(p)), dtype=torch.uint8, device="cuda")
    args = (t.data_ptr(), t.data_ptr(), t.data_ptr(), ctypes.byref(p), ctypes.byref(bufs.out))
    assert L.ganet_decode_batch_packed(*args, None, 100, ws.data_ptr(), ws.numel(), None) == _lib.GANET_ERR_NULL
    assert L.ganet_decode_batch_packed(*args, bufs.packed.data_ptr(), 15, ws.data_ptr(), ws.numel(),
                                       None) == _lib.GANET_ERR_WORKSPACE
    assert L.ganet_decode_batch_packed(*args, bufs.packed.data_ptr(), bufs.packed_words, ws.data_ptr(),
                                       ws.numel() - 256, None) == _lib.GANET_ERR_WORKSPACE
    torch.cuda.synchronize()
