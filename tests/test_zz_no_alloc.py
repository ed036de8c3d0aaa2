"""ganet_decode_batch_noalloc (the entry point that allocates nothing: its keypoint
bitmap sits in a static device array of the library) against ganet_decode_batch,
through the C-ABI, bit for bit, on two calls in a row - the second on a smaller batch
of smaller images, so that stale words of the first call's bitmap would show."""
import ctypes

import numpy as np
import pytest
import torch

from ganet_amd import _lib, decode, synth

pytestmark = pytest.mark.gpu


def _run(L, heat, offset, error, caps, noalloc):
    B, _, H, W = heat.shape
    p = decode._params(B, H, W, 0.05, 1.0, 4.0, 4.0, caps)
    bufs = decode.DecodeBuffers(B, caps[0], caps[1], caps[2], heat.device)
    bufs.dev.fill_(-7)
    stream = torch.cuda.current_stream(heat.device).cuda_stream
    if noalloc:
        rc = L.ganet_decode_batch_noalloc(heat.data_ptr(), offset.data_ptr(), error.data_ptr(),
                                          ctypes.byref(p), ctypes.byref(bufs.out), stream)
        _lib.check(rc, "ganet_decode_batch_noalloc")
    else:
        n = L.ganet_decode_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(n, dtype=torch.uint8, device=heat.device)
        rc = L.ganet_decode_batch(heat.data_ptr(), offset.data_ptr(), error.data_ptr(),
                                  ctypes.byref(p), ctypes.byref(bufs.out), ws.data_ptr(), n, stream)
        _lib.check(rc, "ganet_decode_batch")
    bufs.pack(L, p, stream)
    return bufs.fetch()


def _inputs(B, H, W, seed):
    h, o, e = synth.decode_inputs(B, H, W, lanes=16, seed=seed)
    dev = torch.device("cuda", 0)
    return tuple(torch.as_tensor(np.asarray(a)).to(dev) for a in (h, o, e))


def test_noalloc_matches_workspace_entry_point_twice():
    L = _lib.lib()
    # images above 8192 pixels take the bitmap in device memory (the scratch under test)
    for B, H, W, seed in ((6, 160, 400, 11), (3, 96, 120, 29)):
        heat, offset, error = _inputs(B, H, W, seed)
        caps = decode._default_caps(H, W, True, H * W // 2, 2048, 2048)
        hdr_a, rec_a = _run(L, heat, offset, error, caps, noalloc=False)
        hdr_b, rec_b = _run(L, heat, offset, error, caps, noalloc=True)
        assert int(hdr_a[:, 0].min()) > 0
        assert not (hdr_a[:, 3] & 7).any()
        assert np.array_equal(hdr_a, hdr_b)
        assert np.array_equal(rec_a, rec_b)


def test_noalloc_rejects_batches_beyond_the_static_array():
    L = _lib.lib()
    p = decode._params(64, 1280, 3200, 0.05, 1.0, 4.0, 4.0, (1024, 64, 64))   # 32 MiB of bitmap
    out = _lib.DecodeOut()
    for n, _ in _lib.DecodeOut._fields_:
        setattr(out, n, 256)
    fake = ctypes.c_void_p(256)
    rc = L.ganet_decode_batch_noalloc(fake, fake, fake, ctypes.byref(p), ctypes.byref(out), None)
    assert rc == _lib.GANET_ERR_WORKSPACE
