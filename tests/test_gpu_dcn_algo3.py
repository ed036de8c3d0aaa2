"""GPU parity of algorithm 3 (the contract-then-sample form as one persistent launch,
csrc/deform_conv_stream.hip).  It runs the arithmetic of algorithm 2 in the same order, so
the bar is bit-identity with algorithm 2 (itself checked against the oracle in
test_gpu_dcn.py), plus the oracle tolerance on one case; `check=True` also asserts that no
wait inside the launch ran out."""
import numpy as np
import pytest
import torch

from ganet_amd import dcn, synth
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def run(x, off, w, K, algo, dtype=torch.bfloat16, channels_last=True, mask=None, bias=None, **kw):
    xt = torch.from_numpy(x).cuda().to(dtype)
    if channels_last:
        xt = xt.contiguous(memory_format=torch.channels_last)
    out = dcn.deform_conv_forward(
        xt, torch.from_numpy(off).cuda(), torch.from_numpy(w).cuda().to(dtype),
        (1, 1), (0, (K - 1) // 2), (1, 1), 1, 1,
        mask=None if mask is None else torch.from_numpy(mask).cuda(),
        bias=None if bias is None else torch.from_numpy(bias).cuda().to(dtype),
        algo=algo, check=True, **kw)
    torch.cuda.synchronize()
    return out


# B, Cin, H, W, K: more than one image per queue (B > 8), fewer images than queues, ragged
# pixel counts (H*W not a multiple of the 32/64/128/256-pixel units), 1/2/3 kinds of workgroup
SHAPES = [
    (40, 256, 40, 100, 7),    # the LFA: 5 images per queue, 2 kinds (224 columns each)
    (3, 256, 13, 37, 7),      # 3 of 8 queues hold an image
    (9, 256, 20, 50, 9),      # K 9: 3 kinds of 192 columns, queue 0 holds two images
    (11, 128, 17, 23, 7),     # Cin 128: all 448 columns fit one kind of workgroup
    (8, 128, 9, 31, 5),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_algo3_bit_identical_to_algo2(shape):
    B, Cin, H, W, K = shape
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=3)
    a2 = run(x, pts, w, K, 2, offset_is_points=True, fuse_relu=True)
    a3 = run(x, pts, w, K, 3, offset_is_points=True, fuse_relu=True)
    assert torch.equal(a2, a3)
    # a second call on the same tensors: the counters are zeroed per launch
    assert torch.equal(a3, run(x, pts, w, K, 3, offset_is_points=True, fuse_relu=True))


def test_algo3_against_the_oracle_with_far_samples():
    """Offsets of several rows: sampling waves wait for bands far ahead of their pixels."""
    B, Cin, H, W, K = 10, 256, 24, 40, 7
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, Cin, 1, K)) / np.sqrt(Cin * K)).astype(np.float32)
    off = (rng.standard_normal((B, 2 * K, H, W)) * 9.0).astype(np.float32)
    xr = torch.from_numpy(x).bfloat16().float().numpy()
    wr = torch.from_numpy(w).bfloat16().float().numpy()
    ref = O.deform_conv_forward(xr, off, wr, (1, 1), (0, 3), (1, 1), nthreads=16)
    got = run(x, off, w, K, 3)
    assert torch.equal(got, run(x, off, w, K, 2))
    err = np.abs(got.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 1e-2     # bf16 tolerance of the north star


@pytest.mark.parametrize("dtype,channels_last", [(torch.float16, False), (torch.bfloat16, False),
                                                 (torch.float32, True)])
def test_algo3_mask_bias_layouts_and_dtypes(dtype, channels_last):
    """DCNv2 mask + bias + ReLU, NCHW tensors (x copied channel-last into the workspace, the
    output written NCHW), f16, and float32 tensors contracted in one bf16 pass."""
    B, Cin, H, W, K = 5, 256, 10, 25, 7
    rng = np.random.default_rng(11)
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=9)
    mask = rng.uniform(0, 1, (B, K, H, W)).astype(np.float32)
    bias = rng.standard_normal(64).astype(np.float32)
    kw = dict(mask=mask, bias=bias, fuse_relu=True, dtype=dtype, channels_last=channels_last)
    if dtype == torch.float32:
        kw["mfma_bf16"] = True
    assert torch.equal(run(x, off, w, K, 2, **kw), run(x, off, w, K, 3, **kw))


def test_algo3_refused_where_it_does_not_apply():
    x, off, w, pts = synth.lfa_inputs(2, 256, 8, 12, 7, seed=1)
    with pytest.raises(Exception):
        run(x, off, w, 7, 3, dtype=torch.float32)          # exact / split float32 arithmetic


def test_algo3_several_launches_per_call(monkeypatch):
    """A Y budget of 1 MB: the batch is walked in launches of 8 images (one per queue), the
    last one ragged; the control block is cleared between them, the status word only once."""
    monkeypatch.setenv("GANET_STREAM_Y_MB", "1")
    B, Cin, H, W, K = 20, 128, 17, 23, 7
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=4)
    a3 = run(x, pts, w, K, 3, offset_is_points=True, fuse_relu=True)
    monkeypatch.delenv("GANET_STREAM_Y_MB")
    assert torch.equal(a3, run(x, pts, w, K, 2, offset_is_points=True, fuse_relu=True))
