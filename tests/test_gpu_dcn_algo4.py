"""GPU parity of algorithm 4 (csrc/deform_conv_ring.hip: the contract-then-sample form as one
launch with the per-tap products kept in the forming XCD's L2, tap-major, static schedule).  It
forms the same products in the same order as algorithm 2 and blends them in the same order, so the
bar is bit-identity with algorithm 2 (itself checked against the oracle in test_gpu_dcn.py), plus
the oracle tolerance; `check=True` also asserts that no wait inside the launch ran out and that
every image was taken by some team."""
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


# B, Cin, H, W, K: fewer images than teams (8), an odd number of images per team (a half of the
# last pair idles), several pairs per team, ragged pixel counts (H*W not a multiple of 16), the
# largest map a team takes (64x64 = 4096 pixels), Cin 128 and 64-wide K chunks, K 9 and K 5
SHAPES = [
    (40, 256, 40, 100, 7),    # the LFA: 5 images per team = 3 pairs, the last one half empty
    (3, 256, 13, 37, 7),      # 3 of 8 teams hold an image
    (17, 256, 20, 50, 9),     # K 9; team 0 holds three images
    (11, 128, 17, 23, 7),     # Cin 128
    (2, 256, 64, 64, 7),      # 4096 pixels: every wave of a half has work
    (33, 128, 9, 31, 5),      # K 5
]


@pytest.mark.parametrize("shape", SHAPES)
def test_algo4_bit_identical_to_algo2(shape):
    B, Cin, H, W, K = shape
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=3)
    a2 = run(x, pts, w, K, 2, offset_is_points=True, fuse_relu=True)
    a4 = run(x, pts, w, K, 4, offset_is_points=True, fuse_relu=True)
    assert torch.equal(a2, a4)
    # a second call on the same tensors: the control block is zeroed per launch, the ring's old
    # contents are never read
    assert torch.equal(a4, run(x, pts, w, K, 4, offset_is_points=True, fuse_relu=True))


def test_algo4_against_the_oracle_with_far_samples():
    """Offsets of several rows and samples outside the map: corners anywhere in the plane."""
    B, Cin, H, W, K = 10, 256, 24, 40, 7
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, Cin, 1, K)) / np.sqrt(Cin * K)).astype(np.float32)
    off = (rng.standard_normal((B, 2 * K, H, W)) * 9.0).astype(np.float32)
    xr = torch.from_numpy(x).bfloat16().float().numpy()
    wr = torch.from_numpy(w).bfloat16().float().numpy()
    ref = O.deform_conv_forward(xr, off, wr, (1, 1), (0, 3), (1, 1), nthreads=16)
    got = run(x, off, w, K, 4)
    assert torch.equal(got, run(x, off, w, K, 2))
    err = np.abs(got.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 1e-2     # bf16 tolerance of the north star


@pytest.mark.parametrize("dtype,channels_last", [(torch.float16, False), (torch.bfloat16, False),
                                                 (torch.float32, True)])
def test_algo4_mask_bias_layouts_and_dtypes(dtype, channels_last):
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
    assert torch.equal(run(x, off, w, K, 2, **kw), run(x, off, w, K, 4, **kw))


def test_algo4_fewer_output_channels():
    """Cout 48 < 64: padded weight rows, ragged channel stores."""
    B, Cin, H, W, K = 4, 256, 12, 20, 7
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=2)
    w = w[:48]
    for cl in (True, False):
        assert torch.equal(run(x, off, w, K, 2, channels_last=cl), run(x, off, w, K, 4, channels_last=cl))


def test_algo4_refused_where_it_does_not_apply():
    x, off, w, pts = synth.lfa_inputs(2, 256, 8, 12, 7, seed=1)
    with pytest.raises(Exception):
        run(x, off, w, 7, 4, dtype=torch.float32)          # exact / split float32 arithmetic
    x, off, w, pts = synth.lfa_inputs(1, 256, 80, 200, 9, seed=1)
    with pytest.raises(Exception):
        run(x, off, w, 9, 4)                               # 16000 pixels: more than a team's plane
