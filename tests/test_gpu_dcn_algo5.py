"""GPU parity of algorithm 5 (small batches: the taps of a pixel tile on different wavefronts,
dcn_taps_kernel) against the CPU oracle, through the C-ABI.  Tolerances as everywhere: float32
<= 1e-4, 16-bit <= 1e-2 of the output's scale, the oracle reading the rounded inputs."""
import ctypes

import numpy as np
import pytest
import torch

from ganet_amd import _lib, dcn, synth
from oracle import cpu_oracle as O
from test_gpu_dcn import TOL_16, TOL_F32, make, rel_err, run_gpu

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, H, W, Cout, kH, kW, stride, pad, dil, dg
    (1, 256, 40, 100, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1),   # BASELINE's C1
    (2, 64, 12, 20, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1),
    (2, 64, 11, 14, 96, 3, 3, (2, 2), (1, 1), (1, 1), 1),     # 9 taps on 8 waves, two output blocks, stride 2
    (1, 48, 10, 12, 20, 3, 3, (1, 1), (2, 2), (2, 2), 1),     # padded Cin / Cout, dilation
    (3, 32, 7, 9, 8, 1, 9, (1, 1), (0, 4), (1, 1), 1),        # tiles that straddle images
    (1, 128, 6, 10, 64, 1, 1, (1, 1), (0, 0), (1, 1), 1),     # one tap: one wavefront
    (1, 192, 9, 13, 40, 4, 5, (1, 1), (2, 2), (1, 1), 1),     # 20 taps: three per wavefront at most
]


@pytest.mark.parametrize("case", CASES)
def test_algo5_matches_oracle(case):
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w = make(*case, seed=23)
    ref = O.deform_conv_forward(x, off, w, st, pad, dil, 1, dg, nthreads=4)
    for cl in (False, True):
        got = run_gpu(x, off, w, st, pad, dil, dg, channels_last=cl, algo=5)
        assert got.shape == ref.shape
        assert rel_err(got, ref) <= TOL_F32
    for dtype in (torch.bfloat16, torch.float16):
        xr = torch.from_numpy(x).to(dtype).float().numpy()
        wr = torch.from_numpy(w).to(dtype).float().numpy()
        ref16 = O.deform_conv_forward(xr, off, wr, st, pad, dil, 1, dg, nthreads=4)
        for cl in (False, True):
            got = run_gpu(x, off, w, st, pad, dil, dg, dtype=dtype, channels_last=cl, algo=5)
            assert rel_err(got, ref16) <= TOL_16
    got = run_gpu(x, off, w, st, pad, dil, dg, mfma_bf16=True, algo=5)   # float32 tensors, one bf16 pass
    assert rel_err(got, ref) <= TOL_16


def test_algo5_mask_bias_relu_points_and_groups():
    rng = np.random.default_rng(5)
    B, Cin, H, W, Cout, K = 2, 256, 10, 25, 64, 7
    x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, Cout=Cout, seed=9)
    mask = rng.uniform(0, 1, (B, K, H, W)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = np.maximum(O.deform_conv_forward(x, off, w, (1, 1), (0, 3), (1, 1), mask=mask, bias=bias), 0)
    for cl in (False, True):
        got = run_gpu(x, off, w, (1, 1), (0, 3), (1, 1), mask=mask, bias=bias, algo=5,
                      fuse_relu=True, channels_last=cl)
        assert rel_err(got, ref) <= TOL_F32
        gotp = run_gpu(x, pts, w, (1, 1), (0, 3), (1, 1), mask=mask, bias=bias, algo=5,
                       fuse_relu=True, channels_last=cl, offset_is_points=True)
        assert rel_err(gotp, ref) <= TOL_F32
    # convolution groups: the packed weights are block diagonal
    xg, offg, _ = make(2, 64, 9, 11, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1, seed=3)
    wg = (rng.standard_normal((64, 16, 3, 3)) / 12).astype(np.float32)
    refg = O.deform_conv_forward(xg, offg, wg, (1, 1), (1, 1), (1, 1), 4, 1)
    out = dcn.deform_conv_forward(torch.from_numpy(xg).cuda(), torch.from_numpy(offg).cuda(),
                                  torch.from_numpy(wg).cuda(), (1, 1), (1, 1), (1, 1), 4, 1, algo=5)
    assert rel_err(out.cpu().numpy(), refg) <= TOL_F32


def test_algo5_is_what_small_batches_get_and_is_repeatable():
    """The automatic choice at BASELINE's C1 is algorithm 5; at C3 it stays algorithm 2; two
    launches on the same inputs give the same bits (the partial blocks are summed in a fixed
    order); two deformable groups and the exact-float32 mode are refused for an explicit 5."""
    L = _lib.lib()

    def params(B, dtype=_lib.GANET_BF16, dg=1, mfma=0, algo=0):
        p = _lib.DcnParams()
        p.B, p.Cin, p.H, p.W, p.Cout = B, 256, 40, 100, 64
        p.kH, p.kW, p.sH, p.sW, p.pH, p.pW, p.dH, p.dW = 1, 7, 1, 1, 0, 3, 1, 1
        p.groups, p.deformable_groups = 1, dg
        p.dtype, p.mfma_bf16, p.algo = dtype, mfma, algo
        p.x_layout = p.out_layout = _lib.GANET_NHWC
        return p

    assert L.ganet_deform_conv_algorithm(ctypes.byref(params(1))) == 5
    assert L.ganet_deform_conv_algorithm(ctypes.byref(params(1, _lib.GANET_F32))) == 5
    assert L.ganet_deform_conv_algorithm(ctypes.byref(params(128))) == 2
    assert L.ganet_deform_conv_algorithm(ctypes.byref(params(1, dg=2, algo=5))) == 0
    assert L.ganet_deform_conv_algorithm(ctypes.byref(params(1, _lib.GANET_F32, mfma=2, algo=5))) == 0
    # the packed weights are the ones the other choice reads: same size at every batch
    assert (L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(params(1)))
            == L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(params(128))))
    x, off, w, _ = synth.lfa_inputs(1, 256, 40, 100, 7, Cout=64, seed=2)
    for dtype in (torch.float32, torch.bfloat16):
        a = run_gpu(x, off, w, (1, 1), (0, 3), (1, 1), dtype=dtype, channels_last=True)
        b = run_gpu(x, off, w, (1, 1), (0, 3), (1, 1), dtype=dtype, channels_last=True, algo=5)
        assert np.array_equal(a, b)
