"""Pins the C restatement of the deform-conv forward (oracle/ganet_oracle.c).

The reference holds no fixtures for this op and its CUDA extension cannot run
here, so the oracle is cross-checked against (i) an independent PyTorch
formulation (integer-index gather of the four neighbours + einsum) and
(ii) known-answer cases.  Parity with the reference itself stays UNPINNED.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ganet_amd import synth
from oracle import cpu_oracle as O


def torch_deform_conv(x, offset, weight, stride, padding, dilation, dg=1, mask=None,
                      bias=None):
    """Independent float64 formulation: explicit neighbour gather."""
    x = torch.as_tensor(x, dtype=torch.float64)
    offset = torch.as_tensor(offset, dtype=torch.float64)
    weight = torch.as_tensor(weight, dtype=torch.float64)
    if mask is not None:
        mask = torch.as_tensor(mask, dtype=torch.float64)
    if bias is not None:
        bias = torch.as_tensor(bias, dtype=torch.float64)
    return torch_deform_conv_t(x, offset, weight, stride, padding, dilation, dg, mask,
                               bias).numpy()


def torch_deform_conv_t(x, offset, weight, stride, padding, dilation, dg=1, mask=None,
                        bias=None):
    """The same on float64 tensors, differentiable (autograd is the checker of the
    oracle's backward)."""
    B, C, H, W = x.shape
    Co, _, kH, kW = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kH - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kW - 1) + 1)) // stride[1] + 1
    ys = torch.arange(Ho, dtype=torch.float64).view(1, Ho, 1) * stride[0] - padding[0]
    xs = torch.arange(Wo, dtype=torch.float64).view(1, 1, Wo) * stride[1] - padding[1]
    cpg = C // dg
    cols = [[None] * (kH * kW) for _ in range(dg)]
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        for i in range(kH):
            for j in range(kW):
                t = i * kW + j
                oh = offset[:, g * 2 * kH * kW + 2 * t]
                ow = offset[:, g * 2 * kH * kW + 2 * t + 1]
                # float32 coordinates like the kernel, then exact float64 blend
                hy = (ys + i * dilation[0]).float() + oh.float()
                wx = (xs + j * dilation[1]).float() + ow.float()
                inside = (hy > -1) & (wx > -1) & (hy < H) & (wx < W)
                h0 = torch.floor(hy)
                w0 = torch.floor(wx)
                lh = (hy - h0).double()
                lw = (wx - w0).double()
                h0 = h0.long()
                w0 = w0.long()
                val = 0
                for (dh, dw, wt) in [(0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                                     (1, 0, lh * (1 - lw)), (1, 1, lh * lw)]:
                    hh, ww = h0 + dh, w0 + dw
                    ok = inside & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
                    idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).view(B, 1, -1)
                    v = torch.gather(xg.reshape(B, cpg, H * W), 2,
                                     idx.expand(B, cpg, -1)).view(B, cpg, Ho, Wo)
                    val = val + (wt * ok).unsqueeze(1) * v
                if mask is not None:
                    val = val * mask[:, g * kH * kW + t].unsqueeze(1)
                cols[g][t] = val
    cols = torch.stack([torch.stack(cols[g], dim=2) for g in range(dg)], dim=1)
    cols = cols.reshape(B, C, kH, kW, Ho, Wo)
    out = torch.einsum("ockl,bcklhw->bohw", weight.view(Co, C, kH, kW), cols)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


CASES = [
    # B, Cin, H, W, Cout, kH, kW, stride, pad, dil, dg
    (2, 16, 9, 13, 8, 1, 7, (1, 1), (0, 3), (1, 1), 1),     # the LFA's 1xK form
    (1, 8, 10, 12, 6, 3, 3, (1, 1), (1, 1), (1, 1), 1),     # DCN 3x3
    (2, 8, 11, 14, 4, 3, 3, (2, 2), (1, 1), (1, 1), 2),     # stride 2, 2 deformable groups
    (1, 4, 12, 12, 4, 3, 3, (1, 1), (2, 2), (2, 2), 1),     # dilation 2
    (1, 6, 7, 9, 5, 1, 9, (1, 1), (0, 4), (1, 1), 3),       # K=9, 3 deformable groups
]


@pytest.mark.parametrize("case", CASES)
def test_oracle_vs_torch_formulation(case):
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = rng.standard_normal((Cout, Cin, kH, kW)).astype(np.float32) * 0.2
    Ho = O.dcn_out_size(H, pad[0], dil[0], kH, st[0])
    Wo = O.dcn_out_size(W, pad[1], dil[1], kW, st[1])
    off = (rng.standard_normal((B, dg * 2 * kH * kW, Ho, Wo)) * 2.5).astype(np.float32)
    got = O.deform_conv_forward(x, off, w, st, pad, dil, 1, dg)
    ref = torch_deform_conv(x, off, w, st, pad, dil, dg)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-5 * scale
    # float32-accumulating variant agrees too
    got32 = O.deform_conv_forward(x, off, w, st, pad, dil, 1, dg, acc64=False)
    assert np.abs(got32 - ref).max() <= 1e-4 * scale


def test_oracle_modulated_vs_torch_formulation():
    rng = np.random.default_rng(2)
    B, Cin, H, W, Cout, k, dg = 2, 8, 9, 11, 6, 3, 2
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = rng.standard_normal((Cout, Cin, k, k)).astype(np.float32) * 0.2
    off = (rng.standard_normal((B, dg * 2 * k * k, H, W)) * 2).astype(np.float32)
    mask = rng.uniform(0, 1, (B, dg * k * k, H, W)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    got = O.deform_conv_forward(x, off, w, (1, 1), (1, 1), (1, 1), 1, dg, mask=mask,
                                bias=bias)
    ref = torch_deform_conv(x, off, w, (1, 1), (1, 1), (1, 1), dg, mask=mask, bias=bias)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_zero_offsets_equal_plain_convolution():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 10, 16)).astype(np.float32)
    w = rng.standard_normal((5, 8, 1, 7)).astype(np.float32)
    off = np.zeros((2, 14, 10, 16), np.float32)
    got = O.deform_conv_forward(x, off, w, (1, 1), (0, 3), (1, 1))
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                   padding=(0, 3)).numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_integer_offsets_equal_shifted_convolution():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 4, 12, 14)).astype(np.float32)
    w = rng.standard_normal((3, 4, 3, 3)).astype(np.float32)
    off = np.zeros((1, 18, 12, 14), np.float32)
    off[:, 0::2] = 2.0   # dy = +2
    off[:, 1::2] = -3.0  # dx = -3
    got = O.deform_conv_forward(x, off, w, (1, 1), (1, 1), (1, 1))
    # sampling x at (h+2, w-3) with zero outside == convolving a shifted image
    xs = np.zeros_like(x)
    xs[:, :, :10, 3:] = x[:, :, 2:, :11]
    ref = F.conv2d(torch.from_numpy(xs).double(), torch.from_numpy(w).double(),
                   padding=1).numpy()
    # rows/cols whose taps cross the true border differ by construction: compare interior
    assert np.abs(got[:, :, 1:8, 5:13] - ref[:, :, 1:8, 5:13]).max() <= 1e-5


def test_samples_exactly_on_the_border_are_zero():
    # h_im == -1, == H, w_im == -1, == W must contribute nothing
    # (strict comparisons at deform_conv_cuda_kernel.cu:229)
    x = np.ones((1, 1, 6, 8), np.float32)
    w = np.ones((1, 1, 1, 1), np.float32)
    off = np.zeros((1, 2, 6, 8), np.float32)
    off[0, 0, 0, :] = -1.0          # row 0 samples h = -1
    off[0, 0, 5, :] = 1.0           # row 5 samples h = 6 == H
    off[0, 1, 2, 0] = -1.0          # w = -1
    off[0, 1, 3, 7] = 1.0           # w = 8 == W
    off[0, 0, 1, 3] = -1.5          # h = -0.5: half weight from row 0
    got = O.deform_conv_forward(x, off, w)
    assert np.all(got[0, 0, 0] == 0) and np.all(got[0, 0, 5] == 0)
    assert got[0, 0, 2, 0] == 0 and got[0, 0, 3, 7] == 0
    assert got[0, 0, 1, 3] == 0.5
    assert got[0, 0, 4, 4] == 1.0


def test_bilinear_is_exact_on_a_linear_ramp():
    H, W = 9, 11
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    x = (0.5 * yy + 0.25 * xx).astype(np.float32)[None, None]
    w = np.ones((1, 1, 1, 1), np.float32)
    rng = np.random.default_rng(5)
    off = rng.uniform(-0.75, 0.75, (1, 2, H, W)).astype(np.float32)
    got = O.deform_conv_forward(x, off, w)[0, 0]
    hs, ws = yy + off[0, 0], xx + off[0, 1]
    inside = (hs >= 0) & (hs <= H - 1) & (ws >= 0) & (ws <= W - 1)
    ref = 0.5 * hs + 0.25 * ws
    assert np.abs(got - ref)[inside].max() <= 1e-5


def test_lfa_synthetic_shape_runs_and_has_out_of_bounds_samples():
    x, off, w, pts = synth.lfa_inputs(1, 16, 20, 50, 7, Cout=8, seed=0)
    got = O.deform_conv_forward(x, off, w, (1, 1), (0, 3), (1, 1))
    assert got.shape == (1, 8, 20, 50) and np.isfinite(got).all()
    # base subtraction: offset = pts - [0,-p, 0,-p+1, ...]
    assert np.allclose(off[0, 1::2, 3, 4] + np.arange(-3, 4), pts[0, 1::2, 3, 4])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("modulated", [False, True])
def test_oracle_backward_vs_autograd(case, modulated):
    """oracle_deform_conv_backward (the CUDA code's col2im / col2im_coord / weight GEMM
    restated) against autograd through the float64 formulation above."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    rng = np.random.default_rng(123)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, kH, kW)) * 0.2).astype(np.float32)
    Ho = O.dcn_out_size(H, pad[0], dil[0], kH, st[0])
    Wo = O.dcn_out_size(W, pad[1], dil[1], kW, st[1])
    off = (rng.standard_normal((B, dg * 2 * kH * kW, Ho, Wo)) * 1.7).astype(np.float32)
    mask = rng.uniform(0.1, 1, (B, dg * kH * kW, Ho, Wo)).astype(np.float32) if modulated else None
    go = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    got = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)

    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    toff = torch.tensor(off, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    tm = torch.tensor(mask, dtype=torch.float64, requires_grad=True) if modulated else None
    tb = torch.zeros(Cout, dtype=torch.float64, requires_grad=True) if modulated else None
    out = torch_deform_conv_t(tx, toff, tw, st, pad, dil, dg, tm, tb)
    (out * torch.tensor(go, dtype=torch.float64)).sum().backward()

    def close(a, b, what):
        scale = max(np.abs(b).max(), 1e-30)
        assert np.abs(a - b).max() / scale < 2e-5, what
    close(got["grad_x"], tx.grad.numpy(), "grad_x")
    close(got["grad_offset"], toff.grad.numpy(), "grad_offset")
    close(got["grad_weight"], tw.grad.numpy(), "grad_weight")
    if modulated:
        close(got["grad_mask"], tm.grad.numpy(), "grad_mask")
        close(got["grad_bias"], tb.grad.numpy(), "grad_bias")
