"""GPU parity of the deform-conv backward, through the C-ABI, against the
CPU oracle (oracle/ganet_oracle.c::oracle_deform_conv_backward, itself checked against
autograd in tests/test_oracle_dcn.py).  float32; tolerance 1e-4 of each gradient's scale
(fp32 atomic sums in an unspecified order on the GPU, double sums in the oracle)."""
import numpy as np
import pytest
import torch

from ganet_amd import _lib, dcn
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = [
    # B, Cin, H, W, Cout, kH, kW, stride, pad, dil, dg
    (2, 64, 9, 13, 32, 1, 7, (1, 1), (0, 3), (1, 1), 1),      # LFA form, small
    (1, 32, 10, 12, 16, 3, 3, (1, 1), (1, 1), (1, 1), 1),     # DCN 3x3
    (2, 64, 11, 14, 24, 3, 3, (2, 2), (1, 1), (1, 1), 2),     # stride 2, 2 deformable groups
    (1, 20, 12, 12, 8, 3, 3, (1, 1), (2, 2), (2, 2), 1),      # Cin padded to 32, dilation 2
    (1, 32, 7, 9, 70, 1, 9, (1, 1), (0, 4), (1, 1), 1),       # Cout > 64 (two output blocks)
    (3, 256, 8, 25, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1),     # the LFA's channel counts
]


def make(case, seed, modulated):
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, kH, kW)) * 0.2).astype(np.float32)
    Ho = O.dcn_out_size(H, pad[0], dil[0], kH, st[0])
    Wo = O.dcn_out_size(W, pad[1], dil[1], kW, st[1])
    off = (rng.standard_normal((B, dg * 2 * kH * kW, Ho, Wo)) * 1.7).astype(np.float32)
    mask = rng.uniform(0.1, 1, (B, dg * kH * kW, Ho, Wo)).astype(np.float32) if modulated else None
    go = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    return x, off, w, mask, go


def close(got, ref, what):
    got = got.detach().float().cpu().numpy()
    assert got.shape == ref.shape, what
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
    assert err <= TOL, "%s: relative error %.3g" % (what, err)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("channels_last", [False, True])
def test_backward_matches_oracle(case, modulated, channels_last):
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 5, modulated)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    tm = torch.from_numpy(mask).to(dev) if modulated else None
    if channels_last:
        tx = tx.contiguous(memory_format=torch.channels_last)
        tgo = tgo.contiguous(memory_format=torch.channels_last)
    gi, goff, gm, gw, gb = dcn.deform_conv_backward(
        tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm, need_bias=modulated)
    close(gi, ref["grad_x"], "grad_x")
    close(goff, ref["grad_offset"], "grad_offset")
    close(gw, ref["grad_weight"], "grad_weight")
    if modulated:
        close(gm, ref["grad_mask"], "grad_mask")
        close(gb, ref["grad_bias"], "grad_bias")
    assert gi.is_contiguous(memory_format=torch.channels_last) == (
        channels_last and tx.is_contiguous(memory_format=torch.channels_last))


COMMUTED = [
    # shapes the commuted form takes (one deformable group, stride 1, Cout and Cin in 64-blocks)
    (3, 256, 8, 25, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1),     # the LFA's channel counts
    (2, 256, 9, 11, 128, 3, 3, (1, 1), (1, 1), (1, 1), 1),    # two output blocks, 3x3
    (2, 128, 10, 12, 64, 3, 3, (1, 1), (2, 2), (2, 2), 1),    # dilation 2
    (1, 64, 7, 9, 64, 1, 9, (1, 1), (0, 4), (1, 1), 1),       # Cout == Cin: only when forced
    (2, 192, 12, 10, 64, 1, 5, (1, 1), (0, 0), (1, 1), 1),    # output smaller than input
    # the other tile counts of the split weight-gradient kernel (T*Cout/64 = 5, 4, 6, 2, 1)
    (1, 128, 6, 7, 64, 1, 5, (1, 1), (0, 2), (1, 1), 1),
    (1, 128, 6, 7, 64, 2, 2, (1, 1), (0, 0), (1, 1), 1),
    (1, 128, 5, 9, 128, 1, 3, (1, 1), (0, 1), (1, 1), 1),
    (1, 128, 5, 9, 128, 1, 1, (1, 1), (0, 0), (1, 1), 1),
    (1, 128, 5, 9, 64, 1, 1, (1, 1), (0, 0), (1, 1), 1),
]


@pytest.mark.parametrize("case", COMMUTED)
@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("algo", [1, 2])
def test_backward_direct_and_commuted_forms(case, modulated, algo):
    """algo 1 = columns formed and consumed on chip, algo 2 = Cout-wide rows through the
    per-tap products: both against the oracle, both memory formats."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 23, modulated)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    dev = torch.device("cuda", 0)
    for channels_last in (False, True):
        tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
        tm = torch.from_numpy(mask).to(dev) if modulated else None
        if channels_last:
            tx = tx.contiguous(memory_format=torch.channels_last)
            tgo = tgo.contiguous(memory_format=torch.channels_last)
        gi, goff, gm, gw, gb = dcn.deform_conv_backward(
            tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm, need_bias=modulated, algo=algo)
        close(gi, ref["grad_x"], "grad_x")
        close(goff, ref["grad_offset"], "grad_offset")
        close(gw, ref["grad_weight"], "grad_weight")
        if modulated:
            close(gm, ref["grad_mask"], "grad_mask")
            close(gb, ref["grad_bias"], "grad_bias")
    # weight gradient alone (no per-tap products, no coordinate gradients)
    _, _, _, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                              need_input=False, algo=algo)
    close(gw, ref["grad_weight"], "grad_weight (alone)")


def test_commuted_form_when_every_sample_lands_on_one_spot():
    """All pixels and taps sample the same position: the LDS list of the binned grad_Y
    kernel (2560 entries per 128-pixel tile) overflows and the inline path takes over."""
    B, Cin, H, W, Cout, kH, kW = 1, 128, 20, 60, 64, 1, 3
    st, pad, dil = (1, 1), (0, 1), (1, 1)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, kH, kW)) * 0.2).astype(np.float32)
    go = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    off = np.zeros((B, 2 * kW, H, W), np.float32)
    hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for t in range(kW):
        off[0, 2 * t] = 3.3 - hh
        off[0, 2 * t + 1] = 7.6 - (ww - pad[1] + t)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, 1)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    for algo in (1, 2):
        gi, goff, _, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, 1, algo=algo)
        close(gi, ref["grad_x"], "grad_x")
        close(goff, ref["grad_offset"], "grad_offset")
        close(gw, ref["grad_weight"], "grad_weight")


def test_commuted_form_walks_the_batch_in_chunks(monkeypatch):
    """A 1 MB budget for the per-tap products makes two chunks of the 3-image case (the
    default budget holds 35 LFA-sized images, so small cases never leave the first chunk)."""
    monkeypatch.setenv("GANET_BWD_BUDGET_MB", "1")
    case = COMMUTED[0]
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 31, True)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo, tm = (torch.from_numpy(a).to(dev) for a in (x, off, w, go, mask))
    gi, goff, gm, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                                   algo=2)
    close(gi, ref["grad_x"], "grad_x")
    close(goff, ref["grad_offset"], "grad_offset")
    close(gm, ref["grad_mask"], "grad_mask")
    close(gw, ref["grad_weight"], "grad_weight")


def test_weight_gradient_planes_chunks_and_atomic_flush_agree(monkeypatch):
    """The split weight-gradient kernel leaves its accumulators in planes of partial sums that a
    second kernel adds in plane order; with several chunks the later ones add into the planes;
    GANET_BWD_WGRAD_ATOMIC=1 (the A/B knob) keeps the atomic flush.  All three agree with the
    oracle and with one another to rounding (grad_Y's LDS bins add in arrival order, so not
    even two runs of one path are bit-identical)."""
    case = COMMUTED[0]
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, _, go = make(case, 37, False)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg)["grad_weight"]
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))

    def gw():
        return dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, algo=2,
                                        need_input=False)[3]

    whole = gw()
    close(whole, ref, "grad_weight, planes")
    monkeypatch.setenv("GANET_BWD_BUDGET_MB", "1")       # two chunks: the second adds into the planes
    close(gw(), ref, "grad_weight, planes, two chunks")
    monkeypatch.delenv("GANET_BWD_BUDGET_MB")
    monkeypatch.setenv("GANET_BWD_WGRAD_ATOMIC", "1")
    close(gw(), ref, "grad_weight, atomic flush")


def test_commuted_form_on_a_large_map():
    """80x200 (the C4 feature map): 125 tiles of 128 pixels per image and tap in the binned
    grad_Y kernel, several 16-pixel steps per share in the weight-gradient kernel."""
    case = (1, 128, 80, 200, 64, 1, 3, (1, 1), (0, 1), (1, 1), 1)
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, _, go = make(case, 41, False)
    off *= 3.0                                   # samples reach a few rows away
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    tx = tx.contiguous(memory_format=torch.channels_last)
    gi, goff, _, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, algo=2)
    close(gi, ref["grad_x"], "grad_x")
    close(goff, ref["grad_offset"], "grad_offset")
    close(gw, ref["grad_weight"], "grad_weight")


def test_commuted_form_with_the_atomic_scatter(monkeypatch):
    """Maps above 512 tiles of 128 pixels form grad_Y through global atomics instead of the
    LDS bins; the knob takes that path on a small case."""
    monkeypatch.setenv("GANET_BWD_SCATTER_ATOMIC", "1")
    case = COMMUTED[1]
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 37, True)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo, tm = (torch.from_numpy(a).to(dev) for a in (x, off, w, go, mask))
    gi, goff, gm, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                                   algo=2)
    close(gi, ref["grad_x"], "grad_x")
    close(goff, ref["grad_offset"], "grad_offset")
    close(gm, ref["grad_mask"], "grad_mask")
    close(gw, ref["grad_weight"], "grad_weight")
    _, _, _, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                              need_input=False, algo=2)
    close(gw, ref["grad_weight"], "grad_weight (alone)")


def test_commuted_form_is_refused_where_it_does_not_apply():
    dev = torch.device("cuda", 0)
    x = torch.zeros(1, 64, 6, 6, device=dev)
    with pytest.raises(_lib.GanetError):     # stride 2
        dcn.deform_conv_backward(x, torch.zeros(1, 18, 3, 3, device=dev),
                                 torch.zeros(64, 64, 3, 3, device=dev),
                                 torch.zeros(1, 64, 3, 3, device=dev), (2, 2), (1, 1), (1, 1),
                                 algo=2)


def test_autograd_through_the_modules():
    case = CASES[0]
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 9, True)
    dev = torch.device("cuda", 0)
    # DCNv1 module (DeformConv1D as the LFA uses it)
    m = dcn.DeformConv1D(Cin, Cout, kW, stride=1, padding=pad[1], dilation=1).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w))
    tx = torch.from_numpy(x).to(dev).requires_grad_(True)
    toff = torch.from_numpy(off).to(dev).requires_grad_(True)
    out = m(tx, toff)
    out.backward(torch.from_numpy(go).to(dev))
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg)
    close(tx.grad, ref["grad_x"], "grad_x")
    close(toff.grad, ref["grad_offset"], "grad_offset")
    close(m.weight.grad, ref["grad_weight"], "grad_weight")
    # weight-only gradient (input and offset detached)
    m.weight.grad = None
    m(tx.detach(), toff.detach()).backward(torch.from_numpy(go).to(dev))
    close(m.weight.grad, ref["grad_weight"], "grad_weight (alone)")
    # DCNv2 with bias
    tb = torch.zeros(Cout, device=dev, requires_grad=True)
    tw = torch.from_numpy(w).to(dev).requires_grad_(True)
    tm = torch.from_numpy(mask).to(dev).requires_grad_(True)
    tx.grad = toff.grad = None
    out = dcn.modulated_deform_conv(tx, toff, tm, tw, tb, 1, pad, 1, 1, dg)
    out.backward(torch.from_numpy(go).to(dev))
    ref2 = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    close(tx.grad, ref2["grad_x"], "v2 grad_x")
    close(toff.grad, ref2["grad_offset"], "v2 grad_offset")
    close(tm.grad, ref2["grad_mask"], "v2 grad_mask")
    close(tw.grad, ref2["grad_weight"], "v2 grad_weight")
    close(tb.grad, ref2["grad_bias"], "v2 grad_bias")


def test_backward_rejects_what_it_does_not_build():
    dev = torch.device("cuda", 0)
    # deformable groups of 16 channels (tiles of 32 would straddle them): status UNSUPPORTED
    x = torch.zeros(1, 32, 6, 6, device=dev)
    with pytest.raises(_lib.GanetError):
        dcn.deform_conv_backward(x, torch.zeros(1, 36, 6, 6, device=dev),
                                 torch.zeros(8, 32, 3, 3, device=dev),
                                 torch.zeros(1, 8, 6, 6, device=dev), (1, 1), (1, 1), (1, 1),
                                 1, 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5]])
@pytest.mark.parametrize("modulated,channels_last", [(False, True), (True, False)])
def test_backward_16bit(dtype, case, modulated, channels_last):
    """bf16 / f16 tensors (the reference dispatches half in its backward,
    deform_conv_cuda_kernel.cu:259): x, weight and grad_out are widened inside the library, so
    against the oracle on the SAME rounded values the only differences are float32 summation
    order and one rounding of each gradient to 16 bits: 1e-2 of the gradient's scale (the
    north star's 16-bit tolerance), grad_offset / grad_mask stay float32 (1e-4)."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 8, modulated)
    dev = torch.device("cuda", 0)
    tx, tw, tgo = (torch.from_numpy(a).to(dev).to(dtype) for a in (x, w, go))
    ref = O.deform_conv_backward(tx.float().cpu().numpy(), off, tw.float().cpu().numpy(),
                                 tgo.float().cpu().numpy(), st, pad, dil, 1, dg, mask=mask)
    toff = torch.from_numpy(off).to(dev)
    tm = torch.from_numpy(mask).to(dev) if modulated else None
    if channels_last:
        tx = tx.contiguous(memory_format=torch.channels_last)
        tgo = tgo.contiguous(memory_format=torch.channels_last)
    gi, goff, gm, gw, gb = dcn.deform_conv_backward(
        tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm, need_bias=modulated)
    assert gi.dtype == dtype and gw.dtype == dtype and goff.dtype == torch.float32

    def rel(got, want):
        got = got.detach().float().cpu().numpy()
        return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)

    assert rel(gi, ref["grad_x"]) <= 1e-2
    assert rel(gw, ref["grad_weight"]) <= 1e-2
    assert rel(goff, ref["grad_offset"]) <= 1e-4
    if modulated:
        assert rel(gm, ref["grad_mask"]) <= 1e-4
        assert rel(gb, ref["grad_bias"]) <= 1e-2


GROUPED = [
    # B, Cin, H, W, Cout, kH, kW, stride, pad, dil, dg, groups
    (2, 64, 9, 13, 32, 3, 3, (1, 1), (1, 1), (1, 1), 1, 2),
    (1, 128, 8, 10, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1, 4),
    (2, 64, 11, 14, 24, 3, 3, (2, 2), (1, 1), (1, 1), 2, 2),   # with 2 deformable groups
    (2, 256, 8, 12, 64, 1, 7, (1, 1), (0, 3), (1, 1), 1, 2),   # the commuted forms' shape
]


@pytest.mark.parametrize("case", GROUPED)
@pytest.mark.parametrize("modulated", [False, True])
def test_conv_groups_forward_and_backward(case, modulated):
    """Convolution groups > 1 (one addmm_ per group in the reference,
    deform_conv_cuda.cpp:227-236, :439-470): weight [Cout, Cin/groups, kH, kW]; forward and
    all gradients against the oracle, float32."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg, g = case
    x, off, _, mask, go = make(case[:-1], 13, modulated)
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((Cout, Cin // g, kH, kW)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) if modulated else None
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    tm = torch.from_numpy(mask).to(dev) if modulated else None
    fref = O.deform_conv_forward(x, off, w, st, pad, dil, g, dg, mask=mask, bias=bias)
    for algo in (0, 1):
        out = dcn.deform_conv_forward(tx, toff, tw, st, pad, dil, g, dg, mask=tm,
                                      bias=None if bias is None else torch.from_numpy(bias).to(dev),
                                      algo=algo)
        close(out, fref, "forward algo %d" % algo)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, g, dg, mask=mask)
    gi, goff, gm, gw, gb = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, g, dg,
                                                    mask=tm, need_bias=modulated)
    assert tuple(gw.shape) == (Cout, Cin // g, kH, kW)
    close(gi, ref["grad_x"], "grad_x")
    close(goff, ref["grad_offset"], "grad_offset")
    close(gw, ref["grad_weight"], "grad_weight")
    if modulated:
        close(gm, ref["grad_mask"], "grad_mask")
        close(gb, ref["grad_bias"], "grad_bias")
    # and through the module API
    m = dcn.DeformConv(Cin, Cout, (kH, kW), st, pad, dil, groups=g, deformable_groups=dg).to(dev)
    with torch.no_grad():
        m.weight.copy_(tw)
    if not modulated:
        close(m(tx, toff), fref, "DeformConv(groups=%d)" % g)


KEPT = [COMMUTED[0], COMMUTED[1], COMMUTED[2], COMMUTED[4]]   # 2*Cout <= Cin: commuted by default


@pytest.mark.parametrize("case", KEPT)
@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("channels_last", [False, True])
def test_backward_on_the_forwards_products(case, modulated, channels_last):
    """GANET_DCN_ALGO_KEEP_Y + ganet_deform_conv_backward_y: the forward leaves the per-tap
    products of the whole batch in its workspace, the backward reads them instead of
    contracting x again.  Same output, same gradients (the products are formed by the same
    three bf16 passes; sums through atomics differ in order only), and the oracle's."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 21, modulated)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    tm = torch.from_numpy(mask).to(dev) if modulated else None
    tb = torch.zeros(Cout, device=dev) if modulated else None
    if channels_last:
        tx = tx.contiguous(memory_format=torch.channels_last)
        tgo = tgo.contiguous(memory_format=torch.channels_last)
    keep = {}
    out_k = dcn.deform_conv_forward(tx, toff, tw, st, pad, dil, 1, dg, mask=tm, bias=tb, keep_y=keep)
    # against algorithm 2 without the bit (by itself the library takes algorithm 5 at these sizes)
    out = dcn.deform_conv_forward(tx, toff, tw, st, pad, dil, 1, dg, mask=tm, bias=tb, algo=2)
    assert "y" in keep and keep["y"].numel() == B * H * W * kH * kW * Cout
    assert torch.equal(out, out_k)
    with_y = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                      need_bias=modulated, y_saved=keep["y"])
    plain = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg, mask=tm,
                                     need_bias=modulated)
    for a, b in zip(with_y, plain):
        assert (a is None) == (b is None)
        if a is not None:
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 1e-5 * max(scale, 1e-30)   # order of the atomic sums
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg, mask=mask)
    close(with_y[0], ref["grad_x"], "grad_x")
    close(with_y[1], ref["grad_offset"], "grad_offset")
    close(with_y[3], ref["grad_weight"], "grad_weight")
    if modulated:
        close(with_y[2], ref["grad_mask"], "grad_mask")


def test_products_are_kept_only_where_the_library_can():
    dev = torch.device("cuda", 0)
    # Cout 32 (not a 64-block), bf16 tensors, stride 2: the holder stays empty, the call works
    for (Cin, Cout, dtype, st) in [(64, 32, torch.float32, (1, 1)), (256, 64, torch.bfloat16, (1, 1)),
                                   (256, 64, torch.float32, (2, 2))]:
        x = torch.randn(2, Cin, 8, 12, device=dev).to(dtype)
        w = torch.randn(Cout, Cin, 1, 7, device=dev).to(dtype) * 0.1
        Wo = (12 + 6 - 7) // st[1] + 1
        Ho = (8 - 1) // st[0] + 1
        off = torch.randn(2, 14, Ho, Wo, device=dev)
        keep = {}
        out = dcn.deform_conv_forward(x, off, w, st, (0, 3), (1, 1), keep_y=keep)
        assert keep == {}
        assert torch.equal(out, dcn.deform_conv_forward(x, off, w, st, (0, 3), (1, 1)))
    # the workspace-size limit of the Python layer
    old = dcn.config.keep_y_max_bytes
    try:
        dcn.config.keep_y_max_bytes = 1024
        x = torch.randn(2, 256, 8, 12, device=dev)
        w = torch.randn(64, 256, 1, 7, device=dev) * 0.1
        keep = {}
        dcn.deform_conv_forward(x, torch.randn(2, 14, 8, 12, device=dev), w, (1, 1), (0, 3), (1, 1),
                                keep_y=keep)
        assert keep == {}
    finally:
        dcn.config.keep_y_max_bytes = old


def test_autograd_keeps_the_products_at_the_lfa_shape():
    """DeformConv1D at the LFA's channel counts: the autograd path keeps the products by
    default; gradients against the oracle, and equal to the path with keeping switched off."""
    case = CASES[5]
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 13, False)
    dev = torch.device("cuda", 0)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg)
    grads = []
    for keep in (True, False):
        old = dcn.config.keep_y_for_backward
        dcn.config.keep_y_for_backward = keep
        try:
            m = dcn.DeformConv1D(Cin, Cout, kW, stride=1, padding=pad[1], dilation=1).to(dev)
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(w))
            tx = torch.from_numpy(x).to(dev).requires_grad_(True)
            toff = torch.from_numpy(off).to(dev).requires_grad_(True)
            out = m(tx, toff)
            kept = out.grad_fn.y_saved is not None
            assert kept == keep
            out.backward(torch.from_numpy(go).to(dev))
        finally:
            dcn.config.keep_y_for_backward = old
        close(tx.grad, ref["grad_x"], "grad_x")
        close(toff.grad, ref["grad_offset"], "grad_offset")
        close(m.weight.grad, ref["grad_weight"], "grad_weight")
        grads.append((tx.grad, toff.grad, m.weight.grad))
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5], COMMUTED[1]])
@pytest.mark.parametrize("channels_last", [False, True])
def test_grad_x_written_or_added_into(case, channels_last):
    """GANET_DCN_BWD_OVERWRITE_X (the Python layer's default): grad_x comes back complete in
    an uninitialised tensor; without the bit the call adds into the caller's zero-filled
    tensor as the reference does (deform_conv.py:73-75).  Direct and commuted forms, both
    layouts."""
    B, Cin, H, W, Cout, kH, kW, st, pad, dil, dg = case
    x, off, w, mask, go = make(case, 17, False)
    ref = O.deform_conv_backward(x, off, w, go, st, pad, dil, 1, dg)
    dev = torch.device("cuda", 0)
    tx, toff, tw, tgo = (torch.from_numpy(a).to(dev) for a in (x, off, w, go))
    if channels_last:
        tx = tx.contiguous(memory_format=torch.channels_last)
        tgo = tgo.contiguous(memory_format=torch.channels_last)
    # poison the allocator's next blocks so that a missing store would show
    junk = [torch.full_like(tx, float("nan")) for _ in range(3)]
    del junk
    for overwrite in (True, False):
        gi, goff, _, gw, _ = dcn.deform_conv_backward(tx, toff, tw, tgo, st, pad, dil, 1, dg,
                                                      overwrite=overwrite)
        close(gi, ref["grad_x"], "grad_x (overwrite=%s)" % overwrite)
        close(goff, ref["grad_offset"], "grad_offset")
        close(gw, ref["grad_weight"], "grad_weight")
