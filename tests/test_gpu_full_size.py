"""The LFA at BASELINE.json's full per-GPU sizes (C2, C3, C4, C5), through the
C-ABI: against the CPU oracle on whole batches where it finishes in seconds (C2, C3) and on
a sample of images elsewhere, plus the size-independent properties the operator offers -
images are independent (a permuted batch gives the permuted result, bit for bit, across
the Y chunks and the two-row-block GEMM tiles) and the operator is linear in x (float32).
Tolerances as the north star states, 1e-4 (float32) / 1e-2 (bf16), read two ways: the largest
error against the output scale, and - element-wise, the stricter reading - every output's error
against the sum of the magnitudes of its own terms (sum_t sum_c |w| * blend(|x|): the same
oracle on |x| and |w|, the blend weights being non-negative).  Round 3 adds the per-GPU shares
of BASELINE's sharded configs that had never run: C4 at 512 and 256 images (its 2- and 4-GPU
shares; x is 4.2 GB at 512), C5 at stride 4 with 32 images."""
import numpy as np
import pytest
import torch

from ganet_amd import dcn, synth
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    ref = np.asarray(ref, np.float64)
    return np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)


def lfa(xt, pt, wt, K):
    out = dcn.deform_conv_forward(xt, pt, wt, (1, 1), (0, (K - 1) // 2), (1, 1),
                                  offset_is_points=True, fuse_relu=True)
    torch.cuda.synchronize()
    return out


def inputs(B, H, W, K, dtype, nseed=8, seed=0, Cin=256):
    """B images built from `nseed` seeded ones (each scaled differently so no two are
    equal): the generator costs more than the kernels at these sizes."""
    nseed = min(nseed, B)
    x, off, w, pts = synth.lfa_inputs(nseed, Cin, H, W, K, seed=seed)
    reps = (B + nseed - 1) // nseed
    scale = (1.0 + 0.03125 * np.arange(reps, dtype=np.float32)).repeat(nseed)[:B]
    xt = torch.from_numpy(x).cuda().repeat(reps, 1, 1, 1)[:B] * torch.from_numpy(scale).cuda()[:, None, None, None]
    pt = torch.from_numpy(pts).cuda().repeat(reps, 1, 1, 1)[:B].contiguous()
    xt = xt.to(dtype).contiguous(memory_format=torch.channels_last)
    return xt, pt, torch.from_numpy(w).cuda().to(dtype), pts


def oracle_images(xt, pts, wt, K, idx):
    x = xt[idx].float().cpu().numpy()
    w = wt.float().cpu().numpy()
    p = (K - 1) // 2
    base = np.zeros((2 * K, 1, 1), np.float32)
    base[1::2, 0, 0] = np.arange(-p, p + 1, dtype=np.float32)
    off = pts[np.asarray(idx) % pts.shape[0]] - base[None]
    ref = O.deform_conv_forward(np.ascontiguousarray(x), np.ascontiguousarray(off), w,
                                (1, 1), (0, p), (1, 1), nthreads=16)
    mag = O.deform_conv_forward(np.ascontiguousarray(np.abs(x)), np.ascontiguousarray(off),
                                np.abs(w), (1, 1), (0, p), (1, 1), nthreads=16)
    return np.maximum(ref, 0), mag


def term_err(got, ref, mag):
    """Largest error of an output against the sum of the magnitudes of its own terms (ReLU
    does not increase a difference)."""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    return (d / np.maximum(np.asarray(mag, np.float64), 1e-30)).max()


@pytest.mark.parametrize("name,B,H,W,K,dtype,tol,sample,Cin", [
    ("c3", 128, 40, 100, 7, torch.bfloat16, 1e-2, None, 256),   # whole batch against the oracle
    ("c2", 32, 40, 100, 7, torch.float32, 1e-4, None, 256),
    ("c2_c64", 32, 40, 100, 7, torch.float32, 1e-4, None, 64),  # BASELINE config 2's "C=64"
    ("c4", 128, 80, 200, 9, torch.bfloat16, 1e-2, [0, 6, 7, 63, 64, 127], 256),
    ("c5", 32, 160, 400, 7, torch.bfloat16, 1e-2, [0, 15, 31], 256),
    # C5 at stride 4: 320x800 maps, K 9 - one image's per-tap products are 295 MB, so
    # every Y chunk is a single image
    ("c5_s4", 4, 320, 800, 9, torch.bfloat16, 1e-2, [0, 3], 256),
    # the per-GPU shares of the sharded configs (BASELINE configs 4 and 5)
    ("c4_share2", 512, 80, 200, 9, torch.bfloat16, 1e-2, [0, 255, 256, 511], 256),
    ("c4_share4", 256, 80, 200, 9, torch.bfloat16, 1e-2, [0, 128, 255], 256),
    ("c5_s4_share8", 32, 320, 800, 9, torch.bfloat16, 1e-2, [0, 31], 256),
])
def test_full_size_against_oracle_and_permutation(name, B, H, W, K, dtype, tol, sample, Cin):
    xt, pt, wt, pts = inputs(B, H, W, K, dtype, Cin=Cin, nseed=4 if H * W > 100000 else 8)
    out = lfa(xt, pt, wt, K)
    idx = list(range(B)) if sample is None else sample
    ref, mag = oracle_images(xt, pts, wt, K, idx)
    got = out[idx].float().cpu().numpy()
    assert rel_err(got, ref) <= tol, name
    assert term_err(got, ref, mag) <= tol, name
    # images are independent: a permuted batch gives the permuted result, bit for bit
    perm = torch.from_numpy(np.random.default_rng(3).permutation(B)).cuda()
    outp = lfa(xt[perm].contiguous(memory_format=torch.channels_last), pt[perm].contiguous(), wt, K)
    assert torch.equal(outp, out[perm]), name


def test_full_size_linearity_fp32():
    """relu off: LFA(a*x1 + b*x2) = a*LFA(x1) + b*LFA(x2) at C2's size (float32-class
    arithmetic on the split GEMM: 1e-4 of the scale)."""
    B, H, W, K = 32, 40, 100, 7
    x1, pt, wt, _ = inputs(B, H, W, K, torch.float32, seed=0)
    x2, _, _, _ = inputs(B, H, W, K, torch.float32, seed=50)

    def run(x):
        o = dcn.deform_conv_forward(x, pt, wt, (1, 1), (0, 3), (1, 1), offset_is_points=True)
        torch.cuda.synchronize()
        return o.double()

    a, b = 0.75, -1.5
    lhs = run((a * x1 + b * x2).contiguous(memory_format=torch.channels_last))
    rhs = a * run(x1) + b * run(x2)
    err = (lhs - rhs).abs().max().item() / rhs.abs().max().item()
    assert err <= 1e-4, err
