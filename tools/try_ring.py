"""Development check of algorithm 4 (csrc/deform_conv_ring.hip: Y kept in the forming XCD's L2,
tap-major, static schedule) against algorithm 2 and the oracle on the GPU, with event timings.

  python tools/try_ring.py [B] [iters]        C3 shape (40x100, Cin 256, K 7, bf16), B images
  SMALL=1 python tools/try_ring.py            a few small shapes against the oracle first
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

from ganet_amd import _lib, dcn, synth


def run(xt, pt, wt, K, algo, iters=0):
    L = _lib.lib()
    B, Cin, H, W = xt.shape
    p = _lib.DcnParams()
    p.B, p.Cin, p.H, p.W, p.Cout = B, Cin, H, W, wt.shape[0]
    p.kH, p.kW, p.sH, p.sW, p.pH, p.pW, p.dH, p.dW = 1, K, 1, 1, 0, (K - 1) // 2, 1, 1
    p.groups = p.deformable_groups = 1
    p.dtype = dcn._DTYPES[xt.dtype]
    p.x_layout = p.out_layout = _lib.GANET_NHWC
    p.fuse_relu, p.offset_is_points, p.algo = 1, 1, algo
    n = L.ganet_deform_conv_workspace_bytes(ctypes.byref(p))
    assert n > 0, "unsupported"
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    out = torch.empty((B, wt.shape[0], H, W), dtype=xt.dtype, device="cuda",
                      memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = L.ganet_deform_conv_forward(xt.data_ptr(), pt.data_ptr(), None, wt.data_ptr(), None,
                                         out.data_ptr(), ctypes.byref(p), ws.data_ptr(), n, st)
        _lib.check(rc, "forward")
    call()
    done = torch.cuda.Event()
    done.record()
    t0 = time.time()
    while not done.query():
        if time.time() - t0 > 20:
            print("HUNG algo", algo, flush=True)
            os._exit(3)
        time.sleep(0.02)
    torch.cuda.synchronize()
    so = L.ganet_deform_conv_status_offset(ctypes.byref(p))
    ctl = None
    if so != ctypes.c_size_t(-1).value:
        ctl = ws[so:so + 4096].view(torch.int32).cpu().numpy()
    ms = None
    if iters:
        for _ in range(10):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        if so != ctypes.c_size_t(-1).value:
            ctl = ws[so:so + 4096].view(torch.int32).cpu().numpy()
    return out, ctl, ms


def describe(ctl):
    return "status=%d finished=%d images=%d join=%s bar=%s" % (
        ctl[0], ctl[32], ctl[64], ctl[96:96 + 8 * 32:32].tolist(),
        ctl[96 + 8 * 32:96 + 24 * 32:32].tolist())


def small():
    from oracle import cpu_oracle as O
    ok = True
    for (B, Cin, H, W, K) in [(3, 256, 13, 37, 7), (40, 256, 40, 100, 7), (17, 256, 20, 50, 9),
                              (11, 128, 17, 23, 7), (1, 256, 64, 64, 7), (130, 256, 40, 100, 7)]:
        x, off, w, pts = synth.lfa_inputs(B, Cin, H, W, K, seed=3)
        xt = torch.from_numpy(x).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        pt = torch.from_numpy(pts).cuda()
        wt = torch.from_numpy(w).cuda().bfloat16()
        o2, _, _ = run(xt, pt, wt, K, 2)
        o4, ctl, _ = run(xt, pt, wt, K, 4)
        o4b, ctl2, _ = run(xt, pt, wt, K, 4)
        xr = xt.float().cpu().numpy()
        wr = wt.float().cpu().numpy()
        ref = O.deform_conv_forward(xr, off, wr, (1, 1), (0, (K - 1) // 2), (1, 1), nthreads=16)
        ref = np.maximum(ref, 0)
        e4 = np.abs(o4.float().cpu().numpy() - ref).max() / np.abs(ref).max()
        e2 = np.abs(o2.float().cpu().numpy() - ref).max() / np.abs(ref).max()
        d = (o2.float() - o4.float()).abs().max().item()
        nd = (o2 != o4).sum().item()
        rep = torch.equal(o4, o4b)
        print((B, Cin, H, W, K), "err4 %.2e err2 %.2e  |a2-a4| %.3e  differing %d of %d  repeat-identical %s"
              % (e4, e2, d, nd, o2.numel(), rep), describe(ctl), flush=True)
        ok = ok and e4 <= 1e-2 and ctl[0] == 0 and ctl2[0] == 0 and rep
    return ok


def main():
    if os.environ.get("SMALL"):
        ok = small()
        print("SMALL", "ok" if ok else "FAILED", flush=True)
        if not ok:
            sys.exit(1)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    H, W, K = 40, 100, 7
    x, off, w, pts = synth.lfa_inputs(8, 256, H, W, K, seed=0)
    reps = (B + 7) // 8
    xt = torch.from_numpy(x).cuda().repeat(reps, 1, 1, 1)[:B]
    xt = (xt * (1 + 0.01 * torch.arange(B, device="cuda")[:, None, None, None])).bfloat16()
    xt = xt.contiguous(memory_format=torch.channels_last)
    pt = torch.from_numpy(pts).cuda().repeat(reps, 1, 1, 1)[:B].contiguous()
    wt = torch.from_numpy(w).cuda().bfloat16()
    if os.environ.get("ONLY4"):
        o4, ctl, t4 = run(xt, pt, wt, K, 4, iters)
        print("algo 4 only: %.4f ms" % t4, describe(ctl), flush=True)
        return
    o2, _, t2 = run(xt, pt, wt, K, 2, iters)
    print("algo 2: %.4f ms" % t2, flush=True)
    o4, ctl, t4 = run(xt, pt, wt, K, 4, iters)
    print("algo 4: %.4f ms" % t4, describe(ctl), flush=True)
    d = (o2.float() - o4.float()).abs().max().item()
    nd = (o2 != o4).sum().item()
    print("identical:", torch.equal(o2, o4), "max diff", d, "differing", nd, "of", o2.numel(),
          "scale", o2.float().abs().max().item())
    sys.exit(0 if ctl[0] == 0 and d <= 0.02 * o2.float().abs().max().item() else 1)


if __name__ == "__main__":
    main()
