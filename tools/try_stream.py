"""Development check of algorithm 3 (one persistent producer/consumer launch) against
algorithm 2 on the GPU: outputs must be bit-identical (same arithmetic), and the control
block's error word must stay 0.  Prints event timings of both."""
import ctypes
import sys

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
    side = torch.cuda.Stream()
    import time
    t0 = time.time()
    while not done.query():
        if time.time() - t0 > 6:
            with torch.cuda.stream(side):
                c = ws[:4 * 352].view(torch.int32).to("cpu", non_blocking=True)
                side.synchronize()
            print("HUNG algo", algo, "ctl", c[0].item(), c[32].item(), c[64:72].tolist(), c[96::32].tolist(), flush=True)
            import os
            os._exit(3)
        time.sleep(0.05)
    torch.cuda.synchronize()
    ctl = ws[:4 * 352].view(torch.int32).cpu().numpy()
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
    return out, ctl, ms


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    H, W, K = 40, 100, 7
    x, off, w, pts = synth.lfa_inputs(8, 256, H, W, K, seed=0)
    reps = (B + 7) // 8
    xt = torch.from_numpy(x).cuda().repeat(reps, 1, 1, 1)[:B]
    xt = (xt * (1 + 0.01 * torch.arange(B, device="cuda")[:, None, None, None])).bfloat16()
    xt = xt.contiguous(memory_format=torch.channels_last)
    pt = torch.from_numpy(pts).cuda().repeat(reps, 1, 1, 1)[:B].contiguous()
    wt = torch.from_numpy(w).cuda().bfloat16()
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import os
    if os.environ.get("ONLY2"):
        o2, _, t2 = run(xt, pt, wt, K, 2, iters)
        print("algo 2 only: %.4f ms" % t2, flush=True)
        return
    if os.environ.get("ONLY3"):
        o3, ctl, t3 = run(xt, pt, wt, K, 3, iters)
        print("algo 3 only: %.4f ms err=%d" % (t3, int(ctl[0])), flush=True)
        if os.environ.get("STAMPS"):
            st = ctl[8:24].view(np.uint64).astype(np.float64)
            names = ["control", "x load", "mfma+store issue", "drain+count", "corners", "poll", "sample", "blocked"]
            tot = st.sum()
            print("  per-wave phase time (us at 2.1 GHz), share:")
            for n, v in zip(names, st):
                print("   %-18s %8.1f  %5.1f%%" % (n, v / 4096 / 2100.0, 100 * v / max(tot, 1)))
        return
    o2, _, t2 = run(xt, pt, wt, K, 2, iters)
    print("algo 2: %.4f ms" % t2, flush=True)
    o3, ctl, t3 = run(xt, pt, wt, K, 3, iters)
    print("algo 3: %.4f ms  err=%d owner=%s joined=%s" % (t3, int(ctl[0]), ctl[32:40].tolist(), ctl[64:72].tolist()), flush=True)
    same = torch.equal(o2, o3)
    d = (o2.float() - o3.float()).abs().max().item()
    print("identical:", same, "max diff", d, "scale", o2.float().abs().max().item())
    sys.exit(0 if same and ctl[0] == 0 else 1)


if __name__ == "__main__":
    main()
