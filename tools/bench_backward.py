#!/usr/bin/env python3
"""Times the deform-conv backward at the LFA's shapes (fp32, training): images/s for
grad_input+grad_offset and grad_weight, beside the forward.  Not the headline metric;
tools/ only.  Usage: python tools/bench_backward.py [--batch 32] [--steps 20]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ganet_amd import dcn, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--algo", type=int, default=0, help="0 automatic, 1 direct, 2 commuted")
    ap.add_argument("--hw", type=int, nargs=2, default=[40, 100], help="feature map H W")
    ap.add_argument("--k", type=int, default=7, help="taps of the 1xK kernel")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, Cin, H, W, K, Cout = args.batch, 256, args.hw[0], args.hw[1], args.k, 64
    x, off, w, _ = synth.lfa_inputs(B, Cin, H, W, K, Cout=Cout, seed=0)
    tx, toff, tw = (torch.from_numpy(a).to(dev) for a in (x, off, w))
    if args.channels_last:
        tx = tx.contiguous(memory_format=torch.channels_last)
    out = dcn.deform_conv_forward(tx, toff, tw, (1, 1), (0, (K - 1) // 2), (1, 1))
    go = torch.randn_like(out)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps

    t_f = timed(lambda: dcn.deform_conv_forward(tx, toff, tw, (1, 1), (0, (K - 1) // 2), (1, 1)))
    t_i = timed(lambda: dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), (0, (K - 1) // 2), (1, 1),
                                                 need_input=True, need_weight=False,
                                                 algo=args.algo))
    t_w = timed(lambda: dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), (0, (K - 1) // 2), (1, 1),
                                                 need_input=False, need_weight=True,
                                                 algo=args.algo))
    t_b = timed(lambda: dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), (0, (K - 1) // 2), (1, 1),
                                                 need_input=True, need_weight=True,
                                                 algo=args.algo))
    # the forward keeping its per-tap products, the backward reading them
    # (GANET_DCN_ALGO_KEEP_Y / ganet_deform_conv_backward_y)
    keep = {}
    t_fk = timed(lambda: dcn.deform_conv_forward(tx, toff, tw, (1, 1), (0, (K - 1) // 2), (1, 1),
                                                 keep_y=keep))
    t_bk = None
    if "y" in keep:
        t_bk = timed(lambda: dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), (0, (K - 1) // 2),
                                                      (1, 1), need_input=True, need_weight=True,
                                                      algo=args.algo, y_saved=keep["y"]))
    print(json.dumps(dict(batch=B, hw=[H, W], k=K, channels_last=args.channels_last, algo=args.algo,
                          forward_ms=t_f, backward_input_ms=t_i, backward_weight_ms=t_w,
                          backward_both_ms=t_b,
                          train_images_per_s=B / ((t_f + t_b) * 1e-3),
                          forward_keep_y_ms=t_fk if "y" in keep else None,
                          backward_both_with_y_ms=t_bk,
                          train_images_per_s_with_y=(B / ((t_fk + t_bk) * 1e-3)) if t_bk else None)))


if __name__ == "__main__":
    main()
