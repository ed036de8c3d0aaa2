"""Decode + grouping + pack alone (no LFA beside it): event time per batch.
usage: python tools/time_decode.py <shape key of synth.SHAPES> <batch> [iters]"""
import sys

import torch

from ganet_amd import synth
from ganet_amd.pipeline import HotPath


def main():
    key, B = sys.argv[1], int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    cfg = synth.SHAPES[key]
    H, W = cfg["H"], cfg["W"]
    dev = torch.device("cuda", 0)
    heat, off, err = (torch.from_numpy(a).to(dev) for a in
                      synth.decode_inputs(B, H, W, lanes=cfg["lanes"], seed=0))
    T = float(cfg["T"])
    hp = HotPath(B, H, W, Cin=64, Cout=64, K=cfg["K"], dtype=torch.bfloat16, device=dev,
                 thr=cfg["thr"], root_thr=cfg["root_thr"], cluster_thr=T, by_center_thr=T,
                 max_kpts=min(H * W, 8192 if H * W > 64000 else (4096 if cfg["lanes"] > 4 else 1024)),
                 max_centers=256, side_decode=False)
    for _ in range(10):
        hp.decode(heat, off, err)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hp.decode(heat, off, err)
    e1.record()
    torch.cuda.synchronize()
    r = hp.result()
    print("%s B=%d: %.4f ms per batch; kpts/img %.0f centres/img %.0f groups/img %.1f overflow %d"
          % (key, B, e0.elapsed_time(e1) / iters, r.counts[:, 0].mean(), r.counts[:, 1].mean(),
             r.counts[:, 2].mean(), int(r.counts[:, 3].max())))


if __name__ == "__main__":
    main()
