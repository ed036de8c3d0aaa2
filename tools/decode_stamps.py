"""Phase times inside decode_group_kernel<0> (image 0), from the stamp build of the library:
  GANET_DECODE_DEFS=-DGANET_DECODE_STAMPS python -m ganet_amd.build        # prints the variant's path
  GANET_HIP_LIB=<that path> PYTHONPATH=. python tools/decode_stamps.py <shape key> <batch>
The constant 100 MHz counter is read by thread 0 of workgroup 0 at the phase boundaries."""
import ctypes
import sys

import numpy as np
import torch

from ganet_amd import _lib, synth
from ganet_amd.pipeline import HotPath

NAMES = ["heat -> LDS", "NMS + bitmap", "compaction", "gathers + lists", "centre sort", "centre clustering",
         "record words (atomic)", "group means", "keypoint -> group", "counts, header, finish"]


def main():
    key, B = sys.argv[1], int(sys.argv[2])
    cfg = synth.SHAPES[key]
    H, W = cfg["H"], cfg["W"]
    dev = torch.device("cuda", 0)
    heat, off, err = (torch.from_numpy(a).to(dev) for a in
                      synth.decode_inputs(B, H, W, lanes=cfg["lanes"], seed=0))
    T = float(cfg["T"])
    hp = HotPath(B, H, W, Cin=64, Cout=64, K=cfg["K"], dtype=torch.bfloat16, device=dev,
                 thr=cfg["thr"], root_thr=cfg["root_thr"], cluster_thr=T, by_center_thr=T,
                 max_kpts=min(H * W, 8192 if H * W > 64000 else 1024), max_centers=256,
                 side_decode=False)
    L = _lib.lib()
    acc = np.zeros(10)
    n = 50
    for i in range(n + 10):
        hp.decode(heat, off, err)
        torch.cuda.synchronize()
        st = (ctypes.c_ulonglong * 16)()
        assert L.ganet_debug_decode_stamps(st) == 0
        t = np.array(st[:11], dtype=np.float64)
        if i >= 10:
            acc += np.diff(t) / 100.0      # us
    acc /= n
    if H * W > 8192:
        # large images: the lists come from the grid-wide kernels, the group kernel (MODE 2) only
        # sorts and clusters the centres - its one meaningful row
        r = hp.result()
        print("%s B=%d: serial centre clustering in decode_group_kernel<2>, image 0: %.2f us "
              "(%d centres, %d groups; batch mean %.0f centres)"
              % (key, B, acc[5], int(r.counts[0, 1]), int(r.counts[0, 2]), r.counts[:, 1].mean()))
        return
    print("%s B=%d: phases of workgroup 0 (us)" % (key, B))
    for name, v in zip(NAMES, acc):
        print("  %-26s %6.2f" % (name, v))
    print("  %-26s %6.2f" % ("total", acc.sum()))


if __name__ == "__main__":
    main()
