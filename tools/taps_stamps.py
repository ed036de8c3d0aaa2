#!/usr/bin/env python3
"""Phase stamps of algorithm 5's kernel (dcn_taps_kernel) on the stamp build:
  GANET_TAPS_DEFS=-DGANET_TAPS_STAMPS python -m ganet_amd.build        (prints the library's path)
  GANET_HIP_LIB=<that path> python tools/taps_stamps.py [bf16|fp32] [batch]
Prints, for the first and last workgroup's first and last wave, microseconds since the earliest
of the four starts: start, coordinates in the table, gathers issued, chunk 0 contracted (one
stamp per four chunks), block parked, barrier passed, outputs stored."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ganet_amd import dcn, synth  # noqa: E402

dt = torch.float32 if (sys.argv[1:] or ["bf16"])[0] == "fp32" else torch.bfloat16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x, off, w, _ = synth.lfa_inputs(B, 256, 40, 100, 7, Cout=64, seed=0)
tx = torch.from_numpy(x).cuda().to(dt).contiguous(memory_format=torch.channels_last)
toff, tw = torch.from_numpy(off).cuda(), torch.from_numpy(w).cuda().to(dt)
from ganet_amd import _lib  # noqa: E402

L = ctypes.CDLL(os.environ["GANET_HIP_LIB"])
for k in range(5):
    dcn.deform_conv_forward(tx, toff, tw, (1, 1), (0, 3), (1, 1), algo=5)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 48)()
    assert L.ganet_debug_taps_stamps(st) == 0
    t0 = min(st[12 * j] for j in range(4))
    print("launch", k)
    for j, name in enumerate(("first wg, wave 0", "first wg, last wave", "last wg, wave 0", "last wg, last wave")):
        row = [st[12 * j + i] for i in range(12)]
        print("  %-20s" % name, " ".join("%6.2f" % ((v - t0) / 100.0) for v in row if v >= t0))
