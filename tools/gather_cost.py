"""CPU cost of sharding.NodeGather on rank 0 for N ranks of one node (no GPU involved): every
other rank offers a block of C3's size (128 images, 209 KB of CULane text) every `period` ms,
rank 0 gathers `steps` steps; prints rank 0's CPU time and wall time per step.
  python tools/gather_cost.py [ranks=8] [steps=400] [period_ms=0.3] [copy=1]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganet_amd.sharding import NodeGather  # noqa: E402


def work(rank, world, steps, period, q):
    g = NodeGather(rank, world, "cost%d" % os.getppid())
    text = bytes(209000)
    lengths = np.full(128, 209000 // 128, np.int32)
    lengths[-1] += 209000 - int(lengths.sum())
    buf = bytearray(16 << 20)
    for _ in range(20):
        g.gather(text, lengths, tag=rank, into=buf)
    c0, t0 = time.thread_time(), time.perf_counter()
    nxt = t0
    for _ in range(steps):
        if rank:
            nxt += period
            while time.perf_counter() < nxt:
                pass
        g.gather(text, lengths, tag=rank, into=buf)
    q.put((rank, (time.thread_time() - c0) / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3))
    g.close()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    period = float(sys.argv[3]) * 1e-3 if len(sys.argv) > 3 else 0.3e-3
    q = mp.Queue()
    ps = [mp.Process(target=work, args=(r, world, steps, period, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    print("%d ranks, a block per rank every %.2f ms: rank 0 spends %.3f ms of CPU per step, %.3f ms wall"
          % (world, period * 1e3, res[0][1], res[0][2]))


if __name__ == "__main__":
    main()
