"""Probe: does the LFA pair (GEMM = write-heavy, sampling = read-heavy) gain from two half-batches
in flight on two streams, plain or confined to disjoint halves of the CUs
(hipExtStreamCreateWithCUMask)?  Prints ms per 128 images for each arrangement."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from ganet_amd import synth  # noqa: E402
from ganet_amd.pipeline import HotPath  # noqa: E402

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
B, H, W, K = 128, 40, 100, 7


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def make(nb, seed):
    x, off, w, pts = synth.lfa_inputs(min(nb, 8), 256, H, W, K, seed=seed)
    reps = nb // x.shape[0]
    xt = torch.from_numpy(x).to(dev).repeat(reps, 1, 1, 1).bfloat16().contiguous(
        memory_format=torch.channels_last)
    pt = torch.from_numpy(pts).to(dev).repeat(reps, 1, 1, 1).contiguous()
    hp = HotPath(nb, H, W, K=K, dtype=torch.bfloat16, device=dev)
    hp.set_weight(torch.from_numpy(w).to(dev))
    return hp, xt, pt


def run(name, jobs, streams, iters=200):
    def once():
        for (hp, x, p), s in zip(jobs, streams):
            with torch.cuda.stream(s):
                hp.lfa(x, p)
    for _ in range(20):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        once()
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    b.record()
    torch.cuda.synchronize()
    print("%-46s %.4f ms per %d images" % (name, a.elapsed_time(b) / iters, B), flush=True)


full = make(B, 0)
halves = [make(B // 2, 0), make(B // 2, 100)]
quarters = [make(B // 4, 200 * i) for i in range(4)]
cur = torch.cuda.current_stream()
run("one stream, 128 images", [full], [cur])
run("one stream, 2 x 64 images", halves, [cur, cur])
s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
run("two plain streams, 64 + 64", halves, s2)
s4 = [torch.cuda.Stream(device=dev) for _ in range(4)]
run("four plain streams, 4 x 32", quarters, s4)
alt = [masked_stream([0x55555555] * 8), masked_stream([0xAAAAAAAA] * 8)]
run("two masked streams, alternating CUs", halves, alt)
half = [masked_stream([0xFFFFFFFF] * 4 + [0] * 4), masked_stream([0] * 4 + [0xFFFFFFFF] * 4)]
run("two masked streams, low / high 128 bits", halves, half)
alt16 = [masked_stream([0x0000FFFF] * 8), masked_stream([0xFFFF0000] * 8)]
run("two masked streams, alternating 16-CU groups", halves, alt16)
run("one stream, 128 images (again)", [full], [cur])
