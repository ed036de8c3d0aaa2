"""Two host threads call the library at once on streams of their own (separate tensors and
workspaces): the C-ABI keeps no per-call state outside its arguments, so every result must
equal the one a single thread gets.  (The one shared thing, the static device array of
ganet_decode_batch_noalloc, is documented as racing and is not exercised here.)"""
import threading

import numpy as np
import pytest
import torch

from ganet_amd import dcn, synth
from ganet_amd import decode as D

pytestmark = pytest.mark.gpu


def test_concurrent_calls_from_two_threads():
    dev = torch.device("cuda", 0)
    cfg = synth.SHAPES["s8_culane"]
    H, W, K = cfg["H"], cfg["W"], cfg["K"]
    jobs = []
    for seed, (B, dtype) in enumerate([(24, torch.bfloat16), (9, torch.float32)]):
        x, off, w, pts = synth.lfa_inputs(B, 256, H, W, K, seed=seed)
        heat, o, e = synth.decode_inputs(B, H, W, lanes=cfg["lanes"], seed=seed)
        jobs.append(dict(
            x=torch.from_numpy(x).to(dev).to(dtype).contiguous(memory_format=torch.channels_last),
            pts=torch.from_numpy(pts).to(dev), w=torch.from_numpy(w).to(dev).to(dtype),
            heat=torch.from_numpy(heat).to(dev), o=torch.from_numpy(o).to(dev),
            e=torch.from_numpy(e).to(dev)))

    def work(j, reps, out):
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for _ in range(reps):
                y = dcn.deform_conv_forward(j["x"], j["pts"], j["w"], (1, 1), (0, (K - 1) // 2), (1, 1),
                                            fuse_relu=True, offset_is_points=True)
                r = D.decode_batch(j["heat"], j["o"], j["e"], thr=cfg["thr"], root_thr=cfg["root_thr"],
                                   cluster_thr=float(cfg["T"]), cluster_by_center_thr=float(cfg["T"]))
            stream.synchronize()
        out.append((y, r))

    torch.cuda.synchronize()
    alone = []
    for j in jobs:
        got = []
        work(j, 1, got)
        alone.append(got[0])
    together = [[], []]
    errors = []

    def guarded(j, out):
        try:
            work(j, 25, out)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=guarded, args=(jobs[i], together[i])) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (y0, r0), out in zip(alone, together):
        y1, r1 = out[0]
        assert torch.equal(y0, y1)
        assert np.array_equal(r0.header, r1.header)
        assert np.array_equal(r0.records, r1.records)
