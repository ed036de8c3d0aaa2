#!/usr/bin/env python3
"""bench.py - images/s of GANet's hot path (LFA + keypoint decode) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3] [--dtype bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without a launcher (WORLD_SIZE unset) and --gpus N > 1 this process starts N rank processes
itself, one per GPU, as the reference's tools/dist_test.sh:13-24 does with
torch.distributed.launch; it never touches the GPU, forwards rank 0's JSON line and exits
with the worst rank's code.

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: the LFA (relu(DeformConv1D(x, pts - base)), one C-ABI call)
followed by decode + lane grouping of the same batch and the async D2H copy of
the compact lane lists.  Images are independent, so on N GPUs each rank owns
its own batch of the same size (weak scaling) and NO collective is on the data
path; torch.distributed is used only for the barrier / max-over-ranks timing
the driver's contract asks for.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "fp32": 157.3}

WORKLOADS = {
    # name: shape key in ganet_amd.synth.SHAPES, per-GPU batch, Cin, default dtype, decode?
    "c1": dict(shape="s8_tusimple", B=1, Cin=256, dtype="fp32", decode=True,
               desc="C1 GANet-S TuSimple 320x800, batch 1: LFA (Cin256->64, K7, 40x100) + "
                    "keypoint decode/grouping, one image per step (latency case; the "
                    "reference's own speed test, tools/ganet/speed_test.py:57-65, is batch 1)"),
    "c3": dict(shape="s8_culane", B=128, Cin=256, dtype="bf16", decode=True,
               desc="C3 full head hot path: LFA (Cin256->64, K7) + NMS/decode/grouping, "
                    "CULane 320x800 (40x100 @ s8), batch 128 per GPU"),
    "c2": dict(shape="s8_tusimple", B=32, Cin=256, dtype="fp32", decode=False,
               desc="C2 LFA alone: Cin256->64, K7, 40x100, batch 32, fp32"),
    "c2_c64": dict(shape="s8_tusimple", B=32, Cin=64, dtype="fp32", decode=False,
                   desc="C2 variant: LFA alone with C=64 in/out, K7, 40x100, batch 32, fp32"),
    "c4": dict(shape="s4_culane", B=128, Cin=256, dtype="bf16", decode=True,
               desc="C4 GANet-L head shapes: LFA (Cin256->64, K9) + decode, 80x200 @ s4, "
                    "batch 128 per GPU (1024 over 8)"),
    "c5": dict(shape="dense_s8", B=32, Cin=256, dtype="bf16", decode=True,
               desc="C5 dense stress: 1280x3200 frames (160x400 @ s8), 16 lanes, thr 0.05, "
                    "batch 32 per GPU (256 over 8)"),
    "c5_s4": dict(shape="dense_s4", B=32, Cin=256, dtype="bf16", decode=True,
                  desc="C5 dense stress at stride 4: 1280x3200 frames (320x800 @ s4), K9, 16 lanes, "
                       "thr 0.05, batch 32 per GPU (256 over 8); one image per Y chunk"),
}
TORCH_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}
ESIZE = {"bf16": 2, "f16": 2, "fp32": 4}


def lfa_algorithmic(B, Cin, Cout, H, W, K, dtype):
    """Algorithmic HBM bytes and MFMA FLOPs of one LFA launch:
    x + offsets(f32) + out once per image, weights once per launch."""
    e = ESIZE[dtype]
    hw = H * W
    per_image = e * Cin * hw + 4 * 2 * K * hw + e * Cout * hw
    nbytes = B * per_image + e * Cout * Cin * K
    flops = B * 2.0 * Cout * Cin * K * hw
    return nbytes, flops


def decode_algorithmic(B, H, W):
    """5 float32 planes read once per image."""
    return B * 5 * 4 * H * W


def build_inputs(wl, dtype, dev, seed):
    from ganet_amd import synth
    cfg = synth.SHAPES[wl["shape"]]
    B, H, W, K, Cin = wl["B"], cfg["H"], cfg["W"], cfg["K"], wl["Cin"]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn((B, Cin, H, W), generator=g, device=dev, dtype=torch.float32)
    x = x.to(TORCH_DTYPE[dtype]).contiguous(memory_format=torch.channels_last)
    pts, _ = synth.lfa_points(B, H, W, K, seed=seed)
    w = synth.lfa_weight(Cin, K, 64, seed=seed)
    d = dict(x=x, pts=torch.from_numpy(pts).to(dev), w=torch.from_numpy(w).to(dev),
             cfg=cfg, pts_host=pts, w_host=w)
    if wl["decode"]:
        heat, off, err = synth.decode_inputs(B, H, W, lanes=cfg["lanes"], seed=seed)
        d.update(heat=torch.from_numpy(heat).to(dev), off=torch.from_numpy(off).to(dev),
                 err=torch.from_numpy(err).to(dev), heat_host=heat, off_host=off,
                 err_host=err)
    return d


def cpu_baseline_and_parity(wl, dtype, d, hp, n_images, threads):
    """Rank 0, N=1: the CPU oracle ("port" of the reference's algorithm) timed on
    a bounded sample of the SAME inputs, and used as the checker of the GPU
    result of that sample.  Never part of the measured GPU path."""
    from oracle import cpu_oracle as O
    cfg = d["cfg"]
    K = cfg["K"]
    n = min(n_images, wl["B"])
    x_host = d["x"][:n].float().cpu().numpy()            # what the kernel read
    w_host = d["w"].to(TORCH_DTYPE[dtype]).float().cpu().numpy()
    base = np.zeros((2 * K, 1, 1), np.float32)
    base[1::2, 0, 0] = np.arange(-(K - 1) // 2, (K - 1) // 2 + 1)
    off_host = d["pts_host"][:n] - base[None]
    t0 = time.perf_counter()
    ref = O.deform_conv_forward(x_host, off_host, w_host, (1, 1), (0, (K - 1) // 2),
                                (1, 1), acc64=False, nthreads=threads)
    ref = np.maximum(ref, 0)
    t_lfa = time.perf_counter() - t0
    t_dec = 0.0
    parity = {}
    got = hp.feature[:n].float().cpu().numpy()
    parity["lfa_rel_err"] = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
    parity["lfa_tolerance"] = 1e-4 if dtype == "fp32" else 1e-2
    parity["lfa_ok"] = bool(parity["lfa_rel_err"] <= parity["lfa_tolerance"])
    # the stricter, element-wise yardstick: every output's error against the sum of the
    # magnitudes of ITS OWN terms, sum_t sum_c |w| * blend(|x|) (the blend weights are >= 0, so
    # the same oracle on |x|, |w| gives it), on a part of the sample
    ns = min(n, 32)
    mag = O.deform_conv_forward(np.abs(x_host[:ns]), off_host[:ns], np.abs(w_host), (1, 1),
                                (0, (K - 1) // 2), (1, 1), acc64=False, nthreads=threads)
    parity["lfa_rel_err_vs_terms"] = float(
        (np.abs(got[:ns] - ref[:ns]) / np.maximum(mag, 1e-30)).max())
    parity["lfa_rel_err_vs_terms_images"] = ns
    parity["lfa_ok"] = bool(parity["lfa_ok"] and
                            parity["lfa_rel_err_vs_terms"] <= parity["lfa_tolerance"])
    if wl["decode"]:
        T = float(cfg["T"])
        t0 = time.perf_counter()
        O.decode_group_batch(d["heat_host"][:n], d["off_host"][:n], d["err_host"][:n],
                             cfg["thr"], cfg["root_thr"], T, T, nthreads=threads)
        t_dec = time.perf_counter() - t0
        res = hp.result()
        ok = True
        for b in range(min(n, 8)):
            c, kxy, al, rt = O.decode(d["heat_host"][b], d["off_host"][b],
                                      d["err_host"][b], cfg["thr"], cfg["root_thr"])
            gid, means, cgid, ng = O.group_fast(c, rt, T, T)
            img = res.image(b)
            ok &= bool(np.array_equal(img["centers"], c)
                       and img["align"].tobytes() == al.tobytes()
                       and img["root"].tobytes() == rt.tobytes()
                       and np.array_equal(img["kpt_gid"], gid)
                       and np.array_equal(img["group_center"], means))
        parity["decode_bit_exact"] = ok
    total = t_lfa + t_dec
    # the same port on ONE thread, on the first images of the sample
    n1 = min(n, 8)
    t0 = time.perf_counter()
    O.deform_conv_forward(x_host[:n1], off_host[:n1], w_host, (1, 1), (0, (K - 1) // 2), (1, 1),
                          acc64=False, nthreads=1)
    if wl["decode"]:
        O.decode_group_batch(d["heat_host"][:n1], d["off_host"][:n1], d["err_host"][:n1],
                             cfg["thr"], cfg["root_thr"], float(cfg["T"]), float(cfg["T"]),
                             nthreads=1)
    t_single = time.perf_counter() - t0
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=n / total, unit="images/s", cores=threads, kind="port",
                single_thread=dict(value=n1 / t_single, unit="images/s", images=n1),
                nproc=os.cpu_count(), cpu_share=len(os.sched_getaffinity(0)), cpu_model=model,
                sample="%d of the %d images of the timed batch (same seeded inputs): "
                       "oracle LFA fp32 %.2fs + decode/grouping %.3fs, OpenMP over images"
                       % (n, wl["B"], t_lfa, t_dec),
                lfa_images_per_s=n / t_lfa,
                decode_images_per_s=(n / t_dec) if t_dec > 0 else None), parity


def lfa_backward_side_measurement(dev, Cin, H, W, K, B=32, steps=20):
    """SURVEY 8 row f4, outside the timed region and not part of `value`: all gradients of
    the LFA's deform-conv (float32, channel-last, 32 images) through
    ganet_deform_conv_backward.  Never lets a failure touch the bench line."""
    try:
        import torch
        from ganet_amd import dcn, synth
        x, off, w, _ = synth.lfa_inputs(B, Cin, H, W, K, Cout=64, seed=7)
        tx, toff, tw = (torch.from_numpy(a).to(dev) for a in (x, off, w))
        tx = tx.contiguous(memory_format=torch.channels_last)
        pad = (0, (K - 1) // 2)
        go = torch.randn_like(dcn.deform_conv_forward(tx, toff, tw, (1, 1), pad, (1, 1)))

        def run():
            dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), pad, (1, 1))

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        res = {"what": "grad input+offset+weight, float32, channel-last, batch %d, %dx%d, K %d" % (B, H, W, K),
               "ms": ms, "images_per_s": B / (ms * 1e-3), "steps": steps}
        # the same with the forward's per-tap products kept (GANET_DCN_ALGO_KEEP_Y)
        keep = {}
        dcn.deform_conv_forward(tx, toff, tw, (1, 1), pad, (1, 1), keep_y=keep)
        if "y" in keep:
            def run_y():
                dcn.deform_conv_backward(tx, toff, tw, go, (1, 1), pad, (1, 1), y_saved=keep["y"])
            for _ in range(3):
                run_y()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                run_y()
            e1.record()
            torch.cuda.synchronize()
            res["ms_with_forward_products"] = e0.elapsed_time(e1) / steps
        return res
    except Exception as exc:  # noqa: BLE001 - a side measurement must not break the line
        return {"error": repr(exc)}


def spawn_ranks(n, argv):
    """One child process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in its environment),
    started from a parent that has not initialised the GPU; nothing re-execs.  Rank 0's
    stdout is this process's stdout.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others in a barrier: once one has failed, the rest (the
    # exact processes started above) get a grace period and are then ended
    worst, failed_at, live = 0, None, list(procs)
    while live:
        for pr in list(live):
            rc = pr.poll()
            if rc is None:
                continue
            live.remove(pr)
            if rc != 0 and worst == 0:
                worst = rc if rc > 0 else 1
                failed_at = time.monotonic()
        if failed_at is not None and live and time.monotonic() - failed_at > float(
                os.environ.get("GANET_BENCH_SPAWN_GRACE", "20")):
            for pr in live:
                pr.kill()
            failed_at = time.monotonic() + 1e9
        time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=300,
                    help="untimed steps before the timed region, warm-up included (clock ramp)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "fp32"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--cpu-images", type=int, default=128)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-backward", action="store_true",
                    help="skip the side measurement of the LFA backward (SURVEY 8 f4)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ablate", type=int, default=0, help="profiling only")
    ap.add_argument("--algo", type=int, default=0,
                    help="LFA algorithm: 0 auto, 1 sample-then-contract, 2 contract-then-sample, "
                         "3 / 4 that pair as one persistent launch (opt-in experiments)")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: do not gather the formatted lane lists on rank 0 (A/B)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: gather every k-th step's formatted lane lists on rank 0")
    ap.add_argument("--fp32-mode", default="auto", choices=["auto", "bf16", "exact"],
                    help="fp32 tensors: auto = float32-class accuracy (three bf16 MFMA passes "
                         "over hi/lo splits where the contract-then-sample path applies), "
                         "bf16 = one bf16 pass, exact = float32 MFMA only")
    ap.add_argument("--pts", default="synthetic", choices=["synthetic", "zero"])
    ap.add_argument("--device-pack", action="store_true",
                    help="pack in device memory and copy D2H (default: the pack kernel "
                         "writes the pinned host block directly)")
    ap.add_argument("--lfa-priority", action="store_true",
                    help="run the LFA on a high-priority stream (the decode beside it keeps the default "
                         "priority and fills what the LFA leaves idle)")
    ap.add_argument("--graph-fork", action="store_true",
                    help="with --graph: record the LFA and the decode as two branches of the graph "
                         "(measured slower at batch 1)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step in a HIP graph and replay it (small batches: "
                         "removes the host launch cost)")
    ap.add_argument("--single-stream", action="store_true",
                    help="LFA and decode on one stream (default: decode/grouping/pack run on "
                         "a stream of their own beside the LFA - they do not depend on it)")
    ap.add_argument("--no-consume", dest="consume", action="store_false",
                    help="stop the timed region at the packed records in pinned host memory "
                         "(default: consume them, as the reference's speed test includes "
                         "PostProcessor, tools/ganet/speed_test.py:57-65)")
    ap.add_argument("--consume", dest="consume", action="store_true", default=True,
                    help="inside the timed region, turn every step's records into the final "
                         "result format on the host (lane_post_process + adjust_result + "
                         "CULane .lines.txt text, C++ behind the C-ABI) on two host threads "
                         "beside the enqueueing one")
    ap.add_argument("--format-threads", type=int, default=0,
                    help="host threads of the formatting routines (0: CPU share, at most 16)")
    ap.add_argument("--result-blocks", type=int, default=2,
                    help="result blocks of the step, used in turn: with --consume, step i+n is "
                         "enqueued only after step i has been formatted")
    ap.add_argument("--consume-threads", type=int, default=2)
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the LFA / decode of every n-th timed step with HIP events: "
                         "each record is a packet on its stream, four of them per step cost "
                         "16-19 us of a 0.30 ms step (tools/ab_events.sh); the kernel times "
                         "are the mean over the bracketed steps of the timed region")
    ap.add_argument("--overlap-copy", action="store_true",
                    help="copy the lane lists D2H on a second stream")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("GANET_BENCH_SPAWN_ONLY"):
        # test hook (tests/test_bench_spawn.py): show what a rank was started with, touch no GPU
        print(json.dumps({k: os.environ.get(k) for k in
                          ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}),
              flush=True)
        if os.environ["GANET_BENCH_SPAWN_ONLY"] == "rank1_dies":   # the others "wait in a barrier"
            if os.environ.get("RANK") == "1":
                raise SystemExit(3)
            time.sleep(600)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # rehearsal knobs (one-GPU box): GANET_BENCH_DEVICE pins every rank to one device,
    # GANET_BENCH_BACKEND=gloo replaces RCCL for the barrier / max-over-ranks
    if os.environ.get("GANET_BENCH_DEVICE"):
        local_rank = int(os.environ["GANET_BENCH_DEVICE"])
    backend = os.environ.get("GANET_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if args.lfa_priority:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    dev = torch.device("cuda", local_rank)
    dist = None
    # GANET_BENCH_FORCE_DIST: rehearsal of the RCCL rendezvous / barrier / max with one rank
    if world > 1 or os.environ.get("GANET_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from ganet_amd import _lib
    from ganet_amd.pipeline import HotPath
    _lib.lib()

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    dtype = args.dtype or wl["dtype"]
    d = build_inputs(wl, dtype, dev, args.seed + 1000 * rank)
    cfg = d["cfg"]
    B, H, W, K, Cin = wl["B"], cfg["H"], cfg["W"], cfg["K"], wl["Cin"]
    T = float(cfg["T"])
    hp = HotPath(B, H, W, Cin=Cin, Cout=64, K=K, dtype=TORCH_DTYPE[dtype], device=dev,
                 thr=cfg["thr"], root_thr=cfg["root_thr"], cluster_thr=T, by_center_thr=T,
                 max_kpts=min(H * W, 8192 if H * W > 64000 else
                              (4096 if cfg["lanes"] > 4 else 1024)),
                 max_centers=256, overlap_copy=args.overlap_copy,
                 mfma_bf16={"auto": 0, "bf16": 1, "exact": 2}[args.fp32_mode],
                 side_decode=wl["decode"] and not (args.single_stream or args.graph
                                                   or args.overlap_copy),
                 ablate=args.ablate, algo=args.algo, pack_to_host=not args.device_pack,
                 result_blocks=args.result_blocks)
    if args.pts == "zero":
        d["pts"].zero_()
    hp.set_weight(d["w"])

    def one_step(ev=None):
        if hp.side is not None:
            # two streams: the LFA on the current one, decode beside it (HotPath.step)
            cur = torch.cuda.current_stream(dev)
            hp.side_wait()
            hp.inputs_ready.record(cur)
            if ev is not None:
                ev[0].record()
            hp.lfa(d["x"], d["pts"])
            if ev is not None:
                ev[1].record()
            with torch.cuda.stream(hp.side):
                hp.side.wait_event(hp.inputs_ready)
                if ev is not None:
                    ev[3].record()
                hp.decode(d["heat"], d["off"], d["err"])
                hp.side_mark()
                if ev is not None:
                    ev[2].record()
            return
        if ev is not None:
            ev[0].record()
        hp.lfa(d["x"], d["pts"])
        if ev is not None:
            ev[1].record()
            ev[3].record()
        if wl["decode"]:
            hp.decode(d["heat"], d["off"], d["err"])
            if ev is not None:
                ev[2].record()

    # Device spin-up, before the W warm-up steps and reported as `spinup_steps`: the clocks of
    # an idle MI355X ramp over the first ~100 steps of this workload (20 timed steps after 5
    # warm-up steps read 0.346 ms, 200 after 20 read 0.310, anything after 300 reads 0.307),
    # so a short run would time the ramp instead of the kernels.
    spinup = max(0, args.spinup - args.warmup)
    for _ in range(spinup):
        one_step()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        # the step allocates nothing and synchronises nothing, so it captures as is
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        if wl["decode"]:
            # --graph-fork: the LFA and the decode as two branches of the graph
            graph = hp.capture(d["x"], d["pts"], d["heat"], d["off"], d["err"],
                               fork=args.graph_fork)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one_step()
        graph.replay()
        torch.cuda.synchronize()
    if wl["decode"]:
        if hp.last is None:      # --spinup 0 --warmup 0: the capacity check needs one result
            one_step()
        over = int(np.bitwise_or.reduce(hp.result().counts[:, 3]))
        if over:
            raise SystemExit("decode capacity overflow in the bench configuration")

    # records -> final result format on the host: what the reference does per
    # image in Python after grouping (post_process.py:216-242, culane/test_dataset.py:79-148)
    fmt_threads = args.format_threads or min(16, len(os.sched_getaffinity(0)))
    fmt_crop, fmt_shape, fmt_thr = (0, 270, 1640, 590), (8 * H if cfg["ds"] == 8 else 4 * H,
                                                         8 * W if cfg["ds"] == 8 else 4 * W), 4

    import threading
    fmt_local = threading.local()     # every formatting thread keeps its own output buffers

    # N > 1: every (k-th) step's formatted lane lists are gathered on rank 0 in image order, as
    # north_star's path ends ("per-image lane lists gathered on the host"; the reference:
    # tools/ganet/tusimple/test_dataset.py:252-264, mmdet/apis/test.py:122-166).  A host thread
    # of its own owns a gloo group for it, so neither the enqueueing thread nor the GPUs'
    # streams wait for the gather; the timed region ends when the last gather has.
    gather_q, gather_thread, gather_group = None, None, None
    gathered = {"steps": 0, "images": 0, "bytes": 0, "lanes": 0, "error": None}
    if dist is not None and world > 1 and wl["decode"] and not args.no_gather and not args.graph:
        import queue
        from ganet_amd import sharding as _sh
        # one node (the driver's case): shared-memory rings; ranks on several nodes: a gloo group
        one_node = int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world
        if one_node and not os.environ.get("GANET_BENCH_GATHER_GLOO"):
            gather_group = _sh.NodeGather(rank, world, os.environ.get("MASTER_PORT", "0"))
        else:
            gather_group = dist.new_group(backend="gloo") if backend != "gloo" else None
        gather_q = queue.Queue()

    gather_buf = bytearray(16 << 20) if rank == 0 else None

    def gather_loop():
        from ganet_amd import sharding
        try:
            while True:
                item = gather_q.get()
                if item is None:
                    return
                if isinstance(gather_group, sharding.NodeGather):
                    # every rank sends its lane count beside its block; rank 0 fills one
                    # buffer that it keeps from step to step
                    res = gather_group.gather(item[0], item[1], tag=item[2], into=gather_buf)
                    lanes = sum(gather_group.last_tags)
                else:
                    res = sharding.gather_text_blocks(item[0], item[1].tolist(), dst=0, group=gather_group)
                    lanes = res[0].count(b"\n") if res is not None else 0   # one line per lane
                if res is not None:
                    gathered["steps"] += 1
                    gathered["images"] += len(res[1])
                    gathered["bytes"] += len(res[0])
                    gathered["lanes"] += lanes
                    if gathered["steps"] == 1 and isinstance(gather_group, sharding.NodeGather):
                        gathered["first_step_lanes_by_rank"] = list(gather_group.last_tags)
        except Exception as exc:  # noqa: BLE001
            gathered["error"] = repr(exc)

    def consume(slot, want_text=False):
        from ganet_amd import formats
        scratch = getattr(fmt_local, "scratch", None)
        if scratch is None:
            scratch = fmt_local.scratch = {}
        bl = formats.format_batch(hp.records(slot), B, True, cfg["ds"], fmt_crop, fmt_shape,
                                  fmt_thr, threads=fmt_threads, scratch=scratch)
        buf, start, length = bl.culane_text_buffer(scratch=scratch)
        if gather_q is not None and want_text:
            # (image b's text is buf[start[b] : start[b] + length[b]]; packed back to back here)
            mv = memoryview(buf)
            text = b"".join([mv[a:a + n] for a, n in zip(start.tolist(), length.tolist())])
            nl = int(bl.n_lanes().sum())
            return nl, len(text), (text, np.asarray(length, np.int32), nl)
        return int(bl.n_lanes().sum()), int(length.sum()), None

    if graph is not None or not wl["decode"]:
        args.consume = False     # nothing to format / no per-step result block under a graph
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)]
              if i % max(1, args.event_every) == 0 else None for i in range(args.steps)]
    consumed = [0, 0]
    # the host side is warmed like the device side: worker threads started, the formatting
    # module imported and its first calls (buffer growth, ctypes thunks) made on a warm-up
    # step's records, all before the timed region (a 20-step run would otherwise time them)
    pool = None
    if args.consume:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=args.consume_threads)
        for fut in [pool.submit(consume, hp.last) for _ in range(2 * args.consume_threads)]:
            fut.result()
    if gather_q is not None and not args.consume:
        gather_q = None          # nothing is formatted inside the timed region: nothing to gather
    if gather_q is not None:
        import threading as _th
        gather_thread = _th.Thread(target=gather_loop, daemon=True)
        gather_thread.start()

    def take(fut):
        nl, nb, text = fut.result()
        consumed[0] += nl
        consumed[1] += nb
        if text is not None:
            gather_q.put(text)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # --consume: two host threads format finished steps while this one enqueues the next
    # (the library calls release the GIL); the result blocks are used in turn, so step
    # i + result_blocks is enqueued only after step i has been formatted
    pending = {}
    for i in range(args.steps):
        if pool is not None and hp.cur in pending:
            take(pending.pop(hp.cur))
        if graph is not None:
            graph.replay()
        else:
            one_step(events[i])
        if pool is not None:
            pending[(hp.cur - 1) % len(hp.slots)] = pool.submit(
                consume, hp.last, gather_q is not None and i % max(1, args.gather_every) == 0)
    for fut in pending.values():
        take(fut)
    if pool is not None:
        pool.shutdown()
    if gather_thread is not None:
        gather_q.put(None)
        gather_thread.join()
        if gathered["error"]:
            raise SystemExit("gather of the lane lists failed: " + gathered["error"])
    torch.cuda.synchronize()          # all streams of the device, copies included
    hp.check_lfa()                    # algorithms 3 / 4: no wait inside a launch ran out
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu",
                         dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if graph is not None:
        # per-stage times are not observable inside a replayed graph: time the stages
        # eagerly once more, outside the timed region
        for i in range(args.steps):
            one_step(events[i])
        torch.cuda.synchronize()
    timed = [e for e in events if e is not None]
    lfa_all = np.array([e[0].elapsed_time(e[1]) for e in timed])
    dec_all = np.array([e[3].elapsed_time(e[2]) for e in timed]) if wl["decode"] else np.zeros(1)
    lfa_ms = float(lfa_all.mean())
    dec_ms = float(dec_all.mean()) if wl["decode"] else 0.0

    def spread(a):      # median and the 10th / 90th percentile of the bracketed steps
        return {"p10": float(np.percentile(a, 10)), "median": float(np.median(a)),
                "p90": float(np.percentile(a, 90)), "samples": int(a.size)}

    if rank == 0:
        nbytes, flops = lfa_algorithmic(B, Cin, 64, H, W, K, dtype)
        # float32 tensors on algorithms 1 and 2 are contracted as three bf16 MFMA passes
        # (hi*hi + hi*lo + lo*hi): price the matrix work as 3x the FLOPs at the bf16 peak
        split = dtype == "fp32" and hp.lfa_algorithm in (1, 2, 5) and args.fp32_mode == "auto"
        arith = dtype
        if split:
            flops *= 3.0
            arith = "bf16"
        elif dtype == "fp32" and args.fp32_mode == "bf16":
            arith = "bf16"
        peak_tflops = MFMA_PEAK_TFLOPS[arith]
        t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
        t_mfma = flops / (peak_tflops * 1e12)
        lfa_s = lfa_ms * 1e-3
        hbm_achieved = nbytes / lfa_s / 1e9
        mfma_achieved = flops / lfa_s / 1e12
        # measured L2<->fabric bytes per LFA call: written by tools/traffic_json.py from the
        # PMC passes of tools/profile.sh; only quoted while the kernel sources are the ones
        # that were profiled
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                key = "%s_%s" % (args.workload, dtype)
                if dtype == "fp32" and args.fp32_mode != "auto":
                    key += "_" + args.fp32_mode
                if args.algo:
                    key += "_algo%d" % args.algo
                ent = json.load(open(tfile)).get(key)
                if isinstance(ent, dict):
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    from traffic_json import kernel_source_hash
                    if ent.get("kernel_sources") == kernel_source_hash():
                        traffic = ent["bytes"]
                        traffic_source = "%s @ %s" % (ent.get("summary"), ent.get("head", "")[:7])
                    else:
                        traffic_source = ("stale: %s was measured on other kernel sources (%s)"
                                          % (ent.get("summary"), ent.get("head", "")[:7]))
            except Exception as exc:  # noqa: BLE001
                traffic, traffic_source = None, "unreadable: %r" % (exc,)
        if t_mfma >= t_hbm:
            roof = dict(bound="mfma", achieved=mfma_achieved,
                        peak=peak_tflops, unit="TFLOP/s",
                        frac=mfma_achieved / peak_tflops)
        else:
            roof = dict(bound="hbm", achieved=hbm_achieved, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=hbm_achieved / HBM_PEAK_GBS)
        kernels = ("dcn_fused_kernel" if hp.lfa_algorithm == 1 else
                   "lfa_stream_kernel" if hp.lfa_algorithm == 3 else
                   "lfa_ring_kernel" if hp.lfa_algorithm == 4 else
                   "dcn_taps_kernel" if hp.lfa_algorithm == 5 else
                   "%s + cts_sample_kernel (all chunk pairs of one LFA launch)"
                   % ("cts_gemm_split_kernel" if split else
                      "cts_gemm_ov_kernel" if dtype != "fp32" and Cin in (64, 128, 256)
                      and os.environ.get("GANET_CTS_GEMM", "o")[0] != "t" else "cts_gemm_kernel"))
        roof.update(traffic=traffic, traffic_source=traffic_source, kernel=kernels, lfa_algorithm=hp.lfa_algorithm,
                    kernel_ms=lfa_ms, algorithmic_bytes=nbytes, algorithmic_flops=flops,
                    hbm_gbs=hbm_achieved, hbm_frac=hbm_achieved / HBM_PEAK_GBS,
                    mfma_tflops=mfma_achieved,
                    mfma_frac=mfma_achieved / peak_tflops,
                    mfma_arithmetic=("3 bf16 passes over hi/lo splits, fp32 accumulate"
                                     if split else arith),
                    roofline_time_frac=max(t_hbm, t_mfma) / lfa_s)
        out = {
            "metric": "images/sec (LFA+decode, 320x800)" if wl["decode"]
                      else "images/sec (LFA alone)",
            "value": world * B * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup_steps": spinup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "us_per_step": elapsed / args.steps * 1e6,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.workload,
                       "batch_per_gpu": B, "global_batch": world * B,
                       "feature_hw": [H, W], "cin": Cin, "cout": 64, "k": K,
                       "layout": "NHWC (channels_last) resident in HBM",
                       "fp32_mode": args.fp32_mode if dtype == "fp32" else None,
                       "launch": "hip graph replay" if graph is not None else "eager",
                       "consume": ("every step's records formatted on the host inside the timed "
                                   "region, two host threads beside the GPU") if args.consume else
                                  "records land in pinned host memory; formatting timed beside "
                                  "(host_format)",
                       "streams": ("LFA and decode on two streams" if hp.side is not None
                                   else "LFA and decode as two branches of the graph"
                                   if graph is not None and wl["decode"] and args.graph_fork
                                   else "one stream"),
                       "gather": (("formatted lane lists (CULane text) of every %s step gathered on "
                                   "rank 0 in image order by a host thread (shared-memory rings within "
                                   "the node, gloo across nodes), inside the timed region" % ("" if args.gather_every <= 1 else
                                                     "%d-th" % args.gather_every))
                                  if gather_thread is not None else
                                  ("none (one rank)" if world == 1 else "off")),
                       "inputs": "the same HBM-resident synthetic tensors every step (x: %.0f MB per rank, "
                                 "Infinity Cache 256 MiB); %d untimed spin-up steps before the %d warm-up steps"
                                 % (d["x"].numel() * d["x"].element_size() / 1e6, spinup, args.warmup),
                       "parallelism": "image-sharded x%d, no collective on the data path" % world},
            "roofline": roof,
            "lfa_ms": lfa_ms,
            "decode_ms": dec_ms,
            "lfa_ms_spread": spread(lfa_all),
            "decode_ms_spread": spread(dec_all) if wl["decode"] else None,
            "kernel_time_samples": len(timed),   # steps of the timed region bracketed by events
            "lfa_images_per_s_per_gpu": B / lfa_s,
        }
        if wl["decode"]:
            dbytes = decode_algorithmic(B, H, W)
            out["decode_images_per_s_per_gpu"] = B / (dec_ms * 1e-3)
            out["decode_hbm_gbs"] = dbytes / (dec_ms * 1e-3) / 1e9
        if wl["decode"] and graph is None:
            # host formatting of the timed batch, outside the timed region unless --consume
            slot = hp.last
            consume(slot)
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                nl, nb, _ = consume(slot)
            t_fmt = (time.perf_counter() - t0) / reps
            gpu_ips = B * args.steps / elapsed
            out["host_format"] = {
                "value": B / t_fmt, "unit": "images/s", "cores": fmt_threads,
                "ms_per_batch": t_fmt * 1e3, "lanes_per_batch": nl, "text_bytes_per_batch": nb,
                "what": "records -> lane_post_process + adjust_result (int lanes, more than "
                        "%d points) + CULane .lines.txt text, ganet_lanes_from_packed + "
                        "ganet_culane_text" % fmt_thr,
                "inside_timed_region": bool(args.consume),
                "end_to_end_bound": "gpu" if B / t_fmt >= gpu_ips else "host formatting",
            }
            if args.consume:
                out["host_format"]["lanes_in_timed_region"] = consumed[0]
            if gather_thread is not None:
                out["gather"] = {"steps": gathered["steps"], "images": gathered["images"],
                                 "bytes": gathered["bytes"],
                                 "bytes_per_step": gathered["bytes"] / max(1, gathered["steps"]),
                                 "images_per_step": gathered["images"] / max(1, gathered["steps"]),
                                 "lanes_per_step": gathered["lanes"] / max(1, gathered["steps"]),
                                 "first_step_lanes_by_rank": gathered.get("first_step_lanes_by_rank"),
                                 "rank0_images_per_step": B,
                                 "rank0_lanes_per_step": consumed[0] / max(1, args.steps)}
        if world == 1 and not args.no_cpu_baseline:
            # every core this process may use: its affinity mask, or what the box grants
            # through OMP_NUM_THREADS where that is smaller (nproc shows the whole machine)
            avail = len(os.sched_getaffinity(0))
            omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            threads = args.cpu_threads or (min(avail, omp) if omp > 0 else avail)
            base, parity = cpu_baseline_and_parity(wl, dtype, d, hp, args.cpu_images,
                                                   threads)
            out["cpu_baseline"] = base
            out["parity"] = parity
        if world == 1 and not args.no_backward:
            out["lfa_backward"] = lfa_backward_side_measurement(dev, Cin, H, W, K)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
