"""N>1 path on CPU: world_size 2 over gloo.  The per-image work is done by the
CPU oracle here (test stand-in for the GPU kernels); what is under test is the
block split, the absence of any data-path collective and the host gather."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from ganet_amd import sharding, synth


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _lanes_of_block(lo, hi, B):
    from oracle import cpu_oracle as O
    cfg = synth.SHAPES["s8_tusimple"]
    heat, off, err = synth.decode_inputs(B, cfg["H"], cfg["W"], lanes=4, seed=21)
    out = []
    for b in range(lo, hi):
        c, kxy, al, rt = O.decode(heat[b], off[b], err[b], cfg["thr"], 1.0)
        gid, means, cgid, ng = O.group_fast(c, rt, cfg["T"], cfg["T"])
        out.append((b, gid.tobytes(), means.tobytes(), al.tobytes()))
    return out


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = sharding.run_sharded(lambda lo, hi: _lanes_of_block(lo, hi, B), B)
        if rank == 0:
            q.put(res)
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_equals_single_process():
    B, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _lanes_of_block(0, B, B)
    assert [g[0] for g in got] == list(range(B))
    assert got == want


def _text_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(B, rank, world)
        texts = [("image %d\n" % b * (b % 3)).encode() for b in range(lo, hi)]   # some are empty
        res = sharding.gather_text_blocks(b"".join(texts), [len(t) for t in texts])
        if rank == 0:
            q.put(res)
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_text_blocks_are_gathered_in_image_order():
    """What bench.py --gpus N does with every step's formatted lane lists: rank 0 ends up with
    the texts of all images, rank 0's block first (the order of shard_range)."""
    B, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_text_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    text, lengths = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [("image %d\n" % b * (b % 3)).encode() for b in range(B)]
    assert lengths == [len(t) for t in want]
    assert text == b"".join(want)
    with pytest.raises(ValueError):
        sharding.gather_text_blocks(b"abc", [1, 1])
    # no process group: the local block comes back as it is
    assert sharding.gather_text_blocks(b"abc", [1, 2]) == (b"abc", [1, 2])


def _node_worker(rank, world, key, B, steps, q, block_bytes):
    g = sharding.NodeGather(rank, world, key, block_bytes=block_bytes, slots=4)
    try:
        got = []
        for step in range(steps):
            lo, hi = sharding.shard_range(B, rank, world)
            texts = [("s%d i%d\n" % (step, b) * (b % 3)).encode() for b in range(lo, hi)]
            # odd steps: into a buffer that rank 0 keeps (the text is then a view of it)
            keep = bytearray(4096) if step % 2 else None
            res = g.gather(b"".join(texts), [len(t) for t in texts], tag=100 * step + rank, into=keep)
            assert (res is None) == (rank != 0)
            if rank == 0:
                assert g.last_tags == [100 * step + r for r in range(world)]
                if keep is not None:
                    assert isinstance(res[0], memoryview) and res[0].obj is keep
                got.append((bytes(res[0]), res[1]))
        if rank == 0:
            q.put(got)
    finally:
        g.close()


@pytest.mark.parametrize("block_bytes", [1 << 16, 48])
def test_node_gather_keeps_image_order(block_bytes):
    """bench.py's transport on one node: three ranks, more steps than a ring has slots (so the
    credits are exercised), rank 0 gets every step's blocks in image order; no process group
    involved.  With 48-byte slots some blocks fit their ring and some travel through the socket."""
    B, world, steps = 11, 3, 19
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "t%d_%d" % (os.getpid(), block_bytes)
    procs = [ctx.Process(target=_node_worker, args=(r, world, key, B, steps, q, block_bytes))
             for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for step, (text, lengths) in enumerate(got):
        want = [("s%d i%d\n" % (step, b) * (b % 3)).encode() for b in range(B)]
        assert lengths == [len(t) for t in want]
        assert text == b"".join(want)
    solo = sharding.NodeGather(0, 1, "x")
    assert solo.gather(b"ab", [2]) == (b"ab", [2])
