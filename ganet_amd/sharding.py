"""Image sharding of the hot path over the GPUs of one node.

The path has no cross-image term, so the multi-GPU model is the reference's own
(tools/dist_test.sh:13-24, tools/ganet/tusimple/test_dataset.py:252-264): one
process per GPU, each owns a contiguous block of the images, weights are
replicated, and only the per-image lane lists are gathered on the host.  No
collective runs on the data path; `gather_lane_lists` is control-plane traffic
(pickled Python objects over the process group, a few KB per image).
"""
import os
import socket
import struct
import time

import torch.distributed as dist


def shard_range(n_images, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_images, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_lane_lists(local_lists, dst=0, group=None):
    """Concatenate per-image results of all ranks in image order on rank `dst`.

    `local_lists` is this rank's list (one entry per image of its block, in
    order).  Returns the full list on `dst`, None elsewhere.  Works on any
    backend (gloo on CPU, nccl = RCCL on GPUs) because only host objects move.
    """
    if not dist.is_available() or not dist.is_initialized():
        return list(local_lists)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(list(local_lists), gathered, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for part in gathered:
        out.extend(part)
    return out


def gather_text_blocks(text, lengths, dst=0, group=None):
    """Gather every rank's formatted result block on rank `dst`, in image order.

    `text`: the bytes of this rank's images back to back (e.g. the CULane `.lines.txt` texts
    of formats.BatchLanes.culane_text_buffer, image b's text being `lengths[b]` bytes long);
    `lengths`: one entry per image of the rank's block.  Returns (bytes, lengths) of all
    images on `dst` - rank 0's block first, then rank 1's, ...: the order of shard_range -
    and None elsewhere.  Host objects only (the reference gathers pickled result lists the same
    way, mmdet/apis/test.py:122-166); give it a gloo group when the default one is RCCL so that
    the GPUs' streams are not involved."""
    lengths = [int(n) for n in lengths]
    text = bytes(text)
    if sum(lengths) != len(text):
        raise ValueError("lengths do not add up to the text")
    if not dist.is_available() or not dist.is_initialized():
        return text, lengths
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) != "gloo":
        # host objects over a GPU backend: the generic (pickling) path
        gathered = [None] * world if rank == dst else None
        dist.gather_object((text, lengths), gathered, dst=dst, group=group)
        if rank != dst:
            return None
        return b"".join(g[0] for g in gathered), [n for g in gathered for n in g[1]]
    # gloo: two collectives on plain byte tensors, no pickling - the sizes (tiny), then one
    # gather of [lengths as int32 | text] padded to the largest block
    import numpy as np
    import torch
    payload = np.asarray(lengths, np.int32).tobytes() + text
    mine = torch.tensor([len(payload), len(lengths)], dtype=torch.int64)
    sizes = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    cap = max(int(t[0]) for t in sizes)
    send = torch.zeros(max(cap, 1), dtype=torch.uint8)
    if payload:
        send[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    recv = [torch.empty(max(cap, 1), dtype=torch.uint8) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    texts, all_lengths = [], []
    for t, r in zip(sizes, recv):
        nbytes, nimg = int(t[0]), int(t[1])
        raw = r.numpy()[:nbytes].tobytes()
        all_lengths.extend(np.frombuffer(raw[:4 * nimg], np.int32).tolist())
        texts.append(raw[4 * nimg:])
    return b"".join(texts), all_lengths


class NodeGather(object):
    """The same gather for the ranks of ONE node (north_star's case: the 8 GPUs of one MI355X
    node) without a process group: a gloo collective costs milliseconds per call on loopback
    TCP, a step of the hot path 0.3 ms.  Every other rank owns a small ring of `slots` blocks
    of `block_bytes` in shared memory (a file under /dev/shm that is unlinked as soon as rank 0
    has mapped it, so nothing outlives the processes) and a Unix-domain stream socket to rank 0
    (abstract address derived from `key`; every rank passes the same key, e.g. the rendezvous
    port).  `gather(text, lengths)` is called once per step by every rank, in the same order:
    a rank copies its block into its next free slot and writes a 32-byte header to the socket;
    rank 0 reads the headers in rank order, copies each block from the ring into its place in
    one buffer and returns a one-byte credit.  So a block is copied twice (once by its owner,
    once by rank 0) and the socket carries only headers and credits; its system calls also order
    the ring's plain stores and loads between the two processes.  A rank waits only when rank 0
    is `slots` steps behind it.  A block larger than `block_bytes` travels through the socket
    itself.  Rank 0 returns (text, lengths) of all images in image order (text: a bytearray), the
    others None.  No
    GPU, no stream and no process group is involved."""

    HEAD = struct.Struct("<qqqq")         # bytes of text, images, 1 = in the ring / 0 = follows inline, tag

    def __init__(self, rank, world, key, timeout=60.0, block_bytes=2 << 20, slots=8):
        import mmap
        self.rank, self.world = rank, world
        self.cap, self.slots = int(block_bytes), int(slots)
        self.peers, self.rings = [], []
        self.sent = self.credit = 0
        self.last_tags = []
        if world == 1:
            return
        addr = "\0ganet_gather_%s_%d" % (key, os.getuid())
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(addr)
            srv.listen(world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                r, cap, slots, n = struct.unpack("<qqqq", self._read(conn, 32))
                path = self._read(conn, n).decode()
                fd = os.open(path, os.O_RDWR)
                try:
                    ring = mmap.mmap(fd, cap * slots)
                finally:
                    os.close(fd)
                conn.sendall(b"m")                      # mapped: the owner may unlink the file
                by_rank[r] = (conn, ring, cap, slots)
            srv.close()
            for r in range(1, world):
                self.peers.append(by_rank[r][0])
                self.rings.append(by_rank[r][1:])
            self.taken = [0] * (world - 1)
            self.owed = [0] * (world - 1)
        else:
            path = "/dev/shm/ganet_gather_%s_%d_%d" % (key, os.getuid(), rank)
            fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
            try:
                os.ftruncate(fd, self.cap * self.slots)
                ring = mmap.mmap(fd, self.cap * self.slots)
            finally:
                os.close(fd)
            try:
                t0 = time.monotonic()
                while True:
                    so = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                    try:
                        so.connect(addr)
                        break
                    except OSError:
                        so.close()
                        if time.monotonic() - t0 > timeout:
                            raise
                        time.sleep(0.01)
                so.settimeout(timeout)
                name = path.encode()
                so.sendall(struct.pack("<qqqq", rank, self.cap, self.slots, len(name)) + name)
                self._read(so, 1)
            finally:
                os.unlink(path)
            self.peers, self.rings = [so], [ring]

    @staticmethod
    def _read_into(conn, view):
        got, n = 0, len(view)
        while got < n:
            k = conn.recv_into(view[got:], n - got)
            if k == 0:
                raise ConnectionError("a rank closed its gather socket")
            got += k

    @classmethod
    def _read(cls, conn, n):
        buf = bytearray(n)
        cls._read_into(conn, memoryview(buf))
        return bytes(buf)

    def gather(self, text, lengths, tag=0, into=None):
        """tag: one integer that travels with the block (the bench sends its lane count, so
        that rank 0 need not scan the text); rank 0 finds every rank's in `last_tags`.
        into: a bytearray that rank 0 fills instead of allocating one per step (used when it
        is large enough; the returned text is then a memoryview of its first bytes and is valid
        until the next call with the same buffer)."""
        import numpy as np
        lengths = np.ascontiguousarray(lengths, dtype=np.int32).reshape(-1)
        if int(lengths.sum(dtype=np.int64)) != len(text):
            raise ValueError("lengths do not add up to the text")
        if self.world == 1:
            self.last_tags = [int(tag)]
            return bytes(text), lengths.tolist()
        if self.rank != 0:
            so, ring = self.peers[0], self.rings[0]
            need = lengths.nbytes + len(text)
            if need > self.cap:
                so.sendall(self.HEAD.pack(len(text), len(lengths), 0, int(tag)) + lengths.tobytes()
                           + bytes(text))
                return None
            while self.sent - self.credit >= self.slots:  # rank 0 is `slots` blocks behind: wait
                got = so.recv(64)                         # one byte per slot that it has freed
                if not got:
                    raise ConnectionError("rank 0 closed its gather socket")
                self.credit += len(got)
            at = (self.sent % self.slots) * self.cap
            ring[at:at + lengths.nbytes] = lengths.tobytes()
            np.frombuffer(ring, np.uint8, len(text), at + lengths.nbytes)[:] = np.frombuffer(text, np.uint8)
            self.sent += 1
            so.sendall(self.HEAD.pack(len(text), len(lengths), 1, int(tag)))
            return None
        heads = []
        for conn in self.peers:
            nbytes, nimg, in_ring, peer_tag = self.HEAD.unpack(self._read(conn, self.HEAD.size))
            if nbytes < 0 or nimg < 0:
                raise ConnectionError("bad gather header")
            heads.append((nbytes, nimg, in_ring, peer_tag))
            if not in_ring:                               # oversized block: it follows in the socket
                heads[-1] += (self._read(conn, 4 * nimg), self._read(conn, nbytes))
        # numpy does the large copies: it lets go of the interpreter lock meanwhile, and the thread
        # that launches the GPU work of this rank runs on
        total = len(text) + sum(h[0] for h in heads)
        out = into if into is not None and len(into) >= total else bytearray(total)
        dst = np.frombuffer(out, np.uint8)
        parts = [lengths]
        dst[:len(text)] = np.frombuffer(text, np.uint8)
        at = len(text)
        for i, (conn, h) in enumerate(zip(self.peers, heads)):
            nbytes, nimg, in_ring = h[:3]
            if in_ring:
                ring, cap, slots = self.rings[i]
                src = (self.taken[i] % slots) * cap
                self.taken[i] += 1
                parts.append(np.frombuffer(ring, np.int32, nimg, src).copy())
                dst[at:at + nbytes] = np.frombuffer(ring, np.uint8, nbytes, src + 4 * nimg)
                # credits go back in batches of half a ring: each write to the socket may wake
                # the owner, and a wake-up costs more than the copy
                self.owed[i] += 1
                if self.owed[i] >= max(1, slots // 2):
                    try:
                        conn.sendall(b"c" * self.owed[i])
                    except OSError:
                        pass                              # its owner has finished and gone
                    self.owed[i] = 0
            else:
                parts.append(np.frombuffer(h[4], np.int32))
                dst[at:at + nbytes] = np.frombuffer(h[5], np.uint8)
            at += nbytes
        del dst
        self.last_tags = [int(tag)] + [h[3] for h in heads]
        all_lengths = np.concatenate(parts).tolist()
        return (memoryview(out)[:total] if out is into else out), all_lengths

    def close(self):
        for conn in self.peers:
            conn.close()
        for ring in self.rings:
            try:
                (ring[0] if isinstance(ring, tuple) else ring).close()
            except (BufferError, ValueError):
                pass
        self.peers, self.rings = [], []


def run_sharded(process_block, n_images, group=None, dst=0):
    """Run `process_block(lo, hi) -> list` on this rank's block and gather."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(n_images, rank, world)
    local = process_block(lo, hi)
    if len(local) != hi - lo:
        raise RuntimeError("process_block returned %d results for %d images"
                           % (len(local), hi - lo))
    return gather_lane_lists(local, dst=dst, group=group)
