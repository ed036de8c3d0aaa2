"""Image sharding of the hot path over the GPUs of one node.

The path has no cross-image term, so the multi-GPU model is the reference's own
(tools/dist_test.sh:13-24, tools/ganet/tusimple/test_dataset.py:252-264): one
process per GPU, each owns a contiguous block of the images, weights are
replicated, and only the per-image lane lists are gathered on the host.  No
collective runs on the data path; `gather_lane_lists` is control-plane traffic
(pickled Python objects over the process group, a few KB per image).
"""
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
    gathered = [None] * world if rank == dst else None
    dist.gather_object((text, lengths), gathered, dst=dst, group=group)
    if rank != dst:
        return None
    return b"".join(g[0] for g in gathered), [n for g in gathered for n in g[1]]


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
