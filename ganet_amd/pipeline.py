"""Allocation-free steady-state driver of the hot path for one GPU.

`HotPath` owns every buffer the C-ABI needs (packed weights, workspaces, the
LFA output, the decode output block and its pinned host mirror), so a serving
loop - or bench.py - enqueues one step with no allocator traffic:

    LFA   : feature = relu(DeformConv1D(x, pts - base))   (lanepoints_conv.py:170-172)
    decode: NMS + compaction + grouping for the whole batch, then ONE async
            D2H copy of the compact result block (ganet_head.py:203-238,
            post_process.py:141-194)

Images are independent: on N GPUs each rank owns a contiguous block of the
batch and runs its own HotPath; nothing is exchanged between ranks.
"""
import ctypes

import torch

from . import _lib
from .decode import DecodeBuffers, DecodeResult, _params, records_in_image_order


class HotPath(object):

    def __init__(self, B, H, W, Cin=256, Cout=64, K=7, dtype=torch.bfloat16,
                 device=None, thr=0.4, root_thr=1.0, cluster_thr=5.0,
                 by_center_thr=5.0, max_kpts=1024, max_centers=128,
                 channels_last=True, mfma_bf16=False, fuse_relu=True,
                 offset_is_points=True, expect_kpts=192, expect_centers=16,
                 overlap_copy=False, ablate=0, algo=0, pack_to_host=True,
                 side_decode=False, heat_is_logits=False, result_blocks=2):
        self.L = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.B, self.H, self.W, self.K = B, H, W, K
        self.dtype = dtype
        self.channels_last = channels_last
        p = _lib.DcnParams()
        p.B, p.Cin, p.H, p.W, p.Cout = B, Cin, H, W, Cout
        p.kH, p.kW, p.sH, p.sW = 1, K, 1, 1
        p.pH, p.pW, p.dH, p.dW = 0, (K - 1) // 2, 1, 1
        p.groups = p.deformable_groups = 1
        p.dtype = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
        p.x_layout = p.out_layout = 1 if channels_last else 0
        # float32 only: 0 float32-class accuracy (may use three bf16 passes), 1 one bf16
        # pass, 2 exact float32 MFMA (ganet_dcn_params.mfma_bf16)
        p.mfma_bf16 = int(mfma_bf16) if dtype == torch.float32 else 0
        p.fuse_relu = 1 if fuse_relu else 0
        p.offset_is_points = 1 if offset_is_points else 0
        p.weight_is_packed = 1
        p.debug = ablate         # profiling builds only (see csrc/deform_conv.hip)
        p.algo = algo
        self.p = p
        with torch.cuda.device(self.device):
            nb = self.L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(p))
            if nb == 0:
                raise ValueError("unsupported LFA configuration")
            self.packed = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self.ws_bytes = self.L.ganet_deform_conv_workspace_bytes(ctypes.byref(p))
            # 1 = fused sample-then-contract kernel, 2 = contract-then-sample pair, 3 / 4 = that
            # pair as one persistent launch (both leave a status word in the workspace)
            self.lfa_algorithm = self.L.ganet_deform_conv_algorithm(ctypes.byref(p))
            off = self.L.ganet_deform_conv_status_offset(ctypes.byref(p))
            self._status_off = None if off == ctypes.c_size_t(-1).value else off
            self.ws = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8,
                                  device=self.device)
            self.feature = torch.empty(
                (B, Cout, H, W), dtype=dtype, device=self.device,
                memory_format=torch.channels_last if channels_last
                else torch.contiguous_format)
            by_center = by_center_thr is not None
            caps = (max_kpts, max_centers, max_centers if by_center else max_kpts)
            # heat_is_logits: the head's raw keypoint map goes in; clamp(sigmoid(.)) of
            # ganet_head.py:291 happens inside the decode (ganet_decode_params.flags bit3)
            self.dp = _params(B, H, W, thr, root_thr, cluster_thr, by_center_thr, caps,
                              flags=8 if heat_is_logits else 0)
            # result blocks + pinned mirrors, used in turn: the D2H copy of step i runs
            # while step i+1 computes, and a host thread may still be reading step i's
            # records while steps up to i + result_blocks - 1 are enqueued
            assert result_blocks >= 2
            self.slots = []
            for _ in range(result_blocks):
                bufs = DecodeBuffers(B, caps[0], caps[1], caps[2], self.device)
                # the async copy moves the header plus a budget of record words
                # sized for the expected load; result() fetches any remainder
                budget = bufs.header_words + B * (
                    7 * min(expect_kpts, caps[0]) + 6 * min(expect_centers, caps[1]))
                budget = min(budget, bufs.packed_words)
                host = torch.empty(bufs.packed_words, dtype=torch.int32, pin_memory=True)
                bufs.budget = budget
                self.slots.append(dict(bufs=bufs, host=host, ready=torch.cuda.Event(),
                                       done=torch.cuda.Event(), copied=torch.cuda.Event()))
            self.cur = 0
            self.last = None
            self.bufs = self.slots[0]["bufs"]
            # (zero-filled: the counters of ganet_decode_batch_packed start at zero, and every
            # call leaves them there)
            self.dws_bytes = self.L.ganet_decode_packed_workspace_bytes(ctypes.byref(self.dp))
            self.dws = torch.zeros(max(self.dws_bytes, 256), dtype=torch.uint8,
                                   device=self.device)
            self.overlap_copy = overlap_copy
            # side_decode: decode + grouping + pack go to a stream of their own.  They do
            # not depend on the LFA of the same step, are latency-bound (a few dozen
            # workgroups) and slot into the LFA kernels' ramps and tails instead of
            # extending the step.
            self.side = torch.cuda.Stream(device=self.device) if side_decode else None
            self.inputs_ready = torch.cuda.Event() if side_decode else None
            self.side_done = torch.cuda.Event() if side_decode else None
            self._side_busy = False
            self.pack_to_host = pack_to_host and not overlap_copy
            self.copy_stream = (torch.cuda.Stream(device=self.device) if overlap_copy
                                else None)
        self.params = dict(thr=thr, root_thr=root_thr, cluster_thr=cluster_thr,
                           by_center_thr=by_center_thr)
        self._has_weight = False

    def set_weight(self, weight):
        """weight [Cout, Cin, 1, K] in the module's dtype; packed once."""
        weight = weight.detach().to(device=self.device, dtype=self.dtype).contiguous()
        q = _lib.DcnParams.from_buffer_copy(self.p)
        q.weight_is_packed = 0
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.L.ganet_deform_conv_pack_weight(
            weight.data_ptr(), ctypes.byref(q), self.packed.data_ptr(),
            self.packed.numel(), stream)
        _lib.check(rc, "ganet_deform_conv_pack_weight")
        torch.cuda.current_stream(self.device).synchronize()
        self._has_weight = True

    def lfa(self, x, pts):
        """Enqueue the LFA on the current stream; returns the (reused) output."""
        assert self._has_weight, "call set_weight first"
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.L.ganet_deform_conv_forward(
            x.data_ptr(), pts.data_ptr(), None, self.packed.data_ptr(), None,
            self.feature.data_ptr(), ctypes.byref(self.p), self.ws.data_ptr(),
            self.ws_bytes, stream)
        _lib.check(rc, "ganet_deform_conv_forward")
        return self.feature

    def lfa_status(self):
        """The status word of the persistent LFA launches (algorithms 3 and 4), 0 for the other
        algorithms.  Synchronises the device.  Non-zero: a bounded wait inside the launch ran out
        (the GPU could not keep its workgroups resident, e.g. beside other streams' kernels) and
        `feature` is invalid."""
        if self._status_off is None:
            return 0
        torch.cuda.synchronize(self.device)
        return int(self.ws[self._status_off:self._status_off + 4].view(torch.int32).item())

    def check_lfa(self):
        """Raises if the last LFA call left a failure word (see lfa_status)."""
        word = self.lfa_status()
        if word != 0:
            raise RuntimeError("the persistent LFA launch (algorithm %d) gave up waiting (status "
                               "word %d): its output is invalid" % (self.lfa_algorithm, word))

    def decode(self, heat, offset, error, copy_to_host=True):
        """Enqueue decode + grouping on the current stream, then the D2H copy of
        the result block on the copy stream (it overlaps the next step)."""
        slot = self.slots[self.cur]
        cur_stream = torch.cuda.current_stream(self.device)
        # the block is free again once its previous copy has finished
        if self.overlap_copy:
            cur_stream.wait_event(slot["copied"])
        if copy_to_host and self.pack_to_host:
            # the record stream goes straight into the pinned host block, from the kernel
            # that groups the keypoints where one workgroup per image does the whole decode
            # (one launch), else from the pack kernel behind the decode's: no copy engine
            rc = self.L.ganet_decode_batch_packed(
                heat.data_ptr(), offset.data_ptr(), error.data_ptr(),
                ctypes.byref(self.dp), ctypes.byref(slot["bufs"].out),
                slot["host"].data_ptr(), slot["host"].numel(),
                self.dws.data_ptr(), self.dws_bytes, cur_stream.cuda_stream)
            _lib.check(rc, "ganet_decode_batch_packed")
            slot["direct"] = True
            slot["ready"].record(cur_stream)
            self.last = slot
            self.cur = (self.cur + 1) % len(self.slots)
            return
        rc = self.L.ganet_decode_batch(
            heat.data_ptr(), offset.data_ptr(), error.data_ptr(),
            ctypes.byref(self.dp), ctypes.byref(slot["bufs"].out),
            self.dws.data_ptr(), self.dws_bytes, cur_stream.cuda_stream)
        _lib.check(rc, "ganet_decode_batch")
        slot["direct"] = False
        slot["bufs"].pack(self.L, self.dp, cur_stream.cuda_stream)
        if copy_to_host:
            n = slot["bufs"].budget
            if self.overlap_copy:
                slot["done"].record(cur_stream)
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(slot["done"])
                    slot["host"][:n].copy_(slot["bufs"].packed[:n], non_blocking=True)
                    slot["copied"].record(self.copy_stream)
            else:
                slot["host"][:n].copy_(slot["bufs"].packed[:n], non_blocking=True)
        slot["ready"].record(self.copy_stream if (copy_to_host and self.overlap_copy)
                             else cur_stream)
        self.last = slot
        self.cur = (self.cur + 1) % len(self.slots)

    def step(self, x, pts, heat, offset, error):
        """One pass of the hot path.  Lifetime contract with side_decode: decode, grouping and
        pack read heat/offset/error on the side stream; the caller's stream is made to wait
        for them at the start of the NEXT step (side_wait()), so those tensors must stay
        alive and unchanged until then (or until result()); a caller that frees or rewrites
        them earlier calls side_wait() itself first."""
        if self.side is None:
            self.lfa(x, pts)
            self.decode(heat, offset, error)
            return
        # the step's inputs are ready where the current stream stands now
        cur = torch.cuda.current_stream(self.device)
        self.side_wait()
        self.inputs_ready.record(cur)
        self.lfa(x, pts)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.inputs_ready)
            self.decode(heat, offset, error)
            self.side_mark()

    def side_wait(self):
        """The current stream waits for the side stream's last decode (no host wait)."""
        if self.side is not None and self._side_busy:
            torch.cuda.current_stream(self.device).wait_event(self.side_done)

    def side_mark(self):
        """Called on the side stream after a decode was enqueued there."""
        self.side_done.record(torch.cuda.current_stream(self.device))
        self._side_busy = True

    def capture(self, x, pts, heat, offset, error, fork=False):
        """Record one step on these (fixed) input tensors into a HIP graph and return
        it; `graph.replay()` then enqueues the whole step with one host call (at batch 1
        the host launch cost is a quarter of the step).  The step allocates nothing and
        never synchronises, so it captures as it stands; `result()` works after a replay
        as after `step()`.  fork=True: the LFA and the decode do not depend on one another, so
        they are recorded as two branches of the graph (a second stream forks off the
        capturing one and joins it again).  Measured SLOWER at batch 1 on MI355X / ROCm 7.2
        (41-53 against 29 us per replay, profiles/r03g_bench_c1_bf16_graph*.json: the two
        branches run on two hardware queues and the fork and the join cost more than the
        17 us of overlap they buy), hence off by default."""
        assert not self.overlap_copy and self.side is None, "capture needs the single-stream flow"
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):       # warm-up outside the capture
            self.step(x, pts, heat, offset, error)
        cur.wait_stream(side)
        cur.synchronize()
        first = self.cur
        graph = torch.cuda.CUDAGraph()
        branch = torch.cuda.Stream(device=self.device) if fork else None
        with torch.cuda.graph(graph):
            self.cur = first                # the graph always writes this result block
            if branch is None:
                self.step(x, pts, heat, offset, error)
            else:
                main = torch.cuda.current_stream(self.device)
                branch.wait_stream(main)    # fork
                self.lfa(x, pts)
                with torch.cuda.stream(branch):
                    self.decode(heat, offset, error)
                main.wait_stream(branch)    # join
        self.cur = (first + 1) % len(self.slots)
        self._graph_branch = branch         # keep the stream alive as long as the graph may
        return graph

    def result(self, slot=None):
        """Wait for a step's lane lists and view them.  slot=None: the last step, after
        everything enqueued so far; a slot (HotPath.last, remembered by the caller right
        after a step) is waited for alone, so the next step may already be running."""
        if slot is None:
            torch.cuda.current_stream(self.device).synchronize()
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            if self.side is not None:
                self.side.synchronize()
            self.check_lfa()            # (algorithms 3 and 4 only; free for the others)
            slot = self.last
        else:
            slot["ready"].synchronize()
        host, total = self._host_stream(slot)
        hw = slot["bufs"].header_words
        header = host[:hw].reshape(self.B, _lib.DECODE_COUNT_STRIDE).copy()
        records = host[hw:total].copy()
        if slot.get("direct"):    # (one-launch decode: records sit in arrival order)
            header, records = records_in_image_order(header, records)
        return DecodeResult(header, records, self.params, grouped=True)

    def _host_stream(self, slot):
        bufs, host = slot["bufs"], slot["host"].numpy()
        total = bufs.header_words + int(host[6])
        if not slot.get("direct") and total > bufs.budget:   # beyond the async budget
            slot["host"][bufs.budget:total].copy_(bufs.packed[bufs.budget:total])
        return host, total

    def records(self, slot):
        """The record stream of a step (header + records, the layout of ganet_decode_pack)
        as a view of the pinned host block - the input of formats.format_batch.  Waits for
        that step only; valid until the block comes round again (result_blocks steps later)."""
        slot["ready"].synchronize()
        host, total = self._host_stream(slot)
        return host[:total]
