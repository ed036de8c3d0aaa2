"""The Lane-aware Feature Aggregator's tail as one fused operator.

Mirror of LanePointsConv.forward_single lines 170-172
(mmdet/models/dense_heads/lanepoints_conv.py:141-174):

    dcn_offset  = pts_out_init - dcn_base_offset
    feature_out = relu(DeformConv1D(cls_feat, dcn_offset))

The reference runs an elementwise subtract kernel, the deform-conv op and a
ReLU kernel; here the subtraction happens while the sampling coordinates are
formed and the ReLU in the contraction's epilogue.
"""
import torch
import torch.nn as nn

from torch.autograd import Function

from .dcn import DeformConv1D, _keep_holder, deform_conv_backward, deform_conv_forward


def dcn_base_offset(num_points, dtype=torch.float32, device=None):
    """The 1 x K kernel grid DeformConv1D adds to its offsets, as (dy, dx) pairs per tap:
    dy = 0, dx = tap - (K - 1) // 2, shaped [1, 2K, 1, 1] (what lanepoints_conv.py:63-74
    builds and :170 subtracts from the regressed points)."""
    half = (num_points - 1) // 2
    grid = torch.zeros(num_points, 2, dtype=torch.float64)
    grid[:, 1] = torch.arange(-half, half + 1, dtype=torch.float64)
    return grid.reshape(1, 2 * num_points, 1, 1).to(dtype=dtype, device=device)


class _FusedLfaTail(Function):
    """relu(DeformConv1D(x, pts - base)) as ONE forward call with gradients: the backward
    masks grad_output with the ReLU (out > 0) and runs ganet_deform_conv_backward on
    offset = pts - base; d(offset)/d(pts) = 1, so the offset gradient is the gradient of
    `pts_out_init` (which the reference trains through, lanepoints_conv.py:159-172)."""

    @staticmethod
    def forward(ctx, cls_feat, pts_out_init, weight, stride, padding, dilation, num_points):
        # float32 training: the per-tap products stay in the forward's workspace for the
        # backward (they do not depend on the offsets, fused subtraction or not)
        kept = _keep_holder(ctx, cls_feat, 1)
        out = deform_conv_forward(cls_feat, pts_out_init, weight, stride, padding, dilation,
                                  1, 1, fuse_relu=True, offset_is_points=True, keep_y=kept)
        ctx.y_saved = kept.get("y") if kept is not None else None
        ctx.geometry = (stride, padding, dilation, num_points)
        ctx.save_for_backward(cls_feat, pts_out_init, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        cls_feat, pts, weight, out = ctx.saved_tensors
        stride, padding, dilation, num_points = ctx.geometry
        grad_pre = (grad_output * (out > 0).to(grad_output.dtype)).contiguous(
            memory_format=torch.channels_last if out.is_contiguous(
                memory_format=torch.channels_last) and not out.is_contiguous()
            else torch.contiguous_format)
        offset = (pts - dcn_base_offset(num_points, pts.dtype, pts.device)).contiguous()
        want = ctx.needs_input_grad
        g_in, g_off, _, g_w, _ = deform_conv_backward(
            cls_feat, offset, weight, grad_pre, stride, padding, dilation, 1, 1,
            need_input=want[0] or want[1], need_weight=want[2], y_saved=ctx.y_saved)
        ctx.y_saved = None
        return g_in, g_off, g_w, None, None, None, None


class LaneAwareAggregator(nn.Module):
    """`reppoints_cls_conv` + the offset subtraction + ReLU of LanePointsConv.

    Holds the same parameter as the reference's DeformConv1D (`conv.weight`,
    [Cout, Cin, 1, K], no bias), so a reference state dict loads unchanged
    under the key `reppoints_cls_conv.weight` -> `conv.weight`.
    """

    def __init__(self, feat_channels=256, point_feat_channels=64, num_points=7):
        super().__init__()
        assert num_points % 2 == 1, 'The points number should be an odd number.'
        self.num_points = num_points
        self.dcn_pad = int((num_points - 1) / 2)
        self.conv = DeformConv1D(feat_channels, point_feat_channels, num_points,
                                 1, self.dcn_pad)

    def forward(self, cls_feat, pts_out_init):
        c = self.conv
        if torch.is_grad_enabled() and (cls_feat.requires_grad or pts_out_init.requires_grad
                                        or c.weight.requires_grad):
            # training: the same fused forward, recorded for autograd
            return _FusedLfaTail.apply(cls_feat, pts_out_init, c.weight, c.stride, c.padding,
                                       c.dilation, self.num_points)
        return deform_conv_forward(cls_feat, pts_out_init, c.weight, c.stride, c.padding,
                                   c.dilation, 1, 1, fuse_relu=True, offset_is_points=True)

    def forward_unfused(self, cls_feat, pts_out_init):
        """The reference's three-step form (for parity tests)."""
        base = dcn_base_offset(self.num_points, pts_out_init.dtype,
                               pts_out_init.device)
        return torch.relu(self.conv(cls_feat, (pts_out_init - base).contiguous()))
