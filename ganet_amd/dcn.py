"""Host-side mirror of the reference's deformable-convolution module API
(mmdet/ops/dcn/deform_conv.py) on top of the gfx950 C-ABI.

Same names, constructor arguments, parameter names/shapes, state-dict keys and
exception behaviour as the reference:
  DeformConvFunction / deform_conv        deform_conv.py:15-111, 187
  ModulatedDeformConvFunction / modulated_deform_conv   :114-184, 188
  DeformConv :191-253, DeformConv1D :255-317, DeformConvPack :320-387
  ModulatedDeformConv :390-437, ModulatedDeformConvPack :440-512
Only names, signatures, attribute names, state-dict keys and exception types/messages
are the reference's (they are the drop-in contract); the bodies are written against the
C-ABI.  Forward and backward run the HIP kernels (csrc/deform_conv*.hip); there is no
eager / CPU fallback: CPU tensors raise NotImplementedError like the reference.  The backward
takes float32, bfloat16 and float16 tensors (16-bit ones are widened inside the library).
'DCN' / 'DCNv2' are registered with mmcv's CONV_LAYERS only on request:
register_with_mmcv().
"""
import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.nn.modules.utils import _pair, _single

from . import _lib

_DTYPES = {torch.float32: _lib.GANET_F32, torch.bfloat16: _lib.GANET_BF16,
           torch.float16: _lib.GANET_F16}


class _Config(object):
    """How float32 tensors are contracted (ganet_dcn_params.mfma_bf16):
    default - float32-class accuracy: the exact float32 MFMA, or, where the
    contract-then-sample algorithm applies (the LFA's 256 -> 64), three bf16 MFMA
    passes over hi/lo splits of both operands (~4e-6 of the output scale);
    mfma_bf16 = True - operands rounded to bf16, one pass (~1e-2);
    exact_fp32 = True - the float32 MFMA only."""
    mfma_bf16 = False
    exact_fp32 = False
    # Re-use the kernel-layout copy of a weight across calls while the tensor's
    # (data_ptr, _version) is unchanged.  Off by default: writes through
    # `weight.data` do not bump `_version`, so a stale copy could be used.
    cache_packed_weights = False
    # Training (float32 tensors, the LFA's shapes): the forward leaves its per-tap products
    # in its workspace and the backward reads them instead of contracting x again
    # (GANET_DCN_ALGO_KEEP_Y / ganet_deform_conv_backward_y).  The workspace then lives until
    # the backward: B*H*W*kH*kW*Cout*4 bytes, kept only up to keep_y_max_bytes.
    keep_y_for_backward = True
    keep_y_max_bytes = 1 << 30


config = _Config()


def _is_channels_last(t):
    return (t.dim() == 4 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last))


_PACKED = {}          # (weight identity) -> packed device tensor
_PACKED_MAX = 32


def _packed_weight(L, weight, p):
    """Weights repacked for the kernel ([tap][Cout][Cin]); cached while the
    parameter is unchanged (inference: one pack per module, not per call)."""
    # the packed layout depends on the algorithm the library picks for `p` (plain
    # [tap][Cout][Cin], or hi/lo bf16 planes for the split float32 GEMM), so the choice and
    # the packed size are part of the key, next to the tensor's identity
    key = (weight.data_ptr(), weight._version, weight.dtype, weight.device,
           tuple(weight.shape), p.mfma_bf16, p.deformable_groups, p.algo,
           L.ganet_deform_conv_algorithm(ctypes.byref(p)),
           L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(p)),
           (p.B, p.H, p.W, p.sH, p.sW, p.x_layout))
    hit = _PACKED.get(key)
    if hit is not None:
        return hit
    nbytes = L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(p))
    if nbytes == 0:
        return None  # invalid / unsupported: let the forward call name the reason
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    stream = torch.cuda.current_stream(weight.device).cuda_stream
    rc = L.ganet_deform_conv_pack_weight(weight.data_ptr(), ctypes.byref(p),
                                         packed.data_ptr(), nbytes, stream)
    _lib.check(rc, "ganet_deform_conv_pack_weight")
    if len(_PACKED) >= _PACKED_MAX:
        _PACKED.pop(next(iter(_PACKED)))
    # keep the source tensor alive with its packed copy so data_ptr stays unique
    _PACKED[key] = (packed, weight)
    return _PACKED[key]


def deform_conv_forward(input, offset, weight, stride=(1, 1), padding=(0, 0),
                        dilation=(1, 1), groups=1, deformable_groups=1,
                        mask=None, bias=None, fuse_relu=False,
                        offset_is_points=False, mfma_bf16=None, algo=0, check=False,
                        keep_y=None):
    """Raw forward on the HIP library.  `input` NCHW (contiguous or
    channels_last) on a GPU; returns a tensor of the same memory format.
    algo: ganet_dcn_params.algo (0 = the library chooses: 5 for calls of one round of
    workgroups such as batch 1, 2 where contract-then-sample applies, else 1).
    keep_y: a dict; where the library can keep the per-tap products of the whole batch
    (GANET_DCN_ALGO_KEEP_Y) it gets keep_y["y"] = that float32 tensor (a view of the call's
    workspace) for deform_conv_backward(..., y_saved=...), otherwise it stays empty.
    check=True synchronises and raises if the algorithm left a failure word in its
    workspace (algorithms 3 and 4: ganet_deform_conv_status_offset)."""
    if not input.is_cuda:
        raise NotImplementedError
    if input.dtype not in _DTYPES:
        raise TypeError("deform_conv: unsupported dtype %s" % input.dtype)
    L = _lib.lib()
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    nhwc = _is_channels_last(input)
    if not nhwc:
        input = input.contiguous()
    weight = weight.contiguous()
    if weight.dtype != input.dtype:
        weight = weight.to(input.dtype)
    offset = offset.contiguous()
    if offset.dtype != torch.float32:
        offset = offset.float()
    if mask is not None:
        mask = mask.contiguous()
        if mask.dtype != torch.float32:
            mask = mask.float()
    if bias is not None:
        bias = bias.contiguous().to(input.dtype)
    B, Cin, H, W = input.shape
    Cout, _, kH, kW = weight.shape
    p = _lib.DcnParams()
    p.B, p.Cin, p.H, p.W, p.Cout = B, Cin, H, W, Cout
    p.kH, p.kW = kH, kW
    p.sH, p.sW = stride
    p.pH, p.pW = padding
    p.dH, p.dW = dilation
    p.groups, p.deformable_groups = groups, deformable_groups
    p.dtype = _DTYPES[input.dtype]
    p.x_layout = _lib.GANET_NHWC if nhwc else _lib.GANET_NCHW
    p.out_layout = p.x_layout
    use_bf16 = config.mfma_bf16 if mfma_bf16 is None else mfma_bf16
    if input.dtype != torch.float32:
        p.mfma_bf16 = 0
    elif use_bf16 == "exact" or (not use_bf16 and config.exact_fp32):
        p.mfma_bf16 = 2
    else:
        p.mfma_bf16 = 1 if use_bf16 else 0
    p.fuse_relu = 1 if fuse_relu else 0
    p.offset_is_points = 1 if offset_is_points else 0
    p.algo = algo
    y_off = None
    if keep_y is not None and algo in (0, 2):
        p.algo = algo | _lib.GANET_DCN_ALGO_KEEP_Y
        y_off = L.ganet_deform_conv_y_offset(ctypes.byref(p))
        if (y_off == ctypes.c_size_t(-1).value
                or B * H * W * kH * kW * Cout * 4 > config.keep_y_max_bytes):
            p.algo, y_off = algo, None
    Ho = L.ganet_dcn_out_size(H, p.pH, p.dH, kH, p.sH)
    Wo = L.ganet_dcn_out_size(W, p.pW, p.dW, kW, p.sW)
    if Ho < 1 or Wo < 1:
        raise ValueError("convolution input is too small (output would be "
                         "%dx%dx%dx%d)" % (B, Cout, Ho, Wo))
    if weight.shape[1] * groups != Cin:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d"
                           % (weight.shape[1] * groups, Cin))
    if tuple(offset.shape) != (B, deformable_groups * 2 * kH * kW, Ho, Wo):
        if offset.shape[0] != B:
            raise RuntimeError("invalid batch size of offset")
        if offset.shape[1] != deformable_groups * 2 * kH * kW:
            raise RuntimeError("invalid number of channels of offset")
        raise RuntimeError(
            "invalid spatial size of offset, expected height: %d width: %d, but "
            "got height: %d width: %d" % (Ho, Wo, offset.shape[2], offset.shape[3]))
    if mask is not None and tuple(mask.shape) != (B, deformable_groups * kH * kW, Ho, Wo):
        raise RuntimeError("invalid shape of mask")
    out = torch.empty((B, Cout, Ho, Wo), dtype=input.dtype, device=input.device,
                      memory_format=torch.channels_last if nhwc
                      else torch.contiguous_format)
    with torch.cuda.device(input.device):
        w_ptr = weight.data_ptr()
        if config.cache_packed_weights and (
                not weight.requires_grad or not torch.is_grad_enabled()):
            packed = _packed_weight(L, weight.detach(), p)
            if packed is not None:
                p.weight_is_packed = 1
                w_ptr = packed[0].data_ptr()
        ws_bytes = L.ganet_deform_conv_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=input.device)
        stream = torch.cuda.current_stream(input.device).cuda_stream
        rc = L.ganet_deform_conv_forward(
            input.data_ptr(), offset.data_ptr(),
            mask.data_ptr() if mask is not None else None, w_ptr,
            bias.data_ptr() if bias is not None else None, out.data_ptr(),
            ctypes.byref(p), ws.data_ptr(), ws_bytes, stream)
    if rc == _lib.GANET_ERR_BAD_SHAPE and mask is None and (H < kH or W < kW):
        raise RuntimeError("input image is smaller than kernel")
    _lib.check(rc, "ganet_deform_conv_forward")
    if y_off is not None:
        nbytes = B * H * W * kH * kW * Cout * 4
        keep_y["y"] = ws[y_off:y_off + nbytes].view(torch.float32)
    if check:
        off = L.ganet_deform_conv_status_offset(ctypes.byref(p))
        if off != ctypes.c_size_t(-1).value:
            word = int(ws[off:off + 4].view(torch.int32).item())   # synchronises
            if word != 0:
                raise RuntimeError("ganet_deform_conv_forward: the persistent launch (algorithm "
                                   "3 / 4) gave up waiting (status word %d); the output is "
                                   "invalid" % word)
    return out


def deform_conv_backward(input, offset, weight, grad_output, stride=(1, 1),
                         padding=(0, 0), dilation=(1, 1), groups=1, deformable_groups=1,
                         mask=None, need_input=True, need_weight=True, need_bias=False,
                         algo=0, y_saved=None, overwrite=True):
    """Raw backward on the HIP library (float32).  Returns
    (grad_input, grad_offset, grad_mask, grad_weight, grad_bias); entries that were not
    asked for (or have no meaning without a mask) are None.  grad_input has the memory
    format of `input`.  algo: 0 automatic, 1 direct (columns formed and consumed on chip),
    2 commuted (Cout-wide rows through the per-tap products; where it applies).
    y_saved: the forward's kept per-tap products (deform_conv_forward(keep_y=...)) of the
    same input, offset and weight; used where the commuted form runs, ignored elsewhere."""
    if not grad_output.is_cuda or not input.is_cuda:
        raise NotImplementedError
    if input.dtype not in _DTYPES:
        raise TypeError("deform_conv: unsupported dtype %s" % input.dtype)
    L = _lib.lib()
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    nhwc = _is_channels_last(input)
    if not nhwc:
        input = input.contiguous()
    # x, weight, grad_output and the gradients of x / weight / bias share the tensors' dtype
    # (16-bit ones are widened inside the library, gradients formed in float32 and rounded
    # once); offsets, masks and their gradients are float32 as in the forward
    weight = weight.contiguous().to(input.dtype)
    offset = offset.contiguous().float()
    if mask is not None:
        mask = mask.contiguous().float()
    go_nhwc = _is_channels_last(grad_output)
    grad_output = grad_output.to(input.dtype)
    if not go_nhwc:
        grad_output = grad_output.contiguous()
    B, Cin, H, W = input.shape
    Cout, _, kH, kW = weight.shape
    p = _lib.DcnParams()
    p.B, p.Cin, p.H, p.W, p.Cout = B, Cin, H, W, Cout
    p.kH, p.kW = kH, kW
    p.sH, p.sW = stride
    p.pH, p.pW = padding
    p.dH, p.dW = dilation
    p.groups, p.deformable_groups = groups, deformable_groups
    p.dtype = _DTYPES[input.dtype]
    p.x_layout = _lib.GANET_NHWC if nhwc else _lib.GANET_NCHW
    p.out_layout = _lib.GANET_NHWC if go_nhwc else _lib.GANET_NCHW
    # overwrite: grad_input is written, not added into, and needs no zero-fill here
    # (GANET_DCN_BWD_OVERWRITE_X); False: the reference's zero-fill-then-add contract
    p.algo = algo | (_lib.GANET_DCN_BWD_OVERWRITE_X if overwrite else 0)
    grad_input = grad_offset = grad_mask = grad_weight = grad_bias = None
    if need_input:                                 # both keep the memory format
        grad_input = torch.empty_like(input) if overwrite else torch.zeros_like(input)
        grad_offset = torch.empty_like(offset)
        if mask is not None:
            grad_mask = torch.empty_like(mask)
    if need_weight:
        grad_weight = torch.zeros_like(weight)
    if need_bias:
        grad_bias = torch.zeros(Cout, dtype=input.dtype, device=input.device)

    def ptr(t):
        return t.data_ptr() if t is not None else None

    with torch.cuda.device(input.device):
        ws_bytes = L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=input.device)
        stream = torch.cuda.current_stream(input.device).cuda_stream
        args = (input.data_ptr(), offset.data_ptr(), ptr(mask), weight.data_ptr(),
                grad_output.data_ptr(), ptr(grad_input), ptr(grad_offset), ptr(grad_mask),
                ptr(grad_weight), ptr(grad_bias), ctypes.byref(p), ws.data_ptr(),
                max(ws_bytes, 256) if ws_bytes else 0, stream)
        rc = _lib.GANET_ERR_UNSUPPORTED
        if y_saved is not None and need_input and input.dtype == torch.float32:
            assert y_saved.is_cuda and y_saved.dtype == torch.float32 and y_saved.is_contiguous() \
                and y_saved.numel() == B * H * W * kH * kW * Cout
            rc = L.ganet_deform_conv_backward_y(*args, y_saved.data_ptr())
        if rc == _lib.GANET_ERR_UNSUPPORTED:      # no saved products, or the direct form runs
            rc = L.ganet_deform_conv_backward(*args)
    _lib.check(rc, "ganet_deform_conv_backward")
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias


def _conv_out_hw(size_hw, kernel_hw, stride, padding, dilation):
    """Output height/width by the library's rule (ganet_dcn_out_size =
    deform_conv.py:98-111); 0 where Python's floor division would not be positive."""
    L = _lib.lib()
    return tuple(L.ganet_dcn_out_size(int(size_hw[d]), int(padding[d]), int(dilation[d]),
                                      int(kernel_hw[d]), int(stride[d])) for d in (0, 1))


def _check_im2col_step(im2col_step, batch):
    # the kernels do not chunk the batch; the reference's divisibility contract is kept
    # for API parity (deform_conv.py:49-51)
    assert batch % min(im2col_step, batch) == 0, 'im2col step must divide batchsize'


def _keep_holder(ctx, input, n_coords):
    """A dict for deform_conv_forward(keep_y=...) when this call's backward will form the
    gradients of the input / offsets / mask (the part of the backward that reads the per-tap
    products), None otherwise.  The tensors lead the Function's arguments: input, offset
    (, mask)."""
    if not config.keep_y_for_backward or input.dtype != torch.float32:
        return None
    if not any(ctx.needs_input_grad[:1 + n_coords]):
        return None
    return {}


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {input.dim()}'
                             'D tensor instead.')
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(input, offset, weight)
        # same order of complaints as the reference: empty output, CPU tensor, im2col step
        DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride)
        if not input.is_cuda:
            raise NotImplementedError
        _check_im2col_step(im2col_step, input.shape[0])
        kept = _keep_holder(ctx, input, 1)
        out = deform_conv_forward(input, offset, weight, ctx.stride, ctx.padding,
                                  ctx.dilation, groups, deformable_groups, keep_y=kept)
        ctx.y_saved = kept.get("y") if kept is not None else None
        return out

    @staticmethod
    def backward(ctx, grad_output):
        # deform_conv.py:60-96: gradients of input, offset and weight, None for the rest
        input, offset, weight = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError
        _check_im2col_step(ctx.im2col_step, input.shape[0])
        want = ctx.needs_input_grad
        g_in, g_off, _, g_w, _ = deform_conv_backward(
            input, offset, weight, grad_output, ctx.stride, ctx.padding, ctx.dilation,
            ctx.groups, ctx.deformable_groups, need_input=want[0] or want[1],
            need_weight=want[2], y_saved=ctx.y_saved)
        ctx.y_saved = None
        if g_off is not None:
            g_off = g_off.to(offset.dtype)
        return (g_in, g_off, g_w) + (None,) * 6

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        """(B, Cout, Hout, Wout); ValueError with the reference's text if a side is < 1."""
        hw = _conv_out_hw(input.shape[2:], weight.shape[2:], stride, padding, dilation)
        shape = (input.size(0), weight.size(0)) + hw
        if min(shape) <= 0:
            # the message shows Python's floor-division sizes, as the reference's does
            shown = shape[:2] + tuple(
                (input.size(d + 2) + 2 * padding[d]
                 - (dilation[d] * (weight.size(d + 2) - 1) + 1)) // stride[d] + 1
                for d in (0, 1))
            raise ValueError('convolution input is too small (output would be '
                             + 'x'.join(str(v) for v in shown) + ')')
        return shape


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError
        if any(t.requires_grad for t in (input, offset, mask, weight)):
            # five saved tensors like the reference, which stores a one-element stand-in
            # when there is no bias (deform_conv.py:130-137)
            ctx.save_for_backward(input, offset, mask, weight,
                                  bias if ctx.with_bias else input.new_empty(1))
        kept = _keep_holder(ctx, input, 2)
        out = deform_conv_forward(input, offset, weight, _pair(stride), _pair(padding),
                                  _pair(dilation), groups, deformable_groups, mask=mask,
                                  bias=bias, keep_y=kept)
        ctx.y_saved = kept.get("y") if kept is not None else None
        return out

    @staticmethod
    def backward(ctx, grad_output):
        # deform_conv.py:152-172
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, _bias = ctx.saved_tensors
        grads = deform_conv_backward(
            input, offset, weight, grad_output, _pair(ctx.stride), _pair(ctx.padding),
            _pair(ctx.dilation), ctx.groups, ctx.deformable_groups, mask=mask,
            need_input=True, need_weight=True, need_bias=ctx.with_bias, y_saved=ctx.y_saved)
        ctx.y_saved = None
        g_in, g_off, g_mask, g_w, g_b = grads
        g_off, g_mask = g_off.to(offset.dtype), g_mask.to(mask.dtype)
        return (g_in, g_off, g_mask, g_w, g_b if ctx.with_bias else None) + (None,) * 5

    @staticmethod
    def _infer_shape(ctx, input, weight):
        hw = _conv_out_hw(input.shape[2:4], weight.shape[2:4], _pair(ctx.stride),
                          _pair(ctx.padding), _pair(ctx.dilation))
        return (input.size(0), weight.size(0)) + hw


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


def _set_conv_attributes(m, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, deformable_groups):
    """Attribute names of the reference modules (and the two nn.Conv2d look-alikes mmcv's
    wrappers read), the `weight` parameter [Cout, Cin/groups, kH, kW]."""
    m.in_channels, m.out_channels = in_channels, out_channels
    m.kernel_size, m.stride, m.padding, m.dilation = kernel_size, stride, padding, dilation
    m.groups, m.deformable_groups = groups, deformable_groups
    m.transposed = False
    m.output_padding = _single(0)
    m.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *kernel_size))


def _uniform_fan_in(m):
    """weight ~ U(+-1/sqrt(Cin * kH * kW)) (deform_conv.py:225-230, :294-299, :423-430)."""
    bound = 1.0 / math.sqrt(m.in_channels * m.kernel_size[0] * m.kernel_size[1])
    with torch.no_grad():
        m.weight.uniform_(-bound, bound)


class _DeformConvBase(nn.Module):
    """Shared body of DeformConv / DeformConv1D (deform_conv.py:191-317)."""

    def _init(self, in_channels, out_channels, kernel_size, stride, padding,
              dilation, groups, deformable_groups, bias):
        assert not bias
        assert in_channels % groups == 0, \
            f'in_channels {in_channels} is not divisible by groups {groups}'
        assert out_channels % groups == 0, \
            f'out_channels {out_channels} is not divisible ' \
            f'by groups {groups}'
        _set_conv_attributes(self, in_channels, out_channels, kernel_size, stride, padding,
                             dilation, groups, deformable_groups)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in(self)

    def forward(self, x, offset):
        # A map smaller than the kernel is zero-padded at the bottom/right up to the kernel
        # size (offsets alike) and the result cropped back (deform_conv.py:237-253, :301-317).
        grow_h = max(self.kernel_size[0] - x.size(2), 0)
        grow_w = max(self.kernel_size[1] - x.size(3), 0)
        if grow_h or grow_w:
            x = F.pad(x, (0, grow_w, 0, grow_h)).contiguous()
            offset = F.pad(offset, (0, grow_w, 0, grow_h)).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding,
                          self.dilation, self.groups, self.deformable_groups)
        if grow_h or grow_w:
            out = out[:, :, :out.size(2) - grow_h, :out.size(3) - grow_w].contiguous()
        return out


class DeformConv(_DeformConvBase):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, deformable_groups=1,
                 bias=False):
        super(DeformConv, self).__init__()
        self._init(in_channels, out_channels, _pair(kernel_size), _pair(stride),
                   _pair(padding), _pair(dilation), groups, deformable_groups,
                   bias)


class DeformConv1D(_DeformConvBase):
    """1 x K deformable convolution, the LFA's aggregation operator
    (deform_conv.py:255-317; used at lanepoints_conv.py:104-106)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, deformable_groups=1,
                 bias=False):
        super(DeformConv1D, self).__init__()
        self._init(in_channels, out_channels, (1, kernel_size), (1, stride),
                   (0, padding), (1, dilation), groups, deformable_groups, bias)


class _OffsetPackMixin(object):
    """The `conv_offset` branch of DeformConvPack / ModulatedDeformConvPack: an ordinary
    convolution with the module's own geometry that predicts `maps_per_tap` values per
    kernel tap and deformable group, zero-initialised (deform_conv.py:339-360, :455-484);
    checkpoints older than version 2 call it `<name>_offset` (:362-377, :486-502)."""

    _version = 2

    def _add_conv_offset(self, maps_per_tap):
        taps = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * maps_per_tap * taps,
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.init_offset()

    def init_offset(self):
        for t in (self.conv_offset.weight, self.conv_offset.bias):
            nn.init.zeros_(t)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict,
                              missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get('version', None)
        if version is None or version < 2:
            for leaf in ('weight', 'bias'):
                new, old = prefix + 'conv_offset.' + leaf, prefix[:-1] + '_offset.' + leaf
                if new not in state_dict and old in state_dict:
                    state_dict[new] = state_dict.pop(old)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict,
                                      missing_keys, unexpected_keys, error_msgs)


class DeformConvPack(_OffsetPackMixin, DeformConv):
    """Deformable conv that predicts its own offsets (deform_conv.py:320-387)."""

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        self._add_conv_offset(2)          # (dy, dx) per tap

    def forward(self, x):
        return deform_conv(x, self.conv_offset(x), self.weight, self.stride, self.padding,
                           self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, deformable_groups=1,
                 bias=True):
        super(ModulatedDeformConv, self).__init__()
        # stride / padding / dilation stay as given (ints), as the reference keeps them
        _set_conv_attributes(self, in_channels, out_channels, _pair(kernel_size), stride,
                             padding, dilation, groups, deformable_groups)
        self.with_bias = bias
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in(self)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias,
                                     self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(_OffsetPackMixin, ModulatedDeformConv):

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        self._add_conv_offset(3)          # (dy, dx, mask logit) per tap

    def forward(self, x):
        # channel thirds: y offsets, x offsets, mask logits (deform_conv.py:476-484)
        planes = self.conv_offset(x)
        third = planes.size(1) // 3
        offset, logits = planes[:, :2 * third], planes[:, 2 * third:]
        return modulated_deform_conv(x, offset, logits.sigmoid(), self.weight, self.bias,
                                     self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


def register_with_mmcv(force=False):
    """Register DeformConvPack / ModulatedDeformConvPack as 'DCN' / 'DCNv2' in mmcv's
    CONV_LAYERS (deform_conv.py:320, :440).  An explicit call, not an import side effect;
    force=True replaces an existing registration.  False when mmcv is absent."""
    try:
        from mmcv.cnn import CONV_LAYERS
    except Exception:
        return False
    CONV_LAYERS.register_module('DCN', module=DeformConvPack, force=force)
    CONV_LAYERS.register_module('DCNv2', module=ModulatedDeformConvPack, force=force)
    return True
