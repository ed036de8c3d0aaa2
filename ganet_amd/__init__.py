"""ganet_amd - MI355X (gfx950) implementation of GANet's hot path.

  ganet_amd.dcn           DeformConv / DeformConv1D / DeformConvPack /
                          ModulatedDeformConv* (mirror of mmdet/ops/dcn)
  ganet_amd.decode        ktdet_decode_fast, decode_batch (mirror of
                          GANetHeadFast.ktdet_decode_fast)
  ganet_amd.post_process  PostProcessor (mirror of tools/ganet/post_process.py)
  ganet_amd.formats       adjust_result, tusimple_convert_formal, out_result (mirror of
                          the per-image formatting in tools/ganet/{tusimple,culane}/
                          test_dataset.py and tools/ganet/common.py)
  ganet_amd.lfa           the LFA tail: offsets -> DeformConv1D -> ReLU, fused
  ganet_amd.synth         seeded synthetic inputs of the hot path's shapes

Everything computes in hand-written HIP kernels behind the C-ABI declared in
include/ganet_hip.h (ganet_amd/libganet_hip.so).  There is no CPU fallback:
using an operator without the built library, or on CPU tensors, raises.
"""
from . import _lib  # noqa: F401

__all__ = ["dcn", "decode", "post_process", "formats", "lfa", "synth", "library_path"]


def library_path():
    return _lib.LIB_PATH
