"""ctypes binding of ganet_amd/libganet_hip.so (the C-ABI in include/ganet_hip.h).

There is no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GANET_HIP_LIB: another build of the same library (tools/variants.py, tuning runs)
LIB_PATH = os.environ.get("GANET_HIP_LIB") or os.path.join(_HERE, "libganet_hip.so")

GANET_OK = 0
GANET_ERR_NULL = -1
GANET_ERR_BAD_SHAPE = -2
GANET_ERR_UNSUPPORTED = -3
GANET_ERR_WORKSPACE = -4
GANET_ERR_HIP = -5
GANET_ERR_TOO_SMALL = -6
GANET_ERR_NOT_FINITE = -7

GANET_F32, GANET_BF16, GANET_F16 = 0, 1, 2
GANET_NCHW, GANET_NHWC = 0, 1
GANET_DCN_ALGO_KEEP_Y = 0x100
GANET_DCN_BWD_OVERWRITE_X = 0x200
DECODE_MAX_CENTERS = 2048
DECODE_COUNT_STRIDE = 8

EXPORTS = (
    "ganet_status_string", "ganet_abi_version", "ganet_dcn_out_size",
    "ganet_deform_conv_workspace_bytes", "ganet_deform_conv_forward",
    "ganet_deform_conv_packed_weight_bytes", "ganet_deform_conv_pack_weight",
    "ganet_decode_workspace_bytes", "ganet_decode_batch", "ganet_group_batch",
    "ganet_decode_pack", "ganet_deform_conv_algorithm", "ganet_deform_conv_status_offset",
    "ganet_deform_conv_y_offset", "ganet_deform_conv_backward_y",
    "ganet_deform_conv_backward_workspace_bytes", "ganet_deform_conv_backward",
    "ganet_decode_batch_noalloc", "ganet_decode_packed_workspace_bytes",
    "ganet_decode_batch_packed",
    "ganet_lanes_from_packed", "ganet_tusimple_from_lanes", "ganet_culane_text",
)


class LaneFormat(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "by_center", "downscale", "points_thr", "float32_scalars", "threads",
        "img_h", "img_w")] + [(n, ctypes.c_double) for n in (
            "crop_x0", "crop_y0", "crop_x1", "crop_y1")] + [("packed_words", ctypes.c_int64)]


class DcnParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "Cin", "H", "W", "Cout", "kH", "kW", "sH", "sW", "pH", "pW", "dH",
        "dW", "groups", "deformable_groups", "dtype", "x_layout", "out_layout",
        "mfma_bf16", "fuse_relu", "offset_is_points", "weight_is_packed",
        "algo", "debug")]


class DecodeParams(ctypes.Structure):
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("kpt_thr", ctypes.c_float), ("root_thr", ctypes.c_float),
        ("cluster_thr", ctypes.c_double), ("by_center_thr", ctypes.c_double),
        ("max_kpts", ctypes.c_int32), ("max_centers", ctypes.c_int32),
        ("max_groups", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 4)]


class DecodeOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "counts", "centers_xy", "center_gid", "kpt_xy", "kpt_align", "kpt_root",
        "kpt_gid", "group_center", "group_count")]


class GanetError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__("%s failed: %s (status %d)" % (
            where, status_string(status), status))


_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ganet_amd: %s is missing - build it with `python -m ganet_amd.build` "
                "(there is no CPU or eager fallback)" % LIB_PATH)
        # torch carries its own libamdhip64; it has to be in the process BEFORE this
        # library is, so that both bind to ONE HIP runtime.  Loaded the other way round
        # (this library first, against /opt/rocm's runtime, torch's afterwards) the
        # streams and pointers torch hands over belong to a different runtime and every
        # launch fails with GANET_ERR_HIP.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.ganet_status_string.restype = ctypes.c_char_p
        L.ganet_status_string.argtypes = [ctypes.c_int]
        L.ganet_abi_version.restype = ctypes.c_int
        L.ganet_dcn_out_size.restype = ctypes.c_int
        L.ganet_dcn_out_size.argtypes = [ctypes.c_int] * 5
        L.ganet_deform_conv_workspace_bytes.restype = ctypes.c_size_t
        L.ganet_deform_conv_workspace_bytes.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_status_offset.restype = ctypes.c_size_t
        L.ganet_deform_conv_status_offset.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_y_offset.restype = ctypes.c_size_t
        L.ganet_deform_conv_y_offset.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_algorithm.restype = ctypes.c_int
        L.ganet_deform_conv_algorithm.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_forward.restype = ctypes.c_int
        L.ganet_deform_conv_forward.argtypes = [
            ctypes.c_void_p] * 6 + [ctypes.POINTER(DcnParams), ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_deform_conv_backward_workspace_bytes.restype = ctypes.c_size_t
        L.ganet_deform_conv_backward_workspace_bytes.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_backward.restype = ctypes.c_int
        L.ganet_deform_conv_backward.argtypes = [
            ctypes.c_void_p] * 10 + [ctypes.POINTER(DcnParams), ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_deform_conv_backward_y.restype = ctypes.c_int
        L.ganet_deform_conv_backward_y.argtypes = [
            ctypes.c_void_p] * 10 + [ctypes.POINTER(DcnParams), ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.ganet_deform_conv_packed_weight_bytes.restype = ctypes.c_size_t
        L.ganet_deform_conv_packed_weight_bytes.argtypes = [ctypes.POINTER(DcnParams)]
        L.ganet_deform_conv_pack_weight.restype = ctypes.c_int
        L.ganet_deform_conv_pack_weight.argtypes = [
            ctypes.c_void_p, ctypes.POINTER(DcnParams), ctypes.c_void_p,
            ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_decode_workspace_bytes.restype = ctypes.c_size_t
        L.ganet_decode_workspace_bytes.argtypes = [ctypes.POINTER(DecodeParams)]
        L.ganet_decode_batch.restype = ctypes.c_int
        L.ganet_decode_batch.argtypes = [
            ctypes.c_void_p] * 3 + [ctypes.POINTER(DecodeParams),
                                    ctypes.POINTER(DecodeOut), ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_decode_batch_noalloc.restype = ctypes.c_int
        L.ganet_decode_batch_noalloc.argtypes = [
            ctypes.c_void_p] * 3 + [ctypes.POINTER(DecodeParams),
                                    ctypes.POINTER(DecodeOut), ctypes.c_void_p]
        L.ganet_group_batch.restype = ctypes.c_int
        L.ganet_group_batch.argtypes = [ctypes.POINTER(DecodeParams),
                                        ctypes.POINTER(DecodeOut), ctypes.c_void_p]
        L.ganet_decode_pack.restype = ctypes.c_int
        L.ganet_decode_pack.argtypes = [ctypes.POINTER(DecodeParams),
                                       ctypes.POINTER(DecodeOut), ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_decode_packed_workspace_bytes.restype = ctypes.c_size_t
        L.ganet_decode_packed_workspace_bytes.argtypes = [ctypes.POINTER(DecodeParams)]
        L.ganet_decode_batch_packed.restype = ctypes.c_int
        L.ganet_decode_batch_packed.argtypes = [
            ctypes.c_void_p] * 3 + [ctypes.POINTER(DecodeParams), ctypes.POINTER(DecodeOut),
                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_lanes_from_packed.restype = ctypes.c_int
        L.ganet_lanes_from_packed.argtypes = [
            ctypes.c_void_p, ctypes.POINTER(LaneFormat), ctypes.c_void_p, ctypes.c_size_t,
            ctypes.POINTER(ctypes.c_size_t)]
        L.ganet_tusimple_from_lanes.restype = ctypes.c_int
        L.ganet_tusimple_from_lanes.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.ganet_culane_text.restype = ctypes.c_int
        L.ganet_culane_text.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
            ctypes.c_void_p]
        if L.ganet_abi_version() != 1:
            raise RuntimeError("ganet_amd: ABI version mismatch")
        _lib = L
    return _lib


def status_string(status):
    return lib().ganet_status_string(int(status)).decode()


def check(status, where):
    if status != GANET_OK:
        raise GanetError(status, where)
