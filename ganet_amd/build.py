"""Builds ganet_amd/libganet_hip.so (gfx950) from ganet_amd/csrc/*.hip with hipcc.

In-tree build: the .so sits next to this file, is git-ignored and travels to the
GPU box with the source snapshot.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
_PRODUCT_LIB = os.path.join(HERE, "libganet_hip.so")
_PRODUCT_OBJ = os.path.join(HERE, "csrc", "build")
LIB, OBJ_DIR = _PRODUCT_LIB, _PRODUCT_OBJ

ARCH = "gfx950"
COMMON = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
          "-Wall", "-Wno-unused-function"]
# decode.hip must round like numpy / CPython: no fused multiply-add contraction.
PER_FILE = {"decode.hip": ["-ffp-contract=off"], "lane_format.hip": ["-ffp-contract=off"]}
SOURCES = ["api.hip", "decode.hip", "deform_conv.hip", "deform_conv_cts.hip", "deform_conv_stream.hip", "deform_conv_ring.hip",
           "deform_conv_bwd.hip", "lane_format.hip"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _variant_flags():
    """Extra -D flags of the profiling / debug builds asked for through the environment."""
    extra = ["-DGANET_ABLATE"] if os.environ.get("GANET_ABLATE") else []  # profiling build
    extra += os.environ.get("GANET_RING_DEFS", "").split()      # -DGANET_RING_... (ablations)
    extra += os.environ.get("GANET_DECODE_DEFS", "").split()    # -DGANET_DECODE_STAMPS
    extra += os.environ.get("GANET_TAPS_DEFS", "").split()      # -DGANET_TAPS_STAMPS
    if os.environ.get("GANET_STREAM_DEBUG"):
        extra.append("-DGANET_STREAM_DEBUG")
        extra += os.environ.get("GANET_STREAM_DEFS", "").split()
    return extra


def build_lib(force=False, verbose=False):
    """The product library; with variant flags in the environment (GANET_ABLATE, GANET_RING_DEFS,
    GANET_DECODE_DEFS, GANET_TAPS_DEFS, GANET_STREAM_DEBUG) a SEPARATE library libganet_hip_<tag>.so with objects of
    its own, so that a debug or ablation build can never be picked up as the product library
    afterwards (load it with GANET_HIP_LIB=<path>)."""
    global LIB, OBJ_DIR
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = _variant_flags()
    LIB, OBJ_DIR = _PRODUCT_LIB, _PRODUCT_OBJ
    if extra:
        import hashlib
        tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8]
        LIB = os.path.join(HERE, "libganet_hip_%s.so" % tag)
        OBJ_DIR = os.path.join(HERE, "csrc", "build_" + tag)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(ROOT, "include", "ganet_hip.h"), os.path.join(CSRC, "dcn_device.h"),
               os.path.join(CSRC, "common.h")]
    srcs = [os.path.join(CSRC, src) for src in SOURCES]
    if not force and os.path.exists(LIB) and \
            not any(_newer(f, LIB) for f in srcs + headers):
        # the library is newer than everything it is built from: on the GPU box the
        # objects do not travel (.gpurunignore) but the library does, and is used as it is
        return LIB
    objs, relink = [], force or not os.path.exists(LIB)
    # flags named after one source file only rebuild that file; the rest are the product's objects
    only = {"-DGANET_RING": "deform_conv_ring.hip", "-DGANET_STREAM": "deform_conv_stream.hip",
            "-DGANET_DECODE": "decode.hip", "-DGANET_TAPS": "deform_conv.hip"}
    touched = {v for f in extra for k, v in only.items() if f.startswith(k)}
    narrow = bool(extra) and all(any(f.startswith(k) for k in only) for f in extra)
    if narrow:
        build_lib_product = [os.path.join(_PRODUCT_OBJ, src.replace(".hip", ".o")) for src in SOURCES]
        if not all(os.path.exists(o) for o in build_lib_product):
            raise RuntimeError("build the product library first (python -m ganet_amd.build)")
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if narrow and src not in touched:
            objs.append(os.path.join(_PRODUCT_OBJ, src.replace(".hip", ".o")))
            continue
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc] + COMMON + extra + PER_FILE.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            relink = True
    if relink:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
