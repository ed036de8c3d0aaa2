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
LIB = os.path.join(HERE, "libganet_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "build")

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


def build_lib(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DGANET_ABLATE"] if os.environ.get("GANET_ABLATE") else []  # profiling build
    if os.environ.get("GANET_RING_DEBUG"):
        extra.append("-DGANET_RING_DEBUG")
    if os.environ.get("GANET_STREAM_DEBUG"):
        extra.append("-DGANET_STREAM_DEBUG")
        extra += os.environ.get("GANET_STREAM_DEFS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(ROOT, "include", "ganet_hip.h"), os.path.join(CSRC, "dcn_device.h"),
               os.path.join(CSRC, "common.h")]
    srcs = [os.path.join(CSRC, src) for src in SOURCES]
    if not force and not extra and os.path.exists(LIB) and \
            not any(_newer(f, LIB) for f in srcs + headers):
        # the library is newer than everything it is built from: on the GPU box the
        # objects do not travel (.gpurunignore) but the library does, and is used as it is
        return LIB
    objs, relink = [], force or not os.path.exists(LIB)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
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
    build_lib(force="--force" in sys.argv, verbose=True)
    print(LIB)
