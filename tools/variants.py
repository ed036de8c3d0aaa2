#!/usr/bin/env python3
"""Builds tuning variants of the library: tools/variants.py name:-DFLAG=1,-DX=2 ...
Each becomes variants/lib_<name>.so (git-ignored, travels to the GPU box); run one with
GANET_HIP_LIB=variants/lib_<name>.so python bench.py ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ganet_amd import build as B  # noqa: E402

OUT = os.path.join(ROOT, "variants")
# files a variant flag can touch (VARIANT_FILES=a.hip,b.hip to change)
FILES = os.environ.get("VARIANT_FILES", "deform_conv_cts.hip").split(",")


def build(spec):
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    d = os.path.join(OUT, "obj_" + name)
    os.makedirs(d, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        o = os.path.join(d, src.replace(".hip", ".o"))
        objs.append(o)
        # only the files a flag can touch are rebuilt per variant
        base = os.path.join(B.OBJ_DIR, src.replace(".hip", ".o"))
        if src in FILES or not os.path.exists(base):
            with open(os.path.join(d, src + ".log"), "w") as log:
                subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.COMMON + flags + B.PER_FILE.get(src, []) +
                                      ["-c", os.path.join(B.CSRC, src), "-o", o], stderr=log)
        else:
            objs[-1] = base
    lib = os.path.join(OUT, "lib_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    B.build_lib()
    with ThreadPoolExecutor(4) as ex:
        for lib in ex.map(build, sys.argv[1:]):
            print(lib)
