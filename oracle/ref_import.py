"""TEST INFRASTRUCTURE ONLY - container-side loader for the reference's own decode.

Loads `mmdet/models/dense_heads/ganet_head.py` and `tools/ganet/post_process.py`
from the read-only reference mount with `importlib`, after installing inert
stand-ins for the third-party imports those two files make but never use on
this path (`cv2`, `mmcv.Timer`, the mmdet HEADS registry).  Nothing of the
reference is copied: the files are executed where they lie.

This module only works where `/root/reference` exists (the build container).
It is used by `oracle/make_golden.py` to produce `tests/golden/*.npz` and by
the `not gpu` tests to cross-check the C restatement when the mount is there.
It never travels to the GPU box in a form that runs there, and the product
package (`ganet_amd/`) never imports it.
"""
import functools
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GANET_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(
        REFERENCE_ROOT, "mmdet/models/dense_heads/ganet_head.py"))


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package is not None:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@functools.lru_cache(maxsize=1)
def load_reference():
    """Return (ganet_head module, post_process module) of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not mounted at %s" % REFERENCE_ROOT)
    import torch

    sys.dont_write_bytecode = True
    saved = {k: sys.modules.get(k) for k in (
        "cv2", "mmcv", "mmdet", "mmdet.models", "mmdet.models.builder",
        "mmdet.models.dense_heads")}

    class _Registry(object):
        def register_module(self, cls=None, **kw):
            if cls is None:
                return lambda c: c
            return cls

    cv2 = types.ModuleType("cv2")
    mmcv = types.ModuleType("mmcv")
    mmcv.Timer = object
    mmdet = types.ModuleType("mmdet")
    mmdet.__path__ = []
    models = types.ModuleType("mmdet.models")
    models.__path__ = []
    builder = types.ModuleType("mmdet.models.builder")
    builder.HEADS = _Registry()
    heads = types.ModuleType("mmdet.models.dense_heads")
    heads.__path__ = []
    sys.modules.update({
        "cv2": cv2, "mmcv": mmcv, "mmdet": mmdet, "mmdet.models": models,
        "mmdet.models.builder": builder, "mmdet.models.dense_heads": heads})
    try:
        base = os.path.join(REFERENCE_ROOT, "mmdet/models/dense_heads")
        _load("mmdet.models.dense_heads.ctnet_head",
              os.path.join(base, "ctnet_head.py"), "mmdet.models.dense_heads")
        ganet_head = _load("mmdet.models.dense_heads.ganet_head",
                           os.path.join(base, "ganet_head.py"),
                           "mmdet.models.dense_heads")
        # make_coordmat defaults to device='cuda' (ganet_head.py:40) and is
        # called without a device (ganet_head.py:225): bind it to the CPU.
        ganet_head.make_coordmat = functools.partial(
            ganet_head.make_coordmat, device=torch.device("cpu"))
        post_process = _load(
            "_ganet_reference_post_process",
            os.path.join(REFERENCE_ROOT, "tools/ganet/post_process.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ganet_head, post_process


def reference_decode(heat, offset, error, thr, root_thr=1):
    """Run the reference `ktdet_decode_fast` (ganet_head.py:203-238) on CPU
    tensors shaped [1,1,H,W], [1,2,H,W], [1,2,H,W]."""
    ganet_head, _ = load_reference()
    fn = ganet_head.GANetHeadFast.ktdet_decode_fast
    return fn(None, heat, offset, error, thr=thr, root_thr=root_thr)


def reference_post_process(cpt_seeds, kpt_seeds, downscale, cluster_thr,
                           cluster_by_center_thr, group_fast=True):
    """Run the reference `PostProcessor` (post_process.py:197-262)."""
    _, pp = load_reference()
    proc = pp.PostProcessor(cluster_thr=cluster_thr,
                            cluster_by_center_thr=cluster_by_center_thr,
                            group_fast=group_fast)
    return proc((cpt_seeds, kpt_seeds), downscale)


def reference_group_points_fast(kpt_seeds, cpt_seeds, thr, by_center_thr):
    """Run the reference `group_points_fast` (post_process.py:141-194)."""
    _, pp = load_reference()
    return pp.group_points_fast(kpt_seeds, cpt_seeds, thr, by_center_thr)


@functools.lru_cache(maxsize=1)
def load_reference_formats():
    """The reference's result formatting for the two datasets, executed where it lies:
      tools/ganet/common.py                   get_line_intersection, tusimple_convert_formal
      tools/ganet/tusimple/test_dataset.py    adjust_result            (:49-84)
      tools/ganet/culane/test_dataset.py      adjust_result, out_result (:79-148)
    common.py imports nothing and is loaded whole.  The two test_dataset.py scripts import
    the whole training stack at module level, so only the named function definitions are
    compiled out of the parsed file (no text of them is kept anywhere)."""
    import ast

    if not reference_available():
        raise RuntimeError("reference tree not mounted at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    common = _load("_ganet_reference_common",
                   os.path.join(REFERENCE_ROOT, "tools/ganet/common.py"))

    def functions_of(path, names):
        with open(path, "r") as f:
            tree = ast.parse(f.read(), filename=path)
        keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        ns = {}
        exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
        return ns

    tus = functions_of(os.path.join(REFERENCE_ROOT, "tools/ganet/tusimple/test_dataset.py"),
                       ("adjust_result",))
    cul = functions_of(os.path.join(REFERENCE_ROOT, "tools/ganet/culane/test_dataset.py"),
                       ("adjust_result", "out_result"))
    return dict(get_line_intersection=common.get_line_intersection,
                tusimple_convert_formal=common.tusimple_convert_formal,
                adjust_result_tusimple=tus["adjust_result"],
                adjust_result_culane=cul["adjust_result"],
                out_result=cul["out_result"])
