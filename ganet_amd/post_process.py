"""Host-side mirror of tools/ganet/post_process.py::PostProcessor (lines 197-262).

Same constructor and `__call__(output, downscale) -> (lanes, cluster_centers)`
contract and the same Python structures / numeric types:
  lanes           list of dict(id_class=<group index>, points=[[f32, f32]...],
                  centers=[[f64, f64]...])  for groups with more than one point
  cluster_centers list of dict(id_class=i, center=[int64, int64])
The grouping itself (group_points_fast, post_process.py:141-194) runs on the
GPU (csrc/decode.hip); this file only formats its result
(lane_post_process, post_process.py:216-242).  The dict-seed `group_points`
path (group_fast=False) is out of scope: every shipped config sets
group_fast=True (configs/*/test_common_s*.py:7), and the path cannot run in the
reference either - group_points reads item['align'] / ['score'] / ['center']
(post_process.py:128) while the decode that would feed it returns tuples
(ganet_head.py:191), on a multi-offset branch no shipped head takes (:302).
"""
import numpy as np

from . import decode as _decode


def lanes_from_image(img, downscale):
    """lane_post_process (post_process.py:216-242) from one decoded + grouped
    image (a dict from DecodeResult.image)."""
    gid = img["kpt_gid"]
    align = np.asarray(img["align"], dtype=np.float32)
    root = np.asarray(img["root"], dtype=np.float32).astype(np.float64)
    lanes = []
    n_groups = img["n_groups"]
    if len(gid):
        order = np.argsort(gid, kind="stable")
        sorted_gid = gid[order]
        starts = np.searchsorted(sorted_gid, np.arange(n_groups), side="left")
        ends = np.searchsorted(sorted_gid, np.arange(n_groups), side="right")
    else:
        order = np.zeros(0, np.int64)
        starts = ends = np.zeros(n_groups, np.int64)
    for g in range(n_groups):
        idx = order[starts[g]:ends[g]]
        if len(idx) > 1:
            pts = align[idx] * downscale          # float32 * python int -> float32
            ctr = root[idx] * downscale           # float64
            lanes.append(dict(
                id_class=g,
                points=[[p[0], p[1]] for p in pts],
                centers=[[c[0], c[1]] for c in ctr]))
    cluster_centers = []
    for i, c in enumerate(np.asarray(img["group_center"], dtype=np.int64)):
        cluster_centers.append(dict(id_class=i,
                                    center=[c[0] * downscale, c[1] * downscale]))
    return lanes, cluster_centers


class PostProcessor(object):

    def __init__(self,
                 min_points=5,
                 hm_downscale=16,
                 mask_downscale=8,
                 use_offset=True,
                 cluster_thr=4,
                 cluster_by_center_thr=None,
                 group_fast=False,
                 **kwargs):
        self.min_points = min_points
        self.hm_downscale = hm_downscale
        self.mask_downscale = mask_downscale
        self.cluster_thr = cluster_thr
        self.cluster_by_center_thr = cluster_by_center_thr
        self.group_fast = group_fast

    def lane_post_process(self, img, downscale):
        return lanes_from_image(img, downscale)

    def __call__(self, output, downscale):
        if self.group_fast is not True:
            raise NotImplementedError(
                "ganet_amd.PostProcessor: only group_fast=True is built (the "
                "dict-seed group_points path is unused by every shipped config)")
        if isinstance(output, _decode.DecodeResult):
            res = output
            if not res.grouped:
                res = _decode.group_on_device(res, self.cluster_thr,
                                              self.cluster_by_center_thr)
            if res.B != 1:
                return [lanes_from_image(res.image(b), downscale)
                        for b in range(res.B)]
            return lanes_from_image(res.image(0), downscale)
        output = list(output)
        cpt_seeds, kpt_seeds = output[0], output[1]
        res = getattr(kpt_seeds, "ganet_result", None)
        if res is not None and res.buffers is not None:
            res = _decode.group_on_device(res, self.cluster_thr,
                                          self.cluster_by_center_thr)
        else:
            res = _decode.group_seeds_on_device(cpt_seeds, kpt_seeds,
                                                self.cluster_thr,
                                                self.cluster_by_center_thr)
        return lanes_from_image(res.image(0), downscale)
