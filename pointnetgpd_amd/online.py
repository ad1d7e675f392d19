"""Online grasp scoring: scene cloud + grasp candidates in, ranked grasps out, all on one GPU.

This is the per-frame loop of the reference's ROS node (apps/kinect2grasp.py:712-849) with the
Python loops removed: collect_pc for all grasps, `repeat` subsample/pad draws per grasp, one
batched network pass over G*repeat candidates, softmax/arg-max, the repeat vote and the final
descending sort of the good grasps.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import crop as _crop
from .scoring import score_candidates


@dataclass
class GraspScores:
    label: torch.Tensor     # int32 [G]   voted class (0 for grasps with too few points)
    score: torch.Tensor     # float32 [G] mean probability of the "good" class over agreeing repeats
    valid: torch.Tensor     # uint8 [G]   0 where fewer than min_points points fell inside the gripper
    counts: torch.Tensor    # int32 [G]   points inside the gripper
    order: torch.Tensor     # int64 [n_good] indices of the good grasps, best score first


def score_grasps(model, scene, grasps, limits, repeat: int = 3, seed: int = 0, sel=None,
                 min_points: int = _crop.MIN_POINTS) -> GraspScores:
    """scene float32 [S,3] (device), grasps float64 [G,5,3]; `model` a PointNetCls in eval mode."""
    P = model.num_points
    k = model.fc3.out_features
    best = 2 if k == 3 else 1                                   # apps/kinect2grasp.py:803-807

    def rank(pts, valid):
        pred, probs, _ = score_candidates(model, pts)
        label, score = _crop.vote(pred, probs, valid, repeat)
        good = torch.nonzero((label == best) & (valid != 0)).flatten()          # reads the device: the frame's one sync
        # sorted(..., key=score, reverse=True) is stable (apps/kinect2grasp.py:842-849)
        order = good[torch.sort(score[good], descending=True, stable=True).indices]
        return label, score, order

    if sel is None:
        # crop and subsample in one pass; nothing is read back until the ranking
        pts, valid, counts, overflow = _crop.crop_subsample(grasps, scene, limits, P, repeat=repeat, seed=seed,
                                                            min_points=min_points)
        label, score, order = rank(pts, valid)
        if int(overflow.item()) == 0:
            return GraspScores(label, score, valid, counts, order)
    # given draws, or a grasp with more points than the one-pass kernel holds: the three-call path
    res = _crop.collect_pc(grasps, scene, limits, return_points=False)
    pts, valid = _crop.subsample(res, P, repeat=repeat, sel=sel, seed=seed, min_points=min_points)
    label, score, order = rank(pts, valid)
    return GraspScores(label, score, valid, res.counts, order)
