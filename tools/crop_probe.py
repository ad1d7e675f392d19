"""Time of the one-pass crop + subsample stage (pngpd_crop_subsample) at the online sizes, with its output checked
against the three-call path.  python tools/crop_probe.py [S] [G] [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnetgpd_amd import crop as hc, synth  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    P, R = 750, 3
    limits = hc.box_limits_from_gripper(synth.GRIPPER_BOX[0], synth.GRIPPER_BOX[1] + 0.04, 0.02, synth.GRIPPER_BOX[2])
    scene = torch.from_numpy(synth.scene_points(0, S, 0.5)).cuda()
    grasps = torch.from_numpy(synth.random_grasps(1000, G, 0.5, 0.06)).cuda()
    res = hc.collect_pc(grasps, scene, limits, return_points=False)
    ref, valid_ref = hc.subsample(res, P, repeat=R, seed=0)
    pts, valid, counts, overflow = hc.crop_subsample(grasps, scene, limits, P, repeat=R, seed=0)
    same = bool(torch.equal(pts, ref) and torch.equal(valid, valid_ref) and torch.equal(counts, res.counts))
    for _ in range(10):
        hc.crop_subsample(grasps, scene, limits, P, repeat=R, seed=0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        hc.crop_subsample(grasps, scene, limits, P, repeat=R, seed=0)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    print(f"S={S} G={G}: crop+subsample median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f}  max {ms[-1]:.4f}  "
          f"equal_to_three_calls={same} overflow={int(overflow.item())} mean_cnt={float(counts.float().mean()):.1f}")
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
