"""Golden vectors for the in-gripper crop, produced by the numpy restatement that keeps the
reference's own numpy calls (oracle.crop_oracle.crop_numpy; apps/kinect2grasp.py cannot be
imported).  Run in the build container:  python tests/golden/make_crop_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import crop_oracle as co  # noqa: E402
from pointnetgpd_amd import synth  # noqa: E402

# dyadic gripper so that points can sit EXACTLY on the six planes
HAND = dict(hand_depth=0.0625, hand_outer_diameter=0.125, finger_width=0.015625, hand_height=0.03125)


def adversarial_case():
    """Axis-aligned grasps with float32-exact centres plus points exactly on / one ulp inside each plane."""
    limits = co.box_limits(co.hand_points(**HAND))
    ylo, yhi, zlo, zhi, xlo, xhi = limits
    centre = np.array([0.25, 0.125, 0.0625])
    frames = [np.eye(3), np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]]), np.array([[0, 0, 2.0], [0, 4.0, 0], [-0.5, 0, 0]])]
    grasps = np.zeros((len(frames), 5, 3))
    for g, F in enumerate(frames):
        grasps[g, 4] = centre
        grasps[g, 1:4] = F            # rows: approach, binormal, minor
    mid = np.array([(xlo + xhi) / 2, (ylo + yhi) / 2, (zlo + zhi) / 2])
    local = [mid]
    for axis, (lo, hi) in enumerate(((xlo, xhi), (ylo, yhi), (zlo, zhi))):
        for v in (lo, hi):
            for eps in (0.0, 1.0, -1.0):      # on the plane, one float32 ulp either side
                q = mid.copy()
                q[axis] = v
                w = (centre + q).astype(np.float32)
                if eps:
                    w[axis] = np.nextafter(w[axis], np.float32(np.inf if eps > 0 else -np.inf))
                local.append(w.astype(np.float64) - centre)
    pts = []
    for F in frames:                       # express the local points in the world for every frame
        A = F / np.linalg.norm(F, axis=1, keepdims=True)
        pts.append((np.array(local) @ A) + centre)
    scene = np.vstack(pts).astype(np.float32)
    return grasps, scene, limits


def main():
    S, G, seed, cube = 20000, 32, 51, 0.25
    limits = co.box_limits(co.hand_points(hand_depth=0.06, hand_outer_diameter=0.12, finger_width=0.02, hand_height=0.02))
    scene = synth.scene_points(seed, S, cube)
    grasps = synth.random_grasps(seed, G, cube, 0.03)
    idx, pts = co.crop_numpy(grasps, scene, limits)
    counts = np.array([len(i) for i in idx])
    print("random case: counts min/mean/max", counts.min(), counts.mean(), counts.max())
    digest = np.array([hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() for p in pts])
    np.savez_compressed(os.path.join(HERE, "crop_random.npz"), meta=np.array([S, G, seed]), cube=cube, limits=limits,
                        counts=counts, idx=np.concatenate(idx).astype(np.int32), pts_head=np.concatenate(pts[:6]),
                        pts_sha256=digest)

    grasps, scene, limits = adversarial_case()
    idx, pts = co.crop_numpy(grasps, scene, limits)
    print("adversarial case: scene", scene.shape, "counts", [len(i) for i in idx])
    np.savez_compressed(os.path.join(HERE, "crop_planes.npz"), grasps=grasps, scene=scene, limits=limits,
                        counts=np.array([len(i) for i in idx]), idx=np.concatenate(idx).astype(np.int32),
                        pts=np.concatenate(pts))


if __name__ == "__main__":
    main()
