"""TEST INFRASTRUCTURE ONLY - CPU restatement of the online in-gripper crop and the voting loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates (float64 numpy, like the reference):
  * check_collision_square  apps/kinect2grasp.py:299-344  (way="p_open", use_dataset_py=False)
  * collect_pc              apps/kinect2grasp.py:377-438  (grasp rows 4,1,2,3 of a [G,5,3] array)
  * subsample / pad         apps/kinect2grasp.py:746-774  (>= P: without replacement, else with)
  * repeat / vote           apps/kinect2grasp.py:764, 799-817

Pinning.  apps/kinect2grasp.py cannot be imported (rospy, dexnet, pcl at import time), but
check_collision_square itself needs only numpy: tests/golden/make_crop_golden.py (build container only) takes
that one function definition out of the reference file's syntax tree, runs it in collect_pc's order and stores
its outputs as tests/golden/crop_*.npz.  tests/test_crop_oracle.py checks every form below against those files
bit for bit: `crop_numpy` keeps the reference's numpy calls (np.linalg.norm, np.dot of a 3x3 by 3xS matrix);
`crop_pinned` spells out the operation order that those calls execute on the build container's numpy/OpenBLAS
(a k-ordered fused-multiply-add chain) with no library call left; oracle/crop_oracle.c is the same pinned
order in C (used for large inputs).  `subsample` and the score half of `vote` equal a run of the reference's own
statements (:766-774, :803-817; tests/golden/crop_loop_reference.npz); the vote's `mode(predict)[0][0]` is not
evaluable under this image's scipy, its rule (smallest of the most frequent classes) is restated, unpinned.

The gripper corner table comes from dex-net's get_hand_points (an un-vendored dependency,
github.com/Hymwgk/dex-net, no pinned revision); `hand_points` restates its published construction
and is marked parity-unpinned.  All crop functions take the six box limits explicitly.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

MIN_POINTS = 20  # minimal_points_send_to_point_net, apps/kinect2grasp.py:746


def hand_points(hand_depth, hand_outer_diameter, finger_width, hand_height):
    """Corner table p[21,3] of the gripper model at the origin with approach +x, binormal +y.

    PARITY UNPINNED: restated from the published dex-net GpgGraspSamplerPcl.get_hand_points and
    the corner numbering drawn in the reference's data/*.png; only rows 1, 2, 4, 8 are used here.
    """
    hd, hod, fw, hh = (float(v) for v in (hand_depth, hand_outer_diameter, finger_width, hand_height))
    ow = hod - 2.0 * fw
    a, b, m = np.eye(3)
    c = np.zeros(3)
    p5_6 = m * hh * 0.5 + c
    p7_8 = -m * hh * 0.5 + c
    p5, p6 = -b * ow * 0.5 + p5_6, b * ow * 0.5 + p5_6
    p7, p8 = b * ow * 0.5 + p7_8, -b * ow * 0.5 + p7_8
    p1, p2, p3, p4 = (a * hd + q for q in (p5, p6, p7, p8))
    p9, p10, p11, p12 = -b * fw + p1, -b * fw + p4, -b * fw + p5, -b * fw + p8
    p13, p14, p15, p16 = b * fw + p2, b * fw + p3, b * fw + p6, b * fw + p7
    p17, p18, p19, p20 = -a * hh + p11, -a * hh + p15, -a * hh + p16, -a * hh + p12
    return np.vstack([c, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13, p14, p15, p16, p17, p18, p19, p20])


def box_limits(p):
    """(ylo, yhi, zlo, zhi, xlo, xhi) of the open region, from corners 1, 2, 4, 8
    (apps/kinect2grasp.py:323, 332-337)."""
    s1, s2, s4, s8 = p[1], p[2], p[4], p[8]
    return np.array([s1[1], s2[1], s4[2], s1[2], s8[0], s4[0]], dtype=np.float64)


def split_grasps(grasps):
    """[G,5,3] (or anything reshapeable to it) -> centre, approach, binormal, minor (rows 4,1,2,3)."""
    g = np.asarray(grasps, dtype=np.float64).reshape(-1, 5, 3)
    return g[:, 4, :], g[:, 1, :], g[:, 2, :], g[:, 3, :]


def frame_numpy(approach, binormal, minor):
    rows = []
    for v in (approach, binormal, minor):
        v = np.asarray(v, dtype=np.float64).reshape(1, 3)
        rows.append(v / np.linalg.norm(v))
    return np.vstack(rows)                     # rows = unit axes = transpose of hstack of columns


def crop_one_numpy(centre, approach, binormal, minor, scene, limits):
    """One grasp: (ascending indices int64, gripper-frame coordinates float64 [S,3])."""
    R = frame_numpy(approach, binormal, minor)
    d = scene - np.asarray(centre, dtype=np.float64).reshape(1, 3)      # float32 scene promoted
    pg = np.dot(R, d.T).T
    ylo, yhi, zlo, zhi, xlo, xhi = limits
    inside = ((ylo < pg[:, 1]) & (yhi > pg[:, 1]) & (zhi > pg[:, 2]) & (zlo < pg[:, 2])
              & (xhi > pg[:, 0]) & (xlo < pg[:, 0]))
    return np.where(inside)[0], pg


def crop_numpy(grasps, scene, limits):
    """collect_pc: per grasp the index array and the cropped points in the gripper frame."""
    centre, a, b, m = split_grasps(grasps)
    idx, pts = [], []
    for g in range(len(centre)):
        i, pg = crop_one_numpy(centre[g], a[g], b[g], m[g], scene, limits)
        idx.append(i)
        pts.append(pg[i])
    return idx, pts


def _fma(a, b, c):
    """Exact fused multiply-add on float64 arrays via the C helper (numpy has none)."""
    lib = _c_lib()
    a, b, c = np.broadcast_arrays(np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(c, np.float64))
    a, b, c = (np.ascontiguousarray(v) for v in (a, b, c))
    out = np.empty_like(a)
    P = ctypes.POINTER(ctypes.c_double)
    lib.oracle_fma(a.ctypes.data_as(P), b.ctypes.data_as(P), c.ctypes.data_as(P), out.ctypes.data_as(P),
                   ctypes.c_long(a.size))
    return out


def frame_pinned(approach, binormal, minor):
    """Unit axes with the operation order spelt out: n = sqrt(fma(z,z,fma(y,y,x*x))), v / n."""
    rows = []
    for v in (approach, binormal, minor):
        v = np.asarray(v, dtype=np.float64).reshape(3)
        n2 = _fma(v[2], v[2], _fma(v[1], v[1], v[0] * v[0]))
        rows.append(v / np.sqrt(n2))
    return np.vstack(rows)


def crop_one_pinned(centre, approach, binormal, minor, scene, limits):
    R = frame_pinned(approach, binormal, minor)
    d = scene.astype(np.float64) - np.asarray(centre, dtype=np.float64).reshape(1, 3)
    pg = np.empty_like(d)
    for r in range(3):
        pg[:, r] = _fma(R[r, 2], d[:, 2], _fma(R[r, 1], d[:, 1], R[r, 0] * d[:, 0]))
    ylo, yhi, zlo, zhi, xlo, xhi = limits
    inside = ((ylo < pg[:, 1]) & (yhi > pg[:, 1]) & (zhi > pg[:, 2]) & (zlo < pg[:, 2])
              & (xhi > pg[:, 0]) & (xlo < pg[:, 0]))
    return np.where(inside)[0], pg


# ---------------------------------------------------------------------------------------------
# plain-C twin (oracle/crop_oracle.c), same pinned order; used for large scenes and as CPU baseline
# ---------------------------------------------------------------------------------------------
_LIB = None


def _c_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libcrop_oracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run python -c 'import __graft_entry__ as g; g.build()'")
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_crop_one.restype = ctypes.c_long
    return _LIB


def crop_c(grasps, scene, limits):
    """Same result as crop_numpy, computed by the C restatement."""
    lib = _c_lib()
    centre, a, b, m = split_grasps(grasps)
    scene = np.ascontiguousarray(scene, dtype=np.float32)
    S = scene.shape[0]
    lim = np.ascontiguousarray(limits, dtype=np.float64)
    idx_buf = np.empty(S, dtype=np.int64)
    pts_buf = np.empty((S, 3), dtype=np.float64)
    PD, PF, PL = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
    idx, pts = [], []
    for g in range(len(centre)):
        frame = np.ascontiguousarray(np.concatenate([centre[g], a[g], b[g], m[g]]), dtype=np.float64)
        n = lib.oracle_crop_one(scene.ctypes.data_as(PF), ctypes.c_long(S), frame.ctypes.data_as(PD),
                                lim.ctypes.data_as(PD), idx_buf.ctypes.data_as(PL), pts_buf.ctypes.data_as(PD))
        idx.append(idx_buf[:n].copy())
        pts.append(pts_buf[:n].copy())
    return idx, pts


# ---------------------------------------------------------------------------------------------
# subsample / pad and the repeat vote
# ---------------------------------------------------------------------------------------------
def subsample(points_g, sel):
    """apps/kinect2grasp.py:766-774 with the random draw made explicit: `sel` is the array that
    np.random.choice(len(points_g), P, replace=len(points_g) < P) returned."""
    return points_g[np.asarray(sel)]


def vote(preds, probs, k):
    """apps/kinect2grasp.py:799-817 for one grasp: preds [R] ints, probs [R,1,k] float32.
    Returns (voted label, mean probability of the 'best' class over the agreeing repeats)."""
    preds = [int(p) for p in preds]
    values, counts = np.unique(preds, return_counts=True)
    label = int(values[np.argmax(counts)])            # scipy.stats.mode: smallest of the most common
    best = 2 if k == 3 else 1
    grasp_score = np.array(probs)
    agree = np.where(np.array(preds) == label)
    return label, np.mean(grasp_score[agree][:, 0, best])
