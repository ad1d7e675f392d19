"""CPU restatement of the training-data crop.  TEST INFRASTRUCTURE ONLY: imported by tests/, never
by the product path.

collect_pc_ (model/dataset.py:767-924, the `their_way = False` branch, projection off) and the
resampling / labelling of PointGraspOneViewDataset.__getitem__ (:1098-1131), per grasp record, with
the reference's numpy calls.  model/dataset.py imports pcl, dexnet and autolab_core at module level (none
present in this image) and reads YCB data from disk, and the reference holds no fixture for it.
PINNED as far as the reference can run here: frame_of_record and the gripper-frame coordinates equal, bit
for bit, what the reference's own statements :776-843 return (tests/golden/make_dataset_golden.py takes them out
of the method's syntax tree; tests/golden/dataset_frames_reference.npz).  The six comparisons are those of the
online crop, pinned by its own reference run; getitem_tail equals a run of the statements :1098-1131 with
numpy's generator seeded alike (tests/golden/dataset_tail_reference.npz).  PARITY UNPINNED: the corner table of
dex-net's get_hand_points (an argument here).  ags.gripper.hand_depth is an argument.
"""
import numpy as np


def frame_of_record(grasp, transform, hand_depth):
    """(bottom_ [3], matrix [3,3]) of :776-835 for one grasp record, float64."""
    center = grasp[0:3]
    axis = grasp[3:6]
    angle = grasp[7]
    axis = axis / np.linalg.norm(axis)
    binormal = axis
    cos_t = np.cos(angle)
    sin_t = np.sin(angle)
    R1 = np.c_[[cos_t, 0, sin_t], [0, 1, 0], [-sin_t, 0, cos_t]]
    axis_y = axis
    axis_x = np.array([axis_y[1], -axis_y[0], 0])
    if np.linalg.norm(axis_x) == 0:
        axis_x = np.array([1, 0, 0])
    axis_x = axis_x / np.linalg.norm(axis_x)
    axis_y = axis_y / np.linalg.norm(axis_y)
    axis_z = np.cross(axis_x, axis_y)
    R2 = np.c_[axis_x, np.c_[axis_y, axis_z]]
    approach = R2.dot(R1)[:, 0]
    approach = approach / np.linalg.norm(approach)
    minor_normal = np.cross(axis, approach)
    bottom = center - hand_depth * approach
    bottom_ = (np.dot(transform, np.array([bottom[0], bottom[1], bottom[2], 1]).T))[:3]
    binormal_ = (np.dot(transform, np.array([binormal[0], binormal[1], binormal[2], 0])))[:3].reshape(3, 1)
    approach_ = (np.dot(transform, np.array([approach[0], approach[1], approach[2], 0])))[:3].reshape(3, 1)
    minor_normal_ = (np.dot(transform, np.array([minor_normal[0], minor_normal[1], minor_normal[2], 0])))[:3].reshape(3, 1)
    matrix = np.hstack([approach_, binormal_, minor_normal_]).T
    return bottom_, matrix


def frames_of_records(grasps, transform, hand_depth):
    """[G,12] float64: bottom_ followed by the rows of `matrix`, the layout pngpd_crop_count_frames takes."""
    out = np.empty((len(grasps), 12), dtype=np.float64)
    for g, rec in enumerate(np.asarray(grasps, dtype=np.float64)):
        b, m = frame_of_record(rec, np.asarray(transform, dtype=np.float64), hand_depth)
        out[g, :3] = b
        out[g, 3:] = m.reshape(9)
    return out


def collect_pc_(grasp, pc, transform, hand_depth, hand_points, min_point_limit=50):
    """Returns (in_ind, pc_c2g[in_ind]) or (in_ind, None) when fewer than min_point_limit points (:918)."""
    bottom_, matrix = frame_of_record(grasp, transform, hand_depth)
    pc_c2g = (np.dot(matrix, (pc - bottom_).T)).T                       # :843
    s1, s2, s4, s8 = hand_points[1], hand_points[2], hand_points[4], hand_points[8]
    a1 = s1[1] < pc_c2g[:, 1]
    a2 = s2[1] > pc_c2g[:, 1]
    a3 = s1[2] > pc_c2g[:, 2]
    a4 = s4[2] < pc_c2g[:, 2]
    a5 = s4[0] > pc_c2g[:, 0]
    a6 = s8[0] < pc_c2g[:, 0]
    a = np.vstack([a1, a2, a3, a4, a5, a6])
    in_ind = np.where(np.sum(a, axis=0) == len(a))[0]
    if len(in_ind) < min_point_limit:
        return in_ind, None
    return in_ind, pc_c2g[in_ind]


def getitem_tail(grasp, grasp_pc, grasp_points_num, thresh_good, thresh_bad, choice):
    """:1098-1131 after collect_pc_.  `choice(n, size, replace)` stands for np.random.choice.
    Returns (grasp_pc [3,N] float64, label) or None."""
    if grasp_pc is None:
        return None
    level_score, refine_score = grasp[-2:]
    if len(grasp_pc) > grasp_points_num:
        grasp_pc = grasp_pc[choice(len(grasp_pc), grasp_points_num, False)].T
    else:
        grasp_pc = grasp_pc[choice(len(grasp_pc), grasp_points_num, True)].T
    score = level_score + refine_score * 0.01
    if score >= thresh_bad:
        label = 0
    elif score <= thresh_good:
        label = 1
    else:
        return None
    return grasp_pc, label
