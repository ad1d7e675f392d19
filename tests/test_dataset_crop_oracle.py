"""CPU checks of the training-data crop restatement (oracle/dataset_crop_oracle.py) on a case that
can be verified by hand, and of the host-side label helper of the product mirror."""
import numpy as np

from oracle import crop_oracle as co
from oracle import dataset_crop_oracle as do
from pointnetgpd_amd import dataset_crop as dc
from pointnetgpd_amd import synth


def _hand_points():
    return co.hand_points(0.06, 0.10, 0.01, 0.02)


def test_axis_aligned_grasp_identity_transform():
    # closing axis +y, angle 0: axis_x = (1,0,0) -> approach = +x, minor = y x x = -z ... the frame is
    # (x, y, -z) up to sign, origin = centre - hand_depth * x.  Points are tested in that frame.
    hp = _hand_points()
    grasp = np.array([0.1, 0.0, 0.0, 0.0, 2.0, 0.0, 0.05, 0.0, 0.0, 0.0])
    pc = np.array([[0.07, 0.0, 0.0],      # 3 cm in front of the bottom: inside
                   [0.03, 0.0, 0.0],      # behind the bottom (x_g = -0.01): outside
                   [0.07, 0.05, 0.0],     # beyond the finger gap: outside
                   [0.07, 0.0, 0.02]], dtype=np.float32)   # above the hand height: outside
    ind, pts = do.collect_pc_(grasp, pc, np.eye(4), 0.06, hp, min_point_limit=1)
    np.testing.assert_array_equal(ind, [0])
    np.testing.assert_allclose(pts[0], [0.03, 0.0, 0.0], atol=1e-9)


def test_frames_are_rigid_images_of_an_orthonormal_frame():
    """The frame of a record, independent of how it is computed: orthonormal rows with the closing axis second and
    right-handed; the origin lies hand_depth behind the contact centre along the approach row, in the view."""
    grasps, T, pc = synth.dataset_view(3, 2000, 12)
    frames = do.frames_of_records(grasps, T, 0.06)
    for g in range(len(grasps)):
        m = frames[g, 3:].reshape(3, 3)
        np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-14)          # T is rigid here
        np.testing.assert_allclose(np.cross(m[0], m[1]), -m[2], atol=1e-14)  # minor = binormal x approach
        axis = grasps[g, 3:6] / np.linalg.norm(grasps[g, 3:6])
        np.testing.assert_allclose(m[1], T[:3, :3] @ axis, atol=1e-14)
        centre_view = T[:3, :3] @ grasps[g, :3] + T[:3, 3]
        np.testing.assert_allclose(frames[g, :3], centre_view - 0.06 * m[0], atol=1e-14)
    # the record with its axis along z takes the (1,0,0) branch of :798-799
    assert abs(grasps[0, 3]) + abs(grasps[0, 4]) == 0.0


def test_frames_of_records_is_the_frame_collect_pc_uses():
    grasps, T, pc = synth.dataset_view(3, 2000, 12)
    hp = _hand_points()
    frames = do.frames_of_records(grasps, T, 0.06)
    for g in range(len(grasps)):
        ind, pts = do.collect_pc_(grasps[g], pc, T, 0.06, hp, min_point_limit=0)
        m = frames[g, 3:].reshape(3, 3)
        want = (np.dot(m, (pc - frames[g, :3]).T)).T[ind]
        np.testing.assert_array_equal(pts, want)


def test_labels():
    g = np.zeros((4, 10))
    g[:, -2] = [0.2, 0.6, 0.4, 1.0]
    g[:, -1] = [0.0, 0.0, 50.0, 0.0]      # 0.4 + 0.5 = 0.9
    lab = dc.labels_from_scores(g, thresh_good=0.3, thresh_bad=0.6)
    np.testing.assert_array_equal(lab, [1, 0, 0, 0])
    lab = dc.labels_from_scores(g, thresh_good=0.3, thresh_bad=0.95)
    np.testing.assert_array_equal(lab, [1, -1, -1, 0])


def test_getitem_tail_resample_rule():
    pts = np.arange(30, dtype=np.float64).reshape(10, 3)
    calls = []

    def choice(n, size, replace):
        calls.append((n, size, replace))
        return np.arange(size) % n

    g = np.zeros(10)
    out = do.getitem_tail(g, pts, 10, 0.3, 0.6, choice)       # len == N: WITH replacement (:1104 uses >)
    assert calls[-1] == (10, 10, True) and out[0].shape == (3, 10) and out[1] == 1
    do.getitem_tail(g, pts, 9, 0.3, 0.6, choice)
    assert calls[-1] == (10, 9, False)
    g[-2] = 0.45
    assert do.getitem_tail(g, pts, 9, 0.3, 0.6, choice) is None
    assert do.getitem_tail(g, None, 9, 0.3, 0.6, choice) is None


def test_restatement_regression_vectors():
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_crop_small.npz"))
    grasps, T, pc = synth.dataset_view(int(d["seed"]), int(d["S"]), int(d["G"]))
    hp = _hand_points()
    for g in range(len(grasps)):
        ind, gpc = do.collect_pc_(grasps[g], pc, T, 0.06, hp, min_point_limit=0)
        assert len(ind) == int(d["counts"][g])
        if len(ind) >= 4:
            np.testing.assert_array_equal(gpc[:4], d["first_points"][g])


def test_frames_and_local_coordinates_equal_the_reference_run():
    """tests/golden/dataset_frames_reference.npz holds what the reference's own statements of collect_pc_
    (model/dataset.py:776-843, taken out of the method's syntax tree and run by tests/golden/make_dataset_golden.py)
    return for 40 grasp records - axes along z, along x and y, of arbitrary length, angles 0 and pi/2 among them -
    under a rigid and a sheared 4x4 transform: `matrix`, `bottom_` and the gripper-frame coordinates of 96 points.
    The restatement returns the same doubles, bit for bit."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_frames_reference.npz"))
    rec, pc, hd = d["records"], d["pc"], float(d["hand_depth"])
    for name in ("rigid", "sheared"):
        T = d[f"transform_{name}"]
        frames = do.frames_of_records(rec, T, hd)
        np.testing.assert_array_equal(frames[:, :3], d[f"bottom_{name}"])
        np.testing.assert_array_equal(frames[:, 3:].reshape(-1, 3, 3), d[f"matrix_{name}"])
        for g, r in enumerate(rec):
            bottom_, matrix = do.frame_of_record(r, T, hd)
            np.testing.assert_array_equal((np.dot(matrix, (pc - bottom_).T)).T, d[f"pc_c2g_{name}"][g])


def test_getitem_tail_equals_the_reference_run():
    """tests/golden/dataset_tail_reference.npz: the statements of PointGraspOneViewDataset.__getitem__ after
    collect_pc_ (model/dataset.py:1098-1131), taken out of the method and run with numpy's global generator seeded per
    case - clouds of more than, exactly, and fewer than grasp_points_num points; scores on both sides of and between
    the two thresholds.  The restatement, drawing from the same seeded generator, returns the same arrays and labels."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_tail_reference.npz"))
    N, good, bad = int(d["N"]), float(d["thresh_good"]), float(d["thresh_bad"])
    kept = 0
    for i in range(int(d["count"])):
        np.random.seed(1000 + i)
        res = do.getitem_tail(d[f"grasp_{i}"], d[f"pc_{i}"], N, good, bad,
                              lambda n, size, replace: np.random.choice(n, size=size, replace=replace))
        assert (res is not None) == bool(d[f"kept_{i}"])
        if res is not None:
            kept += 1
            np.testing.assert_array_equal(res[0], d[f"out_{i}"])
            assert res[1] == int(d[f"label_{i}"])
    assert 0 < kept < int(d["count"])
