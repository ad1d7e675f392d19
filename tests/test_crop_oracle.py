"""CPU: the crop oracle's three forms (reference numpy calls / pinned fma order / plain C) agree bit for bit with
the committed golden vectors - outputs of the reference's own check_collision_square (apps/kinect2grasp.py:299-374),
taken out of the reference file and run by tests/golden/make_crop_golden.py - and honour the strict inequalities."""
import hashlib
import os

import numpy as np

from oracle import crop_oracle as co
from pointnetgpd_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def random_case():
    d = np.load(os.path.join(GOLDEN, "crop_random.npz"))
    S, G, seed = (int(v) for v in d["meta"])
    cube = float(d["cube"])
    return d, synth.scene_points(seed, S, cube), synth.random_grasps(seed, G, cube, 0.03), d["limits"]


def test_numpy_form_matches_golden():
    d, scene, grasps, limits = random_case()
    idx, pts = co.crop_numpy(grasps, scene, limits)
    np.testing.assert_array_equal([len(i) for i in idx], d["counts"])
    np.testing.assert_array_equal(np.concatenate(idx), d["idx"])
    np.testing.assert_array_equal(np.concatenate(pts[:6]), d["pts_head"])
    for p, h in zip(pts, d["pts_sha256"]):
        assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == str(h)


def test_pinned_and_c_forms_are_bit_identical_to_numpy_form():
    d, scene, grasps, limits = random_case()
    idx_c, pts_c = co.crop_c(grasps, scene, limits)
    np.testing.assert_array_equal(np.concatenate(idx_c), d["idx"])
    np.testing.assert_array_equal(np.concatenate(pts_c[:6]), d["pts_head"])
    for p, h in zip(pts_c, d["pts_sha256"]):
        assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == str(h)
    centre, a, b, m = co.split_grasps(grasps)
    for g in (0, 5, 17):
        i_np, pg_np = co.crop_one_numpy(centre[g], a[g], b[g], m[g], scene, limits)
        i_pin, pg_pin = co.crop_one_pinned(centre[g], a[g], b[g], m[g], scene, limits)
        np.testing.assert_array_equal(i_np, i_pin)
        np.testing.assert_array_equal(pg_np, pg_pin)          # every one of the S x 3 doubles
        np.testing.assert_array_equal(co.frame_numpy(a[g], b[g], m[g]), co.frame_pinned(a[g], b[g], m[g]))


def test_points_on_the_planes_are_excluded():
    d = np.load(os.path.join(GOLDEN, "crop_planes.npz"))
    for fn in (co.crop_numpy, co.crop_c):
        idx, pts = fn(d["grasps"], d["scene"], d["limits"])
        np.testing.assert_array_equal([len(i) for i in idx], d["counts"])
        np.testing.assert_array_equal(np.concatenate(idx), d["idx"])
        np.testing.assert_array_equal(np.concatenate(pts), d["pts"])
    # frame 0 is the identity: its 19 local points are scene rows 0..18 = centre, then per plane
    # (on, +1 ulp, -1 ulp).  On-plane points must be out; exactly one neighbour of each is in.
    idx0 = set(co.crop_numpy(d["grasps"][:1], d["scene"], d["limits"])[0][0].tolist())
    assert 0 in idx0
    for plane in range(6):
        on, up, down = 1 + 3 * plane, 2 + 3 * plane, 3 + 3 * plane
        assert on not in idx0
        assert (up in idx0) != (down in idx0)


def test_near_plane_golden_of_the_reference_function():
    """Oblique grasps with axes of random length and points within a few float32 ulps of each plane."""
    d = np.load(os.path.join(GOLDEN, "crop_near_planes.npz"))
    assert 0 < d["counts"].min() and d["counts"].sum() < d["scene"].shape[0] * len(d["grasps"])
    for fn in (co.crop_numpy, co.crop_c):
        idx, pts = fn(d["grasps"], d["scene"], d["limits"])
        np.testing.assert_array_equal([len(i) for i in idx], d["counts"])
        np.testing.assert_array_equal(np.concatenate(idx), d["idx"])
        for p, h in zip(pts, d["pts_sha256"]):
            assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == str(h)
    centre, a, b, m = co.split_grasps(d["grasps"])
    start = np.concatenate([[0], np.cumsum(d["counts"])])
    for g in (0, 7, 23):
        i_pin, _ = co.crop_one_pinned(centre[g], a[g], b[g], m[g], d["scene"], d["limits"])
        np.testing.assert_array_equal(i_pin, d["idx"][start[g]:start[g + 1]])


def test_empty_crop_and_grasp_reshape():
    limits = co.box_limits(co.hand_points(0.06, 0.12, 0.02, 0.02))
    scene = synth.scene_points(3, 100, 0.1)
    far = np.zeros((1, 5, 3))
    far[0, 4] = [5, 5, 5]
    far[0, 1:4] = np.eye(3)
    idx, pts = co.crop_numpy(far.reshape(-1), scene, limits)     # flat list form, kinect2grasp.py:389
    assert len(idx) == 1 and idx[0].size == 0 and pts[0].shape == (0, 3)


def test_box_limits_of_hand_model():
    lim = co.box_limits(co.hand_points(hand_depth=0.06, hand_outer_diameter=0.12, finger_width=0.02, hand_height=0.02))
    np.testing.assert_allclose(lim, [-0.04, 0.04, -0.01, 0.01, 0.0, 0.06])


def test_vote_semantics():
    probs = np.array([[[0.3, 0.7]], [[0.6, 0.4]], [[0.2, 0.8]]], dtype=np.float32)
    label, score = co.vote([1, 0, 1], probs, 2)
    assert label == 1 and score == np.mean(np.array([0.7, 0.8], dtype=np.float32))
    label, score = co.vote([0, 1, 2], np.ones((3, 1, 3), np.float32) / 3, 3)   # three-way tie -> smallest
    assert label == 0


REFERENCE_FILE = "/root/reference/apps/kinect2grasp.py"


def _reference_check_collision_square():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_crop_golden", os.path.join(GOLDEN, "make_crop_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.reference_function("check_collision_square"), mod


import pytest  # noqa: E402


@pytest.mark.skipif(not os.path.exists(REFERENCE_FILE), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("seed", [3, 14, 159, 2653])
def test_live_against_the_reference_function_on_fresh_random_cases(seed):
    """Beyond the committed vectors: in the build container the reference's own check_collision_square (taken out of
    its file's syntax tree, numpy its only global) runs next to the restatement on cases drawn here - oblique grasps
    with axes of random length, scenes of mixed density, a few points pushed to within ulps of a plane - and both
    forms of the restatement return the same index lists and the same float64 coordinates, bit for bit."""
    fn, mod = _reference_check_collision_square()
    rng = np.random.default_rng(seed)
    S, G, cube = int(rng.integers(500, 4000)), 12, 0.2
    p = co.hand_points(hand_depth=0.06, hand_outer_diameter=0.12, finger_width=0.02, hand_height=0.02)   # near_plane_case's
    limits = co.box_limits(p)
    scene = synth.scene_points(seed, S, cube)
    grasps = synth.random_grasps(seed, G, cube, 0.03).reshape(G, 5, 3).copy()
    grasps[:, 1:4, :] *= rng.uniform(0.3, 3.0, size=(G, 3, 1))
    ng, nscene, _ = mod.near_plane_case(seed=seed, G=2, per_plane=6)
    grasps = np.concatenate([grasps, ng.reshape(-1, 5, 3)])
    scene = np.vstack([scene, nscene]).astype(np.float32)
    ref_idx, ref_pts = mod.reference_collect_pc(grasps, scene, p)
    for form in (co.crop_numpy, co.crop_c):
        idx, pts = form(grasps, scene, limits)
        assert sum(len(i) for i in idx) > 0
        for g in range(len(grasps)):
            np.testing.assert_array_equal(idx[g], ref_idx[g])
            np.testing.assert_array_equal(pts[g], ref_pts[g])


def test_draw_rule_and_vote_score_equal_the_reference_statements():
    """tests/golden/crop_loop_reference.npz: three statements of the reference's scoring loop (apps/kinect2grasp.py:
    766-774 and 803-817), taken out of its `__main__` block and run: the draw of input_points_num points (without
    replacement when the grasp holds at least that many, with replacement below) under numpy's seeded generator, and
    the best-class rule with the mean score over the repeats that agree with the vote.  The vote itself is
    `mode(predict)[0][0]`, which scipy >= 1.11 no longer evaluates; its rule (smallest of the most frequent) is
    supplied to both sides."""
    d = np.load(os.path.join(GOLDEN, "crop_loop_reference.npz"))
    P = int(d["P"])
    below = 0
    for i in range(int(d["n_clouds"])):
        cloud = d[f"cloud_{i}"]
        np.random.seed(500 + i)
        sel = np.random.choice(len(cloud), P, replace=len(cloud) < P)
        np.testing.assert_array_equal(co.subsample(cloud, sel), d[f"drawn_{i}"])
        below += len(cloud) < P
    assert 0 < below < int(d["n_clouds"])
    for i in range(int(d["n_votes"])):
        label, score = co.vote(d[f"preds_{i}"], d[f"probs_{i}"], int(d[f"k_{i}"]))
        assert label == int(d[f"label_{i}"])
        assert (2 if int(d[f"k_{i}"]) == 3 else 1) == int(d[f"best_{i}"])
        assert score == d[f"score_{i}"] and np.asarray(score).dtype == d[f"score_{i}"].dtype


def test_vote_tie_rule_equals_scipy_mode_with_its_legacy_output_shape():
    """apps/kinect2grasp.py:799 reads `mode(predict)[0][0]`.  scipy >= 1.11 returns scalars from that call (the two
    subscripts then fail), but the same function with `keepdims=True` gives the arrays the reference indexes; its
    tie rule — the smallest of the most frequent values — is the one the restatement and the device vote use."""
    from scipy.stats import mode
    rng = np.random.default_rng(11)
    ties = 0
    for _ in range(400):
        k = int(rng.integers(2, 4))
        R = int(rng.integers(1, 8))
        preds = [int(v) for v in rng.integers(0, k, R)]
        probs = rng.random((R, 1, k), dtype=np.float32)
        want = int(mode(preds, keepdims=True)[0][0])
        label, _ = co.vote(preds, probs, k)
        assert label == want
        counts = np.bincount(preds, minlength=k)
        ties += int((counts == counts.max()).sum() > 1)
    assert ties > 50
