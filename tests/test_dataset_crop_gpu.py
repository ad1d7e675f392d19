"""GPU parity of the training-data crop (pngpd_crop_count_frames / _emit / _gather_dataset through the
C ABI) against the restatement of collect_pc_ + __getitem__ (model/dataset.py:767-924, 1098-1131):
index sets and gripper-frame float64 coordinates bit-exact for given frames, resampled float32 input bit-exact
for given draws; the frames the device builds from grasp records (pngpd_dataset_frames) within a stated bound of
the restatement's and of a run of the reference's own statements (tests/golden/dataset_frames_reference.npz)."""
import numpy as np
import pytest
import torch

from oracle import crop_oracle as co
from oracle import dataset_crop_oracle as do
from pointnetgpd_amd import crop as hc
from pointnetgpd_amd import dataset_crop as dc
from pointnetgpd_amd import synth

pytestmark = pytest.mark.gpu

HD, OW, FW, HH = 0.06, 0.08, 0.01, 0.02


def _hp():
    return co.hand_points(HD, OW + 2 * FW, FW, HH)


def _setup(seed, S, G):
    grasps, T, pc = synth.dataset_view(seed, S, G)
    hp = _hp()
    limits = hc.box_limits_from_hand_points(hp)
    frames = do.frames_of_records(grasps, T, HD)        # the restatement's frames: the per-point tests start from them
    return grasps, T, pc, hp, limits, frames


# The device evaluates the frame arithmetic in its own order and with its own sin / cos; entries are O(1).
FRAME_TOL = 8 * np.finfo(np.float64).eps


def test_device_frames_match_the_restatement():
    for seed, G, stride in ((11, 64, 10), (15, 1000, 10), (16, 7, 8)):
        grasps, T, pc = synth.dataset_view(seed, 1000, G)
        T = T.copy()
        if seed == 15:
            T[:3, :3] *= 1.5                              # a transform with scale: the axes are not re-normalised
        want = do.frames_of_records(grasps, T, HD)
        got = dc.frames_from_configs(grasps[:, :stride], T, HD).cpu().numpy()
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err <= FRAME_TOL * max(1.0, np.abs(want).max()), err
    # the axis-along-z record takes the (1,0,0) branch (model/dataset.py:798-799)
    assert abs(grasps[0, 3]) + abs(grasps[0, 4]) == 0.0


def test_crop_from_device_built_frames_equals_the_restatement():
    """Records -> frames -> crop entirely on the device: same index sets as collect_pc_ on this seeded view (no
    point of it lies within 1e-12 of a plane, checked below, so a last-place difference in a frame cannot move it)."""
    grasps, T, pc = synth.dataset_view(11, 200_000, 64)
    hp = _hp()
    limits = hc.box_limits_from_hand_points(hp)
    frames = dc.frames_from_configs(grasps, T, HD)
    res = dc.collect_pc_frames(frames, torch.from_numpy(pc).cuda(), limits)
    idx, pts = res.index_lists(), res.point_lists()
    total = 0
    for g in range(len(grasps)):
        b, m = do.frame_of_record(grasps[g], T, HD)
        pg = np.dot(m, (pc - b).T).T
        margin = np.min(np.abs(np.stack([pg[:, 1] - limits[0], pg[:, 1] - limits[1], pg[:, 2] - limits[2],
                                         pg[:, 2] - limits[3], pg[:, 0] - limits[4], pg[:, 0] - limits[5]])))
        assert margin > 1e-12
        ind, gpc = do.collect_pc_(grasps[g], pc, T, HD, hp, min_point_limit=0)
        np.testing.assert_array_equal(idx[g], ind)
        if len(ind):
            assert np.abs(pts[g] - pg[ind]).max() <= 1e-14
        total += len(ind)
    assert total > 1000


def test_index_sets_and_points_bit_exact():
    grasps, T, pc, hp, limits, frames = _setup(11, 200_000, 64)
    res = dc.collect_pc_frames(frames, torch.from_numpy(pc).cuda(), limits)
    idx, pts = res.index_lists(), res.point_lists()
    nonempty = 0
    for g in range(len(grasps)):
        ind, _ = do.collect_pc_(grasps[g], pc, T, HD, hp, min_point_limit=0)
        np.testing.assert_array_equal(idx[g], ind)
        m = frames[g, 3:].reshape(3, 3)
        want = (np.dot(m, (pc - frames[g, :3]).T)).T[ind]
        np.testing.assert_array_equal(pts[g], want)
        nonempty += len(ind) > 0
    assert nonempty > 32


def test_frames_with_scale_and_shear_are_cropped_like_the_restatement():
    """collect_pc_ applies the view transform to the axes without re-normalising them (model/dataset.py:831-835): under
    a transform that shrinks and shears, the region is larger in the view than its nominal size and no longer a
    rectangular box.  The culled traversal must still find every point the all-pairs arithmetic finds."""
    grasps, T, pc = synth.dataset_view(17, 150_000, 48)
    A = np.array([[0.6, 0.15, 0.0], [0.0, 0.55, -0.1], [0.05, 0.0, 0.7]])
    T = T.copy()
    T[:3, :3] = A @ T[:3, :3]
    hp = _hp()
    limits = hc.box_limits_from_hand_points(hp)
    frames = do.frames_of_records(grasps, T, HD)
    res = dc.collect_pc_frames(frames, torch.from_numpy(pc).cuda(), limits)
    idx, pts = res.index_lists(), res.point_lists()
    total = 0
    for g in range(len(grasps)):
        ind, _ = do.collect_pc_(grasps[g], pc, T, HD, hp, min_point_limit=0)
        np.testing.assert_array_equal(idx[g], ind)
        m = frames[g, 3:].reshape(3, 3)
        np.testing.assert_array_equal(pts[g], (np.dot(m, (pc - frames[g, :3]).T)).T[ind])
        total += len(ind)
    assert total > 2000


def test_resample_replays_given_draws_and_min_point_limit():
    grasps, T, pc, hp, limits, frames = _setup(12, 14_000, 48)     # ~50 points per gripper: both sides of the limit
    N = 45
    res = dc.collect_pc_frames(frames, torch.from_numpy(pc).cuda(), limits)
    cnt = res.counts.cpu().numpy()
    rng = np.random.default_rng(0)
    sel = np.zeros((len(grasps), N), dtype=np.int32)
    for g, n in enumerate(cnt):
        if n >= dc.MIN_POINT_LIMIT:
            sel[g] = rng.choice(n, size=N, replace=not (n > N))
    data, valid = dc.resample(res, N, sel=sel)
    data, valid = data.cpu().numpy(), valid.cpu().numpy()
    assert data.shape == (len(grasps), 3, N)
    checked = 0
    for g in range(len(grasps)):
        ind, gpc = do.collect_pc_(grasps[g], pc, T, HD, hp)
        if gpc is None:
            assert valid[g] == 0 and not data[g].any()
            continue
        out = do.getitem_tail(np.array([0.0] * 10), gpc, N, 0.3, 0.6, lambda n, size, replace: sel[g])
        np.testing.assert_array_equal(data[g], out[0].astype(np.float32))
        assert valid[g] == 1
        checked += 1
    assert checked > 10 and (valid == 0).any()


def test_device_draws_follow_the_strictly_more_rule():
    grasps, T, pc, hp, limits, frames = _setup(13, 300_000, 32)
    res = dc.collect_pc_frames(frames, torch.from_numpy(pc).cuda(), limits)
    cnt = res.counts.cpu().numpy()
    g_eq = int(np.argmax(cnt))
    N = int(cnt[g_eq])                     # this grasp has exactly N points: WITH replacement (dataset.py:1104)
    assert N >= dc.MIN_POINT_LIMIT
    _, valid, sel = dc.resample(res, N, seed=7, return_selection=True)
    sel = sel.cpu().numpy()
    for g in range(len(grasps)):
        if cnt[g] < dc.MIN_POINT_LIMIT:
            assert (sel[g] == -1).all()
            continue
        twin = hc.draw_selection(7, g, 0, int(cnt[g]), N, replace_when_equal=True)
        np.testing.assert_array_equal(sel[g], twin)
        assert sel[g].min() >= 0 and sel[g].max() < cnt[g]
    # the online rule would have drawn a permutation for the n == N grasp; the dataset rule does not
    assert len(np.unique(sel[g_eq])) < N
    # fewer points than N is always with replacement; a grasp with more than N would be a partial permutation
    _, _, sel2 = dc.resample(res, N - 1, seed=7, return_selection=True)
    assert len(np.unique(sel2.cpu().numpy()[g_eq])) == N - 1


def test_make_training_batch_feeds_the_network_layout():
    grasps, T, pc, hp, limits, frames = _setup(14, 300_000, 40)
    batch = dc.make_training_batch(grasps, torch.from_numpy(pc).cuda(), T, limits, HD, 750, 0.3, 0.6, seed=1)
    lab = dc.labels_from_scores(grasps, 0.3, 0.6)
    assert batch.data.shape[1:] == (3, 750) and batch.data.dtype == torch.float32
    assert batch.data.shape[0] == batch.label.shape[0] == len(batch.kept) > 0
    np.testing.assert_array_equal(batch.label.cpu().numpy(), lab[batch.kept])
    assert (lab[batch.kept] >= 0).all()
    # every kept sample lies inside the gripper box
    d = batch.data.double().cpu().numpy()
    assert (d[:, 1] > limits[0] - 1e-6).all() and (d[:, 1] < limits[1] + 1e-6).all()
    assert (d[:, 0] > limits[4] - 1e-6).all() and (d[:, 0] < limits[5] + 1e-6).all()


def test_device_frames_match_the_reference_run():
    """The same records and transforms as tests/golden/dataset_frames_reference.npz (outputs of the reference's own
    statements, model/dataset.py:776-843): the frames the device builds lie within the stated bound of them."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_frames_reference.npz"))
    rec, hd = d["records"], float(d["hand_depth"])
    for name in ("rigid", "sheared"):
        T = d[f"transform_{name}"]
        got = dc.frames_from_configs(rec, T, hd).cpu().numpy()
        want = np.concatenate([d[f"bottom_{name}"], d[f"matrix_{name}"].reshape(len(rec), 9)], axis=1)
        err = np.abs(got - want).max()
        print(f"{name}: max |device - reference| = {err:.2e}")
        assert err <= FRAME_TOL * max(1.0, np.abs(want).max()), err
