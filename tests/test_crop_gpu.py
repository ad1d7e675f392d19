"""GPU parity of the in-gripper crop / subsample / vote path (through the C ABI) against the
float64 oracle: index sets and gripper-frame coordinates must be BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

from oracle import crop_oracle as co
from oracle import pointnet_oracle as po
from pointnetgpd_amd import crop as hc
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
from pointnetgpd_amd.scoring import score_candidates

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIMITS = co.box_limits(co.hand_points(hand_depth=0.06, hand_outer_diameter=0.12, finger_width=0.02, hand_height=0.02))


def assert_same_crop(res, idx_ref, pts_ref):
    got_idx = res.index_lists()
    got_pts = res.point_lists()
    assert len(got_idx) == len(idx_ref)
    np.testing.assert_array_equal(res.counts.cpu().numpy(), [len(i) for i in idx_ref])
    for g, (a, b) in enumerate(zip(got_idx, idx_ref)):
        assert a.dtype == np.int64
        np.testing.assert_array_equal(a, b, err_msg=f"grasp {g}")
    for g, (a, b) in enumerate(zip(got_pts, pts_ref)):
        assert a.dtype == np.float64
        np.testing.assert_array_equal(a, b, err_msg=f"grasp {g}: gripper-frame points differ")


def test_golden_random_case():
    d = np.load(os.path.join(GOLDEN, "crop_random.npz"))
    S, G, seed = (int(v) for v in d["meta"])
    cube = float(d["cube"])
    scene, grasps = synth.scene_points(seed, S, cube), synth.random_grasps(seed, G, cube, 0.03)
    res = hc.collect_pc(grasps, torch.from_numpy(scene).cuda(), d["limits"])
    np.testing.assert_array_equal(res.counts.cpu().numpy(), d["counts"])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), d["idx"])
    n_head = int(d["counts"][:6].sum())
    np.testing.assert_array_equal(res.points_g.cpu().numpy()[:n_head], d["pts_head"])


def test_golden_near_plane_case():
    """Outputs of the reference's own check_collision_square on oblique grasps (axes not of unit length) with points
    a few float32 ulps either side of every plane: index sets and gripper-frame float64 points, bit for bit."""
    import hashlib
    d = np.load(os.path.join(GOLDEN, "crop_near_planes.npz"))
    res = hc.collect_pc(d["grasps"], torch.from_numpy(d["scene"]).cuda(), d["limits"])
    np.testing.assert_array_equal(res.counts.cpu().numpy(), d["counts"])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), d["idx"])
    for p, h in zip(res.point_lists(), d["pts_sha256"]):
        assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == str(h)


def test_golden_points_on_planes_are_excluded():
    d = np.load(os.path.join(GOLDEN, "crop_planes.npz"))
    res = hc.collect_pc(d["grasps"], torch.from_numpy(d["scene"]).cuda(), d["limits"])
    np.testing.assert_array_equal(res.counts.cpu().numpy(), d["counts"])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), d["idx"])
    np.testing.assert_array_equal(res.points_g.cpu().numpy(), d["pts"])


@pytest.mark.parametrize("S,G", [(200_000, 100), (1023, 3), (1025, 33), (50, 1)])
def test_against_c_oracle(S, G):
    seed = 61 + G
    scene, grasps = synth.scene_points(seed, S, 0.3), synth.random_grasps(seed, G, 0.3, 0.03)
    idx_ref, pts_ref = co.crop_c(grasps, scene, LIMITS)
    res = hc.collect_pc(grasps.reshape(-1), torch.from_numpy(scene).cuda(), LIMITS)   # flat form as in :385-389
    assert_same_crop(res, idx_ref, pts_ref)


def test_empty_crops():
    scene = synth.scene_points(5, 5000, 0.1)
    grasps = synth.random_grasps(5, 4, 0.1, 0.03)
    grasps[:, 4] += 10.0                     # far away: nothing inside
    res = hc.collect_pc(grasps, torch.from_numpy(scene).cuda(), LIMITS)
    assert res.indices.numel() == 0 and int(res.counts.sum()) == 0
    pts, valid = hc.subsample(res, 750, repeat=3)
    assert not valid.any() and float(pts.abs().max()) == 0.0


def make_crop(S=60_000, G=24, seed=71, cube=0.2):
    scene, grasps = synth.scene_points(seed, S, cube), synth.random_grasps(seed, G, cube, 0.03)
    return scene, grasps, hc.collect_pc(grasps, torch.from_numpy(scene).cuda(), LIMITS)


def test_subsample_with_given_selection_matches_reference_rule():
    """points_modify = in_ind_points[ii][np.random.choice(...)] -> FloatTensor (kinect2grasp.py:766-774)."""
    P, R = 500, 3
    scene, grasps, res = make_crop()
    idx_ref, pts_ref = co.crop_c(grasps, scene, LIMITS)
    rng = np.random.default_rng(0)
    counts = [len(i) for i in idx_ref]
    assert min(counts) >= co.MIN_POINTS and any(c < P for c in counts) and any(c >= P for c in counts)
    sel = np.stack([rng.choice(c, P, replace=c < P) for c in counts for _ in range(R)]).astype(np.int32)
    pts, valid = hc.subsample(res, P, repeat=R, sel=sel)
    assert valid.all()
    expect = np.stack([co.subsample(pts_ref[g], sel[g * R + r]).astype(np.float32) for g in range(len(counts)) for r in range(R)])
    np.testing.assert_array_equal(pts.cpu().numpy(), expect)


def test_device_selection_matches_host_twin_and_sampling_rules():
    P, R, seed = 500, 3, 1234
    scene, grasps, res = make_crop()
    counts = res.counts.cpu().numpy()
    pts, valid, sel = hc.subsample(res, P, repeat=R, seed=seed, return_selection=True)
    sel = sel.cpu().numpy()
    for g, n in enumerate(counts):
        for r in range(R):
            row = sel[g * R + r]
            np.testing.assert_array_equal(row, hc.draw_selection(seed, g, r, int(n), P))
            assert row.min() >= 0 and row.max() < n
            if n >= P:
                assert len(set(row.tolist())) == P          # replace=False
    assert not np.array_equal(sel[0], sel[1])               # repeats differ
    # the gathered points are the selected cropped points
    _, pts_ref = co.crop_c(grasps, scene, LIMITS)
    np.testing.assert_array_equal(pts[4].cpu().numpy(), pts_ref[1][sel[4]].astype(np.float32))


def test_min_points_rule():
    scene, grasps, res = make_crop(S=3000, G=16, seed=73)      # sparse scene: ~10-20 points per grasp
    counts = res.counts.cpu().numpy()
    assert (counts < 20).any() and (counts >= 20).any()
    pts, valid = hc.subsample(res, 750, repeat=3, seed=1)
    np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), counts >= 20)


def test_vote_matches_reference_rule():
    rng = np.random.default_rng(3)
    for k in (2, 3):
        G, R = 200, 3
        probs = rng.dirichlet(np.ones(k), size=G * R).astype(np.float32)
        pred = probs.argmax(1).astype(np.int64)
        valid = (rng.random(G) > 0.2).astype(np.uint8)
        label, score = hc.vote(torch.from_numpy(pred).cuda(), torch.from_numpy(probs).cuda(),
                               torch.from_numpy(valid).cuda(), R)
        label, score = label.cpu().numpy(), score.cpu().numpy()
        for g in range(G):
            if not valid[g]:
                assert label[g] == 0 and score[g] == 0.0
                continue
            l_ref, s_ref = co.vote(pred[g * R:(g + 1) * R], probs[g * R:(g + 1) * R].reshape(R, 1, k), k)
            assert label[g] == l_ref
            assert score[g] == np.float32(s_ref), (g, score[g], s_ref)


def test_end_to_end_online_scoring_small_scene():
    """crop -> subsample(given draws) -> network -> softmax/argmax -> vote, against the oracle chain."""
    P, R, k, seed = 500, 3, 2, 81
    scene, grasps, res = make_crop(S=80_000, G=12, seed=seed)
    idx_ref, pts_ref = co.crop_c(grasps, scene, LIMITS)
    rng = np.random.default_rng(1)
    counts = [len(i) for i in idx_ref]
    sel = np.stack([rng.choice(c, P, replace=c < P) for c in counts for _ in range(R)]).astype(np.int32)
    sd = synth.torch_state_dict(seed, k, 0.02)
    m = PointNetCls(P, 3, k)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    pts, valid = hc.subsample(res, P, repeat=R, sel=sel)
    pred, probs, _ = score_candidates(m, pts)
    label, score = hc.vote(pred, probs, valid, R)
    oracle = po.build_cls(P, k, sd)
    for g in range(len(counts)):
        preds, prs = [], []
        for r in range(R):
            p_, pr_ = po.oracle_test_network(oracle, co.subsample(pts_ref[g], sel[g * R + r]))
            preds.append(int(p_[0]))
            prs.append(pr_)
        l_ref, s_ref = co.vote(preds, prs, k)
        margins = [abs(float(pr[0, 0] - pr[0, 1])) for pr in prs]
        if min(margins) > 1e-3:
            assert int(label[g]) == l_ref
            assert abs(float(score[g]) - float(s_ref)) <= 1e-4


def test_full_size_scene_properties():
    """S = 1,000,000 points x G = 10,000 grasps: ascending indices, counts add up, sampled grasps
    equal the C oracle bit for bit."""
    S, G, seed = 1_000_000, 10_000, 91
    scene, grasps = synth.scene_points(seed, S, 0.5), synth.random_grasps(seed, G, 0.5, 0.06)
    res = hc.collect_pc(grasps, torch.from_numpy(scene).cuda(), LIMITS, return_points=False)
    counts, off, idx = res.counts.cpu().numpy(), res.offsets.cpu().numpy(), res.indices.cpu().numpy()
    assert off[0] == 0 and off[-1] == counts.sum() == idx.size
    np.testing.assert_array_equal(np.diff(off), counts)
    assert idx.min() >= 0 and idx.max() < S
    starts = np.zeros(idx.size, dtype=bool)
    starts[off[:-1][counts > 0]] = True
    assert (np.diff(idx.astype(np.int64)) > 0)[~starts[1:]].all()      # ascending inside every grasp
    assert 600 < counts.mean() < 950
    pick = [0, 1, 4999, 9999]
    idx_ref, _ = co.crop_c(grasps[pick], scene, LIMITS)
    for g, ref in zip(pick, idx_ref):
        np.testing.assert_array_equal(idx[off[g]:off[g + 1]], ref)


def test_config5_online_scoring_at_full_size():
    """S = 1,000,000 points x G = 10,000 grasps through online.score_grasps (crop + 3 draws + network + vote + rank)
    with a classifier centred so that about half of the grasps are good: the one-pass crop + subsample equals the
    three-call path for all 10,000 grasps bit for bit, the ranking is the descending stable sort of the good
    grasps, and sampled grasps equal the oracle chain (C crop oracle -> the device's draws replayed by their host
    twin -> eager float32 network -> softmax / arg-max -> vote)."""
    from pointnetgpd_amd.online import score_grasps
    S, G, P, k, R, seed = 1_000_000, 10_000, 750, 2, 3, 93
    scene = synth.scene_points(seed, S, 0.5)
    grasps = synth.random_grasps(seed, G, 0.5, 0.06)
    scene_dev = torch.from_numpy(scene).cuda()
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, 0.02))
    m = m.cuda().eval()
    synth.balance_head(m, hc.crop_subsample(grasps[:512], scene_dev, LIMITS, P, repeat=1, seed=1)[0])
    out = score_grasps(m, scene_dev, grasps, LIMITS, repeat=R, seed=5)
    label, score, valid = out.label.cpu().numpy(), out.score.cpu().numpy(), out.valid.cpu().numpy()
    n_valid, n_good = int(valid.sum()), int(((label == 1) & (valid != 0)).sum())
    assert n_valid > 0.95 * G and 0.2 * n_valid < n_good < 0.8 * n_valid, (n_valid, n_good)
    order = out.order.cpu().numpy()
    assert set(order.tolist()) == set(np.nonzero((label == 1) & (valid != 0))[0].tolist())
    assert (np.diff(score[order]) <= 0).all()
    ties = np.nonzero(np.diff(score[order]) == 0)[0]
    assert (order[ties] < order[ties + 1]).all()                       # stable: equal scores keep the grasp order
    # one pass == three calls, every grasp
    res = hc.collect_pc(grasps, scene_dev, LIMITS, return_points=False)
    pts_ref, valid_ref = hc.subsample(res, P, repeat=R, seed=5)
    pts, valid1, counts, overflow = hc.crop_subsample(grasps, scene_dev, LIMITS, P, repeat=R, seed=5)
    assert int(overflow.item()) == 0
    assert torch.equal(counts, res.counts) and torch.equal(valid1, valid_ref) and torch.equal(pts, pts_ref)
    assert torch.equal(out.counts, res.counts) and torch.equal(out.valid, valid_ref)
    # sampled grasps against the oracle chain
    sd = {n: v.detach().cpu() for n, v in m.state_dict().items()}
    oracle = po.build_cls(P, k, sd)
    good, bad = order[:: max(1, len(order) // 10)][:10], np.nonzero((label == 0) & (valid != 0))[0][::97][:10]
    pick = sorted(set(good.tolist() + bad.tolist() + [0, G - 1]))
    idx_ref, pts_g_ref = co.crop_c(grasps[pick], scene, LIMITS)
    cnt = res.counts.cpu().numpy()
    checked = 0
    for j, g in enumerate(pick):
        assert len(idx_ref[j]) == cnt[g]
        if cnt[g] < 20:
            assert valid[g] == 0 and label[g] == 0 and score[g] == 0.0
            continue
        preds, prs = [], []
        for r in range(R):
            sel = hc.draw_selection(5, g, r, int(cnt[g]), P)
            np.testing.assert_array_equal(pts[g * R + r].cpu().numpy(), pts_g_ref[j][sel].astype(np.float32))
            p_, pr_ = po.oracle_test_network(oracle, co.subsample(pts_g_ref[j], sel))
            preds.append(int(p_[0]))
            prs.append(pr_)
        l_ref, s_ref = co.vote(preds, prs, k)
        if min(abs(float(pr[0, 0] - pr[0, 1])) for pr in prs) > 1e-3:
            assert int(label[g]) == l_ref, g
            assert abs(float(score[g]) - float(s_ref)) <= 1e-4, g
            checked += 1
    assert checked >= 12


def test_online_pipeline_ranks_good_grasps():
    from pointnetgpd_amd.dist import score_grasps_sharded
    from pointnetgpd_amd.online import score_grasps
    P, k, seed = 500, 2, 83
    scene = torch.from_numpy(synth.scene_points(seed, 80_000, 0.2)).cuda()
    grasps = synth.random_grasps(seed, 40, 0.2, 0.03)
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, 0.02))
    m = m.cuda().eval()
    out = score_grasps(m, scene, grasps, LIMITS, repeat=3, seed=5)
    label, score, valid = score_grasps_sharded(m, scene, grasps, LIMITS, repeat=3, seed=5)
    assert torch.equal(out.label, label) and torch.equal(out.score, score) and torch.equal(out.valid, valid)
    s = out.score[out.order].cpu().numpy()
    assert (np.diff(s) <= 0).all()
    assert set(out.order.cpu().tolist()) == set(np.nonzero((out.label.cpu().numpy() == 1) & (out.valid.cpu().numpy() != 0))[0].tolist())
    # a slice scored with its global offset draws what the full run drew
    from pointnetgpd_amd import crop as hc2
    res = hc2.collect_pc(grasps[10:20], scene, LIMITS, return_points=False)
    pts_slice, _ = hc2.subsample(res, P, repeat=3, seed=5, grasp_offset=10)
    res_all = hc2.collect_pc(grasps, scene, LIMITS, return_points=False)
    pts_all, _ = hc2.subsample(res_all, P, repeat=3, seed=5)
    assert torch.equal(pts_slice, pts_all[30:60])


@pytest.mark.parametrize("S,G,cube,P", [(80_000, 40, 0.2, 500), (300_000, 200, 0.4, 750), (2_000, 10, 3.0, 64),
                                        (1_000_000, 500, 0.5, 750)])
def test_one_pass_crop_subsample_equals_the_three_calls(S, G, cube, P):
    seed = 17 + G
    scene = torch.from_numpy(synth.scene_points(seed, S, cube)).cuda()
    grasps = synth.random_grasps(seed, G, cube, 0.03)
    res = hc.collect_pc(grasps, scene, LIMITS, return_points=False)
    pts_ref, valid_ref = hc.subsample(res, P, repeat=3, seed=9, grasp_offset=7)
    pts, valid, counts, overflow = hc.crop_subsample(grasps, scene, LIMITS, P, repeat=3, seed=9, grasp_offset=7)
    assert int(overflow.item()) == 0
    assert torch.equal(counts, res.counts) and torch.equal(valid, valid_ref)
    assert torch.equal(pts, pts_ref)                      # bit for bit: same list order, same draws, same arithmetic


def test_one_pass_crop_subsample_reports_grasps_it_cannot_hold():
    """A dense scene puts more than 1536 points into some grasps: the one-pass kernel says so, its counts stay
    exact, and score_grasps falls back on the three calls."""
    from pointnetgpd_amd.online import score_grasps
    from pointnetgpd_amd.dist import score_grasps_sharded
    P, k, seed = 500, 2, 29
    scene = torch.from_numpy(synth.scene_points(seed, 90_000, 0.05)).cuda()
    grasps = synth.random_grasps(seed, 12, 0.05, 0.0)
    res = hc.collect_pc(grasps, scene, LIMITS, return_points=False)
    assert int(res.counts.max()) > 1536
    pts, valid, counts, overflow = hc.crop_subsample(grasps, scene, LIMITS, P, repeat=3, seed=5)
    assert int(overflow.item()) == 1
    assert torch.equal(counts, res.counts)
    pts_ref, valid_ref = hc.subsample(res, P, repeat=3, seed=5)
    assert torch.equal(valid, valid_ref)
    small = (res.counts <= 1536).nonzero().flatten().cpu().tolist()
    for g in small:                                        # grasps that fit are complete all the same
        assert torch.equal(pts[3 * g:3 * g + 3], pts_ref[3 * g:3 * g + 3])
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, 0.02))
    m = m.cuda().eval()
    out = score_grasps(m, scene, grasps, LIMITS, repeat=3, seed=5)
    pred, probs, _ = score_candidates(m, pts_ref)
    label, score = hc.vote(pred, probs, valid_ref, 3)
    assert torch.equal(out.label, label) and torch.equal(out.score, score) and torch.equal(out.valid, valid_ref)
    assert torch.equal(out.counts, res.counts)
    label2, score2, valid2 = score_grasps_sharded(m, scene, grasps, LIMITS, repeat=3, seed=5)
    assert torch.equal(label2, label) and torch.equal(score2, score) and torch.equal(valid2, valid_ref)


@pytest.mark.parametrize("S,G,cube", [(300_000, 200, 0.4), (5_000, 50, 0.05), (2_000, 10, 3.0), (90_000, 12, 0.05)])
def test_grid_culled_pass_equals_all_pairs_pass(S, G, cube, monkeypatch):
    """The spatially culled mask pass and the brute-force all-pairs pass give identical crops
    (dense scene, a scene smaller than one gripper, a sparse scene much larger than the grid, and a scene so
    dense that a gripper holds more points than a wave sorts in LDS: the index-slab path)."""
    seed = 97 + G
    scene = torch.from_numpy(synth.scene_points(seed, S, cube)).cuda()
    grasps = synth.random_grasps(seed, G, cube, min(0.03, cube / 4))
    monkeypatch.setenv("PNGPD_CROP_BRUTE", "1")
    brute = hc.collect_pc(grasps, scene, LIMITS)
    monkeypatch.setenv("PNGPD_CROP_BRUTE", "0")
    culled = hc.collect_pc(grasps, scene, LIMITS)
    assert torch.equal(brute.counts, culled.counts)
    assert torch.equal(brute.indices, culled.indices)
    assert torch.equal(brute.points_g, culled.points_g)
    if cube < 1.0:
        assert int(culled.counts.sum()) > 0


def test_grid_culling_with_degenerate_scenes():
    pts = np.tile(np.array([[0.1, 0.1, 0.1]], dtype=np.float32), (500, 1))         # all points identical
    g = np.zeros((2, 5, 3))
    g[:, 1:4] = np.eye(3)
    g[0, 4] = [0.07, 0.1, 0.1]      # point lies 3 cm in front of the palm: inside
    g[1, 4] = [0.5, 0.5, 0.5]       # far away: empty
    res = hc.collect_pc(g, torch.from_numpy(pts).cuda(), LIMITS)
    np.testing.assert_array_equal(res.counts.cpu().numpy(), [500, 0])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), np.arange(500))
    # 3000 hits: above half the LDS list capacity (sorted in place by the bitonic network)
    pts = np.tile(np.array([[0.1, 0.1, 0.1]], dtype=np.float32), (3000, 1))
    res = hc.collect_pc(g, torch.from_numpy(pts).cuda(), LIMITS)
    np.testing.assert_array_equal(res.indices.cpu().numpy(), np.arange(3000))
    # clustered indices: 150 hits at the front and one at the far end of a large scene put more than 64 hits
    # into one bucket of the bucket sort (its fallback path)
    pts = np.full((300_000, 3), 2.0, dtype=np.float32)
    pts[:150] = [0.1, 0.1, 0.1]
    pts[-1] = [0.1, 0.1, 0.1]
    res = hc.collect_pc(g, torch.from_numpy(pts).cuda(), LIMITS)
    np.testing.assert_array_equal(res.counts.cpu().numpy(), [151, 0])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), np.r_[np.arange(150), 299_999])
    pts = np.tile(np.array([[0.1, 0.1, 0.1]], dtype=np.float32), (140_000, 1))     # one cell, two index slabs
    res = hc.collect_pc(g, torch.from_numpy(pts).cuda(), LIMITS)
    np.testing.assert_array_equal(res.counts.cpu().numpy(), [140_000, 0])
    np.testing.assert_array_equal(res.indices.cpu().numpy(), np.arange(140_000))
