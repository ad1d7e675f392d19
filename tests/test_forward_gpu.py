"""GPU parity of the HIP forward path (through the C ABI) against the reference's golden vectors
and against the CPU oracle on fresh seeded inputs.

Tolerances: exact-f32 MFMA path and split-bf16 path ("bf16x3": hi + lo operands, three bf16 MFMAs per product)
1e-4 absolute on log-probabilities and on the 3x3 transform; plain bf16 path 2e-2 absolute on log-probabilities;
arg-max agreement on candidates whose two best classes are further apart than the tolerance.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls, PointNetfeat, STN3d
from pointnetgpd_amd.scoring import score_candidates, test_network as hip_test_network

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EVAL_FILES = sorted(glob.glob(os.path.join(GOLDEN, "eval_*.npz")))
FP32_ATOL = 1e-4
BF16_ATOL = 2e-2
X3_ATOL = 1e-4          # operands exact to 2^-18: measured 6e-6 .. 2.1e-5 on the golden files
F32_GRADE = [("fp32", FP32_ATOL), ("bf16x3", X3_ATOL)]
ALL_MODES = F32_GRADE + [("bf16", BF16_ATOL)]


def hip_model(P, k, seed, scale, precision, gain=2.0):
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, scale, gain), strict=True)
    m = m.cuda().eval()
    m.precision = precision
    return m


def top1_agree(logp, ref_logp, tol):
    ref_sorted = np.sort(ref_logp, axis=1)
    decided = (ref_sorted[:, -1] - ref_sorted[:, -2]) > 2 * tol
    return (logp.argmax(1) == ref_logp.argmax(1))[decided].all(), int(decided.sum())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("path", EVAL_FILES, ids=[os.path.basename(p)[5:-4] for p in EVAL_FILES])
def test_fp32_matches_reference_golden(path, precision):
    d = np.load(path)
    P, k, seed, B = (int(v) for v in d["meta"])
    m = hip_model(P, k, seed, float(d["input_scale"]), precision, float(d["gain"]))
    x = torch.from_numpy(synth.candidate_points(seed, B, P, str(d["kind"]))).cuda()
    logp, trans = m(x)
    feat, trans2 = m.feat(x)
    stn_trans = m.feat.stn(x)
    torch.cuda.synchronize()
    err = np.abs(logp.cpu().numpy() - d["logp"]).max()
    print(f"{precision} max |dlogp| = {err:.3e}")
    assert err <= FP32_ATOL
    np.testing.assert_allclose(trans.cpu().numpy(), d["trans"], rtol=0, atol=FP32_ATOL)
    np.testing.assert_allclose(trans2.cpu().numpy(), d["trans"], rtol=0, atol=FP32_ATOL)
    np.testing.assert_allclose(stn_trans.cpu().numpy(), d["trans"], rtol=0, atol=FP32_ATOL)
    np.testing.assert_allclose(feat.cpu().numpy(), d["feat"], rtol=2e-5 if precision == "fp32" else 1e-4, atol=FP32_ATOL)
    ok, n = top1_agree(logp.cpu().numpy(), d["logp"], FP32_ATOL)
    assert ok and n > 0


@pytest.mark.parametrize("path", EVAL_FILES, ids=[os.path.basename(p)[5:-4] for p in EVAL_FILES])
def test_bf16_matches_reference_golden(path):
    d = np.load(path)
    P, k, seed, B = (int(v) for v in d["meta"])
    m = hip_model(P, k, seed, float(d["input_scale"]), "bf16", float(d["gain"]))
    x = torch.from_numpy(synth.candidate_points(seed, B, P, str(d["kind"]))).cuda()
    logp, trans = m(x)
    torch.cuda.synchronize()
    err = np.abs(logp.cpu().numpy() - d["logp"]).max()
    print(f"bf16 max |dlogp| = {err:.3e}, max |dtrans| = {np.abs(trans.cpu().numpy() - d['trans']).max():.3e}")
    assert err <= BF16_ATOL
    ok, _ = top1_agree(logp.cpu().numpy(), d["logp"], BF16_ATOL)
    assert ok


@pytest.mark.parametrize("B", [1, 3, 64])
@pytest.mark.parametrize("precision,tol", ALL_MODES)
def test_against_oracle_fresh_inputs(B, precision, tol):
    P, k, seed = 750, 2, 31
    sd = synth.torch_state_dict(seed, k, 0.02)
    oracle = po.build_cls(P, k, sd)
    m = hip_model(P, k, seed, 0.02, precision)
    x = torch.from_numpy(synth.candidate_points(seed + B, B, P, "gripper"))
    with torch.no_grad():
        ref_logp, ref_trans = oracle(x)
    logp, trans = m(x.cuda())
    err = (logp.cpu() - ref_logp).abs().max().item()
    print(f"{precision} B={B}: max |dlogp| = {err:.3e}")
    assert err <= tol
    assert (trans.cpu() - ref_trans).abs().max().item() <= (tol if precision != "bf16" else 5e-2)
    ok, _ = top1_agree(logp.cpu().numpy(), ref_logp.numpy(), tol)
    assert ok


RAGGED = [(1, 1), (2, 2500), (2, 15), (3, 16), (3, 17), (5, 127), (5, 128), (5, 129), (5, 255), (5, 256), (5, 257), (4, 513),
          (63, 40), (64, 40), (65, 40), (128, 33), (129, 33), (257, 48), (300, 300), (600, 33)]


@pytest.mark.parametrize("B,P", RAGGED, ids=[f"B{b}_P{p}" for b, p in RAGGED])
@pytest.mark.parametrize("precision,tol", ALL_MODES)
def test_ragged_sizes(B, P, precision, tol):
    """Point counts around the 16-point tile and 256-point batch edges (padding = duplicates of the
    last point) and candidate counts around the number of CUs (the bf16 kernel is persistent: one
    workgroup per CU strides over the candidates; bf16x3: two workgroups per candidate, batches of 128 points)."""
    k, seed = 2, 41
    sd = synth.torch_state_dict(seed, k, 0.02)
    oracle = po.build_cls(P, k, sd)
    m = hip_model(P, k, seed, 0.02, precision)
    x = torch.from_numpy(synth.candidate_points(seed + B + P, B, P, "gripper"))
    with torch.no_grad():
        ref_logp, ref_trans = oracle(x)
    logp, trans = m(x.cuda())
    err = (logp.cpu() - ref_logp).abs().max().item()
    assert err <= tol, err
    assert (trans.cpu() - ref_trans).abs().max().item() <= (tol if precision != "bf16" else 5e-2)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_points_layout_is_bitwise_equivalent(precision):
    m = hip_model(750, 2, 33, 1.0, precision, 1.7)
    x = torch.from_numpy(synth.candidate_points(33, 5, 750, "normal")).cuda()
    a, ta = m(x)
    b, tb = m.forward_points(x.transpose(1, 2).contiguous())
    assert torch.equal(a, b) and torch.equal(ta, tb)


def test_scoring_shim_matches_test_network_semantics():
    """Row i of the batched score equals main_test.py:73-83 applied to candidate i."""
    d = np.load(EVAL_FILES[0])
    P, k, seed, B = (int(v) for v in d["meta"])
    m = hip_model(P, k, seed, float(d["input_scale"]), "fp32", float(d["gain"]))
    x = synth.candidate_points(seed, B, P, str(d["kind"]))
    pts = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    pred, probs, _ = score_candidates(m, pts)
    np.testing.assert_allclose(probs.cpu().numpy(), d["probs"], atol=FP32_ATOL)
    np.testing.assert_array_equal(pred.cpu().numpy(), d["pred"])
    p1, pr1 = hip_test_network(m, x[2].T.astype(np.float64))
    assert pr1.shape == (1, k) and int(p1[0]) == int(d["pred"][2])
    np.testing.assert_allclose(pr1[0], d["probs"][2], atol=FP32_ATOL)


def test_wrong_point_count_is_rejected():
    m = hip_model(750, 2, 1, 1.0, "fp32")
    with pytest.raises(ValueError, match="num_points"):
        m(torch.zeros(2, 3, 500, device="cuda"))


def test_weights_repacked_after_in_place_update():
    m = hip_model(750, 2, 35, 1.0, "fp32")
    x = torch.from_numpy(synth.candidate_points(35, 2, 750, "normal")).cuda()
    a, _ = m(x)
    with torch.no_grad():
        m.fc3.bias.add_(torch.tensor([0.5, -0.5], device="cuda"))
    b, _ = m(x)
    assert not torch.equal(a, b)


def test_weights_repacked_after_storage_swap():
    """`param.data = t` (vector_to_parameters, EMA / SWA swaps) gives a parameter another storage without bumping its
    version counter; the next eval forward must score with the new values."""
    m = hip_model(750, 2, 35, 1.0, "fp32")
    x = torch.from_numpy(synth.candidate_points(35, 2, 750, "normal")).cuda()
    a, _ = m(x)
    version = m.fc3.weight._version
    m.fc3.weight.data = (m.fc3.weight.data * 0.5).clone()
    assert m.fc3.weight._version == version          # nothing but the address tells
    b, _ = m(x)
    assert not torch.equal(a, b)
    flat = torch.nn.utils.parameters_to_vector(m.parameters()).detach().clone()
    torch.nn.utils.vector_to_parameters(flat * 1.01, m.parameters())
    c, _ = m(x)
    assert not torch.equal(b, c)
    oracle = po.build_cls(750, 2, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        ref, _ = oracle(x.cpu())
    assert (c.cpu() - ref).abs().max().item() <= FP32_ATOL


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_full_size_properties(precision):
    """B=4096, P=750 in every arithmetic mode: results do not depend on batch position, batch size or
    on the order of the points inside a candidate (the max-pool is symmetric) - bit for bit."""
    B = 4096
    P, seed = 750, 41
    m = hip_model(P, 2, seed, 0.02, precision)
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "gripper")).cuda()
    logp, trans = m(x)
    assert torch.isfinite(logp).all()
    assert torch.allclose(logp.exp().sum(1), torch.ones(B, device="cuda"), atol=1e-5)
    idx = torch.tensor([0, 1, 255, 256, 1023, B // 2, B // 2 + 1, 3000, 3071, 3072, 4000, B - 2, B - 1], device="cuda")
    sub, sub_t = m(x[idx].contiguous())
    assert torch.equal(sub, logp[idx]) and torch.equal(sub_t, trans[idx])
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(0)).cuda()
    shuf, _ = m(x[idx][:, :, perm].contiguous())
    assert torch.equal(shuf, logp[idx])
    # oracle on a handful of rows of the big batch
    oracle = po.build_cls(P, 2, synth.torch_state_dict(seed, 2, 0.02))
    with torch.no_grad():
        ref, _ = oracle(x[idx].cpu())
    tol = BF16_ATOL if precision == "bf16" else FP32_ATOL
    assert (logp[idx].cpu() - ref).abs().max().item() <= tol


def test_standalone_submodules():
    sd = synth.torch_state_dict(37, 2, 1.0, 1.7)
    x = torch.from_numpy(synth.candidate_points(37, 4, 750, "normal"))
    oracle = po.build_cls(750, 2, sd)
    with torch.no_grad():
        ref_feat, ref_trans = oracle.feat(x)
    stn = STN3d(750, 3)
    stn.load_state_dict({k[len("feat.stn."):]: v for k, v in sd.items() if k.startswith("feat.stn.")})
    feat = PointNetfeat(750, 3)
    feat.load_state_dict({k[len("feat."):]: v for k, v in sd.items() if k.startswith("feat.")})
    t = stn.cuda().eval()(x.cuda())
    f, t2 = feat.cuda().eval()(x.cuda())
    assert (t.cpu() - ref_trans).abs().max().item() <= FP32_ATOL
    assert (t2.cpu() - ref_trans).abs().max().item() <= FP32_ATOL
    np.testing.assert_allclose(f.cpu().numpy(), ref_feat.numpy(), rtol=2e-5, atol=FP32_ATOL)


def test_config3_full_size_p500_sharded_batch():
    """B = 65,536 candidates of P = 500 points in bf16 (the largest batched configuration): finite,
    normalised, and any contiguous shard scored alone gives bit-identical rows (what makes the
    multi-GPU split of the batch exact)."""
    B, P, seed = 65_536, 500, 43
    m = hip_model(P, 2, seed, 0.02, "bf16", 1.7)
    g = torch.Generator(device="cuda").manual_seed(seed)
    hd, ow, hh = synth.GRIPPER_BOX
    x = torch.rand((B, 3, P), generator=g, device="cuda")
    x = x * torch.tensor([hd, ow, hh], device="cuda").view(1, 3, 1) + torch.tensor([0.0, -ow / 2, -hh / 2], device="cuda").view(1, 3, 1)
    logp, trans = m(x)
    assert torch.isfinite(logp).all() and torch.isfinite(trans).all()
    assert torch.allclose(logp.exp().sum(1), torch.ones(B, device="cuda"), atol=1e-5)
    for lo, hi in ((0, 8192), (57_344, 65_536), (30_000, 30_003)):
        part, part_t = m(x[lo:hi].contiguous())
        assert torch.equal(part, logp[lo:hi]) and torch.equal(part_t, trans[lo:hi])
    oracle = po.build_cls(P, 2, synth.torch_state_dict(seed, 2, 0.02, 1.7))
    pick = torch.randperm(B, generator=torch.Generator().manual_seed(seed))[:253]
    idx = torch.cat([torch.tensor([0, 12_345, B - 1]), pick]).cuda()          # 256 rows spread over the whole batch
    with torch.no_grad():
        ref, ref_t = oracle(x[idx].cpu())
    assert (logp[idx].cpu() - ref).abs().max().item() <= BF16_ATOL
    # the same rows in the float32-grade default mode, scored as a batch of their own
    m.precision = "bf16x3"
    sub, sub_t = m(x[idx].contiguous())
    assert (sub.cpu() - ref).abs().max().item() <= X3_ATOL and (sub_t.cpu() - ref_t).abs().max().item() <= X3_ATOL


def test_p1000_full_view_variant():
    """num_points = 1000 is what the reference's full-view scripts build (main_fullv.py:51,106)."""
    P, seed = 1000, 45
    sd = synth.torch_state_dict(seed, 2, 1.0, 1.7)
    x = torch.from_numpy(synth.candidate_points(seed, 3, P, "normal"))
    with torch.no_grad():
        ref, _ = po.build_cls(P, 2, sd)(x)
    for precision, tol in ALL_MODES:
        m = PointNetCls(P, 3, 2)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        m.precision = precision
        logp, _ = m(x.cuda())
        assert (logp.cpu() - ref).abs().max().item() <= tol


def test_bf16_error_against_logit_magnitude():
    """The 2e-2 absolute bar of the bf16 path is met on networks whose class scores differ by a few units; the error
    itself is proportional to the scores (bf16 rounds h1, h2 and the weights to 2^-9 relative).  Networks of growing
    weight gain show it: the error stays below 1 % of the largest score difference at every gain, the absolute bar
    fails beyond a score difference of about 5, and the exact-f32 mode holds its 1e-4 relative to the same scale.
    The split-bf16 mode (operands exact to 2^-18) stays within 4e-5 of the score difference, so it meets the 2e-2 absolute
    bar up to differences of several hundred (asserted at 77) and 1e-4 where plain bf16 meets 2e-2."""
    P, k, seed, B = 750, 2, 47, 64
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "gripper"))
    rows = []
    for gain in (1.5, 2.0, 2.5, 3.0, 3.5):
        sd = synth.torch_state_dict(seed, k, 0.02, gain)
        with torch.no_grad():
            ref = po.build_cls(P, k, sd)(x)[0].double()
        spread = (ref.max(1).values - ref.min(1).values).max().item()       # = largest |logit_0 - logit_1|
        errs = {}
        for precision in ("bf16", "fp32", "bf16x3"):
            logp = hip_model(P, k, seed, 0.02, precision, gain)(x.cuda())[0].cpu().double()
            errs[precision] = (logp - ref).abs().max().item()
        rows.append((gain, spread, errs["bf16"], errs["fp32"], errs["bf16x3"]))
        print(f"gain {gain}: largest score difference {spread:8.3f}  bf16 err {errs['bf16']:.3e} "
              f"({100 * errs['bf16'] / max(spread, 1.0):.2f} %)  fp32 err {errs['fp32']:.3e}  bf16x3 err {errs['bf16x3']:.3e}")
    for gain, spread, e16, e32, ex3 in rows:
        assert e16 <= 0.01 * max(spread, 2.0), (gain, spread, e16)
        assert e32 <= 1e-4 * max(1.0, spread / 5.0), (gain, spread, e32)
        assert ex3 <= max(1e-4, 4e-5 * spread), (gain, spread, ex3)
        if spread <= 500.0:
            assert ex3 <= BF16_ATOL, (gain, spread, ex3)
    assert rows[-1][1] > 5.0            # the sweep does reach scores where the absolute bar no longer applies
    assert any(77.0 <= r[1] <= 500.0 for r in rows)     # and one where only the float32-grade modes hold 2e-2


def test_main_test_command_line(tmp_path, capsys):
    """The command line of main_test.py on a checkpoint in the reference's container: ten draws of num_points points
    from one cloud, scored in one batch, majority class printed; the classes equal per-draw calls of test_network."""
    from pointnetgpd_amd import compat, scoring
    P, k, seed = 500, 3, 61
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, 0.02), strict=True)
    path = str(tmp_path / "pointgpd_3class.model")
    compat.save_checkpoint(m, path, old_container=True)
    cloud = synth.candidate_points(seed, 1, 900, "gripper")[0].T.copy()          # [900,3]
    np.save(tmp_path / "cloud.npy", cloud)
    result = scoring.main(["--load-model", path, "--points", str(tmp_path / "cloud.npy")])
    out = capsys.readouterr().out
    assert "Test result: %d" % result in out and "voting:" in out
    model = compat.load_checkpoint(path)
    winner, votes = scoring.vote_on_cloud(model, cloud, repeat=10, seed=5)
    rng = np.random.default_rng(5)
    c = cloud.astype(np.float32)
    singles = []
    for _ in range(10):
        c = c[rng.choice(len(c), P, replace=len(c) < P)]
        singles.append(int(scoring.test_network(model, c)[0][0]))
    assert list(votes) == singles and votes.shape == (10,)
    counts = np.bincount(votes, minlength=k)
    assert counts[winner] == counts.max() and winner == int(np.argmax(counts))


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("P", [750, 500, 241, 1000, 2500, 16])
def test_small_batches_cut_over_points_are_bitwise_equal_to_rows_of_a_large_batch(precision, P):
    """A batch of at most 8 candidates (main_test.py:73-83 scores one per call) runs the conv stacks cut over channel
    pieces AND point groups, with the FC layer behind each stack merging the groups' partial maxima.  A maximum does
    not depend on how its operands are grouped, so every output - log-probabilities, transform and the merged global
    feature - must equal, bit for bit, the same candidates' rows of a batch of 64 (which is not cut over points)."""
    seed = 23
    m = hip_model(P, 2, seed, 0.02, precision)
    x = torch.from_numpy(synth.candidate_points(seed, 64, P, "gripper")).cuda()
    logp, trans, feat = m._infer(x, 0, True, True)
    for B in (1, 2, 3, 8):
        rows = torch.arange(5, 5 + B, device="cuda")
        sub_logp, sub_trans, sub_feat = m._infer(x[rows].contiguous(), 0, True, True)
        assert torch.equal(sub_logp, logp[rows]), (precision, P, B)
        assert torch.equal(sub_trans, trans[rows]), (precision, P, B)
        assert torch.equal(sub_feat, feat[rows]), (precision, P, B)
    # modules without a head: the trunk's stack has no FC layer behind it and is not cut over points; the STN's is
    sd = {k[len("feat."):]: v for k, v in synth.torch_state_dict(seed, 2, 0.02).items() if k.startswith("feat.")}
    f = PointNetfeat(P, 3)
    f.load_state_dict(sd)
    f = f.cuda().eval()
    f.precision = precision
    g64, t64 = f(x)
    g1, t1 = f(x[7:8].contiguous())
    assert torch.equal(g1, g64[7:8]) and torch.equal(t1, t64[7:8])
    assert torch.equal(g64, feat) and torch.equal(t64, trans)


@pytest.mark.parametrize("k,seed", [(2, 41), (3, 42)])
def test_weights_folded_on_the_device_give_the_model_folded_on_the_host_bit_for_bit(monkeypatch, k, seed):
    """Parameters that already live on the GPU are folded and packed there (pngpd_model_load_device: no copy to the
    host, no wait); PNGPD_HOST_FOLD=1 sends them through the host loader instead.  Same float64 operations in the same
    order: log-probabilities, transform and global feature of the two models are equal bit for bit in all three
    arithmetic modes, for the classifier and for the two stand-alone modules, also with a non-contiguous and a
    half-precision parameter in the table."""
    P, B = 750, 19
    sd = synth.torch_state_dict(seed, k, 1.0, 2.0)
    x = torch.from_numpy(synth.candidate_points(seed, B, P)).cuda()
    outs = {}
    for how in ("host", "device"):
        if how == "host":
            monkeypatch.setenv("PNGPD_HOST_FOLD", "1")
        else:
            monkeypatch.delenv("PNGPD_HOST_FOLD", raising=False)
        m = PointNetCls(P, 3, k)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        stn, feat = STN3d(P, 3), PointNetfeat(P, 3)
        stn.load_state_dict({n[len("feat.stn."):]: v for n, v in sd.items() if n.startswith("feat.stn.")})
        feat.load_state_dict({n[len("feat."):]: v for n, v in sd.items() if n.startswith("feat.")})
        stn, feat = stn.cuda().eval(), feat.cuda().eval()
        # a non-contiguous weight (a transposed view of the transposed storage) and a half-precision BatchNorm scale
        w = m.fc2.weight.data
        m.fc2.weight.data = w.t().contiguous().t()
        assert not m.fc2.weight.is_contiguous()
        m.bn1.weight.data = m.bn1.weight.data.half()
        got = []
        for precision in ("fp32", "bf16x3", "bf16"):
            for mod in (m, stn, feat):
                mod.precision = precision
            logp, trans = m(x)
            got += [logp, trans, stn(x), feat(x)[0]]
        torch.cuda.synchronize()
        outs[how] = [t.cpu().numpy().copy() for t in got]
    for a, b in zip(outs["host"], outs["device"]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


def test_device_weight_table_with_a_host_pointer_is_refused():
    """A host address in the device table would fault the GPU inside a kernel: every pointer is looked up first."""
    import ctypes
    from pointnetgpd_amd import _lib
    P, k = 64, 2
    m = hip_model(P, k, 43, 1.0, "fp32")
    m(torch.zeros(2, 3, P, device="cuda"))                      # creates the handle
    h = m._handle(torch.device("cuda", torch.cuda.current_device()))
    table = (ctypes.c_void_p * _lib.SLOTS)()
    host = torch.zeros(1 << 20)
    for i in range(_lib.SLOTS):
        table[i] = host.data_ptr()
    rc = h.lib.pngpd_model_load_device(h.ptr, table, _lib.SLOTS, None)
    assert rc == 1, rc                                           # PNGPD_ERR_INVALID
    assert b"device" in h.lib.pngpd_last_error()
    # the refused call launched nothing and left the model as it was
    logp, _ = m(torch.zeros(2, 3, P, device="cuda"))
    torch.cuda.synchronize()
    assert torch.isfinite(logp).all()
