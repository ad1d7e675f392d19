"""CPU: the oracle restatement against golden vectors produced by the real reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EVAL_FILES = sorted(glob.glob(os.path.join(GOLDEN, "eval_*.npz")))


def load_case(path):
    d = np.load(path)
    P, k, seed, B = (int(v) for v in d["meta"])
    return d, P, k, seed, B, str(d["kind"]), float(d["input_scale"]), float(d["gain"])


def test_golden_files_present():
    assert len(EVAL_FILES) == 4


@pytest.mark.parametrize("path", EVAL_FILES, ids=[os.path.basename(p)[5:-4] for p in EVAL_FILES])
def test_oracle_matches_reference_eval(path):
    d, P, k, seed, B, kind, scale, gain = load_case(path)
    m = po.build_cls(P, k, synth.torch_state_dict(seed, k, scale, gain))
    x = torch.from_numpy(synth.candidate_points(seed, B, P, kind))
    with torch.no_grad():
        logp, trans = m(x)
        feat, _ = m.feat(x)
        pooled = m.feat.stn.pooled(x)
    # same ATen ops in the same order: only thread-count dependent summation differs
    np.testing.assert_allclose(logp.numpy(), d["logp"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(trans.numpy(), d["trans"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(feat.numpy(), d["feat"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pooled.numpy(), d["stn_pooled"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("path", EVAL_FILES[:2], ids=["gripper", "normal"])
def test_oracle_test_network_semantics(path):
    """main_test.py:73-83 applied row by row equals the stored probs / pred."""
    d, P, k, seed, B, kind, scale, gain = load_case(path)
    m = po.build_cls(P, k, synth.torch_state_dict(seed, k, scale, gain))
    x = synth.candidate_points(seed, B, P, kind)
    for i in range(3):
        pred, probs = po.oracle_test_network(m, x[i].T.astype(np.float64))
        assert probs.shape == (1, k)
        np.testing.assert_allclose(probs[0], d["probs"][i], atol=2e-5)
        assert int(pred[0]) == int(d["pred"][i])


def test_oracle_state_dict_keys():
    m = po.OraclePointNetCls(750, 3, 2)
    sd = synth.make_state_dict(1, 2)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert len(sd) == 74
    n_learn = sum(int(np.prod(v.shape)) for n, v in sd.items()
                  if not n.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_learn == 1_604_363


def test_oracle_rejects_wrong_point_count():
    m = po.build_cls(750, 2, synth.torch_state_dict(1, 2))
    with pytest.raises(Exception):
        with torch.no_grad():
            m(torch.zeros(2, 3, 500))
