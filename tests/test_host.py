"""CPU: host-side logic - ABI surface, synthetic data determinism, module/state_dict contract."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from pointnetgpd_amd import _lib, synth
from pointnetgpd_amd.pointnet import PointNetCls, PointNetfeat, STN3d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(pngpd_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 8
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table out of sync with include/pngpd.h"


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.pngpd_abi_version() >= 1
    assert isinstance(lib.pngpd_last_error(), bytes)


def test_synth_is_deterministic_and_shard_invariant():
    a = synth.candidate_points(5, 6, 100, "gripper")
    b = synth.candidate_points(5, 3, 100, "gripper", first=3)
    np.testing.assert_array_equal(a[3:], b)
    hd, ow, hh = synth.GRIPPER_BOX
    assert a[:, 0].min() > 0 and a[:, 0].max() < hd
    assert np.abs(a[:, 1]).max() < ow / 2 and np.abs(a[:, 2]).max() < hh / 2
    n = synth.candidate_points(5, 64, 750, "normal")
    assert abs(float(n.mean())) < 0.3 and np.isfinite(n).all()
    w1 = synth.make_state_dict(3, 2)["fc1.weight"]
    w2 = synth.make_state_dict(3, 2)["fc1.weight"]
    np.testing.assert_array_equal(w1, w2)
    assert not np.array_equal(w1, synth.make_state_dict(4, 2)["fc1.weight"])


def test_module_contract_matches_reference_layout():
    m = PointNetCls(num_points=750, input_chann=3, k=2)
    sd = m.state_dict()
    assert len(sd) == 74
    assert list(sd)[:4] == ["feat.stn.conv1.weight", "feat.stn.conv1.bias", "feat.stn.conv2.weight",
                            "feat.stn.conv2.bias"]
    assert sd["feat.conv3.weight"].shape == (1024, 128, 1)
    assert sd["fc3.weight"].shape == (2, 256)
    m.load_state_dict(synth.torch_state_dict(1, 2), strict=True)
    assert len(list(m.parameters())) == 44
    assert PointNetCls(500, 3, 3).fc3.out_features == 3
    assert isinstance(m.feat, PointNetfeat) and isinstance(m.feat.stn, STN3d)


def test_no_cpu_fallback():
    m = PointNetCls(750, 3, 2).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 3, 750))


def test_shape_errors_before_any_launch():
    m = PointNetCls(750, 3, 2).eval()
    with pytest.raises(ValueError):
        m._check_input(torch.zeros(2, 3), _lib.LAYOUT_C3P)


def test_pickle_and_deepcopy_drop_device_handles():
    import copy
    import pickle
    m = PointNetCls(64, 3, 2)
    m.__dict__["_hip"] = object()
    c = copy.deepcopy(m)
    assert "_hip" not in c.__dict__
    r = pickle.loads(pickle.dumps(m))
    assert "_hip" not in r.__dict__ and list(r.state_dict()) == list(m.state_dict())
