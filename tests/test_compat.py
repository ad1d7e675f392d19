"""Checkpoint loading (pointnetgpd_amd.compat).  The legacy whole-module pickles are fabricated here
from the CPU restatement with the class path the reference's files carry (`model.pointnet.*`,
main_1v.py:350); the reference's own checkpoints are not available to this build."""
import collections
import sys
import types

import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from pointnetgpd_amd import compat, synth
from pointnetgpd_amd import pointnet as pn


@pytest.fixture
def legacy_pickle(tmp_path):
    """torch.save(model) of a restated PointNetCls whose classes claim to live in model.pointnet."""
    def make(P, k, seed, data_parallel=False):
        m = po.build_cls(P, k, synth.torch_state_dict(seed, k, 0.02))
        classes = [type(m), type(m.feat), type(m.feat.stn)]
        old = [c.__module__ for c in classes]
        old_names = [c.__qualname__ for c in classes]
        pkg, mod = types.ModuleType("model"), types.ModuleType("model.pointnet")
        for c, name in zip(classes, ("PointNetCls", "PointNetfeat", "STN3d")):
            c.__module__, c.__qualname__ = "model.pointnet", name
            setattr(mod, name, c)
        pkg.pointnet = mod
        sys.modules["model"], sys.modules["model.pointnet"] = pkg, mod
        path = str(tmp_path / f"legacy_{P}_{k}_{int(data_parallel)}.model")
        try:
            torch.save(torch.nn.DataParallel(m) if data_parallel else m, path)
        finally:
            for c, mo, qn in zip(classes, old, old_names):
                c.__module__, c.__qualname__ = mo, qn
            sys.modules.pop("model", None)
            sys.modules.pop("model.pointnet", None)
        return path, m
    return make


@pytest.mark.parametrize("data_parallel", [False, True])
def test_whole_module_pickle_maps_onto_native_classes(legacy_pickle, data_parallel):
    path, ref = legacy_pickle(750, 2, 5, data_parallel)
    m = compat.load_checkpoint(path, device="cpu")
    assert type(m) is pn.PointNetCls and type(m.feat) is pn.PointNetfeat and type(m.feat.stn) is pn.STN3d
    assert m.num_points == 750 and not m.training and m.fc3.out_features == 2
    got, want = m.state_dict(), ref.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert "model" not in sys.modules and "model.pointnet" not in sys.modules


def test_state_dict_files(tmp_path):
    sd = synth.torch_state_dict(7, 3, 0.02)
    p1, p2 = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    torch.save(sd, p1)
    torch.save(collections.OrderedDict(("module." + k, v) for k, v in sd.items()), p2)
    for p in (p1, p2):
        m = compat.load_checkpoint(p, num_points=500, device="cpu")
        assert m.num_points == 500 and m.fc3.out_features == 3
        assert torch.equal(m.state_dict()["feat.stn.fc3.weight"], sd["feat.stn.fc3.weight"])
    with pytest.raises(ValueError, match="num_points"):
        compat.load_checkpoint(p1, device="cpu")


def test_wrong_num_points_and_foreign_pickles_are_refused(legacy_pickle, tmp_path):
    path, _ = legacy_pickle(500, 2, 6)
    with pytest.raises(ValueError, match="num_points=500"):
        compat.load_checkpoint(path, num_points=750, device="cpu")
    bad = str(tmp_path / "bad.model")
    torch.save(types.SimpleNamespace(x=1), bad)               # a class outside the allow-list
    with pytest.raises(Exception):
        compat.load_checkpoint(bad, device="cpu")


@pytest.mark.gpu
def test_loaded_legacy_checkpoint_scores_like_the_restatement(legacy_pickle):
    path, ref = legacy_pickle(750, 2, 8)
    m = compat.load_checkpoint(path)                          # cuda, eval
    x = torch.from_numpy(synth.candidate_points(8, 6, 750, "gripper"))
    with torch.no_grad():
        want, _ = ref.eval()(x)
    got, _ = m(x.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)


def test_save_checkpoint_round_trip_and_class_path(tmp_path):
    """Whole-module pickles written here carry the reference's class path (so the reference's torch.load resolves
    them to ITS classes) and read back bit-exactly; the model object is left untouched."""
    m = pn.PointNetCls(64, 3, 3)
    m.load_state_dict(synth.torch_state_dict(4, 3, 1.0))
    path = str(tmp_path / "saved.model")
    compat.save_checkpoint(m, path)
    assert type(m) is pn.PointNetCls and type(m.feat.stn) is pn.STN3d
    assert "model" not in sys.modules and "model.pointnet" not in sys.modules
    import zipfile
    with zipfile.ZipFile(path) as z:
        pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"model.pointnet" in pkl and b"pointnetgpd_amd" not in pkl
    back = compat.load_checkpoint(path, device="cpu")
    assert type(back) is pn.PointNetCls and back.num_points == 64 and back.fc3.out_features == 3
    for (k, a), (_, b) in zip(m.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a, b), k


def test_old_container_is_refused_with_an_explanation(tmp_path):
    """main_1v.py:350 writes torch's pre-1.6 non-zip container; torch's restricted unpickler cannot read whole
    modules from it and the loader does not fall back to unrestricted unpickling."""
    import warnings
    m = pn.PointNetCls(32, 3, 2)
    path = str(tmp_path / "old.model")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        compat.save_checkpoint(m, path, old_container=True)
    with pytest.raises(ValueError, match="non-zip container"):
        compat.load_checkpoint(path, device="cpu")
    # a state_dict in the old container holds tensors only and loads
    sd_path = str(tmp_path / "old_sd.pth")
    torch.save(m.state_dict(), sd_path, _use_new_zipfile_serialization=False)
    back = compat.load_checkpoint(sd_path, num_points=32, device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), back.state_dict().values()))
