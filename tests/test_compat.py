"""Checkpoint loading (pointnetgpd_amd.compat).  The legacy whole-module pickles are fabricated here
from the CPU restatement with the class path the reference's files carry (`model.pointnet.*`,
main_1v.py:350); the reference's own checkpoints are not available to this build."""
import collections
import os
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
    assert sorted(got.keys()) == sorted(want.keys())
    assert list(got.keys()) == list(pn.PointNetCls(750, 3, 2).state_dict().keys())      # rebuilt: native key order
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


def test_old_container_round_trip(tmp_path):
    """main_1v.py:350 writes torch's pre-1.6 non-zip container (`_use_new_zipfile_serialization=False`): whole
    modules, bare or wrapped in DataParallel, and state_dicts load from it."""
    m = pn.PointNetCls(32, 3, 2)
    m.load_state_dict(synth.torch_state_dict(9, 2, 1.0))
    for wrap in (False, True):
        path = str(tmp_path / f"old{int(wrap)}.model")
        compat.save_checkpoint(torch.nn.DataParallel(m) if wrap else m, path, old_container=True)
        import zipfile
        assert not zipfile.is_zipfile(path) and compat._is_legacy_container(path)
        back = compat.load_checkpoint(path, device="cpu")
        assert type(back) is pn.PointNetCls and type(back.feat.stn) is pn.STN3d and back.num_points == 32
        assert list(back.state_dict()) == list(m.state_dict())
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), back.state_dict().values()))
        with pytest.raises(ValueError, match="num_points=32"):
            compat.load_checkpoint(path, num_points=750, device="cpu")
    assert type(m) is pn.PointNetCls and "model" not in sys.modules
    sd_path = str(tmp_path / "old_sd.pth")
    torch.save(m.state_dict(), sd_path, _use_new_zipfile_serialization=False)
    back = compat.load_checkpoint(sd_path, num_points=32, device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), back.state_dict().values()))


class _Hostile:
    def __reduce__(self):
        import os
        return (os.getcwd, ())


def test_legacy_reader_resolves_nothing_outside_its_table(tmp_path):
    """The pre-1.6 container is read by an own restricted unpickler: a file that names any other callable or class
    is refused before anything is called, and a damaged file is reported."""
    import pickle
    import warnings
    bad = str(tmp_path / "hostile.model")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.save({"weights": torch.zeros(3), "x": _Hostile()}, bad, _use_new_zipfile_serialization=False)
    assert compat._is_legacy_container(bad)
    with pytest.raises(pickle.UnpicklingError, match="getcwd"):
        compat.load_checkpoint(bad, device="cpu")
    other = str(tmp_path / "other_module.model")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.save(torch.nn.Sequential(torch.nn.Tanh()), other, _use_new_zipfile_serialization=False)
    with pytest.raises(pickle.UnpicklingError, match="Sequential|Tanh"):
        compat.load_checkpoint(other, device="cpu")
    m = pn.PointNetCls(16, 3, 2)
    good = str(tmp_path / "good.model")
    compat.save_checkpoint(m, good, old_container=True)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.model")
    open(cut, "wb").write(blob[:len(blob) - 1000])
    with pytest.raises((ValueError, EOFError, pickle.UnpicklingError)):
        compat.load_checkpoint(cut, device="cpu")
    # a storage record that claims more bytes than the file holds is refused before anything is allocated:
    # the element count of the first storage (a BININT right after its location string) is raised to 2**30
    marker = b"cpu"
    at = blob.index(marker, blob.index(b"storage")) + len(marker)
    while blob[at:at + 1] not in (b"K", b"M", b"J"):          # skip the memo opcode to the count's BININT1/2/4
        at += 1
    width = {b"K": 1, b"M": 2, b"J": 4}[blob[at:at + 1]]
    huge = blob[:at] + b"J" + (2 ** 30).to_bytes(4, "little") + blob[at + 1 + width:]
    big = str(tmp_path / "huge.model")
    open(big, "wb").write(huge)
    with pytest.raises(pickle.UnpicklingError, match="more than the file holds"):
        compat.load_checkpoint(big, device="cpu")


def test_legacy_reader_checks_a_tensor_against_its_storage_before_building_it(tmp_path):
    """A tensor record whose shape reaches beyond its storage is refused before Tensor.set_ could grow the storage to
    fit it: a 12-byte storage with a declared shape of 2**28 floats must not allocate a gigabyte."""
    import pickle
    import warnings
    path = str(tmp_path / "small.model")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.save({"weights": torch.zeros(3)}, path, _use_new_zipfile_serialization=False)
    blob = open(path, "rb").read()
    at = blob.index(b"K\x03\x85", blob.index(b"_rebuild_tensor_v2"))       # the size tuple (3,)
    forged = blob[:at] + b"J" + (2 ** 28).to_bytes(4, "little") + blob[at + 2:]
    bad = str(tmp_path / "forged.model")
    open(bad, "wb").write(forged)
    with pytest.raises(pickle.UnpicklingError, match="does not fit its storage"):
        compat.load_checkpoint(bad, device="cpu")
    store = compat._LegacyStorage(torch.UntypedStorage(16), torch.float32)
    for args in ((0, (1000, 1000), (1000, 1)), (3, (2,), (1,)), (0, (2, 3), (4, 1)), (-1, (2,), (1,)), (0, (2,), (-1,))):
        with pytest.raises(pickle.UnpicklingError):
            compat._rebuild_tensor_v2(store, *args)
    assert store.untyped().nbytes() == 16
    assert compat._rebuild_tensor_v2(store, 1, (3,), (1,)).shape == (3,)
    assert compat._rebuild_tensor_v2(store, 0, (0, 99), (99, 1)).numel() == 0


def test_saved_modules_carry_no_process_local_state(tmp_path):
    """The cached tensor list that decides when device weights are repacked is process-local: it must not travel
    into whole-module files (which the reference's torch.load is meant to read) nor into deep copies."""
    import copy
    m = pn.PointNetCls(16, 3, 2)
    m._weights_signature()
    m.feat._weights_signature()
    assert "_sig_cache" in m.__dict__
    assert "_sig_cache" not in copy.deepcopy(m).__dict__
    for old in (True, False):
        path = str(tmp_path / f"after_forward_{old}.model")
        compat.save_checkpoint(m, path, old_container=old)
        assert b"_sig_cache" not in open(path, "rb").read()
        assert b"_hip" not in open(path, "rb").read()


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "model")),
                    reason="needs the reference checkout (build container only)")
def test_checkpoints_written_by_the_reference_classes_load_unchanged(tmp_path):
    """What main_1v.py:350 does, with the reference's own class imported from its checkout: torch.save(model, path,
    _use_new_zipfile_serialization=False), bare and DataParallel-wrapped (main_test.py:69 unwraps), plus the zip
    container.  All three load, and the reference's torch.load reads what save_checkpoint writes."""
    import subprocess
    import textwrap
    # the reference classes are imported in a child process so that `model.pointnet` never enters this one
    script = textwrap.dedent(f"""
        import sys, warnings
        sys.dont_write_bytecode = True
        sys.path.insert(0, {REFERENCE!r}); sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        import torch
        from model.pointnet import PointNetCls
        from pointnetgpd_amd import synth
        m = PointNetCls(750, 3, 2)
        m.load_state_dict(synth.torch_state_dict(5, 2, 0.02))
        warnings.simplefilter("ignore")
        torch.save(m, sys.argv[1] + "/ref_old.model", _use_new_zipfile_serialization=False)
        torch.save(torch.nn.DataParallel(m), sys.argv[1] + "/ref_old_dp.model", _use_new_zipfile_serialization=False)
        torch.save(m, sys.argv[1] + "/ref_zip.model")
        if len(sys.argv) > 2:                      # second call: read back what save_checkpoint wrote
            back = torch.load(sys.argv[2], map_location="cpu", weights_only=False)
            assert type(back).__module__ == "model.pointnet" and type(back) is PointNetCls
            x = torch.from_numpy(synth.candidate_points(5, 3, 750, "gripper"))
            a, _ = back.eval()(x); b, _ = m.eval()(x)
            assert torch.equal(a, b)
            print("reference-read-ok")
    """)
    subprocess.run([sys.executable, "-c", script, str(tmp_path)], check=True)
    want = synth.torch_state_dict(5, 2, 0.02)
    for name in ("ref_old.model", "ref_old_dp.model", "ref_zip.model"):
        back = compat.load_checkpoint(str(tmp_path / name), device="cpu")
        assert type(back) is pn.PointNetCls and back.num_points == 750 and not back.training
        got = back.state_dict()
        assert sorted(got) == sorted(want) and list(got) == list(pn.PointNetCls(750, 3, 2).state_dict())
        for k in want:
            assert torch.equal(got[k], want[k].to(got[k].dtype).reshape(got[k].shape)), k
    ours = str(tmp_path / "ours_old.model")
    m = pn.PointNetCls(750, 3, 2)
    m.load_state_dict(want)
    compat.save_checkpoint(m, ours, old_container=True)
    out = subprocess.run([sys.executable, "-c", script, str(tmp_path), ours], check=True, capture_output=True, text=True)
    assert "reference-read-ok" in out.stdout
