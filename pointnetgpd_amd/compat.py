"""Loading PointNetGPD checkpoints into the MI355X modules.

The reference saves the WHOLE module with `torch.save(model, path)` (main_1v.py:350; the model may be
wrapped in nn.DataParallel, :207) and loads it back with `torch.load(path)` (main_1v.py:195-197,
main_test.py:48-56), so its checkpoints are pickles that name the classes `model.pointnet.PointNetCls`,
`model.pointnet.PointNetfeat` and `model.pointnet.STN3d`.  `load_checkpoint` reads

  * such whole-module pickles, through torch's restricted (`weights_only=True`) unpickler with an
    allow-list: the three class names above are mapped onto the classes of pointnetgpd_amd.pointnet
    (their attribute layout is the reference's, so the pickled `__dict__` fits), the torch.nn layers a
    PointNet contains and nn.DataParallel (unwrapped).  Nothing else in the file can be instantiated;
  * plain `state_dict` files (with or without the `module.` prefix of DataParallel).

Files in torch's pre-1.6 non-zip container are refused with an explanation (load_checkpoint.__doc__);
`save_checkpoint` writes whole-module pickles under the reference's class path that both sides can read.

The shipped checkpoints of the reference are not available to this build, so this is verified on
pickles of the same structure made by the tests, not on the originals.
"""
from __future__ import annotations

import collections
import sys
import types

import torch
from torch import nn

from . import pointnet as _pn

_LEGACY_MODULE = "model.pointnet"


def _legacy_classes():
    """Subclasses that carry the reference's class path; the unpickler's allow-list is keyed by it."""
    out = {}
    for name in ("STN3d", "PointNetfeat", "PointNetCls"):
        base = getattr(_pn, name)
        out[name] = type(name, (base,), {"__module__": _LEGACY_MODULE, "__doc__": base.__doc__})
    return out


def _to_native(module: nn.Module) -> nn.Module:
    """Re-class the legacy-path instances (recursively) as the classes of pointnetgpd_amd.pointnet."""
    for m in module.modules():
        for name in ("STN3d", "PointNetfeat", "PointNetCls"):
            if type(m).__module__ == _LEGACY_MODULE and type(m).__name__ == name:
                m.__class__ = getattr(_pn, name)
    return module


def _from_state_dict(sd, num_points):
    sd = collections.OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
    if "fc3.weight" not in sd or "feat.stn.conv1.weight" not in sd:
        raise ValueError("state_dict does not look like a PointNetCls (missing fc3.weight / feat.stn.conv1.weight)")
    if num_points is None:
        raise ValueError("a state_dict does not record num_points: pass num_points=750 (or 500 / 1000)")
    k = sd["fc3.weight"].shape[0]
    cin = sd["feat.conv1.weight"].shape[1]
    model = _pn.PointNetCls(num_points=num_points, input_chann=cin, k=k)
    model.load_state_dict(sd, strict=True)
    return model


class _LegacyPath:
    """While active, `model.pointnet.{STN3d,PointNetfeat,PointNetCls}` resolve to subclasses of this package's
    modules - for nothing but one load or save."""

    def __enter__(self):
        self.classes = _legacy_classes()
        self._saved = {k: sys.modules.get(k) for k in ("model", _LEGACY_MODULE)}
        pkg, mod = types.ModuleType("model"), types.ModuleType(_LEGACY_MODULE)
        for name, cls in self.classes.items():
            setattr(mod, name, cls)
        pkg.pointnet = mod
        sys.modules["model"], sys.modules[_LEGACY_MODULE] = pkg, mod
        return self

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        return False


def save_checkpoint(model, path, old_container=False):
    """Whole-module pickle (`torch.save(model, path)`, main_1v.py:350) under the reference's class path
    `model.pointnet.PointNetCls`: the file loads with `load_checkpoint` and, since the attribute layout is the
    reference's, with the reference's own `torch.load` (main_1v.py:201, main_test.py:56).

    old_container=True writes torch's pre-1.6 non-zip container, which is what the reference asks for
    (`_use_new_zipfile_serialization=False`); `load_checkpoint` cannot read that one back (see there), the
    reference can."""
    natives = [(m, type(m)) for m in model.modules() if type(m) in (_pn.STN3d, _pn.PointNetfeat, _pn.PointNetCls)]
    with _LegacyPath() as lp:
        try:
            for m, cls in natives:
                m.__class__ = lp.classes[cls.__name__]
            torch.save(model, path, _use_new_zipfile_serialization=not old_container)
        finally:
            for m, cls in natives:
                m.__class__ = cls


def load_checkpoint(path, num_points=None, device="cuda"):
    """Returns a pointnetgpd_amd.pointnet.PointNetCls in eval mode on `device`.

    num_points is only needed for state_dict files (a whole-module pickle carries it).

    State_dict files load from either container.  Whole-module pickles in torch's pre-1.6 NON-zip container (`_use_new_zipfile_serialization=False`, which
    main_1v.py:350 asks for) are refused: that container records module classes through pickle persistent ids,
    which torch's restricted unpickler rejects, and this loader does not fall back to unrestricted unpickling.
    Re-save such a file where it was made (`torch.save(model.state_dict(), path)` or a plain `torch.save(model,
    path)`), or point `load_checkpoint` at the state_dict."""
    import pickle
    import zipfile
    with _LegacyPath() as lp:
        allowed = list(lp.classes.values()) + [_pn.STN3d, _pn.PointNetfeat, _pn.PointNetCls, nn.Conv1d, nn.BatchNorm1d,
                                               nn.Linear, nn.MaxPool1d, nn.ReLU, nn.DataParallel, nn.Dropout,
                                               collections.OrderedDict, torch.device]
        try:
            with torch.serialization.safe_globals(allowed):
                obj = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            if not zipfile.is_zipfile(path) and "persistent_load" in str(e):
                raise ValueError(
                    f"{path} holds a whole module in torch's pre-1.6 non-zip container; it can only be read by "
                    "unrestricted unpickling, which this loader does not do. Re-save it with "
                    "torch.save(model.state_dict(), ...) or torch.save(model, ...) in the environment that wrote "
                    "it.") from None
            raise
    if isinstance(obj, nn.DataParallel):
        obj = obj.module
    if isinstance(obj, dict):
        if "state_dict" in obj and isinstance(obj["state_dict"], dict):
            obj = obj["state_dict"]
        model = _from_state_dict(obj, num_points)
    elif isinstance(obj, nn.Module):
        model = _to_native(obj)
        if not isinstance(model, _pn.PointNetCls):
            raise ValueError(f"checkpoint holds a {type(model).__name__}, expected PointNetCls")
        if num_points is not None and int(model.num_points) != int(num_points):
            raise ValueError(f"checkpoint was built with num_points={model.num_points}, asked for {num_points}")
    else:
        raise ValueError(f"unsupported checkpoint content: {type(obj).__name__}")
    return model.to(device).eval()
