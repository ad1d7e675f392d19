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

Whole modules in torch's pre-1.6 non-zip container - what main_1v.py:350 writes, since it passes
`_use_new_zipfile_serialization=False` - are read by `_load_legacy_container`, an own restricted reader of that
container (torch's restricted unpickler rejects its module records): the same allow-list, nothing else can be
resolved or called.  No file is ever unpickled without restrictions.
`save_checkpoint` writes whole-module pickles under the reference's class path, in either container, that both
sides can read.

The shipped checkpoints of the reference are not available to this build, so this is verified on
pickles of the same structure made by the tests, not on the originals.
"""
from __future__ import annotations

import collections
import os
import pickle
import struct
import sys
import types
import warnings
import zipfile

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


def _collect_tensors(module, prefix, out):
    """state_dict of an unpickled module tree, read from the three dictionaries every torch version keeps
    (`_parameters`, `_buffers`, `_modules`) instead of through methods whose needs changed between versions."""
    d = module.__dict__
    for n, p in (d.get("_parameters") or {}).items():
        if p is not None:
            out[prefix + n] = p.detach()
    for n, b in (d.get("_buffers") or {}).items():
        if b is not None:
            out[prefix + n] = b.detach()
    for n, m in (d.get("_modules") or {}).items():
        if m is not None:
            _collect_tensors(m, prefix + n + ".", out)
    return out


def _to_native(module: nn.Module) -> nn.Module:
    """A fresh pointnetgpd_amd.pointnet.PointNetCls holding the tensors of an unpickled legacy-path PointNetCls.
    Rebuilding (rather than re-classing the unpickled object) makes the result independent of the attribute layout
    of the torch version that wrote the file; BatchNorm layers saved before torch 0.4.1 have no batch counter."""
    if not (type(module).__module__ == _LEGACY_MODULE and type(module).__name__ == "PointNetCls") and \
            not isinstance(module, _pn.PointNetCls):
        raise ValueError(f"checkpoint holds a {type(module).__name__}, expected PointNetCls")
    sd = _collect_tensors(module, "", collections.OrderedDict())
    num_points = module.__dict__.get("num_points")
    if num_points is None:
        raise ValueError("the pickled PointNetCls records no num_points")
    model = _from_state_dict(sd, int(num_points), allow_missing_counters=True)
    return model


def _from_state_dict(sd, num_points, allow_missing_counters=False):
    sd = collections.OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
    if "fc3.weight" not in sd or "feat.stn.conv1.weight" not in sd:
        raise ValueError("state_dict does not look like a PointNetCls (missing fc3.weight / feat.stn.conv1.weight)")
    if num_points is None:
        raise ValueError("a state_dict does not record num_points: pass num_points=750 (or 500 / 1000)")
    k = sd["fc3.weight"].shape[0]
    cin = sd["feat.conv1.weight"].shape[1]
    model = _pn.PointNetCls(num_points=num_points, input_chann=cin, k=k)
    if allow_missing_counters:
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if unexpected or any(not m.endswith("num_batches_tracked") for m in missing):
            raise ValueError(f"checkpoint does not match PointNetCls: missing {missing}, unexpected {unexpected}")
    else:
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
    (`_use_new_zipfile_serialization=False`, main_1v.py:350)."""
    natives = [(m, type(m)) for m in model.modules() if type(m) in (_pn.STN3d, _pn.PointNetfeat, _pn.PointNetCls)]
    with _LegacyPath() as lp:
        try:
            for m, cls in natives:
                m.__class__ = lp.classes[cls.__name__]
            with warnings.catch_warnings():
                # the old container tries to store each class's source text and warns when a class has none
                warnings.filterwarnings("ignore", message="Couldn't retrieve source code")
                torch.save(model, path, _use_new_zipfile_serialization=not old_container)
        finally:
            for m, cls in natives:
                m.__class__ = cls


# ---- torch's pre-1.6 container ---------------------------------------------------------------------------------
# Layout (torch/serialization.py, _legacy_save): four pickles in a row - a magic number, a protocol version, a
# dictionary describing the writer's machine, the object - then a pickled list of storage keys and, per key, an
# int64 element count followed by the raw elements.  Inside the object pickle two kinds of persistent id occur:
#   ('storage', storage class, key, location, element count, view)   a tensor's memory, filled in afterwards
#   ('module', class, source file, source text)                      first mention of an nn.Module subclass
_LEGACY_MAGIC = 0x1950A86A20F9469CFC6C
_LEGACY_PROTOCOL = 1001

_STORAGE_DTYPES = {"FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
                   "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16,
                   "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool}


class _StorageType:
    """What `torch.FloatStorage` etc. resolve to while a legacy container is read: just the element type."""

    def __init__(self, name):
        self.name, self.dtype = name, _STORAGE_DTYPES[name]


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    # Tensor.set_ grows a resizable storage to fit the view it is given, so a file could declare a tiny storage and
    # a huge shape and have the memory allocated here, before any later check: the extent of the view is compared
    # with the storage first.
    size, stride = tuple(size), tuple(stride)
    if len(size) != len(stride) or any((not isinstance(v, int)) or v < 0 for v in size + stride) \
            or not isinstance(storage_offset, int) or storage_offset < 0:
        raise pickle.UnpicklingError("tensor record with negative or non-integer size, stride or offset")
    extent = 0 if 0 in size else storage_offset + sum((n - 1) * st for n, st in zip(size, stride)) + 1
    itemsize = torch.empty(0, dtype=storage.dtype).element_size()
    if extent * itemsize > storage.untyped().nbytes():
        raise pickle.UnpicklingError(f"tensor record of {extent} elements does not fit its storage of "
                                     f"{storage.untyped().nbytes()} bytes")
    return torch.empty(0, dtype=storage.dtype).set_(storage.untyped(), storage_offset, size, stride)


def _rebuild_tensor(storage, storage_offset, size, stride):
    return _rebuild_tensor_v2(storage, storage_offset, size, stride)


def _rebuild_parameter(data, requires_grad=True, backward_hooks=None):
    return nn.Parameter(data, requires_grad=bool(requires_grad))


class _LegacyStorage:
    """A typed window on an untyped storage whose bytes arrive after the object pickle has been read."""

    def __init__(self, untyped, dtype):
        self._untyped, self.dtype = untyped, dtype

    def untyped(self):
        return self._untyped


class _RestrictedUnpickler(pickle.Unpickler):
    """pickle.Unpickler that resolves nothing outside a fixed table and understands the two persistent ids above."""

    def __init__(self, f, legacy_classes, encoding="utf-8", max_storage_bytes=None):
        super().__init__(f, encoding=encoding)
        self._budget = max_storage_bytes             # storages of a file cannot be larger than the file
        table = {
            ("collections", "OrderedDict"): collections.OrderedDict,
            ("builtins", "set"): set,                    # nn.Module._non_persistent_buffers_set (torch >= 1.6)
            ("__builtin__", "set"): set,                 # the same under protocol 2's Python 2 module names
            ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
            ("torch._utils", "_rebuild_tensor"): _rebuild_tensor,
            ("torch._utils", "_rebuild_parameter"): _rebuild_parameter,
            ("torch", "device"): torch.device,
            ("torch", "Size"): torch.Size,
            ("torch.nn.modules.conv", "Conv1d"): nn.Conv1d,
            ("torch.nn.modules.batchnorm", "BatchNorm1d"): nn.BatchNorm1d,
            ("torch.nn.modules.linear", "Linear"): nn.Linear,
            ("torch.nn.modules.pooling", "MaxPool1d"): nn.MaxPool1d,
            ("torch.nn.modules.activation", "ReLU"): nn.ReLU,
            ("torch.nn.modules.dropout", "Dropout"): nn.Dropout,
            ("torch.nn.parallel.data_parallel", "DataParallel"): nn.DataParallel,
        }
        for name in _STORAGE_DTYPES:
            table[("torch", name)] = _StorageType(name)
        for name, cls in legacy_classes.items():
            table[(_LEGACY_MODULE, name)] = cls
        self._table = table
        self._module_classes = set(legacy_classes.values()) | {nn.Conv1d, nn.BatchNorm1d, nn.Linear, nn.MaxPool1d,
                                                                 nn.ReLU, nn.Dropout, nn.DataParallel}
        self.storages = {}

    def find_class(self, module, name):
        try:
            return self._table[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}, which a PointNetGPD checkpoint "
                                         "has no reason to contain; refusing to load it") from None

    def persistent_load(self, saved_id):
        if not isinstance(saved_id, tuple) or not saved_id:
            raise pickle.UnpicklingError("malformed persistent id")
        kind = saved_id[0].decode("ascii") if isinstance(saved_id[0], bytes) else saved_id[0]
        if kind == "module":
            cls = saved_id[1]
            if cls not in self._module_classes:
                raise pickle.UnpicklingError("module record names a class outside the allow-list")
            return cls                                   # the recorded source text is not looked at
        if kind == "storage":
            storage_type, key, _location, numel, view = saved_id[1:6]
            if not isinstance(storage_type, _StorageType):
                raise pickle.UnpicklingError("storage record without a known storage type")
            key = key.decode("ascii") if isinstance(key, bytes) else str(key)
            if key not in self.storages:
                nbytes = int(numel) * torch.empty(0, dtype=storage_type.dtype).element_size()
                if nbytes < 0 or (self._budget is not None and nbytes > self._budget):
                    raise pickle.UnpicklingError(f"storage {key} claims {nbytes} bytes, more than the file holds")
                if self._budget is not None:
                    self._budget -= nbytes
                self.storages[key] = _LegacyStorage(torch.UntypedStorage(nbytes), storage_type.dtype)
            root = self.storages[key]
            if view is not None:
                raise pickle.UnpicklingError("storage views are not expected in a PointNetGPD checkpoint")
            return root
        raise pickle.UnpicklingError(f"unknown persistent id kind {kind!r}")


def _is_legacy_container(path) -> bool:
    if zipfile.is_zipfile(path):
        return False
    with open(path, "rb") as f:
        head = f.read(16)
    # protocol-2 pickle of the magic number: PROTO 2, LONG1, 10 bytes little-endian
    return head[:2] == b"\x80\x02" and head[2:3] == b"\x8a" and \
        int.from_bytes(head[4:4 + head[3]], "little") == _LEGACY_MAGIC


def _load_legacy_container(path, legacy_classes, encoding="utf-8"):
    """The object stored in a pre-1.6 torch file, read with the restricted unpickler above."""
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size

        def restricted():
            return _RestrictedUnpickler(f, legacy_classes, encoding, max_storage_bytes=size)
        if restricted().load() != _LEGACY_MAGIC:
            raise ValueError(f"{path}: not a torch checkpoint (bad magic number)")
        if restricted().load() != _LEGACY_PROTOCOL:
            raise ValueError(f"{path}: unsupported legacy protocol version")
        info = restricted().load()
        little = info.get("little_endian", info.get(b"little_endian", True)) if isinstance(info, dict) else True
        if not little:
            raise ValueError(f"{path}: written on a big-endian machine")
        up = restricted()
        obj = up.load()
        keys = restricted().load()
        for key in keys:
            key = key.decode("ascii") if isinstance(key, bytes) else str(key)
            if key not in up.storages:
                raise ValueError(f"{path}: storage {key} is not referenced by the object")
            st = up.storages[key]
            (numel,) = struct.unpack("<q", f.read(8))
            nbytes = numel * torch.empty(0, dtype=st.dtype).element_size()
            if nbytes != st.untyped().nbytes():
                raise ValueError(f"{path}: storage {key} holds {nbytes} bytes, the object expects {st.untyped().nbytes()}")
            data = f.read(nbytes)
            if len(data) != nbytes:
                raise ValueError(f"{path}: truncated (storage {key})")
            if nbytes:
                dst = torch.empty(0, dtype=torch.uint8).set_(st.untyped(), 0, (nbytes,), (1,))
                dst.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        if set(up.storages) - {k.decode("ascii") if isinstance(k, bytes) else str(k) for k in keys}:
            raise ValueError(f"{path}: the object refers to storages the file does not hold")
    return obj


def load_checkpoint(path, num_points=None, device="cuda", encoding="utf-8"):
    """Returns a pointnetgpd_amd.pointnet.PointNetCls in eval mode on `device`.

    num_points is only needed for state_dict files (a whole-module pickle carries it).  Whole modules
    (`torch.save(model, ...)`, possibly wrapped in nn.DataParallel) load from torch's zip container and from the
    pre-1.6 non-zip container that main_1v.py:350 writes; state_dict files from either.  `encoding` is the one
    torch.load takes for files written under Python 2 ("latin1").  Nothing is unpickled without restrictions: a
    file that refers to anything outside the classes a PointNet consists of is refused."""
    with _LegacyPath() as lp:
        if _is_legacy_container(path):
            obj = _load_legacy_container(path, lp.classes, encoding)
        else:
            allowed = list(lp.classes.values()) + [_pn.STN3d, _pn.PointNetfeat, _pn.PointNetCls, nn.Conv1d,
                                                   nn.BatchNorm1d, nn.Linear, nn.MaxPool1d, nn.ReLU, nn.DataParallel,
                                                   nn.Dropout, collections.OrderedDict, torch.device]
            with torch.serialization.safe_globals(allowed):
                obj = torch.load(path, map_location="cpu", weights_only=True, encoding=encoding)
    if isinstance(obj, nn.DataParallel):
        obj = obj.__dict__["_modules"]["module"]
    if isinstance(obj, dict):
        if "state_dict" in obj and isinstance(obj["state_dict"], dict):
            obj = obj["state_dict"]
        model = _from_state_dict(obj, num_points)
    elif isinstance(obj, nn.Module):
        model = _to_native(obj)
        if num_points is not None and int(model.num_points) != int(num_points):
            raise ValueError(f"checkpoint was built with num_points={model.num_points}, asked for {num_points}")
    else:
        raise ValueError(f"unsupported checkpoint content: {type(obj).__name__}")
    return model.to(device).eval()
