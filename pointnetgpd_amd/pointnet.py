"""Drop-in replacements for STN3d / PointNetfeat / PointNetCls (reference model/pointnet.py:8-45,
122-152, 174-190) whose forward runs hand-written gfx950 kernels through the C ABI in
include/pngpd.h.

What is preserved from the reference: constructor arguments (num_points, input_chann, k /
global_feat), sub-module names and registration order - hence the 74 state_dict keys, shapes and
their order - forward signatures and return values, .eval()/.train()/.to()/.cuda()/parameters().

What is different: there is no CPU execution path.  The parameters live in ordinary
Conv1d/Linear/BatchNorm1d containers, but forward() never calls them; it hands device pointers to
the library.  A CPU tensor, or a missing libpngpd.so, raises.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from . import _lib

# Arithmetic of the two conv stacks in eval mode (include/pngpd.h): "fp32" = exact-f32 MFMA, "bf16x3" = split-bf16
# operands at float32 grade (1e-4 on the log-probabilities whatever their magnitude, about four times the fp32
# mode's throughput), "bf16" = plain bf16 operands (about three times faster again; its error is 0.4 % of the class
# scores, i.e. 2e-2 only while they differ by a few units).  The default is the fastest mode that holds 1e-4.
_DEFAULT_PRECISION = os.environ.get("PNGPD_PRECISION", "bf16x3")

# Bumped whenever library kernels change parameters or running statistics behind torch's back
# (training step, fused Adam): torch's per-tensor version counters do not see those writes.
_GENERATION = [0]
# Bumped whenever the tensors of a module may have been replaced (see _HipBacked._weights_signature).
_STRUCTURE = [0]


def invalidate_packed_weights() -> None:
    """Force every module to re-fold and re-pack its weights on the next eval-mode forward."""
    _GENERATION[0] += 1
    _STRUCTURE[0] += 1


class _Handle:
    """Owns one pngpd_model (device weights) and the per-batch-size workspaces."""

    def __init__(self, device: torch.device, num_points: int, k: int, parts: int):
        self.lib = _lib.load()
        self.device = device
        self.ptr = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.pngpd_model_create(ctypes.byref(self.ptr), num_points, k, parts))
        self.signature = None
        self._ws = {}

    def workspace(self, B: int) -> torch.Tensor:
        ws = self._ws.get(B)
        if ws is None:
            nbytes = self.lib.pngpd_forward_workspace_bytes(self.ptr, B)
            if len(self._ws) > 8:
                self._ws.clear()
            ws = self._ws[B] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def __del__(self):
        try:
            if self.ptr:
                self.lib.pngpd_model_destroy(self.ptr)
                self.ptr = ctypes.c_void_p()
        except Exception:
            pass


class _HipBacked(nn.Module):
    """Shared plumbing: weight-table assembly, repack-on-change, launch."""

    _PARTS = 0
    precision = _DEFAULT_PRECISION  # "fp32", "bf16x3" or "bf16"; an instance attribute overrides

    # unit name (C ABI order) -> (layer, batchnorm or None); filled by subclasses
    def _units(self) -> dict:
        raise NotImplementedError

    def _num_classes(self) -> int:
        return 2   # unused by models without a head

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_hip", None)   # device handles never travel through pickle / deepcopy,
        state.pop("_sig_cache", None)   # nor does the process-local list of tensors and device addresses
        return state

    def _tensors(self):
        for layer, bn in self._units().values():
            yield layer.weight
            yield layer.bias
            if bn is not None:
                yield from (bn.weight, bn.bias, bn.running_mean, bn.running_var)

    # ---- when do the packed device weights have to be rebuilt? -------------------------------------------------
    # Values change in place (optimizer steps, load_state_dict, manual edits under no_grad): every such write bumps
    # the tensor's autograd version counter.  A tensor's STORAGE changes without a version bump through
    # `param.data = t` (torch.nn.utils.vector_to_parameters, EMA / SWA weight swaps, flattening parameters into one
    # buffer).  Both are polled on every forward - one attribute read and one data_ptr() each, from a cached list of
    # the module's tensors.  The tensor OBJECTS change through .to() / .cuda() / .float() (nn.Module._apply),
    # load_state_dict(assign=True) or by assigning a new sub-module or Parameter to a module of this package; those
    # paths bump _STRUCTURE below, which drops the cached list.  What nothing here can see is a Parameter assigned
    # straight onto a plain torch layer (`model.fc3.weight = nn.Parameter(...)`): call invalidate_packed_weights()
    # after that.
    def _apply(self, fn, *args, **kwargs):
        _STRUCTURE[0] += 1
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        _STRUCTURE[0] += 1
        return super().load_state_dict(*args, **kwargs)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, torch.Tensor)):
            _STRUCTURE[0] += 1
        super().__setattr__(name, value)

    def _weights_signature(self):
        cache = self.__dict__.get("_sig_cache")
        if cache is None or cache[0] != _STRUCTURE[0]:
            cache = self.__dict__["_sig_cache"] = (_STRUCTURE[0], list(self._tensors()))
        versions = 0
        ptrs = 0
        for i, t in enumerate(cache[1]):
            versions += t._version
            ptrs ^= t.data_ptr() * (2 * i + 1)      # order-sensitive digest of the storages' addresses
        return (_GENERATION[0], versions, ptrs)

    def _handle(self, device: torch.device) -> _Handle:
        h = self.__dict__.get("_hip")
        if h is None or h.device != device:
            h = self.__dict__["_hip"] = _Handle(device, self.num_points, self._num_classes(), self._PARTS)
        sig = self._weights_signature()
        if h.signature != sig:
            self._upload(h)
            h.signature = sig
        return h

    def _upload(self, h: _Handle) -> None:
        units = self._units()
        table = (ctypes.c_void_p * _lib.SLOTS)()
        slots, parts = [], []
        for ui, uname in enumerate(_lib.UNITS):
            if uname not in units:
                continue
            layer, bn = units[uname]
            fields = [layer.weight, layer.bias]
            fields += [bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [None] * 4
            for fi, t in enumerate(fields):
                if t is not None:
                    slots.append(ui * len(_lib.FIELDS) + fi)
                    parts.append(t.detach().to(torch.float32).reshape(-1))
        with torch.cuda.device(h.device):
            stream = torch.cuda.current_stream(h.device).cuda_stream
            if all(t.device == h.device for t in parts) and os.environ.get("PNGPD_HOST_FOLD", "0") != "1":
                # the parameters already live on the GPU (a model that was just trained there, or moved with .cuda()):
                # BatchNorm is folded and the images are built by kernels on the stream, nothing crosses to the host and
                # nothing waits.  `parts` holds the (possibly converted) tensors until the call has queued its kernels;
                # their memory is recycled in stream order.
                parts = [t.contiguous() for t in parts]
                for slot, part in zip(slots, parts):
                    table[slot] = part.data_ptr()
                _lib.check(h.lib.pngpd_model_load_device(h.ptr, table, _lib.SLOTS, ctypes.c_void_p(stream)))
                return
            # one device-to-host transfer for all tensors (68 separate ones cost a synchronisation each)
            host = torch.cat(parts).cpu()
            off = 0
            for slot, part in zip(slots, parts):
                table[slot] = host.data_ptr() + 4 * off
                off += part.numel()
            _lib.check(h.lib.pngpd_model_load(h.ptr, table, _lib.SLOTS, ctypes.c_void_p(stream)))

    def _check_input(self, x: torch.Tensor, layout: int):
        if not isinstance(x, torch.Tensor) or x.dim() != 3:
            raise ValueError("expected a 3-D tensor of candidate points")
        if not x.is_cuda:
            raise RuntimeError("pointnetgpd_amd runs on MI355X only: move the module and its input to a "
                               "ROCm device (there is no CPU path)")
        chan_dim, pts_dim = (1, 2) if layout == _lib.LAYOUT_C3P else (2, 1)
        if x.shape[chan_dim] != 3:
            raise ValueError(f"input_chann must be 3, got {x.shape[chan_dim]}")
        if x.shape[pts_dim] != self.num_points:
            raise ValueError(f"input has {x.shape[pts_dim]} points per candidate but the module was built "
                             f"with num_points={self.num_points}")
        return x.detach().to(torch.float32).contiguous()

    def _infer(self, x: torch.Tensor, layout: int, want_logp: bool, want_feat: bool):
        x = self._check_input(x, layout)
        B = x.shape[0]
        dev = x.device
        h = self._handle(dev)
        prec = _lib.PRECISIONS[str(self.precision).lower()]
        k = self._num_classes()
        logp = torch.empty((B, k), dtype=torch.float32, device=dev) if want_logp else None
        feat = torch.empty((B, 1024), dtype=torch.float32, device=dev) if want_feat else None
        trans = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
        ws = h.workspace(B)
        if torch.cuda.current_device() != dev.index:      # the library works on the current device
            with torch.cuda.device(dev):
                return self._launch(h, x, layout, B, prec, logp, trans, feat, ws)
        return self._launch(h, x, layout, B, prec, logp, trans, feat, ws)

    def _launch(self, h, x, layout, B, prec, logp, trans, feat, ws):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = h.lib.pngpd_forward_infer(h.ptr, x.data_ptr(), layout, B, self.num_points, prec,
                                       logp.data_ptr() if logp is not None else None, trans.data_ptr(),
                                       feat.data_ptr() if feat is not None else None, ws.data_ptr(), ws.numel(), stream)
        if rc != _lib.OK:
            _lib.check(rc)
        return logp, trans, feat


def _conv(cin, cout):
    return nn.Conv1d(cin, cout, 1)


class STN3d(_HipBacked):
    """3x3 input transform regressor; forward(x[B,3,P]) -> [B,3,3]."""

    _PARTS = _lib.PART_STN

    def __init__(self, num_points=2500, input_chann=3):
        super().__init__()
        self.num_points = num_points
        self.conv1, self.conv2, self.conv3 = _conv(input_chann, 64), _conv(64, 128), _conv(128, 1024)
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, 9)
        self.relu = nn.ReLU()
        for i, c in enumerate((64, 128, 1024, 512, 256), start=1):
            setattr(self, f"bn{i}", nn.BatchNorm1d(c))

    def _stn_units(self):
        return {"stn_conv1": (self.conv1, self.bn1), "stn_conv2": (self.conv2, self.bn2),
                "stn_conv3": (self.conv3, self.bn3), "stn_fc1": (self.fc1, self.bn4),
                "stn_fc2": (self.fc2, self.bn5), "stn_fc3": (self.fc3, None)}

    _units = _stn_units

    def forward(self, x):
        if self.training:                       # batch-statistics BatchNorm + autograd
            from .training import forward_train_mode
            return forward_train_mode(self, x)
        return self._infer(x, _lib.LAYOUT_C3P, False, False)[1]


class PointNetfeat(_HipBacked):
    """Global feature extractor; forward(x[B,3,P]) -> (feat[B,1024], trans[B,3,3])."""

    _PARTS = _lib.PART_STN | _lib.PART_TRUNK

    def __init__(self, num_points=2500, input_chann=3, global_feat=True):
        super().__init__()
        self.stn = STN3d(num_points=num_points, input_chann=input_chann)
        self.conv1, self.conv2, self.conv3 = _conv(input_chann, 64), _conv(64, 128), _conv(128, 1024)
        for i, c in enumerate((64, 128, 1024), start=1):
            setattr(self, f"bn{i}", nn.BatchNorm1d(c))
        self.mp1 = nn.MaxPool1d(num_points)
        self.num_points = num_points
        self.global_feat = global_feat

    def _feat_units(self):
        u = self.stn._stn_units()
        u.update({"feat_conv1": (self.conv1, self.bn1), "feat_conv2": (self.conv2, self.bn2),
                  "feat_conv3": (self.conv3, self.bn3)})
        return u

    _units = _feat_units

    def forward(self, x):
        if not self.global_feat:
            raise NotImplementedError("global_feat=False (per-point features for PointNetDenseCls) is not "
                                      "part of the grasp-scoring path")
        if self.training:                       # batch-statistics BatchNorm + autograd
            from .training import forward_train_mode
            return forward_train_mode(self, x)
        _, trans, feat = self._infer(x, _lib.LAYOUT_C3P, False, True)
        return feat, trans


class PointNetCls(_HipBacked):
    """Grasp classifier; forward(x[B,3,P]) -> (log_probs[B,k], trans[B,3,3])."""

    _PARTS = _lib.PART_STN | _lib.PART_TRUNK | _lib.PART_HEAD

    def __init__(self, num_points=2500, input_chann=3, k=2):
        super().__init__()
        self.num_points = num_points
        self.feat = PointNetfeat(num_points, input_chann=input_chann, global_feat=True)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k)
        self.bn1, self.bn2 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)
        self.relu = nn.ReLU()

    def _units(self):
        u = self.feat._feat_units()
        u.update({"fc1": (self.fc1, self.bn1), "fc2": (self.fc2, self.bn2), "fc3": (self.fc3, None)})
        return u

    def _num_classes(self) -> int:
        return self.fc3.out_features

    def forward(self, x):
        if self.training:
            # batch-statistics BatchNorm + autograd (main_1v.py:266-272)
            from .training import forward_train_mode
            return forward_train_mode(self, x)
        logp, trans, _ = self._infer(x, _lib.LAYOUT_C3P, True, False)
        return logp, trans

    def forward_points(self, points, return_feat=False):
        """Same network on points laid out [B,P,3] (what the in-gripper crop produces)."""
        logp, trans, feat = self._infer(points, _lib.LAYOUT_P3, True, return_feat)
        return (logp, trans, feat) if return_feat else (logp, trans)
