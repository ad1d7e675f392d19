"""Training step of the reference (main_1v.py:237-302) on the HIP path.

Two ways in, same kernels:

* drop-in: a PointNetCls in train() mode called as ``output, _ = model(data)`` goes through
  `TrainFunction`, a torch.autograd.Function whose forward/backward call pngpd_train_forward /
  pngpd_train_backward; ``F.nll_loss(output, target).backward()`` and ``torch.optim.Adam`` work
  unchanged (main_1v.py:266-274).
* fused: `Trainer.step(data, target)` keeps all parameters, gradients and Adam moments in flat
  buffers, runs forward, NLL, backward and a fused Adam update.  When several ranks train
  data-parallel the flat gradient buffer is reduced over RCCL in three buckets - head, trunk,
  transform regressor, the order in which the backward pass finishes them - each started as soon as
  its part of the backward has been queued, so the exchange of one bucket runs under the kernels of the
  next part.  Every rank starts from rank 0's weights and BatchNorm buffers (broadcast when the Trainer
  is built); BatchNorm statistics then stay per-replica, like nn.DataParallel in the reference
  (main_1v.py:224).
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib
from .pointnet import invalidate_packed_weights

BN_MOMENTUM = 0.1


def _unit_tensors(model):
    """(params table order) unit -> [weight, bias, bn_weight, bn_bias, running_mean, running_var]."""
    out = []
    units = model._units()
    for uname in _lib.UNITS:
        if uname not in units:                 # a model without this part (standalone STN3d / PointNetfeat)
            out.append([None] * 6)
            continue
        layer, bn = units[uname]
        row = [layer.weight, layer.bias]
        row += [bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [None] * 4
        out.append(row)
    return out


def _table(rows, getter):
    t = (ctypes.c_void_p * _lib.SLOTS)()
    for ui, row in enumerate(rows):
        for fi, x in enumerate(row):
            v = getter(x, fi) if x is not None else None
            if v is not None:
                if v.dtype != torch.float32 or not v.is_contiguous() or not v.is_cuda:
                    raise ValueError("training needs contiguous float32 ROCm tensors")
                t[ui * len(_lib.FIELDS) + fi] = v.data_ptr()
    return t


def _bump_num_batches(model):
    bufs = [m.num_batches_tracked for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    torch._foreach_add_(bufs, 1)


def _check(model, x):
    if not x.is_cuda:
        raise RuntimeError("pointnetgpd_amd trains on MI355X only (there is no CPU path)")
    if x.dim() != 3 or x.shape[1] != 3 or x.shape[2] != model.num_points:
        raise ValueError(f"expected data [B,3,{model.num_points}], got {tuple(x.shape)}")
    if x.shape[0] < 2:
        raise ValueError("train-mode BatchNorm needs more than one sample per batch (main_1v.py:258)")
    return x.detach().to(torch.float32).contiguous()


def _classes(model):
    return model.fc3.out_features if (model._PARTS & _lib.PART_HEAD) else 2     # unused without a head


def train_forward(model, x, ws=None, conv3_ms=None):
    """Train-mode forward of a PointNetCls, PointNetfeat or STN3d; returns (logp, trans, workspace, x) - for a model
    without a head (logp is None then) followed by feat.  Updates running statistics.
    conv3_ms: a list that receives the durations (ms, HIP events) of the two conv3 product/max kernels."""
    lib = _lib.load()
    x = _check(model, x)
    B, _, P = x.shape
    parts = model._PARTS
    k = _classes(model)
    dev = x.device
    nbytes = lib.pngpd_train_workspace_bytes(B, P, k)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rows = _unit_tensors(model)
    params = _table(rows, lambda t, fi: t.data)
    logp = torch.empty((B, k), dtype=torch.float32, device=dev) if parts & _lib.PART_HEAD else None
    trans = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    feat = torch.empty((B, 1024), dtype=torch.float32, device=dev) if parts == (_lib.PART_STN | _lib.PART_TRUNK) else None

    def ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        args = (B, P, k, ptr(x), params, _lib.SLOTS, BN_MOMENTUM, ptr(logp), ptr(trans), ptr(feat), ptr(ws), ws.numel())
        if conv3_ms is None:
            _lib.check(lib.pngpd_train_forward_parts(parts, *args, stream))
        else:
            pair = (ctypes.c_float * 2)()
            _lib.check(lib.pngpd_train_forward_timed(*args, pair, stream))
            conv3_ms += [pair[0], pair[1]]
    _bump_num_batches(model)
    invalidate_packed_weights()        # running statistics changed under torch's version counters
    if parts & _lib.PART_HEAD:
        return logp, trans, ws, x
    return logp, trans, ws, x, feat


def critical_points(ws, B, P, k=2):
    """(stn_arg, trunk_arg), int32 [B,1024] each: the point that holds every channel's maximum in the two
    max-pools (model/pointnet.py:33, :147) of the last train-mode forward that used workspace `ws`."""
    lib = _lib.load()
    stn = torch.empty((B, 1024), dtype=torch.int32, device=ws.device)
    trunk = torch.empty_like(stn)
    with torch.cuda.device(ws.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(ws.device).cuda_stream)
        _lib.check(lib.pngpd_train_argmax(B, P, k, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                          ctypes.c_void_p(stn.data_ptr()), ctypes.c_void_p(trunk.data_ptr()), stream))
    return stn, trunk


SAVED = ("stn_h1", "stn_h2", "stn_pooled", "stn_a1", "stn_a2", "trunk_h1", "trunk_h2", "trunk_pooled", "head_a1",
         "head_a2")


def saved_activation(ws, B, P, k, name):
    """View (no copy) of an activation the last train-mode forward saved in workspace `ws` for the backward pass:
    one of SAVED; per-point layers come as [B*P, C] rows ordered (candidate, point)."""
    lib = _lib.load()
    ptr, count = ctypes.c_void_p(), ctypes.c_size_t()
    _lib.check(lib.pngpd_train_saved(B, P, k, ctypes.c_void_p(ws.data_ptr()), ws.numel(), SAVED.index(name),
                                     ctypes.byref(ptr), ctypes.byref(count)))
    off = ptr.value - ws.data_ptr()
    flat = ws[off:off + 4 * count.value].view(torch.float32)
    rows = B if ("pooled" in name or "_a" in name) else B * P
    return flat.view(rows, count.value // rows)


PARTS_ALL = _lib.PART_STN | _lib.PART_TRUNK | _lib.PART_HEAD


def train_backward(model, x, logp, dlogp, dtrans, ws, grad_of, parts=None, dfeat=None):
    """Backward of train_forward.  grad_of(param) -> tensor that receives that parameter's gradient.
    `parts` (mask of _lib.PART_*, default: all the model owns) runs a segment only; segments must follow each other
    as HEAD, TRUNK, STN.  dfeat: gradient w.r.t. the global feature of a model without a head."""
    lib = _lib.load()
    B, _, P = x.shape
    k = _classes(model)
    if parts is None:
        parts = model._PARTS
    rows = _unit_tensors(model)
    params = _table(rows, lambda t, fi: t.data)
    grads = _table(rows, lambda t, fi: grad_of(t) if fi < 4 else None)

    def ptr(t):
        return ctypes.c_void_p(t.to(torch.float32).contiguous().data_ptr()) if t is not None else None

    keep = [t.to(torch.float32).contiguous() if t is not None else None for t in (dlogp, dtrans, dfeat)]
    with torch.cuda.device(x.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.pngpd_train_backward_part(B, P, k, ctypes.c_void_p(x.data_ptr()), params, grads, _lib.SLOTS,
                                                 ptr(logp), ptr(keep[0]), ptr(keep[1]), ptr(keep[2]),
                                                 ctypes.c_void_p(ws.data_ptr()), ws.numel(), parts, stream))


class TrainFunction(torch.autograd.Function):
    """autograd bridge: forward(model, x, *parameters) -> (logp, trans) for a PointNetCls, (feat, trans) for a
    PointNetfeat, trans for an STN3d (model/pointnet.py:185-190, :135-149, :28-45 in train mode)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        out = train_forward(model, x)
        logp, trans, ws, xc = out[:4]
        ctx.model, ctx.ws, ctx.x, ctx.logp, ctx.params = model, ws, xc, logp, params
        if model._PARTS & _lib.PART_HEAD:
            return logp, trans
        if model._PARTS & _lib.PART_TRUNK:
            return out[4], trans
        return trans

    @staticmethod
    def backward(ctx, *douts):
        model = ctx.model
        grads = {id(p): torch.empty_like(p) for p in ctx.params}
        grad_of = lambda p: grads[id(p)]        # noqa: E731
        if model._PARTS & _lib.PART_HEAD:
            dlogp, dtrans = douts
            if dlogp is None:
                dlogp = torch.zeros_like(ctx.logp)
            train_backward(model, ctx.x, ctx.logp, dlogp, dtrans, ctx.ws, grad_of)
        elif model._PARTS & _lib.PART_TRUNK:
            dfeat, dtrans = douts
            if dfeat is None:
                dfeat = torch.zeros((ctx.x.shape[0], 1024), dtype=torch.float32, device=ctx.x.device)
            train_backward(model, ctx.x, None, None, dtrans, ctx.ws, grad_of, dfeat=dfeat)
        else:
            (dtrans,) = douts
            if dtrans is None:
                dtrans = torch.zeros((ctx.x.shape[0], 3, 3), dtype=torch.float32, device=ctx.x.device)
            train_backward(model, ctx.x, None, None, dtrans, ctx.ws, grad_of)
        ctx.ws = None
        return (None, None) + tuple(grads[id(p)] for p in ctx.params)


def forward_train_mode(model, x):
    """What PointNetCls / PointNetfeat / STN3d .forward do in train() mode."""
    params = [t for row in _unit_tensors(model) for t in row[:4] if t is not None]
    return TrainFunction.apply(model, x, *params)


def nll_loss(logp, target):
    """(loss[1], dlogp[B,k]) of F.nll_loss(logp, target) with mean reduction, on the device."""
    lib = _lib.load()
    B, k = logp.shape
    target = target.to(torch.int64).reshape(-1).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logp.device)
    dlogp = torch.empty_like(logp)
    with torch.cuda.device(logp.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(logp.device).cuda_stream)
        _lib.check(lib.pngpd_nll_loss(ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(target.data_ptr()), B, k,
                                      ctypes.c_void_p(loss.data_ptr()), ctypes.c_void_p(dlogp.data_ptr()), stream))
    return loss, dlogp


def step_lr(base_lr: float, epoch: int, step_size: int = 30, gamma: float = 0.5) -> float:
    """Learning rate the reference trains epoch `epoch` (0-based) with: StepLR(30, 0.5) whose
    step() is called at the START of every epoch (main_1v.py:233, 239)."""
    return base_lr * gamma ** ((epoch + 1) // step_size)


class Trainer:
    """Flat-buffer trainer: Adam(lr=0.005, betas=(0.9, 0.999), eps=1e-8) as in main_1v.py:226."""

    def __init__(self, model, lr=0.005, betas=(0.9, 0.999), eps=1e-8, process_group=None):
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.group = process_group
        self.step_count = 0
        named = list(model.named_parameters())
        self.params = [p for _, p in named]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("move the model to a ROCm device before building a Trainer")
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self._gview = {}
        owner = []                 # which part of the network every parameter belongs to, in flat order
        for name, p in named:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)         # parameters become views of the bucket
            g = self.flat_g[off:off + k].view_as(p)
            p.grad = g
            self._gview[id(p)] = g
            part = _lib.PART_STN if name.startswith("feat.stn.") else (
                _lib.PART_TRUNK if name.startswith("feat.") else _lib.PART_HEAD)
            owner.append((part, off, off + k))
            off += k
        # One contiguous range of the flat gradient per part (registration order is STN, trunk, head), listed in the
        # order the backward pass finishes them.
        self.buckets = []
        for part in (_lib.PART_HEAD, _lib.PART_TRUNK, _lib.PART_STN):
            spans = [(lo, hi) for q, lo, hi in owner if q == part]
            lo, hi = spans[0][0], spans[-1][1]
            if sum(b - a for a, b in spans) != hi - lo:
                raise RuntimeError("parameters of one network part are not contiguous in registration order")
            self.buckets.append((part, lo, hi))
        self._ws = None
        self.conv3_ms = None      # set to a list to collect the conv3 kernel durations of every step
        self.wait_events = None   # set to a list to collect (before, after) events around the waits for the collectives
        self.always_bucketed = False   # part-by-part backward and its collectives even on one rank (for tests)
        if self.world_size() > 1:
            self.broadcast_state()

    def world_size(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _device_collectives(self):
        return dist.get_backend(self.group) == "nccl"

    def broadcast_state(self, src=0):
        """Every rank takes rank `src`'s parameters and BatchNorm buffers (running statistics and batch counters):
        replicas built from different random initialisations would otherwise apply averaged gradients to different
        weights and never agree."""
        tensors = [self.flat_p]
        tensors += [b for m in self.model.modules() if isinstance(m, torch.nn.BatchNorm1d)
                    for b in (m.running_mean, m.running_var, m.num_batches_tracked)]
        for t in tensors:
            if self._device_collectives():
                dist.broadcast(t, src=src, group=self.group)
            else:                      # a host-memory backend (gloo): stage through the host
                host = t.detach().cpu()
                dist.broadcast(host, src=src, group=self.group)
                t.copy_(host)
        invalidate_packed_weights()

    def _reduce_bucket(self, lo, hi):
        """Sum flat_g[lo:hi] over the ranks (Adam divides by the world size).  Returns a work handle to wait on, or
        None when the exchange has already completed."""
        g = self.flat_g[lo:hi]
        if self._device_collectives():
            # RCCL runs the collective on its own stream, ordered behind what is queued on the current stream so
            # far; kernels queued afterwards overlap with it
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        host = g.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        g.copy_(host)
        return None

    def gradients(self, data, target):
        """Forward, loss and backward of one batch; leaves in flat_g the gradient SUMMED over the ranks.
        Returns (loss[1] tensor of this rank's batch, logp[B,k])."""
        model = self.model
        if not model.training:
            raise RuntimeError("Trainer.step needs model.train()")
        logp, _, self._ws, x = train_forward(model, data, self._ws, self.conv3_ms)
        loss, dlogp = nll_loss(logp, target)
        grad_of = lambda p: self._gview[id(p)]      # noqa: E731
        if self.world_size() == 1 and not self.always_bucketed:
            train_backward(model, x, logp, dlogp, None, self._ws, grad_of)
            return loss, logp
        pending = []
        for part, lo, hi in self.buckets:           # head, trunk, transform regressor
            train_backward(model, x, logp, dlogp, None, self._ws, grad_of, parts=part)
            pending.append(self._reduce_bucket(lo, hi))
        timed = self.wait_events is not None and self._device_collectives()
        if timed:
            before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            before.record()
        for work in pending:
            if work is not None:
                work.wait()                         # the current stream waits for the collective's stream
        if timed:
            after.record()
            self.wait_events.append((before, after))
        return loss, logp

    def apply(self, world=None):
        """The Adam update from flat_g, which holds the sum of `world` ranks' gradients."""
        world = self.world_size() if world is None else world
        self.step_count += 1
        lib = _lib.load()
        dev = self.flat_p.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.pngpd_adam_step(ctypes.c_void_p(self.flat_p.data_ptr()), ctypes.c_void_p(self.flat_g.data_ptr()),
                                           ctypes.c_void_p(self.exp_avg.data_ptr()),
                                           ctypes.c_void_p(self.exp_avg_sq.data_ptr()), self.flat_p.numel(), self.lr,
                                           self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / world, stream))
        invalidate_packed_weights()    # the fused update bypasses torch's version counters

    def step(self, data, target):
        """One training step; returns (loss[1] tensor, logp[B,k])."""
        loss, logp = self.gradients(data, target)
        self.apply()
        return loss, logp
