"""Hand-run diagnostic: per-parameter gradient error and Adam sign flips of one training step against the
CPU oracle (python tests/diag_train_step.py [P k seed B gain])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls

a = sys.argv[1:]
P, k, seed, B = (int(v) for v in a[:4]) if len(a) >= 4 else (750, 2, 21, 4)
gain = float(a[4]) if len(a) > 4 else 1.7
x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
sd = synth.torch_state_dict(seed, k, 1.0, gain)
o = po.build_cls(P, k, sd, eval_mode=False)
out, _ = o(x)
l = F.nll_loss(out, target)
l.backward()
ref = {n: p.grad.numpy() for n, p in o.named_parameters()}
m = PointNetCls(P, 3, k)
m.load_state_dict(sd)
m = m.cuda().train()
out2, _ = m(x.cuda())
l2 = F.nll_loss(out2, target.cuda())
l2.backward()
print("loss", l.item(), l2.item(), "logp err", (out2.detach().cpu() - out.detach()).abs().max().item())
for n, p in m.named_parameters():
    g = p.grad.cpu().numpy()
    r = ref[n]
    e = np.abs(g - r).max() / max(np.abs(r).max(), 1e-12)
    flips = float(np.mean(np.sign(g) != np.sign(r)))
    print(f"{n:28s} relerr {e:9.2e}  |ref|max {np.abs(r).max():9.2e} |got|max {np.abs(g).max():9.2e} sign flips {flips:7.4f}")

# forward intermediates: pooled trunk feature of the HIP path against the oracle, and how close the oracle's
# head pre-activations are to a ReLU flip
import ctypes
from pointnetgpd_amd import _lib, training
o2 = po.build_cls(P, k, sd, eval_mode=False)
with torch.no_grad():
    feat_ref, trans_ref = o2.feat(x)
    z1 = o2.bn1(o2.fc1(feat_ref))
    print("oracle head bn1 pre-activation closest to zero:", z1.abs().min().item())
m2 = PointNetCls(P, 3, k)
m2.load_state_dict(sd)
m2 = m2.cuda().train()
lib = _lib.load()
xc = x.cuda().contiguous()
nbytes = lib.pngpd_train_workspace_bytes(B, P, k)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
rows = training._unit_tensors(m2)
params = training._table(rows, lambda t, fi: t.data)
logp = torch.empty((B, k), device="cuda")
trans = torch.empty((B, 3, 3), device="cuda")
feat = torch.empty((B, 1024), device="cuda")
_lib.check(lib.pngpd_train_forward(B, P, k, ctypes.c_void_p(xc.data_ptr()), params, _lib.SLOTS, 0.1,
                                   ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(trans.data_ptr()),
                                   ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
d = (feat.cpu() - feat_ref).abs()
print("trunk pooled: max abs err", d.max().item(), "ref max", feat_ref.abs().max().item(), "trans err",
      (trans.cpu() - trans_ref).abs().max().item())
o3 = po.build_cls(P, k, sd, eval_mode=False)
with torch.no_grad():
    o3(x)          # one train-mode forward of the oracle updates its running statistics
osd, msd = o3.state_dict(), m2.state_dict()
for n in osd:
    if n.endswith(("running_var", "running_mean")):
        r = osd[n].numpy()
        g = msd[n].cpu().numpy()
        r0 = sd[n].numpy()      # the update is 0.1 * batch statistic: error relative to that part
        print(f"{n:32s} max rel err of the batch statistic {np.max(np.abs(g - r) / (np.abs(r - 0.9 * r0) + 1e-9)):9.2e}")

# the same step in float64: how far is each float32 implementation from it?
o64 = po.build_cls(P, k, sd, eval_mode=False).double()
out64, _ = o64(x.double())
F.nll_loss(out64, target).backward()
ref64 = {n: p.grad.numpy() for n, p in o64.named_parameters()}
print("against the float64 step: torch float32 | HIP")
for n, p in m.named_parameters():
    r = ref64[n]
    den = max(np.abs(r).max(), 1e-12)
    print(f"{n:28s} {np.abs(ref[n] - r).max() / den:9.2e} | {np.abs(p.grad.cpu().numpy() - r).max() / den:9.2e}")
with torch.no_grad():
    o4 = po.build_cls(P, k, sd, eval_mode=False).double()
    xd = x.double()
    trans = o4.feat.stn(xd)
    xt = torch.bmm(xd.transpose(2, 1), trans).transpose(2, 1)
    h = torch.relu(o4.feat.bn1(o4.feat.conv1(xt)))
    h = torch.relu(o4.feat.bn2(o4.feat.conv2(h)))
    y = o4.feat.bn3(o4.feat.conv3(h))
    top2 = y.topk(2, dim=2).values
    gap = (top2[..., 0] - top2[..., 1]) / y.abs().amax(dim=2).clamp_min(1e-9)
    print("trunk max-pool: (b,c) pairs whose two largest values are within 1e-6 / 1e-5 / 1e-4 relative:",
          int((gap < 1e-6).sum()), int((gap < 1e-5).sum()), int((gap < 1e-4).sum()), "of", gap.numel())
