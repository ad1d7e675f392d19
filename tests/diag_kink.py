"""Hand-run: which elements of feat.bn2.bias / feat.bn1.bias differ most from the float32 oracle (P B k seed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
P, B, k, seed = (int(v) for v in sys.argv[1:5]); gain = 1.3
x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
sd = synth.torch_state_dict(seed, k, 1.0, gain)
o = po.build_cls(P, k, sd, eval_mode=False).double()
out, _ = o(x.double()); F.nll_loss(out, target).backward()
m = PointNetCls(P, 3, k); m.load_state_dict(sd); m = m.cuda().train()
out2, _ = m(x.cuda()); F.nll_loss(out2, target.cuda()).backward()
for n in ("feat.bn2.bias", "feat.bn1.bias", "feat.stn.bn2.bias", "feat.stn.bn1.bias"):
    r = o.get_parameter(n).grad.numpy(); g = m.get_parameter(n).grad.cpu().numpy().astype(np.float64)
    e = np.abs(g - r) / np.abs(r).max()
    top = np.argsort(-e)[:3]
    print(n, [(int(i), f"{e[i]:.1e}") for i in top])
