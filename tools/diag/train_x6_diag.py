"""Hand-run: gradients of one train step (P=500, B=6, k=3, seed 77 and P=257, B=9, k=2, seed 7) from the HIP path
against a float64 oracle, with the float32 oracle beside it; pooled activations against float64."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pointnet_oracle as po  # noqa: E402
from pointnetgpd_amd import synth, training  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402
from pointnetgpd_amd.training import nll_loss  # noqa: E402

for P, B, k, seed, gain in ((500, 6, 3, 77, 1.7), (257, 9, 2, 7, 1.3)):
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    res = {}
    pooled = {}
    for dt in (torch.float32, torch.float64):
        o = po.build_cls(P, k, sd, eval_mode=False).to(dt)
        feats = {}
        o.feat.stn.mp1.register_forward_hook(lambda m, i, out: feats.__setitem__("stn", out.detach().clone()))
        o.feat.mp1.register_forward_hook(lambda m, i, out: feats.__setitem__("trunk", out.detach().clone()))
        out, _ = o(x.to(dt))
        F.nll_loss(out, target).backward()
        res[dt] = {n: p.grad.double() for n, p in o.named_parameters()}
        pooled[dt] = {a: v.reshape(B, 1024).double() for a, v in feats.items()}
    m = PointNetCls(P, 3, k)
    m.load_state_dict(sd)
    m = m.cuda().train()
    logp, trans, ws, xc = training.train_forward(m, x.cuda())
    loss, dlogp = nll_loss(logp, target.cuda())
    grads = {id(p): torch.zeros_like(p) for p in m.parameters()}
    training.train_backward(m, xc, logp, dlogp, None, ws, lambda p: grads[id(p)])
    hp = {"stn": training.saved_activation(ws, B, P, k, "stn_pooled").cpu().double(),
          "trunk": training.saved_activation(ws, B, P, k, "trunk_pooled").cpu().double()}
    print(f"--- P={P} B={B} seed={seed} mode={os.environ.get('PNGPD_TRAIN_CONV3', 'x6')}")
    for a in ("stn", "trunk"):
        ref = pooled[torch.float64][a]
        if a == "stn":
            ref = ref.clamp_min(0)          # the HIP path stores the pooled value after the ReLU
        sc = float(ref.abs().max())
        print(f"pooled {a}: hip {float((hp[a] - ref).abs().max()) / sc:.2e}  f32 oracle "
              f"{float((pooled[torch.float32][a].clamp_min(0) if a == 'stn' else pooled[torch.float32][a]).sub(ref).abs().max()) / sc:.2e}")
    worst = []
    for n, p in m.named_parameters():
        g64 = res[torch.float64][n]
        sc = float(g64.abs().max())
        if sc < 1e-6:
            continue
        eh = float((grads[id(p)].cpu().double() - g64).abs().max()) / sc
        eo = float((res[torch.float32][n] - g64).abs().max()) / sc
        worst.append((eh, eo, n))
    worst.sort(reverse=True)
    for eh, eo, n in worst[:6]:
        print(f"  {n:26s} hip {eh:.2e}  f32 oracle {eo:.2e}")
