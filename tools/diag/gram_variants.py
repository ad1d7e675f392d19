"""Diagnostic: BatchNorm statistics of the two conv3 layers after one training step with the Gram product on the float64
matrix instruction (PNGPD_TRAIN_GRAM=f64) and on bf16 pieces (default); each variant runs in a process of its own
because the library reads the switch once.      python tools/diag/gram_variants.py [B]
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(out, B):
    import torch
    from pointnetgpd_amd import synth
    from pointnetgpd_amd.pointnet import PointNetCls
    from pointnetgpd_amd.training import Trainer
    P, k = 750, 2
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(5, k, 0.02), strict=True)
    m = m.cuda().train()
    x = torch.from_numpy(synth.candidate_points(9, B, P)).cuda()
    t = torch.randint(0, k, (B,), generator=torch.Generator().manual_seed(1)).cuda()
    tr = Trainer(m, lr=1e-3)
    loss, logp = tr.step(x, t)
    torch.cuda.synchronize()
    from pointnetgpd_amd.training import critical_points, saved_activation
    stn_arg, trunk_arg = critical_points(tr._ws, B, P, k)
    masks = {f"mask_{n}": (saved_activation(tr._ws, B, P, k, n) > 0).cpu().numpy() for n in ("stn_h2", "trunk_h2", "stn_pooled", "head_a1")}
    np.savez(out, stn_arg=stn_arg.cpu().numpy(), trunk_arg=trunk_arg.cpu().numpy(), **masks, loss=loss.cpu().numpy(), logp=logp.detach().cpu().numpy(), grad=tr.flat_g.cpu().numpy(),
             **{n.replace(".", "_"): b.detach().cpu().numpy() for n, b in m.named_buffers() if "conv3" not in n and "bn3" in n})


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    res = {}
    for v in ("f64", "x6"):
        out = f"/tmp/gram_{v}.npz"
        env = dict(os.environ, PNGPD_TRAIN_GRAM=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", out, str(B)], check=True, env=env)
        res[v] = np.load(out)
    for key in res["f64"].files:
        if key.endswith("_arg") or key.startswith("mask_"):
            d = res["f64"][key] != res["x6"][key]
            print(f"{key:40s} {int(d.sum())} of {d.size} decisions differ")
            continue
        a, b = res["f64"][key].astype(np.float64), res["x6"][key].astype(np.float64)
        scale = np.abs(a).max() + 1e-30
        print(f"{key:40s} max |diff| {np.abs(a - b).max():.3e}   relative to largest entry {np.abs(a - b).max() / scale:.3e}")
