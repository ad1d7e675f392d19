import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
P,k,seed,B,gain=750,2,21,4,1.7
x=torch.from_numpy(synth.candidate_points(seed,B,P,"normal"))
target=torch.from_numpy((synth.uniform(seed,"labels",(B,))<0.5).astype(np.int64))
sd=synth.torch_state_dict(seed,k,1.0,gain)
o=po.build_cls(P,k,sd,eval_mode=False)
out,_=o(x); l=F.nll_loss(out,target); l.backward()
ref={n:p.grad.numpy() for n,p in o.named_parameters()}
m=PointNetCls(P,3,k); m.load_state_dict(sd); m=m.cuda().train()
out2,_=m(x.cuda()); l2=F.nll_loss(out2,target.cuda()); l2.backward()
print("loss",l.item(),l2.item())
for n,p in m.named_parameters():
    g=p.grad.cpu().numpy(); r=ref[n]
    e=np.abs(g-r).max()/max(np.abs(r).max(),1e-12)
    print(f"{n:28s} relerr {e:9.2e}  |ref|max {np.abs(r).max():9.2e} |got|max {np.abs(g).max():9.2e}")
