"""CPU emulation of the bf16 conv stacks: which rounding contributes how much to the log-prob error."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def fold(conv, bn):
    s = bn.weight.double()/torch.sqrt(bn.running_var.double()+1e-5)
    W = (conv.weight.double().squeeze(-1)*s[:,None]).float(); b=((conv.bias.double()-bn.running_mean.double())*s+bn.bias.double()).float()
    return W,b
def stack(mod, x, last_relu, r):   # x [B,3,P]; r: dict of flags h1,w2,h2,w3
    W1,b1=fold(mod.conv1,mod.bn1); W2,b2=fold(mod.conv2,mod.bn2); W3,b3=fold(mod.conv3,mod.bn3)
    h1=F.relu(torch.einsum('oc,bcp->bop',W1,x)+b1[None,:,None])
    if r['h1']: h1=bf(h1)
    if r['w2']: W2=bf(W2)
    h2=F.relu(torch.einsum('oc,bcp->bop',W2,h1)+b2[None,:,None])
    if r['h2']: h2=bf(h2)
    if r['w3']: W3=bf(W3)
    y=torch.einsum('oc,bcp->bop',W3,h2).max(2).values+b3[None]
    return F.relu(y) if last_relu else y
def fc(x, lins, bns):
    for lin,bn in zip(lins[:2],bns):
        x=F.relu(bn(lin(x)))
    return lins[2](x)
def run(m,x,rs,rt):
    stn=m.feat.stn
    t=fc(stack(stn,x,True,rs),[stn.fc1,stn.fc2,stn.fc3],[stn.bn4,stn.bn5])+torch.eye(3).reshape(1,9)
    xt=torch.bmm(x.transpose(2,1),t.view(-1,3,3)).transpose(2,1)
    return F.log_softmax(fc(stack(m.feat,xt,False,rt),[m.fc1,m.fc2,m.fc3],[m.bn1,m.bn2]),-1), t
none=dict(h1=0,w2=0,h2=0,w3=0); allr=dict(h1=1,w2=1,h2=1,w3=1)
for tag,P,k,seed,kind,scale,gain in [("grip g2.0",500,2,43,"gripper",0.02,2.0),("norm g1.7",750,2,12,"normal",1.0,1.7)]:
    m=po.build_cls(P,k,synth.torch_state_dict(seed,k,scale,gain))
    x=torch.from_numpy(synth.candidate_points(seed,32,P,kind))
    with torch.no_grad():
        ref,tref=run(m,x,none,none)
        print(tag,"logit gap mean %.2f"%(ref[:,0]-ref[:,1]).abs().mean().item())
        def err(rs,rt):
            lp,t=run(m,x,rs,rt); return (lp-ref).abs().max().item(), (t-tref).abs().max().item()
        print("  all bf16          ",err(allr,allr))
        print("  stn only          ",err(allr,none))
        print("  trunk only        ",err(none,allr))
        for key in ['h1','w2','h2','w3']:
            r=dict(none); r[key]=1
            print("  only %s both stacks"%key, err(r,r))
        r=dict(allr); r['h1']=0; r['w2']=0
        print("  L2 exact (h1,w2 fp32) both", err(r,r))
