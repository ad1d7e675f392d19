import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
m = PointNetCls(750, 3, 2); m.load_state_dict(synth.torch_state_dict(0, 2, 0.02)); m = m.cuda().eval(); m.precision = sys.argv[1]
x = torch.from_numpy(synth.candidate_points(0, int(sys.argv[2]), 750)).cuda()
for _ in range(200): m(x)
torch.cuda.synchronize()
