"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference network in eager PyTorch.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (pointnetgpd_amd) never does.

It restates, op for op, the fp32 eager sequence of the reference so that it can stand in for the
reference on machines where /root/reference does not exist:

  * STN3d         model/pointnet.py:8-45    conv1d(k=1)+BN+ReLU x3, MaxPool1d(num_points),
                                            Linear+BN+ReLU x2, Linear, + flattened 3x3 identity
  * PointNetfeat  model/pointnet.py:122-152 bmm(x^T, trans)^T, conv+BN+ReLU x2, conv+BN (no ReLU),
                                            MaxPool1d(num_points)
  * PointNetCls   model/pointnet.py:174-190 Linear+BN+ReLU x2, Linear, log_softmax
  * test_network  main_test.py:73-83        [P,3] numpy -> [1,3,P] float tensor -> softmax, argmax

Pinning: tests/golden/make_golden.py imports the real reference classes in the build container,
loads the same seeded state_dict into both and stores the reference's outputs under tests/golden/;
tests/test_oracle.py checks this restatement against those files.  The reference ships no tests or
golden vectors of its own for this path.

The attribute names below are dictated by the reference state_dict (74 keys); the modules are
assembled from a table rather than written out.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_POINT_MLP = ((3, 64), (64, 128), (128, 1024))
_REGRESSOR = ((1024, 512), (512, 256))


def _add_point_mlp(mod: nn.Module, input_chann: int) -> None:
    for i, (cin, cout) in enumerate(_POINT_MLP, start=1):
        setattr(mod, f"conv{i}", nn.Conv1d(input_chann if i == 1 else cin, cout, 1))
        setattr(mod, f"bn{i}", nn.BatchNorm1d(cout))


def _point_mlp(mod: nn.Module, x: torch.Tensor, last_relu: bool) -> torch.Tensor:
    for i in (1, 2, 3):
        x = getattr(mod, f"bn{i}")(getattr(mod, f"conv{i}")(x))
        if i < 3 or last_relu:
            x = F.relu(x)
    return x


class OracleSTN3d(nn.Module):
    def __init__(self, num_points=2500, input_chann=3):
        super().__init__()
        self.num_points = num_points
        _add_point_mlp(self, input_chann)
        self.mp1 = nn.MaxPool1d(num_points)
        for i, (cin, cout) in enumerate(_REGRESSOR, start=1):
            setattr(self, f"fc{i}", nn.Linear(cin, cout))
            setattr(self, f"bn{i + 3}", nn.BatchNorm1d(cout))
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()

    def pooled(self, x):
        return self.mp1(_point_mlp(self, x, last_relu=True)).view(-1, 1024)

    def forward(self, x):
        g = self.pooled(x)
        g = F.relu(self.bn4(self.fc1(g)))
        g = F.relu(self.bn5(self.fc2(g)))
        g = self.fc3(g)
        eye = torch.eye(3, dtype=g.dtype, device=g.device).reshape(1, 9)
        return (g + eye).view(-1, 3, 3)


class OraclePointNetfeat(nn.Module):
    def __init__(self, num_points=2500, input_chann=3, global_feat=True):
        super().__init__()
        self.stn = OracleSTN3d(num_points=num_points, input_chann=input_chann)
        _add_point_mlp(self, input_chann)
        self.mp1 = nn.MaxPool1d(num_points)
        self.num_points = num_points
        self.global_feat = global_feat

    def forward(self, x):
        trans = self.stn(x)
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
        x = self.mp1(_point_mlp(self, x, last_relu=False)).view(-1, 1024)
        if not self.global_feat:
            raise NotImplementedError("per-point features are outside the scored path")
        return x, trans


class OraclePointNetCls(nn.Module):
    def __init__(self, num_points=2500, input_chann=3, k=2):
        super().__init__()
        self.num_points = num_points
        self.feat = OraclePointNetfeat(num_points, input_chann=input_chann, global_feat=True)
        dims = (1024, 512, 256)
        for i in (1, 2):
            setattr(self, f"fc{i}", nn.Linear(dims[i - 1], dims[i]))
            setattr(self, f"bn{i}", nn.BatchNorm1d(dims[i]))
        self.fc3 = nn.Linear(256, k)
        self.relu = nn.ReLU()

    def forward(self, x):
        g, trans = self.feat(x)
        g = F.relu(self.bn1(self.fc1(g)))
        g = F.relu(self.bn2(self.fc2(g)))
        return F.log_softmax(self.fc3(g), dim=-1), trans


def oracle_test_network(model: nn.Module, local_pc: np.ndarray):
    """main_test.py:73-83 for one candidate: local_pc is [P,3]; returns (pred[1], probs[1,k])."""
    t = torch.FloatTensor(local_pc.T[np.newaxis, ...])
    with torch.no_grad():
        logp, _ = model(t)
    probs = logp.softmax(1)
    pred = probs.max(1, keepdim=True)[1]
    return pred[0], probs.numpy()


def build_cls(num_points: int, k: int, state_dict: dict, eval_mode: bool = True) -> OraclePointNetCls:
    m = OraclePointNetCls(num_points=num_points, input_chann=3, k=k)
    m.load_state_dict(state_dict, strict=True)
    return m.eval() if eval_mode else m.train()
