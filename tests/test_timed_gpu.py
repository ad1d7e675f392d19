"""The diagnostic entry points bench.py takes its roofline figures from: same results as the plain calls,
plausible kernel durations."""
import ctypes

import pytest
import torch

from pointnetgpd_amd import _lib, synth, training
from pointnetgpd_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu


def _model(P, k, train):
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(3, k, 0.02, 1.0))
    m = m.cuda()
    return m.train() if train else m.eval()


def test_timed_inference_forward_equals_the_plain_one():
    B, P, k = 37, 300, 2
    m = _model(P, k, False)
    m.precision = "bf16"
    x = torch.from_numpy(synth.candidate_points(5, B, P)).cuda()
    logp_ref, trans_ref = m(x)
    h = m._handle(x.device)
    ws = h.workspace(B)
    logp = torch.empty((B, k), dtype=torch.float32, device=x.device)
    trans = torch.empty((B, 9), dtype=torch.float32, device=x.device)
    pair = (ctypes.c_float * 2)()
    _lib.check(h.lib.pngpd_forward_infer_timed(h.ptr, ctypes.c_void_p(x.data_ptr()), _lib.LAYOUT_C3P, B, P,
                                               _lib.PRECISIONS["bf16"], ctypes.c_void_p(logp.data_ptr()),
                                               ctypes.c_void_p(trans.data_ptr()), None, ctypes.c_void_p(ws.data_ptr()),
                                               ws.numel(), pair, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(logp, logp_ref) and torch.equal(trans.view(B, 3, 3), trans_ref)
    assert 0.0 < pair[0] < 50.0 and 0.0 < pair[1] < 50.0


def test_timed_training_forward_equals_the_plain_one():
    B, P, k = 6, 130, 2
    x = torch.from_numpy(synth.candidate_points(7, B, P)).cuda()
    a, b = _model(P, k, True), _model(P, k, True)
    logp_ref, trans_ref, _, _ = training.train_forward(a, x)
    ms = []
    logp, trans, _, _ = training.train_forward(b, x, conv3_ms=ms)
    assert torch.equal(logp, logp_ref) and torch.equal(trans, trans_ref)
    assert torch.equal(a.feat.bn3.running_var, b.feat.bn3.running_var)
    assert len(ms) == 2 and all(0.0 < v < 50.0 for v in ms)
