"""Error behaviour of the C ABI: a bad call returns its status code with a message and launches nothing
(the reference fails hard on the same mistakes: MaxPool1d(num_points) on a wrong P, model/pointnet.py:132)."""
import ctypes

import pytest
import torch

from pointnetgpd_amd import _lib, synth
from pointnetgpd_amd.pointnet import PointNetCls, STN3d

pytestmark = pytest.mark.gpu
ERR_INVALID, ERR_SHAPE, ERR_HIP, ERR_STATE, ERR_WORKSPACE = 1, 2, 3, 4, 5


def vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def last_error(lib):
    return lib.pngpd_last_error().decode()


@pytest.fixture(scope="module")
def loaded():
    P, k, B = 64, 2, 4
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(1, k, 0.02))
    m = m.cuda().eval()
    x = torch.from_numpy(synth.candidate_points(1, B, P)).cuda()
    m(x)                                            # creates the handle and loads the weights
    h = m._handle(x.device)
    return m, h, x, B, P, k


def infer(h, x, B, P, layout=_lib.LAYOUT_C3P, precision=_lib.FP32, logp=True, ws=None, ws_bytes=None, ptr=None):
    out = torch.empty((B, 2), dtype=torch.float32, device="cuda") if logp else None
    ws = h.workspace(B) if ws is None else ws
    return h.lib.pngpd_forward_infer(h.ptr if ptr is None else ptr, vp(x), layout, B, P, precision, vp(out), None, None,
                                     vp(ws), ws.numel() if ws_bytes is None else ws_bytes,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def test_forward_rejects_bad_calls(loaded):
    m, h, x, B, P, k = loaded
    assert infer(h, x, B, P) == _lib.OK
    assert infer(h, x, B, P + 1) == ERR_SHAPE and "num_points=64" in last_error(h.lib)
    assert infer(h, x, 0, P) == ERR_SHAPE
    assert infer(h, x, B, P, layout=7) == ERR_INVALID
    assert infer(h, x, B, P, precision=9) == ERR_INVALID
    assert infer(h, None, B, P) == ERR_INVALID
    assert infer(h, x, B, P, ws_bytes=16) == ERR_WORKSPACE and "workspace" in last_error(h.lib)


def test_forward_needs_loaded_weights_and_the_requested_part(loaded):
    m, h, x, B, P, k = loaded
    lib = h.lib
    fresh = ctypes.c_void_p()
    assert lib.pngpd_model_create(ctypes.byref(fresh), P, k, 7) == _lib.OK
    try:
        assert infer(h, x, B, P, ptr=fresh) == ERR_STATE and "not loaded" in last_error(lib)
    finally:
        lib.pngpd_model_destroy(fresh)
    assert lib.pngpd_model_create(ctypes.byref(fresh), P, 99, 7) == ERR_SHAPE
    assert lib.pngpd_model_create(ctypes.byref(fresh), P, k, 2) == ERR_INVALID          # a trunk without its STN
    stn = STN3d(P, 3).cuda().eval()
    stn(x)
    hs = stn._handle(x.device)
    assert infer(hs, x, B, P, ws=h.workspace(B)) == ERR_STATE and "head" in last_error(lib)


def test_crop_rejects_bad_calls():
    lib = _lib.load()
    S, G = 1000, 4
    scene = torch.zeros((S, 3), dtype=torch.float32, device="cuda")
    grasps = torch.zeros((G, 5, 3), dtype=torch.float64, device="cuda")
    lim = (ctypes.c_double * 6)(-1, 1, -1, 1, -1, 1)
    nbytes = lib.pngpd_crop_workspace_bytes(S, G)
    assert nbytes > 0 and lib.pngpd_crop_workspace_bytes(0, G) == 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    cnt = torch.empty(G, dtype=torch.int32, device="cuda")
    off = torch.empty(G + 1, dtype=torch.int64, device="cuda")
    idx = torch.empty(16, dtype=torch.int32, device="cuda")
    pts = torch.empty((G * 3, 8, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty(G, dtype=torch.uint8, device="cuda")
    flag = torch.empty(1, dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.pngpd_crop_count(vp(scene), S, vp(grasps), G, lim, vp(ws), nbytes - 1, vp(cnt), vp(off), st) == ERR_WORKSPACE
    assert lib.pngpd_crop_count(vp(scene), S, None, G, lim, vp(ws), nbytes, vp(cnt), vp(off), st) == ERR_INVALID
    assert lib.pngpd_crop_count(vp(scene), 0, vp(grasps), G, lim, vp(ws), nbytes, vp(cnt), vp(off), st) == ERR_SHAPE
    assert lib.pngpd_crop_emit(vp(scene), S, G, lim, vp(ws), nbytes, vp(cnt), vp(off), vp(idx), -1, None, st) == ERR_SHAPE
    assert lib.pngpd_crop_subsample(vp(scene), S, vp(grasps), G, lim, vp(ws), nbytes, 0, 8, 20, 0, 0, vp(pts), vp(valid),
                                    vp(cnt), vp(flag), st) == ERR_SHAPE
    assert lib.pngpd_crop_subsample(vp(scene), S, vp(grasps), G, lim, vp(ws), nbytes, 3, 8, 20, 0, 0, vp(pts), vp(valid),
                                    vp(cnt), None, st) == ERR_INVALID
    assert lib.pngpd_vote(vp(cnt), vp(cnt), None, G, 3, 1, vp(cnt), vp(cnt), st) == ERR_SHAPE
    rec = torch.zeros((G, 10), dtype=torch.float64, device="cuda")
    frames = torch.empty((G, 12), dtype=torch.float64, device="cuda")
    T = (ctypes.c_double * 16)(*([0.0] * 16))
    assert lib.pngpd_dataset_frames(vp(rec), G, 7, T, 0.06, vp(frames), st) == ERR_SHAPE and "8 fields" in last_error(lib)
    assert lib.pngpd_dataset_frames(vp(rec), G, 10, None, 0.06, vp(frames), st) == ERR_INVALID
    assert lib.pngpd_dataset_frames(vp(rec), G, 10, T, 0.06, vp(frames), st) == _lib.OK


def test_training_rejects_bad_calls():
    lib = _lib.load()
    B, P, k = 4, 32, 2
    assert lib.pngpd_train_workspace_bytes(0, P, k) == 0
    nbytes = lib.pngpd_train_workspace_bytes(B, P, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    x = torch.zeros((B, 3, P), dtype=torch.float32, device="cuda")
    logp = torch.empty((B, k), dtype=torch.float32, device="cuda")
    table = (ctypes.c_void_p * _lib.SLOTS)()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # a table without tensors, a batch of one, a short table, a short workspace
    assert lib.pngpd_train_forward(B, P, k, vp(x), table, _lib.SLOTS, 0.1, vp(logp), None, None, vp(ws), nbytes, st) == ERR_INVALID
    assert "unit 0" in last_error(lib)
    assert lib.pngpd_train_forward(1, P, k, vp(x), table, _lib.SLOTS, 0.1, vp(logp), None, None, vp(ws), nbytes, st) == ERR_SHAPE
    assert lib.pngpd_train_forward(B, P, k, vp(x), table, 5, 0.1, vp(logp), None, None, vp(ws), nbytes, st) == ERR_INVALID
    assert lib.pngpd_train_forward(B, P, k, vp(x), table, _lib.SLOTS, 0.1, vp(logp), None, None, vp(ws), 64, st) == ERR_WORKSPACE
    full = (ctypes.c_void_p * _lib.SLOTS)(*[x.data_ptr()] * _lib.SLOTS)
    for mask in (0, 8, 15):
        assert lib.pngpd_train_backward_part(B, P, k, vp(x), full, full, _lib.SLOTS, vp(logp), vp(logp), None, None,
                                             vp(ws), nbytes, mask, st) == ERR_INVALID and "part mask" in last_error(lib)
    assert lib.pngpd_train_backward_part(B, P, k, vp(x), full, full, _lib.SLOTS, vp(logp), vp(logp), None, vp(x), vp(ws),
                                         nbytes, 7, st) == ERR_INVALID and "dfeat_dev" in last_error(lib)
    assert lib.pngpd_train_forward_parts(2, B, P, k, vp(x), full, _lib.SLOTS, 0.1, vp(logp), None, None, vp(ws), nbytes,
                                         st) == ERR_INVALID
    assert lib.pngpd_adam_step(vp(x), vp(x), vp(x), vp(x), 0, 0.1, 0.9, 0.999, 1e-8, 1, 1.0, st) == ERR_SHAPE
    assert lib.pngpd_nll_loss(vp(logp), None, B, k, vp(logp), vp(logp), st) == ERR_INVALID
