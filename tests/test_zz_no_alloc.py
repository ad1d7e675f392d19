"""pngpd_forward_infer_noalloc (the entry point for callers that cannot allocate device memory) against
pngpd_forward_infer, through the C ABI: bit-identical outputs on two calls in a row, the second with a smaller
batch, in both arithmetic modes; a batch beyond its stated maximum is refused."""
import ctypes

import pytest
import torch

from pointnetgpd_amd import _lib, synth
from pointnetgpd_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu


def vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _outputs(B, k):
    return (torch.full((B, k), float("nan"), device="cuda"), torch.full((B, 3, 3), float("nan"), device="cuda"),
            torch.full((B, 1024), float("nan"), device="cuda"))


@pytest.mark.parametrize("precision", [_lib.FP32, _lib.BF16, _lib.BF16X3])
def test_no_alloc_forward_equals_forward_infer(precision):
    P, k = 750, 2
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(11, k, 0.02))
    m = m.cuda().eval()
    x_all = torch.from_numpy(synth.candidate_points(11, _lib.NOALLOC_MAX_B, P, "gripper")).cuda()
    m(x_all[:2])                                       # creates the handle, uploads the weights
    h = m._handle(x_all.device)
    lib = h.lib
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B in (_lib.NOALLOC_MAX_B, 37, 1):              # each call after the first is smaller than the one before
        x = x_all[:B].contiguous()
        ws = h.workspace(B)
        ref = _outputs(B, k)
        got = _outputs(B, k)
        _lib.check(lib.pngpd_forward_infer(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, precision, vp(ref[0]), vp(ref[1]),
                                           vp(ref[2]), vp(ws), ws.numel(), st))
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        _lib.check(lib.pngpd_forward_infer_noalloc(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, precision, vp(got[0]),
                                                   vp(got[1]), vp(got[2]), st))
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        for r, g, name in zip(ref, got, ("logp", "trans", "feat")):
            assert not torch.isnan(g).any(), name
            assert torch.equal(r, g), f"B={B} {name}: max diff {(r - g).abs().max().item()}"
    # outputs the caller does not want: the transform then sits in the library's own array
    B = 5
    x = x_all[:B].contiguous()
    ref, got = _outputs(B, k), _outputs(B, k)
    ws = h.workspace(B)
    _lib.check(lib.pngpd_forward_infer(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, precision, vp(ref[0]), None, None, vp(ws),
                                       ws.numel(), st))
    _lib.check(lib.pngpd_forward_infer_noalloc(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, precision, vp(got[0]), None, None,
                                               st))
    torch.cuda.synchronize()
    assert torch.equal(ref[0], got[0])


def test_no_alloc_forward_refuses_a_batch_beyond_its_maximum():
    P, k = 64, 2
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(3, k, 0.02))
    m = m.cuda().eval()
    B = _lib.NOALLOC_MAX_B + 1
    x = torch.from_numpy(synth.candidate_points(3, B, P)).cuda()
    m(x[:2])
    h = m._handle(x.device)
    out = torch.empty((B, k), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = h.lib.pngpd_forward_infer_noalloc(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, _lib.FP32, vp(out), None, None, st)
    assert rc == 2 and "at most 256" in h.lib.pngpd_last_error().decode()
    rc = h.lib.pngpd_forward_infer_noalloc(h.ptr, vp(x), _lib.LAYOUT_C3P, B - 1, P + 1, _lib.FP32, vp(out), None, None, st)
    assert rc == 2 and "num_points" in h.lib.pngpd_last_error().decode()


def test_no_alloc_forward_refuses_to_overlap_on_two_streams():
    """All forwards on a GPU share the one scratch array.  A call on another stream while the previous forward is
    still running on the GPU would corrupt it: the library refuses the call (PNGPD_ERR_STATE) instead; once the
    first stream has drained, the same call goes through and gives the same rows."""
    P, k, B = 750, 2, _lib.NOALLOC_MAX_B
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(5, k, 0.02))
    m = m.cuda().eval()
    x = torch.from_numpy(synth.candidate_points(5, B, P)).cuda()
    ref = m(x)[0]
    h = m._handle(x.device)
    prec = _lib.PRECISIONS[str(m.precision)]
    out1, out2 = torch.empty((B, k), device="cuda"), torch.empty((B, k), device="cuda")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    st1, st2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
    call = h.lib.pngpd_forward_infer_noalloc
    for _ in range(300):        # a queue the GPU needs tens of milliseconds for; enqueueing it takes the host far less
        assert call(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, prec, vp(out1), None, None, st1) == 0
    rc = call(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, prec, vp(out2), None, None, st2)
    assert rc == 4 and "another stream" in h.lib.pngpd_last_error().decode()
    s1.synchronize()
    assert call(h.ptr, vp(x), _lib.LAYOUT_C3P, B, P, prec, vp(out2), None, None, st2) == 0
    s2.synchronize()
    assert torch.equal(out1, ref) and torch.equal(out2, ref)
