"""MFMA lane-map probes: the formulas the fused kernels use, checked on real hardware."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe():
    import torch  # noqa: F401  (loads the HIP runtime the library must share)
    path = os.path.join(ROOT, "tests", "_probe.so")
    assert os.path.exists(path), "tests/_probe.so missing: run python -c 'import __graft_entry__ as g; g.build()'"
    return ctypes.CDLL(path)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf16_round(a):
    import torch
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).bfloat16().float().numpy()


def test_axpy_on_torch_stream(probe):
    import torch
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rc = probe.probe_axpy(_ptr(x), ctypes.c_float(2.0), 1000, _stream())
    assert rc == 0
    s.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32) * 2 + 1)


def test_mfma_bf16_layout(probe):
    import torch
    rng = np.random.default_rng(0)
    A = _bf16_round(rng.integers(-4, 5, (32, 16)))
    B = _bf16_round(rng.integers(-4, 5, (16, 32)))
    D = torch.zeros(32, 32, device="cuda")
    assert probe.probe_mfma_bf16(_ptr(_dev(A)), _ptr(_dev(B)), _ptr(D), _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(D.cpu().numpy(), A @ B)


def test_mfma_f32_layout(probe):
    import torch
    rng = np.random.default_rng(1)
    A = rng.integers(-4, 5, (32, 2)).astype(np.float32)
    B = rng.integers(-4, 5, (2, 32)).astype(np.float32)
    D = torch.zeros(32, 32, device="cuda")
    assert probe.probe_mfma_f32(_ptr(_dev(A)), _ptr(_dev(B)), _ptr(D), _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(D.cpu().numpy(), A @ B)


@pytest.mark.parametrize("name", ["probe_chain_f32"])
def test_accumulator_as_operand(probe, name):
    import torch
    rng = np.random.default_rng(2)
    W = rng.integers(-3, 4, (32, 16)).astype(np.float32)
    H = rng.integers(-3, 4, (16, 32)).astype(np.float32)
    V = rng.integers(-3, 4, (32, 32)).astype(np.float32)
    Z = torch.zeros(32, 32, device="cuda")
    assert getattr(probe, name)(_ptr(_dev(W)), _ptr(_dev(H)), _ptr(_dev(V)), _ptr(Z), _stream()) == 0
    torch.cuda.synchronize()
    X = np.maximum(W @ H, 0)          # small integers: exact in bf16 (|x| <= 144 < 256)
    np.testing.assert_array_equal(Z.cpu().numpy(), X.T @ V)


def test_mfma_bf16_16x16x32_layout(probe):
    import torch
    rng = np.random.default_rng(3)
    A = _bf16_round(rng.integers(-4, 5, (16, 32)))
    B = _bf16_round(rng.integers(-4, 5, (32, 16)))
    D = torch.zeros(16, 16, device="cuda")
    assert probe.probe_mfma_bf16_16(_ptr(_dev(A)), _ptr(_dev(B)), _ptr(D), _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(D.cpu().numpy(), A @ B)


def test_accumulator_as_operand_16x16x32(probe):
    import torch
    rng = np.random.default_rng(4)
    W = rng.integers(-2, 3, (32, 32)).astype(np.float32)
    H = rng.integers(-2, 3, (32, 16)).astype(np.float32)
    V = rng.integers(-3, 4, (32, 16)).astype(np.float32)
    Z = torch.zeros(16, 16, device="cuda")
    assert probe.probe_chain_bf16_16(_ptr(_dev(W)), _ptr(_dev(H)), _ptr(_dev(V)), _ptr(Z), _stream()) == 0
    torch.cuda.synchronize()
    X = np.maximum(W @ H, 0)          # |x| <= 128: exact in bf16
    np.testing.assert_array_equal(Z.cpu().numpy(), X.T @ V)
