"""Batched counterpart of the reference's per-candidate scoring shim.

test_network() in the reference (main_test.py:73-83) takes ONE candidate as a [P,3] numpy array,
runs a batch-1 forward and returns (arg-max class, softmax probabilities).  The online loop calls
it three times per grasp (apps/kinect2grasp.py:764-797).  Here the whole candidate set is scored
in one launch sequence; row i of the result equals test_network applied to candidate i.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


def score_logp(logp: torch.Tensor):
    """softmax(log-probs) and first-index arg-max on the device: (pred[B] int64, probs[B,k])."""
    if not logp.is_cuda:
        raise RuntimeError("score_logp needs a ROCm tensor (there is no CPU path)")
    logp = logp.contiguous().float()
    B, k = logp.shape
    probs = torch.empty_like(logp)
    pred = torch.empty(B, dtype=torch.int64, device=logp.device)
    lib = _lib.load()
    with torch.cuda.device(logp.device):
        stream = torch.cuda.current_stream(logp.device).cuda_stream
        _lib.check(lib.pngpd_score(ctypes.c_void_p(logp.data_ptr()), B, k, ctypes.c_void_p(probs.data_ptr()),
                                   ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(stream)))
    return pred, probs


def score_candidates(model, points: torch.Tensor):
    """points [B,P,3] on the device -> (pred[B], probs[B,k], trans[B,3,3])."""
    logp, trans = model.forward_points(points)
    pred, probs = score_logp(logp)
    return pred, probs, trans


def test_network(model_, local_pc):
    """Signature-compatible with main_test.py:73-83: one [P,3] numpy cloud in, (pred[1], probs[1,k]) out.

    The tensor is moved to the model's device, scored through the batched path with B=1, and the
    probabilities come back as numpy like in the reference.
    """
    dev = next(model_.parameters()).device
    pts = torch.as_tensor(np.asarray(local_pc), dtype=torch.float32, device=dev)[None]
    pred, probs, _ = score_candidates(model_, pts)
    return pred[:1], probs.cpu().numpy()


test_network.__test__ = False  # not a pytest test
