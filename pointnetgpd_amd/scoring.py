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


def vote_on_cloud(model, local_pc, repeat=10, seed=None):
    """main_test.py:86-109 for one gripper-frame cloud [n,3]: `repeat` draws of num_points points (without
    replacement when the cloud has enough, with replacement otherwise; each draw is taken from the previous one, as
    the reference's loop does), every draw scored, the most frequent class returned with the per-draw classes.
    The draws are scored in one batch instead of one forward each."""
    rng = np.random.default_rng(seed)
    num_point = int(model.num_points)
    cloud = np.asarray(local_pc, dtype=np.float32)
    draws = []
    for _ in range(repeat):
        cloud = cloud[rng.choice(len(cloud), num_point, replace=len(cloud) < num_point)]
        draws.append(cloud)
    dev = next(model.parameters()).device
    pred, _, _ = score_candidates(model, torch.as_tensor(np.stack(draws), device=dev))
    votes = pred.cpu().numpy()
    classes, counts = np.unique(votes, return_counts=True)
    return int(classes[np.argmax(counts)]), votes          # scipy.stats.mode: the smallest of the most frequent


def main(argv=None):
    """Command line of main_test.py (same flags; --cuda is implied, there is no CPU path), plus --points FILE for
    a real cloud (.npy, [n,3], gripper frame) in place of the reference's random test cloud."""
    import argparse
    import time
    from . import compat
    ap = argparse.ArgumentParser(description="pointnetGPD")
    ap.add_argument("--cuda", action="store_true", default=True)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--load-model", type=str, default="../data/pointgpd_3class.model")
    ap.add_argument("--show_final_grasp", action="store_true", default=False)
    ap.add_argument("--tray_grasp", action="store_true", default=False)
    ap.add_argument("--using_mp", action="store_true", default=False)
    ap.add_argument("--model_type", type=str)
    ap.add_argument("--points", type=str, default=None)
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16x3", "bf16"])
    args = ap.parse_args(argv)
    by_type = {"500": "./data/1v_500_2class_ourway2sample.model", "750": "./data/1v_750_2class_ourway2sample.model",
               "3class": "./data/pointgpd_3class.model"}
    if args.model_type in by_type:
        args.load_model = by_type[args.model_type]
    else:
        print("Using default model file")
    model = compat.load_checkpoint(args.load_model, device=f"cuda:{max(args.gpu, 0)}")
    if args.precision:
        model.precision = args.precision
    print("load model {}".format(args.load_model))
    if args.points:
        local_pc = np.load(args.points)
        seed = int(time.time())
    else:
        seed = int(time.time())
        local_pc = np.random.default_rng(seed).random((500, 3))          # test only, as in the reference
    predict, votes = vote_on_cloud(model, local_pc, repeat=10, seed=seed)
    print("voting: ", list(votes))
    print("Test result:", predict)
    return predict


if __name__ == "__main__":
    main()
