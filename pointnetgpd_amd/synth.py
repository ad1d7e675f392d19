"""Deterministic synthetic weights and inputs.

The reference's checkpoints (data/1v_*_2class_ourway2sample.model) are not available, so tests and
the benchmark use seeded synthetic parameters.  Everything here is a pure function of (seed, name)
built on a counter-based splitmix64 generator, so the same tensors are regenerated bit-for-bit on
any machine instead of being shipped as files.
"""
from __future__ import annotations

import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser applied to an array of uint64 counters."""
    with np.errstate(over="ignore"):
        z = (counter + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    tag = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64(((seed & 0xFFFFFFFF) << 32) | tag)


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0, dtype=np.float32) -> np.ndarray:
    """U[lo, hi) values, 53-bit resolution, keyed by (seed, name)."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    base = splitmix64(np.array([_stream_base(seed, name)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _MASK
    bits = splitmix64(ctr) >> np.uint64(11)
    u = bits.astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).reshape(shape).astype(dtype)


def normal(seed: int, name: str, shape, dtype=np.float32) -> np.ndarray:
    """N(0,1) by Box-Muller on two keyed uniform streams."""
    u1 = uniform(seed, name + "/u1", shape, dtype=np.float64)
    u2 = uniform(seed, name + "/u2", shape, dtype=np.float64)
    return (np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)).astype(dtype)


# (attribute prefix, out, in, has_bn, bn attribute) for the 12 layer units of PointNetCls, in the
# order of the C ABI's weight table.
CLS_UNITS = (
    ("feat.stn.conv1", 64, 3, "feat.stn.bn1"),
    ("feat.stn.conv2", 128, 64, "feat.stn.bn2"),
    ("feat.stn.conv3", 1024, 128, "feat.stn.bn3"),
    ("feat.stn.fc1", 512, 1024, "feat.stn.bn4"),
    ("feat.stn.fc2", 256, 512, "feat.stn.bn5"),
    ("feat.stn.fc3", 9, 256, None),
    ("feat.conv1", 64, 3, "feat.bn1"),
    ("feat.conv2", 128, 64, "feat.bn2"),
    ("feat.conv3", 1024, 128, "feat.bn3"),
    ("fc1", 512, 1024, "bn1"),
    ("fc2", 256, 512, "bn2"),
    ("fc3", None, 256, None),  # out = k
)


def make_state_dict(seed: int, k: int = 2, input_scale: float = 1.0, gain: float = 2.0) -> dict:
    """A full 74-entry PointNetCls state_dict as numpy arrays.

    Layer weights are `gain` times the U(+-1/sqrt(fan_in)) scale of torch's default init (gain 2
    spreads the class scores of different candidates by O(1) instead of 1e-3); the first layer of
    each stack is divided by `input_scale` so that inputs of that magnitude (e.g. centimetre-scale
    gripper coordinates) still drive the network.  BatchNorm affine parameters and running statistics are
    randomised (about one channel in ten gets a negative gamma) so that folding mistakes show.
    """
    sd = {}
    for name, n_out, n_in, bn in CLS_UNITS:
        n_out = k if n_out is None else n_out
        bound = 1.0 / np.sqrt(n_in)
        wshape = (n_out, n_in, 1) if "conv" in name else (n_out, n_in)
        w = uniform(seed, name + ".weight", wshape, -bound, bound) * np.float32(gain)
        if name.endswith("conv1"):
            w = (w / np.float32(input_scale)).astype(np.float32)
        sd[name + ".weight"] = w
        sd[name + ".bias"] = uniform(seed, name + ".bias", (n_out,), -bound, bound)
        if bn is not None:
            gamma = uniform(seed, bn + ".weight", (n_out,), 0.6, 1.4)
            flip = uniform(seed, bn + ".flip", (n_out,)) < 0.1
            sd[bn + ".weight"] = np.where(flip, -gamma, gamma).astype(np.float32)
            sd[bn + ".bias"] = uniform(seed, bn + ".bias", (n_out,), -0.2, 0.2)
            sd[bn + ".running_mean"] = uniform(seed, bn + ".running_mean", (n_out,), -0.2, 0.2)
            sd[bn + ".running_var"] = uniform(seed, bn + ".running_var", (n_out,), 0.6, 1.6)
            sd[bn + ".num_batches_tracked"] = np.array(0, dtype=np.int64)
    return sd


def torch_state_dict(seed: int, k: int = 2, input_scale: float = 1.0, gain: float = 2.0) -> dict:
    import torch

    sd = make_state_dict(seed, k, input_scale, gain)
    return {n: torch.from_numpy(np.ascontiguousarray(v)) for n, v in sd.items()}


GRIPPER_BOX = (0.06, 0.08, 0.02)  # hand_depth, open width, hand_height in metres (synthetic default)


def candidate_points(seed: int, B: int, P: int, kind: str = "gripper", first: int = 0) -> np.ndarray:
    """Points of candidates [first, first+B) as float32 [B,3,P].

    kind="gripper": each candidate fills a random sub-box of the open-gripper region in the
    gripper frame (x in (0,hd), y in (-ow/2,ow/2), z in (-hh/2,hh/2)), so candidates differ in
    shape, not only in noise; kind="normal": N(mu_i, sigma_i^2) with per-candidate mu, sigma.
    Candidate i depends only on (seed, i), so any sharding of the batch sees the same data.
    """
    ids = np.arange(first, first + B, dtype=np.uint64)
    base = splitmix64(np.array([_stream_base(seed, "cand/" + kind)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        cbase = splitmix64((ids * np.uint64(0x2545F4914F6CDD1D) + base) & _MASK)      # [B]
        ctr = (cbase[:, None] + np.arange(3 * P + 8, dtype=np.uint64)[None, :]) & _MASK
    u = (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    par, u = u[:, :8], u[:, 8:].reshape(B, 3, P)
    if kind == "gripper":
        size = np.array(GRIPPER_BOX)[None, :]
        lo_box = np.array([0.0, -GRIPPER_BOX[1] / 2, -GRIPPER_BOX[2] / 2])[None, :]
        half = (0.15 + 0.35 * par[:, 0:3]) * size / 2                                    # [B,3]
        centre = lo_box + half + par[:, 3:6] * (size - 2 * half)
        pts = centre[:, :, None] + (2 * u - 1) * half[:, :, None]
    elif kind == "normal":
        u2 = (splitmix64((ctr[:, 8:] ^ np.uint64(0xA5A5A5A5A5A5A5A5))) >> np.uint64(11)).astype(np.float64)
        u2 = (u2 * (1.0 / 9007199254740992.0)).reshape(B, 3, P)
        z = np.sqrt(-2.0 * np.log1p(-u)) * np.cos(2.0 * np.pi * u2)
        mu = 2.0 * par[:, 0:3] - 1.0
        sigma = 0.3 + 1.2 * par[:, 3:6]
        pts = mu[:, :, None] + sigma[:, :, None] * z
    else:
        raise ValueError(f"unknown kind {kind!r}")
    return pts.astype(np.float32)


def scene_points(seed: int, S: int, cube: float = 0.5) -> np.ndarray:
    """Scene cloud float32 [S,3], uniform in [0,cube)^3 (the voxel-centre cloud of the online loop
    is float32 too, apps/kinect2grasp.py:159)."""
    return uniform(seed, "scene", (S, 3), 0.0, cube)


def elongated_scene_points(seed: int, S: int, cube: float = 0.5) -> np.ndarray:
    """scene_points squeezed to a flat slab away from the origin: a bounding box far from cubic, which the
    voxel grid of the reference grows to a cube around itself (apps/voxelgrid.py:119-123)."""
    pts = scene_points(seed, S, cube).copy()
    pts[:, 1] *= np.float32(0.37)
    pts[:, 2] = pts[:, 2] * np.float32(0.11) + np.float32(0.4)
    return pts


def random_grasps(seed: int, G: int, cube: float = 0.5, margin: float = 0.06) -> np.ndarray:
    """Grasp array float64 [G,5,3] in the layout collect_pc reads (apps/kinect2grasp.py:389-396):
    row 4 = bottom centre, rows 1,2,3 = approach, binormal, minor axis.  Axes form a random
    right-handed orthogonal frame, each scaled by a random length so that normalisation matters;
    row 0 (unused by the crop) holds the centre again."""
    q = normal(seed, "grasp/quat", (G, 4), dtype=np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)   # [G,3,3]
    lengths = uniform(seed, "grasp/len", (G, 3), 0.5, 2.0, dtype=np.float64)
    centre = uniform(seed, "grasp/centre", (G, 3), margin, cube - margin, dtype=np.float64)
    g = np.zeros((G, 5, 3), dtype=np.float64)
    g[:, 0] = centre
    g[:, 4] = centre
    for r in range(3):
        g[:, 1 + r] = R[:, :, r] * lengths[:, r:r + 1]     # columns of R are the frame axes
    return g


def dataset_view(seed: int, S: int, G: int, cube: float = 0.3):
    """One synthetic (object, view) of the training-data crop: grasp records float64 [G,10]
    (contact centre, closing axis (not unit), width, angle, level score, refine score; the layout
    collect_pc_ and __getitem__ read, model/dataset.py:776-779, 1100), a rigid 4x4 transform, and the
    view's cloud float32 [S,3], uniform in a cube around the transformed object."""
    c = uniform(seed, "ds.center", (G, 3), 0.08, cube - 0.08).astype(np.float64)
    axis = uniform(seed, "ds.axis", (G, 3), -1.0, 1.0).astype(np.float64) * 3.0
    axis[0] = [0.0, 0.0, 2.0]                      # the degenerate branch of :798 (axis along z)
    width = uniform(seed, "ds.width", (G,), 0.02, 0.08).astype(np.float64)
    angle = uniform(seed, "ds.angle", (G,), -np.pi, np.pi).astype(np.float64)
    level = np.floor(uniform(seed, "ds.level", (G,), 0.0, 5.0)).astype(np.float64) * 0.2
    refine = uniform(seed, "ds.refine", (G,), 0.0, 1.0).astype(np.float64)
    grasps = np.concatenate([c, axis, width[:, None], angle[:, None], level[:, None], refine[:, None]], axis=1)
    q = uniform(seed, "ds.quat", (4,), -1.0, 1.0).astype(np.float64)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = uniform(seed, "ds.trans", (3,), -0.5, 0.5).astype(np.float64)
    local = uniform(seed, "ds.cloud", (S, 3), 0.0, cube).astype(np.float64)
    pc = (local @ R.T + T[:3, 3]).astype(np.float32)
    return grasps, T, pc


def balance_head(model, points) -> float:
    """Shift the last classifier bias of a PointNetCls so that about half of `points` ([B,P,3] on the model's device)
    land in the highest class.  Seeded synthetic weights put nearly every candidate into one class, which would leave
    the vote, the good-grasp filter and the ranking of the online loop with nothing to do.  Returns the shift."""
    import torch
    was_training = model.training
    model.eval()
    with torch.no_grad():
        logp = model.forward_points(points)[0]
        k = logp.shape[1]
        rest = logp[:, :k - 1].max(dim=1).values
        shift = float((rest - logp[:, k - 1]).median())
        model.fc3.bias[k - 1] += shift             # in place through autograd's counter: the packed weights are rebuilt
    model.train(was_training)
    return shift
