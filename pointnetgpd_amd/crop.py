"""Device-side counterpart of the reference's online crop loop.

`collect_pc(grasp_, pc)` in the reference (apps/kinect2grasp.py:377-438) loops over the grasps in
Python and, for each, runs check_collision_square (:299-344) over the whole scene with numpy.
Here the same float64 arithmetic runs once on the GPU for all grasps; the index sets come back
bit-identical and in ascending order.  The six box limits are explicit because the gripper corner
table comes from an external dex-net call (ags.get_hand_points, :401) in the reference.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

MIN_POINTS = 20   # minimal_points_send_to_point_net (apps/kinect2grasp.py:746)


def box_limits_from_gripper(hand_depth, hand_outer_diameter, finger_width, hand_height):
    """(ylo, yhi, zlo, zhi, xlo, xhi) of the open gripper region in the gripper frame: the values
    corners 1, 2, 4, 8 of dex-net's hand model take for a gripper at the origin with approach +x
    and binormal +y (apps/kinect2grasp.py:323, 401)."""
    ow = float(hand_outer_diameter) - 2.0 * float(finger_width)
    hh, hd = float(hand_height), float(hand_depth)
    return np.array([-ow * 0.5, ow * 0.5, -hh * 0.5, hh * 0.5, 0.0, hd], dtype=np.float64)


def box_limits_from_hand_points(p):
    """Same six limits read from a [21,3] corner table `p` (rows 1, 2, 4, 8; :323, :332-337)."""
    p = np.asarray(p, dtype=np.float64)
    return np.array([p[1][1], p[2][1], p[4][2], p[1][2], p[8][0], p[4][0]], dtype=np.float64)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@dataclass
class CropResult:
    """All tensors live on the device."""
    counts: torch.Tensor        # int32 [G]
    offsets: torch.Tensor       # int64 [G+1]
    indices: torch.Tensor       # int32 [total]  ascending per grasp
    points_g: torch.Tensor      # float64 [total,3] or None
    scene: torch.Tensor
    workspace: torch.Tensor
    n_grasps: int

    def index_lists(self):
        """Per-grasp int64 index arrays on the host, like collect_pc's in_ind_."""
        off = self.offsets.cpu().numpy()
        idx = self.indices.cpu().numpy().astype(np.int64)
        return [idx[off[g]:off[g + 1]] for g in range(self.n_grasps)]

    def point_lists(self):
        """Per-grasp float64 [cnt,3] gripper-frame points on the host, like in_ind_points_."""
        off = self.offsets.cpu().numpy()
        pts = self.points_g.cpu().numpy()
        return [pts[off[g]:off[g + 1]] for g in range(self.n_grasps)]


def _device_inputs(grasp_, pc):
    """(scene float32 [S,3], grasps float64 [G,5,3]) on the scene's device (the current ROCm device for host data);
    grasps may come as anything that reshapes to [G,5,3] (apps/kinect2grasp.py:385-389)."""
    dev = pc.device if isinstance(pc, torch.Tensor) and pc.is_cuda else torch.device("cuda", torch.cuda.current_device())
    scene = torch.as_tensor(pc, dtype=torch.float32, device=dev).contiguous()
    if scene.dim() != 2 or scene.shape[1] != 3:
        raise ValueError("scene must be [S,3]")
    if isinstance(grasp_, torch.Tensor) and grasp_.is_cuda:
        grasps = grasp_.to(device=dev, dtype=torch.float64).reshape(-1, 5, 3).contiguous()
    else:
        host = np.asarray(grasp_.cpu() if isinstance(grasp_, torch.Tensor) else grasp_, dtype=np.float64)
        grasps = torch.as_tensor(host.reshape(-1, 5, 3), device=dev).contiguous()
    return scene, grasps, dev


def collect_pc(grasp_, pc, limits, return_points=True) -> CropResult:
    """Crop the scene `pc` (float32 [S,3]) with every grasp of `grasp_` ([G,5,3] float64 or
    anything that reshapes to it, as in apps/kinect2grasp.py:385-389)."""
    lib = _lib.load()
    scene, grasps, dev = _device_inputs(grasp_, pc)
    S, G = scene.shape[0], grasps.shape[0]
    lim = (ctypes.c_double * 6)(*[float(v) for v in limits])
    nbytes = lib.pngpd_crop_workspace_bytes(S, G)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(G, dtype=torch.int32, device=dev)
    offsets = torch.empty(G + 1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_crop_count(_ptr(scene), S, _ptr(grasps), G, lim, _ptr(ws), nbytes, _ptr(counts),
                                        _ptr(offsets), stream))
        total = int(offsets[G].item())          # the one host sync: the output size is data dependent
        indices = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        points_g = torch.empty((max(total, 1), 3), dtype=torch.float64, device=dev) if return_points else None
        _lib.check(lib.pngpd_crop_emit(_ptr(scene), S, G, lim, _ptr(ws), nbytes, _ptr(counts), _ptr(offsets),
                                       _ptr(indices), total, _ptr(points_g), stream))
    return CropResult(counts, offsets, indices[:total], points_g[:total] if return_points else None, scene, ws, G)


def subsample(crop: CropResult, num_points: int, repeat: int = 3, sel=None, seed: int = 0,
              min_points: int = MIN_POINTS, return_selection: bool = False, grasp_offset: int = 0):
    """Network input for every (grasp, repeat): float32 [G*repeat, P, 3] in the gripper frame, plus
    valid uint8 [G] (0 where the grasp has fewer than `min_points` points, :746-751).

    `sel` int32 [G*repeat, P] reproduces given np.random.choice results (:768-774); without it the
    library draws on the device (without replacement when cnt >= P, with replacement otherwise).
    """
    lib = _lib.load()
    dev = crop.scene.device
    G, S = crop.n_grasps, crop.scene.shape[0]
    pts = torch.empty((G * repeat, num_points, 3), dtype=torch.float32, device=dev)
    valid = torch.empty(G, dtype=torch.uint8, device=dev)
    sel_t = None
    if sel is not None:
        sel_t = torch.as_tensor(sel, dtype=torch.int32, device=dev).contiguous()
        if sel_t.shape != (G * repeat, num_points):
            raise ValueError(f"sel must be [{G * repeat},{num_points}]")
    sel_out = torch.empty((G * repeat, num_points), dtype=torch.int32, device=dev) if return_selection else None
    idx = crop.indices if crop.indices.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_crop_gather(_ptr(crop.scene), S, G, _ptr(crop.workspace), crop.workspace.numel(),
                                         _ptr(idx), _ptr(crop.offsets), _ptr(crop.counts), repeat, num_points,
                                         min_points, _ptr(sel_t), ctypes.c_uint64(seed & (2 ** 64 - 1)), grasp_offset,
                                         _ptr(pts),
                                         _ptr(sel_out), _ptr(valid), stream))
    return (pts, valid, sel_out) if return_selection else (pts, valid)


def crop_subsample(grasp_, pc, limits, num_points: int, repeat: int = 3, seed: int = 0,
                   min_points: int = MIN_POINTS, grasp_offset: int = 0):
    """collect_pc + subsample in one device pass, without the index lists and without a host read in between
    (apps/kinect2grasp.py:712, 746-774 for all grasps): returns (pts float32 [G*repeat, P, 3], valid uint8 [G],
    counts int32 [G], overflow int32 [1]).  The draws are the device generator's, so the result equals
    subsample(collect_pc(...), seed=seed).  overflow[0] != 0 means some grasp held more points than the kernel
    keeps on chip and its rows were not written: call collect_pc + subsample then (score_grasps does)."""
    lib = _lib.load()
    scene, grasps, dev = _device_inputs(grasp_, pc)
    S, G = scene.shape[0], grasps.shape[0]
    lim = (ctypes.c_double * 6)(*[float(v) for v in limits])
    nbytes = lib.pngpd_crop_workspace_bytes(S, G)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pts = torch.empty((G * repeat, num_points, 3), dtype=torch.float32, device=dev)
    valid = torch.empty(G, dtype=torch.uint8, device=dev)
    counts = torch.empty(G, dtype=torch.int32, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_crop_subsample(_ptr(scene), S, _ptr(grasps), G, lim, _ptr(ws), nbytes, repeat, num_points,
                                            min_points, ctypes.c_uint64(seed & (2 ** 64 - 1)), grasp_offset, _ptr(pts),
                                            _ptr(valid), _ptr(counts), _ptr(overflow), stream))
    return pts, valid, counts, overflow


def vote(pred: torch.Tensor, probs: torch.Tensor, valid, repeat: int):
    """Repeat vote of apps/kinect2grasp.py:799-817 for all grasps: (label int32 [G], score float32 [G])."""
    lib = _lib.load()
    dev = pred.device
    k = probs.shape[1]
    G = pred.numel() // repeat
    pred = pred.to(torch.int64).contiguous()
    probs = probs.to(torch.float32).contiguous()
    label = torch.empty(G, dtype=torch.int32, device=dev)
    score = torch.empty(G, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_vote(_ptr(pred), _ptr(probs), _ptr(valid), G, repeat, k, _ptr(label), _ptr(score), stream))
    return label, score


# ---- host twin of the library's selection generator (csrc/crop.hip: draw_selection) ------------
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = (np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def _mix32(x):
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=np.uint32)
        x = x ^ (x >> np.uint32(15))
        x = x * np.uint32(0x2C1B3C6D)
        x = x ^ (x >> np.uint32(12))
        x = x * np.uint32(0x297A2D39)
        return x ^ (x >> np.uint32(15))


def _feistel_f(half, key):
    with np.errstate(over="ignore"):
        return ((np.asarray(half, dtype=np.uint32) ^ np.uint32(key)) * np.uint32(0x9E3779B1)) >> np.uint32(16)


def draw_selection(seed: int, g: int, r: int, n: int, P: int, replace_when_equal: bool = False) -> np.ndarray:
    """The P positions the device draws for (grasp g, repeat r) when the grasp holds n points (csrc/crop_select.h).
    n >= P: the first P numbers below n of pi(0), pi(1), ... for a keyed 4-round Feistel permutation pi of
    [0, 2^ceil(log2 n)); n < P: independent uniform draws.
    replace_when_equal: the dataset's rule (n == P drawn with replacement, model/dataset.py:1104)."""
    with np.errstate(over="ignore"):
        inner = _mix64((np.uint64(g) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(r)) & _M)
        key = _mix64(np.uint64(seed & (2 ** 64 - 1)) ^ inner)
        rk = [np.uint32(int(_mix64((key + (np.uint64(t + 1) << np.uint64(32))) & _M)) >> 32) for t in range(4)]
        if n < P + int(replace_when_equal):
            i = np.arange(P, dtype=np.uint32)
            u = _mix32(i * np.uint32(0x9E3779B9) + rk[0]).astype(np.uint64)
            return ((u * np.uint64(n)) >> np.uint64(32)).astype(np.int32)
        b = 2
        while (1 << b) < n:
            b += 1
        a = b >> 1
        a_mask, c_mask = np.uint32((1 << a) - 1), np.uint32((1 << (b - a)) - 1)
        t = np.arange(1 << b, dtype=np.uint32)
        hi, lo = t >> np.uint32(a), t & a_mask
        hi = hi ^ (_feistel_f(lo, rk[0]) & c_mask)
        lo = lo ^ (_feistel_f(hi, rk[1]) & a_mask)
        hi = hi ^ (_feistel_f(lo, rk[2]) & c_mask)
        lo = lo ^ (_feistel_f(hi, rk[3]) & a_mask)
        x = (hi << np.uint32(a)) | lo
        return x[x < np.uint32(n)][:P].astype(np.int32)
