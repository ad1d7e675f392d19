"""Device-side counterpart of the training-data crop of PointGraspOneViewDataset.

`collect_pc_(grasp, pc, transform, ...)` (model/dataset.py:767-924) builds a gripper frame from an
8-float grasp record (contact centre, closing axis, width, angle) and the 4x4 transform of the
view, crops the view's cloud with the same six planes as the online crop, and `__getitem__`
(:1098-1131) resamples the crop to N points and labels it from the grasp's score.  The reference
does this per sample in 32 numpy worker processes (main_1v.py:158).

Here everything runs on the GPU: `pngpd_dataset_frames` builds the frames of all grasp records of one
(object, view), `pngpd_crop_count_frames` / `pngpd_crop_emit` crop the cloud by every frame and
`pngpd_crop_gather_dataset` resamples.  The six box limits are explicit for the same reason as in
pointnetgpd_amd.crop (the corner table comes from dex-net).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .crop import CropResult, _ptr

MIN_POINT_LIMIT = 50      # model/dataset.py: self.min_point_limit


def frames_from_configs(grasps, transform, hand_depth, device=None) -> torch.Tensor:
    """Frames float64 [G,12] on the device (origin `bottom_`, then the rows approach_, binormal_, minor_normal_ of
    `matrix`; model/dataset.py:776-835) for grasp records [G, >=8] (centre, axis, width, angle, ...)."""
    lib = _lib.load()
    if isinstance(grasps, torch.Tensor) and grasps.is_cuda:
        dev = grasps.device
    else:
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    rec = torch.as_tensor(grasps, dtype=torch.float64, device=dev)
    if rec.dim() == 1:
        rec = rec[None]
    if rec.dim() != 2 or rec.shape[1] < 8 or rec.shape[0] < 1:
        raise ValueError(f"grasp records must be [G, >=8], got {tuple(rec.shape)}")
    rec = rec.contiguous()
    T = np.ascontiguousarray(np.asarray(transform, dtype=np.float64))
    if T.shape != (4, 4):
        raise ValueError(f"transform must be 4x4, got {T.shape}")
    G = rec.shape[0]
    frames = torch.empty((G, 12), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_dataset_frames(_ptr(rec), G, rec.shape[1], T.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                            float(hand_depth), _ptr(frames), stream))
    return frames


def labels_from_scores(grasps, thresh_good, thresh_bad) -> np.ndarray:
    """model/dataset.py:1100, 1117-1124: score = level + 0.01 * refine (the last two fields);
    0 = bad (score >= thresh_bad), 1 = good (score <= thresh_good), -1 = dropped (in between)."""
    grasps = np.asarray(grasps, dtype=np.float64)
    score = grasps[:, -2] + grasps[:, -1] * 0.01
    return np.where(score >= thresh_bad, 0, np.where(score <= thresh_good, 1, -1)).astype(np.int64)


def collect_pc_frames(frames, pc, limits, return_points=True) -> CropResult:
    """Crop the view cloud `pc` (float32 [S,3]) with frames [G,12] as they are (no re-normalisation)."""
    lib = _lib.load()
    if isinstance(pc, torch.Tensor) and not pc.is_cuda:
        raise RuntimeError("pointnetgpd_amd.dataset_crop runs on the GPU only; got a CPU tensor")
    dev = pc.device if isinstance(pc, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
    scene = torch.as_tensor(pc, dtype=torch.float32, device=dev).contiguous()
    if isinstance(frames, torch.Tensor):
        fr = frames.to(device=dev, dtype=torch.float64).reshape(-1, 12).contiguous()
    else:
        fr = torch.as_tensor(np.asarray(frames, dtype=np.float64).reshape(-1, 12), device=dev).contiguous()
    if scene.dim() != 2 or scene.shape[1] != 3:
        raise ValueError("pc must be [S,3]")
    S, G = scene.shape[0], fr.shape[0]
    lim = (ctypes.c_double * 6)(*[float(v) for v in limits])
    nbytes = lib.pngpd_crop_workspace_bytes(S, G)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(G, dtype=torch.int32, device=dev)
    offsets = torch.empty(G + 1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_crop_count_frames(_ptr(scene), S, _ptr(fr), G, lim, _ptr(ws), nbytes, _ptr(counts),
                                               _ptr(offsets), stream))
        total = int(offsets[G].item())
        indices = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        points_g = torch.empty((max(total, 1), 3), dtype=torch.float64, device=dev) if return_points else None
        _lib.check(lib.pngpd_crop_emit(_ptr(scene), S, G, lim, _ptr(ws), nbytes, _ptr(counts), _ptr(offsets),
                                       _ptr(indices), total, _ptr(points_g), stream))
    return CropResult(counts, offsets, indices[:total], points_g[:total] if return_points else None, scene, ws, G)


def resample(crop: CropResult, num_points: int, sel=None, seed: int = 0, min_point_limit: int = MIN_POINT_LIMIT,
             return_selection: bool = False, grasp_offset: int = 0):
    """float32 [G,3,N] (the layout PointNetCls takes) and valid uint8 [G] (0 = the dataset returns None).
    `sel` int32 [G,N] replays given np.random.choice results; otherwise the device draws: without
    replacement only when cnt > N (model/dataset.py:1104)."""
    lib = _lib.load()
    dev = crop.scene.device
    G, S = crop.n_grasps, crop.scene.shape[0]
    pts = torch.empty((G, 3, num_points), dtype=torch.float32, device=dev)
    valid = torch.empty(G, dtype=torch.uint8, device=dev)
    sel_t = None
    if sel is not None:
        sel_t = torch.as_tensor(sel, dtype=torch.int32, device=dev).contiguous()
        if sel_t.shape != (G, num_points):
            raise ValueError(f"sel must be [{G},{num_points}]")
    sel_out = torch.empty((G, num_points), dtype=torch.int32, device=dev) if return_selection else None
    idx = crop.indices if crop.indices.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pngpd_crop_gather_dataset(_ptr(crop.scene), S, G, _ptr(crop.workspace), crop.workspace.numel(),
                                                 _ptr(idx), _ptr(crop.offsets), _ptr(crop.counts), num_points,
                                                 min_point_limit, _ptr(sel_t), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                                 grasp_offset, _ptr(pts), _ptr(sel_out), _ptr(valid), stream))
    return (pts, valid, sel_out) if return_selection else (pts, valid)


@dataclass
class TrainingBatch:
    data: torch.Tensor      # float32 [B,3,N] on the device
    label: torch.Tensor     # int64 [B] on the device
    kept: np.ndarray        # indices into the grasp records of the samples that survived


def make_training_batch(grasps, pc, transform, limits, hand_depth, num_points, thresh_good, thresh_bad,
                        seed: int = 0, min_point_limit: int = MIN_POINT_LIMIT) -> TrainingBatch:
    """All samples one (object, view) pair yields: what __getitem__ returns for each grasp record,
    stacked; records whose crop is too small or whose score lies between the thresholds are dropped
    (the reference's collate function drops the None items, main_1v.py:127-129)."""
    dev = pc.device if isinstance(pc, torch.Tensor) else None
    frames = frames_from_configs(grasps, transform, hand_depth, device=dev)
    labels = labels_from_scores(grasps, thresh_good, thresh_bad)
    crop = collect_pc_frames(frames, pc, limits, return_points=False)
    data, valid = resample(crop, num_points, seed=seed, min_point_limit=min_point_limit)
    keep = valid.bool() & torch.from_numpy(labels >= 0).to(valid.device)
    kept = torch.nonzero(keep).flatten()
    return TrainingBatch(data[kept].contiguous(), torch.from_numpy(labels).to(valid.device)[kept], kept.cpu().numpy())
