"""Device-side voxel down-sampling of the scene, the step in front of the crop.

Mirrors `get_voxel_fun(points_, n)` (apps/kinect2grasp.py:145-150), i.e.
`VoxelGrid(points, n_x=n, n_y=n, n_z=n).compute()` (apps/voxelgrid.py:115-160) followed by
`np.unique(voxel_centers[voxel_n], axis=0)`, and the retry loop around it in `cal_grasp`
(apps/kinect2grasp.py:165-179).  The type of the bin edges is numpy's choice (the reference does not pin
numpy): `np.linspace` on float32 scalars yields float64 edges under the numpy 1.x promotion rules the
reference was written for (`edges="float64"`, the default) and float32 edges under numpy >= 2
(`edges="float32"`); points within one float32 ulp of an edge, and the last bit of a centre, differ.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

N_VOXEL = 500             # apps/kinect2grasp.py: n_voxel
MAX_N = 1600


EDGES = {"float64": _lib.EDGES_F64, "float32": _lib.EDGES_F32}


def get_voxel_fun(points: torch.Tensor, n: int, stream=None, edges: str = "float64") -> torch.Tensor:
    """points [S,3] float32 on the GPU -> unique voxel centres [M,3] float32, lexicographic order."""
    if edges not in EDGES:
        raise ValueError(f"edges must be 'float64' (numpy 1.x) or 'float32' (numpy >= 2), got {edges!r}")
    if not points.is_cuda:
        raise RuntimeError("pointnetgpd_amd.voxel runs on the GPU only; got a CPU tensor")
    if points.dim() != 2 or points.shape[1] != 3 or points.shape[0] < 1:
        raise ValueError(f"points must be [S,3] with S >= 1, got {tuple(points.shape)}")
    if not 1 <= int(n) <= MAX_N:
        raise ValueError(f"n must be in [1, {MAX_N}], got {n}")
    pts = points.to(torch.float32).contiguous()      # cal_grasp: points_.astype(np.float32)
    S = pts.shape[0]
    lib = _lib.load()
    dev = pts.device
    with torch.cuda.device(dev):
        ws_bytes = lib.pngpd_voxel_workspace_bytes(int(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        cap = min(S, int(n) ** 3)
        out = torch.empty((cap, 3), dtype=torch.float32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(lib.pngpd_voxel_unique(ctypes.c_void_p(pts.data_ptr()), S, int(n), EDGES[edges],
                                          ctypes.c_void_p(ws.data_ptr()),
                                          ws_bytes, ctypes.c_void_p(out.data_ptr()), cap,
                                          ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(st)))
        m = int(count.item())
    assert m <= cap
    return out[:m]


def voxelize_scene(points: torch.Tensor, n: int = N_VOXEL, edges: str = "float64") -> torch.Tensor:
    """The scene pre-processing of cal_grasp (apps/kinect2grasp.py:165-179): one pass at `n`; for
    clouds under 2000 points, finer grids (n += 100) until all but 15 points keep a voxel of their own."""
    voxels = get_voxel_fun(points, n, edges=edges)
    if points.shape[0] < 2000:
        while voxels.shape[0] < points.shape[0] - 15 and n <= MAX_N:
            voxels = get_voxel_fun(points, n, edges=edges)
            n = n + 100
    return voxels
