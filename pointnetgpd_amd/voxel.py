"""Device-side voxel down-sampling of the scene, the step in front of the crop.

Mirrors `get_voxel_fun(points_, n)` (apps/kinect2grasp.py:145-150), i.e.
`VoxelGrid(points, n_x=n, n_y=n, n_z=n).compute()` (apps/voxelgrid.py:115-160) followed by
`np.unique(voxel_centers[voxel_n], axis=0)`, and the retry loop around it in `cal_grasp`
(apps/kinect2grasp.py:165-179).  The type of the bin edges is numpy's choice (the reference does not pin
numpy): `np.linspace` on float32 scalars yields float64 edges under the numpy 1.x promotion rules the
reference was written for (`edges="float64"`) and float32 edges under numpy >= 2 (`edges="float32"`); points within
one float32 ulp of an edge, and the last bit of a centre, differ.  The default is what the numpy installed beside this
package would make the reference compute (`default_edges()`).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

N_VOXEL = 500             # apps/kinect2grasp.py: n_voxel
MAX_N = 1600


EDGES = {"float64": _lib.EDGES_F64, "float32": _lib.EDGES_F32}


def default_edges() -> str:
    """The edge type the reference's np.linspace call produces under the numpy installed here: float32 from numpy 2
    on (the form pinned bit for bit by a run of the reference's own code), float64 before."""
    import numpy
    return "float32" if int(numpy.__version__.split(".")[0]) >= 2 else "float64"


_WORKSPACES = {}        # (device index, n) -> workspace tensor, so that a frame loop allocates nothing


def voxel_unique_device(points: torch.Tensor, n: int, stream=None, edges: str = None, out: torch.Tensor = None,
                        count: torch.Tensor = None):
    """get_voxel_fun without the host in the loop: returns (out [capacity,3] float32, count [1] int32), both on the
    device; rows [0, count) of `out` are the unique voxel centres in lexicographic order, the rest is not written.
    Nothing is read back, so a caller can queue the crop behind it (the crop's kernels take the point count from
    memory) or read `count` when it needs the number.  `out` / `count` may be passed in to reuse buffers; capacity
    = out.shape[0], default min(S, n^3), which always suffices."""
    edges = default_edges() if edges is None else edges
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
        ws = _WORKSPACES.get((dev.index, int(n)))
        if ws is None:
            if len(_WORKSPACES) > 8:
                _WORKSPACES.clear()
            ws = _WORKSPACES[(dev.index, int(n))] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if out is None:
            out = torch.empty((min(S, int(n) ** 3), 3), dtype=torch.float32, device=dev)
        if count is None:
            count = torch.empty(1, dtype=torch.int32, device=dev)
        if out.dtype != torch.float32 or out.dim() != 2 or out.shape[1] != 3 or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous [capacity,3] float32 tensor on the points' device")
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(lib.pngpd_voxel_unique(ctypes.c_void_p(pts.data_ptr()), S, int(n), EDGES[edges],
                                          ctypes.c_void_p(ws.data_ptr()),
                                          ws_bytes, ctypes.c_void_p(out.data_ptr()), out.shape[0],
                                          ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(st)))
    return out, count


def get_voxel_fun(points: torch.Tensor, n: int, stream=None, edges: str = None) -> torch.Tensor:
    """points [S,3] float32 on the GPU -> unique voxel centres [M,3] float32, lexicographic order (one host read of
    the count, to give the result its exact shape as the reference does)."""
    out, count = voxel_unique_device(points, n, stream, edges)
    m = int(count.item())
    assert m <= out.shape[0]
    return out[:m]


def voxelize_scene(points: torch.Tensor, n: int = N_VOXEL, edges: str = None) -> torch.Tensor:
    """The scene pre-processing of cal_grasp (apps/kinect2grasp.py:165-179): one pass at `n`; for
    clouds under 2000 points, finer grids (n += 100) until all but 15 points keep a voxel of their own."""
    voxels = get_voxel_fun(points, n, edges=edges)
    if points.shape[0] < 2000:
        while voxels.shape[0] < points.shape[0] - 15 and n <= MAX_N:
            voxels = get_voxel_fun(points, n, edges=edges)
            n = n + 100
    return voxels
