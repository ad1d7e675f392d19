"""CPU restatement of the scene voxel down-sampling.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path.

Follows VoxelGrid.compute (apps/voxelgrid.py:115-160) and get_voxel_fun
(apps/kinect2grasp.py:145-150) line by line with numpy.

Pinning.  apps/voxelgrid.py imports numba at module level (absent from this image), but `cartesian`, the
`VoxelGrid` class and `get_voxel_fun` need only numpy: tests/golden/make_voxel_golden.py (build container only)
takes those three definitions out of the reference files' syntax trees, runs them and stores their outputs as
tests/golden/voxel_reference.npz.  That run uses this image's numpy 2.2, under which np.linspace on two float32
scalars stays in float32: it pins `edges="float32"` below.  Under the numpy 1.x releases the reference was
written for the same call promotes to float64 (0-d float32 * 1.0 inside linspace): `edges="float64"` forces
that by calling np.linspace on Python floats; it differs from the pinned form in that one call only and has no
reference-run fixture of its own (no numpy 1.x here).
"""
import numpy as np


def cartesian(arrays):
    """apps/voxelgrid.py:40-86."""
    arrays = [np.asarray(x) for x in arrays]
    ix = np.indices([len(x) for x in arrays]).reshape(len(arrays), -1).T
    out = np.empty_like(ix, dtype=arrays[0].dtype)
    for k, arr in enumerate(arrays):
        out[:, k] = arr[ix[:, k]]
    return out


def voxel_grid(points, n, edges="float64"):
    """apps/voxelgrid.py:115-160 with n_x = n_y = n_z = n, sizes None, regular_bounding_box True.
    Returns (voxel_n [S] int64, segments (3 x [n+1]), mid-points (3 x [n])), both of type `edges`."""
    points = np.asarray(points)
    assert points.dtype == np.float32
    xyzmin = points.min(0)
    xyzmax = points.max(0)
    margin = max(xyzmax - xyzmin) - (xyzmax - xyzmin)          # :121
    xyzmin = xyzmin - margin / 2                               # :122
    xyzmax = xyzmax + margin / 2                               # :123
    assert xyzmin.dtype == np.float32
    if edges == "float64":
        segments = [np.linspace(float(xyzmin[i]), float(xyzmax[i]), num=n + 1) for i in range(3)]   # :140, numpy 1.x
    else:
        assert int(np.__version__.split(".")[0]) >= 2, "float32 edges are what numpy >= 2 computes"
        segments = [np.linspace(xyzmin[i], xyzmax[i], num=n + 1) for i in range(3)]                 # :140, numpy >= 2
        assert segments[0].dtype == np.float32
    vox = [np.clip(np.searchsorted(segments[i], points[:, i]) - 1, 0, n) for i in range(3)]   # :153-155
    voxel_n = np.ravel_multi_index(vox, [n, n, n])             # :156
    mids = [(segments[i][1:] + segments[i][:-1]) / 2 for i in range(3)]                        # :159
    return voxel_n, segments, mids


def get_voxel_fun(points, n, edges="float64"):
    """apps/kinect2grasp.py:145-150.  The [n^3,3] centre table of the reference is not built; the
    rows it would be indexed at are formed directly (same values: cartesian() is an outer indexing)."""
    voxel_n, _, mids = voxel_grid(points, n, edges)
    ix, iy, iz = np.unravel_index(voxel_n, (n, n, n))
    rows = np.stack([mids[0][ix], mids[1][iy], mids[2][iz]], axis=1).astype(np.float32)          # :160, :148
    return np.unique(rows, axis=0)                             # :149


def get_voxel_fun_full_table(points, n, edges="float64"):
    """The literal form (materialises voxel_centers); only for small n, to check the shortcut above."""
    voxel_n, _, mids = voxel_grid(points, n, edges)
    centers = cartesian(mids).astype(np.float32)
    return np.unique(centers[voxel_n], axis=0)
