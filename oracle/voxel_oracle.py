"""CPU restatement of the scene voxel down-sampling.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path.

Follows VoxelGrid.compute (apps/voxelgrid.py:115-160) and get_voxel_fun
(apps/kinect2grasp.py:145-150) line by line with numpy.  PARITY UNPINNED: apps/voxelgrid.py imports
numba at module level, which this image does not have, and the reference holds no fixture for it;
the restatement is checked against hand-computed cases in tests/test_voxel_oracle.py.

One deliberate pin: np.linspace is called on Python floats so that the bin edges are float64 under
any numpy version.  That is what the reference computed with the numpy 1.x releases it was written
against (0-d float32 * 1.0 promoted to float64 inside linspace); numpy 2 keeps float32 there.
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


def voxel_grid(points, n):
    """apps/voxelgrid.py:115-160 with n_x = n_y = n_z = n, sizes None, regular_bounding_box True.
    Returns (voxel_n [S] int64, segments (3 x [n+1] float64), mid-points (3 x [n] float64))."""
    points = np.asarray(points)
    assert points.dtype == np.float32
    xyzmin = points.min(0)
    xyzmax = points.max(0)
    margin = max(xyzmax - xyzmin) - (xyzmax - xyzmin)          # :121
    xyzmin = xyzmin - margin / 2                               # :122
    xyzmax = xyzmax + margin / 2                               # :123
    assert xyzmin.dtype == np.float32
    segments = [np.linspace(float(xyzmin[i]), float(xyzmax[i]), num=n + 1) for i in range(3)]   # :140
    vox = [np.clip(np.searchsorted(segments[i], points[:, i]) - 1, 0, n) for i in range(3)]   # :153-155
    voxel_n = np.ravel_multi_index(vox, [n, n, n])             # :156
    mids = [(segments[i][1:] + segments[i][:-1]) / 2 for i in range(3)]                        # :159
    return voxel_n, segments, mids


def get_voxel_fun(points, n):
    """apps/kinect2grasp.py:145-150.  The [n^3,3] centre table of the reference is not built; the
    rows it would be indexed at are formed directly (same values: cartesian() is an outer indexing)."""
    voxel_n, _, mids = voxel_grid(points, n)
    ix, iy, iz = np.unravel_index(voxel_n, (n, n, n))
    rows = np.stack([mids[0][ix], mids[1][iy], mids[2][iz]], axis=1).astype(np.float32)          # :160, :148
    return np.unique(rows, axis=0)                             # :149


def get_voxel_fun_full_table(points, n):
    """The literal form (materialises voxel_centers); only for small n, to check the shortcut above."""
    voxel_n, _, mids = voxel_grid(points, n)
    centers = cartesian(mids).astype(np.float32)
    return np.unique(centers[voxel_n], axis=0)
