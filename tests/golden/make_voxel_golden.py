"""Golden vectors for the scene voxel down-sampling, produced by the REFERENCE'S OWN definitions.

Build container only (needs /root/reference):  python tests/golden/make_voxel_golden.py

apps/voxelgrid.py imports numba at module level, which this image lacks, but `cartesian` (apps/voxelgrid.py:40-86),
the `VoxelGrid` class (:89-160; only __init__ and compute are called) and `get_voxel_fun`
(apps/kinect2grasp.py:145-150) need only numpy.  They are taken out of the reference files' syntax trees, compiled
with numpy as their only outside name and run on seeded clouds; the .npz holds the seeds and their outputs, nothing
of the reference's text.

The run uses this image's numpy (2.2): np.linspace on two float32 scalars stays in float32 there, so these vectors
pin the float32-edge form (PNGPD_EDGES_F32, oracle edges="float32").  The numpy 1.x form differs in that call only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from make_crop_golden import reference_function  # noqa: E402
from pointnetgpd_amd import synth  # noqa: E402

VOXELGRID_FILE = "/root/reference/apps/voxelgrid.py"

CASES = [  # (seed, S, cube, n)
    (21, 4000, 0.3, 1), (21, 4000, 0.3, 3), (21, 4000, 0.3, 16), (21, 4000, 0.3, 40),
    (22, 30000, 0.5, 64), (23, 1500, 0.2, 120),
]


def reference_get_voxel_fun():
    cartesian = reference_function("cartesian", VOXELGRID_FILE)
    voxel_grid_class = reference_function("VoxelGrid", VOXELGRID_FILE, {"cartesian": cartesian})
    return reference_function("get_voxel_fun", extra_globals={"voxelgrid": types.SimpleNamespace(VoxelGrid=voxel_grid_class)})


def main():
    get_voxel_fun = reference_get_voxel_fun()
    out = {"numpy_version": np.__version__, "cases": np.array(CASES, dtype=np.float64)}
    for i, (seed, S, cube, n) in enumerate(CASES):
        pts = synth.scene_points(seed, S, cube) if i % 2 == 0 else synth.elongated_scene_points(seed, S, cube)
        v = get_voxel_fun(pts, n)
        assert v.dtype == np.float32
        print(f"case {i}: S={S} n={n} -> {len(v)} centres")
        out[f"out_{i}"] = v
    np.savez_compressed(os.path.join(HERE, "voxel_reference.npz"), **out)


if __name__ == "__main__":
    main()
