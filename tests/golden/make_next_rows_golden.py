"""Regression vectors for the two "next" rows whose reference modules cannot be imported here
(apps/voxelgrid.py needs numba; model/dataset.py needs pcl/dexnet/autolab_core): the outputs of the
numpy restatements in oracle/ on small seeded inputs.  They pin the restatements against accidental
change; they are NOT outputs of the reference (parity unpinned, see the oracle headers).

    python tests/golden/make_next_rows_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import crop_oracle as co            # noqa: E402
from oracle import dataset_crop_oracle as do    # noqa: E402
from oracle import voxel_oracle as vo           # noqa: E402
from pointnetgpd_amd import synth               # noqa: E402


def main():
    pts = synth.scene_points(21, 4000, 0.3)
    np.savez_compressed(os.path.join(HERE, "voxel_small.npz"), seed=21, S=4000, cube=0.3, n=np.array([1, 3, 16, 40]),
                        **{f"out_n{n}": vo.get_voxel_fun(pts, n) for n in (1, 3, 16, 40)})

    grasps, T, pc = synth.dataset_view(22, 60_000, 10)
    hp = co.hand_points(0.06, 0.10, 0.01, 0.02)
    counts, first = [], []
    for g in grasps:
        ind, gpc = do.collect_pc_(g, pc, T, 0.06, hp, min_point_limit=0)
        counts.append(len(ind))
        first.append(gpc[:4] if len(ind) >= 4 else np.zeros((4, 3)))
    np.savez_compressed(os.path.join(HERE, "dataset_crop_small.npz"), seed=22, S=60_000, G=10,
                        counts=np.array(counts), first_points=np.array(first))


if __name__ == "__main__":
    main()
