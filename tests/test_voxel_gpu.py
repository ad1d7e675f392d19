"""GPU parity of pngpd_voxel_unique (through the C ABI) against the CPU restatement of
get_voxel_fun (apps/kinect2grasp.py:145-150): the unique float32 voxel centres, bit-exact and in
the same order, for both edge types; the float32-edge form also against outputs of the reference's own
definitions (tests/golden/voxel_reference.npz, written by tests/golden/make_voxel_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import voxel_oracle as vo
from pointnetgpd_amd import synth, voxel

pytestmark = pytest.mark.gpu


def _check(pts, n):
    for edges in ("float64", "float32"):
        got = voxel.get_voxel_fun(torch.from_numpy(pts).cuda(), n, edges=edges).cpu().numpy()
        want = vo.get_voxel_fun(pts, n, edges=edges)
        assert got.shape == want.shape, (edges, got.shape, want.shape)
        np.testing.assert_array_equal(got, want, err_msg=edges)
    return len(want)


def test_reference_run_golden():
    """Outputs of the reference's own cartesian / VoxelGrid / get_voxel_fun (numpy 2: float32 edges)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxel_reference.npz"))
    for i, (seed, S, cube, n) in enumerate(d["cases"]):
        make = synth.scene_points if i % 2 == 0 else synth.elongated_scene_points
        pts = make(int(seed), int(S), float(cube))
        got = voxel.get_voxel_fun(torch.from_numpy(pts).cuda(), int(n), edges="float32").cpu().numpy()
        want = d[f"out_{i}"]
        assert got.shape == want.shape, (i, got.shape, want.shape)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 500])
def test_random_cloud(n):
    rng = np.random.default_rng(n)
    pts = (rng.random((20000, 3)) * np.array([0.5, 0.3, 0.1]) + np.array([0.2, -0.4, 0.9])).astype(np.float32)
    m = _check(pts, n)
    assert 1 <= m <= 20000


def test_points_on_bin_edges_and_corners():
    # a lattice whose points sit exactly on the bin edges of the n=8 grid over [0, 1]^3
    g = np.linspace(0, 1, 9, dtype=np.float32)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    assert _check(pts, 8) == 8 ** 3
    _check(pts, 5)
    _check(pts, 16)


def test_degenerate_and_tiny_clouds():
    _check(np.array([[0.3, -0.2, 1.5]] * 7, dtype=np.float32), 10)
    _check(np.array([[0.0, 0.0, 0.0]], dtype=np.float32), 3)
    _check(np.array([[0, 0, 0], [1, 0, 0]], dtype=np.float32), 4)            # flat in y, z: the box is still a cube


def test_float32_centres_that_coincide_are_one_row():
    # a 1e-4 wide cloud far from the origin: at n=1600 the bin width is below one float32 ulp of the
    # coordinates, so neighbouring bins share a float32 centre and np.unique merges them
    rng = np.random.default_rng(5)
    pts = (rng.random((4000, 3)) * 1e-4 + 37.0).astype(np.float32)
    m = _check(pts, 1600)
    assert m < 4000


def test_scene_size_and_idempotence():
    pts = synth.scene_points(0, 1_000_000)
    t = torch.from_numpy(pts).cuda()
    v = voxel.get_voxel_fun(t, 500)          # default edges: what the numpy installed here makes the reference compute
    assert voxel.default_edges() == ("float32" if int(np.__version__.split(".")[0]) >= 2 else "float64")
    want = vo.get_voxel_fun(pts, 500, edges=voxel.default_edges())
    np.testing.assert_array_equal(v.cpu().numpy(), want)
    # the same call with the count left on the device: same rows, nothing read back, buffers reused
    buf = torch.full((len(pts), 3), -7.0, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(2):
        out, count = voxel.voxel_unique_device(t, 500, out=buf, count=cnt)
        assert out.data_ptr() == buf.data_ptr() and int(count.item()) == len(want)
        assert torch.equal(out[:len(want)], v) and bool((out[len(want):] == -7.0).all())
    for edges in ("float64", "float32"):
        np.testing.assert_array_equal(voxel.get_voxel_fun(t, 500, edges=edges).cpu().numpy(), vo.get_voxel_fun(pts, 500, edges=edges))
    # the centres of a grid are fixed points of the same grid only if the box does not move; what
    # always holds: a second pass cannot produce more rows than the first, and rows stay sorted
    v2 = voxel.get_voxel_fun(v, 500)
    assert v2.shape[0] <= v.shape[0]
    a = v.cpu().numpy()
    assert (np.lexsort((a[:, 2], a[:, 1], a[:, 0])) == np.arange(len(a))).all()


def test_small_cloud_retry_loop():
    rng = np.random.default_rng(9)
    pts = rng.random((500, 3)).astype(np.float32) * 0.2
    v = voxel.voxelize_scene(torch.from_numpy(pts).cuda(), 20)
    assert v.shape[0] >= 500 - 15


def test_rejects_bad_arguments():
    t = torch.zeros(10, 3, device="cuda")
    with pytest.raises(ValueError):
        voxel.get_voxel_fun(t, 0)
    with pytest.raises(ValueError):
        voxel.get_voxel_fun(t, 1601)
    with pytest.raises(RuntimeError):
        voxel.get_voxel_fun(torch.zeros(10, 3), 4)
