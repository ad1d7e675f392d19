"""CPU, world_size 2 over gloo: the candidate sharding used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointnetgpd_amd.dist import gather_concat, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 10_000, 65_536):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_total, rank, world)
        # stand-in for the per-rank scoring: a deterministic function of the GLOBAL candidate id
        ids = torch.arange(lo, hi, dtype=torch.float32)
        score = torch.sin(ids) * 0.5 + 0.5
        label = (ids.to(torch.int64) % 2).to(torch.int32)
        full_score = gather_concat(score, n_total)
        full_label = gather_concat(label, n_total)
        # flat gradient bucket: the one collective of the training step (sum, then / world)
        bucket = torch.full((1000,), float(rank + 1))
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        bucket /= world
        np.save(os.path.join(out_dir, f"score{rank}.npy"), full_score.numpy())
        np.save(os.path.join(out_dir, f"label{rank}.npy"), full_label.numpy())
        np.save(os.path.join(out_dir, f"bucket{rank}.npy"), bucket.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [11, 64])
def test_two_rank_gather_is_shard_invariant(tmp_path, n_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    ids = np.arange(n_total, dtype=np.float32)
    expect = (torch.sin(torch.from_numpy(ids)) * 0.5 + 0.5).numpy()
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"score{r}.npy"), expect)
        np.testing.assert_array_equal(np.load(tmp_path / f"label{r}.npy"), (ids.astype(np.int64) % 2).astype(np.int32))
        np.testing.assert_array_equal(np.load(tmp_path / f"bucket{r}.npy"), np.full(1000, 1.5, dtype=np.float32))
