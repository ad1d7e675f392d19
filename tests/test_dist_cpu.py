"""CPU, world_size 2 over gloo: the candidate sharding used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointnetgpd_amd.dist import broadcast_int, gather_concat, shard_range
from pointnetgpd_amd.train_loop import ArrayDataset, EpochLoader


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


def _loaders(n, world, bs, drop_last):
    ds = ArrayDataset(np.zeros((n, 3, 4), np.float32), np.arange(n, dtype=np.int64))
    return [EpochLoader(ds, bs, shuffle=True, drop_last=drop_last, seed=3, rank=r, world=world) for r in range(world)]


@pytest.mark.parametrize("n,world,bs", [(63, 2, 16), (999, 8, 125), (17, 2, 4), (7, 8, 2), (64, 1, 16)])
@pytest.mark.parametrize("drop_last", [True, False])
def test_epoch_loader_gives_every_rank_the_same_steps(n, world, bs, drop_last):
    """A rank with one step more than the others would wait for ever in its gradient all-reduce: step counts AND the
    size of every batch (a batch of one is skipped by the loop) are the same on all ranks, shares are disjoint."""
    loaders = _loaders(n, world, bs, drop_last)
    assert len({len(ld) for ld in loaders}) == 1
    seen, sizes = [], []
    for ld in loaders:
        batches = [t.numpy() for _, t in ld]
        assert len(batches) == len(ld)
        sizes.append([len(b) for b in batches])
        seen += [int(v) for b in batches for v in b]
    assert all(s == sizes[0] for s in sizes)
    assert len(seen) == len(set(seen))                              # no sample on two ranks
    share = n // world
    per_rank = (share // bs) * bs if drop_last else share
    assert len(seen) == per_rank * world
    # the next epoch draws another permutation, the same on every rank
    again = [int(v) for ld in loaders for _, t in ld for v in t.numpy()]
    assert len(again) == len(set(again)) and (world * per_rank < 4 or again != seen)


def _loop_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = broadcast_int(1234 + 1000 * rank)                    # rank 0's value everywhere
        ld = _loaders(63, world, 16, True)[rank]
        steps = 0
        for data, target in ld:
            if len(data) <= 1:
                continue
            t = torch.ones(4)
            dist.all_reduce(t)                                      # one collective per step, as Trainer.step has
            steps += 1
        np.save(os.path.join(out_dir, f"loop{rank}.npy"), np.array([seed, steps]))
    finally:
        dist.destroy_process_group()


def test_two_rank_epoch_runs_the_same_number_of_collectives(tmp_path):
    world = 2
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "loop0.npy"), np.load(tmp_path / "loop1.npy")
    assert a[0] == b[0] == 1234
    assert a[1] == b[1] == 1
