"""Two ranks (two processes sharing the one GPU of the test box, gloo with host staging - RCCL needs one GPU per
rank) through the product's own multi-rank code: Trainer (weights broadcast from rank 0, gradient buckets reduced
part by part, fused Adam), dist.score_grasps_sharded, and train_loop under torchrun's environment."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

P, K, B = 64, 2, 6
STEPS = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank, step):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    x = torch.randn(B, 3, P, generator=g) * 0.03
    t = torch.randint(0, K, (B,), generator=g)
    return x, t


def _fresh_model(seed):
    from pointnetgpd_amd.pointnet import PointNetCls
    torch.manual_seed(seed)                       # every rank initialises differently, like separate processes do
    m = PointNetCls(P, 3, K)
    for mod in m.modules():                       # non-trivial BatchNorm state, different per seed
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    return m.cuda().train()


def _train_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnetgpd_amd.training import Trainer
        m = _fresh_model(seed=rank)
        before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        tr = Trainer(m, lr=0.005)
        after_init = tr.flat_p.cpu().clone()
        stats0 = torch.cat([b.detach().reshape(-1).double() for b in m.buffers()]).cpu()
        losses = []
        for step in range(STEPS):
            x, t = _batch(rank, step)
            loss, _ = tr.step(x.cuda(), t.cuda())
            losses.append(loss.item())
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"train{rank}.npz"), before=before.numpy(), after_init=after_init.numpy(),
                 stats0=stats0.numpy(), final=tr.flat_p.cpu().numpy(), grad=tr.flat_g.cpu().numpy(), losses=np.array(losses))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_equals_one_process_on_the_summed_gradient(tmp_path):
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"train{i}.npz") for i in range(world)]
    # different initialisations before the Trainer, rank 0's on both ranks after it (parameters and BN buffers)
    assert np.abs(r[0]["before"] - r[1]["before"]).max() > 1e-3
    np.testing.assert_array_equal(r[0]["after_init"], r[0]["before"])
    np.testing.assert_array_equal(r[1]["after_init"], r[0]["before"])
    np.testing.assert_array_equal(r[0]["stats0"], r[1]["stats0"])
    # the replicas stay identical through the steps, and hold the same summed gradient
    np.testing.assert_array_equal(r[0]["final"], r[1]["final"])
    np.testing.assert_array_equal(r[0]["grad"], r[1]["grad"])
    assert np.abs(r[0]["final"] - r[0]["before"]).max() > 1e-3

    # one process, no process group: two replicas of rank 0's model, each on its rank's batches; the gradients are
    # summed by hand and both apply Adam with 1/2.  Same kernels, same order: bit-identical parameters.
    from pointnetgpd_amd.training import Trainer
    reps = [_fresh_model(seed=0), _fresh_model(seed=0)]
    trs = [Trainer(m, lr=0.005) for m in reps]
    for step in range(STEPS):
        losses = []
        for rank, tr in enumerate(trs):
            x, t = _batch(rank, step)
            losses.append(tr.gradients(x.cuda(), t.cuda())[0].item())
        total = trs[0].flat_g + trs[1].flat_g
        for rank, tr in enumerate(trs):
            tr.flat_g.copy_(total)
            tr.apply(world=2)
            assert losses[rank] == r[rank]["losses"][step]
    np.testing.assert_array_equal(trs[0].flat_p.cpu().numpy(), r[0]["final"])
    np.testing.assert_array_equal(trs[1].flat_p.cpu().numpy(), r[1]["final"])


def _score_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnetgpd_amd.dist import score_grasps_sharded
        model, scene, grasps, limits = _scoring_case()
        label, score, valid = score_grasps_sharded(model, scene, grasps, limits, repeat=3, seed=5)
        np.savez(os.path.join(out_dir, f"score{rank}.npz"), label=label.cpu().numpy(), score=score.cpu().numpy(),
                 valid=valid.cpu().numpy())
    finally:
        dist.destroy_process_group()


def _scoring_case():
    from oracle import crop_oracle as co            # hand_points / box_limits only: inputs
    from pointnetgpd_amd import synth
    from pointnetgpd_amd.pointnet import PointNetCls
    limits = co.box_limits(co.hand_points(hand_depth=0.06, hand_outer_diameter=0.12, finger_width=0.02, hand_height=0.02))
    m = PointNetCls(750, 3, 2)
    m.load_state_dict(synth.torch_state_dict(4, 2, 0.02))
    m = m.cuda().eval()
    scene = torch.from_numpy(synth.scene_points(9, 300_000, 0.3)).cuda()
    grasps = synth.random_grasps(9, 41, 0.3, 0.05)           # 41: the two shards differ in size
    return m, scene, grasps, limits


def test_sharded_scoring_equals_single_process_scoring(tmp_path):
    world = 2
    mp.spawn(_score_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from pointnetgpd_amd.online import score_grasps
    model, scene, grasps, limits = _scoring_case()
    one = score_grasps(model, scene, grasps, limits, repeat=3, seed=5)
    assert int(one.valid.sum()) > 10
    for rank in range(world):
        got = np.load(tmp_path / f"score{rank}.npz")
        np.testing.assert_array_equal(got["label"], one.label.cpu().numpy())
        np.testing.assert_array_equal(got["score"], one.score.cpu().numpy())
        np.testing.assert_array_equal(got["valid"], one.valid.cpu().numpy())


def _loop_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PNGPD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from pointnetgpd_amd import train_loop
    torch.manual_seed(100 + rank)
    # 63 samples over 2 ranks in batches of 16: 31 each -> one step per rank and epoch (an unequal split would give
    # rank 0 two steps and leave it waiting in its all-reduce)
    train_loop.main(["--mode", "train", "--synthetic", "63", "--points", "64", "--batch-size", "16", "--epoch", "2",
                     "--model-path", os.path.join(out_dir, f"models{rank}"), "--log-dir", os.path.join(out_dir, f"log{rank}")])
    import torch.distributed as dist
    from pointnetgpd_amd.pointnet import PointNetCls  # noqa: F401
    dist.destroy_process_group()


def test_train_loop_two_ranks_finishes_and_saves(tmp_path):
    world = 2
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    saved = sorted(os.listdir(tmp_path / "models0"))
    assert len(saved) == 2 and all(f.endswith(".model") for f in saved)
    assert not os.path.exists(tmp_path / "models1")            # only rank 0 writes checkpoints


def _rccl_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from pointnetgpd_amd.training import Trainer
        res = {}
        for name, bucketed in (("plain", False), ("bucketed", True)):
            tr = Trainer(_fresh_model(seed=0), lr=0.005)
            tr.always_bucketed = bucketed
            if bucketed:
                tr.broadcast_state()                  # the device-side broadcast as well
            for step in range(STEPS):
                x, t = _batch(0, step)
                tr.step(x.cuda(), t.cuda())
            torch.cuda.synchronize()
            res[name] = tr.flat_p.cpu().numpy()
            res[name + "_g"] = tr.flat_g.cpu().numpy()
        np.savez(os.path.join(out_dir, "rccl.npz"), **res)
    finally:
        dist.destroy_process_group()


def test_bucketed_backward_over_rccl_equals_the_single_call(tmp_path):
    """What a multi-GPU run executes - pngpd_train_backward_part three times, an asynchronous RCCL all_reduce per
    bucket queued behind each part, the waits, Adam - on a one-rank RCCL group (the test box has one GPU): the same
    parameters and gradients, bit for bit, as the single-call backward."""
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    np.testing.assert_array_equal(r["plain"], r["bucketed"])
    np.testing.assert_array_equal(r["plain_g"], r["bucketed_g"])
