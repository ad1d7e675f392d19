"""Epoch loops of the reference's training script (main_1v.py:237-356) around the MI355X modules.

What is kept from the reference, because results depend on it:
  * the learning rate of epoch e is the one StepLR(step_size=30, gamma=0.5) holds AFTER its step() at the start
    of the epoch (main_1v.py:233, 239): `training.step_lr`;
  * a batch with a single sample is skipped (main_1v.py:258), data goes through .float(), targets through
    .long().squeeze() (:262); train accuracy = correct / samples actually seen (:261, :302);
  * the test loop sums the NLL over samples and divides by len(loader.dataset), accuracy divides by the samples
    seen (main_1v.py:305-328); its loader yields (data, target, obj_name);
  * scalars 'train_loss' (every log_interval batches, at batch_idx + epoch*len(loader)), 'train_acc', 'test_acc',
    'test_loss' (per epoch) (main_1v.py:295-297, 344-346);
  * every save_interval epochs the WHOLE module is pickled (main_1v.py:347-351) under the reference's class path,
    in torch's pre-1.6 container like the reference (`_use_new_zipfile_serialization=False`), so
    `compat.load_checkpoint` and the reference's own torch.load both read it back.

What is not reproduced: PointGraspOneViewDataset (it needs dex-net, autolab_core and python-pcl, model/dataset.py:5-29).
`main` trains on `.npz` shards holding `points[n,3,P]` float32 and `labels[n]` int64 (what that dataset's
__getitem__ returns, model/dataset.py:1128-1131) or on a seeded synthetic set; `dataset_crop.make_training_batch`
produces such arrays from raw clouds and grasp configurations on the GPU.

Several GPUs: one process per GPU (torchrun).  Every rank takes an EQUAL contiguous share of each epoch's
permutation (the n mod world samples left over sit the epoch out), so all ranks run the same number of steps with
the same batch sizes - a rank with one step more would wait for ever in its gradient all-reduce; the permutation
seed is rank 0's, and `Trainer` starts every rank from rank 0's weights and all-reduces the gradients.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import dist as _dist
from .training import Trainer, step_lr


class ScalarLog:
    """add_scalar(tag, value, step) like tensorboard's SummaryWriter (main_1v.py:88); one JSON object per line.

    torch.utils.tensorboard is used as well when it is importable (it is not in every ROCm image)."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "scalars.jsonl")
        self._f = open(self.path, "a")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(logdir)
        except Exception:
            self._tb = None

    def add_scalar(self, tag, scalar_value, global_step=None):
        self._f.write(json.dumps({"tag": tag, "value": float(scalar_value), "step": global_step,
                                  "time": time.time()}) + "\n")
        self._f.flush()
        if self._tb is not None:
            self._tb.add_scalar(tag, scalar_value, global_step)

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()


class ArrayDataset(torch.utils.data.Dataset):
    """(points[n,3,P], labels[n] [, names[n]]) held in memory; items as PointGraspOneViewDataset returns them."""

    def __init__(self, points, labels, names=None):
        if points.ndim != 3 or points.shape[1] != 3 or len(points) != len(labels):
            raise ValueError(f"expected points [n,3,P] and labels [n], got {points.shape} and {labels.shape}")
        self.points, self.labels, self.names = points, labels, names

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        if self.names is None:
            return self.points[i], self.labels[i]
        return self.points[i], self.labels[i], self.names[i]


class EpochLoader:
    """Minimal batch iterator over an ArrayDataset with what the loops need from a DataLoader: len(), .dataset,
    shuffle per epoch, drop_last, and the rank's share of every batch order when several processes train."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=False, seed=0, rank=0, world=1):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.seed, self.rank, self.world, self.epoch = seed, rank, world, 0

    def _share(self):
        """Samples per rank and epoch: the same number on every rank."""
        return len(self.dataset) // self.world

    def __len__(self):
        n = self._share()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        n = len(self.dataset)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        self.epoch += 1
        share = self._share()
        order = order[self.rank * share:(self.rank + 1) * share]
        ds = self.dataset
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            data = torch.from_numpy(np.ascontiguousarray(ds.points[idx]))
            target = torch.from_numpy(np.ascontiguousarray(ds.labels[idx]))
            if ds.names is None:
                yield data, target
            else:
                yield data, target, [ds.names[j] for j in idx]


def train_epoch(model, loader, trainer, epoch, base_lr=0.005, log_interval=10, logger=None, tag="default",
                device="cuda", out=print):
    """One epoch (main_1v.py:237-302); returns the training accuracy."""
    trainer.lr = step_lr(base_lr, epoch)
    model.train()
    correct = 0
    dataset_size = 0
    batch_size = getattr(loader, "batch_size", None) or 1
    k = model.fc3.out_features
    for batch_idx, (data, target) in enumerate(loader):
        if len(data) <= 1:            # the same on every rank: EpochLoader gives all ranks equal batch sizes
            continue
        if int(target.min()) < 0 or int(target.max()) >= k:
            raise ValueError(f"batch {batch_idx}: labels must lie in [0, {k}), got [{int(target.min())}, "
                             f"{int(target.max())}] (F.nll_loss raises on such a target, main_1v.py:270)")
        dataset_size += data.shape[0]
        data, target = data.float().to(device), target.long().squeeze().to(device)
        loss, output = trainer.step(data, target)
        pred = output.max(1, keepdim=True)[1]
        correct += int(pred.eq(target.view_as(pred)).long().sum())
        if batch_idx % log_interval == 0:
            n = len(loader.dataset)
            out("Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f}\t{}".format(
                epoch, batch_idx * batch_size, n, 100.0 * batch_idx * batch_size / n, loss.item(), tag))
            if logger is not None:
                logger.add_scalar("train_loss", loss.item(), batch_idx + epoch * len(loader))
    return float(correct) / float(dataset_size) if dataset_size else 0.0


@torch.no_grad()
def test_epoch(model, loader, device="cuda"):
    """Evaluation pass (main_1v.py:305-328); returns (accuracy, mean NLL, [(obj_name, pred, target), ...])."""
    model.eval()
    test_loss = 0.0
    correct = 0
    dataset_size = 0
    res = []
    for batch in loader:
        data, target = batch[0], batch[1]
        names = batch[2] if len(batch) > 2 else [None] * len(data)
        dataset_size += data.shape[0]
        data, target = data.float().to(device), target.long().squeeze().to(device)
        output, _ = model(data)
        test_loss += F.nll_loss(output, target.reshape(-1), reduction="sum").item()
        pred = output.max(1, keepdim=True)[1]
        correct += int(pred.eq(target.view_as(pred)).long().sum())
        for i, j, k in zip(names, pred.cpu().numpy(), target.reshape(-1).cpu().numpy()):
            res.append((i, int(j[0]), int(k)))
    test_loss /= len(loader.dataset)
    acc = float(correct) / float(dataset_size) if dataset_size else 0.0
    return acc, test_loss, res


def save_model(model, path, old_container=True):
    """Whole-module pickle under the reference's class path in the container the reference writes
    (torch.save(model, path, _use_new_zipfile_serialization=False), main_1v.py:350)."""
    from .compat import save_checkpoint
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_checkpoint(model, path, old_container=old_container)


def _load_arrays(path, tag, num_points):
    files = sorted(f for f in os.listdir(path) if f.startswith(tag) and f.endswith(".npz"))
    if not files:
        raise FileNotFoundError(f"no {tag}*.npz shards under {path}")
    pts, lab = [], []
    for f in files:
        with np.load(os.path.join(path, f)) as d:
            pts.append(np.asarray(d["points"], np.float32))
            lab.append(np.asarray(d["labels"], np.int64))
    pts, lab = np.concatenate(pts), np.concatenate(lab)
    if pts.shape[2] != num_points:
        raise ValueError(f"shards hold {pts.shape[2]} points per grasp, the model is built for {num_points}")
    return pts, lab


def _synthetic(n, num_points, k, seed):
    """Two separable blobs per class in the gripper box: enough for the loops to show a falling loss."""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, k, size=n).astype(np.int64)
    pts = rng.normal(0.0, 0.01, size=(n, 3, num_points)).astype(np.float32)
    pts[:, 0, :] += (0.01 * (lab[:, None] + 1)).astype(np.float32)
    return pts, lab


def build_parser():
    ap = argparse.ArgumentParser(description="pointnetGPD on MI355X")      # flags of main_1v.py:35-59
    ap.add_argument("--tag", type=str, default="default")
    ap.add_argument("--epoch", type=int, default=200)
    ap.add_argument("--mode", choices=["train", "test"], required=True)
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--cuda", action="store_true", help="accepted for compatibility: there is no CPU path")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--load-model", type=str, default="")
    ap.add_argument("--load-epoch", type=int, default=-1)
    ap.add_argument("--model-path", type=str, default="./assets/learned_models")
    ap.add_argument("--data-path", type=str, default="")
    ap.add_argument("--log-interval", type=int, default=10)
    ap.add_argument("--save-interval", type=int, default=1)
    # not in the reference: its constants (main_1v.py:118, 207) and a data source that needs no dex-net
    ap.add_argument("--points", type=int, default=750)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--synthetic", type=int, default=0, help="train/test on this many seeded synthetic samples")
    ap.add_argument("--log-dir", type=str, default="./assets/log")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    from .compat import load_checkpoint
    from .pointnet import PointNetCls

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL across processes needs here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(os.environ.get("PNGPD_DIST_BACKEND", "nccl"))
    torch.cuda.manual_seed(1)                                              # main_1v.py:78-79

    if args.synthetic:
        train_arrays = _synthetic(args.synthetic, args.points, args.classes, 1)
        test_arrays = _synthetic(max(args.synthetic // 4, 2), args.points, args.classes, 2)
    else:
        if not args.data_path:
            raise SystemExit("--data-path (npz shards) or --synthetic N is required")
        train_arrays = _load_arrays(args.data_path, "train", args.points) if args.mode == "train" else None
        test_arrays = _load_arrays(args.data_path, "test", args.points)
    test_names = np.array([f"obj{i}" for i in range(len(test_arrays[1]))])
    test_loader = EpochLoader(ArrayDataset(test_arrays[0], test_arrays[1], test_names), args.batch_size, shuffle=True)

    is_resume = bool(args.load_model and args.load_epoch != -1)
    if is_resume or args.mode == "test":
        if not args.load_model:
            raise SystemExit("--mode test needs --load-model")
        model = load_checkpoint(args.load_model, num_points=args.points, device=device)   # main_1v.py:199-204
        print("load model {}".format(args.load_model))
    else:
        model = PointNetCls(num_points=args.points, input_chann=3, k=args.classes).to(device)

    if args.mode == "test":
        print("testing...")
        acc, loss, _ = test_epoch(model, test_loader, device)
        print("Test done, acc={}, loss={}".format(acc, loss))
        return acc, loss

    current_time = time.strftime("%Y-%m-%dT%H:%M", time.localtime())
    logger = ScalarLog(os.path.join(args.log_dir, current_time)) if rank == 0 else None
    # one shuffle seed for all ranks (rank 0's clock): with different seeds the ranks' shares would overlap
    seed = _dist.broadcast_int(int(time.time()) if not args.synthetic else 0, device)
    train_loader = EpochLoader(ArrayDataset(*train_arrays), args.batch_size, shuffle=True, drop_last=True,
                               seed=seed, rank=rank, world=world)
    trainer = Trainer(model, lr=args.lr)
    acc = loss = None
    for epoch_i in range(int(is_resume) * args.load_epoch, args.epoch):
        acc_train = train_epoch(model, train_loader, trainer, epoch_i, args.lr, args.log_interval, logger, args.tag,
                                device, out=print if rank == 0 else (lambda *_: None))
        if rank != 0:
            continue
        print("Train done, acc={}".format(acc_train))
        acc, loss, _ = test_epoch(model, test_loader, device)
        print("Test done, acc={}, loss={}".format(acc, loss))
        logger.add_scalar("train_acc", acc_train, epoch_i)
        logger.add_scalar("test_acc", acc, epoch_i)
        logger.add_scalar("test_loss", loss, epoch_i)
        if epoch_i % args.save_interval == 0:
            path = os.path.join(args.model_path, current_time + "_{}.model".format(epoch_i))
            save_model(model, path)
            print("Save model @ {}".format(path))
    if logger is not None:
        logger.close()
    return acc, loss


if __name__ == "__main__":
    main()
