"""Epoch loops of main_1v.py:237-356 (pointnetgpd_amd.train_loop): host logic on the CPU with the oracle model
standing in for the network, and the whole driver on the GPU."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth, train_loop
from pointnetgpd_amd.training import step_lr


class TorchTrainer:
    """What train_epoch needs from a trainer (.lr, .step -> (loss, log-probabilities)), on torch.optim.Adam."""

    def __init__(self, model, lr):
        self.model, self.lr = model, lr
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.lrs = []

    def step(self, data, target):
        for g in self.opt.param_groups:
            g["lr"] = self.lr
        self.lrs.append(self.lr)
        self.opt.zero_grad()
        out, _ = self.model(data)
        loss = F.nll_loss(out, target)
        loss.backward()
        self.opt.step()
        return loss.detach(), out.detach()


class Log:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step=None):
        self.rows.append((tag, float(value), step))


def small_set(n, P=32, k=2, seed=3):
    return train_loop._synthetic(n, P, k, seed)


def test_train_epoch_follows_the_reference_loop():
    P, k = 32, 2
    pts, lab = small_set(21, P, k)
    model = po.build_cls(P, k, synth.torch_state_dict(5, k, 1.0, 1.0), eval_mode=False)
    loader = train_loop.EpochLoader(train_loop.ArrayDataset(pts, lab), batch_size=4, shuffle=False, drop_last=False)
    assert len(loader) == 6                                   # 5 full batches + one of a single sample
    tr, log, lines = TorchTrainer(model, 0.005), Log(), []
    acc = train_loop.train_epoch(model, loader, tr, epoch=29, base_lr=0.005, log_interval=2, logger=log, tag="t",
                                 device="cpu", out=lines.append)
    assert len(tr.lrs) == 5                                   # the single-sample batch is skipped (main_1v.py:258)
    assert set(tr.lrs) == {step_lr(0.005, 29)} == {0.0025}    # StepLR stepped at the start of epoch 29 -> 30 steps
    assert 0.0 <= acc <= 1.0
    # accuracy counts the 20 samples seen, not the 21 of the dataset (main_1v.py:261, 302)
    assert abs(acc * 20 - round(acc * 20)) < 1e-9
    assert [r[2] for r in log.rows] == [0 + 29 * 6, 2 + 29 * 6, 4 + 29 * 6] and all(r[0] == "train_loss" for r in log.rows)
    assert lines[0].startswith("Train Epoch: 29 [0/21 (0%)]\tLoss: ") and lines[0].endswith("\tt")


def test_test_epoch_normalises_like_the_reference():
    P, k = 32, 3
    pts, lab = small_set(10, P, k)
    names = np.array([f"o{i}" for i in range(10)])
    model = po.build_cls(P, k, synth.torch_state_dict(6, k, 1.0, 1.0))
    loader = train_loop.EpochLoader(train_loop.ArrayDataset(pts, lab, names), batch_size=4, shuffle=False)
    acc, loss, res = train_loop.test_epoch(model, loader, device="cpu")
    with torch.no_grad():
        out, _ = model(torch.from_numpy(pts))
    ref_loss = F.nll_loss(out, torch.from_numpy(lab), reduction="sum").item() / 10
    ref_acc = float((out.argmax(1).numpy() == lab).mean())
    assert abs(loss - ref_loss) < 1e-5 and abs(acc - ref_acc) < 1e-12
    assert [r[0] for r in res] == list(names) and [r[2] for r in res] == list(lab)
    assert not model.training


def test_epoch_loader_shards_cover_the_epoch_once():
    pts, lab = small_set(23, 8)
    lab = np.arange(23)
    seen = []
    for rank in range(2):
        ld = train_loop.EpochLoader(train_loop.ArrayDataset(pts, lab), batch_size=4, shuffle=True, seed=7, rank=rank, world=2)
        for _, t in ld:
            seen += t.tolist()
    # equal shares: 11 samples per rank, the 23rd sits this epoch out; nothing is seen twice
    assert len(seen) == 22 and len(set(seen)) == 22 and set(seen) <= set(range(23))
    ld = train_loop.EpochLoader(train_loop.ArrayDataset(pts, lab), batch_size=4, shuffle=True, drop_last=True, seed=7)
    a = [t.tolist() for _, t in ld]
    b = [t.tolist() for _, t in ld]
    assert len(a) == 5 and a != b                              # a new permutation every epoch


def test_scalar_log_and_flags(tmp_path):
    log = train_loop.ScalarLog(str(tmp_path / "log"))
    log.add_scalar("test_acc", 0.5, 3)
    log.close()
    row = json.loads(open(log.path).read().splitlines()[0])
    assert row["tag"] == "test_acc" and row["value"] == 0.5 and row["step"] == 3
    flags = {a.option_strings[0] for a in train_loop.build_parser()._actions if a.option_strings}
    assert {"--tag", "--epoch", "--mode", "--batch-size", "--cuda", "--gpu", "--lr", "--load-model", "--load-epoch",
            "--model-path", "--data-path", "--log-interval", "--save-interval"} <= flags      # main_1v.py:35-59
    with pytest.raises(SystemExit):
        train_loop.build_parser().parse_args([])               # --mode is required, as in the reference


@pytest.mark.gpu
def test_driver_trains_saves_and_tests_on_the_gpu(tmp_path, capsys):
    common = ["--synthetic", "96", "--points", "64", "--batch-size", "16", "--model-path", str(tmp_path / "models"),
              "--log-dir", str(tmp_path / "log")]
    acc, loss = train_loop.main(["--mode", "train", "--epoch", "3", "--cuda"] + common)
    out = capsys.readouterr().out
    assert out.count("Train done, acc=") == 3 and out.count("Save model @ ") == 3
    rows = [json.loads(l) for l in open(next((tmp_path / "log").glob("*/scalars.jsonl")))]
    train_losses = [r["value"] for r in rows if r["tag"] == "train_loss"]
    assert train_losses[-1] < train_losses[0]                  # separable blobs: the loss falls
    assert {r["tag"] for r in rows} == {"train_loss", "train_acc", "test_acc", "test_loss"}
    # the last checkpoint, read back by the compat loader in --mode test, gives the last epoch's test result
    last = sorted((tmp_path / "models").glob("*_2.model"))[0]
    acc2, loss2 = train_loop.main(["--mode", "test", "--load-model", str(last)] + common)
    assert abs(acc2 - acc) < 1e-12 and abs(loss2 - loss) < 1e-5
    # and the oracle agrees with the HIP model on that checkpoint's weights
    from pointnetgpd_amd.compat import load_checkpoint
    m = load_checkpoint(str(last), device="cuda")
    own = po.build_cls(64, 2, {n: v.detach().cpu() for n, v in m.state_dict().items()})
    tp, tl = train_loop._synthetic(24, 64, 2, 2)
    with torch.no_grad():
        ref, _ = own(torch.from_numpy(tp))
    got, _ = m(torch.from_numpy(tp).cuda())
    assert (got.cpu() - ref).abs().max().item() < 1e-4 * max(1.0, float(ref.abs().max()))
    assert abs(F.nll_loss(ref, torch.from_numpy(tl), reduction="sum").item() / 24 - loss2) < 1e-4
