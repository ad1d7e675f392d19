"""GPU parity of the training step (main_1v.py:262-274) against the reference's golden one-step
vectors and against the CPU oracle: loss, gradients, post-Adam parameters, running statistics."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
from pointnetgpd_amd.training import Trainer, nll_loss, step_lr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def load_golden():
    d = np.load(os.path.join(GOLDEN, "train_step_p750_k2.npz"))
    P, k, seed, B = (int(v) for v in d["meta"])
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy(d["target"])
    return d, P, k, seed, B, float(d["gain"]), x, target


def hip_train_model(P, k, seed, gain):
    m = PointNetCls(P, 3, k)
    m.load_state_dict(synth.torch_state_dict(seed, k, 1.0, gain))
    return m.cuda().train()


def test_autograd_path_matches_reference_golden_step():
    """output,_ = model(data); F.nll_loss(...).backward(); torch.optim.Adam step - unchanged caller code."""
    d, P, k, seed, B, gain, x, target = load_golden()
    m = hip_train_model(P, k, seed, gain)
    opt = torch.optim.Adam(m.parameters(), lr=0.005)
    opt.zero_grad()
    out, _ = m(x.cuda())
    loss = F.nll_loss(out, target.cuda())
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-5
    np.testing.assert_allclose(out.detach().cpu().numpy(), d["logp"], atol=5e-5)
    grads = {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}
    worst = 0.0
    for key in d.files:
        if key.startswith("grad/"):
            n = key[5:]
            if np.abs(d[key]).max() < 1e-5:
                # parameters whose true gradient is zero (a shift that the next BatchNorm removes):
                # the reference holds rounding noise there
                assert np.abs(grads[n]).max() < 1e-5, n
                continue
            e = rel_err(grads[n], d[key])
            worst = max(worst, e)
            assert e < 2e-3, (n, e)
        elif key.startswith("gradslice/"):
            n = key[10:]
            g = grads[n].reshape(grads[n].shape[0], -1)
            assert rel_err(g[:4, :8], d[key]) < 5e-3, n
            s = d["gradsum/" + n]
            assert abs(np.abs(g.astype(np.float64)).sum() - s[1]) / s[1] < 1e-3, n
    print("worst relative gradient error", worst)
    opt.step()
    sd = {n: v.detach().cpu().numpy() for n, v in m.state_dict().items()}
    for key in d.files:
        if key.startswith("stat/"):
            np.testing.assert_allclose(sd[key[5:]], d[key], rtol=1e-4, atol=1e-5, err_msg=key)
        elif key.startswith("param/"):
            n = key[6:]
            clear = np.abs(d["grad/" + n]) > 1e-5          # Adam's first step is lr*sign(g): skip |g| ~ 0
            np.testing.assert_allclose(sd[n][clear], d[key][clear], atol=2e-5, err_msg=key)
    assert int(sd["feat.stn.bn1.num_batches_tracked"]) == 1


def test_fused_trainer_matches_autograd_path_and_oracle():
    """Two fused steps (forward, NLL, backward, Adam), each compared with a float64 step of the restated network that
    starts from the HIP model's own state (parameters, running statistics, Adam moments - per-step parity, which is
    what a drop-in has to offer) and is forced onto the decisions the HIP forward took (arg-max rows, ReLU masks;
    checked to be right to rounding): loss, every gradient, the post-Adam parameters and the running statistics.
    A plain float32 reference cannot arbitrate here: with 10^4 max decisions per step one is usually tied to within
    float32 rounding, and then two correct implementations route that gradient to different rows."""
    from pointnetgpd_amd import training
    P, k, seed, B = 500, 3, 77, 6
    gain = 1.7
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd0 = synth.torch_state_dict(seed, k, 1.0, gain)
    m = PointNetCls(P, 3, k)
    m.load_state_dict(sd0)
    m = m.cuda().train()
    tr = Trainer(m, lr=0.005)

    def hip_moments():
        out, off = {}, 0
        for name, p in m.named_parameters():       # the order Trainer lays its flat buffers out in
            n = p.numel()
            out[name] = (tr.exp_avg[off:off + n].cpu().clone(), tr.exp_avg_sq[off:off + n].cpu().clone())
            off += n
        return out

    def hip_grads():
        out, off = {}, 0
        for name, p in m.named_parameters():
            out[name] = tr.flat_g[off:off + p.numel()].cpu().view_as(p).double()
            off += p.numel()
        return out

    for step in (1, 2):
        sd = {n: v.detach().cpu().clone() for n, v in m.state_dict().items()}
        moments = hip_moments() if step > 1 else None
        loss = tr.step(x.cuda(), target.cuda())[0].item()
        ws = tr._ws                                    # holds the decisions of the forward of this step
        arg_stn, arg_trunk = (a.cpu() for a in training.critical_points(ws, B, P, k))
        masks = {n: (training.saved_activation(ws, B, P, k, n) > 0).cpu() for n in training.SAVED}
        o = po.build_cls(P, k, sd, eval_mode=False).double()
        opt = torch.optim.Adam(o.parameters(), lr=0.005)
        if moments is not None:
            for n, p in o.named_parameters():
                opt.state[p] = {"step": torch.tensor(float(step - 1)), "exp_avg": moments[n][0].view_as(p).double(),
                                "exp_avg_sq": moments[n][1].view_as(p).double()}
        ref_logp, kinks, pools = forced_forward(o, x.double(), arg_stn, arg_trunk, masks)
        for chosen, true_max in pools:                 # the chosen row holds the maximum
            assert float((true_max - chosen).detach().max()) < 2e-6 * float(true_max.detach().abs().max())
        for pre, mask in kinks:                        # a mask differs from the float64 sign only at ~0
            wrong = (pre.detach() > 0) != mask
            if wrong.any():
                assert float(pre.detach()[wrong].abs().max()) < 2e-5 * float(pre.detach().abs().max())
        ref_loss = F.nll_loss(ref_logp, target)
        ref_loss.backward()
        assert abs(loss - ref_loss.item()) < 2e-5 * max(1.0, abs(ref_loss.item()))
        g64 = {n: p.grad.clone() for n, p in o.named_parameters()}
        opt.step()
        gh, got_sd, ref_sd = hip_grads(), m.state_dict(), o.state_dict()
        worst_g = worst_p = 0.0
        for n, g in g64.items():
            scale = float(g.abs().max())
            if scale < 1e-7:
                assert float(gh[n].abs().max()) < 1e-5, n
                continue
            worst_g = max(worst_g, float((gh[n] - g).abs().max()) / scale)
            clear = g.abs() > 1e-4                    # below that the Adam update is the sign of rounding noise
            if clear.any():
                worst_p = max(worst_p, float((got_sd[n].cpu().double() - ref_sd[n]).abs()[clear].max()))
        print(f"step {step}: gradients within {worst_g:.2e} of each tensor's largest entry, post-Adam parameters "
              f"within {worst_p:.2e} of the float64 step")
        assert worst_g < 1e-4 and worst_p < 5e-6, (step, worst_g, worst_p)
        for n in ref_sd:
            if n.endswith(("running_mean", "running_var")):
                np.testing.assert_allclose(got_sd[n].cpu().numpy(), ref_sd[n].numpy(), rtol=2e-4, atol=2e-5, err_msg=n)
    # eval after training must re-fold / re-pack the weights the fused update wrote: compare with the
    # oracle evaluated on the HIP model's OWN post-training state_dict
    m.eval()
    own = po.build_cls(P, k, {n: v.detach().cpu() for n, v in m.state_dict().items()})
    with torch.no_grad():
        ref_logp, _ = own(x)
    logp, _ = m(x.cuda())
    scale = max(1.0, float(ref_logp.abs().max()))
    assert (logp.cpu() - ref_logp).abs().max().item() < 1e-4 * scale


def test_nll_loss_kernel():
    logp = torch.log_softmax(torch.randn(37, 3, generator=torch.Generator().manual_seed(0)), 1)
    target = torch.randint(0, 3, (37,), generator=torch.Generator().manual_seed(1))
    loss, dlogp = nll_loss(logp.cuda(), target.cuda())
    ref = logp.clone().requires_grad_(True)
    l = F.nll_loss(ref, target)
    l.backward()
    assert abs(loss.item() - l.item()) < 1e-6
    np.testing.assert_allclose(dlogp.cpu().numpy(), ref.grad.numpy(), atol=1e-7)


def test_nll_loss_label_out_of_range_is_not_used_as_an_index():
    """F.nll_loss raises on a target outside [0, k); the kernel must not read out of bounds: NaN loss, and the
    offending row gets no gradient while the others keep theirs."""
    logp = torch.log_softmax(torch.randn(8, 2, generator=torch.Generator().manual_seed(0)), 1).cuda()
    for bad in (2, -1, 10 ** 12):
        target = torch.tensor([0, 1, 1, bad, 0, 0, 1, 1], dtype=torch.int64).cuda()
        loss, dlogp = nll_loss(logp, target)
        assert torch.isnan(loss).all()
        d = dlogp.cpu()
        assert (d[3] == 0).all() and d[0, 0] == -1.0 / 8 and d[1, 1] == -1.0 / 8


def test_batch_of_one_is_rejected_like_the_reference_skips_it():
    m = hip_train_model(750, 2, 1, 1.7)
    with pytest.raises(ValueError, match="more than one"):
        m(torch.zeros(1, 3, 750, device="cuda"))


def test_step_lr_schedule():
    assert step_lr(0.005, 0) == 0.005 and step_lr(0.005, 28) == 0.005
    assert step_lr(0.005, 29) == 0.0025 and step_lr(0.005, 59) == 0.00125


def test_every_gradient_against_oracle():
    P, k, seed, B, gain = 750, 2, 23, 5, 1.7
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) < 0.5).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    oracle = po.build_cls(P, k, sd, eval_mode=False)
    out, _ = oracle(x)
    F.nll_loss(out, target).backward()
    m = hip_train_model(P, k, seed, gain)
    out2, trans2 = m(x.cuda())
    F.nll_loss(out2, target.cuda()).backward()
    ref = {n: p.grad.numpy() for n, p in oracle.named_parameters()}
    checked = 0
    for n, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        if np.abs(ref[n]).max() < 1e-5:
            assert np.abs(g).max() < 1e-5, n
        else:
            assert rel_err(g, ref[n]) < 2e-3, (n, rel_err(g, ref[n]))
            checked += 1
    assert checked >= 30
    # running statistics of all ten BatchNorm layers
    ref_sd, got_sd = oracle.state_dict(), m.state_dict()
    for n in ref_sd:
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(got_sd[n].cpu().numpy(), ref_sd[n].numpy(), rtol=2e-4, atol=2e-5, err_msg=n)


@pytest.mark.parametrize("P,B,k,seed", [(5, 16, 2, 5), (40, 8, 3, 6), (257, 9, 2, 7)])
def test_gradients_and_statistics_at_odd_shapes(P, B, k, seed):
    """Shapes that leave point tiles, channel splits and per-wave row ranges of the conv3 path partly empty: the
    continuous outputs (log-probabilities, transform, running statistics) against the float32 oracle.  The gradients
    of these shapes are checked in test_gradients_with_the_kernels_own_decisions: the max over points and the ReLUs
    make them discontinuous, and where two points tie to within float32 rounding two correct implementations route
    the gradient to different rows (P=257, B=9: the float32 oracle itself is 1.5e-3 away from a float64 step)."""
    gain = 1.3
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    oracle = po.build_cls(P, k, sd, eval_mode=False)
    out, trans = oracle(x)
    F.nll_loss(out, target).backward()
    m = hip_train_model(P, k, seed, gain)
    out2, trans2 = m(x.cuda())
    F.nll_loss(out2, target.cuda()).backward()
    np.testing.assert_allclose(out2.detach().cpu().numpy(), out.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(trans2.detach().cpu().numpy(), trans.detach().numpy(), atol=1e-4)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    ref_sd, got_sd = oracle.state_dict(), m.state_dict()
    for n in ref_sd:
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(got_sd[n].cpu().numpy(), ref_sd[n].numpy(), rtol=2e-4, atol=2e-5, err_msg=n)


def test_zero_and_tiny_batchnorm_scale_on_the_pooled_layers():
    """gamma == 0 makes every point of a candidate tie (the first one wins, as in torch) and leaves only dgamma
    non-zero for that channel; a gamma that is tiny against beta must not lose the normalised value either."""
    P, k, seed, B, gain = 200, 2, 31, 7, 1.3
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    for name in ("feat.bn3.weight", "feat.stn.bn3.weight"):
        w = sd[name].clone()
        w[::7] = 0.0
        w[3::7] = 1e-4 * torch.sign(w[3::7])
        sd[name] = w
    oracle = po.build_cls(P, k, sd, eval_mode=False)
    out, _ = oracle(x)
    F.nll_loss(out, target).backward()
    m = PointNetCls(P, 3, k)
    m.load_state_dict(sd)
    m = m.cuda().train()
    out2, _ = m(x.cuda())
    F.nll_loss(out2, target.cuda()).backward()
    np.testing.assert_allclose(out2.detach().cpu().numpy(), out.detach().numpy(), atol=1e-4)
    ref = {n: p.grad.numpy() for n, p in oracle.named_parameters()}
    for n, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        if np.abs(ref[n]).max() < 1e-5:
            assert np.abs(g).max() < 1e-5, n
        else:
            assert rel_err(g, ref[n]) < 2e-3, (n, rel_err(g, ref[n]))
    for name in ("feat.bn3.weight", "feat.stn.bn3.weight"):       # the channels in question, one by one
        g, r = m.get_parameter(name).grad.cpu().numpy(), ref[name]
        scale = np.abs(r).max()
        assert np.abs(g[::7] - r[::7]).max() < 2e-3 * scale and np.abs(g[3::7] - r[3::7]).max() < 2e-3 * scale, name


def forced_forward(o, x, arg_stn, arg_trunk, masks):
    """Train-mode forward of the restated network (float64) that takes every non-smooth decision from outside: the
    two max-pools read the GIVEN point of each (candidate, channel), every ReLU passes where the GIVEN mask is set.
    Returns (logp, list of (pre-activation, mask) pairs, list of (chosen value, true maximum) pairs)."""
    B, _, P = x.shape
    kinks, pools = [], []

    def gate(pre, mask):
        kinks.append((pre, mask))
        return pre * mask.to(pre.dtype)

    def rows(m, C):                                     # [B*P, C] rows -> [B, C, P]
        return m.reshape(B, P, C).permute(0, 2, 1)

    def stack(mod, h, name, arg, relu_out):
        h = gate(mod.bn1(mod.conv1(h)), rows(masks[name + "_h1"], 64))
        h = gate(mod.bn2(mod.conv2(h)), rows(masks[name + "_h2"], 128))
        y = mod.bn3(mod.conv3(h))
        v = y.gather(2, arg.long().unsqueeze(2)).squeeze(2)
        pools.append((v, y.max(dim=2).values))
        return gate(v, masks[name + "_pooled"]) if relu_out else v

    stn = o.feat.stn
    ps = stack(stn, x, "stn", arg_stn, True)
    h = gate(stn.bn4(stn.fc1(ps)), masks["stn_a1"])
    h = gate(stn.bn5(stn.fc2(h)), masks["stn_a2"])
    iden = torch.eye(3, dtype=x.dtype).reshape(1, 9)
    trans = (stn.fc3(h) + iden).reshape(-1, 3, 3)
    forced_forward.trans, forced_forward.feat = trans, None
    if arg_trunk is None:                               # the transform regressor alone
        return None, kinks, pools
    xt = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
    pt = stack(o.feat, xt, "trunk", arg_trunk, False)
    forced_forward.feat = pt
    if "head_a1" not in masks:                          # a model without a head
        return None, kinks, pools
    h = gate(o.bn1(o.fc1(pt)), masks["head_a1"])
    h = gate(o.bn2(o.fc2(h)), masks["head_a2"])
    return F.log_softmax(o.fc3(h), dim=1), kinks, pools


@pytest.mark.parametrize("which", ["feat", "stn"])
def test_standalone_submodules_in_train_mode(which):
    """PointNetfeat and STN3d on their own in train() mode (model/pointnet.py:135-149, :28-45 with batch-statistics
    BatchNorm), forward and backward through autograd, against a float64 step of the restated modules forced onto the
    decisions the HIP forward took: outputs, every gradient, running statistics."""
    from pointnetgpd_amd import training
    from pointnetgpd_amd.pointnet import PointNetfeat, STN3d
    P, B, seed, gain = 200, 12, 43, 1.3
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    sd = synth.torch_state_dict(seed, 2, 1.0, gain)
    g = torch.Generator().manual_seed(seed)
    w_feat, w_trans = torch.randn(B, 1024, generator=g), torch.randn(B, 3, 3, generator=g)
    if which == "feat":
        m = PointNetfeat(P, 3)
        m.load_state_dict({n[len("feat."):]: v for n, v in sd.items() if n.startswith("feat.")})
        names = ("stn_h1", "stn_h2", "stn_pooled", "stn_a1", "stn_a2", "trunk_h1", "trunk_h2", "trunk_pooled")
    else:
        m = STN3d(P, 3)
        m.load_state_dict({n[len("feat.stn."):]: v for n, v in sd.items() if n.startswith("feat.stn.")})
        names = ("stn_h1", "stn_h2", "stn_pooled", "stn_a1", "stn_a2")
    m = m.cuda().train()
    # the autograd route a caller takes
    if which == "feat":
        feat, trans = m(x.cuda())
        loss = (feat * w_feat.cuda()).sum() + (trans * w_trans.cuda()).sum()
    else:
        trans = m(x.cuda())
        loss = (trans * w_trans.cuda()).sum()
    loss.backward()
    got = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
    stats = {n: v.detach().cpu().clone() for n, v in m.state_dict().items()}
    # the same forward again on a twin (identical state before the step) to read the decisions it took
    twin = type(m)(P, 3)
    cut = len("feat.") if which == "feat" else len("feat.stn.")
    twin.load_state_dict({n[cut:]: v for n, v in sd.items() if n.startswith("feat." if which == "feat" else "feat.stn.")})
    twin = twin.cuda().train()
    out = training.train_forward(twin, x.cuda())
    ws = out[2]
    arg_stn, arg_trunk = (a.cpu() for a in training.critical_points(ws, B, P, 2))
    masks = {n: (training.saved_activation(ws, B, P, 2, n) > 0).cpu() for n in names}
    o = po.build_cls(P, 2, sd, eval_mode=False).double()
    _, kinks, pools = forced_forward(o, x.double(), arg_stn, arg_trunk if which == "feat" else None, masks)
    for chosen, true_max in pools:
        assert float((true_max - chosen).detach().max()) < 2e-6 * float(true_max.detach().abs().max())
    for pre, mask in kinks:
        wrong = (pre.detach() > 0) != mask
        if wrong.any():
            assert float(pre.detach()[wrong].abs().max()) < 2e-5 * float(pre.detach().abs().max())
    ref_trans = forced_forward.trans
    np.testing.assert_allclose(trans.detach().cpu().numpy(), ref_trans.detach().numpy(), atol=1e-4)
    ref_loss = (ref_trans * w_trans.double()).sum()
    if which == "feat":
        np.testing.assert_allclose(feat.detach().cpu().numpy(), forced_forward.feat.detach().numpy(), rtol=2e-5, atol=1e-4)
        ref_loss = ref_loss + (forced_forward.feat * w_feat.double()).sum()
    ref_loss.backward()
    prefix = "feat." if which == "feat" else "feat.stn."
    checked = 0
    for n, gh in got.items():
        r = o.get_parameter(prefix + n).grad
        scale = float(r.abs().max())
        if scale < 1e-7 * max(1.0, float(w_trans.abs().max())):
            assert float(gh.abs().max()) < 1e-4, n
            continue
        assert float((gh - r).abs().max()) / scale < 1e-4, (n, float((gh - r).abs().max()) / scale)
        checked += 1
    assert checked >= (16 if which == "feat" else 10)
    ref_sd = o.state_dict()
    for n, v in stats.items():
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(v.numpy(), ref_sd[prefix + n].numpy(), rtol=2e-4, atol=2e-5, err_msg=n)
        elif n.endswith("num_batches_tracked"):
            assert int(v) == 1


@pytest.mark.parametrize("P,B,k,seed", [(130, 70, 2, 8), (64, 70, 2, 8), (64, 70, 2, 13), (37, 150, 2, 9),
                                        (300, 20, 3, 10), (750, 24, 2, 11), (750, 256, 2, 12),
                                        (5, 16, 2, 5), (40, 8, 3, 6), (257, 9, 2, 7), (500, 6, 3, 77)])
def test_gradients_with_the_kernels_own_decisions(P, B, k, seed):
    """A PointNet step is piecewise smooth: the max over points and every ReLU switch where two values tie to within
    float32 rounding, and with ~10^6 such decisions per step a few always do.  Two correct float32 implementations
    then differ by O(1e-3 .. 1e-2) in isolated gradient entries and everything upstream of them (seed 8 at P=64,
    B=70: trunk bn2 of candidate 55, channel 48, point 32 is -7.8e-7), so a plain comparison at realistic sizes
    tests luck.  This test reads the decisions the HIP forward took - arg-max rows (pngpd_train_argmax) and ReLU
    masks (signs of the saved activations, pngpd_train_saved) - checks that each is right to rounding (the chosen
    row holds the maximum, a mask disagrees with the float64 sign only where the pre-activation is ~0), and then
    compares every gradient with a float64 step of the restated network forced onto the same decisions."""
    from pointnetgpd_amd import training
    gain = 1.3
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    m = hip_train_model(P, k, seed, gain)
    logp, trans, ws, xc = training.train_forward(m, x.cuda())
    loss, dlogp = nll_loss(logp, target.cuda())
    grads = {id(p): torch.zeros_like(p) for p in m.parameters()}
    arg_stn, arg_trunk = (a.cpu() for a in training.critical_points(ws, B, P, k))
    masks = {n: (training.saved_activation(ws, B, P, k, n) > 0).cpu() for n in training.SAVED}
    training.train_backward(m, xc, logp, dlogp, None, ws, lambda p: grads[id(p)])
    assert int(arg_stn.min()) >= 0 and int(arg_stn.max()) < P and int(arg_trunk.min()) >= 0 and int(arg_trunk.max()) < P

    o = po.build_cls(P, k, sd, eval_mode=False).double()
    ref_logp, kinks, pools = forced_forward(o, x.double(), arg_stn, arg_trunk, masks)
    for chosen, true_max in pools:                          # the chosen row holds the maximum
        assert float((true_max - chosen).detach().max()) < 2e-5 * float(true_max.detach().abs().max())
    flipped = 0
    for pre, mask in kinks:                                 # a mask differs from the float64 sign only at ~0
        pre = pre.detach()
        wrong = (pre > 0) != mask
        flipped += int(wrong.sum())
        if wrong.any():
            assert float(pre[wrong].abs().max()) < 2e-5 * float(pre.abs().max())
    print(f"P={P} B={B}: {flipped} ReLU decisions differ from the float64 forward")
    assert (logp.cpu().double() - ref_logp).abs().max().item() < 5e-5
    F.nll_loss(ref_logp, target).backward()
    assert abs(loss.item() - F.nll_loss(ref_logp, target).item()) < 2e-5
    bad = []
    for n, p in m.named_parameters():
        r = o.get_parameter(n).grad.numpy()
        g = grads[id(p)].cpu().numpy()
        if np.abs(r).max() < 1e-7:
            if np.abs(g).max() >= 1e-5:
                bad.append((n, float(np.abs(g).max())))
        elif rel_err(g, r) >= 1e-4:
            bad.append((n, float(rel_err(g, r))))
    assert not bad, bad


def test_config4_trainer_step_at_full_size():
    """One fused Trainer.step at the training configuration (B = 256 per GPU, P = 750) against the float32 oracle
    step from the same state: loss, log-probabilities, running statistics of all ten BatchNorm layers and the batch
    counters.  (Gradients at this size are compared in test_gradients_with_the_kernels_own_decisions[750-256-...]:
    with 10^8 max / ReLU decisions per step a plain float32 comparison of gradients tests luck.)  Post-Adam
    parameters are compared where the oracle's gradient is clear of zero AND the two float32 gradients agree in
    sign and to 1 %: there Adam's first update is lr * g / (|g| + eps), the same number for both."""
    P, k, seed, B, gain = 750, 2, 12, 256, 1.3
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal"))
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64))
    sd = synth.torch_state_dict(seed, k, 1.0, gain)
    oracle = po.build_cls(P, k, sd, eval_mode=False)
    opt = torch.optim.Adam(oracle.parameters(), lr=0.005)
    out, _ = oracle(x)
    loss_ref = F.nll_loss(out, target)
    loss_ref.backward()
    g_ref = {n: p.grad.clone() for n, p in oracle.named_parameters()}
    opt.step()
    m = hip_train_model(P, k, seed, gain)
    tr = Trainer(m, lr=0.005)
    loss, logp = tr.step(x.cuda(), target.cuda())
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    np.testing.assert_allclose(logp.cpu().numpy(), out.detach().numpy(), atol=1e-4)
    ref_sd, got_sd = oracle.state_dict(), m.state_dict()
    for n in ref_sd:
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(got_sd[n].cpu().numpy(), ref_sd[n].numpy(), rtol=2e-4, atol=2e-5, err_msg=n)
        elif n.endswith("num_batches_tracked"):
            assert int(got_sd[n]) == int(ref_sd[n]) == 1
    off, compared = 0, 0
    for n, p in m.named_parameters():
        g = tr.flat_g[off:off + p.numel()].cpu().view_as(p)
        off += p.numel()
        r = g_ref[n]
        clear = (r.abs() > 1e-4) & ((g - r).abs() <= 0.01 * r.abs())
        if clear.any():
            d = (got_sd[n].cpu() - ref_sd[n]).abs()[clear]
            assert float(d.max()) < 1e-4, (n, float(d.max()))
            compared += int(clear.sum())
    assert compared > 100_000


def test_gram_on_bf16_pieces_agrees_with_the_float64_gram(monkeypatch):
    """The statistics of the two conv3 BatchNorms come from a 128x128 Gram matrix.  Its default kernel multiplies
    three-piece bfloat16 splits on the bf16 matrix pipe and adds float32 partial sums of at most 64 rows into double;
    PNGPD_TRAIN_GRAM=f64 selects the float64 matrix instruction.  Both from the same state on the same batch: running
    statistics equal to 1e-6 of their largest entry (float32 storage rounds at 6e-8), log-probabilities to 2e-5."""
    B, P, k = 48, 750, 2
    x = torch.from_numpy(synth.candidate_points(77, B, P)).cuda()
    out = {}
    for variant in ("f64", "x6"):
        monkeypatch.setenv("PNGPD_TRAIN_GRAM", variant)
        m = PointNetCls(P, 3, k)
        m.load_state_dict(synth.torch_state_dict(21, k, 0.02), strict=True)
        m = m.cuda().train()
        with torch.no_grad():
            logp, _ = m(x)
        torch.cuda.synchronize()
        out[variant] = (logp.double().cpu(), {n: b.detach().double().cpu() for n, b in m.named_buffers() if n.endswith(("bn3.running_mean", "bn3.running_var"))})
    assert len(out["x6"][1]) == 4
    for name, ref in out["f64"][1].items():
        got = out["x6"][1][name]
        rel = ((got - ref).abs().max() / ref.abs().max()).item()
        print(f"{name}: max difference relative to the largest entry {rel:.2e}")
        assert rel <= 1e-6, (name, rel)
    assert (out["x6"][0] - out["f64"][0]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("act_scale,w_scale", [(1.0, 1.0), (3.0e3, 2.0e-4), (1.0e-3, 5.0e2), (6.0e4, 1.0)])
def test_conv3_on_float16_pieces_agrees_with_the_other_products(monkeypatch, act_scale, w_scale):
    """The conv3 product of the training step runs, by default, as three products of float16 pieces with both operands
    scaled by powers of two into float16's range (conv3_pool_train_ws_kernel); PNGPD_TRAIN_CONV3=x6 selects six
    products of bfloat16 pieces, =f32 the float32 matrix instruction.  All three from the same state on the same
    batch, with conv2's BatchNorm scale and conv3's weights moved by `act_scale` / `w_scale` so that activations up to
    several 1e5 and weight rows down to 1e-6 pass through the scaling: the STN's pooled maxima agree to 2e-6 of their
    largest entry (the float32 rounding of a 128-term sum), the trunk's to 2e-5, log-probabilities and transform to 2e-5."""
    from pointnetgpd_amd import training
    B, P, k = 24, 750, 2
    x = torch.from_numpy(synth.candidate_points(55, B, P)).cuda()
    sd = synth.torch_state_dict(31, k, 0.02)
    for pre in ("feat.stn.", "feat."):
        sd[pre + "bn2.weight"] = sd[pre + "bn2.weight"] * act_scale
        sd[pre + "bn2.bias"] = sd[pre + "bn2.bias"] * act_scale
        sd[pre + "conv3.weight"] = sd[pre + "conv3.weight"] * w_scale
    out = {}
    for variant in ("f32", "x6", "f16"):
        if variant == "f16":
            monkeypatch.delenv("PNGPD_TRAIN_CONV3", raising=False)       # the default
        else:
            monkeypatch.setenv("PNGPD_TRAIN_CONV3", variant)
        m = PointNetCls(P, 3, k)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        logp, trans, ws, _ = training.train_forward(m, x)
        torch.cuda.synchronize()
        out[variant] = {"logp": logp.double().cpu(), "trans": trans.double().cpu()}
        for name in ("stn_pooled", "trunk_pooled"):
            out[variant][name] = training.saved_activation(ws, B, P, k, name).double().cpu().clone()
        assert torch.isfinite(out[variant]["logp"]).all(), variant
    for variant in ("x6", "f16"):
        for name in ("stn_pooled", "trunk_pooled"):
            ref, got = out["f32"][name], out[variant][name]
            rel = ((got - ref).abs().max() / ref.abs().max()).item()
            print(f"{variant} {name}: max difference relative to the largest entry {rel:.2e}")
            # the trunk's maxima also carry the difference of the 3x3 transform, itself the output of the STN's maxima
            # through two train-mode BatchNorms over 24 rows
            assert rel <= (2e-6 if name == "stn_pooled" else 2e-5), (variant, name, rel)
        assert (out[variant]["logp"] - out["f32"]["logp"]).abs().max().item() <= 2e-5
        assert (out[variant]["trans"] - out["f32"]["trans"]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("P,B,k,seed", [(750, 24, 2, 21), (37, 150, 2, 22), (5, 16, 3, 23), (129, 3, 2, 24)])
def test_backward_of_the_per_point_layers_in_one_pass_agrees_with_the_kernel_per_step_form(monkeypatch, P, B, k, seed):
    """By default the backward of conv2 / bn2 / relu and the column sums of bn1's backward run as one pass over dh2
    (conv2_bwd_fused_kernel) and the forward of conv1 / bn1 / relu / conv2 as one pass over the points
    (conv12_fwd_kernel); PNGPD_TRAIN_BWD2=split and PNGPD_TRAIN_FWD12=split run a kernel per step.  Same state, same batch, same forward:
    both forms take identical ReLU decisions (the mask is the sign of the same recomputed expression), so every
    gradient agrees to float32 rounding of its sums - 2e-5 of the gradient's largest entry."""
    from pointnetgpd_amd import training
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal")).cuda()
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64)).cuda()
    got, fwd = {}, {}
    # both forms with their products on the float32 matrix instruction: this test is about the fusion; the split-bfloat16
    # products of the fused form have a test of their own below
    monkeypatch.setenv("PNGPD_TRAIN_DGRAD", "f32")
    for form in ("split", "fused"):
        if form == "split":
            monkeypatch.setenv("PNGPD_TRAIN_BWD2", "split")
            monkeypatch.setenv("PNGPD_TRAIN_FWD12", "split")
        else:
            monkeypatch.delenv("PNGPD_TRAIN_BWD2", raising=False)
            monkeypatch.delenv("PNGPD_TRAIN_FWD12", raising=False)
        m = hip_train_model(P, k, seed, 1.3)
        logp, trans, ws, xc = training.train_forward(m, x)
        fwd[form] = [logp.cpu().clone(), trans.cpu().clone()] + [
            training.saved_activation(ws, B, P, k, n).cpu().clone() for n in training.SAVED]
        fwd[form] += [b.detach().cpu().clone() for b in m.buffers()]
        loss, dlogp = nll_loss(logp, target)
        grads = {n: torch.zeros_like(p) for n, p in m.named_parameters()}
        by_id = {id(p): grads[n] for n, p in m.named_parameters()}
        training.train_backward(m, xc, logp, dlogp, None, ws, lambda p: by_id[id(p)])
        torch.cuda.synchronize()
        got[form] = {n: g.double().cpu() for n, g in grads.items()}
    # the forward of conv1 / bn1 / relu / conv2 in one pass (conv12_fwd_kernel) runs the same operations in the same
    # order as the two-kernel form: outputs, saved activations and running statistics are equal bit for bit
    for a, b in zip(fwd["split"], fwd["fused"]):
        assert torch.equal(a, b)
    worst = ("", 0.0)
    for n, ref in got["split"].items():
        g = got["fused"][n]
        assert torch.isfinite(g).all(), n
        scale = float(ref.abs().max())
        err = float((g - ref).abs().max())
        if scale < 1e-7:
            assert err < 1e-6, (n, err)
            continue
        if err / scale > worst[1]:
            worst = (n, err / scale)
        assert err <= 2e-5 * scale, (n, err / scale)
    print(f"P={P} B={B}: largest difference {worst[1]:.2e} of the gradient's largest entry ({worst[0]})")


@pytest.mark.parametrize("P,B,k,seed", [(750, 24, 2, 31), (37, 150, 3, 32)])
def test_dense_input_gradient_on_split_bfloat16_agrees_with_the_float32_product(monkeypatch, P, B, k, seed):
    """The two tall products of a stack's backward that make gradient terms - the dense part of conv3's input gradient
    (h2 times a 128x128 matrix, dense_dgrad_x3_kernel) and dh1 = dy2 W2 inside conv2_bwd_fused_kernel - run as three
    products of bfloat16 pieces per float32 product (16 significand bits per operand); PNGPD_TRAIN_DGRAD=f32 keeps
    them on the float32 matrix instruction.  Same state, same batch, same forward: every gradient of the two forms
    agrees to 1e-4 of its largest entry (measured: below 1e-5)."""
    from pointnetgpd_amd import training
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "normal")).cuda()
    target = torch.from_numpy((synth.uniform(seed, "labels", (B,)) * k).astype(np.int64)).cuda()
    got = {}
    for form in ("f32", "x3"):
        if form == "f32":
            monkeypatch.setenv("PNGPD_TRAIN_DGRAD", "f32")
        else:
            monkeypatch.delenv("PNGPD_TRAIN_DGRAD", raising=False)
        m = hip_train_model(P, k, seed, 1.3)
        logp, trans, ws, xc = training.train_forward(m, x)
        loss, dlogp = nll_loss(logp, target)
        grads = {n: torch.zeros_like(p) for n, p in m.named_parameters()}
        by_id = {id(p): grads[n] for n, p in m.named_parameters()}
        training.train_backward(m, xc, logp, dlogp, None, ws, lambda p: by_id[id(p)])
        torch.cuda.synchronize()
        got[form] = {n: g.double().cpu() for n, g in grads.items()}
    worst = ("", 0.0)
    for n, ref in got["f32"].items():
        g = got["x3"][n]
        assert torch.isfinite(g).all(), n
        scale = float(ref.abs().max())
        err = float((g - ref).abs().max())
        if scale < 1e-7:
            assert err < 1e-6, (n, err)
            continue
        if err / scale > worst[1]:
            worst = (n, err / scale)
        assert err <= 1e-4 * scale, (n, err / scale)
    print(f"P={P} B={B}: largest difference {worst[1]:.2e} of the gradient's largest entry ({worst[0]})")
