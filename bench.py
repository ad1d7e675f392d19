#!/usr/bin/env python3
"""Headline benchmark: grasp candidates scored per second at P=750 on N MI355X.

One "step" = one full eval forward of PointNetCls(750,3,2) (STN stack, STN regressor, trunk
stack, classifier head, log-softmax) over a batch of 4096 synthetic candidates per GPU that are
already resident in HBM.  Candidates are sharded data-parallel: every rank scores its own 4096
(weak scaling), no collective is on the data path.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the fused bf16 conv stack)
from a live HIP-event measurement; `cpu_baseline` times the CPU oracle (the reference's op
sequence in eager PyTorch) on the host cores of the same box.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# the host driver supports dmabuf IPC only: RCCL across processes fails without this (set before the GPU is touched)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointnetgpd_amd import _lib, synth  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402

# algorithmic work, BN folded, 1 MAC = 2 flop (bias / ReLU / max / softmax not counted)
FLOP_PER_POINT_STACK = 2 * (3 * 64 + 64 * 128 + 128 * 1024)          # 278,912
FLOP_FC_STN = 2 * (1024 * 512 + 512 * 256 + 256 * 9)
# dense MFMA peaks, MI355X.  bf16x3 computes every product as three bf16 piece products, so what the matrix pipe can
# deliver of such products is the dense bf16 peak / 3 (`achieved` stays algorithmic flops over time).
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": round(2500.0 / 3.0, 1)}
PEAK_NOTE = {"bf16x3": " (float32-grade products as 3 bf16 piece products; peak = dense bf16 peak / 3)"}


def flop_per_candidate(P, k):
    return P * (2 * FLOP_PER_POINT_STACK + 18) + FLOP_FC_STN + 2 * (1024 * 512 + 512 * 256 + 256 * k)


def device_points(seed, B, P, rank, device, first=None):
    """Candidate points generated on the device: uniform in the synthetic gripper box, keyed by
    (seed, global candidate id) so that the data of a candidate does not depend on the sharding.
    `first` = global id of this rank's first candidate (default: rank * B)."""
    g = torch.Generator(device=device)
    out = torch.empty((B, 3, P), dtype=torch.float32, device=device)
    hd, ow, hh = synth.GRIPPER_BOX
    scale = torch.tensor([hd, ow, hh], device=device).view(1, 3, 1)
    shift = torch.tensor([0.0, -ow / 2, -hh / 2], device=device).view(1, 3, 1)
    chunk = 1024
    for c0 in range(0, B, chunk):
        g.manual_seed(seed * 1_000_003 + ((rank * B if first is None else first) + c0))
        n = min(chunk, B - c0)
        out[c0:c0 + n] = torch.rand((n, 3, P), generator=g, device=device) * scale + shift
    return out


def cpu_baseline(P, k, seed, seconds_budget=15.0):
    """Oracle (eager fp32 PyTorch on the host CPU), B=64 batches - the reference's batched eval shape."""
    from oracle import pointnet_oracle as po
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    m = po.build_cls(P, k, synth.torch_state_dict(seed, k, 0.02))
    x = torch.from_numpy(synth.candidate_points(seed, 64, P, "gripper"))
    with torch.no_grad():
        m(x)
        t0 = time.perf_counter()
        m(x)
        one = time.perf_counter() - t0
        iters = int(max(3, min(40, seconds_budget / max(one, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(iters):
            m(x)
        dt = time.perf_counter() - t0
    return {"value": round(64 * iters / dt, 2), "unit": "candidates/s", "cores": threads, "kind": "port",
            "sample": f"oracle PointNetCls({P},3,{k}).eval() fp32, {iters} batches of 64 candidates, "
                      f"torch {torch.__version__} CPU"}


def cpu_baseline_train(P, k, seconds_budget=15.0):
    """Oracle training step (eager fp32 PyTorch: forward, nll_loss, backward, Adam) on the host CPU."""
    from oracle import pointnet_oracle as po
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    Bc = 64
    m = po.build_cls(P, k, synth.torch_state_dict(0, k, 0.02, 1.0), eval_mode=False)
    opt = torch.optim.Adam(m.parameters(), lr=0.005)
    x = torch.from_numpy(synth.candidate_points(0, Bc, P, "gripper"))
    target = torch.from_numpy((synth.uniform(0, "bench/labels", (Bc,)) < 0.5).astype("int64"))

    def one():
        opt.zero_grad()
        out, _ = m(x)
        torch.nn.functional.nll_loss(out, target).backward()
        opt.step()

    one()
    t0 = time.perf_counter()
    one()
    dt1 = time.perf_counter() - t0
    iters = int(max(2, min(20, seconds_budget / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(Bc * iters / dt, 2), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"oracle PointNetCls({P},3,{k}).train(): {iters} steps of batch {Bc} (forward, nll_loss, backward, "
                      f"Adam), torch {torch.__version__} CPU"}


def cpu_baseline_online(scene, grasps, limits, P, k, R, n_grasps=200):
    """The reference's online loop on the host CPU for the first `n_grasps` grasps: numpy float64 crop over the
    whole scene per grasp, then R draws of P points scored one candidate at a time (batch 1)."""
    import numpy as np
    from oracle import crop_oracle as co
    from oracle import pointnet_oracle as po
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    m = po.build_cls(P, k, synth.torch_state_dict(0, k, 0.02))
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    idx, pts = co.crop_numpy(grasps[:n_grasps], scene, limits)
    t_crop = time.perf_counter() - t0
    scored = 0
    t0 = time.perf_counter()
    for pg in pts:
        if len(pg) < 20:
            continue
        for _ in range(R):
            sel = rng.choice(len(pg), P, replace=len(pg) < P)
            po.oracle_test_network(m, pg[sel])
        scored += 1
    t_score = time.perf_counter() - t0
    return {"value": round(n_grasps / (t_crop + t_score), 3), "unit": "grasps/s", "cores": threads, "kind": "port",
            "sample": f"first {n_grasps} grasps on the full {len(scene)}-point scene: numpy crop {t_crop:.2f} s + "
                      f"{scored}x{R} batch-1 oracle forwards {t_score:.2f} s, torch {torch.__version__} CPU"}


def online_workload(args, dev, rank, n_gpus, dist):
    """Config 5: 1M-point scene x 10k grasps per GPU: crop + subsample (3 repeats) + scoring + vote."""
    from pointnetgpd_amd import crop as hc
    from pointnetgpd_amd.online import score_grasps

    S, G, P, k, R = args.scene_points, args.grasps, args.points, args.classes, 3
    limits = hc.box_limits_from_gripper(synth.GRIPPER_BOX[0], synth.GRIPPER_BOX[1] + 0.04, 0.02, synth.GRIPPER_BOX[2])
    scene = torch.from_numpy(synth.scene_points(0, S, 0.5)).to(dev)
    grasps = synth.random_grasps(1000 + rank, G, 0.5, 0.06)
    model = PointNetCls(P, 3, k)
    model.load_state_dict(synth.torch_state_dict(0, k, 0.02), strict=True)
    model = model.to(dev).eval()
    model.precision = args.precision
    grasps_dev = torch.from_numpy(grasps).to(dev)
    # synthetic weights call every grasp bad: centre the classifier so that the vote, the good-grasp filter and the
    # ranking have real work (about half of the grasps come out good)
    synth.balance_head(model, hc.crop_subsample(grasps_dev[:512], scene, limits, P, repeat=1, seed=1)[0])

    def step():
        return score_grasps(model, scene, grasps_dev, limits, repeat=R, seed=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    # stage split with events on the launch stream: crop + subsample in one pass, then scoring and vote
    # (mean of 20 passes queued back to back: one pass alone, started on an idle device, mostly measures its seven
    # launches)
    from pointnetgpd_amd.scoring import score_candidates
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 20
    e[0].record()
    for _ in range(reps):
        pts, valid, counts, overflow = hc.crop_subsample(grasps_dev, scene, limits, P, repeat=R, seed=0)
    e[1].record()
    for _ in range(reps):
        pred, probs, _ = score_candidates(model, pts)
        hc.vote(pred, probs, valid, R)
    e[2].record()
    torch.cuda.synchronize(dev)
    assert int(overflow.item()) == 0
    crop_ms, score_ms = e[0].elapsed_time(e[1]) / reps, e[1].elapsed_time(e[2]) / reps
    total_idx = int(counts.sum().item())
    # algorithmic bytes of the stage: scene once, grasps, counts, and the network input it writes
    crop_bytes = S * 12 + G * 15 * 8 + G * 4 + G * R * P * 3 * 4
    # HBM bytes of the stage per frame from the committed PMC passes, when they ran this configuration
    crop_traffic = None
    tpath = os.path.join(ROOT, "profiles", "r2_crop_stage_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t["config"] == {"S": S, "G": G, "R": R, "P": P}:
            crop_traffic = t["hbm_bytes_per_frame"]
    if rank == 0:
        crop_gbps = crop_bytes / (crop_ms * 1e-3) / 1e9
        line = {
            "metric": "grasps scored/sec, online crop + scoring", "value": round(n_gpus * G * args.steps / elapsed, 1),
            "unit": "grasps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 crop + " + args.precision, "data": "synthetic",
            "config": {"workload": f"scene {S} pts x {G} grasps per GPU, crop+subsample x{R}+PointNetCls P={P}+vote",
                       "crop_ms": round(crop_ms, 3), "score_ms": round(score_ms, 3), "pair_tests_per_s": round(S * G / (crop_ms * 1e-3), 1),
                       "crop_algorithmic_GBps": round(crop_bytes / (crop_ms * 1e-3) / 1e9, 2),
                       "mean_points_per_grasp": round(total_idx / G, 1), "good_grasps": int(out.order.numel()),
                       "process_group": group_info(dist)},
            # the crop + subsample stage against the HBM roofline (its main kernel is bound by vector-instruction
            # issue, profiles/r2_crop_stage.md); the stage that dominates the frame is the scoring, whose kernel the
            # score workload reports
            "roofline": {"bound": "hbm", "kernel": "crop + subsample (grid build and k_cull_gather)",
                         "achieved": round(crop_gbps, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(crop_gbps / 8000.0, 5), "traffic": crop_traffic, "stage_ms": round(crop_ms, 3),
                         "bytes_per_frame": crop_bytes},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_online(scene.cpu().numpy(), grasps, limits, P, k, R)
        return line
    return None


def train_workload(args, dev, rank, n_gpus, dist):
    """Config 4: training step (fwd + bwd, NLL, Adam), P=750, batch 256 per GPU, three gradient buckets all-reduced."""
    from pointnetgpd_amd.training import Trainer

    B, P, k = args.train_batch, args.points, args.classes
    model = PointNetCls(P, 3, k)
    model.load_state_dict(synth.torch_state_dict(0, k, 0.02, 1.0), strict=True)
    model = model.to(dev).train()
    trainer = Trainer(model, lr=0.005)
    x = device_points(0, B, P, rank, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(rank)
    target = (torch.rand(B, generator=g, device=dev) < 0.5).to(torch.int64)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        trainer.step(x, target)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.step(x, target)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    # dominant kernel of the step: the fused conv3 product / max / arg-max kernel, timed with HIP events on the
    # launch stream inside 20 further full steps (two launches per step)
    trainer.conv3_ms = []
    trainer.wait_events = []          # HIP events around the waits for the gradient all-reduces (N > 1 only)
    for _ in range(20):
        trainer.step(x, target)
    torch.cuda.synchronize(dev)
    conv3_ms = sum(trainer.conv3_ms) / len(trainer.conv3_ms)
    trainer.conv3_ms = None
    # how long the compute stream stood still for the collectives: what of the three bucket all-reduces is NOT hidden
    # behind the backward kernels (the last bucket, mostly)
    waits = [a.elapsed_time(b) for a, b in trainer.wait_events]
    trainer.wait_events = None
    exposed_ms = max_over_ranks(dist, sum(waits) / len(waits), dev) if waits else None
    if rank == 0:
        conv3_flop = 2.0 * 128 * 1024 * B * P
        conv3_tf = conv3_flop / (conv3_ms * 1e-3) / 1e12
        # flop of the step as the reference executes it (3 x forward: forward, data and weight gradients); the HIP
        # step does less arithmetic than that (low-rank dense backward of conv3), so this is an equivalent rate
        flop = 3.0 * B * flop_per_candidate(P, k)
        line = {
            "metric": "training samples/sec (fwd+bwd+NLL+Adam)", "value": round(n_gpus * B * args.steps / elapsed, 1),
            "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": {"workload": f"PointNetCls {k}-class train step, P={P}, batch {B} per GPU, Adam lr 0.005, "
                                   "fp32 gradient all-reduced in three buckets over RCCL",
                       "loss": round(float(loss.item()), 5), "reference_equivalent_tflops": round(flop * args.steps / elapsed / 1e12, 2),
                       "process_group": group_info(dist),
                       "exposed_allreduce_wait_ms": None if exposed_ms is None else round(exposed_ms, 4)},
            "roofline": _train_roofline(conv3_tf, conv3_ms, conv3_flop, B, P),
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_train(P, k)
        return line
    return None


def _train_roofline(conv3_tf, conv3_ms, conv3_flop, B, P):
    """Roofline entry of the training step's dominant kernel, the fused conv3 product / max / arg-max.  Its default
    form computes float32-grade products on the 16-bit matrix pipe: every operand is split into two float16 pieces
    (power-of-two scaled into float16's range) and three piece products are accumulated, so a launch executes 3 x the
    algorithmic flops on v_mfma_f32_16x16x32_f16.  `achieved` is algorithmic flops over time, as for every other line;
    `peak` is what the pipe can deliver of such products, the dense 16-bit peak / 3.  PNGPD_TRAIN_CONV3=x6 selects six
    products of bfloat16 pieces (peak / 6), =f32 the float32 matrix instruction (peak 157.3)."""
    mode = os.environ.get("PNGPD_TRAIN_CONV3", "")
    if mode.startswith("f"):
        peak, kernel, mult = PEAK_TFLOPS["fp32"], "conv3_pool_train_kernel (v_mfma_f32_32x32x2_f32)", None
    elif mode.startswith("x"):
        peak, mult = round(PEAK_TFLOPS["bf16"] / 6.0, 1), 6.0
        kernel = ("conv3_pool_train_x6_kernel (float32-grade products as 6 bf16 piece products on "
                  "v_mfma_f32_32x32x16_bf16; peak = dense bf16 peak / 6)")
    else:
        peak, mult = round(PEAK_TFLOPS["bf16"] / 3.0, 1), 3.0
        kernel = ("conv3_pool_train_ws_kernel (float32-grade products as 3 float16 piece products on "
                  "v_mfma_f32_16x16x32_f16, weights stationary; peak = dense 16-bit peak / 3)")
    # HBM bytes per launch from the committed PMC passes, when they ran this kernel at this size
    traffic = None
    tname = "r3_train_conv3_ws_traffic.json" if mult == 3.0 else "r2_train_conv3_traffic.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and mult is not None:
        t = json.load(open(tpath))
        if t["config"] == {"B": B, "P": P}:
            traffic = t["hbm_bytes_per_launch"]
    return {"bound": "mfma", "kernel": kernel, "achieved": round(conv3_tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(conv3_tf / peak, 4), "traffic": traffic, "kernel_ms": round(conv3_ms, 4),
            "flop_per_launch": conv3_flop, "executed_16bit_tflops": round(mult * conv3_tf, 1) if mult else None}


def _voxel_traffic(S, n):
    tpath = os.path.join(ROOT, "profiles", "r3_voxel_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t["config"] == {"S": S, "n": n}:
            return t["hbm_bytes_per_call"]
    return None


def voxel_workload(args, dev, rank, n_gpus, dist):
    """Scene pre-processing in front of the crop: voxel down-sampling of a 1M-point scene at n=500."""
    from pointnetgpd_amd import voxel as hv

    S, n = args.scene_points, args.voxel_n
    scene = torch.from_numpy(synth.scene_points(rank, S, 0.5)).to(dev)
    # the reference's call shape (exact-size result: one host read of the count per call), for the record
    for _ in range(args.warmup):
        out = hv.get_voxel_fun(scene, n)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = hv.get_voxel_fun(scene, n)
    torch.cuda.synchronize(dev)
    with_host_read_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    M = out.shape[0]
    # timed: the same kernels with the count left on the device (what a frame loop queues in front of the crop)
    buf = torch.empty((S, 3), dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    for _ in range(args.warmup):
        hv.voxel_unique_device(scene, n, out=buf, count=cnt)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hv.voxel_unique_device(scene, n, out=buf, count=cnt)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    assert int(cnt.item()) == M
    bitmap = n ** 3 // 8
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        gbps = (S * 12 + M * 12 + 3 * bitmap) / (ms * 1e-3) / 1e9
        line = {
            "metric": "scene points voxel-down-sampled/sec", "value": round(n_gpus * S * args.steps / elapsed, 1),
            "unit": "points/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points, " + ("f32" if hv.default_edges() == "float32" else "f64") + " edges",
            "data": "synthetic",
            "config": {"workload": f"get_voxel_fun: {S} points, n={n}, {M} unique centres, count left on the device",
                       "ms_per_call_with_host_read_of_count": round(with_host_read_ms, 4),
                       "algorithmic_GBps": round((S * 12 + M * 12) / (ms * 1e-3) / 1e9, 2),
                       "bitmap_GBps": round(gbps, 2)},
            # whole call (five kernels) against the HBM roofline; bytes = points in (twice: bounds, marking), centres
            # out, and the occupancy bitmap written, counted and expanded; traffic = HBM bytes per call from the
            # committed counter passes, when they ran this size
            "roofline": {"bound": "hbm", "kernel": "get_voxel_fun (all kernels of the call)", "achieved": round(gbps, 2),
                         "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
                         "traffic": _voxel_traffic(S, n), "bytes_per_call": S * 12 + M * 12 + 3 * bitmap},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import voxel_oracle as vo
            pts = scene.cpu().numpy()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                vo.get_voxel_fun(pts, n)
            dt = (time.perf_counter() - t0) / reps
            line["cpu_baseline"] = {"value": round(S / dt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                                    "sample": f"numpy restatement of get_voxel_fun on the same {S} points, n={n}, {reps} calls"}
        return line
    return None


def per_rank_batch(args, rank, n_gpus):
    """Candidates this rank scores per step: --batch each (weak scaling), or its contiguous share of --global-batch
    (strong scaling: the total is fixed, dist.shard_range cuts it)."""
    if args.global_batch:
        from pointnetgpd_amd.dist import shard_range
        lo, hi = shard_range(args.global_batch, rank, n_gpus)
        return hi - lo
    return args.batch


def group_info(dist):
    """What the ranks are connected by: backend and rank count of the process group (RCCL = torch's "nccl")."""
    if dist is None:
        return {"backend": None, "ranks": 1}
    return {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "ranks": dist.get_world_size()}


def score_workload(args, dev, rank, n_gpus, dist, B, P, precision, steps, warmup, full):
    """Batch scoring (configs 2 and 3): `steps` eval forwards of PointNetCls over B candidates per rank that are
    resident in HBM.  full=False is the short form used for the extra legs of the default run: no PCIe-inclusive
    rate, no CPU baseline, fewer timed conv-stack launches."""
    k, seed = args.classes, 0
    model = PointNetCls(P, 3, k)
    model.load_state_dict(synth.torch_state_dict(seed, k, 0.02), strict=True)
    model = model.to(dev).eval()
    model.precision = precision
    first = 0
    if args.global_batch:
        from pointnetgpd_amd.dist import shard_range
        first = shard_range(args.global_batch, rank, n_gpus)[0]
    x = device_points(seed, B, P, rank, dev, first=first if args.global_batch else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The matrix-core clock takes ~30 ms of sustained load to settle after an idle period (launch
    # durations of one kernel drift by 20 % over the first 40 launches), so a short --warmup is
    # topped up to SETTLE_STEPS untimed steps; the timed region is exactly `steps` steps.
    settle = max(warmup, SETTLE_STEPS) if full else warmup
    for _ in range(settle):
        model(x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        logp, trans = model(x)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    assert torch.isfinite(logp).all()

    pcie_rate = None
    if full:
        # the same step when the candidates start in (pinned) host memory: one H2D copy of x per step, not
        # overlapped with compute.  Reported beside the headline, never as `value`.
        x_host = x.cpu().pin_memory()
        for _ in range(5):
            x.copy_(x_host, non_blocking=True)
            model(x)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        host_steps = 50
        for _ in range(host_steps):
            x.copy_(x_host, non_blocking=True)
            model(x)
        torch.cuda.synchronize(dev)
        pcie_rate = B * host_steps / (time.perf_counter() - t1)

    # dominant kernel: the fused conv stack, timed inside full scoring steps: every step brackets its two
    # conv-stack launches (STN, trunk) with HIP events on the launch stream (pngpd_forward_infer_timed)
    h = model._handle(dev)
    ws = h.workspace(B)
    stream = torch.cuda.current_stream(dev).cuda_stream
    prec = _lib.PRECISIONS[precision]
    logp_t = torch.empty((B, k), dtype=torch.float32, device=dev)
    trans_t = torch.empty((B, 9), dtype=torch.float32, device=dev)
    pair = (ctypes.c_float * 2)()
    conv_ms = []
    skip, timed = (10, 50) if full else (3, 10)
    for i in range(skip + timed):
        _lib.check(h.lib.pngpd_forward_infer_timed(h.ptr, ctypes.c_void_p(x.data_ptr()), _lib.LAYOUT_C3P, B, P, prec,
                                                   ctypes.c_void_p(logp_t.data_ptr()), ctypes.c_void_p(trans_t.data_ptr()),
                                                   None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), pair,
                                                   ctypes.c_void_p(stream)))
        if i >= skip:
            conv_ms += [pair[0], pair[1]]
    kernel_ms = sum(conv_ms) / len(conv_ms)          # mean duration of one conv-stack launch
    flop_launch = float(B) * P * FLOP_PER_POINT_STACK
    achieved = flop_launch / (kernel_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    # HBM bytes per launch of that kernel come from PMC passes (FETCH_SIZE / WRITE_SIZE, corrected for
    # gfx950) that cannot run inside this process; the committed measurement is reported when it is
    # for this exact configuration, else null.
    traffic = None
    for name in ("r3_conv_stack_traffic.json", "r3_conv_stack_x3_traffic.json", "r2_conv_stack_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if traffic is None and os.path.exists(tpath):
            t = json.load(open(tpath))
            if t["config"] == {"B": B, "P": P, "precision": precision}:
                traffic = t["hbm_bytes_per_launch"]
    if rank != 0:
        return None
    total_per_step = float(args.global_batch) if args.global_batch else float(n_gpus) * B
    total = total_per_step * steps
    fc = {"bf16": "split-bf16 (fp32-grade)", "bf16x3": "split-bf16 (fp32-grade)", "fp32": "fp32"}[precision]
    line = {
        "metric": f"grasp candidates scored/sec at N={P}",
        "value": round(total / elapsed, 1),
        "unit": "candidates/s",
        "n_gpus": n_gpus,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(1e3 * elapsed / steps, 4),
        "higher_is_better": True,
        "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None,
        "dtype": precision,
        "data": "synthetic",
        "config": {"workload": f"PointNetCls {k}-class eval forward, P={P}, batch {B} per GPU"
                               + (f" (global batch {args.global_batch} cut over {n_gpus} ranks)" if args.global_batch else "")
                               + f", {precision} fused conv/max-pool kernels + {fc} FC, synthetic weights",
                   "batch_per_gpu": B, "points": P, "classes": k, "parallelism": f"dp{n_gpus}",
                   "process_group": group_info(dist),
                   "untimed_steps_before_timing": settle,
                   "whole_net_tflops": round(total / elapsed * flop_per_candidate(P, k) / 1e12, 2)},
        "roofline": {"bound": "mfma", "kernel": f"conv_stack_{precision}" + PEAK_NOTE.get(precision, ""),
                     "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                     "kernel_ms": round(kernel_ms, 4), "flop_per_launch": flop_launch},
    }
    if pcie_rate is not None:
        line["config"]["pcie_inclusive_candidates_per_s_per_gpu"] = round(pcie_rate, 1)
    if full and n_gpus == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(P, k, seed)
    return line


def _brief(line):
    """What an extra leg contributes to the headline line: its throughput and its dominant kernel's roofline."""
    if line is None:
        return None
    r = line["roofline"]
    out = {"workload": line["config"]["workload"], "metric": line["metric"], "value": line["value"], "unit": line["unit"],
           "ms_per_step": line["ms_per_step"], "steps": line["steps"], "warmup": line["warmup"], "dtype": line["dtype"],
           "scaling": line["scaling"],
           "roofline": {key: r[key] for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if key in r}}
    for key in ("kernel_ms", "stage_ms"):
        if key in r:
            out["roofline"][key] = r[key]
    for key in ("crop_ms", "score_ms", "loss", "exposed_allreduce_wait_ms"):
        if key in line["config"]:
            out[key] = line["config"][key]
    return out


def other_workloads(args, dev, rank, n_gpus, dist):
    """Short legs of the other configurations, run behind the headline with the same per-GPU sizes on every rank:
    the split-bf16 and exact-f32 modes of the headline batch, P=500 (config 3's per-GPU share at 8 GPUs), the training
    step (config 4) and the online frame (config 5).  Each is timed like the headline (barrier + synchronise on both
    sides, max over ranks); none runs a CPU baseline."""
    import copy
    a = copy.copy(args)
    a.global_batch, a.no_cpu_baseline = 0, True
    legs = {}
    legs["score_bf16x3_b4096_p750"] = _brief(score_workload(a, dev, rank, n_gpus, dist, B=4096, P=750, precision="bf16x3",
                                                            steps=30, warmup=10, full=False))
    legs["score_fp32_b4096_p750"] = _brief(score_workload(a, dev, rank, n_gpus, dist, B=4096, P=750, precision="fp32",
                                                          steps=20, warmup=5, full=False))
    legs["score_bf16_b8192_p500"] = _brief(score_workload(a, dev, rank, n_gpus, dist, B=8192, P=500, precision="bf16",
                                                          steps=30, warmup=10, full=False))
    a.points, a.train_batch, a.steps, a.warmup = 750, 256, 30, 15
    legs["train_b256_p750"] = _brief(train_workload(a, dev, rank, n_gpus, dist))
    a.scene_points, a.grasps, a.steps, a.warmup, a.precision = 1_000_000, 10_000, 10, 3, "bf16"
    legs["online_s1m_g10k"] = _brief(online_workload(a, dev, rank, n_gpus, dist))
    return legs


SETTLE_STEPS = 50


def max_over_ranks(dist, elapsed, dev):
    """MAX of the per-rank wall time of the timed region (device tensor under RCCL, host tensor under gloo)."""
    if dist is None:
        return elapsed
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="candidates per GPU")
    ap.add_argument("--points", type=int, default=750)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="score workload: only the headline, without the short legs of the other configurations")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="score workload: total candidates per step, cut over the ranks (strong scaling; e.g. "
                         "--global-batch 65536 --points 500); default: --batch per GPU (weak scaling)")
    ap.add_argument("--workload", default="score", choices=["score", "online", "train", "voxel"],
                    help="score: headline batch scoring; online: 1M-point scene x 10k grasps crop+score; "
                         "train: training step, batch 256 per GPU")
    ap.add_argument("--train-batch", type=int, default=256)
    ap.add_argument("--scene-points", type=int, default=1_000_000)
    ap.add_argument("--grasps", type=int, default=10_000)
    ap.add_argument("--voxel-n", type=int, default=500)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("PNGPD_BENCH_FORCE_GROUP") == "1":     # the second: a one-rank RCCL rehearsal
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PNGPD_BENCH_BACKEND=gloo rehearses the multi-process flow on a box with fewer GPUs than
        # ranks (ranks then share devices); the driver's runs use RCCL, one rank per GPU.
        backend = os.environ.get("PNGPD_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and local_rank >= ndev:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
        dev_index = local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    n_gpus = world
    if n_gpus != args.gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    if args.workload in ("online", "train", "voxel"):
        line = {"online": online_workload, "train": train_workload, "voxel": voxel_workload}[args.workload](args, dev, rank, n_gpus, dist)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    line = score_workload(args, dev, rank, n_gpus, dist, B=per_rank_batch(args, rank, n_gpus), P=args.points,
                          precision=args.precision, steps=args.steps, warmup=args.warmup, full=True)
    if not args.no_other_workloads:
        others = other_workloads(args, dev, rank, n_gpus, dist)
        if rank == 0:
            line["config"]["other_workloads"] = others
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
