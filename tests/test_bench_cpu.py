"""The CPU-baseline legs of bench.py run (tiny samples) and report the fields the bench line promises."""
import importlib.util
import os

import numpy as np

from pointnetgpd_amd import crop as hc
from pointnetgpd_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

FIELDS = {"value", "unit", "cores", "kind", "sample"}


def test_scoring_baseline():
    out = bench.cpu_baseline(64, 2, 0, seconds_budget=0.2)
    assert set(out) == FIELDS and out["value"] > 0 and out["kind"] == "port" and out["unit"] == "candidates/s"


def test_training_baseline():
    out = bench.cpu_baseline_train(32, 2, seconds_budget=0.2)
    assert set(out) == FIELDS and out["value"] > 0 and out["unit"] == "samples/s"


def test_online_baseline():
    scene = synth.scene_points(0, 20_000, 0.2)
    grasps = synth.random_grasps(1, 6, 0.2, 0.03)
    limits = hc.box_limits_from_gripper(0.06, 0.12, 0.02, 0.02)
    out = bench.cpu_baseline_online(scene, grasps, np.asarray(limits), 64, 2, 3, n_grasps=6)
    assert set(out) == FIELDS and out["value"] > 0 and out["unit"] == "grasps/s"


def test_strong_scaling_switch_cuts_the_global_batch_exactly():
    """--global-batch: the ranks' shares are contiguous, differ by at most one and add up to the total; without it
    every rank scores --batch candidates (weak scaling)."""
    import types
    for total, world in ((65_536, 8), (65_536, 3), (10, 4), (7, 8)):
        args = types.SimpleNamespace(global_batch=total, batch=4096)
        shares = [bench.per_rank_batch(args, r, world) for r in range(world)]
        assert sum(shares) == total and max(shares) - min(shares) <= 1
    args = types.SimpleNamespace(global_batch=0, batch=4096)
    assert [bench.per_rank_batch(args, r, 4) for r in range(4)] == [4096] * 4
    assert bench.group_info(None) == {"backend": None, "ranks": 1}


def test_extra_legs_keep_value_time_and_roofline():
    line = {"metric": "m", "value": 1.0, "unit": "u", "ms_per_step": 2.0, "steps": 3, "warmup": 1, "dtype": "bf16",
            "scaling": "weak", "config": {"workload": "w", "crop_ms": 0.3, "loss": 0.1, "exposed_allreduce_wait_ms": None},
            "roofline": {"bound": "mfma", "kernel": "k", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5,
                         "traffic": None, "kernel_ms": 0.4, "flop_per_launch": 1.0}}
    b = bench._brief(line)
    assert b["value"] == 1.0 and b["ms_per_step"] == 2.0 and b["roofline"]["frac"] == 0.5 and b["roofline"]["kernel_ms"] == 0.4
    assert b["crop_ms"] == 0.3 and b["loss"] == 0.1 and "exposed_allreduce_wait_ms" in b and bench._brief(None) is None
    assert bench.PEAK_TFLOPS["bf16x3"] == round(bench.PEAK_TFLOPS["bf16"] / 3.0, 1)
