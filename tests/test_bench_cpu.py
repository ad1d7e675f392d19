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
