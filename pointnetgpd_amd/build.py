"""In-tree build of the HIP library (gfx950 only) and of the test-side helpers.

Everything is compiled with explicit ``hipcc``/``gcc`` command lines so that the built
``.so`` files sit next to their sources and travel with the tree; nothing is JIT-cached.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointnetgpd_amd", "csrc")
LIB_PATH = os.path.join(ROOT, "pointnetgpd_amd", "libpngpd.so")
PROBE_PATH = os.path.join(ROOT, "tests", "_probe.so")
ORACLE_C_PATH = os.path.join(ROOT, "oracle", "_build", "libcrop_oracle.so")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"

LIB_SOURCES = [
    "api.cpp",
    "pack.cpp",
    "pack_device.hip",
    "launch_util.cpp",
    "conv_stack.hip",
    "conv_stack_x3.hip",
    "fc.hip",
    "crop.hip",
    "crop_grid.hip",
    "voxel.hip",
    "train.hip",
    "train_conv3.hip",
    "train_bwd2.hip",
    "tall_gemm.hip",
    "train_api.cpp",
]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _hip_flags():
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall",
            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def build_library(force: bool = False, verbose_resources: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hdrs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not force and not _newer(LIB_PATH, srcs + hdrs):
        return LIB_PATH
    objs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + _hip_flags()
            if s.endswith(".cpp"):
                cmd += ["-x", "hip"]
            if verbose_resources:
                cmd += ["-Rpass-analysis=kernel-resource-usage"]
            _run(cmd + ["-c", s, "-o", o])
        objs.append(o)
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


def build_probe(force: bool = False) -> str:
    src = os.path.join(ROOT, "tests", "csrc", "probe.hip")
    if force or _newer(PROBE_PATH, [src]):
        _run([HIPCC] + _hip_flags() + ["-shared", src, "-o", PROBE_PATH])
    return PROBE_PATH


def build_oracle_c(force: bool = False) -> str:
    """Compile the plain-C crop restatement used only by tests/bench as the checker."""
    src = os.path.join(ROOT, "oracle", "crop_oracle.c")
    if not os.path.exists(src):
        return ""
    os.makedirs(os.path.dirname(ORACLE_C_PATH), exist_ok=True)
    if force or _newer(ORACLE_C_PATH, [src]):
        _run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", ORACLE_C_PATH, "-lm"])
    return ORACLE_C_PATH


EXAMPLE_PATH = os.path.join(ROOT, "examples", "score_candidates")


def build_example(force: bool = False) -> str:
    """The plain-C host of examples/score_candidates.c (gcc, linked against libpngpd.so and the HIP runtime)."""
    src = os.path.join(ROOT, "examples", "score_candidates.c")
    if not os.path.exists(src):
        return ""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if force or _newer(EXAMPLE_PATH, [src, LIB_PATH, os.path.join(ROOT, "include", "pngpd.h")]):
        _run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
              "-I", os.path.join(ROOT, "include"), src, "-o", EXAMPLE_PATH,
              "-L", os.path.dirname(LIB_PATH), "-lpngpd", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
              "-Wl,-rpath," + os.path.dirname(LIB_PATH), "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return EXAMPLE_PATH


def build_all(force: bool = False) -> None:
    build_library(force)
    build_example(force)
    build_probe(force)
    build_oracle_c(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
