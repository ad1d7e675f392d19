"""Diagnostic: where a workgroup of the bf16 conv-stack kernel spends its shader cycles.

    python tools/diag/stamps.py --build     (CPU box: builds tools/diag/libpngpd_diag.so with -DPNGPD_STAMPS)
    python tools/diag/stamps.py             (GPU box: runs B=4096, P=750 and prints per-wave segments)

The diagnostic library is the product library recompiled with in-kernel clock stamps; the product
build contains none of that code.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
def _opt(name, default):
    for i, f in enumerate(sys.argv):
        if f == name:
            return sys.argv[i + 1]
    return default


PRECISION = _opt("--precision", "bf16")
DIAG = os.path.join(ROOT, "tools", "diag", _opt("--lib", "libpngpd_diag") + ".so")


def build():
    from pointnetgpd_amd import build as B
    objs = []
    out = os.path.join(ROOT, "tools", "diag", "_obj")
    os.makedirs(out, exist_ok=True)
    for s in B.LIB_SOURCES:
        src = os.path.join(B.CSRC, s)
        o = os.path.join(out, os.path.splitext(s)[0] + ".o")
        cmd = [B.HIPCC] + B._hip_flags() + ["-DPNGPD_STAMPS"] + [f for f in sys.argv[1:] if f.startswith("-D")]
        if s.endswith(".cpp"):
            cmd += ["-x", "hip"]
        subprocess.run(cmd + ["-c", src, "-o", o], check=True)
        objs.append(o)
    subprocess.run([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", DIAG] + objs, check=True)
    print("built", DIAG)


def run(B=4096, P=750):
    import numpy as np
    import torch
    from pointnetgpd_amd import _lib, synth
    _lib.LIB_PATH = DIAG
    from pointnetgpd_amd.pointnet import PointNetCls
    dev = torch.device("cuda", 0)
    model = PointNetCls(P, 3, 2)
    model.load_state_dict(synth.torch_state_dict(0, 2, 0.02), strict=True)
    model = model.to(dev).eval()
    model.precision = PRECISION
    x = torch.from_numpy(synth.candidate_points(0, B, P)).to(dev)
    for _ in range(300):
        model(x)
    torch.cuda.synchronize()
    lib = _lib.load()
    waves, nb, ne = 8, 6, 5
    stamps = (ctypes.c_ulonglong * (waves * nb * ne))()
    clock = (ctypes.c_ulonglong * 4)()
    rc = lib.pngpd_debug_read_stamps(stamps, clock)
    assert rc == 0, rc
    s = np.array(stamps, dtype=np.int64).reshape(waves, nb, ne)
    cyc, rt = clock[2] - clock[0], clock[3] - clock[1]
    print(f"kernel: {cyc} shader cycles, {rt / 100.0:.1f} us -> {cyc / rt * 100.0:.0f} MHz")
    t0 = s[:, :, 0].min()
    print("per wave, per batch: start, consume-before-produce, produce, consume-after, barrier wait, total")
    for b in range(nb):
        for w in range(waves):
            e = s[w, b]
            print(f"  batch {b} wave {w}: start {e[0] - t0:7d} | pre {e[1] - e[0]:6d} | produce {e[2] - e[1]:6d} | "
                  f"post {e[3] - e[2]:6d} | barrier {e[4] - e[3]:6d} | total {e[4] - e[0]:6d}")
        print()


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        run()
