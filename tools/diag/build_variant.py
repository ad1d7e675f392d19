"""Diagnostic: build tools/diag/libpngpd_<name>.so with extra -D flags.   python tools/diag/build_variant.py NAME [-DFLAG ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointnetgpd_amd import build as B
name = sys.argv[1]
flags = [f for f in sys.argv[2:] if f.startswith("-D")]
out = os.path.join(ROOT, "tools", "diag", "_obj", name)
os.makedirs(out, exist_ok=True)
objs = []
for s in B.LIB_SOURCES:
    src = os.path.join(B.CSRC, s)
    o = os.path.join(out, os.path.splitext(s)[0] + ".o")
    cmd = [B.HIPCC] + B._hip_flags() + flags
    if s.endswith(".cpp"):
        cmd += ["-x", "hip"]
    subprocess.run(cmd + ["-c", src, "-o", o], check=True)
    objs.append(o)
lib = os.path.join(ROOT, "tools", "diag", f"libpngpd_{name}.so")
subprocess.run([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print("built", lib)
