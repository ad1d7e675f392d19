"""The C ABI used from a plain C program (examples/score_candidates.c: no Python, no torch in the
process) gives the same numbers as the Python modules that wrap the same library."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from pointnetgpd_amd import _lib, synth
from pointnetgpd_amd.pointnet import PointNetCls

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "score_candidates")

# state_dict prefixes of the 12 units, in the slot order of include/pngpd.h
PREFIX = {"stn_conv1": ("feat.stn.conv1", "feat.stn.bn1"), "stn_conv2": ("feat.stn.conv2", "feat.stn.bn2"),
          "stn_conv3": ("feat.stn.conv3", "feat.stn.bn3"), "stn_fc1": ("feat.stn.fc1", "feat.stn.bn4"),
          "stn_fc2": ("feat.stn.fc2", "feat.stn.bn5"), "stn_fc3": ("feat.stn.fc3", None),
          "feat_conv1": ("feat.conv1", "feat.bn1"), "feat_conv2": ("feat.conv2", "feat.bn2"),
          "feat_conv3": ("feat.conv3", "feat.bn3"), "fc1": ("fc1", "bn1"), "fc2": ("fc2", "bn2"), "fc3": ("fc3", None)}


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_c_program_matches_python_modules(tmp_path, precision):
    assert os.path.exists(EXE), "examples/score_candidates missing: run python -c 'import __graft_entry__ as g; g.build()'"
    P, k, B, seed = 750, 2, 37, 51
    sd = synth.torch_state_dict(seed, k, 0.02)
    x = synth.candidate_points(seed, B, P, "gripper")
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as f:
        f.write(struct.pack("<4i", P, k, B, _lib.PRECISIONS[precision]))
        for unit in _lib.UNITS:
            layer, bn = PREFIX[unit]
            names = [f"{layer}.weight", f"{layer}.bias"]
            names += [f"{bn}.weight", f"{bn}.bias", f"{bn}.running_mean", f"{bn}.running_var"] if bn else [None] * 4
            for n in names:
                if n is None:
                    f.write(struct.pack("<q", 0))
                else:
                    t = sd[n].detach().cpu().numpy().astype(np.float32).ravel()
                    f.write(struct.pack("<q", t.size))
                    f.write(t.tobytes())
        f.write(np.ascontiguousarray(x, dtype=np.float32).tobytes())
    r = subprocess.run([EXE, src, dst], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = np.fromfile(dst, dtype=np.float32)
    logp_c, trans_c = out[:B * k].reshape(B, k), out[B * k:].reshape(B, 3, 3)

    m = PointNetCls(P, 3, k)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    logp, trans = m(torch.from_numpy(x).cuda())
    np.testing.assert_array_equal(logp_c, logp.cpu().numpy())
    np.testing.assert_array_equal(trans_c, trans.cpu().numpy())
