"""Golden vectors for the frame construction of the training-data crop, produced by the REFERENCE'S OWN statements.

Build container only (needs /root/reference):  python tests/golden/make_dataset_golden.py

model/dataset.py cannot be imported (pcl, dexnet, autolab_core at module level), and collect_pc_
(model/dataset.py:767-924) calls dex-net's get_hand_points for the box limits.  But everything before that call -
the statements from `center = grasp[0:3]` to `pc_c2g = ...` (:776-843): axis normalisation, the two rotation
matrices, the approach and minor axes, the bottom centre, the 4x4 transform of centre and axes, the matrix and the
gripper-frame coordinates of every point - is numpy only, apart from one number read as `ags.gripper.hand_depth`.
This script takes exactly those statements out of the method's syntax tree, wraps them in a function of
(grasp, pc, transform, ags) that returns (matrix, bottom_, pc_c2g), compiles it with numpy as its only global and
runs it; `ags` is a holder for that one number.  Nothing of the reference's text is stored.  The six comparisons
that follow in the reference are those of the online crop (pinned by make_crop_golden.py); the corner table they
read (dex-net) stays an argument.
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REFERENCE_FILE = "/root/reference/model/dataset.py"


def reference_frame_part(path=REFERENCE_FILE):
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PointGraspOneViewDataset"]
    assert len(cls) == 1
    fn = [n for n in cls[0].body if isinstance(n, ast.FunctionDef) and n.name == "collect_pc_"]
    assert len(fn) == 1
    body = []
    for stmt in fn[0].body:
        if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant) and isinstance(stmt.value.value, str):
            continue                                       # docstrings and string-literal comments
        body.append(stmt)
        if isinstance(stmt, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "pc_c2g" for t in stmt.targets):
            break
    else:
        raise AssertionError("no assignment to pc_c2g in collect_pc_")
    names = {n.id for s in body for n in ast.walk(s) if isinstance(n, ast.Name)}
    assert "mlab" not in names and "self" not in names, names
    ret = ast.Return(value=ast.Tuple(elts=[ast.Name(id=n, ctx=ast.Load()) for n in ("matrix", "bottom_", "pc_c2g")],
                                     ctx=ast.Load()))
    args = ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in ("grasp", "pc", "transform", "ags")],
                         kwonlyargs=[], kw_defaults=[], defaults=[])
    func = ast.FunctionDef(name="frame_part", args=args, body=body + [ret], decorator_list=[], lineno=fn[0].lineno,
                           col_offset=0)
    module = ast.fix_missing_locations(ast.Module(body=[func], type_ignores=[]))
    ns = {"np": np, "__name__": "reference_extract"}
    exec(compile(module, path, "exec"), ns)
    return ns["frame_part"]


def reference_getitem_tail(path=REFERENCE_FILE):
    """The statements of PointGraspOneViewDataset.__getitem__ after collect_pc_ (model/dataset.py:1098-1131: the
    None check, resampling to grasp_points_num points, the score and the label) as a function of
    (self, grasp, grasp_pc, obj_grasp); `self` only carries the dataset's own numbers (projection, grasp_points_num,
    thresh_good, thresh_bad, with_obj)."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PointGraspOneViewDataset"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__getitem__"][0]
    start = None
    for i, stmt in enumerate(fn.body):
        if isinstance(stmt, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "grasp_pc" for t in stmt.targets):
            start = i + 1                                  # the statement after `grasp_pc = self.collect_pc_(...)`
    assert start is not None
    body = [st for st in fn.body[start:]
            if not (isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant) and isinstance(st.value.value, str))]
    args = ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in ("self", "grasp", "grasp_pc", "obj_grasp")],
                         kwonlyargs=[], kw_defaults=[], defaults=[])
    func = ast.FunctionDef(name="getitem_tail", args=args, body=body, decorator_list=[], lineno=fn.lineno, col_offset=0)
    module = ast.fix_missing_locations(ast.Module(body=[func], type_ignores=[]))
    ns = {"np": np, "__name__": "reference_extract"}
    exec(compile(module, path, "exec"), ns)
    return ns["getitem_tail"]


def tail_cases(seed=17, N=64):
    rng = np.random.default_rng(seed)
    out = []
    for n, level, refine in ((200, 0.1, 0.2), (N + 1, 0.9, 0.5), (N, 0.3, 1.0), (N - 1, 0.55, 0.0), (50, 0.6, 0.0),
                             (57, 0.4, 20.0), (300, 0.5, 0.0)):
        pc = rng.normal(size=(n, 3))
        grasp = np.concatenate([rng.normal(size=8), [level, refine]])
        out.append((grasp, pc))
    return out, N, 0.5, 0.6                                 # thresh_good, thresh_bad (main_1v.py:34-35 style values)


def cases(seed=91, G=40, S=96):
    rng = np.random.default_rng(seed)
    rec = np.zeros((G, 10))
    rec[:, 0:3] = rng.uniform(-0.1, 0.1, size=(G, 3))                 # contact centre
    axis = rng.normal(size=(G, 3))
    axis[0] = [0.0, 0.0, 1.0]                                          # axis along z: the `norm(axis_x) == 0` branch
    axis[1] = [0.0, 0.0, -2.5]
    axis[2] = [1.0, 0.0, 0.0]
    axis[3] = [0.0, 3.0, 0.0]
    axis[4:] *= rng.uniform(0.2, 5.0, size=(G - 4, 1))                  # not unit: the reference normalises
    rec[:, 3:6] = axis
    rec[:, 6] = rng.uniform(0.02, 0.085, size=G)                       # width
    rec[:, 7] = rng.uniform(-np.pi, np.pi, size=G)                     # angle
    rec[5, 7] = 0.0
    rec[6, 7] = np.pi / 2
    rec[:, 8:10] = rng.uniform(0, 1, size=(G, 2))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    rigid = np.eye(4)
    rigid[:3, :3] = q * np.sign(np.linalg.det(q))
    rigid[:3, 3] = rng.uniform(-0.3, 0.3, size=3)
    sheared = rigid.copy()
    sheared[:3, :3] = sheared[:3, :3] @ np.array([[1.3, 0.1, 0.0], [0.0, 0.8, 0.05], [0.0, 0.0, 1.1]])
    pc = rng.uniform(-0.25, 0.25, size=(S, 3)).astype(np.float32)
    return rec, {"rigid": rigid, "sheared": sheared}, pc, 0.06


def main():
    frame_part = reference_frame_part()
    rec, transforms, pc, hand_depth = cases()
    ags = types.SimpleNamespace(gripper=types.SimpleNamespace(hand_depth=hand_depth))
    out = {"records": rec, "pc": pc, "hand_depth": hand_depth}
    for name, T in transforms.items():
        mats, bots, local = [], [], []
        for g in rec:
            matrix, bottom_, pc_c2g = frame_part(g, pc, T, ags)
            mats.append(np.asarray(matrix, dtype=np.float64))
            bots.append(np.asarray(bottom_, dtype=np.float64))
            local.append(np.asarray(pc_c2g, dtype=np.float64))
        out[f"transform_{name}"] = T
        out[f"matrix_{name}"] = np.stack(mats)
        out[f"bottom_{name}"] = np.stack(bots)
        out[f"pc_c2g_{name}"] = np.stack(local)
        print(name, "matrix", out[f"matrix_{name}"].shape, "pc_c2g", out[f"pc_c2g_{name}"].shape)
    np.savez_compressed(os.path.join(HERE, "dataset_frames_reference.npz"), **out)

    tail = reference_getitem_tail()
    tcases, N, good, bad = tail_cases()
    self_ = types.SimpleNamespace(projection=False, grasp_points_num=N, thresh_good=good, thresh_bad=bad, with_obj=False)
    t = {"N": N, "thresh_good": good, "thresh_bad": bad, "count": len(tcases)}
    for i, (grasp, pc) in enumerate(tcases):
        np.random.seed(1000 + i)
        res = tail(self_, grasp, pc, "obj")
        t[f"grasp_{i}"] = grasp
        t[f"pc_{i}"] = pc
        t[f"kept_{i}"] = res is not None
        if res is not None:
            t[f"out_{i}"] = np.asarray(res[0])
            t[f"label_{i}"] = res[1]
        print("tail case", i, "n =", len(pc), "->", None if res is None else (np.asarray(res[0]).shape, res[1]))
    np.savez_compressed(os.path.join(HERE, "dataset_tail_reference.npz"), **t)


if __name__ == "__main__":
    main()
