"""Helpers to load tests/golden/*.npz fixtures (made by tests/golden/make_golden.py)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """Single-layer fixtures (make_golden.py); the relational ones (make_golden_rel.py) are rel_*."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # grad_*: make_golden_grad.py (test_backward_golden.py); net_* / relgrad_*: make_golden_nets.py (test_nets_golden.py);
    # train_*: make_golden_train.py (test_train_golden.py)
    return [n for n in names if not n.startswith(("rel_", "grad_", "net_", "relgrad_", "train_"))]


def rel_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "rel_*.npz")))


def load_rel_golden(name):
    """-> dict(meta, x {type: array}, ei {(src, rel, dst): [2, E] (source ids, target ids)}, params, out)."""
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    edge_types = [tuple(k) for k in meta["edge_types"]]
    return dict(name=name, meta=meta,
                x={k: z[f"x_{k}"] for k in meta["node_types"]},
                out={k: z[f"out_{k}"] for k in meta["node_types"]},
                ei={k: z[f"ei_{i}"] for i, k in enumerate(edge_types)},
                params={k[2:]: z[k] for k in z.files if k.startswith("p_")})


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(str(z["meta"]))
    params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    return dict(name=name, meta=meta, x=z["x"], edge_index=z["edge_index"], out=z["out"], params=params)


def oracle_forward(g, orc, return_intermediates=False):
    """Run the numpy oracle on a golden fixture's inputs/params."""
    m, p = g["meta"], g["params"]
    if m["kind"] == "lay":
        bw = [p[f"bases_weight.{b}"] for b in range(m["B"])]
        return orc.efficient_graph_conv_forward(
            g["x"], g["edge_index"], bw, p["comb_weights.weight"], p["comb_weights.bias"], p.get("bias"),
            m["H"], m["aggrs"], softmax_weights=m["softmax"], sigmoid_weights=m["sigmoid"],
            hardtanh_weights=m["hardtanh"], add_self_loops=m["add_self_loops"],
            return_intermediates=return_intermediates)
    return orc.egconv_forward(
        g["x"], g["edge_index"], p["bases_weight"], p["comb_weight.weight"], p["comb_weight.bias"], p.get("bias"),
        m["H"], m["B"], m["aggrs"], add_self_loops=m["add_self_loops"], sigmoid=m["sigmoid"],
        return_intermediates=return_intermediates)


def float64_forward(g):
    """The layer of a fixture-shaped dict evaluated in FLOAT64 (oracle/egc_torch_ref.py on double tensors, no gradients): what
    the element-wise checks of the shape sweeps hold the HIP output against."""
    import torch
    from oracle import egc_torch_ref as tref
    m, p = g["meta"], g["params"]
    t = lambda a: None if a is None else torch.from_numpy(np.asarray(a)).double()      # noqa: E731
    x = t(g["x"])
    with torch.no_grad():
        if m["kind"] == "lay":
            out = tref.efficient_graph_conv_forward(x, g["edge_index"], [t(p[f"bases_weight.{b}"]) for b in range(m["B"])],
                                                    t(p["comb_weights.weight"]), t(p["comb_weights.bias"]), t(p.get("bias")), m["H"],
                                                    m["aggrs"], softmax=m["softmax"], sigmoid=m["sigmoid"], hardtanh=m["hardtanh"],
                                                    add_self_loops=m["add_self_loops"])
        else:
            out = tref.egconv_forward(x, g["edge_index"], t(p["bases_weight"]), t(p["comb_weight.weight"]), t(p["comb_weight.bias"]),
                                      t(p.get("bias")), m["H"], m["B"], m["aggrs"], add_self_loops=m["add_self_loops"],
                                      sigmoid=m["sigmoid"])
    return out.numpy()


def rel_err(a, b):
    """max |a-b| / max(1, max|b|)  -- the scale-relative error used across the parity tests."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def elementwise_excess(a, b, tol=1e-5):
    """Element-wise companion of rel_err (VERDICT r2 next #4c): the largest |a - b| / (tol |b| + tol s_row), with
    s_row = max(1, max_j |b[row, j]|) the scale of the element's OWN output row -- rel_err alone measures every element
    against the largest output of the whole array and says nothing about small rows next to large ones.  <= 1 passes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    b2 = b.reshape(b.shape[0], -1) if b.ndim > 1 else b.reshape(1, -1)
    a2 = a.reshape(b2.shape)
    s_row = np.maximum(1.0, np.abs(b2).max(axis=1, keepdims=True))
    return float((np.abs(a2 - b2) / (tol * np.abs(b2) + tol * s_row)).max())
