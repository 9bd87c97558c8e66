#!/usr/bin/env python3
"""Generate tests/golden/rel_*.npz by running the REFERENCE's own REGConv (experiments/rmag/models.py:75-148).

Same arrangement as make_golden.py (build container only; the absent third-party packages are the shims
defined there, whose numerical bodies are the restatements in oracle/egc_oracle.py).  REGConv itself is
constructible -- the reference's ``super(self)`` bug (rmag/models.py:161) is in the REGC wrapper only.
Only vectors are committed.  Usage:  python tests/golden/make_golden_rel.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASES = [
    dict(name="rel_small", fin=24, fout=32, H=4, B=2, sizes=dict(author=37, field_of_study=11, institution=5, paper=29),
         edges=60, seed=1),
    dict(name="rel_mag_shape", fin=128, fout=64, H=8, B=4, sizes=dict(author=150, field_of_study=40, institution=12, paper=120),
         edges=900, seed=2),
]


def main():
    mg.install_shims()
    mg.SparseTensor.matmul = lambda self, x, reduce="sum": mg.shim_sparse_matmul(self, x, reduce)
    spec = importlib.util.spec_from_file_location("ref_rmag_models", "/root/reference/experiments/rmag/models.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    manifest = {}
    for c in CASES:
        rng = np.random.default_rng(c["seed"])
        torch.manual_seed(c["seed"])
        conv = ref.REGConv(c["fin"], c["fout"], c["H"], c["B"])
        with torch.no_grad():  # default Linear biases are tiny; make every term visible
            for lin in list(conv.rel_combs.values()) + list(conv.root_combs.values()):
                lin.bias.normal_(std=0.3)
        x_dict = {k: torch.randn(n, c["fin"]) for k, n in c["sizes"].items()}
        adj, save = {}, {}
        for i, key in enumerate(ref.EDGE_TYPES):
            n_src, n_dst = c["sizes"][key[0]], c["sizes"][key[2]]
            e = c["edges"] if i != 1 else 7           # one very sparse relation: most targets have no in-edge
            src = rng.integers(0, n_src, size=e)
            dst = rng.integers(0, max(1, n_dst - 3), size=e)   # the last targets never receive anything
            if i == 4:
                dst[: e // 3] = 0                      # a long row (degree > 32 in the larger case)
            adj[key] = mg.SparseTensor(row=torch.from_numpy(dst), col=torch.from_numpy(src), sparse_sizes=(n_dst, n_src))
            save[f"ei_{i}"] = np.stack([src, dst]).astype(np.int64)
        with torch.no_grad():
            out = conv(x_dict, adj)
        arrays = dict(save)
        for k, v in x_dict.items():
            arrays[f"x_{k}"] = v.numpy()
            arrays[f"out_{k}"] = out[k].numpy()
        for k, v in conv.state_dict().items():
            arrays[f"p_{k}"] = v.numpy()
        meta = dict(fin=c["fin"], fout=c["fout"], H=c["H"], B=c["B"], node_types=list(ref.NODE_TYPES),
                    edge_types=[list(k) for k in ref.EDGE_TYPES])
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{c['name']}.npz"), **arrays)
        manifest[c["name"]] = meta
        print(c["name"], {k: tuple(v.shape) for k, v in out.items()})
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST_rel.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
