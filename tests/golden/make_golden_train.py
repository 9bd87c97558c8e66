"""Training-loop fixtures from the reference's OWN train / evaluate code (last row): tests/golden/train_*.npz.

What runs here is /root/reference/experiments/zinc/configs.py as it stands, imported under the shims of make_golden.py /
make_golden_grad.py / make_golden_nets.py plus stub modules for the experiment harness it names at import time (exptune,
ray.tune, torch_geometric's DataLoader / ZINC dataset -- none of them on the numeric path):

    cfg = ZincConfig()                                   zinc/configs.py:93-96
    optimizer = cfg.optimizer(model, hparams)            zinc/configs.py:128-129   Adam(lr, weight_decay)
    extra = cfg.extra_setup(model, optimizer, hparams)   zinc/configs.py:131-139   ReduceLROnPlateau(min, 0.5, patience 10, min_lr 1e-5)
    per iteration: cfg.train(...)  -> train()            zinc/configs.py:53-72,144-145
                   cfg.val(...)    -> evaluate() + lr_scheduler.step(val_loss)     zinc/configs.py:75-90,147-151
    at the end:    cfg.test(...)                         zinc/configs.py:153-154

on the reference's EgcZincNet (zinc/models.py:92-135) over seeded synthetic ZINC-shaped batches (egc_amd.workloads), in float32 and
in float64 from the same initial state.  Saved: the batches, the initial state dict, per-iteration train loss / validation
loss / learning rate, the test loss and the final parameters, for both dtypes.  The float32-float64 distance of the
reference's own runs calibrates the bound the HIP path is held to (tests/test_train_golden.py).

Run from the repository root IN THIS CONTAINER (needs /root/reference):  python tests/golden/make_golden_train.py
"""
from __future__ import annotations

import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402,F401
import make_golden_nets as mgn  # noqa: E402
from egc_amd.workloads import zinc_like_batch  # noqa: E402

REF = mgn.REF


class _Anything:
    """Harness names the config module mentions (settings, search strategies, schedulers, summaries): constructible, inert."""

    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k


class _Metric:
    def __init__(self, name, mode):
        self.name, self.mode = name, mode


class _ExperimentConfig:
    def __init__(self, debug_mode=False):
        self.debug_mode = debug_mode


def _stub(name, **attrs):
    m = types.ModuleType(name)
    def other(attr):                                # any other public name: an inert class
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything
    m.__getattr__ = other
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    mgn.install()
    _stub("exptune")
    _stub("exptune.exptune", Metric=_Metric, ExperimentConfig=_ExperimentConfig)
    _stub("exptune.hyperparams")
    _stub("exptune.search_strategies")
    _stub("exptune.summaries")
    _stub("exptune.summaries.final_run_summaries")
    _stub("exptune.utils")
    _stub("ray")
    _stub("ray.tune")
    _stub("ray.tune.schedulers")
    _stub("torch_geometric.data")
    _stub("torch_geometric.datasets")
    mgn._load("experiments.exp_config", os.path.join(REF, "exp_config.py"))
    zm = mgn._load("experiments.zinc.models", os.path.join(REF, "zinc", "models.py"))
    zc = mgn._load("experiments.zinc.configs", os.path.join(REF, "zinc", "configs.py"))
    return zm, zc


class Batch(types.SimpleNamespace):
    """What the loops touch of a torch_geometric Batch: .x, .edge_index, .batch, .y and .to(device)."""

    def to(self, device):
        return self


def _batches(seed, sizes, dtype):
    out = []
    for k, n_graphs in enumerate(sizes):
        atom, ei, n, batch = zinc_like_batch(n_graphs, seed=seed + k)
        rng = np.random.default_rng(seed + 100 + k)
        y = torch.from_numpy(rng.standard_normal(n_graphs).astype(np.float32)).to(dtype)     # ZINC's y: one float per graph
        out.append(Batch(x=atom.view(-1, 1), edge_index=ei, batch=batch, y=y))
    return out


def _run(zm, zc, case, dtype, state):
    torch.manual_seed(0)
    net = zm.EgcZincNet(case["hidden"], case["layers"], 0.0, True, readout="mean", heads=case["H"], bases=case["B"],
                        aggrs=case["aggrs"])
    net.load_state_dict(state)
    net = net.to(dtype)
    data = {"train": _batches(case["seed"], case["train_sizes"], dtype), "val": _batches(case["seed"] + 1000, case["val_sizes"], dtype),
            "test": _batches(case["seed"] + 2000, case["test_sizes"], dtype)}
    cfg = zc.ZincConfig()
    hparams = {"lr": case["lr"], "wd": case["wd"], "batch_size": case["train_sizes"][0]}
    opt = cfg.optimizer(net, hparams)
    with redirect_stdout(io.StringIO()):            # print_model_parameters
        extra = cfg.extra_setup(net, opt, hparams)
    extra.device = torch.device("cpu")
    tr, va, lr = [], [], []
    for it in range(case["iterations"]):
        m, _ = cfg.train(net, opt, data, extra, it)
        tr.append(m["train_loss"])
        m, _ = cfg.val(net, data, extra, it)
        va.append(m["val_loss"])
        lr.append(opt.param_groups[0]["lr"])
    te = cfg.test(net, data, extra)[0]["test_loss"]
    final = {k: v.detach().double().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    return np.array(tr), np.array(va), np.array(lr), float(te), final, data


CASES = [
    dict(name="train_zinc_egcs", hidden=32, layers=4, H=4, B=2, aggrs=["symadd"], seed=31, lr=0.01, wd=1e-4,
         train_sizes=[10, 10, 9], val_sizes=[8], test_sizes=[7], iterations=8),
    dict(name="train_zinc_egcm", hidden=24, layers=4, H=2, B=2, aggrs=["add", "std", "max"], seed=32, lr=0.005, wd=1e-4,
         train_sizes=[12, 11], val_sizes=[9], test_sizes=[6], iterations=6),
    # long enough for the validation loss to stall: ReduceLROnPlateau halves the rate inside the run
    # four bases of 16 channels (the one-launch training path of a batch: egc_layer_backward_batch_fused_f32)
    dict(name="train_zinc_b64", hidden=64, layers=4, H=4, B=4, aggrs=["symadd", "max", "mean"], seed=34, lr=0.005, wd=1e-4,
         train_sizes=[10, 10, 9], val_sizes=[8], test_sizes=[7], iterations=8),
    dict(name="train_zinc_plateau", hidden=16, layers=4, H=2, B=2, aggrs=["symadd", "max"], seed=33, lr=0.03, wd=0.0,
         train_sizes=[6, 6], val_sizes=[6], test_sizes=[6], iterations=26),
]


def main():
    zm, zc = install()
    manifest = {}
    for case in CASES:
        torch.manual_seed(case["seed"])
        rng = np.random.default_rng(case["seed"])
        net0 = zm.EgcZincNet(case["hidden"], case["layers"], 0.0, True, readout="mean", heads=case["H"], bases=case["B"],
                             aggrs=case["aggrs"])
        mgn._randomise(net0, rng)
        state = {k: v.detach().clone() for k, v in net0.state_dict().items()}
        r32 = _run(zm, zc, case, torch.float32, state)
        r64 = _run(zm, zc, case, torch.float64, state)
        arrays = {f"param:{k}": v.numpy() for k, v in state.items()}
        for split in ("train", "val", "test"):
            for i, b in enumerate(r64[5][split]):
                arrays[f"{split}{i}:atom"] = b.x.view(-1).numpy()
                arrays[f"{split}{i}:edge_index"] = b.edge_index.numpy()
                arrays[f"{split}{i}:batch"] = b.batch.numpy()
                arrays[f"{split}{i}:y"] = b.y.double().numpy()
        for tag, r in (("32", r32), ("64", r64)):
            arrays[f"train_loss{tag}"], arrays[f"val_loss{tag}"], arrays[f"lr{tag}"] = r[0], r[1], r[2]
            arrays[f"test_loss{tag}"] = np.array(r[3])
            for k, v in r[4].items():
                arrays[f"final{tag}:{k}"] = v
        meta = {k: v for k, v in case.items() if k != "name"}
        meta.update(net="EgcZincNet", ref="zinc/configs.py:53-90,128-154; zinc/models.py:17-74,92-135",
                    n_train=len(case["train_sizes"]), n_val=len(case["val_sizes"]), n_test=len(case["test_sizes"]))
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **arrays)
        d_tr = float(np.abs(r32[0] - r64[0]).max()), float(np.abs(r32[1] - r64[1]).max())
        d_par = max(float(np.abs(r32[4][k] - r64[4][k]).max()) for k in r64[4])
        manifest[case["name"]] = dict(meta, f32_vs_f64_train_loss=d_tr[0], f32_vs_f64_val_loss=d_tr[1],
                                      f32_vs_f64_final_params=d_par, lr_first=float(r64[2][0]), lr_last=float(r64[2][-1]),
                                      train_loss_first=float(r64[0][0]), train_loss_last=float(r64[0][-1]))
        print(case["name"], "train", r64[0][0], "->", r64[0][-1], "val", r64[1][0], "->", r64[1][-1], "lr", r64[2][0], "->", r64[2][-1],
              "| f32 vs f64: losses", d_tr, "params", d_par)
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST_TRAIN.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
