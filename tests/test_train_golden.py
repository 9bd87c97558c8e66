"""The reference's own TRAINING LOOP as a fixture (tests/golden/train_*.npz, make_golden_train.py: zinc/configs.py's
ZincConfig.optimizer / extra_setup / train / val / test run under shims on EgcZincNet over seeded ZINC-shaped batches, in
float32 and float64) -- SURVEY.md 8c's last row: "the build's own counterpart of zinc train / evaluate ... must reproduce".

CPU: the counterpart loop (tests/callers.py: train_epoch / evaluate / fit_zinc) around the counterpart net over the
differentiable CPU restatement of the layer, in float64, reproduces the reference's float64 trajectory -- training loss,
validation loss and learning rate of every iteration (the ReduceLROnPlateau halving inside train_zinc_plateau included) and
the test loss -- to 1e-8: loop, optimizer / scheduler wiring, BatchNorm mode switching and loss are pinned to the
reference's code.
GPU: the same loop on the gfx950 layers (plain modules and FusedEGCBlock + the segmented-mean readout) in float32 against the
float64 trajectory: the first ten iterations bounded per quantity by max(1e-5, 5 x (training loss) / 10 x (validation loss) the distance
between the reference's OWN float32 and float64 trajectories over those iterations), the rest by 2e-3 (a long run amplifies rounding differences
chaotically), and the learning-rate schedule equal wherever the reference's two runs agree on it."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import egc_amd
from callers import ZincNetLike, fit_zinc
from golden_util import GOLDEN_DIR
from test_nets_golden import _restated


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "train_*.npz")))


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _data(z, meta, dev, dtype):
    data = {}
    for split, count in (("train", meta["n_train"]), ("val", meta["n_val"]), ("test", meta["n_test"])):
        data[split] = []
        for i in range(count):
            batch = torch.from_numpy(z[f"{split}{i}:batch"])
            data[split].append(dict(atom=torch.from_numpy(z[f"{split}{i}:atom"]).to(dev),
                                    edge_index=torch.from_numpy(z[f"{split}{i}:edge_index"]).to(dev),
                                    batch=batch.to(dev), n_graphs=int(batch.max()) + 1,
                                    y=torch.from_numpy(z[f"{split}{i}:y"]).to(dev, dtype)))
    return data


def _net(z, meta):
    net = ZincNetLike(meta["hidden"], meta["layers"], lambda d: egc_amd.EfficientGraphConv(
        d, d, num_heads=meta["H"], num_bases=meta["B"], softmax_weights=False, aggrs=meta["aggrs"]), residual=True)
    net.load_state_dict({k[len("param:"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}, strict=True)
    return net


@pytest.mark.parametrize("name", _names())
def test_counterpart_loop_reproduces_the_reference_trajectory_in_float64(name):
    z, meta = _load(name)
    net = _net(z, meta).double()
    data = _data(z, meta, torch.device("cpu"), torch.float64)

    def forward(b):
        return net(b["atom"], b["edge_index"], b["batch"], b["n_graphs"], conv_fn=_restated)
    tr, va, lrs, te = fit_zinc(forward, [net], net.parameters(), data, meta["lr"], meta["wd"], meta["iterations"])
    assert np.abs(np.array(tr) - z["train_loss64"]).max() <= 1e-8
    assert np.abs(np.array(va) - z["val_loss64"]).max() <= 1e-8
    assert np.array_equal(np.array(lrs), z["lr64"])
    assert abs(te - float(z["test_loss64"])) <= 1e-8
    if name == "train_zinc_plateau":
        assert lrs[0] == meta["lr"] and lrs[-1] == meta["lr"] / 2        # the scheduler acted inside the run


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True, "batch"])
@pytest.mark.parametrize("name", _names())
def test_hip_layers_follow_the_reference_trajectory(name, fused):
    """fused = "batch": every batch of the loop handed to the blocks as an egc_amd.GraphBatch -- the one-launch forward, and for
    train_zinc_b64 (four bases of 16 channels) the one-launch backward of the batch path."""
    z, meta = _load(name)
    man = json.load(open(os.path.join(GOLDEN_DIR, "MANIFEST_TRAIN.json")))[name]
    dev = torch.device("cuda:0")
    net = _net(z, meta).to(dev)
    data = _data(z, meta, dev, torch.float32)
    kw, modules = {}, [net]
    if fused:
        blocks = {}

        def fuse(conv, bn, residual):
            if id(conv) not in blocks:
                blocks[id(conv)] = egc_amd.FusedEGCBlock(conv, bn, relu=True, residual=residual)
            blocks[id(conv)].train(bn.training)
            return blocks[id(conv)]
        kw = dict(fuse=fuse, pool=lambda x, batch, n_graphs: egc_amd.global_mean_pool(x, batch, n_graphs))

    ran = set()

    def forward(b):
        if fused != "batch":
            return net(b["atom"], b["edge_index"], b["batch"], b["n_graphs"], **kw)
        sizes = torch.bincount(b["batch"], minlength=b["n_graphs"])
        gb = egc_amd.GraphBatch(b["edge_index"], batch=b["batch"], num_graphs=b["n_graphs"], max_nodes=int(sizes.max()))
        out = net(b["atom"], gb, b["batch"], b["n_graphs"], **kw)
        gb.check()
        if torch.is_grad_enabled():        # (the training path settles both launches' tiles in the forward)
            ran.update(k[-1] for k, v in gb._setups.items() if isinstance(k, tuple) and isinstance(k[-1], str) and v)
        return out
    tr, va, lrs, te = fit_zinc(forward, modules, net.parameters(), data, meta["lr"], meta["wd"], meta["iterations"])
    if fused == "batch" and name == "train_zinc_b64":
        assert "fused_bwd" in ran, ran          # the one-launch backward, not a fallback
    # Calibrated bounds over the first ten iterations; beyond that a training run is a chaotic map of its rounding errors
    # (train_zinc_plateau: the reference's own float32 and float64 validation losses are 4e-2 apart after 26 iterations of
    # Adam at lr 0.03, although every single step agrees to 1e-6) and the whole trajectory is held to 2e-3 on losses of
    # 0.1 ... 1 -- what pins the late part is the learning-rate schedule below, which follows the validation losses.
    head = slice(0, 10)
    d_tr, d_va = np.abs(np.array(tr) - z["train_loss64"]), np.abs(np.array(va) - z["val_loss64"])
    r_tr, r_va = np.abs(z["train_loss32"] - z["train_loss64"]), np.abs(z["val_loss32"] - z["val_loss64"])
    tol_tr = max(1e-5, 5 * float(r_tr[head].max()))
    # (validation runs on BatchNorm's RUNNING statistics, which integrate the differences of every preceding step, and the
    # std net is the sensitive one -- DESIGN.md 3.2: measured 4.1e-5 against the reference's own 8.1e-6, run to run 3.5-4.2e-5)
    tol_va = max(1e-5, 10 * float(r_va[head].max()))
    if {"std", "var"} & set(meta["aggrs"]):
        # the std net's validation loss moves with the ORDER of the float atomics on hub rows (3.5e-5 ... 9e-5 run to run on
        # the same binary): the bound tests/test_backward_gpu.py gives std / var layers (gtol)
        tol_va = max(tol_va, 2e-4)
    assert d_tr[head].max() <= tol_tr, (d_tr, tol_tr)
    assert d_va[head].max() <= tol_va, (d_va, tol_va)
    assert d_tr.max() <= max(2e-3, 5 * man["f32_vs_f64_train_loss"]) and d_va.max() <= max(2e-3, 5 * man["f32_vs_f64_val_loss"])
    assert abs(te - float(z["test_loss64"])) <= max(2e-3, 5 * man["f32_vs_f64_val_loss"])
    if np.array_equal(z["lr32"], z["lr64"]):       # the reference's own float32 run takes the same schedule: so must this one
        assert np.array_equal(np.array(lrs), z["lr64"])
    assert tr[-1] < tr[0]
