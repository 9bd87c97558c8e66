#!/usr/bin/env python3
"""std / var layers against the reference-generated fixtures (tests/golden/) in the two modes of the library: the default
(squares about the row's first entry) and EGC_STDVAR_REFERENCE=1 (the reference's float32 mean(x^2) - mean(x)^2 on the general
kernels).  Prints, per fixture: scale-relative error, element-wise excess at 1e-5 against the fixture and against float64."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import elementwise_excess, float64_forward, golden_names, load_golden, rel_err  # noqa: E402
from test_parity_gpu import build_layer, run_layer  # noqa: E402

dev = torch.device("cuda:0")
for name in golden_names():
    g = load_golden(name)
    if not any(a in ("std", "var") for a in g["meta"]["aggrs"]):
        continue
    r64 = float64_forward(g)
    line = f"{name:28s} fixture vs float64: excess {elementwise_excess(g['out'], r64, 1e-5):8.3f} |"
    for mode in ("0", "1"):
        os.environ["EGC_STDVAR_REFERENCE"] = mode
        out = run_layer(build_layer(g["meta"], g["params"], dev), g, dev)
        line += (f" {'reference formula' if mode == '1' else 'default'}: rel {rel_err(out, g['out']):.2e} excess vs fixture "
                 f"{elementwise_excess(out, g['out'], 1e-5):8.3f} vs float64 {elementwise_excess(out, r64, 1e-5):8.3f} |")
    print(line)
