#!/usr/bin/env python
"""Cost of the inducing-count padding: cfg-2-shaped model (3 layers, S=20, minibatch 1000) at M = 96, 100, 112, 128 and 200, 224,
256, 300, 320 — ms per training step.  M = 100 executes as Mp = 112 (not 128), M = 300 as 320 (not 512)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from doubly_stochastic_dgp.dgp import DGP  # noqa: E402
from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian  # noqa: E402

rng = np.random.default_rng(0)
X, Y = rng.standard_normal((7372, 8)), rng.standard_normal((7372, 1))
for M in (96, 100, 112, 128, 200, 224, 256, 300, 320, 512):
    Z = X[rng.permutation(7372)[:M]] + 0.01 * rng.standard_normal((M, 8))
    model = DGP(X, Y, Z, [RBF(8) for _ in range(3)], Gaussian(), num_samples=20, minibatch_size=1000)
    for layer in model.layers[:-1]:
        layer.q_sqrt = layer.q_sqrt.value * 1e-5
    for _ in range(10):
        model.train_step(0.01)
    torch.cuda.synchronize()
    n = 100 if M <= 128 else 40
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            model.train_step(0.01)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    print(json.dumps({"M": M, "ms_per_step": round(best, 4)}), flush=True)
    del model
