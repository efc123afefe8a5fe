#!/usr/bin/env python
"""Throughput of the BASELINE.json config shapes on ONE MI355X (configs 4 and 5 are 8-GPU configs: the per-GPU shard of
their minibatch is timed).  Synthetic data of the stated shapes; a step = gather + ELBO + gradient + optimiser step(s).
Prints one JSON line per config (kept under profiles/)."""
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
from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian, MultiClass  # noqa: E402
from doubly_stochastic_dgp.training import NatGradOptimizer  # noqa: E402

CONFIGS = [
    dict(name="cfg1: 1-layer (SVGP), M=50, S=1, mb=100", n=7372, D=8, widths=[8], M=50, S=1, mb=100, steps=300),
    dict(name="cfg2: 3-layer, M=128, S=20, mb=1000", n=7372, D=8, widths=[8, 8, 8], M=128, S=20, mb=1000, steps=200),
    dict(name="cfg3: 5-layer protein-shaped, M=256, S=20, mb=2000", n=41157, D=9, widths=[9] * 5, M=256, S=20, mb=2000, steps=30),
    dict(name="cfg4: MNIST-shaped 784-30-30-10, MultiClass, M=512, S=10, mb=4096/8 per GPU", n=6000, D=784,
         widths=[784, 30, 30], M=512, S=10, mb=512, steps=10, classes=10, var=2.0, ls=2.0),
    dict(name="cfg5: 3-layer, M=1024, S=50, mb=1000/8 per GPU, + natgrad(last layer)", n=7372, D=8, widths=[8, 8, 8],
         M=1024, S=50, mb=125, steps=5, natgrad=0.1),
]
# beyond BASELINE.json (ids continue after the configs): inducing counts above the chain kernels' 1024
EXTRA = [
    dict(name="x6: 3-layer, M=2048, S=20, mb=250 (GEMM-formulated passes only)", n=7372, D=8, widths=[8, 8, 8], M=2048, S=20, mb=250, steps=5),
    dict(name="x7: 2-layer, M=1536, S=10, mb=1000", n=7372, D=8, widths=[8, 8], M=1536, S=10, mb=1000, steps=5),
    dict(name="x8: config-5 shape with white=True (no natural-gradient step)", n=7372, D=8, widths=[8, 8, 8], M=1024, S=50, mb=125, steps=5,
         white=True),
    dict(name="x9: config-4 shape with white=True", n=6000, D=784, widths=[784, 30, 30], M=512, S=10, mb=512, steps=10, classes=10, var=2.0,
         ls=2.0, white=True),
]


def build(cfg_id):
    """(model, step) of config `cfg_id` (1-based) without timing — used by tools/ab_kernels.py."""
    return _build((CONFIGS + EXTRA)[cfg_id - 1])


def _build(cfg):
    rng = np.random.default_rng(0)
    n, D = cfg["n"], cfg["D"]
    if cfg.get("classes"):
        X = rng.uniform(size=(n, D)) * (rng.uniform(size=(n, D)) < 0.19)
        Y = rng.integers(0, cfg["classes"], size=(n, 1)).astype(np.float64)
        lik = MultiClass(cfg["classes"])
    else:
        X = rng.standard_normal((n, D))
        Y = rng.standard_normal((n, 1))
        lik = Gaussian()
    Z = X[rng.permutation(n)[:cfg["M"]]] + 0.01 * rng.standard_normal((cfg["M"], D))
    kernels = [RBF(w, variance=cfg.get("var", 1.0), lengthscales=cfg.get("ls", 1.0)) for w in cfg["widths"]]
    model = DGP(X, Y, Z, kernels, lik, num_outputs=cfg.get("classes"), num_samples=cfg["S"], minibatch_size=cfg["mb"],
                white=cfg.get("white", False))
    for layer in model.layers[:-1]:
        layer.q_sqrt = layer.q_sqrt.value * 1e-5
    last = model.layers[-1]
    ng = NatGradOptimizer(cfg["natgrad"]) if cfg.get("natgrad") else None

    def step():
        if ng is not None:
            model._build_likelihood(with_grad=True, grad_from_layer=len(model.layers) - 1, grad_q_only=True)   # as NatGradOptimizer.minimize does
            model.engine().natgrad_step(len(model.layers) - 1, ng.gamma, check=False)
        model.train_step(0.01)

    return model, step


def run(cfg):
    model, step = _build(cfg)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cfg["steps"]):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    elbo = model.train_step(0.01, sync=True)
    return dict(config=cfg["name"], steps_per_s=round(cfg["steps"] / dt, 3), ms_per_step=round(1e3 * dt / cfg["steps"], 3),
                elbo_finite=bool(np.isfinite(elbo)), n_gpus=1)


if __name__ == "__main__":
    only = sys.argv[1:] or None
    for i, cfg in enumerate(CONFIGS + EXTRA):
        if (only and str(i + 1) not in only) or (not only and i >= len(CONFIGS)):
            continue
        print(json.dumps(run(cfg)), flush=True)
