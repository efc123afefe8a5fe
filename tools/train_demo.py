#!/usr/bin/env python
"""Sanity run of the whole training loop on the cfg-2 workload (synthetic kin8nm-shaped data): Adam on everything with
natural-gradient steps on the last layer interleaved (demo_regression_UCI.ipynb:360-366), ELBO and test log-likelihood
printed as it goes.  Usage: python tools/train_demo.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic data + Z recipe of the benchmark)
from doubly_stochastic_dgp.dgp import DGP  # noqa: E402
from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian  # noqa: E402


def main(steps):
    X, Y = bench.make_synthetic(8192, 8, seed=0)
    Xs, Ys, X, Y = X[7372:], Y[7372:], X[:7372], Y[:7372]
    Z = bench.default_Z(X, 128, seed=0)
    model = DGP(X, Y, Z, [RBF(8), RBF(8), RBF(8)], Gaussian(variance=0.1), num_samples=20, minibatch_size=1000)
    for layer in model.layers[:-1]:
        layer.q_sqrt = layer.q_sqrt.value * 1e-5
    eng = model.engine()
    last = len(model.layers) - 1
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        if it % 2 == 0:      # natural-gradient step on the last layer's (q_mu, q_sqrt), then Adam on everything
            model._build_likelihood(with_grad=True)
            eng.natgrad_step(last, 0.05, check=False)
        elbo = model.train_step(0.01, sync=(it % 250 == 0))
        if it % 250 == 0:
            ll = np.mean(model.predict_density(Xs, Ys, 50))
            print(f"step {it:5d}  elbo {elbo:12.3f}  test log-lik {ll:8.4f}  lik var {float(model.likelihood.likelihood.variance.value):.4f}"
                  f"  {it / (time.perf_counter() - t0):7.1f} it/s", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
