"""HIP path against the CPU oracle at the FULL sizes of BASELINE.json configs[1..4].

The oracle's numbers were computed once in the build container (tests/golden/make_golden_full.py, minutes per case) and are
committed as tests/golden/golden_full_<case>.npz; here the same seeded inputs are rebuilt and the device path is compared with
them: ELBO, per-layer KL, every layer's mean / variance, EVERY gradient block (norm, a fixed pseudo-random projection over
all entries, leading and trailing corners) and, for config 5, the last layer after one natural-gradient step.  These are the
sizes at which the 8 / 16-wave chains run several rounds, the d-split hand-over, the multi-round split-K and the two-round
reduction engage — the slices of tests/test_golden.py stay below all of them.

Tolerances (fp64): ELBO 1e-9 relative; activations 1e-9 of the layer's scale; gradients 1e-7 of the block's scale (the
reference's own bar is 1e-6 / 1e-7, tests/test_dgp.py:101-106)."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from tests.golden import full_cases as FC

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(HERE, f"golden_full_{name}.npz")
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run python -m tests.golden.make_golden_full {name} in the build container")
    return np.load(path)


@pytest.mark.parametrize("name", list(FC.FULL))
def test_full_size_against_oracle_fixture(name):
    g = _load(name)
    spec, state, model, X, Y, zs, c = FC.build(name)
    S, L = c["S"], c["L"]
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(L):
        FC.compare(f"Fmean{l}", Fm[l], g, 1e-9, name)
        FC.compare(f"Fvar{l}", Fv[l], g, 1e-9, name)
    assert_allclose([layer.KL() for layer in model.layers], g["kls"], rtol=1e-9)
    elbo = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(elbo, float(g["elbo"]), rtol=1e-9)
    grads = model.engine().gradient_dict()
    keys = sorted({k.split(".", 1)[1].rsplit(".", 1)[0] for k in g.files if k.startswith("grad.")})
    assert keys and set(keys) <= set(grads), (keys, sorted(grads))
    for k in keys:
        FC.compare("grad." + k, -np.asarray(grads[k]), g, 1e-7, name)      # the device gradient is of loss = -ELBO
    if c.get("natgrad"):
        from doubly_stochastic_dgp.training import NatGradOptimizer
        last = model.layers[-1]
        NatGradOptimizer(c["natgrad"]).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
        FC.compare("ng.q_mu", last.q_mu.value, g, 1e-7, name)
        FC.compare("ng.q_sqrt", last.q_sqrt.value, g, 1e-7, name)
