"""Golden-vector tests.  CPU: the oracle still reproduces the committed vectors (tests/golden/make_golden.py).
GPU: the HIP path reproduces them through the host mirror / C-ABI."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import model as OM
from tests.golden import cases

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(HERE, f"golden_{name}.npz"))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_reproduces_golden(name):
    g = _load(name)
    c, X, Y, Z, specs, zs = cases.inputs(name)
    spec, state, _, X, Y, zs, c = cases.build(name)
    assert_allclose(OM.elbo(spec, state, X, Y, zs, c["S"], num_data=c["num_data"]), g["elbo"], rtol=1e-10)
    _, Fm, Fv = OM.propagate(spec, state, X, zs, c["S"])
    assert_allclose(Fm[-1], g[f"Fmean{c['L'] - 1}"], rtol=1e-9, atol=1e-11)
    assert_allclose(Fv[-1], g[f"Fvar{c['L'] - 1}"], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.CASES))
def test_hip_reproduces_golden(name):
    g = _load(name)
    spec, state, model, X, Y, zs, c = cases.build(name)
    S, L = c["S"], c["L"]
    tol = 1e-7 if c.get("demo_scale") else 1e-9      # q_sqrt*1e-5 makes var cancellation-dominated
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(L):
        assert_allclose(Fm[l], g[f"Fmean{l}"], rtol=tol, atol=tol * 0.1)
        assert_allclose(Fv[l], g[f"Fvar{l}"], rtol=tol, atol=tol * 0.1)
        assert_allclose(Fs[l], g[f"F{l}"], rtol=tol, atol=tol * 0.1)
    assert_allclose([layer.KL() for layer in model.layers], g["kls"], rtol=1e-9)
    elbo = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(elbo, g["elbo"], rtol=tol)
    grads = model.engine().gradient_dict()
    gtol = 1e-5 if c.get("demo_scale") else 1e-7
    for key in g.files:
        if key.startswith("grad."):
            k = key[5:]
            ref = -g[key]
            assert np.max(np.abs(grads[k] - ref)) <= gtol * (np.max(np.abs(ref)) + 1e-12), k
        elif key.startswith("gradnorm."):
            k = key[9:]
            assert_allclose(np.linalg.norm(grads[k]), g[key], rtol=gtol)
            ref = -g["gradblock." + k]
            assert np.max(np.abs(grads[k][:, :16, :16] - ref)) <= gtol * (np.max(np.abs(ref)) + 1e-12), k
