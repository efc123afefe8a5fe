"""The GEMM-formulated layer passes (csrc/layer_gemm.hip: the default from Mp = 512) forced onto oracle-checkable shapes.

`DSDGP_FORCE=gemm_mp=16` (read when the device model is created) routes EVERY layer (white = False and True) through layer_fwd_gemm_launch /
layer_bwd_gemm_launch: the K(Z, X) tile kernel, the triangular / batched / reduce-mode k_pgemm launches with their column-norm
epilogues, the thin products, the forward epilogue, the element-wise reverse kernel and the per-row kernel — at the shapes of the
golden fixtures, where every layer mean / variance / sample, the ELBO and every gradient block have oracle values.  The same path at
its production sizes is covered by tests/test_gpu_full_size.py (config-4 and config-5 shards) and the large-M tests of
test_gpu_parity.py / test_gpu_round2.py.

Tolerances as in tests/test_golden.py: 1e-9 (1e-7 where the demo's q_sqrt * 1e-5 makes the variance cancellation-dominated), gradients
1e-7 (1e-5) of the block's largest entry."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import dgp_oracle as O
from oracle import model as OM
from tests.golden import cases
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FORCE = "gemm_mp=16"


@pytest.mark.parametrize("name", list(cases.CASES))
def test_gemm_path_matches_oracle_vectors(monkeypatch, name):
    monkeypatch.setenv("DSDGP_FORCE", FORCE)
    g = np.load(os.path.join(HERE, f"golden_{name}.npz"))
    spec, state, model, X, Y, zs, c = cases.build(name)
    S, L = c["S"], c["L"]
    tol = 1e-7 if c.get("demo_scale") else 1e-9
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(L):
        assert_allclose(Fm[l], g[f"Fmean{l}"], rtol=tol, atol=tol * 0.1)
        assert_allclose(Fv[l], g[f"Fvar{l}"], rtol=tol, atol=tol * 0.1)
        assert_allclose(Fs[l], g[f"F{l}"], rtol=tol, atol=tol * 0.1)
    elbo = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(elbo, g["elbo"], rtol=tol)
    grads = model.engine().gradient_dict()
    gtol = 1e-5 if c.get("demo_scale") else 1e-7
    checked = 0
    for key in g.files:
        if key.startswith("grad."):
            k, ref = key[5:], -g[key]
            assert np.max(np.abs(grads[k] - ref)) <= gtol * (np.max(np.abs(ref)) + 1e-12), k
            checked += 1
        elif key.startswith("gradnorm."):
            k = key[9:]
            assert_allclose(np.linalg.norm(grads[k]), g[key], rtol=gtol)
            ref = -g["gradblock." + k]
            assert np.max(np.abs(grads[k][:, :16, :16] - ref)) <= gtol * (np.max(np.abs(ref)) + 1e-12), k
            checked += 1
    assert checked >= 4 * L


def test_gemm_path_training_steps_and_predictions(monkeypatch):
    """Three Adam steps on minibatches gathered on the device, then predict_f: the GEMM-formulated model follows the chain model
    (same seeds, same Philox draws) to rounding — the two differ only in summation order."""
    rng = np.random.RandomState(11)
    N, D, M, S = 300, 4, 48, 6
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2)]
    out = {}
    for tag, force in (("chain", "gemm_mp=0"), ("gemm", FORCE)):
        monkeypatch.setenv("DSDGP_FORCE", force)
        _, _, model = make_case(X, Y, Z, specs, S=S, num_data=N, minibatch_size=128)
        for _ in range(3):
            model.train_step(0.01)
        elbo = model.train_step(0.01, sync=True)
        mean, var = model.predict_f(X[:50], 5)
        out[tag] = (elbo, np.asarray(mean), np.asarray(var), model.layers[0].q_mu.value.copy(), model.layers[1].feature.Z.value.copy())
    for a, b in zip(out["chain"], out["gemm"]):
        assert_allclose(b, a, rtol=1e-8, atol=1e-10)
    assert not np.array_equal(out["chain"][1], out["gemm"][1])      # (a different summation order: the forced path really ran)


def test_gemm_path_natural_gradient_step(monkeypatch):
    """NatGradOptimizer on the last layer of a GEMM-formulated model against the oracle's step (the pruned reverse pass + the
    one-factorisation update of model_extras.hpp)."""
    from doubly_stochastic_dgp.training import NatGradOptimizer
    monkeypatch.setenv("DSDGP_FORCE", FORCE)
    rng = np.random.RandomState(12)
    N, D, M, S = 40, 3, 24, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D) * 1.5
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("rbf", D, 1.2, 0.8)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=200)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    _, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=200)
    mu, sq = O.natgrad_step(state["l1.q_mu"], state["l1.q_sqrt"], -g["l1.q_mu"], -g["l1.q_sqrt"], 0.1)
    last = model.layers[-1]
    NatGradOptimizer(0.1).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    assert_allclose(last.q_mu.value, mu, rtol=1e-6, atol=1e-8)
    assert_allclose(last.q_sqrt.value, sq, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("mean_linear", [False, True])
def test_gemm_path_wide_output_layer(monkeypatch, mean_linear):
    """More than 32 outputs per layer (q_mu^T a and the Linear mean function then take k_pgemm / the per-row form instead of the thin
    kernel): ELBO and every gradient block of the GEMM-formulated model against the chain model on the same inputs."""
    rng = np.random.RandomState(13)
    N, D, M, S, DO = 40, 5, 20, 2, 40
    X, Y = rng.randn(N, D), rng.randn(N, DO)
    Z = rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 1.2)] if not mean_linear else [kern_spec("rbf", D, 1.0, 1.2), kern_spec("rbf", 3, 1.0, 1.0)]
    Yt = Y if not mean_linear else Y[:, :2]
    zs = [rng.randn(S, N, DO)] if not mean_linear else [rng.randn(S, N, 3), rng.randn(S, N, 2)]
    out = {}
    for tag, force in (("chain", "gemm_mp=0"), ("gemm", FORCE)):
        monkeypatch.setenv("DSDGP_FORCE", force)
        _, _, model = make_case(X, Yt, Z, specs, S=S, num_data=100)
        e = model._build_likelihood(X, Yt, zs=zs, with_grad=True)
        out[tag] = (e, {k: np.asarray(v).copy() for k, v in model.engine().gradient_dict().items()})
    assert_allclose(out["gemm"][0], out["chain"][0], rtol=1e-10)
    for k, v in out["chain"][1].items():
        assert np.max(np.abs(out["gemm"][1][k] - v)) <= 1e-8 * (np.max(np.abs(v)) + 1e-12), k
