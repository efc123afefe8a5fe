"""Golden-vector tests.  Every committed .npz carries source = "oracle": the vectors are OUTPUTS OF THE CPU ORACLE
(tests/golden/make_golden.py), not of the reference — GPflow 1.1.1 / TF 1.8 cannot run here, DESIGN.md section 3 ("parity
unpinned").  CPU: the oracle is stable against its own committed vectors (a regression guard, not a parity claim).
GPU: the HIP path reproduces them through the host mirror / C-ABI (HIP == oracle)."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import model as OM
from tests.golden import cases, extras

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(HERE, f"golden_{name}.npz"))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_is_stable_against_its_committed_vectors(name):
    g = _load(name)
    c, X, Y, Z, specs, zs = cases.inputs(name)
    spec, state, _, X, Y, zs, c = cases.build(name)
    assert_allclose(OM.elbo(spec, state, X, Y, zs, c["S"], num_data=c["num_data"]), g["elbo"], rtol=1e-10)
    _, Fm, Fv = OM.propagate(spec, state, X, zs, c["S"])
    assert_allclose(Fm[-1], g[f"Fmean{c['L'] - 1}"], rtol=1e-9, atol=1e-11)
    assert_allclose(Fv[-1], g[f"Fvar{c['L'] - 1}"], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.CASES))
def test_hip_matches_oracle_vectors(name):
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


# ---------------------------------------------------------------- the wider set (tests/golden/extras.py): predict_y / density, full_cov,
# three Adam steps, one natural-gradient step
@pytest.mark.parametrize("name", ["svgp_matern52", "two_layer_1d", "ard_white_sum", "multiclass", "bernoulli"])
def test_oracle_is_stable_against_its_committed_extras(name):
    g = _load(name)
    spec, state, _, X, Y, zs, c = cases.build(name)
    got = extras.oracle_extras(spec, state, X, Y, zs, c)
    assert sorted(k for k in g.files if k.startswith("x.")) == sorted(got)
    for k, v in got.items():
        assert_allclose(v, g[k], rtol=1e-9, atol=1e-11, err_msg=k)


def test_every_fixture_holds_the_extras():
    for name in cases.CASES:
        g = _load(name)
        assert str(g["source"]).startswith("oracle"), name      # a fixture computed by the reference itself would say "reference"
        need = {"x.predy_mean", "x.predy_var", "x.preddens", "x.fc_Fmean", "x.fc_Fvar", "x.adam3_elbo", "x.adam3_q_mu"}
        assert need <= set(g.files), name
        if extras._is_gaussian(cases.CASES[name]):
            assert "x.ng_q_mu" in g.files and ("x.ng_q_sqrt" in g.files or "x.ng_q_sqrt_norm" in g.files), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.CASES))
def test_hip_matches_oracle_extras(name):
    from doubly_stochastic_dgp.training import NatGradOptimizer
    g = _load(name)
    spec, state, model, X, Y, zs, c = cases.build(name)
    S, L = c["S"], c["L"]
    tol = 1e-6 if c.get("demo_scale") else 1e-8

    def close(a, key, t=tol):
        ref = g[key]
        assert np.max(np.abs(np.asarray(a) - ref)) <= t * (np.max(np.abs(ref)) + 1e-12), key

    Fmean, Fvar = model._build_predict(X, S=S, zs=zs)
    pm, pv = model.likelihood.predict_mean_and_var(Fmean, Fvar)                       # dgp.py:117-119
    close(pm, "x.predy_mean"); close(pv, "x.predy_var")
    dens = model.likelihood.predict_density_logmeanexp(Fmean, Fvar, np.asarray(Y, dtype=np.float64))   # dgp.py:121-126
    close(dens, "x.preddens")
    _, Fmc, Fvc = model.propagate(X, full_cov=True, S=S, zs=zs)                        # dgp.py:113-114
    close(Fmc[-1], "x.fc_Fmean", 10 * tol); close(Fvc[-1], "x.fc_Fvar", 10 * tol)
    if extras._is_gaussian(c):                                                        # one natural-gradient step, gamma = 0.1
        last = model.layers[-1]
        NatGradOptimizer(extras.NG_GAMMA).minimize(model, var_list=[(last.q_mu, last.q_sqrt)], maxiter=1, X=X, Y=Y, zs=zs)
        model.engine().sync_to_host()
        ngt = 1e-4 if c.get("demo_scale") else 1e-6
        close(last.q_mu.value, "x.ng_q_mu", ngt)
        sq = np.asarray(last.q_sqrt.value)
        if "x.ng_q_sqrt" in g.files:
            close(sq, "x.ng_q_sqrt", ngt)
        else:
            assert_allclose(np.linalg.norm(sq), g["x.ng_q_sqrt_norm"], rtol=ngt)
            close(sq[:, :16, :16], "x.ng_q_sqrt_block", ngt)
    # three Adam(0.01) steps on -ELBO with the same draws (a fresh model: the natural-gradient step above moved the last layer)
    spec, state, model, X, Y, zs, c = cases.build(name)
    for _ in range(extras.ADAM_STEPS):
        model.train_step(extras.ADAM_LR, X=X, Y=Y, zs=zs)
    elbo = model._build_likelihood(X, Y, zs=zs)
    assert_allclose(elbo, g["x.adam3_elbo"], rtol=1e-5 if c.get("demo_scale") else 1e-7)
    model.engine().sync_to_host()
    close(model.layers[-1].q_mu.value, "x.adam3_q_mu", 1e-5 if c.get("demo_scale") else 1e-6)


# ---------------------------------------------------------------- full-size fixtures (tests/golden/full_cases.py; GPU side: tests/test_gpu_full_size.py)
def test_full_size_fixtures_are_complete_and_say_where_they_come_from():
    from tests.golden import full_cases as FC
    for name, c in FC.FULL.items():
        g = np.load(os.path.join(HERE, f"golden_full_{name}.npz"))
        assert str(g["source"]) == "oracle", name
        assert g["kls"].shape == (c["L"],) and np.isfinite(float(g["elbo"]))
        for l in range(c["L"]):
            assert any(k.startswith(f"grad.l{l}.q_sqrt.") for k in g.files), (name, l)
            assert any(k.startswith(f"Fvar{l}.") for k in g.files), (name, l)
        assert ("ng.q_mu.full" in g.files or "ng.q_mu.norm" in g.files) == bool(c.get("natgrad"))


def test_full_size_generator_and_test_build_the_same_model():
    """the generator (no device model) and the GPU test (tests/helpers.make_case) draw identical parameters from the case's seed; the
    oracle at the smallest full-size case reproduces its stored summary"""
    from tests.golden import full_cases as FC, make_golden_full as MG
    spec, state, model, X, Y, zs, c = FC.build("cfg2_unscaled")
    spec2, state2, X2, Y2, zs2, _ = MG.oracle_case("cfg2_unscaled")
    assert all(np.array_equal(state[k], state2[k]) for k in state) and np.array_equal(X, X2) and np.array_equal(Y, Y2)
    g = np.load(os.path.join(HERE, "golden_full_cfg2_unscaled.npz"))
    elbo, grad = OM.elbo_and_grad(spec, state, X, Y, zs, c["S"], num_data=c["num_data"])
    assert_allclose(elbo, float(g["elbo"]), rtol=1e-11)
    for k, v in grad.items():
        FC.compare("grad." + k, v, g, 1e-9, "oracle")
