"""Round 4: the (q_mu, q_sqrt)-only reverse pass (dsdgp_model_set_grad_q_only — what tf.gradients does for NatGradOptimizer's var_list,
demos/demo_regression_UCI.ipynb:360-366): the lowest layer of the pass runs no backward chain and only the products that read the
forward pass's A; its q gradients must be the ones of the full reverse pass, bit for bit."""
import numpy as np
import pytest

from doubly_stochastic_dgp import _lib
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu


def _model(rng, L=3, N=160, D=4, M=40, S=5, DY=2):
    X, Y = rng.randn(N, D), rng.randn(N, DY)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)][:L]
    _, _, model = make_case(X, Y, Z, specs, S=S, num_data=N)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, DY)]
    return model, X, Y, zs


@pytest.mark.parametrize("force", ["gemm_mp=0", "gemm_mp=16"])
@pytest.mark.parametrize("first", [0, 1, 2])
def test_q_only_gradients_equal_the_full_reverse_pass(monkeypatch, force, first):
    monkeypatch.setenv("DSDGP_FORCE", force)
    model, X, Y, zs = _model(np.random.RandomState(21))
    e_full = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    g_full = {k: np.asarray(v).copy() for k, v in model.engine().gradient_dict().items()}
    e_q = model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=first, grad_q_only=True)
    g_q = model.engine().gradient_dict()
    assert e_q == e_full

    def same(a, b, k):
        if first > 0:
            assert np.array_equal(a, b), k
        else:       # a first layer that skips its chain gets its upstream adjoints from k_adj_prep instead of the chain's prologue:
            assert np.max(np.abs(a - b)) <= 1e-12 * (np.max(np.abs(b)) + 1e-300), k      # the S samples are summed in another order

    for l in range(first, 3):
        for name in ("q_mu", "q_sqrt"):
            k = f"l{l}.{name}"
            same(np.asarray(g_q[k]), g_full[k], k)
    # layers above the lowest one of the pass ran their chains: every entry of theirs is complete
    for l in range(first + 1, 3):
        for k in g_full:
            if k.startswith(f"l{l}."):
                assert np.array_equal(np.asarray(g_q[k]), g_full[k]), k


def test_an_adam_step_is_refused_after_a_q_only_gradient_and_accepted_after_a_full_one():
    model, X, Y, zs = _model(np.random.RandomState(22), L=2)
    eng = model.engine()
    model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=0, grad_q_only=True)
    with pytest.raises(_lib.DsdgpError):
        eng.adam_step(0.01)
    model._build_likelihood(X, Y, zs=zs, with_grad=True)
    eng.adam_step(0.01)
    model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)          # the one-call step resets the restriction by itself


# ---------------------------------------------------------------- M > 1024 (GEMM-formulated passes only; white = False)
@pytest.mark.parametrize("M,white", [(1100, False), (1300, False), (1100, True)])          # padded to 1152 = 9 x 128 and 1408 = 11 x 128
def test_more_than_1024_inducing_points(M, white):
    """Above the chain kernels' M = 1024 (LDS-resident activations) a model runs the GEMM-formulated passes, the multi-workgroup
    Cholesky / inverse and the grouped parameter products at the larger size: layer outputs, ELBO, every gradient block and (white =
    False) a natural-gradient step of the last layer against the oracle (as tests/test_gpu_parity.py does at M = 1024)."""
    from numpy.testing import assert_allclose
    from oracle import dgp_oracle as O
    from oracle import model as OM
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(43)
    N, D, S = 48, 8, 2
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = rng.randn(M, D) * 1.5
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("matern52", D, 0.9, 1.3)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=5000, white=white)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(2):
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=5000)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - grads[k])) <= 1e-6 * (np.max(np.abs(g[k])) + 1e-12), k
    if not white:
        mu, sq = O.natgrad_step(state["l1.q_mu"], state["l1.q_sqrt"], -g["l1.q_mu"], -g["l1.q_sqrt"], 0.1)
        last = model.layers[-1]
        NatGradOptimizer(0.1).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
        assert_allclose(last.q_mu.value, mu, rtol=1e-6, atol=1e-8)
        assert_allclose(last.q_sqrt.value, sq, rtol=1e-6, atol=1e-8)
    e0 = model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)
    assert np.isfinite(e0)


@pytest.mark.parametrize("M,white,force", [(2100, True, None), (2100, False, None), (1100, False, "gemm_mp=0")])
def test_unsupported_inducing_counts_fail_loudly(monkeypatch, M, white, force):
    if force:
        monkeypatch.setenv("DSDGP_FORCE", force)
    rng = np.random.RandomState(44)
    N, D = 16, 3
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = rng.randn(M, D) * 2.0
    with pytest.raises((_lib.DsdgpError, NotImplementedError)):
        _, _, model = make_case(X, Y, Z, [kern_spec("rbf", D, 1.0, 1.0)], S=1, num_data=N, white=white)
        model._build_likelihood(X, Y, zs=[rng.randn(1, N, 1)])
