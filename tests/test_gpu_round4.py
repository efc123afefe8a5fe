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
