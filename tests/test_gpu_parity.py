"""-m gpu parity tests: HIP path (through the C-ABI / host mirror) vs the CPU oracle on identical seeded inputs.

Tolerances (fp64; Cholesky is not bit-stable — SURVEY §8c): primitives rtol 1e-11..1e-9 (stated per test);
per-layer mean/var rtol 1e-9 atol 1e-10; ELBO rtol 1e-9; gradients rtol 1e-7 of the largest entry (the oracle side is
torch autograd of the *reference-form* op sequence, which itself carries ~1e-10 cancellation noise).
"""
import ctypes as C

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec, make_case, product_kernel, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from doubly_stochastic_dgp.engine import Context
    return Context.get()


def _dev(ctx, a):
    return ctx.to_device(np.ascontiguousarray(a, dtype=np.float64))


def _p(t):
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------- primitives
@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(16, 16, 4), (50, 37, 19), (128, 128, 128), (200, 65, 130)])
def test_gemm(ctx, tA, tB, m, n, k):
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(m + n + k)
    A = rng.randn(*((k, m) if tA else (m, k)))
    B = rng.randn(*((n, k) if tB else (k, n)))
    Cc = rng.randn(m, n)
    dA, dB, dC = _dev(ctx, A), _dev(ctx, B), _dev(ctx, Cc)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gemm(ctx.handle, tA, tB, m, n, k, 1.5, _p(dA), A.shape[1], _p(dB), B.shape[1], -0.5, _p(dC), n))
    ctx.sync()
    ref = 1.5 * (A.T if tA else A) @ (B.T if tB else B) - 0.5 * Cc
    assert_allclose(dC.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * k)


# 570 -> 576 = 4 x 128 + 64: ragged last diagonal block; 177 .. : the blocked look-ahead sequence (180 -> 192 = 128 + 64); 1536: 12 blocks;
# 2176 = 17 blocks: the plain blocked sequence (the look-ahead form stops at 16)
@pytest.mark.parametrize("n", [5, 16, 19, 50, 100, 128, 176, 180, 192, 200, 256, 300, 320, 448, 512, 570, 600, 1024, 1536, 2176])
def test_potrf(ctx, n):
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n)
    A0 = rng.randn(n, n)
    A0 = A0 @ A0.T + n * np.eye(n)
    batch = 3
    A = np.stack([A0 * (1 + b) for b in range(batch)])
    dA = _dev(ctx, A)
    info = C.c_int(-1)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_potrf(ctx.handle, batch, n, _p(dA), n, n * n, C.byref(info)))
    ctx.sync()
    L = dA.cpu().numpy()
    assert info.value == 0
    for b in range(batch):
        assert np.allclose(np.triu(L[b], 1), 0.0)
        assert np.linalg.norm(L[b] @ L[b].T - A[b]) / np.linalg.norm(A[b]) <= 1e-14 * n      # SURVEY §8c bar
        assert_allclose(L[b], np.linalg.cholesky(A[b]), rtol=1e-11, atol=1e-11)


def test_potrf_not_spd(ctx):
    from doubly_stochastic_dgp import _lib
    A = np.eye(20)
    A[7, 7] = -1.0
    dA = _dev(ctx, A)
    info = C.c_int(0)
    ctx.torch.cuda.current_stream().synchronize()
    rc = ctx.lib.dsdgp_potrf(ctx.handle, 1, 20, _p(dA), 20, 400, C.byref(info))
    assert rc == _lib.ERR_NOT_SPD and info.value == 8


@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("n,nrhs", [(19, 7), (128, 1000), (50, 333)])
def test_trsm(ctx, trans, n, nrhs):
    import scipy.linalg as sla
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n)
    L = np.tril(rng.randn(n, n)) + 4 * np.eye(n)
    B = rng.randn(n, nrhs)
    dL, dB = _dev(ctx, L), _dev(ctx, B)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_trsm(ctx.handle, trans, n, nrhs, _p(dL), n, _p(dB), nrhs))
    ctx.sync()
    ref = sla.solve_triangular(L, B, lower=True, trans=trans)
    assert_allclose(dB.cpu().numpy(), ref, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
@pytest.mark.parametrize("ard", [False, True])
def test_gram(ctx, kind, ard):
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(3)
    D, n, n2 = 8, 37, 1003
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = (0.5 + rng.rand(D)) if ard else np.array([0.8])
    k = O.Kern(kind, D, variance=1.7, lengthscales=ls if ard else float(ls[0]), ARD=ard, white_variance=0.3)
    spec = _lib.KernelSpec(kind={"rbf": 0, "matern52": 1}[kind], input_dim=D, ard=int(ard), has_white=1, variance=1.7,
                           white_variance=0.3, lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    dX, dX2 = _dev(ctx, X), _dev(ctx, X2)
    out = ctx.empty(n, n2)
    outs = ctx.empty(n, n)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, _p(dX2), n2, 0.0, _p(out), n2))
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, None, 0, 1e-6, _p(outs), n))
    ctx.sync()
    # direct-difference r2 (HIP) vs expand-the-square r2 (GPflow form in the oracle): ~1e-15 * |x|^2 apart
    assert_allclose(out.cpu().numpy(), k.K(O.NP, X, X2), rtol=1e-11, atol=1e-13)
    Ks = outs.cpu().numpy()
    assert_allclose(Ks, k.K(O.NP, X) + 1e-6 * np.eye(n), rtol=1e-11, atol=1e-13)
    assert np.array_equal(Ks, Ks.T)


def test_randn_statistics(ctx):
    from doubly_stochastic_dgp import _lib
    n = 1 << 20
    out = ctx.empty(n)
    _lib.check(ctx.lib.dsdgp_randn(ctx.handle, C.c_uint64(7), C.c_uint64(3), n, _p(out)))
    ctx.sync()
    z = out.cpu().numpy()
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1) < 0.01
    assert abs(np.mean(z ** 3)) < 0.02 and abs(np.mean(z ** 4) - 3) < 0.05
    out2 = ctx.empty(n)
    _lib.check(ctx.lib.dsdgp_randn(ctx.handle, C.c_uint64(7), C.c_uint64(3), n, _p(out2)))
    ctx.sync()
    assert np.array_equal(z, out2.cpu().numpy())                      # counter-based: reproducible


# ---------------------------------------------------------------- layer level (layers.py:178-246)
CASES = [
    # M, D_in, D_out, kind, white, ARD
    (19, 2, 3, "matern52", True, False),
    (19, 2, 3, "matern52", False, False),
    (50, 8, 1, "rbf", False, False),
    (100, 5, 5, "rbf", False, True),
    (128, 8, 8, "rbf", False, False),
    (128, 8, 8, "rbf", True, False),
    (200, 9, 9, "rbf", False, False),
]


@pytest.mark.parametrize("M,Din,Dout,kind,white,ard", CASES)
def test_layer_conditional_and_KL(M, Din, Dout, kind, white, ard):
    rng = np.random.RandomState(M + Din)
    N = 333
    X, Y = rng.randn(N, Din), rng.randn(N, Dout)
    Z = X[rng.permutation(N)[:M]] + 0.01 * rng.randn(M, Din)
    ls = (0.8 + rng.rand(Din)) if ard else 1.1
    specs = [kern_spec(kind, Din, 1.3, ls, ard, white_variance=0.05)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=1)
    om = OM.build(O.NP, spec, state)
    layer_o = om.layers[0]
    Xs = rng.randn(77, Din)
    mo, vo = layer_o.conditional_ND(O.NP, Xs)
    m, v = model.layers[0].conditional_ND(Xs)
    assert_allclose(m, mo, rtol=1e-9, atol=1e-10)
    assert_allclose(v, vo, rtol=1e-9, atol=1e-10)
    assert_allclose(model.layers[0].KL(), layer_o.KL(O.NP), rtol=1e-10)


def test_cholesky_failure_raises():
    from doubly_stochastic_dgp import _lib, settings
    rng = np.random.RandomState(0)
    X = rng.randn(30, 2)
    Z = np.concatenate([X[:10], X[:10]])                               # duplicated inducing points, no jitter
    specs = [kern_spec("rbf", 2)]
    with settings.temp_jitter(1e-6):
        spec, state, model = make_case(X, rng.randn(30, 1), Z, specs, S=1, randomize=False)
    with settings.temp_jitter(-1e-3):
        with pytest.raises(_lib.CholeskyError):
            model.layers[0].conditional_ND(X)
        # asynchronous training steps surface the failure at the next synchronising step / at the end of minimize()
        from doubly_stochastic_dgp.training import AdamOptimizer
        model.train_step(0.01)
        with pytest.raises(_lib.CholeskyError):
            model.train_step(0.01, sync=True)
        with pytest.raises(_lib.CholeskyError):
            AdamOptimizer(0.01).minimize(model, maxiter=3)


# ---------------------------------------------------------------- model level (dgp.py:61-98)
def _three_layer(N=200, D=8, M=128, S=4, kind="rbf", q_sqrt_scale=None, seed=0, num_data=None):
    rng = np.random.RandomState(seed)
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[rng.permutation(N)[:M]] + 0.01 * rng.randn(M, D) if M <= N else rng.randn(M, D)
    specs = [kern_spec(kind, D, 1.0, 1.0)] * 3
    spec, state, model = make_case(X, Y, Z, specs, S=S, q_sqrt_scale=q_sqrt_scale, num_data=num_data, seed=seed)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 1)]
    return X, Y, spec, state, model, zs


def test_propagate_three_layers():
    X, Y, spec, state, model, zs = _three_layer()
    Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, 4)
    Fs, Fm, Fv = model.propagate(X, S=4, zs=zs)
    for l in range(3):
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-9, atol=1e-10)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-9, atol=1e-10)
        assert_allclose(Fs[l], Fs_o[l], rtol=1e-9, atol=1e-10)


def test_propagate_broadcast_z():
    # DGP_Quad-style z of shape (S,1,D) (dgp.py:148,153)
    X, Y, spec, state, model, zs = _three_layer(N=50, M=30, S=3)
    zb = [zs[0][:, :1, :], zs[1][:, :1, :], np.zeros((1, 1, 1))]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zb, 3)
    _, Fm, Fv = model.propagate(X, S=3, zs=zb)
    assert_allclose(Fm[-1], Fm_o[-1], rtol=1e-9, atol=1e-10)
    assert_allclose(Fv[-1], Fv_o[-1], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_elbo_value(kind):
    X, Y, spec, state, model, zs = _three_layer(kind=kind, num_data=7372)
    ref = OM.elbo(spec, state, X, Y, zs, 4, num_data=7372)
    got = model.compute_log_likelihood(X, Y, zs=zs)
    assert_allclose(got, ref, rtol=1e-9)
    e = model.E_log_p_Y(X, Y, zs=zs)
    om = OM.build(O.NP, spec, state, 4, 7372)
    assert_allclose(e, om.E_log_p_Y(O.NP, X, Y, zs), rtol=1e-9, atol=1e-10)


def test_elbo_demo_init_small_q_sqrt():
    # demos shrink inner q_sqrt by 1e-5 (demo_regression_UCI.ipynb:183): cancellation-dominated variances
    X, Y, spec, state, model, zs = _three_layer(q_sqrt_scale=1e-5)
    ref = OM.elbo(spec, state, X, Y, zs, 4)
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), ref, rtol=1e-7)


def _grad_check(X, Y, spec, state, model, zs, S, num_data=None, tol=1e-7):
    ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=num_data)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    g = model.engine().gradient_dict()
    worst = {}
    for k in gref:
        worst[k] = np.max(np.abs(-gref[k] - g[k])) / (np.max(np.abs(gref[k])) + 1e-12)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"gradient mismatch (rel to max entry): {bad}; all: {worst}"


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_gradients_three_layers(kind):
    X, Y, spec, state, model, zs = _three_layer(N=150, D=4, M=40, S=3, kind=kind, num_data=1000)
    _grad_check(X, Y, spec, state, model, zs, 3, num_data=1000)


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_gradients_white(kind):
    # white=True: dl/dKu goes through the Cholesky adjoint instead of the Ku^-1 shortcut
    rng = np.random.RandomState(9)
    N, D, M, S = 120, 3, 30, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec(kind, D, 1.1, 0.9, white_variance=0.03), kern_spec(kind, D, 0.8, 1.2)]
    spec, state, model = make_case(X, Y, Z, specs, white=True, S=S, num_data=600)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    _grad_check(X, Y, spec, state, model, zs, S, num_data=600)


def test_gradients_cfg2_shape_slice():
    X, Y, spec, state, model, zs = _three_layer(N=64, D=8, M=128, S=4, num_data=7372, seed=3)
    _grad_check(X, Y, spec, state, model, zs, 4, num_data=7372)


def test_gradients_ard_white_stepdown():
    rng = np.random.RandomState(5)
    N, S = 90, 2
    X, Y = rng.randn(N, 6), rng.randn(N, 2)
    Z = X[:25].copy()
    specs = [kern_spec("rbf", 6, 1.2, 0.8 + rng.rand(6), True, white_variance=0.02),
             kern_spec("rbf", 3, 0.9, 1.3, False, white_variance=0.01),
             kern_spec("rbf", 3, 1.1, 0.7 + rng.rand(3), True)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=500)
    zs = [rng.randn(S, N, 3), rng.randn(S, N, 3), rng.randn(S, N, 2)]
    _grad_check(X, Y, spec, state, model, zs, S, num_data=500)


def test_adam_training_steps_match_oracle():
    X, Y, spec, state, model, zs = _three_layer(N=80, D=3, M=20, S=2, num_data=400)
    keys = sorted(state.keys())
    th = {k: state[k].copy() for k in keys}
    m = {k: np.zeros_like(state[k]) for k in keys}
    v = {k: np.zeros_like(state[k]) for k in keys}
    for t in range(1, 4):
        _, g = OM.elbo_and_grad(spec, th, X, Y, zs, 2, num_data=400)
        for k in keys:
            O.adam_step(th[k], -g[k], m[k], v[k], t, lr=0.01)
        model.train_step(0.01, X=X, Y=Y, zs=zs)
    ref = OM.elbo(spec, th, X, Y, zs, 2, num_data=400)
    got = model.compute_log_likelihood(X, Y, zs=zs)
    assert_allclose(got, ref, rtol=1e-7)
    assert_allclose(model.layers[1].q_mu.value, th["l1.q_mu"], rtol=1e-6, atol=1e-8)
    assert_allclose(model.likelihood.likelihood.variance.value, O.positive_forward(O.NP, th["lik_variance_raw"]), rtol=1e-7)


def test_trainable_flags_respected():
    X, Y, spec, state, model, zs = _three_layer(N=60, D=3, M=16, S=2)
    model.layers[0].feature.Z.set_trainable(False)
    model.likelihood.likelihood.variance.set_trainable(False)
    Z0 = model.layers[0].feature.Z.read_value()
    q0 = model.layers[0].q_mu.read_value()
    model.train_step(0.01, X=X, Y=Y, zs=zs)
    assert np.array_equal(model.layers[0].feature.Z.value, Z0)
    assert not np.array_equal(model.layers[0].q_mu.value, q0)
    assert_allclose(model.likelihood.likelihood.variance.value, 0.1, rtol=1e-12)


def test_mc_elbo_unbiased_vs_explicit_z():
    # device Philox draws: the mean of many stochastic ELBOs must sit within 4 s.e. of the mean over explicit z draws
    X, Y, spec, state, model, zs = _three_layer(N=40, D=2, M=16, S=8)
    vals = np.array([model.compute_log_likelihood(X, Y) for _ in range(200)])
    rng = np.random.RandomState(11)
    ref = np.array([OM.elbo(spec, state, X, Y, [rng.randn(8, 40, 2), rng.randn(8, 40, 2), np.zeros((1, 1, 1))], 8) for _ in range(200)])
    se = np.sqrt(vals.var() / len(vals) + ref.var() / len(ref))
    assert abs(vals.mean() - ref.mean()) < 4 * se


def test_predict_wrappers():
    X, Y, spec, state, model, zs = _three_layer(N=50, D=3, M=16, S=5)
    om = OM.build(O.NP, spec, state, 5)
    _, Fm, Fv = om.propagate(O.NP, X, zs[:2] + [np.zeros((1, 1, 1))], S=5)
    lik = om.likelihood
    m, v = model._build_predict(X, S=5, zs=zs[:2] + [None])
    assert_allclose(m, Fm[-1], rtol=1e-9, atol=1e-10)
    my, vy = model.likelihood.predict_mean_and_var(m, v)
    assert_allclose(vy, Fv[-1] + 0.1, rtol=1e-9)
    dens = model.likelihood.predict_density_logmeanexp(m, v, Y)
    l = lik.predict_density(O.NP, Fm[-1], Fv[-1], Y)
    from scipy.special import logsumexp
    assert_allclose(dens, logsumexp(l - np.log(5), axis=0), rtol=1e-9, atol=1e-10)
    mf, vf = model.predict_f(X, 7)
    assert mf.shape == (7, 50, 1) and vf.shape == (7, 50, 1)


def test_minibatch_gather_pairs_rows():
    rng = np.random.RandomState(0)
    X = rng.randn(500, 3)
    Y = X[:, :1] * 2.0
    spec, state, model = make_case(X, Y, X[:10].copy(), [kern_spec("rbf", 3)], S=1, minibatch_size=64)
    for _ in range(10):
        Xb, Yb = model.next_minibatch()
        assert Xb.shape == (64, 3)
        assert np.array_equal(Yb.cpu().numpy(), Xb.cpu().numpy()[:, :1] * 2.0)


# ---------------------------------------------------------------- natural gradients
@pytest.mark.parametrize("white", [False, True])
def test_natgrad_step_matches_oracle(white):
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(12)
    N, D, M, S = 80, 2, 20, 2
    X, Y = rng.randn(N, D), rng.randn(N, 3)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("rbf", D, 1.2, 0.8)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=300)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 3)]
    _, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=300)
    mu, sq = O.natgrad_step(state["l1.q_mu"], state["l1.q_sqrt"], -g["l1.q_mu"], -g["l1.q_sqrt"], 0.1)
    last = model.layers[-1]
    NatGradOptimizer(0.1).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    assert_allclose(last.q_mu.value, mu, rtol=1e-7, atol=1e-9)
    assert_allclose(last.q_sqrt.value, sq, rtol=1e-7, atol=1e-9)
    state2 = dict(state)
    state2["l1.q_mu"], state2["l1.q_sqrt"] = mu, sq
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, state2, X, Y, zs, S, num_data=300), rtol=1e-8)


def test_natgrad_gamma1_gives_collapsed_bound():
    # tests/test_collapsed.py:57-104 on the device: gamma = 1, Gaussian likelihood, one step => SGPR collapsed bound
    import math
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(8)
    N, M, s2 = 40, 9, 0.2
    X = rng.uniform(size=(N, 1)) * 4
    Y = np.sin(2 * X) + 0.3 * rng.randn(N, 1)
    Z = np.linspace(0, 4, M)[:, None]
    specs = [kern_spec("rbf", 1, 1.3, 0.7)]
    spec, state, model = make_case(X, Y, Z, specs, S=1, lik_var=s2)
    zs = [np.zeros((1, N, 1))]
    layer = model.layers[0]
    NatGradOptimizer(1.0).minimize(model, var_list=[[layer.q_mu, layer.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    elbo = model.compute_log_likelihood(X, Y, zs=zs)
    kern = O.Kern("rbf", 1, variance=1.3, lengthscales=0.7)
    Kuu = kern.K(O.NP, Z) + 1e-6 * np.eye(M)
    Kuf = kern.K(O.NP, Z, X)
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    Cm = Qff + s2 * np.eye(N)
    _, ld = np.linalg.slogdet(Cm)
    bound = (-0.5 * Y.T @ np.linalg.solve(Cm, Y)).item() - 0.5 * ld - 0.5 * N * math.log(2 * math.pi) \
        - 0.5 / s2 * (kern.Kdiag(O.NP, X).sum() - np.trace(Qff))
    assert_allclose(elbo, bound, rtol=1e-7)


# ---------------------------------------------------------------- MultiClass / RobustMax
@pytest.mark.parametrize("white", [True, False])
def test_multiclass_elbo_gradients_and_predictions(white):
    # shapes of tests/test_dgp.py:56-63 (K = 3 classes, num_outputs = K), two layers
    rng = np.random.RandomState(21)
    N, D, M, S, K = 60, 2, 19, 3, 3
    X = rng.uniform(size=(N, D))
    Y = rng.choice([0.0, 1.0, 2.0], N).reshape(N, 1)
    Z = X[:M].copy()
    specs = [kern_spec("matern52", D, 1.0, 0.5), kern_spec("matern52", D, 1.0, 0.5)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=200, num_classes=K)
    zs = [rng.randn(S, N, D), rng.randn(S, N, K)]
    _grad_check(X, Y, spec, state, model, zs, S, num_data=200)
    om = OM.build(O.NP, spec, state, S, 200)
    assert_allclose(model.E_log_p_Y(X, Y, zs=zs), om.E_log_p_Y(O.NP, X, Y, zs), rtol=1e-9, atol=1e-11)
    _, Fm, Fv = om.propagate(O.NP, X, zs, S=S)
    m, v = model._build_predict(X, S=S, zs=zs)
    assert_allclose(model.likelihood.predict_density_logmeanexp(m, v, Y), om.predict_density(O.NP, X, Y, zs, S), rtol=1e-9,
                    atol=1e-11)
    pm, pv = model.likelihood.predict_mean_and_var(m, v)
    rm, rv = om.likelihood.predict_mean_and_var(O.NP, Fm[-1], Fv[-1])
    assert_allclose(pm, rm, rtol=1e-9, atol=1e-12)
    assert_allclose(pv, rv, rtol=1e-8, atol=1e-12)
    assert_allclose(pm.sum(-1), 1.0, atol=2e-3)      # RobustMax probabilities ~ sum to one


# ---------------------------------------------------------------- full_cov=True
@pytest.mark.parametrize("white", [False, True])
def test_full_cov_propagation(white):
    # dgp.py:104-114 / layers.py:206-209 / utils.py:43-51; reference check: tests/test_dgp.py:99,116-117
    rng = np.random.RandomState(31)
    N, D, M, S = 33, 2, 19, 2
    X = rng.uniform(size=(50, D))
    Y = rng.randn(50, 3)
    Z = X[:M].copy()
    specs = [kern_spec("rbf", D, 1.0, 0.6, white_variance=0.01), kern_spec("matern52", D, 1.2, 0.5)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S)
    Xs = rng.uniform(size=(N, D))
    zs = [rng.randn(S, N, D), rng.randn(S, N, 3)]
    Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, Xs, zs, S, full_cov=True)
    Fs, Fm, Fv = model.propagate(Xs, full_cov=True, S=S, zs=zs)
    for l in range(2):
        assert Fv[l].shape == (S, N, N, Fm[l].shape[-1])
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fs[l], Fs_o[l], rtol=1e-6, atol=1e-7)       # Cholesky of an N x N covariance with jitter 1e-6
    m, v = model.predict_f_full_cov(Xs, 1)
    assert m.shape == (1, N, 3) and v.shape == (1, N, N, 3)
    # diagonal of the full covariance == the diagonal-path variance
    md, vd = model.layers[0].conditional_ND(Xs)
    mf, vf = model.layers[0].conditional_ND(Xs, full_cov=True)
    assert_allclose(np.einsum("iid->id", vf), vd, rtol=1e-9, atol=1e-11)
    assert_allclose(mf, md, rtol=1e-10, atol=1e-12)


# ---------------------------------------------------------------- large M / wide inputs (config-4/5-shaped slices)
def test_cfg4_shape_mnist_like_multiclass():
    # BASELINE.json configs[3] shape: widths 784 -> 30 (fixed PCA Linear mean) -> 30 (Identity) -> 10, M = 512,
    # MultiClass(10) (demo_mnist.ipynb:99-104).  Slice: N = 32 of 300 rows, S = 2.
    rng = np.random.RandomState(40)
    Ndata, N, S, M, K = 300, 32, 2, 512, 10
    Xall = rng.uniform(size=(Ndata, 784)) * (rng.uniform(size=(Ndata, 784)) < 0.19)
    Yall = rng.choice(np.arange(K, dtype=np.float64), Ndata).reshape(Ndata, 1)
    Z = Xall[rng.permutation(Ndata)[:M % Ndata or Ndata]]
    Z = np.concatenate([Z, Xall[:M - Z.shape[0]] + 0.05 * rng.randn(M - Z.shape[0], 784)]) if Z.shape[0] < M else Z
    specs = [kern_spec("rbf", 784, 2.0, 2.0), kern_spec("rbf", 30, 2.0, 2.0), kern_spec("rbf", 30, 2.0, 2.0)]
    spec, state, model = make_case(Xall, Yall, Z, specs, S=S, num_data=60000, num_classes=K)
    X, Y = Xall[:N], Yall[:N]
    zs = [rng.randn(S, N, 30), rng.randn(S, N, 30), rng.randn(S, N, K)]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(3):
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)
    _grad_check(X, Y, spec, state, model, zs, S, num_data=60000, tol=1e-6)


def test_cfg5_shape_M1024():
    # BASELINE.json configs[4] shape: 3 layers 8 -> 8 -> 8 -> 1, M = 1024 (slice N = 32, S = 2) + one natural-gradient step
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(41)
    N, D, M, S = 32, 8, 1024, 2
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = rng.randn(M, D) * 1.5
    specs = [kern_spec("rbf", D, 1.0, 1.0)] * 3
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=7372)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 1)]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    assert_allclose(Fm[-1], Fm_o[-1], rtol=1e-8, atol=1e-9)
    assert_allclose(Fv[-1], Fv_o[-1], rtol=1e-8, atol=1e-9)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=7372)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:                                     # EVERY parameter block of every layer
        assert np.max(np.abs(-g[k] - grads[k])) <= 1e-6 * (np.max(np.abs(g[k])) + 1e-12), k
    mu, sq = O.natgrad_step(state["l2.q_mu"], state["l2.q_sqrt"], -g["l2.q_mu"], -g["l2.q_sqrt"], 0.1)
    last = model.layers[-1]
    NatGradOptimizer(0.1).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    assert_allclose(last.q_mu.value, mu, rtol=1e-6, atol=1e-8)
    assert_allclose(last.q_sqrt.value, sq, rtol=1e-6, atol=1e-8)


# ---------------------------------------------------------------- input propagation (layer_initializations.py:55-79)
def _input_prop_case(white=False):
    from doubly_stochastic_dgp import settings
    from doubly_stochastic_dgp.dgp import DGP_Base
    from doubly_stochastic_dgp.gpflow_compat import Gaussian
    from doubly_stochastic_dgp.layer_initializations import init_layers_input_prop
    rng = np.random.RandomState(31)
    N, D, M, S = 40, 2, 14, 3
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", 2, 1.3, 0.9), kern_spec("matern52", 5, 0.8, 1.4), kern_spec("rbf", 4, 1.1, 1.2, ARD=True)]
    np.random.seed(5)
    pads = [np.random.randn(M, k["input_dim"] - D) for k in specs]
    lds = O.init_layers_input_prop(X, Y, Z, specs, pads, white=white)
    for l in lds:
        l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
        l["q_sqrt"] = l["q_sqrt"] * 0.7 + 0.05 * np.tril(rng.randn(*l["q_sqrt"].shape))
    sl, state = OM.state_from_layers(lds, lik_variance=0.2)
    spec = dict(jitter=1e-6, white=white, likelihood="gaussian", layers=sl, num_classes=None)
    np.random.seed(5)
    with settings.temp_jitter(1e-6):
        layers = init_layers_input_prop(X, Y, Z, [product_kernel(k) for k in specs], white=white)
        model = DGP_Base(X, Y, Gaussian(variance=0.2), layers, num_samples=S, num_data=123)
    for l, layer in zip(lds, model.layers):
        assert_allclose(layer.feature.Z.value, l["Z"], rtol=0, atol=0)
        layer.q_mu = l["q_mu"]
        layer.q_sqrt = l["q_sqrt"]
    zs = [rng.randn(S, N, 3), rng.randn(S, N, 2), rng.randn(S, N, 1)]
    return X, Y, spec, state, model, zs, S


@pytest.mark.parametrize("white", [False, True])
def test_input_propagation_matches_oracle(white):
    X, Y, spec, state, model, zs, S = _input_prop_case(white)
    assert [l.input_prop_dim for l in model.layers] == [2, 2, None]
    Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    assert Fs[0].shape == (S, 40, 5) and Fs[1].shape == (S, 40, 4) and Fs[2].shape == (S, 40, 1)
    for l in range(3):
        assert_allclose(Fs[l], Fs_o[l], rtol=1e-9, atol=1e-10)
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-9, atol=1e-10)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-9, atol=1e-10)
    assert np.all(Fs[1][:, :, :2] == X[None]) and np.all(Fv[0][:, :, :2] == 0.0)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=123)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-7, (k, err)


def test_input_propagation_full_cov():
    X, Y, spec, state, model, zs, S = _input_prop_case()
    Xs = X[:9]
    z9 = [z[:, :9] for z in zs]
    Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, Xs, z9, S, full_cov=True)
    Fs, Fm, Fv = model.propagate(Xs, full_cov=True, S=S, zs=z9)
    for l in range(3):
        assert Fv[l].shape == Fv_o[l].shape
        assert_allclose(Fs[l], Fs_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)


# ---------------------------------------------------------------- DGP_Quad (dgp.py:129-166)
@pytest.mark.parametrize("num_classes", [None, 3])
def test_dgp_quad_matches_oracle(num_classes):
    # quadrature over the inner layers: (S,1,D) Gauss-Hermite z's and weights in place of the MC mean — value, E_log_p_Y
    # and every gradient against the oracle's weighted restatement
    from doubly_stochastic_dgp.dgp import DGP_Quad
    rng = np.random.RandomState(21)
    N, D, M, H = 30, 2, 12, 7
    X = rng.randn(N, D)
    Y = rng.randint(0, 3, size=(N, 1)).astype(float) if num_classes else rng.randn(N, 1)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 0.8), kern_spec("matern52", D, 1.2, 1.1)]
    spec, state, model = make_case(X, Y, Z, specs, S=1, num_data=90, num_classes=num_classes)
    quad = DGP_Quad(X, Y, model.likelihood.likelihood, model.layers, H=H, num_data=90)
    assert quad.D_quad == 2 and quad.num_samples == H ** 2
    zs, w = O.quad_points(H, [2])
    assert_allclose(w.sum(), 1.0, rtol=1e-12)
    for a, b in zip(quad.gh_x, zs):
        assert_allclose(a, b, rtol=0, atol=0)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, H ** 2, num_data=90, sample_weights=w)
    got = quad._build_likelihood(X, Y, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = quad.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-7, (k, err)
    om = OM.build(O.NP, spec, state, H ** 2, 90, sample_weights=w)
    assert_allclose(quad.E_log_p_Y(X, Y), om.E_log_p_Y(O.NP, X, Y, zs), rtol=1e-9, atol=1e-11)


def test_dgp_quad_vs_monte_carlo():
    # tests/test_dgp.py:120-174: the sampled bound agrees with quadrature within 3 standard errors (1-D inner layer,
    # z shared across the N points exactly as the reference's (S,1,D) quadrature nodes are)
    from doubly_stochastic_dgp.dgp import DGP_Quad
    rng = np.random.RandomState(22)
    N, M = 2, 2
    X, Y = rng.randn(N, 1), rng.randn(N, 1)
    specs = [kern_spec("rbf", 1, 1.0, 0.3)] * 2
    spec, state, model = make_case(X, Y, X.copy(), specs, S=300, lik_var=0.01)
    quad = DGP_Quad(X, Y, model.likelihood.likelihood, model.layers, H=100)
    q = quad.compute_log_likelihood(X, Y)
    vals = []
    for r in range(60):
        z0 = np.random.RandomState(500 + r).randn(300, 1, 1)
        vals.append(model.compute_log_likelihood(X, Y, zs=[z0, np.zeros((1, 1, 1))]))
    mean, se = np.mean(vals), np.std(vals) / np.sqrt(len(vals))
    assert abs(q - mean) < 3 * se + 1e-9, (q, mean, se)


def test_large_M_white_multiworkgroup_cholesky():
    # M = 520 (padded 1024) with white=True: the multi-workgroup blocked Cholesky/inverse path, the white-adjoint GEMMs that
    # read Lu densely (its upper blocks must be exact zeros), and a batched (D_out = 3) natural-gradient step on top
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(43)
    N, D, M, S = 24, 3, 520, 2
    X, Y = rng.randn(N, D), rng.randn(N, 3)
    Z = rng.randn(M, D) * 2.0
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("matern52", D, 1.3, 0.9)]
    spec, state, model = make_case(X, Y, Z, specs, white=True, S=S, num_data=500)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 3)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=500)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-6, (k, err)
    mu, sq = O.natgrad_step(state["l1.q_mu"], state["l1.q_sqrt"], -g["l1.q_mu"], -g["l1.q_sqrt"], 0.1)
    last = model.layers[-1]
    NatGradOptimizer(0.1).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    assert_allclose(last.q_mu.value, mu, rtol=1e-6, atol=1e-8)
    assert_allclose(last.q_sqrt.value, sq, rtol=1e-6, atol=1e-8)


def test_potrf_large_not_spd(ctx):
    from doubly_stochastic_dgp import _lib
    n = 640
    A = np.eye(n)
    A[300, 300] = -1.0
    dA = _dev(ctx, A)
    info = C.c_int(0)
    ctx.torch.cuda.current_stream().synchronize()
    rc = ctx.lib.dsdgp_potrf(ctx.handle, 1, n, _p(dA), n, n * n, C.byref(info))
    assert rc == _lib.ERR_NOT_SPD and info.value == 301


# ---------------------------------------------------------------- full BASELINE.json sizes
def test_full_size_cfg2_against_oracle():
    # configs[1] at full size: N = 1000, S = 20, M = 128, 3 layers, inner q_sqrt * 1e-5 (the benchmark workload itself)
    rng = np.random.RandomState(50)
    N, D, M, S = 1000, 8, 128, 20
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[rng.permutation(N)[:M]] + 0.05 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", D)] * 3, S=S, num_data=7372, q_sqrt_scale=1e-5, lik_var=1.0)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), np.zeros((1, 1, 1))]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=7372)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-5, (k, err)        # q_sqrt*1e-5 => cancellation-dominated variances


def test_full_size_cfg3_properties():
    # configs[2] at full size (5 layers, D = 9, M = 256, S = 20, minibatch 2000): the oracle would need minutes, so check
    # size-independent properties of the device path instead
    rng = np.random.RandomState(51)
    N, D, M, S = 2000, 9, 256, 20
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[rng.permutation(N)[:M]] + 0.05 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", D, 1.0, 1.5)] * 5, S=S, num_data=41157)
    zs = [rng.randn(S, N, D) for _ in range(4)] + [np.zeros((1, 1, 1))]
    e1 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    g1 = model.engine().grad.cpu().numpy().copy()
    e2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    g2 = model.engine().grad.cpu().numpy().copy()
    assert e1 == e2 and np.array_equal(g1, g2)                          # bitwise deterministic (no atomics)
    # (i) permuting the minibatch rows (with their noise) leaves the ELBO unchanged
    perm = rng.permutation(N)
    zp = [z[:, perm, :] for z in zs[:4]] + [zs[4]]
    e3 = model._build_likelihood(X[perm], Y[perm], zs=zp)
    assert_allclose(e3, e1, rtol=1e-11)
    # (ii) the S-sample estimator is the mean of the S single-sample estimators (same KL): rows are independent
    eng = model.engine()
    out_all = eng.elbo(X, Y, S, zs=zs, data_scale=1.0, kl_weight=0.0)
    acc = 0.0
    for s_ in (0, 7, 19):
        out_s = eng.elbo(X, Y, 1, zs=[z[s_:s_ + 1] for z in zs[:4]] + [zs[4]], data_scale=1.0, kl_weight=0.0)
        acc += out_s[1]
    _, Fm, Fv = eng.propagate(X, S, zs=zs, want=("mean", "var"))
    from doubly_stochastic_dgp.utils import BroadcastingLikelihood
    ve = model.likelihood.variational_expectations_mean(Fm[-1].cpu().numpy(), Fv[-1].cpu().numpy(), Y)
    assert_allclose(ve.sum(), out_all[1], rtol=1e-11)
    ve3 = np.mean([model.likelihood.variational_expectations_mean(Fm[-1].cpu().numpy()[s_:s_ + 1], Fv[-1].cpu().numpy()[s_:s_ + 1], Y).sum()
                   for s_ in (0, 7, 19)])
    assert_allclose(acc / 3.0, ve3, rtol=1e-10)
    # (iii) num_data enters only through the linear data scale (dgp.py:96-98)
    o1 = eng.elbo(X, Y, S, zs=zs, data_scale=1.0, kl_weight=1.0)
    o2 = eng.elbo(X, Y, S, zs=zs, data_scale=3.0, kl_weight=1.0)
    assert_allclose(o2[1], 3.0 * o1[1], rtol=1e-13)
    assert_allclose(o2[0], 3.0 * o1[1] - o1[2], rtol=1e-12)


# ---------------------------------------------------------------- reference edge cases
def test_step_up_single_point():
    # tests/test_dgp.py:176-183 TestStepUp: 1 x 1 data, kernels RBF(1) -> RBF(2): [I | 0] Linear mean, M = 1
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian
    X = np.zeros((1, 1))
    model = DGP(X, X, X, [RBF(1), RBF(2)], Gaussian())
    assert model.layers[0].mean_function.kind == "linear" and model.layers[0].num_outputs == 2
    zs = [np.array([[[0.3, -0.2]]]), np.zeros((1, 1, 1))]
    got = model.compute_log_likelihood(X, X, zs=zs)
    specs = [kern_spec("rbf", 1), kern_spec("rbf", 2)]
    lds = O.init_layers_linear(X, X, X, specs)
    sl, state = OM.state_from_layers(lds, lik_variance=1.0)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl)
    assert_allclose(got, OM.elbo(spec, state, X, X, zs, 1), rtol=1e-9)
    assert np.isfinite(model.compute_log_likelihood())          # stochastic path with device-generated z


@pytest.mark.parametrize("N,M,S", [(1, 5, 1), (2, 17, 3), (33, 31, 2), (257, 65, 5)])
def test_ragged_sizes(N, M, S):
    # single rows, row counts and inducing counts that are not multiples of the 16-row / 16-column MFMA blocks, S = 1:
    # padding rows, identity-padded Ku and the partial last row block must not leak into values or gradients
    rng = np.random.RandomState(100 + N)
    D = 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D)
    specs = [kern_spec("matern52", D, 1.1, 0.9), kern_spec("rbf", D, 0.7, 1.3, ARD=True)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=5 * N)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(2):     # random Z at M = 65: cond(Ku) ~ 1e7, so 1e-8 here (the reference's own bar is 1e-6, tests/test_dgp.py:101-106)
        assert_allclose(Fs[l], Fs_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=5 * N)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-7, (k, err)


def test_trainable_linear_mean_function():
    # [UPSTREAM] mean_functions.Linear(A, b) passed as the final mean function (dgp.py:187) with A and b free: forward with the
    # bias, gradients of A and b ([X;1]^T MB^T through the split-K launch), an Adam step, and set_trainable(False) on b
    from doubly_stochastic_dgp import settings
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import Gaussian, Linear
    rng = np.random.RandomState(61)
    N, D, M, S = 70, 3, 20, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("matern52", D, 1.2, 0.8)]
    A0, b0 = 0.3 * rng.randn(D, 2), 0.5 * rng.randn(2)
    lds = O.init_layers_linear(X, Y, Z, specs)
    for l in lds:
        l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
        l["q_sqrt"] = l["q_sqrt"] * 0.7 + 0.05 * np.tril(rng.randn(*l["q_sqrt"].shape))
    lds[-1]["mean"] = O.MeanFn("linear", A=A0, b=b0)
    lds[-1]["mean_trainable"] = True
    sl, state = OM.state_from_layers(lds, lik_variance=0.2)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl, num_classes=None)
    with settings.temp_jitter(1e-6):
        model = DGP(X, Y, Z, [product_kernel(k) for k in specs], Gaussian(variance=0.2), mean_function=Linear(A0, b0),
                    num_samples=S, num_data=300)
    for l, layer in zip(lds, model.layers):
        layer.q_mu = l["q_mu"]
        layer.q_sqrt = l["q_sqrt"]
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    assert_allclose(Fm[-1], Fm_o[-1], rtol=1e-9, atol=1e-10)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=300)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    assert set(g) == set(grads)
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-7, (k, err)
    assert np.max(np.abs(grads["l1.mean_b"])) > 0
    # one Adam step moves A and b; with b frozen it stays put
    model.layers[-1].mean_function.b.set_trainable(False)
    model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)
    model.engine().sync_to_host()
    assert_allclose(model.layers[-1].mean_function.b.value, b0, rtol=0, atol=0)
    assert np.max(np.abs(model.layers[-1].mean_function.A.value - A0)) > 1e-3


@pytest.mark.parametrize("L", [2, 4])
def test_gradients_with_stream_overlap(L):
    # large enough (n * S * Mp >= 2^20) that the side-stream overlap, the fused adjoint hand-off between inner layers and the
    # likelihood-written last-layer adjoints are all active; 2 layers (no inner layer) and 4 layers (two inner layers)
    rng = np.random.RandomState(70 + L)
    N, D, M, S = 1500, 4, 64, 12
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[rng.permutation(N)[:M]] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 1.2)] * L
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=5000, lik_var=0.3)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 1)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=5000)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - grads[k])) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-7, (k, err)


def test_empty_input_fails_loudly():
    from doubly_stochastic_dgp import _lib
    X, Y, spec, state, model, zs = _three_layer(N=20, M=10, S=2)
    with pytest.raises(_lib.DsdgpError):
        model.propagate(np.zeros((0, X.shape[1])), S=2)
    with pytest.raises((ValueError, _lib.DsdgpError)):
        model.compute_log_likelihood(np.zeros((0, X.shape[1])), np.zeros((0, Y.shape[1])))
    # the model is still usable afterwards
    assert np.isfinite(model.compute_log_likelihood(X, Y, zs=zs))


def test_stream_overlap_is_bitwise_neutral():
    # Race detector for the side-stream overlap (weight-gradient products, parameter-only algebra, RNG, finalize): every
    # reduction on the path is fixed-order, so 60 optimiser steps with and without overlap must give IDENTICAL parameters.
    import os
    rng = np.random.RandomState(77)
    N, D, M, S = 3000, 8, 128, 20
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    out = []
    for no_overlap in ("1", "0"):
        os.environ["DSDGP_NO_OVERLAP"] = no_overlap
        try:
            spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", D)] * 3, S=S, num_data=N, q_sqrt_scale=1e-3,
                                           minibatch_size=1000)
            for _ in range(60):
                model.train_step(0.01)
            elbo = model.train_step(0.01, sync=True)
            eng = model.engine()
            eng.sync_to_host()
            out.append((elbo, np.concatenate([np.ravel(l.q_mu.value) for l in model.layers]),
                        np.concatenate([np.ravel(l.feature.Z.value) for l in model.layers]),
                        np.concatenate([np.ravel(l.q_sqrt.value) for l in model.layers])))
        finally:
            os.environ["DSDGP_NO_OVERLAP"] = "0"
    assert np.isfinite(out[0][0])
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1:], out[1][1:]):
        assert np.array_equal(a, b)
