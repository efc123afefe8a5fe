"""-m gpu tests added in round 2: host/device coherence fixes (minibatch epochs, optimiser state across model re-creation,
fixed Linear maps, label validation), the layer-level sample_from_conditional (layers.py:76-119), triangular solves at
n up to 1024, and an ill-conditioning suite at the conditioning k-means inducing points reach after training.

Tolerances are stated per test; the reference's own bar is rtol = atol = 1e-7 (1 layer) / 1e-6 (2 layers)
(/root/reference/tests/test_dgp.py:101-106).
"""
import ctypes as C
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from doubly_stochastic_dgp.engine import Context
    return Context.get()


def _dev(ctx, a):
    return ctx.to_device(np.ascontiguousarray(a, dtype=np.float64))


def _p(t):
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------- host / device coherence
def test_minibatch_rows_follow_the_permutation_across_epochs():
    """N % B != 0: minibatches straddle epoch boundaries; every gathered batch must be X[perm-stream indices] — sampling
    without replacement inside each epoch, rows of X and Y paired (dgp.py:50-52)."""
    from doubly_stochastic_dgp.dgp import Minibatch
    rng = np.random.RandomState(0)
    N, B = 25, 10
    X = rng.randn(N, 3)
    Y = np.arange(N, dtype=np.float64)[:, None]
    spec, state, model = make_case(X, Y, X[:8].copy(), [kern_spec("rbf", 3)], S=1, minibatch_size=B)
    ref = Minibatch(N, B, seed=0)              # the model's own stream (seed 0, dgp.py:51-52)
    seen = []
    for _ in range(23):
        want = ref.next_indices()
        Xb, Yb = model.next_minibatch()
        assert np.array_equal(Yb.cpu().numpy()[:, 0], want.astype(np.float64))
        assert np.array_equal(Xb.cpu().numpy(), X[want])
        seen.append(want)
    seen = np.concatenate(seen)
    for e in range(len(seen) // N):
        assert sorted(seen[e * N:(e + 1) * N]) == list(range(N))


def _small_model(seed=0, N=60, D=3, M=16, S=2):
    rng = np.random.RandomState(seed)
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.0, 1.0)] * 2
    spec, state, model = make_case(X, Y, Z, specs, S=S, seed=seed)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    return X, Y, spec, state, model, zs


def test_adam_state_survives_a_larger_predict():
    """predict_* with more rows / samples than the training shape re-creates the device model; the Adam moments, the step
    count and theta must carry over (TF keeps its Adam slots across predictions)."""
    X, Y, _, _, a, zs = _small_model()
    _, _, _, _, b, _ = _small_model()
    Xnew = np.random.RandomState(5).randn(4 * X.shape[0], X.shape[1])
    for _ in range(3):
        a.train_step(0.01, X=X, Y=Y, zs=zs)
        b.train_step(0.01, X=X, Y=Y, zs=zs)
    m, v = a.predict_f(Xnew, 9)                 # n = 240 > 60, S = 9 > 2 -> _ensure() grows the workspace
    assert m.shape == (9, Xnew.shape[0], 1) and np.all(np.isfinite(m)) and np.all(v > 0)
    assert a.engine().adam_t == 3
    for _ in range(3):
        a.train_step(0.01, X=X, Y=Y, zs=zs)
        b.train_step(0.01, X=X, Y=Y, zs=zs)
    # not bitwise: the larger workspace changes split-K counts / the alg_g choice, i.e. summation orders (~1e-15 relative)
    for la, lb in zip(a.layers, b.layers):
        assert_allclose(la.q_mu.value, lb.q_mu.value, rtol=1e-9, atol=1e-12)
        assert_allclose(la.q_sqrt.value, lb.q_sqrt.value, rtol=1e-9, atol=1e-12)
        assert_allclose(la.feature.Z.value, lb.feature.Z.value, rtol=1e-9, atol=1e-12)
    assert_allclose(a.likelihood.likelihood.variance.value, b.likelihood.likelihood.variance.value, rtol=1e-9)
    # a wiped optimiser state would restart Adam's bias correction: the first step after it moves every entry by ~lr
    assert a.engine().adam_t == 6


def test_fixed_linear_map_assignment_reaches_the_device():
    """init_layers_linear's PCA map is a fixed device constant (layer_initializations.py:41-42); assigning a new A, or a
    non-zero bias, must still be seen by the next evaluation."""
    rng = np.random.RandomState(2)
    N, S = 40, 2
    X, Y = rng.randn(N, 5), rng.randn(N, 1)
    Z = X[:12].copy()
    specs = [kern_spec("rbf", 5), kern_spec("rbf", 2)]
    spec, state, model = make_case(X, Y, Z, specs, S=S)
    zs = [rng.randn(S, N, 2), rng.randn(S, N, 1)]
    assert model.layers[0].mean_function.kind == "linear"
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, state, X, Y, zs, S), rtol=1e-9)
    W2 = rng.randn(5, 2)
    model.layers[0].mean_function.A = W2
    spec["layers"][0]["mean_A"] = W2
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, state, X, Y, zs, S), rtol=1e-9)
    bias = np.array([0.3, -0.2])
    model.layers[0].mean_function.b = bias       # a biased map moves into theta (structure change)
    spec["layers"][0]["mean_trainable"] = True
    state["l0.mean_A"], state["l0.mean_b"] = W2.copy(), bias.copy()
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, state, X, Y, zs, S), rtol=1e-9)


def test_multiclass_labels_checked_before_launch():
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import MultiClass, RBF
    rng = np.random.RandomState(0)
    X = rng.randn(20, 2)
    with pytest.raises(ValueError):
        DGP(X, np.full((20, 1), 3.0), X[:5], [RBF(2)], MultiClass(3), num_outputs=3)
    model = DGP(X, rng.randint(0, 3, (20, 1)).astype(float), X[:5], [RBF(2)], MultiClass(3), num_outputs=3)
    assert np.isfinite(model.compute_log_likelihood())
    with pytest.raises(ValueError):
        model.compute_log_likelihood(X, np.eye(3)[rng.randint(0, 3, 20)])      # one-hot targets passed by mistake


# ---------------------------------------------------------------- layer-level surface (layers.py:52-119)
@pytest.mark.parametrize("white", [False, True])
def test_layer_sample_from_conditional_diag(white):
    """Layer.sample_from_conditional(X, z, full_cov=False) at LAYER level: conditional_SND + reparameterize as separate
    device calls, against the oracle's layer."""
    rng = np.random.RandomState(4)
    N, D, M, S, Dout = 33, 3, 12, 4, 2
    X, Y = rng.randn(N, D), rng.randn(N, Dout)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec("matern52", D, 1.3, 0.7)], white=white, S=S)
    om = OM.build(O.NP, spec, state, S)
    XS = rng.randn(S, N, D)
    z = rng.randn(S, N, Dout)
    f, m, v = model.layers[0].sample_from_conditional(XS, z=z, full_cov=False)
    fo, mo, vo = om.layers[0].sample_from_conditional(O.NP, XS, z=z, full_cov=False)
    assert f.shape == (S, N, Dout)
    assert_allclose(m, mo, rtol=1e-9, atol=1e-10)
    assert_allclose(v, vo, rtol=1e-9, atol=1e-10)
    assert_allclose(f, fo, rtol=1e-9, atol=1e-10)
    # z = None draws from the device generator: same moments, different sample
    f2, m2, v2 = model.layers[0].sample_from_conditional(XS, z=None)
    assert_allclose(m2, mo, rtol=1e-9, atol=1e-10)
    zz = (f2 - m2) / np.sqrt(v2 + spec["jitter"])
    assert abs(zz.mean()) < 0.2 and 0.7 < zz.std() < 1.3


# ---------------------------------------------------------------- triangular solves at the sizes cfg 4 / 5 use
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("n,nrhs", [(256, 500), (512, 300), (1024, 200),
                                    (256, 2048), (600, 2056), (1024, 4096), (1366, 2304)])      # from 2048 columns: the left-looking form
def test_trsm_large(ctx, trans, n, nrhs):
    """dsdgp_trsm is a blocked substitution (16 x 16 diagonal inverses, 128-row panels, MFMA GEMM updates — right-looking, or
    left-looking through the LDS-tiled kernel for one large system with thousands of columns); forward error scales with cond(L).
    Well-conditioned L: rtol 1e-10; the residual bar |L x - b| <= 1e-13 n |L| |x| is what a backward-stable solve would meet."""
    import scipy.linalg as sla
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n)
    L = np.tril(rng.randn(n, n)) / np.sqrt(n) + 2.0 * np.eye(n)
    B = rng.randn(n, nrhs)
    dL, dB = _dev(ctx, L), _dev(ctx, B)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_trsm(ctx.handle, trans, n, nrhs, _p(dL), n, _p(dB), nrhs))
    ctx.sync()
    Xs = dB.cpu().numpy()
    ref = sla.solve_triangular(L, B, lower=True, trans=trans)
    assert_allclose(Xs, ref, rtol=1e-10, atol=1e-11)
    Lop = L.T if trans else L
    assert np.linalg.norm(Lop @ Xs - B) <= 1e-13 * n * np.linalg.norm(L) * np.linalg.norm(Xs)


# ---------------------------------------------------------------- ill-conditioning
def _near_duplicate_case(white, sep, N=48, D=2, M=24, S=3, jitter=1e-6, seed=7):
    """Inducing points in near-duplicate pairs `sep` apart (what k-means centres of clustered data drift to during
    training): cond(Ku) ~ 2 / (jitter + sep^2)."""
    rng = np.random.RandomState(seed)
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    base = rng.randn(M // 2, D)
    Z = np.concatenate([base, base + sep * rng.randn(M // 2, D)])
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("rbf", D, 1.0, 1.0)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, jitter=jitter, seed=seed)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    K = O.Kern("rbf", D).K(O.NP, Z) + jitter * np.eye(M)
    return X, Y, spec, state, model, zs, np.linalg.cond(K)


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("sep,tol", [(1e-2, 1e-7), (1e-4, 1e-6), (0.0, 1e-6)])
def test_ill_conditioned_inducing_points(white, sep, tol):
    """Explicit Lu^-1 / Ku^-1 products vs the oracle's backward-stable trsm form at cond(Ku) up to ~1e7 (jitter 1e-6 bounds
    it: duplicated points give 2 / jitter).  Bar = the reference's own test tolerance (1e-7 one layer, 1e-6 two layers,
    tests/test_dgp.py:101-106); the forward error of BOTH forms grows like cond * eps here."""
    from doubly_stochastic_dgp import settings
    X, Y, spec, state, model, zs, cond = _near_duplicate_case(white, sep)
    assert cond > (1e3 if sep >= 1e-2 else 1e6)
    with settings.temp_jitter(1e-6):
        Fs_o, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, 3)
        Fs, Fm, Fv = model.propagate(X, S=3, zs=zs)
        for l in range(2):
            assert_allclose(Fm[l], Fm_o[l], rtol=tol, atol=tol)
            assert_allclose(Fv[l], Fv_o[l], rtol=tol, atol=tol)
        ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, 3)
        got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
        assert_allclose(got, ref, rtol=tol)
        assert_allclose([layer.KL() for layer in model.layers], [l.KL(O.NP) for l in OM.build(O.NP, spec, state, 3).layers],
                        rtol=tol)
        g = model.engine().gradient_dict()
        # gradients of an ill-conditioned Ku are themselves ill-conditioned (both sides lose cond * eps): 1e3 x the value bar
        for k in gref:
            scale = np.max(np.abs(gref[k])) + 1e-12
            assert np.max(np.abs(-gref[k] - g[k])) <= 1e3 * tol * scale, (k, cond)


def test_ill_conditioned_smaller_jitter_1e10():
    """jitter 1e-9 with 1e-5-separated pairs: cond(Ku) ~ 1e9..1e10.  Values stay within 1e-4 of the oracle (cond * eps ~ 1e-6
    per solve, two chained layers); a non-finite or wildly different result would flag a breakdown of the explicit inverse."""
    from doubly_stochastic_dgp import settings
    with settings.temp_jitter(1e-9):
        X, Y, spec, state, model, zs, cond = _near_duplicate_case(False, 1e-5, jitter=1e-9)
        assert cond > 1e9
        _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, 3)
        _, Fm, Fv = model.propagate(X, S=3, zs=zs)
        assert np.all(np.isfinite(Fm[-1])) and np.all(np.isfinite(Fv[-1]))
        assert_allclose(Fm[-1], Fm_o[-1], rtol=1e-4, atol=1e-4)
        assert_allclose(Fv[-1], Fv_o[-1], rtol=1e-4, atol=1e-4)
        assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, state, X, Y, zs, 3), rtol=1e-4)


def test_negative_variance_gives_nan_like_the_reference():
    """utils.py:41 takes (var + jitter) ** 0.5 with NO clamp: a variance below -jitter yields NaN samples, not a silently
    clamped value.  Forced here with q_sqrt = 0 and a negative jitter (var = Kdiag - k^T Ku^-1 k ~ 0 at the inducing points)."""
    from doubly_stochastic_dgp import settings
    rng = np.random.RandomState(1)
    N, D, M, S = 20, 2, 10, 2
    X = rng.randn(N, D)
    X[:M] = X[:M]                                   # the first M rows ARE the inducing points: var there ~ 0
    spec, state, model = make_case(X, rng.randn(N, 1), X[:M].copy(), [kern_spec("rbf", D)], S=S, randomize=False)
    model.layers[0].q_sqrt = np.zeros((1, M, M))
    state["l0.q_sqrt"] = np.zeros((1, M, M))
    z = [rng.randn(S, N, 1)]
    mean, var = model.layers[0].conditional_ND(X)
    assert np.all(np.abs(var[:M]) < 1e-4)
    with settings.temp_jitter(1e-6):
        spec["jitter"] = 1e-6
        Fs, _, Fv = model.propagate(X, S=S, zs=z)
        Fs_o, _, Fv_o = OM.propagate(spec, state, X, z, S)
        assert_allclose(Fv[0], Fv_o[0], rtol=1e-6, atol=1e-9)
    from doubly_stochastic_dgp.utils import reparameterize
    v = np.array(Fv[0])
    v[0, :M] = -1e-3                                # below -jitter: the reference's sqrt returns NaN there
    out = reparameterize(np.zeros_like(v), v, z[0])
    assert np.all(np.isnan(out[0, :M])) and np.all(np.isfinite(out[0, M:]))


# ---------------------------------------------------------------- Csave backward chain / alg_g on small shapes
@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("M,D", [(20, 3), (40, 4), (128, 8), (200, 5)])
def test_csave_chain_and_alg_g_gradients(monkeypatch, white, M, D):
    """The production heuristics keep the c_d-saving backward chain and the algebraic dl/dKu assembly for large launches; force
    both onto small, oracle-checkable shapes (every Mp instance 32..256, both ownership tables, white and non-white)."""
    monkeypatch.setenv("DSDGP_FORCE", "save_c=2,cs_min_blocks=0,cs_min_dout=1,alg_g=1")      # read when the model is created
    rng = np.random.RandomState(M + D)
    N, S = 70, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D) * 1.3
    specs = [kern_spec("rbf", D, 1.1, 1.4), kern_spec("matern52", D, 0.9, 1.2), kern_spec("rbf", D, 1.0, 1.0)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=500)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 2)]
    ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=500)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    g = model.engine().gradient_dict()
    for k in gref:
        assert np.max(np.abs(-gref[k] - g[k])) <= 1e-7 * (np.max(np.abs(gref[k])) + 1e-12), k


@pytest.mark.parametrize("n,n2,D", [(130, 2050, 30), (8, 512, 1), (257, 1026, 9)])
def test_gram_paired_store_path(ctx, n, n2, D):
    """dsdgp_gram with an even leading dimension takes the 16-byte paired-store path; several passes over D (D > 8)."""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n)
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = 0.5 + rng.rand(D)
    k = O.Kern("matern52", D, variance=1.3, lengthscales=ls, ARD=True)
    spec = _lib.KernelSpec(kind=1, input_dim=D, ard=1, has_white=0, variance=1.3, white_variance=0.0,
                           lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    dX, dX2 = _dev(ctx, X), _dev(ctx, X2)
    out = ctx.empty(n, n2)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, _p(dX2), n2, 0.0, _p(out), n2))
    ctx.sync()
    assert_allclose(out.cpu().numpy(), k.K(O.NP, X, X2), rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(512, 512, 512), (600, 520, 300), (1024, 513, 77), (700, 1030, 1024), (2048, 2100, 96), (2050, 2048, 130)])
def test_gemm_large_tile_kernel(ctx, tA, tB, m, n, k):
    """dsdgp_gemm around the kernel switch: up to 1024 the 64 x 64 kernels, with both output dimensions >= 2048 the 128 x 128
    double-buffered kernel (k_gemm_big); ragged edges, every transpose combination, alpha / beta."""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(m + n + k + tA + 2 * tB)
    A = rng.randn(*((k, m) if tA else (m, k)))
    B = rng.randn(*((n, k) if tB else (k, n)))
    Cc = rng.randn(m, n)
    dA, dB, dC = _dev(ctx, A), _dev(ctx, B), _dev(ctx, Cc)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gemm(ctx.handle, tA, tB, m, n, k, 0.75, _p(dA), A.shape[1], _p(dB), B.shape[1], -1.25, _p(dC), n))
    ctx.sync()
    ref = 0.75 * (A.T if tA else A) @ (B.T if tB else B) - 1.25 * Cc
    assert_allclose(dC.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * k)


# ---------------------------------------------------------------- full per-GPU shards of the 8-GPU configs (BASELINE configs[3], [4])
def _shard_properties(model, X, Y, zs, S, n_layers, check_grad_perm=True):
    """Size-independent properties of one full shard (the oracle would need minutes at these sizes):
    bitwise determinism, row-permutation invariance of value AND gradient, estimator linearity over samples, linear data scale."""
    rng = np.random.RandomState(0)
    N = X.shape[0]
    e1 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    g1 = model.engine().grad.cpu().numpy().copy()
    e2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    g2 = model.engine().grad.cpu().numpy().copy()
    assert np.isfinite(e1) and np.all(np.isfinite(g1))
    assert e1 == e2 and np.array_equal(g1, g2)                          # fixed-order reductions only: bitwise deterministic
    perm = rng.permutation(N)
    zp = [z[:, perm, :] if z.shape[1] == N else z for z in zs]
    e3 = model._build_likelihood(X[perm], Y[perm], zs=zp, with_grad=check_grad_perm)
    assert_allclose(e3, e1, rtol=1e-10)
    if check_grad_perm:                                                 # same rows in another order: the same gradient up to summation order
        g3 = model.engine().grad.cpu().numpy()
        assert np.max(np.abs(g3 - g1)) <= 1e-8 * np.max(np.abs(g1))
    eng = model.engine()
    o1 = eng.elbo(X, Y, S, zs=zs, data_scale=1.0, kl_weight=1.0)
    o3 = eng.elbo(X, Y, S, zs=zs, data_scale=3.0, kl_weight=0.5)
    assert_allclose(o3[1], 3.0 * o1[1], rtol=1e-13)                     # dgp.py:96-98: linear data scale
    assert_allclose(o3[2], 0.5 * o1[2], rtol=1e-13)
    assert_allclose(o3[0], o3[1] - o3[2], rtol=1e-12)
    # mean over S of the single-sample data terms == the S-sample data term (rows are independent through all layers)
    acc = [eng.elbo(X, Y, 1, zs=[z[s_:s_ + 1] if z.shape[0] == S else z for z in zs], data_scale=1.0, kl_weight=0.0)[1]
           for s_ in range(S)]
    assert_allclose(np.mean(acc), o1[1], rtol=1e-10)
    return e1, g1


def test_full_shard_cfg4_properties():
    """configs[3]: MNIST-shaped 784 -> 30 -> 30 -> 10, M = 512, MultiClass(10), S = 10, minibatch 4096 over 8 GPUs = 512 rows
    per GPU (the shard tools/bench_configs.py times: wide-input first layer at 512 rows, 8-wave M = 512 chains at 5120 rows)."""
    rng = np.random.RandomState(60)
    Ndata, N, S, M, K = 1200, 512, 10, 512, 10
    Xall = rng.uniform(size=(Ndata, 784)) * (rng.uniform(size=(Ndata, 784)) < 0.19)
    Yall = rng.randint(0, K, size=(Ndata, 1)).astype(np.float64)
    Z = Xall[rng.permutation(Ndata)[:M]] + 0.01 * rng.randn(M, 784)
    specs = [kern_spec("rbf", 784, 2.0, 2.0), kern_spec("rbf", 30, 2.0, 2.0), kern_spec("rbf", 30, 2.0, 2.0)]
    spec, state, model = make_case(Xall, Yall, Z, specs, S=S, num_data=60000, num_classes=K, randomize=False)
    X, Y = Xall[:N], Yall[:N]
    zs = [rng.randn(S, N, 30), rng.randn(S, N, 30), rng.randn(S, N, K)]
    _shard_properties(model, X, Y, zs, S, 3)
    # the gradient is an ascent direction of the bound on this shard: one small sign-step of Adam (first step = lr * sign(g))
    # raises it, by about lr * |g|_1
    e0 = model.compute_log_likelihood(X, Y, zs=zs)
    g1n = float(np.sum(np.abs(model.engine().grad.cpu().numpy())))
    model.train_step(1e-6, X=X, Y=Y, zs=zs)
    e1 = model.compute_log_likelihood(X, Y, zs=zs)
    assert e1 > e0 and (e1 - e0) < 2.0e-6 * g1n


def test_full_shard_cfg5_properties_and_natgrad():
    """configs[4]: 8 -> 8 -> 8 -> 1, M = 1024, S = 50, minibatch 1000 over 8 GPUs = 125 rows per GPU (16-wave M = 1024 chains at
    6250 rows, multi-workgroup Cholesky, large-tile GEMMs) + a natural-gradient step on the last layer."""
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(61)
    N, D, M, S = 125, 8, 1024, 50
    Xall, Yall = rng.randn(2000, D), rng.randn(2000, 1)
    Z = Xall[rng.permutation(2000)[:M]] + 0.05 * rng.randn(M, D)
    spec, state, model = make_case(Xall, Yall, Z, [kern_spec("rbf", D)] * 3, S=S, num_data=7372, q_sqrt_scale=1e-5, lik_var=1.0)
    X, Y = Xall[:N], Yall[:N]
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), np.zeros((1, 1, 1))]
    e1, g1 = _shard_properties(model, X, Y, zs, S, 3)
    # natural-gradient step with gamma = 1 on the (Gaussian-likelihood) last layer maximises the bound over its q(u):
    # no other q(u) for that layer gives a higher value (tests/test_collapsed.py:57-104 is the 1-layer instance of this)
    last = model.layers[-1]
    NatGradOptimizer(1.0).minimize(model, var_list=[[last.q_mu, last.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    e_opt = model.compute_log_likelihood(X, Y, zs=zs)
    assert e_opt > e1
    q_mu, q_sqrt = last.q_mu.read_value(), last.q_sqrt.read_value()
    for _ in range(3):
        last.q_mu = q_mu + 1e-3 * rng.randn(*q_mu.shape)
        assert model.compute_log_likelihood(X, Y, zs=zs) < e_opt
    last.q_mu = q_mu
    last.q_sqrt = q_sqrt * (1.0 + 1e-3)
    assert model.compute_log_likelihood(X, Y, zs=zs) < e_opt


# ---------------------------------------------------------------- inducing counts that are not powers of two
@pytest.mark.parametrize("M,Mp", [(100, 112), (200, 224), (300, 320), (70, 80), (140, 160), (600, 640)])
def test_non_power_of_two_inducing_counts(M, Mp):
    """M = 100 is the reference demo size (demos/run_regression.py:57); the padded size the device works on is the next
    multiple of 16 / 32 / 64 / 128, not the next power of two, and values + gradients match the oracle as for any other M."""
    rng = np.random.RandomState(M)
    N, D, S = 50, 4, 2
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D) * 1.4
    specs = [kern_spec("rbf", D, 1.2, 1.3), kern_spec("matern52", D, 0.9, 1.1)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=777)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=777)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    g = model.engine().gradient_dict()
    for k in gref:
        assert np.max(np.abs(-gref[k] - g[k])) <= 1e-7 * (np.max(np.abs(gref[k])) + 1e-12), k
    # the workspace is sized by the padded count: (D_out x Mp x Mp) blocks, so Mp shows in its size
    import ctypes
    from doubly_stochastic_dgp import _lib
    eng = model.engine()
    nb = ctypes.c_int64()
    _lib.check(eng.lib.dsdgp_model_workspace_bytes(ctypes.byref(eng.desc), 64, 1, ctypes.byref(nb)))
    d_pow2 = 1 << int(np.ceil(np.log2(M)))
    assert Mp < d_pow2 or M == d_pow2


# ---------------------------------------------------------------- Bernoulli likelihood (reference tests/test_dgp.py:48-54)
@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("white", [True, False])
def test_bernoulli_elbo_gradients_and_predictions(L, white):
    """Shapes of the reference's test_bernoulli (Y in {-1, 1}, L = 1 and 2, white=True there; white=False added): ELBO and every
    gradient block vs the oracle's torch autograd, E_log_p_Y, predict_density, predict_y."""
    from tests.test_gpu_parity import _grad_check
    rng = np.random.RandomState(48 + L)
    N, D, M, S, DY = 50, 2, 19, 3, 2
    X = rng.uniform(size=(N, D))
    Y = rng.choice([-1.0, 1.0], N * DY).reshape(N, DY)
    Z = X[:M].copy()
    specs = [kern_spec("matern52", D, 1.0, 0.5, white_variance=0.01) for _ in range(L)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=200, bernoulli=True)
    widths = [D] * (L - 1) + [DY]
    zs = [rng.randn(S, N, w) for w in widths]
    _grad_check(X, Y, spec, state, model, zs, S, num_data=200)
    om = OM.build(O.NP, spec, state, S, 200)
    assert_allclose(model.E_log_p_Y(X, Y, zs=zs), om.E_log_p_Y(O.NP, X, Y, zs), rtol=1e-10, atol=1e-12)
    _, Fm, Fv = om.propagate(O.NP, X, zs, S=S)
    m, v = model._build_predict(X, S=S, zs=zs)
    assert_allclose(model.likelihood.predict_density_logmeanexp(m, v, Y), om.predict_density(O.NP, X, Y, zs, S), rtol=1e-10,
                    atol=1e-12)
    pm, pv = model.likelihood.predict_mean_and_var(m, v)
    rm, rv = om.likelihood.predict_mean_and_var(O.NP, Fm[-1], Fv[-1])
    assert_allclose(pm, rm, rtol=1e-12, atol=1e-14)
    assert_allclose(pv, rv, rtol=1e-10, atol=1e-14)
    assert np.all(pm > 1e-3 - 1e-12) and np.all(pm < 1 - 1e-3 + 1e-12)       # probit's 1e-3 floor / ceiling


def test_bernoulli_var_exp_primitive_and_edge_targets():
    """dsdgp_bernoulli_var_exp vs the oracle: targets 1 select p, every other value (0, -1, 2.5) 1 - p; quadrature weights
    replace the mean over S; a negative variance gives NaN as upstream's sqrt does."""
    from doubly_stochastic_dgp.gpflow_compat import Bernoulli
    from doubly_stochastic_dgp.utils import BroadcastingLikelihood
    rng = np.random.RandomState(3)
    S, N, D = 4, 37, 3
    mu, var = 2.0 * rng.randn(S, N, D), rng.uniform(1e-6, 4.0, size=(S, N, D))
    Y = rng.choice([1.0, 0.0, -1.0, 2.5], N * D).reshape(N, D)
    lik, ol = BroadcastingLikelihood(Bernoulli()), O.Bernoulli()
    ve = ol.variational_expectations(O.NP, mu, var, Y)
    assert_allclose(lik.variational_expectations_mean(mu, var, Y), ve.mean(0), rtol=1e-12, atol=1e-14)
    w = rng.uniform(size=S)
    assert_allclose(lik.variational_expectations_mean(mu, var, Y, weights=w), (ve * w[:, None, None]).sum(0), rtol=1e-12, atol=1e-14)
    from scipy.special import logsumexp
    assert_allclose(lik.predict_density_logmeanexp(mu, var, Y), logsumexp(ol.predict_density(O.NP, mu, var, Y), axis=0) - np.log(S),
                    rtol=1e-12, atol=1e-14)
    bad = var.copy()
    bad[1, 5, 2] = -0.1
    out = lik.variational_expectations_mean(mu, bad, Y)
    assert np.isnan(out[5, 2]) and np.isfinite(np.delete(out.ravel(), 5 * D + 2)).all()


def test_bernoulli_training_decreases_loss_and_classifies():
    """Two-layer Bernoulli DGP on a separable toy problem: 200 Adam steps raise the ELBO and the predictive mean separates the
    classes (an end-to-end run of the path the reference exercises only through compare_to_single_layer)."""
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import RBF, Bernoulli
    rng = np.random.RandomState(0)
    N = 200
    X = rng.uniform(-2, 2, size=(N, 2))
    Y = np.where(X[:, :1] * X[:, 1:] > 0, 1.0, -1.0)
    model = DGP(X, Y, X[:20].copy(), [RBF(2, lengthscales=1.0), RBF(2, lengthscales=1.0)], Bernoulli(), num_samples=5)
    e0 = np.mean([model.compute_log_likelihood() for _ in range(5)])
    for _ in range(300):
        model.train_step(0.02)
    e1 = np.mean([model.compute_log_likelihood() for _ in range(5)])
    assert e1 > e0 + 10.0
    p, _ = model.predict_y(X, 20)
    acc = np.mean((p.mean(0) > 0.5) == (Y == 1.0))
    assert acc > 0.9


def test_backward_d_split_forced_on_small_shapes(monkeypatch):
    """The d-split of the backward chain (several workgroups per row block share the per-output loop and hand their partial
    tiles over through global memory) is used from Mp = 512 by default; DSDGP_FORCE=bwd_split=2 forces it onto small shapes: dense
    S_d form (M = 32, 100), Csave form (save_c=2), white and non-white, D_out not divisible by the split."""
    from tests.test_gpu_parity import _grad_check
    rng = np.random.RandomState(77)
    for M, white, save_c in ((32, False, "1"), (100, True, "1"), (64, False, "2")):
        monkeypatch.setenv("DSDGP_FORCE", f"bwd_split=2,save_c={save_c},cs_min_blocks=0,cs_min_dout=1")
        N, D, S, DY = 70, 3, 3, 5
        X, Y = rng.randn(N, D), rng.randn(N, DY)
        Z = X[rng.permutation(N)[:min(M, N)]] if M <= N else np.vstack([X, rng.randn(M - N, D)])
        Z = Z + 0.01 * rng.randn(*Z.shape)
        specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("rbf", D, 0.8, 1.2)]
        spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=300)
        zs = [rng.randn(S, N, D), rng.randn(S, N, DY)]
        _grad_check(X, Y, spec, state, model, zs, S, num_data=300)
        # a second evaluation on the same model: the arrival counters were reset by the last workgroups
        e1 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
        e2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
        assert e1 == e2


def test_factorisation_kept_after_natgrad_step_is_bitwise_neutral():
    """dsdgp_model_track_theta: an evaluation that follows a natural-gradient step alone keeps Lu / Lu^-1 / Ku^-1 (Z and the kernel
    hyper-parameters did not move).  Same alternating Adam / NatGrad schedule on two identical models, one of which reports a
    (fictitious) write to theta before every evaluation and therefore refactorises every time: identical ELBOs and parameters."""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(8)
    N, D, M, S = 120, 3, 40, 4
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("rbf", D, 0.8, 1.2)]
    _, _, a = make_case(X, Y, Z, specs, S=S, num_data=500)
    _, _, b = make_case(X, Y, Z, specs, S=S, num_data=500)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    vals = []
    for model, refactor in ((a, False), (b, True)):
        eng = model.engine()
        out = []
        for it in range(3):
            if refactor:
                eng._upload_if_needed()
                _lib.check(eng.lib.dsdgp_model_theta_changed(eng.model))
            model.train_step(0.01, X=X, Y=Y, zs=zs)                       # Adam: invalidates everything
            if refactor:
                _lib.check(eng.lib.dsdgp_model_theta_changed(eng.model))
            model._build_likelihood(X, Y, zs=zs, with_grad=True)          # gradient for the natural-gradient step
            eng.natgrad_step(len(model.layers) - 1, 0.1)
            if refactor:
                _lib.check(eng.lib.dsdgp_model_theta_changed(eng.model))
            out.append(model._build_likelihood(X, Y, zs=zs, with_grad=True))   # follows the natgrad step alone: Ku side reused
        vals.append((out, eng.theta.cpu().numpy().copy(), eng.grad.cpu().numpy().copy()))
    assert vals[0][0] == vals[1][0]
    assert np.array_equal(vals[0][1], vals[1][1])
    assert np.array_equal(vals[0][2], vals[1][2])


def test_natgrad_with_pruned_reverse_pass_matches_full_gradient():
    """NatGradOptimizer.minimize evaluates the gradient w.r.t. its var_list only (dsdgp_model_set_grad_first_layer: the reverse
    pass stops below the lowest layer in it, as tf.gradients does).  The natural-gradient step from the pruned pass must be the
    one from the full gradient, bit for bit, and a following full-gradient evaluation must be unaffected."""
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(18)
    N, D, M, S = 90, 3, 30, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)]
    _, _, a = make_case(X, Y, Z, specs, S=S, num_data=400)
    _, _, b = make_case(X, Y, Z, specs, S=S, num_data=400)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 2)]
    last = len(a.layers) - 1
    # a: the optimiser's own (pruned) path ; b: full gradient, then the same step
    NatGradOptimizer(0.05).minimize(a, var_list=[[a.layers[-1].q_mu, a.layers[-1].q_sqrt]], maxiter=2, X=X, Y=Y, zs=zs)
    for _ in range(2):
        b._build_likelihood(X, Y, zs=zs, with_grad=True)
        b.engine().natgrad_step(last, 0.05)
    assert np.array_equal(a.engine().theta.cpu().numpy(), b.engine().theta.cpu().numpy())
    ea = a._build_likelihood(X, Y, zs=zs, with_grad=True)
    eb = b._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert ea == eb
    assert np.array_equal(a.engine().grad.cpu().numpy(), b.engine().grad.cpu().numpy())
    # two layers in the var_list: pruned below the lower one
    NatGradOptimizer(0.02).minimize(a, var_list=[[l.q_mu, l.q_sqrt] for l in a.layers[1:]], X=X, Y=Y, zs=zs)
    b._build_likelihood(X, Y, zs=zs, with_grad=True)
    for l in (1, 2):
        b.engine().natgrad_step(l, 0.02)
    assert np.array_equal(a.engine().theta.cpu().numpy(), b.engine().theta.cpu().numpy())


def test_adam_step_refused_after_pruned_gradient():
    """A pruned reverse pass leaves the lower layers' gradient entries stale: dsdgp_model_adam_step refuses to use them."""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(2)
    N, D, M, S = 40, 2, 16, 2
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    specs = [kern_spec("rbf", D, 1.0, 1.0), kern_spec("rbf", D, 1.0, 1.0)]
    _, _, model = make_case(X, Y, X[:M].copy(), specs, S=S)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=1)
    with pytest.raises(_lib.DsdgpError):
        model.engine().adam_step(0.01)
    model._build_likelihood(X, Y, zs=zs, with_grad=True)          # a full gradient again
    model.engine().adam_step(0.01)
