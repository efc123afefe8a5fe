"""Round 5: the two launch forms of the materialised Gram build at the sizes that select them (the barrier-free streaming form only
runs from 8192 tiles of 64 x 32), the launch counter, and the open items of the round-4 advice: a (q_mu, q_sqrt)-only reverse pass as
the FIRST evaluation of a model and after a change of the minibatch shape, a layer conditional on more rows than the model's scratch
holds (GEMM-formulated layers go through in chunks), and the largest padded inducing count the library accepts (Mp = 2048)."""
import ctypes as C

import numpy as np
import pytest
from numpy.testing import assert_allclose

from doubly_stochastic_dgp import _lib
from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from doubly_stochastic_dgp.engine import Context
    return Context.get()


def _p(t):
    return C.c_void_p(t.data_ptr())


def _spec(kind, D, ls, var, ard):
    return _lib.KernelSpec(kind={"rbf": 0, "matern52": 1}[kind], input_dim=D, ard=int(ard), has_white=0, variance=var, white_variance=0.0,
                           lengthscales=ls.ctypes.data_as(_lib.c_double_p))


# ---------------------------------------------------------------- Gram build: streaming form (layers.py:184 as a materialised matrix)
@pytest.mark.parametrize("kind,n,n2,D,ld_extra", [("rbf", 520, 33000, 8, 0),         # ragged rows (520 = 8 x 64 + 8) and columns
                                                    ("rbf", 513, 32801, 8, 0),         # odd leading dimension: scalar stores everywhere
                                                    ("matern52", 1024, 16400, 5, 2),   # D < 8 (zero-padded k-steps), wider buffer
                                                    ("rbf", 640, 26240, 7, 0)])
def test_gram_streaming_form_against_the_oracle(ctx, kind, n, n2, D, ld_extra):
    """>= 8192 tiles and D <= 8: k_gram_mfma3 (wave-private staging, hand-counted s_waitcnt behind the stores).  Every entry against
    the numpy oracle; columns beyond n2 of a wider buffer untouched."""
    rng = np.random.RandomState(n + n2 + D)
    assert ((n + 63) // 64) * ((n2 + 31) // 32) >= 8192
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = 0.8 + rng.rand(D)
    k = O.Kern(kind, D, variance=1.3, lengthscales=ls, ARD=True)
    ld = n2 + ld_extra
    out = ctx.to_device(np.full((n, ld), 7.0))
    dX, dX2 = ctx.to_device(X), ctx.to_device(X2)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(_spec(kind, D, ls, 1.3, True)), _p(dX), n, _p(dX2), n2, 0.0, _p(out), ld))
    ctx.sync()
    got = out.cpu().numpy()
    assert_allclose(got[:, :n2], k.K(O.NP, X, X2), rtol=1e-11, atol=1e-13)
    assert np.all(got[:, n2:] == 7.0)


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_gram_streaming_form_symmetric_call(ctx, kind):
    """K(X, X) + jitter I at n = 4100 (65 x 129 tiles): exactly symmetric, one value on the diagonal, equal to the oracle — the
    diagonal tiles take the slow path of the streaming form, the others the fast one."""
    rng = np.random.RandomState(5)
    n, D = 4100, 6
    X = rng.randn(n, D)
    ls = np.array([1.1])
    k = O.Kern(kind, D, variance=0.9, lengthscales=1.1, ARD=False)
    out = ctx.empty(n, n)
    dX = ctx.to_device(X)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(_spec(kind, D, ls, 0.9, False)), _p(dX), n, None, 0, 1e-6, _p(out), n))
    ctx.sync()
    Ks = out.cpu().numpy()
    assert np.array_equal(Ks, Ks.T)
    dg = np.diag(Ks)
    assert np.all(dg == dg[0])
    assert_allclose(Ks, k.K(O.NP, X) + 1e-6 * np.eye(n), rtol=1e-11, atol=1e-13)


def test_gram_both_forms_agree_bitwise_on_a_shared_block(ctx):
    """The streaming form and the sub-chunk form are two launches of ONE tile arithmetic: a 512 x 33 000 result (streaming) and the
    512 x 4000 result of its first columns (sub-chunk form) carry the same bits where they overlap."""
    rng = np.random.RandomState(8)
    n, n2, n2s, D = 512, 33000, 4000, 8
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = np.array([0.9])
    sp = _spec("rbf", D, ls, 1.0, False)
    dX, dX2 = ctx.to_device(X), ctx.to_device(X2)
    big, small = ctx.empty(n, n2), ctx.empty(n, n2s)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(sp), _p(dX), n, _p(dX2), n2, 0.0, _p(big), n2))
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(sp), _p(dX), n, _p(dX2), n2s, 0.0, _p(small), n2s))
    ctx.sync()
    assert np.array_equal(big.cpu().numpy()[:, :n2s], small.cpu().numpy())


# ---------------------------------------------------------------- launch counter
def test_launch_counter_counts_the_launches_of_a_step():
    rng = np.random.RandomState(3)
    N, D, M, S = 64, 3, 16, 2
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    _, _, model = make_case(X, Y, X[:M] + 0.01 * rng.randn(M, D), [kern_spec("rbf", D), kern_spec("rbf", D)], S=S, num_data=N)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    lib = _lib.load()
    model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)
    c0 = lib.dsdgp_launch_count()
    for _ in range(5):
        model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)
    per_step = (lib.dsdgp_launch_count() - c0) / 5.0
    assert per_step == int(per_step) and 5 <= per_step <= 40, per_step


# ---------------------------------------------------------------- (q_mu, q_sqrt)-only pass on a FRESH model / a new shape
def _qmodel(rng, N=96, D=4, M=24, S=3, L=3):
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)][:L]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=4 * N)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 2)]
    return spec, state, model, X, Y, zs


@pytest.mark.parametrize("force", ["gemm_mp=0", "gemm_mp=16"])
@pytest.mark.parametrize("first", [0, 1, 2])
def test_q_only_pass_as_the_first_evaluation_of_a_model(monkeypatch, force, first):
    """No full reverse pass has run before: whatever the skipped chain / products would have left (E, GW, [X^T;1], partial sums) does
    not exist yet.  The (q_mu, q_sqrt) entries of the layers >= first must still be the oracle's, and finite."""
    monkeypatch.setenv("DSDGP_FORCE", force)
    spec, state, model, X, Y, zs = _qmodel(np.random.RandomState(31 + first))
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, 3, num_data=4 * X.shape[0])
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=first, grad_q_only=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for l in range(first, 3):
        for name in ("q_mu", "q_sqrt"):
            k = f"l{l}.{name}"
            a = np.asarray(grads[k])
            assert np.all(np.isfinite(a)), k
            assert np.max(np.abs(-g[k] - a)) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), k


def test_q_only_pass_after_a_change_of_the_minibatch_shape():
    """ensure_plan re-plans the split-K jobs for the new (n, S): a q-only evaluation right after the change reads none of the old plan's
    partial sums."""
    rng = np.random.RandomState(37)
    spec, state, model, X, Y, zs = _qmodel(rng)
    model._build_likelihood(X, Y, zs=zs, with_grad=True)                       # full pass at 96 rows
    n2 = 40
    X2, Y2 = X[:n2], Y[:n2]
    zs2 = [z[:, :n2] for z in zs]
    ref, g = OM.elbo_and_grad(spec, state, X2, Y2, zs2, 3, num_data=4 * X.shape[0])
    got = model._build_likelihood(X2, Y2, zs=zs2, with_grad=True, grad_from_layer=1, grad_q_only=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for l in (1, 2):
        for name in ("q_mu", "q_sqrt"):
            k = f"l{l}.{name}"
            assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), k


# ---------------------------------------------------------------- conditional on more rows than the model's scratch
@pytest.mark.parametrize("force", ["gemm_mp=16", "gemm_mp=0"])
def test_layer_conditional_on_more_rows_than_the_training_extent(monkeypatch, force):
    """conditional_ND (layers.py:178) of one layer on 5 x the rows the model was sized for: the GEMM-formulated pass walks them in
    chunks of its scratch, the chains take any row count."""
    monkeypatch.setenv("DSDGP_FORCE", force)
    rng = np.random.RandomState(41)
    N, D, M, S = 48, 4, 32, 2
    X, Y = rng.randn(N, D), rng.randn(N, 3)
    spec, state, model = make_case(X, Y, X[:M] + 0.05 * rng.randn(M, D), [kern_spec("rbf", D, 1.2, 0.8)], S=S, num_data=N)
    model._build_likelihood(X, Y, zs=[rng.randn(S, N, 3)])                     # sizes the workspace for (N, S)
    eng = model.engine()
    n = 5 * S * N + 7
    Xn = rng.randn(n, D)
    # straight through the C-ABI: the host mirror would re-create the model with a larger extent first (Engine._ensure)
    dX, mean, var = eng.ctx.to_device(Xn), eng.ctx.empty(n, 3), eng.ctx.empty(n, 3)
    eng.ctx.torch.cuda.current_stream().synchronize()
    _lib.check(eng.lib.dsdgp_model_layer_conditional(eng.model, 0, _p(dX), n, _p(mean), _p(var)))
    eng.ctx.sync()
    om = OM.build(O.NP, spec, state, S, N)
    mo, vo = om.layers[0].conditional_ND(O.NP, Xn)
    assert_allclose(mean.cpu().numpy(), mo, rtol=1e-9, atol=1e-10)
    assert_allclose(var.cpu().numpy(), vo, rtol=1e-9, atol=1e-10)


# ---------------------------------------------------------------- the largest padded inducing count (Mp = 2048)
def test_largest_inducing_count_mp_2048():
    """M = 2000 pads to Mp = 2048 = DSDGP_MAX_MP: blocked Cholesky / inverse on 16 x 128-blocks, the weight-gradient tilings, the GEMM
    passes and the natural-gradient scratch at the largest size the library accepts — layer outputs, ELBO and every gradient block
    against the oracle (small N: the oracle's cost is the M^3 algebra)."""
    rng = np.random.RandomState(47)
    N, D, S, M = 32, 6, 2, 2000
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = rng.randn(M, D) * 1.6
    spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", D, 1.0, 1.0), kern_spec("rbf", D, 0.9, 1.2)], S=S, num_data=3000)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    for l in range(2):
        assert_allclose(Fm[l], Fm_o[l], rtol=1e-8, atol=1e-9)
        assert_allclose(Fv[l], Fv_o[l], rtol=1e-8, atol=1e-9)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=3000)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - grads[k])) <= 1e-6 * (np.max(np.abs(g[k])) + 1e-12), k
    assert np.isfinite(model.train_step(0.01, X=X, Y=Y, zs=zs, sync=True))


# ---------------------------------------------------------------- backward chain at M = 128 on more row blocks than one round of the
# 8-wave instance holds (> 768): the 4-wave instance with the paired d-loop (layer_sm.hip: SM_BWD_RESIDENT_8W)
@pytest.mark.parametrize("white", [False, True])
def test_backward_chain_m128_beyond_one_resident_round(white):
    """(S n)/16 = 782 row blocks for the inner layers: dl/d(everything) against the oracle's reverse pass (layers.py:71-114 through
    dgp.py:139-147), RBF and Matern52 layers, D_out = 5 / 5 / 2."""
    rng = np.random.RandomState(53 + int(white))
    N, D, M, S = 1250, 5, 128, 10
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=4 * N, white=white)
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 2)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=4 * N)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:                                     # EVERY parameter block of every layer
        assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), k


# ---------------------------------------------------------------- reverse pass of a deep model: the upper layers' weight-gradient
# products follow the lowest layer's on the main stream (model_schedule.hpp: defer_upper, from four layers in the pass)
@pytest.mark.parametrize("force", ["wg_defer=0", "wg_defer=2", ""])
def test_five_layer_reverse_pass_in_both_product_schedules(monkeypatch, force):
    """n S Mp = 512 x 8 x 256 = 2^20: the stream-overlapped schedule is on.  Five layers (dgp.py:139-147 looped over the layers), every
    gradient block against the oracle's reverse pass with the products of layers 2 .. 4 launched the old way (side stream, behind an
    event at the end of each layer's chain), deferred, and by the default rule (deferred: five layers >= four)."""
    if force:
        monkeypatch.setenv("DSDGP_FORCE", force)
    else:
        monkeypatch.delenv("DSDGP_FORCE", raising=False)
    rng = np.random.RandomState(61)
    N, D, M, S, L = 512, 4, 256, 8, 5
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[rng.permutation(N)[:M]] + 0.05 * rng.randn(M, D)
    kinds = ["rbf", "matern52", "rbf", "rbf", "matern52"]
    specs = [kern_spec(k, D, 0.9 + 0.1 * i, 1.0 + 0.05 * i) for i, k in enumerate(kinds)]
    spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=3 * N)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 1)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=3 * N)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), k
    # a second evaluation (the plan and every buffer reused) returns the same bits
    got2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    grads2 = model.engine().gradient_dict()
    assert got2 == got
    for k in g:
        assert np.array_equal(np.asarray(grads[k]), np.asarray(grads2[k])), k
