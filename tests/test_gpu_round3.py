"""-m gpu tests added in round 3: the reverse pass's stream schedule (weight-gradient products of a layer under the next layer's
backward chain, per-layer pipelined tail, reduction ahead of the join), the fused head kernel, the small-matrix GEMM path, the blocked / batched
triangular solve and the fused tail.  Every schedule variant must reproduce the serial schedule bit for bit (all reductions on
the path are fixed-order); values are compared with the oracle elsewhere (tests/test_gpu_parity.py).
"""
import ctypes as C

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


def _train_state(monkeypatch, force, no_overlap, steps=12, white=False):
    """Parameters after `steps` Adam steps of a 3-layer model large enough for the two-stream schedule (n S Mp >= 2^20)."""
    import os
    monkeypatch.setenv("DSDGP_FORCE", force)
    os.environ["DSDGP_NO_OVERLAP"] = "1" if no_overlap else "0"
    try:
        rng = np.random.RandomState(5)
        N, D, M, S = 1000, 5, 96, 14
        X, Y = rng.randn(N, D), rng.randn(N, 2)
        Z = X[:M] + 0.05 * rng.randn(M, D)
        specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)]
        _, _, model = make_case(X, Y, Z, specs, S=S, num_data=N, q_sqrt_scale=1e-2, minibatch_size=800, white=white)
        for _ in range(steps):
            model.train_step(0.01)
        elbo = model.train_step(0.01, sync=True)
        eng = model.engine()
        g = eng.grad.cpu().numpy().copy()
        eng.sync_to_host()
        th = np.concatenate([np.ravel(l.q_mu.value) for l in model.layers] + [np.ravel(l.feature.Z.value) for l in model.layers] +
                            [np.ravel(l.q_sqrt.value) for l in model.layers])
        return elbo, g, th
    finally:
        os.environ["DSDGP_NO_OVERLAP"] = "0"


@pytest.mark.parametrize("white", [False, True])
def test_reverse_pass_schedules_are_bitwise_neutral(monkeypatch, white):
    """serial (one stream) == weight-gradient products under the next chain == + per-layer pipelined tail == reduction after / ahead
    of the stream join == plain event record behind the head launch."""
    ref = _train_state(monkeypatch, "pipe_tail=0", True, white=white)
    assert np.isfinite(ref[0])
    for force in ("pipe_tail=0", "pipe_tail=1", "red_ahead=0", "red_ahead=1,ext_ev=0"):
        got = _train_state(monkeypatch, force, False, white=white)
        assert got[0] == ref[0], force
        assert np.array_equal(got[1], ref[1]), force
        assert np.array_equal(got[2], ref[2]), force


def test_pipelined_tail_with_pruned_reverse_pass(monkeypatch):
    """grad_from_layer > 0 (NatGradOptimizer's var_list): the schedule variants agree on the entries of the participating layers."""
    outs = []
    for force in ("pipe_tail=0", "pipe_tail=1"):
        monkeypatch.setenv("DSDGP_FORCE", force)
        rng = np.random.RandomState(9)
        N, D, M, S = 800, 4, 128, 12
        X, Y = rng.randn(N, D), rng.randn(N, 1)
        Z = X[:M] + 0.05 * rng.randn(M, D)
        specs = [kern_spec("rbf", D), kern_spec("rbf", D), kern_spec("rbf", D)]
        _, _, model = make_case(X, Y, Z, specs, S=S, num_data=N)
        zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 1)]
        e = model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=2)
        g = model.engine().gradient_dict()
        outs.append((e, g["l2.q_mu"].copy(), g["l2.q_sqrt"].copy()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


# ---------------------------------------------------------------- blocked / batched triangular solve (K4)
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("n,nrhs,batch,shared", [(128, 257, 8, True), (100, 64, 3, False), (300, 50, 2, True), (17, 5, 4, False)])
def test_trsm_batched(ctx, trans, n, nrhs, batch, shared):
    """layers.py:239: tf.matrix_triangular_solve(Lu_tiled, q_sqrt) — `batch` right-hand sides against one shared L (strideL = 0) or
    their own; ragged n and nrhs (partial 16-blocks, partial 64-column workgroups, several 128-row panels)."""
    import scipy.linalg as sla
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n + batch)
    nL = 1 if shared else batch
    L = np.tril(rng.randn(nL, n, n)) + 4 * np.eye(n)
    L = L + np.triu(rng.randn(nL, n, n), 1) * 0 + np.triu(np.full((nL, n, n), 7.0), 1)      # garbage above the diagonal: never read
    B = rng.randn(batch, n, nrhs)
    dL, dB = _dev(ctx, L), _dev(ctx, B)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_trsm_batched(ctx.handle, trans, n, nrhs, batch, _p(dL), n, 0 if shared else n * n, _p(dB), nrhs, n * nrhs))
    ctx.sync()
    got = dB.cpu().numpy()
    for b in range(batch):
        ref = sla.solve_triangular(np.tril(L[0 if shared else b]), B[b], lower=True, trans=trans)
        assert_allclose(got[b], ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())      # random L: the solutions grow to 1e4


def test_trsm_strided_views(ctx):
    """leading dimensions larger than the matrices (sub-matrix views), as a caller holding padded buffers passes them"""
    import scipy.linalg as sla
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(0)
    n, nrhs, ldl, ldb = 200, 90, 256, 128
    Lbuf, Bbuf = rng.randn(n, ldl), rng.randn(n, ldb)
    Lbuf[:, :n] = np.tril(Lbuf[:, :n]) + 5 * np.eye(n)
    dL, dB = _dev(ctx, Lbuf), _dev(ctx, Bbuf)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_trsm(ctx.handle, 0, n, nrhs, _p(dL), ldl, _p(dB), ldb))
    ctx.sync()
    got = dB.cpu().numpy()
    assert_allclose(got[:, :nrhs], sla.solve_triangular(Lbuf[:, :n], Bbuf[:, :nrhs], lower=True), rtol=1e-10, atol=1e-11)
    assert np.array_equal(got[:, nrhs:], Bbuf[:, nrhs:])          # columns beyond nrhs untouched


# ---------------------------------------------------------------- materialised Gram matrices: the MFMA form (D <= 32) and its edges
@pytest.mark.parametrize("kind", ["rbf", "matern52"])
@pytest.mark.parametrize("n,n2,D", [(5, 33, 1), (64, 1000, 8), (65, 999, 16), (130, 515, 17), (40, 2049, 32), (70, 300, 33)])
def test_gram_mfma_shapes(ctx, kind, n, n2, D):
    """k_gram_mfma: one / two / four / eight k-steps (D = 1 .. 32), rows and columns that do not fill the 16 x 32 tiles, odd leading
    dimensions (scalar stores), and D = 33 (falls back to the direct-difference kernel); the symmetric call is EXACTLY symmetric with
    the jitter on its diagonal and nothing else there."""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(n + n2 + D)
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = 0.7 + rng.rand(D)
    k = O.Kern(kind, D, variance=1.4, lengthscales=ls, ARD=True)
    spec = _lib.KernelSpec(kind={"rbf": 0, "matern52": 1}[kind], input_dim=D, ard=1, has_white=0, variance=1.4, white_variance=0.0,
                           lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    dX, dX2 = _dev(ctx, X), _dev(ctx, X2)
    out, outs = ctx.empty(n, n2), ctx.empty(n, n)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, _p(dX2), n2, 0.0, _p(out), n2))
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, None, 0, 1e-6, _p(outs), n))
    ctx.sync()
    assert_allclose(out.cpu().numpy(), k.K(O.NP, X, X2), rtol=1e-11, atol=1e-13)
    Ks = outs.cpu().numpy()
    assert_allclose(Ks, k.K(O.NP, X) + 1e-6 * np.eye(n), rtol=1e-11, atol=1e-13)
    assert np.array_equal(Ks, Ks.T)
    dg = np.diag(Ks)                                  # r2 is exactly 0 on the diagonal: one value, k(0) + jitter
    assert np.all(dg == dg[0])
    if kind == "rbf":
        assert abs(dg[0] - (1.4 + 1e-6)) <= 4e-16     # (Matern-5/2: r = sqrt(r2 + 1e-12) upstream, k(0) is not the variance exactly)


def test_gram_into_a_wider_buffer(ctx):
    """ld_out > n2: the columns beyond n2 stay untouched (16-byte pair stores must not spill over a row's end)"""
    from doubly_stochastic_dgp import _lib
    rng = np.random.RandomState(0)
    n, n2, D, ld = 50, 101, 8, 128
    X, X2 = rng.randn(n, D), rng.randn(n2, D)
    ls = np.array([0.9])
    k = O.Kern("rbf", D, variance=1.0, lengthscales=0.9, ARD=False)
    spec = _lib.KernelSpec(kind=0, input_dim=D, ard=0, has_white=0, variance=1.0, white_variance=0.0, lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    buf = np.full((n, ld), 7.0)
    dX, dX2, dout = _dev(ctx, X), _dev(ctx, X2), _dev(ctx, buf)
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), _p(dX), n, _p(dX2), n2, 0.0, _p(dout), ld))
    ctx.sync()
    got = dout.cpu().numpy()
    assert_allclose(got[:, :n2], k.K(O.NP, X, X2), rtol=1e-11, atol=1e-13)
    assert np.all(got[:, n2:] == 7.0)


# ---------------------------------------------------------------- forward-only evaluations in whitened coordinates
@pytest.mark.parametrize("M,L", [(24, 2), (128, 3), (200, 2)])
def test_forward_only_whitened_form_equals_plain_form(monkeypatch, M, L):
    """predict / ELBO-value evaluations of a non-white model skip a = Lu^-T a1 (mean = a1^T Lu^-1 q_mu, var through Lu^-1 q_sqrt_d);
    DSDGP_FORCE=white_fwd=0 keeps the plain form: same layer outputs to rounding, and both equal the oracle."""
    rng = np.random.RandomState(M)
    N, D, S = 150, 4, 3
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[rng.permutation(N)[:min(M, N)]] + 0.05 * rng.randn(min(M, N), D) if M <= N else rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.2, 0.9)] + [kern_spec("matern52", D, 0.9, 1.1)] * (L - 1)
    zs = [rng.randn(S, N, D)] * (L - 1) + [rng.randn(S, N, 2)]
    outs = []
    for force in ("white_fwd=1", "white_fwd=0"):
        monkeypatch.setenv("DSDGP_FORCE", force)
        spec, state, model = make_case(X, Y, Z, specs, S=S, num_data=N)
        Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
        e = model._build_likelihood(X, Y, zs=zs)
        outs.append((Fs, Fm, Fv, e))
    for l in range(L):
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert_allclose(a[l], b[l], rtol=1e-11, atol=1e-12)
    assert_allclose(outs[0][3], outs[1][3], rtol=1e-12)
    _, Fm_o, Fv_o = OM.propagate(spec, state, X, zs, S)
    assert_allclose(outs[0][1][-1], Fm_o[-1], rtol=1e-9, atol=1e-11)
    assert_allclose(outs[0][2][-1], Fv_o[-1], rtol=1e-9, atol=1e-11)
    assert_allclose(outs[0][3], OM.elbo(spec, state, X, Y, zs, S, num_data=N), rtol=1e-10)


def test_chain_prologue_and_epilogue_fusions_equal_their_kernels(monkeypatch):
    """the first layer's upstream adjoints inside its backward chain's prologue (adj_fuse) and the Gaussian variational expectations
    with their adjoints inside the last forward chain's epilogue (lik_fuse) against the separate launches k_adj_prep / k_lik_gauss:
    same ELBO and gradient up to the order of a few sums."""
    rng = np.random.RandomState(11)
    N, D, M, S = 900, 5, 64, 16
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    specs = [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2), kern_spec("rbf", D, 0.9, 1.0)]
    zs = [rng.randn(S, N, D), rng.randn(S, N, D), rng.randn(S, N, 2)]
    outs = []
    for force in ("adj_fuse=1,lik_fuse=1", "adj_fuse=0,lik_fuse=1", "adj_fuse=1,lik_fuse=0", "adj_fuse=0,lik_fuse=0"):
        monkeypatch.setenv("DSDGP_FORCE", force)
        _, _, model = make_case(X, Y, Z, specs, S=S, num_data=5 * N, q_sqrt_scale=1e-2)
        e = model._build_likelihood(X, Y, zs=zs, with_grad=True)
        outs.append((e, model.engine().grad.cpu().numpy().copy()))
    for e, g in outs[1:]:
        assert_allclose(e, outs[0][0], rtol=1e-13)
        assert np.max(np.abs(g - outs[0][1])) <= 1e-11 * np.max(np.abs(outs[0][1]))
