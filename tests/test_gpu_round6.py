"""Round 6: the reference's own edge case on the HIP path — `settings.jitter = 1e-18` on the shapes of its DGP == SVGP test
(/root/reference/tests/test_dgp.py:7-11,29-34,65-117: N = 19, Z = X, Matern52 with lengthscale 0.5, D_Y = 3, q_sqrt = 1e-3 I, Gaussian
likelihood with variance 0.01; the two-layer model with the inner kernel's variance at 1e-24 through a Parameter WITHOUT transform),
a pivot that is exactly zero at that jitter, and the full-batch form of the training step (`minibatch_size=None`) with an explicit
`num_data` (dgp.py:49,54-55)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from doubly_stochastic_dgp import _lib, settings
from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu
NP = O.NP


def _reference_setup():
    """tests/test_dgp.py:29-43 (np.random.seed(0), then the draws in the reference's order)."""
    rng = np.random.RandomState(0)
    N, Ns, D_X, D_Y = 19, 20, 2, 3
    X = rng.uniform(size=(N, D_X))
    Xs = rng.uniform(size=(Ns, D_X))
    q_mu = rng.randn(N, D_Y)
    q_sqrt = 0.001 * np.eye(N)[None, :, :] * np.ones((D_Y, 1, 1))
    Y = rng.randn(N, D_Y)
    return X, Xs, q_mu, q_sqrt, Y


def _build_pair(L, white, jitter=1e-18):
    """(HIP model, oracle model, spec, state) of tests/test_dgp.py:66-91 with L layers."""
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import Gaussian, Matern52, Parameter
    X, Xs, q_mu, q_sqrt, Y = _reference_setup()
    k_out = dict(kind="matern52", input_dim=2, variance=1.0, lengthscales=0.5, ARD=False, white_variance=None)
    k_in = dict(k_out, variance=1e-24)
    specs = [k_in] * (L - 1) + [k_out]
    lds = O.init_layers_linear(X, Y, X, specs, white=white, jitter=jitter)
    lds[-1]["q_mu"], lds[-1]["q_sqrt"] = q_mu, q_sqrt
    for ld in lds[:-1]:
        ld["kvar_identity"] = True
    sl, state = OM.state_from_layers(lds, lik_variance=0.01)
    spec = dict(jitter=jitter, white=white, likelihood="gaussian", layers=sl)
    kerns = []
    for _ in range(L - 1):
        k = Matern52(2, lengthscales=0.5)
        k.variance = Parameter(1e-24)                 # the reference's NoTransformMatern52 (tests/test_dgp.py:79-85)
        kerns.append(k)
    kerns.append(Matern52(2, lengthscales=0.5))
    with settings.temp_jitter(jitter):
        lik = Gaussian()
        lik.variance = 0.01
        model = DGP(X, Y, X, kerns, lik, white=white, num_samples=2)
    model.layers[-1].q_mu = q_mu
    model.layers[-1].q_sqrt = q_sqrt
    return model, spec, state, (X, Xs, Y)


@pytest.mark.parametrize("white", [True, False])
@pytest.mark.parametrize("L", [1, 2])
def test_reference_edge_case_jitter_1e18_against_the_oracle(L, white):
    """HIP path against the oracle at the reference's jitter and its own bars (rtol = atol = 1e-7 one layer, 1e-6 two): the bound with
    explicit z, mean / variance / full covariance at the test points, every gradient block.  The in-kernel LDS Cholesky factors
    K(X, X) + 1e-18 I here — the GPU suite otherwise runs at 1e-6."""
    jitter = 1e-18
    tol = 1e-7 if L == 1 else 1e-6
    model, spec, state, (X, Xs, Y) = _build_pair(L, white, jitter)
    rng = np.random.RandomState(6)
    zs = [rng.randn(2, 19, 2)] * (L - 1) + [rng.randn(2, 19, 3)]
    zs_s = [np.zeros((1, 20, 2))] * (L - 1) + [np.zeros((1, 20, 3))]
    with settings.temp_jitter(jitter):
        ref, gref = OM.elbo_and_grad(spec, state, X, Y, zs, 2)
        got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
        assert np.isfinite(got)
        assert_allclose(got, ref, rtol=tol, atol=tol)
        g = model.engine().gradient_dict()
        for k in gref:
            assert np.all(np.isfinite(g[k])), k
            if L == 2 and k.startswith("l0."):
                continue          # the numerically absent inner layer: adjoints between 1e-23 (Z) and 1e12 (variance = 1e-24), no relative bar
            err = np.max(np.abs(-gref[k] - g[k])) / (np.max(np.abs(gref[k])) + 1e-12)
            assert err <= 10 * tol, (k, err)
        _, Fm_o, Fv_o = OM.propagate(spec, state, Xs, zs_s, 1)
        m, v = model._build_predict(Xs, S=1, zs=zs_s)
        assert_allclose(m, Fm_o[-1], rtol=tol, atol=tol)
        assert_allclose(v, Fv_o[-1], rtol=tol, atol=tol)
        _, Fm_f, Fv_f = OM.propagate(spec, state, Xs, zs_s, 1, full_cov=True)
        mf, vf = model._build_predict(Xs, full_cov=True, S=1, zs=zs_s)
        assert_allclose(mf, Fm_f[-1], rtol=tol, atol=tol)
        assert_allclose(vf, Fv_f[-1], rtol=10 * tol, atol=10 * tol)


@pytest.mark.parametrize("white", [True, False])
def test_reference_identity_two_layers_equal_one_on_the_device(white):
    """The reference's assertion itself (tests/test_dgp.py:86-117, L = 2 vs the single layer), both sides on the HIP path."""
    jitter = 1e-18
    m1, _, _, (X, Xs, Y) = _build_pair(1, white, jitter)
    m2, _, _, _ = _build_pair(2, white, jitter)
    rng = np.random.RandomState(7)
    z_in, z_out = rng.randn(2, 19, 2), rng.randn(2, 19, 3)
    with settings.temp_jitter(jitter):
        L1 = m1.compute_log_likelihood(X, Y, zs=[z_out])
        L2 = m2.compute_log_likelihood(X, Y, zs=[z_in, z_out])
        assert_allclose(L1, L2, rtol=1e-6, atol=1e-6)
        p1, v1 = m1.predict_f(Xs, 1)
        p2, v2 = m2.predict_f(Xs, 1)
        assert_allclose(p2[0], p1[0], rtol=1e-6, atol=1e-6)
        assert_allclose(v2[0], v1[0], rtol=1e-6, atol=1e-6)
        y1, yv1 = m1.predict_y(Xs, 1)
        y2, yv2 = m2.predict_y(Xs, 1)
        assert_allclose(y2, y1, rtol=1e-6, atol=1e-6)
        assert_allclose(yv2, yv1, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M", [20, 300])
def test_exactly_zero_pivot_at_jitter_1e18_raises(M):
    """Two identical inducing points, one input dimension: K[0][0] = 1 + 1e-18 = 1 and K[1][0] = 1 exactly, so pivot 1 of the
    factorisation is exactly 0 — "Cholesky decomposition was not successful" ([UPSTREAM] tf.cholesky), never a NaN result.  M = 20: the
    LDS factorisation of the head launch; M = 300: the blocked multi-workgroup one."""
    rng = np.random.RandomState(3)
    X = rng.randn(400, 1)
    Z = np.concatenate([X[:1], X[:1], X[1:M - 1]])
    with settings.temp_jitter(1e-6):
        spec, state, model = make_case(X, rng.randn(400, 1), Z, [kern_spec("rbf", 1)], S=1, randomize=False)
    with settings.temp_jitter(1e-18):
        with pytest.raises(_lib.CholeskyError):
            model.layers[0].conditional_ND(X)
        with pytest.raises(_lib.CholeskyError):
            model.compute_log_likelihood(X, model.Y_data)
        model.train_step(0.01)                                   # asynchronous: reported by the next synchronising step
        with pytest.raises(_lib.CholeskyError):
            model.train_step(0.01, sync=True)
    with settings.temp_jitter(1e-6):                             # and the model is usable again once the jitter is sane
        assert np.isfinite(model.compute_log_likelihood(X, model.Y_data))


@pytest.mark.parametrize("L", [1, 2])
def test_full_batch_training_step_with_num_data_override(L):
    """`DGP(..., minibatch_size=None, num_data=5000)` on 150 rows (dgp.py:49: `num_data or X.shape[0]`, dgp.py:54-55: no Minibatch) through
    `train_step` with no data arguments: three Adam steps against the oracle's, scale = 5000 / 150 (dgp.py:96).  One layer: the bound does
    not depend on z (Gaussian likelihood reads mean and variance), so the device's own draws are used; two layers: explicit z."""
    rng = np.random.RandomState(12)
    N, D, M, S, num_data = 150, 3, 24, 3, 5000
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", D)] * L, S=S, num_data=num_data, minibatch_size=None, seed=2)
    assert model.num_data == num_data and model.minibatch_size is None
    zs = [rng.randn(S, N, D)] * (L - 1) + [rng.randn(S, N, 1)]
    keys = sorted(state.keys())
    th = {k: state[k].copy() for k in keys}
    m = {k: np.zeros_like(state[k]) for k in keys}
    v = {k: np.zeros_like(state[k]) for k in keys}
    for t in range(1, 4):
        _, g = OM.elbo_and_grad(spec, th, X, Y, zs, S, num_data=num_data)
        for k in keys:
            O.adam_step(th[k], -g[k], m[k], v[k], t, lr=0.01)
        if L == 1:
            model.train_step(0.01)
        else:
            model.train_step(0.01, zs=zs)
    ref = OM.elbo(spec, th, X, Y, zs, S, num_data=num_data)
    got = model.compute_log_likelihood(zs=zs)                    # X = None: the full data again
    assert_allclose(got, ref, rtol=1e-7)
    assert_allclose(model.layers[-1].q_mu.value, th[f"l{L - 1}.q_mu"], rtol=1e-6, atol=1e-8)
    assert_allclose(model.likelihood.likelihood.variance.value, O.positive_forward(O.NP, th["lik_variance_raw"]), rtol=1e-7)
    # the same model without the override scales the data term by 1 instead of 5000 / 150
    spec1, state1, model1 = make_case(X, Y, Z, [kern_spec("rbf", D)] * L, S=S, num_data=None, minibatch_size=None, seed=2)
    assert_allclose(model1.compute_log_likelihood(zs=zs), OM.elbo(spec1, state1, X, Y, zs, S), rtol=1e-9)


# ---------------------------------------------------------------- the last layer of a training step as one launch (layer_last.hip)
@pytest.mark.parametrize("force", ["last_fuse=0", ""])
@pytest.mark.parametrize("shape", [dict(N=203, D=8, M=128, S=4, L=3, kind="rbf"),            # ragged rows, config-2 slice
                                   dict(N=181, D=5, M=120, S=3, L=2, kind="matern52"),       # padded inducing rows, two layers (dX path)
                                   dict(N=270, D=9, M=256, S=4, L=3, kind="rbf"),            # the eight-wave instance (config 3's last layer)
                                   dict(N=345, D=3, M=250, S=3, L=2, kind="matern52"),
                                   dict(N=40, D=4, M=128, S=2, L=2, kind="rbf")])            # too few rows for the algebraic dl/dKu: the two chains
def test_last_layer_fused_launch_against_the_oracle_and_the_two_chains(monkeypatch, force, shape):
    """D_out = 1, Gaussian likelihood, non-white: forward chain + likelihood + reverse pass of the last layer in ONE launch (its abar
    from the triangular pair q_sqrt (q_sqrt^T a) with c still in registers) against the oracle's autograd, with the launch on (default)
    and off (`last_fuse=0`: the two chains).  Launch count: one fewer with the fusion."""
    if force:
        monkeypatch.setenv("DSDGP_FORCE", force)
    else:
        monkeypatch.delenv("DSDGP_FORCE", raising=False)
    from doubly_stochastic_dgp.engine import Context
    c = shape
    rng = np.random.RandomState(c["N"] + c["M"])
    N, D, M, S, L = c["N"], c["D"], c["M"], c["S"], c["L"]
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = (X[rng.permutation(N)[:M]] if M <= N else rng.randn(M, D)) + 0.02 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec(c["kind"], D, 1.1, 1.2)] * L, S=S, num_data=4 * N, seed=4)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 1)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=4 * N)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), k
    lib = Context.get().lib
    lib.dsdgp_launch_count.restype = __import__("ctypes").c_int64
    n0 = lib.dsdgp_launch_count()
    got2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    n1 = lib.dsdgp_launch_count()
    assert got2 == got                                            # same bits on a second evaluation
    test_last_layer_fused_launch_against_the_oracle_and_the_two_chains.launches[(force, c["M"], c["N"])] = n1 - n0
    both = test_last_layer_fused_launch_against_the_oracle_and_the_two_chains.launches
    if ("", c["M"], c["N"]) in both and ("last_fuse=0", c["M"], c["N"]) in both:
        # the launch is taken where the layer's dl/dKu is assembled algebraically (4 D_out Mp <= S N rows: no E store in the chain)
        fused = 4 * (128 if M <= 128 else 256) <= S * N
        assert both[("", c["M"], c["N"])] == both[("last_fuse=0", c["M"], c["N"])] - (1 if fused else 0)
    # three optimiser steps through the fused launch track the oracle's Adam
    keys = sorted(state.keys())
    th = {k: state[k].copy() for k in keys}
    mm = {k: np.zeros_like(state[k]) for k in keys}
    vv = {k: np.zeros_like(state[k]) for k in keys}
    for t in range(1, 3):
        _, gg = OM.elbo_and_grad(spec, th, X, Y, zs, S, num_data=4 * N)
        for k in keys:
            O.adam_step(th[k], -gg[k], mm[k], vv[k], t, lr=0.01)
        model.train_step(0.01, X=X, Y=Y, zs=zs)
    assert_allclose(model.compute_log_likelihood(X, Y, zs=zs), OM.elbo(spec, th, X, Y, zs, S, num_data=4 * N), rtol=1e-7)


test_last_layer_fused_launch_against_the_oracle_and_the_two_chains.launches = {}


# ---------------------------------------------------------------- split-K reduction inside the weight-gradient launch (wgrad.hip, WgradJob::fin)
@pytest.mark.parametrize("force", ["wg_red=2", "wg_red=0", ""])
@pytest.mark.parametrize("shape", [dict(N=300, D=5, M=100, S=5, L=3, kind="rbf", white=False),      # Mp = 112 on 128-row tiles: fin_rows < tile rows, several splits
                                   dict(N=64, D=3, M=200, S=2, L=2, kind="matern52", white=False),  # 4 x 4 tiles of 64, few rows: one split by the plan
                                   dict(N=150, D=4, M=130, S=3, L=2, kind="rbf", white=True),       # white: the adjoint of Lu reads E A^T (non-symmetric job)
                                   dict(N=40, D=20, M=64, S=2, L=2, kind="rbf", white=False)])      # wide thin job (D_in + 1 > 16 columns), dense dl/dKu
def test_in_launch_split_k_reduction_against_the_oracle(monkeypatch, force, shape):
    """The weight-gradient launch adding its own split-K partials (last arrival per tile, tickets; `wg_red=2`: every plan), the direct
    register -> result form of one-split plans (default) and the reduction launch (`wg_red=0`) give the oracle's gradient — symmetric
    P_d tiles with their mirrors, the diagonal tiles' uncomputed upper blocks, thin jobs with partial column tiles, E A^T — and the
    same ELBO bits on a second evaluation (tickets back at zero)."""
    if force:
        monkeypatch.setenv("DSDGP_FORCE", force)
    else:
        monkeypatch.delenv("DSDGP_FORCE", raising=False)
    c = shape
    rng = np.random.RandomState(c["N"] + c["M"])
    N, D, M, S, L = c["N"], c["D"], c["M"], c["S"], c["L"]
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = (X[rng.permutation(N)[:M]] if M <= N else rng.randn(M, D)) + 0.02 * rng.randn(M, D)
    spec, state, model = make_case(X, Y, Z, [kern_spec(c["kind"], D, 1.1, 1.2)] * L, S=S, num_data=3 * N, seed=6, white=c["white"])
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 2)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=3 * N)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-9)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-7 * (np.max(np.abs(g[k])) + 1e-12), (force, k)
    g1 = {k: np.asarray(v).copy() for k, v in grads.items()}
    got2 = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert got2 == got
    g2 = model.engine().gradient_dict()
    for k in g1:
        assert np.array_equal(g1[k], np.asarray(g2[k])), (force, k)


# ---------------------------------------------------------------- blocked Cholesky with look-ahead (linalg.hip: k_chol_panel, side workgroups)
@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("M,L", [(180, 1), (192, 2), (250, 1), (300, 3), (448, 1), (570, 2)])
def test_lookahead_blocked_cholesky_in_the_model_against_the_oracle(M, L, white):
    """Ku's factor, inverse factor (block rows formed by side workgroups of the factor launches, the last row split over two launches)
    and log det through the look-ahead sequence from Mp = 192 on — 128 + 64 ragged (180 -> 192), two full blocks (250 -> 256), 320, 448
    (ragged again), 570 -> 640; one matrix and batches of L — against the oracle: ELBO, every gradient (white = True reads Lu itself: the
    parked panels must be moved back and the upper blocks zeroed), then a natural-gradient step (its own factorisation + inverse, whose
    factor is never moved back)."""
    from doubly_stochastic_dgp.training import NatGradOptimizer
    rng = np.random.RandomState(M + L)
    N, D, S = 24, 3, 2
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D) * 2.0
    specs = [kern_spec("rbf", D, 1.0, 1.0)] * (L - 1) + [kern_spec("matern52", D, 1.3, 0.9)]
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=500)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 2)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=500)
    got = model._build_likelihood(X, Y, zs=zs, with_grad=True)
    assert_allclose(got, ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        err = np.max(np.abs(-g[k] - np.asarray(grads[k]))) / (np.max(np.abs(g[k])) + 1e-12)
        assert err <= 1e-6, (k, err)
    # the natural-gradient step from the initial (q_mu, q_sqrt): make_case's random lower-triangular q_sqrt is exponentially
    # ill-conditioned at these M (the one-workgroup kernels miss rtol 1e-6 on it just the same)
    spec, state, model = make_case(X, Y, Z, specs, white=white, S=S, num_data=500, randomize=False)
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=500)
    assert_allclose(model._build_likelihood(X, Y, zs=zs, with_grad=True), ref, rtol=1e-8)
    last = "l%d" % (L - 1)
    mu, sq = O.natgrad_step(state[last + ".q_mu"], state[last + ".q_sqrt"], -g[last + ".q_mu"], -g[last + ".q_sqrt"], 0.1)
    lay = model.layers[-1]
    NatGradOptimizer(0.1).minimize(model, var_list=[[lay.q_mu, lay.q_sqrt]], maxiter=1, X=X, Y=Y, zs=zs)
    assert_allclose(lay.q_mu.value, mu, rtol=1e-6, atol=1e-8)
    assert_allclose(lay.q_sqrt.value, sq, rtol=1e-6, atol=1e-8)
    # a second evaluation on the new parameters (Ku unchanged: its factor is kept; q moved) still matches the oracle
    state2 = dict(state)
    state2[last + ".q_mu"], state2[last + ".q_sqrt"] = mu, sq
    assert_allclose(model._build_likelihood(X, Y, zs=zs), OM.elbo(spec, state2, X, Y, zs, S, num_data=500), rtol=1e-7)


@pytest.mark.parametrize("M,L", [(300, 2), (570, 1)])
def test_plain_blocked_sequence_still_matches_the_oracle(monkeypatch, M, L):
    """`DSDGP_CHOL_LOOKAHEAD=0` (read when a plan is built): panel and trailing update as GEMM launches, the inverse by recursive doubling —
    the form that remains for more than 16 blocks — against the oracle, white = True (reads Lu)."""
    monkeypatch.setenv("DSDGP_CHOL_LOOKAHEAD", "0")
    rng = np.random.RandomState(M)
    N, D, S = 24, 3, 2
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    Z = rng.randn(M, D) * 2.0
    spec, state, model = make_case(X, Y, Z, [kern_spec("matern52", D, 1.3, 0.9)] * L, white=True, S=S, num_data=500)
    zs = [rng.randn(S, N, D) for _ in range(L - 1)] + [rng.randn(S, N, 2)]
    ref, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=500)
    assert_allclose(model._build_likelihood(X, Y, zs=zs, with_grad=True), ref, rtol=1e-8)
    grads = model.engine().gradient_dict()
    for k in g:
        assert np.max(np.abs(-g[k] - np.asarray(grads[k]))) <= 1e-6 * (np.max(np.abs(g[k])) + 1e-12), k


def test_lookahead_cholesky_batch_with_one_indefinite_matrix():
    """dsdgp_potrf on a batch whose middle matrix loses positive definiteness in its third block: the call reports that pivot, the
    other matrices of the batch come out factored."""
    import ctypes as C
    from doubly_stochastic_dgp.engine import Context
    ctx = Context.get()
    n, batch = 640, 3
    rng = np.random.RandomState(4)
    A0 = rng.randn(n, n)
    A0 = A0 @ A0.T + n * np.eye(n)
    A = np.stack([A0, A0.copy(), 2.0 * A0])
    A[1, 300, :] = 0.0
    A[1, :, 300] = 0.0
    A[1, 300, 300] = -1.0
    dA = ctx.to_device(A)
    info = C.c_int(0)
    ctx.sync()
    rc = ctx.lib.dsdgp_potrf(ctx.handle, batch, n, C.c_void_p(dA.data_ptr()), n, n * n, C.byref(info))
    assert rc == _lib.ERR_NOT_SPD and info.value == 301
    L = dA.cpu().numpy()
    for b in (0, 2):
        assert_allclose(np.tril(L[b]), np.linalg.cholesky(A[b]), rtol=1e-10, atol=1e-10)


def test_lookahead_cholesky_reports_the_failing_pivot_of_a_later_block():
    import ctypes as C
    from doubly_stochastic_dgp.engine import Context
    ctx = Context.get()
    n = 200                                     # padded 256: two blocks, the bad pivot in the second
    A = np.eye(n)
    A[150, 150] = -1.0
    dA = ctx.to_device(A)
    info = C.c_int(0)
    ctx.sync()
    rc = ctx.lib.dsdgp_potrf(ctx.handle, 1, n, C.c_void_p(dA.data_ptr()), n, n * n, C.byref(info))
    assert rc == _lib.ERR_NOT_SPD and info.value == 151
