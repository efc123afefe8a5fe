"""Pins the CPU oracle relationally (SURVEY §4 T1–T8): every identity the reference's own tests assert
(tests/test_dgp.py, tests/test_collapsed.py, tests/test_utils.py under /root/reference) is re-stated with an
*independent* closed form on the other side, because neither GPflow nor TF can be imported here.
"""
import math

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import dgp_oracle as O
from oracle import model as OM

NP = O.NP


def _svgp_closed_form(kern, X, Y, Z, q_mu, q_sqrt, lik_var, white, jitter, Xs):
    """Textbook SVGP (Hensman et al. 2013) bound and predictive, written with dense inverses —
    deliberately NOT the trsm/SK/B op sequence of layers.py."""
    M, D = q_mu.shape
    Kuu = kern.K(NP, Z) + jitter * np.eye(M)
    Luu = np.linalg.cholesky(Kuu)
    if white:                      # q(v)=N(q_mu, q_sqrt q_sqrt^T), u = L v
        m = Luu @ q_mu
        Ssq = np.stack([Luu @ np.tril(q_sqrt[d]) for d in range(D)])
    else:
        m, Ssq = q_mu, np.stack([np.tril(q_sqrt[d]) for d in range(D)])
    S = np.stack([Ssq[d] @ Ssq[d].T for d in range(D)])
    Kinv = np.linalg.inv(Kuu)

    def predict(Xq, full=False):
        Kfu = kern.K(NP, Xq, Z)
        Kff = kern.K(NP, Xq) if full else kern.Kdiag(NP, Xq)
        P = Kfu @ Kinv
        mu = P @ m
        if full:
            cov = np.stack([Kff - P @ Kfu.T + P @ S[d] @ P.T for d in range(D)], -1)
            return mu, cov
        var = np.stack([Kff - np.sum(P * Kfu, 1) + np.sum((P @ S[d]) * P, 1) for d in range(D)], 1)
        return mu, var

    mu, var = predict(X)
    ve = -0.5 * math.log(2 * math.pi) - 0.5 * math.log(lik_var) - 0.5 * ((Y - mu) ** 2 + var) / lik_var
    # textbook KL(N(m,S) || N(0,Kuu)) per output
    KL = 0.0
    for d in range(D):
        _, ld_S = np.linalg.slogdet(S[d])
        _, ld_K = np.linalg.slogdet(Kuu)
        KL += 0.5 * (np.trace(Kinv @ S[d]) + m[:, d] @ Kinv @ m[:, d] - M + ld_K - ld_S)
    return ve.sum() - KL, predict, KL


def _setup(seed=0, N=19, Ns=20, D_X=2, D_Y=3):
    rng = np.random.RandomState(seed)
    X = rng.uniform(size=(N, D_X))
    Xs = rng.uniform(size=(Ns, D_X))
    q_mu = rng.randn(N, D_Y)
    q_sqrt = np.tril(0.3 * rng.randn(D_Y, N, N)) + 0.5 * np.eye(N)[None]
    Y = rng.randn(N, D_Y)
    return X, Xs, q_mu, q_sqrt, Y


# T1 — reference tests/test_dgp.py:65-117: a 1-layer DS-DGP *is* SVGP
@pytest.mark.parametrize("white", [True, False])
@pytest.mark.parametrize("kind", ["matern52", "rbf"])
def test_T1_single_layer_equals_svgp(white, kind):
    X, Xs, q_mu, q_sqrt, Y = _setup()
    jitter = 1e-8
    kern = O.Kern(kind, 2, lengthscales=0.5)
    layer = O.SVGPLayer(kern, X.copy(), q_mu, q_sqrt, O.MeanFn("zero"), white=white, jitter=jitter)
    lik = O.Gaussian(0.01)
    m = O.DGPOracle([layer], lik, num_samples=2)
    z = [np.random.RandomState(1).randn(2, 19, 3)]
    L_dgp = m.build_likelihood(NP, X, Y, z)
    L_ref, predict, _ = _svgp_closed_form(kern, X, Y, X, q_mu, q_sqrt, 0.01, white, jitter, Xs)
    assert_allclose(L_dgp, L_ref, rtol=1e-7, atol=1e-7)                # reference bar: test_dgp.py:101-103
    _, Fm, Fv = m.propagate(NP, Xs, [np.zeros((1, 20, 3))], S=1)
    mu, var = predict(Xs)
    assert_allclose(Fm[0][0], mu, rtol=1e-7, atol=1e-7)
    assert_allclose(Fv[0][0], var, rtol=1e-7, atol=1e-7)
    _, Fm, Fv = m.propagate(NP, Xs, [np.zeros((1, 20, 3))], S=1, full_cov=True)
    mu, cov = predict(Xs, full=True)
    # the dense-inverse closed form itself loses digits on the ill-conditioned Z=X RBF case
    assert_allclose(Fv[0][0], cov, rtol=2e-6, atol=1e-6)


# T2 — white ≡ non-white under (q_mu,q_sqrt) -> (Lu q_mu, Lu q_sqrt)
def test_T2_white_equals_nonwhite():
    X, Xs, q_mu, q_sqrt, Y = _setup(1)
    kern = O.Kern("rbf", 2, variance=1.3, lengthscales=0.7)
    jitter = 1e-8
    Lu = np.linalg.cholesky(kern.K(NP, X) + jitter * np.eye(19))
    lw = O.SVGPLayer(kern, X, q_mu, q_sqrt, O.MeanFn("zero"), white=True, jitter=jitter)
    ln = O.SVGPLayer(kern, X, Lu @ q_mu, np.stack([Lu @ np.tril(q) for q in q_sqrt]), O.MeanFn("zero"),
                     white=False, jitter=jitter)
    mw, vw = lw.conditional_ND(NP, Xs)
    mn, vn = ln.conditional_ND(NP, Xs)
    assert_allclose(mw, mn, rtol=1e-9, atol=1e-10)
    assert_allclose(vw, vn, rtol=1e-8, atol=1e-10)
    assert_allclose(lw.KL(NP), ln.KL(NP), rtol=1e-10)


# T3 — KL() ≡ textbook Gaussian KL (layers.py:221-246)
@pytest.mark.parametrize("white", [True, False])
def test_T3_KL_textbook(white):
    X, Xs, q_mu, q_sqrt, Y = _setup(2)
    kern = O.Kern("rbf", 2, lengthscales=0.6)
    layer = O.SVGPLayer(kern, X, q_mu, q_sqrt, O.MeanFn("zero"), white=white, jitter=1e-8)
    _, _, KL = _svgp_closed_form(kern, X, Y, X, q_mu, q_sqrt, 1.0, white, 1e-8, Xs)
    assert_allclose(layer.KL(NP), KL, rtol=1e-7)   # dense-inverse textbook side is the less accurate one


# T5 — Z=X, tiny jitter, optimal q(u) ⇒ exact GPR marginal likelihood (tests/test_collapsed.py:30-54)
def test_T5_optimal_q_gives_gpr():
    rng = np.random.RandomState(3)
    N, s2 = 12, 0.1
    X = rng.uniform(size=(N, 1)) * 3
    Y = np.sin(3 * X) + 0.1 * rng.randn(N, 1)
    kern = O.Kern("rbf", 1, lengthscales=0.8)
    jitter = 1e-9
    Kuu = kern.K(NP, X) + jitter * np.eye(N)
    G = np.linalg.solve(Kuu + s2 * np.eye(N), Kuu)
    Sig = Kuu - Kuu @ G                                                # optimal S for Z=X (posterior cov at X)
    Sig = 0.5 * (Sig + Sig.T)
    m_opt = Kuu @ np.linalg.solve(Kuu + s2 * np.eye(N), Y)
    layer = O.SVGPLayer(kern, X, m_opt, np.linalg.cholesky(Sig)[None], O.MeanFn("zero"), white=False, jitter=jitter)
    elbo = O.DGPOracle([layer], O.Gaussian(s2)).build_likelihood(NP, X, Y, [np.zeros((1, N, 1))])
    Kn = kern.K(NP, X) + s2 * np.eye(N)
    _, ld = np.linalg.slogdet(Kn)
    lml = float((-0.5 * Y.T @ np.linalg.solve(Kn, Y)).item() - 0.5 * ld - 0.5 * N * math.log(2 * math.pi))
    assert_allclose(elbo, lml, rtol=1e-5)                              # reference bar tests/test_collapsed.py:52-54


# T6 — MC estimator unbiased vs Gauss–Hermite over the inner layer (tests/test_dgp.py:120-174)
def test_T6_mc_matches_quadrature():
    rng = np.random.RandomState(0)
    N = 2
    X = rng.uniform(size=(N, 1))
    Y = np.sin(20 * X) + rng.randn(N, 1) * 0.001
    specs = [dict(kind="rbf", input_dim=1, variance=1.0, lengthscales=0.3, ARD=False, white_variance=None)] * 2
    lds = O.init_layers_linear(X, Y, X, specs)
    for ld in lds:
        ld["q_mu"] = 0.3 * rng.randn(*ld["q_mu"].shape)
        ld["q_sqrt"] = ld["q_sqrt"] * 0.5
    mk = lambda S: O.DGPOracle([O.SVGPLayer(l["kern"], l["Z"], l["q_mu"], l["q_sqrt"], l["mean"]) for l in lds],
                               O.Gaussian(0.01), num_samples=S)
    H = 200
    gx, gw = np.polynomial.hermite.hermgauss(H)
    # DGP_Quad mechanism (dgp.py:137-166): deterministic zs of shape (S,1,D) injected into propagate.
    # The inner-layer noise is *per data point* (z has shape S,N,D), so with N=2 the quadrature is a 2-D
    # tensor grid over the two points' independent z — which is what D_quad enumerates for D=1 only when the
    # noise is shared; the reference shares z across N (shape S,1,D).  Re-state exactly that:
    m = mk(H)
    zs = [(gx * 2 ** 0.5)[:, None, None], np.zeros((1, 1, 1))]
    _, Fm, Fv = m.propagate(NP, X, zs, S=H)
    ve = m.likelihood.variational_expectations(NP, Fm[-1], Fv[-1], Y)
    quad = np.sum(ve * (gw / math.sqrt(math.pi))[:, None, None], 0).sum()
    # MC with z shared across N exactly like the quadrature (so both estimate the same integral)
    S, reps = 200, 300
    vals = []
    for r in range(reps):
        z0 = np.random.RandomState(100 + r).randn(S, 1, 1)
        _, Fm, Fv = mk(S).propagate(NP, X, [z0, np.zeros((1, 1, 1))], S=S)
        vals.append(np.mean(m.likelihood.variational_expectations(NP, Fm[-1], Fv[-1], Y), 0).sum())
    mean, se = np.mean(vals), np.std(vals) / math.sqrt(reps)
    assert abs(quad - mean) < 4 * se + 1e-9


# T7 — reparameterize known answers (tests/test_utils.py:181-206, commented out upstream)
def test_T7_reparameterize():
    rng = np.random.RandomState(4)
    S, N, D = 3, 4, 2
    mean, var, z = rng.randn(S, N, D), rng.rand(S, N, D), rng.randn(S, N, D)
    assert_allclose(O.reparameterize(NP, mean, var, z), mean + z * (var + 1e-6) ** 0.5, rtol=1e-15)
    A = rng.randn(S, D, N, N)
    cov = A @ A.transpose(0, 1, 3, 2)                                   # S,D,N,N
    var_full = cov.transpose(0, 2, 3, 1)                                # S,N,N,D
    f = O.reparameterize(NP, mean, var_full, z, full_cov=True)
    for s in range(S):
        for d in range(D):
            L = np.linalg.cholesky(cov[s, d] + 1e-6 * np.eye(N))
            assert_allclose(f[s, :, d], mean[s, :, d] + L @ z[s, :, d], rtol=1e-12)
    assert O.reparameterize(NP, mean, None, z) is mean


# T8 — L=2 with inner kernel variance 1e-24 + Identity mean ≡ L=1 (tests/test_dgp.py:79-88)
@pytest.mark.parametrize("white", [True, False])
def test_T8_two_layer_identity_inner(white):
    X, Xs, q_mu, q_sqrt, Y = _setup(5)
    jitter = 1e-18
    k_out = dict(kind="matern52", input_dim=2, variance=1.0, lengthscales=0.5, ARD=False, white_variance=None)
    k_in = dict(k_out, variance=1e-24)
    q_sqrt = 1e-3 * np.eye(19)[None] * np.ones((3, 1, 1))
    lds2 = O.init_layers_linear(X, Y, X, [k_in, k_out], white=white, jitter=jitter)
    lds1 = O.init_layers_linear(X, Y, X, [k_out], white=white, jitter=jitter)
    for lds in (lds1, lds2):
        lds[-1]["q_mu"], lds[-1]["q_sqrt"] = q_mu, q_sqrt
    mk = lambda lds: O.DGPOracle([O.SVGPLayer(l["kern"], l["Z"], l["q_mu"], l["q_sqrt"], l["mean"], white=white,
                                             jitter=jitter) for l in lds], O.Gaussian(0.01), num_samples=2)
    rng = np.random.RandomState(6)
    z2 = [rng.randn(2, 19, 2), rng.randn(2, 19, 3)]
    L2 = mk(lds2).build_likelihood(NP, X, Y, z2)
    L1 = mk(lds1).build_likelihood(NP, X, Y, z2[1:])
    assert_allclose(L1, L2, rtol=1e-6, atol=1e-6)                       # reference bar test_dgp.py:104-106


# step-up initialisation smoke (tests/test_dgp.py:176-183)
def test_step_up_init():
    X = np.zeros((1, 1))
    specs = [dict(kind="rbf", input_dim=1, variance=1.0, lengthscales=1.0, ARD=False, white_variance=None),
             dict(kind="rbf", input_dim=2, variance=1.0, lengthscales=1.0, ARD=False, white_variance=None)]
    lds = O.init_layers_linear(X, X, X, specs)
    assert lds[0]["mean"].kind == "linear" and lds[0]["mean"].A.shape == (1, 2)
    m = O.DGPOracle([O.SVGPLayer(l["kern"], l["Z"], l["q_mu"], l["q_sqrt"], l["mean"]) for l in lds], O.Gaussian(1.0))
    v = m.build_likelihood(NP, X, X, [np.zeros((1, 1, 2)), np.zeros((1, 1, 1))])
    assert np.isfinite(v)


# torch backend ≡ numpy backend, and autograd ≡ central finite differences
def test_torch_backend_and_gradients():
    rng = np.random.RandomState(7)
    X, Y = rng.randn(30, 3), rng.randn(30, 1)
    Z = X[:10].copy()
    specs = [dict(kind="rbf", input_dim=3, variance=1.2, lengthscales=0.9, ARD=False, white_variance=None)] * 2
    lds = O.init_layers_linear(X, Y, Z, specs)
    for l in lds:
        l["q_mu"] = 0.1 * rng.randn(*l["q_mu"].shape)
    sl, state = OM.state_from_layers(lds, lik_variance=0.3)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl)
    zs = [rng.randn(4, 30, 3), rng.randn(4, 30, 1)]
    v_np = OM.elbo(spec, state, X, Y, zs, 4, num_data=100)
    v_th, g = OM.elbo_and_grad(spec, state, X, Y, zs, 4, num_data=100)
    assert_allclose(v_np, v_th, rtol=1e-12)
    for key, idx in [("l0.Z", (2, 1)), ("l1.q_mu", (3, 0)), ("l0.q_sqrt", (1, 4, 2)), ("l1.kern_lengthscales_raw", ()),
                     ("l0.kern_variance_raw", ()), ("lik_variance_raw", ())]:
        h = 1e-6
        sp, sm = {k: v.copy() for k, v in state.items()}, {k: v.copy() for k, v in state.items()}
        sp[key][idx] += h
        sm[key][idx] -= h
        fd = (OM.elbo(spec, sp, X, Y, zs, 4, 100) - OM.elbo(spec, sm, X, Y, zs, 4, 100)) / (2 * h)
        assert_allclose(g[key][idx], fd, rtol=2e-5, atol=1e-6)
    assert np.allclose(np.triu(g["l0.q_sqrt"], 1), 0.0)


def test_adam_known_answer():
    th, m, v = np.array([1.0, -2.0]), np.zeros(2), np.zeros(2)
    g = np.array([0.5, -0.25])
    O.adam_step(th, g, m, v, 1, lr=0.01)
    # first Adam step moves each coordinate by lr*sign(g) (up to eps)
    assert_allclose(th, [1.0 - 0.01, -2.0 + 0.01], atol=1e-7)


# T4 — one natural-gradient step with gamma=1 on the last layer of a Gaussian-likelihood model lands on the collapsed
# (SGPR) bound (tests/test_collapsed.py:57-104; bound formula layers.py:371-402 re-stated in dense textbook form)
@pytest.mark.parametrize("white", [False, True])
def test_T4_natgrad_gamma1_gives_collapsed_bound(white):
    rng = np.random.RandomState(8)
    N, M, s2 = 40, 9, 0.2
    X = rng.uniform(size=(N, 1)) * 4
    Y = np.sin(2 * X) + 0.3 * rng.randn(N, 1)
    Z = np.linspace(0, 4, M)[:, None]
    specs = [dict(kind="rbf", input_dim=1, variance=1.3, lengthscales=0.7, ARD=False, white_variance=None)]
    lds = O.init_layers_linear(X, Y, Z, specs, white=white)
    lds[0]["q_mu"] = 0.3 * rng.randn(M, 1)
    sl, state = OM.state_from_layers(lds, lik_variance=s2)
    spec = dict(jitter=1e-6, white=white, likelihood="gaussian", layers=sl)
    zs = [np.zeros((1, N, 1))]
    _, g = OM.elbo_and_grad(spec, state, X, Y, zs, 1)
    mu, sq = O.natgrad_step(state["l0.q_mu"], state["l0.q_sqrt"], -g["l0.q_mu"], -g["l0.q_sqrt"], 1.0)
    state2 = dict(state)
    state2["l0.q_mu"], state2["l0.q_sqrt"] = mu, sq
    elbo = OM.elbo(spec, state2, X, Y, zs, 1)
    kern = O.Kern("rbf", 1, variance=1.3, lengthscales=0.7)
    Kuu = kern.K(NP, Z) + 1e-6 * np.eye(M)
    Kuf = kern.K(NP, Z, X)
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    C = Qff + s2 * np.eye(N)
    _, ld = np.linalg.slogdet(C)
    bound = (-0.5 * Y.T @ np.linalg.solve(C, Y)).item() - 0.5 * ld - 0.5 * N * math.log(2 * math.pi) \
        - 0.5 / s2 * (kern.Kdiag(NP, X).sum() - np.trace(Qff))
    assert_allclose(elbo, bound, rtol=1e-7)


def test_input_propagation_restatement():
    # layers.py:105-117 + layer_initializations.py:55-79: the propagated inputs are copied in front of samples/means with zero
    # variance, widths chain as kern.input_dim, and the autograd gradient agrees with central differences
    rng = np.random.RandomState(9)
    N, D, M, S = 6, 2, 5, 2
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.01
    specs = [dict(kind="rbf", input_dim=2, variance=1.3, lengthscales=0.9, ARD=False, white_variance=None),
             dict(kind="rbf", input_dim=4, variance=0.8, lengthscales=1.1, ARD=False, white_variance=None)]
    pads = [rng.randn(M, 0), rng.randn(M, 2)]
    lds = O.init_layers_input_prop(X, Y, Z, specs, pads)
    assert lds[0]["Z"].shape == (M, 2) and lds[1]["Z"].shape == (M, 4) and lds[0]["q_mu"].shape == (M, 2)
    assert_allclose(lds[1]["Z"][:, 2:], pads[1] * 2.0 * 1.3 ** 0.5, rtol=1e-15)
    for l in lds:
        l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
        l["q_sqrt"] = l["q_sqrt"] * 0.7
    sl, state = OM.state_from_layers(lds, lik_variance=0.2)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl, num_classes=None)
    zs = [rng.randn(S, N, 2), rng.randn(S, N, 1)]
    Fs, Fm, Fv = OM.propagate(spec, state, X, zs, S)
    assert Fs[0].shape == (S, N, 4)
    assert np.all(Fs[0][:, :, :2] == X[None]) and np.all(Fm[0][:, :, :2] == X[None]) and np.all(Fv[0][:, :, :2] == 0)
    _, _, Fvf = OM.propagate(spec, state, X, zs, S, full_cov=True)
    assert Fvf[0].shape == (S, N, N, 4) and np.all(Fvf[0][..., :2] == 0)
    v, g = OM.elbo_and_grad(spec, state, X, Y, zs, S)
    for k, idx in (("l0.q_mu", (1, 0)), ("l0.Z", (2, 1)), ("l1.Z", (0, 3))):
        e = np.zeros_like(state[k]); e[idx] = 1e-6
        sp, sm = dict(state), dict(state)
        sp[k], sm[k] = state[k] + e, state[k] - e
        fd = (OM.elbo(spec, sp, X, Y, zs, S) - OM.elbo(spec, sm, X, Y, zs, S)) / 2e-6
        assert abs(fd - g[k][idx]) <= 1e-6 * max(1.0, abs(fd)), (k, fd, g[k][idx])


# ---- independent pins of the [UPSTREAM] formulas the oracle restates (GPflow itself cannot be imported here) ----------------
def test_T9_kernels_match_scikit_learn():
    # RBF / Matern-5/2 Gram matrices against scikit-learn's independent implementation of the same published formulas
    # (isotropic and ARD lengthscales); White: sigma^2 I on K(X), zero on K(X, X2), sigma^2 on Kdiag
    from sklearn.gaussian_process import kernels as SK
    rng = np.random.RandomState(3)
    X, X2 = rng.randn(17, 4), rng.randn(9, 4)
    for ls in (0.7, np.array([0.5, 1.0, 1.5, 2.0])):
        ard = not np.isscalar(ls)
        for kind, sk in (("rbf", SK.RBF(length_scale=ls)), ("matern52", SK.Matern(length_scale=ls, nu=2.5))):
            k = O.Kern(kind, 4, variance=1.7, lengthscales=ls, ARD=ard)
            assert_allclose(k.K(NP, X, X2), 1.7 * sk(X, X2), rtol=1e-9, atol=1e-12)
            off = ~np.eye(17, dtype=bool)      # (Matern: sqrt(r2 + 1e-12) regularisation on the diagonal only)
            assert_allclose(k.K(NP, X)[off], (1.7 * sk(X))[off], rtol=1e-9, atol=1e-12)
            assert_allclose(k.Kdiag(NP, X), np.full(17, 1.7), rtol=1e-6)
    kw = O.Kern("rbf", 4, variance=1.0, lengthscales=1.0, white_variance=0.3)
    assert_allclose(kw.K(NP, X) - O.Kern("rbf", 4).K(NP, X), 0.3 * np.eye(17), atol=1e-14)
    assert_allclose(kw.K(NP, X, X2), O.Kern("rbf", 4).K(NP, X, X2), atol=1e-15)
    assert_allclose(kw.Kdiag(NP, X), np.full(17, 1.3), rtol=1e-15)


def test_T10_gaussian_variational_expectation_by_quadrature():
    # E_{N(f; mu, v)} log N(y; f, s2) against adaptive quadrature (scipy), and the predictive density against its definition
    from scipy import integrate, stats
    lik = O.Gaussian(0.37)
    rng = np.random.RandomState(4)
    for _ in range(5):
        mu, v, y = rng.randn(), rng.rand() + 0.05, rng.randn()
        ve = float(lik.variational_expectations(NP, np.array([[[mu]]]), np.array([[[v]]]), np.array([[y]]))[0, 0, 0])
        ref, _ = integrate.quad(lambda f: stats.norm.pdf(f, mu, math.sqrt(v)) * stats.norm.logpdf(y, f, math.sqrt(0.37)),
                                mu - 12 * math.sqrt(v), mu + 12 * math.sqrt(v))
        assert_allclose(ve, ref, rtol=1e-9)
        pd = float(lik.predict_density(NP, np.array([[[mu]]]), np.array([[[v]]]), np.array([[y]]))[0, 0, 0])
        assert_allclose(pd, stats.norm.logpdf(y, mu, math.sqrt(v + 0.37)), rtol=1e-12)


def test_T11_robustmax_probability_by_monte_carlo():
    # [UPSTREAM] RobustMax.prob_is_largest (20-point Gauss-Hermite x erf products): P(f_y = max_k f_k) for independent
    # Gaussians, against a 2e6-sample Monte-Carlo estimate and the exact two-class closed form
    rng = np.random.RandomState(5)
    lik = O.MultiClass(4)
    mu, var = rng.randn(3, 4), rng.rand(3, 4) + 0.2
    Y = np.array([[0.0], [2.0], [3.0]])
    p = lik._prob_is_largest(NP, Y, mu, var)
    f = mu[:, None, :] + np.sqrt(var)[:, None, :] * rng.randn(3, 2_000_000, 4)
    mc = np.array([(np.argmax(f[i], 1) == int(Y[i, 0])).mean() for i in range(3)])
    assert np.all(np.abs(p - mc) < 5 * np.sqrt(mc * (1 - mc) / 2e6) + 3e-4)      # 1e-4 cdf clipping of the upstream formula
    from scipy import stats
    lik2 = O.MultiClass(2)
    m2, v2 = np.array([[0.3, -0.4]]), np.array([[0.5, 0.8]])
    exact = stats.norm.cdf((0.3 + 0.4) / math.sqrt(0.5 + 0.8))
    assert abs(float(lik2._prob_is_largest(NP, np.array([[0.0]]), m2, v2)[0]) - exact) < 3e-4


def test_trainable_linear_mean_restatement():
    # [UPSTREAM] mean_functions.Linear(A, b) as free parameters of the last layer: autograd of the oracle vs central
    # differences, and the bias shifts the predictive mean by exactly b
    rng = np.random.RandomState(12)
    N, D, M, S = 7, 2, 5, 2
    X, Y = rng.randn(N, D), rng.randn(N, 2)
    specs = [dict(kind="rbf", input_dim=D, variance=1.0, lengthscales=1.0, ARD=False, white_variance=None)] * 2
    lds = O.init_layers_linear(X, Y, X[:M] + 0.01, specs)
    for l in lds:
        l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
    A0, b0 = 0.4 * rng.randn(D, 2), np.array([0.7, -0.2])
    lds[-1]["mean"] = O.MeanFn("linear", A=A0, b=b0)
    lds[-1]["mean_trainable"] = True
    sl, state = OM.state_from_layers(lds, lik_variance=0.3)
    assert state["l1.mean_A"].shape == (D, 2) and state["l1.mean_b"].shape == (2,)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl, num_classes=None)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 2)]
    _, Fm, _ = OM.propagate(spec, state, X, zs, S)
    st0 = dict(state); st0["l1.mean_b"] = np.zeros(2)
    _, Fm0, _ = OM.propagate(spec, st0, X, zs, S)
    assert_allclose(Fm[-1] - Fm0[-1], np.broadcast_to(b0, Fm[-1].shape), rtol=1e-12, atol=1e-12)
    v, g = OM.elbo_and_grad(spec, state, X, Y, zs, S)
    for k, idx in (("l1.mean_A", (1, 0)), ("l1.mean_b", (1,)), ("l0.q_mu", (2, 1))):
        e = np.zeros_like(state[k]); e[idx] = 1e-6
        sp, sm = dict(state), dict(state)
        sp[k], sm[k] = state[k] + e, state[k] - e
        fd = (OM.elbo(spec, sp, X, Y, zs, S) - OM.elbo(spec, sm, X, Y, zs, S)) / 2e-6
        assert abs(fd - g[k][idx]) <= 1e-6 * max(1.0, abs(fd)), (k, fd, g[k][idx])


def test_T12_bernoulli_probit_restatement():
    """Oracle Bernoulli (gpflow 1.1.1 Bernoulli() of /root/reference/tests/test_dgp.py:48-54): (a) the 20-point Gauss-Hermite
    variational expectation written the way upstream does — (R x H) log-densities times the (H x 1) weights — equals the oracle's
    loop; (b) it agrees with adaptive quadrature of log p(y|f) N(f|mu, var) to the rule's accuracy; (c) predict_mean_and_var is
    the probit closed form, whose mean agrees with adaptive quadrature of probit(f) N(f|mu, var)."""
    from scipy import integrate, special
    rng = np.random.RandomState(12)
    S, N, D = 2, 7, 2
    mu, var = rng.randn(S, N, D), rng.uniform(0.05, 1.5, size=(S, N, D))
    Y = rng.choice([-1.0, 1.0], N * D).reshape(N, D)
    lik = O.Bernoulli()
    ve = lik.variational_expectations(O.NP, mu, var, Y)
    assert ve.shape == (S, N, D)
    # (a) upstream's matmul form
    gx, gw = np.polynomial.hermite.hermgauss(20)
    Xq = gx[None, :] * np.sqrt(2.0 * var.reshape(-1, 1)) + mu.reshape(-1, 1)
    p = 0.5 * (1.0 + special.erf(Xq / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3
    Yt = np.tile(np.broadcast_to(Y, (S, N, D)).reshape(-1, 1), (1, 20))
    logp = np.log(np.where(Yt == 1, p, 1 - p))
    assert_allclose(ve.reshape(-1, 1), logp @ (gw.reshape(-1, 1) / np.sqrt(np.pi)), rtol=1e-13)

    # (b), (c) adaptive quadrature on a few entries
    def probit(f):
        return 0.5 * (1.0 + special.erf(f / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3
    pm, pv = lik.predict_mean_and_var(O.NP, mu, var)
    assert_allclose(pv, pm - pm ** 2, rtol=1e-14)
    for idx in [(0, 0, 0), (1, 3, 1), (0, 6, 1)]:
        m0, v0, y0 = mu[idx], var[idx], Y[idx[1], idx[2]]
        pdf = lambda f: np.exp(-0.5 * (f - m0) ** 2 / v0) / np.sqrt(2 * np.pi * v0)
        q = integrate.quad(lambda f: np.log(probit(f) if y0 == 1 else 1 - probit(f)) * pdf(f), m0 - 12 * np.sqrt(v0),
                           m0 + 12 * np.sqrt(v0), epsabs=1e-12)[0]
        assert abs(ve[idx] - q) < 1e-6
        qm = integrate.quad(lambda f: probit(f) * pdf(f), m0 - 12 * np.sqrt(v0), m0 + 12 * np.sqrt(v0), epsabs=1e-12)[0]
        assert abs(pm[idx] - qm) < 1e-9        # E[Phi(f)] = Phi(mu / sqrt(1 + var)) exactly; the 1e-3 mixing is linear
    # predict_density: log-density at the predictive mean, targets other than 1 select 1 - p
    pd = lik.predict_density(O.NP, mu, var, Y)
    assert_allclose(pd, np.log(np.where(np.broadcast_to(Y, pm.shape) == 1, pm, 1 - pm)), rtol=1e-14)


def test_bernoulli_single_layer_elbo_and_torch_gradient():
    """One-layer DGP with the Bernoulli likelihood (the L = 1 case of tests/test_dgp.py:48-54): numpy and torch backends agree
    and the torch gradient of the ELBO w.r.t. q_mu matches a central difference."""
    import torch
    from tests.helpers import kern_spec
    rng = np.random.RandomState(5)
    N, D, M, S = 20, 2, 7, 3
    X = rng.uniform(size=(N, D))
    Y = rng.choice([-1.0, 1.0], N).reshape(N, 1)
    lds = O.init_layers_linear(X, Y, X[:M].copy(), [kern_spec("rbf", D, 1.0, 0.7)], white=True)
    lds[0]["q_mu"] = 0.3 * rng.randn(*lds[0]["q_mu"].shape)
    sl, state = OM.state_from_layers(lds, likelihood="bernoulli")
    spec = dict(jitter=1e-6, white=True, likelihood="bernoulli", layers=sl, num_classes=None)
    zs = [rng.randn(S, N, 1)]
    e = OM.elbo(spec, state, X, Y, zs, S, num_data=50)
    assert np.isfinite(e)
    st = {k: torch.tensor(v, dtype=torch.float64, requires_grad=(k == "l0.q_mu")) for k, v in state.items()}
    et = OM.build(O.TH, spec, st, S, 50).build_likelihood(O.TH, torch.tensor(X), torch.tensor(Y), [torch.tensor(z) for z in zs])
    assert abs(float(et) - e) < 1e-10 * max(1.0, abs(e))
    et.backward()
    g = st["l0.q_mu"].grad.numpy()
    h = 1e-6
    for i in (0, 3):
        sp, sm = dict(state), dict(state)
        sp["l0.q_mu"] = state["l0.q_mu"].copy(); sp["l0.q_mu"][i, 0] += h
        sm["l0.q_mu"] = state["l0.q_mu"].copy(); sm["l0.q_mu"][i, 0] -= h
        fd = (OM.elbo(spec, sp, X, Y, zs, S, num_data=50) - OM.elbo(spec, sm, X, Y, zs, S, num_data=50)) / (2 * h)
        assert abs(fd - g[i, 0]) < 1e-6 * max(1.0, abs(fd))


# T13 — the ONE number the reference tree itself holds for code on this path: demos/demo_step_function.ipynb (cells 2-3) stores
#   "Objective function value: -59.181069"
# for GPR(X, Y, RBF(1, lengthscales=0.2)) + ScipyOptimizer().minimize on np.random.seed(0) data (X ~ U(0,1)^50, then Z ~ U(0,1)^25
# drawn from the same stream, then Y = step(X) + 0.01 randn) — the minimum of the negative log marginal likelihood over GPflow's
# unconstrained (softplus + 1e-6) variables, started from variance 1, lengthscale 0.2, noise variance 1.  Only the number and this
# recipe are kept here, no notebook text.
#
# What it pins: the oracle's RBF kernel ([UPSTREAM] square_dist + exp form), the softplus transform, the Gaussian
# log-marginal-likelihood / Cholesky composition and their torch gradients — L-BFGS from the same start must reach the same
# minimum value — and, through identity T5 at THOSE hyper-parameters, the oracle's one-layer ELBO code (conditional_ND, KL,
# variational expectations) at the reference's own tolerance.  What it does not pin: anything specific to more than one layer
# (propagate, reparameterize), the minibatch scaling, Matern52 / White / ARD, MultiClass / Bernoulli.
REFERENCE_GPR_OBJECTIVE = -59.181069


def _step_function_data():
    st = np.random.RandomState(0)                     # np.random.seed(0) of the notebook: the same legacy stream
    N, M = 50, 25
    X = st.uniform(0, 1, N)[:, None]
    st.uniform(0, 1, M)                               # Z of the notebook: drawn between X and the noise of Y
    Y = np.reshape([0.0 if x < 0.5 else 1.0 for x in X], X.shape) + st.randn(*X.shape) * 1e-2
    return X, Y


def test_T13_reference_held_gpr_objective():
    import torch
    from scipy.optimize import minimize
    X, Y = _step_function_data()
    N = X.shape[0]
    Xt, Yt = torch.tensor(X), torch.tensor(Y)

    def objective(raw):
        r = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
        var, ls, noise = (O.positive_forward(O.TH, r[i]) for i in range(3))
        K = O.Kern("rbf", 1, variance=var, lengthscales=ls).K(O.TH, Xt) + noise * torch.eye(N, dtype=torch.float64)
        L = torch.linalg.cholesky(K)
        a = torch.linalg.solve_triangular(L, Yt, upper=False)
        f = 0.5 * (a * a).sum() + torch.log(torch.diagonal(L)).sum() + 0.5 * N * math.log(2 * math.pi)
        f.backward()
        return float(f.detach()), r.grad.numpy().copy()

    x0 = O.positive_backward_np(np.array([1.0, 0.2, 1.0]))
    res = minimize(objective, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=1000, ftol=1e-14, gtol=1e-10))
    assert abs(res.fun - REFERENCE_GPR_OBJECTIVE) < 5e-7, res.fun            # all eight printed digits
    # ... and the oracle's ELBO machinery at that optimum: Z = X with the optimal q(u) gives the same number (T5)
    var, ls, s2 = (float(O.positive_forward(NP, np.array(v))) for v in res.x)
    kern = O.Kern("rbf", 1, variance=var, lengthscales=ls)
    jitter = 1e-10
    Kuu = kern.K(NP, X) + jitter * np.eye(N)
    Sig = Kuu - Kuu @ np.linalg.solve(Kuu + s2 * np.eye(N), Kuu)
    Sig = 0.5 * (Sig + Sig.T) + 1e-12 * np.eye(N)
    m_opt = Kuu @ np.linalg.solve(Kuu + s2 * np.eye(N), Y)
    layer = O.SVGPLayer(kern, X, m_opt, np.linalg.cholesky(Sig)[None], O.MeanFn("zero"), white=False, jitter=jitter)
    elbo = O.DGPOracle([layer], O.Gaussian(s2)).build_likelihood(NP, X, Y, [np.zeros((1, N, 1))])
    assert_allclose(-elbo, REFERENCE_GPR_OBJECTIVE, rtol=1e-5)               # reference bar tests/test_collapsed.py:52-54


# ---------------------------------------------------------------- Poisson / Exponential / StudentT ([UPSTREAM] gpflow 1.1.1 likelihoods.py)
@pytest.mark.parametrize("name", ["poisson", "exponential", "student_t", "gamma", "beta"])
def test_T14_further_likelihood_restatements(name):
    """(a) log densities against scipy.stats; (b) the exp-link closed forms of the variational expectations against the base class's
    20-point Gauss-Hermite rule on the same log density and against adaptive quadrature; (c) predict_density / predict_mean_and_var
    against adaptive quadrature of exp(logp) and of the conditional moments; (d) numpy and torch backends agree."""
    import math
    import torch
    from scipy import integrate, stats
    lik = {"poisson": O.Poisson(binsize=1.7), "exponential": O.Exponential(), "student_t": O.StudentT(0.6, 4.0), "gamma": O.Gamma(2.3),
           "beta": O.Beta(3.1)}[name]
    rng = np.random.RandomState(7)
    for _ in range(4):
        mu, v = float(rng.randn() * 0.7), float(0.05 + rng.rand() * 0.6)
        y = {"poisson": float(rng.randint(0, 6)), "exponential": float(rng.rand() * 3 + 0.1), "student_t": float(rng.randn()),
             "gamma": float(rng.rand() * 3 + 0.1), "beta": float(rng.uniform(0.05, 0.95))}[name]
        Fmu, Fvar, Y = np.array([[[mu]]]), np.array([[[v]]]), np.array([[y]])
        f0 = float(rng.randn())
        lp = float(np.ravel(lik.logp(O.NP, np.array(f0), np.array(y)))[0])
        ref = {"poisson": lambda: stats.poisson.logpmf(y, 1.7 * math.exp(f0)), "exponential": lambda: stats.expon.logpdf(y, scale=math.exp(f0)),
               "student_t": lambda: stats.t.logpdf(y, 4.0, loc=f0, scale=0.6), "gamma": lambda: stats.gamma.logpdf(y, 2.3, scale=math.exp(f0)),
               "beta": lambda: stats.beta.logpdf(y, 3.1 * float(O.Bernoulli._probit(O.NP, np.array(f0))),
                                                 3.1 * (1.0 - float(O.Bernoulli._probit(O.NP, np.array(f0)))))}[name]()
        assert abs(lp - ref) < 1e-12 * max(1.0, abs(ref))
        dens = lambda x: stats.norm.pdf(x, mu, math.sqrt(v))
        lpf = lambda x: float(np.ravel(lik.logp(O.NP, np.array(x), np.array(y)))[0])
        ve = float(lik.variational_expectations(O.NP, Fmu, Fvar, Y)[0, 0, 0])
        ve_gh = float(O._QuadratureLikelihood.variational_expectations(lik, O.NP, Fmu, Fvar, Y)[0, 0, 0])
        ve_q = integrate.quad(lambda x: lpf(x) * dens(x), mu - 12 * math.sqrt(v), mu + 12 * math.sqrt(v), epsabs=1e-13, epsrel=1e-13)[0]
        assert abs(ve - ve_q) < 2e-6 * max(1.0, abs(ve_q)) and abs(ve - ve_gh) < 2e-6 * max(1.0, abs(ve))
        pd = float(lik.predict_density(O.NP, Fmu, Fvar, Y)[0, 0, 0])
        pd_q = math.log(integrate.quad(lambda x: math.exp(lpf(x)) * dens(x), mu - 12 * math.sqrt(v), mu + 12 * math.sqrt(v), epsabs=1e-14,
                                       epsrel=1e-13)[0])
        assert abs(pd - pd_q) < 1e-4 * max(1.0, abs(pd_q))
        m, s2 = lik.predict_mean_and_var(O.NP, Fmu, Fvar)
        cm = lambda x: float(np.ravel(lik.conditional_mean(O.NP, np.array([x])))[0])
        cv = lambda x: float(np.ravel(lik.conditional_variance(O.NP, np.array([x])))[0])
        em = integrate.quad(lambda x: cm(x) * dens(x), mu - 12 * math.sqrt(v), mu + 12 * math.sqrt(v), epsabs=1e-13, epsrel=1e-13)[0]
        eq = integrate.quad(lambda x: (cv(x) + cm(x) ** 2) * dens(x), mu - 12 * math.sqrt(v), mu + 12 * math.sqrt(v), epsabs=1e-13, epsrel=1e-13)[0]
        assert abs(float(m.ravel()[0]) - em) < 1e-6 * max(1.0, abs(em)) and abs(float(s2.ravel()[0]) - (eq - em ** 2)) < 1e-5 * max(1.0, eq)
        lt = {"poisson": O.Poisson(binsize=1.7), "exponential": O.Exponential(), "student_t": O.StudentT(torch.tensor(0.6, dtype=torch.float64), 4.0),
              "gamma": O.Gamma(torch.tensor(2.3, dtype=torch.float64)), "beta": O.Beta(torch.tensor(3.1, dtype=torch.float64))}[name]
        a = (torch.tensor(Fmu), torch.tensor(Fvar), torch.tensor(Y))
        assert abs(float(lt.variational_expectations(O.TH, *a)) - ve) < 1e-13 * max(1.0, abs(ve))
        assert abs(float(lt.predict_density(O.TH, *a)) - pd) < 1e-13 * max(1.0, abs(pd))


@pytest.mark.parametrize("name", ["poisson", "exponential", "student_t", "gamma", "beta"])
def test_further_likelihoods_single_layer_elbo_and_torch_gradient(name):
    """One-layer DGP with each further likelihood: numpy and torch backends agree on the ELBO and the torch gradient matches central
    differences for q_mu, Z and (StudentT) the raw scale."""
    rng = np.random.RandomState(11)
    N, D, M, S = 12, 2, 6, 3
    X = rng.randn(N, D)
    Y = {"poisson": rng.poisson(2.0, (N, 1)).astype(float), "exponential": rng.exponential(1.5, (N, 1)), "student_t": rng.randn(N, 1),
         "gamma": rng.exponential(1.5, (N, 1)), "beta": rng.uniform(0.05, 0.95, (N, 1))}[name]
    lds = O.init_layers_linear(X, Y, X[:M].copy(), [dict(kind="rbf", input_dim=D, variance=1.2, lengthscales=0.9, ARD=False, white_variance=None)],
                               white=False, jitter=1e-6)
    lds[0]["q_mu"] = 0.3 * rng.randn(M, 1)
    sl, state = OM.state_from_layers(lds, lik_variance=0.8, likelihood=name)
    spec = dict(jitter=1e-6, white=False, likelihood=name, layers=sl, num_classes=None, lik_aux={"poisson": 1.5, "student_t": 5.0}.get(name))
    zs = [rng.randn(S, N, 1)]
    e = OM.elbo(spec, state, X, Y, zs, S, num_data=40)
    et, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=40)
    assert abs(et - e) < 1e-10 * max(1.0, abs(e))
    keys = ["l0.q_mu", "l0.Z"] + (["lik_variance_raw"] if name in ("student_t", "gamma", "beta") else [])
    assert (name in ("student_t", "gamma", "beta")) == ("lik_variance_raw" in state)
    for k in keys:
        st = {kk: np.array(vv, dtype=np.float64) for kk, vv in state.items()}
        idx = tuple(0 for _ in range(st[k].ndim))
        h = 1e-6
        st[k][idx] += h
        ep = OM.elbo(spec, st, X, Y, zs, S, num_data=40)
        st[k][idx] -= 2 * h
        em = OM.elbo(spec, st, X, Y, zs, S, num_data=40)
        fd = (ep - em) / (2 * h)
        assert abs(fd - np.asarray(g[k])[idx]) < 1e-5 * max(1.0, abs(fd)), (k, fd, np.asarray(g[k])[idx])
