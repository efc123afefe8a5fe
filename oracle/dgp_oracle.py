"""CPU restatement of the doubly-stochastic DGP hot path (TEST INFRASTRUCTURE — see oracle/__init__.py).

PARITY UNPINNED by numeric vectors (reference needs gpflow==1.1.1 + tensorflow==1.8, neither importable
here); pinned relationally by tests/test_oracle_identities.py (the identities the reference's own tests assert,
T1-T8) and, for the restated [UPSTREAM] formulas, against independent implementations in the image
(scikit-learn kernels, scipy quadrature, Monte Carlo: T9-T11).

Every function follows the reference *op for op* ("reference form": two triangular solves, SK, B = SK·A,
sum(A∘B)) and cites the reference file:line (paths under /root/reference/).  [UPSTREAM] marks GPflow 1.1.1 /
TF 1.8 behaviour that is not in the reference tree and is re-stated from the published formulas.

The same code runs under two array backends:
  * ``NP``  — numpy/scipy float64: the value oracle;
  * ``TH``  — torch CPU float64: identical op sequence, used only to obtain reverse-mode gradients
              (the analogue of ``tf.gradients``) for the backward-pass parity tests and as the timed
              CPU baseline of a full training step (forward + gradient + Adam).
"""
import math
import numpy as np
import scipy.linalg as _sla

DEFAULT_JITTER = 1e-6          # [UPSTREAM] gpflow settings.numerics.jitter_level default
SOFTPLUS_LOWER = 1e-6          # [UPSTREAM] gpflow.transforms.positive = Log1pe(lower=1e-6)


# --------------------------------------------------------------------------------------------------
# array backends
# --------------------------------------------------------------------------------------------------
class NP:
    name = "numpy"

    @staticmethod
    def asarray(x):
        return np.asarray(x, dtype=np.float64)

    @staticmethod
    def cholesky(a):
        return np.linalg.cholesky(a)

    @staticmethod
    def trsm(l, b, lower=True):
        """solve tri(l) x = b ; batched over leading dims like tf.matrix_triangular_solve [UPSTREAM]"""
        if l.ndim == 2:
            return _sla.solve_triangular(l, b, lower=lower)
        return np.stack([_sla.solve_triangular(li, bi, lower=lower) for li, bi in zip(l, b)])

    @staticmethod
    def t(a):
        return np.swapaxes(a, -1, -2)

    eye = staticmethod(lambda n: np.eye(n))
    exp = staticmethod(np.exp)
    log = staticmethod(np.log)
    sqrt = staticmethod(np.sqrt)
    sum = staticmethod(lambda a, axis=None: np.sum(a, axis=axis))
    mean = staticmethod(lambda a, axis=None: np.mean(a, axis=axis))
    zeros = staticmethod(lambda *s: np.zeros(s))
    ones = staticmethod(lambda *s: np.ones(s))
    diagonal = staticmethod(lambda a: np.diagonal(a, axis1=-2, axis2=-1))
    tile = staticmethod(lambda a, reps: np.tile(a, reps))
    reshape = staticmethod(lambda a, s: np.reshape(a, s))
    stack = staticmethod(lambda xs: np.stack(xs))
    concat = staticmethod(lambda xs, axis: np.concatenate(xs, axis))
    tril = staticmethod(np.tril)
    softplus = staticmethod(lambda x: np.logaddexp(0.0, x))
    erf = staticmethod(lambda x: __import__("scipy.special", fromlist=["erf"]).erf(x))
    clip_min = staticmethod(lambda x, lo: np.maximum(x, lo))
    logsumexp = staticmethod(lambda a, axis: __import__("scipy.special", fromlist=["logsumexp"]).logsumexp(a, axis=axis))
    prod = staticmethod(lambda a, axis: np.prod(a, axis=axis))
    where = staticmethod(np.where)


class TH:
    name = "torch"
    import torch as _t

    @staticmethod
    def asarray(x):
        t = TH._t
        return x if isinstance(x, t.Tensor) else t.as_tensor(np.asarray(x, dtype=np.float64))

    @staticmethod
    def cholesky(a):
        return TH._t.linalg.cholesky(a)

    @staticmethod
    def trsm(l, b, lower=True):
        return TH._t.linalg.solve_triangular(l, b, upper=not lower)

    @staticmethod
    def t(a):
        return a.transpose(-1, -2)

    eye = staticmethod(lambda n: TH._t.eye(n, dtype=TH._t.float64))
    exp = staticmethod(lambda x: TH._t.exp(x))
    log = staticmethod(lambda x: TH._t.log(x))
    sqrt = staticmethod(lambda x: TH._t.sqrt(x))
    sum = staticmethod(lambda a, axis=None: a.sum() if axis is None else a.sum(dim=axis))
    mean = staticmethod(lambda a, axis=None: a.mean() if axis is None else a.mean(dim=axis))
    zeros = staticmethod(lambda *s: TH._t.zeros(*s, dtype=TH._t.float64))
    ones = staticmethod(lambda *s: TH._t.ones(*s, dtype=TH._t.float64))
    diagonal = staticmethod(lambda a: TH._t.diagonal(a, dim1=-2, dim2=-1))
    tile = staticmethod(lambda a, reps: a.repeat(*reps))
    reshape = staticmethod(lambda a, s: a.reshape(*s))
    stack = staticmethod(lambda xs: TH._t.stack(list(xs)))
    concat = staticmethod(lambda xs, axis: TH._t.cat(list(xs), dim=axis))
    tril = staticmethod(lambda a: TH._t.tril(a))
    softplus = staticmethod(lambda x: TH._t.nn.functional.softplus(x, threshold=1e9))
    erf = staticmethod(lambda x: TH._t.erf(x))
    clip_min = staticmethod(lambda x, lo: TH._t.clamp(x, min=lo))
    logsumexp = staticmethod(lambda a, axis: TH._t.logsumexp(a, dim=axis))
    prod = staticmethod(lambda a, axis: TH._t.prod(a, dim=axis))
    where = staticmethod(lambda c, a, b: TH._t.where(c, a, b))


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] gpflow.transforms.positive  (softplus + 1e-6) — the unconstrained <-> constrained map of
# every variance / lengthscale on the path ("Gradient is taken w.r.t. the
# unconstrained free variables")
# --------------------------------------------------------------------------------------------------
def positive_forward(xp, raw):
    return xp.softplus(raw) + SOFTPLUS_LOWER


def positive_backward_np(y):
    y = np.asarray(y, dtype=np.float64) - SOFTPLUS_LOWER
    return y + np.log(-np.expm1(-y))


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] gpflow 1.1.1 kernels.py — Stationary / RBF / Matern52 / White / Sum
# (consumed at layers.py:161,171,184,213 through feature.Kuu/Kuf, kern.K, kern.Kdiag)
# --------------------------------------------------------------------------------------------------
class Kern:
    """kind in {'rbf','matern52'}; optional White summand (k + White) -> white_variance."""

    def __init__(self, kind, input_dim, variance=1.0, lengthscales=1.0, ARD=False, white_variance=None):
        self.kind, self.input_dim, self.ARD = kind, int(input_dim), bool(ARD)
        self.variance = variance
        self.lengthscales = lengthscales
        self.white_variance = white_variance      # None: no White summand

    def _sqdist(self, xp, X, X2):
        # [UPSTREAM] Stationary.square_dist: expand-the-square form on X/ℓ, no clamp
        ls = self.lengthscales
        Xs = X / ls
        Xss = xp.sum(Xs * Xs, 1)
        if X2 is None:
            d = -2.0 * (Xs @ xp.t(Xs))
            return d + Xss[:, None] + Xss[None, :]
        X2s = X2 / ls
        X2ss = xp.sum(X2s * X2s, 1)
        return -2.0 * (Xs @ xp.t(X2s)) + Xss[:, None] + X2ss[None, :]

    def K(self, xp, X, X2=None):
        r2 = self._sqdist(xp, X, X2)
        if self.kind == "rbf":
            k = self.variance * xp.exp(-0.5 * r2)                       # [UPSTREAM] RBF.K
        elif self.kind == "matern52":
            r = xp.sqrt(r2 + 1e-12)                                     # [UPSTREAM] euclid_dist
            s5 = math.sqrt(5.0)
            k = self.variance * (1.0 + s5 * r + 5.0 / 3.0 * (r * r)) * xp.exp(-s5 * r)
        else:
            raise ValueError(self.kind)
        if self.white_variance is not None and X2 is None:
            k = k + self.white_variance * xp.eye(X.shape[0])            # [UPSTREAM] White.K(X)
        return k                                                        # White.K(X, X2) = 0

    def Kdiag(self, xp, X):
        kd = self.variance * xp.ones(X.shape[0])                        # [UPSTREAM] Stationary.Kdiag
        if self.white_variance is not None:
            kd = kd + self.white_variance
        return kd


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] gpflow mean functions Zero / Identity / Linear  (layers.py:219)
# --------------------------------------------------------------------------------------------------
class MeanFn:
    def __init__(self, kind, A=None, b=None):
        self.kind, self.A, self.b = kind, A, b

    def __call__(self, xp, X, num_outputs):
        if self.kind == "zero":
            return xp.zeros(X.shape[0], num_outputs)                    # broadcasts like tf zeros (N,1)
        if self.kind == "identity":
            return X
        if self.kind == "linear":
            out = X @ xp.asarray(self.A)
            return out if self.b is None else out + xp.asarray(self.b)
        raise ValueError(self.kind)


# --------------------------------------------------------------------------------------------------
# layers.py:122-246  SVGP_Layer
# --------------------------------------------------------------------------------------------------
class SVGPLayer:
    def __init__(self, kern, Z, q_mu, q_sqrt, mean_function, white=False, jitter=DEFAULT_JITTER, input_prop_dim=None):
        self.input_prop_dim = input_prop_dim                             # layers.py:36-50
        self.kern, self.Z, self.q_mu, self.q_sqrt = kern, Z, q_mu, q_sqrt
        self.mean_function, self.white, self.jitter = mean_function, white, jitter
        self.num_inducing = Z.shape[0]
        self.num_outputs = q_mu.shape[1]
        self._chol = None

    # layers.py:167-175 build_cholesky_if_needed (memoised: shared by conditional_ND and KL)
    def build_cholesky(self, xp):
        if self._chol is None:
            Ku = self.kern.K(xp, self.Z) + self.jitter * xp.eye(self.num_inducing)   # layers.py:171
            Lu = xp.cholesky(Ku)                                                      # layers.py:172
            self._chol = (Ku, Lu)
        return self._chol

    # layers.py:178-219 conditional_ND, reference form
    def conditional_ND(self, xp, X, full_cov=False):
        Ku, Lu = self.build_cholesky(xp)
        D, M = self.num_outputs, self.num_inducing
        Kuf = self.kern.K(xp, self.Z, X)                                # layers.py:184
        A = xp.trsm(Lu, Kuf, lower=True)                                # layers.py:186
        if not self.white:
            A = xp.trsm(xp.t(Lu), A, lower=False)                       # layers.py:188
        mean = xp.t(A) @ self.q_mu                                      # layers.py:190
        A_tiled = xp.stack([A] * D)                                     # layers.py:192
        I = xp.eye(M)
        SK = -(xp.stack([I] * D) if self.white else xp.stack([Ku] * D))  # layers.py:195-198
        q_sqrt = xp.tril(self.q_sqrt)                                   # LowerTriangular transform, layers.py:150
        SK = SK + q_sqrt @ xp.t(q_sqrt)                                 # layers.py:201
        B = SK @ A_tiled                                                # layers.py:204
        if full_cov:
            delta = xp.t(A_tiled) @ B                                   # layers.py:208  (D,N,N)
            Kff = self.kern.K(xp, X)                                    # layers.py:209
            var = Kff[None] + delta                                     # layers.py:216
            var = var.transpose(2, 1, 0) if xp is NP else var.permute(2, 1, 0)   # tf.transpose reverses axes -> (N,N,D)
        else:
            delta = xp.sum(A_tiled * B, 1)                              # layers.py:212  (D,N)
            Kff = self.kern.Kdiag(xp, X)                                # layers.py:213
            var = xp.t(Kff[None] + delta)                               # layers.py:216-217 (N,D)
        return mean + self.mean_function(xp, X, D), var                 # layers.py:219

    # layers.py:221-246 KL
    def KL(self, xp):
        Ku, Lu = self.build_cholesky(xp)
        D, M = self.num_outputs, self.num_inducing
        q_sqrt = xp.tril(self.q_sqrt)
        KL = -0.5 * D * M                                               # layers.py:234
        KL = KL - 0.5 * xp.sum(xp.log(xp.diagonal(q_sqrt) ** 2))       # layers.py:235
        if not self.white:
            KL = KL + xp.sum(xp.log(xp.diagonal(Lu))) * D               # layers.py:238
            Lu_t = xp.stack([Lu] * D)
            KL = KL + 0.5 * xp.sum(xp.trsm(Lu_t, q_sqrt, lower=True) ** 2)   # layers.py:239
            Kinv_m = xp.trsm(xp.t(Lu), xp.trsm(Lu, self.q_mu, lower=True), lower=False)  # cholesky_solve :240
            KL = KL + 0.5 * xp.sum(self.q_mu * Kinv_m)                  # layers.py:241
        else:
            KL = KL + 0.5 * xp.sum(q_sqrt ** 2)                         # layers.py:243
            KL = KL + 0.5 * xp.sum(self.q_mu ** 2)                      # layers.py:244
        return KL

    # layers.py:52-74 conditional_SND
    def conditional_SND(self, xp, X, full_cov=False):
        S, N, Din = X.shape
        if full_cov:
            ms, vs = zip(*[self.conditional_ND(xp, X[s], full_cov=True) for s in range(S)])  # tf.map_fn :66-69
            return xp.stack(ms), xp.stack(vs)
        mean, var = self.conditional_ND(xp, xp.reshape(X, (S * N, Din)))           # layers.py:71-73
        return xp.reshape(mean, (S, N, self.num_outputs)), xp.reshape(var, (S, N, self.num_outputs))

    # layers.py:76-119 sample_from_conditional
    def sample_from_conditional(self, xp, X, z, full_cov=False):
        mean, var = self.conditional_SND(xp, X, full_cov=full_cov)
        samples = reparameterize(xp, mean, var, z, full_cov=full_cov, jitter=self.jitter)  # layers.py:103
        if self.input_prop_dim:                                          # layers.py:105-117
            X_prop = X[:, :, :self.input_prop_dim]
            samples = xp.concat([X_prop, samples], 2)
            mean = xp.concat([X_prop, mean], 2)
            if full_cov:
                S, N = X.shape[0], X.shape[1]
                var = xp.concat([xp.zeros(S, N, N, self.input_prop_dim), var], 3)
            else:
                var = xp.concat([X_prop * 0.0, var], 2)
        return samples, mean, var


# utils.py:22-51 reparameterize
def reparameterize(xp, mean, var, z, full_cov=False, jitter=DEFAULT_JITTER):
    if var is None:
        return mean                                                     # utils.py:37-38
    if not full_cov:
        return mean + z * (var + jitter) ** 0.5                         # utils.py:40-41 (no clamp)
    S, N, D = mean.shape                                                # utils.py:43-51
    perm = (lambda a, p: a.transpose(*p)) if xp is NP else (lambda a, p: a.permute(*p))
    mean_t = perm(mean, (0, 2, 1))
    var_t = perm(var, (0, 3, 1, 2))
    chol = xp.cholesky(var_t + jitter * xp.eye(N)[None, None])
    z_t = perm(z, (0, 2, 1))[..., None]
    f = mean_t + (chol @ z_t)[..., 0]
    return perm(f, (0, 2, 1))


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] likelihoods (dgp.py:89 via utils.py:54-121 BroadcastingLikelihood)
# --------------------------------------------------------------------------------------------------
class Gaussian:
    kind = "gaussian"

    def __init__(self, variance=1.0):
        self.variance = variance

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        # [UPSTREAM] Gaussian.variational_expectations ; Y broadcast as Y[None] (utils.py:72-73)
        return (-0.5 * math.log(2 * math.pi) - 0.5 * xp.log(self.variance * xp.ones(1))
                - 0.5 * ((Y - Fmu) ** 2 + Fvar) / self.variance)

    def predict_mean_and_var(self, xp, Fmu, Fvar):
        return Fmu, Fvar + self.variance

    def predict_density(self, xp, Fmu, Fvar, Y):
        v = Fvar + self.variance
        return -0.5 * math.log(2 * math.pi) - 0.5 * xp.log(v) - 0.5 * (Y - Fmu) ** 2 / v


class MultiClass:
    """[UPSTREAM] MultiClass(K) with RobustMax(eps=1e-3), 20-point Gauss–Hermite."""
    kind = "multiclass"

    def __init__(self, num_classes, epsilon=1e-3, num_gauss_hermite_points=20):
        self.K, self.eps, self.H = num_classes, epsilon, num_gauss_hermite_points

    def _prob_is_largest(self, xp, Y, mu, var):
        gh_x, gh_w = np.polynomial.hermite.hermgauss(self.H)
        gh_w = gh_w / math.sqrt(math.pi)
        Yi = np.asarray(Y if xp is NP else Y.detach().numpy()).astype(np.int64).reshape(-1)
        R = mu.shape[0]
        oh = np.zeros((R, self.K))
        oh[np.arange(R), Yi] = 1.0
        oh = xp.asarray(oh)
        mu_sel = xp.sum(oh * mu, 1)[:, None]
        var_sel = xp.sum(oh * var, 1)[:, None]
        X = mu_sel + xp.asarray(gh_x)[None, :] * xp.sqrt(xp.clip_min(2.0 * var_sel, 1e-10))      # (R,H)
        dist = (X[:, None, :] - mu[:, :, None]) / xp.sqrt(xp.clip_min(var, 1e-10))[:, :, None]   # (R,K,H)
        cdf = 0.5 * (1.0 + xp.erf(dist / math.sqrt(2.0)))
        cdf = cdf * (1 - 2e-4) + 1e-4
        cdf = cdf * (1.0 - oh[:, :, None]) + oh[:, :, None]
        return xp.prod(cdf, 1) @ xp.asarray(gh_w)                                                  # (R,)

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        # BroadcastingLikelihood flatten/tile, utils.py:76-86
        S, N, D = Fmu.shape
        Yt = xp.reshape(xp.stack([xp.asarray(Y)] * S), (S * N, -1))
        p = self._prob_is_largest(xp, Yt, xp.reshape(Fmu, (S * N, D)), xp.reshape(Fvar, (S * N, D)))
        ve = p * math.log(1 - self.eps) + (1.0 - p) * math.log(self.eps / (self.K - 1.0))
        return xp.reshape(ve, (S, N, 1))

    def predict_density(self, xp, Fmu, Fvar, Y):
        S, N, D = Fmu.shape
        Yt = xp.reshape(xp.stack([xp.asarray(Y)] * S), (S * N, -1))
        p = self._prob_is_largest(xp, Yt, xp.reshape(Fmu, (S * N, D)), xp.reshape(Fvar, (S * N, D)))
        return xp.reshape(xp.log(p * (1 - self.eps) + (1.0 - p) * (self.eps / (self.K - 1.0))), (S, N, 1))

    def predict_mean_and_var(self, xp, Fmu, Fvar):
        S, N, D = Fmu.shape
        ps = []
        for k in range(self.K):
            Yk = np.full((N, 1), float(k))
            ps.append(xp.exp(self.predict_density(xp, Fmu, Fvar, Yk))[..., 0])
        ps = xp.stack(ps)
        ps = ps.transpose(1, 2, 0) if xp is NP else ps.permute(1, 2, 0)
        return ps, ps - ps ** 2


class Bernoulli:
    """[UPSTREAM] gpflow 1.1.1 Bernoulli() as /root/reference/tests/test_dgp.py:48-54 builds it: probit link
    p(f) = Phi(f) (1 - 2e-3) + 1e-3, log-density log(p if y == 1 else 1 - p); variational expectations by the base
    Likelihood's 20-point Gauss-Hermite rule (X = mu + sqrt(2 var) x, weights / sqrt(pi)); predict_mean_and_var in the
    probit closed form p = probit(mu / sqrt(1 + var)), var = p - p^2; predict_density = log-density at that p.
    BroadcastingLikelihood (utils.py:76-86) flattens to (S*N, D) with Y tiled — elementwise, so shapes are kept here."""
    kind = "bernoulli"

    def __init__(self, num_gauss_hermite_points=20):
        self.H = num_gauss_hermite_points

    @staticmethod
    def _probit(xp, x):
        return 0.5 * (1.0 + xp.erf(x / math.sqrt(2.0))) * (1 - 2e-3) + 1e-3

    @staticmethod
    def _one(xp, Y, like):
        """1.0 where the target equals 1 ([UPSTREAM] tf.equal(Y, 1)), broadcast over the leading sample axis of `like`"""
        y = np.asarray(Y.detach().numpy() if hasattr(Y, "detach") else Y, dtype=np.float64)
        return xp.asarray(np.broadcast_to((y == 1.0).astype(np.float64), tuple(like.shape)).copy())

    @staticmethod
    def _logp(xp, p, one):
        return xp.log(one * p + (1.0 - one) * (1.0 - p))

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        gh_x, gh_w = np.polynomial.hermite.hermgauss(self.H)
        gh_w = gh_w / math.sqrt(math.pi)
        one = self._one(xp, Y, Fmu)
        out = 0.0
        for x, w in zip(gh_x, gh_w):
            F = Fmu + xp.sqrt(2.0 * Fvar) * float(x)
            out = out + float(w) * self._logp(xp, self._probit(xp, F), one)
        return out

    def predict_mean_and_var(self, xp, Fmu, Fvar):
        p = self._probit(xp, Fmu / xp.sqrt(1.0 + Fvar))
        return p, p - p ** 2

    def predict_density(self, xp, Fmu, Fvar, Y):
        p = self.predict_mean_and_var(xp, Fmu, Fvar)[0]
        return self._logp(xp, p, self._one(xp, Y, p))


class _QuadratureLikelihood:
    """[UPSTREAM] gpflow 1.1.1 likelihoods.Likelihood: the base class's 20-point Gauss-Hermite rule for everything a subclass does
    not override — X = mu + sqrt(2 var) x_k, weights w_k / sqrt(pi):
        variational_expectations = sum_k w_k logp(X_k, Y)
        predict_density          = log sum_k w_k exp(logp(X_k, Y))
        predict_mean_and_var     : E_y = sum_k w_k cm(X_k),  V_y = sum_k w_k (cv(X_k) + cm(X_k)^2) - E_y^2
    (cm / cv = conditional_mean / conditional_variance).  BroadcastingLikelihood (utils.py:76-86) flattens to (S*N, D) with Y tiled:
    elementwise, so shapes are kept here and Y broadcasts over the leading sample axis."""
    H = 20

    def _gh(self):
        gh_x, gh_w = np.polynomial.hermite.hermgauss(self.H)
        return gh_x, gh_w / math.sqrt(math.pi)

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        gh_x, gh_w = self._gh()
        out = 0.0
        for x, w in zip(gh_x, gh_w):
            out = out + float(w) * self.logp(xp, Fmu + xp.sqrt(2.0 * Fvar) * float(x), Y)
        return out

    def predict_density(self, xp, Fmu, Fvar, Y):
        gh_x, gh_w = self._gh()
        out = 0.0
        for x, w in zip(gh_x, gh_w):
            out = out + float(w) * xp.exp(self.logp(xp, Fmu + xp.sqrt(2.0 * Fvar) * float(x), Y))
        return xp.log(out)

    def predict_mean_and_var(self, xp, Fmu, Fvar):
        gh_x, gh_w = self._gh()
        e, q = 0.0, 0.0
        for x, w in zip(gh_x, gh_w):
            X = Fmu + xp.sqrt(2.0 * Fvar) * float(x)
            cm = self.conditional_mean(xp, X)
            e = e + float(w) * cm
            q = q + float(w) * (self.conditional_variance(xp, X) + cm ** 2)
        return e, q - e ** 2


def _lgamma(xp, x):
    if xp is NP:
        return __import__("scipy.special", fromlist=["gammaln"]).gammaln(x)
    return TH._t.lgamma(x)


class Poisson(_QuadratureLikelihood):
    """[UPSTREAM] gpflow 1.1.1 Poisson(invlink=tf.exp, binsize=1.): logp = Y log(lam) - lam - lgamma(Y + 1), lam = exp(F) binsize;
    conditional mean = variance = lam; with the exp link the variational expectations are closed-form:
    Y Fmu - exp(Fmu + Fvar / 2) binsize - lgamma(Y + 1) + Y log(binsize)."""
    kind = "poisson"

    def __init__(self, binsize=1.0):
        self.binsize = float(binsize)

    def logp(self, xp, F, Y):
        Y = xp.asarray(Y)
        return Y * (F + math.log(self.binsize)) - xp.exp(F) * self.binsize - _lgamma(xp, Y + 1.0)

    def conditional_mean(self, xp, F):
        return xp.exp(F) * self.binsize

    conditional_variance = conditional_mean

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        Y = xp.asarray(Y)
        return Y * Fmu - xp.exp(Fmu + Fvar / 2.0) * self.binsize - _lgamma(xp, Y + 1.0) + Y * math.log(self.binsize)


class Exponential(_QuadratureLikelihood):
    """[UPSTREAM] gpflow 1.1.1 Exponential(invlink=tf.exp): logp = -Y / scale - log(scale), scale = exp(F); conditional mean = scale,
    variance = scale^2; closed-form variational expectations with the exp link: -exp(-Fmu + Fvar / 2) Y - Fmu."""
    kind = "exponential"

    def logp(self, xp, F, Y):
        return -xp.asarray(Y) * xp.exp(-F) - F

    def conditional_mean(self, xp, F):
        return xp.exp(F)

    def conditional_variance(self, xp, F):
        return xp.exp(F) ** 2

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        return -xp.exp(-Fmu + Fvar / 2.0) * xp.asarray(Y) - Fmu


class StudentT(_QuadratureLikelihood):
    """[UPSTREAM] gpflow 1.1.1 StudentT(scale=1.0, deg_free=3.0): `scale` a positive Parameter (trainable), `deg_free` a constant;
    logp = lgamma((nu + 1) / 2) - lgamma(nu / 2) - (log nu + log pi) / 2 - log(scale) - (nu + 1) / 2 log(1 + ((Y - F) / scale)^2 / nu);
    conditional mean F, conditional variance scale^2 nu / (nu - 2); everything else by the base class's quadrature."""
    kind = "student_t"

    def __init__(self, scale=1.0, deg_free=3.0):
        self.scale, self.deg_free = scale, float(deg_free)

    def logp(self, xp, F, Y):
        nu = self.deg_free
        const = math.lgamma((nu + 1.0) / 2.0) - math.lgamma(nu / 2.0) - 0.5 * (math.log(nu) + math.log(math.pi))
        sc = self.scale * xp.ones(1)
        return const - xp.log(sc) - 0.5 * (nu + 1.0) * xp.log(1.0 + (1.0 / nu) * ((xp.asarray(Y) - F) / sc) ** 2)

    def conditional_mean(self, xp, F):
        return F

    def conditional_variance(self, xp, F):
        return (self.scale * xp.ones(1)) ** 2 * (self.deg_free / (self.deg_free - 2.0)) + 0.0 * F


class Gamma(_QuadratureLikelihood):
    """[UPSTREAM] gpflow 1.1.1 Gamma(invlink=tf.exp): `shape` a positive Parameter (1.0, trainable);
    logp = -shape log(scale) - lgamma(shape) + (shape - 1) log Y - Y / scale with scale = exp(F); conditional mean shape scale,
    variance shape scale^2; closed-form variational expectations with the exp link:
    -shape Fmu - lgamma(shape) + (shape - 1) log Y - Y exp(-Fmu + Fvar / 2)."""
    kind = "gamma"

    def __init__(self, shape=1.0):
        self.shape = shape

    def logp(self, xp, F, Y):
        sh, Y = self.shape * xp.ones(1), xp.asarray(Y)
        return -sh * F - _lgamma(xp, sh) + (sh - 1.0) * xp.log(Y) - Y * xp.exp(-F)

    def conditional_mean(self, xp, F):
        return (self.shape * xp.ones(1)) * xp.exp(F)

    def conditional_variance(self, xp, F):
        return (self.shape * xp.ones(1)) * xp.exp(F) ** 2

    def variational_expectations(self, xp, Fmu, Fvar, Y):
        sh, Y = self.shape * xp.ones(1), xp.asarray(Y)
        return -sh * Fmu - _lgamma(xp, sh) + (sh - 1.0) * xp.log(Y) - Y * xp.exp(-Fmu + Fvar / 2.0)


class Beta(_QuadratureLikelihood):
    """[UPSTREAM] gpflow 1.1.1 Beta(invlink=probit, scale=1.0): `scale` a positive Parameter (trainable); mean = probit(F) (the
    Bernoulli's probit with its 1e-3 floor), alpha = mean scale, beta = scale - alpha,
    logp = (alpha - 1) log y + (beta - 1) log(1 - y) + lgamma(alpha + beta) - lgamma(alpha) - lgamma(beta) with y clipped to
    [1e-6, 1 - 1e-6]; conditional mean = mean, variance (mean - mean^2) / (scale + 1); everything by the base class's quadrature."""
    kind = "beta"

    def __init__(self, scale=1.0):
        self.scale = scale

    def logp(self, xp, F, Y):
        sc = self.scale * xp.ones(1)
        mean = Bernoulli._probit(xp, F)
        alpha = mean * sc
        beta = sc - alpha
        y = np.clip(np.asarray(Y.detach().numpy() if hasattr(Y, "detach") else Y, dtype=np.float64), 1e-6, 1 - 1e-6)
        y = xp.asarray(y)
        return ((alpha - 1.0) * xp.log(y) + (beta - 1.0) * xp.log(1.0 - y) + _lgamma(xp, alpha + beta) - _lgamma(xp, alpha)
                - _lgamma(xp, beta))

    def conditional_mean(self, xp, F):
        return Bernoulli._probit(xp, F)

    def conditional_variance(self, xp, F):
        mean = Bernoulli._probit(xp, F)
        return (mean - mean ** 2) / (self.scale * xp.ones(1) + 1.0)


# --------------------------------------------------------------------------------------------------
# dgp.py:42-126  DGP_Base
# --------------------------------------------------------------------------------------------------
class DGPOracle:
    def __init__(self, layers, likelihood, num_samples=1, num_data=None, sample_weights=None):
        self.layers, self.likelihood = layers, likelihood
        self.num_samples, self.num_data = num_samples, num_data
        self.sample_weights = sample_weights          # DGP_Quad (dgp.py:145): gh_w, replaces the mean over S by a weighted sum

    # dgp.py:61-76 propagate
    def propagate(self, xp, X, zs, full_cov=False, S=1):
        sX = xp.stack([X] * S)                                          # dgp.py:63
        Fs, Fmeans, Fvars = [], [], []
        F = sX
        for layer, z in zip(self.layers, zs):                           # dgp.py:69-74
            F, Fmean, Fvar = layer.sample_from_conditional(xp, F, z, full_cov=full_cov)
            Fs.append(F); Fmeans.append(Fmean); Fvars.append(Fvar)
        return Fs, Fmeans, Fvars

    # dgp.py:83-90 E_log_p_Y
    def E_log_p_Y(self, xp, X, Y, zs):
        _, Fmeans, Fvars = self.propagate(xp, X, zs, full_cov=False, S=self.num_samples)
        var_exp = self.likelihood.variational_expectations(xp, Fmeans[-1], Fvars[-1], Y)   # dgp.py:89
        if self.sample_weights is not None:                             # DGP_Quad.E_log_p_Y, dgp.py:165-166
            w = xp.asarray(self.sample_weights)
            return xp.sum(var_exp * w[:, None, None], 0)
        return xp.mean(var_exp, 0)                                      # dgp.py:90

    # dgp.py:92-98 _build_likelihood
    def build_likelihood(self, xp, X, Y, zs):
        L = xp.sum(self.E_log_p_Y(xp, X, Y, zs))                        # dgp.py:94
        KL = sum(layer.KL(xp) for layer in self.layers)                 # dgp.py:95
        scale = float(self.num_data if self.num_data is not None else X.shape[0]) / float(X.shape[0])
        return L * scale - KL                                           # dgp.py:96-98

    # dgp.py:121-126 predict_density
    def predict_density(self, xp, Xnew, Ynew, zs, S):
        _, Fmeans, Fvars = self.propagate(xp, Xnew, zs, S=S)
        l = self.likelihood.predict_density(xp, Fmeans[-1], Fvars[-1], Ynew)
        return xp.logsumexp(l - math.log(S), 0)


def quad_points(H, layer_widths):
    """DGP_Quad.__init__ (dgp.py:141-157): tensor-product Gauss-Hermite points split per inner layer as (S,1,D) whitened
    z's plus a dummy for the last layer, and the weights gh_w (summing to one).  [UPSTREAM] gpflow.quadrature.mvhermgauss
    restated through numpy's hermgauss."""
    import itertools
    D = int(sum(layer_widths))
    gx, gw = np.polynomial.hermite.hermgauss(H)
    x = np.array(list(itertools.product(*(gx,) * D))).reshape(H ** D, D) * 2.0 ** 0.5
    w = np.prod(np.array(list(itertools.product(*(gw,) * D))).reshape(H ** D, D), axis=1) * np.pi ** (-0.5 * D)
    zs, s = [], 0
    for d in layer_widths:
        zs.append(x[:, None, s:s + d])
        s += d
    zs.append(np.zeros((1, 1, 1)))
    return zs, w


# --------------------------------------------------------------------------------------------------
# layer_initializations.py:16-52 init_layers_linear  (numpy only; host-side one-off)
# --------------------------------------------------------------------------------------------------
def init_layers_linear(X, Y, Z, kern_specs, num_outputs=None, final_mean="zero", white=False,
                       jitter=DEFAULT_JITTER):
    """kern_specs: list of dicts(kind,input_dim,variance,lengthscales,ARD,white_variance).
    Returns a list of dicts describing each layer (kern, Z, q_mu, q_sqrt, mean)."""
    num_outputs = num_outputs or Y.shape[1]
    out = []
    X_run, Z_run = X.copy(), Z.copy()

    def make(spec, Zl, dim_out, mf):
        kern = Kern(**spec)
        M = Zl.shape[0]
        q_mu = np.zeros((M, dim_out))                                   # layers.py:146-147
        q_sqrt = np.tile(np.eye(M)[None], [dim_out, 1, 1])              # layers.py:149
        if not white:                                                   # layers.py:160-163
            Ku = kern.K(NP, Zl)
            Lu = np.linalg.cholesky(Ku + np.eye(M) * jitter)
            q_sqrt = np.tile(Lu[None], [dim_out, 1, 1])
        return dict(kern=kern, Z=Zl.copy(), q_mu=q_mu, q_sqrt=q_sqrt, mean=mf, white=white)

    for s_in, s_out in zip(kern_specs[:-1], kern_specs[1:]):
        d_in, d_out = s_in["input_dim"], s_out["input_dim"]
        if d_in == d_out:
            mf = MeanFn("identity")                                     # layer_initializations.py:30-31
            W = None
        else:
            if d_in > d_out:                                            # :34-36 PCA projection
                _, _, V = np.linalg.svd(X_run, full_matrices=False)
                W = V[:d_out, :].T
            else:                                                       # :38-39 identity + zero padding
                W = np.concatenate([np.eye(d_in), np.zeros((d_in, d_out - d_in))], 1)
            mf = MeanFn("linear", A=W)                                  # :41-42 (not trainable)
        out.append(make(s_in, Z_run, d_out, mf))
        if W is not None:                                               # :46-48
            Z_run = Z_run.dot(W)
            X_run = X_run.dot(W)
    out.append(make(kern_specs[-1], Z_run, num_outputs, MeanFn(final_mean)))   # :51
    return out


# layer_initializations.py:55-79 init_layers_input_prop; `pads` replaces the reference's np.random.randn draws (one (M, dim_in - D)
# standard-normal array per layer) so that both sides of a parity test see the same inducing inputs
def init_layers_input_prop(X, Y, Z, kern_specs, pads, num_outputs=None, final_mean="zero", white=False,
                           jitter=DEFAULT_JITTER):
    num_outputs = num_outputs or Y.shape[1]
    D = X.shape[1]
    out = []

    def make(spec, Zl, dim_out, mf, prop):
        kern = Kern(**spec)
        M = Zl.shape[0]
        q_mu = np.zeros((M, dim_out))
        q_sqrt = np.tile(np.eye(M)[None], [dim_out, 1, 1])
        if not white:
            Lu = np.linalg.cholesky(kern.K(NP, Zl) + np.eye(M) * jitter)
            q_sqrt = np.tile(Lu[None], [dim_out, 1, 1])
        return dict(kern=kern, Z=Zl.copy(), q_mu=q_mu, q_sqrt=q_sqrt, mean=mf, white=white, input_prop_dim=prop)

    for i, (s_in, s_out) in enumerate(zip(kern_specs[:-1], kern_specs[1:])):
        dim_in, dim_out = s_in["input_dim"], s_out["input_dim"] - D                    # :65-66
        std_in = float(s_in["variance"]) ** 0.5                                          # :67
        Zp = np.concatenate([Z, pads[i] * 2.0 * std_in], 1)                              # :68-69
        out.append(make(s_in, Zp, dim_out, MeanFn("zero"), D))                           # :70
    dim_in = kern_specs[-1]["input_dim"]
    std_in = float(kern_specs[-2]["variance"]) ** 0.5 if dim_in > D else 1.0             # :73
    Zp = np.concatenate([Z, pads[-1] * 2.0 * std_in], 1)
    out.append(make(kern_specs[-1], Zp, num_outputs, MeanFn(final_mean), None))          # :77
    return out


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] tf.train.AdamOptimizer update
# --------------------------------------------------------------------------------------------------
def adam_step(theta, grad, m, v, t, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    """theta <- theta - lr_t * m/(sqrt(v)+eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t). t counts from 1."""
    m[...] = b1 * m + (1 - b1) * grad
    v[...] = b2 * v + (1 - b2) * grad * grad
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    theta[...] = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta


# --------------------------------------------------------------------------------------------------
# [UPSTREAM] gpflow.training.NatGradOptimizer(gamma) step on one layer's (q_mu, q_sqrt)   (SURVEY §8f row 1,
# Appendix C; used at demos/demo_regression_UCI.ipynb:360-366, tests/test_collapsed.py:100)
# --------------------------------------------------------------------------------------------------
def natgrad_step(q_mu, q_sqrt, g_mu, g_sqrt, gamma):
    """One natural-gradient step of a *minimised* loss on D independent Gaussians N(q_mu[:,d], T_d T_d^T).
    g_mu (M,D), g_sqrt (D,M,M): d loss / d q_mu, d loss / d q_sqrt (lower-triangular).  Returns (q_mu+, q_sqrt+)."""
    M, D = q_mu.shape
    mu_new, sq_new = np.empty_like(q_mu), np.empty_like(q_sqrt)
    for d in range(D):
        T = np.tril(q_sqrt[d])
        m = q_mu[:, d]
        Tinv = _sla.solve_triangular(T, np.eye(M), lower=True)
        H = T.T @ np.tril(g_sqrt[d])
        Phi = np.tril(H) - 0.5 * np.diag(np.diag(H))
        Sbar = Tinv.T @ Phi @ Tinv
        Sbar = 0.5 * (Sbar + Sbar.T)                        # d loss / d S  (Cholesky adjoint)
        g1 = g_mu[:, d] - 2.0 * Sbar @ m                    # d loss / d eta_1 ; d loss / d eta_2 = Sbar
        Sinv = Tinv.T @ Tinv
        theta1 = Sinv @ m - gamma * g1
        A = Sinv + 2.0 * gamma * Sbar                       # -2 theta_2
        LA = np.linalg.cholesky(A)
        LAinv = _sla.solve_triangular(LA, np.eye(M), lower=True)
        Splus = LAinv.T @ LAinv
        mu_new[:, d] = Splus @ theta1
        sq_new[d] = np.linalg.cholesky(Splus)
    return mu_new, sq_new
