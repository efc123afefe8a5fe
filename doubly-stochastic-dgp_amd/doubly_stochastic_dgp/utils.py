"""reparameterize (utils.py:22-51) and BroadcastingLikelihood (utils.py:54-121) of the reference, on the device."""
import ctypes as C

import numpy as np

from . import settings


def reparameterize(mean, var, z, full_cov=False):
    """mean + z * sqrt(var + jitter) for the diagonal case (utils.py:40-41); var=None returns mean (utils.py:37-38)."""
    if var is None:
        return mean
    from . import _lib
    from .engine import Context, ptr
    ctx = Context.get()
    if full_cov:                                                  # utils.py:43-51
        mean = np.asarray(mean, dtype=np.float64)
        S, N, D = mean.shape
        m, v = ctx.to_device(mean), ctx.to_device(var)
        zz = ctx.to_device(np.broadcast_to(z, mean.shape))
        out = ctx.empty(S, N, D)
        _lib.check(ctx.lib.dsdgp_reparameterize_full(ctx.handle, ptr(m), ptr(v), ptr(zz), float(settings.jitter), N, D, S,
                                                     ptr(out)))
        ctx.sync()
        return out.cpu().numpy()
    m, v, zz = (ctx.to_device(np.broadcast_to(a, np.shape(mean))) for a in (mean, var, z))
    out = ctx.empty(*np.shape(mean))
    _lib.check(ctx.lib.dsdgp_reparameterize(ctx.handle, ptr(m), ptr(v), ptr(zz), float(settings.jitter), m.numel(),
                                            ptr(out)))
    ctx.sync()
    return out.cpu().numpy()


def mvhermgauss(H, D):
    """[UPSTREAM] gpflow.quadrature.mvhermgauss: the H**D tensor-product Gauss-Hermite nodes (H**D, D) and weights (H**D,)
    for the weight function exp(-|x|^2) (consumed by DGP_Quad, dgp.py:143-145)."""
    import itertools
    gh_x, gh_w = np.polynomial.hermite.hermgauss(int(H))
    x = np.array(list(itertools.product(*(gh_x,) * int(D))))
    w = np.prod(np.array(list(itertools.product(*(gh_w,) * int(D)))), axis=1)
    return x.reshape(int(H) ** int(D), int(D)), w


class BroadcastingLikelihood:
    """Wrapper giving every likelihood method (S,N,D) semantics with Y of shape (N,D) (utils.py:54-121): the Gaussian
    broadcasts Y[None]; every other likelihood is evaluated on the flattened (S*N, D) arrays with Y tiled S times
    (utils.py:76-86).  Evaluated by libdsdgp; `.likelihood` is the wrapped object
    (`model.likelihood.likelihood.variance`)."""

    def __init__(self, likelihood):
        self.likelihood = likelihood
        from .gpflow_compat import Bernoulli, Beta, Exponential, Gamma, Gaussian, MultiClass, Poisson, StudentT
        self.needs_broadcasting = not isinstance(likelihood, Gaussian)
        self.bernoulli = isinstance(likelihood, Bernoulli)
        # Poisson / Exponential / Gamma (exp link), StudentT, Beta: elementwise like Bernoulli, evaluated by dsdgp_lik_var_exp / dsdgp_lik_predict
        self.generic = isinstance(likelihood, (Poisson, Exponential, StudentT, Gamma, Beta))
        if not isinstance(likelihood, (Gaussian, MultiClass, Bernoulli, Poisson, Exponential, StudentT, Gamma, Beta)):
            raise NotImplementedError(f"likelihood {type(likelihood).__name__} is not on the built path "
                                      "(Gaussian, MultiClass, Bernoulli, Poisson, Exponential, StudentT, Gamma, Beta are)")

    def generic_args(self):
        """(kind, p0, p1) of dsdgp_lik_var_exp / dsdgp_lik_predict: p0 = StudentT.scale / Gamma.shape / Beta.scale, p1 = Poisson.binsize /
        StudentT.deg_free."""
        from . import _lib
        from .gpflow_compat import Beta, Exponential, Gamma, Poisson
        lik = self.likelihood
        if isinstance(lik, Poisson):
            return _lib.LIK_POISSON, 1.0, lik.binsize
        if isinstance(lik, Exponential):
            return _lib.LIK_EXPONENTIAL, 1.0, 1.0
        if isinstance(lik, Gamma):
            return _lib.LIK_GAMMA, float(lik.shape.value), 1.0
        if isinstance(lik, Beta):
            return _lib.LIK_BETA, float(lik.scale.value), 1.0
        return _lib.LIK_STUDENT_T, float(lik.scale.value), lik.deg_free

    def check_targets(self, Y):
        """MultiClass: Y must hold integer class labels in [0, num_classes) — the device kernel indexes its per-class
        accumulators with them ([UPSTREAM] tf.one_hot / gather would error or zero-fill; one-hot or NaN targets are a bug)."""
        if not self.needs_broadcasting:
            return
        Y = np.asarray(Y, dtype=np.float64)
        if self.bernoulli or self.generic:
            # [UPSTREAM] tf.where(tf.equal(Y, 1), p, 1 - p): any finite target is accepted (the reference test draws -1 / 1); the
            # count / positive-target likelihoods evaluate their log densities on whatever finite targets they are given, as upstream
            if Y.ndim != 2 or not np.all(np.isfinite(Y)):
                raise ValueError(f"{type(self.likelihood).__name__} targets must be a finite (N, D) array, got shape {Y.shape}")
            return
        K = self.likelihood.num_classes
        if Y.ndim != 2 or Y.shape[1] != 1:
            raise ValueError(f"MultiClass targets must have shape (N, 1) with labels in [0, {K}), got {Y.shape}")
        if not np.all(np.isfinite(Y)) or np.any(Y != np.floor(Y)) or np.any(Y < 0) or np.any(Y >= K):
            raise ValueError(f"MultiClass targets must be integer labels in [0, {K})")

    def _run(self, mode, Fmu, Fvar, Y, weights=None):
        self.check_targets(Y)
        from . import _lib
        from .engine import Context, ptr
        ctx = Context.get()
        Fmu = np.asarray(Fmu, dtype=np.float64)
        S, N, D = Fmu.shape
        m, v, y = ctx.to_device(Fmu), ctx.to_device(np.broadcast_to(Fvar, Fmu.shape)), ctx.to_device(Y)
        w = None
        if weights is not None:
            if mode != 0 or np.shape(weights) != (S,):
                raise ValueError("sample weights apply to the variational expectations and must have shape (S,)")
            w = ctx.to_device(np.asarray(weights, dtype=np.float64))
        wp = ptr(w) if w is not None else None
        if not self.needs_broadcasting:
            out = ctx.empty(N, D)
            lv = float(self.likelihood.variance.value)
            if mode == 0:
                _lib.check(ctx.lib.dsdgp_gauss_var_exp(ctx.handle, ptr(m), ptr(v), ptr(y), N, S, D, lv, wp, ptr(out)))
            else:
                _lib.check(ctx.lib.dsdgp_gauss_predict_density(ctx.handle, ptr(m), ptr(v), ptr(y), N, S, D, lv, ptr(out)))
        elif self.bernoulli:
            out = ctx.empty(N, D)
            _lib.check(ctx.lib.dsdgp_bernoulli_var_exp(ctx.handle, ptr(m), ptr(v), ptr(y), N, S, D, mode, wp, ptr(out)))
        elif self.generic:
            out = ctx.empty(N, D)
            kind, p0, p1 = self.generic_args()
            _lib.check(ctx.lib.dsdgp_lik_var_exp(ctx.handle, kind, p0, p1, ptr(m), ptr(v), ptr(y), N, S, D, mode, wp, ptr(out)))
        else:
            out = ctx.empty(N, 1)
            _lib.check(ctx.lib.dsdgp_multiclass_var_exp(ctx.handle, ptr(m), ptr(v), ptr(y), N, S, D, mode, wp, ptr(out)))
        ctx.sync()
        return out.cpu().numpy()

    def variational_expectations_mean(self, Fmu, Fvar, Y, weights=None):
        """reduce_mean over S of variational_expectations (dgp.py:89-90); with `weights` (S,) the weighted sum over S of
        DGP_Quad.E_log_p_Y (dgp.py:165-166)."""
        return self._run(0, Fmu, Fvar, Y, weights)

    def predict_density_logmeanexp(self, Fmu, Fvar, Y):
        """logsumexp_S(predict_density) - log S (dgp.py:124-126)."""
        return self._run(1, Fmu, Fvar, Y)

    def predict_mean_and_var(self, Fmu, Fvar):
        from . import _lib
        from .engine import Context, ptr
        ctx = Context.get()
        Fmu = np.asarray(Fmu, dtype=np.float64)
        v = ctx.to_device(np.broadcast_to(np.asarray(Fvar, dtype=np.float64), Fmu.shape))
        if not self.needs_broadcasting:
            out = ctx.empty(*v.shape)
            _lib.check(ctx.lib.dsdgp_add_scalar(ctx.handle, ptr(v), float(self.likelihood.variance.value), v.numel(),
                                                ptr(out)))
            ctx.sync()
            return Fmu, out.cpu().numpy()
        S, N, K = Fmu.shape
        m = ctx.to_device(Fmu)
        om, ov = ctx.empty(S, N, K), ctx.empty(S, N, K)
        if self.bernoulli:
            _lib.check(ctx.lib.dsdgp_bernoulli_predict(ctx.handle, ptr(m), ptr(v), S * N * K, ptr(om), ptr(ov)))
            ctx.sync()
            return om.cpu().numpy(), ov.cpu().numpy()
        if self.generic:
            kind, p0, p1 = self.generic_args()
            _lib.check(ctx.lib.dsdgp_lik_predict(ctx.handle, kind, p0, p1, ptr(m), ptr(v), S * N * K, ptr(om), ptr(ov)))
            ctx.sync()
            return om.cpu().numpy(), ov.cpu().numpy()
        _lib.check(ctx.lib.dsdgp_multiclass_predict(ctx.handle, ptr(m), ptr(v), S * N, K, ptr(om), ptr(ov)))
        ctx.sync()
        return om.cpu().numpy(), ov.cpu().numpy()
