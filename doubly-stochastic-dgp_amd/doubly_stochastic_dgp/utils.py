"""reparameterize (utils.py:22-51) and BroadcastingLikelihood (utils.py:54-121) of the reference, on the device."""
import ctypes as C

import numpy as np

from . import settings


def reparameterize(mean, var, z, full_cov=False):
    """mean + z * sqrt(var + jitter) for the diagonal case (utils.py:40-41); var=None returns mean (utils.py:37-38)."""
    if var is None:
        return mean
    if full_cov:
        raise NotImplementedError("full_cov reparameterisation is a 'next' row (SURVEY §8f)")
    from . import _lib
    from .engine import Context, ptr
    ctx = Context.get()
    m, v, zz = (ctx.to_device(np.broadcast_to(a, np.shape(mean))) for a in (mean, var, z))
    out = ctx.empty(*np.shape(mean))
    ctx.torch.cuda.current_stream().synchronize()
    _lib.check(ctx.lib.dsdgp_reparameterize(ctx.handle, ptr(m), ptr(v), ptr(zz), float(settings.jitter), m.numel(),
                                            ptr(out)))
    ctx.sync()
    return out.cpu().numpy()


class BroadcastingLikelihood:
    """Wrapper giving every likelihood method (S,N,D) semantics with Y of shape (N,D) (utils.py:54-121).  The Gaussian
    case is evaluated by libdsdgp; `.likelihood` is the wrapped object (`model.likelihood.likelihood.variance`)."""

    def __init__(self, likelihood):
        self.likelihood = likelihood
        from .gpflow_compat import Gaussian
        self.needs_broadcasting = not isinstance(likelihood, Gaussian)

    def _gauss(self):
        if self.needs_broadcasting:
            raise NotImplementedError("only the Gaussian likelihood is built so far (MultiClass is SURVEY §8f rank 2)")
        return float(self.likelihood.variance.value)

    def _run(self, fn, Fmu, Fvar, Y):
        from . import _lib
        from .engine import Context, ptr
        ctx = Context.get()
        Fmu = np.asarray(Fmu, dtype=np.float64)
        S, N, D = Fmu.shape
        m, v, y = ctx.to_device(Fmu), ctx.to_device(np.broadcast_to(Fvar, Fmu.shape)), ctx.to_device(Y)
        out = ctx.empty(N, D)
        ctx.torch.cuda.current_stream().synchronize()
        _lib.check(getattr(ctx.lib, fn)(ctx.handle, ptr(m), ptr(v), ptr(y), N, S, D, self._gauss(), ptr(out)))
        ctx.sync()
        return out.cpu().numpy()

    def variational_expectations_mean(self, Fmu, Fvar, Y):
        """reduce_mean over S of variational_expectations (dgp.py:89-90)."""
        return self._run("dsdgp_gauss_var_exp", Fmu, Fvar, Y)

    def predict_density_logmeanexp(self, Fmu, Fvar, Y):
        """logsumexp_S(predict_density) - log S (dgp.py:124-126)."""
        return self._run("dsdgp_gauss_predict_density", Fmu, Fvar, Y)

    def predict_mean_and_var(self, Fmu, Fvar):
        from . import _lib
        from .engine import Context, ptr
        ctx = Context.get()
        v = ctx.to_device(np.asarray(Fvar, dtype=np.float64))
        out = ctx.empty(*v.shape)
        ctx.torch.cuda.current_stream().synchronize()
        _lib.check(ctx.lib.dsdgp_add_scalar(ctx.handle, ptr(v), self._gauss(), v.numel(), ptr(out)))
        ctx.sync()
        return np.asarray(Fmu, dtype=np.float64), out.cpu().numpy()
