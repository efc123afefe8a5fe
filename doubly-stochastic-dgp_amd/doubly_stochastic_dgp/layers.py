"""Host mirror of the reference's layer classes (layers.py:36-246): `Layer`, `SVGP_Layer` with the same constructor,
attributes (`q_mu`, `q_sqrt`, `feature.Z`, `kern`, `mean_function`, `num_outputs`, `white`) and methods
(`conditional_ND`, `conditional_SND`, `sample_from_conditional`, `KL`).  All arithmetic runs in libdsdgp (HIP)."""
import numpy as np

from . import settings
from .gpflow_compat import InducingPoints, Parameter, Parameterized, split_kernel
from .utils import reparameterize


def _host_K_symm(kern, Z):
    """kern.compute_K_symm(Z) for the one-off q(u)=p(u) initialisation (layers.py:160-163); the reference does this
    step on the host with numpy too."""
    stat, white = split_kernel(kern)
    ls = np.asarray(stat.lengthscales._value, dtype=np.float64)
    Zs = Z / ls
    d = Zs[:, None, :] - Zs[None, :, :]
    r2 = np.sum(d * d, -1)
    var = float(stat.variance._value)
    if stat.kind == "rbf":
        K = var * np.exp(-0.5 * r2)
    else:
        r = np.sqrt(r2 + 1e-12)
        K = var * (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * r2) * np.exp(-np.sqrt(5.0) * r)
    if white is not None:
        K = K + float(white.variance._value) * np.eye(Z.shape[0])
    return K


def concat_input_prop(X, p, samples, mean, var, full_cov=False):
    """layers.py:105-117: prepend the propagated inputs X[..., :p] to the samples and means, and zeros to the variances
    (pure copies — no arithmetic)."""
    X_prop = X[:, :, :p]
    samples = np.concatenate([X_prop, samples], 2)
    mean = np.concatenate([X_prop, mean], 2)
    if full_cov:
        zeros = np.zeros(var.shape[:3] + (p,))
        var = np.concatenate([zeros, var], 3)
    else:
        var = np.concatenate([np.zeros_like(X_prop), var], 2)
    return samples, mean, var


class Layer(Parameterized):
    def __init__(self, input_prop_dim=None, **kwargs):
        """input_prop_dim: the first dimensions of X to propagate (layers.py:36-50); None = no input propagation."""
        self.input_prop_dim = input_prop_dim

    def conditional_ND(self, X, full_cov=False):
        raise NotImplementedError

    def KL(self):
        return 0.0

    # layers.py:52-74
    def conditional_SND(self, X, full_cov=False):
        X = np.asarray(X, dtype=np.float64)
        if full_cov:                                              # tf.map_fn over S, layers.py:66-69
            ms, vs = zip(*[self.conditional_ND(X[s], full_cov=True) for s in range(X.shape[0])])
            return np.stack(ms), np.stack(vs)
        S, N, D = X.shape
        mean, var = self.conditional_ND(X.reshape(S * N, D))
        return mean.reshape(S, N, self.num_outputs), var.reshape(S, N, self.num_outputs)

    # layers.py:76-119
    def sample_from_conditional(self, X, z=None, full_cov=False):
        mean, var = self.conditional_SND(X, full_cov=full_cov)
        if z is None:
            z = self._engine().randn(mean.shape)
        z = np.broadcast_to(np.asarray(z, dtype=np.float64), mean.shape)
        samples = reparameterize(mean, var, z, full_cov=full_cov)
        if self.input_prop_dim:                                   # layers.py:105-117
            samples, mean, var = concat_input_prop(np.asarray(X, dtype=np.float64), self.input_prop_dim, samples, mean, var,
                                                   full_cov)
        return samples, mean, var


class SVGP_Layer(Layer):
    def __init__(self, kern, Z, num_outputs, mean_function, white=False, input_prop_dim=None, **kwargs):
        Layer.__init__(self, input_prop_dim, **kwargs)
        Z = np.array(Z, dtype=np.float64)
        self.num_inducing = Z.shape[0]
        self.q_mu = Parameter(np.zeros((self.num_inducing, num_outputs)))                     # layers.py:146-147
        q_sqrt = np.tile(np.eye(self.num_inducing)[None, :, :], [num_outputs, 1, 1])          # layers.py:149
        self.q_sqrt = Parameter(q_sqrt, transform="tril")
        self.feature = InducingPoints(Z)
        self.kern = kern
        self.mean_function = mean_function
        self.num_outputs = int(num_outputs)
        self.white = bool(white)
        if not self.white:                                                                    # layers.py:160-163
            Ku = _host_K_symm(kern, Z)
            Lu = np.linalg.cholesky(Ku + np.eye(Z.shape[0]) * settings.jitter)
            self.q_sqrt = np.tile(Lu[None, :, :], [num_outputs, 1, 1])
        object.__setattr__(self, "_standalone_engine", None)
        object.__setattr__(self, "_model_engine", None)     # (engine, index) once owned by a DGP_Base

    def _engine(self):
        if self._model_engine is not None:
            return self._model_engine[0]
        if self._standalone_engine is None:
            from .engine import Engine
            from .gpflow_compat import Gaussian
            object.__setattr__(self, "_standalone_engine", Engine([self], Gaussian(), self.white))
        return self._standalone_engine

    def _index(self):
        return self._model_engine[1] if self._model_engine is not None else 0

    # layers.py:178-219
    def conditional_ND(self, X, full_cov=False):
        eng = self._engine()
        if full_cov:                                              # layers.py:206-209: var is (N, N, D_out)
            mean, var = eng.layer_conditional_full(self._index(), np.asarray(X, dtype=np.float64))
            eng.ctx.sync()
            return mean.cpu().numpy(), var.cpu().numpy()
        mean, var = eng.layer_conditional(self._index(), np.asarray(X, dtype=np.float64))
        eng.ctx.sync()
        return mean.cpu().numpy(), var.cpu().numpy()

    # layers.py:221-246
    def KL(self):
        return self._engine().layer_kl(self._index())
