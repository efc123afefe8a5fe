"""Minimal *argument objects* with GPflow 1.1.1's constructor signatures — kernels, likelihoods, mean functions,
Parameter / Parameterized — so that user code written against the reference keeps working:

    from doubly_stochastic_dgp.gpflow_compat import RBF, Matern52, White, Gaussian, Zero, Identity, Linear

They carry plain numpy parameter values and hyper-parameter bookkeeping only; every evaluation on the DGP path happens
in libdsdgp (HIP).  [UPSTREAM] = behaviour of GPflow 1.1.1 that the reference relies on.
"""
import numpy as np

from . import settings

SOFTPLUS_LOWER = 1e-6     # [UPSTREAM] transforms.positive = Log1pe(lower=1e-6)


def positive_backward(y):
    y = np.asarray(y, dtype=np.float64) - SOFTPLUS_LOWER
    if np.any(y <= 0):
        raise ValueError("positive-transformed parameter must exceed 1e-6")
    return y + np.log(-np.expm1(-y))


def positive_forward(x):
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64)) + SOFTPLUS_LOWER


class Parameter:
    """[UPSTREAM] gpflow.params.Parameter: `.value` / `.read_value()`, assignment through the parent attribute,
    `.trainable` / `.set_trainable()`.  transform in {None, 'positive', 'tril'}."""

    def __init__(self, value, transform=None, trainable=True):
        self._value = np.array(value, dtype=np.float64)
        self.transform = transform
        self.trainable = bool(trainable)
        self._owners = []            # engines to notify on host-side assignment / to pull from on read

    # -- host <-> device coherence hooks (set by engine.Engine)
    def _pull(self):
        for eng in self._owners:
            eng.sync_to_host()

    def _touch(self):
        for eng in self._owners:
            eng.mark_host_dirty()

    @property
    def value(self):
        self._pull()
        return self._value

    def read_value(self):
        return self.value.copy()

    def assign(self, v):
        self._pull()
        v = np.array(v, dtype=np.float64)
        if v.shape != self._value.shape:
            v = np.broadcast_to(v, self._value.shape).copy()
        self._value = v
        self._touch()

    @property
    def shape(self):
        return self._value.shape

    def set_trainable(self, flag):
        self.trainable = bool(flag)
        for eng in self._owners:
            eng.mark_structure_dirty()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value, dtype=dtype)

    def __repr__(self):
        return f"Parameter({self._value!r}, transform={self.transform}, trainable={self.trainable})"


class Parameterized:
    """Attribute assignment onto an existing Parameter assigns its value (GPflow semantics used by the reference tests:
    `m.layers[-1].q_mu = q_mu`, `lik.variance = 0.01`)."""

    def __setattr__(self, name, value):
        cur = self.__dict__.get(name)
        if isinstance(cur, Parameter) and not isinstance(value, Parameter):
            cur.assign(value)
        else:
            object.__setattr__(self, name, value)

    def parameters(self):
        for v in self.__dict__.values():
            if isinstance(v, Parameter):
                yield v
            elif isinstance(v, Parameterized):
                yield from v.parameters()
            elif isinstance(v, (list, tuple)):
                for e in v:
                    if isinstance(e, Parameterized):
                        yield from e.parameters()

    def set_trainable(self, flag):
        for p in self.parameters():
            p.set_trainable(flag)


# --------------------------------------------------------------------------------------------------
# kernels
# --------------------------------------------------------------------------------------------------
class Kernel(Parameterized):
    def __add__(self, other):
        return Sum(self, other)


class Stationary(Kernel):
    kind = None

    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False, name=None):
        if active_dims is not None:
            raise NotImplementedError("active_dims is not used by the DGP path")
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        if lengthscales is None:
            lengthscales = 1.0
        ls = np.asarray(lengthscales, dtype=np.float64)
        if self.ARD:
            ls = np.broadcast_to(ls, (self.input_dim,)).copy()
        else:
            ls = ls.reshape(())
        self.variance = Parameter(variance, transform="positive")
        self.lengthscales = Parameter(ls, transform="positive")


class RBF(Stationary):
    kind = "rbf"


SquaredExponential = RBF


class Matern52(Stationary):
    kind = "matern52"


class White(Kernel):
    def __init__(self, input_dim, variance=1.0, active_dims=None, name=None):
        self.input_dim = int(input_dim)
        self.variance = Parameter(variance, transform="positive")


class Sum(Kernel):
    """k1 + k2; only Stationary + White is meaningful on the DGP path (demo_step_function.ipynb:111)."""

    def __init__(self, k1, k2):
        parts = []
        for k in (k1, k2):
            parts.extend(k.kernels if isinstance(k, Sum) else [k])
        self.kernels = parts
        self.input_dim = max(k.input_dim for k in parts)

    def split(self):
        stat = [k for k in self.kernels if isinstance(k, Stationary)]
        white = [k for k in self.kernels if isinstance(k, White)]
        if len(stat) != 1 or len(white) > 1 or len(stat) + len(white) != len(self.kernels):
            raise NotImplementedError("only <stationary> + White sums are supported")
        return stat[0], (white[0] if white else None)


def split_kernel(kern):
    if isinstance(kern, Sum):
        return kern.split()
    if isinstance(kern, Stationary):
        return kern, None
    raise NotImplementedError(f"kernel {type(kern).__name__} is not on the DGP path")


# --------------------------------------------------------------------------------------------------
# mean functions
# --------------------------------------------------------------------------------------------------
class MeanFunction(Parameterized):
    pass


class Zero(MeanFunction):
    kind = "zero"

    def __init__(self, output_dim=1):
        self.output_dim = output_dim


class Identity(MeanFunction):
    kind = "identity"

    def __init__(self, input_dim=None):
        self.input_dim = input_dim


class Linear(MeanFunction):
    """[UPSTREAM] gpflow.mean_functions.Linear: y = X A + b with A (D_in, D_out) and b (D_out).  Both are free parameters
    unless fixed with set_trainable(False) (init_layers_linear fixes its PCA / padding maps, layer_initializations.py:41-42);
    trainable or biased instances are optimised on the device like every other parameter."""
    kind = "linear"

    def __init__(self, A=None, b=None):
        A = np.ones((1, 1)) if A is None else np.atleast_2d(np.asarray(A, dtype=np.float64))
        b = np.zeros(A.shape[1]) if b is None else np.broadcast_to(np.asarray(b, dtype=np.float64).ravel(), (A.shape[1],)).copy()
        self.A = Parameter(A)
        self.b = Parameter(b)

    def in_theta(self):
        """Whether A / b must live in the optimiser's parameter vector (else: a fixed map without bias)."""
        return self.A.trainable or self.b.trainable or bool(np.any(self.b.value != 0.0))


# --------------------------------------------------------------------------------------------------
# likelihoods
# --------------------------------------------------------------------------------------------------
class Likelihood(Parameterized):
    pass


class Gaussian(Likelihood):
    kind = "gaussian"

    def __init__(self, variance=1.0, var=None):
        self.variance = Parameter(variance if var is None else var, transform="positive")


class MultiClass(Likelihood):
    kind = "multiclass"

    def __init__(self, num_classes, invlink=None):
        self.num_classes = int(num_classes)
        self.epsilon = 1e-3


class Bernoulli(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.Bernoulli() with the probit link (/root/reference/tests/test_dgp.py:48-54): targets 1 select
    p = probit(f), every other value (the test draws -1 / 1) 1 - p; no parameters."""
    kind = "bernoulli"

    def __init__(self, invlink=None):
        if invlink is not None:
            raise NotImplementedError("only the default probit link is on the built path")


def _exp_link_only(invlink):
    if invlink is not None and invlink is not np.exp and getattr(invlink, "__name__", "") != "exp":
        raise NotImplementedError("only the default exp link is on the built path")


class Poisson(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.Poisson(invlink=tf.exp, binsize=1.): counts with rate exp(f) * binsize (no free parameter)."""
    kind = "poisson"

    def __init__(self, invlink=None, binsize=1.0):
        _exp_link_only(invlink)
        self.binsize = float(binsize)
        if not self.binsize > 0.0:
            raise ValueError("binsize must be positive")


class Exponential(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.Exponential(invlink=tf.exp): positive targets with scale exp(f) (no free parameter)."""
    kind = "exponential"

    def __init__(self, invlink=None):
        _exp_link_only(invlink)


class StudentT(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.StudentT(scale=1.0, deg_free=3.0): `scale` is a positive (trainable) Parameter, `deg_free` a
    constant."""
    kind = "student_t"

    def __init__(self, scale=1.0, deg_free=3.0):
        self.deg_free = float(deg_free)
        if not self.deg_free > 0.0:
            raise ValueError("deg_free must be positive")
        self.scale = Parameter(scale, transform="positive")


class Gamma(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.Gamma(invlink=tf.exp): positive targets, scale exp(f); `shape` is a positive (trainable)
    Parameter, 1.0 upstream (settable here for convenience)."""
    kind = "gamma"

    def __init__(self, invlink=None, shape=1.0):
        _exp_link_only(invlink)
        self.shape = Parameter(shape, transform="positive")


class Beta(Likelihood):
    """[UPSTREAM] gpflow.likelihoods.Beta(invlink=probit, scale=1.0): targets in (0, 1) (clipped to [1e-6, 1 - 1e-6]), mean probit(f);
    `scale` is a positive (trainable) Parameter."""
    kind = "beta"

    def __init__(self, invlink=None, scale=1.0):
        if invlink is not None:
            raise NotImplementedError("only the default probit link is on the built path")
        self.scale = Parameter(scale, transform="positive")


class InducingPoints(Parameterized):
    """[UPSTREAM] gpflow.features.InducingPoints — holder of Z (layers.py:153)."""

    def __init__(self, Z):
        self.Z = Parameter(Z)

    def __len__(self):
        return self.Z.shape[0]
