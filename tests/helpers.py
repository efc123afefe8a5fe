"""Shared builders: the same (spec, state) feeds the CPU oracle and the HIP-backed host mirror."""
import numpy as np

from oracle import dgp_oracle as O
from oracle import model as OM


def kern_spec(kind, D, variance=1.0, lengthscales=1.0, ARD=False, white_variance=None):
    return dict(kind=kind, input_dim=D, variance=variance, lengthscales=lengthscales, ARD=ARD,
                white_variance=white_variance)


def product_kernel(spec):
    from doubly_stochastic_dgp.gpflow_compat import RBF, Matern52, White
    cls = {"rbf": RBF, "matern52": Matern52}[spec["kind"]]
    k = cls(spec["input_dim"], variance=spec["variance"], lengthscales=spec["lengthscales"], ARD=spec["ARD"])
    if spec.get("white_variance") is not None:
        k = k + White(spec["input_dim"], variance=spec["white_variance"])
    return k


def make_case(X, Y, Z, kern_specs, white=False, jitter=1e-6, lik_var=0.1, S=2, num_data=None, seed=0,
              randomize=True, q_sqrt_scale=None, minibatch_size=None, num_classes=None, bernoulli=False, likelihood=None,
              lik_aux=None):
    """Returns (spec, state, model): oracle description and the product DGP with identical parameters."""
    from doubly_stochastic_dgp import settings
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import Bernoulli, Beta, Exponential, Gamma, Gaussian, MultiClass, Poisson, StudentT
    rng = np.random.RandomState(seed)
    lds = O.init_layers_linear(X, Y, Z, kern_specs, white=white, jitter=jitter, num_outputs=num_classes)
    for i, l in enumerate(lds):
        if randomize:
            l["q_mu"] = 0.3 * rng.randn(*l["q_mu"].shape)
            D, M = l["q_sqrt"].shape[0], l["q_sqrt"].shape[1]
            l["q_sqrt"] = l["q_sqrt"] * 0.7 + 0.05 * np.tril(rng.randn(D, M, M))
        if q_sqrt_scale is not None and i < len(lds) - 1:
            l["q_sqrt"] = l["q_sqrt"] * q_sqrt_scale
    # likelihood: "poisson" (lik_aux = binsize) / "exponential" / "student_t" (lik_var = scale, lik_aux = deg_free) / "gamma" (lik_var =
    # shape) / "beta" (lik_var = scale)
    likname = likelihood or ("multiclass" if num_classes else ("bernoulli" if bernoulli else "gaussian"))
    sl, state = OM.state_from_layers(lds, lik_variance=lik_var, likelihood=likname)
    spec = dict(jitter=jitter, white=white, likelihood=likname, layers=sl, num_classes=num_classes, lik_aux=lik_aux)
    with settings.temp_jitter(jitter):
        if likname == "poisson":
            lik = Poisson(binsize=lik_aux or 1.0)
        elif likname == "exponential":
            lik = Exponential()
        elif likname == "student_t":
            lik = StudentT(scale=lik_var, deg_free=lik_aux or 3.0)
        elif likname == "gamma":
            lik = Gamma(shape=lik_var)
        elif likname == "beta":
            lik = Beta(scale=lik_var)
        else:
            lik = MultiClass(num_classes) if num_classes else (Bernoulli() if bernoulli else Gaussian(variance=lik_var))
        model = DGP(X, Y, Z, [product_kernel(k) for k in kern_specs], lik, white=white, num_outputs=num_classes,
                    num_samples=S, num_data=num_data, minibatch_size=minibatch_size)
    for l, layer in zip(lds, model.layers):
        layer.q_mu = l["q_mu"]
        layer.q_sqrt = l["q_sqrt"]
    return spec, state, model


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
