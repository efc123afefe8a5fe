"""Spec + state-dict front end of the CPU oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

A model is described by a plain ``spec`` dict and a ``state`` dict of *unconstrained* numpy arrays (the
free variables GPflow would optimise):

spec  = {"jitter": 1e-6, "white": False, "likelihood": "gaussian"|"multiclass"|"bernoulli"|"poisson"|"exponential"|"student_t"|"gamma"|"beta",
         "num_classes": K, "lik_aux": Poisson binsize / StudentT deg_free (optional),
         "layers": [{"kind": "rbf"|"matern52", "input_dim": D_in, "ARD": bool, "has_white": bool,
                     "mean": "zero"|"identity"|"linear", "mean_A": ndarray|None, "kvar_identity": bool (optional)}, ...]}
state = {"l{i}.Z": (M,D_in), "l{i}.q_mu": (M,D_out), "l{i}.q_sqrt": (D_out,M,M)  [tril part is the free var],
         "l{i}.kern_variance_raw": (), "l{i}.kern_lengthscales_raw": () or (D_in,),
         "l{i}.white_variance_raw": () [if has_white], "lik_variance_raw": () [gaussian: variance; student_t / beta: scale; gamma: shape]}
"""
import math
import numpy as np

from . import dgp_oracle as O


def constrained_to_raw(y):
    return O.positive_backward_np(y)


def state_from_layers(layer_dicts, lik_variance=1.0, likelihood="gaussian"):
    """layer_dicts as produced by dgp_oracle.init_layers_linear -> (spec_layers, state)."""
    spec_layers, state = [], {}
    for i, ld in enumerate(layer_dicts):
        k = ld["kern"]
        spec_layers.append(dict(kind=k.kind, input_dim=k.input_dim, ARD=k.ARD,
                                has_white=k.white_variance is not None,
                                mean=ld["mean"].kind, mean_A=ld["mean"].A, mean_trainable=bool(ld.get("mean_trainable")),
                                input_prop_dim=ld.get("input_prop_dim")))
        if ld.get("mean_trainable"):    # [UPSTREAM] mean_functions.Linear(A, b) as free parameters (identity transform)
            state[f"l{i}.mean_A"] = np.array(ld["mean"].A, dtype=np.float64)
            state[f"l{i}.mean_b"] = np.array(ld["mean"].b if ld["mean"].b is not None else np.zeros(ld["mean"].A.shape[1]),
                                             dtype=np.float64)
        state[f"l{i}.Z"] = np.array(ld["Z"], dtype=np.float64)
        state[f"l{i}.q_mu"] = np.array(ld["q_mu"], dtype=np.float64)
        state[f"l{i}.q_sqrt"] = np.array(ld["q_sqrt"], dtype=np.float64)
        if ld.get("kvar_identity"):
            spec_layers[-1]["kvar_identity"] = True
            state[f"l{i}.kern_variance_raw"] = np.array(k.variance, dtype=np.float64)
        else:
            state[f"l{i}.kern_variance_raw"] = np.array(constrained_to_raw(k.variance))
        ls = np.asarray(k.lengthscales, dtype=np.float64)
        if k.ARD and ls.ndim == 0:
            ls = np.full((k.input_dim,), float(ls))
        state[f"l{i}.kern_lengthscales_raw"] = np.array(constrained_to_raw(ls))
        if k.white_variance is not None:
            state[f"l{i}.white_variance_raw"] = np.array(constrained_to_raw(k.white_variance))
    if likelihood in ("gaussian", "student_t", "gamma", "beta"):      # the likelihood's one positive parameter (variance / scale / shape)
        state["lik_variance_raw"] = np.array(constrained_to_raw(lik_variance))
    return spec_layers, state


def build(xp, spec, state, num_samples=1, num_data=None, sample_weights=None):
    """Instantiate the oracle model under backend ``xp`` from (spec, state)."""
    layers = []
    for i, ls in enumerate(spec["layers"]):
        g = lambda n: xp.asarray(state[f"l{i}.{n}"])
        kern = O.Kern(ls["kind"], ls["input_dim"],
                      # "kvar_identity": a variance Parameter without transform (reference tests/test_dgp.py:79-85,
                      # NoTransformMatern52 with variance 1e-24): the state entry is the variance itself
                      variance=(g("kern_variance_raw") if ls.get("kvar_identity")
                                else O.positive_forward(xp, g("kern_variance_raw"))),
                      lengthscales=O.positive_forward(xp, g("kern_lengthscales_raw")),
                      ARD=ls["ARD"],
                      white_variance=(O.positive_forward(xp, g("white_variance_raw"))
                                      if ls.get("has_white") else None))
        if ls.get("mean_trainable"):
            mf = O.MeanFn(ls["mean"], A=g("mean_A"), b=g("mean_b"))
        else:
            mf = O.MeanFn(ls["mean"], A=ls.get("mean_A"))
        layers.append(O.SVGPLayer(kern, g("Z"), g("q_mu"), g("q_sqrt"), mf,
                                  white=spec["white"], jitter=spec["jitter"],
                                  input_prop_dim=ls.get("input_prop_dim")))
    if spec["likelihood"] == "gaussian":
        lik = O.Gaussian(O.positive_forward(xp, xp.asarray(state["lik_variance_raw"])))
    elif spec["likelihood"] == "bernoulli":
        lik = O.Bernoulli()
    elif spec["likelihood"] == "poisson":
        lik = O.Poisson(binsize=spec.get("lik_aux") or 1.0)
    elif spec["likelihood"] == "exponential":
        lik = O.Exponential()
    elif spec["likelihood"] == "student_t":
        lik = O.StudentT(O.positive_forward(xp, xp.asarray(state["lik_variance_raw"])), deg_free=spec.get("lik_aux") or 3.0)
    elif spec["likelihood"] == "gamma":
        lik = O.Gamma(O.positive_forward(xp, xp.asarray(state["lik_variance_raw"])))
    elif spec["likelihood"] == "beta":
        lik = O.Beta(O.positive_forward(xp, xp.asarray(state["lik_variance_raw"])))
    else:
        lik = O.MultiClass(spec["num_classes"])
    return O.DGPOracle(layers, lik, num_samples=num_samples, num_data=num_data, sample_weights=sample_weights)


def elbo(spec, state, X, Y, zs, num_samples, num_data=None, sample_weights=None):
    m = build(O.NP, spec, state, num_samples, num_data, sample_weights)
    return float(m.build_likelihood(O.NP, np.asarray(X, float), np.asarray(Y, float), zs))


def elbo_and_grad(spec, state, X, Y, zs, num_samples, num_data=None, sample_weights=None):
    """ELBO and d(ELBO)/d(state) via torch CPU float64 autograd on the identical op sequence
    (stands in for tf.gradients [UPSTREAM]).  q_sqrt gradients are lower-triangular by construction."""
    import torch
    leaves = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in state.items()}
    m = build(O.TH, spec, leaves, num_samples, num_data,
              None if sample_weights is None else torch.as_tensor(np.asarray(sample_weights, dtype=np.float64)))
    zs_t = [None if z is None else torch.as_tensor(np.asarray(z, dtype=np.float64)) for z in zs]
    val = m.build_likelihood(O.TH, torch.as_tensor(np.asarray(X, float)), torch.as_tensor(np.asarray(Y, float)), zs_t)
    val.backward()
    grads = {k: (t.grad.numpy().copy() if t.grad is not None else np.zeros_like(state[k])) for k, t in leaves.items()}
    return float(val.detach()), grads


def elbo_and_grad_sharded(spec, state, X, Y, zs, num_samples, data_scale, kl_weight):
    """One data-parallel rank's term  data_scale * sum E_log_p_Y - kl_weight * sum KL  and its gradient, flattened in
    sorted-key order (the layout contract of distributed.allreduce_flat: [grad | elbo])."""
    import torch
    leaves = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in state.items()}
    m = build(O.TH, spec, leaves, num_samples, None)
    Xt, Yt = torch.as_tensor(np.asarray(X, float)), torch.as_tensor(np.asarray(Y, float))
    zs_t = [torch.as_tensor(np.asarray(z, dtype=np.float64)) for z in zs]
    L = O.TH.sum(m.E_log_p_Y(O.TH, Xt, Yt, zs_t))
    KL = sum(layer.KL(O.TH) for layer in m.layers)
    val = L * data_scale - kl_weight * KL
    val.backward()
    flat = np.concatenate([leaves[k].grad.numpy().ravel() for k in sorted(leaves)] + [[float(val.detach())]])
    return flat


def propagate(spec, state, X, zs, S, full_cov=False):
    m = build(O.NP, spec, state, S)
    return m.propagate(O.NP, np.asarray(X, float), zs, full_cov=full_cov, S=S)
