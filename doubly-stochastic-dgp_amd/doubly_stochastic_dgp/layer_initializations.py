"""Layer construction for DGP (host-side, one-off).  Same two entry points and argument meaning as the reference's
layer_initializations.py:16-52 / :55-79, organised differently: the inter-layer width maps are PLANNED first (one
`_width_map` per boundary) and the layers are then built from the plan."""
import numpy as np

from .gpflow_compat import Identity, Linear, Zero
from .layers import SVGP_Layer


def _width_map(d_from, d_to, cloud):
    """The fixed linear map between two layer widths, or None when they agree (identity mean function).
    narrower: the top `d_to` right singular vectors of `cloud` (the data as seen at this depth) — a PCA projection;
    wider:    [I | 0], the extra coordinates start at zero."""
    if d_from == d_to:
        return None
    if d_from > d_to:
        right = np.linalg.svd(cloud, full_matrices=False)[2]
        return right[:d_to].T.copy()
    out = np.zeros((d_from, d_to))
    out[np.arange(d_from), np.arange(d_from)] = 1.0
    return out


def _plan_widths(X, Z, widths):
    """[(inducing inputs, output width, map-or-None)] for every inner boundary, plus the inducing inputs of the last layer.
    Data and inducing inputs are carried through the maps together so that each layer's Z lives in that layer's input space."""
    cloud, ind = np.array(X, dtype=np.float64), np.array(Z, dtype=np.float64)
    plan = []
    for d_from, d_to in zip(widths[:-1], widths[1:]):
        T = _width_map(d_from, d_to, cloud)
        plan.append((ind, d_to, T))
        if T is not None:
            cloud, ind = cloud @ T, ind @ T
    return plan, ind


def _frozen_linear(T):
    mf = Linear(T)
    mf.set_trainable(False)          # the PCA / padding maps are constants of the model (layer_initializations.py:41-42)
    return mf


def init_layers_linear(X, Y, Z, kernels, num_outputs=None, mean_function=None, Layer=SVGP_Layer, white=False):
    """Inner layers get an identity mean where consecutive kernels share their input width and a fixed linear map where they do
    not (see `_width_map`); the last layer gets `mean_function` (default Zero) and `num_outputs` (default: columns of Y)."""
    final_mean = Zero() if mean_function is None else mean_function
    n_out = num_outputs or Y.shape[1]
    plan, Z_last = _plan_widths(X, Z, [k.input_dim for k in kernels])
    layers = [Layer(k, Zl, width, Identity() if T is None else _frozen_linear(T), white=white)
              for k, (Zl, width, T) in zip(kernels[:-1], plan)]
    layers.append(Layer(kernels[-1], Z_last, n_out, final_mean, white=white))
    return layers


def _padded_inducing(Z, extra, scale):
    """Z with `extra` further columns drawn N(0, (2 scale)^2) from numpy's GLOBAL generator, as the reference's np.random.randn
    does (layer_initializations.py:67,76) — seed it for reproducibility."""
    return np.concatenate([Z, np.random.randn(Z.shape[0], extra) * (2.0 * scale)], 1)


def init_layers_input_prop(X, Y, Z, kernels, num_outputs=None, mean_function=None, Layer=SVGP_Layer, white=False):
    """layer_initializations.py:55-79: every inner layer forwards the D data inputs beside its own outputs
    (Layer(input_prop_dim=D), zero mean), so kernel l sees D + (its layer's own width) inputs; the inducing inputs of the
    additional coordinates are random, scaled by the standard deviation of the kernel that produces them."""
    final_mean = Zero() if mean_function is None else mean_function
    n_out = num_outputs or Y.shape[1]
    D = X.shape[1]
    sd = [float(np.asarray(k.variance.value)) ** 0.5 for k in kernels]
    layers = []
    for l in range(len(kernels) - 1):
        k = kernels[l]
        layers.append(Layer(k, _padded_inducing(Z, k.input_dim - D, sd[l]), kernels[l + 1].input_dim - D, Zero(), white=white,
                            input_prop_dim=D))
    k = kernels[-1]
    layers.append(Layer(k, _padded_inducing(Z, k.input_dim - D, sd[-2] if k.input_dim > D else 1.0), n_out, final_mean, white=white))
    return layers
