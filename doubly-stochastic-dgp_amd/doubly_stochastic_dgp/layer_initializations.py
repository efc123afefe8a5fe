"""Layer construction for DGP — mirrors layer_initializations.py:16-52 of the reference (host-side, one-off)."""
import numpy as np

from .gpflow_compat import Identity, Linear, Zero
from .layers import SVGP_Layer


def init_layers_linear(X, Y, Z, kernels, num_outputs=None, mean_function=None, Layer=SVGP_Layer, white=False):
    """Identity mean when widths agree, fixed PCA projection when stepping down, [I | 0] padding when stepping up;
    Z (and the running X used for the PCA) are pushed through the same maps."""
    mean_function = Zero() if mean_function is None else mean_function
    num_outputs = num_outputs or Y.shape[1]
    layers = []
    X_running, Z_running = np.array(X, dtype=np.float64), np.array(Z, dtype=np.float64)
    for kern_in, kern_out in zip(kernels[:-1], kernels[1:]):
        dim_in, dim_out = kern_in.input_dim, kern_out.input_dim
        W = None
        if dim_in == dim_out:
            mf = Identity()
        else:
            if dim_in > dim_out:
                _, _, V = np.linalg.svd(X_running, full_matrices=False)
                W = V[:dim_out, :].T
            else:
                W = np.concatenate([np.eye(dim_in), np.zeros((dim_in, dim_out - dim_in))], 1)
            mf = Linear(W)
            mf.set_trainable(False)
        layers.append(Layer(kern_in, Z_running, dim_out, mf, white=white))
        if W is not None:
            Z_running = Z_running.dot(W)
            X_running = X_running.dot(W)
    layers.append(Layer(kernels[-1], Z_running, num_outputs, mean_function, white=white))
    return layers


def init_layers_input_prop(X, Y, Z, kernels, num_outputs=None, mean_function=None, Layer=SVGP_Layer, white=False):
    """layer_initializations.py:55-79: every inner layer propagates the D data inputs alongside its outputs
    (Layer(input_prop_dim=D), zero mean); the inducing inputs of the extra dimensions are drawn N(0, (2 std)^2) with numpy's
    global RNG exactly as the reference does (np.random.randn) — seed it for reproducibility."""
    mean_function = Zero() if mean_function is None else mean_function
    num_outputs = num_outputs or Y.shape[1]
    D, M = X.shape[1], Z.shape[0]
    layers = []
    for kern_in, kern_out in zip(kernels[:-1], kernels[1:]):
        dim_in, dim_out = kern_in.input_dim, kern_out.input_dim - D
        std_in = float(np.asarray(kern_in.variance.value)) ** 0.5
        pad = np.random.randn(M, dim_in - D) * 2.0 * std_in
        layers.append(Layer(kern_in, np.concatenate([Z, pad], 1), dim_out, Zero(), white=white, input_prop_dim=D))
    dim_in = kernels[-1].input_dim
    std_in = float(np.asarray(kernels[-2].variance.value)) ** 0.5 if dim_in > D else 1.0
    pad = np.random.randn(M, dim_in - D) * 2.0 * std_in
    layers.append(Layer(kernels[-1], np.concatenate([Z, pad], 1), num_outputs, mean_function, white=white))
    return layers
