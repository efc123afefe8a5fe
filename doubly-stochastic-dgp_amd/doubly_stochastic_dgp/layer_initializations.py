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
