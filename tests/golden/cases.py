"""Deterministic case recipes for the golden fixtures.  Inputs are regenerated from legacy numpy RandomState seeds
(bit-stable across numpy versions); only the ORACLE OUTPUTS are stored in the .npz files next to this module."""
import numpy as np

from tests.helpers import kern_spec

CASES = {
    # reference test shapes: tests/test_dgp.py:29-34,66 (N=19, D_X=2, D_Y=3, Z=X, Matern52 l=0.5)
    "svgp_matern52": dict(N=19, D=2, DY=3, M=19, S=2, L=1, kind="matern52", ls=0.5, var=1.0, white=False, lik=0.01,
                          jitter=1e-6, seed=0, num_data=None, zx=True),
    "svgp_matern52_white": dict(N=19, D=2, DY=3, M=19, S=2, L=1, kind="matern52", ls=0.5, var=1.0, white=True, lik=0.01,
                                jitter=1e-6, seed=0, num_data=None, zx=True),
    # tests/test_dgp.py:122-127 (N=2, RBF l=0.1... widened to l=0.3 so Kuu is well conditioned), 2 layers
    "two_layer_1d": dict(N=2, D=1, DY=1, M=2, S=5, L=2, kind="rbf", ls=0.3, var=1.0, white=False, lik=0.01, jitter=1e-6,
                         seed=1, num_data=None, zx=True),
    # config-1-shaped: 1 layer, M=50, S=1, minibatch 100 of 7372
    "cfg1_slice": dict(N=100, D=8, DY=1, M=50, S=1, L=1, kind="rbf", ls=1.0, var=1.0, white=False, lik=1.0, jitter=1e-6,
                       seed=2, num_data=7372, zx=False),
    # config-2-shaped slice: 3 layers, M=128, S=4, N=64 of 7372, inner q_sqrt * 1e-5 (demo_regression_UCI.ipynb:183)
    "cfg2_slice": dict(N=64, D=8, DY=1, M=128, S=4, L=3, kind="rbf", ls=1.0, var=1.0, white=False, lik=1.0, jitter=1e-6,
                       seed=3, num_data=7372, zx=False, demo_scale=1e-5),
    # config-3-shaped slice: 5 layers, D=9, M=256, S=2
    "cfg3_slice": dict(N=48, D=9, DY=1, M=256, S=2, L=5, kind="rbf", ls=1.5, var=1.0, white=False, lik=1.0, jitter=1e-6,
                       seed=4, num_data=41157, zx=False),
    # Matern-5/2 with ARD lengthscales + White kernel (Sum), whitened layers, two layers with a step down 3 -> 2
    "ard_white_sum": dict(N=30, D=3, DY=2, M=12, S=3, L=2, kind="matern52", ls=[0.6, 1.0, 1.7], var=1.3, white=True, lik=0.1,
                          jitter=1e-6, seed=5, num_data=90, zx=False, ard=True, wvar=0.05, dims=[3, 2]),
    # MultiClass(3) / RobustMax likelihood, three layers (demo_mnist.ipynb:99-104 in miniature)
    "multiclass": dict(N=24, D=2, DY=1, M=10, S=2, L=3, kind="rbf", ls=1.1, var=2.0, white=False, lik=None, jitter=1e-6,
                       seed=6, num_data=100, zx=False, classes=3),
    # Bernoulli() likelihood, targets in {-1, 1}, two whitened layers: tests/test_dgp.py:48-54 (N=19, D_X=2, D_Y=3, Z=X)
    "bernoulli": dict(N=19, D=2, DY=3, M=19, S=2, L=2, kind="matern52", ls=0.5, var=1.0, white=True, lik=None, jitter=1e-6,
                      seed=7, num_data=None, zx=True, bernoulli=True),
}


def inputs(name):
    c = CASES[name]
    rng = np.random.RandomState(c["seed"])
    N, D, M, S, L = c["N"], c["D"], c["M"], c["S"], c["L"]
    X = rng.uniform(size=(N, D)) if c["zx"] else rng.randn(N, D)
    Y = rng.randn(N, c["DY"])
    Z = X.copy() if c["zx"] else rng.randn(M, D) * 1.2
    if c.get("classes"):
        Y = rng.randint(0, c["classes"], size=(N, 1)).astype(np.float64)
    if c.get("bernoulli"):
        Y = rng.choice([-1.0, 1.0], N * c["DY"]).reshape(N, c["DY"])
    kdims = c.get("dims") or [D] * L                       # kernel input dims per layer (step-down cases)
    specs = []
    for d in kdims:
        ls = c["ls"]
        if c.get("ard"):
            ls = list(np.asarray(ls, dtype=np.float64)[:d])
        specs.append(kern_spec(c["kind"], d, c["var"], ls, ARD=bool(c.get("ard")), white_variance=c.get("wvar")))
    dims = list(kdims[1:]) + [c.get("classes") or c["DY"]]
    zs = [rng.randn(S, N, d) for d in dims]
    return c, X, Y, Z, specs, zs


def build(name):
    """(spec, state, model-or-None, X, Y, zs, case): model is built only when the HIP library can be used."""
    from tests.helpers import make_case
    c, X, Y, Z, specs, zs = inputs(name)
    spec, state, model = make_case(X, Y, Z, specs, white=c["white"], jitter=c["jitter"], lik_var=c["lik"] or 1.0, S=c["S"],
                                   num_data=c["num_data"], seed=c["seed"], q_sqrt_scale=c.get("demo_scale"),
                                   num_classes=c.get("classes"), bernoulli=bool(c.get("bernoulli")))
    return spec, state, model, X, Y, zs, c
