"""[UPSTREAM] gpflow.training.AdamOptimizer mirror: `AdamOptimizer(0.01).minimize(model, maxiter=K)`
(demos/demo_regression_UCI.ipynb:324).  The step itself (gradient + Adam update) runs in libdsdgp."""


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon

    def minimize(self, model, maxiter=1000):
        n = int(maxiter)
        for it in range(n):
            # steps are asynchronous; the last one synchronises and surfaces a failed Kuu factorisation (CholeskyError)
            model.train_step(self.lr, self.b1, self.b2, self.eps, sync=(it == n - 1))


class NatGradOptimizer:
    """[UPSTREAM] gpflow.training.NatGradOptimizer(gamma): natural-gradient steps on chosen layers' (q_mu, q_sqrt)
    (demos/demo_regression_UCI.ipynb:360-366, using_natural_gradients.ipynb:139-145, tests/test_collapsed.py:100).
    `var_list` is a list of [q_mu, q_sqrt] Parameter pairs, as in the reference."""

    def __init__(self, gamma):
        self.gamma = float(gamma)

    def _layer_indices(self, model, var_list):
        idx = []
        for pair in var_list:
            ids = {id(p) for p in pair}
            hit = [i for i, layer in enumerate(model.layers) if {id(layer.q_mu), id(layer.q_sqrt)} == ids]
            if len(hit) != 1:
                raise ValueError("each var_list entry must be the [q_mu, q_sqrt] pair of one layer of the model")
            idx.append(hit[0])
        return idx

    def minimize(self, model, var_list, maxiter=1, X=None, Y=None, zs=None):
        layers = self._layer_indices(model, var_list)
        eng = model.engine()
        for _ in range(int(maxiter)):
            # tf.gradients w.r.t. var_list only: the reverse pass stops below the lowest layer in it, and — var_list holding nothing
            # but (q_mu, q_sqrt) pairs — it never visits that layer's Kuf / Kuu adjoints (they feed Z and the kernel hyper-parameters)
            model._build_likelihood(X, Y, zs=zs, with_grad=True, grad_from_layer=min(layers), grad_q_only=True)
            for l in layers:
                eng.natgrad_step(l, self.gamma)
        eng.ctx.sync()
