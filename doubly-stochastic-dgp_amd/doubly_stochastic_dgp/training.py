"""[UPSTREAM] gpflow.training.AdamOptimizer mirror: `AdamOptimizer(0.01).minimize(model, maxiter=K)`
(demos/demo_regression_UCI.ipynb:324).  The step itself (gradient + Adam update) runs in libdsdgp."""


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon

    def minimize(self, model, maxiter=1000):
        for _ in range(int(maxiter)):
            model.train_step(self.lr, self.b1, self.b2, self.eps)
        model.engine().ctx.sync()
