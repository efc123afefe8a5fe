#!/usr/bin/env python
"""Factor + inverse of Kuu at M = 1024 through the model path (what bench.py's sub_rooflines.potrf_trtri times): HIP events around the
launch sequence (ProfScope "potrf"), mean of `reps` calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
import numpy as np  # noqa: E402
from doubly_stochastic_dgp import _lib  # noqa: E402
from doubly_stochastic_dgp.dgp import DGP  # noqa: E402
from doubly_stochastic_dgp.engine import Context  # noqa: E402
from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian  # noqa: E402

ctx = Context.get()
rng = np.random.default_rng(1)
ctx.prof_enable(True)
for M in [int(a) for a in sys.argv[1:]] or [1024]:
    Xs = rng.standard_normal((M + 64, 8))
    m1 = DGP(Xs, Xs[:, :1], Xs[:M] + 0.01 * rng.standard_normal((M, 8)), [RBF(8)], Gaussian(), num_samples=1)
    e1 = m1.engine()
    e1.prepare()
    for rep in range(3):
        ctx.prof_read("potrf")
        reps = 5
        for _ in range(reps):
            _lib.check(e1.lib.dsdgp_model_theta_changed(e1.model))
            e1._needs_prepare = True
            e1.prepare()
        ms, cnt = ctx.prof_read("potrf")
        print(f"M={M}: {1e3 * ms / reps:.1f} us per factor + inverse", flush=True)
