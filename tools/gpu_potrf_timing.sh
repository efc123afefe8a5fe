#!/bin/bash
mkdir -p gpurun_out
DSDGP_POTRF_TIMING=1 python - <<'PY' > gpurun_out/potrf_timing.log 2>&1
import sys, numpy as np
sys.path.insert(0, "doubly-stochastic-dgp_amd"); sys.path.insert(0, ".")
from tests.helpers import kern_spec, make_case
rng = np.random.RandomState(0)
X, Y = rng.randn(300, 8), rng.randn(300, 1)
Z = rng.randn(128, 8)
spec, state, model = make_case(X, Y, Z, [kern_spec("rbf", 8)] * 3, S=2)
model.engine().prepare()
model.layers[0].q_mu = model.layers[0].q_mu.value + 0.1
model.engine().prepare()
PY
cat gpurun_out/potrf_timing.log | tail -5
