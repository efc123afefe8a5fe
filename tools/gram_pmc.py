#!/usr/bin/env python
"""dsdgp_gram launches for the PMC passes of the Gram sub-roofline (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
from doubly_stochastic_dgp import _lib  # noqa: E402
from doubly_stochastic_dgp.engine import Context  # noqa: E402
ctx = Context.get()
rng = np.random.default_rng(1)
ls = np.ones(1)
for (M, R, D) in ((128, 20000, 8), (256, 40000, 9), (512, 40960, 30), (1024, 50000, 8)):
    Z, X = ctx.to_device(rng.standard_normal((M, D))), ctx.to_device(rng.standard_normal((R, D)))
    o = ctx.empty(M, R)
    spec = _lib.KernelSpec(kind=0, input_dim=D, ard=0, has_white=0, variance=1.0, white_variance=0.0, lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    for _ in range(4):
        _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), C.c_void_p(Z.data_ptr()), M, C.c_void_p(X.data_ptr()), R, 0.0, C.c_void_p(o.data_ptr()), R))
    ctx.sync()
