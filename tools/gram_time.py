#!/usr/bin/env python
"""dsdgp_gram timed like tools/gram_bench.hip (20 launches between two events, no per-launch events) — library vs microbench."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from doubly_stochastic_dgp import _lib  # noqa: E402
from doubly_stochastic_dgp.engine import Context  # noqa: E402
ctx = Context.get()
rng = np.random.default_rng(1)
ls = np.ones(1)
for (M, R, D, zero) in ((128, 20000, 8, 0), (256, 40000, 9, 0), (512, 40960, 30, 0), (1024, 50000, 8, 0), (1024, 50000, 8, 1)):
    Z = ctx.to_device(rng.standard_normal((M, D)) * (0 if zero else 1))
    X = ctx.to_device(rng.standard_normal((R, D)) * (0 if zero else 1))
    o = ctx.empty(M, R)
    spec = _lib.KernelSpec(kind=0, input_dim=D, ard=0, has_white=0, variance=1.0, white_variance=0.0, lengthscales=ls.ctypes.data_as(_lib.c_double_p))
    def call():
        _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), C.c_void_p(Z.data_ptr()), M, C.c_void_p(X.data_ptr()), R, 0.0, C.c_void_p(o.data_ptr()), R))
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        call()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"n={M} n2={R} D={D} zero={zero}: {us:.1f} us  {8.0 * M * R / us / 1e3:.0f} GB/s", flush=True)
