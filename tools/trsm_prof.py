#!/usr/bin/env python
"""dsdgp_trsm at n x nrhs (default 1024 x 50000) a few times — run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
import torch  # noqa: E402
from doubly_stochastic_dgp import _lib  # noqa: E402
from doubly_stochastic_dgp.engine import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
ctx = Context.get()
L = torch.tril(torch.randn(n, n, dtype=torch.float64, device="cuda")) / n ** 0.5 + 2.0 * torch.eye(n, dtype=torch.float64, device="cuda")
B0 = torch.randn(n, nrhs, dtype=torch.float64, device="cuda")
for trans in (0, 1):
    for it in range(4):
        B = B0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(ctx.lib.dsdgp_trsm(ctx.handle, trans, n, nrhs, C.c_void_p(L.data_ptr()), n, C.c_void_p(B.data_ptr()), nrhs))
        ctx.sync()
        dt = time.perf_counter() - t0
        print(f"trsm trans={trans} n={n} nrhs={nrhs}: {dt * 1e6:.0f} us  {n * n * nrhs / dt / 1e12:.1f} TFLOP/s (n^2 nrhs)")
    ref = torch.linalg.solve_triangular(L.T if trans else L, B0, upper=bool(trans))
    print("relerr", float((B - ref).abs().max() / ref.abs().max()))
