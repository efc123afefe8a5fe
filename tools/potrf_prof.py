#!/usr/bin/env python
"""dsdgp_potrf at n = 1024 (one matrix) a few times — run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
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
ctx = Context.get()
A0 = torch.randn(n, n, dtype=torch.float64, device="cuda")
A0 = A0 @ A0.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
for it in range(6):
    A = A0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(ctx.lib.dsdgp_potrf(ctx.handle, 1, n, C.c_void_p(A.data_ptr()), n, n * n, C.cast(info.data_ptr(), C.POINTER(C.c_int))))
    ctx.sync()
    print(f"potrf n={n}: {(time.perf_counter() - t0) * 1e6:.0f} us (wall, incl. plan build)")
ref = torch.linalg.cholesky(A0)
print("relerr", float((torch.tril(A) - ref).abs().max() / ref.abs().max()))
