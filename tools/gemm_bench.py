#!/usr/bin/env python
"""dsdgp_gemm throughput at the shapes of the M x M algebra and of the trsm panel updates: TFLOP/s against the 78.6 fp64 MFMA peak."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
import torch  # noqa: E402
from doubly_stochastic_dgp import _lib  # noqa: E402
from doubly_stochastic_dgp.engine import Context  # noqa: E402

SHAPES = [(1024, 1024, 1024, 0, 0), (1024, 1024, 1024, 1, 0), (1024, 1024, 1024, 0, 1), (512, 512, 512, 0, 0), (2048, 2048, 2048, 0, 0),
          (896, 50000, 128, 0, 0), (4096, 4096, 256, 0, 0), (256, 256, 256, 0, 0)]


def main():
    ctx = Context.get()
    for m, n, k, ta, tb in SHAPES:
        A = torch.randn((k, m) if ta else (m, k), dtype=torch.float64, device="cuda")
        B = torch.randn((n, k) if tb else (k, n), dtype=torch.float64, device="cuda")
        Cm = torch.zeros(m, n, dtype=torch.float64, device="cuda")
        args = (ctx.handle, ta, tb, m, n, k, 1.0, C.c_void_p(A.data_ptr()), A.shape[1], C.c_void_p(B.data_ptr()), B.shape[1], 0.0,
                C.c_void_p(Cm.data_ptr()), n)
        for _ in range(3):
            _lib.check(ctx.lib.dsdgp_gemm(*args))
        ctx.sync()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.lib.dsdgp_gemm(*args)
        ctx.sync()
        us = (time.perf_counter() - t0) / reps * 1e6
        ref = (A.T if ta else A) @ (B.T if tb else B)
        err = float((Cm - ref).abs().max() / ref.abs().max())
        print(f"m={m} n={n} k={k} tA={ta} tB={tb}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:6.2f} TFLOP/s  frac {2.0 * m * n * k / us / 1e6 / 78.6:.3f}  relerr {err:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
