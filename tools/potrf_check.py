import ctypes as C, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
import torch
from doubly_stochastic_dgp import _lib
from doubly_stochastic_dgp.engine import Context
ctx = Context.get()
info = torch.zeros(1, dtype=torch.int32, device="cuda")
torch.manual_seed(0)
for n, batch in ((192, 1), (256, 2), (500, 1), (512, 1), (570, 1), (576, 3), (640, 1), (1000, 1), (1024, 1), (1024, 3), (1088, 1), (1536, 1), (2048, 1), (2112, 1)):
    A0 = torch.randn(batch, n, n, dtype=torch.float64, device="cuda")
    A0 = A0 @ A0.transpose(1, 2) + n * torch.eye(n, dtype=torch.float64, device="cuda")
    best = 1e9
    for it in range(4):
        A = A0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(ctx.lib.dsdgp_potrf(ctx.handle, batch, n, C.c_void_p(A.data_ptr()), n, n * n, None))
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e6)
    ref = torch.linalg.cholesky(A0)
    print(f"n={n} batch={batch}: {best:.0f} us wall, relerr {float((A - ref).abs().max() / ref.abs().max()):.3e}", flush=True)
