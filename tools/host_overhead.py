import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/doubly-stochastic-dgp_amd"); sys.path.insert(0, "/root/repo/tools")
import torch

# host-side cost of one training step (python + ctypes + hipLaunch) against the step's GPU time
import bench_configs as BC
for cid in (1, 2):
    model, step = BC.build(cid)
    for _ in range(50): step()
    torch.cuda.synchronize()
    n = 8     # few enough steps that the launch queue never back-pressures the host
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"cfg{cid} (n={n}): enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
