#!/usr/bin/env python
"""A/B helper: per-kernel HIP-event times (ms/step) of one BASELINE config shape with the library selected by DSDGP_LIB_PATH."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC  # noqa: E402


def main(cfg_id):
    import torch
    model, step = BC.build(cfg_id)
    for _ in range(3):
        step()
    n = 10 if cfg_id >= 3 else 300
    reps = []
    for _ in range(3):                      # best of three timed batches (run-to-run spread at cfg 2 is ~3 %)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / n * 1e3)
    ms = min(reps)
    ctx = model.engine().ctx
    os.environ["DSDGP_NO_OVERLAP"] = "1"
    ctx.prof_enable(True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    out = {"cfg": cfg_id, "ms_per_step": round(ms, 4), "ms_reps": [round(r, 4) for r in reps]}
    for name in ("layer_fwd", "layer_bwd", "layer_last", "wgrad", "gemm", "potrf"):
        t, cnt = ctx.prof_read(name)
        out[name] = round(t / 5, 3)
    print(out)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        main(int(a))
