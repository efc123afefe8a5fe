#!/usr/bin/env python
"""A/B of DSDGP_FORCE settings (read when a model is created) on one BASELINE config shape, in ONE process:
    python tools/ab_force.py 2 "pipe_tail=0" "pipe_tail=1" "red_ahead=0"
prints ms/step (best and all of three batches) per setting, interleaved twice so that clock drift shows."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC  # noqa: E402


def time_setting(cfg_id, force, n):
    import torch
    os.environ["DSDGP_FORCE"] = force
    model, step = BC.build(cfg_id)
    for _ in range(10):
        step()
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / n * 1e3)
    elbo = model.train_step(0.01, sync=True)
    return dict(cfg=cfg_id, force=force, ms_best=round(min(reps), 4), ms=[round(r, 4) for r in reps],
                steps_per_s=round(1e3 / min(reps), 1), elbo=elbo)


if __name__ == "__main__":
    cfg_id = int(sys.argv[1])
    n = 300 if cfg_id <= 2 else 10
    for rnd in range(2):
        for force in sys.argv[2:]:
            print(json.dumps(time_setting(cfg_id, force, n)), flush=True)
