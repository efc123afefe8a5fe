#!/usr/bin/env python
"""Step time of the config-2 model on the per-GPU shard sizes of a STRONG-scaling run (global minibatch 1000 x S = 20 split over
N = 1 / 2 / 4 / 8 ranks: 1000 / 500 / 250 / 125 rows per rank), measured on ONE GPU without the exchange step.  Input of the
modelled scaling table in DESIGN.md §7 (no multi-GPU curve can be measured from this container)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
import bench as B  # noqa: E402


def main():
    import torch
    for world in (1, 2, 4, 8):
        cfg = dict(B.CFG)
        mb = cfg["mb"] // world
        model, X, Y, Z = B.build_model(cfg, 0, 1, mb)
        for _ in range(20):
            model.train_step(0.01)
        reps = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                model.train_step(0.01)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 300 * 1e3)
        # the two halves a data-parallel step is made of (the all-reduce sits between them)
        eng = model.engine()
        Xb, Yb = model.next_minibatch()
        for _ in range(10):
            eng.elbo(Xb, Yb, cfg["S"], seed=1, with_grad=True, sync=False)
            eng.adam_step(0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            eng.elbo(Xb, Yb, cfg["S"], seed=i, with_grad=True, sync=False)
            eng.adam_step(0.01)
        torch.cuda.synchronize()
        split_ms = (time.perf_counter() - t0) / 200 * 1e3
        print(json.dumps(dict(world=world, rows_per_rank=mb, ms_per_step_fused=round(min(reps), 4),
                              ms_per_step_elbo_plus_adam=round(split_ms, 4), n_theta=int(eng.n_theta))), flush=True)


if __name__ == "__main__":
    main()
