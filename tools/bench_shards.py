#!/usr/bin/env python
"""Step time of the config-2 model on the per-GPU shard sizes of a STRONG-scaling run (global minibatch 1000 x S = 20 split over
N = 1 / 2 / 4 / 8 ranks: 1000 / 500 / 250 / 125 rows per rank), measured on ONE GPU: without the exchange step, and with the
data-parallel code path on a ONE-RANK RCCL group (the collectives are identities, what is measured is everything around them:
the host-side cost of one flat all-reduce per step vs. one per layer issued from the library's bucket callback).  Input of the
modelled scaling table in DESIGN.md section 7 (no multi-GPU curve can be measured from this container)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    sys.path.insert(0, p)
import bench as B  # noqa: E402


def dp_step_ms(cfg, mb, bucketed):
    import torch
    from doubly_stochastic_dgp.distributed import attach
    model, X, Y, Z = B.build_model(cfg, 0, 1, mb)
    attach(model, 0, 1, bucketed=bucketed)
    for _ in range(20):
        model.train_step(0.01)
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            model.train_step(0.01)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 200 * 1e3)
    return round(min(reps), 4)


def main():
    import torch
    import torch.distributed as dist
    from doubly_stochastic_dgp.engine import Context
    Context.get()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
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
                              ms_per_step_elbo_plus_adam=round(split_ms, 4),
                              ms_per_step_dp_flat_1rank=dp_step_ms(cfg, mb, False), ms_per_step_dp_bucketed_1rank=dp_step_ms(cfg, mb, True),
                              n_theta=int(eng.n_theta))), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
