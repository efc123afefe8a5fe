#!/usr/bin/env python
"""Headline benchmark: ELBO-steps/sec of the 3-layer doubly-stochastic DGP (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = minibatch gather + forward ELBO (dgp.py:92-98) + reverse-mode gradient w.r.t. every trainable parameter +
Adam(0.01) update — i.e. one `session.run(opt_op)` of demos/demo_regression_UCI.ipynb:324 — on kin8nm-SHAPED synthetic
data (real UCI data cannot be downloaded): N_data=7372, D=8, widths 8->8->8->1, RBF, M=128, S=20, minibatch 1000, fp64,
white=False, inner q_sqrt * 1e-5.  Inputs are resident in HBM before the timed region.

N>1: one process per GPU (torch.distributed / RCCL); every rank takes its own 1000-row minibatch x all S samples (rows are
independent through all layers), the flat gradient is summed with ONE all-reduce per step; value = N * global steps/s
(minibatch-1000 ELBO steps per second, whole job) -> "scaling": "weak".

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (HIP-event timed dominant kernel) and
`cpu_baseline` (the CPU oracle — a port of the GPflow/TF op sequence, NOT TF itself — timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet FP64 matrix (= FP64 vector) peak; microbench in profiles/
CFG = dict(n_data=7372, D=8, M=128, S=20, mb=1000, L=3)


def layer_shapes(cfg):
    """(R_l, D_in, D_out) per layer; layer 0 sees the N distinct rows only (its S copies are identical, dgp.py:63)."""
    D, S, mb = cfg["D"], cfg["S"], cfg["mb"]
    return [(mb, D, D), (S * mb, D, D), (S * mb, D, 1)]


def algorithmic_flops(cfg):
    """SURVEY §8d contract figures, per kernel class and per step (non-white, w=2)."""
    M = cfg["M"]
    fwd, wg = [], []
    for R, Din, Dout in layer_shapes(cfg):
        f = M * R * (2 * Din + 4) + 2 * M * M * R + 2 * M * R * Dout + 2 * M * R + Dout * (M * M * R + 2 * M * R) + 4 * R * Dout
        fwd.append(f)
        wg.append((2 + Dout) * M * M * R)
    small = sum(M * M * (Din + 2) + M ** 3 / 3 + Dout * M ** 3 / 3 + 2 * M * M * Dout for _, Din, Dout in layer_shapes(cfg))
    return dict(layer_fwd=sum(fwd), layer_bwd=sum(fwd), wgrad=sum(wg), small=small,
                step=3 * (sum(fwd) + small))


def make_synthetic(n, d, seed=0):
    """kin8nm-SHAPED synthetic regression data (SURVEY §8d recipe): X ~ N(0,1), Y = standardise(sin(Xw1) + 0.1 (Xw2)^2
    + 0.1 eps) — mimics demos/datasets.py:74-83 standardisation.  Real UCI data cannot be downloaded (no network)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w1, w2 = rng.standard_normal(d), rng.standard_normal(d)
    Y = np.sin(X @ w1) + 0.1 * (X @ w2) ** 2 + 0.1 * rng.standard_normal(n)
    Y = ((Y - Y.mean()) / (Y.std() + 1e-6))[:, None]
    return X, Y


def default_Z(X, M, seed=0):
    """kmeans2(minit='points') as demos/run_regression.py:57, with a permutation fallback."""
    try:
        from scipy.cluster.vq import kmeans2
        return kmeans2(X, M, minit="points", seed=seed)[0]
    except Exception:
        rng = np.random.default_rng(seed)
        return X[rng.permutation(X.shape[0])[:M]].copy()


def build_model(cfg, rank, world):
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian
    X, Y = make_synthetic(cfg["n_data"], cfg["D"], seed=0)
    Z = default_Z(X, cfg["M"], seed=0)
    kernels = [RBF(cfg["D"]) for _ in range(cfg["L"])]
    model = DGP(X, Y, Z, kernels, Gaussian(), num_samples=cfg["S"], minibatch_size=cfg["mb"])
    for layer in model.layers[:-1]:
        layer.q_sqrt = layer.q_sqrt.value * 1e-5                      # demo_regression_UCI.ipynb:183
    if world > 1:
        from doubly_stochastic_dgp.distributed import attach
        attach(model, rank, world)
    return model, X, Y, Z


def cpu_baseline(cfg, X, Y, Z, budget_s=20.0):
    """CPU restatement of the reference op sequence (forward + autograd + Adam) on this box's host cores."""
    import torch
    from oracle import dgp_oracle as O, model as OM
    specs = [dict(kind="rbf", input_dim=cfg["D"], variance=1.0, lengthscales=1.0, ARD=False, white_variance=None)] * cfg["L"]
    lds = O.init_layers_linear(X, Y, Z, specs)
    for l in lds[:-1]:
        l["q_sqrt"] = l["q_sqrt"] * 1e-5
    sl, state = OM.state_from_layers(lds, lik_variance=1.0)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl)
    rng = np.random.default_rng(0)
    m = {k: np.zeros_like(v) for k, v in state.items()}
    v = {k: np.zeros_like(vv) for k, vv in state.items()}
    S, mb = cfg["S"], cfg["mb"]
    t0 = time.perf_counter()
    steps = 0
    while True:
        idx = rng.permutation(X.shape[0])[:mb]
        zs = [rng.standard_normal((S, mb, cfg["D"])), rng.standard_normal((S, mb, cfg["D"])), np.zeros((1, 1, 1))]
        _, g = OM.elbo_and_grad(spec, state, X[idx], Y[idx], zs, S, num_data=X.shape[0])
        for k in state:
            O.adam_step(state[k], -g[k], m[k], v[k], steps + 1)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 50:
            break
    return dict(value=steps / el, unit="steps/s", cores=int(torch.get_num_threads()), kind="port",
                sample=f"{steps} full training steps of the same workload (numpy/torch-CPU fp64 restatement of the "
                       f"GPflow/TF op sequence incl. autograd + Adam; not TF itself), {el:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    cfg = CFG
    model, X, Y, Z = build_model(cfg, rank, world)
    eng = model.engine()
    ctx = eng.ctx

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.train_step(0.01)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step(0.01)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elbo = model.train_step(0.01, sync=True)

    # secondary metrics (untimed region of the contract): forward-only ELBO evals/s and predict_f rows/s
    n_ev = 50
    Xb, Yb = model.next_minibatch()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(n_ev):
        eng.elbo(Xb, Yb, cfg["S"], seed=i, data_scale=1.0, with_grad=False, sync=False)
    torch.cuda.synchronize()
    evals_per_s = n_ev / (time.perf_counter() - t1)
    Xs = ctx.to_device(X[:1000])
    eng.propagate(Xs, 100, seed=1, want=("mean", "var"))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(20):
        eng.propagate(Xs, 100, seed=i, want=("mean", "var"))
    torch.cuda.synchronize()
    predict_rows_per_s = 20 * 100 * 1000 / (time.perf_counter() - t1)

    # per-kernel HIP-event timing on the launch streams (separate loop: events perturb the pipeline slightly); the
    # wgrad/backward side-stream overlap is switched off here so that each duration is the kernel's own
    os.environ["DSDGP_NO_OVERLAP"] = "1"
    ctx.prof_enable(True)
    nprof = 20
    for _ in range(nprof):
        model.train_step(0.01)
    torch.cuda.synchronize()
    prof = {}
    for name in ("layer_fwd", "layer_bwd", "wgrad", "gemm", "potrf"):
        ms, cnt = ctx.prof_read(name)
        prof[name] = dict(ms_per_step=ms / nprof, launches_per_step=cnt / nprof)
    ctx.prof_enable(False)
    os.environ["DSDGP_NO_OVERLAP"] = "0"
    fl = algorithmic_flops(cfg)
    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.md: 2*FETCH_SIZE + WRITE_SIZE,
    # same command, gfx950 correction of MI355X_MICROARCH.md); None when the profile is not shipped
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)
        for name, key in (("layer_fwd", "k_layer_fwd_sm"), ("layer_bwd", "k_layer_bwd_sm"), ("wgrad", "k_wgrad<4, 4>")):
            hit = [v for k, v in pmc.items() if k.startswith(key)]
            if hit:
                traffic[name] = round(hit[0]["traffic_MB_per_launch"] * 1e6)
    except Exception:
        pass
    roof_all = {}
    for name in ("layer_fwd", "layer_bwd", "wgrad"):
        ms = prof[name]["ms_per_step"]
        ach = fl[name] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof_all[name] = dict(bound="mfma", achieved=round(ach, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                              frac=round(ach / FP64_MFMA_PEAK_TFLOPS, 4), traffic=traffic.get(name),
                              ms_per_step=round(ms, 4), launches_per_step=prof[name]["launches_per_step"],
                              algorithmic_gflop_per_step=round(fl[name] / 1e9, 3))
    dominant = max(roof_all, key=lambda k: roof_all[k]["ms_per_step"])
    roofline = dict(roof_all[dominant], kernel=dominant)

    if rank == 0:
        steps_per_s = args.steps / dt
        out = {
            "metric": "ELBO-steps/sec", "value": round(steps_per_s * world, 3), "unit": "steps/s (minibatch-1000 ELBO+grad+Adam steps, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-layer DS-DGP, kin8nm-shaped (7372x8), RBF, M=128, S=20, minibatch=1000 per GPU, "
                                   "fp64, white=False (BASELINE.json configs[1])",
                       "per_gpu_minibatch": cfg["mb"], "global_batch": cfg["mb"] * world, "num_samples": cfg["S"],
                       "inducing": cfg["M"], "layers": cfg["L"], "parallelism": f"row-sharded dp{world}"},
            "roofline": roofline, "roofline_all": roof_all, "kernel_ms_per_step": prof,
            "step_fraction_of_fp64_peak": round(fl["step"] * steps_per_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
            "elbo_evals_per_s": round(evals_per_s, 2), "predict_f_rows_per_s": round(predict_rows_per_s, 1),
            "final_elbo": elbo,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, X, Y, Z)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
