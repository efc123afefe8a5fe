#!/usr/bin/env python
"""Headline benchmark: ELBO-steps/sec of the 3-layer doubly-stochastic DGP (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = minibatch gather + forward ELBO (dgp.py:92-98) + reverse-mode gradient w.r.t. every trainable parameter +
Adam(0.01) update — i.e. one `session.run(opt_op)` of demos/demo_regression_UCI.ipynb:324 — on kin8nm-SHAPED synthetic
data (real UCI data cannot be downloaded): N_data=7372, D=8, widths 8->8->8->1, RBF, M=128, S=20, minibatch 1000, fp64,
white=False, inner q_sqrt * 1e-5.  Inputs are resident in HBM before the timed region.

N>1: one process per GPU (torch.distributed / RCCL).  Default `--scaling strong` (BASELINE's metric at N GPUs): the GLOBAL
minibatch stays 1000 rows x S=20, every rank takes 1000/N rows x all S samples (rows are independent through all layers), the
flat gradient is summed with one all-reduce per step and every rank applies the same Adam update; value = global steps/s
(NO x N).  `--scaling weak` keeps 1000 rows per GPU (global batch 1000 N) and reports N x steps/s.

Timing protocol: the secondary measurements (forward-only evals/s, predict_f rows/s, per-kernel HIP-event times, the
sub-rooflines, the other configs) run FIRST and bring the GPU to its steady clocks; then W untimed warm-up steps, then EXACTLY K
steps between barrier + synchronize -> `value`.  After the timed region: the CPU baseline (rank 0, N = 1), a fixed loop of 1200
single steps bracketed by HIP events (median / p10 / p90 -> `step_time`), and at N = 1 the flat data-parallel step on the 500 / 250 /
125-row shards of a 2 / 4 / 8-GPU strong-scaling run over a one-rank RCCL group (`shard_steps`: the strong-scaling ceiling).

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


def algorithmic_flops(cfg, shapes=None):
    """SURVEY §8d contract figures, per kernel class and per step (non-white, w=2)."""
    M = cfg["M"]
    fwd, wg = [], []
    shapes = shapes or layer_shapes(cfg)
    for R, Din, Dout in shapes:
        f = M * R * (2 * Din + 4) + 2 * M * M * R + 2 * M * R * Dout + 2 * M * R + Dout * (M * M * R + 2 * M * R) + 4 * R * Dout
        fwd.append(f)
        wg.append((2 + Dout) * M * M * R)
    small = sum(M * M * (Din + 2) + M ** 3 / 3 + Dout * M ** 3 / 3 + 2 * M * M * Dout for _, Din, Dout in shapes)
    return dict(layer_fwd=sum(fwd), layer_bwd=sum(fwd), wgrad=sum(wg), small=small,
                step=3 * (sum(fwd) + small), per_layer=fwd)


def slot_flops(fl, fx, fused_last):
    """Algorithmic / executed flops of the timed kernel classes.  With the fused last-layer launch (csrc/layer_last.hip: forward chain +
    likelihood + reverse pass of the D_out = 1 layer in one kernel) the `layer_fwd` / `layer_bwd` slots hold the layers below it and
    `layer_last` both halves of the last layer (SURVEY 8d counts the reverse pass as 1 x the forward figure)."""
    if not fused_last:
        return ({k: fl[k] for k in ("layer_fwd", "layer_bwd", "wgrad")}, {k: fx[k] for k in ("layer_fwd", "layer_bwd", "wgrad")})
    last = fl["per_layer"][-1]
    a = dict(layer_fwd=fl["layer_fwd"] - last, layer_bwd=fl["layer_bwd"] - last, layer_last=2 * last, wgrad=fl["wgrad"])
    x = dict(layer_fwd=fx["layer_fwd"] - fx["last_fwd"], layer_bwd=fx["layer_bwd"] - fx["last_bwd"], layer_last=fx["last_fused"],
             wgrad=fx["wgrad"])
    return a, x


def chain_d_split(nblk, D_out):
    """csrc/model_types.hpp chain_d_split: workgroups per row block of a forward-chain launch with few row blocks"""
    ds = 1
    if nblk < 256:
        ds = min(4, max(1, 512 // nblk))
    elif nblk < 512:
        ds = min(1024 // nblk, D_out // 5)
    return max(1, min(ds, D_out))


def executed_flops_chain(M, shapes, white=False):
    """EXECUTED v_mfma_f64_16x16x4_f64 flops (2048 per instruction) of the fused chains and the split-K weight-gradient launch at
    padded inducing counts <= 256, counted from the loop structure of csrc/layer_sm_impl.hpp / wgrad.hip: what the pipe
    actually issues, beside SURVEY 8d's ALGORITHMIC count — the backward d-loop is dense (2 M^2 per row and output where the contract
    counts M^2), triangular products run at 16-row granularity, a d-split forward launch repeats its prologue per workgroup, the
    symmetric P_d skip their upper blocks and the alg_g layers have no E A^T product."""
    Mp = -(-M // 16) * 16 if M <= 128 else -(-M // 32) * 32
    nb = Mp // 16
    tri = 4 * nb * (nb + 1) // 2
    fwd = bwd = wg = 0
    last_fwd = last_bwd = last_fused = 0
    for li, (R, Din, Dout) in enumerate(shapes):
        blocks = -(-R // 16)
        k16 = -(-Din // 16)
        nw4 = Mp <= 128 and blocks > 160          # the 4-wave forward instance (8 waves: Mp = 256, and launches of <= 160 row blocks)
        sq = 4 * nb * k16
        pro = sq + tri + (0 if white else tri) + (4 * nb * -(-Dout // 16) if nw4 else 0)
        f_l = blocks * (chain_d_split(blocks, Dout) * pro + Dout * tri)
        fwd += f_l
        dp4 = -(-Dout // 4) * 4
        b_l = blocks * (Dout * 4 * nb * nb + (dp4 // 4) * nb + (tri if white else 4 * nb * nb) + sq + 8 * nb * k16)
        bwd += b_l
        if li == len(shapes) - 1:
            # the fused last-layer launch (layer_last.hip, D_out = 1): distances, a1, a, c = q_sqrt^T a | q_sqrt c (triangular, where the
            # two chains run the dense S a), Ku^-1 abar (dense), distances again, the hyper-parameter / dX sums; mean and q_mu mbar on the VALU
            last_fwd, last_bwd = f_l, b_l
            last_fused = blocks * (sq + 3 * tri + tri + 4 * nb * nb + sq + 8 * nb * k16)
        Mw = -(-Mp // 64) * 64
        ti = Mw // 64
        alg_g = 4 * Dout * Mp <= R
        dp16, dinp16 = -(-Dout // 16) * 16, -(-Din // 16) * 16
        per_chunk = Dout * (64 * ti * (ti - 1) // 2 + 40 * ti) + ti * dp16 + ti * dinp16 + (0 if alg_g else 64 * ti * ti)
        wg += blocks * per_chunk
    return dict(layer_fwd=2048.0 * fwd, layer_bwd=2048.0 * bwd, wgrad=2048.0 * wg, last_fwd=2048.0 * last_fwd, last_bwd=2048.0 * last_bwd,
                last_fused=2048.0 * last_fused)


def executed_profile():
    """executed MFMA flops per step of every config shape from the committed PMC pass (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 summed over
    one step's launches, tools/executed_flops.py) — only when the profile was taken from the kernel sources of the loaded library"""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_executed_flops.json")) as f:
            ex = json.load(f)
        if ex.get("csrc_sha256_16") != csrc_hash():
            return None, "profiles/r06_executed_flops.json was taken from other kernel sources than this build: not reported"
        return ex, "profiles/r06_executed_flops.json (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 x 512, serial schedule, separate run)"
    except Exception:
        return None, None


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


def build_model(cfg, rank, world, mb_local, bucketed=None):
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian
    X, Y = make_synthetic(cfg["n_data"], cfg["D"], seed=0)
    Z = default_Z(X, cfg["M"], seed=0)
    kernels = [RBF(cfg["D"]) for _ in range(cfg["L"])]
    model = DGP(X, Y, Z, kernels, Gaussian(), num_samples=cfg["S"], minibatch_size=mb_local)
    for layer in model.layers[:-1]:
        layer.q_sqrt = layer.q_sqrt.value * 1e-5                      # demo_regression_UCI.ipynb:183
    if world > 1:
        from doubly_stochastic_dgp.distributed import attach
        attach(model, rank, world, bucketed=bucketed)      # None: by gradient size (distributed.BUCKET_MIN_BYTES: flat at config 2)
    return model, X, Y, Z


def cpu_baseline(cfg, X, Y, Z, budget_s=10.0):
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


def sub_rooflines(ctx):
    """The two sub-rooflines north_star names besides the fused chain:
    * Gram build (a MATERIALISED Kuf, layers.py:184, dsdgp_gram): HBM-bound, algorithmic bytes 8 n n2 + 8 D (n + n2), against
      8 TB/s; HIP-event timed here, FETCH/WRITE_SIZE counters of the same launches in profiles/r02_gram_pmc.md;
    * Cholesky + triangular inverse of Kuu at M = 1024 (layers.py:172,186-188: the multi-workgroup blocked path),
      2 M^3 / 3 flops against the 78.6 TFLOP/s fp64 MFMA peak."""
    import ctypes as C
    from doubly_stochastic_dgp import _lib
    rng = np.random.default_rng(1)
    out = {"gram": [], "potrf_trtri": []}
    ls = np.ones(1)
    ctx.prof_enable(True)
    for (M, R, D) in ((128, 20000, 8), (256, 40000, 9), (512, 40960, 30), (1024, 50000, 8)):
        Z, X = ctx.to_device(rng.standard_normal((M, D))), ctx.to_device(rng.standard_normal((R, D)))
        o = ctx.empty(M, R)
        spec = _lib.KernelSpec(kind=0, input_dim=D, ard=0, has_white=0, variance=1.0, white_variance=0.0,
                               lengthscales=ls.ctypes.data_as(_lib.c_double_p))

        def call():
            _lib.check(ctx.lib.dsdgp_gram(ctx.handle, C.byref(spec), C.c_void_p(Z.data_ptr()), M, C.c_void_p(X.data_ptr()), R,
                                          0.0, C.c_void_p(o.data_ptr()), R))
        for _ in range(3):
            call()
        ctx.prof_read("gram")
        for _ in range(20):
            call()
        ms, cnt = ctx.prof_read("gram")        # HIP events around every launch on the launch stream (ProfScope), mean of 20
        nbytes = 8 * M * R + 8 * D * (M + R)
        gbs = nbytes / (ms / cnt * 1e-3) / 1e9
        out["gram"].append(dict(n=M, n2=R, D=D, us_per_launch=round(1e3 * ms / cnt, 2), algorithmic_MB=round(nbytes / 1e6, 1),
                                bound="hbm", achieved=round(gbs, 1), peak=8000.0, unit="GB/s", frac=round(gbs / 8000.0, 4)))
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian
    for M in (128, 1024):
        Xs = rng.standard_normal((M + 64, 8))
        m1 = DGP(Xs, Xs[:, :1], Xs[:M] + 0.01 * rng.standard_normal((M, 8)), [RBF(8)], Gaussian(), num_samples=1)
        e1 = m1.engine()
        e1.prepare()
        ctx.prof_read("potrf")
        reps = 5
        for _ in range(reps):
            _lib.check(e1.lib.dsdgp_model_theta_changed(e1.model))     # else the unchanged Ku keeps its factor (dsdgp_model_track_theta)
            e1._needs_prepare = True
            e1.prepare()
        ms, cnt = ctx.prof_read("potrf")
        fl = 2.0 * M ** 3 / 3.0
        tf = fl / (max(ms, 1e-9) / reps * 1e-3) / 1e12
        out["potrf_trtri"].append(dict(n=M, us=round(1e3 * ms / reps, 1), algorithmic_gflop=round(fl / 1e9, 3), bound="mfma",
                                       achieved=round(tf, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                       frac=round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                       note="one matrix; look-ahead launch sequence (8 factor launches carrying the wide updates and the "
                                            "inverse's block rows, 7 panel launches), latency-bound" if M >= 512 else
                                            "one LDS-resident workgroup; latency-bound"))
    ctx.prof_enable(False)
    # K4: the blocked triangular solve at the two right-hand-side shapes north_star names (Kuu^-1/2 Kuf: M x S N), n^2 nrhs flops
    import torch
    out["trsm"] = []
    for (M, R) in ((128, 20000), (1024, 50000)):
        L = np.tril(rng.standard_normal((M, M))) / np.sqrt(M) + 2.0 * np.eye(M)
        dL, dB = ctx.to_device(L), ctx.to_device(rng.standard_normal((M, R)))

        def solve():
            _lib.check(ctx.lib.dsdgp_trsm(ctx.handle, 0, M, R, C.c_void_p(dL.data_ptr()), M, C.c_void_p(dB.data_ptr()), R))
        for _ in range(2):
            solve()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # ctx stream == torch's current stream
        reps = 10
        e0.record()
        for _ in range(reps):
            solve()                 # (in place: the iterates shrink towards 0, the flop count does not depend on the values)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        fl = float(M) * M * R
        tf = fl / (us * 1e-6) / 1e12
        out["trsm"].append(dict(n=M, nrhs=R, us=round(us, 1), algorithmic_gflop=round(fl / 1e9, 3), bound="mfma", achieved=round(tf, 3),
                                peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                note="16 x 16 diagonal inverses + 128-row panels (one wave per 16 columns) + MFMA GEMM updates"
                                     + ("; left-looking: one LDS-tiled product over the solved rows per panel" if M >= 256 and R >= 2048 else "")))
    return out


def all_configs():
    """Throughput of the OTHER BASELINE.json configs on this one GPU (configs[0], [2] whole; [3], [4] as the per-GPU shard of their
    8-GPU minibatch — tools/bench_configs.py): two passes of >= 20 steps each (the faster one reported), every step bracketed by HIP events on the launch stream (median / p10 /
    p90) next to the wall-clock rate, kernel launches per step (dsdgp_launch_count), the fraction of the fp64 MFMA peak by SURVEY 8d's
    F_step of that shape and — from the committed PMC pass — by the flops the MFMA pipe executed.  Secondary lines of the N = 1 JSON;
    the contract's `value` stays configs[1]."""
    import torch
    from doubly_stochastic_dgp import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as BC
    ex, ex_src = executed_profile()
    lib = _lib.load()
    out = []
    for i, c in enumerate(BC.CONFIGS):
        if i == 1:
            continue
        nsteps = max(20, c["steps"])
        model, step = BC._build(c)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        # two passes of nsteps steps, the faster one reported (the launch-bound config 1 is host-paced: one stall of the host thread in
        # a 30 ms pass halves its rate)
        dt, ts = None, None
        for _rep in range(2):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
            l0 = lib.dsdgp_launch_count()
            t0 = time.perf_counter()
            evs[0].record()
            for k in range(nsteps):
                step()
                evs[k + 1].record()
            torch.cuda.synchronize()
            dt_rep = time.perf_counter() - t0
            launches = (lib.dsdgp_launch_count() - l0) / nsteps
            if dt is None or dt_rep < dt:
                dt = dt_rep
                ts = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(nsteps)])
        elbo = model.train_step(0.01, sync=True)
        widths = c["widths"]
        douts = list(widths[1:]) + [c.get("classes") or 1]
        shapes = [((c["mb"] if l == 0 else c["mb"] * c["S"]), widths[l], douts[l]) for l in range(len(widths))]
        fstep = algorithmic_flops(dict(M=c["M"]), shapes)["step"]
        sps = nsteps / dt
        tf = fstep * sps / 1e12
        key = f"cfg{i + 1}"
        exg = ex["configs"][key]["executed_gflop_per_step"] if ex and key in ex.get("configs", {}) else None
        out.append(dict(config=c["name"], steps=nsteps, steps_per_s=round(sps, 3), ms_per_step=round(1e3 * dt / nsteps, 3),
                        step_time=dict(median_ms=round(float(np.median(ts)), 4), p10_ms=round(float(np.percentile(ts, 10)), 4),
                                       p90_ms=round(float(np.percentile(ts, 90)), 4)),
                        launches_per_step=round(launches, 1),
                        algorithmic_gflop_per_step=round(fstep / 1e9, 1), achieved_tflops=round(tf, 2),
                        frac_of_fp64_peak=round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                        executed_gflop_per_step=exg,
                        frac_executed=None if exg is None else round(exg * 1e9 * sps / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                        executed_source=ex_src, elbo_finite=bool(np.isfinite(elbo)),
                        note=("one optimiser step = natural-gradient evaluation + step on the last layer, then an Adam step: two forward "
                              "passes, F_step counts one" if c.get("natgrad") else None)))
        del model, step
        torch.cuda.empty_cache()
    return out


def shard_steps(cfg):
    """Strong-scaling ceiling measured on ONE GPU: the flat data-parallel training step (ELBO + gradient, one all-reduce of the whole
    gradient buffer, Adam — doubly_stochastic_dgp.distributed) on the per-rank shards of the global 1000-row minibatch at N = 2 / 4 / 8,
    over a ONE-RANK RCCL group: the collective is an identity, everything around it is timed.  What an N-GPU run adds is the
    all-reduce's own latency (2.3 MB over xGMI).  200 steps, best of three."""
    import socket
    import torch
    import torch.distributed as dist
    from doubly_stochastic_dgp.distributed import attach
    created = False
    out = []
    try:
        if not dist.is_initialized():
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
            created = True
        for world in (2, 4, 8):
            rows = cfg["mb"] // world
            model, _, _, _ = build_model(cfg, 0, 1, rows)
            attach(model, 0, 1, bucketed=False)
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
            out.append(dict(n_gpus_modelled=world, rows_per_rank=rows, ms_per_step_flat_1rank_rccl=round(min(reps), 4),
                            steps_per_s_ceiling=round(1e3 / min(reps), 1)))
            del model
            torch.cuda.empty_cache()
    except Exception as e:                      # a secondary line: never takes the contract's JSON line down
        out.append(dict(error=f"{type(e).__name__}: {e}"))
    finally:
        if created:
            dist.destroy_process_group()
    return out


def csrc_hash():
    """sha256 over the kernel sources: ties a committed PMC traffic profile to the build it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "doubly-stochastic-dgp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) or name == "Makefile":
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def self_launch(n):
    """Re-run this command line as `n` ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>.  Returns the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this platform
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = global minibatch fixed at 1000 (BASELINE's metric), weak = 1000 rows per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (profiler runs)")
    ap.add_argument("--allreduce", choices=["auto", "flat", "bucketed"], default="auto",
                    help="N > 1: one flat gradient all-reduce per step, or one per layer from inside the reverse pass; auto = by gradient size")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher of its own N ranks (one process per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1 at a free port) and passes their exit status on; rank 0 prints the JSON line
        sys.exit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it spawns its own ranks) or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    one_device = os.environ.get("DSDGP_BENCH_ONE_DEVICE") == "1"
    if world > 1 and not one_device and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, this node shows {torch.cuda.device_count()} "
                         f"(one process per GPU over RCCL; no oversubscription outside the DSDGP_BENCH_ONE_DEVICE test rig)")
    # test rig only (validating the N > 1 code path on a 1-GPU box): DSDGP_BENCH_BACKEND=gloo + DSDGP_BENCH_ONE_DEVICE=1 put
    # every rank on cuda:0 and stage the all-reduce through the host; the driver's runs use RCCL, one GPU per rank
    backend = os.environ.get("DSDGP_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = dict(CFG)
    if args.scaling == "strong" and world > 1:
        if cfg["mb"] % world:
            raise SystemExit(f"strong scaling needs the global minibatch {cfg['mb']} divisible by {world}")
        mb_local = cfg["mb"] // world
    else:
        mb_local = cfg["mb"]
    model, X, Y, Z = build_model(cfg, rank, world, mb_local, {"auto": None, "flat": False, "bucketed": True}[args.allreduce])
    eng = model.engine()
    ctx = eng.ctx
    cfg_local = dict(cfg, mb=mb_local)        # per-GPU shard: what one launch of this rank processes

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(3):
        model.train_step(0.01)
    torch.cuda.synchronize()

    evals_per_s = predict_rows_per_s = predict_fwd = None
    prof = {}
    sub = {}
    others = None
    steady = None
    if not args.no_extras:
        # secondary metrics: forward-only ELBO evals/s and predict_f rows/s (per rank)
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
        # the forward chains of predict_f alone (HIP events on the launch stream): roofline entry of the second named metric
        ctx.prof_enable(True)
        ctx.prof_read("layer_fwd")
        for i in range(10):
            eng.propagate(Xs, 100, seed=i, want=("mean", "var"))
        torch.cuda.synchronize()
        pf_ms, pf_cnt = ctx.prof_read("layer_fwd")
        ctx.prof_enable(False)
        predict_fwd = dict(ms_per_call=pf_ms / 10, launches_per_call=pf_cnt / 10)

        # per-kernel HIP-event timing on the launch streams (events perturb the pipeline slightly, hence a separate loop); the
        # wgrad/backward side-stream overlap is switched off here so that each duration is the kernel's own
        os.environ["DSDGP_NO_OVERLAP"] = "1"
        ctx.prof_enable(True)
        nprof = 20
        for _ in range(nprof):
            model.train_step(0.01)
        torch.cuda.synchronize()
        for name in ("layer_fwd", "layer_bwd", "layer_last", "wgrad", "gemm", "potrf"):
            ms, cnt = ctx.prof_read(name)
            prof[name] = dict(ms_per_step=ms / nprof, launches_per_step=cnt / nprof)
        prof["layer_last"]["note"] = ("forward chain + Gaussian likelihood + reverse pass of the last (D_out = 1) layer in one launch "
                                      "(csrc/layer_last.hip); 0 launches: the two chains ran instead")
        prof["potrf"]["note"] = ("the fused head launch k_head: parameter transforms + Ku + Cholesky + inverse factor + N(0,1) draws + "
                                 "minibatch gather")
        ctx.prof_enable(False)
        os.environ["DSDGP_NO_OVERLAP"] = "0"
        if rank == 0:
            sub = sub_rooflines(ctx)
            if world == 1:
                others = all_configs()
    # ---- the contract's timed region: W warm-up steps, then EXACTLY K steps
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
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elbo = model.train_step(0.01, sync=True)
    from doubly_stochastic_dgp import _lib
    lib = _lib.load()
    l0 = lib.dsdgp_launch_count()
    for _ in range(10):
        model.train_step(0.01)
    torch.cuda.synchronize()
    launches_per_step = (lib.dsdgp_launch_count() - l0) / 10.0
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(cfg, X, Y, Z)
    if not args.no_extras:
        # steady-state distribution: single steps bracketed by events on the launch stream (ctx stream == torch's current stream),
        # a fixed sample of four batches of 300
        nb_ev, n_batches = 300, 4
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nb_ev + 1)]
        for _ in range(20):
            model.train_step(0.01)
        all_ts = []
        t_s0 = time.perf_counter()
        while True:
            evs[0].record()
            for i in range(nb_ev):
                model.train_step(0.01)
                evs[i + 1].record()
            torch.cuda.synchronize()
            all_ts.append(np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(nb_ev)]))
            if len(all_ts) >= n_batches:
                break
        ts = np.concatenate(all_ts)
        steady = dict(n=int(ts.size), median_ms=round(float(np.median(ts)), 4), p10_ms=round(float(np.percentile(ts, 10)), 4),
                      p90_ms=round(float(np.percentile(ts, 90)), 4), mean_ms=round(float(ts.mean()), 4),
                      wall_s=round(time.perf_counter() - t_s0, 2),
                      note="per-step HIP events on the launch stream, run after the timed region (and after the CPU baseline at N = 1); "
                           "the gradient all-reduce (N>1) is inside each step")
    shards = None
    if rank == 0 and world == 1 and not args.no_extras:
        shards = shard_steps(cfg)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        rccl_ranks = dist.get_world_size()

    fl = algorithmic_flops(cfg_local)
    fl_global = algorithmic_flops(cfg)
    # HBM bytes per launch: NOT measured in this process (PMC counters need rocprofv3) — read from the committed summary of
    # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this very command (2*FETCH + WRITE, the gfx950
    # correction of MI355X_MICROARCH.md) and labelled with its source; null when no current profile is shipped
    traffic, traffic_source = {}, None
    for cand in ("r06_pmc_traffic.json",):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                pmc = json.load(f)
            if pmc.pop("csrc_sha256_16", None) != csrc_hash():
                traffic_source = f"profiles/{cand} was taken from other kernel sources than this build: traffic not reported"
                break
            for name, key in (("layer_fwd", "k_layer_fwd_sm"), ("layer_bwd", "k_layer_bwd_sm"), ("layer_last", "k_layer_last"), ("wgrad", "k_wgrad")):
                hit = [v for k, v in pmc.items() if k.startswith(key)]
                if hit:
                    n_l = sum(h["launches"] for h in hit)
                    traffic[name] = round(sum(h["traffic_MB_per_launch"] * h["launches"] for h in hit) / n_l * 1e6)
            traffic_source = f"profiles/{cand} (separate rocprofv3 --pmc passes of this command, not measured in this run)"
            break
        except Exception:
            pass
    roof_all = {}
    fx = executed_flops_chain(cfg["M"], layer_shapes(cfg_local))
    fused_last = prof.get("layer_last", {}).get("launches_per_step", 0) > 0
    fl_slot, fx_slot = slot_flops(fl, fx, fused_last)
    for name in ("layer_fwd", "layer_bwd", "layer_last", "wgrad"):
        if name not in prof or name not in fl_slot:
            continue
        ms = prof[name]["ms_per_step"]
        ach = fl_slot[name] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        achx = fx_slot[name] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof_all[name] = dict(bound="mfma", achieved=round(ach, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                              frac=round(ach / FP64_MFMA_PEAK_TFLOPS, 4), traffic=traffic.get(name), traffic_source=traffic_source,
                              ms_per_step=round(ms, 4), launches_per_step=prof[name]["launches_per_step"],
                              algorithmic_gflop_per_step=round(fl_slot[name] / 1e9, 3),
                              executed_gflop_per_step=round(fx_slot[name] / 1e9, 3), frac_executed=round(achx / FP64_MFMA_PEAK_TFLOPS, 4),
                              executed_note="MFMA instructions the launch issues x 2048 flops, counted from the kernels' loop structure "
                                            "(bench.executed_flops_chain): dense backward d-loop, 16-row triangular granularity, "
                                            "symmetric / alg_g savings of the weight-gradient products")
    if predict_fwd is not None:
        # predict_f(S = 100, 1000 test points): forward chains only, layer 0 on 1000 rows, layers 1.. on 100 000
        cfgp = dict(cfg, S=100, mb=1000)
        flp = algorithmic_flops(cfgp)["layer_fwd"]
        fxp = executed_flops_chain(cfg["M"], layer_shapes(cfgp), white=True)["layer_fwd"]
        ach = flp / (predict_fwd["ms_per_call"] * 1e-3) / 1e12
        roof_all["predict_f_fwd"] = dict(bound="mfma", achieved=round(ach, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                         frac=round(ach / FP64_MFMA_PEAK_TFLOPS, 4), traffic=None, traffic_source=None,
                                         ms_per_step=round(predict_fwd["ms_per_call"], 4),
                                         launches_per_step=predict_fwd["launches_per_call"],
                                         algorithmic_gflop_per_step=round(flp / 1e9, 3),
                                         executed_gflop_per_step=round(fxp / 1e9, 3),
                                         frac_executed=round(fxp / (predict_fwd["ms_per_call"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                                         executed_note="whitened forward-only chains: one triangular product less per row block")
    roofline = None
    if roof_all:
        dominant = max((k for k in roof_all if k != "predict_f_fwd"), key=lambda k: roof_all[k]["ms_per_step"])
        roofline = dict(roof_all[dominant], kernel=dominant)

    if rank == 0:
        steps_per_s = args.steps / dt
        ex, ex_src = executed_profile()
        step_ex = ex["configs"]["cfg2"]["executed_gflop_per_step"] if (ex and world == 1 and "cfg2" in ex.get("configs", {})) else None
        if step_ex is None:       # chains + weight-gradient products only (the M x M algebra and the head launch are ~3 % more at this shape)
            step_ex = round(sum(fx_slot.values()) / 1e9, 3)
            ex_src = "static count of the chain + weight-gradient launches (bench.executed_flops_chain); M x M algebra and head not included"
        step_ex_frac = round(step_ex * 1e9 * steps_per_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)
        weak = args.scaling == "weak" and world > 1
        value = steps_per_s * (world if weak else 1)
        out = {
            "metric": "ELBO-steps/sec", "value": round(value, 3),
            "unit": ("steps/s (minibatch-1000-per-GPU ELBO+grad+Adam steps x GPUs, whole job)" if weak else
                     "steps/s (global-minibatch-1000 ELBO+grad+Adam steps, whole job)"),
            "n_gpus": world, "rccl_ranks": rccl_ranks, "visible_gpus": torch.cuda.device_count(), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "3-layer DS-DGP, kin8nm-shaped (7372x8), RBF, M=128, S=20, global minibatch "
                                   f"{mb_local * world} ({mb_local} rows per GPU), fp64, white=False (BASELINE.json configs[1])",
                       "per_gpu_minibatch": mb_local, "global_batch": mb_local * world, "num_samples": cfg["S"],
                       "inducing": cfg["M"], "layers": cfg["L"], "parallelism": f"row-sharded dp{world}",
                       "gradient_exchange": ("none" if world == 1 else
                                             ("one all-reduce per layer (bucketed)" if model._dist_buckets()["on"] else "one flat all-reduce"))},
            "roofline": roofline, "roofline_all": roof_all, "sub_rooflines": sub, "all_configs": others, "kernel_ms_per_step": prof,
            "step_time": steady, "launches_per_step": launches_per_step,
            "shard_steps": shards,
            "shard_steps_note": "flat data-parallel step of the 500 / 250 / 125-row shards (N = 2 / 4 / 8 of the global 1000-row minibatch) on "
                                "this one GPU over a one-rank RCCL group: an N-GPU strong-scaling run cannot beat 1 / (this + the 2.3 MB "
                                "all-reduce's latency)",
            "step_fraction_of_fp64_peak": round(algorithmic_flops(dict(cfg, mb=mb_local * world))["step"] * steps_per_s / world
                                                / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
            "step_executed_gflop": step_ex, "step_fraction_executed": step_ex_frac, "step_executed_source": ex_src,
            "elbo_evals_per_s": None if evals_per_s is None else round(evals_per_s, 2),
            "elbo_evals_note": "forward-only ELBO at FIXED parameters: the factorisation of Ku and the parameter-side products are "
                               "kept between evaluations (dsdgp_model_track_theta) and the chains run in whitened coordinates; the "
                               "training step above refactorises every step",
            "predict_f_rows_per_s": None if predict_rows_per_s is None else round(predict_rows_per_s, 1),
            "final_elbo": elbo,
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
