"""world_size = 2 on ONE GPU: two processes share cuda:0 and run the PRODUCT data-parallel path — attach(), the flat
[gradient | elbo, data, KL, info] all-reduce of Engine.gradbuf (over gloo here: RCCL needs one device per rank) and Adam on the
HIP engine — for three steps on disjoint row shards of one minibatch; the parameters must equal the single-process run on the
whole minibatch (dgp.py:92-98: KL counted once, data term scaled by num_data / N_global)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    rng = np.random.RandomState(3)
    N, D, M, S = 96, 4, 24, 3
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M] + 0.01 * rng.randn(M, D)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    return X, Y, Z, zs, S, D


def _model(X, Y, Z, S, D):
    return make_case(X, Y, Z, [kern_spec("rbf", D, 1.1, 0.9), kern_spec("matern52", D, 0.8, 1.2)], S=S, num_data=5000, seed=1)[2]


def _worker(rank, world, port, out_path, bucketed=False):
    import torch
    import torch.distributed as dist
    from doubly_stochastic_dgp.distributed import attach
    from doubly_stochastic_dgp.engine import Context
    Context.get()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Y, Z, zs, S, D = _problem()
    n_local = X.shape[0] // world
    sl = slice(rank * n_local, (rank + 1) * n_local)
    model = _model(X, Y, Z, S, D)
    attach(model, rank, world, bucketed=bucketed)
    zl = [z[:, sl, :] for z in zs]
    elbo = model._build_likelihood(X[sl], Y[sl], zs=zl, with_grad=True)       # all-reduced value: the GLOBAL elbo
    out = model.engine().out4.cpu().numpy().copy()
    for _ in range(3):
        model.train_step(0.01, X=X[sl], Y=Y[sl], zs=zl)
    last = model.train_step(0.01, X=X[sl], Y=Y[sl], zs=zl, sync=True)
    if rank == 0:
        eng = model.engine()
        eng.ctx.sync()
        np.savez(out_path, elbo=elbo, out=out, last=last, q_mu0=model.layers[0].q_mu.value, q_sqrt1=model.layers[1].q_sqrt.value,
                 Z0=model.layers[0].feature.Z.value, lik=model.likelihood.likelihood.variance.value,
                 theta=eng.theta.cpu().numpy(), grad=eng.gradbuf.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(tmp_path, bucketed):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / f"rank0_{int(bucketed)}.npz")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "doubly-stochastic-dgp_amd")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", str(port), out_path, str(int(bucketed))], env=env,
                              cwd=ROOT) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=500) == 0
    return np.load(out_path)


@pytest.mark.timeout(900)
def test_bucketed_exchange_equals_flat_exchange_bitwise(tmp_path):
    """One all-reduce per layer issued from inside the reverse pass (dsdgp_model_set_bucket_callback) against the single flat
    all-reduce after it: identical parameters, gradient buffer and result scalars after four optimiser steps (two ranks: a + b)."""
    flat, buck = _run_two_ranks(tmp_path, False), _run_two_ranks(tmp_path, True)
    for k in ("theta", "grad", "out", "elbo", "last"):
        assert np.array_equal(flat[k], buck[k]), k


@pytest.mark.timeout(600)
def test_two_ranks_equal_single_process(tmp_path):
    got = _run_two_ranks(tmp_path, True)
    X, Y, Z, zs, S, D = _problem()
    ref = _model(X, Y, Z, S, D)
    e = ref._build_likelihood(X, Y, zs=zs, with_grad=True)
    o = ref.engine().out4.cpu().numpy()
    assert_allclose(got["elbo"], e, rtol=1e-12)
    assert_allclose(got["out"][:3], o[:3], rtol=1e-12)          # elbo, data term, KL (weighted 1/world per rank) sum to the global values
    assert got["out"][3] == 0.0
    for _ in range(3):
        ref.train_step(0.01, X=X, Y=Y, zs=zs)
    last = ref.train_step(0.01, X=X, Y=Y, zs=zs, sync=True)
    assert_allclose(got["last"], last, rtol=1e-9)
    assert_allclose(got["q_mu0"], ref.layers[0].q_mu.value, rtol=1e-9, atol=1e-12)
    assert_allclose(got["q_sqrt1"], ref.layers[1].q_sqrt.value, rtol=1e-9, atol=1e-12)
    assert_allclose(got["Z0"], ref.layers[0].feature.Z.value, rtol=1e-9, atol=1e-12)
    assert_allclose(got["lik"], ref.likelihood.likelihood.variance.value, rtol=1e-9)


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], bool(int(sys.argv[5])) if len(sys.argv) > 5 else False)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,launcher", [("flat", "self"), ("bucketed", "torchrun")])
def test_bench_contract_two_ranks_one_device(tmp_path, mode, launcher):
    """bench.py's N > 1 path both ways the driver may start it — plain `python bench.py --gpus 2` (bench.py spawns its own ranks
    under torch.distributed.run) and an explicit torch.distributed.run launch — on the test rig of a 1-GPU box: both ranks on
    cuda:0 over gloo.  One JSON line from rank 0, whole-job value, the rank count it observed, the exchange mode that was asked for."""
    import json
    env = dict(os.environ, DSDGP_BENCH_BACKEND="gloo", DSDGP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extras",
            "--allreduce", mode]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["rccl_ranks"] == 2 and b["steps"] == 6 and b["value"] > 0 and b["scaling"] == "strong"
    assert b["config"]["per_gpu_minibatch"] == 500 and b["config"]["global_batch"] == 1000
    assert ("per layer" in b["config"]["gradient_exchange"]) == (mode == "bucketed")
    assert np.isfinite(b["final_elbo"])
