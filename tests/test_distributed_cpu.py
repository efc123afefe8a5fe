"""N>1 path on CPU: world_size-2 gloo process group.  Each rank evaluates its row shard with
(data_scale, kl_weight) = shard_terms(...); ONE flat all-reduce(sum) of [gradient | elbo] must reproduce the
single-process full-minibatch ELBO and gradient exactly (KL counted once, data term scaled by num_data / N_global)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dgp_oracle as O
from oracle import model as OM
from tests.helpers import kern_spec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.RandomState(0)
    N, D, M, S = 40, 3, 12, 3
    X, Y = rng.randn(N, D), rng.randn(N, 1)
    Z = X[:M].copy()
    lds = O.init_layers_linear(X, Y, Z, [kern_spec("rbf", D)] * 2)
    for l in lds:
        l["q_mu"] = 0.2 * rng.randn(*l["q_mu"].shape)
    sl, state = OM.state_from_layers(lds, lik_variance=0.2)
    spec = dict(jitter=1e-6, white=False, likelihood="gaussian", layers=sl)
    zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
    return spec, state, X, Y, zs, S


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "doubly-stochastic-dgp_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from doubly_stochastic_dgp.distributed import allreduce_flat, shard_terms
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec, state, X, Y, zs, S = _problem()
    N = X.shape[0]
    n_local = N // world
    sl = slice(rank * n_local, (rank + 1) * n_local)
    scale, klw = shard_terms(num_data=1000, n_local=n_local, world=world)
    flat = OM.elbo_and_grad_sharded(spec, state, X[sl], Y[sl], [z[:, sl, :] for z in zs], S, scale, klw)
    buf = torch.as_tensor(flat.copy())
    allreduce_flat(buf, world)
    if rank == 0:
        q.put(buf.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sharded_allreduce_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spec, state, X, Y, zs, S = _problem()
    ref_elbo, g = OM.elbo_and_grad(spec, state, X, Y, zs, S, num_data=1000)
    ref = np.concatenate([g[k].ravel() for k in sorted(g)] + [[ref_elbo]])
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_shard_terms_and_rank_streams():
    from doubly_stochastic_dgp.dgp import DGP
    from doubly_stochastic_dgp.distributed import attach, shard_terms
    from doubly_stochastic_dgp.gpflow_compat import RBF, Gaussian
    assert shard_terms(7372, 1000, 8) == (7372 / 8000.0, 0.125)
    X = np.random.RandomState(0).randn(300, 2)
    models = []
    for r in range(2):
        m = DGP(X, X[:, :1], X[:8], [RBF(2)], Gaussian(), minibatch_size=50)
        attach(m, r, 2)
        models.append(m)
    i0, i1 = models[0]._minibatch.next_indices(), models[1]._minibatch.next_indices()
    assert not np.array_equal(i0, i1)                  # ranks draw different minibatches
    assert models[0]._dist[0] == 0 and models[1]._dist[1] == 2


def test_exchange_form_is_chosen_by_gradient_size():
    """attach(bucketed=None): flat all-reduce below BUCKET_MIN_BYTES of gradient, per-layer buckets above (world > 1 only);
    an explicit bucketed=True / False overrides.  Exercised with stand-ins for the engine: no GPU, no process group."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "doubly-stochastic-dgp_amd"))
    from doubly_stochastic_dgp import distributed as D

    class FakeLib:
        def __init__(self):
            self.installed = 0

        def dsdgp_model_set_bucket_callback(self, model, fn, user):
            self.installed += 1
            return 0

    class FakeEng:
        def __init__(self, n_theta):
            self.n_theta, self.lib, self.model, self.generation = n_theta, FakeLib(), C.c_void_p(1234), 1

    class FakeModel:
        minibatch_size = None

    small, big = (D.BUCKET_MIN_BYTES // 8) - 1, D.BUCKET_MIN_BYTES // 8
    for world, bucketed, n_theta, want in [(2, None, small, 0), (2, None, big, 1), (1, None, big, 0), (2, True, small, 1),
                                           (2, False, big, 0), (8, None, 283783, 0), (8, None, 25_000_000, 1)]:
        m = FakeModel()
        D.attach(m, 0, world, bucketed=bucketed)
        eng = FakeEng(n_theta)
        m._dist_before_elbo(eng)
        assert eng.lib.installed == want, (world, bucketed, n_theta)
        assert m._dist_buckets()["on"] == bool(want)
        m._dist_before_elbo(eng)                       # a second evaluation on the same device model does not re-install
        assert eng.lib.installed == want
        # the device model is re-created at the SAME address (a freed handle coming back from the allocator): the callback must be
        # installed again — keyed on the engine's generation counter, not on the pointer
        eng.generation += 1
        for fn in getattr(eng, "_post_create", ()):    # what Engine._ensure runs right after dsdgp_model_create
            fn(eng)
        assert eng.lib.installed == 2 * want
        m._dist_before_elbo(eng)
        assert eng.lib.installed == 2 * want
        eng.generation += 1                            # ... and if the engine's hook list were bypassed, the evaluation hook catches it
        m._dist_before_elbo(eng)
        assert eng.lib.installed == 3 * want
