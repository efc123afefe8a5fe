"""Exercises the data-parallel code path (distributed.attach -> flat all-reduce over RCCL on the shared stream) on the single
GPU of the test box with a 1-rank NCCL process group: same numbers as the non-distributed step, no hang, fp64 all-reduce OK.
(The world_size-2 arithmetic of the sharding is covered on CPU by tests/test_distributed_cpu.py.)"""
import os
import socket

import numpy as np
import pytest
from numpy.testing import assert_allclose

from tests.helpers import kern_spec, make_case

pytestmark = pytest.mark.gpu


def test_attach_allreduce_single_rank_nccl():
    import torch
    import torch.distributed as dist
    from doubly_stochastic_dgp.distributed import attach
    from doubly_stochastic_dgp.engine import Context
    Context.get()                                   # installs the shared stream before the process group is created
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        rng = np.random.RandomState(0)
        N, D, M, S = 200, 4, 32, 3
        X, Y = rng.randn(N, D), rng.randn(N, 1)
        Z = X[:M].copy()
        zs = [rng.randn(S, N, D), rng.randn(S, N, 1)]
        _, _, ref = make_case(X, Y, Z, [kern_spec("rbf", D)] * 2, S=S, num_data=1000)
        _, _, dp = make_case(X, Y, Z, [kern_spec("rbf", D)] * 2, S=S, num_data=1000)
        attach(dp, 0, 1)
        e_ref = ref._build_likelihood(X, Y, zs=zs, with_grad=True)
        e_dp = dp._build_likelihood(X, Y, zs=zs, with_grad=True)
        assert_allclose(e_dp, e_ref, rtol=1e-13)
        assert np.array_equal(dp.engine().grad.cpu().numpy(), ref.engine().grad.cpu().numpy())
        for _ in range(3):
            ref.train_step(0.01, X=X, Y=Y, zs=zs)
            dp.train_step(0.01, X=X, Y=Y, zs=zs)
        assert_allclose(dp.layers[0].q_mu.value, ref.layers[0].q_mu.value, rtol=1e-12, atol=1e-14)
        t = torch.ones(4, dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        assert float(t.sum().item()) == 4.0
        # the BUCKETED exchange (one RCCL all-reduce per layer, issued from the library's callback on the stream that produced the
        # layer's gradient: torch.cuda.ExternalStream + async_op) — small model: every bucket on the main stream; large model
        # (n S Mp >= 2^20): the upper layers' buckets on the side stream, under the lower layers' backward chains
        for (N2, D2, M2, S2, L2) in ((200, 4, 32, 3, 2), (1000, 5, 96, 14, 3)):
            rng = np.random.RandomState(1)
            X2, Y2 = rng.randn(N2, D2), rng.randn(N2, 1)
            Z2 = X2[:M2] + 0.05 * rng.randn(M2, D2)
            zs2 = [rng.randn(S2, N2, D2)] * (L2 - 1) + [rng.randn(S2, N2, 1)]
            _, _, ref2 = make_case(X2, Y2, Z2, [kern_spec("rbf", D2)] * L2, S=S2, num_data=5000)
            _, _, dpb = make_case(X2, Y2, Z2, [kern_spec("rbf", D2)] * L2, S=S2, num_data=5000)
            attach(dpb, 0, 1, bucketed=True)
            e_ref = ref2._build_likelihood(X2, Y2, zs=zs2, with_grad=True)
            e_b = dpb._build_likelihood(X2, Y2, zs=zs2, with_grad=True)
            assert dpb._dist_buckets()["count"] >= L2 + 1, "the bucket callback did not run"
            assert_allclose(e_b, e_ref, rtol=1e-13)
            assert_allclose(dpb.engine().grad.cpu().numpy(), ref2.engine().grad.cpu().numpy(), rtol=1e-12, atol=1e-14)
            for _ in range(3):
                ref2.train_step(0.01, X=X2, Y=Y2, zs=zs2)
                dpb.train_step(0.01, X=X2, Y=Y2, zs=zs2)
            ref2.engine().sync_to_host(); dpb.engine().sync_to_host()
            assert_allclose(dpb.layers[0].q_mu.value, ref2.layers[0].q_mu.value, rtol=1e-10, atol=1e-13)
            assert_allclose(dpb.layers[-1].q_sqrt.value, ref2.layers[-1].q_sqrt.value, rtol=1e-10, atol=1e-13)
    finally:
        dist.destroy_process_group()


def test_cabi_allreduce_single_rank_rccl():
    """dsdgp_allreduce (include/dsdgp.h) on a 1-rank RCCL communicator created through RCCL's own C API — what a non-torch
    host does: ncclGetUniqueId -> ncclCommInitRank -> dsdgp_allreduce on the ctx stream.  Sum over one rank = identity."""
    import ctypes as C
    import torch
    from doubly_stochastic_dgp import _lib
    from doubly_stochastic_dgp.engine import Context
    ctx = Context.get()
    rccl = None
    for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "no librccl.so on this box"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        x = np.arange(1000, dtype=np.float64) * 0.5 - 3.0
        t = ctx.to_device(x)
        ctx.torch.cuda.current_stream().synchronize()
        _lib.check(ctx.lib.dsdgp_allreduce(ctx.handle, comm, C.c_void_p(t.data_ptr()), t.numel()))
        ctx.sync()
        assert np.array_equal(t.cpu().numpy(), x)
        rc = ctx.lib.dsdgp_allreduce(ctx.handle, None, C.c_void_p(t.data_ptr()), t.numel())
        assert rc == -1                                   # DSDGP_ERR_BAD_ARG: no communicator
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
