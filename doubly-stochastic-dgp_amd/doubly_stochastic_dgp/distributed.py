"""Row-sharded data parallelism for the DGP ELBO: with full_cov=False every row of the (S*N, D) activation
is independent through all layers, so each rank (one process per GPU) takes its own minibatch rows x all S samples, holds
all parameters, recomputes the (tiny) Kuu/Lu/KL algebra redundantly and exchanges ONE flat buffer per step:

    [ d loss / d theta (n_theta doubles) | elbo | data term | KL | cholesky info ]   -- all-reduce(sum) over RCCL/xGMI

Each rank evaluates  loss_r = -(num_data / (n_local * world)) * sum_{rows of r} E_log_p_Y  +  (1/world) * sum_l KL_l,
so the plain SUM over ranks is exactly the single-process loss on the concatenated minibatch (dgp.py:96-98) and its
gradient: KL is counted once, the data term is scaled by num_data / N_global.
"""


def shard_terms(num_data, n_local, world):
    """(data_scale, kl_weight) of one rank."""
    if n_local <= 0:
        raise ValueError("empty minibatch: the ELBO scale num_data / N (dgp.py:96-98) is undefined for N = 0")
    return float(num_data) / float(n_local * world), 1.0 / float(world)


def allreduce_flat(buf, world):
    """Sum `buf` (gradient + 4 scalars, one contiguous tensor) over the default process group."""
    if world == 1:
        return buf
    import torch.distributed as dist
    if buf.is_cuda and dist.get_backend() == "gloo":
        # test rig (two processes sharing one GPU, no RCCL peer): stage through the host; production runs use backend
        # "nccl" (= RCCL) and reduce the device buffer in place on the shared stream
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buf.copy_(host)
        return buf
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def attach(model, rank, world):
    """Make `model` (DGP_Base) a data-parallel replica: rank-specific minibatch stream and Philox stream, gradient
    all-reduce before the Adam step.  Requires torch.distributed to be initialised (backend 'nccl' == RCCL on ROCm)."""
    from .dgp import Minibatch
    if model.minibatch_size:
        model._minibatch = Minibatch(model.X_data.shape[0], model.minibatch_size, seed=rank)

    def allreduce(eng, with_grad, sync=True):
        allreduce_flat(eng.gradbuf, world)
        if not sync:
            return None
        eng.ctx.sync()
        out = eng.out4.cpu().numpy().copy()
        out[3] /= world          # every rank factorises the same Kuu: the summed info is world x the failing pivot index
        return out

    object.__setattr__(model, "_dist", (rank, world, allreduce))
    return model
