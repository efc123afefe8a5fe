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


class _Buckets:
    """Per-layer gradient buckets (dsdgp_model_set_bucket_callback): the library calls `self._cb` from inside dsdgp_model_elbo as soon
    as a bucket's producers are enqueued; the all-reduce of that segment goes out at once, on the producing stream, and runs under
    the backward chains of the layers below.  `finish()` makes the engine's stream wait for every collective before the Adam step."""

    def __init__(self, eng, world):
        from . import _lib
        self.eng, self.world, self.works, self.count, self.error = eng, world, [], 0, None
        self._cb = _lib.BUCKET_FN(self._on_bucket)          # keep the ctypes thunk alive as long as the model

    def install(self):
        import ctypes as C
        from . import _lib
        _lib.check(self.eng.lib.dsdgp_model_set_bucket_callback(self.eng.model, C.cast(self._cb, C.c_void_p), None))
        # keyed on the engine's model GENERATION, not on the handle's address: Engine._ensure / _upload_if_needed destroy and re-create
        # the device model (a predict_* call with more rows than the training shape), and `new dsdgp_model` may get the freed address
        # back — a pointer comparison would then skip the re-installation on some ranks and the ranks would issue different collectives
        self._gen = self.eng.generation
        hooks = self.eng.__dict__.setdefault("_post_create", [])
        if self._reinstall not in hooks:
            hooks.append(self._reinstall)          # the engine re-installs the callback right after every dsdgp_model_create

    def _reinstall(self, eng):
        if eng is self.eng and eng.model is not None:
            self.install()

    def _on_bucket(self, user, bucket, ptr, count, stream):
        # (called from inside a ctypes call: an exception raised here would only be printed — keep it for finish())
        try:
            self._exchange(bucket, ptr, count, stream)
        except BaseException as e:  # noqa: BLE001
            if self.error is None:
                self.error = e

    def _exchange(self, bucket, ptr, count, stream):
        import torch
        import torch.distributed as dist
        eng = self.eng
        off = (ptr - eng.gradbuf.data_ptr()) // 8
        if 0 <= off and off + count <= eng.gradbuf.numel():
            view = eng.gradbuf[off:off + count]
        else:                                           # (result scalars outside the gradient buffer: not the engine's layout)
            raise RuntimeError("bucket outside Engine.gradbuf")
        self.count += 1
        ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
        if dist.get_backend() == "gloo":
            # test rig (two processes on one GPU, no RCCL peer): stage through the host, synchronously
            ext.synchronize()
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            view.copy_(host)
            return
        with torch.cuda.stream(ext):
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()                                    # the current (engine) stream waits for the collective's stream
        self.works = []
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError("gradient bucket exchange failed") from err


# gradient bytes from which the per-layer exchange is the default.  Measured on one MI355X with a one-rank RCCL group (the collectives
# are identities: everything AROUND them is what is timed, tools/bench_shards.py, profiles/r03_strong_scaling_shards.jsonl): the
# bucketed step costs 0.07 .. 0.13 ms more than the flat one at config 2 (four callbacks into Python, per-layer assembly launches,
# four collectives instead of one) — more than the whole 2.3 MB all-reduce it could hide.  From tens of MB of gradient (configs 3 / 5:
# 24 / 200 MB) the exchange is milliseconds and the per-layer form hides all but the lowest layer's.
BUCKET_MIN_BYTES = 16 << 20


def attach(model, rank, world, bucketed=None):
    """Make `model` (DGP_Base) a data-parallel replica: rank-specific minibatch stream and Philox stream, gradient
    all-reduce before the Adam step.  Requires torch.distributed to be initialised (backend 'nccl' == RCCL on ROCm).
    bucketed: one all-reduce per layer, issued from inside the reverse pass on the stream that produced the layer's gradient — it
    overlaps the lower layers' backward chains — instead of one flat all-reduce after the pass.  Default (None): on for world > 1
    when the gradient is at least BUCKET_MIN_BYTES (see above) and the library supports it for the model."""
    from .dgp import Minibatch
    if model.minibatch_size:
        model._minibatch = Minibatch(model.X_data.shape[0], model.minibatch_size, seed=rank)
    state = {"buckets": None, "model": None, "last_count": 0}

    def allreduce(eng, with_grad, sync=True):
        b = state["buckets"]
        state["last_count"] = b.count if b is not None else 0
        if b is not None and b.error is not None and b.count == 0:
            b.finish()                                  # raises: the first bucket already failed
        if b is not None and with_grad and b.count > 0:
            b.finish()                                  # the reverse pass already exchanged every bucket
            b.count = 0
        else:
            allreduce_flat(eng.gradbuf, world)
        if not sync:
            return None
        eng.ctx.sync()
        out = eng.out4.cpu().numpy().copy()
        out[3] /= world          # every rank factorises the same Kuu: the summed info is world x the failing pivot index
        return out

    def before_elbo(eng):
        """(re)install the callback on the engine's current device model (a model re-creation drops it)"""
        want = (world > 1 and eng.n_theta * 8 >= BUCKET_MIN_BYTES) if bucketed is None else bool(bucketed)
        if not want:
            return
        b = state["buckets"]
        if b is None or b.eng is not eng:
            b = _Buckets(eng, world)
            b.install()
            state["buckets"] = b
        elif b._gen != eng.generation:                  # (belt and braces: the engine's post-create hook re-installs it already)
            b.install()
        b.count = 0

    object.__setattr__(model, "_dist", (rank, world, allreduce))
    object.__setattr__(model, "_dist_before_elbo", before_elbo)
    # (tests) how many buckets the last reverse pass handed out
    object.__setattr__(model, "_dist_buckets", lambda: {"count": state["last_count"], "on": state["buckets"] is not None})
    return model
