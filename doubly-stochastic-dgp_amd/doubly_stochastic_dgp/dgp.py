"""Host mirror of the reference's model classes (dgp.py:35-192): `DGP_Base`, `DGP` with the same constructors and the
`propagate / _build_predict / E_log_p_Y / _build_likelihood / compute_log_likelihood / predict_*` surface.  Every
evaluation is one libdsdgp call (HIP kernels on gfx950); this file holds shapes, minibatching and bookkeeping only."""
import ctypes as C

import numpy as np

from . import _lib
from .gpflow_compat import Gaussian, Parameterized, Zero
from .layer_initializations import init_layers_linear
from .distributed import shard_terms
from .utils import BroadcastingLikelihood


class Minibatch:
    """[UPSTREAM] gpflow.params.Minibatch(value, batch_size, seed): epoch-wise shuffled minibatches.  X and Y use the
    same seed (dgp.py:51-52) so rows stay paired; TF's shuffle order itself is not reproducible, so this uses its own
    numpy Generator.  The data stays resident on the device; only the int64 row indices are produced on the host."""

    def __init__(self, n_rows, batch_size, seed=0):
        self.n_rows, self.batch_size = int(n_rows), int(batch_size)
        self.rng = np.random.default_rng(seed)
        self._perm = None
        self._pos = 0
        self._epoch = 0

    def _new_epoch(self):
        """Draw the next epoch's permutation.  EVERY regeneration bumps the epoch counter: callers key their device copy of
        the permutation on it (a minibatch that straddles an epoch boundary regenerates inside next_indices)."""
        self._perm = self.rng.permutation(self.n_rows)
        self._pos = 0
        self._epoch += 1

    def next_span(self):
        """(epoch permutation, start) of the next minibatch when it lies inside one epoch, else None (caller falls back to
        next_indices).  Lets the device keep one uploaded permutation per epoch instead of one index upload per step."""
        if self._perm is None or self._pos >= self.n_rows:
            self._new_epoch()
        if self._pos + self.batch_size > self.n_rows:
            return None
        start = self._pos
        self._pos += self.batch_size
        return self._perm, start, self._epoch

    def next_chunk(self, k):
        """The indices of the next k minibatches, concatenated (k * batch_size,)."""
        return np.concatenate([self.next_indices() for _ in range(int(k))])

    def next_indices(self):
        out = []
        need = self.batch_size
        while need > 0:
            if self._perm is None or self._pos >= self.n_rows:
                self._new_epoch()
            take = min(need, self.n_rows - self._pos)
            out.append(self._perm[self._pos:self._pos + take])
            self._pos += take
            need -= take
        return np.concatenate(out).astype(np.int64)


class DGP_Base(Parameterized):
    """dgp.py:35-126."""

    def __init__(self, X, Y, likelihood, layers, minibatch_size=None, num_samples=1, num_data=None, **kwargs):
        self.num_samples = int(num_samples)
        self.X_data = np.ascontiguousarray(X, dtype=np.float64)
        self.Y_data = np.ascontiguousarray(Y, dtype=np.float64)
        self.num_data = num_data or self.X_data.shape[0]                        # dgp.py:49
        self.minibatch_size = int(minibatch_size) if minibatch_size else None
        self._minibatch = Minibatch(self.X_data.shape[0], self.minibatch_size, seed=0) if self.minibatch_size else None
        self.likelihood = BroadcastingLikelihood(likelihood)                    # dgp.py:57
        self.likelihood.check_targets(self.Y_data)
        self.layers = list(layers)                                              # dgp.py:59
        self.white = bool(self.layers[0].white) if self.layers else False
        object.__setattr__(self, "_eng", None)
        object.__setattr__(self, "_dev_data", None)
        self._seed = 0
        # data-parallel hooks (distributed.py): (rank, world_size, all_reduce_fn)
        object.__setattr__(self, "_dist", None)

    # ------------------------------------------------------------------ engine / data plumbing
    def engine(self):
        if self._eng is None:
            from .engine import Engine
            n0 = self.minibatch_size or self.X_data.shape[0]
            eng = Engine(self.layers, self.likelihood.likelihood, self.white, n_max=n0, s_max=self.num_samples)
            for i, layer in enumerate(self.layers):
                object.__setattr__(layer, "_model_engine", (eng, i))
            object.__setattr__(self, "_eng", eng)
        return self._eng

    def _device_data(self):
        if self._dev_data is None:
            ctx = self.engine().ctx
            object.__setattr__(self, "_dev_data", (ctx.to_device(self.X_data), ctx.to_device(self.Y_data)))
        return self._dev_data

    def _next_seed(self):
        self._seed += 1
        return self._seed

    def _next_index_span(self):
        """(device index tensor, offset, n) of the next minibatch.  The row indices of the next CHUNK minibatches are drawn on the
        host in one go and uploaded once (pinned, asynchronous); a step then only moves an offset.  One upload per epoch (or per step
        when a minibatch straddles an epoch boundary — 2 of every 7 steps at 7372 / 1000) stalled the host behind the stream and the
        GPU behind the host: ~50 us on each such step."""
        Xd, _ = self._device_data()
        ctx = self.engine().ctx
        mb = self._minibatch
        if getattr(self, "_idx_src", None) is not mb or self._idx_pos >= self._idx_cnt:
            k = max(1, min(512, (1 << 18) // max(1, mb.batch_size)))
            host = ctx.torch.from_numpy(mb.next_chunk(k))
            try:
                host = host.pin_memory()
            except RuntimeError:
                pass
            self._idx_host = host                                  # alive until the copy has run
            # on the library's stream (ctx.tstream), whatever stream the caller has made current: the gather that reads the indices
            # is ordered behind this copy only there
            with ctx.torch.cuda.stream(ctx.tstream):
                self._idx_dev = host.to(Xd.device, non_blocking=True)
            self._idx_src, self._idx_pos, self._idx_cnt = mb, 0, k
        off = self._idx_pos * mb.batch_size
        self._idx_pos += 1
        return self._idx_dev, off, mb.batch_size

    def next_minibatch(self):
        """Device tensors (Xb, Yb) of the next minibatch (dgp.py:51-52), or the full data when not minibatching."""
        Xd, Yd = self._device_data()
        if self._minibatch is None:
            return Xd, Yd
        ctx = self.engine().ctx
        idx, off, n = self._next_index_span()
        Xb, Yb = ctx.empty(n, Xd.shape[1]), ctx.empty(n, Yd.shape[1])
        _lib.check(ctx.lib.dsdgp_gather_rows2(ctx.handle, C.c_void_p(Xd.data_ptr()), Xd.shape[1], C.c_void_p(Xb.data_ptr()),
                                              C.c_void_p(Yd.data_ptr()), Yd.shape[1], C.c_void_p(Yb.data_ptr()),
                                              C.c_void_p(idx.data_ptr()), n, off))
        self._keep_idx = idx
        return Xb, Yb

    @staticmethod
    def _np(ts):
        return [t.cpu().numpy() for t in ts]

    # ------------------------------------------------------------------ dgp.py:61-76
    def propagate(self, X, full_cov=False, S=1, zs=None):
        eng = self.engine()
        if full_cov:
            # plotting path (dgp.py:104-114): the layer loop of dgp.py:69-74 with (S,N,N,D) covariances; every piece of
            # arithmetic is a libdsdgp call (layers.py / utils.py mirrors), the loop itself stays on the host
            F = np.tile(np.asarray(X, dtype=np.float64)[None], [int(S), 1, 1])
            Fs, Fmeans, Fvars = [], [], []
            zs = zs or [None] * len(self.layers)
            for layer, z in zip(self.layers, zs):
                F, Fmean, Fvar = layer.sample_from_conditional(F, z=z, full_cov=True)
                Fs.append(F); Fmeans.append(Fmean); Fvars.append(Fvar)
            return Fs, Fmeans, Fvars
        Fs, Fmeans, Fvars = eng.propagate(X, int(S), zs=zs, seed=self._next_seed())
        eng.ctx.sync()
        Fs, Fmeans, Fvars = self._np(Fs), self._np(Fmeans), self._np(Fvars)
        if any(layer.input_prop_dim for layer in self.layers):
            # the device hands the concatenated [X_prop | F] to the next layer itself; the returned lists get the same
            # copies prepended here (layers.py:105-117)
            from .layers import concat_input_prop
            Xh = X.cpu().numpy() if hasattr(X, "data_ptr") else np.asarray(X, dtype=np.float64)
            Fin = np.tile(Xh[None], [int(S), 1, 1])
            for l, layer in enumerate(self.layers):
                if layer.input_prop_dim:
                    Fs[l], Fmeans[l], Fvars[l] = concat_input_prop(Fin, layer.input_prop_dim, Fs[l], Fmeans[l], Fvars[l])
                Fin = Fs[l]
        return Fs, Fmeans, Fvars

    # dgp.py:78-81
    def _build_predict(self, X, full_cov=False, S=1, zs=None):
        if full_cov:
            _, Fmeans, Fvars = self.propagate(X, full_cov=True, S=S, zs=zs)
            return Fmeans[-1], Fvars[-1]
        eng = self.engine()
        _, Fmeans, Fvars = eng.propagate(X, int(S), zs=zs, seed=self._next_seed(), want=("mean", "var"))
        eng.ctx.sync()
        return Fmeans[-1].cpu().numpy(), Fvars[-1].cpu().numpy()

    # dgp.py:83-90
    def E_log_p_Y(self, X, Y, zs=None):
        Fmean, Fvar = self._build_predict(X, full_cov=False, S=self.num_samples, zs=zs)
        return self.likelihood.variational_expectations_mean(Fmean, Fvar, np.asarray(Y, dtype=np.float64))

    # dgp.py:92-98
    def _build_likelihood(self, X=None, Y=None, zs=None, with_grad=False, grad_from_layer=0, grad_q_only=False):
        eng = self.engine()
        if X is None:
            X, Y = self.next_minibatch()
        elif not hasattr(Y, "data_ptr"):
            self.likelihood.check_targets(Y)
        n_local = X.shape[0]
        rank, world, allreduce = self._dist if self._dist else (0, 1, None)
        scale, klw = shard_terms(self.num_data, n_local, world)                 # dgp.py:96-97
        hook = getattr(self, "_dist_before_elbo", None)
        if hook is not None and allreduce is not None:
            eng._ensure(n_local, self.num_samples)
            eng._upload_if_needed()            # may re-create the device model (layout / jitter change): before the hook looks at it
            hook(eng)
        out = eng.elbo(X, Y, self.num_samples, zs=zs, seed=self._next_seed() * world + rank, data_scale=scale,
                       kl_weight=klw, with_grad=with_grad, sync=allreduce is None, grad_from_layer=grad_from_layer,
                       grad_q_only=grad_q_only)
        if allreduce is not None:
            out = allreduce(eng, with_grad)
            if out[3] != 0.0:          # every rank factorises the same Kuu ([UPSTREAM] tf.cholesky raises)
                raise _lib.CholeskyError(f"Cholesky decomposition was not successful (Kuu pivot {int(out[3])})")
        return float(out[0])

    def compute_log_likelihood(self, X=None, Y=None, zs=None):
        """[UPSTREAM] Model.compute_log_likelihood: evaluates _build_likelihood (one MC draw of the ELBO)."""
        return self._build_likelihood(X, Y, zs)

    def train_step(self, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, X=None, Y=None, zs=None, sync=False):
        """One optimiser step of -ELBO: minibatch gather + forward + reverse-mode gradient + Adam
        (one `session.run(opt_op)` of demos/demo_regression_UCI.ipynb:324).  Returns the ELBO if sync=True."""
        eng = self.engine()
        rank, world, allreduce = self._dist if self._dist else (0, 1, None)
        if X is None and zs is None and allreduce is None and self._minibatch is not None:
            # the whole step — minibatch gather, ELBO, gradient, Adam — in one library call
            Xd, Yd = self._device_data()
            idx, off, n = self._next_index_span()
            scale, klw = shard_terms(self.num_data, n, 1)
            eng.train_step_minibatch(Xd, Yd, idx, off, n, self.num_samples, seed=self._next_seed(), data_scale=scale, kl_weight=klw,
                                     lr=lr, beta1=beta1, beta2=beta2, eps=eps)
            return self._sync_result(eng, None, world) if sync else None
        if X is None:
            X, Y = self.next_minibatch()
        elif not hasattr(Y, "data_ptr"):
            self.likelihood.check_targets(Y)
        n_local = X.shape[0]
        scale, klw = shard_terms(self.num_data, n_local, world)
        if allreduce is None:
            # single process: ELBO, gradient and Adam update in one library call (the update rides in the reverse pass's last launch)
            eng.train_step(X, Y, self.num_samples, zs=zs, seed=self._next_seed(), data_scale=scale, kl_weight=klw, lr=lr, beta1=beta1,
                           beta2=beta2, eps=eps)
            out = None
        else:
            hook = getattr(self, "_dist_before_elbo", None)
            if hook is not None:
                eng._ensure(n_local, self.num_samples)
                eng._upload_if_needed()
                hook(eng)
            out = eng.elbo(X, Y, self.num_samples, zs=zs, seed=self._next_seed() * world + rank, data_scale=scale,
                           kl_weight=klw, with_grad=True, sync=False)
            out = allreduce(eng, True, sync=sync)
            eng.adam_step(lr, beta1, beta2, eps)
        return self._sync_result(eng, out, world) if sync else None

    @staticmethod
    def _sync_result(eng, out, world):
        eng.ctx.sync()
        o = eng.out4.cpu().numpy() if out is None else out
        if out is None and world > 1:
            o = o.copy(); o[3] /= world
        if o[3] != 0.0:      # asynchronous steps report a failed Kuu factorisation here ([UPSTREAM] tf.cholesky raises)
            raise _lib.CholeskyError(f"Cholesky decomposition was not successful (Kuu pivot {int(o[3])})")
        return float(o[0])

    # ------------------------------------------------------------------ dgp.py:100-126
    def predict_f(self, Xnew, num_samples):
        return self._build_predict(Xnew, full_cov=False, S=num_samples)

    def predict_f_full_cov(self, Xnew, num_samples):
        return self._build_predict(Xnew, full_cov=True, S=num_samples)

    def predict_all_layers(self, Xnew, num_samples):
        return self.propagate(Xnew, full_cov=False, S=num_samples)

    def predict_all_layers_full_cov(self, Xnew, num_samples):
        return self.propagate(Xnew, full_cov=True, S=num_samples)

    def predict_y(self, Xnew, num_samples):
        Fmean, Fvar = self._build_predict(Xnew, full_cov=False, S=num_samples)
        return self.likelihood.predict_mean_and_var(Fmean, Fvar)

    def predict_density(self, Xnew, Ynew, num_samples):
        Fmean, Fvar = self._build_predict(Xnew, full_cov=False, S=num_samples)
        return self.likelihood.predict_density_logmeanexp(Fmean, Fvar, np.asarray(Ynew, dtype=np.float64))


class DGP_Quad(DGP_Base):
    """A DGP evaluated with Gauss-Hermite quadrature over the inner layers instead of Monte-Carlo samples (dgp.py:129-166):
    H**D_quad deterministic whitened points, D_quad = the summed inner-layer widths, injected through `zs` with shape
    (S,1,D) and combined with the quadrature weights in place of the mean over S.  Exponential in D_quad — a test oracle for
    the sampler (tests/test_dgp.py:120-174), evaluated (and differentiated) by the same device path as DGP."""

    def __init__(self, *args, H=100, **kwargs):
        DGP_Base.__init__(self, *args, **kwargs)
        from .utils import mvhermgauss
        self.H = int(H)
        self.D_quad = sum(layer.q_mu.shape[1] for layer in self.layers[:-1])        # dgp.py:142
        gh_x, gh_w = mvhermgauss(self.H, self.D_quad)
        gh_x = gh_x * 2.0 ** 0.5                                                     # dgp.py:144
        self.gh_w = gh_w * np.pi ** (-0.5 * self.D_quad)                             # dgp.py:145
        self.gh_x, s = [], 0
        for layer in self.layers[:-1]:                                               # dgp.py:149-154: (S,1,D) slices
            e = s + layer.q_mu.shape[1]
            self.gh_x.append(np.ascontiguousarray(gh_x[:, None, s:e]))
            s = e
        self.gh_x.append(np.zeros((1, 1, 1)))                                        # dgp.py:157 (never used)
        self.num_samples = self.H ** self.D_quad                                     # dgp.py:164

    def engine(self):
        eng = DGP_Base.engine(self)
        if getattr(eng, "_sample_w", None) is None:
            eng.set_sample_weights(eng.ctx.to_device(self.gh_w))
        return eng

    def E_log_p_Y(self, X, Y, zs=None):                                              # dgp.py:160-166
        Fmean, Fvar = self._build_predict(X, full_cov=False, S=self.num_samples, zs=self.gh_x)
        return self.likelihood.variational_expectations_mean(Fmean, Fvar, np.asarray(Y, dtype=np.float64), weights=self.gh_w)

    def _build_likelihood(self, X=None, Y=None, zs=None, with_grad=False, grad_from_layer=0, grad_q_only=False):
        return DGP_Base._build_likelihood(self, X, Y, zs=self.gh_x, with_grad=with_grad, grad_from_layer=grad_from_layer,
                                          grad_q_only=grad_q_only)

    def train_step(self, *args, **kwargs):
        kwargs["zs"] = self.gh_x
        return DGP_Base.train_step(self, *args, **kwargs)


class DGP(DGP_Base):
    """The doubly-stochastic DGP with linear/identity mean functions (dgp.py:169-192)."""

    def __init__(self, X, Y, Z, kernels, likelihood, num_outputs=None, mean_function=None, white=False, **kwargs):
        layers = init_layers_linear(X, Y, Z, kernels, num_outputs=num_outputs,
                                    mean_function=Zero() if mean_function is None else mean_function, white=white)
        DGP_Base.__init__(self, X, Y, likelihood, layers, **kwargs)
