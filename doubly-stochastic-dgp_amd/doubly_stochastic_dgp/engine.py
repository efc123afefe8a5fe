"""Device engine: owns the flat unconstrained parameter vector theta, the workspace and the libdsdgp model handle for a
list of SVGP layers + likelihood, and keeps host `Parameter` objects coherent with it.

torch is used ONLY as the device allocator / stream provider (and, in distributed.py, for the RCCL process group);
no torch op runs on the hot path.
"""
import ctypes as C

import numpy as np

from . import _lib, settings
from .gpflow_compat import (Bernoulli, Beta, Exponential, Gamma, Gaussian, MultiClass, Parameter, Poisson, StudentT, positive_backward, positive_forward, split_kernel)

_KIND = {"rbf": _lib.KERN_RBF, "matern52": _lib.KERN_MATERN52}
_MEAN = {"zero": _lib.MEAN_ZERO, "identity": _lib.MEAN_IDENTITY, "linear": _lib.MEAN_LINEAR}


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.DsdgpError("no MI355X visible (torch.cuda.is_available() is False): the DGP compute path has no CPU "
                              "fallback")
    return torch


class Context:
    """One libdsdgp context per process, bound to torch's current HIP stream on cuda:<device>."""
    _inst = None

    def __init__(self, device=None):
        torch = _torch()
        self.lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else device
        self.torch = torch
        # ONE stream for everything: torch's default stream has handle 0, which the C-ABI reads as "create a private
        # stream", and a private stream is not ordered with torch's copies nor with the RCCL all-reduce.  So create a
        # dedicated stream, make it torch's current stream for this process, and give its handle to libdsdgp: kernels,
        # H2D/D2H copies and collectives are then ordered by the stream itself.
        self.tstream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.tstream)
        h = C.c_void_p()
        _lib.check(self.lib.dsdgp_ctx_create(C.byref(h), self.device, C.c_void_p(self.tstream.cuda_stream)))
        self.handle = h

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = Context()
        return cls._inst

    def sync(self):
        _lib.check(self.lib.dsdgp_sync(self.handle))

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=f"cuda:{self.device}")

    def to_device(self, arr):
        a = np.array(arr, dtype=np.float64, order="C", copy=True) if not getattr(arr, "flags", None) or not arr.flags.writeable \
            else np.ascontiguousarray(arr, dtype=np.float64)
        return self.torch.as_tensor(a).to(f"cuda:{self.device}")

    def prof_enable(self, on=True):
        _lib.check(self.lib.dsdgp_prof_enable(self.handle, int(on)))

    def prof_read(self, name, reset=True):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(self.lib.dsdgp_prof_read(self.handle, name.encode(), C.byref(ms), C.byref(cnt), int(reset)))
        return ms.value, cnt.value


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Engine:
    def __init__(self, layers, likelihood, white, n_max=1024, s_max=1):
        self.ctx = Context.get()
        self.lib = self.ctx.lib
        self.layers, self.likelihood, self.white = list(layers), likelihood, bool(white)
        self.jitter = float(settings.jitter)
        self.model = None
        self.n_max, self.s_max = 0, 0
        self._host_dirty = True        # host Parameter values newer than device theta
        self._dev_dirty = False        # device theta newer than host Parameter values
        self._structure_dirty = False
        self._needs_prepare = True
        self._sample_w = None
        self._build_layout()
        self._ensure(n_max, s_max)

    # ------------------------------------------------------------------ parameter layout
    def _params_of(self, layer):
        stat, white = split_kernel(layer.kern)
        return stat, white

    def _build_layout(self):
        off = 0
        self.entries = []      # (Parameter, offset, count, kind) ; kind in {'id','pos','qsqrt'}
        self.desc = _lib.ModelDesc()
        d = self.desc
        d.L = len(self.layers)
        if d.L > _lib.DSDGP_MAX_LAYERS:
            raise ValueError("too many layers")
        d.white = int(self.white)
        d.jitter = self.jitter
        self._mean_A = []
        for l, layer in enumerate(self.layers):
            stat, wk = self._params_of(layer)
            ld = d.layers[l]
            M, Din = layer.feature.Z.shape
            Dout = layer.num_outputs
            ld.M, ld.D_in, ld.D_out = M, Din, Dout
            ld.kern_kind = _KIND[stat.kind]
            ld.ard = int(stat.ARD)
            ld.has_white = int(wk is not None)
            ld.mean_kind = _MEAN[layer.mean_function.kind]
            ld.off_mean_A = ld.off_mean_b = -1
            mf_in_theta = False
            if layer.mean_function.kind == "linear":
                A = np.asarray(layer.mean_function.A._value, dtype=np.float64)
                if A.shape != (Din, Dout):
                    raise ValueError("Linear mean function has the wrong shape")
                mf_in_theta = layer.mean_function.in_theta()
                if not mf_in_theta:                       # fixed map, no bias: a constant device array
                    tA = self.ctx.to_device(A)
                    self._mean_A.append((layer.mean_function, tA))
                    ld.mean_A = tA.data_ptr()
                    # still owned: a later `mean_function.A = W`, a non-zero b or set_trainable(True) must reach the device
                    for p in (layer.mean_function.A, layer.mean_function.b):
                        if self not in p._owners:
                            p._owners.append(self)

            def add(p, kind, name):
                nonlocal off
                cnt = int(np.prod(p.shape)) if p.shape else 1
                self.entries.append((p, off, cnt, kind))
                setattr(ld, "off_" + name, off)
                if self not in p._owners:
                    p._owners.append(self)
                off += cnt
                return p.trainable

            ld.input_prop_dim = int(getattr(layer, "input_prop_dim", None) or 0)
            if ld.input_prop_dim and l == len(self.layers) - 1 and len(self.layers) > 1:
                raise NotImplementedError("the last layer of a DGP does not propagate inputs "
                                          "(layer_initializations.py:75-78)")
            ld.trainable_Z = int(add(layer.feature.Z, "id", "Z"))
            ld.trainable_q_mu = int(add(layer.q_mu, "id", "q_mu"))
            ld.trainable_q_sqrt = int(add(layer.q_sqrt, "qsqrt", "q_sqrt"))
            # a variance Parameter WITHOUT the positive transform (tests/test_dgp.py:79-85: NoTransformMatern52, variance 1e-24) is
            # its own free variable on the device too
            ld.kvar_identity = int(getattr(stat.variance, "transform", "positive") is None)
            ld.trainable_kvar = int(add(stat.variance, "id" if ld.kvar_identity else "pos", "kvar"))
            ld.trainable_kls = int(add(stat.lengthscales, "pos", "kls"))
            if wk is not None:
                ld.trainable_wvar = int(add(wk.variance, "pos", "wvar"))
            if mf_in_theta:
                ld.trainable_mean_A = int(add(layer.mean_function.A, "id", "mean_A"))
                ld.trainable_mean_b = int(add(layer.mean_function.b, "id", "mean_b"))
        if isinstance(self.likelihood, Gaussian):
            d.lik_kind = _lib.LIK_GAUSSIAN
            p = self.likelihood.variance
            self.entries.append((p, off, 1, "pos"))
            if self not in p._owners:
                p._owners.append(self)
            d.off_lik_var = off
            d.trainable_lik_var = int(p.trainable)
            off += 1
        elif isinstance(self.likelihood, MultiClass):
            d.lik_kind = _lib.LIK_MULTICLASS
            d.num_classes = self.likelihood.num_classes
            d.off_lik_var = -1
        elif isinstance(self.likelihood, Bernoulli):
            d.lik_kind = _lib.LIK_BERNOULLI
            d.off_lik_var = -1
        elif isinstance(self.likelihood, Poisson):
            d.lik_kind = _lib.LIK_POISSON
            d.lik_aux = self.likelihood.binsize
            d.off_lik_var = -1
        elif isinstance(self.likelihood, Exponential):
            d.lik_kind = _lib.LIK_EXPONENTIAL
            d.off_lik_var = -1
        elif isinstance(self.likelihood, (StudentT, Gamma, Beta)):
            if isinstance(self.likelihood, StudentT):
                d.lik_kind = _lib.LIK_STUDENT_T
                d.lik_aux = self.likelihood.deg_free
            else:
                d.lik_kind = _lib.LIK_GAMMA if isinstance(self.likelihood, Gamma) else _lib.LIK_BETA
            # the likelihood's one positive parameter (StudentT.scale, Gamma.shape, Beta.scale): the slot Gaussian.variance takes
            p = self.likelihood.shape if isinstance(self.likelihood, Gamma) else self.likelihood.scale
            self.entries.append((p, off, 1, "pos"))
            if self not in p._owners:
                p._owners.append(self)
            d.off_lik_var = off
            d.trainable_lik_var = int(p.trainable)
            off += 1
        else:
            raise NotImplementedError(type(self.likelihood).__name__)
        d.n_theta = off
        self.n_theta = off

    def _pack_host(self):
        th = np.zeros(self.n_theta)
        for p, off, cnt, kind in self.entries:
            v = np.asarray(p._value, dtype=np.float64)
            if kind == "pos":
                v = positive_backward(v)
            elif kind == "qsqrt":
                v = np.tril(v)
            th[off:off + cnt] = np.ravel(v)
        return th

    def _unpack_host(self, th):
        for p, off, cnt, kind in self.entries:
            v = th[off:off + cnt].reshape(p.shape)
            if kind == "pos":
                v = positive_forward(v)
            p._value = np.array(v, dtype=np.float64)

    # ------------------------------------------------------------------ coherence
    def mark_host_dirty(self):
        self._host_dirty = True

    def mark_structure_dirty(self):
        self._structure_dirty = True

    def sync_to_host(self):
        if self._dev_dirty and self.model is not None:
            self.ctx.sync()
            self._unpack_host(self.theta.cpu().numpy())
            self._dev_dirty = False

    def _upload_if_needed(self):
        if self._host_dirty and any(mf.in_theta() for mf, _ in self._mean_A):
            self._structure_dirty = True       # a fixed Linear map gained a bias / became trainable: it moves into theta
        if settings.jitter != self.jitter or self._structure_dirty:
            self.sync_to_host()
            self.jitter = float(settings.jitter)
            self._structure_dirty = False
            for p, *_ in self.entries:
                if self in p._owners:
                    p._owners.remove(self)
            for mf, _ in self._mean_A:
                for p in (mf.A, mf.b):
                    if self in p._owners:
                        p._owners.remove(self)
            self._build_layout()
            n, s = self.n_max, self.s_max
            self._destroy()
            self._host_dirty = True
            self._ensure(n, s)
        if self._host_dirty:
            for mf, tA in self._mean_A:            # fixed maps: refresh the constant device copy in place (same pointer)
                tA.copy_(self.ctx.torch.as_tensor(np.ascontiguousarray(mf.A._value, dtype=np.float64)))
            th = self._pack_host()
            self.theta.copy_(self.ctx.torch.as_tensor(th))
            _lib.check(self.lib.dsdgp_model_theta_changed(self.model))     # our side of dsdgp_model_track_theta
            self._host_dirty = False
            self._dev_dirty = False
            self._needs_prepare = True

    def _prepare_checked(self):
        """build_cholesky_if_needed (layers.py:167-175) once per parameter change, surfacing tf.cholesky failures."""
        self._upload_if_needed()
        if self._needs_prepare:
            info = C.c_int(0)
            _lib.check(self.lib.dsdgp_model_prepare(self.model, C.byref(info)))
            self._needs_prepare = False

    # ------------------------------------------------------------------ model lifetime
    def _destroy(self):
        if self.model is not None:
            _lib.check(self.lib.dsdgp_model_destroy(self.model))
            self.model = None
        self.n_max = self.s_max = 0

    def _ensure(self, n, s):
        """(Re)create the device model for up to n rows x s samples.  Only the workspace depends on (n, s): theta, the
        gradient buffer and the Adam moments / step count are allocated once per parameter layout and SURVIVE a re-creation
        (a predict_* call with more rows or samples than the training shape must not reset the optimiser — TF keeps its
        Adam slots across predictions)."""
        if self.model is not None and n <= self.n_max and s <= self.s_max:
            return
        n_max, s_max = max(n, self.n_max), max(s, self.s_max)
        self._destroy()
        torch = self.ctx.torch
        nbytes = C.c_int64()
        _lib.check(self.lib.dsdgp_model_workspace_bytes(C.byref(self.desc), n_max, s_max, C.byref(nbytes)))
        dev = f"cuda:{self.ctx.device}"
        self.workspace = None                      # release the old workspace before allocating the larger one
        self.workspace = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=dev)
        base = self.workspace.data_ptr()
        self._ws_ptr = (base + 255) // 256 * 256
        fresh = getattr(self, "theta", None) is None or self.theta.numel() != self.n_theta
        if fresh:
            self.theta = torch.zeros(self.n_theta, dtype=torch.float64, device=dev)
            # gradient and the 4 result scalars share one buffer so that data-parallel runs need ONE all-reduce per step
            self.gradbuf = torch.zeros(self.n_theta + 4, dtype=torch.float64, device=dev)
            self.grad = self.gradbuf[:self.n_theta]
            self.adam_m = torch.zeros(self.n_theta, dtype=torch.float64, device=dev)
            self.adam_v = torch.zeros(self.n_theta, dtype=torch.float64, device=dev)
            self.out4 = self.gradbuf[self.n_theta:]
            self.adam_t = 0
            self._host_dirty = True
        h = C.c_void_p()
        _lib.check(self.lib.dsdgp_model_create(self.ctx.handle, C.byref(self.desc), n_max, s_max, ptr(self.theta),
                                               ptr(self.grad), ptr(self.adam_m), ptr(self.adam_v),
                                               C.c_void_p(self._ws_ptr), nbytes.value, C.byref(h)))
        self.model = h
        self.generation = getattr(self, "generation", 0) + 1     # identifies THIS device model (a freed handle's address can come back)
        self._grad_first = 0
        self._grad_q_only = False
        # this engine is the only writer of theta besides the library's own optimiser steps and reports its uploads
        # (_upload_if_needed): an evaluation after a natural-gradient step alone keeps the factorisation of Ku
        _lib.check(self.lib.dsdgp_model_track_theta(self.model, 1))
        self.n_max, self.s_max = n_max, s_max
        if getattr(self, "_sample_w", None) is not None:       # DGP_Quad weights survive a model re-creation
            self.set_sample_weights(self._sample_w)
        for fn in getattr(self, "_post_create", ()):            # per-model settings owned by others (the data-parallel bucket callback)
            fn(self)
        self._needs_prepare = True

    # ------------------------------------------------------------------ compute
    def _zs_args(self, zs, S, n):
        """zs: list (per layer) of None or arrays broadcastable to (S, n, D_out) -> (ptr array, stride array, keepalive)"""
        L = len(self.layers)
        if zs is None:
            return None, None, []
        ptrs = (C.c_void_p * L)()
        strides = (C.c_int64 * (3 * L))()
        keep = []
        for l, z in enumerate(zs):
            if z is None:
                ptrs[l] = None
                continue
            D = self.layers[l].num_outputs
            if hasattr(z, "data_ptr"):
                t = z
                shape = tuple(t.shape)
            else:
                z = np.asarray(z, dtype=np.float64)
                if z.ndim != 3:
                    raise ValueError("z must be rank-3, broadcastable to (S, N, D_out)")
                shape = z.shape
                t = self.ctx.to_device(z)
            for want, got in zip((S, n, D), shape):
                if got not in (1, want):
                    raise ValueError(f"z of shape {shape} does not broadcast to {(S, n, D)}")
            keep.append(t)
            ptrs[l] = t.data_ptr()
            st = (shape[1] * shape[2], shape[2], 1)
            for k in range(3):
                strides[3 * l + k] = 0 if shape[k] == 1 else st[k]
        return ptrs, strides, keep

    def prepare(self):
        self._prepare_checked()

    def propagate(self, X, S, zs=None, seed=0, want=("F", "mean", "var")):
        """X: numpy (n, D) or device tensor. Returns lists of device tensors (S, n, D_out_l)."""
        Xd = X if hasattr(X, "data_ptr") else self.ctx.to_device(X)
        n = Xd.shape[0]
        self._ensure(n, S)
        self._prepare_checked()
        L = len(self.layers)
        outs = {}
        arrs = {}
        for key in ("F", "mean", "var"):
            if key in want:
                ts = [self.ctx.empty(S, n, layer.num_outputs) for layer in self.layers]
                a = (C.c_void_p * L)(*[t.data_ptr() for t in ts])
            else:
                ts, a = None, None
            outs[key], arrs[key] = ts, a
        zp, zst, keep = self._zs_args(zs, S, n)
        _lib.check(self.lib.dsdgp_model_propagate(self.model, ptr(Xd), n, S, zp, zst, C.c_uint64(seed), arrs["F"],
                                                  arrs["mean"], arrs["var"]))
        return outs["F"], outs["mean"], outs["var"]

    def elbo(self, X, Y, S, zs=None, seed=0, data_scale=1.0, kl_weight=1.0, with_grad=False, sync=True, grad_from_layer=0,
             grad_q_only=False):
        """grad_from_layer = l > 0: the reverse pass stops below layer l, as tf.gradients(loss, var_list) does for a var_list of
        upper-layer (q_mu, q_sqrt) pairs (NatGradOptimizer); the gradient entries of the lower layers are then NOT updated.
        grad_q_only: var_list holds nothing but (q_mu, q_sqrt) pairs — only those entries of the layers >= l are produced (the other
        entries of these layers are undefined afterwards; an Adam step is refused until a full gradient has been evaluated)."""
        Xd = X if hasattr(X, "data_ptr") else self.ctx.to_device(X)
        Yd = Y if hasattr(Y, "data_ptr") else self.ctx.to_device(Y)
        n = Xd.shape[0]
        self._check_targets_shape(Yd, n)
        self._ensure(n, S)
        self._upload_if_needed()
        zp, zst, keep = self._zs_args(zs, S, n)
        gfl = int(grad_from_layer) if with_grad else 0
        if gfl != getattr(self, "_grad_first", 0):
            _lib.check(self.lib.dsdgp_model_set_grad_first_layer(self.model, gfl))
            self._grad_first = gfl
        qo = bool(grad_q_only) and bool(with_grad)
        if qo != getattr(self, "_grad_q_only", False):
            _lib.check(self.lib.dsdgp_model_set_grad_q_only(self.model, int(qo)))
            self._grad_q_only = qo
        _lib.check(self.lib.dsdgp_model_elbo(self.model, ptr(Xd), ptr(Yd), n, S, zp, zst, C.c_uint64(seed),
                                             float(data_scale), float(kl_weight), int(with_grad), ptr(self.out4)))
        if not sync:
            return None
        self.ctx.sync()
        out = self.out4.cpu().numpy()
        if out[3] != 0.0:
            raise _lib.CholeskyError(f"Cholesky decomposition was not successful (Kuu pivot {int(out[3])})")
        return out

    def adam_step(self, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
        # the library refuses a pruned gradient: count the step only once it has been taken
        _lib.check(self.lib.dsdgp_model_adam_step(self.model, lr, beta1, beta2, eps, self.adam_t + 1))
        self.adam_t += 1
        self._dev_dirty = True
        self._needs_prepare = True

    def train_step(self, X, Y, S, zs=None, seed=0, data_scale=1.0, kl_weight=1.0, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
        """ELBO + gradient + Adam update in one library call (single-process training); asynchronous, result scalars in out4."""
        Xd = X if hasattr(X, "data_ptr") else self.ctx.to_device(X)
        Yd = Y if hasattr(Y, "data_ptr") else self.ctx.to_device(Y)
        n = Xd.shape[0]
        self._check_targets_shape(Yd, n)
        self._ensure(n, S)
        self._upload_if_needed()
        zp, zst, keep = self._zs_args(zs, S, n)
        if getattr(self, "_grad_first", 0) != 0:
            _lib.check(self.lib.dsdgp_model_set_grad_first_layer(self.model, 0))
            self._grad_first = 0
        if getattr(self, "_grad_q_only", False):
            _lib.check(self.lib.dsdgp_model_set_grad_q_only(self.model, 0))
            self._grad_q_only = False
        _lib.check(self.lib.dsdgp_model_train_step(self.model, ptr(Xd), ptr(Yd), n, S, zp, zst, C.c_uint64(seed), float(data_scale),
                                                   float(kl_weight), lr, beta1, beta2, eps, self.adam_t + 1, ptr(self.out4)))
        self.adam_t += 1
        self._dev_dirty = True
        self._needs_prepare = True

    def train_step_minibatch(self, Xall, Yall, idx, idx_offset, n, S, seed=0, data_scale=1.0, kl_weight=1.0, lr=0.01, beta1=0.9,
                             beta2=0.999, eps=1e-8):
        """train_step on rows idx[idx_offset : idx_offset + n] of the device-resident data (gather inside the step's first launch)."""
        self._check_targets_shape(Yall, Yall.shape[0])
        if Xall.shape[0] != Yall.shape[0]:
            raise ValueError(f"X has {Xall.shape[0]} rows, Y {Yall.shape[0]}")
        if idx_offset < 0 or idx_offset + n > idx.numel():
            raise ValueError(f"index span [{idx_offset}, {idx_offset + n}) outside the index buffer of {idx.numel()} entries")
        self._ensure(n, S)
        self._upload_if_needed()
        if getattr(self, "_grad_first", 0) != 0:
            _lib.check(self.lib.dsdgp_model_set_grad_first_layer(self.model, 0))
            self._grad_first = 0
        if getattr(self, "_grad_q_only", False):
            _lib.check(self.lib.dsdgp_model_set_grad_q_only(self.model, 0))
            self._grad_q_only = False
        _lib.check(self.lib.dsdgp_model_train_step_minibatch(self.model, ptr(Xall), ptr(Yall), ptr(idx), int(idx_offset), int(n), S,
                                                             C.c_uint64(seed), float(data_scale), float(kl_weight), lr, beta1, beta2,
                                                             eps, self.adam_t + 1, ptr(self.out4)))
        self.adam_t += 1
        self._dev_dirty = True
        self._needs_prepare = True

    def _check_targets_shape(self, Yd, n):
        """Y must be (n, D_out of the last layer) for the element-wise likelihoods, (n, 1) labels for MultiClass: the likelihood
        kernels index Y[(row % n) * DY + d] and would read out of bounds otherwise."""
        want = 1 if isinstance(self.likelihood, MultiClass) else self.layers[-1].num_outputs
        if tuple(Yd.shape) != (n, want):
            raise ValueError(f"Y has shape {tuple(Yd.shape)}, expected {(n, want)} (rows of X x outputs of the last layer)")

    def natgrad_step(self, l, gamma, check=True):
        """[UPSTREAM] NatGradOptimizer(gamma) step on layer l's (q_mu, q_sqrt) from the gradient of the last
        elbo(with_grad=True)."""
        info = C.c_int(0)
        _lib.check(self.lib.dsdgp_model_natgrad_step(self.model, l, float(gamma), C.byref(info) if check else None))
        self._dev_dirty = True
        self._needs_prepare = True

    def set_sample_weights(self, w_dev):
        """DGP_Quad (dgp.py:160-166): weight the S propagated samples by w (device tensor, kept alive here) instead of 1/S;
        None restores the Monte-Carlo mean."""
        self._sample_w = w_dev
        _lib.check(self.lib.dsdgp_model_set_sample_weights(self.model, ptr(w_dev) if w_dev is not None else None,
                                                           int(w_dev.numel()) if w_dev is not None else 0))

    def layer_kl(self, l):
        self._prepare_checked()
        out = self.ctx.empty(1)
        _lib.check(self.lib.dsdgp_model_layer_kl(self.model, l, ptr(out)))
        self.ctx.sync()
        return float(out.cpu().numpy()[0])

    def layer_conditional(self, l, X):
        Xd = X if hasattr(X, "data_ptr") else self.ctx.to_device(X)
        n = Xd.shape[0]
        self._ensure(max(n, 1), 1)
        self._prepare_checked()
        D = self.layers[l].num_outputs
        mean, var = self.ctx.empty(n, D), self.ctx.empty(n, D)
        _lib.check(self.lib.dsdgp_model_layer_conditional(self.model, l, ptr(Xd), n, ptr(mean), ptr(var)))
        return mean, var

    def randn(self, shape, seed=None):
        """N(0,1) draws from the device Philox generator (tf.random_normal, layers.py:101-102) as a numpy array."""
        n = int(np.prod(shape))
        out = self.ctx.empty(n)
        self._rng_calls = getattr(self, "_rng_calls", 0) + 1
        _lib.check(self.lib.dsdgp_randn(self.ctx.handle, C.c_uint64(0x5eed if seed is None else seed),
                                        C.c_uint64(1 << 32 | self._rng_calls), n, ptr(out)))
        self.ctx.sync()
        return out.cpu().numpy().reshape(shape)

    def layer_conditional_full(self, l, X):
        """conditional_ND(X, full_cov=True): mean (n, D), var (n, n, D) (layers.py:206-209)."""
        Xd = X if hasattr(X, "data_ptr") else self.ctx.to_device(X)
        n = Xd.shape[0]
        self._ensure(max(n, 1), 1)
        self._prepare_checked()
        D = self.layers[l].num_outputs
        mean, var = self.ctx.empty(n, D), self.ctx.empty(n, n, D)
        _lib.check(self.lib.dsdgp_model_layer_conditional_full(self.model, l, ptr(Xd), n, ptr(mean), ptr(var)))
        return mean, var

    def gradient_dict(self):
        """d loss / d (unconstrained) parameters of the last elbo(with_grad=True), keyed like oracle/model.py."""
        self.ctx.sync()
        g = self.grad.cpu().numpy()
        out = {}
        names = {}
        for l, layer in enumerate(self.layers):
            stat, wk = self._params_of(layer)
            names[id(layer.feature.Z)] = f"l{l}.Z"
            names[id(layer.q_mu)] = f"l{l}.q_mu"
            names[id(layer.q_sqrt)] = f"l{l}.q_sqrt"
            names[id(stat.variance)] = f"l{l}.kern_variance_raw"
            names[id(stat.lengthscales)] = f"l{l}.kern_lengthscales_raw"
            if wk is not None:
                names[id(wk.variance)] = f"l{l}.white_variance_raw"
            if layer.mean_function.kind == "linear":
                names[id(layer.mean_function.A)] = f"l{l}.mean_A"
                names[id(layer.mean_function.b)] = f"l{l}.mean_b"
        if isinstance(self.likelihood, Gaussian):
            names[id(self.likelihood.variance)] = "lik_variance_raw"
        if isinstance(self.likelihood, (StudentT, Beta)):
            names[id(self.likelihood.scale)] = "lik_variance_raw"
        if isinstance(self.likelihood, Gamma):
            names[id(self.likelihood.shape)] = "lik_variance_raw"
        for p, off, cnt, kind in self.entries:
            out[names[id(p)]] = g[off:off + cnt].reshape(p.shape).copy()
        return out
