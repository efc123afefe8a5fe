"""ctypes binding of libdsdgp.so (include/dsdgp.h).  There is NO CPU fallback: if the HIP library is missing or no
MI355X is visible, importing the compute path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSDGP_LIB_PATH: development override (A/B builds of the kernels); the shipped library is csrc/libdsdgp.so
_LIB_PATH = os.environ.get("DSDGP_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libdsdgp.so")

DSDGP_MAX_LAYERS = 16
KERN_RBF, KERN_MATERN52 = 0, 1
MEAN_ZERO, MEAN_IDENTITY, MEAN_LINEAR = 0, 1, 2
LIK_GAUSSIAN, LIK_MULTICLASS, LIK_BERNOULLI, LIK_POISSON, LIK_EXPONENTIAL, LIK_STUDENT_T, LIK_GAMMA, LIK_BETA = 0, 1, 2, 3, 4, 5, 6, 7
ERR_NOT_SPD = -2
ERR_RCCL = -6

c_double_p = C.POINTER(C.c_double)


class DsdgpError(RuntimeError):
    pass


class CholeskyError(DsdgpError):
    """mirrors TF's InvalidArgumentError 'Cholesky decomposition was not successful' (layers.py:172)"""


class KernelSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("input_dim", C.c_int32), ("ard", C.c_int32), ("has_white", C.c_int32),
                ("variance", C.c_double), ("white_variance", C.c_double), ("lengthscales", c_double_p)]


class LayerDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("D_in", C.c_int32), ("D_out", C.c_int32),
                ("kern_kind", C.c_int32), ("ard", C.c_int32), ("has_white", C.c_int32),
                ("mean_kind", C.c_int32),
                ("trainable_Z", C.c_int32), ("trainable_q_mu", C.c_int32), ("trainable_q_sqrt", C.c_int32),
                ("trainable_kvar", C.c_int32), ("trainable_kls", C.c_int32), ("trainable_wvar", C.c_int32),
                ("input_prop_dim", C.c_int32),
                ("trainable_mean_A", C.c_int32), ("trainable_mean_b", C.c_int32),
                ("kvar_identity", C.c_int32), ("reserved0", C.c_int32),
                ("mean_A", C.c_void_p),
                ("off_Z", C.c_int64), ("off_q_mu", C.c_int64), ("off_q_sqrt", C.c_int64),
                ("off_kvar", C.c_int64), ("off_kls", C.c_int64), ("off_wvar", C.c_int64),
                ("off_mean_A", C.c_int64), ("off_mean_b", C.c_int64)]


class ModelDesc(C.Structure):
    _fields_ = [("L", C.c_int32), ("white", C.c_int32), ("lik_kind", C.c_int32), ("num_classes", C.c_int32),
                ("trainable_lik_var", C.c_int32), ("reserved", C.c_int32),
                ("jitter", C.c_double), ("lik_aux", C.c_double), ("off_lik_var", C.c_int64), ("n_theta", C.c_int64),
                ("layers", LayerDesc * DSDGP_MAX_LAYERS)]


_lib = None

_PROTOS = {
    "dsdgp_version": (C.c_int, []),
    "dsdgp_sizeof_layer_desc": (C.c_int, []),
    "dsdgp_sizeof_model_desc": (C.c_int, []),
    "dsdgp_last_error": (C.c_char_p, []),
    "dsdgp_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "dsdgp_ctx_destroy": (C.c_int, [C.c_void_p]),
    "dsdgp_sync": (C.c_int, [C.c_void_p]),
    "dsdgp_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dsdgp_prof_read": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "dsdgp_launch_count": (C.c_int64, []),
    "dsdgp_gram": (C.c_int, [C.c_void_p, C.POINTER(KernelSpec), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                             C.c_double, C.c_void_p, C.c_int64]),
    "dsdgp_potrf": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int)]),
    "dsdgp_trsm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "dsdgp_trsm_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_int64]),
    "dsdgp_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                             C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64]),
    "dsdgp_model_workspace_bytes": (C.c_int, [C.POINTER(ModelDesc), C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "dsdgp_model_create": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "dsdgp_model_destroy": (C.c_int, [C.c_void_p]),
    "dsdgp_model_prepare": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dsdgp_model_propagate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dsdgp_model_elbo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int64), C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "dsdgp_model_adam_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "dsdgp_model_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_int64), C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int64, C.c_void_p]),
    "dsdgp_model_train_step_minibatch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                                   C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                                   C.c_int64, C.c_void_p]),
    "dsdgp_model_set_bucket_callback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsdgp_model_set_sample_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "dsdgp_model_natgrad_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_int)]),
    "dsdgp_model_layer_kl": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dsdgp_model_layer_conditional": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dsdgp_model_layer_conditional_full": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dsdgp_reparameterize_full": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_int32,
                                            C.c_int32, C.c_void_p]),
    "dsdgp_reparameterize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p]),
    "dsdgp_randn": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "dsdgp_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "dsdgp_gather_rows2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_int64]),
    "dsdgp_gauss_var_exp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_double, C.c_void_p, C.c_void_p]),
    "dsdgp_gauss_predict_density": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                              C.c_int32, C.c_double, C.c_void_p]),
    "dsdgp_add_scalar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p]),
    "dsdgp_multiclass_var_exp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_int, C.c_void_p, C.c_void_p]),
    "dsdgp_multiclass_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dsdgp_bernoulli_var_exp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int, C.c_void_p, C.c_void_p]),
    "dsdgp_bernoulli_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dsdgp_lik_var_exp": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "dsdgp_lik_predict": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    "dsdgp_model_set_grad_first_layer": (C.c_int, [C.c_void_p, C.c_int32]),
    "dsdgp_model_set_grad_q_only": (C.c_int, [C.c_void_p, C.c_int32]),
    "dsdgp_model_track_theta": (C.c_int, [C.c_void_p, C.c_int]),
    "dsdgp_model_theta_changed": (C.c_int, [C.c_void_p]),
    "dsdgp_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
}

BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p)

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())


def lib_path():
    return _LIB_PATH


def load():
    """dlopen libdsdgp.so and attach prototypes.  Raises if the library was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise DsdgpError(f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc == 0:
        return
    msg = load().dsdgp_last_error().decode("utf-8", "replace")
    if rc == ERR_NOT_SPD:
        raise CholeskyError(msg)
    raise DsdgpError(f"libdsdgp error {rc}: {msg}")
