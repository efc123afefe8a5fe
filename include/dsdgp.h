/*
 * dsdgp.h — C-ABI of the MI355X-native doubly-stochastic DGP hot path (libdsdgp.so, gfx950).
 *
 * The reference (UCL-SML/Doubly-Stochastic-DGP) has NO FFI / plugin interface: its hot path is reached only
 * through Python classes that compose GPflow/TF ops.  Each entry point below therefore cites the reference
 * *call site* (file:line under /root/reference) whose TF/GPflow op sequence it replaces.  A maintainer binds
 * these with ctypes; the shipped host mirror lives in
 * doubly-stochastic-dgp_amd/doubly_stochastic_dgp/.
 *
 * Conventions
 *   - every function returns int: 0 = DSDGP_OK, <0 = dsdgp_status; dsdgp_last_error() gives a thread-local
 *     message (the reference surfaces errors as Python exceptions from session.run, e.g. "Cholesky
 *     decomposition was not successful" — the host mirror re-raises DSDGP_ERR_NOT_SPD the same way).
 *   - all data pointers are DEVICE pointers to row-major float64 unless marked "host"; the caller owns every
 *     buffer (including workspaces); the library never frees caller memory and keeps model-bound pointers only
 *     until dsdgp_model_destroy.
 *   - work is enqueued asynchronously on the ctx stream; dsdgp_sync() waits for it.
 *   - all arithmetic is float64 (settings.float_type, layers.py:68).
 */
#ifndef DSDGP_H
#define DSDGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsdgp_ctx dsdgp_ctx;
typedef struct dsdgp_model dsdgp_model;

typedef enum {
  DSDGP_OK = 0,
  DSDGP_ERR_BAD_ARG = -1,
  DSDGP_ERR_NOT_SPD = -2,      /* tf.cholesky failure, layers.py:172 */
  DSDGP_ERR_HIP = -3,
  DSDGP_ERR_UNSUPPORTED = -4,
  DSDGP_ERR_WORKSPACE = -5,
  DSDGP_ERR_RCCL = -6          /* dsdgp_allreduce: RCCL missing or a collective failed */
} dsdgp_status;

enum { DSDGP_KERN_RBF = 0, DSDGP_KERN_MATERN52 = 1 };            /* [UPSTREAM] gpflow.kernels */
enum { DSDGP_MEAN_ZERO = 0, DSDGP_MEAN_IDENTITY = 1, DSDGP_MEAN_LINEAR = 2 }; /* layer_initializations.py:30-42,51 */
enum { DSDGP_LIK_GAUSSIAN = 0, DSDGP_LIK_MULTICLASS = 1, DSDGP_LIK_BERNOULLI = 2,     /* dgp.py:57, utils.py:54-93,
                                                                                       * tests/test_dgp.py:48-54 */
       /* further likelihoods utils.py:54-121 can wrap ([UPSTREAM] gpflow 1.1.1 likelihoods.py), exp links / StudentT: */
       DSDGP_LIK_POISSON = 3, DSDGP_LIK_EXPONENTIAL = 4, DSDGP_LIK_STUDENT_T = 5, DSDGP_LIK_GAMMA = 6, DSDGP_LIK_BETA = 7 };

#define DSDGP_MAX_LAYERS 16

/* ---- context ---------------------------------------------------------------------------------------- */
int dsdgp_version(void);
/* sizeof(dsdgp_layer_desc) / sizeof(dsdgp_model_desc) as compiled: a binding checks its struct mirrors against these. */
int dsdgp_sizeof_layer_desc(void);
int dsdgp_sizeof_model_desc(void);
const char* dsdgp_last_error(void);
/* stream: a hipStream_t (as void*) to enqueue on, or NULL for a library-owned stream. */
int dsdgp_ctx_create(dsdgp_ctx** out, int device, void* stream);
int dsdgp_ctx_destroy(dsdgp_ctx* ctx);
int dsdgp_sync(dsdgp_ctx* ctx);
/* HIP-event timing of the most recent launch of a named kernel class on the ctx stream (bench.py roofline):
 * enable, run, then read the accumulated milliseconds and launch count. name in
 * {"layer_fwd","layer_bwd","wgrad","gram","potrf","gemm"}. */
int dsdgp_prof_enable(dsdgp_ctx* ctx, int on);
int dsdgp_prof_read(dsdgp_ctx* ctx, const char* name, double* total_ms, int64_t* launches, int reset);
/* Kernel launches this library has enqueued in this process so far (all contexts; memsets / copies not counted): the difference
 * around a loop of steps is that loop's launches per step (bench.py `launches_per_step`).  No reference counterpart: measurement aid. */
int64_t dsdgp_launch_count(void);

/* ---- primitives (north_star: Gram assembly, blocked Cholesky, trsm) ----------------------------------- */
/* Kernel hyper-parameters, host side.  lengthscales: host pointer to 1 (ard=0) or input_dim (ard=1) values. */
typedef struct {
  int32_t kind;            /* DSDGP_KERN_* */
  int32_t input_dim;
  int32_t ard;
  int32_t has_white;       /* k + White(white_variance): only K(X) and Kdiag see it */
  double variance;
  double white_variance;
  const double* lengthscales; /* host */
} dsdgp_kernel;

/* K1/K3 — feature.Kuu(kern, jitter) / feature.Kuf(kern, X) (layers.py:171,184 -> [UPSTREAM] kern.K):
 *   X2 == NULL : out[n,n]  = k(X,X) + (white_variance + jitter) I          (symmetric)
 *   else       : out[n,n2] = k(X,X2)                                       (White contributes 0)   */
int dsdgp_gram(dsdgp_ctx* ctx, const dsdgp_kernel* kern, const double* X, int64_t n, const double* X2, int64_t n2,
               double jitter, double* out, int64_t ld_out);

/* K2 — tf.cholesky (layers.py:172): in-place lower Cholesky of `batch` n x n SPD matrices (upper part zeroed).
 * info (host, may be NULL): 0 ok, i>0 = first non-positive pivot (1-based) in some batch entry.
 * Blocked right-looking factorisation; the trailing update is an fp64-MFMA syrk/gemm. */
int dsdgp_potrf(dsdgp_ctx* ctx, int batch, int n, double* A, int64_t lda, int64_t stride, int* info);

/* K4 — tf.matrix_triangular_solve(L, B, lower=True) and its transposed form (layers.py:186,188):
 *   trans=0: B <- L^{-1} B ;  trans=1: B <- L^{-T} B.   L: n x n lower (only the lower triangle is read), B: n x nrhs, in place.
 * Blocked: 16 x 16 diagonal-block inverses, 128-row panels solved one wave per 16 right-hand-side columns (MFMA products chained in
 * registers), MFMA GEMM updates of the rows below / above a panel.  n^2 nrhs flops; any n, any nrhs; asynchronous. */
int dsdgp_trsm(dsdgp_ctx* ctx, int trans, int n, int64_t nrhs, const double* L, int64_t ldl, double* B, int64_t ldb);
/* The batched form of layers.py:239, tf.matrix_triangular_solve(self.Lu_tiled, self.q_sqrt): `batch` right-hand-side matrices
 * strideB doubles apart, their L matrices strideL apart — strideL = 0: ONE L shared by the batch (the reference tiles Lu D_out
 * times, layers.py:173; nothing is tiled here). */
int dsdgp_trsm_batched(dsdgp_ctx* ctx, int trans, int n, int64_t nrhs, int batch, const double* L, int64_t ldl, int64_t strideL,
                       double* B, int64_t ldb, int64_t strideB);

/* plain fp64-MFMA GEMM used by the M x M algebra: C = alpha op(A) op(B) + beta C (row-major). */
int dsdgp_gemm(dsdgp_ctx* ctx, int transA, int transB, int m, int n, int k, double alpha, const double* A,
               int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);

/* ---- model: DGP_Base / SVGP_Layer (dgp.py:42-126, layers.py:122-246) ----------------------------------- */
typedef struct {
  int32_t M, D_in, D_out;          /* SVGP_Layer.__init__ layers.py:123-165 */
  int32_t kern_kind, ard, has_white;
  int32_t mean_kind;               /* DSDGP_MEAN_* */
  int32_t trainable_Z, trainable_q_mu, trainable_q_sqrt, trainable_kvar, trainable_kls, trainable_wvar;
  int32_t input_prop_dim;          /* Layer(input_prop_dim) layers.py:36-50,105-117: the next layer sees [X[:, :p] | samples] */
  int32_t trainable_mean_A, trainable_mean_b;   /* only meaningful when the corresponding off_mean_* >= 0 */
  int32_t kvar_identity;           /* 1: theta[off_kvar] IS the kernel variance (a Parameter without transform — the reference's own
                                      tests build such a kernel to reach variance 1e-24, tests/test_dgp.py:79-85); 0: softplus + 1e-6 */
  int32_t reserved0;
  const double* mean_A;            /* device, (D_in x D_out) for DSDGP_MEAN_LINEAR when fixed (layer_initializations.py:41-42);
                                      ignored when off_mean_A >= 0 */
  /* offsets (in doubles) into the flat unconstrained parameter vector theta: */
  int64_t off_Z;                   /* (M, D_in)                       feature.Z           layers.py:153 */
  int64_t off_q_mu;                /* (M, D_out)                      layers.py:146-147 */
  int64_t off_q_sqrt;              /* (D_out, M, M) dense; tril part is the free variable  layers.py:149-151 */
  int64_t off_kvar;                /* scalar, softplus^-1(variance)   [UPSTREAM] transforms.positive (the variance itself if kvar_identity) */
  int64_t off_kls;                 /* 1 or D_in values, softplus^-1(lengthscales) */
  int64_t off_wvar;                /* scalar (if has_white) */
  /* [UPSTREAM] mean_functions.Linear(A, b) as a free parameter (a user-supplied final mean function, dgp.py:187): */
  int64_t off_mean_A;              /* (D_in, D_out) row-major inside theta, or -1: use the fixed mean_A pointer */
  int64_t off_mean_b;              /* (D_out) inside theta, or -1: no bias */
} dsdgp_layer_desc;

typedef struct {
  int32_t L;
  int32_t white;                   /* DGP(..., white=False) dgp.py:187 */
  int32_t lik_kind;
  int32_t num_classes;
  int32_t trainable_lik_var;
  int32_t reserved;
  double jitter;                   /* settings.jitter, layers.py:171, utils.py:41 */
  double lik_aux;                  /* the likelihood's constant: Poisson.binsize, StudentT.deg_free (else ignored) */
  int64_t off_lik_var;             /* scalar, softplus^-1 of the likelihood's positive parameter: Gaussian.variance, StudentT.scale,
                                      Gamma.shape, Beta.scale */
  int64_t n_theta;
  dsdgp_layer_desc layers[DSDGP_MAX_LAYERS];
} dsdgp_model_desc;

/* Workspace needed for minibatches of up to n_max rows and s_max samples. */
int dsdgp_model_workspace_bytes(const dsdgp_model_desc* desc, int64_t n_max, int32_t s_max, int64_t* bytes);
/* theta/grad/adam_m/adam_v: device, n_theta doubles each, 16-byte aligned (DSDGP_ERR_BAD_ARG otherwise: the optimiser reads them as
 * pairs); grad/adam_* may be NULL for predict-only models.  workspace: 256-byte aligned. */
int dsdgp_model_create(dsdgp_ctx* ctx, const dsdgp_model_desc* desc, int64_t n_max, int32_t s_max, double* theta,
                       double* grad, double* adam_m, double* adam_v, void* workspace, int64_t workspace_bytes,
                       dsdgp_model** out);
int dsdgp_model_destroy(dsdgp_model* m);

/* build_cholesky_if_needed for every layer (layers.py:167-175): Ku, Lu (+ Lu^-1, Ku^-1) from the current theta.
 * Must precede propagate / layer calls after theta changed; dsdgp_model_elbo calls it itself.
 * info (host, may be NULL) is filled after an implicit sync with the Cholesky status. */
int dsdgp_model_prepare(dsdgp_model* m, int* info);

/* DGP_Base.propagate (dgp.py:61-76) with full_cov=False.
 *   X (n x D_in0) device. zs: host array of L device pointers (or NULL entries / NULL array): explicit N(0,1) draws
 *   for layer l; element (s,i,d) is read at zs[l][s*zstride[3l] + i*zstride[3l+1] + d*zstride[3l+2]] (strides in
 *   doubles, 0 = broadcast; dgp.py:62,68 "zs" injection).  NULL -> on-device Philox draws from `seed`.
 *   Fs/Fmeans/Fvars: host arrays of L device pointers ((S*n) x D_out_l each, row (s*n+i)) or NULL / NULL entries. */
int dsdgp_model_propagate(dsdgp_model* m, const double* X, int64_t n, int32_t S, const double* const* zs,
                          const int64_t* zstride, uint64_t seed, double* const* Fs, double* const* Fmeans,
                          double* const* Fvars);

/* DGP_Base._build_likelihood (dgp.py:92-98) on an explicit minibatch, optionally with the reverse-mode gradient of
 * loss = -(data_scale * sum_n E_log_p_Y - kl_weight * sum_l KL_l) w.r.t. theta written to `grad`
 * (stands in for tf.gradients [UPSTREAM]).  data_scale = num_data / n_global (dgp.py:96-97); kl_weight = 1, or
 * 1/world_size when the row-sharded data-parallel all-reduce sums ranks.
 * out (device, 4 doubles): [elbo_local, data_term (scaled), kl_weight * KL_sum, potrf_info (0 ok, else 1-based failing pivot)].
 * Every entry but the last sums over data-parallel ranks to the single-process value; potrf_info is identical on all ranks
 * (parameters are replicated), i.e. the sum is world_size * pivot. */
int dsdgp_model_elbo(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S,
                     const double* const* zs, const int64_t* zstride, uint64_t seed, double data_scale,
                     double kl_weight, int with_grad, double* out);

/* DGP_Quad (dgp.py:129-166): replace the Monte-Carlo mean over the S propagated samples of dsdgp_model_elbo by the
 * weighted sum  sum_s w[s] (.)  (Gauss-Hermite weights, summing to one).  w: device, S doubles, borrowed until cleared;
 * NULL restores the mean.  dsdgp_model_elbo then requires its S argument to equal this S. */
int dsdgp_model_set_sample_weights(dsdgp_model* m, const double* w, int32_t S);

/* [UPSTREAM] tf.train.AdamOptimizer step on theta using grad (t counts from 1). Non-trainable entries are skipped. */
int dsdgp_model_adam_step(dsdgp_model* m, double lr, double beta1, double beta2, double eps, int64_t t);

/* One optimiser step of -ELBO in ONE call — `session.run(opt_op)` of demos/demo_regression_UCI.ipynb:324 with
 * opt_op = AdamOptimizer(lr).minimize(model): dsdgp_model_elbo(with_grad = 1) followed by the Adam update (t counts from 1), the
 * update applied by the last launch of the reverse pass instead of a launch of its own.  Same arguments and `out` as
 * dsdgp_model_elbo; `grad` holds the gradient the update used.  Single-process training only: a data-parallel step needs the
 * all-reduce between the two halves (dsdgp_model_elbo, dsdgp_allreduce, dsdgp_model_adam_step).  DSDGP_ERR_BAD_ARG while
 * dsdgp_model_set_grad_first_layer / dsdgp_model_set_grad_q_only restrict the reverse pass (the caller lifts the restriction first:
 * the step does not do it on the caller's behalf). */
int dsdgp_model_train_step(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S, const double* const* zs,
                           const int64_t* zstride, uint64_t seed, double data_scale, double kl_weight, double lr, double beta1,
                           double beta2, double eps, int64_t t, double* out);

/* dsdgp_model_train_step on the minibatch rows idx[idx_offset .. idx_offset + n) of the resident data X_all (rows x D_in of the first
 * layer) / Y_all (rows x D_Y): the two Minibatch iterators of dgp.py:51-52 and the optimiser step in one call; the gather runs inside
 * the step's first launch.  idx: device int64.  Fresh N(0,1) draws (Philox, `seed`); no explicit zs. */
int dsdgp_model_train_step_minibatch(dsdgp_model* m, const double* X_all, const double* Y_all, const int64_t* idx, int64_t idx_offset,
                                     int64_t n, int32_t S, uint64_t seed, double data_scale, double kl_weight, double lr, double beta1,
                                     double beta2, double eps, int64_t t, double* out);

/* [UPSTREAM] gpflow.training.NatGradOptimizer(gamma) step on layer l's (q_mu, q_sqrt), using the loss gradient left in
 * `grad` by the last dsdgp_model_elbo(with_grad=1) (demos/demo_regression_UCI.ipynb:360-366, tests/test_collapsed.py:100):
 * per output, natural parameters theta <- theta - gamma dL/d eta, then back to (mean, Cholesky factor).
 * info (host, may be NULL): non-zero if the updated covariance is not SPD. */
int dsdgp_model_natgrad_step(dsdgp_model* m, int32_t l, double gamma, int* info);

/* [UPSTREAM] tf.gradients(loss, var_list) with var_list = the (q_mu, q_sqrt) of the upper layer(s), as
 * NatGradOptimizer.minimize(var_list=[[last.q_mu, last.q_sqrt]]) builds it: TensorFlow prunes the reverse pass below the
 * lowest layer in var_list.  After this call dsdgp_model_elbo(with_grad=1) runs the reverse pass for layers >= first only;
 * the gradient entries of the layers below keep their previous content (dsdgp_model_adam_step then fails with
 * DSDGP_ERR_BAD_ARG until a full gradient has been evaluated again).
 * first = 0 restores the full gradient.  Ignored (full gradient) for white=True models. */
int dsdgp_model_set_grad_first_layer(dsdgp_model* m, int32_t first);

/* [UPSTREAM] the same tf.gradients(loss, var_list) when var_list holds NOTHING BUT (q_mu, q_sqrt) pairs — what
 * NatGradOptimizer.minimize always passes (demos/demo_regression_UCI.ipynb:360-366, tests/test_collapsed.py:100): TensorFlow's reverse
 * pass then never visits Kuf / Kuu of the lowest layer in var_list (their adjoints feed Z, the kernel hyper-parameters and the layers
 * below only).  on != 0: dsdgp_model_elbo(with_grad=1) leaves complete (q_mu, q_sqrt) entries for the layers >= first
 * (dsdgp_model_set_grad_first_layer) and UNDEFINED values in their other entries.  dsdgp_model_adam_step fails with DSDGP_ERR_BAD_ARG
 * until a full gradient has been evaluated again; dsdgp_model_train_step[_minibatch] (which would evaluate its own gradient under
 * the restriction) fails with DSDGP_ERR_BAD_ARG for as long as the restriction is set: call dsdgp_model_set_grad_q_only(m, 0) and
 * dsdgp_model_set_grad_first_layer(m, 0) first.  on = 0 restores the full gradient.  Ignored for white=True models. */
int dsdgp_model_set_grad_q_only(dsdgp_model* m, int32_t on);

/* Optional contract for callers that alternate optimisers (demo_regression_UCI.ipynb:360-366: one Adam step on the hyper-parameters,
 * one natural-gradient step on the last layer).  theta is caller-owned, so every evaluation normally rebuilds Ku, its Cholesky
 * factor and the inverses (layers.py:167-175 build_cholesky_if_needed).  With tracking enabled the caller promises to report its
 * own writes to theta through dsdgp_model_theta_changed; an evaluation whose only change since the previous one came from
 * dsdgp_model_natgrad_step — which leaves Z and the kernel hyper-parameters untouched — then keeps Lu, Lu^-1 and Ku^-1 and only
 * rebuilds what depends on (q_mu, q_sqrt).  dsdgp_model_adam_step always invalidates. */
int dsdgp_model_track_theta(dsdgp_model* m, int enable);
int dsdgp_model_theta_changed(dsdgp_model* m);

/* SVGP_Layer.KL (layers.py:221-246) of layer l after dsdgp_model_prepare; out: device scalar. */
int dsdgp_model_layer_kl(dsdgp_model* m, int32_t l, double* out);

/* SVGP_Layer.conditional_ND (layers.py:178-219, full_cov=False) of layer l on X (n x D_in_l):
 * mean, var: (n x D_out_l). */
int dsdgp_model_layer_conditional(dsdgp_model* m, int32_t l, const double* X, int64_t n, double* mean, double* var);

/* SVGP_Layer.conditional_ND(X, full_cov=True) (layers.py:206-209,216-219): mean (n x D_out), var (n x n x D_out). */
int dsdgp_model_layer_conditional_full(dsdgp_model* m, int32_t l, const double* X, int64_t n, double* mean, double* var);
/* utils.reparameterize, full covariance (utils.py:43-51): mean, z, out (S x n x D); var (S x n x n x D):
 * out[s,:,d] = mean[s,:,d] + chol(var[s,:,:,d] + jitter I) z[s,:,d]. */
int dsdgp_reparameterize_full(dsdgp_ctx* ctx, const double* mean, const double* var, const double* z, double jitter,
                              int64_t n, int32_t D, int32_t S, double* out);

/* utils.reparameterize, diagonal case (utils.py:40-41): out = mean + z * sqrt(var + jitter), count elements. */
int dsdgp_reparameterize(dsdgp_ctx* ctx, const double* mean, const double* var, const double* z, double jitter,
                         int64_t count, double* out);

/* N(0,1) float64 draws (Philox4x32-10 + Box–Muller), replaces tf.random_normal (layers.py:101-102). */
int dsdgp_randn(dsdgp_ctx* ctx, uint64_t seed, uint64_t stream, int64_t count, double* out);

/* Bucketed exchange of the row-sharded ELBO's gradient.  With a callback set, dsdgp_model_elbo(with_grad = 1) finishes
 * every layer's gradient right behind that layer's weight-gradient products — reduction, products, assembly, hyper-parameter sums —
 * and calls `fn(user, bucket, ptr, count, stream)` on the calling host thread as soon as the kernels that produce the bucket are
 * ENQUEUED: bucket = L-1, L-2, ..., 0 (layer l's contiguous segment of `grad`), then bucket = L (likelihood-variance entry + the four
 * result scalars when `out` = grad + n_theta; else the entry, then bucket L+1 = the scalars).  `stream` is the HIP stream the segment is
 * produced on: a collective enqueued there (ncclAllReduce(ptr, ptr, count, ncclDouble, ncclSum, comm, stream), or a torch.distributed
 * call under torch.cuda.ExternalStream(stream)) is ordered behind its producers and runs UNDER the backward chains of the layers
 * below.  The caller makes its optimiser step wait for the collectives it issued.  fn = NULL restores the single flat buffer
 * (one dsdgp_allreduce).  Ignored for white=True / wide-input models and while dsdgp_model_set_grad_first_layer restricts the pass. */
typedef void (*dsdgp_bucket_fn)(void* user, int32_t bucket, double* ptr, int64_t count, void* stream);
int dsdgp_model_set_bucket_callback(dsdgp_model* m, dsdgp_bucket_fn fn, void* user);

/* Multi-GPU exchange step of the row-sharded ELBO (the step being sharded is dgp.py:92-98): in-place SUM of
 * `count` doubles of `buf` over the ranks of `comm` on the ctx stream — ONE call per training step on the flat
 * [gradient (n_theta) | elbo, data term, kl_weight * KL, potrf_info] buffer (`grad` of dsdgp_model_create with `out` of
 * dsdgp_model_elbo pointing at grad + n_theta), between dsdgp_model_elbo(with_grad = 1, kl_weight = 1 / world,
 * data_scale = num_data / (n_local * world)) and dsdgp_model_adam_step.
 * comm: an RCCL communicator (ncclComm_t, one rank per GPU / process) created by the CALLER with RCCL's own
 * ncclGetUniqueId / ncclCommInitRank; the library resolves ncclAllReduce from the RCCL already loaded in the process (or
 * librccl.so) at first use, so libdsdgp.so itself carries no link-time dependency on RCCL.  Asynchronous.
 * Returns DSDGP_ERR_RCCL when RCCL is unavailable or reports an error (see dsdgp_last_error). */
int dsdgp_allreduce(dsdgp_ctx* ctx, void* comm, double* buf, int64_t count);

/* Minibatch gather ([UPSTREAM] gpflow.params.Minibatch, dgp.py:51-52): dst[i,:] = src[idx[idx_offset+i],:], cols wide.
 * idx: device int64.  Index generation stays on the host (own RNG; TF's shuffle order is not reproducible). */
int dsdgp_gather_rows(dsdgp_ctx* ctx, const double* src, int64_t cols, const int64_t* idx, int64_t n, int64_t idx_offset,
                      double* dst);
/* The X and Y minibatches of one step in one launch (the two Minibatch iterators of dgp.py:51-52 share their seed). */
int dsdgp_gather_rows2(dsdgp_ctx* ctx, const double* srcX, int64_t colsX, double* dstX, const double* srcY, int64_t colsY,
                       double* dstY, const int64_t* idx, int64_t n, int64_t idx_offset);

/* DGP_Base.E_log_p_Y with the Gaussian likelihood (dgp.py:83-90): out[i,d] = mean_s varexp(mean[s,i,d], var[s,i,d], Y[i,d]).
 * mean/var: (S*n x DY); Y, out: (n x DY); lik_var: host scalar.
 * sample_w (device, S doubles, may be NULL): DGP_Quad.E_log_p_Y (dgp.py:160-166) — out = sum_s sample_w[s] varexp_s
 * instead of the mean over s. */
int dsdgp_gauss_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int32_t S,
                        int32_t DY, double lik_var, const double* sample_w, double* out);
/* DGP_Base.predict_density, Gaussian (dgp.py:121-126): out[i,d] = logsumexp_s log N(Y | mean, var + lik_var) - log S. */
int dsdgp_gauss_predict_density(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n,
                                int32_t S, int32_t DY, double lik_var, double* out);
/* [UPSTREAM] MultiClass(K)/RobustMax through BroadcastingLikelihood (utils.py:76-93):
 * mode 0: out[i] = mean_s variational expectation ; mode 1: out[i] = logsumexp_s log density - log S.
 * mean/var: (S*n x K); Y: (n x 1) class labels stored as doubles; out: (n x 1). */
int dsdgp_multiclass_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int32_t S,
                             int32_t K, int mode, const double* sample_w, double* out);
/* MultiClass.predict_mean_and_var: out_mean[r,k] = predictive class probability, out_var = p - p^2; R rows. */
int dsdgp_multiclass_predict(dsdgp_ctx* ctx, const double* mean, const double* var, int64_t R, int32_t K, double* out_mean,
                             double* out_var);

/* [UPSTREAM] Bernoulli() (probit link, 20-point Gauss-Hermite variational expectations) through BroadcastingLikelihood
 * (utils.py:76-86; exercised by /root/reference/tests/test_dgp.py:48-54).  Targets: 1 selects p, anything else 1 - p.
 * mode 0: out[i,d] = mean_s (or sum_s sample_w[s]) variational expectation ; mode 1: out[i,d] = logsumexp_s log density - log S.
 * mean/var: (S x n x DY); Y, out: (n x DY). */
int dsdgp_bernoulli_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int32_t S,
                            int32_t DY, int mode, const double* sample_w, double* out);
/* Bernoulli.predict_mean_and_var (probit closed form): out_mean = probit(mean / sqrt(1 + var)), out_var = p - p^2. */
int dsdgp_bernoulli_predict(dsdgp_ctx* ctx, const double* mean, const double* var, int64_t count, double* out_mean,
                            double* out_var);

/* BroadcastingLikelihood over Poisson(invlink=exp, binsize) / Exponential(invlink=exp) / Gamma(invlink=exp; shape) /
 * StudentT(scale, deg_free) / Beta(invlink=probit, scale)
 * (utils.py:76-93,95-121; [UPSTREAM] gpflow 1.1.1 likelihoods: closed-form variational expectations with the exp link, the base
 * class's 20-point Gauss-Hermite rule otherwise).  kind = DSDGP_LIK_POISSON .. DSDGP_LIK_BETA; p0 = StudentT.scale / Gamma.shape /
 * Beta.scale, p1 = Poisson.binsize / StudentT.deg_free (ignored where the likelihood has no such parameter).  Modes, shapes: as above. */
int dsdgp_lik_var_exp(dsdgp_ctx* ctx, int32_t kind, double p0, double p1, const double* mean, const double* var, const double* Y,
                      int64_t n, int32_t S, int32_t DY, int mode, const double* sample_w, double* out);
/* Likelihood.predict_mean_and_var by the same rule: E_y = sum w cm(f_k), V_y = sum w (cv(f_k) + cm(f_k)^2) - E_y^2. */
int dsdgp_lik_predict(dsdgp_ctx* ctx, int32_t kind, double p0, double p1, const double* mean, const double* var, int64_t count,
                      double* out_mean, double* out_var);

/* out = in + value (Gaussian.predict_mean_and_var adds the noise variance, dgp.py:116-119). */
int dsdgp_add_scalar(dsdgp_ctx* ctx, const double* in, double value, int64_t count, double* out);

#ifdef __cplusplus
}
#endif
#endif /* DSDGP_H */
