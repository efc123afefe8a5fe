// Shared host/device helpers of libdsdgp (gfx950 / CDNA4 only; fp64 throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/dsdgp.h"

// ------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------
void dsdgp_set_error(const char* fmt, ...);

#define DS_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      dsdgp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return DSDGP_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)

#define DS_CHECK_ARG(cond)                                                  \
  do {                                                                      \
    if (!(cond)) {                                                          \
      dsdgp_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
      return DSDGP_ERR_BAD_ARG;                                             \
    }                                                                       \
  } while (0)

// every kernel launch of the library goes through this: a process-wide count of launches (dsdgp_launch_count: launches per step in
// bench.py's line; a relaxed atomic add, nothing else)
#include <atomic>
extern std::atomic<long long> g_dsdgp_launches;
#define DS_LAUNCH(...)                                                   \
  do {                                                                   \
    g_dsdgp_launches.fetch_add(1, std::memory_order_relaxed);            \
    hipLaunchKernelGGL(__VA_ARGS__);                                     \
  } while (0)

#define DS_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != DSDGP_OK) return rc__; \
  } while (0)

// ------------------------------------------------------------------------------------------------------
// context: one per (host thread, device); owns the stream (unless borrowed) and a grow-only scratch used by the
// *primitive* entry points only (the model path runs entirely inside the caller's workspace).
// ------------------------------------------------------------------------------------------------------
struct ProfSlot {
  double ms = 0.0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct dsdgp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t side = nullptr;      // second stream of the models of this context (created with the first model)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int prof_on = 0;
  std::map<std::string, ProfSlot> prof;
  // pinned staging ring for the small host->device descriptor uploads of the primitive entry points (kernel hyper-parameters,
  // GEMM / factorisation descriptors): the copy is enqueued from pinned memory and the call returns — no stream synchronisation
  char* pin = nullptr;
  size_t pin_bytes = 0, pin_off = 0;
  // dsdgp_potrf, n >= 512: the blocked factorisation's plan (descriptor arrays + scratch on the device) of the last call, reused
  // while matrix order, batch and scratch address repeat (a plan build costs two hipMallocs and a synchronous upload: ~150 us)
  void* potrf_plan = nullptr;
  void (*potrf_plan_free)(void*) = nullptr;
  int64_t potrf_key[4] = {0, 0, 0, 0};
};

int ctx_scratch(dsdgp_ctx* ctx, size_t bytes, void** out);
// asynchronous upload of a small host object to device memory `dst` on the ctx stream (the host copy may die on return)
int ctx_upload(dsdgp_ctx* ctx, void* dst, const void* src, size_t bytes);

// RAII-ish profiling bracket: records HIP events on the ctx stream around a launch when profiling is enabled.
struct ProfScope {
  dsdgp_ctx* ctx;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st;
  ProfScope(dsdgp_ctx* c, const char* n, hipStream_t stream = nullptr) : ctx(c), name(n), st(stream ? stream : c->stream) {
    if (ctx->prof_on) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, st);
    }
  }
  ~ProfScope() {
    if (a) {
      hipEventRecord(b, st);
      ctx->prof[name].pending.push_back({a, b});
    }
  }
};

// A thread's wave index is the same in all 64 lanes of its wave, but `threadIdx.x >> 6` is a per-lane value to the compiler: loops whose
// bounds depend on it (the triangular k ranges of the chains, the row ranges of the split-K products) become DIVERGENT loops — exec-mask
// loop control, no unrolling (the "loop not unrolled" warnings), 64-bit VALU address arithmetic per load, load -> wait -> MFMA per
// k-block.  Read through an SGPR the index is uniform: scalar loop control and the requested unrolling / prefetch distance.
#ifdef __HIPCC__
#define DS_WAVE_ID(tid) __builtin_amdgcn_readfirstlane((int)(tid) >> 6)
#endif
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Padded inducing count used by every device-side M x M operand (identity pad on Ku, zero pad elsewhere).  The chain kernels
// exist for every multiple of 16 up to 128, of 32 up to 256, of 64 up to 512 and of 128 up to 1024: M = 100 (the reference
// demo size, demos/run_regression.py:57) runs as 112, not 128; M = 300 as 320, not 512.
static inline int pad_M(int M) {
  if (M <= 32) return 32;
  if (M <= 128) return (int)round_up(M, 16);
  if (M <= 256) return (int)round_up(M, 32);
  if (M <= 512) return (int)round_up(M, 64);
  return (int)round_up(M, 128);
}
// largest padded inducing count: up to 1024 the LDS-resident chains, above it the GEMM-formulated passes only
#define DSDGP_MAX_MP 2048
// Row count of the M-major per-row intermediates (A, E, GW) and of the weight-gradient results: whole 64 x 64 split-K tiles
static inline int pad_Mw(int Mp) { return (int)round_up(Mp, 64); }

// ------------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
typedef double d4 __attribute__((ext_vector_type(4)));
// Pointers that reach a kernel through a descriptor in MEMORY (job lists, GemmProblem, PotrfItem, LayerDev) have no known
// address space: the compiler emits flat_load / flat_store, which count on lgkmcnt as well as vmcnt — every LDS wait then also
// waits for outstanding global traffic.  Hot loops cast such pointers to the global (1) or LDS (3) address space.
typedef const double __attribute__((address_space(1)))* gcptr;
typedef double __attribute__((address_space(1)))* gptr;
typedef double __attribute__((address_space(3)))* lptr;

// Index of the segment a workgroup belongs to in a launch built from a device-resident list (jobs / problems with ascending start
// offsets): bisection, log2(n) dependent (scalar) loads.  The linear walk every grouped kernel started with costs one dependent L2 round
// trip per list entry — up to ~35 of them (the split-K reduction of a three-layer model) in front of a 20 us launch's last workgroups.
#define DS_FIND_SEGMENT(idx, list, n, field, key)            \
  do {                                                       \
    int lo__ = 0, hi__ = (n)-1;                              \
    while (lo__ < hi__) {                                    \
      const int mid__ = (lo__ + hi__ + 1) >> 1;              \
      if ((key) >= (list)[mid__].field) lo__ = mid__;        \
      else hi__ = mid__ - 1;                                 \
    }                                                        \
    idx = lo__;                                              \
  } while (0)

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), wave64.
//   lane l: g = l>>4, c = l&15.   A operand = A[i=c][k=g];  B operand = B[k=g][j=c];
//   D reg t (0..3) = D[row = g + 4t][col = c].
// Note the D layout of a 16x16 block *is* the B-operand layout of four consecutive k-steps (k = g + 4t), which is
// what lets the layer chain feed one GEMM's result straight into the next without touching LDS.
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- [UPSTREAM] Bernoulli likelihood, probit link (/root/reference/tests/test_dgp.py:48-54 builds it; its variational
// expectations are the base Likelihood's 20-point Gauss-Hermite rule, its predictions the probit closed form)
// probit(x) = Phi(x) (1 - 2e-3) + 1e-3
__device__ __forceinline__ double bern_probit(double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)) * (1.0 - 2e-3) + 1e-3; }
// log Bernoulli(y | p): y == 1 selects p, every other target 1 - p
__device__ __forceinline__ double bern_logp(double p, double y) { return log(y == 1.0 ? p : 1.0 - p); }
// variational expectation int log p(y | f) N(f | mu, v) df with np.polynomial.hermite.hermgauss(20) (weights / sqrt(pi));
// dmu / dv = its derivatives.  No clamp on v: a negative variance gives NaN, as upstream's sqrt does.
__device__ __forceinline__ double bern_var_exp(double mu, double v, double y, double* dmu, double* dv) {
  constexpr double GX[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323, 1.7385377121165861, 2.2549740020892757,
                             2.7888060584281305, 3.3478545673832163, 3.944764040115625, 4.603682449550744, 5.387480890011233};
  constexpr double GW[10] = {0.2607930634495549, 0.16173933398399998, 0.0615063720639769, 0.013997837447101022,
                             0.00183010313108049, 0.00012882627996192928, 4.402121090230851e-06, 6.127490259982928e-08,
                             2.4820623623151755e-10, 1.2578006724379234e-13};
  const double sd = sqrt(2.0 * v);
  const double sgn = (y == 1.0) ? 1.0 : -1.0;
  double ve = 0.0, gm = 0.0, gv = 0.0;
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    const double x = (k < 10) ? -GX[9 - k] : GX[k - 10];
    const double w = (k < 10) ? GW[9 - k] : GW[k - 10];
    const double f = mu + sd * x;
    const double p = bern_probit(f);
    const double q = (y == 1.0) ? p : 1.0 - p;
    ve += w * log(q);
    const double dl = sgn * (1.0 - 2e-3) * 0.39894228040143267794 * exp(-0.5 * f * f) / q;      // d log q / d f
    gm += w * dl;
    gv += w * dl * x;
  }
  *dmu = gm;
  *dv = gv / sd;
  return ve;
}

// ---- [UPSTREAM] further GPflow 1.1.1 likelihoods behind BroadcastingLikelihood (utils.py:54-121 wraps ANY likelihood): Poisson and
// Exponential / Gamma with the exp link, StudentT, Beta.  kind = DSDGP_LIK_*; p0 = StudentT.scale / Gamma.shape / Beta.scale,
// p1 = Poisson.binsize / StudentT.deg_free.
// digamma(x), x > 0: recurrence up to x >= 8, then the asymptotic series (error < 1e-15 there)
__device__ __forceinline__ double digamma_d(double x) {
  double r = 0.0;
  while (x < 8.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double i = 1.0 / x, i2 = i * i;
  return r + log(x) - 0.5 * i -
         i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0 - i2 * (691.0 / 32760.0 - i2 / 12.0))))));
}
// log p(y | f).  Gamma (kind 6, exp link): p0 = shape.  Beta (kind 7, the Bernoulli's probit link): p0 = scale, y clipped to [1e-6, 1 - 1e-6].
__device__ __forceinline__ double lik_logp(int kind, double f, double y, double p0, double p1) {
  if (kind == 6) return -p0 * f - lgamma(p0) + (p0 - 1.0) * log(y) - y * exp(-f);
  if (kind == 7) {
    const double mean = bern_probit(f), al = mean * p0, be = p0 - al, yc = fmin(fmax(y, 1e-6), 1.0 - 1e-6);
    return (al - 1.0) * log(yc) + (be - 1.0) * log(1.0 - yc) + lgamma(al + be) - lgamma(al) - lgamma(be);
  }
  if (kind == 3) return y * (f + log(p1)) - exp(f) * p1 - lgamma(y + 1.0);      // Poisson: y log(lam) - lam - lgamma(y + 1), lam = exp(f) binsize
  if (kind == 4) return -y * exp(-f) - f;                                       // Exponential: -y / scale - log(scale), scale = exp(f)
  const double nu = p1, z = (y - f) / p0;                                       // StudentT
  return lgamma(0.5 * (nu + 1.0)) - lgamma(0.5 * nu) - 0.5 * (log(nu) + 1.1447298858494001741) - log(p0) -
         0.5 * (nu + 1.0) * log(1.0 + z * z / nu);
}
// conditional mean / variance of y given f
__device__ __forceinline__ void lik_cond(int kind, double f, double p0, double p1, double* cm, double* cv) {
  if (kind == 3) { *cm = *cv = exp(f) * p1; return; }
  if (kind == 4) { const double e = exp(f); *cm = e; *cv = e * e; return; }
  if (kind == 6) { const double e = exp(f); *cm = p0 * e; *cv = p0 * e * e; return; }
  if (kind == 7) { const double m = bern_probit(f); *cm = m; *cv = (m - m * m) / (p0 + 1.0); return; }
  *cm = f;
  *cv = p0 * p0 * (p1 / (p1 - 2.0));
}
#define DSDGP_GH20_X {0.24534070830090124, 0.7374737285453944, 1.234076215395323, 1.7385377121165861, 2.2549740020892757, \
                      2.7888060584281305, 3.3478545673832163, 3.944764040115625, 4.603682449550744, 5.387480890011233}
#define DSDGP_GH20_W {0.2607930634495549, 0.16173933398399998, 0.0615063720639769, 0.013997837447101022, 0.00183010313108049, \
                      0.00012882627996192928, 4.402121090230851e-06, 6.127490259982928e-08, 2.4820623623151755e-10, 1.2578006724379234e-13}
// variational expectation int log p(y | f) N(f | mu, v) df and its derivatives w.r.t. mu, v and p0: the closed forms GPflow uses with
// the exp link (Poisson, Exponential), the base Likelihood's 20-point Gauss-Hermite rule (weights / sqrt(pi)) for StudentT
__device__ __forceinline__ double lik_var_exp(int kind, double mu, double v, double y, double p0, double p1, double* dmu, double* dv,
                                              double* dp0) {
  *dp0 = 0.0;
  if (kind == 3) {
    const double e = exp(mu + 0.5 * v) * p1;
    *dmu = y - e;
    *dv = -0.5 * e;
    return y * mu - e - lgamma(y + 1.0) + y * log(p1);
  }
  if (kind == 4) {
    const double e = exp(-mu + 0.5 * v) * y;
    *dmu = e - 1.0;
    *dv = -0.5 * e;
    return -e - mu;
  }
  if (kind == 6) {      // Gamma, exp link: -shape mu - lgamma(shape) + (shape - 1) log y - y exp(-mu + v / 2)
    const double e = exp(-mu + 0.5 * v) * y;
    *dmu = e - p0;
    *dv = -0.5 * e;
    *dp0 = -mu - digamma_d(p0) + log(y);
    return -p0 * mu - lgamma(p0) + (p0 - 1.0) * log(y) - e;
  }
  constexpr double GX[10] = DSDGP_GH20_X;
  constexpr double GW[10] = DSDGP_GH20_W;
  if (kind == 7) {      // Beta: quadrature of the log density, its derivatives through alpha = probit(f) scale, beta = scale - alpha
    const double sd = sqrt(2.0 * v), yc = fmin(fmax(y, 1e-6), 1.0 - 1e-6), ly = log(yc), l1y = log(1.0 - yc);
    const double lgs = lgamma(p0), dgs = digamma_d(p0);
    double ve = 0.0, gm = 0.0, gv = 0.0, gp = 0.0;
#pragma unroll 1
    for (int k = 0; k < 20; ++k) {
      const double x = (k < 10) ? -GX[9 - k] : GX[k - 10];
      const double w = (k < 10) ? GW[9 - k] : GW[k - 10];
      const double f = mu + sd * x;
      const double mean = bern_probit(f), al = mean * p0, be = p0 - al;
      ve += w * ((al - 1.0) * ly + (be - 1.0) * l1y + lgs - lgamma(al) - lgamma(be));
      const double da = digamma_d(al), db = digamma_d(be);
      const double dl = (1.0 - 2e-3) * 0.39894228040143267794 * exp(-0.5 * f * f) * p0 * (ly - l1y - da + db);      // d log p / d f
      gm += w * dl;
      gv += w * dl * x;
      gp += w * (mean * ly + (1.0 - mean) * l1y + dgs - mean * da - (1.0 - mean) * db);
    }
    *dmu = gm;
    *dv = gv / sd;
    *dp0 = gp;
    return ve;
  }
  const double sd = sqrt(2.0 * v), nu = p1;
  const double c0 = lgamma(0.5 * (nu + 1.0)) - lgamma(0.5 * nu) - 0.5 * (log(nu) + 1.1447298858494001741) - log(p0);
  double ve = 0.0, gm = 0.0, gv = 0.0, gp = 0.0;
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    const double x = (k < 10) ? -GX[9 - k] : GX[k - 10];
    const double w = (k < 10) ? GW[9 - k] : GW[k - 10];
    const double r = y - (mu + sd * x), den = nu * p0 * p0 + r * r;
    ve += w * (c0 - 0.5 * (nu + 1.0) * log(den / (nu * p0 * p0)));
    const double dl = (nu + 1.0) * r / den;                       // d log p / d f
    gm += w * dl;
    gv += w * dl * x;
    gp += w * (-1.0 / p0 + (nu + 1.0) * r * r / (p0 * den));      // d log p / d scale
  }
  *dmu = gm;
  *dv = gv / sd;
  *dp0 = gp;
  return ve;
}
// log int p(y | f) N(f | mu, v) df (predict_density) by the same rule
__device__ __forceinline__ double lik_log_density(int kind, double mu, double v, double y, double p0, double p1) {
  constexpr double GX[10] = DSDGP_GH20_X;
  constexpr double GW[10] = DSDGP_GH20_W;
  const double sd = sqrt(2.0 * v);
  double s = 0.0;
#pragma unroll 1
  for (int k = 0; k < 20; ++k) {
    const double x = (k < 10) ? -GX[9 - k] : GX[k - 10];
    const double w = (k < 10) ? GW[9 - k] : GW[k - 10];
    s += w * exp(lik_logp(kind, mu + sd * x, y, p0, p1));
  }
  return log(s);
}
// predict_mean_and_var: E_y = sum w cm(f_k), V_y = sum w (cv(f_k) + cm(f_k)^2) - E_y^2
__device__ __forceinline__ void lik_predict(int kind, double mu, double v, double p0, double p1, double* ey, double* vy) {
  constexpr double GX[10] = DSDGP_GH20_X;
  constexpr double GW[10] = DSDGP_GH20_W;
  const double sd = sqrt(2.0 * v);
  double e = 0.0, q = 0.0;
#pragma unroll 1
  for (int k = 0; k < 20; ++k) {
    const double x = (k < 10) ? -GX[9 - k] : GX[k - 10];
    const double w = (k < 10) ? GW[9 - k] : GW[k - 10];
    double cm, cv;
    lik_cond(kind, mu + sd * x, p0, p1, &cm, &cv);
    e += w * cm;
    q += w * (cv + cm * cm);
  }
  *ey = e;
  *vy = q - e * e;
}

// Cross-lane sums WITHOUT the LDS: __shfl_xor compiles to ds_bpermute_b32 pairs (an LDS round trip of 100-200 cycles per step when
// 32 waves share the CU); the per-input-dimension reductions at the end of the backward chain were six such dependent steps per
// dimension and took 23 K of the 61 K clocks of a D_out = 1 workgroup (DSDGP_BWD_TIMING, profiles/r02_chain_phases.txt, before/after in DESIGN.md 5.1).
// gfx950 has v_permlane16_swap / v_permlane32_swap (exchange odd/even 16-lane rows, upper/lower 32 lanes) and the gfx9 DPP row
// shifts / row broadcasts, all plain VALU instructions.
//
// sum over the four 16-lane groups (same c): afterwards every lane holds the total for its column c.
__device__ __forceinline__ double sum_groups(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);     // {this row | its neighbour row} in either order
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double y = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  const unsigned lo2 = __double2loint(y), hi2 = __double2hiint(y);
  const auto c = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
  const auto d = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_or_zero(double x) {      // the DPP-moved value; lanes without a source (or masked off) read 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// full 64-lane sum, returned in every lane (inclusive scan by row_shr 1/2/4/8, row_bcast 15/31, then lane 63 read back)
__device__ __forceinline__ double sum_wave(double x) {
  x += dpp_or_zero<0x111, 0xf>(x);
  x += dpp_or_zero<0x112, 0xf>(x);
  x += dpp_or_zero<0x114, 0xf>(x);
  x += dpp_or_zero<0x118, 0xf>(x);
  x += dpp_or_zero<0x142, 0xa>(x);
  x += dpp_or_zero<0x143, 0xc>(x);
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(x), 63);
  const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(x), 63);
  return __hiloint2double(hi, lo);
}

// broadcast of lane `L` (compile-time constant) of a double through two v_readlane_b32 (far cheaper than the
// ds_bpermute pair behind __shfl)
template <int L>
__device__ __forceinline__ double bcast_lane(double x) {
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(x), L);
  const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(x), L);
  return __hiloint2double((int)hi, (int)lo);
}

// hyper-parameter block layout (device doubles) produced by k_prep for every layer:
//   hyp[0]=variance hyp[1]=white_variance hyp[2]=kdiag (=variance+white) hyp[3]=sigmoid(raw var) hyp[4]=sigmoid(raw white)
//   hyp[8 + j]            = 1/lengthscale_j                   (j < D_in)
//   hyp[8 + D_in + j]     = sigmoid(raw lengthscale_j)        (chain rule of the softplus transform)
#define HYP_VAR 0
#define HYP_WVAR 1
#define HYP_KDIAG 2
#define HYP_DVAR 3
#define HYP_DWVAR 4
#define HYP_ILS 8

// stationary kernel value and d k / d r2 from the scaled squared distance r2 (variance included).
//   RBF      [UPSTREAM]: k = s2 exp(-r2/2)
//   Matern52 [UPSTREAM]: r = sqrt(r2 + 1e-12); k = s2 (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)
template <int KIND>
__device__ __forceinline__ double kern_val(double r2, double s2) {
  if (KIND == DSDGP_KERN_RBF) return s2 * exp(-0.5 * r2);
  const double s5 = 2.23606797749978969641;
  double r = sqrt(r2 + 1e-12);
  return s2 * (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * exp(-s5 * r);
}
template <int KIND>
__device__ __forceinline__ void kern_val_grad(double r2, double s2, double& k, double& dk) {
  if (KIND == DSDGP_KERN_RBF) {
    k = s2 * exp(-0.5 * r2);
    dk = -0.5 * k;
    return;
  }
  const double s5 = 2.23606797749978969641;
  double r = sqrt(r2 + 1e-12);
  double e = s2 * exp(-s5 * r);
  k = (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * e;
  dk = -(5.0 / 6.0) * (1.0 + s5 * r) * e;
}
__device__ __forceinline__ double kern_val_rt(int kind, double r2, double s2) {
  return kind == DSDGP_KERN_RBF ? kern_val<DSDGP_KERN_RBF>(r2, s2) : kern_val<DSDGP_KERN_MATERN52>(r2, s2);
}
#endif
