// [UPSTREAM] gpflow.likelihoods.MultiClass(K) with RobustMax(eps = 1e-3), 20-point Gauss–Hermite (SURVEY §2.1 K9',
// Appendix B), reached through BroadcastingLikelihood's flatten/tile (utils.py:76-93):
//   X_h = mu_y + x_h sqrt(clip(2 v_y, 1e-10));  cdf_kh = (1 + erf((X_h - mu_k)/sqrt(2 clip(v_k,1e-10))))/2 * (1-2e-4) + 1e-4 (k != y)
//   p = sum_h w_h/sqrt(pi) prod_{k != y} cdf_kh ;  var_exp = p log(1-eps) + (1-p) log(eps/(K-1))
// Four lanes per (sample,row), five quadrature nodes each; the 20 x K erf evaluations are ALU-bound; per-class gradient accumulators
// sit in LDS.
#include "common.hpp"

#define MC_H 20
#define MC_KMAX 32
#define MC_T 64

__constant__ double c_gh_x[MC_H];
__constant__ double c_gh_w[MC_H];   // w_h / sqrt(pi)
static bool g_gh_ready = false;

static int ensure_gh(hipStream_t st) {
  if (g_gh_ready) return DSDGP_OK;
  // Gauss–Hermite nodes/weights, n = 20 (numpy.polynomial.hermite.hermgauss(20)); symmetric, listed once
  static const double xpos[10] = {0.2453407083009012499, 0.7374737285453943587, 1.2340762153953230079, 1.7385377121165862068,
                                  2.2549740020892756723, 2.7888060584281304806, 3.3478545673832163269, 3.9447640401156252104,
                                  4.6036824495507442731, 5.3874808900112328620};
  static const double wpos[10] = {4.6224366960061008965e-1, 2.8667550536283412972e-1, 1.0901720602002331250e-1,
                                  2.4810520887463643070e-2, 3.2437733422378566463e-3, 2.2833863601635308670e-4,
                                  7.8025564785320636941e-6, 1.0860693707692815356e-7, 4.3993409922731805536e-10,
                                  2.2293936455341516100e-13};
  double x[MC_H], w[MC_H];
  const double isp = 0.56418958354775628695;   // 1/sqrt(pi)
  for (int i = 0; i < 10; ++i) {
    x[10 + i] = xpos[i];  w[10 + i] = wpos[i] * isp;
    x[9 - i] = -xpos[i];  w[9 - i] = wpos[i] * isp;
  }
  DS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_gh_x), x, sizeof(x), 0, hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_gh_w), w, sizeof(w), 0, hipMemcpyHostToDevice, st));
  DS_HIP(hipStreamSynchronize(st));
  g_gh_ready = true;
  return DSDGP_OK;
}

// mode 0: out[row] = var_exp ; mode 1: out[row] = log predictive density ; optional adjoints dmean/dvar = w * d(-ve)/d(mu, v)
// A workgroup (one wave) takes MC_RW rows; lane (row rr, group hg) evaluates MC_H / MC_HG of the row's quadrature nodes, the four
// partial sums of a row meet in LDS and are added in group order.  (One thread per row walked 20 x (K - 1) erf twice — 80 waves for the
// 5120 rows of config 4's shard, 153 us; the cdf values of a node are now kept for its gradient pass.)
#define MC_RW 16
#define MC_HG 4
static_assert(MC_RW * MC_HG == MC_T && MC_H % MC_HG == 0, "lane map");
__global__ __launch_bounds__(MC_T) void k_multiclass(const double* __restrict__ mean, const double* __restrict__ var,
                                                     const double* __restrict__ Y, int64_t n, int64_t R, int K, double eps,
                                                     int mode, double wgt, double* __restrict__ out,
                                                     double* __restrict__ dmean, double* __restrict__ dvar,
                                                     int y_override) {
  __shared__ double gmu[MC_KMAX * MC_T];
  __shared__ double gv[MC_KMAX * MC_T];
  __shared__ double cdfs[MC_KMAX * MC_T];
  __shared__ double ps[MC_T], gsys[MC_T];
  const int tid = threadIdx.x, rr = tid & (MC_RW - 1), hg = tid / MC_RW;
  const int64_t row = (int64_t)blockIdx.x * MC_RW + rr;
  const bool live = row < R;
  const int64_t rc = live ? row : R - 1;                      // (every lane reaches the barrier)
  const int y = y_override >= 0 ? y_override : (int)Y[rc % n];
  const double* mu = mean + rc * K;
  const double* vv = var + rc * K;
  const double vy = fmax(vv[y], 0.5e-10);                     // clip(2 v_y, 1e-10)
  const double sy = sqrt(2.0 * vy);
  for (int k = 0; k < K; ++k) gmu[k * MC_T + tid] = gv[k * MC_T + tid] = 0.0;
  double p = 0.0, gsy = 0.0;                                  // gsy: dp / d(sqrt(2 v_y))
  const double isp = 0.56418958354775628695;
  for (int h = hg * (MC_H / MC_HG); h < (hg + 1) * (MC_H / MC_HG); ++h) {
    const double X = mu[y] + c_gh_x[h] * sy;
    double P = 1.0;
    for (int k = 0; k < K; ++k) {
      if (k == y) continue;
      const double vk = fmax(vv[k], 1e-10);
      const double u = (X - mu[k]) * rsqrt(2.0 * vk);
      const double cdf = 0.5 * (1.0 + erf(u)) * (1.0 - 2e-4) + 1e-4;
      cdfs[k * MC_T + tid] = cdf;
      P *= cdf;
    }
    p = fma(c_gh_w[h], P, p);
    if (dmean) {
      for (int k = 0; k < K; ++k) {
        if (k == y) continue;
        const double vk = fmax(vv[k], 1e-10);
        const double rs = rsqrt(2.0 * vk);
        const double u = (X - mu[k]) * rs;
        const double cdf = cdfs[k * MC_T + tid];
        const double t = c_gh_w[h] * (P / cdf) * (1.0 - 2e-4) * isp * exp(-u * u);   // w_h dP/du_k
        gmu[k * MC_T + tid] -= t * rs;                                                // du/dmu_k = -rs
        if (vv[k] > 1e-10) gv[k * MC_T + tid] -= t * u / (2.0 * vk);                  // du/dv_k = -u / (2 v_k)
        gmu[y * MC_T + tid] += t * rs;                                                // dX/dmu_y = 1
        gsy += t * rs * c_gh_x[h];                                                    // dX/dsy = x_h
      }
    }
  }
  ps[tid] = p;
  gsys[tid] = gsy;
  __syncthreads();
  if (!live) return;
  auto row_sum = [&](const double* v) { return ((v[rr] + v[rr + MC_RW]) + v[rr + 2 * MC_RW]) + v[rr + 3 * MC_RW]; };
  const double l1 = log(1.0 - eps), l0 = log(eps / (K - 1.0));
  if (hg == 0) {
    const double pt = row_sum(ps);
    if (mode == 0)
      out[row] = pt * l1 + (1.0 - pt) * l0;
    else
      out[row] = log(pt * (1.0 - eps) + (1.0 - pt) * (eps / (K - 1.0)));
  }
  if (dmean) {
    const double gsy_t = row_sum(gsys);
    const double s = -wgt * (l1 - l0);                       // d loss / d p  (loss = -wgt * var_exp)
    for (int k = hg; k < K; k += MC_HG) {                    // the row's classes over its four lanes
      const double gm = row_sum(gmu + k * MC_T);
      double gw = row_sum(gv + k * MC_T);
      if (k == y && vv[y] > 0.5e-10) gw += gsy_t / sy;       // d sy / d v_y = 1 / sy
      dmean[row * K + k] = s * gm;
      dvar[row * K + k] = s * gw;
    }
  }
}

int multiclass_launch(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int64_t R, int K,
                      int mode, double wgt, double* out, double* dmean, double* dvar, int y_override) {
  if (K < 2 || K > MC_KMAX) {
    dsdgp_set_error("MultiClass: K=%d outside [2, %d]", K, MC_KMAX);
    return DSDGP_ERR_UNSUPPORTED;
  }
  DS_TRY(ensure_gh(ctx->stream));
  DS_LAUNCH(k_multiclass, dim3(ceil_div(R, MC_RW)), dim3(MC_T), 0, ctx->stream, mean, var, Y, n, R, K, 1e-3, mode,
                     wgt, out, dmean, dvar, y_override);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// out[i] (n x 1) = mean_s var_exp (mode 0) or logsumexp_s density - log S (mode 1), from per-(s,i) values tmp (S*n)
__global__ void k_over_samples(const double* __restrict__ tmp, int64_t n, int S, int mode, const double* __restrict__ sw,
                               double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (mode == 0) {
      double a = 0.0;
      for (int s = 0; s < S; ++s) a += (sw ? sw[s] : 1.0) * tmp[(int64_t)s * n + i];
      out[i] = sw ? a : a / S;
    } else {
      double mx = -1.0 / 0.0;
      for (int s = 0; s < S; ++s) mx = fmax(mx, tmp[(int64_t)s * n + i]);
      double a = 0.0;
      for (int s = 0; s < S; ++s) a += exp(tmp[(int64_t)s * n + i] - mx);
      out[i] = mx + log(a) - log((double)S);
    }
  }
}

extern "C" int dsdgp_multiclass_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n,
                                        int32_t S, int32_t K, int mode, const double* sample_w, double* out) {
  DS_CHECK_ARG(ctx && mean && var && Y && out && n > 0 && S > 0 && (mode == 0 || mode == 1));
  void* scr;
  DS_TRY(ctx_scratch(ctx, (size_t)S * n * sizeof(double), &scr));
  DS_TRY(multiclass_launch(ctx, mean, var, Y, n, (int64_t)S * n, K, mode, 0.0, (double*)scr, nullptr, nullptr, -1));
  DS_LAUNCH(k_over_samples, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, (const double*)scr, n, S, mode,
                     mode == 0 ? sample_w : nullptr, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// predict_mean_and_var: ps[row, k] = predictive probability of class k ; out_var = ps - ps^2   (Appendix B)
__global__ void k_exp_inplace(double* __restrict__ x, double* __restrict__ v, int64_t R, int K, int k) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
    const double p = exp(x[i * K + k]);
    x[i * K + k] = p;
    v[i * K + k] = p - p * p;
  }
}
__global__ void k_scatter_col(const double* __restrict__ src, double* __restrict__ dst, int64_t R, int K, int k) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x)
    dst[i * K + k] = src[i];
}
extern "C" int dsdgp_multiclass_predict(dsdgp_ctx* ctx, const double* mean, const double* var, int64_t R, int32_t K,
                                        double* out_mean, double* out_var) {
  DS_CHECK_ARG(ctx && mean && var && out_mean && out_var && R > 0);
  void* scr;
  DS_TRY(ctx_scratch(ctx, (size_t)R * sizeof(double), &scr));
  const int nb = (int)std::min<int64_t>(2048, ceil_div(R, 256));
  for (int k = 0; k < K; ++k) {
    DS_TRY(multiclass_launch(ctx, mean, var, nullptr, R, R, K, 1, 0.0, (double*)scr, nullptr, nullptr, k));
    DS_LAUNCH(k_scatter_col, dim3(nb), dim3(256), 0, ctx->stream, (const double*)scr, out_mean, R, K, k);
    DS_LAUNCH(k_exp_inplace, dim3(nb), dim3(256), 0, ctx->stream, out_mean, out_var, R, K, k);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
