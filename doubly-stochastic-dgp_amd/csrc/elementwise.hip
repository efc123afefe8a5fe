// Thin elementwise / gather entry points around the hot path (minibatch gather, Gaussian likelihood wrappers).
#include "common.hpp"

__global__ void k_gather_rows(const double* __restrict__ src, int64_t cols, const int64_t* __restrict__ idx, int64_t n,
                              double* __restrict__ dst) {
  const int64_t total = n * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    dst[i] = src[idx[r] * cols + c];
  }
}

extern "C" int dsdgp_gather_rows(dsdgp_ctx* ctx, const double* src, int64_t cols, const int64_t* idx, int64_t n,
                                 int64_t idx_offset, double* dst) {
  DS_CHECK_ARG(ctx && src && idx && dst && n > 0 && cols > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(n * cols, 256));
  DS_LAUNCH(k_gather_rows, dim3(nb), dim3(256), 0, ctx->stream, src, cols, idx + idx_offset, n, dst);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// X and Y rows of one minibatch in one launch (Minibatch(X), Minibatch(Y) share their seed, dgp.py:51-52)
__global__ void k_gather_rows2(const double* __restrict__ a, int64_t ca, double* __restrict__ da, const double* __restrict__ b,
                               int64_t cb, double* __restrict__ db, const int64_t* __restrict__ idx, int64_t n) {
  const int64_t ct = ca + cb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * ct; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ct, j = i % ct;
    if (j < ca)
      da[r * ca + j] = a[idx[r] * ca + j];
    else
      db[r * cb + (j - ca)] = b[idx[r] * cb + (j - ca)];
  }
}
extern "C" int dsdgp_gather_rows2(dsdgp_ctx* ctx, const double* srcX, int64_t colsX, double* dstX, const double* srcY,
                                  int64_t colsY, double* dstY, const int64_t* idx, int64_t n, int64_t idx_offset) {
  DS_CHECK_ARG(ctx && srcX && srcY && dstX && dstY && idx && n > 0 && colsX > 0 && colsY > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(n * (colsX + colsY), 256));
  DS_LAUNCH(k_gather_rows2, dim3(nb), dim3(256), 0, ctx->stream, srcX, colsX, dstX, srcY, colsY, dstY, idx + idx_offset, n);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// mode 0: mean_s variational expectation ; mode 1: logsumexp_s predictive log density - log S
__global__ void k_gauss_over_samples(const double* __restrict__ mean, const double* __restrict__ var,
                                     const double* __restrict__ Y, int64_t n, int S, int DY, double s2, int mode,
                                     const double* __restrict__ sw, double* __restrict__ out) {
  const int64_t total = n * DY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double y = Y[i];
    if (mode == 0) {
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double mu = mean[(int64_t)s * total + i], v = var[(int64_t)s * total + i];
        const double ve = -0.91893853320467274178 - 0.5 * log(s2) - 0.5 * ((y - mu) * (y - mu) + v) / s2;
        acc += sw ? sw[s] * ve : ve;
      }
      out[i] = sw ? acc : acc / S;      // quadrature weights (DGP_Quad.E_log_p_Y, dgp.py:160-166) or the MC mean (dgp.py:90)
    } else {
      double mx = -1.0 / 0.0;
      for (int s = 0; s < S; ++s) {
        const double mu = mean[(int64_t)s * total + i], v = var[(int64_t)s * total + i] + s2;
        const double l = -0.91893853320467274178 - 0.5 * log(v) - 0.5 * (y - mu) * (y - mu) / v;
        mx = l > mx ? l : mx;
      }
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double mu = mean[(int64_t)s * total + i], v = var[(int64_t)s * total + i] + s2;
        const double l = -0.91893853320467274178 - 0.5 * log(v) - 0.5 * (y - mu) * (y - mu) / v;
        acc += exp(l - mx);
      }
      out[i] = mx + log(acc) - log((double)S);
    }
  }
}

static int gauss_over_samples(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int S,
                              int DY, double s2, int mode, const double* sw, double* out) {
  DS_CHECK_ARG(ctx && mean && var && Y && out && n > 0 && S > 0 && DY > 0 && s2 > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(n * DY, 256));
  DS_LAUNCH(k_gauss_over_samples, dim3(nb), dim3(256), 0, ctx->stream, mean, var, Y, n, S, DY, s2, mode, sw, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

extern "C" int dsdgp_gauss_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n,
                                   int32_t S, int32_t DY, double lik_var, const double* sample_w, double* out) {
  return gauss_over_samples(ctx, mean, var, Y, n, S, DY, lik_var, 0, sample_w, out);
}
extern "C" int dsdgp_gauss_predict_density(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y,
                                           int64_t n, int32_t S, int32_t DY, double lik_var, double* out) {
  return gauss_over_samples(ctx, mean, var, Y, n, S, DY, lik_var, 1, nullptr, out);
}

// ---- Bernoulli (probit) through BroadcastingLikelihood: mode 0 mean_s variational expectation, mode 1 logmeanexp_s density
__global__ void k_bern_over_samples(const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ Y,
                                    int64_t n, int S, int DY, int mode, const double* __restrict__ sw, double* __restrict__ out) {
  const int64_t total = n * DY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double y = Y[i];
    if (mode == 0) {
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        double dm, dv;
        const double ve = bern_var_exp(mean[(int64_t)s * total + i], var[(int64_t)s * total + i], y, &dm, &dv);
        acc += sw ? sw[s] * ve : ve;
      }
      out[i] = sw ? acc : acc / S;
    } else {
      double mx = -1.0 / 0.0;
      for (int s = 0; s < S; ++s) {
        const double l = bern_logp(bern_probit(mean[(int64_t)s * total + i] / sqrt(1.0 + var[(int64_t)s * total + i])), y);
        mx = l > mx ? l : mx;
      }
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double l = bern_logp(bern_probit(mean[(int64_t)s * total + i] / sqrt(1.0 + var[(int64_t)s * total + i])), y);
        acc += exp(l - mx);
      }
      out[i] = mx + log(acc) - log((double)S);
    }
  }
}
extern "C" int dsdgp_bernoulli_var_exp(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n,
                                       int32_t S, int32_t DY, int mode, const double* sample_w, double* out) {
  DS_CHECK_ARG(ctx && mean && var && Y && out && n > 0 && S > 0 && DY > 0 && (mode == 0 || mode == 1));
  DS_CHECK_ARG(mode == 0 || !sample_w);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(n * DY, 256));
  DS_LAUNCH(k_bern_over_samples, dim3(nb), dim3(256), 0, ctx->stream, mean, var, Y, n, S, DY, mode, sample_w, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
__global__ void k_bern_predict(const double* __restrict__ mean, const double* __restrict__ var, int64_t count,
                               double* __restrict__ om, double* __restrict__ ov) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double p = bern_probit(mean[i] / sqrt(1.0 + var[i]));
    om[i] = p;
    ov[i] = p - p * p;
  }
}
extern "C" int dsdgp_bernoulli_predict(dsdgp_ctx* ctx, const double* mean, const double* var, int64_t count, double* out_mean,
                                       double* out_var) {
  DS_CHECK_ARG(ctx && mean && var && out_mean && out_var && count > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(count, 256));
  DS_LAUNCH(k_bern_predict, dim3(nb), dim3(256), 0, ctx->stream, mean, var, count, out_mean, out_var);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// ---- Poisson / Exponential / Gamma (exp link), StudentT and Beta through BroadcastingLikelihood: the same two reductions over the samples
__global__ void k_lik_over_samples(int kind, double p0, double p1, const double* __restrict__ mean, const double* __restrict__ var,
                                   const double* __restrict__ Y, int64_t n, int S, int DY, int mode, const double* __restrict__ sw,
                                   double* __restrict__ out) {
  const int64_t total = n * DY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double y = Y[i];
    if (mode == 0) {
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        double dm, dv, dp;
        const double ve = lik_var_exp(kind, mean[(int64_t)s * total + i], var[(int64_t)s * total + i], y, p0, p1, &dm, &dv, &dp);
        acc += sw ? sw[s] * ve : ve;
      }
      out[i] = sw ? acc : acc / S;
    } else {
      double mx = -1.0 / 0.0;
      for (int s = 0; s < S; ++s) {
        const double l = lik_log_density(kind, mean[(int64_t)s * total + i], var[(int64_t)s * total + i], y, p0, p1);
        mx = l > mx ? l : mx;
      }
      double acc = 0.0;
      for (int s = 0; s < S; ++s) {
        const double l = lik_log_density(kind, mean[(int64_t)s * total + i], var[(int64_t)s * total + i], y, p0, p1);
        acc += exp(l - mx);
      }
      out[i] = mx + log(acc) - log((double)S);
    }
  }
}
static bool lik_quad_kind_ok(int kind, double p0, double p1) {
  if (kind == DSDGP_LIK_POISSON) return p1 > 0.0;
  if (kind == DSDGP_LIK_EXPONENTIAL) return true;
  if (kind == DSDGP_LIK_GAMMA || kind == DSDGP_LIK_BETA) return p0 > 0.0;
  return kind == DSDGP_LIK_STUDENT_T && p0 > 0.0 && p1 > 0.0;
}
extern "C" int dsdgp_lik_var_exp(dsdgp_ctx* ctx, int32_t kind, double p0, double p1, const double* mean, const double* var,
                                 const double* Y, int64_t n, int32_t S, int32_t DY, int mode, const double* sample_w, double* out) {
  DS_CHECK_ARG(ctx && mean && var && Y && out && n > 0 && S > 0 && DY > 0 && (mode == 0 || mode == 1));
  DS_CHECK_ARG(mode == 0 || !sample_w);
  DS_CHECK_ARG(lik_quad_kind_ok(kind, p0, p1));
  const int nb = (int)std::min<int64_t>(4096, ceil_div(n * DY, 256));
  DS_LAUNCH(k_lik_over_samples, dim3(nb), dim3(256), 0, ctx->stream, (int)kind, p0, p1, mean, var, Y, n, S, DY, mode, sample_w, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
__global__ void k_lik_predict(int kind, double p0, double p1, const double* __restrict__ mean, const double* __restrict__ var,
                              int64_t count, double* __restrict__ om, double* __restrict__ ov) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    double e, q;
    lik_predict(kind, mean[i], var[i], p0, p1, &e, &q);
    om[i] = e;
    ov[i] = q;
  }
}
extern "C" int dsdgp_lik_predict(dsdgp_ctx* ctx, int32_t kind, double p0, double p1, const double* mean, const double* var,
                                 int64_t count, double* out_mean, double* out_var) {
  DS_CHECK_ARG(ctx && mean && var && out_mean && out_var && count > 0);
  DS_CHECK_ARG(lik_quad_kind_ok(kind, p0, p1));
  const int nb = (int)std::min<int64_t>(4096, ceil_div(count, 256));
  DS_LAUNCH(k_lik_predict, dim3(nb), dim3(256), 0, ctx->stream, (int)kind, p0, p1, mean, var, count, out_mean, out_var);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

__global__ void k_add_scalar(const double* __restrict__ in, double v, int64_t count, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] + v;
}
extern "C" int dsdgp_add_scalar(dsdgp_ctx* ctx, const double* in, double value, int64_t count, double* out) {
  DS_CHECK_ARG(ctx && in && out && count > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(count, 256));
  DS_LAUNCH(k_add_scalar, dim3(nb), dim3(256), 0, ctx->stream, in, value, count, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
