// Materialised Gram assembly: feature.Kuu / feature.Kuf (layers.py:171,184 -> [UPSTREAM] kern.K).
// HBM-bound: each workgroup builds a 16 x 256 output tile from LDS-staged, lengthscale-scaled row tiles of X and X2
// (pairwise squared distances by direct differences, so r2 >= 0 and K(X,X) is exactly symmetric), and writes it with
// fully coalesced 2 KB row segments.  Algorithmic bytes = 8 * n * n2 (output) + 8 * D * (n + n2) (inputs).
#include "common.hpp"

#define GR_TI 16
#define GR_TJ 256
#define GR_DC 16

template <int KIND>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ X, int64_t n, const double* __restrict__ X2,
                                              int64_t n2, int D, const double* __restrict__ hyp, double diag_add,
                                              int symmetric, double* __restrict__ out, int64_t ld) {
  __shared__ double xi[GR_TI * GR_DC];
  __shared__ double xj[GR_TJ * (GR_DC + 1)];
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * GR_TJ, i0 = (int64_t)blockIdx.y * GR_TI;
  const double s2 = hyp[HYP_VAR];
  const double* ils = hyp + HYP_ILS;
  double r2[GR_TI];
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) r2[ii] = 0.0;
  for (int d0 = 0; d0 < D; d0 += GR_DC) {
    {
      const int ii = tid / GR_DC, dd = tid % GR_DC;
      double v = 0.0;
      if (i0 + ii < n && d0 + dd < D) v = X[(i0 + ii) * D + d0 + dd] * ils[d0 + dd];
      xi[tid] = v;
    }
#pragma unroll
    for (int q = 0; q < GR_DC; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / GR_DC, dd = idx % GR_DC;
      double v = 0.0;
      if (j0 + row < n2 && d0 + dd < D) v = X2[(j0 + row) * D + d0 + dd] * ils[d0 + dd];
      xj[row * (GR_DC + 1) + dd] = v;
    }
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < GR_DC; ++dd) {
      const double v = xj[tid * (GR_DC + 1) + dd];
#pragma unroll
      for (int ii = 0; ii < GR_TI; ++ii) {
        const double df = xi[ii * GR_DC + dd] - v;
        r2[ii] = fma(df, df, r2[ii]);
      }
    }
    __syncthreads();
  }
  const int64_t j = j0 + tid;
  if (j < n2) {
#pragma unroll
    for (int ii = 0; ii < GR_TI; ++ii) {
      const int64_t i = i0 + ii;
      if (i < n) {
        double k = kern_val<KIND>(r2[ii], s2);
        if (symmetric && i == j) k += diag_add;
        out[i * ld + j] = k;
      }
    }
  }
}

int gram_launch(dsdgp_ctx* ctx, int kind, const double* X, int64_t n, const double* X2, int64_t n2, int D,
                const double* hyp_dev, double diag_add, int symmetric, double* out, int64_t ld) {
  ProfScope ps(ctx, "gram");
  dim3 grid(ceil_div(n2, GR_TJ), ceil_div(n, GR_TI));
  if (kind == DSDGP_KERN_RBF)
    hipLaunchKernelGGL(k_gram<DSDGP_KERN_RBF>, grid, dim3(256), 0, ctx->stream, X, n, X2, n2, D, hyp_dev, diag_add,
                       symmetric, out, ld);
  else
    hipLaunchKernelGGL(k_gram<DSDGP_KERN_MATERN52>, grid, dim3(256), 0, ctx->stream, X, n, X2, n2, D, hyp_dev,
                       diag_add, symmetric, out, ld);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

extern "C" int dsdgp_gram(dsdgp_ctx* ctx, const dsdgp_kernel* kern, const double* X, int64_t n, const double* X2,
                          int64_t n2, double jitter, double* out, int64_t ld_out) {
  DS_CHECK_ARG(ctx && kern && X && out && n > 0);
  DS_CHECK_ARG(kern->kind == DSDGP_KERN_RBF || kern->kind == DSDGP_KERN_MATERN52);
  DS_CHECK_ARG(kern->input_dim > 0 && kern->lengthscales);
  const int D = kern->input_dim;
  const int symmetric = (X2 == nullptr);
  if (symmetric) {
    X2 = X;
    n2 = n;
  }
  DS_CHECK_ARG(n2 > 0 && ld_out >= n2);
  std::vector<double> hyp(HYP_ILS + 2 * D, 0.0);
  hyp[HYP_VAR] = kern->variance;
  hyp[HYP_WVAR] = kern->has_white ? kern->white_variance : 0.0;
  hyp[HYP_KDIAG] = hyp[HYP_VAR] + hyp[HYP_WVAR];
  for (int j = 0; j < D; ++j) hyp[HYP_ILS + j] = 1.0 / kern->lengthscales[kern->ard ? j : 0];
  void* scr;
  DS_TRY(ctx_scratch(ctx, hyp.size() * sizeof(double), &scr));
  DS_TRY(ctx_upload(ctx, scr, hyp.data(), hyp.size() * sizeof(double)));     // asynchronous (pinned staging ring)
  const double diag_add = symmetric ? (hyp[HYP_WVAR] + jitter) : 0.0;
  return gram_launch(ctx, kern->kind, X, n, X2, n2, D, (const double*)scr, diag_add, symmetric, out, ld_out);
}
