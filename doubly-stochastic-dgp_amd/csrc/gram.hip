// Materialised Gram assembly: feature.Kuu / feature.Kuf (layers.py:171,184 -> [UPSTREAM] kern.K).
// HBM-bound: each workgroup builds an 8 x 512 output tile (pairwise squared distances by direct differences, so r2 >= 0 and
// K(X,X) is exactly symmetric) and writes it as 4 KB row segments of 16-byte non-temporal stores.  Algorithmic bytes = 8 * n * n2 (output) + 8 * D * (n + n2) (inputs).
#include "common.hpp"

#define GR_TI 8      // rows (X / inducing side) per workgroup
#define GR_TJ 512    // columns (X2 / data side) per workgroup: two per thread -> 16-byte stores, 4 KB runs per output row
#define GR_DC 8      // input dimensions per pass

#define GR_LD (GR_TJ + 8)   // LDS row stride of the transposed X2 tile (doubles)

// Each thread owns two adjacent columns and all GR_TI rows of the tile.  Per pass of GR_DC input dimensions the workgroup's
// 512 X2 rows are read COALESCED (8 threads = one 64-byte row chunk: 4-8 cache lines per wave-instruction; one thread reading
// its own rows with 8-byte loads touches 64 lines per instruction and the kernel turns texture-address-bound, measured 3x
// slower), scaled by 1/lengthscale and parked TRANSPOSED in LDS ([dim][row]: a thread's two columns are one 16-byte read);
// the GR_TI scaled X rows are LDS broadcasts; distances accumulate in registers and the 16 kernel values leave as 8 double2
// non-temporal stores (1 KB per wave-instruction, 4 KB runs per output row).
template <int KIND>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ X, int64_t n, const double* __restrict__ X2,
                                              int64_t n2, int D, const double* __restrict__ hyp, double diag_add,
                                              int symmetric, double* __restrict__ out, int64_t ld) {
  __shared__ __attribute__((aligned(16))) double xi[GR_TI * GR_DC];
  __shared__ __attribute__((aligned(16))) double xj[GR_DC * GR_LD];
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int64_t jbase = (int64_t)blockIdx.x * GR_TJ, j0 = jbase + 2 * tid, i0 = (int64_t)blockIdx.y * GR_TI;
  const double s2 = hyp[HYP_VAR];
  const double* ils = hyp + HYP_ILS;
  double ra[GR_TI], rb[GR_TI];
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) ra[ii] = rb[ii] = 0.0;
  const int sdd = tid % GR_DC, srow = tid / GR_DC;      // staging role: dimension sdd of rows srow + 32 q
  for (int d0 = 0; d0 < D; d0 += GR_DC) {
    const int dc = (D - d0 < GR_DC) ? D - d0 : GR_DC;
    if (d0 > 0) __syncthreads();
    const int dq = d0 + (sdd < dc ? sdd : dc - 1);                       // clamped: every load is unconditional
    const double sc = sdd < dc ? ils[dq] : 0.0;
    double st[GR_TJ / 32];
#pragma unroll
    for (int q = 0; q < GR_TJ / 32; ++q) {
      int64_t row = jbase + srow + 32 * q;
      if (row > n2 - 1) row = n2 - 1;
      st[q] = X2[row * D + dq];
    }
    if (tid < GR_TI * GR_DC) {
      const int ii = tid / GR_DC;
      const int64_t row = (i0 + ii < n) ? i0 + ii : n - 1;
      xi[tid] = X[row * D + dq] * sc;
    }
    // products are rounded by the LDS store on BOTH sides (no fused z - x * il): (i, j) and (j, i) see the same bits and
    // K(X, X) is exactly symmetric
#pragma unroll
    for (int q = 0; q < GR_TJ / 32; ++q) xj[sdd * GR_LD + srow + 32 * q] = st[q] * sc;
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < GR_DC; ++dd) {
      const d2 xx = *reinterpret_cast<const d2*>(&xj[dd * GR_LD + 2 * tid]);
#pragma unroll
      for (int ii = 0; ii < GR_TI; ++ii) {
        const double z = xi[ii * GR_DC + dd];
        const double da = z - xx[0], db = z - xx[1];
        ra[ii] = fma(da, da, ra[ii]);
        rb[ii] = fma(db, db, rb[ii]);
      }
    }
  }
  if (j0 >= n2) return;
  const bool pair = (j0 + 1 < n2) && ((ld & 1) == 0);       // 16-byte aligned pair store
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) {          // values first (straight-line: the exps of all rows interleave) ...
    const int64_t i = i0 + ii;
    ra[ii] = kern_val<KIND>(ra[ii], s2) + ((symmetric && i == j0) ? diag_add : 0.0);
    rb[ii] = kern_val<KIND>(rb[ii], s2) + ((symmetric && i == j0 + 1) ? diag_add : 0.0);
  }
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) {          // ... then the stores
    const int64_t i = i0 + ii;
    if (i < n) {
      double* o = out + i * ld + j0;
      if (pair) {
        __builtin_nontemporal_store((d2){ra[ii], rb[ii]}, reinterpret_cast<d2*>(o));
      } else {
        o[0] = ra[ii];
        if (j0 + 1 < n2) o[1] = rb[ii];
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
