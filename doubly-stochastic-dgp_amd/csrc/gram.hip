// Materialised Gram assembly: feature.Kuu / feature.Kuf (layers.py:171,184 -> [UPSTREAM] kern.K).  Two kernels: k_gram_mfma (D <= 32,
// distances through the MFMA pipe, below) and k_gram (any D, direct differences):
// HBM-bound: each workgroup builds an 8 x 512 output tile (pairwise squared distances by direct differences, so r2 >= 0 and
// K(X,X) is exactly symmetric) and writes it as 4 KB row segments of 16-byte non-temporal stores.  Algorithmic bytes = 8 * n * n2 (output) + 8 * D * (n + n2) (inputs).
#include <algorithm>

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
  const bool pair = (j0 + 1 < n2) && ((ld & 1) == 0) && (((uintptr_t)out & 15) == 0);       // 16-byte aligned pair store
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

// ---------------------------------------------------------------------------------------------------------------------
// D <= 32: the distances through the fp64 MFMA pipe.  The form above spends 2 D + ~28 fp64 VALU operations per 8 output bytes — at
// D = 8 that is the VALU limit and the HBM limit at the same point (~7.7 TB/s), at D = 30 the VALU caps it at 3.7 TB/s (measured
// 3.5 and 1.5).  Here r2 = |x_i|^2 + |x_j|^2 - 2 G_ij with the Gram block G on the MFMA pipe (ceil(D / 4) instructions per
// 16 x 16 outputs), clamped at 0 and exactly 0 on the diagonal of K(X, X) — the form of the chains' Kuf tile and of the head
// launch's Ku.  What is left per output is the kernel function: the store stream is the bound.
//  * a wave owns 16 rows (i) and walks 32-column tiles: TWO accumulators whose B operands are the X2 rows j0 + 2c and j0 + 2c + 1, so
//    that lane (g, c) ends with the ADJACENT outputs (i, j0 + 2c), (i, j0 + 2c + 1) of rows i = g + 4t: one 16-byte non-temporal
//    store per row, 256 contiguous bytes per 16 lanes;
//  * K(X, X) stays exactly symmetric: the scaled coordinates are rounded once (x * (1 / l)), both norms of a pair come from the same
//    sequential sum (scaled_norm), G_ij and G_ji add the same products in the same order.
// grid (column chunks, ceil(n / 64)), 256 threads; jtiles 32-column tiles per workgroup.
__device__ __forceinline__ double scaled_norm(const double* __restrict__ row, int D, const double* __restrict__ ils) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double v = row[d] * ils[d];
    s = fma(v, v, s);
  }
  return s;
}
template <int KIND, int KS>
__global__ __launch_bounds__(256) void k_gram_mfma(const double* __restrict__ X, int64_t n, const double* __restrict__ X2, int64_t n2,
                                                   int D, const double* __restrict__ hyp, double diag_add, int symmetric,
                                                   double* __restrict__ out, int64_t ld, int jtiles) {
  // LDS: the workgroup's X2 rows scaled by 1 / lengthscale, [column][4 KS + 1] (odd stride: the B-operand reads of a half-wave — rows
  // 2c, k index g — fall on 32 distinct 8-byte banks), zero-padded beyond D; then their squared norms.  The rows are ONE contiguous
  // block of X2, read with coalesced loads (a lane fetching its own rows with 8-byte loads touches one cache line per lane and
  // instruction: texture-address-bound — the first version of this kernel did, and ran slower than k_gram).
  extern __shared__ __attribute__((aligned(16))) double gm_dyn[];
  constexpr int LDP = 4 * KS + 1;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int cols = jtiles * 32;
  double* xs = gm_dyn;
  double* nrm = gm_dyn + (size_t)cols * LDP;
  const int64_t jbase = (int64_t)blockIdx.x * cols;
  const double s2 = hyp[HYP_VAR];
  const double* __restrict__ ils = hyp + HYP_ILS;
  {
    const int64_t avail = (n2 - jbase < cols ? n2 - jbase : cols) * (int64_t)D;       // doubles of X2 that exist for this chunk
    const double* __restrict__ src = X2 + jbase * D;
    for (int e = tid; e < cols * D; e += 256) {
      const int col = e / D, d = e - col * D;
      xs[col * LDP + d] = (e < avail) ? src[e] * ils[d] : 0.0;
    }
    if (D < 4 * KS)
      for (int e = tid; e < cols * (4 * KS - D); e += 256) {
        const int col = e / (4 * KS - D), d = D + e - col * (4 * KS - D);
        xs[col * LDP + d] = 0.0;
      }
  }
  __syncthreads();
  for (int col = tid; col < cols; col += 256) {
    double sn = 0.0;
    for (int d = 0; d < D; ++d) sn = fma(xs[col * LDP + d], xs[col * LDP + d], sn);
    nrm[col] = sn;
  }
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.y * 64 + 16 * wave;
  if (i0 >= n) return;
  // A operand: row i0 + c scaled the same way, k = 4 s + g; its norm by the same sequential sum as the columns'
  double a[KS], ni[4];
  {
    const int64_t ir = (i0 + c < n) ? i0 + c : n - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = (4 * s + g < D) ? X[ir * D + 4 * s + g] * ils[4 * s + g] : 0.0;
    const double nr = scaled_norm(X + ir * D, D, ils);       // every lane: the norm of row i0 + c (four copies, no divergence)
#pragma unroll
    for (int t = 0; t < 4; ++t) ni[t] = __shfl(nr, g + 4 * t);
  }
  const bool even_ld = (ld & 1) == 0 && (((uintptr_t)out & 15) == 0);     // 16-byte stores need an aligned base as well (any double* is accepted)
  for (int jt = 0; jt < jtiles; ++jt) {
    const int64_t j0 = jbase + 32 * jt;
    if (j0 >= n2) break;
    const int64_t jA = j0 + 2 * c, jB = jA + 1;
    const double* __restrict__ pa = xs + (32 * jt + 2 * c) * LDP + g;
    d4 GA = (d4){0, 0, 0, 0}, GB = (d4){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      GA = mfma_f64(a[s], pa[4 * s], GA);
      GB = mfma_f64(a[s], pa[LDP + 4 * s], GB);
    }
    const double nA = nrm[32 * jt + 2 * c], nB = nrm[32 * jt + 2 * c + 1];
    double ka[4], kb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t i = i0 + g + 4 * t;
      double ra = fmax(ni[t] + nA - 2.0 * GA[t], 0.0), rb = fmax(ni[t] + nB - 2.0 * GB[t], 0.0);
      const bool da = symmetric && i == jA, db = symmetric && i == jB;
      if (da) ra = 0.0;
      if (db) rb = 0.0;
      ka[t] = kern_val<KIND>(ra, s2) + (da ? diag_add : 0.0);
      kb[t] = kern_val<KIND>(rb, s2) + (db ? diag_add : 0.0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t i = i0 + g + 4 * t;
      if (i < n && jA < n2) {
        double* o = out + i * ld + jA;
        if (even_ld && jB < n2) {
          __builtin_nontemporal_store((d2){ka[t], kb[t]}, reinterpret_cast<d2*>(o));
        } else {
          o[0] = ka[t];
          if (jB < n2) o[1] = kb[t];
        }
      }
    }
  }
}

template <int KIND>
static void gram_mfma_go(hipStream_t st, int ks, dim3 grid, const double* X, int64_t n, const double* X2, int64_t n2, int D,
                         const double* hyp, double diag_add, int symmetric, double* out, int64_t ld, int jtiles) {
  const int KSp = ks <= 2 ? 2 : (ks <= 4 ? 4 : 8);
  const size_t lds = (size_t)jtiles * 32 * (4 * KSp + 2) * sizeof(double);
  if (KSp == 2)
    hipLaunchKernelGGL((k_gram_mfma<KIND, 2>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, jtiles);
  else if (KSp == 4)
    hipLaunchKernelGGL((k_gram_mfma<KIND, 4>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, jtiles);
  else
    hipLaunchKernelGGL((k_gram_mfma<KIND, 8>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, jtiles);
}

int gram_launch(dsdgp_ctx* ctx, int kind, const double* X, int64_t n, const double* X2, int64_t n2, int D,
                const double* hyp_dev, double diag_add, int symmetric, double* out, int64_t ld) {
  ProfScope ps(ctx, "gram");
  if (D <= 32) {
    // about 2048 workgroups where the problem has them (eight light workgroups per CU), at most 16 column tiles (512 columns) each
    const int64_t ct = ceil_div(n2, 32), rb = ceil_div(n, 64);
    const int ksp = D <= 8 ? 2 : (D <= 16 ? 4 : 8);
    const int jmax = std::max(1, (int)((40 << 10) / (32 * (4 * ksp + 2) * sizeof(double))));      // <= 40 KB of LDS per workgroup
    int jtiles = (int)std::min<int64_t>(std::min(16, jmax), std::max<int64_t>(1, ct * rb / 2048));
    dim3 grid((unsigned)ceil_div(ct, jtiles), (unsigned)rb);
    const int ks = ceil_div(D, 4);
    if (kind == DSDGP_KERN_RBF)
      gram_mfma_go<DSDGP_KERN_RBF>(ctx->stream, ks, grid, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld, jtiles);
    else
      gram_mfma_go<DSDGP_KERN_MATERN52>(ctx->stream, ks, grid, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld, jtiles);
    DS_HIP(hipGetLastError());
    return DSDGP_OK;
  }
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
