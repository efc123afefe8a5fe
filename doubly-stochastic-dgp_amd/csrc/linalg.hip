// fp64 MFMA dense kernels for the M x M algebra of the DS-DGP path (gfx950).
//   k_gemm_grouped  : grouped/batched GEMM, 64x64 tiles, LDS-staged, v_mfma_f64_16x16x4_f64
//   k_potrf_trtri   : blocked right-looking Cholesky (tf.cholesky, layers.py:172) with MFMA trailing update, followed by
//                     the blocked triangular inverse that turns the two tf.matrix_triangular_solve calls of
//                     layers.py:186,188 into MFMA products inside the layer chain kernel.
#include <stdlib.h>
#include <algorithm>

#include "linalg.hpp"
#include "chol_lds.hpp"

#define GT 64
#define GK 16
#define GLD 81  // odd LDS row stride (doubles): conflict-free k-major writes, <=2-way on the fragment reads

// Interior tiles of k_gemm_grouped (the whole 64 x 64 tile and every 16-step of its k range inside the matrices, even leading
// dimensions, 16-byte aligned bases): the (batch, k-tile) pipeline with the operand orientation as template parameters and ONE 32-byte
// load per operand, thread and step.  The generic loop below decides the orientation per element at run time and guards every element:
// eight 8-byte loads per thread and step, each behind its own branch, and ~250 VALU instructions of index arithmetic per sixteen MFMAs
// (profiles/r05_gemm_grouped_isa.md) — 33 TFLOP/s on the 1024 x 1024 x 1024 products of config 5.
template <bool TA, bool TB>
__device__ __forceinline__ void gg_interior(const GemmProblem& P, int m0, int n0, int b0, int ks_lo, int ksteps, int nsteps, double* As,
                                            double* Bs, int tid, int g, int c, int wr, int wc, d4 (&acc)[2][2]) {
  typedef const d4 __attribute__((address_space(1)))* gd4;
  // A tile 64 (m) x 16 (k): row-major A (!TA): thread -> row tid >> 2, k offset 4 (tid & 3); A^T stored (TA): row k = tid >> 4, m offset 4 (tid & 15)
  const int ar = TA ? (tid >> 4) : (tid >> 2), ao = TA ? 4 * (tid & 15) : 4 * (tid & 3);
  const int br = TB ? (tid >> 2) : (tid >> 4), bo = TB ? 4 * (tid & 3) : 4 * (tid & 15);
  d4 ra, rb;
  auto gload = [&](int step) {
    const int b = b0 + step / ksteps, k0 = (ks_lo + step % ksteps) * GK;
    gcptr A = (gcptr)(P.A + (int64_t)b * P.sA);
    gcptr B = (gcptr)(P.B + (int64_t)b * P.sB);
    ra = TA ? *reinterpret_cast<gd4>(A + (int64_t)(k0 + ar) * P.lda + m0 + ao) : *reinterpret_cast<gd4>(A + (int64_t)(m0 + ar) * P.lda + k0 + ao);
    rb = TB ? *reinterpret_cast<gd4>(B + (int64_t)(n0 + br) * P.ldb + k0 + bo) : *reinterpret_cast<gd4>(B + (int64_t)(k0 + br) * P.ldb + n0 + bo);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (TA) As[ar * GLD + ao + j] = ra[j]; else As[(ao + j) * GLD + ar] = ra[j];
      if (TB) Bs[(bo + j) * GLD + br] = rb[j]; else Bs[br * GLD + bo + j] = rb[j];
    }
  };
  gload(0);
  for (int step = 0; step < nsteps; ++step) {
    lstore();
    __syncthreads();
    if (step + 1 < nsteps) gload(step + 1);
#pragma unroll
    for (int k4 = 0; k4 < GK; k4 += 4) {
      const double a0 = As[(k4 + g) * GLD + wr * 32 + c];
      const double a1 = As[(k4 + g) * GLD + wr * 32 + 16 + c];
      const double b0v = Bs[(k4 + g) * GLD + wc * 32 + c];
      const double b1v = Bs[(k4 + g) * GLD + wc * 32 + 16 + c];
      acc[0][0] = mfma_f64(a0, b0v, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1v, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0v, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1v, acc[1][1]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_gemm_grouped(const GemmProblem* __restrict__ probs, int nprob) {
  __shared__ double As[GK * GLD];
  __shared__ double Bs[GK * GLD];
  const int bid = blockIdx.x;
  int p = 0, t;
  if (probs[0].order) {          // longest-processing-time order of the launch (gemm_plan_lpt)
    p = probs[0].order[2 * bid];
    t = probs[0].order[2 * bid + 1];
  } else {
    DS_FIND_SEGMENT(p, probs, nprob, tile_start, bid);
    t = bid - probs[p].tile_start;
  }
  const GemmProblem P = probs[p];
  const int tiles = P.tiles_m * P.tiles_n;
  int b0, b1;
  if (P.batch_reduce) {
    b0 = 0;
    b1 = P.batch;
  } else {
    b0 = t / tiles;
    b1 = b0 + 1;
    t = t % tiles;
  }
  const int m0 = (t / P.tiles_n) * GT, n0 = (t % P.tiles_n) * GT;
  if (P.lower_only && n0 > m0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0, 0, 0, 0};

  // software pipeline over (batch, k-tile) steps: the global loads of step t+1 are issued before the MFMAs of step t
  int kmin = 0, kmax = P.k;
  if (P.tri & 1) kmin = max(kmin, n0);
  if (P.tri & 2) kmax = min(kmax, m0 + GT);
  if (P.tri & 4) kmax = min(kmax, n0 + GT);
  if (P.tri & 8) kmin = max(kmin, m0);
  const int ks_lo = kmin / GK;
  const int ksteps = max(0, (kmax + GK - 1) / GK - ks_lo);
  const int nsteps = (b1 - b0) * ksteps;
  double ra[4], rb[4];
  auto gload = [&](int step) {
    const int b = b0 + step / ksteps, k0 = (ks_lo + step % ksteps) * GK;
    gcptr A = (gcptr)(P.A + (int64_t)b * P.sA);
    gcptr B = (gcptr)(P.B + (int64_t)b * P.sB);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int mm, kk;
      if (!P.transA) { kk = tid & 15; mm = (tid >> 4) + 16 * i; } else { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
      double v = 0.0;
      if (m0 + mm < P.m && k0 + kk < P.k)
        v = P.transA ? A[(int64_t)(k0 + kk) * P.lda + m0 + mm] : A[(int64_t)(m0 + mm) * P.lda + k0 + kk];
      ra[i] = v;
      int nn, k2;
      if (!P.transB) { nn = tid & 63; k2 = (tid >> 6) + 4 * i; } else { k2 = tid & 15; nn = (tid >> 4) + 16 * i; }
      double w = 0.0;
      if (n0 + nn < P.n && k0 + k2 < P.k)
        w = P.transB ? B[(int64_t)(n0 + nn) * P.ldb + k0 + k2] : B[(int64_t)(k0 + k2) * P.ldb + n0 + nn];
      rb[i] = w;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int mm, kk;
      if (!P.transA) { kk = tid & 15; mm = (tid >> 4) + 16 * i; } else { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
      As[kk * GLD + mm] = ra[i];
      int nn, k2;
      if (!P.transB) { nn = tid & 63; k2 = (tid >> 6) + 4 * i; } else { k2 = tid & 15; nn = (tid >> 4) + 16 * i; }
      Bs[k2 * GLD + nn] = rb[i];
    }
  };
  const bool interior = nsteps > 0 && m0 + GT <= P.m && n0 + GT <= P.n && (ks_lo + ksteps) * GK <= P.k && ((P.lda | P.ldb) & 1) == 0 &&
                        ((P.sA | P.sB | P.lda | P.ldb) & 3) == 0 && (((uintptr_t)P.A | (uintptr_t)P.B) & 31) == 0;
  if (interior) {
    if (P.transA) {
      if (P.transB) gg_interior<true, true>(P, m0, n0, b0, ks_lo, ksteps, nsteps, As, Bs, tid, g, c, wr, wc, acc);
      else gg_interior<true, false>(P, m0, n0, b0, ks_lo, ksteps, nsteps, As, Bs, tid, g, c, wr, wc, acc);
    } else {
      if (P.transB) gg_interior<false, true>(P, m0, n0, b0, ks_lo, ksteps, nsteps, As, Bs, tid, g, c, wr, wc, acc);
      else gg_interior<false, false>(P, m0, n0, b0, ks_lo, ksteps, nsteps, As, Bs, tid, g, c, wr, wc, acc);
    }
  } else {
  if (nsteps > 0) gload(0);
  for (int step = 0; step < nsteps; ++step) {
    lstore();
    __syncthreads();
    if (step + 1 < nsteps) gload(step + 1);
#pragma unroll
    for (int k4 = 0; k4 < GK; k4 += 4) {
      const double a0 = As[(k4 + g) * GLD + wr * 32 + c];
      const double a1 = As[(k4 + g) * GLD + wr * 32 + 16 + c];
      const double b0v = Bs[(k4 + g) * GLD + wc * 32 + c];
      const double b1v = Bs[(k4 + g) * GLD + wc * 32 + 16 + c];
      acc[0][0] = mfma_f64(a0, b0v, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1v, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0v, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1v, acc[1][1]);
    }
    __syncthreads();
  }
  }
  gptr C = (gptr)(P.C + (P.batch_reduce ? 0 : (int64_t)b0 * P.sC));
  // beta C: all sixteen values requested together (clamped addresses), not one round trip per element behind its own branch
  double cold[2][2][4];
  if (P.beta != 0.0) {
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wr * 32 + ib * 16 + g + 4 * r, P.m - 1);
          const int col = min(n0 + wc * 32 + jb * 16 + c, P.n - 1);
          cold[ib][jb][r] = C[(int64_t)row * P.ldc + col];
        }
  }
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + ib * 16 + g + 4 * r;
        const int col = n0 + wc * 32 + jb * 16 + c;
        if (row < P.m && col < P.n) {
          double v = P.alpha * acc[ib][jb][r];
          if (P.beta != 0.0) v += P.beta * cold[ib][jb][r];
          C[(int64_t)row * P.ldc + col] = v;
          if ((P.tri & 16) && n0 < m0) C[(int64_t)col * P.ldc + row] = v;     // mirror of a symmetric result
        }
      }
}

// ------------------------------------------------------------------------------------------------------
// Large-tile variant for the M >= 512 algebra (Ku^-1, Lu^-1 q_sqrt, U_d, U_d U_d^T, P_d T_d, natural-gradient products: up to
// 2 M^3 flops each, D_out of them per layer): 128 x 128 output tile per workgroup, each of the 4 waves a 64 x 64 quadrant
// (16 accumulators), K in steps of 16 through a DOUBLE-BUFFERED LDS pair (the global loads of step t+1 are issued before the
// 64 MFMAs of step t, one barrier per step), 8 LDS operand reads per 16 MFMAs.  Same problem descriptor, structure hints and
// epilogue as k_gemm_grouped (which keeps the small / thin problems: it wastes less on partial tiles).
// ------------------------------------------------------------------------------------------------------
#define BT 128
#define BK 16
#define BLD 144   // LDS row stride (doubles): 128 + 16 pad -> the g and g+1 k-rows of a fragment read sit 32 banks apart (conflict-free)

__global__ __launch_bounds__(256) void k_gemm_big(const GemmProblem* __restrict__ probs, int nprob) {
  __shared__ __attribute__((aligned(16))) double As[2][BK * BLD];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * BLD];
  const int bid = blockIdx.x;
  int p = 0, t;
  if (probs[0].order) {          // longest-processing-time order of the launch (gemm_plan_lpt)
    p = probs[0].order[2 * bid];
    t = probs[0].order[2 * bid + 1];
  } else {
    DS_FIND_SEGMENT(p, probs, nprob, tile_start, bid);
    t = bid - probs[p].tile_start;
  }
  const GemmProblem P = probs[p];
  const int tiles = P.tiles_m * P.tiles_n;
  int b0, b1;
  if (P.batch_reduce) {
    b0 = 0;
    b1 = P.batch;
  } else {
    b0 = t / tiles;
    b1 = b0 + 1;
    t = t % tiles;
  }
  const int m0 = (t / P.tiles_n) * BT, n0 = (t % P.tiles_n) * BT;
  if (P.lower_only && n0 > m0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  int kmin = 0, kmax = P.k;
  if (P.tri & 1) kmin = max(kmin, n0);
  if (P.tri & 2) kmax = min(kmax, m0 + BT);
  if (P.tri & 4) kmax = min(kmax, n0 + BT);
  if (P.tri & 8) kmin = max(kmin, m0);
  const int ks_lo = kmin / BK;
  const int ksteps = max(0, (kmax + BK - 1) / BK - ks_lo);
  const int nsteps = (b1 - b0) * ksteps;
  // staging roles.  "row-contiguous" operand (A not transposed / B transposed: element [x][k], k contiguous): thread takes
  // row x = tid / 2, k = 8 (tid & 1) .. + 7.  "k-major" operand (A transposed / B not transposed: [k][x]): k = tid / 16,
  // x = 8 (tid & 15) .. + 7.  Either way 64 contiguous bytes per thread.
  double ra[8], rb[8];
  auto gload = [&](int step) {
    const int b = b0 + step / ksteps, k0 = (ks_lo + step % ksteps) * BK;
    gcptr A = (gcptr)(P.A + (int64_t)b * P.sA);
    gcptr B = (gcptr)(P.B + (int64_t)b * P.sB);
    // interior chunks load UNCONDITIONALLY (eight loads in flight); a per-element guard puts every load in its own basic block
    // and serialises eight memory round trips per operand (measured: 10x slower than the 64 x 64 kernel)
    auto load8 = [&](gcptr src, bool row_ok, int first, int limit, double (&r)[8]) {
      if (row_ok && first + 8 <= limit) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = src[u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = (row_ok && first + u < limit) ? src[u] : 0.0;
      }
    };
    if (!P.transA) {
      const int mm = m0 + (tid >> 1), kk = k0 + 8 * (tid & 1);
      load8(A + (int64_t)mm * P.lda + kk, mm < P.m, kk, P.k, ra);
    } else {
      const int kk = k0 + (tid >> 4), mm = m0 + 8 * (tid & 15);
      load8(A + (int64_t)kk * P.lda + mm, kk < P.k, mm, P.m, ra);
    }
    if (P.transB) {
      const int nn = n0 + (tid >> 1), kk = k0 + 8 * (tid & 1);
      load8(B + (int64_t)nn * P.ldb + kk, nn < P.n, kk, P.k, rb);
    } else {
      const int kk = k0 + (tid >> 4), nn = n0 + 8 * (tid & 15);
      load8(B + (int64_t)kk * P.ldb + nn, kk < P.k, nn, P.n, rb);
    }
  };
  auto lstore = [&](int buf) {
    if (!P.transA) {
      const int mm = tid >> 1, kk = 8 * (tid & 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) As[buf][(kk + u) * BLD + mm] = ra[u];
    } else {
      const int kk = tid >> 4, mm = 8 * (tid & 15);
#pragma unroll
      for (int u = 0; u < 8; ++u) As[buf][kk * BLD + mm + u] = ra[u];
    }
    if (P.transB) {
      const int nn = tid >> 1, kk = 8 * (tid & 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) Bs[buf][(kk + u) * BLD + nn] = rb[u];
    } else {
      const int kk = tid >> 4, nn = 8 * (tid & 15);
#pragma unroll
      for (int u = 0; u < 8; ++u) Bs[buf][kk * BLD + nn + u] = rb[u];
    }
  };
  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) gload(step + 1);
#pragma unroll
    for (int k4 = 0; k4 < BK; k4 += 4) {
      double a[4], bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[buf][(k4 + g) * BLD + wr * 64 + 16 * i + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[j] = Bs[buf][(k4 + g) * BLD + wc * 64 + 16 * j + c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(a[i], bq[j], acc[i][j]);
    }
    if (step + 1 < nsteps) lstore(buf ^ 1);     // the other buffer: its last readers finished before the previous barrier
    __syncthreads();
  }
  gptr C = (gptr)(P.C + (P.batch_reduce ? 0 : (int64_t)b0 * P.sC));
#pragma unroll
  for (int ib = 0; ib < 4; ++ib)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + ib * 16 + g + 4 * r;
        const int col = n0 + wc * 64 + jb * 16 + c;
        if (row < P.m && col < P.n) {
          double v = P.alpha * acc[ib][jb][r];
          if (P.beta != 0.0) v += P.beta * C[(int64_t)row * P.ldc + col];
          C[(int64_t)row * P.ldc + col] = v;
          if ((P.tri & 16) && n0 < m0) C[(int64_t)col * P.ldc + row] = v;     // mirror of a symmetric result
        }
      }
}

// ------------------------------------------------------------------------------------------------------
// Short-K variant for the M <= 256 algebra of the training step (Ku^-1 = Lu^-T Lu^-1, Lu^-1 q_sqrt_d, S_d, U_d, U_d U_d^T, P_d T_d, GS_d
// at M = 128: a few dozen 64 x 64 tiles of 4 MFLOP problems).  The LDS-staged kernel above pays one global round trip + two
// barriers per 16-wide k-tile with a single tile of prefetch: eight dependent round trips = 14 us for a 128^3 problem, four times
// per step.  Here there is NO LDS and NO barrier: each wave owns a 32 x 32 quarter of the tile and feeds the MFMAs straight from
// global memory, as the weight-gradient products do — the operand that is contiguous along k comes in as ONE 32-byte load per lane
// and 16-wide k-block (lane (g, c) takes k = 4 g .. 4 g + 3 and uses them in MFMA steps s = 0..3; the other operand's rows 4 g + s are
// four 8-byte loads, 128 contiguous bytes across the 16 c-lanes), all loads of four k-blocks in flight at once.  Same descriptor,
// structure hints and epilogue as k_gemm_grouped.
// ------------------------------------------------------------------------------------------------------
// k loop of one wave's 32 x 32 quarter; TA / TB as template parameters so that the loop body is ONE basic block (a run-time
// transpose test inside it put every load behind its own branch: the round trips serialised and the kernel measured 20 us for
// a 128^3 problem, slower than the LDS-staged one)
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_small_loop(gcptr A, gcptr B, int64_t lda, int64_t ldb, const int (&ra)[2], const int (&cb)[2],
                                                int kb_lo, int kb_hi, int g, d4 (&acc)[2][2]) {
  typedef const d4 __attribute__((address_space(1)))* gd4;
#pragma unroll 4
  for (int kb = kb_lo; kb < kb_hi; ++kb) {
    const int k0 = 16 * kb + 4 * g;
    d4 av[2], bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (!TA) {
        av[i] = *reinterpret_cast<gd4>(A + (int64_t)ra[i] * lda + k0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) av[i][s] = A[(int64_t)(k0 + s) * lda + ra[i]];
      }
      if constexpr (TB) {
        bv[i] = *reinterpret_cast<gd4>(B + (int64_t)cb[i] * ldb + k0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[i][s] = B[(int64_t)(k0 + s) * ldb + cb[i]];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64(av[i][s], bv[j][s], acc[i][j]);
  }
}

__global__ __launch_bounds__(256) void k_gemm_small(const GemmProblem* __restrict__ probs, int nprob) {
  const int bid = blockIdx.x;
  int p = 0, t;
  if (probs[0].order) {          // longest-processing-time order of the launch (gemm_plan_lpt)
    p = probs[0].order[2 * bid];
    t = probs[0].order[2 * bid + 1];
  } else {
    DS_FIND_SEGMENT(p, probs, nprob, tile_start, bid);
    t = bid - probs[p].tile_start;
  }
  const GemmProblem P = probs[p];
  const int tiles = P.tiles_m * P.tiles_n;
  int b0, b1;
  if (P.batch_reduce) {
    b0 = 0;
    b1 = P.batch;
  } else {
    b0 = t / tiles;
    b1 = b0 + 1;
    t = t % tiles;
  }
  const int m0 = (t / P.tiles_n) * GT, n0 = (t % P.tiles_n) * GT;
  if (P.lower_only && n0 > m0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  int kmin = 0, kmax = P.k;
  if (P.tri & 1) kmin = max(kmin, n0 + 32 * wc);
  if (P.tri & 2) kmax = min(kmax, m0 + 32 * wr + 32);
  if (P.tri & 4) kmax = min(kmax, n0 + 32 * wc + 32);
  if (P.tri & 8) kmin = max(kmin, m0 + 32 * wr);
  const int kb_lo = kmin / 16, kb_hi = (min(kmax, P.k) + 15) / 16;
  // rows / columns of this wave's 2 x 2 blocks, clamped for the loads (partial tiles: thin right-hand sides, M = 16 * odd): the
  // clamped lanes compute a copy of the last row / column, which the guarded epilogue drops
  int ra[2], cb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra[i] = min(m0 + 32 * wr + 16 * i + c, P.m - 1);
    cb[i] = min(n0 + 32 * wc + 16 * i + c, P.n - 1);
  }
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  for (int b = b0; b < b1; ++b) {
    gcptr A = (gcptr)(P.A + (int64_t)b * P.sA);
    gcptr B = (gcptr)(P.B + (int64_t)b * P.sB);
    if (!P.transA && !P.transB) gemm_small_loop<false, false>(A, B, P.lda, P.ldb, ra, cb, kb_lo, kb_hi, g, acc);
    else if (!P.transA) gemm_small_loop<false, true>(A, B, P.lda, P.ldb, ra, cb, kb_lo, kb_hi, g, acc);
    else if (!P.transB) gemm_small_loop<true, false>(A, B, P.lda, P.ldb, ra, cb, kb_lo, kb_hi, g, acc);
    else gemm_small_loop<true, true>(A, B, P.lda, P.ldb, ra, cb, kb_lo, kb_hi, g, acc);
  }
  gptr Cp = (gptr)(P.C + (P.batch_reduce ? 0 : (int64_t)b0 * P.sC));
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + ib * 16 + g + 4 * r;
        const int col = n0 + wc * 32 + jb * 16 + c;
        if (row < P.m && col < P.n) {
          double v = P.alpha * acc[ib][jb][r];
          if (P.beta != 0.0) v += P.beta * Cp[(int64_t)row * P.ldc + col];
          Cp[(int64_t)row * P.ldc + col] = v;
          if ((P.tri & 16) && n0 < m0) Cp[(int64_t)col * P.ldc + row] = v;     // mirror of a symmetric result
        }
      }
}

// Launches whose largest problem is at least GEMM_BIG_MIN in both output dimensions take the 128 x 128 kernel; the planned
// tile count carries the choice in GEMM_BIG_FLAG / GEMM_SMALL_FLAG so that every call site keeps passing plan -> launch unchanged.
// Round 3 re-measured the threshold (512 since round 2): the 64 x 64 kernels win up to 1024 — a single 1024^3 product is 64 tiles of 128 x 128
// on 256 CUs (200 us) against 256 tiles of 64 x 64 (93 us); batched D_out x 1024^3 products with triangular / symmetric hints lose less
// to the hints' tile granularity (config 5: gemm 5.10 -> 4.30 ms per step, config 4: 1.74 -> 1.63); from 2048 the large tile leads
// (2048^3: 43 TFLOP/s, 4096 x 4096 x 256: 49).
#define GEMM_BIG_FLAG (1 << 30)
#define GEMM_BIG_MIN 2048
// Launches whose every problem is short in k (<= GEMM_SMALL_K, whole 16-blocks, 32-byte aligned rows) take the LDS-free kernel.
#define GEMM_SMALL_FLAG (1 << 29)
#define GEMM_SMALL_K 256
static bool gemm_small_ok(const GemmProblem& P) {
  if (P.k > GEMM_SMALL_K || P.k % 16 != 0 || P.m < 1 || P.n < 1) return false;
  auto al = [](const double* p, int64_t ld, int64_t st) { return ((uintptr_t)p & 31) == 0 && ld % 4 == 0 && st % 4 == 0; };
  if (!P.transA && !al(P.A, P.lda, P.sA)) return false;      // the operand contiguous along k is read with 32-byte loads
  if (P.transB && !al(P.B, P.ldb, P.sB)) return false;
  if (P.C == P.A || P.C == P.B) return false;                // in-place updates: a wave would overwrite what its neighbour still reads
  return true;
}

int gemm_plan(GemmProblem* host, int nprob, int allow_big) {
  int big = allow_big == 2, small = nprob > 0;          // 2: the 128 x 128 kernel whatever the sizes (in-place 128-column panels)
  static const int big_min = getenv("DSDGP_GEMM_BIG_MIN") ? atoi(getenv("DSDGP_GEMM_BIG_MIN")) : GEMM_BIG_MIN;   // (A/B aid)
  if (allow_big == 1)
    for (int i = 0; i < nprob; ++i)
      if (host[i].m >= big_min && host[i].n >= big_min && host[i].k >= 64) big = 1;
  for (int i = 0; i < nprob; ++i) small = small && gemm_small_ok(host[i]);
  if (big) small = 0;
  const int T = big ? BT : GT;
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    GemmProblem& P = host[i];
    P.tiles_m = ceil_div(P.m, T);
    P.tiles_n = ceil_div(P.n, T);
    P.tile_start = total;
    total += P.tiles_m * P.tiles_n * (P.batch_reduce ? 1 : P.batch);
  }
  return big ? (total | GEMM_BIG_FLAG) : (small ? (total | GEMM_SMALL_FLAG) : total);
}

int gemm_plan_lpt(GemmProblem* host, int nprob, std::vector<int32_t>& order, int allow_big) {
  const int planned = gemm_plan(host, nprob, allow_big);
  const bool big = (planned & GEMM_BIG_FLAG) != 0;
  const int T = big ? BT : GT;
  struct Tile { int work, p, t; };
  std::vector<Tile> tiles;
  for (int p = 0; p < nprob; ++p) {
    const GemmProblem& P = host[p];
    const int per = P.tiles_m * P.tiles_n, nb = P.batch_reduce ? 1 : P.batch;
    for (int t = 0; t < per * nb; ++t) {
      const int tt = t % per;
      const int m0 = (tt / P.tiles_n) * T, n0 = (tt % P.tiles_n) * T;
      if (P.lower_only && n0 > m0) continue;
      // the k range the kernels give this tile (the LDS-free kernel trims per 32 x 32 quarter: its widest quarter has this range)
      int kmin = 0, kmax = P.k;
      if (P.tri & 1) kmin = std::max(kmin, n0);
      if (P.tri & 2) kmax = std::min(kmax, m0 + T);
      if (P.tri & 4) kmax = std::min(kmax, n0 + T);
      if (P.tri & 8) kmin = std::max(kmin, m0);
      const int steps = std::max(0, (kmax + 15) / 16 - kmin / 16) * (P.batch_reduce ? P.batch : 1);
      tiles.push_back(Tile{steps, p, t});
    }
  }
  std::stable_sort(tiles.begin(), tiles.end(), [](const Tile& a, const Tile& b) { return a.work > b.work; });
  order.clear();
  order.reserve(2 * tiles.size());
  for (const Tile& t : tiles) {
    order.push_back(t.p);
    order.push_back(t.t);
  }
  return (int)tiles.size() | (planned & (GEMM_BIG_FLAG | GEMM_SMALL_FLAG));
}

static void gemm_dispatch(const GemmProblem* dev, int nprob, int planned, hipStream_t st) {
  const bool big = (planned & GEMM_BIG_FLAG) != 0, small = (planned & GEMM_SMALL_FLAG) != 0;
  const int total_tiles = planned & ~(GEMM_BIG_FLAG | GEMM_SMALL_FLAG);
  if (total_tiles <= 0) return;
  if (big)
    DS_LAUNCH(k_gemm_big, dim3(total_tiles), dim3(256), 0, st, dev, nprob);
  else if (small)
    DS_LAUNCH(k_gemm_small, dim3(total_tiles), dim3(256), 0, st, dev, nprob);
  else
    DS_LAUNCH(k_gemm_grouped, dim3(total_tiles), dim3(256), 0, st, dev, nprob);
}
int gemm_launch(dsdgp_ctx* ctx, const GemmProblem* dev, int nprob, int total_tiles, hipStream_t stream) {
  if ((total_tiles & ~(GEMM_BIG_FLAG | GEMM_SMALL_FLAG)) <= 0) return DSDGP_OK;
  hipStream_t st = stream ? stream : ctx->stream;
  ProfScope ps(ctx, "gemm", st);
  gemm_dispatch(dev, nprob, total_tiles, st);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// Blocked Cholesky + triangular inverse, one 256-thread workgroup per matrix, NB = 16.
// ------------------------------------------------------------------------------------------------------
// INLDS: the working matrix lives in LDS (n <= 128: n x (n+4) doubles <= 135 KB of the 160 KB), so the ~35 dependent
// phases of the factorisation pay LDS latency instead of L2 round trips; the result is written back at the end.
// INLDS: the working matrix lives in LDS (n <= 128: n x (n+4) doubles <= 135 KB of the 160 KB), so the dependent phases
// of the factorisation pay LDS latency instead of L2 round trips; the factor is written back at the end.
// Triangular inverse: block-column forward substitution — wave w owns block columns {w, nb-1-w, ...} of X = L^-1 and keeps
// them in registers (the D layout of one 16x16 product is the B operand of the next), so there is no inter-wave
// dependency and no barrier:  X_jj = Ljj^-1,  X_ij = -Lii^-1 * sum_{k=j}^{i-1} L_ik X_kj.
#define POTRF_MAXNB 16
template <bool INLDS> struct WPtr { typedef gptr type; };
template <> struct WPtr<true> { typedef lptr type; };
template <bool INLDS>
__global__ __launch_bounds__(256) void k_potrf_trtri(const PotrfItem* __restrict__ items, int nb_max) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* Ld = dyn;                     // 16 x 17 : current diagonal block (+ reciprocal diagonal in column 16)
  double* s_red = dyn + 16 * 17;        // 4
  double* Xdall = dyn + 16 * 17 + 8;    // nb x 16 x 17 : inverses of all diagonal blocks
  __shared__ int s_info;
  const PotrfItem it = items[blockIdx.x];
  const int n = it.n, nb = n / 16;
  const int ld = INLDS ? n + 4 : it.ld;
  typename WPtr<INLDS>::type W;            // LDS-resident or global working matrix, never a generic (flat) pointer
  if constexpr (INLDS) W = (lptr)(Xdall + nb_max * 16 * 17); else W = (gptr)it.W;
  gptr Linv = (gptr)it.Linv;
  gptr LinvT = (gptr)it.LinvT;
  const int ldi = it.ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  long long tph[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#define PH(i) do { long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; } while (0)
  if (tid == 0) s_info = 0;
  if (INLDS) {
    // stage the lower triangle (the factorisation never reads above the diagonal blocks): 16-byte loads, 8 in flight
    typedef double d2v __attribute__((ext_vector_type(2)));
    const int halfn = n / 2, tot = n * halfn;
    for (int base = 0; base < tot; base += 256 * 8) {
      d2v v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int i = idx / halfn, j2 = idx % halfn;
        v[u] = (d2v){0.0, 0.0};
        if (idx < tot && 2 * j2 <= (i | 15)) v[u] = *reinterpret_cast<const d2v*>(it.W + (int64_t)i * it.ld + 2 * j2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int i = idx / halfn, j2 = idx % halfn;
        if (idx < tot && 2 * j2 <= (i | 15)) { W[i * ld + 2 * j2] = v[u][0]; W[i * ld + 2 * j2 + 1] = v[u][1]; }
      }
    }
  }
  __syncthreads();
  PH(5);

  // Diagonal block jb: Cholesky by wave 0 (lane i < 16 owns row i in registers; pivots/columns broadcast with v_readlane),
  // then column `lane` of X = L_jj^{-1} by forward substitution, column-oriented so that the 15 updates after each solved
  // entry are independent (critical path 16 steps instead of 136 chained FMAs); LDS reads are broadcasts.  Wave 0 only.
  auto factor_diag = [&](int jb) {
    const int j0 = jb * 16;
    double* Xd = Xdall + jb * 16 * 17;
    {
      const int i = c;
      double a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = W[(int64_t)(j0 + i) * ld + j0 + j];
      double myinv = 0.0;
      int bad = 0;
      Chol16<0>::run(a, i, myinv, bad);
      if (bad && lane == 0 && s_info == 0) s_info = j0 + bad;
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const double v = (j <= i) ? a[j] : 0.0;
          Ld[i * 17 + j] = v;
          W[(int64_t)(j0 + i) * ld + j0 + j] = v;
        }
        Ld[i * 17 + 16] = myinv;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);          // single wave: its own LDS stores are visible to its own later reads once drained
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) {
      double x[16], sacc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        x[k] = sacc[k] * Ld[k * 17 + 16];
#pragma unroll
        for (int i = k + 1; i < 16; ++i) sacc[i] = fma(-Ld[i * 17 + k], x[k], sacc[i]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Xd[i * 17 + lane] = x[i];
    }
  };
  // one trailing tile: A_ik -= L_ij L_kj^T
  auto trail_tile = [&](int ib, int kb, int j0) {
    d4 acc;
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = W[(int64_t)(ib * 16 + g + 4 * r) * ld + kb * 16 + c];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      av[s] = W[(int64_t)(ib * 16 + c) * ld + j0 + 4 * s + g];
      bv[s] = W[(int64_t)(kb * 16 + c) * ld + j0 + 4 * s + g];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_f64(-av[s], bv[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) W[(int64_t)(ib * 16 + g + 4 * r) * ld + kb * 16 + c] = acc[r];
  };

  if (wave == 0) factor_diag(0);
  __syncthreads();
  PH(0);
  for (int jb = 0; jb < nb; ++jb) {
    const int j0 = jb * 16;
    double* Xd = Xdall + jb * 16 * 17;
    // (b) panel: L_ij = A_ij * L_jj^{-T}   (16x16 MFMA products)
    for (int ib = jb + 1 + wave; ib < nb; ib += 4) {
      d4 acc = (d4){0, 0, 0, 0};
      double av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = W[(int64_t)(ib * 16 + c) * ld + j0 + 4 * s + g];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma_f64(av[s], Xd[c * 17 + 4 * s + g], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) W[(int64_t)(ib * 16 + g + 4 * r) * ld + j0 + c] = acc[r];
    }
    __syncthreads();
    PH(2);
    // (c) trailing update (syrk) with look-ahead: wave 0 updates the next diagonal tile first and factors it (Cholesky +
    // inverse, a ~7000-cycle dependent chain) while waves 1-3 update the other tiles — the factorisation of block jb+1 is
    // off the critical path whenever the rest of the trailing update is at least as long
    const int nt = nb - jb - 1;
    if (nt > 0) {
      const int cnt = nt * (nt + 1) / 2;      // tile id 0 = (jb+1, jb+1)
      if (wave == 0) {
        trail_tile(jb + 1, jb + 1, j0);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        factor_diag(jb + 1);
      } else {
        // two tiles per iteration: their LDS loads, 4-MFMA chains and stores interleave (a lone tile is a ~2000-cycle
        // load -> dependent MFMAs -> store chain; the trailing update was 55 % of the factorisation)
        auto tile_of = [&](int id, int& ib2, int& kb2) {
          ib2 = 0;
          while ((ib2 + 1) * (ib2 + 2) / 2 <= id) ++ib2;
          kb2 = id - ib2 * (ib2 + 1) / 2;
        };
        int id = wave;
        for (; id + 3 < cnt; id += 6) {
          int ia, ka, ib_, kb_;
          tile_of(id, ia, ka);
          tile_of(id + 3, ib_, kb_);
          const int ra0 = (jb + 1 + ia) * 16, ca0 = (jb + 1 + ka) * 16, rb0 = (jb + 1 + ib_) * 16, cb0 = (jb + 1 + kb_) * 16;
          d4 accA, accB;
          double a1[4], b1[4], a2[4], b2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            accA[r] = W[(int64_t)(ra0 + g + 4 * r) * ld + ca0 + c];
            accB[r] = W[(int64_t)(rb0 + g + 4 * r) * ld + cb0 + c];
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            a1[s4] = W[(int64_t)(ra0 + c) * ld + j0 + 4 * s4 + g];
            b1[s4] = W[(int64_t)(ca0 + c) * ld + j0 + 4 * s4 + g];
            a2[s4] = W[(int64_t)(rb0 + c) * ld + j0 + 4 * s4 + g];
            b2[s4] = W[(int64_t)(cb0 + c) * ld + j0 + 4 * s4 + g];
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            accA = mfma_f64(-a1[s4], b1[s4], accA);
            accB = mfma_f64(-a2[s4], b2[s4], accB);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            W[(int64_t)(ra0 + g + 4 * r) * ld + ca0 + c] = accA[r];
            W[(int64_t)(rb0 + g + 4 * r) * ld + cb0 + c] = accB[r];
          }
        }
        for (; id < cnt; id += 3) {
          int ib2, kb2;
          tile_of(id, ib2, kb2);
          trail_tile(jb + 1 + ib2, jb + 1 + kb2, j0);
        }
      }
    }
    __syncthreads();
    PH(3);
  }
  // logdet over the real (unpadded) part
  {
    double s = 0.0;
    for (int i = tid; i < it.nreal; i += 256) s += 2.0 * log(W[(int64_t)i * ld + i]);
    s = sum_wave(s);
    if (lane == 0) s_red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      const double ldv = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      if (it.pad & 16) {          // block of a larger matrix: accumulate (launches of one matrix are sequential)
        it.scal[0] += ldv;
        if (s_info && it.scal[1] == 0.0) it.scal[1] = (double)(s_info + it.info_offset);
      } else {
        it.scal[0] = ldv;
        it.scal[1] = (double)s_info;
      }
    }
  }
  // write the factor back (LDS variant) with a zeroed strict upper triangle — unless the caller never reads it
  if (!(INLDS && (it.pad & 8)))
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    const double v = (j > i) ? 0.0 : W[(int64_t)i * ld + j];
    it.W[(int64_t)i * it.ld + j] = v;
  }
  if (!Linv) return;
  PH(4);
  if (nb > POTRF_MAXNB) {
    // large matrices (n > 256): block-row recurrence through global memory, one barrier per block row
    __syncthreads();
    for (int jb = 0; jb < nb; ++jb) {          // diagonal blocks of the inverse
      const double* Xi = Xdall + jb * 16 * 17;
      for (int idx = tid; idx < 256; idx += 256) {
        const int i = idx >> 4, j = idx & 15;
        const double v = Xi[i * 17 + j];
        Linv[(int64_t)(jb * 16 + i) * ldi + jb * 16 + j] = v;
        if (LinvT) LinvT[(int64_t)(jb * 16 + j) * ldi + jb * 16 + i] = v;
      }
    }
    __syncthreads();
    for (int ib = 1; ib < nb; ++ib) {
      const double* Xi = Xdall + ib * 16 * 17;
      for (int jb2 = wave; jb2 < ib; jb2 += 4) {
        d4 S0 = (d4){0, 0, 0, 0};
        for (int kb = jb2; kb < ib; ++kb) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double av = W[(int64_t)(ib * 16 + c) * ld + kb * 16 + 4 * s + g];
            const double bv = Linv[(int64_t)(kb * 16 + 4 * s + g) * ldi + jb2 * 16 + c];
            S0 = mfma_f64(av, bv, S0);
          }
        }
        d4 R = (d4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) R = mfma_f64(-Xi[c * 17 + 4 * s + g], S0[s], R);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          Linv[(int64_t)(ib * 16 + g + 4 * t) * ldi + jb2 * 16 + c] = R[t];
          if (LinvT) LinvT[(int64_t)(jb2 * 16 + c) * ldi + ib * 16 + g + 4 * t] = R[t];
        }
      }
      __syncthreads();
    }
  } else
  // block-column forward substitution, columns paired (w, nb-1-w, w+4, ...) so the four waves carry equal work
  for (int q = 0; q < (nb + 3) / 4; ++q) {
    const int jcol = ((q & 1) == 0) ? (q / 2) * 8 + wave : (q / 2) * 8 + 7 - wave;
    if (jcol >= nb) continue;
    d4 x[POTRF_MAXNB];
#pragma unroll
    for (int rel = 0; rel < POTRF_MAXNB; ++rel) {
      const int ib = jcol + rel;
      if (ib < nb) {
        const double* Xi = Xdall + ib * 16 * 17;
        if (rel == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) x[0][t] = Xi[(g + 4 * t) * 17 + c];
        } else {
          d4 S0 = (d4){0, 0, 0, 0}, S1 = (d4){0, 0, 0, 0};
#pragma unroll
          for (int r2 = 0; r2 < rel; ++r2) {
            const int kb = jcol + r2;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const double a = W[(int64_t)(ib * 16 + c) * ld + kb * 16 + 4 * s + g];
              if (r2 & 1)
                S1 = mfma_f64(a, x[r2][s], S1);
              else
                S0 = mfma_f64(a, x[r2][s], S0);
            }
          }
          S0 += S1;
          d4 R = (d4){0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 4; ++s) R = mfma_f64(-Xi[c * 17 + 4 * s + g], S0[s], R);
          x[rel] = R;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          Linv[(int64_t)(ib * 16 + g + 4 * t) * ldi + jcol * 16 + c] = x[rel][t];
          if (LinvT) LinvT[(int64_t)(jcol * 16 + c) * ldi + ib * 16 + g + 4 * t] = x[rel][t];
        }
      }
    }
  }
  PH(5);
  if (tid == 0 && it.pad == 7)
    for (int q = 0; q < 6; ++q) it.scal[2 + q] = (double)tph[q];
#undef PH
}

int potrf_launch(dsdgp_ctx* ctx, const PotrfItem* dev_items, int nitems, int n_max) {
  ProfScope ps(ctx, "potrf");
  const int nb_max = n_max / 16;
  const size_t base = (16 * 17 + 8 + (size_t)nb_max * 16 * 17) * sizeof(double);
  if (n_max <= 128) {
    const size_t lds = base + (size_t)n_max * (n_max + 4) * sizeof(double);
    if (lds > 64 * 1024)
      {
        static int lds_set = 0;   // the attribute is sticky: one driver call per instance and size
        if ((int)lds > lds_set) {
          DS_HIP(hipFuncSetAttribute((const void*)k_potrf_trtri<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          lds_set = (int)lds;
        }
      }
    DS_LAUNCH(k_potrf_trtri<true>, dim3(nitems), dim3(256), lds, ctx->stream, dev_items, nb_max);
  } else {
    if (base > 64 * 1024)
      {
        static int lds_set = 0;   // the attribute is sticky: one driver call per instance and size
        if ((int)base > lds_set) {
          DS_HIP(hipFuncSetAttribute((const void*)k_potrf_trtri<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)base));
          lds_set = (int)base;
        }
      }
    DS_LAUNCH(k_potrf_trtri<false>, dim3(nitems), dim3(256), base, ctx->stream, dev_items, nb_max);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// pad / unpad helpers for the primitive entry points (arbitrary n -> multiple of 16 with an identity pad)
// ------------------------------------------------------------------------------------------------------
__global__ void k_pad_spd(const double* __restrict__ A, int64_t lda, int64_t strideA, int n, double* __restrict__ P,
                          int np) {
  const double* Ab = A + (int64_t)blockIdx.y * strideA;
  double* Pb = P + (int64_t)blockIdx.y * np * np;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < np * np; idx += gridDim.x * blockDim.x) {
    const int i = idx / np, j = idx % np;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < n && j < n) v = (j <= i) ? Ab[(int64_t)i * lda + j] : Ab[(int64_t)j * lda + i];
    Pb[idx] = v;
  }
}
__global__ void k_unpad(const double* __restrict__ P, int np, double* __restrict__ A, int64_t lda, int64_t strideA,
                        int n) {
  const double* Pb = P + (int64_t)blockIdx.y * np * np;
  double* Ab = A + (int64_t)blockIdx.y * strideA;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
    const int i = idx / n, j = idx % n;
    Ab[(int64_t)i * lda + j] = Pb[(int64_t)i * np + j];
  }
}

extern "C" int dsdgp_potrf(dsdgp_ctx* ctx, int batch, int n, double* A, int64_t lda, int64_t stride, int* info) {
  DS_CHECK_ARG(ctx && A && batch > 0 && n > 0 && lda >= n);
  const int np = n > 176 ? (int)round_up(n, 64) : (int)round_up(n, 16);      // from 192 on: the blocked multi-workgroup sequence
  const size_t mat_bytes = (size_t)batch * np * np * sizeof(double);
  const size_t item_bytes = round_up(batch * sizeof(PotrfItem), 256);
  const size_t scal_bytes = round_up((size_t)batch * 2 * sizeof(double), 256);
  void* scr;
  DS_TRY(ctx_scratch(ctx, mat_bytes + item_bytes + scal_bytes, &scr));
  double* P = (double*)scr;
  PotrfItem* items_d = (PotrfItem*)((char*)scr + mat_bytes);
  double* scal_d = (double*)((char*)scr + mat_bytes + item_bytes);
  std::vector<PotrfItem> items(batch);
  for (int b = 0; b < batch; ++b) {
    items[b] = PotrfItem{P + (size_t)b * np * np, nullptr, nullptr, scal_d + 2 * b, np, np, n, 0, 0, 0};
  }
  DS_TRY(ctx_upload(ctx, items_d, items.data(), batch * sizeof(PotrfItem)));   // asynchronous (pinned staging ring)
  DS_LAUNCH(k_pad_spd, dim3(ceil_div(np * np, 256), batch), dim3(256), 0, ctx->stream, A, lda, stride, n, P, np);
  if (np >= 192 && np % 64 == 0) {
    // large matrices: multi-workgroup blocked path; the plan of the last call is kept in the context (the model path pre-builds its plans)
    const int64_t key[4] = {(int64_t)(uintptr_t)P, np, batch, n};
    if (!ctx->potrf_plan || memcmp(key, ctx->potrf_key, sizeof(key)) != 0) {
      DS_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->potrf_plan) ctx->potrf_plan_free(ctx->potrf_plan);
      ctx->potrf_plan = nullptr;
      BigChol* plan = new BigChol();
      const int rc = bigchol_build(ctx, *plan, P, nullptr, nullptr, scal_d, batch, (int64_t)np * np, 2, np, n, nullptr, false);
      if (rc != DSDGP_OK) {
        bigchol_free(*plan);
        delete plan;
        return rc;
      }
      ctx->potrf_plan = plan;
      ctx->potrf_plan_free = [](void* p) { bigchol_free(*(BigChol*)p); delete (BigChol*)p; };
      memcpy(ctx->potrf_key, key, sizeof(key));
    }
    DS_TRY(bigchol_run(ctx, *(BigChol*)ctx->potrf_plan));
  } else {
    DS_TRY(potrf_launch(ctx, items_d, batch, np));
  }
  DS_LAUNCH(k_unpad, dim3(ceil_div(n * n, 256), batch), dim3(256), 0, ctx->stream, P, np, A, lda, stride, n);
  DS_HIP(hipGetLastError());
  if (info) {
    std::vector<double> sc(2 * batch);
    DS_HIP(hipMemcpyAsync(sc.data(), scal_d, sc.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
    *info = 0;
    for (int b = 0; b < batch; ++b)
      if (sc[2 * b + 1] != 0.0 && *info == 0) *info = (int)sc[2 * b + 1];
    if (*info) {
      dsdgp_set_error("Cholesky decomposition was not successful (pivot %d)", *info);
      return DSDGP_ERR_NOT_SPD;
    }
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_gemm(dsdgp_ctx* ctx, int transA, int transB, int m, int n, int k, double alpha, const double* A,
                          int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  DS_CHECK_ARG(ctx && A && B && C && m > 0 && n > 0 && k > 0);
  GemmProblem P{};
  P.A = A; P.B = B; P.C = C;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc;
  P.m = m; P.n = n; P.k = k;
  P.transA = transA; P.transB = transB;
  P.batch = 1; P.batch_reduce = 0;
  P.alpha = alpha; P.beta = beta;
  const int total = gemm_plan(&P, 1);
  void* scr;
  DS_TRY(ctx_scratch(ctx, sizeof(GemmProblem), &scr));
  DS_TRY(ctx_upload(ctx, scr, &P, sizeof(P)));
  return gemm_launch(ctx, (const GemmProblem*)scr, 1, total);
}

// inverse of padded lower-triangular matrices by the blocked recurrence of k_potrf_trtri (natural-gradient step: T^-1 of q_sqrt)
__global__ __launch_bounds__(256) void k_trtri_only(double* __restrict__ Wb, double* __restrict__ Linvb, int n, int64_t stride) {
  // W: padded lower-triangular L (n multiple of 16, identity pad), one matrix per workgroup. Recurrence of k_potrf_trtri.
  double* __restrict__ W = Wb + (int64_t)blockIdx.x * stride;
  double* __restrict__ Linv = Linvb + (int64_t)blockIdx.x * stride;
  __shared__ double Ld[16 * 17];
  const int ld = n, nb = n / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  for (int jb = 0; jb < nb; ++jb) {
    const int j0 = jb * 16;
    __syncthreads();
    for (int idx = tid; idx < 256; idx += 256) Ld[(idx >> 4) * 17 + (idx & 15)] = W[(int64_t)(j0 + (idx >> 4)) * ld + j0 + (idx & 15)];
    __syncthreads();
    if (wave == 0 && lane < 16) {
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Ld[i * 17 + k] * x[k];
        x[i] = s / Ld[i * 17 + i];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Linv[(int64_t)(j0 + i) * ld + j0 + lane] = x[i];
    }
  }
  __syncthreads();
  for (int ib = 1; ib < nb; ++ib) {
    for (int jb2 = wave; jb2 < ib; jb2 += 4) {
      d4 S = (d4){0, 0, 0, 0};
      for (int kb = jb2; kb < ib; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double a = W[(int64_t)(ib * 16 + c) * ld + kb * 16 + 4 * s + g];
          const double b = Linv[(int64_t)(kb * 16 + 4 * s + g) * ld + jb2 * 16 + c];
          S = mfma_f64(a, b, S);
        }
      }
      d4 R = (d4){0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) R = mfma_f64(-Linv[(int64_t)(ib * 16 + c) * ld + ib * 16 + 4 * s + g], S[s], R);
#pragma unroll
      for (int r = 0; r < 4; ++r) Linv[(int64_t)(ib * 16 + g + 4 * r) * ld + jb2 * 16 + c] = R[r];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    if (j > i) Linv[(int64_t)i * ld + j] = 0.0;
  }
}

int trtri_launch(dsdgp_ctx* ctx, double* W, double* Linv, int n, int64_t stride, int batch) {
  DS_LAUNCH(k_trtri_only, dim3(batch), dim3(256), 0, ctx->stream, W, Linv, n, stride);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// multi-workgroup blocked Cholesky / triangular inverse (see linalg.hpp)
// ------------------------------------------------------------------------------------------------------
// zero the 64x64 blocks strictly above the block diagonal (the blocked factorisation only maintains the lower blocks)
// end of the blocked factorisation: the panels parked transposed above the block diagonal (BCB x BCB blocks) go to their places below
// it and the upper blocks are zeroed.  One 16 x 16 tile per workgroup pass, through LDS: both sides run along rows.  grid (x, batch)
__global__ __launch_bounds__(256) void k_mirror_panels(double* __restrict__ W, int n, int64_t stride, int bs) {
  __shared__ double tt[16][17];
  double* Wb = W + (int64_t)blockIdx.y * stride;
  const int nt = n / 16, ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  for (int t = blockIdx.x; t < nt * nt; t += gridDim.x) {
    const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;          // tile of the UPPER part: rows i0.., columns j0..
    if (j0 / bs <= i0 / bs) continue;                           // (uniform per workgroup)
    __syncthreads();
    tt[ti][tj] = Wb[(int64_t)(i0 + ti) * n + j0 + tj];
    Wb[(int64_t)(i0 + ti) * n + j0 + tj] = 0.0;
    __syncthreads();
    Wb[(int64_t)(j0 + ti) * n + i0 + tj] = tt[tj][ti];
  }
}
__global__ void k_transpose_lower(const double* __restrict__ X, double* __restrict__ XT, int n, int64_t stride) {
  const double* Xb = X + (int64_t)blockIdx.y * stride;
  double* Tb = XT + (int64_t)blockIdx.y * stride;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n * n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    Tb[idx] = (j >= i) ? Xb[(int64_t)j * n + i] : 0.0;      // XT[i][j] = X[j][i], zero below the diagonal
  }
}
// inverse of the 64x64 lower-triangular diagonal blocks of already-triangular matrices (from_tri mode), one WG per block
__global__ __launch_bounds__(256) void k_trtri_diag64(const PotrfItem* __restrict__ items) {
  __shared__ __attribute__((aligned(16))) double Lb[64 * 64];
  __shared__ double Xb[64 * 65];
  __shared__ double rinv[64];
  const PotrfItem it = items[blockIdx.x];
  const int tid = threadIdx.x;
  {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = u * 256 + tid, i = idx >> 6, j = idx & 63;
      v[u] = (j <= i) ? it.W[(int64_t)i * it.ld + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = u * 256 + tid, i = idx >> 6, j = idx & 63;
      Lb[idx] = v[u];
      if (i == j) rinv[i] = 1.0 / v[u];
    }
  }
  __syncthreads();
  if (tid < 64) {
    // column tid of X by forward substitution, the column in REGISTERS: row i of L is read at wave-uniform addresses (LDS broadcast) and the
    // loop is fully unrolled — 2016 FMAs; the entries above the diagonal come out as exact zeros by themselves (all their terms are zero).
    // (the rolled form — every term two dependent LDS round trips — took 97 us)
    const int col = tid;
    double x[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      double s0 = (i == col) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k + 1 < i; k += 2) {
        s0 -= Lb[i * 64 + k] * x[k];
        s1 -= Lb[i * 64 + k + 1] * x[k + 1];
      }
      if (i & 1) s0 -= Lb[i * 64 + i - 1] * x[i - 1];
      x[i] = (s0 + s1) * rinv[i];
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) Xb[i * 65 + col] = x[i];
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    it.Linv[(int64_t)i * it.ld + j] = Xb[i * 65 + j];
  }
}

// factor + inverse of ONE diagonal block (n = 128, or 64 at a ragged end) of the blocked factorisation, LDS-resident (chol_lds.hpp):
// ~25 us for 128 columns where k_potrf_trtri took 23 us for 64.  PotrfItem as for k_potrf_trtri (pad bit 16: accumulate into scal).
//
// Look-ahead form of the blocked factorisation (round 6): the launch that factors diagonal block q also carries, in further workgroups, the
// WIDE part of the trailing update with panel q - 1 (block columns >= q + 1: nothing the factorisation of block q reads), so that only
// k_chol_panel (solve + update of block column q + 1) sits between two diagonal blocks on the critical path.  Workgroups [0, nchol) factor,
// side task mb * nwide + t updates 64 x 64 sub-tile t (lower triangle of the 64-row strips from row (q + 1) 128 on) of matrix mb:
// A_ij -= L_i,q-1 L_j,q-1^T with both operands read from the panel parked TRANSPOSED above the block diagonal (rows (q - 1) 128 ..).
struct CholWide {
  double* W;
  int64_t stride;
  int32_t n, q, nwide, nchol;
};
__device__ __forceinline__ void chol_wide_update(const CholWide wa, int idx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int mb = idx / wa.nwide;
  idx -= mb * wa.nwide;
  int si = 0;
  while ((si + 1) * (si + 2) / 2 <= idx) ++si;
  const int sj = idx - si * (si + 1) / 2;
  const int64_t n = wa.n;
  const int base = (wa.q + 1) * 128, pK = (wa.q - 1) * 128;
  const int ri = base + 64 * si + 16 * (wave >> 1), rj = base + 64 * sj + 32 * (wave & 1);
  gptr Wb = (gptr)(wa.W + (int64_t)mb * wa.stride);
  // (the products are summed on their own and subtracted once: accumulating into the large diagonal entries rounds at their magnitude)
  d4 c0, c1, acc0 = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    c0[t] = Wb[(ri + g + 4 * t) * n + rj + c];
    c1[t] = Wb[(ri + g + 4 * t) * n + rj + 16 + c];
  }
  gcptr ma = (gcptr)(Wb + (pK + g) * n + ri + c), mq = (gcptr)(Wb + (pK + g) * n + rj + c);
#pragma unroll 1
  for (int s0 = 0; s0 < 32; s0 += 8) {
    double av[8], b0[8], b1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = ma[4 * (s0 + u) * n];
      b0[u] = mq[4 * (s0 + u) * n];
      b1[u] = mq[4 * (s0 + u) * n + 16];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = mfma_f64(av[u], b0[u], acc0);
      acc1 = mfma_f64(av[u], b1[u], acc1);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    Wb[(ri + g + 4 * t) * n + rj + c] = c0[t] - acc0[t];
    Wb[(ri + g + 4 * t) * n + rj + 16 + c] = c1[t] - acc1[t];
  }
}

// X = L^-1 inside the look-ahead sequence (round 6): block row i of X — X_ij = -X_ii sum_{k = j}^{i-1} L_ik X_kj, everything on its right
// final once diagonal block i is factored — is formed by further workgroups of the launch that factors block i + 1 (the last row by a
// launch of its own behind the last block), instead of the 2 log2(n / 128) dependent product launches behind the factorisation.  A
// workgroup owns a 16-column slab of X_ij: wave w the 16 rows 16 w .. of T = sum_k L_ik X_kj (L read from the panels parked transposed
// above the block diagonal, 32 (i - j) k-steps), the slab through LDS, then the rows of -X_ii T (k <= its own rows: X_ii lower
// triangular).  j = i (with a transposed output only): the slab of X_ii itself goes to the transposed copy.  8 (i [+ 1]) workgroups per matrix.
struct CholXrow {
  const double* W;
  double* Linv;
  double* LinvT;      // may be NULL
  int64_t stride;
  int32_t n, i, nx, batch;      // nx slabs per matrix
  // 0: the whole row.  The LAST row is split over two launches: 1 = the terms k <= i - 2 of T (inside the launch that factors block i; the
  // sums wait in X_ij's own place), 2 = the term k = i - 1 on top of them and the product with X_ii (behind that launch)
  int32_t mode, pad;
};
__device__ __forceinline__ void chol_xrow(const CholXrow xa, lptr T, int idx) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int mb = idx / xa.nx;
  idx -= mb * xa.nx;
  const int j = idx >> 3, cs = idx & 7, i = xa.i;
  const int64_t n = xa.n;
  const int rows = xa.n - 128 * i < 128 ? xa.n - 128 * i : 128;      // 64 at a ragged end
  gcptr Wb = (gcptr)(xa.W + (int64_t)mb * xa.stride);
  gptr Xb = (gptr)(xa.Linv + (int64_t)mb * xa.stride);
  gptr XT = xa.LinvT ? (gptr)(xa.LinvT + (int64_t)mb * xa.stride) : (gptr) nullptr;
  if (j == i) {      // X_ii^T, columns 16 cs .. of X_ii (rows from 16 cs on: below the block diagonal of the slab nothing is stored)
    for (int e = tid; e < rows * 16; e += CHOL_THREADS) {
      const int r = e >> 4, cc = e & 15;
      if (r >= 16 * cs && 16 * cs < rows) XT[(128 * i + 16 * cs + cc) * n + 128 * i + r] = Xb[(128 * i + r) * n + 128 * i + 16 * cs + cc];
    }
    return;
  }
  const bool live = 16 * w < rows;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  typedef const d4 __attribute__((address_space(1)))* g4ptr;
  // fragments of X_ii for the second product: requested in front of the first (they depend on nothing here)
  d4 xf[8];
  if (live && xa.mode != 1) {
    gcptr xp = (gcptr)(Xb + (128 * i + 16 * w + c) * n + 128 * i + 4 * g);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
      if (kc <= wu) xf[kc] = *reinterpret_cast<g4ptr>(xp + 16 * kc);
  }
  gptr own = Xb + (128 * i + 16 * w + g) * n + 128 * j + 16 * cs + c;      // X_ij's rows g + 4 t of this wave's block
  d4 acc = (d4){0, 0, 0, 0}, acc1 = (d4){0, 0, 0, 0};
  if (live) {
    const int kb0 = (xa.mode == 2) ? i - 1 : j, kb1 = (xa.mode == 1) ? i - 1 : i;
    if (xa.mode == 2 && j < i - 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = own[4 * t * n];
    }
    gcptr ap = Wb + (128 * kb0 + g) * n + 128 * i + 16 * w + c;      // L_ik^T: row = k index, column = row of L
    gcptr bp = (gcptr)(Xb + (128 * kb0 + g) * n + 128 * j + 16 * cs + c);
    const int steps = 32 * (kb1 - kb0);
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += 16) {
      double av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        av[u] = ap[4 * (s0 + u) * n];
        bv[u] = bp[4 * (s0 + u) * n];
      }
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        acc = mfma_f64(av[u], bv[u], acc);
        acc1 = mfma_f64(av[u + 1], bv[u + 1], acc1);
      }
    }
    acc += acc1;
    if (xa.mode == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) own[4 * t * n] = acc[t];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) T[(16 * w + g + 4 * t) * 17 + c] = acc[t];
    }
  }
  if (xa.mode == 1) return;
  __syncthreads();
  if (!live) return;
  d4 r0 = (d4){0, 0, 0, 0}, r1 = (d4){0, 0, 0, 0};
#pragma unroll
  for (int kc = 0; kc < 8; kc += 2) {
    if (kc <= wu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) r0 = mfma_f64(xf[kc][t], T[(16 * kc + 4 * g + t) * 17 + c], r0);
    }
    if (kc + 1 <= wu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) r1 = mfma_f64(xf[kc + 1][t], T[(16 * (kc + 1) + 4 * g + t) * 17 + c], r1);
    }
  }
  r0 += r1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    own[4 * t * n] = -r0[t];
    if (XT) XT[(128 * j + 16 * cs + c) * n + 128 * i + 16 * w + g + 4 * t] = -r0[t];
  }
}
// the last block row of X (behind the last factor launch)
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_xrow(const CholXrow xa) {
  __shared__ double T[128 * 17];
  for (int t = blockIdx.x; t < xa.batch * xa.nx; t += gridDim.x) {
    chol_xrow(xa, (lptr)T, t);
    __syncthreads();
  }
}

__global__ __launch_bounds__(CHOL_THREADS) void k_chol_block(const PotrfItem it0, const int64_t sW, const int64_t sInv, const int64_t sScal,
                                                             const CholWide wa, const CholXrow xa, const CholXrow xb) {
  extern __shared__ __attribute__((aligned(16))) double chol_dyn[];
  __shared__ int s_info;
  if ((int)blockIdx.x >= wa.nchol) {
    // side tasks (wide-update sub-tiles, then slabs of the inverse's block row), dealt round-robin over the launch's further workgroups:
    // each of them holds a CU (the launch's LDS size), so the host caps their number
    const int nw = wa.nchol * wa.nwide, nxa = xa.batch * xa.nx, total = nw + nxa + xb.batch * xb.nx;
    for (int t = (int)blockIdx.x - wa.nchol; t < total; t += (int)gridDim.x - wa.nchol) {
      if (t < nw) {
        chol_wide_update(wa, t);
      } else {
        if (t < nw + nxa) chol_xrow(xa, (lptr)chol_dyn, t - nw);
        else chol_xrow(xb, (lptr)chol_dyn, t - nw - nxa);
        __syncthreads();
      }
    }
    return;
  }
  // the item of matrix blockIdx.x from the one of matrix 0 (kernel arguments: no dependent load in front of the block's own loads)
  PotrfItem it = it0;
  it.W += (int64_t)blockIdx.x * sW;
  it.Linv += (int64_t)blockIdx.x * sInv;
  if (it.scal) it.scal += (int64_t)blockIdx.x * sScal;
  const int n = it.n, ld = n + 5, nb = n >> 4, nreal = it.nreal;      // n = 128, or 64 at a ragged end
  const int sh = (n == 128) ? 7 : 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lptr W = (lptr)chol_dyn;
  lptr Xd = (lptr)(chol_dyn + n * ld);
  lptr red = (lptr)(chol_dyn + n * ld + nb * 16 * 17);
  gptr Wg = (gptr)it.W;
  if (tid == 0) s_info = 0;
  // accumulated logdet / info of the earlier blocks: requested with the block's loads (the read-modify-write at the end waits for nothing)
  double sc0 = 0.0, sc1 = 0.0;
  if (tid == 0 && it.scal && (it.pad & 16)) {
    sc0 = it.scal[0];
    sc1 = it.scal[1];
  }
  // lower block triangle, the 16 x 16 diagonal blocks whole (mirrored from their lower halves); identity beyond the real order.  ONE round
  // trip: every load of the thread in flight at once, and only the blocks that are kept are requested (the mirrored reads of the upper
  // blocks — a cache line per lane — were 44 % of the requests and most of the lines)
  {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int idx = u * CHOL_THREADS + tid;
      const int i = idx >> sh, j = idx & (n - 1);
      const bool need = idx < n * n && (j >> 4) <= (i >> 4) && i < nreal && j < nreal;
      v[u] = need ? Wg[(j <= i) ? (int64_t)i * it.ld + j : (int64_t)j * it.ld + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int idx = u * CHOL_THREADS + tid;
      const int i = idx >> sh, j = idx & (n - 1);
      if (idx < n * n && (j >> 4) <= (i >> 4)) W[i * ld + j] = (i < nreal && j < nreal) ? v[u] : ((i == j) ? 1.0 : 0.0);
    }
  }
  asm volatile("" : "+v"(sc0), "+v"(sc1));
  __syncthreads();
  long long tl = 0;
  lds_chol_inverse<false>(W, Xd, n, ld, (gptr)it.Linv, (gptr) nullptr, (int64_t)it.ld, &s_info, nullptr, tl);
  double s = 0.0;
  for (int i = tid; i < nreal; i += CHOL_THREADS) s += 2.0 * log(W[i * ld + i]);
  s = sum_wave(s);
  if (lane == 0) red[wave] = s;
  if (!(it.pad & 8))
    for (int idx = tid; idx < n * n; idx += CHOL_THREADS) {
      const int i = idx >> sh, j = idx & (n - 1);
      Wg[(int64_t)i * it.ld + j] = (j <= i) ? W[i * ld + j] : 0.0;
    }
  __syncthreads();
  if (tid == 0 && it.scal) {
    double ldv = 0.0;
    for (int w = 0; w < CHOL_NW; ++w) ldv += red[w];
    if (it.pad & 16) {
      it.scal[0] = sc0 + ldv;
      if (s_info && sc1 == 0.0) it.scal[1] = (double)(it.info_offset + s_info);
    } else {
      it.scal[0] = ldv;
      it.scal[1] = s_info ? (double)(it.info_offset + s_info) : 0.0;
    }
  }
}

// Between two diagonal blocks of the look-ahead form: panel solve and the update of block column p + 1 in ONE launch.  A workgroup owns a
// 32 x 32 sub-tile (strip s of the rows below block p, strip b of block p + 1's rows: b <= s inside the diagonal tile) and computes both
// 32-row strips of the panel it needs ITSELF — La = A_s,p L_pp^-T, Lb = A_b,p L_pp^-T from the inverse the factor launch left, column
// block jc of the result summing the chunks k <= jc only (L_pp^-1 lower triangular) — so nothing in the launch waits for another
// workgroup: 144 MFMAs per wave, then 8 for the tile.  Operands go from L2 straight to MFMA registers as in k_wgrad_coop (a lane loads
// 4 consecutive k of one row).  The strips pass through LDS to become operands of the tile update; the
// b = 0 workgroups park their La TRANSPOSED above the block diagonal (where the wide update of the next factor launch and
// k_mirror_panels expect the panel).  grid (tasks, batch).
struct CholPanel {
  double* W;
  const double* dinv;      // L_pp^-1 of matrix 0 (lower triangular, upper 16 x 16 blocks never read), leading dimension n
  int64_t stride, dinv_stride;
  int32_t n, p;
};
#define CPL_LD 132
__global__ __launch_bounds__(256) void k_chol_panel(const CholPanel a) {
  __shared__ __attribute__((aligned(32))) double S[64 * CPL_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int64_t n = a.n;
  const int r0 = (a.p + 1) * 128, pK = a.p * 128;
  const int nrem = a.n - r0, nbb = (nrem < 128 ? nrem : 128) >> 5;
  int idx = blockIdx.x, s, b;
  const int ndiag = nbb * (nbb + 1) / 2;
  if (idx < ndiag) {
    s = 0;
    while ((s + 1) * (s + 2) / 2 <= idx) ++s;
    b = idx - s * (s + 1) / 2;
  } else {
    idx -= ndiag;
    s = nbb + idx / nbb;
    b = idx % nbb;
  }
  gptr Wb = (gptr)(a.W + (int64_t)blockIdx.y * a.stride);
  gcptr Li = (gcptr)(a.dinv + (int64_t)blockIdx.y * a.dinv_stride);
  const int ra = r0 + 32 * s, rb = r0 + 32 * b;
  const int bi = w >> 1, bj = w & 1;
  gptr Cp = Wb + (ra + 16 * bi + g) * n + rb + 16 * bj + c;
  d4 cin, cacc = (d4){0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) cin[t] = Cp[4 * t * n];
  typedef const d4 __attribute__((address_space(1)))* g4ptr;
  // wave w owns 16 rows of the two strips (w < 2: La, else Lb) and all eight column blocks of them: its eight A fragments are requested up
  // front (one round trip), the 36 fragments of L_pp^-1 (the same for every wave and workgroup: L2 hits) stream through a ring RING steps
  // ahead of the MFMAs that read them (a step = 4 MFMAs: left to the scheduler the loads sink to their uses and every step pays the trip)
  gcptr rowp = (gcptr)(Wb + ((w < 2 ? ra + 16 * w : rb + 16 * (w - 2)) + c) * n + pK + 4 * g);
  d4 pa[8];
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) pa[kc] = *reinterpret_cast<g4ptr>(rowp + 16 * kc);
  constexpr int RING = 8;
  d4 q[RING];
  gcptr lip = Li + c * n + 4 * g;
  // step order: chunk-major (kc = 0: jc = 0 .. 7, kc = 1: jc = 1 .. 7, ...): consecutive MFMAs go to different accumulators
  constexpr int SJC[36] = {0, 1, 2, 3, 4, 5, 6, 7, 1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};
  constexpr int SKC[36] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};
  d4 acc[8];
#pragma unroll
  for (int jc = 0; jc < 8; ++jc) acc[jc] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int st = 0; st < RING; ++st) q[st] = *reinterpret_cast<g4ptr>(lip + 16 * SJC[st] * n + 16 * SKC[st]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int st = 0; st < 36; ++st) {
    const int jc = SJC[st], kc = SKC[st];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[jc] = mfma_f64(pa[kc][t], q[st % RING][t], acc[jc]);
    __builtin_amdgcn_sched_barrier(0);
    if (st + RING < 36) q[st % RING] = *reinterpret_cast<g4ptr>(lip + 16 * SJC[st + RING] * n + 16 * SKC[st + RING]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int jc = 0; jc < 8; ++jc)
#pragma unroll
    for (int t = 0; t < 4; ++t) S[(16 * w + g + 4 * t) * CPL_LD + 16 * jc + c] = acc[jc][t];
  __syncthreads();
  {
    const d4* Sa = reinterpret_cast<const d4*>(S + (16 * bi + c) * CPL_LD + 4 * g);
    const d4* Sb = reinterpret_cast<const d4*>(S + (32 + 16 * bj + c) * CPL_LD + 4 * g);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      const d4 av = Sa[4 * kc], bv = Sb[4 * kc];
#pragma unroll
      for (int t = 0; t < 4; ++t) cacc = mfma_f64(av[t], bv[t], cacc);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) Cp[4 * t * n] = cin[t] - cacc[t];
  if (b == 0) {
    gptr Mp = Wb + (int64_t)pK * n + ra + (tid & 31);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int k = it * 8 + (tid >> 5);
      Mp[k * n] = S[(tid & 31) * CPL_LD + k];
    }
  }
}

#define BCB 128     // diagonal block of the blocked factorisation
int bigchol_build(dsdgp_ctx* ctx, BigChol& P, double* W, double* Linv, double* LinvT, double* scal, int batch, int64_t stride,
                  int64_t scal_stride, int n, int nreal, double* Tbuf, bool from_tri, bool need_factor) {
  DS_CHECK_ARG(n % 64 == 0 && n >= 128 && batch >= 1);
  P.need_factor = need_factor;
  P.n = n; P.batch = batch; P.nb = from_tri ? n / 64 : ceil_div(n, BCB);
  P.W = W; P.Linv = Linv; P.LinvT = LinvT; P.scal = scal; P.Tbuf = Tbuf;
  P.stride = stride; P.scal_stride = scal_stride;
  P.want_inverse = Linv != nullptr;
  P.from_tri = from_tri;
  const int nb = P.nb;
  // plan-owned scratch (BCB x n per matrix) unless the caller supplies one
  const int bs = from_tri ? 64 : BCB;
  const size_t tb = Tbuf ? 0 : (size_t)batch * bs * n * sizeof(double);
  void* tblock = nullptr;
  if (tb) {
    DS_HIP(hipMalloc(&tblock, tb));
    DS_HIP(hipMemset(tblock, 0, tb));
    Tbuf = (double*)tblock;
    P.Tbuf = Tbuf;
  }
  P.tbuf_block = tblock;
  std::vector<PotrfItem> items((size_t)nb * batch);
  for (int p = 0; p < nb; ++p)
    for (int b = 0; b < batch; ++b) {
      const int64_t off = (int64_t)b * stride + (int64_t)p * bs * n + p * bs;
      const int bn = std::min(bs, n - p * bs);                // 64 at a ragged end (n a multiple of 64)
      int nr = nreal - p * bs;
      nr = nr < 0 ? 0 : (nr > bn ? bn : nr);
      // without a requested inverse the inverses of the diagonal blocks (needed by the panel solve) live in Tbuf
      double* dinv = Linv ? Linv + off : Tbuf + (int64_t)b * bs * n + p * bs;
      // (block 0 SETS logdet / info, the later blocks accumulate: no memset of `scal` in front of the sequence)
      items[(size_t)p * batch + b] = PotrfItem{W + off, dinv, nullptr, scal ? scal + b * scal_stride : nullptr,
                                               bn, n, nr, p == 0 ? 0 : 16, p * bs, 0};
    }
  std::vector<GemmProblem> gp;
  P.tiles.clear();
  P.nprob.clear();
  P.first.clear();
  // one launch = a list of problems planned together (tile_start relative to the launch)
  auto add_launch = [&](std::vector<GemmProblem>& list, int allow_big) {
    P.tiles.push_back(gemm_plan(list.data(), (int)list.size(), allow_big));
    P.nprob.push_back((int)list.size());
    P.first.push_back((int)gp.size());
    for (auto& g : list) gp.push_back(g);
  };
  auto add = [&](GemmProblem& g, int mode) {
    std::vector<GemmProblem> one{g};
    add_launch(one, mode);
  };
  auto mk = [&](const double* A, const double* B, double* C, int m, int nn, int k, int tA, int tB, double alpha, double beta, int lower) {
    GemmProblem g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.m = m; g.n = nn; g.k = k;
    g.lda = n; g.ldb = n; g.ldc = n;
    g.transA = tA; g.transB = tB; g.batch = batch; g.sA = stride; g.sB = stride; g.sC = stride; g.batch_reduce = 0;
    g.alpha = alpha; g.beta = beta; g.lower_only = lower;
    return g;
  };
  const bool lookahead_off = getenv("DSDGP_CHOL_LOOKAHEAD") && atoi(getenv("DSDGP_CHOL_LOOKAHEAD")) == 0;      // (A/B aid; read per plan)
  // the wide-update workgroups of the look-ahead form each hold a whole CU (the factor launch's LDS): beyond 16 blocks the plain sequence
  P.lookahead = !from_tri && !lookahead_off && P.nb >= 2 && P.nb <= 16;
  P.dinv = Linv ? Linv : Tbuf;
  P.dinv_stride = Linv ? stride : (int64_t)bs * n;
  P.dinv_step = Linv ? (int64_t)BCB * n + BCB : BCB;
  if (!from_tri && !P.lookahead) {
    for (int p = 0; p + 1 < P.nb; ++p) {
      const int rem = n - (p + 1) * BCB;
      double* panel = W + (int64_t)(p + 1) * BCB * n + p * BCB;
      // the panel, TRANSPOSED, into the mirror blocks above the diagonal (unused by the factorisation):  L_ip^T = L_pp^-1 A_ip^T.
      // Out of place, so the LDS-free 64 x 64 kernel can take it (16 us; in place only a kernel whose workgroup owns all 128 columns
      // of its rows is safe — the 128 x 128 kernel at 8 barrier-separated k-steps: 26 us).  k_mirror_panels moves the panels
      // back below the diagonal at the end.
      double* mirror = W + (int64_t)p * BCB * n + (p + 1) * BCB;
      GemmProblem g1 = mk(Linv ? Linv + (int64_t)p * BCB * n + p * BCB : Tbuf + p * BCB, panel, mirror, BCB, rem, BCB, 0, 1, 1.0, 0.0, 0);
      if (!Linv) g1.sA = (int64_t)BCB * n;
      g1.tri = 2;                                        // L_pp^-1 lower-triangular
      add(g1, 0);
      // A_ij -= L_ip L_jp^T  (lower tiles only), both operands read from the transposed panel
      GemmProblem g2 = mk(mirror, mirror, W + (int64_t)(p + 1) * BCB * n + (p + 1) * BCB, rem, rem, BCB, 1, 0, -1.0, 1.0, 1);
      add(g2, 0);
    }
  }
  P.xrows = P.want_inverse && P.lookahead;      // block rows of X inside the factor launches (chol_xrow) instead
  if (P.want_inverse && !P.xrows) {
    // X = L^-1 by RECURSIVE DOUBLING over the 64 x 64 diagonal inverses the factor kernel left on Linv's diagonal: at block
    // size s every pair of adjacent diagonal blocks [X11 0; X21 X22] gets X21 = -X22 (L21 X11), all n / 2s pairs (and all
    // matrices of the batch) in ONE grouped launch per product: 2 log2(n / 64) launches (8 at n = 1024) instead of the
    // 2 (n / 64 - 1) of the block-row recurrence, and the upper levels are big enough for the 128 x 128 kernel.
    // T = L21 X11 is parked in the mirror position of the X21 block inside the scratch matrix S (LinvT when the caller wants
    // it — the final transpose overwrites it — else a plan-owned n x n).
    double* S = LinvT;
    if (!S) {
      DS_HIP(hipMalloc(&P.inv_block, (size_t)batch * n * n * sizeof(double)));
      S = (double*)P.inv_block;
    }
    const int64_t sS = LinvT ? stride : (int64_t)n * n;
    for (int sblk = from_tri ? 64 : BCB; sblk < n; sblk *= 2) {
      std::vector<GemmProblem> l1, l2;
      for (int r0 = sblk; r0 < n; r0 += 2 * sblk) {
        const int c0 = r0 - sblk;
        const int rows = (n - r0 < sblk) ? n - r0 : sblk;      // ragged last pair (n / 64 not a power of two)
        GemmProblem g1 = mk(W + (int64_t)r0 * n + c0, Linv + (int64_t)c0 * n + c0, S + (int64_t)r0 * n + c0, rows, sblk, sblk, 0, 0,
                            1.0, 0.0, 0);
        g1.sC = sS; g1.tri = 1;                                 // X11 lower-triangular: k >= n
        l1.push_back(g1);
        GemmProblem g2 = mk(Linv + (int64_t)r0 * n + r0, S + (int64_t)r0 * n + c0, Linv + (int64_t)r0 * n + c0, rows, sblk, rows, 0, 0,
                            -1.0, 0.0, 0);
        g2.sB = sS; g2.tri = 2;                                 // X22 lower-triangular: k <= m
        l2.push_back(g2);
      }
      add_launch(l1, sblk <= 256 ? 0 : 1);      // short K: as above
      add_launch(l2, sblk <= 256 ? 0 : 1);
    }
  }
  P.items0.clear();
  for (int p = 0; p < nb; ++p) {      // matrix 0's item per block: k_chol_block's argument
    P.items0.push_back(items[(size_t)p * batch]);
    // nobody reads L: the panel solve, the wide update and the rows of the inverse all work from L_pp^-1 — the block's own write-back goes too
    if (P.xrows && !P.need_factor) P.items0.back().pad |= 8;
  }
  const size_t ib = round_up(items.size() * sizeof(PotrfItem), 256), gb = round_up(gp.size() * sizeof(GemmProblem) + 256, 256);
  DS_HIP(hipMalloc(&P.dev_block, ib + gb));
  P.diag_items = (PotrfItem*)P.dev_block;
  P.gp = (GemmProblem*)((char*)P.dev_block + ib);
  DS_HIP(hipMemcpy(P.diag_items, items.data(), items.size() * sizeof(PotrfItem), hipMemcpyHostToDevice));
  if (!gp.empty()) DS_HIP(hipMemcpy(P.gp, gp.data(), gp.size() * sizeof(GemmProblem), hipMemcpyHostToDevice));
  return DSDGP_OK;
}

void bigchol_free(BigChol& P) {
  if (P.inv_block) hipFree(P.inv_block);
  P.inv_block = nullptr;
  if (P.dev_block) hipFree(P.dev_block);
  if (P.tbuf_block) hipFree(P.tbuf_block);
  P.dev_block = nullptr;
  P.tbuf_block = nullptr;
}

int bigchol_run(dsdgp_ctx* ctx, const BigChol& P) {
  ProfScope ps(ctx, "potrf");
  const int nb = P.nb, batch = P.batch;
  int gi = 0;
  auto launch = [&](int i) { gemm_dispatch(P.gp + P.first[i], P.nprob[i], P.tiles[i], ctx->stream); };
  if (!P.from_tri) {
    const size_t lds = chol_lds_bytes(BCB);
    static bool lds_set = false;      // the attribute is sticky: one driver call
    if (!lds_set) {
      DS_HIP(hipFuncSetAttribute((const void*)k_chol_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_set = true;
    }
    for (int p = 0; p < nb; ++p) {
      CholWide wa{P.W, P.stride, P.n, p, 0, batch};
      if (P.lookahead && p >= 1) {
        const int ns64 = (P.n - (p + 1) * BCB) / 64;      // 64-row strips from block p + 1 on
        wa.nwide = ns64 > 0 ? ns64 * (ns64 + 1) / 2 : 0;
      }
      CholXrow xa{P.W, P.Linv, P.LinvT, P.stride, P.n, p - 1, 0, batch, 0, 0}, xb = xa;
      if (P.xrows && p >= 1) xa.nx = 8 * (p - 1 + (P.LinvT ? 1 : 0));
      if (P.xrows && p == nb - 1) {      // the early terms of the last row
        xb.i = p;
        xb.nx = 8 * (p - 1);
        xb.mode = 1;
      }
      const int side = std::min(batch * (wa.nwide + xa.nx + xb.nx), std::max(32, 248 - batch));      // (a CU each: one round of them)
      DS_LAUNCH(k_chol_block, dim3(batch + side), dim3(CHOL_THREADS), lds, ctx->stream, P.items0[p], P.stride, P.dinv_stride,
                P.scal_stride, wa, xa, xb);
      if (p + 1 < nb) {
        if (P.lookahead) {
          const int nrem = P.n - (p + 1) * BCB, nbb = std::min(nrem, BCB) / 32, ns = nrem / 32;
          const CholPanel pa{P.W, P.dinv + (int64_t)p * P.dinv_step, P.stride, P.dinv_stride, P.n, p};
          DS_LAUNCH(k_chol_panel, dim3(nbb * (nbb + 1) / 2 + (ns - nbb) * nbb, batch), dim3(256), 0, ctx->stream, pa);
        } else {
          launch(gi++);
          launch(gi++);
        }
      }
    }
    if (P.xrows) {
      CholXrow xa{P.W, P.Linv, P.LinvT, P.stride, P.n, nb - 1, 8 * (nb - 1 + (P.LinvT ? 1 : 0)), batch, 2, 0};
      if (xa.nx) DS_LAUNCH(k_chol_xrow, dim3(std::min(batch * xa.nx, 2048)), dim3(CHOL_THREADS), 0, ctx->stream, xa);
    }
    // (with the inverse formed from the parked panels nothing but a caller that reads L itself needs them moved below the diagonal)
    if (P.need_factor || !P.xrows)
      DS_LAUNCH(k_mirror_panels, dim3(std::min(1024, (P.n / 16) * (P.n / 16)), batch), dim3(256), 0, ctx->stream, P.W, P.n, P.stride, BCB);
  } else if (P.want_inverse) {
    DS_LAUNCH(k_trtri_diag64, dim3(nb * batch), dim3(256), 0, ctx->stream, P.diag_items);
  }
  if (P.want_inverse && !P.xrows) {
    while (gi < (int)P.tiles.size()) launch(gi++);
    if (P.LinvT)
      DS_LAUNCH(k_transpose_lower, dim3(256, batch), dim3(256), 0, ctx->stream, P.Linv, P.LinvT, P.n, P.stride);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
