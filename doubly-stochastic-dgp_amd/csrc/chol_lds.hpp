// Cholesky factor and inverse factor of ONE LDS-resident matrix (n <= 128, n a multiple of 16) by a 512-thread workgroup — the
// factorisation core of the head launch (head_impl.hpp: config-2-sized models) and of the 128 x 128 diagonal blocks of the multi-
// workgroup blocked factorisation (linalg.hip: k_chol128, M >= 256).  Organised around the chain of n dependent pivots:
//  * Panel factorisation by ROWS-IN-LANES: a wave holds the 16 rows of the diagonal block in lanes 0..15 and 48 rows of the panel
//    below it in lanes 16..63 (one row of 16 columns per lane, Chol16 of linalg.hpp).  The pivot loop that factors the diagonal
//    block scales and updates the panel rows in the same instructions, so the panel comes out as L_ij = A_ij L_jj^-T with NO inverse
//    of the diagonal block, no MFMA panel product and no barrier between "factor" and "panel".  Up to three waves share a tall
//    panel; each repeats the 16 diagonal rows (no inter-wave traffic inside the pivot loop).
//  * Right-looking trailing update on the MFMA pipe with look-ahead: after the tiles of block column jb+1 are updated (all waves, one
//    tile each), the panel waves factor that column while the other waves finish the remaining tiles — they touch disjoint columns.
//  * X = L^-1: the inverse of diagonal block jb by wave 7 beside the panel waves (the same pivot loop on identity rows), then the blocks
//    of row jb of X by the idle waves (X_ij = -X_ii sum_k L_ik X_kj; the D layout of one product is the B operand of the next),
//    parked transposed in the unused upper triangle of the LDS matrix and streamed to global memory as they are produced.
//  * Odd leading dimension (n + 5): a row-per-lane access (panel loads / stores) hits 32 distinct 8-byte banks per half-wave.
#pragma once
#include "linalg.hpp"

#define CHOL_THREADS 512
#define CHOL_NW 8
#define CHOL_MAX_N 128

// LDS doubles: n x (n + 5) working matrix | nb x (16 x 17) diagonal-block inverses (scratch before the factorisation) | n + 32 scratch
static inline size_t chol_lds_bytes(int n) {
  const int nb = n / 16;
  return ((size_t)n * (n + 5) + (size_t)nb * 16 * 17 + n + 32) * sizeof(double);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL store of the wave (vmcnt(0)),
// i.e. for the write acknowledgements of the inverse blocks streamed out under the factorisation (measured: 1.4 K clocks of barrier
// wait per block column)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// W: n x ld (ld = n + 5) in LDS: the lower block triangle of the SPD matrix with its 16 x 16 diagonal blocks WHOLE; on return L in the
// lower triangle (diagonal blocks zero above the diagonal), X^T parked above.  Xd: nb x 272 LDS scratch.  Linv / LinvT: global outputs
// with leading dimension ldo (only their lower / upper halves are written).  *s_info (LDS, zero on entry): 1-based index of a failing
// pivot.  tscal != NULL: phase clocks of thread 0 into tscal[4..10] (debug aid).  Ends behind a workgroup barrier.
template <bool HAS_T>
__device__ __forceinline__ void lds_chol_inverse(lptr W, lptr Xd, const int n, const int ld, gptr Linv, gptr LinvT, const int64_t ldo,
                                                 int* s_info, double* tscal, long long& tlast) {
  const int nb = n >> 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
#define CHOL_STAMP(i) do { if (tscal && tid == 0) { const long long tn = __builtin_amdgcn_s_memtime(); tscal[2 + (i)] = (double)(tn - tlast); tlast = tn; } } while (0)
  // one panel: lanes 0..15 <- rows of diagonal block jb, lanes 16..63 <- 48 rows below it (slab `wave`); waves 0 .. npw-1.
  // The rows below the diagonal block are stored at once; the factored diagonal block itself only after the phase's barrier
  // (panel_commit): the other panel waves and wave 7 read the UNFACTORED block in this phase.
  double pa[16];
  auto panel = [&](int jb) {
    const int j0 = jb << 4;
    const int r = (lane < 16) ? j0 + lane : j0 + 16 + 48 * wave + (lane - 16);
    const bool valid = r < n;
    const int rl = valid ? r : n - 1;
#pragma unroll
    for (int j = 0; j < 16; ++j) pa[j] = W[rl * ld + j0 + j];
    double myinv = 0.0;
    int bad = 0;
    __builtin_amdgcn_s_setprio(3);        // the pivot chain is the critical path: ahead of the wave that shares this SIMD
    Chol16<0>::run(pa, lane < 16 ? lane : 99, myinv, bad);
    __builtin_amdgcn_s_setprio(0);
    if (lane >= 16 && valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) W[r * ld + j0 + j] = pa[j];
    }
    if (wave == 0 && lane == 0 && bad && *s_info == 0) *s_info = j0 + bad;
  };
  auto panel_commit = [&](int jb) {                      // wave 0, after the barrier that ends the panel's phase
    if (wave != 0 || lane >= 16) return;
    const int j0 = jb << 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) W[(j0 + lane) * ld + j0 + j] = (j <= lane) ? pa[j] : 0.0;
  };
  // one trailing tile: A_ik -= L_i,jb L_k,jb^T
  auto trail_tile = [&](int ib, int kb, int j0) {
    d4 acc;
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = W[(ib * 16 + g + 4 * r) * ld + kb * 16 + c];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      av[s] = W[(ib * 16 + c) * ld + j0 + 4 * s + g];
      bv[s] = W[(kb * 16 + c) * ld + j0 + 4 * s + g];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_f64(-av[s], bv[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) W[(ib * 16 + g + 4 * r) * ld + kb * 16 + c] = acc[r];
  };
  // X = L^-1 is built DURING the factorisation by the waves that are neither on the panel nor short of trailing tiles, and goes straight
  // to global memory (Lu^-1 and Lu^-T; their zero halves are never written: the workspace is zeroed when the model is created).
  // inverse of diagonal block jb with the SAME pivot loop as the panel: lanes 0..15 hold the rows of the block, lanes 16..31 the rows of
  // the identity — "panel rows" that come out as I L_jj^-T = X_jj^T.  Run by wave 7 beside the panel waves (same input, same phase):
  // X_jj costs nothing on the critical path and is ready one phase earlier than a forward substitution after the factorisation.
  auto panel_inv = [&](int jb) {
    const int j0 = jb << 4, col = lane & 15;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double w = W[(j0 + col) * ld + j0 + j];
      a[j] = (lane < 16) ? w : ((lane < 32 && j == col) ? 1.0 : 0.0);
    }
    double myinv = 0.0;
    int bad = 0;
    Chol16<0>::run(a, lane < 16 ? lane : 99, myinv, bad);
    if (lane >= 16 && lane < 32) {      // a[j] = X^T[col][j] = X[j][col]
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        Xd[jb * 272 + j * 17 + col] = a[j];
        Linv[(int64_t)(j0 + j) * ldo + j0 + col] = a[j];
        if constexpr (HAS_T) LinvT[(int64_t)(j0 + col) * ldo + j0 + j] = a[j];
      }
    }
  };
  // block (i, j), j < i:  X_ij = -X_ii sum_{k = j}^{i-1} L_ik X_kj  (X_jj from Xd, X_kj for k > j parked transposed at W[16 j + c][16 k + r]);
  // the D layout of one product is the B operand of the next
  auto xblock = [&](int i, int j) {
    d4 S0 = (d4){0, 0, 0, 0}, S1 = (d4){0, 0, 0, 0};
    {
      double av[4], bv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        av[s] = W[(16 * i + c) * ld + 16 * j + 4 * s + g];
        bv[s] = Xd[j * 272 + (4 * s + g) * 17 + c];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) S0 = mfma_f64(av[s], bv[s], S0);
    }
    for (int k = j + 1; k < i; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        av[s] = W[(16 * i + c) * ld + 16 * k + 4 * s + g];
        bv[s] = W[(16 * j + c) * ld + 16 * k + 4 * s + g];
      }
      if ((k - j) & 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) S1 = mfma_f64(av[s], bv[s], S1);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) S0 = mfma_f64(av[s], bv[s], S0);
      }
    }
    double xi[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xi[s] = -Xd[i * 272 + c * 17 + 4 * s + g];
    S0 += S1;
    d4 R = (d4){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) R = mfma_f64(xi[s], S0[s], R);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      W[(16 * j + c) * ld + 16 * i + g + 4 * t] = R[t];
      Linv[(int64_t)(16 * i + g + 4 * t) * ldo + 16 * j + c] = R[t];
      if constexpr (HAS_T) LinvT[(int64_t)(16 * j + c) * ldo + 16 * i + g + 4 * t] = R[t];
    }
  };
  auto panel_waves = [&](int jb) {                      // waves that hold rows of block column jb's panel (wave 0 always: the diagonal)
    const int prow = n - ((jb + 1) << 4);
    return prow > 0 ? (prow + 47) / 48 : 1;
  };

  if (wave < panel_waves(0)) panel(0);
  else if (wave == CHOL_NW - 1) panel_inv(0);
  lds_barrier();
  CHOL_STAMP(2);     // first panel
  long long tq[4] = {0, 0, 0, 0}, tl = 0;
#define CHOL_Q(i) do { if (tscal && tid == 0) { const long long tn = __builtin_amdgcn_s_memtime(); tq[i] += tn - tl; tl = tn; } } while (0)
  if (tscal && tid == 0) tl = __builtin_amdgcn_s_memtime();
  for (int jb = 0; jb + 1 < nb; ++jb) {
    const int j0 = jb << 4;
    panel_commit(jb);
    // (A) tiles of block column jb + 1, one per wave
    {
      const int ib = jb + 1 + wave;
      if (ib < nb) trail_tile(ib, jb + 1, j0);
    }
    CHOL_Q(0);
    lds_barrier();
    CHOL_Q(1);
    // (B) panel jb + 1 by the panel waves, the inverse of its diagonal block by wave 7; the other waves share the blocks of row jb of X
    // (all its inputs are final) and the tiles of the block columns >= jb + 2 (disjoint from the panel)
    const int npw = panel_waves(jb + 1);
    if (wave < npw) {
      panel(jb + 1);
    } else if (wave == CHOL_NW - 1) {
      panel_inv(jb + 1);
    } else {
      const int nx = jb;
      const int nt = nb - jb - 2;
      const int cnt = nt * (nt + 1) / 2;
      for (int task = wave - npw; task < nx + cnt; task += CHOL_NW - 1 - npw) {
        if (task < nx) {
          xblock(jb, task);
        } else {
          const int id = task - nx;
          int ib2 = 0;
          while ((ib2 + 1) * (ib2 + 2) / 2 <= id) ++ib2;
          const int kb2 = id - ib2 * (ib2 + 1) / 2;
          trail_tile(jb + 2 + ib2, jb + 2 + kb2, j0);
        }
      }
    }
    CHOL_Q(2);
    lds_barrier();
    CHOL_Q(3);
  }
  if (tscal && tid == 0) { tscal[7] = (double)tq[0]; tscal[8] = (double)tq[1]; tscal[9] = (double)tq[2]; tscal[10] = (double)tq[3]; }
  CHOL_STAMP(3);     // remaining block columns (with the inverse of all but the last block row underneath)
  // ---- the last block row of X, one block per wave
  panel_commit(nb - 1);
  if (wave < nb - 1) xblock(nb - 1, wave);
  lds_barrier();
#undef CHOL_STAMP
#undef CHOL_Q
}
