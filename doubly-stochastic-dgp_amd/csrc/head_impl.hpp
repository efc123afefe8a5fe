// Head of an ELBO evaluation in ONE launch for models whose layers all have Mp <= 128 and D_in <= 16 (included by model.hip):
//   block column x = 0 of grid (1 + nprep, L):  Ku = K(Z,Z) + (white + jitter) I  (layers.py:171)  ->  Lu = chol(Ku)  (layers.py:172)
//                                               ->  Lu^-1, Lu^-T (the two triangular solves of layers.py:186,188 as products), log det
//   block columns x >= 1                     :  the parameter transforms / paddings of prep_body
// Before round 3 this was k_prep_kuu (8.6 us) + k_potrf_trtri (48.6 us, three workgroups on a 256-CU chip at the head of every step).
//
// The factorisation never leaves the LDS (n x (n + 5) doubles, 136 KB at n = 128) and is organised around the one thing that bounds
// it — the chain of n dependent pivots:
//  * Panel factorisation by ROWS-IN-LANES: a wave holds the 16 rows of the diagonal block in lanes 0..15 and 48 rows of the panel
//    below it in lanes 16..63 (one row of 16 columns per lane, Chol16 of linalg.hpp).  The pivot loop that factors the diagonal
//    block scales and updates the panel rows in the same instructions, so the panel comes out as L_ij = A_ij L_jj^-T with NO inverse
//    of the diagonal block, no MFMA panel product and no barrier between "factor" and "panel" (the old kernel: Cholesky of the
//    block, its inverse by forward substitution, barrier, MFMA product: ~8 K clocks per block column on the critical path).  Up to three
//    waves share a tall panel; each repeats the 16 diagonal rows (no inter-wave traffic inside the pivot loop).
//  * Right-looking trailing update on the MFMA pipe with look-ahead: after the tiles of block column jb+1 are updated (all waves, one
//    tile each), the panel waves factor that column while the other waves finish the remaining tiles — they touch disjoint columns.
//  * Lu^-1: the inverses of the nb diagonal blocks in parallel (one wave each, column-per-lane forward substitution), then block column
//    j of X = L^-1 by wave j (X_ij = -X_ii sum_k L_ik X_kj, the D layout of one product is the B operand of the next: registers only),
//    parked transposed in the unused upper triangle of the LDS matrix, then one coalesced write of Lu^-1 and Lu^-T.
//  * Odd leading dimension (n + 5): a row-per-lane access (panel loads / stores) then hits 32 distinct 8-byte banks per half-wave; the
//    MFMA fragment reads stay <= 3-way.
#pragma once

#define HEAD_THREADS 512
#define HEAD_NW 8
#define HEAD_MAX_N 128
#define HEAD_MAX_DIN 16

static inline size_t head_lds_bytes(int n) {
  const int nb = n / 16;
  return ((size_t)n * (n + 5) + (size_t)nb * 16 * 17 + n + 32) * sizeof(double);
}

__device__ __forceinline__ void head_factor(const LayerDev& v, const double* __restrict__ theta, double jitter, int white, double* dyn) {
  const int n = v.Mp, nb = n >> 4, M = v.M, Din = v.D_in, ld = n + 5;
  lptr W = (lptr)dyn;                                    // n x ld working matrix
  lptr Xd = (lptr)(dyn + n * ld);                        // nb x (16 x 17): Z / l staging first, inverses of the diagonal blocks later
  lptr dinv = (lptr)(dyn + n * ld + nb * 16 * 17);       // n: reciprocals of diag(L)
  lptr red = dinv + n;                                   // 8 partial sums
  __shared__ int s_info;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  if (tid == 0) s_info = 0;
  const double var = softplus_d(theta[v.off_kvar]) + SOFTPLUS_LOWER;
  const double wvar = v.has_white ? softplus_d(theta[v.off_wvar]) + SOFTPLUS_LOWER : 0.0;
  // ---- Z / lengthscale, zero-padded to 16 columns (row stride 17)
  for (int idx = tid; idx < n * 16; idx += HEAD_THREADS) {
    const int mrow = idx >> 4, d = idx & 15;
    double z = 0.0;
    if (mrow < M && d < Din)
      z = theta[v.off_Z + (int64_t)mrow * Din + d] / (softplus_d(theta[v.off_kls + (v.ard ? d : 0)]) + SOFTPLUS_LOWER);
    Xd[mrow * 17 + d] = z;
  }
  __syncthreads();
  // ---- Ku (lower block triangle; diagonal blocks whole) into the LDS, scaled squared distances to global for the adjoint (k_asm_kbar)
  {
    gptr R2 = (gptr)v.R2;
    for (int idx = tid; idx < n * nb; idx += HEAD_THREADS) {
      const int i = idx % n, q = idx / n;
      if (q > (i >> 4)) continue;
      double zi[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) zi[d] = Xd[i * 17 + d];
#pragma unroll 4
      for (int e = 0; e < 16; ++e) {
        const int j = 16 * q + e;
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
          if (d < Din) {
            const double df = zi[d] - Xd[j * 17 + d];
            r2 = fma(df, df, r2);
          }
        }
        double k = (i == j) ? 1.0 : 0.0;
        if (i < M && j < M) {
          k = kern_val_rt(v.kern_kind, r2, var);
          if (i == j) k += wvar + jitter;
        }
        W[i * ld + j] = k;
        R2[(int64_t)i * n + j] = r2;
        if (q < (i >> 4)) R2[(int64_t)j * n + i] = r2;
      }
    }
  }
  __syncthreads();

  // one panel: lanes 0..15 <- rows of diagonal block jb, lanes 16..63 <- 48 rows below it (slab `wave`); waves 0 .. npw-1
  auto panel = [&](int jb) {
    const int j0 = jb << 4;
    const int r = (lane < 16) ? j0 + lane : j0 + 16 + 48 * wave + (lane - 16);
    const bool valid = r < n;
    const int rl = valid ? r : n - 1;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = W[rl * ld + j0 + j];
    double myinv = 0.0;
    int bad = 0;
    Chol16<0>::run(a, lane < 16 ? lane : 99, myinv, bad);
    if (lane < 16) {
      if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) W[r * ld + j0 + j] = (j <= lane) ? a[j] : 0.0;
        dinv[r] = myinv;
        if (bad && lane == 0 && s_info == 0) s_info = j0 + bad;
      }
    } else if (valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) W[r * ld + j0 + j] = a[j];
    }
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
  auto panel_waves = [&](int jb) {                      // waves that hold rows of block column jb's panel (wave 0 always: the diagonal)
    const int prow = n - ((jb + 1) << 4);
    return prow > 0 ? (prow + 47) / 48 : 1;
  };

  if (wave < panel_waves(0)) panel(0);
  __syncthreads();
  for (int jb = 0; jb + 1 < nb; ++jb) {
    const int j0 = jb << 4;
    // (A) tiles of block column jb + 1, one per wave
    {
      const int ib = jb + 1 + wave;
      if (ib < nb) trail_tile(ib, jb + 1, j0);
    }
    __syncthreads();
    // (B) panel jb + 1 by the panel waves; the other waves update the tiles of the block columns >= jb + 2 (disjoint from the panel)
    const int npw = panel_waves(jb + 1);
    if (wave < npw) {
      panel(jb + 1);
    } else {
      const int nt = nb - jb - 2;
      const int cnt = nt * (nt + 1) / 2;
      for (int id = wave - npw; id < cnt; id += HEAD_NW - npw) {
        int ib2 = 0;
        while ((ib2 + 1) * (ib2 + 2) / 2 <= id) ++ib2;
        const int kb2 = id - ib2 * (ib2 + 1) / 2;
        trail_tile(jb + 2 + ib2, jb + 2 + kb2, j0);
      }
    }
    __syncthreads();
  }
  // ---- log det over the real (unpadded) part; the factor itself only for the white=True adjoint (gp_w1 reads Lu from Kp)
  {
    double s = 0.0;
    for (int i = tid; i < M; i += HEAD_THREADS) s += 2.0 * log(W[i * ld + i]);
    s = sum_wave(s);
    if (lane == 0) red[wave] = s;
  }
  if (white) {
    gptr Kp = (gptr)v.Kp;
    for (int idx = tid; idx < n * n; idx += HEAD_THREADS) {
      const int i = idx / n, j = idx % n;
      Kp[idx] = (j > i) ? 0.0 : W[i * ld + j];
    }
  }
  // ---- inverses of the diagonal blocks: wave w -> block w, lane = column, forward substitution down the column
  if (wave < nb && lane < 16) {
    const int j0 = wave << 4;
    double x[16], sacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      x[k] = sacc[k] * dinv[j0 + k];
#pragma unroll
      for (int i = k + 1; i < 16; ++i) sacc[i] = fma(-W[(j0 + i) * ld + j0 + k], x[k], sacc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Xd[wave * 272 + i * 17 + lane] = x[i];
  }
  __syncthreads();
  if (tid == 0) {
    double ldv = 0.0;
    for (int w = 0; w < HEAD_NW; ++w) ldv += red[w];
    v.scal[0] = ldv;
    v.scal[1] = (double)s_info;
  }
  // ---- block column `wave` of X = L^-1, rows below the diagonal block; X_ij^T parked at W[(16 j + c)][16 i + r] (upper triangle)
  if (wave < nb) {
    const int jcol = wave;
    d4 x[HEAD_MAX_N / 16];
#pragma unroll
    for (int rel = 0; rel < HEAD_MAX_N / 16; ++rel) {
      const int ib = jcol + rel;
      if (ib < nb) {
        lptr Xi = Xd + ib * 272;
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
              const double a = W[(ib * 16 + c) * ld + kb * 16 + 4 * s + g];
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
#pragma unroll
          for (int t = 0; t < 4; ++t) W[(jcol * 16 + c) * ld + ib * 16 + g + 4 * t] = R[t];
        }
      }
    }
  }
  __syncthreads();
  // ---- Lu^-1 and Lu^-T, whole matrices (zeros included), coalesced
  {
    gptr Linv = (gptr)v.Linv;
    gptr LinvT = (gptr)v.LinvT;
    for (int idx = tid; idx < n * n; idx += HEAD_THREADS) {
      const int r = idx / n, cc = idx % n, rb = r >> 4, cb = cc >> 4;
      double lo = 0.0, up = 0.0;
      if (rb > cb) lo = W[cc * ld + r];
      else if (rb == cb) lo = Xd[rb * 272 + (r & 15) * 17 + (cc & 15)];
      if (cb > rb) up = W[r * ld + cc];
      else if (rb == cb) up = Xd[rb * 272 + (cc & 15) * 17 + (r & 15)];
      Linv[idx] = lo;
      LinvT[idx] = up;
    }
  }
}

// grid (1 + nprep + hr.nblk, L), HEAD_THREADS threads, head_lds_bytes(max Mp) of dynamic LDS
__global__ __launch_bounds__(HEAD_THREADS) void k_head(const double* __restrict__ theta, const LayerDev* __restrict__ layers,
                                                       double* __restrict__ lik_const, int64_t off_lik, int lik_gauss, double jitter,
                                                       int nprep, int keep_kuu, int white, const HeadRand hr) {
  extern __shared__ __attribute__((aligned(16))) double head_dyn[];
  const int bx = (int)blockIdx.x, l = (int)blockIdx.y;
  if (bx > nprep) {                   // N(0,1) draws of layer l (Philox stream l, as dsdgp_randn / k_randn number them)
    if (hr.count[l] > 0)
      randn_body(hr.seed, (uint64_t)l, hr.count[l], hr.out[l], (int64_t)(bx - 1 - nprep) * HEAD_THREADS + threadIdx.x, (int64_t)hr.nblk * HEAD_THREADS);
    return;
  }
  const LayerDev v = layers[l];
  if (bx > 0)
    prep_body(v, theta, lik_const, off_lik, lik_gauss, bx - 1, nprep);
  else if (!keep_kuu)                 // keep_kuu: the factor of the unchanged Ku stays in place (dsdgp_model_track_theta)
    head_factor(v, theta, jitter, white, head_dyn);
}
