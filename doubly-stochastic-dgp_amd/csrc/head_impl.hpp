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

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL store of the wave (vmcnt(0)),
// i.e. for the write acknowledgements of the Lu^-1 blocks and distances streamed out under the factorisation (measured: 1.4 K clocks
// of barrier wait per block column)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define HEAD_STAMP(i) do { if (timing && tid == 0) { const long long tn = __builtin_amdgcn_s_memtime(); v.scal[2 + (i)] = (double)(tn - tlast); tlast = tn; } } while (0)
__device__ __forceinline__ void head_factor(const LayerDev& v, const double* __restrict__ theta, double jitter, int white, int timing,
                                            double* dyn) {
  const int n = v.Mp, nb = n >> 4, M = v.M, Din = v.D_in, ld = n + 5;
  lptr W = (lptr)dyn;                                    // n x ld working matrix: L in the lower triangle, X^T = L^-T parked in the upper
  lptr Xd = (lptr)(dyn + n * ld);                        // nb x (16 x 17): Z / l staging first, inverses of the diagonal blocks later
  lptr dinv = (lptr)(dyn + n * ld + nb * 16 * 17);       // n: |z / l|^2 for the distances
  lptr red = dinv + n;                                   // [0..7] partial sums, [16..31] 1 / lengthscale
  gptr Linv = (gptr)v.Linv;
  gptr LinvT = (gptr)v.LinvT;
  __shared__ int s_info;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  if (tid == 0) s_info = 0;
  long long tlast = timing ? __builtin_amdgcn_s_memtime() : 0;
  const double var = softplus_d(theta[v.off_kvar]) + SOFTPLUS_LOWER;
  const double wvar = v.has_white ? softplus_d(theta[v.off_wvar]) + SOFTPLUS_LOWER : 0.0;
  // ---- Z / lengthscale, zero-padded to 16 columns (row stride 17): the Z loads are in flight while 16 threads transform the lengthscales
  double zreg[HEAD_MAX_N * 16 / HEAD_THREADS];
#pragma unroll
  for (int u = 0; u < HEAD_MAX_N * 16 / HEAD_THREADS; ++u) {
    const int idx = tid + u * HEAD_THREADS, mrow = idx >> 4, d = idx & 15;
    zreg[u] = (mrow < M && d < Din) ? theta[v.off_Z + (int64_t)mrow * Din + d] : 0.0;
  }
  if (tid < 16) red[16 + tid] = (tid < Din) ? 1.0 / (softplus_d(theta[v.off_kls + (v.ard ? tid : 0)]) + SOFTPLUS_LOWER) : 0.0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < HEAD_MAX_N * 16 / HEAD_THREADS; ++u) {
    const int idx = tid + u * HEAD_THREADS, mrow = idx >> 4, d = idx & 15;
    if (mrow < n) Xd[mrow * 17 + d] = zreg[u] * red[16 + d];
  }
  __syncthreads();
  if (tid < n) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      const double x = Xd[tid * 17 + d];
      s = fma(x, x, s);
    }
    dinv[tid] = s;
  }
  __syncthreads();
  HEAD_STAMP(0);     // kernel start, descriptor / parameter loads, Z staging
  // ---- Ku (lower block triangle; diagonal blocks whole) into the LDS, one 16 x 16 block pair per wave and pass: the scaled squared
  // distances as |z_i|^2 + |z_j|^2 - 2 z_i . z_j with the Gram block on the MFMA pipe (clamped at 0, exactly 0 on the diagonal; the
  // same form as the chains' Kuf tile), also stored to global memory for the adjoint (k_asm_kbar).  The element-by-element form
  // (16 LDS reads, 16 FMAs and the index arithmetic per element) took 29 K clocks here, this one is bounded by the exp per element.
  {
    gptr R2 = (gptr)v.R2;
    const int npair = nb * (nb + 1) / 2;
    for (int p = wave; p < npair; p += HEAD_NW) {
      int ib = 0;
      while ((ib + 1) * (ib + 2) / 2 <= p) ++ib;
      const int q = p - ib * (ib + 1) / 2;
      d4 G = (d4){0, 0, 0, 0};
      {
        double za[4], zb[4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          za[s] = Xd[(16 * ib + c) * 17 + 4 * s + g];
          zb[s] = Xd[(16 * q + c) * 17 + 4 * s + g];
        }
        if (Din > 8) {
#pragma unroll
          for (int s = 2; s < 4; ++s) {
            za[s] = Xd[(16 * ib + c) * 17 + 4 * s + g];
            zb[s] = Xd[(16 * q + c) * 17 + 4 * s + g];
          }
        }
        G = mfma_f64(za[0], zb[0], G);
        G = mfma_f64(za[1], zb[1], G);
        if (Din > 8) {
          G = mfma_f64(za[2], zb[2], G);
          G = mfma_f64(za[3], zb[3], G);
        }
      }
      const int j = 16 * q + c;
      const double nj = dinv[j];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 16 * ib + g + 4 * t;
        double r2 = fmax(dinv[i] + nj - 2.0 * G[t], 0.0);
        if (i == j) r2 = 0.0;
        double k = (i == j) ? 1.0 : 0.0;
        if (i < M && j < M) {
          k = kern_val_rt(v.kern_kind, r2, var);
          if (i == j) k += wvar + jitter;
        }
        W[i * ld + j] = k;
        R2[(int64_t)i * n + j] = r2;
        if (q < ib) R2[(int64_t)j * n + i] = r2;
      }
    }
  }
  __syncthreads();
  HEAD_STAMP(1);     // Ku

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
    if (wave == 0 && lane == 0 && bad && s_info == 0) s_info = j0 + bad;
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
        Linv[(int64_t)(j0 + j) * n + j0 + col] = a[j];
        LinvT[(int64_t)(j0 + col) * n + j0 + j] = a[j];
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
      Linv[(int64_t)(16 * i + g + 4 * t) * n + 16 * j + c] = R[t];
      LinvT[(int64_t)(16 * j + c) * n + 16 * i + g + 4 * t] = R[t];
    }
  };
  auto panel_waves = [&](int jb) {                      // waves that hold rows of block column jb's panel (wave 0 always: the diagonal)
    const int prow = n - ((jb + 1) << 4);
    return prow > 0 ? (prow + 47) / 48 : 1;
  };

  if (wave < panel_waves(0)) panel(0);
  else if (wave == HEAD_NW - 1) panel_inv(0);
  lds_barrier();
  HEAD_STAMP(2);     // first panel
  long long tq[4] = {0, 0, 0, 0}, tl = 0;
#define HEAD_Q(i) do { if (timing && tid == 0) { const long long tn = __builtin_amdgcn_s_memtime(); tq[i] += tn - tl; tl = tn; } } while (0)
  if (timing && tid == 0) tl = __builtin_amdgcn_s_memtime();
  for (int jb = 0; jb + 1 < nb; ++jb) {
    const int j0 = jb << 4;
    panel_commit(jb);
    // (A) tiles of block column jb + 1, one per wave
    {
      const int ib = jb + 1 + wave;
      if (ib < nb) trail_tile(ib, jb + 1, j0);
    }
    HEAD_Q(0);
    lds_barrier();
    HEAD_Q(1);
    // (B) panel jb + 1 by the panel waves, the inverse of its diagonal block by wave 7; the other waves share the blocks of row jb of X
    // (all its inputs are final) and the tiles of the block columns >= jb + 2 (disjoint from the panel)
    const int npw = panel_waves(jb + 1);
    if (wave < npw) {
      panel(jb + 1);
    } else if (wave == HEAD_NW - 1) {
      panel_inv(jb + 1);
    } else {
      const int nx = jb;
      const int nt = nb - jb - 2;
      const int cnt = nt * (nt + 1) / 2;
      for (int task = wave - npw; task < nx + cnt; task += HEAD_NW - 1 - npw) {
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
    HEAD_Q(2);
    lds_barrier();
    HEAD_Q(3);
  }
  if (timing && tid == 0) { v.scal[7] = (double)tq[0]; v.scal[8] = (double)tq[1]; v.scal[9] = (double)tq[2]; v.scal[10] = (double)tq[3]; }
  HEAD_STAMP(3);     // remaining block columns (with the inverse of all but the last block row underneath)
  // ---- the last block row of X, one block per wave
  panel_commit(nb - 1);
  if (wave < nb - 1) xblock(nb - 1, wave);
  lds_barrier();
  // ---- log det over the real (unpadded) part; the factor itself only for the white=True adjoint (gp_w1 reads Lu from Kp)
  {
    double s = 0.0;
    for (int i = tid; i < M; i += HEAD_THREADS) s += 2.0 * log(W[i * ld + i]);
    s = sum_wave(s);
    if (lane == 0) red[wave] = s;
  }
  if (white) {
    const int ccol = tid & (HEAD_MAX_N - 1), rgrp = tid >> 7;
    if (ccol < n) {
      gptr Kp = (gptr)v.Kp;
      for (int i = rgrp; i < n; i += HEAD_THREADS / HEAD_MAX_N) Kp[i * n + ccol] = (ccol > i) ? 0.0 : W[i * ld + ccol];
    }
  }
  __syncthreads();
  if (tid == 0) {
    double ldv = 0.0;
    for (int w = 0; w < HEAD_NW; ++w) ldv += red[w];
    v.scal[0] = ldv;
    v.scal[1] = (double)s_info;
  }
  HEAD_STAMP(4);     // log det, last block row of the inverse
}

// grid (1 + nprep + hr.nblk + hg.nblk, L), HEAD_THREADS threads, head_lds_bytes(max Mp) of dynamic LDS
__global__ __launch_bounds__(HEAD_THREADS) void k_head(const double* __restrict__ theta, const LayerDev* __restrict__ layers,
                                                       double* __restrict__ lik_const, int64_t off_lik, int lik_gauss, double jitter,
                                                       int nprep, int keep_kuu, int white, int timing, const HeadRand hr, const HeadGather hg) {
  extern __shared__ __attribute__((aligned(16))) double head_dyn[];
  const int bx = (int)blockIdx.x, l = (int)blockIdx.y;
  if (bx > nprep + hr.nblk) {         // minibatch rows (Minibatch(X), Minibatch(Y) of dgp.py:51-52), block row 0 only
    if (l != 0) return;
    const int64_t ct = hg.dx + hg.dy, nth = (int64_t)hg.nblk * HEAD_THREADS;
    for (int64_t i = (int64_t)(bx - 1 - nprep - hr.nblk) * HEAD_THREADS + threadIdx.x; i < hg.n * ct; i += nth) {
      const int64_t r = i / ct, j = i % ct;
      if (j < hg.dx)
        hg.Xd[r * hg.dx + j] = hg.Xs[hg.idx[r] * hg.dx + j];
      else
        hg.Yd[r * hg.dy + (j - hg.dx)] = hg.Ys[hg.idx[r] * hg.dy + (j - hg.dx)];
    }
    return;
  }
  if (bx > nprep) {                   // N(0,1) draws of layer l (Philox stream l, as dsdgp_randn / k_randn number them)
    if (hr.count[l] > 0)
      randn_body(hr.seed, (uint64_t)l, hr.count[l], hr.out[l], (int64_t)(bx - 1 - nprep) * HEAD_THREADS + threadIdx.x, (int64_t)hr.nblk * HEAD_THREADS);
    return;
  }
  const LayerDev v = layers[l];
  if (bx > 0)
    prep_body(v, theta, lik_const, off_lik, lik_gauss, bx - 1, nprep);
  else if (!keep_kuu)                 // keep_kuu: the factor of the unchanged Ku stays in place (dsdgp_model_track_theta)
    head_factor(v, theta, jitter, white, timing, head_dyn);
}
