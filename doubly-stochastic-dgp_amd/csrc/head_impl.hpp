// Head of an ELBO evaluation in ONE launch for models whose layers all have Mp <= 128 and D_in <= 16 (included by model.hip):
//   block column x = 0 of grid (1 + nprep, L):  Ku = K(Z,Z) + (white + jitter) I  (layers.py:171)  ->  Lu = chol(Ku)  (layers.py:172)
//                                               ->  Lu^-1, Lu^-T (the two triangular solves of layers.py:186,188 as products), log det
//   block columns x >= 1                     :  the parameter transforms / paddings of prep_body
// Before round 3 this was k_prep_kuu (8.6 us) + k_potrf_trtri (48.6 us, three workgroups on a 256-CU chip at the head of every step).
//
// The factorisation never leaves the LDS (n x (n + 5) doubles, 136 KB at n = 128): chol_lds.hpp (lds_chol_inverse) — panel factorisation
// by rows-in-lanes, MFMA trailing update with look-ahead, Lu^-1 built underneath and streamed out as Lu^-1 and Lu^-T.
#pragma once
#include "chol_lds.hpp"

#define HEAD_THREADS 512
#define HEAD_NW 8
#define HEAD_MAX_N 128
#define HEAD_MAX_DIN 16

static inline size_t head_lds_bytes(int n) { return chol_lds_bytes(n); }
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
  const double var = kvar_of(v, theta);
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

  lds_chol_inverse<true>(W, Xd, n, ld, Linv, LinvT, (int64_t)n, &s_info, timing ? v.scal : nullptr, tlast);
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
    prep_body<false>(v, theta, lik_const, off_lik, lik_gauss, bx - 1, nprep);
  else if (!keep_kuu)                 // keep_kuu: the factor of the unchanged Ku stays in place (dsdgp_model_track_theta)
    head_factor(v, theta, jitter, white, timing, head_dyn);
}
