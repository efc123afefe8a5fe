// Device kernels of the model path that are not chains or GEMMs: parameter transforms and Ku (k_prep*, k_head), KL (k_kl_*), likelihoods,
// random draws, reparameterisation, upstream adjoints, split-K reduction, gradient assembly (k_asm_*), the fused tail (value, hyper-parameter
// gradients, Adam).  Part of the model translation unit.
#pragma once
// ------------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double softplus_d(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }
// kernel variance from its free variable: [UPSTREAM] transforms.positive, or the value itself for a Parameter without transform
// (dsdgp_layer_desc::kvar_identity — tests/test_dgp.py:79-85 builds a Matern52 with variance 1e-24 that way)
__device__ __forceinline__ double kvar_of(const LayerDev& v, const double* __restrict__ theta) {
  const double rv = theta[v.off_kvar];
  return v.kvar_identity ? rv : softplus_d(rv) + SOFTPLUS_LOWER;
}

// parameter transforms + padding (LowerTriangular / positive transforms of layers.py:150 and [UPSTREAM] kernels)
// WAVE_TILES: k_prep_kuu (256 threads, large models); the head launch (512 threads, M <= 128, its LDS is the factorisation's) takes
// the one-tile-per-workgroup form with the small buffer
template <bool WAVE_TILES>
__device__ void prep_body(const LayerDev& v, const double* __restrict__ theta, double* __restrict__ lik_const, int64_t off_lik,
                          int lik_gauss, int bx, int nprep) {
  const int tid0 = bx * blockDim.x + threadIdx.x, nth = nprep * blockDim.x;
  if (tid0 == 0) {
    const double rv = theta[v.off_kvar];
    const double var = kvar_of(v, theta);
    double wv = 0.0, dwv = 0.0;
    if (v.has_white) {
      const double rw = theta[v.off_wvar];
      wv = softplus_d(rw) + SOFTPLUS_LOWER;
      dwv = sigmoid_d(rw);
    }
    v.hyp[HYP_VAR] = var; v.hyp[HYP_WVAR] = wv; v.hyp[HYP_KDIAG] = var + wv;
    v.hyp[HYP_DVAR] = v.kvar_identity ? 1.0 : sigmoid_d(rv); v.hyp[HYP_DWVAR] = dwv;
    if (blockIdx.y == 0 && lik_gauss) {   // (grid y = layer)
      const double rl = theta[off_lik];
      lik_const[0] = softplus_d(rl) + SOFTPLUS_LOWER;
      lik_const[1] = sigmoid_d(rl);
    }
  }
  for (int j = tid0; j < v.D_in; j += nth) {
    const double rl = theta[v.off_kls + (v.ard ? j : 0)];
    v.hyp[HYP_ILS + j] = 1.0 / (softplus_d(rl) + SOFTPLUS_LOWER);
    v.hyp[HYP_ILS + v.D_in + j] = sigmoid_d(rl);
  }
  const int Mp = v.Mp, M = v.M;
  if (v.D_in > WIDE_DIN)
    for (int idx = tid0; idx < Mp * v.DinP16; idx += nth) {
      const int i = idx / v.DinP16, q = idx % v.DinP16;
      v.Zp1[idx] = (i < M) ? (q < v.D_in ? theta[v.off_Z + (int64_t)i * v.D_in + q] : (q == v.D_in ? 1.0 : 0.0)) : 0.0;
    }
  for (int idx = tid0; idx < Mp * v.D_in; idx += nth) {
    const double z = (idx / v.D_in < M) ? theta[v.off_Z + idx] : 0.0;
    const double rl = theta[v.off_kls + (v.ard ? idx % v.D_in : 0)];
    v.Zp[idx] = z;
    v.Zs[idx] = z / (softplus_d(rl) + SOFTPLUS_LOWER);
  }
  if (v.need_tpt) {
    // padded factor and its transpose, 16 x 16 tiles through LDS: both stores run along rows (the element-wise form stored the
    // transpose with stride Mp — 8 M scattered 8-byte stores per layer at M = 1024, D_out = 8: most of this launch's 195 us).
    // ONE TILE PER WAVE (lane = row lane / 4, four columns): the transposition is wave-private, no workgroup barrier, and the
    // loads of the next tile are in flight while this one is stored.
    if constexpr (WAVE_TILES) {
    __shared__ double tt[4][16][17];
    const int nt = Mp / 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tr = lane >> 2, tc = (lane & 3) * 4;
    // only the tiles on or below the diagonal: above it Tp is zero and so is the mirror tile of TpT — constant since the workspace was
    // zeroed at model creation (nothing else writes these arrays), so a third of this pass's 3 x 8 D_out Mp^2 bytes is never moved
    const int ntl = nt * (nt + 1) / 2;
    const int64_t ntile = (int64_t)v.D_out * ntl;
    for (int64_t tile = (int64_t)bx * 4 + wave; tile < ntile; tile += (int64_t)nprep * 4) {
      const int d = (int)(tile / ntl), rem = (int)(tile % ntl);
      int ti = (int)((sqrt(8.0 * rem + 1.0) - 1.0) * 0.5);
      while ((ti + 1) * (ti + 2) / 2 <= rem) ++ti;
      while (ti * (ti + 1) / 2 > rem) --ti;
      const int i0 = ti * 16, j0 = (rem - ti * (ti + 1) / 2) * 16;
      const int i = i0 + tr;
      double t4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + tc + u;
        const bool in = i < M && j <= i;
        t4[u] = in ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v.Tp[((int64_t)d * Mp + i) * Mp + j0 + tc + u] = t4[u];
        tt[wave][tr][tc + u] = t4[u];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's LDS stores have landed (wave-private tile: no barrier)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 4; ++u) v.TpT[((int64_t)d * Mp + j0 + tr) * Mp + i0 + tc + u] = tt[wave][tc + u][tr];
      __builtin_amdgcn_wave_barrier();
    }
    } else {
    __shared__ double tt[16][17];
    const int nt = Mp / 16, ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int ntl = nt * (nt + 1) / 2;      // (tiles on or below the diagonal only: see above)
    for (int tile = bx; tile < v.D_out * ntl; tile += nprep) {
      const int d = tile / ntl, rem = tile % ntl;
      int tb = 0;
      while ((tb + 1) * (tb + 2) / 2 <= rem) ++tb;
      const int i0 = tb * 16, j0 = (rem - tb * (tb + 1) / 2) * 16;
      const int i = i0 + ti, j = j0 + tj;
      const bool on = threadIdx.x < 256;          // (the head launch runs this body with 512 threads per block)
      const double t = (on && i < M && j <= i) ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
      if (on) v.Tp[((int64_t)d * Mp + i) * Mp + j] = t;
      __syncthreads();
      if (on) tt[ti][tj] = t;
      __syncthreads();
      if (on) v.TpT[((int64_t)d * Mp + j0 + ti) * Mp + i0 + tj] = tt[tj][ti];
    }
    }
  } else {
    for (int idx = tid0; idx < v.D_out * Mp * Mp; idx += nth) {
      const int d = idx / (Mp * Mp), rem = idx % (Mp * Mp), i = rem / Mp, j = rem % Mp;
      v.Tp[idx] = (i < M && j <= i) ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
    }
  }
  for (int idx = tid0; idx < Mp * v.D_out; idx += nth)
    v.qmu[idx] = (idx / v.D_out < M) ? theta[v.off_q_mu + idx] : 0.0;
  for (int idx = tid0; idx < Mp * v.DP4; idx += nth) {
    const int i = idx / v.DP4, d = idx % v.DP4;
    v.qmu4[idx] = (i < M && d < v.D_out) ? theta[v.off_q_mu + (int64_t)i * v.D_out + d] : 0.0;
  }
}

// Ku = K(Z,Z) + (white + jitter) I   (layers.py:171), identity on the padding
__device__ void kuu_body(const LayerDev& v, const double* __restrict__ theta, double jitter, int bx, int nbx) {
  // 16 x 16 output tile per workgroup pass; the two 16-row panels of Z are staged through LDS in 32-column chunks so that
  // wide inputs (784-d MNIST layer) read Z coalesced.  Also stores the scaled squared distances for the adjoint (k_asm_kbar).
  // Reads the raw parameters (not k_prep's outputs): both roles run in ONE launch.
  __shared__ double Zi[16][33], Zj[16][33], ils_s[32];
  const int Mp = v.Mp, nt = Mp / 16, Din = v.D_in, M = v.M;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const double var = kvar_of(v, theta);
  const double wvar = v.has_white ? softplus_d(theta[v.off_wvar]) + SOFTPLUS_LOWER : 0.0;
  for (int tile = bx; tile < nt * nt; tile += nbx) {
    const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
    double r2 = 0.0;
    for (int q0 = 0; q0 < Din; q0 += 32) {
      __syncthreads();
      if (tid < 32 && q0 + tid < Din) ils_s[tid] = 1.0 / (softplus_d(theta[v.off_kls + (v.ard ? q0 + tid : 0)]) + SOFTPLUS_LOWER);
      for (int e = tid; e < 512; e += 256) {
        const int r = e >> 5, c = e & 31;
        const bool ok = q0 + c < Din;
        Zi[r][c] = (ok && i0 + r < M) ? theta[v.off_Z + (int64_t)(i0 + r) * Din + q0 + c] : 0.0;
        Zj[r][c] = (ok && j0 + r < M) ? theta[v.off_Z + (int64_t)(j0 + r) * Din + q0 + c] : 0.0;
      }
      __syncthreads();
      const int qn = min(32, Din - q0);
      for (int c = 0; c < qn; ++c) {
        const double df = (Zi[ti][c] - Zj[tj][c]) * ils_s[c];
        r2 = fma(df, df, r2);
      }
    }
    const int i = i0 + ti, j = j0 + tj;
    double k = (i == j) ? 1.0 : 0.0;
    if (i < M && j < M) {
      k = kern_val_rt(v.kern_kind, r2, var);
      if (i == j) k += wvar + jitter;
    }
    v.Kp[(int64_t)i * Mp + j] = k;
    v.R2[(int64_t)i * Mp + j] = r2;
  }
}

// ONE launch for the parameter transforms / padding (first PREP_BLOCKS block columns) and Ku (the rest), grid (x, L)
__global__ __launch_bounds__(256) void k_prep_kuu(const double* __restrict__ theta, const LayerDev* __restrict__ layers,
                                                  double* __restrict__ lik_const, int64_t off_lik, int lik_gauss, double jitter,
                                                  int nprep, int keep_kuu) {
  const LayerDev v = layers[blockIdx.y];
  if ((int)blockIdx.x < nprep)
    prep_body<true>(v, theta, lik_const, off_lik, lik_gauss, blockIdx.x, nprep);
  else if (!keep_kuu)                 // keep_kuu: the factor of the unchanged Ku stays in place (dsdgp_model_track_theta)
    kuu_body(v, theta, jitter, blockIdx.x - nprep, gridDim.x - nprep);
}

// Philox4x32-10 + Box–Muller: replaces tf.random_normal (layers.py:101-102)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
// pairs t0, t0 + nth, ... of the stream (seed, stream): out[2 i], out[2 i + 1] from counter i
__device__ __forceinline__ void randn_body(uint64_t seed, uint64_t stream, int64_t count, double* __restrict__ out, int64_t t0, int64_t nth) {
  const int64_t npairs = (count + 1) / 2;
  for (int64_t i = t0; i < npairs; i += nth) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    const uint64_t a = ((uint64_t)c[1] << 32) | c[0], b = ((uint64_t)c[3] << 32) | c[2];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    out[2 * i] = rad * cs;
    if (2 * i + 1 < count) out[2 * i + 1] = rad * sn;
  }
}
// fresh N(0,1) draws of the inner layers generated inside the head launch (they depend on nothing)
struct HeadRand {
  double* out[DSDGP_MAX_LAYERS];
  int64_t count[DSDGP_MAX_LAYERS];      // 0: this layer takes no draw from here
  uint64_t seed;
  int32_t nblk;                         // block columns of the launch that generate draws
};

// the minibatch rows gathered inside the head launch too (dsdgp_model_train_step_minibatch): they depend on nothing but the indices
struct HeadGather {
  const double *Xs, *Ys;       // whole data (rows x dx / rows x dy)
  const int64_t* idx;          // row indices of this minibatch (already offset)
  double *Xd, *Yd;             // (n x dx), (n x dy)
  int64_t n;
  int32_t dx, dy, nblk;
};

#include "head_impl.hpp"

__device__ double block_sum_256(double x, double* sh) {
  x = sum_wave(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// SVGP_Layer.KL (layers.py:221-246); V = Lu^-1 q_sqrt_d and nL = Lu^-1 q_mu come from the grouped GEMM.
// grid (NPART, L): deterministic two-stage reduction.
__global__ __launch_bounds__(256) void k_kl_part(const LayerDev* __restrict__ layers) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  double acc = 0.0;
  for (int64_t idx = t0; idx < (int64_t)v.D_out * M; idx += nth) {
    const int d = (int)(idx / M), i = (int)(idx % M);
    const double t = v.Tp[((int64_t)d * Mp + i) * Mp + i];
    acc -= 0.5 * log(t * t);                                            // layers.py:235
  }
  if (!v.white) {
    {   // 1/2 |V|_F^2: 16-byte loads, four independent partial sums (a lone dependent chain ran at ~80 GB/s at M = 1024)
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2* V2 = reinterpret_cast<const d2*>(v.V);
      const int64_t n2 = (int64_t)v.D_out * Mp * Mp / 2;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int64_t idx = t0;
      for (; idx + 3 * nth < n2; idx += 4 * nth) {
        const d2 x0 = V2[idx], x1 = V2[idx + nth], x2 = V2[idx + 2 * nth], x3 = V2[idx + 3 * nth];
        a0 = fma(x0[0], x0[0], fma(x0[1], x0[1], a0));
        a1 = fma(x1[0], x1[0], fma(x1[1], x1[1], a1));
        a2 = fma(x2[0], x2[0], fma(x2[1], x2[1], a2));
        a3 = fma(x3[0], x3[0], fma(x3[1], x3[1], a3));
      }
      for (; idx < n2; idx += nth) {
        const d2 x0 = V2[idx];
        a0 = fma(x0[0], x0[0], fma(x0[1], x0[1], a0));
      }
      acc += 0.5 * ((a0 + a1) + (a2 + a3));                                                                           // :239
    }
    for (int64_t idx = t0; idx < (int64_t)Mp * v.DP4; idx += nth) acc = fma(0.5 * v.nL[idx], v.nL[idx], acc);       // :240-241
  } else {
    for (int64_t idx = t0; idx < (int64_t)v.D_out * Mp * Mp; idx += nth) acc = fma(0.5 * v.Tp[idx], v.Tp[idx], acc);  // :243
    for (int64_t idx = t0; idx < (int64_t)Mp * v.D_out; idx += nth) acc = fma(0.5 * v.qmu[idx], v.qmu[idx], acc);    // :244
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) v.klpart[blockIdx.x] = tot;
}
// KL of every layer from the partial sums of k_kl_part, by ALL 256 threads of the workgroup that forms the ELBO value (k_tail /
// k_finalize): the loads of all layers first, then one fixed-order block reduction per layer.  (A launch of its own for this sat on
// the side stream of every step: k_kl_final, 4.5 us + a launch boundary; one thread walking the partials serially cost the tail 7 us.)
// Returns sum_l KL_l (valid in thread 0) and leaves KL_l in klv[0] of each layer.
__device__ __forceinline__ double layers_kl_value(const LayerDev* __restrict__ layers, int L, double* sh) {
  double tot = 0.0;
  for (int l0 = 0; l0 < L; l0 += 4) {         // four layers' loads in flight, then their reductions
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = 0.0;
      if (l0 + u < L)
        for (int b = threadIdx.x; b < layers[l0 + u].kl_parts; b += 256) x[u] += layers[l0 + u].klpart[b];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (l0 + u < L) {
        const LayerDev& v = layers[l0 + u];
        double kl = block_sum_256(x[u], sh) - 0.5 * v.D_out * v.M;          // layers.py:234
        if (!v.white) kl += 0.5 * v.D_out * v.scal[0];                      // layers.py:238 (sum log diag Lu = logdet/2)
        if (threadIdx.x == 0) v.klv[0] = kl;
        tot += kl;
      }
  }
  return tot;
}
// (dsdgp_model_layer_kl only)
__global__ __launch_bounds__(256) void k_kl_final(const LayerDev* __restrict__ layers, int L) {
  __shared__ double sh[4];
  layers_kl_value(layers, L, sh);
}

// [UPSTREAM] Gaussian.variational_expectations (dgp.py:89-90) and its adjoints w.r.t. the last layer's mean/var.
__global__ __launch_bounds__(256) void k_lik_gauss(const double* __restrict__ mean, const double* __restrict__ var,
                                                   const double* __restrict__ Y, int64_t n, int S, int DY,
                                                   const double* __restrict__ lik_const, double w,
                                                   const double* __restrict__ sw, double* __restrict__ part,
                                                   double* __restrict__ dmean, double* __restrict__ dvar,
                                                   double* __restrict__ MBt, double* __restrict__ VBt, int64_t ldt) {
  // MBt / VBt (DY x ldt, or NULL): the adjoints stored transposed and zero-padded, i.e. already in the form the last layer's
  // backward chain reads (k_adj_prep's job when that layer has one output row per input row)
  __shared__ double sh[4];
  const double s2 = lik_const[0];
  const int64_t total = (int64_t)S * n * DY;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double ve = 0.0, dl = 0.0;
  if (idx < total) {
    const int64_t row = idx / DY;
    const int dd = (int)(idx % DY);
    const double y = Y[(row % n) * DY + dd];
    const double mu = mean[idx], v = var[idx];
    const double q = (y - mu) * (y - mu) + v;
    const double f = sw ? sw[row / n] * S : 1.0;     // quadrature weight relative to the MC mean's 1/S (dgp.py:166)
    ve = f * (-0.91893853320467274178 - 0.5 * log(s2) - 0.5 * q / s2);
    dl = f * (-0.5 / s2 + 0.5 * q / (s2 * s2));
    if (dmean) {
      dmean[idx] = -w * f * (y - mu) / s2;
      dvar[idx] = 0.5 * w * f / s2;
    }
    if (MBt) {
      MBt[(int64_t)dd * ldt + row] = -w * f * (y - mu) / s2;
      VBt[(int64_t)dd * ldt + row] = 0.5 * w * f / s2;
    }
  } else if (MBt && idx < ldt * DY) {      // rows of the 16-row padding
    MBt[(idx % DY) * ldt + idx / DY] = 0.0;
    VBt[(idx % DY) * ldt + idx / DY] = 0.0;
  }
  const double a = block_sum_256(ve, sh);
  const double b = block_sum_256(dl, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = b;
  }
}

// [UPSTREAM] Bernoulli (probit) variational expectations and their adjoints w.r.t. the last layer's mean / var; same outputs as
// k_lik_gauss (the likelihood has no parameter: the second partial is zero)
__global__ __launch_bounds__(256) void k_lik_bern(const double* __restrict__ mean, const double* __restrict__ var,
                                                  const double* __restrict__ Y, int64_t n, int S, int DY, double w,
                                                  const double* __restrict__ sw, double* __restrict__ part,
                                                  double* __restrict__ dmean, double* __restrict__ dvar,
                                                  double* __restrict__ MBt, double* __restrict__ VBt, int64_t ldt) {
  __shared__ double sh[4];
  const int64_t total = (int64_t)S * n * DY;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double ve = 0.0;
  if (idx < total) {
    const int64_t row = idx / DY;
    const int dd = (int)(idx % DY);
    const double y = Y[(row % n) * DY + dd];
    const double f = sw ? sw[row / n] * S : 1.0;
    double dm, dv;
    ve = f * bern_var_exp(mean[idx], var[idx], y, &dm, &dv);
    if (dmean) {
      dmean[idx] = -w * f * dm;
      dvar[idx] = -w * f * dv;
    }
    if (MBt) {
      MBt[(int64_t)dd * ldt + row] = -w * f * dm;
      VBt[(int64_t)dd * ldt + row] = -w * f * dv;
    }
  } else if (MBt && idx < ldt * DY) {      // rows of the 16-row padding
    MBt[(idx % DY) * ldt + idx / DY] = 0.0;
    VBt[(idx % DY) * ldt + idx / DY] = 0.0;
  }
  const double a = block_sum_256(ve, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = 0.0;
  }
}

// [UPSTREAM] Poisson / Exponential / Gamma (exp link), StudentT and Beta variational expectations (common.hpp: lik_var_exp) and their
// adjoints; same outputs as k_lik_gauss — the second partial is the derivative w.r.t. the likelihood's positive parameter
// (lik_const[0]: StudentT.scale, Gamma.shape, Beta.scale), zero for Poisson / Exponential
__global__ __launch_bounds__(256) void k_lik_gen(int kind, const double* __restrict__ lik_const, double aux,
                                                 const double* __restrict__ mean, const double* __restrict__ var,
                                                 const double* __restrict__ Y, int64_t n, int S, int DY, double w,
                                                 const double* __restrict__ sw, double* __restrict__ part,
                                                 double* __restrict__ dmean, double* __restrict__ dvar,
                                                 double* __restrict__ MBt, double* __restrict__ VBt, int64_t ldt) {
  __shared__ double sh[4];
  const double p0 = (kind == DSDGP_LIK_STUDENT_T || kind == DSDGP_LIK_GAMMA || kind == DSDGP_LIK_BETA) ? lik_const[0] : 1.0;
  const int64_t total = (int64_t)S * n * DY;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double ve = 0.0, dl = 0.0;
  if (idx < total) {
    const int64_t row = idx / DY;
    const int dd = (int)(idx % DY);
    const double y = Y[(row % n) * DY + dd];
    const double f = sw ? sw[row / n] * S : 1.0;
    double dm, dv, dp;
    ve = f * lik_var_exp(kind, mean[idx], var[idx], y, p0, aux, &dm, &dv, &dp);
    dl = f * dp;
    if (dmean) {
      dmean[idx] = -w * f * dm;
      dvar[idx] = -w * f * dv;
    }
    if (MBt) {
      MBt[(int64_t)dd * ldt + row] = -w * f * dm;
      VBt[(int64_t)dd * ldt + row] = -w * f * dv;
    }
  } else if (MBt && idx < ldt * DY) {      // rows of the 16-row padding
    MBt[(idx % DY) * ldt + idx / DY] = 0.0;
    VBt[(idx % DY) * ldt + idx / DY] = 0.0;
  }
  const double a = block_sum_256(ve, sh);
  const double b = block_sum_256(dl, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = b;
  }
}

// per-sample quadrature weights applied to per-row values (R = S*n rows) and the (R x K) adjoints (MultiClass + DGP_Quad)
__global__ void k_scale_by_sample(const double* __restrict__ sw, int64_t n, int S, int K, int64_t R, double* __restrict__ ve,
                                  double* __restrict__ dmean, double* __restrict__ dvar) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * K; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / K;
    const double f = sw[row / n] * S;
    if (i % K == 0) ve[row] *= f;
    if (dmean) {
      dmean[i] *= f;
      dvar[i] *= f;
    }
  }
}

// block partial sums of a vector (MultiClass variational expectations), same [blocks][2] layout as k_lik_gauss
__global__ __launch_bounds__(256) void k_partial_sum(const double* __restrict__ x, int64_t count, double* __restrict__ part) {
  __shared__ double sh[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double a = block_sum_256(i < count ? x[i] : 0.0, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = 0.0;
  }
}

// ELBO = data_scale/S * sum ve - kl_weight * sum KL   (dgp.py:92-98)
__global__ __launch_bounds__(256) void k_finalize(const LayerDev* __restrict__ layers, int L, const double* __restrict__ part,
                                                  int nblocks, double w, double kl_weight, const double* __restrict__ lik_const,
                                                  double* __restrict__ grad, int64_t off_lik, int with_grad,
                                                  double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += part[2 * i];
    b += part[2 * i + 1];
  }
  a = block_sum_256(a, sh);
  b = block_sum_256(b, sh);
  const double kl = layers_kl_value(layers, L, sh);
  if (threadIdx.x == 0) {
    double info = 0.0;
    for (int l = 0; l < L; ++l)
      if (layers[l].scal[1] != 0.0 && info == 0.0) info = layers[l].scal[1];
    out[0] = w * a - kl_weight * kl;
    out[1] = w * a;
    out[2] = kl_weight * kl;   // weighted like out[0]: the data-parallel SUM over ranks (kl_weight = 1/world) is then KL itself
    out[3] = info;
    if (with_grad && grad && off_lik >= 0) grad[off_lik] = -w * b * lik_const[1];
  }
}

__global__ void k_randn(uint64_t seed, uint64_t stream, int64_t count, double* __restrict__ out) {
  randn_body(seed, stream, count, out, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

__global__ void k_reparam(const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ z,
                          double jitter, int64_t count, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = mean[i] + z[i] * sqrt(var[i] + jitter);
}

// upstream adjoints of one layer, transposed to the M-major layout of the chain kernels:
//   MB[d][r] = sum_s (dF + dmean)[s,r,d]
//   VB[d][r] = sum_s (dF * z / (2 sqrt(var + jitter)) + dvar)[s,r,d]       (utils.py:41 reverse)
//   XT1      = [X^T ; 1]  (for dl/dZ = GW [X | 1])
// input propagation (layers.py:105-110): next-layer input = [ X[:, :prop] | samples ]; the `rep` output rows of one input
// row (layer 0 is evaluated once per data row) read the same X row
__global__ void k_concat_prop(const double* __restrict__ Xin, int64_t Rin, int D_in, int prop, const double* __restrict__ F,
                              int D_out, int64_t R, double* __restrict__ out) {
  const int W = prop + D_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * W; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t orow = i / W;
    const int j = (int)(i % W);
    out[i] = (j < prop) ? Xin[(orow % Rin) * D_in + j] : F[orow * D_out + (j - prop)];
  }
}

__global__ void k_adj_prep(const double* __restrict__ dF, const double* __restrict__ dmean, const double* __restrict__ dvar,
                           const double* __restrict__ z, int64_t zs_s, int64_t zs_n, int64_t zs_d, int64_t n_inner,
                           const double* __restrict__ var, const double* __restrict__ X, int64_t Rin, int rep, int D_in,
                           int D_out, int DP16, int DinP16, double jitter, int64_t ld, double* __restrict__ MB,
                           double* __restrict__ VB, double* __restrict__ XT1, int ldf, int offf) {
  // dF: adjoint of the next layer's input, (rows x ldf) with this layer's samples at column offset offf (input propagation)
  // grid: x over rows, y over the d / j index (max(DP16, DinP16) slices)
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ld) return;
  const bool ok = r < Rin;
  const int d = blockIdx.y;
  if (d < DP16) {
    double mb = 0.0, vb = 0.0;
    if (ok && d < D_out) {
      if (dF) {
        // all `rep` samples of layer 0 share var (the S input copies are identical): hoist the rsqrt, keep 4 loads in flight
        const double hv = 0.5 * rsqrt(var[r * D_out + d] + jitter);
        double m4[4] = {0, 0, 0, 0}, v4[4] = {0, 0, 0, 0};
        int s = 0;
        for (; s + 4 <= rep; s += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t orow = (int64_t)(s + u) * Rin + r;
            const double f = dF[orow * ldf + offf + d];
            m4[u] += f;
            v4[u] = fma(f, z[(orow / n_inner) * zs_s + (orow % n_inner) * zs_n + d * zs_d], v4[u]);
          }
        }
        for (; s < rep; ++s) {
          const int64_t orow = (int64_t)s * Rin + r;
          const double f = dF[orow * ldf + offf + d];
          m4[0] += f;
          v4[0] = fma(f, z[(orow / n_inner) * zs_s + (orow % n_inner) * zs_n + d * zs_d], v4[0]);
        }
        mb = (m4[0] + m4[1]) + (m4[2] + m4[3]);
        vb = ((v4[0] + v4[1]) + (v4[2] + v4[3])) * hv;
      }
      if (dmean) {
        for (int s = 0; s < rep; ++s) {
          const int64_t o = ((int64_t)s * Rin + r) * D_out + d;
          mb += dmean[o];
          vb += dvar[o];
        }
      }
    }
    MB[(int64_t)d * ld + r] = mb;
    VB[(int64_t)d * ld + r] = vb;
  }
  if (d < DinP16) {
    double v = 0.0;
    if (ok) v = (d < D_in) ? X[r * D_in + d] : (d == D_in ? 1.0 : 0.0);
    XT1[(int64_t)d * ld + r] = v;
  }
}

__global__ __launch_bounds__(256) void k_reduce_grouped(const RedJob* __restrict__ jobs, int njobs, int blk0) {
  // blk0: first block of this launch in the numbering of the whole job list (per-layer launches of a sub-range)
  __shared__ double sh[4];
  const int bx = (int)blockIdx.x + blk0;
  int jb = 0;
  DS_FIND_SEGMENT(jb, jobs, njobs, blk_start, bx);
  const RedJob J = jobs[jb];
  if (J.wide) {
    const int64_t i = bx - J.blk_start;     // one workgroup per output element
    double s = 0.0;
    for (int sp = threadIdx.x; sp < J.nsplit; sp += 256) s += J.part[(int64_t)sp * J.pstride + i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) J.out[i] = s;
    return;
  }
  const int rwave = threadIdx.x >> 6;
  if (J.ways == 64) {
    // symmetric result with ONE or TWO splits (large M: the weight-gradient launch has more tiles than the chip has workgroup slots,
    // choose_nsplit returns 1 — configs 4 / 5): nothing to reduce, the job is a copy of the lower 64 x 64 tiles plus their mirror.
    // One tile per workgroup, 16-byte loads along rows, the tile parked in LDS (odd stride), both stores along rows.  The 16 x 16 form
    // below moved 2 KB per workgroup with three of its four waves adding zeros: 226 us per step at config 4 (1.3 TB/s).
    // Same sums as that form (p0, or p0 + p1): bit-identical.
    extern __shared__ __attribute__((aligned(16))) double red_dyn[];      // 64 x 65 doubles, only launches whose plan has such jobs carry it
    double (*t64)[65] = reinterpret_cast<double (*)[65]>(red_dyn);
    typedef double d2 __attribute__((ext_vector_type(2)));
    int t = bx - J.blk_start, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
    const double* __restrict__ src = J.part + (int64_t)(64 * ti + r) * J.in_ld + 64 * tj + c0;
    d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const d2*>(src + 2 * u);
    if (J.nsplit == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += *reinterpret_cast<const d2*>(src + J.pstride + 2 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      t64[r][c0 + 2 * u] = v[u][0];
      t64[r][c0 + 2 * u + 1] = v[u][1];
    }
    __syncthreads();
    double* __restrict__ o = J.out + (int64_t)(64 * ti + r) * J.out_ld + 64 * tj + c0;
    const bool diag = ti == tj;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      d2 w = v[u];
      // diagonal tile: its 16 x 16 blocks above the block diagonal were not computed — they mirror the blocks below
      if (diag && ((c0 + 2 * u) >> 4) > (r >> 4)) w = (d2){t64[c0 + 2 * u][r], t64[c0 + 2 * u + 1][r]};
      *reinterpret_cast<d2*>(o + 2 * u) = w;
    }
    if (!diag) {
      double* __restrict__ om = J.out + (int64_t)(64 * tj + r) * J.out_ld + 64 * ti + c0;
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<d2*>(om + 2 * u) = (d2){t64[c0 + 2 * u][r], t64[c0 + 2 * u + 1][r]};
    }
    return;
  }
  if (J.ways == 16) {
    // symmetric result, one 16 x 16 tile on or below the diagonal per workgroup: each wave sums a quarter of the splits for the whole
    // tile (16-byte loads, four splits in flight per wave), the four partial tiles meet in LDS, and the tile goes out twice —
    // as it is and transposed to its mirror position — BOTH along rows.  (The element-wise form stored the mirror with stride
    // out_ld: 64 cache lines per store instruction for ~47 % of the outputs.)
    __shared__ double tl[4][16][17];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, rr = lane >> 3, cp = (lane & 7) * 2;      // lane = (row rr + 8 e, column pair cp): 16-byte loads
    int t = bx - J.blk_start, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const double* __restrict__ src = J.part + (int64_t)(16 * ti + rr) * J.in_ld + 16 * tj + cp;
    const int64_t estep = (int64_t)8 * J.in_ld;
    d2 s[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u][0] = s[u][1] = (d2){0, 0};
    int sp = rwave;
    // EIGHT splits = sixteen 16-byte loads in flight per lane and pass (round 5; four before): the splits beyond the end are clamped
    // to the last one and masked, so that every pass is one batch of loads — config 2's 23 splits were a batch of four and two more
    // round trips one after the other, on the critical path at the end of the step.  Accumulator u still takes the splits
    // rwave + 4 (u + 4 k): the sums are the same numbers in the same order.
    for (; sp < J.nsplit; sp += 32) {
      d2 x[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int spu = sp + 4 * u;
        const double* __restrict__ p = src + (int64_t)(spu < J.nsplit ? spu : J.nsplit - 1) * J.pstride;
        x[u][0] = *reinterpret_cast<const d2*>(p);
        x[u][1] = *reinterpret_cast<const d2*>(p + estep);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (sp + 4 * u < J.nsplit) {
          s[u & 3][0] += x[u][0];
          s[u & 3][1] += x[u][1];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const d2 v2 = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
      tl[rwave][rr + 8 * e][cp] = v2[0];
      tl[rwave][rr + 8 * e][cp + 1] = v2[1];
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const double v = (tl[0][r][cc] + tl[1][r][cc]) + (tl[2][r][cc] + tl[3][r][cc]);
    J.out[(int64_t)(16 * ti + r) * J.out_ld + 16 * tj + cc] = v;
    if (ti != tj) {
      __syncthreads();
      tl[0][r][cc] = v;
      __syncthreads();
      J.out[(int64_t)(16 * tj + r) * J.out_ld + 16 * ti + cc] = tl[0][cc][r];
    }
    return;
  }
  if (J.ways == 8) {
    // the 4-way form with TWO adjacent outputs per lane: 16-byte loads (8-byte lanes moved ~2 TB/s of the ~50 MB of partials that
    // config 2 reduces per step).  Same summation order per output as ways = 4: the result is bit-identical.
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ d2 sh2[4][64];
    const int lane = threadIdx.x & 63;
    const int64_t i0 = (int64_t)(bx - J.blk_start) * 128 + 2 * lane;
    const bool live = i0 < J.count;
    int64_t i = live ? i0 : 0, o = i, ostep = 1;
    if (live && J.out_ld > 0) {
      const int64_t r = i0 / J.out_ld, cc = i0 % J.out_ld;
      i = r * J.in_ld + cc;
      if (J.sym_n > 0) {
        const int64_t ti = r / J.sym_tile, tj = cc / J.sym_tile;
        if (tj > ti) {                       // see the scalar form below: read the lower tile's local element, store transposed
          const int64_t lr = r % J.sym_tile, lc = cc % J.sym_tile;
          i = (tj * J.sym_tile + lr) * J.in_ld + ti * J.sym_tile + lc;
          o = (ti * J.sym_tile + lc) * J.out_ld + tj * J.sym_tile + lr;
          ostep = J.out_ld;
        }
      }
    }
    d2 s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = (d2){0, 0};
    int sp = rwave;
    for (; sp + 28 < J.nsplit; sp += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += *reinterpret_cast<const d2*>(J.part + (int64_t)(sp + u * 4) * J.pstride + i);
    }
    if (sp < J.nsplit) {      // the remainder (up to seven splits) as ONE batch of clamped loads, added to s[0] in the old order — the loop
                              // they used to be walked one round trip after the other (all of config 2's 23 splits: six per wave)
      d2 x[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int spu = sp + 4 * u;
        x[u] = *reinterpret_cast<const d2*>(J.part + (int64_t)(spu < J.nsplit ? spu : J.nsplit - 1) * J.pstride + i);
      }
#pragma unroll
      for (int u = 0; u < 7; ++u)
        if (sp + 4 * u < J.nsplit) s[0] += x[u];
    }
    sh2[rwave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rwave == 0 && live) {
      const d2 t = (sh2[0][lane] + sh2[1][lane]) + (sh2[2][lane] + sh2[3][lane]);
      J.out[o] = t[0];
      J.out[o + ostep] = t[1];
    }
    return;
  }
  const bool four = J.ways == 4;
  const int64_t i0 = four ? (int64_t)(bx - J.blk_start) * 64 + (threadIdx.x & 63) : (int64_t)(bx - J.blk_start) * 256 + threadIdx.x;
  if (!four && i0 >= J.count) return;
  const bool live = i0 < J.count;
  int64_t i = live ? i0 : 0, o = i;
  if (live)
  if (J.out_ld > 0) {
    const int64_t r = i0 / J.out_ld, cc = i0 % J.out_ld;
    i = r * J.in_ld + cc;
    if (J.sym_n > 0) {
      // tiles above the diagonal were not computed: the thread that would own element (lr, lc) of upper tile (ti, tj) sums the
      // SAME local element of the lower tile (tj, ti) — coalesced partial reads — and stores it at the transposed position
      const int64_t ti = r / J.sym_tile, tj = cc / J.sym_tile;
      if (tj > ti) {
        const int64_t lr = r % J.sym_tile, lc = cc % J.sym_tile;
        i = (tj * J.sym_tile + lr) * J.in_ld + ti * J.sym_tile + lc;
        o = (ti * J.sym_tile + lc) * J.out_ld + tj * J.sym_tile + lr;
      }
    }
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int st = four ? 4 : 1;
  int sp = four ? rwave : 0;
  for (; sp + 7 * st < J.nsplit; sp += 8 * st) {     // eight independent loads in flight; fixed order -> deterministic
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += J.part[(int64_t)(sp + u * st) * J.pstride + i];
  }
  if (sp < J.nsplit) {        // (remainder as one batch of clamped loads, same order of additions: see ways == 8)
    double x[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int spu = sp + u * st;
      x[u] = J.part[(int64_t)(spu < J.nsplit ? spu : J.nsplit - 1) * J.pstride + i];
    }
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (sp + u * st < J.nsplit) s[0] += x[u];
  }
  const double tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (!four) {
    J.out[o] = tot;
    return;
  }
  __shared__ double sh4[4][64];
  sh4[rwave][threadIdx.x & 63] = tot;
  __syncthreads();
  if (rwave == 0 && live) J.out[o] = (sh4[0][threadIdx.x] + sh4[1][threadIdx.x]) + (sh4[2][threadIdx.x] + sh4[3][threadIdx.x]);
}

// dl/dKu = -sym(G) + kl_w (D/2 Ku^-1 - 1/2 sum_d U_d U_d^T - 1/2 n n^T),  U_d = Ku^-1 q_sqrt_d, n = Ku^-1 q_mu
// then wm = Kbar ∘ dk/dr2 and wk = Kbar ∘ k / variance for the Gram adjoint.
// FOLD (D_in <= 32 and one element per thread): the Ku-side hyper-parameter partial sums of k_asm_hyp_part are taken here,
// one row of hyp2part per workgroup, while kbar / wm / wk are still in registers (one launch less on the step's tail).
__global__ __launch_bounds__(256) void k_asm_kbar(const LayerDev* __restrict__ layers, double kl_w) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M, Din = v.D_in;
  const double* G = v.bigred;
  const bool fold = v.hyp_parts > 0;
  double a_sum = 0.0, tr_sum = 0.0, wm_keep = 0.0;
  int i_keep = 0, j_keep = 0;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    double kb = 0.0, wm = 0.0, wk = 0.0;
    if (i < M && j < M) {
      double nn = 0.0, uu = 0.0;
      if (!v.white) {
        // eight outputs of loads in flight (clamped, masked; the sums in the old order): the rolled loop walked D_out dependent round
        // trips per element — 30 at config 4, where this launch moved 147 MB in 183 us
        const int64_t MMk = (int64_t)Mp * Mp;
        for (int d0 = 0; d0 < v.D_out; d0 += 8) {
          double xu[8], xa[8], xb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int d = d0 + u < v.D_out ? d0 + u : v.D_out - 1;
            xu[u] = v.UU[d * MMk + idx];
            xa[u] = v.n4[i * v.DP4 + d];
            xb[u] = v.n4[j * v.DP4 + d];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (d0 + u < v.D_out) {
              nn += xa[u] * xb[u];
              uu += xu[u];
            }
        }
      }
      if (v.white) {
        kb = 0.5 * (v.wX[i * Mp + j] + v.wX[j * Mp + i]);   // KL(white) does not depend on Ku (layers.py:243-244)
      } else {
        double gsym;
        if (v.alg_g) {
          // sym(sum_r e a^T) = sum_d (GS_d + GS_d^T - P_d) + 1/2 (n t^T + t n^T),  t = A mbar^T (thinq)
          double gs = 0.0, nt = 0.0;
          const int64_t MMg = (int64_t)Mp * Mp;
          for (int d0 = 0; d0 < v.D_out; d0 += 4) {      // (four outputs of loads in flight, the sums in the old order)
            double g1[4], g2[4], pb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int64_t o = (int64_t)(d0 + u < v.D_out ? d0 + u : v.D_out - 1) * MMg;
              g1[u] = v.GS[o + idx];
              g2[u] = v.GS[o + j * Mp + i];
              pb[u] = v.bigred[MMg + o + idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (d0 + u < v.D_out) {
                const int d = d0 + u;
                gs += (g1[u] + g2[u]) - pb[u];
                nt += v.n4[i * v.DP4 + d] * v.thinq[j * v.DP16 + d] + v.n4[j * v.DP4 + d] * v.thinq[i * v.DP16 + d];
              }
          }
          gsym = gs + 0.5 * nt;
        } else {
          gsym = 0.5 * (G[i * Mp + j] + G[j * Mp + i]);
        }
        kb = -gsym + kl_w * (0.5 * v.D_out * v.Kinv[idx] - 0.5 * uu - 0.5 * nn);
      }
      const double r2 = v.R2[idx];
      double k, dk;
      if (v.kern_kind == DSDGP_KERN_RBF)
        kern_val_grad<DSDGP_KERN_RBF>(r2, v.hyp[HYP_VAR], k, dk);
      else
        kern_val_grad<DSDGP_KERN_MATERN52>(r2, v.hyp[HYP_VAR], k, dk);
      wm = kb * dk;
      wk = kb * k / v.hyp[HYP_VAR];
      a_sum += wk;
      if (i == j) tr_sum += kb;
      wm_keep = wm; i_keep = i; j_keep = j;
    }
    v.Kbar[idx] = kb;
    v.wm[idx] = wm;
    v.wk[idx] = wk;
  }
  if (!fold) return;
  double* out = v.hyp2part + (int64_t)blockIdx.x * (Din + 2);
  a_sum = block_sum_256(a_sum, sh);
  tr_sum = block_sum_256(tr_sum, sh);
  if (threadIdx.x == 0) {
    out[0] = a_sum;
    out[1] = tr_sum;
  }
  for (int q = 0; q < Din; ++q) {
    const double df = v.Zp[i_keep * Din + q] - v.Zp[j_keep * Din + q];
    const double sq = block_sum_256(wm_keep * df * df, sh);
    if (threadIdx.x == 0) out[2 + q] = sq;
  }
}

// white=True helpers: Lu_bar = -tril(G) ; Phi(H) = tril(H) with halved diagonal (in place)
__global__ void k_white_lbar(const LayerDev* __restrict__ layers) {
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    v.wLbar[idx] = (i < v.M && j <= i) ? -v.bigred[idx] : 0.0;
  }
}
__global__ void k_white_phi(const LayerDev* __restrict__ layers) {
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    const double h = v.wH[idx];
    v.wH[idx] = (j < i) ? h : (j == i ? 0.5 * h : 0.0);
  }
}

// final assembly of d loss / d theta: elementwise part, grid (blocks, L)
__device__ void asm_hyp_final(const LayerDev& v, double* __restrict__ grad);
__global__ __launch_bounds__(256) void k_asm_params(const LayerDev* __restrict__ layers, double* __restrict__ grad, double kl_w) {
  const LayerDev v = layers[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {   // extra block: hyper-parameter gradients of this layer
    asm_hyp_final(v, grad);
    return;
  }
  const int Mp = v.Mp, M = v.M, Din = v.D_in, Dout = v.D_out;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)(gridDim.x - 1) * blockDim.x;
  const double* ils = v.hyp + HYP_ILS;
  // q_sqrt: 2 tril(P_d T_d) + kl_w (Ku^-1 T_d - diag(1/T_ii))
  // A WAVE per row (d, i) and 512 columns: lane l takes the eight columns 8 l .. 8 l + 7 — 16-byte loads of the part on or below the
  // diagonal only, 16-byte stores of the whole row (zeros right of the diagonal).  The element-wise grid-stride loop paid two 64-bit
  // divisions per element and walked its 15 elements per thread one dependent round trip after the other: 148 us at config 4 for
  // 147 MB read + 147 MB written (1.9 TB/s).
  {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int64_t w0 = t0 >> 6, nw = nth >> 6;
    const int nchunk = (M + 511) / 512;
    const double* Uw = v.white ? v.Tp : v.U;
    for (int64_t wi = w0; wi < (int64_t)Dout * M * nchunk; wi += nw) {
      const int row = (int)(wi / nchunk), ch = (int)(wi - (int64_t)row * nchunk);
      const int d = row / M, i = row - d * M;
      const int j0 = 512 * ch + 8 * lane;
      if (j0 >= M) continue;
      gcptr PT = (gcptr)(v.PT + ((int64_t)d * Mp + i) * Mp + j0), U = (gcptr)(Uw + ((int64_t)d * Mp + i) * Mp + j0);
      d2 pt[4], uu[4];
      const bool any = j0 <= i;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pt[u] = (d2){0, 0};
        uu[u] = (d2){0, 0};
        if (any && j0 + 2 * u <= i) {       // (a pair that straddles the diagonal is loaded whole: Mp is even, the row is padded)
          pt[u] = *reinterpret_cast<const d2 __attribute__((address_space(1)))*>(PT + 2 * u);
          uu[u] = *reinterpret_cast<const d2 __attribute__((address_space(1)))*>(U + 2 * u);
        }
      }
      const bool diag = i >= j0 && i < j0 + 8;
      const double tinv = diag ? 1.0 / v.Tp[((int64_t)d * Mp + i) * Mp + i] : 0.0;
      double* gr = grad + v.off_q_sqrt + (int64_t)row * M + j0;
      const bool al = (((uintptr_t)gr) & 15) == 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double g0 = 0.0, g1 = 0.0;
        const int ja = j0 + 2 * u, jb = ja + 1;
        if (ja <= i) g0 = 2.0 * pt[u][0] + kl_w * (uu[u][0] - (ja == i ? tinv : 0.0));
        if (jb <= i) g1 = 2.0 * pt[u][1] + kl_w * (uu[u][1] - (jb == i ? tinv : 0.0));
        if (al && jb < M) {
          *reinterpret_cast<d2*>(gr + 2 * u) = (d2){g0, g1};
        } else {
          if (ja < M) gr[2 * u] = g0;
          if (jb < M) gr[2 * u + 1] = g1;
        }
      }
    }
  }
  // trainable Linear mean function: rows j < D_in of [X;1]^T MB^T are d loss / d A, row D_in is d loss / d b
  if (v.meanAB) {
    if (v.off_mean_A >= 0)
      for (int64_t idx = t0; idx < (int64_t)Din * Dout; idx += nth) grad[v.off_mean_A + idx] = v.meanAB[(idx / Dout) * v.DP16 + idx % Dout];
    if (v.off_mean_b >= 0)
      for (int64_t idx = t0; idx < Dout; idx += nth) grad[v.off_mean_b + idx] = v.meanAB[(int64_t)Din * v.DP16 + idx];
  }
  // q_mu: A mbar + kl_w Ku^-1 q_mu
  for (int64_t idx = t0; idx < (int64_t)M * Dout; idx += nth) {
    const int i = (int)(idx / Dout), d = (int)(idx % Dout);
    grad[v.off_q_mu + idx] = v.thinq[i * v.DP16 + d] + kl_w * (v.white ? v.qmu4[i * v.DP4 + d] : v.n4[i * v.DP4 + d]);
  }
  // Z: through Kuf (GW [X|1]) and through Ku (wm).  Wide inputs: sum_j wm_ij (z_iq - z_jq) = rowsum_i z_iq - (wm Z)_iq with
  // WZ = wm [Z | 1] from the MFMA GEMM; otherwise one wavefront per (i, q), lanes stride over j
  if (Din > WIDE_DIN) {
    for (int64_t idx = t0; idx < (int64_t)M * Din; idx += nth) {
      const int i = (int)(idx / Din), q = (int)(idx % Din);
      const double zi = v.Zp[idx];
      const double s = v.WZ[(int64_t)i * v.DinP16 + Din] * zi - v.WZ[(int64_t)i * v.DinP16 + q];
      const double il2 = ils[q] * ils[q];
      grad[v.off_Z + idx] = 4.0 * il2 * s - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
    }
  } else {
    const int lane = threadIdx.x & 63;
    const int64_t w0 = t0 >> 6, nw = nth >> 6;
    for (int64_t idx = w0; idx < (int64_t)M * Din; idx += nw) {
      const int i = (int)(idx / Din), q = (int)(idx % Din);
      const double zi = v.Zp[i * Din + q];
      double s = 0.0;
      for (int j0 = lane; j0 < M; j0 += 512) {      // (eight steps of loads in flight, the terms in the old order)
        double w[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 64 * u < M ? j0 + 64 * u : M - 1;
          w[u] = v.wm[i * Mp + j];
          z[u] = v.Zp[j * Din + q];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + 64 * u < M) s = fma(w[u], zi - z[u], s);
      }
      s = sum_wave(s);
      if (lane == 0) {
        const double il2 = ils[q] * ils[q];
        grad[v.off_Z + idx] = 4.0 * il2 * s - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
      }
    }
  }
}

// Ku-side hyper-parameter partial sums, grid (NPART, L): part[b] = { sum wk, trace Kbar, sum_q wm (z_i-z_j)_q^2 ... }
__global__ __launch_bounds__(256) void k_asm_hyp_part(const LayerDev* __restrict__ layers) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M, Din = v.D_in;
  if (v.hyp_parts > 0) return;   // folded into k_asm_kbar
  const int t0 = blockIdx.x * 256 + threadIdx.x, nth = NPART * 256;
  double* out = v.hyp2part + (int64_t)blockIdx.x * (Din + 2);
  double a = 0.0, tr = 0.0;
  for (int idx = t0; idx < M * M; idx += nth) {
    const int i = idx / M, j = idx % M;
    a += v.wk[i * Mp + j];
    if (i == j) tr += v.Kbar[i * Mp + i];
  }
  a = block_sum_256(a, sh);
  tr = block_sum_256(tr, sh);
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = tr;
  }
  if (Din > WIDE_DIN) {
    // sum_ij wm_ij (z_iq - z_jq)^2 = 2 sum_i z_iq (rowsum_i z_iq - (wm Z)_iq)   (wm symmetric); rows i = b mod NPART
    for (int q = threadIdx.x; q < Din; q += 256) {
      double s = 0.0;
      for (int i = blockIdx.x; i < M; i += NPART) {
        const double zi = v.Zp[(int64_t)i * Din + q];
        s = fma(2.0 * zi, v.WZ[(int64_t)i * v.DinP16 + Din] * zi - v.WZ[(int64_t)i * v.DinP16 + q], s);
      }
      out[2 + q] = s;
    }
    return;
  }
  for (int q = 0; q < Din; ++q) {
    double s = 0.0;
    for (int idx = t0; idx < M * M; idx += nth) {
      const int i = idx / M, j = idx % M;
      const double df = v.Zp[i * Din + q] - v.Zp[j * Din + q];
      s = fma(v.wm[i * Mp + j], df * df, s);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[2 + q] = s;
  }
}
// kernel hyper-parameter gradients from the partial sums (one workgroup per layer; runs as the LAST block row of
// k_asm_params — its inputs come from kernels that precede that launch)
__device__ void asm_hyp_final(const LayerDev& v, double* __restrict__ grad) {
  __shared__ double sh[4];
  __shared__ double gl_s[64];
  const int Din = v.D_in, parts = abs(v.hyp_parts), stride = Din + 2;
  const double* ils = v.hyp + HYP_ILS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    double a = 0.0, tr = 0.0;
    for (int b0 = threadIdx.x; b0 < parts; b0 += 4 * 256) {      // (loads of four steps in flight, sums in the old order)
      double xa[4], xt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 256 * u < parts ? b0 + 256 * u : parts - 1;
        xa[u] = v.hyp2part[b * stride];
        xt[u] = v.hyp2part[b * stride + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (b0 + 256 * u < parts) {
          a += xa[u];
          tr += xt[u];
        }
    }
    a = block_sum_256(a, sh);
    tr = block_sum_256(tr, sh);
    if (threadIdx.x == 0) {
      grad[v.off_kvar] = (a + v.hyp_red[0] + v.hyp_red[1]) * v.hyp[HYP_DVAR];
      if (v.has_white) grad[v.off_wvar] = (tr + v.hyp_red[1]) * v.hyp[HYP_DWVAR];
    }
  }
  double iso = 0.0;
  if (Din <= 64) {
    // few lengthscales: one wavefront per q, lanes over the partial rows (a serial walk over the rows is latency-bound)
    for (int q = wave; q < Din; q += 4) {
      double s = 0.0;
      // (eight steps of loads in flight, added in the old order: with up to 1024 partial rows the rolled loop was 16 dependent round
      // trips per lengthscale and this ONE block set the duration of the unfused tail's last launch)
      for (int b0 = lane; b0 < parts; b0 += 512) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v.hyp2part[(b0 + 64 * u < parts ? b0 + 64 * u : parts - 1) * stride + 2 + q];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (b0 + 64 * u < parts) s += x[u];
      }
      s = sum_wave(s);
      if (lane == 0) gl_s[q] = -2.0 * ils[q] * ils[q] * ils[q] * s + v.hyp_red[2 + q];
    }
    __syncthreads();
    if (threadIdx.x < Din) {
      const double gl = gl_s[threadIdx.x];
      if (v.ard)
        grad[v.off_kls + threadIdx.x] = gl * v.hyp[HYP_ILS + Din + threadIdx.x];
      else
        iso = gl;
    }
  } else {
    for (int q = threadIdx.x; q < Din; q += 256) {
      double s = 0.0;
      // (eight partial rows of loads in flight, added in row order: the rolled loop was `parts` dependent round trips per lengthscale —
      // this ONE block set the duration of the whole launch at wide inputs: 148 us at config 4)
      for (int b0 = 0; b0 < parts; b0 += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v.hyp2part[(b0 + u < parts ? b0 + u : parts - 1) * stride + 2 + q];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (b0 + u < parts) s += x[u];
      }
      const double gl = -2.0 * ils[q] * ils[q] * ils[q] * s + v.hyp_red[2 + q];
      if (v.ard)
        grad[v.off_kls + q] = gl * v.hyp[HYP_ILS + Din + q];
      else
        iso += gl;
    }
  }
  iso = block_sum_256(iso, sh);
  if (!v.ard && threadIdx.x == 0) grad[v.off_kls] = iso * v.hyp[HYP_ILS + Din];
}
// Adam over entries [0, n) whose mask equals `want` (0: any non-zero mask), grid-stride over PAIRS: every array is read with one
// 16-byte load per pair, all five loads of an iteration in flight before the first use (the element-at-a-time form tested the mask
// first — two dependent round trips per element, one element in flight per thread: 2.6 TB/s on the 8.7 M parameters of config 4).
// All arrays are 16-byte aligned (workspace / caller allocations) and index 2 k is even.
__device__ __forceinline__ void adam_sweep(double* __restrict__ theta, const double* __restrict__ grad, double* __restrict__ m,
                                           double* __restrict__ v, const double* __restrict__ mask, int64_t n, double lr_t, double b1,
                                           double b2, double eps, double want, int64_t first, int64_t nthreads) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int64_t npair = n >> 1;
  // Four pairs per thread and pass: their masks first (one batch of loads), then gradient, moments and parameters of the pairs that have
  // an entry of this sweep (one batch), then the updates.  A pair without one — the structurally zero half of every q_sqrt above the
  // diagonal is masked out: about half of a large model's entries — costs its 16 mask bytes instead of 64 read and 48 written.
  for (int64_t k0 = first; k0 < npair; k0 += 4 * nthreads) {
    d2 mk[4];
    bool act[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t k = k0 + u * nthreads;
      mk[u] = *reinterpret_cast<const d2*>(mask + 2 * (k < npair ? k : first));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t k = k0 + u * nthreads;
      act[u] = k < npair && (want == 0.0 ? (mk[u][0] != 0.0 || mk[u][1] != 0.0) : (mk[u][0] == want || mk[u][1] == want));
    }
    d2 g[4], mm[4], vv[4], th[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t k = k0 + u * nthreads;
      if (act[u]) {
        g[u] = *reinterpret_cast<const d2*>(grad + 2 * k);
        mm[u] = *reinterpret_cast<const d2*>(m + 2 * k);
        vv[u] = *reinterpret_cast<const d2*>(v + 2 * k);
        th[u] = *reinterpret_cast<const d2*>(theta + 2 * k);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!act[u]) continue;
      const int64_t k = k0 + u * nthreads;
      bool on[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        on[e] = want == 0.0 ? mk[u][e] != 0.0 : mk[u][e] == want;
        if (on[e]) {
          const double mi = b1 * mm[u][e] + (1.0 - b1) * g[u][e];
          const double vi = b2 * vv[u][e] + (1.0 - b2) * g[u][e] * g[u][e];
          mm[u][e] = mi;
          vv[u][e] = vi;
          th[u][e] -= lr_t * mi / (sqrt(vi) + eps);
        }
      }
      // an entry that is not this sweep's is NOT written back: in k_tail its owner (a hyper-parameter / likelihood block) updates it concurrently
      if (on[0] && on[1]) {
        *reinterpret_cast<d2*>(m + 2 * k) = mm[u];
        *reinterpret_cast<d2*>(v + 2 * k) = vv[u];
        *reinterpret_cast<d2*>(theta + 2 * k) = th[u];
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (on[e]) {
            m[2 * k + e] = mm[u][e];
            v[2 * k + e] = vv[u][e];
            theta[2 * k + e] = th[u][e];
          }
      }
    }
  }
  if ((n & 1) && first == 0) {
    const int64_t i = n - 1;
    const bool on = want == 0.0 ? mask[i] != 0.0 : mask[i] == want;
    if (on) {
      const double gi = grad[i];
      const double mi = b1 * m[i] + (1.0 - b1) * gi;
      const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      theta[i] -= lr_t * mi / (sqrt(vi) + eps);
    }
  }
}
// A failed Kuu factorisation ([UPSTREAM] tf.cholesky raises inside session.run: the optimiser op never runs) leaves the parameters
// and the Adam state untouched: the asynchronous step reports the pivot at the next synchronising call and the model is still usable.
__device__ __forceinline__ bool chol_failed(const LayerDev* __restrict__ layers, int L) {
  bool bad = false;
  for (int l = 0; l < L; ++l) bad |= layers[l].scal[1] != 0.0;
  return bad;
}
__global__ void k_adam(double* __restrict__ theta, const double* __restrict__ grad, double* __restrict__ m,
                       double* __restrict__ v, const double* __restrict__ mask, int64_t n, double lr_t, double b1,
                       double b2, double eps, const LayerDev* __restrict__ layers, int L) {
  if (chol_failed(layers, L)) return;
  adam_sweep(theta, grad, m, v, mask, n, lr_t, b1, b2, eps, 0.0, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
             (int64_t)gridDim.x * blockDim.x);
}

// ------------------------------------------------------------------------------------------------------
// Fused tail (non-white models whose layers all have D_in <= WIDE_DIN): k_asm_kbar + k_asm_params become ONE pass with a wave per
// inducing row, k_finalize + the hyper-parameter reduction + (single-process training) the Adam update a second small launch.
// Before: reduce 22 us -> P_d T_d 13 -> k_asm_kbar 15 -> k_asm_params 7 -> k_adam 5 (+ k_finalize 5 on the side stream and its join).
// ------------------------------------------------------------------------------------------------------
// Row i of dl/dKu = -sym(G) + kl_w (D/2 Ku^-1 - 1/2 sum_d U_d U_d^T - 1/2 n n^T) (k_asm_kbar's formula) goes to an LDS row as
// wm = Kbar ∘ dk/dr2; from it the Z gradient of row i and this row's partial sums of the kernel hyper-parameter gradients (hyp2part
// row i: sum wk, Kbar_ii, sum_j wm_ij (z_iq - z_jq)^2), then the q_mu / q_sqrt gradient rows.  One WORKGROUP per inducing row: its
// four waves share the row of Kbar, then split the input dimensions and the (output, column) pairs of the q_sqrt rows — a wave per
// row walked ten dependent memory round trips one after the other (30 us).  grid (M_max, layers), 256 threads, mp_max doubles of LDS.
__global__ __launch_bounds__(256) void k_asm_rows(const LayerDev* __restrict__ layers, double* __restrict__ grad, double kl_w, int mp_max, int pre_on) {
  extern __shared__ __attribute__((aligned(16))) double asm_dyn[];
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = (int)blockIdx.x;
  const int Mp = v.Mp, M = v.M, Din = v.D_in, Dout = v.D_out;
  if (i >= M) return;
  lptr wm = (lptr)asm_dyn;
  const int64_t MM = (int64_t)Mp * Mp;
  const double kvar = v.hyp[HYP_VAR];
  // Round 6 — small layers (M <= 128, D_in <= 8, D_out M <= 1024: configs 1 / 2): everything the LATER phases read — the Z columns and
  // thinz of the Z-gradient phase, the q_mu row, the q_sqrt row's P_d T_d / U_d / T_d — is requested here, before the first phase
  // computes anything, from clamped addresses.  The phases then run on registers: the launch was five dependent memory round trips per
  // workgroup (a 12 us launch on the critical tail of the step for a few hundred KB of data), now two.
  const bool pre = pre_on && (M <= 128) && (Din <= 8) && (Dout * M <= 1024);
  double pz[2][2] = {{0, 0}, {0, 0}}, pzi[2] = {0, 0}, ptz[2] = {0, 0}, ptz1 = 0.0, pq_t = 0.0, pq_n = 0.0;
  double p4p[4] = {0, 0, 0, 0}, p4u[4] = {0, 0, 0, 0}, p4t[4] = {0, 0, 0, 0};
  if (pre) {
    gcptr Zp = (gcptr)v.Zp, thinz = (gcptr)v.thinz, thinq = (gcptr)v.thinq, n4 = (gcptr)v.n4, PT = (gcptr)v.PT, U = (gcptr)v.U, Tp = (gcptr)v.Tp;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wave + 4 * u, qc = q < Din ? q : Din - 1;
      pzi[u] = Zp[i * Din + qc];
      ptz[u] = thinz[i * v.DinP16 + qc];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int j = lane + 64 * w;
        pz[u][w] = Zp[(j < M ? j : M - 1) * Din + qc];
      }
    }
    ptz1 = thinz[i * v.DinP16 + Din];
    const int dq = tid < Dout ? tid : Dout - 1;
    pq_t = thinq[i * v.DP16 + dq];
    pq_n = n4[i * v.DP4 + dq];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int e = tid + 256 * u;
      e = e < Dout * M ? e : Dout * M - 1;
      const int d = e / M, j = e - d * M;
      const int64_t p = d * MM + (int64_t)i * Mp + (j <= i ? j : i);
      p4p[u] = PT[p];
      p4u[u] = U[p];
      p4t[u] = Tp[p];
    }
  }
  double a_sum = 0.0, tr = 0.0;
  for (int j = tid; j < M; j += 256) {
    const int64_t idx = (int64_t)i * Mp + j, idt = (int64_t)j * Mp + i;
    double nn = 0.0, uu = 0.0, gsym;
    if (v.alg_g) {
      // sym(sum_r e a^T) = sum_d (GS_d + GS_d^T - P_d) + 1/2 (n t^T + t n^T),  t = A mbar^T (thinq)
      // ONE loop over the outputs, unrolled: every load of four outputs is in flight before the first use (two loops of
      // one output per iteration walked 2 D_out dependent round trips)
      double gs = 0.0, nt = 0.0;
#pragma unroll 4
      for (int d = 0; d < Dout; ++d) {
        const double ni = v.n4[i * v.DP4 + d], nj = v.n4[j * v.DP4 + d];
        const double g1 = v.GS[d * MM + idx], g2 = v.GS[d * MM + idt], pd = v.bigred[MM + d * MM + idx];
        const double tj = v.thinq[j * v.DP16 + d], ti2 = v.thinq[i * v.DP16 + d];
        nn = fma(ni, nj, nn);
        uu += v.UU[d * MM + idx];
        gs += (g1 + g2) - pd;
        nt += ni * tj + nj * ti2;
      }
      gsym = gs + 0.5 * nt;
    } else {
#pragma unroll 4
      for (int d = 0; d < Dout; ++d) {
        nn = fma(v.n4[i * v.DP4 + d], v.n4[j * v.DP4 + d], nn);
        uu += v.UU[d * MM + idx];
      }
      gsym = 0.5 * (v.bigred[idx] + v.bigred[idt]);
    }
    const double kb = -gsym + kl_w * (0.5 * Dout * v.Kinv[idx] - 0.5 * uu - 0.5 * nn);
    double k, dk;
    if (v.kern_kind == DSDGP_KERN_RBF)
      kern_val_grad<DSDGP_KERN_RBF>(v.R2[idx], kvar, k, dk);
    else
      kern_val_grad<DSDGP_KERN_MATERN52>(v.R2[idx], kvar, k, dk);
    wm[j] = kb * dk;
    a_sum += kb * k / kvar;
    if (j == i) tr = kb;
  }
  a_sum = block_sum_256(a_sum, sh);           // (its barriers also publish the wm row)
  tr = block_sum_256(tr, sh);
  double* __restrict__ hp = v.hyp2part + (int64_t)i * (Din + 2);
  if (tid == 0) {
    hp[0] = a_sum;
    hp[1] = tr;
  }
  const double* __restrict__ ils = v.hyp + HYP_ILS;
  if (pre) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wave + 4 * u;
      if (q >= Din) break;
      const double zi = pzi[u];
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int j = lane + 64 * w;
        if (j < M) {
          const double df = zi - pz[u][w], wj = wm[j];
          s1 = fma(wj, df, s1);
          s2 = fma(wj * df, df, s2);
        }
      }
      s1 = sum_wave(s1);
      s2 = sum_wave(s2);
      if (lane == 0) {
        const double il2 = ils[q] * ils[q];
        grad[v.off_Z + (int64_t)i * Din + q] = 4.0 * il2 * s1 - 2.0 * il2 * (ptz[u] - zi * ptz1);
        hp[2 + q] = s2;
      }
    }
    if (tid < Dout) grad[v.off_q_mu + (int64_t)i * Dout + tid] = pq_t + kl_w * pq_n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      if (e < Dout * M) {
        const int d = e / M, j = e - d * M;
        const double gq = 2.0 * p4p[u] + kl_w * (p4u[u] - (i == j ? 1.0 / p4t[u] : 0.0));
        grad[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] = (j <= i) ? gq : 0.0;
      }
    }
  } else {
  for (int q = wave; q < Din; q += 4) {
    const double zi = v.Zp[i * Din + q];
    double s1 = 0.0, s2 = 0.0;
    for (int j = lane; j < M; j += 64) {
      const double df = zi - v.Zp[j * Din + q], w = wm[j];
      s1 = fma(w, df, s1);
      s2 = fma(w * df, df, s2);
    }
    s1 = sum_wave(s1);
    s2 = sum_wave(s2);
    if (lane == 0) {
      const double il2 = ils[q] * ils[q];
      grad[v.off_Z + (int64_t)i * Din + q] = 4.0 * il2 * s1 - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
      hp[2 + q] = s2;
    }
  }
  // q_mu: A mbar + kl_w Ku^-1 q_mu
  for (int d = tid; d < Dout; d += 256) grad[v.off_q_mu + (int64_t)i * Dout + d] = v.thinq[i * v.DP16 + d] + kl_w * v.n4[i * v.DP4 + d];
  // q_sqrt: 2 tril(P_d T_d) + kl_w (Ku^-1 T_d - diag(1/T_ii)); clamped (unconditional) loads so that several are in flight
  // Row i of every output d: thread t takes the columns t + 256 u.  The loads of output d + 1 are requested before the stores of output d
  // (with store and loads in one loop body the stores — which may alias the loads for all the compiler knows — kept every iteration's
  // loads behind the previous iteration's stores: D_out M / 256 dependent round trips per workgroup, 1.4 TB/s at config 5); columns
  // right of the diagonal are stored as the zeros the contract promises without loading anything.
  {
    constexpr int NU = 4;                   // M <= 1024 on this path (fused tail: D_in <= WIDE_DIN, any M; larger M loops)
    for (int j0 = 0; j0 < M; j0 += 256 * NU) {
      double pt[NU], uu[NU], td = 0.0;
      auto request = [&](int d) {
        gcptr PT = (gcptr)(v.PT + d * MM + (int64_t)i * Mp), U = (gcptr)(v.U + d * MM + (int64_t)i * Mp);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int j = j0 + tid + 256 * u, jc = j <= i ? j : i;
          pt[u] = PT[jc];
          uu[u] = U[jc];
        }
        td = ((gcptr)v.Tp)[d * MM + (int64_t)i * Mp + i];
      };
      request(0);
      for (int d = 0; d < Dout; ++d) {
        double gq[NU];
        const double tinv = 1.0 / td;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int j = j0 + tid + 256 * u;
          gq[u] = (j <= i) ? 2.0 * pt[u] + kl_w * (uu[u] - (i == j ? tinv : 0.0)) : 0.0;
        }
        if (d + 1 < Dout) request(d + 1);
        double* gr = grad + v.off_q_sqrt + ((int64_t)d * M + i) * M;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int j = j0 + tid + 256 * u;
          if (j < M) gr[j] = gq[u];
        }
      }
    }
  }
  }
  // trainable Linear mean function: rows j < D_in of [X;1]^T MB^T are d loss / d A, row D_in is d loss / d b
  if (v.meanAB) {
    if (v.off_mean_A >= 0)
      for (int64_t idx = (int64_t)i * 256 + tid; idx < (int64_t)Din * Dout; idx += (int64_t)M * 256)
        grad[v.off_mean_A + idx] = v.meanAB[(idx / Dout) * v.DP16 + idx % Dout];
    if (v.off_mean_b >= 0 && i == 0)
      for (int idx = tid; idx < Dout; idx += 256) grad[v.off_mean_b + idx] = v.meanAB[(int64_t)Din * v.DP16 + idx];
  }
}

struct AdamArgs {
  double* theta; double* m; double* v; const double* mask;
  int64_t n;
  double lr_t, b1, b2, eps;
  int32_t on;
};
__device__ __forceinline__ void adam_one(const AdamArgs& A, int64_t i, double g) {
  const double mi = A.b1 * A.m[i] + (1.0 - A.b1) * g;
  const double vi = A.b2 * A.v[i] + (1.0 - A.b2) * g * g;
  A.m[i] = mi;
  A.v[i] = vi;
  A.theta[i] -= A.lr_t * mi / (sqrt(vi) + A.eps);
}
struct FinArgs {
  const double* part; int nblocks; double w, kl_weight; const double* lik_const; int64_t off_lik; double* out; int L;
  int do_fin;      // 0: no ELBO-value block in this launch (per-layer launches of the bucketed data-parallel tail)
};
// blocks 0 .. La-1: kernel hyper-parameter gradients of layer first + b from the row partials (asm_hyp_final);
// block La: ELBO value + likelihood-variance gradient (k_finalize's job);  blocks > La (only with A.on): Adam on every entry those
// blocks do not own (mask 1), the owners apply it to theirs (mask 2) right after writing the gradient.
__global__ __launch_bounds__(256) void k_tail(const LayerDev* __restrict__ layers_all, int first, int La, double* __restrict__ grad,
                                              const FinArgs F, const AdamArgs A) {
  __shared__ double sh[4];
  const int b = (int)blockIdx.x;
  const bool adam_on = A.on && !chol_failed(layers_all, F.L);
  if (b < La) {
    const LayerDev v = layers_all[first + b];
    asm_hyp_final(v, grad);
    if (adam_on) {
      __syncthreads();      // (the values were written by threads of this block: re-read below by the same threads that wrote them)
      const int Din = v.D_in;
      if (threadIdx.x == 0) {
        if (A.mask[v.off_kvar] != 0.0) adam_one(A, v.off_kvar, grad[v.off_kvar]);
        if (v.has_white && A.mask[v.off_wvar] != 0.0) adam_one(A, v.off_wvar, grad[v.off_wvar]);
        if (!v.ard && A.mask[v.off_kls] != 0.0) adam_one(A, v.off_kls, grad[v.off_kls]);
      }
      if (v.ard && (int)threadIdx.x < Din && A.mask[v.off_kls + threadIdx.x] != 0.0)
        adam_one(A, v.off_kls + threadIdx.x, grad[v.off_kls + threadIdx.x]);
    }
    return;
  }
  const int nf = F.do_fin ? 1 : 0;
  if (b == La && nf) {
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < F.nblocks; i += 256) {
      a += F.part[2 * i];
      c += F.part[2 * i + 1];
    }
    a = block_sum_256(a, sh);
    c = block_sum_256(c, sh);
    const double kl = layers_kl_value(layers_all, F.L, sh);
    if (threadIdx.x == 0) {
      double info = 0.0;
      for (int l = 0; l < F.L; ++l)
        if (layers_all[l].scal[1] != 0.0 && info == 0.0) info = layers_all[l].scal[1];
      F.out[0] = F.w * a - F.kl_weight * kl;
      F.out[1] = F.w * a;
      F.out[2] = F.kl_weight * kl;
      F.out[3] = info;
      if (F.off_lik >= 0) {
        const double g = -F.w * c * F.lik_const[1];
        grad[F.off_lik] = g;
        if (adam_on && A.mask[F.off_lik] != 0.0) adam_one(A, F.off_lik, g);
      }
    }
    return;
  }
  if (!adam_on) return;
  adam_sweep(A.theta, grad, A.m, A.v, A.mask, A.n, A.lr_t, A.b1, A.b2, A.eps, 1.0, (int64_t)(b - La - nf) * 256 + threadIdx.x,
             (int64_t)(gridDim.x - La - nf) * 256);
}
