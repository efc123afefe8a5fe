// SVGP_Layer hot path (layers.py:178-219 conditional_ND, utils.py:40-41 reparameterize) as register-resident fp64-MFMA
// chains on gfx950.
//
// Orientation: everything is kept "M-major" — a wavefront owns 16 data rows r (MFMA columns) and the full padded
// inducing dimension Mp (MFMA rows).  With  Kuf[:, r] = k_r  the layer is
//     a1 = Lu^{-1} k ,  a = Lu^{-T} a1 ,  c_d = q_sqrt_d^T a ,
//     mean_d = a . q_mu[:,d] + m(x) ,  var_d = kdiag - |a1|^2 + |c_d|^2 ,  F_d = mean_d + z sqrt(var_d + jitter)
// (the "cheaper equivalent form" of SURVEY Appendix A.6 — algebraically identical to the reference's
//  SK = q_sqrt q_sqrt^T - Ku,  B = SK A,  sum(A∘B), but all three products are triangular MFMA products and the
//  D x M x R intermediates A_tiled / B never exist).  The weights (Lu^{-T}, Lu^{-1}, q_sqrt_d) stream from L2 as MFMA
// A-operands with 128-B coalesced reads; the activations never leave registers because the D layout of
// v_mfma_f64_16x16x4_f64 (row = 4*reg + lane/16) is exactly the B-operand layout of the next product's four k-steps.
#include "layer.hpp"
#include <stdlib.h>

// LDS: [zs Mp*Din][xs 4*16*(Din+1)] (padded to 16 B) [two weight-panel buffers of 16 x (Mp+16) doubles]
static inline size_t chain_front_doubles(int Mp, int D_in) { return (size_t)round_up(Mp * D_in + 4 * 16 * (D_in + 1), 2); }
size_t layer_fwd_lds_bytes(int Mp, int D_in) {
  return (chain_front_doubles(Mp, D_in) + 2 * 16 * (size_t)(Mp + 16)) * sizeof(double);
}

typedef double d2 __attribute__((ext_vector_type(2)));

// Weight-panel pipeline shared by the 4 waves of a workgroup.  A panel is 16 consecutive rows (the k-block of an MFMA
// product) of an Mp x Mp row-major weight matrix.  Every thread owns one (row, MPB-column) slot: it loads the slot of
// panel p+1 from L2/HBM into registers before the MFMAs of panel p are issued and stores it to the other LDS buffer
// afterwards — one barrier per panel, global latency hidden behind a whole panel of MFMAs, and each weight element
// crosses the L2->CU path once per workgroup instead of once per wave.  Row stride Mp+16 doubles keeps the four
// 16-lane groups of an A-fragment read on disjoint banks.
template <int MPB>
struct PanelPipe {
  static constexpr int Mp = MPB * 16, LDW = Mp + 16, PANEL = 16 * LDW;
  double* buf;
  int cur;
  int row, col0, blk;
  bool have;
  d2 pre[MPB / 2];
  __device__ __forceinline__ PanelPipe(double* b, int tid) : buf(b), cur(0), have(false) {
    row = tid >> 4;
    col0 = (tid & 15) * MPB;
    blk = col0 >> 4;
  }
  // issue the loads of panel `kb` of W, column blocks [lo, hi]
  __device__ __forceinline__ void prefetch(const double* __restrict__ W, int kb, int lo, int hi) {
    have = (blk >= lo && blk <= hi);
    if (have) {
      const d2* __restrict__ src = reinterpret_cast<const d2*>(W + (int64_t)(16 * kb + row) * Mp + col0);
#pragma unroll
      for (int j = 0; j < MPB / 2; ++j) pre[j] = src[j];
    }
  }
  // publish the prefetched panel and make it current
  __device__ __forceinline__ void commit() {
    if (have) {
      d2* dst = reinterpret_cast<d2*>(buf + (cur ^ 1) * PANEL + row * LDW + col0);
#pragma unroll
      for (int j = 0; j < MPB / 2; ++j) dst[j] = pre[j];
    }
    __syncthreads();
    cur ^= 1;
  }
  // A-operand fragment: W[16 kb + 4 s + g][16 ib + c]
  __device__ __forceinline__ double frag(int s, int ib, int g, int c) const {
    return buf[cur * PANEL + (4 * s + g) * LDW + 16 * ib + c];
  }
};

// scaled squared distances between the wave's 16 rows (LDS xs, lane column c) and all Mp inducing points (LDS zs),
// in D layout: r2[kb][t] <-> m = 16 kb + g + 4 t.
template <int MPB>
__device__ __forceinline__ void sqdist_tile(const double* __restrict__ zs, const double* __restrict__ xs, int Din, int g,
                                            int c, d4 (&r2)[MPB]) {
#pragma unroll
  for (int kb = 0; kb < MPB; ++kb) r2[kb] = (d4){0, 0, 0, 0};
  for (int j = 0; j < Din; ++j) {
    const double xv = xs[c * (Din + 1) + j];
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double df = zs[(16 * kb + g + 4 * t) * Din + j] - xv;
        r2[kb][t] = fma(df, df, r2[kb][t]);
      }
  }
}

template <int MPB, int KIND, bool WHITE>
__global__ __launch_bounds__(256, (MPB <= 8 ? 2 : 1)) void k_layer_fwd(const LayerFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16;
  const int Din = a.D_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  double* zs = smem;
  double* xs = smem + Mp * Din + wave * 16 * (Din + 1);
  PanelPipe<MPB> P(smem + (size_t)((Mp * Din + 4 * 16 * (Din + 1) + 1) / 2 * 2), tid);
  P.prefetch(a.LinvT, 0, 0, MPB - 1);
  const double* ils = a.hyp + HYP_ILS;
  for (int idx = tid; idx < Mp * Din; idx += 256) zs[idx] = a.Zp[idx] * ils[idx % Din];
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  for (int idx = lane; idx < 16 * Din; idx += 64) {
    const int rr = idx / Din, j = idx % Din;
    int64_t row = r0 + rr;
    if (row > a.Rin - 1) row = a.Rin - 1;
    xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
  }
  P.commit();   // barrier: zs/xs and panel 0 of Lu^{-T} visible
  // NOTE: no early exit — every wave takes part in the panel barriers; out-of-range rows are clamped and never stored.
  const int64_t r = r0 + c;
  const bool rvalid = r < a.Rin;
  const int64_t rc = rvalid ? r : a.Rin - 1;
  const double s2 = a.hyp[HYP_VAR];

  // --- Kuf tile (layers.py:184) in registers
  d4 kreg[MPB];
  sqdist_tile<MPB>(zs, xs, Din, g, c, kreg);
#pragma unroll
  for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      kreg[kb][t] = (16 * kb + g + 4 * t < a.M) ? kern_val<KIND>(kreg[kb][t], s2) : 0.0;

  // --- a1 = Lu^{-1} k   (layers.py:186)   A-operand element [i][k] = Linv[i][k] = LinvT[k][i], i-blocks >= k-block
  d4 a1[MPB];
#pragma unroll
  for (int ib = 0; ib < MPB; ++ib) a1[ib] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < MPB; ++kb) {
    if (kb + 1 < MPB)
      P.prefetch(a.LinvT, kb + 1, kb + 1, MPB - 1);
    else
      P.prefetch(WHITE ? a.Tp : a.Linv, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double bv = kreg[kb][s];
#pragma unroll
      for (int ib = kb; ib < MPB; ++ib) a1[ib] = mfma_f64(P.frag(s, ib, g, c), bv, a1[ib]);
    }
    P.commit();
  }
  double s1 = 0.0;
#pragma unroll
  for (int ib = 0; ib < MPB; ++ib)
#pragma unroll
    for (int t = 0; t < 4; ++t) s1 = fma(a1[ib][t], a1[ib][t], s1);
  s1 = sum_groups(s1);

  // --- a = Lu^{-T} a1   (layers.py:188, non-white)   A-operand element [i][k] = Linv[k][i], i-blocks <= k-block
  d4 av[MPB];
  if (WHITE) {
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib) av[ib] = a1[ib];
  } else {
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib) av[ib] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb) {
      if (kb + 1 < MPB)
        P.prefetch(a.Linv, kb + 1, 0, kb + 1);
      else
        P.prefetch(a.Tp, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = a1[kb][s];
#pragma unroll
        for (int ib = 0; ib <= kb; ++ib) av[ib] = mfma_f64(P.frag(s, ib, g, c), bv, av[ib]);
      }
      P.commit();
    }
  }
  if (a.Asave && r0 < a.ldA) {
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib)
#pragma unroll
      for (int t = 0; t < 4; ++t) a.Asave[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r] = rvalid ? av[ib][t] : 0.0;
  }

  const double kdiag = a.hyp[HYP_KDIAG];
  for (int d = 0; d < a.D_out; ++d) {
    // --- c_d = q_sqrt_d^T a ; |c_d|^2   (replaces SK/B of layers.py:195-212)
    d4 cacc[MPB];
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib) cacc[ib] = (d4){0, 0, 0, 0};
    const double* __restrict__ Td = a.Tp + (int64_t)d * Mp * Mp;
    const bool more = d + 1 < a.D_out;
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb) {
      if (kb + 1 < MPB)
        P.prefetch(Td, kb + 1, 0, kb + 1);
      else if (more)
        P.prefetch(Td + (int64_t)Mp * Mp, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = av[kb][s];
#pragma unroll
        for (int ib = 0; ib <= kb; ++ib) cacc[ib] = mfma_f64(P.frag(s, ib, g, c), bv, cacc[ib]);
      }
      if (kb + 1 < MPB || more) P.commit();
    }
    double s2sum = 0.0, mu = 0.0;
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s2sum = fma(cacc[ib][t], cacc[ib][t], s2sum);
        mu = fma(av[ib][t], a.qmu[(16 * ib + g + 4 * t) * a.D_out + d], mu);   // layers.py:190
      }
    s2sum = sum_groups(s2sum);
    mu = sum_groups(mu);
    const double var = kdiag - s1 + s2sum;                                     // layers.py:212-217
    if (a.mean_kind == DSDGP_MEAN_IDENTITY) {                                  // layers.py:219
      mu += a.X[rc * Din + d];
    } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
      double acc = 0.0;
      for (int j = 0; j < Din; ++j) acc = fma(a.X[rc * Din + j], a.mean_A[j * a.D_out + d], acc);
      mu += acc + (a.mean_b ? a.mean_b[d] : 0.0);
    }
    if (rvalid) {
      for (int s = g; s < a.rep; s += 4) {
        const int64_t orow = (int64_t)s * a.Rin + r;
        const int64_t o = orow * a.D_out + d;
        if (a.mean) a.mean[o] = mu;
        if (a.var) a.var[o] = var;
        if (a.F && a.z) {
          const double zv = a.z[(orow / a.n_inner) * a.zs_s + (orow % a.n_inner) * a.zs_n + d * a.zs_d];
          a.F[o] = mu + zv * sqrt(var + a.jitter);                             // utils.py:41 (no clamp)
        }
      }
    }
  }
}

template <int MPB, int KIND>
static int fwd_dispatch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int white) {
  const int Mp = MPB * 16;
  const size_t lds = layer_fwd_lds_bytes(Mp, a.D_in);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_fwd: D_in=%d with M_pad=%d needs %zu B LDS (>160 KiB): large-D_in path not built yet", a.D_in,
                    Mp, lds);
    return DSDGP_ERR_UNSUPPORTED;
  }
  const int nblk = ceil_div(a.Rin, 64);
  ProfScope ps(ctx, "layer_fwd");
  if (white) {
    if (lds > 64 * 1024)
      DS_HIP(hipFuncSetAttribute((const void*)k_layer_fwd<MPB, KIND, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_layer_fwd<MPB, KIND, true>), dim3(nblk), dim3(256), lds, ctx->stream, a);
  } else {
    if (lds > 64 * 1024)
      DS_HIP(hipFuncSetAttribute((const void*)k_layer_fwd<MPB, KIND, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_layer_fwd<MPB, KIND, false>), dim3(nblk), dim3(256), lds, ctx->stream, a);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

int layer_fwd_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white) {
#define FWD_CASE(MPB)                                                                           \
  case MPB * 16:                                                                                \
    return kern_kind == DSDGP_KERN_RBF ? fwd_dispatch<MPB, DSDGP_KERN_RBF>(ctx, a, white)       \
                                       : fwd_dispatch<MPB, DSDGP_KERN_MATERN52>(ctx, a, white);
  switch (Mp) {
    FWD_CASE(2)
    FWD_CASE(4)
    FWD_CASE(8)
    FWD_CASE(16)
    default:
      dsdgp_set_error("layer_fwd: padded inducing count %d not built (supported: 32,64,128,256)", Mp);
      return DSDGP_ERR_UNSUPPORTED;
  }
#undef FWD_CASE
}

// ------------------------------------------------------------------------------------------------------
// Backward chain (non-white).  Per data row, with the upstream adjoints mbar_d = dl/dmean_d, vbar_d = dl/dvar_d,
// g = sum_d vbar_d (SURVEY Appendix C re-derived in terms of Ku^{-1} so that no Cholesky adjoint is needed):
//   abar = sum_d 2 vbar_d S_d a + sum_d q_mu[:,d] mbar_d          (S_d = q_sqrt_d q_sqrt_d^T)
//   b    = Ku^{-1} abar ;  e = b - g a ;  kbar = e - g a          (kbar = dl/dKuf[:, r])
//   dl/dKu(data) = -sym(sum_r e_r a_r^T)      -> k_wgrad(E, A)
//   dl/dS_d      =  sum_r vbar_d a_r a_r^T    -> k_wgrad(A, A, scale = vbar_d)
//   dl/dq_mu     =  sum_r a_r mbar_r^T        -> k_wgrad(A, MB)
//   GW = kbar ∘ dk/dr2 drives dl/dX (here), dl/dZ (k_wgrad(GW, [X|1])) and the lengthscale / variance partials.
// ------------------------------------------------------------------------------------------------------
template <int MPB, int KIND, int OCC, bool WHITE>
__global__ __launch_bounds__(256, OCC) void k_layer_bwd(const LayerBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16;
  const int Din = a.D_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  double* zs = smem;
  double* xs = smem + Mp * Din + wave * 16 * (Din + 1);
  PanelPipe<MPB> P(smem + (size_t)((Mp * Din + 4 * 16 * (Din + 1) + 1) / 2 * 2), tid);
  P.prefetch(a.Sd, 0, 0, MPB - 1);
  const double* ils = a.hyp + HYP_ILS;
  for (int idx = tid; idx < Mp * Din; idx += 256) zs[idx] = a.Zp[idx] * ils[idx % Din];
  const int64_t wg = (int64_t)blockIdx.x * 4 + wave;
  const int64_t r0 = wg * 16;
  for (int idx = lane; idx < 16 * Din; idx += 64) {
    const int rr = idx / Din, j = idx % Din;
    int64_t row = r0 + rr;
    if (row > a.Rin - 1) row = a.Rin - 1;
    xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
  }
  P.commit();
  const bool wactive = r0 < a.ldA;          // waves beyond the padded row count only keep the panel barriers alive
  const int64_t r = wactive ? r0 + c : 0;
  const bool rvalid = wactive && (r < a.Rin);
  const double s2 = a.hyp[HYP_VAR];

  d4 av[MPB], acc[MPB];
#pragma unroll
  for (int ib = 0; ib < MPB; ++ib) {
    acc[ib] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) av[ib][t] = a.Asave[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r];
  }
  double gsum = 0.0;
  for (int d = 0; d < a.D_out; ++d) {
    const double vd = a.VB[(int64_t)d * a.ldA + r];
    gsum += vd;
    const double vd2 = 2.0 * vd;
    const double* __restrict__ Sd = a.Sd + (int64_t)d * Mp * Mp;
    const double* __restrict__ nxt = (d + 1 < a.D_out) ? Sd + (int64_t)Mp * Mp : (WHITE ? a.Linv : a.Kinv);
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb) {
      if (kb + 1 < MPB)
        P.prefetch(Sd, kb + 1, 0, MPB - 1);
      else
        P.prefetch(nxt, 0, 0, (WHITE && d + 1 >= a.D_out) ? 0 : MPB - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = av[kb][s] * vd2;
#pragma unroll
        for (int ib = 0; ib < MPB; ++ib) acc[ib] = mfma_f64(P.frag(s, ib, g, c), bv, acc[ib]);
      }
      P.commit();
    }
  }
  for (int sp = 0; sp < a.DP4 / 4; ++sp) {
    const double bv = a.MB[(int64_t)(4 * sp + g) * a.ldA + r];
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib) acc[ib] = mfma_f64(a.qmu4[(16 * ib + c) * a.DP4 + 4 * sp + g], bv, acc[ib]);
  }
  d4 bb[MPB];
#pragma unroll
  for (int ib = 0; ib < MPB; ++ib) bb[ib] = (d4){0, 0, 0, 0};
  if (WHITE) {
    // white: var_d = kdiag - |a1|^2 + a1^T S_d a1, so a1bar = acc - 2 g a1 and kbar = Lu^{-T} a1bar (i-blocks <= k-block)
#pragma unroll
    for (int ib = 0; ib < MPB; ++ib)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[ib][t] -= 2.0 * gsum * av[ib][t];
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb) {
      if (kb + 1 < MPB) P.prefetch(a.Linv, kb + 1, 0, kb + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = acc[kb][s];
#pragma unroll
        for (int ib = 0; ib <= kb; ++ib) bb[ib] = mfma_f64(P.frag(s, ib, g, c), bv, bb[ib]);
      }
      if (kb + 1 < MPB) P.commit();
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb) {
      if (kb + 1 < MPB) P.prefetch(a.Kinv, kb + 1, 0, MPB - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = acc[kb][s];
#pragma unroll
        for (int ib = 0; ib < MPB; ++ib) bb[ib] = mfma_f64(P.frag(s, ib, g, c), bv, bb[ib]);
      }
      if (kb + 1 < MPB) P.commit();
    }
  }
  if (!wactive) return;   // no barriers below
#pragma unroll
  for (int ib = 0; ib < MPB; ++ib)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // non-white: E = b - g a with dl/dKu = -sym(E A^T), kbar = E - g a ;  white: E = kbar with dl/dLu = -tril(E A1^T)
      const double e = WHITE ? bb[ib][t] : bb[ib][t] - gsum * av[ib][t];
      a.E[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r] = e;
      bb[ib][t] = WHITE ? e : e - gsum * av[ib][t];  // kbar
    }

  // recompute the Kuf tile and its r2-derivative; GW = kbar * dk/dr2
  d4 wv[MPB];
  sqdist_tile<MPB>(zs, xs, Din, g, c, wv);
  double svar = 0.0;
#pragma unroll
  for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double k, dk;
      kern_val_grad<KIND>(wv[kb][t], s2, k, dk);
      const bool ok = rvalid && (16 * kb + g + 4 * t < a.M);
      const double kb_ = bb[kb][t];
      svar += ok ? kb_ * k : 0.0;
      const double w = ok ? kb_ * dk : 0.0;
      wv[kb][t] = w;
      a.GW[(int64_t)(16 * kb + g + 4 * t) * a.ldA + r] = w;
    }
  svar = sum_wave(svar);
  const double gk = sum_wave((rvalid && g == 0) ? gsum : 0.0);
  double* hp = a.hyp_part + wg * (Din + 2);
  if (lane == 0) {
    hp[0] = svar / s2;   // d loss / d variance through Kuf
    hp[1] = gk;          // d loss / d kdiag  (Kdiag = variance (+ white), layers.py:213)
  }
  for (int j = 0; j < Din; ++j) {
    const double xv = xs[c * (Din + 1) + j];
    double sx = 0.0, sl = 0.0;
#pragma unroll
    for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double df = xv - zs[(16 * kb + g + 4 * t) * Din + j];
        const double wdf = wv[kb][t] * df;
        sx += wdf;
        sl = fma(wdf, df, sl);
      }
    sx = sum_groups(sx);
    sl = sum_wave(sl);
    if (lane == 0) hp[2 + j] = -2.0 * ils[j] * sl;   // d r2 / d l_j = -2 (x_j - z_j)^2 / l_j^3
    if (a.dX && rvalid && g == 0) {
      double dx = 2.0 * ils[j] * sx;                  // d r2 / d x_j = 2 (x_j - z_j) / l_j^2
      if (a.mean_kind == DSDGP_MEAN_IDENTITY) {
        dx += a.MB[(int64_t)j * a.ldA + r];
      } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
        for (int d = 0; d < a.D_out; ++d) dx = fma(a.mean_A[j * a.D_out + d], a.MB[(int64_t)d * a.ldA + r], dx);
      }
      a.dX[r * Din + j] = dx;
    }
  }
}

template <int MPB, int KIND, bool WHITE>
static int bwd_dispatch(dsdgp_ctx* ctx, const LayerBwdArgs& a) {
  const int Mp = MPB * 16;
  const size_t lds = layer_fwd_lds_bytes(Mp, a.D_in);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_bwd: D_in=%d too large for the fused path", a.D_in);
    return DSDGP_ERR_UNSUPPORTED;
  }
  const int nblk = ceil_div(a.ldA, 64);
  ProfScope ps(ctx, "layer_bwd");
  static const int occ2 = getenv("DSDGP_BWD_OCC2") ? atoi(getenv("DSDGP_BWD_OCC2")) : 0;
  if (occ2 && MPB <= 8) {
    hipLaunchKernelGGL((k_layer_bwd<MPB, KIND, (MPB <= 8 ? 2 : 1), WHITE>), dim3(nblk), dim3(256), lds, ctx->stream, a);
  } else {
    if (lds > 64 * 1024)
      DS_HIP(hipFuncSetAttribute((const void*)k_layer_bwd<MPB, KIND, 1, WHITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_layer_bwd<MPB, KIND, 1, WHITE>), dim3(nblk), dim3(256), lds, ctx->stream, a);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

int layer_bwd_launch(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white) {
#define BWD_CASE(MPB)                                                                                          \
  case MPB * 16:                                                                                               \
    if (white)                                                                                                 \
      return kern_kind == DSDGP_KERN_RBF ? bwd_dispatch<MPB, DSDGP_KERN_RBF, true>(ctx, a)                     \
                                         : bwd_dispatch<MPB, DSDGP_KERN_MATERN52, true>(ctx, a);               \
    return kern_kind == DSDGP_KERN_RBF ? bwd_dispatch<MPB, DSDGP_KERN_RBF, false>(ctx, a)                      \
                                       : bwd_dispatch<MPB, DSDGP_KERN_MATERN52, false>(ctx, a);
  switch (Mp) {
    BWD_CASE(2)
    BWD_CASE(4)
    BWD_CASE(8)
    BWD_CASE(16)
    default:
      dsdgp_set_error("layer_bwd: padded inducing count %d not built", Mp);
      return DSDGP_ERR_UNSUPPORTED;
  }
#undef BWD_CASE
}

// ------------------------------------------------------------------------------------------------------
// Split-K "weight gradient" products over the data rows: out = P diag(scale) Q^T with P (rowsP x R), Q (rowsQ x R)
// both M-major.  One wavefront per (job, split, tile); operands go straight from L2/HBM to MFMA registers: each lane
// loads 4 consecutive r (32 B) of one row, and the t-th of them is the k-operand of the t-th MFMA, so a 16-row x
// 16-r fragment costs two 16-B loads per lane and no LDS.
// ------------------------------------------------------------------------------------------------------
// K loop of one (split, tile) task; GUARD: only the first njv of the NJ column blocks of Q exist (thin products A MB^T,
// GW [X|1]^T ride in the same launch as the M x M products, their Q has 16..DinP16 rows)
// DIAG: diagonal tile of a symmetric result (P == Q): only the 16x16 blocks on or below the block diagonal are formed
// (10 of 16 MFMAs per k-step at NI = NJ = 4); the reduction mirrors at 16-block granularity.
template <int NI, int NJ, bool GUARD, bool DIAG>
__device__ __forceinline__ void wgrad_loop(gcptr Pp, gcptr Qp, gcptr scale, int64_t ld, int64_t c_lo, int64_t c_hi, int g,
                                           int njv, d4 (&acc)[NI][NJ]) {
  for (int64_t ch = c_lo; ch < c_hi; ++ch) {
    const int64_t rb = ch * 16;
    d4 pa[NI], qb[NJ];
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) pa[ii] = *reinterpret_cast<const d4 __attribute__((address_space(1)))*>(Pp + (int64_t)16 * ii * ld + rb);
    if constexpr (DIAG) {
      // diagonal tile of a symmetric product: Q's rows ARE P's rows — no second load (a third less operand traffic per P_d at Mw = 128;
      // the launch moves ~5.6 TB/s out of L2 / Infinity Cache and did not get faster with deeper prefetch: bandwidth, not latency)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) qb[jj] = pa[jj];
    } else {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (!GUARD || jj < njv) qb[jj] = *reinterpret_cast<const d4 __attribute__((address_space(1)))*>(Qp + (int64_t)16 * jj * ld + rb);
    }
    if (scale) {
      const d4 sc = *reinterpret_cast<const d4 __attribute__((address_space(1)))*>(scale + rb + 4 * g);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (!GUARD || jj < njv) qb[jj] *= sc;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ii = 0; ii < NI; ++ii)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          if ((!GUARD || jj < njv) && (!DIAG || jj <= ii)) acc[ii][jj] = mfma_f64(pa[ii][t], qb[jj][t], acc[ii][jj]);
  }
}

template <int NI, int NJ>
__global__ __launch_bounds__(256) void k_wgrad(const WgradJob* __restrict__ jobs, int njobs, int nsplit, int64_t ld,
                                               int64_t Rp, int total_tasks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x * 4 + wave;
  if (w >= total_tasks) return;
  int jb = 0;
  while (jb + 1 < njobs && w >= jobs[jb + 1].task_start) ++jb;
  const WgradJob J = jobs[jb];
  int local = w - J.task_start;
  int split, tile_i, tile_j, ns_eff = nsplit;
  if (J.sym) {
    // off-diagonal tiles first (nsplit K ranges each), then the diagonal tiles (ns_diag longer ranges: they skip 6 of 16 blocks)
    const int n_off = J.ti * (J.ti - 1) / 2;
    if (local < nsplit * n_off) {
      split = local / n_off;
      local = local % n_off;
      tile_i = 1;
      while (tile_i * (tile_i + 1) / 2 <= local) ++tile_i;
      tile_j = local - tile_i * (tile_i - 1) / 2;
    } else {
      local -= nsplit * n_off;
      split = local / J.ti;
      tile_i = tile_j = local % J.ti;
      ns_eff = J.ns_diag;
    }
  } else {
    const int tiles = J.ti * J.tj;
    split = local / tiles;
    local = local % tiles;
    tile_i = local / J.tj;
    tile_j = local % J.tj;
  }
  const int64_t nch = Rp / 16;
  const int64_t c_lo = split * nch / ns_eff, c_hi = (split + 1) * nch / ns_eff;
  const int njv = (J.qrows16 - NJ * tile_j < NJ) ? J.qrows16 - NJ * tile_j : NJ;    // column blocks of this tile that exist
  d4 acc[NI][NJ];
#pragma unroll
  for (int ii = 0; ii < NI; ++ii)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) acc[ii][jj] = (d4){0, 0, 0, 0};
  gcptr Pp = (gcptr)(J.P + (int64_t)(16 * NI * tile_i + c) * ld + 4 * g);      // job descriptors live in memory: see gcptr
  gcptr Qp = (gcptr)(J.Q + (int64_t)(16 * NJ * tile_j + c) * ld + 4 * g);
  const bool diag = J.sym && tile_i == tile_j && NI == NJ;
  if (diag)
    wgrad_loop<NI, NJ, false, true>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  else if (njv == NJ)
    wgrad_loop<NI, NJ, false, false>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  else
    wgrad_loop<NI, NJ, true, false>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  const int rowsP = 16 * NI * J.ti;
  gptr o = (gptr)(J.out + (int64_t)split * rowsP * J.ldo);
#pragma unroll
  for (int ii = 0; ii < NI; ++ii)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      if (jj < njv && !(diag && jj > ii)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          o[(int64_t)(16 * (NI * tile_i + ii) + g + 4 * t) * J.ldo + 16 * (NJ * tile_j + jj) + c] = acc[ii][jj][t];
      }
}

// Cooperative form: ONE WORKGROUP per (job, split, tile) task; its four waves take a quarter of the split's row range each and
// reduce their accumulators through LDS in a fixed order ((w0 + w2) + (w1 + w3)) before wave 0 stores the partial.  Same
// wave-level parallelism with a quarter of the split-K partials (less traffic for k_reduce_grouped, which is bandwidth /
// latency-bound), or twice the waves per SIMD at half the partials — fp64 MFMA needs >= 2 waves per SIMD for its pipe rate.
template <int NI, int NJ>
__global__ __launch_bounds__(256) void k_wgrad_coop(const WgradJob* __restrict__ jobs, int njobs, int nsplit, int64_t ld,
                                                    int64_t Rp, int total_tasks) {
  __shared__ double red[2 * NI * NJ * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x;
  int jb = 0;
  while (jb + 1 < njobs && w >= jobs[jb + 1].task_start) ++jb;
  const WgradJob J = jobs[jb];
  int local = w - J.task_start;
  int split, tile_i, tile_j, ns_eff = nsplit;
  if (J.sym) {
    const int n_off = J.ti * (J.ti - 1) / 2;
    if (local < nsplit * n_off) {
      split = local / n_off;
      local = local % n_off;
      tile_i = 1;
      while (tile_i * (tile_i + 1) / 2 <= local) ++tile_i;
      tile_j = local - tile_i * (tile_i - 1) / 2;
    } else {
      local -= nsplit * n_off;
      split = local / J.ti;
      tile_i = tile_j = local % J.ti;
      ns_eff = J.ns_diag;
    }
  } else {
    const int tiles = J.ti * J.tj;
    split = local / tiles;
    local = local % tiles;
    tile_i = local / J.tj;
    tile_j = local % J.tj;
  }
  const int64_t nch = Rp / 16;
  const int64_t s_lo = split * nch / ns_eff, s_hi = (split + 1) * nch / ns_eff;
  const int64_t c_lo = s_lo + (s_hi - s_lo) * wave / 4, c_hi = s_lo + (s_hi - s_lo) * (wave + 1) / 4;
  const int njv = (J.qrows16 - NJ * tile_j < NJ) ? J.qrows16 - NJ * tile_j : NJ;
  d4 acc[NI][NJ];
#pragma unroll
  for (int ii = 0; ii < NI; ++ii)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) acc[ii][jj] = (d4){0, 0, 0, 0};
  gcptr Pp = (gcptr)(J.P + (int64_t)(16 * NI * tile_i + c) * ld + 4 * g);      // job descriptors live in memory: see gcptr
  gcptr Qp = (gcptr)(J.Q + (int64_t)(16 * NJ * tile_j + c) * ld + 4 * g);
  const bool diag = J.sym && tile_i == tile_j && NI == NJ;
  if (diag)
    wgrad_loop<NI, NJ, false, true>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  else if (njv == NJ)
    wgrad_loop<NI, NJ, false, false>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  else
    wgrad_loop<NI, NJ, true, false>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, njv, acc);
  // fixed-order tree over the four waves: slot s of a wave's accumulator lives at red[region][s][lane]
  constexpr int NS = NI * NJ * 4;
  if (wave >= 2) {
    double* r = red + (wave - 2) * NS * 64 + lane;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int t = 0; t < 4; ++t) r[((ii * NJ + jj) * 4 + t) * 64] = acc[ii][jj][t];
  }
  __syncthreads();
  if (wave < 2) {
    const double* r = red + wave * NS * 64 + lane;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[ii][jj][t] += r[((ii * NJ + jj) * 4 + t) * 64];
  }
  __syncthreads();
  if (wave == 1) {
    double* r = red + lane;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int t = 0; t < 4; ++t) r[((ii * NJ + jj) * 4 + t) * 64] = acc[ii][jj][t];
  }
  __syncthreads();
  if (wave != 0) return;
  const double* r = red + lane;
  const int rowsP = 16 * NI * J.ti;
  gptr o = (gptr)(J.out + (int64_t)split * rowsP * J.ldo);
#pragma unroll
  for (int ii = 0; ii < NI; ++ii)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      if (jj < njv && !(diag && jj > ii)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          o[(int64_t)(16 * (NI * tile_i + ii) + g + 4 * t) * J.ldo + 16 * (NJ * tile_j + jj) + c] =
              acc[ii][jj][t] + r[((ii * NJ + jj) * 4 + t) * 64];
      }
}

// ------------------------------------------------------------------------------------------------------
// 128 x 128 tiles for the symmetric products P_d = A diag(v_d) A^T.  The 64 x 64 form above is bound by operand bandwidth out of
// L2 / Infinity Cache (every wave streams its own 64 + 64 rows: 256 bytes per MFMA); here ONE workgroup owns a 128 x 128 tile, its
// four waves the 64 x 64 sub-tiles, and each 16-data-row chunk of the 128 (+128) operand rows is staged ONCE through LDS
// (double-buffered, one barrier per chunk): 64 bytes per MFMA on a diagonal tile, 128 off the diagonal.  LDS layout [row][16 + 2]:
// the MFMA fragments (lane (g, c): row 16 ii + c, data row 4 t + g of k-step t — any bijection of k works) are conflict-free
// 8-byte reads, the staging stores 16-byte writes.  The sub-tile above the diagonal of a diagonal tile is not computed
// (k_reduce_grouped mirrors at 16-block granularity, as for the 64 x 64 form); all K splits of a job cover every tile.
// MEASURED SLOWER and therefore OFF by default (DSDGP_WGRAD_T128=1 enables it; gradients parity-tested): weight-gradient time per
// step 0.129 -> 0.169 ms at config 2, 2.49 -> 3.03 ms at config 3.  A third to a half of the operand traffic does not pay for one
// barrier and one LDS round trip per 16-row chunk with 64 MFMAs per wave in between (two workgroups per CU: 74 KB of LDS each).
#define T128_LD 18
int wgrad_t128_lds_bytes() { return (4 * 128 * T128_LD + 2 * 16) * (int)sizeof(double); }
__global__ __launch_bounds__(256) void k_wgrad_t128(const WgradJob* __restrict__ jobs, int njobs, int nsplit, int64_t ld, int64_t Rp) {
  extern __shared__ __attribute__((aligned(16))) double tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  int jb = 0;
  while (jb + 1 < njobs && (int)blockIdx.x >= jobs[jb + 1].task_start) ++jb;
  const WgradJob J = jobs[jb];
  int local = (int)blockIdx.x - J.task_start;
  const int T = J.ti, ntile = T * (T + 1) / 2;
  const int split = local / ntile;
  local %= ntile;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= local) ++ti;
  const int tj = local - ti * (ti + 1) / 2;
  const bool diag128 = ti == tj;
  const int si = wave >> 1, sj = wave & 1;
  const bool idle = diag128 && sj > si;        // the 64 x 64 sub-tile above the diagonal
  const bool dsub = diag128 && si == sj;       // diagonal sub-tile: 16-blocks with jj <= ii only
  const int64_t nch = Rp / 16;
  const int64_t c_lo = split * nch / nsplit, c_hi = (split + 1) * nch / nsplit;
  double* Pt = tsm;                            // [2][128][T128_LD]
  double* Qt = tsm + 2 * 128 * T128_LD;        // [2][128][T128_LD]  (unused on a diagonal tile: Q's rows are P's rows)
  double* Sc = tsm + 4 * 128 * T128_LD;        // [2][16]
  // staging: thread -> (row = tid / 2, half = tid % 2): data rows 8 half .. 8 half + 7 of its operand row (64 contiguous bytes)
  const int srow = tid >> 1, sh = tid & 1;
  typedef const d4 __attribute__((address_space(1)))* gd4;
  gcptr Pg = (gcptr)(J.P + (int64_t)(128 * ti + srow) * ld + 8 * sh);
  gcptr Qg = (gcptr)(J.Q + (int64_t)(128 * tj + srow) * ld + 8 * sh);
  gcptr Sg = (gcptr)J.scale;
  d4 p0 = (d4){0, 0, 0, 0}, p1 = p0, q0 = p0, q1 = p0;
  double sv = 1.0;
  auto gload = [&](int64_t ch) {
    const int64_t rb = ch * 16;
    p0 = *reinterpret_cast<gd4>(Pg + rb);
    p1 = *reinterpret_cast<gd4>(Pg + rb + 4);
    if (!diag128) {
      q0 = *reinterpret_cast<gd4>(Qg + rb);
      q1 = *reinterpret_cast<gd4>(Qg + rb + 4);
    }
    if (Sg && tid < 16) sv = Sg[rb + tid];
  };
  auto lstore = [&](int buf) {
    double* pd = Pt + (buf * 128 + srow) * T128_LD + 8 * sh;
    *reinterpret_cast<d4*>(pd) = p0;
    *reinterpret_cast<d4*>(pd + 4) = p1;
    if (!diag128) {
      double* qd = Qt + (buf * 128 + srow) * T128_LD + 8 * sh;
      *reinterpret_cast<d4*>(qd) = q0;
      *reinterpret_cast<d4*>(qd + 4) = q1;
    }
    if (tid < 16) Sc[buf * 16 + tid] = sv;
  };
  d4 acc[4][4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = (d4){0, 0, 0, 0};
  if (c_lo < c_hi) {
    gload(c_lo);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t ch = c_lo; ch < c_hi; ++ch) {
    const bool more = ch + 1 < c_hi;
    if (more) gload(ch + 1);                   // in flight during the MFMAs of this chunk
    if (!idle) {
      const double* __restrict__ pb = Pt + (cur * 128 + 64 * si + c) * T128_LD;
      const double* __restrict__ qb = (diag128 ? Pt : Qt) + (cur * 128 + 64 * sj + c) * T128_LD;
      const double* __restrict__ sc = Sc + cur * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 4 * t + g;
        const double s = sc[k];
        double pa[4], qv[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) pa[ii] = pb[16 * ii * T128_LD + k];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) qv[jj] = qb[16 * jj * T128_LD + k] * s;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            if (!dsub || jj <= ii) acc[ii][jj] = mfma_f64(pa[ii], qv[jj], acc[ii][jj]);
      }
    }
    if (more) lstore(cur ^ 1);                 // the other buffer: every wave finished reading it before the previous barrier
    __syncthreads();
    cur ^= 1;
  }
  if (idle) return;
  gptr o = (gptr)(J.out + (int64_t)split * (128 * T) * J.ldo);
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (!dsub || jj <= ii) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          o[(int64_t)(128 * ti + 64 * si + 16 * ii + g + 4 * t) * J.ldo + 128 * tj + 64 * sj + 16 * jj + c] = acc[ii][jj][t];
      }
}
int wgrad_t128_enabled() {
  static const int on = getenv("DSDGP_WGRAD_T128") ? atoi(getenv("DSDGP_WGRAD_T128")) : 0;
  return on;
}
int wgrad_t128_launch(dsdgp_ctx* ctx, const WgradJob* jobs_dev, int njobs, int total_tasks, int nsplit, int64_t ld, int64_t Rp,
                      hipStream_t stream) {
  if (total_tasks <= 0) return DSDGP_OK;
  hipStream_t st = stream ? stream : ctx->stream;
  ProfScope ps(ctx, "wgrad", st);
  static bool attr = false;
  if (!attr) {
    DS_HIP(hipFuncSetAttribute((const void*)k_wgrad_t128, hipFuncAttributeMaxDynamicSharedMemorySize, wgrad_t128_lds_bytes()));
    attr = true;
  }
  hipLaunchKernelGGL(k_wgrad_t128, dim3(total_tasks), dim3(256), wgrad_t128_lds_bytes(), st, jobs_dev, njobs, nsplit, ld, Rp);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

int wgrad_coop_enabled() {
  static const int on = getenv("DSDGP_WGRAD_COOP") ? atoi(getenv("DSDGP_WGRAD_COOP")) : 1;
  return on;
}

int wgrad_launch(dsdgp_ctx* ctx, const WgradJob* jobs_dev, int njobs, int total_tasks, int nsplit, int64_t ld, int64_t Rp,
                 int NI, int NJ, hipStream_t stream) {
  hipStream_t st = stream ? stream : ctx->stream;
  ProfScope ps(ctx, "wgrad", st);
  const bool coop = wgrad_coop_enabled();
  const dim3 grid(coop ? total_tasks : ceil_div(total_tasks, 4)), blk(256);
#define WG_CASE(I, Jn)                                                                                              \
  if (NI == I && NJ == Jn) {                                                                                        \
    if (coop)                                                                                                       \
      hipLaunchKernelGGL((k_wgrad_coop<I, Jn>), grid, blk, 0, st, jobs_dev, njobs, nsplit, ld, Rp, total_tasks);    \
    else                                                                                                            \
      hipLaunchKernelGGL((k_wgrad<I, Jn>), grid, blk, 0, st, jobs_dev, njobs, nsplit, ld, Rp, total_tasks);         \
    DS_HIP(hipGetLastError());                                                                                      \
    return DSDGP_OK;                                                                                                \
  }
  WG_CASE(4, 4)
  WG_CASE(2, 2)
  WG_CASE(4, 1)
  WG_CASE(2, 1)
#undef WG_CASE
  dsdgp_set_error("wgrad: tile shape %dx%d not built", NI, NJ);
  return DSDGP_ERR_UNSUPPORTED;
}
