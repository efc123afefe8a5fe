// "Split-M" variant of the SVGP_Layer chain (same math as layer.hip; layers.py:178-219 + utils.py:40-41).
//
// Why: v_mfma_f64_16x16x4_f64 needs >= 2 wavefronts per SIMD to run at its pipe rate (a lone wave issues one MFMA per
// ~59 ns, two waves one per ~45 ns per SIMD — profiles/r01_mfma_f64_microbench.txt), but a 20 000-row layer only has 1250
// sixteen-row blocks for 1024 SIMDs.  Here the 4 waves of a workgroup cooperate on ONE block of 16*CB data rows: wave w
// owns the output row-blocks {w, 7-w, 8+w, 15-w, ...} of every product (paired so that triangular products carry equal
// work), the activations live in a ping-pong LDS buffer in MFMA B-operand order ([k][16 rows], bank-conflict free), and
// each wave streams only ITS weight columns from L2 — no weight element is fetched twice inside a workgroup.  That gives
// 4x more, 4x shorter wave-tasks (5000 for cfg 2): full MFMA-pipe occupancy, ~98 % balance, and a 4x shorter critical path
// for the small first layer.
#include <stdlib.h>

#include "layer.hpp"

template <int MPB>
struct Own {
  static constexpr int NQ = (MPB >= 4) ? MPB / 4 : 1;
  static __device__ __forceinline__ int ib(int wave, int q) {
    if (MPB >= 8) return (q & 1) ? (q >> 1) * 8 + 7 - wave : (q >> 1) * 8 + wave;
    return wave;
  }
  static __device__ __forceinline__ bool active(int wave) { return MPB >= 4 || wave < MPB; }
};

// LDS carve (doubles): xs | act | red      (Z/l is read through L1/L2; ONE in-place activation buffer)
struct SmLds {
  int xs, act, red, total;
};
static inline SmLds sm_lds(int Mp, int D_in, int D_out, int CB) {
  SmLds L;
  int o = 0;
  L.xs = o; o += 16 * CB * (D_in + 1);
  o = (int)round_up(o, 2);
  L.act = o; o += CB * Mp * 16;
  L.red = o;
  const int red_fwd = 4 * CB * 16 * (1 + D_out) + 2 * 4 * CB * 16;      // s1 + mu[D_out] + 2 x s2
  const int red_bwd = 4 * CB * 16 * D_in;                               // dX partials
  o += red_fwd > red_bwd ? red_fwd : red_bwd;
  L.total = o;
  return L;
}

template <int MPB, int KIND, bool WHITE, int CB>
__global__ __launch_bounds__(256) void k_layer_fwd_sm(const LayerFwdArgs a, const SmLds L) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16, NQ = Own<MPB>::NQ;
  const int Din = a.D_in, Dout = a.D_out;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const double* __restrict__ zs = a.Zs;
  double* xs = smem + L.xs;
  double* actA = smem + L.act;                            // single in-place activation buffer ([k][16 rows])
  double* actB = actA;
  double* red_s1 = smem + L.red;                          // [4][CB][16]
  double* red_mu = red_s1 + 4 * CB * 16;                  // [4][CB][Dout][16]
  double* red_s2 = red_mu + 4 * CB * 16 * Dout;           // [2][4][CB][16]
  const double* ils = a.hyp + HYP_ILS;
  const int64_t r0 = (int64_t)blockIdx.x * 16 * CB;
  for (int idx = tid; idx < 16 * CB * Din; idx += 256) {
    const int rr = idx / Din, j = idx % Din;
    int64_t row = r0 + rr;
    if (row > a.Rin - 1) row = a.Rin - 1;
    xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
  }
  __syncthreads();
  const bool act = Own<MPB>::active(wave);
  const double s2 = a.hyp[HYP_VAR];

  // --- Kuf tile (layers.py:184): own row-blocks -> act
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = 16 * ib + g + 4 * t;
          double r2 = 0.0;
          for (int j = 0; j < Din; ++j) {
            const double df = zs[m * Din + j] - xs[(cb * 16 + c) * (Din + 1) + j];
            r2 = fma(df, df, r2);
          }
          actA[(cb * Mp + m) * 16 + c] = (m < a.M) ? kern_val<KIND>(r2, s2) : 0.0;
        }
    }
  }
  __syncthreads();

  d4 acc[NQ][CB];
  // --- a1 = Lu^{-1} k (layers.py:186): out block ib sums k-blocks kb <= ib ; weights LinvT[k][i]
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[q][cb] = (d4){0, 0, 0, 0};
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
      const double* __restrict__ W = a.LinvT + 16 * ib + c + g * Mp;
#pragma unroll 2
      for (int kb = 0; kb <= ib; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double av = W[(16 * kb + 4 * s) * Mp];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[q][cb] = mfma_f64(av, actA[(cb * Mp + 16 * kb + 4 * s + g) * 16 + c], acc[q][cb]);
        }
      }
    }
  }
  {
    double p[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      p[cb] = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) p[cb] = fma(acc[q][cb][t], acc[q][cb][t], p[cb]);
      p[cb] = sum_groups(p[cb]);
      if (g == 0) red_s1[(wave * CB + cb) * 16 + c] = act ? p[cb] : 0.0;
    }
  }
  __syncthreads();   // every wave has finished reading k before a1 overwrites it in place
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) actB[(cb * Mp + 16 * ib + g + 4 * t) * 16 + c] = acc[q][cb][t];
    }
  }
  __syncthreads();
  const double* actIn = actB;   // holds "a" for the q_sqrt stage (white: a = a1)
  // --- a = Lu^{-T} a1 (layers.py:188): out block ib sums kb >= ib ; weights Linv[k][i]
  if (!WHITE) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[q][cb] = (d4){0, 0, 0, 0};
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB>::ib(wave, q);
        const double* __restrict__ W = a.Linv + 16 * ib + c + g * Mp;
#pragma unroll 2
        for (int kb = ib; kb < MPB; ++kb) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double av = W[(16 * kb + 4 * s) * Mp];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
              acc[q][cb] = mfma_f64(av, actB[(cb * Mp + 16 * kb + 4 * s + g) * 16 + c], acc[q][cb]);
          }
        }
      }
    }
    __syncthreads();   // a1 fully consumed -> overwrite with a
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int t = 0; t < 4; ++t) actA[(cb * Mp + 16 * ib + g + 4 * t) * 16 + c] = acc[q][cb][t];
      }
    }
    actIn = actA;
  }
  // acc now holds this wave's rows of "a": save for the backward pass, and the partial mean a . q_mu (layers.py:190)
  if (act && a.Asave && blockIdx.y == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int64_t r = r0 + cb * 16 + c;
        if (r < a.ldA) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            a.Asave[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r] = (r < a.Rin) ? acc[q][cb][t] : 0.0;
        }
      }
    }
  }
  for (int d = 0; d < Dout; ++d) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      double mu = 0.0;
      if (act) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
          for (int t = 0; t < 4; ++t) mu = fma(acc[q][cb][t], a.qmu[(16 * ib + g + 4 * t) * Dout + d], mu);
        }
      }
      mu = sum_groups(mu);
      if (g == 0) red_mu[((wave * CB + cb) * Dout + d) * 16 + c] = mu;
    }
  }
  __syncthreads();

  const double kdiag = a.hyp[HYP_KDIAG];
  // small launches (the N-row first layer) spread their D_out products over gridDim.y workgroups per row block
  const int dchunk = (Dout + (int)gridDim.y - 1) / (int)gridDim.y;
  const int d_lo = (int)blockIdx.y * dchunk, d_hi = (d_lo + dchunk < Dout) ? d_lo + dchunk : Dout;
  for (int d = d_lo; d < d_hi; ++d) {
    // --- c_d = q_sqrt_d^T a ; |c_d|^2 (replaces SK/B of layers.py:195-212): out block ib sums kb >= ib
    d4 cacc[NQ][CB];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) cacc[q][cb] = (d4){0, 0, 0, 0};
    if (act) {
      const double* __restrict__ Td = a.Tp + (int64_t)d * Mp * Mp;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB>::ib(wave, q);
        const double* __restrict__ W = Td + 16 * ib + c + g * Mp;
#pragma unroll 2
        for (int kb = ib; kb < MPB; ++kb) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double av = W[(16 * kb + 4 * s) * Mp];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
              cacc[q][cb] = mfma_f64(av, actIn[(cb * Mp + 16 * kb + 4 * s + g) * 16 + c], cacc[q][cb]);
          }
        }
      }
    }
    double* rs2 = red_s2 + (d & 1) * 4 * CB * 16;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      double p = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) p = fma(cacc[q][cb][t], cacc[q][cb][t], p);
      p = sum_groups(p);
      if (g == 0) rs2[(wave * CB + cb) * 16 + c] = act ? p : 0.0;
    }
    __syncthreads();
    if (wave == (d & 3)) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int64_t r = r0 + cb * 16 + c;
        if (r < a.Rin) {
          double s1 = 0.0, s2sum = 0.0, mu = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            s1 += red_s1[(w * CB + cb) * 16 + c];
            s2sum += rs2[(w * CB + cb) * 16 + c];
            mu += red_mu[((w * CB + cb) * Dout + d) * 16 + c];
          }
          const double var = kdiag - s1 + s2sum;                               // layers.py:212-217
          if (a.mean_kind == DSDGP_MEAN_IDENTITY) {                            // layers.py:219
            mu += a.X[r * Din + d];
          } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
            double m2 = 0.0;
            for (int j = 0; j < Din; ++j) m2 = fma(a.X[r * Din + j], a.mean_A[j * Dout + d], m2);
            mu += m2;
          }
          for (int s = g; s < a.rep; s += 4) {
            const int64_t orow = (int64_t)s * a.Rin + r;
            const int64_t o = orow * Dout + d;
            if (a.mean) a.mean[o] = mu;
            if (a.var) a.var[o] = var;
            if (a.F && a.z) {
              const double zv = a.z[(orow / a.n_inner) * a.zs_s + (orow % a.n_inner) * a.zs_n + d * a.zs_d];
              a.F[o] = mu + zv * sqrt(var + a.jitter);                         // utils.py:41 (no clamp)
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward (math: see layer.hip).  hyp_part gets one partial row per WAVE: index (blockIdx.x * 4 + wave).
// ------------------------------------------------------------------------------------------------------
template <int MPB, int KIND, bool WHITE, int CB>
__global__ __launch_bounds__(256) void k_layer_bwd_sm(const LayerBwdArgs a, const SmLds L) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16, NQ = Own<MPB>::NQ;
  const int Din = a.D_in, Dout = a.D_out;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const double* __restrict__ zs = a.Zs;
  double* xs = smem + L.xs;
  double* actA = smem + L.act;
  double* actB = actA;
  double* redx = smem + L.red;                            // [4][CB][Din][16]
  const double* ils = a.hyp + HYP_ILS;
  const int64_t r0 = (int64_t)blockIdx.x * 16 * CB;
  for (int idx = tid; idx < 16 * CB * Din; idx += 256) {
    const int rr = idx / Din, j = idx % Din;
    int64_t row = r0 + rr;
    if (row > a.Rin - 1) row = a.Rin - 1;
    xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
  }
  const bool act = Own<MPB>::active(wave);
  const double s2 = a.hyp[HYP_VAR];
  int64_t r[CB];
  bool rin[CB], rvalid[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    r[cb] = r0 + cb * 16 + c;
    rin[cb] = r[cb] < a.ldA;
    rvalid[cb] = r[cb] < a.Rin;
  }
  d4 av[NQ][CB], acc[NQ][CB];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      acc[q][cb] = (d4){0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double v = (act && rin[cb]) ? a.Asave[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r[cb]] : 0.0;
        av[q][cb][t] = v;
        if (act) actA[(cb * Mp + 16 * ib + g + 4 * t) * 16 + c] = v;
      }
    }
  }
  __syncthreads();
  double gsum[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) gsum[cb] = 0.0;
  for (int d = 0; d < Dout; ++d) {
    double vd2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const double vd = rin[cb] ? a.VB[(int64_t)d * a.ldA + r[cb]] : 0.0;
      gsum[cb] += vd;
      vd2[cb] = 2.0 * vd;
    }
    if (act) {
      const double* __restrict__ Sd = a.Sd + (int64_t)d * Mp * Mp;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB>::ib(wave, q);
        const double* __restrict__ W = Sd + 16 * ib + c + g * Mp;
#pragma unroll 2
        for (int kb = 0; kb < MPB; ++kb) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double wv = W[(16 * kb + 4 * s) * Mp];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
              acc[q][cb] = mfma_f64(wv, actA[(cb * Mp + 16 * kb + 4 * s + g) * 16 + c] * vd2[cb], acc[q][cb]);
          }
        }
      }
    }
  }
  if (act) {
    for (int sp = 0; sp < a.DP4 / 4; ++sp) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const double bv = rin[cb] ? a.MB[(int64_t)(4 * sp + g) * a.ldA + r[cb]] : 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB>::ib(wave, q);
          acc[q][cb] = mfma_f64(a.qmu4[(16 * ib + c) * a.DP4 + 4 * sp + g], bv, acc[q][cb]);
        }
      }
    }
  }
  __syncthreads();   // "a" fully consumed from LDS -> overwrite with abar in place
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (WHITE) acc[q][cb][t] -= 2.0 * gsum[cb] * av[q][cb][t];
          actB[(cb * Mp + 16 * ib + g + 4 * t) * 16 + c] = acc[q][cb][t];
        }
    }
  }
  __syncthreads();
  // b = Ku^{-1} abar (dense)   |   white: kbar = Lu^{-T} a1bar (k-blocks >= own block)
  d4 bb[NQ][CB];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bb[q][cb] = (d4){0, 0, 0, 0};
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
      const double* __restrict__ W = (WHITE ? a.Linv : a.Kinv) + 16 * ib + c + g * Mp;
#pragma unroll 2
      for (int kb = (WHITE ? ib : 0); kb < MPB; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double wv = W[(16 * kb + 4 * s) * Mp];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            bb[q][cb] = mfma_f64(wv, actB[(cb * Mp + 16 * kb + 4 * s + g) * 16 + c], bb[q][cb]);
        }
      }
    }
  }
  // E, kbar, GW and the hyper-parameter / input-gradient partial sums over this wave's inducing rows
  double svar = 0.0;
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = 16 * ib + g + 4 * t;
          const double e = WHITE ? bb[q][cb][t] : bb[q][cb][t] - gsum[cb] * av[q][cb][t];
          const double kbar = WHITE ? e : e - gsum[cb] * av[q][cb][t];
          double r2 = 0.0;
          for (int j = 0; j < Din; ++j) {
            const double df = zs[m * Din + j] - xs[(cb * 16 + c) * (Din + 1) + j];
            r2 = fma(df, df, r2);
          }
          double k, dk;
          kern_val_grad<KIND>(r2, s2, k, dk);
          const bool ok = rvalid[cb] && (m < a.M);
          svar += ok ? kbar * k : 0.0;
          const double w = ok ? kbar * dk : 0.0;
          bb[q][cb][t] = w;
          if (rin[cb]) {
            a.E[(int64_t)m * a.ldA + r[cb]] = e;
            a.GW[(int64_t)m * a.ldA + r[cb]] = w;
          }
        }
    }
  }
  svar = sum_wave(svar);
  double gk = 0.0;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) gk += (rvalid[cb] && g == 0 && wave == 0) ? gsum[cb] : 0.0;
  gk = sum_wave(gk);
  double* hp = a.hyp_part + ((int64_t)blockIdx.x * 4 + wave) * (Din + 2);
  if (lane == 0) {
    hp[0] = svar / s2;
    hp[1] = gk;
  }
  for (int j = 0; j < Din; ++j) {
    double sl = 0.0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      double sx = 0.0;
      if (act) {
        const double xv = xs[(cb * 16 + c) * (Din + 1) + j];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB>::ib(wave, q);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const double df = xv - zs[(16 * ib + g + 4 * t) * Din + j];
            const double wdf = bb[q][cb][t] * df;
            sx += wdf;
            sl = fma(wdf, df, sl);
          }
        }
      }
      sx = sum_groups(sx);
      if (g == 0) redx[((wave * CB + cb) * Din + j) * 16 + c] = sx;
    }
    sl = sum_wave(sl);
    if (lane == 0) hp[2 + j] = -2.0 * ils[j] * sl;
  }
  if (!a.dX) return;
  __syncthreads();
  for (int idx = tid; idx < CB * 16 * Din; idx += 256) {
    const int j = idx % Din, rr = idx / Din, cb = rr / 16, cc = rr % 16;
    const int64_t row = r0 + rr;
    if (row < a.Rin) {
      double sx = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) sx += redx[((w * CB + cb) * Din + j) * 16 + cc];
      double dx = 2.0 * ils[j] * sx;
      if (a.mean_kind == DSDGP_MEAN_IDENTITY) {
        dx += a.MB[(int64_t)j * a.ldA + row];
      } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
        for (int d = 0; d < Dout; ++d) dx = fma(a.mean_A[j * Dout + d], a.MB[(int64_t)d * a.ldA + row], dx);
      }
      a.dX[row * Din + j] = dx;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------------
int sm_chain_cb() {
  static const int cb = getenv("DSDGP_SM_CB") ? atoi(getenv("DSDGP_SM_CB")) : 1;
  return cb == 2 ? 2 : 1;
}
int sm_chain_enabled() {
  static const int on = getenv("DSDGP_CHAIN_SM") ? atoi(getenv("DSDGP_CHAIN_SM")) : 1;
  return on;
}
int64_t sm_hyp_parts(int64_t ld) { return 4 * (int64_t)ceil_div(ld, 16 * sm_chain_cb()); }

template <int MPB, int KIND, bool WHITE, int CB>
static int fwd_sm_go(dsdgp_ctx* ctx, const LayerFwdArgs& a) {
  const SmLds L = sm_lds(MPB * 16, a.D_in, a.D_out, CB);
  const size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_fwd(sm): needs %zu B LDS", lds);
    return DSDGP_ERR_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    DS_HIP(hipFuncSetAttribute((const void*)k_layer_fwd_sm<MPB, KIND, WHITE, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(ctx, "layer_fwd");
  const int nrow = ceil_div(a.Rin, 16 * CB);
  int ds = a.d_split > 0 ? a.d_split : 1;
  if (ds > a.D_out) ds = a.D_out;
  hipLaunchKernelGGL((k_layer_fwd_sm<MPB, KIND, WHITE, CB>), dim3(nrow, ds), dim3(256), lds, ctx->stream, a, L);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
template <int MPB, int KIND, bool WHITE, int CB>
static int bwd_sm_go(dsdgp_ctx* ctx, const LayerBwdArgs& a) {
  const SmLds L = sm_lds(MPB * 16, a.D_in, a.D_out, CB);
  const size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_bwd(sm): needs %zu B LDS", lds);
    return DSDGP_ERR_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    DS_HIP(hipFuncSetAttribute((const void*)k_layer_bwd_sm<MPB, KIND, WHITE, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(ctx, "layer_bwd");
  hipLaunchKernelGGL((k_layer_bwd_sm<MPB, KIND, WHITE, CB>), dim3(ceil_div(a.ldA, 16 * CB)), dim3(256), lds, ctx->stream, a, L);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

#define SM_DISPATCH(FN, ARGS)                                                                      \
  const int cb = sm_chain_cb();                                                                    \
  switch (Mp) {                                                                                    \
    SM_CASE(FN, 2, ARGS) SM_CASE(FN, 4, ARGS) SM_CASE(FN, 8, ARGS) SM_CASE(FN, 16, ARGS)           \
    default:                                                                                       \
      dsdgp_set_error("layer chain (sm): padded inducing count %d not built", Mp);                 \
      return DSDGP_ERR_UNSUPPORTED;                                                                \
  }
#define SM_CASE(FN, MPB, ARGS)                                                                     \
  case MPB * 16:                                                                                   \
    if (kern_kind == DSDGP_KERN_RBF) {                                                             \
      if (white) return cb == 2 ? FN<MPB, DSDGP_KERN_RBF, true, 2> ARGS : FN<MPB, DSDGP_KERN_RBF, true, 1> ARGS;   \
      return cb == 2 ? FN<MPB, DSDGP_KERN_RBF, false, 2> ARGS : FN<MPB, DSDGP_KERN_RBF, false, 1> ARGS;            \
    } else {                                                                                       \
      if (white) return cb == 2 ? FN<MPB, DSDGP_KERN_MATERN52, true, 2> ARGS : FN<MPB, DSDGP_KERN_MATERN52, true, 1> ARGS; \
      return cb == 2 ? FN<MPB, DSDGP_KERN_MATERN52, false, 2> ARGS : FN<MPB, DSDGP_KERN_MATERN52, false, 1> ARGS;  \
    }

int layer_fwd_sm_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white) {
  SM_DISPATCH(fwd_sm_go, (ctx, a))
}
int layer_bwd_sm_launch(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white) {
  SM_DISPATCH(bwd_sm_go, (ctx, a))
}
