// The LAST layer of a training step as ONE launch: forward chain (layers.py:184-219) -> Gaussian variational expectations and their
// adjoints (dgp.py:89-90) -> reverse mode of the same 16-row block, with the state the two chains share kept on the chip.
//
// Why (profiles/r05_timeline_step.txt, config 2): the D_out = 1 forward chain (38 us), a 6.5 us dependent-launch gap and its backward
// chain (42 us) touch the same row blocks; the backward chain starts by re-reading the `a` tile the forward chain stored, re-staging
// x / l and — the expensive part — forming abar's variance part as the DENSE product 2 vbar S a (S = q_sqrt q_sqrt^T, 2 M^2 flops per
// row) because the triangular pair q_sqrt (q_sqrt^T a) needs c = q_sqrt^T a, which only the forward chain has.  In one workgroup c is
// still in registers when the adjoints of the row block are known:
//     abar = 2 vbar q_sqrt c + q_mu mbar            (one TRIANGULAR product: 72 instead of 128 MFMAs per wave at M = 128)
//     kbar = Ku^-1 abar - 2 vbar a                  (dense; the accumulators start at -2 vbar a, so `a` dies there)
// and nothing travels through HBM between the halves: x / l stays staged in LDS, `a` is stored once (the weight-gradient products
// read it) and never read back, the launch boundary and the second ramp-up are gone.  The distances are recomputed for k and dk / dr2
// as in the unfused chain (4 % of a workgroup; keeping eight more doubles per lane across both chains costs an occupancy step).
//
// Scope: D_out = 1, Gaussian likelihood, non-white, narrow inputs, the paired instances (two row blocks per wave: Mp = 128 on four
// waves, Mp = 256 on eight), Zero mean function (what init_layers_linear gives a last layer), algebraic dl/dKu assembly (no E), one
// output row per input row (an inner-layer input).  Everything else
// takes the two chains (model_schedule.hpp decides; DSDGP_FORCE=last_fuse=0 switches this launch off).  Same outputs as the pair:
// mean / var, lik_part, lik_MB / lik_VB, Asave, [X^T;1], GW, hyp_part, dX or MBp / VBp — the weight-gradient products, the reductions
// and the assembly read the same buffers (not bit-identical to the pair: abar is a triangular pair here, a dense product there, and the
// likelihood sums are taken in another order — tests/test_gpu_round6.py compares both schedules and each with the oracle).
#include "layer_sm_impl.hpp"

struct LastLds {
  int xs, act, red, mv, total;
};
static inline LastLds last_lds(int Mp, int D_in, int NW) {
  LastLds L;
  int o = 0;
  L.xs = o; o += 16 * (D_in + 1);
  o = (int)round_up(o, 2);
  L.act = o; o += Mp * 16;
  L.red = o;
  // red: [s1 | s2 | mu] (3 x NW x 16) and the distance code's wave-private scratch (NW x 32) in the forward half, the dX partials
  // (NW x D_in x 16) in the reverse half
  int red = 3 * NW * 16;
  if (red < NW * 32) red = NW * 32;
  if (red < NW * 16 * D_in) red = NW * 16 * D_in;
  o += red;
  L.mv = o; o += 32;              // mbar[16] | vbar[16] of the row block: alive from the likelihood epilogue to the end
  L.total = o;
  return L;
}

template <int MPB, int NW, int KIND>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 5 : 4)) void k_layer_last(const LayerFwdArgs a, const LayerBwdArgs b, const LastLds L,
                                                                            const int hyp_rows) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16;
  static_assert(Own<MPB, NW>::NQ == 2 && MPB % NW == 0, "paired instances only");
  const int Din = a.D_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = DS_WAVE_ID(tid);
  const int g = lane >> 4, c = lane & 15;
  const int ib0 = Own<MPB, NW>::ib(wave, 0), ib1 = Own<MPB, NW>::ib(wave, 1);
  double* xs = smem + L.xs;
  double* actb = smem + L.act;
  double* red_s1 = smem + L.red;               // [NW][16]
  double* red_s2 = red_s1 + NW * 16;           // [NW][16]
  double* red_mu = red_s2 + NW * 16;           // [NW][16]
  double* redx = smem + L.red;                 // reverse half: [NW][D_in][16]
  double* mv = smem + L.mv;
  const double* ils = a.hyp + HYP_ILS;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const double s2 = a.hyp[HYP_VAR];
  const int64_t r = r0 + c;
  const bool rin = r < a.ldA, rvalid = r < a.Rin;

  // ------------------------------------------------------------------ forward half (k_layer_fwd_sm, D_out = 1)
  if (a.XT1) {
    for (int idx = tid; idx < 16 * (Din + 1); idx += NW * 64) {
      const int j = idx >> 4, rr = idx & 15;
      const int64_t rw = r0 + rr;
      if (rw < a.ldA) a.XT1[(int64_t)j * a.ldA + rw] = (rw < a.Rin) ? (j < Din ? a.X[rw * Din + j] : 1.0) : 0.0;
    }
  }
  for (int idx = tid; idx < 16 * Din; idx += NW * 64) {
    const int rr = idx / Din, j = idx % Din;
    int64_t row = r0 + rr;
    if (row > a.Rin - 1) row = a.Rin - 1;
    xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
  }
  __syncthreads();
  {
    d4 r2[2];
    sqdist_narrow<2, MPB, NW>(a.Zs, xs, Din, wave, g, c, true, red_s1, r2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ib = q ? ib1 : ib0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = 16 * ib + g + 4 * t;
        const double kv = kern_val<KIND>(r2[q][t], s2);
        actb[m * 16 + c] = (m < a.M) ? kv : 0.0;
      }
    }
  }
  __syncthreads();
  d4 acc[2];
  acc[0] = acc[1] = (d4){0, 0, 0, 0};
  chain_range2<Mp, false>(a.LinvT, actb, ib0, ib1, MPB, g, c, acc[0], acc[1]);           // a1 = Lu^-1 k (layers.py:186)
  {
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) p = fma(acc[q][t], acc[q][t], p);
    p = sum_groups(p);
    if (g == 0) red_s1[wave * 16 + c] = p;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) actb[(16 * (q ? ib1 : ib0) + g + 4 * t) * 16 + c] = acc[q][t];
  __syncthreads();
  acc[0] = acc[1] = (d4){0, 0, 0, 0};
  chain_range2<Mp, true>(a.Linv, actb, ib0, ib1, MPB, g, c, acc[0], acc[1]);             // a = Lu^-T a1 (layers.py:188)
  __syncthreads();
  {
    // a -> LDS; the partial mean a . q_mu of this wave's rows (layers.py:190)
    double mu = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ib = q ? ib1 : ib0;
      double qv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qv[t] = a.qmu[16 * ib + g + 4 * t];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        actb[(16 * ib + g + 4 * t) * 16 + c] = acc[q][t];
        mu = fma(acc[q][t], qv[t], mu);
      }
    }
    mu = sum_groups(mu);
    if (g == 0) red_mu[wave * 16 + c] = mu;
  }
  __syncthreads();
  d4 cacc[2];
  cacc[0] = cacc[1] = (d4){0, 0, 0, 0};
  chain_range2<Mp, true>(a.Tp, actb, ib0, ib1, MPB, g, c, cacc[0], cacc[1]);             // c = q_sqrt^T a
  {
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) p = fma(cacc[q][t], cacc[q][t], p);
    p = sum_groups(p);
    if (g == 0) red_s2[wave * 16 + c] = p;
  }
  __syncthreads();        // partials published; every wave is done reading `a` from LDS
  // mean / var of the 16 rows, Gaussian variational expectations and their adjoints: lanes 0 .. 15 of wave 0, one row each
  if (wave == 0) {
    const bool on = lane < 16;
    const int cc = lane & 15;
    const int64_t rw = r0 + cc;
    const bool live = on && rw < a.Rin;
    double s1 = 0.0, s2sum = 0.0, mu = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      s1 += red_s1[w * 16 + cc];
      s2sum += red_s2[w * 16 + cc];
      mu += red_mu[w * 16 + cc];
    }
    const double var = a.hyp[HYP_KDIAG] - s1 + s2sum;                                    // layers.py:212-217
    const int64_t rc = live ? rw : 0;
    const double lik_s2 = a.lik_const[0];
    const double y = a.lik_Y[rc % a.n_inner];
    const double qq = (y - mu) * (y - mu) + var;
    double ve = live ? (-0.91893853320467274178 - 0.5 * log(lik_s2)) - 0.5 * qq / lik_s2 : 0.0;
    double dl = live ? -0.5 / lik_s2 + 0.5 * qq / (lik_s2 * lik_s2) : 0.0;
    const double mb = live ? -a.lik_w * (y - mu) / lik_s2 : 0.0;
    const double vb = live ? 0.5 * a.lik_w / lik_s2 : 0.0;
    if (live) {
      if (a.mean) a.mean[rw] = mu;
      if (a.var) a.var[rw] = var;
    }
    if (on) {
      if (rw < a.lik_ld) {
        a.lik_MB[rw] = mb;
        a.lik_VB[rw] = vb;
      }
      mv[cc] = mb;
      mv[16 + cc] = vb;
    }
    ve = sum_wave(ve);
    dl = sum_wave(dl);
    if (lane == 0) {
      a.lik_part[2 * (int64_t)blockIdx.x] = ve;
      a.lik_part[2 * (int64_t)blockIdx.x + 1] = dl;
    }
  }
  // c -> LDS (the B operand of the next product); `a` to global memory for the weight-gradient products — from the registers, now:
  // the accumulators are re-used below
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) actb[(16 * (q ? ib1 : ib0) + g + 4 * t) * 16 + c] = cacc[q][t];
  if (rin) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) a.Asave[(int64_t)(16 * (q ? ib1 : ib0) + g + 4 * t) * a.ldA + r] = rvalid ? acc[q][t] : 0.0;
  }
  __syncthreads();

  // ------------------------------------------------------------------ reverse half (k_layer_bwd_sm, D_out = 1, no E)
  const double md = mv[c], vd = mv[16 + c];          // upstream adjoints of data row c (0 beyond Rin); sum_d vbar_d = vd
  d4 y[2];
  y[0] = y[1] = (d4){0, 0, 0, 0};
  chain_range2<Mp, false>(b.TpT, actb, ib0, ib1, MPB, g, c, y[0], y[1]);                 // q_sqrt c = S a
  {
    // abar = 2 vbar S a + q_mu mbar
    const double vd2 = 2.0 * vd;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ib = q ? ib1 : ib0;
      double qv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qv[t] = a.qmu[16 * ib + g + 4 * t];
#pragma unroll
      for (int t = 0; t < 4; ++t) y[q][t] = fma(vd2, y[q][t], qv[t] * md);
    }
  }
  __syncthreads();        // c fully consumed -> abar in its place
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      actb[(16 * (q ? ib1 : ib0) + g + 4 * t) * 16 + c] = y[q][t];
      acc[q][t] = -2.0 * vd * acc[q][t];             // kbar = Ku^-1 abar - 2 g a: the product below accumulates onto -2 g a
    }
  __syncthreads();
  chain_dense2<Mp, MPB>(b.Kinv, actb, ib0, ib1, g, c, acc[0], acc[1]);
  // recompute the Kuf tile for GW = kbar * dk/dr2 (x / l is still staged)
  d4 r2[2];
  sqdist_narrow<2, MPB, NW>(b.Zs, xs, Din, wave, g, c, true, redx, r2);
  __syncthreads();        // the distance code's scratch lives where the dX partials go
  double svar = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ib = q ? ib1 : ib0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = 16 * ib + g + 4 * t;
      double k, dk;
      kern_val_grad<KIND>(r2[q][t], s2, k, dk);
      const bool ok = rvalid && (m < b.M);
      const double kbar = acc[q][t];
      svar += ok ? kbar * k : 0.0;
      acc[q][t] = ok ? kbar * dk : 0.0;
    }
  }
  svar = sum_wave(svar);
  const double gk = sum_wave((rvalid && g == 0 && wave == 0) ? vd : 0.0);
  double* hp = b.hyp_part + ((int64_t)blockIdx.x * hyp_rows + wave) * (Din + 2);
  if (lane == 0) {
    hp[0] = svar / s2;
    hp[1] = gk;
  }
  // the reduction plan counts hyp_rows partial rows per row block (the waves of the unfused backward instance): the surplus reads zero
  for (int e = tid; e < (hyp_rows - NW) * (Din + 2); e += NW * 64) b.hyp_part[((int64_t)blockIdx.x * hyp_rows + NW) * (Din + 2) + e] = 0.0;
  {
    const int jn = Din;
    const bool single = 16 * jn <= NW * 64;
    double pre_z = 0.0, pre_var = 1.0;
    if (b.MBp && single && tid < 16 * jn) {
      const int j = tid / 16, cc = tid % 16;
      const int64_t row = r0 + cc;
      const int d = j - b.prop;
      if (row < b.Rin && d >= 0) {
        pre_z = b.zp[(row / b.n_inner) * b.zp_s + (row % b.n_inner) * b.zp_n + d * b.zp_d];
        pre_var = b.varp[row * b.Dp + d];
      }
    }
    double w1 = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) w1 += acc[q][t];
    w1 = sum_groups(w1);
    for (int kk = 0; kk < jn; kk += 16) {
      d4 wz = (d4){0, 0, 0, 0}, z2 = (d4){0, 0, 0, 0};
      const int jc = (kk + c < jn) ? kk + c : jn - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ib = q ? ib1 : ib0;
        double zv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) zv[t] = b.Zs[(int64_t)(16 * ib + g + 4 * t) * Din + jc];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          wz = mfma_f64(zv[t], acc[q][t], wz);
          z2 = mfma_f64(zv[t] * zv[t], acc[q][t], z2);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = kk + g + 4 * t;
        const bool in = j < jn;
        const double xv = xs[c * (jn + 1) + (in ? j : jn - 1)];
        if ((b.dX || b.MBp) && in) redx[(wave * jn + j) * 16 + c] = fma(xv, w1, -wz[t]);
        double sl = fma(xv * xv, w1, fma(-2.0 * xv, wz[t], z2[t]));
        sl += dpp_or_zero<0x111, 0xf>(sl);
        sl += dpp_or_zero<0x112, 0xf>(sl);
        sl += dpp_or_zero<0x114, 0xf>(sl);
        sl += dpp_or_zero<0x118, 0xf>(sl);
        if (c == 15 && in) hp[2 + j] = -2.0 * ils[j] * sl;
      }
    }
    if (b.dX || b.MBp) {
      __syncthreads();
      for (int idx = tid; idx < 16 * jn; idx += NW * 64) {
        const int j = b.MBp ? idx / 16 : idx % jn, cc = b.MBp ? idx % 16 : idx / jn;
        const int64_t row = r0 + cc;
        if (row < b.Rin) {
          double sx = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) sx += redx[(w * jn + j) * 16 + cc];
          double dx = 2.0 * ils[j] * sx;
          if (b.MBp) {
            const int d = j - b.prop;
            if (d >= 0) {
              const double zv = single ? pre_z : b.zp[(row / b.n_inner) * b.zp_s + (row % b.n_inner) * b.zp_n + d * b.zp_d];
              const double vv = single ? pre_var : b.varp[row * b.Dp + d];
              b.MBp[(int64_t)d * b.ldA + row] = dx;
              b.VBp[(int64_t)d * b.ldA + row] = dx * zv * 0.5 * rsqrt(vv + b.jitter);
            }
          } else {
            b.dX[row * Din + j] = dx;
          }
        } else if (b.MBp && row < b.ldA && j >= b.prop) {
          b.MBp[(int64_t)(j - b.prop) * b.ldA + row] = 0.0;
          b.VBp[(int64_t)(j - b.prop) * b.ldA + row] = 0.0;
        }
      }
    }
  }
  if (rin) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) b.GW[(int64_t)(16 * (q ? ib1 : ib0) + g + 4 * t) * b.ldA + r] = acc[q][t];
  }
}

template <int MPB, int NW, int KIND>
static int last_go(dsdgp_ctx* ctx, const LayerFwdArgs& a, const LayerBwdArgs& b, int hyp_rows) {
  const LastLds L = last_lds(MPB * 16, a.D_in, NW);
  const size_t lds = (size_t)L.total * sizeof(double);
  ProfScope ps(ctx, "layer_last");
  DS_LAUNCH((k_layer_last<MPB, NW, KIND>), dim3(ceil_div(a.ldA, 16)), dim3(NW * 64), lds, ctx->stream, a, b, L, hyp_rows);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// shapes this launch exists for (the caller checks the model-level conditions: layer position, likelihood, gradient set)
int layer_last_built(int Mp, int D_in, int D_out) { return (Mp == 128 || Mp == 256) && D_out == 1 && D_in <= 16; }
// waves per row block of the instance for this padded inducing count
int layer_last_waves(int Mp) { return Mp == 128 ? 4 : 8; }

int layer_last_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, const LayerBwdArgs& b, int Mp, int kern_kind, int hyp_rows) {
  if (!layer_last_built(Mp, a.D_in, a.D_out) || hyp_rows < layer_last_waves(Mp) || !a.lik_Y || !a.Asave || a.rep != 1 || a.F || b.E ||
      a.Csave || b.Csave || a.d_split > 1 || b.d_split > 1 || b.up_dF ||
      a.mean_kind != DSDGP_MEAN_ZERO || b.mean_kind != DSDGP_MEAN_ZERO) {
    dsdgp_set_error("layer_last: launch outside the fused last layer's scope (internal)");
    return DSDGP_ERR_UNSUPPORTED;
  }
  const bool rbf = kern_kind == DSDGP_KERN_RBF;
  if (Mp == 128) return rbf ? last_go<8, 4, DSDGP_KERN_RBF>(ctx, a, b, hyp_rows) : last_go<8, 4, DSDGP_KERN_MATERN52>(ctx, a, b, hyp_rows);
  return rbf ? last_go<16, 8, DSDGP_KERN_RBF>(ctx, a, b, hyp_rows) : last_go<16, 8, DSDGP_KERN_MATERN52>(ctx, a, b, hyp_rows);
}
