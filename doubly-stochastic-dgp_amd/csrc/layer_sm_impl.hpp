// "Split-M" SVGP_Layer chain (layers.py:178-219 + utils.py:40-41; math identical to layer.hip).
//
// Why: v_mfma_f64_16x16x4_f64 needs >= 2 wavefronts per SIMD to run at its pipe rate (a lone wave issues one MFMA per
// ~59 ns, two waves one per ~45 ns per SIMD — profiles/r01_mfma_f64_microbench.txt), but a 20 000-row layer only has 1250
// sixteen-row blocks for 1024 SIMDs.  Here the NW waves of a workgroup cooperate on ONE block of 16 data rows: wave w owns
// the output row-blocks {w, P-1-w, P+w, 2P-1-w, ...} (P = 2 NW; paired so that triangular products carry equal work), the
// activations live in ONE in-place LDS buffer in MFMA B-operand order ([k][16 rows]: 128-B rows, bank-conflict free), each
// wave streams only ITS weight columns from L2 (no weight element is fetched twice inside a workgroup), and Z/l is read
// through L1.  ~23 KB of LDS and < 100 VGPRs at M = 128 => 4-5 workgroups per CU: latency is hidden by thread-level
// parallelism (explicit prefetching was measured slower because it costs occupancy).
//   NW = 4  : Mp <= 256          NW = 8 : Mp = 512 (64 KB activation buffer)       NW = 16 : Mp = 1024 (128 KB)
// Wide inputs (D_in > 64, e.g. the 784-pixel first MNIST layer) stage x/l through LDS in 64-column chunks.
#pragma once
#include <stdlib.h>

#include "layer.hpp"

#define XCH 64   // columns of x/l staged per chunk

// Operand order of the chain GEMMs — two variants, chosen per instance (measured, tools/ab_kernels.py):
//  * D4 (NW >= 8, i.e. Mp >= 512): every weight matrix is read as plain rows W[i][k]: lane (g, c) of the wave that owns output
//    row-block ib fetches W[16 ib + c][16 kb + 4 g .. + 3] with ONE 32-byte load and feeds its four values to the four MFMAs of
//    the k-block, i.e. MFMA step s contracts k = 16 kb + 4 g + s (any bijection of k inside a 16-block is allowed as long as A
//    and B agree).  The matching B row is 4 g + s, so the activation rows are stored PERMUTED in LDS (row 4 a + b of a block
//    lives at slot 4 b + a): the B read of step s is slot 16 kb + 4 s + g, conflict-free with an immediate offset, and the loop
//    carries one global load and no address arithmetic per four MFMAs (2x on the M = 512 / 1024 chains).
//  * scalar (NW = 4, Mp <= 256): one 8-byte load per MFMA from the transposed matrix W^T[k][i] (k = 16 kb + 4 s + g, natural
//    LDS order): each load instruction covers 4 full cache lines, which wins while the whole weight set is L1/L2-hot.
template <bool D4>
__device__ __forceinline__ int act_slot(int m) {
  return D4 ? ((m & ~15) | ((m & 3) << 2) | ((m >> 2) & 3)) : m;
}
// slot of row (g + 4 t) of block ib — the MFMA D-layout rows this lane holds
template <bool D4>
__device__ __forceinline__ int out_slot(int ib, int g, int t) {
  return D4 ? 16 * ib + 4 * g + t : 16 * ib + g + 4 * t;
}

// acc += W[16 ib + c][16 kb .. 16 kb + 15] . act[16 kb .. 16 kb + 15][c]
//   WR = W as rows [i][k], WT = the same matrix transposed ([k][i]); Mp = leading dimension of both
template <int Mp, bool D4>
__device__ __forceinline__ d4 chain_block(const double* __restrict__ WR, const double* __restrict__ WT,
                                          const double* __restrict__ actb, int ib, int kb, int g, int c, d4 acc) {
  if constexpr (D4) {
    const d4 w4 = *reinterpret_cast<const d4*>(WR + (int64_t)(16 * ib + c) * Mp + 16 * kb + 4 * g);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_f64(w4[s], actb[(16 * kb + 4 * s + g) * 16 + c], acc);
  } else {
    const double* __restrict__ w = WT + (int64_t)(16 * kb + g) * Mp + 16 * ib + c;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_f64(w[(int64_t)(4 * s) * Mp], actb[(16 * kb + 4 * s + g) * 16 + c], acc);
  }
  return acc;
}

// acc += sum_{kb = lo}^{hi - 1} W-block(ib, kb) . act-block(kb): the triangular products of the forward chain (k ranges that depend on
// the wave's row block, i.e. run-time trip counts).  Scalar operand order (Mp <= 256): SOFTWARE-PIPELINED by hand, two k-blocks deep —
// the four weight loads of block kb + 1 are issued before the four MFMAs of block kb.  Left to `#pragma unroll 4` the compiler keeps
// one k-block per iteration (the loop is divergent to it: its bounds depend on threadIdx.x >> 6; "loop not unrolled" warnings) and
// every block is load x 4 -> s_waitcnt -> MFMA x 4: one L1 / L2 round trip in front of every four MFMAs, hidden only by the other waves
// of the SIMD (profiles/r05_fwd_chain_isa.md: 47 TFLOP/s executed in the forward chain at M = 256 against 68 in the backward chain,
// whose loops have compile-time bounds and come out twelve loads deep).
template <int Mp, bool D4>
__device__ __forceinline__ d4 chain_range(const double* __restrict__ WR, const double* __restrict__ WT,
                                          const double* __restrict__ actb, int ib, int lo, int hi, int g, int c, d4 acc) {
  if constexpr (D4) {
#pragma unroll 4
    for (int kb = lo; kb < hi; ++kb) acc = chain_block<Mp, true>(WR, WT, actb, ib, kb, g, c, acc);
    return acc;
  } else {
    if (lo >= hi) return acc;
    // address = uniform row base (SGPR pair: kb, s are wave-uniform) + a 32-bit per-lane offset: the scalar-base form of global_load,
    // no 64-bit VALU address arithmetic and no address registers per load
    const unsigned off = (unsigned)(g * Mp + 16 * ib + c);
    auto wld = [&](int kbx, int s) { return (WT + (size_t)(16 * kbx + 4 * s) * Mp)[off]; };
    double wa[4], wb[4];
    int kb = lo;
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[s] = wld(kb, s);
    while (kb + 1 < hi) {
#pragma unroll
      for (int s = 0; s < 4; ++s) wb[s] = wld(kb + 1, s);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma_f64(wa[s], actb[(16 * kb + 4 * s + g) * 16 + c], acc);
      ++kb;
      if (kb + 1 < hi) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wa[s] = wld(kb + 1, s);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma_f64(wb[s], actb[(16 * kb + 4 * s + g) * 16 + c], acc);
      ++kb;
    }
    if (kb < hi) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma_f64(wa[s], actb[(16 * kb + 4 * s + g) * 16 + c], acc);
    }
    return acc;
  }
}

// The same for the TWO row blocks a wave owns (NQ = 2: ib0 = wave, ib1 = 2 NW - 1 - wave), interleaved: where both k ranges overlap the
// wave runs two independent accumulator chains on ONE activation fragment per k-step (half the LDS reads, eight weight loads in flight
// ahead of eight MFMAs), before that the longer range alone.  UPPER: k ranges [ib, hi) (Lu^-T, q_sqrt^T products); else [0, ib + 1).
template <int Mp, bool UPPER>
__device__ __forceinline__ void chain_range2(const double* __restrict__ WT, const double* __restrict__ actb, int ib0, int ib1, int nkb,
                                             int g, int c, d4& acc0, d4& acc1) {
  // ib0 < ib1.  UPPER: chain 0 covers [ib0, nkb), chain 1 [ib1, nkb): chain 0 alone on [ib0, ib1), both on [ib1, nkb).
  //             else : chain 0 covers [0, ib0], chain 1 [0, ib1]     : both on [0, ib0], chain 1 alone on (ib0, ib1].
  const unsigned off0 = (unsigned)(g * Mp + 16 * ib0 + c), off1 = (unsigned)(g * Mp + 16 * ib1 + c);    // (scalar row base + lane offset)
  auto wld = [&](int kbx, int s, unsigned off) { return (WT + (size_t)(16 * kbx + 4 * s) * Mp)[off]; };
  if constexpr (UPPER) {
    acc0 = chain_range<Mp, false>(nullptr, WT, actb, ib0, ib0, ib1, g, c, acc0);
  }
  const int lo = UPPER ? ib1 : 0, hi = UPPER ? nkb : ib0 + 1;
  if (lo < hi) {
    // ONE register set per chain, refilled in place: the weight of k-step s of block kb + 1 is requested right behind the MFMA that
    // consumed the one of block kb (eight loads in flight per wave; a second register set — two blocks deep — costs 16 VGPRs and with
    // them an occupancy step at M = 128 and M = 256)
    double a0[4], a1[4];
    int kb = lo;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a0[s] = wld(kb, s, off0);
      a1[s] = wld(kb, s, off1);
    }
    for (; kb + 1 < hi; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double bv = actb[(16 * kb + 4 * s + g) * 16 + c];
        acc0 = mfma_f64(a0[s], bv, acc0);
        acc1 = mfma_f64(a1[s], bv, acc1);
        a0[s] = wld(kb + 1, s, off0);
        a1[s] = wld(kb + 1, s, off1);
      }
      // pin the interleave: two MFMAs, then the two loads that refill their weight registers, four times.  Without it the scheduler
      // sinks all eight loads to the end of the block (behind the eighth MFMA) and the next block starts by waiting for them.
      // (A full scheduling barrier per step with the activation fragment requested one step ahead — the backward d-loop's form —
      // measured slower here: +2 registers per chain, config 2 0.542 -> 0.551 ms per step.)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double bv = actb[(16 * kb + 4 * s + g) * 16 + c];
      acc0 = mfma_f64(a0[s], bv, acc0);
      acc1 = mfma_f64(a1[s], bv, acc1);
    }
  }
  if constexpr (!UPPER) {
    acc1 = chain_range<Mp, false>(nullptr, WT, actb, ib1, ib0 + 1, ib1 + 1, g, c, acc1);
  }
}

// Dense product for the two row blocks of a wave (k-blocks 0 .. MPB - 1 of both, compile-time trip count): the paired form of
// chain_range2 without the triangular heads / tails.
template <int Mp, int MPB>
__device__ __forceinline__ void chain_dense2(const double* __restrict__ WT, const double* __restrict__ actb, int ib0, int ib1, int g, int c,
                                             d4& acc0, d4& acc1) {
  const unsigned off0 = (unsigned)(g * Mp + 16 * ib0 + c), off1 = (unsigned)(g * Mp + 16 * ib1 + c);
  auto wld = [&](int kbx, int s, unsigned off) { return (WT + (size_t)(16 * kbx + 4 * s) * Mp)[off]; };
  double a0[4], a1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a0[s] = wld(0, s, off0);
    a1[s] = wld(0, s, off1);
  }
  double bv = actb[g * 16 + c];
#pragma unroll
  for (int kb = 0; kb < MPB; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kn = (s == 3) ? (kb + 1) % MPB : kb, sn = (s + 1) & 3;
      const double bvn = actb[(16 * kn + 4 * sn + g) * 16 + c];
      acc0 = mfma_f64(a0[s], bv, acc0);
      acc1 = mfma_f64(a1[s], bv, acc1);
      if (kb + 1 < MPB) {
        a0[s] = wld(kb + 1, s, off0);
        a1[s] = wld(kb + 1, s, off1);
      }
      bv = bvn;
      if (kb + 1 < MPB) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int MPB, int NW>
struct Own {
  // NQ row blocks per wave; MPB need not be a multiple of NW (inducing counts padded to 16 / 32 / 64 / 128 instead of powers of
  // two): the tail slots of some waves then point past the last block and are skipped (skip(ib)); the test folds away when
  // MPB % NW == 0.
  static constexpr int NQ = (MPB + NW - 1) / NW;
  static __device__ __forceinline__ int ib(int wave, int q) {
    if (NQ >= 2) return (q & 1) ? (q >> 1) * (2 * NW) + (2 * NW - 1) - wave : (q >> 1) * (2 * NW) + wave;
    return wave;
  }
  static __device__ __forceinline__ bool skip(int ib) { return (MPB % NW != 0) && ib >= MPB; }
  static __device__ __forceinline__ bool active(int wave) { return MPB >= NW || wave < MPB; }
};

// Row-block ownership of the Csave backward d-loop (abar += q_sqrt_d cbar_d, a TRIANGULAR product: row block ib costs ib + 1
// k-blocks).  8-wave workgroups at Mp = 128 / 256 split into four LIGHT waves, which also stage cbar_d from global memory into
// LDS, and four HEAVY waves, which never issue those loads: vmcnt retires loads in order, so a wave that has the (HBM-latency)
// cbar_{d+1} loads in flight stalls at its first (L2-latency) weight fetch — with every wave staging, each output paid the full
// HBM latency (measured: 67 % of the wave cycles parked, backward chain 1.45x SLOWER than the dense S_d form).  Waves w and w + 4
// share a SIMD: light wave s + heavy wave 4 + s carry the same MFMA count on every SIMD (9 blocks at Mp = 128, 34 at Mp = 256).
template <int MPB, int NW>
struct OwnCS {
  static constexpr bool CUSTOM = (NW == 8) && (MPB == 8 || MPB == 16);
  static constexpr int NQ = CUSTOM ? MPB / 8 : Own<MPB, NW>::NQ;
  static constexpr int NLOAD = CUSTOM ? 4 : NW;      // waves that stage cbar_d
  static __device__ __forceinline__ int ib(int wave, int q) {
    if constexpr (!CUSTOM) return Own<MPB, NW>::ib(wave, q);
    if constexpr (MPB == 8) return wave < 4 ? wave : 11 - wave;                         // light: 0..3 ; heavy: 7, 6, 5, 4
    if (wave < 4) return q == 0 ? 7 - wave : wave;                                      // light: (7 - s, s)
    return q == 0 ? 19 - wave : 4 + wave;                                               // heavy: (15 - s, 8 + s), s = wave - 4
  }
  static __device__ __forceinline__ bool active(int wave) { return CUSTOM || Own<MPB, NW>::active(wave); }
};

// outputs per epilogue group of the forward chain: their variance / mean partials are parked in LDS, ONE barrier per group,
// then all threads write the group's mean / var / F as contiguous runs (the per-output form paid a barrier and 16 scattered
// 8-byte stores per output).  Double-buffered by group parity.  NW = 16 keeps one output per group (LDS is full at M = 1024).
static inline constexpr int sm_db(int NW) { return NW == 4 ? 8 : (NW == 8 ? 4 : 1); }

// LDS carve (doubles): xs | act | red
struct SmLds {
  int xs, act, red, total;
};
static inline SmLds sm_lds(int Mp, int D_in, int D_out, int NW, bool wide, int nbuf = 1) {
  SmLds L;
  int o = 0;
  const int xch = D_in < XCH ? D_in : XCH;
  L.xs = o; o += 16 * (xch + 1);
  o = (int)round_up(o, 2);
  L.act = o; o += Mp * 16;
  L.red = o;                        // nbuf = 2 (Csave backward chain, Mp <= 256): the second staging buffer; it is free again when the
                                    // epilogue needs `red`, so the two share the space
  const bool mu_early = (NW == 4) && !wide;
  const int db = sm_db(NW);
  const int red_fwd = NW * 16 + 2 * db * NW * 16 + (mu_early ? NW * 16 * D_out : 2 * db * NW * 16);   // s1 | 2 x [db] s2 | mean partials
  const int red_bwd = NW * 16 * xch;                            // dX partials of one chunk
  int red = red_fwd > red_bwd ? red_fwd : red_bwd;
  if (nbuf == 2 && red < Mp * 16) red = Mp * 16;
  o += red;
  L.total = o;
  return L;
}

// One staged chunk (jn columns of x / l in xs, row stride jn + 1; Z columns j0 .. j0 + jn) of the MFMA distance form below, any jn:
// clamped (unconditional) loads, masked afterwards.
template <int NQ, int MPB, int NW>
__device__ __forceinline__ void sqdist_chunk_masked(const double* __restrict__ zs, const double* xs, int Din, int j0, int jn, int wave,
                                                    int g, int c, d4 (&zx)[NQ], double (&zsq)[NQ], double& xx) {
  for (int kk = 0; kk < jn; kk += 16) {
    // dimension kk + 4 s + g in k-step s (round 6; 4 g + s before): a group with fewer than 16 dimensions left — D_in = 8 is two k-steps —
    // issues only the k-steps that carry data.  With 4 g + s every step of a D_in <= 8 layer was half zeros: four MFMAs and four Z loads
    // per row block where two do.  (Any bijection of the 16 dimensions onto (g, s) is allowed as long as A and B agree.)
    const int ns = (jn - kk >= 16) ? 4 : (jn - kk + 3) >> 2;           // wave-uniform
    double b[4];
    int jc[4];
    bool in[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = kk + 4 * s + g;
      in[s] = j < jn;
      jc[s] = in[s] ? j : jn - 1;
      const double v = xs[c * (jn + 1) + jc[s]];
      b[s] = in[s] ? v : 0.0;
      xx = fma(b[s], b[s], xx);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB, NW>::ib(wave, q);
      if (Own<MPB, NW>::skip(ib)) continue;
      const double* __restrict__ zr = zs + (int64_t)(16 * ib + c) * Din + j0;
      double av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < ns) {
          const double v = zr[jc[s]];
          av[s] = in[s] ? v : 0.0;
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < ns) {
          zx[q] = mfma_f64(av[s], b[s], zx[q]);
          zsq[q] = fma(av[s], av[s], zsq[q]);
        }
      }
    }
  }
}
// |z|^2 (per lane for row 16 ib + c) folded over g and turned into the accumulator layout through wave-private LDS; r2 >= 0
template <int NQ>
__device__ __forceinline__ void sqdist_finish(const d4 (&zx)[NQ], const double (&zsq)[NQ], double xx, double* scratch, int wave, int g,
                                              int c, d4 (&r2)[NQ]) {
  xx = sum_groups(xx);          // lane (g, c) covered the dimensions g, 4 + g, 8 + g, 12 + g of every group of 16 (4 g .. 4 g + 3 in the wide path)
  double* zrow = scratch + wave * NQ * 16;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double zq = sum_groups(zsq[q]);          // |z|^2 of row 16 ib + c
    if (g == 0) zrow[q * 16 + c] = zq;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double v = zrow[q * 16 + g + 4 * t] + xx - 2.0 * zx[q][t];
      r2[q][t] = v > 0.0 ? v : 0.0;
    }
}
// narrow inputs (D_in <= XCH, x / l already staged in xs by the caller, barrier passed): the same form in one chunk
template <int NQ, int MPB, int NW>
__device__ __forceinline__ void sqdist_narrow(const double* __restrict__ zs, const double* xs, int Din, int wave, int g, int c, bool act,
                                              double* scratch, d4 (&r2)[NQ]) {
  d4 zx[NQ];
  double zsq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { zx[q] = (d4){0, 0, 0, 0}; zsq[q] = 0.0; }
  double xx = 0.0;
  if (act) sqdist_chunk_masked<NQ, MPB, NW>(zs, xs, Din, 0, Din, wave, g, c, zx, zsq, xx);
  sqdist_finish<NQ>(zx, zsq, xx, scratch, wave, g, c, r2);
}

// scaled squared distances of this wave's inducing rows to the block's 16 data rows, D layout, chunked over D_in (wide inputs).
// r2[m][c] = |z_m|^2 + |x_c|^2 - 2 z_m . x_c on the MFMA pipe: lane (g, c) feeds A = Zs[16 ib + c][k], B = xs[c][k] with
// k = 16-dimension group + 4 g + s in k-step s (any bijection of k works as long as A and B agree), so the product lands in the
// D layout (row g + 4 t of block ib, column c) the chains want; |z_m|^2 is summed per lane for row 16 ib + c, folded over g and
// turned from that A-operand order into the D layout through 16 doubles of LDS per row block (`scratch`, wave-private: same wave
// writes and reads, LDS operations of a wave execute in order); |x_c|^2 from a per-lane sum folded over g.  The element-by-element form (one L2 load and one FMA per (m, c, dimension) and lane)
// took 1.37 M of the 3.25 M clocks of the 784-dimensional first layer of config 4 (DSDGP_FWD_TIMING; the current clocks are in profiles/r02_chain_phases.txt) and as much again
// in the backward chain.  Absolute error ~ 1e-16 (|z|^2 + |x|^2); the result is clamped at 0.
template <int NQ, int MPB, int NW>
__device__ __forceinline__ void sm_sqdist(const double* __restrict__ zs, const double* __restrict__ X,
                                          const double* __restrict__ ils, double* xs, int Din, int64_t r0, int64_t Rin,
                                          int tid, int wave, int g, int c, bool act, double* scratch, d4 (&r2)[NQ]) {
  d4 zx[NQ];
  double zsq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { zx[q] = (d4){0, 0, 0, 0}; zsq[q] = 0.0; }
  double xx = 0.0;
  for (int j0 = 0; j0 < Din; j0 += XCH) {
    const int jn = (Din - j0 < XCH) ? Din - j0 : XCH;
    if (j0 > 0) __syncthreads();
    for (int idx = tid; idx < 16 * jn; idx += NW * 64) {
      const int rr = idx / jn, j = idx % jn;
      int64_t row = r0 + rr;
      if (row > Rin - 1) row = Rin - 1;
      xs[rr * (jn + 1) + j] = X[row * Din + j0 + j] * ils[j0 + j];
    }
    __syncthreads();
    if (act) {
      if ((jn & 15) == 0 && (Din & 3) == 0) {
        // whole groups of 16 dimensions: one 32-byte load per lane and row block, all groups of the chunk unrolled so that their
        // loads are in flight together (the wide instances run one workgroup per CU: registers are plentiful)
#pragma unroll
        for (int u = 0; u < XCH / 16; ++u) {
          const int kk = 16 * u;
          if (kk >= jn) break;
          double b[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            b[s] = xs[c * (jn + 1) + kk + 4 * g + s];
            xx = fma(b[s], b[s], xx);
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ib = Own<MPB, NW>::ib(wave, q);
            if (Own<MPB, NW>::skip(ib)) continue;
            const d4 av = *reinterpret_cast<const d4*>(zs + (int64_t)(16 * ib + c) * Din + j0 + kk + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              zx[q] = mfma_f64(av[s], b[s], zx[q]);
              zsq[q] = fma(av[s], av[s], zsq[q]);
            }
          }
        }
      } else {
        sqdist_chunk_masked<NQ, MPB, NW>(zs, xs, Din, j0, jn, wave, g, c, zx, zsq, xx);
      }
    }
  }
  sqdist_finish<NQ>(zx, zsq, xx, scratch, wave, g, c, r2);
}

// debug aid of the phase-clock launches.  The stamps are s_memrealtime (constant 100 MHz, one time base for the whole chip): s_memtime
// counts shader clocks but its base differs from CU to CU (tools/xcd_probe.hip: spreads of 5e7 clocks inside one XCD), so starts and
// ends of different workgroups could not be compared.  Reports the launch span, every workgroup's start delay and own duration, and
// the workgroups that start late (after a quarter of the span: a second round).
static inline void phase_span_report(const std::vector<unsigned long long>& h, int nwg, int last_slot) {
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int w = 0; w < nwg; ++w) {
    t0 = std::min(t0, h[(size_t)w * 8]);
    t1 = std::max(t1, h[(size_t)w * 8 + last_slot]);
  }
  const double span = (double)(t1 - t0) * 0.01;
  double dur = 0, start = 0, start_max = 0, dur_max = 0, late = 0;
  int nlate = 0;
  for (int w = 0; w < nwg; ++w) {
    const double s = (double)(h[(size_t)w * 8] - t0) * 0.01, d = (double)(h[(size_t)w * 8 + last_slot] - h[(size_t)w * 8]) * 0.01;
    dur += d; start += s;
    start_max = std::max(start_max, s); dur_max = std::max(dur_max, d);
    if (s > 0.25 * span) { ++nlate; late += d; }
  }
  fprintf(stderr, "[span] %d wgs: launch %.1f us | wg duration avg %.1f max %.1f us | start avg %.1f max %.1f us | %d late starters (avg duration %.1f us)\n",
          nwg, span, dur / nwg, dur_max, start / nwg, start_max, nlate, nlate ? late / nlate : 0.0);
  if (nwg >= 512) {      // mean duration by launch slot (w >> 8: the order in which a CU received its workgroups) and by XCD (w & 7)
    double ds[8] = {0}, dx[8] = {0};
    int ns[8] = {0}, nx[8] = {0};
    for (int w = 0; w < nwg; ++w) {
      const double d = (double)(h[(size_t)w * 8 + last_slot] - h[(size_t)w * 8]) * 0.01;
      const int k = std::min(w >> 8, 7);
      ds[k] += d; ++ns[k]; dx[w & 7] += d; ++nx[w & 7];
    }
    fprintf(stderr, "[span]   mean duration by slot w >> 8:");
    for (int k = 0; k < 8 && ns[k]; ++k) fprintf(stderr, " %.1f", ds[k] / ns[k]);
    fprintf(stderr, " | by XCD w & 7:");
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", dx[x] / std::max(nx[x], 1));
    fprintf(stderr, "\n");
  }
}
#define FWD_STAMP(i) do { if (a.phase_clk && tid == 0 && blockIdx.y == 0) a.phase_clk[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// occupancy the forward instances are held to.  The 8-wave M = 256 instance sits at 120-126 VGPRs = two workgroups per CU; unrelated
// edits (a debug stamp) moved it to 131 = ONE workgroup per CU (config 3: forward chains 16.3 -> 20.1 ms per 5 steps) — pinned at 128.
template <int MPB, int NW, bool WIDE, bool LIK>
constexpr int fwd_min_waves() { return (LIK && NW == 4) ? 5 : ((NW == 8 && MPB == 16 && !WIDE && !LIK) ? 4 : 1); }
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE, bool LIK>
__global__ __launch_bounds__(NW * 64, (fwd_min_waves<MPB, NW, WIDE, LIK>())) void k_layer_fwd_sm(const LayerFwdArgs a, const SmLds L) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16, NQ = Own<MPB, NW>::NQ;
  constexpr bool D4 = (MPB > 16);
  constexpr bool ILV = D4 || (MPB >= 16);   // interleave the NQ row-block chains inside the k loop (measured: helps from NQ = 4)
  const int Din = a.D_in, Dout = a.D_out;
  const int qld = a.qmu_ld ? a.qmu_ld : Dout;
  const int tid = threadIdx.x, lane = tid & 63, wave = DS_WAVE_ID(tid);
  const int g = lane >> 4, c = lane & 15;
  double* xs = smem + L.xs;
  double* actb = smem + L.act;                            // single in-place activation buffer ([k][16 rows])
  double* red_s1 = smem + L.red;                          // [NW][16]
  constexpr int DB = sm_db(NW);
  double* red_s2 = red_s1 + NW * 16;                      // [2][DB][NW][16]
  double* red_mu = red_s2 + 2 * DB * NW * 16;             // MU_EARLY: [NW][Dout][16] ; else [2][DB][NW][16]
  constexpr bool MU_EARLY = (NW == 4) && !WIDE;           // small-M kernels: all mean partials before the q_sqrt loop
  const double* ils = a.hyp + HYP_ILS;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const bool act = Own<MPB, NW>::active(wave);
  const double s2 = a.hyp[HYP_VAR];
  FWD_STAMP(0);

  // [X^T ; 1] of this block for the Z-gradient product of the backward pass (coalesced 128-byte runs along the rows)
  if (a.XT1 && blockIdx.y == 0) {
    for (int idx = tid; idx < 16 * (Din + 1); idx += NW * 64) {
      const int j = idx >> 4, rr = idx & 15;
      const int64_t r = r0 + rr;
      if (r < a.ldA) a.XT1[(int64_t)j * a.ldA + r] = (r < a.Rin) ? (j < Din ? a.X[r * Din + j] : 1.0) : 0.0;
    }
  }
  // --- Kuf tile (layers.py:184): own row-blocks -> act
  if constexpr (!WIDE) {
    // D_in <= XCH: one staging pass, distances accumulated element by element (fewest live registers)
    for (int idx = tid; idx < 16 * Din; idx += NW * 64) {
      const int rr = idx / Din, j = idx % Din;
      int64_t row = r0 + rr;
      if (row > a.Rin - 1) row = a.Rin - 1;
      xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
    }
    __syncthreads();
    // distances on the MFMA pipe (|z|^2 + |x|^2 - 2 z.x, see sm_sqdist): the element-by-element form — one L1 / L2 load and one FMA per
    // (inducing row, data row, dimension) and lane — was 17 K of the 61 K clocks of a D_out = 1 workgroup at config 2
    d4 r2[NQ];
    sqdist_narrow<NQ, MPB, NW>(a.Zs, xs, Din, wave, g, c, act, red_s1, r2);
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = 16 * ib + g + 4 * t;
          const double kv = kern_val<KIND>(r2[q][t], s2);      // unconditional: behind `m < M` every exp sat in its own exec branch
          actb[act_slot<D4>(m) * 16 + c] = (m < a.M) ? kv : 0.0;
        }
      }
    }
  } else {
    d4 r2[NQ];
    sm_sqdist<NQ, MPB, NW>(a.Zs, a.X, ils, xs, Din, r0, a.Rin, tid, wave, g, c, act, red_s1, r2);
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = 16 * ib + g + 4 * t;
          const double kv = kern_val<KIND>(r2[q][t], s2);      // unconditional: behind `m < M` every exp sat in its own exec branch
          actb[act_slot<D4>(m) * 16 + c] = (m < a.M) ? kv : 0.0;
        }
      }
    }
  }
  __syncthreads();
  FWD_STAMP(1);      // Kuf tile in LDS

  d4 acc[NQ];
  // --- a1 = Lu^{-1} k (layers.py:186): out block ib sums k-blocks kb <= ib ; weights LinvT[k][i]
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = (d4){0, 0, 0, 0};
  // two row blocks per wave, both real, scalar operand order: the interleaved pair (chain_range2); else one chain after the other
  constexpr bool PAIR = (NQ == 2) && !D4 && (MPB % NW == 0);
  if (act) {
    if constexpr (PAIR) {
      chain_range2<Mp, false>(a.LinvT, actb, Own<MPB, NW>::ib(wave, 0), Own<MPB, NW>::ib(wave, 1), MPB, g, c, acc[0], acc[1]);
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
        acc[q] = chain_range<Mp, D4>(a.Linv, a.LinvT, actb, ib, 0, ib + 1, g, c, acc[q]);
      }
    }
  }
  {
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) p = fma(acc[q][t], acc[q][t], p);
    p = sum_groups(p);
    if (g == 0) red_s1[wave * 16 + c] = act ? p : 0.0;
  }
  __syncthreads();   // every wave has finished reading k before a1 overwrites it in place
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB, NW>::ib(wave, q);
      if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) actb[out_slot<D4>(ib, g, t) * 16 + c] = acc[q][t];
    }
  }
  __syncthreads();
  FWD_STAMP(2);      // a1, |a1|^2
  // --- a = Lu^{-T} a1 (layers.py:188): out block ib sums kb >= ib ; weights Linv[k][i]   (white: a = a1)
  if (!WHITE) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = (d4){0, 0, 0, 0};
    if (act) {
      if constexpr (PAIR) {
        chain_range2<Mp, true>(a.Linv, actb, Own<MPB, NW>::ib(wave, 0), Own<MPB, NW>::ib(wave, 1), MPB, g, c, acc[0], acc[1]);
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
          acc[q] = chain_range<Mp, D4>(a.LinvT, a.Linv, actb, ib, ib, MPB, g, c, acc[q]);
        }
      }
    }
    __syncthreads();   // a1 fully consumed -> overwrite with a
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) actb[out_slot<D4>(ib, g, t) * 16 + c] = acc[q][t];
      }
    }
  }
  // acc holds this wave's rows of "a" (saved for the backward pass at the END of the kernel, see there); the partial mean
  // a . q_mu (layers.py:190)
  FWD_STAMP(3);      // a
  if constexpr (MU_EARLY) {
    // mean partials mu[d][c] = sum_m q_mu[m][d] a[m][c] over this wave's rows on the MFMA pipe, 16 outputs at a time: A = q_mu^T
    // (lane (g, c) loads row 16 ib + g + 4 t, output c), B = a in the accumulator layout.  The per-output scalar form (eight loads,
    // eight FMAs and a cross-lane fold per output) was 16 K of the 169 K clocks of a D_out = 8 workgroup at config 2.
    for (int d0 = 0; d0 < Dout; d0 += 16) {
      d4 mu4 = (d4){0, 0, 0, 0};
      if (act) {
        const bool din = d0 + c < Dout;
        const int dc = din ? d0 + c : Dout - 1;                  // clamped (unconditional) loads, masked afterwards
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
          double qv[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) qv[t] = a.qmu[(int64_t)(16 * ib + g + 4 * t) * qld + dc];       // layers.py:190
#pragma unroll
          for (int t = 0; t < 4; ++t) mu4 = mfma_f64(din ? qv[t] : 0.0, acc[q][t], mu4);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int d = d0 + g + 4 * t;
        if (d < Dout) red_mu[(wave * Dout + d) * 16 + c] = mu4[t];
      }
    }
  }
  __syncthreads();
  FWD_STAMP(4);      // mean partials

  const double kdiag = a.hyp[HYP_KDIAG];
  const double lik_s2 = LIK ? a.lik_const[0] : 1.0;
  const double lik_c0 = -0.91893853320467274178 - 0.5 * log(lik_s2);
  double lik_ve = 0.0, lik_dl = 0.0;
  // small launches (the N-row first layer) spread their D_out products over gridDim.y workgroups per row block
  const int dchunk = (Dout + (int)gridDim.y - 1) / (int)gridDim.y;
  const int d_lo = (int)blockIdx.y * dchunk, d_hi = (d_lo + dchunk < Dout) ? d_lo + dchunk : Dout;
  for (int d0 = d_lo, grp = 0; d0 < d_hi; d0 += DB, ++grp) {
    const int gs = (d_hi - d0 < DB) ? d_hi - d0 : DB;
    double* rs2 = red_s2 + (grp & 1) * DB * NW * 16;
    double* rmu_g = red_mu + (grp & 1) * DB * NW * 16;     // !MU_EARLY only
    for (int dd = 0; dd < gs; ++dd) {
      const int d = d0 + dd;
      // --- c_d = q_sqrt_d^T a ; |c_d|^2 (replaces SK/B of layers.py:195-212): out block ib sums kb >= ib
      d4 cacc[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) cacc[q] = (d4){0, 0, 0, 0};
      if (act) {
        const double* __restrict__ TdT = a.TpT + (int64_t)d * Mp * Mp;
        const double* __restrict__ Td = a.Tp + (int64_t)d * Mp * Mp;
        if constexpr (PAIR) {
          chain_range2<Mp, true>(Td, actb, Own<MPB, NW>::ib(wave, 0), Own<MPB, NW>::ib(wave, 1), MPB, g, c, cacc[0], cacc[1]);
        } else {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ib = Own<MPB, NW>::ib(wave, q);
            if (Own<MPB, NW>::skip(ib)) continue;
            cacc[q] = chain_range<Mp, D4>(TdT, Td, actb, ib, ib, MPB, g, c, cacc[q]);
          }
        }
      }
      double p = 0.0, mu = 0.0;
      if (act) {
        if (a.Csave) {
          // c_d for the backward chain, BLOCK-major: the (Mp x 16) tile of (row block, output d) is one contiguous 16*Mp*8-byte
          // run stored in LDS slot order, so this store and the backward staging are plain streaming copies (an M-major
          // layout like Asave put every 128-byte run on its own page: TLB- and DRAM-page-hostile at D_out*Mp runs per block)
          const int64_t r = r0 + c;
          double* __restrict__ Cd = a.Csave + ((int64_t)blockIdx.x * Dout + d) * (Mp * 16) + c;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ib = Own<MPB, NW>::ib(wave, q);
            if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const double cv = (r < a.Rin) ? cacc[q][t] : 0.0;
              Cd[out_slot<D4>(ib, g, t) * 16] = cv;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            p = fma(cacc[q][t], cacc[q][t], p);
            if constexpr (!MU_EARLY) mu = fma(acc[q][t], a.qmu[(int64_t)(16 * ib + g + 4 * t) * qld + d], mu);   // layers.py:190
          }
        }
      }
      p = sum_groups(p);
      if constexpr (!MU_EARLY) mu = sum_groups(mu);
      if (g == 0) {
        rs2[(dd * NW + wave) * 16 + c] = p;
        if constexpr (!MU_EARLY) rmu_g[(dd * NW + wave) * 16 + c] = mu;
      }
    }
    __syncthreads();
    // epilogue of the group: thread e <-> (sample s, row cc, output dd); a row's gs outputs are contiguous in mean / var / F.
    // The first layer writes `rep` = S output rows per input row: those are spread over the threads too (one thread walking
    // its S rows serially — z load, three stores each — was a third of that latency-bound launch); a Linear mean function keeps
    // the per-row form (its D_in-long dot product is not worth repeating per sample).
    const bool flat = a.rep > 1 && a.mean_kind != DSDGP_MEAN_LINEAR;
    const int n_items = 16 * gs * (flat ? a.rep : 1);
    for (int e = tid; e < n_items; e += NW * 64) {
      const int s0 = flat ? e / (16 * gs) : 0, e2 = e % (16 * gs);
      const int cc = e2 / gs, dd = e2 % gs, d = d0 + dd;
      const int64_t r = r0 + cc;
      if (r >= a.Rin) {
        if (LIK && r < a.lik_ld && s0 == 0) {          // rows of the 16-row padding of the transposed adjoints
          a.lik_MB[(int64_t)d * a.lik_ld + r] = 0.0;
          a.lik_VB[(int64_t)d * a.lik_ld + r] = 0.0;
        }
        continue;
      }
      double s1 = 0.0, s2sum = 0.0, mu = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s1 += red_s1[w * 16 + cc];
        s2sum += rs2[(dd * NW + w) * 16 + cc];
        mu += MU_EARLY ? red_mu[(w * Dout + d) * 16 + cc] : rmu_g[(dd * NW + w) * 16 + cc];
      }
      const double var = kdiag - s1 + s2sum;                               // layers.py:212-217
      if (a.mean_kind == DSDGP_MEAN_IDENTITY) {                            // layers.py:219
        mu += a.X[r * Din + d];
      } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
        double m2 = 0.0;
        for (int j = 0; j < Din; ++j) m2 = fma(a.X[r * Din + j], a.mean_A[(int64_t)j * Dout + d], m2);
        mu += m2 + (a.mean_b ? a.mean_b[d] : 0.0);
      }
      const double sd = sqrt(var + a.jitter);
      const int s_lo = flat ? s0 : 0, s_hi = flat ? s0 + 1 : a.rep;
      for (int s = s_lo; s < s_hi; ++s) {
        const int64_t orow = (int64_t)s * a.Rin + r;
        const int64_t o = orow * Dout + d;
        if (a.mean) a.mean[o] = mu;
        if (a.var) a.var[o] = var;
        if (a.F && a.z) {
          const double zv = a.z[(orow / a.n_inner) * a.zs_s + (orow % a.n_inner) * a.zs_n + d * a.zs_d];
          a.F[o] = mu + zv * sd;                                           // utils.py:41 (no clamp)
        }
        if constexpr (LIK) {
          const double y = a.lik_Y[(orow % a.n_inner) * Dout + d];
          const double q = (y - mu) * (y - mu) + var;
          lik_ve += lik_c0 - 0.5 * q / lik_s2;
          lik_dl += -0.5 / lik_s2 + 0.5 * q / (lik_s2 * lik_s2);
          a.lik_MB[(int64_t)d * a.lik_ld + orow] = -a.lik_w * (y - mu) / lik_s2;
          a.lik_VB[(int64_t)d * a.lik_ld + orow] = 0.5 * a.lik_w / lik_s2;
        }
      }
    }
  }
  if constexpr (LIK) {     // this workgroup's share of the variational expectations (fixed-order sums: waves, then the NW wave totals)
    lik_ve = sum_wave(lik_ve);
    lik_dl = sum_wave(lik_dl);
    __syncthreads();
    if (lane == 0) {
      red_s1[2 * wave] = lik_ve;
      red_s1[2 * wave + 1] = lik_dl;
    }
    __syncthreads();
    if (tid == 0) {
      double sv = 0.0, sd2 = 0.0;
      for (int w = 0; w < NW; ++w) {
        sv += red_s1[2 * w];
        sd2 += red_s1[2 * w + 1];
      }
      const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
      a.lik_part[2 * wg] = sv;
      a.lik_part[2 * wg + 1] = sd2;
    }
  }
  FWD_STAMP(5);      // per-output products and epilogue
  // "a" for the backward pass.  Stored LAST: vmcnt retires in order on gfx9, so a store issued before the per-output loop made that
  // loop's first weight loads wait for the store's write acknowledgement.  The tile is re-read from the activation buffer (this
  // thread's own slots, intact since the second chain) — no register stays live for it.
  if (act && a.Asave && blockIdx.y == 0) {
    const int64_t r = r0 + c;
    if (r < a.ldA) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          a.Asave[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r] = (r < a.Rin) ? actb[out_slot<D4>(ib, g, t) * 16 + c] : 0.0;
      }
    }
  }
  FWD_STAMP(6);
}

// ------------------------------------------------------------------------------------------------------
// backward (math: see layer.hip).  hyp_part gets one partial row per WAVE: index (blockIdx.x * NW + wave).
// ------------------------------------------------------------------------------------------------------
// CS: abar's variance part from the saved c_d (triangular q_sqrt_d products, staged through LDS) instead of dense S_d a
#define BWD_STAMP(i) do { if (a.phase_clk && tid == 0) a.phase_clk[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// occupancy the 8-wave instances are held to: 80 VGPRs = three workgroups per CU at Mp <= 128, 128 VGPRs = two at Mp <= 256 (the
// allocator lands one to three registers above those steps otherwise, which halves the resident workgroups)
template <int MPB, int NW, bool WIDE, bool CS>
constexpr int bwd_min_waves() { return (NW == 8 && !WIDE && !CS) ? (MPB <= 8 ? 6 : 4) : ((NW == 4 && MPB == 8 && !WIDE && !CS) ? 5 : 1); }
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE, bool CS>
__global__ __launch_bounds__(NW * 64, (bwd_min_waves<MPB, NW, WIDE, CS>())) void k_layer_bwd_sm(const LayerBwdArgs a, const SmLds L) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int Mp = MPB * 16, NQ = Own<MPB, NW>::NQ;
  constexpr bool D4 = (MPB > 16);
  constexpr bool ILV = D4 || (MPB >= 16);   // interleave the NQ row-block chains inside the k loop (measured: helps from NQ = 4)
  const int Din = a.D_in, Dout = a.D_out;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const double* __restrict__ zs = a.Zs;
  double* xs = smem + L.xs;
  double* actb = smem + L.act;
  double* redx = smem + L.red;                            // [NW][chunk][16]
  const double* ils = a.hyp + HYP_ILS;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const bool act = Own<MPB, NW>::active(wave);
  const double s2 = a.hyp[HYP_VAR];
  const int64_t r = r0 + c;
  const bool rin = r < a.ldA, rvalid = r < a.Rin;
  BWD_STAMP(0);
  if (a.up_dF) {
    // upstream adjoints of this row block from the adjoint of the next layer's input (see LayerBwdArgs::up_dF).  (row, output) pairs
    // x C chunks of the rep samples over the threads, partials through the (still unused) reduction scratch, fixed summation order.
    const int npair = 16 * Dout, T = NW * 64;
    const int cap = (L.total - L.red) / (2 * npair);
    int C = T / npair;
    if (C > a.up_rep) C = a.up_rep;
    if (C > cap) C = cap;
    if (C < 1) C = 1;
    for (int e = tid; e < npair * C; e += T) {
      const int pair = e % npair, ch = e / npair;
      const int rr = pair / Dout, d = pair - rr * Dout;
      const int64_t rw = r0 + rr;
      double ms = 0.0, vs = 0.0;
      if (rw < a.Rin) {
        const int s_lo = ch * a.up_rep / C, s_hi = (ch + 1) * a.up_rep / C;
        for (int s = s_lo; s < s_hi; ++s) {
          const int64_t orow = (int64_t)s * a.Rin + rw;
          const double f = a.up_dF[orow * a.up_ld + a.up_off + d];
          ms += f;
          vs = fma(f, a.up_z[(orow / a.up_n_inner) * a.up_zs + (orow % a.up_n_inner) * a.up_zn + d * a.up_zd], vs);
        }
      }
      redx[2 * e] = ms;
      redx[2 * e + 1] = vs;
    }
    __syncthreads();
    for (int pair = tid; pair < npair; pair += T) {
      const int rr = pair / Dout, d = pair - rr * Dout;
      const int64_t rw = r0 + rr;
      double ms = 0.0, vs = 0.0;
      for (int ch = 0; ch < C; ++ch) {
        ms += redx[2 * (ch * npair + pair)];
        vs += redx[2 * (ch * npair + pair) + 1];
      }
      if (rw < a.Rin) {
        a.MBw[(int64_t)d * a.ldA + rw] = ms;
        a.VBw[(int64_t)d * a.ldA + rw] = vs * 0.5 * rsqrt(a.up_var[rw * Dout + d] + a.up_jitter);
      } else if (rw < a.ldA) {
        a.MBw[(int64_t)d * a.ldA + rw] = 0.0;
        a.VBw[(int64_t)d * a.ldA + rw] = 0.0;
      }
    }
    __syncthreads();       // (waits for the stores; the lines of this row block are read below by this workgroup only)
  }
  d4 av[NQ], acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ib = Own<MPB, NW>::ib(wave, q);
    acc[q] = av[q] = (d4){0, 0, 0, 0};
    if (Own<MPB, NW>::skip(ib)) continue;
    // unconditional (clamped) loads, masked afterwards: behind `act && rin` every load sat in its own exec branch with an
    // s_waitcnt vmcnt(0) in front of the next one — eight serialised round trips at the head of every workgroup
    const bool live = act && rin;
    const double* __restrict__ ap = a.Asave + (int64_t)(16 * (act ? ib : 0) + g) * a.ldA + (rin ? r : 0);
    double ld[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ld[t] = ap[(int64_t)(4 * t) * a.ldA];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double v = live ? ld[t] : 0.0;
      av[q][t] = v;
      if (!CS && act) actb[out_slot<D4>(ib, g, t) * 16 + c] = v;
    }
  }
  // x / l of this row block (needed for the kernel recompute after the chains): requested now so that its latency hides behind them.
  // xs is not touched by anything in between and every path below passes a barrier before reading it.
  if constexpr (!WIDE) {
    for (int idx = tid; idx < 16 * Din; idx += NW * 64) {
      const int rr = idx / Din, j = idx % Din;
      int64_t row = r0 + rr;
      if (row > a.Rin - 1) row = a.Rin - 1;
      xs[rr * (Din + 1) + j] = a.X[row * Din + j] * ils[j];
    }
  }
  // the first two k-steps of the mean part's upstream adjoints, for the same reason
  const double bv0l = a.MB[(int64_t)g * a.ldA + (rin ? r : 0)];
  const double bv1l = a.MB[(int64_t)(a.DP4 > 4 ? 4 + g : g) * a.ldA + (rin ? r : 0)];
  const double bv0 = (act && rin) ? bv0l : 0.0;
  const double bv1 = (act && rin && a.DP4 > 4) ? bv1l : 0.0;
  double gsum = 0.0;
  // d-split (LayerBwdArgs::d_split): this workgroup's outputs of the d-loop, and the hand-over of the partial tiles
  const int n_split = (int)gridDim.y;
  const int dchunk = (Dout + n_split - 1) / n_split;
  const int d_lo = (int)blockIdx.y * dchunk, d_hi = (d_lo + dchunk < Dout) ? d_lo + dchunk : Dout;
  __shared__ int s_ticket;
  // T: this wave's accumulator tiles, row block of tile q = ib_of(q) (negative: none).  Returns false in the workgroups that are done.
  // hand-over without fences: the partial tiles go out as 8-byte agent-scope (sc1, write-through) stores and come back as sc1 loads —
  // valid across XCDs on their own (MI355X_MICROARCH.md, inter-workgroup visibility: "8-B agent atomics both sides").  The
  // __threadfence() pair of rounds 1-2 wrote back / invalidated a whole L2 per workgroup: +57 us on the 63-row-block first layer
  // of config 2, which kept the split off below Mp = 512.
  auto merge_split = [&](auto& T, auto ib_of) -> bool {
    if (n_split == 1) return true;
    constexpr int NT = sizeof(T) / sizeof(T[0]);
    const int64_t pstride = (int64_t)Mp * 16 + 16;
    double* mine = a.part + ((int64_t)blockIdx.x * n_split + blockIdx.y) * pstride;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int ib = ib_of(q);
      if (ib < 0) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) __hip_atomic_store(&mine[(16 * ib + g + 4 * t) * 16 + c], T[q][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0 && g == 0) __hip_atomic_store(&mine[Mp * 16 + c], gsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores have left before the ticket is drawn
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.part_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != n_split - 1) return false;
    if (tid == 0) __hip_atomic_store(a.part_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every split has arrived: ready for the next launch
    const double* base = a.part + (int64_t)blockIdx.x * n_split * pstride;
#pragma unroll
    for (int q = 0; q < NT; ++q) T[q] = (d4){0, 0, 0, 0};
    gsum = 0.0;
    for (int y = 0; y < n_split; ++y) {                    // fixed order: the sum does not depend on which workgroup arrives last
      const double* p = base + y * pstride;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int ib = ib_of(q);
        if (ib < 0) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) T[q][t] += __hip_atomic_load(&p[(16 * ib + g + 4 * t) * 16 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      gsum += __hip_atomic_load(&p[Mp * 16 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
  };
  BWD_STAMP(1);      // A tile loaded
  if constexpr (CS) {
    // cbar_d = 2 vbar_d c_d staged into LDS ([k][16 rows], B-operand order), abar += q_sqrt_d cbar_d: out block ib sums kb <= ib.
    // Mp <= 256: two staging buffers (the loads of output d+1 fly during the products of d, one barrier per output);
    // Mp >= 512: LDS holds one buffer (two barriers per output, loads still prefetched).
    using OC = OwnCS<MPB, NW>;
    constexpr bool DBUF = (MPB <= 16);
    constexpr int NLT = OC::NLOAD * 64;             // staging threads
    constexpr int NS = Mp * 16 / (NLT * 2);         // staged element PAIRS per staging thread (16-byte loads / LDS stores)
    double* buf1 = DBUF ? smem + L.red : actb;
    const bool stager = tid < NLT;
    const int e0 = 2 * tid;                         // pair i of this thread: elements e0 + i * 2 NLT (+1) of the tile; rows scc, scc + 1
    const int scc = e0 & 15;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 tmp[NS];
    auto stage_load = [&](int d) {
      if (!stager) return;
      const d2 v2 = 2.0 * *reinterpret_cast<const d2*>(a.VB + (int64_t)d * a.ldA + r0 + scc);
      const double* __restrict__ Cd = a.Csave + ((int64_t)blockIdx.x * Dout + d) * (Mp * 16) + e0;
#pragma unroll
      for (int i = 0; i < NS; ++i) tmp[i] = v2 * *reinterpret_cast<const d2*>(Cd + i * 2 * NLT);
    };
    auto stage_store = [&](double* buf) {
      if (!stager) return;
#pragma unroll
      for (int i = 0; i < NS; ++i) *reinterpret_cast<d2*>(buf + e0 + i * 2 * NLT) = tmp[i];
    };
    constexpr bool PREF = (NW < 16);            // NW = 16 (1024 threads, 128-VGPR cap): no prefetch across the products (spills)
    d4 cacc[OC::NQ];
#pragma unroll
    for (int q = 0; q < OC::NQ; ++q) cacc[q] = (d4){0, 0, 0, 0};
    const bool cact = OC::active(wave);
    if (PREF && d_lo < d_hi) stage_load(d_lo);
    if (DBUF && d_lo < d_hi) stage_store(actb);
    for (int d = d_lo; d < d_hi; ++d) {
      gsum += rin ? a.VB[(int64_t)d * a.ldA + r] : 0.0;
      const int dpar = (d - d_lo) & 1;
      double* cur = (DBUF && dpar) ? buf1 : actb;
      if (DBUF) {
        if (d + 1 < d_hi) stage_load(d + 1);
        __syncthreads();
      } else {
        __syncthreads();          // the products of output d-1 are done with the buffer
        if (!PREF) stage_load(d);
        stage_store(actb);
        __syncthreads();
        if (PREF && d + 1 < d_hi) stage_load(d + 1);
      }
      if (cact) {
        const double* __restrict__ Td = a.Tp + (int64_t)d * Mp * Mp;
        const double* __restrict__ TdT = a.TpT + (int64_t)d * Mp * Mp;
#pragma unroll
        for (int q = 0; q < OC::NQ; ++q) {
          const int ib = OC::ib(wave, q);
          if (!OC::CUSTOM && Own<MPB, NW>::skip(ib)) continue;
#pragma unroll 4
          for (int kb = 0; kb <= ib; ++kb) cacc[q] = chain_block<Mp, D4>(Td, TdT, cur, ib, kb, g, c, cacc[q]);
        }
      }
      if (DBUF && d + 1 < d_hi) stage_store(dpar ? actb : buf1);
    }
    if (!merge_split(cacc, [&](int q) { const int ib = OC::ib(wave, q); return (cact && (OC::CUSTOM || !Own<MPB, NW>::skip(ib))) ? ib : -1; })) return;
    // mean part (abar += q_mu mbar) and hand-over of abar through LDS, under the d-loop's ownership
    if (cact) {
      for (int sp = 0; sp < a.DP4 / 4; ++sp) {
        const double bv = sp == 0 ? bv0 : (sp == 1 ? bv1 : (rin ? a.MB[(int64_t)(4 * sp + g) * a.ldA + r] : 0.0));
#pragma unroll
        for (int q = 0; q < OC::NQ; ++q) {
          const int ib = OC::ib(wave, q);
          if (!OC::CUSTOM && Own<MPB, NW>::skip(ib)) continue;
          cacc[q] = mfma_f64(a.qmu4[(int64_t)(16 * ib + c) * a.DP4 + 4 * sp + g], bv, cacc[q]);
        }
      }
    }
    __syncthreads();   // the last staged operand is consumed -> abar goes into the first buffer
    if (cact) {
#pragma unroll
      for (int q = 0; q < OC::NQ; ++q) {
        const int ib = OC::ib(wave, q);
        if (!OC::CUSTOM && Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) actb[out_slot<D4>(ib, g, t) * 16 + c] = cacc[q][t];
      }
    }
    __syncthreads();
    if (WHITE) {       // a1bar = abar - 2 gsum a1, applied by the waves that hold those rows of a1 (standard ownership)
      if (act) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
          for (int t = 0; t < 4; ++t) actb[out_slot<D4>(ib, g, t) * 16 + c] -= 2.0 * gsum * av[q][t];
        }
      }
      __syncthreads();
    }
  } else {
  __syncthreads();
  // NQ = 2 without the interleaved form (Mp = 128 on four waves): the two row-block chains of the wave run side by side on one
  // activation fragment per k-step, ONE weight register set refilled in place right behind the MFMAs that consumed it, the refill
  // crossing from the last k-block of output d into the first of d + 1 (the S_d are contiguous: k-block t = d MPB + kb of one tall
  // matrix).  The interleave is pinned (chain_range2 has the story); 96 VGPRs = five workgroups per CU.
  constexpr bool PAIRD = (NQ == 2) && !D4 && (MPB % NW == 0);
  if constexpr (PAIRD) {
    const int ib0 = Own<MPB, NW>::ib(wave, 0), ib1 = Own<MPB, NW>::ib(wave, 1);
    const unsigned off0 = (unsigned)(g * Mp + 16 * ib0 + c), off1 = (unsigned)(g * Mp + 16 * ib1 + c);
    const int t_last = d_hi * MPB - 1;
    auto wld = [&](int t, int s, unsigned off) { return (a.Sd + (size_t)(16 * t + 4 * s) * Mp)[off]; };
    double w0[4], w1[4];
    if (d_lo < d_hi) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        w0[s] = wld(d_lo * MPB, s, off0);
        w1[s] = wld(d_lo * MPB, s, off1);
      }
    }
    // vbar_d of the NEXT output is requested inside the k loop of this one, unconditionally (clamped): at the top of the loop it sat
    // behind an exec branch and its first use drained the whole load queue (s_waitcnt vmcnt(0)) once per output
    const int64_t rc = rin ? r : 0;
    double vd_next = d_lo < d_hi ? a.VB[(int64_t)d_lo * a.ldA + rc] : 0.0;
    for (int d = d_lo; d < d_hi; ++d) {
      const double vd = rin ? vd_next : 0.0;      // (select, not a product: padded rows get an exact 0 whatever the clamped load returned)
      gsum += vd;
      const double vd2 = 2.0 * vd;
      d4 y0 = (d4){0, 0, 0, 0}, y1 = (d4){0, 0, 0, 0};
      double bv = actb[g * 16 + c];
#pragma unroll
      for (int kb = 0; kb < MPB; ++kb) {
        int tn = d * MPB + kb + 1;              // (the last refill of the launch re-reads its own block)
        if (kb == MPB - 1 && tn > t_last) tn = t_last;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // one step = the activation fragment of the NEXT step requested, two MFMAs, their two weight registers refilled; a full
          // scheduling barrier behind every step: with group barriers alone the scheduler still chose WHICH loads go into a group,
          // and on the 8-wave M = 256 instance it picked the ones needed next (queue depth 1-2 instead of 8)
          const int kn = (s == 3) ? (kb + 1) % MPB : kb, sn = (s + 1) & 3;
          const double bvn = actb[(16 * kn + 4 * sn + g) * 16 + c];
          y0 = mfma_f64(w0[s], bv, y0);
          y1 = mfma_f64(w1[s], bv, y1);
          w0[s] = wld(tn, s, off0);
          w1[s] = wld(tn, s, off1);
          bv = bvn;
          if (kb == 1 && s == 0) vd_next = a.VB[(int64_t)(d + 1 < d_hi ? d + 1 : d) * a.ldA + rc];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][t] = fma(vd2, y0[t], acc[0][t]);
        acc[1][t] = fma(vd2, y1[t], acc[1][t]);
      }
    }
  } else
  for (int d = d_lo; d < d_hi; ++d) {
    const double vd = rin ? a.VB[(int64_t)d * a.ldA + r] : 0.0;
    gsum += vd;
    const double vd2 = 2.0 * vd;
    if (act) {
      const double* __restrict__ Sd = a.Sd + (int64_t)d * Mp * Mp;
      if constexpr (ILV) {
        // y_d = S_d a for this wave's row blocks (the NQ chains interleaved), then abar += 2 vbar_d(column) * y_d
        d4 y[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] = (d4){0, 0, 0, 0};
#pragma unroll 4
        for (int kb = 0; kb < MPB; ++kb) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int ibq = Own<MPB, NW>::ib(wave, q);
            if (!Own<MPB, NW>::skip(ibq)) y[q] = chain_block<Mp, D4>(Sd, Sd, actb, ibq, kb, g, c, y[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[q][t] = fma(vd2, y[q][t], acc[q][t]);
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
          // y_d = S_d a, then abar += 2 vbar_d y_d (four FMAs per output).  The earlier form scaled the B operand instead
          // (one v_mul_f64 per MFMA): fp64 VALU work competes with the fp64 MFMAs — that multiply alone cost 16 % of the
          // loop (tools/chain_loop_bench.hip: 113 -> 95 us at the D_out = 8 launch of config 2).
          const double* __restrict__ W = Sd + 16 * ib + c + (int64_t)g * Mp;
          d4 y = (d4){0, 0, 0, 0};
#pragma unroll 4
          for (int kb = 0; kb < MPB; ++kb) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              y = mfma_f64(W[(int64_t)(16 * kb + 4 * s) * Mp], actb[(16 * kb + 4 * s + g) * 16 + c], y);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[q][t] = fma(vd2, y[t], acc[q][t]);
        }
      }
    }
  }
  if (!merge_split(acc, [&](int q) { const int ib = Own<MPB, NW>::ib(wave, q); return (act && !Own<MPB, NW>::skip(ib)) ? ib : -1; })) return;
  }
  BWD_STAMP(2);      // d-loop done (and, with a d-split, the partial tiles merged)
  if constexpr (!CS) {
  if (act) {
    for (int sp = 0; sp < a.DP4 / 4; ++sp) {
      const double bv = sp == 0 ? bv0 : (sp == 1 ? bv1 : (rin ? a.MB[(int64_t)(4 * sp + g) * a.ldA + r] : 0.0));
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
        acc[q] = mfma_f64(a.qmu4[(int64_t)(16 * ib + c) * a.DP4 + 4 * sp + g], bv, acc[q]);
      }
    }
  }
  __syncthreads();   // "a" fully consumed from LDS -> overwrite with abar in place
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB, NW>::ib(wave, q);
      if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (WHITE) acc[q][t] -= 2.0 * gsum * av[q][t];
        actb[out_slot<D4>(ib, g, t) * 16 + c] = acc[q][t];
      }
    }
  }
  __syncthreads();
  }
  BWD_STAMP(3);      // mean part, abar in LDS
  // b = Ku^{-1} abar (dense)   |   white: kbar = Lu^{-T} a1bar (k-blocks >= own block)
  d4 bb[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) bb[q] = (d4){0, 0, 0, 0};
  constexpr bool PAIRK = (NQ == 2) && !D4 && !CS && (MPB % NW == 0);     // (the d-loop's paired form, see there)
  if constexpr (PAIRK) {
    if constexpr (WHITE) {
      const int wu = DS_WAVE_ID(tid);                                       // run-time k ranges: wave-uniform bounds
      chain_range2<Mp, true>(a.Linv, actb, Own<MPB, NW>::ib(wu, 0), Own<MPB, NW>::ib(wu, 1), MPB, g, c, bb[0], bb[1]);
    } else {
      chain_dense2<Mp, MPB>(a.Kinv, actb, Own<MPB, NW>::ib(wave, 0), Own<MPB, NW>::ib(wave, 1), g, c, bb[0], bb[1]);
    }
  } else
  if (act) {
    if (WHITE || !ILV) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ib = Own<MPB, NW>::ib(wave, q);
        if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll 4
        for (int kb = (WHITE ? ib : 0); kb < MPB; ++kb)
          bb[q] = WHITE ? chain_block<Mp, D4>(a.LinvT, a.Linv, actb, ib, kb, g, c, bb[q])
                        : chain_block<Mp, D4>(a.Kinv, a.Kinv, actb, ib, kb, g, c, bb[q]);
      }
    } else {
#pragma unroll 4
      for (int kb = 0; kb < MPB; ++kb) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ibq = Own<MPB, NW>::ib(wave, q);
          if (!Own<MPB, NW>::skip(ibq)) bb[q] = chain_block<Mp, D4>(a.Kinv, a.Kinv, actb, ibq, kb, g, c, bb[q]);
        }
      }
    }
  }
  BWD_STAMP(4);      // Ku^-1 product
  // E, kbar; recompute the Kuf tile for GW = kbar * dk/dr2
  d4 r2[NQ];
  if constexpr (WIDE) sm_sqdist<NQ, MPB, NW>(zs, a.X, ils, xs, Din, r0, a.Rin, tid, wave, g, c, act, redx, r2);
  else sqdist_narrow<NQ, MPB, NW>(zs, xs, Din, wave, g, c, act, redx, r2);      // x / l staged at kernel start
  // the wave-private scratch of the distance code lives where the dX partials go: a wave that runs ahead into the reductions below
  // must not overwrite the scratch of one that is still here (seen as a non-deterministic gradient with the Csave chain)
  __syncthreads();
  BWD_STAMP(5);      // (wide inputs: distances chunked through LDS)
  double svar = 0.0;
  if (act) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB, NW>::ib(wave, q);
      if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = 16 * ib + g + 4 * t;
        const double e = WHITE ? bb[q][t] : bb[q][t] - gsum * av[q][t];
        const double kbar = WHITE ? e : e - gsum * av[q][t];
        const double r2v = r2[q][t];
        double k, dk;
        kern_val_grad<KIND>(r2v, s2, k, dk);
        const bool ok = rvalid && (m < a.M);
        svar += ok ? kbar * k : 0.0;
        const double w = ok ? kbar * dk : 0.0;
        bb[q][t] = w;
        // NULL: d loss / d Ku is assembled from the P_d (model.hip, alg_g).  GW is stored at the end of the kernel: a store issued here
        // makes the Z loads of the reductions below wait for its write acknowledgement (vmcnt retires in order).
        if (rin && a.E) a.E[(int64_t)m * a.ldA + r] = e;
      }
    }
  }
  BWD_STAMP(6);      // kernel recompute, E / GW stores issued
  svar = sum_wave(svar);
  const double gk = sum_wave((rvalid && g == 0 && wave == 0) ? gsum : 0.0);
  double* hp = a.hyp_part + ((int64_t)blockIdx.x * NW + wave) * (Din + 2);
  if (lane == 0) {
    hp[0] = svar / s2;
    hp[1] = gk;
  }
  // lengthscale partials and d loss / d X, chunked over D_in like the distance computation
  for (int j0 = 0; j0 < Din; j0 += XCH) {
    const int jn = (Din - j0 < XCH) ? Din - j0 : XCH;
    if (WIDE) {   // single-chunk kernels: xs still holds x/l staged above
      __syncthreads();
      for (int idx = tid; idx < 16 * jn; idx += NW * 64) {
        const int rr = idx / jn, j = idx % jn;
        int64_t row = r0 + rr;
        if (row > a.Rin - 1) row = a.Rin - 1;
        xs[rr * (jn + 1) + j] = a.X[row * Din + j0 + j] * ils[j0 + j];
      }
      __syncthreads();
    }
    // epilogue operands of this thread's item (one item per thread when 16 jn <= threads): requested now, they arrive during the
    // reductions below instead of after the barrier
    const bool single = 16 * jn <= NW * 64;
    double pre_mb = 0.0, pre_z = 0.0, pre_var = 1.0;
    if ((a.dX || a.MBp) && single && tid < 16 * jn) {
      const int j = a.MBp ? tid / 16 : tid % jn, cc = a.MBp ? tid % 16 : tid / jn;
      const int64_t row = r0 + cc;
      if (row < a.Rin) {
        if (a.mean_kind == DSDGP_MEAN_IDENTITY) pre_mb = a.MB[(int64_t)(j0 + j) * a.ldA + row];
        const int d = j0 + j - a.prop;
        if (a.MBp && d >= 0) {
          pre_z = a.zp[(row / a.n_inner) * a.zp_s + (row % a.n_inner) * a.zp_n + d * a.zp_d];
          pre_var = a.varp[row * a.Dp + d];
        }
      }
    }
    // The sums over this wave's inducing rows run on the MFMA pipe, 16 input dimensions at a time.
    //   WZ[j][c] = sum_m z[m][j] w[m][c],  Z2[j][c] = sum_m z[m][j]^2 w[m][c]   (A = Zs^T tile: lane (g, c) loads row 16 ib + g + 4 t,
    //   dimension c of the group — 128-byte runs; B = w, which sits in the accumulator layout = the B layout of k-steps t)
    //   d X partial: sum_m w (x - z) = x W1 - WZ          lengthscale partial: sum_m,c w (x - z)^2 = sum_c (x^2 W1 - 2 x WZ + Z2)
    // with W1[c] = sum_m w[m][c].  One element-by-element pass per dimension (16 L2 loads, two cross-lane reductions) took
    // 1.93 M of the 5.2 M clocks of the 784-dimensional first layer of config 4 and 22 K of the 59 K clocks of a D_out = 1 workgroup
    // of config 2 (profiles/r02_chain_phases.txt); this form 0.41 M and 12 K.
    double w1 = 0.0;
    if (act) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) w1 += bb[q][t];           // rows beyond M and skipped blocks hold 0
    }
    w1 = sum_groups(w1);
    for (int kk = 0; kk < jn; kk += 16) {
      d4 wz = (d4){0, 0, 0, 0}, z2 = (d4){0, 0, 0, 0};
      if (act) {
        const int jc = (kk + c < jn) ? kk + c : jn - 1;       // clamped (unconditional) loads; dimensions past jn are dropped below
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int ib = Own<MPB, NW>::ib(wave, q);
          if (Own<MPB, NW>::skip(ib)) continue;
          double zv[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) zv[t] = zs[(int64_t)(16 * ib + g + 4 * t) * Din + j0 + jc];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            wz = mfma_f64(zv[t], bb[q][t], wz);
            z2 = mfma_f64(zv[t] * zv[t], bb[q][t], z2);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = kk + g + 4 * t;                           // accumulator row = dimension kk + g + 4 t, column c
        const bool in = j < jn;
        const double xv = xs[c * (jn + 1) + (in ? j : jn - 1)];
        if ((a.dX || a.MBp) && in) redx[(wave * jn + j) * 16 + c] = act ? fma(xv, w1, -wz[t]) : 0.0;
        double sl = act ? fma(xv * xv, w1, fma(-2.0 * xv, wz[t], z2[t])) : 0.0;
        sl += dpp_or_zero<0x111, 0xf>(sl);                      // sum over the 16 columns of this row of lanes (-> lane c = 15)
        sl += dpp_or_zero<0x112, 0xf>(sl);
        sl += dpp_or_zero<0x114, 0xf>(sl);
        sl += dpp_or_zero<0x118, 0xf>(sl);
        if (c == 15 && in) hp[2 + j0 + j] = -2.0 * ils[j0 + j] * sl;
      }
    }
    if (a.dX || a.MBp) {
      __syncthreads();
      for (int idx = tid; idx < 16 * jn; idx += NW * 64) {
        // dX is row-major (j fastest); the fused transposed adjoints are M-major (row fastest): keep the stores coalesced
        const int j = a.MBp ? idx / 16 : idx % jn, cc = a.MBp ? idx % 16 : idx / jn;
        const int64_t row = r0 + cc;
        if (row < a.Rin) {
          double sx = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) sx += redx[(w * jn + j) * 16 + cc];
          double dx = 2.0 * ils[j0 + j] * sx;
          if (a.mean_kind == DSDGP_MEAN_IDENTITY) {
            dx += single ? pre_mb : a.MB[(int64_t)(j0 + j) * a.ldA + row];
          } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
            for (int d = 0; d < Dout; ++d) dx = fma(a.mean_A[(int64_t)(j0 + j) * Dout + d], a.MB[(int64_t)d * a.ldA + row], dx);
          }
          if (a.MBp) {
            const int d = j0 + j - a.prop;
            if (d >= 0) {
              const double zv = single ? pre_z : a.zp[(row / a.n_inner) * a.zp_s + (row % a.n_inner) * a.zp_n + d * a.zp_d];
              const double vv = single ? pre_var : a.varp[row * a.Dp + d];
              a.MBp[(int64_t)d * a.ldA + row] = dx;
              a.VBp[(int64_t)d * a.ldA + row] = dx * zv * 0.5 * rsqrt(vv + a.jitter);
            }
          } else {
            a.dX[row * Din + j0 + j] = dx;
          }
        } else if (a.MBp && row < a.ldA && j0 + j >= a.prop) {
          a.MBp[(int64_t)(j0 + j - a.prop) * a.ldA + row] = 0.0;
          a.VBp[(int64_t)(j0 + j - a.prop) * a.ldA + row] = 0.0;
        }
      }
    }
  }
  if (act && rin) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ib = Own<MPB, NW>::ib(wave, q);
      if (Own<MPB, NW>::skip(ib)) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) a.GW[(int64_t)(16 * ib + g + 4 * t) * a.ldA + r] = bb[q][t];
    }
  }
  BWD_STAMP(7);      // hyper-parameter partials, dX / transposed adjoints, GW
}


// ---- launch helpers (one instance per template combination) and dispatch macros shared by the layer_sm_*.hip parts
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE, bool LIK>
static int fwd_sm_go2(dsdgp_ctx* ctx, const LayerFwdArgs& a) {
  const SmLds L = sm_lds(MPB * 16, a.D_in, a.D_out, NW, WIDE);
  const size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_fwd(sm): needs %zu B LDS (> 160 KiB): D_out too large for this M", lds);
    return DSDGP_ERR_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    {
      static int lds_set = 0;   // the attribute is sticky: one driver call per instance and size
      if ((int)lds > lds_set) {
        DS_HIP(hipFuncSetAttribute((const void*)k_layer_fwd_sm<MPB, NW, KIND, WHITE, WIDE, LIK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = (int)lds;
      }
    }
  ProfScope ps(ctx, "layer_fwd");
  const int nrow = ceil_div(a.Rin, 16);
  int ds = a.d_split > 0 ? a.d_split : 1;
  if (ds > a.D_out) ds = a.D_out;
  static const bool timing = getenv("DSDGP_FWD_TIMING") != nullptr;
  if (timing) {      // debug aid, synchronous: per-phase times (s_memrealtime, 100 MHz) averaged over the workgroups of this launch
    unsigned long long* clk = nullptr;
    DS_HIP(hipMalloc(&clk, (size_t)nrow * 8 * sizeof(unsigned long long)));
    LayerFwdArgs b = a;
    b.phase_clk = clk;
    DS_LAUNCH((k_layer_fwd_sm<MPB, NW, KIND, WHITE, WIDE, LIK>), dim3(nrow, ds), dim3(NW * 64), lds, ctx->stream, b, L);
    DS_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h((size_t)nrow * 8);
    DS_HIP(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    hipFree(clk);
    double ph[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < nrow; ++w)
      for (int i = 0; i < 6; ++i) ph[i] += (double)(h[(size_t)w * 8 + i + 1] - h[(size_t)w * 8 + i]);
    phase_span_report(h, nrow, 6);
    const double u = 0.01 / nrow;      // 100 MHz ticks -> us per workgroup
    fprintf(stderr, "[fwd phases] Mp=%d NW=%d D_out=%d wgs=%dx%d  us/wg: Kuf tile %.2f | a1 %.2f | a %.2f | mean partials %.2f | per-output + epilogue %.2f | "
            "Asave %.2f | sum %.2f\n", MPB * 16, NW, a.D_out, nrow, ds, ph[0] * u, ph[1] * u, ph[2] * u, ph[3] * u, ph[4] * u,
            ph[5] * u, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]) * u);
    return DSDGP_OK;
  }
  DS_LAUNCH((k_layer_fwd_sm<MPB, NW, KIND, WHITE, WIDE, LIK>), dim3(nrow, ds), dim3(NW * 64), lds, ctx->stream, a, L);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE>
static int fwd_sm_go(dsdgp_ctx* ctx, const LayerFwdArgs& a) {
  if (a.lik_Y) return fwd_sm_go2<MPB, NW, KIND, WHITE, WIDE, true>(ctx, a);
  return fwd_sm_go2<MPB, NW, KIND, WHITE, WIDE, false>(ctx, a);
}
// DSDGP_BWD_TIMING=1 (debug aid, synchronous): per-phase times (s_memrealtime, 100 MHz) of every backward-chain launch, averaged over its workgroups
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE, bool CS>
static int bwd_phase_timing(dsdgp_ctx* ctx, const LayerBwdArgs& a0, const SmLds& L, size_t lds) {
  const int nwg = (int)ceil_div(a0.ldA, 16);
  unsigned long long* clk = nullptr;
  DS_HIP(hipMalloc(&clk, (size_t)nwg * 8 * sizeof(unsigned long long)));
  LayerBwdArgs a = a0;
  a.phase_clk = clk;
  DS_LAUNCH((k_layer_bwd_sm<MPB, NW, KIND, WHITE, WIDE, CS>), dim3(nwg, a.d_split > 1 ? a.d_split : 1), dim3(NW * 64), lds, ctx->stream, a, L);
  DS_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<unsigned long long> h((size_t)nwg * 8);
  DS_HIP(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  hipFree(clk);
  double ph[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < nwg; ++w) {
    for (int i = 0; i < 7; ++i) ph[i] += (double)(h[(size_t)w * 8 + i + 1] - h[(size_t)w * 8 + i]);
  }
  phase_span_report(h, nwg, 7);
  const double u = 0.01 / nwg;      // 100 MHz ticks -> us per workgroup
  fprintf(stderr, "[bwd phases] Mp=%d NW=%d D_out=%d wgs=%d  us/wg: A-tile %.2f | d-loop %.2f | mean+abar %.2f | Kinv %.2f | x-stage %.2f | "
          "kernel+E/GW %.2f | hyp+dX %.2f | sum %.2f\n", MPB * 16, NW, a.D_out, nwg, ph[0] * u, ph[1] * u, ph[2] * u, ph[3] * u, ph[4] * u, ph[5] * u,
          ph[6] * u, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6]) * u);
  return DSDGP_OK;
}
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE, bool CS>
static int bwd_sm_go2(dsdgp_ctx* ctx, const LayerBwdArgs& a) {
  const SmLds L = sm_lds(MPB * 16, a.D_in, a.D_out, NW, WIDE, (CS && MPB <= 16) ? 2 : 1);
  const size_t lds = (size_t)L.total * sizeof(double);
  if (lds > 160 * 1024) {
    dsdgp_set_error("layer_bwd(sm): needs %zu B LDS (> 160 KiB)", lds);
    return DSDGP_ERR_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    {
      static int lds_set = 0;   // the attribute is sticky: one driver call per instance and size
      if ((int)lds > lds_set) {
        DS_HIP(hipFuncSetAttribute((const void*)k_layer_bwd_sm<MPB, NW, KIND, WHITE, WIDE, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = (int)lds;
      }
    }
  ProfScope ps(ctx, "layer_bwd");
  static const bool timing = getenv("DSDGP_BWD_TIMING") != nullptr;
  if (timing) return bwd_phase_timing<MPB, NW, KIND, WHITE, WIDE, CS>(ctx, a, L, lds);
  DS_LAUNCH((k_layer_bwd_sm<MPB, NW, KIND, WHITE, WIDE, CS>), dim3(ceil_div(a.ldA, 16), a.d_split > 1 ? a.d_split : 1), dim3(NW * 64), lds,
                     ctx->stream, a, L);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
// Csave instances are built for the sizes that use them by default (Mp >= 320) and for the parity-tested small ones
template <int MPB, int NW>
constexpr bool sm_cs_inst() { return MPB > 16 || MPB == 2 || MPB == 4 || MPB == 8 || MPB == 16; }
template <int MPB, int NW, int KIND, bool WHITE, bool WIDE>
static int bwd_sm_go(dsdgp_ctx* ctx, const LayerBwdArgs& a) {
  if constexpr (sm_cs_inst<MPB, NW>()) {
    if (a.Csave) return bwd_sm_go2<MPB, NW, KIND, WHITE, WIDE, true>(ctx, a);
  }
  return bwd_sm_go2<MPB, NW, KIND, WHITE, WIDE, false>(ctx, a);
}



#define SM_CASE2(FN, MPB, NW, KIND, ARGS)                                                              \
  if (wide) return white ? FN<MPB, NW, KIND, true, true> ARGS : FN<MPB, NW, KIND, false, true> ARGS;      \
  return white ? FN<MPB, NW, KIND, true, false> ARGS : FN<MPB, NW, KIND, false, false> ARGS;
#define SM_CASE(FN, MPB, NW, ARGS)                                           \
  case MPB * 16:                                                             \
    if (kern_kind == DSDGP_KERN_RBF) { SM_CASE2(FN, MPB, NW, DSDGP_KERN_RBF, ARGS) }   \
    SM_CASE2(FN, MPB, NW, DSDGP_KERN_MATERN52, ARGS)
#define SM_SMALL_CASE(FN, MPB, ARGS)                                                                       \
  if (Mp == MPB * 16) {                                                                                    \
    if (kern_kind == DSDGP_KERN_RBF)                                                                       \
      return white ? FN<MPB, 8, DSDGP_KERN_RBF, true, false> ARGS : FN<MPB, 8, DSDGP_KERN_RBF, false, false> ARGS; \
    return white ? FN<MPB, 8, DSDGP_KERN_MATERN52, true, false> ARGS : FN<MPB, 8, DSDGP_KERN_MATERN52, false, false> ARGS; \
  }
#define SM_NOT_BUILT                                                                     \
  dsdgp_set_error("layer chain: padded inducing count %d not built (32..1024)", Mp);     \
  return DSDGP_ERR_UNSUPPORTED;
