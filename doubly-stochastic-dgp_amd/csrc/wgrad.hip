// Split-K "weight gradient" products over the data rows of the DS-DGP reverse pass (reverse mode of layers.py:184-217 w.r.t. Ku,
// q_sqrt, q_mu, Z: sums over the rows of the minibatch x samples) on gfx950 fp64 MFMA.
#include "layer.hpp"
#include <stdlib.h>

// out = P diag(scale) Q^T with P (rowsP x R), Q (rowsQ x R) both M-major.  Operands go straight from L2 / Infinity Cache to MFMA
// registers: each lane loads 4 consecutive r (32 B) of one row, and the t-th of them is the k-operand of the t-th MFMA, so a
// 16-row x 16-r fragment costs two 16-B loads per lane and no LDS.
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


// The same K loop with the operands refilled IN PLACE, half a chunk at a time (round 5).  A lane's 32-byte fragment is two 16-byte
// halves: k-steps t = 0, 1 read the first, t = 2, 3 the second.  The loop above loads a whole chunk, waits, and runs its 64 MFMAs — with
// two waves per SIMD (236 VGPRs) nothing but the other wave covers the round trip, and a second register set for prefetching does not
// fit.  Here the first halves of chunk ch + 1 are requested right behind the MFMAs of t = 0, 1 of chunk ch (into the registers those
// just read) and are due 32 MFMAs later, the second halves likewise: half a chunk of loads is always in flight under half a chunk of
// MFMAs, with the register count of the plain loop.  Every phase is fenced with a full scheduling barrier (left to the scheduler the
// refill form came out at 360 registers).  DIAG: no Q loads — the Q registers hold the scaled copies of P.  Non-GUARD tiles only.
typedef double wd2 __attribute__((ext_vector_type(2)));
template <int NI, int NJ, bool DIAG>
__device__ __forceinline__ void wgrad_loop_rolling(gcptr Pp, gcptr Qp, gcptr scale, int64_t ld, int64_t c_lo, int64_t c_hi, int g,
                                                   d4 (&acc)[NI][NJ]) {
  if (c_lo >= c_hi) return;
  typedef const wd2 __attribute__((address_space(1)))* h2ptr;
  wd2 p[2][NI], q[2][NJ], sc[2];
  auto request = [&](int64_t ch, int h) {
    const int64_t rb = ch * 16 + 2 * h;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) p[h][ii] = *reinterpret_cast<h2ptr>(Pp + (int64_t)16 * ii * ld + rb);
    if constexpr (!DIAG) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) q[h][jj] = *reinterpret_cast<h2ptr>(Qp + (int64_t)16 * jj * ld + rb);
    }
    if (scale) sc[h] = *reinterpret_cast<h2ptr>(scale + rb + 4 * g);
  };
  request(c_lo, 0);
  request(c_lo, 1);
  for (int64_t ch = c_lo; ch < c_hi; ++ch) {
    const int64_t nx = ch + 1 < c_hi ? ch + 1 : ch;        // (the last chunk re-requests itself: unused)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (DIAG) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) q[h][jj] = scale ? p[h][jj] * sc[h] : p[h][jj];
      } else if (scale) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) q[h][jj] *= sc[h];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
            if (!DIAG || jj <= ii) acc[ii][jj] = mfma_f64(p[h][ii][t], q[h][jj][t], acc[ii][jj]);
      __builtin_amdgcn_sched_barrier(0);
      request(nx, h);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ONE WORKGROUP per (job, split, tile) task; its four waves take a quarter of the split's row range each and reduce their
// accumulators through LDS in a fixed order ((w0 + w2) + (w1 + w3)) before wave 0 stores the partial: a quarter of the split-K
// partials of a one-wave-per-task form at the same wave-level parallelism (fp64 MFMA needs >= 2 waves per SIMD for its pipe rate).
// Forms measured slower and removed in round 3: one wave per task, 32 x 32 tiles, 128 x 128 tiles staged through LDS.
template <int NI, int NJ>
__global__ __launch_bounds__(256, 2) void k_wgrad_coop(const WgradJob* __restrict__ jobs, int njobs, int nsplit, int64_t ld,
                                                    int64_t Rp, int total_tasks) {
  // [2][NS][64] accumulator hand-over of the four waves; the in-launch reduction parks the finished 64 x 65 tile in the second half
  __shared__ double red[2 * NI * NJ * 4 * 64 + 128];
  __shared__ int s_ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x;
  int jb = 0;
  DS_FIND_SEGMENT(jb, jobs, njobs, task_start, w);
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
    wgrad_loop_rolling<NI, NJ, true>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, acc);
  else if (njv == NJ)
    wgrad_loop_rolling<NI, NJ, false>(Pp, Qp, (gcptr)J.scale, ld, c_lo, c_hi, g, acc);
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
  const double* r = red + lane;
  const int rowsP = 16 * NI * J.ti;
  if (!J.fin) {
    if (wave != 0) return;
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
    return;
  }
  // ---- in-launch split-K reduction (WgradJob::fin).  The workgroup's tile goes to LDS first (T, 64 x 65: the second half of `red`, free since
  // wave 1 took wave 3's accumulators) — the accumulators are dead from here on — and everything below moves whole 512-byte rows of it:
  // thread (wave, lane) owns column `lane` of the rows wave + 4 k.
  static_assert(NI == 4 && NJ == 4, "tile staging assumes 64 x 64 tiles");
  double* T = red + NS * 64;
  if (wave == 0) {
#pragma unroll
    for (int ii = 0; ii < NI; ++ii)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (jj < njv && !(diag && jj > ii)) {
#pragma unroll
          for (int t = 0; t < 4; ++t) T[(16 * ii + g + 4 * t) * 65 + 16 * jj + c] = acc[ii][jj][t] + r[((ii * NJ + jj) * 4 + t) * 64];
        }
  }
  __syncthreads();
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  const int cc = lane;
  // blocks of the tile that exist: the first njv column blocks; on or below the block diagonal for the diagonal tile of a symmetric result
  const bool col_ok = cc < 16 * njv;
  if (ns_eff > 1) {
    // hand-over without fences (the d-split of the backward chain, layer_sm_impl.hpp: merge_split): partials out as 8-byte agent-scope
    // stores, one s_waitcnt before the ticket, agent-scope loads by the last arrival — valid across XCDs on their own
    double* ob = J.out + (int64_t)split * rowsP * J.ldo + (int64_t)(64 * tile_i) * J.ldo + 64 * tile_j;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int rr = wu + 4 * k;
      if (col_ok && !(diag && (cc >> 4) > (rr >> 4)))
        __hip_atomic_store(ob + (int64_t)rr * J.ldo + cc, T[rr * 65 + cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tix = J.sym ? (tile_i == tile_j ? J.ti * (J.ti - 1) / 2 + tile_i : tile_i * (tile_i - 1) / 2 + tile_j) : tile_i * J.tj + tile_j;
      const int tk = __hip_atomic_fetch_add(J.tick + tix, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == ns_eff - 1) __hip_atomic_store(J.tick + tix, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every split has arrived
      s_ticket = tk;
    }
    __syncthreads();
    if (s_ticket != ns_eff - 1) return;
    // last arrival: the tile's partials added in split order, RB splits of loads in flight (uniform base + lane offset: a 64-bit
    // pointer per load would take the batch's registers twice over)
    constexpr int RB = 2;      // (more in flight would spill: this path is the non-default one, see ensure_plan)
    const double* ub = J.out + (int64_t)(64 * tile_i) * J.ldo + 64 * tile_j;
    const int64_t ps = (int64_t)rowsP * J.ldo;
    double sum[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sum[k] = 0.0;
    for (int s0 = 0; s0 < ns_eff; s0 += RB) {
      double x[RB][16];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int sp = s0 + u < ns_eff ? s0 + u : ns_eff - 1;
        const double* us = ub + sp * ps;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int rr = wu + 4 * k;
          x[u][k] = (col_ok && !(diag && (cc >> 4) > (rr >> 4))) ? __hip_atomic_load(us + (int64_t)rr * J.ldo + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < RB; ++u)
        if (s0 + u < ns_eff) {
#pragma unroll
          for (int k = 0; k < 16; ++k) sum[k] += x[u][k];
        }
    }
    // (every wave is past the barrier behind its reads of T)
#pragma unroll
    for (int k = 0; k < 16; ++k) T[(wu + 4 * k) * 65 + cc] = sum[k];
    __syncthreads();
  }
  {
    // a wave instruction stores one 512-byte row of the tile; the mirror tile reads T transposed (odd stride: conflict-free)
    const int gc = 64 * tile_j + cc;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int rr = wu + 4 * k;
      const int gr = 64 * tile_i + rr;
      if (gr < J.fin_rows && gc < J.fin_cols) {
        // diagonal tile of a symmetric result: the blocks above the block diagonal were not formed — they mirror the ones below
        const double v = (diag && (cc >> 4) > (rr >> 4)) ? T[cc * 65 + rr] : T[rr * 65 + cc];
        J.fin[(int64_t)gr * J.fin_ld + gc] = v;
      }
    }
    if (J.sym && !diag) {
      const int mc = 64 * tile_i + cc;      // mirror tile (tile_j, tile_i): element (rr, cc) = T[cc][rr]
#pragma unroll 4
      for (int k = 0; k < 16; ++k) {
        const int rr = wu + 4 * k;
        const int mr = 64 * tile_j + rr;
        if (mr < J.fin_rows && mc < J.fin_cols) J.fin[(int64_t)mr * J.fin_ld + mc] = T[cc * 65 + rr];
      }
    }
  }
}

int wgrad_launch(dsdgp_ctx* ctx, const WgradJob* jobs_dev, int njobs, int total_tasks, int nsplit, int64_t ld, int64_t Rp,
                 hipStream_t stream) {
  if (total_tasks <= 0) return DSDGP_OK;
  hipStream_t st = stream ? stream : ctx->stream;
  ProfScope ps(ctx, "wgrad", st);
  DS_LAUNCH((k_wgrad_coop<4, 4>), dim3(total_tasks), dim3(256), 0, st, jobs_dev, njobs, nsplit, ld, Rp, total_tasks);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
