// Materialised Gram assembly: feature.Kuu / feature.Kuf (layers.py:171,184 -> [UPSTREAM] kern.K).  D <= 32: distances through the MFMA
// pipe (k_gram_mfma2 / k_gram_mfma3 below, two launch forms of one tile arithmetic); any D: k_gram (direct differences):
// HBM-bound: each workgroup builds an 8 x 512 output tile (pairwise squared distances by direct differences, so r2 >= 0 and
// K(X,X) is exactly symmetric) and writes it as 4 KB row segments of 16-byte non-temporal stores.  Algorithmic bytes = 8 * n * n2 (output) + 8 * D * (n + n2) (inputs).
#include <algorithm>

#include "common.hpp"

#define GR_TI 8      // rows (X / inducing side) per workgroup
#define GR_TJ 512    // columns (X2 / data side) per workgroup: two per thread -> 16-byte stores, 4 KB runs per output row
#define GR_DC 8      // input dimensions per pass

#define GR_LD (GR_TJ + 8)   // LDS row stride of the transposed X2 tile (doubles)

// Each thread owns two adjacent columns and all GR_TI rows of the tile.  Per pass of GR_DC input dimensions the workgroup's
// 512 X2 rows are read COALESCED (8 threads = one 64-byte row chunk: 4-8 cache lines per wave-instruction; one thread reading
// its own rows with 8-byte loads touches 64 lines per instruction and the kernel turns texture-address-bound, measured 3x
// slower), scaled by 1/lengthscale and parked TRANSPOSED in LDS ([dim][row]: a thread's two columns are one 16-byte read);
// the GR_TI scaled X rows are LDS broadcasts; distances accumulate in registers and the 16 kernel values leave as 8 double2
// non-temporal stores (1 KB per wave-instruction, 4 KB runs per output row).
template <int KIND>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ X, int64_t n, const double* __restrict__ X2,
                                              int64_t n2, int D, const double* __restrict__ hyp, double diag_add,
                                              int symmetric, double* __restrict__ out, int64_t ld) {
  __shared__ __attribute__((aligned(16))) double xi[GR_TI * GR_DC];
  __shared__ __attribute__((aligned(16))) double xj[GR_DC * GR_LD];
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int64_t jbase = (int64_t)blockIdx.x * GR_TJ, j0 = jbase + 2 * tid, i0 = (int64_t)blockIdx.y * GR_TI;
  const double s2 = hyp[HYP_VAR];
  const double* ils = hyp + HYP_ILS;
  double ra[GR_TI], rb[GR_TI];
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) ra[ii] = rb[ii] = 0.0;
  const int sdd = tid % GR_DC, srow = tid / GR_DC;      // staging role: dimension sdd of rows srow + 32 q
  for (int d0 = 0; d0 < D; d0 += GR_DC) {
    const int dc = (D - d0 < GR_DC) ? D - d0 : GR_DC;
    if (d0 > 0) __syncthreads();
    const int dq = d0 + (sdd < dc ? sdd : dc - 1);                       // clamped: every load is unconditional
    const double sc = sdd < dc ? ils[dq] : 0.0;
    double st[GR_TJ / 32];
#pragma unroll
    for (int q = 0; q < GR_TJ / 32; ++q) {
      int64_t row = jbase + srow + 32 * q;
      if (row > n2 - 1) row = n2 - 1;
      st[q] = X2[row * D + dq];
    }
    if (tid < GR_TI * GR_DC) {
      const int ii = tid / GR_DC;
      const int64_t row = (i0 + ii < n) ? i0 + ii : n - 1;
      xi[tid] = X[row * D + dq] * sc;
    }
    // products are rounded by the LDS store on BOTH sides (no fused z - x * il): (i, j) and (j, i) see the same bits and
    // K(X, X) is exactly symmetric
#pragma unroll
    for (int q = 0; q < GR_TJ / 32; ++q) xj[sdd * GR_LD + srow + 32 * q] = st[q] * sc;
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < GR_DC; ++dd) {
      const d2 xx = *reinterpret_cast<const d2*>(&xj[dd * GR_LD + 2 * tid]);
#pragma unroll
      for (int ii = 0; ii < GR_TI; ++ii) {
        const double z = xi[ii * GR_DC + dd];
        const double da = z - xx[0], db = z - xx[1];
        ra[ii] = fma(da, da, ra[ii]);
        rb[ii] = fma(db, db, rb[ii]);
      }
    }
  }
  if (j0 >= n2) return;
  const bool pair = (j0 + 1 < n2) && ((ld & 1) == 0) && (((uintptr_t)out & 15) == 0);       // 16-byte aligned pair store
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) {          // values first (straight-line: the exps of all rows interleave) ...
    const int64_t i = i0 + ii;
    ra[ii] = kern_val<KIND>(ra[ii], s2) + ((symmetric && i == j0) ? diag_add : 0.0);
    rb[ii] = kern_val<KIND>(rb[ii], s2) + ((symmetric && i == j0 + 1) ? diag_add : 0.0);
  }
#pragma unroll
  for (int ii = 0; ii < GR_TI; ++ii) {          // ... then the stores
    const int64_t i = i0 + ii;
    if (i < n) {
      double* o = out + i * ld + j0;
      if (pair) {
        __builtin_nontemporal_store((d2){ra[ii], rb[ii]}, reinterpret_cast<d2*>(o));
      } else {
        o[0] = ra[ii];
        if (j0 + 1 < n2) o[1] = rb[ii];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// D <= 32: the distances through the fp64 MFMA pipe.  The form above spends 2 D + ~28 fp64 VALU operations per 8 output bytes — at
// D = 8 that is the VALU limit and the HBM limit at the same point (~7.7 TB/s), at D = 30 the VALU caps it at 3.7 TB/s (measured
// 3.5 and 1.5).  Here r2 = |x_i|^2 + |x_j|^2 - 2 G_ij with the Gram block G on the MFMA pipe (ceil(D / 4) instructions per
// 16 x 16 outputs), clamped at 0 and exactly 0 on the diagonal of K(X, X) — the form of the chains' Kuf tile and of the head
// launch's Ku.  What is left per output is the kernel function: the store stream is the bound.
//  * a wave owns 16 rows (i) and walks 32-column tiles: TWO accumulators whose B operands are the X2 rows j0 + 2c and j0 + 2c + 1, so
//    that lane (g, c) ends with the ADJACENT outputs (i, j0 + 2c), (i, j0 + 2c + 1) of rows i = g + 4t: one 16-byte non-temporal
//    store per row, 256 contiguous bytes per 16 lanes;
//  * K(X, X) stays exactly symmetric: the scaled coordinates are rounded once (x * (1 / l)), both norms of a pair come from the same
//    sequential sum (scaled_norm), G_ij and G_ji add the same products in the same order.
// Two launch forms of this tile arithmetic follow (the round-3 form — column chunks of <= 16 tiles staged at once, 2096 workgroups at
// 1024 x 50 000 — was replaced by them in round 5).
__device__ __forceinline__ double scaled_norm(const double* __restrict__ row, int D, const double* __restrict__ ils) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double v = row[d] * ils[d];
    s = fma(v, v, s);
  }
  return s;
}
// Form 1 (k_gram_mfma2) — equal work per workgroup in ONE round.  The round-3 launch cut the columns into chunks of <= 16 tiles
// staged at once (30 KB of LDS at D = 8: five workgroups per CU, 2096 workgroups at 1024 x 50 000 = 1.6 rounds, every workgroup 40 %
// prologue: stage -> barrier -> norms -> barrier -> A rows -> 12 tiles: 100 us).  Here (94 us; 512 x 40 960 x 30: 87 -> 69 us)
//  * grid (gx, row blocks of 64) with gx x row blocks <= the resident workgroups (256 CUs x occupancy): workgroup (x, y) owns the
//    column tiles [x ct / gx, (x + 1) ct / gx) of row block y — within one tile of each other, no second round, no tail;
//  * the tiles are staged in sub-chunks of SC (20 KB of LDS at D <= 8: the register file, not the LDS, sets the occupancy): thread t
//    fetches column t of the NEXT sub-chunk into registers (coalesced: consecutive threads read consecutive D-double rows) while the
//    waves work on the current one, then scales, stores and sums its norm in the order scaled_norm uses (K(X, X) stays exactly
//    symmetric: same roundings on both sides);
//  * the A rows and their norms are requested before the first staging barrier.
template <int KIND, int KS, int SC>
__global__ __launch_bounds__(256) void k_gram_mfma2(const double* __restrict__ X, int64_t n, const double* __restrict__ X2, int64_t n2,
                                                    int D, const double* __restrict__ hyp, double diag_add, int symmetric,
                                                    double* __restrict__ out, int64_t ld, int64_t ct) {
  extern __shared__ __attribute__((aligned(16))) double gm_dyn[];
  constexpr int LDP = 4 * KS + 1, COLS = SC * 32;
  static_assert(COLS <= 256, "one staging thread per column");
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  double* xs = gm_dyn;
  double* nrm = gm_dyn + (size_t)COLS * LDP;
  const double s2 = hyp[HYP_VAR];
  const double* __restrict__ ils = hyp + HYP_ILS;
  const int64_t t_lo = (int64_t)blockIdx.x * ct / gridDim.x, t_hi = (int64_t)(blockIdx.x + 1) * ct / gridDim.x;
  const int64_t i0 = (int64_t)blockIdx.y * 64 + 16 * wave;
  const bool row_ok = i0 < n;
  // next sub-chunk's column of this thread (clamped, unconditional loads; masked when committed)
  double pre[4 * KS];
  bool pre_ok = false;
  auto prefetch = [&](int64_t tile0) {
    if (tid >= COLS) return;
    const int64_t col = tile0 * 32 + tid;
    pre_ok = col < n2;
    const double* __restrict__ src = X2 + (pre_ok ? col : n2 - 1) * D;
#pragma unroll
    for (int d = 0; d < 4 * KS; ++d) pre[d] = src[d < D ? d : D - 1];
  };
  auto commit = [&]() {
    if (tid >= COLS) return;
    double sn = 0.0;
#pragma unroll
    for (int d = 0; d < 4 * KS; ++d) {
      const double v = (pre_ok && d < D) ? pre[d] * ils[d < D ? d : D - 1] : 0.0;
      xs[tid * LDP + d] = v;
      sn = fma(v, v, sn);            // zeros beyond D add nothing: the sum of scaled_norm, term by term
    }
    nrm[tid] = sn;
  };
  prefetch(t_lo);
  // A operand: row i0 + c scaled the same way, k = 4 s + g; its norm by the same sequential sum as the columns'
  double a[KS], ni[4];
  {
    const int64_t ir = (i0 + c < n) ? i0 + c : n - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = (4 * s + g < D) ? X[ir * D + 4 * s + g] * ils[4 * s + g] : 0.0;
    const double nr = scaled_norm(X + ir * D, D, ils);
#pragma unroll
    for (int t = 0; t < 4; ++t) ni[t] = __shfl(nr, g + 4 * t);
  }
  const bool even_ld = (ld & 1) == 0 && (((uintptr_t)out & 15) == 0);
  const bool full_rows = i0 + 16 <= n;
  double* __restrict__ orow = out + (i0 + g) * ld + 2 * c;         // this lane's outputs of a tile: orow + j0 + 4 t ld (t = 0..3)
  const int64_t ld4 = 4 * ld;
  for (int64_t tc = t_lo; tc < t_hi; tc += SC) {
    __syncthreads();                 // the previous sub-chunk is consumed
    commit();
    __syncthreads();
    if (tc + SC < t_hi) prefetch(tc + SC);
    const int nt = (int)((t_hi - tc < SC) ? t_hi - tc : SC);
    if (!row_ok) continue;
    for (int jt = 0; jt < nt; ++jt) {
      const int64_t j0 = (tc + jt) * 32;
      const double* __restrict__ pa = xs + (32 * jt + 2 * c) * LDP + g;
      d4 GA = (d4){0, 0, 0, 0}, GB = (d4){0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        GA = mfma_f64(a[s], pa[4 * s], GA);
        GB = mfma_f64(a[s], pa[LDP + 4 * s], GB);
      }
      const double nA = nrm[32 * jt + 2 * c], nB = nrm[32 * jt + 2 * c + 1];
      // whole tile inside the matrix, 16-byte stores possible, no diagonal element in it (wave-uniform): nothing but the kernel
      // function per output, two rows (four values) at a time, each pair stored as soon as it exists (all eight values interleaved
      // hold ~20 more registers live, which beside the staging registers costs waves per SIMD)
      const bool fast = even_ld && full_rows && j0 + 32 <= n2 && !(symmetric && j0 < i0 + 16 && j0 + 32 > i0);
      if (fast) {
        double* __restrict__ o = orow + j0;
#pragma unroll
        for (int tp = 0; tp < 4; tp += 2) {
          const double k0 = kern_val<KIND>(fmax(ni[tp] + nA - 2.0 * GA[tp], 0.0), s2);
          const double k1 = kern_val<KIND>(fmax(ni[tp] + nB - 2.0 * GB[tp], 0.0), s2);
          const double k2 = kern_val<KIND>(fmax(ni[tp + 1] + nA - 2.0 * GA[tp + 1], 0.0), s2);
          const double k3 = kern_val<KIND>(fmax(ni[tp + 1] + nB - 2.0 * GB[tp + 1], 0.0), s2);
          __builtin_nontemporal_store((d2){k0, k1}, reinterpret_cast<d2*>(o + tp * ld4));
          __builtin_nontemporal_store((d2){k2, k3}, reinterpret_cast<d2*>(o + (tp + 1) * ld4));
          __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
      // edge / diagonal tiles: one row group at a time (rolled: this path must not set the kernel's register count)
      const int64_t jA = j0 + 2 * c, jB = jA + 1;
      const double nit[4] = {ni[0], ni[1], ni[2], ni[3]};
#pragma unroll 1
      for (int t = 0; t < 4; ++t) {
        const int64_t i = i0 + g + 4 * t;
        const double gat = t == 0 ? GA[0] : (t == 1 ? GA[1] : (t == 2 ? GA[2] : GA[3]));
        const double gbt = t == 0 ? GB[0] : (t == 1 ? GB[1] : (t == 2 ? GB[2] : GB[3]));
        const double nt_ = t == 0 ? nit[0] : (t == 1 ? nit[1] : (t == 2 ? nit[2] : nit[3]));
        double ra = fmax(nt_ + nA - 2.0 * gat, 0.0), rb = fmax(nt_ + nB - 2.0 * gbt, 0.0);
        const bool da = symmetric && i == jA, db = symmetric && i == jB;
        if (da) ra = 0.0;
        if (db) rb = 0.0;
        const double ka = kern_val<KIND>(ra, s2) + (da ? diag_add : 0.0);
        const double kb = kern_val<KIND>(rb, s2) + (db ? diag_add : 0.0);
        if (i < n && jA < n2) {
          double* o = out + i * ld + jA;
          if (even_ld && jB < n2) {
            __builtin_nontemporal_store((d2){ka, kb}, reinterpret_cast<d2*>(o));
          } else {
            o[0] = ka;
            if (jB < n2) o[1] = kb;
          }
        }
      }
    }
  }
}

// Form 2 (k_gram_mfma3, D <= 8 and large results) — no workgroup barrier in the streaming loop.  In form 1 the four waves of a workgroup meet at two
// barriers per staged sub-chunk, and a wave held up by the store queue holds up the other three (1024 x 50 000: 94 us against 76 us for
// the bare store pattern, and no faster on all-zero inputs: not the arithmetic).  Here the workgroup stages its 64 A rows once (the
// only barrier); after that every wave is on its own: it owns the column tiles t_lo + wave, + 4, ... of the workgroup's range, stages
// one tile at a time in a wave-private LDS slot (lane (h, l): dimensions [h 2 KS, (h + 1) 2 KS) of column l — the next tile's values wait in
// 2 KS registers while the current one is worked on; the norm's summation chain runs through the lower half-wave and is continued by the
// upper one, term for term the sum of scaled_norm) and walks the four 16-row groups of the block for each tile with the B operands in
// registers.  Same tile arithmetic, same 4 x 256-byte non-temporal store per row group; adjacent waves write adjacent 256-byte segments.
template <int KIND, int KS>
__global__ __launch_bounds__(256) void k_gram_mfma3(const double* __restrict__ X, int64_t n, const double* __restrict__ X2, int64_t n2,
                                                    int D, const double* __restrict__ hyp, double diag_add, int symmetric,
                                                    double* __restrict__ out, int64_t ld, int64_t ct) {
  extern __shared__ __attribute__((aligned(16))) double gm_dyn[];
  constexpr int LDP = 4 * KS + 1, HK = 2 * KS;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int half = lane >> 5, lc = lane & 31;
  double* as = gm_dyn;                                   // [64][LDP] scaled A rows of this block
  double* na = as + 64 * LDP;                            // [64] their norms
  double* il = na + 64;                                  // [4 KS] 1 / lengthscale, zero beyond D (read per tile: NOT from global memory —
                                                         // a vector load in the streaming loop waits behind every store issued before it)
  double* xw = il + 4 * KS + wave * (32 * LDP + 32);     // wave-private: [32][LDP] scaled columns of the current tile
  double* nw = xw + 32 * LDP;                            //               [32] their norms
  const double s2 = hyp[HYP_VAR];
  const double* __restrict__ ils = hyp + HYP_ILS;
  const int64_t t_lo = (int64_t)blockIdx.x * ct / gridDim.x, t_hi = (int64_t)(blockIdx.x + 1) * ct / gridDim.x;
  const int64_t i0b = (int64_t)blockIdx.y * 64;
  double pre[HK];
  bool pre_ok = false;
  auto prefetch = [&](int64_t tile) {
    const int64_t col = tile * 32 + lc;
    pre_ok = col < n2;
    const double* __restrict__ src = X2 + (pre_ok ? col : n2 - 1) * D;
    // Issued as inline assembly and waited for by hand (wait_prefetch): vmcnt retires in issue order, so the values are there once at
    // most the 16 stores of the tile issued behind these loads are outstanding.  Left to the compiler the wait is s_waitcnt vmcnt(0) —
    // it cannot know how many stores follow the loads — i.e. a full drain of the wave's store queue (write acknowledgements under a
    // saturated HBM) before every tile: 4.5 instead of 5.x TB/s at 1024 x 50 000.
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      const int d = half * HK + k;
      const double* __restrict__ pk = src + (d < D ? d : D - 1);
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre[k]) : "v"(pk) : "memory");
    }
  };
  // stores_behind: the tile worked on since the prefetch took the all-fast path (exactly 16 store instructions issued after the loads)
  auto wait_prefetch = [&](bool stores_behind) {
    if (stores_behind) {
      if constexpr (HK == 4) asm volatile("s_waitcnt vmcnt(16)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3])::"memory");
      else if constexpr (HK == 8)
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]), "+v"(pre[6]), "+v"(pre[7])::"memory");
      else
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]), "+v"(pre[6]), "+v"(pre[7]),
                     "+v"(pre[8]), "+v"(pre[9]), "+v"(pre[10]), "+v"(pre[11]), "+v"(pre[12]), "+v"(pre[13]), "+v"(pre[14]), "+v"(pre[15])::"memory");
    } else {
      if constexpr (HK == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3])::"memory");
      else if constexpr (HK == 8)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]), "+v"(pre[6]), "+v"(pre[7])::"memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]), "+v"(pre[6]), "+v"(pre[7]),
                     "+v"(pre[8]), "+v"(pre[9]), "+v"(pre[10]), "+v"(pre[11]), "+v"(pre[12]), "+v"(pre[13]), "+v"(pre[14]), "+v"(pre[15])::"memory");
    }
  };
  auto commit = [&]() {
    double v[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      const int d = half * HK + k;
      v[k] = pre_ok ? pre[k] * il[d] : 0.0;
      xw[lc * LDP + d] = v[k];
    }
    double s0 = 0.0;
#pragma unroll
    for (int k = 0; k < HK; ++k) s0 = fma(v[k], v[k], s0);
    double s1 = __shfl(s0, lc);                 // the lower half-wave's partial sum of this column ...
#pragma unroll
    for (int k = 0; k < HK; ++k) s1 = fma(v[k], v[k], s1);          // ... continued by the upper half-wave with its dimensions
    if (half) nw[lc] = s1;
  };
  int64_t t = t_lo + wave;
  if (t < t_hi) prefetch(t);
  if (tid >= 64 && tid < 64 + 4 * KS) il[tid - 64] = (tid - 64 < D) ? ils[tid - 64] : 0.0;
  if (tid < 64) {
    const int64_t ir = (i0b + tid < n) ? i0b + tid : n - 1;
    double sn = 0.0;
#pragma unroll
    for (int d = 0; d < 4 * KS; ++d) {
      const double v = (d < D) ? X[ir * D + d] * ils[d < D ? d : D - 1] : 0.0;
      as[tid * LDP + d] = v;
      sn = fma(v, v, sn);
    }
    na[tid] = sn;
  }
  __syncthreads();
  const bool even_ld = (ld & 1) == 0 && (((uintptr_t)out & 15) == 0);
  const int64_t ld4 = 4 * ld;
  bool stores_behind = false;         // (the first tile's values: waited for with vmcnt(0))
  for (; t < t_hi; t += 4) {
    wait_prefetch(stores_behind);
    commit();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (t + 4 < t_hi) prefetch(t + 4);
    double bA[KS], bB[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bA[s] = xw[(2 * c) * LDP + 4 * s + g];
      bB[s] = xw[(2 * c + 1) * LDP + 4 * s + g];
    }
    const double nA = nw[2 * c], nB = nw[2 * c + 1];
    const int64_t j0 = t * 32;
    // whole tile column inside the matrix for all four row groups, 16-byte stores possible, no diagonal element in it (wave-uniform):
    // nothing but the kernel function per output, and EXACTLY 16 store instructions (what wait_prefetch counts on)
    const bool fast = even_ld && i0b + 64 <= n && j0 + 32 <= n2 && !(symmetric && j0 < i0b + 64 && j0 + 32 > i0b);
    stores_behind = fast;
    if (fast) {
#pragma unroll 1
      for (int rg = 0; rg < 4; ++rg) {
        d4 GA = (d4){0, 0, 0, 0}, GB = (d4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double a = as[(16 * rg + c) * LDP + 4 * s + g];
          GA = mfma_f64(a, bA[s], GA);
          GB = mfma_f64(a, bB[s], GB);
        }
        double ni[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) ni[tt] = na[16 * rg + g + 4 * tt];
        // two rows (four kernel values) at a time, each pair stored as soon as it exists
        double* __restrict__ o = out + (i0b + 16 * rg + g) * ld + j0 + 2 * c;
#pragma unroll
        for (int tp = 0; tp < 4; tp += 2) {
          const double k0 = kern_val<KIND>(fmax(ni[tp] + nA - 2.0 * GA[tp], 0.0), s2);
          const double k1 = kern_val<KIND>(fmax(ni[tp] + nB - 2.0 * GB[tp], 0.0), s2);
          const double k2 = kern_val<KIND>(fmax(ni[tp + 1] + nA - 2.0 * GA[tp + 1], 0.0), s2);
          const double k3 = kern_val<KIND>(fmax(ni[tp + 1] + nB - 2.0 * GB[tp + 1], 0.0), s2);
          __builtin_nontemporal_store((d2){k0, k1}, reinterpret_cast<d2*>(o + tp * ld4));
          __builtin_nontemporal_store((d2){k2, k3}, reinterpret_cast<d2*>(o + (tp + 1) * ld4));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // edge / diagonal tiles: one row at a time (rolled: this path must not set the kernel's register count)
      const int64_t jA = j0 + 2 * c, jB = jA + 1;
#pragma unroll 1
      for (int rg = 0; rg < 4; ++rg) {
        const int64_t i0 = i0b + 16 * rg;
        if (i0 >= n) break;
        d4 GA = (d4){0, 0, 0, 0}, GB = (d4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double a = as[(16 * rg + c) * LDP + 4 * s + g];
          GA = mfma_f64(a, bA[s], GA);
          GB = mfma_f64(a, bB[s], GB);
        }
#pragma unroll 1
        for (int tt = 0; tt < 4; ++tt) {
          const int64_t i = i0 + g + 4 * tt;
          const double gat = tt == 0 ? GA[0] : (tt == 1 ? GA[1] : (tt == 2 ? GA[2] : GA[3]));
          const double gbt = tt == 0 ? GB[0] : (tt == 1 ? GB[1] : (tt == 2 ? GB[2] : GB[3]));
          const double nt_ = na[16 * rg + g + 4 * tt];
          double ra = fmax(nt_ + nA - 2.0 * gat, 0.0), rb = fmax(nt_ + nB - 2.0 * gbt, 0.0);
          const bool da = symmetric && i == jA, db = symmetric && i == jB;
          if (da) ra = 0.0;
          if (db) rb = 0.0;
          const double ka = kern_val<KIND>(ra, s2) + (da ? diag_add : 0.0);
          const double kb = kern_val<KIND>(rb, s2) + (db ? diag_add : 0.0);
          if (i < n && jA < n2) {
            double* o = out + i * ld + jA;
            if (even_ld && jB < n2) {
              __builtin_nontemporal_store((d2){ka, kb}, reinterpret_cast<d2*>(o));
            } else {
              o[0] = ka;
              if (jB < n2) o[1] = kb;
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();      // (the tile's LDS reads are issued before the next commit's stores: same wave, program order)
  }
}

template <int KIND>
static void gram_mfma3_go(hipStream_t st, int ks, const double* X, int64_t n, const double* X2, int64_t n2, int D, const double* hyp,
                          double diag_add, int symmetric, double* out, int64_t ld) {
  const int64_t ct = ceil_div(n2, 32), rb = ceil_div(n, 64);
  const int KSp = ks <= 2 ? 2 : (ks <= 4 ? 4 : 8);
  const size_t lds = ((size_t)64 * (4 * KSp + 1) + 64 + 4 * KSp + 4 * (32 * (4 * KSp + 1) + 32)) * sizeof(double);
  // 8 workgroups per CU: two rounds at the four waves per SIMD the instance's 106 VGPRs allow.  Sweep at 1024 x 50 000 (three processes
  // each, the spread between processes is +- 4 us): 4 per CU 83 - 93 us, 6: 87 - 91, 8: 83 - 84, 10: 83 - 87, 12: 89 - 93, 16: 84 - 89
  const int occ = 8;
  int64_t gx = std::max<int64_t>(1, (int64_t)256 * occ / rb);
  gx = std::min(gx, std::max<int64_t>(1, ct / 8));           // small problems: at least two tiles per wave
  gx = std::min(gx, ct);
  dim3 grid((unsigned)gx, (unsigned)rb);
  if (KSp == 2)
    DS_LAUNCH((k_gram_mfma3<KIND, 2>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
  else if (KSp == 4)
    DS_LAUNCH((k_gram_mfma3<KIND, 4>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
  else
    DS_LAUNCH((k_gram_mfma3<KIND, 8>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
}

template <int KIND>
static void gram_mfma2_go(hipStream_t st, int ks, const double* X, int64_t n, const double* X2, int64_t n2, int D, const double* hyp,
                          double diag_add, int symmetric, double* out, int64_t ld) {
  const int64_t ct = ceil_div(n2, 32), rb = ceil_div(n, 64);
  const int KSp = ks <= 2 ? 2 : (ks <= 4 ? 4 : 8);
  const int sc = KSp == 8 ? 4 : 8;
  const size_t lds = ((size_t)sc * 32 * (4 * KSp + 1) + sc * 32) * sizeof(double);
  // resident workgroups per CU, by the registers of the instance (93 / 115 / 174 VGPRs at KS = 2 / 4 / 8: 5 / 4 / 2 waves per SIMD;
  // Matern-5/2 one step lower at KS = 2) and by the LDS of a sub-chunk
  const int occ_reg = KSp == 2 ? (KIND == DSDGP_KERN_RBF ? 5 : 4) : (KSp == 4 ? 4 : 2);
  const int occ = (int)std::min<size_t>(occ_reg, (160 << 10) / (lds + 512));
  int64_t gx = std::max<int64_t>(1, (int64_t)256 * occ / rb);
  gx = std::min(gx, std::max<int64_t>(1, ct / 4));           // small problems: at least four tiles per workgroup
  gx = std::min(gx, ct);
  dim3 grid((unsigned)gx, (unsigned)rb);
  if (KSp == 2)
    DS_LAUNCH((k_gram_mfma2<KIND, 2, 8>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
  else if (KSp == 4)
    DS_LAUNCH((k_gram_mfma2<KIND, 4, 8>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
  else
    DS_LAUNCH((k_gram_mfma2<KIND, 8, 4>), grid, dim3(256), lds, st, X, n, X2, n2, D, hyp, diag_add, symmetric, out, ld, ct);
}

#ifndef DSDGP_GRAM_STREAM_VALIDATED
#define DSDGP_GRAM_STREAM_VALIDATED 0
#endif
int gram_launch(dsdgp_ctx* ctx, int kind, const double* X, int64_t n, const double* X2, int64_t n2, int D,
                const double* hyp_dev, double diag_add, int symmetric, double* out, int64_t ld) {
  ProfScope ps(ctx, "gram");
  if (D <= 32) {
    // D <= 8 and a large result (>= 8192 tiles of 64 x 32): the barrier-free streaming launch; else the sub-chunk launch (measured,
    // tools/gram_time.py: 1024 x 50 000 x 8: 83 - 84 us against 93 - 94; 128 x 20 000 x 8: 16.2 against 14.1; 256 x 40 000 x 9: 30.7
    // against 26.9; 512 x 40 960 x 30: 75 against 69.5)
    const int ks = ceil_div(D, 4);
    // The streaming form waits for its prefetched column values with a HAND-COUNTED s_waitcnt (k_gram_mfma3: the loads are inline
    // assembly, the compiler does not know they are in flight).  That is only as good as the ISA the compiler emitted around it, so the
    // form is on by default only in a library built with the compiler it was validated with (Makefile: DSDGP_GRAM_STREAM_VALIDATED — the
    // bitwise test of both forms, tests/test_gpu_round5.py::test_gram_both_forms_agree_bitwise_on_a_shared_block, green with that hipcc);
    // DSDGP_GRAM_STREAM=0 / 1 overrides either way (1 after running that test with the new compiler).
    static const bool stream_ok = getenv("DSDGP_GRAM_STREAM") ? atoi(getenv("DSDGP_GRAM_STREAM")) != 0 : (DSDGP_GRAM_STREAM_VALIDATED != 0);
    const bool stream = stream_ok && ks <= 2 && (int64_t)ceil_div(n2, 32) * ceil_div(n, 64) >= 8192;
    if (kind == DSDGP_KERN_RBF) {
      if (stream) gram_mfma3_go<DSDGP_KERN_RBF>(ctx->stream, ks, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld);
      else gram_mfma2_go<DSDGP_KERN_RBF>(ctx->stream, ks, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld);
    } else {
      if (stream) gram_mfma3_go<DSDGP_KERN_MATERN52>(ctx->stream, ks, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld);
      else gram_mfma2_go<DSDGP_KERN_MATERN52>(ctx->stream, ks, X, n, X2, n2, D, hyp_dev, diag_add, symmetric, out, ld);
    }
    DS_HIP(hipGetLastError());
    return DSDGP_OK;
  }
  dim3 grid(ceil_div(n2, GR_TJ), ceil_div(n, GR_TI));
  if (kind == DSDGP_KERN_RBF)
    DS_LAUNCH(k_gram<DSDGP_KERN_RBF>, grid, dim3(256), 0, ctx->stream, X, n, X2, n2, D, hyp_dev, diag_add,
                       symmetric, out, ld);
  else
    DS_LAUNCH(k_gram<DSDGP_KERN_MATERN52>, grid, dim3(256), 0, ctx->stream, X, n, X2, n2, D, hyp_dev,
                       diag_add, symmetric, out, ld);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

extern "C" int dsdgp_gram(dsdgp_ctx* ctx, const dsdgp_kernel* kern, const double* X, int64_t n, const double* X2,
                          int64_t n2, double jitter, double* out, int64_t ld_out) {
  DS_CHECK_ARG(ctx && kern && X && out && n > 0);
  DS_CHECK_ARG(kern->kind == DSDGP_KERN_RBF || kern->kind == DSDGP_KERN_MATERN52);
  DS_CHECK_ARG(kern->input_dim > 0 && kern->lengthscales);
  const int D = kern->input_dim;
  const int symmetric = (X2 == nullptr);
  if (symmetric) {
    X2 = X;
    n2 = n;
  }
  DS_CHECK_ARG(n2 > 0 && ld_out >= n2);
  std::vector<double> hyp(HYP_ILS + 2 * D, 0.0);
  hyp[HYP_VAR] = kern->variance;
  hyp[HYP_WVAR] = kern->has_white ? kern->white_variance : 0.0;
  hyp[HYP_KDIAG] = hyp[HYP_VAR] + hyp[HYP_WVAR];
  for (int j = 0; j < D; ++j) hyp[HYP_ILS + j] = 1.0 / kern->lengthscales[kern->ard ? j : 0];
  void* scr;
  DS_TRY(ctx_scratch(ctx, hyp.size() * sizeof(double), &scr));
  DS_TRY(ctx_upload(ctx, scr, hyp.data(), hyp.size() * sizeof(double)));     // asynchronous (pinned staging ring)
  const double diag_add = symmetric ? (hyp[HYP_WVAR] + jitter) : 0.0;
  return gram_launch(ctx, kern->kind, X, n, X2, n2, D, (const double*)scr, diag_add, symmetric, out, ld_out);
}
