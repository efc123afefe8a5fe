// SVGP_Layer.conditional_ND + reparameterize (layers.py:178-219, utils.py:40-41) and their reverse mode for LARGE inducing counts,
// formulated as whole-layer MFMA GEMMs (see layer.hpp, "GEMM-formulated layer passes").  gfx950 only.
//
// forward (white = False; math of DESIGN.md section 2):
//   K   = k(Z, X)                       k_kuf<KIND, false, TS>        (Mp x ld, rows >= M and columns >= Rin are zero)
//   a1  = Lu^-1 K                       k_pgemm, W lower-triangular   + column sums of squares per 128-row tile  -> |a1|^2
//   a   = Lu^-T a1                      k_pgemm, W upper-triangular   -> Asave
//   c_d = q_sqrt_d^T a   (all d)        k_pgemm, batched, W upper     + column sums of squares -> |c_d|^2 ; stored only for the reverse pass
//   mu  = q_mu^T a (+ mean_A^T X^T)     k_thin on the transposed, padded q_mu (k_pgemm above 32 outputs)
//   mean / var / F                      k_gl_epilogue  (var = kdiag - |a1|^2 + |c_d|^2, utils.py:41)
//   (products of a launch too small to fill a good part of the chip are cut along k: pgemm_split, k_gl_sum, k_gl_colsq)
// backward:
//   abar = sum_d q_sqrt_d (2 vbar_d c_d) + q_mu mbar     ONE k_pgemm: W lower-triangular, B scaled per column, batch items summed, the
//                                                        q_mu mbar term as tail product; the (output, k) steps of a tile in up to 8
//                                                        pieces when the tiles alone would not fill the chip (partial sums in Pb)
//          (no c_d kept: abar = sum_d S_d (2 vbar_d a), dense)
//   b    = Ku^-1 abar                                    k_gl_sum (the pieces, fixed order), k_pgemm
//   e = b - g a, kbar = e - g a, GW = kbar dk/dr2, E      k_kuf<KIND, true, TS> (recomputes r2), sum kbar k per block
//   ZZ^T GW with ZZ = [Z/l | (Z/l)^2 | 1]                k_thin (k_pgemm for wide inputs) -> sums over the inducing rows for dX and the lengthscales
//   dX / transposed adjoints of the layer below, hyp_part k_gl_bwd_rows
// white = True (layers.py:186-188 without the second solve): a1 stands where a does (no Lu^-T product forward); backward
//   a1bar = abar - 2 (sum_d vbar_d) a1 (k_gl_white_abar), kbar = Lu^-T a1bar (ONE triangular k_pgemm instead of the dense Ku^-1 abar), e = kbar
#include <type_traits>

#include "layer.hpp"
#include <algorithm>

#define PT 128     // output tile (both dimensions)
#define PK 16      // k per staging step
#define PLD 144    // LDS row stride in doubles: the g and g + 1 k-rows of a fragment read sit 32 banks apart (conflict-free ds_read_b64)
typedef double d2 __attribute__((ext_vector_type(2)));

// One product of a layer pass:  C[o] = alpha * sum over a range of (batch item, k) steps of  W[b] (m x k, row-major) B'[b] (k x n)
//   B' = B (k x n, row-major) with its columns scaled by bs_mul * bscale[b][col] (or as it is).
// reduce_batch = 0: every batch item is its own output o = b;  1: the batch items are summed into ONE output (o = 0), followed by the
//   TAIL product W2 (m x k2) B2 (k2 x n) (unscaled, dense: the q_mu mbar term of abar) when k2 > 0.
// groups >= 1: the step range of an output is cut into `groups` contiguous pieces, piece g goes to C + o sC + g sCg (partial sums:
//   split-K / split-batch; the caller adds them in a fixed order).  colsq (groups == 1 only): colsq[(o tiles_m + tile row) ldq + col] =
//   sum over the tile's rows of C^2.
// tri = 2: W lower-triangular (k ranges over [0, tile row's last row]) ; 8: upper-triangular (k from the tile's first row) ; 0: dense.
struct PGemm {
  const double* W;
  const double* B;
  double* C;
  const double* bscale;
  double* colsq;
  const double* W2;
  const double* B2;
  int64_t ldw, ldb, ldc, sW, sB, sC, sS, ldq, sCg, ldw2;
  int32_t m, n, k, batch, reduce_batch, groups, tri, store, k2, accum;      // accum: C += alpha W B (dsdgp_trsm's panel updates)      // (pad: alignment)
  int32_t tiles_m, tiles_n;
  double alpha, bs_mul;     // bs_mul multiplies the column scales
};

// TN: tile width (128, or 64 for launches that would not fill the chip with 128-wide tiles: twice the tiles, the four waves take
// 64 x 32 each).  Two workgroups per CU either way (LDS), i.e. two waves per SIMD — what the fp64 MFMA pipe needs to run at its rate.
template <int TN>
__global__ __launch_bounds__(256) void k_pgemm(const PGemm P) {
  constexpr int PLB = TN + 16;         // LDS row stride of the B tile: like PLD, the k-rows g and g + 1 sit 32 banks apart
  constexpr int NB = TN / 16;          // doubles of a B row per staging thread (8 / 4)
  constexpr int NJ = TN / 32;          // 16-column blocks per wave
  __shared__ __attribute__((aligned(16))) double As[2][PK * PLD];
  __shared__ __attribute__((aligned(16))) double Bs[2][PK * PLB];
  const int G = P.groups > 1 ? P.groups : 1;
  const int nb_out = P.reduce_batch ? 1 : P.batch;
  const int Z = nb_out * G;
  int t = blockIdx.x;
  const int tn = t % P.tiles_n;
  t /= P.tiles_n;
  const int z = t % Z, tmi = t / Z;
  // workgroups are dispatched in index order: the tile rows with the longest k range go first (longest-processing-time order)
  const int tm = (P.tri == 2) ? P.tiles_m - 1 - tmi : tmi;
  const int o = z / G, gidx = z - o * G;
  const int m0 = tm * PT, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  d4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  int kmin = 0, kmax = P.k;
  if (P.tri == 2) kmax = min(kmax, m0 + PT);
  if (P.tri == 8) kmin = m0;
  const int ks_lo = kmin / PK;
  const int ksteps = max(0, (kmax + PK - 1) / PK - ks_lo);
  const int nmain = (P.reduce_batch ? P.batch : 1) * ksteps;
  const int T = nmain + ((P.reduce_batch && P.k2 > 0) ? (P.k2 + PK - 1) / PK : 0);
  const int s_lo = (int)((int64_t)gidx * T / G), s_hi = (int)((int64_t)(gidx + 1) * T / G);
  // staging roles: W rows (k contiguous): thread takes row m0 + tid / 2, k = 8 (tid & 1) .. + 7 ; B rows (n contiguous): k = tid / 16,
  // n = n0 + NB (tid & 15) .. + NB - 1.  16-byte loads.
  const int am = m0 + (tid >> 1), ak = 8 * (tid & 1);
  const int bk = tid >> 4, bn = n0 + NB * (tid & 15);
  const bool a_row = am < P.m, b_col = bn + NB <= P.n;
  double ra[8], rb[NB], sv[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) sv[u] = 1.0;
  int kq = 0;          // first k of the W chunk in flight (for the triangle mask applied when it is stored to LDS)
  int tri_q = 0;       // triangle mask of the chunk in flight (none on the tail product)
  double bs_q = 1.0;   // multiplier of the column scales of the chunk in flight (1 on the tail product)
  // gload only ISSUES the loads of a step; everything that consumes the loaded values (triangle mask, scales) happens in lstore,
  // i.e. after the MFMAs of the step that runs meanwhile — a use inside gload would put the memory round trip in front of them
  // FAST (the whole tile inside the matrices, every k chunk full): the loads are unconditional.  With the guards, the loaded registers
  // met the zero-filled alternative in a phi, the compiler resolved it with register COPIES of the loaded values — and a copy needs its
  // source: an `s_waitcnt vmcnt(0)` sat between the loads of step s + 1 and the MFMAs of step s, every step (ISA: sixteen v_mov_b64
  // behind the wait), i.e. the memory round trip the two-stage pipeline exists to hide was paid in full.
  auto gload = [&](int step, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const bool tail = step >= nmain;
    const int bi = tail ? 0 : step / ksteps;
    const int b = P.reduce_batch ? bi : o;
    const int k0 = tail ? (step - nmain) * PK : (ks_lo + step - bi * ksteps) * PK;
    const int kdim = tail ? P.k2 : P.k;
    const int64_t ldw = tail ? P.ldw2 : P.ldw;
    kq = k0 + ak;
    tri_q = tail ? 0 : P.tri;
    bs_q = tail ? 1.0 : P.bs_mul;
    {
      gcptr src = (gcptr)((tail ? P.W2 : P.W + (int64_t)b * P.sW) + (int64_t)am * ldw + k0 + ak);
      if (FAST || (a_row && k0 + ak + 8 <= kdim)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const d2 v = *reinterpret_cast<const d2 __attribute__((address_space(1)))*>(src + 2 * u);
          ra[2 * u] = v[0];
          ra[2 * u + 1] = v[1];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = (a_row && k0 + ak + u < kdim) ? src[u] : 0.0;
      }
    }
    {
      const bool ok = FAST || (b_col && k0 + bk < kdim);
      gcptr src = (gcptr)((tail ? P.B2 : P.B + (int64_t)b * P.sB) + (int64_t)(k0 + bk) * P.ldb + bn);
      if (FAST || ok) {
#pragma unroll
        for (int u = 0; u < NB / 2; ++u) {
          const d2 v = *reinterpret_cast<const d2 __attribute__((address_space(1)))*>(src + 2 * u);
          rb[2 * u] = v[0];
          rb[2 * u + 1] = v[1];
        }
      } else {
#pragma unroll
        for (int u = 0; u < NB; ++u) rb[u] = 0.0;
      }
      if (P.bscale) {
        if (tail) {
          if (step == nmain || step == s_lo) {
#pragma unroll
            for (int u = 0; u < NB; ++u) sv[u] = 1.0;
          }
        } else if ((step == bi * ksteps || step == s_lo) && b_col) {      // first k-step of batch item b in this piece: its column scales
          gcptr sp = (gcptr)(P.bscale + (int64_t)b * P.sS + bn);
#pragma unroll
          for (int u = 0; u < NB; ++u) sv[u] = sp[u];
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    const int mm = tid >> 1;
    // entries of the other triangle inside the diagonal tile are not trusted to be zero in memory
    if (tri_q == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) ra[u] = (kq + u <= am) ? ra[u] : 0.0;
    } else if (tri_q == 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) ra[u] = (kq + u >= am) ? ra[u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) As[buf][(ak + u) * PLD + mm] = ra[u];
    if (P.bscale) {
#pragma unroll
      for (int u = 0; u < NB; ++u) rb[u] *= sv[u] * bs_q;
    }
    double* brow = &Bs[buf][bk * PLB + NB * (tid & 15)];
#pragma unroll
    for (int u = 0; u < NB / 2; ++u) *reinterpret_cast<d2*>(brow + 2 * u) = (d2){rb[2 * u], rb[2 * u + 1]};
  };
  const bool fast = m0 + PT <= P.m && n0 + TN <= P.n && P.k % PK == 0 && (!(P.reduce_batch && P.k2 > 0) || P.k2 % PK == 0);
  auto run = [&](auto fast_tag) {
  if (s_hi > s_lo) {
    gload(s_lo, fast_tag);
    lstore(0);
  }
  __syncthreads();
  for (int step = s_lo; step < s_hi; ++step) {
    const int buf = (step - s_lo) & 1;
    if (step + 1 < s_hi) gload(step + 1, fast_tag);
#pragma unroll
    for (int k4 = 0; k4 < PK; k4 += 4) {
      double a[4], bq[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[buf][(k4 + g) * PLD + wr * 64 + 16 * i + c];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bq[j] = Bs[buf][(k4 + g) * PLB + wc * (TN / 2) + 16 * j + c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_f64(a[i], bq[j], acc[i][j]);
    }
    if (step + 1 < s_hi) lstore(buf ^ 1);     // the other buffer: its last readers passed the previous barrier
    __syncthreads();
  }
  };
  if (fast) run(std::true_type{});
  else run(std::false_type{});
  if (P.store) {
    gptr C = (gptr)(P.C + (int64_t)o * P.sC + (int64_t)gidx * P.sCg);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wr * 64 + ib * 16 + g + 4 * r;
          const int col = n0 + wc * (TN / 2) + jb * 16 + c;
          if (row < P.m && col < P.n) {
            const double v = P.alpha * acc[ib][jb][r];
            C[(int64_t)row * P.ldc + col] = P.accum ? C[(int64_t)row * P.ldc + col] + v : v;
          }
        }
  }
  if (P.colsq) {
    // column sums of squares over this tile's rows: lanes fold their 16 rows, the four row groups fold through permlane swaps, the
    // two waves that share the columns through LDS (fixed order).  Rows >= m hold zero (masked W rows).
    double* cs = &As[0][0];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      double s = 0.0;
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = P.alpha * acc[ib][jb][r];
          s = fma(v, v, s);
        }
      s = sum_groups(s);
      if (g == 0) cs[wr * TN + wc * (TN / 2) + jb * 16 + c] = s;
    }
    __syncthreads();
    if (tid < TN) {
      const int col = n0 + tid;
      if (col < P.n) P.colsq[((int64_t)o * P.tiles_m + tm) * P.ldq + col] = cs[tid] + cs[TN + tid];
    }
  }
}

// Thin products C (mt x n) = W (mt x k) B (k x n) with mt = 16 or 32 rows (q_mu^T a; ZZ^T GW at narrow inputs): memory-bound on B, which a
// 128-row tile would read for 4 - 8x the MFMA work it needs.  One workgroup per 32 columns, its four waves a quarter of k each with
// both operands straight from global memory (B: 128-byte runs per 16 lanes; W is small and cache-resident), partial tiles added
// through LDS in wave order.
#define THC 32
template <int NIT>
__global__ __launch_bounds__(256) void k_thin(const double* __restrict__ W, int64_t ldw, const double* __restrict__ B, int64_t ldb,
                                              double* __restrict__ C, int64_t ldc, int mt, int n, int k, int accumulate) {
  __shared__ double red[4][4][4][64];          // [wave][tile][reg][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.x * THC;
  constexpr int nit = NIT;                     // 1 or 2 row tiles (mt / 16)
  const int kq = (((k + 3) / 4 + 3) / 4) * 4;  // k per wave: a quarter of k rounded up to whole MFMA steps (k < 4: all of it in wave 0)
  const int k_lo = wave * kq, k_hi = min(k, k_lo + kq);
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  // lane c holds the ADJACENT columns n0 + 2c, n0 + 2c + 1 (one 16-byte load; column tile j = the columns of parity j): with 8-byte
  // loads and four steps in flight a CU had 32 KB outstanding, half of what the HBM latency asks for (config 5: 33 us for 51 MB)
  const int cc = n0 + 2 * c;
  const bool cin = cc < n;           // (n is even: thin_launch)
  const int colc = cin ? cc : 0;
  // TU steps per trip: every load of the trip is issued before its first MFMA (with the row-tile count a run-time value the loop
  // kept a branch, was not unrolled, and each step waited for its own loads: 30 us for a FOUR-workgroup launch)
  constexpr int TU = 8;
  for (int kk = k_lo; kk < k_hi; kk += 4 * TU) {
    d2 bv[TU];
    double a0[TU], a1[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int kr = kk + 4 * u + g;
      const int krc = kr < k_hi ? kr : k_lo;
      bv[u] = *reinterpret_cast<const d2*>(B + (int64_t)krc * ldb + colc);
      a0[u] = W[(int64_t)c * ldw + krc];
      if constexpr (NIT > 1) a1[u] = W[(int64_t)(16 + c) * ldw + krc];
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const bool kin = kk + 4 * u + g < k_hi;
      const double a0m = kin ? a0[u] : 0.0;
      const double b0v = cin ? bv[u].x : 0.0, b1v = cin ? bv[u].y : 0.0;
      acc[0][0] = mfma_f64(a0m, b0v, acc[0][0]);
      acc[0][1] = mfma_f64(a0m, b1v, acc[0][1]);
      if constexpr (NIT > 1) {
        const double a1m = kin ? a1[u] : 0.0;
        acc[1][0] = mfma_f64(a1m, b0v, acc[1][0]);
        acc[1][1] = mfma_f64(a1m, b1v, acc[1][1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][2 * i + j][r][lane] = acc[i][j][r];
  __syncthreads();
  // 4 tiles x 4 registers x 64 lanes = 1024 sums over the four waves, 4 per thread
  for (int e = tid; e < 1024; e += 256) {
    const int l = e & 63, r = (e >> 6) & 3, tl = e >> 8;
    const int i = tl >> 1, j = tl & 1;
    if (i >= nit) continue;
    const double v = (red[0][tl][r][l] + red[1][tl][r][l]) + (red[2][tl][r][l] + red[3][tl][r][l]);
    const int row = 16 * i + (l >> 4) + 4 * r, col = n0 + 2 * (l & 15) + j;
    if (col < n) C[(int64_t)row * ldc + col] = accumulate ? C[(int64_t)row * ldc + col] + v : v;
  }
}
static int thin_launch(dsdgp_ctx* ctx, const double* W, int64_t ldw, const double* B, int64_t ldb, double* C, int64_t ldc, int mt, int n, int k,
                       int accumulate = 0) {
  DS_CHECK_ARG((n & 1) == 0 && (ldb & 1) == 0 && ((uintptr_t)B & 15) == 0);        // (padded row counts: multiples of 16)
  if (mt > 16) DS_LAUNCH(k_thin<2>, dim3(ceil_div(n, THC)), dim3(256), 0, ctx->stream, W, ldw, B, ldb, C, ldc, mt, n, k, accumulate);
  else DS_LAUNCH(k_thin<1>, dim3(ceil_div(n, THC)), dim3(256), 0, ctx->stream, W, ldw, B, ldb, C, ldc, mt, n, k, accumulate);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

static int pgemm_launch(dsdgp_ctx* ctx, PGemm P) {
  P.tiles_m = ceil_div(P.m, PT);
  const int G = P.groups > 1 ? P.groups : 1;
  const int Z = (P.reduce_batch ? 1 : P.batch) * G;
  // 128-wide tiles when they fill the chip (two workgroups per CU), else 64-wide ones
  static const int narrow_below = getenv("DSDGP_PGEMM_NARROW_BELOW") ? atoi(getenv("DSDGP_PGEMM_NARROW_BELOW")) : 512;   // (A/B aid)
  const bool narrow = (int64_t)P.tiles_m * ceil_div(P.n, PT) * Z < narrow_below;
  P.tiles_n = ceil_div(P.n, narrow ? 64 : PT);
  const int64_t blocks = (int64_t)P.tiles_m * P.tiles_n * Z;
  if (blocks <= 0) return DSDGP_OK;
  if (narrow) DS_LAUNCH(k_pgemm<64>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, P);
  else DS_LAUNCH(k_pgemm<128>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, P);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// An independent product whose tiles would leave most of the chip idle — the 125-row first layer of a config-5 shard is 16 tiles of up to
// 64 dependent k-steps: ~105 us per launch whatever its size — is cut along k into G pieces (partial results in `pb`), added in a fixed
// order by k_gl_sum; the column sums of squares then come from the sum (k_gl_colsq).  Plain launch when the tiles fill a good part of
// the chip or k is short.
static int pgemm_split(dsdgp_ctx* ctx, PGemm P, double* pb, int64_t pb_doubles);
// C (m x n) += alpha W (m x k) B (k x n), row-major, through the LDS-tiled kernel (linalg.hpp: the panel updates of dsdgp_trsm).
// 16-byte staging loads: W, B 16-byte aligned, even leading dimensions, n a multiple of 8.
int pgemm_accum(dsdgp_ctx* ctx, const double* W, int64_t ldw, const double* B, int64_t ldb, double* C, int64_t ldc, int m, int n, int k, double alpha) {
  DS_CHECK_ARG(W && B && C && m > 0 && n > 0 && k > 0 && (n & 7) == 0 && (ldw & 1) == 0 && (ldb & 1) == 0 && (((uintptr_t)W | (uintptr_t)B) & 15) == 0);
  PGemm P{};
  P.W = W; P.B = B; P.C = C; P.ldw = ldw; P.ldb = ldb; P.ldc = ldc; P.m = m; P.n = n; P.k = k; P.batch = 1; P.store = 1; P.alpha = alpha; P.accum = 1;
  return pgemm_launch(ctx, P);
}

// ------------------------------------------------------------------------------------------------------
// Kuf tile kernel.  64 x 64 outputs per workgroup, thread (ty, tx) owns rows ty + 16 i, columns tx + 16 j; Z / l and x / l of the tile
// staged through LDS in chunks of KD input dimensions ([dimension][64 + 1], Z reads broadcast over tx, x reads conflict-free);
// squared distances by direct differences (D_in subtract + FMA pairs per output: fp64 VALU work that is < 1 % of the layer's MFMA flops).
//   BWD = false: K[m][r] = k(r2) for m < M, r < Rin, zero in the padding.
//   BWD = true : e = b - g a, kbar = e - g a, GW = kbar dk/dr2 (0 in the padding), E = e, svar[block] = sum kbar k.
// ------------------------------------------------------------------------------------------------------
#define KT 64
#define KD 32
struct KufArgs {
  const double* Zs;     // (Mp x D_in) Z / lengthscale
  const double* X;      // (Rin x D_in)
  const double* hyp;
  int64_t Rin, ld;
  int32_t M, Mp, D_in, D_out;
  double* K;            // forward out (Mp x ld)
  const double* A;      // backward in: a (Mp x ld), b = Ku^-1 abar (Mp x ld)
  const double* Bm;
  const double* VB;     // (D_out x ld)
  double* E;            // or NULL
  double* GW;
  double* svar;         // [blocks]
  int32_t white;        // backward: e = kbar = Bm (no -g a terms: they are part of a1bar)
};
// TS: tile edge, 64 or 32 (small launches — the 512 x 512 tile of a 784-pixel first layer is 64 tiles of 64 x 64 on 256 CUs, each a
// chain of 784 staged dimensions — take 32 x 32 tiles: four times the workgroups, a quarter of the chain each)
template <int KIND, bool BWD, int TS>
__global__ __launch_bounds__(256) void k_kuf(const KufArgs a) {
  constexpr int NI = TS / 16;
  // dimensions staged per round: a round costs a global round trip whatever its size (one workgroup per CU on the small launches),
  // so the 32 x 32 form — the 784-pixel layer's — takes 128 at a time (7 rounds instead of 25)
  constexpr int KDT = TS == 32 ? 4 * KD : KD;
  constexpr int NV = KDT / 32;
  __shared__ double zt[KDT][TS + 1];
  __shared__ double xt[KDT][TS + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * TS;
  const int m0 = blockIdx.y * TS;
  const double* ils = a.hyp + HYP_ILS;
  const double s2 = a.hyp[HYP_VAR];
  double r2[NI][NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) r2[i][j] = 0.0;
  // staging: thread = (dimension tid & 31 of the chunk, rows tid >> 5 + 8 u); the loads of chunk c + 1 are issued before the distance
  // updates of chunk c (the 784-pixel layer walks 25 chunks: with the load -> LDS -> barrier -> compute rounds one after the other the
  // 512 x 512 tile took 102 us, most of it exposed load latency)
  constexpr int NS = TS / 8;
  static_assert(KD == 32, "staging map");
  const int sd = tid & 31, sr = tid >> 5;
  double zr[NV][NS], xr[NV][NS], ilr[NV];
  // (gload only ISSUES loads — clamped addresses, no use of the values: a `* il` or a zero-select next to a load makes the compiler wait
  // for each pair in turn, sixteen round trips per chunk; scale and padding are applied when the chunk goes to LDS)
  auto gload = [&](int d0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int dd = min(d0 + 32 * v + sd, a.D_in - 1);
      ilr[v] = ils[dd];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int rr = sr + 8 * u;
        const int mrow = min(m0 + rr, a.Mp - 1);
        const int64_t xrow = min<int64_t>(r0 + rr, a.Rin - 1);
        zr[v][u] = a.Zs[(int64_t)mrow * a.D_in + dd];
        xr[v][u] = a.X[xrow * a.D_in + dd];
      }
    }
  };
  gload(0);
  for (int d0 = 0; d0 < a.D_in; d0 += KDT) {
    const int dn = min(KDT, a.D_in - d0);
    if (d0 > 0) __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const bool ok = d0 + 32 * v + sd < a.D_in;
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        zt[32 * v + sd][sr + 8 * u] = ok ? zr[v][u] : 0.0;
        xt[32 * v + sd][sr + 8 * u] = ok ? xr[v][u] * ilr[v] : 0.0;
      }
    }
    __syncthreads();
    if (d0 + KDT < a.D_in) gload(d0 + KDT);
    // UD dimensions per trip, their LDS reads issued together (one workgroup per CU on the small launches: a read -> use round trip
    // per dimension was the other half of those 102 us); the chunk's tail is zero on both sides
    constexpr int UD = NI == 2 ? 8 : 1;        // (64 x 64 tiles: 16 accumulators + 4 x 8 operands measured slower than the plain loop)
    for (int db = 0; db < dn; db += UD) {
      double zv[UD][NI], xv[UD][NI];
#pragma unroll
      for (int u = 0; u < UD; ++u) {
#pragma unroll
        for (int i = 0; i < NI; ++i) zv[u][i] = zt[db + u][ty + 16 * i];
#pragma unroll
        for (int j = 0; j < NI; ++j) xv[u][j] = xt[db + u][tx + 16 * j];
      }
#pragma unroll
      for (int u = 0; u < UD; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const double df = zv[u][i] - xv[u][j];
            r2[i][j] = fma(df, df, r2[i][j]);
          }
    }
  }
  double sv = 0.0;
  double gs[NI];
  double avs[BWD ? NI : 1][BWD ? NI : 1], bvs[BWD ? NI : 1][BWD ? NI : 1];
  if constexpr (BWD) {
    // every load of the tile first (clamped addresses): the stores below may alias them as far as the compiler knows, so a load next
    // to its use waited behind the previous element's stores — NI^2 round trips per thread
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int64_t off = (int64_t)min(m0 + ty + 16 * i, a.Mp - 1) * a.ld + min<int64_t>(r0 + tx + 16 * j, a.ld - 1);
        avs[i][j] = a.A[off];
        bvs[i][j] = a.Bm[off];
      }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t r = min<int64_t>(r0 + tx + 16 * j, a.ld - 1);
      double s = 0.0;
#pragma unroll 8
      for (int d = 0; d < a.D_out; ++d) s += a.VB[(int64_t)d * a.ld + r];
      gs[j] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= a.Mp) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t r = r0 + tx + 16 * j;
      if (r >= a.ld) continue;
      const bool ok = (m < a.M) && (r < a.Rin);
      if constexpr (!BWD) {
        a.K[(int64_t)m * a.ld + r] = ok ? kern_val<KIND>(r2[i][j], s2) : 0.0;
      } else {
        const double av = avs[i][j];
        const double e = a.white ? bvs[i][j] : bvs[i][j] - gs[j] * av;
        const double kbar = a.white ? e : e - gs[j] * av;
        double k, dk;
        kern_val_grad<KIND>(r2[i][j], s2, k, dk);
        sv += ok ? kbar * k : 0.0;
        a.GW[(int64_t)m * a.ld + r] = ok ? kbar * dk : 0.0;
        if (a.E) a.E[(int64_t)m * a.ld + r] = e;
      }
    }
  }
  if constexpr (BWD) {
    sv = sum_wave(sv);
    if ((tid & 63) == 0) red[tid >> 6] = sv;
    __syncthreads();
    if (tid == 0) a.svar[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// out[o] = sum of `n` buffers `stride` doubles apart (fixed order), two doubles per thread; grid.y = outputs (ostride apart in `part`,
// ostride_out apart in `out`)
__global__ void k_gl_sum(const double* __restrict__ part, int n, int64_t stride, int64_t count2, double* __restrict__ out, int64_t ostride,
                         int64_t ostride_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count2) return;
  const double* p = part + (int64_t)blockIdx.y * ostride;
  d2 s = reinterpret_cast<const d2*>(p)[i];
  int u = 1;
  for (; u + 3 < n; u += 4) {          // four loads in flight, added in index order
    d2 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = reinterpret_cast<const d2*>(p + (int64_t)(u + t) * stride)[i];
#pragma unroll
    for (int t = 0; t < 4; ++t) s += v[t];
  }
  for (; u < n; ++u) s += reinterpret_cast<const d2*>(p + (int64_t)u * stride)[i];
  reinterpret_cast<d2*>(out + (int64_t)blockIdx.y * ostride_out)[i] = s;
}
// column sums of squares of C (m x n) per 128-row tile (what k_pgemm's epilogue leaves when a product is not split): grid (64-column
// blocks, tiles, outputs), thread = (row group 0..3, column), fixed-order tree over the four row groups
__global__ __launch_bounds__(256) void k_gl_colsq(const double* __restrict__ C, int64_t sC, int64_t ldc, int m, int n, double* __restrict__ colsq,
                                                  int64_t ldq) {
  __shared__ double red[4][64];
  const int tid = threadIdx.x, cc = tid & 63, rr = tid >> 6;
  const int col = blockIdx.x * 64 + cc, tm = blockIdx.y, o = blockIdx.z;
  double sq = 0.0;
  if (col < n) {
    const double* p = C + (int64_t)o * sC + col;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int row = tm * PT + rr + 4 * i;
      const double v = row < m ? p[(int64_t)row * ldc] : 0.0;
      sq = fma(v, v, sq);
    }
  }
  red[rr][cc] = sq;
  __syncthreads();
  if (rr == 0 && col < n) colsq[((int64_t)o * gridDim.y + tm) * ldq + col] = (red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc]);
}
// (also the transposed, padded map of a Linear mean function: qmu = mean_A (D_in x D_out), Mp = D_in)
__global__ void k_gl_qmut(const double* __restrict__ qmu, int qld, int Mp, int D_out, int rows16, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows16 * Mp) return;
  const int d = idx / Mp, m = idx - d * Mp;
  out[idx] = d < D_out ? qmu[(int64_t)m * qld + d] : 0.0;
}
__global__ void k_gl_xt1(const double* __restrict__ X, int64_t Rin, int D_in, int64_t ld, double* __restrict__ XT1) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(D_in + 1) * ld) return;
  const int j = (int)(idx / ld);
  const int64_t r = idx - (int64_t)j * ld;
  XT1[idx] = (r < Rin) ? (j < D_in ? X[r * D_in + j] : 1.0) : 0.0;
}
// ZZ = [Z/l | (Z/l)^2 | 1]^T as (nzz16 x Mp) rows (the W operand of the ZZ^T GW product), rows >= M of Z/l are zero
__global__ void k_gl_zz(const double* __restrict__ Zs, int M, int Mp, int D_in, int nzz16, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nzz16 * Mp) return;
  const int j = idx / Mp, m = idx - j * Mp;
  double v = 0.0;
  if (m < M) {
    if (j < D_in) v = Zs[(int64_t)m * D_in + j];
    else if (j < 2 * D_in) { const double z = Zs[(int64_t)m * D_in + j - D_in]; v = z * z; }
    else if (j == 2 * D_in) v = 1.0;
  }
  out[idx] = v;
}

// forward epilogue.  A workgroup takes 64 data rows; the per-tile-row sums of squares and q_mu^T a arrive M-major (coalesced along
// the rows), go through LDS and leave row-major (a row's outputs are contiguous in mean / var / F), 32 outputs at a time.
// The first layer writes `rep` output rows per input row.
#define GL_EPI_ROWS 64
#define GL_EPI_DC 32
// lin_done: the Linear mean function's X mean_A is already part of MUT (thin product on [X^T ; 1]); only the bias is added here
// sum of the per-tile-row partials p[t * ld], t < n, in tile order; four loads in flight
__device__ __forceinline__ double tile_sum(const double* __restrict__ p, int64_t ld, int n) {
  double v = 0.0;
  int t = 0;
  for (; t + 3 < n; t += 4) {
    double w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = p[(int64_t)(t + u) * ld];
#pragma unroll
    for (int u = 0; u < 4; ++u) v += w[u];
  }
  for (; t < n; ++t) v += p[(int64_t)t * ld];
  return v;
}
template <bool LIK>
__global__ __launch_bounds__(256) void k_gl_epilogue(const LayerFwdArgs a, const double* __restrict__ colsq, int tiles_m, const double* __restrict__ MUT,
                                                     int lin_done) {
  __shared__ double red[8];
  __shared__ double s1s[GL_EPI_ROWS];
  __shared__ double s2s[GL_EPI_DC][GL_EPI_ROWS + 1];
  __shared__ double mus[GL_EPI_DC][GL_EPI_ROWS + 1];
  const int Dout = a.D_out, Din = a.D_in, tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * GL_EPI_ROWS;
  const double kdiag = a.hyp[HYP_KDIAG];
  const double lik_s2 = LIK ? a.lik_const[0] : 1.0;
  const double lik_c0 = -0.91893853320467274178 - 0.5 * log(lik_s2);
  double lik_ve = 0.0, lik_dl = 0.0;
  if (tid < GL_EPI_ROWS) {
    const int64_t r = r0 + tid;
    s1s[tid] = r < a.ldA ? tile_sum(colsq + r, a.ldA, tiles_m) : 0.0;
  }
  for (int d0 = 0; d0 < Dout; d0 += GL_EPI_DC) {
    const int dn = min(GL_EPI_DC, Dout - d0);
    __syncthreads();
    for (int idx = tid; idx < dn * GL_EPI_ROWS; idx += 256) {
      const int dd = idx / GL_EPI_ROWS, rr = idx - dd * GL_EPI_ROWS;
      const int64_t r = r0 + rr;
      double v = 0.0, mu = 0.0;
      if (r < a.ldA) {
        mu = MUT[(int64_t)(d0 + dd) * a.ldA + r];                           // layers.py:190
        v = tile_sum(colsq + (int64_t)(1 + d0 + dd) * tiles_m * a.ldA + r, a.ldA, tiles_m);
      }
      s2s[dd][rr] = v;
      mus[dd][rr] = mu;
    }
    __syncthreads();
    // the first layer writes `rep` = S output rows per input row: the samples are cut over gridDim.y workgroups (one workgroup per 64
    // input rows walking all S — a z load and three stores each — was 50 us of a two-workgroup launch at config 5)
    const int s_lo = (int)((int64_t)blockIdx.y * a.rep / gridDim.y), s_hi = (int)((int64_t)(blockIdx.y + 1) * a.rep / gridDim.y);
    for (int e = tid; e < dn * GL_EPI_ROWS; e += 256) {
      const int rr = e / dn, dd = e - rr * dn, d = d0 + dd;
      const int64_t r = r0 + rr;
      if (r >= a.Rin) {
        if (LIK && r < a.lik_ld) {          // rows of the 16-row padding of the transposed adjoints
          a.lik_MB[(int64_t)d * a.lik_ld + r] = 0.0;
          a.lik_VB[(int64_t)d * a.lik_ld + r] = 0.0;
        }
        continue;
      }
      const double var = kdiag - s1s[rr] + s2s[dd][rr];                     // layers.py:212-217
      double mu = mus[dd][rr];
      if (a.mean_kind == DSDGP_MEAN_IDENTITY) {                            // layers.py:219
        mu += a.X[r * Din + d];
      } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
        double m2 = 0.0;
        if (!lin_done)
          for (int j = 0; j < Din; ++j) m2 = fma(a.X[r * Din + j], a.mean_A[(int64_t)j * Dout + d], m2);
        mu += m2 + (a.mean_b ? a.mean_b[d] : 0.0);
      }
      const double sd = sqrt(var + a.jitter);
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
  if constexpr (LIK) {
    lik_ve = sum_wave(lik_ve);
    lik_dl = sum_wave(lik_dl);
    if ((tid & 63) == 0) {
      red[2 * (tid >> 6)] = lik_ve;
      red[2 * (tid >> 6) + 1] = lik_dl;
    }
    __syncthreads();
    if (tid == 0) {
      a.lik_part[2 * (int64_t)blockIdx.x] = (red[0] + red[2]) + (red[4] + red[6]);
      a.lik_part[2 * (int64_t)blockIdx.x + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
  }
}
static int pgemm_split(dsdgp_ctx* ctx, PGemm P, double* pb, int64_t pb_doubles) {
  const int tiles_m = ceil_div(P.m, PT);
  const int Z0 = P.batch;
  const int64_t tiles = (int64_t)tiles_m * ceil_div(P.n, 64) * Z0;
  const int ksteps = ceil_div(P.k, PK);
  const int64_t one = (int64_t)P.m * P.ldc;
  // up to 127 tiles: towards one workgroup per CU; 128 - 255 tiles of a long product (M = 1024: the 8 outputs of config 5's 125-row first
  // layer, 128 workgroups of 36 - 64 steps, 113 us): towards two per CU
  const bool mid = tiles >= 128 && tiles < 256 && ksteps >= 64;
  int G = (int)std::min<int64_t>(8, std::min<int64_t>(ksteps / 8, ((mid ? 512 : 256) + tiles - 1) / tiles));
  if (G > 1 && (int64_t)G * Z0 * one > pb_doubles) G = (int)(pb_doubles / (Z0 * one));
  if ((tiles >= 128 && !mid) || P.groups > 1 || P.reduce_batch || G < 2 || (one & 1)) return pgemm_launch(ctx, P);
  PGemm Q = P;
  Q.C = pb; Q.sC = (int64_t)G * one; Q.sCg = one; Q.groups = G; Q.store = 1; Q.colsq = nullptr;
  DS_TRY(pgemm_launch(ctx, Q));
  // the sum goes to the product's own output, or (not stored: only the norms are wanted) over the first piece of each output
  double* sum = P.store ? P.C : pb;
  const int64_t sum_stride = P.store ? P.sC : (int64_t)G * one;
  DS_LAUNCH(k_gl_sum, dim3(ceil_div(one / 2, 256), Z0), dim3(256), 0, ctx->stream, pb, G, one, one / 2, sum, (int64_t)G * one, sum_stride);
  if (P.colsq)
    DS_LAUNCH(k_gl_colsq, dim3(ceil_div(P.n, 64), tiles_m, Z0), dim3(256), 0, ctx->stream, sum, sum_stride, P.ldc, P.m, P.n, P.colsq, P.ldq);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
int layer_gemm_lik_blocks(int64_t Rin, int D_out) { return ceil_div(round_up(Rin, 16), GL_EPI_ROWS); }

// backward, per data row: dX (or the transposed adjoints of the layer below) and this block's hyper-parameter partial sums from
// OUT = ZZ^T GW:  WZ[j][r] = sum_m w z_mj,  Z2[j][r] = sum_m w z_mj^2,  W1[r] = sum_m w  (w = GW[m][r], z = Z / l)
//   d X partial      sum_m w (x - z)   = x W1 - WZ
//   lengthscale part sum_m w (x - z)^2 = x^2 W1 - 2 x WZ + Z2          (x = X / l)
// hyp_part row of row block i: [0, sum_r g_r, -2 (1/l_j) sum_r (...)_j]; rows nb .. nb + nsv - 1 carry [svar / s2, 0, ...] of the
// element-wise kernel's blocks (written by the workgroups with blockIdx.x >= nb).
// grid (row blocks + svar blocks, chunks of GL_JC input dimensions): a chunk's workgroup owns columns 2 + j of its row block's hyp_part
// row for its own j (chunk 0 also columns 0, 1), so that the 784 dimensions of a wide first layer spread over 49 workgroups per row block
#define GL_BR 256
#define GL_JC 16
__global__ __launch_bounds__(GL_BR) void k_gl_bwd_rows(const LayerBwdArgs a, const double* __restrict__ OUT, const double* __restrict__ svar,
                                                       int nsv, int nb) {
  __shared__ double red[2][GL_BR / 64];
  const int Din = a.D_in, Dout = a.D_out;
  const int tid = threadIdx.x;
  const double* ils = a.hyp + HYP_ILS;
  const double s2 = a.hyp[HYP_VAR];
  if ((int)blockIdx.x >= nb) {       // the svar rows: [svar / s2, 0, ...], stored along the rows (a thread per row walked Din + 2 scattered
                                     // stores: 100 us of one workgroup on the 784-pixel layer), the chunks of grid.y sharing them
    const int64_t p0 = ((int64_t)blockIdx.x - nb) * GL_BR;
    const int wd = Din + 2;
    const int64_t cnt = (int64_t)min<int64_t>(GL_BR, nsv - p0) * wd;
    for (int64_t idx = (int64_t)blockIdx.y * GL_BR + tid; idx < cnt; idx += (int64_t)gridDim.y * GL_BR) {
      const int64_t p = p0 + idx / wd;
      const int col = (int)(idx % wd);
      a.hyp_part[(int64_t)nb * wd + p0 * wd + idx] = col == 0 ? svar[p] / s2 : 0.0;
    }
    return;
  }
  const int64_t r = (int64_t)blockIdx.x * GL_BR + tid;
  const bool rv = r < a.Rin, rl = r < a.ldA;
  double* hp = a.hyp_part + (int64_t)blockIdx.x * (Din + 2);
  int par = 0;
  auto block_sum = [&](double v) -> double {      // (alternating scratch: one barrier per sum)
    v = sum_wave(v);
    if ((tid & 63) == 0) red[par][tid >> 6] = v;
    __syncthreads();
    const double t = (red[par][0] + red[par][1]) + (red[par][2] + red[par][3]);
    par ^= 1;
    return t;
  };
  if (blockIdx.y == 0) {
    double g = 0.0;
    if (rv)
      for (int d = 0; d < Dout; ++d) g += a.VB[(int64_t)d * a.ldA + r];
    const double gk = block_sum(g);
    if (tid == 0) {
      hp[0] = 0.0;
      hp[1] = gk;
    }
  }
  const double w1 = rl ? OUT[(int64_t)(2 * Din) * a.ldA + r] : 0.0;
  const int j_lo = blockIdx.y * GL_JC;
  // every load of the chunk first (the rows of X are Din doubles apart: a load -> sum -> barrier sequence per dimension walked GL_JC
  // dependent round trips, 102 us on the 784-pixel layer), then the GL_JC block sums
  double xs[GL_JC], wzs[GL_JC], z2s[GL_JC];
#pragma unroll
  for (int t = 0; t < GL_JC; ++t) {
    const int j = j_lo + t;
    const bool ok = rl && j < Din;
    xs[t] = (ok && rv) ? a.X[r * Din + j] : 0.0;
    wzs[t] = ok ? OUT[(int64_t)j * a.ldA + r] : 0.0;
    z2s[t] = ok ? OUT[(int64_t)(Din + j) * a.ldA + r] : 0.0;
  }
  double sls[GL_JC];
#pragma unroll
  for (int t = 0; t < GL_JC; ++t) {
    const int j = j_lo + t;
    double sl = 0.0;
    if (rl && j < Din) {
      const double xv = xs[t] * ils[j];
      const double wz = wzs[t], z2 = z2s[t];
      sl = rv ? fma(xv * xv, w1, fma(-2.0 * xv, wz, z2)) : 0.0;
      if (a.dX || a.MBp) {
        if (rv) {
          double dx = 2.0 * ils[j] * fma(xv, w1, -wz);
          if (a.mean_kind == DSDGP_MEAN_IDENTITY) {
            dx += a.MB[(int64_t)j * a.ldA + r];
          } else if (a.mean_kind == DSDGP_MEAN_LINEAR) {
            for (int d = 0; d < Dout; ++d) dx = fma(a.mean_A[(int64_t)j * Dout + d], a.MB[(int64_t)d * a.ldA + r], dx);
          }
          if (a.MBp) {
            const int d = j - a.prop;
            if (d >= 0) {
              const double zv = a.zp[(r / a.n_inner) * a.zp_s + (r % a.n_inner) * a.zp_n + d * a.zp_d];
              const double vv = a.varp[r * a.Dp + d];
              a.MBp[(int64_t)d * a.ldA + r] = dx;
              a.VBp[(int64_t)d * a.ldA + r] = dx * zv * 0.5 * rsqrt(vv + a.jitter);
            }
          } else {
            a.dX[r * Din + j] = dx;
          }
        } else if (a.MBp && j >= a.prop) {
          a.MBp[(int64_t)(j - a.prop) * a.ldA + r] = 0.0;
          a.VBp[(int64_t)(j - a.prop) * a.ldA + r] = 0.0;
        }
      }
    }
    sls[t] = sl;
  }
#pragma unroll
  for (int t = 0; t < GL_JC; ++t) {
    const int j = j_lo + t;
    if (j >= Din) break;            // (uniform over the workgroup)
    const double tot = block_sum(sls[t]);
    if (tid == 0) hp[2 + j] = -2.0 * ils[j] * tot;
  }
}
// hyp_part rows the GEMM-formulated backward writes for ld padded rows: one per block of k_gl_bwd_rows + one per block of k_kuf
static inline int kuf_tile(int64_t ld, int Mp);
int layer_gemm_hyp_parts(int64_t ld, int Mp) {
  const int ts = kuf_tile(ld, Mp);
  return ceil_div(ld, GL_BR) + ceil_div(ld, ts) * ceil_div(Mp, ts);
}

// tile edge of the Kuf kernels for a launch of these extents (the same rule sizes the svar partials of the backward form)
static inline int kuf_tile(int64_t ld, int Mp) { return ((int64_t)ceil_div(ld, KT) * ceil_div(Mp, KT) >= 256) ? KT : 32; }
template <bool BWD>
static int kuf_launch(dsdgp_ctx* ctx, int kern_kind, const KufArgs& k) {
  const int ts = kuf_tile(k.ld, k.Mp);
  const dim3 grid(ceil_div(k.ld, ts), ceil_div(k.Mp, ts));
  if (kern_kind == DSDGP_KERN_RBF) {
    if (ts == KT) DS_LAUNCH((k_kuf<DSDGP_KERN_RBF, BWD, KT>), grid, dim3(256), 0, ctx->stream, k);
    else DS_LAUNCH((k_kuf<DSDGP_KERN_RBF, BWD, 32>), grid, dim3(256), 0, ctx->stream, k);
  } else {
    if (ts == KT) DS_LAUNCH((k_kuf<DSDGP_KERN_MATERN52, BWD, KT>), grid, dim3(256), 0, ctx->stream, k);
    else DS_LAUNCH((k_kuf<DSDGP_KERN_MATERN52, BWD, 32>), grid, dim3(256), 0, ctx->stream, k);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

int layer_fwd_gemm_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white, const GemmLayerWs& ws) {
  ProfScope ps(ctx, "layer_fwd");
  const int64_t ld = a.ldA;
  const int Dout = a.D_out, tiles_m = ceil_div(Mp, PT);
  const int64_t MM = (int64_t)Mp * Mp, ML = (int64_t)Mp * ld;
  hipStream_t st = ctx->stream;
  // K = k(Z, X)
  KufArgs k{};
  k.Zs = a.Zs; k.X = a.X; k.hyp = a.hyp; k.Rin = a.Rin; k.ld = ld; k.M = a.M; k.Mp = Mp; k.D_in = a.D_in; k.D_out = Dout; k.K = ws.T1;
  DS_TRY(kuf_launch<false>(ctx, kern_kind, k));
  // [X^T ; 1]: the Z-gradient product of the reverse pass reads it, and so does the Linear mean function's product below
  const int rows16 = (int)round_up(Dout, 16);
  const bool lin_thin = a.mean_kind == DSDGP_MEAN_LINEAR && rows16 <= 32;
  double* XT = a.XT1 ? a.XT1 : (lin_thin ? ws.OUTt : nullptr);
  if (XT) {
    const int64_t cnt = (int64_t)(a.D_in + 1) * ld;
    DS_LAUNCH(k_gl_xt1, dim3(ceil_div(cnt, 256)), dim3(256), 0, st, a.X, a.Rin, a.D_in, ld, XT);
  }
  DS_LAUNCH(k_gl_qmut, dim3(ceil_div((int64_t)rows16 * Mp, 256)), dim3(256), 0, st, a.qmu, a.qmu_ld ? a.qmu_ld : Dout, Mp, Dout, rows16,
                     ws.qmuT);
  DS_HIP(hipGetLastError());
  // a1 = Lu^-1 K (layers.py:186), |a1|^2 per tile row.  a = Lu^-T a1 (layers.py:188) goes straight to Asave when the pass keeps it.
  // white = True (layers.py:186-188 without the second solve): a1 takes a's place in everything below
  double* Aout = a.Asave ? a.Asave : (white ? ws.T2 : ws.T1);      // (K is dead once a1 exists)
  PGemm P{};
  P.W = a.Linv; P.B = ws.T1; P.C = white ? Aout : ws.T2; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = Mp; P.n = (int)ld; P.k = Mp; P.batch = 1;
  P.tri = 2; P.store = 1; P.alpha = 1.0; P.colsq = ws.colsq; P.ldq = ld;
  DS_TRY(pgemm_split(ctx, P, ws.Pb, ws.pb_doubles));
  if (!white) {
    P = PGemm{};
    P.W = a.LinvT; P.B = ws.T2; P.C = Aout; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = Mp; P.n = (int)ld; P.k = Mp; P.batch = 1;
    P.tri = 8; P.store = 1; P.alpha = 1.0;
    DS_TRY(pgemm_split(ctx, P, ws.Pb, ws.pb_doubles));
  }
  // c_d = q_sqrt_d^T a for every output in one launch: |c_d|^2 per tile row; c_d itself only when the reverse pass wants it
  P = PGemm{};
  P.W = a.TpT; P.sW = MM; P.B = Aout; P.sB = 0; P.C = a.Csave; P.sC = ML; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = Mp; P.n = (int)ld; P.k = Mp;
  P.batch = Dout; P.tri = 8; P.store = a.Csave ? 1 : 0; P.alpha = 1.0; P.colsq = ws.colsq + (int64_t)tiles_m * ld; P.ldq = ld;
  DS_TRY(pgemm_split(ctx, P, ws.Pb, ws.pb_doubles));
  // q_mu^T a (layers.py:190)
  if (rows16 <= 32) {
    DS_TRY(thin_launch(ctx, ws.qmuT, Mp, Aout, ld, ws.MUT, ld, rows16, (int)ld, Mp));
  } else {
    P = PGemm{};
    P.W = ws.qmuT; P.B = Aout; P.C = ws.MUT; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = rows16; P.n = (int)ld; P.k = Mp; P.batch = 1;
    P.tri = 0; P.store = 1; P.alpha = 1.0;
    DS_TRY(pgemm_launch(ctx, P));
  }
  if (lin_thin) {
    // Linear mean function (layers.py:219; the PCA step-down of a 784-pixel first layer): MUT += mean_A^T X^T as a thin product — the
    // per-(row, output) dot product over D_in strided reads was 0.97 ms per step at config 4
    DS_LAUNCH(k_gl_qmut, dim3(ceil_div((int64_t)rows16 * a.D_in, 256)), dim3(256), 0, st, a.mean_A, Dout, a.D_in, Dout, rows16, ws.ZZ);
    DS_HIP(hipGetLastError());
    DS_TRY(thin_launch(ctx, ws.ZZ, a.D_in, XT, ld, ws.MUT, ld, rows16, (int)ld, a.D_in, 1));
  }
  const int nb = ceil_div(ld, GL_EPI_ROWS);
  if (a.lik_Y) {
    DS_LAUNCH(k_gl_epilogue<true>, dim3(nb), dim3(256), 0, st, a, ws.colsq, tiles_m, ws.MUT, lin_thin ? 1 : 0);
  } else {
    const int ny = std::max(1, std::min(a.rep, 512 / nb));      // sample chunks of a first layer: ~512 workgroups
    DS_LAUNCH(k_gl_epilogue<false>, dim3(nb, ny), dim3(256), 0, st, a, ws.colsq, tiles_m, ws.MUT, lin_thin ? 1 : 0);
  }
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// white = True: abar (Mp x ld, in place) -= 2 (sum_d vbar_d[r]) a1[m][r]; grid (256-column blocks, row groups)
__global__ __launch_bounds__(256) void k_gl_white_abar(double* __restrict__ T, const double* __restrict__ A1, const double* __restrict__ VB, int Mp,
                                                       int64_t ld, int Dout) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= ld) return;
  double g = 0.0;
#pragma unroll 4
  for (int d = 0; d < Dout; ++d) g += VB[(int64_t)d * ld + r];
  g *= 2.0;
  for (int m = blockIdx.y; m < Mp; m += gridDim.y) T[(int64_t)m * ld + r] -= g * A1[(int64_t)m * ld + r];
}

int layer_bwd_gemm_launch(dsdgp_ctx* ctx, const LayerBwdArgs& b, int Mp, int kern_kind, int white, const GemmLayerWs& ws) {
  ProfScope ps(ctx, "layer_bwd");
  const int64_t ld = b.ldA;
  const int Dout = b.D_out, Din = b.D_in;
  const int64_t MM = (int64_t)Mp * Mp, ML = (int64_t)Mp * ld;
  hipStream_t st = ctx->stream;
  // abar = sum_d (...) + q_mu mbar (the tail product): the (output, k) steps of a tile cut into `groups` contiguous pieces when one
  // workgroup per tile would not fill the chip with 128-wide tiles (partial sums in Pb, added in a fixed order)
  const int tiles128 = ceil_div(Mp, PT) * ceil_div(ld, PT);
  int groups = (512 + tiles128 - 1) / tiles128;           // (128-wide tiles run at 50 TFLOP/s when they fill the chip, 64-wide ones at 31 - 38)
  // (more pieces fit when the launch is small: the 512-row first layer of config 4 — 16 tiles — ran 278 us as 32 x 8 workgroups of
  // 64-wide tiles, one per CU; as 16 x 32 of 128-wide ones it fills the chip)
  const int gmax = (int)std::min<int64_t>(32, std::max<int64_t>(GL_MAX_GROUPS, ws.pb_doubles / ML));
  if (groups > gmax) groups = gmax;
  if (groups < 1) groups = 1;
  PGemm P{};
  if (b.Csave) {      // abar += q_sqrt_d (2 vbar_d c_d): triangular
    P.W = b.Tp; P.B = b.Csave; P.sB = ML; P.tri = 2;
  } else {            // abar += S_d (2 vbar_d a): dense
    P.W = b.Sd; P.B = b.Asave; P.sB = 0; P.tri = 0;
  }
  P.sW = MM; P.C = groups > 1 ? ws.Pb : ws.T2; P.sCg = ML; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = Mp; P.n = (int)ld; P.k = Mp; P.batch = Dout;
  P.reduce_batch = 1; P.groups = groups; P.store = 1; P.alpha = 1.0; P.bscale = b.VB; P.sS = ld; P.bs_mul = 2.0;   // columns scaled by 2 vbar_d
  P.W2 = b.qmu4; P.B2 = b.MB; P.k2 = b.DP4; P.ldw2 = b.DP4;
  DS_TRY(pgemm_launch(ctx, P));
  if (groups > 1) {
    DS_LAUNCH(k_gl_sum, dim3(ceil_div(ML / 2, 256)), dim3(256), 0, st, ws.Pb, groups, ML, ML / 2, ws.T2, (int64_t)0, (int64_t)0);
    DS_HIP(hipGetLastError());
  }
  P = PGemm{};        // b = Ku^-1 abar   |   white: kbar = Lu^-T a1bar,  a1bar = abar - 2 (sum_d vbar_d) a1 (the -|a1|^2 term of the variance)
  if (white) {
    DS_LAUNCH(k_gl_white_abar, dim3(ceil_div(ld, 256), std::min(Mp / 16, 64)), dim3(256), 0, st, ws.T2, b.Asave, b.VB, Mp, ld, Dout);
    DS_HIP(hipGetLastError());
    P.W = b.LinvT; P.tri = 8;
  } else {
    P.W = b.Kinv;
  }
  P.B = ws.T2; P.C = ws.T1; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = Mp; P.n = (int)ld; P.k = Mp;
  P.batch = 1; P.store = 1; P.alpha = 1.0;
  DS_TRY(pgemm_split(ctx, P, ws.Pb, ws.pb_doubles));
  // e, kbar, GW (+ E) with the kernel recomputed
  KufArgs k{};
  k.Zs = b.Zs; k.X = b.X; k.hyp = b.hyp; k.Rin = b.Rin; k.ld = ld; k.M = b.M; k.Mp = Mp; k.D_in = Din; k.D_out = Dout;
  k.A = b.Asave; k.Bm = ws.T1; k.VB = b.VB; k.E = b.E; k.GW = b.GW; k.svar = ws.svar; k.white = white;
  DS_TRY(kuf_launch<true>(ctx, kern_kind, k));
  const int kts = kuf_tile(ld, Mp);
  const int nsv = ceil_div(ld, kts) * ceil_div(Mp, kts);
  // sums over the inducing rows: OUT = ZZ^T GW
  const int nzz = 2 * Din + 1, nzz16 = (int)round_up(nzz, 16);
  DS_LAUNCH(k_gl_zz, dim3(ceil_div((int64_t)nzz16 * Mp, 256)), dim3(256), 0, st, b.Zs, b.M, Mp, Din, nzz16, ws.ZZ);
  DS_HIP(hipGetLastError());
  if (nzz16 <= 32) {
    DS_TRY(thin_launch(ctx, ws.ZZ, Mp, b.GW, ld, ws.OUTt, ld, nzz16, (int)ld, Mp));
  } else {
    P = PGemm{};
    P.W = ws.ZZ; P.B = b.GW; P.C = ws.OUTt; P.ldw = Mp; P.ldb = ld; P.ldc = ld; P.m = nzz16; P.n = (int)ld; P.k = Mp; P.batch = 1; P.store = 1; P.alpha = 1.0;
    DS_TRY(pgemm_launch(ctx, P));
  }
  const int nb = ceil_div(ld, GL_BR);
  DS_LAUNCH(k_gl_bwd_rows, dim3(nb + ceil_div(nsv, GL_BR), ceil_div(Din, GL_JC)), dim3(GL_BR), 0, st, b, ws.OUTt, ws.svar, nsv, nb);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}
