// K4 — tf.matrix_triangular_solve(L, B, lower=True) (layers.py:186,188) and its batched use (layers.py:239:
// tf.matrix_triangular_solve(self.Lu_tiled, self.q_sqrt)) as a BLOCKED solve on the fp64 MFMA pipe (gfx950):
//     trans = 0:  B <- L^-1 B          trans = 1:  B <- L^-T B          L: n x n lower, B: n x nrhs, row-major, `batch` of them
//
//  * the 16 x 16 diagonal blocks of L are inverted once (k_trsm_diag: one lane per column, forward substitution) — n^2 / 16 * 16
//    flops, nothing like the n^3 / 3 of a full inverse (rounds 1-2 inverted all of L with ONE workgroup and multiplied densely:
//    2x a solve's flops and a serial 25 ms at n = 1024);
//  * panels of <= 128 rows are solved by k_trsm_panel, ONE WAVE PER 16 RIGHT-HAND-SIDE COLUMNS: the wave walks the panel's 16-row
//    blocks, X_i = D_i (B_i - sum_k L_ik X_k), with every X_k of its columns in registers — the D layout of one MFMA product is
//    the B-operand layout of the next, so nothing goes through LDS and no wave waits for another; thousands of waves in flight at
//    the right-hand-side counts of the layer (nrhs = S N = 20 000 at config 2);
//  * between panels the remaining rows take B_rest -= L_rest,panel X_panel as ONE grouped MFMA GEMM launch per panel (all matrices
//    of the batch in it; the LDS-free 64 x 64 kernel at K = 128) — for n > 128 that is where the n^2 nrhs flops are.
// Flop count = n^2 nrhs per matrix (the triangular count; bench.py `sub_rooflines.trsm` quotes it against the 78.6 TFLOP/s peak).
#include <vector>

#include "linalg.hpp"

#define TRSM_PANEL 128

// inverse of the 16 x 16 diagonal blocks: grid (nb, batch), 64 threads (16 active: lane = column)
__global__ __launch_bounds__(64) void k_trsm_diag(const double* __restrict__ L, int64_t ldl, int64_t strideL, int n, double* __restrict__ Dinv,
                                                 int64_t strideD) {
  __shared__ double Ld[16 * 17];
  const int jb = blockIdx.x, j0 = jb * 16, lane = threadIdx.x;
  const double* __restrict__ Lb = L + (int64_t)blockIdx.y * strideL;
  for (int idx = lane; idx < 256; idx += 64) {
    const int i = idx >> 4, j = idx & 15;
    double v = (i == j) ? 1.0 : 0.0;                                   // identity pad beyond n
    if (j0 + i < n && j0 + j < n) v = (j <= i) ? Lb[(int64_t)(j0 + i) * ldl + j0 + j] : 0.0;
    Ld[i * 17 + j] = v;
  }
  __syncthreads();
  if (lane >= 16) return;
  double x[16], sacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    x[k] = sacc[k] / Ld[k * 17 + k];
#pragma unroll
    for (int i = k + 1; i < 16; ++i) sacc[i] = fma(-Ld[i * 17 + k], x[k], sacc[i]);
  }
  double* __restrict__ D = Dinv + (int64_t)blockIdx.y * strideD + (int64_t)jb * 256;
#pragma unroll
  for (int i = 0; i < 16; ++i) D[i * 16 + lane] = x[i];
}

struct TrsmPanel {
  const double* L; int64_t ldl, strideL;
  const double* Dinv; int64_t strideD;
  double* B; int64_t ldb, strideB;
  int32_t n, r0, pn;          // rows r0 .. r0 + pn of the system (pn <= TRSM_PANEL, multiple of 16 except at the end)
  int64_t nrhs;
  int32_t trans;
};

// grid (ceil(nrhs / 64), batch), 256 threads: wave w of block bx owns columns 64 bx + 16 w .. + 15
__global__ __launch_bounds__(256) void k_trsm_panel(const TrsmPanel P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int64_t col0 = (int64_t)blockIdx.x * 64 + 16 * wave;
  if (col0 >= P.nrhs) return;
  gcptr L = (gcptr)(P.L + (int64_t)blockIdx.y * P.strideL);
  gcptr Dv = (gcptr)(P.Dinv + (int64_t)blockIdx.y * P.strideD);
  gptr B = (gptr)(P.B + (int64_t)blockIdx.y * P.strideB);
  const int pb = (P.pn + 15) / 16, b0 = P.r0 / 16;
  const int64_t col = col0 + c;
  const bool cok = col < P.nrhs;
  const int64_t colc = cok ? col : P.nrhs - 1;
  constexpr int MAXB = TRSM_PANEL / 16;
  d4 x[MAXB];
  // block order: forward (trans = 0) or backward (trans = 1)
#pragma unroll
  for (int step = 0; step < MAXB; ++step) {
    if (step < pb) {
      const int i = P.trans ? pb - 1 - step : step;                // block row inside the panel
      const int rI = P.r0 + 16 * i;
      d4 acc0, acc1 = (d4){0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = rI + g + 4 * t;
        acc0[t] = (r < P.n) ? B[(int64_t)r * P.ldb + colc] : 0.0;
      }
      // acc -= op(L)_ik X_k over the blocks already solved (k < i forward, k > i backward)
#pragma unroll
      for (int s2 = 0; s2 < MAXB; ++s2) {
        if (s2 < step) {
          const int k = P.trans ? pb - 1 - s2 : s2;
          const int rK = P.r0 + 16 * k;
          double av[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            // A operand [row c][k index 4 s + g] of op(L)_ik:  L[rI + c][rK + 4 s + g]  or  L[rK + 4 s + g][rI + c]
            const int ra = P.trans ? rK + 4 * s + g : rI + c, ca = P.trans ? rI + c : rK + 4 * s + g;
            av[s] = (ra < P.n && ca < P.n) ? L[(int64_t)ra * P.ldl + ca] : 0.0;
          }
          if (s2 & 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc1 = mfma_f64(-av[s], x[s2][s], acc1);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc0 = mfma_f64(-av[s], x[s2][s], acc0);
          }
        }
      }
      acc0 += acc1;
      // X_i = D_i acc  (trans: D_i^T acc);  A operand [row c][k 4 s + g] = D[c][4 s + g]  or  D[4 s + g][c]
      gcptr Di = Dv + (int64_t)(b0 + i) * 256;
      d4 xi = (d4){0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double dv = P.trans ? Di[(4 * s + g) * 16 + c] : Di[c * 16 + 4 * s + g];
        xi = mfma_f64(dv, acc0[s], xi);
      }
      x[step] = xi;
      if (cok) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = rI + g + 4 * t;
          if (r < P.n) B[(int64_t)r * P.ldb + col] = xi[t];
        }
      }
    }
  }
}

// batched entry: L matrices strideL apart (0: one L shared by the batch, as the tiled Lu of layers.py:173,239), B matrices strideB apart
// full transpose (n x n, row-major) through 32 x 32 LDS tiles
__global__ __launch_bounds__(256) void k_trsm_transpose(const double* __restrict__ L, int64_t ldl, int n, double* __restrict__ T) {
  __shared__ double tl[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) tl[r][tx] = (i0 + r < n && j0 + tx < n) ? L[(int64_t)(i0 + r) * ldl + j0 + tx] : 0.0;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (j0 + r < n && i0 + tx < n) T[(int64_t)(j0 + r) * n + i0 + tx] = tl[tx][r];
}

// One large system (n >= 256, thousands of right-hand sides): LEFT-looking.  Panel p first takes the contributions of all solved panels
// as ONE long product through the LDS-tiled 128-row kernel of the layer passes (pgemm_accum: k = the rows solved so far, each
// right-hand-side row read once per panel), then k_trsm_panel solves its 128 rows.  The right-looking form below updates ALL remaining
// rows after every panel with a K = 128 product: (n / 128)^2 / 2 read-modify-write passes over 128-row slabs of B (2.9 GB at
// n = 1024, nrhs = 50 000) on the LDS-free short-K kernel — 23 TFLOP/s there, 26 - 30 here (measured no better: 256-row
// super-panels, whose products take the kernel's 128-wide tiles but fill the chip 1.5 times).  trans = 1 walks the panels upwards on a transposed copy
// of L (the tiled kernel reads its left operand along rows).
static int trsm_left_looking(dsdgp_ctx* ctx, int trans, int n, int64_t nrhs, const double* L, int64_t ldl, double* B, int64_t ldb) {
  const int nb = ceil_div(n, 16), npanel = ceil_div(n, TRSM_PANEL);
  void* scr;
  const size_t dbytes = round_up((size_t)nb * 256 * sizeof(double), 256);
  DS_TRY(ctx_scratch(ctx, dbytes + (trans ? (size_t)n * n * sizeof(double) : 0), &scr));
  double* Dinv = (double*)scr;
  double* LT = (double*)((char*)scr + dbytes);
  DS_LAUNCH(k_trsm_diag, dim3(nb, 1), dim3(64), 0, ctx->stream, L, ldl, (int64_t)0, n, Dinv, (int64_t)nb * 256);
  if (trans) DS_LAUNCH(k_trsm_transpose, dim3(ceil_div(n, 32), ceil_div(n, 32)), dim3(256), 0, ctx->stream, L, ldl, n, LT);
  DS_HIP(hipGetLastError());
  for (int pi = 0; pi < npanel; ++pi) {
    const int p = trans ? npanel - 1 - pi : pi;
    const int r0 = p * TRSM_PANEL, pn = std::min(TRSM_PANEL, n - r0);
    if (!trans && r0 > 0)            // B_p -= L[p rows, : r0] X[: r0]
      DS_TRY(pgemm_accum(ctx, L + (int64_t)r0 * ldl, ldl, B, ldb, B + (int64_t)r0 * ldb, ldb, pn, (int)nrhs, r0, -1.0));
    if (trans && r0 + pn < n)        // B_p -= L[r0 + pn :, p columns]^T X[r0 + pn :]
      DS_TRY(pgemm_accum(ctx, LT + (int64_t)r0 * n + r0 + pn, n, B + (int64_t)(r0 + pn) * ldb, ldb, B + (int64_t)r0 * ldb, ldb, pn, (int)nrhs,
                         n - (r0 + pn), -1.0));
    TrsmPanel P{L, ldl, 0, Dinv, 0, B, ldb, 0, n, r0, pn, nrhs, trans ? 1 : 0};
    ProfScope ps(ctx, "trsm");
    DS_LAUNCH(k_trsm_panel, dim3(ceil_div(nrhs, 64), 1), dim3(256), 0, ctx->stream, P);
    DS_HIP(hipGetLastError());
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_trsm_batched(dsdgp_ctx* ctx, int trans, int n, int64_t nrhs, int batch, const double* L, int64_t ldl,
                                  int64_t strideL, double* B, int64_t ldb, int64_t strideB) {
  DS_CHECK_ARG(ctx && L && B && n > 0 && nrhs > 0 && batch > 0 && ldl >= n && ldb >= nrhs);
  DS_CHECK_ARG(nrhs <= 0x7fffffff && ldb <= 0x7fffffff && ldl <= 0x7fffffff);      // the update GEMMs carry 32-bit extents
  if (batch == 1 && n >= 256 && nrhs >= 2048 && (nrhs & 7) == 0 && (ldl & 1) == 0 && (ldb & 1) == 0 && (n & 1) == 0 &&
      (((uintptr_t)L | (uintptr_t)B) & 15) == 0)
    return trsm_left_looking(ctx, trans, n, nrhs, L, ldl, B, ldb);
  const int nb = ceil_div(n, 16);
  const int nL = strideL == 0 ? 1 : batch;
  const int npanel = ceil_div(n, TRSM_PANEL);
  void* scr;
  const size_t dbytes = round_up((size_t)nL * nb * 256 * sizeof(double), 256);
  DS_TRY(ctx_scratch(ctx, dbytes + (size_t)npanel * sizeof(GemmProblem) + 256, &scr));
  double* Dinv = (double*)scr;
  GemmProblem* gp = (GemmProblem*)((char*)scr + dbytes);
  const int64_t strideD = strideL == 0 ? 0 : (int64_t)nb * 256;
  DS_LAUNCH(k_trsm_diag, dim3(nb, nL), dim3(64), 0, ctx->stream, L, ldl, strideL, n, Dinv, (int64_t)nb * 256);
  DS_HIP(hipGetLastError());
  // the GEMM descriptors of all panels in one upload
  std::vector<GemmProblem> probs(npanel);
  std::vector<int> tiles(npanel, 0);
  for (int pi = 0; pi < npanel; ++pi) {
    const int p = trans ? npanel - 1 - pi : pi;
    const int r0 = p * TRSM_PANEL, pn = std::min(TRSM_PANEL, n - r0);
    GemmProblem& G = probs[pi];
    memset(&G, 0, sizeof(G));
    G.batch = batch; G.sA = strideL; G.sB = strideB; G.sC = strideB;
    G.lda = ldl; G.ldb = ldb; G.ldc = ldb;
    G.alpha = -1.0; G.beta = 1.0;
    G.n = (int)nrhs; G.k = pn;
    if (!trans) {                       // rows below the panel:  B[r0 + pn :, :] -= L[r0 + pn :, r0 : r0 + pn] X_panel
      G.m = n - (r0 + pn);
      G.A = L + (int64_t)(r0 + pn) * ldl + r0; G.transA = 0;
      G.B = B + (int64_t)r0 * ldb; G.C = B + (int64_t)(r0 + pn) * ldb;
    } else {                            // rows above the panel:  B[: r0, :] -= L[r0 : r0 + pn, : r0]^T X_panel
      G.m = r0;
      G.A = L + (int64_t)r0 * ldl; G.transA = 1;
      G.B = B + (int64_t)r0 * ldb; G.C = B;
    }
    if (G.m > 0) tiles[pi] = gemm_plan(&G, 1);
  }
  DS_TRY(ctx_upload(ctx, gp, probs.data(), probs.size() * sizeof(GemmProblem)));
  for (int pi = 0; pi < npanel; ++pi) {
    const int p = trans ? npanel - 1 - pi : pi;
    TrsmPanel P{L, ldl, strideL, Dinv, strideD, B, ldb, strideB, n, p * TRSM_PANEL, std::min(TRSM_PANEL, n - p * TRSM_PANEL), nrhs, trans ? 1 : 0};
    {
      ProfScope ps(ctx, "trsm");
      DS_LAUNCH(k_trsm_panel, dim3(ceil_div(nrhs, 64), batch), dim3(256), 0, ctx->stream, P);
      DS_HIP(hipGetLastError());
    }
    if (probs[pi].m > 0) DS_TRY(gemm_launch(ctx, gp + pi, 1, tiles[pi]));
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_trsm(dsdgp_ctx* ctx, int trans, int n, int64_t nrhs, const double* L, int64_t ldl, double* B, int64_t ldb) {
  return dsdgp_trsm_batched(ctx, trans, n, nrhs, 1, L, ldl, 0, B, ldb, 0);
}
