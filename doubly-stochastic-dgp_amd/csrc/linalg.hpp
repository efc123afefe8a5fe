// Small dense fp64 linear algebra on gfx950: grouped MFMA GEMM, blocked Cholesky + triangular inverse.
#pragma once
#include "common.hpp"

// One GEMM of a grouped launch: C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]   (row-major),
// or, with batch_reduce, C = alpha * sum_b op(A[b]) op(B[b]) + beta * C.
struct GemmProblem {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int32_t m, n, k;
  int32_t transA, transB;
  int32_t batch, batch_reduce;
  int32_t tiles_m, tiles_n, tile_start;
  double alpha, beta;
  int32_t lower_only;        // skip 64x64 tiles strictly above the diagonal (syrk-style updates, symmetric / triangular results)
  int32_t tri;               // structure hints that trim the k range of a tile (zeros are never multiplied):
                             //   1: B[k][n] lower-triangular (k >= n)      2: A[m][k] lower-triangular (k <= m)
                             //   4: transB with B[n][k] lower-triangular (k <= n)   8: A[m][k] upper-triangular (k >= m)
                             //  16: with lower_only — also store the transposed tile (symmetric result, full matrix wanted)
  // Longest-processing-time tile order of the launch this problem belongs to (gemm_plan_lpt; the same pointer in every problem of the
  // launch, NULL: workgroup b takes tile b): workgroup b takes tile order[2 b + 1] of problem order[2 b].  The structure hints give the
  // tiles of one launch k ranges from one to M / 16 steps; in index order the longest ones of the last problem start last and the launch
  // ends with a few workgroups running alone (the four 17 x 1024^3 launches of a config-5 step: ~800 us each against ~400 us of work).
  const int32_t* order;
  int32_t n_order, pad_order;
};

// One matrix of a batched factorisation launch (n multiple of 16; rows/cols >= nreal carry an identity pad).
struct PotrfItem {
  double* W;      // in: SPD matrix (lower part read); out: lower Cholesky factor, upper zeroed
  double* Linv;   // out: W^-1 (lower), may alias nothing else; may be NULL (skip inverse)
  double* LinvT;  // out: transpose of Linv (may be NULL)
  double* scal;   // out: [0] = sum_i<nreal 2 log L_ii (= logdet), [1] = info (0 ok, else 1-based bad pivot)
  int32_t n, ld, nreal, pad;   // pad bits: 8 = skip factor write-back, 16 = accumulate into scal (blocked driver), 7 = timing
  int32_t info_offset, pad2;   // added to a failing pivot index (position of this block inside a larger matrix)
};

// Fills tile_start/tiles_* of `host` problems, returns the total number of tiles (64 x 64, or 128 x 128 with the large-tile
// kernel flagged in bit 30 when the launch holds a problem of at least 512 x 512 and allow_big): pass it to gemm_launch as is.
int gemm_plan(GemmProblem* host, int nprob, int allow_big = 1);     // 0: 64 x 64 kernels only, 1: by size, 2: the 128 x 128 kernel
// gemm_plan + the launch's tiles that do any work (lower_only drops the ones above the diagonal), heaviest first: `order` gets
// (problem, tile) pairs; the caller puts them in device memory and points every problem's `order` / `n_order` at them.  The returned
// count (with the kernel flags) is the number of pairs = workgroups to launch.
int gemm_plan_lpt(GemmProblem* host, int nprob, std::vector<int32_t>& order, int allow_big = 1);
// Launch over problems already resident in device memory (`dev`), described by the planned `host` copy.
int gemm_launch(dsdgp_ctx* ctx, const GemmProblem* dev, int nprob, int total_tiles, hipStream_t stream = nullptr);
// layer_gemm.hip: C (m x n) += alpha W (m x k) B (k x n), row-major, 128 x 128 (or 128 x 64) LDS-tiled fp64-MFMA tiles over the whole k
// range (W, B 16-byte aligned, even leading dimensions, n a multiple of 8)
int pgemm_accum(dsdgp_ctx* ctx, const double* W, int64_t ldw, const double* B, int64_t ldb, double* C, int64_t ldc, int m, int n, int k, double alpha);
// n_max: largest (padded) matrix order among the items; <= 128 selects the LDS-resident variant
int potrf_launch(dsdgp_ctx* ctx, const PotrfItem* dev_items, int nitems, int n_max);
// batched inverse of padded lower-triangular matrices (n multiple of 16, identity pad), one workgroup each
int trtri_launch(dsdgp_ctx* ctx, double* W, double* Linv, int n, int64_t stride, int batch);

// ---- multi-workgroup blocked Cholesky + triangular inverse for large matrices (n >= 512, multiple of 64) ----
// Right-looking with NB = 64: diagonal blocks by k_potrf_trtri (LDS variant), panel solve and trailing syrk as grouped
// MFMA GEMM launches over all 64x64 tiles, inverse by the blocked recurrence X_i,: = -X_ii (L_i,: X) (two GEMMs per block
// row).  All descriptors are built once (static pointers) so a run is a fixed, fully asynchronous launch sequence.
struct BigChol {
  int n = 0, batch = 0, nb = 0;
  bool want_inverse = false, from_tri = false;   // from_tri: input is already a lower-triangular factor (inverse only)
  double* W = nullptr; double* Linv = nullptr; double* LinvT = nullptr; double* scal = nullptr; double* Tbuf = nullptr;
  int64_t stride = 0, scal_stride = 0;
  bool lookahead = false;              // k_chol_panel + wide update inside the factor launches (linalg.hip)
  bool need_factor = true;             // false: only the inverse / log det are read (the panels may stay parked above the diagonal)
  bool xrows = false;                  // ... and the block rows of the inverse (chol_xrow) instead of the recursive doubling
  const double* dinv = nullptr;        // inverse of diagonal block 0 of matrix 0; block p: + p * dinv_step, matrix b: + b * dinv_stride
  int64_t dinv_stride = 0, dinv_step = 0;
  PotrfItem* diag_items = nullptr;     // device: nb * batch
  std::vector<PotrfItem> items0;       // host: matrix 0's item of every diagonal block
  GemmProblem* gp = nullptr;           // device: per panel {solve, trailing}, then per block row {T, X}
  std::vector<int> tiles;              // per launch: planned tile count (bit 30: large-tile kernel)
  std::vector<int> nprob, first;       // per launch: number of problems and index of the first one in gp
  void* inv_block = nullptr;           // plan-owned scratch of the recursive-doubling inverse (when no LinvT is requested)
  void* dev_block = nullptr;
  void* tbuf_block = nullptr;         // plan-owned Tbuf (when the caller passed none)
};
// W/Linv/LinvT: `batch` matrices of order n (leading dimension n) `stride` doubles apart; scal: 2 doubles per matrix
// (`scal_stride` apart); Tbuf: batch * 64 * n doubles of scratch (needed when want_inverse).
int bigchol_build(dsdgp_ctx* ctx, BigChol& P, double* W, double* Linv, double* LinvT, double* scal, int batch, int64_t stride,
                  int64_t scal_stride, int n, int nreal, double* Tbuf, bool from_tri, bool need_factor = true);
int bigchol_run(dsdgp_ctx* ctx, const BigChol& P);
void bigchol_free(BigChol& P);

#ifdef __HIPCC__
// ---- Cholesky of a 16-column panel held one ROW per lane (a[j] = column j of this lane's row).  Lanes 0..15 hold the 16 x 16
// diagonal block; any further lanes hold rows of the panel BELOW it and come out as L_ij = A_ij L_jj^-T for free: at pivot J every
// lane scales its column-J entry by 1 / l_JJ and subtracts l_iJ l_KJ from its entries K > J, where l_KJ is read from diagonal lane K
// with v_readlane (lane indices are compile-time constants).  One hardware v_rsq_f64 + two Newton steps per pivot (the library
// rsqrt() expands to a ~800-cycle sqrt + divide chain), no f64 division, no exec-mask branches in the 16-pivot chain.
template <int J, int K>
struct Chol16Upd {
  static __device__ __forceinline__ void run(double (&a)[16], double lij) {
    const double lkj = bcast_lane<K>(lij);
    a[K] -= lij * lkj;
    Chol16Upd<J, K + 1>::run(a, lij);
  }
};
template <int J>
struct Chol16Upd<J, 16> {
  static __device__ __forceinline__ void run(double (&)[16], double) {}
};
template <int J>
struct Chol16 {
  static __device__ __forceinline__ void run(double (&a)[16], int i, double& myinv, int& bad) {
    const double ajj = bcast_lane<J>(a[J]);
    if (!(ajj > 0.0) && bad == 0) bad = J + 1;
    // hardware v_rsq_f64 seed + two Newton steps (the library rsqrt() expands to a ~800-cycle sqrt + divide chain,
    // 16 of them in sequence per diagonal block dominated the whole factorisation)
    double inv = __builtin_amdgcn_rsq(ajj);
    inv = inv * fma(-0.5 * ajj * inv, inv, 1.5);
    inv = inv * fma(-0.5 * ajj * inv, inv, 1.5);
    myinv = (i == J) ? inv : myinv;
    // no (i >= J) masking here: lanes above the diagonal carry garbage that never reaches a lower lane and is
    // zeroed when the block is stored — keeps the 16-pivot chain free of exec-mask branches
    const double lij = a[J] * inv;
    a[J] = lij;
    Chol16Upd<J, J + 1>::run(a, lij);
    Chol16<J + 1>::run(a, i, myinv, bad);
  }
};
template <>
struct Chol16<16> {
  static __device__ __forceinline__ void run(double (&)[16], int, double&, int&) {}
};

#endif
