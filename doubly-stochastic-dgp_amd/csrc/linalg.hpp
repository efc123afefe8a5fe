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
};

// One matrix of a batched factorisation launch (n multiple of 16; rows/cols >= nreal carry an identity pad).
struct PotrfItem {
  double* W;      // in: SPD matrix (lower part read); out: lower Cholesky factor, upper zeroed
  double* Linv;   // out: W^-1 (lower), may alias nothing else; may be NULL (skip inverse)
  double* LinvT;  // out: transpose of Linv (may be NULL)
  double* scal;   // out: [0] = sum_i<nreal 2 log L_ii (= logdet), [1] = info (0 ok, else 1-based bad pivot)
  int32_t n, ld, nreal, pad;
};

// Fills tile_start/tiles_* of `host` problems, returns the total number of 64x64 tiles.
int gemm_plan(GemmProblem* host, int nprob);
// Launch over problems already resident in device memory (`dev`), described by the planned `host` copy.
int gemm_launch(dsdgp_ctx* ctx, const GemmProblem* dev, int nprob, int total_tiles);
// n_max: largest (padded) matrix order among the items; <= 128 selects the LDS-resident variant
int potrf_launch(dsdgp_ctx* ctx, const PotrfItem* dev_items, int nitems, int n_max);
// batched inverse of padded lower-triangular matrices (n multiple of 16, identity pad), one workgroup each
int trtri_launch(dsdgp_ctx* ctx, double* W, double* Linv, int n, int64_t stride, int batch);
