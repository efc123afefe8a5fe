// Natural-gradient step ([UPSTREAM] NatGradOptimizer) and the full_cov = True path (layers.py:206-219, utils.py:43-51): kernels and entry
// points.  Part of the model translation unit.
#pragma once
// ------------------------------------------------------------------------------------------------------
// Natural-gradient step on one layer's (q_mu, q_sqrt)  — [UPSTREAM] gpflow.training.NatGradOptimizer(gamma)
// (demos/demo_regression_UCI.ipynb:360-366, tests/test_collapsed.py:100).
// Per output d:  Sbar = sym(T^-T Phi(T^T Tbar) T^-1);  theta1 = S^-1 m - gamma (mbar - 2 Sbar m);
//                A = S^-1 + 2 gamma Sbar (= -2 theta2);  S+ = A^-1;  m+ = S+ theta1;  T+ = chol(S+).
// ONE factorisation for the last two lines (round 4; rounds 1-3 factored A, formed S+ = A^-1 with a GEMM and factored S+ again): with
// J the index reversal, J A J = Lr Lr^T gives A = U U^T with U = J Lr J UPPER-triangular, so S+ = U^-T U^-1 and its lower Cholesky
// factor IS U^-T = J Lr^-T J (positive diagonal: the factor is unique):  T+[i][j] = Lr^-1[M-1-j][M-1-i],  m+ = T+ (T+^T theta1).
// The blocked factorisation of a 1024 x 1024 matrix is a ~0.7 ms launch sequence (linalg.hip): config 5 runs three per step instead of four.
// ------------------------------------------------------------------------------------------------------
__global__ void k_ng_prep(const LayerDev* __restrict__ layers, int l, const double* __restrict__ grad) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int64_t tot = (int64_t)v.D_out * Mp * Mp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx / ((int64_t)Mp * Mp)), rem = (int)(idx % ((int64_t)Mp * Mp)), i = rem / Mp, j = rem % Mp;
    const bool in = i < M && j <= i;
    v.ngTbar[idx] = in ? grad[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
    v.ngTI[idx] = (i < M) ? v.Tp[idx] : (i == j ? 1.0 : 0.0);
  }
}
__global__ void k_ng_phi(const LayerDev* __restrict__ layers, int l) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp;
  const int64_t tot = (int64_t)v.D_out * Mp * Mp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int rem = (int)(idx % ((int64_t)Mp * Mp)), i = rem / Mp, j = rem % Mp;
    const double h = v.ngH[idx];
    v.ngH[idx] = (j < i) ? h : (j == i ? 0.5 * h : 0.0);
  }
}
// A = S^-1 + 2 gamma Sbar, written INDEX-REVERSED inside its M x M block (J A J), identity pad ; Sbar stored (symmetrised) into ngY
__global__ void k_ng_assemble(const LayerDev* __restrict__ layers, int l, double gamma) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int64_t MM = (int64_t)Mp * Mp, tot = (int64_t)v.D_out * MM;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t base = idx / MM * MM;
    const int rem = (int)(idx % MM), i = rem / Mp, j = rem % Mp;
    double sb = 0.0, a = (i == j) ? 1.0 : 0.0;
    if (i < M && j < M) {
      sb = 0.5 * (v.ngX[base + i * Mp + j] + v.ngX[base + j * Mp + i]);
      a = v.ngSinv[idx] + 2.0 * gamma * sb;
    }
    v.ngY[idx] = sb;
    if (i < M && j < M) v.ngA[base + (int64_t)(M - 1 - i) * Mp + (M - 1 - j)] = a;
    else v.ngA[idx] = a;
  }
}
// matrix-vector products of the natural-gradient step: ONE WAVE PER ROW (the lanes walk the row: coalesced; a thread per row read
// M strided doubles one after the other — 275 us / 159 us at M = 1024).  blocks of 256 threads = 4 rows.
__global__ __launch_bounds__(256) void k_ng_theta1(const LayerDev* __restrict__ layers, int l, const double* __restrict__ grad, double gamma) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= v.D_out * M) return;
  const int d = row / M, i = row % M;
  const double* Sinv = v.ngSinv + (int64_t)d * Mp * Mp + (int64_t)i * Mp;
  const double* Sbar = v.ngY + (int64_t)d * Mp * Mp + (int64_t)i * Mp;
  double s1 = 0.0, s2 = 0.0;
  for (int j = lane; j < M; j += 64) {
    const double mj = v.qmu[j * v.D_out + d];
    s1 = fma(Sinv[j], mj, s1);
    s2 = fma(Sbar[j], mj, s2);
  }
  s1 = sum_wave(s1);
  s2 = sum_wave(s2);
  if (lane == 0) v.ngTheta1[d * Mp + i] = s1 - gamma * (grad[v.off_q_mu + (int64_t)i * v.D_out + d] - 2.0 * s2);
}
// m+ = T+ (T+^T theta1) with T+[i][j] = Lr^-1[M-1-j][M-1-i]: both products walk ROWS of the inverse factor / its transpose
//   w[j]  = sum_i T+[i][j] theta1[i] = sum_c Lr^-1[M-1-j][c] theta1[M-1-c]         (second = 0: row M-1-j of ngLAinv)
//   m+[i] = sum_j T+[i][j] w[j]      = sum_c Lr^-T[M-1-i][c] w[M-1-c]              (second = 1: row M-1-i of ngLAinvT)
__global__ __launch_bounds__(256) void k_ng_mu(const LayerDev* __restrict__ layers, int l, double* __restrict__ theta, int second) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= v.D_out * M) return;
  const int d = row / M, i = row % M;
  const double* R = (second ? v.ngLAinvT : v.ngLAinv) + (int64_t)d * Mp * Mp + (int64_t)(M - 1 - i) * Mp;
  const double* x = (second ? v.ngV : v.ngTheta1) + d * Mp;
  // only the triangle that holds the factor is read: columns <= the row of Lr^-1, >= the row of its transpose
  const int r = M - 1 - i;
  const int c_lo = second ? r : 0, c_hi = second ? M : r + 1;
  double s = 0.0;
  for (int c = c_lo + lane; c < c_hi; c += 64) s = fma(R[c], x[M - 1 - c], s);
  s = sum_wave(s);
  if (lane == 0) {
    if (second) theta[v.off_q_mu + (int64_t)i * v.D_out + d] = s;
    else v.ngV[d * Mp + i] = s;
  }
}
__global__ void k_ng_write(const LayerDev* __restrict__ layers, int l, double* __restrict__ theta) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int64_t tot = (int64_t)v.D_out * M * M;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx / ((int64_t)M * M)), rem = (int)(idx % ((int64_t)M * M)), i = rem / M, j = rem % M;
    theta[v.off_q_sqrt + idx] = (j <= i) ? v.ngLAinvT[((int64_t)d * Mp + (M - 1 - i)) * Mp + (M - 1 - j)] : 0.0;
  }
}

extern "C" int dsdgp_model_natgrad_step(dsdgp_model* m, int32_t l, double gamma, int* info) {
  DS_CHECK_ARG(m && m->grad && l >= 0 && l < m->desc.L && gamma > 0);
  dsdgp_ctx* ctx = m->ctx;
  LayerState& St = m->L[l];
  const LayerDev& v = St.dev;
  const int64_t MM = (int64_t)v.Mp * v.Mp;
  const int nb = (int)std::min<int64_t>(1024, ceil_div(v.D_out * MM, 256));
  // Tp / qmu must reflect the current theta
  if (!m->prepared) DS_TRY(prepare_async(m));
  DS_LAUNCH(k_ng_prep, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, m->grad);
  DS_HIP(hipGetLastError());
  if (St.big) DS_TRY(bigchol_run(ctx, St.big_ngT));
  else DS_TRY(trtri_launch(ctx, v.ngTI, v.ngTinv, v.Mp, MM, v.D_out));
  DS_TRY(gemm_launch(ctx, St.ng_gp, 2, St.ng_t1));
  DS_LAUNCH(k_ng_phi, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l);
  DS_TRY(gemm_launch(ctx, St.ng_gp + 2, 1, St.ng_t2));
  DS_TRY(gemm_launch(ctx, St.ng_gp + 3, 1, St.ng_t3));
  DS_LAUNCH(k_ng_assemble, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, gamma);
  DS_LAUNCH(k_ng_theta1, dim3(ceil_div(v.D_out * v.M, 4)), dim3(256), 0, ctx->stream, m->layers_dev, l, m->grad,
                     gamma);
  DS_HIP(hipGetLastError());
  if (St.big) DS_TRY(bigchol_run(ctx, St.big_ngA));
  else DS_TRY(potrf_launch(ctx, St.ng_items, v.D_out, v.Mp));
  DS_LAUNCH(k_ng_mu, dim3(ceil_div(v.D_out * v.M, 4)), dim3(256), 0, ctx->stream, m->layers_dev, l, m->theta, 0);
  DS_LAUNCH(k_ng_mu, dim3(ceil_div(v.D_out * v.M, 4)), dim3(256), 0, ctx->stream, m->layers_dev, l, m->theta, 1);
  DS_LAUNCH(k_ng_write, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, m->theta);
  DS_HIP(hipGetLastError());
  m->prepared = false;
  m->q_dirty = (m->q_dirty == -1 || m->q_dirty == l) ? l : -2;     // Z and the kernel hyper-parameters are untouched: kuu_valid stays
  if (info) {
    std::vector<double> sc(2 * v.D_out);
    DS_HIP(hipMemcpyAsync(sc.data(), v.ngScal, sc.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
    *info = 0;
    for (int d = 0; d < v.D_out; ++d)
      if (sc[2 * d + 1] != 0.0 && *info == 0) *info = (int)sc[2 * d + 1];
    if (*info) {
      dsdgp_set_error("natural-gradient step left q(u) covariance non-SPD (gamma too large?): pivot %d", *info);
      return DSDGP_ERR_NOT_SPD;
    }
  }
  return DSDGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// full_cov=True: SVGP_Layer.conditional_ND(full_cov=True) (layers.py:206-209,216-219) and
// reparameterize(full_cov=True) (utils.py:43-51).  Plot-sized inputs; composed from the gram / grouped-GEMM / potrf
// kernels (same algebra as the diagonal path: var_d = Kff - A1^T A1 + (T_d^T A)^T (T_d^T A)).
// ------------------------------------------------------------------------------------------------------
__global__ void k_fullcov_combine(const double* __restrict__ Kff, const double* __restrict__ Q, const double* __restrict__ P,
                                  int64_t n, int D, double* __restrict__ var) {
  const int64_t tot = n * n * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D);
    const int64_t ij = idx / D;
    var[idx] = Kff[ij] - Q[ij] + P[(int64_t)d * n * n + ij];      // (n, n, D) layout, layers.py:216-217
  }
}
__global__ void k_add_mean_fn(double* __restrict__ mean, const double* __restrict__ X, int64_t n, int D_in, int D_out,
                              int mean_kind, const double* __restrict__ A, const double* __restrict__ bias) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * D_out; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / D_out;
    const int d = (int)(idx % D_out);
    double v = mean[idx];
    if (mean_kind == DSDGP_MEAN_IDENTITY) {
      v += X[i * D_in + d];
    } else if (mean_kind == DSDGP_MEAN_LINEAR) {
      for (int j = 0; j < D_in; ++j) v = fma(X[i * D_in + j], A[j * D_out + d], v);
      if (bias) v += bias[d];
    }
    mean[idx] = v;
  }
}

static int run_gemms(dsdgp_ctx* ctx, std::vector<GemmProblem>& probs, GemmProblem* dev) {
  const int total = gemm_plan(probs.data(), (int)probs.size());
  DS_HIP(hipMemcpyAsync(dev, probs.data(), probs.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, ctx->stream));
  DS_HIP(hipStreamSynchronize(ctx->stream));
  return gemm_launch(ctx, dev, (int)probs.size(), total);
}

extern "C" int dsdgp_model_layer_conditional_full(dsdgp_model* m, int32_t l, const double* X, int64_t n, double* mean,
                                                  double* var) {
  DS_CHECK_ARG(m && X && mean && var && l >= 0 && l < m->desc.L && n > 0);
  if (!m->prepared) DS_TRY(prepare_async(m));
  dsdgp_ctx* ctx = m->ctx;
  LayerState& St = m->L[l];
  const LayerDev& v = St.dev;
  const int Mp = v.Mp, D = v.D_out;
  const int64_t MN = (int64_t)Mp * n, NN = n * n;
  // scratch: Kuf, A1, A (Mp x n) | C (D x Mp x n) | Kff, Q (n x n) | P (D x n x n) | gemm problem list
  const size_t bytes = (size_t)((3 + D) * MN + (2 + D) * NN) * sizeof(double) + 8 * sizeof(GemmProblem) + 256;
  void* scr;
  DS_TRY(ctx_scratch(ctx, bytes, &scr));
  double* Kuf = (double*)scr;
  double* A1 = Kuf + MN;
  double* A = A1 + MN;
  double* Cd = A + MN;
  double* Kff = Cd + (int64_t)D * MN;
  double* Q = Kff + NN;
  double* P = Q + NN;
  GemmProblem* gp = (GemmProblem*)(((uintptr_t)(P + (int64_t)D * NN) + 255) & ~(uintptr_t)255);
  DS_HIP(hipMemsetAsync(Kuf, 0, MN * sizeof(double), ctx->stream));
  DS_TRY(gram_launch(ctx, v.kern_kind, v.Zp, v.M, X, n, v.D_in, v.hyp, 0.0, 0, Kuf, n));                 // layers.py:184
  // Kff = kern.K(X) (layers.py:209): White contributes on the diagonal, no jitter.  hyp[HYP_WVAR] lives on the device:
  // build with diag_add = 0 and add the white variance in the combine step through Q (subtract a negative).
  DS_TRY(gram_launch(ctx, v.kern_kind, X, n, X, n, v.D_in, v.hyp, 0.0, 1, Kff, n));
  std::vector<GemmProblem> g1(1), g2, g3;
  fill_gemm(g1[0], v.Linv, Kuf, A1, Mp, (int)n, Mp, Mp, (int)n, (int)n, 0, 0, 1, 0, 0, 0, 0);            // layers.py:186
  DS_TRY(run_gemms(ctx, g1, gp));
  const double* Ause = A1;
  if (!m->desc.white) {
    std::vector<GemmProblem> ga(1);
    fill_gemm(ga[0], v.LinvT, A1, A, Mp, (int)n, Mp, Mp, (int)n, (int)n, 0, 0, 1, 0, 0, 0, 0);           // layers.py:188
    DS_TRY(run_gemms(ctx, ga, gp));
    Ause = A;
  }
  GemmProblem pm, pc, pq;
  fill_gemm(pm, Ause, v.qmu, mean, (int)n, D, Mp, (int)n, D, D, 1, 0, 1, 0, 0, 0, 0);                    // layers.py:190
  fill_gemm(pc, v.Tp, Ause, Cd, Mp, (int)n, Mp, Mp, (int)n, (int)n, 1, 0, D, (int64_t)Mp * Mp, 0, MN, 0);  // q_sqrt_d^T A
  fill_gemm(pq, A1, A1, Q, (int)n, (int)n, Mp, (int)n, (int)n, (int)n, 1, 0, 1, 0, 0, 0, 0);             // A1^T A1
  g2 = {pm, pc, pq};
  DS_TRY(run_gemms(ctx, g2, gp));
  GemmProblem pp;
  fill_gemm(pp, Cd, Cd, P, (int)n, (int)n, Mp, (int)n, (int)n, (int)n, 1, 0, D, MN, MN, NN, 0);
  g3 = {pp};
  DS_TRY(run_gemms(ctx, g3, gp));
  const int nb = (int)std::min<int64_t>(2048, ceil_div(NN * D, 256));
  DS_LAUNCH(k_fullcov_combine, dim3(nb), dim3(256), 0, ctx->stream, Kff, Q, P, n, D, var);
  DS_LAUNCH(k_add_mean_fn, dim3(ceil_div(n * D, 256)), dim3(256), 0, ctx->stream, mean, X, n, v.D_in, D,
                     St.d.mean_kind, St.meanA, St.meanb);
  DS_HIP(hipGetLastError());
  if (v.has_white) {
    // add the White variance on the diagonal of every output's covariance (Kff of a Sum kernel)
    extern __global__ void k_add_diag_dev(double*, int64_t, int, const double*);
    DS_LAUNCH(k_add_diag_dev, dim3(ceil_div(n * D, 256)), dim3(256), 0, ctx->stream, var, n, D, v.hyp + HYP_WVAR);
    DS_HIP(hipGetLastError());
  }
  return DSDGP_OK;
}

__global__ void k_add_diag_dev(double* __restrict__ var, int64_t n, int D, const double* __restrict__ val) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * D) return;
  const int64_t i = idx / D;
  const int d = (int)(idx % D);
  var[(i * n + i) * D + d] += val[0];
}

// reparameterize(full_cov=True) (utils.py:43-51): f[s,:,d] = mean[s,:,d] + chol(var[s,:,:,d] + jitter I) z[s,:,d]
__global__ void k_fullcov_gather(const double* __restrict__ var, int64_t n, int D, int S, double jitter, double* __restrict__ out) {
  const int64_t tot = (int64_t)S * D * n * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx % n, i = (idx / n) % n, d = (idx / (n * n)) % D, s = idx / (n * n * D);
    out[idx] = var[((s * n + i) * n + j) * D + d] + (i == j ? jitter : 0.0);          // SNND -> SDNN (+ jitter I)
  }
}
__global__ void k_fullcov_sample(const double* __restrict__ Lc, const double* __restrict__ mean, const double* __restrict__ z,
                                 int64_t n, int D, int S, double* __restrict__ out) {
  const int64_t tot = (int64_t)S * n * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = idx % D, i = (idx / D) % n, s = idx / (D * n);
    const double* Lrow = Lc + ((s * D + d) * n + i) * n;
    double acc = mean[idx];
    for (int64_t j = 0; j <= i; ++j) acc = fma(Lrow[j], z[(s * n + j) * D + d], acc);
    out[idx] = acc;
  }
}

extern "C" int dsdgp_reparameterize_full(dsdgp_ctx* ctx, const double* mean, const double* var, const double* z, double jitter,
                                         int64_t n, int32_t D, int32_t S, double* out) {
  DS_CHECK_ARG(ctx && mean && var && z && out && n > 0 && D > 0 && S > 0);
  const int64_t nmat = (int64_t)S * D;
  double* Lc = nullptr;
  DS_HIP(hipMallocAsync((void**)&Lc, nmat * n * n * sizeof(double), ctx->stream));
  const int nb = (int)std::min<int64_t>(4096, ceil_div(nmat * n * n, 256));
  DS_LAUNCH(k_fullcov_gather, dim3(nb), dim3(256), 0, ctx->stream, var, n, D, S, jitter, Lc);
  DS_HIP(hipGetLastError());
  int info = 0;
  int rc = dsdgp_potrf(ctx, (int)nmat, (int)n, Lc, n, n * n, &info);
  if (rc == DSDGP_OK) {
    DS_LAUNCH(k_fullcov_sample, dim3(ceil_div((int64_t)S * n * D, 256)), dim3(256), 0, ctx->stream, Lc, mean, z, n, D, S,
                       out);
    if (hipGetLastError() != hipSuccess) rc = DSDGP_ERR_HIP;
  }
  hipFreeAsync(Lc, ctx->stream);
  return rc;
}

