// DGP_Base / SVGP_Layer orchestration (dgp.py:42-126, layers.py:122-246) over the HIP kernels of this library.
// One C call enqueues a whole ELBO evaluation (+ reverse-mode gradient) on the ctx stream; nothing here touches the host
// between kernels except reading the final scalars when the caller asks for them.
#include "layer.hpp"
#include "linalg.hpp"
#include <stdlib.h>
#include <hip/hip_ext.h>

int multiclass_launch(dsdgp_ctx* ctx, const double* mean, const double* var, const double* Y, int64_t n, int64_t R, int K,
                      int mode, double wgt, double* out, double* dmean, double* dvar, int y_override);
int gram_launch(dsdgp_ctx* ctx, int kind, const double* X, int64_t n, const double* X2, int64_t n2, int D,
                const double* hyp_dev, double diag_add, int symmetric, double* out, int64_t ld);

#define SOFTPLUS_LOWER 1e-6  // [UPSTREAM] gpflow.transforms.positive

#include "model_types.hpp"      // descriptors, policies
#include "model_layout.hpp"     // workspace layout
#include "model_kernels.hpp"    // device kernels (transforms, KL, likelihoods, reduction, assembly, tail)

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int validate_desc(const dsdgp_model_desc* d) {
  DS_CHECK_ARG(d && d->L >= 1 && d->L <= DSDGP_MAX_LAYERS && d->n_theta > 0);
  DS_CHECK_ARG(d->lik_kind == DSDGP_LIK_GAUSSIAN || d->lik_kind == DSDGP_LIK_MULTICLASS || d->lik_kind == DSDGP_LIK_BERNOULLI ||
               lik_is_generic(d->lik_kind));
  DS_CHECK_ARG((d->lik_kind != DSDGP_LIK_POISSON && d->lik_kind != DSDGP_LIK_STUDENT_T) || d->lik_aux > 0.0);      // binsize / deg_free
  DS_CHECK_ARG(!lik_has_param(d->lik_kind) || (d->off_lik_var >= 0 && d->off_lik_var < d->n_theta));
  for (int l = 0; l < d->L; ++l) {
    const dsdgp_layer_desc& y = d->layers[l];
    DS_CHECK_ARG(y.M >= 1 && y.D_in >= 1 && y.D_out >= 1);
    DS_CHECK_ARG(y.kern_kind == DSDGP_KERN_RBF || y.kern_kind == DSDGP_KERN_MATERN52);
    // the chain kernels hold a row block's activations in LDS and are built up to Mp = 1024; above it the GEMM-formulated passes
    // (layer_gemm.hip) are the only form
    if (pad_M(y.M) > DSDGP_MAX_MP) {
      dsdgp_set_error("layer %d: M=%d inducing points: built up to M = %d", l, y.M, DSDGP_MAX_MP);
      return DSDGP_ERR_UNSUPPORTED;
    }
    DS_CHECK_ARG(y.input_prop_dim >= 0 && y.input_prop_dim <= y.D_in);
    if (l > 0) DS_CHECK_ARG(y.D_in == d->layers[l - 1].D_out + d->layers[l - 1].input_prop_dim);   // layers.py:105-110
    if (y.mean_kind == DSDGP_MEAN_IDENTITY) DS_CHECK_ARG(y.D_in == y.D_out);
    if (y.mean_kind == DSDGP_MEAN_LINEAR) DS_CHECK_ARG(y.mean_A != nullptr || y.off_mean_A >= 0);
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_model_workspace_bytes(const dsdgp_model_desc* desc, int64_t n_max, int32_t s_max, int64_t* bytes) {
  DS_TRY(validate_desc(desc));
  DS_CHECK_ARG(bytes && n_max > 0 && s_max > 0);
  dsdgp_model tmp{};
  parse_force(&tmp);
  tmp.desc = *desc;
  tmp.n_max = n_max;
  tmp.s_max = s_max;
  size_t total = 0;
  layout(&tmp, nullptr, &total);
  *bytes = (int64_t)total;
  return DSDGP_OK;
}

static void fill_gemm(GemmProblem& P, const double* A, const double* B, double* C, int m, int n, int k, int lda, int ldb,
                      int ldc, int tA, int tB, int batch, int64_t sA, int64_t sB, int64_t sC, int reduce) {
  memset(&P, 0, sizeof(P));
  P.A = A; P.B = B; P.C = C;
  P.m = m; P.n = n; P.k = k;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc;
  P.transA = tA; P.transB = tB;
  P.batch = batch; P.sA = sA; P.sB = sB; P.sC = sC; P.batch_reduce = reduce;
  P.alpha = 1.0; P.beta = 0.0;
}

extern "C" int dsdgp_model_create(dsdgp_ctx* ctx, const dsdgp_model_desc* desc, int64_t n_max, int32_t s_max,
                                  double* theta, double* grad, double* adam_m, double* adam_v, void* workspace,
                                  int64_t workspace_bytes, dsdgp_model** out) {
  DS_CHECK_ARG(ctx && out && theta && workspace);
  DS_TRY(validate_desc(desc));
  DS_CHECK_ARG(((uintptr_t)workspace & 255) == 0);
  // the optimiser sweeps theta / grad / adam_m / adam_v with 16-byte vector accesses (adam_sweep)
  DS_CHECK_ARG(((uintptr_t)theta & 15) == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)adam_m & 15) == 0 && ((uintptr_t)adam_v & 15) == 0);
  dsdgp_model* m = new dsdgp_model();
  parse_force(m);
  m->ctx = ctx;
  m->desc = *desc;
  m->n_max = n_max;
  m->s_max = s_max;
  m->theta = theta; m->grad = grad; m->adam_m = adam_m; m->adam_v = adam_v;
  size_t total = 0;
  layout(m, (char*)workspace, &total);
  for (int l = 0; l < desc->L; ++l)
    if (m->L[l].dev.Mp > 1024 && !m->L[l].gemm) {      // (only reachable with the DSDGP_FORCE=gemm_mp=... test hook)
      dsdgp_set_error("layer %d: M=%d needs the GEMM-formulated passes, which DSDGP_FORCE switched off", l, desc->layers[l].M);
      delete m;
      return DSDGP_ERR_UNSUPPORTED;
    }
  if ((int64_t)total > workspace_bytes) {
    dsdgp_set_error("workspace too small: need %zu bytes, got %lld", total, (long long)workspace_bytes);
    delete m;
    return DSDGP_ERR_WORKSPACE;
  }
  hipStream_t st = ctx->stream;
  DS_HIP(hipMemsetAsync(workspace, 0, total, st));
  for (int l = 0; l < desc->L; ++l) {
    LayerState& S = m->L[l];
    S.meanA = (S.dev.off_mean_A >= 0) ? theta + S.dev.off_mean_A : S.d.mean_A;
    S.meanb = (S.dev.off_mean_b >= 0) ? theta + S.dev.off_mean_b : nullptr;
  }
  const int L = desc->L;
  std::vector<LayerDev> ld(L);
  std::vector<PotrfItem> items(L);
  std::vector<GemmProblem> gf, g1, g2, w1, w2, w3, wz, gpt;
  // plan a launch and give it its longest-processing-time tile list (device copy in the model's pool; in index order when the pool is full)
  m->gemm_order_used = 0;
  auto plan_lpt = [&](GemmProblem* probs, int n) -> int {
    std::vector<int32_t> order;
    const int total = gemm_plan_lpt(probs, n, order);
    if (order.empty() || m->gemm_order_used + (int64_t)order.size() > m->gemm_order_cap) return gemm_plan(probs, n);
    int32_t* dst = m->gemm_order + m->gemm_order_used;
    // (on the model's stream: behind the asynchronous clearing of the workspace)
    if (hipMemcpyAsync(dst, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return gemm_plan(probs, n);
    for (int i = 0; i < n; ++i) {
      probs[i].order = dst;
      probs[i].n_order = (int32_t)(order.size() / 2);
    }
    m->gemm_order_used += (int64_t)order.size();
    return total;
  };
  int64_t asm_elems = 0;
  for (int l = 0; l < L; ++l) {
    const LayerDev& v = m->L[l].dev;
    ld[l] = v;
    items[l] = PotrfItem{v.Kp, v.Linv, v.LinvT, v.scal, v.Mp, v.Mp, v.M,
                         getenv("DSDGP_POTRF_TIMING") ? 7 : (desc->white ? 0 : 8) /* Lu itself is only read by the white adjoint */, 0, 0};
    const int Mp = v.Mp;
    const int64_t MM = (int64_t)Mp * Mp;
    GemmProblem P;
    const size_t gf0 = gf.size() + 1, g10 = g1.size(), g20 = g2.size(), gp0 = gpt.size();   // this layer's q-dependent problems (Ku^-1 excluded)
    fill_gemm(P, v.LinvT, v.Linv, v.Kinv, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, 1, 0, 0, 0, 0);              // Ku^-1
    P.lower_only = 1; P.tri = 8 | 1 | 16;                                                              //   upper x lower, symmetric
    gf.push_back(P);
    fill_gemm(P, v.Linv, v.Tp, v.V, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, 0, MM, MM, 0);            // Lu^-1 q_sqrt
    P.lower_only = 1; P.tri = 2 | 1;                                                                   //   lower x lower = lower
    gf.push_back(P);
    fill_gemm(P, v.Linv, v.qmu4, v.nL, Mp, v.DP4, Mp, Mp, v.DP4, v.DP4, 0, 0, 1, 0, 0, 0, 0);        // Lu^-1 q_mu
    P.tri = 2;
    gf.push_back(P);
    // S_d = q_sqrt_d q_sqrt_d^T feeds the dense backward chain and KS_d; layers whose backward chain always takes the Csave
    // form (Mp >= 512) and that do not assemble dl/dKu algebraically never read it
    if (!(Mp > 256 && save_c_enabled(m, Mp)) || v.alg_g) {
      fill_gemm(P, v.Tp, v.Tp, v.Sd, Mp, Mp, Mp, Mp, Mp, Mp, 0, 1, v.D_out, MM, MM, MM, 0);            // S_d
      P.lower_only = 1; P.tri = 2 | 4 | 16;                                                              //   lower x lower^T, symmetric
      gf.push_back(P);
    }
    fill_gemm(P, v.Kinv, v.Tp, v.U, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, 0, MM, MM, 0);            // U_d
    P.tri = 1;
    g1.push_back(P);
    fill_gemm(P, v.Kinv, v.qmu4, v.n4, Mp, v.DP4, Mp, Mp, v.DP4, v.DP4, 0, 0, 1, 0, 0, 0, 0);        // n
    g1.push_back(P);
    fill_gemm(P, v.bigred + MM, v.Tp, v.PT, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, MM, MM, MM, 0);   // P_d T_d
    P.lower_only = 1; P.tri = 1;                                                                       //   only tril(P_d T_d) is read
    gpt.push_back(P);
    if (v.alg_g) {
      fill_gemm(P, v.Kinv, v.Sd, v.KS, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, 0, MM, MM, 0);         // KS_d = Ku^-1 S_d
      g1.push_back(P);
      fill_gemm(P, v.KS, v.bigred + MM, v.GS, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, MM, MM, MM, 0); // GS_d = KS_d P_d
      gpt.push_back(P);
    }
    fill_gemm(P, v.U, v.U, v.UU, Mp, Mp, Mp, Mp, Mp, Mp, 0, 1, v.D_out, MM, MM, MM, 0);              // U_d U_d^T
    P.lower_only = 1; P.tri = 16;
    g2.push_back(P);
    if (v.D_in > WIDE_DIN) {
      fill_gemm(P, v.wm, v.Zp1, v.WZ, Mp, v.DinP16, Mp, Mp, v.DinP16, v.DinP16, 0, 0, 1, 0, 0, 0, 0);  // wm [Z | 1]
      wz.push_back(P);
    }
    {
      LayerState& Sq = m->L[l];
      std::vector<GemmProblem> lq(gf.begin() + gf0, gf.end());
      Sq.lq_nf = (int)lq.size();
      Sq.lq_tf = plan_lpt(lq.data(), Sq.lq_nf);
      std::vector<GemmProblem> q1(g1.begin() + g10, g1.end()), q2(g2.begin() + g20, g2.end());
      Sq.lq_n1 = (int)q1.size(); Sq.lq_t1 = plan_lpt(q1.data(), Sq.lq_n1);
      Sq.lq_n2 = (int)q2.size(); Sq.lq_t2 = plan_lpt(q2.data(), Sq.lq_n2);
      std::vector<GemmProblem> qp(gpt.begin() + gp0, gpt.end());
      Sq.lq_np = (int)qp.size(); Sq.lq_tp = plan_lpt(qp.data(), Sq.lq_np);
      lq.insert(lq.end(), q1.begin(), q1.end());
      lq.insert(lq.end(), q2.begin(), q2.end());
      lq.insert(lq.end(), qp.begin(), qp.end());
      DS_CHECK_ARG(lq.size() <= 12);
      DS_HIP(hipMemcpyAsync(Sq.lq, lq.data(), lq.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
      DS_HIP(hipStreamSynchronize(st));
    }
    m->kuu_blocks = std::max(m->kuu_blocks, std::min(1024, (Mp / 16) * (Mp / 16)));
    asm_elems = std::max<int64_t>(asm_elems, (int64_t)v.D_out * v.M * v.M);
    m->prep_blocks = (int)std::max<int64_t>(m->prep_blocks, std::min<int64_t>(2048, (int64_t)v.D_out * Mp * Mp / 2048));
    // white=True: d l/d Ku from d l/d Lu = -tril(G) through the Cholesky adjoint  Ku_bar = sym(Lu^-T Phi(Lu^T Lu_bar) Lu^-1)
    fill_gemm(P, v.bigred + MM, v.Tp, v.PT, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, MM, MM, MM, 0);   // P_d T_d
    w1.push_back(P);
    fill_gemm(P, v.Kp, v.wLbar, v.wH, Mp, Mp, Mp, Mp, Mp, Mp, 1, 0, 1, 0, 0, 0, 0);                  // Lu^T Lu_bar
    w1.push_back(P);
    fill_gemm(P, v.wH, v.Linv, v.wY, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, 1, 0, 0, 0, 0);                   // Phi Lu^-1
    w2.push_back(P);
    fill_gemm(P, v.Linv, v.wY, v.wX, Mp, Mp, Mp, Mp, Mp, Mp, 1, 0, 1, 0, 0, 0, 0);                   // Lu^-T (Phi Lu^-1)
    w3.push_back(P);
    {
      LayerState& St = m->L[l];
      GemmProblem ng[4];
      fill_gemm(ng[0], v.ngTI, v.ngTbar, v.ngH, Mp, Mp, Mp, Mp, Mp, Mp, 1, 0, v.D_out, MM, MM, MM, 0);        // T^T Tbar
      fill_gemm(ng[1], v.ngTinv, v.ngTinv, v.ngSinv, Mp, Mp, Mp, Mp, Mp, Mp, 1, 0, v.D_out, MM, MM, MM, 0);   // S^-1
      ng[0].lower_only = 1; ng[0].tri = 8 | 1;          // upper x lower; k_ng_phi keeps tril(H) only
      ng[1].lower_only = 1; ng[1].tri = 8 | 1 | 16;     // upper x lower, symmetric (as Ku^-1)
      St.ng_t1 = plan_lpt(ng, 2);
      fill_gemm(ng[2], v.ngH, v.ngTinv, v.ngY, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, MM, MM, MM, 0);         // Phi T^-1
      ng[2].tri = 2 | 1;                                // lower x lower = lower (the tiles above the diagonal come out as zeros: ng[3] reads them)
      St.ng_t2 = plan_lpt(ng + 2, 1);
      fill_gemm(ng[3], v.ngTinv, v.ngY, v.ngX, Mp, Mp, Mp, Mp, Mp, Mp, 1, 0, v.D_out, MM, MM, MM, 0);         // T^-T Phi T^-1
      ng[3].tri = 8 | 1;                                // upper x lower
      St.ng_t3 = plan_lpt(ng + 3, 1);
      DS_HIP(hipMemcpyAsync(St.ng_gp, ng, sizeof(ng), hipMemcpyHostToDevice, st));
      std::vector<PotrfItem> it(v.D_out);
      for (int d = 0; d < v.D_out; ++d)
        it[d] = PotrfItem{v.ngA + d * MM, v.ngLAinv + d * MM, v.ngLAinvT + d * MM, v.ngScal + 2 * d, Mp, Mp, v.M, 0, 0, 0};
      DS_HIP(hipMemcpyAsync(St.ng_items, it.data(), it.size() * sizeof(PotrfItem), hipMemcpyHostToDevice, st));
      DS_HIP(hipStreamSynchronize(st));
      St.big = Mp >= big_mp(m->uniform_big) && Mp % 64 == 0;
      if (St.big) {
        // (Lu itself is read by the white = True adjoint only, the factor of the natural-gradient step's A by nobody)
        const bool need_lu = desc->white != 0;
        if (!m->uniform_big) DS_TRY(bigchol_build(ctx, St.big_k, v.Kp, v.Linv, v.LinvT, v.scal, 1, MM, 2, Mp, v.M, nullptr, false, need_lu));
        else if (l == 0) DS_TRY(bigchol_build(ctx, m->big_all, v.Kp, v.Linv, v.LinvT, v.scal, L, MM, 8, Mp, v.M, nullptr, false, need_lu));
        DS_TRY(bigchol_build(ctx, St.big_ngA, v.ngA, v.ngLAinv, v.ngLAinvT, v.ngScal, v.D_out, MM, 2, Mp, v.M, nullptr, false, false));
        DS_TRY(bigchol_build(ctx, St.big_ngT, v.ngTI, v.ngTinv, nullptr, nullptr, v.D_out, MM, 0, Mp, v.M, nullptr, true));
      }
    }
  }
  m->tail_ok = m->force.tail != 0 && !desc->white;
  for (int l = 0; l < L; ++l) {
    m->tail_ok = m->tail_ok && m->L[l].dev.D_in <= WIDE_DIN;
    m->mp_max_all = std::max(m->mp_max_all, (int)m->L[l].dev.Mp);
    m->m_max_all = std::max(m->m_max_all, (int)m->L[l].dev.M);
  }
  for (int l = 0; l < L; ++l) {
    LayerDev& v = m->L[l].dev;
    const bool fold = v.D_in <= WIDE_DIN && (int64_t)m->kuu_blocks * 256 >= (int64_t)v.Mp * v.Mp;
    v.hyp_parts = m->tail_ok ? v.M : (fold ? m->kuu_blocks : -NPART);      // fused tail: one partial row per inducing row (k_asm_rows)
    ld[l].hyp_parts = v.hyp_parts;
    v.kl_parts = ld[l].kl_parts = m->mp_max_all >= 512 ? 512 : NPART;       // = the grid of k_kl_part (prepare_async)
    if (!fold) m->need_hyp_part = true;
  }
  m->head_ok = m->force.head != 0 && !m->uniform_big;
  for (int l = 0; l < L; ++l) m->head_ok = m->head_ok && m->L[l].dev.Mp <= HEAD_MAX_N && m->L[l].dev.D_in <= HEAD_MAX_DIN;
  m->n_fwd = (int)gf.size(); m->t_fwd = plan_lpt(gf.data(), m->n_fwd);
  m->n_bwd1 = (int)g1.size(); m->t_bwd1 = plan_lpt(g1.data(), m->n_bwd1);
  m->n_pt = (int)gpt.size(); m->t_pt = plan_lpt(gpt.data(), m->n_pt);
  DS_HIP(hipMemcpyAsync(m->gp_pt, gpt.data(), gpt.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  m->n_bwd2 = (int)g2.size(); m->t_bwd2 = plan_lpt(g2.data(), m->n_bwd2);
  m->n_wz = (int)wz.size();
  if (m->n_wz) {
    m->t_wz = plan_lpt(wz.data(), m->n_wz);
    DS_HIP(hipMemcpyAsync(m->gp_wz, wz.data(), wz.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  }
  m->asm_blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(64, asm_elems / 1024));
  m->n_w = L;
  m->t_w1 = plan_lpt(w1.data(), 2 * L); m->t_w2 = plan_lpt(w2.data(), L); m->t_w3 = plan_lpt(w3.data(), L);
  DS_HIP(hipMemcpyAsync(m->gp_w1, w1.data(), w1.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->gp_w2, w2.data(), w2.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->gp_w3, w3.data(), w3.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->layers_dev, ld.data(), L * sizeof(LayerDev), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->potrf_items, items.data(), L * sizeof(PotrfItem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->gp_fwd, gf.data(), gf.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->gp_bwd1, g1.data(), g1.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  DS_HIP(hipMemcpyAsync(m->gp_bwd2, g2.data(), g2.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, st));
  // trainable mask (set_trainable(False) of the reference, e.g. tests/test_dgp.py:141-145)
  std::vector<double> mask(desc->n_theta, 0.0);
  auto mark = [&](int64_t off, int64_t cnt, int on) {
    for (int64_t i = 0; i < cnt; ++i) mask[off + i] = (double)on;
  };
  for (int l = 0; l < L; ++l) {
    const dsdgp_layer_desc& y = desc->layers[l];
    mark(y.off_Z, (int64_t)y.M * y.D_in, y.trainable_Z);
    mark(y.off_q_mu, (int64_t)y.M * y.D_out, y.trainable_q_mu);
    // q_sqrt: the entries on or below the diagonal only — above it the parameter is structurally zero with a zero gradient (its Adam
    // update is the identity), and at large M that half of theta is most of what the optimiser sweep would move
    if (y.trainable_q_sqrt)
      for (int64_t d = 0; d < y.D_out; ++d)
        for (int64_t i = 0; i < y.M; ++i) mark(y.off_q_sqrt + (d * y.M + i) * y.M, i + 1, 1);
    // (2: entries whose gradient k_tail's hyper-parameter / likelihood blocks produce and, in a fused training step, update themselves)
    mark(y.off_kvar, 1, 2 * y.trainable_kvar);
    mark(y.off_kls, y.ard ? y.D_in : 1, 2 * y.trainable_kls);
    if (y.has_white) mark(y.off_wvar, 1, 2 * y.trainable_wvar);
    if (y.mean_kind == DSDGP_MEAN_LINEAR && y.off_mean_A >= 0) mark(y.off_mean_A, (int64_t)y.D_in * y.D_out, y.trainable_mean_A);
    if (y.mean_kind == DSDGP_MEAN_LINEAR && y.off_mean_b >= 0) mark(y.off_mean_b, y.D_out, y.trainable_mean_b);
  }
  if (lik_has_param(desc->lik_kind)) mark(desc->off_lik_var, 1, 2 * desc->trainable_lik_var);
  DS_HIP(hipMemcpyAsync(m->mask, mask.data(), mask.size() * sizeof(double), hipMemcpyHostToDevice, st));
  DS_HIP(hipStreamSynchronize(st));
  m->overlap = !(getenv("DSDGP_NO_OVERLAP") && atoi(getenv("DSDGP_NO_OVERLAP")));
  // ONE side stream per context, shared by its models: a stream per model left the mapping of streams to hardware queues to the
  // order in which models had been created and destroyed (a process that had built several models occasionally ran a later one
  // 15 - 140 % slower, tools/ab_force.py)
  if (!ctx->side) DS_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  m->side = ctx->side;

  const unsigned evf = hipEventDisableTiming;   // (+ hipEventDisableSystemFence measured neutral in round 5: 0.558 ms either way)
  for (int l = 0; l < L; ++l) {
    DS_HIP(hipEventCreateWithFlags(&m->ev_bwd[l], evf));
  }
  DS_HIP(hipEventCreateWithFlags(&m->ev_side, evf));
  DS_HIP(hipEventCreateWithFlags(&m->ev_fork, evf));
  DS_HIP(hipEventCreateWithFlags(&m->ev_prep_side, evf));
  DS_HIP(hipEventCreateWithFlags(&m->ev_z, evf));
  DS_HIP(hipEventCreateWithFlags(&m->ev_kinv, evf));
  m->prepared = false;
  m->plan_n = -1;
  m->plan_S = -1;
  *out = m;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_destroy(dsdgp_model* m) {
  if (m) {
    hipStreamSynchronize(m->ctx->stream);
    hipStreamSynchronize(m->side);

    for (int l = 0; l < m->desc.L; ++l) {
      hipEventDestroy(m->ev_bwd[l]);
    }
    hipEventDestroy(m->ev_side);

    hipEventDestroy(m->ev_fork); hipEventDestroy(m->ev_prep_side); hipEventDestroy(m->ev_z); hipEventDestroy(m->ev_kinv);
    for (int l = 0; l < m->desc.L; ++l) {
      if (l == 0) bigchol_free(m->big_all);
      bigchol_free(m->L[l].big_k); bigchol_free(m->L[l].big_ngA); bigchol_free(m->L[l].big_ngT);
    }
    delete m;
  }
  return DSDGP_OK;
}

#include "model_schedule.hpp"   // prepare / forward / plan / backward / ELBO / training step

extern "C" int dsdgp_model_train_step(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S, const double* const* zs,
                                      const int64_t* zstride, uint64_t seed, double data_scale, double kl_weight, double lr, double beta1,
                                      double beta2, double eps, int64_t t, double* out) {
  DS_CHECK_ARG(X && Y);
  return train_step_impl(m, X, Y, n, S, zs, zstride, seed, data_scale, kl_weight, lr, beta1, beta2, eps, t, out, nullptr);
}
// The same step on rows idx[idx_offset .. idx_offset + n) of the resident data: the gather ([UPSTREAM] gpflow.params.Minibatch,
// dgp.py:51-52) happens inside the step's first launch instead of a launch of its own.
extern "C" int dsdgp_model_train_step_minibatch(dsdgp_model* m, const double* X_all, const double* Y_all, const int64_t* idx,
                                                int64_t idx_offset, int64_t n, int32_t S, uint64_t seed, double data_scale, double kl_weight,
                                                double lr, double beta1, double beta2, double eps, int64_t t, double* out) {
  DS_CHECK_ARG(X_all && Y_all && idx && idx_offset >= 0);
  const GatherSrc gs{X_all, Y_all, idx + idx_offset};
  return train_step_impl(m, nullptr, nullptr, n, S, nullptr, nullptr, seed, data_scale, kl_weight, lr, beta1, beta2, eps, t, out, &gs);
}

extern "C" int dsdgp_model_set_bucket_callback(dsdgp_model* m, dsdgp_bucket_fn fn, void* user) {
  DS_CHECK_ARG(m != nullptr);
  if (fn) {
    // the buckets are contiguous SEGMENTS of theta: layer l owns [off_Z_l, off_Z_{l+1}), the likelihood variance follows the last
    // layer's segment.  A descriptor with another ordering would hand out segments that are incomplete or not yet produced.
    const int L = m->desc.L;
    const int64_t lik_lo = lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : m->desc.n_theta;
    for (int l = 0; l < L; ++l) {
      const dsdgp_layer_desc& d = m->L[l].d;
      const int64_t lo = d.off_Z, hi = (l + 1 < L) ? m->L[l + 1].d.off_Z : lik_lo;
      const int64_t offs[] = {d.off_q_mu, d.off_q_sqrt, d.off_kvar, d.off_kls, d.has_white ? d.off_wvar : lo,
                              d.off_mean_A >= 0 ? d.off_mean_A : lo, d.off_mean_b >= 0 ? d.off_mean_b : lo};
      bool ok = lo < hi;
      for (int64_t o : offs) ok = ok && o >= lo && o < hi;
      if (!ok) {
        dsdgp_set_error("dsdgp_model_set_bucket_callback: the parameters of layer %d do not form one contiguous segment of theta "
                        "ordered [layer 0 | layer 1 | ... | likelihood]", l);
        return DSDGP_ERR_UNSUPPORTED;
      }
    }
  }
  m->bucket_fn = fn;
  m->bucket_user = user;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_set_grad_first_layer(dsdgp_model* m, int32_t first) {
  DS_CHECK_ARG(m && first >= 0 && first < m->desc.L);
  m->grad_first = first;
  return DSDGP_OK;
}
extern "C" int dsdgp_model_set_grad_q_only(dsdgp_model* m, int32_t on) {
  DS_CHECK_ARG(m != nullptr);
  m->grad_q_only = on != 0;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_track_theta(dsdgp_model* m, int enable) {
  DS_CHECK_ARG(m != nullptr);
  m->track_theta = enable != 0;
  m->kuu_valid = false;
  m->q_dirty = -2;
  return DSDGP_OK;
}
extern "C" int dsdgp_model_theta_changed(dsdgp_model* m) {
  DS_CHECK_ARG(m != nullptr);
  m->prepared = false;
  m->kuu_valid = false;
  m->q_dirty = -2;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_layer_kl(dsdgp_model* m, int32_t l, double* out) {
  DS_CHECK_ARG(m && out && l >= 0 && l < m->desc.L);
  if (!m->prepared) DS_TRY(prepare_async(m));
  DS_LAUNCH(k_kl_final, dim3(1), dim3(256), 0, m->ctx->stream, m->layers_dev, m->desc.L);
  DS_HIP(hipGetLastError());
  DS_HIP(hipMemcpyAsync(out, m->L[l].dev.klv, sizeof(double), hipMemcpyDeviceToDevice, m->ctx->stream));
  return DSDGP_OK;
}

extern "C" int dsdgp_model_layer_conditional(dsdgp_model* m, int32_t l, const double* X, int64_t n, double* mean,
                                             double* var) {
  DS_CHECK_ARG(m && X && mean && var && l >= 0 && l < m->desc.L && n > 0);
  if (!m->prepared) DS_TRY(prepare_async(m));
  LayerState& St = m->L[l];
  const LayerDev& v = St.dev;
  LayerFwdArgs a{};
  a.X = X; a.Rin = n; a.rep = 1;
  a.D_in = v.D_in; a.D_out = v.D_out; a.M = v.M;
  a.Zp = v.Zp; a.Zs = v.Zs; a.hyp = v.hyp; a.LinvT = v.LinvT; a.Linv = v.Linv; a.Tp = v.Tp; a.TpT = v.TpT; a.qmu = v.qmu;
  a.mean_kind = St.d.mean_kind; a.mean_A = St.meanA; a.mean_b = St.meanb;
  a.jitter = m->desc.jitter;
  a.n_inner = n;
  a.mean = mean; a.var = var;
  a.ldA = round_up(n, 16);
  if (St.gemm) {
    // the GEMM-formulated pass works in the model's scratch (gws: sized for ld_max rows): more rows than that go through in chunks —
    // rows are independent (layers.py:71-74), mean / var are row-major, and above Mp = 1024 no chain instance exists to fall back on
    const int64_t chunk = St.ld_max - St.ld_max % 16;
    DS_CHECK_ARG(chunk > 0);
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      LayerFwdArgs c = a;
      c.Rin = std::min<int64_t>(chunk, n - r0);
      c.n_inner = c.Rin;
      c.X = X + r0 * v.D_in;
      c.mean = mean + r0 * v.D_out;
      c.var = var + r0 * v.D_out;
      c.ldA = round_up(c.Rin, 16);
      DS_TRY(layer_fwd_gemm_launch(m->ctx, c, v.Mp, v.kern_kind, m->desc.white ? 1 : 0, m->gws));
    }
    return DSDGP_OK;
  }
  return layer_fwd_sm_launch(m->ctx, a, v.Mp, v.kern_kind, m->desc.white);
}

extern "C" int dsdgp_reparameterize(dsdgp_ctx* ctx, const double* mean, const double* var, const double* z, double jitter,
                                    int64_t count, double* out) {
  DS_CHECK_ARG(ctx && mean && var && z && out && count > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(count, 256));
  DS_LAUNCH(k_reparam, dim3(nb), dim3(256), 0, ctx->stream, mean, var, z, jitter, count, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

extern "C" int dsdgp_randn(dsdgp_ctx* ctx, uint64_t seed, uint64_t stream, int64_t count, double* out) {
  DS_CHECK_ARG(ctx && out && count > 0);
  return randn_async(ctx, seed, stream, count, out);
}

#include "model_extras.hpp"     // natural gradients, full covariance
