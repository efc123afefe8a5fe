// Workspace layout of a model (dsdgp_model_workspace_bytes / dsdgp_model_create): every device buffer of the path carved from the
// caller's allocation by one bump allocator, so that sizing (base == NULL) and carving are the same code.  Part of the model translation unit.
#pragma once
static void layout(dsdgp_model* m, char* base, size_t* total) {
  Bump b{base, 0};
  const dsdgp_model_desc& D = m->desc;
  m->layers_dev = b.take<LayerDev>(D.L);
  m->mask = b.take<double>(D.n_theta);
  m->lik_const = b.take<double>(8);
  m->scal4 = b.take<double>(8);
  const int64_t Rlast = (int64_t)m->s_max * m->n_max;
  m->lik_blocks_max = std::max(ceil_div((Rlast + 16) * D.layers[D.L - 1].D_out, 256) + 1,   // + the 16-row padding written by the fused adjoint path
                               4 * ceil_div(Rlast, 16) + 4);                                 // likelihood inside the last chain: one pair per workgroup
  m->lik_part = b.take<double>((size_t)m->lik_blocks_max * 2);
  m->Xmb = b.take<double>((size_t)m->n_max * D.layers[0].D_in);
  m->Ymb = b.take<double>((size_t)m->n_max * D.layers[D.L - 1].D_out);
  m->lik_dmean = b.take<double>((size_t)Rlast * D.layers[D.L - 1].D_out);
  m->lik_dvar = b.take<double>((size_t)Rlast * D.layers[D.L - 1].D_out);
  m->potrf_items = b.take<PotrfItem>(D.L);
  m->gp_fwd = b.take<GemmProblem>(4 * D.L);
  m->gp_bwd1 = b.take<GemmProblem>(4 * D.L);
  m->gp_pt = b.take<GemmProblem>(2 * D.L);
  m->gp_bwd2 = b.take<GemmProblem>(D.L);
  m->gp_wz = b.take<GemmProblem>(D.L);
  m->gp_w1 = b.take<GemmProblem>(2 * D.L); m->gp_w2 = b.take<GemmProblem>(D.L); m->gp_w3 = b.take<GemmProblem>(D.L);
  m->rjobs_cap = 0;
  for (int l = 0; l < D.L; ++l) m->rjobs_cap += D.layers[l].D_out + 5;
  m->rjobs = b.take<RedJob>(m->rjobs_cap);
  // layers with the same (large) M keep their Ku / Lu^-1 / Lu^-T contiguous so that one batched factorisation serves them all
  bool uniform = D.L > 1 && pad_M(D.layers[0].M) >= big_mp(true);
  for (int l = 1; l < D.L; ++l) uniform = uniform && D.layers[l].M == D.layers[0].M;
  uniform = uniform && pad_M(D.layers[0].M) % 64 == 0;
  m->uniform_big = uniform;
  double *Kp_all = nullptr, *Linv_all = nullptr, *LinvT_all = nullptr, *scal_all = nullptr;
  if (uniform) {
    const size_t MM0 = (size_t)pad_M(D.layers[0].M) * pad_M(D.layers[0].M);
    Kp_all = b.take<double>(D.L * MM0); Linv_all = b.take<double>(D.L * MM0); LinvT_all = b.take<double>(D.L * MM0);
    scal_all = b.take<double>(D.L * 8);
  }
  for (int l = 0; l < D.L; ++l) {
    LayerState& S = m->L[l];
    const dsdgp_layer_desc& d = D.layers[l];
    S.d = d;
    LayerDev& v = S.dev;
    v.M = d.M; v.Mp = pad_M(d.M); v.D_in = d.D_in; v.D_out = d.D_out;
    v.DP4 = (int)round_up(d.D_out, 4); v.DP16 = (int)round_up(d.D_out, 16); v.DinP16 = (int)round_up(d.D_in + 1, 16);
    v.kern_kind = d.kern_kind; v.ard = d.ard; v.has_white = d.has_white; v.white = D.white; v.hyp_parts = -NPART;
    v.off_Z = d.off_Z; v.off_q_mu = d.off_q_mu; v.off_q_sqrt = d.off_q_sqrt;
    v.off_kvar = d.off_kvar; v.off_kls = d.off_kls; v.off_wvar = d.off_wvar; v.kvar_identity = d.kvar_identity ? 1 : 0;
    const size_t Mp = v.Mp, MM = Mp * Mp;
    v.Zp = b.take<double>(Mp * d.D_in);
    v.Zs = b.take<double>(Mp * d.D_in);
    v.hyp = b.take<double>(HYP_ILS + 2 * d.D_in + 8);
    v.Tp = b.take<double>(d.D_out * MM);
    v.TpT = b.take<double>(d.D_out * MM);
    v.qmu = b.take<double>(Mp * d.D_out);
    v.qmu4 = b.take<double>(Mp * v.DP4);
    if (uniform) {
      v.Kp = Kp_all + l * MM; v.Linv = Linv_all + l * MM; v.LinvT = LinvT_all + l * MM; v.scal = scal_all + l * 8;
    } else {
      v.Kp = b.take<double>(MM); v.Linv = b.take<double>(MM); v.LinvT = b.take<double>(MM); v.scal = b.take<double>(16);
    }
    v.Kinv = b.take<double>(MM);
    v.V = b.take<double>(d.D_out * MM); v.nL = b.take<double>(Mp * v.DP4); v.Sd = b.take<double>(d.D_out * MM);
    v.klv = b.take<double>(8);
    v.U = b.take<double>(d.D_out * MM); v.n4 = b.take<double>(Mp * v.DP4); v.PT = b.take<double>(d.D_out * MM);
    v.UU = b.take<double>(d.D_out * MM); v.Kbar = b.take<double>(MM); v.wm = b.take<double>(MM); v.wk = b.take<double>(MM);
    v.bigred = b.take<double>((1 + d.D_out) * MM);
    v.thinq = b.take<double>(Mp * v.DP16);
    v.thinz = b.take<double>(Mp * v.DinP16);
    v.hyp_red = b.take<double>(d.D_in + 2 + 8);
    v.klpart = b.take<double>(512);
    v.ngTI = b.take<double>(d.D_out * MM); v.ngTinv = b.take<double>(d.D_out * MM); v.ngTbar = b.take<double>(d.D_out * MM);
    v.ngH = b.take<double>(d.D_out * MM); v.ngY = b.take<double>(d.D_out * MM); v.ngX = b.take<double>(d.D_out * MM);
    v.ngSinv = b.take<double>(d.D_out * MM); v.ngA = b.take<double>(d.D_out * MM); v.ngLAinv = b.take<double>(d.D_out * MM);
    v.ngLAinvT = b.take<double>(d.D_out * MM); v.ngV = b.take<double>(d.D_out * Mp);
    v.ngTheta1 = b.take<double>(d.D_out * Mp); v.ngScal = b.take<double>(4 * d.D_out + 8);
    v.wLbar = b.take<double>(MM); v.wH = b.take<double>(MM); v.wY = b.take<double>(MM); v.wX = b.take<double>(MM);
    v.hyp2part = b.take<double>(1024 * (d.D_in + 2));
    {
      const int alg_env = m->force.alg_g;   // -1: heuristic, 0: never, 1: always
      const int64_t R_l = (l == 0) ? m->n_max : (int64_t)m->s_max * m->n_max;
      v.alg_g = (!D.white && (alg_env == 1 || (alg_env < 0 && (int64_t)4 * d.D_out * v.Mp <= R_l))) ? 1 : 0;
      v.need_tpt = (v.Mp > 256 || save_c_enabled(m, v.Mp) || (m->force.gemm_mp > 0 && v.Mp >= m->force.gemm_mp)) ? 1 : 0;
      // the fused last layer (layer_last.hip) forms abar's variance part as q_sqrt (q_sqrt^T a): it reads q_sqrt^T as rows
      if (l == D.L - 1 && D.L >= 2 && !D.white && m->force.last_fuse != 0 && layer_last_built(v.Mp, d.D_in, d.D_out)) v.need_tpt = 1;
      v.KS = v.alg_g ? b.take<double>(d.D_out * MM) : nullptr;
      v.GS = v.alg_g ? b.take<double>(d.D_out * MM) : nullptr;
    }
    v.off_mean_A = (d.mean_kind == DSDGP_MEAN_LINEAR) ? d.off_mean_A : -1;
    v.off_mean_b = (d.mean_kind == DSDGP_MEAN_LINEAR) ? d.off_mean_b : -1;
    S.mean_grad = (v.off_mean_A >= 0 && d.trainable_mean_A) || (v.off_mean_b >= 0 && d.trainable_mean_b);
    const int mrows = 64 * ceil_div(v.DinP16, 64);
    v.meanAB = S.mean_grad ? b.take<double>((size_t)mrows * v.DP16) : nullptr;
    v.R2 = b.take<double>(MM);
    v.Zp1 = b.take<double>(Mp * v.DinP16); v.WZ = b.take<double>(Mp * v.DinP16);
    S.R_max = (int64_t)m->s_max * m->n_max;
    const int64_t Rin_max = (l == 0) ? m->n_max : S.R_max;
    S.ld_max = round_up(Rin_max, 16);
    const size_t Mw = pad_Mw(v.Mp);      // rows Mp..Mw-1 stay zero (never written): whole tiles for the weight-gradient products
    S.A = b.take<double>(Mw * S.ld_max); S.E = b.take<double>(Mw * S.ld_max); S.GW = b.take<double>(Mw * S.ld_max);
    S.C = save_c_enabled(m, (int)Mp) ? b.take<double>((size_t)d.D_out * Mp * S.ld_max) : nullptr;
    S.VB = b.take<double>(v.DP16 * S.ld_max); S.MB = b.take<double>(v.DP16 * S.ld_max);
    S.XT1 = b.take<double>((size_t)round_up(v.DinP16, 64) * S.ld_max);   // rows >= DinP16 stay zero: whole 64-row tiles for the mean-gradient product
    S.F = b.take<double>(S.R_max * d.D_out); S.mean = b.take<double>(S.R_max * d.D_out);
    S.var = b.take<double>(S.R_max * d.D_out); S.zbuf = b.take<double>(S.R_max * d.D_out + 2);
    S.prop = (l + 1 < D.L) ? d.input_prop_dim : 0;     // the last layer's concatenation is host glue (nothing consumes it)
    S.dF = b.take<double>(S.R_max * (d.D_out + S.prop));
    S.Xcat = S.prop ? b.take<double>(S.R_max * (d.D_out + S.prop)) : nullptr;
    int NI, ti;
    wgrad_shapes(v.Mp, NI, ti);
    const int tj_big = ti;
    S.nsplit_big_max = choose_nsplit((v.alg_g ? 0 : ti * tj_big) + d.D_out * (ti * (ti - 1) / 2) + (int)ceil(d.D_out * ti * (NI + 1) / (2.0 * NI)),
                                     S.ld_max / 16, 512);
    S.nsplit_thin_max = choose_nsplit(ti * (v.DP16 / 16 + v.DinP16 / 16), S.ld_max / 16, 256);
    S.part_big = b.take<double>((size_t)S.nsplit_big_max * (1 + d.D_out) * Mw * Mw);
    S.part_thin = b.take<double>((size_t)S.nsplit_big_max * Mw * (v.DP16 + v.DinP16));
    S.part_mean = S.mean_grad ? b.take<double>((size_t)S.nsplit_big_max * mrows * v.DP16) : nullptr;
    S.gemm = m->force.gemm_mp > 0 && v.Mp >= m->force.gemm_mp;
    S.hyp_part = b.take<double>((size_t)(std::max<int64_t>(std::max<int64_t>(sm_hyp_parts(S.ld_max, v.Mp, d.D_in), layer_gemm_hyp_parts(S.ld_max, v.Mp)),
                                                           8 * 160) + 16) * (d.D_in + 2));
    // d-split of the backward chain on small launches (at most 1024 workgroups): partial abar tiles + arrival counters
    S.bpart = b.take<double>((size_t)1024 * (Mp * 16 + 16));
    S.bcnt = b.take<int>(512);
    S.lq = b.take<GemmProblem>(12);
    S.wj = b.take<WgradJob>(d.D_out + 4);
    {
      const int tq = (int)ceil_div(v.DP16 / 16, 4), tz = (int)ceil_div(v.DinP16 / 16, 4);
      S.wtick_cap = (d.D_out + 1) * ti * ti + ti * (tq + tz) + tz * tq + 16;
      S.wtick = b.take<int32_t>(S.wtick_cap);
    }
    S.ng_gp = b.take<GemmProblem>(4);
    S.ng_items = b.take<PotrfItem>(d.D_out);
  }
  {   // tile lists of the grouped M x M launches: every problem list is planned at most three times (whole model, per layer, natural
      // gradients), a problem has at most D_out x (Mw / 64)^2 tiles
    int64_t cap = 0;
    for (int l = 0; l < D.L; ++l) {
      const int64_t t64 = ceil_div(pad_Mw(m->L[l].dev.Mp), 64);
      cap += (int64_t)(20 * D.layers[l].D_out + 24) * t64 * t64;
    }
    m->gemm_order_cap = 2 * 2 * cap;
    m->gemm_order = b.take<int32_t>((size_t)m->gemm_order_cap);
  }
  {   // scratch of the GEMM-formulated layers, sized for the largest of them
    int64_t ML = 0, cq = 0, mut = 0, qt = 0, zz = 0, ot = 0, sv = 0;
    for (int l = 0; l < D.L; ++l) {
      const LayerState& S = m->L[l];
      if (!S.gemm) continue;
      const LayerDev& v = S.dev;
      const int64_t ld = S.ld_max, nzz16 = round_up(2 * v.D_in + 1, 16);
      ML = std::max<int64_t>(ML, (int64_t)v.Mp * ld);
      cq = std::max<int64_t>(cq, (int64_t)(1 + v.D_out) * ceil_div(v.Mp, 128) * ld);
      mut = std::max<int64_t>(mut, (int64_t)v.DP16 * ld);
      qt = std::max<int64_t>(qt, (int64_t)v.DP16 * v.Mp);
      zz = std::max<int64_t>(zz, nzz16 * v.Mp);
      ot = std::max<int64_t>(ot, nzz16 * ld);
      sv = std::max<int64_t>(sv, (int64_t)ceil_div(ld, 32) * ceil_div(v.Mp, 32));
    }
    if (ML > 0) {
      m->gws.T1 = b.take<double>(ML); m->gws.T2 = b.take<double>(ML); m->gws.Pb = b.take<double>((GL_MAX_GROUPS + 1) * ML);
      m->gws.pb_doubles = (GL_MAX_GROUPS + 1) * ML;
      m->gws.colsq = b.take<double>(cq); m->gws.MUT = b.take<double>(mut); m->gws.qmuT = b.take<double>(qt);
      m->gws.ZZ = b.take<double>(zz); m->gws.OUTt = b.take<double>(ot); m->gws.svar = b.take<double>(sv + 16);
    }
  }
  *total = (size_t)round_up((int64_t)b.off, 256);
}
