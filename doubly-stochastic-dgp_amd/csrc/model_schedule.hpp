// Launch schedule of one evaluation: prepare (Ku, factor, parameter products; main / side stream), forward layers (dgp.py:61-76), plan of
// the weight-gradient products and reductions, reverse pass with the stream overlap and the data-parallel buckets, ELBO / training step.
// Part of the model translation unit.
#pragma once
// Side-stream overlap pays for its cross-stream events (a few microseconds each) only when the kernels are long enough:
// tiny models (cfg 1: 100 rows, M = 50: n S Mp = 6400) are launch-latency-bound and run 40 % faster on a single stream.  The threshold
// (DSDGP_FORCE overlap_min, default 2^18; 2^20 until round 6) sits below the 125- and 250-row shards of config 2's strong-scaling run
// (n S Mp = 3.2e5 / 6.4e5): overlapped they take 0.297 / 0.342 ms per data-parallel step against 0.317 / 0.368 on one stream.
static bool overlap_on(const dsdgp_model* m, int64_t n, int S) {
  const char* no = getenv("DSDGP_NO_OVERLAP");   // read per call so that a profiler can serialise the kernels
  if (!m->overlap || (no && atoi(no))) return false;
  int mp_max = 0;
  for (int l = 0; l < m->desc.L; ++l) mp_max = std::max(mp_max, (int)m->L[l].dev.Mp);
  return n * S * (int64_t)mp_max >= (int64_t)m->force.overlap_min;
}

// main stream waits for the parameter-only side work of the last prepare (no-op when nothing is pending)
static int join_prep(dsdgp_model* m) {
  if (m->side_pending) {
    DS_HIP(hipStreamWaitEvent(m->ctx->stream, m->ev_prep_side, 0));
    m->side_pending = false;
  }
  return DSDGP_OK;
}

// Parameter transforms, Ku, its Cholesky / inverse factor (main stream: the forward chain needs exactly these), then the
// parameter-only rest — Ku^-1, S_d, Lu^-1 q_sqrt, KL and, for a gradient step, U_d = Ku^-1 q_sqrt_d, n = Ku^-1 q_mu and
// U_d U_d^T — which nothing needs before the backward pass / the final reduction: with `side` it runs on the side stream
// concurrently with the forward layers and the caller joins (join_prep) where it is first consumed.
// (`side` = run that part on the side stream.)
static int prepare_async(dsdgp_model* m, bool with_grad = false, bool side = false, const HeadRand* hr = nullptr,
                         const HeadGather* hg = nullptr) {
  dsdgp_ctx* ctx = m->ctx;
  const int L = m->desc.L;
  DS_TRY(join_prep(m));
  const bool keep_kuu = m->track_theta && m->kuu_valid;
  // dsdgp_model_track_theta and no change since an evaluation that produced everything this one needs: the factor AND the
  // parameter-side products (Ku^-1, Lu^-1 q_sqrt, KL, ...) stay — a forward-only evaluation at fixed parameters is the chains alone
  const bool unchanged = keep_kuu && m->q_dirty == -1 && (m->prepared_grad || !with_grad);
  int mp_max = 0;
  for (int l = 0; l < L; ++l) mp_max = std::max(mp_max, (int)m->L[l].dev.Mp);
  bool head_event = false;
  if (m->head_ok) {
    ProfScope ps(ctx, "potrf");
    const size_t lds = head_lds_bytes(mp_max);
    static size_t lds_set = 0;     // the attribute is sticky: one driver call per size
    if (lds > lds_set) {
      DS_HIP(hipFuncSetAttribute((const void*)k_head, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_set = lds;
    }
    HeadRand none{};
    const HeadRand& r = hr ? *hr : none;
    HeadGather gnone{};
    const HeadGather& gq = hg ? *hg : gnone;
    const int nprep = std::max(32, m->prep_blocks / 2);
    // with a side stream the launch carries the fork event itself (hipExtLaunchKernel attaches it to the dispatch's completion signal):
    // a separate hipEventRecord puts a marker packet on this stream that the next kernel queues behind (~6 us, profiles/r03_timeline_*)
    head_event = side && m->force.ext_ev != 0;
    g_dsdgp_launches.fetch_add(1, std::memory_order_relaxed);
    hipExtLaunchKernelGGL(k_head, dim3(1 + nprep + r.nblk + gq.nblk, L), dim3(HEAD_THREADS), (uint32_t)lds, ctx->stream, nullptr,
                          head_event ? m->ev_fork : nullptr, 0, (const double*)m->theta, (const LayerDev*)m->layers_dev, m->lik_const,
                          (int64_t)m->desc.off_lik_var, lik_has_param(m->desc.lik_kind) ? 1 : 0, m->desc.jitter, nprep,
                          keep_kuu ? 1 : 0, m->desc.white ? 1 : 0, getenv("DSDGP_POTRF_TIMING") ? 1 : 0, r, gq);
    DS_HIP(hipGetLastError());
  } else if (!unchanged) {
    DS_LAUNCH(k_prep_kuu, dim3(m->prep_blocks + (keep_kuu ? 0 : m->kuu_blocks), L), dim3(256), 0, ctx->stream, m->theta, m->layers_dev,
                       m->lik_const, m->desc.off_lik_var, lik_has_param(m->desc.lik_kind) ? 1 : 0, m->desc.jitter,
                       m->prep_blocks, keep_kuu ? 1 : 0);
    DS_HIP(hipGetLastError());
  }
  if (m->head_ok) {
    // (factorised inside k_head)
  } else if (keep_kuu) {
    // Z and the kernel hyper-parameters are those of the previous evaluation: Lu, Lu^-1, log det stay
  } else if (m->uniform_big) {
    DS_TRY(bigchol_run(ctx, m->big_all));
  } else if (mp_max >= big_mp(false)) {
    for (int l = 0; l < L; ++l) {
      if (m->L[l].big) DS_TRY(bigchol_run(ctx, m->L[l].big_k));
      else DS_TRY(potrf_launch(ctx, m->potrf_items + l, 1, m->L[l].dev.Mp));
    }
  } else {
    DS_TRY(potrf_launch(ctx, m->potrf_items, L, mp_max));
  }
  if (unchanged) {
    m->prepared = true;
    return DSDGP_OK;
  }
  hipStream_t st = ctx->stream;
  if (side) {
    if (!head_event) DS_HIP(hipEventRecord(m->ev_fork, ctx->stream));
    DS_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    st = m->side;
  }
  const int klb = mp_max >= 512 ? 512 : NPART;    // M = 512 / 1024: V alone is 8..64 MB per layer — 32 workgroups were latency-bound
  // only layer lq's (q_mu, q_sqrt) moved since an evaluation that produced everything this one needs: its products alone
  const int gfirst = (with_grad && !m->desc.white) ? m->grad_first : 0;
  // The forward-side products (Ku^-1, Lu^-1 q_sqrt, Lu^-1 q_mu, S_d, the KL sums) and the gradient-side ones (U_d, n, U_d U_d^T) are
  // decided separately: after a natural-gradient step on layer lf the forward side of every other layer is still that of the previous
  // evaluation, whatever that evaluation did on the gradient side (a pruned natural-gradient evaluation leaves the lower layers' U_d
  // undone: the Adam evaluation that follows redoes the gradient side of ALL layers, but the forward side of layer lf only — it used to
  // redo all 17 Lu^-1 q_sqrt_d of a config-5 model as well)
  const int lf = (keep_kuu && m->q_dirty >= 0) ? m->q_dirty : -1;
  const int lq = (lf >= 0 && (m->prepared_grad || !with_grad)) ? lf : -1;
  if (keep_kuu && m->q_dirty == -1) {
    // nothing moved since the last evaluation (it lacked the gradient side only): the forward side stands as it is
  } else if (lf >= 0) {
    LayerState& Sq = m->L[lf];
    DS_TRY(gemm_launch(ctx, Sq.lq, Sq.lq_nf, Sq.lq_tf, st));
    DS_LAUNCH(k_kl_part, dim3(klb, 1), dim3(256), 0, st, m->layers_dev + lf);
  } else {
    DS_TRY(gemm_launch(ctx, m->gp_fwd, m->n_fwd, m->t_fwd, st));
    DS_LAUNCH(k_kl_part, dim3(klb, L), dim3(256), 0, st, m->layers_dev);
  }
  DS_HIP(hipGetLastError());
  m->kinv_pending = false;
  if (side && with_grad && !m->desc.white) {
    // Ku^-1, S_d and Lu^-1 q_sqrt are in the launch above — everything the backward CHAINS read of the side stream's products (the
    // rest — KS_d, U_d, n, U_d U_d^T — feeds the assembly at the end of the step).  The reverse pass waits for THIS event, long since
    // signalled when the forward chains end; the whole side stream's work ends together with the last inner forward chain, and a
    // just-in-time cross-stream wait there cost ~12 us of idle main stream in front of the first launch of the reverse pass
    // (profiles/r06_timeline_step.txt).  The rest is covered by the join at the end of the reverse pass (the side stream is in order).
    DS_HIP(hipEventRecord(m->ev_kinv, m->side));
    m->kinv_pending = true;
  }
  if (with_grad && !m->desc.white) {
    if (lq >= 0) {
      LayerState& Sq = m->L[lq];
      DS_TRY(gemm_launch(ctx, Sq.lq + Sq.lq_nf, Sq.lq_n1, Sq.lq_t1, st));
      DS_TRY(gemm_launch(ctx, Sq.lq + Sq.lq_nf + Sq.lq_n1, Sq.lq_n2, Sq.lq_t2, st));
    } else if (gfirst > 0) {
      for (int l = gfirst; l < L; ++l) {
        LayerState& Sq = m->L[l];
        DS_TRY(gemm_launch(ctx, Sq.lq + Sq.lq_nf, Sq.lq_n1, Sq.lq_t1, st));
        DS_TRY(gemm_launch(ctx, Sq.lq + Sq.lq_nf + Sq.lq_n1, Sq.lq_n2, Sq.lq_t2, st));
      }
    } else {
      DS_TRY(gemm_launch(ctx, m->gp_bwd1, m->n_bwd1, m->t_bwd1, st));
      DS_TRY(gemm_launch(ctx, m->gp_bwd2, m->n_bwd2, m->t_bwd2, st));
    }
  }
  if (side) {
    DS_HIP(hipEventRecord(m->ev_prep_side, m->side));
    m->side_pending = true;
  }
  m->prepared = true;
  m->prepared_grad = with_grad && gfirst == 0;     // (a partial prepare with_grad required the previous one to have had it)
  m->kuu_valid = true;
  m->q_dirty = -1;
  return DSDGP_OK;
}

static int read_info(dsdgp_model* m, int* info) {
  if (!info) return DSDGP_OK;
  *info = 0;
  if (getenv("DSDGP_POTRF_TIMING")) {   // debug aid: per-phase shader cycles of layer 0's factorisation
    double sc[12];
    hipMemcpyAsync(sc, m->L[0].dev.scal, sizeof(sc), hipMemcpyDeviceToHost, m->ctx->stream);
    hipStreamSynchronize(m->ctx->stream);
    if (m->head_ok)
      fprintf(stderr, "[head cycles] start + Z staging %.0f | Ku %.0f | first panel %.0f | block columns 1.. (+ inverse rows) %.0f | logdet + last two "
              "inverse rows %.0f || wave 0 in the loop: tile %.0f, barrier %.0f, panel %.0f, barrier %.0f\n", sc[2], sc[3], sc[4], sc[5], sc[6], sc[7],
              sc[8], sc[9], sc[10]);
    else
      fprintf(stderr, "[potrf cycles] factor %.0f inverse %.0f panel %.0f trailing %.0f logdet+writeback %.0f copyin+trtri %.0f\n", sc[2],
              sc[3], sc[4], sc[5], sc[6], sc[7]);
  }
  for (int l = 0; l < m->desc.L; ++l) {
    double sc[2];
    DS_HIP(hipMemcpyAsync(sc, m->L[l].dev.scal, sizeof(sc), hipMemcpyDeviceToHost, m->ctx->stream));
    DS_HIP(hipStreamSynchronize(m->ctx->stream));
    if (sc[1] != 0.0 && *info == 0) *info = (int)sc[1];
  }
  if (*info) {
    dsdgp_set_error("Cholesky decomposition was not successful (layer Kuu pivot %d)", *info);
    return DSDGP_ERR_NOT_SPD;
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_model_prepare(dsdgp_model* m, int* info) {
  DS_CHECK_ARG(m != nullptr);
  DS_TRY(prepare_async(m));
  return read_info(m, info);
}

static int randn_async(dsdgp_ctx* ctx, uint64_t seed, uint64_t stream, int64_t count, double* out, hipStream_t st = nullptr) {
  const int nb = (int)std::min<int64_t>(2048, ceil_div((count + 1) / 2, 256));
  DS_LAUNCH(k_randn, dim3(nb > 0 ? nb : 1), dim3(256), 0, st ? st : ctx->stream, seed, stream, count, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// Whether the last layer's forward chain, the Gaussian likelihood and its reverse pass run as ONE launch (layer_last.hip) in this step.
// `a`: the forward arguments as forward_layers built them.
static bool last_chain_fusable(const dsdgp_model* m, const LayerState& St, const LayerFwdArgs& a) {
  const LayerDev& v = St.dev;
  const int L = m->desc.L;
  if (m->force.last_fuse == 0 || m->desc.white || St.gemm || L < 2) return false;
  if (!layer_last_built(v.Mp, v.D_in, v.D_out) || !v.alg_g || !v.need_tpt) return false;
  if (!a.lik_Y || !a.Asave || a.Csave || a.F || a.rep != 1 || a.d_split > 1 || a.qmu_ld) return false;
  if (a.mean_kind != DSDGP_MEAN_ZERO) return false;
  if (m->grad_q_only && m->grad_first == L - 1) return false;     // that layer runs no backward chain at all
  if (m->force.bwd_split >= 2) return false;
  static const bool timing = getenv("DSDGP_FWD_TIMING") || getenv("DSDGP_BWD_TIMING");      // the phase clocks are the two chains'
  if (timing) return false;
  return ceil_div(a.Rin, 16) >= m->force.last_min_blocks;
}

// dgp.py:61-76 propagate
static int forward_layers(dsdgp_model* m, const double* X, int64_t n, int S, const double* const* zs,
                          const int64_t* zstride, uint64_t seed, bool save, bool need_last_F, double* const* Fs,
                          double* const* Fmeans, double* const* Fvars, bool z_ready = false, const double* lik_Y = nullptr,
                          double lik_w = 0.0, int* lik_nblocks = nullptr) {
  dsdgp_ctx* ctx = m->ctx;
  const int L = m->desc.L;
  DS_CHECK_ARG(n > 0 && n <= m->n_max && S > 0 && S <= m->s_max);
  const double* Xin = X;
  // Forward-only evaluations of a non-white model (predict_f, ELBO values) run in WHITENED coordinates: with V_d = Lu^-1 q_sqrt_d and
  // nL = Lu^-1 q_mu — both formed for the KL term anyway — mean = a1^T nL and var = kdiag - |a1|^2 + |V_d^T a1|^2 (V_d is
  // lower-triangular like q_sqrt_d), so the chain skips a = Lu^-T a1 (layers.py:188): one of 2 + D_out triangular products per row
  // block, a third of the MFMA work of a D_out = 1 layer.  The training pass keeps `a` (the reverse pass is written in terms of it).
  // Mp <= 256: the larger instances read the factor transposed, which exists for q_sqrt only.
  const bool wf_ok = !m->desc.white && !save && m->force.white_fwd != 0;
  m->last_deferred = false;
  if (wf_ok) DS_TRY(join_prep(m));          // V, nL come from the parameter products (side stream in the overlapped schedule)
  for (int l = 0; l < L; ++l) {
    LayerState& St = m->L[l];
    const LayerDev& v = St.dev;
    const int64_t Rin = (l == 0) ? n : (int64_t)S * n;
    const int rep = (l == 0) ? S : 1;
    const bool last = (l == L - 1);
    const bool want_F = !last || need_last_F;
    const bool wf = wf_ok && v.Mp <= 256 && !St.gemm;
    LayerFwdArgs a{};
    a.X = Xin; a.Rin = Rin; a.rep = rep;
    a.D_in = v.D_in; a.D_out = v.D_out; a.M = v.M;
    a.Zp = v.Zp; a.Zs = v.Zs; a.hyp = v.hyp; a.LinvT = v.LinvT; a.Linv = v.Linv; a.Tp = v.Tp; a.TpT = v.TpT; a.qmu = v.qmu;
    if (wf) { a.Tp = v.V; a.qmu = v.nL; a.qmu_ld = v.DP4; }
    a.mean_kind = St.d.mean_kind; a.mean_A = St.meanA; a.mean_b = St.meanb;
    a.jitter = m->desc.jitter;
    a.n_inner = n;
    a.z = nullptr;
    if (want_F) {
      if (zs && zs[l]) {
        a.z = zs[l];
        a.zs_s = zstride[3 * l]; a.zs_n = zstride[3 * l + 1]; a.zs_d = zstride[3 * l + 2];
      } else {
        if (!(z_ready && !last)) DS_TRY(randn_async(ctx, seed, (uint64_t)l, (int64_t)S * n * v.D_out, St.zbuf));
        a.z = St.zbuf;
        a.zs_s = n * v.D_out; a.zs_n = v.D_out; a.zs_d = 1;
      }
    }
    a.F = want_F ? ((Fs && Fs[l]) ? Fs[l] : St.F) : nullptr;
    a.mean = (Fmeans && Fmeans[l]) ? Fmeans[l] : St.mean;
    a.var = (Fvars && Fvars[l]) ? Fvars[l] : St.var;
    a.ldA = round_up(Rin, 16);
    const bool save_l = save && (m->desc.white || l >= m->grad_first);    // layers below the pruned reverse pass keep nothing for it
    a.Asave = save_l ? St.A : nullptr;
    // c_d is kept for the backward chain where that pays: enough row blocks to hide the extra latency per output (the N-row first
    // layer is a latency-bound launch) and enough outputs for the halved d-loop to matter
    // (from Mp = 512 one output's product outlasts the staging latency even on a handful of row blocks: always)
    St.c_used = save_l && St.C && sm_cs_built(v.Mp) &&
                (v.Mp > 256 || ((Rin + 15) / 16 > m->force.cs_min_blocks && v.D_out >= m->force.cs_min_dout && v.D_out <= m->force.cs_max_dout));
    if (St.gemm) St.c_used = save_l && St.C != nullptr;      // the triangular abar product halves the largest GEMM of the reverse pass
    // (q, q_sqrt)-only gradients: the lowest layer of the reverse pass runs no backward chain (backward_layers), so it keeps neither
    const bool q_lowest = save_l && m->grad_q_only && !m->desc.white && l == m->grad_first;
    if (q_lowest) St.c_used = false;
    a.Csave = St.c_used ? St.C : nullptr;
    a.XT1 = (save_l && !q_lowest) ? St.XT1 : nullptr;
    {
      const int64_t nblk = (Rin + 15) / 16;
      a.d_split = chain_d_split(nblk, v.D_out);
      if (last && lik_Y) {      // Gaussian variational expectations + adjoints in this chain's epilogue
        a.lik_Y = lik_Y; a.lik_const = m->lik_const; a.lik_w = lik_w; a.lik_part = m->lik_part;
        a.lik_MB = St.MB; a.lik_VB = St.VB; a.lik_ld = round_up(Rin, 16);
        *lik_nblocks = St.gemm ? layer_gemm_lik_blocks(Rin, v.D_out) : (int)nblk * a.d_split;
      }
    }
    if (last && last_chain_fusable(m, St, a)) {
      m->last_fwd = a;               // launched by backward_layers together with this layer's reverse pass
      m->last_deferred = true;
    } else if (St.gemm) DS_TRY(layer_fwd_gemm_launch(ctx, a, v.Mp, v.kern_kind, m->desc.white ? 1 : 0, m->gws));
    else DS_TRY(layer_fwd_sm_launch(ctx, a, v.Mp, v.kern_kind, m->desc.white || wf));
    St.z_used = a.z; St.zs_s = a.zs_s; St.zs_n = a.zs_n; St.zs_d = a.zs_d;
    St.X_used = Xin; St.Rin_used = Rin; St.rep_used = rep; St.ld_used = a.ldA;
    if (St.prop && !last) {
      const int64_t R = (int64_t)S * n, cnt = R * (v.D_out + St.prop);
      DS_LAUNCH(k_concat_prop, dim3((int)std::min<int64_t>(4096, ceil_div(cnt, 256))), dim3(256), 0, ctx->stream, Xin, Rin,
                         v.D_in, St.prop, a.F, v.D_out, R, St.Xcat);
      DS_HIP(hipGetLastError());
      Xin = St.Xcat;
    } else {
      Xin = a.F;
    }
  }
  return DSDGP_OK;
}

extern "C" int dsdgp_model_propagate(dsdgp_model* m, const double* X, int64_t n, int32_t S, const double* const* zs,
                                     const int64_t* zstride, uint64_t seed, double* const* Fs, double* const* Fmeans,
                                     double* const* Fvars) {
  DS_CHECK_ARG(m && X);
  DS_CHECK_ARG(!zs || zstride);
  if (!m->prepared) DS_TRY(prepare_async(m));
  return forward_layers(m, X, n, S, zs, zstride, seed, false, true, Fs, Fmeans, Fvars);
}

// (re)build the split-K job lists for minibatch shape (n, S); uploaded once, reused by every step of that shape
static int ensure_plan(dsdgp_model* m, int64_t n, int S) {
  if (m->plan_n == n && m->plan_S == S) return DSDGP_OK;
  dsdgp_ctx* ctx = m->ctx;
  const int L = m->desc.L;
  std::vector<RedJob> red;
  for (int l = 0; l < L; ++l) {
    LayerState& St = m->L[l];
    const LayerDev& v = St.dev;
    const int64_t Rin = (l == 0) ? n : (int64_t)S * n;
    const int64_t ld = round_up(Rin, 16), nch = ld / 16;
    int NI, ti;
    wgrad_shapes(v.Mp, NI, ti);
    const int64_t MM = (int64_t)v.Mp * v.Mp;
    const int Mw = pad_Mw(v.Mp);
    const int64_t MMw = (int64_t)Mw * Mw;
    // G = E A^T is full; the D_out P_d are symmetric: off-diagonal tiles cost 1, diagonal tiles (NI+1)/(2 NI) and get that fraction
    // of the K splits, so every task carries about the same number of MFMAs.  alg_g layers have no G job (k_asm_kbar
    // assembles sum_r e a^T from the P_d).
    const int n_off = ti * (ti - 1) / 2;
    const double dfrac = (NI + 1) / (2.0 * NI);
    int ns = choose_nsplit((v.alg_g ? 0 : ti * ti) + v.D_out * n_off + (int)ceil(v.D_out * ti * dfrac), nch, 512);
    if (ns > St.nsplit_big_max) ns = St.nsplit_big_max;
    St.ns_big = ns;
    St.ns_thin = ns;
    const int ns_diag = std::max(1, (int)ceil(ns * dfrac));
    const int tjq = ceil_div(v.DP16 / 16, NI), tjz = ceil_div(v.DinP16 / 16, NI);
    double* const out_q = St.part_thin;
    double* const out_z = St.part_thin + (int64_t)ns * Mw * v.DP16;
    std::vector<WgradJob> jobsA, jobsB;
    std::vector<RedJob> redA, redB;
    int startA = 0, startB = 0;
    // in-launch split-K reduction (WgradJob::fin): a job's tiles draw their tickets from St.wtick, the result goes where the reduction
    // launch used to put it and the job leaves k_reduce_grouped's list.  By default only where the plan has ONE split (large M: more
    // tiles than workgroup slots — the "reduction" was a copy-and-mirror pass over every P_d): the tile goes from registers to its
    // place(s) with no partial, no ticket.  With several splits every workgroup of the launch would wait for its partial's write
    // acknowledgement and draw a ticket before it can leave — measured at config 2 (23 splits of ~17 us tasks, wg_red = 2): the three
    // product launches 0.114 -> 0.294 ms per step, the step 0.510 -> 0.618 ms; config 3 +2 %.  (wg_red = 0: never; 2: always — parity tests.)
    const bool wfuse = m->force.wg_red >= 2 || (m->force.wg_red == 1 && ns == 1);
    int tick_off = 0;
    auto fuse = [&](WgradJob& J, double* fin, int fin_ld, int fin_rows, int fin_cols) {
      if (!wfuse) return;
      J.fin = fin; J.fin_ld = fin_ld; J.fin_rows = fin_rows; J.fin_cols = fin_cols;
      J.tick = St.wtick + tick_off;
      tick_off += J.sym ? J.ti * (J.ti + 1) / 2 : J.ti * J.tj;
    };
    // ---- A jobs: operands from the forward chain (A, [X^T;1]) and from the producer of this layer's upstream adjoints (VB, MB)
    for (int j = 1; j <= v.D_out; ++j) {
      WgradJob J{};
      J.P = St.A; J.Q = St.A;
      J.scale = St.VB + (int64_t)(j - 1) * ld;
      J.out = St.part_big + (int64_t)j * ns * MMw;
      J.ti = ti; J.tj = ti; J.ldo = Mw; J.task_start = startA;
      J.sym = 1; J.qrows16 = Mw / 16;                     // P_d = sum_r vbar_d a a^T is symmetric
      J.ns_diag = ns_diag; J.pad = 0;
      startA += ns * n_off + ns_diag * ti;
      fuse(J, v.bigred + (int64_t)j * MM, v.Mp, v.Mp, v.Mp);
      jobsA.push_back(J);
      if (!wfuse) redA.push_back(RedJob{J.out, v.bigred + (int64_t)j * MM, MM, ns, 0, 0, v.Mp, 16, MMw, Mw, v.Mp});   // mirror at 16-block granularity
    }
    {
      WgradJob J{St.A, St.MB, nullptr, out_q, ti, tjq, v.DP16, startA, 0, v.DP16 / 16, 0, 0};      // A mbar^T -> q_mu
      fuse(J, v.thinq, v.DP16, v.Mp, v.DP16);
      jobsA.push_back(J);
    }
    startA += ns * ti * tjq;
    if (!wfuse) redA.push_back(RedJob{out_q, v.thinq, (int64_t)v.Mp * v.DP16, ns, 0, 0, 0, 0, (int64_t)Mw * v.DP16, 0, 0});
    if (St.mean_grad) {
      // trainable Linear mean function: d loss / d [A ; b] = [X ; 1]^T MB^T  (XT1 is zero-padded to whole 16*NI-row tiles)
      const int tim = ceil_div(v.DinP16 / 16, NI), tjm = ceil_div(v.DP16 / 16, NI);
      WgradJob J{St.XT1, St.MB, nullptr, St.part_mean, tim, tjm, v.DP16, startA, 0, v.DP16 / 16, 0, 0};
      fuse(J, v.meanAB, v.DP16, 16 * NI * tim, v.DP16);
      jobsA.push_back(J);
      startA += ns * tim * tjm;
      if (!wfuse) redA.push_back(RedJob{St.part_mean, v.meanAB, (int64_t)16 * NI * tim * v.DP16, ns, 0, 0, 0, 0, (int64_t)16 * NI * tim * v.DP16, 0, 0});
    }
    // ---- B jobs: operands from this layer's backward chain (E, GW)
    if (!v.alg_g) {
      WgradJob J{};
      J.P = St.E; J.Q = St.A; J.scale = nullptr;
      J.out = St.part_big;
      J.ti = ti; J.tj = ti; J.ldo = Mw; J.task_start = startB;
      J.sym = 0; J.qrows16 = Mw / 16; J.ns_diag = ns_diag; J.pad = 0;
      startB += ns * ti * ti;
      fuse(J, v.bigred, v.Mp, v.Mp, v.Mp);
      jobsB.push_back(J);
      if (!wfuse) redB.push_back(RedJob{J.out, v.bigred, MM, ns, 0, 0, 0, 16, MMw, Mw, v.Mp});
    }
    {
      WgradJob J{St.GW, St.XT1, nullptr, out_z, ti, tjz, v.DinP16, startB, 0, v.DinP16 / 16, 0, 0};   // GW [X|1]^T -> Z
      fuse(J, v.thinz, v.DinP16, v.Mp, v.DinP16);
      jobsB.push_back(J);
    }
    startB += ns * ti * tjz;
    if (!wfuse) redB.push_back(RedJob{out_z, v.thinz, (int64_t)v.Mp * v.DinP16, ns, 0, 0, 0, 0, (int64_t)Mw * v.DinP16, 0, 0});
    redB.push_back(RedJob{St.hyp_part, v.hyp_red, (int64_t)v.D_in + 2, St.gemm ? layer_gemm_hyp_parts(ld, v.Mp) : (int)sm_hyp_parts(ld, v.Mp, v.D_in), 0, 1, 0, 0, (int64_t)v.D_in + 2, 0, 0});
    // one list [A | B] with cumulative task numbers: one launch per layer
    std::vector<WgradJob> jobs(jobsA);
    for (WgradJob J : jobsB) {
      J.task_start += startA;
      jobs.push_back(J);
    }
    St.njobs = (int)jobs.size();
    St.njobsA = (int)jobsA.size();
    St.totA = startA;
    St.tot_big = startA + startB;
    St.tot_thin = 0;
    St.red_off = (int)red.size();
    red.insert(red.end(), redA.begin(), redA.end());
    red.insert(red.end(), redB.begin(), redB.end());
    St.red_n = (int)red.size() - St.red_off;
    if (tick_off > St.wtick_cap) {
      dsdgp_set_error("internal: weight-gradient ticket list overflow (%d > %d)", tick_off, St.wtick_cap);
      return DSDGP_ERR_WORKSPACE;
    }
    DS_HIP(hipMemsetAsync(St.wtick, 0, (size_t)St.wtick_cap * sizeof(int32_t), ctx->stream));
    // diagonal tiles fill only their first ns_diag partial slots: the rest must read as zero under the new plan
    DS_HIP(hipMemsetAsync(St.part_big, 0, (size_t)St.nsplit_big_max * (1 + v.D_out) * MMw * sizeof(double), ctx->stream));
    DS_HIP(hipMemcpyAsync(St.wj, jobs.data(), jobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
  }
  int blocks = 0;
  bool any64 = false;
  for (auto& r : red) {
    r.ways = (!r.wide && r.nsplit >= 12) ? 4 : 0;
    const bool even = r.count % 2 == 0 && r.pstride % 2 == 0 && r.in_ld % 2 == 0 && r.out_ld % 2 == 0 && r.sym_tile % 2 == 0 &&
                      ((uintptr_t)r.part & 15) == 0;
    if (r.ways == 4 && even) r.ways = 8;
    const int64_t rows = r.out_ld > 0 ? r.count / r.out_ld : 0;
    const bool tiled = !r.wide && r.sym_n > 0 && r.sym_tile == 16 && r.out_ld > 0 && rows == r.out_ld && rows % 16 == 0 && even;
    if (tiled) r.ways = 16;
    // one or two splits of a symmetric result on whole 64-tiles: the copy-and-mirror form (one 64 x 64 tile per workgroup)
    const bool tiled64 = tiled && r.nsplit <= 2 && rows % 64 == 0 && r.sym_n % 64 == 0 && ((uintptr_t)r.out & 15) == 0;
    if (tiled64) {
      r.ways = 64;
      any64 = true;
    }
    r.blk_start = blocks;
    blocks += r.wide ? (int)r.count
                     : (tiled64 ? (int)((rows / 64) * (rows / 64 + 1) / 2)
                                : (tiled ? (int)((rows / 16) * (rows / 16 + 1) / 2) : ceil_div(r.count, r.ways == 8 ? 128 : (r.ways == 4 ? 64 : 256))));
  }
  for (int l = 0; l < L; ++l) {
    LayerState& St = m->L[l];
    auto blk_at = [&](int idx) { return idx < (int)red.size() ? red[idx].blk_start : blocks; };
    St.red_blk0 = blk_at(St.red_off);
    St.red_blkn = blk_at(St.red_off + St.red_n) - St.red_blk0;
  }
  if ((int)red.size() > m->rjobs_cap) {
    dsdgp_set_error("internal: reduction job list overflow");
    return DSDGP_ERR_WORKSPACE;
  }
  DS_HIP(hipMemcpyAsync(m->rjobs, red.data(), red.size() * sizeof(RedJob), hipMemcpyHostToDevice, ctx->stream));
  DS_HIP(hipStreamSynchronize(ctx->stream));
  m->n_red = (int)red.size();
  m->red_blocks = blocks;
  m->red_lds = any64 ? 64 * 65 * sizeof(double) : 0;      // dynamic LDS of k_reduce_grouped: the 64 x 64 copy-and-mirror tile
  m->plan_n = n;
  m->plan_S = S;
  return DSDGP_OK;
}

static int launch_finalize(dsdgp_model* m, hipStream_t st) {
  DS_LAUNCH(k_finalize, dim3(1), dim3(256), 0, st, m->layers_dev, m->desc.L, m->lik_part, m->fin.nblocks, m->fin.w,
                     m->fin.kl_weight, m->lik_const, m->grad,
                     lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : (int64_t)-1, m->fin.with_grad, m->fin.out);
  DS_HIP(hipGetLastError());
  m->fin.done = true;
  return DSDGP_OK;
}

// Reverse pass.  Streams (when the launches are long enough to pay for cross-stream events, overlap_on):
//   main : backward chain L-1, L-2, ..., gfirst, the weight-gradient products of layer gfirst, its split-K reduction | join | the
//          other layers' reduction, P_d T_d / GS_d products, gradient assembly, value + Adam
//   side : the weight-gradient products of layer l behind an event at the end of ITS backward chain, i.e. under the chain of layer
//          l - 1 (they fill the MFMA pipe while that chain's workgroups sit in their load / reduction phases and in the launch's
//          tail); with the pipelined tail (data-parallel buckets) also that layer's reduction, products and assembly.
// Every reduction is fixed-order, so the schedule does not change a bit of the result (tests/test_gpu_parity.py:
// test_stream_overlap_is_bitwise_neutral, tests/test_gpu_round3.py).
static int backward_layers(dsdgp_model* m, int64_t n, int S, double kl_weight) {
  dsdgp_ctx* ctx = m->ctx;
  const int L = m->desc.L;
  DS_TRY(ensure_plan(m, n, S));
  const bool overlap = overlap_on(m, n, S);
  // Ku^-1, S_d (and KS, U, UU for the assembly below) come from the side stream.  In the overlapped schedule the chains wait for the event
  // behind the first of those launches only (prepare_async: ev_kinv); the join with the side stream at the end of this function — which
  // the assembly launches follow — covers the rest.  The per-layer tails (buckets) read KS / U inside the loop: they take the full join.
  const int gfirst0 = m->desc.white ? 0 : m->grad_first;
  const bool tail_in_loop = (m->bucket_fn != nullptr && m->tail_ok && gfirst0 == 0 && !m->fuse_adam.on) ||
                            (overlap && m->force.pipe_tail != 0 && !m->desc.white);
  const bool kinv_only = overlap && !tail_in_loop && !m->desc.white && m->kinv_pending && m->side_pending;
  if (kinv_only) DS_HIP(hipStreamWaitEvent(ctx->stream, m->ev_kinv, 0));
  else DS_TRY(join_prep(m));
  const int gfirst = m->desc.white ? 0 : m->grad_first;     // reverse mode stops below this layer (dsdgp_model_set_grad_first_layer)
  // data-parallel buckets: every layer's reduction, products, assembly and hyper-parameter gradients right behind its weight-gradient
  // products, then the caller's collective on that layer's segment of the gradient, on the stream the segment was produced on — the
  // exchange of the upper layers runs under the lower layers' backward chains (dsdgp_model_set_bucket_callback)
  const bool bucketed = m->bucket_fn != nullptr && m->tail_ok && gfirst == 0 && !m->fuse_adam.on;
  const bool pipelined = bucketed || (overlap && m->force.pipe_tail != 0 && !m->desc.white);
  // split-K reduction + P_d T_d / GS_d products of one layer right behind its weight-gradient products (pipelined tail)
  auto layer_tail = [&](LayerState& St, hipStream_t st) -> int {
    DS_LAUNCH(k_reduce_grouped, dim3(St.red_blkn), dim3(256), m->red_lds, st, m->rjobs + St.red_off, St.red_n, St.red_blk0);
    DS_HIP(hipGetLastError());
    DS_TRY(gemm_launch(ctx, St.lq + St.lq_nf + St.lq_n1 + St.lq_n2, St.lq_np, St.lq_tp, st));
    if (bucketed) {
      const int l = (int)(&St - m->L);
      const LayerDev* lay1 = m->layers_dev + l;
      DS_LAUNCH(k_asm_rows, dim3(St.dev.M, 1), dim3(256), (size_t)m->mp_max_all * sizeof(double), st, lay1, m->grad, kl_weight,
                         m->mp_max_all, m->force.asm_pre);
      FinArgs F{};
      AdamArgs A{};
      DS_LAUNCH(k_tail, dim3(1), dim3(256), 0, st, m->layers_dev, l, 1, m->grad, F, A);
      DS_HIP(hipGetLastError());
      // this layer's parameters are one contiguous segment of theta: [off_Z, next layer's off_Z) (the last layer's ends where the
      // likelihood variance or the vector ends)
      const int64_t lo = St.d.off_Z;
      const int64_t hi = (l + 1 < L) ? m->L[l + 1].d.off_Z
                                     : (lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : m->desc.n_theta);
      m->bucket_fn(m->bucket_user, l, m->grad + lo, hi - lo, (void*)st);
    }
    return DSDGP_OK;
  };
  // (q_mu, q_sqrt)-only gradients (dsdgp_model_set_grad_q_only): the lowest layer of the pass needs d/dq_mu = sum_r a mbar^T and
  // d/dq_sqrt_d = 2 tril(P_d q_sqrt_d), P_d = sum_r vbar_d a a^T — the forward pass's A and the upstream adjoints, nothing the backward
  // chain produces (E, GW, dX feed Z, the kernel hyper-parameters and the layers below): no chain, and only the A jobs of its products
  const bool q_only = m->grad_q_only && !m->desc.white;
  auto launch_wgrad = [&](LayerState& Sx, hipStream_t st) -> int {
    const int64_t ldx = Sx.ld_used;
    if (q_only && &Sx == &m->L[gfirst]) return wgrad_launch(ctx, Sx.wj, Sx.njobsA, Sx.totA, Sx.ns_big, ldx, ldx, st);
    return wgrad_launch(ctx, Sx.wj, Sx.njobs, Sx.tot_big, Sx.ns_big, ldx, ldx, st);
  };
  // Deep models (round 5): only the products of layer gfirst + 1 go to the side stream (they run under the lowest chain, the one launch
  // with too few row blocks to fill the chip); those of the layers above follow the lowest layer's products on the main stream — no
  // event record / wait pair per layer (6 + 12 us) and no products squeezed in beside a chain that saturates the MFMA pipe anyway.
  // 5-layer config 3: 7.60 -> 7.37 ms per step; the 3-layer configs (one deferred launch) are within +-1 %, so the rule starts at
  // four layers in the pass.  (DSDGP_FORCE wg_defer = 0 / 2: off / on at any depth.)
  const bool defer_upper = overlap && !pipelined && (m->force.wg_defer == 2 || (m->force.wg_defer < 0 && L - gfirst >= 4));
  for (int l = L - 1; l >= gfirst; --l) {
    LayerState& St = m->L[l];
    const LayerDev& v = St.dev;
    const bool last = (l == L - 1);
    const int64_t Rin = St.Rin_used, ld = St.ld_used;
    const int rep = St.rep_used;
    // transposed upstream adjoints MB / VB (+ [X^T ; 1]): written by the producer where one exists — the likelihood kernel
    // for the last layer, the next layer's backward chain for inner layers — else (first layer: S output rows per input
    // row; MultiClass) by k_adj_prep
    const bool fused = (last && m->fused_last) || (!last && l >= 1);
    // ... or, for a first layer below others, by this layer's own backward chain in its prologue (LayerBwdArgs::up_dF)
    const bool skip_chain = q_only && l == gfirst;
    const bool in_chain = !fused && !last && !skip_chain && m->force.adj_fuse != 0 && !St.c_used && !St.gemm &&
                          sm_adj_fusable(v.Mp, ld / 16, v.D_in, v.D_out);
    if (!fused && !in_chain)
      DS_LAUNCH(k_adj_prep, dim3(ceil_div(ld, 256), std::max(v.DP16, v.DinP16)), dim3(256), 0, ctx->stream, last ? nullptr : St.dF,
                         last ? m->lik_dmean : nullptr, last ? m->lik_dvar : nullptr, St.z_used, St.zs_s, St.zs_n,
                         St.zs_d, n, St.var, St.X_used, Rin, rep, v.D_in, v.D_out, v.DP16, v.DinP16, m->desc.jitter, ld,
                         St.MB, St.VB, St.XT1, v.D_out + St.prop, St.prop);
    DS_HIP(hipGetLastError());
    // the lowest layer of the reverse pass keeps its products on the main stream: nothing is left to run them under, and the
    // join below then waits for side-stream work that finished long ago instead of for a just-in-time signal
    const bool on_main = overlap && l == gfirst && L - gfirst > 1;
    LayerBwdArgs b{};
    b.X = St.X_used; b.Rin = Rin; b.D_in = v.D_in; b.D_out = v.D_out; b.M = v.M; b.DP4 = v.DP4;
    b.Zp = v.Zp; b.Zs = v.Zs; b.hyp = v.hyp; b.Kinv = v.Kinv; b.Linv = v.Linv; b.LinvT = v.LinvT; b.Sd = v.Sd; b.qmu4 = v.qmu4;
    b.Asave = St.A; b.Csave = St.c_used ? St.C : nullptr; b.Tp = v.Tp; b.TpT = v.TpT; b.ldA = ld; b.VB = St.VB; b.MB = St.MB;
    b.E = v.alg_g ? nullptr : St.E; b.GW = St.GW;
    b.dX = (l > gfirst) ? m->L[l - 1].dF : nullptr;
    if (l >= 2 && l > gfirst) {   // the previous layer is an inner layer: hand it its transposed adjoints directly
      LayerState& Pv = m->L[l - 1];
      b.dX = nullptr;
      b.MBp = Pv.MB; b.VBp = Pv.VB;
      b.zp = Pv.z_used; b.zp_s = Pv.zs_s; b.zp_n = Pv.zs_n; b.zp_d = Pv.zs_d; b.n_inner = n;
      b.varp = Pv.var; b.Dp = Pv.dev.D_out; b.prop = Pv.prop; b.jitter = m->desc.jitter;
    }
    b.mean_kind = St.d.mean_kind; b.mean_A = St.meanA;
    b.hyp_part = St.hyp_part;
    if (in_chain) {
      b.up_dF = St.dF; b.up_rep = rep; b.up_ld = v.D_out + St.prop; b.up_off = St.prop;
      b.up_z = St.z_used; b.up_zs = St.zs_s; b.up_zn = St.zs_n; b.up_zd = St.zs_d; b.up_n_inner = n;
      b.up_var = St.var; b.up_jitter = m->desc.jitter; b.MBw = St.MB; b.VBw = St.VB;
    }
    {   // few row blocks (the N-row first layer, small shards): spread the d-loop over up to four workgroups per row block.
        // Only from Mp = 512 (bwd_split = 2 forces it everywhere, 0 disables): every workgroup of a split repeats the chain's prologue
        // and epilogue phases.  63-row-block first layer of config 2 (M = 128): +57 us with the __threadfence() hand-over of
        // round 2, still +6 us per step with the fence-free sc1 hand-over of round 3 (cutting the upper layer's weight-gradient
        // task list over both streams to use the shortened chain: +10..+20 us); -0.9 ms on the 32-row-block, 30-output first
        // layer of config 4 (M = 512)
      const int64_t nblk = ld / 16;
      const bool want = m->force.bwd_split >= 2 || (m->force.bwd_split == 1 && v.Mp > 256);
      const int ds = (want && St.bpart) ? chain_d_split(nblk, v.D_out) : 1;
      b.d_split = ds; b.part = St.bpart; b.part_cnt = St.bcnt;
    }
    if (last && m->last_deferred) {
      DS_TRY(layer_last_launch(ctx, m->last_fwd, b, v.Mp, v.kern_kind, (int)(sm_hyp_parts(ld, v.Mp, v.D_in) / (ld / 16))));
      m->last_deferred = false;
    } else if (skip_chain) { /* nothing downstream of this layer's chain is wanted */ }
    else if (St.gemm) DS_TRY(layer_bwd_gemm_launch(ctx, b, v.Mp, v.kern_kind, m->desc.white ? 1 : 0, m->gws));
    else DS_TRY(layer_bwd_sm_launch(ctx, b, v.Mp, v.kern_kind, m->desc.white));
    // in-launch split-K reduction: what is left for k_reduce_grouped of this layer are the chain's hyper-parameter partials — added
    // right behind the chain (a handful of workgroups beside the side stream's products) instead of in a launch behind the final join
    if (!overlap || on_main) {
      DS_TRY(launch_wgrad(St, ctx->stream));
      if (defer_upper)
        for (int lu = gfirst + 2; lu < L; ++lu) DS_TRY(launch_wgrad(m->L[lu], ctx->stream));
      if (pipelined) DS_TRY(layer_tail(St, ctx->stream));
      continue;
    }
    if (defer_upper && l >= gfirst + 2) continue;
    // ONE event per chain boundary (an event record costs the recording stream ~6 us): behind it the side stream takes this layer's
    // products, which then run under the NEXT layer's backward chain.  Measured slower and removed in round 3: the products of a layer
    // ahead of its own chain's end (they need only the upstream adjoints), completion events attached to the chain / product launches
    // (hipExtLaunchKernelGGL: +3 us), a cap on the split count.
    hipStream_t ss = m->side;
    DS_HIP(hipEventRecord(m->ev_bwd[l], ctx->stream));
    DS_HIP(hipStreamWaitEvent(ss, m->ev_bwd[l], 0));
    DS_TRY(launch_wgrad(St, ss));
    if (pipelined) DS_TRY(layer_tail(St, ss));
  }
  // the lowest layer's products ran on the main stream: its split-K reduction goes ahead of the join, so that the side stream's
  // completion signal (a just-in-time cross-stream wait costs ~12 us of idle time) travels while the main stream works
  const bool red_ahead = overlap && !pipelined && gfirst == 0 && L > 1 && m->force.red_ahead != 0;
  if (red_ahead) {
    LayerState& S0 = m->L[0];
    DS_LAUNCH(k_reduce_grouped, dim3(S0.red_blkn), dim3(256), m->red_lds, ctx->stream, m->rjobs + S0.red_off, S0.red_n, S0.red_blk0);
    DS_HIP(hipGetLastError());
  }
  if (overlap) {
    // value + likelihood-variance gradient: needs the likelihood partials (main, before ev_bwd) and KL (side).  AFTER the
    // weight-gradient launches: its single workgroup was observed to sit for > 1 ms behind the co-running large-M chain, and
    // everything queued behind it on this stream waited with it
    if (!m->fin.done && !m->tail_ok) DS_TRY(launch_finalize(m, m->side));
    DS_HIP(hipEventRecord(m->ev_side, m->side));
    DS_HIP(hipStreamWaitEvent(ctx->stream, m->ev_side, 0));
    m->side_pending = false;      // (in order behind the parameter products: this join is theirs too)

  }
  if (!pipelined) {
    if (red_ahead) {
      LayerState& S1 = m->L[1];
      DS_LAUNCH(k_reduce_grouped, dim3(m->red_blocks - S1.red_blk0), dim3(256), m->red_lds, ctx->stream, m->rjobs + S1.red_off,
                         m->n_red - S1.red_off, S1.red_blk0);
    } else {
      // (the partial sums of the layers below a pruned pass are stale: their jobs — ordered by layer — are left out)
      const LayerState& Sg = m->L[gfirst];
      DS_LAUNCH(k_reduce_grouped, dim3(m->red_blocks - Sg.red_blk0), dim3(256), m->red_lds, ctx->stream, m->rjobs + Sg.red_off,
                         m->n_red - Sg.red_off, Sg.red_blk0);
    }
    DS_HIP(hipGetLastError());
    if (m->desc.white) {
      DS_LAUNCH(k_white_lbar, dim3(32, L), dim3(256), 0, ctx->stream, m->layers_dev);
      DS_TRY(gemm_launch(ctx, m->gp_w1, 2 * L, m->t_w1));
      DS_LAUNCH(k_white_phi, dim3(32, L), dim3(256), 0, ctx->stream, m->layers_dev);
      DS_TRY(gemm_launch(ctx, m->gp_w2, L, m->t_w2));
      DS_TRY(gemm_launch(ctx, m->gp_w3, L, m->t_w3));
    } else if (gfirst > 0) {
      for (int l = gfirst; l < L; ++l) {
        LayerState& Sq = m->L[l];
        DS_TRY(gemm_launch(ctx, Sq.lq + Sq.lq_nf + Sq.lq_n1 + Sq.lq_n2, Sq.lq_np, Sq.lq_tp));
      }
    } else {
      DS_TRY(gemm_launch(ctx, m->gp_pt, m->n_pt, m->t_pt));
    }
  }
  // the assembly of the layers that took part (their gradient entries; those of the layers below gfirst keep their old content)
  const LayerDev* lay = m->layers_dev + gfirst;
  const int La = L - gfirst;
  if (bucketed) {
    // last bucket: likelihood-variance gradient and the four result scalars (contiguous behind the layers' segments when `out`
    // is grad + n_theta, as the contract of dsdgp_allreduce asks)
    FinArgs F{m->lik_part, m->fin.nblocks, m->fin.w, m->fin.kl_weight, m->lik_const,
              lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : (int64_t)-1, m->fin.out, L, 1};
    AdamArgs A{};
    DS_LAUNCH(k_tail, dim3(1), dim3(256), 0, ctx->stream, m->layers_dev, 0, 0, m->grad, F, A);
    DS_HIP(hipGetLastError());
    m->fin.done = true;
    const int64_t lo = lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : m->desc.n_theta;
    const bool tail_scalars = m->fin.out == m->grad + m->desc.n_theta;
    m->bucket_fn(m->bucket_user, L, m->grad + lo, (m->desc.n_theta - lo) + (tail_scalars ? 4 : 0), (void*)ctx->stream);
    if (!tail_scalars) m->bucket_fn(m->bucket_user, L + 1, m->fin.out, 4, (void*)ctx->stream);
    return DSDGP_OK;
  }
  if (m->tail_ok) {
    DS_LAUNCH(k_asm_rows, dim3(m->m_max_all, La), dim3(256), (size_t)m->mp_max_all * sizeof(double), ctx->stream, lay, m->grad,
                       kl_weight, m->mp_max_all, m->force.asm_pre);
    FinArgs F{m->lik_part, m->fin.nblocks, m->fin.w, m->fin.kl_weight, m->lik_const,
              lik_has_param(m->desc.lik_kind) ? m->desc.off_lik_var : (int64_t)-1, m->fin.out, L, 1};
    AdamArgs A{m->theta, m->adam_m, m->adam_v, m->mask, m->desc.n_theta, m->fuse_adam.lr_t, m->fuse_adam.b1, m->fuse_adam.b2,
               m->fuse_adam.eps, (m->fuse_adam.on && gfirst == 0) ? 1 : 0};
    const int nadam = A.on ? (int)std::min<int64_t>(2048, ceil_div(m->desc.n_theta, 512)) : 0;      // a thread per pair of entries, eight workgroups per CU
    DS_LAUNCH(k_tail, dim3(La + 1 + nadam), dim3(256), 0, ctx->stream, m->layers_dev, gfirst, La, m->grad, F, A);
    DS_HIP(hipGetLastError());
    m->fin.done = true;
    return DSDGP_OK;
  }
  DS_LAUNCH(k_asm_kbar, dim3(m->kuu_blocks, La), dim3(256), 0, ctx->stream, lay, kl_weight);
  if (m->n_wz) DS_TRY(gemm_launch(ctx, m->gp_wz, m->n_wz, m->t_wz));   // (wm of the layers below gfirst is stale: their WZ is never read)
  if (m->need_hyp_part) DS_LAUNCH(k_asm_hyp_part, dim3(NPART, La), dim3(256), 0, ctx->stream, lay);
  DS_LAUNCH(k_asm_params, dim3(m->asm_blocks + 1, La), dim3(256), 0, ctx->stream, lay, m->grad, kl_weight);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

// minibatch to gather before the evaluation (dsdgp_model_train_step_minibatch): rows idx[0..n) of the resident data
struct GatherSrc {
  const double *Xs, *Ys;
  const int64_t* idx;
};
static int elbo_impl(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S, const double* const* zs,
                     const int64_t* zstride, uint64_t seed, double data_scale, double kl_weight, int with_grad, double* out,
                     const GatherSrc* gs) {
  DS_CHECK_ARG(m && out && ((X && Y) || gs));
  DS_CHECK_ARG(!zs || zstride);
  DS_CHECK_ARG(!m->sample_w || S == m->sample_w_S);
  if (with_grad) {
    DS_CHECK_ARG(m->grad != nullptr);
  }
  dsdgp_ctx* ctx = m->ctx;
  const int L = m->desc.L;
  // fresh N(0,1) draws do not depend on the parameters: generate them on the side stream while Ku is factorised
  bool z_side = false, z_head = false;
  const bool ovl = overlap_on(m, n, S);
  HeadRand hr{};
  if (m->head_ok) {
    // ... or, with the fused head launch, in spare block columns of that launch (no second stream, no events)
    hr.seed = seed;
    for (int l = 0; l + 1 < L; ++l)
      if (!(zs && zs[l])) {
        hr.out[l] = m->L[l].zbuf;
        hr.count[l] = (int64_t)S * n * m->L[l].dev.D_out;
        hr.nblk = std::max<int>(hr.nblk, (int)std::min<int64_t>(64, ceil_div((hr.count[l] + 1) / 2, 4 * HEAD_THREADS)));
        z_head = true;
      }
  } else if (ovl) {
    DS_HIP(hipEventRecord(m->ev_fork, ctx->stream));     // after the previous step's readers of zbuf
    DS_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    for (int l = 0; l + 1 < L; ++l)
      if (!(zs && zs[l])) {
        DS_TRY(randn_async(ctx, seed, (uint64_t)l, (int64_t)S * n * m->L[l].dev.D_out, m->L[l].zbuf, m->side));
        z_side = true;
      }
    if (z_side) DS_HIP(hipEventRecord(m->ev_z, m->side));
  }
  HeadGather hg{};
  if (gs) {
    DS_CHECK_ARG(n > 0 && n <= m->n_max);
    const int dx = m->desc.layers[0].D_in, dy = (m->desc.lik_kind == DSDGP_LIK_MULTICLASS) ? 1 : m->desc.layers[L - 1].D_out;
    if (m->head_ok) {
      hg = HeadGather{gs->Xs, gs->Ys, gs->idx, m->Xmb, m->Ymb, n, dx, dy, (int)std::min<int64_t>(16, ceil_div(n * (dx + dy), 2 * HEAD_THREADS))};
    } else {
      DS_TRY(dsdgp_gather_rows2(ctx, gs->Xs, dx, m->Xmb, gs->Ys, dy, m->Ymb, gs->idx, n, 0));
    }
    X = m->Xmb;
    Y = m->Ymb;
  }
  DS_TRY(prepare_async(m, with_grad != 0, ovl, z_head ? &hr : nullptr, (gs && m->head_ok) ? &hg : nullptr));
  if (z_side) DS_HIP(hipStreamWaitEvent(ctx->stream, m->ev_z, 0));
  LayerState& last = m->L[L - 1];
  const int DY = last.dev.D_out;
  const int64_t total = (int64_t)S * n * DY;
  int nblocks = ceil_div(total, 256);
  const double w = data_scale / (double)S;
  // the last layer of a deep model has one output row per input row: its transposed adjoints come straight from the
  // likelihood kernel (no k_adj_prep launch on the critical path) — or, Gaussian likelihood without quadrature weights, from the
  // last forward chain's own epilogue (no likelihood launch either)
  const bool elementwise = m->desc.lik_kind == DSDGP_LIK_GAUSSIAN || m->desc.lik_kind == DSDGP_LIK_BERNOULLI || lik_is_generic(m->desc.lik_kind);
  m->fused_last = with_grad && L > 1 && elementwise;
  const bool lik_in_chain = m->fused_last && m->desc.lik_kind == DSDGP_LIK_GAUSSIAN && !m->sample_w && m->force.lik_fuse != 0;
  int lik_nb = 0;
  DS_TRY(forward_layers(m, X, n, S, zs, zstride, seed, with_grad != 0, false, nullptr, nullptr, nullptr, z_side || z_head,
                        lik_in_chain ? Y : nullptr, w, &lik_nb));
  if (lik_in_chain) {
    nblocks = lik_nb;
  } else if (elementwise) {
    const int64_t ldt = round_up((int64_t)S * n, 16);
    if (m->fused_last) nblocks = ceil_div(ldt * DY, 256);
    double* dm = (with_grad && !m->fused_last) ? m->lik_dmean : nullptr;
    double* dv = (with_grad && !m->fused_last) ? m->lik_dvar : nullptr;
    double* mbt = m->fused_last ? last.MB : nullptr;
    double* vbt = m->fused_last ? last.VB : nullptr;
    if (m->desc.lik_kind == DSDGP_LIK_GAUSSIAN)
      DS_LAUNCH(k_lik_gauss, dim3(nblocks), dim3(256), 0, ctx->stream, last.mean, last.var, Y, n, S, DY, m->lik_const,
                         w, m->sample_w, m->lik_part, dm, dv, mbt, vbt, ldt);
    else if (m->desc.lik_kind == DSDGP_LIK_BERNOULLI)
      DS_LAUNCH(k_lik_bern, dim3(nblocks), dim3(256), 0, ctx->stream, last.mean, last.var, Y, n, S, DY, w, m->sample_w,
                         m->lik_part, dm, dv, mbt, vbt, ldt);
    else
      DS_LAUNCH(k_lik_gen, dim3(nblocks), dim3(256), 0, ctx->stream, (int)m->desc.lik_kind, (const double*)m->lik_const, m->desc.lik_aux,
                         last.mean, last.var, Y, n, S, DY, w, m->sample_w, m->lik_part, dm, dv, mbt, vbt, ldt);
  } else {
    // MultiClass: Y is (n x 1) labels, the last layer has K = num_classes outputs; ve per (s, i) row -> last.F scratch
    DS_CHECK_ARG(DY == m->desc.num_classes);
    const int64_t R = (int64_t)S * n;
    DS_TRY(multiclass_launch(ctx, last.mean, last.var, Y, n, R, DY, 0, w, last.F, with_grad ? m->lik_dmean : nullptr,
                             with_grad ? m->lik_dvar : nullptr, -1));
    if (m->sample_w) {
      const int64_t cnt = R * DY;
      DS_LAUNCH(k_scale_by_sample, dim3((int)std::min<int64_t>(2048, ceil_div(cnt, 256))), dim3(256), 0, ctx->stream,
                         m->sample_w, n, S, DY, R, last.F, with_grad ? m->lik_dmean : nullptr, with_grad ? m->lik_dvar : nullptr);
    }
    nblocks = ceil_div(R, 256);
    DS_LAUNCH(k_partial_sum, dim3(nblocks), dim3(256), 0, ctx->stream, last.F, R, m->lik_part);
  }
  DS_HIP(hipGetLastError());
  m->fin.nblocks = nblocks; m->fin.w = w; m->fin.kl_weight = kl_weight; m->fin.with_grad = with_grad; m->fin.out = out;
  m->fin.done = false;
  if (with_grad) {
    DS_TRY(backward_layers(m, n, S, kl_weight));
    m->grad_pruned = !m->desc.white && (m->grad_first > 0 || m->grad_q_only);
  }
  if (!m->fin.done) {
    DS_TRY(join_prep(m));   // KL values
    DS_TRY(launch_finalize(m, ctx->stream));
  }
  m->prepared = true;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_elbo(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S,
                                const double* const* zs, const int64_t* zstride, uint64_t seed, double data_scale,
                                double kl_weight, int with_grad, double* out) {
  DS_CHECK_ARG(X && Y);
  return elbo_impl(m, X, Y, n, S, zs, zstride, seed, data_scale, kl_weight, with_grad, out, nullptr);
}

extern "C" int dsdgp_model_set_sample_weights(dsdgp_model* m, const double* w, int32_t S) {
  DS_CHECK_ARG(m && (!w || (S > 0 && S <= m->s_max)));
  m->sample_w = w;
  m->sample_w_S = w ? S : 0;
  return DSDGP_OK;
}

extern "C" int dsdgp_model_adam_step(dsdgp_model* m, double lr, double beta1, double beta2, double eps, int64_t t) {
  DS_CHECK_ARG(m && m->grad && m->adam_m && m->adam_v && t >= 1);
  if (m->grad_pruned) {
    dsdgp_set_error("dsdgp_model_adam_step: the last gradient was evaluated for layers >= %d only (dsdgp_model_set_grad_first_layer)", m->grad_first);
    return DSDGP_ERR_BAD_ARG;
  }
  const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
  const int64_t n = m->desc.n_theta;
  const int nb = (int)std::min<int64_t>(2048, ceil_div(n, 512));      // a thread per pair of entries (adam_sweep), eight workgroups per CU
  DS_LAUNCH(k_adam, dim3(nb), dim3(256), 0, m->ctx->stream, m->theta, m->grad, m->adam_m, m->adam_v, m->mask, n,
                     lr_t, beta1, beta2, eps, (const LayerDev*)m->layers_dev, m->desc.L);
  DS_HIP(hipGetLastError());
  m->prepared = false;
  m->kuu_valid = false;
  m->q_dirty = -2;
  return DSDGP_OK;
}

// One optimiser step in one call: ELBO + gradient with the Adam update applied by the tail launch of the reverse pass (no separate
// k_adam launch; `session.run(opt_op)` of demos/demo_regression_UCI.ipynb:324).  Falls back to elbo + adam_step where the fused tail
// does not apply (white=True, wide inputs).
static int train_step_impl(dsdgp_model* m, const double* X, const double* Y, int64_t n, int32_t S, const double* const* zs,
                           const int64_t* zstride, uint64_t seed, double data_scale, double kl_weight, double lr, double beta1,
                           double beta2, double eps, int64_t t, double* out, const GatherSrc* gs) {
  DS_CHECK_ARG(m && m->grad && m->adam_m && m->adam_v && t >= 1);
  if (!m->desc.white && (m->grad_first > 0 || m->grad_q_only)) {
    dsdgp_set_error("dsdgp_model_train_step: the reverse pass is restricted to layers >= %d%s (dsdgp_model_set_grad_first_layer / _q_only)",
                    m->grad_first, m->grad_q_only ? ", (q_mu, q_sqrt) only" : "");
    return DSDGP_ERR_BAD_ARG;
  }
  if (!m->tail_ok) {
    DS_TRY(elbo_impl(m, X, Y, n, S, zs, zstride, seed, data_scale, kl_weight, 1, out, gs));
    return dsdgp_model_adam_step(m, lr, beta1, beta2, eps, t);
  }
  m->fuse_adam.on = 1;
  m->fuse_adam.lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
  m->fuse_adam.b1 = beta1; m->fuse_adam.b2 = beta2; m->fuse_adam.eps = eps;
  const int rc = elbo_impl(m, X, Y, n, S, zs, zstride, seed, data_scale, kl_weight, 1, out, gs);
  m->fuse_adam.on = 0;
  DS_TRY(rc);
  m->prepared = false;
  m->kuu_valid = false;
  m->q_dirty = -2;
  return DSDGP_OK;
}
