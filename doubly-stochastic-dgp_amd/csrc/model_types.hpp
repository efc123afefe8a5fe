// Descriptors of the model path: the device-resident per-layer record, reduction / weight-gradient job records, host-side layer and
// model state, and the size policies (padding, split counts, chain / GEMM path selection).  Part of the model translation unit
// (model.hip includes it; nothing else does).
#pragma once
// ------------------------------------------------------------------------------------------------------
// device-resident per-layer descriptor shared by all the small per-layer kernels
// ------------------------------------------------------------------------------------------------------
struct LayerDev {
  int32_t M, Mp, D_in, D_out, DP4, DP16, DinP16, kern_kind, ard, has_white, white, hyp_parts;   // rows of hyp2part per backward: > 0 written by k_asm_kbar (folded), < 0: -NPART rows by k_asm_hyp_part
  int32_t kl_parts, kvar_identity;   // partial sums k_kl_part leaves in klpart (the launch's block columns); dsdgp_layer_desc::kvar_identity
  int64_t off_Z, off_q_mu, off_q_sqrt, off_kvar, off_kls, off_wvar;
  double *Zp, *Zs, *hyp, *Tp, *TpT, *qmu, *qmu4;
  double *Kp, *Linv, *LinvT, *Kinv, *scal;
  double *V, *nL, *Sd, *klv;
  double *U, *n4, *PT, *UU, *Kbar, *wm, *wk;
  double *bigred, *thinq, *thinz, *hyp_red;
  double* meanAB;              // (64 ti x DP16) [X;1]^T MB^T: gradient of a trainable Linear mean function (rows: D_in of A, then b)
  int64_t off_mean_A, off_mean_b;   // -1: not a free parameter
  double *R2, *Zp1, *WZ;       // scaled squared distances of Z (Mp x Mp); [Z | 1] and wm [Z | 1] (Mp x DinP16, D_in > 32 only)
  double *klpart, *hyp2part;   // [NPART] KL partial sums ; [NPART][D_in + 2] Ku-side hyper-parameter partials
  // alg_g: sum_r e_r a_r^T is assembled from the (already needed) P_d and A mbar^T instead of a split-K product over the rows:
  //   sum_r e a^T = sum_d (2 Ku^-1 S_d - I) P_d + n (A mbar^T)^T   (e = Ku^-1 abar - g a, abar = sum_d 2 vbar_d S_d a + q_mu mbar)
  // KS_d = Ku^-1 S_d (parameter-only, side stream), GS_d = KS_d P_d (same launch as P_d T_d).  Chosen per layer when the row
  // count dwarfs D_out * M (2 D_out M^3 flops instead of 2 M^2 R, and the chain stops writing E).
  double *KS, *GS;
  int32_t alg_g, need_tpt;   // need_tpt: some chain kernel reads q_sqrt^T (Mp >= 512 row-oriented loads; the Csave backward chain)
  double *wLbar, *wH, *wY, *wX;  // white=True: Cholesky-adjoint temporaries (Mp x Mp each)
  // natural-gradient temporaries, (D_out x Mp x Mp) each unless noted
  double *ngTI, *ngTinv, *ngTbar, *ngH, *ngY, *ngX, *ngSinv, *ngA, *ngLAinv, *ngLAinvT, *ngV /* D_out x Mp */, *ngTheta1 /* D_out x Mp */, *ngScal /* 2 x D_out */;
};
#define NPART 32
#define PREP_BLOCKS 64      // minimum; large models take more (dsdgp_model::prep_blocks: ~2048 elements of q_sqrt per thread block pass)
#define WIDE_DIN 32   // layers with D_in above this take the GEMM form of the Ku-side Z / lengthscale adjoints

struct RedJob {
  const double* part;
  double* out;
  int64_t count;
  int32_t nsplit, blk_start;
  int32_t wide;        // wide: few outputs, many splits -> one workgroup per output element (fixed-order tree)
  int32_t sym_n;       // > 0: symmetric result whose tiles above the diagonal were not computed (mirrored here)
  int32_t sym_tile;
  int64_t pstride;     // elements between consecutive splits of `part`
  int32_t in_ld, out_ld;   // > 0: 2-D result, `part` rows have leading dimension in_ld (the weight-gradient products run on
                           // whole 64-row tiles: Mw = round_up(Mp, 64)), `out` rows out_ld; 0: linear
  int32_t ways;            // 4: a workgroup owns 64 outputs, each of its four waves a quarter of the splits (many splits, few outputs:
                           // one thread per output walked 156 splits eight at a time — 20 dependent round trips); else one thread each
};

struct LayerState {
  dsdgp_layer_desc d;
  LayerDev dev;
  int64_t R_max;    // max output rows (s_max * n_max)
  int64_t ld_max;
  int nsplit_big_max, nsplit_thin_max;
  double *A, *C, *E, *GW, *VB, *MB, *XT1;
  double *F, *mean, *var, *zbuf, *dF;
  const double *meanA, *meanb;   // Linear mean function: A (fixed device array or inside theta), bias or NULL
  int njobs;                     // weight-gradient jobs of this layer in the current plan
  int njobsA = 0, totA = 0;      // ... of which the first njobsA (tasks [0, totA)) read only the forward pass's A and the upstream adjoints
  double* part_mean;             // split-K partials of the mean-function gradient product (only when it is trainable)
  bool mean_grad;
  double* Xcat;     // [X_prop | F] handed to the next layer when input propagation is on (layers.py:105-110)
  int prop;
  double *part_big, *part_thin, *hyp_part;
  int32_t* wtick = nullptr;      // arrival counters of the in-launch split-K reduction, one per (job, tile) of this layer's products
  int wtick_cap = 0;
  int red_off = 0, red_n = 0, red_blk0 = 0, red_blkn = 0;   // this layer's range of the reduction job list / of its blocks
  double* bpart = nullptr;      // backward-chain d-split: partial abar tiles [row block][split][Mp * 16 + 16]
  int* bcnt = nullptr;          //   arrival counters per row block (zero between launches)
  bool big = false;    // Mp >= big_mp(), a multiple of 64: multi-workgroup blocked factorisations (linalg.hpp BigChol)
  BigChol big_k, big_ngA, big_ngT;
  GemmProblem* ng_gp;  // device: 4 natural-gradient GEMM problems (H, Sinv | Y | X)
  PotrfItem* ng_items; // device: D_out factorisation items (the index-reversed A_d)
  int ng_t1, ng_t2, ng_t3;
  // the products of THIS layer that depend on (q_mu, q_sqrt) only, as launches of their own (prepare after a natural-gradient
  // step on this layer alone: the other layers' S_d / U_d / ... are still those of the previous evaluation)
  GemmProblem* lq = nullptr;
  int lq_nf = 0, lq_tf = 0, lq_n1 = 0, lq_t1 = 0, lq_n2 = 0, lq_t2 = 0, lq_np = 0, lq_tp = 0;   // forward / U, n, KS / U U^T / P_d T_d, GS_d
  // weight-gradient jobs of ONE launch per layer, rebuilt when (n, S) changes: [A | B] — the A jobs (P_d = A diag(vbar_d) A^T, A mbar^T,
  // the mean-function product) read what the forward chain and the producer of this layer's upstream adjoints left, the B jobs
  // (E A^T, GW [X|1]^T) the backward chain's outputs
  WgradJob* wj;
  int ns_big, ns_thin, tot_big, tot_thin;
  // z actually used by the last forward (for the backward pass)
  const double* z_used;
  int64_t zs_s, zs_n, zs_d;
  const double* X_used;
  int64_t Rin_used;
  int rep_used;
  int64_t ld_used;
  bool c_used = false;   // the last forward stored c_d (Csave) for the backward chain
  bool gemm = false;     // this layer's passes are whole-layer GEMMs (layer_gemm.hip) instead of the fused chains
};

struct dsdgp_model {
  dsdgp_ctx* ctx;
  dsdgp_model_desc desc;
  int64_t n_max;
  int s_max;
  double *theta, *grad, *adam_m, *adam_v;
  LayerState L[DSDGP_MAX_LAYERS];
  LayerDev* layers_dev;
  double* mask;
  double* lik_const;   // [0] = variance, [1] = sigmoid(raw)
  double* lik_part;    // [blocks][2]
  int lik_blocks_max;
  double *lik_dmean, *lik_dvar;
  double* scal4;       // internal copy of out
  PotrfItem* potrf_items;
  GemmProblem *gp_fwd, *gp_bwd1, *gp_bwd2, *gp_w1, *gp_w2, *gp_w3;
  GemmProblem* gp_pt;   // P_d T_d (the only KL/q_sqrt GEMM that depends on the backward pass)
  int n_pt = 0, t_pt = 0;
  hipEvent_t ev_fork, ev_prep_side, ev_z;
  hipEvent_t ev_kinv;           // side stream, behind the forward-side parameter products (Ku^-1 ...): all the fused last-layer launch needs of them
  bool kinv_pending = false;
  int32_t* gemm_order = nullptr;   // device pool of the longest-processing-time tile lists of the grouped M x M launches (gemm_plan_lpt)
  int64_t gemm_order_cap = 0, gemm_order_used = 0;
  GemmLayerWs gws{};           // scratch of the GEMM-formulated layers (one set per model: the layers run one after the other)
  bool prepared_grad = false;  // the last prepare also produced U_d, n, U_d U_d^T
  bool track_theta = false;    // dsdgp_model_track_theta: the caller reports its writes to theta
  bool kuu_valid = false;      // Lu / Lu^-1 / Ku^-1 belong to the Z and kernel hyper-parameters currently in theta
  int grad_first = 0;          // dsdgp_model_set_grad_first_layer: reverse mode stops below this layer
  bool grad_pruned = false;    // the gradient buffer holds a pruned reverse pass (entries of the lower layers are stale)
  bool grad_q_only = false;    // dsdgp_model_set_grad_q_only: only the (q_mu, q_sqrt) entries of the layers >= grad_first are wanted
  int q_dirty = -2;            // with kuu_valid: -1 nothing changed, l >= 0 only layer l's (q_mu, q_sqrt) changed, -2 unknown / several
  bool side_pending = false;   // parameter-only work (Ku^-1, S_d, KL, U, UU) still running on the side stream
  int n_fwd, n_bwd1, n_bwd2, t_fwd, t_bwd1, t_bwd2, n_w, t_w1, t_w2, t_w3;
  const double* sample_w = nullptr;   // DGP_Quad quadrature weights (borrowed), NULL = Monte-Carlo mean
  int sample_w_S = 0;
  GemmProblem* gp_wz;   // wm [Z | 1] of the layers with D_in > WIDE_DIN
  int n_wz = 0, t_wz = 0, kuu_blocks = 32, asm_blocks = 64, prep_blocks = PREP_BLOCKS;
  bool uniform_big = false;     // all layers share M and Mp >= 192: ONE batched multi-workgroup Cholesky for all layers
  BigChol big_all;
  bool need_hyp_part = false;
  bool tail_ok = false;         // non-white, every D_in <= WIDE_DIN: gradient assembly in k_asm_rows + k_tail (one wave per inducing row)
  struct { int on; double lr_t, b1, b2, eps; } fuse_adam = {0, 0, 0, 0, 0};   // dsdgp_model_train_step: Adam applied inside k_tail
  int mp_max_all = 0, m_max_all = 0;
  // data-parallel buckets (dsdgp_model_set_bucket_callback): one per layer in reverse order, then the likelihood / result scalars
  dsdgp_bucket_fn bucket_fn = nullptr;
  void* bucket_user = nullptr;
  double *Xmb = nullptr, *Ymb = nullptr;   // gathered minibatch of dsdgp_model_train_step_minibatch (n_max x D_in of layer 0 / x DY)
  bool head_ok = false;         // every layer has Mp <= 128 and D_in <= 16: parameter transforms, Ku, its factorisation and inverse
                                // factor (and the inner layers' N(0,1) draws) in ONE launch (k_head, head_impl.hpp)
  bool last_deferred = false;   // the last layer's forward chain was not launched by forward_layers: it runs fused with its reverse pass (layer_last.hip)
  LayerFwdArgs last_fwd;        //   ... with these arguments
  bool fused_last = false;      // the last layer's MB / VB were written by the likelihood kernel of this step
  // arguments of the pending k_finalize (value + likelihood-variance gradient): launched on the side stream beside the
  // backward chain when streams overlap, otherwise on the main stream after it
  struct { int nblocks; double w, kl_weight; int with_grad; double* out; bool done; } fin;   // some layer does not fold its Ku-side hyper-parameter partials into k_asm_kbar
  RedJob* rjobs;       // device: split reductions of every layer, rebuilt when (n, S) changes
  int rjobs_cap, n_red, red_blocks;
  size_t red_lds = 0;
  int64_t plan_n;
  int plan_S;
  bool prepared;
  // side stream: the weight-gradient products of layer l overlap the backward chain of layer l-1 (disjoint buffers)
  hipStream_t side;
  hipEvent_t ev_bwd[DSDGP_MAX_LAYERS];
  hipEvent_t ev_side;
  bool overlap;
  // DSDGP_FORCE="key=value,...": test hooks that force the large-launch variants onto small, oracle-checkable shapes (read when
  // the model is created).  save_c: Csave backward chain 0 never / 1 for Mp > 256 / 2 every size with an instance (thresholds
  // cs_min_blocks, cs_min_dout); alg_g: algebraic dl/dKu assembly -1 heuristic / 0 never / 1 always; bwd_split: d-split of the
  // backward chain 0 off / 1 from Mp = 512 / 2 everywhere; pipe_tail: per-layer reduction + P_d T_d products behind each layer's
  // weight-gradient products 0 / 1; head / tail / adj_fuse / lik_fuse = 0: the unfused launches (parity tests of the fusions);
  // ext_ev = 0: plain event record behind the head launch; red_ahead = 0: one split-K reduction after the stream join;
  // white_fwd = 0: forward-only evaluations in plain coordinates.  last_fuse = 0: the last layer of a training step as its two chains
  // instead of the fused launch (layer_last.hip); last_min_blocks: fewest row blocks for which the fused launch is taken.  gemm_mp: smallest padded inducing count whose layers take the
  // GEMM-formulated passes (layer_gemm.hip) instead of the fused chains, 0 = never (parity tests force it onto small shapes).
  struct Force { int save_c = 1, cs_min_blocks = 160, cs_min_dout = 3, cs_max_dout = 1 << 20, wg_defer = -1, alg_g = -1, bwd_split = 1, pipe_tail = 0, head = 1, tail = 1, adj_fuse = 1, ext_ev = 1, lik_fuse = 1, red_ahead = 1, white_fwd = 1, gemm_mp = 512, last_fuse = 1, last_min_blocks = 1, asm_pre = 1, overlap_min = 1 << 18, wg_red = 1; } force;
};
static void parse_force(dsdgp_model* m) {
  const char* e = getenv("DSDGP_FORCE");
  if (!e) return;
  std::string str(e);
  size_t pos = 0;
  while (pos < str.size()) {
    size_t end = str.find(',', pos);
    if (end == std::string::npos) end = str.size();
    const std::string kv = str.substr(pos, end - pos);
    const size_t eq = kv.find('=');
    if (eq != std::string::npos) {
      const std::string k = kv.substr(0, eq);
      const int v = atoi(kv.c_str() + eq + 1);
      if (k == "save_c") m->force.save_c = v;
      else if (k == "cs_min_blocks") m->force.cs_min_blocks = v;
      else if (k == "cs_min_dout") m->force.cs_min_dout = v;
      else if (k == "cs_max_dout") m->force.cs_max_dout = v;
      else if (k == "wg_defer") m->force.wg_defer = v;
      else if (k == "alg_g") m->force.alg_g = v;
      else if (k == "bwd_split") m->force.bwd_split = v;
      else if (k == "red_ahead") m->force.red_ahead = v;
      else if (k == "white_fwd") m->force.white_fwd = v;
      else if (k == "pipe_tail") m->force.pipe_tail = v;
      else if (k == "head") m->force.head = v;
      else if (k == "tail") m->force.tail = v;
      else if (k == "adj_fuse") m->force.adj_fuse = v;
      else if (k == "ext_ev") m->force.ext_ev = v;
      else if (k == "lik_fuse") m->force.lik_fuse = v;
      else if (k == "gemm_mp") m->force.gemm_mp = v;
      else if (k == "last_fuse") m->force.last_fuse = v;
      else if (k == "last_min_blocks") m->force.last_min_blocks = v;
      else if (k == "asm_pre") m->force.asm_pre = v;
      else if (k == "overlap_min") m->force.overlap_min = v;
      else if (k == "wg_red") m->force.wg_red = v;
    }
    pos = end + 1;
  }
}

struct Bump {
  char* base;
  size_t off;
  template <class T>
  T* take(size_t count) {
    off = (size_t)round_up((int64_t)off, 256);
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Csave backward chain: the forward chain keeps c_d = q_sqrt_d^T a (D_out x Mp doubles per row) so that the backward chain's
// abar = sum_d 2 vbar_d S_d a becomes the triangular product sum_d q_sqrt_d (2 vbar_d c_d): half the MFMAs of that loop.  Measured
// (profiles/r02_fp64_mfma_notes.md): a clear win from Mp = 512 (cfg 4 +10 %, cfg 5 +14 %); at Mp = 128 / 256 the d-loop turns from
// MFMA-throughput-bound into latency-bound and the chain gets no faster, so those sizes keep the S_d form.
static bool save_c_enabled(const dsdgp_model* m, int Mp) {
  return m->force.save_c >= 2 || (m->force.save_c == 1 && Mp > 256);
}
// workgroups per row block of a chain launch with few row blocks (the d-split).  < 256 row blocks (first layers, small shards): up to
// four, ~512 workgroups.  256..511 row blocks (the per-GPU shards of configs 4 / 5: two rounds on 256 CUs, the second a quarter to a
// half full): two or three pack better, but every workgroup repeats the prologue (Kuf tile and the two triangular chains), so only
// when each keeps at least five outputs (measured: config 4, D_out = 30, -6.6 % per step; config 5, D_out = 8, +6.7 % without the rule)
static int chain_d_split(int64_t nblk, int D_out) {
  int ds = 1;
  if (nblk < 256) ds = (int)std::min<int64_t>(4, std::max<int64_t>(1, 512 / nblk));
  else if (nblk < 512) ds = (int)std::min<int64_t>(1024 / nblk, D_out / 5);
  if (ds > D_out) ds = D_out;
  return ds < 1 ? 1 : ds;
}
// padded inducing count from which the multi-workgroup blocked Cholesky / inverse (look-ahead sequence, linalg.hip) replaces the
// one-workgroup kernel: 192 (Mp a multiple of 64).  Measured, factor + inverse of one matrix: Mp = 192 180 -> 77 us, 256 371 -> 101,
// 448 1990 -> 181.  Layers that share M are factorised as ONE batch.
// the likelihood owns one positive parameter in theta (Gaussian.variance / StudentT.scale: desc.off_lik_var, lik_const[0])
static inline bool lik_has_param(int kind) {
  return kind == DSDGP_LIK_GAUSSIAN || kind == DSDGP_LIK_STUDENT_T || kind == DSDGP_LIK_GAMMA || kind == DSDGP_LIK_BETA;
}
// Poisson / Exponential / StudentT / Gamma / Beta: elementwise, evaluated by lik_var_exp (common.hpp)
static inline bool lik_is_generic(int kind) { return kind >= DSDGP_LIK_POISSON && kind <= DSDGP_LIK_BETA; }
static int big_mp(bool uniform) {
  static const int nonuniform = getenv("DSDGP_BIG_MP") ? atoi(getenv("DSDGP_BIG_MP")) : 192;      // (A/B aid)
  return uniform ? 192 : nonuniform;
}
// K splits of a weight-gradient launch: about `target_tasks` workgroup tasks (512 = two workgroups of four waves per CU), every
// wave at least two 16-row chunks
static int choose_nsplit(int tiles_per_split, int64_t nchunks, int target_tasks) {
  int ns = target_tasks / (tiles_per_split > 0 ? tiles_per_split : 1);
  if (ns < 1) ns = 1;
  int64_t cap = nchunks / 8;
  if (cap < 1) cap = 1;
  if (ns > cap) ns = (int)cap;
  return ns;
}
// 64 x 64 tiles (NI = 4 blocks of 16) on Mw = round_up(Mp, 64) rows (zero rows beyond Mp)
static void wgrad_shapes(int Mp, int& NI, int& ti) {
  NI = 4;
  ti = pad_Mw(Mp) / 64;
}
