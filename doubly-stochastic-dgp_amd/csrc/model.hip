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

// ------------------------------------------------------------------------------------------------------
// device-resident per-layer descriptor shared by all the small per-layer kernels
// ------------------------------------------------------------------------------------------------------
struct LayerDev {
  int32_t M, Mp, D_in, D_out, DP4, DP16, DinP16, kern_kind, ard, has_white, white, hyp_parts;   // rows of hyp2part per backward: > 0 written by k_asm_kbar (folded), < 0: -NPART rows by k_asm_hyp_part
  int32_t kl_parts, pad_kl;       // partial sums k_kl_part leaves in klpart (the launch's block columns)
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
  double *ngTI, *ngTinv, *ngTbar, *ngH, *ngY, *ngX, *ngSinv, *ngA, *ngLAinv, *ngLAinvT, *ngSplus, *ngTheta1 /* D_out x Mp */, *ngScal /* 4 x D_out */;
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
  double* part_mean;             // split-K partials of the mean-function gradient product (only when it is trainable)
  bool mean_grad;
  double* Xcat;     // [X_prop | F] handed to the next layer when input propagation is on (layers.py:105-110)
  int prop;
  double *part_big, *part_thin, *hyp_part;
  int red_off = 0, red_n = 0, red_blk0 = 0, red_blkn = 0;   // this layer's range of the reduction job list / of its blocks
  double* bpart = nullptr;      // backward-chain d-split: partial abar tiles [row block][split][Mp * 16 + 16]
  int* bcnt = nullptr;          //   arrival counters per row block (zero between launches)
  bool big = false;    // Mp >= 512: multi-workgroup blocked factorisations (linalg.hpp BigChol)
  BigChol big_k, big_ngA, big_ngS, big_ngT;
  GemmProblem* ng_gp;  // device: 5 natural-gradient GEMM problems (H, Sinv | Y | X | Splus)
  PotrfItem* ng_items; // device: 2 * D_out factorisation items (A_d, then Splus_d)
  int ng_t1, ng_t2, ng_t3, ng_t4;
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
  int32_t* gemm_order = nullptr;   // device pool of the longest-processing-time tile lists of the grouped M x M launches (gemm_plan_lpt)
  int64_t gemm_order_cap = 0, gemm_order_used = 0;
  GemmLayerWs gws{};           // scratch of the GEMM-formulated layers (one set per model: the layers run one after the other)
  bool prepared_grad = false;  // the last prepare also produced U_d, n, U_d U_d^T
  bool track_theta = false;    // dsdgp_model_track_theta: the caller reports its writes to theta
  bool kuu_valid = false;      // Lu / Lu^-1 / Ku^-1 belong to the Z and kernel hyper-parameters currently in theta
  int grad_first = 0;          // dsdgp_model_set_grad_first_layer: reverse mode stops below this layer
  bool grad_pruned = false;    // the gradient buffer holds a pruned reverse pass (entries of the lower layers are stale)
  int q_dirty = -2;            // with kuu_valid: -1 nothing changed, l >= 0 only layer l's (q_mu, q_sqrt) changed, -2 unknown / several
  bool side_pending = false;   // parameter-only work (Ku^-1, S_d, KL, U, UU) still running on the side stream
  int n_fwd, n_bwd1, n_bwd2, t_fwd, t_bwd1, t_bwd2, n_w, t_w1, t_w2, t_w3;
  const double* sample_w = nullptr;   // DGP_Quad quadrature weights (borrowed), NULL = Monte-Carlo mean
  int sample_w_S = 0;
  GemmProblem* gp_wz;   // wm [Z | 1] of the layers with D_in > WIDE_DIN
  int n_wz = 0, t_wz = 0, kuu_blocks = 32, asm_blocks = 64, prep_blocks = PREP_BLOCKS;
  bool uniform_big = false;     // all layers share M and Mp >= 256: ONE batched multi-workgroup Cholesky for all layers
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
  bool fused_last = false;      // the last layer's MB / VB were written by the likelihood kernel of this step
  // arguments of the pending k_finalize (value + likelihood-variance gradient): launched on the side stream beside the
  // backward chain when streams overlap, otherwise on the main stream after it
  struct { int nblocks; double w, kl_weight; int with_grad; double* out; bool done; } fin;   // some layer does not fold its Ku-side hyper-parameter partials into k_asm_kbar
  RedJob* rjobs;       // device: split reductions of every layer, rebuilt when (n, S) changes
  int rjobs_cap, n_red, red_blocks;
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
  // white_fwd = 0: forward-only evaluations in plain coordinates.  gemm_mp: smallest padded inducing count whose layers take the
  // GEMM-formulated passes (layer_gemm.hip) instead of the fused chains, 0 = never (parity tests force it onto small shapes).
  struct Force { int save_c = 1, cs_min_blocks = 160, cs_min_dout = 3, alg_g = -1, bwd_split = 1, pipe_tail = 0, head = 1, tail = 1, adj_fuse = 1, ext_ev = 1, lik_fuse = 1, red_ahead = 1, white_fwd = 1, gemm_mp = 512; } force;
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
// padded inducing count from which the multi-workgroup blocked Cholesky / inverse replaces the one-workgroup kernel: 512, or
// 256 when all layers share M and are factorised as ONE batch (measured: config 3 +2 %; per-layer sequences at 256 would lose
// to the single launch that factors all layers side by side)
static int big_mp(bool uniform) { return uniform ? 256 : 512; }
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
    v.off_kvar = d.off_kvar; v.off_kls = d.off_kls; v.off_wvar = d.off_wvar;
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
    v.ngLAinvT = b.take<double>(d.D_out * MM); v.ngSplus = b.take<double>(d.D_out * MM);
    v.ngTheta1 = b.take<double>(d.D_out * Mp); v.ngScal = b.take<double>(4 * d.D_out + 8);
    v.wLbar = b.take<double>(MM); v.wH = b.take<double>(MM); v.wY = b.take<double>(MM); v.wX = b.take<double>(MM);
    v.hyp2part = b.take<double>(1024 * (d.D_in + 2));
    {
      const int alg_env = m->force.alg_g;   // -1: heuristic, 0: never, 1: always
      const int64_t R_l = (l == 0) ? m->n_max : (int64_t)m->s_max * m->n_max;
      v.alg_g = (!D.white && (alg_env == 1 || (alg_env < 0 && (int64_t)4 * d.D_out * v.Mp <= R_l))) ? 1 : 0;
      v.need_tpt = (v.Mp > 256 || save_c_enabled(m, v.Mp) || (!D.white && m->force.gemm_mp > 0 && v.Mp >= m->force.gemm_mp)) ? 1 : 0;
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
    S.gemm = !D.white && m->force.gemm_mp > 0 && v.Mp >= m->force.gemm_mp;
    S.hyp_part = b.take<double>((size_t)(std::max<int64_t>(std::max<int64_t>(sm_hyp_parts(S.ld_max, v.Mp, d.D_in), layer_gemm_hyp_parts(S.ld_max, v.Mp)),
                                                           8 * 160) + 16) * (d.D_in + 2));
    // d-split of the backward chain on small launches (at most 1024 workgroups): partial abar tiles + arrival counters
    S.bpart = b.take<double>((size_t)1024 * (Mp * 16 + 16));
    S.bcnt = b.take<int>(512);
    S.lq = b.take<GemmProblem>(12);
    S.wj = b.take<WgradJob>(d.D_out + 4);
    S.ng_gp = b.take<GemmProblem>(5);
    S.ng_items = b.take<PotrfItem>(2 * d.D_out);
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

// ------------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double softplus_d(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

// parameter transforms + padding (LowerTriangular / positive transforms of layers.py:150 and [UPSTREAM] kernels)
// WAVE_TILES: k_prep_kuu (256 threads, large models); the head launch (512 threads, M <= 128, its LDS is the factorisation's) takes
// the one-tile-per-workgroup form with the small buffer
template <bool WAVE_TILES>
__device__ void prep_body(const LayerDev& v, const double* __restrict__ theta, double* __restrict__ lik_const, int64_t off_lik,
                          int lik_gauss, int bx, int nprep) {
  const int tid0 = bx * blockDim.x + threadIdx.x, nth = nprep * blockDim.x;
  if (tid0 == 0) {
    const double rv = theta[v.off_kvar];
    const double var = softplus_d(rv) + SOFTPLUS_LOWER;
    double wv = 0.0, dwv = 0.0;
    if (v.has_white) {
      const double rw = theta[v.off_wvar];
      wv = softplus_d(rw) + SOFTPLUS_LOWER;
      dwv = sigmoid_d(rw);
    }
    v.hyp[HYP_VAR] = var; v.hyp[HYP_WVAR] = wv; v.hyp[HYP_KDIAG] = var + wv;
    v.hyp[HYP_DVAR] = sigmoid_d(rv); v.hyp[HYP_DWVAR] = dwv;
    if (blockIdx.y == 0 && lik_gauss) {   // (grid y = layer)
      const double rl = theta[off_lik];
      lik_const[0] = softplus_d(rl) + SOFTPLUS_LOWER;
      lik_const[1] = sigmoid_d(rl);
    }
  }
  for (int j = tid0; j < v.D_in; j += nth) {
    const double rl = theta[v.off_kls + (v.ard ? j : 0)];
    v.hyp[HYP_ILS + j] = 1.0 / (softplus_d(rl) + SOFTPLUS_LOWER);
    v.hyp[HYP_ILS + v.D_in + j] = sigmoid_d(rl);
  }
  const int Mp = v.Mp, M = v.M;
  if (v.D_in > WIDE_DIN)
    for (int idx = tid0; idx < Mp * v.DinP16; idx += nth) {
      const int i = idx / v.DinP16, q = idx % v.DinP16;
      v.Zp1[idx] = (i < M) ? (q < v.D_in ? theta[v.off_Z + (int64_t)i * v.D_in + q] : (q == v.D_in ? 1.0 : 0.0)) : 0.0;
    }
  for (int idx = tid0; idx < Mp * v.D_in; idx += nth) {
    const double z = (idx / v.D_in < M) ? theta[v.off_Z + idx] : 0.0;
    const double rl = theta[v.off_kls + (v.ard ? idx % v.D_in : 0)];
    v.Zp[idx] = z;
    v.Zs[idx] = z / (softplus_d(rl) + SOFTPLUS_LOWER);
  }
  if (v.need_tpt) {
    // padded factor and its transpose, 16 x 16 tiles through LDS: both stores run along rows (the element-wise form stored the
    // transpose with stride Mp — 8 M scattered 8-byte stores per layer at M = 1024, D_out = 8: most of this launch's 195 us).
    // ONE TILE PER WAVE (lane = row lane / 4, four columns): the transposition is wave-private, no workgroup barrier, and the
    // loads of the next tile are in flight while this one is stored.
    if constexpr (WAVE_TILES) {
    __shared__ double tt[4][16][17];
    const int nt = Mp / 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tr = lane >> 2, tc = (lane & 3) * 4;
    const int64_t ntile = (int64_t)v.D_out * nt * nt;
    for (int64_t tile = (int64_t)bx * 4 + wave; tile < ntile; tile += (int64_t)nprep * 4) {
      const int d = (int)(tile / (nt * nt)), rem = (int)(tile % (nt * nt)), i0 = (rem / nt) * 16, j0 = (rem % nt) * 16;
      const int i = i0 + tr;
      double t4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + tc + u;
        const bool in = i < M && j <= i;
        t4[u] = in ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v.Tp[((int64_t)d * Mp + i) * Mp + j0 + tc + u] = t4[u];
        tt[wave][tr][tc + u] = t4[u];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's LDS stores have landed (wave-private tile: no barrier)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 4; ++u) v.TpT[((int64_t)d * Mp + j0 + tr) * Mp + i0 + tc + u] = tt[wave][tc + u][tr];
      __builtin_amdgcn_wave_barrier();
    }
    } else {
    __shared__ double tt[16][17];
    const int nt = Mp / 16, ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    for (int tile = bx; tile < v.D_out * nt * nt; tile += nprep) {
      const int d = tile / (nt * nt), rem = tile % (nt * nt), i0 = (rem / nt) * 16, j0 = (rem % nt) * 16;
      const int i = i0 + ti, j = j0 + tj;
      const bool on = threadIdx.x < 256;          // (the head launch runs this body with 512 threads per block)
      const double t = (on && i < M && j <= i) ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
      if (on) v.Tp[((int64_t)d * Mp + i) * Mp + j] = t;
      __syncthreads();
      if (on) tt[ti][tj] = t;
      __syncthreads();
      if (on) v.TpT[((int64_t)d * Mp + j0 + ti) * Mp + i0 + tj] = tt[tj][ti];
    }
    }
  } else {
    for (int idx = tid0; idx < v.D_out * Mp * Mp; idx += nth) {
      const int d = idx / (Mp * Mp), rem = idx % (Mp * Mp), i = rem / Mp, j = rem % Mp;
      v.Tp[idx] = (i < M && j <= i) ? theta[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] : 0.0;
    }
  }
  for (int idx = tid0; idx < Mp * v.D_out; idx += nth)
    v.qmu[idx] = (idx / v.D_out < M) ? theta[v.off_q_mu + idx] : 0.0;
  for (int idx = tid0; idx < Mp * v.DP4; idx += nth) {
    const int i = idx / v.DP4, d = idx % v.DP4;
    v.qmu4[idx] = (i < M && d < v.D_out) ? theta[v.off_q_mu + (int64_t)i * v.D_out + d] : 0.0;
  }
}

// Ku = K(Z,Z) + (white + jitter) I   (layers.py:171), identity on the padding
__device__ void kuu_body(const LayerDev& v, const double* __restrict__ theta, double jitter, int bx, int nbx) {
  // 16 x 16 output tile per workgroup pass; the two 16-row panels of Z are staged through LDS in 32-column chunks so that
  // wide inputs (784-d MNIST layer) read Z coalesced.  Also stores the scaled squared distances for the adjoint (k_asm_kbar).
  // Reads the raw parameters (not k_prep's outputs): both roles run in ONE launch.
  __shared__ double Zi[16][33], Zj[16][33], ils_s[32];
  const int Mp = v.Mp, nt = Mp / 16, Din = v.D_in, M = v.M;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const double var = softplus_d(theta[v.off_kvar]) + SOFTPLUS_LOWER;
  const double wvar = v.has_white ? softplus_d(theta[v.off_wvar]) + SOFTPLUS_LOWER : 0.0;
  for (int tile = bx; tile < nt * nt; tile += nbx) {
    const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
    double r2 = 0.0;
    for (int q0 = 0; q0 < Din; q0 += 32) {
      __syncthreads();
      if (tid < 32 && q0 + tid < Din) ils_s[tid] = 1.0 / (softplus_d(theta[v.off_kls + (v.ard ? q0 + tid : 0)]) + SOFTPLUS_LOWER);
      for (int e = tid; e < 512; e += 256) {
        const int r = e >> 5, c = e & 31;
        const bool ok = q0 + c < Din;
        Zi[r][c] = (ok && i0 + r < M) ? theta[v.off_Z + (int64_t)(i0 + r) * Din + q0 + c] : 0.0;
        Zj[r][c] = (ok && j0 + r < M) ? theta[v.off_Z + (int64_t)(j0 + r) * Din + q0 + c] : 0.0;
      }
      __syncthreads();
      const int qn = min(32, Din - q0);
      for (int c = 0; c < qn; ++c) {
        const double df = (Zi[ti][c] - Zj[tj][c]) * ils_s[c];
        r2 = fma(df, df, r2);
      }
    }
    const int i = i0 + ti, j = j0 + tj;
    double k = (i == j) ? 1.0 : 0.0;
    if (i < M && j < M) {
      k = kern_val_rt(v.kern_kind, r2, var);
      if (i == j) k += wvar + jitter;
    }
    v.Kp[(int64_t)i * Mp + j] = k;
    v.R2[(int64_t)i * Mp + j] = r2;
  }
}

// ONE launch for the parameter transforms / padding (first PREP_BLOCKS block columns) and Ku (the rest), grid (x, L)
__global__ __launch_bounds__(256) void k_prep_kuu(const double* __restrict__ theta, const LayerDev* __restrict__ layers,
                                                  double* __restrict__ lik_const, int64_t off_lik, int lik_gauss, double jitter,
                                                  int nprep, int keep_kuu) {
  const LayerDev v = layers[blockIdx.y];
  if ((int)blockIdx.x < nprep)
    prep_body<true>(v, theta, lik_const, off_lik, lik_gauss, blockIdx.x, nprep);
  else if (!keep_kuu)                 // keep_kuu: the factor of the unchanged Ku stays in place (dsdgp_model_track_theta)
    kuu_body(v, theta, jitter, blockIdx.x - nprep, gridDim.x - nprep);
}

// Philox4x32-10 + Box–Muller: replaces tf.random_normal (layers.py:101-102)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
// pairs t0, t0 + nth, ... of the stream (seed, stream): out[2 i], out[2 i + 1] from counter i
__device__ __forceinline__ void randn_body(uint64_t seed, uint64_t stream, int64_t count, double* __restrict__ out, int64_t t0, int64_t nth) {
  const int64_t npairs = (count + 1) / 2;
  for (int64_t i = t0; i < npairs; i += nth) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    const uint64_t a = ((uint64_t)c[1] << 32) | c[0], b = ((uint64_t)c[3] << 32) | c[2];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    out[2 * i] = rad * cs;
    if (2 * i + 1 < count) out[2 * i + 1] = rad * sn;
  }
}
// fresh N(0,1) draws of the inner layers generated inside the head launch (they depend on nothing)
struct HeadRand {
  double* out[DSDGP_MAX_LAYERS];
  int64_t count[DSDGP_MAX_LAYERS];      // 0: this layer takes no draw from here
  uint64_t seed;
  int32_t nblk;                         // block columns of the launch that generate draws
};

// the minibatch rows gathered inside the head launch too (dsdgp_model_train_step_minibatch): they depend on nothing but the indices
struct HeadGather {
  const double *Xs, *Ys;       // whole data (rows x dx / rows x dy)
  const int64_t* idx;          // row indices of this minibatch (already offset)
  double *Xd, *Yd;             // (n x dx), (n x dy)
  int64_t n;
  int32_t dx, dy, nblk;
};

#include "head_impl.hpp"

__device__ double block_sum_256(double x, double* sh) {
  x = sum_wave(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// SVGP_Layer.KL (layers.py:221-246); V = Lu^-1 q_sqrt_d and nL = Lu^-1 q_mu come from the grouped GEMM.
// grid (NPART, L): deterministic two-stage reduction.
__global__ __launch_bounds__(256) void k_kl_part(const LayerDev* __restrict__ layers) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  double acc = 0.0;
  for (int64_t idx = t0; idx < (int64_t)v.D_out * M; idx += nth) {
    const int d = (int)(idx / M), i = (int)(idx % M);
    const double t = v.Tp[((int64_t)d * Mp + i) * Mp + i];
    acc -= 0.5 * log(t * t);                                            // layers.py:235
  }
  if (!v.white) {
    {   // 1/2 |V|_F^2: 16-byte loads, four independent partial sums (a lone dependent chain ran at ~80 GB/s at M = 1024)
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2* V2 = reinterpret_cast<const d2*>(v.V);
      const int64_t n2 = (int64_t)v.D_out * Mp * Mp / 2;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int64_t idx = t0;
      for (; idx + 3 * nth < n2; idx += 4 * nth) {
        const d2 x0 = V2[idx], x1 = V2[idx + nth], x2 = V2[idx + 2 * nth], x3 = V2[idx + 3 * nth];
        a0 = fma(x0[0], x0[0], fma(x0[1], x0[1], a0));
        a1 = fma(x1[0], x1[0], fma(x1[1], x1[1], a1));
        a2 = fma(x2[0], x2[0], fma(x2[1], x2[1], a2));
        a3 = fma(x3[0], x3[0], fma(x3[1], x3[1], a3));
      }
      for (; idx < n2; idx += nth) {
        const d2 x0 = V2[idx];
        a0 = fma(x0[0], x0[0], fma(x0[1], x0[1], a0));
      }
      acc += 0.5 * ((a0 + a1) + (a2 + a3));                                                                           // :239
    }
    for (int64_t idx = t0; idx < (int64_t)Mp * v.DP4; idx += nth) acc = fma(0.5 * v.nL[idx], v.nL[idx], acc);       // :240-241
  } else {
    for (int64_t idx = t0; idx < (int64_t)v.D_out * Mp * Mp; idx += nth) acc = fma(0.5 * v.Tp[idx], v.Tp[idx], acc);  // :243
    for (int64_t idx = t0; idx < (int64_t)Mp * v.D_out; idx += nth) acc = fma(0.5 * v.qmu[idx], v.qmu[idx], acc);    // :244
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) v.klpart[blockIdx.x] = tot;
}
// KL of every layer from the partial sums of k_kl_part, by ALL 256 threads of the workgroup that forms the ELBO value (k_tail /
// k_finalize): the loads of all layers first, then one fixed-order block reduction per layer.  (A launch of its own for this sat on
// the side stream of every step: k_kl_final, 4.5 us + a launch boundary; one thread walking the partials serially cost the tail 7 us.)
// Returns sum_l KL_l (valid in thread 0) and leaves KL_l in klv[0] of each layer.
__device__ __forceinline__ double layers_kl_value(const LayerDev* __restrict__ layers, int L, double* sh) {
  double tot = 0.0;
  for (int l0 = 0; l0 < L; l0 += 4) {         // four layers' loads in flight, then their reductions
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = 0.0;
      if (l0 + u < L)
        for (int b = threadIdx.x; b < layers[l0 + u].kl_parts; b += 256) x[u] += layers[l0 + u].klpart[b];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (l0 + u < L) {
        const LayerDev& v = layers[l0 + u];
        double kl = block_sum_256(x[u], sh) - 0.5 * v.D_out * v.M;          // layers.py:234
        if (!v.white) kl += 0.5 * v.D_out * v.scal[0];                      // layers.py:238 (sum log diag Lu = logdet/2)
        if (threadIdx.x == 0) v.klv[0] = kl;
        tot += kl;
      }
  }
  return tot;
}
// (dsdgp_model_layer_kl only)
__global__ __launch_bounds__(256) void k_kl_final(const LayerDev* __restrict__ layers, int L) {
  __shared__ double sh[4];
  layers_kl_value(layers, L, sh);
}

// [UPSTREAM] Gaussian.variational_expectations (dgp.py:89-90) and its adjoints w.r.t. the last layer's mean/var.
__global__ __launch_bounds__(256) void k_lik_gauss(const double* __restrict__ mean, const double* __restrict__ var,
                                                   const double* __restrict__ Y, int64_t n, int S, int DY,
                                                   const double* __restrict__ lik_const, double w,
                                                   const double* __restrict__ sw, double* __restrict__ part,
                                                   double* __restrict__ dmean, double* __restrict__ dvar,
                                                   double* __restrict__ MBt, double* __restrict__ VBt, int64_t ldt) {
  // MBt / VBt (DY x ldt, or NULL): the adjoints stored transposed and zero-padded, i.e. already in the form the last layer's
  // backward chain reads (k_adj_prep's job when that layer has one output row per input row)
  __shared__ double sh[4];
  const double s2 = lik_const[0];
  const int64_t total = (int64_t)S * n * DY;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double ve = 0.0, dl = 0.0;
  if (idx < total) {
    const int64_t row = idx / DY;
    const int dd = (int)(idx % DY);
    const double y = Y[(row % n) * DY + dd];
    const double mu = mean[idx], v = var[idx];
    const double q = (y - mu) * (y - mu) + v;
    const double f = sw ? sw[row / n] * S : 1.0;     // quadrature weight relative to the MC mean's 1/S (dgp.py:166)
    ve = f * (-0.91893853320467274178 - 0.5 * log(s2) - 0.5 * q / s2);
    dl = f * (-0.5 / s2 + 0.5 * q / (s2 * s2));
    if (dmean) {
      dmean[idx] = -w * f * (y - mu) / s2;
      dvar[idx] = 0.5 * w * f / s2;
    }
    if (MBt) {
      MBt[(int64_t)dd * ldt + row] = -w * f * (y - mu) / s2;
      VBt[(int64_t)dd * ldt + row] = 0.5 * w * f / s2;
    }
  } else if (MBt && idx < ldt * DY) {      // rows of the 16-row padding
    MBt[(idx % DY) * ldt + idx / DY] = 0.0;
    VBt[(idx % DY) * ldt + idx / DY] = 0.0;
  }
  const double a = block_sum_256(ve, sh);
  const double b = block_sum_256(dl, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = b;
  }
}

// [UPSTREAM] Bernoulli (probit) variational expectations and their adjoints w.r.t. the last layer's mean / var; same outputs as
// k_lik_gauss (the likelihood has no parameter: the second partial is zero)
__global__ __launch_bounds__(256) void k_lik_bern(const double* __restrict__ mean, const double* __restrict__ var,
                                                  const double* __restrict__ Y, int64_t n, int S, int DY, double w,
                                                  const double* __restrict__ sw, double* __restrict__ part,
                                                  double* __restrict__ dmean, double* __restrict__ dvar,
                                                  double* __restrict__ MBt, double* __restrict__ VBt, int64_t ldt) {
  __shared__ double sh[4];
  const int64_t total = (int64_t)S * n * DY;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double ve = 0.0;
  if (idx < total) {
    const int64_t row = idx / DY;
    const int dd = (int)(idx % DY);
    const double y = Y[(row % n) * DY + dd];
    const double f = sw ? sw[row / n] * S : 1.0;
    double dm, dv;
    ve = f * bern_var_exp(mean[idx], var[idx], y, &dm, &dv);
    if (dmean) {
      dmean[idx] = -w * f * dm;
      dvar[idx] = -w * f * dv;
    }
    if (MBt) {
      MBt[(int64_t)dd * ldt + row] = -w * f * dm;
      VBt[(int64_t)dd * ldt + row] = -w * f * dv;
    }
  } else if (MBt && idx < ldt * DY) {      // rows of the 16-row padding
    MBt[(idx % DY) * ldt + idx / DY] = 0.0;
    VBt[(idx % DY) * ldt + idx / DY] = 0.0;
  }
  const double a = block_sum_256(ve, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = 0.0;
  }
}

// per-sample quadrature weights applied to per-row values (R = S*n rows) and the (R x K) adjoints (MultiClass + DGP_Quad)
__global__ void k_scale_by_sample(const double* __restrict__ sw, int64_t n, int S, int K, int64_t R, double* __restrict__ ve,
                                  double* __restrict__ dmean, double* __restrict__ dvar) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * K; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / K;
    const double f = sw[row / n] * S;
    if (i % K == 0) ve[row] *= f;
    if (dmean) {
      dmean[i] *= f;
      dvar[i] *= f;
    }
  }
}

// block partial sums of a vector (MultiClass variational expectations), same [blocks][2] layout as k_lik_gauss
__global__ __launch_bounds__(256) void k_partial_sum(const double* __restrict__ x, int64_t count, double* __restrict__ part) {
  __shared__ double sh[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double a = block_sum_256(i < count ? x[i] : 0.0, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = 0.0;
  }
}

// ELBO = data_scale/S * sum ve - kl_weight * sum KL   (dgp.py:92-98)
__global__ __launch_bounds__(256) void k_finalize(const LayerDev* __restrict__ layers, int L, const double* __restrict__ part,
                                                  int nblocks, double w, double kl_weight, const double* __restrict__ lik_const,
                                                  double* __restrict__ grad, int64_t off_lik, int with_grad,
                                                  double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += part[2 * i];
    b += part[2 * i + 1];
  }
  a = block_sum_256(a, sh);
  b = block_sum_256(b, sh);
  const double kl = layers_kl_value(layers, L, sh);
  if (threadIdx.x == 0) {
    double info = 0.0;
    for (int l = 0; l < L; ++l)
      if (layers[l].scal[1] != 0.0 && info == 0.0) info = layers[l].scal[1];
    out[0] = w * a - kl_weight * kl;
    out[1] = w * a;
    out[2] = kl_weight * kl;   // weighted like out[0]: the data-parallel SUM over ranks (kl_weight = 1/world) is then KL itself
    out[3] = info;
    if (with_grad && grad && off_lik >= 0) grad[off_lik] = -w * b * lik_const[1];
  }
}

__global__ void k_randn(uint64_t seed, uint64_t stream, int64_t count, double* __restrict__ out) {
  randn_body(seed, stream, count, out, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

__global__ void k_reparam(const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ z,
                          double jitter, int64_t count, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = mean[i] + z[i] * sqrt(var[i] + jitter);
}

// upstream adjoints of one layer, transposed to the M-major layout of the chain kernels:
//   MB[d][r] = sum_s (dF + dmean)[s,r,d]
//   VB[d][r] = sum_s (dF * z / (2 sqrt(var + jitter)) + dvar)[s,r,d]       (utils.py:41 reverse)
//   XT1      = [X^T ; 1]  (for dl/dZ = GW [X | 1])
// input propagation (layers.py:105-110): next-layer input = [ X[:, :prop] | samples ]; the `rep` output rows of one input
// row (layer 0 is evaluated once per data row) read the same X row
__global__ void k_concat_prop(const double* __restrict__ Xin, int64_t Rin, int D_in, int prop, const double* __restrict__ F,
                              int D_out, int64_t R, double* __restrict__ out) {
  const int W = prop + D_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * W; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t orow = i / W;
    const int j = (int)(i % W);
    out[i] = (j < prop) ? Xin[(orow % Rin) * D_in + j] : F[orow * D_out + (j - prop)];
  }
}

__global__ void k_adj_prep(const double* __restrict__ dF, const double* __restrict__ dmean, const double* __restrict__ dvar,
                           const double* __restrict__ z, int64_t zs_s, int64_t zs_n, int64_t zs_d, int64_t n_inner,
                           const double* __restrict__ var, const double* __restrict__ X, int64_t Rin, int rep, int D_in,
                           int D_out, int DP16, int DinP16, double jitter, int64_t ld, double* __restrict__ MB,
                           double* __restrict__ VB, double* __restrict__ XT1, int ldf, int offf) {
  // dF: adjoint of the next layer's input, (rows x ldf) with this layer's samples at column offset offf (input propagation)
  // grid: x over rows, y over the d / j index (max(DP16, DinP16) slices)
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ld) return;
  const bool ok = r < Rin;
  const int d = blockIdx.y;
  if (d < DP16) {
    double mb = 0.0, vb = 0.0;
    if (ok && d < D_out) {
      if (dF) {
        // all `rep` samples of layer 0 share var (the S input copies are identical): hoist the rsqrt, keep 4 loads in flight
        const double hv = 0.5 * rsqrt(var[r * D_out + d] + jitter);
        double m4[4] = {0, 0, 0, 0}, v4[4] = {0, 0, 0, 0};
        int s = 0;
        for (; s + 4 <= rep; s += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t orow = (int64_t)(s + u) * Rin + r;
            const double f = dF[orow * ldf + offf + d];
            m4[u] += f;
            v4[u] = fma(f, z[(orow / n_inner) * zs_s + (orow % n_inner) * zs_n + d * zs_d], v4[u]);
          }
        }
        for (; s < rep; ++s) {
          const int64_t orow = (int64_t)s * Rin + r;
          const double f = dF[orow * ldf + offf + d];
          m4[0] += f;
          v4[0] = fma(f, z[(orow / n_inner) * zs_s + (orow % n_inner) * zs_n + d * zs_d], v4[0]);
        }
        mb = (m4[0] + m4[1]) + (m4[2] + m4[3]);
        vb = ((v4[0] + v4[1]) + (v4[2] + v4[3])) * hv;
      }
      if (dmean) {
        for (int s = 0; s < rep; ++s) {
          const int64_t o = ((int64_t)s * Rin + r) * D_out + d;
          mb += dmean[o];
          vb += dvar[o];
        }
      }
    }
    MB[(int64_t)d * ld + r] = mb;
    VB[(int64_t)d * ld + r] = vb;
  }
  if (d < DinP16) {
    double v = 0.0;
    if (ok) v = (d < D_in) ? X[r * D_in + d] : (d == D_in ? 1.0 : 0.0);
    XT1[(int64_t)d * ld + r] = v;
  }
}

__global__ __launch_bounds__(256) void k_reduce_grouped(const RedJob* __restrict__ jobs, int njobs, int blk0) {
  // blk0: first block of this launch in the numbering of the whole job list (per-layer launches of a sub-range)
  __shared__ double sh[4];
  const int bx = (int)blockIdx.x + blk0;
  int jb = 0;
  while (jb + 1 < njobs && bx >= jobs[jb + 1].blk_start) ++jb;
  const RedJob J = jobs[jb];
  if (J.wide) {
    const int64_t i = bx - J.blk_start;     // one workgroup per output element
    double s = 0.0;
    for (int sp = threadIdx.x; sp < J.nsplit; sp += 256) s += J.part[(int64_t)sp * J.pstride + i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) J.out[i] = s;
    return;
  }
  const int rwave = threadIdx.x >> 6;
  if (J.ways == 16) {
    // symmetric result, one 16 x 16 tile on or below the diagonal per workgroup: each wave sums a quarter of the splits for the whole
    // tile (16-byte loads, four splits in flight per wave), the four partial tiles meet in LDS, and the tile goes out twice —
    // as it is and transposed to its mirror position — BOTH along rows.  (The element-wise form stored the mirror with stride
    // out_ld: 64 cache lines per store instruction for ~47 % of the outputs.)
    __shared__ double tl[4][16][17];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, rr = lane >> 3, cp = (lane & 7) * 2;      // lane = (row rr + 8 e, column pair cp): 16-byte loads
    int t = bx - J.blk_start, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const double* __restrict__ src = J.part + (int64_t)(16 * ti + rr) * J.in_ld + 16 * tj + cp;
    const int64_t estep = (int64_t)8 * J.in_ld;
    d2 s[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u][0] = s[u][1] = (d2){0, 0};
    int sp = rwave;
    for (; sp + 12 < J.nsplit; sp += 16) {          // four splits = eight 16-byte loads in flight per lane
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* __restrict__ p = src + (int64_t)(sp + 4 * u) * J.pstride;
        s[u][0] += *reinterpret_cast<const d2*>(p);
        s[u][1] += *reinterpret_cast<const d2*>(p + estep);
      }
    }
    for (int u = 0; sp < J.nsplit; sp += 4, ++u) {   // at most three more: into the accumulator their position in a full group would use
      const double* __restrict__ p = src + (int64_t)sp * J.pstride;
      s[u][0] += *reinterpret_cast<const d2*>(p);
      s[u][1] += *reinterpret_cast<const d2*>(p + estep);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const d2 v2 = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
      tl[rwave][rr + 8 * e][cp] = v2[0];
      tl[rwave][rr + 8 * e][cp + 1] = v2[1];
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const double v = (tl[0][r][cc] + tl[1][r][cc]) + (tl[2][r][cc] + tl[3][r][cc]);
    J.out[(int64_t)(16 * ti + r) * J.out_ld + 16 * tj + cc] = v;
    if (ti != tj) {
      __syncthreads();
      tl[0][r][cc] = v;
      __syncthreads();
      J.out[(int64_t)(16 * tj + r) * J.out_ld + 16 * ti + cc] = tl[0][cc][r];
    }
    return;
  }
  if (J.ways == 8) {
    // the 4-way form with TWO adjacent outputs per lane: 16-byte loads (8-byte lanes moved ~2 TB/s of the ~50 MB of partials that
    // config 2 reduces per step).  Same summation order per output as ways = 4: the result is bit-identical.
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ d2 sh2[4][64];
    const int lane = threadIdx.x & 63;
    const int64_t i0 = (int64_t)(bx - J.blk_start) * 128 + 2 * lane;
    const bool live = i0 < J.count;
    int64_t i = live ? i0 : 0, o = i, ostep = 1;
    if (live && J.out_ld > 0) {
      const int64_t r = i0 / J.out_ld, cc = i0 % J.out_ld;
      i = r * J.in_ld + cc;
      if (J.sym_n > 0) {
        const int64_t ti = r / J.sym_tile, tj = cc / J.sym_tile;
        if (tj > ti) {                       // see the scalar form below: read the lower tile's local element, store transposed
          const int64_t lr = r % J.sym_tile, lc = cc % J.sym_tile;
          i = (tj * J.sym_tile + lr) * J.in_ld + ti * J.sym_tile + lc;
          o = (ti * J.sym_tile + lc) * J.out_ld + tj * J.sym_tile + lr;
          ostep = J.out_ld;
        }
      }
    }
    d2 s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = (d2){0, 0};
    int sp = rwave;
    for (; sp + 28 < J.nsplit; sp += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += *reinterpret_cast<const d2*>(J.part + (int64_t)(sp + u * 4) * J.pstride + i);
    }
    for (; sp < J.nsplit; sp += 4) s[0] += *reinterpret_cast<const d2*>(J.part + (int64_t)sp * J.pstride + i);
    sh2[rwave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rwave == 0 && live) {
      const d2 t = (sh2[0][lane] + sh2[1][lane]) + (sh2[2][lane] + sh2[3][lane]);
      J.out[o] = t[0];
      J.out[o + ostep] = t[1];
    }
    return;
  }
  const bool four = J.ways == 4;
  const int64_t i0 = four ? (int64_t)(bx - J.blk_start) * 64 + (threadIdx.x & 63) : (int64_t)(bx - J.blk_start) * 256 + threadIdx.x;
  if (!four && i0 >= J.count) return;
  const bool live = i0 < J.count;
  int64_t i = live ? i0 : 0, o = i;
  if (live)
  if (J.out_ld > 0) {
    const int64_t r = i0 / J.out_ld, cc = i0 % J.out_ld;
    i = r * J.in_ld + cc;
    if (J.sym_n > 0) {
      // tiles above the diagonal were not computed: the thread that would own element (lr, lc) of upper tile (ti, tj) sums the
      // SAME local element of the lower tile (tj, ti) — coalesced partial reads — and stores it at the transposed position
      const int64_t ti = r / J.sym_tile, tj = cc / J.sym_tile;
      if (tj > ti) {
        const int64_t lr = r % J.sym_tile, lc = cc % J.sym_tile;
        i = (tj * J.sym_tile + lr) * J.in_ld + ti * J.sym_tile + lc;
        o = (ti * J.sym_tile + lc) * J.out_ld + tj * J.sym_tile + lr;
      }
    }
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int st = four ? 4 : 1;
  int sp = four ? rwave : 0;
  for (; sp + 7 * st < J.nsplit; sp += 8 * st) {     // eight independent loads in flight; fixed order -> deterministic
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += J.part[(int64_t)(sp + u * st) * J.pstride + i];
  }
  for (; sp < J.nsplit; sp += st) s[0] += J.part[(int64_t)sp * J.pstride + i];
  const double tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (!four) {
    J.out[o] = tot;
    return;
  }
  __shared__ double sh4[4][64];
  sh4[rwave][threadIdx.x & 63] = tot;
  __syncthreads();
  if (rwave == 0 && live) J.out[o] = (sh4[0][threadIdx.x] + sh4[1][threadIdx.x]) + (sh4[2][threadIdx.x] + sh4[3][threadIdx.x]);
}

// dl/dKu = -sym(G) + kl_w (D/2 Ku^-1 - 1/2 sum_d U_d U_d^T - 1/2 n n^T),  U_d = Ku^-1 q_sqrt_d, n = Ku^-1 q_mu
// then wm = Kbar ∘ dk/dr2 and wk = Kbar ∘ k / variance for the Gram adjoint.
// FOLD (D_in <= 32 and one element per thread): the Ku-side hyper-parameter partial sums of k_asm_hyp_part are taken here,
// one row of hyp2part per workgroup, while kbar / wm / wk are still in registers (one launch less on the step's tail).
__global__ __launch_bounds__(256) void k_asm_kbar(const LayerDev* __restrict__ layers, double kl_w) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M, Din = v.D_in;
  const double* G = v.bigred;
  const bool fold = v.hyp_parts > 0;
  double a_sum = 0.0, tr_sum = 0.0, wm_keep = 0.0;
  int i_keep = 0, j_keep = 0;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    double kb = 0.0, wm = 0.0, wk = 0.0;
    if (i < M && j < M) {
      double nn = 0.0, uu = 0.0;
      if (!v.white)
        for (int d = 0; d < v.D_out; ++d) {
          nn += v.n4[i * v.DP4 + d] * v.n4[j * v.DP4 + d];
          uu += v.UU[(int64_t)d * Mp * Mp + idx];
        }
      if (v.white) {
        kb = 0.5 * (v.wX[i * Mp + j] + v.wX[j * Mp + i]);   // KL(white) does not depend on Ku (layers.py:243-244)
      } else {
        double gsym;
        if (v.alg_g) {
          // sym(sum_r e a^T) = sum_d (GS_d + GS_d^T - P_d) + 1/2 (n t^T + t n^T),  t = A mbar^T (thinq)
          double gs = 0.0, nt = 0.0;
          for (int d = 0; d < v.D_out; ++d) {
            const int64_t o = (int64_t)d * Mp * Mp;
            gs += (v.GS[o + idx] + v.GS[o + j * Mp + i]) - v.bigred[(int64_t)Mp * Mp + o + idx];
            nt += v.n4[i * v.DP4 + d] * v.thinq[j * v.DP16 + d] + v.n4[j * v.DP4 + d] * v.thinq[i * v.DP16 + d];
          }
          gsym = gs + 0.5 * nt;
        } else {
          gsym = 0.5 * (G[i * Mp + j] + G[j * Mp + i]);
        }
        kb = -gsym + kl_w * (0.5 * v.D_out * v.Kinv[idx] - 0.5 * uu - 0.5 * nn);
      }
      const double r2 = v.R2[idx];
      double k, dk;
      if (v.kern_kind == DSDGP_KERN_RBF)
        kern_val_grad<DSDGP_KERN_RBF>(r2, v.hyp[HYP_VAR], k, dk);
      else
        kern_val_grad<DSDGP_KERN_MATERN52>(r2, v.hyp[HYP_VAR], k, dk);
      wm = kb * dk;
      wk = kb * k / v.hyp[HYP_VAR];
      a_sum += wk;
      if (i == j) tr_sum += kb;
      wm_keep = wm; i_keep = i; j_keep = j;
    }
    v.Kbar[idx] = kb;
    v.wm[idx] = wm;
    v.wk[idx] = wk;
  }
  if (!fold) return;
  double* out = v.hyp2part + (int64_t)blockIdx.x * (Din + 2);
  a_sum = block_sum_256(a_sum, sh);
  tr_sum = block_sum_256(tr_sum, sh);
  if (threadIdx.x == 0) {
    out[0] = a_sum;
    out[1] = tr_sum;
  }
  for (int q = 0; q < Din; ++q) {
    const double df = v.Zp[i_keep * Din + q] - v.Zp[j_keep * Din + q];
    const double sq = block_sum_256(wm_keep * df * df, sh);
    if (threadIdx.x == 0) out[2 + q] = sq;
  }
}

// white=True helpers: Lu_bar = -tril(G) ; Phi(H) = tril(H) with halved diagonal (in place)
__global__ void k_white_lbar(const LayerDev* __restrict__ layers) {
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    v.wLbar[idx] = (i < v.M && j <= i) ? -v.bigred[idx] : 0.0;
  }
}
__global__ void k_white_phi(const LayerDev* __restrict__ layers) {
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Mp * Mp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Mp, j = idx % Mp;
    const double h = v.wH[idx];
    v.wH[idx] = (j < i) ? h : (j == i ? 0.5 * h : 0.0);
  }
}

// final assembly of d loss / d theta: elementwise part, grid (blocks, L)
__device__ void asm_hyp_final(const LayerDev& v, double* __restrict__ grad);
__global__ __launch_bounds__(256) void k_asm_params(const LayerDev* __restrict__ layers, double* __restrict__ grad, double kl_w) {
  const LayerDev v = layers[blockIdx.y];
  if (blockIdx.x == gridDim.x - 1) {   // extra block: hyper-parameter gradients of this layer
    asm_hyp_final(v, grad);
    return;
  }
  const int Mp = v.Mp, M = v.M, Din = v.D_in, Dout = v.D_out;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)(gridDim.x - 1) * blockDim.x;
  const double* ils = v.hyp + HYP_ILS;
  // q_sqrt: 2 tril(P_d T_d) + kl_w (Ku^-1 T_d - diag(1/T_ii))
  for (int64_t idx = t0; idx < (int64_t)Dout * M * M; idx += nth) {
    const int d = (int)(idx / ((int64_t)M * M)), rem = (int)(idx % ((int64_t)M * M)), i = rem / M, j = rem % M;
    double gq = 0.0;
    if (j <= i) {
      const int64_t p = ((int64_t)d * Mp + i) * Mp + j;
      gq = 2.0 * v.PT[p] + kl_w * ((v.white ? v.Tp[p] : v.U[p]) - (i == j ? 1.0 / v.Tp[p] : 0.0));
    }
    grad[v.off_q_sqrt + idx] = gq;
  }
  // trainable Linear mean function: rows j < D_in of [X;1]^T MB^T are d loss / d A, row D_in is d loss / d b
  if (v.meanAB) {
    if (v.off_mean_A >= 0)
      for (int64_t idx = t0; idx < (int64_t)Din * Dout; idx += nth) grad[v.off_mean_A + idx] = v.meanAB[(idx / Dout) * v.DP16 + idx % Dout];
    if (v.off_mean_b >= 0)
      for (int64_t idx = t0; idx < Dout; idx += nth) grad[v.off_mean_b + idx] = v.meanAB[(int64_t)Din * v.DP16 + idx];
  }
  // q_mu: A mbar + kl_w Ku^-1 q_mu
  for (int64_t idx = t0; idx < (int64_t)M * Dout; idx += nth) {
    const int i = (int)(idx / Dout), d = (int)(idx % Dout);
    grad[v.off_q_mu + idx] = v.thinq[i * v.DP16 + d] + kl_w * (v.white ? v.qmu4[i * v.DP4 + d] : v.n4[i * v.DP4 + d]);
  }
  // Z: through Kuf (GW [X|1]) and through Ku (wm).  Wide inputs: sum_j wm_ij (z_iq - z_jq) = rowsum_i z_iq - (wm Z)_iq with
  // WZ = wm [Z | 1] from the MFMA GEMM; otherwise one wavefront per (i, q), lanes stride over j
  if (Din > WIDE_DIN) {
    for (int64_t idx = t0; idx < (int64_t)M * Din; idx += nth) {
      const int i = (int)(idx / Din), q = (int)(idx % Din);
      const double zi = v.Zp[idx];
      const double s = v.WZ[(int64_t)i * v.DinP16 + Din] * zi - v.WZ[(int64_t)i * v.DinP16 + q];
      const double il2 = ils[q] * ils[q];
      grad[v.off_Z + idx] = 4.0 * il2 * s - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
    }
  } else {
    const int lane = threadIdx.x & 63;
    const int64_t w0 = t0 >> 6, nw = nth >> 6;
    for (int64_t idx = w0; idx < (int64_t)M * Din; idx += nw) {
      const int i = (int)(idx / Din), q = (int)(idx % Din);
      const double zi = v.Zp[i * Din + q];
      double s = 0.0;
      for (int j = lane; j < M; j += 64) s = fma(v.wm[i * Mp + j], zi - v.Zp[j * Din + q], s);
      s = sum_wave(s);
      if (lane == 0) {
        const double il2 = ils[q] * ils[q];
        grad[v.off_Z + idx] = 4.0 * il2 * s - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
      }
    }
  }
}

// Ku-side hyper-parameter partial sums, grid (NPART, L): part[b] = { sum wk, trace Kbar, sum_q wm (z_i-z_j)_q^2 ... }
__global__ __launch_bounds__(256) void k_asm_hyp_part(const LayerDev* __restrict__ layers) {
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int Mp = v.Mp, M = v.M, Din = v.D_in;
  if (v.hyp_parts > 0) return;   // folded into k_asm_kbar
  const int t0 = blockIdx.x * 256 + threadIdx.x, nth = NPART * 256;
  double* out = v.hyp2part + (int64_t)blockIdx.x * (Din + 2);
  double a = 0.0, tr = 0.0;
  for (int idx = t0; idx < M * M; idx += nth) {
    const int i = idx / M, j = idx % M;
    a += v.wk[i * Mp + j];
    if (i == j) tr += v.Kbar[i * Mp + i];
  }
  a = block_sum_256(a, sh);
  tr = block_sum_256(tr, sh);
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = tr;
  }
  if (Din > WIDE_DIN) {
    // sum_ij wm_ij (z_iq - z_jq)^2 = 2 sum_i z_iq (rowsum_i z_iq - (wm Z)_iq)   (wm symmetric); rows i = b mod NPART
    for (int q = threadIdx.x; q < Din; q += 256) {
      double s = 0.0;
      for (int i = blockIdx.x; i < M; i += NPART) {
        const double zi = v.Zp[(int64_t)i * Din + q];
        s = fma(2.0 * zi, v.WZ[(int64_t)i * v.DinP16 + Din] * zi - v.WZ[(int64_t)i * v.DinP16 + q], s);
      }
      out[2 + q] = s;
    }
    return;
  }
  for (int q = 0; q < Din; ++q) {
    double s = 0.0;
    for (int idx = t0; idx < M * M; idx += nth) {
      const int i = idx / M, j = idx % M;
      const double df = v.Zp[i * Din + q] - v.Zp[j * Din + q];
      s = fma(v.wm[i * Mp + j], df * df, s);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[2 + q] = s;
  }
}
// kernel hyper-parameter gradients from the partial sums (one workgroup per layer; runs as the LAST block row of
// k_asm_params — its inputs come from kernels that precede that launch)
__device__ void asm_hyp_final(const LayerDev& v, double* __restrict__ grad) {
  __shared__ double sh[4];
  __shared__ double gl_s[64];
  const int Din = v.D_in, parts = abs(v.hyp_parts), stride = Din + 2;
  const double* ils = v.hyp + HYP_ILS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    double a = 0.0, tr = 0.0;
    for (int b = threadIdx.x; b < parts; b += 256) {
      a += v.hyp2part[b * stride];
      tr += v.hyp2part[b * stride + 1];
    }
    a = block_sum_256(a, sh);
    tr = block_sum_256(tr, sh);
    if (threadIdx.x == 0) {
      grad[v.off_kvar] = (a + v.hyp_red[0] + v.hyp_red[1]) * v.hyp[HYP_DVAR];
      if (v.has_white) grad[v.off_wvar] = (tr + v.hyp_red[1]) * v.hyp[HYP_DWVAR];
    }
  }
  double iso = 0.0;
  if (Din <= 64) {
    // few lengthscales: one wavefront per q, lanes over the partial rows (a serial walk over the rows is latency-bound)
    for (int q = wave; q < Din; q += 4) {
      double s = 0.0;
      for (int b = lane; b < parts; b += 64) s += v.hyp2part[b * stride + 2 + q];
      s = sum_wave(s);
      if (lane == 0) gl_s[q] = -2.0 * ils[q] * ils[q] * ils[q] * s + v.hyp_red[2 + q];
    }
    __syncthreads();
    if (threadIdx.x < Din) {
      const double gl = gl_s[threadIdx.x];
      if (v.ard)
        grad[v.off_kls + threadIdx.x] = gl * v.hyp[HYP_ILS + Din + threadIdx.x];
      else
        iso = gl;
    }
  } else {
    for (int q = threadIdx.x; q < Din; q += 256) {
      double s = 0.0;
      for (int b = 0; b < parts; ++b) s += v.hyp2part[b * stride + 2 + q];
      const double gl = -2.0 * ils[q] * ils[q] * ils[q] * s + v.hyp_red[2 + q];
      if (v.ard)
        grad[v.off_kls + q] = gl * v.hyp[HYP_ILS + Din + q];
      else
        iso += gl;
    }
  }
  iso = block_sum_256(iso, sh);
  if (!v.ard && threadIdx.x == 0) grad[v.off_kls] = iso * v.hyp[HYP_ILS + Din];
}
__global__ void k_adam(double* __restrict__ theta, const double* __restrict__ grad, double* __restrict__ m,
                       double* __restrict__ v, const double* __restrict__ mask, int64_t n, double lr_t, double b1,
                       double b2, double eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (mask[i] == 0.0) continue;
    const double g = grad[i];
    const double mi = b1 * m[i] + (1.0 - b1) * g;
    const double vi = b2 * v[i] + (1.0 - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= lr_t * mi / (sqrt(vi) + eps);
  }
}

// ------------------------------------------------------------------------------------------------------
// Fused tail (non-white models whose layers all have D_in <= WIDE_DIN): k_asm_kbar + k_asm_params become ONE pass with a wave per
// inducing row, k_finalize + the hyper-parameter reduction + (single-process training) the Adam update a second small launch.
// Before: reduce 22 us -> P_d T_d 13 -> k_asm_kbar 15 -> k_asm_params 7 -> k_adam 5 (+ k_finalize 5 on the side stream and its join).
// ------------------------------------------------------------------------------------------------------
// Row i of dl/dKu = -sym(G) + kl_w (D/2 Ku^-1 - 1/2 sum_d U_d U_d^T - 1/2 n n^T) (k_asm_kbar's formula) goes to an LDS row as
// wm = Kbar ∘ dk/dr2; from it the Z gradient of row i and this row's partial sums of the kernel hyper-parameter gradients (hyp2part
// row i: sum wk, Kbar_ii, sum_j wm_ij (z_iq - z_jq)^2), then the q_mu / q_sqrt gradient rows.  One WORKGROUP per inducing row: its
// four waves share the row of Kbar, then split the input dimensions and the (output, column) pairs of the q_sqrt rows — a wave per
// row walked ten dependent memory round trips one after the other (30 us).  grid (M_max, layers), 256 threads, mp_max doubles of LDS.
__global__ __launch_bounds__(256) void k_asm_rows(const LayerDev* __restrict__ layers, double* __restrict__ grad, double kl_w, int mp_max) {
  extern __shared__ __attribute__((aligned(16))) double asm_dyn[];
  __shared__ double sh[4];
  const LayerDev v = layers[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = (int)blockIdx.x;
  const int Mp = v.Mp, M = v.M, Din = v.D_in, Dout = v.D_out;
  if (i >= M) return;
  lptr wm = (lptr)asm_dyn;
  const int64_t MM = (int64_t)Mp * Mp;
  const double kvar = v.hyp[HYP_VAR];
  double a_sum = 0.0, tr = 0.0;
  for (int j = tid; j < M; j += 256) {
    const int64_t idx = (int64_t)i * Mp + j, idt = (int64_t)j * Mp + i;
    double nn = 0.0, uu = 0.0, gsym;
    if (v.alg_g) {
      // sym(sum_r e a^T) = sum_d (GS_d + GS_d^T - P_d) + 1/2 (n t^T + t n^T),  t = A mbar^T (thinq)
      // ONE loop over the outputs, unrolled: every load of four outputs is in flight before the first use (two loops of
      // one output per iteration walked 2 D_out dependent round trips)
      double gs = 0.0, nt = 0.0;
#pragma unroll 4
      for (int d = 0; d < Dout; ++d) {
        const double ni = v.n4[i * v.DP4 + d], nj = v.n4[j * v.DP4 + d];
        const double g1 = v.GS[d * MM + idx], g2 = v.GS[d * MM + idt], pd = v.bigred[MM + d * MM + idx];
        const double tj = v.thinq[j * v.DP16 + d], ti2 = v.thinq[i * v.DP16 + d];
        nn = fma(ni, nj, nn);
        uu += v.UU[d * MM + idx];
        gs += (g1 + g2) - pd;
        nt += ni * tj + nj * ti2;
      }
      gsym = gs + 0.5 * nt;
    } else {
#pragma unroll 4
      for (int d = 0; d < Dout; ++d) {
        nn = fma(v.n4[i * v.DP4 + d], v.n4[j * v.DP4 + d], nn);
        uu += v.UU[d * MM + idx];
      }
      gsym = 0.5 * (v.bigred[idx] + v.bigred[idt]);
    }
    const double kb = -gsym + kl_w * (0.5 * Dout * v.Kinv[idx] - 0.5 * uu - 0.5 * nn);
    double k, dk;
    if (v.kern_kind == DSDGP_KERN_RBF)
      kern_val_grad<DSDGP_KERN_RBF>(v.R2[idx], kvar, k, dk);
    else
      kern_val_grad<DSDGP_KERN_MATERN52>(v.R2[idx], kvar, k, dk);
    wm[j] = kb * dk;
    a_sum += kb * k / kvar;
    if (j == i) tr = kb;
  }
  a_sum = block_sum_256(a_sum, sh);           // (its barriers also publish the wm row)
  tr = block_sum_256(tr, sh);
  double* __restrict__ hp = v.hyp2part + (int64_t)i * (Din + 2);
  if (tid == 0) {
    hp[0] = a_sum;
    hp[1] = tr;
  }
  const double* __restrict__ ils = v.hyp + HYP_ILS;
  for (int q = wave; q < Din; q += 4) {
    const double zi = v.Zp[i * Din + q];
    double s1 = 0.0, s2 = 0.0;
    for (int j = lane; j < M; j += 64) {
      const double df = zi - v.Zp[j * Din + q], w = wm[j];
      s1 = fma(w, df, s1);
      s2 = fma(w * df, df, s2);
    }
    s1 = sum_wave(s1);
    s2 = sum_wave(s2);
    if (lane == 0) {
      const double il2 = ils[q] * ils[q];
      grad[v.off_Z + (int64_t)i * Din + q] = 4.0 * il2 * s1 - 2.0 * il2 * (v.thinz[i * v.DinP16 + q] - zi * v.thinz[i * v.DinP16 + Din]);
      hp[2 + q] = s2;
    }
  }
  // q_mu: A mbar + kl_w Ku^-1 q_mu
  for (int d = tid; d < Dout; d += 256) grad[v.off_q_mu + (int64_t)i * Dout + d] = v.thinq[i * v.DP16 + d] + kl_w * v.n4[i * v.DP4 + d];
  // q_sqrt: 2 tril(P_d T_d) + kl_w (Ku^-1 T_d - diag(1/T_ii)); clamped (unconditional) loads so that several are in flight
  for (int e = tid; e < Dout * M; e += 256) {
    const int d = e / M, j = e - d * M;
    const int64_t p = d * MM + (int64_t)i * Mp + (j <= i ? j : i);
    const double gq = 2.0 * v.PT[p] + kl_w * (v.U[p] - (i == j ? 1.0 / v.Tp[p] : 0.0));
    grad[v.off_q_sqrt + ((int64_t)d * M + i) * M + j] = (j <= i) ? gq : 0.0;
  }
  // trainable Linear mean function: rows j < D_in of [X;1]^T MB^T are d loss / d A, row D_in is d loss / d b
  if (v.meanAB) {
    if (v.off_mean_A >= 0)
      for (int64_t idx = (int64_t)i * 256 + tid; idx < (int64_t)Din * Dout; idx += (int64_t)M * 256)
        grad[v.off_mean_A + idx] = v.meanAB[(idx / Dout) * v.DP16 + idx % Dout];
    if (v.off_mean_b >= 0 && i == 0)
      for (int idx = tid; idx < Dout; idx += 256) grad[v.off_mean_b + idx] = v.meanAB[(int64_t)Din * v.DP16 + idx];
  }
}

struct AdamArgs {
  double* theta; double* m; double* v; const double* mask;
  int64_t n;
  double lr_t, b1, b2, eps;
  int32_t on;
};
__device__ __forceinline__ void adam_one(const AdamArgs& A, int64_t i, double g) {
  const double mi = A.b1 * A.m[i] + (1.0 - A.b1) * g;
  const double vi = A.b2 * A.v[i] + (1.0 - A.b2) * g * g;
  A.m[i] = mi;
  A.v[i] = vi;
  A.theta[i] -= A.lr_t * mi / (sqrt(vi) + A.eps);
}
struct FinArgs {
  const double* part; int nblocks; double w, kl_weight; const double* lik_const; int64_t off_lik; double* out; int L;
  int do_fin;      // 0: no ELBO-value block in this launch (per-layer launches of the bucketed data-parallel tail)
};
// blocks 0 .. La-1: kernel hyper-parameter gradients of layer first + b from the row partials (asm_hyp_final);
// block La: ELBO value + likelihood-variance gradient (k_finalize's job);  blocks > La (only with A.on): Adam on every entry those
// blocks do not own (mask 1), the owners apply it to theirs (mask 2) right after writing the gradient.
__global__ __launch_bounds__(256) void k_tail(const LayerDev* __restrict__ layers_all, int first, int La, double* __restrict__ grad,
                                              const FinArgs F, const AdamArgs A) {
  __shared__ double sh[4];
  const int b = (int)blockIdx.x;
  if (b < La) {
    const LayerDev v = layers_all[first + b];
    asm_hyp_final(v, grad);
    if (A.on) {
      __syncthreads();      // (the values were written by threads of this block: re-read below by the same threads that wrote them)
      const int Din = v.D_in;
      if (threadIdx.x == 0) {
        if (A.mask[v.off_kvar] != 0.0) adam_one(A, v.off_kvar, grad[v.off_kvar]);
        if (v.has_white && A.mask[v.off_wvar] != 0.0) adam_one(A, v.off_wvar, grad[v.off_wvar]);
        if (!v.ard && A.mask[v.off_kls] != 0.0) adam_one(A, v.off_kls, grad[v.off_kls]);
      }
      if (v.ard && (int)threadIdx.x < Din && A.mask[v.off_kls + threadIdx.x] != 0.0)
        adam_one(A, v.off_kls + threadIdx.x, grad[v.off_kls + threadIdx.x]);
    }
    return;
  }
  const int nf = F.do_fin ? 1 : 0;
  if (b == La && nf) {
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < F.nblocks; i += 256) {
      a += F.part[2 * i];
      c += F.part[2 * i + 1];
    }
    a = block_sum_256(a, sh);
    c = block_sum_256(c, sh);
    const double kl = layers_kl_value(layers_all, F.L, sh);
    if (threadIdx.x == 0) {
      double info = 0.0;
      for (int l = 0; l < F.L; ++l)
        if (layers_all[l].scal[1] != 0.0 && info == 0.0) info = layers_all[l].scal[1];
      F.out[0] = F.w * a - F.kl_weight * kl;
      F.out[1] = F.w * a;
      F.out[2] = F.kl_weight * kl;
      F.out[3] = info;
      if (F.off_lik >= 0) {
        const double g = -F.w * c * F.lik_const[1];
        grad[F.off_lik] = g;
        if (A.on && A.mask[F.off_lik] != 0.0) adam_one(A, F.off_lik, g);
      }
    }
    return;
  }
  const int64_t nth = (int64_t)(gridDim.x - La - nf) * 256;
  for (int64_t i = (int64_t)(b - La - nf) * 256 + threadIdx.x; i < A.n; i += nth)
    if (A.mask[i] == 1.0) adam_one(A, i, grad[i]);
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int validate_desc(const dsdgp_model_desc* d) {
  DS_CHECK_ARG(d && d->L >= 1 && d->L <= DSDGP_MAX_LAYERS && d->n_theta > 0);
  DS_CHECK_ARG(d->lik_kind == DSDGP_LIK_GAUSSIAN || d->lik_kind == DSDGP_LIK_MULTICLASS || d->lik_kind == DSDGP_LIK_BERNOULLI);
  for (int l = 0; l < d->L; ++l) {
    const dsdgp_layer_desc& y = d->layers[l];
    DS_CHECK_ARG(y.M >= 1 && y.D_in >= 1 && y.D_out >= 1);
    DS_CHECK_ARG(y.kern_kind == DSDGP_KERN_RBF || y.kern_kind == DSDGP_KERN_MATERN52);
    if (pad_M(y.M) > 1024) {
      dsdgp_set_error("layer %d: M=%d inducing points exceeds the built chain kernels (<= 1024)", l, y.M);
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
  dsdgp_model* m = new dsdgp_model();
  parse_force(m);
  m->ctx = ctx;
  m->desc = *desc;
  m->n_max = n_max;
  m->s_max = s_max;
  m->theta = theta; m->grad = grad; m->adam_m = adam_m; m->adam_v = adam_v;
  size_t total = 0;
  layout(m, (char*)workspace, &total);
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
      GemmProblem ng[5];
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
      fill_gemm(ng[4], v.ngLAinvT, v.ngLAinv, v.ngSplus, Mp, Mp, Mp, Mp, Mp, Mp, 0, 0, v.D_out, MM, MM, MM, 0);  // S+
      ng[4].lower_only = 1; ng[4].tri = 8 | 1 | 16;     // upper x lower, symmetric
      St.ng_t4 = plan_lpt(ng + 4, 1);
      DS_HIP(hipMemcpyAsync(St.ng_gp, ng, sizeof(ng), hipMemcpyHostToDevice, st));
      std::vector<PotrfItem> it(2 * v.D_out);
      for (int d = 0; d < v.D_out; ++d) {
        it[d] = PotrfItem{v.ngA + d * MM, v.ngLAinv + d * MM, v.ngLAinvT + d * MM, v.ngScal + 2 * d, Mp, Mp, v.M, 0, 0, 0};
        it[v.D_out + d] = PotrfItem{v.ngSplus + d * MM, nullptr, nullptr, v.ngScal + 2 * v.D_out + 2 * d, Mp, Mp, v.M, 0, 0, 0};
      }
      DS_HIP(hipMemcpyAsync(St.ng_items, it.data(), it.size() * sizeof(PotrfItem), hipMemcpyHostToDevice, st));
      DS_HIP(hipStreamSynchronize(st));
      St.big = Mp >= big_mp(m->uniform_big);
      if (St.big) {
        if (!m->uniform_big) DS_TRY(bigchol_build(ctx, St.big_k, v.Kp, v.Linv, v.LinvT, v.scal, 1, MM, 2, Mp, v.M, nullptr, false));
        else if (l == 0) DS_TRY(bigchol_build(ctx, m->big_all, v.Kp, v.Linv, v.LinvT, v.scal, L, MM, 8, Mp, v.M, nullptr, false));
        DS_TRY(bigchol_build(ctx, St.big_ngA, v.ngA, v.ngLAinv, v.ngLAinvT, v.ngScal, v.D_out, MM, 2, Mp, v.M, nullptr, false));
        DS_TRY(bigchol_build(ctx, St.big_ngS, v.ngSplus, nullptr, nullptr, v.ngScal + 2 * v.D_out, v.D_out, MM, 2, Mp, v.M, nullptr, false));
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
    mark(y.off_q_sqrt, (int64_t)y.D_out * y.M * y.M, y.trainable_q_sqrt);
    // (2: entries whose gradient k_tail's hyper-parameter / likelihood blocks produce and, in a fused training step, update themselves)
    mark(y.off_kvar, 1, 2 * y.trainable_kvar);
    mark(y.off_kls, y.ard ? y.D_in : 1, 2 * y.trainable_kls);
    if (y.has_white) mark(y.off_wvar, 1, 2 * y.trainable_wvar);
    if (y.mean_kind == DSDGP_MEAN_LINEAR && y.off_mean_A >= 0) mark(y.off_mean_A, (int64_t)y.D_in * y.D_out, y.trainable_mean_A);
    if (y.mean_kind == DSDGP_MEAN_LINEAR && y.off_mean_b >= 0) mark(y.off_mean_b, y.D_out, y.trainable_mean_b);
  }
  if (desc->lik_kind == DSDGP_LIK_GAUSSIAN) mark(desc->off_lik_var, 1, 2 * desc->trainable_lik_var);
  DS_HIP(hipMemcpyAsync(m->mask, mask.data(), mask.size() * sizeof(double), hipMemcpyHostToDevice, st));
  DS_HIP(hipStreamSynchronize(st));
  m->overlap = !(getenv("DSDGP_NO_OVERLAP") && atoi(getenv("DSDGP_NO_OVERLAP")));
  // ONE side stream per context, shared by its models: a stream per model left the mapping of streams to hardware queues to the
  // order in which models had been created and destroyed (a process that had built several models occasionally ran a later one
  // 15 - 140 % slower, tools/ab_force.py)
  if (!ctx->side) DS_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  m->side = ctx->side;

  for (int l = 0; l < L; ++l) {
    DS_HIP(hipEventCreateWithFlags(&m->ev_bwd[l], hipEventDisableTiming));
  }
  DS_HIP(hipEventCreateWithFlags(&m->ev_side, hipEventDisableTiming));
  DS_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  DS_HIP(hipEventCreateWithFlags(&m->ev_prep_side, hipEventDisableTiming));
  DS_HIP(hipEventCreateWithFlags(&m->ev_z, hipEventDisableTiming));
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

    hipEventDestroy(m->ev_fork); hipEventDestroy(m->ev_prep_side); hipEventDestroy(m->ev_z);
    for (int l = 0; l < m->desc.L; ++l) {
      if (l == 0) bigchol_free(m->big_all);
      bigchol_free(m->L[l].big_k); bigchol_free(m->L[l].big_ngA); bigchol_free(m->L[l].big_ngS); bigchol_free(m->L[l].big_ngT);
    }
    delete m;
  }
  return DSDGP_OK;
}

// Side-stream overlap pays for its cross-stream events (a few microseconds each) only when the kernels are long enough:
// tiny models (cfg 1: 100 rows, M = 50) are launch-latency-bound and run 40 % faster on a single stream.
static bool overlap_on(const dsdgp_model* m, int64_t n, int S) {
  const char* no = getenv("DSDGP_NO_OVERLAP");   // read per call so that a profiler can serialise the kernels
  if (!m->overlap || (no && atoi(no))) return false;
  int mp_max = 0;
  for (int l = 0; l < m->desc.L; ++l) mp_max = std::max(mp_max, (int)m->L[l].dev.Mp);
  return n * S * (int64_t)mp_max >= (int64_t)1 << 20;
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
    hipExtLaunchKernelGGL(k_head, dim3(1 + nprep + r.nblk + gq.nblk, L), dim3(HEAD_THREADS), (uint32_t)lds, ctx->stream, nullptr,
                          head_event ? m->ev_fork : nullptr, 0, (const double*)m->theta, (const LayerDev*)m->layers_dev, m->lik_const,
                          (int64_t)m->desc.off_lik_var, m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? 1 : 0, m->desc.jitter, nprep,
                          keep_kuu ? 1 : 0, m->desc.white ? 1 : 0, getenv("DSDGP_POTRF_TIMING") ? 1 : 0, r, gq);
    DS_HIP(hipGetLastError());
  } else if (!unchanged) {
    hipLaunchKernelGGL(k_prep_kuu, dim3(m->prep_blocks + (keep_kuu ? 0 : m->kuu_blocks), L), dim3(256), 0, ctx->stream, m->theta, m->layers_dev,
                       m->lik_const, m->desc.off_lik_var, m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? 1 : 0, m->desc.jitter,
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
  const int lq = (keep_kuu && m->q_dirty >= 0 && (m->prepared_grad || !with_grad)) ? m->q_dirty : -1;
  if (lq >= 0) {
    LayerState& Sq = m->L[lq];
    DS_TRY(gemm_launch(ctx, Sq.lq, Sq.lq_nf, Sq.lq_tf, st));
    hipLaunchKernelGGL(k_kl_part, dim3(klb, 1), dim3(256), 0, st, m->layers_dev + lq);
  } else {
    DS_TRY(gemm_launch(ctx, m->gp_fwd, m->n_fwd, m->t_fwd, st));
    hipLaunchKernelGGL(k_kl_part, dim3(klb, L), dim3(256), 0, st, m->layers_dev);
  }
  DS_HIP(hipGetLastError());
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
  hipLaunchKernelGGL(k_randn, dim3(nb > 0 ? nb : 1), dim3(256), 0, st ? st : ctx->stream, seed, stream, count, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
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
                (v.Mp > 256 || ((Rin + 15) / 16 > m->force.cs_min_blocks && v.D_out >= m->force.cs_min_dout));
    if (St.gemm) St.c_used = save_l && St.C != nullptr;      // the triangular abar product halves the largest GEMM of the reverse pass
    a.Csave = St.c_used ? St.C : nullptr;
    a.XT1 = save_l ? St.XT1 : nullptr;
    {
      const int64_t nblk = (Rin + 15) / 16;
      a.d_split = chain_d_split(nblk, v.D_out);
      if (last && lik_Y) {      // Gaussian variational expectations + adjoints in this chain's epilogue
        a.lik_Y = lik_Y; a.lik_const = m->lik_const; a.lik_w = lik_w; a.lik_part = m->lik_part;
        a.lik_MB = St.MB; a.lik_VB = St.VB; a.lik_ld = round_up(Rin, 16);
        *lik_nblocks = St.gemm ? layer_gemm_lik_blocks(Rin, v.D_out) : (int)nblk * a.d_split;
      }
    }
    if (St.gemm) DS_TRY(layer_fwd_gemm_launch(ctx, a, v.Mp, v.kern_kind, m->gws));
    else DS_TRY(layer_fwd_sm_launch(ctx, a, v.Mp, v.kern_kind, m->desc.white || wf));
    St.z_used = a.z; St.zs_s = a.zs_s; St.zs_n = a.zs_n; St.zs_d = a.zs_d;
    St.X_used = Xin; St.Rin_used = Rin; St.rep_used = rep; St.ld_used = a.ldA;
    if (St.prop && !last) {
      const int64_t R = (int64_t)S * n, cnt = R * (v.D_out + St.prop);
      hipLaunchKernelGGL(k_concat_prop, dim3((int)std::min<int64_t>(4096, ceil_div(cnt, 256))), dim3(256), 0, ctx->stream, Xin, Rin,
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
      jobsA.push_back(J);
      redA.push_back(RedJob{J.out, v.bigred + (int64_t)j * MM, MM, ns, 0, 0, v.Mp, 16, MMw, Mw, v.Mp});   // mirror at 16-block granularity
    }
    jobsA.push_back(WgradJob{St.A, St.MB, nullptr, out_q, ti, tjq, v.DP16, startA, 0, v.DP16 / 16, 0, 0});      // A mbar^T -> q_mu
    startA += ns * ti * tjq;
    redA.push_back(RedJob{out_q, v.thinq, (int64_t)v.Mp * v.DP16, ns, 0, 0, 0, 0, (int64_t)Mw * v.DP16, 0, 0});
    if (St.mean_grad) {
      // trainable Linear mean function: d loss / d [A ; b] = [X ; 1]^T MB^T  (XT1 is zero-padded to whole 16*NI-row tiles)
      const int tim = ceil_div(v.DinP16 / 16, NI), tjm = ceil_div(v.DP16 / 16, NI);
      jobsA.push_back(WgradJob{St.XT1, St.MB, nullptr, St.part_mean, tim, tjm, v.DP16, startA, 0, v.DP16 / 16, 0, 0});
      startA += ns * tim * tjm;
      redA.push_back(RedJob{St.part_mean, v.meanAB, (int64_t)16 * NI * tim * v.DP16, ns, 0, 0, 0, 0, (int64_t)16 * NI * tim * v.DP16, 0, 0});
    }
    // ---- B jobs: operands from this layer's backward chain (E, GW)
    if (!v.alg_g) {
      WgradJob J{};
      J.P = St.E; J.Q = St.A; J.scale = nullptr;
      J.out = St.part_big;
      J.ti = ti; J.tj = ti; J.ldo = Mw; J.task_start = startB;
      J.sym = 0; J.qrows16 = Mw / 16; J.ns_diag = ns_diag; J.pad = 0;
      startB += ns * ti * ti;
      jobsB.push_back(J);
      redB.push_back(RedJob{J.out, v.bigred, MM, ns, 0, 0, 0, 16, MMw, Mw, v.Mp});
    }
    jobsB.push_back(WgradJob{St.GW, St.XT1, nullptr, out_z, ti, tjz, v.DinP16, startB, 0, v.DinP16 / 16, 0, 0});   // GW [X|1]^T -> Z
    startB += ns * ti * tjz;
    redB.push_back(RedJob{out_z, v.thinz, (int64_t)v.Mp * v.DinP16, ns, 0, 0, 0, 0, (int64_t)Mw * v.DinP16, 0, 0});
    redB.push_back(RedJob{St.hyp_part, v.hyp_red, (int64_t)v.D_in + 2, St.gemm ? layer_gemm_hyp_parts(ld, v.Mp) : (int)sm_hyp_parts(ld, v.Mp, v.D_in), 0, 1, 0, 0, (int64_t)v.D_in + 2, 0, 0});
    // one list [A | B] with cumulative task numbers: one launch per layer
    std::vector<WgradJob> jobs(jobsA);
    for (WgradJob J : jobsB) {
      J.task_start += startA;
      jobs.push_back(J);
    }
    St.njobs = (int)jobs.size();
    St.tot_big = startA + startB;
    St.tot_thin = 0;
    St.red_off = (int)red.size();
    red.insert(red.end(), redA.begin(), redA.end());
    red.insert(red.end(), redB.begin(), redB.end());
    St.red_n = (int)red.size() - St.red_off;
    // diagonal tiles fill only their first ns_diag partial slots: the rest must read as zero under the new plan
    DS_HIP(hipMemsetAsync(St.part_big, 0, (size_t)St.nsplit_big_max * (1 + v.D_out) * MMw * sizeof(double), ctx->stream));
    DS_HIP(hipMemcpyAsync(St.wj, jobs.data(), jobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
  }
  int blocks = 0;
  for (auto& r : red) {
    r.ways = (!r.wide && r.nsplit >= 12) ? 4 : 0;
    const bool even = r.count % 2 == 0 && r.pstride % 2 == 0 && r.in_ld % 2 == 0 && r.out_ld % 2 == 0 && r.sym_tile % 2 == 0 &&
                      ((uintptr_t)r.part & 15) == 0;
    if (r.ways == 4 && even) r.ways = 8;
    const int64_t rows = r.out_ld > 0 ? r.count / r.out_ld : 0;
    const bool tiled = !r.wide && r.sym_n > 0 && r.sym_tile == 16 && r.out_ld > 0 && rows == r.out_ld && rows % 16 == 0 && even;
    if (tiled) r.ways = 16;
    r.blk_start = blocks;
    blocks += r.wide ? (int)r.count
                     : (tiled ? (int)((rows / 16) * (rows / 16 + 1) / 2) : ceil_div(r.count, r.ways == 8 ? 128 : (r.ways == 4 ? 64 : 256)));
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
  m->plan_n = n;
  m->plan_S = S;
  return DSDGP_OK;
}

static int launch_finalize(dsdgp_model* m, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, m->layers_dev, m->desc.L, m->lik_part, m->fin.nblocks, m->fin.w,
                     m->fin.kl_weight, m->lik_const, m->grad,
                     m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : (int64_t)-1, m->fin.with_grad, m->fin.out);
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
  DS_TRY(join_prep(m));   // Ku^-1, S_d (and U, UU for the assembly below) come from the side stream
  const int gfirst = m->desc.white ? 0 : m->grad_first;     // reverse mode stops below this layer (dsdgp_model_set_grad_first_layer)
  // data-parallel buckets: every layer's reduction, products, assembly and hyper-parameter gradients right behind its weight-gradient
  // products, then the caller's collective on that layer's segment of the gradient, on the stream the segment was produced on — the
  // exchange of the upper layers runs under the lower layers' backward chains (dsdgp_model_set_bucket_callback)
  const bool bucketed = m->bucket_fn != nullptr && m->tail_ok && gfirst == 0 && !m->fuse_adam.on;
  const bool pipelined = bucketed || (overlap && m->force.pipe_tail != 0 && !m->desc.white);
  // split-K reduction + P_d T_d / GS_d products of one layer right behind its weight-gradient products (pipelined tail)
  auto layer_tail = [&](LayerState& St, hipStream_t st) -> int {
    hipLaunchKernelGGL(k_reduce_grouped, dim3(St.red_blkn), dim3(256), 0, st, m->rjobs + St.red_off, St.red_n, St.red_blk0);
    DS_HIP(hipGetLastError());
    DS_TRY(gemm_launch(ctx, St.lq + St.lq_nf + St.lq_n1 + St.lq_n2, St.lq_np, St.lq_tp, st));
    if (bucketed) {
      const int l = (int)(&St - m->L);
      const LayerDev* lay1 = m->layers_dev + l;
      hipLaunchKernelGGL(k_asm_rows, dim3(St.dev.M, 1), dim3(256), (size_t)m->mp_max_all * sizeof(double), st, lay1, m->grad, kl_weight,
                         m->mp_max_all);
      FinArgs F{};
      AdamArgs A{};
      hipLaunchKernelGGL(k_tail, dim3(1), dim3(256), 0, st, m->layers_dev, l, 1, m->grad, F, A);
      DS_HIP(hipGetLastError());
      // this layer's parameters are one contiguous segment of theta: [off_Z, next layer's off_Z) (the last layer's ends where the
      // likelihood variance or the vector ends)
      const int64_t lo = St.d.off_Z;
      const int64_t hi = (l + 1 < L) ? m->L[l + 1].d.off_Z
                                     : (m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : m->desc.n_theta);
      m->bucket_fn(m->bucket_user, l, m->grad + lo, hi - lo, (void*)st);
    }
    return DSDGP_OK;
  };
  auto launch_wgrad = [&](LayerState& Sx, hipStream_t st) -> int {
    const int64_t ldx = Sx.ld_used;
    return wgrad_launch(ctx, Sx.wj, Sx.njobs, Sx.tot_big, Sx.ns_big, ldx, ldx, st);
  };
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
    const bool in_chain = !fused && !last && m->force.adj_fuse != 0 && !St.c_used && !St.gemm && sm_adj_fusable(v.Mp, ld / 16, v.D_in, v.D_out);
    if (!fused && !in_chain)
      hipLaunchKernelGGL(k_adj_prep, dim3(ceil_div(ld, 256), std::max(v.DP16, v.DinP16)), dim3(256), 0, ctx->stream, last ? nullptr : St.dF,
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
    if (St.gemm) DS_TRY(layer_bwd_gemm_launch(ctx, b, v.Mp, v.kern_kind, m->gws));
    else DS_TRY(layer_bwd_sm_launch(ctx, b, v.Mp, v.kern_kind, m->desc.white));
    if (!overlap || on_main) {
      DS_TRY(launch_wgrad(St, ctx->stream));
      if (pipelined) DS_TRY(layer_tail(St, ctx->stream));
      continue;
    }
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
    hipLaunchKernelGGL(k_reduce_grouped, dim3(S0.red_blkn), dim3(256), 0, ctx->stream, m->rjobs + S0.red_off, S0.red_n, S0.red_blk0);
    DS_HIP(hipGetLastError());
  }
  if (overlap) {
    // value + likelihood-variance gradient: needs the likelihood partials (main, before ev_bwd) and KL (side).  AFTER the
    // weight-gradient launches: its single workgroup was observed to sit for > 1 ms behind the co-running large-M chain, and
    // everything queued behind it on this stream waited with it
    if (!m->fin.done && !m->tail_ok) DS_TRY(launch_finalize(m, m->side));
    DS_HIP(hipEventRecord(m->ev_side, m->side));
    DS_HIP(hipStreamWaitEvent(ctx->stream, m->ev_side, 0));

  }
  if (!pipelined) {
    if (red_ahead) {
      LayerState& S1 = m->L[1];
      hipLaunchKernelGGL(k_reduce_grouped, dim3(m->red_blocks - S1.red_blk0), dim3(256), 0, ctx->stream, m->rjobs + S1.red_off,
                         m->n_red - S1.red_off, S1.red_blk0);
    } else {
      hipLaunchKernelGGL(k_reduce_grouped, dim3(m->red_blocks), dim3(256), 0, ctx->stream, m->rjobs, m->n_red, 0);
    }
    DS_HIP(hipGetLastError());
    if (m->desc.white) {
      hipLaunchKernelGGL(k_white_lbar, dim3(32, L), dim3(256), 0, ctx->stream, m->layers_dev);
      DS_TRY(gemm_launch(ctx, m->gp_w1, 2 * L, m->t_w1));
      hipLaunchKernelGGL(k_white_phi, dim3(32, L), dim3(256), 0, ctx->stream, m->layers_dev);
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
              m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : (int64_t)-1, m->fin.out, L, 1};
    AdamArgs A{};
    hipLaunchKernelGGL(k_tail, dim3(1), dim3(256), 0, ctx->stream, m->layers_dev, 0, 0, m->grad, F, A);
    DS_HIP(hipGetLastError());
    m->fin.done = true;
    const int64_t lo = m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : m->desc.n_theta;
    const bool tail_scalars = m->fin.out == m->grad + m->desc.n_theta;
    m->bucket_fn(m->bucket_user, L, m->grad + lo, (m->desc.n_theta - lo) + (tail_scalars ? 4 : 0), (void*)ctx->stream);
    if (!tail_scalars) m->bucket_fn(m->bucket_user, L + 1, m->fin.out, 4, (void*)ctx->stream);
    return DSDGP_OK;
  }
  if (m->tail_ok) {
    hipLaunchKernelGGL(k_asm_rows, dim3(m->m_max_all, La), dim3(256), (size_t)m->mp_max_all * sizeof(double), ctx->stream, lay, m->grad,
                       kl_weight, m->mp_max_all);
    FinArgs F{m->lik_part, m->fin.nblocks, m->fin.w, m->fin.kl_weight, m->lik_const,
              m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : (int64_t)-1, m->fin.out, L, 1};
    AdamArgs A{m->theta, m->adam_m, m->adam_v, m->mask, m->desc.n_theta, m->fuse_adam.lr_t, m->fuse_adam.b1, m->fuse_adam.b2,
               m->fuse_adam.eps, (m->fuse_adam.on && gfirst == 0) ? 1 : 0};
    const int nadam = A.on ? (int)std::min<int64_t>(512, ceil_div(m->desc.n_theta, 256)) : 0;
    hipLaunchKernelGGL(k_tail, dim3(La + 1 + nadam), dim3(256), 0, ctx->stream, m->layers_dev, gfirst, La, m->grad, F, A);
    DS_HIP(hipGetLastError());
    m->fin.done = true;
    return DSDGP_OK;
  }
  hipLaunchKernelGGL(k_asm_kbar, dim3(m->kuu_blocks, La), dim3(256), 0, ctx->stream, lay, kl_weight);
  if (m->n_wz) DS_TRY(gemm_launch(ctx, m->gp_wz, m->n_wz, m->t_wz));   // (wm of the layers below gfirst is stale: their WZ is never read)
  if (m->need_hyp_part) hipLaunchKernelGGL(k_asm_hyp_part, dim3(NPART, La), dim3(256), 0, ctx->stream, lay);
  hipLaunchKernelGGL(k_asm_params, dim3(m->asm_blocks + 1, La), dim3(256), 0, ctx->stream, lay, m->grad, kl_weight);
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
  const bool elementwise = m->desc.lik_kind == DSDGP_LIK_GAUSSIAN || m->desc.lik_kind == DSDGP_LIK_BERNOULLI;
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
      hipLaunchKernelGGL(k_lik_gauss, dim3(nblocks), dim3(256), 0, ctx->stream, last.mean, last.var, Y, n, S, DY, m->lik_const,
                         w, m->sample_w, m->lik_part, dm, dv, mbt, vbt, ldt);
    else
      hipLaunchKernelGGL(k_lik_bern, dim3(nblocks), dim3(256), 0, ctx->stream, last.mean, last.var, Y, n, S, DY, w, m->sample_w,
                         m->lik_part, dm, dv, mbt, vbt, ldt);
  } else {
    // MultiClass: Y is (n x 1) labels, the last layer has K = num_classes outputs; ve per (s, i) row -> last.F scratch
    DS_CHECK_ARG(DY == m->desc.num_classes);
    const int64_t R = (int64_t)S * n;
    DS_TRY(multiclass_launch(ctx, last.mean, last.var, Y, n, R, DY, 0, w, last.F, with_grad ? m->lik_dmean : nullptr,
                             with_grad ? m->lik_dvar : nullptr, -1));
    if (m->sample_w) {
      const int64_t cnt = R * DY;
      hipLaunchKernelGGL(k_scale_by_sample, dim3((int)std::min<int64_t>(2048, ceil_div(cnt, 256))), dim3(256), 0, ctx->stream,
                         m->sample_w, n, S, DY, R, last.F, with_grad ? m->lik_dmean : nullptr, with_grad ? m->lik_dvar : nullptr);
    }
    nblocks = ceil_div(R, 256);
    hipLaunchKernelGGL(k_partial_sum, dim3(nblocks), dim3(256), 0, ctx->stream, last.F, R, m->lik_part);
  }
  DS_HIP(hipGetLastError());
  m->fin.nblocks = nblocks; m->fin.w = w; m->fin.kl_weight = kl_weight; m->fin.with_grad = with_grad; m->fin.out = out;
  m->fin.done = false;
  if (with_grad) {
    DS_TRY(backward_layers(m, n, S, kl_weight));
    m->grad_pruned = !m->desc.white && m->grad_first > 0;
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
  const int nb = (int)std::min<int64_t>(1024, ceil_div(n, 256));
  hipLaunchKernelGGL(k_adam, dim3(nb), dim3(256), 0, m->ctx->stream, m->theta, m->grad, m->adam_m, m->adam_v, m->mask, n,
                     lr_t, beta1, beta2, eps);
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
  if (!m->desc.white && m->grad_first > 0) {
    dsdgp_set_error("dsdgp_model_train_step: the reverse pass is restricted to layers >= %d (dsdgp_model_set_grad_first_layer)", m->grad_first);
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
    const int64_t lik_lo = m->desc.lik_kind == DSDGP_LIK_GAUSSIAN ? m->desc.off_lik_var : m->desc.n_theta;
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
  hipLaunchKernelGGL(k_kl_final, dim3(1), dim3(256), 0, m->ctx->stream, m->layers_dev, m->desc.L);
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
  if (St.gemm && a.ldA <= St.ld_max) return layer_fwd_gemm_launch(m->ctx, a, v.Mp, v.kern_kind, m->gws);
  return layer_fwd_sm_launch(m->ctx, a, v.Mp, v.kern_kind, m->desc.white);
}

extern "C" int dsdgp_reparameterize(dsdgp_ctx* ctx, const double* mean, const double* var, const double* z, double jitter,
                                    int64_t count, double* out) {
  DS_CHECK_ARG(ctx && mean && var && z && out && count > 0);
  const int nb = (int)std::min<int64_t>(4096, ceil_div(count, 256));
  hipLaunchKernelGGL(k_reparam, dim3(nb), dim3(256), 0, ctx->stream, mean, var, z, jitter, count, out);
  DS_HIP(hipGetLastError());
  return DSDGP_OK;
}

extern "C" int dsdgp_randn(dsdgp_ctx* ctx, uint64_t seed, uint64_t stream, int64_t count, double* out) {
  DS_CHECK_ARG(ctx && out && count > 0);
  return randn_async(ctx, seed, stream, count, out);
}

// ------------------------------------------------------------------------------------------------------
// Natural-gradient step on one layer's (q_mu, q_sqrt)  — [UPSTREAM] gpflow.training.NatGradOptimizer(gamma)
// (demos/demo_regression_UCI.ipynb:360-366, tests/test_collapsed.py:100).
// Per output d:  Sbar = sym(T^-T Phi(T^T Tbar) T^-1);  theta1 = S^-1 m - gamma (mbar - 2 Sbar m);
//                A = S^-1 + 2 gamma Sbar (= -2 theta2);  S+ = A^-1;  m+ = S+ theta1;  T+ = chol(S+).
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
// A = S^-1 + 2 gamma Sbar with identity pad ; Sbar stored (symmetrised) into ngY
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
    v.ngA[idx] = a;
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
__global__ __launch_bounds__(256) void k_ng_mu(const LayerDev* __restrict__ layers, int l, double* __restrict__ theta) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= v.D_out * M) return;
  const int d = row / M, i = row % M;
  const double* Sp = v.ngSplus + (int64_t)d * Mp * Mp + (int64_t)i * Mp;
  double s = 0.0;
  for (int j = lane; j < M; j += 64) s = fma(Sp[j], v.ngTheta1[d * Mp + j], s);
  s = sum_wave(s);
  if (lane == 0) theta[v.off_q_mu + (int64_t)i * v.D_out + d] = s;
}
__global__ void k_ng_write(const LayerDev* __restrict__ layers, int l, double* __restrict__ theta) {
  const LayerDev v = layers[l];
  const int Mp = v.Mp, M = v.M;
  const int64_t tot = (int64_t)v.D_out * M * M;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx / ((int64_t)M * M)), rem = (int)(idx % ((int64_t)M * M)), i = rem / M, j = rem % M;
    theta[v.off_q_sqrt + idx] = (j <= i) ? v.ngSplus[((int64_t)d * Mp + i) * Mp + j] : 0.0;
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
  hipLaunchKernelGGL(k_ng_prep, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, m->grad);
  DS_HIP(hipGetLastError());
  if (St.big) DS_TRY(bigchol_run(ctx, St.big_ngT));
  else DS_TRY(trtri_launch(ctx, v.ngTI, v.ngTinv, v.Mp, MM, v.D_out));
  DS_TRY(gemm_launch(ctx, St.ng_gp, 2, St.ng_t1));
  hipLaunchKernelGGL(k_ng_phi, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l);
  DS_TRY(gemm_launch(ctx, St.ng_gp + 2, 1, St.ng_t2));
  DS_TRY(gemm_launch(ctx, St.ng_gp + 3, 1, St.ng_t3));
  hipLaunchKernelGGL(k_ng_assemble, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, gamma);
  hipLaunchKernelGGL(k_ng_theta1, dim3(ceil_div(v.D_out * v.M, 4)), dim3(256), 0, ctx->stream, m->layers_dev, l, m->grad,
                     gamma);
  DS_HIP(hipGetLastError());
  if (St.big) DS_TRY(bigchol_run(ctx, St.big_ngA));
  else DS_TRY(potrf_launch(ctx, St.ng_items, v.D_out, v.Mp));
  DS_TRY(gemm_launch(ctx, St.ng_gp + 4, 1, St.ng_t4));
  hipLaunchKernelGGL(k_ng_mu, dim3(ceil_div(v.D_out * v.M, 4)), dim3(256), 0, ctx->stream, m->layers_dev, l, m->theta);
  DS_HIP(hipGetLastError());
  if (St.big) DS_TRY(bigchol_run(ctx, St.big_ngS));
  else DS_TRY(potrf_launch(ctx, St.ng_items + v.D_out, v.D_out, v.Mp));
  hipLaunchKernelGGL(k_ng_write, dim3(nb), dim3(256), 0, ctx->stream, m->layers_dev, l, m->theta);
  DS_HIP(hipGetLastError());
  m->prepared = false;
  m->q_dirty = (m->q_dirty == -1 || m->q_dirty == l) ? l : -2;     // Z and the kernel hyper-parameters are untouched: kuu_valid stays
  if (info) {
    std::vector<double> sc(4 * v.D_out);
    DS_HIP(hipMemcpyAsync(sc.data(), v.ngScal, sc.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
    *info = 0;
    for (int d = 0; d < 2 * v.D_out; ++d)
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
  hipLaunchKernelGGL(k_fullcov_combine, dim3(nb), dim3(256), 0, ctx->stream, Kff, Q, P, n, D, var);
  hipLaunchKernelGGL(k_add_mean_fn, dim3(ceil_div(n * D, 256)), dim3(256), 0, ctx->stream, mean, X, n, v.D_in, D,
                     St.d.mean_kind, St.meanA, St.meanb);
  DS_HIP(hipGetLastError());
  if (v.has_white) {
    // add the White variance on the diagonal of every output's covariance (Kff of a Sum kernel)
    extern __global__ void k_add_diag_dev(double*, int64_t, int, const double*);
    hipLaunchKernelGGL(k_add_diag_dev, dim3(ceil_div(n * D, 256)), dim3(256), 0, ctx->stream, var, n, D, v.hyp + HYP_WVAR);
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
  hipLaunchKernelGGL(k_fullcov_gather, dim3(nb), dim3(256), 0, ctx->stream, var, n, D, S, jitter, Lc);
  DS_HIP(hipGetLastError());
  int info = 0;
  int rc = dsdgp_potrf(ctx, (int)nmat, (int)n, Lc, n, n * n, &info);
  if (rc == DSDGP_OK) {
    hipLaunchKernelGGL(k_fullcov_sample, dim3(ceil_div((int64_t)S * n * D, 256)), dim3(256), 0, ctx->stream, Lc, mean, z, n, D, S,
                       out);
    if (hipGetLastError() != hipSuccess) rc = DSDGP_ERR_HIP;
  }
  hipFreeAsync(Lc, ctx->stream);
  return rc;
}
