// SVGP_Layer hot path on gfx950: argument blocks of the chain kernels and of the split-K weight-gradient products.
#pragma once
#include "common.hpp"

// SVGP_Layer.conditional_ND + reparameterize (layers.py:178-219, utils.py:40-41), fused.
struct LayerFwdArgs {
  const double* X;   // (Rin x D_in) row-major inputs (for layer 0: the N minibatch rows, shared by all S samples)
  int64_t Rin;
  int32_t rep;       // S for layer 0 (tf.tile of dgp.py:63 is never materialised), 1 otherwise
  int32_t D_in, D_out, M;
  const double* Zp;     // (Mp x D_in), rows >= M zero
  const double* Zs;     // (Mp x D_in)  Z / lengthscale (split-M kernels read it through L1/L2 instead of staging it in LDS)
  const double* hyp;    // see common.hpp HYP_*
  const double* LinvT;  // (Mp x Mp)  Lu^{-T}
  const double* Linv;   // (Mp x Mp)  Lu^{-1}
  const double* Tp;     // (D_out x Mp x Mp) lower-triangular q_sqrt, zero padded
  const double* TpT;    // (D_out x Mp x Mp) its transpose (split-M kernels read every weight as rows [i][k])
  const double* qmu;    // (Mp x D_out)
  int32_t qmu_ld;       // row stride of qmu (0: D_out)
  int32_t mean_kind;
  const double* mean_A; // (D_in x D_out) of the Linear mean function
  const double* mean_b; // (D_out) bias of the Linear mean function or NULL
  const double* z;      // N(0,1) draws, element (s,i,d) at z[s*zs_s + i*zs_n + d*zs_d]; NULL -> F not produced
  int64_t zs_s, zs_n, zs_d;
  int64_t n_inner;      // minibatch rows per sample: output row o = s*n_inner + i
  double jitter;
  double* F;            // (rep*Rin x D_out) samples or NULL
  double* mean;         // (rep*Rin x D_out) or NULL
  double* var;          // (rep*Rin x D_out) or NULL
  double* Asave;        // (Mp x ldA): A = Ku^{-1} Kuf (white: Lu^{-1} Kuf), kept for the backward pass, or NULL
  double* Csave;        // [row block][D_out][Mp slots][16 rows]: c_d = q_sqrt_d^T a of every output in LDS slot order, kept for the
                        // backward pass (split-M kernels), or NULL
  int64_t ldA;
  int32_t d_split;      // split-M kernels: gridDim.y workgroups share a row block, each taking D_out/d_split outputs
  double* XT1;          // split-M kernels, training: (DinP16 x ldA) [X^T ; 1] for the Z-gradient product, or NULL
  unsigned long long* phase_clk;   // debug aid (DSDGP_FWD_TIMING): [workgroup][8] s_memrealtime (100 MHz) stamps of the forward chain's phases, or NULL
  // last layer of a training step with the Gaussian likelihood: [UPSTREAM] Gaussian.variational_expectations (dgp.py:89-90) and its
  // adjoints in this chain's epilogue (k_lik_gauss's job: one launch less between the forward and the reverse pass).  lik_Y NULL: off.
  const double* lik_Y;             // (n_inner x D_out) targets
  const double* lik_const;         // [0] = likelihood variance
  double lik_w;                    // data_scale / S
  double* lik_part;                // [workgroup][2]: sum of the variational expectations, sum of d / d variance
  double *lik_MB, *lik_VB;         // (D_out x lik_ld) transposed adjoints w.r.t. mean / var of this layer (the backward chain reads them)
  int64_t lik_ld;
};

struct LayerBwdArgs {
  const double* X;
  int64_t Rin;
  int32_t D_in, D_out, M, DP4;
  const double* Zp;
  const double* Zs;     // (Mp x D_in)  Z / lengthscale
  const double* hyp;
  const double* Kinv;   // (Mp x Mp)
  const double* Linv;   // (Mp x Mp)   (white path)
  const double* LinvT;  // (Mp x Mp)   (white path, split-M kernels)
  const double* Sd;     // (D_out x Mp x Mp)  q_sqrt q_sqrt^T
  const double* qmu4;   // (Mp x DP4)
  const double* Asave;
  const double* Csave;  // [row block][D_out][Mp][16] from the forward chain: abar += q_sqrt_d (2 vbar_d c_d) — a triangular product — replaces
                        // the dense 2 vbar_d S_d a (split-M kernels; NULL selects the S_d form)
  const double* Tp;     // (D_out x Mp x Mp) q_sqrt, and its transpose (Csave form)
  const double* TpT;
  int64_t ldA;
  const double* VB;     // (D_out.. x ldA)  d loss / d var, transposed, zero beyond Rin
  const double* MB;     // (>=DP4 x ldA)    d loss / d mean, transposed, zero padded
  double* E;            // (Mp x ldA)  out: so that d loss/d Ku (data) = -sym(E A^T)
  double* GW;           // (Mp x ldA)  out: Kuf-bar * dk/dr2
  double* dX;           // (Rin x D_in) out or NULL
  int32_t mean_kind;
  const double* mean_A;
  double* hyp_part;     // [nwaves][D_in + 2]: sum kbar*k, sum vbar, lengthscale partials
  // split-M kernels: when the PREVIOUS layer is an inner layer (one output row per input row) its transposed upstream
  // adjoints are written here directly (k_adj_prep's job): MBp[d][r] = dX[r][prop+d], VBp[d][r] = dX * z / (2 sqrt(var+jitter))
  double* MBp;          // (>= Dp x ldA) or NULL (then dX is written instead)
  double* VBp;
  const double* zp;     // the previous layer's N(0,1) draws, element (row, d) at zp[(row / n_inner) * zp_s + (row % n_inner) * zp_n + d * zp_d]
  int64_t zp_s, zp_n, zp_d, n_inner;
  const double* varp;   // the previous layer's variances (Rin x Dp)
  int32_t Dp, prop;     // previous layer's D_out, input_prop_dim
  double jitter;
  unsigned long long* phase_clk;   // debug aid (DSDGP_BWD_TIMING): [workgroup][8] s_memrealtime (100 MHz) stamps of the backward chain's phases, or NULL
  // split-M kernels, small launches: gridDim.y = d_split workgroups share a row block, each takes D_out / d_split outputs of the d-loop
  // and leaves its partial abar tile (+ its share of sum_d vbar_d) in `part` ([row block][d_split][Mp * 16 + 16]); the workgroup that
  // arrives last (ticket from part_cnt[row block], reset by it) adds the partials in split order — a fixed order, whoever is
  // last — and runs the rest of the chain
  int32_t d_split;
  double* part;
  int* part_cnt;
  // a layer with `up_rep` output rows per input row (the first layer: S samples share mean / var) gets its transposed upstream adjoints
  // from the adjoint of the next layer's input right here, in the chain's prologue (k_adj_prep's job, one launch less on the
  // critical path):  MB[d][r] = sum_s dF[s Rin + r][off + d],  VB[d][r] = sum_s dF[..] z[..] / (2 sqrt(var[r][d] + jitter)); written
  // to MB / VB (the weight-gradient products read them too) by the workgroup that owns the row block.  NULL: MB / VB are ready.
  const double* up_dF;
  int32_t up_rep, up_ld, up_off;
  const double* up_z;          // this layer's draws, element (row, d) at up_z[(row / n_inner) * up_zs + (row % n_inner) * up_zn + d * up_zd]
  int64_t up_zs, up_zn, up_zd, up_n_inner;
  const double* up_var;        // this layer's variances (Rin x D_out)
  double up_jitter;
  double *MBw, *VBw;           // = MB / VB, writable
};
// whether the backward chain of this shape can take the adjoint prologue (its LDS reduction scratch holds 2 x 16 x D_out partials)
int sm_adj_fusable(int Mp, int64_t nblk, int D_in, int D_out);

// out[split][i][j] = sum_{r in split} P[i][r] * scale[r] * Q[j][r]
struct WgradJob {
  const double* P;
  const double* Q;
  const double* scale;  // or NULL
  double* out;          // [nsplit][rowsP][ldo]
  int32_t ti, tj;       // tiles in i / j (tile = 16*NI x 16*NJ; the last j tile may be partial, see qrows16)
  int32_t ldo;
  int32_t task_start;
  int32_t sym;          // P == Q (symmetric result): only tiles with tile_j <= tile_i are computed
  int32_t qrows16;      // rows of Q / 16 (= columns of the result / 16)
  int32_t ns_diag;      // sym: K splits of the (cheaper, 10/16) diagonal tiles; their tasks follow the off-diagonal ones
  int32_t pad;
  // In-launch split-K reduction (round 6): with `fin` set the LAST of a tile's splits to arrive (ticket counter `tick[tile]`, zero between
  // launches) adds the tile's partials in split order and writes rows < fin_rows, columns < fin_cols of the result (leading dimension
  // fin_ld) — and, for sym, the mirror tile / the uncomputed upper blocks of a diagonal tile — so that no reduction launch follows the
  // products.  One split: the tile goes from registers to `fin`, `out` is not touched.  NULL: partials only (k_reduce_grouped adds them).
  double* fin;
  int32_t* tick;
  int32_t fin_ld, fin_rows, fin_cols, pad2;
};

// jobs_dev: device copy of `njobs` jobs with task_start filled (64 x 64 tiles, one workgroup per (job, split, tile) task)
int wgrad_launch(dsdgp_ctx* ctx, const WgradJob* jobs_dev, int njobs, int total_tasks, int nsplit, int64_t ld, int64_t Rp,
                 hipStream_t stream = nullptr);
// chain kernels (layer_sm.hip): the NW waves of a workgroup cooperate on one block of 16 rows
int layer_fwd_sm_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white);
int layer_bwd_sm_launch(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white);
// the last layer of a training step as one launch (layer_last.hip): forward chain + Gaussian likelihood + reverse mode of the same row block
int layer_last_built(int Mp, int D_in, int D_out);
int layer_last_waves(int Mp);
int layer_last_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, const LayerBwdArgs& b, int Mp, int kern_kind, int hyp_rows);
int sm_cs_built(int Mp);     // the split-M backward chain has a Csave instance for this padded inducing count
int64_t sm_hyp_parts(int64_t ld, int Mp, int D_in);   // number of hyp_part rows the split-M backward writes for ld padded rows

// ------------------------------------------------------------------------------------------------------
// GEMM-formulated layer passes for LARGE inducing counts (layer_gemm.hip; Mp >= 512 by default).
// The fused chains keep one block of 16 data rows per workgroup, so every weight element fetched from L2 / Infinity Cache feeds 16
// columns of one MFMA — at Mp >= 512 the weights of a layer (2 + D_out triangular M x M matrices, 10 - 40 MB) no longer sit in L1 / L2
// and the chains run at 0.16 - 0.31 of the fp64 MFMA peak (profiles/r03_kernel_stats_cfg{4,5}.md).  Here each of the layer's products
// is ONE LDS-tiled MFMA GEMM over the whole row range (128 x 128 output tiles: every weight and every activation element is used 128
// times per fetch), with the per-row reductions (|a1|^2, |c_d|^2) taken in the GEMM epilogue; the intermediates (Kuf, a1, a, c_d:
// Mp x R doubles each, 51 MB at config 5) live in HBM / Infinity Cache between the launches.
// Same argument blocks and same outputs as the chains (A, C, mean / var / F, E, GW, hyp_part, dX / MBp / VBp), so the
// weight-gradient products, the assembly and the optimiser are untouched.  C (c_d for the backward pass) is laid out [d][Mp][ldA].
// ------------------------------------------------------------------------------------------------------
struct GemmLayerWs {
  double* T1;        // (Mp_max x ld_max)  forward: Kuf            backward: Ku^-1 abar
  double* T2;        // (Mp_max x ld_max)  forward: a1 = Lu^-1 Kuf backward: abar
  double* Pb;        // (GL_MAX_GROUPS + 1) x (Mp_max x ld_max): partial abar of the d-groups + the mean part
  double* colsq;     // (1 + D_out_max) x tiles_m_max x ld_max: per-tile-row column sums of squares (|a1|^2, |c_d|^2)
  double* MUT;       // (DP16_max x ld_max)  q_mu^T a
  double* qmuT;      // (DP16_max x Mp_max)  q_mu^T, zero padded
  double* ZZ;        // (Mp_max x nzz_max)   [Z/l | (Z/l)^2 | 1]
  double* OUTt;      // (nzz_max x ld_max)   ZZ^T GW
  double* svar;      // per block of the element-wise backward kernel: sum kbar k
  int64_t pb_doubles; // capacity of Pb
};
#define GL_MAX_GROUPS 4
// doubles of each GemmLayerWs array for a model whose gemm-path layers have at most these extents
struct GemmLayerExtents { int64_t Mp, ld, D_out, D_in; };
int layer_gemm_hyp_parts(int64_t ld, int Mp);
int layer_gemm_lik_blocks(int64_t Rin, int D_out);
int layer_fwd_gemm_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white, const GemmLayerWs& ws);
int layer_bwd_gemm_launch(dsdgp_ctx* ctx, const LayerBwdArgs& b, int Mp, int kern_kind, int white, const GemmLayerWs& ws);
