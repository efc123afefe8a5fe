// Dispatch of the SVGP_Layer chain kernels (layer_sm_impl.hpp).  The instances are compiled in three parts
// (layer_sm_{fwd,bwd}_{a,b,c}.hip: padded inducing counts 32..112, 128..256, 320..1024, each direction on its own) so that they
// build in parallel.
#include <stdlib.h>

#include "layer.hpp"

#define XCH 64
int layer_fwd_sm_a(dsdgp_ctx*, const LayerFwdArgs&, int, int, int, int);
int layer_fwd_sm_b(dsdgp_ctx*, const LayerFwdArgs&, int, int, int, int);
int layer_fwd_sm_c(dsdgp_ctx*, const LayerFwdArgs&, int, int, int, int);
int layer_bwd_sm_a(dsdgp_ctx*, const LayerBwdArgs&, int, int, int, int);
int layer_bwd_sm_b(dsdgp_ctx*, const LayerBwdArgs&, int, int, int, int);
int layer_bwd_sm_c(dsdgp_ctx*, const LayerBwdArgs&, int, int, int, int);

// waves per 16-row block.  Launches with few row blocks (the N-row first layer: 63 blocks at N = 1000) are latency-bound —
// one dependent MFMA chain per wave on a mostly idle chip — so they take twice the waves per block (half the chain each).
#define SM_SMALL_BLOCKS 160
#define SM_BWD_RESIDENT_8W 768
static inline bool sm_small(int Mp, int64_t nblk, int D_in, bool bwd) {
  // measured (tools/ab_kernels.py): M = 256 — the 8-wave form (two row blocks per wave instead of four) wins at every size
  // (-7 % on both chains); M = 128 — it wins for the backward chain while one round of it holds the launch (see below), for the
  // forward chain only on small launches (the 4-wave forward instance has the early-mean specialisation)
  // Mp = 160 .. 224 follow the M = 256 rule (tools/bench_padding.py: with 4 waves M = 224 ran no faster than M = 256 with 8)
  // round 5, backward chain at M = 128: the 8-wave instance holds 77 VGPRs = three workgroups per CU = 768 resident row blocks; a
  // launch with more runs in two rounds, the second one latency-bound on a third of the chip (config 2: 1250 blocks, 482 late
  // starters, 139 us).  The 4-wave instance with the paired d-loop (96 VGPRs, five workgroups per CU) keeps 1280 blocks resident with
  // the MFMA pipe saturated in the d-loop: 124 us.  (DSDGP_SM_BWD_LIM128: A/B aid.)
  static const int64_t bwd_lim128 = getenv("DSDGP_SM_BWD_LIM128") ? atoll(getenv("DSDGP_SM_BWD_LIM128")) : SM_BWD_RESIDENT_8W;
  const int64_t lim = Mp > 128 ? ((int64_t)1 << 40) : (bwd ? bwd_lim128 : SM_SMALL_BLOCKS);
  return Mp >= 128 && Mp <= 256 && nblk <= lim && D_in <= XCH;
}
// waves per row block of the instance that a launch of this shape takes
static inline int sm_nw(int Mp, int64_t nblk, int D_in, bool bwd) {
  return (Mp > 512 || (Mp == 512 && D_in <= XCH && !bwd)) ? 16 : (Mp > 256 ? 8 : (sm_small(Mp, nblk, D_in, bwd) ? 8 : 4));
}
// rows of hyp_part the backward chain writes (one per wave)
int64_t sm_hyp_parts(int64_t ld, int Mp, int D_in) { return (int64_t)sm_nw(Mp, ceil_div(ld, 16), D_in, true) * ceil_div(ld, 16); }
// the adjoint prologue (LayerBwdArgs::up_dF) parks 2 x 16 x D_out partial sums in the chain's reduction scratch
int sm_adj_fusable(int Mp, int64_t nblk, int D_in, int D_out) {
  if (D_in > XCH) return 0;
  const int NW = sm_nw(Mp, nblk, D_in, true);
  const int xch = D_in < XCH ? D_in : XCH;
  return NW * 16 * xch >= 2 * 16 * D_out;
}
// padded inducing counts whose backward chain has a Csave instance (layer_sm_impl.hpp: sm_cs_inst)
int sm_cs_built(int Mp) { return Mp > 256 || Mp == 32 || Mp == 64 || Mp == 128 || Mp == 256; }

int layer_fwd_sm_launch(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white) {
  const int small = sm_small(Mp, ceil_div(a.Rin, 16), a.D_in, false);
  if (Mp < 128) return layer_fwd_sm_a(ctx, a, Mp, kern_kind, white, small);
  if (Mp <= 256) return layer_fwd_sm_b(ctx, a, Mp, kern_kind, white, small);
  return layer_fwd_sm_c(ctx, a, Mp, kern_kind, white, small);
}
int layer_bwd_sm_launch(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white) {
  const int small = sm_small(Mp, ceil_div(a.ldA, 16), a.D_in, true);
  if (Mp < 128) return layer_bwd_sm_a(ctx, a, Mp, kern_kind, white, small);
  if (Mp <= 256) return layer_bwd_sm_b(ctx, a, Mp, kern_kind, white, small);
  return layer_bwd_sm_c(ctx, a, Mp, kern_kind, white, small);
}
