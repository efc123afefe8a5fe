// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

// (forward instances only: compiled next to the backward instances of the same range the forward kernels came out with 2 - 5 more VGPRs
// and spill code — 128 + 12 B instead of 126 at M = 256 —, so each direction has its own translation unit since round 5)
int layer_fwd_sm_b(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  if (small) {
    SM_SMALL_CASE(fwd_sm_go, 8, (ctx, a))
    SM_SMALL_CASE(fwd_sm_go, 10, (ctx, a))
    SM_SMALL_CASE(fwd_sm_go, 12, (ctx, a))
    SM_SMALL_CASE(fwd_sm_go, 14, (ctx, a))
    SM_SMALL_CASE(fwd_sm_go, 16, (ctx, a))
  }
  switch (Mp) {
    SM_CASE(fwd_sm_go, 8, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 10, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 12, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 14, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 16, 4, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
