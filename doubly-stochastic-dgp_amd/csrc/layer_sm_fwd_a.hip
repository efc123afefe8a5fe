// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

// (forward instances only: compiled next to the backward instances of the same range the forward kernels came out with 2 - 5 more VGPRs
// and spill code — 128 + 12 B instead of 126 at M = 256 —, so each direction has its own translation unit since round 5)
int layer_fwd_sm_a(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  switch (Mp) {
    SM_CASE(fwd_sm_go, 2, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 3, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 4, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 5, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 6, 4, (ctx, a))
    SM_CASE(fwd_sm_go, 7, 4, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
