// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

int layer_bwd_sm_b(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  if (small) {
    SM_SMALL_CASE(bwd_sm_go, 8, (ctx, a))
    SM_SMALL_CASE(bwd_sm_go, 10, (ctx, a))
    SM_SMALL_CASE(bwd_sm_go, 12, (ctx, a))
    SM_SMALL_CASE(bwd_sm_go, 14, (ctx, a))
    SM_SMALL_CASE(bwd_sm_go, 16, (ctx, a))
  }
  switch (Mp) {
    SM_CASE(bwd_sm_go, 8, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 10, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 12, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 14, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 16, 4, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
