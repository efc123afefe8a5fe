// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

int layer_bwd_sm_a(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  switch (Mp) {
    SM_CASE(bwd_sm_go, 2, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 3, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 4, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 5, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 6, 4, (ctx, a))
    SM_CASE(bwd_sm_go, 7, 4, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
