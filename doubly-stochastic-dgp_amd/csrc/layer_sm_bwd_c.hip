// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

int layer_bwd_sm_c(dsdgp_ctx* ctx, const LayerBwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  switch (Mp) {
    SM_CASE(bwd_sm_go, 20, 8, (ctx, a))
    SM_CASE(bwd_sm_go, 24, 8, (ctx, a))
    SM_CASE(bwd_sm_go, 28, 8, (ctx, a))
    SM_CASE(bwd_sm_go, 32, 8, (ctx, a))
    SM_CASE(bwd_sm_go, 40, 16, (ctx, a))
    SM_CASE(bwd_sm_go, 48, 16, (ctx, a))
    SM_CASE(bwd_sm_go, 56, 16, (ctx, a))
    SM_CASE(bwd_sm_go, 64, 16, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
