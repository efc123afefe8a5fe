// Split-M chain kernels, instances for one range of padded inducing counts (see layer_sm_impl.hpp; layer_sm.hip dispatches).
#include "layer_sm_impl.hpp"

// (forward instances only: compiled next to the backward instances of the same range the forward kernels came out with 2 - 5 more VGPRs
// and spill code — 128 + 12 B instead of 126 at M = 256 —, so each direction has its own translation unit since round 5)
int layer_fwd_sm_c(dsdgp_ctx* ctx, const LayerFwdArgs& a, int Mp, int kern_kind, int white, int small) {
  const bool wide = a.D_in > XCH;
  if (Mp == 512 && !wide) {      // forward chain only: 16 waves (four per SIMD at one workgroup per CU) measured -8 % there and +2 % on the
                                 // backward chain (config 4); the wide instance's staging would not fit the LDS with 16 waves
    switch (Mp) {
      SM_CASE(fwd_sm_go, 32, 16, (ctx, a))
      default: break;
    }
  }
  switch (Mp) {
    SM_CASE(fwd_sm_go, 20, 8, (ctx, a))
    SM_CASE(fwd_sm_go, 24, 8, (ctx, a))
    SM_CASE(fwd_sm_go, 28, 8, (ctx, a))
    SM_CASE(fwd_sm_go, 32, 8, (ctx, a))
    SM_CASE(fwd_sm_go, 40, 16, (ctx, a))
    SM_CASE(fwd_sm_go, 48, 16, (ctx, a))
    SM_CASE(fwd_sm_go, 56, 16, (ctx, a))
    SM_CASE(fwd_sm_go, 64, 16, (ctx, a))
    default: break;
  }
  SM_NOT_BUILT
}
