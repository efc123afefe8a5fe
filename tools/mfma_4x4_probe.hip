// Decodes the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 and whether the CBSZ / ABID block broadcast applies to it:
// for every pair (pa, pb) of one-hot A / B lanes, which output lanes become non-zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int CBSZ, int ABID>
__global__ void k(unsigned long long* out) {
  const int l = threadIdx.x;
  for (int pa = 0; pa < 64; ++pa)
    for (int pb = 0; pb < 64; ++pb) {
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == pa ? 1.0 : 0.0, l == pb ? 1.0 : 0.0, 0.0, CBSZ, ABID, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0) out[pa * 64 + pb] = m;
    }
}
static unsigned long long H[4096];
template <int CBSZ, int ABID>
void probe() {
  unsigned long long* d;
  hipMalloc(&d, sizeof(H));
  k<CBSZ, ABID><<<1, 64>>>(d);
  hipMemcpy(H, d, sizeof(H), hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d: for A lane pa: the B lanes it meets -> output lane(s)\n", CBSZ, ABID);
  for (int pa = 0; pa < 64; ++pa) {
    if (!(pa < 8 || pa == 16 || pa == 17 || pa == 20 || pa == 32 || pa == 48 || pa == 63)) continue;
    printf("  pa=%2d:", pa);
    for (int pb = 0; pb < 64; ++pb) {
      const unsigned long long m = H[pa * 64 + pb];
      if (!m) continue;
      printf(" pb=%d->", pb);
      for (int o = 0; o < 64; ++o) if (m >> o & 1) printf("%d,", o);
    }
    printf("\n");
  }
  hipFree(d);
}
int main() {
  probe<0, 0>(); probe<2, 0>(); probe<2, 1>();
  return 0;
}
