// Which XCD does workgroup w of a 1-D launch run on, and which workgroups share an s_memtime base?  (hipcc --offload-arch=gfx950 -O2)
// Prints the XCC_ID (hwreg 20, bits 3:0), the CU / SE ids (HW_ID) and s_memtime / s_memrealtime of the first workgroups of a launch
// that fills the chip, and the spread of (s_memtime - k * s_memrealtime) per XCD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void k_probe(unsigned long long* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[blockIdx.x * 4 + 0] = xcc;
    out[blockIdx.x * 4 + 1] = hwid;
    out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

int main() {
  const int n = 2048;
  unsigned long long* d;
  hipMalloc(&d, n * 4 * sizeof(unsigned long long));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe, dim3(n), dim3(256), 0, 0, d, 200);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(n * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  printf("w : xcc  (first 48 workgroups)\n");
  for (int w = 0; w < 48; ++w) printf("%d:%llu ", w, h[w * 4] & 15);
  printf("\n");
  int match = 0;
  for (int w = 0; w < n; ++w) match += ((h[w * 4] & 15) == (unsigned)(w % 8));
  printf("workgroups with xcc == w %% 8: %d of %d\n", match, n);
  // per XCC: spread of memtime - 24 * memrealtime (2.4 GHz vs 100 MHz) among its workgroups
  for (int x = 0; x < 8; ++x) {
    double lo = 1e300, hi = -1e300; int cnt = 0;
    for (int w = 0; w < n; ++w) if ((int)(h[w * 4] & 15) == x) {
      const double v = (double)h[w * 4 + 2] - 24.0 * (double)h[w * 4 + 3];
      lo = std::min(lo, v); hi = std::max(hi, v); ++cnt;
    }
    printf("xcc %d: %d workgroups, memtime - 24 realtime spread %.3g clocks\n", x, cnt, hi - lo);
  }
  return 0;
}
