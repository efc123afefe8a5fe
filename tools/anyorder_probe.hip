// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950 (kernels of ONE stream running concurrently, no events)?
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
// K1 (ordered) -> K2 (ordered) -> K3 (any order) -> K4 (ordered), every kernel spins `us` microseconds on 64 workgroups and stamps
// s_memrealtime (100 MHz) at start / end.  Expected if the flag works: K3 starts with K2 (after K1 ended), K4 starts after both ended.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(unsigned long long* stamp, int slot, int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * sizeof(unsigned long long));
  hipStream_t s;
  hipStreamCreate(&s);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemsetAsync(d, 0, 64 * sizeof(unsigned long long), s);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, d, 0, 50);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, d, 1, 100);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 2, 60);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, d, 3, 20);
    hipStreamSynchronize(s);
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double t0 = (double)h[0];
    printf("rep %d:", rep);
    for (int k = 0; k < 4; ++k) printf("  K%d %.1f..%.1f us", k + 1, (h[2 * k] - t0) * 0.01, (h[2 * k + 1] - t0) * 0.01);
    printf("\n");
  }
  return 0;
}
