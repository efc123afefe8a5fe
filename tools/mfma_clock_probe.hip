// Does the shader clock drop under fp64 MFMA load?  rocm-smi reports 2.39 GHz during a saturating v_mfma_f64_16x16x4_f64 loop
// (profiles/r02_mfma_power_samples.txt), yet the loop sustains only 47-49 TFLOP/s.  Here every kernel reads s_memtime (shader-clock
// counter) and s_memrealtime (constant 100 MHz) at its start and end: ticks / real time = the clock the CUs actually ran at.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k_loop(double* out, unsigned long long* clk, int iters) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  d4 acc[4];
  double s[4];
  int iv = threadIdx.x;
  for (int i = 0; i < 4; ++i) { acc[i] = (d4){0, 0, 0, 0}; s[i] = 1.0 + i; }
  const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 1) {
        acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][1], 0, 0, 0);
        acc[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][2], 0, 0, 0);
        acc[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][3], 0, 0, 0);
      }
      if (MODE == 2) { s[i] = fma(s[i], a, b); s[i] = fma(s[i], b, a); s[i] = fma(s[i], a, b); s[i] = fma(s[i], b, a); }
      if (MODE == 3) { iv = iv * 1664525 + 1013904223; iv ^= iv >> 7; iv = iv * 22695477 + 1; iv ^= iv >> 5; }
    }
  }
  double r = iv;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, int iters, double flop_per_iter_per_wave) {
  const int blocks = 256 * 4;      // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  double* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_loop<MODE><<<blocks, 256>>>(out, clk, iters / 50); hipDeviceSynchronize();
  hipEventRecord(e0); k_loop<MODE><<<blocks, 256>>>(out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[2 * 1024];
  hipMemcpy(h, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double t = 0, r = 0;
  for (int i = 0; i < blocks; ++i) { t += h[2 * i]; r += h[2 * i + 1]; }
  t /= blocks; r /= blocks;
  const double tf = flop_per_iter_per_wave * iters * blocks * 4 / (ms * 1e-3) * 1e-12;
  printf("%-28s %8.2f ms   s_memtime %.4g ticks = %7.1f MHz by the events, %7.1f MHz by s_memrealtime (100 MHz)   %6.1f TFLOP/s\n", name, ms, t,
         t / (ms * 1e3), t / (r / 100.0), tf);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<3>("integer VALU", 400000, 0);
    run<2>("v_fma_f64", 400000, 4 * 4 * 2 * 64.0);
    run<1>("v_mfma_f64_4x4x4_4b_f64", 400000, 16 * 512.0);
    run<0>("v_mfma_f64_16x16x4_f64", 100000, 4 * 2048.0);
  }
  return 0;
}
