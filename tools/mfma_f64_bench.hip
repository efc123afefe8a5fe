// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (roofline denominator check for DESIGN.md).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  double* out;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * 4;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * NACC * waves_per_simd);
  printf("acc=%d waves/simd=%d  %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", NACC, waves_per_simd, flops / ms * 1e-9, cyc);
  hipFree(out);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<4>(2); run<8>(2);
  run<4>(3); run<4>(4); run<2>(4); run<8>(4); run<1>(8); run<2>(8); run<4>(8);
  return 0;
}
