// Separates "cycles per MFMA" from "clock under load" for v_mfma_f64_16x16x4_f64: each wave brackets its MFMA loop with
// s_memtime (shader-clock ticks) while the host times the same launch with HIP events; short (~0.1 ms) and long (~0.3 s) runs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* ticks, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;
  double* out;
  unsigned long long* ticks;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  hipMalloc(&ticks, blocks * 4 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, ticks, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, ticks, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(blocks * 4 * sizeof(unsigned long long));
  hipMemcpy(h, ticks, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks * 4; ++i) avg += (double)h[i];
  avg /= blocks * 4;
  const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * 4;
  const double mfma_per_simd = (double)NACC * iters * waves_per_simd;
  printf("acc=%d waves/simd=%d iters=%6d  %.3f ms  %.2f TFLOP/s | ticks/wave %.0f -> %.1f ticks per MFMA per SIMD | tick rate %.1f MHz (ticks/wall)\n",
         NACC, waves_per_simd, iters, ms, flops / ms * 1e-9, avg, avg / mfma_per_simd, avg / (ms * 1e3));
  free(h);
  hipFree(out);
  hipFree(ticks);
}
int main(int argc, char** argv) {
  if (argc > 1) {   // long mode: ~N seconds of the saturating configuration, for power / clock sampling from outside
    const int secs = atoi(argv[1]);
    for (int i = 0; i < secs * 14; ++i) run<8>(2, 100000);
    return 0;
  }
  run<8>(2, 100); run<8>(2, 100); run<8>(2, 400); run<8>(2, 2000); run<8>(2, 20000); run<8>(2, 100000);
  run<4>(4, 200); run<4>(4, 20000);
  run<8>(1, 200); run<8>(1, 20000);
  return 0;
}
