// Which fp64 instruction form sustains the most on gfx950?  v_mfma_f64_16x16x4_f64 (2048 flop), v_mfma_f64_4x4x4_4b_f64 (4 blocks of
// 4x4x4: 512 flop) and plain v_fma_f64 (128 flop per wave-instruction), each with 8 independent accumulators at 1 / 2 / 4 waves
// per SIMD.  The datasheet quotes 78.6 TFLOP/s for both the matrix and the vector path.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, s = 0;
  if (MODE == 0) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (MODE == 1) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) s += acc[i];
  } else {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], b, a);
    for (int i = 0; i < 16; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(int wps, const char* name, double flop_per_inst, int inst_per_iter) {
  const int blocks = 256 * wps, iters = 20000;
  double* out;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)inst_per_iter * iters * blocks * 4;
  printf("%-26s waves/simd=%d  %.2f TFLOP/s  (%.1f cycles per wave-instruction per SIMD @2.4GHz)\n", name, wps, insts * flop_per_inst / ms * 1e-9,
         ms * 1e-3 * 2.4e9 / ((double)inst_per_iter * iters * wps));
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>(w, "v_mfma_f64_16x16x4_f64", 2048.0, 8);
  for (int w : {1, 2, 4, 8}) run<1>(w, "v_mfma_f64_4x4x4_4b_f64", 512.0, 8);
  for (int w : {1, 2, 4, 8}) run<2>(w, "v_fma_f64 (vector)", 128.0, 16);
  return 0;
}
