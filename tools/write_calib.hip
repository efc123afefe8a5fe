// Calibration of the WRITE_SIZE / FETCH_SIZE counters on KNOWN byte counts in the chain kernels' store pattern (the
// backward chain's WRITE_SIZE is above the bytes its source stores).  Kernels, each moving exactly 128 x 20 000 doubles = 20.48 MB:
//   w_mmajor : the chains' GW / Asave pattern — a workgroup owns 16 data rows; a store instruction writes four 128-byte runs in four
//              rows of the M-major matrix (lane (g, c): row 16 ib + g + 4 t, column r0 + c)
//   w_linear : contiguous 16-byte stores
//   r_only   : reads such a matrix (16-byte loads), writes 8 bytes per workgroup
//   rw       : reads one matrix, writes another in the M-major pattern (what a backward chain does with A and GW)
// Launch order (stream order, one queue): w_mmajor(A) r_only(A) w_linear(B) r_only(B) w_mmajor(A) w_mmajor(B) r_only(A) rw(A->B) r_only(B)
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace ... ; rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   (tools/gpu_r6_k.sh prints per dispatch)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define MP 128
#define R 20000
__global__ __launch_bounds__(256) void w_mmajor(double* __restrict__ out, double v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int64_t r = (int64_t)blockIdx.x * 16 + c;
  for (int q = 0; q < 2; ++q) {
    const int ib = wave + 4 * q;
    for (int t = 0; t < 4; ++t) out[(int64_t)(16 * ib + g + 4 * t) * R + r] = v + t;
  }
}
__global__ __launch_bounds__(256) void w_linear(double* __restrict__ out, double v) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i < (int64_t)MP * R) *reinterpret_cast<d2*>(out + i) = (d2){v, v + 1};
}
__global__ __launch_bounds__(256) void r_only(const double* __restrict__ in, double* __restrict__ sink) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  double s = 0;
  if (i < (int64_t)MP * R) {
    const d2 x = *reinterpret_cast<const d2*>(in + i);
    s = x[0] + x[1];
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) sink[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void rw(const double* __restrict__ in, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int64_t r = (int64_t)blockIdx.x * 16 + c;
  double x[2][4];
  for (int q = 0; q < 2; ++q)
    for (int t = 0; t < 4; ++t) x[q][t] = in[(int64_t)(16 * (wave + 4 * q) + g + 4 * t) * R + r];
  for (int q = 0; q < 2; ++q)
    for (int t = 0; t < 4; ++t) out[(int64_t)(16 * (wave + 4 * q) + g + 4 * t) * R + r] = 2.0 * x[q][t];
}
int main() {
  double *A, *B, *S;
  const size_t bytes = (size_t)MP * R * sizeof(double);
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&S, 1 << 20);
  hipMemset(A, 0, bytes); hipMemset(B, 0, bytes);
  hipDeviceSynchronize();
  const int nlin = (MP * R / 2 + 255) / 256;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(w_mmajor, dim3(R / 16), dim3(256), 0, 0, A, 1.0 + rep);
    hipLaunchKernelGGL(r_only, dim3(nlin), dim3(256), 0, 0, A, S);
    hipLaunchKernelGGL(w_linear, dim3(nlin), dim3(256), 0, 0, B, 2.0 + rep);
    hipLaunchKernelGGL(r_only, dim3(nlin), dim3(256), 0, 0, B, S);
    hipLaunchKernelGGL(w_mmajor, dim3(R / 16), dim3(256), 0, 0, A, 3.0 + rep);
    hipLaunchKernelGGL(w_mmajor, dim3(R / 16), dim3(256), 0, 0, B, 4.0 + rep);
    hipLaunchKernelGGL(r_only, dim3(nlin), dim3(256), 0, 0, A, S);
    hipLaunchKernelGGL(rw, dim3(R / 16), dim3(256), 0, 0, A, B);
    hipLaunchKernelGGL(r_only, dim3(nlin), dim3(256), 0, 0, B, S);
    hipDeviceSynchronize();
  }
  printf("done: every kernel moves %.2f MB\n", bytes / 1e6);
  return 0;
}
