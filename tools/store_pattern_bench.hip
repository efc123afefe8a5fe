// Which store pattern does a 1024 x 50 000 fp64 write stream want on MI355X?  (reference for k_gram_mfma: DESIGN 5.5)
//   A: a wave instruction writes 1 KB of ONE row (lane * 16 B), non-temporal          B: the Gram kernel's pattern — lane (g, c) writes
//   16 B at (row g + 4 t, columns 2c, 2c + 1): four 256-byte runs in four rows per instruction      C: B with plain stores
//   D: 64 lanes x 16 B over TWO rows (512-byte runs)        E: A with plain stores
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define N 1024
#define N2 50000
template <int MODE>
__global__ __launch_bounds__(256) void k(double* __restrict__ out, int64_t ld, int jtiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int64_t i0 = (int64_t)blockIdx.y * 64 + 16 * wave;
  const int64_t jbase = (int64_t)blockIdx.x * jtiles * 32;
  const d2 v = {1.0 + lane, 2.0};
  if (MODE == 0 || MODE == 4) {
    // the wave's 16 rows x (32 jtiles) columns as runs of 128 columns (1 KB) per instruction
    const int64_t cols = (int64_t)jtiles * 32;
    for (int r = 0; r < 16; ++r)
      for (int64_t j = 2 * lane; j < cols; j += 128) {
        if (jbase + j + 1 < N2) {
          d2* o = reinterpret_cast<d2*>(out + (i0 + r) * ld + jbase + j);
          if (MODE == 0) __builtin_nontemporal_store(v, o); else *o = v;
        }
      }
  } else if (MODE == 3) {
    const int64_t cols = (int64_t)jtiles * 32;
    const int half = lane >> 5, l32 = lane & 31;
    for (int r = 0; r < 16; r += 2)
      for (int64_t j = 2 * l32; j < cols; j += 64) {
        if (jbase + j + 1 < N2) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(out + (i0 + r + half) * ld + jbase + j));
      }
  } else {
    for (int jt = 0; jt < jtiles; ++jt) {
      const int64_t j0 = jbase + 32 * jt + 2 * c;
      if (j0 + 1 >= N2) break;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        d2* o = reinterpret_cast<d2*>(out + (i0 + g + 4 * t) * ld + j0);
        if (MODE == 1) __builtin_nontemporal_store(v, o); else *o = v;
      }
    }
  }
}
template <int MODE>
static void run(const char* name, double* out, int jtiles) {
  dim3 grid((N2 / 32 + jtiles - 1) / jtiles, N / 64);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, out, (int64_t)N2, jtiles);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, out, (int64_t)N2, jtiles);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / 20;
  printf("%-44s jtiles %2d: %7.1f us  %5.2f TB/s\n", name, jtiles, us, 8.0 * N * N2 / us / 1e6);
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * N * N2);
  for (int jt : {4, 12, 16}) {
    run<0>("A  1 KB runs of one row, non-temporal", out, jt);
    run<4>("E  1 KB runs of one row, plain", out, jt);
    run<3>("D  512-byte runs in two rows, non-temporal", out, jt);
    run<1>("B  256-byte runs in four rows (gram), nt", out, jt);
    run<2>("C  256-byte runs in four rows (gram), plain", out, jt);
  }
  return 0;
}
