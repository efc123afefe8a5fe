// A 16x16x4 fp64 product out of FOUR v_mfma_f64_4x4x4_4b_f64 (which sustain ~70-75 TFLOP/s against 47-49 for v_mfma_f64_16x16x4_f64,
// tools/mfma_f64_variants.hip): same A / B lane layout as the big instruction (tools/mfma_4x4_probe.hip: A[4 blk + i][k] sits in lane
// 16 k + 4 blk + i, B[k][4 blk + j] in lane 16 k + 4 blk + j, D_blk[i][j] in lane 16 i + 4 blk + j), B rotated by 4 / 8 / 12 lanes
// inside each 16-lane row (DPP row_ror) for the off-diagonal 4x4 blocks; the four accumulators come out in a rotated layout that
// unrot() turns into the big instruction's D layout.  Checks the emulation against the native instruction (both rotation directions)
// and times both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp(double old, double x) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANK, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
#define ROR(n) (0x120 + (n))
// DIR = 0: B for accumulator n is row_ror(b, 16 - 4n); DIR = 1: row_ror(b, 4n)
template <int DIR>
__device__ __forceinline__ d4 mfma_r(double a, double b, d4 acc) {
  const double b1 = DIR ? dpp<ROR(4), 0xf>(b, b) : dpp<ROR(12), 0xf>(b, b);
  const double b2 = dpp<ROR(8), 0xf>(b, b);
  const double b3 = DIR ? dpp<ROR(12), 0xf>(b, b) : dpp<ROR(4), 0xf>(b, b);
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b2, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b3, acc[3], 0, 0, 0);
  return acc;
}
// rotated accumulators -> D layout of v_mfma_f64_16x16x4_f64 (reg t, lane 16 g + c  <->  D[g + 4 t][c]):
// out[t] in bank J (lanes 4J..4J+3 of every row) = acc[n], n = (J - t) mod 4, moved by n banks
template <int DIR>
__device__ __forceinline__ d4 unrot(d4 r) {
  d4 o;
#define MOVE(t, n, J) o[t] = dpp<ROR(DIR ? (16 - 4 * (n)) % 16 : 4 * (n)), (1 << (J))>(o[t], r[n]);
  o[0] = r[0]; o[1] = r[0]; o[2] = r[0]; o[3] = r[0];          // bank t of out[t] = acc[0] as is (n = 0); the other banks are overwritten
  MOVE(0, 1, 1) MOVE(0, 2, 2) MOVE(0, 3, 3)
  MOVE(1, 1, 2) MOVE(1, 2, 3) MOVE(1, 3, 0)
  MOVE(2, 1, 3) MOVE(2, 2, 0) MOVE(2, 3, 1)
  MOVE(3, 1, 0) MOVE(3, 2, 1) MOVE(3, 3, 2)
#undef MOVE
  return o;
}
template <int DIR>
__global__ void k_check(const double* A, const double* B, double* out) {
  const int l = threadIdx.x;
  d4 nat = (d4){0, 0, 0, 0}, em = (d4){0, 0, 0, 0};
  for (int s = 0; s < 8; ++s) {
    nat = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s * 64 + l], B[s * 64 + l], nat, 0, 0, 0);
    em = mfma_r<DIR>(A[s * 64 + l], B[s * 64 + l], em);
  }
  em = unrot<DIR>(em);
  double e = 0;
  for (int t = 0; t < 4; ++t) e = fmax(e, fabs(nat[t] - em[t]));
  out[l] = e;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_time(double* out, int iters) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b + i, acc[i], 0, 0, 0);
      else acc[i] = mfma_r<0>(a, b + i, acc[i]);
    }
  double s = 0;
  for (int i = 0; i < 4; ++i) { d4 r = MODE ? unrot<0>(acc[i]) : acc[i]; s += r[0] + r[1] + r[2] + r[3]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void timeit(int wps) {
  const int blocks = 256 * wps, iters = 20000;
  double* out; hipMalloc(&out, blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_time<MODE><<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
  hipEventRecord(e0); k_time<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s waves/simd=%d: %.2f TFLOP/s\n", MODE ? "4 x 4x4x4 + DPP rotations" : "native 16x16x4           ", wps, 2048.0 * 4 * iters * blocks * 4 / ms * 1e-9);
  hipFree(out);
}
int main() {
  double hA[512], hB[512], hO[64];
  srand(3);
  for (int i = 0; i < 512; ++i) { hA[i] = (rand() % 2000) / 1000.0 - 1; hB[i] = (rand() % 2000) / 1000.0 - 1; }
  double *A, *B, *O; hipMalloc(&A, 4096); hipMalloc(&B, 4096); hipMalloc(&O, 512);
  hipMemcpy(A, hA, 4096, hipMemcpyHostToDevice); hipMemcpy(B, hB, 4096, hipMemcpyHostToDevice);
  k_check<0><<<1, 64>>>(A, B, O); hipMemcpy(hO, O, 512, hipMemcpyDeviceToHost);
  double e0 = 0; for (int i = 0; i < 64; ++i) e0 = fmax(e0, hO[i]);
  k_check<1><<<1, 64>>>(A, B, O); hipMemcpy(hO, O, 512, hipMemcpyDeviceToHost);
  double e1 = 0; for (int i = 0; i < 64; ++i) e1 = fmax(e1, hO[i]);
  printf("max |native - emulated|: DIR=0 %.3e   DIR=1 %.3e\n", e0, e1);
  for (int w : {1, 2, 4}) { timeit<0>(w); timeit<1>(w); }
  return 0;
}
