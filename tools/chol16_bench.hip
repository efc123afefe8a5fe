// Microbenchmark of the 16-pivot panel loop of the fused head kernel (head_impl.hpp): one wave, rows in lanes, data re-read from LDS
// every repetition.  Variants: 0 = two Newton steps per pivot (linalg.hpp Chol16), 1 = one cubic (Halley) step, 2 = pivots in pairs
// (one rsq latency per two pivots: rsq(a) and rsq(a c - b^2) are independent), 3 = pairs + Halley, 9 = load / store only.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/chol16_bench tools/chol16_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>

template <int L>
__device__ __forceinline__ double bcast_lane(double x) {
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(x), L);
  const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(x), L);
  return __hiloint2double((int)hi, (int)lo);
}
template <bool HALLEY>
__device__ __forceinline__ double rsq_ref(double a) {
  double y = __builtin_amdgcn_rsq(a);
  if (HALLEY) {
    const double e = fma(-(a * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
  }
  y = y * fma(-0.5 * a * y, y, 1.5);
  y = y * fma(-0.5 * a * y, y, 1.5);
  return y;
}
template <int J, int K, bool H> struct Upd1 {
  static __device__ __forceinline__ void run(double (&a)[16], double lij) {
    a[K] -= lij * bcast_lane<K>(lij);
    Upd1<J, K + 1, H>::run(a, lij);
  }
};
template <int J, bool H> struct Upd1<J, 16, H> { static __device__ __forceinline__ void run(double (&)[16], double) {} };
template <int J, bool H> struct C1 {
  static __device__ __forceinline__ void run(double (&a)[16]) {
    const double ajj = bcast_lane<J>(a[J]);
    const double inv = rsq_ref<H>(ajj);
    const double lij = a[J] * inv;
    a[J] = lij;
    Upd1<J, J + 1, H>::run(a, lij);
    C1<J + 1, H>::run(a);
  }
};
template <bool H> struct C1<16, H> { static __device__ __forceinline__ void run(double (&)[16]) {} };

template <int J, int K> struct Upd2 {
  static __device__ __forceinline__ void run(double (&a)[16], double l1, double l2) {
    a[K] = fma(-l2, bcast_lane<K>(l2), fma(-l1, bcast_lane<K>(l1), a[K]));
    Upd2<J, K + 1>::run(a, l1, l2);
  }
};
template <int J> struct Upd2<J, 16> { static __device__ __forceinline__ void run(double (&)[16], double, double) {} };
template <int J, bool H> struct C2 {
  static __device__ __forceinline__ void run(double (&a)[16]) {
    const double p = bcast_lane<J>(a[J]), q = bcast_lane<J + 1>(a[J]), r = bcast_lane<J + 1>(a[J + 1]);
    const double det = fma(p, r, -q * q);
    const double r1 = rsq_ref<H>(p), rd = rsq_ref<H>(det);
    const double t = q * (r1 * r1), r2 = rd * (p * r1);
    const double u = a[J], w = fma(-u, t, a[J + 1]);
    const double l1 = u * r1, l2 = w * r2;
    a[J] = l1; a[J + 1] = l2;
    Upd2<J, J + 2>::run(a, l1, l2);
    C2<J + 2, H>::run(a);
  }
};
template <bool H> struct C2<16, H> { static __device__ __forceinline__ void run(double (&)[16]) {} };

template <int V>
__global__ __launch_bounds__(64) void k_bench(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ clk, int reps) {
  __shared__ double W[64 * 17];
  const int lane = threadIdx.x;
  for (int j = 0; j < 16; ++j) W[lane * 17 + j] = A[lane * 16 + j];
  __syncthreads();
  double acc = 0.0;
  double a[16];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = W[lane * 17 + j];
    if (V == 0) C1<0, false>::run(a);
    if (V == 1) C1<0, true>::run(a);
    if (V == 2) C2<0, false>::run(a);
    if (V == 3) C2<0, true>::run(a);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += a[j];
    __builtin_amdgcn_wave_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int j = 0; j < 16; ++j) out[lane * 16 + j] = a[j];
  out[64 * 16 + lane] = acc;
  if (lane == 0) clk[0] = t1 - t0;
}


typedef double d4 __attribute__((ext_vector_type(4)));
// co-run test: 8 waves; waves in `chain_mask` run the pivot loop, waves in `mfma_mask` a dependent fp64 MFMA loop, the rest idle
__global__ __launch_bounds__(512) void k_corun(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ clk, int reps,
                                               int chain_mask, int mfma_mask) {
  __shared__ double W[64 * 17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) for (int j = 0; j < 16; ++j) W[lane * 17 + j] = A[lane * 16 + j];
  __syncthreads();
  double acc = 0.0;
  double a[16];
  const long long t0 = __builtin_amdgcn_s_memtime();
  if ((chain_mask >> wave) & 1) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = W[lane * 17 + j];
      C1<0, false>::run(a);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += a[j];
      __builtin_amdgcn_wave_barrier();
    }
  } else if ((mfma_mask >> wave) & 1) {
    d4 c = (d4){0, 0, 0, 0};
    double x = W[lane & 15], y = W[17 + (lane & 15)];
    for (int r = 0; r < reps * 40; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0);
    acc = c[0] + c[1] + c[2] + c[3];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[tid] = acc;
  clk[wave] = t1 - t0;
}

int main() {
  const int n = 64;
  std::vector<double> A(n * 16), ref(n * 16), got(n * 16 + 64);
  // rows 0..15: SPD block B = G G^T + 16 I ; rows 16..63: random panel rows
  srand(1);
  double G[16][16];
  for (auto& row : G) for (double& x : row) x = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = (i == j) ? 4.0 : 0.0;
      for (int k = 0; k < 16; ++k) s += G[i][k] * G[j][k];
      A[i * 16 + j] = s;
    }
  for (int i = 16; i < n; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = rand() / (double)RAND_MAX - 0.5;
  // host reference: L of the block, then L_panel = A_panel L^-T
  ref = A;
  for (int j = 0; j < 16; ++j) {
    const double d = sqrt(ref[j * 16 + j]);
    for (int i = 0; i < n; ++i) if (i >= j) ref[i * 16 + j] /= d; 
    for (int k = j + 1; k < 16; ++k)
      for (int i = 0; i < n; ++i) if (i >= k) ref[i * 16 + k] -= ref[i * 16 + j] * ref[k * 16 + j];
  }
  double *dA, *dO; long long* dC;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, got.size() * 8); hipMalloc(&dC, 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  const int reps = 2000;
  long long base = 0;
  for (int v : {9, 0, 1, 2, 3}) {
    for (int it = 0; it < 2; ++it) {
      if (v == 9) hipLaunchKernelGGL(k_bench<9>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
      if (v == 0) hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
      if (v == 1) hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
      if (v == 2) hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
      if (v == 3) hipLaunchKernelGGL(k_bench<3>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
      hipDeviceSynchronize();
    }
    long long c;
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    hipMemcpy(got.data(), dO, got.size() * 8, hipMemcpyDeviceToHost);
    double err = 0.0;
    if (v != 9)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < 16; ++j)
          if (i >= 16 || j <= i) err = fmax(err, fabs(got[i * 16 + j] - ref[i * 16 + j]));
    if (v == 9) base = c;
    printf("variant %d: %.1f memtime ticks per panel (minus load/store loop: %.1f), max abs err %.3e\n", v, (double)c / reps,
           (double)(c - base) / reps, err);
  }
  long long* dC8; hipMalloc(&dC8, 64);
  double* dO8; hipMalloc(&dO8, 512 * 8);
  struct { int cm, mm; const char* what; } cases[] = {{1, 0, "wave 0 alone"}, {0x11, 0, "waves 0 + 4 (same SIMD), both pivot loops"},
    {0x3, 0, "waves 0 + 1 (different SIMDs), both pivot loops"}, {1, 0x10, "wave 0 pivot loop + wave 4 fp64 MFMA loop (same SIMD)"},
    {1, 0x02, "wave 0 pivot loop + wave 1 fp64 MFMA loop (other SIMD)"}, {1, 0xF0, "wave 0 pivot loop + waves 4..7 MFMA"}};
  for (auto& cs : cases) {
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k_corun, dim3(1), dim3(512), 0, 0, dA, dO8, dC8, reps, cs.cm, cs.mm); hipDeviceSynchronize(); }
    long long c8[8];
    hipMemcpy(c8, dC8, 64, hipMemcpyDeviceToHost);
    printf("%-60s: wave 0 %.1f ticks per panel", cs.what, (double)c8[0] / reps);
    for (int w = 1; w < 8; ++w) if (((cs.cm | cs.mm) >> w) & 1) printf(", wave %d %.1f", w, (double)c8[w] / reps);
    printf("\n");
  }
  return 0;
}
