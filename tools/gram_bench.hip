// Where does the materialised Gram build (k_gram, csrc/gram.hip) lose its bandwidth?  Same tile shape and store pattern, three
// bodies: MODE 0 = stores only (write roof of the pattern), 1 = distances without exp, 2 = full RBF value.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast tools/gram_bench.hip -o tools/bin/gram_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define TI 8
#define TJ 512
#define DC 8
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(const double* __restrict__ X, int64_t n, const double* __restrict__ X2, int64_t n2, int D,
                                         double* __restrict__ out, int64_t ld) {
  __shared__ double xi[TI * DC];
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * TJ + 2 * tid, i0 = (int64_t)blockIdx.y * TI;
  double ra[TI], rb[TI];
#pragma unroll
  for (int ii = 0; ii < TI; ++ii) ra[ii] = rb[ii] = 0.0;
  if (MODE >= 1) {
    for (int d0 = 0; d0 < D; d0 += DC) {
      if (d0 > 0) __syncthreads();
      if (tid < TI * DC) xi[tid] = X[(i0 + tid / DC) * D + d0 + tid % DC];
      double xa[DC], xb[DC];
#pragma unroll
      for (int dd = 0; dd < DC; ++dd) { xa[dd] = X2[j0 * D + d0 + dd]; xb[dd] = X2[(j0 + 1) * D + d0 + dd]; }
      __syncthreads();
#pragma unroll
      for (int dd = 0; dd < DC; ++dd)
#pragma unroll
        for (int ii = 0; ii < TI; ++ii) {
          const double z = xi[ii * DC + dd];
          const double da = z - xa[dd], db = z - xb[dd];
          ra[ii] = fma(da, da, ra[ii]); rb[ii] = fma(db, db, rb[ii]);
        }
    }
  }
#pragma unroll
  for (int ii = 0; ii < TI; ++ii) {
    double ka = ra[ii], kb = rb[ii];
    if (MODE == 2) { ka = exp(-0.5 * ka); kb = exp(-0.5 * kb); }
    d2* o = reinterpret_cast<d2*>(out + (i0 + ii) * ld + j0);
    if (NT) __builtin_nontemporal_store((d2){ka, kb}, o); else *o = (d2){ka, kb};
  }
}
template <int MODE, bool NT>
void run(const double* X, int n, const double* X2, int n2, int D, double* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid(n2 / TJ, n / TI);
  for (int i = 0; i < 3; ++i) k<MODE, NT><<<grid, 256>>>(X, n, X2, n2, D, out, n2);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k<MODE, NT><<<grid, 256>>>(X, n, X2, n2, D, out, n2);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = 8.0 * n * n2;
  printf("n=%d n2=%d D=%d mode=%d nt=%d  %.1f us  %.0f GB/s\n", n, n2, D, MODE, (int)NT, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}
int main() {
  const int shapes[3][3] = {{128, 20480, 8}, {1024, 50176, 8}, {512, 40960, 32}};
  for (auto& s : shapes) {
    const int n = s[0], n2 = s[1], D = s[2];
    double *X, *X2, *out;
    hipMalloc(&X, 8.0 * n * D); hipMalloc(&X2, 8.0 * n2 * D); hipMalloc(&out, 8.0 * n * n2);
    hipMemset(X, 0, 8.0 * n * D); hipMemset(X2, 0, 8.0 * n2 * D);
    run<0, false>(X, n, X2, n2, D, out); run<0, true>(X, n, X2, n2, D, out);
    run<1, true>(X, n, X2, n2, D, out); run<2, true>(X, n, X2, n2, D, out); run<2, false>(X, n, X2, n2, D, out);
    hipFree(X); hipFree(X2); hipFree(out);
  }
  return 0;
}
