// Bisects which feature of the library's k_gram costs the factor 3 against the bare kernel of gram_bench.hip.
//   F bit0: lengthscale scaling + partial-pass logic, bit1: index clamps / bounds, bit2: opaque rounding asm, bit3: library epilogue
#include <hip/hip_runtime.h>
#include <stdio.h>
#define TI 8
#define TJ 512
#define DC 8
typedef double d2 __attribute__((ext_vector_type(2)));
template <int F>
__global__ __launch_bounds__(256) void k(const double* __restrict__ X, int64_t n, const double* __restrict__ X2, int64_t n2, int D,
                                         const double* __restrict__ hyp, double diag_add, int symmetric, double* __restrict__ out, int64_t ld) {
  __shared__ double xi[TI * DC];
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * TJ + 2 * tid, i0 = (int64_t)blockIdx.y * TI;
  const double s2 = hyp[0];
  const double* ils = hyp + 8;
  const int64_t ja = (F & 2) ? (j0 < n2 ? j0 : n2 - 1) : j0, jb = (F & 2) ? (j0 + 1 < n2 ? j0 + 1 : n2 - 1) : j0 + 1;
  double ra[TI], rb[TI];
#pragma unroll
  for (int ii = 0; ii < TI; ++ii) ra[ii] = rb[ii] = 0.0;
  for (int d0 = 0; d0 < D; d0 += DC) {
    const int dc = (F & 1) ? ((D - d0 < DC) ? D - d0 : DC) : DC;
    if (d0 > 0) __syncthreads();
    if (tid < TI * DC) {
      const int ii = tid / DC, dd = tid % DC;
      double v = 0.0;
      if (F & 1) { if ((!(F & 2) || i0 + ii < n) && dd < dc) v = X[(i0 + ii) * D + d0 + dd] * ils[d0 + dd]; }
      else v = X[(i0 + ii) * D + d0 + dd];
      xi[tid] = v;
    }
    double xa[DC], xb[DC], il[DC];
#pragma unroll
    for (int dd = 0; dd < DC; ++dd) {
      const int dq = d0 + ((F & 1) ? (dd < dc ? dd : dc - 1) : dd);
      il[dd] = (F & 1) ? ils[dq] : 1.0;
      xa[dd] = X2[ja * D + dq];
      xb[dd] = X2[jb * D + dq];
    }
    if (F & 1) {
#pragma unroll
      for (int dd = 0; dd < DC; ++dd) {
        const double sc = dd < dc ? il[dd] : 0.0;
        double ta = xa[dd] * sc, tb = xb[dd] * sc;
        if (F & 4) asm("" : "+v"(ta), "+v"(tb));
        xa[dd] = ta; xb[dd] = tb;
      }
    }
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
  if (F & 8) {
    if (j0 >= n2) return;
    const bool pair = (j0 + 1 < n2) && ((ld & 1) == 0);
#pragma unroll
    for (int ii = 0; ii < TI; ++ii) {
      const int64_t i = i0 + ii;
      ra[ii] = s2 * exp(-0.5 * ra[ii]) + ((symmetric && i == j0) ? diag_add : 0.0);
      rb[ii] = s2 * exp(-0.5 * rb[ii]) + ((symmetric && i == j0 + 1) ? diag_add : 0.0);
    }
#pragma unroll
    for (int ii = 0; ii < TI; ++ii) {
      const int64_t i = i0 + ii;
      if (i < n) {
        double* o = out + i * ld + j0;
        if (pair) __builtin_nontemporal_store((d2){ra[ii], rb[ii]}, reinterpret_cast<d2*>(o));
        else { o[0] = ra[ii]; if (j0 + 1 < n2) o[1] = rb[ii]; }
      }
    }
  } else {
#pragma unroll
    for (int ii = 0; ii < TI; ++ii) {
      d2* o = reinterpret_cast<d2*>(out + (i0 + ii) * ld + j0);
      __builtin_nontemporal_store((d2){s2 * exp(-0.5 * ra[ii]), s2 * exp(-0.5 * rb[ii])}, o);
    }
  }
}
template <int F>
void run(const double* X, int n, const double* X2, int n2, int D, const double* hyp, double* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid(n2 / TJ, n / TI);
  for (int i = 0; i < 3; ++i) k<F><<<grid, 256>>>(X, n, X2, n2, D, hyp, 0.0, 0, out, n2);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) k<F><<<grid, 256>>>(X, n, X2, n2, D, hyp, 0.0, 0, out, n2);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("n=%d n2=%d D=%d F=%2d  %.1f us  %.0f GB/s\n", n, n2, D, F, ms / 20 * 1e3, 8.0 * n * n2 / (ms / 20 * 1e-3) / 1e9);
}
int main() {
  const int shapes[2][3] = {{1024, 50176, 8}, {512, 40960, 32}};
  for (auto& s : shapes) {
    const int n = s[0], n2 = s[1], D = s[2];
    double *X, *X2, *out, *hyp;
    hipMalloc(&X, 8.0 * n * D); hipMalloc(&X2, 8.0 * n2 * D); hipMalloc(&out, 8.0 * n * n2); hipMalloc(&hyp, 8 * (16 + 2 * D));
    double* hx = (double*)malloc(8.0 * n2 * D);
    for (size_t i = 0; i < (size_t)n2 * D; ++i) hx[i] = ((i * 2654435761u) % 1000) * 0.002 - 1.0;
    double hh[128]; for (int i = 0; i < 128; ++i) hh[i] = 1.0;
    hipMemcpy(hyp, hh, 8 * (16 + 2 * D), hipMemcpyHostToDevice);
    hipMemcpy(X2, hx, 8.0 * n2 * D, hipMemcpyHostToDevice); hipMemcpy(X, hx, 8.0 * n * D, hipMemcpyHostToDevice);
    run<0>(X, n, X2, n2, D, hyp, out); run<1>(X, n, X2, n2, D, hyp, out); run<3>(X, n, X2, n2, D, hyp, out);
    run<7>(X, n, X2, n2, D, hyp, out); run<15>(X, n, X2, n2, D, hyp, out); run<8>(X, n, X2, n2, D, hyp, out);
    hipFree(X); hipFree(X2); hipFree(out); hipFree(hyp); free(hx);
  }
  return 0;
}
