// The LIBRARY's k_gram (csrc/gram.hip, compiled into this binary with the library's flags) timed like tools/gram_bench.hip:
// separates "kernel" from "call path" (ctypes, descriptor upload) when the two disagree.
#include "../doubly-stochastic-dgp_amd/csrc/gram.hip"
#include <stdarg.h>
void dsdgp_set_error(const char* fmt, ...) {}
int ctx_scratch(dsdgp_ctx*, size_t, void**) { return 0; }
int ctx_upload(dsdgp_ctx*, void*, const void*, size_t) { return 0; }
int main() {
  dsdgp_ctx ctx;
  hipStreamCreate(&ctx.stream);
  const int shapes[3][3] = {{128, 20480, 8}, {1024, 50176, 8}, {512, 40960, 32}};
  for (auto& s : shapes) {
    const int n = s[0], n2 = s[1], D = s[2];
    double *X, *X2, *out, *hyp;
    hipMalloc(&X, 8.0 * n * D); hipMalloc(&X2, 8.0 * n2 * D); hipMalloc(&out, 8.0 * n * n2); hipMalloc(&hyp, 8 * (HYP_ILS + 2 * D + 8));
    std::vector<double> h(HYP_ILS + 2 * D + 8, 1.0), hx((size_t)n2 * D);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 2654435761u) % 1000) * 0.002 - 1.0;
    hipMemcpy(hyp, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(X2, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(X, hx.data(), (size_t)n * D * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) gram_launch(&ctx, 0, X, n, X2, n2, D, hyp, 0.0, 0, out, n2);
    hipEventRecord(a, ctx.stream);
    for (int i = 0; i < 20; ++i) gram_launch(&ctx, 0, X, n, X2, n2, D, hyp, 0.0, 0, out, n2);
    hipEventRecord(b, ctx.stream); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("library k_gram n=%d n2=%d D=%d: %.1f us  %.0f GB/s\n", n, n2, D, ms / 20 * 1e3, 8.0 * n * n2 / (ms / 20 * 1e-3) / 1e9);
  }
  return 0;
}
