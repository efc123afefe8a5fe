// What bounds the dense d-loop of the backward chain (abar += sum_d 2 vbar_d S_d a, Mp = 128, 8 waves per 16-row block)?
// The loop is rebuilt here in both MFMA forms with its three ingredients switchable: the weight stream from L2 (global loads),
// the activation reads from LDS, the MFMAs.  Same launch shape as config 2's D_out = 8 layer: 1250 workgroups x 512 threads,
// 35 KB LDS, 8 outputs.   build: hipcc --offload-arch=gfx950 -O3 -I doubly-stochastic-dgp_amd/csrc -I include tools/chain_loop_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"

// ---- the fast fp64 form of a 16x16x4 product ---------------------------------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 sustains ~70-75 TFLOP/s where v_mfma_f64_16x16x4_f64 is issue-limited at 47-49 (tools/mfma_f64_variants.hip,
// profiles/r02_mfma_f64_variants.txt).  It computes four independent 4x4x4 blocks with the SAME operand lane layout as the big
// instruction (tools/mfma_4x4_probe.hip): A[4 blk + i][k] in lane 16 k + 4 blk + i, B[k][4 blk + j] in lane 16 k + 4 blk + j,
// D_blk[i][j] in lane 16 i + 4 blk + j — i.e. only the diagonal 4x4 blocks of the 16x16 product.  Feeding B rotated by 4 n columns
// (lane c gets column (c + 4 n) & 15: an LDS re-read at the rotated column) gives the n-th block diagonal, so FOUR small MFMAs on
// four accumulators equal one big one (bit-exact, tools/mfma_emul_test.hip).  The accumulators are in a ROTATED layout:
//   acc[n], lane (g, c)  =  D[4 (c >> 2) + g][4 (((c >> 2) + n) & 3) + (c & 3)]
// unrot() converts to the big instruction's D layout (reg t, lane (g, c) = D[g + 4 t][c]) with 12 bank-masked DPP moves.
// The form only pays where the B re-reads are shared by several row blocks or the loop is otherwise MFMA-issue-bound.
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ d4 mfma_rot(double a, double b0, double b1, double b2, double b3, d4 acc) {
  acc[0] = mfma4(a, b0, acc[0]);
  acc[1] = mfma4(a, b1, acc[1]);
  acc[2] = mfma4(a, b2, acc[2]);
  acc[3] = mfma4(a, b3, acc[3]);
  return acc;
}
__device__ __forceinline__ int rot_col(int c, int n) { return 4 * (((c >> 2) + n) & 3) + (c & 3); }
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_mov(double old, double x) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
#define DSDGP_ROW_ROR(n) (0x120 + (n))
// rotated accumulators -> D layout of the big instruction: out[t], bank J (lanes 4 J .. 4 J + 3 of every 16-lane row) = acc[(J - t) & 3]
// moved up by that many banks
__device__ __forceinline__ d4 unrot(d4 r) {
  d4 o;
  o[0] = r[0]; o[1] = r[0]; o[2] = r[0]; o[3] = r[0];          // bank t of out[t] is acc[0] as it stands; the other banks are overwritten
#define DSDGP_MOVE(t, n, J) o[t] = dpp_mov<DSDGP_ROW_ROR(4 * (n)), (1 << (J))>(o[t], r[n]);
  DSDGP_MOVE(0, 1, 1) DSDGP_MOVE(0, 2, 2) DSDGP_MOVE(0, 3, 3)
  DSDGP_MOVE(1, 1, 2) DSDGP_MOVE(1, 2, 3) DSDGP_MOVE(1, 3, 0)
  DSDGP_MOVE(2, 1, 3) DSDGP_MOVE(2, 2, 0) DSDGP_MOVE(2, 3, 1)
  DSDGP_MOVE(3, 1, 0) DSDGP_MOVE(3, 2, 1) DSDGP_MOVE(3, 3, 2)
#undef DSDGP_MOVE
  return o;
}
// per-lane values s[n] attached to the elements of a rotated tile -> lane (g, c): the sum over the four row blocks of column c
// (rows 4 J + g, J = 0..3); sum_groups() of the result is the full 16-row column sum
__device__ __forceinline__ double rot_colsum(d4 s) {
  double x = s[0];
  x += dpp_mov<DSDGP_ROW_ROR(4), 0xf>(s[1], s[1]);
  x += dpp_mov<DSDGP_ROW_ROR(8), 0xf>(s[2], s[2]);
  x += dpp_mov<DSDGP_ROW_ROR(12), 0xf>(s[3], s[3]);
  return x;
}


constexpr int Mp = 128, MPB = 8, NWV = 8;

// ---- the committed form: one accumulator, v_mfma_f64_16x16x4_f64, compiler-scheduled (#pragma unroll 4 over k-blocks)
template <bool GL, bool LD, int COLS>
__global__ __launch_bounds__(512) void k_big(const double* __restrict__ S, double* out, int Dout) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  for (int i = tid; i < Mp * 16 * COLS; i += 512) smem[i] = 1e-3 * (i % 97);
  __syncthreads();
  d4 acc[COLS];
  for (int t = 0; t < COLS; ++t) acc[t] = (d4){0, 0, 0, 0};
  const double w0 = 1.0 + 1e-6 * lane;
  for (int d = 0; d < Dout; ++d) {
    const double* __restrict__ W = S + (int64_t)d * Mp * Mp + 16 * wave + c + (int64_t)g * Mp;
    const double vd2 = 1.0 + d;
#pragma unroll 4
    for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double w = GL ? W[(int64_t)(16 * kb + 4 * s) * Mp] : w0;
#pragma unroll
        for (int t = 0; t < COLS; ++t) {
          const double b = LD ? smem[t * Mp * 16 + (16 * kb + 4 * s + g) * 16 + c] * vd2 : vd2;
          acc[t] = mfma_f64(w, b, acc[t]);
        }
      }
  }
  double r = 0;
  for (int t = 0; t < COLS; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[(size_t)blockIdx.x * 512 + tid] = r;
}

// ---- the fast form: four v_mfma_f64_4x4x4_4b_f64 per product, hand-pipelined (layer_sm_impl.hpp: FastD)
struct WF { double w[16]; };
struct BT { double b[8]; };
template <int OFF>
__device__ __forceinline__ void lds_issue(BT& t, const unsigned (&la)[4]) {
  asm volatile(
      "ds_read_b64 %0, %8 offset:%12\n\tds_read_b64 %1, %9 offset:%12\n\tds_read_b64 %2, %10 offset:%12\n\tds_read_b64 %3, %11 offset:%12\n\t"
      "ds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %9 offset:%13\n\tds_read_b64 %6, %10 offset:%13\n\tds_read_b64 %7, %11 offset:%13"
      : "=&v"(t.b[0]), "=&v"(t.b[1]), "=&v"(t.b[2]), "=&v"(t.b[3]), "=&v"(t.b[4]), "=&v"(t.b[5]), "=&v"(t.b[6]), "=&v"(t.b[7])
      : "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "n"(OFF), "n"(OFF + 4 * 128));
}
__device__ __forceinline__ void lds_wait(BT& t) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.b[0]), "+v"(t.b[1]), "+v"(t.b[2]), "+v"(t.b[3]), "+v"(t.b[4]), "+v"(t.b[5]), "+v"(t.b[6]), "+v"(t.b[7]));
}
template <bool GL, bool LD, int GIC, int I, int GIN>
__device__ __forceinline__ void batch(const WF& fc, WF& fn, const double* Sn, BT& bc, BT& bn, const unsigned (&la)[4], unsigned boff, d4& y) {
  constexpr int noff = (I + 1 < 8) ? (16 * (GIC * 4 + (I + 1) / 2) + 8 * ((I + 1) & 1)) * 128 : (16 * (GIN * 4)) * 128;
  if (LD) lds_issue<noff>(bn, la);
  __builtin_amdgcn_sched_barrier(0);
  if (GL) {
#pragma unroll
    for (int j = 2 * I; j < 2 * I + 2; ++j) {
      typedef const char __attribute__((address_space(1)))* gbytes;
      typedef const double __attribute__((address_space(1)))* gdbl;
      const gbytes base = (gbytes)(Sn + (16 * (GIN * 4 + (j >> 2)) + 4 * (j & 3)) * Mp);
      fn.w[j] = *(gdbl)(base + boff);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = 4 * (I / 2) + 2 * (I & 1) + h;
    y = mfma_rot(fc.w[j], bc.b[4 * h], bc.b[4 * h + 1], bc.b[4 * h + 2], bc.b[4 * h + 3], y);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (LD) lds_wait(bn);
}
template <bool GL, bool LD, int GIC, int GIN>
__device__ __forceinline__ void step(const WF& fc, WF& fn, const double* Sn, BT& bc, BT& bn, const unsigned (&la)[4], unsigned boff, d4& y) {
  batch<GL, LD, GIC, 0, GIN>(fc, fn, Sn, bc, bn, la, boff, y);
  batch<GL, LD, GIC, 1, GIN>(fc, fn, Sn, bn, bc, la, boff, y);
  batch<GL, LD, GIC, 2, GIN>(fc, fn, Sn, bc, bn, la, boff, y);
  batch<GL, LD, GIC, 3, GIN>(fc, fn, Sn, bn, bc, la, boff, y);
  batch<GL, LD, GIC, 4, GIN>(fc, fn, Sn, bc, bn, la, boff, y);
  batch<GL, LD, GIC, 5, GIN>(fc, fn, Sn, bn, bc, la, boff, y);
  batch<GL, LD, GIC, 6, GIN>(fc, fn, Sn, bc, bn, la, boff, y);
  batch<GL, LD, GIC, 7, GIN>(fc, fn, Sn, bn, bc, la, boff, y);
}
template <bool GL, bool LD>
__global__ __launch_bounds__(512, 4) void k_fast(const double* __restrict__ S, double* out, int Dout) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  for (int i = tid; i < Mp * 16; i += 512) smem[i] = 1e-3 * (i % 97);
  __syncthreads();
  unsigned boff = (unsigned)(g * Mp + c + 16 * wave) * 8u;
  unsigned la[4];
  for (int n = 0; n < 4; ++n) la[n] = (unsigned)(uintptr_t)(lptr)(smem + g * 16 + ((c + 4 * n) & 15));
  WF fa, fb;
  BT bc, bn;
  for (int j = 0; j < 16; ++j) fa.w[j] = fb.w[j] = 1.0 + 1e-6 * (lane + j);
  for (int j = 0; j < 8; ++j) bc.b[j] = bn.b[j] = 1.0 + j;
  if (LD) { lds_issue<0>(bc, la); lds_wait(bc); }
  d4 racc = (d4){0, 0, 0, 0};
#pragma unroll 1
  for (int d = 0; d < Dout; ++d) {
    const double* Sd = S + (int64_t)d * Mp * Mp;
    const double* Sn = S + (int64_t)(d + 1 < Dout ? d + 1 : d) * Mp * Mp;
    asm volatile("" : "+s"(Sd), "+s"(Sn));
    asm volatile("" : "+v"(boff));
    d4 y = (d4){0, 0, 0, 0};
    step<GL, LD, 0, 1>(fa, fb, Sd, bc, bn, la, boff, y);
    step<GL, LD, 1, 0>(fb, fa, Sn, bc, bn, la, boff, y);
    racc += (1.0 + d) * y;
    __builtin_amdgcn_sched_barrier(0);
  }
  racc = unrot(racc);
  out[(size_t)blockIdx.x * 512 + tid] = racc[0] + racc[1] + racc[2] + racc[3];
}


// ---- variants of the committed form (what is the 30 % between the full loop and the MFMAs alone made of?)
//   VAR 1: no per-product v_mul_f64 (y_d accumulated, abar += vd2 * y_d once per output)
//   VAR 2: VAR 1 + two accumulators (even / odd k-steps)
//   VAR 3: VAR 1 + explicit register prefetch: the 16 weights of the NEXT four k-blocks are loaded before the 16 MFMAs of these
//   VAR 4: VAR 3 + the 16 activation values of the next four k-blocks prefetched too
template <int VAR>
__global__ __launch_bounds__(512) void k_var(const double* __restrict__ S, double* out, int Dout) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  for (int i = tid; i < Mp * 16; i += 512) smem[i] = 1e-3 * (i % 97);
  __syncthreads();
  d4 acc = (d4){0, 0, 0, 0};
  const double* __restrict__ act = smem + g * 16 + c;
  if (VAR <= 2) {
    for (int d = 0; d < Dout; ++d) {
      const double* __restrict__ W = S + (int64_t)d * Mp * Mp + 16 * wave + c + (int64_t)g * Mp;
      d4 y0 = (d4){0, 0, 0, 0}, y1 = (d4){0, 0, 0, 0};
#pragma unroll 4
      for (int kb = 0; kb < MPB; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double w = W[(int64_t)(16 * kb + 4 * s) * Mp];
          const double b = act[(16 * kb + 4 * s) * 16];
          if (VAR == 2 && (s & 1)) y1 = mfma_f64(w, b, y1); else y0 = mfma_f64(w, b, y0);
        }
      acc += (1.0 + d) * (y0 + y1);
    }
  } else {
    double wn[16], bn[16];
    {
      const double* __restrict__ W = S + 16 * wave + c + (int64_t)g * Mp;
#pragma unroll
      for (int j = 0; j < 16; ++j) wn[j] = W[(int64_t)(4 * j) * Mp];
      if (VAR == 4)
#pragma unroll
        for (int j = 0; j < 16; ++j) bn[j] = act[(4 * j) * 16];
    }
    const int nh = 2 * Dout;                 // half-outputs: four k-blocks each
    d4 y = (d4){0, 0, 0, 0};
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      double wc[16], bc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { wc[j] = wn[j]; bc[j] = bn[j]; }
      const int hn = h + 1 < nh ? h + 1 : h;
      const double* __restrict__ W = S + (int64_t)(hn >> 1) * Mp * Mp + (int64_t)(64 * (hn & 1)) * Mp + 16 * wave + c + (int64_t)g * Mp;
#pragma unroll
      for (int j = 0; j < 16; ++j) wn[j] = W[(int64_t)(4 * j) * Mp];
      if (VAR == 4) {
#pragma unroll
        for (int j = 0; j < 16; ++j) bn[j] = act[(64 * (hn & 1) + 4 * j) * 16];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double b = (VAR == 4) ? bc[j] : act[(64 * (h & 1) + 4 * j) * 16];
        y = mfma_f64(wc[j], b, y);
      }
      if (h & 1) { acc += (1.0 + (h >> 1)) * y; y = (d4){0, 0, 0, 0}; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[(size_t)blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <typename F>
void timeit(const char* name, F launch, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.1f us   %6.1f TFLOP/s\n", name, ms / reps * 1e3, flops / (ms / reps * 1e-3) * 1e-12);
}
int main() {
  const int Dout = 8, blocks = 1250;
  double *S, *out;
  hipMalloc(&S, (size_t)Dout * Mp * Mp * 8); hipMalloc(&out, (size_t)blocks * 512 * 8);
  double* h = (double*)malloc((size_t)Dout * Mp * Mp * 8);
  for (size_t i = 0; i < (size_t)Dout * Mp * Mp; ++i) h[i] = 1e-3 * (i % 101);
  hipMemcpy(S, h, (size_t)Dout * Mp * Mp * 8, hipMemcpyHostToDevice);
  const double fl = 2.0 * Mp * Mp * 16 * Dout * blocks;
  const size_t lds = 34944, lds2 = 34944 + Mp * 16 * 8;
#define BIG(GL, LD) timeit("16x16x4  global=" #GL " lds=" #LD, [&] { k_big<GL, LD, 1><<<blocks, 512, lds>>>(S, out, Dout); }, fl)
#define FAST(GL, LD) timeit("4x4x4x4  global=" #GL " lds=" #LD, [&] { k_fast<GL, LD><<<blocks, 512, lds>>>(S, out, Dout); }, fl)
  BIG(true, true); BIG(false, true); BIG(true, false); BIG(false, false);
  FAST(true, true); FAST(false, true); FAST(true, false); FAST(false, false);
  // 32 data rows per workgroup (two activation tiles per weight fragment): half the weight stream per flop
  timeit("16x16x4  32 rows per workgroup, global=true lds=true", [&] { k_big<true, true, 2><<<blocks / 2, 512, lds2>>>(S, out, Dout); }, fl);
  timeit("16x16x4  32 rows per workgroup, global=false lds=true", [&] { k_big<false, true, 2><<<blocks / 2, 512, lds2>>>(S, out, Dout); }, fl);
#define VARI(V, TXT) timeit("16x16x4  " TXT, [&] { k_var<V><<<blocks, 512, lds>>>(S, out, Dout); }, fl)
  VARI(1, "no v_mul (y_d accumulated)"); VARI(2, "no v_mul, two accumulators"); VARI(3, "no v_mul, weights prefetched one group ahead");
  VARI(4, "no v_mul, weights and activations prefetched");
  return 0;
}
