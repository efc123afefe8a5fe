// Context, error reporting, scratch and HIP-event profiling of libdsdgp.
#include <dlfcn.h>
#include <stdarg.h>

#include "common.hpp"

static thread_local char g_err[1024] = "";

void dsdgp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dsdgp_last_error(void) { return g_err; }
extern "C" int dsdgp_version(void) { return 100; }
extern "C" int dsdgp_sizeof_layer_desc(void) { return (int)sizeof(dsdgp_layer_desc); }
extern "C" int dsdgp_sizeof_model_desc(void) { return (int)sizeof(dsdgp_model_desc); }

extern "C" int dsdgp_ctx_create(dsdgp_ctx** out, int device, void* stream) {
  DS_CHECK_ARG(out != nullptr);
  int ndev = 0;
  DS_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    dsdgp_set_error("no HIP device visible: libdsdgp has no CPU fallback");
    return DSDGP_ERR_HIP;
  }
  DS_CHECK_ARG(device >= 0 && device < ndev);
  DS_HIP(hipSetDevice(device));
  dsdgp_ctx* c = new dsdgp_ctx();
  c->device = device;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    DS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  *out = c;
  return DSDGP_OK;
}

extern "C" int dsdgp_ctx_destroy(dsdgp_ctx* ctx) {
  if (!ctx) return DSDGP_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->prof)
    for (auto& ev : kv.second.pending) {
      hipEventDestroy(ev.first);
      hipEventDestroy(ev.second);
    }
  if (ctx->potrf_plan && ctx->potrf_plan_free) ctx->potrf_plan_free(ctx->potrf_plan);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->pin) hipHostFree(ctx->pin);
  if (ctx->side) {
    hipStreamSynchronize(ctx->side);
    hipStreamDestroy(ctx->side);
  }
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return DSDGP_OK;
}

extern "C" int dsdgp_sync(dsdgp_ctx* ctx) {
  DS_CHECK_ARG(ctx != nullptr);
  DS_HIP(hipStreamSynchronize(ctx->stream));
  return DSDGP_OK;
}

int ctx_scratch(dsdgp_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    DS_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) DS_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    const size_t want = round_up(bytes, 1 << 20);
    DS_HIP(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return DSDGP_OK;
}

int ctx_upload(dsdgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  const size_t RING = 1 << 20;
  if (!ctx->pin) {
    DS_HIP(hipHostMalloc((void**)&ctx->pin, RING, hipHostMallocDefault));
    ctx->pin_bytes = RING;
    ctx->pin_off = 0;
  }
  const size_t need = (size_t)round_up((int64_t)bytes, 64);
  if (need > ctx->pin_bytes) {        // larger than the ring: a plain (synchronised) copy
    DS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    DS_HIP(hipStreamSynchronize(ctx->stream));
    return DSDGP_OK;
  }
  if (ctx->pin_off + need > ctx->pin_bytes) {
    DS_HIP(hipStreamSynchronize(ctx->stream));     // wrap-around: once per MiB of descriptors
    ctx->pin_off = 0;
  }
  char* slot = ctx->pin + ctx->pin_off;
  ctx->pin_off += need;
  memcpy(slot, src, bytes);
  DS_HIP(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
  return DSDGP_OK;
}

extern "C" int dsdgp_prof_enable(dsdgp_ctx* ctx, int on) {
  DS_CHECK_ARG(ctx != nullptr);
  ctx->prof_on = on;
  return DSDGP_OK;
}

std::atomic<long long> g_dsdgp_launches{0};
extern "C" int64_t dsdgp_launch_count(void) { return (int64_t)g_dsdgp_launches.load(std::memory_order_relaxed); }

extern "C" int dsdgp_prof_read(dsdgp_ctx* ctx, const char* name, double* total_ms, int64_t* launches, int reset) {
  DS_CHECK_ARG(ctx && name);
  DS_HIP(hipStreamSynchronize(ctx->stream));
  ProfSlot& s = ctx->prof[name];
  for (auto& ev : s.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      s.ms += ms;
      s.launches += 1;
    }
    hipEventDestroy(ev.first);
    hipEventDestroy(ev.second);
  }
  s.pending.clear();
  if (total_ms) *total_ms = s.ms;
  if (launches) *launches = s.launches;
  if (reset) {
    s.ms = 0.0;
    s.launches = 0;
  }
  return DSDGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// RCCL all-reduce of the flat [gradient | scalars] buffer (include/dsdgp.h).  ncclAllReduce is resolved at first use from
// the RCCL already in the process (a torch process ships its own librccl) or from librccl.so: no link-time dependency.
// ------------------------------------------------------------------------------------------------------
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
static nccl_allreduce_fn g_allreduce = nullptr;
static nccl_errstr_fn g_errstr = nullptr;

static int resolve_rccl() {
  if (g_allreduce) return DSDGP_OK;
  void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
  void* h = nullptr;
  if (!sym) {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (h) sym = dlsym(h, "ncclAllReduce");
  }
  if (!sym) {
    const char* why = dlerror();      // (one call: dlerror() clears the error state)
    dsdgp_set_error("dsdgp_allreduce: RCCL (ncclAllReduce) is not loadable in this process: %s", why ? why : "symbol not found");
    return DSDGP_ERR_RCCL;
  }
  g_allreduce = (nccl_allreduce_fn)sym;
  g_errstr = (nccl_errstr_fn)(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
  return DSDGP_OK;
}

extern "C" int dsdgp_allreduce(dsdgp_ctx* ctx, void* comm, double* buf, int64_t count) {
  DS_CHECK_ARG(ctx && comm && buf && count > 0);
  DS_TRY(resolve_rccl());
  const int ncclFloat64 = 8, ncclSum = 0;   // rccl.h: ncclDataType_t / ncclRedOp_t
  const int rc = g_allreduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, comm, ctx->stream);
  if (rc != 0) {
    dsdgp_set_error("dsdgp_allreduce: ncclAllReduce failed: %s", g_errstr ? g_errstr(rc) : "unknown RCCL error");
    return DSDGP_ERR_RCCL;
  }
  return DSDGP_OK;
}
