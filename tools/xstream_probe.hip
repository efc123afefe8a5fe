// What does a cross-stream dependency cost the stream that is NOT waiting?  Event record / wait pairs against a flag in signal memory that
// the producer kernel writes itself and the other stream waits for with hipStreamWaitValue64 (no packet on the producer's stream).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/xstream_probe.hip -o tools/bin/libxstream_probe.so   (python: ctypes, torch's runtime)
//   hipcc --offload-arch=gfx950 -O2 -DPROBE_MAIN tools/xstream_probe.hip -o /tmp/xstream_probe               (standalone: /opt/rocm's)
// Every kernel spins `us` microseconds on 256 workgroups and stamps s_memrealtime (100 MHz) at its first start / last end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>

__global__ void spin(unsigned long long* stamp, int slot, int us, unsigned int* counter, unsigned long long* flag, unsigned long long val) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) atomicMin(&stamp[2 * slot], t0);
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    atomicMax(&stamp[2 * slot + 1], __builtin_amdgcn_s_memrealtime());
    if (flag) {
      __threadfence();
      const unsigned int prev = atomicAdd(counter, 1u);
      if (prev == gridDim.x - 1) {
        *counter = 0;
        __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define LAUNCH(st, slot, us, cnt, flg, val) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, d, slot, us, cnt, flg, (unsigned long long)(val))

extern "C" int xstream_probe_run(void) {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned long long* d;
  unsigned int* cnt;
  CK(hipMalloc(&d, 64 * sizeof(unsigned long long)));
  CK(hipMalloc(&cnt, 64));
  CK(hipMemset(cnt, 0, 64));
  unsigned long long* flag = nullptr;
  if (can) {
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
  }
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev, ev2, evs;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&evs, hipEventDisableTiming | hipEventDisableSystemFence));
  unsigned long long init[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
  unsigned long long val = 0;
  const char* names[] = {"0 same stream: K1 K2 (K3 free on s2)",
                         "1 fork by event: s1 K1 record K2 | s2 wait K3",
                         "2 fork by event on K1's completion signal (hipExtLaunchKernel stopEvent)",
                         "3 fork by flag: s1 K1(writes flag) K2 | s2 waitValue K3",
                         "4 join by event, long done: s2 K3(10us) record | s1 K1(50us) wait K2",
                         "5 join by flag, long done: s2 K3(10us, writes flag) | s1 K1(50us) waitValue K2",
                         "6 join by event, just in time: s2 K3(70us) record | s1 K1(50us) wait K2",
                         "7 join by flag, just in time: s2 K3(70us, writes flag) | s1 K1(50us) waitValue K2",
                         "8 fork by event without system fence",
                         "9 fork by hipStreamWriteValue64 on s1 behind K1 | s2 waitValue K3"};
  for (int mode = 0; mode < 10; ++mode) {
    if (!can && (mode == 3 || mode == 5 || mode == 7 || mode == 9)) continue;
    printf("%s\n", names[mode]);
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      ++val;
      // warm both queues so that neither starts from idle
      LAUNCH(s1, 3, 5, nullptr, nullptr, 0);
      LAUNCH(s2, 3, 5, nullptr, nullptr, 0);
      switch (mode) {
        case 0:
          LAUNCH(s1, 0, 50, nullptr, nullptr, 0);
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          LAUNCH(s2, 2, 20, nullptr, nullptr, 0);
          break;
        case 1:
        case 8:
          LAUNCH(s1, 0, 50, nullptr, nullptr, 0);
          CK(hipEventRecord(mode == 1 ? ev : evs, s1));
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          CK(hipStreamWaitEvent(s2, mode == 1 ? ev : evs, 0));
          LAUNCH(s2, 2, 20, nullptr, nullptr, 0);
          break;
        case 2:
          hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, nullptr, ev, 0, d, 0, 50, (unsigned int*)nullptr, (unsigned long long*)nullptr, 0ull);
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          CK(hipStreamWaitEvent(s2, ev, 0));
          LAUNCH(s2, 2, 20, nullptr, nullptr, 0);
          break;
        case 3:
          CK(hipStreamWaitValue64(s2, flag, val, hipStreamWaitValueGte, ~0ull));
          LAUNCH(s2, 2, 20, nullptr, nullptr, 0);
          LAUNCH(s1, 0, 50, cnt, flag, val);
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          break;
        case 4:
        case 6:
          LAUNCH(s2, 2, mode == 4 ? 10 : 70, nullptr, nullptr, 0);
          CK(hipEventRecord(ev2, s2));
          LAUNCH(s1, 0, 50, nullptr, nullptr, 0);
          CK(hipStreamWaitEvent(s1, ev2, 0));
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          break;
        case 5:
        case 7:
          LAUNCH(s2, 2, mode == 5 ? 10 : 70, cnt, flag, val);
          LAUNCH(s1, 0, 50, nullptr, nullptr, 0);
          CK(hipStreamWaitValue64(s1, flag, val, hipStreamWaitValueGte, ~0ull));
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          break;
        case 9:
          CK(hipStreamWaitValue64(s2, flag, val, hipStreamWaitValueGte, ~0ull));
          LAUNCH(s2, 2, 20, nullptr, nullptr, 0);
          LAUNCH(s1, 0, 50, nullptr, nullptr, 0);
          CK(hipStreamWriteValue64(s1, flag, val, 0));
          LAUNCH(s1, 1, 20, nullptr, nullptr, 0);
          break;
      }
      CK(hipStreamSynchronize(s1));
      CK(hipStreamSynchronize(s2));
      unsigned long long h[8];
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      const double t0 = (double)h[0];
      printf("   K1 %.1f..%.1f  K2(s1) %.1f..%.1f  K3(s2) %.1f..%.1f   | K1end->K2 %.2f  K1end->K3 %.2f  K3end->K2 %.2f us\n", 0.0, (h[1] - t0) * 0.01,
             (h[2] - t0) * 0.01, (h[3] - t0) * 0.01, ((double)h[4] - t0) * 0.01, ((double)h[5] - t0) * 0.01, ((double)h[2] - (double)h[1]) * 0.01,
             ((double)h[4] - (double)h[1]) * 0.01, ((double)h[2] - (double)h[5]) * 0.01);
    }
  }
  return 0;
}
#ifdef PROBE_MAIN
int main() { return xstream_probe_run(); }
#endif
