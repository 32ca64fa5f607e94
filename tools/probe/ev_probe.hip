// What does a cross-stream hand-over cost the PRODUCING stream?  A chain of N dependent kernels on stream A; after each one a
// second stream B is told to wait for it and runs a tiny kernel.  (a) no hand-over, (b) hipEventRecord + hipStreamWaitEvent,
// (c) the kernel's own completion signal as the event: hipExtLaunchKernelGGL(..., stopEvent) + hipStreamWaitEvent.
// Prints the wall time of the chain on A per kernel (HIP events around the chain) for two kernel durations.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  float *a, *b;
  CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22));
  CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
  const int N = 200;
  hipEvent_t t0, t1, ev[2], evx[2];
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evx[i], hipEventDisableTiming | hipEventDisableSystemFence)); }
  for (int iters : {200, 20000}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, A));
        for (int i = 0; i < N; ++i) {
          hipEvent_t e = (mode == 3 ? evx : ev)[i & 1];
          if (mode == 0) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, a, iters);
          } else if (mode == 1) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, a, iters);
            CK(hipEventRecord(e, A));
            CK(hipStreamWaitEvent(B, e, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, b, 100);
          } else {
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, nullptr, e, 0, a, iters);
            CK(hipStreamWaitEvent(B, e, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, b, 100);
          }
        }
        CK(hipEventRecord(t1, A));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        if (ms < best) best = ms;
      }
      const char* names[] = {"no hand-over", "hipEventRecord + wait", "ext launch stopEvent + wait", "ext launch stopEvent (no system fence) + wait"};
      printf("iters %6d  %-48s %.2f us per chain kernel\n", iters, names[mode], best * 1e3f / N);
    }
  }
  // ---- cost of a wait that is already satisfied (the forward's level waits, the join): A waits for an event B recorded long ago
  {
    hipEvent_t done;
    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, b, 100);
    CK(hipEventRecord(done, B));
    CK(hipDeviceSynchronize());
    for (int iters : {200, 20000}) {
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(t0, A));
          for (int i = 0; i < N; ++i) {
            if (mode == 1) CK(hipStreamWaitEvent(A, done, 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, a, iters);
          }
          CK(hipEventRecord(t1, A));
          CK(hipDeviceSynchronize());
          float ms; CK(hipEventElapsedTime(&ms, t0, t1));
          if (ms < best) best = ms;
        }
        printf("iters %6d  %-48s %.2f us per chain kernel\n", iters, mode ? "satisfied hipStreamWaitEvent before each kernel" : "plain chain", best * 1e3f / N);
      }
    }
  }
  return 0;
}
