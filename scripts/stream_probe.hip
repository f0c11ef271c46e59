// Do a train of small dependent kernels on one stream and a chip-filling kernel on another run
// concurrently on this GPU/runtime?  Wall time of: train alone, big alone, both.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void small(float *p, int n) {
  float x = p[threadIdx.x];
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  p[threadIdx.x] = x;
}
template <int LDSB> __global__ void __launch_bounds__(256) big(float *p, int n) {
  __shared__ float sh[LDSB / 4];
  float x = threadIdx.x;
  sh[threadIdx.x] = x;
  __syncthreads();
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, sh[(threadIdx.x + i) & 255]);
  if (x == 12345.f) p[0] = x;
}
int main() {
  float *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t a, b; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  for (int prio = 0; prio < 2; prio++) {
    if (prio) { CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo)); }
    else { CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); }
    auto run = [&](bool train, bool bigk) -> double {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      if (bigk) hipLaunchKernelGGL(big<73728>, dim3(8192), dim3(256), 0, b, d, 6000);
      if (train) for (int i = 0; i < 300; i++) hipLaunchKernelGGL(small, dim3(4), dim3(64), 0, a, d + 1024, 1500);
      CK(hipDeviceSynchronize());
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    run(true, true);
    for (int rep = 0; rep < 2; rep++)
      printf("%s streams: train alone %.2f ms, big alone %.2f ms, both %.2f ms\n", prio ? "prioritised" : "plain", run(true, false), run(false, true), run(true, true));
    CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
  }
  return 0;
}
