// Shader clock seen by latency-bound single-wave kernels (s_memtime ticks vs the 100 MHz wall clock),
// alone, in a train of tiny launches, and next to a kernel that keeps the other CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(float *out, long long *t, int n) {
  long long c0 = clock64(), w0 = wall_clock64();
  float x = out[0];
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  long long c1 = clock64(), w1 = wall_clock64();
  out[1] = x;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
__global__ void burn(float *out, int n) {
  float x = threadIdx.x;
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  if (x == 12345.f) out[0] = x;
}
int main() {
  float *d; long long *t;
  hipMalloc(&d, 1024); hipMalloc(&t, 16 * 4096); hipMemset(d, 0, 1024);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  long long h[2 * 4096];
  auto report = [&](const char *tag, int nl) {
    hipDeviceSynchronize();
    hipMemcpy(h, t, sizeof(long long) * 2 * nl, hipMemcpyDeviceToHost);
    double c = 0, w = 0; for (int i = 0; i < nl; i++) { c += h[2 * i]; w += h[2 * i + 1]; }
    printf("%-44s ticks=%.0f wall100MHz=%.0f  -> %.0f MHz (if s_memtime = shader clock), %.2f us per launch body\n", tag, c / nl, w / nl, c / w * 100.0, w / nl / 100.0);
  };
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, 200000); report("single long chain (200k fma)", 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s1);
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, 2000);
    hipEventRecord(e1, s1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    report("train of 1000 launches x 2000 fma", 1); printf("   train: %.2f us per launch end to end\n", ms);
    hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, s2, d, 4000000);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, 2000);
    hipStreamSynchronize(s1);
    report("2000 fma next to a chip-filling kernel", 1);
    hipDeviceSynchronize();
  }
  return 0;
}
