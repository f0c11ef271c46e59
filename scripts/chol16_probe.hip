// ticks of one chol16_invert sweep (one wave, registers only)
#include "../rustrobotics_amd/csrc/kernels.hip.h"
#include <cstdio>
using namespace rrpgo;
template <typename T> __global__ void __launch_bounds__(64) probe(T *buf, long long *st) {
  const int lane = threadIdx.x;
  T x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = buf[c * 64 + lane];
  long long t0 = clock64();
#pragma unroll
  for (int c = 0; c < 16; c++) asm volatile("" : "+v"(x[c]));
  __builtin_amdgcn_sched_barrier(0);
  bool bad = chol16_invert<T>(x, lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 16; c++) asm volatile("" : "+v"(x[c]));
  long long t1 = clock64();
#pragma unroll
  for (int c = 0; c < 16; c++) buf[c * 64 + lane] = x[c];
  if (lane == 0) { st[0] = t1 - t0; st[1] = bad; }
}
template <typename T> void run(const char *name) {
  T h[1024]; for (int c = 0; c < 16; c++) for (int l = 0; l < 64; l++) h[c * 64 + l] = (l < 16) ? (l == c ? 4.0 + c : (c < l ? 0.1 / (1 + l + c) : 0)) : (((l - 16) & 15) == c ? 1 : 0);
  T *d; long long *st; hipMalloc(&d, sizeof(h)); hipMalloc(&st, 16);
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<T>, dim3(1), dim3(64), 0, 0, d, st);
    long long s[2]; hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
    printf("%s chol16_invert: %lld ticks (bad=%lld)\n", name, s[0], s[1]);
  }
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
