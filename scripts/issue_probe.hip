// Per-instruction cost seen by ONE wave (and by several waves of one workgroup) on gfx950:
// s_memtime ticks per instruction for dependent / independent VALU, readlane + fma pairs, LDS round
// trips, MFMA chains, barriers and dependent global loads.  Numbers feed the latency model in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 256;   // unrolled instructions per measurement

template <int MODE> __global__ void probe(float *buf, long long *out, int *idx) {
  __shared__ float lds[4096];
  float x = buf[threadIdx.x], y = x + 1.f, z = x + 2.f, w = x + 3.f;
  double dx = x;
  f4 acc = {x, y, z, w}, acc2 = acc, acc3 = acc, acc4 = acc;
  d4 dacc = {x, y, z, w};
  lds[threadIdx.x] = x;
  __syncthreads();
  int p = idx[threadIdx.x & 63];
  long long t0 = clock64();
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
  } else if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(y));
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(z));
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(w));
    }
  } else if (MODE == 2) {   // dependent f64 fma
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(dx));
  } else if (MODE == 3) {   // readlane -> fma pairs, independent of each other
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      int s;
      asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(x), "n"(i & 31));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y) : "s"(s), "v"(x));
    }
  } else if (MODE == 4) {   // dependent LDS round trips
#pragma unroll
    for (int i = 0; i < N / 4; i++) { p = __float_as_int(lds[p & 4095]) & 4095; }
  } else if (MODE == 5) {   // dependent MFMA f32 16x16x4
#pragma unroll
    for (int i = 0; i < N / 4; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
  } else if (MODE == 6) {   // 4 independent MFMA f32 16x16x4 chains
#pragma unroll
    for (int i = 0; i < N / 16; i++) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc3, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc4, 0, 0, 0);
    }
  } else if (MODE == 7) {   // dependent MFMA f64 16x16x4
#pragma unroll
    for (int i = 0; i < N / 4; i++) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(dx, dx, dacc, 0, 0, 0);
  } else if (MODE == 8) {   // barriers
#pragma unroll
    for (int i = 0; i < N / 4; i++) __syncthreads();
  } else if (MODE == 9) {   // dependent global loads (pointer chase inside one cache-resident page)
#pragma unroll
    for (int i = 0; i < N / 8; i++) p = idx[p & 63];
  } else if (MODE == 10) {  // scalar ALU chain
    int s = __builtin_amdgcn_readfirstlane(p);
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
    p = s;
  } else if (MODE == 11) {  // LDS write then read by another lane (what a layout change costs)
#pragma unroll
    for (int i = 0; i < N / 8; i++) { lds[threadIdx.x] = x; __syncthreads(); x += lds[(threadIdx.x + 17) & 63]; }
  }
  long long t1 = clock64();
  buf[threadIdx.x] = x + y + z + w + (float)dx + acc[0] + acc2[1] + acc3[2] + acc4[3] + (float)dacc[0] + p;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> int run(const char *name, int count, float *buf, long long *out, int *idx) {
  for (int threads : {64, 256, 512, 1024}) {
    if (MODE != 8 && MODE != 0 && MODE != 3 && MODE != 5 && MODE != 4 && threads > 64) continue;
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, buf, out, idx);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, buf, out, idx);
    CK(hipDeviceSynchronize());
    long long h[16]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-46s waves=%2d  %7.1f ticks per op (wave 0), %7.1f (last wave)\n", name, threads / 64, (double)h[0] / count, (double)h[threads / 64 - 1] / count);
  }
  return 0;
}
int main() {
  float *buf; long long *out; int *idx;
  CK(hipMalloc(&buf, 4096 * 4)); CK(hipMalloc(&out, 16 * 8)); CK(hipMalloc(&idx, 64 * 4));
  CK(hipMemset(buf, 0, 4096 * 4));
  int h[64]; for (int i = 0; i < 64; i++) h[i] = (i * 7 + 3) & 63;
  CK(hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>("dependent v_fma_f32", N, buf, out, idx);
  run<1>("4 independent v_fma_f32 chains", N, buf, out, idx);
  run<2>("dependent v_fma_f64", N, buf, out, idx);
  run<3>("v_readlane + v_fma pair (per pair)", N / 2, buf, out, idx);
  run<4>("dependent LDS read", N / 4, buf, out, idx);
  run<5>("dependent mfma_f32_16x16x4", N / 4, buf, out, idx);
  run<6>("4 independent mfma_f32_16x16x4 chains (per mfma)", N / 4, buf, out, idx);
  run<7>("dependent mfma_f64_16x16x4", N / 4, buf, out, idx);
  run<8>("__syncthreads", N / 4, buf, out, idx);
  run<9>("dependent global load (cached)", N / 8, buf, out, idx);
  run<10>("dependent s_add_u32", N, buf, out, idx);
  run<11>("LDS write + barrier + LDS read other lane", N / 8, buf, out, idx);
  return 0;
}
