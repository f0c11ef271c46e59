// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned *d; hipMalloc(&d, 4 * 16 * 4); unsigned h[64];
  for (int threads : {512, 1024, 256}) {
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(2), dim3(threads), 60000, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; b++) {
      printf("%4d threads, workgroup %d: wave -> simd:", threads, b);
      for (int w = 0; w < threads / 64; w++) printf(" %d", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %d)\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
