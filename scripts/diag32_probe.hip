// Where does a one-wave k_big_diag32-style launch spend its time?  s_memtime stamps around the phases of
// diag32_factor_invert on a synthetic SPD block, plus a check against a host Cholesky + inverse.
#include "../rustrobotics_amd/csrc/kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace rrpgo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T> __global__ void __launch_bounds__(64) probe(T *F, int M, int nb, T *Wt, int *err, long long *st) {
  __shared__ T Dl[DIAG32_LDS];
  const int lane = threadIdx.x;
  long long t0 = clock64();
  for (int e = lane; e < 32 * 32; e += 64) {   // the image contract: zeros above the diagonal, identity padding
    const int c = e >> 5, r = e & 31;
    Dl[c * 33 + r] = (r < nb && c < nb && r >= c) ? F[(int64_t)c * M + r] : ((r == c && r >= nb) ? (T)1 : (T)0);
  }
  diag32_init_tables<T>(Dl);
  __syncthreads();
  long long t1 = clock64();
  __builtin_amdgcn_sched_barrier(0);
  diag32_factor_invert<T>(Dl, nb, F, M, Wt, err);
  __builtin_amdgcn_sched_barrier(0);
  __threadfence();
  long long t2 = clock64();
  if (lane == 0) { st[0] = t1 - t0; st[1] = t2 - t1; }
}

template <typename T> int run(const char *name, int nb) {
  const int M = 40;
  std::vector<double> A(32 * 32, 0.0), L(32 * 32, 0.0), W(32 * 32, 0.0);
  for (int i = 0; i < nb; i++)
    for (int j = 0; j <= i; j++) { double v = 1.0 / (1.0 + i + j) + (i == j ? 2.0 + 0.1 * i : 0.0); A[j * 32 + i] = v; A[i * 32 + j] = v; }
  for (int j = 0; j < nb; j++) {   // host Cholesky
    double d = A[j * 32 + j];
    for (int k = 0; k < j; k++) d -= L[k * 32 + j] * L[k * 32 + j];
    L[j * 32 + j] = std::sqrt(d);
    for (int i = j + 1; i < nb; i++) {
      double v = A[j * 32 + i];
      for (int k = 0; k < j; k++) v -= L[k * 32 + i] * L[k * 32 + j];
      L[j * 32 + i] = v / L[j * 32 + j];
    }
  }
  for (int c = 0; c < nb; c++)     // W = L^-1, column c
    for (int r = c; r < nb; r++) {
      double v = r == c ? 1.0 : 0.0;
      for (int k = c; k < r; k++) v -= L[k * 32 + r] * W[c * 32 + k];
      W[c * 32 + r] = v / L[r * 32 + r];
    }
  std::vector<T> hF(M * M, (T)0), hW(1024);
  for (int c = 0; c < nb; c++) for (int r = c; r < nb; r++) hF[c * M + r] = (T)A[c * 32 + r];
  T *dF, *dW; int *derr; long long *dst;
  CK(hipMalloc(&dF, sizeof(T) * M * M)); CK(hipMalloc(&dW, sizeof(T) * 1024)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&dst, 64));
  CK(hipMemset(derr, 0, 4));
  long long st[2]; float ms = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemcpy(dF, hF.data(), sizeof(T) * M * M, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<T>, dim3(1), dim3(64), 0, 0, dF, M, nb, dW, derr, dst);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    printf("%s nb=%d rep %d: load %lld ticks, factor+invert+store %lld ticks, events %.1f us\n", name, nb, rep, st[0], st[1], ms * 1e3);
  }
  std::vector<T> oF(M * M);
  CK(hipMemcpy(oF.data(), dF, sizeof(T) * M * M, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hW.data(), dW, sizeof(T) * 1024, hipMemcpyDeviceToHost));
  int herr; CK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
  double eL = 0, eW = 0;
  for (int c = 0; c < nb; c++) for (int r = c; r < nb; r++) {
    eL = std::fmax(eL, std::fabs((double)oF[c * M + r] - L[c * 32 + r]));
    eW = std::fmax(eW, std::fabs((double)hW[c * 32 + r] - W[c * 32 + r]));   // Wt[j*32 + c'] = W(c', j): j = c, c' = r
  }
  for (int j = 0; j < 32; j++) for (int c = 0; c < 32; c++) {
    double want = (c >= nb || j >= nb) ? (c == j ? 1.0 : 0.0) : (c >= j ? W[j * 32 + c] : 0.0);
    eW = std::fmax(eW, std::fabs((double)hW[j * 32 + c] - want));
  }
  printf("%s nb=%d: max|L - ref| = %.3g, max|W - ref| = %.3g, err flag %d\n", name, nb, eL, eW, herr);
  return (eL < (sizeof(T) == 8 ? 1e-12 : 1e-4) && eW < (sizeof(T) == 8 ? 1e-12 : 1e-4) && herr == 0) ? 0 : 1;
}
int main() {
  int bad = 0;
  bad |= run<double>("f64", 32); bad |= run<float>("f32", 32);
  bad |= run<double>("f64", 21); bad |= run<float>("f32", 7);
  printf(bad ? "FAILED\n" : "OK\n");
  return bad;
}
