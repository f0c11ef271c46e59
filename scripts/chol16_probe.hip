// The two 16 x 16 factor-and-invert sweeps of one wave, timed (s_memtime ticks at 100 MHz * 24 = shader clocks reported by clock64)
// and checked against a host Cholesky: chol16_invert (rows in registers, v_readlane broadcasts) and chol16_mfma (the block in
// MFMA accumulator layout, one rank-1 MFMA per column).  Exit code 0 = both within tolerance.
#include "../rustrobotics_amd/csrc/kernels.hip.h"
using namespace rrpgo;
// The two retired forms the probe compares chol16_dpp2 with (removed from kernels.hip.h in r05, kept here as the reference):
// 16 x 16 Cholesky AND inverse in the registers of one wave: lanes 0..15 hold the rows of the block,
// lanes 16..31 the rows of an identity, so the column sweep that turns the block into L turns the
// identity into L^-T (v_readlane broadcasts, rsqrt + Newton, no LDS and no barrier on the chain).
// `lane` is the lane index modulo 32 (lanes 32..63 mirror 0..31).
template <typename T> __device__ __forceinline__ bool chol16_invert(T (&x)[16], int lane) {
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    T d = lane_bcast(x[k], k);
    if (!(d > (T)0)) { bad = true; d = (T)1; }
    const T inv = chain_rsqrt(d);
    const T lik = lane >= k ? x[k] * inv : (T)0;
    x[k] = lik;
#pragma unroll
    for (int j = k + 1; j < 16; j++) x[j] -= lik * lane_bcast(lik, j);
  }
  return bad;
}

// 16 x 16 Cholesky AND inverse on the matrix cores, the block in ACCUMULATOR layout: register r of lane l holds
// element (MM::row(l, r), l & 15) of a 16 x 16 matrix -- what a trailing-update tile leaves behind, so the diagonal
// block goes from the update into the factorisation without an LDS round trip.
//   D   in: the SYMMETRIC block (both triangles)             destroyed
//   Lt  out: L^T  (element (k, i) = L(i, k), zero for i < k)
//   W   out: L^-1 (element (k, i) = W(k, i), zero for i > k)
// Column k: row k of D lives in ONE register of the 16 lanes of one lane group -- by symmetry it is column k, i.e. in the
// a / b operand position of an MFMA for k-slot (lane >> 4): the rank-1 update D -= l l^T is ONE v_mfma (the other three
// k-slots multiply zeros), and so is the forward substitution that carries the inverse along, F -= l w_k^T (F starts as
// the identity; w_k = row k of F over L_kk is row k of L^-1).  On the chain of a column: v_readlane of the pivot, rsqrt,
// one multiply, one MFMA -- the 15 - k (v_readlane, v_readlane, FMA) triples per column of the register sweep
// (chol16_invert) are gone.  Columns nb.. (a partial last block) are skipped: Lt and W are zero there.
template <typename T>
__device__ __forceinline__ bool chol16_mfma(typename Mfma16<T>::Acc &D, typename Mfma16<T>::Acc &Lt, typename Mfma16<T>::Acc &W, int nb) {
  using MM = Mfma16<T>;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  typename MM::Acc F;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    F[r] = MM::row(lane, r) == li ? (T)1 : (T)0;
    Lt[r] = 0;
    W[r] = 0;
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    constexpr bool f64 = sizeof(T) == 8;
    const int gk = f64 ? (k & 3) : (k >> 2), rk = f64 ? (k >> 2) : (k & 3);   // lane group and register of row k
    if (k < nb) {
      T d = lane_bcast(D[rk], 16 * gk + k);
      if (!(d > (T)0)) { bad = true; d = (T)1; }
      const T vk = (lk == gk && li >= k) ? D[rk] : (T)0;   // row k = column k, from the diagonal down (off the chain: D is there before 1/sqrt)
      const T fk = lk == gk ? F[rk] : (T)0;
      const T inv = chain_rsqrt(d);
      const T l = vk * inv, w = fk * inv;
      D = MM::mma(l, -l, D);
      F = MM::mma(l, -w, F);
      Lt[rk] += l;
      W[rk] += w;
    }
  }
  return bad;
}


#include <cmath>
#include <cstdio>
#include <vector>
template <typename T> __global__ void __launch_bounds__(64) probe_regs(T *buf, long long *st) {
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
// A: 16 x 16 symmetric, row-major; outputs L (lower) and W = L^-1 row-major
template <typename T> __global__ void __launch_bounds__(64) probe_mfma(const T *A, T *Lout, T *Wout, int nb, long long *st) {
  using MM = Mfma16<T>;
  const int lane = threadIdx.x, li = lane & 15;
  typename MM::Acc D, Lt, W;
#pragma unroll
  for (int r = 0; r < 4; r++) D[r] = A[MM::row(lane, r) * 16 + li];
  long long t0 = clock64();
#pragma unroll
  for (int r = 0; r < 4; r++) asm volatile("" : "+v"(D[r]));
  __builtin_amdgcn_sched_barrier(0);
  bool bad = chol16_mfma<T>(D, Lt, W, nb);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; r++) { asm volatile("" : "+v"(Lt[r])); asm volatile("" : "+v"(W[r])); }
  long long t1 = clock64();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int k = MM::row(lane, r);
    Lout[li * 16 + k] = Lt[r];   // Lt(k, i) = L(i, k)
    Wout[k * 16 + li] = W[r];
  }
  if (lane == 0) { st[0] = t1 - t0; st[1] = bad; }
}
template <typename T> int run(const char *name, int nb, double tol) {
  std::vector<double> A(256, 0.0), L(256, 0.0), W(256, 0.0);
  for (int i = 0; i < nb; i++)
    for (int j = 0; j <= i; j++) { double v = 1.0 / (1.0 + i + j) + (i == j ? 2.0 + 0.3 * i : 0.0); A[i * 16 + j] = v; A[j * 16 + i] = v; }
  for (int j = 0; j < nb; j++) {
    double d = A[j * 16 + j];
    for (int k = 0; k < j; k++) d -= L[j * 16 + k] * L[j * 16 + k];
    L[j * 16 + j] = std::sqrt(d);
    for (int i = j + 1; i < nb; i++) {
      double v = A[i * 16 + j];
      for (int k = 0; k < j; k++) v -= L[i * 16 + k] * L[j * 16 + k];
      L[i * 16 + j] = v / L[j * 16 + j];
    }
  }
  for (int c = 0; c < nb; c++)
    for (int r = c; r < nb; r++) {
      double v = r == c ? 1.0 : 0.0;
      for (int k = c; k < r; k++) v -= L[r * 16 + k] * W[k * 16 + c];
      W[r * 16 + c] = v / L[r * 16 + r];
    }
  std::vector<T> hA(256), hL(256), hW(256);
  for (int i = 0; i < 256; i++) hA[i] = (T)A[i];
  T *dA, *dL, *dW; long long *st;
  hipMalloc(&dA, 256 * sizeof(T)); hipMalloc(&dL, 256 * sizeof(T)); hipMalloc(&dW, 256 * sizeof(T)); hipMalloc(&st, 16);
  long long s[2] = {0, 0};
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(dA, hA.data(), 256 * sizeof(T), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_mfma<T>, dim3(1), dim3(64), 0, 0, dA, dL, dW, nb, st);
    hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
  }
  hipMemcpy(hL.data(), dL, 256 * sizeof(T), hipMemcpyDeviceToHost);
  hipMemcpy(hW.data(), dW, 256 * sizeof(T), hipMemcpyDeviceToHost);
  double el = 0, ew = 0;
  for (int i = 0; i < nb; i++)
    for (int j = 0; j < nb; j++) { el = std::fmax(el, std::fabs((double)hL[i * 16 + j] - L[i * 16 + j])); ew = std::fmax(ew, std::fabs((double)hW[i * 16 + j] - W[i * 16 + j])); }
  const bool ok = el < tol && ew < tol && s[1] == 0;
  printf("%s nb=%2d chol16_mfma: %lld clocks, max |dL| %.2e, max |dW| %.2e, bad=%lld %s\n", name, nb, s[0], el, ew, s[1], ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
// chol16_dpp2: same register layout as chol16_invert (lanes 0..15 block rows, 16..31 identity rows)
template <typename T> __global__ void __launch_bounds__(64) probe_dpp2(T *buf, long long *st) {
  const int lane = threadIdx.x;
  T x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = buf[c * 64 + lane];
  long long t0 = clock64();
#pragma unroll
  for (int c = 0; c < 16; c++) asm volatile("" : "+v"(x[c]));
  __builtin_amdgcn_sched_barrier(0);
  bool bad = chol16_dpp2<T>(x);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 16; c++) asm volatile("" : "+v"(x[c]));
  long long t1 = clock64();
#pragma unroll
  for (int c = 0; c < 16; c++) buf[c * 64 + lane] = x[c];
  if (lane == 0) { st[0] = t1 - t0; st[1] = bad; }
}
template <typename T> int check_dpp2(const char *name, double tol) {
  T h[1024], r0[1024], r1[1024];
  for (int c = 0; c < 16; c++) for (int l = 0; l < 64; l++) { const int q = l & 15; h[c * 64 + l] = ((l & 16) == 0) ? (q == c ? 4.0 + c : 0.1 / (1 + q + c)) : (q == c ? 1 : 0); }
  T *d; long long *st; hipMalloc(&d, sizeof(h)); hipMalloc(&st, 16);
  long long s[2];
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_regs<T>, dim3(1), dim3(64), 0, 0, d, st);
  hipMemcpy(r0, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_dpp2<T>, dim3(1), dim3(64), 0, 0, d, st);
    hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
  }
  hipMemcpy(r1, d, sizeof(h), hipMemcpyDeviceToHost);
  double e = 0;   // compare with the register sweep: lower triangle of L (lanes 0..15) and all of W (lanes 16..31)
  for (int c = 0; c < 16; c++) for (int l = 0; l < 32; l++) { if (l < 16 && c > l) continue; e = std::fmax(e, std::fabs((double)r0[c * 64 + l] - (double)r1[c * 64 + l])); }
  const bool ok = e < tol && s[1] == 0;
  printf("%s chol16_dpp2: %lld clocks, max |d| vs the register sweep %.2e %s\n", name, s[0], e, ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
template <typename T> void time_regs(const char *name) {
  T h[1024]; for (int c = 0; c < 16; c++) for (int l = 0; l < 64; l++) h[c * 64 + l] = (l < 16) ? (l == c ? 4.0 + c : (c < l ? 0.1 / (1 + l + c) : 0)) : (((l - 16) & 15) == c ? 1 : 0);
  T *d; long long *st; hipMalloc(&d, sizeof(h)); hipMalloc(&st, 16);
  long long s[2];
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_regs<T>, dim3(1), dim3(64), 0, 0, d, st);
    hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
  }
  printf("%s chol16_invert (register sweep): %lld clocks (bad=%lld)\n", name, s[0], s[1]);
}
int main() {
  int fail = 0;
  time_regs<float>("f32"); time_regs<double>("f64");
  fail += check_dpp2<float>("f32", 2e-5); fail += check_dpp2<double>("f64", 1e-13);
  for (int nb : {16, 9, 1}) { fail += run<double>("f64", nb, 1e-13); fail += run<float>("f32", nb, 2e-5); }
  printf(fail ? "FAILED\n" : "all OK\n");
  return fail;
}
