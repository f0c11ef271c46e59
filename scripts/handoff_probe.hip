// In-launch hand-off probe for the big-front dataflow kernel (k_big_flow): which load / store flavours hand a
// 4 KB payload from one workgroup to another INSIDE a launch without stale reads, when 128-byte lines are SHARED
// by two producers (half a line each, written at different times) and the consumer's caches already hold the
// line -- the situation of two neighbouring tiles of a front.  Prints stale counts and the round-trip time.
//   hipcc --offload-arch=gfx950 -O3 scripts/handoff_probe.hip -o handoff_probe && ./handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LINES = 32, ELEMS = LINES * 32, PASSES = ELEMS / 64;
constexpr unsigned SPIN_MAX = 4000000u;

__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_flag(const unsigned *p, unsigned v, unsigned *tmo) {
  for (unsigned s = 0; s < SPIN_MAX; s++) {
    if (ld_flag(p) == v) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  if (threadIdx.x == 0) atomicAdd(tmo, 1u);
  return false;
}

template <int LOADF> __device__ __forceinline__ float ld(const float *p) {
  if (LOADF == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1
  if (LOADF == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // sc0 sc1
  return *p;
}
template <int STOREF> __device__ __forceinline__ void st(float *p, float v) {
  if (STOREF == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1
  else *p = v;
}

// mode bits: LOADF (0 plain, 1 sc1, 2 sc0 sc1), STOREF (0 plain, 1 sc1), ACQ (agent acquire before the reads),
// REL (agent release before the flag)
template <int LOADF, int STOREF, bool ACQ, bool REL>
__global__ void __launch_bounds__(64) k_probe(float *buf, unsigned *flags, int *xcc, long long *lat, unsigned *stale, unsigned *tmo,
                                             int rounds, int stride) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int g3 = b / (3 * stride), rem = b % (3 * stride);
  const int role = rem / stride, g = g3 * stride + rem % stride;   // role 0 consumer, 1 / 2 producers of half 0 / 1
  if (lane == 0) {
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b] = id & 15;
  }
  float *reg = buf + (size_t)g * ELEMS;
  unsigned *fl = flags + (size_t)g * 4 * 32;   // ready1, flag1, ready2, flag2: one 128-byte line each
  unsigned bad = 0;
  long long tsum = 0;
  float dummy = 0;
  for (int r = 1; r <= rounds; r++) {
    if (role == 0) {
      for (int p = 0; p < PASSES; p++) dummy += ld<LOADF>(reg + p * 64 + lane);   // the lines are in this CU's / XCD's caches now
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int h = 0; h < 2; h++) {
        const long long t0 = wall_clock64();
        if (lane == 0) st_flag(fl + (2 * h) * 32, (unsigned)r);
        if (!wait_flag(fl + (2 * h + 1) * 32, (unsigned)r, tmo)) return;
        if (ACQ) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float v[PASSES];
        for (int p = 0; p < PASSES; p++) v[p] = ld<LOADF>(reg + p * 64 + lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tsum += wall_clock64() - t0;
        const int half = (lane >> 4) & 1;
        for (int p = 0; p < PASSES; p++)
          if (half <= h && v[p] != (float)r) bad++;
      }
    } else {
      const int h = role - 1;
      if (!wait_flag(fl + (2 * h) * 32, (unsigned)r, tmo)) return;
      if (((lane >> 4) & 1) == h)
        for (int p = 0; p < PASSES; p++) st<STOREF>(reg + p * 64 + lane, (float)r);
      if (REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) st_flag(fl + (2 * h + 1) * 32, (unsigned)r);
    }
  }
  if (role == 0) {
    atomicAdd(stale, bad);
    if (lane == 0) lat[g] = tsum;
    if (dummy == 123.456f) buf[0] = dummy;
  }
}

template <int LOADF, int STOREF, bool ACQ, bool REL> void run(const char *name, int stride) {
  const int nb = 240, groups = nb / 3, rounds = 200;
  float *buf; unsigned *flags, *stale, *tmo; int *xcc; long long *lat;
  hipMalloc(&buf, sizeof(float) * groups * ELEMS); hipMemset(buf, 0, sizeof(float) * groups * ELEMS);
  hipMalloc(&flags, 4 * 128 * groups); hipMemset(flags, 0, 4 * 128 * groups);
  hipMalloc(&stale, 4); hipMemset(stale, 0, 4);
  hipMalloc(&tmo, 4); hipMemset(tmo, 0, 4);
  hipMalloc(&xcc, 4 * nb); hipMalloc(&lat, 8 * groups);
  hipLaunchKernelGGL((k_probe<LOADF, STOREF, ACQ, REL>), dim3(nb), dim3(64), 0, 0, buf, flags, xcc, lat, stale, tmo, rounds, stride);
  hipError_t e = hipDeviceSynchronize();
  unsigned hs = 0, ht = 0; std::vector<int> hx(nb); std::vector<long long> hl(groups);
  hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost); hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), xcc, 4 * nb, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), lat, 8 * groups, hipMemcpyDeviceToHost);
  int cross = 0;
  for (int b = 0; b < nb; b++) {
    const int g3 = b / (3 * stride), rem = b % (3 * stride);
    if (rem / stride == 0) { const int p1 = b + stride; cross += hx[b] != hx[p1]; (void)g3; }
  }
  double t = 0; for (long long v : hl) t += (double)v;
  printf("%-58s stride %d: stale %8u of %d  timeouts %u  cross-XCD pairs %d/%d  round trip %.2f us  (%s)\n", name, stride, hs,
         groups * rounds * 2 * PASSES * 64 * 3 / 4, ht, cross, groups, t / groups / rounds / 2 / 100.0, hipGetErrorString(e));
  hipFree(buf); hipFree(flags); hipFree(stale); hipFree(tmo); hipFree(xcc); hipFree(lat);
}

int main() {
  for (int stride : {1, 8}) {
    run<1, 1, false, false>("sc1 stores, sc1 loads, no fence", stride);
    run<2, 1, false, false>("sc1 stores, sc0 sc1 loads, no fence", stride);
    run<0, 1, true, false>("sc1 stores, acquire(agent) + plain loads", stride);
    run<0, 0, true, true>("plain stores + release(agent), acquire(agent) + plain loads", stride);
    run<1, 0, false, true>("plain stores + release(agent), sc1 loads", stride);
    run<0, 1, false, false>("sc1 stores, plain loads, no fence   [expected stale]", stride);
    run<1, 0, false, false>("plain stores, sc1 loads, no fence   [expected stale]", stride);
  }
  return 0;
}
