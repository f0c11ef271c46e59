// In-launch hand-off probe for the big-front dataflow kernel (k_big_flow): which load / store flavours hand a
// 4 KB payload from one workgroup to another INSIDE a launch without stale reads, when 128-byte lines are SHARED
// by two producers (half a line each, written at different times) and the consumer's caches already hold the
// line -- the situation of two neighbouring tiles of a front.  Prints stale counts and the round-trip time.
//   hipcc --offload-arch=gfx950 -O3 scripts/handoff_probe.hip -o handoff_probe && ./handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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

int main8(int rounds);

// ---- r05: the payload-as-flag hand-off of k_solve_flow (kernels.hip.h, x_wait) for 8-BYTE payloads --------------------------
// An entry of x goes from "pending" to its value in ONE aligned 8-byte sc1 store; the consumer polls the entry itself.  x_wait
// trusts that such a store is single-copy atomic (a poll never sees one 32-bit half new and the other half old).  Here every
// round r writes (hi, lo) = (H | r, L ^ r) over round r - 1's pair; the consumer polls until the HIGH word is round r's and
// counts entries whose LOW word is still round r - 1's (torn).  FORM 0: __hip_atomic_store / load (global_store_dwordx2 sc1),
// FORM 1: raw buffer store / load b64 with the sc1 bit (Sc1Buf).  WIDE: the producer writes its 512 entries as 64 lanes x 8
// passes (one entry per lane and store, as solve_front does).
typedef unsigned pr_u2 __attribute__((ext_vector_type(2)));
template <int FORM> __device__ __forceinline__ void st8(double *base, int i, unsigned hi, unsigned lo) {
  if (FORM == 0) __hip_atomic_store(base + i, __hiloint2double((int)hi, (int)lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 512 * 8, 0x00020000);
    const pr_u2 w = {lo, hi};
    __builtin_amdgcn_raw_buffer_store_b64(w, r, i * 8, 0, 16);
  }
}
template <int FORM> __device__ __forceinline__ void ld8(const double *base, int i, unsigned &hi, unsigned &lo) {
  if (FORM == 0) {
    const double v = __hip_atomic_load(const_cast<double *>(base) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi = (unsigned)__double2hiint(v); lo = (unsigned)__double2loint(v);
  } else {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 512 * 8, 0x00020000);
    const pr_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, i * 8, 0, 16);
    hi = w.y; lo = w.x;
  }
}
template <int FORM>
__global__ void __launch_bounds__(64) k_probe8(double *buf, unsigned *go, unsigned long long *torn, unsigned long long *seen, unsigned *tmo, int rounds, int stride) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int g2 = b / (2 * stride), rem = b % (2 * stride);
  const int role = rem / stride, g = g2 * stride + rem % stride;   // role 0 consumer, 1 producer
  double *reg = buf + (size_t)g * 512;
  unsigned *fl = go + (size_t)g * 32;
  constexpr unsigned H = 0x40100000u, L = 0x5a5a0000u;
  unsigned long long bad = 0, n = 0;
  for (int r = 1; r <= rounds; r++) {
    if (role == 1) {
      // the consumer has checked round r - 1 (otherwise a fast producer would overwrite what is being polled)
      if (!wait_flag(fl, (unsigned)(r - 1), tmo)) return;
      for (int p = 0; p < 8; p++) st8<FORM>(reg, p * 64 + lane, H | (unsigned)r, L ^ (unsigned)r);
    } else {
      for (int p = 0; p < 8; p++) {
        unsigned hi, lo;
        unsigned s = 0;
        do { asm volatile("" ::: "memory"); ld8<FORM>(reg, p * 64 + lane, hi, lo); } while (hi != (H | (unsigned)r) && ++s < SPIN_MAX);   // (the clobber: a raw buffer load is no volatile access, the compiler may hoist it out of a poll)
        if (s >= SPIN_MAX) { if (lane == 0) atomicAdd(tmo, 1u); return; }
        bad += lo != (L ^ (unsigned)r);
        n++;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) st_flag(fl, (unsigned)r);
    }
  }
  if (role == 0) { atomicAdd(torn, bad); atomicAdd(seen, n); }
}
template <int FORM> void run8(const char *name, int stride, int rounds) {
  const int nb = 240, groups = nb / 2;
  double *buf; unsigned *go, *tmo; unsigned long long *torn, *seen;
  hipMalloc(&buf, 8 * 512 * groups); hipMemset(buf, 0, 8 * 512 * groups);
  hipMalloc(&go, 128 * groups); hipMemset(go, 0, 128 * groups);
  hipMalloc(&torn, 8); hipMemset(torn, 0, 8); hipMalloc(&seen, 8); hipMemset(seen, 0, 8);
  hipMalloc(&tmo, 4); hipMemset(tmo, 0, 4);
  hipLaunchKernelGGL((k_probe8<FORM>), dim3(nb), dim3(64), 0, 0, buf, go, torn, seen, tmo, rounds, stride);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long ht = 0, hs = 0; unsigned hto = 0;
  hipMemcpy(&ht, torn, 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, seen, 8, hipMemcpyDeviceToHost); hipMemcpy(&hto, tmo, 4, hipMemcpyDeviceToHost);
  printf("8-byte payload-as-flag, %-44s stride %d: torn %llu of %llu hand-offs  timeouts %u  (%s)\n", name, stride, ht, hs, hto, hipGetErrorString(e));
  hipFree(buf); hipFree(go); hipFree(torn); hipFree(seen); hipFree(tmo);
}
int main8(int rounds) {
  for (int stride : {1, 8}) {
    run8<0>("atomic store / load, agent scope (dwordx2 sc1)", stride, rounds);
    run8<1>("raw buffer store / load b64, sc1", stride, rounds);
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1) return main8(atoi(argv[1]));   // ./handoff_probe ROUNDS: the 8-byte payload-as-flag probe only (122880 hand-offs per round)
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
