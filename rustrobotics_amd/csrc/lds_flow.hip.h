// lds_flow.hip.h -- k_factor_flow / k_solve_flow: the multifrontal factorisation and the back substitution of a graph
// whose fronts ALL live in LDS (intel.g2o, M3500, dlr: up to a few thousand poses) as ONE launch each.
//
// The level schedule (k_factor_tasks / k_solve_tasks, one launch per level of the task tree) pays for every level the
// slowest task of that level plus a kernel boundary, and a front's workgroup cannot begin before the launch of its
// level does.  Here a front's workgroup is resident from the start of the launch:
//
//   tasks      every subtree cheaper than a threshold (walked in postorder by one workgroup, exactly as before) and every
//              front above those as a task of its own (symbolic.cpp, build_flow_schedule)
//   tickets    workgroups draw tasks from ONE atomic counter; the list is in the start order of a list schedule of the
//              task tree, a topological order: a task only waits for tasks with smaller tickets, which are finished or
//              held by a running workgroup -- no deadlock whatever the grid size, the dispatch order or other kernels
//              on the device
//   hand-off   factorisation: a front's packed update matrix is written through to memory (sc1 stores), then its flag is
//              set; the parent waits for the flag right before THAT child's extend-add -- zeroing its LDS image, adding
//              the H entries and the right-hand side, the earlier children and the child's scatter map are behind it by
//              then -- and reads the update matrix with sc1 loads.  The panel (which only the back substitution reads,
//              a later launch) is copied out after the flag.  Back substitution: no flags -- an entry of x is its own
//              flag (the linearisation kernel fills x with a NaN payload, X_PENDING_WORD; a front spins on the entries it
//              gathers: x_wait, kernels.hip.h), and everything that does not depend on x -- L21, the row indices, an LDS
//              image of L11 and the inverse diagonal blocks -- is requested before that wait.
//   order      children are added in a fixed order (symbolic.cpp: by the cost model's finish time), so every sum -- and
//              the result, bit for bit -- is the same as in the level schedule (process_front / solve_front are the same
//              code with FLOW = true).
//
// Every wait is bounded in time (dep_wait, kernels.hip.h).  Replaces umfpack.factorize / umfpack.solve
// (reference src/mapping/pose_graph_optimization.rs:138-141).
#pragma once

namespace rrpgo {

struct alignas(128) LdsFlowTask {   // what a workgroup loads per ticket: ONE 128-byte line
  SnMeta m;                         // record of the task's first front (factorisation) / last front (back substitution)
  int32_t sn;                       // that front
  int32_t sn_begin, sn_end;         // the task's fronts: task_sn[sn_begin .. sn_end)
  // k_solve_flow only -- a front beyond LDS with a narrow pivot block (kind > 0; the task is that ONE front):
  int32_t kind;                     //   0: LDS fronts (solve_front); 1: GEMV unit (64 columns x one row slice of L21^T x[rows]); 2: the front's L11
  int32_t slices;                   //   row slices of the level's k_big_gemv_partial launch (the sums are formed slice by slice: the same bits)
  int32_t bx, by;                   //   GEMV unit: column block, slice
  int32_t n_units;                  //   kind 2: the GEMV units to wait for (the front's solve flag word counts them)
  int32_t pad[8];
};
static_assert(sizeof(LdsFlowTask) == 128, "LdsFlowTask is one 128-byte line");

// the next ticket, the same in every thread of the workgroup (one atomic, broadcast through LDS; the barriers inside the
// task that follows separate this read from the next write)
__device__ __forceinline__ int lds_flow_ticket(unsigned *ticket, int *slot) {
  if (threadIdx.x == 0) *slot = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*slot);
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_factor_flow(FactorArgs<T> a, const LdsFlowTask *tasks, int n_tasks, unsigned *ticket) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ T dinv[W16_SCR];   // inverse of the current 16 x 16 diagonal block + an identity (diag16_factor_invert_full)
  __shared__ int ticket_slot[2];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const bool stopped = opt_stopped(a.err);   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule (the load is
                                             // back before the first ticket is: nothing waits for it)
  init_w16_identity<T>(dinv, threadIdx.x, THREADS);
  if (THREADS > 256 && wave_index() == 0) __builtin_amdgcn_s_setprio(RRPGO_CHAIN_PRIO);
  for (int round = 0;; round++) {
    const int tk = lds_flow_ticket(ticket, &ticket_slot[round & 1]);
    if (tk >= n_tasks || stopped) break;
    const LdsFlowTask tr = tasks[tk];
    int snext = tr.sn;
    SnMeta mnext = tr.m;
    for (int si = tr.sn_begin; si < tr.sn_end; si++) {
      const int s = snext;
      const SnMeta m = mnext;
      if (si + 1 < tr.sn_end) {   // the next front's record in flight under this front
        snext = a.task_sn[si + 1];
        mnext = a.sn_meta[snext];
      }
      process_front<T, THREADS, false, true>(a, s, m, smem, smem + (m.nc + m.nr + 1) * m.nc, 0, dinv);
    }
  }
}

// a counter of finished units behind a front's solve flag word: bounded like dep_wait
__device__ __forceinline__ bool dep_wait_count(const unsigned *p, unsigned need, int *err, unsigned long long max_ticks) {
  bool ok = true;
  if (dep_flag_ld(p) < need) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; spins++) {
      __builtin_amdgcn_s_sleep(1);
      if (dep_flag_ld(p) >= need) break;
      if ((spins & 31u) == 0u) {
        const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e != 0 || wall_clock64() - t0 > max_ticks) {
          if (e == 0 && (threadIdx.x & 63) == 0) atomicOr(err, DEVERR_FLOW_TIMEOUT);
          ok = false;
          break;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return ok;
}

// Fronts beyond LDS with narrow pivot blocks (sphere2500: five of six levels) are tasks of this launch too, in place of a
// k_big_gemv_partial + k_solve_mid launch pair per level (~24 us per level): per front its GEMV units (k_big_gemv_partial's
// decomposition and sums: 64 columns x one row slice, spread over the workgroups of the launch; x[rows] waited for entry by entry)
// count themselves in the front's solve flag word, the front's L11 task waits for the count and solves (solve_big_front).
// `part`: the partial sums, [slice][N] (N = dim).
template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_solve_flow(FactorArgs<T> a, const LdsFlowTask *tasks, int n_tasks, unsigned *ticket, T *part, int64_t N) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ int ticket_slot[2];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const bool stopped = opt_stopped(a.err);   // as k_factor_flow
  for (int round = 0;; round++) {
    const int tk = lds_flow_ticket(ticket, &ticket_slot[round & 1]);
    if (tk >= n_tasks || stopped) break;
    const LdsFlowTask tr = tasks[tk];
    if (tr.kind == 1) {
      big_gemv_unit<T, true>(a, tr.m, part, N, tr.slices, tr.bx, tr.by, smem);
      dep_drain();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(a.dep_flags + a.parent_dep_self + tr.sn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    if (tr.kind == 2) {
      dep_wait_count(a.dep_flags + a.parent_dep_self + tr.sn, (unsigned)tr.n_units, a.err, a.wait_ticks);
      solve_big_front<T, THREADS, true>(a, tr.m, smem, part, N, tr.slices);
      continue;
    }
    int snext = tr.sn;
    SnMeta mnext = tr.m;
    for (int si = tr.sn_end - 1; si >= tr.sn_begin; si--) {
      const int s = snext;
      const SnMeta m = mnext;
      if (si > tr.sn_begin) {
        snext = a.task_sn[si - 1];
        mnext = a.sn_meta[snext];
      }
      solve_front<T, THREADS, true, true>(a, s, m, smem);
    }
  }
}

// (the flags and the two tickets are zeroed and the solution vector marked pending by the linearisation kernel of the same
// iteration: LinArgs::zero_words, fill_words; the edge-parallel linearisation launches this instead)
__global__ void __launch_bounds__(256) k_fill_words(unsigned *w, int n, unsigned v) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) w[i] = v;
}

// rr_pgo_set_state with the state of the previous call: the poses come back from their device-side copy (16-byte words)
__global__ void __launch_bounds__(256) k_copy_words16(const uint4 *src, uint4 *dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace rrpgo
