// flow.hip.h -- k_big_flow: the fronts beyond LDS of ONE tree level as ONE launch of ticket-ordered tile tasks.
//
// The launch sequence of kernels.hip.h ("huge fronts") is level-synchronous twice over: every 32-column step of
// the panel chain and every 128-column trailing update is a launch of its own, so a level of few fronts is a chain
// of ~8 us launches that leave most of the chip idle (r02: the six top levels of the 1M-edge lattice took 3.4 of
// 5.8 ms at MfmaUtil 1.8 %).  Here the same arithmetic (same device functions; in the exact mode the same order of
// every sum: bit-identical to the launch sequence) runs as a dataflow:
//
//   task PANEL(front, kb, row-block group)   four waves = four 32-row blocks below the 32-column block at kb:
//        left-looking update over the step's range of earlier blocks (its super-panel's; for the first two blocks of
//        a super-panel, in the fast mode, the previous super-panel's too), X = A W^T; the wave that owns rows
//        kb+32..kb+63 then forms, factors and inverts the NEXT diagonal block (the chain) -- the body of k_big_panel32
//   task UPDATE(front, K0, tile)             one 64 x 64 tile of the K = 128 trailing update of super-panel K0
//        (the first one of a front gathers its tile from the children); in the exact mode tile (0, 0) also factors
//        and inverts the next super-panel's first diagonal block -- the body of k_big_update
//   task DIAG0(front)                        the front's first diagonal block
//
// Workgroups draw tasks from ONE atomic ticket; the task list is in a topological order (the start order of a list
// schedule of the task DAG, host side: Engine::build_flow_levels), so a task only ever waits for tasks with smaller
// tickets, which are finished or held by a running workgroup: no deadlock whatever the grid size or the dispatch
// order.  Completion is published per unit (X of a row block, a tile) in flag words; payload moves with sc1
// (write-through / L1-bypassing) accesses, flags with agent-scope relaxed atomics after the storing wave's
// s_waitcnt vmcnt(0) (MI355X_MICROARCH.md, inter-workgroup visibility; scripts/handoff_probe.hip measured this form
// stale-free with lines shared by two producers, 0.8 us per hop against ~1.5-2 us per kernel boundary).  The one
// hand-off on the chain, W of a diagonal block, is polled in place (k_flow_reset marks the blocks, see there).  Every
// spin is bounded: a wait that runs out sets DEVERR_FLOW_TIMEOUT, after which every wait returns at once and the
// launch drains.
// Replaces the reference's umfpack.factorize (src/mapping/pose_graph_optimization.rs:138) for those fronts.
#pragma once

namespace rrpgo {

constexpr int FLOW_MAX_BACK = 5;   // blocks a PANEL step looks back over at most: three of its own super-panel, or four of the previous one and one
struct FlowFront {       // one front of a flow level: where its flags live (indices into FlowArgs::flags)
  int32_t wf;            // wf[b]: W of 32-column block b is in winv (and every row the chain read for it is in F)
  int32_t pf, pstride;   // pf[b * pstride + rb]: X of row block rb of block b is in F
  int32_t uf, ustride;   // uf[sp * ustride + bx (bx + 1) / 2 + by]: tile (bx, by) of super-panel sp's update is in F (tiles of 64 or 128: FlowLevel::nt)
  // cross-level form (ONE launch for every level, r05; -1 / 0 otherwise):
  int32_t bf, nbuild;    // bf: counter of finished BUILD tasks of this front; the one that counts to nbuild sets the flag word bf + 1, which the
                         // front's DIAG0 and PANEL tasks wait for (its UPDATE tasks wait for PANEL flags: nothing to add)
  int32_t done;          // counter of finished UPDATE tasks: the parent's BUILD tasks wait for it (FlowArgs::child_done holds word and target)
};
static_assert(sizeof(FlowFront) == 32, "FlowFront is one 32-byte record");

struct FlowTask {        // 16 bytes, one scalar load
  int32_t kind_front;    // kind << 24 | front slot of the level
  int32_t p0, p1, p2;    // PANEL: kb, first row block, K0;  UPDATE: K0, bx, by;  DIAG0: -
};
constexpr int FLOW_PANEL = 0, FLOW_UPDATE = 1, FLOW_DIAG0 = 2, FLOW_BUILD = 3;   // BUILD(front, first column): cross-level form only
constexpr int FLOW_BUILD_COLS = 4;   // pivot columns per BUILD task: one per wave, k_big_build's decomposition
constexpr int FLOW_GROUP = 4;   // row blocks per PANEL task (one per wave)

struct alignas(128) FlowRec {   // what a workgroup loads per ticket: the task and its front's records in ONE 128-byte line
  FlowTask t;                   // (the front's records in tables of their own were a second, dependent round trip
  FlowFront ff;                 // per task: ~0.8 us of a 6 us task on the levels that are bound by their task count)
  SnMeta m;
  int32_t pad[4];
};
static_assert(sizeof(FlowRec) == 128, "FlowRec is one 128-byte line");
template <typename T> struct FlowArgs {   // everything the launch reads: a slim kernel-argument block (the ticket loop keeps all of it in SGPRs)
  const FlowRec *tasks;
  unsigned *ticket;
  unsigned *flags;
  int n_tasks;
  int gather;   // the first trailing update of a front gathers its tiles right of big_built_cols from the children
  int schur_tile;   // > 0: the UPDATE tiles stop at the front's Schur origin (big_schur_origin with this tile edge); the Schur
                    // complement is formed by ONE k_big_schur launch behind this one (K = nc, high occupancy); 0: UPDATE tiles reach through it
  int exact;    // 1: the next super-panel's first diagonal block comes out of tile (0, 0) of the trailing update, exactly as
                // in the launch sequence (bit-identical results); 0: the chain wave forms it itself, left-looking over the
                // whole super-panel like every other diagonal block -- same sums in another order, and no tile on the chain
  const ChildMeta *child_meta;
  const int32_t *scat;
  T *lvals, *uvals, *xch, *winv;
  T *xnew;      // per 32-column block of a flow front (laid out like winv): the chain wave's X block, polled in place by the next step (see flow_panel_wave)
  int *err;
  unsigned long long wait_ticks;   // bound of one wait, 100 MHz wall-clock ticks (RR_PGO_FLOW_TIMEOUT_MS; default 2 s)
  unsigned long long *trace;   // diagnostic build (-DRRPGO_FLOW_TRACE): [ticket][wave][4] wall-clock stamps, else null
  // cross-level form: what the BUILD tasks read (k_big_build's arguments)
  const int2 *child_done;      // per child_meta entry: (word of the child's `done` counter, UPDATE tasks it has), (-1, 0): factored by an earlier launch
  const int32_t *fasm_colptr, *fasm_dst, *fasm_src, *perm;
  const T *hvals, *b;
};

// stamps of a task's wave: 0 = ticket drawn, 1 = dependencies met, 2 = X / tile stored, 3 = flags set (100 MHz wall clock)
#ifdef RRPGO_FLOW_TRACE
#define RRPGO_FLOW_MARK(fa, t, wv, slot)                                                                     \
  do {                                                                                                        \
    if ((fa).trace && (lane) == 0) (fa).trace[((size_t)(t) * 4 + (size_t)(wv)) * 4 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define RRPGO_FLOW_MARK(fa, t, wv, slot) do { } while (0)
#endif


__device__ __forceinline__ unsigned flow_flag_ld(const unsigned *p) { return __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flow_flag_set(unsigned *p) { __hip_atomic_store(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave waits until every flag its lanes name (nullptr = none) is set.  Bounded; false = gave up or the launch
// is draining after an error (the caller goes on: addresses never depend on data, the results are reported invalid).
__device__ __forceinline__ bool flow_wait(const unsigned *p, int *err, unsigned long long max_ticks) {
  bool ok = true;
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; spins++) {
    const unsigned v = p ? flow_flag_ld(p) : 1u;
    if (__all(v != 0u)) break;
    if ((spins & 63u) == 63u) {
      if (t0 == 0) t0 = wall_clock64();   // (the bound is wall-clock time, not a poll count: a poll's cost varies with clocks, contention and profilers)
      const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || wall_clock64() - t0 > max_ticks) {
        if (e == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicOr(err, DEVERR_FLOW_TIMEOUT);
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the payload loads below the poll
  return ok;
}
// wait until a counter has reached `need` (wave-uniform word).  Bounded like flow_wait.
__device__ __forceinline__ bool flow_wait_ge(const unsigned *word, unsigned need, int *err, unsigned long long max_ticks) {
  bool ok = true;
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; spins++) {
    if (flow_flag_ld(word) >= need) break;
    if ((spins & 63u) == 63u) {
      if (t0 == 0) t0 = wall_clock64();
      const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || wall_clock64() - t0 > max_ticks) {
        if (e == 0) atomicOr(err, DEVERR_FLOW_TIMEOUT);
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return ok;
}
// the mark k_flow_reset leaves in a W block that is not there yet (see there)
__device__ __forceinline__ bool flow_w_unset(float v) { return __float_as_uint(v) == 0xffffffffu; }
// fp64: EITHER half still holding the mark counts as not there (an aligned 8-byte sc1 store was never seen torn on this chip --
// scripts/handoff_probe.hip, 0 of 1e9 -- but the poll does not rest on it).  A finished value whose low word happens to be all ones
// (one in 2^32; the high word cannot be: that is a NaN) keeps the poll going for flow_w_patience polls, then it is taken as read.
__device__ __forceinline__ bool flow_w_unset(double v) { return (unsigned)__double2hiint(v) == 0xffffffffu || (unsigned)__double2loint(v) == 0xffffffffu; }
__device__ __forceinline__ bool flow_w_hi_unset(float v) { return flow_w_unset(v); }
__device__ __forceinline__ bool flow_w_hi_unset(double v) { return (unsigned)__double2hiint(v) == 0xffffffffu; }
constexpr unsigned flow_w_patience = 16;   // polls with only low words at the mark before the values are accepted
// after this wave's payload stores: drain them, then the caller sets its flags
__device__ __forceinline__ void flow_drain() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ int flow_tri(int bx, int by) { return bx * (bx + 1) / 2 + by; }

// ---- DIAG0: the very first diagonal block of a front, by one wave (k_big_diag32's arithmetic).  The block stays as
// assembled in F: nothing reads a diagonal block of L once its W exists (the back substitution works with the inverses).
template <typename T>
__device__ __forceinline__ void flow_diag0_wave(const FlowArgs<T> &fa, const FlowFront &ff, const SnMeta &m, T *Sh, int tid) {
  const int nb = min(BIG_NB, m.nc);
  const int M = m.nc + m.nr + 1;
  T *F = fa.lvals + m.loff;
  const int lane = tid & 63;
  const Sc1Buf<T> fbuf(F, (uint32_t)((int64_t)M * M * (int64_t)sizeof(T)));
  T dv[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int e = t * 64 + lane, c = e >> 5, r = e & 31;
    dv[t] = fbuf.ld((uint32_t)(min(c, nb - 1) * M + min(r, nb - 1)) * (uint32_t)sizeof(T));
  }
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int e = t * 64 + lane, c = e >> 5, r = e & 31;
    Sh[c * 33 + r] = (r < nb && c < nb && r >= c) ? dv[t] : ((r == c && r >= nb) ? (T)1 : (T)0);
  }
  diag32_init_tables<T>(Sh);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  diag32_factor_invert<T, false, true>(Sh, nb, F, M, fa.winv + (int64_t)m.wblk * 256, fa.err, false, true);
  flow_drain();
  if (lane == 0) flow_flag_set(fa.flags + ff.wf);
}

// ---- BUILD (cross-level form): FLOW_BUILD_COLS pivot columns of a front, one per wave -- k_big_build's arithmetic (the sum, in
// child order, of what the children hold for every entry, then the column's H entries and its right-hand-side entry on top),
// with the children's entries read past L1 and the column written through: children and parent are fronts of ONE launch.  The
// task waits for the `done` counters of the children that are fronts of this launch (the others were factored by k_factor_flow,
// an earlier launch), and counts itself in the front's build counter when its columns are in memory.
template <typename T>
__device__ __forceinline__ void flow_build_task(const FlowArgs<T> &fa, const FlowFront &ff, const SnMeta &m, int j0, int tid, int ticket) {
  const int M = m.nc + m.nr + 1;
  T *F = fa.lvals + m.loff;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const ChildMeta *cm = fa.child_meta + m.child_begin;
  if (wave == 0) {
    for (int q0 = 0; q0 < m.child_count; q0 += 64) {   // (a front has a handful of children: one pass)
      const int q = q0 + lane;
      int2 cd = make_int2(-1, 0);
      if (q < m.child_count) cd = fa.child_done[m.child_begin + q];
      // lanes poll their own child's counter; the wave goes on when every lane's has reached its target
      bool ok = true;
      unsigned long long t0 = 0;
      for (unsigned spins = 0;; spins++) {
        const bool there = cd.x < 0 || flow_flag_ld(fa.flags + cd.x) >= (unsigned)cd.y;
        if (__all(there)) break;
        if ((spins & 63u) == 63u) {
          if (t0 == 0) t0 = wall_clock64();
          const int e = __hip_atomic_load(fa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (e != 0 || wall_clock64() - t0 > fa.wait_ticks) {
            if (e == 0 && lane == 0) atomicOr(fa.err, DEVERR_FLOW_TIMEOUT);
            ok = false;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      (void)ok;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  RRPGO_FLOW_MARK(fa, ticket, wave, 1);
  const int Jend = big_built_cols(m.nc, M);
  const int J = j0 + wave;
  if (J < Jend) {
    T *col = F + (int64_t)J * M;
    if (m.child_count <= 2) big_build_column<T, 2, true, FlowArgs<T>>(fa, cm, m.child_count, M, J, col, lane);
    else big_build_column<T, 4, true, FlowArgs<T>>(fa, cm, m.child_count, M, J, col, lane);
    if (J < m.nc) {
      // the column's H entries and its right-hand-side entry on top of what this wave's other lanes have just stored
      // (k_big_build, with_h: the same sums)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int t0 = fa.fasm_colptr[m.col0 + J], t1 = fa.fasm_colptr[m.col0 + J + 1];
      for (int t = t0 + lane; t < t1; t += 64) {
        T *p = F + fa.fasm_dst[t];
        mem_st<true>(p, mem_ld<true>(p) + fa.hvals[fa.fasm_src[t]]);
      }
      if (lane == 0) {
        T *p = F + (int64_t)J * M + (M - 1);
        mem_st<true>(p, mem_ld<true>(p) + fa.b[fa.perm[m.col0 + J]]);
      }
    }
  }
  flow_drain();
  __syncthreads();
  if (tid == 0) {
    // every BUILD task drains its stores before it counts itself: whoever counts last knows all the front's pivot columns are in memory
    const unsigned before = __hip_atomic_fetch_add(fa.flags + ff.bf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == (unsigned)ff.nbuild) flow_flag_set(fa.flags + ff.bf + 1);
  }
  RRPGO_FLOW_MARK(fa, ticket, wave, 3);
}

// ---- PANEL: one wave = 32 rows below the block at kb (k_big_panel32's arithmetic, sc1 accesses, flags).
// The order of work is the chain's schedule.  Everything that does not depend on the diagonal block being factored
// right now comes first -- the left-looking update from the super-panel's earlier blocks, oldest first, each block's
// operands requested one block ahead (two register sets); only the NEWEST block waits for its X (published one
// step ago, by the chain wave BEFORE it factors).  Then the wave waits for W, multiplies, stores and publishes its X;
// the wave that owns the next diagonal block's rows goes on: last term of that block, factor and invert, publish W.
// So one step of the chain costs: hop + newest block's loads and MFMAs + hop + W load + X + factor-and-invert,
// with the first half running under the previous step's factor-and-invert.
template <typename T, int TS /* edge of a trailing-update tile: 64 or 128 */>
__device__ __forceinline__ void flow_panel_wave(const FlowArgs<T> &fa, const FlowFront &ff, const SnMeta &m, int kb,
                                                int K0raw, int rowblk, T *Sh, int tid, int ticket, const unsigned *built = nullptr) {
  static_assert(BIG_NB == 32 && BIG_SUPER == 128, "written for 32-column blocks in 128-column super-panels");
  using MM = Mfma16<T>;
  const int nb = min(BIG_NB, m.nc - kb);
  const int M = m.nc + m.nr + 1;
  const int R0 = kb + nb + rowblk * 32;
  if (R0 >= M) return;
  T *F = fa.lvals + m.loff;
  T *Wt = fa.winv + (int64_t)m.wblk * 256 + (kb / BIG_NB) * 1024;
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // K0 = first column of the left-looking range: the super-panel's first column, or -- fast mode, the first two blocks
  // of a super-panel (FlowTask::p2, build_flow_levels) -- the PREVIOUS super-panel's, whose update then skips these 64
  // columns: the step after a super-panel's end does not wait for a tile of that update (6.8 us on the chain).  The
  // next diagonal block's range starts `skip` blocks later (the second block of such a super-panel forms the third,
  // whose columns the previous update did cover).
  const int skip = K0raw >> 24, K0 = K0raw & 0xffffff;
  const int super_end = min(K0 + BIG_SUPER, m.nc);   // (exact mode only: K0 is the super-panel's first column there)
  const int kn = kb + BIG_NB;
  const int blk = kb / BIG_NB, q = (kb - K0) / BIG_NB, sp = K0 / BIG_SUPER;
  // this wave also prepares the next diagonal block (nb == 32 then); across the super-panel's end only in the fast mode
  const bool look = rowblk == 0 && kn < (fa.exact ? super_end : m.nc);
  // flags of block (blk - j) that block's term of the update needs: my rows there (a partial last block is not
  // aligned with the whole blocks' row blocks and straddles two) and the rows of the diagonal block (row block j - 1)
  auto block_flag = [&](int j, int which) -> const unsigned * {   // which: 0, 1 = my rows; 2 = the diagonal block's rows
    if (which == 2) return fa.flags + ff.pf + (blk - j) * ff.pstride + (j - 1);
    const int r0p = kb - 32 * j + 32;
    const int rb = (R0 - r0p) / 32 + which;
    return rb <= (min(R0 + 31, M - 1) - r0p) / 32 ? fa.flags + ff.pf + (blk - j) * ff.pstride + rb : nullptr;
  };
  {
    // ---- first wait: the C tiles (previous super-panel's trailing update) and every block but the newest
    const unsigned *fp = nullptr;
    if (lane < 3 * (FLOW_MAX_BACK - 1)) {
      const int j = lane / 3 + 2;   // blocks blk - 2, blk - 3, ... (a long range: not blk - 2, see wait_second)
      if (j <= q && (j > 2 || q <= 3)) fp = block_flag(j, lane % 3);
    } else if (lane < 3 * (FLOW_MAX_BACK - 1) + 2) {
      const int bx = (R0 - K0) / TS + (lane - 3 * (FLOW_MAX_BACK - 1)), bxe = (min(R0 + 31, M - 1) - K0) / TS;
      if (sp > 0 && bx <= bxe) fp = fa.flags + ff.uf + (sp - 1) * ff.ustride + flow_tri(bx, (kb - K0) / TS);
    } else if (lane == 3 * (FLOW_MAX_BACK - 1) + 2) {
      const int d = (kn - K0) / TS;   // the update before the next block's range (a later range: look_tile below)
      if (look && skip == 0 && K0 > 0) fp = fa.flags + ff.uf + (K0 / BIG_SUPER - 1) * ff.ustride + flow_tri(d, d);
    } else if (lane == 63) {
      fp = built;   // cross-level form: the front's pivot columns (the C tiles below are read from them)
    }
    flow_wait(fp, fa.err, fa.wait_ticks);
  }
  RRPGO_FLOW_MARK(fa, ticket, tid >> 6, 1);
  constexpr uint32_t SZ = (uint32_t)sizeof(T);
  const Sc1Buf<T> fbuf(F, (uint32_t)((int64_t)M * M * (int64_t)sizeof(T))), wbuf(Wt, 1024u * SZ);
  auto ldF = [&](int col, int row) { return fbuf.ld((uint32_t)(col * M + row) * SZ); };   // element (row, col) of the front
  int irow[2];
  irow[0] = min(R0 + li, M - 1);
  irow[1] = min(R0 + 16 + li, M - 1);
  const int nblk = q;
  const int arow0 = kb + min(li, nb - 1), arow1 = kb + min(16 + li, nb - 1);
  const T am0 = li < nb ? (T)-1 : (T)0, am1 = 16 + li < nb ? (T)-1 : (T)0;
  T av[2][8][2], bv[2][8][2];
  const uint32_t colb = (uint32_t)((K0 + lk) * M) * SZ;
  const uint32_t oa0 = colb + (uint32_t)arow0 * SZ, oa1 = colb + (uint32_t)arow1 * SZ;
  const uint32_t ob0 = colb + (uint32_t)irow[0] * SZ, ob1 = colb + (uint32_t)irow[1] * SZ;
  const uint32_t kstep = (uint32_t)(4 * M) * SZ;
  auto fetch = [&](int b, T (*xa)[2], T (*xb)[2]) {
    uint32_t d = (uint32_t)(b * 8) * kstep;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      xa[s4][0] = fbuf.ld(oa0 + d) * am0;
      xa[s4][1] = fbuf.ld(oa1 + d) * am1;
      xb[s4][0] = fbuf.ld(ob0 + d);
      xb[s4][1] = fbuf.ld(ob1 + d);
      d += kstep;
    }
  };
  // The newest block's operand from the DIAGONAL block's rows is the X block the previous step's chain wave has just formed: the one
  // hand-off every wave of this step waits for.  It does not come through F (store, drain, flag, poll, load) but through a copy the
  // chain wave stores FIRST (xnew, below) and that is polled in place like W: k_flow_reset marked it, each entry goes from the mark
  // to its value in one store (r05: the drain + flag hop were ~1.7 us of every chain step).
  const Sc1Buf<T> xbuf(fa.xnew + (int64_t)m.wblk * 256 + (int64_t)max(blk - 1, 0) * 1024, 1024u * SZ);
  auto fetch_newest = [&](int b, T (*xa)[2], T (*xb)[2]) {
    uint32_t d = (uint32_t)(b * 8) * kstep;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      xb[s4][0] = fbuf.ld(ob0 + d);
      xb[s4][1] = fbuf.ld(ob1 + d);
      d += kstep;
    }
    const uint32_t r0 = (uint32_t)min(li, nb - 1), r1 = (uint32_t)min(16 + li, nb - 1);
    unsigned long long xt0 = 0;
    unsigned low_only = 0;
    for (unsigned spins = 0;; spins++) {
      bool unset = false, hi_unset = false;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s4 = 0; s4 < 8; s4++) {
        const uint32_t c = (uint32_t)(4 * s4 + lk) * 32u;
        xa[s4][0] = xbuf.ld((c + r0) * SZ);
        xa[s4][1] = xbuf.ld((c + r1) * SZ);
        unset = unset || flow_w_unset(xa[s4][0]) || flow_w_unset(xa[s4][1]);
        hi_unset = hi_unset || flow_w_hi_unset(xa[s4][0]) || flow_w_hi_unset(xa[s4][1]);
      }
      if (!__any(unset)) break;
      if (!__any(hi_unset) && ++low_only > flow_w_patience) break;
      if ((spins & 63u) == 63u) {
        if (xt0 == 0) xt0 = wall_clock64();
        const int e = __hip_atomic_load(fa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e != 0 || wall_clock64() - xt0 > fa.wait_ticks) {
          if (e == 0 && lane == 0) atomicOr(fa.err, DEVERR_FLOW_TIMEOUT);
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      xa[s4][0] *= am0;
      xa[s4][1] *= am1;
    }
  };
  auto wait_newest = [&] {   // X of block blk - 1, my rows (the diagonal block's rows: fetch_newest)
    flow_wait(lane < 2 ? block_flag(1, lane) : nullptr, fa.err, fa.wait_ticks);
  };
  // a range of more than three blocks reaches into the previous super-panel: its blocks are there long before block
  // blk - 2 is, and at ~2 us of loads per block the pre-work has to start on them at once to stay off the chain
  auto wait_second = [&] {
    if (q > 3) flow_wait(lane < 3 ? block_flag(2, lane) : nullptr, fa.err, fa.wait_ticks);
  };
  typename MM::Acc acc[2][2], nxt[2][2];
#pragma unroll
  for (int jb = 0; jb < 2; jb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int j = 16 * jb + MM::row(lane, r);
#pragma unroll
      for (int ib = 0; ib < 2; ib++) acc[ib][jb][r] = ldF(kb + min(j, nb - 1), irow[ib]);
    }
  if (nblk == 1) { wait_newest(); fetch_newest(0, av[0], bv[0]); }
  else if (nblk > 0) fetch(0, av[0], bv[0]);
  auto load_next_diag = [&] {
#pragma unroll
    for (int jb = 0; jb < 2; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cn = min(kn + 16 * jb + MM::row(lane, r), M - 1);
#pragma unroll
        for (int ib = 0; ib < 2; ib++) nxt[ib][jb][r] = ldF(cn, irow[ib]);
      }
  };
  if (look && skip == 0) load_next_diag();
#pragma unroll
  for (int jb = 0; jb < 2; jb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int j = 16 * jb + MM::row(lane, r);
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        const T v = pin(acc[ib][jb][r]);
        acc[ib][jb][r] = j < nb ? v : (T)0;
      }
    }
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
#pragma unroll
  for (int b = 0; b < FLOW_MAX_BACK; b++) {
    if (b < nblk) {
      const int slot = b & 1;
      if (b + 1 < nblk) {   // the next block's operands are in flight under this block's MFMAs
        if (b + 2 == nblk) wait_newest();
        if (b + 3 == nblk) wait_second();
        if (b + 2 == nblk) fetch_newest(b + 1, av[slot ^ 1], bv[slot ^ 1]);
        else fetch(b + 1, av[slot ^ 1], bv[slot ^ 1]);
      }
      if (look && skip > 0 && b + 1 == skip) {
        // the next diagonal block's range starts with the next block; the tile of the update before it that holds the
        // diagonal block is finished late (it needs the previous super-panel's last X): waited for now, not before the
        // pre-work, and loaded together with the newest block's operands
        const int Kn = K0 + BIG_NB * skip, d = (kn - Kn) / TS;
        flow_wait(lane == 0 ? fa.flags + ff.uf + (Kn / BIG_SUPER - 1) * ff.ustride + flow_tri(d, d) : nullptr, fa.err, fa.wait_ticks);
        load_next_diag();
      }
#pragma unroll
      for (int s4 = 0; s4 < 8; s4++) {
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
          for (int jb = 0; jb < 2; jb++) acc[ib][jb] = MM::mma(av[slot][s4][jb], bv[slot][s4][ib], acc[ib][jb]);
        if (look && b >= skip) {
#pragma unroll
          for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int jb = 0; jb <= ib; jb++) nxt[ib][jb] = MM::mma(-bv[slot][s4][jb], bv[slot][s4][ib], nxt[ib][jb]);
        }
      }
    }
  }
  // ---- now the diagonal block: wait for its inverse
  // (polled in place: k_flow_reset marked the block, see there; bounded like flow_wait)
  T wv[3][4];
  unsigned long long wt0 = 0;
  unsigned low_only = 0;
  for (unsigned spins = 0;; spins++) {
    bool unset = false, hi_unset = false;
    asm volatile("" ::: "memory");   // the raw buffer loads below are not volatile accesses: nothing else keeps them inside the poll
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cb = t == 0 ? 0 : 1, jb = t == 2 ? 1 : 0;
        wv[t][r] = wbuf.ld((uint32_t)((16 * jb + MM::row(lane, r)) * 32 + 16 * cb + li) * SZ);
        unset = unset || flow_w_unset(wv[t][r]);
        hi_unset = hi_unset || flow_w_hi_unset(wv[t][r]);
      }
    if (!__any(unset)) break;
    if (!__any(hi_unset) && ++low_only > flow_w_patience) break;
    if ((spins & 63u) == 63u) {
      if (wt0 == 0) wt0 = wall_clock64();
      const int e = __hip_atomic_load(fa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || wall_clock64() - wt0 > fa.wait_ticks) {
        if (e == 0 && lane == 0) atomicOr(fa.err, DEVERR_FLOW_TIMEOUT);
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  RRPGO_FLOW_MARK(fa, ticket, tid >> 6, 2);
  typename MM::Acc out[2][2];
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    out[ib][0] = typename MM::Acc{0, 0, 0, 0};
    out[ib][1] = typename MM::Acc{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      out[ib][0] = MM::mma(wv[0][r], acc[ib][0][r], out[ib][0]);
      out[ib][1] = MM::mma(wv[1][r], acc[ib][0][r], out[ib][1]);
      out[ib][1] = MM::mma(wv[2][r], acc[ib][1][r], out[ib][1]);
    }
  }
  if (nb == BIG_NB && R0 + 32 <= M) {
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t co = (uint32_t)((kb + 16 * cb + MM::row(lane, r)) * M + R0 + li) * SZ;
        fbuf.st(co, (T)out[0][cb][r]);
        fbuf.st(co + 16u * SZ, (T)out[1][cb][r]);
      }
  } else {
#pragma unroll
    for (int ib = 0; ib < 2; ib++)
#pragma unroll
      for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int c = 16 * cb + MM::row(lane, r), i = R0 + 16 * ib + li;
          if (i < M && c < nb) fbuf.st((uint32_t)((kb + c) * M + i) * SZ, (T)out[ib][cb][r]);
        }
  }
  unsigned *pflag = fa.flags + ff.pf + blk * ff.pstride + rowblk;
  if (look) {
    // the copy the next step polls (all 1024 entries: rows past the front's end hold clamped, finite values nobody uses unmasked)
    const Sc1Buf<T> xout(fa.xnew + (int64_t)m.wblk * 256 + (int64_t)blk * 1024, 1024u * SZ);
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t co = (uint32_t)((16 * cb + MM::row(lane, r)) * 32 + li) * SZ;
        xout.st(co, (T)out[0][cb][r]);
        xout.st(co + 16u * SZ, (T)out[1][cb][r]);
      }
  }
  if (!look) {
    flow_drain();
    if (lane == 0) flow_flag_set(pflag);
    RRPGO_FLOW_MARK(fa, ticket, tid >> 6, 3);
    return;
  }
  // next diagonal block: the last 32 columns of its update are this wave's own X
#pragma unroll
  for (int cb = 0; cb < 2; cb++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int ib = 0; ib < 2; ib++)
#pragma unroll
        for (int jb = 0; jb <= ib; jb++) nxt[ib][jb] = MM::mma(-out[jb][cb][r], out[ib][cb][r], nxt[ib][jb]);
  const int nbn = min(BIG_NB, m.nc - kn);
  sh_image_from_acc<T>(Sh, nxt, nbn);
  diag32_init_tables<T>(Sh);
  // X of this row block is what the NEXT step's pre-work reads (these rows are its diagonal block's): it is published from INSIDE the
  // factorisation, after the first 16 x 16 sweep -- its stores (written through to memory: ~1 us) have landed by then, and the chain
  // has not stood still for them (r05: the drain used to sit here, in front of the factorisation)
  wave_sync();
  auto publish_x = [&] {
    flow_drain();
    if (lane == 0) flow_flag_set(pflag);
  };
  // the factored block itself is not stored: nothing reads a diagonal block of L once its W exists (the back
  // substitution works with the inverses), and across a super-panel's end tile (0, 0) of the update still owns it
  diag32_factor_invert<T, false, true, decltype(publish_x)>(Sh, nbn, F + (int64_t)kn * M + kn, M, fa.winv + (int64_t)m.wblk * 256 + (kn / BIG_NB) * 1024, fa.err,
                                                            false, true, publish_x);
  flow_drain();   // W is in memory
  if (lane == 0) flow_flag_set(fa.flags + ff.wf + blk + 1);
  RRPGO_FLOW_MARK(fa, ticket, tid >> 6, 3);
}

// Tickets and completion flags are zeroed by a KERNEL at the start of every factorisation, not by a memset node: inside
// the captured stage graphs of a sharded handle a hipMemsetAsync node was not ordered before the kernels behind it on
// replay (r03: the second replay of a stage found the first one's flags still set -- tasks did not wait, the ticket
// was past the end -- while eager launches and the unsharded graph, where hundreds of microseconds of other kernels
// sit between the two, were fine).
// It also marks the W blocks of the range's flow fronts as NOT THERE YET (workgroups flag_wgs.. = entries of `wfill`: offset
// and count of scalars in winv, four blocks each): a PANEL wave polls its W block itself instead of a flag followed by the block (one
// round trip to L2 on the chain instead of two).  The mark is the all-ones pattern, a NaN no arithmetic produces
// (hardware NaNs are 0x7fc00000 / propagated payloads of the operands); every store of a W block writes all of its
// 1024 entries, each entry goes from the mark to its value in one store, so a block without a mark is complete.
template <typename T>
__global__ void __launch_bounds__(256) k_flow_reset(unsigned *words, int64_t n, int flag_wgs, T *winv, const int64_t *wfill, T *xnew) {
  if ((int)blockIdx.x < flag_wgs) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)flag_wgs * 256) words[i] = 0u;
  } else {
    // one entry of `wfill` per workgroup: offset and count (<= 4096, a multiple of 1024) of scalars in winv
    const int e = (int)blockIdx.x - flag_wgs;
    const int64_t off = wfill[2 * e], cnt = wfill[2 * e + 1];
    using U = typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type;
    U *w = reinterpret_cast<U *>(winv + off), *xw = reinterpret_cast<U *>(xnew + off);
    for (int64_t i = threadIdx.x; i < cnt; i += 256) { w[i] = ~(U)0; xw[i] = ~(U)0; }   // (the chain waves' X blocks are polled in place like W)
  }
}
#ifndef RRPGO_FLOW_DEPTH
#define RRPGO_FLOW_DEPTH 4   // k-chunks of a trailing-update tile requested ahead of the MFMAs (the launch sequence's k_big_update uses 1 at
#endif                       // six workgroups per CU; here two workgroups per CU have to cover a chunk's memory round trip themselves)
#ifndef RRPGO_FLOW_WAVES
#define RRPGO_FLOW_WAVES 2   // fp32: waves per SIMD the register allocation aims at (= workgroups per CU): the panel wave keeps ~200 values in flight;
#endif                       // fp64 (two registers per value) runs one workgroup per CU

// NT = MFMA tiles per wave and dimension of a trailing-update tile: 2 -> 64 x 64 tiles per task (levels of few tasks: more of
// them in flight), 4 -> 128 x 128 (levels with thousands of tiles: the flow kernel runs two workgroups per CU, and at
// that occupancy only the large tile -- sixteen accumulators per wave, four times the MFMAs per staged chunk -- keeps
// the matrix cores fed while the next chunk is on its way)
// XL: the cross-level form (r05) -- the list holds the tasks of EVERY level of fronts beyond LDS, level after level: BUILD tasks
// form a front's pivot columns once its children (fronts of earlier levels, earlier tickets) are done, a front's other tasks
// wait for its BUILD tasks, and whatever is gathered from a child is read past L1.  One launch instead of three per level;
// graphs of a few dozen big fronts (sphere2500: 46 in six levels) -- what k_factor_flow did for the LDS fronts.
template <typename T, int NT, bool XL = false> __global__ void __launch_bounds__(256, (sizeof(T) == 4 ? RRPGO_FLOW_WAVES : 1)) k_big_flow(FlowArgs<T> fa) {
  if (opt_stopped(fa.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  using MM = Mfma16<T>;
  using UT = UpdTile<T, NT>;
  constexpr int TS = UT::TILE;
  static_assert(UT::SMEM >= DIAG32_LDS, "one LDS region serves the tile staging and the diagonal-block images");
  __shared__ T smem[UT::SMEM];
  __shared__ unsigned s_ticket;
  for (;;) {
    // Everything a task derives from the thread index is derived from THIS copy: the compiler would otherwise hoist
    // the per-lane constants of both task bodies out of the ticket loop and keep them all live (k_big_flow needed
    // more than 256 VGPRs that way; the panel wave alone needs 180, the tile 80).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_ticket = atomicAdd(fa.ticket, 1u);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane((int)s_ticket);
    __syncthreads();   // everybody has read it (and is done with smem) before the next round rewrites it
    if (t >= fa.n_tasks) return;
    RRPGO_FLOW_MARK(fa, t, wave, 0);
    const FlowRec rec = fa.tasks[t];
    const FlowTask tk = rec.t;
    const int kind = tk.kind_front >> 24;
    const SnMeta m = rec.m;
    const FlowFront ff = rec.ff;
    if constexpr (XL) {
      if (kind == FLOW_BUILD) {
        flow_build_task<T>(fa, ff, m, tk.p0, tid, t);
        continue;
      }
    }
    // (cross-level form: the flag that says the front's pivot columns are built -- a PANEL wave reads them before any other wait)
    const unsigned *built = XL ? fa.flags + ff.bf + 1 : nullptr;
    if (kind == FLOW_PANEL) {
      flow_panel_wave<T, TS>(fa, ff, m, tk.p0, tk.p2, tk.p1 + wave, smem, tid, t, built);
      continue;
    }
    if (kind == FLOW_DIAG0) {
      if (wave == 0) {
        if constexpr (XL) flow_wait(lane == 0 ? built : nullptr, fa.err, fa.wait_ticks);
        flow_diag0_wave<T>(fa, ff, m, smem, tid);
      }
      RRPGO_FLOW_MARK(fa, t, wave, 3);
      continue;
    }
    // ---- UPDATE: tile (bx, by) of the trailing update of the super-panel at K0
    const int K0 = tk.p0, bx = tk.p1, by = tk.p2;
    const int M = m.nc + m.nr + 1;
    const int ke = min(K0 + BIG_SUPER, m.nc), sp = K0 / BIG_SUPER;
    const int t0 = ke;
    const int I0 = t0 + bx * TS, J0 = t0 + by * TS;
    // with the Schur complement left to k_big_schur the last super-panel only owes the strip [nc, origin) (see k_big_update)
    const int jmax = (fa.schur_tile && ke == m.nc) ? min(big_schur_origin(m.nc, fa.schur_tile), M) : M;
    if (wave == 0) {
      // X of the rows of both operand strips for every 32-column block of the super-panel: lane = block (4) x strip (2)
      // x row block (up to TS / 32 + 1 = 5 of them when the strip is not aligned with the block's row blocks)
      {
        const unsigned *fp = nullptr;
        const int qq = lane >> 4, strip = (lane >> 3) & 1, k = lane & 7;
        const int kb = K0 + 32 * qq;
        if (kb < ke) {
          const int nbq = min(BIG_NB, m.nc - kb), r0 = kb + nbq;
          const int lo = strip ? J0 : I0, hi = min(lo + TS - 1, M - 1);
          const int rb = (lo - r0) / 32 + k;
          if (rb <= (hi - r0) / 32) fp = fa.flags + ff.pf + (kb / BIG_NB) * ff.pstride + rb;
        }
        flow_wait(fp, fa.err, fa.wait_ticks);
      }
      if (sp > 0) {   // the tiles of the previous super-panel's update under this one (its grid starts at K0)
        const unsigned *fp = nullptr;
        if (lane < 4) {
          const int o = K0;
          const int bxl = (I0 - o) / TS, bxh = (min(I0 + TS - 1, M - 1) - o) / TS, byl = (J0 - o) / TS, byh = (min(J0 + TS, jmax) - 1 - o) / TS;
          const int pbx = bxl + (lane >> 1), pby = byl + (lane & 1);
          if (pbx <= bxh && pby <= byh && pby <= pbx) fp = fa.flags + ff.uf + (sp - 1) * ff.ustride + flow_tri(pbx, pby);
        }
        flow_wait(fp, fa.err, fa.wait_ticks);
      }
    }
    __syncthreads();
    RRPGO_FLOW_MARK(fa, t, wave, 1);
    T *F = fa.lvals + m.loff;
    typename MM::Acc acc[NT][NT];
    TileGather<T> tg{nullptr, -1, nullptr, nullptr, nullptr, nullptr};
    if (fa.gather && K0 == 0 && J0 >= big_built_cols(m.nc, M)) tg = TileGather<T>{fa.child_meta + m.child_begin, m.child_count, fa.scat, fa.lvals, fa.uvals, fa.xch};
    const bool have = big_update_tile<T, NT, RRPGO_FLOW_DEPTH, true, false, XL>(F, M, K0, ke, jmax, I0, J0, smem, acc, nullptr, false, tg, tid);
    RRPGO_FLOW_MARK(fa, t, wave, 2);
    flow_drain();
    __syncthreads();   // every wave's part of the tile is in memory
    if (tid == 64) {
      flow_flag_set(fa.flags + ff.uf + sp * ff.ustride + flow_tri(bx, by));
      if constexpr (XL) __hip_atomic_fetch_add(fa.flags + ff.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one more of the front's UPDATE tasks: its parent counts them
    }
    // exact mode, tile (0, 0): its first wave holds the next super-panel's first diagonal block: factor and invert it here
    if (fa.exact && bx == 0 && by == 0 && t0 < m.nc && wave == 0 && have) {
      T *Sh = smem;
      const int nbn = min(BIG_NB, m.nc - t0);
      if constexpr (NT == 2) sh_image_from_acc<T>(Sh, acc, nbn);
      else {
        const typename MM::Acc corner[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
        sh_image_from_acc<T>(Sh, corner, nbn);
      }
      diag32_init_tables<T>(Sh);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      diag32_factor_invert<T, false, true>(Sh, nbn, F + (int64_t)t0 * M + t0, M, fa.winv + (int64_t)m.wblk * 256 + (int64_t)(t0 / BIG_NB) * 1024, fa.err,
                                           false, true);
      flow_drain();
      if (lane == 0) flow_flag_set(fa.flags + ff.wf + t0 / BIG_NB);
    }
    RRPGO_FLOW_MARK(fa, t, wave, 3);
  }
}

// ---- k_big_solve_flow: the back substitution of a level's WIDE pivot blocks as ONE launch ---------------------------
// k_big_solve_sp is one launch per 128-column super-panel, right to left: workgroup 0 of a front solves the super-panel,
// the others fold the solution of the super-panel to its right into the columns further left -- 12 dependent launches of
// 7.9 us for the lattice's root, 41 per iteration in all.  Here the same steps are tasks of one launch, drawn from a
// ticket like k_big_flow's: CHAIN(front, step) solves a super-panel and publishes its x, FOLD(front, group, step) applies
// the x of the super-panel to the right to 64 columns further left.  Two counters per front order them: xdone
// (super-panels solved) and fdone[g] (folds group g has applied, incl. the initialisation at step 0); a chain step
// waits for the two groups that cover its 128 columns -- their folds used the x of the step before last, so they have
// had a whole chain step.  The list is ordered by step (chain first, then the folds nearest to the diagonal), so every
// task waits only for smaller tickets: no deadlock whatever the grid size or what else runs on the GPU (several handles
// may replay such launches side by side).  Same arithmetic, same order of every sum as k_big_solve_sp: bit-identical.
// Payload (x, 128 values per step) with sc1 accesses, counters with agent-scope atomics after a drain; bounded spins.
struct SolveFlowFront { int32_t xdone, fdone; };   // indices into the flag words: xdone; fdone + g
struct SolveFlowTask { int32_t front, ell, group; };   // group < 0: CHAIN
__device__ __forceinline__ bool solve_flow_wait(const unsigned *word, unsigned need, int *err, unsigned long long max_ticks) {
  bool ok = true;
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; spins++) {
    if (flow_flag_ld(word) >= need) break;
    if ((spins & 63u) == 63u) {
      if (t0 == 0) t0 = wall_clock64();
      const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || wall_clock64() - t0 > max_ticks) {
        if (e == 0) atomicOr(err, DEVERR_FLOW_TIMEOUT);
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return ok;
}
template <typename T> __global__ void __launch_bounds__(1024) k_big_solve_flow(FactorArgs<T> a, const T *part, int64_t N, int R, unsigned *flags,
                                                                             const SolveFlowFront *fronts, const SolveFlowTask *tasks, int n_tasks,
                                                                             unsigned *ticket) {
  if (opt_stopped(a.err)) return;
  __shared__ T xf[BIG_SUPER];              // this super-panel: t, then x
  __shared__ T Ws[4 * 32 * 33];            // the super-panel's four W_b, staged transposed
  __shared__ unsigned s_ticket;
  constexpr int NW = 16;
  constexpr uint32_t SZ = (uint32_t)sizeof(T);
  for (;;) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const int tk = __builtin_amdgcn_readfirstlane((int)s_ticket);
    __syncthreads();
    if (tk >= n_tasks) return;
    const SolveFlowTask task = tasks[tk];
    const SnMeta m = a.task_meta[a.task_begin + task.front];
    const SolveFlowFront sf = fronts[task.front];
    const int nc = m.nc, M = nc + m.nr + 1;
    const int S = (nc + BIG_SUPER - 1) / BIG_SUPER;
    const T *Lg = a.lvals + m.loff;
    T *xg = a.x + m.col0;
    const Sc1Buf<T> xbuf(xg, (uint32_t)nc * SZ);
    auto t_init = [&](int j) {   // y1 - L21^T x[rows], slices subtracted in slice order
      T t = Lg[(int64_t)j * M + (M - 1)];
      if (m.nr > 0)
        for (int r = 0; r < R; r++) t -= part[(int64_t)r * N + m.col0 + j];
      return t;
    };
    unsigned *xdone = flags + sf.xdone;
    const int ell = task.ell, sp = S - 1 - ell;
    const int K0 = BIG_SUPER * sp, K1 = min(nc, K0 + BIG_SUPER), K2 = min(nc, K1 + BIG_SUPER);
    const int nright = K2 - K1;   // rows of the super-panel to the right (ell > 0)
    if (task.group >= 0) {
      // ---- FOLD: 64 columns left of the current super-panel, 4 per wave
      const int g = task.group, c_lo = 64 * g, c_hi = min(K0, c_lo + 64);
      unsigned *fdone = flags + sf.fdone + g;
      if (ell == 0) {
        for (int j = c_lo + tid; j < c_hi; j += 1024) xbuf.st((uint32_t)j * SZ, t_init(j));
      } else {
        // the fold operands do not depend on the solution: request them before the wait
        T la[4], lb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = min(c_lo + wave + NW * q, c_hi - 1);
          const T *col = Lg + (int64_t)i * M + K1;
          la[q] = col[min(lane, nright - 1)];
          lb[q] = col[min(lane + 64, nright - 1)];
        }
        if (wave == 0) {
          solve_flow_wait(xdone, (unsigned)ell, a.err, a.wait_ticks);   // x of the super-panel to the right
          solve_flow_wait(fdone, (unsigned)ell, a.err, a.wait_ticks);   // this group's previous fold (another workgroup may have run it): the sums go in step order
        }
        __syncthreads();
        const T va = xbuf.ld((uint32_t)(K1 + min(lane, nright - 1)) * SZ), vb = xbuf.ld((uint32_t)(K1 + min(lane + 64, nright - 1)) * SZ);
        const T xa = lane < nright ? va : (T)0, xb = lane + 64 < nright ? vb : (T)0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = c_lo + wave + NW * q;
          const T sum = wave_sum63<T>(la[q] * xa + lb[q] * xb);
          if (lane == 63 && i < c_hi) xbuf.st((uint32_t)i * SZ, xbuf.ld((uint32_t)i * SZ) - sum);
        }
      }
      flow_drain();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(fdone, (unsigned)(ell + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // ---- CHAIN: this super-panel.  Every load whose address does not depend on a solution is requested up front
    const T *Wb = a.winv + (int64_t)m.wblk * 256;
    const bool have_right = sp + 1 < S;
    const int w = K1 - K0;
    const int b_hi = (K1 + 31) / 32 - 1, b_lo = K0 / 32;
    T fa_[8], fb_[8];
    if (have_right) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const T *col = Lg + (int64_t)(K0 + min(wave + NW * c, w - 1)) * M + K1;
        fa_[c] = col[min(lane, nright - 1)];
        fb_[c] = col[min(lane + 64, nright - 1)];
      }
    }
    T wreg[4], lv[4][3];
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const int b = max(b_hi - bb, b_lo);
      wreg[bb] = Wb[(int64_t)b * 1024 + tid];
      const int c0 = 32 * b, cw = min(32, nc - c0), ncols = c0 - K0;
      const T *base = Lg + c0 + min(l32, cw - 1);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int i = 2 * (wave + NW * q) + half;
        lv[bb][q] = base[(int64_t)(K0 + min(i, max(ncols - 1, 0))) * M];
      }
    }
    // the previous super-panel's x, and the folds of the steps before last into my 128 columns (groups K0 / 64, K0 / 64 + 1)
    if (ell > 0 && wave == 0) {
      solve_flow_wait(xdone, (unsigned)ell, a.err, a.wait_ticks);
      solve_flow_wait(flags + sf.fdone + K0 / 64, (unsigned)ell, a.err, a.wait_ticks);
      if (K0 + 64 < K1) solve_flow_wait(flags + sf.fdone + K0 / 64 + 1, (unsigned)ell, a.err, a.wait_ticks);
    }
    __syncthreads();
    T xa = 0, xb = 0;
    if (have_right) {
      const T va = xbuf.ld((uint32_t)(K1 + min(lane, nright - 1)) * SZ), vb = xbuf.ld((uint32_t)(K1 + min(lane + 64, nright - 1)) * SZ);
      xa = lane < nright ? va : (T)0;
      xb = lane + 64 < nright ? vb : (T)0;
    }
    if (tid < w) xf[tid] = ell == 0 ? t_init(K0 + tid) : xbuf.ld((uint32_t)(K0 + tid) * SZ);
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const int c = tid >> 5, j = tid & 31;
      Ws[bb * (32 * 33) + j * 33 + c] = wreg[bb];
    }
    __syncthreads();
    if (have_right) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int i = wave + NW * c;
        const T sum = wave_sum63<T>(fa_[c] * xa + fb_[c] * xb);
        if (lane == 63 && i < w) xf[i] -= sum;
      }
      __syncthreads();
    }
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const int b = b_hi - bb;
      if (b < b_lo) break;   // uniform
      const int c0 = 32 * b, cw = min(32, nc - c0), o = c0 - K0, ncols = c0 - K0;
      if (tid < 64) {
        const T *ws = Ws + bb * (32 * 33) + l32;
        const T tv = pin(xf[o + min(l32, cw - 1)]);
        const T v = l32 < cw ? tv : (T)0;
        T wv[32];
#pragma unroll
        for (int j = 0; j < 32; j++) wv[j] = ws[j * 33];
        T x = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) x += wv[j] * lane_bcast(v, j);
        if (lane < cw) xf[o + lane] = x;
      }
      __syncthreads();
      if (ncols > 0) {   // uniform
        const T xv = pin(xf[o + min(l32, cw - 1)]);
        const T xj = l32 < cw ? xv : (T)0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int i = 2 * (wave + NW * q) + half;
          const T sum = half_wave_sum<T>(lv[bb][q] * xj);
          if (l32 == 31 && i < ncols) xf[i] -= sum;
        }
        __syncthreads();
      }
    }
    if (tid < w) xbuf.st((uint32_t)(K0 + tid) * SZ, xf[tid]);
    flow_drain();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(xdone, (unsigned)(ell + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace rrpgo
