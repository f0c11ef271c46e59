// kernels.hip.h -- gfx950 device code for the pose-graph Gauss-Newton path.
//
//   k_linearize      error + Jacobians + J^T W J / J^T W e + chi2     (reference :434-486,165-192,537-574)
//   k_factor_tasks   multifrontal supernodal Cholesky, fronts in LDS  (replaces umfpack.factorize, :138)
//   k_solve_tasks    back substitution down the supernode tree         (replaces umfpack.solve, :141)
//   k_update         update_nodes + |dx|^2                             (reference :229-245,273)
//   k_finalize_slot  fixed-order reduction of the chi2 / |dx|^2 partials (pgo_api.hip)
//   k_big_* / k_solve_mid / k_big_flow (flow.hip.h)  fronts beyond LDS (see "huge fronts" below)
//   k_linearize_se3 / k_update_se3, k_pack_boundary / k_pack_shared / k_sum_shared   SE(3), sharding over ranks
//
// Wavefront = 64 lanes.  Cross-workgroup dependencies are kernel boundaries on one stream -- except inside the dataflow
// launches (k_factor_flow / k_solve_flow, lds_flow.hip.h; k_big_flow / k_big_solve_flow, flow.hip.h), whose workgroups hand
// update matrices, tiles and solutions to each other behind flags: sc1 payload, agent-scope flag, bounded waits (dep_wait
// below, flow_wait there).  Everywhere else a workgroup only reads what it wrote itself or what an earlier launch wrote.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#ifndef RRPGO_CHAIN_PRIO
#define RRPGO_CHAIN_PRIO 3   // s_setprio of the wave that carries a workgroup's dependent chain
#endif

namespace rrpgo {

constexpr unsigned X_PENDING_WORD = 0x7ff8deadu;   // see x_wait

template <typename T> struct VecT;
template <> struct VecT<float> { using V4 = float4; using V2 = float2; };
template <> struct VecT<double> { using V4 = double4; using V2 = double2; };

#ifndef RRPGO_LIN_GROUP
#define RRPGO_LIN_GROUP 8
#endif
constexpr int LIN_GROUP = RRPGO_LIN_GROUP;   // lanes cooperating on one node in k_linearize (power of two <= 64)
constexpr int LIN_THREADS = 256;
constexpr int UPD_THREADS = 256;

// Diagnostic build only (-DRRPGO_TRACE with -DRRPGO_STAMPS): first workgroup of a big-front launch
// appends (tag, wall clock at entry) to a trace region behind the stamps.
#if defined(RRPGO_TRACE) && defined(RRPGO_STAMPS)
#define RRPGO_TRACE_MARK(a, tag)                                                                       \
  do {                                                                                                  \
    if ((a).trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {       \
      const unsigned long long i_ = atomicAdd((a).trace, 1ull);                                         \
      if (i_ < 190000) { (a).trace[2 + 2 * i_] = (tag); (a).trace[3 + 2 * i_] = wall_clock64(); }        \
    }                                                                                                   \
  } while (0)
#else
#define RRPGO_TRACE_MARK(a, tag) do { } while (0)
#endif

#if defined(RRPGO_TRACE) && defined(RRPGO_STAMPS)
#define RRPGO_PHASE_MARK(a, cond, tag)                                                                  \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if ((a).trace && (cond) && threadIdx.x == 0 && blockIdx.y == 0) {                                   \
      const unsigned long long i_ = atomicAdd((a).trace, 1ull);                                         \
      if (i_ < 190000) { (a).trace[2 + 2 * i_] = (tag); (a).trace[3 + 2 * i_] = clock64(); }            \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  } while (0)
#else
#define RRPGO_PHASE_MARK(a, cond, tag) do { } while (0)
#endif

// Diagnostic build only (-DRRPGO_STAMPS): thread 0 of a workgroup records the
// 100 MHz wall clock at the phase boundaries of every front it processes.
#ifdef RRPGO_STAMPS
#define RRPGO_STAMP(a, s, slot)                                                     \
  do {                                                                               \
    if (threadIdx.x == 0 && (a).stamps) (a).stamps[(int64_t)(s) * 12 + (slot)] = wall_clock64(); \
  } while (0)
#define RRPGO_ACC_DECL() unsigned long long acc_t0_ = 0
#define RRPGO_ACC_BEGIN() acc_t0_ = wall_clock64()
#define RRPGO_ACC_END(ptr, slot)                                                     \
  do {                                                                               \
    if (threadIdx.x == 0 && (ptr)) (ptr)[slot] += wall_clock64() - acc_t0_;         \
  } while (0)
#ifdef RRPGO_STAMPS_SOLVE   // the back substitution reuses slots 0..4: stamp it OR the factorisation
#define RRPGO_STAMP_SOLVE(a, s, slot) RRPGO_STAMP(a, s, slot)
#else
#define RRPGO_STAMP_SOLVE(a, s, slot) do { } while (0)
#endif
#else
#define RRPGO_STAMP(a, s, slot) do { } while (0)
#define RRPGO_STAMP_SOLVE(a, s, slot) do { } while (0)
#define RRPGO_ACC_DECL() do { } while (0)
#define RRPGO_ACC_BEGIN() do { } while (0)
#define RRPGO_ACC_END(ptr, slot) do { } while (0)
#endif

// The wave's index in its workgroup as a SCALAR: `threadIdx.x >> 6` lives in a vector register, and every
// tile index, column offset and loop bound derived from it would be 64-lane VALU arithmetic.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// device-side error flags (sticky, read by the host at sync points)
enum : int { DEVERR_NOT_SPD = 1, DEVERR_FLOW_TIMEOUT = 2 };

// ---- the loop state of ONE rr_pgo_optimize call, on the device (reference pose_graph_optimization.rs:247-303).
// The host enqueues iterations ahead of the device and never synchronises inside the loop: the stop rule (:298-300), the
// Levenberg-Marquardt accept / reject (:275-282) and the failure of a factorisation (:271) are decided here, by the
// kernel that finishes an iteration, which also publishes the iteration's (chi2, |dx|) in a ring the host polls
// (pinned, host-coherent memory).  The STOP WORD is the int behind the sticky error flag (err[1]): once it is set,
// every later launch of the same call returns at its first instruction (the compute kernels) or leaves the state alone
// (k_update), so an iteration that was enqueued before the host saw the stop costs a handful of empty launches.
struct OptCtrl {
  double lambda;        // :254; x 2 on reject, / 2 on accept
  double last_error;    // :255, :284
  double tolerance;     // :253
  unsigned long long seq;   // items of this call published so far
  int reject;           // Levenberg-Marquardt: the step just applied is to be undone (:276-279); read by the undo launch
  int pad;
};
struct alignas(32) OptSlot {   // one published item; `seq` (1-based index of the item) is stored last, system-scope release
  double chi2, norm;
  int flags, pad;
  unsigned long long seq;
};
enum : int { OPT_RING = 8 };
enum : int { OPT_STOP = 1,      // the stop rule fired in this item: it is the last iteration of the call
             OPT_SKIPPED = 2,   // the item found the stop word set: only its chi2 (of the final state) means anything
             OPT_ERR_SHIFT = 8  // flags >> 8 = the device error flag (DEVERR_*) when the item's factorisation failed
};
__device__ __forceinline__ void opt_publish(OptCtrl *c, OptSlot *ring, double chi2, double norm, int flags) {
  const unsigned long long n = c->seq + 1;
  OptSlot *s = ring + (n - 1) % OPT_RING;
  s->chi2 = chi2;
  s->norm = norm;
  s->flags = flags;
  __hip_atomic_store(&s->seq, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  c->seq = n;
}
// the stop word of the running rr_pgo_optimize call (0 outside one): `err` is the engine's two-int error block
__device__ __forceinline__ bool opt_stopped(const int *err) { return err && err[1] != 0; }


// TC = type of the state, the measurements and all factor arithmetic; T = type H and b are stored in
// (TC == T, or TC = double with T = float: "mixed" mode, exact gradient + single-precision factor)
// One SE(2) edge as ONE record (64 bytes in fp32, a 128-byte line in fp64): the pull form visits every edge from both
// endpoints, and out of five arrays a visit touched five partly used 64-byte sectors (r03, PMC: 294 MB per launch on the
// 1M-edge lattice)
template <typename TC> struct EdgeRec {
  int32_t from, to;
  int64_t slot;                        // (offset into hvals << 1) | transposed
  typename VecT<TC>::V4 meas;          // SE2: x, y, cos, sin | SE2_XY: x, y, 0, 0
  typename VecT<TC>::V4 info_a;        // i11 i12 i13 i22
  typename VecT<TC>::V2 info_b;        // i23 i33
};
static_assert(sizeof(EdgeRec<float>) == 64 && sizeof(EdgeRec<double>) == 128, "EdgeRec is four / eight sixteen-byte loads of one line");
template <typename T, typename TC = T> struct LinArgs {
  int n_nodes;
  const EdgeRec<TC> *e_rec;              // per edge: what the arrays below hold, as one record (k_linearize reads this)
  const typename VecT<TC>::V4 *pose;     // x, y, cos, sin  (XY landmarks: x, y, -, -)
  const int2 *e_idx;                     // from, to
  const typename VecT<TC>::V4 *e_meas;   // SE2: x, y, cos, sin | SE2_XY: x, y, 0, 0
  const typename VecT<TC>::V4 *e_info_a; // i11 i12 i13 i22
  const typename VecT<TC>::V2 *e_info_b; // i23 i33
  const int64_t *e_slot;                 // (offset into hvals << 1) | transposed
  const int32_t *inc_ptr;
  const int2 *inc_list;                  // per incidence: (edge << 4 | offdiag << 3 | owns << 2 | kind << 1 | role, the OTHER endpoint's node) --
                                         // the far pose is requested together with the edge record, not after it (sharded runs: see Engine)
  const int32_t *node_list;              // sharded runs: the nodes this rank linearises (own + shared), else null;
                                         // n_nodes is then the length of the list
  const uint8_t *node_dim;               // 3 (SE2) or 2 (XY)
  const int32_t *node_offset;            // reference scalar offset
  const int64_t *diag_off;
  T *hvals;
  T *b;                                  // reference scalar order, already negated (:361)
  double *chi2_partial;                  // one per workgroup
  int anchor;                            // node that gets the 1e7 prior (:330-336), -1 none
  TC lambda;                             // added to every diagonal entry when > 0 (LM, :362-366)
  int write_system;                      // 0: chi2 only
  const uint8_t *adds_diag;              // sharded runs: per node, 1 = this rank adds prior / lambda (shared nodes: rank 0)
  unsigned *zero_words;                  // flags and tickets of the dataflow launches that follow (lds_flow.hip.h): zeroed here,
  int n_zero_words;                      // one launch ahead of their first use (0: none)
  unsigned *fill_words;                  // the solution vector of k_solve_flow: every word set to X_PENDING_WORD here (a front
  int n_fill_words;                      // of that launch waits for its ancestors' entries themselves, x_wait) (0: none)
  OptCtrl *ctrl;                         // rr_pgo_optimize's device-side loop state (null outside one) ...
  int lambda_from_ctrl;                  // ... 1: lambda = ctrl->lambda (Levenberg-Marquardt: the host does not know it)
  int reset_ctrl;                        // ... 1: the first launch of a call: its first thread resets the loop state
  double reset_lambda, reset_tolerance;  //      (:253-254) and the stop word err[1] -- nothing in this launch reads either
  int *err;
  int publish;                           // ... 1: a Gauss-Newton iteration's linearisation: when the stop word is set, its LAST
                                         //      workgroup publishes chi2 (of the final state: the call's last error entry) at once --
                                         //      the host returns while the iteration's other launches drain empty;
                                         //      2: a chi2-only item: always published from here (and kept as OptCtrl::last_error)
  OptSlot *ring_host;
  int *blocks_done;                      // zero between launches
};
template <typename A> __device__ __forceinline__ void opt_reset_in_first_thread(const A &a) {
  if (a.reset_ctrl && blockIdx.x == 0 && threadIdx.x == 0) {
    a.ctrl->lambda = a.reset_lambda;
    a.ctrl->tolerance = a.reset_tolerance;
    a.ctrl->last_error = 0.0;
    a.ctrl->seq = 0ull;
    a.ctrl->reject = 0;
    a.err[1] = 0;
  }
}

// ---------------------------------------------------------------- factor maths

// Error and Jacobians of one 2D edge, zero padded to 3x3.
// kind 0: pose-pose (:434-447,457-486); kind 1: pose-landmark (:449-455,516-535).
template <typename T>
__device__ __forceinline__ void edge_linearize_2d(int kind, const typename VecT<T>::V4 &x1,
                                                  const typename VecT<T>::V4 &x2,
                                                  const typename VecT<T>::V4 &z, T e[3], T A[3][3],
                                                  T B[3][3]) {
  const T dx = x2.x - x1.x, dy = x2.y - x1.y;
  if (kind == 0) {
    // z^-1, x1^-1 (nalgebra Isometry::inverse), then (z^-1 * x1^-1) * x2
    const T zic = z.z, zis = -z.w;
    const T zitx = zic * (-z.x) - zis * (-z.y), zity = zis * (-z.x) + zic * (-z.y);
    const T xic = x1.z, xis = -x1.w;
    const T xitx = xic * (-x1.x) - xis * (-x1.y), xity = xis * (-x1.x) + xic * (-x1.y);
    const T mc = zic * xic - zis * xis, ms = zic * xis + zis * xic;
    const T mtx = zitx + (zic * xitx - zis * xity), mty = zity + (zis * xitx + zic * xity);
    const T etx = mtx + (mc * x2.x - ms * x2.y), ety = mty + (ms * x2.x + mc * x2.y);
    const T ec = mc * x2.z - ms * x2.w, es = mc * x2.w + ms * x2.z;
    e[0] = etx; e[1] = ety; e[2] = atan2(es, ec);
    // M = Rz^T R1^T ; a12 = M * (dy, -dx)
    const T a12x = mc * dy + ms * dx, a12y = ms * dy - mc * dx;
    A[0][0] = -mc; A[0][1] = ms;  A[0][2] = a12x;
    A[1][0] = -ms; A[1][1] = -mc; A[1][2] = a12y;
    A[2][0] = 0;   A[2][1] = 0;   A[2][2] = -1;
    B[0][0] = mc;  B[0][1] = -ms; B[0][2] = 0;
    B[1][0] = ms;  B[1][1] = mc;  B[1][2] = 0;
    B[2][0] = 0;   B[2][1] = 0;   B[2][2] = 1;
  } else {
    const T r = x1.z, i = x1.w;
    e[0] = (r * dx + i * dy) - z.x;
    e[1] = (-i * dx + r * dy) - z.y;
    e[2] = 0;
    A[0][0] = -r; A[0][1] = -i; A[0][2] = -i * dx + r * dy;
    A[1][0] = i;  A[1][1] = -r; A[1][2] = -r * dx - i * dy;
    A[2][0] = 0;  A[2][1] = 0;  A[2][2] = 0;
    B[0][0] = r;  B[0][1] = i;  B[0][2] = 0;
    B[1][0] = -i; B[1][1] = r;  B[1][2] = 0;
    B[2][0] = 0;  B[2][1] = 0;  B[2][2] = 0;
  }
}

template <typename T> __device__ __forceinline__ T group_sum8(T v) {   // sum over the LIN_GROUP lanes of a node
#pragma unroll
  for (int o = 1; o < LIN_GROUP; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int THREADS> __device__ __forceinline__ T block_sum(T v, T *scratch) {
  // wave reduction then one value per wave through LDS; result valid in thread 0
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int wave = wave_index(), lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < THREADS / 64; w++) r += scratch[w];
  return r;
}

// The last workgroup of an update launch also does the fixed-order reduction of the chi2 / |dx|^2 partials into
// the (chi2, |dx|) ring slot (what k_finalize_slot does in a launch of its own): one launch less per iteration.
struct FinArgs {
  int enabled;                 // 0: the caller launches k_finalize_slot itself
  const double *chi_partial;   // written by the linearisation of this iteration (an earlier launch)
  int n_chi;
  double *hist;                // ring of (chi2, |dx|) pairs
  int *counter;                // slot counter, advanced when `advance`
  int advance, ring;
  int *blocks_done;            // zero between launches
  OptCtrl *ctrl;               // rr_pgo_optimize without host round trips: the device-side loop state (null: none) ...
  OptSlot *ring_host;          // ... and the host-visible ring the iteration is published in
  int *err;                    // the engine's error block: [0] sticky error flag, [1] stop word
};
// Gauss-Newton inside rr_pgo_optimize (one thread): ct = chi2 of the state BEFORE this step (errors[i], :286), sqrt(nt) = |dx|
// (:273); the stop rule, and the end of the call after a failed factorisation
__device__ __forceinline__ void finalize_publish(const FinArgs &f, double ct, double nt) {
  if (!f.ctrl) return;
  const int e = f.err[0];
  if (f.err[1]) return;   // enqueued behind the iteration that met the stop rule: this item's linearisation has published chi2 of the final state
  const double nrm = sqrt(nt);
  const bool stop = e != 0 || nrm < f.ctrl->tolerance;   // :298-300; a failed factorisation ends the call (:271)
  if (stop) f.err[1] = 1;
  opt_publish(f.ctrl, f.ring_host, ct, nrm, (stop && !e ? OPT_STOP : 0) | (e << OPT_ERR_SHIFT));
}
template <int THREADS>
__device__ __forceinline__ void finalize_in_last_block(const FinArgs &f, const double *norm_partial, int n_norm, double *red) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __threadfence();                                  // this block's partial is out before it is counted
    is_last = atomicAdd(f.blocks_done, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();                                    // the other blocks' partials are visible
  double c = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < f.n_chi; i += THREADS) c += f.chi_partial[i];
  for (int i = threadIdx.x; i < n_norm; i += THREADS) n += norm_partial[i];
  const double ct = block_sum<double, THREADS>(c, red);
  const double nt = block_sum<double, THREADS>(n, red);
  if (threadIdx.x == 0) {
    const int slot = *f.counter % f.ring;
    if (f.n_chi > 0) f.hist[2 * slot] = ct;
    f.hist[2 * slot + 1] = sqrt(nt);
    if (f.advance) *f.counter = *f.counter + 1;
    *f.blocks_done = 0;
    finalize_publish(f, ct, nt);
  }
}

// rr_pgo_optimize's items that are not a Gauss-Newton iteration (one workgroup of 256; the sums in k_finalize_slot's order):
//   mode 0   chi2 of the current state (the linearisation before this launch wrote the partials): Gauss-Newton's entry behind
//            the last iteration when the stop rule never fired; Levenberg-Marquardt's initial error (:255) -> last_error
//   mode 1   the end of a Levenberg-Marquardt iteration (:274-300): error = chi2 of the state AFTER the step, |dx| from the
//            update's partials; reject (last_error < error) => `reject` for the undo launch behind this one, lambda x 2, else
//            lambda / 2; last_error = error either way (:284); the stop rule
struct OptItemArgs {
  const double *chi_partial, *norm_partial;
  int n_chi, n_norm, mode;
  OptCtrl *ctrl;
  OptSlot *ring;
  int *err;
};
__global__ void __launch_bounds__(256) k_opt_item(OptItemArgs a) {
  __shared__ double red[4];
  double c = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < a.n_chi; i += 256) c += a.chi_partial[i];
  for (int i = threadIdx.x; i < a.n_norm; i += 256) n += a.norm_partial[i];
  const double ct = block_sum<double, 256>(c, red);
  const double nt = block_sum<double, 256>(n, red);
  if (threadIdx.x != 0) return;
  OptCtrl *k = a.ctrl;
  k->reject = 0;
  if (a.err[1]) { opt_publish(k, a.ring, ct, 0.0, OPT_SKIPPED); return; }
  if (a.mode == 0) {
    k->last_error = ct;
    opt_publish(k, a.ring, ct, 0.0, 0);
    return;
  }
  const int e = a.err[0];
  if (e) {   // the step was not applied (k_update saw the flag): no decision to take, the call ends
    a.err[1] = 1;
    opt_publish(k, a.ring, ct, 0.0, e << OPT_ERR_SHIFT);
    return;
  }
  const double nrm = sqrt(nt);
  if (k->last_error < ct) { k->reject = 1; k->lambda *= 2.0; }   // :276-279
  else k->lambda /= 2.0;                                           // :281
  k->last_error = ct;                                              // :284
  const bool stop = nrm < k->tolerance;
  if (stop) a.err[1] = 1;
  opt_publish(k, a.ring, ct, nrm, stop ? OPT_STOP : 0);
}

// (the sum in k_finalize_slot's order: 256 threads, stride 256, block_sum)
template <typename A> __device__ __forceinline__ void opt_publish_chi2_in_last_block(const A &a, double *red) {
  if (!a.publish) return;
  const bool stopped = !a.reset_ctrl && a.err[1] != 0;   // (the call's first launch resets the word: never stopped)
  if (a.publish == 1 && !stopped) return;
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(a.blocks_done, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  double c = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += LIN_THREADS) c += a.chi2_partial[i];
  const double ct = block_sum<double, LIN_THREADS>(c, red);
  if (threadIdx.x == 0) {
    *a.blocks_done = 0;
    if (!stopped) a.ctrl->last_error = ct;
    opt_publish(a.ctrl, a.ring_host, ct, 0.0, stopped ? OPT_SKIPPED : 0);
  }
}

// One group of LIN_GROUP lanes per node pulls the node's incident edges
// (deterministic: no atomics, fixed summation order), builds the node's
// diagonal block and right-hand side; the lane holding an edge in its `from`
// role also writes the off-diagonal block and the edge's chi2 term.
template <typename TO, typename T>
__global__ void __launch_bounds__(LIN_THREADS) k_linearize(LinArgs<TO, T> a) {
  using V4 = typename VecT<T>::V4;
  using V2 = typename VecT<T>::V2;
  __shared__ double red[LIN_THREADS / 64];
  const int gid = blockIdx.x * LIN_THREADS + threadIdx.x;
  for (int i = gid; i < a.n_zero_words; i += gridDim.x * LIN_THREADS) a.zero_words[i] = 0u;
  for (int i = gid; i < a.n_fill_words; i += gridDim.x * LIN_THREADS) a.fill_words[i] = X_PENDING_WORD;
  opt_reset_in_first_thread(a);
  const T lambda = a.lambda_from_ctrl ? (T)a.ctrl->lambda : a.lambda;
  const int slot = gid / LIN_GROUP, sub = gid % LIN_GROUP;
  const int node = slot < a.n_nodes ? (a.node_list ? a.node_list[slot] : slot) : -1;
  double chi = 0.0;
  T hd[6] = {0, 0, 0, 0, 0, 0};  // 00 10 11 20 21 22
  T bv[3] = {0, 0, 0};
  int nd = 0;
  if (node >= 0) {
    nd = a.node_dim[node];
    const V4 self = a.pose[node];
    const int q1 = a.inc_ptr[node + 1];
    for (int q = a.inc_ptr[node] + sub; q < q1; q += LIN_GROUP) {
      const int2 inc = a.inc_list[q];
      const int ent = inc.x;
      // sharded runs: an edge contributes to diagonal blocks, right-hand side and chi2 on ONE rank (bit 2, its
      // owner -- the other ranks' copies of a far endpoint are stale); the off-diagonal block of an edge between
      // two shared nodes is written by every rank (bit 3)
      if (!(ent & 12)) continue;
      const bool owns = ent & 4;
      const int k = ent >> 4, kind = (ent >> 1) & 1, role = ent & 1;
      const EdgeRec<T> rec = a.e_rec[k];
      const V4 other = a.pose[inc.y];
      const V4 z = rec.meas;
      const V4 wa = rec.info_a;
      const V2 wb = rec.info_b;
      const T W[3][3] = {{wa.x, wa.y, wa.z}, {wa.y, wa.w, wb.x}, {wa.z, wb.x, wb.y}};
      T e[3], A[3][3], B[3][3];
      edge_linearize_2d<T>(kind, role ? other : self, role ? self : other, z, e, A, B);
      // J = A (from role) or B (to role);  JW = J^T W
      T JW[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          T s = 0;
#pragma unroll
          for (int r = 0; r < 3; r++) s += (role ? B[r][i] : A[r][i]) * W[r][j];
          JW[i][j] = s;
        }
      if (a.write_system && owns) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) {
            T s = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) s += JW[i][r] * (role ? B[r][j] : A[r][j]);
            hd[t++] += s;
          }
#pragma unroll
        for (int i = 0; i < 3; i++) bv[i] += JW[i][0] * e[0] + JW[i][1] * e[1] + JW[i][2] * e[2];
      }
      if (role == 0) {
        // chi2 term e^T W e (:555,568), accumulated in f64
        T we0 = W[0][0] * e[0] + W[0][1] * e[1] + W[0][2] * e[2];
        T we1 = W[1][0] * e[0] + W[1][1] * e[1] + W[1][2] * e[2];
        T we2 = W[2][0] * e[0] + W[2][1] * e[1] + W[2][2] * e[2];
        if (owns) chi += (double)(e[0] * we0 + e[1] * we1 + e[2] * we2);   // every edge's term is owned by one rank
        if (a.write_system && (ent & 8)) {
          // off-diagonal block H[from rows, to cols] = A^T W B
          const int64_t so = rec.slot;
          TO *dst = a.hvals + (so >> 1);
          const bool tr = so & 1;
          const int d2 = kind ? 2 : 3;
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              if (j >= d2) continue;
              T s = JW[i][0] * B[0][j] + JW[i][1] * B[1][j] + JW[i][2] * B[2][j];
              dst[tr ? j * 3 + i : i * d2 + j] = (TO)s;
            }
        }
      }
    }
  }
  if (a.write_system) {
#pragma unroll
    for (int t = 0; t < 6; t++) hd[t] = group_sum8(hd[t]);
#pragma unroll
    for (int t = 0; t < 3; t++) bv[t] = group_sum8(bv[t]);
    if (node >= 0 && sub == 0) {
      T add = lambda;
      if (node == a.anchor) add += (T)10000000.0;
      if (a.adds_diag && !a.adds_diag[node]) add = 0;
      TO *d = a.hvals + a.diag_off[node];
      if (nd == 3) {
        d[0] = (TO)(hd[0] + add); d[1] = (TO)hd[1];         d[2] = (TO)hd[3];
        d[3] = (TO)hd[1];         d[4] = (TO)(hd[2] + add); d[5] = (TO)hd[4];
        d[6] = (TO)hd[3];         d[7] = (TO)hd[4];         d[8] = (TO)(hd[5] + add);
      } else {
        d[0] = (TO)(hd[0] + add); d[1] = (TO)hd[1];
        d[2] = (TO)hd[1];         d[3] = (TO)(hd[2] + add);
      }
      TO *bo = a.b + a.node_offset[node];
      for (int t = 0; t < nd; t++) bo[t] = (TO)(-bv[t]);
    }
  }
  double tot = block_sum<double, LIN_THREADS>(chi, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = tot;
  opt_publish_chi2_in_last_block(a, red);
}

// ---- the EDGE-PARALLEL form of the same linearisation (the north star's wording; RR_PGO_EDGE_LINEARIZE=1) ----
// Kept as the measured alternative to k_linearize, not as the default: see DESIGN.md "Linearisation: pull vs
// edge-parallel".  One thread per edge evaluates the factor ONCE (the pull form evaluates it from both ends):
//   off-diagonal block   stored directly (one writer per block)
//   from-node terms      edges of one `from` node are consecutive in file order on the lattice: a segmented scan over
//                        the wave (head flags, fixed order) adds them up and the last lane of a run issues ONE set of
//                        atomic adds per run
//   to-node terms        atomic adds, one set per edge (the far endpoints of a wave's edges are all different)
//   chi2                 workgroup sum in f64, like the pull form
// k_lin_init clears diagonal blocks and rhs first (and places prior / lambda), k_lin_finish mirrors the lower
// triangles.  Floating-point atomics make the summation order -- hence the last bits of H, b and every later
// iterate -- vary from run to run; the pull form is bit-reproducible.
template <typename TO, typename T>
__global__ void __launch_bounds__(256) k_lin_init(LinArgs<TO, T> a) {
  const int node = blockIdx.x * 256 + threadIdx.x;
  if (node >= a.n_nodes) return;
  const int nd = a.node_dim[node];
  T add = a.lambda;
  if (node == a.anchor) add += (T)10000000.0;
  TO *d = a.hvals + a.diag_off[node];
  for (int t = 0; t < nd * nd; t++) d[t] = (t / nd == t % nd) ? (TO)add : (TO)0;
  TO *bo = a.b + a.node_offset[node];
  for (int t = 0; t < nd; t++) bo[t] = 0;
}
template <typename TO, typename T>
__global__ void __launch_bounds__(256) k_lin_finish(LinArgs<TO, T> a) {
  const int node = blockIdx.x * 256 + threadIdx.x;
  if (node >= a.n_nodes) return;
  const int nd = a.node_dim[node];
  TO *d = a.hvals + a.diag_off[node];
  for (int i = 0; i < nd; i++)
    for (int j = i + 1; j < nd; j++) d[i * nd + j] = d[j * nd + i];
}
template <typename TO, typename T>
__global__ void __launch_bounds__(LIN_THREADS) k_linearize_edges(LinArgs<TO, T> a, int n_edges) {
  using V4 = typename VecT<T>::V4;
  using V2 = typename VecT<T>::V2;
  __shared__ double red[LIN_THREADS / 64];
  const int k = blockIdx.x * LIN_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = k < n_edges;
  const int kc = live ? k : n_edges - 1;
  const int2 ft = a.e_idx[kc];
  const int kind = a.node_dim[ft.y] == 2 ? 1 : 0;   // an XY landmark is always the `to` end (g2o.rs:98-115)
  const V4 x1 = a.pose[ft.x], x2 = a.pose[ft.y];
  const V4 z = a.e_meas[kc];
  const V4 wa = a.e_info_a[kc];
  const V2 wb = a.e_info_b[kc];
  const T W[3][3] = {{wa.x, wa.y, wa.z}, {wa.y, wa.w, wb.x}, {wa.z, wb.x, wb.y}};
  T e[3], A[3][3], B[3][3];
  edge_linearize_2d<T>(kind, x1, x2, z, e, A, B);
  const T we0 = W[0][0] * e[0] + W[0][1] * e[1] + W[0][2] * e[2];
  const T we1 = W[1][0] * e[0] + W[1][1] * e[1] + W[1][2] * e[2];
  const T we2 = W[2][0] * e[0] + W[2][1] * e[1] + W[2][2] * e[2];
  double chi = live ? (double)(e[0] * we0 + e[1] * we1 + e[2] * we2) : 0.0;
  if (a.write_system) {
    T AW[3][3], BW[3][3];   // J^T W
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        T sa = 0, sb = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) { sa += A[r][i] * W[r][j]; sb += B[r][i] * W[r][j]; }
        AW[i][j] = sa;
        BW[i][j] = sb;
      }
    // nine terms per end: lower triangle of J^T W J (00 10 11 20 21 22) and -J^T W e
    T fa[9], fb[9];
    int t = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        T sa = 0, sb = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) { sa += AW[i][r] * A[r][j]; sb += BW[i][r] * B[r][j]; }
        fa[t] = live ? sa : (T)0;
        fb[t] = live ? sb : (T)0;
        t++;
      }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      fa[6 + i] = live ? -(AW[i][0] * e[0] + AW[i][1] * e[1] + AW[i][2] * e[2]) : (T)0;
      fb[6 + i] = live ? -(BW[i][0] * e[0] + BW[i][1] * e[1] + BW[i][2] * e[2]) : (T)0;
    }
    if (live) {
      // off-diagonal block H[from rows, to cols] = A^T W B
      const int64_t so = a.e_slot[k];
      TO *dst = a.hvals + (so >> 1);
      const bool tr = so & 1;
      const int d2 = kind ? 2 : 3;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          if (j >= d2) continue;
          const T sv = AW[i][0] * B[0][j] + AW[i][1] * B[1][j] + AW[i][2] * B[2][j];
          dst[tr ? j * 3 + i : i * d2 + j] = (TO)sv;
        }
    }
    // from-node terms: segmented inclusive scan over the wave (head flag = first lane of a run of equal `from`)
    const int key = live ? ft.x : -1;
    const int kprev = __shfl_up(key, 1);
    bool head = lane == 0 || kprev != key;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      T other[9];
#pragma unroll
      for (int q = 0; q < 9; q++) other[q] = __shfl_up(fa[q], o);
      const bool ohead = __shfl_up((int)head, o) != 0;
      if (lane >= o && !head) {
#pragma unroll
        for (int q = 0; q < 9; q++) fa[q] += other[q];
        head = ohead;
      }
    }
    const int knext = __shfl_down(key, 1);
    const bool tail = live && (lane == 63 || knext != key);
    auto add_node = [&](int node, const T (&f)[9]) {
      const int nd = a.node_dim[node];
      TO *d = a.hvals + a.diag_off[node];
      TO *bo = a.b + a.node_offset[node];
      if (nd == 3) {
        unsafeAtomicAdd(d + 0, (TO)f[0]); unsafeAtomicAdd(d + 3, (TO)f[1]); unsafeAtomicAdd(d + 4, (TO)f[2]);
        unsafeAtomicAdd(d + 6, (TO)f[3]); unsafeAtomicAdd(d + 7, (TO)f[4]); unsafeAtomicAdd(d + 8, (TO)f[5]);
        unsafeAtomicAdd(bo + 0, (TO)f[6]); unsafeAtomicAdd(bo + 1, (TO)f[7]); unsafeAtomicAdd(bo + 2, (TO)f[8]);
      } else {
        unsafeAtomicAdd(d + 0, (TO)f[0]); unsafeAtomicAdd(d + 2, (TO)f[1]); unsafeAtomicAdd(d + 3, (TO)f[2]);
        unsafeAtomicAdd(bo + 0, (TO)f[6]); unsafeAtomicAdd(bo + 1, (TO)f[7]);
      }
    };
    if (tail) add_node(ft.x, fa);
    if (live) add_node(ft.y, fb);
  }
  const double tot = block_sum<double, LIN_THREADS>(chi, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = tot;
}

// ---- the edge-parallel form exactly as the north star words it (RR_PGO_EDGE_LINEARIZE=2): ONE WAVEFRONT PER EDGE.
// The wave reads the edge's record (one 64 / 128-byte line) and its two endpoint poses with wave-uniform loads, evaluates error
// and Jacobians once, STAGES A, B, the information matrix and the error in LDS, and then lane l forms output scalar l out of the
// staged operands (dynamic indices are LDS addresses, not register selects):
//   lanes 0-5 / 6-8     lower triangle of A^T W A / -A^T W e      -> the from-node's accumulator
//   lanes 9-14 / 15-17  lower triangle of B^T W B / -B^T W e      -> the to-node's accumulator
//   lanes 18-26         A^T W B, the off-diagonal block            -> stored (one writer per block)
//   lane 27             e^T W e
// The scatter-add is LDS-REDUCED: a workgroup takes 64 consecutive edges (file order: runs of one from-node), keeps one
// accumulator of 9 scalars per node it meets in an LDS table (open addressing, ds atomics), and flushes every occupied entry
// with ONE set of global atomic adds at the end.  k_lin_init / k_lin_finish as for k_linearize_edges; floating-point atomics
// make the last bits vary from run to run (the pull form is bit-reproducible and is the product).
constexpr int WE_EDGES = 64, WE_SLOTS = 128;
template <typename T> __device__ __forceinline__ void lds_atomic_add(T *p, T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename TO, typename T>
__global__ void __launch_bounds__(256) k_linearize_wave_edges(LinArgs<TO, T> a, int n_edges) {
  using V4 = typename VecT<T>::V4;
  using V2 = typename VecT<T>::V2;
  __shared__ int slot_node[WE_SLOTS];
  __shared__ T slot_acc[WE_SLOTS][9];     // 00 10 11 20 21 22 of J^T W J, then -J^T W e
  __shared__ T stage[4][40];              // per wave: A (9) | B (9) | W (9) | e (3) | W e (3)
  __shared__ double red[4];
  const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63;
  for (int s = tid; s < WE_SLOTS; s += 256) {
    slot_node[s] = -1;
#pragma unroll
    for (int q = 0; q < 9; q++) slot_acc[s][q] = 0;
  }
  __syncthreads();
  T *st = stage[wave];
  double chi = 0.0;
  const int e0 = blockIdx.x * WE_EDGES;
  for (int t = 0; t < WE_EDGES / 4; t++) {
    const int k = e0 + 4 * t + wave;      // wave-uniform
    if (k >= n_edges) break;
    const EdgeRec<T> rec = a.e_rec[k];
    const int kind = a.node_dim[rec.to] == 2 ? 1 : 0;
    const V4 x1 = a.pose[rec.from], x2 = a.pose[rec.to];
    const V4 wa = rec.info_a;
    const V2 wb = rec.info_b;
    const T W[3][3] = {{wa.x, wa.y, wa.z}, {wa.y, wa.w, wb.x}, {wa.z, wb.x, wb.y}};
    T e[3], A[3][3], B[3][3];
    edge_linearize_2d<T>(kind, x1, x2, rec.meas, e, A, B);
    // stage the operands (every lane holds the same values: lane q of the first 39 writes entry q)
    T mine = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) {
      mine = lane == q ? A[q / 3][q % 3] : mine;
      mine = lane == 9 + q ? B[q / 3][q % 3] : mine;
      mine = lane == 18 + q ? W[q / 3][q % 3] : mine;
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      mine = lane == 27 + q ? e[q] : mine;
      mine = lane == 30 + q ? W[q][0] * e[0] + W[q][1] * e[1] + W[q][2] * e[2] : mine;
    }
    if (lane < 33) st[lane] = mine;
    // the two accumulators of this edge: lane 0 finds (or claims) the from-node's entry, lane 1 the to-node's
    int slot = 0;
    if (lane < 2 && a.write_system) {
      const int node = lane == 0 ? rec.from : rec.to;
      int h = (int)((unsigned)node * 2654435761u >> 25) & (WE_SLOTS - 1);
      for (;;) {
        const int prev = atomicCAS(&slot_node[h], -1, node);
        if (prev == -1 || prev == node) break;
        h = (h + 1) & (WE_SLOTS - 1);   // (at most 2 x WE_EDGES distinct nodes per workgroup: the table never fills)
      }
      slot = h;
    }
    const int slot_f = __shfl(slot, 0), slot_t = __shfl(slot, 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T *As = st, *Bs = st + 9, *Ws = st + 18, *es = st + 27, *wes = st + 30;
    // lane -> (i, j) of a lower triangle: 0:(0,0) 1:(1,0) 2:(1,1) 3:(2,0) 4:(2,1) 5:(2,2)
    auto tri_i = [](int q) { return q < 1 ? 0 : q < 3 ? 1 : 2; };
    auto tri_j = [](int q) { return q < 1 ? 0 : q < 3 ? q - 1 : q - 3; };
    auto jwj = [&](const T *J, const T *K, int i, int j) {   // (J^T W K)(i, j)
      T sum = 0;
      for (int r = 0; r < 3; r++) {
        T jw = 0;
        for (int c = 0; c < 3; c++) jw += J[c * 3 + i] * Ws[c * 3 + r];
        sum += jw * K[r * 3 + j];
      }
      return sum;
    };
    auto jwe = [&](const T *J, int i) { return -(J[0 * 3 + i] * wes[0] + J[1 * 3 + i] * wes[1] + J[2 * 3 + i] * wes[2]); };
    if (a.write_system) {
      if (lane < 6) lds_atomic_add(&slot_acc[slot_f][lane], jwj(As, As, tri_i(lane), tri_j(lane)));
      else if (lane < 9) lds_atomic_add(&slot_acc[slot_f][lane], jwe(As, lane - 6));
      else if (lane < 15) lds_atomic_add(&slot_acc[slot_t][lane - 9], jwj(Bs, Bs, tri_i(lane - 9), tri_j(lane - 9)));
      else if (lane < 18) lds_atomic_add(&slot_acc[slot_t][lane - 9], jwe(Bs, lane - 15));
      else if (lane < 27) {
        const int q = lane - 18, i = q / 3, j = q % 3, d2 = kind ? 2 : 3;
        if (j < d2) {
          const int64_t so = rec.slot;
          TO *dst = a.hvals + (so >> 1);
          dst[(so & 1) ? j * 3 + i : i * d2 + j] = (TO)jwj(As, Bs, i, j);
        }
      }
    }
    if (lane == 27) chi += (double)(es[0] * wes[0] + es[1] * wes[1] + es[2] * wes[2]);
    __builtin_amdgcn_wave_barrier();   // the stage is rewritten by the next edge
  }
  __syncthreads();
  if (a.write_system)
    for (int s = tid; s < WE_SLOTS; s += 256) {
      const int node = slot_node[s];
      if (node < 0) continue;
      const T *f = slot_acc[s];
      const int nd = a.node_dim[node];
      TO *d = a.hvals + a.diag_off[node];
      TO *bo = a.b + a.node_offset[node];
      if (nd == 3) {
        unsafeAtomicAdd(d + 0, (TO)f[0]); unsafeAtomicAdd(d + 3, (TO)f[1]); unsafeAtomicAdd(d + 4, (TO)f[2]);
        unsafeAtomicAdd(d + 6, (TO)f[3]); unsafeAtomicAdd(d + 7, (TO)f[4]); unsafeAtomicAdd(d + 8, (TO)f[5]);
        unsafeAtomicAdd(bo + 0, (TO)f[6]); unsafeAtomicAdd(bo + 1, (TO)f[7]); unsafeAtomicAdd(bo + 2, (TO)f[8]);
      } else {
        unsafeAtomicAdd(d + 0, (TO)f[0]); unsafeAtomicAdd(d + 2, (TO)f[1]); unsafeAtomicAdd(d + 3, (TO)f[2]);
        unsafeAtomicAdd(bo + 0, (TO)f[6]); unsafeAtomicAdd(bo + 1, (TO)f[7]);
      }
    }
  const double tot = block_sum<double, 256>(chi, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = tot;
}

// ------------------------------------------------------------------ SE(3)
// NOT reference behaviour: the reference's SE(3) path is todo!() (pose_graph_optimization.rs:241,
// 357,570).  Build-defined, g2o file convention, identical to the oracle's definition:
//   E = Z^-1 * Xi^-1 * Xj ,  e = [ t_E ; sign(w_E) * vec(q_E) ]                (6 scalars)
//   update  X <- X * (dt, Exp(dw)) :  t += R dt ,  q <- q (x) exp(dw)            (right increments)
// Jacobians (cf. the structure hinted at :488-514: A = [-Ra, Ra*skew(t_b); 0, ..], B = [Re, 0; 0, ..]):
//   B = [ R_E , 0 ; 0 , (s/2)(w_E I + [v_E]x) ]
//   A = [ -Rz^T , Rz^T [t_C]x ; 0 , -(s/2) vec3x3( L(q_z^-1) R(q_C) ) ] ,  C = Xi^-1 Xj
template <typename T, typename TC = T> struct LinArgs3 {
  int n_nodes;
  const typename VecT<TC>::V4 *pose;    // 2 per node: (tx,ty,tz,-), (qx,qy,qz,qw)
  const int2 *e_idx;
  const typename VecT<TC>::V4 *e_meas;  // 2 per edge, same packing
  const TC *e_info;                     // 21 per edge, row-major upper triangle
  const int64_t *e_slot;
  const int32_t *inc_ptr;
  const int2 *inc_list;                 // as LinArgs::inc_list: (edge << 4 | offdiag << 3 | owns << 2 | role, the other endpoint's node)
  const int32_t *node_list;             // sharded runs: the nodes this rank linearises, else null
  const int32_t *node_offset;
  const int64_t *diag_off;
  T *hvals;
  T *b;
  double *chi2_partial;
  int anchor;
  TC lambda;
  int write_system;
  const uint8_t *adds_diag;             // sharded runs: per node, 1 = this rank adds prior / lambda
  unsigned *zero_words;                 // as LinArgs::zero_words
  int n_zero_words;
  unsigned *fill_words;                 // as LinArgs::fill_words
  int n_fill_words;
  OptCtrl *ctrl;                        // as LinArgs::ctrl ...
  int lambda_from_ctrl, reset_ctrl;
  double reset_lambda, reset_tolerance;
  int *err;
  int publish;
  OptSlot *ring_host;
  int *blocks_done;
};

template <typename T> __device__ __forceinline__ void q_mul(const T a[4], const T b[4], T r[4]) {
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
template <typename T> __device__ __forceinline__ void q_rot(const T q[4], const T v[3], T r[3]) {
  const T cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  const T dx = q[1] * cz - q[2] * cy, dy = q[2] * cx - q[0] * cz, dz = q[0] * cy - q[1] * cx;
  r[0] = v[0] + 2 * (q[3] * cx + dx);
  r[1] = v[1] + 2 * (q[3] * cy + dy);
  r[2] = v[2] + 2 * (q[3] * cz + dz);
}
template <typename T> __device__ __forceinline__ void q_to_rot(const T q[4], T R[3][3]) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - z * w);     R[0][2] = 2 * (x * z + y * w);
  R[1][0] = 2 * (x * y + z * w);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - x * w);
  R[2][0] = 2 * (x * z - y * w);     R[2][1] = 2 * (y * z + x * w);     R[2][2] = 1 - 2 * (x * x + y * y);
}

// error (6) and the requested Jacobian (role 0: A w.r.t. Xi, role 1: B w.r.t. Xj), 6 x 6 row-major
template <typename T>
__device__ void edge_linearize_3d(int role, const T ti[3], const T qi[4], const T tj[3], const T qj[4],
                                  const T tz[3], const T qz[4], T e[6], T J[6][6]) {
  const T qic[4] = {-qi[0], -qi[1], -qi[2], qi[3]};
  const T qzc[4] = {-qz[0], -qz[1], -qz[2], qz[3]};
  // C = Xi^-1 Xj
  T d[3] = {tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2]}, tC[3], qC[4];
  q_rot(qic, d, tC);
  q_mul(qic, qj, qC);
  // E = Z^-1 C
  T d2[3] = {tC[0] - tz[0], tC[1] - tz[1], tC[2] - tz[2]}, tE[3], qE[4];
  q_rot(qzc, d2, tE);
  q_mul(qzc, qC, qE);
  const T sgn = qE[3] < 0 ? (T)-1 : (T)1;
  e[0] = tE[0]; e[1] = tE[1]; e[2] = tE[2];
  e[3] = sgn * qE[0]; e[4] = sgn * qE[1]; e[5] = sgn * qE[2];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 6; c++) J[r][c] = 0;
  if (role) {
    T RE[3][3];
    q_to_rot(qE, RE);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) J[r][c] = RE[r][c];
    const T h = (T)0.5 * sgn, w = qE[3], vx = qE[0], vy = qE[1], vz = qE[2];
    J[3][3] = h * w;   J[3][4] = -h * vz; J[3][5] = h * vy;
    J[4][3] = h * vz;  J[4][4] = h * w;   J[4][5] = -h * vx;
    J[5][3] = -h * vy; J[5][4] = h * vx;  J[5][5] = h * w;
  } else {
    T RZt[3][3];
    q_to_rot(qzc, RZt);   // rotation of Z^-1 = Rz^T
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) J[r][c] = -RZt[r][c];
      // Rz^T [tC]x : column k = Rz^T (tC x e_k)... [tC]x u = tC x u
      J[r][3] = RZt[r][1] * tC[2] - RZt[r][2] * tC[1];
      J[r][4] = RZt[r][2] * tC[0] - RZt[r][0] * tC[2];
      J[r][5] = RZt[r][0] * tC[1] - RZt[r][1] * tC[0];
    }
    // d vec(q_E)/d dw_i = -(1/2) vec( qz^-1 (x) (u,0) (x) qC ), column by column
#pragma unroll
    for (int k = 0; k < 3; k++) {
      T u[4] = {k == 0 ? (T)1 : (T)0, k == 1 ? (T)1 : (T)0, k == 2 ? (T)1 : (T)0, (T)0}, t1[4], t2[4];
      q_mul(qzc, u, t1);
      q_mul(t1, qC, t2);
      J[3][3 + k] = (T)-0.5 * sgn * t2[0];
      J[4][3 + k] = (T)-0.5 * sgn * t2[1];
      J[5][3 + k] = (T)-0.5 * sgn * t2[2];
    }
  }
}

template <typename TO, typename T>
__global__ void __launch_bounds__(LIN_THREADS) k_linearize_se3(LinArgs3<TO, T> a) {
  using V4 = typename VecT<T>::V4;
  __shared__ double red[LIN_THREADS / 64];
  const int gid = blockIdx.x * LIN_THREADS + threadIdx.x;
  for (int i = gid; i < a.n_zero_words; i += gridDim.x * LIN_THREADS) a.zero_words[i] = 0u;
  for (int i = gid; i < a.n_fill_words; i += gridDim.x * LIN_THREADS) a.fill_words[i] = X_PENDING_WORD;
  opt_reset_in_first_thread(a);
  const T lambda = a.lambda_from_ctrl ? (T)a.ctrl->lambda : a.lambda;
  const int slot = gid / LIN_GROUP, sub = gid % LIN_GROUP;
  const int node = slot < a.n_nodes ? (a.node_list ? a.node_list[slot] : slot) : -1;
  double chi = 0.0;
  T hd[21], bv[6];   // lower triangle of the 6 x 6 diagonal block, row-major: (0,0),(1,0),(1,1),...
#pragma unroll
  for (int t = 0; t < 21; t++) hd[t] = 0;
#pragma unroll
  for (int t = 0; t < 6; t++) bv[t] = 0;
  if (node >= 0) {
    const V4 st = a.pose[2 * node], sq = a.pose[2 * node + 1];
    const T ts[3] = {st.x, st.y, st.z}, qs[4] = {sq.x, sq.y, sq.z, sq.w};
    const int q1 = a.inc_ptr[node + 1];
    for (int q = a.inc_ptr[node] + sub; q < q1; q += LIN_GROUP) {
      const int2 inc = a.inc_list[q];
      const int ent = inc.x;
      if (!(ent & 12)) continue;   // sharded runs: see k_linearize
      const bool owns = ent & 4;
      const int k = ent >> 4, role = ent & 1;
      const int other = inc.y;
      const V4 ot = a.pose[2 * other], oq = a.pose[2 * other + 1];
      const T to[3] = {ot.x, ot.y, ot.z}, qo[4] = {oq.x, oq.y, oq.z, oq.w};
      const V4 zt = a.e_meas[2 * k], zq = a.e_meas[2 * k + 1];
      const T tz[3] = {zt.x, zt.y, zt.z}, qz[4] = {zq.x, zq.y, zq.z, zq.w};
      T W[6][6];
      {
        const T *w = a.e_info + (int64_t)k * 21;
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = i; j < 6; j++) { W[i][j] = w[t]; W[j][i] = w[t]; t++; }
      }
      T e[6], J[6][6];
      if (role) edge_linearize_3d<T>(1, to, qo, ts, qs, tz, qz, e, J);
      else edge_linearize_3d<T>(0, ts, qs, to, qo, tz, qz, e, J);
      T JW[6][6];   // J^T W
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
          T sacc = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) sacc += J[r][i] * W[r][j];
          JW[i][j] = sacc;
        }
      if (a.write_system && owns) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) {
            T sacc = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) sacc += JW[i][r] * J[r][j];
            hd[t++] += sacc;
          }
#pragma unroll
        for (int i = 0; i < 6; i++) {
          T sacc = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) sacc += JW[i][r] * e[r];
          bv[i] += sacc;
        }
      }
      if (role == 0) {
        T c2 = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          T we = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) we += W[i][r] * e[r];
          c2 += e[i] * we;
        }
        if (owns) chi += (double)c2;
        if (a.write_system && (ent & 8)) {
          // off-diagonal block H[from rows, to cols] = A^T W B: B of the same edge
          T e2[6], Bm[6][6];
          edge_linearize_3d<T>(1, ts, qs, to, qo, tz, qz, e2, Bm);
          const int64_t so = a.e_slot[k];
          TO *dst = a.hvals + (so >> 1);
          const bool tr = so & 1;
#pragma unroll
          for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) {
              T sacc = 0;
#pragma unroll
              for (int r = 0; r < 6; r++) sacc += JW[i][r] * Bm[r][j];
              dst[tr ? j * 6 + i : i * 6 + j] = (TO)sacc;
            }
        }
      }
    }
  }
  if (a.write_system) {
#pragma unroll
    for (int t = 0; t < 21; t++) hd[t] = group_sum8(hd[t]);
#pragma unroll
    for (int t = 0; t < 6; t++) bv[t] = group_sum8(bv[t]);
    if (node >= 0 && sub == 0) {
      T add = lambda;
      if (node == a.anchor) add += (T)10000000.0;
      if (a.adds_diag && !a.adds_diag[node]) add = 0;
      TO *d = a.hvals + a.diag_off[node];
      int t = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
          const T v = hd[t++] + (i == j ? add : (T)0);
          d[i * 6 + j] = (TO)v;
          d[j * 6 + i] = (TO)v;
        }
      TO *bo = a.b + a.node_offset[node];
#pragma unroll
      for (int i = 0; i < 6; i++) bo[i] = (TO)(-bv[i]);
    }
  }
  double tot = block_sum<double, LIN_THREADS>(chi, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = tot;
  opt_publish_chi2_in_last_block(a, red);
}

template <typename T, typename TC = T> struct UpdArgs3 {
  int n_nodes;
  typename VecT<TC>::V4 *pose;
  const int32_t *node_pcol, *node_offset;
  const T *x, *dx_ref_in;
  T *dx_ref_out;
  TC sign;
  double *norm_partial;
  int export_only;     // 1: only write dx_ref_out
  const int *err;      // sticky device error flag: a failed factorisation must not touch the state
  const int32_t *node_list;    // sharded runs: the nodes this rank updates (own + shared), n_nodes = its length
  const uint8_t *norm_counts;  // sharded runs: per node, 1 = this rank adds the node's |dx|^2 (every node counted once)
  const int *gate;             // as UpdArgs::gate
  FinArgs fin;
};

// X <- X * (dt, Exp(dw)):  t += R dt ;  q <- normalise( q (x) exp(dw) )
template <typename TO, typename T>
__global__ void __launch_bounds__(UPD_THREADS) k_update_se3(UpdArgs3<TO, T> a) {
  __shared__ double red[UPD_THREADS / 64];
  const int slot = blockIdx.x * UPD_THREADS + threadIdx.x;
  const int node = slot < a.n_nodes ? (a.node_list ? a.node_list[slot] : slot) : -1;
  double nrm = 0.0;
  if (a.gate && *a.gate == 0) return;
  const bool failed = a.err && (a.err[0] != 0 || a.err[1] != 0);   // as k_update
  if (node >= 0 && !failed) {
    T d[6];
    const TO *src = a.dx_ref_in ? a.dx_ref_in + a.node_offset[node] : a.x + a.node_pcol[node];
#pragma unroll
    for (int t = 0; t < 6; t++) d[t] = (T)src[t];
    if (a.dx_ref_out) {
      TO *dst = a.dx_ref_out + a.node_offset[node];
#pragma unroll
      for (int t = 0; t < 6; t++) dst[t] = src[t];
    }
    if (a.export_only) return;
#pragma unroll
    for (int t = 0; t < 6; t++) { nrm += (double)d[t] * (double)d[t]; d[t] *= a.sign; }
    if (a.norm_counts && !a.norm_counts[node]) nrm = 0.0;
    auto pt = a.pose[2 * node], pq = a.pose[2 * node + 1];
    const T q[4] = {pq.x, pq.y, pq.z, pq.w};
    T rt[3];
    q_rot(q, d, rt);
    pt.x += rt[0]; pt.y += rt[1]; pt.z += rt[2];
    const T th = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    T dq[4];
    if (th < (T)1e-12) {
      dq[0] = (T)0.5 * d[3]; dq[1] = (T)0.5 * d[4]; dq[2] = (T)0.5 * d[5]; dq[3] = 1;
    } else {
      const T sc = sin((T)0.5 * th) / th;
      dq[0] = sc * d[3]; dq[1] = sc * d[4]; dq[2] = sc * d[5]; dq[3] = cos((T)0.5 * th);
    }
    T qn[4];
    q_mul(q, dq, qn);
    const T inv = (T)1 / sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    pq.x = qn[0] * inv; pq.y = qn[1] * inv; pq.z = qn[2] * inv; pq.w = qn[3] * inv;
    a.pose[2 * node] = pt;
    a.pose[2 * node + 1] = pq;
  }
  if (a.export_only) return;
  double tot = block_sum<double, UPD_THREADS>(nrm, red);
  if (threadIdx.x == 0) a.norm_partial[blockIdx.x] = tot;
  if (a.fin.enabled) finalize_in_last_block<UPD_THREADS>(a.fin, a.norm_partial, (int)gridDim.x, red);
}

// ------------------------------------------------------------ multifrontal

// Everything a workgroup needs to know about one front, one 64-byte record
// (fetched with a single scalar load instead of a chain of table lookups).
struct SnMeta {
  int32_t nc, nr, col0, uld;        // pivot columns, rows below, first permuted column, ld of U
                                    // (0 = packed in uvals, > 0 = in place in lvals, < 0 = packed in xch)
  int32_t asm_begin, asm_count;     // flat assembly entries (fasm_src / fasm_dst)
  int32_t dup_begin, dup_count;     // entries of parallel-edge blocks, added serially
  int32_t child_begin, child_count; // into child_meta
  int32_t rows_ptr, wblk;           // into sn_rows; first 16 x 16 block of this front in winv
  int64_t loff, uoff;               // panel / update-matrix offsets
};
static_assert(sizeof(SnMeta) == 64, "SnMeta must stay one 64-byte record");

struct ChildMeta {
  int64_t uoff;      // the child's update matrix: uvals (uld == 0), lvals in place (uld > 0), xch (uld < 0)
  int64_t scat_ptr;  // scatter map into the parent's LDS image, -1 = use rel
  int64_t rel_ptr;
  int32_t ncu, uld;  // nrows + 1 (rhs row), leading dimension (0 = packed)
};
static_assert(sizeof(ChildMeta) == 32, "ChildMeta must stay one 32-byte record");

template <typename T> struct FactorArgs {
  const int32_t *task_ptr, *task_sn;
  int task_begin;
  const SnMeta *sn_meta;
  const SnMeta *task_meta;   // record of the first front of every task: one load instead of three for the
                             // one-front tasks of the big-front kernels
  const ChildMeta *child_meta;
  const int32_t *fasm_src, *fasm_dst, *fdup_src, *fdup_dst;
  const int32_t *fasm_colptr;   // fronts beyond LDS: start of every permuted pivot column's entries in fasm_*
  const int32_t *scat, *rel;
  const int32_t *perm;      // permuted scalar -> reference scalar
  const int32_t *sn_rows;
  const T *hvals;
  const T *b;
  T *lvals;                 // factor panels (and whole big fronts)
  T *uvals;                 // packed update matrices
  T *xch;                   // sharded runs: exchange buffer of the boundary fronts' packed update matrices
  T *x;                     // solution, permuted order
  T *winv;                  // inverse diagonal blocks, kept for the back substitution: 16 x 16 per 16 columns of an
                            // LDS front ([block][j][c] = W(j, c)), 32 x 32 per 32 columns of a big front ([block][j][c] = W(c, j))
  int *err;
  unsigned long long *stamps;  // [S][12], diagnostic builds only (else null)
  unsigned long long *trace;   // diagnostic trace region (else null)
  // dataflow launches of the LDS fronts (lds_flow.hip.h; null / 0 elsewhere)
  const int32_t *child_dep;    // per child_meta entry: index in dep_flags of the flag that says the child's update matrix is in memory,
                               // -1 = the child was factored earlier in the same task (by this workgroup)
  const int32_t *parent_dep;   // per front, back substitution: index of the flag that says the parent's solution is in memory, -1 = none
                               // (a root, or the parent was solved earlier in the same task)
  unsigned *dep_flags;         // [0, S): factor flags, [S, 2 S): solve flags; zeroed before every factorisation
  int parent_dep_self;         // S: a front's own solve flag is dep_flags[S + s]
  unsigned long long wait_ticks;   // bound of one wait, 100 MHz ticks
  int solve_lds;                   // k_solve_flow: scalars of dynamic LDS (decides, front by front, whether its L11 image fits)
};

// ---- hand-offs between workgroups of ONE launch (lds_flow.hip.h; flow.hip.h has the same policy for the fronts beyond
// LDS): payload with sc1 accesses (written through to memory, read past this CU's L1), the flag with agent-scope relaxed
// atomics after the storing waves' s_waitcnt vmcnt(0) and a workgroup barrier.  A wait is bounded in TIME (100 MHz wall
// clock): when it runs out -- or another wait already has -- DEVERR_FLOW_TIMEOUT is set and every later wait returns at
// once, so the launch drains; addresses never depend on data, k_update does not apply a step while a flag is set.
__device__ __forceinline__ unsigned dep_flag_ld(const unsigned *p) { return __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dep_flag_set(unsigned *p) { __hip_atomic_store(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dep_drain() {   // after this wave's payload stores, before the barrier that precedes the flag
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool dep_wait(const unsigned *p, int *err, unsigned long long max_ticks) {   // wave-uniform p
  bool ok = true;
  if (dep_flag_ld(p) == 0u) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; spins++) {
      __builtin_amdgcn_s_sleep(1);
      if (dep_flag_ld(p) != 0u) break;
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
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the payload loads below the poll
  return ok;
}

// update-matrix element (i >= j) of an n x n lower triangle: packed columns, or
// a plain column-major square when ld > 0
__device__ __forceinline__ int64_t tri_index(int n, int ld, int i, int j) {
  return ld > 0 ? (int64_t)j * ld + i : (int64_t)j * n - (int64_t)j * (j - 1) / 2 + (i - j);
}

// ---- MFMA 16x16x4 tiles (f64: v_mfma_f64_16x16x4_f64, f32: v_mfma_f32_16x16x4_f32).
// A operand: lane l holds A[l & 15][l >> 4]; B operand: lane l holds B[l >> 4][l & 15];
// result register `reg` of lane l is D[row(l, reg)][l & 15] -- the row map differs
// between the f64 and the f32 instruction (cdna_hip_programming.md, Fragment layout).
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
  typedef double Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma16<float> {
  typedef float Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// One wave:  C(i, j) -= sum_{k in [ka, kb)} X[i][k] * X[j][k]  for the 16 x 16 tile
// i in [i0, i0+16), j in [j0, j0+16), restricted to i < imax, j < jmax, i >= j.
// X is column-major with leading dimension ldx.  The product is formed as
// D = (-X_j) * X_i^T so that a lane's four results are four different columns j
// of one row i: consecutive lanes touch consecutive i (contiguous in memory).
// Operands of four k-steps are fetched before the four dependent MFMAs issue.
template <typename T> __device__ __forceinline__ T pin(T v);
struct StoreInPlace { template <typename T> __device__ __forceinline__ void operator()(T *p, T v) const { *p = v; } };
template <typename T, typename CAddr, typename Store = StoreInPlace>
__device__ __forceinline__ void tile_rank_update(const T *X, int ldx, int i0, int j0, int imax, int jmax,
                                                 int ka, int kb, CAddr caddr, Store store = Store()) {
  using MM = Mfma16<T>;
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int i = i0 + li;
  // loads go to clamped addresses (a predicated load is a branch): rows past the tile's range only
  // feed entries that are never stored, columns past kb are zeroed on one operand.
  // The accumulator holds -C (one negation on the way in and out instead of one per operand).
  const int ic = min(i, imax - 1);
  typename MM::Acc acc;
  bool valid[4];
  T *pc[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = j0 + MM::row(lane, r);
    valid[r] = i < imax && j < jmax && i >= j;
    pc[r] = caddr(ic, min(min(j, jmax - 1), ic));   // the true address wherever valid
    acc[r] = -*pc[r];
  }
  // whole 16-column chunks need no clamp: one pointer per operand, bumped by 4 columns per load
  const int nfull = (kb - ka) >> 4;
  const int step = 4 * ldx;
  const T *xi = X + ic + (ka + lk) * ldx, *xj = X + min(j0 + li, jmax - 1) + (ka + lk) * ldx;
  T av[4], bv[4], an[4], bn[4];
  auto fetch = [&](T *xa, T *xb) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      xb[q] = xi[q * step];
      xa[q] = xj[q * step];
    }
    xi += 4 * step;
    xj += 4 * step;
  };
  if (nfull > 0) fetch(av, bv);
  for (int c = 0; c < nfull; c++) {   // operands of the next four k-steps are in flight under these MFMAs
    if (c + 1 < nfull) fetch(an, bn);
#pragma unroll
    for (int q = 0; q < 4; q++) acc = MM::mma(av[q], bv[q], acc);
#pragma unroll
    for (int q = 0; q < 4; q++) { av[q] = an[q]; bv[q] = bn[q]; }
  }
  const int rem = (kb - ka) & 15;
  if (rem > 0) {   // the last 1..15 columns: clamped column index, zero on one operand
#pragma unroll
    for (int q = 0; q < 4; q++) {   // the eight loads together, then pinned (a pin per load made each wait for its own round trip)
      const int off = (min(4 * q + lk, rem - 1) - lk) * ldx;
      bv[q] = xi[off];
      av[q] = xj[off];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      bv[q] = pin(bv[q]);
      const T va = pin(av[q]);
      av[q] = 4 * q + lk < rem ? va : (T)0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (4 * q < rem) acc = MM::mma(av[q], bv[q], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (valid[r]) store(pc[r], -acc[r]);
}

// Keeps an unconditional load unconditional: without it the compiler sinks a load whose value is only
// used under a select back into a branch (s_and_saveexec + s_cbranch per access).
template <typename T> __device__ __forceinline__ T pin(T v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Memory policy of the big-front device functions.  SC1 = false: plain accesses (every dependency of the launch is
// a kernel boundary).  SC1 = true (k_big_flow: tasks of ONE launch hand tiles to each other): agent-scope relaxed
// atomics = global_load / global_store ... sc1 -- the store is written through to memory, the load bypasses this
// CU's L1 and is served coherently by L2; no release / acquire fence is needed around them (MI355X_MICROARCH.md,
// inter-workgroup visibility; measured for this path with lines shared by two producers: scripts/handoff_probe.hip,
// profiles/archive/r03_handoff_probe.txt).
template <bool SC1, typename T> __device__ __forceinline__ T mem_ld(const T *p) {
  if constexpr (SC1) return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1, typename T> __device__ __forceinline__ void mem_st(T *p, T v) {
  if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// An entry of the solution vector that k_solve_flow has not produced yet holds X_PENDING_WORD in each of its 32-bit words
// (the linearisation kernel of the iteration puts it there): a NaN with a payload no arithmetic produces, in fp32 and fp64
// alike.  x_wait returns the entry once it is there -- the payload is its own flag: no separate flag round trip, and the
// producer neither drains its stores nor meets at a barrier before the consumers may go on.  Bounded in time like dep_wait.
__device__ __forceinline__ bool x_pending(float v) { return __builtin_bit_cast(unsigned, v) == X_PENDING_WORD; }
// fp64: the HIGH word alone decides (0x7ff8dead........ is a NaN whatever the low word is, so no value is mistaken for the
// mark).  An aligned 8-byte sc1 store was never seen torn on this chip (scripts/handoff_probe.hip, 8-byte payload-as-flag
// form: 0 of 1e9 hand-offs, profiles/r05_handoff_probe.txt), but x_wait does not rest on that: an entry whose high word is
// there while its low word still reads as the mark -- a torn store, or one value in 2^32 -- is read again until two
// consecutive reads agree (x_half).
__device__ __forceinline__ bool x_pending(double v) { return (unsigned)__double2hiint(v) == X_PENDING_WORD; }
__device__ __forceinline__ bool x_half(float) { return false; }
__device__ __forceinline__ bool x_half(double v) { return (unsigned)__double2loint(v) == X_PENDING_WORD; }
template <typename T> __device__ __forceinline__ T x_wait(const T *p, int *err, unsigned long long max_ticks) {
  T v = mem_ld<true>(p);
  if (x_pending(v)) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; spins++) {
      __builtin_amdgcn_s_sleep(1);
      v = mem_ld<true>(p);
      if (!x_pending(v)) break;
      if ((spins & 31u) == 0u) {
        const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e != 0 || wall_clock64() - t0 > max_ticks) {
          if (e == 0) atomicOr(err, DEVERR_FLOW_TIMEOUT);
          break;
        }
      }
    }
  }
  if (x_half(v)) {   // (never taken in fp32; in fp64 only for a torn store or a value whose low word equals the mark)
    for (int k = 0; k < 64; k++) {
      __builtin_amdgcn_s_sleep(1);
      const T w = mem_ld<true>(p);
      const bool same = __builtin_bit_cast(unsigned long long, (double)w) == __builtin_bit_cast(unsigned long long, (double)v);
      v = w;
      if (same) break;
    }
  }
  return v;
}

// The same policy through a buffer resource (a front, a W slot): sc1 accesses by 32-bit byte offset from a uniform
// base -- no 64-bit address arithmetic per access, immediate offsets fold into the instruction, and unlike atomics the
// compiler may schedule them freely (relaxed atomics are "ordered" memory operations to the scheduler: every address
// of a batch of loads stays live until its load has issued in source order, which cost k_big_flow ~100 VGPRs).
// aux = 16 is the sc1 bit of gfx940+ buffer instructions.
typedef unsigned rr_u2 __attribute__((ext_vector_type(2)));
template <typename T> struct Sc1Buf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Sc1Buf(const T *base, uint32_t bytes)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base), 0, (int)bytes, 0x00020000)) {}
  __device__ __forceinline__ T ld(uint32_t off) const {
    if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 16));
    else {
      const rr_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 16);
      return __hiloint2double((int)w.y, (int)w.x);
    }
  }
  __device__ __forceinline__ void st(uint32_t off, T v) const {
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, 16);
    else {
      const rr_u2 w = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
      __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)off, 0, 16);
    }
  }
};

// value of `v` in lane `src` (wave-uniform src): v_readlane, no LDS crossbar trip
__device__ __forceinline__ double lane_bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// Sum over the 64 lanes of a wave, result valid in lane 63: row shifts + row broadcasts on the DPP
// path (VALU speed; the ds_bpermute route of __shfl_down costs an LDS crossbar trip per step).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_get(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <typename T> __device__ __forceinline__ T wave_sum63(T v) {
  v += dpp_get<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row sum
  v += dpp_get<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}

// 1/sqrt(d): hardware seed + Newton steps (the library sqrt + divide pair is a
// ~40-instruction dependent chain and sits on the critical path of every column)
__device__ __forceinline__ double fast_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ float fast_rsqrt(float d) {
  float y = __builtin_amdgcn_rsqf(d);
  return y * (1.5f - 0.5f * d * y * y);
}

// 1/sqrt(d) with ONE third-order correction of the hardware seed (fp64: v_rsq_f64 is good to ~2^-23, two Newton
// steps were eight dependent operations on the chain of every pivot column; e = 1 - d y^2, y (1 + e/2 + 3 e^2/8) is
// four and leaves a relative error of (5/16) e^3 ~ 2^-70)
__device__ __forceinline__ double chain_rsqrt(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ float chain_rsqrt(float d) { return fast_rsqrt(d); }

// acc += bcast * b, where bcast is the value of `a` in lane J of the calling lane's 16-lane row: ONE instruction (DPP
// row_newbcast on the multiply-add itself; gfx90a+ allows it on the 64-bit FMA too) instead of v_readlane (x 2 in fp64)
// + FMA.  The caller keeps two wait states between the VALU write of `a` and the first use (s_nop 1).
template <int J> __device__ __forceinline__ void fmac_bcast(double &acc, double a, double b) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}
template <int J> __device__ __forceinline__ void fmac_bcast(float &acc, float a, float b) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}
// ---- the same sweep with the identity rows in the NEXT 16-lane row instead of in registers of the same lanes (the layout
// of chol16_invert: lanes 0..15 of every 32 hold the rows of the block, lanes 16..31 the rows of an identity): one
// multiply-add per column entry serves both, and the wave needs 16 registers per lane, not 32 -- the kernels that run
// 16 waves per workgroup have 128 VGPRs.  row_newbcast reads lane J of the lane's OWN row, so the identity lanes need the
// block rows' multipliers L(J, K) in their row: v_permlane16_swap (gfx950) copies the even rows of a register into the odd
// rows, two instructions per fp64 column.
template <bool NEG> __device__ __forceinline__ double even_rows_to_odd(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v) ^ (NEG ? 0x80000000u : 0u);
  unsigned lo2 = lo, hi2 = hi;
  // (the assembler's hazard recogniser does not look into inline assembly: keep the wait states a VALU write -> permlane read needs here)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
  return __hiloint2double((int)hi, (int)lo);
}
template <bool NEG> __device__ __forceinline__ float even_rows_to_odd(float v) {
  unsigned a = __float_as_uint(v) ^ (NEG ? 0x80000000u : 0u), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __uint_as_float(a);
}
template <typename T, int K, int J> struct Chol16Upd2 {
  static __device__ __forceinline__ void run(T (&x)[16], T nb, T l) {
    fmac_bcast<J>(x[J], nb, l);
    Chol16Upd2<T, K, J + 1>::run(x, nb, l);
  }
};
template <typename T, int K> struct Chol16Upd2<T, K, 16> {
  static __device__ __forceinline__ void run(T (&)[16], T, T) {}
};
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// 1/sqrt(d) in five stages (the operations of chain_rsqrt, one dependent step each), so that the sweep can put a few of a
// column's multiply-adds between two stages: a wave issues in order, and the inline-assembly multiply-adds are scheduling
// barriers to the compiler -- what is not interleaved in the source is not interleaved at all.
template <typename T> struct RsqStages;
template <> struct RsqStages<double> {
  double d, y, hy, e, t, u;
  __device__ __forceinline__ void s0(double dd) { d = dd; y = __builtin_amdgcn_rsq(d); }
  __device__ __forceinline__ void s1() { hy = d * y; }
  __device__ __forceinline__ void s2() { e = __builtin_fma(-hy, y, 1.0); }
  __device__ __forceinline__ void s3() { t = __builtin_fma(0.375, e, 0.5); u = y * e; }
  __device__ __forceinline__ double s4() { return __builtin_fma(u, t, y); }
};
template <> struct RsqStages<float> {
  float d, y, hy, e;
  __device__ __forceinline__ void s0(float dd) { d = dd; y = __builtin_amdgcn_rsqf(d); }
  __device__ __forceinline__ void s1() { hy = 0.5f * d * y; }
  __device__ __forceinline__ void s2() { e = __builtin_fmaf(-hy, y, 1.5f); }
  __device__ __forceinline__ void s3() {}
  __device__ __forceinline__ float s4() { return y * e; }
};
template <typename T, int K, int J, int JE> struct Chol16UpdR {   // the multiply-adds of column K for the entries J .. JE - 1
  static __device__ __forceinline__ void run(T (&x)[16], T nb, T l) {
    if constexpr (J < JE && J < 16) {
      if constexpr (J > K) fmac_bcast<J>(x[J], nb, l);
      Chol16UpdR<T, K, J + 1, JE>::run(x, nb, l);
    }
  }
};
template <typename T, int K> struct Chol16Col2 {
  // inv = 1 / sqrt(pivot K), already formed while column K - 1 was being applied
  static __device__ __forceinline__ void run(T (&x)[16], T inv, int ncols, bool &bad) {
    if (K >= ncols) return;   // (columns past a partial block are identity columns: sweeping them changes nothing)
    const T l = x[K] * inv;   // block rows: L(q, K) (garbage above the diagonal, see chol16_dpp); identity rows: their multiplier
    x[K] = l;
    T nb = even_rows_to_odd<true>(l);    // -L(., K) in both rows of every pair
    asm volatile("s_nop 1" : "+v"(nb));  // VALU write -> DPP read of the same register
    T invn = (T)1;
    if constexpr (K + 1 < 16) {
      // The NEXT pivot ahead of this column's update: d' = x[K+1](K+1) - L(K+1, K)^2 from two broadcast values, rounded
      // exactly as the multiply-add below rounds that entry; its 1/sqrt is formed in stages BETWEEN the column's
      // multiply-adds, so the chain of a column is multiply -> FMA -> rsqrt stages and the fifteen multiply-adds, the row
      // copy and the wait states fill its latencies instead of following it.
      const T la = lane_bcast(l, K + 1), b = lane_bcast(x[K + 1], K + 1);
      const T dn = fma_t(-la, la, b);
      bad = bad || !(dn > (T)0);
      RsqStages<T> r;
      __builtin_amdgcn_sched_barrier(0);
      r.s0(dn);
      __builtin_amdgcn_sched_barrier(0);
      Chol16UpdR<T, K, K + 1, K + 4>::run(x, nb, l);
      __builtin_amdgcn_sched_barrier(0);
      r.s1();
      __builtin_amdgcn_sched_barrier(0);
      Chol16UpdR<T, K, K + 4, K + 7>::run(x, nb, l);
      __builtin_amdgcn_sched_barrier(0);
      r.s2();
      __builtin_amdgcn_sched_barrier(0);
      Chol16UpdR<T, K, K + 7, K + 10>::run(x, nb, l);
      __builtin_amdgcn_sched_barrier(0);
      r.s3();
      __builtin_amdgcn_sched_barrier(0);
      Chol16UpdR<T, K, K + 10, K + 13>::run(x, nb, l);
      __builtin_amdgcn_sched_barrier(0);
      invn = r.s4();
      __builtin_amdgcn_sched_barrier(0);
      Chol16UpdR<T, K, K + 13, 16>::run(x, nb, l);
    }
    Chol16Col2<T, K + 1>::run(x, invn, ncols, bad);
  }
};
template <typename T> struct Chol16Col2<T, 16> {
  static __device__ __forceinline__ void run(T (&)[16], T, int, bool &) {}
};
// lanes (mod 32) 0..15: row q of the block in, row q of L out (GARBAGE above the diagonal); lanes 16..31: row q of an
// identity in, x[c] = W(c, q) out.  ncols < 16: the block is padded with an identity from column ncols on.
template <typename T> __device__ __forceinline__ bool chol16_dpp2(T (&x)[16], int ncols = 16) {
  const T d0 = lane_bcast(x[0], 0);
  bool bad = !(d0 > (T)0);
  Chol16Col2<T, 0>::run(x, chain_rsqrt(d0), ncols, bad);
  return bad;
}

// The same for a FULL block whose strict upper triangle in P is zero (fronts are zeroed before assembly
// and every later store is to i >= j) and with a 16 x 16 identity at wscr + 16 * 17: lanes 0..15 stream
// the rows of the block, lanes 16..31 the rows of the identity, through ONE per-lane (base, stride) pair
// -- no select on the way in or out (an fp64 select is two v_cndmask; the general version spends ~380
// instructions, 1.3 us of a lone wave, around the 1.9 us sweep).  W goes to LDS only; the caller copies
// it to global memory for the back substitution with another wave (copy_w16).
constexpr int W16_SCR = 16 * 17 + 256;   // LDS scalars: W scratch + identity
template <typename T>
__device__ __forceinline__ void diag16_factor_invert_full(T *P, int M, int k0, int *err, T *wscr) {
  const int lane = threadIdx.x & 63, ll = lane & 31, q = lane & 15;
  const bool rowlane = ll < 16;
  const T *src = rowlane ? P + k0 * M + k0 + q : wscr + 16 * 17 + q;   // element (q, c) at src[c * sstride]
  const int sstride = rowlane ? M : 16;
  T x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = src[c * sstride];
  const bool bad = chol16_dpp2<T>(x);   // (leaves garbage above the diagonal of the factored block: nothing reads it)
  if (bad && lane == 0) atomicOr(err, DEVERR_NOT_SPD);
  T *dst = rowlane ? P + k0 * M + k0 + q : wscr + q * 17;              // L(q, c) in place | W(c, q)
  const int dstride = rowlane ? M : 1;
#pragma unroll
  for (int c = 0; c < 16; c++) dst[c * dstride] = x[c];
}
// The same for a PARTIAL last block (nb < 16 columns), padded with an identity: rows q >= nb stream identity
// rows too, columns c >= nb of the block's own rows are redirected to a zero entry of the identity on the
// way in and to a dump slot (padding column of wscr) on the way out -- a select on a 32-bit LDS address per
// column instead of nested fp64 selects.
template <typename T>
__device__ __forceinline__ void diag16_factor_invert_part(T *P, int M, int k0, int nb, int *err, T *wscr) {
  const int lane = threadIdx.x & 63, ll = lane & 31, q = lane & 15;
  const bool rowlane = ll < 16;
  const bool prow = rowlane && q < nb;   // lanes that hold a row of the block
  T *ident = wscr + 16 * 17;
  const T *src = prow ? P + k0 * M + k0 + q : ident + q;
  const int sstride = prow ? M : 16;
  T x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const T *pc = (prow && c >= nb) ? ident + 1 : src + c * sstride;
    x[c] = *pc;
  }
  const bool bad = chol16_dpp2<T>(x, nb);
  if (bad && lane == 0) atomicOr(err, DEVERR_NOT_SPD);
  T *dump = wscr + 16;
  T *dst = prow ? P + k0 * M + k0 + q : rowlane ? dump : wscr + q * 17;
  const int dstride = prow ? M : rowlane ? 0 : 1;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    T *pc = (prow && c >= nb) ? dump : dst + c * dstride;
    *pc = x[c];
  }
}
// wscr (W(c, q) at [q * 17 + c]) -> wout[c * 16 + q], one wave
template <typename T> __device__ __forceinline__ void copy_w16(const T *wscr, T *wout) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e = lane + 64 * u;
    wout[e] = wscr[(e & 15) * 17 + (e >> 4)];
  }
}
template <typename T> __device__ __forceinline__ void init_w16_identity(T *wscr, int tid, int nthreads) {
  for (int t = tid; t < 256; t += nthreads) wscr[16 * 17 + t] = (t >> 4) == (t & 15) ? (T)1 : (T)0;
}

// Workgroup-wide partial Cholesky of the leading nc columns of the M x nc panel P (column-major,
// ld M; rows nc.. are the off-diagonal rows and the rhs row), blocked by 16 columns:
//   diagonal block   registers of wave 0 (row per lane, v_readlane broadcasts, rsqrt + Newton), which
//                    also yields W = L11^-1 from 16 identity rows riding along
//   rows below       X = A W^T on the matrix cores, one 16-row tile per wave at a time
//   trailing update  16 x 16 MFMA tiles, one per wave at a time
// with a one-block LOOK-AHEAD: after the triangular solve of block k only the next block column is
// updated by everybody; the rest of block k's update runs on waves 1.. while wave 0 already factors
// diagonal block k+1 (the two touch disjoint columns).
// FAST: the strict upper triangles of P's diagonal blocks are zero and wscr has W16_SCR scalars with the
// identity set (init_w16_identity) -- full blocks then take diag16_factor_invert_full.
// nu > 0: the front's (nu x nu) update matrix lives at uaddr(i, j) (rows nc.. of the panel are its operand):
// the Schur complement of every block but the last is then applied here, block by block, by the waves that
// idle while wave 0 factors the next diagonal block; the caller applies the last block (columns
// 16 * ((nc - 1) / 16) .. nc) after the panel.
struct NoUAddr { __device__ float *operator()(int, int) const { return nullptr; } };
template <typename T, int THREADS, bool FAST = false, typename UAddr = NoUAddr>
__device__ __forceinline__ void panel_factor(T *P, int M, int nc, int *err, T *wscr /* 16 * 17 (FAST: W16_SCR) scalars of LDS */, T *wout,
                             unsigned long long *acc = nullptr, int nu = 0, UAddr uaddr = UAddr(), bool defer_u = false) {
  using MM = Mfma16<T>;
  constexpr int NB = 16;
  constexpr int NW = THREADS / 64;
  const int tid = threadIdx.x;
  const int wave = wave_index(), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  RRPGO_ACC_DECL();
  int pend_k0 = -1;   // block whose "rest" update (block columns 1.. of its trailing part) is still owed
  auto rest_update = [&](int pk0, int first_wave, int nwaves) {
    // tiles with jb >= 1 of the update of block pk0 (full 16 columns by construction)
    const int js = pk0 + NB;
    const int nj = (nc - js + 15) >> 4, ni = (M - js + 15) >> 4;
    // tiles (ib >= jb, jb >= 1) dealt round-robin over the waves (balanced: no wave draws the skipped ones)
    int total = 0;
    for (int jb = 1; jb < nj; jb++) total += max(ni - jb, 0);
    const int nt = (nu + 15) >> 4;   // tiles of the update matrix (lower triangle), after the panel's own
    const int utotal = defer_u ? 0 : nt * (nt + 1) / 2;   // deferred: the caller subtracts ALL pivot columns from the update matrix in one pass
    for (int t = wave - first_wave; t < total + utotal; t += nwaves) {
      if (t < total) {
        int jb = 1, rem = t;
        while (rem >= ni - jb) { rem -= ni - jb; jb++; }
        const int ib = jb + rem;
        tile_rank_update<T>(P, M, js + 16 * ib, js + 16 * jb, M, nc, pk0, pk0 + NB,
                            [&](int i, int j) { return P + j * M + i; });
      } else {
        if constexpr (!std::is_same<UAddr, NoUAddr>::value) {
          int jb = 0, rem = t - total;
          while (rem >= nt - jb) { rem -= nt - jb; jb++; }
          const int ib = jb + rem;
          tile_rank_update<T>(P + nc, M, 16 * ib, 16 * jb, nu, nu, pk0, pk0 + NB, uaddr);
        }
      }
    }
  };
  for (int k0 = 0; k0 < nc; k0 += NB) {
    const int nb = min(NB, nc - k0);
    RRPGO_ACC_BEGIN();
    constexpr bool fast = FAST;
    auto diag = [&] {
      static_assert(FAST, "the general diagonal-block form (fronts factored in place in global memory by this routine) was retired in r05");
      if (nb == NB) diag16_factor_invert_full<T>(P, M, k0, err, wscr);
      else diag16_factor_invert_part<T>(P, M, k0, nb, err, wscr);
    };
    if (NW == 1) {
      if (pend_k0 >= 0) rest_update(pend_k0, 0, 1);
      diag();
    } else if (wave == 0) {
      diag();
    } else if (pend_k0 >= 0) {
      rest_update(pend_k0, 1, NW - 1);
    }
    pend_k0 = -1;
    __syncthreads();
    RRPGO_ACC_END(acc, 7);
    RRPGO_ACC_BEGIN();
    // rows below the diagonal block: X(i, c) = sum_j A(i, j) W(c, j); tile rows = c, tile columns = i
    {
      const int r0 = k0 + nb;
      const int ntiles = (M - r0 + 15) >> 4;
      T wa[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) wa[s4] = wscr[(4 * s4 + lk) * 17 + li];
      if (fast && wout && wave == NW - 1) copy_w16<T>(wscr, wout + (k0 >> 4) * 256);   // kept for the back substitution
      for (int ib = wave; ib < ntiles; ib += NW) {
        const int i = r0 + 16 * ib + li;
        const T *arow = P + min(i, M - 1);
        typename MM::Acc x = {0, 0, 0, 0};
        T av[4];   // the four operands requested together (a pin per load, as it was, made each wait for its own round trip)
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) av[s4] = arow[(k0 + min(4 * s4 + lk, nb - 1)) * M];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) av[s4] = pin(av[s4]);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int j = 4 * s4 + lk;
          x = MM::mma(wa[s4], j < nb ? av[s4] : (T)0, x);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int c = MM::row(lane, r);
          if (i < M && c < nb) P[(k0 + c) * M + i] = x[r];
        }
      }
    }
    __syncthreads();
    RRPGO_ACC_END(acc, 8);
    // next block column only (jb == 0): P(i,j) -= L(i, blk) L(j, blk)^T, j in [k0+nb, k0+nb+16)
    const int js = k0 + nb;
    if (js < nc) {
      RRPGO_ACC_BEGIN();
      const int ni = (M - js + 15) >> 4;
      for (int ib = wave; ib < ni; ib += NW)
        tile_rank_update<T>(P, M, js + 16 * ib, js, M, nc, k0, k0 + nb,
                            [&](int i, int j) { return P + j * M + i; });
      // no barrier here: the first wave owns tile 0 (the next diagonal block) and goes straight on to
      // factor it; the other tiles of this block column are first read by the triangular solve of the next
      // block, which comes after the barrier that follows that factorisation
      RRPGO_ACC_END(acc, 9);
      if (js + NB < nc || nu > 0) pend_k0 = k0;   // more block columns to the right (or the update matrix): owed, done under the next diagonal block
    }
  }
}


// Assemble, factor and publish one front.  P is the M x nc pivot panel
// (column-major, ld M), U the (nr+1) x (nr+1) update matrix (packed lower when
// uld == 0, column-major with leading dimension uld otherwise).  For fronts
// held in LDS (P and U contiguous: U = P + M*nc) both are copied to global
// memory at the end; a front that lives in global memory (IN_PLACE) is already
// where it has to be.
// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the vector
// memory counter, i.e. it would wait for global loads that were requested early on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FLOW (lds_flow.hip.h: fronts of several workgroups of ONE launch): a child that another workgroup factors is waited
// for right before ITS extend-add -- everything that does not depend on it (zeroing, H entries, rhs, the children before
// it, its scatter map) is done by then; update matrices move with sc1 accesses; the front's flag is set as soon as its
// update matrix is in memory, the panel is copied out after that.  The arithmetic is the same in both forms.
template <typename T, int THREADS, bool IN_PLACE, bool FLOW = false>
__device__ void process_front(const FactorArgs<T> &a, int s, const SnMeta &m, T *P, T *U, int uld, T *dinv) {
  static_assert(!(FLOW && IN_PLACE), "the dataflow form is for fronts in LDS");
  const int tid = threadIdx.x;
  const int nc = m.nc, nr = m.nr;
  const int M = nc + nr + 1, nu = nr + 1;
  const int psize = M * nc, usize = nu * (nu + 1) / 2;
  RRPGO_STAMP(a, s, 0);
#ifdef RRPGO_STAMPS
  if (FLOW && tid == 0 && a.stamps) a.stamps[(int64_t)s * 12 + 10] = 0;
#endif
  // ---- original entries of H that live in this front's pivot columns and the rhs row: index, then
  // value -- two dependent global round trips (~1.2 us).  The first APRE entries of every thread are
  // requested BEFORE the zeroing pass and its barrier, which hides them.
  constexpr int APRE = 2;
  const int32_t *asrc = a.fasm_src + m.asm_begin, *adst = a.fasm_dst + m.asm_begin;
  int pd[APRE];
  T pv[APRE], pb = 0;
#pragma unroll
  for (int u = 0; u < APRE; u++) {
    const int t = tid + u * THREADS;
    pd[u] = -1;
    pv[u] = 0;
    if (t < m.asm_count) { pd[u] = adst[t]; pv[u] = a.hvals[asrc[t]]; }
  }
  if (tid < nc) pb = a.b[a.perm[m.col0 + tid]];
  ChildMeta cnext{};
  int dnext = -1;   // FLOW: flag of the child cnext, -1 = nothing to wait for
  if (m.child_count > 0) {
    cnext = a.child_meta[m.child_begin];
    if (FLOW) dnext = a.child_dep[m.child_begin];
  }
  if (IN_PLACE) {
    for (int64_t t = tid; t < (int64_t)M * M; t += THREADS) P[t] = 0;  // whole front, ld M
  } else {
    if constexpr (sizeof(T) == 4) {
      // fp32 (the 150 KB fronts of the large graphs): 16-byte stores (P is 16-byte aligned), a quarter of the LDS
      // instructions; the last 1 - 3 scalars one by one.  (fp64: the small graphs' fronts, measured 0.4 % slower this way.)
      using V4z = typename VecT<T>::V4;
      const int n4 = (psize + usize) >> 2;
      V4z *P4 = reinterpret_cast<V4z *>(P);
      for (int t = tid; t < n4; t += THREADS) P4[t] = V4z{0, 0, 0, 0};
      if (tid < ((psize + usize) & 3)) P[4 * n4 + tid] = 0;
    } else {
      for (int t = tid; t < psize + usize; t += THREADS) P[t] = 0;
    }
  }
  constexpr int EPRE = 4;
  int ed[EPRE];
  T ev[EPRE];
  auto child_u = [&](const ChildMeta &c) { return (c.uld > 0 ? a.lvals : c.uld < 0 ? a.xch : a.uvals) + c.uoff; };
  // a child's update matrix: plain loads, or (FLOW) sc1 loads by 32-bit byte offset -- another workgroup of this launch wrote it
  auto uc_ld = [&](const T *Uc, const Sc1Buf<T> &ub, int t) -> T {
    if constexpr (FLOW) return ub.ld((uint32_t)t * (uint32_t)sizeof(T));
    else return Uc[t];
  };
  auto prefetch_map = [&](const ChildMeta &c) {
#pragma unroll
    for (int u = 0; u < EPRE; u++) { ed[u] = -1; ev[u] = 0; }
    if (IN_PLACE || c.scat_ptr < 0) return;
    const int32_t *map = a.scat + c.scat_ptr;
    const int cnt = c.ncu * (c.ncu + 1) / 2;
#pragma unroll
    for (int u = 0; u < EPRE; u++) {
      const int t = tid + u * THREADS;
      if (t < cnt) ed[u] = map[t];
    }
  };
  auto prefetch_val = [&](const ChildMeta &c) {
    if (IN_PLACE || c.scat_ptr < 0) return;
    const T *Uc = child_u(c);
    const int cnt = c.ncu * (c.ncu + 1) / 2;
    [[maybe_unused]] const Sc1Buf<T> ub(Uc, FLOW ? (uint32_t)cnt * (uint32_t)sizeof(T) : 0u);
#pragma unroll
    for (int u = 0; u < EPRE; u++) {
      const int t = tid + u * THREADS;
      if (t < cnt) ev[u] = uc_ld(Uc, ub, t);
    }
  };
  auto prefetch = [&](const ChildMeta &c, int dep) {   // the values only when the child's update matrix is known to be there
    prefetch_map(c);
    if (!FLOW || dep < 0) prefetch_val(c);
  };
  if (m.child_count > 0) prefetch(cnext, dnext);   // under the barrier and the assembly stores
  if (IN_PLACE) __syncthreads();
  else lds_barrier();
  RRPGO_STAMP(a, s, 1);
  // plain stores: every destination is hit once
  {
#pragma unroll
    for (int u = 0; u < APRE; u++)
      if (pd[u] >= 0) P[pd[u]] = pv[u];
    for (int t = tid + APRE * THREADS; t < m.asm_count; t += THREADS) P[adst[t]] = a.hvals[asrc[t]];
    if (tid < nc) P[tid * M + (M - 1)] = pb;
    for (int j = tid + THREADS; j < nc; j += THREADS) P[j * M + (M - 1)] = a.b[a.perm[m.col0 + j]];
  }
  if (m.dup_count > 0) {  // blocks of parallel edges (rare): serial, fixed order
    __syncthreads();
    if (tid == 0)
      for (int t = 0; t < m.dup_count; t++) P[a.fdup_dst[m.dup_begin + t]] += a.hvals[a.fdup_src[m.dup_begin + t]];
  }
  RRPGO_STAMP(a, s, 2);
  // ---- extend-add of the children's update matrices, fixed child order.  A packed child into an LDS
  // parent has one precomputed destination per element; the first EPRE elements per thread of child q+1
  // are requested while child q is being added (each child is otherwise a global round trip + a barrier).
  for (int q = 0; q < m.child_count; q++) {
    const ChildMeta c = cnext;
    const int dep = dnext;
    if (q + 1 < m.child_count) {   // the next record in flight under this child
      cnext = a.child_meta[m.child_begin + q + 1];
      if (FLOW) dnext = a.child_dep[m.child_begin + q + 1];
    }
    const T *Uc = child_u(c);
    if (FLOW && dep >= 0) {   // every wave for itself; the barrier below joins them
#ifdef RRPGO_STAMPS
      const unsigned long long w0_ = wall_clock64();
#endif
      dep_wait(a.dep_flags + dep, a.err, a.wait_ticks);
#ifdef RRPGO_STAMPS
      if (tid == 0 && a.stamps) a.stamps[(int64_t)s * 12 + 10] += wall_clock64() - w0_;   // time spent waiting for children
#endif
    }
    __syncthreads();
    if (!IN_PLACE && c.scat_ptr >= 0) {
      // packed child, LDS parent: one precomputed destination per element, coalesced
      const int32_t *map = a.scat + c.scat_ptr;
      const int cnt = c.ncu * (c.ncu + 1) / 2;
      [[maybe_unused]] const Sc1Buf<T> ub(Uc, FLOW ? (uint32_t)cnt * (uint32_t)sizeof(T) : 0u);
      if (FLOW && dep >= 0) prefetch_val(c);   // (its map entries were requested before the wait)
#pragma unroll
      for (int u = 0; u < EPRE; u++)
        if (ed[u] >= 0) P[ed[u]] += ev[u];
      if (q + 1 < m.child_count) prefetch(cnext, dnext);
      int t = tid + EPRE * THREADS;
      for (; t + 3 * THREADS < cnt; t += 4 * THREADS) {
        const int d0 = map[t], d1 = map[t + THREADS], d2 = map[t + 2 * THREADS], d3 = map[t + 3 * THREADS];
        const T v0 = uc_ld(Uc, ub, t), v1 = uc_ld(Uc, ub, t + THREADS), v2 = uc_ld(Uc, ub, t + 2 * THREADS), v3 = uc_ld(Uc, ub, t + 3 * THREADS);
        if (d0 >= 0) P[d0] += v0;
        if (d1 >= 0) P[d1] += v1;
        if (d2 >= 0) P[d2] += v2;
        if (d3 >= 0) P[d3] += v3;
      }
      for (; t < cnt; t += THREADS) {
        const int d0 = map[t];
        if (d0 >= 0) P[d0] += uc_ld(Uc, ub, t);
      }
    } else {
      if (q + 1 < m.child_count) prefetch(cnext, dnext);
      const int32_t *rel = a.rel + c.rel_ptr;
      const int ncu = c.ncu;
      for (int t = tid; t < ncu * ncu; t += THREADS) {
        const int j = t / ncu, i = t - j * ncu;
        if (i < j || t == ncu * ncu - 1) continue;  // lower triangle; (rhs, rhs) corner is never used
        const T v = Uc[tri_index(ncu, c.uld, i, j)];
        const int li = rel[i], lj = rel[j];
        if (lj < nc) P[(int64_t)lj * M + li] += v;
        else U[tri_index(nu, uld, li - nc, lj - nc)] += v;
      }
    }
  }
  __syncthreads();
  RRPGO_STAMP(a, s, 3);
  // ---- partial factorisation + Schur complement
  // A wide update matrix (fp32 fronts of the large graphs: ~240 rows below ~50 pivot columns) is not updated block by block
  // under the diagonal chain -- every 16 columns would move the whole triangle through the MFMA pipe and back into LDS
  // (5 us per block, r03 stamps) -- but once, by ALL pivot columns, after the panel: the same FMAs in the same order
  // (bit-identical), the triangle read and written once.  Lattice -30 us per iteration; narrow fronts keep the overlap
  // (intel with every front deferred: -2 %), and the fp64 instantiation is compiled without the option.
#ifndef RRPGO_DEFER_U_MIN
#define RRPGO_DEFER_U_MIN 128
#endif
  const bool defer_u = sizeof(T) == 4 && nu >= RRPGO_DEFER_U_MIN && nc > 16;
  auto uaddr = [&](int i, int j) {   // 32-bit index arithmetic for the packed triangle in LDS
    return IN_PLACE ? U + ((int64_t)j * uld + i) : U + (j * nu - ((j * (j - 1)) >> 1) + (i - j));
  };
  {
#ifdef RRPGO_STAMPS
  if (tid == 0 && a.stamps) { a.stamps[(int64_t)s * 12 + 7] = 0; a.stamps[(int64_t)s * 12 + 8] = 0; a.stamps[(int64_t)s * 12 + 9] = 0; }
  panel_factor<T, THREADS, !IN_PLACE>(P, M, nc, a.err, dinv, a.winv + (int64_t)m.wblk * 256, a.stamps ? a.stamps + (int64_t)s * 12 : nullptr, nu, uaddr, defer_u);
#else
  panel_factor<T, THREADS, !IN_PLACE>(P, M, nc, a.err, dinv, a.winv + (int64_t)m.wblk * 256, nullptr, nu, uaddr, defer_u);
#endif
  }
  RRPGO_STAMP(a, s, 4);
  {
    // U(i,j) -= sum_k L21[i][k] L21[j][k]  (i >= j), 16 x 16 tiles on the matrix cores: the columns of the
    // LAST 16-column block only, the earlier blocks went in under the diagonal chain (panel_factor)
    const int klast = defer_u ? 0 : ((nc - 1) >> 4) << 4;
    const int nt = (nu + 15) >> 4;
    const int wave = wave_index();
    // the tiles of the lower triangle dealt round-robin: t-th tile of the column-major enumeration
    // FLOW: this pass leaves every entry of the update matrix final, and nothing reads it from LDS again: the results go
    // straight to its place in memory (written through), not back into LDS for a copy-out pass of their own
    [[maybe_unused]] T *Ug0 = (m.uld < 0 ? a.xch : a.uvals) + m.uoff;
    [[maybe_unused]] const Sc1Buf<T> ub0(Ug0, (FLOW && !IN_PLACE) ? (uint32_t)usize * (uint32_t)sizeof(T) : 0u);
    auto to_memory = [&](T *p, T v) { ub0.st((uint32_t)(p - U) * (uint32_t)sizeof(T), v); };
    for (int t = wave; t < nt * (nt + 1) / 2; t += THREADS / 64) {
      int jb = 0, rem = t;
      while (rem >= nt - jb) { rem -= nt - jb; jb++; }
      const int ib = jb + rem;
      if constexpr (FLOW && !IN_PLACE) tile_rank_update<T>(P + nc, M, 16 * ib, 16 * jb, nu, nu, klast, nc, uaddr, to_memory);
      else tile_rank_update<T>(P + nc, M, 16 * ib, 16 * jb, nu, nu, klast, nc, uaddr);
    }
  }
  if constexpr (FLOW && !IN_PLACE) dep_drain();
  __syncthreads();
  RRPGO_STAMP(a, s, 5);
  if (!IN_PLACE) {
    using V2 = typename VecT<T>::V2;
    // panel -> L storage, packed update -> U storage (both offsets are multiples of 4 scalars)
    T *Lg = a.lvals + m.loff;
    T *Ug = (m.uld < 0 ? a.xch : a.uvals) + m.uoff;
    if constexpr (FLOW) {
      // the update matrix is in memory (the pass above, drained ahead of the barrier): the parent's workgroup may go on --
      // the panel (read by the back substitution, a later launch) is copied out behind the flag
      (void)Ug;
      if (tid == 0) dep_flag_set(a.dep_flags + s);
      RRPGO_STAMP(a, s, 11);
    }
    if ((psize & 1) == 0) {
      const V2 *sp = reinterpret_cast<const V2 *>(P);
      V2 *dp = reinterpret_cast<V2 *>(Lg);
      for (int t = tid; t < (psize >> 1); t += THREADS) dp[t] = sp[t];
    } else {
      for (int t = tid; t < psize; t += THREADS) Lg[t] = P[t];
    }
    if constexpr (!FLOW)
      for (int t = tid; t < usize; t += THREADS) Ug[t] = U[t];
    __syncthreads();
  }
  RRPGO_STAMP(a, s, 6);
}

// One workgroup per task; a task is a list of supernodes in elimination order
// whose fronts are assembled, factored and pushed to global memory one after
// the other, entirely out of LDS.
template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_factor_tasks(FactorArgs<T> a) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ T dinv[W16_SCR];   // inverse of the current 16 x 16 diagonal block + an identity (diag16_factor_invert_full)
  T *smem = reinterpret_cast<T *>(smem_raw);
  init_w16_identity<T>(dinv, threadIdx.x, THREADS);   // process_front has barriers before the first use
  // the wave that carries every diagonal block shares its SIMD's issue slots with waves 4, 8, 12 of the workgroup:
  // it is served first (measured: intel.g2o +0.7 %, M3500 +0.9 %)
  if (THREADS > 256 && wave_index() == 0) __builtin_amdgcn_s_setprio(RRPGO_CHAIN_PRIO);
  const int task = a.task_begin + blockIdx.x;
  // the record of the NEXT front is requested before the current one is processed: index, then record, are
  // two dependent scalar loads (~0.5 us) that would otherwise sit between any two fronts of a task
  const int sbeg = a.task_ptr[task], send = a.task_ptr[task + 1];
  int snext = a.task_sn[sbeg];
  SnMeta mnext = a.sn_meta[snext];
  for (int si = sbeg; si < send; si++) {
    const int s = snext;
    const SnMeta m = mnext;
    if (si + 1 < send) {
      snext = a.task_sn[si + 1];
      mnext = a.sn_meta[snext];
    }
    process_front<T, THREADS, false>(a, s, m, smem, smem + (m.nc + m.nr + 1) * m.nc, 0, dinv);
  }
}

// ---- huge fronts: many workgroups per front, one launch per phase, the huge fronts of one
// level batched in the launch (grid y or z = front slot; the level's fronts are the single-front
// tasks task_begin, task_begin+1, ...).  Each M x M front lies in L storage (column-major, ld M).
//   k_big_build + k_big_assemble          ONE gather pass builds the pivot columns, then the H entries and the rhs
//   per 128-column super-panel (left-looking inside it):
//       4 x k_big_panel32   rows below a 32-column block: update from the super-panel's earlier columns,
//                           multiply by the inverse diagonal block; its first wave prepares the next block
//                           (the level's very first block: inside its first launch, every workgroup for itself)
//       k_big_update        everything right of the super-panel (K = 128, the dense MFMA contraction); its tile (0, 0)
//                           factors and inverts the next super-panel's first diagonal block
//   levels of few fronts and few tasks: k_big_flow (flow.hip.h) runs all of that as ONE launch of ticket-ordered tasks
//   back substitution: k_big_gemv_partial (L21^T x over the chip, row slices), then k_big_solve_sp (wide pivot blocks,
//   per 128 columns over the chip) or k_solve_mid (one workgroup per front; sums the slices, L11)
#ifndef RRPGO_BIG_NB
#define RRPGO_BIG_NB 32
#endif
#ifndef RRPGO_BIG_SUPER
#define RRPGO_BIG_SUPER 128
#endif
constexpr int BIG_NB = RRPGO_BIG_NB;        // pivot block width
constexpr int BIG_SUPER = RRPGO_BIG_SUPER;  // super-panel width = K of the big trailing update

// columns [0, big_built_cols) of a front are written by k_big_build when the first trailing update gathers the rest
// from the children: the pivot columns, rounded up to the tile grid of that update (tiles start at column 128 when
// nc > 128 and are 64 or 128 wide)
// Schur origin of a front whose trailing updates are split (schur_split): rows / columns from here on are formed by ONE
// pass of k_big_schur over all pivot columns.  A front of a single super-panel: nc (that pass is its only update);
// else nc rounded up to the tile grid the earlier super-panels' tiles share (their last tile columns reach to it).
__device__ __host__ __forceinline__ int big_schur_origin(int nc, int tile) {   // tile: k_big_schur's tile edge, 64 or 128
  return nc <= BIG_SUPER ? nc : ((nc + tile - 1) & ~(tile - 1));   // (with 128 this is big_built_cols: a tile is gathered or loaded as a whole)
}
__device__ __host__ __forceinline__ int big_built_cols(int nc, int M) {
  const int r = ((nc + 127) / 128) * 128;
  return nc <= BIG_SUPER ? nc : (r < M ? r : M);
}

// add == 0: plain stores into the zeroed front (before the extend-adds); add == 1: on top of what k_big_build
// gathered from the children (every destination appears once in the list, so neither form needs atomics)
template <typename T> __global__ void __launch_bounds__(256) k_big_assemble(FactorArgs<T> a, int add) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.y];
  T *F = a.lvals + m.loff;
  const int M = m.nc + m.nr + 1;
  const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
  const int32_t *src = a.fasm_src + m.asm_begin, *dst = a.fasm_dst + m.asm_begin;
  if (add) {
    for (int t = gid; t < m.asm_count; t += gsz) F[dst[t]] += a.hvals[src[t]];
    for (int j = gid; j < m.nc; j += gsz) F[(int64_t)j * M + (M - 1)] += a.b[a.perm[m.col0 + j]];
  } else {
    for (int t = gid; t < m.asm_count; t += gsz) F[dst[t]] = a.hvals[src[t]];
    for (int j = gid; j < m.nc; j += gsz) F[(int64_t)j * M + (M - 1)] = a.b[a.perm[m.col0 + j]];
  }
}

// Zeroing + every extend-add of a level in ONE pass: a wave per column of the parent, lanes over its rows; each
// entry is the sum, in child order, of what the children hold for it (inverse maps: ChildMeta::scat_ptr), written
// once.  The front is never read here: against a zeroing pass + one read-modify-write launch per child (r01) this moves
// (parent once + children once) instead of (parent 1 + 2 x children-that-touch-it) times.  Rows (J & ~63)..J-1 of
// column J are cleared too (the diagonal-block kernels read whole squares).
// QB children per batch: all their index loads are requested before the first value load, so a batch pays two
// dependent round trips; a child without this column (jq < 0) is read at a clamped address and masked.
// Args: any argument block with scat, lvals, uvals, xch (FactorArgs, FlowArgs).  SC1 (k_big_flow's cross-level form: children
// and parent are fronts of ONE launch): the children's entries are read past this CU's L1 and the column is written through.
template <typename T, int QB, bool SC1 = false, typename Args = FactorArgs<T>>
__device__ __forceinline__ void big_build_column(const Args &a, const ChildMeta *cm, int nkids, int M, int J, T *col, int lane) {
  for (int r0 = (J & ~63) + lane; r0 < M; r0 += 256) {
    T acc[4] = {0, 0, 0, 0};
    for (int q0 = 0; q0 < nkids; q0 += QB) {
      int jq[QB], last[QB], iq[QB][4];
      const T *ucol[QB];
#pragma unroll
      for (int qq = 0; qq < QB; qq++) {
        const ChildMeta c = cm[min(q0 + qq, nkids - 1)];
        const int32_t *inv = a.scat + c.scat_ptr;
        const int j = __builtin_amdgcn_readfirstlane(inv[J]);   // the child's column for this parent column (wave-uniform)
        jq[qq] = q0 + qq < nkids ? j : -1;
        const int jc = max(jq[qq], 0);
        const T *Uc = (c.uld > 0 ? a.lvals : c.uld < 0 ? a.xch : a.uvals) + c.uoff;
        ucol[qq] = Uc + (c.uld > 0 ? (int64_t)jc * c.uld : (int64_t)jc * c.ncu - (int64_t)jc * (jc - 1) / 2 - jc);
        last[qq] = c.ncu - 1;       // the (rhs, rhs) corner is never used (and never written by an LDS child)
#pragma unroll
        for (int u = 0; u < 4; u++) iq[qq][u] = inv[min(r0 + 64 * u, M - 1)];
      }
      T uv[QB][4];
#pragma unroll
      for (int qq = 0; qq < QB; qq++)
#pragma unroll
        for (int u = 0; u < 4; u++) uv[qq][u] = mem_ld<SC1>(ucol[qq] + max(iq[qq][u], max(jq[qq], 0)));   // rows above the column map below jq
#pragma unroll
      for (int qq = 0; qq < QB; qq++)
#pragma unroll
        for (int u = 0; u < 4; u++)
          acc[u] += (jq[qq] >= 0 && iq[qq][u] >= jq[qq] && !(iq[qq][u] == last[qq] && jq[qq] == last[qq])) ? uv[qq][u] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (r0 + 64 * u < M) mem_st<SC1>(col + r0 + 64 * u, acc[u]);
  }
}

// pivot_only: columns [0, big_built_cols) only -- the first trailing update forms the rest (k_big_update, gather).
// with_h: the wave that wrote a pivot column then adds the column's H entries and its right-hand-side entry itself
// (fasm_* sorted by destination, fasm_colptr) -- what the k_big_assemble launch of every level did (r03: 11 launches,
// 71 us per iteration of the 1M-edge lattice).  The entries go on top of what this wave's OTHER lanes have just stored:
// the stores are drained first (write-through: they are in L2 then) and the values read back past the CU's L1, which
// may still hold the previous iteration's line.  Same sums as k_big_assemble: bit-identical.
template <typename T> __global__ void __launch_bounds__(256) k_big_build(FactorArgs<T> a, int pivot_only, int with_h) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.y];
  const int M = m.nc + m.nr + 1;
  T *F = a.lvals + m.loff;
  const int wave = wave_index(), lane = threadIdx.x & 63;
  const ChildMeta *cm = a.child_meta + m.child_begin;
  const int Jend = pivot_only ? big_built_cols(m.nc, M) : M;
  for (int J = blockIdx.x * 4 + wave; J < Jend; J += gridDim.x * 4) {
    T *col = F + (int64_t)J * M;
    if (m.child_count <= 2) big_build_column<T, 2>(a, cm, m.child_count, M, J, col, lane);
    else big_build_column<T, 4>(a, cm, m.child_count, M, J, col, lane);
  }
  if (!with_h) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int J = blockIdx.x * 4 + wave; J < m.nc; J += gridDim.x * 4) {
    const int t0 = a.fasm_colptr[m.col0 + J], t1 = a.fasm_colptr[m.col0 + J + 1];
    for (int t = t0 + lane; t < t1; t += 64) {
      T *p = F + a.fasm_dst[t];
      *p = mem_ld<true>(p) + a.hvals[a.fasm_src[t]];
    }
    if (lane == 0) {
      T *p = F + (int64_t)J * M + (M - 1);
      *p = mem_ld<true>(p) + a.b[a.perm[m.col0 + J]];
    }
  }
}

// blocks of parallel edges (rare): serial, after the plain stores
template <typename T> __global__ void k_big_assemble_dup(FactorArgs<T> a) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.y];
  T *F = a.lvals + m.loff;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int t = 0; t < m.dup_count; t++) F[a.fdup_dst[m.dup_begin + t]] += a.hvals[a.fdup_src[m.dup_begin + t]];
}

// ---- left-looking 32-column blocks inside a 128-column super-panel -------------------------------
// One wave factors the (<= 32)^2 diagonal block and inverts it, as a 2 x 2 recursion over 16 x 16
// blocks: the two diagonal blocks in registers (above), the off-diagonal blocks on the matrix cores
//   L21 = A21 W11^T,  S22 = A22 - L21 L21^T,  W21 = -W22 (L21 W11),   W = L^-1.
// Results move from one MFMA to the next as accumulator registers wherever the contraction index of
// the next product can be taken in the order the accumulator already has (k-slot = MM::row).
// A lone wave pays ~4 ns per instruction and ~30 ns per LDS round trip, and a predicated LDS or global
// access costs a branch: every access below is unconditional (lanes 32..63 mirror lanes 0..31 and
// store the same values, the never-read upper triangles take whatever falls out) and selects do the
// masking.
//   Sh   LDS, DIAG32_LDS scalars: Sh[c * 33 + r] = block(r, c) for r >= c, ZERO for r < c, a partial block (nb < 32)
//        already padded with an identity by the caller; the second half is scratch for W
//   out: F block (in place, zeros above the diagonal) and Wt[j * 32 + c] = W(c, j)
// The caller's part of the contract (all unconditional stores of one wave, before the barrier that precedes
// diag32_factor_invert): the image has ZEROS strictly above the diagonal (sh_image_from_acc / the loaders do
// that in the select they need anyway for the identity padding), a 16 x 16 identity sits behind the two
// images and the never-written (0,1) tile of the W image is zero -- so the sweeps stream block rows and
// identity rows through one per-lane (base, stride) pair and the results leave without a single select
// (an fp64 select is two v_cndmask: the selects used to be ~260 of the ~2100 instructions of this chain).
constexpr int DIAG32_LDS = 2 * 32 * 33 + 256;   // block image | W image | identity
template <typename T> __device__ __forceinline__ void diag32_init_tables(T *Sh) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e = lane + 64 * u;
    Sh[2 * 32 * 33 + e] = (e >> 4) == (e & 15) ? (T)1 : (T)0;          // ident[c * 16 + q]
    Sh[32 * 33 + (16 + (e >> 4)) * 33 + (e & 15)] = (T)0;               // W image, columns 16.., rows 0..15
  }
}
// a 32 x 32 diagonal block held as 2 x 2 accumulator tiles (transposed layout: tile rows = block columns)
// -> the image, identity padding past nbn
template <typename T>
__device__ __forceinline__ void sh_image_from_acc(T *Sh, const typename Mfma16<T>::Acc (&q)[2][2], int nbn) {
  using MM = Mfma16<T>;
  const int lane = threadIdx.x & 63, li = lane & 15;
#pragma unroll
  for (int ib = 0; ib < 2; ib++)
#pragma unroll
    for (int jb = 0; jb < 2; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int jc = 16 * jb + MM::row(lane, r), ir = 16 * ib + li;
        if (jb > ib) Sh[jc * 33 + ir] = (T)0;
        else Sh[jc * 33 + ir] = (ir < nbn && jc < nbn && ir >= jc) ? q[ib][jb][r] : ((ir == jc && ir >= nbn) ? (T)1 : (T)0);
      }
}

// Mid: called once by every lane after the first 16 x 16 sweep (~1 us into the ~2.7 us): k_big_flow's chain wave publishes the X
// block it stored just before the call there -- by then those stores have landed, and the factorisation has not waited for them.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <typename T, bool WG_IS_ONE_WAVE = true, bool SC1 = false, typename Mid = NoMid>
__device__ __forceinline__ void diag32_factor_invert(T *Sh, int nb, T *Fblk, int M, T *Wt, int *err, bool store_l = true,
                                                     bool store_w = true, const Mid mid = Mid{}) {
  // orders this wave's LDS writes before its later reads: a workgroup barrier where the workgroup IS the
  // wave, a wave-level fence where other waves of the workgroup have already left
  auto sync = [] {
    if (WG_IS_ONE_WAVE) lds_barrier();   // LDS traffic only: __syncthreads() would also wait for the caller's global stores (~1 us on the chain)
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  };
  using MM = Mfma16<T>;
  constexpr int WOFF = 32 * 33;
  T *Dl = Sh, *Wl = Sh + WOFF;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const int ll = lane & 31, q = lane & 15;
  const bool rowlane = ll < 16;
  T x[16];
  bool bad;
  const T *ident = Sh + 2 * WOFF;
  const int sstride = rowlane ? 33 : 16;
  // ---- (1,1)
  {
    const T *src = rowlane ? Dl + q : ident + q;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = src[c * sstride];
  }
  bad = chol16_dpp2<T>(x, min(nb, 16));   // (leaves garbage above the diagonal of L: zeroed on the way out, the images below rely on it)
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = (rowlane && c > q) ? (T)0 : x[c];
#pragma unroll
  for (int c = 0; c < 16; c++) Sh[rowlane ? c * 33 + q : WOFF + q * 33 + c] = x[c];   // L11(q, c) | W11(c, q)
  sync();
  mid();
  // ---- L21(i, c) = sum_j A21(i, j) W11(c, j): tile rows = c, tile columns = i
  typename MM::Acc l21 = {0, 0, 0, 0};
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++)
    l21 = MM::mma(Wl[(4 * s4 + lk) * 33 + li], Dl[(4 * s4 + lk) * 33 + 16 + li], l21);
  // ---- S22(i, j) = A22(i, j) - sum_c L21(i, c) L21(j, c): tile rows = j, tile columns = i
  typename MM::Acc s22;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = MM::row(lane, r);
    s22[r] = Dl[(16 + j) * 33 + 16 + li];   // zero above the diagonal (contract)
  }
#pragma unroll
  for (int r = 0; r < 4; r++) s22 = MM::mma(-l21[r], l21[r], s22);
  // ---- T1(i, j) = sum_c L21(i, c) W11(c, j): tile rows = i, tile columns = j
  typename MM::Acc t1 = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; r++) t1 = MM::mma(l21[r], Wl[li * 33 + MM::row(lane, r)], t1);
  sync();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int c = MM::row(lane, r);
    Dl[c * 33 + 16 + li] = l21[r];
    Dl[(16 + c) * 33 + 16 + li] = li >= c ? s22[r] : (T)0;   // the product filled the upper triangle too
  }
  sync();
  // ---- (2,2)
  {
    const T *src = rowlane ? Dl + 16 * 33 + 16 + q : ident + q;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = src[c * sstride];
  }
  bad = chol16_dpp2<T>(x, max(nb - 16, 0)) || bad;
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = (rowlane && c > q) ? (T)0 : x[c];
  if (bad && lane == 0) atomicOr(err, DEVERR_NOT_SPD);
#pragma unroll
  for (int c = 0; c < 16; c++) Sh[rowlane ? (16 + c) * 33 + 16 + q : WOFF + (16 + q) * 33 + 16 + c] = x[c];
  sync();
  // ---- W21(p, j) = -sum_i W22(p, i) T1(i, j): tile rows = p, tile columns = j
  typename MM::Acc w21 = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; r++) w21 = MM::mma(-Wl[(16 + MM::row(lane, r)) * 33 + 16 + li], t1[r], w21);
#pragma unroll
  for (int r = 0; r < 4; r++) Wl[li * 33 + 16 + MM::row(lane, r)] = w21[r];
  sync();
  T lo[16], wo[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {   // e = j * 32 + c'; the strictly upper parts are zero
    const int e = t * 64 + lane, c = e >> 5, r = e & 31;
    lo[t] = Dl[c * 33 + r];   // both images are exactly zero above the diagonal by now
    wo[t] = Wl[c * 33 + r];
  }
  if (store_w) {
#pragma unroll
    for (int t = 0; t < 16; t++) mem_st<SC1>(Wt + t * 64 + lane, wo[t]);
  }
  if (!store_l) return;
  if (nb == BIG_NB) {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 64 + lane, c = e >> 5, r = e & 31;
      mem_st<SC1>(Fblk + (int64_t)c * M + r, lo[t]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 64 + lane, c = e >> 5, r = e & 31;
      if (r < nb && c <= r) mem_st<SC1>(Fblk + (int64_t)c * M + r, lo[t]);
    }
  }
}

// Rows below the 32-column block at kb, one wave per 32 rows, everything on the matrix cores:
//   A  = F[rows, kb:kb+nb] - F[rows, K0:kb] * F[kb:kb+nb, K0:kb]^T     (left-looking update, K <= 96)
//   X  = A * W^T,  W = inverse of the diagonal block (left behind by the previous launch)
// The accumulators hold the transposed tile (MFMA rows = panel columns j, MFMA columns = rows i, the
// contiguous direction).  A k-slot of the second product is whatever panel column the accumulator
// register already holds, so A goes from result to operand without leaving its registers.
// The wave that owns rows kb+32..kb+63 then forms the NEXT diagonal block the same way (its own X is
// the last 32 columns of that update), factors and inverts it: one launch per 32 columns on the chain.
// Loads go to clamped addresses (no branch per load); rows past the front only feed results that are
// never stored.
template <typename T> __global__ void __launch_bounds__(64) k_big_panel32(FactorArgs<T> a, int kb, int K0, int first) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  static_assert(BIG_NB == 32, "the left-looking panel kernels are written for 32-column blocks");
  using MM = Mfma16<T>;
  __shared__ T Sh[DIAG32_LDS];
  RRPGO_TRACE_MARK(a, 200);
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.y];
  if (kb >= m.nc) return;
  const int nb = min(BIG_NB, m.nc - kb);
  const int M = m.nc + m.nr + 1;
  const int R0 = kb + nb + blockIdx.x * 32;
  if (R0 >= M) return;
  T *F = a.lvals + m.loff;
  // every 32-column block has its own W slot (kept for the back substitution), so the slot this launch
  // reads is never the one its first workgroup writes for the next block
  const T *Wt = a.winv + (int64_t)m.wblk * 256 + (kb / BIG_NB) * 1024;
  const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const int super_end = min(K0 + BIG_SUPER, m.nc);
  const int kn = kb + BIG_NB;
  const bool look = blockIdx.x == 0 && kn < super_end;   // nb == 32 here
  RRPGO_PHASE_MARK(a, look, 500);
  // W operand tiles (cb, jb) = (0,0), (1,0), (1,1); (0,1) is zero
  T wv[3][4];
  if (!first) {
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cb = t == 0 ? 0 : 1, jb = t == 2 ? 1 : 0;
        wv[t][r] = Wt[(16 * jb + MM::row(lane, r)) * 32 + 16 * cb + li];
      }
  }
  // first: the very first block of a level.  No trailing update came before it to leave W behind, and a
  // launch of its own for it would sit alone on the chain: every workgroup of this launch factors and inverts
  // the block for itself instead (same arithmetic, same bits), the first one keeps W for the solve.  The block
  // itself stays as assembled in F -- nothing reads a diagonal block of L once its W exists, and storing it
  // here would race with the other workgroups that are still reading it.
  T dv[16];
  if (first) {
    const T *Fblk = F + (int64_t)kb * M + kb;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 64 + lane, c = e >> 5, r = e & 31;
      dv[t] = Fblk[(int64_t)min(c, nb - 1) * M + min(r, nb - 1)];
    }
  }
  int irow[2];   // this lane's two rows, clamped into the front
  irow[0] = min(R0 + li, M - 1);
  irow[1] = min(R0 + 16 + li, M - 1);
  // Operands of the left-looking update, columns K0..kb in whole 32-column blocks (at most three): their
  // loads are requested FIRST, together with everything else this wave reads, so that the launch pays one
  // memory round trip instead of one per block (measured: 3.9 + 4.7 us of a 15.7 us chain step were
  // waits on two to four serial round trips).  fp64 keeps two blocks in registers and refills.
  constexpr int PRE = sizeof(T) == 4 ? 3 : 2;
  const int nblk = (kb - K0) / BIG_NB;
  const int arow0 = kb + min(li, nb - 1), arow1 = kb + min(16 + li, nb - 1);
  const T am0 = li < nb ? (T)-1 : (T)0, am1 = 16 + li < nb ? (T)-1 : (T)0;   // sign and mask of the a-operand
  T av[PRE][8][2], bv[PRE][8][2];
  // 32-bit byte offsets from the (uniform) front base: one VALU add per load instead of 64-bit pointer
  // arithmetic (a front is at most a few hundred MB)
  const char *Fb = reinterpret_cast<const char *>(F);
  const uint32_t colb = (uint32_t)((K0 + lk) * M) * (uint32_t)sizeof(T);
  const uint32_t oa0 = colb + (uint32_t)arow0 * (uint32_t)sizeof(T), oa1 = colb + (uint32_t)arow1 * (uint32_t)sizeof(T);
  const uint32_t ob0 = colb + (uint32_t)irow[0] * (uint32_t)sizeof(T), ob1 = colb + (uint32_t)irow[1] * (uint32_t)sizeof(T);
  const uint32_t kstep = (uint32_t)(4 * M) * (uint32_t)sizeof(T);
  auto ld = [&](uint32_t off) { return *reinterpret_cast<const T *>(Fb + off); };
  auto fetch = [&](int blk, T (*xa)[2], T (*xb)[2]) {
    uint32_t d = (uint32_t)(blk * 8) * kstep;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      xa[s4][0] = ld(oa0 + d) * am0;
      xa[s4][1] = ld(oa1 + d) * am1;
      xb[s4][0] = ld(ob0 + d);
      xb[s4][1] = ld(ob1 + d);
      d += kstep;
    }
  };
  // Request order = the order the results are needed in: a wave has at most 63 vector loads in flight, so
  // the ~140 loads of a chain step return in two or three memory round trips (HBM: the data was written by
  // an earlier launch), and the left-looking MFMAs of block 0 can run under the later ones -- the tile
  // itself first, then the operand blocks, the next diagonal block last.
  typename MM::Acc acc[2][2], nxt[2][2];
#pragma unroll
  for (int jb = 0; jb < 2; jb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int j = 16 * jb + MM::row(lane, r);
      const T *ccol = F + (int64_t)(kb + min(j, nb - 1)) * M;
#pragma unroll
      for (int ib = 0; ib < 2; ib++) acc[ib][jb][r] = ccol[irow[ib]];   // masked below, once it is here
    }
#pragma unroll
  for (int p = 0; p < PRE; p++)
    if (p < nblk) fetch(p, av[p], bv[p]);
  if (look) {
    // next diagonal block (rows = columns = kn..kn+31); entries above its diagonal are never read
#pragma unroll
    for (int jb = 0; jb < 2; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const T *ccol = F + (int64_t)min(kn + 16 * jb + MM::row(lane, r), M - 1) * M;
#pragma unroll
        for (int ib = 0; ib < 2; ib++) nxt[ib][jb][r] = ccol[irow[ib]];
      }
  }
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
  if (first) {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int e = t * 64 + lane, c = e >> 5, r = e & 31;
      Sh[c * 33 + r] = (r < nb && c < nb && r >= c) ? dv[t] : ((r == c && r >= nb) ? (T)1 : (T)0);
    }
    diag32_init_tables<T>(Sh);
    lds_barrier();
    diag32_factor_invert<T>(Sh, nb, F + (int64_t)kb * M + kb, M, const_cast<T *>(Wt), a.err, false, blockIdx.x == 0);
    lds_barrier();
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cb = t == 0 ? 0 : 1, jb = t == 2 ? 1 : 0;
        wv[t][r] = Sh[32 * 33 + (16 * jb + MM::row(lane, r)) * 33 + 16 * cb + li];   // the LDS image of W
      }
    lds_barrier();   // Sh is free again for the next diagonal block's image
  }
  RRPGO_PHASE_MARK(a, look, 501);
#pragma unroll
  for (int blk = 0; blk < BIG_SUPER / BIG_NB - 1; blk++) {
    if (blk < nblk) {
      constexpr int dummy = 0; (void)dummy;
      const int slot = blk % PRE;   // compile-time after unrolling
#pragma unroll
      for (int s4 = 0; s4 < 8; s4++) {
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
          for (int jb = 0; jb < 2; jb++) acc[ib][jb] = MM::mma(av[slot][s4][jb], bv[slot][s4][ib], acc[ib][jb]);
        if (look) {
#pragma unroll
          for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int jb = 0; jb <= ib; jb++) nxt[ib][jb] = MM::mma(-bv[slot][s4][jb], bv[slot][s4][ib], nxt[ib][jb]);
        }
      }
      if (PRE < BIG_SUPER / BIG_NB - 1 && blk + PRE < nblk) fetch(blk + PRE, av[slot], bv[slot]);   // refill the freed set
    }
  }
  RRPGO_PHASE_MARK(a, look, 502);
  // X = A * W^T : out[ib][cb] (rows c of the MFMA tile) = sum_j W[c][j] * A[j][i]
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
        T *ccol = F + (int64_t)(kb + 16 * cb + MM::row(lane, r)) * M + R0 + li;
        ccol[0] = out[0][cb][r];
        ccol[16] = out[1][cb][r];
      }
  } else {
#pragma unroll
    for (int ib = 0; ib < 2; ib++)
#pragma unroll
      for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int c = 16 * cb + MM::row(lane, r), i = R0 + 16 * ib + li;
          if (i < M && c < nb) F[(int64_t)(kb + c) * M + i] = out[ib][cb][r];
        }
  }
  if (!look) return;
  RRPGO_PHASE_MARK(a, look, 503);
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
  lds_barrier();   // not __syncthreads(): the X stores above need not have landed before the next block is factored
  RRPGO_PHASE_MARK(a, look, 504);
  diag32_factor_invert<T>(Sh, nbn, F + (int64_t)kn * M + kn, M, a.winv + (int64_t)m.wblk * 256 + (kn / BIG_NB) * 1024, a.err);
  RRPGO_PHASE_MARK(a, look, 505);
}

#ifndef RRPGO_UPD_WAVES
#define RRPGO_UPD_WAVES 5   // fp32: workgroups per CU the register allocation is told to allow.  The kernels need 80 - 84 VGPRs either way (six
#endif                      // per CU); the bound only steers the scheduler: 3 (r02) lattice 4.63 ms, 4 / 5 / 6: 4.61 / 4.60 / 4.60, 8: 4.62 (r03)
// Rank update  C(i,j) -= sum_{k in [ka,ke)} F(i,k) F(j,k)  over i in [t0, M), j in [t0, jmax), i >= j, for one tile of
// (32 NT) x (32 NT) per workgroup, (16 NT) x (16 NT) per wave as NT x NT MFMA 16x16x4 tiles.  The two operand strips
// are staged through LDS in k-chunks of KC columns, double buffered: the global loads of chunk c + 1 are in flight
// while the MFMAs of chunk c run; one barrier per chunk.  NT = 2 (64 x 64 tiles, 20 KB of LDS in fp32) is what runs;
// NT = 4 (128 x 128) lost every measurement since r02 (two workgroups per CU) and is only kept instantiable.
#ifndef RRPGO_UPD_KC
#define RRPGO_UPD_KC 16     // 32: lattice 4.70 against 4.65 ms (r03)
#endif
template <typename T, int NT> struct UpdTile {
  static constexpr int KC = RRPGO_UPD_KC;   // k-chunk staged in LDS
  static constexpr int TILE = 32 * NT, WTILE = 16 * NT;  // workgroup tile, wave tile
  static constexpr int KSTEP = 256 / TILE;               // k-rows of a chunk loaded per pass of the 256 threads
  static constexpr int LDT = TILE + 16;                  // padded row: the four k-rows a wave reads hit disjoint banks
  static constexpr int NLD = KC / KSTEP;                 // global loads per operand per thread per chunk
  static constexpr int SMEM = 2 * 2 * KC * LDT;          // scalars of LDS: As[2][KC][LDT] | Bs[2][KC][LDT]
};
// One tile of a rank update by the 256 threads of a workgroup: C(I0.., J0..) -= F(I0.., ka:ke) F(J0.., ka:ke)^T,
// restricted to i < M, j < jmax, i >= j.  smem: UpdTile::SMEM scalars.  Returns whether this wave holds part of the
// tile; acc keeps the wave's (updated) part of C for a caller that goes on with it.
// DEPTH = k-chunks whose global loads are in flight ahead of the MFMAs.  1: the chunk after the current one -- the
// throughput shape (few registers, three workgroups per CU).  8: the whole K = 128 strip is requested up front -- for
// launches of a few hundred tiles at the top of the tree, where a tile is a chain of memory round trips (data of the
// previous launch comes back from HBM in ~1.3 us) and nothing else runs on the CU to hide them.
// Where the C tile of a front's FIRST trailing update comes from when its columns were left out of k_big_build
// (n > 0): the children's update matrices, gathered through the inverse maps -- the trailing part of a front is
// then written once, fully formed, instead of built (written), read, updated and written again.
template <typename T> struct TileGather {
  const ChildMeta *cm;
  int n;                       // children to gather from (a front without children starts from zeros); -1 = load the tile from F
  const int32_t *scat;
  const T *lvals, *uvals, *xch;
};
// GSC1: the gathered children are fronts of the SAME launch (k_big_flow's cross-level form): their entries are read past L1.
template <typename T, int NT, int DEPTH = 1, bool SC1 = false, bool LONGK = false, bool GSC1 = false>
__device__ __forceinline__ bool big_update_tile(T *F, int M, int ka, int ke, int jmax, int I0, int J0, T *smem,
                                                typename Mfma16<T>::Acc (&acc)[NT][NT], unsigned long long *trace = nullptr, bool pm = false,
                                                const TileGather<T> gather = TileGather<T>{nullptr, -1, nullptr, nullptr, nullptr, nullptr},
                                                int tid_in = -1 /* k_big_flow: the caller's own copy of threadIdx.x */) {
  struct { unsigned long long *trace; } a{trace};   // for RRPGO_PHASE_MARK (diagnostic builds)
  (void)a; (void)pm;
  using MM = Mfma16<T>;
  using UT = UpdTile<T, NT>;
  constexpr int KC = UT::KC, TILE = UT::TILE, WTILE = UT::WTILE, KSTEP = UT::KSTEP, LDT = UT::LDT, NLD = UT::NLD;
  T (*As)[KC][LDT] = reinterpret_cast<T (*)[KC][LDT]>(smem);                    // As[buf][k][i] =  F(I0 + i, k)
  T (*Bs)[KC][LDT] = reinterpret_cast<T (*)[KC][LDT]>(smem + 2 * KC * LDT);     // Bs[buf][k][j] = -F(J0 + j, k)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, wave = tid_in >= 0 ? __builtin_amdgcn_readfirstlane(tid_in >> 6) : wave_index(), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int wi = (wave & 1) * WTILE, wj = (wave >> 1) * WTILE;
  const int i0 = I0 + wi, j0 = J0 + wj;
  const bool wave_active = i0 < M && j0 < jmax && i0 + WTILE > j0;
  // A tile strictly below the diagonal and inside the front needs no masks at all.  Elsewhere every
  // load goes to a CLAMPED address (no branch per load: out-of-range rows only feed entries that are
  // never stored) and only the stores are predicated.
  const bool interior = I0 + TILE <= M && J0 + TILE <= jmax && I0 >= J0 + TILE;
  // ---- operand staging: thread t loads row (t % TILE) of every KSTEP-th k of the chunk
  // 32-bit byte offsets from the (uniform) front base: one VALU add per load instead of 64-bit pointer arithmetic and
  // selects (a front is a few tens of MB).  The PMC instruction mix of this kernel was 6 VALU + 3 SALU per MFMA.
  const int sr = tid % TILE, sk = tid / TILE;
  const char *Fb = reinterpret_cast<const char *>(F);
  const uint32_t oa = (uint32_t)((ka + sk) * M + min(I0 + sr, M - 1)) * (uint32_t)sizeof(T);
  const uint32_t ob = (uint32_t)((ka + sk) * M + min(J0 + sr, M - 1)) * (uint32_t)sizeof(T);
  const uint32_t qstep = (uint32_t)(KSTEP * M) * (uint32_t)sizeof(T), cstep = (uint32_t)(KC * M) * (uint32_t)sizeof(T);
  [[maybe_unused]] const Sc1Buf<T> fbuf(F, SC1 ? (uint32_t)((int64_t)M * M * (int64_t)sizeof(T)) : 0u);
  auto ldo = [&](uint32_t off) {
    if constexpr (SC1) return fbuf.ld(off);
    else return *reinterpret_cast<const T *>(Fb + off);
  };
  auto sto = [&](uint32_t off, T v) {
    if constexpr (SC1) fbuf.st(off, v);
    else *reinterpret_cast<T *>(const_cast<char *>(Fb) + off) = v;
  };
  const int nk = ke - ka;
  T ra[DEPTH][NLD], rb[DEPTH][NLD];
  auto fetch = [&](int c, T (&xa)[NLD], T (&xb)[NLD]) {
    const uint32_t ca = oa + (uint32_t)c * cstep, cb = ob + (uint32_t)c * cstep;
    if ((c + 1) * KC <= nk) {   // uniform: a whole chunk, no masks
#pragma unroll
      for (int q = 0; q < NLD; q++) {
        xa[q] = ldo(ca + (uint32_t)q * qstep);
        xb[q] = -ldo(cb + (uint32_t)q * qstep);
      }
    } else {   // columns past ke are re-read from a valid column and zeroed by a select
#pragma unroll
      for (int q = 0; q < NLD; q++) {
        const bool kok = c * KC + sk + KSTEP * q < nk;
        const uint32_t off = kok ? (uint32_t)q * qstep : 0u;
        const T va = ldo(ca + off), vb = ldo(cb + off);
        xa[q] = kok ? va : (T)0;
        xb[q] = kok ? -vb : (T)0;
      }
    }
  };
  auto stash = [&](int buf, const T (&xa)[NLD], const T (&xb)[NLD]) {
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      As[buf][sk + KSTEP * q][sr] = xa[q];
      Bs[buf][sk + KSTEP * q][sr] = xb[q];
    }
  };
  const int nchunks = (nk + KC - 1) / KC;
  constexpr int MAXCH = BIG_SUPER / KC;   // K <= 128
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    if (d < nchunks) fetch(d, ra[d], rb[d]);   // requested BEFORE the C tile: a gathered tile pays two dependent round trips of its own
  // ---- accumulators = current C tile
  if (gather.n >= 0) {   // uniform over the workgroup
    // the sum, in child order, of what the children hold for every entry (rows and columns the child does not have
    // map to -1; the maps are monotone, so row >= column implies iq >= jq wherever both exist).  The index vectors of
    // up to four children are staged in LDS by all threads (one coalesced load each), so a tile pays two dependent
    // global round trips (indices, values) however many children it has.
#pragma unroll
    for (int ib = 0; ib < NT; ib++)
#pragma unroll
      for (int jb = 0; jb < NT; jb++) acc[ib][jb] = typename MM::Acc{0, 0, 0, 0};
    // staged per child: TILE row indices | TILE column indices | TILE column OFFSETS (the column's start in the child's
    // packed or square storage: two integer multiplies, done once per column here instead of by every lane)
    int32_t *sidx = reinterpret_cast<int32_t *>(smem);
    for (int q0 = 0; q0 < gather.n; q0 += 4) {
      const int nb = min(4, gather.n - q0);
      if (q0 > 0) __syncthreads();
      for (int e = tid; e < nb * 2 * TILE; e += 256) {
        const int qq = e / (2 * TILE), w = e - qq * 2 * TILE;
        const ChildMeta c = gather.cm[q0 + qq];
        const int p = w < TILE ? min(I0 + w, M - 1) : min(J0 + w - TILE, jmax - 1);
        const int v = (gather.scat + c.scat_ptr)[p];
        sidx[qq * 3 * TILE + w] = v;
        if (w >= TILE) {
          const int jc = max(v, 0);
          sidx[qq * 3 * TILE + TILE + w] = c.uld > 0 ? jc * c.uld : jc * c.ncu - ((jc * (jc - 1)) >> 1) - jc;
        }
      }
      __syncthreads();
      if (wave_active) {
        for (int qq = 0; qq < nb; qq++) {
          const ChildMeta c = gather.cm[q0 + qq];
          const int32_t *si = sidx + qq * 3 * TILE;
          const T *Uc = (c.uld > 0 ? gather.lvals : c.uld < 0 ? gather.xch : gather.uvals) + c.uoff;
          int iq[NT];
#pragma unroll
          for (int ib = 0; ib < NT; ib++) iq[ib] = si[wi + 16 * ib + li];
#pragma unroll
          for (int jb = 0; jb < NT; jb++) {
            int jq[4], co[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              jq[r] = si[TILE + wj + 16 * jb + MM::row(lane, r)];
              co[r] = si[2 * TILE + wj + 16 * jb + MM::row(lane, r)];
            }
            T v[4][NT];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int jc = max(jq[r], 0);
#pragma unroll
              for (int ib = 0; ib < NT; ib++) v[r][ib] = mem_ld<GSC1>(Uc + co[r] + max(iq[ib], jc));   // clamped into the column
            }
            // an entry of the lower triangle exists in the child iff its row AND its column do (monotone maps);
            // entries above the diagonal of a diagonal tile take whatever falls out, they are never stored.  The
            // (rhs, rhs) corner is read like any other entry: it is finite (zero in an LDS child's packed storage,
            // which is never written there) and never used.
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int ib = 0; ib < NT; ib++) acc[ib][jb][r] += (iq[ib] | jq[r]) >= 0 ? v[r][ib] : (T)0;
          }
        }
      }
    }
    __syncthreads();   // the index vectors share the operand staging buffers
  } else if (wave_active && interior) {
#pragma unroll
    for (int jb = 0; jb < NT; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t co = (uint32_t)((j0 + 16 * jb + MM::row(lane, r)) * M + i0 + li) * (uint32_t)sizeof(T);
#pragma unroll
        for (int ib = 0; ib < NT; ib++) acc[ib][jb][r] = ldo(co + (uint32_t)(16 * ib) * (uint32_t)sizeof(T));
      }
  } else if (wave_active) {
#pragma unroll
    for (int jb = 0; jb < NT; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if constexpr (SC1) {
          const uint32_t cj = (uint32_t)(min(j0 + 16 * jb + MM::row(lane, r), jmax - 1) * M);
#pragma unroll
          for (int ib = 0; ib < NT; ib++) acc[ib][jb][r] = ldo((cj + (uint32_t)min(i0 + 16 * ib + li, M - 1)) * (uint32_t)sizeof(T));
        } else {
          const T *ccol = F + (int64_t)min(j0 + 16 * jb + MM::row(lane, r), jmax - 1) * M;
#pragma unroll
          for (int ib = 0; ib < NT; ib++) acc[ib][jb][r] = ccol[min(i0 + 16 * ib + li, M - 1)];
        }
      }
  }
  RRPGO_PHASE_MARK(a, pm, 604);
  // The chunk loop twice: FULL = a whole super-panel (K = 128, every launch above the lowest levels) with all of its
  // conditions resolved at compile time -- the general form pays eight scalar compare-and-branch pairs per chunk
  // (SQ_ACTIVE_INST_SCA read 20-27 % on this kernel).
  auto chunks = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    auto fetch_c = [&](int c, T (&xa)[NLD], T (&xb)[NLD]) {
      if constexpr (FULL) {
        const uint32_t ca = oa + (uint32_t)c * cstep, cb = ob + (uint32_t)c * cstep;
#pragma unroll
        for (int q = 0; q < NLD; q++) {
          xa[q] = ldo(ca + (uint32_t)q * qstep);
          xb[q] = -ldo(cb + (uint32_t)q * qstep);
        }
      } else {
        fetch(c, xa, xb);
      }
    };
    stash(0, ra[0], rb[0]);
    if (FULL ? DEPTH < MAXCH : DEPTH < nchunks) fetch_c(DEPTH, ra[0], rb[0]);
    __syncthreads();
    RRPGO_PHASE_MARK(a, pm, 601);
#pragma unroll
    for (int c = 0; c < MAXCH; c++) {
      if (!FULL && c >= nchunks) break;
      const int buf = c & 1;
      if (wave_active) {
#pragma unroll
        for (int s4 = 0; s4 < KC / 4; s4++) {
          T av[NT], bv[NT];
#pragma unroll
          for (int q = 0; q < NT; q++) {
            bv[q] = As[buf][4 * s4 + lk][wi + 16 * q + li];
            av[q] = Bs[buf][4 * s4 + lk][wj + 16 * q + li];
          }
#pragma unroll
          for (int ib = 0; ib < NT; ib++)
#pragma unroll
            for (int jb = 0; jb < NT; jb++) acc[ib][jb] = MM::mma(av[jb], bv[ib], acc[ib][jb]);
        }
      }
      if (FULL ? c + 1 < MAXCH : c + 1 < nchunks) {
        const int set = (c + 1) % DEPTH;   // compile-time after unrolling
        stash(buf ^ 1, ra[set], rb[set]);
        if (FULL ? c + 1 + DEPTH < MAXCH : c + 1 + DEPTH < nchunks) fetch_c(c + 1 + DEPTH, ra[set], rb[set]);
      }
      __syncthreads();
    }
  };
  // LONGK (k_big_schur: the Schur complement of a front in ONE pass over all its pivot columns, K = nc): the same chunks
  // in the same order as the super-panel by super-panel passes -- 128 is a multiple of the chunk -- as one running
  // double-buffered pipeline of any length; the tile is loaded (or gathered) once and stored once.
  auto chunks_long = [&] {
    static_assert(!LONGK || DEPTH == 1, "the long pipeline keeps one chunk in flight");
    stash(0, ra[0], rb[0]);
    if (1 < nchunks) fetch(1, ra[0], rb[0]);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
      const int buf = c & 1;
      if (wave_active) {
#pragma unroll
        for (int s4 = 0; s4 < KC / 4; s4++) {
          T av[NT], bv[NT];
#pragma unroll
          for (int q = 0; q < NT; q++) {
            bv[q] = As[buf][4 * s4 + lk][wi + 16 * q + li];
            av[q] = Bs[buf][4 * s4 + lk][wj + 16 * q + li];
          }
#pragma unroll
          for (int ib = 0; ib < NT; ib++)
#pragma unroll
            for (int jb = 0; jb < NT; jb++) acc[ib][jb] = MM::mma(av[jb], bv[ib], acc[ib][jb]);
        }
      }
      if (c + 1 < nchunks) {
        stash(buf ^ 1, ra[0], rb[0]);
        if (c + 2 < nchunks) fetch(c + 2, ra[0], rb[0]);
      }
      __syncthreads();
    }
  };
  if constexpr (LONGK) chunks_long();
  else if (nk == BIG_SUPER) chunks(std::true_type{});
  else chunks(std::false_type{});
  RRPGO_PHASE_MARK(a, pm, 602);
  if (!wave_active) return false;
  if (interior) {
#pragma unroll
    for (int jb = 0; jb < NT; jb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t co = (uint32_t)((j0 + 16 * jb + MM::row(lane, r)) * M + i0 + li) * (uint32_t)sizeof(T);
#pragma unroll
        for (int ib = 0; ib < NT; ib++) sto(co + (uint32_t)(16 * ib) * (uint32_t)sizeof(T), (T)acc[ib][jb][r]);
      }
  } else {
#pragma unroll
    for (int ib = 0; ib < NT; ib++)
#pragma unroll
      for (int jb = 0; jb < NT; jb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = i0 + 16 * ib + li, j = j0 + 16 * jb + MM::row(lane, r);
          if (i < M && j < jmax && i >= j) {
            if constexpr (SC1) sto((uint32_t)(j * M + i) * (uint32_t)sizeof(T), (T)acc[ib][jb][r]);
            else F[(int64_t)j * M + i] = acc[ib][jb][r];
          }
        }
  }
  return true;
}

// k_big_update: everything right of the super-panel at kb, Schur complement included (K <= 128), one 64 x 64 tile per
// workgroup; tile t = bx (bx + 1) / 2 + by of a front's lower triangle of tiles.  gather: the launch for a front's FIRST super-panel forms the
// tiles right of big_built_cols from the children instead of loading them (k_big_build was told to leave them out).
template <typename T, int NT, int DEPTH = 1> __global__ void __launch_bounds__(256, (sizeof(T) == 4 ? RRPGO_UPD_WAVES : 2)) k_big_update(FactorArgs<T> a, int kb, int gather, const int32_t *tile_map, int n_tiles, int xcd_remap, int schur_split) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  using MM = Mfma16<T>;
  using UT = UpdTile<T, NT>;
  constexpr int TILE = UT::TILE;
  __shared__ T smem[UT::SMEM];
  RRPGO_TRACE_MARK(a, 101);
  // The launch is a one-dimensional grid over the level's REAL tiles (tile_map[v] = front slot << 16 | tile of the
  // front's lower triangle, fronts one after the other, a front's tiles row by row).  Workgroups are dealt round-robin
  // over the 8 XCDs in dispatch order and every XCD has an L2 of its own: with neighbouring tiles on different XCDs the
  // two 32 KB operand strips of a tile are fetched from HBM by (almost) every tile that uses them.  xcd_remap gives
  // XCD c the c-th CONTIGUOUS eighth of the tile list instead -- whole fronts, or runs of consecutive tile rows, share
  // one L2, where a front's 128-column operand panel (<= 1 MB) stays resident; equal tile counts per XCD.
  // Placement only: the arithmetic of a tile does not change (speed, never correctness: MI355X_MICROARCH.md).
  unsigned v = blockIdx.x;
  if (xcd_remap) {
    const unsigned total = (unsigned)n_tiles, c = v & 7u, base = total >> 3, rem = total & 7u;
    v = c * base + min(c, rem) + (v >> 3);
  }
  const int packed = tile_map[v];
  const unsigned zq = (unsigned)packed >> 16;
  const int t = packed & 0xffff;
  int bx = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (bx * (bx + 1) / 2 > t) bx--;
  while ((bx + 1) * (bx + 2) / 2 <= t) bx++;
  const int by = t - bx * (bx + 1) / 2;
  const SnMeta m = a.task_meta[a.task_begin + zq];
  if (kb >= m.nc) return;
  const int M = m.nc + m.nr + 1;
  const int ke = min((kb / BIG_SUPER) * BIG_SUPER + BIG_SUPER, m.nc);
  const int t0 = ke;
  static_assert(NT == 4 || NT == 2, "tile shapes");
  const int I0 = t0 + bx * TILE, J0 = t0 + by * TILE;
  if (I0 >= M || J0 >= M) return;   // uniform over the workgroup
  // With the Schur complement left to k_big_schur (schur_split) a super-panel's update stops at the front's Schur
  // origin: whole tile columns for every super-panel but the last, whose tile grid starts at nc -- it only owes the
  // strip of columns [nc, origin) the earlier super-panels' last tile column reached into.
  const int jmax = (schur_split && ke == m.nc) ? min(big_schur_origin(m.nc, schur_split), M) : M;   // schur_split = k_big_schur's tile edge, 0 = none
  T *F = a.lvals + m.loff;
  typename MM::Acc acc[NT][NT];
  [[maybe_unused]] const bool pm = bx == 2 && by == 0 && zq == 0;
  RRPGO_PHASE_MARK(a, pm, 600);
  TileGather<T> tg{nullptr, -1, nullptr, nullptr, nullptr, nullptr};
  if (gather && J0 >= big_built_cols(m.nc, M)) tg = TileGather<T>{a.child_meta + m.child_begin, m.child_count, a.scat, a.lvals, a.uvals, a.xch};
  if (!big_update_tile<T, NT, DEPTH>(F, M, kb, ke, jmax, I0, J0, smem, acc, a.trace, pm, tg)) return;
  RRPGO_PHASE_MARK(a, pm, 603);
  // The first wave of the first tile holds the next super-panel's first diagonal block (rows = columns
  // = t0 .. t0+31) in acc[0..1][0..1]: after the tile is stored it factors and inverts that block here,
  // which saves a launch of its own at the head of the next super-panel's chain.
  if (bx == 0 && by == 0 && wave_index() == 0 && t0 < m.nc) {
    T *Sh = smem;   // the staging buffers are idle now (every wave passed the last barrier of the k loop)
    const int nbn = min(BIG_NB, m.nc - t0);
    static_assert(DIAG32_LDS <= UT::SMEM / 2, "the diagonal-block images fit the first operand strip");
    if constexpr (NT == 2) sh_image_from_acc<T>(Sh, acc, nbn);
    else {
      const typename MM::Acc corner[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
      sh_image_from_acc<T>(Sh, corner, nbn);
    }
    diag32_init_tables<T>(Sh);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    diag32_factor_invert<T, false>(Sh, nbn, F + (int64_t)t0 * M + t0, M, a.winv + (int64_t)m.wblk * 256 + (int64_t)(t0 / BIG_NB) * 1024, a.err);
  }
}

// k_big_schur: the Schur complement of every front of a level in ONE pass, K = nc: tile (bx, by) of the front's trailing
// square from its Schur origin on, C = [children's sum, gathered | what k_big_build left] - X[I, 0:nc] X[J, 0:nc]^T.  The
// tile is touched once (r02: once per 128 pivot columns -- read, updated, written), the k loop is nc / 16 chunks long
// instead of 8, and the sums are the same chunks in the same order: results are bit-identical to the pass-per-super-panel
// form.  Same grid as k_big_update: the level's real tiles, XCD c takes the c-th contiguous eighth.
template <typename T, int NT> __global__ void __launch_bounds__(256, (NT == 4 ? (sizeof(T) == 4 ? 2 : 1) : sizeof(T) == 4 ? RRPGO_UPD_WAVES : 2)) k_big_schur(FactorArgs<T> a, int gather, const int32_t *tile_map, int n_tiles, int xcd_remap) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  using MM = Mfma16<T>;
  using UT = UpdTile<T, NT>;
  constexpr int TILE = UT::TILE;
  __shared__ T smem[UT::SMEM];
  RRPGO_TRACE_MARK(a, 102);
  unsigned v = blockIdx.x;
  if (xcd_remap) {
    const unsigned total = (unsigned)n_tiles, c = v & 7u, base = total >> 3, rem = total & 7u;
    v = c * base + min(c, rem) + (v >> 3);
  }
  const int packed = tile_map[v];
  const unsigned zq = (unsigned)packed >> 16;
  const int t = packed & 0xffff;
  int bx = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (bx * (bx + 1) / 2 > t) bx--;
  while ((bx + 1) * (bx + 2) / 2 <= t) bx++;
  const int by = t - bx * (bx + 1) / 2;
  const SnMeta m = a.task_meta[a.task_begin + zq];
  const int M = m.nc + m.nr + 1;
  const int o = big_schur_origin(m.nc, TILE);
  const int I0 = o + bx * TILE, J0 = o + by * TILE;
  if (I0 >= M || J0 >= M) return;
  typename MM::Acc acc[NT][NT];
  TileGather<T> tg{nullptr, -1, nullptr, nullptr, nullptr, nullptr};
  // a tile is gathered as a whole or loaded as a whole: with 128-wide tiles only those that start right of the built columns
  if (gather && J0 >= big_built_cols(m.nc, M)) tg = TileGather<T>{a.child_meta + m.child_begin, m.child_count, a.scat, a.lvals, a.uvals, a.xch};
  big_update_tile<T, NT, 1, false, true>(a.lvals + m.loff, M, 0, m.nc, M, I0, J0, smem, acc, nullptr, false, tg);
}

// a0 / a1 += v[lane k of the 16-lane row] * op[k] over the even / odd k = K .. 15: a 16 x 16 matrix-vector product with the
// vector spread over the lanes of a row and lane m holding column m of the matrix, sixteen DPP multiply-adds in two chains
template <typename T, int K> struct BcastDot16 {
  static __device__ __forceinline__ void run(T &a0, T &a1, T v, const T (&op)[16]) {
    fmac_bcast<K>((K & 1) ? a1 : a0, v, op[K]);
    BcastDot16<T, K + 1>::run(a0, a1, v, op);
  }
};
template <typename T> struct BcastDot16<T, 16> {
  static __device__ __forceinline__ void run(T &, T &, T, const T (&)[16]) {}
};

// the sum of v over the 16 lanes of the calling lane's row by rotations within the row (8, 4, 2, 1): every lane ends with
// the sum of all 16 -- associated in an order that depends on the lane, so ONE lane's result is used (lane 0 of the row:
// ((v0 + v8) + (v4 + v12)) + ((v2 + v10) + (v6 + v14)) + the same of the odd lanes)
template <int ROR> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xf, 0xf, false));
}
template <int ROR> __device__ __forceinline__ double row_ror(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + ROR, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + ROR, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <typename T> __device__ __forceinline__ T row_sum16(T v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  return v;
}

// Back substitution for one supernode:
//   x1 = L11^-T ( y1 - L21^T x[rows] ),  y1 = the rhs row of the factored panel.
// STAGE (fronts of the LDS path): x[rows] and t (then x1) live in LDS, padded with zeros to whole 16s; L21^T x comes
// straight from global memory, one column per 16-lane row; the chain over 16-column blocks reads its operands (the kept
// inverse diagonal blocks W_b and the sub-diagonal blocks of L11) from an LDS image -- all of L11 in k_solve_flow, just
// those blocks in the level schedule -- staged in the same round trip as the L21 loads; details at the code below.
// work: STAGE: (nr -> 16s) + (nc -> 16s) + 2 + the image (pgo_api.hip, step_solve_lds_); else nr + nc + a 64 x 65 chunk.
// FLOW (k_solve_flow, lds_flow.hip.h): a front waits for its ancestors' entries of x themselves (x_wait: an entry that has
// not been produced holds a NaN payload), reads and writes x with sc1 accesses and sets no flag.  Same arithmetic in both
// forms.
template <typename T, int THREADS, bool STAGE, bool FLOW = false>
__device__ void solve_front(const FactorArgs<T> &a, int s, const SnMeta &m, T *work) {
  const int tid = threadIdx.x;
  const int nc = m.nc, nr = m.nr;
  const int M = nc + nr + 1;
  const T *Lg = a.lvals + m.loff;
  const int32_t *rows = a.sn_rows + m.rows_ptr;
  if (STAGE) {
    constexpr int NW = THREADS / 64;
    const int wave = wave_index(), lane = tid & 63, l16 = lane & 15;
    const int nblk = (nc + 15) >> 4, ncp = 16 * nblk;
    const int nrp = (nr + 15) & ~15;
    T *x2 = work;         // nr, padded with zeros to whole 16s
    T *t1 = x2 + nrp;     // 16 nblk (t, then x; zero past nc: a partial last block needs no masks), 16-byte aligned
    // The chain reads its operands from LDS, staged in the same round trip as the loads below:
    //   full image (k_solve_flow, when the launch has the room: a.solve_lds)   L11 and the W blocks: column j at Ll + j ldl,
    //     ldl odd, rows padded with zeros to whole blocks; W_b TRANSPOSED in the place of the diagonal block b, which
    //     nobody reads.  It is staged while the front waits for its parent, and the fold reads it too: a step of the chain
    //     makes no trip to memory.
    //   chain image (otherwise)   per block b, 2 x 16 rows of 17: W_b transposed, then L(b+1, b) transposed, zero past the
    //     front.  The fold's operands come from memory, requested one barrier ahead.
    T *Ll = t1 + ncp;
    const int ldl = ncp + 1;
    constexpr int CBLK = 2 * 16 * 17;
    const bool img = FLOW && nrp + ncp + ncp * ldl + 2 <= a.solve_lds;
    __syncthreads();
    RRPGO_STAMP_SOLVE(a, s, 0);
    // t = y1 - L21^T x[rows]: the 16 lanes of a row take one column, lane q the rows q, q + 16, ... (a row of lanes reads
    // 128 contiguous bytes of the column), the 16 partial sums are added with DPP broadcasts (row_sum16).  A workgroup
    // covers THREADS / 16 columns per pass; the sum of a column does not depend on the workgroup's size.  Nothing of L21
    // depends on x: when the shape allows (PM passes of RM rows per lane) the loads are requested before the front waits
    // for its parent, so that after the wait there are R reads of x from LDS and P (R + 17) multiply-adds per lane.
    constexpr int CPP = THREADS / 16, PM = 3, RM = 8;
    const int q16 = tid & 15, jl = tid >> 4;
    const int R = (nr + 15) >> 4, P = (nc + CPP - 1) / CPP;
    const bool hoist = nr > 0 && P <= PM && R <= RM;
    T lv[PM][RM], gy[PM];
    if (hoist) {
#pragma unroll
      for (int p = 0; p < PM; p++)
        if (p < P) {
          const T *col = Lg + (int64_t)min(jl + p * CPP, nc - 1) * M + nc;
#pragma unroll
          for (int r = 0; r < RM; r++)
            if (r < R) lv[p][r] = col[min(q16 + 16 * r, nr)];   // past the rows: the rhs entry, times the zero padding of x2
          gy[p] = col[nr];
        }
    }
    const int row0 = nr > 0 ? rows[min(tid, nr - 1)] : 0;   // where this thread's x comes from: known before the wait
    if (tid < ncp - nc) t1[nc + tid] = (T)0;
    const T *Wsrc = a.winv + (int64_t)m.wblk * 256;
    if (img) {
      // the W blocks are requested first and stored last: one round trip for a front of up to 10 columns per wave
      auto wstore = [&](int i, T v) {   // W_b(k, m) -> image (row 16 b + k, column 16 b + m)
        const int cb = 16 * (i >> 8), wk = (i >> 4) & 15, wm = i & 15;
        if (i < nblk * 256) Ll[(cb + wm) * ldl + cb + wk] = v;
      };
      T w4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) w4[u] = Wsrc[min(tid + u * THREADS, nblk * 256 - 1)];
      constexpr int SB = 10;   // columns a wave has in flight
      const bool two = ncp > 64;
      for (int j0 = wave; j0 < nc; j0 += NW * SB) {
        T v[SB][2];
#pragma unroll
        for (int u = 0; u < SB; u++) {
          const T *col = Lg + (int64_t)min(j0 + NW * u, nc - 1) * M;
          v[u][0] = col[min(lane, nc - 1)];
          if (two) v[u][1] = col[min(lane + 64, nc - 1)];
        }
#pragma unroll
        for (int u = 0; u < SB; u++) {
          const int j = j0 + NW * u;
          if (j < nc) {
            if (lane < ncp && (lane >> 4) > (j >> 4)) Ll[j * ldl + lane] = lane < nc ? v[u][0] : (T)0;
            if (two && lane + 64 < ncp && ((lane + 64) >> 4) > (j >> 4)) Ll[j * ldl + lane + 64] = lane + 64 < nc ? v[u][1] : (T)0;
          }
        }
      }
      for (int j = wave; j < nc; j += NW)   // fronts wider than 128 columns
        for (int i = lane + 128; i < ncp; i += 64)
          if ((i >> 4) > (j >> 4)) Ll[j * ldl + i] = i < nc ? Lg[(int64_t)j * M + i] : (T)0;
#pragma unroll
      for (int u = 0; u < 4; u++) wstore(tid + u * THREADS, w4[u]);
      for (int i = tid + 4 * THREADS; i < nblk * 256; i += THREADS) wstore(i, Wsrc[i]);
    } else {
      // entry e of the chain image: block b = e >> 9, then the 256 entries of W_b and the 256 of L(b+1, b), both in memory
      // order; lane m of the chain finds its 16 operands k = 0 .. 15 at m * 17 + k
      constexpr int SC = 8;
      const int ne = (2 * nblk - 1) * 256;
      for (int e0 = tid; e0 < ne; e0 += SC * THREADS) {
        T v[SC];
#pragma unroll
        for (int u = 0; u < SC; u++) {
          const int e = min(e0 + u * THREADS, ne - 1);
          const int b = e >> 9, hi = (e >> 4) & 15, lo = e & 15;   // L: (m, k); W: (k, m) -- both contiguous in memory
          const T *src = (e & 256) ? Lg + (int64_t)(16 * b + hi) * M + min(16 * b + 16 + lo, nc - 1) : Wsrc + b * 256 + (e & 255);
          v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < SC; u++) {
          const int e = e0 + u * THREADS;
          const int b = e >> 9, hi = (e >> 4) & 15, lo = e & 15;
          if (e < ne) {
            if (e & 256) Ll[b * CBLK + 272 + hi * 17 + lo] = 16 * b + 16 + lo >= nc ? (T)0 : v[u];
            else Ll[b * CBLK + lo * 17 + hi] = v[u];
          }
        }
      }
    }
    if constexpr (FLOW) {
      // every thread waits for its own entry of x (the parent's columns arrive last; the older ancestors' are there)
      const int pdep = a.parent_dep[s];
      if (pdep >= 2 * a.parent_dep_self) dep_wait(a.dep_flags + pdep, a.err, a.wait_ticks);   // failure injection only (rr_pgo_debug_withhold): a word nobody sets
      if (tid < nr) x2[tid] = x_wait(a.x + row0, a.err, a.wait_ticks);
      for (int i = tid + THREADS; i < nr; i += THREADS) x2[i] = x_wait(a.x + rows[i], a.err, a.wait_ticks);
    } else {
      if (tid < nr) x2[tid] = a.x[row0];
      for (int i = tid + THREADS; i < nr; i += THREADS) x2[i] = a.x[rows[i]];
    }
    if (tid < nrp - nr) x2[nr + tid] = (T)0;
    __syncthreads();
    RRPGO_STAMP_SOLVE(a, s, 1);
    if (nr == 0) {
      // a root front has no rows below its pivot block: t1 is its rhs row
      for (int j = tid; j < nc; j += THREADS) t1[j] = Lg[(int64_t)j * M + nc];
    } else if (hoist) {
      T xr[RM];
#pragma unroll
      for (int r = 0; r < RM; r++) xr[r] = x2[min(q16 + 16 * r, nrp - 1)];   // all RM reads at once; those past R are not used
#pragma unroll
      for (int p = 0; p < PM; p++) {
        if (p >= P) break;
        T acc = lv[p][0] * xr[0];
#pragma unroll
        for (int r = 1; r < RM; r++) {
          if (r >= R) break;   // wave-uniform: a branch, not a select per multiply-add
          acc += lv[p][r] * xr[r];
        }
        const T tot = row_sum16(acc);
        const int j = jl + p * CPP;
        if (q16 == 0 && j < nc) t1[j] = gy[p] - tot;
      }
    } else {
      // the same sums with the loads behind the wait: two columns per lane row and eight rows at a time in flight
      for (int j0 = 0; j0 < nc; j0 += 2 * CPP) {
        const int jA = j0 + jl, jB = jA + CPP;
        const T *colA = Lg + (int64_t)min(jA, nc - 1) * M + nc, *colB = Lg + (int64_t)min(jB, nc - 1) * M + nc;
        T accA = 0, accB = 0;
        for (int r0 = 0; r0 < R; r0 += 8) {
          T la8[8], lb8[8], x8[8];
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int i = min(q16 + 16 * (r0 + r), nr);   // past the rows: the rhs entry, times zero
            la8[r] = colA[i];
            lb8[r] = colB[i];
            x8[r] = r0 + r < R ? x2[q16 + 16 * (r0 + r)] : (T)0;
          }
#pragma unroll
          for (int r = 0; r < 8; r++) {
            if (r0 + r >= R) break;
            accA += la8[r] * x8[r];
            accB += lb8[r] * x8[r];
          }
        }
        const T totA = row_sum16(accA), totB = row_sum16(accB);
        if (q16 == 0 && jA < nc) t1[jA] = colA[nr] - totA;
        if (q16 == 0 && jB < nc) t1[jB] = colB[nr] - totB;
      }
    }
    // L11^T x = t, backward, by 16-column blocks with the inverse diagonal blocks W_b = L_bb^-1:
    //   x_b = W_b^T ( t_b - sum_{b' > b} L(b', b)^T x_b' ).
    // The first wave runs the chain: a vector lives spread over the 16 lanes of a row (the four rows of the wave
    // compute the same), lane m holds column m of the operand block, and a 16 x 16 product is sixteen DPP
    // multiply-adds (v_fmac row_newbcast, two chains of eight):
    //   u = t_b - L(b+1, b)^T x_(b+1),   la[k] = L(16 (b+1) + k, 16 b + m)
    //   x_b = W_b^T u,                   wv[k] = W_b(k, m)
    // (the matrix cores need eight DEPENDENT v_mfma per step, ~100 clocks each in fp64).  The other waves fold the
    // finished block into everything two or more blocks to its left, one barrier behind: a fold thread owns one
    // column i, requests its 16 contiguous entries L(c0.., i) BEFORE the barrier that releases the block -- no
    // address depends on x -- and reads x_b with vector loads; four partial sums keep its dependent chain short.
    // So there is one barrier per 16 columns and nobody writes an entry somebody else is reading.
    auto chain = [&](auto img_c) {
      constexpr bool IMG = decltype(img_c)::value;
      T la[16], wv[16];
      auto lblock = [&](int b, T (&dst)[16]) {   // operand block of chain step b (uses block b + 1), b clamped
        const int bc = max(min(b, nblk - 2), 0);
        if constexpr (IMG) {
          const T *col = Ll + (16 * bc + l16) * ldl + 16 * bc + 16;
#pragma unroll
          for (int k = 0; k < 16; k++) dst[k] = col[k];
        } else {
          const T *col = Ll + bc * CBLK + 272 + l16 * 17;
#pragma unroll
          for (int k = 0; k < 16; k++) dst[k] = col[k];
        }
      };
      auto wblock = [&](int b, T (&dst)[16]) {   // W operand of chain step b, b clamped
        const int bc = max(b, 0);
        if constexpr (IMG) {
          const T *col = Ll + (16 * bc + l16) * ldl + 16 * bc;
#pragma unroll
          for (int k = 0; k < 16; k++) dst[k] = col[k];
        } else {
          const T *col = Ll + bc * CBLK + l16 * 17;
#pragma unroll
          for (int k = 0; k < 16; k++) dst[k] = col[k];
        }
      };
      const bool chainer = wave == 0 && lane < 16;   // one 16-lane row carries the chain
      auto chain_preload = [&] {   // operands of the first chain step: nothing here depends on x
        if (chainer) {
          wblock(nblk - 1, wv);
          if (nblk >= 2) lblock(nblk - 2, la);
        }
      };
      __syncthreads();
      chain_preload();
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the loop below starts with nothing of its own in flight
      RRPGO_STAMP_SOLVE(a, s, 2);
      T xprev = 0;
      const int fi = NW == 1 ? tid : tid - 64;            // the fold thread's column
      constexpr int FSTRIDE = NW == 1 ? THREADS : THREADS - 64;
      using XV = T __attribute__((ext_vector_type(16 / sizeof(T))));
      constexpr int XE = 16 / sizeof(T);
      for (int b = nblk - 1; b >= 0; b--) {
        const int c0 = 16 * b, cw = min(16, nc - c0);
        const int lim = c0 - 16;
        T fl[16];
        const bool folder = (NW == 1 || wave > 0) && fi >= 0 && fi < lim;
        auto fold_operands = [&] {
          if constexpr (IMG) {
            const T *lcol = Ll + fi * ldl + c0;
#pragma unroll
            for (int j = 0; j < 16; j++) fl[j] = lcol[j];
          } else {
            const T *lcol = Lg + (int64_t)fi * M + c0;
#pragma unroll
            for (int j = 0; j < 16; j++) fl[j] = lcol[min(j, cw - 1)];   // past the block's width: meets x = 0
          }
        };
        if (NW > 1 && folder) fold_operands();   // requested before the barrier
        if (chainer) {
          T u = t1[c0 + l16];
          if (b + 1 < nblk) {
            T a0 = 0, a1 = 0;
            asm volatile("s_nop 1" : "+v"(xprev));   // VALU write -> DPP read of the same register
            BcastDot16<T, 0>::run(a0, a1, xprev, la);   // xprev is zero past the block's width
            u -= a0 + a1;
            lblock(b - 1, la);
          }
          T x0 = 0, x1 = 0;
          asm volatile("s_nop 1" : "+v"(u));
          BcastDot16<T, 0>::run(x0, x1, u, wv);
          T x = x0 + x1;
          x = l16 < cw ? x : (T)0;
          t1[c0 + lane] = x;
          xprev = x;
          wblock(b - 1, wv);
        }
        __syncthreads();
        if (NW == 1 || wave > 0) {
          if (NW == 1 && folder) fold_operands();
          if (folder) {
            const XV *xp = reinterpret_cast<const XV *>(t1 + c0);
            T p[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 16 / XE; q++) {
              const XV xv = xp[q];
#pragma unroll
              for (int e = 0; e < XE; e++) p[(q * XE + e) & 3] += fl[q * XE + e] * xv[e];
            }
            t1[fi] -= (p[0] + p[1]) + (p[2] + p[3]);
          }
          for (int i = fi + FSTRIDE; i < lim; i += FSTRIDE) {   // fronts wider than the workgroup (not on the LDS path today)
            const T *lcol = IMG ? Ll + i * ldl + c0 : Lg + (int64_t)i * M + c0;
            T tv = t1[i];
            for (int j = 0; j < cw; j++) tv -= lcol[j] * t1[c0 + j];
            t1[i] = tv;
          }
        }
      }
    };
    if (img) chain(std::true_type{});
    else chain(std::false_type{});
    __syncthreads();
    RRPGO_STAMP_SOLVE(a, s, 3);
    for (int j = tid; j < nc; j += THREADS) mem_st<FLOW>(a.x + m.col0 + j, t1[j]);   // FLOW: the entries are their own flags (x_wait)
    __syncthreads();
    RRPGO_STAMP_SOLVE(a, s, 4);
  } else {
    // Front in place in HBM (fronts beyond LDS), left-looking over 64-column chunks from the right:
    //   t_c = t_c - L[pivot rows below the chunk, chunk]^T * xf[those rows]   one contiguous GEMV
    //   x_c = L_cc^-T t_c                                                   64-step chunk solve
    // with xf = x1 (this front's pivots: t on entry, the solution as chunks finish).
    // A wave streams 4 columns, 4 row-blocks at a time (16 independent loads in flight per lane).
    T *xf = work;              // nc
    T *Lc = xf + nc;           // 64 x 65 chunk, transposed, reciprocal diagonal
    constexpr int NW = THREADS / 64;
    const int wave = wave_index(), lane = tid & 63;
    __syncthreads();
    // the product with the rows below the pivot block, y1 - L21^T x[rows], was formed by the
    // multi-workgroup k_big_gemv_* launches and left in x[col0 ..]: only L11 is streamed here
    for (int j = tid; j < nc; j += THREADS) xf[j] = a.x[m.col0 + j];
    for (int c0 = ((nc - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
      const int cw = min(64, nc - c0);
      for (int t = tid; t < cw * cw; t += THREADS) {
        const int c = t / cw, r = t - c * cw;
        if (r > c) Lc[r * 65 + c] = Lg[(int64_t)(c0 + c) * M + c0 + r];
        else if (r == c) Lc[r * 65 + c] = (T)1 / Lg[(int64_t)(c0 + c) * M + c0 + r];
      }
      __syncthreads();   // also publishes xf entries written by the previous chunk
      const int rb = c0 + cw, re = nc;
      for (int jb = 4 * wave; jb < cw; jb += 4 * NW) {
        const T *col[4];
        T acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) col[q] = Lg + (int64_t)(c0 + min(jb + q, cw - 1)) * M;
        int i = rb + lane;
        for (; i + 192 < re; i += 256) {
          T xv[4], lv[4][4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            xv[u] = xf[i + 64 * u];
#pragma unroll
            for (int q = 0; q < 4; q++) lv[q][u] = col[q][i + 64 * u];
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] += lv[q][u] * xv[u];
        }
        for (; i < re; i += 64) {
          const T xv = xf[i];
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] += col[q][i] * xv;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (jb + q < cw) xf[c0 + jb + q] -= acc[q];
        }
      }
      __syncthreads();
      if (tid < 64) {
        T tv = tid < cw ? xf[c0 + tid] : (T)0;
        const T *row = Lc + (cw - 1) * 65;
        T rd = row[cw - 1];
        T rv = tid < cw - 1 ? row[tid] : (T)0;
        for (int jj = cw - 1; jj >= 0; jj--) {
          T nd = 0, nv = 0;
          if (jj > 0) {
            const T *rn = row - 65;
            nd = rn[jj - 1];
            nv = tid < jj - 1 ? rn[tid] : (T)0;
            row = rn;
          }
          const T xj = lane_bcast(tv, jj) * rd;
          const T upd = tv - rv * xj;
          tv = tid == jj ? xj : upd;
          rd = nd;
          rv = nv;
        }
        if (tid < cw) xf[c0 + tid] = tv;
      }
      __syncthreads();   // Lc is restaged and xf[c0..] is read by everybody in the next chunk
    }
    for (int j = tid; j < nc; j += THREADS) a.x[m.col0 + j] = xf[j];
    __syncthreads();
  }
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_solve_tasks(FactorArgs<T> a) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int task = a.task_begin + blockIdx.x;
  const int sbeg = a.task_ptr[task], send = a.task_ptr[task + 1];
  int snext = a.task_sn[send - 1];   // next front's record in flight under the current front (see k_factor_tasks)
  SnMeta mnext = a.sn_meta[snext];
  for (int si = send - 1; si >= sbeg; si--) {
    const int s = snext;
    const SnMeta m = mnext;
    if (si > sbeg) {
      snext = a.task_sn[si - 1];
      mnext = a.sn_meta[snext];
    }
    solve_front<T, THREADS, true>(a, s, m, smem);
  }
}

// mid / huge fronts: one workgroup per front, panel streamed from HBM (task list of single fronts)
// Back substitution of the fronts beyond LDS, first part:  t = y1 - L21^T x[rows]  for every such front
// of a level, spread over the whole chip: workgroup (bx, by, front) takes 64 columns and the by-th of
// R row slices, lanes along the rows (coalesced), a wave 4 columns at a time.  Partial sums go to
// part[by][col0 + j] (fixed slots, summed in order by the solve kernels: deterministic).
// FLOW: a GEMV task of k_solve_flow (lds_flow.hip.h) -- the entries of x are waited for in place (x_wait), the partial sums are
// written through; waves beyond the fourth of a larger workgroup only take part in the barriers.  Same sums either way.
template <typename T, bool FLOW>
__device__ __forceinline__ void big_gemv_unit(const FactorArgs<T> &a, const SnMeta &m, T *part, int64_t N, int R, int bx, int by, T *xs /* 1024 */) {
  const int nc = m.nc, nr = m.nr, M = nc + nr + 1;
  const int j0 = bx * 64;
  const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63;
  const int nthreads = blockDim.x;
  const int i_begin = (int)((int64_t)nr * by / R), i_end = (int)((int64_t)nr * (by + 1) / R);
  const T *Lg = a.lvals + m.loff;
  const int32_t *rows = a.sn_rows + m.rows_ptr;
  T acc[4][4];   // [pass][column of the pass]
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[p][q] = 0;
  for (int ib = i_begin; ib < i_end; ib += 1024) {
    const int cnt = min(1024, i_end - ib);
    __syncthreads();
    for (int t = tid; t < cnt; t += nthreads) {
      if constexpr (FLOW) xs[t] = x_wait(a.x + rows[ib + t], a.err, a.wait_ticks);
      else xs[t] = a.x[rows[ib + t]];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int jc = j0 + 16 * wave + 4 * p;   // this wave's 4 columns of the pass (clamped: no branch per load)
      const T *c0 = Lg + (int64_t)min(jc, nc - 1) * M + nc + ib;
      const T *c1 = Lg + (int64_t)min(jc + 1, nc - 1) * M + nc + ib;
      const T *c2 = Lg + (int64_t)min(jc + 2, nc - 1) * M + nc + ib;
      const T *c3 = Lg + (int64_t)min(jc + 3, nc - 1) * M + nc + ib;
      int i = lane;
      for (; i + 192 < cnt; i += 256) {
        T xv[4], l0[4], l1[4], l2[4], l3[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          xv[u] = xs[i + 64 * u];
          l0[u] = c0[i + 64 * u]; l1[u] = c1[i + 64 * u]; l2[u] = c2[i + 64 * u]; l3[u] = c3[i + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc[p][0] += l0[u] * xv[u]; acc[p][1] += l1[u] * xv[u]; acc[p][2] += l2[u] * xv[u]; acc[p][3] += l3[u] * xv[u];
        }
      }
      for (; i < cnt; i += 64) {
        const T xv = xs[i];
        acc[p][0] += c0[i] * xv; acc[p][1] += c1[i] * xv; acc[p][2] += c2[i] * xv; acc[p][3] += c3[i] * xv;
      }
    }
    }
  }
  if (wave < 4) {
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const T sum = wave_sum63<T>(acc[p][q]);
      const int j = j0 + 16 * wave + 4 * p + q;
      if (lane == 63 && j < nc) mem_st<FLOW>(part + (int64_t)by * N + m.col0 + j, sum);
    }
  }
}
template <typename T> __global__ void __launch_bounds__(256) k_big_gemv_partial(FactorArgs<T> a, T *part, int64_t N, int R) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  __shared__ T xs[1024];
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.z];
  if ((int)blockIdx.x * 64 >= m.nc) return;
  big_gemv_unit<T, false>(a, m, part, N, R, blockIdx.x, blockIdx.y, xs);
}

template <typename T> __device__ __forceinline__ T half_wave_sum(T v) {   // sums of lanes 0..31 / 32..63 in lanes 31 / 63
  v += dpp_get<0x111, 0xf>(v);
  v += dpp_get<0x112, 0xf>(v);
  v += dpp_get<0x114, 0xf>(v);
  v += dpp_get<0x118, 0xf>(v);
  v += dpp_get<0x142, 0xa>(v);
  return v;
}
template <typename T, int THREADS, bool FLOW = false>   // FLOW: a task of k_solve_flow -- the partial sums read past L1, x written through (its entries are their own flags)
__device__ void solve_big_front(const FactorArgs<T> &a, const SnMeta &m, T *work, const T *part, int64_t N, int R) {
  constexpr int NW = THREADS / 64;
  constexpr int NQ = 32;                // column pairs per wave and pass
  const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int nc = m.nc, M = nc + m.nr + 1;
  const T *Lg = a.lvals + m.loff;
  const T *Wb = a.winv + (int64_t)m.wblk * 256;
  const int nblk = (nc + 31) >> 5;
  T *xf = work;                     // nc: t on entry, x as blocks finish
  T *Ws = work + ((nc + 3) & ~3);   // 2 x (32 x 33): W_b staged transposed, Ws[j * 33 + c] = W_b(j, c)
  __syncthreads();
  // t = y1 - L21^T x[rows]: the R row slices of k_big_gemv_partial summed here, in slice order (what a separate
  // separate launch used to do)
  for (int j = tid; j < nc; j += THREADS) {
    T t = Lg[(int64_t)j * M + (M - 1)];
    if (m.nr > 0)
      for (int r0 = 0; r0 < R; r0 += 8) {   // eight slices requested together, subtracted in slice order
        T p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = mem_ld<FLOW>(part + (int64_t)min(r0 + u, R - 1) * N + m.col0 + j);
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (r0 + u < R) t -= p[u];
      }
    xf[j] = t;
  }
  auto stage_w = [&](int b) {   // Wt[c * 32 + j] = W_b(j, c)  ->  Ws[j * 33 + c]
    for (int e = tid; e < 1024; e += THREADS) {   // (k_solve_mid: 1024 threads, one entry each; a task of k_solve_flow: 256 or 512)
      const int c = e >> 5, j = e & 31;
      Ws[(b & 1) * (32 * 33) + j * 33 + c] = Wb[(int64_t)b * 1024 + e];
    }
  };
  stage_w(nblk - 1);
  auto chain_step = [&](int b, int c0, int cw) {   // x_b = W_b^T t_b on the first wave
    if (tid < 64) {
      const T *ws = Ws + (b & 1) * (32 * 33) + l32;
      const T tv = pin(xf[c0 + min(l32, cw - 1)]);
      const T v = l32 < cw ? tv : (T)0;
      T w[32];
#pragma unroll
      for (int j = 0; j < 32; j++) w[j] = ws[j * 33];
      T x = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) x += w[j] * lane_bcast(v, j);   // W is padded with an identity past cw
      if (lane < cw) xf[c0 + lane] = x;
    }
  };
  if constexpr (sizeof(T) == 8) {
    // fp64 (sphere2500, torus3D: fronts of a few hundred columns, where the round trip per block shows).
    // The fold of block b into the columns i < 32 b reads L(32 b + l32, i): a wave takes column pairs
    // i = 2 (wave + NW q) + half, NG pairs per GROUP.  Groups run through all blocks as one stream and the next
    // group -- of this block or the first of the next -- is requested before the current one is summed (two
    // register sets), so the HBM round trip of a block's rows hides under the previous block's fold and chain
    // step instead of following every barrier.  No address depends on x.  (fp32 keeps the per-block requests
    // below: on the 1M-edge lattice the stream was measured 3-5 % slower, +2.4 % on the fp64 datasets.)
    constexpr int NG = NQ / 2;
    T lv0[NG], lv1[NG];
    auto pairs_of = [&](int b) { return ((32 * b + 1) / 2 + NW - 1) / NW; };   // column pairs per wave, uniform
    auto fetch = [&](int b, int q0, T (&dst)[NG]) {
      const int c0 = 32 * b, cw = min(32, nc - c0), npw = pairs_of(b);
      const T *base = Lg + c0 + min(l32, cw - 1);
#pragma unroll
      for (int q = 0; q < NG; q++)
        if (q0 + q < npw) dst[q] = base[(int64_t)min(2 * (wave + NW * (q0 + q)) + half, nc - 1) * M];
    };
    int cur = 0;
    if (nblk >= 2) fetch(nblk - 1, 0, lv0);
    __syncthreads();
    for (int b = nblk - 1; b >= 0; b--) {
      const int c0 = 32 * b, cw = min(32, nc - c0);
      if (b > 0) stage_w(b - 1);
      chain_step(b, c0, cw);
      __syncthreads();
      const T xv = pin(xf[c0 + min(l32, cw - 1)]);
      const T xj = l32 < cw ? xv : (T)0;
      const int npw = pairs_of(b);
      for (int q0 = 0; q0 < npw; q0 += NG) {
        // the set `cur` holds group (b, q0); the next group goes into the other set first
        const bool more = q0 + NG < npw;
        const int nb2 = more ? b : b - 1, nq2 = more ? q0 + NG : 0;
        auto body = [&](T (&use)[NG], T (&other)[NG]) {
          if (nb2 >= 1) fetch(nb2, nq2, other);
#pragma unroll
          for (int q = 0; q < NG; q++)
            if (q0 + q < npw) {
              const int i = 2 * (wave + NW * (q0 + q)) + half;
              const T sum = half_wave_sum<T>(use[q] * xj);
              if (l32 == 31 && i < c0) xf[i] -= sum;
            }
        };
        if (cur == 0) body(lv0, lv1);
        else body(lv1, lv0);
        cur ^= 1;
      }
      __syncthreads();
    }
  } else {
  __syncthreads();
  for (int b = nblk - 1; b >= 0; b--) {
    const int c0 = 32 * b, cw = min(32, nc - c0);
    if (b > 0) stage_w(b - 1);
    // L(c0 + l32, i) for this wave's columns i = 2 (wave + NW q) + half; the first NQ pairs per wave are
    // requested now, any further pass (pivot blocks wider than 2 NW NQ columns) after the block is solved
    const int npw = ((c0 + 1) / 2 + NW - 1) / NW;   // column pairs per wave, uniform
    const T *base = Lg + c0 + min(l32, cw - 1);
    T lv[NQ];
    auto fetch = [&](int q0) {
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (q0 + q < npw) lv[q] = base[(int64_t)min(2 * (wave + NW * (q0 + q)) + half, nc - 1) * M];
    };
    fetch(0);
    if (tid < 64) {
      const T *ws = Ws + (b & 1) * (32 * 33) + l32;
      const T tv = pin(xf[c0 + min(l32, cw - 1)]);
      const T v = l32 < cw ? tv : (T)0;
      T w[32];
#pragma unroll
      for (int j = 0; j < 32; j++) w[j] = ws[j * 33];
      T x = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) x += w[j] * lane_bcast(v, j);   // W is padded with an identity past cw
      if (lane < cw) xf[c0 + lane] = x;
    }
    __syncthreads();
    {
      const T xv = pin(xf[c0 + min(l32, cw - 1)]);
      const T xj = l32 < cw ? xv : (T)0;
      for (int q0 = 0; q0 < npw; q0 += NQ) {
        if (q0 > 0) fetch(q0);
#pragma unroll
        for (int q = 0; q < NQ; q++)
          if (q0 + q < npw) {
            const int i = 2 * (wave + NW * (q0 + q)) + half;
            const T sum = half_wave_sum<T>(lv[q] * xj);
            if (l32 == 31 && i < c0) xf[i] -= sum;
          }
      }
    }
    __syncthreads();
  }
  }
  for (int j = tid; j < nc; j += THREADS) mem_st<FLOW>(a.x + m.col0 + j, xf[j]);
  __syncthreads();
}

// Back substitution of WIDE pivot blocks (the top separators: 600 .. 1500 columns), one launch per 128-column
// super-panel, right to left.  k_solve_mid walks a front's whole L11 through one CU (4.5 MB for the root of the
// 1M-edge lattice, 4.6 us per 32 columns); here launch ell handles super-panel s = S - 1 - ell of every front:
//   workgroup 0       folds x of super-panel s + 1 (solved by the previous launch) into its own 128 columns
//                     (a 128 x 128 block of L), then solves them: four chain steps with the kept inverses
//   workgroups 1..    fold the same x into the columns further left, 64 columns each (the whole chip reads L11)
// The running right-hand side lives in x[col0 ..]; launch 0 (no super-panel to the right yet) initialises it:
// t = y1 - sum of the row slices of k_big_gemv_partial.  Every column is owned by one workgroup per launch and
// the launches are ordered: plain read-modify-write, fixed summation order.
template <typename T> __global__ void __launch_bounds__(1024) k_big_solve_sp(FactorArgs<T> a, int ell, const T *part, int64_t N, int R) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  __shared__ T xf[BIG_SUPER];              // this super-panel: t, then x
  __shared__ T Ws[4 * 32 * 33];            // the super-panel's four W_b, staged transposed
  const SnMeta m = a.task_meta[a.task_begin + blockIdx.y];
  const int nc = m.nc, M = nc + m.nr + 1;
  const int S = (nc + BIG_SUPER - 1) / BIG_SUPER, sp = S - 1 - ell;
  if (sp < 0) return;
  const int K0 = BIG_SUPER * sp, K1 = min(nc, K0 + BIG_SUPER), K2 = min(nc, K1 + BIG_SUPER);
  const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  constexpr int NW = 16;
  const T *Lg = a.lvals + m.loff;
  T *xg = a.x + m.col0;
  auto t_init = [&](int j) {   // y1 - L21^T x[rows], slices subtracted in slice order
    T t = Lg[(int64_t)j * M + (M - 1)];
    if (m.nr > 0)
      for (int r = 0; r < R; r++) t -= part[(int64_t)r * N + m.col0 + j];
    return t;
  };
  const bool have_right = sp + 1 < S;
  const int nright = K2 - K1;   // 1..128 rows of the super-panel to the right
  // x of the super-panel to the right, two rows per lane: rows lane and lane + 64
  T xa = 0, xb = 0;
  if (have_right) {
    const T va = xg[K1 + min(lane, nright - 1)], vb = xg[K1 + min(lane + 64, nright - 1)];
    xa = lane < nright ? va : (T)0;
    xb = lane + 64 < nright ? vb : (T)0;
  }
  // column i: sum over the rows of the right super-panel of L(K1 + j, i) x_j, by one wave
  auto fold_column = [&](int i) {
    const T *col = Lg + (int64_t)i * M + K1;
    const T la = col[min(lane, nright - 1)], lb = col[min(lane + 64, nright - 1)];
    return wave_sum63<T>(la * xa + lb * xb);   // valid in lane 63
  };
  if (blockIdx.x > 0) {
    // ---- columns further left: 64 per workgroup, 4 per wave
    const int c_lo = 64 * (blockIdx.x - 1);
    if (c_lo >= K0) return;
    const int c_hi = min(K0, c_lo + 64);
    if (ell == 0) {   // nothing to the right of this front's last super-panel: initialise the running rhs
      for (int j = c_lo + tid; j < c_hi; j += 1024) xg[j] = t_init(j);
      return;
    }
    for (int i = c_lo + wave; i < c_hi; i += NW) {
      const T sum = fold_column(i);
      if (lane == 63) xg[i] -= sum;
    }
    return;
  }
  // ---- workgroup 0: this super-panel.  Every load whose address does not depend on a solution is requested up
  // front (the fold operands, the four inverse diagonal blocks, the in-panel fold operands): the chain below then
  // runs out of registers and LDS.
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
  const T *Wb = a.winv + (int64_t)m.wblk * 256;
  T wreg[4], lv[4][3];
#pragma unroll
  for (int bb = 0; bb < 4; bb++) {
    const int b = max(b_hi - bb, b_lo);
    wreg[bb] = Wb[(int64_t)b * 1024 + tid];   // Wb[b][c * 32 + j] = W_b(j, c)
    // L(c0 + l32, i) for the columns i in [K0, c0): half a wave per column
    const int c0 = 32 * b, cw = min(32, nc - c0), ncols = c0 - K0;
    const T *base = Lg + c0 + min(l32, cw - 1);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int i = 2 * (wave + NW * q) + half;
      lv[bb][q] = base[(int64_t)(K0 + min(i, max(ncols - 1, 0))) * M];
    }
  }
  if (tid < w) xf[tid] = ell == 0 ? t_init(K0 + tid) : xg[K0 + tid];
#pragma unroll
  for (int bb = 0; bb < 4; bb++) {   // -> Ws[bb][j * 33 + c]
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
    if (tid < 64) {   // x_b = W_b^T t_b (W is padded with an identity past cw)
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
  if (tid < w) xg[K0 + tid] = xf[tid];
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_solve_mid(FactorArgs<T> a, int w32, const T *part, int64_t N, int R) {
  if (opt_stopped(a.err)) return;   // rr_pgo_optimize: enqueued behind the iteration that met the stop rule
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int s = a.task_sn[a.task_ptr[a.task_begin + blockIdx.x]];
  // w32: the front was factored by the 32-column block kernels (inverse diagonal blocks in winv); it also sums
  // the partial products itself.
  if (w32) solve_big_front<T, THREADS>(a, a.sn_meta[s], reinterpret_cast<T *>(smem_raw), part, N, R);
  else solve_front<T, THREADS, false>(a, s, a.sn_meta[s], reinterpret_cast<T *>(smem_raw));
}

// ---- sharding over ranks -----------------------------------------------------------------------
// A boundary front that lives in place in HBM publishes its update matrix by copying the lower
// triangle of its trailing square into the exchange buffer (packed); `list` = (front, offset) pairs.
template <typename T> __global__ void __launch_bounds__(256) k_pack_boundary(FactorArgs<T> a, const int64_t *list) {
  const int s = (int)list[2 * blockIdx.y];
  T *dst = a.xch + list[2 * blockIdx.y + 1];
  const SnMeta m = a.sn_meta[s];
  const int M = m.nc + m.nr + 1, nu = m.nr + 1;
  const T *Usrc = a.lvals + m.uoff;   // element (nc, nc) of the front, ld M
  for (int j = blockIdx.x; j < nu; j += gridDim.x) {
    const int64_t o = (int64_t)j * nu - (int64_t)j * (j - 1) / 2;
    for (int i = j + threadIdx.x; i < nu; i += 256) dst[o + (i - j)] = Usrc[(int64_t)j * M + i];
  }
}

// The shared nodes' diagonal blocks and right-hand-side entries are sums over edges of ALL ranks: every rank
// publishes its partial sums behind its update matrices (k_pack_shared, list = (source offset in hvals or
// ~offset in b, length) per shared node, packed back to back at `off` of the rank's chunk); after the
// all-gather every rank adds the P partials in rank order (k_sum_shared): identical bits everywhere.
template <typename T> __global__ void __launch_bounds__(256) k_pack_shared(const int64_t *src_off, int n, const T *hvals, const T *b, T *dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t so = src_off[i];
  dst[i] = so >= 0 ? hvals[so] : b[~so];
}
// A rank whose own subtrees hit a non-positive pivot in stage 0 publishes that in the last scalar of its chunk
// (k_pack_err); after the all-gather every rank folds the P flags into its own device flag (k_merge_err) BEFORE
// stage 1's k_update tests it: either every rank applies the step or none does, and every rank reports ENOTSPD.
// The scalar carries the flag BITS (small integers, exact in fp32 too): a rank whose stage 0 timed out in a dataflow
// launch makes the group report a timeout, not a matrix that is not positive definite.
template <typename T> __global__ void k_pack_err(const int *err, T *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = (T)(*err & 0xff);
}
template <typename T> __global__ void k_merge_err(int *err, const T *xch, int64_t chunk, int64_t off, int P) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int bits = 0;
  for (int r = 0; r < P; r++) {
    const T v = xch[(int64_t)r * chunk + off];
    bits |= (v >= (T)0 && v < (T)256) ? (int)v : DEVERR_NOT_SPD;   // (anything else -- a NaN from a rank in trouble -- counts as a failure)
  }
  if (bits) atomicOr(err, bits);
}
template <typename T> __global__ void __launch_bounds__(256) k_sum_shared(const int64_t *src_off, int n, T *hvals, T *b, const T *xch, int64_t chunk, int64_t off, int P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T s = 0;
  for (int r = 0; r < P; r++) s += xch[(int64_t)r * chunk + off + i];
  const int64_t so = src_off[i];
  if (so >= 0) hvals[so] = s;
  else b[~so] = s;
}

// ------------------------------------------------------------------ gauge transfer (single-precision factor)
// The reference removes the gauge freedom of the SE(2) graph with a 1e7 prior on ONE node (:330-336).  A
// rigid motion of the whole graph leaves every error unchanged, so J V = 0 for the three fields
//   V = [ (1,0,0) ; (0,1,0) ; (-(y_i - y_o), x_i - x_o, 1) ]_i      (translations, rotation about o)
// and H0 = J^T W J is singular with null space V, b is orthogonal to V, and the reference's solution is the
// ONE solution of H0 dx = b whose anchor entries vanish (V^T (H0 + P) dx = V_a^T 1e7 dx_a = 0, V_a invertible).
// The anchor is a weak hinge, though: "rotate everything but the anchor" costs only the strain of the anchor's
// few edges against a lever of the graph's whole extent -- the smallest eigenvalue is ~1e-5 on the 400 x 250
// lattice against entries of 1e5..1e7, beyond what a single-precision factor resolves.
// With a single-precision factor the engine therefore solves the SAME singular system in another gauge and
// transfers the result:
//   (H0 + V_S diag(mu) V_S^T) y = b,  S = the pivot nodes of the root front (dense anyway: no extra fill);
//        every solution of H0 y = b with V_S^T y_S = 0 solves it, so y = dx + V c exactly;
//   dx = y - V c,  c = y_a  (rotation taken about the anchor's position)   -- re-gauged in k_update.
// Now the hinge is a whole separator.  Exact in exact arithmetic; fp64 handles keep the reference formulation.
template <typename T, typename TC> struct GaugeArgs {
  const typename VecT<TC>::V4 *pose;
  const int32_t *col_node;   // per pivot column of the root front: node << 2 | component
  int nc, M;
  T *F;                      // the root front (in place, ld M)
  T *v;                      // 3 * nc scratch: the three fields on the root's pivot columns
  TC ox, oy;                 // origin of the rotation field (centroid of S at set-up: conditioning only)
  T mu_t, mu_r;
};

template <typename T, typename TC> __global__ void __launch_bounds__(256) k_gauge_vectors(GaugeArgs<T, TC> a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nc) return;
  const int ent = a.col_node[j], node = ent >> 2, comp = ent & 3;
  const auto p = a.pose[node];
  a.v[j] = comp == 0 ? (T)1 : (T)0;
  a.v[a.nc + j] = comp == 1 ? (T)1 : (T)0;
  a.v[2 * a.nc + j] = comp == 0 ? (T)(-(p.y - a.oy)) : comp == 1 ? (T)(p.x - a.ox) : (T)1;
}

// F(i, j) += mu_t (tx_i tx_j + ty_i ty_j) + mu_r rot_i rot_j on the lower triangle of the root's pivot block
template <typename T, typename TC> __global__ void __launch_bounds__(256) k_big_gauge(GaugeArgs<T, TC> a) {
  const int j = blockIdx.x;
  const T *v0 = a.v, *v1 = a.v + a.nc, *v2 = a.v + 2 * a.nc;
  const T a0 = a.mu_t * v0[j], a1 = a.mu_t * v1[j], a2 = a.mu_r * v2[j];
  T *col = a.F + (int64_t)j * a.M;
  for (int i = j + threadIdx.x; i < a.nc; i += 256) col[i] += a0 * v0[i] + a1 * v1[i] + a2 * v2[i];
}

// ------------------------------------------------------------------ update

template <typename T, typename TC = T> struct UpdArgs {
  int n_nodes;
  typename VecT<TC>::V4 *pose;
  const uint8_t *node_dim;
  const int32_t *node_pcol, *node_offset;
  const T *x;          // permuted solution (used when dx_ref_in == nullptr)
  const T *dx_ref_in;  // reference-order step supplied by the caller (rr_pgo_update)
  T *dx_ref_out;       // reference-order copy of the applied step (may be null)
  TC sign;
  double *norm_partial;
  int gauge_anchor;    // >= 0: x is a solution in the root-separator gauge; transfer it to the anchor gauge first
  int export_only;     // 1: only write dx_ref_out (rr_pgo_linearize_solve), the state stays as it is
  const int *err;      // sticky device error flag: a failed factorisation must not touch the state
  const int32_t *node_list;    // sharded runs: the nodes this rank updates (own + shared), n_nodes = its length
  const uint8_t *norm_counts;  // sharded runs: per node, 1 = this rank adds the node's |dx|^2 (every node counted once)
  const int *gate;             // non-null: the launch does nothing unless *gate != 0 (Levenberg-Marquardt's undo inside
                               // rr_pgo_optimize: OptCtrl::reject, decided on the device by the launch before this one)
  FinArgs fin;
};

// update_nodes (:229-245) + |dx|^2 (:273).  The step of a failed factorisation (non-positive pivot) is NOT
// applied: the reference returns Err from solve() at :271 before update_nodes(), its state stays intact.
// one node's update; returns its |dx|^2 term
template <typename TO, typename T>
__device__ __forceinline__ double update_node(const UpdArgs<TO, T> &a, int node) {
  double nrm = 0.0;
  const int nd = a.node_dim[node];
  T d[3] = {0, 0, 0};
  const TO *src = a.dx_ref_in ? a.dx_ref_in + a.node_offset[node] : a.x + a.node_pcol[node];
  for (int t = 0; t < nd; t++) d[t] = (T)src[t];
  auto p = a.pose[node];
  const bool regauge = a.gauge_anchor >= 0 && !a.dx_ref_in;
  if (regauge) {
    // dx = y - V c, c = y_anchor, rotation field about the anchor's position (see "gauge transfer")
    const TO *ya = a.x + a.node_pcol[a.gauge_anchor];
    const T c0 = (T)ya[0], c1 = (T)ya[1], c2 = (T)ya[2];
    const auto pa = a.pose[a.gauge_anchor];
    d[0] = d[0] - c0 + c2 * (p.y - pa.y);
    d[1] = d[1] - c1 - c2 * (p.x - pa.x);
    if (nd == 3) d[2] -= c2;
    if (node == a.gauge_anchor) { d[0] = 0; d[1] = 0; d[2] = 0; }
  }
  if (a.dx_ref_out) {
    TO *dst = a.dx_ref_out + a.node_offset[node];
    for (int t = 0; t < nd; t++) dst[t] = (TO)d[t];
  }
  if (!a.export_only && !(regauge && node == a.gauge_anchor)) {
    if (!a.norm_counts || a.norm_counts[node])
      for (int t = 0; t < nd; t++) nrm += (double)d[t] * (double)d[t];
    p.x += a.sign * d[0];
    p.y += a.sign * d[1];
    if (nd == 3) {  // rotation *= UnitComplex::from_angle(dtheta), no renormalisation (:236)
      const T c = cos(a.sign * d[2]), s = sin(a.sign * d[2]);
      const T re = p.z * c - p.w * s, im = p.z * s + p.w * c;
      p.z = re;
      p.w = im;
    }
    a.pose[node] = p;
  }
  return nrm;
}

template <typename TO, typename T>
__global__ void __launch_bounds__(UPD_THREADS) k_update(UpdArgs<TO, T> a) {
  __shared__ double red[UPD_THREADS / 64];
  const int slot = blockIdx.x * UPD_THREADS + threadIdx.x;
  const int node = slot < a.n_nodes ? (a.node_list ? a.node_list[slot] : slot) : -1;
  double nrm = 0.0;
  if (a.gate && *a.gate == 0) return;
  // a failed factorisation, or a launch enqueued behind the iteration that met the stop rule (:298-300): the state stays
  const bool failed = a.err && (a.err[0] != 0 || a.err[1] != 0);
  if (node >= 0 && !failed) nrm = update_node(a, node);
  if (a.export_only) return;
  double tot = block_sum<double, UPD_THREADS>(nrm, red);
  if (threadIdx.x == 0) a.norm_partial[blockIdx.x] = tot;
  if (a.fin.enabled) finalize_in_last_block<UPD_THREADS>(a.fin, a.norm_partial, (int)gridDim.x, red);
}

}  // namespace rrpgo
