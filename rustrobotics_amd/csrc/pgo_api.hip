// pgo_api.hip -- librr_pgo.so: C ABI (include/rr_pgo.h) over the HIP engine.
//
// Host side of the hot path, mirroring the reference's PoseGraph
// (src/mapping/pose_graph_optimization.rs:214-373): create = parse + upload +
// one-off symbolic analysis; optimize = the GN / LM loop with every kernel of an
// iteration replayed from one hipGraph on the handle's own stream; the host
// reads two scalars (chi2, |dx|) per iteration.
#include <mutex>
#include <queue>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rr_pgo.h"
#include "host_graph.h"
#include "kernels.hip.h"
#include "flow.hip.h"
#include "lds_flow.hip.h"
#include "symbolic.h"
#include "host_threads.h"

#ifndef RRPGO_UPD_DEPTH
#define RRPGO_UPD_DEPTH 1   // k-chunks of the trailing update requested ahead of the MFMAs
#endif
namespace rrpgo {


struct ApiError : std::runtime_error {
  int code;
  ApiError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      throw ApiError(RR_PGO_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// One handle makes ~45 device buffers, and hipMalloc / hipFree cost 50-100 us each (the frees also
// synchronise): a caller that builds a graph, runs ten iterations and drops it -- the reference's own
// criterion bench -- would spend more time there than iterating.  Buffers are carved out of a few large
// chunks owned by the engine instead; a DevBuf only calls hipMalloc itself when no arena is active.
// Chunks of dropped handles are kept per device and handed to the next handle (hipFree synchronises the device and costs
// ~0.1 ms per chunk: "drop" was 0.4 - 0.5 ms of the 7 ms closure the reference's bench times, benches/graph_slam.rs:9-10, a
// loop of new + optimize(10) + drop).  Only small chunks, and only a bounded total: the arenas of the large graphs are freed.
// Nothing reads memory it has not written (scripts/gpu_soak.py rebuilds handles on poisoned memory), so a recycled chunk needs no clearing.
struct ChunkPool {
  struct Idle { int dev; char *base; size_t cap; };
  std::mutex mu;
  std::vector<Idle> idle;
  size_t idle_bytes = 0;
  static constexpr size_t kMaxChunk = 64u << 20, kMaxIdle = 256u << 20;
  char *take(size_t min_bytes, size_t *cap) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    size_t best = idle.size();
    for (size_t i = 0; i < idle.size(); i++)
      if (idle[i].dev == dev && idle[i].cap >= min_bytes && idle[i].cap <= 4 * min_bytes + (1u << 20) && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
    if (best == idle.size()) return nullptr;
    char *p = idle[best].base;
    *cap = idle[best].cap;
    idle_bytes -= idle[best].cap;
    idle.erase(idle.begin() + (long)best);
    return p;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (Idle &c : idle) (void)hipFree(c.base);
    idle.clear();
    idle_bytes = 0;
  }
  void put(int dev, char *base, size_t cap) {   // dev: the device the chunk was allocated on (the caller's current device may be another by now)
    if (cap <= kMaxChunk && dev >= 0) {
      std::lock_guard<std::mutex> lk(mu);
      if (idle_bytes + cap <= kMaxIdle) { idle.push_back(Idle{dev, base, cap}); idle_bytes += cap; return; }
    }
    (void)hipFree(base);
  }
};
static ChunkPool &chunk_pool() { static ChunkPool *p = new ChunkPool; return *p; }   // leaked on purpose, like the stream pool

struct DeviceArena {
  struct Chunk { char *base; size_t cap, used; int dev; };
  std::vector<Chunk> chunks;
  size_t next_chunk = 8u << 20;
  DeviceArena() = default;
  DeviceArena(const DeviceArena &) = delete;
  DeviceArena &operator=(const DeviceArena &) = delete;
  bool stream_failed = false;   // the engine's stream could not be synchronised when it was given back (a device fault, a launch that
                                // never drained): work may still be touching the chunks -- they are freed (hipFree waits for the
                                // device), not handed to the next handle
  // (the engine's stream was synchronised when it went back to its pool: declared after the arena, destroyed before it)
  ~DeviceArena() {
    for (Chunk &c : chunks) {
      if (stream_failed) (void)hipFree(c.base);
      else chunk_pool().put(c.dev, c.base, c.cap);
    }
  }
  void reserve(size_t bytes) { next_chunk = std::max(next_chunk, bytes); }
  void *take(size_t bytes) {
    for (Chunk &c : chunks) {
      const size_t off = (c.used + 255) & ~(size_t)255;
      if (off + bytes <= c.cap) { c.used = off + bytes; return c.base + off; }
    }
    Chunk c{nullptr, std::max(next_chunk, bytes + 256), 0, -1};
    if (hipGetDevice(&c.dev) != hipSuccess) c.dev = -1;
    if (char *p = chunk_pool().take(c.cap, &c.cap)) c.base = p;
    else HIPCHK(hipMalloc((void **)&c.base, c.cap));
    next_chunk = std::max<size_t>(8u << 20, c.cap / 4);
    c.used = bytes;
    chunks.push_back(c);
    return c.base;
  }
};
static thread_local DeviceArena *t_arena = nullptr;   // set while an engine allocates

// hipStreamCreate costs ~2 ms (a hardware queue) and hipStreamDestroy about as much: more than the ten
// Gauss-Newton iterations of the reference's bench on intel.g2o.  Streams of destroyed handles are kept
// per device and handed to the next handle (idle: they were synchronised before being returned).
struct StreamPool {
  std::mutex mu;
  std::vector<std::pair<int, hipStream_t>> idle;
  hipStream_t get() {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < idle.size(); i++)
        if (idle[i].first == dev) { hipStream_t s = idle[i].second; idle.erase(idle.begin() + (long)i); return s; }
    }
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
  }
  bool put(hipStream_t s) {   // false: the stream could not be synchronised (it is destroyed, not pooled)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipStreamDestroy(s); return false; }
    std::lock_guard<std::mutex> lk(mu);
    if (idle.size() < 16) idle.emplace_back(dev, s);
    else (void)hipStreamDestroy(s);
    return true;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &p : idle) (void)hipStreamDestroy(p.second);
    idle.clear();
  }
};
static StreamPool &stream_pool() { static StreamPool *p = new StreamPool; return *p; }   // leaked on purpose: outlives the runtime's teardown order

// RAII holders: an Engine constructor that throws (hipMalloc failure, RR_PGO_EUNSUPPORTED) still releases these
struct PooledStream {
  hipStream_t s = nullptr;
  PooledStream() = default;
  PooledStream(const PooledStream &) = delete;
  PooledStream &operator=(const PooledStream &) = delete;
  bool *failed = nullptr;   // the arena's flag (DeviceArena::stream_failed)
  ~PooledStream() { if (s && !stream_pool().put(s) && failed) *failed = true; }
  void acquire() { if (!s) s = stream_pool().get(); }
  operator hipStream_t() const { return s; }
};
struct EventHolder {
  hipEvent_t e = nullptr;
  EventHolder() = default;
  EventHolder(const EventHolder &) = delete;
  EventHolder &operator=(const EventHolder &) = delete;
  ~EventHolder() { if (e) (void)hipEventDestroy(e); }
  void create(unsigned flags) { if (!e) HIPCHK(hipEventCreateWithFlags(&e, flags)); }
  operator hipEvent_t() const { return e; }
};
template <typename U> struct PinnedBuf {
  U *p = nullptr;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  // coherent (fine-grained): a kernel's system-scope stores are visible to the polling host while the kernel runs
  void alloc(size_t n, bool coherent = true) {
    if (p) { (void)hipHostFree(p); p = nullptr; }
    HIPCHK(hipHostMalloc((void **)&p, n * sizeof(U), coherent ? (hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) : hipHostMallocDefault));
  }
  U &operator[](size_t i) const { return p[i]; }
  operator U *() const { return p; }
};

template <typename U> struct DevBuf {
  U *p = nullptr;
  size_t n = 0;
  bool owned = false;   // false: lives in the engine's arena
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p && owned) (void)hipFree(p); }
  void alloc(size_t count) {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    n = count;
    if (!count) return;
    if (t_arena) { p = (U *)t_arena->take(count * sizeof(U)); owned = false; }
    else { HIPCHK(hipMalloc((void **)&p, count * sizeof(U))); owned = true; }
  }
  void upload(const std::vector<U> &v) {
    alloc(v.size());
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice));
  }
  void zero() { if (n) HIPCHK(hipMemset(p, 0, n * sizeof(U))); }
};

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct EngineBase {
  virtual ~EngineBase() = default;
  virtual void chi2(double *out) = 0;
  virtual void linearize_solve(double lambda, int lm, double *dx_out) = 0;
  virtual void update(const double *dx, double sign) = 0;
  virtual void optimize(int solver, int iters, double *errors, int *n_errors, double *norms) = 0;
  virtual void get_state(double *out) = 0;
  virtual void set_state(const double *st) = 0;
  virtual void assemble(double lambda, int lm, std::vector<double> &hvals, std::vector<double> &b) = 0;
  virtual void iterate_async(int iters) = 0;
  virtual void sync() = 0;
  virtual void profile(int iters, double *ms, int64_t *launches) = 0;
  virtual hipStream_t stream() = 0;
  virtual void read_stamps(std::vector<unsigned long long> &out) = 0;
  virtual void mark_flow_fronts(std::vector<char> &in_flow) = 0;   // fronts whose level runs as k_big_flow
  virtual int schur_tile() const = 0;                              // tile edge of k_big_schur, 0 = no Schur split
  virtual int flow_trace(int level, std::vector<int32_t> &tasks, std::vector<unsigned long long> &stamps, int *nf, double *est_us) = 0;
  // sharded runs
  virtual void exchange_buffer(int which, void **ptr, int64_t *n, int32_t *esize) = 0;
  virtual void set_exchange_buffer(int which, void *ptr, int64_t n) = 0;
  virtual void stage(int stage, double lambda, int lm) = 0;
  virtual void read_last_scalars(double *chi, double *norm) = 0;
  virtual void debug_withhold(int mode) = 0;   // failure injection for the dataflow launches (rr_pgo_debug_withhold)
  int n_launches_per_iter = 0;
};

constexpr int HIST = 4096;  // ring of (chi2, |dx|) pairs written by k_finalize

__global__ void k_finalize_slot(const double *chi_partial, int n_chi, const double *norm_partial,
                                int n_norm, double *hist, int *counter, int advance) {
  __shared__ double red[4];
  double c = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < n_chi; i += 256) c += chi_partial[i];
  for (int i = threadIdx.x; i < n_norm; i += 256) n += norm_partial[i];
  double ct = block_sum<double, 256>(c, red);
  double nt = block_sum<double, 256>(n, red);
  if (threadIdx.x == 0) {
    int slot = *counter % HIST;
    if (n_chi > 0) hist[2 * slot] = ct;
    if (n_norm > 0) hist[2 * slot + 1] = sqrt(nt);
    if (advance) *counter = *counter + 1;
  }
}

// sharded runs: this rank's partial sums (chi2, |dx|^2) for the caller's sum all-reduce; `which` bit 0 = chi2, bit 1 = |dx|^2
__global__ void k_finalize_partial(const double *chi_partial, int n_chi, const double *norm_partial, int n_norm, double *scal) {
  __shared__ double red[4];
  double c = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < n_chi; i += 256) c += chi_partial[i];
  for (int i = threadIdx.x; i < n_norm; i += 256) n += norm_partial[i];
  double ct = block_sum<double, 256>(c, red);
  double nt = block_sum<double, 256>(n, red);
  if (threadIdx.x == 0) {
    if (n_chi > 0) scal[0] = ct;
    scal[1] = n_norm > 0 ? nt : 0.0;
  }
}

// T: type of H, b, L, x (factor + solve).  S: type of the state, the measurements and the
// linearisation arithmetic (S == T, or S = double with T = float: "mixed" mode).
template <typename T, typename S = T> class Engine final : public EngineBase {
  using V4 = typename VecT<S>::V4;
  using V2 = typename VecT<S>::V2;
  const HostGraph &g_;
  const Symbolic &sym_;
  DeviceArena arena_;   // declared before every DevBuf: destroyed after them
  PooledStream stream_;
  hipGraphExec_t gn_exec_ = nullptr;
  // graph data
  DevBuf<V4> pose_, e_meas_, e_info_a_;
  DevBuf<V2> e_info_b_;
  DevBuf<int2> e_idx_;
  DevBuf<EdgeRec<S>> e_rec_;   // SE(2): every edge's from / to / slot / measurement / information as one record (k_linearize)
  DevBuf<int64_t> e_slot_, diag_off_;
  DevBuf<int32_t> inc_ptr_, node_offset_, node_pcol_;
  DevBuf<int2> inc_list_;
  DevBuf<uint8_t> node_dim_;
  DevBuf<S> e_info3_;   // SE(3): 21 information entries per edge
  bool is3d_ = false;
  // sharding over ranks
  int rank_ = 0, world_ = 1;
  bool sharded_ = false;            // driven by rr_pgo_stage (world_size > 1, or opt.sharded with one rank)
  int64_t xch_n_ = 0;               // elements of exchange buffer 0: world * chunk
  DevBuf<int32_t> node_list_;       // nodes this rank linearises and updates: own + shared (null: all)
  DevBuf<uint8_t> norm_counts_;     // per node: this rank adds its |dx|^2 to the partial sum (and prior / lambda to its diagonal)
  DevBuf<int64_t> shared_src_;      // scalars of the shared nodes' diagonal blocks and rhs (k_pack_shared / k_sum_shared)
  int n_shared_ = 0;
  int n_list_ = 0;
  DevBuf<double> scal_own_;         // exchange buffer 1: partial (chi2, |dx|^2) unless the caller binds one
  double *scal_ = nullptr;
  hipGraphExec_t stage_exec_[2] = {nullptr, nullptr};
  DevBuf<T> xch_own_;               // exchange buffer 0 (boundary update matrices) unless the caller binds one
  T *xch_ = nullptr, *x_ptr_ = nullptr;
  int solve_threads_max_ = 512;     // RR_PGO_SOLVE_THREADS=<n>: cap of the back-substitution workgroup size
  static constexpr int factor_threads_max_ = 1024;   // 16-wave workgroups for steps whose fronts exceed 128 rows (8 waves: intel 4134 against 4318 it/s, r04)
  struct UpdMap { int64_t offset; int n_tiles; };
  std::vector<std::vector<UpdMap>> upd_maps_;   // [step][super-panel]: slice of upd_map_buf_ (k_big_update's tile list)
  std::vector<UpdMap> schur_maps_;              // [step]: k_big_schur's tile list
  static constexpr int schur_tile_ = 64;        // k_big_schur's tile edge (128 x 128 tiles, k_big_schur<T, 4>: measured 13 % slower, r03)
  bool schur_split_ = true;                     // RR_PGO_SCHUR_SPLIT=0: every super-panel's update reaches through the Schur complement (r02)
  DevBuf<int32_t> upd_map_buf_;
  static constexpr bool xcd_remap_ = true;   // k_big_update / k_big_schur: one contiguous eighth of the tile list per XCD (dispatch order: FETCH_SIZE 96 against 37 MB per launch, r02)
  int sp_solve_min_nc_ = 256;       // back substitution: levels whose widest pivot block has at least this many columns run k_big_solve_sp (RR_PGO_SP_SOLVE_MIN)
  static constexpr bool gather_update_ = true;   // big fronts: k_big_build builds the pivot columns only, a front's first trailing update gathers its tiles from the children
  // k_big_flow (flow.hip.h): levels of at most flow_max_nf_ big fronts run as ONE launch of ticket-ordered tile tasks
  struct FlowLevel {
    DevBuf<FlowRec> tasks;     // every task with a copy of its front's records (one 128-byte load per task)
    int n_tasks = 0;
    bool schur_split = false;  // the level's Schur complements are a k_big_schur launch behind the flow launch
    std::vector<FlowTask> host_tasks;   // diagnostic builds: the sorted list, for rr_pgo_debug_flow_trace
    DevBuf<unsigned long long> trace;   // diagnostic builds only
    int64_t ticket_word = 0;   // index of the level's ticket in flow_flags_
    int wfill_begin = 0, wfill_end = 0;   // the level's entries in flow_wfill_
    double est_us = 0;         // critical path of the cost model that orders the tasks
  };
  std::vector<std::unique_ptr<FlowLevel>> flow_levels_;   // per step of sym_.steps (null: launch sequence)
  std::vector<SnMeta> host_task_meta_;   // host copy of task_meta_ (one record per task = per front of a big level)
  std::vector<SnMeta> host_sn_meta_;     // host copy of sn_meta_
  DevBuf<int64_t> flow_wfill_;      // per four W blocks of a front of a flow level: offset and count of scalars in winv (k_flow_reset marks them)
  DevBuf<unsigned> flow_flags_;     // tickets + completion flags of every flow level, zeroed at the start of a factorisation
  int flow_max_nf_ = 1 << 20;       // RR_PGO_FLOW=<n> (0: never): levels of at most n fronts ...
  int flow_max_tasks_ = 1 << 20;    // RR_PGO_FLOW_TASKS=<n>: ... and of at most n tasks run as ONE k_big_flow launch: by default every level.
                                    // History on the 1M-edge lattice (r03): with tickets in the order of earliest starts only the levels of 1
                                    // and 2 fronts gained; with the list schedule and the Schur complements left to k_big_schur the levels up
                                    // to 64 fronts; with W polled in place and no tile on the chain every level does (limits 64 / 128 / 256 /
                                    // none: 4.57 / 4.54 / 4.53 / 4.52 ms) -- the launch sequence (k_big_panel32 + k_big_update) remains as the
                                    // parity alternative (RR_PGO_FLOW=0)
  // cross-level form of k_big_flow (flow.hip.h, XL): graphs of a few dozen fronts beyond LDS (sphere2500: 46 in six levels) run ALL
  // their levels as ONE launch -- BUILD tasks in place of k_big_build, Schur complements as UPDATE tasks, parent waits for child by counter
  bool xl_ = false;                 // RR_PGO_FLOW_XL=0: one build + one flow launch per level (the r03 / r04 form; bit-identical)
  int xl_max_fronts_ = 256;         // RR_PGO_FLOW_XL=<n>: graphs of at most n fronts beyond LDS
  int xl_level_tasks_ = 1 << 20;    // RR_PGO_FLOW_XL_TASKS=<n>: ... none of whose levels has more than n tasks in that form
  bool xl_refused_ = false;         // a level went over the task limit under the cross-level plan: planned again level by level
  DevBuf<FlowRec> xl_recs_;
  std::vector<FlowRec> xl_host_;
  DevBuf<int2> xl_child_done_;
  DevBuf<unsigned long long> xl_trace_;   // diagnostic builds only
  int64_t xl_ticket_word_ = 0;
  static constexpr bool fused_assembly_ = true;   // H entries and rhs of the fronts beyond LDS by k_big_build's own waves when its launch is a single round of workgroups (else a k_big_assemble launch)
  static constexpr bool flow_deep_ = true;   // fast mode: panel steps of blocks 0 and 1 look back over the previous super-panel (see build_flow_levels)
  bool flow_exact_ = false;         // RR_PGO_FLOW_EXACT=1: bit-identical to the launch sequence (tile (0, 0) forms the next super-panel's first block)
  struct SolveFlowLevel { DevBuf<SolveFlowFront> fronts; DevBuf<SolveFlowTask> tasks; int n_tasks = 0; int64_t ticket_word = 0; };
  std::vector<std::unique_ptr<SolveFlowLevel>> solve_flow_;   // per step: k_big_solve_flow's tasks and counters (null: k_big_solve_sp launches)
  bool solve_flow_on_ = true;       // RR_PGO_SOLVE_FLOW=0: one k_big_solve_sp launch per 128 columns
  int flow_schur_min_ = 512;        // RR_PGO_FLOW_SCHUR_MIN=<n>: flow levels with at least n Schur tiles leave them to k_big_schur
  int flow_grid_ = 0;               // persistent workgroups of a flow launch (RR_PGO_FLOW_GRID; default CUs x RRPGO_FLOW_WAVES)
  // k_factor_flow / k_solve_flow (lds_flow.hip.h): every front in LDS, the factorisation and the back substitution ONE launch each
  bool lds_flow_ = false;
  DevBuf<int32_t> child_dep_, parent_dep_;
  std::vector<int32_t> host_child_dep_;
  DevBuf<unsigned> dep_flags_;      // [0, S) factor flags, [S, 2 S) solve flags, then the two tickets on lines of their own, then a word nobody sets
  DevBuf<LdsFlowTask> lds_ftasks_, lds_stasks_;
  int lds_n_tasks_ = 0, lds_flow_cus_ = 256;
  std::vector<LdsFlowTask> host_stasks_;   // k_solve_flow's tickets: [per narrow level of fronts beyond LDS, top level first: GEMV units, L11 tasks] + the LDS tasks
  int lds_n_stasks_ = 0;
  std::vector<char> mid_in_flow_;    // per step: the level's back substitution runs as tasks of k_solve_flow, not as launches
  unsigned long long wait_ticks_ = 200000000ull;   // bound of one in-launch wait: 2 s of the 100 MHz wall clock (RR_PGO_FLOW_TIMEOUT_MS)
  // failure injection (rr_pgo_debug_withhold): what was changed, to put it back
  int withheld_child_ = -1, withheld_parent_ = -1, withheld_level_ = -1, withheld_task_ = -1;
  int32_t withheld_parent_val_ = -1;
  FlowRec withheld_rec_{};
  static constexpr int kDeadFlagWords = 4096;   // flag words behind the live ones that no consumer ever looks at
  // gauge transfer (single-precision factor, Gauss-Newton; kernels.hip.h "gauge transfer")
  bool gauge_ok_ = false;            // the root front is a big front with an SE2 pivot node
  bool gauge_now_ = false;           // the system being factored was linearised without the anchor prior
  int gauge_root_ = -1;              // supernode that takes the rank-3 term
  DevBuf<int32_t> gauge_col_node_;
  DevBuf<T> gauge_v_;
  double gauge_ox_ = 0, gauge_oy_ = 0, gauge_mu_t_ = 0, gauge_mu_r_ = 0;
  DevBuf<int64_t> pack_list_;       // (front, exchange offset) of the in-place boundary fronts this rank owns
  int n_pack_ = 0, pack_max_nu_ = 0;
  // numeric
  DevBuf<T> hvals_, b_, x_, dx_ref_, lvals_, uvals_, winv_, xnew_, gemv_part_;
  static constexpr int kGemvSlices = 16;   // row slices of the multi-workgroup L21^T x product
  DevBuf<double> chi_partial_, norm_partial_, hist_;
  DevBuf<int> counter_, err_, blocks_done_;
  DevBuf<unsigned long long> stamps_;  // diagnostic builds only
  // symbolic tables
  DevBuf<int32_t> task_ptr_, task_sn_, fasm_src_, fasm_dst_, fasm_colptr_, fdup_src_, fdup_dst_, scat_, rel_, perm_, sn_rows_;
  DevBuf<SnMeta> sn_meta_, task_meta_;
  DevBuf<ChildMeta> child_meta_;
  std::vector<int> step_solve_lds_;  // scalars of LDS the back-solve of each step needs
  PinnedBuf<double> host_pair_;      // pinned: chi2, |dx|, and (as an int in slot 2) the device error flag; behind them (32-byte
                                     // aligned) the OPT_RING slots rr_pgo_optimize's device-side loop publishes its iterations in
  PinnedBuf<V4> state_stage_;        // rr_pgo_set_state's staging buffer (made on first use) and the event behind its last copy
  EventHolder state_stage_ev_;
  std::vector<double> state_in_;     // ... and the state that call was given
  DevBuf<V4> pose_saved_;            // ... and its device-side copy
  OptSlot *opt_ring_ = nullptr;      // = host_pair_ + 4: host-coherent, written by the device, polled by optimize_pipelined
  DevBuf<OptCtrl> opt_ctrl_;         // the loop state of the running rr_pgo_optimize call (kernels.hip.h, OptCtrl)
  bool opt_active_ = false;          // launches enqueued now belong to a pipelined rr_pgo_optimize call ...
  bool opt_first_ = false;           // ... and the next linearisation is the call's first launch: it resets the loop state
  bool opt_lambda_dev_ = false;      // ... Levenberg-Marquardt: the linearisation reads lambda from the loop state
  int opt_publish_ = 0;              // ... LinArgs::publish of the next linearisation
  double opt_lambda0_ = 0.01;
  bool stop_dirty_ = false;          // the stop word err_[1] may be set: cleared before the next launch outside such a call
  bool sync_optimize_ = false;       // RR_PGO_SYNC_OPTIMIZE=1: rr_pgo_optimize with one host round trip per iteration (the r01 - r05
                                     // form, bit-identical: the parity alternative; always with the edge-parallel linearisation)
  int n_lin_blocks_ = 0, n_upd_blocks_ = 0;
  bool no_graph_ = false, force_graph_ = false;   // RR_PGO_NO_GRAPH=1 / RR_PGO_FORCE_GRAPH=1 (read when the handle is created): plain launches / replays of the captured hipGraph everywhere
  bool edge_lin_ = false;           // RR_PGO_EDGE_LINEARIZE=1: k_linearize_edges (one thread per edge, atomics) instead of the pull form
  bool edge_lin_wave_ = false;      // RR_PGO_EDGE_LINEARIZE=2: k_linearize_wave_edges (one WAVEFRONT per edge, operands staged in LDS, LDS-reduced scatter-add)
  int host_counter_ = 0;             // mirrors the device slot counter

 public:
  Engine(const HostGraph &g, const Symbolic &sym, int rank, int world, bool sharded)
      : g_(g), sym_(sym), rank_(rank), world_(world), sharded_(sharded || world > 1) {
    struct ArenaScope {   // every DevBuf::alloc of this constructor draws from arena_
      explicit ArenaScope(DeviceArena *a) { t_arena = a; }
      ~ArenaScope() { t_arena = nullptr; }
    } scope(&arena_);
    const bool ctimes = getenv("RR_PGO_ANALYZE_TIMES") != nullptr;   // wall time of the phases of this constructor on stderr
    double ct0 = now_ms();
    auto cmark = [&](const char *what) {
      if (!ctimes) return;
      const double t = now_ms();
      std::fprintf(stderr, "engine:  %-12s %8.3f ms\n", what, t - ct0);
      ct0 = t;
    };
    // the factor storage dominates: one chunk sized for it and the value arrays, tables follow in 8 MB chunks
    arena_.reserve((size_t)(sym.l_elems + sym.u_elems + sym.n_hvals + 8 * (int64_t)g.dim + sym.xch_elems) * sizeof(T) + (4u << 20));
    stream_.failed = &arena_.stream_failed;
    stream_.acquire();
    cmark("stream");
    host_pair_.alloc(4 + OPT_RING * sizeof(OptSlot) / sizeof(double));
    opt_ring_ = reinterpret_cast<OptSlot *>(host_pair_.p + 4);
    cmark("pinned");
    const int N = g.n_nodes(), E = g.n_edges();
    // ---- graph arrays
    is3d_ = g.has_se3;
    std::vector<uint8_t> ndim(N);
    for (int i = 0; i < N; i++) ndim[i] = (uint8_t)node_dim(g.node_kind[i]);
    node_dim_.upload(ndim);
    std::vector<int2> eidx(E);
    std::vector<int64_t> eslot(E);
    for (int k = 0; k < E; k++) {
      eidx[k] = int2{g.edge_from[k], g.edge_to[k]};
      eslot[k] = (sym.blk_off[sym.edge_slot[k]] << 1) | (sym.edge_transposed[k] ? 1 : 0);
    }
    e_idx_.upload(eidx);
    e_slot_.upload(eslot);
    if (!is3d_) {
      std::vector<V4> pose(N);
      for (int i = 0; i < N; i++) {
        const double *s = &g.node_state[g.node_state_off[i]];
        if (g.node_kind[i] == NODE_SE2) pose[i] = V4{(S)s[0], (S)s[1], (S)std::cos(s[2]), (S)std::sin(s[2])};
        else pose[i] = V4{(S)s[0], (S)s[1], (S)0, (S)0};
      }
      pose_.upload(pose);
      std::vector<V4> emeas(E), einfa(E);
      std::vector<V2> einfb(E);
      for (int k = 0; k < E; k++) {
        const double *m = &g.edge_meas[g.edge_meas_off[k]];
        const double *w = &g.edge_info[g.edge_info_off[k]];
        if (g.edge_kind[k] == EDGE_SE2) {
          emeas[k] = V4{(S)m[0], (S)m[1], (S)std::cos(m[2]), (S)std::sin(m[2])};
          einfa[k] = V4{(S)w[0], (S)w[1], (S)w[2], (S)w[3]};
          einfb[k] = V2{(S)w[4], (S)w[5]};
        } else {
          emeas[k] = V4{(S)m[0], (S)m[1], (S)0, (S)0};
          einfa[k] = V4{(S)w[0], (S)w[1], (S)0, (S)w[2]};
          einfb[k] = V2{(S)0, (S)0};
        }
      }
      e_meas_.upload(emeas);
      e_info_a_.upload(einfa);
      e_info_b_.upload(einfb);
      std::vector<EdgeRec<S>> erec(E);
      for (int k = 0; k < E; k++) erec[k] = EdgeRec<S>{eidx[k].x, eidx[k].y, eslot[k], emeas[k], einfa[k], einfb[k]};
      e_rec_.upload(erec);
    } else {
      // SE(3): (t, -), (q) pairs; quaternions normalised like UnitQuaternion::from_quaternion
      auto pack7 = [](const double *s, V4 &a, V4 &b) {
        const double n = std::sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5] + s[6] * s[6]);
        a = V4{(S)s[0], (S)s[1], (S)s[2], (S)0};
        b = V4{(S)(s[3] / n), (S)(s[4] / n), (S)(s[5] / n), (S)(s[6] / n)};
      };
      std::vector<V4> pose(2 * (size_t)N), emeas(2 * (size_t)E);
      std::vector<S> einfo(21 * (size_t)E);
      for (int i = 0; i < N; i++) pack7(&g.node_state[g.node_state_off[i]], pose[2 * i], pose[2 * i + 1]);
      for (int k = 0; k < E; k++) {
        pack7(&g.edge_meas[g.edge_meas_off[k]], emeas[2 * k], emeas[2 * k + 1]);
        const double *w = &g.edge_info[g.edge_info_off[k]];
        for (int t = 0; t < 21; t++) einfo[21 * (size_t)k + t] = (S)w[t];
      }
      pose_.upload(pose);
      e_meas_.upload(emeas);
      e_info3_.upload(einfo);
    }
    std::vector<int2> inc(sym.inc_list.size());
    if (E >= (1 << 27)) throw ApiError(RR_PGO_EUNSUPPORTED, "more than 2^27 edges");
    for (size_t q = 0; q < inc.size(); q++) {
      int k = sym.inc_list[q] >> 1, role = sym.inc_list[q] & 1;
      // Sharded runs: an edge's contributions to diagonal blocks, right-hand side and chi2 are computed by ONE
      // rank, its owner = the rank of its from-node, else of its to-node, else (both shared) rank 0 -- only
      // the owner is sure to hold current poses of both endpoints.  The off-diagonal block of an edge between
      // two shared nodes is needed (and computable) on every rank.
      const int pf = world_ > 1 ? sym.node_part[g.edge_from[k]] : 0, pt = world_ > 1 ? sym.node_part[g.edge_to[k]] : 0;
      const int owns = (pf >= 0 ? pf : pt >= 0 ? pt : 0) == rank_ ? 1 : 0;
      const int offd = role == 0 && (owns || (pf < 0 && pt < 0)) ? 1 : 0;
      inc[q] = int2{(k << 4) | (offd << 3) | (owns << 2) | ((g.edge_kind[k] == EDGE_SE2_XY ? 1 : 0) << 1) | role,
                    role ? g.edge_from[k] : g.edge_to[k]};
    }
    inc_ptr_.upload(sym.inc_ptr);
    inc_list_.upload(inc);
    node_offset_.upload(g.node_offset);
    node_pcol_.upload(sym.node_pcol);
    diag_off_.upload(sym.diag_off);
    cmark("graph arrays");
    // ---- numeric buffers
    hvals_.alloc((size_t)sym.n_hvals);
    hvals_.zero();
    b_.alloc((size_t)g.dim);
    x_.alloc((size_t)g.dim);
    dx_ref_.alloc((size_t)g.dim);
    b_.zero(); x_.zero(); dx_ref_.zero();
    x_ptr_ = x_.p;
    n_list_ = N;
    if (sharded_) {
      xch_n_ = std::max<int64_t>(sym.xch_elems, 64 * (int64_t)world_);
      xch_own_.alloc((size_t)xch_n_ + 4);
      xch_own_.zero();
      xch_ = xch_own_.p;
      scal_own_.alloc(2);
      scal_own_.zero();
      scal_ = scal_own_.p;
    }
    if (world_ > 1) {
      std::vector<int32_t> nl;
      std::vector<uint8_t> nc(N);
      std::vector<int64_t> sh;   // the shared nodes' diagonal blocks (offset in hvals) and rhs entries (~offset in b)
      for (int i = 0; i < N; i++) {
        const int pt = sym.node_part[i];
        if (pt == rank_ || pt < 0) nl.push_back(i);
        nc[i] = (pt == rank_ || (pt < 0 && rank_ == 0)) ? 1 : 0;
        if (pt < 0) {
          const int d = node_dim(g.node_kind[i]);
          for (int t = 0; t < d * d; t++) sh.push_back(sym.diag_off[i] + t);
          for (int t = 0; t < d; t++) sh.push_back(~(int64_t)(g.node_offset[i] + t));
        }
      }
      n_list_ = (int)nl.size();
      node_list_.upload(nl);
      norm_counts_.upload(nc);
      n_shared_ = (int)sh.size();
      if (sym.xch_shared_off + n_shared_ > sym.xch_chunk) throw ApiError(RR_PGO_EINVAL, "internal: exchange chunk too small");
      shared_src_.upload(sh);
      std::vector<int64_t> pl;
      for (int f = 0; f < sym.S; f++)
        if (sym.sn_xch_off[f] >= 0 && sym.sn_big[f] && sym.sn_owner[f] == rank_) {
          pl.push_back(f);
          pl.push_back(sym.sn_xch_off[f]);
          pack_max_nu_ = std::max(pack_max_nu_, sym.sn_nrows[f] + 1);
        }
      n_pack_ = (int)pl.size() / 2;
      pack_list_.upload(pl);
    }
    setup_gauge();
    lvals_.alloc((size_t)sym.l_elems + 4);
    uvals_.alloc((size_t)sym.u_elems + 4);
    lvals_.zero(); uvals_.zero();
    {
      bool any_big = false;
      for (const Step &st : sym.steps) any_big = any_big || st.kind != STEP_TASKS;
      gemv_part_.alloc(any_big ? (size_t)kGemvSlices * g.dim : 4);
    }
    if (const char *e = getenv("RR_PGO_SOLVE_THREADS")) solve_threads_max_ = std::atoi(e);
    if (const char *e = getenv("RR_PGO_SP_SOLVE_MIN")) sp_solve_min_nc_ = std::atoi(e);
    if (const char *e = getenv("RR_PGO_SCHUR_SPLIT")) schur_split_ = std::atoi(e) != 0;
    no_graph_ = getenv("RR_PGO_NO_GRAPH") != nullptr;
    force_graph_ = getenv("RR_PGO_FORCE_GRAPH") != nullptr && !no_graph_;
    if (const char *e = getenv("RR_PGO_FLOW")) flow_max_nf_ = std::atoi(e);
    if (const char *e = getenv("RR_PGO_FLOW_EXACT")) flow_exact_ = std::atoi(e) != 0;
    if (const char *e = getenv("RR_PGO_FLOW_TASKS")) flow_max_tasks_ = std::atoi(e);
    if (const char *e = getenv("RR_PGO_SOLVE_FLOW")) solve_flow_on_ = std::atoi(e) != 0;
    if (const char *e = getenv("RR_PGO_FLOW_SCHUR_MIN")) flow_schur_min_ = std::atoi(e);
    n_lin_blocks_ = (int)(((int64_t)n_list_ * LIN_GROUP + LIN_THREADS - 1) / LIN_THREADS);
    edge_lin_ = getenv("RR_PGO_EDGE_LINEARIZE") != nullptr && !is3d_ && !sharded_ && g_.n_edges() > 0;
    edge_lin_wave_ = edge_lin_ && std::atoi(getenv("RR_PGO_EDGE_LINEARIZE")) == 2;
    if (edge_lin_) n_lin_blocks_ = std::max(1, (g_.n_edges() + (edge_lin_wave_ ? WE_EDGES : LIN_THREADS) - 1) / (edge_lin_wave_ ? WE_EDGES : LIN_THREADS));
    n_upd_blocks_ = (n_list_ + UPD_THREADS - 1) / UPD_THREADS;
    chi_partial_.alloc((size_t)n_lin_blocks_);
    norm_partial_.alloc((size_t)n_upd_blocks_);
    hist_.alloc(2 * HIST);
    hist_.zero();
    counter_.alloc(1);
    counter_.zero();
    err_.alloc(2);   // [0] sticky error flag, [1] stop word of the running rr_pgo_optimize call (kernels.hip.h, OptCtrl)
    err_.zero();
    opt_ctrl_.alloc(1);
    opt_ctrl_.zero();
    sync_optimize_ = getenv("RR_PGO_SYNC_OPTIMIZE") != nullptr;
    blocks_done_.alloc(2);   // [0] k_update's last-workgroup counter, [1] k_linearize's
    blocks_done_.zero();
    cmark("numeric");
    // ---- symbolic tables
    task_ptr_.upload(sym.task_ptr);
    task_sn_.upload(sym.task_sn);
    {
      std::vector<SnMeta> meta(sym.S);
      std::vector<ChildMeta> cm(sym.child_list.size());
      int64_t wblk_total = 0;
      for (int f = 0; f < sym.S; f++) {
        SnMeta &m = meta[f];
        m.nc = sym.sn_ncols[f];
        m.nr = sym.sn_nrows[f];
        m.col0 = sym.sn_col0[f];
        m.uld = sym.sn_uld[f];
        m.asm_begin = (int32_t)sym.fasm_ptr[f];
        m.asm_count = (int32_t)(sym.fasm_ptr[f + 1] - sym.fasm_ptr[f]);
        m.dup_begin = (int32_t)sym.fdup_ptr[f];
        m.dup_count = (int32_t)(sym.fdup_ptr[f + 1] - sym.fdup_ptr[f]);
        m.child_begin = sym.child_ptr[f];
        m.child_count = sym.child_ptr[f + 1] - sym.child_ptr[f];
        if (sym.sn_rows_ptr[f] > 0x7fffffffLL) throw ApiError(RR_PGO_EUNSUPPORTED, "row structure exceeds 32-bit indexing");
        m.rows_ptr = (int32_t)sym.sn_rows_ptr[f];
        m.wblk = (int32_t)wblk_total;
        // 16 x 16 blocks per 16 columns; the 32-column block kernels of the big fronts keep 32 x 32 per 32 columns
        wblk_total += sym.sn_big[f] ? 4 * ((sym.sn_ncols[f] + 31) / 32) : (sym.sn_ncols[f] + 15) / 16;
        m.loff = sym.sn_loff[f];
        m.uoff = sym.sn_uoff[f];
        if (sym.sn_xch_off[f] >= 0 && !sym.sn_big[f]) {   // LDS boundary front: writes straight into the exchange buffer
          m.uld = -1;
          m.uoff = sym.sn_xch_off[f];
        }
      }
      for (size_t q = 0; q < cm.size(); q++) {
        const int c = sym.child_list[q];
        cm[q].uoff = sym.sn_uoff[c];
        cm[q].scat_ptr = sym.scat_ptr[c];
        cm[q].rel_ptr = sym.rel_ptr[c];
        cm[q].ncu = sym.sn_nrows[c] + 1;
        cm[q].uld = sym.sn_uld[c];
        if (sym.sn_xch_off[c] >= 0) {   // boundary child: every rank reads it from the (all-reduced) exchange buffer
          cm[q].uld = -1;
          cm[q].uoff = sym.sn_xch_off[c];
        }
      }
      sn_meta_.upload(meta);
      host_sn_meta_ = meta;
      {
        std::vector<SnMeta> tm(sym.task_ptr.size() - 1);
        for (size_t t = 0; t + 1 < sym.task_ptr.size(); t++) tm[t] = meta[sym.task_sn[sym.task_ptr[t]]];
        task_meta_.upload(tm);
        host_task_meta_ = tm;
        // k_big_panel32 / k_big_update / k_big_flow address a front with 32-bit BYTE offsets from its base
        for (const Step &st : sym.steps)
          if (st.kind == STEP_BIG)
            for (int t = st.task_begin; t < st.task_end; t++) {
              const int sn = sym.task_sn[sym.task_ptr[t]];
              const int64_t Mf = sym.sn_ncols[sn] + sym.sn_nrows[sn] + 1;
              if (Mf * Mf * (int64_t)sizeof(T) >= (1LL << 32))
                throw ApiError(RR_PGO_EUNSUPPORTED, "a front of more than 4 GiB exceeds the 32-bit addressing of the big-front kernels");
            }
      }
      if (wblk_total > 0x7fffff00LL) throw ApiError(RR_PGO_EUNSUPPORTED, "too many diagonal blocks");
      winv_.alloc((size_t)wblk_total * 256 + 4);
      winv_.zero();
      {
        bool any_big = false;
        for (const Step &st : sym.steps) any_big = any_big || st.kind == STEP_BIG;
        xnew_.alloc(any_big ? (size_t)wblk_total * 256 + 4 : 4);   // the chain waves' X blocks of the flow fronts (laid out like winv; flow.hip.h)
        xnew_.zero();
      }
      child_meta_.upload(cm);
      if (const char *e = getenv("RR_PGO_FLOW_TIMEOUT_MS")) wait_ticks_ = (unsigned long long)std::max(1.0, std::atof(e) * 1e5);
      if (sym.lds_flow) {
        // dataflow launches of the LDS fronts: who waits for whom (a dependency inside one task needs no flag), and
        // one 128-byte record per ticket
        if (sharded_ || world_ > 1) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: dataflow schedule on a sharded handle");
        lds_flow_ = true;
        // (the dataflow step is step 0: every LDS front; fronts beyond LDS follow level by level as STEP_BIG steps)
        if (sym.steps.empty() || sym.steps[0].kind != STEP_TASKS || sym.steps[0].task_begin != 0)
          throw ApiError(RR_PGO_EUNSUPPORTED, "internal: the dataflow step must come first");
        for (size_t si = 1; si < sym.steps.size(); si++)
          if (sym.steps[si].kind != STEP_BIG) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: a second LDS step behind the dataflow step");
        const int nt = sym.steps[0].task_end;
        std::vector<int32_t> task_of(sym.S, -1);
        for (size_t t = 0; t + 1 < sym.task_ptr.size(); t++)
          for (int q = sym.task_ptr[t]; q < sym.task_ptr[t + 1]; q++) task_of[sym.task_sn[q]] = (int)t;
        host_child_dep_.assign(cm.size(), -1);
        std::vector<int32_t> pdep(sym.S, -1);
        for (int f = 0; f < sym.S; f++) {
          for (int q = sym.child_ptr[f]; q < sym.child_ptr[f + 1]; q++) {
            const int c = sym.child_list[q];
            if (task_of[c] != task_of[f]) host_child_dep_[q] = c;
            else if (c > f) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: child after its parent in one task");
          }
          const int pf = sym.sn_parent[f];
          if (pf >= 0 && !sym.sn_big[pf] && task_of[pf] != task_of[f]) pdep[f] = sym.S + pf;   // (a parent beyond LDS was solved by an earlier launch)
        }
        child_dep_.upload(host_child_dep_);
        parent_dep_.upload(pdep);
        dep_flags_.alloc((size_t)2 * sym.S + 128);
        dep_flags_.zero();
        std::vector<LdsFlowTask> ft(nt), stk(nt);
        for (int t = 0; t < nt; t++) {
          LdsFlowTask r{};
          r.sn_begin = sym.task_ptr[t];
          r.sn_end = sym.task_ptr[t + 1];
          r.sn = sym.task_sn[r.sn_begin];
          r.m = meta[r.sn];
          ft[t] = r;
        }
        if ((int)sym.solve_order.size() != nt) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: solve order of the dataflow schedule");
        for (int k = 0; k < nt; k++) {
          LdsFlowTask r = ft[sym.solve_order[k]];
          r.sn = sym.task_sn[r.sn_end - 1];
          r.m = meta[r.sn];
          stk[k] = r;
        }
        lds_ftasks_.upload(ft);
        host_stasks_ = std::move(stk);   // uploaded by plan_mid_solve_tasks (fronts beyond LDS may join the list)
        lds_n_tasks_ = nt;
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        (void)hipDeviceGetAttribute(&lds_flow_cus_, hipDeviceAttributeMultiprocessorCount, dev);
        lds_flow_cus_ = std::max(lds_flow_cus_, 1);
        if (const char *e = getenv("RR_PGO_LDS_FLOW_GRID")) lds_flow_cus_ = std::max(1, std::atoi(e));   // workgroups of the two dataflow launches (experiments)
      }
    }
    cmark("front tables");
    build_flow_levels();   // (after the front records: every flow task carries a copy of its front's)
    build_update_maps();
    cmark("flow levels");
    fasm_src_.upload(sym.fasm_src);
    fasm_dst_.upload(sym.fasm_dst);
    fasm_colptr_.upload(sym.fasm_colptr);
    fdup_src_.upload(sym.fdup_src);
    fdup_dst_.upload(sym.fdup_dst);
    scat_.upload(sym.scat);
    rel_.upload(sym.rel);
    perm_.upload(sym.perm);
    sn_rows_.upload(sym.sn_rows);
    for (const Step &st : sym.steps) {
      int need = 0;
      if (st.kind == STEP_MID) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: the panel-in-LDS step class has no kernel (SymbolicOptions::panel_budget_elems must be 0)");
      if (st.kind == STEP_TASKS) {
        for (int t = st.task_begin; t < st.task_end; t++)
          for (int q = sym.task_ptr[t]; q < sym.task_ptr[t + 1]; q++) {
            int s = sym.task_sn[q], nc = sym.sn_ncols[s], nr = sym.sn_nrows[s];
            if (nc > 256) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: LDS-path supernode wider than 256 columns");
            // x[rows] and t, padded as solve_front lays them out; k_solve_flow adds an LDS image of L11 where it fits
            const int ncp = (nc + 15) / 16 * 16, base = ((nr + 15) & ~15) + ncp + 2;
            const int cap = (int)((size_t)kMaxLds / sizeof(T));
            const bool image = sym.lds_flow && &st == &sym.steps[0];
            need = std::max(need, image && base + ncp * (ncp + 1) <= cap ? base + ncp * (ncp + 1) : base + ncp / 16 * (2 * 16 * 17));
          }
      } else {
        for (int t = st.task_begin; t < st.task_end; t++) {
          int s = sym.task_sn[sym.task_ptr[t]];
          need = std::max(need, sym.sn_ncols[s] + sym.sn_nrows[s] + 64 * 65 + 4);
        }
      }
      if ((size_t)need * sizeof(T) > (size_t)kMaxLds) throw ApiError(RR_PGO_EUNSUPPORTED, "front too large for the back-substitution scratch");
      step_solve_lds_.push_back(need);
    }
#ifdef RRPGO_STAMPS
    stamps_.alloc((size_t)sym.S * 12 + 400000);   // per-front phase stamps | launch trace
    stamps_.zero();
#endif
    plan_mid_solve_tasks();
    cmark("index tables");
    configure_kernels();
    cmark("kernel attrs");
    n_launches_per_iter = 2;
    for (const Step &st : sym.steps)
      n_launches_per_iter += st.kind == STEP_BIG ? count_big_launches(st) + count_big_solve_launches(st) : 2;
    if (lds_flow_) {   // linearise, k_factor_flow, the levels of fronts beyond LDS, k_solve_flow, update
      n_launches_per_iter = 4 + (xl_ ? 1 : 0);
      for (size_t si = 1; si < sym.steps.size(); si++)
        n_launches_per_iter += (xl_ ? 0 : count_big_launches(sym.steps[si])) + (mid_in_flow_[si] ? 0 : count_big_solve_launches(sym.steps[si]));
    }
  }

  ~Engine() override {
    if (gn_exec_) (void)hipGraphExecDestroy(gn_exec_);   // streams, events and the pinned buffer release themselves
    for (hipGraphExec_t e : stage_exec_) if (e) (void)hipGraphExecDestroy(e);
  }

  hipStream_t stream() override { return stream_; }

 private:
  // ---- k_big_flow, host side: the task list of every level that runs as a dataflow launch.
  // Tasks and their dependencies are those flow.hip.h waits for; the list is sorted by the earliest start time a
  // task can have under a simple cost model with unlimited workgroups (us; the constants are the r02 in-kernel marks).
  // A dependency always FINISHES (so also starts) before its dependant starts, hence the sorted list is a topological
  // order -- all the ticket scheme needs -- and roughly the order in which tasks become ready.
  void build_flow_levels() {
    flow_levels_.clear();
    flow_levels_.resize(sym_.steps.size());
    solve_flow_.clear();
    solve_flow_.resize(sym_.steps.size());
    constexpr double kDiag = 4.0, kPre = 2.5, kNewest = 1.5, kX = 1.5, kLook = 4.5, kTile = 8.0, kTail = 4.5, kHop = 0.7;
    int64_t words = 0;
    int dev = 0, cus = 256;
    HIPCHK(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    flow_grid_ = std::max(cus, 1) * (sizeof(T) == 4 ? RRPGO_FLOW_WAVES : 1);
    if (const char *e = getenv("RR_PGO_FLOW_GRID")) flow_grid_ = std::max(1, std::atoi(e));
    {
      // the cross-level form: every front beyond LDS sits above the LDS fronts (the dataflow schedule's condition), few of them,
      // none with parallel-edge blocks (k_big_assemble_dup has no task form), no gauge term on the root, default flow limits
      if (const char *e = getenv("RR_PGO_FLOW_XL")) xl_max_fronts_ = std::atoi(e);
      int n_big = 0;
      bool dup = false;
      for (const Step &st : sym_.steps)
        if (st.kind == STEP_BIG)
          for (int t = st.task_begin; t < st.task_end; t++) {
            const int sn = sym_.task_sn[sym_.task_ptr[t]];
            n_big++;
            dup = dup || sym_.fdup_ptr[sn + 1] > sym_.fdup_ptr[sn];
          }
      xl_ = lds_flow_ && !sharded_ && world_ == 1 && gather_update_ && fused_assembly_ && !gauge_ok_ && !dup && n_big > 0 && n_big <= xl_max_fronts_ &&
            flow_max_nf_ >= (1 << 20) && flow_max_tasks_ >= (1 << 20) && !xl_refused_;
      if (const char *e = getenv("RR_PGO_FLOW_XL_TASKS")) xl_level_tasks_ = std::atoi(e);
    }
    // first W block of every supernode in winv (as the SnMeta table lays them out), for the marks k_flow_reset writes
    std::vector<int64_t> sn_wblk(sym_.S + 1, 0);
    for (int f = 0; f < sym_.S; f++)
      sn_wblk[f + 1] = sn_wblk[f] + (sym_.sn_big[f] ? 4 * ((sym_.sn_ncols[f] + 31) / 32) : (sym_.sn_ncols[f] + 15) / 16);
    std::vector<int64_t> wfill;
    std::vector<std::vector<FlowTask>> all_tasks(sym_.steps.size());
    std::vector<std::vector<FlowFront>> all_fronts(sym_.steps.size());
    for (size_t si = 0; si < sym_.steps.size(); si++) {
      const Step &st = sym_.steps[si];
      const int nf = st.task_end - st.task_begin;
      if (st.kind != STEP_BIG || nf > flow_max_nf_ || flow_max_nf_ <= 0) continue;
      auto lvl = std::make_unique<FlowLevel>();
      lvl->ticket_word = words;
      // The Schur complements of the level's fronts: as UPDATE tiles of this launch (one per super-panel), or left to ONE
      // k_big_schur launch behind it (K = nc, six workgroups per CU, MfmaUtil 52 - 55 % on the lattice's levels of 4 ... 32
      // fronts).  In the launch they cost twice the workgroup time, but at the top of the tree the launch is
      // bound by its chain and most workgroups idle: the tiles stay in when the list schedule says they fit into 60 % of
      // that idle time (lattice: the level of 2 fronts, -60 us; the level of 4 fronts would lose 70 us)
      int64_t schur_tiles = 0;
      double schur_in_flow_us = 0;   // workgroup time of the Schur tiles as UPDATE tasks
      for (int z = 0; z < nf; z++) {
        const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
        const int nc = sym_.sn_ncols[sn], M = nc + sym_.sn_nrows[sn] + 1, o = big_schur_origin(nc, schur_tile_);
        if (o < M) {
          const int64_t nt = (M - o + 63) / 64;
          schur_tiles += nt * (nt + 1) / 2;
          schur_in_flow_us += (double)(nt * (nt + 1) / 2) * ((nc + BIG_SUPER - 1) / BIG_SUPER) * kTile;
        }
      }
      lvl->schur_split = schur_split_ && schur_tiles >= flow_schur_min_;
      if (xl_) lvl->schur_split = false;   // one launch: the Schur complements are UPDATE tasks
      std::vector<FlowTask> tasks;
      std::vector<FlowFront> fronts(nf);
      double level_end = 0, busy_us = 0;
      bool too_many = false;
      auto plan = [&]() {
      tasks.clear();
      level_end = busy_us = 0;
      // The level's tasks in their natural order (front by front, step by step), each with the flags it waits for and the
      // flags it sets. The ticket order is then the START order of a list schedule of this DAG on the launch's workgroups
      // (priority: longest remaining path), so that a workgroup drawing the next ticket finds what the schedule would
      // have given it: the next chain step of a front ahead of the bulk of the previous super-panel's far tiles.
      struct Gen {
        FlowTask t;
        int need0, need1, prod0, prod1;
        int wflag;             // the W flag a PANEL task waits for after its pre-work, or -1
        float pre, newest;     // PANEL: work on the older blocks / on the newest block, both before W is needed
        float dur;             // W there (or start) -> workgroup free
        float blevel;
      };
      struct Need { int flag, cls; };          // cls 0: needed at the start; 1: the newest block's X (after `pre`)
      struct Prod { int flag; float off; };    // set `off` after W arrived (or after the start)
      std::vector<Gen> gen;
      std::vector<Need> need;
      std::vector<Prod> prod;
      constexpr int TS = 64;           // edge of a trailing-update tile (k_big_flow<T, 2>)
      words = lvl->ticket_word + 32;   // the ticket on a 128-byte line of its own
      const int64_t flag0 = words;
      for (int z = 0; z < nf; z++) {
        const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
        const int nc = sym_.sn_ncols[sn], M = nc + sym_.sn_nrows[sn] + 1;
        const int nblk = (nc + BIG_NB - 1) / BIG_NB, nsp = (nc + BIG_SUPER - 1) / BIG_SUPER;
        const int pstride = (M + 31) / 32 + 1;
        const int ntmax = (M + TS - 1) / TS + 1, ustride = ntmax * (ntmax + 1) / 2;
        FlowFront &ff = fronts[z];
        ff.wf = (int32_t)words; words += nblk + 1;
        ff.pf = (int32_t)words; ff.pstride = pstride; words += (int64_t)nblk * pstride;
        ff.uf = (int32_t)words; ff.ustride = ustride; words += (int64_t)nsp * ustride;
        ff.bf = -1; ff.nbuild = 0; ff.done = -1;
        if (xl_) {   // build counter + "built" flag, and the counter of finished UPDATE tasks, on a line of their own
          words = (words + 31) & ~(int64_t)31;
          ff.bf = (int32_t)words; ff.done = (int32_t)words + 2; words += 3;
          ff.nbuild = (big_built_cols(nc, M) + FLOW_BUILD_COLS - 1) / FLOW_BUILD_COLS;
        }
        words = (words + 31) & ~(int64_t)31;
        if (words > 0x7fffff00LL) throw ApiError(RR_PGO_EUNSUPPORTED, "too many flow flags");
        const int wf = (int)(ff.wf - flag0), pf = (int)(ff.pf - flag0), uf = (int)(ff.uf - flag0);
        auto tri = [](int bx, int by) { return bx * (bx + 1) / 2 + by; };
        auto open_task = [&](FlowTask t) {
          gen.push_back(Gen{t, (int)need.size(), 0, (int)prod.size(), 0, -1, 0.f, 0.f, 0.f, 0.f});
        };
        auto close_task = [&]() { gen.back().need1 = (int)need.size(); gen.back().prod1 = (int)prod.size(); };
        open_task(FlowTask{(FLOW_DIAG0 << 24) | z, 0, 0, 0});
        prod.push_back(Prod{wf + 0, (float)kDiag});
        gen.back().dur = (float)kDiag;
        close_task();
        const int o_schur = lvl->schur_split ? std::min(big_schur_origin(nc, schur_tile_), M) : M;   // UPDATE tiles start left of it
        // fast mode: the first two blocks of a super-panel look back over the previous super-panel too, and that one's
        // update skips these 64 columns (its tile column 0) -- the chain does not wait for a tile at a super-panel's end
        auto deep = [&](int spx) { return flow_deep_ && !flow_exact_ && spx > 0 && spx * BIG_SUPER + 64 <= nc; };
        for (int sp = 0; sp < nsp; sp++) {
          const int K0 = sp * BIG_SUPER, ke = std::min(K0 + BIG_SUPER, nc);
          // the tiles under rows [r0, r1] x columns [c0, c1] of the update whose tile grid starts at column Ko (the update
          // before a left-looking range that starts there); tiles right of the Schur origin are not this launch's
          auto prev_tiles = [&](int Ko, int r0, int r1, int c0, int c1, int cls) {
            if (Ko <= 0) return;
            const int spo = Ko / BIG_SUPER - 1;
            for (int bx = (r0 - Ko) / TS; bx <= (std::min(r1, M - 1) - Ko) / TS; bx++)
              for (int by = (c0 - Ko) / TS; by <= std::min((std::min(c1, M - 1) - Ko) / TS, bx); by++)
                if (Ko + TS * by < o_schur && !(by == 0 && deep(spo + 1))) need.push_back(Need{uf + spo * ustride + tri(bx, by), cls});
          };
          for (int kb = K0; kb < ke; kb += BIG_NB) {
            const int blk = kb / BIG_NB, qb = (kb - K0) / BIG_NB, nb = std::min(BIG_NB, nc - kb), kn = kb + BIG_NB;
            const int Ks = deep(sp) && qb < 2 ? K0 - BIG_SUPER : K0;   // first column of the step's left-looking range
            const int q = (kb - Ks) / BIG_NB;
            const int skip = deep(sp) && qb == 1 ? BIG_SUPER / BIG_NB : 0;   // the NEXT diagonal block's range starts that many blocks later
            const int nrb = (M - (kb + nb) + 31) / 32;
            for (int g0 = 0; g0 < nrb; g0 += FLOW_GROUP) {
              open_task(FlowTask{(FLOW_PANEL << 24) | z, kb, g0, Ks | (skip << 24)});
              Gen &g = gen.back();
              g.wflag = wf + blk;
              g.pre = (float)(q > 1 ? kPre * (q - 1) * 0.5 : 0.5);
              g.newest = (float)(q > 0 ? kNewest : 0.0);
              g.dur = (float)kX;
              for (int w = 0; w < FLOW_GROUP && g0 + w < nrb; w++) {
                const int rb = g0 + w, R0 = kb + nb + 32 * rb;
                const bool look = rb == 0 && kn < (flow_exact_ ? ke : nc);
                // a wave starts with everything that needs neither W nor the newest block (the older blocks and the C
                // tiles), then the newest block's term, then W, then X; the look wave goes on with the next block
                for (int j = 1; j <= q; j++) {
                  const int r0p = kb - 32 * j + 32;
                  need.push_back(Need{pf + (blk - j) * pstride + (j - 1), j == 1});
                  for (int rbp = (R0 - r0p) / 32; rbp <= (std::min(R0 + 31, M - 1) - r0p) / 32; rbp++)
                    need.push_back(Need{pf + (blk - j) * pstride + rbp, j == 1});
                }
                prev_tiles(Ks, R0, R0 + 31, kb, kb + nb - 1, 0);
                if (look) prev_tiles(Ks + BIG_NB * skip, kn, kn + 31, kn, kn + 31, 0);
                prod.push_back(Prod{pf + blk * pstride + rb, (float)kX});   // X is published before the look wave goes on
                if (look) { prod.push_back(Prod{wf + blk + 1, (float)(kX + kLook)}); g.dur = (float)(kX + kLook); }
              }
              close_task();
            }
          }
          // trailing update of the super-panel: tiles of rows / columns >= ke
          const int t0 = ke;
          const int nt = (M - t0 + TS - 1) / TS;
          for (int by = deep(sp + 1) ? 1 : 0; by < nt; by++)       // column-major: the next super-panel's own columns first
            for (int bx = by; bx < nt; bx++) {
              const int I0 = t0 + TS * bx, J0 = t0 + TS * by;
              if (J0 >= o_schur) continue;       // the Schur complement: one k_big_schur launch behind the flow launch
              open_task(FlowTask{(FLOW_UPDATE << 24) | z, K0, bx, by});
              for (int kb = K0; kb < ke; kb += BIG_NB) {
                const int nbq = std::min(BIG_NB, nc - kb), r0 = kb + nbq;
                for (int strip = 0; strip < 2; strip++) {
                  const int lo = strip ? J0 : I0, hi = std::min(lo + TS - 1, M - 1);
                  for (int rb = (lo - r0) / 32; rb <= (hi - r0) / 32; rb++) need.push_back(Need{pf + (kb / BIG_NB) * pstride + rb, 0});
                }
              }
              prev_tiles(K0, I0, I0 + TS - 1, J0, J0 + TS - 1, 0);
              prod.push_back(Prod{uf + sp * ustride + tri(bx, by), (float)kTile});
              gen.back().dur = (float)kTile;
              if (flow_exact_ && bx == 0 && by == 0 && t0 < nc) {
                prod.push_back(Prod{wf + t0 / BIG_NB, (float)(kTile + kTail)});
                gen.back().dur = (float)(kTile + kTail);
              }
              close_task();
            }
        }
      }
      const int n_gen = (int)gen.size();
      if (n_gen > (xl_ ? std::min(flow_max_tasks_, xl_level_tasks_) : flow_max_tasks_)) { too_many = true; return; }
      const int n_flags = (int)(words - flag0);
      // who sets a flag, who waits for it
      std::vector<int> producer(n_flags, -1), cons_ptr(n_flags + 1, 0), cons;
      for (int i = 0; i < n_gen; i++)
        for (int k = gen[i].prod0; k < gen[i].prod1; k++) producer[prod[k].flag] = i;
      auto for_each_need = [&](int i, auto &&fn) {   // (flag, lead): lead = how long after the task's start the flag is needed
        const Gen &g = gen[i];
        for (int k = g.need0; k < g.need1; k++) fn(need[k].flag, need[k].cls ? g.pre : 0.f);
        if (g.wflag >= 0) fn(g.wflag, g.pre + g.newest);
      };
      for (int i = 0; i < n_gen; i++) for_each_need(i, [&](int f, float) { cons_ptr[f + 1]++; });
      for (int f = 0; f < n_flags; f++) cons_ptr[f + 1] += cons_ptr[f];
      cons.resize(cons_ptr[n_flags]);
      {
        std::vector<int> fill(cons_ptr.begin(), cons_ptr.end() - 1);
        for (int i = 0; i < n_gen; i++) for_each_need(i, [&](int f, float) { cons[fill[f]++] = i; });
      }
      // longest remaining path of every task (its consumers come later in the natural order)
      {
        std::vector<float> after(n_flags, 0.f);   // per flag: the longest path that starts when it is set
        for (int i = n_gen - 1; i >= 0; i--) {
          Gen &g = gen[i];
          const float lead = g.pre + g.newest;
          float bl = lead + g.dur;
          for (int k = g.prod0; k < g.prod1; k++) bl = std::max(bl, lead + prod[k].off + after[prod[k].flag]);
          g.blevel = bl;
          for_each_need(i, [&](int f, float ld) { after[f] = std::max(after[f], (float)kHop + bl - ld); });
        }
      }
      // the list schedule
      tasks.reserve(n_gen);
      {
        // half the launch's workgroups: the model's tiles are cheaper than loaded ones, and a chain step drawn a little
        // early only waits, while one drawn late stalls its front (measured: 128 ... 256 slots equal, 512 1 % slower)
        int P = std::max(1, flow_grid_ / 2);
        std::vector<float> ftime(n_flags, 0.f);
        std::vector<int> pending(n_gen, 0);
        for (int i = 0; i < n_gen; i++) for_each_need(i, [&](int f, float) { if (producer[f] >= 0) pending[i]++; });
        using Fut = std::pair<float, int>;                 // (earliest useful start, task)
        std::priority_queue<Fut, std::vector<Fut>, std::greater<Fut>> future;
        auto less_urgent = [&](int x, int y) { return gen[x].blevel != gen[y].blevel ? gen[x].blevel < gen[y].blevel : x > y; };
        std::priority_queue<int, std::vector<int>, decltype(less_urgent)> ready(less_urgent);
        std::priority_queue<float, std::vector<float>, std::greater<float>> slots;
        for (int k = 0; k < P; k++) slots.push(0.f);
        auto becomes_known = [&](int i) {
          float r = 0.f;
          for_each_need(i, [&](int f, float ld) { if (producer[f] >= 0) r = std::max(r, ftime[f] + (float)kHop - ld); });
          future.push(Fut{r, i});
        };
        for (int i = 0; i < n_gen; i++) if (pending[i] == 0) becomes_known(i);
        float now = 0.f;
        for (int done = 0; done < n_gen; done++) {
          now = std::max(now, slots.top());
          while (!future.empty() && future.top().first <= now) { ready.push(future.top().second); future.pop(); }
          if (ready.empty()) {
            if (future.empty()) throw ApiError(RR_PGO_EUNSUPPORTED, "flow task graph is not schedulable");
            now = future.top().first;
            while (!future.empty() && future.top().first <= now) { ready.push(future.top().second); future.pop(); }
          }
          const int i = ready.top();
          ready.pop();
          slots.pop();
          const Gen &g = gen[i];
          float t_new = 0.f, t_w = 0.f;
          for (int k = g.need0; k < g.need1; k++)
            if (need[k].cls && producer[need[k].flag] >= 0) t_new = std::max(t_new, ftime[need[k].flag] + (float)kHop);
          if (g.wflag >= 0 && producer[g.wflag] >= 0) t_w = ftime[g.wflag] + (float)kHop;
          const float go = std::max(std::max(now + g.pre, t_new) + g.newest, t_w);
          slots.push(go + g.dur);
          level_end = std::max(level_end, (double)(go + g.dur));
          busy_us += g.pre + g.newest + g.dur;
          tasks.push_back(g.t);
          for (int k = g.prod0; k < g.prod1; k++) {
            const int f = prod[k].flag;
            ftime[f] = go + prod[k].off;
            for (int c = cons_ptr[f]; c < cons_ptr[f + 1]; c++)
              if (--pending[cons[c]] == 0) becomes_known(cons[c]);
          }
        }
      }
      };   // plan
      plan();
      if (!too_many && !xl_ && lvl->schur_split && busy_us + schur_in_flow_us <= 0.6 * flow_grid_ * level_end) {
        lvl->schur_split = false;
        plan();
        if (too_many) { too_many = false; lvl->schur_split = true; plan(); }
      }
      if (too_many) { words = lvl->ticket_word; continue; }   // throughput-bound level: launch sequence (its flag words are given back)
      if (xl_) {
        // BUILD tasks ahead of the level's other tasks, chunk by chunk over all fronts (every front's first columns first)
        std::vector<FlowTask> bt;
        int max_nb = 0;
        for (int z = 0; z < nf; z++) max_nb = std::max(max_nb, fronts[z].nbuild);
        for (int k = 0; k < max_nb; k++)
          for (int z = 0; z < nf; z++)
            if (k < fronts[z].nbuild) bt.push_back(FlowTask{(FLOW_BUILD << 24) | z, k * FLOW_BUILD_COLS, 0, 0});
        bt.insert(bt.end(), tasks.begin(), tasks.end());
        tasks.swap(bt);
      }
      lvl->n_tasks = (int)tasks.size();
      lvl->est_us = level_end;
      lvl->wfill_begin = (int)(wfill.size() / 2);
      for (int z = 0; z < nf; z++) {
        const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
        const int64_t w0 = sn_wblk[sn] * 256, wn = (int64_t)((sym_.sn_ncols[sn] + BIG_NB - 1) / BIG_NB) * 1024;
        for (int64_t o = 0; o < wn; o += 4096) { wfill.push_back(w0 + o); wfill.push_back(std::min<int64_t>(4096, wn - o)); }   // one workgroup of k_flow_reset each
      }
      lvl->wfill_end = (int)(wfill.size() / 2);
      all_tasks[si] = std::move(tasks);
      all_fronts[si] = std::move(fronts);
      flow_levels_[si] = std::move(lvl);
    }
    if (xl_) {
      // the cross-level form needs a task list for EVERY level of fronts beyond LDS (with all Schur tiles as UPDATE tasks a
      // level has several times the tasks of its per-level plan): a level that went over the limit takes the whole graph back to
      // one build + one flow launch per level, planned afresh with the per-level Schur split -- never a refused handle
      bool complete = true;
      for (size_t si = 0; si < sym_.steps.size(); si++) complete = complete && (sym_.steps[si].kind != STEP_BIG || flow_levels_[si]);
      if (!complete) {
        xl_refused_ = true;
        build_flow_levels();
        return;
      }
    }
    // k_big_solve_flow (back substitution of wide pivot blocks as one launch per level): a ticket and counters per front in
    // the same zeroed block, and the level's task list: by step, the chain tasks first, then the folds, the groups nearest
    // to the current super-panel first (the next chain step waits for those)
    std::vector<std::vector<SolveFlowFront>> all_sf(sym_.steps.size());
    std::vector<std::vector<SolveFlowTask>> all_st(sym_.steps.size());
    if (solve_flow_on_)
      for (size_t si = 0; si < sym_.steps.size(); si++) {
        const Step &st = sym_.steps[si];
        if (st.kind != STEP_BIG) continue;
        const int nf = st.task_end - st.task_begin;
        int max_nc = 1;
        for (int z = 0; z < nf; z++) max_nc = std::max(max_nc, sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]]);
        if (max_nc < sp_solve_min_nc_) continue;
        solve_flow_[si] = std::make_unique<SolveFlowLevel>();
        solve_flow_[si]->ticket_word = words;
        words += 32;
        const int max_S = (max_nc + BIG_SUPER - 1) / BIG_SUPER;
        for (int z = 0; z < nf; z++) {
          const int nc = sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]];
          all_sf[si].push_back(SolveFlowFront{(int32_t)words, (int32_t)(words + 1)});
          words += 1 + (nc + 63) / 64;
          words = (words + 31) & ~(int64_t)31;
        }
        for (int ell = 0; ell < max_S; ell++) {
          for (int z = 0; z < nf; z++) {
            const int nc = sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]];
            if (ell < (nc + BIG_SUPER - 1) / BIG_SUPER) all_st[si].push_back(SolveFlowTask{z, ell, -1});
          }
          for (int z = 0; z < nf; z++) {
            const int nc = sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]];
            const int n_sp = (nc + BIG_SUPER - 1) / BIG_SUPER;
            if (ell >= n_sp) continue;
            const int K0 = BIG_SUPER * (n_sp - 1 - ell);
            for (int g = (K0 + 63) / 64 - 1; g >= 0; g--) all_st[si].push_back(SolveFlowTask{z, ell, g});   // columns [64 g, ...) left of K0
          }
        }
        solve_flow_[si]->n_tasks = (int)all_st[si].size();
      }
    if (words == 0) return;
    wfill.resize(wfill.size() + 2, 0);   // never empty
    flow_wfill_.upload(wfill);
    flow_flags_.alloc((size_t)words + kDeadFlagWords);
    flow_flags_.zero();
    for (size_t si = 0; si < sym_.steps.size(); si++)
      if (solve_flow_[si]) {
        solve_flow_[si]->fronts.upload(all_sf[si]);
        solve_flow_[si]->tasks.upload(all_st[si]);
      }
    for (size_t si = 0; si < sym_.steps.size(); si++)
      if (flow_levels_[si]) {
        {
          const Step &st = sym_.steps[si];
          std::vector<FlowRec> recs(all_tasks[si].size());
          for (size_t i = 0; i < recs.size(); i++) {
            const int slot = all_tasks[si][i].kind_front & 0xffffff;
            recs[i].t = all_tasks[si][i];
            recs[i].ff = all_fronts[si][slot];
            recs[i].m = host_task_meta_[st.task_begin + slot];
            std::memset(recs[i].pad, 0, sizeof(recs[i].pad));
          }
          flow_levels_[si]->tasks.upload(recs);
        }
#ifdef RRPGO_FLOW_TRACE
        flow_levels_[si]->host_tasks = all_tasks[si];
        flow_levels_[si]->trace.alloc(all_tasks[si].size() * 16);
        flow_levels_[si]->trace.zero();
#endif
      }
    if (xl_) {
      // the cross-level list: every level's records, level after level (children have smaller tickets than their parents), and
      // per child_meta entry where the child's `done` counter lives and how many UPDATE tasks it counts
      std::vector<int32_t> sn_done(sym_.S, -1), sn_nupd(sym_.S, 0);
      for (size_t si = 0; si < sym_.steps.size(); si++) {
        if (!flow_levels_[si]) { if (sym_.steps[si].kind == STEP_BIG) xl_ = false; continue; }
        const Step &st = sym_.steps[si];
        for (const FlowTask &t : all_tasks[si]) {
          const int slot = t.kind_front & 0xffffff, sn = sym_.task_sn[sym_.task_ptr[st.task_begin + slot]];
          sn_done[sn] = all_fronts[si][slot].done;
          if ((t.kind_front >> 24) == FLOW_UPDATE) sn_nupd[sn]++;
        }
      }
      if (!xl_) throw ApiError(RR_PGO_EUNSUPPORTED, "internal: a level of fronts beyond LDS without a dataflow task list in the cross-level form");
      std::vector<int2> cd(sym_.child_list.size(), int2{-1, 0});
      for (size_t q = 0; q < cd.size(); q++) {
        const int c = sym_.child_list[q];
        if (sn_done[c] >= 0) cd[q] = int2{sn_done[c], sn_nupd[c]};
      }
      xl_child_done_.upload(cd);
      bool first = true;
      for (size_t si = 0; si < sym_.steps.size(); si++) {
        if (!flow_levels_[si]) continue;
        if (first) { xl_ticket_word_ = flow_levels_[si]->ticket_word; first = false; }
        const Step &st = sym_.steps[si];
        for (const FlowTask &t : all_tasks[si]) {
          const int slot = t.kind_front & 0xffffff;
          FlowRec r{};
          r.t = t;
          r.ff = all_fronts[si][slot];
          r.m = host_task_meta_[st.task_begin + slot];
          xl_host_.push_back(r);
        }
      }
      xl_recs_.upload(xl_host_);
#ifdef RRPGO_FLOW_TRACE
      xl_trace_.alloc(xl_host_.size() * 16);
      xl_trace_.zero();
#endif
    }
  }
  // k_big_update's and k_big_schur's grids: per level of the launch sequence the list of real tiles, front after front
  // -- per super-panel the tiles left of the Schur origin (schur_split_: the Schur complement is ONE pass of k_big_schur
  // at the end of the level; else every tile right of the super-panel), and the Schur tiles themselves
  void build_update_maps() {
    upd_maps_.assign(sym_.steps.size(), {});
    schur_maps_.assign(sym_.steps.size(), UpdMap{0, 0});
    std::vector<int32_t> buf;
    for (size_t si = 0; si < sym_.steps.size(); si++) {
      const Step &st = sym_.steps[si];
      if (st.kind != STEP_BIG) continue;
      const bool flow_level = (bool)flow_levels_[si];   // its per-super-panel tiles are UPDATE tasks; only the Schur pass is a launch
      const int nf = st.task_end - st.task_begin;
      if (nf > 0xffff) throw ApiError(RR_PGO_EUNSUPPORTED, "more than 65535 big fronts in one level");
      int max_nc = 0;
      for (int z = 0; z < nf; z++) max_nc = std::max(max_nc, sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]]);
      auto tri_ok = [](int nt) { if (nt * (nt + 1) / 2 > 0xffff) throw ApiError(RR_PGO_EUNSUPPORTED, "a front of more than 65535 update tiles"); };
      for (int K0 = 0; K0 < max_nc && !flow_level; K0 += BIG_SUPER) {
        UpdMap um{(int64_t)buf.size(), 0};
        for (int z = 0; z < nf; z++) {
          const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
          const int nc = sym_.sn_ncols[sn], M = nc + sym_.sn_nrows[sn] + 1;
          if (K0 >= nc) continue;
          const int t0 = std::min(K0 + BIG_SUPER, nc), nt = (M - t0 + 63) / 64;
          tri_ok(nt);
          const int o = schur_split_ ? std::min(big_schur_origin(nc, schur_tile_), M) : M;   // tiles whose first column is left of it
          for (int bx = 0; bx < nt; bx++)
            for (int by = 0; by <= bx; by++)
              if (t0 + 64 * by < o) buf.push_back((z << 16) | (bx * (bx + 1) / 2 + by));
        }
        um.n_tiles = (int)((int64_t)buf.size() - um.offset);
        upd_maps_[si].push_back(um);
      }
      if (flow_level ? flow_levels_[si]->schur_split : schur_split_) {
        UpdMap um{(int64_t)buf.size(), 0};
        for (int z = 0; z < nf; z++) {
          const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
          const int nc = sym_.sn_ncols[sn], M = nc + sym_.sn_nrows[sn] + 1;
          const int o = big_schur_origin(nc, schur_tile_);
          if (o >= M) continue;
          const int nt = (M - o + schur_tile_ - 1) / schur_tile_;
          tri_ok(nt);
          for (int t = 0; t < nt * (nt + 1) / 2; t++) buf.push_back((z << 16) | t);
        }
        um.n_tiles = (int)((int64_t)buf.size() - um.offset);
        schur_maps_[si] = um;
      }
    }
    if (!buf.empty()) upd_map_buf_.upload(buf);
  }
  const UpdMap &upd_map(const Step &st, int sp) const { return upd_maps_[(size_t)(&st - sym_.steps.data())][(size_t)sp]; }
  bool any_flow(size_t from, size_t to) const {   // a factorisation range that must zero the flag block first
    for (size_t si = from; si < to && si < flow_levels_.size(); si++) if (flow_levels_[si]) return true;
    for (const auto &p : solve_flow_) if (p) return true;   // the back substitution of ANY level may use its counters afterwards
    return false;
  }
  const FlowLevel *flow_of(const Step &st) const {
    const size_t si = (size_t)(&st - sym_.steps.data());
    return si < flow_levels_.size() ? flow_levels_[si].get() : nullptr;
  }
  FlowArgs<T> flow_args() {
    FlowArgs<T> fa;
    fa.flags = flow_flags_.p;
    fa.gather = gather_update_ ? 1 : 0;
    fa.exact = flow_exact_ ? 1 : 0;
    fa.child_meta = child_meta_.p;
    fa.scat = scat_.p;
    fa.lvals = lvals_.p;
    fa.uvals = uvals_.p;
    fa.xch = xch_;
    fa.winv = winv_.p;
    fa.xnew = xnew_.p;
    fa.err = err_.p;
    fa.wait_ticks = wait_ticks_;
    fa.trace = nullptr;
    fa.child_done = xl_child_done_.p;
    fa.fasm_colptr = fasm_colptr_.p;
    fa.fasm_dst = fasm_dst_.p;
    fa.fasm_src = fasm_src_.p;
    fa.perm = perm_.p;
    fa.hvals = hvals_.p;
    fa.b = b_.p;
    return fa;
  }
  void launch_flow(const Step &st, const FlowLevel &lvl) {
    FlowArgs<T> fa = flow_args();
    fa.tasks = lvl.tasks.p;
    fa.ticket = flow_flags_.p + lvl.ticket_word;
    fa.n_tasks = lvl.n_tasks;
    fa.schur_tile = lvl.schur_split ? schur_tile_ : 0;
    fa.trace = lvl.trace.p;
    hipLaunchKernelGGL((k_big_flow<T, 2>), dim3((unsigned)std::min(lvl.n_tasks, flow_grid_)), dim3(256), 0, stream_, fa);
    check_launch("k_big_flow");
  }
  // every level of fronts beyond LDS as ONE launch (the cross-level form: BUILD tasks, Schur complements inside, counters between levels)
  void launch_flow_xl() {
    FlowArgs<T> fa = flow_args();
    fa.tasks = xl_recs_.p;
    fa.ticket = flow_flags_.p + xl_ticket_word_;
    fa.n_tasks = (int)xl_host_.size();
    fa.schur_tile = 0;
    fa.trace = xl_trace_.p;
    hipLaunchKernelGGL((k_big_flow<T, 2, true>), dim3((unsigned)std::min(fa.n_tasks, flow_grid_)), dim3(256), 0, stream_, fa);
    check_launch("k_big_flow (cross-level)");
  }

  // Gauge transfer applies to a single-precision factor (T = float) of an SE(2) graph whose root front is
  // one of the big in-place fronts; RR_PGO_GAUGE=0 keeps the reference's anchor prior (comparison runs).
  void setup_gauge() {
    if (sizeof(T) != 4 || is3d_ || g_.anchor_node < 0 || sym_.S == 0) return;
    if (const char *e = getenv("RR_PGO_GAUGE")) if (std::atoi(e) == 0) return;
    const int root = sym_.S - 1;
    if (!sym_.sn_big[root] || !sym_.sn_huge[root] || sym_.sn_nrows[root] != 0) return;
    std::vector<int32_t> cn;
    double sx = 0, sy = 0, ext = 0;
    int n_se2 = 0, n_nodes = 0;
    for (int p = sym_.sn_first_pos[root]; p < sym_.sn_first_pos[root] + sym_.sn_npos[root]; p++) {
      const int node = sym_.order[p];
      const double *st = &g_.node_state[g_.node_state_off[node]];
      sx += st[0]; sy += st[1]; n_nodes++;
      n_se2 += g_.node_kind[node] == NODE_SE2;
      for (int t = 0; t < node_dim(g_.node_kind[node]); t++) cn.push_back((node << 2) | t);
    }
    if (n_se2 == 0 || (int)cn.size() != sym_.sn_ncols[root]) return;
    gauge_ox_ = sx / n_nodes; gauge_oy_ = sy / n_nodes;
    for (int p = sym_.sn_first_pos[root]; p < sym_.sn_first_pos[root] + sym_.sn_npos[root]; p++) {
      const double *st = &g_.node_state[g_.node_state_off[sym_.order[p]]];
      ext = std::max(ext, std::max(std::fabs(st[0] - gauge_ox_), std::fabs(st[1] - gauge_oy_)));
    }
    // entries of the added term stay at the scale of the information matrices: mu * max|v|^2 = w
    double w = 0;
    for (int k = 0; k < g_.n_edges(); k++) {
      const double *wi = &g_.edge_info[g_.edge_info_off[k]];
      w = std::max(w, std::fabs(wi[0]));
    }
    gauge_mu_t_ = w;
    gauge_mu_r_ = w / std::max(1.0, ext * ext);
    gauge_root_ = root;
    gauge_col_node_.upload(cn);
    gauge_v_.alloc(3 * cn.size());
    gauge_ok_ = true;
  }
  void launch_gauge_term() {
    GaugeArgs<T, S> ga;
    ga.pose = pose_.p;
    ga.col_node = gauge_col_node_.p;
    ga.nc = sym_.sn_ncols[gauge_root_];
    ga.M = ga.nc + sym_.sn_nrows[gauge_root_] + 1;
    ga.F = lvals_.p + sym_.sn_loff[gauge_root_];
    ga.v = gauge_v_.p;
    ga.ox = (S)gauge_ox_; ga.oy = (S)gauge_oy_;
    ga.mu_t = (T)gauge_mu_t_; ga.mu_r = (T)gauge_mu_r_;
    hipLaunchKernelGGL((k_gauge_vectors<T, S>), dim3((ga.nc + 255) / 256), dim3(256), 0, stream_, ga);
    hipLaunchKernelGGL((k_big_gauge<T, S>), dim3(ga.nc), dim3(256), 0, stream_, ga);
    check_launch("k_big_gauge");
  }

  static constexpr int kMaxLds = 160 * 1024 - 9216;   // the kernels also hold static __shared__ scratch (W16_SCR: a W slot + identity, tickets)

  template <int TH> void set_lds_attr() {
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_tasks<T, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_solve_tasks<T, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
  }
  int level_max_nc(const Step &st) const {
    int max_nc = 1;
    for (int t = st.task_begin; t < st.task_end; t++) max_nc = std::max(max_nc, sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[t]]]);
    return max_nc;
  }
  // row slices of a level's k_big_gemv_partial launch: enough workgroups to fill the chip, but never thinner than 128 rows
  // (every slice is a partial sum the solve has to fetch and add per column)
  int gemv_slices(const Step &st) const {
    const int nf = st.task_end - st.task_begin, cb = (level_max_nc(st) + 63) / 64;
    int max_nr = 1;
    for (int t = st.task_begin; t < st.task_end; t++) max_nr = std::max(max_nr, sym_.sn_nrows[sym_.task_sn[sym_.task_ptr[t]]]);
    return std::max(1, std::min(std::min(kGemvSlices, (max_nr + 127) / 128), (768 + nf * cb - 1) / (nf * cb)));
  }
  // Graphs on the dataflow schedule with fronts beyond LDS (sphere2500: 46, six levels): the levels whose pivot blocks are
  // narrower than RR_PGO_SP_SOLVE_MIN columns -- all but the top one or two -- are back-substituted as TASKS of k_solve_flow
  // (lds_flow.hip.h: GEMV units + one L11 task per front) instead of a k_big_gemv_partial + k_solve_mid launch pair per level.
  // The wide top levels stay launches in front of k_solve_flow.  RR_PGO_SOLVE_MID_FLOW=0: every level as launches (the
  // r03 - r05 form; the same sums in the same order: bit-identical).
  void plan_mid_solve_tasks() {
    mid_in_flow_.assign(sym_.steps.size(), 0);
    if (!lds_flow_) return;
    bool on = true;
    if (const char *e = getenv("RR_PGO_SOLVE_MID_FLOW")) on = std::atoi(e) != 0;
    int top = (int)sym_.steps.size() - 1;
    while (on && top >= 1 && level_max_nc(sym_.steps[top]) >= sp_solve_min_nc_) top--;   // the wide top levels
    std::vector<LdsFlowTask> mid;
    int need = 1024;   // a GEMV unit's x[rows] chunk
    for (int i = top; on && i >= 1; i--) {
      const Step &st = sym_.steps[i];
      if (level_max_nc(st) >= sp_solve_min_nc_) { on = false; break; }   // a wide level under a narrow one: keep every level a launch
      const int R = gemv_slices(st);
      for (int pass = 1; pass <= 2; pass++)   // the level's GEMV units first, then its L11 tasks
        for (int t = st.task_begin; t < st.task_end; t++) {
          LdsFlowTask r{};
          r.sn_begin = sym_.task_ptr[t];
          r.sn_end = r.sn_begin + 1;
          r.sn = sym_.task_sn[r.sn_begin];
          r.m = host_sn_meta_[r.sn];
          r.kind = pass;
          r.slices = R;
          const int nc = sym_.sn_ncols[r.sn], nr = sym_.sn_nrows[r.sn];
          const int units = nr > 0 ? (nc + 63) / 64 * R : 0;
          if (pass == 2) {
            r.n_units = units;
            mid.push_back(r);
            need = std::max(need, ((nc + 3) & ~3) + 2 * 32 * 33 + 64);
          } else {
            for (int u = 0; u < units; u++) { r.bx = u / R; r.by = u % R; mid.push_back(r); }
          }
        }
    }
    if (on && !mid.empty() && (size_t)need * sizeof(T) <= (size_t)kMaxLds) {
      for (int i = top; i >= 1; i--) mid_in_flow_[i] = 1;
      step_solve_lds_[0] = std::max(step_solve_lds_[0], need);
      mid.insert(mid.end(), host_stasks_.begin(), host_stasks_.end());
      host_stasks_.swap(mid);
    }
    lds_n_stasks_ = (int)host_stasks_.size();
    lds_stasks_.upload(host_stasks_);
  }

  void configure_kernels() {
    // opt in to > 64 KiB of dynamic LDS
    set_lds_attr<64>();
    set_lds_attr<128>();
    set_lds_attr<256>();
    set_lds_attr<512>();
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_tasks<T, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_solve_mid<T, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_flow<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_flow<T, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_flow<T, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_solve_flow<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
    HIPCHK(hipFuncSetAttribute((const void *)k_solve_flow<T, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
  }

  template <int TH> void launch_factor_tasks(int nt, size_t lds, const FactorArgs<T> &a) {
    hipLaunchKernelGGL((k_factor_tasks<T, TH>), dim3(nt), dim3(TH), lds, stream_, a);
  }
  template <int TH> void launch_solve_tasks(int nt, size_t lds, const FactorArgs<T> &a) {
    hipLaunchKernelGGL((k_solve_tasks<T, TH>), dim3(nt), dim3(TH), lds, stream_, a);
  }

  LinArgs<T, S> lin_args(double lambda, int lm, int write_system, bool reference_prior = false) {
    // Gauss-Newton with a single-precision factor: no anchor prior, the root front carries the gauge term
    if (write_system) gauge_now_ = gauge_ok_ && !lm && !reference_prior;
    LinArgs<T, S> a;
    a.n_nodes = n_list_;
    a.node_list = node_list_.p;
    a.pose = pose_.p;
    a.e_rec = e_rec_.p;
    a.e_idx = e_idx_.p;
    a.e_meas = e_meas_.p;
    a.e_info_a = e_info_a_.p;
    a.e_info_b = e_info_b_.p;
    a.e_slot = e_slot_.p;
    a.inc_ptr = inc_ptr_.p;
    a.inc_list = inc_list_.p;
    a.node_dim = node_dim_.p;
    a.node_offset = node_offset_.p;
    a.diag_off = diag_off_.p;
    a.hvals = hvals_.p;
    a.b = b_.p;
    a.chi2_partial = chi_partial_.p;
    a.anchor = (write_system && gauge_now_) ? -1 : g_.anchor_node;
    a.lambda = lm ? (S)lambda : (S)0;
    a.write_system = write_system;
    a.adds_diag = norm_counts_.p;
    a.zero_words = lds_flow_ ? dep_flags_.p : nullptr;
    a.n_zero_words = lds_flow_ ? 2 * sym_.S + 64 : 0;
    a.fill_words = lds_flow_ ? reinterpret_cast<unsigned *>(x_ptr_) : nullptr;
    a.n_fill_words = lds_flow_ ? (int)((size_t)g_.dim * sizeof(T) / 4) : 0;
    opt_lin_fields(a, lm);
    return a;
  }
  // what a linearisation inside a pipelined rr_pgo_optimize call carries: the reset of the loop state (first launch of
  // the call), lambda from the device (Levenberg-Marquardt)
  template <typename A> void opt_lin_fields(A &a, int lm) {
    a.ctrl = opt_active_ ? opt_ctrl_.p : nullptr;
    a.err = err_.p;
    a.lambda_from_ctrl = (opt_active_ && opt_lambda_dev_ && lm) ? 1 : 0;
    a.reset_ctrl = (opt_active_ && opt_first_) ? 1 : 0;
    a.reset_lambda = opt_lambda0_;
    a.reset_tolerance = 1e-4;   // :253
    a.publish = opt_active_ ? opt_publish_ : 0;
    a.ring_host = opt_ring_;
    a.blocks_done = blocks_done_.p + 1;
    opt_first_ = false;
  }

  FactorArgs<T> factor_args(int task_begin) {
    FactorArgs<T> a;
    a.task_ptr = task_ptr_.p;
    a.task_sn = task_sn_.p;
    a.task_begin = task_begin;
    a.sn_meta = sn_meta_.p;
    a.task_meta = task_meta_.p;
    a.child_meta = child_meta_.p;
    a.fasm_src = fasm_src_.p;
    a.fasm_dst = fasm_dst_.p;
    a.fasm_colptr = fasm_colptr_.p;
    a.fdup_src = fdup_src_.p;
    a.fdup_dst = fdup_dst_.p;
    a.scat = scat_.p;
    a.rel = rel_.p;
    a.perm = perm_.p;
    a.sn_rows = sn_rows_.p;
    a.hvals = hvals_.p;
    a.b = b_.p;
    a.lvals = lvals_.p;
    a.uvals = uvals_.p;
    a.xch = xch_;
    a.x = x_ptr_;
    a.winv = winv_.p;
    a.err = err_.p;
    a.stamps = stamps_.p;
#if defined(RRPGO_TRACE) && defined(RRPGO_STAMPS)
    a.trace = stamps_.p ? stamps_.p + (size_t)sym_.S * 12 : nullptr;
#else
    a.trace = nullptr;
#endif
    a.child_dep = child_dep_.p;
    a.parent_dep = parent_dep_.p;
    a.dep_flags = dep_flags_.p;
    a.parent_dep_self = sym_.S;
    a.wait_ticks = wait_ticks_;
    a.solve_lds = 0;
    return a;
  }

  // Optional per-launch event timing (rr_pgo_profile).
  struct Prof {
    bool on = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double ms[RR_PGO_NUM_KCLASS] = {0};
    int64_t n[RR_PGO_NUM_KCLASS] = {0};
  } prof_;
  void pbegin() { if (prof_.on) HIPCHK(hipEventRecord(prof_.e0, stream_)); }
  void pend(int k, int launches = 1) {   // launches: kernels in the timed segment
    if (!prof_.on) return;
    HIPCHK(hipEventRecord(prof_.e1, stream_));
    HIPCHK(hipEventSynchronize(prof_.e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, prof_.e0, prof_.e1));
    prof_.ms[k] += ms;
    prof_.n[k] += launches;
  }

  // a launch with an invalid configuration fails silently unless asked
  void check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw ApiError(RR_PGO_ENODEVICE, std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
  }

  void launch_linearize(double lambda, int lm, int write_system, bool reference_prior = false) {
    if (stop_dirty_ && !opt_active_) {
      // a pipelined rr_pgo_optimize call may have left its stop word set: every path outside such a call starts here
      HIPCHK(hipMemsetAsync(err_.p + 1, 0, sizeof(int), stream_));
      stop_dirty_ = false;
    }
    pbegin();
    if (!is3d_ && edge_lin_) {
      // the edge-parallel form (experiment knob RR_PGO_EDGE_LINEARIZE): clear + prior, one thread per edge, mirror
      const LinArgs<T, S> la = lin_args(lambda, lm, write_system, reference_prior);
      const unsigned nb = (unsigned)((g_.n_nodes() + 255) / 256);
      if (lds_flow_) {
        hipLaunchKernelGGL(k_fill_words, dim3(4), dim3(256), 0, stream_, dep_flags_.p, 2 * sym_.S + 64, 0u);
        hipLaunchKernelGGL(k_fill_words, dim3(16), dim3(256), 0, stream_, reinterpret_cast<unsigned *>(x_ptr_), (int)((size_t)g_.dim * sizeof(T) / 4), X_PENDING_WORD);
      }
      if (write_system) hipLaunchKernelGGL((k_lin_init<T, S>), dim3(nb), dim3(256), 0, stream_, la);
      if (edge_lin_wave_) hipLaunchKernelGGL((k_linearize_wave_edges<T, S>), dim3(n_lin_blocks_), dim3(256), 0, stream_, la, g_.n_edges());
      else hipLaunchKernelGGL((k_linearize_edges<T, S>), dim3(n_lin_blocks_), dim3(LIN_THREADS), 0, stream_, la, g_.n_edges());
      if (write_system) hipLaunchKernelGGL((k_lin_finish<T, S>), dim3(nb), dim3(256), 0, stream_, la);
    } else if (!is3d_) {
      hipLaunchKernelGGL((k_linearize<T, S>), dim3(n_lin_blocks_), dim3(LIN_THREADS), 0, stream_,
                         lin_args(lambda, lm, write_system, reference_prior));
    } else {
      LinArgs3<T, S> a;
      a.n_nodes = n_list_;
      a.node_list = node_list_.p;
      a.pose = pose_.p;
      a.e_idx = e_idx_.p;
      a.e_meas = e_meas_.p;
      a.e_info = e_info3_.p;
      a.e_slot = e_slot_.p;
      a.inc_ptr = inc_ptr_.p;
      a.inc_list = inc_list_.p;
      a.node_offset = node_offset_.p;
      a.diag_off = diag_off_.p;
      a.hvals = hvals_.p;
      a.b = b_.p;
      a.chi2_partial = chi_partial_.p;
      a.anchor = g_.anchor_node;
      a.lambda = lm ? (S)lambda : (S)0;
      a.write_system = write_system;
      a.adds_diag = norm_counts_.p;
      a.zero_words = lds_flow_ ? dep_flags_.p : nullptr;
      a.n_zero_words = lds_flow_ ? 2 * sym_.S + 64 : 0;
      a.fill_words = lds_flow_ ? reinterpret_cast<unsigned *>(x_ptr_) : nullptr;
      a.n_fill_words = lds_flow_ ? (int)((size_t)g_.dim * sizeof(T) / 4) : 0;
      opt_lin_fields(a, lm);
      hipLaunchKernelGGL((k_linearize_se3<T, S>), dim3(n_lin_blocks_), dim3(LIN_THREADS), 0, stream_, a);
    }
    check_launch("k_linearize");
    pend(RR_PGO_K_LINEARIZE);
  }

  void launch_factor() { launch_factor_range(0, sym_.steps.size()); }
  void launch_factor_range(size_t from, size_t to) {
    // tickets and completion flags of the flow levels (one small kernel per factorisation; flow.hip.h, k_flow_reset)
    if (any_flow(from, to)) {
      // ... and the "not there yet" marks of the W blocks of the range's flow fronts
      int e0 = 0, e1 = 0;
      for (size_t si = from; si < to && si < flow_levels_.size(); si++)
        if (flow_levels_[si]) { if (e1 == 0) e0 = flow_levels_[si]->wfill_begin; e1 = flow_levels_[si]->wfill_end; }
      const int flag_wgs = (int)std::min<int64_t>(((int64_t)flow_flags_.n + 255) / 256, 256);
      hipLaunchKernelGGL(k_flow_reset<T>, dim3((unsigned)(flag_wgs + std::max(e1 - e0, 0))), dim3(256), 0, stream_, flow_flags_.p, (int64_t)flow_flags_.n, flag_wgs,
                         winv_.p, flow_wfill_.p + 2 * e0, xnew_.p);
      check_launch("k_flow_reset");
    }
    if (lds_flow_ && from == 0 && to > 0) {
      // every front in LDS: their whole factorisation as ONE launch of ticket-ordered tasks (lds_flow.hip.h)
      const Step &st = sym_.steps[0];
      pbegin();
      const size_t lds = (size_t)st.max_lds_elems * sizeof(T);
      const FactorArgs<T> a = factor_args(0);
      const int grid = std::min(lds_n_tasks_, lds_flow_cus_);
      unsigned *ticket = dep_flags_.p + 2 * sym_.S;
      if (std::min(st.threads, factor_threads_max_) <= 256)
        hipLaunchKernelGGL((k_factor_flow<T, 256>), dim3(grid), dim3(256), lds, stream_, a, (const LdsFlowTask *)lds_ftasks_.p, lds_n_tasks_, ticket);
      else if (std::min(st.threads, factor_threads_max_) <= 512)
        hipLaunchKernelGGL((k_factor_flow<T, 512>), dim3(grid), dim3(512), lds, stream_, a, (const LdsFlowTask *)lds_ftasks_.p, lds_n_tasks_, ticket);
      else
        hipLaunchKernelGGL((k_factor_flow<T, 1024>), dim3(grid), dim3(1024), lds, stream_, a, (const LdsFlowTask *)lds_ftasks_.p, lds_n_tasks_, ticket);
      check_launch("k_factor_flow");
      pend(RR_PGO_K_FACTOR);
      from = 1;
    }
    if (xl_ && from == 1 && to == sym_.steps.size()) {
      pbegin();
      launch_flow_xl();
      pend(RR_PGO_K_BIG_FLOW);
      return;
    }
    for (size_t si = from; si < to; si++) {
      const Step &st = sym_.steps[si];
      pbegin();
      if (st.kind == STEP_TASKS) {
        const int nt = st.task_end - st.task_begin;
        const size_t lds = (size_t)st.max_lds_elems * sizeof(T);
        FactorArgs<T> a = factor_args(st.task_begin);
        const int fth = step_threads(st, nt, factor_threads_max_);
        if (fth == 64) launch_factor_tasks<64>(nt, lds, a);
        else if (fth == 128) launch_factor_tasks<128>(nt, lds, a);
        else if (fth == 256) launch_factor_tasks<256>(nt, lds, a);
        else if (fth <= 512) launch_factor_tasks<512>(nt, lds, a);
        else launch_factor_tasks<1024>(nt, lds, a);
        pend(RR_PGO_K_FACTOR);
      } else {
        launch_big_level(st, true);
        pend(RR_PGO_K_BIGFRONT);
      }
    }
  }

  // Workgroup size of an LDS-front step: the symbolic phase sizes it for the latency of ONE task (a 16-wave workgroup per
  // front of > 128 rows).  Smaller workgroups for levels of thousands of tasks were measured slower (r02, lattice, 5244
  // tasks: 690 us with 1024 threads, 873 with 512, 1045 with 256).
  int step_threads(const Step &st, int, int cap) const { return std::min(st.threads, cap); }

  // The fronts beyond LDS of one level, batched (grid y / z = front): ONE gather pass builds the pivot columns from
  // the children (k_big_build) and adds the H entries and the right-hand side (k_big_assemble); then either the whole
  // level as one dataflow launch (k_big_flow: levels of few fronts and few tasks), or per 128-column super-panel four
  // left-looking k_big_panel32 launches and one K = 128 trailing update (k_big_update; the first one of a front gathers
  // its tiles from the children, its tile (0, 0) factors and inverts the next super-panel's first diagonal block).
  // do_launch == false only counts the launches.  (The r01 / r02 alternatives -- right-looking K = 32 steps, whole
  // super-panels per chain step, the two-stream far update, zero + extend-add assembly, the one-workgroup panel class --
  // were measured slower and removed in r03: profiles/EXPERIMENTS.md.)
  int launch_big_level(const Step &st, bool do_launch) {
    const FactorArgs<T> a = factor_args(st.task_begin);
    const int nf = st.task_end - st.task_begin;
    int maxM = 0, max_nc = 0, n = 0;
    int64_t max_asm = 0;
    bool any_dup = false;
    std::vector<int> fr(nf);
    for (int z = 0; z < nf; z++) {
      const int s = sym_.task_sn[sym_.task_ptr[st.task_begin + z]];
      fr[z] = s;
      maxM = std::max(maxM, sym_.sn_ncols[s] + sym_.sn_nrows[s] + 1);
      max_nc = std::max(max_nc, sym_.sn_ncols[s]);
      max_asm = std::max<int64_t>(max_asm, std::max<int64_t>(sym_.fasm_ptr[s + 1] - sym_.fasm_ptr[s], sym_.sn_ncols[s]));
      any_dup = any_dup || sym_.fdup_ptr[s + 1] > sym_.fdup_ptr[s];
    }
    auto rows_max = [&](int from) {  // max over fronts still active at column `from` of (M - from)
      int r = 0;
      for (int s : fr)
        if (sym_.sn_ncols[s] > from) r = std::max(r, sym_.sn_ncols[s] + sym_.sn_nrows[s] + 1 - from);
      return r;
    };
    // one pass writes every entry of the level's fronts once: the sum of the children's contributions (gathered
    // through inverse maps), zeros elsewhere; the H entries and the rhs are added on top -- by the building waves
    // themselves when the launch is a single round of workgroups (three dependent loads per column at the end of every
    // workgroup: cheaper than a launch there, dearer on the levels of hundreds of fronts -- lattice 4.62 against 4.68 ms
    // with every level fused, sphere2500 +1 % fused), else by a k_big_assemble launch
    const int build_gx = std::min(std::max(((gather_update_ ? std::min(maxM, ((max_nc + 127) / 128) * 128) : maxM) + 3) / 4, 1), 8192);
    const bool fused = fused_assembly_ && (int64_t)build_gx * nf <= 2048;
    if (do_launch) {
      hipLaunchKernelGGL(k_big_build<T>, dim3((unsigned)build_gx, nf), dim3(256), 0, stream_, a, gather_update_ ? 1 : 0, fused ? 1 : 0);
      check_launch("k_big_build");
      if (!fused) {
        hipLaunchKernelGGL(k_big_assemble<T>, dim3((unsigned)std::min<int64_t>((max_asm + 255) / 256, 2048), nf), dim3(256), 0, stream_, a, 1);
        check_launch("k_big_assemble");
      }
      if (any_dup) hipLaunchKernelGGL(k_big_assemble_dup<T>, dim3(1, nf), dim3(64), 0, stream_, a);
    }
    n += (fused ? 1 : 2) + (any_dup ? 1 : 0);
    if (gauge_ok_) {
      bool has_root = false;
      for (int s : fr) has_root = has_root || s == gauge_root_;
      if (has_root) {
        if (do_launch && gauge_now_) launch_gauge_term();
        n += 2;
      }
    }
    if (do_launch) pend(RR_PGO_K_BIGFRONT);   // closes the build / assemble segment
    if (const FlowLevel *lvl = flow_of(st)) {
      // the whole panel chain and every trailing update of the level: ONE launch of ticket-ordered tasks
      if (do_launch) {
        pbegin();
        launch_flow(st, *lvl);
        pend(RR_PGO_K_BIG_FLOW);
      }
      n++;
      n += launch_schur(st, a, do_launch);
      if (do_launch) pbegin();   // re-arm: the caller closes the level with pend(BIGFRONT)
      return n;
    }
    for (int K0 = 0; K0 < max_nc; K0 += BIG_SUPER) {
      // left-looking inside the super-panel: ONE launch per 32 columns (update from the columns K0..kb, multiply by the
      // inverse diagonal block, next diagonal block).  The level's first block is factored and inverted inside its own
      // row launch (by every workgroup); the first block of a later super-panel comes out of the previous trailing update.
      if (do_launch) pbegin();
      const int k_end = std::min(K0 + BIG_SUPER, max_nc);
      for (int kb = K0; kb < k_end; kb += BIG_NB) {
        const int rb = rows_max(kb) - 1;
        if (do_launch) {
          hipLaunchKernelGGL(k_big_panel32<T>, dim3((std::max(rb, 1) + 31) / 32, nf), dim3(64), 0, stream_, a, kb, K0, kb == 0 ? 1 : 0);
          check_launch("k_big_panel32");
        }
        n++;
      }
      if (do_launch) pend(RR_PGO_K_BIG_PANEL, (k_end - K0 + BIG_NB - 1) / BIG_NB);
      // everything right of the super-panel, Schur complement included: the level's real 64 x 64 tiles as a
      // one-dimensional grid (upd_maps_: front slot and tile of every grid index)
      const UpdMap &um = upd_map(st, K0 / BIG_SUPER);
      if (um.n_tiles == 0) continue;   // schur_split_: a level of single-super-panel fronts has no per-super-panel update at all
      if (do_launch) {
        pbegin();
        hipLaunchKernelGGL((k_big_update<T, 2, RRPGO_UPD_DEPTH>), dim3((unsigned)um.n_tiles), dim3(256), 0, stream_, a, K0,
                           (gather_update_ && K0 == 0) ? 1 : 0, (const int32_t *)upd_map_buf_.p + um.offset, um.n_tiles, xcd_remap_ ? 1 : 0,
                           schur_split_ ? schur_tile_ : 0);
        check_launch("k_big_update");
        pend(RR_PGO_K_BIG_UPDATE);
      }
      n++;
    }
    n += launch_schur(st, a, do_launch);
    if (do_launch) pbegin();   // re-arm: the caller closes the level with pend(BIGFRONT)
    return n;
  }
  // the Schur complements of the level's fronts: ONE pass over all pivot columns (K = nc), each tile written once
  int launch_schur(const Step &st, const FactorArgs<T> &a, bool do_launch) {
    const UpdMap &um = schur_maps_[(size_t)(&st - sym_.steps.data())];
    if (um.n_tiles == 0) return 0;   // no map: no split on this level (or no Schur complement at all)
    if (do_launch) {
      pbegin();
      hipLaunchKernelGGL((k_big_schur<T, 2>), dim3((unsigned)um.n_tiles), dim3(256), 0, stream_, a, gather_update_ ? 1 : 0,
                         (const int32_t *)upd_map_buf_.p + um.offset, um.n_tiles, xcd_remap_ ? 1 : 0);
      check_launch("k_big_schur");
      pend(RR_PGO_K_BIG_UPDATE);
    }
    return 1;
  }
  int count_big_launches(const Step &st) { return launch_big_level(st, false); }
  bool level_has_rows(const Step &st) const {
    for (int t = st.task_begin; t < st.task_end; t++)
      if (sym_.sn_nrows[sym_.task_sn[sym_.task_ptr[t]]] > 0) return true;
    return false;
  }
  int count_big_solve_launches(const Step &st) const {
    int max_nc = 1;
    for (int t = st.task_begin; t < st.task_end; t++) max_nc = std::max(max_nc, sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[t]]]);
    const int gemv = level_has_rows(st) ? 1 : 0;
    if (max_nc >= sp_solve_min_nc_) {
      const size_t si = (size_t)(&st - sym_.steps.data());
      return si < solve_flow_.size() && solve_flow_[si] ? gemv + 1 : gemv + (max_nc + BIG_SUPER - 1) / BIG_SUPER;
    }
    return gemv + 1;
  }

  void launch_solve() {
    launch_solve_steps(lds_flow_ ? 1 : 0);   // the fronts beyond LDS (and, in the level schedule, everything), top down
    if (lds_flow_) {
      const Step &st = sym_.steps[0];
      pbegin();
      const size_t lds = (size_t)step_solve_lds_[0] * sizeof(T);
      FactorArgs<T> a = factor_args(0);
      a.solve_lds = step_solve_lds_[0];
      const int sth = std::min(st.threads, solve_threads_max_);
      const int grid = std::min(lds_n_stasks_, lds_flow_cus_ * (sth <= 256 ? 4 : 2));
      unsigned *ticket = dep_flags_.p + 2 * sym_.S + 32;
      if (sth <= 256)
        hipLaunchKernelGGL((k_solve_flow<T, 256>), dim3(grid), dim3(256), lds, stream_, a, (const LdsFlowTask *)lds_stasks_.p, lds_n_stasks_, ticket, gemv_part_.p, (int64_t)g_.dim);
      else
        hipLaunchKernelGGL((k_solve_flow<T, 512>), dim3(grid), dim3(512), lds, stream_, a, (const LdsFlowTask *)lds_stasks_.p, lds_n_stasks_, ticket, gemv_part_.p, (int64_t)g_.dim);
      check_launch("k_solve_flow");
      pend(RR_PGO_K_SOLVE);
    }
  }
  void launch_solve_steps(int first_step) {
    for (int i = (int)sym_.steps.size() - 1; i >= first_step; i--) {   // shared top fronts first, then this rank's subtrees
      const Step &st = sym_.steps[i];
      if (mid_in_flow_[(size_t)i]) continue;   // its fronts are tasks of k_solve_flow
      const size_t lds = (size_t)step_solve_lds_[i] * sizeof(T);
      pbegin();
      if (st.kind == STEP_TASKS) {
        const int nt = st.task_end - st.task_begin;
        FactorArgs<T> a = factor_args(st.task_begin);
        a.solve_lds = step_solve_lds_[i];
        const int sth = step_threads(st, nt, solve_threads_max_);
        if (sth <= 64) launch_solve_tasks<64>(nt, lds, a);
        else if (sth <= 128) launch_solve_tasks<128>(nt, lds, a);
        else if (sth <= 256) launch_solve_tasks<256>(nt, lds, a);
        else launch_solve_tasks<512>(nt, lds, a);
        pend(RR_PGO_K_SOLVE);
      } else {
        // t = y1 - L21^T x[rows] over the whole chip, then L11: per 128-column super-panel over the chip (wide pivot blocks) or one workgroup per front
        const int nf = st.task_end - st.task_begin;
        int max_nc = 1;
        for (int z = 0; z < nf; z++) max_nc = std::max(max_nc, sym_.sn_ncols[sym_.task_sn[sym_.task_ptr[st.task_begin + z]]]);
        const int cb = (max_nc + 63) / 64;
        const int R = gemv_slices(st);
        const FactorArgs<T> fa = factor_args(st.task_begin);
        if (level_has_rows(st)) {   // (a level of root fronts -- no rows below the pivot block -- has no L21^T x to form: the solves read no partial sums)
          hipLaunchKernelGGL(k_big_gemv_partial<T>, dim3(cb, R, nf), dim3(256), 0, stream_, fa, gemv_part_.p, (int64_t)g_.dim, R);
          check_launch("k_big_gemv_partial");
        }
        if (max_nc >= sp_solve_min_nc_) {
          if (solve_flow_[(size_t)i]) {
            // wide pivot blocks, the whole level's chain steps and folds as ONE launch of ticketed tasks (flow.hip.h)
            const SolveFlowLevel &sl = *solve_flow_[(size_t)i];
            int cus_grid = std::min(sl.n_tasks, std::max(flow_grid_ / (sizeof(T) == 4 ? RRPGO_FLOW_WAVES : 1), 1));
            hipLaunchKernelGGL(k_big_solve_flow<T>, dim3((unsigned)cus_grid), dim3(1024), 0, stream_, fa, (const T *)gemv_part_.p, (int64_t)g_.dim, R,
                               flow_flags_.p, (const SolveFlowFront *)sl.fronts.p, (const SolveFlowTask *)sl.tasks.p, sl.n_tasks,
                               flow_flags_.p + sl.ticket_word);
            check_launch("k_big_solve_flow");
            pend(RR_PGO_K_BIG_SOLVE, 2);
            continue;
          }
          // wide pivot blocks: one launch per 128-column super-panel, L11 read by the whole chip
          const int nsp = (max_nc + BIG_SUPER - 1) / BIG_SUPER;
          for (int ell = 0; ell < nsp; ell++) {
            hipLaunchKernelGGL(k_big_solve_sp<T>, dim3(1 + (max_nc + 63) / 64, nf), dim3(1024), 0, stream_, fa, ell, (const T *)gemv_part_.p, (int64_t)g_.dim, R);
            check_launch("k_big_solve_sp");
          }
          pend(RR_PGO_K_BIG_SOLVE, 1 + nsp);
          continue;
        }
        hipLaunchKernelGGL((k_solve_mid<T, 1024>), dim3(nf), dim3(1024), lds, stream_, fa, 1, (const T *)gemv_part_.p, (int64_t)g_.dim, R);
        pend(RR_PGO_K_BIG_SOLVE, 2);
      }
    }
  }

  void launch_update(const T *dx_ref_in, double sign, bool write_ref, bool export_only = false, bool fused_finalize = false, const int *gate = nullptr) {
    FinArgs fin{};
    if (fused_finalize) {
      fin.enabled = 1; fin.chi_partial = chi_partial_.p; fin.n_chi = n_lin_blocks_; fin.hist = hist_.p; fin.counter = counter_.p;
      fin.advance = 1; fin.ring = HIST; fin.blocks_done = blocks_done_.p;
      if (opt_active_) { fin.ctrl = opt_ctrl_.p; fin.ring_host = opt_ring_; fin.err = err_.p; }
    }
    pbegin();
    if (!is3d_) {
      UpdArgs<T, S> u;
      u.n_nodes = n_list_;
      u.node_list = node_list_.p;
      u.norm_counts = norm_counts_.p;
      u.pose = pose_.p;
      u.node_dim = node_dim_.p;
      u.node_pcol = node_pcol_.p;
      u.node_offset = node_offset_.p;
      u.x = x_ptr_;
      u.dx_ref_in = dx_ref_in;
      u.dx_ref_out = write_ref ? dx_ref_.p : nullptr;
      u.sign = (S)sign;
      u.norm_partial = norm_partial_.p;
      u.gauge_anchor = (!dx_ref_in && gauge_now_) ? g_.anchor_node : -1;
      u.export_only = export_only ? 1 : 0;
      u.err = dx_ref_in ? nullptr : err_.p;
      u.gate = gate;
      u.fin = fin;
      hipLaunchKernelGGL((k_update<T, S>), dim3(n_upd_blocks_), dim3(UPD_THREADS), 0, stream_, u);
    } else {
      UpdArgs3<T, S> u;
      u.n_nodes = n_list_;
      u.node_list = node_list_.p;
      u.norm_counts = norm_counts_.p;
      u.pose = pose_.p;
      u.node_pcol = node_pcol_.p;
      u.node_offset = node_offset_.p;
      u.x = x_ptr_;
      u.dx_ref_in = dx_ref_in;
      u.dx_ref_out = write_ref ? dx_ref_.p : nullptr;
      u.sign = (S)sign;
      u.norm_partial = norm_partial_.p;
      u.export_only = export_only ? 1 : 0;
      u.err = dx_ref_in ? nullptr : err_.p;
      u.gate = gate;
      u.fin = fin;
      hipLaunchKernelGGL((k_update_se3<T, S>), dim3(n_upd_blocks_), dim3(UPD_THREADS), 0, stream_, u);
    }
    check_launch("k_update");
    pend(RR_PGO_K_UPDATE);
  }

  void launch_finalize(bool chi, bool norm, bool advance) {
    pbegin();
    hipLaunchKernelGGL(k_finalize_slot, dim3(1), dim3(256), 0, stream_, chi_partial_.p, chi ? n_lin_blocks_ : 0,
                       norm_partial_.p, norm ? n_upd_blocks_ : 0, hist_.p, counter_.p, advance ? 1 : 0);
    pend(RR_PGO_K_REDUCE);
  }

  // one Gauss-Newton iteration: chi2(state) -> slot.chi, |dx| -> slot.norm, slot++
  void enqueue_gn_iteration() {
    launch_linearize(0.0, 0, 1);
    launch_factor();
    launch_solve();
    launch_update(nullptr, 1.0, true, false, true);   // + the reduction of the chi2 / |dx|^2 partials in its last workgroup
  }

  void ensure_gn_graph() {
    if (gn_exec_) return;
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    try {
      enqueue_gn_iteration();
    } catch (...) {
      // a failed launch must not leave the (pooled) stream in capture mode
      (void)hipStreamEndCapture(stream_, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    HIPCHK(hipStreamEndCapture(stream_, &graph));
    const hipError_t ie = hipGraphInstantiate(&gn_exec_, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { gn_exec_ = nullptr; throw ApiError(RR_PGO_ENODEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
  }

  // One host round trip per iteration: the two scalars and the device error flag come back in
  // two async copies behind a single stream synchronisation.
  void read_slot(int slot, double *chi, double *norm) {
    int *eflag = reinterpret_cast<int *>(host_pair_.p + 2);
    HIPCHK(hipMemcpyAsync(host_pair_.p, hist_.p + 2 * (slot % HIST), 2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(eflag, err_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    if (chi) *chi = host_pair_[0];
    if (norm) *norm = host_pair_[1];
    throw_on_flag(*eflag);
  }

  void throw_on_flag(int e) {
    if (!e) return;
    HIPCHK(hipMemsetAsync(err_.p, 0, sizeof(int), stream_));
    // (a launch that drains after a timed-out hand-off computes on tiles that are not there: whatever else it flags is noise)
    if (e & DEVERR_FLOW_TIMEOUT)
      throw ApiError(RR_PGO_ETIMEOUT, "a hand-off between workgroups of one launch timed out; the step was not applied");
    if (e & DEVERR_NOT_SPD) throw ApiError(RR_PGO_ENOTSPD, "normal matrix is not positive definite (non-positive pivot)");
    throw ApiError(RR_PGO_ENODEVICE, "device-side error flag " + std::to_string(e));
  }

  void check_device_error() {
    int *eflag = reinterpret_cast<int *>(host_pair_.p + 2);
    HIPCHK(hipMemcpyAsync(eflag, err_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    throw_on_flag(*eflag);
  }

  void reset_counter() {
    HIPCHK(hipMemsetAsync(counter_.p, 0, sizeof(int), stream_));
    host_counter_ = 0;
  }

  double chi2_now() {
    reset_counter();
    launch_linearize(0.0, 0, 0);
    launch_finalize(true, false, false);
    double c = 0;
    read_slot(0, &c, nullptr);
    return c;
  }

 public:
  // On a handle that holds ONE RANK's share of a graph (world_size > 1) these three would return rank-partial
  // results (chi2 of the edges this rank owns, partial diagonal blocks of the shared nodes, an update of this
  // rank's nodes only): refused -- rr_pgo_stage(h, 2) + the sum all-reduce of exchange buffer 1 is the chi2 of a
  // sharded graph, rr_pgo_stage(h, 0 / 1) its iteration.
  void refuse_rank_partial(const char *what) const {
    if (world_ > 1)
      throw ApiError(RR_PGO_EUNSUPPORTED, std::string(what) + " on one rank of a sharded graph would be rank-partial: use rr_pgo_stage + the two collectives");
  }
  void chi2(double *out) override { refuse_rank_partial("rr_pgo_chi2"); *out = chi2_now(); }

  void linearize_solve(double lambda, int lm, double *dx_out) override {
    if (sharded_) throw ApiError(RR_PGO_EUNSUPPORTED, "sharded handle: drive it with rr_pgo_stage + the two collectives");
    launch_linearize(lambda, lm, 1);
    launch_factor();
    launch_solve();
    launch_update(nullptr, 1.0, true, true);   // permuted solution -> reference order (and the anchor gauge), state untouched
    std::vector<T> tmp((size_t)g_.dim);
    HIPCHK(hipMemcpyAsync(tmp.data(), dx_ref_.p, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    check_device_error();
    for (int i = 0; i < g_.dim; i++) dx_out[i] = (double)tmp[i];
  }

  void update(const double *dx, double sign) override {
    refuse_rank_partial("rr_pgo_update");
    std::vector<T> tmp((size_t)g_.dim);
    for (int i = 0; i < g_.dim; i++) tmp[i] = (T)dx[i];
    HIPCHK(hipMemcpyAsync(dx_ref_.p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, stream_));
    launch_update(dx_ref_.p, sign, false);
    HIPCHK(hipStreamSynchronize(stream_));
  }

  // optimize(), pose_graph_optimization.rs:247-303
  void optimize(int solver, int iters, double *errors, int *n_errors, double *norms) override {
    if (sharded_) throw ApiError(RR_PGO_EUNSUPPORTED, "sharded handle: drive it with rr_pgo_stage + the two collectives");
    // (an iteration of dozens of launches -- the 1M-edge lattice: 67, 4.4 ms -- gains nothing from the loop on the device: a host
    // round trip is 0.5 % of its iteration, and the item behind the stop would be dozens of empty launches.  Built and measured, r06:
    // 4.55 against 4.50 ms per iteration through optimize(), also with only the next linearisation enqueued ahead.)
    if (!edge_lin_ && !sync_optimize_ && !prof_.on && n_launches_per_iter < 48) {
      optimize_pipelined(solver, iters, errors, n_errors, norms);
      return;
    }
    const double tolerance = 1e-4;  // :253
    int ne = 0;
    if (solver == RR_PGO_GAUSS_NEWTON) {
      // chi2 of the state BEFORE an update comes out of that iteration's own
      // linearisation pass, so errors[i] is known once iteration i has run and
      // one chi2-only pass after the loop supplies the last entry.
      // an iteration of a handful of launches (the dataflow form of the small graphs: four) is enqueued as it is; capturing
      // and instantiating a hipGraph costs more than ten such iterations save (the closure the reference's bench times,
      // benches/graph_slam.rs:9-10, is ONE new() + optimize(10)).  Graphs of dozens of launches replay a captured graph.
      // (optimize() reads two scalars back after every iteration, so the enqueueing is not hidden behind the device as it is
      // in iterate_async: dozens of launches stay one graph launch here.)  RR_PGO_FORCE_GRAPH=1: always the graph.
      const bool eager = (no_graph_ || n_launches_per_iter <= 8) && !gn_exec_ && !force_graph_;
      if (!eager) ensure_gn_graph();
      reset_counter();
      int done = 0;
      for (int i = 0; i < iters; i++) {
        if (eager) enqueue_gn_iteration();
        else HIPCHK(hipGraphLaunch(gn_exec_, stream_));
        double chi, nrm;
        read_slot(i, &chi, &nrm);
        errors[ne++] = chi;
        if (norms) norms[i] = nrm;
        done = i + 1;
        if (nrm < tolerance) break;  // :298-300
      }
      (void)done;
      errors[ne++] = chi2_now();
    } else {
      double lambda = 0.01;  // :254
      double last_error = chi2_now();
      errors[ne++] = last_error;
      for (int i = 0; i < iters; i++) {
        reset_counter();
        launch_linearize(lambda, 1, 1);
        launch_factor();
        launch_solve();
        launch_update(nullptr, 1.0, true);
        launch_finalize(false, true, false);
        launch_linearize(0.0, 0, 0);
        launch_finalize(true, false, false);
        double error, nrm;
        read_slot(0, &error, &nrm);
        if (last_error < error) {  // :276-281
          launch_update(dx_ref_.p, -1.0, false);
          lambda *= 2.0;
        } else {
          lambda /= 2.0;
        }
        last_error = error;  // :284
        if (norms) norms[i] = nrm;
        errors[ne++] = error;
        if (nrm < tolerance) break;
      }
      HIPCHK(hipStreamSynchronize(stream_));
    }
    *n_errors = ne;
  }

  // ---- rr_pgo_optimize without a host round trip per iteration (on the small graphs an iteration is a few launches of
  // 0.1 - 0.6 ms altogether, and a stream synchronisation + two copies per iteration were ~25 us of idle device; handles of
  // fewer than 48 launches per iteration).
  // The loop of :247-303 is cut into ITEMS, each a fixed sequence of launches whose last kernel publishes (chi2, |dx|, flags)
  // in the host-coherent ring and -- on the device -- takes the decisions the reference takes on the host: the stop rule
  // (:298-300, the stop word err[1]: every later launch of the call is empty or leaves the state alone), Levenberg-
  // Marquardt's accept / reject and lambda (:275-282), the end of the call after a failed factorisation (:271).
  //   Gauss-Newton          item k < n: iteration k (k_linearize ... k_update; chi2 of the state BEFORE the step and |dx|);
  //                         item n: chi2 of the final state.  After a stop in iteration k, item k + 1 IS that chi2 (its
  //                         linearisation runs, the rest is skipped): the errors are the reference's list, shifted by one
  //   Levenberg-Marquardt   item 0: the initial error (:255); item k: iteration k - 1 = step, chi2, decision, gated undo
  // The host keeps ONE item queued behind the one that is running (enqueueing an item takes a fraction of its run time), polls
  // the ring, and stops enqueueing when it sees the stop: optimize(100) that stops after 6 iterations pays one skipped item.
  void enqueue_opt_item(bool lm, long k, int iters) {
    OptItemArgs oi{};
    oi.chi_partial = chi_partial_.p; oi.norm_partial = norm_partial_.p; oi.n_chi = n_lin_blocks_; oi.n_norm = 0; oi.mode = 0;
    oi.ctrl = opt_ctrl_.p; oi.ring = opt_ring_; oi.err = err_.p;
    struct Pub { int &p; ~Pub() { p = 0; } } pub{opt_publish_};
    if (!lm) {
      // an iteration: its linearisation publishes only when the stop word is set (then the rest of the item is empty
      // launches); the item behind the last iteration: chi2 only, published by the linearisation's last workgroup
      opt_publish_ = k < iters ? 1 : 2;
      if (k < iters) enqueue_gn_iteration();
      else launch_linearize(0.0, 0, 0);
      return;
    }
    if (k == 0) {
      launch_linearize(0.0, 0, 0);
      hipLaunchKernelGGL(k_opt_item, dim3(1), dim3(256), 0, stream_, oi);
      check_launch("k_opt_item");
      return;
    }
    launch_linearize(0.0, 1, 1);   // lambda: OptCtrl::lambda
    launch_factor();
    launch_solve();
    launch_update(nullptr, 1.0, true);
    launch_linearize(0.0, 0, 0);
    oi.n_norm = n_upd_blocks_; oi.mode = 1;
    hipLaunchKernelGGL(k_opt_item, dim3(1), dim3(256), 0, stream_, oi);
    check_launch("k_opt_item");
    launch_update(dx_ref_.p, -1.0, false, false, false, &opt_ctrl_.p->reject);   // :277, when the device decided so
  }
  // the slot of item `idx` (0-based); false: the stream went idle or failed without publishing it
  bool wait_opt_slot(long idx, OptSlot *out) {
    volatile OptSlot *s = opt_ring_ + idx % OPT_RING;
    const unsigned long long want = (unsigned long long)idx + 1;
    double t_query = now_ms();
    for (unsigned spins = 0;; spins++) {
      if (__atomic_load_n(const_cast<unsigned long long *>(&s->seq), __ATOMIC_ACQUIRE) == want) break;
      __builtin_ia32_pause();
      if ((spins & 1023u) == 1023u && now_ms() - t_query > 1.0) {
        // every wait inside a launch is bounded, so the stream always drains: an idle stream without the slot is a lost launch
        const hipError_t q = hipStreamQuery(stream_);
        if (q == hipSuccess) {
          if (__atomic_load_n(const_cast<unsigned long long *>(&s->seq), __ATOMIC_ACQUIRE) == want) break;
          return false;
        }
        if (q != hipErrorNotReady) throw ApiError(RR_PGO_ENODEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        t_query = now_ms();
      }
    }
    out->chi2 = s->chi2; out->norm = s->norm; out->flags = s->flags; out->seq = want;
    return true;
  }
  void optimize_pipelined(int solver, int iters, double *errors, int *n_errors, double *norms) {
    const bool lm = solver != RR_PGO_GAUSS_NEWTON;
    const long total = (long)iters + 1;
    for (int i = 0; i < OPT_RING; i++) opt_ring_[i].seq = 0ull;   // (nothing of an earlier call is in flight: every call consumes what it enqueued)
    struct Scope {   // whatever happens, the launches behind this call are plain ones again and know about the stop word
      Engine *e;
      bool done = false;
      ~Scope() {
        e->opt_active_ = false; e->opt_first_ = false; e->stop_dirty_ = true;
        // left by an exception (a failed launch, a lost item): what is still enqueued must not publish into the next call's ring
        if (!done) (void)hipStreamSynchronize(e->stream_);
      }
    } scope{this};
    opt_active_ = true; opt_first_ = true; opt_lambda_dev_ = lm; opt_lambda0_ = 0.01;   // :254
    // TWO whole items in the queue: the device never waits for the host, and the item behind the stop is a handful of empty
    // launches behind its linearisation (which publishes the final chi2)
    long enq = 0, seen = 0;   // items enqueued / consumed
    int ne = 0, dev_err = 0;
    bool stop_seen = false;
    while (true) {
      while (enq < total && enq < seen + 2 && !stop_seen) enqueue_opt_item(lm, enq++, iters);
      if (seen >= enq) break;
      OptSlot sl;
      if (!wait_opt_slot(seen, &sl)) {
        HIPCHK(hipStreamSynchronize(stream_));
        check_device_error();
        throw ApiError(RR_PGO_ENODEVICE, "internal: an item of rr_pgo_optimize was never published");
      }
      const long k = seen++;
      const int e = sl.flags >> OPT_ERR_SHIFT;
      if (e) { dev_err = e; stop_seen = true; continue; }
      if (!lm) {
        if (k == iters || (sl.flags & OPT_SKIPPED)) { if (!dev_err) errors[ne++] = sl.chi2; continue; }   // chi2 of the final state
        errors[ne++] = sl.chi2;
        if (norms) norms[k] = sl.norm;
      } else {
        if (sl.flags & OPT_SKIPPED) continue;
        errors[ne++] = sl.chi2;
        if (k > 0 && norms) norms[k - 1] = sl.norm;
      }
      if (sl.flags & OPT_STOP) stop_seen = true;
    }
    scope.done = true;   // every item that was enqueued has been consumed
    if (dev_err) throw_on_flag(dev_err);
    *n_errors = ne;
  }

  void get_state(double *out) override {
    const int N = g_.n_nodes();
    std::vector<V4> pose(pose_.n);
    HIPCHK(hipMemcpyAsync(pose.data(), pose_.p, pose.size() * sizeof(V4), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    for (int i = 0; i < N; i++) {
      if (is3d_) {
        const V4 &t = pose[2 * i], &q = pose[2 * i + 1];
        *out++ = (double)t.x; *out++ = (double)t.y; *out++ = (double)t.z;
        *out++ = (double)q.x; *out++ = (double)q.y; *out++ = (double)q.z; *out++ = (double)q.w;
        continue;
      }
      *out++ = (double)pose[i].x;
      *out++ = (double)pose[i].y;
      if (g_.node_kind[i] == NODE_SE2) *out++ = std::atan2((double)pose[i].w, (double)pose[i].z);
    }
  }

  // (the poses go through a pinned staging buffer into a device-side copy, from there into the state, and the call does not
  // wait for either copy; the state of the previous call is remembered, so setting the SAME state again -- restarting a run,
  // what the benchmark does between two optimize() calls -- costs the host a comparison and one small launch instead of the
  // conversion loop (a cosine and a sine per pose) and a host-to-device copy)
  void restore_saved_state() {
    const int64_t n16 = (int64_t)(pose_.n * sizeof(V4) / 16);
    hipLaunchKernelGGL(k_copy_words16, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, 1024)), dim3(256), 0, stream_,
                       reinterpret_cast<const uint4 *>(pose_saved_.p), reinterpret_cast<uint4 *>(pose_.p), n16);
    check_launch("k_copy_words16");
  }
  void set_state(const double *st) override {
    const int N = g_.n_nodes();
    const size_t n_in = g_.node_state.size();
    if (state_stage_.p && state_in_.size() == n_in && std::memcmp(state_in_.data(), st, n_in * sizeof(double)) == 0) {
      restore_saved_state();
      return;
    }
    if (!state_stage_.p) {
      state_stage_.alloc(pose_.n, false);
      state_stage_ev_.create(hipEventDisableTiming);
      pose_saved_.alloc(pose_.n);
    } else {
      HIPCHK(hipEventSynchronize(state_stage_ev_));   // the previous copy out of the buffer has been made
    }
    state_in_.assign(st, st + n_in);
    V4 *pose = state_stage_.p;
    for (int i = 0; i < N; i++) {
      if (is3d_) {
        const double n = std::sqrt(st[3] * st[3] + st[4] * st[4] + st[5] * st[5] + st[6] * st[6]);
        pose[2 * i] = V4{(S)st[0], (S)st[1], (S)st[2], (S)0};
        pose[2 * i + 1] = V4{(S)(st[3] / n), (S)(st[4] / n), (S)(st[5] / n), (S)(st[6] / n)};
        st += 7;
      } else if (g_.node_kind[i] == NODE_SE2) {
        pose[i] = V4{(S)st[0], (S)st[1], (S)std::cos(st[2]), (S)std::sin(st[2])};
        st += 3;
      } else {
        pose[i] = V4{(S)st[0], (S)st[1], (S)0, (S)0};
        st += 2;
      }
    }
    HIPCHK(hipMemcpyAsync(pose_saved_.p, pose, pose_.n * sizeof(V4), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipEventRecord(state_stage_ev_, stream_));
    restore_saved_state();
  }

  void assemble(double lambda, int lm, std::vector<double> &hv, std::vector<double> &b) override {
    refuse_rank_partial("rr_pgo_assemble");
    launch_linearize(lambda, lm, 1, true);   // always the reference's system (anchor prior), whatever the factor's gauge
    std::vector<T> th((size_t)sym_.n_hvals), tb((size_t)g_.dim);
    HIPCHK(hipMemcpyAsync(th.data(), hvals_.p, th.size() * sizeof(T), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(tb.data(), b_.p, tb.size() * sizeof(T), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    hv.assign(th.begin(), th.end());
    b.assign(tb.begin(), tb.end());
  }

  void iterate_async(int iters) override {
    if (sharded_) throw ApiError(RR_PGO_EUNSUPPORTED, "sharded handle: drive it with rr_pgo_stage + the two collectives");
    // RR_PGO_NO_GRAPH=1: plain launches (rocprofv3 --kernel-trace crashes on replays of graphs
    // with thousands of nodes; profiling runs of the large workloads use this switch)
    // plain launches queue back to back while consecutive graph launches leave the GPU idle for ~8 us each (rocprofv3
    // --kernel-trace, scripts/kernel_gaps.py), and the host stays far ahead of the device even at 67 launches per iteration
    // (the 1M-edge lattice: ~0.2 ms of enqueueing against 4.4 ms of kernels).  Measured: intel + 2 %, sphere2500 + 0.8 %,
    // lattice + 0.7 %.  RR_PGO_FORCE_GRAPH=1: replays of the captured hipGraph (r01 - r03's form).
    if (no_graph_ || !force_graph_) {
      for (int i = 0; i < iters; i++) enqueue_gn_iteration();
      return;
    }
    ensure_gn_graph();
    for (int i = 0; i < iters; i++) HIPCHK(hipGraphLaunch(gn_exec_, stream_));
  }

  void sync() override {
    HIPCHK(hipStreamSynchronize(stream_));
    check_device_error();
  }

  int schur_tile() const override { return schur_split_ ? schur_tile_ : 0; }
  void mark_flow_fronts(std::vector<char> &in_flow) override {
    for (size_t si = 0; si < flow_levels_.size(); si++)
      if (flow_levels_[si])
        for (int t = sym_.steps[si].task_begin; t < sym_.steps[si].task_end; t++)
          in_flow[sym_.task_sn[sym_.task_ptr[t]]] = flow_levels_[si]->schur_split ? 1 : 2;   // 2: its Schur tiles are UPDATE tasks too
  }
  // diagnostic builds: the task list and the stamps of the `level`-th flow level (-1: no such level)
  int flow_trace(int level, std::vector<int32_t> &tasks, std::vector<unsigned long long> &stamps, int *nf, double *est_us) override {
    if (xl_) {
      // the cross-level list as "level 0": the front slot of every task replaced by its SUPERNODE (slots repeat from level to level)
      if (level != 0) return -1;
      tasks.resize(xl_host_.size() * 4);
      size_t i = 0;
      for (size_t si = 0; si < flow_levels_.size(); si++) {
        if (!flow_levels_[si]) continue;
        const Step &st = sym_.steps[si];
        for (int k2 = 0; k2 < flow_levels_[si]->n_tasks; k2++, i++) {
          const FlowTask &t = xl_host_[i].t;
          const int sn = sym_.task_sn[sym_.task_ptr[st.task_begin + (t.kind_front & 0xffffff)]];
          tasks[4 * i] = (t.kind_front & ~0xffffff) | sn;
          tasks[4 * i + 1] = t.p0; tasks[4 * i + 2] = t.p1; tasks[4 * i + 3] = t.p2;
        }
      }
      stamps.resize(xl_trace_.n);
      if (xl_trace_.n) HIPCHK(hipMemcpy(stamps.data(), xl_trace_.p, xl_trace_.n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      *nf = sym_.S;
      *est_us = 0;
      return (int)xl_host_.size();
    }
    int k = 0;
    for (size_t si = 0; si < flow_levels_.size(); si++) {
      if (!flow_levels_[si]) continue;
      if (k++ != level) continue;
      FlowLevel &l = *flow_levels_[si];
      tasks.resize(l.host_tasks.size() * 4);
      if (!l.host_tasks.empty()) std::memcpy(tasks.data(), l.host_tasks.data(), l.host_tasks.size() * sizeof(FlowTask));
      stamps.resize(l.trace.n);
      if (l.trace.n) HIPCHK(hipMemcpy(stamps.data(), l.trace.p, l.trace.n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      *nf = sym_.steps[si].task_end - sym_.steps[si].task_begin;
      *est_us = l.est_us;
      return l.n_tasks;
    }
    return -1;
  }

  void read_stamps(std::vector<unsigned long long> &out) override {
    out.resize(stamps_.n);
    if (stamps_.n) HIPCHK(hipMemcpy(out.data(), stamps_.p, stamps_.n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }

  // ---- sharded execution: one GN iteration = stage 0, all-gather of exchange buffer 0, stage 1, sum all-reduce
  // of exchange buffer 1 (two doubles).  Both collectives are the caller's (RCCL), issued on this stream.
  void exchange_buffer(int which, void **ptr, int64_t *n, int32_t *esize) override {
    if (!sharded_) throw ApiError(RR_PGO_EINVAL, "handle is not sharded");
    if (which == 0) { *ptr = xch_; *n = xch_n_; *esize = (int32_t)sizeof(T); }
    else if (which == 1) { *ptr = scal_; *n = 2; *esize = (int32_t)sizeof(double); }
    else throw ApiError(RR_PGO_EINVAL, "exchange buffer index must be 0 or 1");
  }
  void set_exchange_buffer(int which, void *ptr, int64_t n) override {
    if (!sharded_) throw ApiError(RR_PGO_EINVAL, "handle is not sharded");
    for (hipGraphExec_t &e : stage_exec_) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }   // captured pointers
    if (which == 0 && n >= xch_n_) xch_ = (T *)ptr;
    else if (which == 1 && n >= 2) scal_ = (double *)ptr;
    else throw ApiError(RR_PGO_EINVAL, "exchange buffer index / size mismatch");
  }
  void enqueue_stage(int stg, double lambda, int lm) {
    const size_t n_local = world_ > 1 ? (size_t)sym_.n_local_steps : sym_.steps.size();
    if (stg == 0) {
      // linearise this rank's nodes (own + shared top separators), factor its own subtrees, publish the
      // boundary fronts' update matrices in this rank's chunk of buffer 0
      launch_linearize(lambda, lm, 1);
      if (n_shared_ > 0)   // this rank's partial sums of the shared nodes' diagonal blocks and rhs, behind its update matrices
        hipLaunchKernelGGL(k_pack_shared<T>, dim3((n_shared_ + 255) / 256), dim3(256), 0, stream_, shared_src_.p, n_shared_,
                           (const T *)hvals_.p, (const T *)b_.p, xch_ + (int64_t)rank_ * sym_.xch_chunk + sym_.xch_shared_off);
      launch_factor_range(0, n_local);
      if (n_pack_ > 0) {
        hipLaunchKernelGGL(k_pack_boundary<T>, dim3(std::min(pack_max_nu_, 1024), n_pack_), dim3(256), 0, stream_,
                           factor_args(0), pack_list_.p);
        check_launch("k_pack_boundary");
      }
      if (world_ > 1)   // this rank's error flag travels with its chunk (the group must agree on ENOTSPD)
        hipLaunchKernelGGL(k_pack_err<T>, dim3(1), dim3(64), 0, stream_, (const int *)err_.p,
                           xch_ + (int64_t)rank_ * sym_.xch_chunk + sym_.xch_flag_off);
    } else if (stg == 1) {
      // after the all-gather: the shared top fronts (every rank, identical), the back substitution of the top
      // and of this rank's subtrees, the update of this rank's nodes, its partial chi2 / |dx|^2
      if (n_shared_ > 0)   // the shared nodes' diagonal blocks and rhs: the P partial sums, added in rank order
        hipLaunchKernelGGL(k_sum_shared<T>, dim3((n_shared_ + 255) / 256), dim3(256), 0, stream_, shared_src_.p, n_shared_, hvals_.p, b_.p,
                           (const T *)xch_, sym_.xch_chunk, sym_.xch_shared_off, world_);
      if (world_ > 1)
        hipLaunchKernelGGL(k_merge_err<T>, dim3(1), dim3(64), 0, stream_, err_.p, (const T *)xch_, sym_.xch_chunk, sym_.xch_flag_off, world_);
      launch_factor_range(n_local, sym_.steps.size());
      launch_solve();
      launch_update(nullptr, 1.0, true);
      hipLaunchKernelGGL(k_finalize_partial, dim3(1), dim3(256), 0, stream_, chi_partial_.p, n_lin_blocks_, norm_partial_.p,
                         n_upd_blocks_, scal_);
    } else {
      // chi2 of the current state only (the last entry of optimize()'s error list): partial sum -> buffer 1
      launch_linearize(0.0, 0, 0);
      hipLaunchKernelGGL(k_finalize_partial, dim3(1), dim3(256), 0, stream_, chi_partial_.p, n_lin_blocks_, norm_partial_.p, 0, scal_);
    }
    check_launch("stage");
  }
  void stage(int stg, double lambda, int lm) override {
    if (!sharded_) throw ApiError(RR_PGO_EINVAL, "handle is not sharded");
    if (stg < 0 || stg > 2) throw ApiError(RR_PGO_EINVAL, "stage must be 0, 1 or 2");
    if (stg == 2 || lm || no_graph_) { enqueue_stage(stg, lambda, lm); return; }
    if (!stage_exec_[stg]) {   // Gauss-Newton stages replay a captured graph like the unsharded iteration
      hipGraph_t graph = nullptr;
      if (stg == 1) gauge_now_ = gauge_ok_;   // what stage 0 (lm == 0) set; the capture must not depend on call order
      HIPCHK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      try {
        enqueue_stage(stg, 0.0, 0);
      } catch (...) {
        (void)hipStreamEndCapture(stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      HIPCHK(hipStreamEndCapture(stream_, &graph));
      const hipError_t ie = hipGraphInstantiate(&stage_exec_[stg], graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) { stage_exec_[stg] = nullptr; throw ApiError(RR_PGO_ENODEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
    }
    if (stg == 0) gauge_now_ = gauge_ok_;
    HIPCHK(hipGraphLaunch(stage_exec_[stg], stream_));
  }
  // (chi2, |dx|) of the last stage-1 / stage-2 call, valid once the caller has all-reduced buffer 1
  void read_last_scalars(double *chi, double *norm) override {
    if (!sharded_) throw ApiError(RR_PGO_EINVAL, "handle is not sharded");
    int *eflag = reinterpret_cast<int *>(host_pair_.p + 2);
    HIPCHK(hipMemcpyAsync(host_pair_.p, scal_, 2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(eflag, err_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    *chi = host_pair_[0];
    *norm = std::sqrt(host_pair_[1]);
    throw_on_flag(*eflag);
  }

  // Test-only failure injection: make ONE hand-off of a dataflow launch never arrive, so that the waits behind it run
  // into their time bound (RR_PGO_FLOW_TIMEOUT_MS).  mode 1: k_factor_flow -- the root front's last foreign child's flag
  // is looked for in a word nobody sets; 2: k_solve_flow -- the same for the parent flag of one front; 3: k_big_flow --
  // the first PANEL task of the first flow level publishes its X blocks in the dead words behind the live flags;
  // 0: everything back.  The captured graphs hold pointers, not contents: no re-capture.
  void debug_withhold(int mode) override {
    HIPCHK(hipStreamSynchronize(stream_));
    if (mode == 0) {
      if (withheld_child_ >= 0) {
        HIPCHK(hipMemcpy(child_dep_.p + withheld_child_, &host_child_dep_[withheld_child_], sizeof(int32_t), hipMemcpyHostToDevice));
        withheld_child_ = -1;
      }
      if (withheld_parent_ >= 0) {
        HIPCHK(hipMemcpy(parent_dep_.p + withheld_parent_, &withheld_parent_val_, sizeof(int32_t), hipMemcpyHostToDevice));
        withheld_parent_ = -1;
      }
      if (withheld_level_ >= 0) {
        FlowRec *dst = withheld_level_ == (1 << 30) ? xl_recs_.p : flow_levels_[withheld_level_]->tasks.p;
        HIPCHK(hipMemcpy(dst + withheld_task_, &withheld_rec_, sizeof(FlowRec), hipMemcpyHostToDevice));
        withheld_level_ = -1;
      }
      return;
    }
    if (mode == 1 || mode == 2) {
      if (!lds_flow_) throw ApiError(RR_PGO_EUNSUPPORTED, "this handle does not run the dataflow launches of the LDS fronts");
      const int32_t dead = 2 * sym_.S + 96;   // inside dep_flags_, behind the words the linearisation zeroes, never set
      if (mode == 1) {
        for (int q = (int)host_child_dep_.size() - 1; q >= 0; q--)
          if (host_child_dep_[q] >= 0) {
            HIPCHK(hipMemcpy(child_dep_.p + q, &dead, sizeof(int32_t), hipMemcpyHostToDevice));
            withheld_child_ = q;
            return;
          }
        throw ApiError(RR_PGO_EUNSUPPORTED, "no front waits for a child of another task");
      }
      std::vector<int32_t> pd(sym_.S);
      HIPCHK(hipMemcpy(pd.data(), parent_dep_.p, pd.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int f = 0; f < sym_.S; f++)
        if (pd[f] >= 0) {
          withheld_parent_ = f;
          withheld_parent_val_ = pd[f];
          HIPCHK(hipMemcpy(parent_dep_.p + f, &dead, sizeof(int32_t), hipMemcpyHostToDevice));
          return;
        }
      throw ApiError(RR_PGO_EUNSUPPORTED, "no front waits for a parent of another task");
    }
    if (mode == 3 && xl_) {
      for (size_t t = 0; t < xl_host_.size(); t++)
        if ((xl_host_[t].t.kind_front >> 24) == FLOW_PANEL) {
          withheld_level_ = 1 << 30;   // (the cross-level list)
          withheld_task_ = (int)t;
          withheld_rec_ = xl_host_[t];
          FlowRec bad = xl_host_[t];
          bad.ff.pf = (int32_t)((int64_t)flow_flags_.n - kDeadFlagWords);
          bad.ff.pstride = 1;
          HIPCHK(hipMemcpy(xl_recs_.p + t, &bad, sizeof(FlowRec), hipMemcpyHostToDevice));
          return;
        }
      throw ApiError(RR_PGO_EUNSUPPORTED, "no PANEL task in the cross-level list");
    }
    if (mode == 3) {
      for (size_t si = 0; si < flow_levels_.size(); si++) {
        if (!flow_levels_[si]) continue;
        FlowLevel &l = *flow_levels_[si];
        std::vector<FlowRec> recs((size_t)l.n_tasks);
        HIPCHK(hipMemcpy(recs.data(), l.tasks.p, recs.size() * sizeof(FlowRec), hipMemcpyDeviceToHost));
        for (int t = 0; t < l.n_tasks; t++)
          if ((recs[t].t.kind_front >> 24) == FLOW_PANEL) {
            withheld_level_ = (int)si;
            withheld_task_ = t;
            withheld_rec_ = recs[t];
            FlowRec bad = recs[t];
            bad.ff.pf = (int32_t)((int64_t)flow_flags_.n - kDeadFlagWords);   // its X flags land in the dead words
            bad.ff.pstride = 1;
            HIPCHK(hipMemcpy(l.tasks.p + t, &bad, sizeof(FlowRec), hipMemcpyHostToDevice));
            return;
          }
      }
      throw ApiError(RR_PGO_EUNSUPPORTED, "this handle has no dataflow level of fronts beyond LDS");
    }
    throw ApiError(RR_PGO_EINVAL, "mode must be 0 .. 3");
  }

  void profile(int iters, double *ms, int64_t *launches) override {
    HIPCHK(hipEventCreate(&prof_.e0));
    HIPCHK(hipEventCreate(&prof_.e1));
    for (int k = 0; k < RR_PGO_NUM_KCLASS; k++) { prof_.ms[k] = 0; prof_.n[k] = 0; }
    prof_.on = true;
    try {
      for (int i = 0; i < iters; i++) enqueue_gn_iteration();
      HIPCHK(hipStreamSynchronize(stream_));
    } catch (...) {
      prof_.on = false;
      throw;
    }
    prof_.on = false;
    (void)hipEventDestroy(prof_.e0);
    (void)hipEventDestroy(prof_.e1);
    for (int k = 0; k < RR_PGO_NUM_KCLASS; k++) { ms[k] = prof_.ms[k]; launches[k] = prof_.n[k]; }
  }
};

}  // namespace rrpgo

// ============================================================== C ABI

using namespace rrpgo;

struct rr_pgo {
  HostGraph g;
  std::shared_ptr<const Symbolic> symp;   // shared with the analysis cache (structurally identical graphs analysed once)
  const Symbolic &sym_ref() const { return *symp; }
  rr_pgo_options opt;
  std::unique_ptr<EngineBase> engine;
  rr_pgo_stats stats;
  std::vector<int32_t> blk_rows, blk_cols;   // assemble() block list
  std::vector<int64_t> blk_offs;
};

struct rr_pgo_synth {
  HostGraph g;
};

namespace {

template <typename F> int guarded(F &&f) {
  try {
    f();
    return RR_PGO_OK;
  } catch (const ApiError &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc &) {
    g_last_error = "out of host memory";
    return RR_PGO_ENOMEM;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return RR_PGO_EINVAL;
  }
}

void fill_desc(const HostGraph &g, rr_pgo_graph_desc *d) {
  d->n_nodes = g.n_nodes();
  d->node_kind = g.node_kind.data();
  d->node_id = g.node_id.data();
  d->node_state = g.node_state.data();
  d->n_edges = g.n_edges();
  d->edge_kind = g.edge_kind.data();
  d->edge_from = g.edge_from.data();
  d->edge_to = g.edge_to.data();
  d->edge_meas = g.edge_meas.data();
  d->edge_info = g.edge_info.data();
}

// ---- analysis cache: the last few analyses of this process, keyed by everything they depend on
struct AnalysisCacheEntry {
  uint64_t key;
  int32_t precision, rank, world_size, sharded;
  std::vector<int32_t> node_kind, edge_kind, edge_from, edge_to;
  std::vector<double> node_state;
  std::string env;
  std::shared_ptr<const Symbolic> sym;
};
static std::mutex g_analysis_mu;
static std::vector<AnalysisCacheEntry> g_analysis_cache;   // most recent last; at most four graphs of at most 50 000 edges
extern "C" char **environ;
static std::string analysis_env() {   // every RR_PGO_* switch of the process (the analysis reads a dozen of them)
  std::string e;
  for (char **p = environ; p && *p; p++)
    if (std::strncmp(*p, "RR_PGO_", 7) == 0) { e += *p; e += '\n'; }
  return e;
}
static uint64_t fnv(uint64_t h, const void *data, size_t n) {
  const unsigned char *p = (const unsigned char *)data;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
static uint64_t analysis_cache_key(const HostGraph &g, const rr_pgo_options &opt) {
  if (const char *e = std::getenv("RR_PGO_ANALYSIS_CACHE")) if (std::atoi(e) == 0) return 0;
  if (g.n_edges() > 50000 || g.n_edges() == 0) return 0;   // (the reference's datasets: at most 17 605 edges; the tables of a large graph are not worth keeping)
  uint64_t h = 1469598103934665603ull;
  const int32_t o[4] = {opt.precision, opt.rank, opt.world_size, opt.sharded};
  h = fnv(h, o, sizeof o);
  h = fnv(h, g.node_kind.data(), g.node_kind.size() * 4);
  h = fnv(h, g.edge_kind.data(), g.edge_kind.size() * 4);
  h = fnv(h, g.edge_from.data(), g.edge_from.size() * 4);
  h = fnv(h, g.edge_to.data(), g.edge_to.size() * 4);
  h = fnv(h, g.node_state.data(), g.node_state.size() * 8);
  return h ? h : 1;
}
static bool analysis_same(const AnalysisCacheEntry &c, uint64_t key, const HostGraph &g, const rr_pgo_options &opt, const std::string &env) {
  return c.key == key && c.precision == opt.precision && c.rank == opt.rank && c.world_size == opt.world_size && c.sharded == opt.sharded &&
         c.node_kind == g.node_kind && c.edge_kind == g.edge_kind && c.edge_from == g.edge_from && c.edge_to == g.edge_to && c.node_state == g.node_state && c.env == env;
}
static std::shared_ptr<const Symbolic> analysis_cache_find(uint64_t key, const HostGraph &g, const rr_pgo_options &opt) {
  const std::string env = analysis_env();
  std::lock_guard<std::mutex> lk(g_analysis_mu);
  for (const AnalysisCacheEntry &c : g_analysis_cache)
    if (analysis_same(c, key, g, opt, env)) return c.sym;
  return nullptr;
}
static void analysis_cache_store(uint64_t key, const HostGraph &g, const rr_pgo_options &opt, std::shared_ptr<const Symbolic> sym) {
  AnalysisCacheEntry c{key, opt.precision, opt.rank, opt.world_size, opt.sharded, g.node_kind, g.edge_kind, g.edge_from, g.edge_to, g.node_state, analysis_env(), std::move(sym)};
  std::lock_guard<std::mutex> lk(g_analysis_mu);
  if (g_analysis_cache.size() >= 4) g_analysis_cache.erase(g_analysis_cache.begin());
  g_analysis_cache.push_back(std::move(c));
}

// host part of PoseGraph::new: options, symbolic analysis, the statistics that need no device
void analyze_handle(std::unique_ptr<rr_pgo> &h, const rr_pgo_options *opt_in, double parse_ms) {
  rr_pgo_options opt;
  if (opt_in) opt = *opt_in; else rr_pgo_default_options(&opt);
  h->opt = opt;
  if (opt.precision != RR_PGO_F64 && opt.precision != RR_PGO_F32 && opt.precision != RR_PGO_MIXED)
    throw ApiError(RR_PGO_EINVAL, "bad precision");
  if (opt.world_size > 1) {
    if (opt.world_size & (opt.world_size - 1)) throw ApiError(RR_PGO_EINVAL, "world_size must be a power of two");
    if (opt.world_size > 64) throw ApiError(RR_PGO_EINVAL, "world_size must be at most 64");
    if (opt.rank < 0 || opt.rank >= opt.world_size) throw ApiError(RR_PGO_EINVAL, "rank out of range");
  }
  // symbolic analysis (host only)
  SymbolicOptions so;
  so.lds_budget_elems = opt.precision == RR_PGO_F64 ? 19000 : 38000;
  // nested dissection down to leaves of this many nodes on large graphs (measured on the 400 x 250 lattice, r02:
  // fp32 6.05 ms per step with 48, 6.09 with 40, 6.12 with 56, 6.21 with 64; fp64 -- half the LDS budget per front --
  // 9.78 with 32, 10.41 with 48 or 64)
  const int big_leaf = opt.precision == RR_PGO_F64 ? 32 : 48;
  so.nd_leaf = h->g.n_nodes() <= 6000 ? (1 << 30) : big_leaf;
  so.split_separators = h->g.n_nodes() > 6000;   // wide top fronts: see symbolic.cpp, supernode pass
  if (opt.world_size > 1) {   // sharding needs the nested-dissection top levels
    so.n_parts = opt.world_size;
    so.my_part = opt.rank;
    so.nd_leaf = big_leaf;
    so.pin_node = h->g.anchor_node;   // every rank needs the anchor's entries of the solution (gauge transfer)
  }
  // tuning knobs of the symbolic phase
  // graphs whose fronts all fit LDS: ONE dataflow launch for the factorisation, one for the back substitution (lds_flow.hip.h);
  // RR_PGO_LDS_FLOW=0 keeps one launch per level of the task tree (the parity alternative: bit-identical results)
  so.lds_flow = opt.world_size <= 1 && !opt.sharded;
  if (const char *e = std::getenv("RR_PGO_LDS_FLOW")) so.lds_flow = so.lds_flow && std::atoi(e) != 0;
  if (const char *e = std::getenv("RR_PGO_TASK_US")) so.task_us = std::atof(e);
  if (std::getenv("RR_PGO_NO_GEO")) so.geo_nd = false;
  // small graphs (trajectories with loop closures): the multilevel bisection finds narrower separators (intel: 97 -> 60 nodes along the heaviest root path) than the
  // level sets / coordinate cuts (symbolic.cpp, MultilevelBisection); the lattice's straight cuts are already the best there are
  so.ml_nd = h->g.n_nodes() <= 6000;
  if (const char *e = std::getenv("RR_PGO_ML_ND")) so.ml_nd = std::atoi(e) != 0;
  if (const char *e = std::getenv("RR_PGO_LDS_PIECES")) so.max_lds_pieces = std::max(1, std::atoi(e));
  // RR_PGO_JOIN_SEPARATORS=1: a region's last separator always chained into its parent separator's supernode; =0: never (small graphs:
  // no longer a choice of the candidates below)
  const char *join_env = std::getenv("RR_PGO_JOIN_SEPARATORS");
  if (join_env) so.split_separators = std::atoi(join_env) == 0 && join_env[0] == '0';
  if (const char *e = std::getenv("RR_PGO_ND_LEAF")) so.nd_leaf = std::atoi(e);
  if (const char *e = std::getenv("RR_PGO_BALANCE_BLOCKS")) { so.balance_blocks = std::atoi(e) != 0; if (std::atoi(e) > 1) so.balance_max_rem = std::atoi(e); }
  // the widest front the chain pass may merge into its parent (when the cost model says the parent finishes earlier): 80 pivot
  // columns for 2D graphs, 48 for 6 x 6 blocks -- re-measured on the r05 / r06 trees (profiles/r06_chain_cap_sweep.txt: intel + 3.1 %,
  // M3500 - 0.3 %, dlr + 0.0 % at 80; the SE(3) graphs lose 0.4 - 2 % beyond 48, where the model's error grows with the front)
  if (!h->g.has_se3) so.merge_chain_nc = 80;
  if (const char *e = std::getenv("RR_PGO_MERGE_CHAIN")) { so.merge_chain_nc = std::atoi(e); if (const char *c = std::strchr(e, ',')) so.merge_chain_gain_us = std::atof(c + 1); }
  if (const char *e = std::getenv("RR_PGO_AMALG_NP")) so.amalg_np = std::atoi(e);
  double t0 = now_ms();
  std::string err;
  // The analysis is a function of the graph's structure (and, through the coordinate cuts, of the initial positions), the options
  // and the switches above -- not of the measurements.  A caller that builds the same graph again (the reference's own bench is a
  // loop of PoseGraph::new(file) + optimize(10), benches/graph_slam.rs:9-10; UMFPACK users keep the symbolic object for the same
  // reason) gets the tables of the first analysis.  RR_PGO_ANALYSIS_CACHE=0: every handle is analysed afresh.
  const uint64_t cache_key = analysis_cache_key(h->g, opt);
  if (cache_key != 0)
    if (std::shared_ptr<const Symbolic> hit = analysis_cache_find(cache_key, h->g, opt)) h->symp = hit;
  Symbolic fresh;
  if (h->symp) {
  } else
  if (h->g.n_nodes() <= 6000 && opt.world_size <= 1 && !opt.sharded && !(std::getenv("RR_PGO_ND_LEAF") && std::getenv("RR_PGO_AMALG_NP"))) {
    // Small graphs are bound by the critical path through the supernode tree, not by flops: a few
    // nested-dissection cuts above minimum-degree leaves shorten that path on the larger ones (M3500, dlr,
    // sphere2500: +20..26 % measured) and lengthen it on intel; merging mid-sized fronts (relaxed amalgamation up to 72
    // pivot columns) pays on sphere2500 and costs 2 - 4 % on intel, M3500 and dlr now that the fronts on the chain are merged
    // by their own rule (symbolic.cpp, step 5).  The front cost model ranks the candidates the way the measurements do, so
    // the estimated critical path picks the leaf size and the amalgamation width.  Below 2400 nodes no cut ever won.
    // The candidates are independent host computations: one thread each (r04: the six analyses in sequence were 35 ms
    // of M3500's and 55 ms of dlr's constructor on the GPU box, against 2 - 3 ms of optimize(10)).
    static const int kLeafLevelSets[] = {1 << 30, 3000, 2000, 1400, 1000, 700};
    static const int kLeafMultilevel[] = {1 << 30, 250, 150, 100, 70, 50};   // (500 never won on any graph of the test set)
    struct Cand { int leaf, np; bool split; };
    const bool np_fixed = std::getenv("RR_PGO_AMALG_NP") != nullptr;
    std::vector<Cand> cl;
    for (int li = 0; li < 6; li++) {
      int leaf = so.ml_nd ? kLeafMultilevel[li] : kLeafLevelSets[li];
      if (std::getenv("RR_PGO_ND_LEAF")) { if (leaf != (1 << 30)) continue; leaf = so.nd_leaf; }
      else if (leaf != (1 << 30) && ((!so.ml_nd && h->g.n_nodes() < 2400) || leaf >= h->g.n_nodes())) continue;   // (a leaf size >= the graph is no cut at all)
      // (with the multilevel bisection the undissected tree lost on every graph of 1000+ nodes by 30 - 60 % of the estimate, and its
      // minimum-degree pass over the whole graph is the slowest of the candidate analyses: 5.8 ms on dlr)
      if (so.ml_nd && leaf == (1 << 30) && h->g.n_nodes() >= 1000 && !std::getenv("RR_PGO_ND_LEAF")) continue;
      cl.push_back({leaf, np_fixed ? so.amalg_np : 16, so.split_separators});   // the narrow rule first: it wins wherever every front lives in LDS
    }
    const size_t n_depths = cl.size();
    // the deepest dissections once more with mid-sized fronts merged up to 32 columns (intel, r05: the model's and the measured best)
    if (so.ml_nd && !np_fixed && !std::getenv("RR_PGO_ND_LEAF"))
      for (int leaf : {70, 50})
        if (leaf < h->g.n_nodes()) cl.push_back({leaf, 32, so.split_separators});
    // ... and the deeper dissections with a region's last separator NOT chained into its parent separator's supernode (the rule of the
    // large graphs, symbolic.cpp step 4): the sibling separator then runs beside it instead of before it.  Measured (r05, same kernels):
    // sphere2500 1539 -> 1697 it/s, torus3D 1066 -> 1191, dlr 7369 -> 7852, intel 7382 -> 6962 -- and the estimates say so beforehand
    // (-8 %, -5 %, -3 %, +7 %), so the candidates carry both forms and the model picks.
    const bool split_free = so.ml_nd && !join_env && !std::getenv("RR_PGO_ND_LEAF");
    if (split_free)
      for (int leaf : {100, 70, 50})
        if (leaf < h->g.n_nodes()) cl.push_back({leaf, np_fixed ? so.amalg_np : 16, true});
    Symbolic best;
    double best_crit = -1.0;
    int best_leaf = 0;
    // the candidates differ in where they STOP dissecting, not in how a set is split: the deepest dissection runs once (its halves on
    // threads of their own) and records its splits, every candidate replays them (symbolic.h, NdSplitTable)
    NdSplitTable splits;
    bool shared_splits = false;
    if (so.ml_nd && cl.size() > 1) {
      SymbolicOptions o = so;
      o.nd_leaf = 1 << 30;
      for (const Cand &c : cl) o.nd_leaf = std::min(o.nd_leaf, c.leaf);
      o.nd_record = &splits;
      const double td = now_ms();
      err = dissect_only(h->g, o);
      shared_splits = err.empty();
      if (std::getenv("RR_PGO_ANALYZE_TIMES")) std::fprintf(stderr, "analyze: shared dissection down to %d nodes: %.3f ms (%zu splits)\n", o.nd_leaf, now_ms() - td, splits.map.size());
    }
    auto run = [&](const std::vector<Cand> &list) {
      std::vector<Symbolic> cands(list.size());
      std::vector<std::string> errs(list.size());
      // (one candidate per thread, never more threads than cores: host_threads.h)
      parallel_indices((int)list.size(), (int)list.size(), [&](int c) {
        SymbolicOptions o = so;
        o.nd_leaf = list[c].leaf;
        o.amalg_np = list[c].np;
        o.split_separators = list[c].split;
        if (shared_splits) o.nd_replay = &splits;
        try { errs[c] = analyze(h->g, o, cands[c]); } catch (const std::exception &e) { errs[c] = e.what(); }
      });
      for (size_t c = 0; c < list.size(); c++) {
        if (!errs[c].empty()) { err = errs[c]; return; }
        if (std::getenv("RR_PGO_ANALYZE_TIMES"))
          std::fprintf(stderr, "analyze: nd_leaf %d, amalgamation up to %d columns, separators %s -> estimated critical path %.1f us (%d big fronts, %d supernodes)\n", list[c].leaf,
                       list[c].np, list[c].split ? "apart" : "chained", cands[c].est_critical_us, cands[c].n_big, cands[c].S);
        // a tree with a front beyond LDS below an LDS front loses the dataflow launches (one launch per level instead, and no
        // cross-level launch for its big fronts): the estimate does not see that, the measurement does (torus3D, r05: the
        // candidate with the smallest estimate ran 40 launches per iteration and 2 % slower than r04's tree)
        const double eff = cands[c].est_critical_us * (so.lds_flow && !cands[c].lds_flow ? 1.15 : 1.0);
        if (best_crit < 0 || eff < best_crit) { best_crit = eff; best_leaf = list[c].leaf; best = std::move(cands[c]); }
      }
    };
    if (err.empty()) run(cl);
    // fronts beyond LDS (sphere2500, torus3D: 6 x 6 blocks): there the r01 rule -- mid-sized fronts merge up to 72 columns --
    // still pays, and with another depth than the narrow rule's (torus3D): a second round of the same depths decides
    if (err.empty() && !np_fixed && best.n_big > 0) {
      cl.resize(n_depths);
      for (Cand &c : cl) c.np = 72;
      if (split_free)
        for (size_t i = 0; i < n_depths; i++)
          if (cl[i].leaf != (1 << 30)) cl.push_back({cl[i].leaf, 72, true});
      run(cl);
    }
    (void)best_leaf;
    if (err.empty()) fresh = std::move(best);
  } else {
    err = analyze(h->g, so, fresh);
    if (err.empty() && std::getenv("RR_PGO_ANALYZE_TIMES"))
      std::fprintf(stderr, "analyze: estimated critical path %.1f us (%d big fronts, %d supernodes)\n", fresh.est_critical_us, fresh.n_big, fresh.S);
  }
  if (!err.empty()) throw ApiError(RR_PGO_EINVAL, err);
  if (!h->symp) {
    h->symp = std::make_shared<const Symbolic>(std::move(fresh));
    if (cache_key != 0) analysis_cache_store(cache_key, h->g, opt, h->symp);
  }
  double t1 = now_ms();
  rr_pgo_stats &s = h->stats;
  std::memset(&s, 0, sizeof s);
  const Symbolic &y = h->sym_ref();
  const double sz = opt.precision == RR_PGO_F64 ? 8.0 : 4.0;
  int64_t diag_elems = 0, off_elems = 0;
  for (int i = 0; i < y.N; i++) { int d = node_dim(h->g.node_kind[i]); diag_elems += d * d; }
  off_elems = y.n_hvals - diag_elems;
  s.nnz_h_blocks = y.N + y.n_offblocks;
  s.nnz_l_scalars = y.l_elems;
  s.factor_flops = y.factor_flops;
  s.n_supernodes = y.S;
  s.n_levels = (int)y.steps.size();
  s.max_front = y.max_front;
  s.max_pivot_cols = y.max_pivot_cols;
  s.n_big_fronts = y.n_big;
  s.analyze_ms = t1 - t0;
  s.parse_ms = parse_ms;
  s.abi_version = RR_PGO_ABI_VERSION;
  s.lds_dataflow = y.lds_flow ? 1 : 0;
  // algorithmic bytes per GN iteration, SURVEY.md 8(d): every datum moved once.  The factor is priced at its NONZEROS,
  // nnzblk(L) * d^2 scalars (Symbolic::nnz_l_entries) -- written once by the factorisation, read by the forward and by
  // the backward substitution -- not at what is stored: the supernodal panels carry padding and a front beyond LDS keeps
  // its whole M x M square in place (stored_factor_bytes; 2.2 x the nonzeros on the 1M-edge lattice)
  const double dim = y.dim;
  double edge_stream = 0;
  for (int k = 0; k < h->g.n_edges(); k++)
    edge_stream += 8.0 + sz * (edge_meas_len(h->g.edge_kind[k]) + edge_info_len(h->g.edge_kind[k]));
  s.bytes_linearize = edge_stream + dim * sz /*poses*/ + (diag_elems + off_elems) * sz + dim * sz;
  s.bytes_chi2 = 0;  // fused into the linearisation pass
  s.bytes_factor = (diag_elems + off_elems) * sz + (double)y.nnz_l_entries * sz;
  s.bytes_solve = 2.0 * (double)y.nnz_l_entries * sz + 4.0 * dim * sz;
  s.bytes_update = 3.0 * dim * sz;
  s.stored_factor_bytes = (double)y.l_elems * sz;
}

void build_handle(std::unique_ptr<rr_pgo> &h, const rr_pgo_options *opt_in, double parse_ms) {
  analyze_handle(h, opt_in, parse_ms);
  const rr_pgo_options opt = h->opt;
  // device
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    throw ApiError(RR_PGO_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
  if (opt.device >= 0) {
    if (opt.device >= ndev) throw ApiError(RR_PGO_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(opt.device));
  }
  const int wr = opt.world_size > 1 ? opt.rank : 0, ww = opt.world_size > 1 ? opt.world_size : 1;
  const bool shd = opt.sharded != 0;
  if (opt.precision == RR_PGO_F64) h->engine = std::make_unique<Engine<double>>(h->g, h->sym_ref(), wr, ww, shd);
  else if (opt.precision == RR_PGO_F32) h->engine = std::make_unique<Engine<float>>(h->g, h->sym_ref(), wr, ww, shd);
  else h->engine = std::make_unique<Engine<float, double>>(h->g, h->sym_ref(), wr, ww, shd);
  // stats
  rr_pgo_stats &s = h->stats;
  const Symbolic &y = h->sym_ref();
  s.n_launches_per_iter = h->engine->n_launches_per_iter;
  // trailing updates of the huge fronts: a super-panel of w columns updates the lower triangle of the T rows to its
  // right, w * T (T + 1) / 2 multiply-adds -- counted for the launches of k_big_update and, separately, for the tiles
  // that run inside k_big_flow launches.  (The updates INSIDE a super-panel belong to the panel kernels / PANEL tasks
  // and are not counted here.)
  double buf = 0, bflow = 0;
  std::vector<char> in_flow(y.S, 0);
  h->engine->mark_flow_fronts(in_flow);
  const int sch_tile = h->engine->schur_tile();   // 0: no split (flow levels that keep their Schur tiles are marked 2 in in_flow)
  auto schur_origin_fn = [&](int nc) { return (double)big_schur_origin(nc, sch_tile); };
  const std::function<double(int)> schur_origin_of = sch_tile ? std::function<double(int)>(schur_origin_fn) : std::function<double(int)>();
  for (int f = 0; f < y.S; f++)
    if (y.sn_huge[f]) {
      const double M = y.sn_ncols[f] + y.sn_nrows[f] + 1;
      for (int k0 = 0; k0 < y.sn_ncols[f]; k0 += BIG_SUPER) {
        const double w = std::min<int>(BIG_SUPER, y.sn_ncols[f] - k0), T = M - (k0 + w);
        // with the Schur split the tiles from the Schur origin on belong to k_big_schur (a k_big_update-class launch) on every level
        const double Ts = (schur_origin_of && in_flow[f] != 2) ? std::max(0.0, M - std::max<double>(schur_origin_of(y.sn_ncols[f]), k0 + w)) : 0.0;
        buf += w * Ts * (Ts + 1);
        (in_flow[f] ? bflow : buf) += w * (T * (T + 1) - Ts * (Ts + 1));
      }
    }
  s.big_update_flops = buf;
  s.big_flow_flops = bflow;
}

}  // namespace

extern "C" {

void rr_pgo_default_options(rr_pgo_options *opt) {
  std::memset(opt, 0, sizeof *opt);
  opt->precision = RR_PGO_F64;
  opt->device = -1;
  opt->solver = RR_PGO_GAUSS_NEWTON;
  opt->rank = 0;
  opt->world_size = 1;
}

const char *rr_pgo_last_error(void) { return g_last_error.c_str(); }

int rr_pgo_load_g2o(const char *path, const rr_pgo_options *opt, rr_pgo **out) {
  if (!path || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  *out = nullptr;
  return guarded([&] {
    auto h = std::make_unique<rr_pgo>();
    bool io = false;
    double t0 = now_ms();
    std::string err = load_g2o(path, h->g, io);
    if (!err.empty()) throw ApiError(io ? RR_PGO_EIO : RR_PGO_EPARSE, err);
    build_handle(h, opt, now_ms() - t0);
    *out = h.release();
  });
}

int rr_pgo_create(const rr_pgo_graph_desc *d, const rr_pgo_options *opt, rr_pgo **out) {
  if (!d || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  *out = nullptr;
  return guarded([&] {
    auto h = std::make_unique<rr_pgo>();
    HostGraph &g = h->g;
    if (d->n_nodes < 0 || d->n_edges < 0) throw ApiError(RR_PGO_EINVAL, "negative counts");
    if (d->n_nodes > 0 && (!d->node_kind || !d->node_state)) throw ApiError(RR_PGO_EINVAL, "null node arrays");
    if (d->n_edges > 0 && (!d->edge_kind || !d->edge_from || !d->edge_to || !d->edge_meas || !d->edge_info))
      throw ApiError(RR_PGO_EINVAL, "null edge arrays");
    g.node_kind.assign(d->node_kind, d->node_kind + d->n_nodes);
    g.node_id.resize(d->n_nodes);
    for (int i = 0; i < d->n_nodes; i++) g.node_id[i] = d->node_id ? d->node_id[i] : (uint32_t)i;
    size_t ns = 0;
    for (int i = 0; i < d->n_nodes; i++) {
      if (g.node_kind[i] < 0 || g.node_kind[i] > 2) throw ApiError(RR_PGO_EINVAL, "bad node kind");
      ns += node_state_len(g.node_kind[i]);
    }
    g.node_state.assign(d->node_state, d->node_state + ns);
    g.edge_kind.assign(d->edge_kind, d->edge_kind + d->n_edges);
    g.edge_from.assign(d->edge_from, d->edge_from + d->n_edges);
    g.edge_to.assign(d->edge_to, d->edge_to + d->n_edges);
    size_t nm = 0, ni = 0;
    for (int k = 0; k < d->n_edges; k++) {
      if (g.edge_kind[k] < 0 || g.edge_kind[k] > 2) throw ApiError(RR_PGO_EINVAL, "bad edge kind");
      nm += edge_meas_len(g.edge_kind[k]);
      ni += edge_info_len(g.edge_kind[k]);
    }
    g.edge_meas.assign(d->edge_meas, d->edge_meas + nm);
    g.edge_info.assign(d->edge_info, d->edge_info + ni);
    std::string err = g.finalize();
    if (!err.empty()) throw ApiError(RR_PGO_EINVAL, err);
    build_handle(h, opt, 0.0);
    *out = h.release();
  });
}

void rr_pgo_destroy(rr_pgo *h) { delete h; }

int32_t rr_pgo_num_nodes(const rr_pgo *h) { return h ? h->g.n_nodes() : 0; }
int32_t rr_pgo_num_edges(const rr_pgo *h) { return h ? h->g.n_edges() : 0; }
int32_t rr_pgo_dim(const rr_pgo *h) { return h ? h->g.dim : 0; }
int32_t rr_pgo_state_len(const rr_pgo *h) { return h ? (int32_t)h->g.node_state.size() : 0; }
int32_t rr_pgo_anchor_node(const rr_pgo *h) { return h ? h->g.anchor_node : -1; }

int rr_pgo_get_graph(const rr_pgo *h, rr_pgo_graph_desc *out) {
  if (!h || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  fill_desc(h->g, out);
  return RR_PGO_OK;
}

int rr_pgo_chi2(rr_pgo *h, double *out) {
  if (!h || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->chi2(out); });
}

int rr_pgo_linearize_solve(rr_pgo *h, double lambda, int lm, double *dx_out) {
  if (!h || !dx_out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->linearize_solve(lambda, lm, dx_out); });
}

int rr_pgo_update(rr_pgo *h, const double *dx, double sign) {
  if (!h || !dx) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->update(dx, sign); });
}

int rr_pgo_optimize(rr_pgo *h, int32_t iters, double *errors, int32_t *n_errors, double *norms) {
  if (!h || !errors || !n_errors || iters < 0) { g_last_error = "bad argument"; return RR_PGO_EINVAL; }
  *n_errors = 0;
  return guarded([&] {
    int ne = 0;
    h->engine->optimize(h->opt.solver, iters, errors, &ne, norms);
    *n_errors = ne;
  });
}

int rr_pgo_get_state(rr_pgo *h, double *out) {
  if (!h || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->get_state(out); });
}

int rr_pgo_set_state(rr_pgo *h, const double *st) {
  if (!h || !st) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->set_state(st); });
}

int rr_pgo_assemble(rr_pgo *h, double lambda, int lm, int32_t *n_blocks, int32_t *brow, int32_t *bcol,
                    int64_t *boff, double *bvals, int64_t *n_vals, double *b_out) {
  if (!h) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  const Symbolic &y = h->sym_ref();
  const int64_t nb = y.N + y.n_offblocks;
  if (n_blocks) *n_blocks = (int32_t)nb;
  if (n_vals) *n_vals = y.n_hvals;
  if (!bvals) return RR_PGO_OK;
  return guarded([&] {
    std::vector<double> hv, b;
    h->engine->assemble(lambda, lm, hv, b);
    for (int i = 0; i < y.N; i++) {
      int v = y.order[i];
      if (brow) brow[i] = v;
      if (bcol) bcol[i] = v;
      if (boff) boff[i] = y.diag_off[v];
    }
    for (int64_t s = 0; s < y.n_offblocks; s++) {
      if (brow) brow[y.N + s] = y.blk_row[s];
      if (bcol) bcol[y.N + s] = y.blk_col[s];
      if (boff) boff[y.N + s] = y.blk_off[s];
    }
    std::memcpy(bvals, hv.data(), hv.size() * sizeof(double));
    if (b_out) std::memcpy(b_out, b.data(), b.size() * sizeof(double));
  });
}

int rr_pgo_iterate_async(rr_pgo *h, int32_t iters) {
  if (!h || iters < 0) { g_last_error = "bad argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->iterate_async(iters); });
}

int rr_pgo_sync(rr_pgo *h) {
  if (!h) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->sync(); });
}

int32_t rr_pgo_abi_version(void) { return RR_PGO_ABI_VERSION; }

int rr_pgo_trim(void) {
  chunk_pool().trim();
  stream_pool().trim();
  std::lock_guard<std::mutex> lk(g_analysis_mu);
  g_analysis_cache.clear();
  return RR_PGO_OK;
}

int rr_pgo_debug_withhold(rr_pgo *h, int32_t mode) {
  if (!h) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->debug_withhold(mode); });
}

int rr_pgo_analyze_g2o(const char *path, const rr_pgo_options *opt, rr_pgo_stats *out) {
  if (!path || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] {
    auto h = std::make_unique<rr_pgo>();
    const double t0 = now_ms();
    bool io_error = false;
    const std::string err = load_g2o(path, h->g, io_error);
    if (!err.empty()) throw ApiError(io_error ? RR_PGO_EIO : RR_PGO_EPARSE, err);
    analyze_handle(h, opt, now_ms() - t0);
    *out = h->stats;
  });
}

int rr_pgo_get_stats(const rr_pgo *h, rr_pgo_stats *out) {
  if (!h || !out) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  *out = h->stats;
  return RR_PGO_OK;
}

int rr_pgo_profile(rr_pgo *h, int32_t iters, double *ms_total, int64_t *launches, int32_t n_classes) {
  if (!h || !ms_total || !launches || iters < 0 || n_classes < 0) { g_last_error = "bad argument"; return RR_PGO_EINVAL; }
  return guarded([&] {
    double ms[RR_PGO_NUM_KCLASS];
    int64_t n[RR_PGO_NUM_KCLASS];
    h->engine->profile(iters, ms, n);
    for (int k = 0; k < std::min<int>(n_classes, RR_PGO_NUM_KCLASS); k++) { ms_total[k] = ms[k]; launches[k] = n[k]; }
  });
}

int rr_pgo_synth_grid(int32_t width, int32_t height, int64_t n_edges_target, uint64_t seed_meas,
                      uint64_t seed_init, rr_pgo_synth **out, rr_pgo_graph_desc *desc) {
  if (!out || !desc || width < 2 || height < 2) { g_last_error = "bad argument"; return RR_PGO_EINVAL; }
  return guarded([&] {
    auto s = std::make_unique<rr_pgo_synth>();
    synth_grid(width, height, n_edges_target, seed_meas, seed_init, s->g);
    fill_desc(s->g, desc);
    *out = s.release();
  });
}

void rr_pgo_synth_free(rr_pgo_synth *s) { delete s; }

int rr_pgo_exchange_buffer(rr_pgo *h, int32_t which, void **dev_ptr, int64_t *n_elems, int32_t *elem_size) {
  if (!h || !dev_ptr || !n_elems || !elem_size) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->exchange_buffer(which, dev_ptr, n_elems, elem_size); });
}
int rr_pgo_set_exchange_buffer(rr_pgo *h, int32_t which, void *dev_ptr, int64_t n_elems) {
  if (!h || !dev_ptr) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->set_exchange_buffer(which, dev_ptr, n_elems); });
}
int rr_pgo_stage(rr_pgo *h, int32_t stage, double lambda, int lm) {
  if (!h) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->stage(stage, lambda, lm); });
}
int rr_pgo_stage_scalars(rr_pgo *h, double *chi2, double *norm_dx) {
  if (!h || !chi2 || !norm_dx) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  return guarded([&] { h->engine->read_last_scalars(chi2, norm_dx); });
}
#ifdef RRPGO_FLOW_TRACE
// Diagnostic build only: per supernode (parent, pivot columns, rows below, step) -- out[S][4]
extern "C" int32_t rr_pgo_debug_sn_info(rr_pgo *h, int32_t *out) {
  if (!h) return -1;
  const Symbolic &y = h->sym_ref();
  if (out) {
    std::vector<int> step_of(y.S, -1);
    for (size_t si = 0; si < y.steps.size(); si++)
      for (int t = y.steps[si].task_begin; t < y.steps[si].task_end; t++)
        for (int q = y.task_ptr[t]; q < y.task_ptr[t + 1]; q++) step_of[y.task_sn[q]] = (int)si;
    for (int s2 = 0; s2 < y.S; s2++) { out[4 * s2] = y.sn_parent[s2]; out[4 * s2 + 1] = y.sn_ncols[s2]; out[4 * s2 + 2] = y.sn_nrows[s2]; out[4 * s2 + 3] = step_of[s2]; }
  }
  return y.S;
}
// Diagnostic build only (make ../librr_pgo_trace.so): task list (4 ints per task) and stamps (4 waves x 4 per task) of a flow level
extern "C" int64_t rr_pgo_debug_flow_trace(rr_pgo *h, int32_t level, int32_t *tasks, unsigned long long *stamps, int64_t cap_tasks, int32_t *nf, double *est_us) {
  if (!h) return -1;
  int64_t n = -1;
  guarded([&] {
    std::vector<int32_t> t;
    std::vector<unsigned long long> st;
    n = h->engine->flow_trace(level, t, st, nf, est_us);
    if (n < 0 || !tasks || !stamps) return;
    const int64_t m = std::min<int64_t>(n, cap_tasks);
    std::memcpy(tasks, t.data(), (size_t)m * 4 * sizeof(int32_t));
    std::memcpy(stamps, st.data(), (size_t)m * 16 * sizeof(unsigned long long));
  });
  return n;
}
#endif
#ifdef RRPGO_STAMPS
// Diagnostic build only: the launch trace region behind the stamps (count, pad, then (tag, clock) pairs).
extern "C" int64_t rr_pgo_debug_trace(rr_pgo *h, unsigned long long *out, int64_t cap) {
  if (!h) return -1;
  int64_t n = 0;
  guarded([&] {
    std::vector<unsigned long long> st;
    h->engine->read_stamps(st);
    const size_t off = (size_t)h->sym_ref().S * 12;
    n = std::min<int64_t>(cap, (int64_t)(st.size() - off));
    for (int64_t i = 0; i < n; i++) out[i] = st[off + i];
  });
  return n;
}
// Diagnostic build only: per-supernode phase stamps of the last factorisation
// plus the supernode -> (step, task, ncols, nrows) map.  out: [S][16] doubles.
int rr_pgo_debug_stamps(rr_pgo *h, double *out, int32_t *n_sn) {
  if (!h) return RR_PGO_EINVAL;
  return guarded([&] {
    const Symbolic &y = h->sym_ref();
    *n_sn = y.S;
    if (!out) return;
    std::vector<unsigned long long> st;
    h->engine->read_stamps(st);
    std::vector<int> step_of(y.S, -1), task_of(y.S, -1);
    for (size_t si = 0; si < y.steps.size(); si++) {
      const Step &sp = y.steps[si];
      for (int t = sp.task_begin; t < sp.task_end; t++)   // (a level of fronts beyond LDS: one front per task)
        for (int q = y.task_ptr[t]; q < y.task_ptr[t + 1]; q++) { step_of[y.task_sn[q]] = (int)si; task_of[y.task_sn[q]] = t; }
    }
    for (int s = 0; s < y.S; s++) {
      double *o = out + (size_t)s * 16;
      o[0] = step_of[s]; o[1] = task_of[s]; o[2] = y.sn_ncols[s]; o[3] = y.sn_nrows[s];
      o[4] = y.child_ptr[s + 1] - y.child_ptr[s];
      for (int q = 0; q < 10; q++) o[5 + q] = (double)st[(size_t)s * 12 + q];
      o[15] = y.sn_parent[s];
    }
  });
}
// Diagnostic build only: the raw stamps ([S][12], 100 MHz ticks) + per front: parent, task (= ticket of the dataflow
// schedule), pivot columns, rows below, children.  out: [S][20] doubles.
int rr_pgo_debug_stamps12(rr_pgo *h, double *out, int32_t *n_sn) {
  if (!h) return RR_PGO_EINVAL;
  return guarded([&] {
    const Symbolic &y = h->sym_ref();
    *n_sn = y.S;
    if (!out) return;
    std::vector<unsigned long long> st;
    h->engine->read_stamps(st);
    std::vector<int> task_of(y.S, -1);
    for (size_t t = 0; t + 1 < y.task_ptr.size(); t++)
      for (int q = y.task_ptr[t]; q < y.task_ptr[t + 1]; q++) task_of[y.task_sn[q]] = (int)t;
    for (int s = 0; s < y.S; s++) {
      double *o = out + (size_t)s * 20;
      for (int q = 0; q < 12; q++) o[q] = (double)st[(size_t)s * 12 + q];
      o[12] = y.sn_parent[s]; o[13] = task_of[s]; o[14] = y.sn_ncols[s]; o[15] = y.sn_nrows[s];
      o[16] = y.child_ptr[s + 1] - y.child_ptr[s]; o[17] = y.lds_flow ? 1 : 0; o[18] = o[19] = 0;
    }
  });
}
#endif

void *rr_pgo_stream(rr_pgo *h) { return h && h->engine ? (void *)h->engine->stream() : nullptr; }

int rr_pgo_node_owner(const rr_pgo *h, int32_t *owner) {
  if (!h || !owner) { g_last_error = "null argument"; return RR_PGO_EINVAL; }
  const int ws = h->opt.world_size > 1 ? h->opt.world_size : 1;
  for (int i = 0; i < h->g.n_nodes(); i++) owner[i] = ws > 1 ? h->sym_ref().node_part[i] : 0;
  return RR_PGO_OK;
}

}  // extern "C"
