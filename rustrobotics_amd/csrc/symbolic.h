// symbolic.h -- one-off symbolic analysis of the normal matrix H = J^T W J.
//
// The reference redoes ordering + symbolic + numeric factorisation inside
// UMFPACK on every Gauss-Newton iteration (pose_graph_optimization.rs:130-141)
// although the sparsity pattern never changes.  Here the pattern work is done
// ONCE per graph on the host and turned into flat tables the HIP kernels walk:
//
//   node graph -> nested-dissection ordering (constrained minimum degree in the
//   leaves) -> elimination tree -> relaxed supernodes -> multifrontal fronts
//   (pivot panel + update matrix) -> a level schedule of workgroup tasks.
//
// Everything is at NODE (block) granularity with per-node scalar dims (3 for
// SE2, 2 for XY landmarks, 6 for SE3), expanded to scalar indices at the end.
#pragma once
#include <cstdint>
#include <vector>
#include <unordered_map>
#include <mutex>

#include "host_graph.h"

namespace rrpgo {

// The splits of one nested dissection, for the candidates of one graph that differ in where they STOP dissecting (pgo_api.hip):
// recorded by the deepest, replayed by the others.  A split is looked up by (depth, first node, size) of its set -- unique within one
// dissection, and checked against what was stored.
struct NdSplit {
  int32_t depth = 0, first = 0, n = 0;
  bool as_leaf = false;              // the search found no usable separator: the set is ordered as a leaf
  std::vector<int32_t> left, right, sep;
};
struct NdSplitTable {
  std::unordered_map<uint64_t, NdSplit> map;
  std::mutex mu;                     // (the two halves of a top split are dissected by two threads)
  static uint64_t key(int depth, int first, int n) {
    return (uint64_t)depth * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)first * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)(uint32_t)n * 0x165667B19E3779F9ull;
  }
  void store(int depth, int first, int n, bool as_leaf, const std::vector<int32_t> &l, const std::vector<int32_t> &r, const std::vector<int32_t> &s) {
    std::lock_guard<std::mutex> lock(mu);
    NdSplit &sp = map[key(depth, first, n)];
    sp.depth = depth; sp.first = first; sp.n = n; sp.as_leaf = as_leaf;
    sp.left = l; sp.right = r; sp.sep = s;
  }
  const NdSplit *find(int depth, int first, int n) const {
    auto it = map.find(key(depth, first, n));
    if (it == map.end() || it->second.depth != depth || it->second.first != first || it->second.n != n) return nullptr;
    return &it->second;
  }
};

struct SymbolicOptions {
  NdSplitTable *nd_record = nullptr;        // the dissection stores every split it computes here ...
  const NdSplitTable *nd_replay = nullptr;  // ... and takes the splits it finds here instead of searching (both: see NdSplitTable)
  bool split_separators = false;     // do not chain a region's last separator into its parent separator's supernode
  bool geo_nd = true;                // nested dissection may cut along a coordinate axis (pose graphs are spatial)
  bool ml_nd = false;                // nested dissection also tries a multilevel bisection with a minimum-cover separator (small graphs)
  int threads_shift = 1;             // workgroup size classes: front size is shifted left by this before the lookup
  int max_lds_pieces = 2;            // a supernode beyond the LDS budget is cut into at most this many LDS-sized pieces, else it stays ONE front
                                     // beyond LDS (r03, 1M-edge lattice: 54-column supernodes just over the budget were cut into three 18-column
                                     // fronts, each moving a 237 x 237 update matrix through LDS -- two 160 us tasks in a level of 45 us tasks)
  int n_cus = 256;                   // compute units of the device (the schedule's estimate of a level of many tasks)
  int64_t lds_budget_elems = 19000;  // LDS scalars one workgroup may use for a front (panel + packed update)
  int nd_leaf = 40;         // nested dissection stops below this many nodes
  bool balance_blocks = true;  // fronts a few columns over a multiple of 16 hand their last nodes to their parent (symbolic.cpp, step 5)
  int balance_max_rem = 8;     //   "a few": at most this many columns over
  int merge_chain_nc = 48;     // > 0: a front of at most this many pivot columns joins its parent when it is the parent's child on the longest
  double merge_chain_frac = 0.5;   //   chain, both fit LDS together, the explicit zeros stay below this fraction of the merged front and the cost
  double merge_chain_gain_us = 0.0; //  model's finish time of the parent improves by more than this
  int amalg_np = 72;        // relaxed amalgamation of fronts beyond the small ones: merged pivot columns <= amalg_np and
  double amalg_frac = 0.15; //   explicit zeros <= amalg_frac of the merged front's entries (else only when nearly free)
  int n_parts = 1;          // >1: top ND levels are shared, subtrees owned by ranks (power of two)
  int my_part = 0;          // with n_parts > 1: the rank whose schedule is emitted (own subtrees, then shared top)
  int pin_node = -1;        // with n_parts > 1: this node (the anchor) is made part of the top separator
  double task_us = 0.0;     // subtrees cheaper than this (model us) become one leaf task; 0 = pick by the cost model
  bool lds_flow = false;    // graphs whose fronts all fit LDS (and n_parts == 1): ONE task list in ticket order for the dataflow
                            // launches k_factor_flow / k_solve_flow (lds_flow.hip.h) instead of one step per tree level
  int64_t panel_budget_elems = 0;  // must stay 0: fronts beyond LDS whose pivot panel fits this budget would form a step class of
                                   // their own (STEP_MID); its kernel (r02's one-workgroup k_factor_panel) was measured slower
                                   // than the batched tiled path and removed in r03 -- the engine rejects such steps
};

// One (block) entry of H that has to be added into a front.
struct AsmItem {
  int64_t src;    // offset into the H value array (row-major d_row x d_col block)
  int32_t lrow;   // local scalar row in the front
  int32_t lcol;   // local scalar col in the front (a pivot column)
  int16_t drow, dcol;
  int32_t diag;   // 1: symmetric diagonal block (only i >= j is used)
};

// STEP_TASKS: LDS fronts, one workgroup per task.  STEP_MID: fronts beyond LDS whose pivot panel fits LDS, one
// workgroup each (k_factor_panel), batched (task list of single fronts).  STEP_BIG: the huge fronts of
// one level (task list of single fronts), tiled over many workgroups, a sequence of batched launches.
enum StepKind : int32_t { STEP_TASKS = 0, STEP_BIG = 1, STEP_MID = 2 };
struct Step {
  int32_t kind;
  int32_t task_begin, task_end;  // STEP_TASKS: range in task_ptr
  int32_t sn;                    // unused (-1)
  int32_t max_front;             // largest M = ncols + nrows + 1 among the step's fronts
  int32_t max_lds_elems;         // LDS scalars needed by the largest front of the step
  int32_t threads;               // workgroup size chosen for the step
};

struct Symbolic {
  int N = 0, dim = 0, S = 0;
  // ---- ordering
  std::vector<int32_t> order;      // order[pos] = node
  std::vector<int32_t> pos_of;     // node -> pos
  std::vector<int32_t> node_pcol;  // node -> first permuted scalar column
  std::vector<int32_t> perm;       // permuted scalar -> reference scalar (node_offset + i)
  std::vector<int32_t> node_part;  // owner rank of the node, -1 = shared top separator
  // ---- H storage (block CSR of the permuted lower triangle)
  std::vector<int64_t> diag_off;   // per node: offset of its d x d diagonal block
  int64_t n_offblocks = 0;
  std::vector<int32_t> blk_row, blk_col;  // node ids; blk_col is eliminated first
  std::vector<int64_t> blk_off;    // offset of the d_row x d_col block (row-major)
  int64_t n_hvals = 0;
  std::vector<int32_t> edge_slot;  // per edge: off-diagonal block index
  std::vector<uint8_t> edge_transposed;  // 1: slot stores (A^T W B)^T, i.e. row node = edge.to
  std::vector<uint8_t> slot_shared;      // per block: 1 = extra block of a repeated node pair (parallel edges)
  // incidence lists for the pull-style linearisation: entry = edge*2 + role (0 from, 1 to)
  std::vector<int32_t> inc_ptr, inc_list;
  // ---- supernodes (in elimination order; children precede parents)
  std::vector<int32_t> sn_first_pos, sn_npos;  // pivot nodes = order[first .. first+npos)
  std::vector<int32_t> sn_ncols, sn_nrows, sn_col0, sn_parent, sn_owner;
  std::vector<int64_t> sn_rows_ptr;  // into sn_rows (nrows entries: permuted scalar indices, ascending)
  std::vector<int32_t> sn_rows;
  std::vector<int64_t> sn_loff;      // panel offset in L storage, (M x ncols), ld = M = ncols+nrows+1
  std::vector<int64_t> sn_uoff;      // update matrix offset; packed lower ((nrows+1) square) when sn_uld==0
  std::vector<int32_t> sn_uld;       // 0 = packed lower triangle, else leading dimension (big fronts)
  std::vector<uint8_t> sn_big;       // front does not fit LDS (mid or huge): lives in place in L storage
  std::vector<uint8_t> sn_huge;      // ... and is too large for one workgroup
  int64_t l_elems = 0, u_elems = 0;
  std::vector<int64_t> asm_ptr;
  std::vector<AsmItem> asm_items;
  // flat scalar-level assembly lists (what the kernels read): H value index ->
  // index in the front's panel (lcol*M + lrow); `dup` = blocks of parallel edges
  // that must be ADDED serially after the plain stores
  std::vector<int64_t> fasm_ptr, fdup_ptr;        // per supernode
  std::vector<int32_t> fasm_src, fasm_dst, fdup_src, fdup_dst;
  std::vector<int32_t> fasm_colptr;   // fronts beyond LDS: per permuted pivot column, where its entries start in fasm_* (dim + 1)
  // extend-add scatter maps: for a child with a packed update matrix and a
  // parent that lives in LDS, one destination per packed element: index into
  // the parent's [panel | packed update] LDS image, -1 = skip
  std::vector<int64_t> scat_ptr;                  // per supernode (as a child), -1 = no map
  std::vector<int32_t> scat;
  std::vector<int32_t> child_ptr, child_list;
  std::vector<int64_t> rel_ptr;      // per supernode (as a child): into rel, nrows+1 entries
  std::vector<int32_t> rel;          // local index in the parent's front (last entry = parent's rhs row)
  // ---- schedule
  std::vector<int32_t> task_ptr, task_sn;
  std::vector<Step> steps;           // factor order; the back-solve walks it backwards
  // ---- dataflow schedule of the LDS fronts (SymbolicOptions::lds_flow; all fronts in LDS, one rank): `steps` is ONE
  // STEP_TASKS step whose tasks are in TICKET order -- the start order of a list schedule of the task tree, so every task
  // comes after the tasks of its fronts' children -- and the back substitution draws its tasks in solve_order (parents
  // before children).  Leaf subtrees cheaper than the threshold are one task, every front above them a task of its own.
  bool lds_flow = false;
  std::vector<int32_t> solve_order;  // ticket -> task for k_solve_flow
  double est_factor_us = 0.0, est_solve_us = 0.0;   // the cost model's makespans of the two launches
  // ---- sharding over ranks (n_parts > 1): steps [0, n_local_steps) are this rank's own subtrees,
  // the rest are the shared top fronts every rank factors after the exchange.  Boundary fronts
  // (owned, parent shared) publish their update matrix, packed, in the exchange buffer.
  int32_t n_local_steps = 0;
  std::vector<int64_t> sn_xch_off;   // per supernode: offset in the exchange buffer, -1 = not a boundary front
  int64_t xch_elems = 0;             // n_parts * xch_chunk
  int64_t xch_chunk = 0;             // scalars of one rank's chunk of the exchange buffer (all-gather)
  int64_t xch_shared_off = 0;        // inside a chunk: the rank's partial diagonal blocks + rhs of the shared nodes
  int64_t xch_flag_off = 0;          // inside a chunk: the rank's device error flag after stage 0 (one scalar)
  std::vector<int8_t> col_owner;     // per permuted scalar column: owner rank, -1 = shared
  // ---- stats
  int64_t nnz_l_blocks = 0;          // node-level nonzero blocks of L (no padding)
  int64_t nnz_l_entries = 0;         // their scalars: sum over the blocks of d_row * d_col (nnzblk(L) * d^2)
  int64_t factor_flops = 0;
  int32_t max_front = 0, max_pivot_cols = 0, n_big = 0;
  double est_critical_us = 0.0;
  double task_us_used = 0.0;
};

// Returns "" or an error message.
std::string analyze(const HostGraph &g, const SymbolicOptions &opt, Symbolic &sym);

// The ordering phase alone (nested dissection down to opt.nd_leaf), for its side effect on opt.nd_record.
std::string dissect_only(const HostGraph &g, const SymbolicOptions &opt);

}  // namespace rrpgo
