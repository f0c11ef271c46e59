/*
 * rr_pgo.h -- C ABI of librr_pgo.so, the MI355X (gfx950) pose-graph-optimization
 * backend for RustRobotics' `robotics::mapping::PoseGraph`.
 *
 * The reference crate has no FFI seam of its own for this path;
 * the one it has one level down is russell_sparse -> UMFPACK, an opaque handle
 * with new/factorize/solve/drop returning i32 status codes
 * (reference src/mapping/pose_graph_optimization.rs:130,138,141).  This header
 * follows that model: opaque handle, plain pointers + sizes, int status.
 * Each entry point names the reference interface it replaces (file:line is
 * relative to the reference repository root).  INTEGRATION.md shows the Rust
 * `extern "C"` block + `PoseGraph` wrapper a maintainer would add.
 *
 * Threading: one handle = one caller thread at a time; handles are independent.
 * All functions return RR_PGO_OK (0) or a negative RR_PGO_E* code;
 * rr_pgo_last_error() returns the message of the calling thread's last failure.
 * Nothing here falls back to the CPU: without a HIP device every compute entry
 * point fails with RR_PGO_ENODEVICE.
 */
#ifndef RR_PGO_H
#define RR_PGO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_PGO_ABI_VERSION 4  /* see rr_pgo_abi_version() */

typedef struct rr_pgo rr_pgo; /* opaque: replaces `struct PoseGraph`, pose_graph_optimization.rs:155-163 */

enum {
  RR_PGO_OK = 0,
  RR_PGO_EINVAL = -1,   /* bad argument / malformed description.  Deliberate deviations from the reference, which
                         * accepts these inputs and then fails (or silently misbehaves) in the solver: a self-loop
                         * edge (from == to; the reference sums H_ii, H_ij, H_ji, H_jj into one block) and, for the
                         * loader, a repeated VERTEX id (the reference overwrites the node but keeps the first
                         * offset, leaving structurally empty rows => singular system) are rejected up front:
                         * RR_PGO_EINVAL from rr_pgo_create, RR_PGO_EPARSE from rr_pgo_load_g2o. */
  RR_PGO_EIO = -2,      /* file could not be read                         (Err(io) at g2o.rs:51) */
  RR_PGO_EPARSE = -3,   /* malformed g2o text                             (Err/panic at g2o.rs:53-139) */
  RR_PGO_ENODEVICE = -4,/* no usable HIP device / HIP runtime error */
  RR_PGO_ENOTSPD = -5,  /* factorisation hit a non-positive pivot         (Err from umfpack.factorize, :138).
                         * Like the reference (Err at :271 comes before update_nodes) the handle's state is the one
                         * before the failed iteration: the step is not applied, the caller may retry, e.g. with LM. */
  RR_PGO_ENOMEM = -6,
  RR_PGO_EUNSUPPORTED = -7,
  RR_PGO_ETIMEOUT = -8  /* a wait between workgroups of one launch (the dataflow launches k_factor_flow / k_solve_flow /
                         * k_big_flow / k_big_solve_flow hand fronts, tiles and solutions to each other) ran out of time:
                         * the launch drained without applying the step -- the handle's state is the one before the call, the
                         * handle stays usable.  No counterpart in the reference (its solver is one thread). */
};

/* enum PoseGraphSolver, pose_graph_optimization.rs:28-32 */
enum { RR_PGO_GAUSS_NEWTON = 0, RR_PGO_LEVENBERG_MARQUARDT = 1 };
/* enum Node variants, :147-154 ; enum Edge variants, :20-26 */
enum { RR_PGO_NODE_SE2 = 0, RR_PGO_NODE_XY = 1, RR_PGO_NODE_SE3 = 2 };
enum { RR_PGO_EDGE_SE2 = 0, RR_PGO_EDGE_SE2_XY = 1, RR_PGO_EDGE_SE3 = 2 };
/* arithmetic type of the device path (the reference is f64 throughout).
 * RR_PGO_MIXED: state, measurements, error / Jacobians / gradient and chi2 in f64; H, its factor and
 * the solve in f32.  The gradient is exact, so Gauss-Newton converges to the f64 minimum while the
 * factorisation (all of the cost on large graphs) runs at the f32 rate. */
enum { RR_PGO_F64 = 0, RR_PGO_F32 = 1, RR_PGO_MIXED = 2 };

/* What parse_g2o returns (g2o.rs:35-45: len, edges, lut, nodes), flattened.
 * All arrays are borrowed for the duration of the call only. */
typedef struct rr_pgo_graph_desc {
  int32_t n_nodes;
  const int32_t *node_kind;   /* [n_nodes] RR_PGO_NODE_*; scalar offsets follow this order (g2o.rs:60-77) */
  const uint32_t *node_id;    /* [n_nodes] g2o ids, may be NULL (then id = index) */
  const double *node_state;   /* packed in node order: SE2 x,y,theta | XY x,y | SE3 x,y,z,qx,qy,qz,qw */
  int32_t n_edges;
  const int32_t *edge_kind;   /* [n_edges] RR_PGO_EDGE_*, file order (order defines the prior, :330-336) */
  const int32_t *edge_from;   /* [n_edges] dense node index (lut/nodes lookups of :312-320 done by the caller) */
  const int32_t *edge_to;
  const double *edge_meas;    /* packed in edge order: SE2 x,y,theta | SE2_XY x,y | SE3 x,y,z,qx,qy,qz,qw */
  const double *edge_info;    /* packed upper triangles, row-major: 6 | 3 | 21 values (g2o.rs:82,100,117) */
} rr_pgo_graph_desc;

typedef struct rr_pgo_options {
  int32_t precision;      /* RR_PGO_F64 (default), RR_PGO_F32 or RR_PGO_MIXED */
  int32_t device;         /* HIP device ordinal, -1 = current device */
  int32_t solver;         /* RR_PGO_GAUSS_NEWTON / RR_PGO_LEVENBERG_MARQUARDT (PoseGraph::new's 2nd arg, :215) */
  /* Multi-GPU sharding of ONE graph.  world_size <= 1: single GPU; else a power of two <= 64. */
  int32_t rank, world_size;
  int32_t sharded;        /* 1: build the handle for rr_pgo_stage even with world_size <= 1 (a one-rank group: same
                           * stages, same collectives); world_size > 1 implies it */
  int32_t reserved[10];   /* zero */
} rr_pgo_options;

void rr_pgo_default_options(rr_pgo_options *opt);

/* ---- lifecycle ---------------------------------------------------------- */

/* PoseGraph::new(file_path, solver)  (:215-227) = parse_g2o (g2o.rs:35-143) +
 * upload + one-off symbolic analysis.  Same tag set and failure cases as the
 * reference loader; a panic there is RR_PGO_EPARSE here. */
int rr_pgo_load_g2o(const char *path, const rr_pgo_options *opt, rr_pgo **out);

/* PoseGraph::new for an already parsed graph (what a Rust caller holding the
 * output of its own parse_g2o passes down). */
int rr_pgo_create(const rr_pgo_graph_desc *desc, const rr_pgo_options *opt, rr_pgo **out);

/* Drop for PoseGraph */
void rr_pgo_destroy(rr_pgo *h);

/* message for the last failing call on this thread (Box<dyn Error> text) */
const char *rr_pgo_last_error(void);

/* Process-wide state the library keeps BETWEEN handles, and how to give it back (no counterpart in the reference, which
 * keeps nothing between two PoseGraphs):
 *   - device memory of destroyed handles (chunks of at most 64 MB, at most 256 MB in all), their HIP streams (at most 16)
 *     and the symbolic analysis of the last four graph structures are kept for the next handle: the reference's bench
 *     (benches/graph_slam.rs:9-10) is a loop of new + optimize(10) + drop, and hipMalloc / hipStreamCreate / the analysis
 *     cost more than its ten iterations.  RR_PGO_ANALYSIS_CACHE=0 switches the last one off.
 *   - rr_pgo_load_g2o / rr_pgo_create work on a few host threads of their own for the duration of the call (the parts of a
 *     large file, the halves of the nested dissection, the candidate elimination trees): never more than the host has
 *     cores, all joined before the call returns; a thread the host refuses is not used.
 * rr_pgo_trim() frees everything of the first kind that no live handle is using (safe at any time from any thread; the
 * next constructor pays for allocation and analysis again).  Returns RR_PGO_OK. */
int rr_pgo_trim(void);

/* ---- sizes / fields ------------------------------------------------------ */
int32_t rr_pgo_num_nodes(const rr_pgo *h);  /* nodes.len()  */
int32_t rr_pgo_num_edges(const rr_pgo *h);  /* edges.len()  */
int32_t rr_pgo_dim(const rr_pgo *h);        /* len, :156    */
int32_t rr_pgo_state_len(const rr_pgo *h);  /* entries rr_pgo_get_state writes */
int32_t rr_pgo_anchor_node(const rr_pgo *h);/* from-node of the first pose-pose edge (prior target, :330-336), -1 if none */

/* host copy of the parsed graph in rr_pgo_graph_desc packing (so a caller or a
 * test can hand the identical graph to another implementation).  Pointers stay
 * valid until rr_pgo_destroy. */
int rr_pgo_get_graph(const rr_pgo *h, rr_pgo_graph_desc *out);

/* ---- the hot path -------------------------------------------------------- */

/* global_error(graph)  (:537-574): sum_e e^T Omega e at the current state.
 * f64 result also in f32 mode. */
int rr_pgo_chi2(rr_pgo *h, double *out);

/* build_linear_system(lambda)?.solve()?  (:271 ; linearize_and_solve :371-373
 * is lambda = 0, lm = 0).  Includes the 1e7 prior (:330-336), b = -b (:361) and,
 * when lm != 0, + lambda*I (:362-366).  dx_out: rr_pgo_dim entries, reference
 * scalar order (node offsets). */
int rr_pgo_linearize_solve(rr_pgo *h, double lambda, int lm, double *dx_out);

/* update_nodes(sign * dx)  (:229-245) */
int rr_pgo_update(rr_pgo *h, const double *dx, double sign);

/* optimize(num_iterations, log=false, plot=false)  (:247-303), exact control
 * flow incl. the LM accept/reject quirks (:275-286) and the |dx| < 1e-4 break
 * (:298-300).  errors_out needs num_iterations+1 slots; *n_errors = 1 +
 * iterations executed (the length of the reference's returned Vec<f64>).
 * norms_out (may be NULL): |dx| per executed iteration.
 * The loop runs on the device: the stop rule, Levenberg-Marquardt's accept / reject and lambda are decided by the kernel
 * that finishes an iteration, which publishes (chi2, |dx|) in host-coherent memory; the host enqueues one iteration ahead
 * and polls -- no stream synchronisation inside the call (handles of 48+ launches per iteration, i.e. graphs of tens of
 * thousands of poses, keep one host round trip per iteration: 0.5 % of theirs).  Same bits either way.
 * With log or plot the reference prints / plots BETWEEN iterations (:258-268, :288-296): a shim steps through
 * rr_pgo_linearize_solve / rr_pgo_update / rr_pgo_chi2 instead. */
int rr_pgo_optimize(rr_pgo *h, int32_t num_iterations, double *errors_out,
                    int32_t *n_errors, double *norms_out);

/* State read-back: SE2 -> x, y, atan2(im,re) ; XY -> x, y ; SE3 -> x,y,z,qx,qy,qz,qw,
 * node order.  (Field access on PoseGraph.nodes in the reference.) */
int rr_pgo_get_state(rr_pgo *h, double *out);
/* Overwrite the state (same packing); used to restart a benchmark run.  Returns once the copy is enqueued on the handle's
 * stream (everything else the handle does is ordered behind it); setting the state of the previous call again costs a
 * comparison and one device-side copy. */
int rr_pgo_set_state(rr_pgo *h, const double *state);

/* ---- inspection of the assembled system (parity tests) ------------------- */

/* Runs the linearisation kernels only and returns the assembled normal matrix
 * as a dense-block list: for every stored block s, (row_node, col_node) and
 * d_row x d_col row-major values; plus b (negated).  Call with all output
 * pointers NULL to get the counts.  Values are converted to f64. */
int rr_pgo_assemble(rr_pgo *h, double lambda, int lm, int32_t *n_blocks,
                    int32_t *block_row_node, int32_t *block_col_node,
                    int64_t *block_val_offset, double *block_vals,
                    int64_t *n_vals, double *b_out);

/* ---- measurement --------------------------------------------------------- */

/* Enqueue `iters` Gauss-Newton iterations (linearise, factor, solve, update,
 * chi2) back to back on the handle's stream WITHOUT the convergence break and
 * without host round trips; returns immediately.  rr_pgo_sync waits. */
int rr_pgo_iterate_async(rr_pgo *h, int32_t iters);
int rr_pgo_sync(rr_pgo *h);

typedef struct rr_pgo_stats {
  /* symbolic analysis (done once in create) */
  int64_t nnz_h_blocks;      /* stored blocks of H (diag + lower off-diag)      */
  int64_t nnz_l_scalars;     /* scalars in the supernodal factor incl. padding  */
  int64_t factor_flops;      /* flops of one numeric factorisation               */
  int32_t n_supernodes, n_levels, n_launches_per_iter;
  int32_t max_front, max_pivot_cols;
  int32_t n_big_fronts;      /* fronts taken by the tiled multi-workgroup path   */
  double analyze_ms, parse_ms;
  /* algorithmic bytes of one GN iteration by phase */
  double bytes_linearize, bytes_factor, bytes_solve, bytes_update, bytes_chi2;
  double big_update_flops;   /* flops (2 per multiply-add) of one iteration's k_big_update launches */
  double big_flow_flops;     /* the same count for the trailing-update tiles that run inside k_big_flow launches */
  double stored_factor_bytes;/* bytes of the factor AS STORED (supernodal panels with their padding; fronts beyond LDS: the whole
                              * in-place M x M front) -- NOT what bytes_factor / bytes_solve are computed from: those follow
                              * SURVEY 8(d), nnzblk(L) * d^2 * s, every datum moved once */
  int32_t abi_version;       /* RR_PGO_ABI_VERSION of the library that filled the struct */
  int32_t lds_dataflow;      /* 1: the fronts that live in LDS are factored and solved by the two dataflow launches
                              * (k_factor_flow / k_solve_flow), 0: by one launch per level of the task tree */
} rr_pgo_stats;
int rr_pgo_get_stats(const rr_pgo *h, rr_pgo_stats *out);

/* Host-only: parse + symbolic analysis of a g2o file, the rr_pgo_stats a handle on it would report (fields that depend
 * on the device -- n_launches_per_iter, big_update_flops, big_flow_flops -- are 0).  Needs no HIP device. */
int rr_pgo_analyze_g2o(const char *path, const rr_pgo_options *opt, rr_pgo_stats *out);

/* ABI version: bumped whenever a struct layout, an enum that sizes a caller's array (RR_PGO_NUM_KCLASS) or the meaning
 * of an argument changes.  r03 -> 3 (RR_PGO_NUM_KCLASS 10 -> 11), r04 -> 4 (rr_pgo_stats: stored_factor_bytes,
 * abi_version; RR_PGO_ETIMEOUT; rr_pgo_profile takes the length of the caller's arrays). */
int32_t rr_pgo_abi_version(void);

/* Per-kernel timing measured with HIP events on the handle's own stream.
 * Runs `iters` eager (non-graph) GN iterations with an event pair around every
 * launch and accumulates per kernel class.  The caller passes the length of its arrays (RR_PGO_NUM_KCLASS). */
enum {
  RR_PGO_K_LINEARIZE = 0,   /* k_linearize                                             */
  RR_PGO_K_FACTOR = 1,      /* k_factor_tasks (fronts in LDS)                          */
  RR_PGO_K_SOLVE = 2,       /* k_solve_tasks                                           */
  RR_PGO_K_UPDATE = 3,      /* k_update                                                */
  RR_PGO_K_REDUCE = 4,      /* k_finalize_slot                                         */
  RR_PGO_K_BIGFRONT = 5,    /* huge fronts: k_big_build + k_big_assemble (gather pass), k_flow_reset */
  RR_PGO_K_BIG_PANEL = 6,   /* k_big_panel32 (huge fronts: 32-column chain steps)       */
  RR_PGO_K_BIG_UPDATE = 7,  /* k_big_update + k_big_schur (huge fronts: MFMA trailing updates) */
  RR_PGO_K_MID_FACTOR = 8,  /* retired in r03 (the one-workgroup panel class was removed): always 0   */
  RR_PGO_K_BIG_SOLVE = 9,   /* k_big_gemv_partial + k_big_solve_flow (k_big_solve_sp) / k_solve_mid */
  RR_PGO_K_BIG_FLOW = 10,   /* k_big_flow (huge fronts of a level of few fronts: panels + updates as one dataflow launch) */
  RR_PGO_NUM_KCLASS = 11
};
int rr_pgo_profile(rr_pgo *h, int32_t iters, double *ms_total /*[n_classes]*/,
                   int64_t *launches /*[n_classes]*/, int32_t n_classes /* length of the two arrays: at most that many
                   classes are written (a caller built against an older RR_PGO_NUM_KCLASS is not overrun) */);

/* ---- testing ------------------------------------------------------------- */

/* Failure injection for the dataflow launches (tests only; no counterpart in the reference): make ONE hand-off between
 * workgroups never arrive, so that the waits behind it run into their time bound (environment RR_PGO_FLOW_TIMEOUT_MS,
 * read when the handle is created; default 2000) and the next iteration returns RR_PGO_ETIMEOUT with the state untouched.
 * mode 1: a child front of the factorisation of the LDS fronts (k_factor_flow); 2: a parent front of their back
 * substitution (k_solve_flow); 3: one panel step of the fronts beyond LDS (k_big_flow); 0: put everything back. */
int rr_pgo_debug_withhold(rr_pgo *h, int32_t mode);

/* ---- synthetic workload (BASELINE config 4) -------------------- */

/* Deterministic SE(2) lattice graph: W x H poses in boustrophedon order, the
 * 10-offset stencil (+ part of an 11th) described in SURVEY.md 8(d), trimmed or
 * capped to n_edges_target (<=0: all stencil edges).  Fills a graph description
 * whose arrays are owned by the returned object; free with rr_pgo_synth_free. */
typedef struct rr_pgo_synth rr_pgo_synth;
int rr_pgo_synth_grid(int32_t width, int32_t height, int64_t n_edges_target,
                      uint64_t seed_meas, uint64_t seed_init, rr_pgo_synth **out,
                      rr_pgo_graph_desc *desc);
void rr_pgo_synth_free(rr_pgo_synth *s);

/* ---- sharding ONE graph over ranks ------------------------------ */

/* A handle created with opt.world_size = P > 1 (power of two, <= 64) and opt.rank = r owns the subtrees
 * of partition r of the nested dissection (its nodes, their edges, their fronts); the top log2(P)
 * separator levels -- and the anchor node -- are shared: every rank holds their state and factors their
 * fronts redundantly, so there is no panel traffic on the sequential chain of the top fronts.  Every rank
 * creates its handle from the SAME graph.  One Gauss-Newton iteration is two stages and two collectives
 * (the reference has no counterpart: its only parallel construct is pose_graph_optimization.rs:230):
 *
 *   rr_pgo_stage(h, 0, lambda, lm)   linearise the rank's nodes (:305-369 restricted to own + shared nodes),
 *                                    factor its own subtrees, publish the boundary fronts' update matrices in
 *                                    chunk r of exchange buffer 0
 *   ALL-GATHER of buffer 0           in place: rank r contributes elements [r * n / P, (r + 1) * n / P)
 *   rr_pgo_stage(h, 1, ...)          shared top fronts, back substitution (top + own subtrees), update_nodes
 *                                    (:229-245) for own + shared nodes, partial chi2 / |dx|^2 -> buffer 1
 *   ALL-REDUCE (sum) of buffer 1     two doubles; only the stop rule (:298-300) and the log need them
 *   rr_pgo_stage(h, 2, ...)          chi2 of the current state only (partial -> buffer 1, then the same
 *                                    all-reduce): the last entry of optimize()'s error list
 *
 * Every edge's chi2 term and every node's |dx|^2 is counted by exactly one rank.  A rank's copy of the state
 * is valid for its own and the shared nodes (rr_pgo_node_owner says which); rr_pgo_get_state returns the
 * rank's copy.  The library enqueues on its own stream (rr_pgo_stream): issue the collectives on that stream
 * and an iteration needs no host synchronisation.  Buffers are device memory of element size *elem_size (4 or
 * 8 for buffer 0, 8 for buffer 1); the caller may bind memory it allocated itself (so that its collective
 * library can register it) with rr_pgo_set_exchange_buffer before the first stage. */
int rr_pgo_exchange_buffer(rr_pgo *h, int32_t which, void **dev_ptr, int64_t *n_elems, int32_t *elem_size);
int rr_pgo_set_exchange_buffer(rr_pgo *h, int32_t which, void *dev_ptr, int64_t n_elems);
int rr_pgo_stage(rr_pgo *h, int32_t stage, double lambda, int lm);
/* after the all-reduce of buffer 1: chi2 of the state BEFORE the iteration's update and |dx| of the step
 * (stage 1), or chi2 of the current state (stage 2).  Synchronises the handle's stream. */
int rr_pgo_stage_scalars(rr_pgo *h, double *chi2, double *norm_dx);
void *rr_pgo_stream(rr_pgo *h);                           /* hipStream_t the handle launches on */
/* owner[n_nodes]: rank that owns the node, -1 = shared (top separators, anchor); all 0 on an unsharded handle */
int rr_pgo_node_owner(const rr_pgo *h, int32_t *owner);

#ifdef __cplusplus
}
#endif
#endif
