/* abi_client.c -- a plain C99 client of include/rr_pgo.h (test infrastructure).
 *
 * The callers of the path this library replaces are compiled code (Rust: src/mapping/mod.rs:6,
 * benches/graph_slam.rs:9-10, examples/mapping/pose_graph_optimization.rs:49-50 of the reference); Rust is not in
 * this image, so the boundary is exercised from C instead: the header must compile as C, and a C program drives
 * load -> optimize -> get_state.
 *
 *   abi_client layout
 *       prints `struct field offset size` for every field of the three public structs as THIS compiler lays them
 *       out (tests/test_abi_and_host.py compares them with the ctypes mirror in rustrobotics_amd/_lib.py)
 *   abi_client run <file.g2o> <iterations> <out.bin> [gn|lm]
 *       PoseGraph::new(file)?.optimize(iterations, false, false) (benches/graph_slam.rs:9-10) through the C ABI;
 *       out.bin = int32 n_errors, int32 state_len, then the errors and the state as raw doubles (compared bit for
 *       bit with the Python mirror's result by tests/test_gpu_parity.py)
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rr_pgo.h"

/* the layout the Rust #[repr(C)] structs of INTEGRATION.md and the ctypes mirror assume (LP64) */
_Static_assert(sizeof(rr_pgo_options) == 16 * 4, "rr_pgo_options is sixteen int32");
_Static_assert(offsetof(rr_pgo_options, reserved) == 6 * 4, "rr_pgo_options.reserved follows six int32");
_Static_assert(sizeof(rr_pgo_graph_desc) == 80, "rr_pgo_graph_desc: two int32 (padded) + eight pointers");
_Static_assert(offsetof(rr_pgo_graph_desc, node_kind) == 8 && offsetof(rr_pgo_graph_desc, n_edges) == 32 &&
                   offsetof(rr_pgo_graph_desc, edge_info) == 72,
               "rr_pgo_graph_desc field offsets");
_Static_assert(sizeof(rr_pgo_stats) == 3 * 8 + 6 * 4 + 10 * 8 + 2 * 4, "rr_pgo_stats");
_Static_assert(offsetof(rr_pgo_stats, analyze_ms) == 48 && offsetof(rr_pgo_stats, big_update_flops) == 104 &&
                   offsetof(rr_pgo_stats, big_flow_flops) == 112 && offsetof(rr_pgo_stats, stored_factor_bytes) == 120 &&
                   offsetof(rr_pgo_stats, abi_version) == 128 && offsetof(rr_pgo_stats, lds_dataflow) == 132,
               "rr_pgo_stats field offsets");

#define FIELD(S, f) printf(#S " " #f " %zu %zu\n", offsetof(S, f), sizeof(((S *)0)->f))

static int print_layout(void) {
  printf("rr_pgo_options . 0 %zu\n", sizeof(rr_pgo_options));
  FIELD(rr_pgo_options, precision); FIELD(rr_pgo_options, device); FIELD(rr_pgo_options, solver);
  FIELD(rr_pgo_options, rank); FIELD(rr_pgo_options, world_size); FIELD(rr_pgo_options, sharded);
  FIELD(rr_pgo_options, reserved);
  printf("rr_pgo_graph_desc . 0 %zu\n", sizeof(rr_pgo_graph_desc));
  FIELD(rr_pgo_graph_desc, n_nodes); FIELD(rr_pgo_graph_desc, node_kind); FIELD(rr_pgo_graph_desc, node_id);
  FIELD(rr_pgo_graph_desc, node_state); FIELD(rr_pgo_graph_desc, n_edges); FIELD(rr_pgo_graph_desc, edge_kind);
  FIELD(rr_pgo_graph_desc, edge_from); FIELD(rr_pgo_graph_desc, edge_to); FIELD(rr_pgo_graph_desc, edge_meas);
  FIELD(rr_pgo_graph_desc, edge_info);
  printf("rr_pgo_stats . 0 %zu\n", sizeof(rr_pgo_stats));
  FIELD(rr_pgo_stats, nnz_h_blocks); FIELD(rr_pgo_stats, nnz_l_scalars); FIELD(rr_pgo_stats, factor_flops);
  FIELD(rr_pgo_stats, n_supernodes); FIELD(rr_pgo_stats, n_levels); FIELD(rr_pgo_stats, n_launches_per_iter);
  FIELD(rr_pgo_stats, max_front); FIELD(rr_pgo_stats, max_pivot_cols); FIELD(rr_pgo_stats, n_big_fronts);
  FIELD(rr_pgo_stats, analyze_ms); FIELD(rr_pgo_stats, parse_ms); FIELD(rr_pgo_stats, bytes_linearize);
  FIELD(rr_pgo_stats, bytes_factor); FIELD(rr_pgo_stats, bytes_solve); FIELD(rr_pgo_stats, bytes_update);
  FIELD(rr_pgo_stats, bytes_chi2); FIELD(rr_pgo_stats, big_update_flops); FIELD(rr_pgo_stats, big_flow_flops);
  FIELD(rr_pgo_stats, stored_factor_bytes); FIELD(rr_pgo_stats, abi_version); FIELD(rr_pgo_stats, lds_dataflow);
  printf("enum RR_PGO_NUM_KCLASS %d 0\n", (int)RR_PGO_NUM_KCLASS);
  printf("enum RR_PGO_ABI_VERSION %d 0\n", (int)RR_PGO_ABI_VERSION);
  return 0;
}

static int fail(const char *what, int rc) {
  fprintf(stderr, "abi_client: %s failed with %d: %s\n", what, rc, rr_pgo_last_error());
  return 2;
}

static int run(const char *path, int iterations, const char *out_path, const char *solver) {
  rr_pgo_options opt;
  rr_pgo *h = NULL;
  int rc, n_errors = 0, state_len;
  double *errors, *state;
  FILE *f;
  rr_pgo_default_options(&opt);
  opt.solver = (solver && strcmp(solver, "lm") == 0) ? RR_PGO_LEVENBERG_MARQUARDT : RR_PGO_GAUSS_NEWTON;
  rc = rr_pgo_load_g2o(path, &opt, &h); /* PoseGraph::new, pose_graph_optimization.rs:215-227 */
  if (rc != RR_PGO_OK) return fail("rr_pgo_load_g2o", rc);
  errors = (double *)malloc(sizeof(double) * (size_t)(iterations + 1));
  state_len = rr_pgo_state_len(h);
  state = (double *)malloc(sizeof(double) * (size_t)state_len);
  if (!errors || !state) return 3;
  rc = rr_pgo_optimize(h, iterations, errors, &n_errors, NULL); /* optimize, :247-303 */
  if (rc != RR_PGO_OK) return fail("rr_pgo_optimize", rc);
  rc = rr_pgo_get_state(h, state);
  if (rc != RR_PGO_OK) return fail("rr_pgo_get_state", rc);
  printf("%d nodes, %d edges, dim %d: chi2 %.9g -> %.9g in %d iterations\n", rr_pgo_num_nodes(h), rr_pgo_num_edges(h),
         rr_pgo_dim(h), errors[0], errors[n_errors - 1], n_errors - 1);
  f = fopen(out_path, "wb");
  if (!f) return 4;
  fwrite(&n_errors, sizeof(int), 1, f);
  fwrite(&state_len, sizeof(int), 1, f);
  fwrite(errors, sizeof(double), (size_t)n_errors, f);
  fwrite(state, sizeof(double), (size_t)state_len, f);
  fclose(f);
  rr_pgo_destroy(h); /* Drop */
  free(errors);
  free(state);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && strcmp(argv[1], "layout") == 0) return print_layout();
  if (argc >= 5 && strcmp(argv[1], "run") == 0) return run(argv[2], atoi(argv[3]), argv[4], argc > 5 ? argv[5] : "gn");
  fprintf(stderr, "usage: abi_client layout | abi_client run <file.g2o> <iterations> <out.bin> [gn|lm]\n");
  return 1;
}
