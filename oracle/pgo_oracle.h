/*
 * pgo_oracle.h -- CPU oracle for the pose-graph-optimization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, scalar, fp64 restatement of the
 * reference algorithm (RustRobotics src/mapping/g2o.rs and
 * src/mapping/pose_graph_optimization.rs).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library
 * (rustrobotics_amd/csrc, librr_pgo.so) never links, loads or calls anything in
 * this directory.
 *
 * Parity status: PINNED for SE(2) pose-pose and pose-landmark graphs by the
 * reference's own unit-test goldens (tests/test_oracle_goldens.py lists them with
 * file:line).  UNPINNED for Levenberg-Marquardt, for input_M3500_g2o.g2o and for
 * everything SE(3): the reference has no test (LM, M3500) or no executed code
 * (SE(3) hits todo!() at pose_graph_optimization.rs:241,357,570).
 */
#ifndef PGO_ORACLE_H
#define PGO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct og_graph og_graph;

enum { OG_NODE_SE2 = 0, OG_NODE_XY = 1, OG_NODE_SE3 = 2 };
enum { OG_EDGE_SE2 = 0, OG_EDGE_SE2_XY = 1, OG_EDGE_SE3 = 2 };
enum { OG_GAUSS_NEWTON = 0, OG_LEVENBERG_MARQUARDT = 1 };

/* g2o.rs:35-143.  Returns NULL and fills err on any failure (the reference
 * returns Err or panics; both map to NULL here). */
og_graph *og_load_g2o(const char *path, char *err, int errlen);

/* Build a graph from flat arrays (synthetic graphs; same in-memory semantics as
 * a parsed file).  node_kind[n]; node_state: SE2 x,y,theta | XY x,y | SE3
 * x,y,z,qx,qy,qz,qw packed back to back in node order.  edge_kind[m];
 * edge_from/edge_to are dense node indices; edge_meas and edge_info packed back
 * to back: SE2 (3 meas, 6 upper-tri info), SE2_XY (2, 3), SE3 (7, 21). */
og_graph *og_create(int n_nodes, const int *node_kind, const double *node_state,
                    int n_edges, const int *edge_kind, const int *edge_from,
                    const int *edge_to, const double *edge_meas,
                    const double *edge_info, char *err, int errlen);
void og_free(og_graph *g);

int og_num_nodes(const og_graph *g);
int og_num_edges(const og_graph *g);
int og_dim(const og_graph *g); /* `len` of pose_graph_optimization.rs:156 */
int og_node_kind(const og_graph *g, int node);
int og_node_offset(const og_graph *g, int node);
unsigned og_node_id(const og_graph *g, int node);
int og_edge_kind(const og_graph *g, int edge);
int og_edge_from(const og_graph *g, int edge); /* dense node index */
int og_edge_to(const og_graph *g, int edge);
/* raw flattened copies (for feeding the same graph to the product library) */
void og_get_edge_meas(const og_graph *g, int edge, double *out);
void og_get_edge_info_full(const og_graph *g, int edge, double *out); /* d*d row-major */

/* global_error, pose_graph_optimization.rs:537-574 */
double og_global_error(const og_graph *g);

/* error vector + Jacobians of one edge (row-major A: de x d1, B: de x d2).
 * SE2-SE2: :434-447,457-486 ; SE2-XY: :449-455,516-535 */
int og_linearize_edge(const og_graph *g, int edge, double *A, double *B, double *e);

/* build_linear_system(lambda), :305-369, with COO duplicates summed.
 * Returns the LOWER triangle in CSC (colptr n+1, rowidx, vals) and b (already
 * negated, :361).  Call with vals==NULL to query nnz.  Returns nnz or <0. */
int og_build_system(const og_graph *g, double lambda, int lm, int *colptr,
                    int *rowidx, double *vals, double *b);

/* build_linear_system(lambda)?.solve()?  (:271 / :371-373).  dx has og_dim entries. */
int og_linearize_and_solve(const og_graph *g, double lambda, int lm, double *dx);

/* update_nodes(sign*dx), :229-245 */
void og_update_nodes(og_graph *g, const double *dx, double sign);

/* optimize(num_iterations), :247-303.  errors needs num_iterations+1 slots,
 * norms num_iterations.  Returns number of errors written (1 + iterations run)
 * or <0 on solver failure. */
int og_optimize(og_graph *g, int num_iterations, int solver, double *errors,
                double *norms);

/* State out: SE2 -> x,y,atan2(im,re) ; XY -> x,y ; SE3 -> x,y,z,qx,qy,qz,qw */
void og_get_state(const og_graph *g, double *out);
int og_state_len(const og_graph *g);
/* SE2 rotation as stored (re, im) -- not renormalised by update_nodes (:236) */
void og_get_se2_raw(const og_graph *g, int node, double *out4);

/* stats of the last factorisation (for reporting): nnz(L) scalar */
long og_last_nnz_l(const og_graph *g);

#ifdef __cplusplus
}
#endif
#endif
