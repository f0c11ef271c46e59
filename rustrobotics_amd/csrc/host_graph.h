// host_graph.h -- host-side pose graph, flattened (what the reference keeps as
// Vec<Edge<f64>> + two FxHashMaps, pose_graph_optimization.rs:155-163).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rrpgo {

enum NodeKind : int32_t { NODE_SE2 = 0, NODE_XY = 1, NODE_SE3 = 2 };
enum EdgeKind : int32_t { EDGE_SE2 = 0, EDGE_SE2_XY = 1, EDGE_SE3 = 2 };

inline int node_dim(int kind) { return kind == NODE_SE2 ? 3 : kind == NODE_XY ? 2 : 6; }
inline int node_state_len(int kind) { return kind == NODE_SE2 ? 3 : kind == NODE_XY ? 2 : 7; }
inline int edge_dim(int kind) { return kind == EDGE_SE2 ? 3 : kind == EDGE_SE2_XY ? 2 : 6; }
inline int edge_meas_len(int kind) { return kind == EDGE_SE2 ? 3 : kind == EDGE_SE2_XY ? 2 : 7; }
inline int edge_info_len(int kind) { return kind == EDGE_SE2 ? 6 : kind == EDGE_SE2_XY ? 3 : 21; }

// Same packing as rr_pgo_graph_desc (include/rr_pgo.h).
struct HostGraph {
  std::vector<int32_t> node_kind;
  std::vector<uint32_t> node_id;
  std::vector<int32_t> node_offset;     // scalar offset, vertex file order (g2o.rs:60-77)
  std::vector<int64_t> node_state_off;  // offset into node_state
  std::vector<double> node_state;       // SE2 x,y,theta | XY x,y | SE3 x,y,z,qx,qy,qz,qw
  std::vector<int32_t> edge_kind, edge_from, edge_to;
  std::vector<int64_t> edge_meas_off, edge_info_off;
  std::vector<double> edge_meas, edge_info;
  int32_t dim = 0;          // `len`
  int32_t anchor_node = -1; // from-node of the first pose-pose edge (prior, :330-336)
  bool has_se3 = false, has_2d = false;

  int n_nodes() const { return (int)node_kind.size(); }
  int n_edges() const { return (int)edge_kind.size(); }
  // Recomputes offsets / anchor / flags from the kind + packed arrays; returns
  // an error string (empty = ok) after validating endpoints and kinds.
  std::string finalize();
};

// parse_g2o, g2o.rs:35-143.  Returns "" or an error message; io_error set when
// the file could not be read (Err(io) in the reference) as opposed to malformed.
std::string load_g2o(const char *path, HostGraph &g, bool &io_error);

// BASELINE config 4 generator.
void synth_grid(int width, int height, int64_t n_edges_target, uint64_t seed_meas,
                uint64_t seed_init, HostGraph &g);

}  // namespace rrpgo
