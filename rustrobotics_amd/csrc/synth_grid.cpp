// synth_grid.cpp -- deterministic SE(2) lattice pose graph (BASELINE config 4;
// construction rules fixed in SURVEY.md 8(d)).  Not reference behaviour: the
// reference has no generator and, as written, cannot run a graph this large
// (COO capacity len*len, pose_graph_optimization.rs:113-119).
//
//   * W x H unit lattice; pose index k: row y = k / W, x = k % W on even rows and
//     W-1 - k % W on odd rows (boustrophedon), heading 0 on even rows, pi on odd.
//   * edges, cell-major in (y,x) raster order, offset-minor over the stencil
//     (0,1),(1,0),(-1,1),(1,1),(0,2),(2,0),(-2,1),(-1,2),(1,2),(2,1); then, if
//     n_edges_target asks for more, offset (-2,2) edges in raster order.
//     from = the lower pose index.  400 x 250 -> 992,860 + 7,140 = 1,000,000.
//   * measurement = (x_from^-1 * x_to of the ground truth) + N(0, diag(.05^2,.05^2,.01^2)),
//     drawn in edge order from splitmix64(seed_meas) + Box-Muller;
//     information = diag(400, 400, 10000).
//   * initial guess = ground truth + N(0, diag(.1^2,.1^2,.02^2)) from seed_init.
#include <cmath>

#include "host_graph.h"

namespace rrpgo {

namespace {

struct Rng {
  uint64_t s;
  bool has_spare = false;
  double spare = 0.0;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {  // splitmix64
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  double normal() {  // Box-Muller, both outputs used
    if (has_spare) { has_spare = false; return spare; }
    double u1 = (double)((next() >> 11) + 1) * (1.0 / 9007199254740992.0);
    double u2 = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
    double r = std::sqrt(-2.0 * std::log(u1));
    double a = 6.283185307179586476925286766559 * u2;
    spare = r * std::sin(a);
    has_spare = true;
    return r * std::cos(a);
  }
};

}  // namespace

void synth_grid(int W, int H, int64_t n_edges_target, uint64_t seed_meas, uint64_t seed_init,
                HostGraph &g) {
  g = HostGraph();
  const int N = W * H;
  const double kPi = 3.14159265358979323846;
  auto pose_index = [&](int x, int y) { return y * W + ((y & 1) ? (W - 1 - x) : x); };
  std::vector<double> gx(N), gy(N), gth(N);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int k = pose_index(x, y);
      gx[k] = x;
      gy[k] = y;
      gth[k] = (y & 1) ? kPi : 0.0;
    }
  g.node_kind.assign(N, NODE_SE2);
  g.node_id.resize(N);
  g.node_state.resize((size_t)N * 3);
  Rng ri(seed_init);
  for (int k = 0; k < N; k++) {
    g.node_id[k] = (uint32_t)k;
    g.node_state[3 * (size_t)k + 0] = gx[k] + 0.1 * ri.normal();
    g.node_state[3 * (size_t)k + 1] = gy[k] + 0.1 * ri.normal();
    g.node_state[3 * (size_t)k + 2] = gth[k] + 0.02 * ri.normal();
  }
  static const int off[10][2] = {{0, 1}, {1, 0}, {-1, 1}, {1, 1}, {0, 2},
                                 {2, 0}, {-2, 1}, {-1, 2}, {1, 2}, {2, 1}};
  Rng rm(seed_meas);
  auto add_edge = [&](int a, int b) {
    int from = a < b ? a : b, to = a < b ? b : a;
    // x_from^-1 * x_to : translation R_from^T (t_to - t_from), angle th_to - th_from
    double c = std::cos(gth[from]), s = std::sin(gth[from]);
    double dx = gx[to] - gx[from], dy = gy[to] - gy[from];
    double lx = c * dx + s * dy, ly = -s * dx + c * dy;
    double dth = gth[to] - gth[from];
    g.edge_kind.push_back(EDGE_SE2);
    g.edge_from.push_back(from);
    g.edge_to.push_back(to);
    g.edge_meas.push_back(lx + 0.05 * rm.normal());
    g.edge_meas.push_back(ly + 0.05 * rm.normal());
    g.edge_meas.push_back(dth + 0.01 * rm.normal());
    const double info[6] = {400.0, 0.0, 0.0, 400.0, 0.0, 10000.0};
    g.edge_info.insert(g.edge_info.end(), info, info + 6);
  };
  auto full = [&]() { return n_edges_target > 0 && (int64_t)g.edge_kind.size() >= n_edges_target; };
  for (int y = 0; y < H && !full(); y++)
    for (int x = 0; x < W && !full(); x++)
      for (int o = 0; o < 10 && !full(); o++) {
        int x2 = x + off[o][0], y2 = y + off[o][1];
        if (x2 < 0 || x2 >= W || y2 >= H) continue;
        add_edge(pose_index(x, y), pose_index(x2, y2));
      }
  for (int y = 0; y + 2 < H && n_edges_target > 0 && !full(); y++)
    for (int x = 2; x < W && !full(); x++) add_edge(pose_index(x, y), pose_index(x - 2, y + 2));
  g.finalize();
}

}  // namespace rrpgo
