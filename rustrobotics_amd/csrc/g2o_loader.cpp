// g2o_loader.cpp -- g2o text loader with the reference's semantics
// (reference src/mapping/g2o.rs:35-143):
//   * whole file read, split into lines (a trailing '\r' is stripped like
//     str::lines() does), tokens split on ' ' ONLY, empty tokens dropped (:52);
//   * tags VERTEX_SE2 / VERTEX_XY / VERTEX_SE3:QUAT / EDGE_SE2 / EDGE_SE2_XY /
//     EDGE_SE3:QUAT, anything else is an error (unimplemented!() at :138);
//   * scalar offsets grow by 3 / 2 / 6 in vertex FILE order (:60-77);
//   * edges keep FILE order (:94-95,111-112,135-136);
//   * information matrices are given as row-major upper triangles (:82,100,117).
// A condition that panics in the reference (empty line, wrong value count, edge
// to an unknown vertex) is reported as an error string here.
// SE(3) quaternions are read in the g2o text order qx qy qz qw; the reference's
// iso3 passes them to nalgebra as (w,i,j,k) (g2o.rs:20) but never
// executes SE(3), so there is no behaviour to match.
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string_view>
#include <thread>
#include <unordered_map>

#include "host_graph.h"
#include "host_threads.h"

namespace rrpgo {

namespace {

bool read_file(const char *path, std::string &out) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0) { std::fclose(f); return false; }
  out.resize((size_t)n);
  size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

bool to_u32(std::string_view s, uint32_t &v) {
  if (!s.empty() && s[0] == '+') s.remove_prefix(1);
  if (s.empty()) return false;
  auto r = std::from_chars(s.data(), s.data() + s.size(), v, 10);
  return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

bool to_f64(std::string_view s, double &v) {
  // Rust's f64::from_str: optional sign, decimal / exponent, inf / nan; no hex.  std::from_chars parses
  // the same grammar (correctly rounded, no locale, no copy) except for a leading '+'.
  if (s.empty()) return false;
  if (s[0] == '+') {
    s.remove_prefix(1);
    if (s.empty() || s[0] == '-' || s[0] == '+') return false;
  }
  auto r = std::from_chars(s.data(), s.data() + s.size(), v, std::chars_format::general);
  if (r.ec == std::errc() && r.ptr == s.data() + s.size()) return true;
  // anything unusual (overflow to infinity, underflow, "infinity") takes the strtod route
  if (s.size() > 62) return false;
  char buf[64];
  std::memcpy(buf, s.data(), s.size());
  buf[s.size()] = 0;
  const char *p = buf;
  if (*p == '-') p++;
  if (p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) return false;
  char *end = nullptr;
  v = std::strtod(buf, &end);
  return end != buf && *end == 0;
}

}  // namespace

std::string HostGraph::finalize() {
  const int N = n_nodes(), E = n_edges();
  node_offset.assign(N, 0);
  node_state_off.assign(N, 0);
  int off = 0;
  int64_t soff = 0;
  has_se3 = has_2d = false;
  for (int i = 0; i < N; i++) {
    int k = node_kind[i];
    if (k < NODE_SE2 || k > NODE_SE3) return "bad node kind";
    node_offset[i] = off;
    node_state_off[i] = soff;
    off += node_dim(k);
    soff += node_state_len(k);
    (k == NODE_SE3 ? has_se3 : has_2d) = true;
  }
  dim = off;
  if ((int64_t)node_state.size() != soff) return "node_state length does not match node kinds";
  if (has_se3 && has_2d) return "mixed 2D / 3D graphs are not supported";
  edge_meas_off.assign(E, 0);
  edge_info_off.assign(E, 0);
  int64_t mo = 0, io = 0;
  anchor_node = -1;
  for (int k = 0; k < E; k++) {
    int ek = edge_kind[k];
    if (ek < EDGE_SE2 || ek > EDGE_SE3) return "bad edge kind";
    int a = edge_from[k], b = edge_to[k];
    if (a < 0 || a >= N || b < 0 || b >= N)
      return "edge " + std::to_string(k) + " references an unknown vertex";
    int ka = node_kind[a], kb = node_kind[b];
    bool ok = (ek == EDGE_SE2 && ka == NODE_SE2 && kb == NODE_SE2) ||
              (ek == EDGE_SE2_XY && ka == NODE_SE2 && kb == NODE_XY) ||
              (ek == EDGE_SE3 && ka == NODE_SE3 && kb == NODE_SE3);
    if (!ok)  // unreachable!() in the reference, pose_graph_optimization.rs:315-320,342-347
      return "edge " + std::to_string(k) + ": endpoint kinds do not match the edge kind";
    if (a == b) return "edge " + std::to_string(k) + " is a self loop";
    edge_meas_off[k] = mo;
    edge_info_off[k] = io;
    mo += edge_meas_len(ek);
    io += edge_info_len(ek);
    // prior goes on the from-node of the first pose-pose edge in file order (:330-336)
    if (anchor_node < 0 && (ek == EDGE_SE2 || ek == EDGE_SE3)) anchor_node = a;
  }
  if ((int64_t)edge_meas.size() != mo) return "edge_meas length does not match edge kinds";
  if ((int64_t)edge_info.size() != io) return "edge_info length does not match edge kinds";
  return "";
}

namespace {

// what one thread parses: a run of whole lines
struct ParsedPart {
  std::vector<int32_t> node_kind, edge_kind, node_line;
  std::vector<uint32_t> node_id, from_id, to_id;
  std::vector<double> node_state, edge_meas, edge_info;
  long lines = 0;        // lines consumed (up to and including the failing one)
  std::string error;     // first failure of this part, without the line prefix
};

void parse_part(const char *text, size_t begin, size_t end, ParsedPart &out) {
  std::vector<std::string_view> tok;
  size_t pos = begin;
  auto fail = [&](const std::string &m) { out.error = m; };
  while (pos < end) {
    const char *nl = (const char *)std::memchr(text + pos, '\n', end - pos);
    const size_t eol = nl ? (size_t)(nl - text) : end;
    std::string_view line(text + pos, eol - pos);
    pos = eol + 1;
    out.lines++;
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    tok.clear();
    for (size_t i = 0; i < line.size();) {
      while (i < line.size() && line[i] == ' ') i++;
      size_t j = i;
      while (j < line.size() && line[j] != ' ') j++;
      if (j > i) tok.push_back(line.substr(i, j - i));
      i = j;
    }
    if (tok.empty()) return fail("empty line");  // line[0] panics in the reference (:53)

    int kind = -1, nvals = 0;
    bool is_edge = false;
    if (tok[0] == "VERTEX_SE2") { kind = NODE_SE2; nvals = 3; }
    else if (tok[0] == "VERTEX_XY") { kind = NODE_XY; nvals = 2; }
    else if (tok[0] == "VERTEX_SE3:QUAT") { kind = NODE_SE3; nvals = 7; }
    else if (tok[0] == "EDGE_SE2") { kind = EDGE_SE2; nvals = 9; is_edge = true; }
    else if (tok[0] == "EDGE_SE2_XY") { kind = EDGE_SE2_XY; nvals = 5; is_edge = true; }
    else if (tok[0] == "EDGE_SE3:QUAT") { kind = EDGE_SE3; nvals = 28; is_edge = true; }
    else return fail("unsupported tag '" + std::string(tok[0]) + "'");

    const size_t first = is_edge ? 3 : 2;
    if (tok.size() != first + (size_t)nvals)
      return fail("expected " + std::to_string(nvals) + " values after the ids");
    uint32_t id0 = 0, id1 = 0;
    if (!to_u32(tok[1], id0) || (is_edge && !to_u32(tok[2], id1))) return fail("bad vertex id");
    double v[28];
    for (int i = 0; i < nvals; i++)
      if (!to_f64(tok[first + i], v[i])) return fail("bad number '" + std::string(tok[first + i]) + "'");

    if (!is_edge) {
      out.node_kind.push_back(kind);
      out.node_id.push_back(id0);
      out.node_line.push_back((int32_t)out.lines);
      out.node_state.insert(out.node_state.end(), v, v + nvals);
    } else {
      out.edge_kind.push_back(kind);
      out.from_id.push_back(id0);
      out.to_id.push_back(id1);
      const int nm = edge_meas_len(kind);
      out.edge_meas.insert(out.edge_meas.end(), v, v + nm);
      out.edge_info.insert(out.edge_info.end(), v + nm, v + nvals);
    }
  }
}

}  // namespace

// The text is cut into a few runs of whole lines, parsed by one thread each (the numbers are the cost: ~50 000 from_chars calls
// for intel.g2o, 1 ms of the 7 ms closure the reference's bench times, benches/graph_slam.rs:9-10) and put together in file
// order: vertex order, edge order, the first failure in file order and its line number are those of a sequential pass.
std::string load_g2o(const char *path, HostGraph &g, bool &io_error) {
  io_error = false;
  std::string text;
  if (!read_file(path, text)) {
    io_error = true;
    return std::string("cannot read '") + path + "'";
  }
  g = HostGraph();
  int nparts = (int)std::min<size_t>(4, text.size() / (96u << 10) + 1);
  nparts = std::max(1, std::min<int>(nparts, (int)std::thread::hardware_concurrency()));
  std::vector<size_t> cut(nparts + 1, text.size());
  cut[0] = 0;
  for (int k = 1; k < nparts; k++) {
    size_t p = text.size() * (size_t)k / (size_t)nparts;
    p = std::max(p, cut[k - 1]);
    const size_t nl = text.find('\n', p);
    cut[k] = nl == std::string::npos ? text.size() : nl + 1;
  }
  std::vector<ParsedPart> parts(nparts);
  parallel_indices(nparts, nparts, [&](int k) { parse_part(text.data(), cut[k], cut[k + 1], parts[k]); });
  std::unordered_map<uint32_t, int32_t> index_of;  // id -> dense index (lut + nodes maps of the reference)
  std::vector<uint32_t> from_id, to_id;
  size_t nn = 0, ne = 0;
  for (const ParsedPart &p : parts) { nn += p.node_kind.size(); ne += p.edge_kind.size(); }
  index_of.reserve(nn * 2);
  g.node_kind.reserve(nn); g.node_id.reserve(nn);
  g.edge_kind.reserve(ne); from_id.reserve(ne); to_id.reserve(ne);
  long line0 = 0;
  for (const ParsedPart &p : parts) {
    for (size_t i = 0; i < p.node_kind.size(); i++) {
      // The reference would silently overwrite the hash-map entry and leave the
      // previous offset's rows empty (a singular system): rejected here.
      if (!index_of.emplace(p.node_id[i], (int32_t)g.node_kind.size()).second)
        return "line " + std::to_string(line0 + p.node_line[i]) + ": duplicate vertex id " + std::to_string(p.node_id[i]);
      g.node_kind.push_back(p.node_kind[i]);
      g.node_id.push_back(p.node_id[i]);
    }
    if (!p.error.empty()) return "line " + std::to_string(line0 + p.lines) + ": " + p.error;
    g.node_state.insert(g.node_state.end(), p.node_state.begin(), p.node_state.end());
    g.edge_kind.insert(g.edge_kind.end(), p.edge_kind.begin(), p.edge_kind.end());
    from_id.insert(from_id.end(), p.from_id.begin(), p.from_id.end());
    to_id.insert(to_id.end(), p.to_id.begin(), p.to_id.end());
    g.edge_meas.insert(g.edge_meas.end(), p.edge_meas.begin(), p.edge_meas.end());
    g.edge_info.insert(g.edge_info.end(), p.edge_info.begin(), p.edge_info.end());
    line0 += p.lines;
  }
  const size_t E = g.edge_kind.size();
  g.edge_from.resize(E);
  g.edge_to.resize(E);
  for (size_t k = 0; k < E; k++) {
    auto a = index_of.find(from_id[k]), b = index_of.find(to_id[k]);
    if (a == index_of.end() || b == index_of.end())
      return "edge " + std::to_string(k) + " references an unknown vertex";
    g.edge_from[k] = a->second;
    g.edge_to[k] = b->second;
  }
  return g.finalize();
}

}  // namespace rrpgo
