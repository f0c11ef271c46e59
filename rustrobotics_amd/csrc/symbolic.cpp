// symbolic.cpp -- see symbolic.h.  Host only, runs once per graph.
#include "symbolic.h"
#include "host_threads.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>
#include <thread>
#include <atomic>

namespace rrpgo {

namespace {

struct Adj {
  std::vector<int32_t> ptr, idx;
  int deg(int v) const { return ptr[v + 1] - ptr[v]; }
};

Adj build_adjacency(const HostGraph &g) {
  const int N = g.n_nodes(), E = g.n_edges();
  Adj a;
  a.ptr.assign(N + 1, 0);
  for (int k = 0; k < E; k++) {
    a.ptr[g.edge_from[k] + 1]++;
    a.ptr[g.edge_to[k] + 1]++;
  }
  for (int i = 0; i < N; i++) a.ptr[i + 1] += a.ptr[i];
  std::vector<int32_t> raw(a.ptr[N]), fill(a.ptr.begin(), a.ptr.end() - 1);
  for (int k = 0; k < E; k++) {
    raw[fill[g.edge_from[k]]++] = g.edge_to[k];
    raw[fill[g.edge_to[k]]++] = g.edge_from[k];
  }
  // sort + dedupe each list
  std::vector<int32_t> nptr(N + 1, 0);
  a.idx.reserve(raw.size());
  for (int i = 0; i < N; i++) {
    auto b = raw.begin() + a.ptr[i], e = raw.begin() + a.ptr[i + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    nptr[i] = (int32_t)a.idx.size();
    a.idx.insert(a.idx.end(), b, e);
  }
  nptr[N] = (int32_t)a.idx.size();
  a.ptr.swap(nptr);
  return a;
}

// ---------------------------------------------------------------------------
// Constrained minimum degree on a small local graph: vertices [0, ne) are
// eliminated, vertices [ne, nt) (not yet eliminated separator nodes) only take
// part in the degrees.  Quotient-graph formulation, exact external degree
// weighted by the scalar dimension of every node.
void constrained_min_degree(int ne, int nt, std::vector<std::vector<int32_t>> &A,
                            const std::vector<int32_t> &w, std::vector<int32_t> &out) {
  std::vector<std::vector<int32_t>> E(nt), members(nt);
  std::vector<char> gone(nt, 0), absorbed(nt, 0);
  std::vector<int32_t> deg(nt, 0), mark(nt, 0), mark2(nt, 0);
  int stamp = 0;
  // the queue: one small min-heap of vertex ids per degree (lazy deletion: an entry counts while its vertex is alive and
  // still has that degree) -- the same choice as ONE heap of (degree, id) pairs, smallest id among the smallest degree, at a
  // fraction of its cost (r04: a third of this function's time went into sifting a 13 000-entry heap of pairs)
  std::vector<std::vector<int32_t>> bucket;
  int dmin = 0;
  auto push = [&](int d, int v) {
    if (d >= (int)bucket.size()) bucket.resize((size_t)d + 64);
    auto &b = bucket[d];
    b.push_back(v);
    std::push_heap(b.begin(), b.end(), std::greater<int32_t>());
    dmin = std::min(dmin, d);
  };
  for (int v = 0; v < ne; v++) {
    int d = 0;
    for (int u : A[v]) d += w[u];
    deg[v] = d;
    push(d, v);
  }
  out.clear();
  out.reserve(ne);
  std::vector<int32_t> reach;
  for (int k = 0; k < ne; k++) {
    int p = -1;
    while (p < 0) {
      auto &b = bucket[dmin];
      if (b.empty()) { dmin++; continue; }
      std::pop_heap(b.begin(), b.end(), std::greater<int32_t>());
      const int v = b.back();
      b.pop_back();
      if (!gone[v] && deg[v] == dmin) p = v;
    }
    out.push_back(p);
    gone[p] = 1;
    reach.clear();
    const int sp = ++stamp;
    mark[p] = sp;
    for (int u : A[p])
      if (!gone[u] && mark[u] != sp) { mark[u] = sp; reach.push_back(u); }
    for (int e : E[p]) {
      if (absorbed[e]) continue;
      for (int u : members[e])
        if (!gone[u] && mark[u] != sp) { mark[u] = sp; reach.push_back(u); }
      absorbed[e] = 1;
      std::vector<int32_t>().swap(members[e]);
    }
    members[p] = reach;
    for (int i : reach) {
      auto &ai = A[i];
      size_t wq = 0;
      for (int u : ai)
        if (!gone[u] && mark[u] != sp) ai[wq++] = u;
      ai.resize(wq);
      auto &ei = E[i];
      wq = 0;
      for (int e : ei)
        if (!absorbed[e]) ei[wq++] = e;
      ei.resize(wq);
      ei.push_back(p);
    }
    // exact external degrees of the vertices of the new element.  They all contain the element itself: its weight is
    // taken once (wreach), and only what lies OUTSIDE it is scanned -- the members of the older elements that are not in
    // `reach` (mark == sp) and the vertex's own edges (pruned of `reach` above).  (r04: the same degrees, hence the same
    // ordering, as scanning every element's members for every vertex -- |reach|^2 per step less; intel.g2o 2.5 -> 1.x ms.)
    // An older element that has nothing outside the new one is absorbed by it (it would never add to a degree again).
    int wreach = 0;
    for (int u : reach) wreach += w[u];
    for (int i : reach) {
      auto &ei = E[i];
      if (i >= ne) {   // constrained vertices never enter the queue (their element lists are only pruned)
        continue;
      }
      const int si = ++stamp;
      int d = wreach - w[i];
      for (int u : A[i])
        if (mark2[u] != si) { mark2[u] = si; d += w[u]; }
      size_t wq = 0;
      for (int e : ei) {
        if (e == p) { ei[wq++] = e; continue; }
        if (absorbed[e]) continue;
        int outside = 0;
        auto &me = members[e];
        size_t mq = 0;
        for (int u : me) {
          if (gone[u]) continue;
          me[mq++] = u;   // (eliminated members are dropped on the way: the list is scanned again for the next vertex)
          if (mark[u] != sp) {
            outside++;
            if (mark2[u] != si) { mark2[u] = si; d += w[u]; }
          }
        }
        me.resize(mq);
        if (outside == 0) { absorbed[e] = 1; std::vector<int32_t>().swap(members[e]); }
        else ei[wq++] = e;
      }
      ei.resize(wq);
      deg[i] = d;
      push(d, i);
    }
    std::vector<int32_t>().swap(A[p]);
    std::vector<int32_t>().swap(E[p]);
  }
}

// ---------------------------------------------------------------------------
// Multilevel bisection of a small graph: heavy-edge matching down to a few dozen vertices, greedy graph growing
// from several seeds there, Fiduccia-Mattheyses refinement of the edge cut on the way back up; then the smallest
// vertex separator the edge bisection admits -- a minimum vertex cover of its cut edges (Koenig's theorem on the
// bipartite graph of the two boundaries).  Everything is integer arithmetic with index tie-breaks: the same graph
// gives the same split on every host.  Measured against the breadth-first level sets and coordinate cuts of
// NestedDissection on the trajectory graphs (r05, intel down to leaves of 100 nodes): the separators on the heaviest
// root path add up to 60 nodes instead of 97 (top separator 20 instead of 26, the two below it 7 and 11 instead of
// 13 and 38); a Fiedler-vector bisection with the same cover step, computed offline with scipy, gives 71 (its top
// separator is 12 nodes at a 55 : 45 split -- the score below prefers the better balanced 20).
struct MlGraph {
  int n = 0;
  std::vector<int32_t> ptr, idx, ew, vw;
};

struct MultilevelBisection {
  // One coarsening step.  cmap[v] = coarse vertex of v.
  static void coarsen(const MlGraph &g, int32_t max_vw, MlGraph &c, std::vector<int32_t> &cmap) {
    const int n = g.n;
    std::vector<int32_t> perm(n), match(n, -1), rep;
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return g.ptr[a + 1] - g.ptr[a] < g.ptr[b + 1] - g.ptr[b]; });
    cmap.assign(n, -1);
    rep.reserve(n);
    for (int v : perm) {
      if (match[v] >= 0) continue;
      int best = -1;
      int32_t bw = 0;
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
        const int u = g.idx[p];
        if (match[u] >= 0 || u == v || g.vw[v] + g.vw[u] > max_vw) continue;
        if (best < 0 || g.ew[p] > bw || (g.ew[p] == bw && g.vw[u] < g.vw[best])) { best = u; bw = g.ew[p]; }
      }
      match[v] = best >= 0 ? best : v;
      if (best >= 0) { match[best] = v; cmap[best] = (int32_t)rep.size(); }
      cmap[v] = (int32_t)rep.size();
      rep.push_back(v);
    }
    c.n = (int)rep.size();
    c.vw.assign(c.n, 0);
    c.ptr.assign(c.n + 1, 0);
    c.idx.clear();
    c.ew.clear();
    std::vector<int32_t> slot(c.n, -1);
    for (int cv = 0; cv < c.n; cv++) {
      const int start = (int)c.idx.size();
      const int fv[2] = {rep[cv], match[rep[cv]]};
      for (int t = 0; t < (fv[1] == fv[0] ? 1 : 2); t++) {
        const int f = fv[t];
        c.vw[cv] += g.vw[f];
        for (int p = g.ptr[f]; p < g.ptr[f + 1]; p++) {
          const int cu = cmap[g.idx[p]];
          if (cu == cv) continue;
          if (slot[cu] < start) {
            slot[cu] = (int32_t)c.idx.size();
            c.idx.push_back(cu);
            c.ew.push_back(g.ew[p]);
          } else {
            c.ew[slot[cu]] += g.ew[p];
          }
        }
      }
      c.ptr[cv + 1] = (int32_t)c.idx.size();
    }
  }

  static int64_t cut_of(const MlGraph &g, const std::vector<int8_t> &side) {
    int64_t c = 0;
    for (int v = 0; v < g.n; v++)
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++)
        if (g.idx[p] > v && side[g.idx[p]] != side[v]) c += g.ew[p];
    return c;
  }

  // One Fiduccia-Mattheyses pass: vertices move one at a time in order of gain (each at most once), the best prefix
  // of the move sequence is kept.  A move may not take a side below min_side.  Returns whether anything was kept.
  static bool fm_pass(const MlGraph &g, int64_t min_side, std::vector<int8_t> &side, int64_t W[2], int64_t &cut) {
    const int n = g.n;
    // (the arrays of a pass live as long as the thread: a dissection runs thousands of passes over graphs of a few dozen vertices)
    static thread_local std::vector<int32_t> gain, moves;
    static thread_local std::vector<char> locked;
    static thread_local std::vector<std::pair<int32_t, int32_t>> heap;   // (gain, -vertex): the larger gain first, then the smaller index
    gain.resize(n);
    locked.assign(n, 0);
    heap.clear();
    moves.clear();
    auto push = [&](int32_t gn, int v) { heap.emplace_back(gn, -v); std::push_heap(heap.begin(), heap.end()); };
    for (int v = 0; v < n; v++) {
      int32_t ext = 0, in = 0;
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) (side[g.idx[p]] != side[v] ? ext : in) += g.ew[p];
      gain[v] = ext - in;
      if (ext > 0) push(gain[v], v);
    }
    auto absdiff = [&]() { return W[0] > W[1] ? W[0] - W[1] : W[1] - W[0]; };
    int64_t best_cut = cut, best_bal = absdiff();
    size_t best_len = 0;
    int since = 0;
    const int limit = std::max(24, std::min(n / 6, 200));
    while (!heap.empty() && since < limit) {
      std::pop_heap(heap.begin(), heap.end());
      const auto top = heap.back();
      heap.pop_back();
      const int v = -top.second;
      if (locked[v] || top.first != gain[v]) continue;
      const int a = side[v];
      if (W[a] - g.vw[v] < min_side) continue;
      side[v] = (int8_t)(1 - a);
      W[a] -= g.vw[v];
      W[1 - a] += g.vw[v];
      cut -= gain[v];
      locked[v] = 1;
      moves.push_back(v);
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
        const int u = g.idx[p];
        if (locked[u]) continue;
        gain[u] += (side[u] == a ? 2 : -2) * g.ew[p];
        push(gain[u], u);
      }
      const int64_t bal = absdiff();
      if (cut < best_cut || (cut == best_cut && bal < best_bal)) {
        best_cut = cut;
        best_bal = bal;
        best_len = moves.size();
        since = 0;
      } else {
        since++;
      }
    }
    for (size_t i = moves.size(); i-- > best_len;) {
      const int v = moves[i], a = side[v];
      side[v] = (int8_t)(1 - a);
      W[a] -= g.vw[v];
      W[1 - a] += g.vw[v];
    }
    cut = best_cut;
    return best_len > 0;
  }

  static int64_t refine(const MlGraph &g, int64_t min_side, std::vector<int8_t> &side, int passes) {
    int64_t W[2] = {0, 0};
    for (int v = 0; v < g.n; v++) W[side[v]] += g.vw[v];
    int64_t cut = cut_of(g, side);
    for (int i = 0; i < passes; i++)
      if (!fm_pass(g, min_side, side, W, cut)) break;
    return cut;
  }

  // Region 0 grows from `seed` by the vertex most strongly tied to it until it holds half the weight.
  static void grow(const MlGraph &g, int seed, int64_t half, std::vector<int8_t> &side) {
    const int n = g.n;
    side.assign(n, 1);
    static thread_local std::vector<int32_t> conn, degw;
    conn.assign(n, 0);
    degw.assign(n, 0);
    for (int v = 0; v < n; v++)
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) degw[v] += g.ew[p];
    int64_t W0 = 0;
    int v = seed;
    while (v >= 0) {
      side[v] = 0;
      W0 += g.vw[v];
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) conn[g.idx[p]] += g.ew[p];
      if (W0 >= half) break;
      int best = -1, first_out = -1;
      int32_t bg = 0;
      for (int u = 0; u < n; u++) {
        if (side[u] == 0) continue;
        if (first_out < 0) first_out = u;
        if (conn[u] == 0) continue;
        const int32_t gn = 2 * conn[u] - degw[u];
        if (best < 0 || gn > bg) { best = u; bg = gn; }
      }
      v = best >= 0 ? best : first_out;   // (a graph in several pieces: go on with the next piece)
    }
  }

  // Breadth-first order from `seed`: the cheap initial split for a coarsest graph that stayed large.
  static void grow_bfs(const MlGraph &g, int seed, int64_t half, std::vector<int8_t> &side) {
    const int n = g.n;
    side.assign(n, 1);
    std::vector<int32_t> q;
    std::vector<char> seen(n, 0);
    int64_t W0 = 0;
    int next_unseen = 0;
    q.push_back(seed);
    seen[seed] = 1;
    for (size_t h = 0; W0 < half; h++) {
      if (h == q.size()) {
        while (next_unseen < n && seen[next_unseen]) next_unseen++;
        if (next_unseen == n) break;
        q.push_back(next_unseen);
        seen[next_unseen] = 1;
      }
      const int v = q[h];
      side[v] = 0;
      W0 += g.vw[v];
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++)
        if (!seen[g.idx[p]]) { seen[g.idx[p]] = 1; q.push_back(g.idx[p]); }
    }
  }

  // The coarsening is a function of the graph alone: one hierarchy serves every bisection of it.
  struct Hierarchy {
    std::vector<MlGraph> coarse;
    std::vector<std::vector<int32_t>> cmaps;
    std::vector<std::vector<int8_t>> grown;   // the coarsest graph split by region growing from a few seeds, unrefined
    int64_t total = 0;
  };
  static void build_hierarchy(const MlGraph &g0, Hierarchy &h) {
    h.total = 0;
    for (int v = 0; v < g0.n; v++) h.total += g0.vw[v];
    const int32_t max_vw = (int32_t)std::max<int64_t>(1, h.total / 24);
    const MlGraph *cur = &g0;
    while (cur->n > 48) {
      MlGraph c;
      std::vector<int32_t> cmap;
      coarsen(*cur, max_vw, c, cmap);
      if (c.n * 10 > cur->n * 9) break;
      h.coarse.push_back(std::move(c));
      h.cmaps.push_back(std::move(cmap));
      cur = &h.coarse.back();
    }
    // (four seeds, two passes each: ten seeds and four passes were 60 % of the time of a dissection and bought nothing)
    const MlGraph &g = *cur;
    const int n_seeds = std::min(g.n, 4);
    h.grown.resize(n_seeds);
    for (int s = 0; s < n_seeds; s++) {
      const int seed = (int)((int64_t)s * g.n / n_seeds);
      if (g.n <= 400) grow(g, seed, h.total / 2, h.grown[s]);
      else grow_bfs(g, seed, h.total / 2, h.grown[s]);
    }
  }

  // side[v] in {0, 1}; both sides keep at least min_side of the vertex weight where the graph allows it.
  static void bisect(const MlGraph &g0, const Hierarchy &h, int64_t min_side, std::vector<int8_t> &side) {
    const int64_t total = h.total;
    {
      const MlGraph &g = h.coarse.empty() ? g0 : h.coarse.back();
      std::vector<int8_t> trial;
      int64_t best_cut = -1, best_bal = 0;
      for (const std::vector<int8_t> &start : h.grown) {
        trial = start;
        const int64_t cut = refine(g, min_side, trial, 2);
        int64_t W0 = 0;
        for (int v = 0; v < g.n; v++) if (trial[v] == 0) W0 += g.vw[v];
        if (W0 < min_side || total - W0 < min_side) continue;
        const int64_t bal = W0 * 2 > total ? W0 * 2 - total : total - W0 * 2;
        if (best_cut < 0 || cut < best_cut || (cut == best_cut && bal < best_bal)) { best_cut = cut; best_bal = bal; side = trial; }
      }
      if (best_cut < 0) side = h.grown[0];   // no seed gave two sides of the least weight: keep the first growth as it is
    }
    for (size_t l = h.coarse.size(); l-- > 0;) {
      const MlGraph &fine = l == 0 ? g0 : h.coarse[l - 1];
      std::vector<int8_t> fs(fine.n);
      for (int v = 0; v < fine.n; v++) fs[v] = side[h.cmaps[l][v]];
      side.swap(fs);
      refine(fine, min_side, side, 4);
    }
  }

  // Minimum vertex cover of the cut edges: in_sep[v] = 1 for the vertices of the separator.
  static void cover_separator(const MlGraph &g, const std::vector<int8_t> &side, std::vector<char> &in_sep) {
    const int n = g.n;
    in_sep.assign(n, 0);
    std::vector<int32_t> L, rid(n, -1), R;
    for (int v = 0; v < n; v++) {
      bool b = false;
      for (int p = g.ptr[v]; p < g.ptr[v + 1] && !b; p++) b = side[g.idx[p]] != side[v];
      if (!b) continue;
      if (side[v] == 0) L.push_back(v);
      else { rid[v] = (int32_t)R.size(); R.push_back(v); }
    }
    const int nl = (int)L.size(), nr = (int)R.size();
    std::vector<int32_t> match_l(nl, -1), match_r(nr, -1), stamp(nr, -1);
    // augmenting paths (Kuhn); the depth of the search is bounded by the size of the boundary
    struct Matcher {
      const MlGraph &g;
      const std::vector<int8_t> &side;
      const std::vector<int32_t> &L, &rid;
      std::vector<int32_t> &match_l, &match_r, &stamp;
      bool augment(int l, int tag) {
        const int v = L[l];
        for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
          const int r = rid[g.idx[p]];
          if (r < 0 || side[g.idx[p]] == 0 || stamp[r] == tag) continue;
          stamp[r] = tag;
          if (match_r[r] < 0 || augment(match_r[r], tag)) { match_r[r] = l; match_l[l] = r; return true; }
        }
        return false;
      }
    } matcher{g, side, L, rid, match_l, match_r, stamp};
    for (int l = 0; l < nl; l++) matcher.augment(l, l);
    // Koenig: Z = vertices reachable from the unmatched left vertices along alternating paths
    std::vector<char> zl(nl, 0), zr(nr, 0);
    std::vector<int32_t> stack;
    for (int l = 0; l < nl; l++) if (match_l[l] < 0) { zl[l] = 1; stack.push_back(l); }
    while (!stack.empty()) {
      const int l = stack.back();
      stack.pop_back();
      const int v = L[l];
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
        const int r = rid[g.idx[p]];
        if (r < 0 || side[g.idx[p]] == 0 || zr[r] || match_l[l] == r) continue;
        zr[r] = 1;
        const int l2 = match_r[r];
        if (l2 >= 0 && !zl[l2]) { zl[l2] = 1; stack.push_back(l2); }
      }
    }
    for (int l = 0; l < nl; l++) if (!zl[l]) in_sep[L[l]] = 1;
    for (int r = 0; r < nr; r++) if (zr[r]) in_sep[R[r]] = 1;
  }
};

// ---------------------------------------------------------------------------
// Nested dissection with breadth-first level-set separators.
struct NestedDissection {
  const Adj &adj;
  const std::vector<int32_t> &w;
  const SymbolicOptions &opt;
  int N;
  std::vector<int32_t> set_id;   // which recursion set a node currently belongs to
  std::vector<char> ordered;
  std::vector<int32_t> order;    // output: order[pos] = node
  std::vector<int32_t> part;     // owner partition, -1 shared
  std::vector<int32_t> level, queue;
  std::vector<int8_t> side;      // scratch of the coordinate bisection
  std::vector<int32_t> sep_of;   // separator a node belongs to (-1: inside a leaf)
  const HostGraph *hg = nullptr; // node positions for coordinate bisection (optional)
  bool splits_only = false;      // dissect_only(): the caller wants the recorded splits, not the order -- leaves are not ordered
  std::atomic<int> next_set{1};  // (ids only ever compared for equality: the two halves of a split may be dissected by two threads)
  int part_depth = 0;

  NestedDissection(const Adj &a, const std::vector<int32_t> &ww, const SymbolicOptions &o)
      : adj(a), w(ww), opt(o), N((int)ww.size()), set_id(N, 0), ordered(N, 0), part(N, -1),
        level(N, -1), side(N, -1), sep_of(N, -1) {
    order.reserve(N);
    queue.reserve(N);
    while ((1 << part_depth) < opt.n_parts) part_depth++;
  }

  // BFS inside set `sid` from `root`; fills queue (visit order) and level[].
  // Returns the number of levels.  level[] of visited nodes must be reset by caller.
  int bfs(int root, int sid, std::vector<int32_t> &visit) {
    visit.clear();
    visit.push_back(root);
    level[root] = 0;
    int nlev = 1;
    for (size_t h = 0; h < visit.size(); h++) {
      int v = visit[h];
      for (int p = adj.ptr[v]; p < adj.ptr[v + 1]; p++) {
        int u = adj.idx[p];
        if (set_id[u] == sid && level[u] < 0) {
          level[u] = level[v] + 1;
          nlev = level[u] + 1;
          visit.push_back(u);
        }
      }
    }
    return nlev;
  }

  // `out`: where the order of this part of the graph is collected; `local_id`: scratch of the calling thread, all -1 between calls
  void order_leaf(const std::vector<int32_t> &S, std::vector<int32_t> &out, std::vector<int32_t> &local_id) {
    if (splits_only) {
      for (int v : S) { out.push_back(v); ordered[v] = 1; }
      return;
    }
    // local graph: S first, then not-yet-ordered outside neighbours (ancestor separators)
    const int ne = (int)S.size();
    std::vector<int32_t> verts(S);
    for (int i = 0; i < ne; i++) local_id[S[i]] = i;
    for (int i = 0; i < ne; i++)
      for (int p = adj.ptr[S[i]]; p < adj.ptr[S[i] + 1]; p++) {
        int u = adj.idx[p];
        if (local_id[u] < 0 && !ordered[u]) {
          local_id[u] = (int)verts.size();
          verts.push_back(u);
        }
      }
    const int nt = (int)verts.size();
    std::vector<std::vector<int32_t>> A(nt);
    std::vector<int32_t> lw(nt);
    for (int i = 0; i < nt; i++) lw[i] = w[verts[i]];
    for (int i = 0; i < ne; i++)
      for (int p = adj.ptr[S[i]]; p < adj.ptr[S[i] + 1]; p++) {
        int lu = local_id[adj.idx[p]];
        if (lu >= 0) {
          A[i].push_back(lu);
          if (lu >= ne) A[lu].push_back(i);
        }
      }
    std::vector<int32_t> lo;
    constrained_min_degree(ne, nt, A, lw, lo);
    for (int l : lo) {
      out.push_back(verts[l]);
      ordered[verts[l]] = 1;
    }
    for (int v : verts) local_id[v] = -1;
  }

  void assign_part(const std::vector<int32_t> &S, int depth, int path) {
    int p = depth >= part_depth ? (path >> (depth - part_depth)) : (path << (part_depth - depth));
    for (int v : S) part[v] = p;
  }

  // Coordinate bisection: pose graphs are spatial, and a breadth-first level set from a corner of a
  // rectangular map is an L-shaped front about twice as long as a straight cut.  Split S at the median
  // coordinate of its longest axis; the separator is the lighter of the two boundaries of the cut.
  // Returns the score (separator weight, penalised for imbalance like the level-set search) or < 0.
  double geo_split(const std::vector<int32_t> &S, int sid, std::vector<int32_t> &left,
                   std::vector<int32_t> &right, std::vector<int32_t> &sep) {
    if (!hg || !opt.geo_nd) return -1.0;
    const int n = (int)S.size(), nd = hg->has_se3 ? 3 : 2;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v : S)
      for (int a = 0; a < nd; a++) {
        const double c = hg->node_state[hg->node_state_off[v] + a];
        if (!(c == c)) return -1.0;
        lo[a] = std::min(lo[a], c);
        hi[a] = std::max(hi[a], c);
      }
    double ext = 0;
    for (int a = 0; a < nd; a++) ext = std::max(ext, hi[a] - lo[a]);
    if (!(ext > 0)) return -1.0;
    // candidates: every axis whose extent is at least half the longest, cut at five quantiles around the
    // median; the separator of a cut is the lighter of its two boundaries
    static const double kQuant[5] = {0.5, 0.46, 0.54, 0.42, 0.58};
    std::vector<std::pair<double, int32_t>> key(n);
    double best_score = -1.0;
    int best_ax = -1, best_cut = 0, best_pick = 0;
    for (int ax = 0; ax < nd; ax++) {
      if (hi[ax] - lo[ax] < 0.5 * ext) continue;
      for (int i = 0; i < n; i++) key[i] = {hg->node_state[hg->node_state_off[S[i]] + ax], S[i]};
      std::sort(key.begin(), key.end());
      for (int qi = 0; qi < 5; qi++) {
        int cut = std::min(std::max((int)(kQuant[qi] * n), 1), n - 1);
        {
          // snap to the widest gap between consecutive coordinates nearby: a cut through the middle of a
          // row of poses with almost equal coordinates would be ragged and its boundary a row thicker
          const int win = std::max(8, n / 48);
          int bi = cut;
          double bg = -1.0;
          for (int i = std::max(1, cut - win); i <= std::min(n - 1, cut + win); i++) {
            const double gap = key[i].first - key[i - 1].first;
            if (gap > bg) { bg = gap; bi = i; }
          }
          cut = bi;
        }
        for (int i = 0; i < n; i++) side[key[i].second] = i < cut ? 0 : 1;
        int64_t wb[2] = {0, 0}, nb[2] = {0, 0};
        for (int v : S) {
          const int sv = side[v];
          bool c = false;
          for (int p = adj.ptr[v]; p < adj.ptr[v + 1] && !c; p++) {
            const int u = adj.idx[p];
            c = set_id[u] == sid && side[u] == 1 - sv;
          }
          if (c) { wb[sv] += w[v]; nb[sv]++; }
        }
        const int pick = wb[0] <= wb[1] ? 0 : 1;
        if (nb[pick] == 0) continue;
        const double l = cut - (pick == 0 ? nb[0] : 0), r = (n - cut) - (pick == 1 ? nb[1] : 0);
        if (l <= 0 || r <= 0) continue;
        const double imb = std::fabs(l - r) / (double)n;
        const double sc = (double)wb[pick] * (1.0 + 4.0 * imb * imb) + (imb > 0.6 ? 1e6 * imb : 0.0);
        if (best_score < 0 || sc < best_score) { best_score = sc; best_ax = ax; best_cut = cut; best_pick = pick; }
      }
    }
    double score = -1.0;
    if (best_ax >= 0) {
      for (int i = 0; i < n; i++) key[i] = {hg->node_state[hg->node_state_off[S[i]] + best_ax], S[i]};
      std::sort(key.begin(), key.end());
      for (int i = 0; i < n; i++) side[key[i].second] = i < best_cut ? 0 : 1;
      left.clear(); right.clear(); sep.clear();
      std::vector<int32_t> bnd;
      for (int v : S) {
        if (side[v] != best_pick) continue;
        bool c = false;
        for (int p = adj.ptr[v]; p < adj.ptr[v + 1] && !c; p++) {
          const int u = adj.idx[p];
          c = set_id[u] == sid && side[u] == 1 - best_pick;
        }
        if (c) bnd.push_back(v);
      }
      for (int v : bnd) side[v] = 2;
      for (int v : S) (side[v] == 0 ? left : side[v] == 1 ? right : sep).push_back(v);
      score = best_score;
    }
    for (int v : S) side[v] = -1;
    return score;
  }

  // The multilevel bisection of S (see MultilevelBisection) with its minimum-cover separator; same score as the
  // other two searches.
  double ml_split(const std::vector<int32_t> &S, int sid, std::vector<int32_t> &left,
                  std::vector<int32_t> &right, std::vector<int32_t> &sep, std::vector<int32_t> &local_id) {
    const int n = (int)S.size();
    MlGraph g;
    g.n = n;
    g.ptr.assign(n + 1, 0);
    g.vw.assign(n, 1);
    for (int i = 0; i < n; i++) local_id[S[i]] = i;
    for (int i = 0; i < n; i++) {
      for (int p = adj.ptr[S[i]]; p < adj.ptr[S[i] + 1]; p++)
        if (set_id[adj.idx[p]] == sid) g.idx.push_back(local_id[adj.idx[p]]);
      g.ptr[i + 1] = (int32_t)g.idx.size();
    }
    g.ew.assign(g.idx.size(), 1);
    for (int v : S) local_id[v] = -1;
    // three bisections from one hierarchy, with a looser and a tighter balance: the smallest separator that is not too lopsided
    // wins (the same score as the other searches).  Measured through the front cost model on six trajectory / sphere / garage
    // graphs (r05): one setting 0.655 of the uncut critical path, the best of three 0.611.
    MultilevelBisection::Hierarchy hier;
    MultilevelBisection::build_hierarchy(g, hier);
    static const double kMinSide[3] = {0.30, 0.38, 0.46};
    const bool three = n >= 200;   // (below 200 nodes the middle setting alone: same estimates, a third less time)
    const int n_try = three ? 3 : 1;
    struct Try { std::vector<int8_t> sd; std::vector<char> in_sep; };
    Try tries[3];
    auto run = [&](int t) {
      MultilevelBisection::bisect(g, hier, (int64_t)(kMinSide[three ? t : 1] * n), tries[t].sd);
      MultilevelBisection::cover_separator(g, tries[t].sd, tries[t].in_sep);
    };
    if (splits_only && n >= 400) {
      // the recording dissection has the host to itself (the candidate analyses start when it is done): the three settings of its
      // large splits side by side -- they only read the graph and the hierarchy.  (Beside eight candidate analyses this
      // oversubscribed the host: 6.7 against 4.7 ms, r05.)
      parallel_indices(3, 3, [&](int t) { run(t); });
    } else {
      for (int t = 0; t < n_try; t++) run(t);
    }
    double best = -1.0;
    std::vector<int32_t> tl, tr, ts;
    for (int t = 0; t < n_try; t++) {   // (in the order of the settings, whichever thread finished first: the first best wins)
      tl.clear(); tr.clear(); ts.clear();
      int64_t ws = 0;
      for (int i = 0; i < n; i++) {
        if (tries[t].in_sep[i]) { ts.push_back(S[i]); ws += w[S[i]]; }
        else (tries[t].sd[i] == 0 ? tl : tr).push_back(S[i]);
      }
      if (tl.empty() || tr.empty() || ts.empty()) continue;
      const double imb = std::fabs((double)tl.size() - (double)tr.size()) / (double)n;
      const double sc = (double)ws * (1.0 + 4.0 * imb * imb) + (imb > 0.6 ? 1e6 * imb : 0.0);
      if (best < 0 || sc < best) { best = sc; left.swap(tl); right.swap(tr); sep.swap(ts); }
    }
    return best;
  }

  void dissect(std::vector<int32_t> &S, int depth, int path, std::vector<int32_t> &out, std::vector<int32_t> &local_id) {
    const int n = (int)S.size();
    if (n == 0) return;
    if (depth == part_depth || (depth < part_depth && n <= opt.nd_leaf)) assign_part(S, depth, path);
    if (n <= opt.nd_leaf) {
      order_leaf(S, out, local_id);
      return;
    }
    const int sid = set_id[S[0]];
    // The split of a set is a function of the set alone (not of the leaf size the caller stops at): when several dissection depths of
    // one graph are analysed side by side (the engine's candidates), the deepest is dissected once, recording, and the others replay.
    std::vector<int32_t> left, right, sep;
    bool replayed = false;
    if (opt.nd_replay)
      if (const NdSplit *sp = opt.nd_replay->find(depth, S[0], n)) {
        if (sp->as_leaf) {
          if (depth < part_depth) assign_part(S, depth, path);
          order_leaf(S, out, local_id);
          return;
        }
        left = sp->left;
        right = sp->right;
        sep = sp->sep;
        replayed = true;
      }
    if (!replayed) {
      // connected components
      std::vector<int32_t> visit, comp_of_start, comp_sizes;
      std::vector<std::vector<int32_t>> comps;
      for (int v : S)
        if (level[v] < 0) {
          bfs(v, sid, visit);
          comps.emplace_back(visit);
        }
      for (int v : S) level[v] = -1;
      if (comps.size() > 1) {
        std::sort(comps.begin(), comps.end(),
                  [](const auto &a, const auto &b) { return a.size() > b.size(); });
        size_t nl = 0, nr = 0;
        for (auto &c : comps) {
          auto &dst = nl <= nr ? left : right;
          (nl <= nr ? nl : nr) += c.size();
          dst.insert(dst.end(), c.begin(), c.end());
        }
      } else {
        // pseudo-peripheral root
        int root = S[0], nlev = 0;
        for (int it = 0; it < 6; it++) {
          int nl = bfs(root, sid, visit);
          int far = visit.back(), best_deg = 1 << 30;
          for (size_t i = visit.size(); i-- > 0 && level[visit[i]] == nl - 1;) {
            int d = adj.deg(visit[i]);
            if (d < best_deg) { best_deg = d; far = visit[i]; }
          }
          bool grew = nl > nlev;
          nlev = nl;
          if (!grew && it > 0) break;
          if (it < 5) {
            for (int v : visit) level[v] = -1;
            root = far;
          }
        }
        // make sure levels correspond to `root`
        for (int v : S) level[v] = -1;
        nlev = bfs(root, sid, visit);
        if (nlev < 3) {  // no usable level separator: treat as a leaf
          for (int v : S) level[v] = -1;
          if (opt.nd_record) opt.nd_record->store(depth, S[0], n, true, left, right, sep);
          if (depth < part_depth) assign_part(S, depth, path);
          order_leaf(S, out, local_id);
          return;
        }
        std::vector<int64_t> cnt(nlev, 0), wsum(nlev, 0);
        for (int v : visit) { cnt[level[v]]++; wsum[level[v]] += w[v]; }
        int64_t total = n, acc = 0;
        int best = -1;
        double best_score = 1e300;
        for (int j = 0; j < nlev; j++) {
          int64_t l = acc, r = total - acc - cnt[j];
          acc += cnt[j];
          if (j == 0 || j == nlev - 1) continue;
          double imb = std::fabs((double)l - (double)r) / (double)total;
          double score = (double)wsum[j] * (1.0 + 4.0 * imb * imb) + (imb > 0.6 ? 1e6 * imb : 0.0);
          if (score < best_score) { best_score = score; best = j; }
        }
        for (int v : visit) {
          int lv = level[v];
          if (lv < best) left.push_back(v);
          else if (lv > best) right.push_back(v);
          else {
            bool touches_right = false;
            for (int p = adj.ptr[v]; p < adj.ptr[v + 1] && !touches_right; p++) {
              int u = adj.idx[p];
              touches_right = set_id[u] == sid && level[u] == best + 1;
            }
            (touches_right ? sep : left).push_back(v);
          }
        }
        for (int v : S) level[v] = -1;
        {
          // the straight cut, if the graph has positions and it is the lighter separator
          int64_t ws = 0;
          for (int v : sep) ws += w[v];
          const double imb = std::fabs((double)left.size() - (double)right.size()) / (double)n;
          const double bfs_score = (double)ws * (1.0 + 4.0 * imb * imb) + (imb > 0.6 ? 1e6 * imb : 0.0);
          std::vector<int32_t> gl, gr, gs;
          const double gscore = geo_split(S, sid, gl, gr, gs);
          double cur_score = bfs_score;
          if (gscore >= 0 && gscore < bfs_score) { left.swap(gl); right.swap(gr); sep.swap(gs); cur_score = gscore; }
          if (opt.ml_nd) {
            const double mscore = ml_split(S, sid, gl, gr, gs, local_id);
            if (mscore >= 0 && mscore < cur_score) { left.swap(gl); right.swap(gr); sep.swap(gs); }
          }
        }
      }
      if (opt.nd_record) opt.nd_record->store(depth, S[0], n, false, left, right, sep);
    }
    if (depth == 0 && opt.pin_node >= 0 && part_depth > 0) {
      // sharded runs: the anchor joins the top separator, so that every rank holds its solution entries
      // (the gauge transfer of the single-precision factor needs them on every rank before the update)
      auto drop = [&](std::vector<int32_t> &v) {
        for (size_t i = 0; i < v.size(); i++)
          if (v[i] == opt.pin_node) { v.erase(v.begin() + (long)i); return true; }
        return false;
      };
      if (drop(left) || drop(right)) sep.push_back(opt.pin_node);
    }
    if (depth < part_depth)
      for (int v : sep) part[v] = -1;
    const int left_id = next_set++, right_id = next_set++, zid = next_set++;
    for (int v : left) set_id[v] = left_id;
    for (int v : right) set_id[v] = right_id;
    for (int v : sep) set_id[v] = zid;
    std::vector<int32_t>().swap(S);
    // No edge joins the two halves, so their dissections touch disjoint entries of the per-node arrays (the scratch that maps
    // separator nodes of the ancestors is per thread): the top two levels of a small graph's multilevel dissection -- three
    // quarters of the time of its analysis -- run on up to four threads.  Same order as the sequential run: left, right, separator.
    if (opt.ml_nd && depth < (splits_only ? 3 : 2) && left.size() >= 150 && right.size() >= 150) {   // (alone on the host when only recording: one level more)
      std::vector<int32_t> out_right, scratch_right(N, -1);
      run_beside([&] { dissect(right, depth + 1, path * 2 + 1, out_right, scratch_right); },
                 [&] { dissect(left, depth + 1, path * 2, out, local_id); });
      out.insert(out.end(), out_right.begin(), out_right.end());
    } else {
      dissect(left, depth + 1, path * 2, out, local_id);
      dissect(right, depth + 1, path * 2 + 1, out, local_id);
    }
    for (int v : sep) {
      out.push_back(v);
      ordered[v] = 1;
      sep_of[v] = zid;
    }
  }
};

// Elimination tree of the permuted node graph (Liu, with path compression).
void elimination_tree(const Adj &adj, const std::vector<int32_t> &order,
                      const std::vector<int32_t> &pos_of, std::vector<int32_t> &parent) {
  const int N = (int)order.size();
  parent.assign(N, -1);
  std::vector<int32_t> anc(N, -1);
  for (int j = 0; j < N; j++) {
    int v = order[j];
    for (int p = adj.ptr[v]; p < adj.ptr[v + 1]; p++) {
      int i = pos_of[adj.idx[p]];
      while (i != -1 && i < j) {
        int nx = anc[i];
        anc[i] = j;
        if (nx == -1) parent[i] = j;
        i = nx;
      }
    }
  }
}

void postorder_forest(const std::vector<int32_t> &parent, std::vector<int32_t> &post) {
  const int N = (int)parent.size();
  std::vector<int32_t> head(N, -1), next(N, -1);
  for (int j = N - 1; j >= 0; j--)
    if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
  post.clear();
  post.reserve(N);
  std::vector<int32_t> stack;
  for (int r = 0; r < N; r++) {
    if (parent[r] != -1) continue;
    stack.push_back(r);
    while (!stack.empty()) {
      int v = stack.back();
      int c = head[v];
      if (c != -1) {
        head[v] = next[c];
        stack.push_back(c);
      } else {
        post.push_back(v);
        stack.pop_back();
      }
    }
  }
}

// Least-squares fit to the in-kernel stamps on MI355X (fp64; 966 fronts of intel, M3500 and dlr,
// scripts/fit_front_cost.py, rms residual 0.34 us on a mean of 11 us): a fixed part (zero, assemble,
// store), the 16-column blocks of the pivot chain (diagonal block + triangular solve + look-ahead, all
// latency bound), one global round trip per child in the extend-add, the last block's Schur complement
// and the LDS image that is zeroed and copied out.
double front_cost_us(int nc, int nr, int kids) {
  const double nu = nr + 1;
  const double elems = (double)(nc + nr + 1) * nc + 0.5 * nu * nu;
  return 2.3 + 3.37 * (double)((nc + 15) / 16) + 0.65 * kids + 1.12e-5 * (double)nc * nu * nu + 3.6e-4 * elems;
}

int64_t lds_elems(int nc, int nr) {
  return (int64_t)(nc + nr + 1) * nc + (int64_t)(nr + 1) * (nr + 2) / 2;
}

}  // namespace

// RR_PGO_ANALYZE_TIMES=1: wall time of every phase of analyze() on stderr (diagnostic)
struct PhaseTimer {
  bool on = std::getenv("RR_PGO_ANALYZE_TIMES") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char *what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "analyze: %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

std::string dissect_only(const HostGraph &g, const SymbolicOptions &opt) {
  const int N = g.n_nodes();
  if (N == 0) return "graph has no vertices";
  std::vector<int32_t> w(N);
  for (int i = 0; i < N; i++) w[i] = node_dim(g.node_kind[i]);
  const Adj adj = build_adjacency(g);
  NestedDissection nd(adj, w, opt);
  nd.hg = &g;
  nd.splits_only = true;
  std::vector<int32_t> all(N), scratch(N, -1);
  std::iota(all.begin(), all.end(), 0);
  nd.dissect(all, 0, 0, nd.order, scratch);
  return (int)nd.order.size() == N ? "" : "internal: ordering lost nodes";
}

std::string analyze(const HostGraph &g, const SymbolicOptions &opt, Symbolic &sym) {
  sym = Symbolic();
  const int N = g.n_nodes(), E = g.n_edges();
  if (N == 0) return "graph has no vertices";
  sym.N = N;
  sym.dim = g.dim;
  std::vector<int32_t> w(N);
  for (int i = 0; i < N; i++) w[i] = node_dim(g.node_kind[i]);
  const Adj adj = build_adjacency(g);
  const int64_t lds_budget = opt.lds_budget_elems;

  PhaseTimer ptimer;
  // ---- 1. ordering ---------------------------------------------------------
  std::vector<int32_t> order, pos_of(N), node_sep;
  {
    NestedDissection nd(adj, w, opt);
    nd.hg = &g;
    std::vector<int32_t> all(N);
    std::iota(all.begin(), all.end(), 0);
    std::vector<int32_t> scratch(N, -1);
    nd.dissect(all, 0, 0, nd.order, scratch);
    order.swap(nd.order);
    sym.node_part.swap(nd.part);
    node_sep.swap(nd.sep_of);
    if ((int)order.size() != N) return "internal: ordering lost nodes";
    if (opt.n_parts <= 1) std::fill(sym.node_part.begin(), sym.node_part.end(), 0);
  }
  for (int p = 0; p < N; p++) pos_of[order[p]] = p;

  ptimer.mark("ordering");
  // ---- 2. etree + postorder -------------------------------------------------
  std::vector<int32_t> parent;
  elimination_tree(adj, order, pos_of, parent);
  {
    std::vector<int32_t> post;
    postorder_forest(parent, post);
    std::vector<int32_t> norder(N);
    for (int p = 0; p < N; p++) norder[p] = order[post[p]];
    order.swap(norder);
    for (int p = 0; p < N; p++) pos_of[order[p]] = p;
    elimination_tree(adj, order, pos_of, parent);
  }

  ptimer.mark("etree");
  // ---- 3. node-level column structures -> weighted counts -------------------
  std::vector<int64_t> cc(N, 0);  // scalar rows strictly below the diagonal block
  {
    std::vector<std::vector<int32_t>> st(N);
    std::vector<int32_t> mark(N, -1), head(N, -1), next(N, -1);
    int64_t nblk = 0;
    for (int j = 0; j < N; j++) {
      std::vector<int32_t> &s = st[j];
      mark[j] = j;
      int v = order[j];
      for (int p = adj.ptr[v]; p < adj.ptr[v + 1]; p++) {
        int i = pos_of[adj.idx[p]];
        if (i > j && mark[i] != j) { mark[i] = j; s.push_back(i); }
      }
      for (int c = head[j]; c != -1; c = next[c]) {
        for (int x : st[c])
          if (x != j && mark[x] != j) { mark[x] = j; s.push_back(x); }
        std::vector<int32_t>().swap(st[c]);
      }
      std::sort(s.begin(), s.end());
      int64_t sum = 0;
      for (int x : s) sum += w[order[x]];
      cc[j] = sum;
      nblk += (int64_t)s.size() + 1;
      sym.nnz_l_entries += (int64_t)w[v] * (w[v] + sum);   // the column's d x d diagonal block and the blocks below it, no padding
      if (!s.empty()) {
        int pj = s[0];
        if (parent[j] != pj) return "internal: etree mismatch";
        next[j] = head[pj];
        head[pj] = j;
      }
    }
    sym.nnz_l_blocks = nblk;
  }

  ptimer.mark("colcounts");
  // ---- 4. zero-fill supernodes, then relaxed amalgamation -------------------
  std::vector<int32_t> sn_of(N), s_first, s_last;
  for (int j = 0; j < N; j++) {
    // The last separator of a dissected region has the structure of its parent separator and would
    // chain into one supernode with it; kept apart, it is factored in the same batch as its sibling
    // separator one level down, which shortens the sequential panel chain of the wide top fronts.
    const bool other_sep = j > 0 && opt.split_separators && node_sep[order[j - 1]] >= 0 && node_sep[order[j]] >= 0 &&
                           node_sep[order[j - 1]] != node_sep[order[j]];
    bool join = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + w[order[j]] &&
                sym.node_part[order[j - 1]] == sym.node_part[order[j]] && !other_sep;
    if (!join) {
      s_first.push_back(j);
      s_last.push_back(j);
    } else {
      s_last.back() = j;
    }
    sn_of[j] = (int)s_first.size() - 1;
  }
  const int S0 = (int)s_first.size();
  std::vector<int32_t> sp(S0, -1);
  std::vector<int64_t> npiv(S0, 0), nrow(S0), zeros(S0, 0);
  for (int s = 0; s < S0; s++) {
    for (int j = s_first[s]; j <= s_last[s]; j++) npiv[s] += w[order[j]];
    nrow[s] = cc[s_last[s]];
    int pj = parent[s_last[s]];
    sp[s] = pj < 0 ? -1 : sn_of[pj];
  }
  std::vector<int32_t> merged_into(S0, -1);
  std::vector<std::vector<int32_t>> pre(S0);      // merged members, elimination order
  std::vector<std::vector<int32_t>> kids(S0);
  for (int s = 0; s < S0; s++)
    if (sp[s] >= 0) kids[sp[s]].push_back(s);
  auto part_of_sn = [&](int s) { return sym.node_part[order[s_first[s]]]; };
  for (int p = 0; p < S0; p++) {
    auto &ch = kids[p];
    std::sort(ch.begin(), ch.end(), [&](int a, int b) { return nrow[a] > nrow[b]; });
    for (int c : ch) {
      if (part_of_sn(c) != part_of_sn(p)) continue;
      int64_t np2 = npiv[c] + npiv[p];
      if (opt.split_separators && np2 > 256 && node_sep[order[s_first[c]]] >= 0 && node_sep[order[s_first[p]]] >= 0 &&
          node_sep[order[s_first[c]]] != node_sep[order[s_first[p]]])
        continue;   // wide sibling separators stay fronts of their own (see the supernode pass above)
      int64_t z = npiv[c] * (npiv[p] + nrow[p] - nrow[c]);
      if (z < 0) z = 0;
      int64_t ztot = zeros[c] + zeros[p] + z;
      double T = 0.5 * (double)np2 * (double)(np2 + 1) + (double)np2 * (double)nrow[p];
      double frac = (double)ztot / T;
      bool fits_before = lds_elems((int)npiv[p], (int)nrow[p]) <= lds_budget;
      bool fits_after = lds_elems((int)np2, (int)nrow[p]) <= lds_budget;
      if (fits_before && !fits_after) continue;
      // Per-front fixed cost on the GPU (zero / assemble / extend-add / store,
      // ~3 us) dwarfs the flops of padding zeros in a small front, so small
      // fronts merge eagerly; large ones only when it is (nearly) free.
      const int64_t m2 = np2 + nrow[p];
      bool ok = z == 0 || np2 <= 16 || (m2 <= 64 && frac <= 0.70) || (m2 <= 96 && np2 <= 64 && frac <= 0.50) ||
                (np2 <= opt.amalg_np && frac <= opt.amalg_frac) || frac <= 0.03;
      if (!ok) continue;
      merged_into[c] = p;
      npiv[p] = np2;
      zeros[p] = ztot;
      auto &dst = pre[p];
      dst.insert(dst.end(), pre[c].begin(), pre[c].end());
      dst.push_back(c);
      std::vector<int32_t>().swap(pre[c]);
    }
  }
  // merged tree + final order (DFS postorder over representatives)
  auto rep_of = [&](int s) {
    while (merged_into[s] >= 0) s = merged_into[s];
    return s;
  };
  std::vector<int32_t> rparent(S0, -1);
  for (int s = 0; s < S0; s++)
    if (merged_into[s] < 0 && sp[s] >= 0) rparent[s] = rep_of(sp[s]);
  std::vector<int32_t> final_sn;  // representatives in final elimination order
  {
    std::vector<int32_t> pp(S0, -1), post;
    // restrict to representatives by mapping non-representatives to self-roots, then filter
    for (int s = 0; s < S0; s++) pp[s] = merged_into[s] < 0 ? rparent[s] : -2;
    std::vector<int32_t> idx_of(S0, -1), reps;
    for (int s = 0; s < S0; s++)
      if (pp[s] != -2) { idx_of[s] = (int)reps.size(); reps.push_back(s); }
    std::vector<int32_t> cpar(reps.size());
    for (size_t i = 0; i < reps.size(); i++) cpar[i] = pp[reps[i]] < 0 ? -1 : idx_of[pp[reps[i]]];
    postorder_forest(cpar, post);
    for (int i : post) final_sn.push_back(reps[i]);
  }
  int S = (int)final_sn.size();
  std::vector<int32_t> forder;
  forder.reserve(N);
  sym.sn_first_pos.resize(S);
  sym.sn_npos.resize(S);
  for (int f = 0; f < S; f++) {
    int r = final_sn[f];
    sym.sn_first_pos[f] = (int)forder.size();
    auto emit = [&](int s) {
      for (int j = s_first[s]; j <= s_last[s]; j++) forder.push_back(order[j]);
    };
    for (int m : pre[r]) emit(m);
    emit(r);
    sym.sn_npos[f] = (int)forder.size() - sym.sn_first_pos[f];
  }
  if ((int)forder.size() != N) return "internal: amalgamation lost nodes";
  // A supernode whose front does not fit the LDS budget is cut into a chain of
  // narrower supernodes when that makes every piece fit (same flops, the later
  // pivots simply travel through the first pieces' update matrices) -- into at most
  // opt.max_lds_pieces of them: every piece moves the whole update matrix through LDS.
  {
    std::vector<int32_t> nfirst, nnpos;
    for (int f = 0; f < S; f++) {
      const int a = sym.sn_first_pos[f], n = sym.sn_npos[f];
      const int64_t below = nrow[final_sn[f]];
      int64_t tot = 0;
      for (int p = a; p < a + n; p++) tot += w[forder[p]];
      int pieces = 1;
      if (lds_elems((int)tot, (int)below) > lds_budget && n > 1) {
        for (int np = 2; np <= n && np <= opt.max_lds_pieces && pieces == 1; np++) {
          // np pieces of (almost) equal node counts; check that all fit
          bool ok = true;
          int64_t done = 0;
          int pos = a;
          for (int q = 0; q < np && ok; q++) {
            int cnt = n / np + (q < n % np ? 1 : 0);
            int64_t nc = 0;
            for (int p = pos; p < pos + cnt; p++) nc += w[forder[p]];
            ok = lds_elems((int)nc, (int)(tot - done - nc + below)) <= lds_budget;
            done += nc;
            pos += cnt;
          }
          if (ok) pieces = np;
        }
      }
      int pos = a;
      for (int q = 0; q < pieces; q++) {
        int cnt = n / pieces + (q < n % pieces ? 1 : 0);
        nfirst.push_back(pos);
        nnpos.push_back(cnt);
        pos += cnt;
      }
    }
    sym.sn_first_pos.swap(nfirst);
    sym.sn_npos.swap(nnpos);
  }
  S = (int)sym.sn_first_pos.size();
  sym.S = S;
  order.swap(forder);
  for (int p = 0; p < N; p++) pos_of[order[p]] = p;

  // ---- 5. supernodal symbolic factorisation on the final partition ----------
  std::vector<int32_t> sn_at_pos(N);
  std::vector<std::vector<int32_t>> rows(S);  // node positions
  auto symbolic_rows = [&] {
    for (int f = 0; f < S; f++)
      for (int p = sym.sn_first_pos[f]; p < sym.sn_first_pos[f] + sym.sn_npos[f]; p++) sn_at_pos[p] = f;
    for (auto &r : rows) r.clear();
    sym.sn_parent.assign(S, -1);
    std::vector<int32_t> mark(N, -1), head(S, -1), next(S, -1);
    for (int f = 0; f < S; f++) {
      const int a = sym.sn_first_pos[f], b = a + sym.sn_npos[f];
      auto &r = rows[f];
      for (int p = a; p < b; p++) {
        int v = order[p];
        for (int q = adj.ptr[v]; q < adj.ptr[v + 1]; q++) {
          int i = pos_of[adj.idx[q]];
          if (i >= b && mark[i] != f) { mark[i] = f; r.push_back(i); }
        }
      }
      for (int c = head[f]; c != -1; c = next[c])
        for (int x : rows[c])
          if (x >= b && mark[x] != f) { mark[x] = f; r.push_back(x); }
      std::sort(r.begin(), r.end());
      if (!r.empty()) {
        int pf = sn_at_pos[r[0]];
        sym.sn_parent[f] = pf;
        next[f] = head[pf];
        head[pf] = f;
      }
    }
  };
  symbolic_rows();
  // The kernels walk a front's pivot columns in blocks of 16, and a block costs the same whether it holds 16 columns or one
  // (~3 us in the factorisation, 0.4 us in the back substitution, on the critical path).  A front whose width is a few
  // columns over a multiple of 16 hands its LAST nodes to its parent (they become the parent's first pivots: the same
  // elimination tree, the same fill -- a front's rows are a subset of its parent's front -- so only the cut between two
  // supernodes of a chain moves); the parent may pass a remainder on in turn, like a carry.  Fronts in LDS only.
  if ((opt.balance_blocks || opt.merge_chain_nc > 0) && opt.n_parts <= 1) {
    std::vector<int64_t> ncs, nrs;
    std::vector<int32_t> crit;
    auto measure = [&] {
      ncs.assign(S, 0); nrs.assign(S, 0);
      for (int f = 0; f < S; f++) {
        for (int p = sym.sn_first_pos[f]; p < sym.sn_first_pos[f] + sym.sn_npos[f]; p++) ncs[f] += w[order[p]];
        for (int x : rows[f]) nrs[f] += w[order[x]];
      }
      // a parent's child on the longest chain of blocks: only that one hands nodes up (what the other children would add to
      // the parent would lengthen the chain, and their own partial blocks are off it)
      std::vector<int64_t> chain(S, 0);
      crit.assign(S, -1);
      for (int f = 0; f < S; f++) {
        chain[f] += (ncs[f] + 15) / 16;
        const int pf = sym.sn_parent[f];
        if (pf >= 0 && (crit[pf] < 0 || chain[f] > chain[crit[pf]])) crit[pf] = f;
        if (pf >= 0) chain[pf] = std::max(chain[pf], chain[f]);   // (pf's own blocks are added when its turn comes)
      }
    };
    // out_nodes[f] trailing nodes of front f become the first nodes of its parent; a front left empty disappears
    auto apply_moves = [&](const std::vector<int32_t> &out_nodes) -> bool {
      std::vector<std::vector<int32_t>> incoming(S);
      for (int f = 0; f < S; f++)
        if (out_nodes[f] > 0) {
          const int b = sym.sn_first_pos[f] + sym.sn_npos[f];
          for (int p = b - out_nodes[f]; p < b; p++) incoming[sym.sn_parent[f]].push_back(order[p]);
        }
      std::vector<int32_t> norder, nfirst, nnpos;
      norder.reserve(N);
      for (int f = 0; f < S; f++) {
        const int a = (int)norder.size();
        for (int v : incoming[f]) norder.push_back(v);
        for (int p = sym.sn_first_pos[f]; p < sym.sn_first_pos[f] + sym.sn_npos[f] - out_nodes[f]; p++) norder.push_back(order[p]);
        if ((int)norder.size() > a) { nfirst.push_back(a); nnpos.push_back((int)norder.size() - a); }
      }
      if ((int)norder.size() != N) return false;
      order.swap(norder);
      sym.sn_first_pos.swap(nfirst);
      sym.sn_npos.swap(nnpos);
      S = (int)sym.sn_first_pos.size();
      sym.S = S;
      rows.assign(S, {});
      for (int p = 0; p < N; p++) pos_of[order[p]] = p;
      symbolic_rows();
      return true;
    };
    const auto is_lds = [&](int64_t nc, int64_t nr) { return lds_elems((int)nc, (int)nr) <= lds_budget; };
    if (opt.merge_chain_nc > 0) {
      // A narrow front on the chain joins its parent outright when the cost model says the parent finishes earlier that way:
      // a front on the critical path costs ~4 us of hand-off, Schur complement and copy-out in the factorisation (and ~1.2 us
      // in the back substitution) whatever its width -- but its panel then has to wait for the parent's OTHER children.  With
      // fin = the model's finish time of a front's subtree, f = the parent's child that finishes last, c2 = the finish of the
      // runner-up:   before  fin(f) + cost(p)      merged  max(fin(f) - cost(f), c2) + cost(f + p).
      // (The relaxed amalgamation above weighs zeros against a front's fixed cost everywhere; here only the one child whose
      // chain the parent waits for.)  A front takes one child per pass; a few passes.
      for (int pass = 0; pass < 3; pass++) {
        measure();
        std::vector<int32_t> kids(S, 0), out_nodes(S, 0);
        for (int f = 0; f < S; f++)
          if (sym.sn_parent[f] >= 0) kids[sym.sn_parent[f]]++;
        std::vector<double> fin(S, 0.0), c1(S, 0.0), c2(S, 0.0);
        std::vector<int32_t> last(S, -1);
        std::vector<char> grown(S, 0);
        bool any = false;
        for (int p = 0; p < S; p++) {   // children before parents: c1 / c2 / last of p are complete here
          double cost_p = front_cost_us((int)ncs[p], (int)nrs[p], kids[p]);
          double start = c1[p];
          const int f = last[p];
          if (f >= 0 && !grown[f] && ncs[f] <= opt.merge_chain_nc && is_lds(ncs[f], nrs[f]) && is_lds(ncs[p], nrs[p]) &&
              is_lds(ncs[f] + ncs[p], nrs[p])) {
            const int64_t np2 = ncs[f] + ncs[p];
            const double z = (double)ncs[f] * (double)std::max<int64_t>(ncs[p] + nrs[p] - nrs[f], 0);
            const double T = 0.5 * (double)np2 * (double)(np2 + 1) + (double)np2 * (double)nrs[p];
            const double cost_f = front_cost_us((int)ncs[f], (int)nrs[f], kids[f]);
            const double cost_m = front_cost_us((int)np2, (int)nrs[p], kids[p] - 1 + kids[f]);
            const double merged_fin = std::max(fin[f] - cost_f, c2[p]) + cost_m;
            if (z / T <= opt.merge_chain_frac && fin[f] + cost_p - merged_fin > opt.merge_chain_gain_us) {
              out_nodes[f] = sym.sn_npos[f];
              ncs[p] = np2;
              kids[p] += kids[f] - 1;
              grown[p] = 1;
              any = true;
              start = std::max(fin[f] - cost_f, c2[p]);
              cost_p = cost_m;
            }
          }
          fin[p] = start + cost_p;
          const int pp = sym.sn_parent[p];
          if (pp >= 0) {
            if (fin[p] > c1[pp]) { c2[pp] = c1[pp]; c1[pp] = fin[p]; last[pp] = p; }
            else if (fin[p] > c2[pp]) c2[pp] = fin[p];
          }
        }
        if (!any) break;
        if (!apply_moves(out_nodes)) return "internal: chain merging lost nodes";
      }
    }
    if (opt.balance_blocks) {
      measure();
      std::vector<int32_t> out_nodes(S, 0);   // trailing nodes a front hands to its parent
      bool any = false;
      for (int f = 0; f < S; f++) {
        const int pf = sym.sn_parent[f];
        if (pf < 0 || crit[pf] != f) continue;
        const int64_t r = ncs[f] % 16;
        if (r == 0 || r > opt.balance_max_rem || ncs[f] <= 16) continue;
        if (!is_lds(ncs[f], nrs[f]) || !is_lds(ncs[pf], nrs[pf])) continue;
        int k = 0;
        int64_t m = 0;
        for (int p = sym.sn_first_pos[f] + sym.sn_npos[f] - 1; p > sym.sn_first_pos[f] && m < r; p--) { m += w[order[p]]; k++; }
        if (m < r) continue;
        if (!is_lds(ncs[f] - m, nrs[f] + m) || !is_lds(ncs[pf] + m, nrs[pf])) continue;
        const bool parent_grows = (ncs[pf] + m + 15) / 16 > (ncs[pf] + 15) / 16;
        if (parent_grows && sym.sn_parent[pf] < 0) continue;   // a root has nobody to pass the remainder on to
        out_nodes[f] = k;
        ncs[f] -= m; nrs[f] += m; ncs[pf] += m;
        any = true;
      }
      if (any && !apply_moves(out_nodes)) return "internal: block balancing lost nodes";
    }
  }
  sym.order = order;
  sym.pos_of = pos_of;

  // permuted scalar numbering
  sym.node_pcol.assign(N, 0);
  sym.perm.resize(g.dim);
  {
    int c = 0;
    for (int p = 0; p < N; p++) {
      int v = order[p];
      sym.node_pcol[v] = c;
      for (int d = 0; d < w[v]; d++) sym.perm[c++] = g.node_offset[v] + d;
    }
  }

  ptimer.mark("supernodes");
  sym.sn_ncols.resize(S);
  sym.sn_nrows.resize(S);
  sym.sn_col0.resize(S);
  sym.sn_owner.resize(S);
  sym.sn_rows_ptr.assign(S + 1, 0);
  for (int f = 0; f < S; f++) {
    int nc = 0, nr = 0;
    for (int p = sym.sn_first_pos[f]; p < sym.sn_first_pos[f] + sym.sn_npos[f]; p++) nc += w[order[p]];
    for (int x : rows[f]) nr += w[order[x]];
    sym.sn_ncols[f] = nc;
    sym.sn_nrows[f] = nr;
    sym.sn_col0[f] = sym.node_pcol[order[sym.sn_first_pos[f]]];
    sym.sn_owner[f] = opt.n_parts > 1 ? sym.node_part[order[sym.sn_first_pos[f]]] : 0;
    sym.sn_rows_ptr[f + 1] = sym.sn_rows_ptr[f] + nr;
  }
  sym.sn_rows.resize(sym.sn_rows_ptr[S]);
  for (int f = 0; f < S; f++) {
    int64_t o = sym.sn_rows_ptr[f];
    for (int x : rows[f]) {
      int v = order[x];
      for (int d = 0; d < w[v]; d++) sym.sn_rows[o++] = sym.node_pcol[v] + d;
    }
  }

  ptimer.mark("symbolic");
  // ---- 6. storage layout, big-front classification --------------------------
  sym.sn_loff.resize(S);
  sym.sn_uoff.resize(S);
  sym.sn_uld.resize(S);
  sym.sn_big.resize(S);
  sym.sn_huge.assign(S, 0);
  {
    int64_t lo = 0, uo = 0;
    for (int f = 0; f < S; f++) {
      const int nc = sym.sn_ncols[f], nr = sym.sn_nrows[f], M = nc + nr + 1;
      const bool big = lds_elems(nc, nr) > lds_budget;
      sym.sn_big[f] = big;
      sym.sn_huge[f] = big && (int64_t)M * nc > opt.panel_budget_elems;
      sym.n_big += big;
      sym.max_front = std::max(sym.max_front, nc + nr);
      sym.max_pivot_cols = std::max(sym.max_pivot_cols, nc);
      for (int k = 0; k < nc; k++) sym.factor_flops += (int64_t)(M - 1 - k) * (M - 1 - k) + 2 * (M - k);
      if (!big) {
        sym.sn_loff[f] = lo;
        lo += (int64_t)M * nc;
        sym.sn_uld[f] = 0;
        sym.sn_uoff[f] = uo;
        uo += (int64_t)(nr + 1) * (nr + 2) / 2;
      } else {
        // the whole M x M front lives in L storage; the update matrix is its
        // trailing (nr+1) square, addressed in place with ld = M
        sym.sn_loff[f] = lo;
        sym.sn_uld[f] = M;
        sym.sn_uoff[f] = lo + (int64_t)nc * M + nc;  // element (nc, nc)
        lo += (int64_t)M * M;
      }
      // keep 16-byte alignment for either precision
      lo = (lo + 3) & ~(int64_t)3;
      uo = (uo + 3) & ~(int64_t)3;
    }
    sym.l_elems = lo;
    sym.u_elems = uo;
  }

  // ---- 6b. sharding: boundary fronts and column owners
  sym.sn_xch_off.assign(S, -1);
  sym.col_owner.assign(g.dim, 0);
  sym.xch_chunk = 0;
  if (opt.n_parts > 1) {
    // The boundary fronts' update matrices are exchanged by an ALL-GATHER: rank r writes chunk r of the
    // buffer (its own boundary fronts, packed), so offsets are (owner * chunk + offset inside the chunk)
    // with one chunk size for all ranks (the largest total, rounded to 64 scalars).
    std::vector<int64_t> used(opt.n_parts, 0);
    for (int f = 0; f < S; f++) {
      const int p = sym.sn_parent[f];
      if (sym.sn_owner[f] >= 0 && p >= 0 && sym.sn_owner[p] < 0) {
        int64_t &xo = used[sym.sn_owner[f]];
        xo += (int64_t)(sym.sn_nrows[f] + 1) * (sym.sn_nrows[f] + 2) / 2;
        xo = (xo + 3) & ~(int64_t)3;
      }
    }
    for (int64_t u : used) sym.xch_chunk = std::max(sym.xch_chunk, u);
    // behind the update matrices, every rank's chunk carries its PARTIAL diagonal blocks and right-hand-side
    // entries of the shared nodes (summed over the edges the rank owns): the "Hessian border blocks"
    sym.xch_shared_off = (sym.xch_chunk + 63) & ~(int64_t)63;
    int64_t shared_elems = 0;
    for (int i = 0; i < N; i++)
      if (sym.node_part[i] < 0) shared_elems += (int64_t)w[i] * w[i] + w[i];
    // ... and, last, ONE scalar: the rank's device error flag after stage 0 (a non-positive pivot in a rank's own
    // subtree must stop the update on EVERY rank: the state stays untouched across the whole group)
    sym.xch_flag_off = sym.xch_shared_off + shared_elems;
    sym.xch_chunk = std::max<int64_t>((sym.xch_flag_off + 1 + 63) & ~(int64_t)63, 64);
    std::fill(used.begin(), used.end(), 0);
    for (int f = 0; f < S; f++) {
      for (int c = 0; c < sym.sn_ncols[f]; c++) sym.col_owner[sym.sn_col0[f] + c] = (int8_t)sym.sn_owner[f];
      const int p = sym.sn_parent[f];
      if (sym.sn_owner[f] >= 0 && p >= 0 && sym.sn_owner[p] < 0) {
        int64_t &xo = used[sym.sn_owner[f]];
        sym.sn_xch_off[f] = (int64_t)sym.sn_owner[f] * sym.xch_chunk + xo;
        xo += (int64_t)(sym.sn_nrows[f] + 1) * (sym.sn_nrows[f] + 2) / 2;
        xo = (xo + 3) & ~(int64_t)3;
      }
      if (sym.sn_owner[f] >= 0 && p >= 0 && sym.sn_owner[p] >= 0 && sym.sn_owner[p] != sym.sn_owner[f])
        return "internal: a front's parent belongs to another rank";
      if (sym.sn_owner[f] < 0 && p >= 0 && sym.sn_owner[p] >= 0) return "internal: shared front below an owned one";
    }
    sym.xch_elems = (int64_t)opt.n_parts * sym.xch_chunk;
  }

  ptimer.mark("layout");
  // ---- 7. H block structure + assembly lists --------------------------------
  sym.diag_off.resize(N);
  int64_t hv = 0;
  for (int p = 0; p < N; p++) {
    int v = order[p];
    sym.diag_off[v] = hv;
    hv += (int64_t)w[v] * w[v];
  }
  {
    // unique (col_pos, row_pos) pairs
    std::vector<std::pair<int64_t, int32_t>> keyed(E);
    for (int k = 0; k < E; k++) {
      int pa = pos_of[g.edge_from[k]], pb = pos_of[g.edge_to[k]];
      int c = std::min(pa, pb), r = std::max(pa, pb);
      keyed[k] = {(int64_t)c * N + r, k};
    }
    std::sort(keyed.begin(), keyed.end());
    sym.edge_slot.resize(E);
    sym.edge_transposed.resize(E);
    int64_t last = -1;
    for (auto &[key, k] : keyed) {
      {
        // every edge owns one block; a repeated node pair (parallel edges) gets
        // an extra block flagged `dup`, summed in a serial pass at assembly
        int c = (int)(key / N), r = (int)(key % N);
        sym.blk_col.push_back(order[c]);
        sym.blk_row.push_back(order[r]);
        sym.blk_off.push_back(hv);
        sym.slot_shared.push_back(key == last ? 1 : 0);
        hv += (int64_t)w[order[c]] * w[order[r]];
        last = key;
      }
      sym.edge_slot[k] = (int)sym.blk_col.size() - 1;
      // slot holds H[row node rows, col node cols]; the edge computes H[from rows, to cols]
      sym.edge_transposed[k] = sym.blk_row.back() == g.edge_from[k] ? 0 : 1;
    }
    sym.n_offblocks = (int64_t)sym.blk_col.size();
    sym.n_hvals = hv;
  }
  // incidence lists
  sym.inc_ptr.assign(N + 1, 0);
  for (int k = 0; k < E; k++) {
    sym.inc_ptr[g.edge_from[k] + 1]++;
    sym.inc_ptr[g.edge_to[k] + 1]++;
  }
  for (int i = 0; i < N; i++) sym.inc_ptr[i + 1] += sym.inc_ptr[i];
  sym.inc_list.resize(sym.inc_ptr[N]);
  {
    std::vector<int32_t> fill(sym.inc_ptr.begin(), sym.inc_ptr.end() - 1);
    for (int k = 0; k < E; k++) {
      sym.inc_list[fill[g.edge_from[k]]++] = k * 2;
      sym.inc_list[fill[g.edge_to[k]]++] = k * 2 + 1;
    }
  }
  // local index lookup per front, assembly items, children, rel maps
  sym.asm_ptr.assign(S + 1, 0);
  sym.child_ptr.assign(S + 1, 0);
  sym.rel_ptr.assign(S + 1, 0);
  for (int f = 0; f < S; f++)
    if (sym.sn_parent[f] >= 0) sym.child_ptr[sym.sn_parent[f] + 1]++;
  for (int f = 0; f < S; f++) {
    sym.child_ptr[f + 1] += sym.child_ptr[f];
    sym.rel_ptr[f + 1] = sym.rel_ptr[f] + (sym.sn_parent[f] >= 0 ? sym.sn_nrows[f] + 1 : 0);
  }
  sym.child_list.resize(sym.child_ptr[S]);
  sym.rel.resize(sym.rel_ptr[S]);
  {
    std::vector<int32_t> fill(sym.child_ptr.begin(), sym.child_ptr.end() - 1);
    for (int f = 0; f < S; f++)
      if (sym.sn_parent[f] >= 0) sym.child_list[fill[sym.sn_parent[f]]++] = f;
  }
  if (sym.n_big == 0 && opt.n_parts <= 1) {
    // Graphs whose fronts all live in LDS (the dataflow launches of lds_flow.hip.h): a front adds its children's update
    // matrices in the order in which the cost model expects them to be FINISHED (the subtree with the longest chain
    // last), so that a parent which waits for its children one by one waits for the last one only.  The order is a
    // function of the tree alone -- the same whatever task partition or launch form runs it: sums, hence bits, agree.
    std::vector<double> fin(S, 0.0);
    for (int f = 0; f < S; f++) {
      double latest = 0.0;
      for (int q = sym.child_ptr[f]; q < sym.child_ptr[f + 1]; q++) latest = std::max(latest, fin[sym.child_list[q]]);
      fin[f] = latest + front_cost_us(sym.sn_ncols[f], sym.sn_nrows[f], sym.child_ptr[f + 1] - sym.child_ptr[f]);
      std::stable_sort(sym.child_list.begin() + sym.child_ptr[f], sym.child_list.begin() + sym.child_ptr[f + 1],
                       [&](int32_t a, int32_t b) { return fin[a] < fin[b]; });
    }
  }
  {
    // blocks are sorted by column position, so a supernode's blocks are contiguous
    std::vector<int64_t> blk_begin(S + 1, 0);
    for (int64_t s = 0; s < sym.n_offblocks; s++) blk_begin[sn_at_pos[pos_of[sym.blk_col[s]]] + 1]++;
    for (int f = 0; f < S; f++) blk_begin[f + 1] += blk_begin[f];
    std::vector<int32_t> loc(g.dim, -1);
    for (int f = 0; f < S; f++) {
      const int nc = sym.sn_ncols[f], nr = sym.sn_nrows[f], c0 = sym.sn_col0[f];
      const int32_t *rws = sym.sn_rows.data() + sym.sn_rows_ptr[f];
      for (int i = 0; i < nc; i++) loc[c0 + i] = i;
      for (int i = 0; i < nr; i++) loc[rws[i]] = nc + i;
      for (int p = sym.sn_first_pos[f]; p < sym.sn_first_pos[f] + sym.sn_npos[f]; p++) {
        int v = order[p];
        AsmItem it;
        it.src = sym.diag_off[v];
        it.lrow = it.lcol = sym.node_pcol[v] - c0;
        it.drow = it.dcol = (int16_t)w[v];
        it.diag = 1;
        sym.asm_items.push_back(it);
      }
      for (int64_t s = blk_begin[f]; s < blk_begin[f + 1]; s++) {
        int rn = sym.blk_row[s], cn = sym.blk_col[s];
        AsmItem it;
        it.src = sym.blk_off[s];
        it.lcol = sym.node_pcol[cn] - c0;
        it.lrow = loc[sym.node_pcol[rn]];
        if (it.lrow < 0) return "internal: H block outside its front";
        it.drow = (int16_t)w[rn];
        it.dcol = (int16_t)w[cn];
        it.diag = sym.slot_shared[s] ? 2 : 0;
        sym.asm_items.push_back(it);
      }
      sym.asm_ptr[f + 1] = (int64_t)sym.asm_items.size();
      for (int q = sym.child_ptr[f]; q < sym.child_ptr[f + 1]; q++) {
        int c = sym.child_list[q];
        const int32_t *crows = sym.sn_rows.data() + sym.sn_rows_ptr[c];
        int32_t *rel = sym.rel.data() + sym.rel_ptr[c];
        for (int i = 0; i < sym.sn_nrows[c]; i++) {
          rel[i] = loc[crows[i]];
          if (rel[i] < 0) return "internal: child row missing from parent front";
        }
        rel[sym.sn_nrows[c]] = nc + nr;  // rhs row
      }
      for (int i = 0; i < nc; i++) loc[c0 + i] = -1;
      for (int i = 0; i < nr; i++) loc[rws[i]] = -1;
    }
  }
  // flat lists + scatter maps
  sym.fasm_ptr.assign(S + 1, 0);
  sym.fdup_ptr.assign(S + 1, 0);
  sym.scat_ptr.assign(S, -1);
  for (int f = 0; f < S; f++) {
    const int M = sym.sn_ncols[f] + sym.sn_nrows[f] + 1;
    for (int64_t q = sym.asm_ptr[f]; q < sym.asm_ptr[f + 1]; q++) {
      const AsmItem &it = sym.asm_items[q];
      for (int i = 0; i < it.drow; i++)
        for (int j = 0; j < it.dcol; j++) {
          if (it.diag == 1 && i < j) continue;
          int64_t src = it.src + i * it.dcol + j;
          int64_t dst = (int64_t)(it.lcol + j) * M + it.lrow + i;
          if (src > 0x7fffffffLL || dst > 0x7fffffffLL) return "graph too large for 32-bit assembly indices";
          if (it.diag == 2) { sym.fdup_src.push_back((int32_t)src); sym.fdup_dst.push_back((int32_t)dst); }
          else { sym.fasm_src.push_back((int32_t)src); sym.fasm_dst.push_back((int32_t)dst); }
        }
    }
    sym.fasm_ptr[f + 1] = (int64_t)sym.fasm_src.size();
    sym.fdup_ptr[f + 1] = (int64_t)sym.fdup_src.size();
  }
  // fronts beyond LDS: their entries in column-major order of the destination, and where every pivot column's entries
  // start -- the wave of k_big_build that writes a column adds the column's H entries itself (no k_big_assemble launch)
  sym.fasm_colptr.assign((size_t)g.dim + 1, 0);
  for (int f = 0; f < S; f++) {
    if (!sym.sn_big[f]) continue;
    const int64_t a = sym.fasm_ptr[f], b = sym.fasm_ptr[f + 1];
    const int nc = sym.sn_ncols[f], M = nc + sym.sn_nrows[f] + 1, c0 = sym.sn_col0[f];
    std::vector<std::pair<int32_t, int32_t>> e((size_t)(b - a));
    for (int64_t t = a; t < b; t++) e[(size_t)(t - a)] = {sym.fasm_dst[t], sym.fasm_src[t]};
    std::sort(e.begin(), e.end());
    int64_t t = a;
    for (int J = 0; J < nc; J++) {
      sym.fasm_colptr[c0 + J] = (int32_t)t;
      while (t < b && e[(size_t)(t - a)].first < (int64_t)(J + 1) * M) t++;
    }
    if (t != b) return "internal: assembly entry outside the pivot columns";
    sym.fasm_colptr[c0 + nc] = (int32_t)b;   // (the next front overwrites it with the same value or its own start)
    for (int64_t q = a; q < b; q++) { sym.fasm_dst[q] = e[(size_t)(q - a)].first; sym.fasm_src[q] = e[(size_t)(q - a)].second; }
  }
  {
    // one destination per element of a child's packed update matrix (the LDS image of the parent): sized
    // once, filled column by column with the column's part of the index hoisted
    int64_t total = 0;
    for (int c = 0; c < S; c++) {
      const int p = sym.sn_parent[c];
      if (p < 0 || sym.sn_big[c] || sym.sn_big[p]) continue;
      const int64_t ncu = sym.sn_nrows[c] + 1;
      sym.scat_ptr[c] = total;
      total += ncu * (ncu + 1) / 2;
    }
    sym.scat.resize(total);
    for (int c = 0; c < S; c++) {
      const int p = sym.sn_parent[c];
      if (p < 0 || sym.sn_big[c] || sym.sn_big[p]) continue;
      const int ncu = sym.sn_nrows[c] + 1;
      const int ncp = sym.sn_ncols[p], Mp = ncp + sym.sn_nrows[p] + 1, nup = sym.sn_nrows[p] + 1;
      const int32_t *rel = sym.rel.data() + sym.rel_ptr[c];
      int32_t *out = sym.scat.data() + sym.scat_ptr[c];
      for (int j = 0; j < ncu; j++) {
        const int lj = rel[j];
        // d = lj * Mp + li  (pivot column of the parent)  |  Mp * ncp + b2 * nup - b2 (b2 - 1) / 2 + (li - ncp - b2)
        const int b2 = lj - ncp;
        const int base = lj < ncp ? lj * Mp : Mp * ncp + (b2 * nup - b2 * (b2 - 1) / 2) - ncp - b2;
        for (int i = j; i < ncu; i++) *out++ = base + rel[i];
      }
      out[-1] = -1;   // the (rhs, rhs) corner is never used
    }
    // children of a front beyond LDS: the INVERSE of rel (parent local index -> child local index, -1 where the
    // child has no such row), so that k_big_build can gather every entry of the parent from its children and write
    // it exactly once -- no zeroing pass, no read-modify-write per child.  rel is strictly increasing (both row
    // lists are in elimination order), so the inverse is monotone too and a column's gather is a near-contiguous read.
    int64_t inv_base = total;
    for (int c = 0; c < S; c++) {
      const int p = sym.sn_parent[c];
      if (p < 0 || !sym.sn_big[p]) continue;
      sym.scat_ptr[c] = total;
      total += sym.sn_ncols[p] + sym.sn_nrows[p] + 1;
    }
    if (total > inv_base) {
      sym.scat.resize(total, -1);
      for (int c = 0; c < S; c++) {
        const int p = sym.sn_parent[c];
        if (p < 0 || !sym.sn_big[p]) continue;
        const int ncu = sym.sn_nrows[c] + 1, Mp = sym.sn_ncols[p] + sym.sn_nrows[p] + 1;
        const int32_t *rel = sym.rel.data() + sym.rel_ptr[c];
        int32_t *inv = sym.scat.data() + sym.scat_ptr[c];
        for (int r = 0; r < Mp; r++) inv[r] = -1;
        for (int i = 0; i < ncu; i++) {
          if (i > 0 && rel[i] <= rel[i - 1]) return "internal: child rows not in the parent's order";
          inv[rel[i]] = i;
        }
      }
    }
  }

  ptimer.mark("asm lists");
  // ---- 8. schedule ----------------------------------------------------------
  // The task granularity trades launches (levels) against the serialisation of independent sibling
  // fronts inside one workgroup; the best threshold depends on the tree.  Candidates are scored with
  // the calibrated cost model (sum over steps of launch gap + slowest task) and the best one is kept.
  auto build_schedule = [&](double task_us) {
    sym.task_ptr.clear();
    sym.task_sn.clear();
    sym.steps.clear();
    std::vector<double> cost(S), scost(S), sub(S, 0.0);
    for (int f = 0; f < S; f++) {
      cost[f] = front_cost_us(sym.sn_ncols[f], sym.sn_nrows[f], sym.child_ptr[f + 1] - sym.child_ptr[f]);
      // back substitution of the same front (the solve launches reuse the task partition): staging +
      // product with the rows below + one chain step per 16 columns
      scost[f] = 1.6 + 1.3 * (double)((sym.sn_ncols[f] + 15) / 16);
      sub[f] += cost[f];
      if (sym.sn_parent[f] >= 0) sub[sym.sn_parent[f]] += sub[f];
    }
    std::vector<char> top(S, 0);
    for (int f = 0; f < S; f++) {
      if (sym.sn_big[f] || sub[f] > task_us || (opt.n_parts > 1 && sym.sn_owner[f] < 0)) top[f] = 1;
      if (top[f] && sym.sn_parent[f] >= 0) top[sym.sn_parent[f]] = 1;  // parents come later in order
    }
    // note: a parent has a larger index than all of its descendants, so one
    // ascending pass propagates `top` all the way up.
    std::vector<int32_t> task_of(S, -1), lvl(S, 0);
    std::vector<std::vector<int32_t>> tasks;
    std::vector<int32_t> task_level;
    // leaf tasks: maximal non-top subtrees
    for (int f = 0; f < S; f++) {
      if (top[f]) continue;
      int p = sym.sn_parent[f];
      if (p >= 0 && !top[p]) continue;  // not a subtree root
      task_of[f] = (int)tasks.size();
      tasks.emplace_back();
      task_level.push_back(0);
    }
    for (int f = S - 1; f >= 0; f--)  // push task ids down to descendants
      if (!top[f] && task_of[f] < 0) task_of[f] = task_of[sym.sn_parent[f]];
    for (int f = 0; f < S; f++)
      if (!top[f]) tasks[task_of[f]].push_back(f);
    // top part: levels with chain merging
    for (int f = 0; f < S; f++) {
      if (!top[f]) continue;
      int L = 0, nmax = 0, cmax = -1;
      for (int q = sym.child_ptr[f]; q < sym.child_ptr[f + 1]; q++) {
        int c = sym.child_list[q];
        if (!top[c]) continue;
        // Sharded runs: the levels of the SHARED fronts must not depend on the rank.  Every rank factors them redundantly and
        // keeps its own copy of the shared poses, so the copies only stay equal if the ranks batch (and therefore sum) alike;
        // the depth of a rank's own subtrees under the task threshold IT picked is not the same on every rank.  (r05: rank 4
        // of 8 on sphere2500 batched the five top fronts 2 + 3 + 1 instead of 4 + 1 + 1, its copies of the shared poses drifted
        // from the others' by 1e-7 over six iterations and the converged state sat 4e-8 off the oracle's.)
        if (opt.n_parts > 1 && sym.sn_owner[f] < 0 && sym.sn_owner[c] >= 0) continue;
        if (lvl[c] > L) { L = lvl[c]; nmax = 1; cmax = c; }
        else if (lvl[c] == L) nmax++;
      }
      bool chain = L >= 1 && nmax == 1 && !sym.sn_big[f] && !sym.sn_big[cmax] &&
                   tasks[task_of[cmax]].back() == cmax && sym.sn_owner[f] == sym.sn_owner[cmax];
      if (chain) {
        lvl[f] = L;
        task_of[f] = task_of[cmax];
        tasks[task_of[f]].push_back(f);
      } else {
        lvl[f] = L + 1;
        if (!sym.sn_big[f]) {
          task_of[f] = (int)tasks.size();
          tasks.emplace_back(1, f);
          task_level.push_back(lvl[f]);
        }
      }
    }
    int maxlvl = 0;
    for (int f = 0; f < S; f++) maxlvl = std::max(maxlvl, (int)lvl[f]);
    // emit steps.  Sharded: two passes -- this rank's own fronts (all levels), then the shared ones.
    sym.task_ptr.assign(1, 0);
    double crit = 0.0;
    const int npass = opt.n_parts > 1 ? 2 : 1;
    for (int pass = 0; pass < npass; pass++) {
      auto wanted = [&](int f) {
        if (opt.n_parts <= 1) return true;
        return pass == 0 ? sym.sn_owner[f] == opt.my_part : sym.sn_owner[f] < 0;
      };
      for (int L = 0; L <= maxlvl; L++) {
        Step st{};
        st.kind = STEP_TASKS;
        st.task_begin = (int)sym.task_ptr.size() - 1;
        double worst = 0.0, worst_solve = 0.0, sum = 0.0, sum_solve = 0.0;
        // the level's tasks, the longest first: workgroups are dispatched in index order as CUs become free, so a level
        // of thousands of tasks packs like a longest-processing-time schedule and ends with its shortest tasks (r03, the
        // 2067 leaf tasks of the 1M-edge lattice in the order of the tree: 80 % of the CUs busy, the last 10 % of the
        // launch with a quarter of them)
        std::vector<std::pair<double, size_t>> level_tasks;
        for (size_t t = 0; t < tasks.size(); t++) {
          if (task_level[t] != L || tasks[t].empty() || !wanted(tasks[t][0])) continue;
          double tc = 0.0;
          for (int f : tasks[t]) tc += cost[f];
          level_tasks.emplace_back(-tc, t);
        }
        std::stable_sort(level_tasks.begin(), level_tasks.end());
        for (const auto &lt : level_tasks) {
          const size_t t = lt.second;
          double tc = 0.0, ts = 0.0;
          for (int f : tasks[t]) {
            sym.task_sn.push_back(f);
            tc += cost[f];
            ts += scost[f];
            st.max_front = std::max(st.max_front, sym.sn_ncols[f] + sym.sn_nrows[f] + 1);
            st.max_lds_elems = std::max<int64_t>(st.max_lds_elems, lds_elems(sym.sn_ncols[f], sym.sn_nrows[f]));
          }
          sym.task_ptr.push_back((int)sym.task_sn.size());
          worst = std::max(worst, tc);
          worst_solve = std::max(worst_solve, ts);
          sum += tc;
          sum_solve += ts;
        }
        st.task_end = (int)sym.task_ptr.size() - 1;
        {
          const int mf = st.max_front << opt.threads_shift;   // tuning knob: shift the size classes
          st.threads = mf <= 20 ? 64 : mf <= 48 ? 128 : mf <= 96 ? 256 : mf <= 128 ? 512 : 1024;
        }
        if (st.task_end > st.task_begin) {
          sym.steps.push_back(st);
          // one workgroup per CU: a level of many tasks takes its total work over the CUs, not its longest task
          crit += 1.5 + std::max(worst, sum / opt.n_cus) + 1.5 + std::max(worst_solve, sum_solve / opt.n_cus);
        }
        // fronts beyond LDS at this level: optional one-workgroup class, then the tiled batch
        for (int cls = 0; cls < 2; cls++) {
          Step b{};
          b.kind = cls == 0 ? STEP_MID : STEP_BIG;
          b.sn = -1;
          b.task_begin = (int)sym.task_ptr.size() - 1;
          double w2 = 0.0;
          for (int f = 0; f < S; f++)
            if (sym.sn_big[f] && (sym.sn_huge[f] ? 1 : 0) == cls && lvl[f] == L && wanted(f)) {
              sym.task_sn.push_back(f);
              sym.task_ptr.push_back((int)sym.task_sn.size());
              b.max_front = std::max(b.max_front, sym.sn_ncols[f] + sym.sn_nrows[f] + 1);
              if (cls == 0)   // the panel class keeps M x nc in LDS
                b.max_lds_elems = std::max<int64_t>(b.max_lds_elems, (int64_t)(sym.sn_ncols[f] + sym.sn_nrows[f] + 1) * (sym.sn_ncols[f] + 4) + 4);  // + four staged inverse maps (4-byte entries)
              w2 = std::max(w2, cost[f]);
            }
          b.task_end = (int)sym.task_ptr.size() - 1;
          b.threads = 1024;
          if (b.task_end > b.task_begin) {
            sym.steps.push_back(b);
            crit += 1.5 + w2;
          }
        }
      }
      if (pass == 0) sym.n_local_steps = opt.n_parts > 1 ? (int)sym.steps.size() : 0;
    }
    sym.est_critical_us = crit;
    return crit;
  };
  // ---- 8b. dataflow schedule (lds_flow.hip.h): ONE launch for the factorisation, ONE for the back substitution.
  // Tasks: every maximal subtree cheaper than the threshold (a workgroup walks it in postorder, as in the level
  // schedule), and every front above those a task of its own -- its workgroup does the front's child-independent
  // pre-work (zero the LDS image, H entries, rhs, scatter maps) while the children are still being factored elsewhere.
  // Tickets are handed out in the START order of a list schedule of the task tree on n_cus workgroups (priority:
  // longest remaining path), which is a topological order: a task only ever waits for tasks with smaller tickets.
  std::vector<double> fcost, fscost, fsub, fpre;   // per front, the same for every threshold
  std::vector<int32_t> big_level(S, -1);           // fronts beyond LDS: their level (step) in the level schedule
  auto build_flow_schedule = [&](double task_us, bool keep) {
    if (fcost.empty()) {
      fcost.resize(S); fscost.resize(S); fsub.assign(S, 0.0); fpre.resize(S);
      for (int f = 0; f < S; f++) {
        const int nc = sym.sn_ncols[f], nr = sym.sn_nrows[f];
        fcost[f] = front_cost_us(nc, nr, sym.child_ptr[f + 1] - sym.child_ptr[f]);
        fscost[f] = 1.6 + 1.3 * (double)((nc + 15) / 16);
        // what a front's workgroup can do before its last child is there: zeroing + assembly (the model's fixed part and
        // most of its per-element part) and the extend-add of the earlier children
        fpre[f] = std::min(0.6 * fcost[f], 1.6 + 2.4e-4 * (double)lds_elems(nc, nr) +
                                              0.65 * std::max(0, sym.child_ptr[f + 1] - sym.child_ptr[f] - 1));
        fsub[f] += fcost[f];
        if (sym.sn_parent[f] >= 0) fsub[sym.sn_parent[f]] += fsub[f];
      }
    }
    const std::vector<double> &cost = fcost, &scost = fscost, &sub = fsub, &pre = fpre;
    // fronts beyond LDS are not part of the launch: they sit above every LDS front (a front's parent is at least as
    // large), run level by level behind it, and their LDS children are the roots of the launch's forest
    std::vector<char> top(S, 0);
    for (int f = 0; f < S; f++) {
      if (sym.sn_big[f] || sub[f] > task_us) top[f] = 1;
      if (top[f] && sym.sn_parent[f] >= 0) top[sym.sn_parent[f]] = 1;
    }
    std::vector<int32_t> task_of(S, -1);
    std::vector<std::vector<int32_t>> tasks;
    for (int f = 0; f < S; f++) {
      if (sym.sn_big[f]) continue;
      if (top[f]) { task_of[f] = (int)tasks.size(); tasks.emplace_back(1, f); continue; }
      const int p = sym.sn_parent[f];
      if (p >= 0 && !top[p]) continue;   // inside a leaf subtree: its root collects it below
      task_of[f] = (int)tasks.size();
      tasks.emplace_back();
    }
    for (int f = S - 1; f >= 0; f--)
      if (!top[f] && task_of[f] < 0) task_of[f] = task_of[sym.sn_parent[f]];
    for (int f = 0; f < S; f++)
      if (!top[f]) tasks[task_of[f]].push_back(f);
    const int nt = (int)tasks.size();
    // the task forest: parent task = task of the parent of the task's last front (none: a root, or a front beyond LDS)
    std::vector<int> tparent(nt, -1), ndeps(nt, 0);
    std::vector<double> tpre(nt, 0.0), tdur(nt, 0.0), tsolve(nt, 0.0);
    for (int t = 0; t < nt; t++) {
      const int last = tasks[t].back(), p = sym.sn_parent[last];
      if (p >= 0 && !sym.sn_big[p]) { tparent[t] = task_of[p]; ndeps[task_of[p]]++; }
      double c = 0, sc = 0;
      for (int f : tasks[t]) { c += cost[f]; sc += scost[f]; }
      tpre[t] = top[tasks[t][0]] ? pre[tasks[t][0]] : 0.0;
      tdur[t] = c - tpre[t];
      tsolve[t] = sc;
    }
    constexpr double kHop = 1.5;   // flag + payload round trip between two workgroups of one launch
    const int P = std::max(1, opt.n_cus);
    // list schedule of a forest of tasks whose dependencies point ONE way (deps -> parent for the factorisation,
    // parent -> children for the back substitution); returns the start order and the makespan
    auto list_schedule = [&](const std::vector<std::vector<int>> &succ, const std::vector<int> &n_pred, const std::vector<double> &lead,
                             const std::vector<double> &dur, std::vector<int32_t> &order) {
      std::vector<double> blevel(nt, 0.0), ready(nt, 0.0), finish(nt, 0.0);
      {
        // longest remaining path: successors come later in a topological order of `succ`; compute by memoised DFS
        std::vector<int> state(nt, 0), stack;
        for (int r = 0; r < nt; r++) {
          if (state[r]) continue;
          stack.push_back(r);
          while (!stack.empty()) {
            const int t = stack.back();
            if (state[t] == 0) {
              state[t] = 1;
              for (int s2 : succ[t]) if (!state[s2]) stack.push_back(s2);
            } else {
              stack.pop_back();
              if (state[t] == 2) continue;
              state[t] = 2;
              double b = 0;
              for (int s2 : succ[t]) b = std::max(b, kHop + blevel[s2] - lead[s2]);
              blevel[t] = lead[t] + dur[t] + std::max(0.0, b);
            }
          }
        }
      }
      // tasks whose predecessors have all been started: `later` by the time from which they would not wait, `now_ok` (those
      // that would not wait at the current time) by the longest remaining path
      std::vector<int> pending(n_pred);
      using Fut = std::pair<double, int>;
      std::priority_queue<Fut, std::vector<Fut>, std::greater<Fut>> later;
      auto less_urgent = [&](int x, int y) { return blevel[x] != blevel[y] ? blevel[x] < blevel[y] : x > y; };
      std::priority_queue<int, std::vector<int>, decltype(less_urgent)> now_ok(less_urgent);
      for (int t = 0; t < nt; t++) if (pending[t] == 0) later.push(Fut{ready[t] - lead[t], t});
      std::priority_queue<double, std::vector<double>, std::greater<double>> slots;
      for (int k = 0; k < P; k++) slots.push(0.0);
      order.clear();
      double makespan = 0.0;
      for (int done = 0; done < nt; done++) {
        const double now = slots.top();
        slots.pop();
        while (!later.empty() && later.top().first <= now) { now_ok.push(later.top().second); later.pop(); }
        int best;
        if (!now_ok.empty()) { best = now_ok.top(); now_ok.pop(); }
        else { best = later.top().second; later.pop(); }   // every known task would wait: the one that becomes ready first
        const double go = std::max(now + lead[best], ready[best]);
        finish[best] = go + dur[best];
        makespan = std::max(makespan, finish[best]);
        slots.push(finish[best]);
        order.push_back(best);
        for (int s2 : succ[best]) {
          ready[s2] = std::max(ready[s2], finish[best] + kHop);
          if (--pending[s2] == 0) later.push(Fut{ready[s2] - lead[s2], s2});
        }
      }
      return makespan;
    };
    std::vector<std::vector<int>> up(nt), down(nt);
    std::vector<int> n_pred_solve(nt, 0);
    for (int t = 0; t < nt; t++)
      if (tparent[t] >= 0) { up[t].push_back(tparent[t]); down[tparent[t]].push_back(t); n_pred_solve[t] = 1; }
    std::vector<int32_t> forder, sorder;
    const double f_us = nt ? list_schedule(up, ndeps, tpre, tdur, forder) : 0.0;
    const std::vector<double> zero(nt, 0.0);
    const double s_us = nt ? list_schedule(down, n_pred_solve, zero, tsolve, sorder) : 0.0;
    // the fronts beyond LDS, level by level behind the launch, grouped as the level schedule groups them (big_level:
    // measured on sphere2500, fewer and wider levels -- a level = the big fronts whose big children are all in earlier
    // ones -- cost the big-front launches 4 %); priced like the level schedule does
    const std::vector<int32_t> &blvl = big_level;
    int maxb = -1;
    for (int f = 0; f < S; f++) maxb = std::max(maxb, (int)blvl[f]);
    double crit = 1.5 + f_us + 1.5 + s_us;
    for (int L = 0; L <= maxb; L++) {
      double w2 = 0.0;
      for (int f = 0; f < S; f++) if (blvl[f] == L) w2 = std::max(w2, cost[f]);
      crit += 1.5 + w2;
    }
    if (!keep) return crit;
    sym.task_ptr.assign(1, 0);
    sym.task_sn.clear();
    sym.steps.clear();
    std::vector<int32_t> new_id(nt, -1);
    if (nt > 0) {
      Step st{};
      st.kind = STEP_TASKS;
      st.sn = -1;
      for (int k = 0; k < nt; k++) {
        new_id[forder[k]] = k;
        for (int f : tasks[forder[k]]) {
          sym.task_sn.push_back(f);
          st.max_front = std::max(st.max_front, sym.sn_ncols[f] + sym.sn_nrows[f] + 1);
          st.max_lds_elems = std::max<int64_t>(st.max_lds_elems, lds_elems(sym.sn_ncols[f], sym.sn_nrows[f]));
        }
        sym.task_ptr.push_back((int)sym.task_sn.size());
      }
      st.task_end = nt;
      {
        const int mf = st.max_front << opt.threads_shift;
        st.threads = mf <= 20 ? 64 : mf <= 48 ? 128 : mf <= 96 ? 256 : mf <= 128 ? 512 : 1024;
      }
      sym.steps.push_back(st);
    }
    for (int L = 0; L <= maxb; L++) {
      Step b{};
      b.kind = STEP_BIG;
      b.sn = -1;
      b.task_begin = (int)sym.task_ptr.size() - 1;
      for (int f = 0; f < S; f++)
        if (blvl[f] == L) {
          if (!sym.sn_huge[f]) return -1.0;   // (the one-workgroup class of fronts beyond LDS has no kernel: cannot happen with panel_budget_elems == 0)
          sym.task_sn.push_back(f);
          sym.task_ptr.push_back((int)sym.task_sn.size());
          b.max_front = std::max(b.max_front, sym.sn_ncols[f] + sym.sn_nrows[f] + 1);
        }
      b.task_end = (int)sym.task_ptr.size() - 1;
      b.threads = 1024;
      if (b.task_end > b.task_begin) sym.steps.push_back(b);
    }
    sym.solve_order.resize(nt);
    for (int k = 0; k < nt; k++) sym.solve_order[k] = new_id[sorder[k]];
    sym.lds_flow = nt > 0;
    sym.est_factor_us = f_us;
    sym.est_solve_us = s_us;
    sym.est_critical_us = crit;
    return crit;
  };
  // (the dataflow step holds EVERY front that lives in LDS and runs before the first level of fronts beyond LDS: that needs
  // the big fronts to sit above all LDS fronts.  A front's parent is not always the larger of the two -- a child with a
  // wide pivot block can be beyond LDS under an LDS parent; such a tree keeps the level schedule.)
  bool big_below_lds = false;
  for (int f = 0; f < S && !big_below_lds; f++)
    if (sym.sn_big[f] && sym.sn_parent[f] >= 0 && !sym.sn_big[sym.sn_parent[f]]) big_below_lds = true;
  // ... and it pays where the LDS fronts are bound by latency, not where thousands of tasks keep every CU busy anyway (the
  // 1M-edge lattice: 9 255 LDS fronts, 2 634 tasks in two levels -- as ONE launch of waiting workgroups 555 + 143 us against
  // 521 + 122 us of level launches; sphere2500, 545 LDS fronts: 91 + 38 against 126 + 56 us)
  int n_lds_fronts = 0;
  for (int f = 0; f < S; f++) n_lds_fronts += !sym.sn_big[f];
  if (opt.lds_flow && opt.n_parts <= 1 && S > 0 && !big_below_lds && n_lds_fronts <= 8 * opt.n_cus) {
    if (sym.n_big > 0) {   // the levels of the fronts beyond LDS: from the level schedule
      double best = 1e300, best_t = 90;
      for (double t = 10.0; t < 260.0; t *= 1.12) {
        const double c = build_schedule(t);
        if (c < best) { best = c; best_t = t; }
      }
      build_schedule(best_t);
      int L = 0;
      for (const Step &st : sym.steps)
        if (st.kind == STEP_BIG) {
          for (int t = st.task_begin; t < st.task_end; t++) big_level[sym.task_sn[sym.task_ptr[t]]] = L;
          L++;
        }
    }
    double best = 1e300, best_t = 40;
    if (opt.task_us > 0) best_t = opt.task_us;
    else
      // (the score is a step function of the threshold; the measured optimum was at or near the finest granularity on every
      // reference dataset -- intel and dlr 4, M3500 25 -- so a coarse grid is enough)
      for (double t : {4.0, 8.0, 15.0, 25.0, 40.0, 60.0, 90.0, 140.0}) {
        const double c = build_flow_schedule(t, false);
        if (c < best) { best = c; best_t = t; }
      }
    build_flow_schedule(best_t, true);
    sym.task_us_used = best_t;
  } else if (opt.task_us > 0) {
    build_schedule(opt.task_us);
  } else {
    // the score is a jagged function of the threshold (a subtree flips between "leaf task" and "top"):
    // a dense geometric grid, each candidate costs one O(S) pass
    double best = 1e300, best_t = 90;
    for (double t = 10.0; t < 260.0; t *= 1.12) {
      const double c = build_schedule(t);
      if (c < best) { best = c; best_t = t; }
    }
    build_schedule(best_t);
    sym.task_us_used = best_t;
  }
  ptimer.mark("schedule");
  return "";
}

}  // namespace rrpgo
