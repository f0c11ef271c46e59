// nd_replay_check.cpp -- host-only check of the shared dissection (symbolic.h, NdSplitTable): the engine dissects a small graph ONCE, down
// to the deepest candidate's leaves, recording every split; the candidate analyses replay the splits and stop at their own leaf size.
// A replayed analysis must be the computed one, table for table, at every depth.  usage: nd_replay_check file.g2o ...
#include <cstdio>
#include <cstdlib>
#include "host_graph.h"
#include "symbolic.h"
using namespace rrpgo;
template <class V> unsigned long long hv(const V &v) { unsigned long long h = 1469598103934665603ull; for (auto x : v) { h ^= (unsigned long long)(long long)x; h *= 1099511628211ull; } return h; }
int main(int argc, char **argv) {
  int bad = 0;
  for (int a = 1; a < argc; a++) {
    HostGraph g; bool io;
    load_g2o(argv[a], g, io);
    SymbolicOptions base; base.lds_budget_elems = 19000; base.lds_flow = true; base.amalg_np = 16; base.ml_nd = true;
    NdSplitTable tab;
    { SymbolicOptions o = base; o.nd_leaf = 50; o.nd_record = &tab; std::string e = dissect_only(g, o); if (!e.empty()) { printf("%s\n", e.c_str()); return 1; } }
    for (int leaf : {1 << 30, 250, 150, 100, 70, 50}) {
      Symbolic s1, s2;
      SymbolicOptions o = base; o.nd_leaf = leaf;
      analyze(g, o, s1);
      o.nd_replay = &tab;
      analyze(g, o, s2);
      const bool same = s1.order == s2.order && s1.sn_first_pos == s2.sn_first_pos && s1.sn_parent == s2.sn_parent && s1.est_critical_us == s2.est_critical_us && hv(s1.task_sn) == hv(s2.task_sn);
      if (!same) { printf("%s leaf %d: DIFFERENT\n", argv[a], leaf); bad++; }
    }
    printf("%s: %zu splits recorded\n", argv[a], tab.map.size());
  }
  printf(bad ? "FAIL\n" : "replayed == computed for every depth\n");
  return bad;
}
