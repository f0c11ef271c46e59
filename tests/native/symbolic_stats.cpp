// Prints symbolic-analysis statistics for a g2o file or a synthetic grid (host only).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "host_graph.h"
#include "symbolic.h"
using namespace rrpgo;
int main(int argc, char **argv) {
  HostGraph g;
  if (argc >= 4 && !strcmp(argv[1], "grid")) {
    synth_grid(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoll(argv[4]) : 0, 42, 43, g);
  } else {
    bool io;
    std::string e = load_g2o(argv[1], g, io);
    if (!e.empty()) { printf("load error: %s\n", e.c_str()); return 1; }
  }
  SymbolicOptions opt;
  if (getenv("LEAF")) opt.nd_leaf = atoi(getenv("LEAF"));
  if (getenv("PARTS")) opt.n_parts = atoi(getenv("PARTS"));
  if (getenv("LDS")) opt.lds_budget_elems = atoll(getenv("LDS"));
  if (getenv("TASKUS")) opt.task_us = atof(getenv("TASKUS"));
  Symbolic s;
  auto t0 = std::chrono::steady_clock::now();
  std::string e = analyze(g, opt, s);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!e.empty()) { printf("analyze error: %s\n", e.c_str()); return 1; }
  printf("N=%d E=%d dim=%d | analyze %.1f ms | S=%d Lblocks=%lld l_elems=%lld u_elems=%lld flops=%.3g | maxfront=%d maxpiv=%d big=%d | steps=%zu tasks=%zu est_crit=%.1f us\n",
         s.N, g.n_edges(), s.dim, ms, s.S, (long long)s.nnz_l_blocks, (long long)s.l_elems, (long long)s.u_elems,
         (double)s.factor_flops, s.max_front, s.max_pivot_cols, s.n_big, s.steps.size(), s.task_ptr.size() - 1, s.est_critical_us);
  for (auto &st : s.steps) {
    if (st.kind == STEP_TASKS) printf("  tasks %d (threads %d, maxM %d, lds %d)\n", st.task_end - st.task_begin, st.threads, st.max_front, st.max_lds_elems);
    else printf("  BIG sn %d M=%d nc=%d\n", st.sn, st.max_front, s.sn_ncols[st.sn]);
  }
  return 0;
}
