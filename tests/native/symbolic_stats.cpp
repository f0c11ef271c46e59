// Prints symbolic-analysis statistics for a g2o file or a synthetic grid (host only).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
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
  if (getenv("FLOW")) opt.lds_flow = true;
  if (getenv("API")) {   // the options rr_pgo_create uses for a graph of this size (pgo_api.hip, build_handle), fp32 budget
    opt.lds_budget_elems = getenv("F64") ? 19000 : 38000;
    opt.nd_leaf = g.n_nodes() <= 6000 ? (1 << 30) : (getenv("F64") ? 32 : 48);
    opt.split_separators = g.n_nodes() > 6000;
  }
  Symbolic s;
  auto t0 = std::chrono::steady_clock::now();
  std::string e = analyze(g, opt, s);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!e.empty()) { printf("analyze error: %s\n", e.c_str()); return 1; }
  printf("N=%d E=%d dim=%d | analyze %.1f ms | S=%d Lblocks=%lld l_elems=%lld u_elems=%lld flops=%.3g | maxfront=%d maxpiv=%d big=%d | steps=%zu tasks=%zu est_crit=%.1f us task_us=%.0f\n",
         s.N, g.n_edges(), s.dim, ms, s.S, (long long)s.nnz_l_blocks, (long long)s.l_elems, (long long)s.u_elems,
         (double)s.factor_flops, s.max_front, s.max_pivot_cols, s.n_big, s.steps.size(), s.task_ptr.size() - 1, s.est_critical_us, s.task_us_used);
  if (s.lds_flow) printf("  lds_flow: %zu tasks, est factor %.1f us, solve %.1f us\n", s.task_ptr.size() - 1, s.est_factor_us, s.est_solve_us);
  if (getenv("HIST")) {
    // fronts by size class: count, flops share
    const int edges[] = {64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 1 << 30};
    long long cnt[11] = {0}, big[11] = {0};
    double fl[11] = {0};
    for (int f = 0; f < s.S; f++) {
      int nc = s.sn_ncols[f], M = nc + s.sn_nrows[f] + 1;
      int b = 0;
      while (M > edges[b]) b++;
      cnt[b]++;
      big[b] += s.sn_big[f];
      fl[b] += (double)nc * M * M;
    }
    double tot = 0;
    for (int b = 0; b < 11; b++) tot += fl[b];
    for (int b = 0; b < 11; b++)
      printf("  M<=%-10d fronts %-7lld (big %-6lld) flops %5.1f%%\n", edges[b], cnt[b], big[b], 100 * fl[b] / tot);
    int nsteps_tasks = 0, nsteps_big = 0;
    for (auto &st : s.steps) (st.kind == STEP_BIG ? nsteps_big : nsteps_tasks)++;
    printf("  steps: %d task launches, %d big-front steps\n", nsteps_tasks, nsteps_big);
    return 0;
  }
  if (getenv("TASKHIST"))
    for (auto &st : s.steps) {
      if (st.kind != STEP_TASKS) continue;
      // per task: largest LDS need, largest M, fronts, model cost
      const long long le[] = {2400, 4800, 9600, 19000, 38000, 1 << 30};
      long long cnt[6] = {0}, fr[6] = {0}, mM[6] = {0};
      for (int t = st.task_begin; t < st.task_end; t++) {
        long long lds = 0, M = 0;
        for (int q = s.task_ptr[t]; q < s.task_ptr[t + 1]; q++) {
          int f = s.task_sn[q], nc = s.sn_ncols[f], nr = s.sn_nrows[f];
          lds = std::max<long long>(lds, (long long)(nc + nr + 1) * nc + (long long)(nr + 1) * (nr + 2) / 2);
          M = std::max<long long>(M, nc + nr + 1);
        }
        int b = 0;
        while (lds > le[b]) b++;
        cnt[b]++; fr[b] += s.task_ptr[t + 1] - s.task_ptr[t]; mM[b] = std::max(mM[b], M);
      }
      printf("  step: %d tasks;", st.task_end - st.task_begin);
      for (int b = 0; b < 6; b++) printf(" lds<=%lld: %lld tasks %lld fronts maxM %lld |", le[b], cnt[b], fr[b], mM[b]);
      printf("\n");
    }
  if (getenv("FRONTS"))   // every front of every big step: pivot columns x front order
    for (auto &st : s.steps) {
      if (st.kind != STEP_BIG) continue;
      printf("  big step, %d fronts:", st.task_end - st.task_begin);
      long long fl = 0, st_bytes = 0;
      int shown = 0;
      for (int t = st.task_begin; t < st.task_end; t++) {
        const int f = s.task_sn[s.task_ptr[t]], nc = s.sn_ncols[f], M = nc + s.sn_nrows[f] + 1;
        fl += (long long)nc * M * M; st_bytes += (long long)M * M;
        if (shown++ < 6) printf(" %dx%d(kids %d)", nc, M, s.child_ptr[f + 1] - s.child_ptr[f]);
      }
      printf(" ... nc*M^2 = %.2f G, storage %.1f M elems\n", fl * 1e-9, st_bytes * 1e-6);
    }
  for (auto &st : s.steps) {
    printf("  %s %d (threads %d, maxM %d, lds %d)\n", st.kind == STEP_TASKS ? "tasks" : st.kind == STEP_MID ? "mid" : "HUGE",
           st.task_end - st.task_begin, st.threads, st.max_front, st.max_lds_elems);
  }
  return 0;
}
// (histogram helper appended below main via env HIST=1 is handled in main)
