// mf_host_check.cpp -- TEST HELPER (host only, never part of librr_pgo.so).
//
// Validates the tables produced by rustrobotics_amd/csrc/symbolic.cpp without a
// GPU: fills the H block structure with random SPD values, walks the fronts
// exactly as the HIP kernels do (assembly items, extend-add through the `rel`
// maps, partial Cholesky with the rhs carried as the last row, back
// substitution through sn_rows) in the order the level schedule prescribes,
// and checks  || b - H x ||_inf / ||b||_inf.  Also checks that the schedule is
// a valid topological order (children done in an earlier step, or earlier in
// the same task).
//
// usage: mf_host_check <file.g2o> | grid W H [n_edges]   [env LEAF, PARTS, LDS]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "host_graph.h"
#include "symbolic.h"
using namespace rrpgo;

static int64_t tri(int n, int ld, int i, int j) {
  return ld > 0 ? (int64_t)j * ld + i : (int64_t)j * n - (int64_t)j * (j - 1) / 2 + (i - j);
}

int main(int argc, char **argv) {
  HostGraph g;
  if (argc >= 4 && !strcmp(argv[1], "grid")) {
    synth_grid(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoll(argv[4]) : 0, 42, 43, g);
  } else if (argc >= 2) {
    bool io;
    std::string e = load_g2o(argv[1], g, io);
    if (!e.empty()) { printf("load error: %s\n", e.c_str()); return 2; }
  } else {
    return 2;
  }
  SymbolicOptions opt;
  if (getenv("LEAF")) opt.nd_leaf = atoi(getenv("LEAF"));
  if (getenv("PARTS")) opt.n_parts = atoi(getenv("PARTS"));
  if (getenv("LDS")) opt.lds_budget_elems = atoll(getenv("LDS"));
  if (getenv("FLOW")) opt.lds_flow = true;   // the dataflow schedule of lds_flow.hip.h (tasks of ONE step in ticket order)
  // the chain passes of step 5 (symbolic.cpp): fronts joining their parent / handing it their last nodes
  if (getenv("MERGE_NC")) opt.merge_chain_nc = atoi(getenv("MERGE_NC"));
  if (getenv("MERGE_GAIN")) opt.merge_chain_gain_us = atof(getenv("MERGE_GAIN"));
  if (getenv("BALANCE")) { opt.balance_blocks = atoi(getenv("BALANCE")) != 0; if (atoi(getenv("BALANCE")) > 1) opt.balance_max_rem = atoi(getenv("BALANCE")); }
  if (getenv("AMALG_NP")) opt.amalg_np = atoi(getenv("AMALG_NP"));
  if (getenv("ML")) opt.ml_nd = true;   // the multilevel bisection with a minimum-cover separator (graphs of up to 6000 nodes in the engine)
  if (getenv("PIN")) opt.pin_node = g.anchor_node;   // sharded runs of the engine: the anchor joins the top separator
  Symbolic y;
  std::string e = analyze(g, opt, y);
  if (!e.empty()) { printf("analyze error: %s\n", e.c_str()); return 2; }
  const int N = y.N, dim = y.dim, S = y.S;

  // ---- random SPD values on the block pattern
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> hv((size_t)y.n_hvals, 0.0), b(dim), rowsum(dim, 0.0);
  for (int64_t s = 0; s < y.n_offblocks; s++) {
    int rn = y.blk_row[s], cn = y.blk_col[s];
    int dr = node_dim(g.node_kind[rn]), dc = node_dim(g.node_kind[cn]);
    for (int i = 0; i < dr; i++)
      for (int j = 0; j < dc; j++) {
        double v = U(rng);
        hv[y.blk_off[s] + i * dc + j] = v;
        rowsum[g.node_offset[rn] + i] += std::fabs(v);
        rowsum[g.node_offset[cn] + j] += std::fabs(v);
      }
  }
  for (int v = 0; v < N; v++) {
    int d = node_dim(g.node_kind[v]);
    for (int i = 0; i < d; i++)
      for (int j = 0; j <= i; j++) {
        double x = i == j ? 0.0 : U(rng);
        hv[y.diag_off[v] + i * d + j] = x;
        hv[y.diag_off[v] + j * d + i] = x;
        if (i != j) { rowsum[g.node_offset[v] + i] += std::fabs(x); rowsum[g.node_offset[v] + j] += std::fabs(x); }
      }
    for (int i = 0; i < d; i++) hv[y.diag_off[v] + i * d + i] = rowsum[g.node_offset[v] + i] + 1.0 + std::fabs(U(rng));
  }
  for (int i = 0; i < dim; i++) b[i] = U(rng);

  // ---- schedule check + factor order.  Sharded (PARTS > 1): every rank's schedule is analysed;
  // all ranks' local steps run first (in rank order, they are independent), then the shared top
  // steps once -- which is what the all-reduce of the boundary update matrices makes possible.
  std::vector<int> done_step(S, -1), order;
  std::vector<int> pos_in_task(S, -1);
  int n_sched = 0, step_counter = 0;
  auto run_steps = [&](const Symbolic &ys, size_t from, size_t to) {
    for (size_t si = from; si < to; si++, step_counter++) {
      const Step &st = ys.steps[si];
      for (int t = st.task_begin; t < st.task_end; t++)
        for (int q = ys.task_ptr[t]; q < ys.task_ptr[t + 1]; q++) {
          int s = ys.task_sn[q];
          if (st.kind == STEP_TASKS && ys.sn_big[s]) { printf("FAIL: big front %d inside an LDS task\n", s); exit(1); }
          if (st.kind != STEP_TASKS && ys.task_ptr[t + 1] - ys.task_ptr[t] != 1) { printf("FAIL: bad batch task %d\n", t); exit(1); }
          if (st.kind == STEP_MID && (!ys.sn_big[s] || ys.sn_huge[s])) { printf("FAIL: bad mid front %d\n", s); exit(1); }
          if (st.kind == STEP_BIG && !ys.sn_huge[s]) { printf("FAIL: STEP_BIG on a front that is not huge\n"); exit(1); }
          if (done_step[s] >= 0) { printf("FAIL: supernode %d scheduled twice\n", s); exit(1); }
          for (int cq = ys.child_ptr[s]; cq < ys.child_ptr[s + 1]; cq++) {
            int c = ys.child_list[cq];
            // level schedule: in an earlier step, or earlier in the same task; dataflow schedule (ONE step, tasks in
            // ticket order): earlier in the list, i.e. in a task with a smaller ticket or earlier in the same task
            bool ok = done_step[c] >= 0 && (ys.lds_flow || done_step[c] < step_counter || (st.kind == STEP_TASKS && pos_in_task[c] == t + 1000000 * (int)(&ys != &y)));
            if (!ok) { printf("FAIL: supernode %d scheduled before its child %d\n", s, c); exit(1); }
          }
          done_step[s] = step_counter;
          pos_in_task[s] = t + 1000000 * (int)(&ys != &y);
          order.push_back(s);
          n_sched++;
        }
    }
  };
  if (opt.n_parts <= 1) {
    run_steps(y, 0, y.steps.size());
  } else {
    for (int p = 0; p < opt.n_parts; p++) {
      SymbolicOptions op = opt;
      op.my_part = p;
      Symbolic yp;
      std::string ep = analyze(g, op, yp);
      if (!ep.empty()) { printf("analyze error (rank %d): %s\n", p, ep.c_str()); return 2; }
      if (yp.l_elems != y.l_elems || yp.xch_elems != y.xch_elems || yp.S != y.S) { printf("FAIL: rank layouts differ\n"); return 1; }
      for (int s = 0; s < S; s++)
        if (yp.sn_owner[s] != y.sn_owner[s]) { printf("FAIL: owners differ between ranks\n"); return 1; }
      // every rank factors the SHARED fronts itself: their part of the schedule (which fronts share a step, in which order)
      // must be the same on every rank, or the ranks' copies of the shared poses drift apart by rounding (r05, sphere2500 over 8)
      {
        auto shared_part = [](const Symbolic &z) {
          std::vector<int32_t> v;
          for (size_t i = (size_t)z.n_local_steps; i < z.steps.size(); i++) {
            v.push_back(-1 - z.steps[i].kind);
            for (int t = z.steps[i].task_begin; t < z.steps[i].task_end; t++) {
              v.push_back(-100);
              for (int q = z.task_ptr[t]; q < z.task_ptr[t + 1]; q++) v.push_back(z.task_sn[q]);
            }
          }
          return v;
        };
        static std::vector<int32_t> first;
        if (p == 0) first = shared_part(yp);
        else if (shared_part(yp) != first) { printf("FAIL: rank %d schedules the shared fronts differently from rank 0\n", p); return 1; }
      }
      run_steps(yp, 0, (size_t)yp.n_local_steps);
      if (p == opt.n_parts - 1) run_steps(yp, (size_t)yp.n_local_steps, yp.steps.size());
    }
    int nb = 0;
    for (int s = 0; s < S; s++) nb += y.sn_xch_off[s] >= 0;
    printf("sharded over %d ranks: %d boundary fronts, exchange buffer %lld scalars\n", opt.n_parts, nb, (long long)y.xch_elems);
  }
  if (n_sched != S) { printf("FAIL: schedule covers %d of %d supernodes\n", n_sched, S); return 1; }
  if (getenv("FLOW") && !getenv("FLOW_OPTIONAL") && !y.lds_flow) { printf("FAIL: no dataflow schedule was built\n"); return 1; }
  if (y.lds_flow) {
    // the back substitution's ticket order: a permutation of the tasks in which a front's parent is solved in a task
    // with a smaller ticket, or later in the same task's (reversed) walk
    // step 0 = the dataflow step (every LDS front), then the levels of fronts beyond LDS
    if (y.steps.empty() || y.steps[0].kind != STEP_TASKS || y.steps[0].task_begin != 0) { printf("FAIL: dataflow schedule shape\n"); return 1; }
    for (size_t si = 1; si < y.steps.size(); si++)
      if (y.steps[si].kind != STEP_BIG) { printf("FAIL: an LDS step behind the dataflow step\n"); return 1; }
    const int nt = y.steps[0].task_end;
    if ((int)y.solve_order.size() != nt) { printf("FAIL: dataflow schedule shape\n"); return 1; }
    std::vector<int> ticket_of(nt, -1), task_of(S, -1);
    for (int k = 0; k < nt; k++) {
      const int t = y.solve_order[k];
      if (t < 0 || t >= nt || ticket_of[t] >= 0) { printf("FAIL: solve order is not a permutation\n"); return 1; }
      ticket_of[t] = k;
    }
    for (int t = 0; t < nt; t++)
      for (int q = y.task_ptr[t]; q < y.task_ptr[t + 1]; q++) {
        if (y.sn_big[y.task_sn[q]]) { printf("FAIL: a front beyond LDS in the dataflow step\n"); return 1; }
        task_of[y.task_sn[q]] = t;
      }
    for (int s2 = 0; s2 < S; s2++) {
      const int p2 = y.sn_parent[s2];
      if (p2 < 0 || y.sn_big[s2] || y.sn_big[p2]) continue;   // (fronts beyond LDS are solved by earlier launches)
      if (task_of[p2] != task_of[s2] && ticket_of[task_of[p2]] >= ticket_of[task_of[s2]]) { printf("FAIL: front %d is solved before its parent\n", s2); return 1; }
      if (task_of[p2] == task_of[s2] && p2 < s2) { printf("FAIL: parent before child inside a task\n"); return 1; }
    }
    printf("dataflow schedule: %d tasks, model %.1f + %.1f us\n", nt, y.est_factor_us, y.est_solve_us);
  }

  // ---- the longest chain of the supernode tree in pivot columns (what a small graph's iteration time follows): a guard on the
  // quality of the dissection -- CHAIN_MAX=<columns> fails the run when the chain is longer
  {
    std::vector<int64_t> chain(S, 0);
    int64_t longest = 0;
    for (int s2 = 0; s2 < S; s2++) {   // (children precede parents)
      chain[s2] += y.sn_ncols[s2];
      longest = std::max(longest, chain[s2]);
      const int p2 = y.sn_parent[s2];
      if (p2 >= 0) chain[p2] = std::max(chain[p2], chain[s2]);
    }
    printf("longest chain: %lld pivot columns\n", (long long)longest);
    if (getenv("CHAIN_MAX") && longest > atoll(getenv("CHAIN_MAX"))) { printf("FAIL: the chain is longer than %s columns\n", getenv("CHAIN_MAX")); return 1; }
  }

  std::vector<double> L((size_t)y.l_elems + 4, 0.0), Uv((size_t)y.u_elems + 4, 0.0), x(dim, 0.0);
  std::vector<double> P, Uloc;
  for (int s : order) {
    const int nc = y.sn_ncols[s], nr = y.sn_nrows[s], M = nc + nr + 1, nu = nr + 1;
    P.assign((size_t)M * nc, 0.0);
    Uloc.assign((size_t)nu * (nu + 1) / 2, 0.0);
    for (int64_t q = y.asm_ptr[s]; q < y.asm_ptr[s + 1]; q++) {
      const AsmItem &it = y.asm_items[q];
      for (int i = 0; i < it.drow; i++)
        for (int j = 0; j < it.dcol; j++) {
          if (it.diag == 1 && i < j) continue;
          P[(size_t)(it.lcol + j) * M + it.lrow + i] += hv[it.src + i * it.dcol + j];
        }
    }
    {
      // the flat lists the kernels read must give the same image (entries of parallel edges in the dup list); for a
      // front beyond LDS column by column through fasm_colptr (k_big_build adds a column's entries right after writing it)
      std::vector<double> P2((size_t)M * nc, 0.0);
      for (int64_t t = y.fasm_ptr[s]; t < y.fasm_ptr[s + 1]; t++) P2[(size_t)y.fasm_dst[t]] += hv[y.fasm_src[t]];
      for (int64_t t = y.fdup_ptr[s]; t < y.fdup_ptr[s + 1]; t++) P2[(size_t)y.fdup_dst[t]] += hv[y.fdup_src[t]];
      for (size_t k = 0; k < P2.size(); k++)
        if (P2[k] != P[k]) { printf("FAIL: flat assembly list of supernode %d differs from its items at entry %zu\n", s, k); return 1; }
      if (y.sn_big[s]) {
        const int c0 = y.sn_col0[s];
        if (y.fasm_colptr[c0] != y.fasm_ptr[s] || y.fasm_colptr[c0 + nc] != y.fasm_ptr[s + 1]) { printf("FAIL: column pointers of front %d do not span its list\n", s); return 1; }
        for (int J = 0; J < nc; J++)
          for (int t = y.fasm_colptr[c0 + J]; t < y.fasm_colptr[c0 + J + 1]; t++)
            if (y.fasm_dst[t] / M != J) { printf("FAIL: entry %d of front %d is not in column %d\n", t, s, J); return 1; }
      }
    }
    for (int j = 0; j < nc; j++) P[(size_t)j * M + M - 1] = b[y.perm[y.sn_col0[s] + j]];
    for (int q = y.child_ptr[s]; q < y.child_ptr[s + 1]; q++) {
      int c = y.child_list[q];
      int ncu = y.sn_nrows[c] + 1, cld = y.sn_uld[c];
      const double *Uc = (cld > 0 ? L.data() : Uv.data()) + y.sn_uoff[c];
      const int32_t *rel = y.rel.data() + y.rel_ptr[c];
      for (int j = 0; j < ncu; j++)
        for (int i = j; i < ncu; i++) {
          if (i == ncu - 1 && j == ncu - 1) continue;
          double v = Uc[tri(ncu, cld, i, j)];
          int li = rel[i], lj = rel[j];
          if (li < lj) { printf("FAIL: rel map not monotone\n"); return 1; }
          if (lj < nc) P[(size_t)lj * M + li] += v;
          else Uloc[tri(nu, 0, li - nc, lj - nc)] += v;
        }
    }
    for (int k = 0; k < nc; k++) {
      double d = P[(size_t)k * M + k];
      if (!(d > 0)) { printf("FAIL: non-positive pivot in supernode %d\n", s); return 1; }
      d = std::sqrt(d);
      P[(size_t)k * M + k] = d;
      for (int i = k + 1; i < M; i++) P[(size_t)k * M + i] /= d;
      for (int j = k + 1; j < nc; j++)
        for (int i = j; i < M; i++) P[(size_t)j * M + i] -= P[(size_t)k * M + i] * P[(size_t)k * M + j];
    }
    for (int j = 0; j < nu; j++)
      for (int i = j; i < nu; i++) {
        double sacc = 0;
        for (int k = 0; k < nc; k++) sacc += P[(size_t)k * M + nc + i] * P[(size_t)k * M + nc + j];
        Uloc[tri(nu, 0, i, j)] -= sacc;
      }
    if (!y.sn_big[s]) {
      std::copy(P.begin(), P.end(), L.begin() + y.sn_loff[s]);
      std::copy(Uloc.begin(), Uloc.end(), Uv.begin() + y.sn_uoff[s]);
    } else {
      double *F = L.data() + y.sn_loff[s];
      for (int j = 0; j < nc; j++)
        for (int i = 0; i < M; i++) F[(size_t)j * M + i] = P[(size_t)j * M + i];
      for (int j = 0; j < nu; j++)
        for (int i = j; i < nu; i++) L[y.sn_uoff[s] + tri(nu, M, i, j)] = Uloc[tri(nu, 0, i, j)];
    }
  }
  for (size_t oi = order.size(); oi-- > 0;) {
    int s = order[oi];
    const int nc = y.sn_ncols[s], nr = y.sn_nrows[s], M = nc + nr + 1;
    const double *Lg = L.data() + y.sn_loff[s];
    const int32_t *rows = y.sn_rows.data() + y.sn_rows_ptr[s];
    std::vector<double> t(nc);
    for (int j = 0; j < nc; j++) {
      double sacc = 0;
      for (int i = 0; i < nr; i++) sacc += Lg[(size_t)j * M + nc + i] * x[rows[i]];
      t[j] = Lg[(size_t)j * M + M - 1] - sacc;
    }
    for (int j = nc - 1; j >= 0; j--) {
      double sacc = 0;
      for (int i = j + 1; i < nc; i++) sacc += Lg[(size_t)j * M + i] * t[i];
      t[j] = (t[j] - sacc) / Lg[(size_t)j * M + j];
    }
    for (int j = 0; j < nc; j++) x[y.sn_col0[s] + j] = t[j];
  }
  // ---- residual in reference scalar order
  std::vector<double> xr(dim), r(b);
  for (int i = 0; i < dim; i++) xr[y.perm[i]] = x[i];
  for (int v = 0; v < N; v++) {
    int d = node_dim(g.node_kind[v]), o = g.node_offset[v];
    for (int i = 0; i < d; i++)
      for (int j = 0; j < d; j++) r[o + i] -= hv[y.diag_off[v] + i * d + j] * xr[o + j];
  }
  for (int64_t s = 0; s < y.n_offblocks; s++) {
    int rn = y.blk_row[s], cn = y.blk_col[s];
    int dr = node_dim(g.node_kind[rn]), dc = node_dim(g.node_kind[cn]);
    int ro = g.node_offset[rn], co = g.node_offset[cn];
    for (int i = 0; i < dr; i++)
      for (int j = 0; j < dc; j++) {
        double v = hv[y.blk_off[s] + i * dc + j];
        r[ro + i] -= v * xr[co + j];
        r[co + j] -= v * xr[ro + i];
      }
  }
  double rmax = 0, bmax = 0;
  for (int i = 0; i < dim; i++) { rmax = std::max(rmax, std::fabs(r[i])); bmax = std::max(bmax, std::fabs(b[i])); }
  printf("N=%d dim=%d S=%d steps=%zu big=%d maxfront=%d  rel_residual=%.3e\n", N, dim, S, y.steps.size(), y.n_big,
         y.max_front, rmax / bmax);
  if (!(rmax / bmax < 1e-9)) { printf("FAIL: residual too large\n"); return 1; }
  printf("OK\n");
  return 0;
}
