// kernels.hip.h -- gfx950 device code for the pose-graph Gauss-Newton path.
//
//   k_linearize      error + Jacobians + J^T W J / J^T W e + chi2     (reference :434-486,165-192,537-574)
//   k_factor_tasks   multifrontal supernodal Cholesky, fronts in LDS  (replaces umfpack.factorize, :138)
//   k_solve_tasks    back substitution down the supernode tree         (replaces umfpack.solve, :141)
//   k_update         update_nodes + |dx|^2                             (reference :229-245,273)
//   k_finalize       fixed-order reduction of the chi2 / |dx|^2 partials
//
// Wavefront = 64 lanes.  All cross-workgroup dependencies are kernel boundaries
// on one stream; inside a launch a workgroup only reads what it wrote itself
// or what an earlier launch wrote.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rrpgo {

template <typename T> struct VecT;
template <> struct VecT<float> { using V4 = float4; using V2 = float2; };
template <> struct VecT<double> { using V4 = double4; using V2 = double2; };

constexpr int LIN_GROUP = 8;      // lanes cooperating on one node in k_linearize
constexpr int LIN_THREADS = 256;
constexpr int UPD_THREADS = 256;

// device-side error flags (sticky, read by the host at sync points)
enum : int { DEVERR_NOT_SPD = 1 };

struct AsmItemDev {
  int64_t src;
  int32_t lrow, lcol;
  int16_t drow, dcol;
  int32_t diag;  // 0 off-diagonal block, 1 diagonal block (lower part used), 2 duplicate-edge block (serial pass)
};

template <typename T> struct LinArgs {
  int n_nodes;
  const typename VecT<T>::V4 *pose;     // x, y, cos, sin  (XY landmarks: x, y, -, -)
  const int2 *e_idx;                     // from, to
  const typename VecT<T>::V4 *e_meas;    // SE2: x, y, cos, sin | SE2_XY: x, y, 0, 0
  const typename VecT<T>::V4 *e_info_a;  // i11 i12 i13 i22
  const typename VecT<T>::V2 *e_info_b;  // i23 i33
  const int64_t *e_slot;                 // (offset into hvals << 1) | transposed
  const int32_t *inc_ptr, *inc_list;     // entry = edge << 2 | kind << 1 | role
  const uint8_t *node_dim;               // 3 (SE2) or 2 (XY)
  const int32_t *node_offset;            // reference scalar offset
  const int64_t *diag_off;
  T *hvals;
  T *b;                                  // reference scalar order, already negated (:361)
  double *chi2_partial;                  // one per workgroup
  int anchor;                            // node that gets the 1e7 prior (:330-336), -1 none
  T lambda;                              // added to every diagonal entry when > 0 (LM, :362-366)
  int write_system;                      // 0: chi2 only
};

// ---------------------------------------------------------------- factor maths

// Error and Jacobians of one 2D edge, zero padded to 3x3.
// kind 0: pose-pose (:434-447,457-486); kind 1: pose-landmark (:449-455,516-535).
template <typename T>
__device__ __forceinline__ void edge_linearize_2d(int kind, const typename VecT<T>::V4 &x1,
                                                  const typename VecT<T>::V4 &x2,
                                                  const typename VecT<T>::V4 &z, T e[3], T A[3][3],
                                                  T B[3][3]) {
  const T dx = x2.x - x1.x, dy = x2.y - x1.y;
  if (kind == 0) {
    // z^-1, x1^-1 (nalgebra Isometry::inverse), then (z^-1 * x1^-1) * x2
    const T zic = z.z, zis = -z.w;
    const T zitx = zic * (-z.x) - zis * (-z.y), zity = zis * (-z.x) + zic * (-z.y);
    const T xic = x1.z, xis = -x1.w;
    const T xitx = xic * (-x1.x) - xis * (-x1.y), xity = xis * (-x1.x) + xic * (-x1.y);
    const T mc = zic * xic - zis * xis, ms = zic * xis + zis * xic;
    const T mtx = zitx + (zic * xitx - zis * xity), mty = zity + (zis * xitx + zic * xity);
    const T etx = mtx + (mc * x2.x - ms * x2.y), ety = mty + (ms * x2.x + mc * x2.y);
    const T ec = mc * x2.z - ms * x2.w, es = mc * x2.w + ms * x2.z;
    e[0] = etx; e[1] = ety; e[2] = atan2(es, ec);
    // M = Rz^T R1^T ; a12 = M * (dy, -dx)
    const T a12x = mc * dy + ms * dx, a12y = ms * dy - mc * dx;
    A[0][0] = -mc; A[0][1] = ms;  A[0][2] = a12x;
    A[1][0] = -ms; A[1][1] = -mc; A[1][2] = a12y;
    A[2][0] = 0;   A[2][1] = 0;   A[2][2] = -1;
    B[0][0] = mc;  B[0][1] = -ms; B[0][2] = 0;
    B[1][0] = ms;  B[1][1] = mc;  B[1][2] = 0;
    B[2][0] = 0;   B[2][1] = 0;   B[2][2] = 1;
  } else {
    const T r = x1.z, i = x1.w;
    e[0] = (r * dx + i * dy) - z.x;
    e[1] = (-i * dx + r * dy) - z.y;
    e[2] = 0;
    A[0][0] = -r; A[0][1] = -i; A[0][2] = -i * dx + r * dy;
    A[1][0] = i;  A[1][1] = -r; A[1][2] = -r * dx - i * dy;
    A[2][0] = 0;  A[2][1] = 0;  A[2][2] = 0;
    B[0][0] = r;  B[0][1] = i;  B[0][2] = 0;
    B[1][0] = -i; B[1][1] = r;  B[1][2] = 0;
    B[2][0] = 0;  B[2][1] = 0;  B[2][2] = 0;
  }
}

template <typename T> __device__ __forceinline__ T group_sum8(T v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

template <typename T, int THREADS> __device__ __forceinline__ T block_sum(T v, T *scratch) {
  // wave reduction then one value per wave through LDS; result valid in thread 0
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < THREADS / 64; w++) r += scratch[w];
  return r;
}

// One group of LIN_GROUP lanes per node pulls the node's incident edges
// (deterministic: no atomics, fixed summation order), builds the node's
// diagonal block and right-hand side; the lane holding an edge in its `from`
// role also writes the off-diagonal block and the edge's chi2 term.
template <typename T>
__global__ void __launch_bounds__(LIN_THREADS) k_linearize(LinArgs<T> a) {
  using V4 = typename VecT<T>::V4;
  using V2 = typename VecT<T>::V2;
  __shared__ double red[LIN_THREADS / 64];
  const int gid = blockIdx.x * LIN_THREADS + threadIdx.x;
  const int node = gid / LIN_GROUP, sub = gid % LIN_GROUP;
  double chi = 0.0;
  T hd[6] = {0, 0, 0, 0, 0, 0};  // 00 10 11 20 21 22
  T bv[3] = {0, 0, 0};
  int nd = 0;
  if (node < a.n_nodes) {
    nd = a.node_dim[node];
    const V4 self = a.pose[node];
    const int q1 = a.inc_ptr[node + 1];
    for (int q = a.inc_ptr[node] + sub; q < q1; q += LIN_GROUP) {
      const int ent = a.inc_list[q];
      const int k = ent >> 2, kind = (ent >> 1) & 1, role = ent & 1;
      const int2 ft = a.e_idx[k];
      const V4 other = a.pose[role ? ft.x : ft.y];
      const V4 z = a.e_meas[k];
      const V4 wa = a.e_info_a[k];
      const V2 wb = a.e_info_b[k];
      const T W[3][3] = {{wa.x, wa.y, wa.z}, {wa.y, wa.w, wb.x}, {wa.z, wb.x, wb.y}};
      T e[3], A[3][3], B[3][3];
      edge_linearize_2d<T>(kind, role ? other : self, role ? self : other, z, e, A, B);
      // J = A (from role) or B (to role);  JW = J^T W
      T JW[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          T s = 0;
#pragma unroll
          for (int r = 0; r < 3; r++) s += (role ? B[r][i] : A[r][i]) * W[r][j];
          JW[i][j] = s;
        }
      if (a.write_system) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) {
            T s = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) s += JW[i][r] * (role ? B[r][j] : A[r][j]);
            hd[t++] += s;
          }
#pragma unroll
        for (int i = 0; i < 3; i++) bv[i] += JW[i][0] * e[0] + JW[i][1] * e[1] + JW[i][2] * e[2];
      }
      if (role == 0) {
        // chi2 term e^T W e (:555,568), accumulated in f64
        T we0 = W[0][0] * e[0] + W[0][1] * e[1] + W[0][2] * e[2];
        T we1 = W[1][0] * e[0] + W[1][1] * e[1] + W[1][2] * e[2];
        T we2 = W[2][0] * e[0] + W[2][1] * e[1] + W[2][2] * e[2];
        chi += (double)(e[0] * we0 + e[1] * we1 + e[2] * we2);
        if (a.write_system) {
          // off-diagonal block H[from rows, to cols] = A^T W B
          const int64_t so = a.e_slot[k];
          T *dst = a.hvals + (so >> 1);
          const bool tr = so & 1;
          const int d2 = kind ? 2 : 3;
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              if (j >= d2) continue;
              T s = JW[i][0] * B[0][j] + JW[i][1] * B[1][j] + JW[i][2] * B[2][j];
              dst[tr ? j * 3 + i : i * d2 + j] = s;
            }
        }
      }
    }
  }
  if (a.write_system) {
#pragma unroll
    for (int t = 0; t < 6; t++) hd[t] = group_sum8(hd[t]);
#pragma unroll
    for (int t = 0; t < 3; t++) bv[t] = group_sum8(bv[t]);
    if (node < a.n_nodes && sub == 0) {
      T add = a.lambda;
      if (node == a.anchor) add += (T)10000000.0;
      T *d = a.hvals + a.diag_off[node];
      if (nd == 3) {
        d[0] = hd[0] + add; d[1] = hd[1];       d[2] = hd[3];
        d[3] = hd[1];       d[4] = hd[2] + add; d[5] = hd[4];
        d[6] = hd[3];       d[7] = hd[4];       d[8] = hd[5] + add;
      } else {
        d[0] = hd[0] + add; d[1] = hd[1];
        d[2] = hd[1];       d[3] = hd[2] + add;
      }
      T *bo = a.b + a.node_offset[node];
      for (int t = 0; t < nd; t++) bo[t] = -bv[t];
    }
  }
  double tot = block_sum<double, LIN_THREADS>(chi, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = tot;
}

// ------------------------------------------------------------ multifrontal

template <typename T> struct FactorArgs {
  // schedule
  const int32_t *task_ptr, *task_sn;
  int task_begin;
  // supernodes
  const int32_t *sn_ncols, *sn_nrows, *sn_col0;
  const int64_t *sn_loff, *sn_uoff;
  const int32_t *sn_uld;
  const int64_t *asm_ptr;
  const AsmItemDev *asm_items;
  const int32_t *child_ptr, *child_list;
  const int64_t *rel_ptr;
  const int32_t *rel;
  const int32_t *perm;      // permuted scalar -> reference scalar
  const int64_t *sn_rows_ptr;
  const int32_t *sn_rows;
  // numeric
  const T *hvals;
  const T *b;
  T *lvals;                 // factor panels
  T *uvals;                 // update matrices (packed)
  T *x;                     // solution, permuted order
  int *err;
};

// update-matrix element (i >= j) of an n x n lower triangle: packed columns, or
// a plain column-major square when ld > 0
__device__ __forceinline__ int64_t tri_index(int n, int ld, int i, int j) {
  return ld > 0 ? (int64_t)j * ld + i : (int64_t)j * n - (int64_t)j * (j - 1) / 2 + (i - j);
}

// Workgroup-wide partial Cholesky of the leading nc columns of the M x nc panel
// P (column-major, ld M; rows nc.. are the off-diagonal rows and the rhs row),
// blocked by NB columns: diagonal block in the registers of the first wave
// (row per lane, columns exchanged with wave shuffles), triangular solve with
// one thread per row, rank-NB update of the remaining panel columns.
template <typename T, int THREADS, int NB>
__device__ void panel_factor(T *P, int M, int nc, int *err) {
  const int tid = threadIdx.x;
  for (int k0 = 0; k0 < nc; k0 += NB) {
    const int nb = min(NB, nc - k0);
    if (tid < 64) {
      T r[NB];
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NB; k++)
        r[k] = (tid < nb && k <= tid) ? P[(int64_t)(k0 + k) * M + k0 + tid] : (T)0;
#pragma unroll
      for (int k = 0; k < NB; k++) {
        T d = __shfl(r[k], k);
        if (k < nb && !(d > (T)0)) bad = true;
        if (!(d > (T)0)) d = (T)1;
        const T sq = sqrt(d);
        const T lik = tid > k ? r[k] / sq : (tid == k ? sq : (T)0);
        r[k] = lik;
#pragma unroll
        for (int j = k + 1; j < NB; j++) {
          const T ljk = __shfl(lik, j);
          if (tid >= j) r[j] -= lik * ljk;
        }
      }
#pragma unroll
      for (int k = 0; k < NB; k++)
        if (tid < nb && k <= tid) P[(int64_t)(k0 + k) * M + k0 + tid] = r[k];
      if (bad && tid == 0) atomicOr(err, DEVERR_NOT_SPD);
    }
    __syncthreads();
    // rows below the diagonal block: x * L11^T = row  (forward substitution per row)
    for (int i = k0 + nb + tid; i < M; i += THREADS) {
      T xr[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        if (k < nb) {
          T s = P[(int64_t)(k0 + k) * M + i];
#pragma unroll
          for (int q = 0; q < NB; q++)
            if (q < k) s -= xr[q] * P[(int64_t)(k0 + q) * M + k0 + k];
          xr[k] = s / P[(int64_t)(k0 + k) * M + k0 + k];
          P[(int64_t)(k0 + k) * M + i] = xr[k];
        }
      }
    }
    __syncthreads();
    // update the remaining pivot columns j in [k0+nb, nc), rows i >= j
    const int j0 = k0 + nb;
    const int ncols_left = nc - j0;
    if (ncols_left > 0) {
      const int rows_left = M - j0;
      const int total = ncols_left * rows_left;
      for (int t = tid; t < total; t += THREADS) {
        const int jj = t / rows_left, ii = t - jj * rows_left;
        const int j = j0 + jj, i = j0 + ii;
        if (i < j) continue;
        T s = 0;
#pragma unroll
        for (int k = 0; k < NB; k++)
          if (k < nb) s += P[(int64_t)(k0 + k) * M + i] * P[(int64_t)(k0 + k) * M + j];
        P[(int64_t)j * M + i] -= s;
      }
      __syncthreads();
    }
  }
}

// Assemble, factor and publish one front.  P is the M x nc pivot panel
// (column-major, ld M), U the (nr+1) x (nr+1) update matrix (packed lower when
// uld == 0, column-major with leading dimension uld otherwise).  For fronts
// held in LDS both are copied to global memory at the end; a front that lives
// in global memory (IN_PLACE) is already where it has to be.
template <typename T, int THREADS, int MAXD2, bool IN_PLACE>
__device__ void process_front(const FactorArgs<T> &a, int s, T *P, T *U, int uld) {
  const int tid = threadIdx.x;
  const int nc = a.sn_ncols[s], nr = a.sn_nrows[s];
  const int M = nc + nr + 1, nu = nr + 1;
  const int64_t psize = (int64_t)M * nc;
  if (IN_PLACE) {
    for (int64_t t = tid; t < (int64_t)M * M; t += THREADS) P[t] = 0;  // whole front, ld M
  } else {
    const int64_t usize = (int64_t)nu * (nu + 1) / 2;
    for (int64_t t = tid; t < psize; t += THREADS) P[t] = 0;
    for (int64_t t = tid; t < usize; t += THREADS) U[t] = 0;
  }
  __syncthreads();
  // ---- original entries of H that live in this front's pivot columns
  const int64_t i0 = a.asm_ptr[s], i1 = a.asm_ptr[s + 1];
  for (int64_t t = tid; t < (i1 - i0) * MAXD2; t += THREADS) {
    const AsmItemDev it = a.asm_items[i0 + t / MAXD2];
    const int e = (int)(t % MAXD2);
    const int dr = it.drow, dc = it.dcol;
    if (e >= dr * dc || it.diag == 2) continue;
    const int i = e / dc, j = e - i * dc;
    if (it.diag == 1 && i < j) continue;
    P[(int64_t)(it.lcol + j) * M + it.lrow + i] = a.hvals[it.src + e];
  }
  const int c0 = a.sn_col0[s];
  for (int j = tid; j < nc; j += THREADS) P[(int64_t)j * M + (M - 1)] = a.b[a.perm[c0 + j]];
  __syncthreads();
  if (tid == 0)  // blocks of duplicated edges (rare): serial, fixed order
    for (int64_t q = i0; q < i1; q++) {
      const AsmItemDev it = a.asm_items[q];
      if (it.diag != 2) continue;
      for (int i = 0; i < it.drow; i++)
        for (int j = 0; j < it.dcol; j++)
          P[(int64_t)(it.lcol + j) * M + it.lrow + i] += a.hvals[it.src + i * it.dcol + j];
    }
  // ---- extend-add of the children's update matrices, fixed child order
  for (int q = a.child_ptr[s]; q < a.child_ptr[s + 1]; q++) {
    const int c = a.child_list[q];
    const int ncu = a.sn_nrows[c] + 1;
    const int cld = a.sn_uld[c];
    const T *Uc = (cld > 0 ? a.lvals : a.uvals) + a.sn_uoff[c];
    const int32_t *rel = a.rel + a.rel_ptr[c];
    __syncthreads();
    for (int t = tid; t < ncu * ncu; t += THREADS) {
      const int j = t / ncu, i = t - j * ncu;
      if (i < j || t == ncu * ncu - 1) continue;  // lower triangle; (rhs, rhs) corner is never used
      const T v = Uc[tri_index(ncu, cld, i, j)];
      const int li = rel[i], lj = rel[j];
      if (lj < nc) P[(int64_t)lj * M + li] += v;
      else U[tri_index(nu, uld, li - nc, lj - nc)] += v;
    }
  }
  __syncthreads();
  // ---- partial factorisation + Schur complement
  panel_factor<T, THREADS, 8>(P, M, nc, a.err);
  {
    // U(i,j) -= sum_k P[nc+i][k] P[nc+j][k]   (i >= j); 2 x 2 register tiles
    const int nt = (nu + 1) / 2;
    for (int t = tid; t < nt * nt; t += THREADS) {
      const int tj = t / nt, ti = t - tj * nt;
      if (ti < tj) continue;
      const int i0r = 2 * ti, j0r = 2 * tj;
      const bool i1ok = i0r + 1 < nu, j1ok = j0r + 1 < nu;
      T s00 = 0, s10 = 0, s01 = 0, s11 = 0;
      const T *pi = P + nc + i0r, *pj = P + nc + j0r;
      for (int k = 0; k < nc; k++) {
        const T a0 = pi[(int64_t)k * M], a1 = i1ok ? pi[(int64_t)k * M + 1] : (T)0;
        const T b0 = pj[(int64_t)k * M], b1 = j1ok ? pj[(int64_t)k * M + 1] : (T)0;
        s00 += a0 * b0; s10 += a1 * b0; s01 += a0 * b1; s11 += a1 * b1;
      }
      U[tri_index(nu, uld, i0r, j0r)] -= s00;
      if (i1ok) U[tri_index(nu, uld, i0r + 1, j0r)] -= s10;
      if (j1ok && ti > tj) U[tri_index(nu, uld, i0r, j0r + 1)] -= s01;
      if (i1ok && j1ok) U[tri_index(nu, uld, i0r + 1, j0r + 1)] -= s11;
    }
  }
  __syncthreads();
  if (!IN_PLACE) {
    const int64_t usize = (int64_t)nu * (nu + 1) / 2;
    T *Lg = a.lvals + a.sn_loff[s];
    for (int64_t t = tid; t < psize; t += THREADS) Lg[t] = P[t];
    T *Ug = a.uvals + a.sn_uoff[s];
    for (int64_t t = tid; t < usize; t += THREADS) Ug[t] = U[t];
    __syncthreads();
  }
}

// One workgroup per task; a task is a list of supernodes in elimination order
// whose fronts are assembled, factored and pushed to global memory one after
// the other, entirely out of LDS.
template <typename T, int THREADS, int MAXD2>
__global__ void __launch_bounds__(THREADS) k_factor_tasks(FactorArgs<T> a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int task = a.task_begin + blockIdx.x;
  for (int si = a.task_ptr[task]; si < a.task_ptr[task + 1]; si++) {
    const int s = a.task_sn[si];
    const int nc = a.sn_ncols[s], nr = a.sn_nrows[s];
    process_front<T, THREADS, MAXD2, false>(a, s, smem, smem + (int64_t)(nc + nr + 1) * nc, 0);
  }
}

// Fallback for a front that does not fit in LDS: one workgroup works on the
// front in place in global memory (L storage holds the whole M x M front).
template <typename T, int THREADS, int MAXD2>
__global__ void __launch_bounds__(THREADS) k_factor_big_single(FactorArgs<T> a, int s) {
  T *F = a.lvals + a.sn_loff[s];
  const int M = a.sn_ncols[s] + a.sn_nrows[s] + 1;
  process_front<T, THREADS, MAXD2, true>(a, s, F, a.lvals + a.sn_uoff[s], M);
}

// Back substitution for one supernode:
//   x1 = L11^-T ( y1 - L21^T x[rows] ),  y1 = the rhs row of the factored panel.
// work: LDS scratch of nr + nc scalars (+ nc*nc when STAGE_L11).
template <typename T, int THREADS, bool STAGE_L11>
__device__ void solve_front(const FactorArgs<T> &a, int s, T *work) {
  const int tid = threadIdx.x;
  const int nc = a.sn_ncols[s], nr = a.sn_nrows[s];
  const int M = nc + nr + 1;
  const T *Lg = a.lvals + a.sn_loff[s];
  T *x2 = work;        // nr
  T *t1 = work + nr;   // nc
  T *L11 = t1 + nc;    // nc x nc, ld nc (STAGE_L11 only)
  const int32_t *rows = a.sn_rows + a.sn_rows_ptr[s];
  __syncthreads();
  for (int i = tid; i < nr; i += THREADS) x2[i] = a.x[rows[i]];
  if (STAGE_L11)
    for (int t = tid; t < nc * nc; t += THREADS) {
      const int j = t / nc, i = t - j * nc;
      L11[t] = Lg[(int64_t)j * M + i];
    }
  __syncthreads();
  // t1[j] = y1[j] - sum_i L21[i][j] x2[i] ; one wave per column, lanes over rows
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < nc; j += THREADS / 64) {
      const T *col = Lg + (int64_t)j * M + nc;
      T sacc = 0;
      for (int i = lane; i < nr; i += 64) sacc += col[i] * x2[i];
      for (int o = 32; o > 0; o >>= 1) sacc += __shfl_down(sacc, o);
      if (lane == 0) t1[j] = col[nr] - sacc;
    }
  }
  __syncthreads();
  if (STAGE_L11) {
    // L11^T x = t, backward, in the registers of the first wave: lane l owns
    // entries l, l+64, l+128, l+192 (nc <= 256 on this path)
    if (tid < 64) {
      T tt[4];
#pragma unroll
      for (int q = 0; q < 4; q++) tt[q] = (tid + 64 * q) < nc ? t1[tid + 64 * q] : (T)0;
      for (int j = nc - 1; j >= 0; j--) {
        const int owner = j & 63, slot = j >> 6;
        T mine = slot == 0 ? tt[0] : slot == 1 ? tt[1] : slot == 2 ? tt[2] : tt[3];
        const T xj = __shfl(mine, owner) / L11[j * nc + j];
        if (tid == owner) {
          if (slot == 0) tt[0] = xj; else if (slot == 1) tt[1] = xj; else if (slot == 2) tt[2] = xj; else tt[3] = xj;
        }
        // t[i] -= L(j,i) * x_j for i < j  (row j of L11: stride nc)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = tid + 64 * q;
          if (i < j) tt[q] -= L11[i * nc + j] * xj;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (tid + 64 * q < nc) t1[tid + 64 * q] = tt[q];
    }
  } else {
    // big fronts: column dot products straight from global memory, first wave
    if (tid < 64) {
      for (int j = nc - 1; j >= 0; j--) {
        const T *col = Lg + (int64_t)j * M;
        T sacc = 0;
        for (int i = j + 1 + tid; i < nc; i += 64) sacc += col[i] * t1[i];
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_down(sacc, o);
        if (tid == 0) t1[j] = (t1[j] - sacc) / col[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  const int c0 = a.sn_col0[s];
  for (int j = tid; j < nc; j += THREADS) a.x[c0 + j] = t1[j];
  __syncthreads();
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_solve_tasks(FactorArgs<T> a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int task = a.task_begin + blockIdx.x;
  for (int si = a.task_ptr[task + 1] - 1; si >= a.task_ptr[task]; si--)
    solve_front<T, THREADS, true>(a, a.task_sn[si], smem);
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_solve_big_single(FactorArgs<T> a, int s) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  solve_front<T, THREADS, false>(a, s, reinterpret_cast<T *>(smem_raw));
}

// ------------------------------------------------------------------ update

template <typename T> struct UpdArgs {
  int n_nodes;
  typename VecT<T>::V4 *pose;
  const uint8_t *node_dim;
  const int32_t *node_pcol, *node_offset;
  const T *x;          // permuted solution (used when dx_ref_in == nullptr)
  const T *dx_ref_in;  // reference-order step supplied by the caller (rr_pgo_update)
  T *dx_ref_out;       // reference-order copy of the applied step (may be null)
  T sign;
  double *norm_partial;
};

template <typename T>
__global__ void __launch_bounds__(UPD_THREADS) k_update(UpdArgs<T> a) {
  __shared__ double red[UPD_THREADS / 64];
  const int node = blockIdx.x * UPD_THREADS + threadIdx.x;
  double nrm = 0.0;
  if (node < a.n_nodes) {
    const int nd = a.node_dim[node];
    T d[3] = {0, 0, 0};
    if (a.dx_ref_in) {
      const T *src = a.dx_ref_in + a.node_offset[node];
      for (int t = 0; t < nd; t++) d[t] = src[t];
    } else {
      const T *src = a.x + a.node_pcol[node];
      for (int t = 0; t < nd; t++) d[t] = src[t];
    }
    if (a.dx_ref_out) {
      T *dst = a.dx_ref_out + a.node_offset[node];
      for (int t = 0; t < nd; t++) dst[t] = d[t];
    }
    for (int t = 0; t < nd; t++) nrm += (double)d[t] * (double)d[t];
    auto p = a.pose[node];
    p.x += a.sign * d[0];
    p.y += a.sign * d[1];
    if (nd == 3) {  // rotation *= UnitComplex::from_angle(dtheta), no renormalisation (:236)
      const T c = cos(a.sign * d[2]), s = sin(a.sign * d[2]);
      const T re = p.z * c - p.w * s, im = p.z * s + p.w * c;
      p.z = re;
      p.w = im;
    }
    a.pose[node] = p;
  }
  double tot = block_sum<double, UPD_THREADS>(nrm, red);
  if (threadIdx.x == 0) a.norm_partial[blockIdx.x] = tot;
}

// scalars[0] = chi2 (sum of n_chi partials), scalars[1] = |dx| (sqrt of sum of n_norm partials)
__global__ void __launch_bounds__(256) k_finalize(const double *chi_partial, int n_chi,
                                                  const double *norm_partial, int n_norm,
                                                  double *scalars) {
  __shared__ double red[4];
  double c = 0.0, n = 0.0;
  // fixed order: each thread a strided slice, then the block tree
  for (int i = threadIdx.x; i < n_chi; i += 256) c += chi_partial[i];
  for (int i = threadIdx.x; i < n_norm; i += 256) n += norm_partial[i];
  double ct = block_sum<double, 256>(c, red);
  double nt = block_sum<double, 256>(n, red);
  if (threadIdx.x == 0) {
    if (n_chi > 0) scalars[0] = ct;
    if (n_norm > 0) scalars[1] = sqrt(nt);
  }
}

}  // namespace rrpgo
