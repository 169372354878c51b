// pose_graph.hip -- loop-closure pose graph on gfx950 (MI355X), FP64.
// Replaces the solve inside myslam::Optimizer::solvePoseGraphLoop (reference
// src/optimizer_ceres.cpp:1036-1305; cost functor PoseGraphLoop, include/myslam/optimizer_ceres.h:269-325)
// and the map-point re-anchoring that follows it (:1281-1301).
//
//   k_pg_linearize  one wavefront per key-frame: residuals and closed-form tangent Jacobians of its
//                   incident edges, its 6x6 diagonal block, gradient and one 6x6 off-diagonal block per
//                   neighbour, gathered in adjacency order (deterministic, no atomics)
//   k_pg_damp       A = S H S + D(radius) (lower triangle), rhs = S g        (Jacobi scaling S)
//   vo::chol_factor_solve (csrc/chol.hip)  dense tile Cholesky + both substitutions of the
//                   6(N-1) x 6(N-1) system in one persistent dataflow kernel on the FP64 matrix cores
//   k_pg_model      model cost change  -(g''.s + s^T H'' s / 2)
//   k_pg_candidate  x (+) delta  (EigenQuaternionParameterization::Plus, additive translation)
//   k_pg_cost       sum of squared residuals over the edges
// The trust-region bookkeeping (a handful of scalars per iteration) runs on the host: the routine is
// called once per loop closure.
#include "ba_math.h"
#include "vo_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace vo;

constexpr int NB = vo::kCholPanel;  // Cholesky panel width

// ---------------------------------------------------------------- quaternion helpers (x, y, z, w)
__host__ __device__ __forceinline__ void q_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__host__ __device__ __forceinline__ void q_rot(const double q[4], const double v[3], double o[3]) {  // Eigen _transformVector
  const double uv[3] = {2 * (q[1] * v[2] - q[2] * v[1]), 2 * (q[2] * v[0] - q[0] * v[2]), 2 * (q[0] * v[1] - q[1] * v[0])};
  o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__host__ __device__ __forceinline__ void q_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - z * w), R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w), R[7] = 2 * (y * z + x * w), R[8] = 1 - 2 * (x * x + y * y);
}
__host__ __device__ __forceinline__ void m3(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// PoseGraphLoop residual of one edge (optimizer_ceres.h:281-312) and its tangent Jacobians
// (7 x 6 row-major per node: [rotation delta of EigenQuaternionParameterization, translation])
template <bool WANT_J>
__device__ void pg_edge(const double *x1, double s1, const double *x2, double s2, const double qm[4], const double tm[3],
                        double sm, double r[7], double *J1, double *J2) {
  const double *q1 = x1, *t1 = x1 + 4, *q2 = x2, *t2 = x2 + 4;
  const double q2i[4] = {-q2[0], -q2[1], -q2[2], q2[3]};
  double q12[4], dq[4], w[3], rq[3], rt[3];
  q_mul(q1, q2i, q12);
  q_mul(qm, q12, dq);
  const double t2s[3] = {(1.0 / s2) * t2[0], (1.0 / s2) * t2[1], (1.0 / s2) * t2[2]};
  q_rot(q2i, t2s, w);
  const double t2inv[3] = {-w[0], -w[1], -w[2]};
  q_rot(q1, t2inv, rq);
  const double t12[3] = {s1 * rq[0] + t1[0], s1 * rq[1] + t1[1], s1 * rq[2] + t1[2]};
  q_rot(qm, t12, rt);
  for (int k = 0; k < 3; k++) r[k] = 2.0 * dq[k], r[3 + k] = sm * rt[k] + tm[k];
  r[6] = sm * s1 * (1.0 / s2);
  if (!WANT_J) return;
  double Rm[9], R1[9], R2[9];
  q_to_R(qm, Rm), q_to_R(q1, R1), q_to_R(q2, R2);
  for (int i = 0; i < 42; i++) J1[i] = 0, J2[i] = 0;
  for (int a = 0; a < 3; a++) {
    double e[4] = {0, 0, 0, 0}, tmp[4], col[4];
    e[a] = 1;
    q_mul(qm, e, tmp);
    q_mul(tmp, q12, col);  // d(2 vec(qm (1 + d^) q12)) / d d_a
    for (int k = 0; k < 3; k++) J1[6 * k + a] = 2.0 * col[k];
    q_mul(dq, e, col);     // q2'^-1 = q2^-1 (1 - d^)
    for (int k = 0; k < 3; k++) J2[6 * k + a] = -2.0 * col[k];
  }
  {
    const double X[9] = {0, -rq[2], rq[1], rq[2], 0, -rq[0], -rq[1], rq[0], 0};
    double M[9];
    m3(Rm, X, M);  // d r_t / d delta1 = sm Rm s1 (-2 [R1 t2inv]x),  d r_t / d t1 = sm Rm
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) {
        J1[6 * (3 + k) + a] = -2.0 * sm * s1 * M[3 * k + a];
        J1[6 * (3 + k) + 3 + a] = sm * Rm[3 * k + a];
      }
  }
  {
    const double X[9] = {0, -t2s[2], t2s[1], t2s[2], 0, -t2s[0], -t2s[1], t2s[0], 0};
    const double R2t[9] = {R2[0], R2[3], R2[6], R2[1], R2[4], R2[7], R2[2], R2[5], R2[8]};
    double A[9], Bm[9], Cm[9];
    m3(Rm, R1, A);
    m3(A, R2t, Bm);  // Rm R1 R2^T
    m3(Bm, X, Cm);   // d r_t / d delta2 = -2 sm s1 Rm R1 R2^T [t2/s2]x,  d r_t / d t2 = -sm s1 / s2 Rm R1 R2^T
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) {
        J2[6 * (3 + k) + a] = -2.0 * sm * s1 * Cm[3 * k + a];
        J2[6 * (3 + k) + 3 + a] = -sm * s1 * (1.0 / s2) * Bm[3 * k + a];
      }
  }
}

struct PgDev {
  int n_nodes, n_edges, n;  // n = 6 * free nodes
  int ld;                   // padded leading dimension of the dense matrices (multiple of NB)
  const int *e_i, *e_j;
  const double *qm, *tm, *sm, *scales;
  const int *slot;                     // node -> free index or -1
  const int *adj_start, *adj_edge;     // CSR: incident edges of every node, in edge order
  double *H, *g;                       // unscaled normal equations (full symmetric blocks written)
  double *colscale;                    // Jacobi scale
};

// one wavefront per node
__global__ __launch_bounds__(64) void k_pg_linearize(PgDev P, const double *x, int first) {
  __shared__ double Ja[42], Jb[42], r[7];
  const int a = blockIdx.x, lane = threadIdx.x;
  const int sa = P.slot[a];
  if (sa < 0) return;
  double haa = 0, ga = 0;  // lane < 36: entry (p, q) of the diagonal block; lane < 6: gradient entry
  const int p = lane / 6, q = lane - 6 * p;
  // zero this node's block row (the matrix is rebuilt every linearisation)
  for (int i = lane; i < 6 * P.n; i += 64) {
    const int rr = i / P.n, cc = i - rr * P.n;
    P.H[(long long)(6 * sa + rr) * P.ld + cc] = 0.0;
  }
  __syncthreads();
  for (int k = P.adj_start[a]; k < P.adj_start[a + 1]; k++) {
    const int e = P.adj_edge[k];
    const int i = P.e_i[e], j = P.e_j[e];
    const bool first_side = (i == a);
    const int b = first_side ? j : i;
    if (lane == 0) {
      double J1[42], J2[42], rr[7];
      pg_edge<true>(x + 7 * i, P.scales[i], x + 7 * j, P.scales[j], P.qm + 4 * e, P.tm + 3 * e, P.sm[e], rr, J1, J2);
      for (int t = 0; t < 42; t++) Ja[t] = first_side ? J1[t] : J2[t], Jb[t] = first_side ? J2[t] : J1[t];
      for (int t = 0; t < 7; t++) r[t] = rr[t];
    }
    __syncthreads();
    if (lane < 36) {
      double s = 0, sb = 0;
      for (int t = 0; t < 7; t++) s += Ja[6 * t + p] * Ja[6 * t + q], sb += Ja[6 * t + p] * Jb[6 * t + q];
      haa += s;
      const int sbn = P.slot[b];
      if (sbn >= 0 && b != a) P.H[(long long)(6 * sa + p) * P.ld + 6 * sbn + q] += sb;  // this block row is ours alone
    }
    if (lane < 6) {
      double s = 0;
      for (int t = 0; t < 7; t++) s += Ja[6 * t + lane] * r[t];
      ga += s;
    }
    __syncthreads();
  }
  if (lane < 36) P.H[(long long)(6 * sa + p) * P.ld + 6 * sa + q] = haa;
  if (lane < 6) P.g[6 * sa + lane] = ga;
  if (first && lane < 36 && p == q) P.colscale[6 * sa + p] = 1.0 / (1.0 + sqrt(haa));  // 1 / (1 + ||column||)
}

// A = S H S + D, D = clamp(diag(S H S), 1e-6, 1e32) / radius (lower triangle incl. padding), rhs = S g
__global__ __launch_bounds__(256) void k_pg_damp(PgDev P, double *A, double *rhs, double radius) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(idx / P.ld), j = (int)(idx - (long long)i * P.ld);
  if (i >= P.ld) return;
  double v = 0;
  if (i < P.n && j < P.n) {
    v = P.H[idx] * P.colscale[i] * P.colscale[j];
    if (i == j) v += fmin(fmax(v, 1e-6), 1e32) / radius;
  } else if (i == j) {
    v = 1.0;  // padding: identity keeps the factorisation well defined
  }
  A[idx] = v;
  if (j == 0) A[(long long)P.ld * P.ld + i] = i < P.n ? P.g[i] * P.colscale[i] : 0.0;  // right-hand side: row ld
  (void)rhs;
}

// model cost change of step s = -y:  -(g''.s + s^T H'' s / 2), H'' = S H S (undamped); out[0] += partial
__global__ __launch_bounds__(256) void k_pg_model(PgDev P, const double *y, double *partial) {
  __shared__ double red[4];
  const int i = blockIdx.x;  // one row per workgroup
  double acc = 0;
  for (int j = threadIdx.x; j < P.n; j += 256) acc += P.H[(long long)i * P.ld + j] * P.colscale[j] * y[j];
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double Hy = ((red[0] + red[1]) + red[2]) + red[3];  // (H S y)_i
    const double si = P.colscale[i], yi = y[i];
    // s = -y:  g''.s = -g''_i y_i ;  s^T H'' s = y_i s_i (H S y)_i
    partial[i] = -(-(P.g[i] * si) * yi + 0.5 * yi * si * Hy);
  }
}

__global__ void k_pg_candidate(PgDev P, const double *x, const double *y, double *xc, double *norms /*x^2, step^2 partial per node*/) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= P.n_nodes) return;
  const int s = P.slot[a];
  double d[6] = {0, 0, 0, 0, 0, 0};
  if (s >= 0)
    for (int k = 0; k < 6; k++) d[k] = -y[6 * s + k] * P.colscale[6 * s + k];
  const double *q = x + 7 * a;
  double qn[4] = {q[0], q[1], q[2], q[3]};
  const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nrm > 0.0) {  // ceres::EigenQuaternionParameterization::Plus
    const double sn = sin(nrm) / nrm;
    const double dq[4] = {sn * d[0], sn * d[1], sn * d[2], cos(nrm)};
    q_mul(dq, q, qn);
  }
  double xn2 = 0, st2 = 0;
  for (int k = 0; k < 4; k++) {
    xc[7 * a + k] = qn[k];
    xn2 += q[k] * q[k], st2 += (qn[k] - q[k]) * (qn[k] - q[k]);
  }
  for (int k = 0; k < 3; k++) {
    const double tn = x[7 * a + 4 + k] + d[3 + k];
    xc[7 * a + 4 + k] = tn;
    xn2 += x[7 * a + 4 + k] * x[7 * a + 4 + k], st2 += d[3 + k] * d[3 + k];
  }
  norms[2 * a] = s >= 0 ? xn2 : 0.0, norms[2 * a + 1] = s >= 0 ? st2 : 0.0;
}

__global__ __launch_bounds__(256) void k_pg_cost(PgDev P, const double *x, double *partial) {
  __shared__ double red[4];
  const int e = blockIdx.x * 256 + threadIdx.x;
  double c = 0;
  if (e < P.n_edges) {
    const int i = P.e_i[e], j = P.e_j[e];
    double r[7];
    pg_edge<false>(x + 7 * i, P.scales[i], x + 7 * j, P.scales[j], P.qm + 4 * e, P.tm + 3 * e, P.sm[e], r, nullptr, nullptr);
    for (int k = 0; k < 7; k++) c += 0.5 * r[k] * r[k];
  }
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// corrected = Swr * (Srw * p): optimizer_ceres.cpp:1281-1301; Sim3 as (s, q, t): S p = s R(q) p + t
__global__ void k_sim3_reanchor(int n, const double *pts_in, const int *ref, const double *Srw /*8 per node: q t s*/,
                                const double *Swr, double *pts_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = ref[i];
  double p[3] = {pts_in[3 * i], pts_in[3 * i + 1], pts_in[3 * i + 2]}, a[3], b[3];
  if (k >= 0) {
    const double *S1 = Srw + 8 * k, *S2 = Swr + 8 * k;
    q_rot(S1, p, a);
    for (int c = 0; c < 3; c++) a[c] = S1[7] * a[c] + S1[4 + c];
    q_rot(S2, a, b);
    for (int c = 0; c < 3; c++) p[c] = S2[7] * b[c] + S2[4 + c];
  }
  pts_out[3 * i] = p[0], pts_out[3 * i + 1] = p[1], pts_out[3 * i + 2] = p[2];
}

// All entry points of this file run on the calling host thread's stream with that thread's grow-only scratch
// (vo_common.h): no hipMalloc / hipFree / hipDeviceSynchronize per call, nothing on the NULL stream.
int upload(vo::DevBuf &b, const void *src, size_t bytes, hipStream_t st) {
  return vo::upload(b, src, bytes, st, "pose graph");
}

int sum_device(const vo::DevBuf &b, int n, hipStream_t st, double &out, int stride = 1, int offset = 0) {  // fixed order, on the host
  std::vector<double> h((size_t)n * stride);
  VO_CHECK(vo::copy_d2h(h.data(), b.p, h.size() * 8, st, "pose graph"));
  VO_CHECK(vo::stream_sync(st, "pose graph"));
  double s = 0;
  for (int i = 0; i < n; i++) s += h[(size_t)i * stride + offset];
  out = s;
  return VO_OK;
}

}  // namespace

extern "C" {

// the 64 x 64 tiles that hold a non-zero: a matrix with empty tiles is factored on its sparse plan (the plans of the
// large reduced camera systems go through the same code; this entry point is how the tests reach them).  NULL = dense.
static bool pattern_of_matrix(const std::vector<double> &Ap, int ld, std::vector<unsigned long long> &pattern) {
  const int m = ld / NB;
  pattern.assign((size_t)m, 0ull);
  bool dense = true;
  for (int ti = 0; ti < m; ti++)
    for (int tj = 0; tj <= ti; tj++) {
      bool nz = false;
      for (int r = NB * ti; r < NB * ti + NB && !nz; r++)
        for (int c = NB * tj; c < NB * tj + NB; c++)
          if (Ap[(size_t)r * ld + c] != 0.0) {
            nz = true;
            break;
          }
      if (nz) pattern[ti] |= 1ull << tj;
      else dense = false;
    }
  return dense;
}
static vo::CholPlan *plan_of_matrix(const std::vector<double> &Ap, int ld) {
  std::vector<unsigned long long> pattern;
  return pattern_of_matrix(Ap, ld, pattern) ? nullptr : vo::chol_plan_create(ld / NB, pattern.data());
}

// Test entry of the split (per-rank segment) solve, csrc/chol.hip: the n_ranks shards are emulated one after the other on
// this GPU.  The tile columns [0, c0) hold independent segments -- col_part[j] = the segment of tile column j, owned by rank
// col_part[j] % n_ranks --, the rest the separators.  Every "rank" gets the segment columns it owns; the separator block
// and its part of the right-hand side go to rank 0 alone (the sum over the ranks is the matrix).  Phase 1 per rank, the
// separator blocks summed (the all-reduce), phase 2 and 3 per rank; x is collected from the owners.
int vo_chol_solve_split(int n, const double *A_rowmajor_lower, double *b, int c0, const int32_t *col_part, int n_ranks) {
  if (n <= 0 || !A_rowmajor_lower || !b || !col_part || n_ranks < 1) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  const int ld = (n + NB - 1) / NB * NB, m = ld / NB;
  if (ld > 4096 || c0 < 1 || c0 >= m) return VO_ERR_INVALID;
  std::vector<double> Ap((size_t)(ld + NB) * ld, 0.0);
  for (int i = 0; i < ld; i++) {
    if (i < n) {
      for (int j = 0; j <= i; j++) Ap[(size_t)i * ld + j] = A_rowmajor_lower[(size_t)i * n + j];
      Ap[(size_t)ld * ld + i] = b[i];
    } else {
      Ap[(size_t)i * ld + i] = 1.0;
    }
  }
  std::vector<unsigned long long> pattern;
  pattern_of_matrix(Ap, ld, pattern);
  hipStream_t st = vo::thread_stream();
  const int s0 = NB * c0;
  std::vector<vo::DevBuf> dA((size_t)n_ranks), dws((size_t)n_ranks);
  std::vector<unsigned long long> own((size_t)n_ranks, 0ull);
  for (int j = 0; j < c0; j++) {
    if (col_part[j] < 0) {  // (a negative segment number would index in front of `own`: ADVICE r4)
      vo::set_error("vo_chol_solve_split: col_part[%d] = %d", j, col_part[j]);
      return VO_ERR_INVALID;
    }
    own[(size_t)(col_part[j] % n_ranks)] |= 1ull << j;
  }
  std::vector<vo::CholPlan *> plans;
  auto cleanup = [&](int rc) {
    for (auto *p : plans) vo::chol_plan_destroy(p);
    for (auto &d : dA) d.release();
    for (auto &d : dws) d.release();
    return rc;
  };
  std::vector<double> sum((size_t)(ld + NB) * ld, 0.0), tmp(sum.size());
  int failed = 0;
  for (int r = 0; r < n_ranks; r++) {
    std::vector<double> Ar(Ap);
    if (r != 0)  // the separator block and its right-hand side: rank 0's contribution only
      for (int i = s0; i <= ld; i++)
        for (int j = s0; j < ld; j++) Ar[(size_t)i * ld + j] = 0.0;
    if (upload(dA[r], Ar.data(), Ar.size() * 8, st) != VO_OK || dws[r].reserve(vo::chol_workspace_bytes(ld)) != VO_OK) return cleanup(VO_ERR_HIP);
    (void)hipMemsetAsync(dws[r].p, 0, 4, st);
    vo::CholPlan *p1 = vo::chol_plan_create_split(m, pattern.data(), c0, own[r], 1);
    plans.push_back(p1);
    vo::chol_split_phase(dA[r].as<double>(), ld, dws[r].p, st, p1, 1, c0);
    if (vo::copy_d2h(tmp.data(), dA[r].p, tmp.size() * 8, st, "vo_chol_solve_split") != VO_OK || vo::stream_sync(st, "vo_chol_solve_split") != VO_OK)
      return cleanup(VO_ERR_HIP);
    for (int i = s0; i <= ld; i++)
      for (int j = s0; j < ld; j++) sum[(size_t)i * ld + j] += tmp[(size_t)i * ld + j];
  }
  std::vector<double> x((size_t)ld, 0.0);
  for (int r = 0; r < n_ranks; r++) {
    // the "all-reduced" separator block into this rank's storage (row by row: the block is not contiguous)
    for (int i = s0; i <= ld; i++)
      (void)hipMemcpyAsync(dA[r].as<double>() + (size_t)i * ld + s0, sum.data() + (size_t)i * ld + s0, (size_t)(ld - s0) * 8, hipMemcpyHostToDevice, st);
    vo::CholPlan *p2 = vo::chol_plan_create_split(m, pattern.data(), c0, own[r], 2);
    vo::CholPlan *p3 = vo::chol_plan_create_split(m, pattern.data(), c0, own[r], 3);
    plans.push_back(p2), plans.push_back(p3);
    vo::chol_split_phase(dA[r].as<double>(), ld, dws[r].p, st, p2, 2, c0);
    vo::chol_split_phase(dA[r].as<double>(), ld, dws[r].p, st, p3, 3, c0);
    int f = 0;
    if (vo::copy_d2h(&f, dws[r].p, 4, st, "vo_chol_solve_split") != VO_OK ||
        vo::copy_d2h(tmp.data(), dA[r].as<double>() + (size_t)(ld + 1) * ld, (size_t)ld * 8, st, "vo_chol_solve_split") != VO_OK ||
        vo::stream_sync(st, "vo_chol_solve_split") != VO_OK)
      return cleanup(VO_ERR_HIP);
    failed = std::max(failed, f);
    for (int i = 0; i < ld; i++) {
      const int tj = i / NB;
      const bool mine = tj < c0 ? ((own[r] >> tj) & 1ull) != 0 : r == 0;
      if (mine) x[i] = tmp[i];
    }
  }
  if (failed) {
    vo::set_error(failed == 1 ? "vo_chol_solve_split: matrix is not positive definite" : "vo_chol_solve_split: the factorisation kernel abandoned a wait");
    return cleanup(failed == 1 ? VO_ERR_INVALID : VO_ERR_HIP);
  }
  for (int i = 0; i < n; i++) b[i] = x[i];
  return cleanup(VO_OK);
}

int vo_chol_solve(int n, double *A_rowmajor_lower, double *b) {
  // dense SPD solve on the device through the same blocked kernels (test / utility entry point)
  if (n <= 0 || !A_rowmajor_lower || !b) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  const int ld = (n + NB - 1) / NB * NB;
  if (ld > 4096) {
    vo::set_error("vo_chol_solve: n = %d exceeds 4096", n);
    return VO_ERR_CAPACITY;
  }
  std::vector<double> Ap((size_t)(ld + NB) * ld, 0.0), bp(ld, 0.0);
  for (int i = 0; i < ld; i++) {
    if (i < n) {
      for (int j = 0; j <= i; j++) Ap[(size_t)i * ld + j] = A_rowmajor_lower[(size_t)i * n + j];
      Ap[(size_t)ld * ld + i] = b[i];  // right-hand side: row ld
    } else {
      Ap[(size_t)i * ld + i] = 1.0;
    }
  }
  thread_local vo::ScratchBuf dA, dfail;
  hipStream_t st = vo::thread_stream();
  auto done = [&](int r) { return r; };
  VO_CHECK(upload(dA, Ap.data(), Ap.size() * 8, st));
  VO_CHECK(dfail.reserve(vo::chol_workspace_bytes(ld)));
  VO_HIP_CHECK(hipMemsetAsync(dfail.p, 0, 4, st));
  vo::CholPlan *plan = plan_of_matrix(Ap, ld);
  vo::chol_factor_solve(dA.as<double>(), ld, dfail.p, st, plan);
  int failed = 0;
  const int crc = vo::copy_d2h(&failed, dfail.p, 4, st, "vo_chol_solve");
  const int src = vo::stream_sync(st, "vo_chol_solve");
  vo::chol_plan_destroy(plan);
  VO_CHECK(crc);
  VO_CHECK(src);
  if (failed) {
    vo::set_error(failed == 1 ? "vo_chol_solve: matrix is not positive definite" : "vo_chol_solve: the factorisation kernel abandoned a wait");
    return done(failed == 1 ? VO_ERR_INVALID : VO_ERR_HIP);
  }
  VO_CHECK(vo::copy_d2h(Ap.data(), dA.p, Ap.size() * 8, st, "vo_chol_solve"));
  VO_CHECK(vo::stream_sync(st, "vo_chol_solve"));
  for (int i = 0; i < n; i++) b[i] = Ap[(size_t)(ld + 1) * ld + i];  // solution: row ld + 1
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) A_rowmajor_lower[(size_t)i * n + j] = Ap[(size_t)i * ld + j];
  return done(VO_OK);
}

#ifdef VO_CHOL_STAMPS
// developer entry (tools/chol_stamps.py): vo_chol_solve + the stamp block of the workspace
int vo_chol_debug_solve(int n, double *A_rowmajor_lower, double *b, unsigned long long *stamps, int n_stamps) {
  const int ld = (n + NB - 1) / NB * NB;
  std::vector<double> Ap((size_t)(ld + NB) * ld, 0.0);
  for (int i = 0; i < ld; i++) {
    if (i < n) {
      for (int j = 0; j <= i; j++) Ap[(size_t)i * ld + j] = A_rowmajor_lower[(size_t)i * n + j];
      Ap[(size_t)ld * ld + i] = b[i];
    } else {
      Ap[(size_t)i * ld + i] = 1.0;
    }
  }
  vo::DevBuf dA, dws;
  VO_CHECK(upload(dA, Ap.data(), Ap.size() * 8, nullptr));
  VO_HIP_CHECK(hipDeviceSynchronize());
  const size_t wsb = vo::chol_workspace_bytes(ld);
  VO_CHECK(dws.reserve(wsb));
  VO_HIP_CHECK(hipMemset(dws.p, 0, wsb));
  vo::CholPlan *plan = plan_of_matrix(Ap, ld);
  vo::chol_factor_solve(dA.as<double>(), ld, dws.p, nullptr, plan);
  VO_HIP_CHECK(hipDeviceSynchronize());
  vo::chol_plan_destroy(plan);
  VO_HIP_CHECK(hipMemcpy(Ap.data(), dA.p, Ap.size() * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) b[i] = Ap[(size_t)(ld + 1) * ld + i];
  const int m = ld / NB;
  const size_t soff = wsb - (size_t)2 * m * 16 * 8;
  VO_HIP_CHECK(hipMemcpy(stamps, reinterpret_cast<uint8_t *>(dws.p) + soff, std::min<size_t>((size_t)n_stamps, (size_t)2 * m * 16) * 8, hipMemcpyDeviceToHost));
  int failed = 0;
  VO_HIP_CHECK(hipMemcpy(&failed, dws.p, 4, hipMemcpyDeviceToHost));
  dA.release(), dws.release();
  return failed;
}
#endif

int vo_pose_graph_solve(int n_nodes, double *quats, double *trans, const double *scales, int fixed_node, int n_edges,
                        const int32_t *edge_i, const int32_t *edge_j, const double *q_meas, const double *t_meas,
                        const double *s_meas, int fix_scale, int max_iterations, vo_lm_summary *summary) {
  if (n_nodes < 1 || n_edges < 0 || !quats || !trans || !scales || fixed_node < 0 || fixed_node >= n_nodes ||
      (n_edges > 0 && (!edge_i || !edge_j || !q_meas || !t_meas || !s_meas)))
    return VO_ERR_INVALID;
  if (!fix_scale) {
    vo::set_error("vo_pose_graph_solve: free scales are not supported (the reference always fixes them, loopClosing.cpp:15)");
    return VO_ERR_INVALID;
  }
  for (int e = 0; e < n_edges; e++)
    if (edge_i[e] < 0 || edge_i[e] >= n_nodes || edge_j[e] < 0 || edge_j[e] >= n_nodes || edge_i[e] == edge_j[e]) {
      vo::set_error("vo_pose_graph_solve: edge %d connects %d -> %d", e, edge_i[e], edge_j[e]);
      return VO_ERR_INVALID;
    }
  VO_CHECK(vo::ensure_device());
  vo_lm_summary local = {};
  if (!summary) summary = &local;
  memset(summary, 0, sizeof(*summary));
  // structure: free slots (a node enters the problem when an edge touches it), adjacency in edge order
  std::vector<int> slot(n_nodes, -1), used(n_nodes, 0), adj_start(n_nodes + 1, 0), adj_edge(2 * (size_t)n_edges);
  for (int e = 0; e < n_edges; e++) used[edge_i[e]] = used[edge_j[e]] = 1;
  int nfree = 0;
  for (int a = 0; a < n_nodes; a++)
    if (used[a] && a != fixed_node) slot[a] = nfree++;
  if (nfree == 0 || n_edges == 0) return VO_OK;
  for (int e = 0; e < n_edges; e++) adj_start[edge_i[e] + 1]++, adj_start[edge_j[e] + 1]++;
  for (int a = 0; a < n_nodes; a++) adj_start[a + 1] += adj_start[a];
  {
    std::vector<int> fill(adj_start.begin(), adj_start.end() - 1);
    for (int e = 0; e < n_edges; e++) adj_edge[fill[edge_i[e]]++] = e, adj_edge[fill[edge_j[e]]++] = e;
  }
  const int n = 6 * nfree, ld = (n + NB - 1) / NB * NB;
  if (ld > 4096) {
    vo::set_error("pose graph with %d free key-frames exceeds the dense solver (6N <= 4096)", nfree);
    return VO_ERR_CAPACITY;
  }
  // The normal matrix has a block per edge: along a trajectory a narrow band (spanning tree + covisibility neighbours)
  // plus the loop edges.  Systems of more than a few tiles are factored on the sparse plan of their structure, in the
  // block order vo::chol_choose_order picks (segments of the band factored concurrently).
  vo::CholPlan *plan = nullptr;
  struct PlanGuard {
    vo::CholPlan *&p;
    ~PlanGuard() { vo::chol_plan_destroy(p); }
  } plan_guard{plan};
  if (ld / NB >= 4) {
    std::vector<std::pair<int, int>> pairs;
    for (int e = 0; e < n_edges; e++) {
      const int sa = slot[edge_i[e]], sb = slot[edge_j[e]];
      if (sa >= 0 && sb >= 0 && sa != sb) pairs.push_back({std::max(sa, sb), std::min(sa, sb)});
    }
    const vo::CholOrder o = vo::chol_choose_order(nfree, 6, pairs, ld / NB);
    for (int a = 0; a < n_nodes; a++)
      if (slot[a] >= 0) slot[a] = o.slot_of[slot[a]];
    plan = vo::chol_plan_create(ld / NB, o.pattern.data());
  }
  std::vector<double> x(7 * (size_t)n_nodes);
  for (int a = 0; a < n_nodes; a++) {
    memcpy(&x[7 * a], quats + 4 * a, 32);
    memcpy(&x[7 * a + 4], trans + 3 * a, 24);
  }
  // grow-only, per host thread (150 MB at 500 key-frames; vo_release_thread_scratch() gives it back)
  thread_local vo::ScratchBuf d_ei, d_ej, d_qm, d_tm, d_sm, d_sc, d_slot, d_as, d_ae, d_H, d_g, d_cs, d_A, d_rhs, d_x,
      d_xc, d_part, d_norm, d_fail, d_cost;
  hipStream_t st = vo::thread_stream();
  auto done = [&](int r) { return r; };
#define PG_TRY(expr) VO_CHECK(expr)
  PG_TRY(upload(d_ei, edge_i, (size_t)n_edges * 4, st));
  PG_TRY(upload(d_ej, edge_j, (size_t)n_edges * 4, st));
  PG_TRY(upload(d_qm, q_meas, (size_t)n_edges * 32, st));
  PG_TRY(upload(d_tm, t_meas, (size_t)n_edges * 24, st));
  PG_TRY(upload(d_sm, s_meas, (size_t)n_edges * 8, st));
  PG_TRY(upload(d_sc, scales, (size_t)n_nodes * 8, st));
  PG_TRY(upload(d_slot, slot.data(), (size_t)n_nodes * 4, st));
  PG_TRY(upload(d_as, adj_start.data(), adj_start.size() * 4, st));
  PG_TRY(upload(d_ae, adj_edge.data(), adj_edge.size() * 4, st));
  PG_TRY(upload(d_x, x.data(), x.size() * 8, st));
  PG_TRY(d_xc.reserve(x.size() * 8));
  PG_TRY(d_H.reserve((size_t)ld * ld * 8));
  PG_TRY(d_A.reserve((size_t)(ld + NB) * ld * 8));
  PG_TRY(d_g.reserve((size_t)ld * 8));
  PG_TRY(d_cs.reserve((size_t)ld * 8));
  PG_TRY(d_rhs.reserve((size_t)ld * 8));
  PG_TRY(d_part.reserve((size_t)std::max(ld, 64) * 8));
  PG_TRY(d_norm.reserve((size_t)n_nodes * 16));
  const int cost_blocks = (n_edges + 255) / 256;
  PG_TRY(d_cost.reserve((size_t)cost_blocks * 8));
  PG_TRY(d_fail.reserve(vo::chol_workspace_bytes(ld)));
  PgDev P;
  P.n_nodes = n_nodes, P.n_edges = n_edges, P.n = n, P.ld = ld;
  P.e_i = d_ei.as<int>(), P.e_j = d_ej.as<int>(), P.qm = d_qm.as<double>(), P.tm = d_tm.as<double>();
  P.sm = d_sm.as<double>(), P.scales = d_sc.as<double>(), P.slot = d_slot.as<int>();
  P.adj_start = d_as.as<int>(), P.adj_edge = d_ae.as<int>(), P.H = d_H.as<double>(), P.g = d_g.as<double>();
  P.colscale = d_cs.as<double>();
  auto cost_of = [&](const double *dx, double &c) -> int {
    hipLaunchKernelGGL(k_pg_cost, dim3(cost_blocks), dim3(256), 0, st, P, dx, d_cost.as<double>());
    return sum_device(d_cost, cost_blocks, st, c);
  };
  auto linearize = [&](const double *dx, int first, double &gmax) -> int {
    hipLaunchKernelGGL(k_pg_linearize, dim3(n_nodes), dim3(64), 0, st, P, dx, first);
    std::vector<double> g(n);
    VO_CHECK(vo::copy_d2h(g.data(), d_g.p, (size_t)n * 8, st, "pose graph"));
    VO_CHECK(vo::stream_sync(st, "pose graph"));
    gmax = 0;
    for (double v : g) gmax = std::max(gmax, std::fabs(v));
    return VO_OK;
  };
  // ---- Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy (contract: DESIGN.md section 3)
  double radius = 1e4, decrease = 2.0, x_cost = 0, gmax = 0;
  double *dx = d_x.as<double>(), *dxc = d_xc.as<double>();
  PG_TRY(linearize(dx, 1, gmax));
  PG_TRY(cost_of(dx, x_cost));
  summary->initial_cost = x_cost;
  int iterations = 0, accepted = 0, termination = 0, invalid = 0, chol_retries = 0;
  bool last_ok = false;
  for (int it = 1;; it++) {
    if (it - 1 >= max_iterations) {
      termination = 0;
      break;
    }
    if (last_ok && gmax <= 1e-10) {
      termination = 3;
      break;
    }
    if (radius < 1e-32) {
      termination = 4;
      break;
    }
    iterations = it;
    last_ok = false;
    // damped system, factorisation, solve
    VO_HIP_CHECK(hipMemsetAsync(d_fail.p, 0, 4, st));
    hipLaunchKernelGGL(k_pg_damp, dim3((unsigned)(((long long)ld * ld + 255) / 256)), dim3(256), 0, st, P,
                       d_A.as<double>(), d_rhs.as<double>(), radius);
    VO_HIP_CHECK(hipMemsetAsync(d_A.as<double>() + (size_t)(ld + 1) * ld, 0, (size_t)(NB - 1) * ld * 8, st));  // rows below the rhs
    vo::chol_factor_solve(d_A.as<double>(), ld, d_fail.p, st, plan);
    const double *ysol = d_A.as<double>() + (size_t)(ld + 1) * ld;
    hipLaunchKernelGGL(k_pg_model, dim3(n), dim3(256), 0, st, P, ysol, d_part.as<double>());
    hipLaunchKernelGGL(k_pg_candidate, dim3((n_nodes + 127) / 128), dim3(128), 0, st, P, dx, ysol, dxc,
                       d_norm.as<double>());
    hipLaunchKernelGGL(k_pg_cost, dim3(cost_blocks), dim3(256), 0, st, P, dxc, d_cost.as<double>());  // the candidate's cost
    // one synchronisation per LM iteration: fail flag, model partials, norms and candidate cost come back together
    int failed = 0;
    std::vector<double> h_part(n), h_norm(2 * (size_t)n_nodes), h_cost(cost_blocks);
    PG_TRY(vo::copy_d2h(&failed, d_fail.p, 4, st, "pose graph"));
    PG_TRY(vo::copy_d2h(h_part.data(), d_part.p, h_part.size() * 8, st, "pose graph"));
    PG_TRY(vo::copy_d2h(h_norm.data(), d_norm.p, h_norm.size() * 8, st, "pose graph"));
    PG_TRY(vo::copy_d2h(h_cost.data(), d_cost.p, h_cost.size() * 8, st, "pose graph"));
    PG_TRY(vo::stream_sync(st, "pose graph"));
    if (failed == 2) {
      // the factorisation kernel gave up waiting for a tile (a GPU shared with other work can starve it): that is not
      // "the matrix is not positive definite" -- treating it as an invalid step would shrink the trust region and change
      // the LM path from run to run.  Same radius, same system, once more; a second abandonment is an error.
      if (++chol_retries > 2) {
        vo::set_error("pose graph: the factorisation kernel abandoned a wait three times (GPU oversubscribed?)");
        return VO_ERR_HIP;
      }
      it--;
      continue;
    }
    double model = 0;
    if (!failed)
      for (double v : h_part) model += v;
    if (failed || !(model > 0.0) || !std::isfinite(model)) {
      if (++invalid >= 5) {
        termination = 4;
        break;
      }
      radius /= decrease;
      decrease *= 2.0;
      continue;
    }
    invalid = 0;
    double cand = 0, xn2 = 0, sn2 = 0;
    for (double v : h_cost) cand += v;
    if (!std::isfinite(cand)) cand = 1.7976931348623157e308;
    for (int a = 0; a < n_nodes; a++) xn2 += h_norm[2 * (size_t)a], sn2 += h_norm[2 * (size_t)a + 1];
    const double x_norm = std::sqrt(xn2), step_norm = std::sqrt(sn2);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) {
      termination = 2;
      break;
    }
    const double change = x_cost - cand;
    if (std::fabs(change) <= 1e-6 * x_cost) {
      termination = 1;
      break;
    }
    const double rel = change / model;
    if (rel > 1e-3) {
      std::swap(dx, dxc);
      x_cost = cand;
      PG_TRY(linearize(dx, 0, gmax));
      const double t2 = 2.0 * rel - 1.0;
      radius = std::min(radius / std::max(1.0 / 3.0, 1.0 - t2 * t2 * t2), 1e16);
      decrease = 2.0;
      accepted++;
      last_ok = true;
    } else {
      radius /= decrease;
      decrease *= 2.0;
    }
  }
  summary->iterations = iterations, summary->accepted = accepted, summary->termination = termination;
  summary->final_cost = x_cost, summary->final_radius = radius;
  PG_TRY(vo::copy_d2h(x.data(), dx, x.size() * 8, st, "pose graph"));
  PG_TRY(vo::stream_sync(st, "pose graph"));
  for (int a = 0; a < n_nodes; a++) {
    memcpy(quats + 4 * a, &x[7 * a], 32);
    memcpy(trans + 3 * a, &x[7 * a + 4], 24);
  }
#undef PG_TRY
  return done(VO_OK);
}

int vo_sim3_reanchor_points(int n_points, const double *points_in, const int32_t *ref_node, int n_nodes,
                            const double *S_rw, const double *S_wr, double *points_out) {
  if (n_points < 0 || n_nodes < 0 || (n_points > 0 && (!points_in || !ref_node || !S_rw || !S_wr || !points_out)))
    return VO_ERR_INVALID;
  if (n_points == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf dp, dr, d1, d2, dout;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(upload(dp, points_in, (size_t)n_points * 24, st));
  VO_CHECK(upload(dr, ref_node, (size_t)n_points * 4, st));
  VO_CHECK(upload(d1, S_rw, (size_t)n_nodes * 64, st));
  VO_CHECK(upload(d2, S_wr, (size_t)n_nodes * 64, st));
  VO_CHECK(dout.reserve((size_t)n_points * 24));
  hipLaunchKernelGGL(k_sim3_reanchor, dim3((n_points + 255) / 256), dim3(256), 0, st, n_points, dp.as<double>(),
                     dr.as<int>(), d1.as<double>(), d2.as<double>(), dout.as<double>());
  VO_CHECK(vo::copy_d2h(points_out, dout.p, (size_t)n_points * 24, st, "vo_sim3_reanchor_points"));
  return vo::stream_sync(st, "vo_sim3_reanchor_points");
}

}  // extern "C"
