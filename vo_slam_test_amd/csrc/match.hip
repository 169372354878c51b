// match.hip -- Hamming distance matrix on gfx950 and the guided greedy matchers that consume it.
// Replaces the arithmetic of myslam::Matcher (reference src/matcher.cpp): computeDistance
// (:1240-1256) becomes one all-pairs kernel; the sequential, order-dependent assignment logic of
// searchByProjection (:18-148, :274-353) is replayed on the host over the device-computed matrix
// so that match pairs stay identical to the reference's visiting order.
#include "vo_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace vo;

// K6: D[i][j] = popcount(A_i xor B_j) over 256 bits.  A workgroup owns 512 columns x 128 rows: each thread
// keeps two B descriptors in registers (fetched as four 16-byte loads: a wavefront reads 4 KB contiguous) and
// streams the 128-row A tile from LDS (all lanes read the same address: a broadcast, no bank conflict);
// results leave as one 4-byte store per lane per row, i.e. 256 contiguous bytes per wavefront -- the kernel
// is bound by the 2 bytes/pair it writes and by the 16 vector instructions a distance costs.
constexpr int kHamRows = 128;
typedef uint32_t ham_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_hamming(const uint32_t *A, int na, long long a_stride,
                                                 const uint32_t *B, int nb, long long b_stride,
                                                 uint16_t *D, long long d_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t a[kHamRows][8];
  const int tid = threadIdx.x;
  const long long p = blockIdx.z;
  A += p * a_stride * 8;
  B += p * b_stride * 8;
  D += p * d_stride;
  const int i0 = blockIdx.y * kHamRows;
  const int j0 = (blockIdx.x * 256 + tid) * 2;
  const bool wide = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;  // uniform
  for (int c = tid; c < kHamRows * 2; c += 256) {  // 16-byte chunk c of the tile: row c / 2, half c & 1
    const int r = c >> 1;
    ham_u32x4 v = {0u, 0u, 0u, 0u};
    if (i0 + r < na) {
      const uint32_t *src = A + (long long)(i0 + r) * 8 + 4 * (c & 1);
      if (wide)
        v = *reinterpret_cast<const ham_u32x4 *>(src);
      else
        v = ham_u32x4{src[0], src[1], src[2], src[3]};
    }
    *reinterpret_cast<ham_u32x4 *>(&a[r][4 * (c & 1)]) = v;
  }
  uint32_t b0[8], b1[8];
  {
    ham_u32x4 q[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    // the last column of an odd nb is clamped, not predicated (a predicated load becomes a branch)
    const uint32_t *s0 = B + (long long)min(j0, nb - 1) * 8, *s1 = B + (long long)min(j0 + 1, nb - 1) * 8;
    if (wide) {
      q[0] = reinterpret_cast<const ham_u32x4 *>(s0)[0], q[1] = reinterpret_cast<const ham_u32x4 *>(s0)[1];
      q[2] = reinterpret_cast<const ham_u32x4 *>(s1)[0], q[3] = reinterpret_cast<const ham_u32x4 *>(s1)[1];
    } else {
#pragma unroll
      for (int w = 0; w < 4; w++) q[0][w] = s0[w], q[1][w] = s0[4 + w], q[2][w] = s1[w], q[3][w] = s1[4 + w];
    }
#pragma unroll
    for (int w = 0; w < 4; w++) b0[w] = q[0][w], b0[4 + w] = q[1][w], b1[w] = q[2][w], b1[4 + w] = q[3][w];
  }
  __syncthreads();
  if (j0 >= nb) return;
  const bool pair_store = ((nb & 1) == 0);
  const int rows = min(kHamRows, na - i0);
  uint16_t *o = D + (long long)i0 * nb + j0;
  for (int r = 0; r < rows; r++, o += nb) {
    int d0 = 0, d1 = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const uint32_t av = a[r][w];
      d0 += __popc(av ^ b0[w]);
      d1 += __popc(av ^ b1[w]);
    }
    if (pair_store) {
      *reinterpret_cast<uint32_t *>(o) = (uint32_t)d0 | ((uint32_t)d1 << 16);
    } else {
      o[0] = (uint16_t)d0;
      if (j0 + 1 < nb) o[1] = (uint16_t)d1;
    }
  }
}

int launch_hamming(const uint8_t *a, int na, size_t as, const uint8_t *b, int nb, size_t bs, uint16_t *d,
                   size_t ds, int n_pairs, hipStream_t st) {
  if (na <= 0 || nb <= 0 || n_pairs <= 0) return VO_OK;
  dim3 grid((nb + 511) / 512, (na + kHamRows - 1) / kHamRows, n_pairs);
  hipLaunchKernelGGL(k_hamming, grid, dim3(256), 0, st, reinterpret_cast<const uint32_t *>(a), na, (long long)as,
                     reinterpret_cast<const uint32_t *>(b), nb, (long long)bs, d, (long long)ds);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

// MapPoint::computeDescriptor (mappoint.cpp:118-179): among the N descriptors observing one map point, the one
// whose MEDIAN Hamming distance to all N (itself included, sorted row entry int(0.5 * (N - 1))) is smallest;
// strict < keeps the first.  One workgroup per map point (a batch of points per launch): descriptors staged in
// LDS at a 36-byte pitch (conflict-free column reads), one wavefront per row -- lanes are columns, the row's k-th
// smallest distance by a 9-step bisection over ballot counts -- and an LDS atomicMin over (median << 16 | row).
constexpr int kMedianMax = 1024;  // observations per map point (36 KB of LDS)
__global__ __launch_bounds__(256) void k_median_desc(const uint32_t *desc, const int *offsets, int *best_idx) {
  extern __shared__ uint32_t md_lds[];  // [n][9]
  __shared__ unsigned s_best;
  const int set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = offsets[set], n = offsets[set + 1] - o;
  if (n <= 0) {  // `if (desp.empty()) return;` :136-137 -- nothing selected
    if (tid == 0) best_idx[set] = -1;
    return;
  }
  for (int i = tid; i < n * 8; i += 256) md_lds[(i >> 3) * 9 + (i & 7)] = desc[(long long)o * 8 + i];
  if (tid == 0) s_best = 256u << 16;  // bestMid = 256, bestIdx = 0  :159-160
  __syncthreads();
  const int k = (n - 1) / 2;  // int(0.5 * (N - 1))  :166
  constexpr int NC = kMedianMax / 64;
  for (int row = wave; row < n; row += 4) {
    uint32_t a[8];
#pragma unroll
    for (int w = 0; w < 8; w++) a[w] = md_lds[row * 9 + w];
    int d[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      d[c] = 0x7fff;
      const int j = lane + 64 * c;
      if (64 * c < n && j < n) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) s += __popc(a[w] ^ md_lds[j * 9 + w]);
        d[c] = s;
      }
    }
    int lo = 0, hi = 256;  // smallest v with #(d <= v) >= k + 1
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < NC; c++)
        if (64 * c < n) cnt += __popcll(__builtin_amdgcn_ballot_w64(d[c] <= mid));
      if (cnt >= k + 1) hi = mid; else lo = mid + 1;
    }
    if (lane == 0) atomicMin(&s_best, ((unsigned)lo << 16) | (unsigned)row);
  }
  __syncthreads();
  if (tid == 0) best_idx[set] = (int)(s_best & 0xffffu);
}

// ---- host replay helpers (BoW-node searches: the candidate sets are vocabulary nodes, not grid windows) ----
constexpr int HISTO_LENGTH = 30;               // matcher.cpp:13

// device all-pairs distances for (queries x features), back on the host
int distance_matrix(const uint8_t *q_desc, int nq, const uint8_t *f_desc, int nf, std::vector<uint16_t> &D) {
  D.assign((size_t)nq * nf, 0);
  if (nq <= 0 || nf <= 0) return VO_OK;
  return vo_hamming_matrix(q_desc, nq, f_desc, nf, D.data());
}

void three_max(const std::vector<std::vector<int>> &h, int &i1, int &i2, int &i3) {  // matcher.cpp:1258-1304
  int m1 = 0, m2 = 0, m3 = 0;
  i1 = i2 = i3 = -1;
  for (int i = 0; i < (int)h.size(); i++) {
    const int s = (int)h[i].size();
    if (s > m1) {
      m3 = m2, i3 = i2, m2 = m1, i2 = i1, m1 = s, i1 = i;
    } else if (s > m2) {
      m3 = m2, i3 = i2, m2 = s, i2 = i;
    } else if (s > m3) {
      m3 = s, i3 = i;
    }
  }
  if (m2 < 0.1f * (float)m1)
    i2 = i3 = -1;
  else if (m3 < 0.1f * (float)m1)
    i3 = -1;
}

}  // namespace

extern "C" {

int vo_hamming_matrix_dev(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *d, void *stream) {
  if ((na > 0 && !a) || (nb > 0 && !b) || (na > 0 && nb > 0 && !d) || na < 0 || nb < 0) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  return launch_hamming(a, na, 0, b, nb, 0, d, 0, 1, (hipStream_t)stream);
}

int vo_hamming_matrix_batch_dev(const uint8_t *a, int na, size_t as, const uint8_t *b, int nb, size_t bs,
                                uint16_t *d, size_t ds, int n_pairs, void *stream) {
  if (!a || !b || !d || na < 0 || nb < 0 || n_pairs < 0) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  return launch_hamming(a, na, as, b, nb, bs, d, ds, n_pairs, (hipStream_t)stream);
}

int vo_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *d) {
  if (na < 0 || nb < 0) return VO_ERR_INVALID;
  if (na == 0 || nb == 0) return VO_OK;
  if (!a || !b || !d) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  // per host thread, grow-only: the stateless entry points do not allocate after the first call at a size
  // (never freed: a few MB per calling thread for the life of the process); the calling thread's own stream
  thread_local vo::ScratchBuf da, db, dd;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::upload(da, a, (size_t)na * 32, st, "vo_hamming_matrix"));
  VO_CHECK(vo::upload(db, b, (size_t)nb * 32, st, "vo_hamming_matrix"));
  VO_CHECK(dd.reserve((size_t)na * nb * 2));
  VO_CHECK(launch_hamming(da.as<uint8_t>(), na, 0, db.as<uint8_t>(), nb, 0, dd.as<uint16_t>(), 0, 1, st));
  VO_CHECK(vo::copy_d2h(d, dd.p, (size_t)na * nb * 2, st, "vo_hamming_matrix"));
  return vo::stream_sync(st, "vo_hamming_matrix");
}

int vo_median_descriptor(const uint8_t *desc, int n_sets, const int32_t *offsets, int32_t *best_idx) {
  if (n_sets < 0 || (n_sets > 0 && (!offsets || !best_idx))) return VO_ERR_INVALID;
  if (n_sets == 0) return VO_OK;
  const int total = offsets[n_sets];
  if (total < 0 || (total > 0 && !desc)) return VO_ERR_INVALID;
  int nmax = 0;
  for (int s = 0; s < n_sets; s++) {
    const int n = offsets[s + 1] - offsets[s];
    if (n < 0) return VO_ERR_INVALID;
    nmax = std::max(nmax, n);
  }
  if (nmax > kMedianMax) {
    vo::set_error("vo_median_descriptor: %d observations of one map point exceed %d", nmax, kMedianMax);
    return VO_ERR_CAPACITY;
  }
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf dd, doff, dbest;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::upload(dd, desc, (size_t)total * 32, st, "vo_median_descriptor"));
  VO_CHECK(vo::upload(doff, offsets, (size_t)(n_sets + 1) * 4, st, "vo_median_descriptor"));
  VO_CHECK(dbest.reserve((size_t)n_sets * 4));
  const size_t lds = (size_t)std::max(nmax, 1) * 36;
  hipLaunchKernelGGL(k_median_desc, dim3(n_sets), dim3(256), lds, st, dd.as<uint32_t>(), doff.as<int>(), dbest.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(best_idx, dbest.p, (size_t)n_sets * 4, st, "vo_median_descriptor"));
  return vo::stream_sync(st, "vo_median_descriptor");
}

}  // extern "C"

namespace {

constexpr int TH_LOW = 50;  // matcher.cpp:12

struct RotHist {  // rotation-consistency filter shared by the search routines (:128-145 and friends)
  std::vector<std::vector<int>> bins = std::vector<std::vector<int>>(HISTO_LENGTH);
  void add(float angle_a, float angle_b, int idx, bool cv_round) {
    const float pdf = HISTO_LENGTH / 360.0f;
    float r = angle_a - angle_b;
    if (r < 0) r += 360.0f;
    int bin = cv_round ? (int)lrintf(r * pdf) : (int)roundf(r * pdf);
    if (bin == HISTO_LENGTH) bin = 0;
    bins[bin].push_back(idx);
  }
  template <class F>
  int prune(F &&drop) {  // calls drop(idx) for every entry outside the three dominant bins
    int i1, i2, i3, removed = 0;
    three_max(bins, i1, i2, i3);
    for (int b = 0; b < HISTO_LENGTH; b++)
      if (b != i1 && b != i2 && b != i3)
        for (int idx : bins[b]) {
          drop(idx);
          removed++;
        }
    return removed;
  }
};

// walk two ascending node lists, calling f(ia, ib) for every common node id (:541-544 lower_bound walk)
template <class F>
void for_common_nodes(const vo_bow_view &a, const vo_bow_view &b, F &&f) {
  int ia = 0, ib = 0;
  while (ia < a.n_nodes && ib < b.n_nodes) {
    if (a.node_id[ia] == b.node_id[ib]) {
      f(ia, ib);
      ia++, ib++;
    } else if (a.node_id[ia] < b.node_id[ib]) {
      ia++;
    } else {
      ib++;
    }
  }
}

}  // namespace

extern "C" {

int vo_match_bow(const vo_frame_view *a, const uint8_t *a_valid, const vo_bow_view *an, const vo_frame_view *b,
                 const uint8_t *b_valid, const vo_bow_view *bn, int mode, float ratio, int check_rot, int32_t *match,
                 int *n_matches) {
  if (!a || !b || !an || !bn || !a_valid || !match || !n_matches || (mode != 0 && mode != 1) || (mode == 1 && !b_valid))
    return VO_ERR_INVALID;
  const int nout = mode == 0 ? b->n : a->n;
  for (int i = 0; i < nout; i++) match[i] = -1;
  *n_matches = 0;
  if (a->n == 0 || b->n == 0) return VO_OK;
  std::vector<uint16_t> D;
  VO_CHECK(distance_matrix(a->desc, a->n, b->desc, b->n, D));
  std::vector<uint8_t> taken(b->n, 0);
  RotHist rot;
  int cnt = 0;
  for_common_nodes(*an, *bn, [&](int ia, int ib) {
    for (int s = an->start[ia]; s < an->start[ia + 1]; s++) {
      const int i1 = (int)an->feat[s];
      if (!a_valid[i1]) continue;
      int best1 = 256, best2 = 256, bidx = -1;
      for (int t = bn->start[ib]; t < bn->start[ib + 1]; t++) {
        const int i2 = (int)bn->feat[t];
        if (mode == 0 ? match[i2] >= 0 : (taken[i2] || !b_valid[i2])) continue;
        const int d = D[(size_t)i1 * b->n + i2];
        if (d < best1)
          best2 = best1, best1 = d, bidx = i2;
        else if (d < best2)
          best2 = d;
      }
      if (best1 <= TH_LOW && (float)best1 < ratio * (float)best2) {
        if (mode == 0) {
          match[bidx] = i1;
          if (check_rot) rot.add(a->angle[i1], b->angle[bidx], bidx, true);
        } else {
          match[i1] = bidx;
          taken[bidx] = 1;
          if (check_rot) rot.add(a->angle[i1], b->angle[bidx], i1, false);  // :637 uses round()
        }
        cnt++;
      }
    }
  });
  if (check_rot) cnt -= rot.prune([&](int idx) { match[idx] = -1; });
  *n_matches = cnt;
  return VO_OK;
}

int vo_match_triangulation(const vo_frame_view *a, const uint8_t *a_has, const vo_bow_view *an, const vo_frame_view *b,
                           const uint8_t *b_has, const vo_bow_view *bn, const double F[9], float ex, float ey,
                           const float *scale_factors, int check_rot, int32_t *match12, int *n_matches) {
  if (!a || !b || !an || !bn || !a_has || !b_has || !F || !scale_factors || !match12 || !n_matches) return VO_ERR_INVALID;
  for (int i = 0; i < a->n; i++) match12[i] = -1;
  *n_matches = 0;
  if (a->n == 0 || b->n == 0) return VO_OK;
  std::vector<uint16_t> D;
  VO_CHECK(distance_matrix(a->desc, a->n, b->desc, b->n, D));
  std::vector<uint8_t> taken(b->n, 0);
  RotHist rot;
  int cnt = 0;
  for_common_nodes(*an, *bn, [&](int ia, int ib) {
    for (int s = an->start[ia]; s < an->start[ia + 1]; s++) {
      const int i1 = (int)an->feat[s];
      if (a_has[i1]) continue;
      const bool stereo1 = a->uright[i1] >= 0;
      // epipolar line of feature 1 in image 2: l = F12^T p1 (checkEpipolarConstrain, :1306-1324)
      const double l0 = a->x[i1] * F[0] + a->y[i1] * F[3] + F[6], l1 = a->x[i1] * F[1] + a->y[i1] * F[4] + F[7],
                   l2 = a->x[i1] * F[2] + a->y[i1] * F[5] + F[8];
      const float den = (float)(l0 * l0 + l1 * l1);
      int best = TH_LOW, bidx = -1;
      for (int t = bn->start[ib]; t < bn->start[ib + 1]; t++) {
        const int i2 = (int)bn->feat[t];
        if (taken[i2] || b_has[i2]) continue;
        const int d = D[(size_t)i1 * b->n + i2];
        if (d > TH_LOW || d > best) continue;  // an equal later distance replaces the earlier one (:928)
        const float sigma = scale_factors[b->octave[i2]];
        if (!stereo1 && !(b->uright[i2] >= 0)) {
          const float dx = ex - b->x[i2], dy = ey - b->y[i2];
          if (dx * dx + dy * dy < 100 * sigma) continue;  // too close to the epipole (:932-940)
        }
        if (den == 0) continue;
        const float num = (float)(l0 * b->x[i2] + l1 * b->y[i2] + l2);
        if (num * num / den < 3.84f * sigma * sigma) best = d, bidx = i2;
      }
      if (bidx >= 0) {
        match12[i1] = bidx;
        taken[bidx] = 1;
        cnt++;
        if (check_rot) rot.add(a->angle[i1], b->angle[bidx], i1, false);
      }
    }
  });
  if (check_rot) cnt -= rot.prune([&](int idx) { match12[idx] = -1; });
  *n_matches = cnt;
  return VO_OK;
}


// ---- BoW transform (DBoW3::Vocabulary::transform as called by computeBow, frame.cpp:248-253,
// keyframe.cpp:394-398).  The vocabulary tree lives in HBM as flat arrays; one lane per feature
// walks it: at every level the child with the smallest Hamming distance (first wins ties).
}  // extern "C" (kernels below)

namespace {

struct VocabDev {
  int n_nodes, depth;
  const int *child_start, *children, *word_id;
  const uint32_t *desc;  // 8 dwords per node
  const double *weight;
};

__global__ __launch_bounds__(256) void k_bow_transform(VocabDev V, int n, const uint32_t *feat, int levelsup, int *out_word,
                                                       double *out_weight, int *out_node) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t f[8];
#pragma unroll
  for (int k = 0; k < 8; k++) f[k] = feat[8LL * i + k];
  const int nid_level = V.depth - levelsup;
  int id = 0, level = 0, nid = 0;
  while (V.word_id[id] < 0 && V.child_start[id + 1] > V.child_start[id]) {
    level++;
    const int c0 = V.child_start[id], c1 = V.child_start[id + 1];
    int best = -1, bestd = 1 << 30;
    for (int c = c0; c < c1; c++) {
      const int ch = V.children[c];
      const uint32_t *d = V.desc + 8LL * ch;
      int dist = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) dist += __popc(f[k] ^ d[k]);
      if (dist < bestd) bestd = dist, best = ch;  // strict: the first minimum wins
    }
    id = best;
    if (level == nid_level) nid = id;
  }
  out_word[i] = V.word_id[id];
  out_weight[i] = V.weight[id];
  out_node[i] = nid;
}

}  // namespace

struct vo_vocab {
  VocabDev V{};
  vo::DevBuf b_cs, b_ch, b_wid, b_desc, b_w;
};

extern "C" {

int vo_vocab_create(vo_vocab **out, int n_nodes, int depth_L, const int32_t *child_start, const int32_t *children,
                    const uint8_t *node_desc, const double *node_weight, const int32_t *word_id) {
  if (!out || n_nodes < 1 || depth_L < 0 || !child_start || !node_desc || !node_weight || !word_id) return VO_ERR_INVALID;
  const int n_children = child_start[n_nodes];
  if (n_children < 0 || (n_children > 0 && !children)) return VO_ERR_INVALID;
  for (int i = 0; i < n_nodes; i++)
    if (child_start[i] > child_start[i + 1]) return VO_ERR_INVALID;
  for (int c = 0; c < n_children; c++)
    if (children[c] <= 0 || children[c] >= n_nodes) {  // a child is never the root: guarantees termination
      vo::set_error("vo_vocab_create: child %d out of range", children[c]);
      return VO_ERR_INVALID;
    }
  VO_CHECK(vo::ensure_device());
  vo_vocab *v = new vo_vocab();
  auto up = [](vo::DevBuf &b, const void *src, size_t bytes) -> int {
    VO_CHECK(b.reserve(std::max<size_t>(bytes, 64)));
    if (bytes) VO_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return VO_OK;
  };
  int rc;
  if ((rc = up(v->b_cs, child_start, (size_t)(n_nodes + 1) * 4)) != VO_OK || (rc = up(v->b_ch, children, (size_t)n_children * 4)) != VO_OK ||
      (rc = up(v->b_wid, word_id, (size_t)n_nodes * 4)) != VO_OK || (rc = up(v->b_desc, node_desc, (size_t)n_nodes * 32)) != VO_OK ||
      (rc = up(v->b_w, node_weight, (size_t)n_nodes * 8)) != VO_OK) {
    for (vo::DevBuf *b : {&v->b_cs, &v->b_ch, &v->b_wid, &v->b_desc, &v->b_w}) b->release();
    delete v;
    return rc;
  }
  v->V.n_nodes = n_nodes, v->V.depth = depth_L;
  v->V.child_start = v->b_cs.as<int>(), v->V.children = v->b_ch.as<int>(), v->V.word_id = v->b_wid.as<int>();
  v->V.desc = v->b_desc.as<uint32_t>(), v->V.weight = v->b_w.as<double>();
  *out = v;
  return VO_OK;
}

void vo_vocab_destroy(vo_vocab *v) {
  if (!v) return;
  for (vo::DevBuf *b : {&v->b_cs, &v->b_ch, &v->b_wid, &v->b_desc, &v->b_w}) b->release();
  delete v;
}

int vo_bow_transform(const vo_vocab *v, int n, const uint8_t *desc, int levelsup, int32_t *word_id, double *weight,
                     int32_t *node_id) {
  if (!v || n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) return VO_ERR_INVALID;
  if (n == 0) return VO_OK;
  thread_local vo::ScratchBuf d_f, d_w, d_wt, d_n;
  VO_CHECK(d_f.reserve((size_t)n * 32));
  VO_CHECK(d_w.reserve((size_t)n * 4));
  VO_CHECK(d_wt.reserve((size_t)n * 8));
  VO_CHECK(d_n.reserve((size_t)n * 4));
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::copy_h2d(d_f.p, desc, (size_t)n * 32, st, "vo_bow_transform"));
  hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, st, v->V, n, d_f.as<uint32_t>(), levelsup,
                     d_w.as<int>(), d_wt.as<double>(), d_n.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(word_id, d_w.p, (size_t)n * 4, st, "vo_bow_transform"));
  VO_CHECK(vo::copy_d2h(weight, d_wt.p, (size_t)n * 8, st, "vo_bow_transform"));
  VO_CHECK(vo::copy_d2h(node_id, d_n.p, (size_t)n * 4, st, "vo_bow_transform"));
  return vo::stream_sync(st, "vo_bow_transform");
}

}  // extern "C"
