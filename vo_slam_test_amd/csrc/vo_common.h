// vo_common.h -- shared host/device helpers for the gfx950 library (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <vector>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/vo_hip.h"

namespace vo {

void set_error(const char *fmt, ...);
int ensure_device();  // VO_OK or VO_ERR_NO_DEVICE

#define VO_HIP_CHECK(expr)                                                                 \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      vo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VO_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

#define VO_CHECK(expr)           \
  do {                           \
    int _s = (expr);             \
    if (_s != VO_OK) return _s;  \
  } while (0)

// growable device buffer
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  bool view = false;  // p points into another DevBuf (an arena): never freed through this object
  int reserve(size_t n) {
    if (view) p = nullptr, bytes = 0, view = false;
    if (n <= bytes) return VO_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    VO_HIP_CHECK(hipMalloc(&p, n));
    bytes = n;
    return VO_OK;
  }
  void release() {
    if (p && !view) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    view = false;
  }
  void set_view(void *ptr, size_t n) {  // (an owned allocation is kept aside by the caller or released first)
    if (p && !view) (void)hipFree(p);
    p = ptr, bytes = n, view = true;
  }
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

// Grow-only device scratch of the stateless entry points: one set per host thread (`thread_local ScratchBuf`), kept
// between calls so that the hot path neither allocates nor frees (hipFree synchronises the whole device).  Every
// ScratchBuf registers itself with its thread; vo_release_thread_scratch() frees what the calling thread holds (the
// buffers grow again on the next call), e.g. before a worker thread exits or after a one-off 150 MB pose graph.
struct ScratchBuf : DevBuf {
  ScratchBuf();
};
size_t release_thread_scratch();  // bytes freed

// Grow-only page-locked host staging (per host thread where used): copies to and from it run at full PCIe
// rate and without the runtime's own bounce buffer.  Never freed (thread-exit order vs. runtime teardown).
struct PinnedBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(size_t n) {
    if (n <= bytes) return VO_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    const size_t want = n + n / 2 + 4096;
    VO_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    bytes = want;
    return VO_OK;
  }
  uint8_t *data() const { return reinterpret_cast<uint8_t *>(p); }
};

constexpr int kWave = 64;

// Per host thread: a non-blocking stream for the stateless entry points (vo_hamming_matrix, vo_pose_only_solve,
// vo_sim3_solve, vo_pose_graph_solve, ...).  The reference calls them concurrently from the tracking, local-
// mapping and loop-closing threads: on the legacy NULL stream a 0.4 ms pose-only solve would queue behind a
// 500-key-frame pose graph, and every hipDeviceSynchronize would stall the other threads' streams.  Created on
// first use, never destroyed (thread-exit order vs. runtime teardown).
hipStream_t thread_stream();
// checked copies on a stream (error text names `what`)
int copy_h2d(void *dst, const void *src, size_t bytes, hipStream_t st, const char *what);
int copy_d2h(void *dst, const void *src, size_t bytes, hipStream_t st, const char *what);
int stream_sync(hipStream_t st, const char *what);
// reserve + copy
int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st, const char *what);

// Dense SPD solve on the device (csrc/chol.hip).  A: (ld + 64) rows x ld columns, row-major, ld a multiple of 64.
// Rows 0..ld-1: the matrix, lower triangle used (identity on the padding diagonal); row ld: the right-hand side;
// rows ld+1.. are spare.  The lower triangle is factored in place (L L^T, 64 x 64 tiles, one persistent dataflow
// kernel on the FP64 matrix cores); the solution ends up in row ld + 1.  `workspace`: chol_workspace_bytes(ld) bytes
// of device memory; its first int is the fail flag -- zeroed by the caller, set to 1 when a pivot is not positive,
// 2 when the kernel abandoned a wait.  Enqueues one memset and one kernel; no synchronisation.
constexpr int kCholPanel = 64;
size_t chol_workspace_bytes(int ld);
// `plan` (or NULL = dense): the tile structure of the matrix.  chol_plan_create(m, pattern): m = ld / 64 tile rows, bit k of
// pattern[i] = tile (i, k) of the lower triangle may be non-zero (the fill of the factorisation is added by the symbolic
// pass); tiles outside the plan are never read, written or waited for, and tile columns that do not depend on each
// other are factored concurrently.  chol_symbolic is that pass on its own (for choosing an ordering on the host).
struct CholPlan;
CholPlan *chol_plan_create(int m, const unsigned long long *pattern);
void chol_plan_destroy(CholPlan *p);
void chol_plan_info(const CholPlan *p, int *n_tiles, int *depth, int *n_updates = nullptr);
void chol_symbolic(int m, const unsigned long long *pattern, unsigned long long *lmask, int *depth, int *n_tiles);
void chol_factor_solve(double *A, int ld, void *workspace, hipStream_t st, const CholPlan *plan = nullptr);
// Split solve over the ranks of a sharded system (chol.hip): the segments of a nested-dissection order occupy the tile
// columns [0, c0), the separators [c0, m); `own` = this rank's segment columns.  Phase 1 eliminates them into the separator
// block, phase 2 solves the (rank-summed) separator block, phase 3 substitutes back into the rank's segments.
CholPlan *chol_plan_create_split(int m, const unsigned long long *pattern, int c0, unsigned long long own, int phase);
void chol_split_phase(double *A, int ld, void *workspace, hipStream_t st, const CholPlan *plan, int phase, int c0);
// Order of the nf diagonal blocks (bs rows each; pairs = the off-diagonal blocks (hi, lo) that are non-zero) of a system
// with m tile rows: natural, or a nested dissection of a (cyclic) band when that shortens the chain of dependent tile
// columns by a quarter or more (chol.hip).  force_parts: -1 = choose, 1 = natural, P > 1 = P segments.
struct CholOrder {
  std::vector<int> slot_of;  // natural block index -> position
  int parts = 1, cyclic = 0, sep = 0, depth = 0, tiles = 0;
  std::vector<unsigned long long> pattern;  // tile pattern in the chosen order (chol_plan_create's input)
  std::vector<int> part_of;  // position -> segment index, -1 = separator (empty: natural order, no segments)
  int seg_slots = 0;         // positions [0, seg_slots) are the segments', the rest the separators'
};
CholOrder chol_choose_order(int nf, int bs, const std::vector<std::pair<int, int>> &pairs, int m, int force_parts = -1);

// searchByBoW(KeyFrame*, Frame*) (matcher.cpp:449-559) for B (reference key-frame, current frame) pairs whose current
// frames are resident in a frame store (slots slot0 .. slot0 + B - 1): Frame::computeBow (vocabulary transform, levelsup)
// on the device, the common-node walk on the host (this call synchronises `st` once), one k_node_replay launch with a
// workgroup per pair reading the frames' descriptors and angles in place.  dev_assigned [B][cap]: key-frame feature index
// held by each frame feature or -1; dev_n_matches [B].  (match.hip; the key-frame side is host memory.)
struct RefKeyFrame {
  int n;
  const uint8_t *valid;  // [n] the feature's map point exists and is not bad
  const uint8_t *desc;   // [n][32]
  const float *angle;    // [n]
  const vo_bow_view *nodes;
};
int bow_search_resident(const vo_vocab *v, vo_frames *frames, int slot0, int B, const RefKeyFrame *kfs, float ratio, int check_rot,
                        int levelsup, int32_t *dev_assigned, int cap, int32_t *dev_n_matches, hipStream_t st);
// device views of a frame store's per-slot arrays (guided.hip)
struct FrameStoreView {
  int cap;
  const uint8_t *desc;  // [slots][cap][32]
  const float *angle;   // [slots][cap]
  const int *n;         // [slots]
};
FrameStoreView frame_store_view(const vo_frames *h);

// vo_set_option(VO_OPT_HAMMING_KERNEL) (match.hip): 0 = matrix-core form, 1 = VALU form
void set_hamming_kernel(int v);

// Device addresses of the handles' sticky error flags (NULL before the first use): vo_tracker copies them into its
// result block so that one download answers "pose + counts + did anything overflow" (orb.hip, guided.hip).
const int *orb_error_flag(const vo_orb *h);
const int *guided_error_flag(const vo_frames *h);

}  // namespace vo
