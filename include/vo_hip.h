/*
 * vo_hip.h -- C-ABI of the MI355X (gfx950) implementation of the ORB front end and the
 * bundle-adjustment back end of guisongchen/vo_slam_test.
 *
 * This is the drop-in boundary: the reference has no plugin/FFI interface (its hot path is three
 * C++ classes inside libvo.so), so each entry point below names the reference member it replaces.
 * C++ shims with the reference's class names live in include/myslam_shim/ and call only this
 * header.  Plain pointers and sizes; no C++, torch, OpenCV, Eigen or Ceres types.
 *
 * Conventions
 *   - every function returns VO_OK (0) or a negative vo_status; nothing throws.
 *   - "host" pointers are ordinary CPU memory, "dev" pointers are HIP device memory (HBM).
 *   - handles are not re-entrant (like ORBextractor, which mutates mvImagePyramid); use one
 *     handle per host thread.  Each handle owns one HIP stream unless one is supplied.
 *   - the library fails loudly (VO_ERR_NO_DEVICE) when no gfx950 device is usable; there is no
 *     CPU fallback.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VO_OK = 0,
  VO_ERR_INVALID = -1,   /* bad argument */
  VO_ERR_NO_DEVICE = -2, /* no usable HIP device */
  VO_ERR_HIP = -3,       /* a HIP runtime call failed (vo_last_error() has the text) */
  VO_ERR_CAPACITY = -4,  /* an internal or caller buffer is too small */
  VO_ERR_STOPPED = -5    /* local BA aborted by the stop flag before the first solve */
} vo_status;

const char *vo_last_error(void);
int vo_device_count(void);
const char *vo_version(void);
/* The stateless entry points (vo_hamming_matrix, vo_match_*, vo_pose_only_solve, vo_sim3_solve, vo_pose_graph_solve,
 * vo_chol_solve, ...) keep grow-only device scratch per calling host thread, so that the hot path neither allocates
 * nor frees (the reference's Matcher / Optimizer are called from three threads).  This
 * frees what the calling thread holds and returns the number of bytes; the buffers grow again on the next call.
 * Call it before a worker thread exits, or after a one-off large problem (a 500-key-frame pose graph holds 150 MB). */
size_t vo_release_thread_scratch(void);

/* cv::KeyPoint memory layout (28 bytes) so that the shim can reinterpret the output array. */
typedef struct {
  float x, y;     /* pt, level-0 pixel coordinates */
  float size;     /* 31 * scale[octave], truncated (ORBextractor.cpp:842) */
  float angle;    /* degrees [0,360) from the intensity centroid */
  float response; /* FAST score */
  int32_t octave;
  int32_t class_id; /* -1 */
} vo_keypoint;

/* ------------------------------------------------------------------------------------------
 * ORB extractor  --  replaces ORB_SLAM2::ORBextractor (include/myslam/ORBextractor.h:45-111,
 * src/ORBextractor.cpp).
 * ------------------------------------------------------------------------------------------ */
typedef struct vo_orb vo_orb;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.cpp:414-476; constructed once at visualOdometry.cpp:31 with 1000,1.2,8,20,7). */
int vo_orb_create(vo_orb **out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                  int min_th_fast);
void vo_orb_destroy(vo_orb *h);
/* run on a caller-owned hipStream_t instead of the handle's own stream (NULL = default stream) */
int vo_orb_set_stream(vo_orb *h, void *hip_stream);
/* VO_ORB_OPT_FUSED_LEVEL_PASS (default 0): 1 = every pyramid level whose geometry allows it is processed by ONE kernel that
 * stages a tile once and produces the level's FAST cell results, its blurred tiles and the next level's rows from it
 * (csrc/orb_level_pass.inc); 0 = the three separate kernels per level.  The results are bit-identical; on MI355X the fused
 * pass measures 6 % slower than the three kernels (profiles/r05_ab_fused.txt: more vector instructions and three workgroup
 * barriers), which is why it is opt-in. */
/* VO_ORB_OPT_EARLY_LEVEL0 (default 0): 1 = outside the instrumented mode (vo_orb_set_timing) FAST on the cells of level 0 and
 * level 0's blur -- the work that needs the caller's image only -- start on the extractor's side stream next to the resize chain;
 * 0 = FAST on all cells behind the pyramid, the whole blur on the side stream next to it.  Results are identical; measured
 * (tools/ext_schedule_ab.py): 2.482 against 2.499 ms per 1024 frames for the extraction alone, no difference for the tracked
 * step (every kernel involved is bound by instruction issue: there is nothing to overlap), hence opt-in. */
/* VO_ORB_OPT_BLUR_KERNEL (default 0): which kernel blurs the levels (identical planes either way): 0 = k_blur_mfma, both passes
 * of the 7x7 filter as int8 matrix-core products with banded weight matrices (levels of at least 64 x 16 pixels with 16-byte
 * aligned rows; the others fall back by themselves), 1 = k_blur_groups, the dot-product form on the VALU. */
/* VO_ORB_OPT_DESCRIBE_BLUR (default 0): 0 = no blurred pyramid is made: the descriptor kernel (k_describe_win) brings the
 * 45 x 45 raw window of a key-point into LDS once, takes the orientation moments from it, blurs it in place (int8 matrix-core
 * products, the arithmetic of the plane kernels: identical descriptors) and runs the tests on it;
 * vo_orb_get_level(blurred = 1) then makes the planes when it is asked for them.  1 = the blurred planes are made for every
 * level of every frame (VO_ORB_OPT_BLUR_KERNEL picks the kernel) and the descriptor kernel (k_describe) reads its windows from
 * them.  The fused level pass (VO_ORB_OPT_FUSED_LEVEL_PASS) makes its own blurred tiles: with it the planes are always used. */
enum { VO_ORB_OPT_FUSED_LEVEL_PASS = 1, VO_ORB_OPT_EARLY_LEVEL0 = 2, VO_ORB_OPT_BLUR_KERNEL = 3, VO_ORB_OPT_DESCRIBE_BLUR = 4 };
int vo_orb_set_option(vo_orb *h, int option, int value);
/* (tests and tools) the fused pass's plan for one level of a width x height image: out = {takes the fused pass, tile pitch,
 * tile rows, score rows, workgroups per frame, LDS bytes per workgroup, survivor-list entries, 0} */
int vo_orb_debug_level_pass(vo_orb *h, int width, int height, int level, int out[8]);

/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors (ORBextractor.h:61-76) */
int vo_orb_levels(const vo_orb *h);
float vo_orb_scale_factor(const vo_orb *h);
int vo_orb_scale_factors(const vo_orb *h, float *scale /*nlevels*/, float *inv_scale /*nlevels or NULL*/);
int vo_orb_features_per_level(const vo_orb *h, int *quota /*nlevels*/);
/* upper bound on key-points per frame (the oct-tree may return a few more than nfeatures) */
int vo_orb_max_keypoints(const vo_orb *h);

/* ORBextractor::operator()(image, mask [ignored], keypoints, descriptors)
 * (ORBextractor.cpp:1051-1112; called from Frame::Frame, frame.cpp:22).
 * Host 8-bit grey image in, host key-points (level-major, oct-tree order) and 32-byte descriptors
 * out.  image == NULL or width/height <= 0 returns VO_OK without touching the outputs (:1054). */
int vo_orb_extract(vo_orb *h, const uint8_t *image, int width, int height, int stride,
                   vo_keypoint *keypoints, uint8_t *descriptors, int capacity, int *n_keypoints);

/* The same operator over a batch of frames that is already resident in HBM (bench / pipelines
 * that keep frames on the device).  Frame f starts at dev_images + f*frame_stride_bytes.
 * Outputs are device arrays: key-points [n_frames][capacity], descriptors [n_frames][capacity][32],
 * counts [n_frames].  Asynchronous on the handle's stream; call vo_orb_sync before reading. */
int vo_orb_extract_batch_dev(vo_orb *h, const uint8_t *dev_images, int n_frames, int width,
                             int height, int stride, size_t frame_stride_bytes,
                             vo_keypoint *dev_keypoints, uint8_t *dev_descriptors, int capacity,
                             int32_t *dev_counts);
int vo_orb_sync(vo_orb *h);

/* mvImagePyramid[level] of frame `frame` of the last call (ORBextractor.h:85), unpadded, to host.
 * blurred != 0 returns the 7x7 Gaussian-blurred plane the descriptors were sampled from. */
int vo_orb_get_level(vo_orb *h, int frame, int level, int blurred, uint8_t *dst, int dst_stride,
                     int *width, int *height);
/* FAST candidates of one level of one frame after the per-cell NMS, in reference order
 * (vToDistributeKeys, ORBextractor.cpp:826-833): x,y relative to the (16,16) border, score. */
int vo_orb_get_candidates(vo_orb *h, int frame, int level, float *x, float *y, float *response,
                          int capacity, int *n);
/* per-level key-point counts of one frame of the last call */
int vo_orb_get_level_counts(vo_orb *h, int frame, int32_t *counts /*nlevels*/);

/* Per-stage timing with HIP events recorded on the handle's stream around each stage of every
 * subsequent call (bench.py's live roofline measurement).  Stages: 0 pyramid (all resize launches),
 * 1 FAST cells, 2 oct-tree, 3 offsets (always 0: folded into the descriptor kernel), 4 blur,
 * 5 orientation+descriptor.  With the fused level pass (vo_orb_set_option) stage 0 carries the whole chain of level passes
 * -- pyramid + FAST + blur -- and stages 1 and 4 are empty.  vo_orb_get_timing
 * synchronises, adds the elapsed times of the calls since the last reset to ms[VO_ORB_STAGES],
 * returns the number of timed calls in *n_calls and resets the accumulators. */
#define VO_ORB_STAGES 6
/* Stage hook: `hook(stage, hip_stream, user)` is called on the calling host thread inside vo_orb_extract* each time the launches
 * of a stage (numbering of vo_orb_get_timing: 0 pyramid, 1 FAST, 2 oct-tree, 4 blur, 5 descriptors) have been enqueued on
 * `hip_stream` -- the place to record an event or to make another stream wait, e.g. to start the all-pairs matching of the
 * PREVIOUS batch (matrix cores + HBM writes) on a second stream exactly when this batch's FAST (vector-issue bound) starts:
 * bench.py's extract + match leg does that.  The hook must not synchronise.  NULL removes it. */
typedef void (*vo_orb_stage_hook)(int stage, void *hip_stream, void *user);
int vo_orb_set_stage_hook(vo_orb *h, vo_orb_stage_hook hook, void *user);
int vo_orb_set_timing(vo_orb *h, int enabled);
int vo_orb_get_timing(vo_orb *h, double *ms /*VO_ORB_STAGES*/, int *n_calls);

/* ------------------------------------------------------------------------------------------
 * Matcher  --  replaces the arithmetic of myslam::Matcher (include/myslam/matcher.h:9-45,
 * src/matcher.cpp).  The pointer-graph gather/scatter stays in the C++ shim.
 * ------------------------------------------------------------------------------------------ */

/* Matcher::computeDistance (matcher.cpp:1240-1256) for all pairs: D[i*nb + j] = Hamming(A_i,B_j).
 * Device pointers; descriptors are 32 bytes each, 4-byte aligned. */
int vo_hamming_matrix_dev(const uint8_t *dev_a, int na, const uint8_t *dev_b, int nb,
                          uint16_t *dev_d, void *hip_stream);
/* batched: pair p uses A + p*a_stride, B + p*b_stride, D + p*d_stride (strides in elements of
 * the respective arrays' rows: descriptors / descriptors / uint16) */
int vo_hamming_matrix_batch_dev(const uint8_t *dev_a, int na, size_t a_stride, const uint8_t *dev_b,
                                int nb, size_t b_stride, uint16_t *dev_d, size_t d_stride,
                                int n_pairs, void *hip_stream);
/* host convenience wrapper (copies in/out) -- also what MapPoint::computeDescriptor's N x N
 * median selection (mappoint.cpp:140-151) consumes */
int vo_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *d);

/* MapPoint::computeDescriptor (mappoint.cpp:118-179): for each set s (one map point) of
 * descriptors desc[offsets[s] .. offsets[s+1]) -- the rows kf->descriptors_.row(idx) of its good
 * observers -- the index (within the set) of the descriptor with the smallest median Hamming
 * distance to all members (:140-172); -1 for an empty set (:136-137 returns without choosing).
 * ONE kernel launch for the whole batch (the N x N distances never leave the GPU); at most 1024
 * observations per map point.  Host pointers.  (The scalar Matcher::computeDistance of two
 * descriptors is a host inline popcount in the shim: a GPU round trip per pair would be absurd.) */
int vo_median_descriptor(const uint8_t *desc, int n_sets, const int32_t *offsets, int32_t *best_idx);


/* one frame's features as the matcher sees them (Frame members, frame.h:26-45) */
typedef struct vo_frame_view_s {
  int32_t n;
  const float *x, *y;       /* unKeypoints_[i].pt */
  const int32_t *octave;    /* unKeypoints_[i].octave */
  const float *angle;       /* unKeypoints_[i].angle */
  const float *uright;      /* uRight_[i] */
  const uint8_t *desc;      /* descriptors_ (n x 32) */
  float xmin, ymin, xmax, ymax; /* Camera::xMin_.. (camera.cpp:40-43) */
} vo_frame_view;

/* ------------------------------------------------------------------------------------------
 * Device-resident frames  --  Frame::undistortKeyPoints / findDepth / assignFeaturesToGrid
 * (frame.cpp:36-133) and Frame/KeyFrame::getFeaturesInArea (frame.cpp:199-247,
 * keyframe.cpp:268-312) on the GPU: key-points stay in HBM between the extractor and the matcher.
 * A vo_frames handle stores up to max_frames frames of up to max_features (<= 16384) features:
 * undistorted key-points, octave, angle, uRight, depth, descriptors and the 64 x 48 grid as CSR.
 * ------------------------------------------------------------------------------------------ */
typedef struct vo_frames vo_frames;
int vo_frames_create(vo_frames **out, int max_frames, int max_features);
void vo_frames_destroy(vo_frames *h);
int vo_frames_capacity(const vo_frames *h, int *max_frames, int *max_features);
/* Camera (camera.cpp:8-47): intrinsics = fx, fy, cx, cy, bf; dist_coef = k1, k2, p1, p2, k3 (NULL or
 * k1 == 0: no undistortion, frame.cpp:41-45); image bounds xMin = yMin = 0, xMax = width, yMax = height. */
int vo_frames_set_camera(vo_frames *h, const float intrinsics[5], const float dist_coef[5], float width,
                         float height);
/* Frame::Frame post-processing (frame.cpp:27-32) of n_frames extractor outputs into slots
 * slot0 ..: dev_keypoints [n_frames][capacity], dev_descriptors [n_frames][capacity][32], dev_counts
 * [n_frames] exactly as vo_orb_extract_batch_dev wrote them.  depth_kind 0: no depth (uRight = depth
 * = -1); 1: float32 metres [h][depth_pitch_bytes]; 2: uint16 raw, metres = raw * inv_depth_scale
 * (Mat::convertTo, visualOdometry.cpp:162-163).  cv::undistortPoints = OpenCV 3.x: 5 fixed-point
 * iterations in double.  Asynchronous on hip_stream. */
int vo_frames_build_dev(vo_frames *h, int slot0, int n_frames, const vo_keypoint *dev_keypoints,
                        const uint8_t *dev_descriptors, const int32_t *dev_counts, int capacity,
                        const void *dev_depth, int depth_kind, size_t depth_frame_stride_bytes,
                        int depth_pitch_bytes, float inv_depth_scale, void *hip_stream);
/* Frame::Frame (frame.cpp:14-34) for one host image in one call: ORB extraction (:22), undistortKeyPoints,
 * findDepth, assignFeaturesToGrid (:27-31) on the device -- the image goes up once, nothing comes back but
 * the raw key-points (Frame::keypoints_, written here) and what vo_frames_download then returns for the slot
 * (unKeypoints_, uRight_, depth_, descriptors_, the grid).  The extractor handle is switched to the calling
 * thread's stream.  depth as in vo_frames_build_dev (host pointer). */
int vo_frames_construct(vo_frames *h, int slot, vo_orb *orb, const uint8_t *image, int width, int height, int stride,
                        const void *depth, int depth_kind, int depth_pitch_bytes, float inv_depth_scale,
                        vo_keypoint *keypoints, int capacity, int *n_keypoints);
/* one already post-processed frame from host arrays (builds the grid); depth may be NULL */
int vo_frames_upload(vo_frames *h, int slot, const vo_frame_view *view, const float *depth,
                     void *hip_stream);
/* copy a slot back (tests / shims): any output may be NULL; cell_start has 64*48+1 entries
 * (cell = ix * 48 + iy), cell_items n entries (feature indices in push_back order). */
int vo_frames_download(vo_frames *h, int slot, int *n, float *x, float *y, int32_t *octave, float *angle,
                       float *uright, float *depth, uint8_t *desc, int32_t *cell_start,
                       uint16_t *cell_items, void *hip_stream);

/* Frame::getFeaturesInArea(u, v, radius, minLevel, maxLevel) (frame.cpp:199-247) and
 * KeyFrame::getFeaturesInArea(u, v, radius) (keyframe.cpp:268-312; min_level = max_level = NULL) for
 * n_queries windows of one stored frame: out_idx[q * max_out + k] = k-th feature index of window q in
 * the reference's order (grid column by column, cells top to bottom, insertion order inside a cell),
 * out_count[q] = number of features in the window (may exceed max_out: the list is then truncated).
 * Host arrays in and out; runs on the calling thread's stream.  The searches above do not go through
 * this list -- they walk the same windows inside their candidate kernel -- it is the reference's
 * accessor for other callers and the direct test of the grid. */
int vo_frames_features_in_area(vo_frames *h, int slot, int n_queries, const float *u, const float *v,
                               const float *radius, const int32_t *min_level, const int32_t *max_level,
                               int32_t *out_idx, int max_out, int32_t *out_count);

/* Guided (window) matching on device-resident frames, batched over frames: frame f of the call
 * searches slot slot0 + f with its own block of queries (query q of frame f at index
 * f * stride + q of every array; all device pointers).
 *   mode 0  searchByProjection(Frame*, Frame*)     matcher.cpp:18-148   aux = 1/z, level = last octave
 *   mode 1  searchByProjection(Frame*, MapPoints)  :274-353             aux = trackProj_uR_, viewcos
 *   mode 2  searchByProjection(Frame*, KeyFrame*)  :150-272             radius, dist_threshold
 *   mode 3  fuseMapPoints candidate search         :1064-1106           aux = projected uR; radius = threshold
 *   mode 4  searchBySim3 / fuseByPose inner search :756-786, :1196-1213 max_dist
 *   mode 5  searchByProjection(KeyFrame*, Sim3&)   :356-447 (Q-M1)      radius = th
 * Modes 0, 1, 2, 5 claim features: dev_assigned [n_frames][max_features] in/out (query index per
 * feature or -1; mode 5 starts from -1), dev_feature_mask [n_frames][max_features] or NULL =
 * blocked / holds-a-map-point / occupied on entry.  Modes 3, 4: dev_best_idx [n_frames][stride].
 * dev_n_matches [n_frames].  Every query owns 32 candidate records; pool_per_frame sizes the per-frame
 * overflow area that windows with more gated candidates spill into (0 = 16 per query);
 * vo_match_guided_status reports an exhausted overflow area after the stream has been synchronised. */
typedef struct {
  int32_t n_queries;          /* queries per frame (upper bound when n_per_frame is given) */
  int32_t stride;             /* distance between the query blocks of consecutive frames (>= n_queries) */
  const int32_t *n_per_frame; /* device, or NULL; a NEGATIVE entry leaves the frame out of the call: nothing of it is
                                 read or written (assigned, best_idx, n_matches keep their values) */
  const uint8_t *flags;       /* bit 0 valid, bit 1 the query's map point has observations (see below) */
  const float *u, *v, *aux;
  const int32_t *level;
  const float *angle, *viewcos; /* modes 0, 2 (checkRot) / mode 1 */
  const uint8_t *desc;
} vo_guided_queries;
typedef struct {
  int32_t mode;
  float radius, bf, ratio, dist_threshold;
  int32_t direction, check_rot, n_levels, max_dist;
  const float *scale_factors; /* host, n_levels entries */
  /* mode 0 only, or NULL: trackWithMotion's retry (visualOdometry.cpp:241-245) decided on the device.  A frame that ends
   * the call with fewer than retry_below matches has its dev_assigned entries set back to -1 and
   * retry_n_per_frame[f] = its query count; every other frame gets -1 -- the n_per_frame array of a second call
   * (same queries, wider radius) that then looks at those frames only. */
  int32_t retry_below;
  int32_t *retry_n_per_frame; /* device [n_frames] */
} vo_guided_params;
int vo_match_guided_dev(vo_frames *h, int slot0, int n_frames, const vo_guided_queries *q,
                        const vo_guided_params *p, const uint8_t *dev_feature_mask, int32_t *dev_assigned,
                        int32_t *dev_best_idx, int32_t *dev_n_matches, size_t pool_per_frame,
                        void *hip_stream);
int vo_match_guided_status(vo_frames *h, void *hip_stream);

/* q_flags bit 0: query valid (map point exists, not outlier, projects inside the image);
 *         bit 1: its map point has observe_cnt_ > 0 (claims the feature for later queries). */

/* Matcher::searchByProjection(Frame* cur, Frame* last, radius, checkRot) (matcher.cpp:18-148).
 * direction: 1 = forward, 2 = backward, 0 = neither (:47-48,:70-75).  assigned[cur.n] in/out:
 * query index matched to each feature or -1.  blocked[cur.n]: feature already holds an observed
 * map point.  Host-array form of vo_match_guided_dev mode 0: the frame view and the queries are
 * uploaded in one block, window search, gates, Hamming distances and the ordered claim replay run on
 * the device, only assigned[] comes back.  Returns the match count in *n_matches. */
int vo_match_frame_projection(const vo_frame_view *cur, int nq, const uint8_t *q_flags,
                              const float *q_u, const float *q_v, const float *q_invz,
                              const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                              float radius, float bf, int direction, int check_rot, int n_levels,
                              const float *scale_factors, const uint8_t *blocked,
                              int32_t *assigned, int *n_matches);

/* Matcher::searchByProjection(Frame*, const vector<MapPoint*>&, thRadius) (matcher.cpp:274-353);
 * ratio = Matcher::ratio_. */
int vo_match_local_map(const vo_frame_view *cur, int nq, const uint8_t *q_flags, const float *q_u,
                       const float *q_v, const float *q_ur, const int32_t *q_level,
                       const float *q_viewcos, const uint8_t *q_desc, float th_radius, float ratio,
                       const float *scale_factors, const uint8_t *blocked, int32_t *assigned,
                       int *n_matches);

/* Matcher::searchByProjection(Frame*, KeyFrame*, radius, distThreshold, found, checkRot)
 * (matcher.cpp:150-272, relocalisation top-up).  Query i = key-frame map point i, already projected
 * (flag bit 0: exists, not bad, not in `found`, z > 0, inside the image and its distance range);
 * q_level = MapPoint::predictScale.  has_map_point[cur.n]: the feature holds any map point (:218);
 * features assigned earlier in the call are skipped as well. */
int vo_match_frame_keyframe(const vo_frame_view *cur, int nq, const uint8_t *q_flags, const float *q_u,
                            const float *q_v, const int32_t *q_level, const float *q_angle,
                            const uint8_t *q_desc, float radius, float dist_threshold, int check_rot,
                            const float *scale_factors, const uint8_t *has_map_point, int32_t *assigned,
                            int *n_matches);

/* DBoW3::FeatureVector of one frame as CSR: node ids ascending, node k owns
 * feat[start[k] .. start[k+1]) (frame.h:50, keyframe.h). */
typedef struct {
  int32_t n_nodes;
  const uint32_t *node_id;
  const int32_t *start;
  const uint32_t *feat;
} vo_bow_view;

/* Matcher::searchByBoW(KeyFrame*, Frame*, matches, checkRot) (matcher.cpp:449-559; mode 0:
 * match[b.n] = A index held by each B feature) and searchByBoW(KeyFrame*, KeyFrame*, ...)
 * (:561-677; mode 1: match[a.n] = B index of each A feature).  a_valid / b_valid: the feature has
 * a good map point.  ratio = Matcher::ratio_.  Runs on the device (k_node_replay: the node-by-node
 * claims in the reference's order, distances computed on the fly); b.n <= 16384, else VO_ERR_CAPACITY. */
int vo_match_bow(const vo_frame_view *a, const uint8_t *a_valid, const vo_bow_view *a_nodes,
                 const vo_frame_view *b, const uint8_t *b_valid, const vo_bow_view *b_nodes, int mode,
                 float ratio, int check_rot, int32_t *match, int *n_matches);

/* The same search for n_pairs pairs in ONE launch (one workgroup per pair; frames that appear in several pairs are
 * uploaded once): the searchByBoW calls over the relocalisation candidates (visualOdometry.cpp:354-371) or the loop
 * candidates (loopClosing.cpp:182).  Arrays of n_pairs pointers; match[p] has b[p]->n (mode 0) or a[p]->n (mode 1)
 * entries; results identical to n_pairs single calls. */
int vo_match_bow_batch(int n_pairs, const vo_frame_view *const *a, const uint8_t *const *a_valid,
                       const vo_bow_view *const *a_nodes, const vo_frame_view *const *b, const uint8_t *const *b_valid,
                       const vo_bow_view *const *b_nodes, int mode, float ratio, int check_rot, int32_t *const *match,
                       int *n_matches);

/* Matcher::searchForTriangulation(kf1, kf2, matchIdxs, F12, checkRot) (matcher.cpp:867-1010,
 * called at localMapping.cpp:187).  *_has_map_point: feature already triangulated (skipped).
 * F12 row-major; (ex, ey) = camera centre 1 projected into key-frame 2 (:887-891).
 * match12[a.n] = B index or -1.  Same device kernel as vo_match_bow (epipole / epipolar-line gates per
 * candidate); scale_factors: 8 entries. */
int vo_match_triangulation(const vo_frame_view *a, const uint8_t *a_has_map_point, const vo_bow_view *a_nodes,
                           const vo_frame_view *b, const uint8_t *b_has_map_point, const vo_bow_view *b_nodes,
                           const double F12[9], float ex, float ey, const float *scale_factors,
                           int check_rot, int32_t *match12, int *n_matches);

/* matching part of Matcher::fuseMapPoints (matcher.cpp:1012-1133; the map mutation :1108-1127
 * stays in the shim).  Query = candidate map point projected into the key-frame (flag bit 0:
 * passes the gates of :1029-1062); best_idx[nq] = feature to fuse with or -1. */
int vo_match_fuse(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u,
                  const float *q_v, const float *q_ur, const int32_t *q_level, const uint8_t *q_desc,
                  float threshold, const float *scale_factors, int32_t *best_idx, int *n_matches);

/* Bag-of-words transform: DBoW3::Vocabulary::transform(features, bowVec, featVec, levelsup) as called by
 * Frame::computeBow (frame.cpp:248-253) and KeyFrame::computeBow (keyframe.cpp:394-398), levelsup = 3.
 * The vocabulary tree (parsed on the host from the DBoW3 file) is uploaded once as flat arrays: the
 * children of node i are children[child_start[i] .. child_start[i+1]), node 0 is the root,
 * word_id[i] >= 0 marks a leaf (a word) with weight node_weight[i], node_desc holds one 32-byte ORB
 * descriptor per node.  Per feature: the word, its weight, and the node reached at level
 * depth_L - levelsup (the key of the FeatureVector that searchByBoW / searchForTriangulation walk).
 * The (word -> summed weight) map, its L1 normalisation and the (node -> feature list) map stay in
 * the shim: they are std::map insertions. */
typedef struct vo_vocab vo_vocab;
int vo_vocab_create(vo_vocab **out, int n_nodes, int depth_L, const int32_t *child_start, const int32_t *children,
                    const uint8_t *node_desc, const double *node_weight, const int32_t *word_id);
void vo_vocab_destroy(vo_vocab *v);
int vo_bow_transform(const vo_vocab *v, int n, const uint8_t *desc, int levelsup, int32_t *word_id, double *weight,
                     int32_t *node_id);

/* The vocabulary file itself (DBoW3::Vocabulary(path), vo_run.cpp:87): DBoW3's binary stream
 * (Vocabulary::toStream; plain, or -- what Vocabulary::save(path) writes by default, reference map.cpp:94 -- in QuickLZ
 * level-1 chunks, decoded by a restatement of the published decoder: no library-written file was available to pin it,
 * a stream that does not follow the format is refused), its cv::FileStorage form (.yml / .yml.gz: Vocabulary::save(cv::FileStorage&),
 * what DBoW3::Vocabulary(path) falls back to) or the ORB-SLAM2 text format; builds the device tree. */
int vo_vocab_load(const char *path, vo_vocab **out, int *n_nodes, int *n_words, int *branching_k, int *depth_L);
/* Map::score (map.cpp:335-376): L1 similarity of the query BoW vector (ascending word ids) with every
 * candidate's (CSR: candidate c owns cand_words / cand_values [cand_start[c], cand_start[c+1])); one
 * launch for all candidates of detectLoopCandidates / detectRelocalizationCandidates (:210-333). */
int vo_bow_score(int n_query, const int32_t *query_words, const double *query_values, int n_candidates,
                 const int32_t *cand_start, const int32_t *cand_words, const double *cand_values, double *scores);

/* Sim3Solver (src/sim3Solver.cpp): every RANSAC hypothesis of Sim3Solver::iterate in one launch -- Horn's
 * closed form (computeSim3 :179-252) for the three sampled correspondences triplets[3k..3k+2] and
 * checkInliers (:254-280) over all n correspondences.  The random triplets come from the caller (the
 * reference draws them with rand(), :119-137), as does the sequential pick "first hypothesis whose
 * inlier count beats the threshold" (:141-160).  max_err = the reference's INTEGER thresholds
 * (vector<int>, 9.210 sigma^2 truncated).  Per hypothesis: counts[k], inlier_flags[k * n ..] (or NULL),
 * sims[13 k ..] = R12 row-major (9), t12 (3), s12.
 * All six correspondence arrays NULL: the correspondences the calling host thread passed with its previous call (same n)
 * are still on the device and are used again -- a caller that evaluates one hypothesis per call (to keep its random
 * generator in step with the reference's) uploads them once per Sim3Solver::iterate, not once per trip. */
int vo_sim3_ransac_eval(int n, const double *cam1_points, const double *cam2_points, const double *pixels1,
                        const double *pixels2, const int32_t *max_err1, const int32_t *max_err2,
                        const float cam4[4], int n_hypotheses, const int32_t *triplets, int fix_scale,
                        int32_t *counts, uint8_t *inlier_flags, double *sims);

/* LocalMapping::createNewMapPoints' linear triangulation (localMapping.cpp:234-251), batched: normalised
 * observations xn1 / xn2 [n][2], Tcw1 (3 x 4 float, row-major) of the current key-frame, Tcw2 one pose or
 * one per pair.  points [n][3]; ok[i] = 0 where |x_3| < 1e-8 (:245-246 skips the pair).  The right
 * singular vector is taken from the eigen-decomposition of A^T A in double: agreement with cv::SVD's
 * float Jacobi is to float rounding (stated tolerance 1e-4 relative), not bit-exact. */
int vo_triangulate(int n, const float *xn1, const float *xn2, const float Tcw1[12], const float *Tcw2,
                   int per_pair_pose2, float *points, uint8_t *ok);

/* searchForTriangulation of the current key-frame `a` against ALL its neighbours in one launch (the loop of
 * LocalMapping::createNewMapPoints, localMapping.cpp:160-190: <= 10 neighbours, one call each in the reference): pair p
 * searches b[p] with F12 = F[9p .. 9p+8] and the epipole (ex[p], ey[p]); match12[p] has a->n entries.  The calls are
 * independent in the reference (matchIdxs is local to a neighbour), so the results equal n_pairs single calls. */
int vo_match_triangulation_batch(int n_pairs, const vo_frame_view *a, const uint8_t *a_has_map_point,
                                 const vo_bow_view *a_nodes, const vo_frame_view *const *b,
                                 const uint8_t *const *b_has_map_point, const vo_bow_view *const *b_nodes, const double *F,
                                 const float *ex, const float *ey, const float *scale_factors, int check_rot,
                                 int32_t *const *match12, int *n_matches);

/* cv::cvtColor(CV_RGB2GRAY / CV_BGR2GRAY [/ RGBA / BGRA]) of VisualOdometry::createFrame
 * (visualOdometry.cpp:146-159), OpenCV 3.x fixed point. */
int vo_rgb_to_gray(const uint8_t *src, long long n_pixels, int channels, int first_is_red, uint8_t *dst);
int vo_rgb_to_gray_dev(const uint8_t *dev_src, long long n_pixels, int channels, int first_is_red,
                       uint8_t *dev_dst, void *hip_stream);

/* The reference harness's I/O (test/vo_run.cpp): associate.txt (:24-58), cv::imread of 8-bit colour
 * / 16-bit depth PNGs (:108-109; zlib + PNG filters, non-interlaced), the trajectory files (:154-232,
 * Twc7 = translation + quaternion x y z w per pose, Eigen's default stream format) and the tracking-
 * time report (:138-151). */
typedef struct vo_dataset vo_dataset;
int vo_dataset_open(vo_dataset **out, const char *dataset_dir, int max_frames);
int vo_dataset_size(const vo_dataset *d);
int vo_dataset_entry(const vo_dataset *d, int i, const char **rgb_time, const char **rgb_path,
                     const char **depth_time, const char **depth_path);
void vo_dataset_close(vo_dataset *d);
int vo_png_info(const char *path, int *width, int *height, int *channels, int *bit_depth);
int vo_png_read(const char *path, int as_bgr, void *dst, size_t dst_bytes);
int vo_trajectory_write(const char *path, int n, const char *const *timestamps, const double *Twc7);
int vo_tracking_time_stats(const double *seconds, int n_tracked, double *median, double *mean);

/* Loop-closure searches.  All three project map points into a key-frame and take, per point, the
 * best Hamming match among KeyFrame::getFeaturesInArea(u, v, th * scale[level]) with octave in
 * [level - 1, level]; flag bit 0 of a query = it passed the projection gates of the routine.
 *
 * vo_match_area_best: queries independent -- the inner search of Matcher::searchBySim3
 * (matcher.cpp:756-786, 821-851; max_dist = 100) and of Matcher::fuseByPose (:1196-1213;
 * max_dist = 50; the map mutation :1215-1230 stays in the shim). */
int vo_match_area_best(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u,
                       const float *q_v, const int32_t *q_level, const uint8_t *q_desc, float th,
                       const float *scale_factors, int max_dist, int32_t *best_idx, int *n_matches);

/* Matcher::searchByProjection(KeyFrame*, Sim3&, loopMapPoints, matchMapPoints, th)
 * (matcher.cpp:356-447).  occupied[kf.n]: matchMapPoints[k] non-null on entry; assigned[kf.n] =
 * query that claimed the feature, or -1.  The reference's skip test `matchMapPoints[j]` (:422)
 * indexes with the candidate counter, not the feature index; reproduced. */
int vo_match_sim3_projection(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u,
                             const float *q_v, const int32_t *q_level, const uint8_t *q_desc, int th,
                             const float *scale_factors, const uint8_t *occupied, int32_t *assigned,
                             int *n_matches);

/* Matcher::searchBySim3 (matcher.cpp:679-865): q1 = map point of feature i of key-frame 1 projected
 * into key-frame 2 (flag 0 also for features without a usable point or already matched, :722-727),
 * q2 the reverse; match12[kf1.n] = feature of key-frame 2, kept only when both directions agree. */
int vo_match_sim3_mutual(const vo_frame_view *kf1, const vo_frame_view *kf2, const uint8_t *q1_flags,
                         const float *q1_u, const float *q1_v, const int32_t *q1_level, const uint8_t *q1_desc,
                         const uint8_t *q2_flags, const float *q2_u, const float *q2_v,
                         const int32_t *q2_level, const uint8_t *q2_desc, float th,
                         const float *scale_factors1, const float *scale_factors2, int32_t *match12,
                         int *n_matches);

/* ------------------------------------------------------------------------------------------
 * Optimizer  --  replaces myslam::Optimizer (include/myslam/optimizer_ceres.h:12-97,
 * src/optimizer_ceres.cpp) including the Ceres solve it delegates to.
 * Poses are se3 tangent vectors [upsilon(3); omega(3)] = Sophus SE3::log(Tcw); cam = fx,fy,cx,cy,bf.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t iterations;  /* LM iterations run (successful or not) */
  int32_t accepted;
  int32_t termination; /* 0 max iterations, 1 function tol., 2 parameter tol., 3 gradient tol., 4 failure */
  int32_t reserved;
  double initial_cost, final_cost, final_radius;
} vo_lm_summary;

/* Optimizer::solvePoseOnlySE3(Frame*) (optimizer_ceres.cpp:157-314) for `n_problems` independent
 * frames in one launch (one workgroup per frame).  Problem p owns observations
 * [offsets[p], offsets[p+1]).  obs = (u, v, uR) with uR < 0 for monocular observations.
 * pose [n_problems][6] in/out, outlier[total obs] out, n_inliers[n_problems] out.
 * summaries: NULL or [n_problems][2] (Huber round, plain round).  Host pointers. */
int vo_pose_only_solve(int n_problems, const int32_t *offsets, const double *points,
                       const double *obs, const double *inv_sigma, const double cam[5],
                       double *poses, uint8_t *outlier, int32_t *n_inliers,
                       vo_lm_summary *summaries);
/* device-resident variant (all pointers device memory, asynchronous on hip_stream) */
int vo_pose_only_solve_dev(int n_problems, const int32_t *dev_offsets, int max_obs,
                           const double *dev_points, const double *dev_obs,
                           const double *dev_inv_sigma, const double *dev_cam5, double *dev_poses,
                           uint8_t *dev_outlier, int32_t *dev_n_inliers,
                           vo_lm_summary *dev_summaries, void *hip_stream);

/* the same with explicit ranges: problem p owns observations [ranges[2p], ranges[2p] + ranges[2p+1])
 * (frames laid out at a fixed stride by vo_track_gather_dev) */
int vo_pose_only_solve_ranges_dev(int n_problems, const int32_t *dev_ranges, const double *dev_points,
                                  const double *dev_obs, const double *dev_inv_sigma, const double *dev_cam5,
                                  double *dev_poses, uint8_t *dev_outlier, int32_t *dev_n_inliers,
                                  vo_lm_summary *dev_summaries, void *hip_stream);

/* Tracking glue on the device (a tracked frame stays in HBM from the extractor to the pose):
 *   vo_track_project_dev  projection prologue of searchByProjection(Frame*, Frame*) (matcher.cpp:41-64):
 *       dev_Tcw [n_frames][12] = rotation row-major + translation of the current pose estimate,
 *       dev_points [n_frames][stride][3] the last frame's map points, dev_point_flags bit 0 = point
 *       exists and is no outlier, bit 1 = observe_cnt_ > 0; writes the mode-0 query arrays
 *       (flags, u, v, 1/z) of vo_match_guided_dev.  xmin .. ymax are the int-truncated image bounds.
 *   vo_track_scatter_dev  frame->mappoints_[k] = the query that claimed feature k: copies its world
 *       point into dev_feature_points [n_frames][max_features][3], sets dev_feature_has (and
 *       dev_feature_observed = flag bit 1, the `blocked` mask of the next search).
 *   vo_track_gather_dev   the gather of solvePoseOnlySE3 (optimizer_ceres.cpp:181-202): features that
 *       hold a map point, in feature order, into points / obs (x, y, uRight) / 1/sigma at
 *       [f * max_features ...) with dev_ranges [n_frames][2] = (start, count) for
 *       vo_pose_only_solve_ranges_dev; dev_index (or NULL) = the feature index of every observation. */
int vo_track_project_dev(int n_frames, int n_queries, int stride, const double *dev_Tcw, const double *dev_points,
                         const uint8_t *dev_point_flags, const float cam4[4], int xmin, int xmax, int ymin,
                         int ymax, uint8_t *dev_q_flags, float *dev_u, float *dev_v, float *dev_invz,
                         void *hip_stream);
int vo_track_scatter_dev(vo_frames *h, int slot0, int n_frames, const int32_t *dev_assigned,
                         const double *dev_query_points, const uint8_t *dev_query_flags, int stride,
                         double *dev_feature_points, uint8_t *dev_feature_has, uint8_t *dev_feature_observed,
                         void *hip_stream);
int vo_track_gather_dev(vo_frames *h, int slot0, int n_frames, const double *dev_feature_points,
                        const uint8_t *dev_feature_has, const float *scale_factors, int n_levels,
                        double *dev_points, double *dev_obs, double *dev_inv_sigma, int32_t *dev_ranges,
                        int32_t *dev_index, void *hip_stream);
/* vo_track_scatter_dev followed by vo_track_gather_dev in one launch (same outputs, bit for bit): what the
 * tracker runs between a search and the pose solve that consumes its matches. */
int vo_track_scatter_gather_dev(vo_frames *h, int slot0, int n_frames, const int32_t *dev_assigned,
                                const double *dev_query_points, const uint8_t *dev_query_flags, int stride,
                                double *dev_feature_points, uint8_t *dev_feature_has,
                                uint8_t *dev_feature_observed, const float *scale_factors, int n_levels,
                                double *dev_points, double *dev_obs, double *dev_inv_sigma,
                                int32_t *dev_ranges, int32_t *dev_index, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * The tracked-frame pipeline as one object  --  VisualOdometry::trackWithMotion + trackLocalMap
 * (visualOdometry.cpp:228-251, 286-300, 745-775, 864-886) for a batch of `batch` independent camera
 * streams resident in HBM: extraction, Frame::Frame post-processing, searchByProjection against the last
 * frame's map points (radius 15), solvePoseOnlySE3, cullingOutliersBeforeLocalMap, Frame::isInFrame +
 * MapPoint::predictScale for the local map points WITH THE REFINED POSE, searchByProjection against
 * them (thRadius 3, ratio 0.8), solvePoseOnlySE3, inlier count.  One call enqueues the whole sequence
 * (27 kernel launches) without host synchronisation; batch = 1 with host images is the single-stream form
 * (Frame construction to pose in one call).
 *
 * Streams: the searches and pose solves run on `stream` (NULL: a high-priority stream of the tracker),
 * the extraction on `extract_stream` (NULL: a stream of the tracker, or `stream` itself when
 * single_stream != 0).  Several trackers that share one extract_stream take turns on the extraction
 * while each other's searches and solves run next to it (two batches in flight: bench.py's regime).
 *
 * The retry of trackWithMotion -- fewer than 20 matches: clear, search again at 2 x radius (:241-245) -- runs on the
 * device for the frames that need it (a per-frame flag, a second candidate / replay pass over the flagged frames only).
 * Status bits VO_TRACK_FEW_MATCHES / VO_TRACK_FEW_INLIERS tell the caller that the reference would have left
 * trackWithMotion (:247, :253); the route it takes then is vo_tracker_track_ref_keyframe below (trackRefKeyFrame,
 * :256-277).  Not covered: relocalisation (:307-402, PnP RANSAC over BoW candidates -- control plane around
 * vo_match_bow / vo_match_frame_keyframe / vo_pose_only_solve), and the map-side steps between the two stages:
 * trackLocalMap derives localKeyframes_ / localMappoints_ from frame_curr_->mappoints_ AFTER the first stage's culling
 * (updateLocalKeyFrames / updateLocalMapPoints, :286-291), whereas this call takes the local map BEFORE it starts.  A
 * caller that needs the reference's order runs the two stages as two calls: vo_tracker_track with an empty local map
 * (max_local points, n = 0), derives the local map from VO_TRACKER_ASSIGNED_LAST, then vo_tracker_track_local_map.
 * ------------------------------------------------------------------------------------------ */
typedef struct vo_tracker vo_tracker;
typedef struct {
  int32_t batch, width, height;
  int32_t nfeatures, nlevels, ini_th_fast, min_th_fast; /* 0: 1000, 8, 20, 7 (visualOdometry.cpp:31) */
  float scale_factor;                                   /* <= 1: 1.2 */
  float intrinsics[5];                                  /* fx, fy, cx, cy, bf */
  float dist_coef[5];                                   /* k1, k2, p1, p2, k3 */
  int32_t has_distortion;
  float inv_depth_scale;                                /* metres = raw * inv_depth_scale (depth_kind 2) */
  int32_t max_last, max_local;                          /* capacity: last-frame / local map points per frame */
  int32_t max_features;                                 /* feature slots per frame; 0: the extractor's bound */
  int32_t single_stream;
  void *stream, *extract_stream;                        /* hipStream_t or NULL */
} vo_tracker_config;
typedef struct {
  float radius;     /* searchByProjection(frame, last frame): 15 */
  float th_radius;  /* searchByProjection(frame, local points): 3 (5 just after relocalisation, :793-794) */
  float ratio;      /* Matcher(0.8) */
  int32_t direction; /* matcher.cpp:70-75: 0 none, 1 forward, 2 backward */
  float ref_ratio;   /* vo_tracker_track_ref_keyframe: Matcher(0.7) (:262); <= 0: 0.7 */
  int32_t no_retry;  /* 0: a frame with < 20 matches is cleared and searched again at 2 x radius (:241-245), on the
                        device, before the solve; 1: no second search (the status bit still reports < 20) */
} vo_tracker_params;
enum { VO_TRACK_FEW_MATCHES = 1, VO_TRACK_FEW_INLIERS = 2 };
int vo_tracker_create(vo_tracker **out, const vo_tracker_config *cfg);
void vo_tracker_destroy(vo_tracker *t);
int vo_tracker_info(const vo_tracker *t, int *batch, int *max_features, int *max_keypoints, int *n_levels);
vo_orb *vo_tracker_extractor(vo_tracker *t); /* the handles the tracker owns (accessors, tests) */
vo_frames *vo_tracker_frames(vo_tracker *t);
void *vo_tracker_stream(vo_tracker *t);
/* State of the map as the next batch sees it; host arrays [batch][n][...], copied before the call
 * returns.  Last frame (frame_last_->mappoints_, visualOdometry.cpp:232-238): Tcw12 [batch][12] =
 * rotation row-major + translation of frame_curr_->Tcw_ = Tcl_ * frame_last_->Tcw_; per map point its
 * world position, flags (bit 0: exists and is no outlier, bit 1: observe_cnt_ > 0), the octave, angle
 * and descriptor of the last frame's feature.  Local map (localMappoints_, :745-775): world position,
 * normal vector, minDistance_ / maxDistance_ (mappoint.h), flags (bit 0: exists and is not bad, bit 1:
 * observe_cnt_ > 0), descriptor; link[i] = index of the same MapPoint in the last-frame list or -1
 * (NULL: none) -- a point the first search matched carries visualIdxOfFrame_ == frame id and is
 * skipped (:765). */
int vo_tracker_set_last_frame(vo_tracker *t, int n, const double *Tcw12, const double *points, const uint8_t *flags,
                              const int32_t *octave, const float *angle, const uint8_t *desc);
int vo_tracker_set_local_map(vo_tracker *t, int n, const double *points, const double *normals,
                             const float *min_distance, const float *max_distance, const uint8_t *flags,
                             const int32_t *link, const uint8_t *desc);
/* One batch.  _dev: 8-bit grey images [batch] in device memory (row pitch, frame stride in bytes), depth as
 * in vo_frames_build_dev (depth_kind 0 none, 1 float32 metres, 2 uint16 raw).  vo_tracker_track: the same
 * from host memory (width x height, tightly packed; uploads included: the image ahead of the extraction, the depth -- first
 * read by the frame build -- on a copy stream of the tracker behind the extraction's launches.  Page-locked host buffers must
 * stay untouched until vo_tracker_results or a synchronisation; pageable ones are read before the call returns).
 * Asynchronous; params NULL: 15, 3, 0.8, 0. */
int vo_tracker_track_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                         const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                         const vo_tracker_params *params);
int vo_tracker_track(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind,
                     const vo_tracker_params *params);
/* The two stages as two calls (the reference's order: updateLocalKeyFrames / updateLocalMapPoints derive the local map
 * from the matches of the first stage, visualOdometry.cpp:286-291).  vo_tracker_track_first[_dev]: Frame construction,
 * trackWithMotion's search (+ retry), solvePoseOnlySE3, cullingOutliersBeforeLocalMap; vo_tracker_results then returns
 * the first solve's pose, n_tracked = the culling's observed-inlier count (:250), n_inliers = the solve's return value,
 * the status bits of :247 / :253.  vo_tracker_track_local_map: searchLocalMapPoints, solvePoseOnlySE3 and the inlier
 * count on the state the first stage left, against the local map set in between (vo_tracker_set_local_map). */
int vo_tracker_track_first(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind,
                           const vo_tracker_params *params);
int vo_tracker_track_first_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                               const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                               const vo_tracker_params *params);
int vo_tracker_track_local_map(vo_tracker *t, const vo_tracker_params *params);
/* VisualOdometry::trackRefKeyFrame (visualOdometry.cpp:256-277) -- the route taken when trackWithMotion fails --
 * followed by trackLocalMap (first_stage_only = 0) or on its own (1): Frame construction, Frame::computeBow (vocabulary
 * transform on the device), Matcher(0.7).searchByBoW(keyframe_trackRef_, frame) (k_node_replay, one workgroup per
 * frame of the batch), `match_num < 15` -> VO_TRACK_FEW_MATCHES, the key-frame's map points into the frame's slots,
 * pose = frame_last_->Tcw_, solvePoseOnlySE3, cullingOutliersBeforeLocalMap.  vo_tracker_set_ref_keyframe: per frame of
 * the batch its reference key-frame's features [batch][n]: the map point of every feature (world position; flags bit 0:
 * exists and is not bad, bit 1: observe_cnt_ > 0), the feature's angle and descriptor, its DBoW3::FeatureVector
 * (nodes[batch], levelsup 3) and Tcw12 [batch][12] = frame_last_->Tcw_.  It REPLACES the last-frame state
 * (vo_tracker_set_last_frame): the key-frame's map points take the place of frame_last_->mappoints_ for the rest of the
 * pipeline (`link` of vo_tracker_set_local_map then indexes the key-frame's features).  The vocabulary must outlive
 * the tracker's use of it.  The common-node walk is host work: this route synchronises once inside the call. */
int vo_tracker_set_ref_keyframe(vo_tracker *t, const vo_vocab *vocab, int n, const double *Tcw12, const double *points,
                                const uint8_t *flags, const float *angle, const uint8_t *desc,
                                const vo_bow_view *const *nodes);
int vo_tracker_track_ref_keyframe(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind,
                                  const vo_tracker_params *params, int first_stage_only);
int vo_tracker_track_ref_keyframe_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                                      const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                                      const vo_tracker_params *params, int first_stage_only);
/* Waits for the batch and copies out (any pointer may be NULL): poses as se3 [batch][6] and as Tcw
 * [batch][12]; n_tracked = inliers of the second solve whose map point has observations (inliers_num_,
 * :289-300); n_inliers = the second solve's return value; the two searches' match counts; status bits.
 * Reports sticky stage errors (dropped key-points, exhausted candidate pools) as VO_ERR_CAPACITY. */
int vo_tracker_results(vo_tracker *t, double *poses6, double *Tcw12, int32_t *n_tracked, int32_t *n_inliers,
                       int32_t *n_matches_last, int32_t *n_matches_local, int32_t *status);
/* intermediate state of the last batch (tests, shims): [batch][max_features] / [batch][max_local] arrays */
enum {
  VO_TRACKER_ASSIGNED_LAST = 0,      /* int32: last-frame point index per feature after search 1, or -1 */
  VO_TRACKER_ASSIGNED_LOCAL = 1,     /* int32: local point index per feature claimed by search 2, or -1 */
  VO_TRACKER_POSE_FIRST = 2,         /* double [batch][6]: pose after the first solve */
  VO_TRACKER_INLIERS_FIRST = 3,      /* int32 [batch]: return value of the first solve */
  VO_TRACKER_OBSERVED_INLIERS_FIRST = 4, /* int32 [batch]: cullingOutliersBeforeLocalMap's return value */
  VO_TRACKER_FEATURE_HAS_POINT = 5,  /* uint8: frame->mappoints_[i] != nullptr at the end */
  VO_TRACKER_FEATURE_POINTS = 6,     /* double [..][3]: that map point's position */
  VO_TRACKER_LOCAL_FLAGS = 7, VO_TRACKER_LOCAL_U = 8, VO_TRACKER_LOCAL_V = 9, VO_TRACKER_LOCAL_UR = 10,
  VO_TRACKER_LOCAL_LEVEL = 11, VO_TRACKER_LOCAL_VIEWCOS = 12, /* Frame::isInFrame's outputs per local point */
  VO_TRACKER_KEYPOINT_COUNTS = 13,   /* int32 [batch] */
  VO_TRACKER_FEATURE_OUTLIER = 14    /* uint8: frame->outliers_[i] after the second solve (valid after the local-map stage) */
};
int vo_tracker_get(vo_tracker *t, int what, void *dst, size_t dst_bytes);
int vo_tracker_sync(vo_tracker *t);
/* HIP events on the launching streams around the six stages of every subsequent batch (bench.py):
 * 0 extraction, 1 frame post-processing, 2 search vs last frame, 3 pose solve + culling, 4 isInFrame +
 * search vs local map, 5 pose solve + count.  vo_tracker_get_timing synchronises, returns the summed
 * milliseconds since the last call and the number of timed batches, and resets. */
#define VO_TRACKER_STAGES 6
int vo_tracker_set_timing(vo_tracker *t, int enabled);
int vo_tracker_get_timing(vo_tracker *t, double *ms /*VO_TRACKER_STAGES*/, int *n_calls);

/* Optimizer::solveLoopSim3(keyframe_curr, keyframe_match, inlierMappoints, Scm, fixScaleFlag)
 * (optimizer_ceres.cpp:810-1030) with PoseOnlySim3 / PoseOnlyInverseSim3 (optimizer_ceres.h:211-267):
 * problem 1 (Huber sqrt(10), <= 10 iterations), chi2 > 10 rejection in both images, then 10 (or 5
 * when nothing was rejected) more iterations on the survivors and a final test of every match.
 * Batched: problem p owns matches offsets[p] .. offsets[p+1].  Per match: the map point in the
 * matched key-frame's camera frame (cam_match) with the current key-frame's pixel and 1/sigma, and
 * the current key-frame's camera-frame point (cam_curr) with the matched key-frame's pixel and
 * 1/sigma (:846-878).  poses[6p..] in/out = [angle-axis; t] of Scm, scales[p] its scale
 * (fix_scale: constant, loopClosing.cpp:15).  outlier[i] = 1: inlierMappoints entry nulled.
 * n_inliers[p] = return value; 0 with pose and scale untouched when fewer than 10 matches survive
 * problem 1 (:950-951).  summaries[2p].reserved reports the phase reached: 1 = returned after problem 1
 * (Scm must not be written), 2 = both problems ran (Scm = the result, even with 0 inliers). */
int vo_sim3_solve(int n_problems, const int32_t *offsets, const double *cam_match, const double *pix_curr,
                  const double *inv_sigma_curr, const double *cam_curr, const double *pix_match,
                  const double *inv_sigma_match, const double camera[4], int fix_scale, double *poses,
                  double *scales, uint8_t *outlier, int32_t *n_inliers, vo_lm_summary *summaries /*2 per problem or NULL*/);

/* The solve inside Optimizer::solvePoseGraphLoop (optimizer_ceres.cpp:1036-1305; cost functor
 * PoseGraphLoop, optimizer_ceres.h:269-325): key-frame i = Sim3 S_iw as unit quaternion quats[4i..]
 * (Eigen coefficient order x, y, z, w; ceres::EigenQuaternionParameterization), translation
 * trans[3i..], scale scales[i] (constant: fix_scale must be 1, as loopClosing.cpp:15 always passes);
 * edge e (edge_i -> edge_j) measures S_ji = (q_meas, t_meas, s_meas) (:1094-1236).  fixed_node is
 * keyframe_match (:1238-1240).  No loss, LM, exact solve of the normal equations
 * (SPARSE_NORMAL_CHOLESKY in the reference; here a dense blocked Cholesky with the trailing update on
 * the FP64 matrix cores), max_iterations = 20 in the reference (:1253).  6 * (nodes - 1) <= 4096. */
int vo_pose_graph_solve(int n_nodes, double *quats, double *trans, const double *scales, int fixed_node,
                        int n_edges, const int32_t *edge_i, const int32_t *edge_j, const double *q_meas,
                        const double *t_meas, const double *s_meas, int fix_scale, int max_iterations,
                        vo_lm_summary *summary);

/* Map-point re-anchoring after the pose graph (optimizer_ceres.cpp:1281-1301):
 * out = S_wr[ref] * (S_rw[ref] * p); a Sim3 is 8 doubles: quaternion (x, y, z, w), translation, scale.
 * ref_node[i] < 0 leaves the point unchanged (bad points are skipped, :1284-1285). */
int vo_sim3_reanchor_points(int n_points, const double *points_in, const int32_t *ref_node, int n_nodes,
                            const double *S_rw, const double *S_wr, double *points_out);

/* Dense symmetric positive definite solve on the device (the blocked Cholesky behind
 * vo_pose_graph_solve): A row-major, lower triangle read and overwritten by its factor, b -> x. */
int vo_chol_solve(int n, double *A_rowmajor_lower, double *b);

/* The same solve through the split (per-rank segment) form that a sharded global BA can use (vo_ba_set_shard +
 * vo_ba_set_allreduce with VO_BA_OPT_SEGMENTS), with the n_ranks shards
 * emulated one after the other on this GPU -- a test entry: the 64-column tile columns [0, c0_tiles) hold segments that
 * are independent of each other (col_part[j] = the segment of tile column j, owned by rank col_part[j] % n_ranks), the
 * remaining tile columns the separators.  Per rank: eliminate the own segments into the separator block; the separator
 * blocks are summed (the all-reduce); every rank solves the separators and substitutes back into its segments.  b -> x. */
int vo_chol_solve_split(int n, const double *A_rowmajor_lower, double *b, int c0_tiles, const int32_t *col_part, int n_ranks);

/* Bundle-adjustment problem handle (the arrays Optimizer::solveLocalBAPoseAndPoint gathers at
 * optimizer_ceres.cpp:446-592).  Edges may be given in any order; they are grouped by point
 * internally (stable).  cam_fixed[c] != 0 <=> SetParameterBlockConstant (:578-579). */
typedef struct vo_ba vo_ba;
int vo_ba_create(vo_ba **out, int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points,
                 const double *points, int n_edges, const int32_t *edge_cam,
                 const int32_t *edge_point, const double *edge_obs, const double *edge_inv_sigma,
                 const double cam[5]);
void vo_ba_destroy(vo_ba *h);
/* A NEW problem in an existing handle: the per-key-frame caller (localMapping.cpp:38 builds a different local window every
 * time) keeps ONE handle per thread and resets it instead of create / destroy -- stream, device buffers (grow-only, with
 * headroom), page-locked staging, shard / callback / options stay, so that nothing is allocated or freed on the hot path
 * (hipFree synchronises the whole device, the tracking thread's streams included).  Arguments as vo_ba_create.  Caller-owned
 * reduce buffers (vo_ba_set_reduce_buffers) were sized for the previous problem: the reset forgets them, set them again.  A
 * problem that is rejected (VO_ERR_INVALID: an edge out of range) leaves the handle's previous problem in place. */
int vo_ba_reset(vo_ba *h, int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points, const double *points,
                int n_edges, const int32_t *edge_cam, const int32_t *edge_point, const double *edge_obs,
                const double *edge_inv_sigma, const double cam[5]);
int vo_ba_set_stream(vo_ba *h, void *hip_stream);
/* restrict this handle to the points p with p % n_shards == shard (multi-GPU: one process per
 * GPU, each owning a shard; cameras replicated).  Must precede any solve.
 * (With VO_BA_OPT_SEGMENTS set, an all-reduce callback and a large reduced system whose key-frame order has
 * nested-dissection segments, a point belongs to the rank of the segment it touches instead -- see vo_ba_set_allreduce.) */
int vo_ba_set_shard(vo_ba *h, int shard, int n_shards);
/* Per-handle options; must precede the first use of the handle (any solve, vo_ba_set_state, vo_ba_debug_order) and be the
 * same on every rank of a sharded solve.  A sharded handle with an all-reduce callback checks that with one small
 * handshake all-reduce when it is first used: ranks that disagree on the options, the shard count or the problem fail
 * with VO_ERR_INVALID instead of waiting for each other in mismatched collectives.
 *   VO_BA_OPT_SEGMENTS                 1: per-rank segment factorisation (see vo_ba_set_allreduce); default 0
 *   VO_BA_OPT_COLLECTIVES_AT_ONE_RANK  1: a handle of ONE shard with a callback runs the sharded form of the loop (for
 *                                      bringing a collective up on a one-GPU machine: every call is a sum over one rank)
 *   VO_BA_OPT_ORDER_PARTS              force the number of nested-dissection parts of a large system's key-frame order
 *                                      (1 = natural order; -1 = choose, the default) */
enum { VO_BA_OPT_SEGMENTS = 1, VO_BA_OPT_COLLECTIVES_AT_ONE_RANK = 2, VO_BA_OPT_ORDER_PARTS = 3 };
int vo_ba_set_option(vo_ba *h, int option, int value);
/* Process-wide developer knobs: VO_OPT_BA_GRAPH 1 = replay the LM iteration sequence of an unsharded solve from a
 * hipGraph (default 0: eager launches are faster on this stack); VO_OPT_POSE_BLOCK = threads per
 * frame of the pose-only solver (0 = automatic, 64, 128, 256); VO_OPT_HAMMING_KERNEL = which form of the all-pairs Hamming
 * kernel vo_hamming_matrix* launch: 0 (default) = int8 matrix-core dot products, 1 = the xor / popcount VALU form (identical
 * results); VO_OPT_BA_PAIRS_KERNEL = which kernel gathers the reduced camera system of a LARGE problem
 * (more than 21 free key-frames): 0 (default) = k_ba_pairs_lds, blocks staged through LDS, 1 = k_ba_pairs, lane = couple (the two
 * agree to rounding, not bitwise: the order of the sums differs). */
enum { VO_OPT_BA_GRAPH = 1, VO_OPT_POSE_BLOCK = 2, VO_OPT_HAMMING_KERNEL = 3, VO_OPT_BA_PAIRS_KERNEL = 4 };
int vo_set_option(int option, int value);
/* Multi-GPU from C/C++: the all-reduce the sharded LM loop needs (sum of n doubles at dev_buf over all
 * shards, in place, ordered on hip_stream; returns 0).  With RCCL this is
 *   ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, (hipStream_t)stream)
 * over xGMI.  Once set, vo_ba_solve / vo_ba_local_ba[_enqueue] on a sharded handle run the whole LM
 * schedule with exactly two calls of it per iteration (the reduced camera system, 6 scalars); every
 * rank must make the same calls.  Without it those entry points reject a sharded handle
 * (VO_ERR_INVALID) instead of solving from partial sums.
 * Per-rank segment factorisation (opt-in, vo_ba_set_option(h, VO_BA_OPT_SEGMENTS, 1); large reduced systems only): every rank eliminates the
 * nested-dissection segments it owns, and the calls per iteration become four -- the camera-block extras, the separator
 * block after the elimination, the step, 6 scalars; vo_ba_linearize / vo_ba_step refuse such a handle.  The option
 * takes effect when the handle is first used and must be the same on every rank (checked by a handshake).  Measured with
 * emulated ranks it does more work per rank than the default, which is why it is not the default.
 * VO_BA_OPT_COLLECTIVES_AT_ONE_RANK: a handle of ONE shard with a callback runs the same sharded form of the loop (for
 * bringing a collective up on a one-GPU machine: every call is a sum over one rank; tests/test_gpu_rccl.py). */
typedef int (*vo_allreduce_fn)(void *user, double *dev_buf, size_t n_doubles, void *hip_stream);
int vo_ba_set_allreduce(vo_ba *h, vo_allreduce_fn fn, void *user);
int vo_ba_set_state(vo_ba *h, const double *poses, const double *points);
int vo_ba_get_state(vo_ba *h, double *poses, double *points);
int vo_ba_n_free_cams(const vo_ba *h);
/* The key-frame order a large reduced system (6 nf + 1 > 128) is factored in, chosen when the handle is first used:
 * out = {parts, cyclic, separator key-frames, dependent tile columns on the longest chain, tiles of L, tile rows,
 * tile products L(i,k) L(j,k)^T of the factorisation (2 x 64^3 flop each), first separator tile column of a handle in
 * segment mode or 0}; parts == 1: the natural order.
 * (Tests and tools; all zero tile rows for LDS-sized systems.) */
int vo_ba_debug_order(vo_ba *h, int out[8]);

/* Full Optimizer::solveLocalBAPoseAndPoint numerics (:530-755): Huber LM (5 iterations), float
 * chi2 classification, plain LM (10 iterations) on the inliers, final chi2 pass.
 * stop: NULL or the address of the reference's LIVE `bool stopFlag` (read as a byte, so a C++ caller passes
 * reinterpret_cast<const volatile unsigned char *>(&stopFlag)), polled exactly at :594 and :612.
 * edge_erase[n_edges] (host) out in the caller's edge order.  summaries: NULL or [2]. */
int vo_ba_local_ba(vo_ba *h, const volatile unsigned char *stop, uint8_t *edge_erase,
                   vo_lm_summary *summaries);
/* The same schedule split into "queue everything on the handle's stream" and "wait + fetch results",
 * so that several independent problems (handles) overlap on one GPU. */
int vo_ba_local_ba_enqueue(vo_ba *h, const volatile unsigned char *stop);
int vo_ba_local_ba_finish(vo_ba *h, uint8_t *edge_erase, vo_lm_summary *summaries);
/* One Ceres-style LM solve (ceres::Solve with DENSE_SCHUR at :604 / :699) on the current state.
 * huber_* <= 0 disables the loss.  edge_active: NULL or host mask in caller's edge order. */
int vo_ba_solve(vo_ba *h, double huber_mono, double huber_stereo, int max_iterations,
                const uint8_t *edge_active, vo_lm_summary *summary);

/* Split-phase interface used by the multi-GPU driver and the kernel-level tests:
 *   vo_ba_linearize   evaluate residuals/Jacobians at the current state, assemble the point
 *                     blocks and this shard's contribution to the reduced camera system into the
 *                     device buffer returned by vo_ba_reduced_system(): packed doubles
 *                     [ S (6nf x 6nf row-major) | b (6nf) | cost | |x|^2 ... ] -- the buffer that is
 *                     all-reduced (sum) across shards each LM iteration.
 *   vo_ba_step        Cholesky-solve the (already reduced) system, back-substitute, form the
 *                     candidate state, evaluate its cost -> second small reduced buffer.
 *   vo_ba_update      accept / reject, trust-region radius update, convergence flags.
 * All three are asynchronous on the handle's stream; no host synchronisation inside. */
int vo_ba_lm_begin(vo_ba *h, double huber_mono, double huber_stereo, int max_iterations,
                   const uint8_t *edge_active);
int vo_ba_linearize(vo_ba *h);
int vo_ba_step(vo_ba *h);
int vo_ba_update(vo_ba *h);
int vo_ba_lm_end(vo_ba *h, vo_lm_summary *summary);
/* let the caller own the two all-reduce payload buffers (e.g. torch tensors handed to
 * torch.distributed); sizes as reported by vo_ba_reduced_system / vo_ba_reduced_cost.
 * Must precede vo_ba_lm_begin; vo_ba_reset forgets them (they are sized by the problem). */
int vo_ba_set_reduce_buffers(vo_ba *h, double *dev_system, double *dev_cost);
/* device pointers + element counts of the two all-reduce payloads */
int vo_ba_reduced_system(vo_ba *h, double **dev_ptr, size_t *n_doubles);
int vo_ba_reduced_cost(vo_ba *h, double **dev_ptr, size_t *n_doubles);
/* pieces of the local-BA schedule for drivers that run the LM loop themselves (multi-GPU):
 *   vo_ba_classify(h, 0)  float chi2 test of optimizer_ceres.cpp:618-689 on the current state; the
 *                         device-side edge mask becomes "inlier", outliers are remembered
 *   vo_ba_lm_begin_inliers  like vo_ba_lm_begin but keeps that device-side mask (problem 2, :691-699)
 *   vo_ba_classify(h, 1)  final pass :703-755 (adds to the remembered outliers)
 *   vo_ba_get_edge_outliers  the remembered mask in the caller's edge order (edgeErase) */
int vo_ba_classify(vo_ba *h, int final_pass);
int vo_ba_lm_begin_inliers(vo_ba *h, double huber_mono, double huber_stereo, int max_iterations);
int vo_ba_get_edge_outliers(vo_ba *h, uint8_t *edge_erase);
/* copy out the undamped reduced camera system of the current linearisation (tests):
 * S [6nf*6nf], b [6nf], cost.  point_damping is added to every point-block diagonal. */
int vo_ba_debug_schur(vo_ba *h, double huber_mono, double huber_stereo, double point_damping,
                      const uint8_t *edge_active, double *S, double *b, double *cost);

/* Developer instrumentation: s_memrealtime stamps (100 MHz ticks) written by the BA kernels of a
 * -DVO_BA_STAMPS build (tools/ba_stamps.py); all zero in the product build.  out[48]. */
int vo_ba_debug_stamps(vo_ba *h, unsigned long long *out);

/* SE3 helpers the shims need (Sophus SE3::exp / log as used at :163,:257,:474,:787) */
int vo_se3_exp(const double xi[6], double R_rowmajor[9], double t[3]);
int vo_se3_log(const double R_rowmajor[9], const double t[3], double xi[6]);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
