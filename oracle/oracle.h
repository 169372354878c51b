/*
 * oracle.h -- CPU restatement of the guisongchen/vo_slam_test hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load liboracle.so.  The
 * product path (vo_slam_test_amd/, include/vo_hip.h) never includes, links or calls
 * anything in this directory.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and
 * cannot be compiled in this image (OpenCV, Ceres, Sophus, Eigen, DBoW3 absent;
 * SURVEY.md section 8c).  Every function below cites the reference file:line it
 * follows; third-party arithmetic (OpenCV 3.x FAST / resize / GaussianBlur /
 * fastAtan2 / cvRound, Ceres 1.x LM + Schur, old Sophus SE3) is restated from the
 * published algorithms and the exact variant chosen is written next to it.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orc_keypoint;

/* ---------------- ORB extractor (reference src/ORBextractor.cpp) -------------- */

#define ORC_MAX_LEVELS 16

typedef struct {
  int nfeatures, nlevels, ini_th, min_th;
  float scale_factor;
  float scale[ORC_MAX_LEVELS];      /* mvScaleFactor      ORBextractor.cpp:421-427 */
  float inv_scale[ORC_MAX_LEVELS];  /* mvInvScaleFactor   :430-436 */
  int quota[ORC_MAX_LEVELS];        /* mnFeaturesPerLevel :440-451 */
  int umax[16];                     /* circular patch     :457-475 */
  int8_t pattern[1024];             /* bit_pattern_31_    :154-412 (fed from fixture) */
} orc_orb_params;

void orc_orb_params_init(orc_orb_params *p, int nfeatures, float scale_factor, int nlevels,
                         int ini_th, int min_th, const int8_t *pattern1024);
void orc_level_size(const orc_orb_params *p, int w, int h, int level, int *lw, int *lh);

/* cv::resize(INTER_LINEAR) for CV_8UC1, OpenCV 3.x fixed-point path. */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                          int dh, int dstride);
/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) for CV_8UC1, OpenCV <=3.4.0 8-bit kernel. */
void orc_gaussian7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
/* cv::FAST(img, kps, threshold, nms=true), TYPE_9_16.  Returns count; x,y,score arrays sized cap. */
int orc_fast9_16(const uint8_t *img, int cols, int rows, int stride, int threshold, int nms,
                 int *xs, int *ys, int *scores, int cap);
float orc_fast_atan2(float y, float x);
int orc_cv_round_f(float v);
/* cos/sin of a float angle [rad] as used by the descriptor: deterministic double polynomial,
 * rounded to float (spec in DESIGN.md "steered BRIEF trig").  *_libm = glibc cosf/sinf. */
void orc_cos_sin_f(float angle_rad, float *c, float *s);
void orc_cos_sin_f_libm(float angle_rad, float *c, float *s);

/* ComputeKeyPointsOctTree cell loop for one level (:771-837): candidates in reference order,
 * coordinates relative to (minBorderX, minBorderY) exactly like vToDistributeKeys. */
int orc_level_candidates(const orc_orb_params *p, const uint8_t *img, int w, int h, int stride,
                         float *cx, float *cy, float *cresp, int cap);
/* DistributeOctTree (:545-769).  Returns number of kept keys, indices into the candidate arrays in
 * list order.  Tie rule for equal-size nodes = creation order (Q-E3). */
int orc_distribute_octtree(const float *cx, const float *cy, const float *cresp, int n, int minX,
                           int maxX, int minY, int maxY, int N, int *out_idx, int cap);
float orc_ic_angle(const uint8_t *img, int stride, int px, int py, const int *umax);
void orc_orb_descriptor(const uint8_t *blur, int stride, int px, int py, float angle_deg,
                        const int8_t *pattern, uint8_t *desc32);

void orc_orb_descriptor_libm(const uint8_t *blur, int stride, int px, int py, float angle_deg,
                             const int8_t *pattern, uint8_t *desc32); /* glibc cosf / sinf variant (measurement only) */

/* Full ORBextractor::operator() (:1051-1112).  kps/desc sized for `cap` key-points.
 * If pyr_out != NULL it receives nlevels pointers to malloc'ed unpadded level images (caller frees)
 * and blur_out likewise.  Returns number of key-points or <0 on error. */
int orc_orb_extract(const orc_orb_params *p, const uint8_t *img, int w, int h, int stride,
                    orc_keypoint *kps, uint8_t *desc, int cap, int *n_per_level);

/* ---------------- Matcher (reference src/matcher.cpp, src/frame.cpp) ---------- */

int orc_hamming256(const uint8_t *a, const uint8_t *b); /* matcher.cpp:1240-1256 */
void orc_hamming_matrix(const uint8_t *A, int na, const uint8_t *B, int nb, uint16_t *D);
void orc_three_max(const int *hist_sizes, int L, int *i1, int *i2, int *i3); /* :1258-1304 */

#define ORC_GRID_COLS 64
#define ORC_GRID_ROWS 48
typedef struct {
  int n;
  const float *x, *y;      /* unKeypoints_ pt */
  const int32_t *octave;
  const float *angle;
  const float *uright;     /* uRight_ (<=0: no depth) */
  const uint8_t *desc;     /* n x 32 */
  float xmin, ymin, xmax, ymax, grid_per_px_w, grid_per_px_h;
  /* CSR grid built by orc_frame_build_grid: cell (ix,iy) -> [start[ix*48+iy], start[..+1]) */
  int *cell_start;         /* 64*48+1 */
  int *cell_items;         /* n */
} orc_frame;

void orc_frame_build_grid(orc_frame *f); /* frame.cpp:72-89 */
int orc_features_in_area(const orc_frame *f, float u, float v, float radius, int min_level,
                         int max_level, int *out, int cap); /* frame.cpp:199-247 */

/* Frame post-processing (frame_oracle.c): undistortKeyPoints frame.cpp:36-70 (cv::undistortPoints,
 * OpenCV 3.x: 5 iterations in double), findDepth :108-133, depth Mat::convertTo visualOdometry.cpp:162-163 */
void orc_undistort_points(int n, const float *x, const float *y, const float intr[4], const float dist[5],
                          float *ux, float *uy);
void orc_find_depth(int n, const float *x, const float *y, const float *ux, const float *depth_img, int w, int h,
                    int stride, float bf, float *uright, float *depth);
void orc_depth_to_float(const uint16_t *raw, int n, float inv_scale, float *out);
/* MapPoint::computeDescriptor mappoint.cpp:118-179: index of the median-best descriptor, -1 if n == 0 */
int orc_median_descriptor(const uint8_t *desc, int n);
/* Frame::isInFrame + MapPoint::predictScale (frame.cpp:145-190, mappoint.cpp:182-196, 391-401) */
void orc_is_in_frame(int n, const double pose6[6], const double *pts, const double *normals, const float *min_dist,
                     const float *max_dist, const uint8_t *valid, const float intr5[5], float xmin, float xmax, float ymin,
                     float ymax, float scale_factor_1, int n_levels, uint8_t *flags, float *u_out, float *v_out,
                     float *ur_out, int32_t *level_out, float *viewcos_out);

/* Sim3Solver (sim3Solver.cpp): Horn's closed form :179-252 and one RANSAC hypothesis per sample triplet with
 * checkInliers :254-280 (integer thresholds, float pixel arithmetic); the sequential pick :141-160 is the caller's */
void orc_sim3_horn(const double P1[9], const double P2[9], int fix_scale, double R[9], double t[3], double *s);
void orc_sim3_ransac_eval(int n, const double *pc1, const double *pc2, const double *px1, const double *px2,
                          const int32_t *maxerr1, const int32_t *maxerr2, const float cam[4], int K, const int32_t *triplets,
                          int fix_scale, int32_t *counts, uint8_t *flags, double *sims);
/* localMapping.cpp:234-251 (4 x 4 float SVD triangulation); visualOdometry.cpp:146-159 (cvtColor to grey) */
int orc_triangulate(const float xn1[2], const float xn2[2], const float T1[12], const float T2[12], float out[3]);
void orc_rgb_to_gray(const uint8_t *src, int n_px, int channels, int first_is_red, uint8_t *dst);

/* Matcher::searchByProjection(Frame*,Frame*,radius,checkRot) matcher.cpp:18-148 on flat arrays.
 * Query i carries what the reference reads from frame_last/map point i (already projected):
 * valid[i], u,v (float pixel), invz, last octave, last angle, descriptor, claimed-feature mask is
 * in/out: assigned[idx] = query index or -1; `blocked[idx]` = feature already holds an observed
 * point (mappoints_[idx]->observe_cnt_>0). */
int orc_match_frame_projection(const orc_frame *cur, int nq, const uint8_t *q_valid,
                               const float *q_u, const float *q_v, const float *q_invz,
                               const int32_t *q_octave, const float *q_angle,
                               const uint8_t *q_desc, float radius, float bf, int direction,
                               int check_rot, int n_levels, const float *scale_factors,
                               const uint8_t *blocked, int32_t *assigned);
/* Matcher::searchByProjection(Frame*, vector<MapPoint*>, thRadius) matcher.cpp:274-353. */
int orc_match_local_map(const orc_frame *cur, int nq, const uint8_t *q_valid, const float *q_u,
                        const float *q_v, const float *q_ur, const int32_t *q_level,
                        const float *q_viewcos, const uint8_t *q_desc, float th_radius,
                        float ratio, const float *scale_factors, const uint8_t *blocked,
                        int32_t *assigned);

/* DBoW3::FeatureVector of one frame as CSR (node ids ascending) */
typedef struct {
  int32_t n_nodes;
  const uint32_t *node_id;
  const int32_t *start; /* n_nodes + 1 */
  const uint32_t *feat;
} orc_bow;

int orc_match_frame_keyframe(const orc_frame *cur, int nq, const uint8_t *q_valid, const float *q_u,
                             const float *q_v, const int32_t *q_level, const float *q_angle,
                             const uint8_t *q_desc, float radius, float dist_threshold, int check_rot,
                             const float *scale_factors, const uint8_t *has_mp, int32_t *assigned);
int orc_match_bow(const orc_frame *a, const uint8_t *a_valid, const orc_bow *an, const orc_frame *b,
                  const uint8_t *b_valid, const orc_bow *bn, int mode, float ratio, int check_rot,
                  int32_t *match);
int orc_match_triangulation(const orc_frame *a, const uint8_t *a_has_mp, const orc_bow *an,
                            const orc_frame *b, const uint8_t *b_has_mp, const orc_bow *bn,
                            const double F12[9], float ex, float ey, const float *scale_factors,
                            int check_rot, int32_t *match12);
int orc_match_fuse(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                   const float *q_v, const float *q_ur, const int32_t *q_level, const uint8_t *q_desc,
                   float threshold, const float *scale_factors, int32_t *best_idx);

int orc_match_area_best(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                        const float *q_v, const int32_t *q_level, const uint8_t *q_desc, float th,
                        const float *scale_factors, int max_dist, int32_t *best_idx);
int orc_match_sim3_projection(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                              const float *q_v, const int32_t *q_level, const uint8_t *q_desc, int th,
                              const float *scale_factors, const uint8_t *occupied, int32_t *assigned);
int orc_match_sim3_mutual(const orc_frame *kf1, const orc_frame *kf2, const uint8_t *q1_valid,
                          const float *q1_u, const float *q1_v, const int32_t *q1_level,
                          const uint8_t *q1_desc, const uint8_t *q2_valid, const float *q2_u,
                          const float *q2_v, const int32_t *q2_level, const uint8_t *q2_desc, float th, const float *scale_factors1,
                          const float *scale_factors2, int32_t *match12);

/* DBoW3::Vocabulary::transform as used by computeBow (frame.cpp:248-253, keyframe.cpp:394-398) */
void orc_bow_transform(int depth_L, const int32_t *child_start, const int32_t *children, const uint8_t *node_desc,
                       const double *node_weight, const int32_t *word_id, int n, const uint8_t *desc, int levelsup,
                       int32_t *out_word, double *out_weight, int32_t *out_node);
double orc_bow_score(int n1, const int32_t *w1, const double *v1, int n2, const int32_t *w2, const double *v2);

/* ---------------- Optimizer (reference optimizer_ceres.{h,cpp}) --------------- */

void orc_se3_exp(const double xi[6], double q[4] /*w,x,y,z*/, double t[3]); /* Sophus SE3::exp */
void orc_se3_log(const double q[4], const double t[3], double xi[6]);       /* Sophus SE3::log */
void orc_se3_plus(const double x[6], const double delta[6], double out[6]); /* :44-53 */
void orc_se3_trans_point(const double se3[6], const double pt[3], double out[3]); /* .h:29-95 */
void orc_se3_apply(const double q[4], const double t[3], const double p[3], double out[3]);
void orc_angle_axis_to_R(const double aa[3], double R[9] /*column-major*/);

/* residual + Jacobians of one edge; uR<0 => mono (2 rows) else stereo (3 rows).  Returns rows.
 * Jp row-major rows x 6, Jl row-major rows x 3 (either may be NULL). cam = fx,fy,cx,cy,bf. */
int orc_edge_eval(const double pose[6], const double pt[3], const double obs[3], double inv_sigma,
                  const double cam[5], double r[3], double *Jp, double *Jl);

typedef struct {
  int max_iterations;
  int iterations;          /* iterations actually run (iteration 0 excluded) */
  int accepted;
  double initial_cost, final_cost;
  double final_radius;
  int termination;         /* 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 failure */
  /* optional per-iteration trace (caller supplies arrays of max_iterations+1 or NULL) */
  double *trace_cost, *trace_radius;
  int *trace_accepted;
} orc_lm_summary;

/* Optimizer::solvePoseOnlySE3 (:157-314) on flat arrays.  pose in/out = se3 log [upsilon;omega].
 * outlier[n] out.  Returns inlier count.  sums: optional [2] LM summaries (round 0, round 1). */
int orc_pose_only_solve(int n, const double *pts /*n x 3*/, const double *obs /*n x 3*/,
                        const double *inv_sigma, const double cam[5], double pose[6],
                        uint8_t *outlier, orc_lm_summary *sums);

/* one Ceres-style LM solve of a BA problem with point-block Schur elimination.
 * huber_mono/huber_stereo <= 0 => no loss.  edge_active: NULL or mask.  Edges may come in any
 * order; the oracle processes them in the given order. */
int orc_ba_lm(int n_cams, double *poses /*n_cams x 6 in/out*/, const uint8_t *cam_fixed,
              int n_pts, double *points /*n_pts x 3 in/out*/, int n_edges, const int32_t *e_cam,
              const int32_t *e_pt, const double *e_obs /*n_edges x 3*/, const double *e_inv_sigma,
              const uint8_t *edge_active, const double cam[5], double huber_mono,
              double huber_stereo, int max_iterations, orc_lm_summary *sum);

/* Optimizer::solveLocalBAPoseAndPoint (:446-808) numerics on flat arrays: problem 1 (Huber, 5 it),
 * float chi2 classification, problem 2 (no loss, 10 it), final chi2 pass.  edge_erase[n_edges] out.
 * stop: polled exactly where the reference polls stopFlag (:594, :612).  Returns 0, or 1 if the
 * first poll aborted (no write-back, Q-B8). */
int orc_local_ba(int n_cams, double *poses, const uint8_t *cam_fixed, int n_pts, double *points,
                 int n_edges, const int32_t *e_cam, const int32_t *e_pt, const double *e_obs,
                 const double *e_inv_sigma, const double cam[5], const volatile int *stop,
                 uint8_t *edge_erase, orc_lm_summary *sums /*[2] or NULL*/);

/* Linearisation products at the current point, for K8/K9 checks: Schur S (6nf x 6nf, row-major,
 * nf = number of free cams in index order), rhs b (6nf), cost.  No LM damping, no Jacobi scaling,
 * loss applied if huber>0. */
int orc_ba_schur(int n_cams, const double *poses, const uint8_t *cam_fixed, int n_pts,
                 const double *points, int n_edges, const int32_t *e_cam, const int32_t *e_pt,
                 const double *e_obs, const double *e_inv_sigma, const uint8_t *edge_active,
                 const double cam[5], double huber_mono, double huber_stereo, double point_damping,
                 double *S, double *b, double *cost);

/* PoseOnlySim3 / PoseOnlyInverseSim3 residual blocks of one match at x = [angle-axis, t, s]
 * (optimizer_ceres.h:211-267); Jacobians 2 x 7 row-major or NULL. */
void orc_sim3_eval(const double x[7], const double cam_match[3], const double pix_curr[2], double isig_c,
                   const double cam_curr[3], const double pix_match[2], double isig_m,
                   const double cam[4], double r_fwd[2], double J_fwd[14], double r_inv[2], double J_inv[14]);

/* Optimizer::solveLoopSim3 (optimizer_ceres.cpp:810-1030) on flat arrays.  pose in/out =
 * [angle-axis; t] of Scm, *scale its scale; outlier[n]: inlierMappoints entry nulled.  Returns the
 * inlier count (0 with pose/scale untouched when fewer than 10 matches survive problem 1). */
int orc_sim3_solve(int n, const double *cam_match, const double *pix_curr, const double *isig_curr,
                   const double *cam_curr, const double *pix_match, const double *isig_match,
                   const double cam[4], int fix_scale, double pose[6], double *scale, uint8_t *outlier,
                   orc_lm_summary *sums /*[2] or NULL*/);

/* ceres::EigenQuaternionParameterization::Plus (quaternion order x, y, z, w) */
void orc_quat_plus(const double q[4], const double d[3], double o[4]);
/* PoseGraphLoop residual (optimizer_ceres.h:269-325) of one edge and its tangent Jacobians
 * (7 x 6 row-major per node: [rotation delta, translation]) or NULL */
void orc_pose_graph_edge(const double q1[4], const double t1[3], double s1, const double q2[4], const double t2[3],
                         double s2, const double qm[4], const double tm[3], double sm, double r[7], double *J1,
                         double *J2);
/* Optimizer::solvePoseGraphLoop's solve (optimizer_ceres.cpp:1238-1258) on flat arrays: scales
 * constant (fixScaleFlag), node `fixed` constant, no loss, LM with exact normal-equation solves. */
int orc_pose_graph_solve(int n_nodes, double *quats /*4n: x y z w*/, double *trans /*3n*/, const double *scales,
                         int fixed, int n_edges, const int32_t *e_i, const int32_t *e_j, const double *q_meas,
                         const double *t_meas, const double *s_meas, int max_iterations, orc_lm_summary *sum);

#ifdef __cplusplus
}
#endif
#endif
