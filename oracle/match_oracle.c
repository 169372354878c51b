/*
 * match_oracle.c -- CPU restatement of myslam::Matcher pieces on flat arrays
 * (reference src/matcher.cpp, src/frame.cpp).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Parity unpinned (the reference has no tests); pure integer / float code restated 1:1.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100     /* matcher.cpp:11 */
#define TH_LOW 50       /* :12 */
#define HISTO_LENGTH 30 /* :13 */

/* Matcher::computeDistance, matcher.cpp:1240-1256 (SWAR popcount on 8 x int32) */
int orc_hamming256(const uint8_t *a, const uint8_t *b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    memcpy(&x, a + 4 * i, 4);
    memcpy(&y, b + 4 * i, 4);
    uint32_t v = x ^ y;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

void orc_hamming_matrix(const uint8_t *A, int na, const uint8_t *B, int nb, uint16_t *D) {
  for (int i = 0; i < na; i++)
    for (int j = 0; j < nb; j++)
      D[(size_t)i * nb + j] = (uint16_t)orc_hamming256(A + (size_t)i * 32, B + (size_t)j * 32);
}

/* Matcher::computeThreeMax, matcher.cpp:1258-1304 */
void orc_three_max(const int *hist_sizes, int L, int *ind1, int *ind2, int *ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = hist_sizes[i];
    if (s > max1) {
      max3 = max2;
      *ind3 = *ind2;
      max2 = max1;
      *ind2 = *ind1;
      max1 = s;
      *ind1 = i;
    } else if (s > max2) {
      max3 = max2;
      *ind3 = *ind2;
      max2 = s;
      *ind2 = i;
    } else if (s > max3) {
      max3 = s;
      *ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    *ind2 = -1;
    *ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    *ind3 = -1;
  }
}

/* Frame::assignFeaturesToGrid, frame.cpp:72-89: cell lists keep insertion (feature index) order */
void orc_frame_build_grid(orc_frame *f) {
  const int ncell = ORC_GRID_COLS * ORC_GRID_ROWS;
  int *cell_of = (int *)malloc(sizeof(int) * (f->n > 0 ? f->n : 1));
  memset(f->cell_start, 0, sizeof(int) * (ncell + 1));
  for (int i = 0; i < f->n; i++) {
    const int gx = (int)roundf((f->x[i] - f->xmin) * f->grid_per_px_w);
    const int gy = (int)roundf((f->y[i] - f->ymin) * f->grid_per_px_h);
    if (gx < 0 || gx >= ORC_GRID_COLS || gy < 0 || gy >= ORC_GRID_ROWS) {
      cell_of[i] = -1;
      continue;
    }
    cell_of[i] = gx * ORC_GRID_ROWS + gy;
    f->cell_start[cell_of[i] + 1]++;
  }
  for (int c = 0; c < ncell; c++) f->cell_start[c + 1] += f->cell_start[c];
  int *fill = (int *)calloc(ncell, sizeof(int));
  for (int i = 0; i < f->n; i++) {
    if (cell_of[i] < 0) continue;
    f->cell_items[f->cell_start[cell_of[i]] + fill[cell_of[i]]++] = i;
  }
  free(fill);
  free(cell_of);
}

/* Frame::getFeaturesInArea, frame.cpp:199-247 */
int orc_features_in_area(const orc_frame *f, float u, float v, float radius, int min_level,
                         int max_level, int *out, int cap) {
  int n = 0;
  int minGX = (int)floorf((u - f->xmin - radius) * f->grid_per_px_w);
  if (minGX < 0) minGX = 0;
  if (minGX >= ORC_GRID_COLS) return 0;
  int maxGX = (int)floorf((u - f->xmin + radius) * f->grid_per_px_w);
  if (maxGX > ORC_GRID_COLS - 1) maxGX = ORC_GRID_COLS - 1;
  if (maxGX < 0) return 0;
  int minGY = (int)floorf((v - f->ymin - radius) * f->grid_per_px_h);
  if (minGY < 0) minGY = 0;
  if (minGY >= ORC_GRID_ROWS) return 0;
  int maxGY = (int)floorf((v - f->ymin + radius) * f->grid_per_px_h);
  if (maxGY > ORC_GRID_ROWS - 1) maxGY = ORC_GRID_ROWS - 1;
  if (maxGY < 0) return 0;
  for (int ix = minGX; ix <= maxGX; ix++) {
    for (int iy = minGY; iy <= maxGY; iy++) {
      int c = ix * ORC_GRID_ROWS + iy;
      for (int t = f->cell_start[c]; t < f->cell_start[c + 1]; t++) {
        int k = f->cell_items[t];
        if (f->octave[k] < min_level || f->octave[k] > max_level) continue;
        const float distx = f->x[k] - u;
        const float disty = f->y[k] - v;
        if (fabsf(distx) < radius && fabsf(disty) < radius) {
          if (n < cap) out[n] = k;
          n++;
        }
      }
    }
  }
  return n;
}

/* Matcher::searchByProjection(Frame*,Frame*,radius,checkRot), matcher.cpp:18-148.
 * q_observed[i] = (mp->observe_cnt_ > 0) of query i; blocked[idx] = the feature already held an
 * observed point before the call.  assigned[idx] (in/out) = query index now held, -1 = none.
 * direction: 1 forward (:70-71), 2 backward (:72-73), 0 otherwise (:74-75). */
int orc_match_frame_projection(const orc_frame *cur, int nq, const uint8_t *q_valid,
                               const float *q_u, const float *q_v, const float *q_invz,
                               const int32_t *q_octave, const float *q_angle,
                               const uint8_t *q_desc, float radius, float bf, int direction,
                               int check_rot, int n_levels, const float *scale_factors,
                               const uint8_t *blocked_in, int32_t *assigned) {
  const float pdf = HISTO_LENGTH / 360.0f; /* :14 */
  int match_cnt = 0;
  int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (size_t)(nq > 0 ? nq : 1));
  int hist_n[HISTO_LENGTH];
  memset(hist_n, 0, sizeof(hist_n));
  int *cand = (int *)malloc(sizeof(int) * (cur->n > 0 ? cur->n : 1));
  uint8_t *blocked = (uint8_t *)malloc(cur->n > 0 ? cur->n : 1);
  /* q_observed travels in bit 1 of q_valid (bit 0 = valid) to keep the argument list flat */
  for (int k = 0; k < cur->n; k++) blocked[k] = blocked_in ? blocked_in[k] : 0;
  for (int i = 0; i < nq; i++) {
    if (!(q_valid[i] & 1)) continue;
    const float u = q_u[i], v = q_v[i], invz = q_invz[i];
    const int lastOctave = q_octave[i];
    const float radius_scale = radius * scale_factors[lastOctave];
    int nc;
    if (direction == 1)
      nc = orc_features_in_area(cur, u, v, radius_scale, lastOctave, n_levels, cand, cur->n);
    else if (direction == 2)
      nc = orc_features_in_area(cur, u, v, radius_scale, 0, lastOctave, cand, cur->n);
    else
      nc = orc_features_in_area(cur, u, v, radius_scale, lastOctave - 1, lastOctave + 1, cand, cur->n);
    if (nc == 0) continue;
    int bestDist = 256, bestIdx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (blocked[idx]) continue; /* :87 */
      if (cur->uright[idx] > 0) { /* :90-96 */
        const float u_r = u - bf * invz;
        const float error = fabsf(u_r - cur->uright[idx]);
        if (error > radius_scale) continue;
      }
      const int dist = orc_hamming256(q_desc + (size_t)i * 32, cur->desc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = idx;
      }
    }
    if (bestDist <= TH_HIGH) {
      assigned[bestIdx] = i;
      blocked[bestIdx] = (q_valid[i] >> 1) & 1;
      match_cnt++;
      if (check_rot) {
        float rot = q_angle[i] - cur->angle[bestIdx];
        if (rot < 0) rot += 360.0f;
        int bin = orc_cv_round_f(rot * pdf);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin * nq + hist_n[bin]++] = bestIdx;
      }
    }
  }
  if (check_rot) {
    int i1 = -1, i2 = -1, i3 = -1;
    orc_three_max(hist_n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int b = 0; b < HISTO_LENGTH; b++) {
      if (b != i1 && b != i2 && b != i3)
        for (int j = 0; j < hist_n[b]; j++) {
          assigned[hist[b * nq + j]] = -1;
          match_cnt--;
        }
    }
  }
  free(hist);
  free(cand);
  free(blocked);
  return match_cnt;
}

/* Matcher::searchByProjection(Frame*, const vector<MapPoint*>&, thRadius), matcher.cpp:274-353 */
int orc_match_local_map(const orc_frame *cur, int nq, const uint8_t *q_valid, const float *q_u,
                        const float *q_v, const float *q_ur, const int32_t *q_level,
                        const float *q_viewcos, const uint8_t *q_desc, float th_radius,
                        float ratio, const float *scale_factors, const uint8_t *blocked_in,
                        int32_t *assigned) {
  int match_cnt = 0;
  int *cand = (int *)malloc(sizeof(int) * (cur->n > 0 ? cur->n : 1));
  uint8_t *blocked = (uint8_t *)malloc(cur->n > 0 ? cur->n : 1);
  for (int k = 0; k < cur->n; k++) blocked[k] = blocked_in ? blocked_in[k] : 0;
  for (int im = 0; im < nq; im++) {
    if (!(q_valid[im] & 1)) continue;
    float radius;
    if (q_viewcos[im] > 0.998) radius = 2.5;
    else radius = 4.0;
    radius *= th_radius;
    const int level_predict = q_level[im];
    const float radius_scale = radius * scale_factors[level_predict];
    const int nc = orc_features_in_area(cur, q_u[im], q_v[im], radius_scale, level_predict - 1,
                                        level_predict, cand, cur->n);
    if (nc == 0) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (blocked[idx]) continue; /* :314 */
      if (cur->uright[idx] > 0) {
        const float er = fabsf(q_ur[im] - cur->uright[idx]);
        if (er > radius_scale) continue;
      }
      const int dist = orc_hamming256(q_desc + (size_t)im * 32, cur->desc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = cur->octave[idx];
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = cur->octave[idx];
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > ratio * (float)bestDist2) continue;
      assigned[bestIdx] = im;
      blocked[bestIdx] = (q_valid[im] >> 1) & 1;
      match_cnt++;
    }
  }
  free(cand);
  free(blocked);
  return match_cnt;
}

/* ---------------------------------------------------------------------------------------------
 * Matcher::searchByProjection(Frame*, KeyFrame*, radius, distThreshold, found, checkRot),
 * matcher.cpp:150-272 (relocalisation top-up).  Query i = key-frame map point i already projected
 * (valid: exists, not bad, not in `found`, z>0, inside the image, inside its distance range);
 * q_level = predictScale.  A feature that holds ANY map point is skipped (:218), including the ones
 * assigned earlier in this call.
 * ------------------------------------------------------------------------------------------- */
int orc_match_frame_keyframe(const orc_frame *cur, int nq, const uint8_t *q_valid, const float *q_u,
                             const float *q_v, const int32_t *q_level, const float *q_angle,
                             const uint8_t *q_desc, float radius, float dist_threshold, int check_rot,
                             const float *scale_factors, const uint8_t *has_mp_in, int32_t *assigned) {
  const float pdf = HISTO_LENGTH / 360.0f;
  int match_cnt = 0;
  int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (size_t)(nq > 0 ? nq : 1));
  int hist_n[HISTO_LENGTH];
  memset(hist_n, 0, sizeof(hist_n));
  int *cand = (int *)malloc(sizeof(int) * (cur->n > 0 ? cur->n : 1));
  uint8_t *has = (uint8_t *)malloc(cur->n > 0 ? cur->n : 1);
  for (int k = 0; k < cur->n; k++) has[k] = has_mp_in ? has_mp_in[k] : 0;
  for (int i = 0; i < nq; i++) {
    if (!(q_valid[i] & 1)) continue;
    const int lp = q_level[i];
    const float rs = radius * scale_factors[lp];
    const int nc = orc_features_in_area(cur, q_u[i], q_v[i], rs, lp - 1, lp + 1, cand, cur->n);
    if (nc == 0) continue;
    int bestDist = 256, bestIdx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (has[idx]) continue;
      const int dist = orc_hamming256(q_desc + (size_t)i * 32, cur->desc + (size_t)idx * 32);
      if (dist < bestDist) bestDist = dist, bestIdx = idx;
    }
    if ((float)bestDist <= dist_threshold) { /* int <= float comparison of the reference */
      assigned[bestIdx] = i;
      has[bestIdx] = 1;
      match_cnt++;
      if (check_rot) {
        float rot = q_angle[i] - cur->angle[bestIdx];
        if (rot < 0) rot += 360.0f;
        int bin = orc_cv_round_f(rot * pdf);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin * nq + hist_n[bin]++] = bestIdx;
      }
    }
  }
  if (check_rot) {
    int i1 = -1, i2 = -1, i3 = -1;
    orc_three_max(hist_n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int b = 0; b < HISTO_LENGTH; b++)
      if (b != i1 && b != i2 && b != i3)
        for (int j = 0; j < hist_n[b]; j++) {
          assigned[hist[b * nq + j]] = -1;
          match_cnt--;
        }
  }
  free(hist), free(cand), free(has);
  return match_cnt;
}

/* DBoW3::FeatureVector as CSR: node ids ascending, node k owns feat[start[k] .. start[k+1]) */
static int next_common_node(const orc_bow *a, const orc_bow *b, int *ia, int *ib) {
  while (*ia < a->n_nodes && *ib < b->n_nodes) {
    if (a->node_id[*ia] == b->node_id[*ib]) return 1;
    if (a->node_id[*ia] < b->node_id[*ib]) (*ia)++; /* lower_bound walk, :541-544 */
    else (*ib)++;
  }
  return 0;
}

/* Matcher::searchByBoW(KeyFrame*, Frame*, matches, checkRot) matcher.cpp:449-559 (mode 0) and
 * searchByBoW(KeyFrame*, KeyFrame*, ...) :561-677 (mode 1).
 * mode 0: match[b->n] = A index held by B feature (or -1); a_valid = A feature has a good map point.
 * mode 1: match[a->n] = B index matched to A feature; b_valid likewise; uses round() for the bin. */
int orc_match_bow(const orc_frame *a, const uint8_t *a_valid, const orc_bow *an, const orc_frame *b,
                  const uint8_t *b_valid, const orc_bow *bn, int mode, float ratio, int check_rot,
                  int32_t *match) {
  const float pdf = HISTO_LENGTH / 360.0f;
  const int nout = mode == 0 ? b->n : a->n;
  for (int i = 0; i < nout; i++) match[i] = -1;
  uint8_t *matched2 = (uint8_t *)calloc(b->n > 0 ? b->n : 1, 1);
  int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (size_t)(a->n > 0 ? a->n : 1));
  int hist_n[HISTO_LENGTH];
  memset(hist_n, 0, sizeof(hist_n));
  int cnt = 0, ia = 0, ib = 0;
  while (next_common_node(an, bn, &ia, &ib)) {
    for (int s = an->start[ia]; s < an->start[ia + 1]; s++) {
      const int i1 = (int)an->feat[s];
      if (!a_valid[i1]) continue;
      int best1 = 256, best2 = 256, bidx = -1;
      for (int t = bn->start[ib]; t < bn->start[ib + 1]; t++) {
        const int i2 = (int)bn->feat[t];
        if (mode == 0) {
          if (match[i2] >= 0) continue; /* :487 */
        } else {
          if (matched2[i2] || !b_valid[i2]) continue; /* :605-608 */
        }
        const int d = orc_hamming256(a->desc + (size_t)i1 * 32, b->desc + (size_t)i2 * 32);
        if (d < best1) best2 = best1, best1 = d, bidx = i2;
        else if (d < best2) best2 = d;
      }
      if (best1 <= TH_LOW && (float)best1 < ratio * (float)best2) {
        float rot = a->angle[i1] - b->angle[bidx];
        if (rot < 0) rot += 360.0f;
        int bin = mode == 0 ? orc_cv_round_f(rot * pdf) : (int)roundf(rot * pdf); /* :518 vs :637 */
        if (bin == HISTO_LENGTH) bin = 0;
        if (mode == 0) {
          match[bidx] = i1;
          if (check_rot) hist[bin * a->n + hist_n[bin]++] = bidx;
        } else {
          match[i1] = bidx;
          matched2[bidx] = 1;
          if (check_rot) hist[bin * a->n + hist_n[bin]++] = i1;
        }
        cnt++;
      }
    }
    ia++, ib++;
  }
  if (check_rot) {
    int i1 = -1, i2 = -1, i3 = -1;
    orc_three_max(hist_n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int bb = 0; bb < HISTO_LENGTH; bb++)
      if (bb != i1 && bb != i2 && bb != i3)
        for (int j = 0; j < hist_n[bb]; j++) {
          match[hist[bb * a->n + j]] = -1;
          cnt--;
        }
  }
  free(matched2), free(hist);
  return cnt;
}

/* Matcher::checkEpipolarConstrain, matcher.cpp:1306-1324 (float arithmetic on a double line) */
static int epipolar_ok(float x1, float y1, float x2, float y2, const double F[9], float sigma) {
  const double l0 = x1 * F[0] + y1 * F[3] + F[6], l1 = x1 * F[1] + y1 * F[4] + F[7],
               l2 = x1 * F[2] + y1 * F[5] + F[8]; /* (p1^T F12)^T */
  const float num = (float)(l0 * x2 + l1 * y2 + l2);
  const float den = (float)(l0 * l0 + l1 * l1);
  if (den == 0) return 0;
  const float d2 = num * num / den;
  return d2 < 3.84f * sigma * sigma;
}

/* Matcher::searchForTriangulation(kf1, kf2, matchIdxs, F12, checkRot) matcher.cpp:867-1010.
 * a_has_mp / b_has_mp: the feature already has a map point (then it is skipped).  F12 row-major.
 * match12[a->n] = B index or -1.  (ex, ey) = projection of camera centre 1 into key-frame 2. */
int orc_match_triangulation(const orc_frame *a, const uint8_t *a_has_mp, const orc_bow *an,
                            const orc_frame *b, const uint8_t *b_has_mp, const orc_bow *bn,
                            const double F12[9], float ex, float ey, const float *scale_factors,
                            int check_rot, int32_t *match12) {
  const float pdf = HISTO_LENGTH / 360.0f;
  for (int i = 0; i < a->n; i++) match12[i] = -1;
  uint8_t *matched2 = (uint8_t *)calloc(b->n > 0 ? b->n : 1, 1);
  int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (size_t)(a->n > 0 ? a->n : 1));
  int hist_n[HISTO_LENGTH];
  memset(hist_n, 0, sizeof(hist_n));
  int cnt = 0, ia = 0, ib = 0;
  while (next_common_node(an, bn, &ia, &ib)) {
    for (int s = an->start[ia]; s < an->start[ia + 1]; s++) {
      const int i1 = (int)an->feat[s];
      if (a_has_mp[i1]) continue;
      const int stereo1 = a->uright[i1] >= 0;
      int bestDist = TH_LOW, bidx = -1;
      for (int t = bn->start[ib]; t < bn->start[ib + 1]; t++) {
        const int i2 = (int)bn->feat[t];
        if (matched2[i2] || b_has_mp[i2]) continue;
        const int stereo2 = b->uright[i2] >= 0;
        const int d = orc_hamming256(a->desc + (size_t)i1 * 32, b->desc + (size_t)i2 * 32);
        if (d > TH_LOW || d > bestDist) continue; /* later equal distances replace earlier ones */
        if (!stereo1 && !stereo2) {
          const float dx = ex - b->x[i2], dy = ey - b->y[i2];
          if (dx * dx + dy * dy < 100 * scale_factors[b->octave[i2]]) continue;
        }
        if (epipolar_ok(a->x[i1], a->y[i1], b->x[i2], b->y[i2], F12, scale_factors[b->octave[i2]])) {
          bestDist = d;
          bidx = i2;
        }
      }
      if (bidx >= 0) {
        match12[i1] = bidx;
        matched2[bidx] = 1;
        if (check_rot) {
          float rot = a->angle[i1] - b->angle[bidx];
          if (rot < 0) rot += 360.0f;
          int bin = (int)roundf(rot * pdf);
          if (bin == HISTO_LENGTH) bin = 0;
          hist[bin * a->n + hist_n[bin]++] = i1;
        }
        cnt++;
      }
    }
    ia++, ib++;
  }
  if (check_rot) {
    int i1 = -1, i2 = -1, i3 = -1;
    orc_three_max(hist_n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int bb = 0; bb < HISTO_LENGTH; bb++)
      if (bb != i1 && bb != i2 && bb != i3)
        for (int j = 0; j < hist_n[bb]; j++) {
          match12[hist[bb * a->n + j]] = -1;
          cnt--;
        }
  }
  free(matched2), free(hist);
  return cnt;
}

/* matching part of Matcher::fuseMapPoints (matcher.cpp:1012-1133; the map mutation at :1108-1127
 * stays in the caller).  Query = candidate map point already projected into the key-frame (valid:
 * good, not observed by it, z>=0, inside the image and its distance / viewing-angle range).
 * best_idx[nq] = feature index with the smallest Hamming distance <= TH_LOW after the level and
 * chi2 gates, or -1.  Queries are independent of each other. */
int orc_match_fuse(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                   const float *q_v, const float *q_ur, const int32_t *q_level, const uint8_t *q_desc,
                   float threshold, const float *scale_factors, int32_t *best_idx) {
  int cnt = 0;
  int *cand = (int *)malloc(sizeof(int) * (kf->n > 0 ? kf->n : 1));
  for (int i = 0; i < nq; i++) {
    best_idx[i] = -1;
    if (!(q_valid[i] & 1)) continue;
    const int lp = q_level[i];
    const float radius = threshold * scale_factors[lp];
    /* KeyFrame::getFeaturesInArea(u,v,r) has no level filter (keyframe.cpp:268-312) */
    const int nc = orc_features_in_area(kf, q_u[i], q_v[i], radius, -1000, 1000, cand, kf->n);
    int bestDist = 256, bidx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (kf->octave[idx] < lp - 1 || kf->octave[idx] > lp) continue;
      const float exx = q_u[i] - kf->x[idx], eyy = q_v[i] - kf->y[idx];
      const float invSigma = 1.0f / scale_factors[kf->octave[idx]];
      if (kf->uright[idx] >= 0) {
        const float er = q_ur[i] - kf->uright[idx];
        const float e2 = exx * exx + eyy * eyy + er * er;
        if (e2 * invSigma * invSigma > 7.815f) continue;
      } else {
        const float e2 = exx * exx + eyy * eyy;
        if (e2 * invSigma * invSigma > 5.991f) continue;
      }
      const int d = orc_hamming256(q_desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
      if (d < bestDist) bestDist = d, bidx = idx;
    }
    if (bestDist <= TH_LOW) {
      best_idx[i] = bidx;
      cnt++;
    }
  }
  free(cand);
  return cnt;
}

/* ---- Sim3 / loop-closure searches (M4, M7, M10) ------------------------------------------------
 * All three project map points into a key-frame and take, per point, the best Hamming match among
 * the features of KeyFrame::getFeaturesInArea(u, v, th * scale[level_predict]) whose octave is in
 * [level_predict - 1, level_predict].  The caller supplies the projections (flag bit 0 = the gates
 * of matcher.cpp:380-406 / 728-754 / 1163-1187 passed). */

/* best feature per query, queries independent: the inner search of searchBySim3 (matcher.cpp:756-786,
 * 821-851; max_dist = TH_HIGH) and of fuseByPose (:1196-1213; max_dist = TH_LOW) */
int orc_match_area_best(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                        const float *q_v, const int32_t *q_level, const uint8_t *q_desc, float th,
                        const float *scale_factors, int max_dist, int32_t *best_idx) {
  int cnt = 0;
  int *cand = (int *)malloc(sizeof(int) * (kf->n > 0 ? kf->n : 1));
  for (int i = 0; i < nq; i++) {
    best_idx[i] = -1;
    if (!(q_valid[i] & 1)) continue;
    const int lp = q_level[i];
    const float radius = th * scale_factors[lp];
    const int nc = orc_features_in_area(kf, q_u[i], q_v[i], radius, -1000, 1000, cand, kf->n);
    int bestDist = 256, bidx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (kf->octave[idx] < lp - 1 || kf->octave[idx] > lp) continue;
      const int d = orc_hamming256(q_desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
      if (d < bestDist) bestDist = d, bidx = idx;
    }
    if (bestDist <= max_dist) {
      best_idx[i] = bidx;
      cnt++;
    }
  }
  free(cand);
  return cnt;
}

/* Matcher::searchByProjection(KeyFrame*, Sim3&, loopMapPoints, matchMapPoints, th), matcher.cpp:356-447.
 * occupied[k] != 0 <=> matchMapPoints[k] is non-null on entry; assigned[k] = query that claimed
 * feature k during the call.  Q-M1 (:422): the skip test indexes matchMapPoints with the
 * CANDIDATE COUNTER j, not with the feature index -- reproduced literally. */
int orc_match_sim3_projection(const orc_frame *kf, int nq, const uint8_t *q_valid, const float *q_u,
                              const float *q_v, const int32_t *q_level, const uint8_t *q_desc, int th,
                              const float *scale_factors, const uint8_t *occupied, int32_t *assigned) {
  int cnt = 0;
  int *cand = (int *)malloc(sizeof(int) * (kf->n > 0 ? kf->n : 1));
  uint8_t *occ = (uint8_t *)malloc(kf->n > 0 ? kf->n : 1);
  for (int k = 0; k < kf->n; k++) occ[k] = occupied ? occupied[k] : 0, assigned[k] = -1;
  for (int i = 0; i < nq; i++) {
    if (!(q_valid[i] & 1)) continue;
    const int lp = q_level[i];
    const float radius = (float)th * scale_factors[lp];
    const int nc = orc_features_in_area(kf, q_u[i], q_v[i], radius, -1000, 1000, cand, kf->n);
    int bestDist = 256, bidx = -1;
    for (int j = 0; j < nc; j++) {
      const int idx = cand[j];
      if (occ[j]) continue; /* Q-M1 */
      if (kf->octave[idx] < lp - 1 || kf->octave[idx] > lp) continue;
      const int d = orc_hamming256(q_desc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
      if (d < bestDist) bestDist = d, bidx = idx;
    }
    if (bestDist <= TH_LOW) {
      occ[bidx] = 1; /* matchMapPoints[bestIdx] = mp (:439); a second claim overwrites the first */
      assigned[bidx] = i;
      cnt++;
    }
  }
  free(cand);
  free(occ);
  return cnt;
}

/* Matcher::searchBySim3, matcher.cpp:679-865: q1 = map points of key-frame 1 projected into key-frame
 * 2 (:722-754), q2 the reverse (:790-819); a pair survives when both directions agree (:853-864).
 * match12[i] = feature of key-frame 2 matched to feature i of key-frame 1, or -1. */
int orc_match_sim3_mutual(const orc_frame *kf1, const orc_frame *kf2, const uint8_t *q1_valid,
                          const float *q1_u, const float *q1_v, const int32_t *q1_level,
                          const uint8_t *q1_desc, const uint8_t *q2_valid, const float *q2_u,
                          const float *q2_v, const int32_t *q2_level, const uint8_t *q2_desc, float th, const float *scale_factors1,
                          const float *scale_factors2, int32_t *match12) {
  int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (kf1->n > 0 ? kf1->n : 1));
  int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * (kf2->n > 0 ? kf2->n : 1));
  orc_match_area_best(kf2, kf1->n, q1_valid, q1_u, q1_v, q1_level, q1_desc, th, scale_factors2, TH_HIGH, m1);
  orc_match_area_best(kf1, kf2->n, q2_valid, q2_u, q2_v, q2_level, q2_desc, th, scale_factors1, TH_HIGH, m2);
  int found = 0;
  for (int i = 0; i < kf1->n; i++) {
    match12[i] = -1;
    const int idx2 = m1[i];
    if (idx2 >= 0 && m2[idx2] == i) match12[i] = idx2, found++;
  }
  free(m1);
  free(m2);
  return found;
}

/* ---- N2: DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup) as called by
 * Frame::computeBow / KeyFrame::computeBow (frame.cpp:248-253, keyframe.cpp:394-398; levelsup = 3).
 * DBoW3 0.0.1 is an external, unpinned dependency: its published algorithm is restated -- from the root,
 * at every level take the child with the smallest Hamming distance (first wins ties), remember the node
 * reached at level L - levelsup, stop at a leaf; the leaf's word id and weight go to the BoW vector, the
 * remembered node to the feature vector.  Tree as flat arrays: children of node i are
 * children[child_start[i] .. child_start[i+1]); word_id[i] >= 0 marks a leaf; node 0 is the root. */
void orc_bow_transform(int depth_L, const int32_t *child_start, const int32_t *children, const uint8_t *node_desc,
                       const double *node_weight, const int32_t *word_id, int n, const uint8_t *desc, int levelsup,
                       int32_t *out_word, double *out_weight, int32_t *out_node) {
  const int nid_level = depth_L - levelsup;
  for (int i = 0; i < n; i++) {
    const uint8_t *f = desc + (size_t)i * 32;
    int final_id = 0, level = 0, nid = 0; /* nid_level <= 0: the root */
    while (word_id[final_id] < 0 && child_start[final_id + 1] > child_start[final_id]) {
      level++;
      const int c0 = child_start[final_id], c1 = child_start[final_id + 1];
      int best = children[c0], bestd = orc_hamming256(f, node_desc + (size_t)best * 32);
      for (int c = c0 + 1; c < c1; c++) {
        const int id = children[c];
        const int d = orc_hamming256(f, node_desc + (size_t)id * 32);
        if (d < bestd) bestd = d, best = id;
      }
      final_id = best;
      if (level == nid_level) nid = final_id;
    }
    out_word[i] = word_id[final_id];
    out_weight[i] = node_weight[final_id];
    out_node[i] = nid;
  }
}

/* Map::score (map.cpp:335-376): L1 score of two BoW vectors given as ascending (word, value) lists */
double orc_bow_score(int n1, const int32_t *w1, const double *v1, int n2, const int32_t *w2, const double *v2) {
  double score = 0;
  int i = 0, j = 0;
  while (i < n1 && j < n2) {
    if (w1[i] == w2[j]) {
      score += fabs(v1[i] - v2[j]) - fabs(v1[i]) - fabs(v2[j]);
      i++, j++;
    } else if (w1[i] < w2[j]) {
      i++;
    } else {
      j++;
    }
  }
  return -score / 2.0;
}
