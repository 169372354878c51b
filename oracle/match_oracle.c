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
