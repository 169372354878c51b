/*
 * orb_oracle.c -- CPU restatement of ORB_SLAM2::ORBextractor (reference
 * src/ORBextractor.cpp) in plain C.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED at the OpenCV boundary: the reference has no golden vectors
 * and OpenCV is not installed here.  The OpenCV 3.x algorithms restated below
 * (and the variant picked where versions differ) are the contract:
 *   cv::FAST        TYPE_9_16 scalar path (segment test + cornerScore<16> + 3x3 NMS)
 *   cv::resize      INTER_LINEAR, CV_8U: 11-bit fixed-point coefficients
 *   cv::GaussianBlur 7x7 sigma 2, CV_8U: <=3.4.0 separable filter with the kernel
 *                   quantised to 8 fractional bits per pass, (v + 2^15) >> 16
 *   cv::fastAtan2   3.x scalar polynomial (degrees)
 *   cvRound         round-half-to-even (lrintf / cvtss2si)
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31      /* ORBextractor.cpp:74 */
#define HALF_PATCH_SIZE 15 /* :75 */
#define EDGE_THRESHOLD 19  /* :76 */

int orc_cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
static int cv_ceil_f(float v) {
  int i = (int)v;
  return i + (i < v);
}

/* ------------------------------------------------------------------ E0 ---- */
/* ORBextractor::ORBextractor, ORBextractor.cpp:414-476 */
void orc_orb_params_init(orc_orb_params *p, int nfeatures, float scale_factor, int nlevels,
                         int ini_th, int min_th, const int8_t *pattern1024) {
  memset(p, 0, sizeof(*p));
  p->nfeatures = nfeatures;
  p->nlevels = nlevels;
  p->ini_th = ini_th;
  p->min_th = min_th;
  p->scale_factor = scale_factor;
  /* the member `scaleFactor` is a double initialised from the float argument (:83 of the header) */
  double scaleFactor = (double)scale_factor;
  p->scale[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) p->scale[i] = (float)(p->scale[i - 1] * scaleFactor); /* :423-427 */
  for (int i = 0; i < nlevels; i++) p->inv_scale[i] = 1.0f / p->scale[i];               /* :432-436 */

  float factor = (float)(1.0f / scaleFactor); /* :441 */
  float nDesired =
      nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)); /* :443 */
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    p->quota[level] = orc_cv_round_f(nDesired);
    sum += p->quota[level];
    nDesired *= factor;
  }
  p->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0; /* :451 */

  memcpy(p->pattern, pattern1024, 1024);

  /* umax, :457-475 */
  int v, v0;
  int vmax = cv_floor_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
  int vmin = cv_ceil_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) p->umax[v] = cv_round_d(sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (p->umax[v0] == p->umax[v0 + 1]) ++v0;
    p->umax[v] = v0;
    ++v0;
  }
}

/* level size, ORBextractor.cpp:1119-1120 */
void orc_level_size(const orc_orb_params *p, int w, int h, int level, int *lw, int *lh) {
  float scale = p->inv_scale[level];
  *lw = orc_cv_round_f((float)w * scale);
  *lh = orc_cv_round_f((float)h * scale);
}

/* ------------------------------------------------------------------ E1 ---- */
/* cv::resize INTER_LINEAR 8UC1 (OpenCV 3.x imgproc/resize: resizeGeneric_ with
 * HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>).
 * Called at ORBextractor.cpp:1129. */
static short sat_short_from_float(float v) {
  int i = orc_cv_round_f(v);
  if (i > 32767) i = 32767;
  if (i < -32768) i = -32768;
  return (short)i;
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                          int dh, int dstride) {
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int *xofs = (int *)malloc(sizeof(int) * dw);
  short *ialpha = (short *)malloc(sizeof(short) * dw * 2);
  int *yofs = (int *)malloc(sizeof(int) * dh);
  short *ibeta = (short *)malloc(sizeof(short) * dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx + 1 >= sw) {
      if (dx < xmax) xmax = dx;
      if (sx >= sw - 1) {
        fx = 0;
        sx = sw - 1;
      }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_short_from_float((1.f - fx) * 2048);
    ialpha[dx * 2 + 1] = sat_short_from_float(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = sat_short_from_float((1.f - fy) * 2048);
    ibeta[dy * 2 + 1] = sat_short_from_float(fy * 2048);
  }
  int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
  for (int dy = 0; dy < dh; dy++) {
    int sy = yofs[dy];
    int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
    int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
    const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      if (dx < xmax) {
        int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
        row0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
        row1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
      } else {
        row0[dx] = S0[sx] * 2048;
        row1[dx] = S1[sx] * 2048;
      }
    }
    int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t *D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++) {
      int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
      D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  free(xofs);
  free(ialpha);
  free(yofs);
  free(ibeta);
  free(row0);
  free(row1);
}

/* ------------------------------------------------------------------ E7 ---- */
/* cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) on a cloned ROI, ORBextractor.cpp:1093-1094.
 * getGaussianKernel(7, 2, CV_32F) -> convertTo(CV_32S, 256) -> SymmRowSmallFilter<uchar,int> ->
 * SymmColumnFilter<FixedPtCastEx<int,uchar>(16)>. */
static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * n - 2 - i;
  }
  return i;
}

static void gaussian7_kernel_q8(int k[7]) {
  float cf[7];
  double sum = 0;
  double scale2X = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; i++) {
    double x = i - (7 - 1) * 0.5;
    double t = exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cv_round_d((double)cf[i] * 256.0);
  }
}

void orc_gaussian7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
  int k[7];
  gaussian7_kernel_q8(k); /* {18,34,49,55,49,34,18}: sums to 257, as old OpenCV does */
  int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t *S = src + (size_t)y * sstride;
    for (int x = 0; x < w; x++) {
      int acc = 0;
      for (int i = 0; i < 7; i++) acc += k[i] * S[reflect101(x + i - 3, w)];
      tmp[(size_t)y * w + x] = acc;
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t *D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      int acc = 0;
      for (int j = 0; j < 7; j++) acc += k[j] * tmp[(size_t)reflect101(y + j - 3, h) * w + x];
      int v = (acc + (1 << 15)) >> 16;
      D[x] = (uint8_t)(v > 255 ? 255 : (v < 0 ? 0 : v));
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------ E2 ---- */
/* cv::FAST(..., threshold, nonmaxSuppression=true) = FAST_t<16> (OpenCV 3.x features2d/fast.cpp),
 * called per cell at ORBextractor.cpp:817-823. */
static const int fast_off16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                      {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                      {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

static int fast_corner_score16(const uint8_t *ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[25];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    a = a < d[k + 3] ? a : d[k + 3];
    if (a <= a0) continue;
    a = a < d[k + 4] ? a : d[k + 4];
    a = a < d[k + 5] ? a : d[k + 5];
    a = a < d[k + 6] ? a : d[k + 6];
    a = a < d[k + 7] ? a : d[k + 7];
    a = a < d[k + 8] ? a : d[k + 8];
    int t = a < d[k] ? a : d[k];
    a0 = a0 > t ? a0 : t;
    t = a < d[k + 9] ? a : d[k + 9];
    a0 = a0 > t ? a0 : t;
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    b = b > d[k + 3] ? b : d[k + 3];
    b = b > d[k + 4] ? b : d[k + 4];
    b = b > d[k + 5] ? b : d[k + 5];
    if (b >= b0) continue;
    b = b > d[k + 6] ? b : d[k + 6];
    b = b > d[k + 7] ? b : d[k + 7];
    b = b > d[k + 8] ? b : d[k + 8];
    int t = b > d[k] ? b : d[k];
    b0 = b0 < t ? b0 : t;
    t = b > d[k + 9] ? b : d[k + 9];
    b0 = b0 < t ? b0 : t;
  }
  return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, int cols, int rows, int stride, int threshold, int nms,
                 int *xs, int *ys, int *scores, int cap) {
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = fast_off16[k][0] + fast_off16[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  if (threshold > 255) threshold = 255;
  if (threshold < 0) threshold = 0;
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++)
    threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  uint8_t *bufmem = (uint8_t *)calloc((size_t)cols * 3, 1);
  uint8_t *buf[3] = {bufmem, bufmem + cols, bufmem + cols * 2};
  int *cpmem = (int *)malloc(sizeof(int) * (size_t)(cols + 1) * 3);
  int *cpbuf[3] = {cpmem + 1, cpmem + (cols + 1) + 1, cpmem + 2 * (cols + 1) + 1};
  int nout = 0;
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t *ptr = img + (size_t)i * stride + 3;
    uint8_t *curr = buf[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t *tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        int is_corner = 0;
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                is_corner = 1;
                break;
              }
            } else
              count = 0;
          }
        }
        if (!is_corner && (d & 2)) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                is_corner = 1;
                break;
              }
            } else
              count = 0;
          }
        }
        if (is_corner) {
          cornerpos[ncorners++] = j;
          if (nms) curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = buf[(i - 4 + 3) % 3];
    const uint8_t *pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                   score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                   score > curr[j] && score > curr[j + 1])) {
        if (nout < cap) {
          xs[nout] = j;
          ys[nout] = i - 1;
          scores[nout] = score;
        }
        nout++;
      }
    }
  }
  free(bufmem);
  free(cpmem);
  return nout;
}

/* cell loop of ComputeKeyPointsOctTree, ORBextractor.cpp:776-837 (one level) */
int orc_level_candidates(const orc_orb_params *p, const uint8_t *img, int w, int h, int stride,
                         float *cx, float *cy, float *cresp, int cap) {
  const float W = 30;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols < 1 || nRows < 1) return 0;
  const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
  int tmpcap = (wCell + 6) * (hCell + 6);
  int *xs = (int *)malloc(sizeof(int) * tmpcap * 3), *ys = xs + tmpcap, *sc = ys + tmpcap;
  int n = 0;
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
      const uint8_t *sub = img + (size_t)y0 * stride + x0;
      int m = orc_fast9_16(sub, cw, ch, stride, p->ini_th, 1, xs, ys, sc, tmpcap);
      if (m == 0) m = orc_fast9_16(sub, cw, ch, stride, p->min_th, 1, xs, ys, sc, tmpcap);
      for (int k = 0; k < m; k++) {
        if (n < cap) {
          cx[n] = (float)xs[k] + (float)(j * wCell);
          cy[n] = (float)ys[k] + (float)(i * hCell);
          cresp[n] = (float)sc[k];
        }
        n++;
      }
    }
  }
  free(xs);
  return n;
}

/* ------------------------------------------------------------- E3 / E4 ---- */
/* ExtractorNode::DivideNode (:487-543) and ORBextractor::DistributeOctTree (:545-769), with the
 * std::list emulated by a doubly linked pool so that push_front / erase / iteration order are the
 * reference's.  Q-E3: the reference sorts (size, ExtractorNode*) pairs, i.e. breaks size ties by
 * heap address; this restatement breaks them by node creation order (later-created = larger). */
typedef struct {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  int *keys;
  int nkeys;
  int no_more;
  int prev, next; /* list links (pool indices), -1 = none */
  int seq;
} onode;

typedef struct {
  onode *pool;
  int npool, cap;
  int head, tail, size;
  int seq;
} olist;

static int ol_new(olist *L) {
  if (L->npool == L->cap) {
    L->cap = L->cap ? L->cap * 2 : 256;
    L->pool = (onode *)realloc(L->pool, sizeof(onode) * L->cap);
  }
  onode *nd = &L->pool[L->npool];
  memset(nd, 0, sizeof(*nd));
  nd->prev = nd->next = -1;
  nd->seq = L->seq++;
  return L->npool++;
}
static void ol_push_front(olist *L, int id) {
  L->pool[id].prev = -1;
  L->pool[id].next = L->head;
  if (L->head >= 0) L->pool[L->head].prev = id;
  L->head = id;
  if (L->tail < 0) L->tail = id;
  L->size++;
}
static void ol_push_back(olist *L, int id) {
  L->pool[id].next = -1;
  L->pool[id].prev = L->tail;
  if (L->tail >= 0) L->pool[L->tail].next = id;
  L->tail = id;
  if (L->head < 0) L->head = id;
  L->size++;
}
static int ol_erase(olist *L, int id) { /* returns next */
  int p = L->pool[id].prev, n = L->pool[id].next;
  if (p >= 0) L->pool[p].next = n;
  else L->head = n;
  if (n >= 0) L->pool[n].prev = p;
  else L->tail = p;
  L->size--;
  return n;
}

static void divide_node(olist *L, int id, const float *cx, const float *cy, int ch[4]) {
  /* children are created as detached pool nodes; caller links the non-empty ones */
  for (int c = 0; c < 4; c++) ch[c] = ol_new(L);
  onode *P = &L->pool[id];
  const int halfX = (int)ceilf((float)(P->URx - P->ULx) / 2);
  const int halfY = (int)ceilf((float)(P->BRy - P->ULy) / 2);
  onode *n1 = &L->pool[ch[0]], *n2 = &L->pool[ch[1]], *n3 = &L->pool[ch[2]], *n4 = &L->pool[ch[3]];
  n1->ULx = P->ULx, n1->ULy = P->ULy;
  n1->URx = P->ULx + halfX, n1->URy = P->ULy;
  n1->BLx = P->ULx, n1->BLy = P->ULy + halfY;
  n1->BRx = P->ULx + halfX, n1->BRy = P->ULy + halfY;
  n2->ULx = n1->URx, n2->ULy = n1->URy;
  n2->URx = P->URx, n2->URy = P->URy;
  n2->BLx = n1->BRx, n2->BLy = n1->BRy;
  n2->BRx = P->URx, n2->BRy = P->ULy + halfY;
  n3->ULx = n1->BLx, n3->ULy = n1->BLy;
  n3->URx = n1->BRx, n3->URy = n1->BRy;
  n3->BLx = P->BLx, n3->BLy = P->BLy;
  n3->BRx = n1->BRx, n3->BRy = P->BLy;
  n4->ULx = n3->URx, n4->ULy = n3->URy;
  n4->URx = n2->BRx, n4->URy = n2->BRy;
  n4->BLx = n3->BRx, n4->BLy = n3->BRy;
  n4->BRx = P->BRx, n4->BRy = P->BRy;
  for (int c = 0; c < 4; c++) {
    L->pool[ch[c]].keys = (int *)malloc(sizeof(int) * (P->nkeys > 0 ? P->nkeys : 1));
    L->pool[ch[c]].nkeys = 0;
  }
  for (int i = 0; i < P->nkeys; i++) {
    int k = P->keys[i];
    onode *t;
    if (cx[k] < (float)n1->URx) {
      if (cy[k] < (float)n1->BRy) t = n1;
      else t = n3;
    } else if (cy[k] < (float)n1->BRy)
      t = n2;
    else
      t = n4;
    t->keys[t->nkeys++] = k;
  }
  for (int c = 0; c < 4; c++)
    if (L->pool[ch[c]].nkeys == 1) L->pool[ch[c]].no_more = 1;
}

typedef struct {
  int size, seq, id;
} size_node;
static int cmp_size_node(const void *a, const void *b) {
  const size_node *x = (const size_node *)a, *y = (const size_node *)b;
  if (x->size != y->size) return x->size < y->size ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int orc_distribute_octtree(const float *cx, const float *cy, const float *cresp, int n, int minX,
                           int maxX, int minY, int maxY, int N, int *out_idx, int cap) {
  const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY)); /* :549 */
  if (nIni < 1) return -1; /* reference divides by zero here (portrait images) */
  const float hX = (float)(maxX - minX) / nIni;
  olist L;
  memset(&L, 0, sizeof(L));
  L.head = L.tail = -1;
  int *ini = (int *)malloc(sizeof(int) * nIni);
  for (int i = 0; i < nIni; i++) {
    int id = ol_new(&L);
    onode *ni = &L.pool[id];
    ni->ULx = (int)(hX * (float)i), ni->ULy = 0;
    ni->URx = (int)(hX * (float)(i + 1)), ni->URy = 0;
    ni->BLx = ni->ULx, ni->BLy = maxY - minY;
    ni->BRx = ni->URx, ni->BRy = maxY - minY;
    ni->keys = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    ol_push_back(&L, id);
    ini[i] = id;
  }
  for (int i = 0; i < n; i++) {
    int b = (int)(cx[i] / hX); /* :574 */
    if (b < 0) b = 0;
    if (b >= nIni) b = nIni - 1; /* reference would index out of bounds */
    onode *t = &L.pool[ini[b]];
    t->keys[t->nkeys++] = i;
  }
  for (int lit = L.head; lit >= 0;) { /* :579-590 */
    onode *nd = &L.pool[lit];
    if (nd->nkeys == 1) {
      nd->no_more = 1;
      lit = nd->next;
    } else if (nd->nkeys == 0)
      lit = ol_erase(&L, lit);
    else
      lit = nd->next;
  }
  int bFinish = 0;
  size_node *vsz = (size_node *)malloc(sizeof(size_node) * (size_t)(4 * (n + nIni) + 16));
  size_node *vprev = (size_node *)malloc(sizeof(size_node) * (size_t)(4 * (n + nIni) + 16));
  int nvsz = 0;
  while (!bFinish) {
    int prevSize = L.size;
    int lit = L.head;
    int nToExpand = 0;
    nvsz = 0;
    while (lit >= 0) {
      if (L.pool[lit].no_more) {
        lit = L.pool[lit].next;
        continue;
      }
      int ch[4];
      divide_node(&L, lit, cx, cy, ch);
      for (int c = 0; c < 4; c++) {
        if (L.pool[ch[c]].nkeys > 0) {
          ol_push_front(&L, ch[c]);
          if (L.pool[ch[c]].nkeys > 1) {
            nToExpand++;
            vsz[nvsz].size = L.pool[ch[c]].nkeys;
            vsz[nvsz].seq = L.pool[ch[c]].seq;
            vsz[nvsz].id = ch[c];
            nvsz++;
          }
        }
      }
      lit = ol_erase(&L, lit);
    }
    if (L.size >= N || L.size == prevSize) {
      bFinish = 1;
    } else if (L.size + nToExpand * 3 > N) {
      while (!bFinish) {
        prevSize = L.size;
        int nprev = nvsz;
        memcpy(vprev, vsz, sizeof(size_node) * nprev);
        nvsz = 0;
        qsort(vprev, nprev, sizeof(size_node), cmp_size_node);
        for (int j = nprev - 1; j >= 0; j--) {
          int ch[4];
          divide_node(&L, vprev[j].id, cx, cy, ch);
          for (int c = 0; c < 4; c++) {
            if (L.pool[ch[c]].nkeys > 0) {
              ol_push_front(&L, ch[c]);
              if (L.pool[ch[c]].nkeys > 1) {
                vsz[nvsz].size = L.pool[ch[c]].nkeys;
                vsz[nvsz].seq = L.pool[ch[c]].seq;
                vsz[nvsz].id = ch[c];
                nvsz++;
              }
            }
          }
          ol_erase(&L, vprev[j].id);
          if (L.size >= N) break;
        }
        if (L.size >= N || L.size == prevSize) bFinish = 1;
      }
    }
  }
  /* retain best response per node (:748-766) */
  int nout = 0;
  for (int lit = L.head; lit >= 0; lit = L.pool[lit].next) {
    onode *nd = &L.pool[lit];
    int best = nd->keys[0];
    float maxResponse = cresp[best];
    for (int k = 1; k < nd->nkeys; k++) {
      if (cresp[nd->keys[k]] > maxResponse) {
        best = nd->keys[k];
        maxResponse = cresp[best];
      }
    }
    if (nout < cap) out_idx[nout] = best;
    nout++;
  }
  for (int i = 0; i < L.npool; i++) free(L.pool[i].keys);
  free(L.pool);
  free(ini);
  free(vsz);
  free(vprev);
  return nout;
}

/* ------------------------------------------------------------------ E6 ---- */
/* cv::fastAtan2 (OpenCV 3.x core/mathfuncs_core scalar version), degrees in [0,360) */
float orc_fast_atan2(float y, float x) {
  static const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* IC_Angle, ORBextractor.cpp:79-107 */
float orc_ic_angle(const uint8_t *img, int stride, int px, int py, const int *umax) {
  int m_01 = 0, m_10 = 0;
  const uint8_t *center = img + (size_t)py * stride + px;
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ E8 ---- */
/* cos/sin used by computeOrbDescriptor (:114-115).  The reference calls glibc cosf/sinf through
 * the std:: float overloads.  To make the device and the oracle agree bit-for-bit the contract is:
 * evaluate in double with the fixed-order Cody-Waite + minimax polynomial below and round once to
 * float.  tests/test_oracle_orb.py sweeps this against glibc cosf/sinf (orc_cos_sin_f_libm). */
static double poly_sin(double r) { /* |r| <= pi/4 */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = S6;
  p = p * z + S5;
  p = p * z + S4;
  p = p * z + S3;
  p = p * z + S2;
  p = p * z + S1;
  return r + r * (z * p);
}
static double poly_cos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = C6;
  p = p * z + C5;
  p = p * z + C4;
  p = p * z + C3;
  p = p * z + C2;
  p = p * z + C1;
  return (1.0 - 0.5 * z) + z * (z * p);
}
void orc_cos_sin_f(float angle_rad, float *c, float *s) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
  const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */
  double x = (double)angle_rad;
  double kd = floor(x * TWO_OVER_PI + 0.5);
  int k = (int)kd;
  double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
  double sr = poly_sin(r), cr = poly_cos(r);
  double cs, sn;
  switch (k & 3) {
    case 0: cs = cr, sn = sr; break;
    case 1: cs = -sr, sn = cr; break;
    case 2: cs = -cr, sn = -sr; break;
    default: cs = sr, sn = -cr; break;
  }
  *c = (float)cs;
  *s = (float)sn;
}
void orc_cos_sin_f_libm(float angle_rad, float *c, float *s) {
  *c = cosf(angle_rad);
  *s = sinf(angle_rad);
}

/* computeOrbDescriptor, ORBextractor.cpp:110-151 */
void orc_orb_descriptor(const uint8_t *blur, int stride, int px, int py, float angle_deg,
                        const int8_t *pattern, uint8_t *desc32) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f); /* :109 */
  float angle = angle_deg * factorPI;
  float a, b;
  orc_cos_sin_f(angle, &a, &b);
  const uint8_t *center = blur + (size_t)py * stride + px;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int t = 0; t < 8; t++) {
      const int8_t *pp = pattern + (i * 16 + t * 2) * 2;
      float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
      int t0 = center[orc_cv_round_f(x0 * b + y0 * a) * stride + orc_cv_round_f(x0 * a - y0 * b)];
      int t1 = center[orc_cv_round_f(x1 * b + y1 * a) * stride + orc_cv_round_f(x1 * a - y1 * b)];
      val |= (t0 < t1) << t;
    }
    desc32[i] = (uint8_t)val;
  }
}

/* the same with glibc's cosf / sinf, what a reference binary calls (:114-115): used to MEASURE how many descriptor
 * bytes the correctly-rounded contract above can change (tests/test_oracle_orb.py) */
void orc_orb_descriptor_libm(const uint8_t *blur, int stride, int px, int py, float angle_deg,
                             const int8_t *pattern, uint8_t *desc32) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float angle = angle_deg * factorPI;
  float a, b;
  orc_cos_sin_f_libm(angle, &a, &b);
  const uint8_t *center = blur + (size_t)py * stride + px;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int t = 0; t < 8; t++) {
      const int8_t *pp = pattern + (i * 16 + t * 2) * 2;
      float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
      int t0 = center[orc_cv_round_f(x0 * b + y0 * a) * stride + orc_cv_round_f(x0 * a - y0 * b)];
      int t1 = center[orc_cv_round_f(x1 * b + y1 * a) * stride + orc_cv_round_f(x1 * a - y1 * b)];
      val |= (t0 < t1) << t;
    }
    desc32[i] = (uint8_t)val;
  }
}

/* ------------------------------------------------------------------ E9 ---- */
/* ORBextractor::operator(), ORBextractor.cpp:1051-1112 (+ComputePyramid :1115-1142,
 * ComputeKeyPointsOctTree :771-861).  The 19-px REFLECT_101 padding of each level is never read by
 * any later stage (FAST stays inside [16,w-16), the patch inside [4,w-4), the blur reflects at the
 * ROI edge of a clone) so levels are kept unpadded. */
int orc_orb_extract(const orc_orb_params *p, const uint8_t *img, int w, int h, int stride,
                    orc_keypoint *kps, uint8_t *desc, int cap, int *n_per_level) {
  if (!img || w <= 0 || h <= 0) return 0; /* :1054-1055 */
  int nl = p->nlevels;
  uint8_t *lev[ORC_MAX_LEVELS];
  int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
  for (int l = 0; l < nl; l++) {
    orc_level_size(p, w, h, l, &lw[l], &lh[l]);
    lev[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
    if (l == 0) {
      for (int y = 0; y < h; y++) memcpy(lev[0] + (size_t)y * w, img + (size_t)y * stride, w);
    } else {
      orc_resize_linear_u8(lev[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lev[l], lw[l], lh[l], lw[l]);
    }
  }
  int total = 0;
  int ccap = 0;
  for (int l = 0; l < nl; l++) ccap = lw[l] * lh[l] > ccap ? lw[l] * lh[l] : ccap;
  float *cx = (float *)malloc(sizeof(float) * ccap * 3), *cy = cx + ccap, *cr = cy + ccap;
  int *sel = (int *)malloc(sizeof(int) * (ccap + 1));
  for (int l = 0; l < nl; l++) {
    const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
    const int maxBX = lw[l] - EDGE_THRESHOLD + 3, maxBY = lh[l] - EDGE_THRESHOLD + 3;
    int nc = orc_level_candidates(p, lev[l], lw[l], lh[l], lw[l], cx, cy, cr, ccap);
    int nk = 0;
    if (nc > 0) nk = orc_distribute_octtree(cx, cy, cr, nc, minBX, maxBX, minBY, maxBY, p->quota[l], sel, ccap);
    if (nk < 0) nk = 0;
    if (n_per_level) n_per_level[l] = nk;
    if (nk == 0) continue;
    const int scaledPatchSize = (int)(PATCH_SIZE * p->scale[l]); /* :842 */
    uint8_t *blur = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
    orc_gaussian7_u8(lev[l], lw[l], lh[l], lw[l], blur, lw[l]);
    for (int i = 0; i < nk; i++) {
      if (total >= cap) break;
      orc_keypoint *k = &kps[total];
      k->x = cx[sel[i]] + (float)minBX;
      k->y = cy[sel[i]] + (float)minBY;
      k->response = cr[sel[i]];
      k->octave = l;
      k->class_id = -1;
      k->size = (float)scaledPatchSize;
      int px = orc_cv_round_f(k->x), py = orc_cv_round_f(k->y);
      k->angle = orc_ic_angle(lev[l], lw[l], px, py, p->umax); /* unblurred level, :859-860 */
      orc_orb_descriptor(blur, lw[l], px, py, k->angle, p->pattern, desc + (size_t)total * 32);
      if (l != 0) { /* :1102-1108 */
        float scale = p->scale[l];
        k->x *= scale;
        k->y *= scale;
      }
      total++;
    }
    free(blur);
  }
  free(cx);
  free(sel);
  for (int l = 0; l < nl; l++) free(lev[l]);
  return total;
}
