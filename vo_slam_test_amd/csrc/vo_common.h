// vo_common.h -- shared host/device helpers for the gfx950 library (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

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
  int reserve(size_t n) {
    if (n <= bytes) return VO_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    VO_HIP_CHECK(hipMalloc(&p, n));
    bytes = n;
    return VO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

constexpr int kWave = 64;

// Dense SPD solve on the device (csrc/pose_graph.hip): A is ld x ld row-major with ld a multiple of 64
// (identity on the padding diagonal), lower triangle factored in place (L L^T, 64-wide panels,
// trailing update on the FP64 matrix cores), rhs -> solution.  *fail (device int, zeroed by the
// caller) is set when a pivot is not positive.  Enqueues kernels only; no synchronisation.
constexpr int kCholPanel = 64;
void chol_factor_solve(double *A, int ld, double *rhs, int *fail, hipStream_t st);

}  // namespace vo
