// vo_common.hip -- error reporting and device bring-up shared by every C-ABI entry point.
#include "vo_common.h"

#include <vector>

#include <mutex>

namespace vo {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ensure_device() {
  static std::once_flag once;
  static int status = VO_ERR_NO_DEVICE;
  std::call_once(once, [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
      status = VO_ERR_NO_DEVICE;
      return;
    }
    status = VO_OK;
  });
  if (status != VO_OK)
    set_error("no usable HIP device: this library has no CPU fallback (build target gfx950 / MI355X)");
  return status;
}

static thread_local std::vector<DevBuf *> *g_scratch = nullptr;  // heap: no destructor order issues at thread exit
ScratchBuf::ScratchBuf() {
  if (!g_scratch) g_scratch = new std::vector<DevBuf *>();
  g_scratch->push_back(this);
}
size_t release_thread_scratch() {
  size_t freed = 0;
  if (g_scratch)
    for (DevBuf *b : *g_scratch) freed += b->bytes, b->release();
  return freed;
}

hipStream_t thread_stream() {
  thread_local hipStream_t st = nullptr;
  thread_local bool tried = false;
  if (!tried) {
    tried = true;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      st = nullptr;  // the NULL stream still gives correct results
    }
  }
  return st;
}

int copy_h2d(void *dst, const void *src, size_t bytes, hipStream_t st, const char *what) {
  if (!bytes) return VO_OK;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) {
    set_error("%s: host-to-device copy of %zu bytes failed: %s", what, bytes, hipGetErrorString(e));
    return VO_ERR_HIP;
  }
  return VO_OK;
}

int copy_d2h(void *dst, const void *src, size_t bytes, hipStream_t st, const char *what) {
  if (!bytes) return VO_OK;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) {
    set_error("%s: device-to-host copy of %zu bytes failed: %s", what, bytes, hipGetErrorString(e));
    return VO_ERR_HIP;
  }
  return VO_OK;
}

int stream_sync(hipStream_t st, const char *what) {
  const hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    set_error("%s: kernel or copy failed: %s", what, hipGetErrorString(e));
    return VO_ERR_HIP;
  }
  return VO_OK;
}

int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st, const char *what) {
  VO_CHECK(b.reserve(bytes > 64 ? bytes : 64));
  return copy_h2d(b.p, src, bytes, st, what);
}

}  // namespace vo

extern "C" {
const char *vo_last_error(void) { return vo::g_err; }
int vo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char *vo_version(void) { return "vo_slam_test_amd 0.1 (gfx950)"; }
size_t vo_release_thread_scratch(void) { return vo::release_thread_scratch(); }
}
