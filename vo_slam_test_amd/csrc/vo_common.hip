// vo_common.hip -- error reporting and device bring-up shared by every C-ABI entry point.
#include "vo_common.h"

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

}  // namespace vo

extern "C" {
const char *vo_last_error(void) { return vo::g_err; }
int vo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char *vo_version(void) { return "vo_slam_test_amd 0.1 (gfx950)"; }
}
