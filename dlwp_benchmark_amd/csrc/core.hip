// Library-level entry points: version, error slot, device count.
#include "common.hpp"

namespace dlwp {

std::string& last_error_slot() {
  static thread_local std::string slot;
  return slot;
}

int32_t fail(int32_t code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_slot() = buf;
  return code;
}

}  // namespace dlwp

extern "C" int32_t dlwp_version(void) { return 100; }

extern "C" const char* dlwp_last_error(void) { return dlwp::last_error_slot().c_str(); }

extern "C" int32_t dlwp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    dlwp::fail(DLWP_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}
