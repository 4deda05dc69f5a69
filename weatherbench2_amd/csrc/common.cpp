// Error plumbing of the C ABI (include/wb2hip.h).
#include "common.hpp"
#include "wb2hip.h"

namespace wb2 {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return -1;
}

}  // namespace wb2

extern "C" {
int wb2_version(void) { return WB2_VERSION; }
const char* wb2_last_error(void) { return wb2::error_buffer(); }
}
