// acez_common.hip -- error state, version and device probing of the C ABI (include/acez.h).
#include "acez_common.h"
#include <string.h>

namespace acez {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace acez

extern "C" const char* acez_last_error(void) { return acez::g_err; }
extern "C" const char* acez_version(void) { return "acez 0.1 (gfx950)"; }
extern "C" int acez_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
