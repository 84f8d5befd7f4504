// core.hip -- error plumbing and version of libhipie_mi355.
#include <stdarg.h>

#include "common.h"

namespace hipie {

thread_local char g_err[512] = {0};

int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(HIPIE_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return HIPIE_OK;
}

}  // namespace hipie

extern "C" int hipie_version(void) { return HIPIE_ABI_VERSION; }
extern "C" const char* hipie_last_error(void) { return hipie::g_err; }
