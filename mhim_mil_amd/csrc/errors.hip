// errors.cpp — thread-local last-error string behind mhimx_last_error().
#include <stdarg.h>

#include "common.hpp"

namespace mhimx {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
const std::string& last_error() { return g_err; }
}  // namespace mhimx

extern "C" const char* mhimx_last_error(void) { return mhimx::last_error().c_str(); }
extern "C" int mhimx_version(void) { return MHIMX_VERSION; }
