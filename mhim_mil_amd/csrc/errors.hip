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

// the bag batch the calling thread is enqueueing (common.hpp: BagBatch)
static thread_local BagBatch g_batch = {};
const BagBatch& cur_batch() { return g_batch; }
void set_batch(const BagBatch* b) {
  if (b) g_batch = *b;
  else g_batch = BagBatch{};
}
}  // namespace mhimx

extern "C" const char* mhimx_last_error(void) { return mhimx::last_error().c_str(); }
extern "C" int mhimx_version(void) { return MHIMX_VERSION; }
