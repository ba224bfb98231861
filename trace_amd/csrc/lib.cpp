// Error plumbing + version probe for libtrace_amd.so.
#include <stdarg.h>

#include "common.h"

namespace trace {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace trace

extern "C" const char* trace_last_error(void) { return trace::g_err; }
extern "C" int trace_abi_version(void) { return 1; }
