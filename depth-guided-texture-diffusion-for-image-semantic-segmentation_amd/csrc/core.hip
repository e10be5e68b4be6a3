// core.hip — version + thread-local error text for the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void dgtd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dgtd_version(void) { return 100; }
extern "C" const char* dgtd_last_error(void) { return g_err; }
