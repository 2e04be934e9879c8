// Library-wide state: thread-local error string, version, architecture tag.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ns_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ns_version(void) { return 100; }
extern "C" const char* ns_device_arch(void) { return "gfx950"; }
extern "C" const char* ns_last_error(void) { return g_err; }
