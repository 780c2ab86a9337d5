// Thread-local error text + ABI version for libneusky_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/neusky_hip.h"

static thread_local char g_err[512] = "";

void nsky_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nsky_last_error(void) { return g_err; }
extern "C" int nsky_abi_version(void) { return 1; }
