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
// 2: nsky_gemm_desc gained rowsum_k_limit; nsky_split_planes / nsky_gemm_f32_planes, nsky_main_losses_*, nsky_ddf_losses_* added
extern "C" int nsky_abi_version(void) { return 16; }
