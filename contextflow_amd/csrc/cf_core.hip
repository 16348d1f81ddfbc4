// ABI core: version + thread-local error string.
#include "cf_common.h"
#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_err[512] = "";
}

void cf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int cf_abi_version(void) { return CF_ABI_VERSION; }
const char* cf_last_error(void) { return g_err; }
}
