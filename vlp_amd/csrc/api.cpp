// Host-side ABI plumbing of libvlp_hip.so: version, thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "vlp_hip.h"

static thread_local char g_err[512] = "";

int vlp_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int vlp_version(void) { return VLP_ABI_VERSION; }
extern "C" const char* vlp_last_error_string(void) { return g_err; }
