// Host-side ABI plumbing of libvlp_hip.so: version, thread-local error string.
#include <hip/hip_runtime_api.h>
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

// ---- device selection on entry (common.h) -----------------------------------------------------------------------------------------
static thread_local int t_device = -1;         // device this thread last made current through the library
static int device_count() {
    static int n = [] { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; } return c; }();
    return n;
}
int vlp_current_device(void) {
    if (t_device >= 0) return t_device;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
int vlp_enter_device(const void* p, const char* who) {
    if (p == nullptr) return VLP_OK;           // the entry's own argument check reports the null operand
    if (device_count() == 1) {                 // a single-GPU process: nothing to select, and no driver query per call
        if (t_device != 0) { (void)hipSetDevice(0); t_device = 0; }
        return VLP_OK;
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return vlp_set_error(VLP_ERR_BAD_ARG, "%s: operand %p is not device memory (libvlp_hip has no CPU path)", who, p);
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeUnified)
        return vlp_set_error(VLP_ERR_BAD_ARG, "%s: operand %p is host memory (libvlp_hip has no CPU path)", who, p);
    if (at.device != t_device) {
        if (hipSetDevice(at.device) != hipSuccess) return vlp_set_error(VLP_ERR_HIP, "%s: hipSetDevice(%d): %s", who, at.device, hipGetErrorString(hipGetLastError()));
        t_device = at.device;
    }
    return VLP_OK;
}
