// Host-side ABI plumbing of libvlp_hip.so: version, thread-local error string.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

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
extern "C" int vlp_lab_build(void) {
#ifdef VLP_LAB_BUILD
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* vlp_last_error_string(void) { return g_err; }

// ---- device selection on entry (common.h) -----------------------------------------------------------------------------------------
// The library never trusts a private notion of "the current device": the host (torch.cuda.set_device, `with torch.cuda.device()`) may
// change it between two calls.  Every entry compares the device that owns its first operand with hipGetDevice() (thread-local inside the
// runtime, no driver call), switches if they differ and RESTORES the caller's device on return (VlpDeviceGuard, common.h) -- so torch's
// own bookkeeping of the current device stays true.  Which device owns a pointer is a driver query (hipPointerGetAttributes); it is
// asked once per allocation and remembered per thread as an address range (the caching allocator hands out views of a few dozen
// segments), re-validated every VLP_RANGE_REVALIDATE hits in case an allocation was freed and its addresses re-used on another device.
static int device_count() {
    static int n = [] {
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
        if (const char* e = getenv("VLP_FAKE_DEVICE_COUNT")) { const int f = atoi(e); if (f > 0) c = f; }   // tests: the multi-device path on a 1-GPU box
        return c;
    }();
    return n;
}
int vlp_current_device(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
#define VLP_RANGE_SLOTS 64
#define VLP_RANGE_REVALIDATE 256
struct VlpRange { uintptr_t lo, hi; int dev; unsigned hits; };
static thread_local VlpRange t_ranges[VLP_RANGE_SLOTS];
static thread_local int t_nranges = 0, t_next = 0;
static thread_local unsigned long long t_lookups = 0, t_queries = 0;      // vlp_debug_device_lookup_stats
// force: ask the driver even on a cache hit (the guard does so before it SWITCHES devices on the word of a cached entry: a freed segment
// whose addresses were re-used by an allocation on another device would otherwise send up to VLP_RANGE_REVALIDATE launches to the wrong
// device; a hit that agrees with the caller's current device -- every call of a one-device-per-process job -- costs no driver call)
static int owner_of(const void* p, const char* who, int* dev, bool force = false) {
    const uintptr_t a = (uintptr_t)p;
    ++t_lookups;
    for (int i = 0; i < t_nranges; ++i) {
        VlpRange& r = t_ranges[i];
        if (a >= r.lo && a < r.hi) {
            if (!force && (++r.hits % VLP_RANGE_REVALIDATE)) { *dev = r.dev; return VLP_OK; }
            r = t_ranges[--t_nranges];            // (periodic) re-validation: drop the entry and ask the driver again
            break;
        }
    }
    ++t_queries;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return vlp_set_error(VLP_ERR_BAD_ARG, "%s: operand %p is not device memory (libvlp_hip has no CPU path)", who, p);
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeUnified)
        return vlp_set_error(VLP_ERR_BAD_ARG, "%s: operand %p is host memory (libvlp_hip has no CPU path)", who, p);
    *dev = at.device;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size > 0) {
        VlpRange r = {(uintptr_t)base, (uintptr_t)base + size, at.device, 0u};
        if (t_nranges < VLP_RANGE_SLOTS) t_ranges[t_nranges++] = r;
        else { t_ranges[t_next] = r; t_next = (t_next + 1) % VLP_RANGE_SLOTS; }
    } else {
        (void)hipGetLastError();
    }
    return VLP_OK;
}
VlpDeviceGuard::VlpDeviceGuard(const void* p, const char* who) : prev(-1), rc(VLP_OK) {
    if (p == nullptr) return;                  // the entry's own argument check reports the null operand
    int dev = 0;
    rc = owner_of(p, who, &dev);               // also refuses host pointers, on single-GPU processes too
    if (rc != VLP_OK || device_count() <= 1) return;
    const int cur = vlp_current_device();
    if (dev == cur) return;
    rc = owner_of(p, who, &dev, true);         // about to switch devices: on the driver's word, not a cached range's
    if (rc != VLP_OK || dev == cur) return;
    if (hipSetDevice(dev) != hipSuccess) {
        rc = vlp_set_error(VLP_ERR_HIP, "%s: hipSetDevice(%d): %s", who, dev, hipGetErrorString(hipGetLastError()));
        return;
    }
    prev = cur;
}
VlpDeviceGuard::~VlpDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);   // the caller's current device is left as it was found
}
// test hook: (owner lookups, driver queries) of the calling thread
extern "C" void vlp_debug_device_lookup_stats(unsigned long long* lookups, unsigned long long* driver_queries) {
    if (lookups) *lookups = t_lookups;
    if (driver_queries) *driver_queries = t_queries;
}
