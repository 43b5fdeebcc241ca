// libgimhip runtime glue: version, thread-local error string, launch check.
#include "gim_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void gim_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gim_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gim_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return GIM_ERR_LAUNCH;
    }
    return GIM_OK;
}

extern "C" int gim_version(void) { return 100; }
extern "C" const char* gim_last_error(void) { return g_err; }
extern "C" int gim_ktile_bytes(void) { return 128; }
extern "C" int gim_npad_granule(void) { return 64; }

// fp16 range guard word (gim_common.h): one registration per device, process-wide.  The launch wrappers of the kernels that store
// residual streams read it when they build their arguments, so a captured graph keeps the pointer that was registered at capture.
static std::atomic<int*> g_range_guard[256];
extern "C" int gim_set_range_guard(int32_t* device_word) {
    g_range_guard[GimPerDevice::dev()].store((int*)device_word, std::memory_order_release);
    return GIM_OK;
}
int* gim_range_guard_ptr() { return g_range_guard[GimPerDevice::dev()].load(std::memory_order_acquire); }
