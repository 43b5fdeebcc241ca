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

extern "C" int gim_version(void) { return 113; }   // 110: health word as an argument (no gim_set_range_guard), count[2 + N] layout of gim_coarse_match; 111: gim_token_emit.kv_* (fused KV state), gim_linear_attention_finalize; 112: gim_coarse_args.precand_per_row, the library reads no environment variable; 113: gim_conv_args.split16
extern "C" const char* gim_last_error(void) { return g_err; }
extern "C" int gim_ktile_bytes(void) { return 128; }
extern "C" int gim_npad_granule(void) { return 64; }
