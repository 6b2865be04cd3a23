// Error plumbing and version for libiseg_hip.so.  Entry points never throw and never abort: they return a
// negative status and leave a per-thread message for iseg_last_error().
#include "common.h"
#include "iseg_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = {0};

extern "C" void iseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int iseg_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        iseg_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return ISEG_ERR_HIP;
    }
    return ISEG_OK;
}

extern "C" int iseg_version(void) { return 100; }

extern "C" size_t iseg_last_error(char* buf_h, size_t n) {
    const size_t len = strlen(g_err);
    if (buf_h && n > 0) {
        const size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf_h, g_err, c);
        buf_h[c] = 0;
    }
    return len;
}
