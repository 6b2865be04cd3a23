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

// ---------------------------------------------------------------------------------------------------------------------------
// deferred parameter-gradient reductions (see common.h)
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int DEF_MAX = 24;      // descriptors per batched launch (the table travels in the kernel arguments: 24 x 64 B)

struct DefDesc {
    const float* partials;
    float* out0;
    float* out1;
    int64_t pstride, n, n0;
    int P;
    float scale;
    int block0;      // first workgroup of this descriptor in the batched grid
    int pad;
};

struct DefBatch {
    DefDesc d[DEF_MAX];
    int count, blocks;
};

struct DefState {
    bool on = false;
    char* arena = nullptr;
    size_t cap = 0, used = 0;
    const char* glo = nullptr;
    const char* ghi = nullptr;
    hipStream_t stream = nullptr;
    DefBatch batch{};
} g_def;

// same arithmetic as reduce_rows_kernel (16 columns x 16 row lanes, fixed order), one descriptor per run of workgroups
__global__ __launch_bounds__(256) void reduce_rows_batched_kernel(DefBatch b) {
    __shared__ float red[16][17];
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.d[i + 1].block0) ++i;      // (uniform: the table sits in scalar registers)
    const DefDesc& q = b.d[i];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)((int)blockIdx.x - q.block0) * 16 + tx;
    float s = 0.f;
    if (j < q.n) {
        int p = ty;
        for (; p + 48 < q.P; p += 64) {
            const float a = q.partials[(int64_t)p * q.pstride + j], c = q.partials[(int64_t)(p + 16) * q.pstride + j];
            const float e = q.partials[(int64_t)(p + 32) * q.pstride + j], f = q.partials[(int64_t)(p + 48) * q.pstride + j];
            s += (a + c) + (e + f);
        }
        for (; p < q.P; p += 16) s += q.partials[(int64_t)p * q.pstride + j];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < q.n) {
        float t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][tx];
        t *= q.scale;
        float* dst = j < q.n0 ? q.out0 + j : (q.out1 ? q.out1 + (j - q.n0) : nullptr);
        if (dst) *dst = t + *dst;      // (only accumulating reductions are deferred)
    }
}

int def_flush(hipStream_t stream) {
    if (g_def.batch.count == 0) {
        g_def.used = 0;
        return ISEG_OK;
    }
    hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3(g_def.batch.blocks), dim3(256), 0, stream, g_def.batch);
    g_def.batch.count = 0;
    g_def.batch.blocks = 0;
    g_def.used = 0;      // later producers run behind this launch in stream order: the arena is free again
    return iseg_check_launch("iseg_deferred_flush");
}

}  // namespace

float* iseg_deferred_partials(size_t bytes, const float* out0, const float* out1, int accumulate, hipStream_t stream) {
    if (!g_def.on || !accumulate || stream != g_def.stream || !out0) return nullptr;
    const char* a = (const char*)out0;
    if (a < g_def.glo || a >= g_def.ghi) return nullptr;
    if (out1 && ((const char*)out1 < g_def.glo || (const char*)out1 >= g_def.ghi)) return nullptr;
    bytes = (bytes + 255) / 256 * 256;
    if (bytes > g_def.cap) return nullptr;
    for (int i = 0; i < g_def.batch.count; ++i) {      // two queued reductions into the same gradient would race inside one launch
        const DefDesc& q = g_def.batch.d[i];
        if (q.out0 == out0 || (out1 && q.out1 == out1) || (q.out1 && q.out1 == out0) || (out1 && q.out0 == out1)) {
            def_flush(stream);
            break;
        }
    }
    if (g_def.used + bytes > g_def.cap || g_def.batch.count == DEF_MAX) def_flush(stream);
    float* p = (float*)(g_def.arena + g_def.used);
    g_def.used += bytes;
    return p;
}

void iseg_deferred_push(const float* partials, int P, int64_t pstride, int64_t n, float* out0, float* out1, int64_t n0, float scale,
                        hipStream_t stream) {
    DefBatch& b = g_def.batch;
    if (b.count == DEF_MAX) def_flush(stream);      // (cannot happen after iseg_deferred_partials, kept for safety)
    DefDesc& q = b.d[b.count++];
    q.partials = partials;
    q.out0 = out0;
    q.out1 = out1;
    q.pstride = pstride;
    q.n = n;
    q.n0 = n0;
    q.P = P;
    q.scale = scale;
    q.block0 = b.blocks;
    q.pad = 0;
    b.blocks += (int)((n + 15) / 16);
}

extern "C" int iseg_deferred_begin(void* grad_base, size_t grad_bytes, void* arena, size_t arena_bytes, hipStream_t stream) {
    ISEG_REQUIRE(grad_base && grad_bytes > 0 && arena && arena_bytes >= (1u << 20), "iseg_deferred_begin: bad arguments");
    ISEG_REQUIRE(!g_def.on, "iseg_deferred_begin: already active");
    g_def.on = true;
    g_def.arena = (char*)arena;
    g_def.cap = arena_bytes / 256 * 256;
    g_def.used = 0;
    g_def.glo = (const char*)grad_base;
    g_def.ghi = g_def.glo + grad_bytes;
    g_def.stream = stream;
    g_def.batch.count = 0;
    g_def.batch.blocks = 0;
    return ISEG_OK;
}

extern "C" int iseg_deferred_flush(hipStream_t stream) {
    if (!g_def.on) return ISEG_OK;
    return def_flush(stream);
}

extern "C" int iseg_deferred_end(hipStream_t stream) {
    if (!g_def.on) return ISEG_OK;
    const int rc = def_flush(stream);
    g_def.on = false;
    return rc;
}
