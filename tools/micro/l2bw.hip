// build: hipcc --offload-arch=gfx950 -O3 tools/micro/l2bw.hip -o tools/micro/l2bw   (the binary is git-ignored)
// Read-bandwidth ceiling by footprint (L2 / Infinity Cache / HBM): every workgroup streams the same `bytes`-sized buffer with 16-B
// loads, starting at a different offset.  Used to decide whether the GEMM's tile re-reads (L2 -> CU traffic) are what bound it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n16, int iters, uint4* out) {
    uint4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        size_t i = (i0 + (size_t)it * 977 * 256) % n16;
        for (size_t k = 0; k < 8; ++k) {
            uint4 v = p[i];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            i += stride;
            if (i >= n16) i -= n16;
        }
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t maxb = (size_t)2 << 30;
    uint4 *buf, *out;
    hipMalloc(&buf, maxb);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, maxb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t sizes[] = {1u << 20, 4u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20, 512u << 20, (size_t)2 << 30};
    for (size_t b : sizes) {
        const size_t n16 = b / 16;
        const int grid = 256 * 8, iters = 64;
        rd<<<grid, 256>>>(buf, n16, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rd<<<grid, 256>>>(buf, n16, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * 256 * iters * 8 * 16;
        printf("footprint %6zu MiB: %8.1f GB/s\n", b >> 20, bytes / ms / 1e6);
    }
    return 0;
}
