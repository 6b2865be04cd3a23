// build: hipcc --offload-arch=gfx950 -O3 -w tools/micro/fill_rate.hip -o tools/micro/fill_rate   (the binary is git-ignored)
// What one CU can pull from an L2-resident footprint, by path: global_load_lds_dwordx4 (HBM/L2 -> LDS, no registers: the GEMM rings), plain
// global_load_dwordx4 into registers, and both at once.  One 512-thread workgroup per CU re-reads its own 48 KB window (the K-step of a 256 x 128
// tile) `iters` times with 12 KB per wavefront in flight; the windows of an XCD's 32 CUs are 1.5 MB together (L2 = 4 MB per XCD).  A second
// footprint of 64 MB per XCD (windows walk through it) gives the beyond-L2 rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ buf, size_t window_stride, size_t walk, int iters, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * window_stride;
    uint4 acc = {0, 0, 0, 0};
    size_t off = 0;
    for (int it = 0; it < iters; ++it) {
        const char* src = base + off + wid * 6144 + lane * 16;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int p = 0; p < (MODE == 2 ? 3 : 6); ++p)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 1024), (lds_ptr)(smem + (it & 1) * 49152 + wid * 6144 + p * 1024), 16, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            uint4 v[6];
#pragma unroll
            for (int p = (MODE == 2 ? 3 : 0); p < 6; ++p) v[p] = *reinterpret_cast<const uint4*>(src + p * 1024);
#pragma unroll
            for (int p = (MODE == 2 ? 3 : 0); p < 6; ++p) acc.x ^= v[p].x, acc.y ^= v[p].y, acc.z ^= v[p].z, acc.w ^= v[p].w;
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // one iteration stays in flight
        off += walk;
        if (off + 49152 > window_stride) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = smem[threadIdx.x];
}

template <int MODE> void run(const char* name, const char* buf, size_t window_stride, size_t walk) {
    float* out;
    hipMalloc(&out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // (the register path holds no LDS: two workgroups per CU give it the same 12 loads per wavefront slot in flight as the ring has)
    const int grid = MODE == 1 ? 512 : 256, lds = MODE == 1 ? 1024 : 98304;
    k<MODE><<<grid, 512, lds>>>(buf, MODE == 1 ? window_stride / 2 : window_stride, walk, 50, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 512, lds>>>(buf, MODE == 1 ? window_stride / 2 : window_stride, walk, MODE == 1 ? iters / 2 : iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * 49152;
    printf("%-46s %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9));
    hipFree(out);
}

int main() {
    char* buf;
    const size_t big = (size_t)256 * (2u << 20);      // 2 MB per workgroup: 512 MB
    hipMalloc(&buf, big);
    hipMemset(buf, 1, big);
    printf("L2-resident windows (48 KB per CU, re-read):\n");
    run<0>("global_load_lds_dwordx4", buf, 49152, 0);
    run<1>("global_load_dwordx4 -> registers", buf, 49152, 0);
    run<2>("half by each path", buf, 49152, 0);
    printf("streaming (each workgroup walks 2 MB, 512 MB in all):\n");
    run<0>("global_load_lds_dwordx4", buf, 2u << 20, 49152);
    run<1>("global_load_dwordx4 -> registers", buf, 2u << 20, 49152);
    run<2>("half by each path", buf, 2u << 20, 49152);
    return 0;
}
