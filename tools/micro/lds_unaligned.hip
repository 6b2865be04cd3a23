// Does ds_read_b128 take a 2-byte-aligned LDS address on gfx950 (ROCm 7.2), and at what price?  (Needed for sliding 8-element windows
// over a channel-planar row: the depthwise weight gradient as MFMA products, csrc/dwconv_wgrad_mfma.hip.)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_unaligned tools/micro/lds_unaligned.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void probe(uint32_t* out, int shift_bytes, int iters, uint64_t* cycles) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192 + 64];
    for (int i = threadIdx.x; i < 8192 + 64; i += 256) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (uint32_t)(lane * 32 + shift_bytes);      // 32-byte stride per lane, + shift
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v;
        const uint32_t a = base + (uint32_t)((it & 3) * 2048);
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        acc0 += v[0];
        acc1 += v[1];
        acc2 += v[2];
        acc3 += v[3];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    // last iteration's values for the correctness check (it = iters - 1)
    u32x4 w;
    const uint32_t a = base;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a) : "memory");
    if (blockIdx.x == 0) {
        out[threadIdx.x * 4 + 0] = w[0];
        out[threadIdx.x * 4 + 1] = w[1];
        out[threadIdx.x * 4 + 2] = w[2];
        out[threadIdx.x * 4 + 3] = w[3];
    }
    if (acc0 + acc1 + acc2 + acc3 == 0x12345678u) out[0] = 0;
}

int main() {
    uint32_t* d_out;
    uint64_t* d_cyc;
    hipMalloc(&d_out, 256 * 4 * 4);
    hipMalloc(&d_cyc, 8);
    for (int shift = 0; shift <= 14; shift += 2) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d_out, shift, 4096, d_cyc);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("shift %d: launch failed (%s)\n", shift, hipGetErrorString(hipGetLastError()));
            return 1;
        }
        std::vector<uint32_t> h(1024);
        uint64_t cyc;
        hipMemcpy(h.data(), d_out, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int e0 = (lane * 32 + shift) / 2;      // first uint16 element the lane should see
            for (int j = 0; j < 4; ++j) {
                const uint32_t want = (uint32_t)(e0 + 2 * j) | ((uint32_t)(e0 + 2 * j + 1) << 16);
                if (h[lane * 4 + j] != want) ++bad;
            }
        }
        printf("shift %2d bytes: %s, %.1f cycles per dependent ds_read_b128 (wave 0)\n", shift, bad ? "WRONG VALUES" : "values correct", (double)cyc / 4096);
    }
    return 0;
}
