// Throughput of ds_read_b128 for the access shapes of the depthwise weight-gradient MFMA kernel (csrc/dwconv_wgrad_mfma.hip):
//   mode 0  lane * 16 bytes                 (contiguous, the conflict-free reference)
//   mode 1  A fragment: 2 * (lane & 15) + 16 * (lane >> 4)   -- sixteen overlapping, 2-byte-shifted windows per k-group
//   mode 2  B fragment: (lane & 15) * ROW + 16 * (lane >> 4) with ROW = 272 bytes (136-element rows)
//   mode 3  B fragment with ROW = 256 bytes (unpadded 128-element rows)
// 64 reads in flight per wave, 4 waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void probe(uint32_t* out, int mode, int iters, uint64_t* cycles) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t off;
    if (mode == 0) off = lane * 16;
    else if (mode == 1) off = 2 * (lane & 15) + 16 * (lane >> 4);
    else if (mode == 2) off = (lane & 15) * 272 + 16 * (lane >> 4);
    else off = (lane & 15) * 256 + 16 * (lane >> 4);
    const uint32_t base = (uint32_t)(uintptr_t)lds + off + wid * 8192;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(base), "n"(q * 64) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q][0] ^ v[q][3];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    uint32_t* d_out;
    uint64_t* d_cyc;
    hipMalloc(&d_out, 256 * 256 * 4);
    hipMalloc(&d_cyc, 8);
    const char* names[4] = {"contiguous 16 B per lane", "A: 2-byte shifted windows", "B: rows of 272 B", "B: rows of 256 B"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d_out, mode, 1024, d_cyc);
        hipDeviceSynchronize();
        uint64_t cyc;
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.1f cycles per ds_read_b128 per wave with 4 waves on the CU\n", mode, names[mode], (double)cyc / (1024.0 * 16));
    }
    return 0;
}
