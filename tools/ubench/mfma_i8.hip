// mfma_i8.hip -- issue rate of the i8 matrix instructions the dense Hamming matcher uses (gfx950): clocks per
// v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 on one SIMD, with 1 or 2 independent accumulators per wave and 1, 2, 3 waves
// per SIMD, alone and with the selection's VALU work (v_max_i32 + v_med3_i32 per accumulator element) beside it.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_i8 mfma_i8.hip      output: one JSON object (profiles/r02_mfma_i8.json)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4acc __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 32x32x32 one accumulator, 1: 32x32x32 two accumulators, 2: 16x16x64 four accumulators, 3: mode 1 + selection VALU
__global__ __launch_bounds__(256) void k(unsigned long long *clk, int *sink, int iters)
{
    v4i a = {(int)threadIdx.x, 0x40404040, (int)0xC0C0C0C0, 7}, b = {3, (int)0xC040C040, 5, (int)threadIdx.x};
    v16i c0 = {}, c1 = {};
    v4acc d0 = {}, d1 = {}, d2 = {}, d3 = {};
    int ba = -(1 << 30), sa = -(1 << 30);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        } else if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0); }
            if (MODE == 3) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int x = c0[r], y = c1[r];
                    sa = max(min(ba, sa), min(max(ba, sa), x)); ba = max(ba, x);
                    sa = max(min(ba, sa), min(max(ba, sa), y)); ba = max(ba, y);
                }
                c0[0] = ba & 1; c1[0] = sa & 1;
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, d3, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    int s = ba ^ sa;
    for (int r = 0; r < 16; r++) s ^= c0[r] ^ c1[r];
    for (int r = 0; r < 4; r++) s ^= d0[r] ^ d1[r] ^ d2[r] ^ d3[r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, iters = 2000;
    unsigned long long *d_clk; int *d_sink;
    CK(hipMalloc((void **)&d_clk, 8 * ncu * 4 * 4)); CK(hipMalloc((void **)&d_sink, 4 * ncu * 4 * 256));
    std::vector<unsigned long long> h(ncu * 4 * 4);
    const char *names[4] = {"32x32x32_i8, one accumulator chain", "32x32x32_i8, two accumulators", "16x16x64_i8, four accumulators",
                            "32x32x32_i8 two accumulators + 64 selection VALU per 16 MFMA"};
    printf("{\"device\": \"%s\", \"note\": \"clk = shader clocks per MFMA on one SIMD = loop clocks of a wave / (16 MFMA per iteration * iterations * waves per SIMD)\", \"results\": {\n", prop.gcnArchName);
    for (int mode = 0; mode < 4; mode++) {
        printf("%s  \"%s\": {", mode ? ",\n" : "", names[mode]);
        for (int wps = 1; wps <= 3; wps++) {
            const int blocks = ncu * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_clk, d_sink, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_clk, d_sink, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_clk, d_sink, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d_clk, d_sink, iters);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_clk, 8 * blocks * 4, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + blocks * 4);
            const double med = (double)h[blocks * 2];
            printf("%s\"w%d\": {\"clk\": %.2f, \"ms\": %.4f, \"TOPS\": %.1f}", wps > 1 ? ", " : "", wps, med / (16.0 * iters * wps), ms,
                   (mode == 2 ? 16.0 * 16 * 64 * 2 : 32.0 * 32 * 32 * 2) * 16.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12);
        }
        printf("}");
    }
    printf("\n}}\n");
    return 0;
}
