// wave_launch.hip -- how fast can gfx950 START waves?  A grid of short-lived waves (each spins for a given number of
// clocks, touches no memory) is timed with HIP events for several workgroup sizes, register footprints (the kernel is
// compiled to hold NV vector registers), LDS footprints and wave lifetimes.  If the launch takes longer than
// waves x lifetime / wave slots, the difference is what the dispatcher costs: the per-wave start rate bounds every kernel
// whose waves live for about a microsecond (k_resize_linear_4x4: 1.0 us, k_describe: ~6 us for one keypoint).
// build: hipcc --offload-arch=gfx950 -O2 -o wave_launch wave_launch.hip      output: one JSON object (profiles/r02_wave_launch.json)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int NV, int LDS>
__global__ void k(int *sink, int spin_clocks)
{
    __shared__ int lds[LDS > 0 ? LDS / 4 : 1];
    int v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = threadIdx.x + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < spin_clocks) {
#pragma unroll
        for (int i = 0; i < NV; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v[i]));
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < NV; i++) s ^= v[i];
    if (LDS > 0) { lds[threadIdx.x % (LDS / 4)] = s; __syncthreads(); s ^= lds[(threadIdx.x + 1) % (LDS / 4)]; }
    if (s == 0x7fffffff) sink[0] = s;   // never true in practice: keeps everything alive without a store per wave
}

template <int NV, int LDS>
static double run(int *sink, int wg_threads, int nwaves, int spin, int reps)
{
    const int waves_per_wg = wg_threads / 64, nwg = nwaves / waves_per_wg;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<NV, LDS>), dim3(nwg), dim3(wg_threads), 0, 0, sink, spin);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k<NV, LDS>), dim3(nwg), dim3(wg_threads), 0, 0, sink, spin);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[r], a, b));
    }
    std::sort(ms.begin(), ms.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms[reps / 2] * 1e3;   // us
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int *sink; CK(hipMalloc(&sink, 64));
    const int nwaves = 65536, reps = 15;
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves\": %d, \"note\": \"us per launch of 65536 waves; waves_per_us = start rate when the launch is not bound by wave slots\", \"runs\": [\n", prop.name, prop.multiProcessorCount, nwaves);
    bool first = true;
    auto emit = [&](const char *what, int nv, int lds, int wg, int spin, double us) {
        printf("%s  {\"case\": \"%s\", \"vgprs\": %d, \"lds_bytes\": %d, \"wg_threads\": %d, \"spin_clocks\": %d, \"us\": %.2f, \"waves_per_us\": %.0f}", first ? "" : ",\n", what, nv, lds, wg, spin, us, nwaves / us);
        first = false;
    };
    const int spins[] = {0, 1000, 2500, 5000};
    for (int spin : spins) {
        for (int wg : {64, 256, 512, 1024}) emit("few registers, no LDS", 8, 0, wg, spin, run<8, 0>(sink, wg, nwaves, spin, reps));
        emit("48 registers", 48, 0, 256, spin, run<48, 0>(sink, 256, nwaves, spin, reps));
        emit("72 registers", 72, 0, 256, spin, run<72, 0>(sink, 256, nwaves, spin, reps));
        emit("120 registers", 120, 0, 256, spin, run<120, 0>(sink, 256, nwaves, spin, reps));
        emit("72 registers + 22 KiB LDS", 72, 22528, 256, spin, run<72, 22528>(sink, 256, nwaves, spin, reps));
    }
    // Round 3: the footprints of k_describe (64 registers, 3.5 KiB of LDS per wave) and k_fast_cells (48 registers, 5 KiB) with their
    // wave lifetimes, as one-wave and as four-wave workgroups: launch time against waves x lifetime / 8192 wave slots = what a slot
    // loses between the end of one wave and the start of the next.
    for (int spin : {6000, 14000, 25000}) {
        emit("describe footprint, 1-wave workgroups", 64, 3584, 64, spin, run<64, 3584>(sink, 64, nwaves, spin, reps));
        emit("describe footprint, 4-wave workgroups", 64, 14336, 256, spin, run<64, 14336>(sink, 256, nwaves, spin, reps));
        emit("FAST footprint, 1-wave workgroups", 48, 5120, 64, spin, run<48, 5120>(sink, 64, nwaves, spin, reps));
        emit("FAST footprint, 4-wave workgroups", 48, 20480, 256, spin, run<48, 20480>(sink, 256, nwaves, spin, reps));
    }
    printf("\n]}\n");
    return 0;
}
