// orbx_debug.hip -- calibration helpers for the rocprofv3 traffic counters (not part of the product path).
// MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by exactly 2x on
// gfx950 and other access widths are uncalibrated; tools/calibrate_traffic.py streams a known byte
// count with 4-byte and 16-byte loads and with 4-byte stores under --pmc FETCH_SIZE / WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/orbx.h"

template <typename T>
__global__ void k_dbg_read(const T *__restrict__ p, size_t n, uint32_t *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        const T v = p[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;   // keeps the loads alive
}

__global__ void k_dbg_write(uint32_t *__restrict__ p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}

// mode 0: 4-byte loads, 1: 16-byte loads, 2: 4-byte stores.  Streams `bytes` once per repetition.
extern "C" int orbx_debug_stream(size_t bytes, int mode, int reps)
{
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 256) != hipSuccess) return ORBX_E_HIP;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < reps; r++) {
        if (mode == 0) hipLaunchKernelGGL(k_dbg_read<uint32_t>, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
        else if (mode == 1) hipLaunchKernelGGL(k_dbg_read<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
        else hipLaunchKernelGGL(k_dbg_write, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4);
    }
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(buf); (void)hipFree(sink);
    return e == hipSuccess ? ORBX_OK : ORBX_E_HIP;
}
