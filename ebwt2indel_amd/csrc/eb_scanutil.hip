// eb_scanutil.hip -- exclusive prefix sum over an array of 4 x u64 vectors (letter counts per tile,
// document-array popcounts).  The arrays scanned here are tiny next to the BWT (one vector per
// 32768 symbols), so a single 1024-thread workgroup that walks the array strip by strip is enough;
// no temporary storage is needed.
#include "eb_internal.h"

__global__ void __launch_bounds__(1024) k_scan4(uint64_t* __restrict__ v, uint64_t count) {
    __shared__ uint64_t wave_tot[4][16];
    __shared__ uint64_t carry[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < count; base += 1024) {
        uint64_t i = base + threadIdx.x;
        uint64_t x[4] = {0, 0, 0, 0}, inc[4];
        if (i < count)
            for (int c = 0; c < 4; ++c) x[c] = v[i * 4 + c];
        for (int c = 0; c < 4; ++c) {
            uint64_t s = x[c];
            for (int o = 1; o < 64; o <<= 1) {
                uint64_t t = __shfl_up((unsigned long long)s, o);
                if (lane >= (uint32_t)o) s += t;
            }
            inc[c] = s;
            if (lane == 63) wave_tot[c][wave] = s;
        }
        __syncthreads();
        uint64_t out[4];
        for (int c = 0; c < 4; ++c) {
            uint64_t before = carry[c];
            for (uint32_t w = 0; w < wave; ++w) before += wave_tot[c][w];
            out[c] = before + inc[c] - x[c];
        }
        if (i < count)
            for (int c = 0; c < 4; ++c) v[i * 4 + c] = out[c];
        __syncthreads();
        if (threadIdx.x == 1023)
            for (int c = 0; c < 4; ++c) carry[c] = out[c] + x[c];
        __syncthreads();
    }
}

int eb_scan4_exclusive(uint64_t* d_vec4, uint64_t count, hipStream_t stream) {
    if (count == 0) return EB_OK;
    hipLaunchKernelGGL(k_scan4, dim3(1), dim3(1024), 0, stream, d_vec4, count);
    EB_HIP(hipGetLastError());
    return EB_OK;
}
