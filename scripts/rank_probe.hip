// Micro-benchmark (exploration, not product): what limits random 64-byte block reads on MI355X?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t counts(uint4 c, uint32_t mask) {
    uint32_t d = ~c.z & mask, lo = d & ~c.y, hi = d & c.y;
    return __popc(lo & ~c.x) | (__popc(lo & c.x) << 8) | (__popc(hi & ~c.x) << 16) | (__popc(hi & c.x) << 24);
}
__device__ __forceinline__ uint32_t pmask(uint32_t o, uint32_t j) { uint32_t w = o >> 5; return j < w ? ~0u : (j == w ? ((1u << (o & 31)) - 1) : 0u); }

// gather only: K independent queries per quad per trip, no output
template <int K>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ chunks, const uint64_t* __restrict__ pos, uint64_t m, uint64_t* __restrict__ sink) {
    const uint32_t j = threadIdx.x & 3;
    uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    uint32_t acc = 0;
    for (; q + (K - 1) * stride < m; q += K * stride) {
        uint64_t i[K]; uint4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) i[k] = pos[q + k * stride];
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = chunks[((i[k] >> 7) << 2) + j];
#pragma unroll
        for (int k = 0; k < K; ++k) acc += counts(c[k], pmask(i[k] & 127, j)) + c[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// gather only with a cache policy on the block loads (inline asm: hipcc has no builtin for sc0/sc1 on global loads):
// POLICY 1 = sc1, 2 = sc0 sc1, 3 = nt, 4 = sc0 sc1 nt.  Does a policy make the L2 ask the fabric for 64 bytes instead of a
// 128-byte line (TCC_EA0_RDREQ_64B / _128B under rocprofv3), and does the request rate change?
template <int K, int POLICY>
__global__ void __launch_bounds__(256) k_gather_pol(const uint4* __restrict__ chunks, const uint64_t* __restrict__ pos, uint64_t m, uint64_t* __restrict__ sink) {
    const uint32_t j = threadIdx.x & 3;
    uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    uint32_t acc = 0;
    for (; q + (K - 1) * stride < m; q += K * stride) {
        uint64_t i[K]; uint4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) i[k] = pos[q + k * stride];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint4* a = chunks + ((i[k] >> 7) << 2) + j;
            if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(c[k]) : "v"(a) : "memory");
            else if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(c[k]) : "v"(a) : "memory");
            else if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(c[k]) : "v"(a) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(c[k]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) acc += counts(c[k], pmask(i[k] & 127, j)) + c[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// full rank with output, K in flight
template <int K, bool NT>
__global__ void __launch_bounds__(256) k_rank(const uint4* __restrict__ chunks, const uint64_t* __restrict__ pos, uint64_t m, uint64_t* __restrict__ out4) {
    const uint32_t j = threadIdx.x & 3;
    uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    for (; q + (K - 1) * stride < m; q += K * stride) {
        uint64_t i[K]; uint4 c[K]; uint32_t p[K];
#pragma unroll
        for (int k = 0; k < K; ++k) i[k] = NT ? __builtin_nontemporal_load(&pos[q + k * stride]) : pos[q + k * stride];
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = chunks[((i[k] >> 7) << 2) + j];
#pragma unroll
        for (int k = 0; k < K; ++k) { p[k] = counts(c[k], pmask(i[k] & 127, j)); p[k] += __shfl_xor((int)p[k], 1); p[k] += __shfl_xor((int)p[k], 2); }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint64_t v = (uint64_t)c[k].w + ((p[k] >> (8 * j)) & 0xFF);
            if (NT) __builtin_nontemporal_store(v, &out4[(q + k * stride) * 4 + j]); else out4[(q + k * stride) * 4 + j] = v;
        }
    }
}
// contiguous queries per quad group: quad handles queries q*K .. q*K+K-1 (better output coalescing per wave? same)
__global__ void __launch_bounds__(256) k_write_only(uint64_t m, uint64_t* __restrict__ out4) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; t < m * 4; t += stride) out4[t] = t;
}

template <typename F> float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    uint64_t nblocks = (argc > 1 ? atoll(argv[1]) : 4096) * (1ull << 20) / 64; // table MB
    uint64_t m = 1ull << 27;
    uint4* chunks; uint64_t *pos, *out, *sink;
    CK(hipMalloc(&chunks, nblocks * 64)); CK(hipMemset(chunks, 0x5a, nblocks * 64));
    CK(hipMalloc(&pos, m * 8)); CK(hipMalloc(&out, m * 32)); CK(hipMalloc(&sink, 8));
    std::vector<uint64_t> h(m); uint64_t s = 88172645463325252ull;
    for (uint64_t k = 0; k < m; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[k] = s % (nblocks * 128); }
    CK(hipMemcpy(pos, h.data(), m * 8, hipMemcpyHostToDevice));
    printf("table %.2f GB, %llu queries\n", nblocks * 64 / 1e9, (unsigned long long)m);
    const bool sweep = argc > 2; // table-size sweep (second argument): one grid, one line
    if (sweep) {
        const int g = 256 * 16;
        float a4 = time_ms([&] { hipLaunchKernelGGL(k_gather<4>, dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float r4 = time_ms([&] { hipLaunchKernelGGL((k_rank<4, false>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        printf("table %8.1f MB: gather-only K=4 %.1f G reads/s, rank+output K=4 %.1f GQ/s\n", nblocks * 64 / 1048576.0, m / a4 / 1e6, m / r4 / 1e6);
        return 0;
    }
    int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        float a1 = time_ms([&] { hipLaunchKernelGGL(k_gather<1>, dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float a2 = time_ms([&] { hipLaunchKernelGGL(k_gather<2>, dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float a4 = time_ms([&] { hipLaunchKernelGGL(k_gather<4>, dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float a8 = time_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        printf("grid %5d gather-only  K=1 %.1f  K=2 %.1f  K=4 %.1f  K=8 %.1f GQ/s\n", g, m / a1 / 1e6, m / a2 / 1e6, m / a4 / 1e6, m / a8 / 1e6);
        float r2 = time_ms([&] { hipLaunchKernelGGL((k_rank<2, false>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        float r4 = time_ms([&] { hipLaunchKernelGGL((k_rank<4, false>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        float r8 = time_ms([&] { hipLaunchKernelGGL((k_rank<8, false>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        float n4 = time_ms([&] { hipLaunchKernelGGL((k_rank<4, true>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        float n8 = time_ms([&] { hipLaunchKernelGGL((k_rank<8, true>), dim3(g), dim3(256), 0, 0, chunks, pos, m, out); });
        printf("grid %5d rank+output  K=2 %.1f  K=4 %.1f  K=8 %.1f  nt K=4 %.1f  nt K=8 %.1f GQ/s\n", g, m / r2 / 1e6, m / r4 / 1e6, m / r8 / 1e6, m / n4 / 1e6, m / n8 / 1e6);
    }
    {
        const int g = 256 * 16;
        float p1 = time_ms([&] { hipLaunchKernelGGL((k_gather_pol<4, 1>), dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float p2 = time_ms([&] { hipLaunchKernelGGL((k_gather_pol<4, 2>), dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float p3 = time_ms([&] { hipLaunchKernelGGL((k_gather_pol<4, 3>), dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        float p4 = time_ms([&] { hipLaunchKernelGGL((k_gather_pol<4, 4>), dim3(g), dim3(256), 0, 0, chunks, pos, m, sink); });
        printf("grid %5d gather-only K=4 by cache policy: sc1 %.1f  sc0 sc1 %.1f  nt %.1f  sc0 sc1 nt %.1f GQ/s\n", g, m / p1 / 1e6, m / p2 / 1e6, m / p3 / 1e6, m / p4 / 1e6);
    }
    float w = time_ms([&] { hipLaunchKernelGGL(k_write_only, dim3(256 * 16), dim3(256), 0, 0, m, out); });
    printf("write-only 32 B/query: %.1f GQ/s (%.2f TB/s)\n", m / w / 1e6, m * 32 / w / 1e9);
    return 0;
}
