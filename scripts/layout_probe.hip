// Micro-benchmark (exploration, not product), round 3: two questions about random reads of the packed BWT on MI355X.
//
// (1) LAYOUT A/B.  A = the tree's layout: 64-byte block of 128 symbols with its own four counters, read by FOUR
//     lanes (16 bytes each); two such blocks share a 128-byte L2 line.  B = one block per line: 128 bytes = 256 symbols, ONE set
//     of counters + 16 spare bytes, read by EIGHT lanes.  Same index size (n/2 bytes), same lines touched by any query: what
//     differs is the number of requests and lanes per query.  Patterns: uniform random single queries (the rank metric),
//     six queries within 30 rows (a suffix-tree node), a pair ~20 rows apart (an LF step of a cluster's left context).
// (2) MEMORY TYPE.  Every L2 miss of the cached table moves a whole 128-byte line over the fabric (PMC, round 2), whatever
//     cache policy the load carries.  Does an UNCACHED (hipDeviceMallocUncached) or fine-grained allocation make a 64-byte
//     quad read a 64-byte fabric request -- and serve more of them per second?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_probe scripts/layout_probe.hip && /tmp/layout_probe [table MB]
//   (PMC: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- /tmp/layout_probe 4096 pmc)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t counts(uint4 c, uint32_t mask) {
    uint32_t d = ~c.z & mask, lo = d & ~c.y, hi = d & c.y;
    return __popc(lo & ~c.x) | (__popc(lo & c.x) << 8) | (__popc(hi & ~c.x) << 16) | (__popc(hi & c.x) << 24);
}
__device__ __forceinline__ uint32_t pmask(uint32_t o, uint32_t j) { uint32_t w = o >> 5; return j < w ? ~0u : (j == w ? ((1u << (o & 31)) - 1) : 0u); }

// Q queries per lane group and trip, positions pos[g * Q + k] (a group's queries are adjacent in the position stream).
// LANES = 4: layout A (quad, 64-byte block of 128 symbols); LANES = 8: layout B (eight lanes, 128-byte block of 256 symbols).
// OUT: write the four ranks (32 bytes per query) -- else results only feed a sink.
template <int LANES, int Q, bool OUT>
__global__ void __launch_bounds__(256) k_rank(const uint4* __restrict__ chunks, const uint64_t* __restrict__ pos, uint64_t groups, uint64_t* __restrict__ out4,
                                             uint64_t* __restrict__ sink) {
    const uint32_t j = threadIdx.x & (LANES - 1);
    uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) / LANES;
    uint32_t acc = 0;
    for (; g < groups; g += stride) {
        uint64_t i[Q];
        uint4 c[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) i[k] = pos[g * Q + k];
#pragma unroll
        for (int k = 0; k < Q; ++k) c[k] = LANES == 4 ? chunks[((i[k] >> 7) << 2) + j] : chunks[((i[k] >> 8) << 3) + j];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            uint32_t p = counts(c[k], pmask((uint32_t)(i[k] & (LANES * 32 - 1)), j));
            p += __shfl_xor((int)p, 1);
            p += __shfl_xor((int)p, 2);
            if (LANES == 8) p += __shfl_xor((int)p, 4);
            // layout B: the counter of letter (j & 3) sits in chunk (j & 3); lanes 4..7 fetch it from their partner
            uint32_t w = c[k].w;
            if (LANES == 8) w = __shfl((int)w, (threadIdx.x & 63u & ~7u) + (j & 3u));
            uint64_t v = (uint64_t)w + ((p >> (8 * (j & 3))) & 0xFF);
            if (OUT) { if (j < 4) out4[(g * Q + k) * 4 + j] = v; }
            else acc += (uint32_t)v;
        }
    }
    if (!OUT && acc == 0x12345678u) sink[0] = acc;
}

template <typename F> float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const uint64_t bytes = (argc > 1 ? atoll(argv[1]) : 4096) * (1ull << 20);
    const bool pmc = argc > 2 && !strcmp(argv[2], "pmc"); // one launch of each kernel, for a counter pass
    const uint64_t n = bytes * 2;                         // symbols (n/2 bytes in either layout)
    const uint64_t m = 1ull << 27;                        // queries per pattern
    uint64_t *pos, *out, *sink;
    CK(hipMalloc(&pos, m * 8)); CK(hipMalloc(&out, m * 32)); CK(hipMalloc(&sink, 8));
    std::vector<uint64_t> h(m);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    struct Pattern { const char* name; int q; } pats[] = {{"uniform random, 1 per group", 1}, {"node: 6 within 30 rows", 6}, {"LF pair ~20 rows apart", 2}};
    struct Mem { const char* name; unsigned flag; } mems[] = {{"hipMalloc (cached)", 0}, {"hipDeviceMallocUncached", hipDeviceMallocUncached}, {"hipDeviceMallocFinegrained", hipDeviceMallocFinegrained}};
    printf("table %.2f GB (%llu symbols at 4 bits), %llu queries per pattern\n", bytes / 1e9, (unsigned long long)n, (unsigned long long)m);
    for (const Mem& mem : mems) {
        uint4* chunks = nullptr;
        hipError_t e = mem.flag ? hipExtMallocWithFlags((void**)&chunks, bytes, mem.flag) : hipMalloc((void**)&chunks, bytes);
        if (e != hipSuccess) { printf("%s: allocation failed: %s\n", mem.name, hipGetErrorString(e)); continue; }
        CK(hipMemset(chunks, 0x5a, bytes));
        CK(hipDeviceSynchronize());
        for (const Pattern& pt : pats) {
            const uint64_t groups = m / pt.q;
            const int off6[6] = {0, 5, 11, 17, 23, 30};
            for (uint64_t g = 0; g < groups; ++g) {
                uint64_t base = rnd() % (n - 64);
                for (int k = 0; k < pt.q; ++k) h[g * pt.q + k] = base + (pt.q == 6 ? off6[k] : (pt.q == 2 ? 21 * k : 0));
            }
            CK(hipMemcpy(pos, h.data(), groups * pt.q * 8, hipMemcpyHostToDevice));
            const int grid = 256 * 16;
            const uint64_t mq = groups * pt.q;
            float a = 0, b = 0, ao = 0, bo = 0;
            auto run = [&](auto kern) { return time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, chunks, pos, groups, out, sink); }, pmc ? 1 : 5); };
            if (pt.q == 1) { a = run(k_rank<4, 1, false>); b = run(k_rank<8, 1, false>); ao = run(k_rank<4, 1, true>); bo = run(k_rank<8, 1, true>); }
            if (pt.q == 6) { a = run(k_rank<4, 6, false>); b = run(k_rank<8, 6, false>); ao = run(k_rank<4, 6, true>); bo = run(k_rank<8, 6, true>); }
            if (pt.q == 2) { a = run(k_rank<4, 2, false>); b = run(k_rank<8, 2, false>); ao = run(k_rank<4, 2, true>); bo = run(k_rank<8, 2, true>); }
            printf("%-28s %-28s  A (64-B block, 4 lanes): %6.1f GQ/s, with output %6.1f   |   B (128-B block, 8 lanes): %6.1f GQ/s, with output %6.1f\n", mem.name, pt.name,
                   mq / a / 1e6, mq / ao / 1e6, mq / b / 1e6, mq / bo / 1e6);
            fflush(stdout);
        }
        CK(hipFree(chunks));
    }
    return 0;
}
