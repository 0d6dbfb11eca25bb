// Probe (exploration, not product): would a TWO-symbol LF step -- a pair-alphabet index, one 128-byte line per
// two steps of the left-context extension -- beat k_extend1's one 64-byte block (one 128-byte line) per step?
// The extension is a walk of dependent memory accesses with little arithmetic: what this measures is the memory side of both forms
// on a table of the index's size, with the arithmetic of each form's rank evaluation attached:
//   single : 30 dependent steps per lane, each one random 64-byte block (four 16-byte loads of one line) + the per-letter popcounts of
//            both range ends (what k_extend1 does)
//   pair   : 15 dependent steps per lane, each one random 128-byte line (eight 16-byte loads: b planes, a planes, 4 + 16 counters of a
//            64-symbol group) + the popcounts for "arg-max over b, then over a given b" at both range ends; with probability P2 the range
//            straddles the group and the NEXT line is read too (adjacent: same DRAM page)
// for 1.6 M lanes (the chains of a whole-range pass at 7.55 G positions) and 200 k lanes (a rank's eighth).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/pair_step_probe scripts/pair_step_probe.hip && /tmp/pair_step_probe 4096
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }
__device__ __forceinline__ uint32_t counts4(uint4 c, uint32_t mask) {
    uint32_t d = ~c.z & mask, lo = d & ~c.y, hi = d & c.y;
    return __popc(lo & ~c.x) | (__popc(lo & c.x) << 8) | (__popc(hi & ~c.x) << 16) | (__popc(hi & c.x) << 24);
}

template <int STEPS>
__global__ void __launch_bounds__(256) k_single(const uint4* __restrict__ t, uint64_t n_blocks, uint64_t lanes, uint64_t* __restrict__ sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    uint64_t s = mix(i + 1);
    uint32_t acc = 0;
    for (int k = 0; k < STEPS; ++k) {
        const uint4* b = t + (s & (n_blocks - 1)) * 4; // (tables are powers of two)
        const uint4 c0 = b[0], c1 = b[1], c2 = b[2], c3 = b[3];
        const uint32_t o = (uint32_t)(s >> 40) & 127u, o2 = (o + 13u) & 127u;
        // both ends of the range: four letters x four chunks each
        uint32_t p = counts4(c0, o > 32 ? ~0u : (1u << (o & 31)) - 1) + counts4(c1, o > 64 ? ~0u : (1u << (o & 31)) - 1) + counts4(c2, o > 96 ? ~0u : (1u << (o & 31)) - 1) + counts4(c3, (1u << (o & 31)) - 1);
        uint32_t q = counts4(c0, o2 > 32 ? ~0u : (1u << (o2 & 31)) - 1) + counts4(c1, o2 > 64 ? ~0u : (1u << (o2 & 31)) - 1) + counts4(c2, o2 > 96 ? ~0u : (1u << (o2 & 31)) - 1) + counts4(c3, (1u << (o2 & 31)) - 1);
        const uint32_t d = q - p;
        uint32_t best = d & 0xFF, bj = 0;
        for (uint32_t j = 1; j < 4; ++j) { const uint32_t v = (d >> (8 * j)) & 0xFF; if (v > best) { best = v; bj = j; } }
        acc += best;
        s = mix(s + c0.w + c1.w + c2.w + c3.w + bj + p);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// P2_256: probability (in 1/256) that a step reads the adjacent line as well
template <int STEPS>
__global__ void __launch_bounds__(256) k_pair(const uint4* __restrict__ t, uint64_t n_lines, uint64_t lanes, uint32_t p2_256, uint64_t* __restrict__ sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    uint64_t s = mix(i + 1);
    uint32_t acc = 0;
    for (int k = 0; k < STEPS; ++k) {
        const uint64_t line = s & (n_lines / 2 - 1);
        const uint4* b = t + line * 8;
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = b[j];
        uint4 w[8];
        const bool two = ((uint32_t)(s >> 48) & 255u) < p2_256; // the range's end lies in the next group
        if (two) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = b[8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = v[j];
        }
        // group of 64 symbols: v[0] = b planes x,y (2 x 64 bits), v[1].xy = b plane TERM, v[1].zw / v[2] = a planes, v[3] = 4 b counters, v[4..7] = 16 pair counters
        const uint32_t o = (uint32_t)(s >> 40) & 63u, o2 = (o + 13u) & 63u;
        const uint64_t m1 = (1ull << o) - 1, m2 = (1ull << o2) - 1;
        auto u64 = [](uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; };
        const uint64_t bx = u64(v[0].x, v[0].y), by = u64(v[0].z, v[0].w), bt = u64(v[1].x, v[1].y), ax = u64(v[1].z, v[1].w), ay = u64(v[2].x, v[2].y), at = u64(v[2].z, v[2].w);
        const uint64_t Bx = u64(w[0].x, w[0].y), By = u64(w[0].z, w[0].w), Bt = u64(w[1].x, w[1].y), Ax = u64(w[1].z, w[1].w), Ay = u64(w[2].x, w[2].y), At = u64(w[2].z, w[2].w);
        // arg-max over b: four letters at both ends (selects only: an array indexed at run time would live in scratch memory)
        auto cnt_b = [&](int c) -> uint32_t {
            const uint64_t lm = ~bt & ((c & 2) ? by : ~by) & ((c & 1) ? bx : ~bx), LM = ~Bt & ((c & 2) ? By : ~By) & ((c & 1) ? Bx : ~Bx);
            const uint32_t cl = c == 0 ? v[3].x : c == 1 ? v[3].y : c == 2 ? v[3].z : v[3].w, cr = c == 0 ? w[3].x : c == 1 ? w[3].y : c == 2 ? w[3].z : w[3].w;
            return (cr + (uint32_t)__popcll(LM & m2)) - (cl + (uint32_t)__popcll(lm & m1));
        };
        const uint32_t n0 = cnt_b(0), n1 = cnt_b(1), n2 = cnt_b(2), n3 = cnt_b(3);
        uint32_t bb = 0, nbest = n0;
        if (n1 > nbest) { nbest = n1; bb = 1; }
        if (n2 > nbest) { nbest = n2; bb = 2; }
        if (n3 > nbest) { nbest = n3; bb = 3; }
        // then over a, among the rows whose b is the winner: four pair counters of that b at both ends
        const uint64_t sel = ~bt & ((bb & 2) ? by : ~by) & ((bb & 1) ? bx : ~bx), SEL = ~Bt & ((bb & 2) ? By : ~By) & ((bb & 1) ? Bx : ~Bx);
        const uint4 pc = bb == 0 ? v[4] : bb == 1 ? v[5] : bb == 2 ? v[6] : v[7], PC = bb == 0 ? w[4] : bb == 1 ? w[5] : bb == 2 ? w[6] : w[7];
        auto cnt_a = [&](int c) -> uint32_t {
            const uint64_t lm = sel & ~at & ((c & 2) ? ay : ~ay) & ((c & 1) ? ax : ~ax), LM = SEL & ~At & ((c & 2) ? Ay : ~Ay) & ((c & 1) ? Ax : ~Ax);
            const uint32_t cl = c == 0 ? pc.x : c == 1 ? pc.y : c == 2 ? pc.z : pc.w, cr = c == 0 ? PC.x : c == 1 ? PC.y : c == 2 ? PC.z : PC.w;
            return (cr + (uint32_t)__popcll(LM & m2)) - (cl + (uint32_t)__popcll(lm & m1));
        };
        const uint32_t a0 = cnt_a(0), a1 = cnt_a(1), a2 = cnt_a(2), a3 = cnt_a(3);
        uint32_t ba = 0, nabest = a0;
        if (a1 > nabest) { nabest = a1; ba = 1; }
        if (a2 > nabest) { nabest = a2; ba = 2; }
        if (a3 > nabest) { nabest = a3; ba = 3; }
        acc += nbest + nabest;
        s = mix(s + nbest + nabest + bb * 4 + ba);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const uint64_t mb = argc > 1 ? atoll(argv[1]) : 4096; // a power of two
    const uint64_t bytes = mb << 20;
    uint4* t; uint64_t* sink;
    CK(hipMalloc(&t, bytes + 4096)); CK(hipMemset(t, 0x5a, bytes + 4096)); CK(hipMalloc(&sink, 8));
    printf("table %.2f GB (single: %llu blocks of 64 B; pair: %llu lines of 128 B, i.e. the pair index of the same symbols is 4x the block table)\n", bytes / 1e9,
           (unsigned long long)(bytes / 64), (unsigned long long)(bytes / 128));
    for (uint64_t lanes : {1600000ull, 200000ull}) {
        const unsigned grid = (unsigned)((lanes + 255) / 256);
        const float s30 = time_ms([&] { hipLaunchKernelGGL(k_single<30>, dim3(grid), dim3(256), 0, 0, t, bytes / 64, lanes, sink); });
        printf("%8llu lanes: single step x30 %.3f ms (%.1f G lines/s)", (unsigned long long)lanes, s30, lanes * 30 / s30 / 1e6);
        for (uint32_t p2 : {0u, 51u, 115u}) { // 0 %, 20 %, 45 % of the steps read the adjacent line too
            const float p15 = time_ms([&] { hipLaunchKernelGGL(k_pair<15>, dim3(grid), dim3(256), 0, 0, t, bytes / 128, lanes, p2, sink); });
            printf(" | pair step x15, %2u %% two-line steps: %.3f ms (%.1f G lines/s)", p2 * 100 / 256, p15, lanes * 15 * (1 + p2 / 256.0) / p15 / 1e6);
        }
        printf("\n");
    }
    return 0;
}
