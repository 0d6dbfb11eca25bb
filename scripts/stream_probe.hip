// Micro-benchmark (exploration): streaming-read rate of the k_scan access patterns on MI355X.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// S1: current k_scan pattern: per lane 2 x u64 (two arrays) + 2 x uint4 at 32-byte lane stride (64 positions per lane)
__global__ void __launch_bounds__(256) s1(const uint64_t* __restrict__ K, const uint64_t* __restrict__ M, const uint4* __restrict__ B, uint64_t words, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t tile = blockIdx.x; tile * 256 < words; tile += gridDim.x) {
        uint64_t w = tile * 256 + threadIdx.x;
        if (w < words) { uint64_t k = K[w], m = M[w]; uint4 a = B[2 * w], b = B[2 * w + 1]; acc += (k & ~m) + a.x + a.y + b.x + b.y; }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// S2: 128 positions per lane: 16 B + 16 B + 64 B contiguous per lane
__global__ void __launch_bounds__(256) s2(const uint4* __restrict__ K, const uint4* __restrict__ M, const uint4* __restrict__ B, uint64_t words, uint64_t* sink) {
    uint64_t acc = 0; uint64_t items = words / 2;
    for (uint64_t tile = blockIdx.x; tile * 256 < items; tile += gridDim.x) {
        uint64_t i = tile * 256 + threadIdx.x;
        if (i < items) { uint4 k = K[i], m = M[i]; uint4 a = B[4 * i], b = B[4 * i + 1], c = B[4 * i + 2], d = B[4 * i + 3]; acc += k.x + m.y + a.x + b.y + c.x + d.y + k.z + m.w; }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// S3: everything as unit-stride dwordx4 (what an LDS-staged tile load would issue)
__global__ void __launch_bounds__(256) s3(const uint4* __restrict__ K, const uint4* __restrict__ M, const uint4* __restrict__ B, uint64_t words, uint64_t* sink) {
    uint64_t acc = 0; uint64_t kq = words / 2, bq = words * 2; // uint4 counts
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < bq; i += stride) {
        uint4 a = B[i]; acc += a.x + a.y;
        if (i < kq) { uint4 k = K[i], m = M[i]; acc += k.x + m.y; }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// S4: tile-wise unit stride: per tile of 256 words: K 2 KB, M 2 KB, B 8 KB loaded by the block with dwordx4 (LDS-staging order)
__global__ void __launch_bounds__(256) s4(const uint4* __restrict__ K, const uint4* __restrict__ M, const uint4* __restrict__ B, uint64_t words, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t tile = blockIdx.x; tile * 256 < words; tile += gridDim.x) {
        uint64_t kb = tile * 128, bb = tile * 512;
        uint4 a = B[bb + threadIdx.x], b = B[bb + 256 + threadIdx.x];
        uint4 k = threadIdx.x < 128 ? K[kb + threadIdx.x] : M[kb + threadIdx.x - 128];
        acc += a.x + b.y + k.x;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
template <typename F> float time_ms(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char** argv) {
    uint64_t n = (argc > 1 ? atoll(argv[1]) : 7550) * 1000000ull; uint64_t words = n / 64 / 256 * 256;
    void *K, *M, *B; uint64_t* sink;
    CK(hipMalloc(&K, words * 8)); CK(hipMalloc(&M, words * 8)); CK(hipMalloc(&B, words * 32)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(K, 1, words * 8)); CK(hipMemset(M, 2, words * 8)); CK(hipMemset(B, 3, words * 32));
    double bytes = words * 48.0;
    printf("n=%.2f G positions, %.2f GB per pass\n", words * 64 / 1e9, bytes / 1e9);
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL(s1, dim3(g), dim3(256), 0, 0, (uint64_t*)K, (uint64_t*)M, (uint4*)B, words, sink); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(s2, dim3(g), dim3(256), 0, 0, (uint4*)K, (uint4*)M, (uint4*)B, words, sink); });
        float t3 = time_ms([&] { hipLaunchKernelGGL(s3, dim3(g), dim3(256), 0, 0, (uint4*)K, (uint4*)M, (uint4*)B, words, sink); });
        float t4 = time_ms([&] { hipLaunchKernelGGL(s4, dim3(g), dim3(256), 0, 0, (uint4*)K, (uint4*)M, (uint4*)B, words, sink); });
        printf("grid %5d  S1 current %.2f ms %.2f TB/s | S2 128/lane %.2f ms %.2f TB/s | S3 unit-stride %.2f ms %.2f TB/s | S4 tile unit-stride %.2f ms %.2f TB/s\n",
               g, t1, bytes / t1 / 1e9, t2, bytes / t2 / 1e9, t3, bytes / t3 / 1e9, t4, bytes / t4 / 1e9);
    }
    return 0;
}
