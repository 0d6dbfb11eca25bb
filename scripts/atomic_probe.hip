// atomic_probe.hip -- returning atomicAdd by one thread per workgroup per iteration (the append pattern of the traversal
// kernels): how does the time depend on the number of distinct counters the workgroups are spread over?
// build: hipcc --offload-arch=gfx950 -O2 scripts/atomic_probe.hip -o scripts/_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(unsigned long long* ctr, int n_ctr, int stride_words, int iters, unsigned long long* sink) {
    __shared__ unsigned long long base;
    unsigned long long acc = 0;
    unsigned long long* c = ctr + (size_t)(blockIdx.x % n_ctr) * stride_words;
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) base = atomicAdd(c, 1536ull);
        __syncthreads();
        acc += base;
    }
    if (acc == 12345) sink[0] = acc;
}
int main() {
    unsigned long long *ctr, *sink;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 768, iters = 2000; // 1.5 M atomics
    for (int n_ctr : {1, 2, 4, 8, 16, 64}) {
        for (int stride : {16, 512}) { // 128 B or 4 KB apart
            hipMemset(ctr, 0, 1 << 20);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, n_ctr, stride, 10, sink);
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, n_ctr, stride, iters, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("counters=%2d stride=%4d B: %.3f ms for %d atomics -> %.1f ns each (serialised)\n", n_ctr, stride * 8, ms, blocks * iters, ms * 1e6 / (blocks * iters));
        }
    }
    return 0;
}
