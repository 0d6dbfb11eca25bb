// alloc_probe.hip -- how long do hipMalloc / hipFree / hipMallocAsync of large buffers take on this box?
// build: hipcc --offload-arch=gfx950 -O2 scripts/alloc_probe.hip -o scripts/_alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    const double gbs[] = {0.25, 1, 4, 14, 14, 14, 32, 32};
    for (double gb : gbs) {
        size_t bytes = (size_t)(gb * (1ull << 30));
        void* p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, bytes));
        double t1 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        CK(hipDeviceSynchronize());
        double t2 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        CK(hipDeviceSynchronize());
        double t3 = now();
        CK(hipFree(p));
        double t4 = now();
        printf("hipMalloc %6.2f GB: malloc %.1f ms, first touch %.1f ms, second touch %.1f ms, free %.1f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
    // stream-ordered pool: release threshold = keep everything
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (int rep = 0; rep < 3; ++rep) {
        size_t bytes = 14ull << 30;
        void* p = nullptr;
        double t0 = now();
        CK(hipMallocAsync(&p, bytes, 0));
        CK(hipStreamSynchronize(0));
        double t1 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFreeAsync(p, 0));
        CK(hipStreamSynchronize(0));
        double t3 = now();
        printf("hipMallocAsync 14 GB rep %d: malloc %.1f ms, touch %.1f ms, free %.1f ms\n", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    // many medium pieces instead of one large
    {
        void* ps[56];
        double t0 = now();
        for (int i = 0; i < 56; ++i) CK(hipMalloc(&ps[i], 256ull << 20));
        double t1 = now();
        for (int i = 0; i < 56; ++i) CK(hipFree(ps[i]));
        double t2 = now();
        printf("56 x 256 MB: malloc %.1f ms, free %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    }
    return 0;
}
