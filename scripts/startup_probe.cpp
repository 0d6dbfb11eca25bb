// Probe: what a process pays for the HIP runtime alone, and for having librccl.so (570 MB of code objects) among its libraries.
//   hipcc -O2 -o a scripts/startup_probe.cpp ; hipcc -O2 -o b scripts/startup_probe.cpp -Wl,--no-as-needed -lrccl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
int main() {
    double t0 = now();
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 1 << 20);
    double t1 = now();
    (void)hipFree(p);
    printf("first hipMalloc %.3f s (%s)\n", t1 - t0, hipGetErrorString(e));
    return 0;
}
