// What does ncclCommInitAll hold while it runs?  One thread initialises a communicator on GPU 0 (5.4 s on this pool, round 5);
// the main thread meanwhile times the HIP calls the loader and the traversal make, one kind per line, with timestamps.
// hipcc -O2 -o scripts/_rccl_init_probe scripts/rccl_init_probe.cpp -lrccl -pthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(uint32_t* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main(int argc, char** argv) {
    const bool early = argc > 1 && !strcmp(argv[1], "early"); // early: the communicator comes up before the process's first other HIP call
    const double t0 = now();
    std::atomic<int> done{0};
    ncclComm_t comm;
    int dev = 0;
    std::thread th;
    auto init = [&] { double a = now(); ncclResult_t r = ncclCommInitAll(&comm, 1, &dev); printf("[%.3f] ncclCommInitAll returned %d after %.3f s\n", now() - t0, (int)r, now() - a); done = 1; };
    if (early) th = std::thread(init);
    (void)hipSetDevice(0);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    void *d = nullptr, *h = nullptr;
    const size_t bytes = 32u << 20;
    (void)hipMalloc(&d, bytes);
    (void)hipHostMalloc(&h, bytes, hipHostMallocDefault);
    memset(h, 1, bytes);
    (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st, (uint32_t*)d, bytes / 4);
    (void)hipStreamSynchronize(st);
    printf("[%.3f] first HIP work done\n", now() - t0);
    if (!early) th = std::thread(init);
    struct Stat { const char* name; double worst = 0, sum = 0; int n = 0; double worst_at = 0; } s[6] = {{"hipMemcpyAsync H2D 32 MB (enqueue)"}, {"kernel launch"}, {"hipStreamSynchronize"}, {"hipMalloc+hipFree 64 MB"}, {"hipHostMalloc+Free 8 MB"}, {"hipEventRecord+Query"}};
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    int rounds = 0;
    while (!done || rounds < 20) {
        auto timed = [&](int k, auto&& f) { double a = now(); f(); double dt = now() - a; s[k].sum += dt; ++s[k].n; if (dt > s[k].worst) { s[k].worst = dt; s[k].worst_at = a - t0; } };
        timed(0, [&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); });
        timed(1, [&] { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st, (uint32_t*)d, bytes / 4); });
        timed(2, [&] { (void)hipStreamSynchronize(st); });
        timed(3, [&] { void* q = nullptr; (void)hipMalloc(&q, 64u << 20); (void)hipFree(q); });
        timed(4, [&] { void* q = nullptr; (void)hipHostMalloc(&q, 8u << 20, hipHostMallocDefault); (void)hipHostFree(q); });
        timed(5, [&] { (void)hipEventRecord(ev, st); (void)hipEventQuery(ev); });
        ++rounds;
        if (done && rounds > 2000) break;
    }
    th.join();
    printf("[%.3f] %d rounds while the communicator came up\n", now() - t0, rounds);
    for (auto& x : s) printf("  %-36s n=%d mean %.3f ms worst %.1f ms (at %.3f s)\n", x.name, x.n, x.sum / x.n * 1e3, x.worst * 1e3, x.worst_at);
    return 0;
}
