// Probe: what a fresh process pays for ONE hipMalloc of an index's size, and for the same memory reserved as address space and
// mapped slice by slice (hipMemAddressReserve + hipMemCreate / hipMemMap / hipMemSetAccess) -- could the loader map the index while
// the first pieces of the file are already on their way?     hipcc -O2 -o vmm_probe scripts/vmm_probe.cpp ; ./vmm_probe malloc|vmm [GB] [slices]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(uint4* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(1, 2, 3, 4); }
int main(int argc, char** argv) {
    const bool vmm = argc > 1 && !strcmp(argv[1], "vmm");
    const double gb = argc > 2 ? atof(argv[2]) : 3.8;
    const int slices = argc > 3 ? atoi(argv[3]) : 8;
    CK(hipFree(nullptr)); // runtime up
    size_t bytes = (size_t)(gb * (1ull << 30));
    double t0 = now();
    if (!vmm) {
        void* p = nullptr;
        CK(hipMalloc(&p, bytes));
        double t1 = now();
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        CK(hipDeviceSynchronize());
        printf("hipMalloc %.2f GB: %.1f ms; first touch of all of it by a kernel %.1f ms\n", gb, (t1 - t0) * 1e3, (now() - t1) * 1e3);
        return 0;
    }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t slice = (bytes / slices + gran - 1) / gran * gran;
    bytes = slice * slices;
    void* base = nullptr;
    CK(hipMemAddressReserve(&base, bytes, 0, nullptr, 0));
    double t1 = now();
    printf("granularity %zu, reserve of %.2f GB: %.2f ms; slices of %.0f MB:", gran, bytes / 1e9, (t1 - t0) * 1e3, slice / 1e6);
    hipMemAccessDesc acc;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs(slices);
    for (int s = 0; s < slices; ++s) {
        double a = now();
        CK(hipMemCreate(&hs[s], slice, &prop, 0));
        CK(hipMemMap((char*)base + s * slice, slice, 0, hs[s], 0));
        CK(hipMemSetAccess((char*)base + s * slice, slice, &acc, 1));
        double b = now();
        // a kernel writes the slice just mapped while the rest of the range is still unmapped
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(slice / 16 / 256)), dim3(256), 0, 0, (uint4*)((char*)base + s * slice), slice / 16);
        printf(" %.1f", (b - a) * 1e3);
    }
    CK(hipDeviceSynchronize());
    printf(" ms; all mapped and written after %.1f ms\n", (now() - t0) * 1e3);
    for (int s = 0; s < slices; ++s) { CK(hipMemUnmap((char*)base + s * slice, slice)); CK(hipMemRelease(hs[s])); }
    CK(hipMemAddressFree(base, bytes));
    return 0;
}
