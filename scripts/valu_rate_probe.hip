// Issue rate of the vector instructions k_scan's bit-parallel part is made of, on gfx950: 8 independent chains per lane, 2048 x 256
// lanes, straight-line blocks of 64 instructions.  hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(NAME, DECL, ASM8, SINK)                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint64_t* out, int iters, uint32_t seed) {             \
        DECL                                                                                           \
        for (int it = 0; it < iters; ++it) {                                                           \
            ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8                                                    \
        }                                                                                              \
        SINK                                                                                           \
    }

#define DECL32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
#define SINK32 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
#define DECL64 uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = (uint64_t)seed << 20 | 1;
#define SINK64 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;

#define A_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a##i));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a##i) : "v"(b));
#define A_CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define A_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a##i) : "v"(b));
#define A_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a##i));
#define A_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a##i));
#define A_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(a##i) : "v"(b));
#define A_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define A_CMP64(i) asm volatile("v_cmp_ne_u64 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
#define A_SHL32(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##i));
#define A_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c" : "+v"(a##i) : "v"(b));

#define A_CNDM_S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(b));
#define A_CMPCNDM(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define A_ARITHSEL(i) asm volatile("v_sub_u32 %2, %0, %1\n v_ashrrev_i32 %2, 31, %2\n v_bfi_b32 %0, %2, %1, %0" : "+v"(a##i) : "v"(b), "v"(t##i));
#define DECL32T DECL32 uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
#define DECL32V DECL32 asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
#define DECL32S DECL32 asm volatile("s_mov_b64 s[20:21], 0x55555555" ::: "s20", "s21");
BODY(k_cndmask_vccset, DECL32V, REP8(A_CNDM), SINK32)
BODY(k_cndmask_sgpr, DECL32S, REP8(A_CNDM_S), SINK32)
BODY(k_cmp_cndmask, DECL32, REP8(A_CMPCNDM), SINK32)
BODY(k_arith_select, DECL32T, REP8(A_ARITHSEL), SINK32)
BODY(k_add32, DECL32, REP8(A_ADD32), SINK32)
BODY(k_and32, DECL32, REP8(A_AND32), SINK32)
BODY(k_bcnt, DECL32, REP8(A_BCNT), SINK32)
BODY(k_ffbh, DECL32, REP8(A_FFBH), SINK32)
BODY(k_bfi, DECL32, REP8(A_BFI), SINK32)
BODY(k_cndmask, DECL32, REP8(A_CNDM), SINK32)
BODY(k_shl32, DECL32, REP8(A_SHL32), SINK32)
BODY(k_bitop3, DECL32, REP8(A_BITOP3), SINK32)
BODY(k_addco_pair, DECL32, REP8(A_ADDCO), SINK32)
BODY(k_lshl_add_u64, DECL64, REP8(A_LSHLADD64), SINK64)
BODY(k_lshlrev_b64, DECL64, REP8(A_LSHL64), SINK64)
BODY(k_lshrrev_b64, DECL64, REP8(A_LSHR64), SINK64)
BODY(k_mov_b64, DECL64, REP8(A_MOV64), SINK64)
BODY(k_cmp_ne_u64, DECL64, REP8(A_CMP64), SINK64)

int main() {
    const int grid = 2048, iters = 4096;
    uint64_t* out;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    printf("%s, %d CUs, clock %d kHz; per-SIMD issue = wave instructions x 4 SIMD-cycles nominal\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    struct K { const char* name; void (*f)(uint64_t*, int, uint32_t); int per; } ks[] = {
        {"v_add_u32", k_add32, 1}, {"v_and_b32", k_and32, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bfi_b32", k_bfi, 1},
        {"v_cndmask_b32 (vcc untouched)", k_cndmask, 1}, {"v_cndmask_b32 (vcc set once)", k_cndmask_vccset, 1}, {"v_cndmask_b32_e64 s[20:21]", k_cndmask_sgpr, 1},
        {"v_cmp_gt_u32 + v_cndmask (pair)", k_cmp_cndmask, 2}, {"v_sub + v_ashr + v_bfi (triple)", k_arith_select, 3}, {"v_lshlrev_b32", k_shl32, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_add_co + v_addc (pair)", k_addco_pair, 2},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_mov_b64", k_mov_b64, 1},
        {"v_cmp_ne_u64", k_cmp_ne_u64, 1}};
    for (auto& k : ks) {
        hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, 16, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, out, iters, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)grid * 4 * iters * 64 * k.per; // 64 instructions per iteration and chain set
        const double per_simd_per_s = wave_instr / (pr.multiProcessorCount * 4.0) / (ms * 1e-3);
        printf("%-28s %8.3f ms  %7.1f M wave-instr/s per SIMD  -> %5.2f cycles per instruction at 2.4 GHz\n", k.name, ms, per_simd_per_s / 1e6, 2.4e9 / per_simd_per_s);
    }
    return 0;
}
