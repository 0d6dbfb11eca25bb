// ebwt_gen.hip -- synthetic-input tooling on the GPU (NOT part of the hot path, no reference counterpart):
// builds the eBWT (and document array) of error-free fixed-length reads sampled from known haplotypes,
// at benchmark scale (BASELINE.json configs[1]: 50 M x 150 bp = 7.55 G BWT positions) in seconds.
//
// The reference consumes eBWTs produced by external tools (BCR_LCP_GSA / egap, README.md:38) that do
// not exist offline; sorting 7.55 G read suffixes on the host would take hours.  Because every read is a
// substring of the haplotype text T, the order of the read suffixes W# follows from the suffix array of T:
//     key(W#) = ( lo(W), |W|, read index ),   lo(W) = first SA row of the suffixes of T that start with W.
// (W a proper prefix of W' -> lo(W) <= lo(W') and |W| < |W'|; first mismatch -> disjoint SA intervals;
//  equal strings -> read index, as in the eBWT definition used by ebwt2indel_amd/tools/ebwt_build.cpp.)
// Steps: (1) SA of T on its first 256 symbols by prefix doubling with radix sorts; (2) LCP of adjacent rows,
// capped at 255; (3) for each suffix length len = 0..L: lo_len[j] = max{ j' <= j : LCP[j'] < len } by a
// prefix-max scan, then one 64-bit key per read; (4) the keys go straight into slabs by lo (a counting pass sizes the
// slabs, a second pass places the keys: no n-key staging buffer, so 15 G keys fit one GPU) and keep only lo's offset
// INSIDE the slab -- (lo mod slab, |W|, read index) fits 64 bits whatever the text and collection sizes; each slab is
// radix-sorted on its own; (5) emit the symbol preceding each suffix (TERM for whole reads).
// tests/test_gpu_generator.py checks the result against the suffix-sorting host builder.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

namespace {

thread_local char g_err[512] = "";

#define GEN_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -4;                                                                                    \
        }                                                                                                 \
    } while (0)

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) {
        if (p) { (void)hipFree(p); p = nullptr; }
        return hipMalloc(&p, bytes ? bytes : 1);
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

__device__ __forceinline__ uint32_t code3(uint8_t c) {
    // A,C,G,T -> 1..4; separators (bytes 1,2,...) -> 5,6 (distinct, each occurs once); anything else / past the end -> 0
    return c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : c == 'T' ? 4u : c == 1 ? 5u : c == 2 ? 6u : 7u;
}

// initial key: first 8 symbols, 3 bits each
__global__ void __launch_bounds__(256) k_init_keys(const uint8_t* __restrict__ t, uint32_t m, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t k = 0;
    for (uint32_t d = 0; d < 8; ++d) k = (k << 3) | (i + d < m ? code3(t[i + d]) : 0u);
    keys[i] = k;
    idx[i] = i;
}

// after a sort: head[j] = j+1 where a new key group starts, else 0  (max-scan turns it into group-start index + 1)
__global__ void __launch_bounds__(256) k_group_heads(const uint64_t* __restrict__ keys, uint32_t m, uint32_t* __restrict__ head) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    head[j] = (j == 0 || keys[j] != keys[j - 1]) ? j + 1 : 0u;
}
__global__ void __launch_bounds__(256) k_scatter_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ grp, uint32_t m, uint32_t* __restrict__ rank) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) rank[sa[j]] = grp[j]; // >= 1
}
__global__ void __launch_bounds__(256) k_double_keys(const uint32_t* __restrict__ rank, uint32_t m, uint32_t h, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t hi = rank[i], lo = (uint64_t)i + h < m ? rank[i + h] : 0u;
    keys[i] = (hi << 32) | lo;
    idx[i] = i;
}
__global__ void __launch_bounds__(256) k_inverse(const uint32_t* __restrict__ sa, uint32_t m, uint32_t* __restrict__ inv) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) inv[sa[j]] = j;
}

// LCP of adjacent SA rows, capped at 255
__global__ void __launch_bounds__(256) k_lcp(const uint8_t* __restrict__ t, const uint32_t* __restrict__ sa, uint32_t m, uint8_t* __restrict__ lcp) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (j == 0) { lcp[0] = 0; return; }
    uint32_t a = sa[j - 1], b = sa[j];
    uint32_t lim = 255;
    if (m - a < lim) lim = m - a;
    if (m - b < lim) lim = m - b;
    uint32_t k = 0;
    while (k < lim && t[a + k] == t[b + k]) ++k;
    lcp[j] = (uint8_t)k;
}

__global__ void __launch_bounds__(256) k_lo_seed(const uint8_t* __restrict__ lcp, uint32_t m, uint32_t len, uint32_t* __restrict__ v) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) v[j] = lcp[j] < len ? j : 0u;
}

struct KeyLayout {
    uint32_t rbits;      // read index bits
    uint32_t slab_shift; // slab = lo >> slab_shift; the key holds lo's low slab_shift bits
};

__device__ __forceinline__ uint64_t make_key(uint32_t lo_in_slab, uint32_t len, uint64_t r, uint32_t rbits) {
    return ((uint64_t)lo_in_slab << (8 + rbits)) | ((uint64_t)len << rbits) | r;
}

// One key per read for suffix length `len` (offset i = L - len inside the read).  Pass 1 (keys == nullptr): count the
// keys of every slab; pass 2: place them, slab by slab, at the cursors the counts gave (order inside a slab is free:
// each slab is sorted afterwards).
__global__ void __launch_bounds__(256) k_make_keys(const uint64_t* __restrict__ starts, uint64_t n_reads, uint32_t L, uint32_t len,
                                                   const uint32_t* __restrict__ inv, const uint32_t* __restrict__ lo_len, KeyLayout kl,
                                                   uint64_t* __restrict__ keys, unsigned long long* __restrict__ slab_cursor) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint32_t lo = 0;
    if (len > 0) {
        uint64_t q = starts[r] + (L - len);
        lo = lo_len[inv[q]];
    }
    unsigned long long pos = atomicAdd(&slab_cursor[lo >> kl.slab_shift], 1ull);
    if (keys) keys[pos] = make_key(lo & ((1u << kl.slab_shift) - 1u), len, r, kl.rbits);
}

__global__ void __launch_bounds__(256) k_emit(const uint64_t* __restrict__ keys, uint64_t n, KeyLayout kl, const uint8_t* __restrict__ t,
                                              const uint64_t* __restrict__ starts, uint32_t L, uint8_t term, uint64_t n_first,
                                              uint8_t* __restrict__ bwt, uint8_t* __restrict__ da) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t key = keys[k];
        uint64_t r = key & ((1ull << kl.rbits) - 1ull);
        uint32_t len = (uint32_t)(key >> kl.rbits) & 0xFFu;
        uint32_t i = L - len;
        bwt[k] = i == 0 ? term : t[starts[r] + i - 1];
        if (da) da[k] = r < n_first ? '0' : '1';
    }
}

unsigned grid_of(uint64_t n) { return (unsigned)((n + 255) / 256); }
unsigned grid_capped(uint64_t n) { uint64_t b = (n + 255) / 256; return (unsigned)(b > (1ull << 22) ? (1ull << 22) : b); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

const char* ebgen_last_error(void) { return g_err; }

void ebgen_free(void* d_ptr) {
    if (d_ptr) (void)hipFree(d_ptr);
}

int ebgen_to_host(void* dst, const void* d_src, uint64_t bytes) {
    GEN_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// text      : haplotypes concatenated, each followed by its own separator byte (1 after the first, 2 after the second);
//             only A,C,G,T elsewhere.  text_len < 2^32.
// starts    : n_reads offsets into text; read r = text[starts[r] .. starts[r]+read_len) and must not contain a separator.
// read_len  : <= 254.   n_first: reads [0,n_first) form collection 0 (document array '0'), the rest collection 1.
// Outputs (device memory on `device`, free with ebgen_free): *d_bwt = n_reads*(read_len+1) ASCII symbols;
//             *d_da likewise ('0'/'1') when d_da != NULL.  timings[4] = {suffix array, lcp, keys, sort+emit} seconds.
int ebgen_build(const uint8_t* text, uint64_t text_len, const uint64_t* starts, uint64_t n_reads, uint32_t read_len, uint8_t term,
                uint64_t n_first, int device, void** d_bwt, void** d_da, uint64_t* n_out, double* timings) {
    if (!text || !starts || !d_bwt || !n_out || n_reads == 0 || read_len == 0 || read_len > 254 || text_len >= 0xFFFFFFF0ull) {
        snprintf(g_err, sizeof g_err, "bad arguments");
        return -1;
    }
    GEN_HIP(hipSetDevice(device));
    hipStream_t st = nullptr; // default stream: everything here is sequential
    const uint32_t m = (uint32_t)text_len;
    const uint32_t L = read_len;
    const uint64_t n = n_reads * (uint64_t)(L + 1);
    KeyLayout kl;
    kl.rbits = 1;
    while ((1ull << kl.rbits) < n_reads) ++kl.rbits;
    if (const char* e = getenv("EBGEN_RBITS")) { uint32_t v = (uint32_t)atoi(e); if (v > kl.rbits && v <= 48) kl.rbits = v; } // tests: a wide read-index field
    // slabs: few enough that each is a decent sort, many enough that (lo inside the slab, 8-bit length, read index) fits
    // 64 bits and that the largest slab's sort buffer stays small next to the n keys
    uint32_t n_slabs_target = n > (1ull << 32) ? 4096 : (n > (1ull << 30) ? 256 : (n > (1ull << 24) ? 16 : 1));
    if (const char* e = getenv("EBGEN_SLABS")) n_slabs_target = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : n_slabs_target; // tests
    kl.slab_shift = 0;
    while (((uint64_t)m >> kl.slab_shift) + 1 > n_slabs_target) ++kl.slab_shift;
    while (kl.slab_shift > 1 && kl.slab_shift + 8 + kl.rbits > 64) --kl.slab_shift; // more slabs until the key fits
    if (kl.slab_shift + 8 + kl.rbits > 64) { snprintf(g_err, sizeof g_err, "key does not fit 64 bits (%u read-index bits)", kl.rbits); return -1; }
    if (kl.slab_shift > 31) kl.slab_shift = 31;
    const uint32_t n_slabs = (uint32_t)(((uint64_t)m >> kl.slab_shift) + 1);

    double t0 = now_s();
    Buf d_text, d_starts, d_keys, d_keys2, d_idx, d_idx2, d_rank, d_tmp, d_u32, d_lcp, d_inv;
    GEN_HIP(d_text.alloc((size_t)m + 16));
    GEN_HIP(hipMemset(d_text.p, 0, (size_t)m + 16));
    GEN_HIP(hipMemcpy(d_text.p, text, m, hipMemcpyHostToDevice));
    GEN_HIP(d_starts.alloc(n_reads * 8));
    GEN_HIP(hipMemcpy(d_starts.p, starts, n_reads * 8, hipMemcpyHostToDevice));

    // ---- (1) suffix array of T on its first 256 symbols
    GEN_HIP(d_keys.alloc((size_t)m * 8));
    GEN_HIP(d_keys2.alloc((size_t)m * 8));
    GEN_HIP(d_idx.alloc((size_t)m * 4));
    GEN_HIP(d_idx2.alloc((size_t)m * 4));
    GEN_HIP(d_rank.alloc((size_t)m * 4));
    GEN_HIP(d_u32.alloc((size_t)m * 4));
    size_t sort_tmp = 0, scan_tmp = 0;
    GEN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(),
                                               (size_t)m, 0, 64, st));
    GEN_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, scan_tmp, d_u32.as<uint32_t>(), d_u32.as<uint32_t>(), hipcub::Max(), (size_t)m, st));
    GEN_HIP(d_tmp.alloc(sort_tmp > scan_tmp ? sort_tmp : scan_tmp));
    size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    hipLaunchKernelGGL(k_init_keys, dim3(grid_of(m)), dim3(256), 0, st, d_text.as<uint8_t>(), m, d_keys.as<uint64_t>(), d_idx.as<uint32_t>());
    uint32_t h = 8;
    for (;;) {
        int end_bit = h == 8 ? 24 : 64;
        GEN_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(),
                                                   (size_t)m, 0, end_bit, st));
        // keys2 / idx2 hold the sorted order
        if (h >= 256) break;
        hipLaunchKernelGGL(k_group_heads, dim3(grid_of(m)), dim3(256), 0, st, d_keys2.as<uint64_t>(), m, d_u32.as<uint32_t>());
        GEN_HIP(hipcub::DeviceScan::InclusiveScan(d_tmp.p, tmp_bytes, d_u32.as<uint32_t>(), d_u32.as<uint32_t>(), hipcub::Max(), (size_t)m, st));
        hipLaunchKernelGGL(k_scatter_rank, dim3(grid_of(m)), dim3(256), 0, st, d_idx2.as<uint32_t>(), d_u32.as<uint32_t>(), m, d_rank.as<uint32_t>());
        hipLaunchKernelGGL(k_double_keys, dim3(grid_of(m)), dim3(256), 0, st, d_rank.as<uint32_t>(), m, h, d_keys.as<uint64_t>(), d_idx.as<uint32_t>());
        h *= 2;
    }
    GEN_HIP(hipDeviceSynchronize());
    // d_idx2 = suffix array
    d_keys.release();
    d_keys2.release();
    d_idx.release();
    d_rank.release();
    double t1 = now_s();

    // ---- (2) LCP (capped) and inverse suffix array
    GEN_HIP(d_lcp.alloc(m));
    GEN_HIP(d_inv.alloc((size_t)m * 4));
    hipLaunchKernelGGL(k_lcp, dim3(grid_of(m)), dim3(256), 0, st, d_text.as<uint8_t>(), d_idx2.as<uint32_t>(), m, d_lcp.as<uint8_t>());
    hipLaunchKernelGGL(k_inverse, dim3(grid_of(m)), dim3(256), 0, st, d_idx2.as<uint32_t>(), m, d_inv.as<uint32_t>());
    GEN_HIP(hipDeviceSynchronize());
    d_idx2.release();
    double t2 = now_s();

    // ---- (3) + (4) keys, one suffix length at a time: first only counted per slab, then placed into their slabs
    Buf d_slab_count, d_lo, d_slabbed, d_alt;
    GEN_HIP(d_slab_count.alloc((size_t)n_slabs * 8));
    GEN_HIP(hipMemset(d_slab_count.p, 0, (size_t)n_slabs * 8));
    GEN_HIP(d_lo.alloc((size_t)m * 4));
    std::vector<unsigned long long> cnt(n_slabs), off(n_slabs + 1, 0);
    uint64_t max_slab = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t len = 0; len <= L; ++len) {
            if (len > 0) {
                hipLaunchKernelGGL(k_lo_seed, dim3(grid_of(m)), dim3(256), 0, st, d_lcp.as<uint8_t>(), m, len, d_u32.as<uint32_t>());
                GEN_HIP(hipcub::DeviceScan::InclusiveScan(d_tmp.p, tmp_bytes, d_u32.as<uint32_t>(), d_lo.as<uint32_t>(), hipcub::Max(), (size_t)m, st));
            }
            hipLaunchKernelGGL(k_make_keys, dim3(grid_of(n_reads)), dim3(256), 0, st, d_starts.as<uint64_t>(), n_reads, L, len, d_inv.as<uint32_t>(),
                               d_lo.as<uint32_t>(), kl, pass ? d_slabbed.as<uint64_t>() : (uint64_t*)nullptr, d_slab_count.as<unsigned long long>());
        }
        GEN_HIP(hipDeviceSynchronize());
        GEN_HIP(hipGetLastError());
        if (pass == 0) {
            GEN_HIP(hipMemcpy(cnt.data(), d_slab_count.p, (size_t)n_slabs * 8, hipMemcpyDeviceToHost));
            for (uint32_t s = 0; s < n_slabs; ++s) { off[s + 1] = off[s] + cnt[s]; if (cnt[s] > max_slab) max_slab = cnt[s]; }
            if (off[n_slabs] != n) { snprintf(g_err, sizeof g_err, "internal: slab counts do not add up"); return -6; }
            GEN_HIP(hipMemcpy(d_slab_count.p, off.data(), (size_t)n_slabs * 8, hipMemcpyHostToDevice)); // cursors of pass 2
            GEN_HIP(d_slabbed.alloc(n * 8));
        }
    }
    d_lo.release();
    d_u32.release();
    d_lcp.release();
    d_inv.release();
    double t3 = now_s();

    {
        GEN_HIP(d_alt.alloc(max_slab * 8));
        size_t tb = 0;
        GEN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_slabbed.as<uint64_t>(), d_alt.as<uint64_t>(), (size_t)max_slab, 0, 64, st));
        if (tb > tmp_bytes) { GEN_HIP(d_tmp.alloc(tb)); tmp_bytes = tb; }
        const int key_bits = (int)(kl.slab_shift + 8 + kl.rbits);
        for (uint32_t s = 0; s < n_slabs; ++s) {
            if (cnt[s] == 0) continue;
            uint64_t* seg = d_slabbed.as<uint64_t>() + off[s];
            GEN_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, seg, d_alt.as<uint64_t>(), (size_t)cnt[s], 0, key_bits, st));
            GEN_HIP(hipMemcpyAsync(seg, d_alt.p, cnt[s] * 8, hipMemcpyDeviceToDevice, st));
        }
        GEN_HIP(hipDeviceSynchronize());
        d_alt.release();
    }
    uint64_t* sorted = d_slabbed.as<uint64_t>();

    // ---- (5) emit
    void* out_bwt = nullptr;
    void* out_da = nullptr;
    GEN_HIP(hipMalloc(&out_bwt, n));
    if (d_da) GEN_HIP(hipMalloc(&out_da, n));
    hipLaunchKernelGGL(k_emit, dim3(grid_capped(n)), dim3(256), 0, st, sorted, n, kl, d_text.as<uint8_t>(), d_starts.as<uint64_t>(), L, term, n_first,
                       static_cast<uint8_t*>(out_bwt), static_cast<uint8_t*>(out_da));
    GEN_HIP(hipDeviceSynchronize());
    GEN_HIP(hipGetLastError());
    double t4 = now_s();
    if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; timings[2] = t3 - t2; timings[3] = t4 - t3; }
    *d_bwt = out_bwt;
    if (d_da) *d_da = out_da;
    *n_out = n;
    return 0;
}

}  // extern "C"
