// ebwt_gen_slabs.hip -- synthetic-input tooling on the GPU (NOT part of the hot path, no reference counterpart):
// the eBWT generator for collections whose sort keys do not fit one GPU at once -- BASELINE.json configs[4],
// 1 G x 150 bp DISTINCT reads of a 5 Gb diploid genome = 151 G BWT positions (1.2 TB of 64-bit keys) -- and whose
// haplotype text has 2^32 symbols or more.  Same definition of the result as ebwt_gen.hip / ebwt_build.cpp:
//     key(W#) = ( lo(W), |W|, read index ),   lo(W) = first suffix-array row of the text T' that starts with W,
// for every suffix W of every read; T' = any text that contains every read (here: the haplotypes with separators and
// end padding read as 'A', packed 2 bits per symbol -- 2.5 GB for 10 G symbols).  What is different is that nothing of
// size n, and no suffix array of the whole text, is ever resident:
//   * Suffixes W shorter than P symbols ("short blocks", P <= 6) are whole runs of the output: all reads that end with
//     W, in read order.  One stable radix sort of the reads per length.
//   * Every other suffix belongs to the SLAB of its first P symbols; a slab's interval of suffix-array rows holds the
//     whole interval of each of its W's, so slabs are independent: 4^P of them, each a few hundred million keys.
//   * The rows of a slab come from sorting the text positions of a GROUP (first Pg <= P symbols) by their next 32
//     symbols (one radix sort of 64-bit windows), ties -- the two haplotypes between variants -- by direct comparison
//     up to the read length; LCP of adjacent rows by window comparison.  No prefix doubling, no inverse suffix array.
//   * lo(W) for a row and a length: table of the first row of every (Pg+11)-mer for short W, a walk down the LCP
//     array (a few rows) for long W; the reads that cover a text position come from the sorted start positions.
//   * Output offsets of slabs and short blocks: depth-first order of the 4-ary trie of prefixes, from two histograms.
// The output (n ASCII symbols, optionally the document array) is written slab by slab into one device buffer.
// tests/test_gpu_generator.py: equal to the suffix-sorting host builder and to ebwt_gen.hip for P, Pg in 0..5;
// tests/test_gpu_fullsize.py: the 7.55 G-position input of the bench has the recorded SHA-256 through this path too.
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

#define GS_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -4;                                                                                    \
        }                                                                                                 \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = b ? b : 1;
        return hipMalloc(&p, bytes);
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

constexpr uint32_t PAD = 512;        // symbols of 'A' behind the text: every window of a comparison stays inside the array
constexpr uint32_t TBL_BITS = 22;    // first-row table of a group: 11 symbols behind the group prefix
constexpr uint32_t TIE_CAP = 4096;   // longest run of text positions equal on 32 symbols that is sorted by comparison
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t code2(uint8_t c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; } // 'A', separators, padding: 0

// 32 symbols starting at q; symbol k of the window at bits 63-2k .. 62-2k (integer order = lexicographic order)
__device__ __forceinline__ uint64_t window(const uint64_t* __restrict__ t2, uint64_t q) {
    const uint64_t w = q >> 5;
    const uint32_t s = (uint32_t)(q & 31) * 2;
    const uint64_t a = t2[w];
    if (s == 0) return a;
    return (a << s) | (t2[w + 1] >> (64 - s));
}
__device__ __forceinline__ uint32_t symbol(const uint64_t* __restrict__ t2, uint64_t q) { return (uint32_t)(t2[q >> 5] >> (62 - 2 * (q & 31))) & 3u; }

__global__ void __launch_bounds__(256) k_pack2(const uint8_t* __restrict__ t, uint64_t m, uint64_t n_words, uint64_t* __restrict__ t2) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t base = w * 32;
    uint64_t v = 0;
    if (base + 32 <= m) {
        const uint4 lo = *reinterpret_cast<const uint4*>(t + base), hi = *reinterpret_cast<const uint4*>(t + base + 16);
        const uint32_t words[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint64_t)code2((uint8_t)(words[k] >> (8 * b))) << (62 - 2 * (4 * k + b));
    } else {
        for (uint32_t j = 0; j < 32; ++j)
            if (base + j < m) v |= (uint64_t)code2(t[base + j]) << (62 - 2 * j);
    }
    t2[w] = v;
}

// how many text positions start with each P-mer (P <= 6); one thread per 32 positions
__global__ void __launch_bounds__(256) k_text_hist(const uint64_t* __restrict__ t2, uint64_t m, uint32_t P, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[4096];
    const uint32_t bins = 1u << (2 * P);
    for (uint32_t i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t n_words = (m + 31) / 32;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * 256) {
        const uint64_t a = t2[w], b = t2[w + 1];
        for (uint32_t j = 0; j < 32 && w * 32 + j < m; ++j) {
            const uint64_t win = j ? (a << (2 * j)) | (b >> (64 - 2 * j)) : a;
            atomicAdd(&h[P ? (uint32_t)(win >> (64 - 2 * P)) : 0u], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// per read: the P-mers at offsets 0 .. L-P (one key each in the slab of that P-mer), and the read's last l symbols for l < P
// short[] holds the counts of lengths 1 .. P-1 one after the other: length l at offset (4^l - 4) / 3
__global__ void __launch_bounds__(256) k_read_hist(const uint64_t* __restrict__ t2, const uint64_t* __restrict__ starts, uint64_t n_reads, uint32_t L, uint32_t P,
                                                   unsigned long long* __restrict__ hist, unsigned long long* __restrict__ short_cnt) {
    __shared__ uint32_t h[4096];
    __shared__ uint32_t sh[1368];
    const uint32_t bins = 1u << (2 * P), sbins = P ? ((1u << (2 * P)) - 4) / 3 : 0;
    for (uint32_t i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    for (uint32_t i = threadIdx.x; i < sbins; i += 256) sh[i] = 0;
    __syncthreads();
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * 256) {
        const uint64_t p = starts[r];
        uint64_t q = p;
        uint32_t left = L + 1 - P;   // L >= P is checked by the host
        while (left) {
            const uint64_t cur = window(t2, q);
            const uint32_t c = left < 33 - P ? left : 33 - P;
            for (uint32_t j = 0; j < c; ++j) atomicAdd(&h[P ? (uint32_t)((cur << (2 * j)) >> (64 - 2 * P)) : 0u], 1u);
            q += c;
            left -= c;
        }
        for (uint32_t l = 1; l < P; ++l) atomicAdd(&sh[((1u << (2 * l)) - 4) / 3 + (uint32_t)(window(t2, p + L - l) >> (64 - 2 * l))], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
    for (uint32_t i = threadIdx.x; i < sbins; i += 256)
        if (sh[i]) atomicAdd(&short_cnt[i], (unsigned long long)sh[i]);
}

__global__ void __launch_bounds__(256) k_iota(uint32_t* __restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// C[b] = number of sorted starts below 64 b
__global__ void __launch_bounds__(256) k_ctable(const uint64_t* __restrict__ ss, uint64_t n_reads, uint32_t* __restrict__ C, uint64_t cn) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cn) return;
    const uint64_t x = b << 6;
    uint64_t lo = 0, hi = n_reads;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ss[mid] < x) lo = mid + 1; else hi = mid;
    }
    C[b] = (uint32_t)lo;
}
__device__ __forceinline__ uint32_t lower_bound_start(const uint64_t* __restrict__ ss, const uint32_t* __restrict__ C, uint64_t x) {
    const uint64_t b = x >> 6;
    uint32_t s = C[b];
    const uint32_t e = C[b + 1];
    while (s < e && ss[s] < x) ++s;
    return s;
}

// the text positions of group G (first Pg symbols), in no particular order, with the 32 symbols behind the group prefix
__global__ void __launch_bounds__(256) k_collect(const uint64_t* __restrict__ t2, uint64_t m, uint32_t Pg, uint32_t G, unsigned long long* __restrict__ cursor,
                                                 uint64_t* __restrict__ out_key, uint64_t* __restrict__ out_q) {
    const uint64_t n_words = (m + 31) / 32;
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint64_t a = 0, b = 0, c = 0;
    uint32_t mask = 0;
    if (w < n_words) {
        a = t2[w];
        b = t2[w + 1];
        c = t2[w + 2];
        for (uint32_t j = 0; j < 32 && w * 32 + j < m; ++j) {
            const uint64_t win = j ? (a << (2 * j)) | (b >> (64 - 2 * j)) : a;
            if ((Pg ? (uint32_t)(win >> (64 - 2 * Pg)) : 0u) == G) mask |= 1u << j;
        }
    }
    const uint32_t cnt = __popc(mask);
    uint32_t incl = cnt;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd(cursor, (unsigned long long)total);
    base = __shfl(base, 63);
    uint64_t at = base + incl - cnt;
    while (mask) {
        const uint32_t j = __ffs(mask) - 1;
        mask &= mask - 1;
        const uint32_t s = 2 * (j + Pg);
        uint64_t key;
        if (s == 0) key = a;
        else if (s < 64) key = (a << s) | (b >> (64 - s));
        else if (s == 64) key = b;
        else key = (b << (s - 64)) | (c >> (128 - s));
        out_key[at] = key;
        out_q[at] = w * 32 + j;
        ++at;
    }
}

// order of two text positions whose first Pg + 32 symbols are equal: by the symbols behind them, up to `upto`; then by position
__device__ __forceinline__ bool tie_less(const uint64_t* __restrict__ t2, uint64_t a, uint64_t b, uint32_t from, uint32_t upto) {
    for (uint32_t off = from; off < upto; off += 32) {
        const uint64_t wa = window(t2, a + off), wb = window(t2, b + off);
        if (wa != wb) return wa < wb;
    }
    return a < b;
}
__global__ void __launch_bounds__(256) k_tiefix(const uint64_t* __restrict__ skey, uint64_t* __restrict__ sq, uint64_t M, const uint64_t* __restrict__ t2, uint32_t Pg,
                                                uint32_t upto, uint32_t* __restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= M) return;
    const uint64_t k = skey[i];
    if ((i && skey[i - 1] == k) || skey[i + 1] != k) return;
    uint64_t e = i + 2;
    while (e < M && skey[e] == k && e - i <= TIE_CAP) ++e;
    if (e - i > TIE_CAP) { atomicAdd(err, 1u); return; }
    for (uint64_t x = i + 1; x < e; ++x) { // insertion sort: runs are pairs on a diploid sample
        const uint64_t v = sq[x];
        uint64_t y = x;
        while (y > i && tie_less(t2, v, sq[y - 1], Pg + 32, upto)) { sq[y] = sq[y - 1]; --y; }
        sq[y] = v;
    }
}

// LCP of adjacent rows, capped at 255 (lengths compared with it are <= 254)
__global__ void __launch_bounds__(256) k_lcp(const uint64_t* __restrict__ sq, uint64_t M, const uint64_t* __restrict__ t2, uint8_t* __restrict__ lcp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (i == 0) { lcp[0] = 0; return; }
    const uint64_t a = sq[i - 1], b = sq[i];
    uint32_t k = 0;
    while (k < 256) {
        const uint64_t x = window(t2, a + k) ^ window(t2, b + k);
        if (x) { k += (uint32_t)__clzll((long long)x) >> 1; break; }
        k += 32;
    }
    lcp[i] = (uint8_t)(k > 255 ? 255 : k);
}

// first row of every (Pg + 11)-mer of the group; tbl is EMPTY-filled; then every empty entry takes the next filled one
__global__ void __launch_bounds__(256) k_tbl_heads(const uint64_t* __restrict__ skey, uint64_t M, uint32_t* __restrict__ tbl) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint32_t t = (uint32_t)(skey[i] >> (64 - TBL_BITS));
    if (i == 0 || t != (uint32_t)(skey[i - 1] >> (64 - TBL_BITS))) tbl[t] = (uint32_t)i;
}
__global__ void __launch_bounds__(1024) k_tbl_fill(uint32_t* __restrict__ tbl, uint32_t M) {
    __shared__ uint32_t first[1024];
    constexpr uint32_t chunk = (1u << TBL_BITS) / 1024;
    const uint32_t t = threadIdx.x, lo = t * chunk, hi = lo + chunk;
    uint32_t run = EMPTY;
    for (uint32_t e = hi; e-- > lo;) {
        const uint32_t v = tbl[e];
        if (v != EMPTY) run = v; else if (run != EMPTY) tbl[e] = run;
    }
    first[t] = run;
    __syncthreads();
    if (t == 0) { // carry over the chunks, last to first
        uint32_t carry = M;
        for (uint32_t c = 1024; c-- > 0;) {
            const uint32_t f = first[c];
            first[c] = carry;
            if (f != EMPTY) carry = f;
        }
    }
    __syncthreads();
    const uint32_t carry = first[t];
    for (uint32_t e = hi; e-- > lo && tbl[e] == EMPTY;) tbl[e] = carry;
}

// reads that hold text position q as the start of a suffix of P symbols or more: starts in [q + P - L, q]
__device__ __forceinline__ void covering(const uint64_t* __restrict__ ss, const uint32_t* __restrict__ C, uint64_t q, uint32_t L, uint32_t P, uint32_t* s0, uint32_t* s1) {
    *s1 = lower_bound_start(ss, C, q + 1);
    *s0 = q + P >= L ? lower_bound_start(ss, C, q + P - L) : 0u;
}
__global__ void __launch_bounds__(256) k_count(const uint64_t* __restrict__ sq, uint64_t M, const uint64_t* __restrict__ ss, const uint32_t* __restrict__ C, uint32_t L, uint32_t P,
                                               uint32_t* __restrict__ cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    uint32_t s0, s1;
    covering(ss, C, sq[i], L, P, &s0, &s1);
    cnt[i] = s1 - s0;
}
// sq / skey / lcp point at the slab's first row; tbl holds rows of the group, `ra` is the slab's first row in the group
__global__ void __launch_bounds__(256) k_fill(const uint64_t* __restrict__ sq, const uint64_t* __restrict__ skey, const uint8_t* __restrict__ lcp, uint64_t M,
                                              const uint64_t* __restrict__ ss, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ C,
                                              const uint32_t* __restrict__ tbl, uint32_t ra, uint32_t L, uint32_t P, uint32_t Pg, uint32_t rbits,
                                              const uint32_t* __restrict__ koff, uint64_t* __restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint64_t q = sq[i];
    uint32_t s0, s1;
    covering(ss, C, q, L, P, &s0, &s1);
    const uint32_t top = (uint32_t)(skey[i] >> (64 - TBL_BITS));
    uint64_t j = i;                       // walks down the LCP array as the lengths get shorter
    uint64_t at = koff[i];
    for (uint32_t s = s1; s-- > s0;) {
        const uint32_t len = (uint32_t)(ss[s] + L - q);
        uint64_t lo;
        if (len <= Pg + TBL_BITS / 2) {
            const uint32_t nsym = len - Pg;
            lo = tbl[top & ~((1u << (TBL_BITS - 2 * nsym)) - 1u)] - ra;
        } else {
            while (j > 0 && lcp[j] >= len) --j;
            lo = j;
        }
        keys[at++] = (lo << (8 + rbits)) | ((uint64_t)len << rbits) | perm[s];
    }
}

__global__ void __launch_bounds__(256) k_emit(const uint64_t* __restrict__ keys, uint64_t n, uint32_t rbits, const uint64_t* __restrict__ t2, const uint64_t* __restrict__ starts,
                                              uint32_t L, uint8_t term, uint64_t n_first, uint8_t* __restrict__ bwt, uint8_t* __restrict__ da) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[k];
        const uint64_t r = key & ((1ull << rbits) - 1ull);
        const uint32_t len = (uint32_t)(key >> rbits) & 0xFFu;
        const uint32_t i = L - len;
        bwt[k] = i == 0 ? term : (uint8_t)"ACGT"[symbol(t2, starts[r] + i - 1)];
        if (da) da[k] = r < n_first ? '0' : '1';
    }
}

// ---- the LCP bit vectors of the collection, straight from the definition (an independent statement of what phases 2-3 of the
// product compute by suffix-tree navigation; ebwt2InDel.cpp:344-391 in LCP terms, checked against the CPU oracle in
// tests/test_gpu_generator.py): LCP[i] = length of the common prefix of the read suffixes W of rows i-1 and i (terminators never
// match), thrK[i] = LCP[i] >= K, thrR[i] = LCP[i] >= k_right, minima[i] = LCP[i] < LCP[i-1] and LCP[i] <= LCP[i+1].
// Adjacent keys of a sorted slab: same lo -> the shorter W is a prefix of the longer; else the LCP of the text rows in between.
__global__ void __launch_bounds__(256) k_row_lcp(const uint64_t* __restrict__ keys, uint64_t K, uint32_t rbits, const uint8_t* __restrict__ lcp_text, uint8_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    if (j == 0) { out[0] = 0; return; }   // against the last row of another slab or block: fewer than P symbols (not given here)
    const uint64_t a = keys[j - 1], b = keys[j];
    const uint64_t lo_a = a >> (8 + rbits), lo_b = b >> (8 + rbits);
    const uint32_t len_a = (uint32_t)(a >> rbits) & 0xFFu, len_b = (uint32_t)(b >> rbits) & 0xFFu;
    uint32_t v = len_a < len_b ? len_a : len_b;
    for (uint64_t t = lo_a + 1; t <= lo_b && v; ++t) { const uint32_t x = lcp_text[t]; if (x < v) v = x; }
    out[j] = (uint8_t)v;
}
// one thread per 64 output rows; `first` = output row of the slab's key 0
__global__ void __launch_bounds__(256) k_truth_bits(const uint8_t* __restrict__ row_lcp, uint64_t K, uint64_t first, uint32_t thr_k, uint32_t thr_r,
                                                    unsigned long long* __restrict__ bits_k, unsigned long long* __restrict__ bits_r, unsigned long long* __restrict__ bits_min) {
    const uint64_t w = (first >> 6) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > ((first + K - 1) >> 6)) return;
    unsigned long long mk = 0, mr = 0, mm = 0;
    for (uint32_t b = 0; b < 64; ++b) {
        const uint64_t g = (w << 6) + b;
        if (g < first + 1 || g >= first + K) continue;
        const uint64_t j = g - first;
        const uint32_t v = row_lcp[j];
        if (v >= thr_k) mk |= 1ull << b;
        if (v >= thr_r) mr |= 1ull << b;
        if (j >= 2 && j + 1 < K && v < row_lcp[j - 1] && v <= row_lcp[j + 1]) mm |= 1ull << b;
    }
    if (mk) atomicOr(&bits_k[w], mk);
    if (mr) atomicOr(&bits_r[w], mr);
    if (mm) atomicOr(&bits_min[w], mm);
}

// suffixes of l < P symbols: key = (the l symbols, read index); a stable sort on the symbols keeps the read order
__global__ void __launch_bounds__(256) k_short_keys(const uint64_t* __restrict__ t2, const uint64_t* __restrict__ starts, uint64_t n_reads, uint32_t L, uint32_t l,
                                                    uint32_t rbits, uint64_t* __restrict__ keys) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t code = l ? window(t2, starts[r] + L - l) >> (64 - 2 * l) : 0ull;
    keys[r] = (code << rbits) | r;
}
__global__ void __launch_bounds__(256) k_short_emit(const uint64_t* __restrict__ keys, uint64_t n_reads, uint32_t rbits, uint32_t l, const uint64_t* __restrict__ t2,
                                                    const uint64_t* __restrict__ starts, uint32_t L, const unsigned long long* __restrict__ first_index,
                                                    const unsigned long long* __restrict__ out_offset, uint64_t n_first, uint8_t* __restrict__ bwt, uint8_t* __restrict__ da) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const uint64_t key = keys[i];
    const uint64_t r = key & ((1ull << rbits) - 1ull);
    const uint64_t code = key >> rbits;
    const uint64_t at = out_offset[code] + (i - first_index[code]);
    bwt[at] = (uint8_t)"ACGT"[symbol(t2, starts[r] + L - l - 1)];   // l < P <= L: there is a symbol in front
    if (da) da[at] = r < n_first ? '0' : '1';
}

// ---- synthetic diploid sample made on the device (the host recipe of sim.py, counter-based instead of sequential)
__device__ __forceinline__ uint64_t mix(uint64_t seed, uint64_t x) {
    uint64_t z = seed + (x + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct IndelOfBlock { uint64_t at; int32_t delta; }; // position in the first haplotype; +len inserted before it, -len deleted from it, 0 none
__device__ __forceinline__ IndelOfBlock indel_of(uint64_t b, uint64_t G, uint32_t snp_every, uint32_t indel_every, uint32_t max_indel, uint64_t seed) {
    IndelOfBlock r{0, 0};
    const uint64_t at = b * indel_every + indel_every / 2 + snp_every / 4 + 7;
    if (at >= G || at >= (b + 1) * (uint64_t)indel_every) return r;
    const uint32_t len = 1 + (uint32_t)(mix(seed ^ 0xB1, b) % max_indel);
    r.at = at;
    if (mix(seed ^ 0xA1, b) & 1) r.delta = (int32_t)len;
    else { const uint64_t room = (b + 1) * (uint64_t)indel_every < G ? (b + 1) * (uint64_t)indel_every - at : G - at; r.delta = -(int32_t)(len < room ? len : room); }
    return r;
}
__global__ void __launch_bounds__(256) k_synth_hap1(uint8_t* __restrict__ text, uint64_t G, uint64_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (uint64_t)gridDim.x * blockDim.x) text[i] = (uint8_t)"ACGT"[mix(seed, i) & 3];
}
__global__ void __launch_bounds__(256) k_synth_len2(uint64_t G, uint64_t n_blocks, uint32_t snp_every, uint32_t indel_every, uint32_t max_indel, uint64_t seed,
                                                    unsigned long long* __restrict__ len2) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t lo = b * indel_every, hi = lo + indel_every < G ? lo + indel_every : G;
    len2[b] = (unsigned long long)((int64_t)(hi - lo) + indel_of(b, G, snp_every, indel_every, max_indel, seed).delta);
}
// one workgroup per block of indel_every bases of the first haplotype
__global__ void __launch_bounds__(256) k_synth_hap2(const uint8_t* __restrict__ hap1, uint8_t* __restrict__ hap2, uint64_t G, uint32_t snp_every, uint32_t indel_every,
                                                    uint32_t max_indel, uint64_t seed, const unsigned long long* __restrict__ off2, const unsigned long long* __restrict__ len2) {
    const uint64_t b = blockIdx.x;
    const uint64_t lo = b * indel_every;
    const IndelOfBlock id = indel_of(b, G, snp_every, indel_every, max_indel, seed);
    const uint64_t rel = id.delta ? id.at - lo : ~0ull;
    for (uint64_t o = threadIdx.x; o < len2[b]; o += 256) {
        uint64_t src;
        bool inserted = false;
        if (id.delta > 0) {
            if (o < rel) src = lo + o;
            else if (o < rel + (uint64_t)id.delta) { inserted = true; src = 0; }
            else src = lo + o - (uint64_t)id.delta;
        } else if (id.delta < 0) src = o < rel ? lo + o : lo + o + (uint64_t)(-id.delta);
        else src = lo + o;
        uint8_t c;
        if (inserted) c = (uint8_t)"ACGT"[mix(seed ^ 0xD1, b * 8 + (o - rel)) & 3];
        else {
            c = hap1[src];
            if (snp_every && src % snp_every == snp_every / 2) {
                const uint32_t old = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
                c = (uint8_t)"ACGT"[(old + 1 + (uint32_t)(mix(seed ^ 0xC1, src) % 3)) & 3];
            }
        }
        hap2[off2[b] + o] = c;
    }
}
__global__ void __launch_bounds__(256) k_synth_starts(uint64_t* __restrict__ starts, uint64_t n_reads, uint32_t L, uint64_t off1, uint64_t len1, uint64_t off2, uint64_t len2, uint64_t seed) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t h = mix(seed ^ 0xE1, r);
    starts[r] = (r & 1) ? off2 + h % (len2 - L + 1) : off1 + h % (len1 - L + 1); // read r comes from haplotype r % 2 (sim.sample_read_starts)
}

unsigned grid_of(uint64_t n) { return (unsigned)((n + 255) / 256); }
unsigned grid_capped(uint64_t n) { const uint64_t b = (n + 255) / 256; return (unsigned)(b > (1ull << 20) ? (1ull << 20) : (b ? b : 1)); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
uint32_t bits_for(uint64_t v) { uint32_t b = 1; while (b < 63 && (1ull << b) < v) ++b; return b; }

}  // namespace

extern "C" {

const char* ebgen_slabs_last_error(void) { return g_err; }

// A diploid sample made on `device`: a uniformly random genome of `genome_len` bases, a second haplotype with a substitution
// every `snp_every` bases and an insertion or deletion of 1..max_indel bases every `indel_every` bases, n_reads start positions
// alternating between the haplotypes.  *d_text = haplotype 1, byte 1, haplotype 2, byte 2 (the layout of sim.haplotype_text).
int ebgen_synth_diploid(uint64_t genome_len, uint64_t seed, uint32_t snp_every, uint32_t indel_every, uint32_t max_indel, uint64_t n_reads, uint32_t read_len,
                        int device, void** d_text, uint64_t* text_len, void** d_starts, uint64_t* hap2_len) {
    if (!d_text || !text_len || !d_starts || genome_len <= read_len || indel_every == 0 || max_indel == 0 || max_indel > 7 || n_reads == 0) {
        snprintf(g_err, sizeof g_err, "bad arguments");
        return -1;
    }
    GS_HIP(hipSetDevice(device));
    const uint64_t G = genome_len, nb = (G + indel_every - 1) / indel_every;
    Buf len2, off2, tmp;
    GS_HIP(len2.alloc((nb + 1) * 8));
    GS_HIP(off2.alloc((nb + 1) * 8));
    GS_HIP(hipMemset(len2.p, 0, (nb + 1) * 8));
    hipLaunchKernelGGL(k_synth_len2, dim3(grid_of(nb)), dim3(256), 0, 0, G, nb, snp_every, indel_every, max_indel, seed, len2.as<unsigned long long>());
    size_t tb = 0;
    GS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, len2.as<unsigned long long>(), off2.as<unsigned long long>(), (size_t)(nb + 1), 0));
    GS_HIP(tmp.alloc(tb));
    GS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, len2.as<unsigned long long>(), off2.as<unsigned long long>(), (size_t)(nb + 1), 0));
    unsigned long long H2 = 0;
    GS_HIP(hipMemcpy(&H2, off2.as<unsigned long long>() + nb, 8, hipMemcpyDeviceToHost));
    if (H2 <= read_len) { snprintf(g_err, sizeof g_err, "second haplotype shorter than a read"); return -1; }
    const uint64_t m = G + 1 + H2 + 1;
    void* text = nullptr;
    void* starts = nullptr;
    GS_HIP(hipMalloc(&text, m));
    if (hipMalloc(&starts, n_reads * 8) != hipSuccess) { (void)hipFree(text); snprintf(g_err, sizeof g_err, "out of device memory"); return -4; }
    uint8_t* t = static_cast<uint8_t*>(text);
    hipLaunchKernelGGL(k_synth_hap1, dim3(grid_capped(G)), dim3(256), 0, 0, t, G, seed);
    hipLaunchKernelGGL(k_synth_hap2, dim3((unsigned)nb), dim3(256), 0, 0, t, t + G + 1, G, snp_every, indel_every, max_indel, seed, off2.as<unsigned long long>(),
                       len2.as<unsigned long long>());
    const uint8_t seps[2] = {1, 2};
    GS_HIP(hipMemcpy(t + G, &seps[0], 1, hipMemcpyHostToDevice));
    GS_HIP(hipMemcpy(t + m - 1, &seps[1], 1, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_synth_starts, dim3(grid_of(n_reads)), dim3(256), 0, 0, static_cast<uint64_t*>(starts), n_reads, read_len, 0ull, G, G + 1, (uint64_t)H2, seed);
    GS_HIP(hipDeviceSynchronize());
    GS_HIP(hipGetLastError());
    *d_text = text;
    *text_len = m;
    *d_starts = starts;
    if (hap2_len) *hap2_len = H2;
    return 0;
}

// text / starts: host or device memory (text_on_device / starts_on_device); read r = text[starts[r] .. starts[r] + read_len), A/C/G/T only;
// any other byte of the text (separators) never lies inside a read.  read_len <= 254.  prefix_len = P (slabs), group_len = Pg <= P (units of
// the suffix sorting), < 0: chosen from the sizes.  reads [0, n_first) form collection 0 of the document array.
// Outputs (device memory, free with ebgen_free): *d_bwt = n_reads * (read_len + 1) ASCII symbols, *d_da ('0'/'1') when d_da != NULL.
// timings[5] = {pack + histograms, sorted starts, short blocks, group suffix sorting, slab keys + sort + emit} seconds; info[4] = {P, Pg, max rows of a group, max keys of a slab}.
// truth != NULL: also the three LCP bit vectors of the collection by their definition (thr_k / thr_r >= P), each (n + 63) / 64 + 1 words of device
// memory (truth[0..2] = thrK, thrR, minima; free with ebgen_free).  minima is not decided for the first two rows and the last row of every slab
// and short block: those rows are listed in unknown_rows (host, up to unknown_cap; *n_unknown = how many) and left 0.
int ebgen_build_slabbed(const void* text, uint64_t text_len, int text_on_device, const void* starts, uint64_t n_reads, int starts_on_device, uint32_t read_len,
                        uint8_t term, uint64_t n_first, int prefix_len, int group_len, int device, void** d_bwt, void** d_da, uint64_t* n_out, double* timings,
                        uint64_t* info, int thr_k, int thr_r, void** truth, uint64_t* unknown_rows, uint64_t unknown_cap, uint64_t* n_unknown) {
    if (!text || !starts || !d_bwt || !n_out || n_reads == 0 || read_len == 0 || read_len > 254 || text_len < read_len || n_reads >= (1ull << 32) ||
        text_len >= (1ull << 40)) {
        snprintf(g_err, sizeof g_err, "bad arguments");
        return -1;
    }
    GS_HIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    const uint64_t m = text_len;
    const uint32_t L = read_len;
    const uint64_t n = n_reads * (uint64_t)(L + 1);
    const uint32_t rbits = bits_for(n_reads);
    const uint32_t lo_bits = 64 - 8 - rbits;
    const double t0 = now_s();

    // ---- packed text
    const uint64_t n_words = (m + PAD) / 32 + 3;
    Buf d_t2, d_starts;
    GS_HIP(d_t2.alloc(n_words * 8));
    {
        Buf d_bytes;
        const uint8_t* src = static_cast<const uint8_t*>(text);
        if (!text_on_device) {
            GS_HIP(d_bytes.alloc(m + 32));
            GS_HIP(hipMemcpy(d_bytes.p, text, m, hipMemcpyHostToDevice));
            src = d_bytes.as<uint8_t>();
        }
        hipLaunchKernelGGL(k_pack2, dim3(grid_of(n_words)), dim3(256), 0, st, src, m, n_words, d_t2.as<uint64_t>());
        GS_HIP(hipDeviceSynchronize());
    }
    if (starts_on_device) {
        GS_HIP(d_starts.alloc(n_reads * 8));
        GS_HIP(hipMemcpy(d_starts.p, starts, n_reads * 8, hipMemcpyDeviceToDevice));
    } else {
        const uint64_t* h = static_cast<const uint64_t*>(starts);
        for (uint64_t r = 0; r < n_reads; ++r)
            if (h[r] + L > m) { snprintf(g_err, sizeof g_err, "read %llu leaves the text", (unsigned long long)r); return -1; }
        GS_HIP(d_starts.alloc(n_reads * 8));
        GS_HIP(hipMemcpy(d_starts.p, starts, n_reads * 8, hipMemcpyHostToDevice));
    }

    // ---- sizes: rows per 6-mer of the text -> group and slab prefix lengths
    Buf d_hist, d_short;
    GS_HIP(d_hist.alloc(4096 * 8));
    GS_HIP(hipMemset(d_hist.p, 0, 4096 * 8));
    hipLaunchKernelGGL(k_text_hist, dim3(2048), dim3(256), 0, st, d_t2.as<uint64_t>(), m, 6u, d_hist.as<unsigned long long>());
    std::vector<unsigned long long> rows6(4096);
    GS_HIP(hipMemcpy(rows6.data(), d_hist.p, 4096 * 8, hipMemcpyDeviceToHost));
    auto max_rows = [&](uint32_t p) { // most text positions under one prefix of p <= 6 symbols
        unsigned long long best = 0;
        const uint32_t span = 1u << (2 * (6 - p));
        for (uint32_t x = 0; x < 4096; x += span) {
            unsigned long long s = 0;
            for (uint32_t y = 0; y < span; ++y) s += rows6[x + y];
            if (s > best) best = s;
        }
        return best;
    };
    const double keys_per_row = (double)n / (double)m;
    const uint64_t group_cap = 1ull << 28, key_cap = 1ull << 28;
    int Pg = group_len, P = prefix_len;
    if (Pg < 0) { Pg = 0; while (Pg < 6 && max_rows((uint32_t)Pg) > group_cap) ++Pg; }
    if (P < 0) {
        P = Pg;
        while (P < 6 && (max_rows((uint32_t)P) >= (1ull << (lo_bits < 31 ? lo_bits : 31)) || (double)max_rows((uint32_t)P) * keys_per_row * 1.25 > (double)key_cap)) ++P;
    }
    if (P > 6 || Pg > P || Pg < 0 || (uint32_t)P > L) { snprintf(g_err, sizeof g_err, "prefix lengths %d / %d out of range (group <= slab <= 6 and <= read length)", Pg, P); return -1; }
    if (max_rows((uint32_t)Pg) >= 0xFFFFFFF0ull || max_rows((uint32_t)P) >= (1ull << (lo_bits < 32 ? lo_bits : 32))) {
        snprintf(g_err, sizeof g_err, "a slab of %llu rows does not fit the key (%u bits for the row) or a group has 2^32 rows", max_rows((uint32_t)P), lo_bits);
        return -1;
    }
    if (truth && (thr_k < P || thr_r < P || thr_k < 1 || thr_r < 1 || thr_k > 254 || thr_r > 254 || !unknown_rows || !n_unknown)) {
        snprintf(g_err, sizeof g_err, "LCP thresholds %d / %d must lie in [max(1, slab prefix %d), 254]", thr_k, thr_r, P);
        return -1;
    }
    const uint32_t n_slabs = 1u << (2 * P), n_groups = 1u << (2 * Pg), slabs_per_group = n_slabs / n_groups;
    std::vector<unsigned long long> rows(n_slabs, 0);
    for (uint32_t x = 0; x < 4096; ++x) rows[x >> (2 * (6 - P))] += rows6[x];

    // ---- keys per slab and reads per short suffix
    const uint32_t n_short = P ? ((1u << (2 * P)) - 4) / 3 : 0;
    GS_HIP(d_short.alloc((size_t)(n_short + 1) * 8));
    GS_HIP(hipMemset(d_hist.p, 0, 4096 * 8));
    GS_HIP(hipMemset(d_short.p, 0, (size_t)(n_short + 1) * 8));
    hipLaunchKernelGGL(k_read_hist, dim3(2048), dim3(256), 0, st, d_t2.as<uint64_t>(), d_starts.as<uint64_t>(), n_reads, L, (uint32_t)P, d_hist.as<unsigned long long>(),
                       d_short.as<unsigned long long>());
    std::vector<unsigned long long> keys_of(n_slabs), short_cnt(n_short + 1);
    GS_HIP(hipMemcpy(keys_of.data(), d_hist.p, (size_t)n_slabs * 8, hipMemcpyDeviceToHost));
    GS_HIP(hipMemcpy(short_cnt.data(), d_short.p, (size_t)(n_short + 1) * 8, hipMemcpyDeviceToHost));
    GS_HIP(hipGetLastError());
    // output offsets: depth-first over the prefixes; W# sorts in front of everything that continues W
    std::vector<unsigned long long> slab_off(n_slabs), short_off(n_short + 1);
    unsigned long long pos = 0, max_keys = 0, max_group_rows = 0, max_slab_rows = 0;
    {
        struct Frame { uint32_t l, code, next; };
        std::vector<Frame> stack;
        uint64_t unk = 0;
        auto item = [&](unsigned long long first, unsigned long long count) { // rows whose LCP-minimum bit the generator does not decide
            if (!truth || !count) return;
            for (unsigned long long r : {first, first + 1, first + count - 1})
                if (r < first + count && !(r == first + 1 && count == 2)) { if (unk < unknown_cap) unknown_rows[unk] = r; ++unk; }
        };
        auto enter = [&](uint32_t l, uint32_t code) {
            if (l == (uint32_t)P) { slab_off[code] = pos; item(pos, keys_of[code]); pos += keys_of[code]; return; }
            if (l == 0) { item(pos, n_reads); pos += n_reads; } // the terminators' own suffixes "#": rows 0 .. n_reads-1, read order
            else { const uint32_t at = ((1u << (2 * l)) - 4) / 3 + code; short_off[at] = pos; item(pos, short_cnt[at]); pos += short_cnt[at]; }
            stack.push_back({l, code, 0});
        };
        enter(0, 0);
        while (!stack.empty()) {
            Frame& f = stack.back();
            if (f.next == 4) { stack.pop_back(); continue; }
            const uint32_t l = f.l + 1, code = f.code * 4 + f.next++;
            enter(l, code);
        }
        if (truth) {
            if (unk > unknown_cap) { snprintf(g_err, sizeof g_err, "%llu undecided rows do not fit unknown_rows[%llu]", (unsigned long long)unk, (unsigned long long)unknown_cap); return -1; }
            *n_unknown = unk;
        }
    }
    if (pos != n) { snprintf(g_err, sizeof g_err, "internal: %llu keys counted, %llu expected", pos, (unsigned long long)n); return -6; }
    for (uint32_t x = 0; x < n_slabs; ++x) { if (keys_of[x] > max_keys) max_keys = keys_of[x]; if (rows[x] > max_slab_rows) max_slab_rows = rows[x]; }
    for (uint32_t g = 0; g < n_groups; ++g) {
        unsigned long long s = 0;
        for (uint32_t x = 0; x < slabs_per_group; ++x) s += rows[g * slabs_per_group + x];
        if (s > max_group_rows) max_group_rows = s;
    }
    if (max_keys >= (1ull << 32)) { snprintf(g_err, sizeof g_err, "a slab of %llu keys: choose a longer prefix", max_keys); return -1; }
    if (info) { info[0] = (uint64_t)P; info[1] = (uint64_t)Pg; info[2] = max_group_rows; info[3] = max_keys; }
    const double t1 = now_s();

    // ---- reads by start position
    Buf d_ss, d_perm, d_C, d_tmp;
    const uint64_t cn = (m >> 6) + 2;
    size_t tmp_bytes = 0;
    {
        Buf d_iota, d_ss_in;
        GS_HIP(d_ss.alloc(n_reads * 8));
        GS_HIP(d_perm.alloc(n_reads * 4));
        GS_HIP(d_iota.alloc(n_reads * 4));
        hipLaunchKernelGGL(k_iota, dim3(grid_of(n_reads)), dim3(256), 0, st, d_iota.as<uint32_t>(), n_reads);
        size_t tb = 0;
        GS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_starts.as<uint64_t>(), d_ss.as<uint64_t>(), d_iota.as<uint32_t>(), d_perm.as<uint32_t>(), (size_t)n_reads, 0,
                                                   (int)bits_for(m + 1), st));
        GS_HIP(d_tmp.alloc(tb));
        tmp_bytes = tb;
        GS_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, d_starts.as<uint64_t>(), d_ss.as<uint64_t>(), d_iota.as<uint32_t>(), d_perm.as<uint32_t>(), (size_t)n_reads, 0,
                                                   (int)bits_for(m + 1), st));
        GS_HIP(d_C.alloc(cn * 4));
        hipLaunchKernelGGL(k_ctable, dim3(grid_of(cn)), dim3(256), 0, st, d_ss.as<uint64_t>(), n_reads, d_C.as<uint32_t>(), cn);
        GS_HIP(hipDeviceSynchronize());
        GS_HIP(hipGetLastError());
    }
    const double t2 = now_s();

    // ---- output
    void* out_bwt = nullptr;
    void* out_da = nullptr;
    GS_HIP(hipMalloc(&out_bwt, n));
    struct OutGuard { void** a; void** b; bool keep = false; ~OutGuard() { if (!keep) { if (*a) (void)hipFree(*a); if (*b) (void)hipFree(*b); } } } og{&out_bwt, &out_da};
    if (d_da) GS_HIP(hipMalloc(&out_da, n));
    uint8_t* bwt = static_cast<uint8_t*>(out_bwt);
    uint8_t* da = static_cast<uint8_t*>(out_da);
    const size_t truth_words = (size_t)((n + 63) / 64 + 1);
    void* tv[3] = {nullptr, nullptr, nullptr};
    struct TruthGuard { void** v; bool keep = false; ~TruthGuard() { if (!keep) for (int i = 0; i < 3; ++i) if (v[i]) (void)hipFree(v[i]); } } tg{tv};
    Buf d_rowlcp;
    if (truth) {
        for (auto& v : tv) { GS_HIP(hipMalloc(&v, truth_words * 8)); GS_HIP(hipMemset(v, 0, truth_words * 8)); }
        GS_HIP(d_rowlcp.alloc((size_t)max_keys + 8));
    }

    // ---- short blocks: suffixes of 0 .. P-1 symbols
    {
        Buf d_k, d_k2, d_first, d_off;
        GS_HIP(d_k.alloc(n_reads * 8));
        GS_HIP(d_k2.alloc(n_reads * 8));
        size_t tb = 0;
        GS_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_k.as<uint64_t>(), d_k2.as<uint64_t>(), (size_t)n_reads, 0, 64, st));
        if (tb > tmp_bytes) { GS_HIP(d_tmp.alloc(tb)); tmp_bytes = tb; }
        const uint32_t most = P > 1 ? 1u << (2 * (P - 1)) : 1u;
        GS_HIP(d_first.alloc((size_t)most * 8));
        GS_HIP(d_off.alloc((size_t)most * 8));
        std::vector<unsigned long long> first(most), off(most);
        for (uint32_t l = 0; l < (uint32_t)P; ++l) {
            const uint32_t codes = 1u << (2 * l);
            unsigned long long run = 0;
            for (uint32_t c = 0; c < codes; ++c) {
                first[c] = run;
                if (l == 0) { off[c] = 0; run += n_reads; }
                else { const uint32_t at = ((1u << (2 * l)) - 4) / 3 + c; off[c] = short_off[at]; run += short_cnt[at]; }
            }
            if (run != n_reads) { snprintf(g_err, sizeof g_err, "internal: short suffixes of %u symbols: %llu reads counted", l, run); return -6; }
            GS_HIP(hipMemcpy(d_first.p, first.data(), (size_t)codes * 8, hipMemcpyHostToDevice));
            GS_HIP(hipMemcpy(d_off.p, off.data(), (size_t)codes * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_short_keys, dim3(grid_of(n_reads)), dim3(256), 0, st, d_t2.as<uint64_t>(), d_starts.as<uint64_t>(), n_reads, L, l, rbits, d_k.as<uint64_t>());
            const uint64_t* sorted = d_k.as<uint64_t>();
            if (l) {
                GS_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, d_k.as<uint64_t>(), d_k2.as<uint64_t>(), (size_t)n_reads, (int)rbits, (int)(rbits + 2 * l), st));
                sorted = d_k2.as<uint64_t>();
            }
            hipLaunchKernelGGL(k_short_emit, dim3(grid_of(n_reads)), dim3(256), 0, st, sorted, n_reads, rbits, l, d_t2.as<uint64_t>(), d_starts.as<uint64_t>(), L,
                               d_first.as<unsigned long long>(), d_off.as<unsigned long long>(), n_first, bwt, da);
        }
        GS_HIP(hipDeviceSynchronize());
        GS_HIP(hipGetLastError());
    }
    const double t3 = now_s();

    // ---- groups and their slabs
    Buf d_key, d_key2, d_q, d_q2, d_lcp, d_tbl, d_cursor, d_err, d_cnt, d_koff, d_keys, d_keys2;
    GS_HIP(d_key.alloc((size_t)max_group_rows * 8 + 8));
    GS_HIP(d_key2.alloc((size_t)max_group_rows * 8 + 8));
    GS_HIP(d_q.alloc((size_t)max_group_rows * 8 + 8));
    GS_HIP(d_q2.alloc((size_t)max_group_rows * 8 + 8));
    GS_HIP(d_lcp.alloc((size_t)max_group_rows + 8));
    GS_HIP(d_tbl.alloc((size_t)(1u << TBL_BITS) * 4));
    GS_HIP(d_cursor.alloc(8));
    GS_HIP(d_err.alloc(4));
    GS_HIP(hipMemset(d_err.p, 0, 4));
    GS_HIP(d_cnt.alloc((size_t)max_slab_rows * 4 + 8));
    GS_HIP(d_koff.alloc((size_t)max_slab_rows * 4 + 8));
    GS_HIP(d_keys.alloc((size_t)max_keys * 8 + 8));
    GS_HIP(d_keys2.alloc((size_t)max_keys * 8 + 8));
    {
        size_t a = 0, b = 0, c = 0;
        GS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, a, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_q.as<uint64_t>(), d_q2.as<uint64_t>(), (size_t)max_group_rows, 0, 64, st));
        GS_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, b, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), (size_t)max_keys, 0, 64, st));
        GS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, c, d_cnt.as<uint32_t>(), d_koff.as<uint32_t>(), (size_t)max_slab_rows, st));
        const size_t need = a > b ? (a > c ? a : c) : (b > c ? b : c);
        if (need > tmp_bytes) { GS_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
    }
    double t_groups = 0, t_slabs = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        unsigned long long Mg = 0;
        for (uint32_t x = 0; x < slabs_per_group; ++x) Mg += rows[g * slabs_per_group + x];
        if (Mg == 0) continue;
        const double tg = now_s();
        GS_HIP(hipMemsetAsync(d_cursor.p, 0, 8, st));
        hipLaunchKernelGGL(k_collect, dim3(grid_of((m + 31) / 32)), dim3(256), 0, st, d_t2.as<uint64_t>(), m, (uint32_t)Pg, g, d_cursor.as<unsigned long long>(),
                           d_key.as<uint64_t>(), d_q.as<uint64_t>());
        GS_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_q.as<uint64_t>(), d_q2.as<uint64_t>(), (size_t)Mg, 0, 64, st));
        const uint64_t* skey = d_key2.as<uint64_t>();
        uint64_t* sq = d_q2.as<uint64_t>();
        hipLaunchKernelGGL(k_tiefix, dim3(grid_of(Mg)), dim3(256), 0, st, skey, sq, (uint64_t)Mg, d_t2.as<uint64_t>(), (uint32_t)Pg, L, d_err.as<uint32_t>());
        hipLaunchKernelGGL(k_lcp, dim3(grid_of(Mg)), dim3(256), 0, st, sq, (uint64_t)Mg, d_t2.as<uint64_t>(), d_lcp.as<uint8_t>());
        GS_HIP(hipMemsetAsync(d_tbl.p, 0xFF, (size_t)(1u << TBL_BITS) * 4, st));
        hipLaunchKernelGGL(k_tbl_heads, dim3(grid_of(Mg)), dim3(256), 0, st, skey, (uint64_t)Mg, d_tbl.as<uint32_t>());
        hipLaunchKernelGGL(k_tbl_fill, dim3(1), dim3(1024), 0, st, d_tbl.as<uint32_t>(), (uint32_t)Mg);
        unsigned long long collected = 0;
        uint32_t err = 0;
        GS_HIP(hipMemcpyAsync(&collected, d_cursor.p, 8, hipMemcpyDeviceToHost, st));
        GS_HIP(hipMemcpyAsync(&err, d_err.p, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        GS_HIP(hipGetLastError());
        if (collected != Mg) { snprintf(g_err, sizeof g_err, "internal: group %u holds %llu text positions, %llu expected", g, collected, Mg); return -6; }
        if (err) { snprintf(g_err, sizeof g_err, "the text repeats itself: more than %u positions equal on %d symbols (not a random genome)", TIE_CAP, Pg + 32); return -7; }
        t_groups += now_s() - tg;
        const double ts = now_s();
        unsigned long long ra = 0;
        for (uint32_t x = 0; x < slabs_per_group; ++x) {
            const uint32_t X = g * slabs_per_group + x;
            const unsigned long long M = rows[X], K = keys_of[X];
            if (M && K) {
                hipLaunchKernelGGL(k_count, dim3(grid_of(M)), dim3(256), 0, st, sq + ra, (uint64_t)M, d_ss.as<uint64_t>(), d_C.as<uint32_t>(), L, (uint32_t)P, d_cnt.as<uint32_t>());
                GS_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_cnt.as<uint32_t>(), d_koff.as<uint32_t>(), (size_t)M, st));
                uint32_t last_off = 0, last_cnt = 0;
                GS_HIP(hipMemcpyAsync(&last_off, d_koff.as<uint32_t>() + (M - 1), 4, hipMemcpyDeviceToHost, st));
                GS_HIP(hipMemcpyAsync(&last_cnt, d_cnt.as<uint32_t>() + (M - 1), 4, hipMemcpyDeviceToHost, st));
                GS_HIP(hipStreamSynchronize(st));
                if ((unsigned long long)last_off + last_cnt != K) {
                    snprintf(g_err, sizeof g_err, "internal: slab %u: %llu keys from the reads, %llu from the text positions", X, K, (unsigned long long)last_off + last_cnt);
                    return -6;
                }
                hipLaunchKernelGGL(k_fill, dim3(grid_of(M)), dim3(256), 0, st, sq + ra, skey + ra, d_lcp.as<uint8_t>() + ra, (uint64_t)M, d_ss.as<uint64_t>(), d_perm.as<uint32_t>(),
                                   d_C.as<uint32_t>(), d_tbl.as<uint32_t>(), (uint32_t)ra, L, (uint32_t)P, (uint32_t)Pg, rbits, d_koff.as<uint32_t>(), d_keys.as<uint64_t>());
                const int key_bits = (int)(bits_for(M + 1) + 8 + rbits);
                GS_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), (size_t)K, 0, key_bits > 64 ? 64 : key_bits, st));
                hipLaunchKernelGGL(k_emit, dim3(grid_capped(K)), dim3(256), 0, st, d_keys2.as<uint64_t>(), (uint64_t)K, rbits, d_t2.as<uint64_t>(), d_starts.as<uint64_t>(), L, term,
                                   n_first, bwt + slab_off[X], da ? da + slab_off[X] : nullptr);
                if (truth) {
                    hipLaunchKernelGGL(k_row_lcp, dim3(grid_of(K)), dim3(256), 0, st, d_keys2.as<uint64_t>(), (uint64_t)K, rbits, d_lcp.as<uint8_t>() + ra, d_rowlcp.as<uint8_t>());
                    const uint64_t words = ((slab_off[X] + K - 1) >> 6) - (slab_off[X] >> 6) + 1;
                    hipLaunchKernelGGL(k_truth_bits, dim3(grid_of(words)), dim3(256), 0, st, d_rowlcp.as<uint8_t>(), (uint64_t)K, (uint64_t)slab_off[X], (uint32_t)thr_k, (uint32_t)thr_r,
                                       static_cast<unsigned long long*>(tv[0]), static_cast<unsigned long long*>(tv[1]), static_cast<unsigned long long*>(tv[2]));
                }
            } else if (K) {
                snprintf(g_err, sizeof g_err, "internal: slab %u has keys but no text position", X);
                return -6;
            }
            ra += M;
        }
        GS_HIP(hipStreamSynchronize(st));
        GS_HIP(hipGetLastError());
        t_slabs += now_s() - ts;
    }
    if (timings) { timings[0] = t1 - t0; timings[1] = t2 - t1; timings[2] = t3 - t2; timings[3] = t_groups; timings[4] = t_slabs; }
    og.keep = true;
    if (truth) { tg.keep = true; for (int i = 0; i < 3; ++i) truth[i] = tv[i]; }
    *d_bwt = out_bwt;
    if (d_da) *d_da = out_da;
    *n_out = n;
    return 0;
}

}  // extern "C"
