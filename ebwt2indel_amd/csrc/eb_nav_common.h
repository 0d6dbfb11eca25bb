// eb_nav_common.h -- records, their packed stack form, counters and launch descriptors of the traversal kernels, and the pieces
// the chain-following kernels share (included by eb_navigate.hip only; everything lives in its anonymous namespace).
#pragma once
#include "eb_internal.h"

namespace {


// Records are padded to a multiple of 16 bytes so that lanes move them as uint4.
struct alignas(16) Leaf1 { uint64_t l, r, depth, pad; };                              // sa_leaf (include.hpp:513-527)
struct alignas(16) Node1 { uint64_t b[6]; uint64_t depth, pad; };                     // sa_node (include.hpp:394-413)
struct alignas(16) Leaf2 { uint64_t l1, r1, l2, r2, depth, pad[3]; };                 // pair<sa_leaf,sa_leaf>
struct alignas(16) Node2 { uint64_t a[6]; uint64_t b[6]; uint64_t depth, pad[3]; };   // pair<sa_node,sa_node>

// Records travel through the stacks packed: NF 40-bit positions (n < 2^40, checked by the host) and a 16-bit depth that
// saturates -- the depth is only ever compared with K and k_right (< 65535, checked by the host).  A one-BWT node is
// 32 bytes instead of 64, a two-BWT node 64 instead of 128, a two-BWT leaf 32 instead of 64: half the record traffic of
// the traversal and twice the records per arena.
template <int NF>
struct alignas(16) Packed { uint64_t w[((NF * 40 + 16 + 127) / 128) * 2]; };
template <int NF>
__host__ __device__ __forceinline__ Packed<NF> pack_fields(const uint64_t* f, uint64_t depth) {
    static_assert(((NF * 40) & 63) <= 48, "the depth must not straddle two words");
    Packed<NF> p;
    for (uint64_t& x : p.w) x = 0;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int off = 40 * k, wd = off >> 6, sh = off & 63;
        const uint64_t v = f[k] & ((1ull << 40) - 1ull);
        p.w[wd] |= v << sh;
        if (sh > 24) p.w[wd + 1] |= v >> (64 - sh);
    }
    p.w[(NF * 40) >> 6] |= (depth > 0xFFFFull ? 0xFFFFull : depth) << ((NF * 40) & 63);
    return p;
}
template <int NF>
__host__ __device__ __forceinline__ uint64_t unpack_fields(const Packed<NF>& p, uint64_t* f) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int off = 40 * k, wd = off >> 6, sh = off & 63;
        uint64_t v = p.w[wd] >> sh;
        if (sh > 24) v |= p.w[wd + 1] << (64 - sh);
        f[k] = v & ((1ull << 40) - 1ull);
    }
    return (p.w[(NF * 40) >> 6] >> ((NF * 40) & 63)) & 0xFFFFull;
}
using Node1P = Packed<6>;
using Node2P = Packed<12>;
using Leaf2P = Packed<4>;
static_assert(sizeof(Node1P) == 32 && sizeof(Node2P) == 64 && sizeof(Leaf2P) == 32, "packed record sizes");
__host__ __device__ __forceinline__ Node1P pack_rec(const Node1& n) { return pack_fields<6>(n.b, n.depth); }
__host__ __device__ __forceinline__ Node2P pack_rec(const Node2& n) { return pack_fields<12>(n.a, n.depth); }   // a[6] and b[6] are contiguous
__host__ __device__ __forceinline__ Leaf2P pack_rec(const Leaf2& l) { return pack_fields<4>(&l.l1, l.depth); }
__host__ __device__ __forceinline__ const Leaf1& pack_rec(const Leaf1& l) { return l; }                            // 32 bytes as it is
__host__ __device__ __forceinline__ void unpack_rec(const Node1P& p, Node1& n) { n.depth = unpack_fields<6>(p, n.b); n.pad = 0; }
__host__ __device__ __forceinline__ void unpack_rec(const Node2P& p, Node2& n) { n.depth = unpack_fields<12>(p, n.a); n.pad[0] = n.pad[1] = n.pad[2] = 0; }
__host__ __device__ __forceinline__ void unpack_rec(const Leaf2P& p, Leaf2& l) { l.depth = unpack_fields<4>(p, &l.l1); l.pad[0] = l.pad[1] = l.pad[2] = 0; }
__host__ __device__ __forceinline__ void unpack_rec(const Leaf1& p, Leaf1& l) { l = p; }

struct NavCounters {
    unsigned long long pushed;     // children appended by the current launch
    unsigned long long pad0[7];    // keep the hot word on its own 64-byte line
    unsigned long long taken;      // k_nodes1c: input records handed out so far by the current launch (reset with `pushed`)
    unsigned long long pad1[7];
    unsigned long long leaves, nodes, lcp_values, da_values, minima, ranks, pos_sum, pos_sq_sum;
};

struct NavParams {
    int K, k_right;
    int write;     // 0: expand only (the shared top of the tree on shards other than 0): no bit writes, no statistics
    uint8_t* lcpb; // one byte per position (bit 0: LCP >= K, bit 1: LCP >= k_right) or null -> interleaved bit words `kr`
    // below != 0: the COMPLEMENT is collected -- a writer marks its positions iff its depth is BELOW the threshold, and the
    // vectors are inverted once at the end.  Every position 1..n-1 has exactly one writer (a node border or a leaf
    // interior; lcp_values reaches n, ebwt2InDel.cpp:617,670) and with reads of 100+ bases nearly all writers lie deeper
    // than K and k_right: marking the few shallow ones instead of the many deep ones removes ~95 % of the scattered
    // stores of the traversal (phases 2-3 at 7.55 G positions: 0.58 -> 0.47 s).
    int below;
#ifdef EB_SCAN_EXPERIMENTS
    uint32_t dbg;  // EB_NAV_DEBUG (profiling build only; results are wrong): 1 no LCP stores, 2 no minima, 4 children not appended below depth 40, 8 no DA bits of singleton rows (two BWTs; phase 4 then refuses the flags)
#endif
};
#ifdef EB_SCAN_EXPERIMENTS
#define EB_NDBG(p) ((p).dbg)
#else
#define EB_NDBG(p) 0u
#endif

// What one expansion launch works on.  It lives in DEVICE memory and is written by k_plan (below), so that the host can enqueue
// several launches back to back without knowing how many children the previous one pushed.
struct NavLaunch {
    const void* in;    // `count` records, popped from the top of the source stack (expanded where they lie)
    uint64_t count;
    void* out;         // children are appended here through NavCounters::pushed ...
    uint64_t out_cap;  // ... up to this many
    int32_t write;     // 0: expand only (the shared top of the tree on shards other than 0): no bit writes, no statistics
    int32_t follow;    // k_nodes1c: 1 = a node's first kept child is expanded by the same quad instead of being pushed (0 while a
                       // sharded traversal still walks level by level towards its cut)
};

// Power sums of the LCP positions written, modulo 2^64 (eb_nav_stats.lcp_pos_sum / lcp_pos_sq_sum).  The count alone
// (lcp_values == n) cannot tell "every position 1..n-1 once" from "one twice, one never" -- which the complement form of the
// thresholds would turn into a wrong bit, and atomicOr hides duplicates -- while sum and sum of squares do (round-2 advisor
// finding).  Ranges (leaf interiors) use the closed forms: tri(x) = sum of i < x, sq(x) = sum of i^2 < x.
__host__ __device__ __forceinline__ uint64_t eb_tri(uint64_t x) { return (x & 1ull) ? x * ((x - 1) >> 1) : (x >> 1) * (x - 1); }
__host__ __device__ __forceinline__ uint64_t eb_sq(uint64_t x) {
    // (x-1) x (2x-1) / 6: the halving is exact on one factor; the product is a multiple of 3, and 3 is invertible modulo 2^64
    return eb_tri(x) * (2 * x - 1) * 0xAAAAAAAAAAAAAAABull;
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((unsigned long long)v, o);
    return v;
}

// Statistics: every workgroup owns one row of counters in HBM and adds its totals with plain read-modify-writes
// (launches are sequential; rows are summed on the host at the end) -- no per-wave atomics on shared words.
#define EB_NAV_STATS 8 /* leaves, nodes, lcp_values, da_values, minima, ranks, sum of the LCP positions written, sum of their squares */
__device__ __forceinline__ void flush_stats(unsigned long long* rows, const uint64_t* v /* EB_NAV_STATS values of this lane */) {
    __shared__ unsigned long long sh[EB_NAV_STATS][4];
    for (int k = 0; k < EB_NAV_STATS; ++k) {
        uint64_t s = wave_sum(v[k]);
        if ((threadIdx.x & 63u) == 0) sh[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < EB_NAV_STATS) {
        unsigned long long* p = rows + (uint64_t)blockIdx.x * EB_NAV_STATS + threadIdx.x;
        *p += sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    }
}

__device__ __forceinline__ uint64_t pick4(const uint64_t* F, uint32_t c) { return c == 0 ? F[0] : c == 1 ? F[1] : c == 2 ? F[2] : F[3]; }

// ---- pieces shared by the chain-following kernels (k_nodes1c, k_leaves1c, k_nodes2c, k_leaves2c)
// The launch's input records are handed out through a cursor (NavCounters::taken): a wave takes a chunk with ONE returning atomic,
// its quads take records from the chunk as their chains end.  Everything here is wave-uniform except `want` / `idx`.
struct NavCursor {
    uint64_t next = 0, end = 0; // this wave's chunk
    uint64_t chunk, count;
    bool dry = false;           // the input is used up
    // records per chunk: 16 (one per quad) for small launches, up to 256
    __device__ __forceinline__ NavCursor(uint64_t count_, uint32_t grid) : count(count_) {
        const uint64_t waves = (count + 15) / 16 < (uint64_t)grid * 4 ? (count + 15) / 16 : (uint64_t)grid * 4;
        chunk = count / (waves * 4);
        chunk = chunk < 16 ? 16 : (chunk > 256 ? 256 : chunk);
    }
    // hands input records to the quads that ask for one (`want`, the same in the four lanes of a quad): idx = the record's index,
    // want stays set where the input ran out
    __device__ __forceinline__ void take(NavCounters* cnt, bool& want, uint64_t& idx, uint32_t j, uint32_t lane) {
        for (;;) {
            const unsigned long long m = __ballot(want && j == 0);
            if (!m || dry) break;
            if (next == end) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&cnt->taken, (unsigned long long)chunk);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
                next = base < count ? base : count;
                end = base + chunk < count ? base + chunk : count;
                if (next == end) { dry = true; break; }
            }
            const uint64_t avail = end - next;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << (lane & ~3u)) - 1ull)); // askers in the quads below this one
            if (want && rank < avail) { idx = next + rank; want = false; }
            const uint64_t asked = (uint64_t)__popcll(m);
            next += asked < avail ? asked : avail;
        }
    }
};
// A kept child that does not continue in registers: into the small staging buffer, or -- buffer full: the bushy top of the tree --
// straight to the stack with an atomic of its own.  `make()` packs the record; it is called only when some lane of the wave pushes.
template <typename Rec, int STAGE, typename Make>
__device__ __forceinline__ void nav_push_or_spill(bool px, Make make, uint4* s_buf, uint32_t* s_count, Rec* out, uint64_t out_cap, NavCounters* cnt, uint32_t lane) {
    const unsigned long long pm = __ballot(px);
    if (!pm) return; // (uniform; most turns nobody pushes)
    const Rec rec = make();
    uint32_t wbase = 0;
    if (lane == 0) wbase = atomicAdd(s_count, (uint32_t)__popcll(pm));
    wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
    if (px) {
        constexpr int W = sizeof(Rec) / 16;
        const uint32_t slot = wbase + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
        const uint4* src = reinterpret_cast<const uint4*>(&rec);
        if (slot < (uint32_t)STAGE) {
            for (int q = 0; q < W; ++q) s_buf[slot * W + q] = src[q];
        } else {
            const unsigned long long g = atomicAdd(&cnt->pushed, 1ull);
            if (g < out_cap)
                for (int q = 0; q < W; ++q) eb_store_global(reinterpret_cast<uint4*>(out) + g * W + q, src[q]);
        }
    }
}
// End of a round of turns: the staging buffer is emptied, and the workgroup goes on while any of its waves has work.
template <typename Rec, int STAGE>
__device__ __forceinline__ bool nav_round_end(Stager<Rec, STAGE>& stage, uint32_t* s_count, int* s_alive, Rec* out, uint64_t out_cap, NavCounters* cnt, bool wave_has_work, uint32_t lane) {
    __syncthreads();
    if (threadIdx.x == 0 && *s_count > (uint32_t)STAGE) *s_count = STAGE; // (the excess went straight to the stack)
    stage.flush(out, out_cap, &cnt->pushed);
    if (threadIdx.x == 0) *s_alive = 0;
    __syncthreads();
    if (lane == 0 && wave_has_work) *s_alive = 1;
    __syncthreads();
    const int go_on = *s_alive;
    __syncthreads(); // (nobody resets the flag before everybody has read it)
    return go_on != 0;
}


}  // namespace
