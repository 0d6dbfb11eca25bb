// eb_navigate.hip -- phases 2-3 on the GPU: LCP-threshold / LCP-minima / DA bitvectors.
//
// Replaces navigate_one_bwt (ebwt2InDel.cpp:555-676) and navigate_two_bwts (:679-831).
// The reference walks suffix-tree leaves and right-maximal nodes depth-first with an explicit
// stack, one record at a time.  Every LCP / DA slot has exactly one writer, so the visiting order
// is free; here the stack lives in HBM and is drained in batches: a launch pops up to `batch`
// records from the top, one lane expands one record (<= 6 parallel_rank queries per BWT), and the
// surviving children are pushed back with one wave-aggregated atomic reservation.  Popping from
// the top keeps the footprint of a depth-first walk; a batch of millions keeps HBM busy with
// independent 64-byte block reads.  Bit writes are atomicOr on 64-bit words and are skipped when
// the value is 0 (vectors start zeroed), i.e. for records shallower than K / k_right.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>
#include <type_traits>

#include "eb_config.h"
#include "eb_internal.h"
#include "eb_pieces.h"

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

// ---- phase 2: leaves.  update_LCP_leaf (ebwt2InDel.cpp:344-355) / update_DA (:394-425) and next_leaves
// (dna_bwt.hpp:358-379; two BWTs: ebwt2InDel.cpp:452-472).  Four lanes per leaf: lane j owns letter j --
// it loads the j-th chunk of each rank query and builds the child leaf obtained by prepending "ACGT"[j].
template <int NB>
__global__ void __launch_bounds__(256) k_leaves(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p,
                                               const NavLaunch* __restrict__ la, NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Leaf = typename std::conditional<NB == 1, Leaf1, Leaf2>::type;
    using Rec = typename std::conditional<NB == 1, Leaf1, Leaf2P>::type; // the record as it lies in the stacks
    constexpr int STAGE = 1024;
    const void* in_ = la->in;
    void* out_ = la->out;
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    if ((uint64_t)blockIdx.x * 64 >= count) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const Rec* in = static_cast<const Rec*>(in_);
    Rec* out = static_cast<Rec*>(out_);
    const uint32_t j = threadIdx.x & 3u, q = threadIdx.x >> 2;
    uint64_t n_leaf = 0, n_lcp = 0, n_da = 0, n_rank = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t tiles = (count + 63) / 64;
    uint32_t trips = 0;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint64_t i = tile * 64 + q;
        bool active = i < count;
        Leaf L;
        memset(&L, 0, sizeof L);
        if (active) unpack_rec(eb_load_global(in + i), L);
        Leaf child;
        memset(&child, 0, sizeof child);
        bool keep = false;
        if constexpr (NB == 1) {
            uint64_t a = eb_rank4_quad(x1, L.l, j);
            uint64_t b = eb_rank4_quad(x1, L.r > L.l ? L.r : L.l, j);
            uint64_t Fj = pick4(x1.F, j);
            child.l = Fj + a; child.r = Fj + b; child.depth = L.depth + 1;
            keep = active && (b - a >= 2);
            if (active && p.write && L.r > L.l + 1) {
                // LCP slots l+1 .. r-1 of the leaf: the quad splits the touched words
                uint64_t lo = L.l + 1, hi = L.r;
                bool setK = ((int64_t)L.depth >= p.K) != (p.below != 0), setR = ((int64_t)L.depth >= p.k_right) != (p.below != 0);
                if ((setK || setR) && p.lcpb) {
                    uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                    for (uint64_t i2 = lo + j; i2 < hi; i2 += 4) p.lcpb[i2] = v; // every position has one writer: plain byte stores
                } else if (setK || setR) {
                    uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
                    for (uint64_t w = w0 + j; w <= w1; w += 4) {
                        uint64_t m = ~0ull;
                        if (w == w0) m &= ~0ull << (lo & 63u);
                        if (w == w1) m &= ~0ull >> (63u - ((hi - 1) & 63u));
                        if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w), m);
                        if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w + 1), m);
                    }
                }
                if (j == 0) { n_lcp += hi - lo; n_s1 += eb_tri(hi) - eb_tri(lo); n_s2 += eb_sq(hi) - eb_sq(lo); }
            }
            if (active && p.write && j == 0) { ++n_leaf; n_rank += L.r > L.l ? 2 : 1; }
        } else {
            uint64_t a1 = eb_rank4_quad(x1, L.l1, j), b1 = eb_rank4_quad(x1, L.r1 > L.l1 ? L.r1 : L.l1, j);
            uint64_t a2 = eb_rank4_quad(x2, L.l2, j), b2 = eb_rank4_quad(x2, L.r2 > L.l2 ? L.r2 : L.l2, j);
            uint64_t F1 = pick4(x1.F, j), F2 = pick4(x2.F, j);
            child.l1 = F1 + a1; child.r1 = F1 + b1; child.l2 = F2 + a2; child.r2 = F2 + b2; child.depth = L.depth + 1;
            keep = active && ((b1 - a1) + (b2 - a2) >= 2);
            if (active && p.write) {
                uint64_t s1 = L.l1 + L.l2, s2 = L.l2 + L.r1, e = L.r1 + L.r2;
                bool setK = ((int64_t)L.depth >= p.K) != (p.below != 0), setR = ((int64_t)L.depth >= p.k_right) != (p.below != 0);
                // words of [s1, e): DA = 1 on [s2,e) (second collection), LCP bits on [s1+1, e)
                uint64_t w0 = s1 >> 6, w1 = e > s1 ? (e - 1) >> 6 : w0;
                for (uint64_t w = w0 + j; w <= w1 && e > s1; w += 4) {
                    uint64_t base = w << 6;
                    auto span = [&](uint64_t lo, uint64_t hi) -> uint64_t { // bits of word w inside [lo,hi)
                        if (hi <= lo || hi <= base || lo >= base + 64) return 0ull;
                        uint64_t m = ~0ull;
                        if (lo > base) m &= ~0ull << (lo - base);
                        if (hi < base + 64) m &= ~0ull >> (base + 64 - hi);
                        return m;
                    };
                    uint64_t md = span(s2, e), ml = span(s1 + 1, e);
                    if (md) atomicOr(reinterpret_cast<unsigned long long*>(f.da + w), md);
                    if (!p.lcpb) {
                        if (ml && setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w), ml);
                        if (ml && setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w + 1), ml);
                    }
                }
                if (p.lcpb && (setK || setR)) {
                    uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                    for (uint64_t i2 = s1 + 1 + j; i2 < e; i2 += 4) p.lcpb[i2] = v;
                }
                if (j == 0) {
                    ++n_leaf;
                    n_da += e - s1;
                    if (e > s1 + 1) { n_lcp += e - s1 - 1; n_s1 += eb_tri(e) - eb_tri(s1 + 1); n_s2 += eb_sq(e) - eb_sq(s1 + 1); }
                    n_rank += (L.r1 > L.l1 ? 2 : 1) + (L.r2 > L.l2 ? 2 : 1);
                }
            }
        }
        stage.push(keep, pack_rec(child));
        // every tile adds at most 256 records: flushing every STAGE/256 tiles never overflows the buffer, and the
        // decision is the same in every thread without looking at shared state
        if (++trips % (STAGE / 256) == 0) stage.flush(out, out_cap, &cnt->pushed);
    }
    stage.flush(out, out_cap, &cnt->pushed);
    const uint64_t st[EB_NAV_STATS] = {n_leaf, 0, n_lcp, n_da, 0, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// ---- phase 3, one BWT, software-pipelined (round 3).  One node per quad and trip (four lanes per node: lane j loads the j-th chunk
// of the six boundary blocks, owns letter j -- the Weiner-link child "ACGT"[j] + W -- and writes border j's LCP / minimum bits), and the
// memory operations of the NEXT trips are under way while a trip is worked on:
//   stage A  the record of trip t+1 (loaded during trip t-1) is unpacked; its LCP / minimum bits are set at once -- they
//            depend on the record alone (borders and depth), not on any rank -- and its six block loads are issued;
//   stage C  the record of trip t+2 is requested;
//   stage D  trip t: its six blocks have landed meanwhile (vector-memory operations complete in issue order, so the wait for
//            record t+1 in stage A covered them); ranks, the four Weiner-link children, staging.
// k_nodes issued the record load, waited, issued the block loads, waited, computed: two exposed memory latencies per trip with
// three waves per SIMD to cover them (49 KB of staging per workgroup), ~6.8 us per trip for ~1 us of arithmetic.  Here the only
// vector-memory operations inside stage D are the staging flush's, so nothing in D waits for the loads issued in A / C.  The
// superblock table (a global load per rank in eb_rank4_quad_finish) is read from LDS for that reason.
#define EB_NAV_SUPER_LDS 40 /* superblocks held in LDS (n <= 171 G symbols); beyond that the table is read from memory */
template <bool SUPER_LDS> // the superblock table fits EB_NAV_SUPER_LDS entries of LDS (a select between an LDS and a global address would become flat loads)
__global__ void __launch_bounds__(256) k_nodes1p(eb_dev_index x1, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p, const NavLaunch* __restrict__ la,
                                                NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Node1P;
    constexpr int STAGE = 1536;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_super[EB_NAV_SUPER_LDS * 4];
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const uint64_t tiles = (count + 63) / 64;
    if ((uint64_t)blockIdx.x >= tiles) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (SUPER_LDS) {
        const uint64_t n_super = (x1.n >> EB_SUPER_SHIFT) + 1;
        for (uint32_t k = threadIdx.x; k < n_super * 4; k += blockDim.x) s_super[k] = x1.super[k];
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, q = threadIdx.x >> 2;
    const uint64_t G = gridDim.x;
    uint64_t n_node = 0, n_lcp = 0, n_rank = 0, n_min = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t Fj = pick4(x1.F, j);

    auto load_rec = [&](uint64_t tile) -> Rec {
        Rec r;
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
        const uint64_t i = tile * 64 + q;
        if (tile < tiles && i < count) r = eb_load_global(in + i);
        return r;
    };
    // stage A for one record: border bits (update_lcp_threshold include.hpp:826-860, update_lcp_minima ebwt2InDel.cpp:357-391)
    auto mark = [&](const Node1& N) {
        uint64_t lo = 0, hi = 0;
        for (int k = 0; k < 4; ++k) if (j == (uint32_t)k) { lo = N.b[k]; hi = N.b[k + 1]; }
        if (hi > lo && hi != N.b[5]) {
            const bool setK = ((int64_t)N.depth >= p.K) != (p.below != 0), setR = ((int64_t)N.depth >= p.k_right) != (p.below != 0);
            if (p.lcpb) {
                uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                if (v && !(EB_NDBG(p) & 1u)) p.lcpb[hi] = v;
            } else {
                if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6)), 1ull << (hi & 63u));
                if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6) + 1), 1ull << (hi & 63u));
            }
            ++n_lcp;
            n_s1 += hi;
            n_s2 += hi * hi;
        }
        if (j >= 1 && hi - lo >= 2 && hi < N.b[5] - 1) {
            if (!(EB_NDBG(p) & 2u)) eb_bit_set_atomic(f.minima, hi);
            ++n_min;
        }
    };

    uint64_t tile = blockIdx.x;
    Node1 Nn;          // the node of the next trip (stage A done)
    uint4 cn[6];       // its six blocks, in flight
    bool act_n;
    {
        Rec r0 = load_rec(tile);
        unpack_rec(r0, Nn);
        act_n = tile * 64 + q < count;
        if (act_n && p.write) mark(Nn);
#pragma unroll
        for (int k = 0; k < 6; ++k) cn[k] = eb_rank4_quad_load(x1, Nn.b[k], j);
    }
    Rec rr = load_rec(tile + G); // stage C of the prologue
    uint32_t trips = 0;
    for (; tile < tiles; tile += G) {
        const Node1 Nc = Nn;
        const bool act_c = act_n;
        uint4 cc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) cc[k] = cn[k];
        // stage A: the next trip's record -> node, border bits, block loads
        unpack_rec(rr, Nn);
        act_n = tile + G < tiles && (tile + G) * 64 + q < count;
        if (act_n && p.write) mark(Nn);
#pragma unroll
        for (int k = 0; k < 6; ++k) cn[k] = eb_rank4_quad_load(x1, Nn.b[k], j);
        // stage C: the record after that
        rr = load_rec(tile + 2 * G);
        // stage D: this trip's ranks and children (LF(sa_node) dna_bwt.hpp:323-356, next_nodes :381-404)
        Node1 child;
        child.pad = 0;
        int kids = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint64_t i = Nc.b[k];
            uint32_t pc = eb_chunk_counts(cc[k], eb_prefix_mask32((uint32_t)(i & 127u), j));
            pc = eb_quad_sum(pc);
            uint64_t sup;
            if constexpr (SUPER_LDS) sup = s_super[((i >> EB_SUPER_SHIFT) << 2) + j];
            else sup = x1.super[((i >> EB_SUPER_SHIFT) << 2) + j];
            child.b[k] = Fj + sup + cc[k].w + ((pc >> (8u * j)) & 0xFFu);
        }
        for (int k = 0; k < 5; ++k) kids += child.b[k + 1] > child.b[k];
        child.depth = Nc.depth + 1;
        const bool keep = act_c && kids >= 2; // number_of_children (include.hpp:760-768)
        if (act_c && p.write && j == 0) {
            ++n_node;
            n_rank += 1;
            for (int k = 1; k < 6; ++k) n_rank += Nc.b[k] != Nc.b[k - 1]; // the reference shares the query of equal neighbours
        }
        stage.push(keep, pack_rec(child));
        // every trip adds at most 256 records: flushing every STAGE / 256 trips never overflows the buffer, and the decision is
        // the same in every thread without looking at shared state
        if (++trips % (STAGE / 256) == 0) stage.flush(out, out_cap, &cnt->pushed);
    }
    stage.flush(out, out_cap, &cnt->pushed);
    const uint64_t st[EB_NAV_STATS] = {0, n_node, n_lcp, 0, n_min, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

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

// ---- phase 3, one BWT, CHAINS FOLLOWED IN REGISTERS (round 4).  The suffix tree of a read collection is mostly unary under Weiner
// links: W is right-maximal because some read ends behind it ('#') while the others go on with one letter, and the same holds for
// cW, c'cW, ... until the reads run out -- ~135 nodes in a row whose only kept child is the next one (6.1 G nodes, ~50 M such
// chains at 7.55 G positions).  k_nodes1p sends every one of them through the stack: 32 bytes written, 32 read, packed, staged,
// unpacked.  Here a quad KEEPS the first kept child of its node -- lane jc's six boundaries are spread over the quad by DPP -- and
// expands it in its next turn; only the other kept children (one node in a hundred has any) go to the stack, and a quad takes a
// new record from the launch's input only when its chain ends.  A quad runs TWO chains alternately, so that the block loads of
// one are in flight while the other computes (the distance k_nodes1p keeps between its stages A and D), plus one input record
// fetched ahead.  Input records are handed out through a cursor (NavCounters::taken): a wave takes a chunk with one atomic, its
// quads take records from the chunk as their chains end.  Same bits, same counters: every node is still expanded exactly once.
// A launch may now push more than four records per input record; the planner's room rule no longer bounds that, so an overflow
// (never seen: the whole traversal pushes fewer records than ONE stack holds) ends the traversal with an error and the caller
// walks again with k_nodes1p (flags_build_chains).
template <bool SUPER_LDS>
__global__ void __launch_bounds__(256) k_nodes1c(eb_dev_index x1, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p, const NavLaunch* __restrict__ la,
                                                NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Node1P;
    // Pushes are rare here (one node in a hundred has a second kept child), so the staging buffer is small -- the 48 KB of k_nodes1p
    // would keep the leaf traversal, which runs side by side, off the CUs for the whole of these long launches -- and a record that
    // finds it full (the bushy top of the tree) goes straight to the stack with an atomic of its own.
    constexpr int STAGE = 256;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_super[EB_NAV_SUPER_LDS * 4];
    __shared__ int s_alive;
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const bool follow = la->follow != 0;
    if ((uint64_t)blockIdx.x * 64 >= count) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (SUPER_LDS) {
        const uint64_t n_super = (x1.n >> EB_SUPER_SHIFT) + 1;
        for (uint32_t k = threadIdx.x; k < n_super * 4; k += blockDim.x) s_super[k] = x1.super[k];
    }
    if (threadIdx.x == 0) { s_count = 0; s_alive = 0; }
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint64_t n_node = 0, n_lcp = 0, n_rank = 0, n_min = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t Fj = pick4(x1.F, j);
    NavCursor cursor(count, gridDim.x);

    // border bits of a node (update_lcp_threshold include.hpp:826-860, update_lcp_minima ebwt2InDel.cpp:357-391), as in k_nodes1p
    auto mark = [&](const Node1& N) {
        uint64_t lo = 0, hi = 0;
        for (int k = 0; k < 4; ++k) if (j == (uint32_t)k) { lo = N.b[k]; hi = N.b[k + 1]; }
        if (hi > lo && hi != N.b[5]) {
            const bool setK = ((int64_t)N.depth >= p.K) != (p.below != 0), setR = ((int64_t)N.depth >= p.k_right) != (p.below != 0);
            if (p.lcpb) {
                uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                if (v && !(EB_NDBG(p) & 1u)) p.lcpb[hi] = v;
            } else {
                if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6)), 1ull << (hi & 63u));
                if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6) + 1), 1ull << (hi & 63u));
            }
            ++n_lcp;
            n_s1 += hi;
            n_s2 += hi * hi;
        }
        if (j >= 1 && hi - lo >= 2 && hi < N.b[5] - 1) {
            if (!(EB_NDBG(p) & 2u)) eb_bit_set_atomic(f.minima, hi);
            ++n_min;
        }
    };

    // ---- two chains per quad + one record fetched ahead
    Node1 N0, N1;
    uint4 c0[6], c1[6];
    bool act0 = false, act1 = false, have_spare = false;
    Rec spare;
    spare.w[0] = spare.w[1] = spare.w[2] = spare.w[3] = 0;
    // the block loads of a node's six boundaries (consumed in its chain's next turn)
    auto load6 = [&](const Node1& N, uint4 (&c)[6], bool act) {
        // (Round 5: letting a boundary in the block of its left neighbour take that neighbour's chunk -- most of a unary node's six
        // boundaries coincide -- is SLOWER, 0.35 against 0.27 s per traversal at 7.55 G positions: the duplicate loads hit the CU's
        // L1, the five divergent branches per turn do not come free.  profiles/r05_nav_share_blocks_ab.txt)
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = eb_rank4_quad_load(x1, act ? N.b[k] : 0ull, j);
    };
    auto start = [&](Node1& N, uint4 (&c)[6], bool& act) {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        act = !want;
        Rec r;
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
        if (act) r = eb_load_global(in + idx);
        unpack_rec(r, N);
        if (act && p.write) mark(N);
        load6(N, c, true);
    };
    start(N0, c0, act0);
    start(N1, c1, act1);
    {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        have_spare = !want;
        if (have_spare) spare = eb_load_global(in + idx);
    }
    // one turn of a chain = ranks and children of its node (LF(sa_node) dna_bwt.hpp:323-356, next_nodes :381-404), then its next
    // node: the kept child, or the record fetched ahead.  Six turns (three per chain) between two flushes of the staging buffer.
    for (;;) {
        const bool wave_alive = __ballot(act0 || act1 || have_spare) != 0ull;
        if (wave_alive) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    Node1& N = sl ? N1 : N0;
                    uint4 (&c)[6] = sl ? c1 : c0;
                    bool& act = sl ? act1 : act0;
                    Node1 child;
                    child.pad = 0;
                    int kids = 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const uint64_t i = N.b[k];
                        uint32_t pc = eb_chunk_counts(c[k], eb_prefix_mask32((uint32_t)(i & 127u), j));
                        pc = eb_quad_sum(pc);
                        uint64_t sup;
                        if constexpr (SUPER_LDS) sup = s_super[((i >> EB_SUPER_SHIFT) << 2) + j];
                        else sup = x1.super[((i >> EB_SUPER_SHIFT) << 2) + j];
                        child.b[k] = Fj + sup + c[k].w + ((pc >> (8u * j)) & 0xFFu);
                    }
                    for (int k = 0; k < 5; ++k) kids += child.b[k + 1] > child.b[k];
                    child.depth = N.depth + 1;
                    const bool keep = act && kids >= 2; // number_of_children (include.hpp:760-768)
                    if (act && p.write && j == 0) {
                        ++n_node;
                        n_rank += 1;
                        for (int k = 1; k < 6; ++k) n_rank += N.b[k] != N.b[k - 1]; // the reference shares the query of equal neighbours
                    }
                    // the quad goes on with its first kept child; the others travel through the stack
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return pack_rec(child); }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    if (jc < 4u) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) N.b[k] = eb_quad_pick64(child.b[k], j == jc);
                        N.depth = child.depth;
                    } else {
                        // the chain ended (or the slot was idle): on with the record fetched ahead, and another one is asked for
                        act = have_spare;
                        if (have_spare) { unpack_rec(spare, N); want = true; }
                        have_spare = false;
                    }
                    uint64_t idx = 0;
                    bool asked = want;
                    cursor.take(cnt, want, idx, j, lane);
                    if (asked && !want) { spare = eb_load_global(in + idx); have_spare = true; }
                    // the new node of this chain: its border bits, its block loads (consumed in this chain's next turn)
                    if (act && p.write) mark(N);
                    load6(N, c, act);
                }
            }
        }
        // six turns add at most 6 x 256 records: the staging buffer is emptied here, and the workgroup goes on while any of its waves does
        if (!nav_round_end<Rec, STAGE>(stage, &s_count, &s_alive, out, out_cap, cnt, __ballot(act0 || act1 || have_spare) != 0ull, lane)) break;
    }
    const uint64_t st[EB_NAV_STATS] = {0, n_node, n_lcp, 0, n_min, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// ---- phase 2, one BWT, chains followed in registers like k_nodes1c.  Two reads that end at the same place share every suffix, so
// the leaves W#, cW#, c'cW#, ... form a chain as long as the reads (0.58 G leaves in ~4 M chains at 7.55 G positions, which
// k_leaves walks in lock-step: a hundred launches of 4 M records at the end of the traversal).  A quad keeps the first kept child
// leaf and expands it in its next turn; two chains per quad alternate, so that the two block loads of one are in flight while the
// other computes (k_leaves waits for its record, then for its blocks: nothing overlaps), one input record is fetched ahead.
__global__ void __launch_bounds__(256) k_leaves1c(eb_dev_index x1, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p, const NavLaunch* __restrict__ la,
                                                 NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Leaf1;
    constexpr int STAGE = 256;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ int s_alive;
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const bool follow = la->follow != 0;
    if ((uint64_t)blockIdx.x * 64 >= count) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (threadIdx.x == 0) { s_count = 0; s_alive = 0; }
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint64_t n_leaf = 0, n_lcp = 0, n_rank = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t Fj = pick4(x1.F, j);
    NavCursor cursor(count, gridDim.x);
    // LCP slots l+1 .. r-1 of a leaf (update_LCP_leaf, ebwt2InDel.cpp:344-355): the quad splits the touched words
    auto mark = [&](uint64_t l, uint64_t r, uint64_t depth) {
        if (r <= l + 1) return;
        const uint64_t lo = l + 1, hi = r;
        const bool setK = ((int64_t)depth >= p.K) != (p.below != 0), setR = ((int64_t)depth >= p.k_right) != (p.below != 0);
        if ((setK || setR) && p.lcpb) {
            const uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
            for (uint64_t i2 = lo + j; i2 < hi; i2 += 4) p.lcpb[i2] = v;
        } else if (setK || setR) {
            const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
            for (uint64_t w = w0 + j; w <= w1; w += 4) {
                uint64_t m = ~0ull;
                if (w == w0) m &= ~0ull << (lo & 63u);
                if (w == w1) m &= ~0ull >> (63u - ((hi - 1) & 63u));
                if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w), m);
                if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w + 1), m);
            }
        }
        if (j == 0) { n_lcp += hi - lo; n_s1 += eb_tri(hi) - eb_tri(lo); n_s2 += eb_sq(hi) - eb_sq(lo); }
    };
    struct Chain { uint64_t l, r, depth; uint4 cl, cr; bool act; };
    Chain C0, C1;
    bool have_spare = false;
    Rec spare;
    spare.l = spare.r = spare.depth = spare.pad = 0;
    // a chain's new leaf: its LCP bits, its statistics, its two block loads (consumed in the chain's next turn)
    auto arm = [&](Chain& C) {
        if (C.act && p.write) {
            mark(C.l, C.r, C.depth);
            if (j == 0) { ++n_leaf; n_rank += C.r > C.l ? 2 : 1; }
        }
        C.cl = eb_rank4_quad_load(x1, C.act ? C.l : 0ull, j);
        C.cr = eb_rank4_quad_load(x1, C.act ? (C.r > C.l ? C.r : C.l) : 0ull, j);
    };
    auto start = [&](Chain& C) {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        C.act = !want;
        Rec r;
        r.l = r.r = r.depth = r.pad = 0;
        if (C.act) r = eb_load_global(in + idx);
        C.l = r.l; C.r = r.r; C.depth = r.depth;
        arm(C);
    };
    start(C0);
    start(C1);
    {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        have_spare = !want;
        if (have_spare) spare = eb_load_global(in + idx);
    }
    for (;;) {
        const bool wave_alive = __ballot(C0.act || C1.act || have_spare) != 0ull;
        if (wave_alive) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    Chain& C = sl ? C1 : C0;
                    // next_leaves (dna_bwt.hpp:358-379): the child obtained by prepending letter j
                    const uint64_t a = eb_rank4_quad_finish(x1, C.cl, C.l, j), b = eb_rank4_quad_finish(x1, C.cr, C.r > C.l ? C.r : C.l, j);
                    Rec child;
                    child.l = Fj + a; child.r = Fj + b; child.depth = C.depth + 1; child.pad = 0;
                    const bool keep = C.act && (b - a >= 2);
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return child; }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    if (jc < 4u) {
                        C.l = eb_quad_pick64(child.l, j == jc);
                        C.r = eb_quad_pick64(child.r, j == jc);
                        C.depth = child.depth;
                    } else {
                        C.act = have_spare;
                        if (have_spare) { C.l = spare.l; C.r = spare.r; C.depth = spare.depth; want = true; }
                        have_spare = false;
                    }
                    uint64_t idx = 0;
                    const bool asked = want;
                    cursor.take(cnt, want, idx, j, lane);
                    if (asked && !want) { spare = eb_load_global(in + idx); have_spare = true; }
                    arm(C);
                }
            }
        }
        if (!nav_round_end<Rec, STAGE>(stage, &s_count, &s_alive, out, out_cap, cnt, __ballot(C0.act || C1.act || have_spare) != 0ull, lane)) break;
    }
    const uint64_t st[EB_NAV_STATS] = {n_leaf, 0, n_lcp, 0, 0, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// ---- phase 2, two BWTs, chains followed in registers: k_leaves1c for merged leaves (two sa_leaves: four block loads per turn; the DA
// slots of the leaf's rows are written with its LCP bits).
__global__ void __launch_bounds__(256) k_leaves2c(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p, const NavLaunch* __restrict__ la,
                                                 NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Leaf2P;
    constexpr int STAGE = 256;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ int s_alive;
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const bool follow = la->follow != 0;
    if ((uint64_t)blockIdx.x * 64 >= count) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (threadIdx.x == 0) { s_count = 0; s_alive = 0; }
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint64_t n_leaf = 0, n_lcp = 0, n_da = 0, n_rank = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t F1 = pick4(x1.F, j), F2 = pick4(x2.F, j);
    NavCursor cursor(count, gridDim.x);
    // DA and LCP slots of a merged leaf (update_DA ebwt2InDel.cpp:394-425, update_LCP_leaf :344-355), as in k_leaves<2>: DA = 1 on the
    // second collection's rows [s2, e), LCP bits on [s1 + 1, e); the quad splits the touched words
    auto mark = [&](const Leaf2& L) {
        const uint64_t s1 = L.l1 + L.l2, s2 = L.l2 + L.r1, e = L.r1 + L.r2;
        const bool setK = ((int64_t)L.depth >= p.K) != (p.below != 0), setR = ((int64_t)L.depth >= p.k_right) != (p.below != 0);
        const uint64_t w0 = s1 >> 6, w1 = e > s1 ? (e - 1) >> 6 : w0;
        for (uint64_t w = w0 + j; w <= w1 && e > s1; w += 4) {
            const uint64_t base = w << 6;
            auto span = [&](uint64_t lo, uint64_t hi) -> uint64_t { // bits of word w inside [lo,hi)
                if (hi <= lo || hi <= base || lo >= base + 64) return 0ull;
                uint64_t m = ~0ull;
                if (lo > base) m &= ~0ull << (lo - base);
                if (hi < base + 64) m &= ~0ull >> (base + 64 - hi);
                return m;
            };
            const uint64_t md = span(s2, e), ml = span(s1 + 1, e);
            if (md) atomicOr(reinterpret_cast<unsigned long long*>(f.da + w), md);
            if (!p.lcpb) {
                if (ml && setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w), ml);
                if (ml && setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * w + 1), ml);
            }
        }
        if (p.lcpb && (setK || setR)) {
            const uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
            for (uint64_t i2 = s1 + 1 + j; i2 < e; i2 += 4) p.lcpb[i2] = v;
        }
        if (j == 0) {
            ++n_leaf;
            n_da += e - s1;
            if (e > s1 + 1) { n_lcp += e - s1 - 1; n_s1 += eb_tri(e) - eb_tri(s1 + 1); n_s2 += eb_sq(e) - eb_sq(s1 + 1); }
            n_rank += (L.r1 > L.l1 ? 2 : 1) + (L.r2 > L.l2 ? 2 : 1);
        }
    };
    struct Chain { Leaf2 L; uint4 cl1, cr1, cl2, cr2; bool act; };
    Chain C0, C1;
    bool have_spare = false;
    uint64_t spare = 0; // index of the input record reserved for this quad's next chain
    // a chain's new leaf: its DA / LCP bits, its statistics, its four block loads (consumed in the chain's next turn)
    auto arm = [&](Chain& C) {
        if (C.act && p.write) mark(C.L);
        C.cl1 = eb_rank4_quad_load(x1, C.act ? C.L.l1 : 0ull, j);
        C.cr1 = eb_rank4_quad_load(x1, C.act ? (C.L.r1 > C.L.l1 ? C.L.r1 : C.L.l1) : 0ull, j);
        C.cl2 = eb_rank4_quad_load(x2, C.act ? C.L.l2 : 0ull, j);
        C.cr2 = eb_rank4_quad_load(x2, C.act ? (C.L.r2 > C.L.l2 ? C.L.r2 : C.L.l2) : 0ull, j);
    };
    auto start = [&](Chain& C) {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        C.act = !want;
        Rec r;
        for (uint64_t& w : r.w) w = 0;
        if (C.act) r = eb_load_global(in + idx);
        unpack_rec(r, C.L);
        arm(C);
    };
    start(C0);
    start(C1);
    {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        have_spare = !want;
        spare = idx;
    }
    for (;;) {
        const bool wave_alive = __ballot(C0.act || C1.act || have_spare) != 0ull;
        if (wave_alive) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    Chain& C = sl ? C1 : C0;
                    // next_leaves on both BWTs (ebwt2InDel.cpp:452-472): the child obtained by prepending letter j
                    const uint64_t a1 = eb_rank4_quad_finish(x1, C.cl1, C.L.l1, j), b1 = eb_rank4_quad_finish(x1, C.cr1, C.L.r1 > C.L.l1 ? C.L.r1 : C.L.l1, j);
                    const uint64_t a2 = eb_rank4_quad_finish(x2, C.cl2, C.L.l2, j), b2 = eb_rank4_quad_finish(x2, C.cr2, C.L.r2 > C.L.l2 ? C.L.r2 : C.L.l2, j);
                    Leaf2 child;
                    child.l1 = F1 + a1; child.r1 = F1 + b1; child.l2 = F2 + a2; child.r2 = F2 + b2; child.depth = C.L.depth + 1;
                    child.pad[0] = child.pad[1] = child.pad[2] = 0;
                    const bool keep = C.act && ((b1 - a1) + (b2 - a2) >= 2);
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return pack_rec(child); }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    if (jc < 4u) {
                        C.L.l1 = eb_quad_pick64(child.l1, j == jc);
                        C.L.r1 = eb_quad_pick64(child.r1, j == jc);
                        C.L.l2 = eb_quad_pick64(child.l2, j == jc);
                        C.L.r2 = eb_quad_pick64(child.r2, j == jc);
                        C.L.depth = child.depth;
                    } else {
                        C.act = have_spare;
                        if (have_spare) { unpack_rec(eb_load_global(in + spare), C.L); want = true; }
                        have_spare = false;
                    }
                    uint64_t idx = 0;
                    const bool asked = want;
                    cursor.take(cnt, want, idx, j, lane);
                    if (asked && !want) { spare = idx; have_spare = true; }
                    arm(C);
                }
            }
        }
        if (!nav_round_end<Rec, STAGE>(stage, &s_count, &s_alive, out, out_cap, cnt, __ballot(C0.act || C1.act || have_spare) != 0ull, lane)) break;
    }
    const uint64_t st[EB_NAV_STATS] = {n_leaf, 0, n_lcp, n_da, 0, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// ---- phase 3, two BWTs, software-pipelined the same way (k_nodes1p).  A merged node is two sa_nodes (12 boundaries); as in
// the round-2 kernel a block is loaded ONCE for the boundaries of a node that share it (they are sorted and a few dozen rows apart):
// which boundaries issue a load is decided in stage A and kept as a 12-bit mask for stage D.
template <bool SUPER_LDS>
__global__ void __launch_bounds__(256) k_nodes2p(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p,
                                                const NavLaunch* __restrict__ la, NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Node2P;
    constexpr int STAGE = 768;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_super[2][EB_NAV_SUPER_LDS * 4];
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const uint64_t tiles = (count + 63) / 64;
    if ((uint64_t)blockIdx.x >= tiles) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (SUPER_LDS) {
        const uint64_t ns1 = (x1.n >> EB_SUPER_SHIFT) + 1, ns2 = (x2.n >> EB_SUPER_SHIFT) + 1;
        for (uint32_t k = threadIdx.x; k < ns1 * 4; k += blockDim.x) s_super[0][k] = x1.super[k];
        for (uint32_t k = threadIdx.x; k < ns2 * 4; k += blockDim.x) s_super[1][k] = x2.super[k];
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, q = threadIdx.x >> 2;
    const uint64_t G = gridDim.x;
    uint64_t n_node = 0, n_lcp = 0, n_da = 0, n_rank = 0, n_min = 0, n_s1 = 0, n_s2 = 0;
    const uint64_t F1 = pick4(x1.F, j), F2 = pick4(x2.F, j);

    auto load_rec = [&](uint64_t tile) -> Rec {
        Rec r;
        for (uint64_t& w : r.w) w = 0;
        const uint64_t i = tile * 64 + q;
        if (tile < tiles && i < count) r = eb_load_global(in + i);
        return r;
    };
    // stage A for one record: border bits in merged coordinates (merge_nodes, include.hpp:476-490), DA slots of singleton leaves
    auto mark = [&](const Node2& N) {
        uint64_t lo = 0, hi = 0;
        const uint64_t last = N.a[5] + N.b[5];
        for (int k = 0; k < 4; ++k) if (j == (uint32_t)k) { lo = N.a[k] + N.b[k]; hi = N.a[k + 1] + N.b[k + 1]; }
        if (hi > lo && hi != last) {
            const bool setK = ((int64_t)N.depth >= p.K) != (p.below != 0), setR = ((int64_t)N.depth >= p.k_right) != (p.below != 0);
            if (p.lcpb) {
                uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                if (v && !(EB_NDBG(p) & 1u)) p.lcpb[hi] = v;
            } else {
                if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6)), 1ull << (hi & 63u));
                if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6) + 1), 1ull << (hi & 63u));
            }
            ++n_lcp;
            n_s1 += hi;
            n_s2 += hi * hi;
        }
        if (j >= 1 && hi - lo >= 2 && hi < last - 1) {
            if (!(EB_NDBG(p) & 2u)) eb_bit_set_atomic(f.minima, hi);
            ++n_min;
        }
        // leaves of merged size 1 were not visited in phase 2: write their DA slot now (find_leaves, ebwt2InDel.cpp:474-527)
        for (int k = 0; k < 5; ++k) {
            if ((uint32_t)(k & 3) != j || (k == 4 && j != 0)) continue;
            uint64_t s1 = N.a[k + 1] - N.a[k], s2 = N.b[k + 1] - N.b[k];
            if (s1 + s2 == 1) {
                if (s2 && !(EB_NDBG(p) & 8u)) eb_bit_set_atomic(f.da, N.b[k] + N.a[k + 1]);
                ++n_da;
            }
        }
    };
    // which boundaries start a new block (bit k: a[k], bit 6 + k: b[k]); the others reuse their left neighbour's
    auto load_mask = [&](const Node2& N) -> uint32_t {
        uint32_t m = 1u | (1u << 6);
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            m |= ((N.a[k] >> 7) != (N.a[k - 1] >> 7) ? 1u : 0u) << k;
            m |= ((N.b[k] >> 7) != (N.b[k - 1] >> 7) ? 1u : 0u) << (6 + k);
        }
        return m;
    };

    uint64_t tile = blockIdx.x;
    Node2 Nn;
    uint4 cna[6], cnb[6];
    uint32_t mn = 0;
    bool act_n;
    {
        Rec r0 = load_rec(tile);
        unpack_rec(r0, Nn);
        act_n = tile * 64 + q < count;
        if (act_n && p.write) mark(Nn);
        mn = load_mask(Nn);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if ((mn >> k) & 1u) cna[k] = eb_rank4_quad_load(x1, Nn.a[k], j);
            if ((mn >> (6 + k)) & 1u) cnb[k] = eb_rank4_quad_load(x2, Nn.b[k], j);
        }
    }
    Rec rr = load_rec(tile + G);
    uint32_t trips = 0;
    for (; tile < tiles; tile += G) {
        const Node2 Nc = Nn;
        const bool act_c = act_n;
        const uint32_t mc = mn;
        uint4 cca[6], ccb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { cca[k] = cna[k]; ccb[k] = cnb[k]; }
        // stage A: the next trip's record -> node, bits, block loads
        unpack_rec(rr, Nn);
        act_n = tile + G < tiles && (tile + G) * 64 + q < count;
        if (act_n && p.write) mark(Nn);
        mn = load_mask(Nn);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if ((mn >> k) & 1u) cna[k] = eb_rank4_quad_load(x1, Nn.a[k], j);
            if ((mn >> (6 + k)) & 1u) cnb[k] = eb_rank4_quad_load(x2, Nn.b[k], j);
        }
        // stage C: the record after that
        rr = load_rec(tile + 2 * G);
        // stage D: this trip's ranks and children (LF on both BWTs, next_nodes ebwt2InDel.cpp:529-553)
        Node2 child;
        child.pad[0] = child.pad[1] = child.pad[2] = 0;
        int kids = 0;
        uint4 cur_a = cca[0], cur_b = ccb[0];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if ((mc >> k) & 1u) cur_a = cca[k]; // else: the block of the boundary to the left
            if ((mc >> (6 + k)) & 1u) cur_b = ccb[k];
            const uint64_t ia = Nc.a[k], ib = Nc.b[k];
            uint32_t pa = eb_chunk_counts(cur_a, eb_prefix_mask32((uint32_t)(ia & 127u), j));
            uint32_t pb = eb_chunk_counts(cur_b, eb_prefix_mask32((uint32_t)(ib & 127u), j));
            pa = eb_quad_sum(pa);
            pb = eb_quad_sum(pb);
            uint64_t sa, sb;
            if constexpr (SUPER_LDS) { sa = s_super[0][((ia >> EB_SUPER_SHIFT) << 2) + j]; sb = s_super[1][((ib >> EB_SUPER_SHIFT) << 2) + j]; }
            else { sa = x1.super[((ia >> EB_SUPER_SHIFT) << 2) + j]; sb = x2.super[((ib >> EB_SUPER_SHIFT) << 2) + j]; }
            child.a[k] = F1 + sa + cur_a.w + ((pa >> (8u * j)) & 0xFFu);
            child.b[k] = F2 + sb + cur_b.w + ((pb >> (8u * j)) & 0xFFu);
        }
        for (int k = 0; k < 5; ++k) kids += (child.a[k + 1] > child.a[k]) || (child.b[k + 1] > child.b[k]); // include.hpp:784-792
        child.depth = Nc.depth + 1;
        const bool keep = act_c && kids >= 2;
        if (act_c && p.write && j == 0) {
            ++n_node;
            n_rank += 2;
            for (int k = 1; k < 6; ++k) n_rank += (Nc.a[k] != Nc.a[k - 1]) + (Nc.b[k] != Nc.b[k - 1]);
        }
        stage.push(keep, pack_rec(child));
        if (++trips % (STAGE / 256) == 0) stage.flush(out, out_cap, &cnt->pushed);
    }
    stage.flush(out, out_cap, &cnt->pushed);
    const uint64_t st[EB_NAV_STATS] = {0, n_node, n_lcp, n_da, n_min, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// ---- phase 3, two BWTs, chains followed in registers (k_nodes1c for merged nodes).  ONE chain per quad (two, like k_nodes1c, need
// 256 registers: one wave per SIMD) and one input record fetched ahead; the loads of a turn are covered by the other waves of the
// SIMD.  The first kept child's twelve boundaries are spread over the quad by DPP, the other kept children go to the stack (small
// staging buffer, excess straight to the stack).
template <bool SUPER_LDS>
__global__ void __launch_bounds__(256) k_nodes2c(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p,
                                                const NavLaunch* __restrict__ la, NavCounters* cnt, unsigned long long* __restrict__ stat_rows) {
    using Rec = Node2P;
    constexpr int STAGE = 256;
    __shared__ uint4 s_buf[STAGE * (sizeof(Rec) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_super[2][EB_NAV_SUPER_LDS * 4];
    __shared__ int s_alive;
    const Rec* __restrict__ in = static_cast<const Rec*>(la->in);
    Rec* out = static_cast<Rec*>(la->out);
    const uint64_t count = la->count, out_cap = la->out_cap;
    p.write = la->write;
    const bool follow = la->follow != 0;
    if ((uint64_t)blockIdx.x * 64 >= count) return; // the grid is sized for a full batch: nothing for this workgroup (uniform)
    Stager<Rec, STAGE> stage{s_buf, &s_count, &s_base};
    if (SUPER_LDS) {
        const uint64_t ns1 = (x1.n >> EB_SUPER_SHIFT) + 1, ns2 = (x2.n >> EB_SUPER_SHIFT) + 1;
        for (uint32_t k = threadIdx.x; k < ns1 * 4; k += blockDim.x) s_super[0][k] = x1.super[k];
        for (uint32_t k = threadIdx.x; k < ns2 * 4; k += blockDim.x) s_super[1][k] = x2.super[k];
    }
    if (threadIdx.x == 0) { s_count = 0; s_alive = 0; }
    __syncthreads();
    const uint32_t j = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t n_node = 0, n_lcp = 0, n_da = 0, n_rank = 0, n_min = 0; // per lane and launch: far below 2^32
    uint64_t n_s1 = 0, n_s2 = 0;
    const uint64_t F1 = pick4(x1.F, j), F2 = pick4(x2.F, j);
    NavCursor cursor(count, gridDim.x);

    // border bits in merged coordinates (merge_nodes, include.hpp:476-490), DA slots of singleton leaves: as in k_nodes2p
    auto mark = [&](const Node2& N) {
        uint64_t lo = 0, hi = 0;
        const uint64_t last = N.a[5] + N.b[5];
        for (int k = 0; k < 4; ++k) if (j == (uint32_t)k) { lo = N.a[k] + N.b[k]; hi = N.a[k + 1] + N.b[k + 1]; }
        if (hi > lo && hi != last) {
            const bool setK = ((int64_t)N.depth >= p.K) != (p.below != 0), setR = ((int64_t)N.depth >= p.k_right) != (p.below != 0);
            if (p.lcpb) {
                uint8_t v = (uint8_t)((setK ? 1 : 0) | (setR ? 2 : 0));
                if (v && !(EB_NDBG(p) & 1u)) p.lcpb[hi] = v;
            } else {
                if (setK) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6)), 1ull << (hi & 63u));
                if (setR) atomicOr(reinterpret_cast<unsigned long long*>(kr + 2 * (hi >> 6) + 1), 1ull << (hi & 63u));
            }
            ++n_lcp;
            n_s1 += hi;
            n_s2 += hi * hi;
        }
        if (j >= 1 && hi - lo >= 2 && hi < last - 1) {
            if (!(EB_NDBG(p) & 2u)) eb_bit_set_atomic(f.minima, hi);
            ++n_min;
        }
        // leaves of merged size 1 were not visited in phase 2: write their DA slot now (find_leaves, ebwt2InDel.cpp:474-527)
        for (int k = 0; k < 5; ++k) {
            if ((uint32_t)(k & 3) != j || (k == 4 && j != 0)) continue;
            uint64_t s1 = N.a[k + 1] - N.a[k], s2 = N.b[k + 1] - N.b[k];
            if (s1 + s2 == 1) {
                if (s2 && !(EB_NDBG(p) & 8u)) eb_bit_set_atomic(f.da, N.b[k] + N.a[k + 1]);
                ++n_da;
            }
        }
    };
    // which boundaries start a new block (bit k: a[k], bit 6 + k: b[k]); the others reuse their left neighbour's
    auto load_mask = [&](const Node2& N) -> uint32_t {
        // (all twelve loaded unconditionally, measured in round 5: 0.645 s against 0.62 s, profiles/r05_nav2_loads_ab.txt)
        uint32_t m = 1u | (1u << 6);
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            m |= ((N.a[k] >> 7) != (N.a[k - 1] >> 7) ? 1u : 0u) << k;
            m |= ((N.b[k] >> 7) != (N.b[k - 1] >> 7) ? 1u : 0u) << (6 + k);
        }
        return m;
    };
    Node2 N0;
    uint4 a0[6], b0[6];
    uint32_t m0 = 0;
    bool act0 = false, have_spare = false;
    uint64_t spare = 0; // index of the input record reserved for this quad's next chain (loaded when the chain at hand ends: once in ~135 turns)
    // a chain's new node: its bits, its block loads (consumed in the chain's next turn)
    auto arm = [&](Node2& N, uint4 (&ca)[6], uint4 (&cb)[6], uint32_t& m, bool act) {
        if (act && p.write) mark(N);
        m = load_mask(N);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if ((m >> k) & 1u) ca[k] = eb_rank4_quad_load(x1, act ? N.a[k] : 0ull, j);
            if ((m >> (6 + k)) & 1u) cb[k] = eb_rank4_quad_load(x2, act ? N.b[k] : 0ull, j);
        }
    };
    auto start = [&](Node2& N, uint4 (&ca)[6], uint4 (&cb)[6], uint32_t& m, bool& act) {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        act = !want;
        Rec r;
        for (uint64_t& w : r.w) w = 0;
        if (act) r = eb_load_global(in + idx);
        unpack_rec(r, N);
        arm(N, ca, cb, m, act);
    };
    start(N0, a0, b0, m0, act0);
    {
        bool want = true;
        uint64_t idx = 0;
        cursor.take(cnt, want, idx, j, lane);
        have_spare = !want;
        spare = idx;
    }
    for (;;) {
        const bool wave_alive = __ballot(act0 || have_spare) != 0ull;
        if (wave_alive) {
            for (int rep = 0; rep < 6; ++rep) {
                {
                    Node2& N = N0;
                    uint4 (&ca)[6] = a0;
                    uint4 (&cb)[6] = b0;
                    uint32_t& mc = m0;
                    bool& act = act0;
                    // ranks and children (LF on both BWTs, next_nodes ebwt2InDel.cpp:529-553)
                    Node2 child;
                    child.pad[0] = child.pad[1] = child.pad[2] = 0;
                    int kids = 0;
                    uint4 cur_a = ca[0], cur_b = cb[0];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if ((mc >> k) & 1u) cur_a = ca[k]; // else: the block of the boundary to the left
                        if ((mc >> (6 + k)) & 1u) cur_b = cb[k];
                        const uint64_t ia = N.a[k], ib = N.b[k];
                        uint32_t pa = eb_chunk_counts(cur_a, eb_prefix_mask32((uint32_t)(ia & 127u), j));
                        uint32_t pb = eb_chunk_counts(cur_b, eb_prefix_mask32((uint32_t)(ib & 127u), j));
                        pa = eb_quad_sum(pa);
                        pb = eb_quad_sum(pb);
                        uint64_t sa, sb;
                        if constexpr (SUPER_LDS) { sa = s_super[0][((ia >> EB_SUPER_SHIFT) << 2) + j]; sb = s_super[1][((ib >> EB_SUPER_SHIFT) << 2) + j]; }
                        else { sa = x1.super[((ia >> EB_SUPER_SHIFT) << 2) + j]; sb = x2.super[((ib >> EB_SUPER_SHIFT) << 2) + j]; }
                        child.a[k] = F1 + sa + cur_a.w + ((pa >> (8u * j)) & 0xFFu);
                        child.b[k] = F2 + sb + cur_b.w + ((pb >> (8u * j)) & 0xFFu);
                    }
                    for (int k = 0; k < 5; ++k) kids += (child.a[k + 1] > child.a[k]) || (child.b[k + 1] > child.b[k]); // include.hpp:784-792
                    child.depth = N.depth + 1;
                    const bool keep = act && kids >= 2;
                    if (act && p.write && j == 0) {
                        ++n_node;
                        n_rank += 2;
                        for (int k = 1; k < 6; ++k) n_rank += (N.a[k] != N.a[k - 1]) + (N.b[k] != N.b[k - 1]);
                    }
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return pack_rec(child); }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    if (jc < 4u) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            N.a[k] = eb_quad_pick64(child.a[k], j == jc);
                            N.b[k] = eb_quad_pick64(child.b[k], j == jc);
                        }
                        N.depth = child.depth;
                    } else {
                        act = have_spare;
                        if (have_spare) { unpack_rec(eb_load_global(in + spare), N); want = true; }
                        have_spare = false;
                    }
                    uint64_t idx = 0;
                    const bool asked = want;
                    cursor.take(cnt, want, idx, j, lane);
                    if (asked && !want) { spare = idx; have_spare = true; }
                    arm(N, ca, cb, mc, act);
                }
            }
        }
        if (!nav_round_end<Rec, STAGE>(stage, &s_count, &s_alive, out, out_cap, cnt, __ballot(act0 || have_spare) != 0ull, lane)) break;
    }
    const uint64_t st[EB_NAV_STATS] = {0, n_node, n_lcp, n_da, n_min, n_rank, n_s1, n_s2};
    flush_stats(stat_rows, st);
}

// Sharded traversal: at the first level wide enough, every shard keeps the records whose interval start
// hashes to it -- content-addressed, so all shards agree whatever order their own launches left the level in.
// W2 >= 0: a second word of the record is added to the key (packed two-BWT records: more of the two interval starts).
template <typename Item, int W2>
__global__ void __launch_bounds__(256) k_keep_shard(const NavLaunch* __restrict__ la, uint32_t shard, uint32_t n_shards, NavCounters* cnt) {
    const Item* __restrict__ in = static_cast<const Item*>(la->in);
    Item* __restrict__ out = static_cast<Item*>(la->out);
    const uint64_t count = la->count; // 0 unless k_drain_plan found the level to cut
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < count; base += total) {
        uint64_t i = base + threadIdx.x;
        bool keep = false;
        if (i < count) {
            const uint64_t* w = reinterpret_cast<const uint64_t*>(in + i);
            uint64_t key = w[0];
            if (W2 >= 0) key += w[W2 < 0 ? 0 : W2];
            uint64_t h = key * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            keep = (uint32_t)(h % n_shards) == shard;
        }
        unsigned long long mask = __ballot(keep);
        unsigned long long wbase = 0;
        if (lane == 0 && mask) wbase = atomicAdd(&cnt->pushed, (unsigned long long)__popcll(mask));
        wbase = __shfl(wbase, 0);
        if (keep) {
            uint64_t slot = wbase + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
            const Item v = eb_load_global(in + i);
            const uint4* src = reinterpret_cast<const uint4*>(&v);
            uint4* dst = reinterpret_cast<uint4*>(out + slot);
            for (int k = 0; k < (int)(sizeof(Item) / 16); ++k) eb_store_global(dst + k, src[k]);
        }
    }
}

// after navigation: kr[2w], kr[2w+1] -> thrK[w], thrR[w] (streaming, ~1 ms per 7.5 G positions)
__global__ void __launch_bounds__(256) k_split_kr(const uint4* __restrict__ kr, uint64_t words, uint64_t* __restrict__ thrK, uint64_t* __restrict__ thrR) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = kr[w];
        thrK[w] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        thrR[w] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
}

// after navigation: 64 bytes -> one word of thrK and one of thrR
__global__ void __launch_bounds__(256) k_pack_lcp(const uint4* __restrict__ lcpb, uint64_t words, uint64_t* __restrict__ thrK, uint64_t* __restrict__ thrR) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0, r = 0;
        for (int q = 0; q < 4; ++q) {
            uint4 v = lcpb[w * 4 + q];
            uint32_t d[4] = {v.x, v.y, v.z, v.w};
            for (int t = 0; t < 4; ++t)
                for (int b = 0; b < 4; ++b) {
                    uint32_t byte = (d[t] >> (8 * b)) & 0xFFu;
                    uint32_t pos = q * 16 + t * 4 + b;
                    k |= (uint64_t)(byte & 1u) << pos;
                    r |= (uint64_t)((byte >> 1) & 1u) << pos;
                }
        }
        thrK[w] = k;
        thrR[w] = r;
    }
}

// complement form -> final form: thr = ~collected on positions 1 .. n-1, zero elsewhere (position 0 is never written by
// the reference and stays 0; so do the spare bits behind n)
__global__ void __launch_bounds__(256) k_invert_thr(uint64_t* __restrict__ thrK, uint64_t* __restrict__ thrR, uint64_t n, uint64_t alloc_words) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < alloc_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t m = 0;
        const uint64_t base = w << 6;
        if (base < n) {
            m = ~0ull;
            if (n - base < 64) m &= ~0ull >> (64 - (n - base));
            if (w == 0) m &= ~1ull;
        }
        thrK[w] = ~thrK[w] & m;
        thrR[w] = ~thrR[w] & m;
    }
}

// ones in each group of 8 DA words -> exclusive prefix = ones before the group
__global__ void __launch_bounds__(256) k_da_group_counts(const uint64_t* __restrict__ da, uint64_t words, uint64_t groups,
                                                         uint64_t* __restrict__ vec4) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    uint64_t s = 0;
    for (int k = 0; k < 8; ++k) {
        uint64_t w = g * 8 + k;
        if (w < words) s += (uint64_t)__popcll(da[w]);
    }
    vec4[g * 4] = s;
    vec4[g * 4 + 1] = vec4[g * 4 + 2] = vec4[g * 4 + 3] = 0;
}
__global__ void __launch_bounds__(256) k_take_first(const uint64_t* __restrict__ vec4, uint64_t groups, uint64_t* __restrict__ out) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < groups) out[g] = vec4[g * 4];
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// 64 * per_quad records per 256-thread workgroup and trip; the grid is capped and strides over the batch
unsigned nav_grid(uint64_t count, uint64_t per_quad = 1) {
    uint64_t tiles = (count + 64 * per_quad - 1) / (64 * per_quad);
    const uint64_t cap = 256ull * 8;
    return (unsigned)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}
unsigned nav_grid_for(uint64_t batch) { return nav_grid(batch, 1); } // per_quad > 1 only lowers the number of tiles: the cap is what large batches see

// ---- the drain's bookkeeping, on the DEVICE.  Round 2 ran it on the host: after every launch the host read back how many
// children it had pushed (8 bytes, a blocking round trip: ~700 of them at 7.55 G positions) and chose the next batch.  Now a
// one-thread kernel does exactly that between two expansions, so the host enqueues eight launches back to back, reads the
// state once per group, and always keeps one group queued ahead of the one it waits for: the device never idles on the host.
enum { DRAIN_NONE = 0, DRAIN_EXPAND = 1, DRAIN_SPLIT = 2 };
enum { DRAIN_ERR_CAPACITY = 1, DRAIN_ERR_OVERFLOW = 2 };
#define EB_NAV_OVERFLOW (-100) /* internal: a launch pushed more records than it was given room for (never leaves this file) */
struct DrainState {
    unsigned long long stack[2];          // device addresses of the two LIFO stacks
    unsigned long long top[2];            // records on each
    unsigned long long cap, batch;        // records a stack holds; most records popped per launch
    unsigned long long launches, max_top; // expansions that had work; most records pending at once
    unsigned long long split_min;
    uint32_t item_bytes;
    uint32_t recent;                      // stack that received records last
    uint32_t last, last_src;              // what the previous step was (DRAIN_*) and which stack it popped
    unsigned long long last_base, last_room;
    uint32_t shard, n_shards, split_done, error;
    unsigned long long room_hook;         // tests (EB_NAV_CHAINS=3): most room a launch is TOLD it has, so that it really overflows (0: the truth)
    NavLaunch expand;                     // what the next expansion kernel works on (count 0: nothing)
    NavLaunch split;                      // ... and the next k_keep_shard (count 0: nothing)
};

// books the children of the previous step
__device__ __forceinline__ void drain_account(DrainState* st, NavCounters* cnt) {
    const unsigned long long pushed = cnt->pushed;
    cnt->pushed = 0;
    cnt->taken = 0;
    if (st->last == DRAIN_EXPAND) {
        const uint32_t src = st->last_src, dst = 1u - src;
        if (pushed > st->last_room) st->error = DRAIN_ERR_OVERFLOW;
        st->top[src] = st->last_base;
        st->top[dst] += pushed;
        if (pushed) st->recent = dst;
    } else if (st->last == DRAIN_SPLIT) {
        const uint32_t src = st->last_src, dst = 1u - src;
        st->top[src] = 0;
        st->top[dst] = pushed;
        st->recent = dst;
        st->split_done = 1;
    }
    st->last = DRAIN_NONE;
    if (st->top[0] + st->top[1] > st->max_top) st->max_top = st->top[0] + st->top[1];
}
__global__ void k_drain_account(DrainState* st, NavCounters* cnt) { drain_account(st, cnt); }
// ... and chooses the next step: TWO stacks alternate; a batch is popped from the top of one and expanded in place (the
// kernel reads it where it lies -- no staging copy), its children are pushed onto the other, which becomes the one to pop from
// next.  Popping the most recently pushed records first keeps a depth-first footprint.
__global__ void k_drain_plan(DrainState* st, NavCounters* cnt) {
    drain_account(st, cnt);
    st->expand.count = 0;
    st->split.count = 0;
    if (st->error || st->top[0] + st->top[1] == 0) return;
    const uint32_t recent = st->recent;
    const uint32_t src = st->top[recent] > 0 ? recent : 1u - recent, dst = 1u - src;
    const unsigned long long ib = st->item_bytes;
    if (st->n_shards > 1 && !st->split_done && st->top[dst] == 0 && st->top[src] >= st->split_min * st->n_shards) {
        // so far every launch expanded one whole level: cut it (k_keep_shard)
        st->split.in = reinterpret_cast<const void*>(st->stack[src]);
        st->split.count = st->top[src];
        st->split.out = reinterpret_cast<void*>(st->stack[dst]);
        st->split.out_cap = st->cap;
        st->last = DRAIN_SPLIT;
        st->last_src = src;
        return;
    }
    const unsigned long long room = st->cap - st->top[dst];
    unsigned long long B = st->top[src] < st->batch ? st->top[src] : st->batch;
    if (4 * B > room) B = room / 4; // a record has at most four children
    if (B == 0) { st->error = DRAIN_ERR_CAPACITY; return; }
    const unsigned long long base = st->top[src] - B;
    st->expand.in = reinterpret_cast<const void*>(st->stack[src] + base * ib);
    st->expand.count = B;
    st->expand.out = reinterpret_cast<void*>(st->stack[dst] + st->top[dst] * ib);
    const unsigned long long handed = st->room_hook && st->room_hook < room ? st->room_hook : room;
    st->expand.out_cap = handed;
    st->expand.write = (st->n_shards == 1 || st->split_done || st->shard == 0) ? 1 : 0;
    st->expand.follow = (st->n_shards == 1 || st->split_done) ? 1 : 0;
    st->last = DRAIN_EXPAND;
    st->last_src = src;
    st->last_base = base;
    st->last_room = handed;
    ++st->launches;
}

template <typename Item>
struct Drain {
    // Drains LIFO work stacks in HBM with batched launches.  `launch(const NavLaunch* la, grid)` must append children
    // through NavCounters::pushed.
    int device;
    hipStream_t stream;
    Item* d_stack[2] = {nullptr, nullptr};
    uint64_t cap = 0, batch = 0, top[2] = {0, 0};
    uint64_t launches = 0, max_top = 0;
    // sharded traversal: the tree is walked breadth-first (identically on every shard, bits written by shard 0 only)
    // down to the first level of at least split_min * n_shards records, which is then cut by k_keep_shard
    uint32_t shard = 0, n_shards = 1;
    uint64_t split_min = 4096;
    uint64_t room_hook = 0; // see DrainState

    // The stacks live in one arena shared by both phases (allocating HBM costs ~20 ms per GB, so the arena is sized from
    // the batch, not from n).  A depth-first drain needs a few batches of room (observed: < 7 batches at 7.55 G
    // positions); if a stack ever fills, the planner shrinks the batch -- in the limit a plain depth-first walk that needs
    // 4 x depth records -- so capacity never affects results.
    int init(void* arena, uint64_t arena_bytes, uint64_t n_bound, uint64_t batch_max) {
        batch = batch_max;
        if (batch > n_bound / 2 + 64) batch = n_bound / 2 + 64;
        uint64_t records = arena_bytes / sizeof(Item);
        if (records < 16 * 1024) { eb_set_error("traversal arena too small"); return EB_ERR_NOMEM; }
        cap = records / 2;
        if (batch > cap / 8) batch = cap / 8;
        uint64_t need = n_bound + 4 * batch + 64;   // every pending record is a distinct suffix-tree node/leaf
        if (cap > need) cap = need;
        d_stack[0] = static_cast<Item*>(arena);
        d_stack[1] = d_stack[0] + cap;
        top[0] = top[1] = 0;
        if (eb_cfg().trace_nav) fprintf(stderr, "[nav] 2 stacks x %llu records (%.2f GB) batch=%llu\n", (unsigned long long)cap, 2 * cap * sizeof(Item) / 1e9, (unsigned long long)batch);
        return EB_OK;
    }
    int push_host(const Item& it) {
        EB_HIP(hipMemcpyAsync(d_stack[0] + top[0], &it, sizeof(Item), hipMemcpyHostToDevice, stream));
        EB_HIP(hipStreamSynchronize(stream));
        ++top[0];
        return EB_OK;
    }
    // d_state: sizeof(DrainState) bytes of device memory; h_state: two DrainState in page-locked host memory
    template <int W2, typename Launch>
    int run(NavCounters* d_cnt, DrainState* d_state, DrainState* h_state, Launch launch) {
        const bool trace = eb_cfg().trace_nav;
        const double t_start = now_s();
        const int group = 8; // launches per read-back
        DrainState init;
        memset(&init, 0, sizeof init);
        init.stack[0] = reinterpret_cast<unsigned long long>(d_stack[0]);
        init.stack[1] = reinterpret_cast<unsigned long long>(d_stack[1]);
        init.top[0] = top[0];
        init.top[1] = top[1];
        init.cap = cap;
        init.batch = batch;
        init.split_min = split_min;
        init.item_bytes = (uint32_t)sizeof(Item);
        init.shard = shard;
        init.n_shards = n_shards;
        init.room_hook = room_hook;
        init.max_top = top[0] + top[1];
        h_state[0] = init;
        EB_HIP(hipMemsetAsync(&d_cnt->pushed, 0, sizeof(unsigned long long), stream));
        EB_HIP(hipMemcpyAsync(d_state, &h_state[0], sizeof(DrainState), hipMemcpyHostToDevice, stream));
        EB_HIP(hipStreamSynchronize(stream)); // h_state[0] is reused as a snapshot buffer below
        hipEvent_t ev[2] = {nullptr, nullptr};
        struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } evg{ev};
        EB_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        EB_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        // full-batch grid: the launch does not know its batch (k_drain_plan decides it on the device); workgroups beyond it leave at once
        const unsigned full_grid = nav_grid_for(batch);
        bool split_pending = n_shards > 1;
        auto enqueue_group = [&](int slot) -> int {
            for (int k = 0; k < group; ++k) {
                hipLaunchKernelGGL(k_drain_plan, dim3(1), dim3(1), 0, stream, d_state, d_cnt);
                if (split_pending)
                    hipLaunchKernelGGL((k_keep_shard<Item, W2>), dim3(2048), dim3(256), 0, stream, &d_state->split, shard, n_shards, d_cnt);
                launch(&d_state->expand, full_grid);
            }
            hipLaunchKernelGGL(k_drain_account, dim3(1), dim3(1), 0, stream, d_state, d_cnt);
            EB_HIP(hipGetLastError());
            EB_HIP(hipMemcpyAsync(&h_state[slot], d_state, sizeof(DrainState), hipMemcpyDeviceToHost, stream));
            EB_HIP(hipEventRecord(ev[slot], stream));
            return EB_OK;
        };
        int rc = EB_OK, g = 0;
        if ((rc = enqueue_group(0)) != EB_OK) return rc;
        for (;;) {
            // one group queued ahead of the one waited for
            if ((rc = enqueue_group((g + 1) & 1)) != EB_OK) break;
            EB_HIP(hipEventSynchronize(ev[g & 1]));
            const DrainState& s = h_state[g & 1];
            if (trace)
                fprintf(stderr, "[nav] group %d: launches=%llu pending=%llu max=%llu t=%.3fs\n", g, (unsigned long long)s.launches,
                        (unsigned long long)(s.top[0] + s.top[1]), (unsigned long long)s.max_top, now_s() - t_start);
            if (s.split_done) split_pending = false;
            if (s.error || s.top[0] + s.top[1] == 0) {
                // the group queued ahead finds nothing to do (or the same error): wait for it, it still uses the buffers
                EB_HIP(hipEventSynchronize(ev[(g + 1) & 1]));
                const DrainState& e = h_state[(g + 1) & 1];
                top[0] = e.top[0];
                top[1] = e.top[1];
                launches = e.launches;
                max_top = e.max_top;
                if (e.error == DRAIN_ERR_CAPACITY) { eb_set_error("traversal stack capacity exceeded (cap %llu records)", (unsigned long long)cap); return EB_ERR_NOMEM; }
                if (e.error) { eb_set_error("traversal stack overflow"); return EB_NAV_OVERFLOW; }
                return EB_OK;
            }
            ++g;
        }
        (void)hipStreamSynchronize(stream);
        return rc;
    }
};

}  // namespace

int eb_flags_alloc(uint64_t n, int device, bool with_da, eb_flags** out) {
    int ndev = eb_device_count();
    if (ndev <= 0) { eb_set_error("no HIP device available: the ebwt2InDel hot path runs on the GPU only (no CPU fallback)"); return EB_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { eb_set_error("device %d out of range (have %d)", device, ndev); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(device));
    eb_flags* f = new eb_flags();
    f->device = device;
    f->n = n;
    f->words = eb_bit_words(n);
    size_t bytes = eb_bit_alloc_words(n) * 8;
    hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    uint64_t** vecs[5] = {&f->d_thrK, &f->d_thrR, &f->d_min, with_da ? &f->d_da : nullptr, &f->d_flag};
    for (int i = 0; i < 5 && e == hipSuccess; ++i) {
        if (!vecs[i]) continue;
        e = hipMalloc(vecs[i], bytes);
        if (e == hipSuccess) e = hipMemsetAsync(*vecs[i], 0, bytes, f->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    if (e != hipSuccess) { eb_set_error("flag vector allocation failed: %s", hipGetErrorString(e)); eb_flags_destroy(f); return EB_ERR_NOMEM; }
    *out = f;
    return EB_OK;
}

int eb_flags_build_da_rank(eb_flags* f) {
    if (!f->d_da) return EB_OK;
    EB_HIP(hipSetDevice(f->device));
    uint64_t groups = (f->words + 7) / 8 + 1;
    uint64_t* d_vec = nullptr;
    EB_HIP(hipMalloc(&d_vec, groups * 4 * sizeof(uint64_t)));
    if (f->d_da_rank) { (void)hipFree(f->d_da_rank); f->d_da_rank = nullptr; }
    EB_HIP(hipMalloc(&f->d_da_rank, groups * sizeof(uint64_t)));
    unsigned grid = (unsigned)((groups + 255) / 256);
    hipLaunchKernelGGL(k_da_group_counts, dim3(grid), dim3(256), 0, f->stream, f->d_da, f->words, groups, d_vec);
    int rc = eb_scan4_exclusive(d_vec, groups, f->stream);
    if (rc != EB_OK) return rc;
    hipLaunchKernelGGL(k_take_first, dim3(grid), dim3(256), 0, f->stream, d_vec, groups, f->d_da_rank);
    EB_HIP(hipMemcpyAsync(&f->da_ones, f->d_da_rank + (groups - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, f->stream)); // the extra last group: ones before the end
    EB_HIP(hipStreamSynchronize(f->stream));
    EB_HIP(hipGetLastError());
    EB_HIP(hipFree(d_vec));
    return EB_OK;
}

__global__ void __launch_bounds__(256) k_fuse_flags(const uint64_t* __restrict__ thrK, const uint64_t* __restrict__ minima, uint64_t words,
                                                   uint64_t* __restrict__ flag) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) flag[w] = thrK[w] & ~minima[w];
}

int eb_flags_derive(eb_flags* f) {
    EB_HIP(hipSetDevice(f->device));
    if (f->complemented) { // a traversal (or the sum of its shards) left thrK / thrR in complement form
        hipLaunchKernelGGL(k_invert_thr, dim3(2048), dim3(256), 0, f->stream, f->d_thrK, f->d_thrR, f->n, eb_bit_alloc_words(f->n));
        f->complemented = false;
    }
    // phase 4 reads LCP_threshold[2i] only as `LCP_threshold[2i] && !LCP_minima[i]` (ebwt2InDel.cpp:1618,1632): one word instead of two
    hipLaunchKernelGGL(k_fuse_flags, dim3(2048), dim3(256), 0, f->stream, f->d_thrK, f->d_min, eb_bit_alloc_words(f->n), f->d_flag);
    EB_HIP(hipStreamSynchronize(f->stream));
    EB_HIP(hipGetLastError());
    int rc = eb_flags_build_da_rank(f);
    if (rc != EB_OK) return rc;
    f->derived = true;
    return EB_OK;
}

// derive: finish the vectors (inversion of the complement form, fused flag word, DA ranks) -- eb_flags_build; false: leave them in
// the form the traversal collected them in, for a caller that still has to combine shards (eb_flags_build_shard)
static int flags_build_impl(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                            eb_nav_stats* stats, bool complement, bool derive, bool chains) {
    if (!b1 || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (n_shards == 0 || shard >= n_shards) { eb_set_error("shard %u of %u", shard, n_shards); return EB_ERR_ARG; }
    if (K < 0 || K > 65534 || k_right < 0 || k_right > 65534) { eb_set_error("K and k_right must be below 65535"); return EB_ERR_ARG; }
    if (b1->n + (b2 ? b2->n : 0) >= (1ull << 40)) { eb_set_error("BWT of 2^40 symbols or more: node records hold 40-bit positions"); return EB_ERR_ARG; }
    if (b2 && b2->device != b1->device) { eb_set_error("both indexes must live on the same device"); return EB_ERR_ARG; }
    *out = nullptr;
    EB_HIP(hipSetDevice(b1->device));
    const uint64_t n = b1->n + (b2 ? b2->n : 0);
    const bool trace_alloc = eb_cfg().trace_nav;
    double ta = now_s();
    auto lap = [&](const char* what) {
        if (!trace_alloc) return;
        double t = now_s();
        fprintf(stderr, "[nav] %s: %.1f ms\n", what, (t - ta) * 1e3);
        ta = t;
    };
    eb_flags* f = nullptr;
    int rc = eb_flags_alloc(n, b1->device, b2 != nullptr, &f);
    if (rc != EB_OK) return rc;
    lap("flag vectors allocated + cleared");
    struct Guard { eb_flags* f; ~Guard() { if (f) eb_flags_destroy(f); } } guard{f};

    // per traversal: the hot counter, the drain's state (device) with two snapshots of it (page-locked host), statistics rows
    struct NavBufs {
        NavCounters* cnt = nullptr;
        DrainState* state = nullptr;
        DrainState* h_state = nullptr;
        unsigned long long* rows = nullptr; // EB_NAV_STATS counters per workgroup (nav_grid() launches at most 2048)
        ~NavBufs() {
            if (cnt) (void)hipFree(cnt);
            if (state) (void)hipFree(state);
            if (h_state) (void)hipHostFree(h_state);
            if (rows) (void)hipFree(rows);
        }
        int alloc() {
            EB_HIP(hipMalloc(&cnt, sizeof(NavCounters)));
            EB_HIP(hipMemset(cnt, 0, sizeof(NavCounters)));
            EB_HIP(hipMalloc(&state, sizeof(DrainState)));
            EB_HIP(hipHostMalloc(&h_state, 2 * sizeof(DrainState)));
            EB_HIP(hipMalloc(&rows, 2048 * EB_NAV_STATS * sizeof(unsigned long long)));
            EB_HIP(hipMemset(rows, 0, 2048 * EB_NAV_STATS * sizeof(unsigned long long)));
            return EB_OK;
        }
    };
    // The two traversals are independent (leaves write leaf interiors, nodes write child borders: disjoint positions), so
    // they run side by side -- the leaves on a second stream driven by a second host thread, with counters, statistics
    // rows and stacks of their own: the short leaf launches fill the gaps of the node launches (0.06 s hidden at config 2).
    NavBufs nb1, nb2;
    if ((rc = nb1.alloc()) != EB_OK || (rc = nb2.alloc()) != EB_OK) return rc;
    hipStream_t st2 = nullptr;
    EB_HIP(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    struct Guard5 { hipStream_t s; ~Guard5() { (void)hipStreamDestroy(s); } } guard5{st2};

    eb_dev_index x1 = b1->dev(), x2 = b2 ? b2->dev() : b1->dev();
    eb_dev_flags df = f->dev();
    NavParams p{K, k_right, 1, nullptr, complement ? 1 : 0};
#ifdef EB_SCAN_EXPERIMENTS
    p.dbg = getenv("EB_NAV_DEBUG") ? (uint32_t)atoi(getenv("EB_NAV_DEBUG")) : 0u; // (profiling build only)
#endif
    const uint64_t split_min = eb_cfg().nav_split_min;
    hipStream_t st = f->stream;
    const uint64_t batch_max = eb_cfg().nav_batch;
    // one arena for both traversals: two stacks of half the arena's batches (EB_NAV_BATCH) of the largest record each
    const uint64_t item_bytes = b2 ? sizeof(Node2P) : sizeof(Node1P); // the largest record of the two traversals
    uint64_t arena_batches = eb_cfg().nav_arena_batches + 1; // observed peak: 5 batches pending
    uint64_t eff_batch = batch_max < n / 2 + 64 ? batch_max : n / 2 + 64;
    uint64_t arena_records = arena_batches * eff_batch + (n < (1ull << 20) ? n + 4 * eff_batch + 64 : 0);
    if (arena_records < (1ull << 16)) arena_records = 1ull << 16;
    uint64_t arena_bytes = arena_records * item_bytes + 65536;
    {
        size_t free_b = 0, total_b = 0;
        EB_HIP(hipMemGetInfo(&free_b, &total_b));
        if (arena_bytes > free_b / 2) arena_bytes = free_b / 2;
    }
    // The traversal's temporaries come from the library's scratch cache and go back to it: a freed multi-GB block is wiped by
    // the driver, and the next large hipMalloc in the process then waits for that wipe (seconds -- measured 3.4 s for
    // the 7.55 GB below right after a previous call's hipFree); recycled blocks cost nothing and phase 4 reuses them.
    const int device = b1->device;
    size_t arena_cap = 0;
    void* d_arena = eb_scratch_get(device, arena_bytes, &arena_cap);
    if (!d_arena) { eb_set_error("traversal arena allocation failed (%llu bytes)", (unsigned long long)arena_bytes); return EB_ERR_NOMEM; }
    struct Guard3 { int dev; void* p; size_t cap; ~Guard3() { eb_scratch_put(dev, p, cap); } } guard3{device, d_arena, arena_cap};
    size_t arena2_cap = 0;
    uint64_t arena2_bytes = arena_bytes / 2; // the leaf frontier stays far smaller than the node frontier
    void* d_arena2 = eb_scratch_get(device, arena2_bytes, &arena2_cap);
    struct Guard6 { int dev; void* p; size_t cap; ~Guard6() { if (p) eb_scratch_put(dev, p, cap); } } guard6{device, d_arena2, arena2_cap};
    const bool side_by_side = d_arena2 != nullptr; // (one after the other in one arena when memory is that short)
    lap("arena allocated");
    // Phase 4 follows: what its first job would pay before its scan can start -- scan planes, device
    // temporaries, a page-locked blob: ~50 ms at 7.55 G positions -- is paid HERE, on a side thread, beside the traversal, when the
    // device has the memory to spare (the traversal's own buffers are allocated by now).
    std::thread warm;
    struct WarmGuard { std::thread& t; ~WarmGuard() { if (t.joinable()) t.join(); } } warm_guard{warm};
    {
        size_t free_b = 0, total_b = 0;
        const uint64_t positions = n / (n_shards ? n_shards : 1); // a shard's rank scans its share of the closing positions afterwards
        f->phase4_positions = positions;
        const bool have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        // (two BWTs, one shard: the merged scan planes are built right behind the traversal -- their memory is taken now)
        if (have_info && b2 && derive && n_shards <= 1 && free_b > 4 * (n / 4)) {
            if (hipMalloc(&f->d_mxy_spare, eb_bit_alloc_words(n) * sizeof(uint4)) != hipSuccess) { f->d_mxy_spare = nullptr; (void)hipGetLastError(); }
            else free_b -= eb_bit_alloc_words(n) * sizeof(uint4);
        }
        if (have_info && free_b > 3 * (n / 4 + positions / 8 + (512ull << 20)))
            warm = std::thread([b1, b2, positions] { eb_phase4_warm(b1, b2, positions); });
    }
    // Threshold bits are collected either as one byte per position (plain stores; needs n bytes of spare HBM) or, when
    // memory is short, as interleaved bit words set with atomics; both are converted to the two bit vectors afterwards.
    uint64_t* d_kr = nullptr;
    uint8_t* d_lcpb = nullptr;
    const size_t lcpb_bytes = eb_bit_alloc_words(n) * 64;
    size_t thr_cap = 0;
    {
        size_t free_b = 0, total_b = 0;
        EB_HIP(hipMemGetInfo(&free_b, &total_b));
        // Complement form (the default): writers are few, so atomicOr on the interleaved bit words costs nothing and the n-byte
        // array, its clearing and its packing pass are saved.  The reference's form (EB_NAV_FORM=bytes / words, and the fallback
        // for inputs that are no eBWT) writes every position: bytes with plain stores when n spare bytes of HBM exist.
        bool want_bytes = !complement && eb_cfg().nav_form != EbConfig::NAV_WORDS && lcpb_bytes + (4ull << 30) < free_b;
        if (want_bytes) d_lcpb = static_cast<uint8_t*>(eb_scratch_get(device, lcpb_bytes, &thr_cap));
        if (!d_lcpb) {
            d_kr = static_cast<uint64_t*>(eb_scratch_get(device, eb_bit_alloc_words(n) * 16, &thr_cap));
            if (!d_kr) { eb_set_error("threshold word allocation failed"); return EB_ERR_NOMEM; }
        }
    }
    struct Guard4 { int dev; void* p; size_t cap; ~Guard4() { eb_scratch_put(dev, p, cap); } } guard4{device, d_lcpb ? (void*)d_lcpb : (void*)d_kr, thr_cap};
    if (d_lcpb) EB_HIP(hipMemsetAsync(d_lcpb, 0, lcpb_bytes, f->stream));
    else EB_HIP(hipMemsetAsync(d_kr, 0, eb_bit_alloc_words(n) * 16, f->stream));
    p.lcpb = d_lcpb;
    if (trace_alloc) { EB_HIP(hipStreamSynchronize(f->stream)); lap("threshold bytes allocated + cleared"); }
    eb_nav_stats s;
    memset(&s, 0, sizeof s);
    s.n = n;
    EB_HIP(hipStreamSynchronize(f->stream)); // the cleared threshold words, before either traversal touches them
    uint64_t launches_l = 0, launches_n = 0;
    const uint64_t room_hook = chains && eb_cfg().nav_chains == 3 ? 64 : 0; // tests: the chain-following walk is told it has room for 64 records per launch
    // phase 2 (ebwt2InDel.cpp:577-615 / :708-756); first_leaf = {[0,F_A), 0} (dna_bwt.hpp:313-317)
    auto phase_leaves = [&](hipStream_t sx, NavBufs& nb, void* arena, uint64_t bytes) -> int {
        int r;
        NavCounters* dc = nb.cnt;
        unsigned long long* rows = nb.rows;
        if (!b2) {
            Drain<Leaf1> d;
            d.device = b1->device; d.stream = sx;
            d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
            if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
            if ((r = d.push_host(Leaf1{0, b1->F[0], 0, 0})) != EB_OK) return r;
            r = d.run<-1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
                if (chains) hipLaunchKernelGGL(k_leaves1c, dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else hipLaunchKernelGGL(k_leaves<1>, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            });
            launches_l = d.launches;
            return r;
        }
        Drain<Leaf2P> d;
        d.device = b1->device; d.stream = sx;
        d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
        if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
        if ((r = d.push_host(pack_rec(Leaf2{0, b1->F[0], 0, b2->F[0], 0, {0, 0, 0}}))) != EB_OK) return r;
        r = d.run<1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
            if (chains) hipLaunchKernelGGL(k_leaves2c, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else hipLaunchKernelGGL(k_leaves<2>, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
        });
        launches_l = d.launches;
        return r;
    };
    // phase 3 (ebwt2InDel.cpp:622-673 / :764-823); root (dna_bwt.hpp:296-308)
    auto phase_nodes = [&](hipStream_t sx, NavBufs& nb, void* arena, uint64_t bytes) -> int {
        int r;
        NavCounters* dc = nb.cnt;
        unsigned long long* rows = nb.rows;
        if (!b2) {
            Drain<Node1P> d;
            d.device = b1->device; d.stream = sx;
            d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
            if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
            Node1 root{{0, b1->F[0], b1->F[1], b1->F[2], b1->F[3], b1->n}, 0, 0};
            if ((r = d.push_host(pack_rec(root))) != EB_OK) return r;
            r = d.run<-1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
                const bool super_lds = (b1->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS;
                if (chains && super_lds) hipLaunchKernelGGL((k_nodes1c<true>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else if (chains) hipLaunchKernelGGL((k_nodes1c<false>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else if (super_lds) hipLaunchKernelGGL((k_nodes1p<true>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else hipLaunchKernelGGL((k_nodes1p<false>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
            });
            launches_n = d.launches;
            return r;
        }
        Drain<Node2P> d;
        d.device = b1->device; d.stream = sx;
        d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
        if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
        Node2 root{{0, b1->F[0], b1->F[1], b1->F[2], b1->F[3], b1->n}, {0, b2->F[0], b2->F[1], b2->F[2], b2->F[3], b2->n}, 0, {0, 0, 0}};
        if ((r = d.push_host(pack_rec(root))) != EB_OK) return r;
        const bool super_fits = (b1->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS && (b2->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS;
        r = d.run<1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
            if (chains && super_fits) hipLaunchKernelGGL((k_nodes2c<true>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else if (chains) hipLaunchKernelGGL((k_nodes2c<false>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else if (super_fits) hipLaunchKernelGGL((k_nodes2p<true>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else hipLaunchKernelGGL((k_nodes2p<false>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
        });
        launches_n = d.launches;
        return r;
    };
    double t0 = now_s();
    if (side_by_side) {
        int rc_l = EB_OK;
        std::string err_l;
        double t_l = 0;
        std::thread leaf_thread([&] {
            (void)hipSetDevice(device);
            rc_l = phase_leaves(st2, nb2, d_arena2, arena2_bytes);
            if (rc_l != EB_OK) err_l = eb_last_error(); // the message is thread-local
            t_l = now_s() - t0;
        });
        rc = phase_nodes(st, nb1, d_arena, arena_bytes);
        s.seconds_nodes = now_s() - t0;
        leaf_thread.join();
        s.seconds_leaves = t_l;
        if (rc == EB_OK && rc_l != EB_OK) { eb_set_error("%s", err_l.c_str()); rc = rc_l; }
        if (rc != EB_OK) return rc;
    } else {
        if ((rc = phase_leaves(st, nb1, d_arena, arena_bytes)) != EB_OK) return rc;
        double t1 = now_s();
        s.seconds_leaves = t1 - t0;
        if ((rc = phase_nodes(st, nb1, d_arena, arena_bytes)) != EB_OK) return rc;
        s.seconds_nodes = now_s() - t1;
    }
    const uint64_t launches = launches_l + launches_n;
    if (d_lcpb) hipLaunchKernelGGL(k_pack_lcp, dim3(4096), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_lcpb), f->words, f->d_thrK, f->d_thrR);
    else hipLaunchKernelGGL(k_split_kr, dim3(2048), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_kr), f->words, f->d_thrK, f->d_thrR);
    lap("traversals");
    EB_HIP(hipStreamSynchronize(st));
    EB_HIP(hipGetLastError());
    lap("threshold bits packed");
    s.seconds_traversal = now_s() - t0;
    f->complemented = complement;
    if (derive && (rc = eb_flags_derive(f)) != EB_OK) return rc; // shards: eb_flags_finalize, after the reduction
    // two BWTs: phase 4 streams their scan planes merged by the document array, which is complete now (a sharded build's is not:
    // its first phase-4 job merges them)
    if (derive && b2 && n_shards <= 1 && f->d_mxy_spare && (rc = eb_flags_merge_planes(f, b1, b2)) != EB_OK) return rc;
    NavCounters c;
    memset(&c, 0, sizeof c);
    {
        std::vector<unsigned long long> rows(2048 * EB_NAV_STATS), rows2(2048 * EB_NAV_STATS);
        EB_HIP(hipMemcpy(rows.data(), nb1.rows, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        EB_HIP(hipMemcpy(rows2.data(), nb2.rows, rows2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < rows.size(); ++k) rows[k] += rows2[k];
        for (size_t r = 0; r < 2048; ++r) {
            c.leaves += rows[r * EB_NAV_STATS + 0];
            c.nodes += rows[r * EB_NAV_STATS + 1];
            c.lcp_values += rows[r * EB_NAV_STATS + 2];
            c.da_values += rows[r * EB_NAV_STATS + 3];
            c.minima += rows[r * EB_NAV_STATS + 4];
            c.ranks += rows[r * EB_NAV_STATS + 5];
            c.pos_sum += rows[r * EB_NAV_STATS + 6];
            c.pos_sq_sum += rows[r * EB_NAV_STATS + 7];
        }
    }
    s.leaves = c.leaves;
    s.nodes = c.nodes;
    s.lcp_values = c.lcp_values + (shard == 0 ? 1 : 0); // the reference starts its counter at 1 (ebwt2InDel.cpp:575)
    s.da_values = c.da_values;
    s.minima = c.minima;
    s.rank_queries = c.ranks;
    s.lcp_pos_sum = c.pos_sum;
    s.lcp_pos_sq_sum = c.pos_sq_sum;
    s.batches = launches;
    if (stats) *stats = s;
    guard.f = nullptr;
    *out = f;
    return EB_OK;
}

// Host-side round trip of the packed traversal records (the same __host__ __device__ code the kernels run): lets a
// test without a GPU exercise 40-bit positions beyond 2^32 and the saturating depth.
extern "C" int eb_selftest_pack_fields(int32_t n_fields, const uint64_t* fields, uint64_t depth, uint64_t* fields_out, uint64_t* depth_out,
                                       uint64_t* words_out, uint32_t* n_words) {
    if (!fields || !fields_out || !depth_out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    auto go = [&](auto packed, auto nf_tag) {
        constexpr int NF = decltype(nf_tag)::value;
        packed = pack_fields<NF>(fields, depth);
        *depth_out = unpack_fields<NF>(packed, fields_out);
        const uint32_t nw = (uint32_t)(sizeof(packed) / 8);
        if (n_words) *n_words = nw;
        if (words_out) memcpy(words_out, packed.w, sizeof(packed));
    };
    if (n_fields == 6) go(Node1P{}, std::integral_constant<int, 6>{});
    else if (n_fields == 12) go(Node2P{}, std::integral_constant<int, 12>{});
    else if (n_fields == 4) go(Leaf2P{}, std::integral_constant<int, 4>{});
    else { eb_set_error("record kinds have 4, 6 or 12 position fields"); return EB_ERR_ARG; }
    return EB_OK;
}

extern "C" int eb_nav_covers_every_slot_once(const eb_nav_stats* s) {
    if (!s) return 0;
    // positions 1 .. n-1, each exactly once: their number (the reference's counter starts at 1, ebwt2InDel.cpp:575), their
    // sum and the sum of their squares, modulo 2^64
    return s->lcp_values == s->n && (s->n == 0 || (s->lcp_pos_sum == eb_tri(s->n) && s->lcp_pos_sq_sum == eb_sq(s->n)));
}

// One-BWT node traversal with chains followed in registers (k_nodes1c; EB_NAV_CHAINS=0: k_nodes1p).  A launch of k_nodes1c may
// push more records than the planner left room for (it never has: see the kernel); the traversal then ends with an error and is
// walked again the round-3 way.
static int flags_build_chains(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                              eb_nav_stats* stats, bool complement, bool derive) {
    // EB_NAV_CHAINS (tests): 0 the fallback's kernels at once; 2 the first walk's result is thrown away as if it had overflowed; 3 the
    // first walk is told its stacks have room for 64 records per launch, so that it really overflows (records dropped in mid-launch,
    // error set on the device, the group queued ahead finding nothing to do, the leaf thread joined)
    const int mode = eb_cfg().nav_chains;
    const bool chains = mode != 0;
    int rc = flags_build_impl(b1, b2, K, k_right, shard, n_shards, out, stats, complement, derive, chains);
    if (mode == 2 && rc == EB_OK) { eb_flags_destroy(*out); *out = nullptr; rc = EB_NAV_OVERFLOW; }
    // only an overflow of the chain-following walk is answered by the other kernels; anything else (memory, a device error) would
    // fail the same way again
    if (chains && rc == EB_NAV_OVERFLOW) rc = flags_build_impl(b1, b2, K, k_right, shard, n_shards, out, stats, complement, derive, false);
    return rc == EB_NAV_OVERFLOW ? EB_ERR_STATE : rc;
}

extern "C" int eb_flags_build(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, eb_flags** out, eb_nav_stats* stats) {
    if (!out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    const bool complement = eb_cfg().nav_form == EbConfig::NAV_COMPLEMENT;
    eb_nav_stats s;
    int rc = flags_build_chains(b1, b2, K, k_right, 0, 1, out, &s, complement, true);
    if (rc != EB_OK) return rc;
    if (complement && !eb_nav_covers_every_slot_once(&s)) {
        // Not every LCP slot was written exactly once: the input is not the eBWT of a read collection.  The reference leaves
        // unwritten slots at 0 (vector<bool>(n,false), ebwt2InDel.cpp:571), which the complement form cannot know: walk again
        // the reference's way (writers at or above the thresholds set their bits).
        eb_flags_destroy(*out);
        *out = nullptr;
        rc = flags_build_chains(b1, b2, K, k_right, 0, 1, out, &s, false, true);
        if (rc != EB_OK) return rc;
    }
    if (stats) *stats = s;
    return EB_OK;
}

extern "C" int eb_flags_build_shard(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                                    eb_nav_stats* stats) {
    // shards collect the complement (see NavParams::below): their vectors still add up (one writer per marked bit), and
    // eb_flags_finalize inverts the sum.  Callers check that the shards' lcp_values add up to n (a valid eBWT).
    // ALWAYS shard form, never derived -- also for n_shards == 1: the callers' next step (all-reduce + eb_flags_finalize, or
    // eb_flags_export + eb_flags_import_shard_sum) is the same for any number of shards, and a one-shard build that came
    // back finished would be inverted a second time on the host-exchange paths (round-2 advisor finding).
    return flags_build_chains(b1, b2, K, k_right, shard, n_shards, out, stats, true, false);
}

extern "C" int eb_flags_device_vectors(eb_flags* f, uint64_t* d_vectors[4], uint64_t* words) {
    if (!f || !d_vectors || !words) { eb_set_error("null argument"); return EB_ERR_ARG; }
    d_vectors[0] = f->d_thrK;
    d_vectors[1] = f->d_thrR;
    d_vectors[2] = f->d_min;
    d_vectors[3] = f->d_da;
    *words = f->words;
    return EB_OK;
}

// ones in a bit vector: per-workgroup partial sums, added up by the host
__global__ void __launch_bounds__(256) k_popcount_words(const uint64_t* __restrict__ v, uint64_t words, unsigned long long* __restrict__ partial) {
    unsigned long long s = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) s += (unsigned long long)__popcll(v[w]);
    s = wave_sum(s);
    __shared__ unsigned long long sh[4];
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

extern "C" int eb_flags_popcounts(eb_flags* f, uint64_t ones[4]) {
    if (!f || !ones) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(f->device));
    const unsigned grid = 1024;
    unsigned long long* d_part = nullptr;
    EB_HIP(hipMalloc(&d_part, 4 * grid * sizeof(unsigned long long)));
    struct G { void* p; ~G() { (void)hipFree(p); } } g{d_part};
    const uint64_t* vec[4] = {f->d_thrK, f->d_thrR, f->d_min, f->d_da};
    EB_HIP(hipMemsetAsync(d_part, 0, 4 * grid * sizeof(unsigned long long), f->stream));
    for (int k = 0; k < 4; ++k)
        if (vec[k]) hipLaunchKernelGGL(k_popcount_words, dim3(grid), dim3(256), 0, f->stream, vec[k], f->words, d_part + (size_t)k * grid);
    std::vector<unsigned long long> h(4 * grid);
    EB_HIP(hipMemcpyAsync(h.data(), d_part, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, f->stream));
    EB_HIP(hipStreamSynchronize(f->stream));
    EB_HIP(hipGetLastError());
    for (int k = 0; k < 4; ++k) {
        ones[k] = 0;
        for (unsigned b = 0; b < grid; ++b) ones[k] += h[(size_t)k * grid + b];
    }
    return EB_OK;
}

extern "C" int eb_flags_finalize(eb_flags* f) {
    if (!f) { eb_set_error("null argument"); return EB_ERR_ARG; }
    return eb_flags_derive(f);
}

// '1' -> 1, anything else -> 0: 64 ASCII bytes -> one word (ebwt2InDel.cpp:1503-1508)
__global__ void __launch_bounds__(256) k_pack_da(const uint8_t* __restrict__ ascii, uint64_t len, uint64_t first_word, uint64_t n_words,
                                                uint64_t* __restrict__ da) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    uint64_t off = t * 64, w = 0;
    if (off + 64 <= len) {
        const uint4* p4 = reinterpret_cast<const uint4*>(ascii + off);
        for (int q = 0; q < 4; ++q) {
            uint4 v = p4[q];
            uint32_t d[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 4; ++k)
                for (int b = 0; b < 4; ++b)
                    w |= (uint64_t)(((d[k] >> (8 * b)) & 0xFFu) == '1') << (q * 16 + k * 4 + b);
        }
    } else {
        for (uint64_t k = 0; k < 64 && off + k < len; ++k) w |= (uint64_t)(ascii[off + k] == '1') << k;
    }
    da[first_word + t] = w;
}

// DA[from, n) = 1 (a document-array file that ends early on a '1', see eb_flags_set_da); bits at and beyond n stay 0
__global__ void __launch_bounds__(256) k_fill_da_tail(uint64_t* __restrict__ da, uint64_t from, uint64_t n) {
    const uint64_t w0 = from >> 6, w1 = (n - 1) >> 6;
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t m = ~0ull;
        if (w == w0) m &= ~0ull << (from & 63u);
        if (w == w1) m &= ~0ull >> (63u - ((n - 1) & 63u));
        da[w] |= m; // one thread per word
    }
}

// The document array from one of three places: a host buffer, the bytes of a file (fd), or ASCII bytes already in HBM (d_src).
// '1' -> 1, anything else -> 0 (ebwt2InDel.cpp:1503-1508).  A source shorter than the BWT: the reference's read() fails and
// leaves its one-byte buffer untouched, so every remaining slot repeats the LAST byte of the file (k_fill_da_tail); with an empty
// file that buffer was never written (undefined upstream): zeros here.
// Large host / file sources go through page-locked pieces filled by the loader's reader threads while earlier pieces are uploaded
// and packed, on two streams (eb_pieces.h: what the eBWT itself comes in by) -- round 4 copied the whole array from pageable
// memory in 256 MB pieces, each waited for: ~10 GB/s.
static int set_da_from(eb_flags* f, const uint8_t* host, int fd, const uint8_t* d_src, uint64_t n) {
    EB_HIP(hipSetDevice(f->device));
    const EbConfig& cfg = eb_cfg();
    const size_t bytes = eb_bit_alloc_words(f->n) * 8;
    if (!f->d_da) EB_HIP(hipMalloc(&f->d_da, bytes));
    EB_HIP(hipMemsetAsync(f->d_da, 0, bytes, f->stream));
    const uint64_t lim = n < f->n ? n : f->n;
    bool sticky_one = false;
    if (lim > 0 && lim < f->n) {
        uint8_t last = 0;
        if (host) last = host[lim - 1];
        else if (d_src) EB_HIP(hipMemcpy(&last, d_src + lim - 1, 1, hipMemcpyDeviceToHost));
        else if (pread(fd, &last, 1, (off_t)(lim - 1)) != 1) { eb_set_error("read error in the document array"); return EB_ERR_IO; }
        sticky_one = last == '1';
    }
    // a document array means phase 4 in two-sample form (twice the chains per cluster): its first job's reservations are made now,
    // beside the upload (the traversal reserved for one sample: it could not know)
    std::thread reserve([dev = f->device, n_pos = f->phase4_positions ? f->phase4_positions : f->n] { eb_phase4_reserve(dev, n_pos, true); });
    struct ReserveGuard { std::thread& t; ~ReserveGuard() { if (t.joinable()) t.join(); } } reserve_guard{reserve};
    auto pack = [&](const uint8_t* d_piece, uint64_t pos, uint64_t len, hipStream_t st) {
        const uint64_t n_words = (len + 63) / 64;
        hipLaunchKernelGGL(k_pack_da, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_piece, len, pos / 64, n_words, f->d_da);
    };
    int rc = EB_OK;
    if (d_src) {
        const uint64_t PIECE = 1ull << 36; // (one launch's grid holds 2^32 workgroups of 256 words)
        for (uint64_t pos = 0; pos < lim; pos += PIECE) pack(d_src + pos, pos, lim - pos < PIECE ? lim - pos : PIECE, f->stream);
    } else if (lim) {
        const bool pinned_path = lim >= cfg.pinned_min || !host;
        const uint64_t PIECE = pinned_path ? cfg.pinned_piece : (256ull << 20); // multiples of 64 positions
        const int SLOTS = 4;
        const uint64_t stage_bytes = lim < PIECE ? lim : PIECE;
        const size_t n_pieces = (size_t)((lim + PIECE - 1) / PIECE);
        uint8_t *d_stage[4] = {nullptr, nullptr, nullptr, nullptr}, *h_stage[4] = {nullptr, nullptr, nullptr, nullptr};
        uint64_t h_cap[4] = {0, 0, 0, 0};
        size_t d_cap[4] = {0, 0, 0, 0};
        hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};
        hipStream_t up[2] = {f->stream, nullptr};
        hipError_t e = hipSuccess;
        {
            PieceReaders readers; // (joined before the pieces it fills go back to the pool)
            readers.src = host;
            readers.fd = fd;
            auto piece_len = [&](size_t k) -> uint64_t { const uint64_t pos = (uint64_t)k * PIECE; return lim - pos < PIECE ? lim - pos : PIECE; };
            size_t posted = 0;
            if (pinned_path) readers.start(cfg.copy_threads, n_pieces);
            for (int i = 0; i < SLOTS && (size_t)i < n_pieces && e == hipSuccess; ++i) {
                if (pinned_path) {
                    h_stage[i] = static_cast<uint8_t*>(eb_pinned_get(stage_bytes, &h_cap[i]));
                    if (!h_stage[i]) { e = hipErrorOutOfMemory; break; }
                    readers.post((int)posted, h_stage[i], (uint64_t)posted * PIECE, piece_len(posted));
                    ++posted;
                }
                d_stage[i] = static_cast<uint8_t*>(eb_scratch_get(f->device, stage_bytes, &d_cap[i]));
                if (!d_stage[i]) { e = hipErrorOutOfMemory; break; }
                e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
            }
            if (e == hipSuccess && n_pieces > 1 && hipStreamCreateWithFlags(&up[1], hipStreamNonBlocking) != hipSuccess) up[1] = nullptr;
            if (!up[1]) up[1] = f->stream;
            if (e == hipSuccess && up[1] != f->stream) {
                // the second stream starts behind the memset of the array
                hipEvent_t zeroed = nullptr;
                e = hipEventCreateWithFlags(&zeroed, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(zeroed, f->stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(up[1], zeroed, 0);
                if (zeroed) (void)hipEventDestroy(zeroed);
            }
            for (size_t k = 0; k < n_pieces && e == hipSuccess && rc == EB_OK; ++k) {
                const hipStream_t st = up[k & 1];
                const int slot = (int)(k % SLOTS);
                const uint64_t pos = (uint64_t)k * PIECE, len = piece_len(k);
                const uint8_t* src = host ? host + pos : nullptr;
                if (pinned_path) {
                    if (!readers.wait((int)k)) { eb_set_error("read error in the document array"); rc = EB_ERR_IO; break; }
                    src = h_stage[slot];
                } else if (k >= (size_t)SLOTS && (e = hipEventSynchronize(done[slot])) != hipSuccess) break;
                if ((e = hipMemcpyAsync(d_stage[slot], src, len, hipMemcpyHostToDevice, st)) != hipSuccess) break;
                pack(d_stage[slot], pos, len, st);
                if ((e = hipEventRecord(done[slot], st)) != hipSuccess) break;
                if (pinned_path && k >= 1 && posted < n_pieces) { // the slot of the piece BEFORE this one takes the next piece (as in the index loader)
                    const int free_slot = (int)((k - 1) % SLOTS);
                    if ((e = hipEventSynchronize(done[free_slot])) != hipSuccess) break;
                    readers.post((int)posted, h_stage[free_slot], (uint64_t)posted * PIECE, piece_len(posted));
                    ++posted;
                }
            }
            if (up[1] != f->stream) {
                if (hipStreamSynchronize(up[1]) != hipSuccess && e == hipSuccess) e = hipErrorUnknown;
                (void)hipStreamDestroy(up[1]);
            }
            if (hipStreamSynchronize(f->stream) != hipSuccess && e == hipSuccess) e = hipErrorUnknown;
        }
        for (int i = 0; i < 4; ++i) {
            if (d_stage[i]) eb_scratch_put(f->device, d_stage[i], d_cap[i]);
            if (h_stage[i]) eb_pinned_put(h_stage[i], h_cap[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
        if (rc == EB_OK && e != hipSuccess) { eb_set_error("document array upload failed: %s", hipGetErrorString(e)); rc = EB_ERR_DEVICE; }
        if (rc != EB_OK) return rc;
    }
    if (sticky_one) hipLaunchKernelGGL(k_fill_da_tail, dim3(2048), dim3(256), 0, f->stream, f->d_da, lim, f->n);
    EB_HIP(hipStreamSynchronize(f->stream));
    EB_HIP(hipGetLastError());
    return eb_flags_build_da_rank(f);
}

extern "C" int eb_flags_set_da(eb_flags* f, const uint8_t* ascii01, uint64_t n) {
    if (!f || (!ascii01 && n)) { eb_set_error("null argument"); return EB_ERR_ARG; }
    static const uint8_t empty = 0;
    return set_da_from(f, ascii01 ? ascii01 : &empty, -1, nullptr, n);
}

extern "C" int eb_flags_set_da_device(eb_flags* f, const void* d_ascii01, uint64_t n) {
    if (!f || (!d_ascii01 && n)) { eb_set_error("null argument"); return EB_ERR_ARG; }
    static const uint8_t empty = 0;
    if (!n) return set_da_from(f, &empty, -1, nullptr, 0);
    return set_da_from(f, nullptr, -1, static_cast<const uint8_t*>(d_ascii01), n);
}

extern "C" int eb_flags_set_da_file(eb_flags* f, const char* path) {
    if (!f || !path) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { eb_set_error("Error: could not read file %s", path); return EB_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); eb_set_error("cannot stat %s", path); return EB_ERR_IO; }
    static const uint8_t empty = 0;
    const uint64_t n = (uint64_t)st.st_size;
    int rc = n ? set_da_from(f, nullptr, fd, nullptr, n) : set_da_from(f, &empty, -1, nullptr, 0);
    close(fd);
    return rc;
}

extern "C" uint64_t eb_flags_words(const eb_flags* f) { return f ? f->words : 0; }
extern "C" int eb_flags_has_da(const eb_flags* f) { return f && f->d_da ? 1 : 0; }

extern "C" int eb_flags_export(eb_flags* f, uint64_t* thrK, uint64_t* thrR, uint64_t* minima, uint64_t* da) {
    if (!f) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(f->device));
    size_t bytes = f->words * 8;
    if (thrK) EB_HIP(hipMemcpy(thrK, f->d_thrK, bytes, hipMemcpyDeviceToHost));
    if (thrR) EB_HIP(hipMemcpy(thrR, f->d_thrR, bytes, hipMemcpyDeviceToHost));
    if (minima) EB_HIP(hipMemcpy(minima, f->d_min, bytes, hipMemcpyDeviceToHost));
    if (da) {
        if (!f->d_da) { eb_set_error("flags carry no document array"); return EB_ERR_STATE; }
        EB_HIP(hipMemcpy(da, f->d_da, bytes, hipMemcpyDeviceToHost));
    }
    return EB_OK;
}

static int flags_import_impl(uint64_t n, int device, const uint64_t* thrK, const uint64_t* thrR, const uint64_t* minima, const uint64_t* da,
                             bool complemented, eb_flags** out) {
    if (!out || !thrK || !thrR || !minima) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *out = nullptr;
    eb_flags* f = nullptr;
    int rc = eb_flags_alloc(n, device, da != nullptr, &f);
    if (rc != EB_OK) return rc;
    f->complemented = complemented;
    size_t bytes = f->words * 8;
    hipError_t e = hipMemcpy(f->d_thrK, thrK, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f->d_thrR, thrR, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f->d_min, minima, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && da) e = hipMemcpy(f->d_da, da, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { eb_set_error("flag upload failed: %s", hipGetErrorString(e)); eb_flags_destroy(f); return EB_ERR_DEVICE; }
    if ((rc = eb_flags_derive(f)) != EB_OK) { eb_flags_destroy(f); return rc; }
    *out = f;
    return EB_OK;
}

extern "C" int eb_flags_import(uint64_t n, int device, const uint64_t* thrK, const uint64_t* thrR, const uint64_t* minima,
                               const uint64_t* da, eb_flags** out) {
    return flags_import_impl(n, device, thrK, thrR, minima, da, false, out);
}

extern "C" int eb_flags_import_shard_sum(uint64_t n, int device, const uint64_t* thrK_below, const uint64_t* thrR_below, const uint64_t* minima,
                                         const uint64_t* da, eb_flags** out) {
    return flags_import_impl(n, device, thrK_below, thrR_below, minima, da, true, out);
}

extern "C" void eb_flags_destroy(eb_flags* f) {
    if (!f) return;
    if (f->device >= 0) (void)hipSetDevice(f->device);
    uint64_t* v[6] = {f->d_thrK, f->d_thrR, f->d_min, f->d_da, f->d_da_rank, f->d_flag};
    for (auto p : v)
        if (p) (void)hipFree(p);
    if (f->d_mxy) (void)hipFree(f->d_mxy);
    if (f->d_mxy_spare) (void)hipFree(f->d_mxy_spare);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}
