// eb_nav_one.h -- traversal kernels of ONE BWT (and the generic leaf kernel of both modes): k_leaves, k_nodes1p (every node through
// the stack: the fallback), k_nodes1c / k_leaves1c (chains followed in registers).  Included by eb_navigate.hip only.
#pragma once
#include "eb_nav_common.h"

namespace {

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

// ---- phase 3, one BWT, CHAINS FOLLOWED IN REGISTERS (round 4).  The suffix tree of a read collection is mostly unary under Weiner
// links: W is right-maximal because some read ends behind it ('#') while the others go on with one letter, and the same holds for
// cW, c'cW, ... until the reads run out -- ~67 nodes in a row whose only kept child is the next one (6.1 G nodes in 92 M records through
// the stack at 7.55 G positions; 16 in a row on reads with 1 % errors: profiles/r05_input_stats.txt).  k_nodes1p sends every one of them through the stack: 32 bytes written, 32 read, packed, staged,
// unpacked.  Here a quad KEEPS the first kept child of its node -- lane jc's six boundaries reach the quad through ds_bpermute -- and
// expands it in its next turn; only the other kept children (one node in a hundred has any) go to the stack, and a quad takes a
// new record from the launch's input only when its chain ends.  A quad runs TWO chains alternately, so that the block loads of
// one are in flight while the other computes (the distance k_nodes1p keeps between its stages A and D), plus one input record
// fetched ahead.  Input records are handed out through a cursor (NavCounters::taken): a wave takes a chunk with one atomic, its
// quads take records from the chunk as their chains end.  Same bits, same counters: every node is still expanded exactly once.
// A launch may now push more than four records per input record; the planner's room rule no longer bounds that, so an overflow
// (never seen: the whole traversal pushes fewer records than ONE stack holds) ends the traversal with an error and the caller
// walks again with k_nodes1p (flags_build_chains).
template <bool SUPER_LDS>
// (four waves per SIMD = 128 registers: at three -- 130 registers, nothing spilled -- a traversal takes 0.274 instead of 0.260 s, so the
// kernel is hardly short of waves in flight; at five -- 96 registers -- it spills and takes 0.87 s.  profiles/r05_sq_nav.txt)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_nodes1c(eb_dev_index x1, eb_dev_flags f, uint64_t* __restrict__ kr, NavParams p, const NavLaunch* __restrict__ la,
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
        // the node and its rank queries as the reference counts them (it shares the query of equal neighbouring boundaries): one per
        // DISTINCT boundary = 1 + the non-empty child intervals -- lane j knows about interval j, lane 0 adds the first boundary and T's
        n_rank += (hi > lo ? 1u : 0u) + (j == 0 ? 1u + (N.b[5] != N.b[4] ? 1u : 0u) : 0u);
        n_node += j == 0 ? 1u : 0u;
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
                    // letter j's count before the superblock: read ONCE per node -- its six boundaries lie in one superblock of 2^32
                    // symbols unless the node straddles a superblock boundary (then per boundary: wave-uniform choice)
                    auto super_of = [&](uint64_t i) -> uint64_t {
                        if constexpr (SUPER_LDS) return s_super[((i >> EB_SUPER_SHIFT) << 2) + j];
                        else return x1.super[((i >> EB_SUPER_SHIFT) << 2) + j];
                    };
                    const bool one_super = __ballot((N.b[5] >> EB_SUPER_SHIFT) != (N.b[0] >> EB_SUPER_SHIFT)) == 0ull;
                    const uint64_t base0 = Fj + super_of(N.b[0]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        uint32_t pc = eb_chunk_counts(c[k], eb_prefix_mask32((uint32_t)(N.b[k] & 127u), j));
                        pc = eb_quad_sum(pc);
                        const uint64_t base = one_super ? base0 : Fj + super_of(N.b[k]);
                        child.b[k] = (base + c[k].w) + ((pc >> (8u * j)) & 0xFFu);
                    }
                    for (int k = 0; k < 5; ++k) kids += child.b[k + 1] > child.b[k];
                    child.depth = N.depth + 1;
                    const bool keep = act && kids >= 2; // number_of_children (include.hpp:760-768)
                    // the quad goes on with its first kept child; the others travel through the stack
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return pack_rec(child); }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    // lane jc's six boundaries to the whole quad: twelve ds_bpermute (the LDS crossbar, next to idle here) instead of six
                    // masked quad-ORs of 64 bits -- 36 vector instructions per turn in a kernel whose vector ALUs are busy 92 % of the
                    // time (profiles/r05_sq_nav.txt).
                    if (jc < 4u) { // (uniform inside a quad: the lanes a lane reads from are active with it)
                        const int src = (int)(((lane & ~3u) | jc) << 2);
#pragma unroll
                        for (int k = 0; k < 6; ++k) N.b[k] = eb_bpermute64(src, child.b[k]);
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
// the leaves W#, cW#, c'cW#, ... form a chain as long as the reads (0.58 G leaves in 16 M records through the stack at 7.55 G positions, which
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


}  // namespace
