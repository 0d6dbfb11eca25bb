// eb_nav_two.h -- traversal kernels of TWO BWTs (merged leaves / nodes): k_leaves2c, k_nodes2p (the fallback), k_nodes2c.
// Included by eb_navigate.hip only.
#pragma once
#include "eb_nav_common.h"

namespace {

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
            // rank queries as the reference counts them (equal neighbouring boundaries share one): one per DISTINCT boundary of each BWT =
            // 1 + its non-empty child intervals; the lane that looks at interval k counts it, lane 0 adds the two first boundaries
            n_rank += (s1 ? 1u : 0u) + (s2 ? 1u : 0u);
        }
        if (j == 0) { ++n_node; n_rank += 2; }
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
                    // letter j's counts before the superblocks: read ONCE per node and BWT -- a node's six boundaries lie in one
                    // superblock of 2^32 symbols unless it straddles a superblock boundary (then per boundary: wave-uniform choice)
                    auto super_a = [&](uint64_t i) -> uint64_t {
                        if constexpr (SUPER_LDS) return s_super[0][((i >> EB_SUPER_SHIFT) << 2) + j];
                        else return x1.super[((i >> EB_SUPER_SHIFT) << 2) + j];
                    };
                    auto super_b = [&](uint64_t i) -> uint64_t {
                        if constexpr (SUPER_LDS) return s_super[1][((i >> EB_SUPER_SHIFT) << 2) + j];
                        else return x2.super[((i >> EB_SUPER_SHIFT) << 2) + j];
                    };
                    const bool one_super = __ballot((N.a[5] >> EB_SUPER_SHIFT) != (N.a[0] >> EB_SUPER_SHIFT) || (N.b[5] >> EB_SUPER_SHIFT) != (N.b[0] >> EB_SUPER_SHIFT)) == 0ull;
                    const uint64_t base_a = F1 + super_a(N.a[0]), base_b = F2 + super_b(N.b[0]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if ((mc >> k) & 1u) cur_a = ca[k]; // else: the block of the boundary to the left
                        if ((mc >> (6 + k)) & 1u) cur_b = cb[k];
                        const uint64_t ia = N.a[k], ib = N.b[k];
                        uint32_t pa = eb_chunk_counts(cur_a, eb_prefix_mask32((uint32_t)(ia & 127u), j));
                        uint32_t pb = eb_chunk_counts(cur_b, eb_prefix_mask32((uint32_t)(ib & 127u), j));
                        pa = eb_quad_sum(pa);
                        pb = eb_quad_sum(pb);
                        const uint64_t ba = one_super ? base_a : F1 + super_a(ia), bb = one_super ? base_b : F2 + super_b(ib);
                        child.a[k] = (ba + cur_a.w) + ((pa >> (8u * j)) & 0xFFu);
                        child.b[k] = (bb + cur_b.w) + ((pb >> (8u * j)) & 0xFFu);
                    }
                    for (int k = 0; k < 5; ++k) kids += (child.a[k + 1] > child.a[k]) || (child.b[k + 1] > child.b[k]); // include.hpp:784-792
                    child.depth = N.depth + 1;
                    const bool keep = act && kids >= 2;
                    const uint32_t km = eb_quad_or(keep ? (1u << j) : 0u);
                    const uint32_t jc = follow && km ? (uint32_t)__ffs((int)km) - 1u : 4u;
                    nav_push_or_spill<Rec, STAGE>(keep && j != jc, [&] { return pack_rec(child); }, s_buf, &s_count, out, out_cap, cnt, lane);
                    bool want = false;
                    if (jc < 4u) { // (uniform inside a quad: the lanes a lane reads from are active with it)
                        // lane jc's twelve boundaries to the whole quad through the LDS crossbar (ds_bpermute: no vector-ALU work) instead
                        // of twelve masked quad-ORs of 64 bits -- 72 vector instructions per turn of a kernel whose vector ALUs are busy 89 %
                        // of the time (profiles/r05_sq_nav.txt)
                        const int src = (int)(((lane & ~3u) | jc) << 2);
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            N.a[k] = eb_bpermute64(src, child.a[k]);
                            N.b[k] = eb_bpermute64(src, child.b[k]);
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


}  // namespace
