// eb_scan_kernels.hip -- phase 4, first half: the linear cluster scan over the BWT and the ordering of its survivors.
//
// Replaces the loops of run_one_dataset (ebwt2InDel.cpp:1609-1655), run_two_datasets (:1395-1445),
// run_two_datasets_da (:1510-1560) and the filters of find_variants (:941-1005, :840-934, :1013-1096).
//
//   k_scan           one lane per 64 positions: flag = thrK & ~minima; every 1->0 transition closes a cluster [b,e).
//                    Clusters of length >= 2*mcov_out get their symbol histogram (TERM counts as 'A',
//                    include.hpp:275-289) and the frequent-symbol test; the few survivors are handed over in BWT-ordered
//                    batches (or, the fallback, appended to one list).
//   k_batch_resolve  long-run candidates get their histogram by rank differences; the last workgroup to finish scans the
//                    batches' counts and lays out the job's call set (JobLayout) -- on the device, so that the host enqueues
//                    the whole job without waiting for a count.
//   k_batch_compact  every batch sorts its few dozen members into the dense list and writes their chain offsets and the
//                    work list of the left-context extension.
// The context kernels are in eb_context.hip, the job state machine in eb_scan_job.cpp.
#include <hipcub/hipcub.hpp>

#include "eb_scan_internal.h"

namespace {

// Work-skipping switches for A/B timing of k_scan's parts exist only in the profiling build (make EXPERIMENTS=1, its own
// output directory); in the shipped library EB_DBG() is the constant 0, the branches vanish and no environment variable
// can hollow out the kernel.
#ifdef EB_SCAN_EXPERIMENTS
#define EB_DBG(p) ((p).dbg)
#else
#define EB_DBG(p) 0u
#endif

// the neighbour lane's value (lane i gets lane i-1's, lane 0 gets 0) as a DPP wave shift on a v_mov (v_mov_b32_dpp wave_shr:1)
// instead of __shfl_up, which compiles to ds_bpermute_b32, an LDS round trip: six of them per tile in k_scan (round 4: -1 % on the
// scan and on the step, profiles/r04_dpp_shift_ab.txt)
__device__ __forceinline__ uint64_t eb_wave_shr1_64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t flag_word(const eb_dev_flags& f, uint64_t w) { return f.flag[w]; }

__device__ __forceinline__ uint32_t frequent_mask(const uint32_t* cnt, uint32_t mcov) {
    uint32_t m = 0;
    for (int c = 0; c < 4; ++c) m |= (cnt[c] >= mcov ? 1u : 0u) << c;
    return m;
}

// histogram of [b,e) by rank difference; TERM rows fall into 'A'
__device__ __forceinline__ void hist_by_rank(const eb_dev_index& x, uint64_t b, uint64_t e, uint32_t* cnt) {
    rank4_t rb = eb_rank4(x, b), re = eb_rank4(x, e);
    uint64_t acgt = 0;
    for (int c = 0; c < 4; ++c) { cnt[c] = (uint32_t)(re.v[c] - rb.v[c]); acgt += re.v[c] - rb.v[c]; }
    cnt[0] += (uint32_t)((e - b) - acgt);
}

// mode 3: histogram split by DA (counts[DA[i]][base_to_int(bwt[i])], ebwt2InDel.cpp:1019)
__device__ __forceinline__ void hist_by_da(const eb_dev_index& x, const eb_dev_flags& f, uint64_t b, uint64_t e, uint32_t* c0, uint32_t* c1) {
    for (int c = 0; c < 4; ++c) c0[c] = c1[c] = 0;
    for (uint64_t ci = b >> 5; ci <= (e - 1) >> 5; ++ci) {
        uint4 ch = x.chunks[ci];
        uint64_t lo = ci << 5;
        uint32_t m = 0xFFFFFFFFu;
        if (b > lo) m &= 0xFFFFFFFFu << (uint32_t)(b - lo);
        if (e < lo + 32) m &= 0xFFFFFFFFu >> (uint32_t)(lo + 32 - e);
        uint32_t d = (uint32_t)(f.da[ci >> 1] >> ((ci & 1u) * 32u));
        uint32_t L[4] = {~ch.y & ~ch.x, ~ch.y & ch.x, ch.y & ~ch.x, ch.y & ch.x}; // TERM (x=y=0) lands in 'A'
        for (int c = 0; c < 4; ++c) {
            c1[c] += (uint32_t)__popc(L[c] & m & d);
            c0[c] += (uint32_t)__popc(L[c] & m & ~d);
        }
    }
}

// Letter masks of 64 symbols from their two low code planes; TERM (x=y=0) and padding land in 'A',
// which is exactly base_to_int's mapping (include.hpp:275-289).
struct Planes {
    uint64_t x, y;
    __device__ __forceinline__ uint64_t letter(int c) const {
        return c == 0 ? ~y & ~x : c == 1 ? ~y & x : c == 2 ? y & ~x : y & x;
    }
};

__device__ __forceinline__ Planes load_planes(const uint4* __restrict__ xy, uint64_t w) {
    uint4 v = xy[w]; // scan planes: 16 bytes instead of the two 16-byte chunks (TERM plane and counters are not needed here)
    Planes p;
    p.x = (uint64_t)v.x | ((uint64_t)v.y << 32);
    p.y = (uint64_t)v.z | ((uint64_t)v.w << 32);
    return p;
}

// Geometry of the run that closes at bit `bit` of word w: its begin, and which bits of the word itself (mcur) and of
// the previous word (mprv) it covers; in_regs = false when it starts further back (then b comes from walking the flag
// words in memory and the masks are not meaningful).
struct RunGeom {
    uint64_t b, mcur, mprv;
    bool in_regs;
};
__device__ __forceinline__ RunGeom run_geometry(const eb_dev_flags& f, uint64_t w, uint64_t fw, uint64_t pfw, uint32_t bit) {
    RunGeom g;
    const uint64_t base = w << 6;
    uint64_t below = (1ull << bit) - 1ull;
    uint64_t zeros = ~fw & below;
    g.mprv = 0;
    if (zeros) {
        uint32_t zb = 64u - (uint32_t)__clzll((long long)zeros); // first flagged bit after the nearest 0
        g.b = base + zb;
        g.mcur = below & (~0ull << zb);
        g.in_regs = true;
        return g;
    }
    if (~pfw) {
        uint32_t zb = 64u - (uint32_t)__clzll((long long)~pfw);   // 1..64
        g.b = base - 64 + zb;
        g.mcur = below;
        g.mprv = zb < 64 ? ~0ull << zb : 0ull;
        g.in_regs = true;
        return g;
    }
    // run longer than the two words at hand: walk back through memory
    g.mcur = 0;
    g.in_regs = false;
    uint64_t ww = w - 1;
    g.b = 0;
    while (ww > 0) {
        --ww;
        uint64_t z = ~flag_word(f, ww);
        if (z) { g.b = (ww << 6) + (64u - (uint32_t)__clzll((long long)z)); break; }
    }
    return g;
}

// K3+K4 fused.  One lane per 64 positions; all loads of a tile are issued up front and are independent of each other:
// the fused flag word (8 B), the scan planes of the same 64 positions (16 B) and, with a document array, its word
// (8 B).  What every run needs (CLUST_SIZES, counters) is done by the lane that streams the word, with bit tricks on
// registers.  What only a few runs need is NOT done there -- a wave would wait for its few busy lanes -- but handed
// over through two small per-wave queues in LDS and worked off 64 at a time, one run per lane:
//   hot queue : clusters that can hold two frequent letters at all (found bit-parallel: 1 in 147 on clean reads, 1 in 29 on reads
//               with 1 % errors, profiles/r05_input_stats.txt)                                -> two-popcount necessary test
//   todo queue: clusters that pass it (1 in 351 / 1 in 92)                                    -> histogram, frequent letters, record
// A queued run is a (word, bit) pair; its lane reloads the two flag words and planes (cache hits: just streamed).
// Runs longer than the two words (and the two-BWT mode, whose symbols live in other coordinates) take the memory path.
// Persistent grid; statistics leave as one row per workgroup.
template <int MODE, bool PREFETCH> // PREFETCH: the loads of tile it+1 are issued before tile it is worked on (see the main loop)
__global__ void __launch_bounds__(256) k_scan(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p, uint64_t w_lo,
                                             uint64_t w_count, Survivor* __restrict__ surv, uint64_t surv_cap,
                                             ScanCounters* __restrict__ cnt, unsigned long long* __restrict__ partial,
                                             Survivor* __restrict__ slots, uint32_t* __restrict__ bcount) {
    __shared__ unsigned int sh_hist[202]; // [201]: sink for runs that CLUST_SIZES does not count
    constexpr int STAGE = (int)EB_STAGE_SLOTS; // survivors are ~1 per 10^4 positions: a workgroup rarely fills this before its next flush
    const bool ordered = p.ordered != 0;
    __shared__ uint4 s_buf[STAGE * (sizeof(Survivor) / 16)];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long q_hot[4][128], q_todo[4][128]; // per wave; entry = word << 6 | bit
    Stager<Survivor, STAGE> stage{s_buf, &s_count, &s_base};
    for (int k = threadIdx.x; k < 202; k += blockDim.x) sh_hist[k] = 0;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // ones of a ballot mask in the lanes below this one
    auto ones_below = [](unsigned long long m) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    const uint64_t hi_pos = p.end < p.n ? p.end : p.n;
    // scan planes in the coordinates the flags live in: the index's own (modes 1, 3) or the two BWTs merged by the DA (mode 2)
    const uint4* __restrict__ xy = MODE == 2 ? f.mxy : x1.xy;
    const uint64_t n_bwt_words = MODE == 2 ? f.words : (x1.n_blocks << 1); // 64-symbol words backed by the planes
    uint32_t n_runs32 = 0, n_clusters32 = 0, n_surv32 = 0; // per lane: bounded by the positions a lane streams (64 per tile)
    uint64_t n_bases = 0;
    uint64_t cur_batch = 0;                           // ordered mode: the batch of the tile at hand (wave-uniform)
    uint32_t hot_n = 0, todo_n = 0;                   // queue fill, the same in every lane of the wave
    const uint64_t tiles = (w_count + 255) / 256;
    constexpr bool local_syms = true;

    // ---- a queued cluster: histogram + frequent-symbol filter -> survivor record
    auto analyse = [&](bool have, unsigned long long entry) {
        if (!have) return;
        const uint64_t w = entry >> 6;
        const uint32_t bit = (uint32_t)(entry & 63u);
        const uint64_t fw = flag_word(f, w), pfw = w ? flag_word(f, w - 1) : 0ull;
        RunGeom g = run_geometry(f, w, fw, pfw, bit);
        Survivor s;
        s.b = g.b; s.e = (w << 6) + bit; s.b0 = s.e0 = s.b1 = s.e1 = 0; s.fmask0 = s.fmask1 = 0; s.pad = 0;
        uint32_t c0[4], c1[4];
        bool keep;
        if (!g.in_regs) {
            // run longer than two words: its histogram needs memory; k_long_runs does that later so that the
            // streaming kernel carries no rank code (registers) for this rare case
            s.fmask0 = EB_NEEDS_HIST;
            keep = true;
            atomicAdd(&cnt->n_long, 1ull);
        } else {
            Planes cur = {0, 0}, prv = {0, 0};
            if (w < n_bwt_words) cur = load_planes(xy, w);
            if (g.mprv) prv = load_planes(xy, w - 1);
            if constexpr (MODE == 1) {
                for (int c = 0; c < 4; ++c) c0[c] = (uint32_t)__popcll(cur.letter(c) & g.mcur) + (uint32_t)__popcll(prv.letter(c) & g.mprv);
                s.fmask0 = frequent_mask(c0, p.mcov);
                int nf = __popc(s.fmask0);
                keep = nf >= 2 && !(p.q > 0 && nf > p.q); // ebwt2InDel.cpp:961-966
            } else {
                const uint64_t da_cur = f.da[w], da_prv = g.mprv ? f.da[w - 1] : 0ull;
                for (int c = 0; c < 4; ++c) {
                    uint64_t lc = cur.letter(c) & g.mcur, lp = prv.letter(c) & g.mprv;
                    c1[c] = (uint32_t)__popcll(lc & da_cur) + (uint32_t)__popcll(lp & da_prv);
                    c0[c] = (uint32_t)__popcll(lc & ~da_cur) + (uint32_t)__popcll(lp & ~da_prv);
                }
                s.fmask0 = frequent_mask(c0, p.mcov);
                s.fmask1 = frequent_mask(c1, p.mcov);
                int n0 = __popc(s.fmask0), n1 = __popc(s.fmask1);
                keep = n0 > 0 && n1 > 0 && !(p.q > 0 && (n0 > p.q || n1 > p.q)); // :1042-1052
                // Pairs are emitted only for contexts whose last characters -- the frequent symbols themselves -- differ
                // (:921, :1083), and an empty pair list leaves cluster_nr untouched (:1250): a cluster whose two sides
                // share one single frequent letter can never reach the output, so its contexts are not computed.
                keep = keep && __popc(s.fmask0 | s.fmask1) > 1;
                if (MODE == 2 && keep) { // the chains and the right context run in the two BWTs' own coordinates
                    s.b1 = eb_da_rank(f, s.b); s.e1 = eb_da_rank(f, s.e);
                    s.b0 = s.b - s.b1; s.e0 = s.e - s.e1;
                }
            }
        }
        if (keep && ordered) {
            // straight into the slots of the batch at hand: one cursor per batch (hundreds to thousands of them, shared by the one or
            // two workgroups whose tile runs meet in the batch: no contention).  A wave empties its queues before it enters the
            // next batch, so nothing of this batch is staged later.
            const uint32_t pos = atomicAdd(&bcount[cur_batch], 1u);
            if (pos < EB_BATCH_SLOTS) slots[cur_batch * EB_BATCH_SLOTS + pos] = s;
            else cnt->spilled = 1ull;
            ++n_surv32;
        } else if (keep) {
            // LDS staging (no per-survivor global atomic on the one list cursor); a full buffer spills straight to the list
            uint32_t slot = atomicAdd(&s_count, 1u);
            if (slot < (uint32_t)STAGE) {
                const uint4* src = reinterpret_cast<const uint4*>(&s);
                for (int k = 0; k < (int)(sizeof(Survivor) / 16); ++k) s_buf[slot * (sizeof(Survivor) / 16) + k] = src[k];
            } else {
                unsigned long long gl = atomicAdd(&cnt->nsurv, 1ull);
                if (gl < surv_cap) surv[gl] = s;
            }
        }
    };
    // Queue primitives.  Only this wave touches its queues and every lane executes every call (wave-uniform control
    // flow), so the fill counters live in registers; LDS operations of one wave complete in order.
    auto todo_push = [&](bool have, unsigned long long entry) {
        unsigned long long m = __ballot(have);
        if (have) q_todo[wave][todo_n + ones_below(m)] = entry;
        todo_n += (uint32_t)__popcll(m);
#ifdef EB_SCAN_EXPERIMENTS
        if (lane == 0 && m) atomicAdd(&cnt->n_todo, (unsigned long long)__popcll(m));
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (todo_n >= 64u) {
            todo_n -= 64u;
            unsigned long long e = q_todo[wave][todo_n + lane];
            __builtin_amdgcn_wave_barrier();
            if (!(EB_DBG(p) & 2u)) analyse(true, e);
        }
    };
    // a queued cluster that holds a change of symbol: two letters with >= mcov rows each need >= mcov rows that differ
    // from the majority letter, and a row that differs does so in code plane x or y:
    // (minority of x) + (minority of y) >= mcov is necessary.  Clusters with a stray TERM or two fall out here.
    auto filter = [&](bool have, unsigned long long entry) {
        bool pass = false;
        if (have) {
            const uint64_t w = entry >> 6;
            const uint32_t bit = (uint32_t)(entry & 63u);
            const uint64_t fw = flag_word(f, w), pfw = w ? flag_word(f, w - 1) : 0ull;
            // geometry without the memory walk: a run that fills the whole previous word goes on unfiltered
            const uint64_t below = (1ull << bit) - 1ull;
            const uint64_t zeros = ~fw & below;
            uint64_t mcur = below, mprv = 0;
            uint32_t len;
            pass = true;
            bool in_regs = true;
            if (zeros) {
                uint32_t zb = 64u - (uint32_t)__clzll((long long)zeros);
                mcur = below & (~0ull << zb);
                len = bit - zb;
            } else if (~pfw) {
                uint32_t zb = 64u - (uint32_t)__clzll((long long)~pfw);
                mprv = zb < 64 ? ~0ull << zb : 0ull;
                len = bit + 64u - zb;
            } else {
                in_regs = false;
                len = 0;
            }
            if (in_regs) {
                Planes cur = {0, 0}, prv = {0, 0};
                if (w < n_bwt_words) cur = load_planes(xy, w);
                if (mprv) prv = load_planes(xy, w - 1);
                uint32_t ones_x = (uint32_t)__popcll(cur.x & mcur) + (uint32_t)__popcll(prv.x & mprv);
                uint32_t ones_y = (uint32_t)__popcll(cur.y & mcur) + (uint32_t)__popcll(prv.y & mprv);
                uint32_t dx = ones_x < len - ones_x ? ones_x : len - ones_x;
                uint32_t dy = ones_y < len - ones_y ? ones_y : len - ones_y;
                pass = dx + dy >= p.mcov;
            }
        }
        todo_push(pass, entry);
    };
    auto hot_push = [&](bool have, unsigned long long entry) {
        unsigned long long m = __ballot(have);
        if (have) q_hot[wave][hot_n + ones_below(m)] = entry;
        hot_n += (uint32_t)__popcll(m);
#ifdef EB_SCAN_EXPERIMENTS
        if (lane == 0 && m) atomicAdd(&cnt->n_hot, (unsigned long long)__popcll(m));
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (hot_n >= 64u) {
            hot_n -= 64u;
            unsigned long long e = q_hot[wave][hot_n + lane];
            __builtin_amdgcn_wave_barrier();
            filter(true, e);
        }
    };

    // Everything a lane needs for one tile; all loads are independent of each other (the extra "previous
    // word" loads of lane 0 included), so they are in flight together.
    struct TileData {
        uint64_t fw, l0_pfw;
        Planes cur, l0_prv;
    };
    auto load_tile = [&](uint64_t tile) -> TileData {
        TileData d;
        d.fw = d.l0_pfw = 0;
        d.cur = {0, 0};
        d.l0_prv = {0, 0};
        uint64_t t = tile * 256 + threadIdx.x;
        if (tile < tiles && t < w_count) {
            uint64_t w = w_lo + t;
            d.fw = flag_word(f, w);
            if (local_syms && w < n_bwt_words) d.cur = load_planes(xy, w);
            if (lane == 0 && w > 0) {
                d.l0_pfw = flag_word(f, w - 1);
                if (local_syms) d.l0_prv = load_planes(xy, w - 1);
            }
        }
        return d;
    };
    // what is left in this wave's queues (fewer than 64 entries each): filtered / analysed now
    auto drain = [&]() {
        if (hot_n) {
            bool have = lane < hot_n;
            unsigned long long e = have ? q_hot[wave][lane] : 0ull;
            hot_n = 0;
            __builtin_amdgcn_wave_barrier();
            filter(have, e);
        }
        if (todo_n) {
            bool have = lane < todo_n;
            unsigned long long e = have ? q_todo[wave][lane] : 0ull;
            todo_n = 0;
            __builtin_amdgcn_wave_barrier();
            if (!(EB_DBG(p) & 2u)) analyse(have, e);
        }
    };
    uint32_t trips = 0;
    const uint64_t T = p.tiles_per_wg;
    const uint64_t iters = ordered ? T : (tiles > blockIdx.x ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0);
    // PREFETCH: a software pipeline over the tiles -- the loads of tile it+1 are issued before tile it is worked on (12 more
    // registers).  With the chip to itself (8 workgroups per CU) the kernel is not short of bytes in flight -- its ~480 vector
    // instructions per 64-position word are what it waits for -- and the pipeline LOSES: 1.144 against 1.090 ms on the same box
    // (round 3).  It is for the scan that shares the CUs with the context kernels at 3-4 workgroups per CU (EB_SCAN_OVERLAP),
    // where half the waves must cover the same memory latency.
    auto tile_of = [&](uint64_t it) -> uint64_t {
        return it < iters ? (ordered ? (uint64_t)blockIdx.x * T + it : (uint64_t)blockIdx.x + it * gridDim.x) : tiles; // tiles: nothing to load
    };
    TileData nxt;
    if (PREFETCH) nxt = load_tile(tile_of(0));
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t tile = tile_of(it);
        if (ordered) cur_batch = tile >> p.batch_shift; // batches of 2^batch_shift tiles of the RANGE, whichever workgroups scan them
        TileData td;
        if (PREFETCH) { td = nxt; nxt = load_tile(tile_of(it + 1)); }
        else td = load_tile(tile);
        uint64_t t = tile * 256 + threadIdx.x;
        bool active = t < w_count;
        uint64_t w = w_lo + (active ? t : 0);
        uint64_t fw = td.fw;
        Planes cur = td.cur, prv;
        // previous word: neighbour lane's registers; lane 0 fetched it from memory
        uint64_t pfw = eb_wave_shr1_64(fw);
        prv.x = eb_wave_shr1_64(cur.x);
        prv.y = eb_wave_shr1_64(cur.y);
        if (lane == 0) { pfw = td.l0_pfw; prv = td.l0_prv; }
        const uint64_t ea = active ? ~fw & ((fw << 1) | (pfw >> 63)) : 0ull; // positions e with flag[e]=0 and flag[e-1]=1
        // keep closing positions inside [begin,end) and below n (a run still open at n-1 is dropped, ebwt2InDel.cpp:1609-1655)
        uint64_t ends = ea;
        uint64_t base = w << 6;
        // (only the two waves whose 64 words hold an end of the scanned range need the masks: wave-uniform test on the wave's span)
        const uint64_t span_lo = (w_lo + tile * 256 + (uint64_t)wave * 64) << 6;
        if (span_lo < p.begin || span_lo + 4096 > hi_pos) {
            if (base < p.begin) ends &= (p.begin - base >= 64) ? 0ull : (~0ull << (p.begin - base));
            if (base + 64 > hi_pos) ends &= (hi_pos <= base) ? 0ull : (~0ull >> (base + 64 - hi_pos));
        }
        n_runs32 += (uint32_t)__popcll(ends);
        if (EB_DBG(p) & 4u) continue;
        // ---- bit-parallel: which closing runs can hold two frequent letters at all.  Two letters with >= mcov rows each
        // (:961; the two-sample modes need two distinct frequent letters to print anything, see analyse()) mean that at
        // least mcov rows differ from any one row of the cluster -- take its first row inside this word.  All runs of the
        // word at once:
        //   fill   : adding the first-row bit of every run to the run masks flips exactly the runs whose first row has
        //            a 1 (the carry stops at the zero that ends a run), which spreads that bit over the run;
        //   differ : rows whose (x, y) differ from the spread first row;
        //   >= mcov: the lowest differing row of every run is cleared mcov-1 times (a decrement that starts at each
        //            run's first row and is stopped by the closing zero), what is left marks runs with >= mcov;
        //   collect: the run masks plus a non-empty subset of them carry out into exactly their closing positions.
        // All but one cluster in 147 (one symbol, or a read start or two; one in 29 on reads with 1 % errors) end here.  Rows of a run that lie in the previous word
        // are counted separately and lower the bar for that one run.
        uint64_t hot = ends;
        {
            const uint64_t carry = pfw >> 63;
            const uint64_t first = (fw & ~((fw << 1) | carry)) | (fw & carry);   // first row of every run (part) in this word
            uint64_t fx = fw & ~(fw + (cur.x & first)), fy = fw & ~(fw + (cur.y & first));
            uint64_t d = ((cur.x ^ fx) | (cur.y ^ fy)) & fw;
            // the run that comes in from the previous word: its rows there (bits zbp..63) against the same first row
            uint32_t need0 = p.mcov;                 // differing rows still needed inside this word by that run
            uint64_t inj = 0;
            const uint64_t pz = ~pfw;
            const uint32_t zbp = pz ? 64u - (uint32_t)__clzll((long long)pz) : 0u;
            bool prv_hot = !pz;                      // the run reaches beyond the two words: decided later, from memory
            if (carry) {
                if (pz) {
                    const uint64_t top = ~0ull << zbp; // zbp <= 63 here
                    const uint64_t rx = (cur.x & 1ull) ? ~0ull : 0ull, ry = (cur.y & 1ull) ? ~0ull : 0ull;
                    const uint32_t rp = (uint32_t)__popcll(((prv.x ^ rx) | (prv.y ^ ry)) & top);
                    need0 = rp >= p.mcov ? 0u : p.mcov - rp;
                    // a run that lies in the previous word only (closes at bit 0) has no first row here: any change keeps it
                    const uint64_t pchg = ((prv.x ^ (prv.x >> 1)) | (prv.y ^ (prv.y >> 1))) & pfw & (pfw >> 1) & ~(1ull << 63);
                    prv_hot = (pchg >> zbp) != 0;
                } else {
                    need0 = 0;
                }
                if ((fw & 1ull) && need0 == 0) inj = 1ull;
            }
            const uint32_t clears = p.mcov > 9u ? 8u : p.mcov - 1u;             // fewer clears only weaken the filter
            const uint64_t first_in = first & ~1ull;                             // runs that begin in this word
            for (uint32_t k = 1; k <= clears; ++k) {
                uint64_t from = first_in | ((fw & carry & 1ull) && k < need0 ? 1ull : (first & 1ull & ~carry));
                d &= (d | ea) - from;
            }
            hot = (fw + d + inj) & ends;
            if ((ends & 1ull) && prv_hot) hot |= 1ull;
        }
        // ---- run statistics of every closing position (CLUST_SIZES, counters).  The flagged rows between two
        // consecutive closing positions are exactly the run that closes at the second one, so a run's length is the
        // difference of two prefix popcounts; the word is walked as two 32-bit halves (single-instruction bit ops, and a
        // wave waits for the busiest lane of each half rather than of the whole word).
        {
            uint64_t rest = ea;
            int32_t cprev = 0; // prefix popcount at the previous closing position (negative: rows carried in from the previous word)
            if (pfw >> 63) {
                const uint64_t pz = ~pfw;
                if (pz) cprev = -(int32_t)__clzll((long long)pz);
                else if (rest) {
                    // the first closing run reaches beyond the previous word: its length comes from memory
                    uint32_t bit = (uint32_t)__ffsll((long long)rest) - 1u;
                    rest &= rest - 1;
                    if ((ends >> bit) & 1ull) {
                        RunGeom g = run_geometry(f, w, fw, pfw, bit);
                        uint64_t len = base + bit - g.b;
                        n_bases += len;
                        if (len <= 200 && !(EB_DBG(p) & 1u)) atomicAdd(&sh_hist[len], 1u);
                        if (len >= p.min_len) ++n_clusters32;
                        else hot &= ~(1ull << bit);
                    }
                    cprev = (int32_t)bit; // all rows below `bit` are flagged
                }
            }
            uint32_t nb32 = 0, n_short = 0;
            const int32_t cstart = cprev; // the lengths of the runs walked below add up to the last prefix popcount minus this one
            if (EB_DBG(p) & 16u) rest = 0;
            // branch-free body: runs longer than 200 (and, in the two words that hold the ends of the scanned range, the
            // closing positions outside it) add to a sink slot of the histogram instead of skipping the add
            const bool edge = __ballot((rest & ~ends) != 0ull) != 0ull; // wave-uniform: some lane's word holds an end of the scanned range
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint32_t h = (uint32_t)(fw >> (32 * half));
                const int32_t cbase = half ? (int32_t)__popc((uint32_t)fw) : 0;
                uint32_t eb = (uint32_t)(rest >> (32 * half));
                const uint32_t er = (uint32_t)(ends >> (32 * half));
                uint32_t short_bits = 0;
                if (!edge) {
                    while (eb) {
                        const uint32_t low = eb & (0u - eb);
                        eb ^= low;
                        int32_t c = cbase + (int32_t)__popc(h & (low - 1u));
                        uint32_t len = (uint32_t)(c - cprev);
                        cprev = c;
                        atomicAdd(&sh_hist[len < 201u ? len : 201u], 1u);
                        short_bits |= len < p.min_len ? low : 0u;
                    }
                } else {
                    while (eb) {
                        const uint32_t low = eb & (0u - eb);
                        eb ^= low;
                        int32_t c = cbase + (int32_t)__popc(h & (low - 1u));
                        uint32_t len = (uint32_t)(c - cprev);
                        cprev = c;
                        const bool counted = (er & low) != 0u;
                        nb32 += counted ? len : 0u;
                        atomicAdd(&sh_hist[counted && len < 201u ? len : 201u], 1u);
                        short_bits |= counted && len < p.min_len ? low : 0u;
                    }
                }
                n_short += (uint32_t)__popc(short_bits);
                hot &= ~((uint64_t)short_bits << (32 * half));
            }
            if (!edge) nb32 = (uint32_t)(cprev - cstart); // (the edge form counts only the runs that close inside the scanned range)
            n_bases += nb32;
            n_clusters32 += (uint32_t)__popcll(ends & rest) - n_short; // closing positions walked here, minus the short runs
        }
        // ---- hand the clusters that may matter to the queues, one per lane and round
        if (EB_DBG(p) & 8u) hot = 0;
        for (;;) {
            bool have = hot != 0;
            if (!__ballot(have)) break;
            uint32_t bit = have ? (uint32_t)__ffsll((long long)hot) - 1u : 0u;
            if (have) hot &= hot - 1;
            unsigned long long entry = (w << 6) | bit;
            hot_push(have, entry);
        }
        // Survivors are rare (~1 per 10^4 positions = ~2 per tile): empty the staging buffer every 16 tiles.  No
        // barrier in between (they would serialise the tiles of a workgroup); a buffer that fills up earlier sends
        // the excess straight to the list (analyse), so the schedule never affects the result.
        if (ordered) {
            // a batch ends: every cluster that closed in its tiles must be in its slots before this wave moves on, so the wave
            // empties its queues (two part-filled rounds per wave and batch; no barrier: the batch cursor is global)
            if (((tile + 1) & (((uint64_t)1 << p.batch_shift) - 1u)) == 0 || it + 1 == iters) drain();
        } else if ((++trips & 15u) == 0) {
            __syncthreads();
            if (threadIdx.x == 0 && s_count > (uint32_t)STAGE) s_count = STAGE;
            stage.flush(surv, surv_cap, &cnt->nsurv);
        }
    }
    if (!ordered) {
        drain();
        __syncthreads();
        if (threadIdx.x == 0 && s_count > (uint32_t)STAGE) s_count = STAGE;
        stage.flush(surv, surv_cap, &cnt->nsurv);
    }
    // Statistics leave the kernel as one row of plain stores per workgroup (k_scan_reduce adds the rows up): thousands of
    // waves hammering three counters in one cache line with atomics cost ~0.25 ms per launch, whatever the range size.
    uint64_t n_runs = n_runs32, n_clusters = n_clusters32, n_surv = n_surv32;
    for (int o = 32; o > 0; o >>= 1) {
        n_runs += __shfl_xor((unsigned long long)n_runs, o);
        n_clusters += __shfl_xor((unsigned long long)n_clusters, o);
        n_bases += __shfl_xor((unsigned long long)n_bases, o);
        n_surv += __shfl_xor((unsigned long long)n_surv, o);
    }
    __shared__ unsigned long long sh_red[4][4];
    if (lane == 0) { sh_red[0][threadIdx.x >> 6] = n_runs; sh_red[1][threadIdx.x >> 6] = n_clusters; sh_red[2][threadIdx.x >> 6] = n_bases; sh_red[3][threadIdx.x >> 6] = n_surv; }
    __syncthreads();
    unsigned long long* row = partial + (uint64_t)blockIdx.x * EB_SCAN_PARTIAL;
    if (threadIdx.x < 3) row[threadIdx.x] = sh_red[threadIdx.x][0] + sh_red[threadIdx.x][1] + sh_red[threadIdx.x][2] + sh_red[threadIdx.x][3];
    if (threadIdx.x == 3) row[204] = sh_red[3][0] + sh_red[3][1] + sh_red[3][2] + sh_red[3][3]; // survivors handed over by the ordered scan
    for (int k = threadIdx.x; k < 201; k += blockDim.x) row[3 + k] = (unsigned long long)sh_hist[k] * (unsigned long long)k; // CLUST_SIZES[len] += len
}

// column sums of the per-workgroup statistics rows -> ScanCounters (one workgroup per column)
__global__ void __launch_bounds__(256) k_scan_reduce(const unsigned long long* __restrict__ partial, uint32_t rows, ScanCounters* __restrict__ cnt, int ordered) {
    __shared__ unsigned long long sh[4];
    const uint32_t col = blockIdx.x;
    unsigned long long s = 0;
    for (uint32_t r = threadIdx.x; r < rows; r += blockDim.x) s += partial[(uint64_t)r * EB_SCAN_PARTIAL + col];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = sh[0] + sh[1] + sh[2] + sh[3];
        if (col == 0) cnt->runs_closed = s;
        else if (col == 1) cnt->clusters = s;
        else if (col == 2) cnt->bases = s;
        else if (col == 204) { if (ordered) cnt->nsurv = s; } // the unordered scan counts its survivors with the list cursor
        else cnt->hist[col - 3] = s;
    }
}

// Histogram + frequent-symbol filter of the clusters k_scan could not analyse from registers (runs longer than
// 64..128 positions).  Dropped ones get begin = ~0, which the sort moves behind the real survivors.
__device__ __forceinline__ uint32_t bucket_of(uint64_t b, uint64_t base);
// the histogram and frequent-letter test of one such candidate; false: dropped (begin = ~0)
__device__ __forceinline__ bool resolve_long_run(const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, Survivor& s) {
    uint32_t c0[4], c1[4];
    bool keep;
    if (p.mode == 1) {
        hist_by_rank(x1, s.b, s.e, c0);
        s.fmask0 = frequent_mask(c0, p.mcov);
        s.fmask1 = 0;
        int nf = __popc(s.fmask0);
        keep = nf >= 2 && !(p.q > 0 && nf > p.q);
    } else {
        if (p.mode == 2) {
            // symbols live in the two BWTs' own coordinates: rank differences on both (ebwt2InDel.cpp:846-847)
            s.b1 = eb_da_rank(f, s.b); s.e1 = eb_da_rank(f, s.e);
            s.b0 = s.b - s.b1; s.e0 = s.e - s.e1;
            hist_by_rank(x1, s.b0, s.e0, c0);
            hist_by_rank(x2, s.b1, s.e1, c1);
        } else {
            hist_by_da(x1, f, s.b, s.e, c0, c1);
        }
        s.fmask0 = frequent_mask(c0, p.mcov);
        s.fmask1 = frequent_mask(c1, p.mcov);
        int n0 = __popc(s.fmask0), n1 = __popc(s.fmask1);
        keep = n0 > 0 && n1 > 0 && !(p.q > 0 && (n0 > p.q || n1 > p.q)) && __popc(s.fmask0 | s.fmask1) > 1;
    }
    if (!keep) {
        s.b = ~0ull;
        s.fmask0 = s.fmask1 = 0;
    }
    return keep;
}
__global__ void __launch_bounds__(256) k_long_runs(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p, Survivor* __restrict__ surv, uint64_t n,
                                                  unsigned long long* __restrict__ dropped, uint64_t bucket_base, uint32_t* __restrict__ bucket_cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Survivor s = surv[i];
    if (s.fmask0 != EB_NEEDS_HIST) { // the usual survivor: only counted into its position bucket (see k_bucket_place)
        atomicAdd(&bucket_cnt[bucket_of(s.b, bucket_base)], 1u);
        return;
    }
    if (!resolve_long_run(x1, x2, f, p, s)) atomicAdd(dropped, 1ull);
    else atomicAdd(&bucket_cnt[bucket_of(s.b, bucket_base)], 1u);
    surv[i] = s;
}

// ---- ordered hand-over (ScanParams.ordered): the batches lie in BWT order in `slots`, batch b holding bcount[b] survivors in
// arrival order.  Three launches, no host wait in between -- the counts stay on the device:
// (1) k_batch_resolve, one wave per batch: long-run candidates get their histogram; every batch reports how many survivors it
//     keeps and how many left-context chains they need
__global__ void __launch_bounds__(256) k_batch_resolve(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p, Survivor* __restrict__ slots,
                                                      const uint32_t* __restrict__ bcount, uint64_t n_batches, uint32_t* __restrict__ kept,
                                                      uint32_t* __restrict__ bch, ScanCounters* __restrict__ cnt) {
    const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (b >= n_batches) return; // whole waves
    uint32_t c = bcount[b];
    if (lane == 0) atomicMax(&cnt->max_batch, c);
    if (c > EB_BATCH_SLOTS) { c = EB_BATCH_SLOTS; if (lane == 0) cnt->spilled = 1ull; } // (the scan has said so already)
    uint32_t mine = 0, chains = 0;
    for (uint32_t i = lane; i < c; i += 64u) {
        Survivor* q = slots + b * EB_BATCH_SLOTS + i;
        uint32_t fm0 = q->fmask0, fm1 = q->fmask1;
        bool keep = true;
        if (fm0 == EB_NEEDS_HIST) {
            Survivor s = *q;
            keep = resolve_long_run(x1, x2, f, p, s);
            if (!keep) atomicAdd(&cnt->dropped, 1ull);
            *q = s;
            fm0 = s.fmask0;
            fm1 = s.fmask1;
        }
        mine += keep ? 1u : 0u;
        chains += keep ? (uint32_t)(__popc(fm0) + __popc(fm1)) : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) { mine += (uint32_t)__shfl_xor((int)mine, o); chains += (uint32_t)__shfl_xor((int)chains, o); }
    if (lane == 0) { kept[b] = mine; bch[b] = chains; }
}

// (2) k_batch_offsets, ONE workgroup: exclusive sums of the batches' survivor and chain counts, then the layout of the job's
//     call set (JobLayout: every later kernel reads its bounds and section offsets from it), the image's header, and the first
//     part of the job's report, written straight into the host's page-locked memory.  A batch that overflowed leaves an empty
//     call set behind (the host does the job again with smaller batches).
__global__ void __launch_bounds__(256) k_batch_offsets(const uint32_t* __restrict__ kept, const uint32_t* __restrict__ bch, uint64_t n_batches,
                                                      uint32_t* __restrict__ boff, uint32_t* __restrict__ bchoff, ScanCounters* __restrict__ cnt, ScanParams p,
                                                      uint32_t lw, uint32_t rw, uint8_t* __restrict__ blob, JobReport* __restrict__ report) {
    __shared__ uint32_t wave_tot[2][4];
    __shared__ uint32_t carry[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_batches; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const uint32_t v0 = i < n_batches ? kept[i] : 0u, v1 = i < n_batches ? bch[i] : 0u;
        uint32_t s0 = v0, s1 = v1;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t0 = (uint32_t)__shfl_up((int)s0, o), t1 = (uint32_t)__shfl_up((int)s1, o);
            if (lane >= (uint32_t)o) { s0 += t0; s1 += t1; }
        }
        if (lane == 63) { wave_tot[0][wave] = s0; wave_tot[1][wave] = s1; }
        __syncthreads();
        uint32_t b0 = carry[0], b1 = carry[1];
        for (uint32_t w = 0; w < wave; ++w) { b0 += wave_tot[0][w]; b1 += wave_tot[1][w]; }
        if (i < n_batches) { boff[i] = b0 + s0 - v0; bchoff[i] = b1 + s1 - v1; }
        __syncthreads();
        if (threadIdx.x == 255) { carry[0] = b0 + s0; carry[1] = b1 + s1; }
        __syncthreads();
    }
    const bool spilled = cnt->spilled != 0ull;
    const uint32_t ns = spilled ? 0u : carry[0], nch = spilled ? 0u : carry[1];
    if (threadIdx.x == 0) {
        boff[n_batches] = ns;
        JobLayout L;
        eb_job_layout(L, ns, nch, lw, rw);
        L.spilled = spilled ? 1u : 0u;
        L.max_batch = cnt->max_batch;
        cnt->lay = L;
        report->lay = L;
        report->nsurv_raw = cnt->nsurv;
        report->runs_closed = cnt->runs_closed;
        report->clusters = cnt->clusters;
        report->bases = cnt->bases;
        report->dropped = cnt->dropped;
        uint64_t* h = reinterpret_cast<uint64_t*>(blob);
        h[0] = EB_CALLS_MAGIC; h[1] = ns; h[2] = nch; h[3] = (uint64_t)(uint32_t)p.k_left; h[4] = (uint64_t)(uint32_t)p.k_right; h[5] = h[6] = h[7] = 0;
        if (nch & 1u) *reinterpret_cast<uint32_t*>(blob + L.off_crec + (uint64_t)nch * sizeof(eb_chain)) = 0u; // the padding behind an odd number of 12-byte chains
    }
    for (int k = threadIdx.x; k < 201; k += blockDim.x) report->hist[k] = cnt->hist[k];
}

// (3) k_batch_compact, one workgroup per batch: puts its members in order (rank by counting: a few dozen keys in LDS), writes
//     them to their place in the dense list, their chain offsets (first_chain of the cluster record) and their entries of the
//     left-context extension's work list: one entry (cluster << 3 | side*4 + letter) per frequent symbol.
__global__ void __launch_bounds__(256) k_batch_compact(const Survivor* __restrict__ slots, const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ boff,
                                                      const uint32_t* __restrict__ bchoff, const ScanCounters* __restrict__ cnt, Survivor* __restrict__ sorted,
                                                      uint32_t* __restrict__ choff, uint64_t* __restrict__ chains) {
    __shared__ unsigned long long keys[EB_BATCH_SLOTS];
    __shared__ uint32_t s_fm[EB_BATCH_SLOTS];  // at sorted position r: fmask0 | fmask1 << 4
    __shared__ uint32_t s_off[EB_BATCH_SLOTS]; // chains of this batch before sorted position r
    __shared__ uint32_t wave_tot[4];
    if (cnt->lay.spilled) return;
    const uint64_t b = blockIdx.x;
    uint32_t c = bcount[b];
    c = c < EB_BATCH_SLOTS ? c : EB_BATCH_SLOTS;
    const uint32_t base = boff[b], cbase = bchoff[b];
    const Survivor* q = slots + b * EB_BATCH_SLOTS;
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) keys[i] = q[i].b;
    for (uint32_t i = threadIdx.x; i < EB_BATCH_SLOTS; i += blockDim.x) s_fm[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        const unsigned long long key = keys[i];
        if (key == ~0ull) continue; // a dropped long-run candidate
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; ++j) rank += keys[j] < key ? 1u : 0u; // begins of distinct clusters are distinct
        Survivor v = q[i];
        sorted[base + rank] = v;
        s_fm[rank] = (v.fmask0 & 15u) | ((v.fmask1 & 15u) << 4);
    }
    __syncthreads();
    // exclusive sums of the chain counts in sorted order: two positions per thread (2r, 2r + 1)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n0 = (uint32_t)__popc(s_fm[2 * threadIdx.x]), n1 = (uint32_t)__popc(s_fm[2 * threadIdx.x + 1]);
    uint32_t s = n0 + n1;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)s, o);
        if (lane >= (uint32_t)o) s += t;
    }
    if (lane == 63) wave_tot[wave] = s;
    __syncthreads();
    uint32_t before = s - (n0 + n1);
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    s_off[2 * threadIdx.x] = before;
    s_off[2 * threadIdx.x + 1] = before + n0;
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < EB_BATCH_SLOTS; r += blockDim.x) {
        const uint32_t fm = s_fm[r];
        if (!fm) continue; // (every kept survivor has at least one frequent letter; empty slots have none)
        uint32_t o = cbase + s_off[r];
        choff[base + r] = o;
        const uint64_t i = (uint64_t)(base + r) << 3;
        for (uint32_t sidx = 0; sidx < 8; ++sidx)
            if ((fm >> sidx) & 1u) chains[o++] = i | sidx;
    }
}

// ---- BWT order of the survivors.  They leave the scan in queue order; their begins are spread evenly over the range
// (~1 per 10^4 positions), so instead of a general sort (rocprim falls back to a 10-pass merge sort below 1 M keys:
// 0.2 ms) they are dealt into buckets of 2^EB_BUCKET_SHIFT positions -- count, exclusive scan, place -- and the few
// buckets that receive more than one are put in order by one thread each.  Dropped long-run candidates (begin = ~0)
// are left out here.
#ifndef EB_BUCKET_SHIFT
#define EB_BUCKET_SHIFT 11
#endif
__device__ __forceinline__ uint32_t bucket_of(uint64_t b, uint64_t base) { return b <= base ? 0u : (uint32_t)((b - base) >> EB_BUCKET_SHIFT); }

__global__ void __launch_bounds__(256) k_bucket_place(const Survivor* __restrict__ s, uint64_t n, uint64_t base, const uint32_t* __restrict__ off,
                                                     uint32_t* __restrict__ fill, Survivor* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Survivor v = s[i];
    if (v.b == ~0ull) return;
    uint32_t k = bucket_of(v.b, base);
    out[off[k] + atomicAdd(&fill[k], 1u)] = v;
}
// one thread per bucket puts the buckets with more than one survivor in order: insertion sort for the usual handful,
// an in-place heapsort when a bucket is crowded (a repeat-rich stretch of the BWT, or the adversarial test patterns), so
// that no input turns this into a quadratic crawl
__global__ void __launch_bounds__(256) k_bucket_fix(const uint32_t* __restrict__ off, uint64_t n_buckets, Survivor* __restrict__ out) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    const uint32_t lo = off[k], hi = off[k + 1];
    const uint32_t m = hi - lo;
    if (m <= 24) {
        for (uint32_t i = lo + 1; i < hi; ++i) {
            Survivor v = out[i];
            uint32_t j = i;
            while (j > lo && out[j - 1].b > v.b) { out[j] = out[j - 1]; --j; }
            if (j != i) out[j] = v;
        }
        return;
    }
    Survivor* a = out + lo;
    auto sift = [&](uint32_t root, uint32_t end) { // max-heap on .b over a[0, end)
        Survivor v = a[root];
        for (;;) {
            uint32_t child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && a[child + 1].b > a[child].b) ++child;
            if (a[child].b <= v.b) break;
            a[root] = a[child];
            root = child;
        }
        a[root] = v;
    };
    for (uint32_t i = m / 2; i-- > 0;) sift(i, m);
    for (uint32_t end = m - 1; end > 0; --end) {
        Survivor t = a[0];
        a[0] = a[end];
        a[end] = t;
        sift(0, end);
    }
}
// chains per surviving cluster (slots beyond the survivors stay 0)
__global__ void __launch_bounds__(256) k_surv_chains(const Survivor* __restrict__ sorted, const uint32_t* __restrict__ ns_ptr, uint64_t n_raw,
                                                    uint32_t* __restrict__ n_chains) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_raw) return;
    uint32_t c = 0;
    if (i < *ns_ptr) { Survivor v = sorted[i]; c = (uint32_t)(__popc(v.fmask0) + __popc(v.fmask1)); }
    n_chains[i] = c;
}
// the unordered hand-over's counterpart of k_batch_offsets (one workgroup): survivors kept and their chains -> JobLayout, report
__global__ void __launch_bounds__(256) k_unordered_layout(const uint32_t* __restrict__ ns_ptr, const uint32_t* __restrict__ choff, const uint32_t* __restrict__ nch,
                                                         ScanCounters* __restrict__ cnt, uint32_t lw, uint32_t rw, JobReport* __restrict__ report) {
    if (threadIdx.x == 0) {
        const uint32_t ns = *ns_ptr;
        const uint32_t n_chains = ns ? choff[ns - 1] + nch[ns - 1] : 0u;
        JobLayout L;
        eb_job_layout(L, ns, n_chains, lw, rw);
        cnt->lay = L;
        report->lay = L;
        report->nsurv_raw = cnt->nsurv;
        report->runs_closed = cnt->runs_closed;
        report->clusters = cnt->clusters;
        report->bases = cnt->bases;
        report->dropped = cnt->dropped;
    }
    for (int k = threadIdx.x; k < 201; k += blockDim.x) report->hist[k] = cnt->hist[k];
}
// dense work list for the left-context extension: one entry (cluster << 3 | side*4 + letter) per frequent symbol; the first
// workgroup also writes the image's header
__global__ void __launch_bounds__(256) k_chain_list(const Survivor* __restrict__ s, const uint32_t* __restrict__ offset, uint64_t n, uint64_t* __restrict__ chains,
                                                   const ScanCounters* __restrict__ cnt, ScanParams p, uint8_t* __restrict__ blob) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        const JobLayout L = cnt->lay;
        uint64_t* h = reinterpret_cast<uint64_t*>(blob);
        h[0] = EB_CALLS_MAGIC; h[1] = L.ns; h[2] = L.n_chains; h[3] = (uint64_t)(uint32_t)p.k_left; h[4] = (uint64_t)(uint32_t)p.k_right; h[5] = h[6] = h[7] = 0;
        if (L.n_chains & 1u) *reinterpret_cast<uint32_t*>(blob + L.off_crec + (uint64_t)L.n_chains * sizeof(eb_chain)) = 0u;
    }
    if (i >= n) return;
    Survivor v = s[i];
    uint32_t o = offset[i];
    for (uint32_t sidx = 0; sidx < 8; ++sidx) {
        uint32_t fm = (sidx >> 2) ? v.fmask1 : v.fmask0;
        if ((fm >> (sidx & 3u)) & 1u) chains[o++] = (i << 3) | sidx;
    }
}

// Two-BWT mode: scan planes of the MERGED sequence (row i of the merge comes from bwt2 if DA[i] else bwt1, in order:
// ebwt2InDel.cpp:1431-1432), so that the streaming scan reads symbols in the coordinates its flags live in, exactly
// as with a merged BWT + DA.  Built once per flags object.  One lane per 64 merged rows.
__global__ void __launch_bounds__(256) k_merge_planes(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint4* __restrict__ mxy) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < f.words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t d = f.da[w];
        uint64_t i1 = eb_da_rank(f, w << 6), i0 = (w << 6) - i1;
        uint64_t x = 0, y = 0;
        const uint32_t rows = (uint32_t)((f.n - (w << 6)) < 64 ? (f.n - (w << 6)) : 64);
        for (uint32_t k = 0; k < rows; ++k) {
            uint32_t code = ((d >> k) & 1ull) ? eb_code(x2, i1++) : eb_code(x1, i0++);
            x |= (uint64_t)(code & 1u) << k;          // TERM (code 4) has x = y = 0 and lands in 'A' (include.hpp:275-289)
            y |= (uint64_t)((code >> 1) & 1u) << k;
        }
        mxy[w] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
    }
}

}  // namespace

// ------------------------------------------------------------------ launchers (eb_scan_internal.h)

void eb_launch_merge_planes(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, uint4* mxy) {
    hipLaunchKernelGGL(k_merge_planes, dim3(4096), dim3(256), 0, st, x1, x2, f, mxy);
}

void eb_launch_scan(hipStream_t st, unsigned grid, unsigned dyn_lds, bool prefetch, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f,
                    const ScanParams& p, uint64_t w_lo, uint64_t w_count, Survivor* surv, uint64_t surv_cap, ScanCounters* cnt, unsigned long long* partial,
                    Survivor* slots, uint32_t* bcount, hipEvent_t scan_end) {
    auto kern = prefetch ? (p.mode == 1 ? k_scan<1, true> : (p.mode == 2 ? k_scan<2, true> : k_scan<3, true>))
                         : (p.mode == 1 ? k_scan<1, false> : (p.mode == 2 ? k_scan<2, false> : k_scan<3, false>));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn_lds, st, x1, x2, f, p, w_lo, w_count, surv, surv_cap, cnt, partial, slots, bcount);
    hipLaunchKernelGGL(k_scan_reduce, dim3(EB_SCAN_PARTIAL), dim3(256), 0, st, partial, grid, cnt, p.ordered ? 1 : 0);
    if (scan_end) (void)hipEventRecord(scan_end, st);
}

void eb_launch_order(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, Survivor* slots,
                     const uint32_t* bcount, uint64_t n_batches, uint32_t* kept, ScanCounters* cnt, uint32_t lw, uint32_t rw, uint8_t* blob, JobReport* report,
                     Survivor* sorted, uint32_t* choff, uint64_t* chains) {
    uint32_t *bch = kept + (n_batches + 1), *boff = bch + (n_batches + 1), *bchoff = boff + (n_batches + 1);
    hipLaunchKernelGGL(k_batch_resolve, dim3(eb_grid_of(n_batches * 64)), dim3(256), 0, st, x1, x2, f, p, slots, bcount, n_batches, kept, bch, cnt);
    hipLaunchKernelGGL(k_batch_offsets, dim3(1), dim3(256), 0, st, kept, bch, n_batches, boff, bchoff, cnt, p, lw, rw, blob, report);
    hipLaunchKernelGGL(k_batch_compact, dim3((unsigned)n_batches), dim3(256), 0, st, slots, bcount, boff, bchoff, cnt, sorted, choff, chains);
}

size_t eb_order_tmp_bytes(uint64_t n_buckets, uint64_t ns_raw) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(n_buckets + 1), (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)ns_raw, (hipStream_t)0);
    return a > b ? a : b;
}

// BWT order of an unordered survivor list: deal the survivors into position buckets (the count without the dropped candidates
// comes back with the chain count)
int eb_launch_order_unordered(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, Survivor* surv,
                              uint64_t ns_raw, uint64_t bucket_base, uint64_t n_buckets, uint32_t* bcnt, uint32_t* boff, void* tmp, size_t tmp_bytes,
                              uint32_t* nch, ScanCounters* cnt, uint32_t lw, uint32_t rw, JobReport* report, Survivor* sorted, uint32_t* choff) {
    uint32_t* bfill = bcnt + (n_buckets + 1); // bcnt = [count | fill]
    if (ns_raw == 0) { // nothing survived: an empty call set
        EB_HIP(hipMemsetAsync(boff + n_buckets, 0, 4, st));
        hipLaunchKernelGGL(k_unordered_layout, dim3(1), dim3(256), 0, st, boff + n_buckets, choff, nch, cnt, lw, rw, report);
        EB_HIP(hipGetLastError());
        return EB_OK;
    }
    EB_HIP(hipMemsetAsync(bcnt, 0, (n_buckets + 1) * 8, st));
    // candidates whose run is longer than the scan's two words get their histogram by rank differences (the ones that fail:
    // begin = ~0); every survivor is counted into its bucket
    hipLaunchKernelGGL(k_long_runs, dim3(eb_grid_of(ns_raw)), dim3(256), 0, st, x1, x2, f, p, surv, ns_raw, &cnt->dropped, bucket_base, bcnt);
    EB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, bcnt, boff, (int)(n_buckets + 1), st)); // boff[n_buckets] = survivors kept
    hipLaunchKernelGGL(k_bucket_place, dim3(eb_grid_of(ns_raw)), dim3(256), 0, st, surv, ns_raw, bucket_base, boff, bfill, sorted);
    hipLaunchKernelGGL(k_bucket_fix, dim3(eb_grid_of(n_buckets)), dim3(256), 0, st, boff, n_buckets, sorted);
    hipLaunchKernelGGL(k_surv_chains, dim3(eb_grid_of(ns_raw)), dim3(256), 0, st, sorted, boff + n_buckets, ns_raw, nch);
    EB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, nch, choff, (int)ns_raw, st));
    hipLaunchKernelGGL(k_unordered_layout, dim3(1), dim3(256), 0, st, boff + n_buckets, choff, nch, cnt, lw, rw, report);
    EB_HIP(hipGetLastError());
    return EB_OK;
}

void eb_launch_chain_list(hipStream_t st, const Survivor* sorted, const uint32_t* choff, uint64_t ns, uint64_t* chains, const ScanCounters* cnt, const ScanParams& p,
                          uint8_t* blob) {
    hipLaunchKernelGGL(k_chain_list, dim3(eb_grid_of(ns ? ns : 1)), dim3(256), 0, st, sorted, choff, ns, chains, cnt, p, blob);
}

// loads this file's code object now (the runtime does it with the first launch otherwise: milliseconds in front of the first job)
void eb_scan_kernels_preload() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_scan_reduce));
}
