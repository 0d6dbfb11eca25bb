// eb_scan.hip -- phase 4 on the GPU: cluster scan, frequent-symbol filter, backward left-context
// extension and right-context extraction.
//
// Replaces the loops of run_one_dataset (ebwt2InDel.cpp:1609-1655), run_two_datasets (:1395-1445),
// run_two_datasets_da (:1510-1560) and find_variants (:941-1005, :840-934, :1013-1096) with
// extract_consensus/consensus_letter (:243-319) and extract_dna (:325-342).
//
//   k_scan     one lane per 64 positions: flag = thrK & ~minima; every 1->0 transition closes a
//              cluster [b,e).  Clusters of length >= 2*mcov_out get their symbol histogram
//              (rank difference over the packed BWT; TERM counts as 'A', include.hpp:275-289) and
//              the frequent-symbol test; the few survivors are appended to a list.
//   (sort)     survivors ordered by cluster begin (BWT order) with a device radix sort.
//   k_extend   four lanes per (cluster, frequent symbol) chain: k_left-1 dependent LF(range)
//              steps, each two quad-cooperative rank queries; arg-max with ties A<C<G<T.
//   k_right    one lane per cluster: first position with thrR, then k_right x (F, FL).
// Host text formatting (distance / event_type / to_file) lives in eb_writer.cpp.
#include <hipcub/hipcub.hpp>

#include <stdlib.h>

#include <chrono>

#include "eb_internal.h"

namespace {

struct ScanParams {
    int mode;           // 1 one sample, 2 two BWTs, 3 merged BWT + DA
    uint32_t mcov;      // mcov_out
    uint32_t min_len;   // 2 * mcov_out
    int q;              // max_variants_per_position (0 = unlimited)
    int k_left, k_right;
    uint64_t begin, end; // closing positions owned by this call
    uint64_t n;          // merged length
    // ordered != 0: every workgroup scans ONE contiguous range of tiles_per_wg tiles and hands its survivors over in batches
    // of EB_BATCH_TILES tiles, batch b of workgroup g into slots [ (g * batches_per_wg + b) * EB_BATCH_SLOTS, ... ): the batches
    // are in BWT order as they lie, only their few dozen members need sorting (k_batch_compact).  ordered == 0: the
    // workgroups stride over the tiles and append to one list in arrival order (position buckets put it in order).
    uint32_t ordered, tiles_per_wg;
#ifdef EB_SCAN_EXPERIMENTS
    uint32_t dbg;        // EB_SCAN_DEBUG bits: 1 no length histogram, 2 no cluster analysis, 4 no run loop, 8 no queues, 16 no statistics
#endif
};
// Work-skipping switches for A/B timing of k_scan's parts exist only in the profiling build (make EXPERIMENTS=1, its own
// output directory); in the shipped library EB_DBG() is the constant 0, the branches vanish and no environment variable
// can hollow out the kernel.
#ifdef EB_SCAN_EXPERIMENTS
#define EB_DBG(p) ((p).dbg)
#else
#define EB_DBG(p) 0u
#endif

struct alignas(16) Survivor {
    uint64_t b, e;            // merged coordinates
    uint64_t b0, e0, b1, e1;  // per-BWT ranges (mode 2)
    uint32_t fmask0, fmask1;  // frequent letters per side, bit c = "ACGT"[c]
    uint64_t pad;
};

// the neighbour lane's value (lane i gets lane i-1's, lane 0 gets 0) as a DPP wave shift on a v_mov (v_mov_b32_dpp wave_shr:1)
// instead of __shfl_up, which compiles to ds_bpermute_b32, an LDS round trip: six of them per tile in k_scan (round 4: -1 % on the
// scan and on the step, profiles/r04_dpp_shift_ab.txt)
__device__ __forceinline__ uint64_t eb_wave_shr1_64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
#define EB_NEEDS_HIST 0x80000000u

#define EB_SCAN_PARTIAL 205 /* runs, clusters, bases, hist[201], survivors (ordered scan) */

#define EB_REQ_SLOTS 64
#define EB_BATCH_TILES 128u /* tiles per batch (128 x 256 words x 64 = 2 097 152 positions: ~210 survivors on read data) */
#define EB_BATCH_SLOTS 512u /* survivors a batch can hold; more than that: spilled, the scan runs again unordered */
#define EB_STAGE_SLOTS 128u /* the unordered scan's staging buffer in LDS */
struct ScanCounters {
    unsigned long long nsurv, runs_closed, clusters, bases, n_long, dropped, spilled;
    unsigned long long hist[201];
    unsigned long long req_ext[EB_REQ_SLOTS], req_right[EB_REQ_SLOTS]; // memory requests counted by k_extend / k_right (count_requests)
    unsigned long long req_sum[2];                                     // their sums (k_req_sum), copied to the job's pinned words
};

__device__ __forceinline__ uint64_t flag_word(const eb_dev_flags& f, uint64_t w) { return f.flag[w]; }

// ones in DA[0,p)
__device__ __forceinline__ uint64_t da_rank(const eb_dev_flags& f, uint64_t p) {
    uint64_t g = p >> 9, w = p >> 6;
    uint64_t s = f.da_rank[g];
    for (uint64_t k = g << 3; k < w; ++k) s += (uint64_t)__popcll(f.da[k]);
    uint32_t o = (uint32_t)(p & 63u);
    if (o) s += (uint64_t)__popcll(f.da[w] & ((1ull << o) - 1ull));
    return s;
}

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
//   hot queue : clusters that contain a change of symbol (found bit-parallel, ~1 in 5)   -> two-popcount necessary test
//   todo queue: clusters that pass it (~1 in 30)                                         -> histogram, frequent letters, record
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
                    s.b1 = da_rank(f, s.b); s.e1 = da_rank(f, s.e);
                    s.b0 = s.b - s.b1; s.e0 = s.e - s.e1;
                }
            }
        }
        if (keep && ordered) {
            // straight into the slots of the batch at hand: one cursor per batch (thousands of them: no contention).  A wave
            // empties its queues before it enters the next batch, so nothing of this batch is staged later.
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
        if (ordered) cur_batch = (uint64_t)blockIdx.x * ((T + EB_BATCH_TILES - 1) / EB_BATCH_TILES) + it / EB_BATCH_TILES;
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
        // ~19 of 20 clusters (one symbol, or a read start or two) end here.  Rows of a run that lie in the previous word
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
            if (((it + 1) % EB_BATCH_TILES) == 0 || it + 1 == iters) drain();
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
            s.b1 = da_rank(f, s.b); s.e1 = da_rank(f, s.e);
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

// ---- ordered scan (ScanParams.ordered): the batches lie in BWT order in `slots`, batch b holding bcount[b] survivors in
// arrival order.  One wave per batch.
// (1) long-run candidates get their histogram; every batch reports how many survivors it keeps (kept[n_batches] = 0 for the
//     exclusive scan that follows, whose last element is then the total)
__global__ void __launch_bounds__(256) k_batch_resolve(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p, Survivor* __restrict__ slots,
                                                      const uint32_t* __restrict__ bcount, uint64_t n_batches, uint32_t* __restrict__ kept,
                                                      unsigned long long* __restrict__ dropped) {
    const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (b >= n_batches) return; // whole waves
    const uint32_t c = bcount[b];
    uint32_t mine = 0;
    for (uint32_t i = lane; i < c; i += 64u) {
        Survivor* q = slots + b * EB_BATCH_SLOTS + i;
        if (q->fmask0 == EB_NEEDS_HIST) {
            Survivor s = *q;
            bool keep = resolve_long_run(x1, x2, f, p, s);
            if (!keep) atomicAdd(dropped, 1ull);
            *q = s;
            mine += keep ? 1u : 0u;
        } else
            ++mine;
    }
    for (int o = 32; o > 0; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o);
    if (lane == 0) {
        kept[b] = mine;
        if (b == 0) kept[n_batches] = 0;
    }
}
// (2) each batch puts its members in order (rank by counting: a few dozen keys in LDS) and writes them to their place in the
//     dense list; the chain count of every survivor goes out with it (k_surv_chains of the unordered path).  n_chains[]
//     beyond the survivors kept has been zeroed by the caller.
__global__ void __launch_bounds__(256) k_batch_compact(const Survivor* __restrict__ slots, const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ off,
                                                      uint64_t n_batches, Survivor* __restrict__ sorted, uint32_t* __restrict__ n_chains) {
    __shared__ unsigned long long keys[EB_BATCH_SLOTS];
    const uint64_t b = blockIdx.x; // one workgroup per batch: the rank loop is c compares per member
    const uint32_t c = bcount[b];
    const uint32_t base = off[b];
    const Survivor* q = slots + b * EB_BATCH_SLOTS;
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) keys[i] = q[i].b;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        const unsigned long long key = keys[i];
        if (key == ~0ull) continue; // a dropped long-run candidate
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; ++j) rank += keys[j] < key ? 1u : 0u; // begins of distinct clusters are distinct
        Survivor v = q[i];
        sorted[base + rank] = v;
        n_chains[base + rank] = (uint32_t)(__popc(v.fmask0) + __popc(v.fmask1));
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
// surviving clusters and their chains, for the host (one read-back for both)
__global__ void k_prep_totals(const uint32_t* __restrict__ ns_ptr, const uint32_t* __restrict__ choff, const uint32_t* __restrict__ nch,
                              uint32_t* __restrict__ out2) {
    uint32_t ns = *ns_ptr;
    out2[0] = ns;
    out2[1] = ns ? choff[ns - 1] + nch[ns - 1] : 0u;
}
// dense work list for k_extend: one entry (cluster << 3 | side*4 + letter) per frequent symbol
__global__ void __launch_bounds__(256) k_chain_list(const Survivor* __restrict__ s, const uint32_t* __restrict__ offset, uint64_t n, uint64_t* __restrict__ chains) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Survivor v = s[i];
    uint32_t o = offset[i];
    for (uint32_t sidx = 0; sidx < 8; ++sidx) {
        uint32_t fm = (sidx >> 2) ? v.fmask1 : v.fmask0;
        if ((fm >> (sidx & 3u)) & 1u) chains[o++] = (i << 3) | sidx;
    }
}

__device__ __forceinline__ uint64_t pick4(const uint64_t* F, uint32_t c) { return c == 0 ? F[0] : c == 1 ? F[1] : c == 2 ? F[2] : F[3]; }

// Memory requests issued by the context kernels, counted where they are issued (eb_scan_stats.requests_*): every lane
// of the wave calls this once at the end; one non-returning atomic per wave, spread over EB_REQ_SLOTS words.
__device__ __forceinline__ void count_requests(unsigned long long* __restrict__ slots, uint32_t mine) {
    uint32_t v = mine;
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&slots[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (EB_REQ_SLOTS - 1)], (unsigned long long)v);
}

// Four lanes per chain.  extract_consensus(bwt,range,c,...) ebwt2InDel.cpp:303-319 -> LF(range,c), then
// up to k_left-1 x { LF(range) ; consensus_letter } (:243-282).  Output: eb_chain + bases packed 2 bits each in
// extension order (the writer reverses them, :315).
template <bool TWO> // TWO: two-BWT mode (the only one in which a chain may run in the second index)
__global__ void __launch_bounds__(256) k_extend(eb_dev_index x1, eb_dev_index x2, ScanParams p, const Survivor* __restrict__ surv,
                                               const uint64_t* __restrict__ chains, uint64_t n_chains, eb_chain* __restrict__ crec,
                                               uint64_t* __restrict__ left, uint32_t left_words, unsigned long long* __restrict__ req_slots) {
    const uint32_t j = threadIdx.x & 3u;
    uint64_t ch = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    uint32_t n_req = 0; // 64-byte block requests issued by this quad (counted by its lane 0)
    if (ch < n_chains) { // uniform inside a quad
    uint64_t item = chains[ch];
    uint64_t i = item >> 3;
    uint32_t sidx = (uint32_t)(item & 7u);
    Survivor s = surv[i];
    uint32_t side = sidx >> 2, c = sidx & 3u;
    // the index of this chain: a by-value pick of the few fields used (a reference chosen at run time between two kernel
    // arguments would put both structs into scratch memory, on the path of every dependent step)
    eb_dev_index x = x1;
    if (TWO && side == 1) { x.chunks = x2.chunks; x.super = x2.super; x.n = x2.n; x.n_blocks = x2.n_blocks; for (int k = 0; k < 4; ++k) x.F[k] = x2.F[k]; }
    uint64_t L, R;
    if (TWO) { L = side ? s.b1 : s.b0; R = side ? s.e1 : s.e0; } else { L = s.b; R = s.e; }
    uint64_t* out = left + ch * (uint64_t)left_words;
    // Every lane keeps what belongs to its letter j: F[j], and the rank of j at both ends of the range.  The quad sums of the
    // packed counts, the arg-max and the hand-over of the winner's range go through DPP (quad permutations on the consuming
    // instruction) -- with __shfl they were fourteen ds_bpermute round trips per step, five of them one after the other.
    const uint32_t sh8 = 8u * j;
    const uint64_t Fj = pick4(x.F, j);
    eb_super_cache sc = {0xFFFFFFFFu, 0};
    // LF(range, c)
    uint64_t a, bq;
    eb_rank4_quad_pair_dpp(x, sc, L, R, j, sh8, a, bq);
    n_req += 1u + ((R >> 7) != (L >> 7));
    L = eb_quad_pick64(Fj + a, j == c);
    R = eb_quad_pick64(Fj + bq, j == c);
    uint32_t support = (uint32_t)(R - L);
    uint32_t have = 1;
    uint64_t word = c; // base 0 = the frequent symbol itself
    while (have < (uint32_t)p.k_left) {
        eb_rank4_quad_pair_dpp(x, sc, L, R, j, sh8, a, bq);
        n_req += 1u + ((R >> 7) != (L >> 7));
        const uint64_t len = bq - a;
        // arg-max over the quad, ties to the smaller letter (stable descending sort, ebwt2InDel.cpp:252-255): the maximum of
        // len << 2 | (3 - j); 32-bit keys unless some range of the wave is 2^30 rows or longer (wave-uniform choice)
        uint32_t bj;
        uint64_t bl;
        if (__ballot((len >> 30) != 0ull) == 0ull) {
            const uint32_t key = eb_quad_max(((uint32_t)len << 2) | (3u - j));
            bj = 3u - (key & 3u);
            bl = key >> 2;
        } else {
            const uint64_t key = eb_quad_max64((len << 2) | (uint64_t)(3u - j));
            bj = 3u - ((uint32_t)key & 3u);
            bl = key >> 2;
        }
        if (bl == 0) break; // consensus_letter returns 0 (:257)
        L = eb_quad_pick64(Fj + a, j == bj);
        R = L + bl;
        word |= (uint64_t)bj << (2u * (have & 31u));
        ++have;
        if ((have & 31u) == 0) { // 32 bases collected
            if (j == 0) out[(have >> 5) - 1] = word;
            word = 0;
        }
    }
    if (j == 0) {
        if (have == (uint32_t)p.k_left) {
            if (have & 31u) out[have >> 5] = word;
        } else {
            // a context shorter than k_left is dropped by the writer (ebwt2InDel.cpp:317): leave it blank
            for (uint32_t wi = 0; wi < (have >> 5); ++wi) out[wi] = 0;
        }
        eb_chain k;
        k.side = (uint8_t)side;
        k.sym = (uint8_t)"ACGT"[c];
        k.reserved = 0;
        k.len = have;
        k.support = support;
        crec[ch] = k;
    }
    }
    count_requests(req_slots, j == 0 ? n_req : 0u);
}

// The same chains, ONE lane per chain (round 4).  The quad form above spends ~113 vector instructions per LF step and wave on 16
// chains -- every lane runs the whole instruction stream on a quarter of the block -- and beside the scan the pass is short of
// vector issue as much as of memory lines.  Here a lane loads the whole 64-byte block (four 16-byte loads from one line: one
// fabric request all the same) and counts all four letters at both ends of its range: ~170 instructions per step and wave for
// 64 chains.  Same results by construction (the counts are those of eb_rank4).
template <bool TWO>
__global__ void __launch_bounds__(256) k_extend1(eb_dev_index x1, eb_dev_index x2, ScanParams p, const Survivor* __restrict__ surv,
                                                const uint64_t* __restrict__ chains, uint64_t n_chains, eb_chain* __restrict__ crec,
                                                uint64_t* __restrict__ left, uint32_t left_words, unsigned long long* __restrict__ req_slots) {
    const uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_req = 0;
    if (ch < n_chains) {
        const uint64_t item = chains[ch];
        const uint64_t i = item >> 3;
        const uint32_t sidx = (uint32_t)(item & 7u);
        const Survivor s = surv[i];
        const uint32_t side = sidx >> 2, c = sidx & 3u;
        eb_dev_index x = x1; // by value, see k_extend
        if (TWO && side == 1) { x.chunks = x2.chunks; x.super = x2.super; x.n = x2.n; x.n_blocks = x2.n_blocks; for (int k = 0; k < 4; ++k) x.F[k] = x2.F[k]; }
        uint64_t L, R;
        if (TWO) { L = side ? s.b1 : s.b0; R = side ? s.e1 : s.e0; } else { L = s.b; R = s.e; }
        uint64_t* out = left + ch * (uint64_t)left_words;
        // packed per-letter counts (A | C << 8 | G << 16 | T << 24; at most 128 each) of the symbols before in-block offset o
        auto counts = [](const uint4& c0, const uint4& c1, const uint4& c2, const uint4& c3, uint32_t o) -> uint32_t {
            return eb_chunk_counts(c0, eb_prefix_mask32(o, 0)) + eb_chunk_counts(c1, eb_prefix_mask32(o, 1)) + eb_chunk_counts(c2, eb_prefix_mask32(o, 2)) +
                   eb_chunk_counts(c3, eb_prefix_mask32(o, 3));
        };
        auto letter_w = [](const uint4& c0, const uint4& c1, const uint4& c2, const uint4& c3, uint32_t j) -> uint32_t {
            return j == 0 ? c0.w : j == 1 ? c1.w : j == 2 ? c2.w : c3.w;
        };
        uint64_t sup[4] = {0, 0, 0, 0};
        uint32_t sup_at = 0xFFFFFFFFu; // superblock whose counts `sup` holds
        // one LF(range) (dna_bwt.hpp:138-166): a[j], b[j] = rank of letter j at L and R, without F
        auto ranks = [&](uint64_t lo, uint64_t hi, uint64_t* a, uint64_t* b) {
            const uint4* bl = x.chunks + ((lo >> 7) << 2);
            const uint4 c0 = eb_load_chunk(bl), c1 = eb_load_chunk(bl + 1), c2 = eb_load_chunk(bl + 2), c3 = eb_load_chunk(bl + 3);
            uint4 d0 = c0, d1 = c1, d2 = c2, d3 = c3;
            const bool two = (hi >> 7) != (lo >> 7);
            if (two) {
                const uint4* br = x.chunks + ((hi >> 7) << 2);
                d0 = eb_load_chunk(br); d1 = eb_load_chunk(br + 1); d2 = eb_load_chunk(br + 2); d3 = eb_load_chunk(br + 3);
            }
            n_req += 1u + (two ? 1u : 0u);
            const uint32_t pl = counts(c0, c1, c2, c3, (uint32_t)(lo & 127u)), pr = counts(d0, d1, d2, d3, (uint32_t)(hi & 127u));
            // superblock counts: kept from the previous step (a walk nearly always stays inside one superblock of 2^32 symbols;
            // eight table loads per step otherwise, on the path of every dependent step)
            const uint32_t sbl = (uint32_t)(lo >> EB_SUPER_SHIFT), sbr = (uint32_t)(hi >> EB_SUPER_SHIFT);
            if (sbl != sup_at) {
                sup_at = sbl;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) sup[j] = x.super[((uint64_t)sbl << 2) + j];
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint64_t sr = sbr == sbl ? sup[j] : x.super[((uint64_t)sbr << 2) + j];
                a[j] = sup[j] + letter_w(c0, c1, c2, c3, j) + ((pl >> (8u * j)) & 0xFFu);
                b[j] = sr + letter_w(d0, d1, d2, d3, j) + ((pr >> (8u * j)) & 0xFFu);
            }
        };
        uint64_t a[4], b[4];
        ranks(L, R, a, b);
        // LF(range, c)
        L = pick4(x.F, c) + (c == 0 ? a[0] : c == 1 ? a[1] : c == 2 ? a[2] : a[3]);
        R = pick4(x.F, c) + (c == 0 ? b[0] : c == 1 ? b[1] : c == 2 ? b[2] : b[3]);
        const uint32_t support = (uint32_t)(R - L);
        uint32_t have = 1;
        uint64_t word = c; // base 0 = the frequent symbol itself
        while (have < (uint32_t)p.k_left) {
            ranks(L, R, a, b);
            // largest sub-range, ties to the smaller letter (stable descending sort, ebwt2InDel.cpp:252-255)
            uint64_t bl = b[0] - a[0];
            uint32_t bj = 0;
#pragma unroll
            for (uint32_t j = 1; j < 4; ++j) {
                const uint64_t len = b[j] - a[j];
                if (len > bl) { bl = len; bj = j; }
            }
            if (bl == 0) break; // consensus_letter returns 0 (:257)
            L = pick4(x.F, bj) + (bj == 0 ? a[0] : bj == 1 ? a[1] : bj == 2 ? a[2] : a[3]);
            R = L + bl;
            word |= (uint64_t)bj << (2u * (have & 31u));
            ++have;
            if ((have & 31u) == 0) { out[(have >> 5) - 1] = word; word = 0; }
        }
        if (have == (uint32_t)p.k_left) {
            if (have & 31u) out[have >> 5] = word;
        } else {
            for (uint32_t wi = 0; wi < (have >> 5); ++wi) out[wi] = 0; // a context shorter than k_left is dropped by the writer (:317): blank
        }
        eb_chain k;
        k.side = (uint8_t)side;
        k.sym = (uint8_t)"ACGT"[c];
        k.reserved = 0;
        k.len = have;
        k.support = support;
        crec[ch] = k;
    }
    count_requests(req_slots, n_req);
}

// One lane per cluster: record header + right context (extract_dna, ebwt2InDel.cpp:325-342) at the first
// cluster position whose LCP >= k_right (:979-988; two BWTs :896-912); bases packed 2 bits each in text order.
// Launched behind the left-context extension, straight into the call set's image (ns_ptr: the number of survivors, in device
// memory; gate_chains: the extension's records when the clusters that cannot print are skipped).
template <bool TWO, int LANES> // LANES = 1: one lane per cluster; 4: a quad per cluster (eb_select_quad)
__global__ void __launch_bounds__(256) k_right(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p,
                                              const Survivor* __restrict__ surv, const uint32_t* __restrict__ ns_ptr,
                                              const uint32_t* __restrict__ chain_off, uint32_t chain_base,
                                              eb_cluster* __restrict__ recs, uint64_t* __restrict__ right, uint32_t right_words,
                                              unsigned long long* __restrict__ req_slots, const eb_chain* __restrict__ gate_chains) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint32_t j = LANES == 4 ? threadIdx.x & 3u : 0u, qbase = (threadIdx.x & 63u) & ~3u;
    uint32_t n_req = 0; // directory reads + 64-byte blocks probed by this cluster's FL steps (counted by its first lane)
    if (i < *ns_ptr) {
    Survivor s = surv[i];
    // Gate (large call sets, launched behind k_extend): the writer touches a cluster's right context only when it prints,
    // and it cannot print unless two contexts reached their full length (one sample: v.size() < 2 returns before anything
    // else, ebwt2InDel.cpp:1256; two samples: a pair needs a full context on each side whose last characters differ,
    // :1161-1163 / :921,1083 -- shorter contexts were dropped at :317).  About one surviving cluster in five sits next to
    // a read start and fails this; its 30 FL steps (~75 memory requests) are skipped and the record says so.
    bool walk = true;
    if (gate_chains) {
        const uint32_t n_slots = (uint32_t)(__popc(s.fmask0) + __popc(s.fmask1));
        const eb_chain* k = gate_chains + chain_off[i];
        uint32_t full0 = 0, full1 = 0; // letters (bit = A,C,G,T) with a full-length context, per side
        for (uint32_t t = 0; t < n_slots; ++t) {
            eb_chain c = k[t];
            if (c.len != (uint32_t)p.k_left) continue;
            uint32_t bit = c.sym == 'A' ? 1u : c.sym == 'C' ? 2u : c.sym == 'G' ? 4u : 8u;
            if (c.side) full1 |= bit; else full0 |= bit;
        }
        if (p.mode == 1) walk = __popc(full0) >= 2;
        else walk = full0 && full1 && (__popc(full0 | full1) >= 2); // some a in side 0, b in side 1 with different letters
    }
    // first position in [b,e) with thrR
    uint64_t pos = ~0ull;
    for (uint64_t w = s.b >> 6; w <= (s.e - 1) >> 6; ++w) {
        uint64_t m = f.thrR[w];
        uint64_t lo = w << 6;
        if (s.b > lo) m &= ~0ull << (s.b - lo);
        if (s.e < lo + 64) m &= ~0ull >> (lo + 64 - s.e);
        if (m) { pos = lo + (uint32_t)__ffsll((long long)m) - 1u; break; }
    }
    uint32_t len = 0;
    if (pos != ~0ull && walk) {
        uint64_t row = pos;
        eb_dev_index x = x1; // by value, see k_extend
        if (TWO) {
            const bool second = eb_bit(f.da, pos);
            uint64_t i1 = da_rank(f, pos);
            row = second ? i1 : pos - i1;
            // every field the walk reads, picked one by one (a conditional copy of the arrays left the whole struct in scratch
            // memory in the one-lane form: 176 bytes per lane and a dynamic index on the path of every step)
            x.chunks = second ? x2.chunks : x1.chunks; x.super = second ? x2.super : x1.super; x.sel_dir = second ? x2.sel_dir : x1.sel_dir;
            x.n = second ? x2.n : x1.n; x.n_blocks = second ? x2.n_blocks : x1.n_blocks; // (the select's guess is clamped to the index it walks in)
            x.sel_shift = second ? x2.sel_shift : x1.sel_shift; x.sel_blk_bits = second ? x2.sel_blk_bits : x1.sel_blk_bits; // (the two indexes may differ in size)
            x.psi = second ? x2.psi : x1.psi;
#pragma unroll
            for (int k = 0; k < 4; ++k) { x.F[k] = second ? x2.F[k] : x1.F[k]; x.sel_off[k] = second ? x2.sel_off[k] : x1.sel_off[k]; x.psi_off[k] = second ? x2.psi_off[k] : x1.psi_off[k]; }
        }
        uint64_t* out = right + i * (uint64_t)right_words;
        uint64_t word = 0;
        int c = eb_F_letter(x, row);
        while (c >= 0 && len < (uint32_t)p.k_right) {
            word |= (uint64_t)c << (2u * (len & 31u));
            ++len;
            if ((len & 31u) == 0) { if (j == 0) out[(len >> 5) - 1] = word; word = 0; }
            if (LANES == 4) row = eb_select_quad(x, row - eb_F_of(x, c), c, j, qbase, n_req);
            else row = eb_FL(x, row, c, n_req);
            c = eb_F_letter(x, row);
        }
        if ((len & 31u) && j == 0) out[len >> 5] = word;
    }
    eb_cluster r;
    r.begin = s.b;
    r.end = s.e;
    r.first_chain = chain_off[i] + chain_base;
    r.n_slots = (uint16_t)(__popc(s.fmask0) + __popc(s.fmask1));
    r.right_found = pos != ~0ull ? 1 : 0;
    r.right_len = len;
    r.reserved = walk ? 0u : EB_RIGHT_SKIPPED;
    if (j == 0) recs[i] = r;
    }
    count_requests(req_slots, j == 0 ? n_req : 0u);
}

__global__ void k_req_sum(ScanCounters* __restrict__ c) {
    unsigned long long a = c->req_ext[threadIdx.x], b = c->req_right[threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (threadIdx.x == 0) { c->req_sum[0] = a; c->req_sum[1] = b; }
}

// Two-BWT mode: scan planes of the MERGED sequence (row i of the merge comes from bwt2 if DA[i] else bwt1, in order:
// ebwt2InDel.cpp:1431-1432), so that the streaming scan reads symbols in the coordinates its flags live in, exactly
// as with a merged BWT + DA.  Built once per flags object.  One lane per 64 merged rows.
__global__ void __launch_bounds__(256) k_merge_planes(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, uint4* __restrict__ mxy) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < f.words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t d = f.da[w];
        uint64_t i1 = da_rank(f, w << 6), i0 = (w << 6) - i1;
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

#define EB_PINNED_RING_BYTES (2ull << 30) /* most host memory a pipelining caller's ring of eight page-locked blobs may take */
#define EB_JOB_PINNED_BYTES (512 + 16 * (sizeof(ScanCounters) + 64)) /* page-locked words of one job (<= 16 sub-shards) */
struct DevMem { // per-call temporary from the caching scratch allocator
    void* p = nullptr;
    size_t cap = 0;
    int device = 0;
    ~DevMem() { if (p) eb_scratch_put(device, p, cap); }
    int alloc(size_t bytes) {
        if (p && cap >= bytes) return EB_OK;
        if (p) { eb_scratch_put(device, p, cap); p = nullptr; }
        p = eb_scratch_get(device, bytes, &cap);
        if (!p) { eb_set_error("device scratch allocation of %zu bytes failed", bytes); return EB_ERR_NOMEM; }
        return EB_OK;
    }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

unsigned grid_of(uint64_t threads) { return (unsigned)((threads + 255) / 256); }
// persistent scan grid: 8 workgroups per CU, each striding over 256-word tiles
unsigned scan_grid(uint64_t words, unsigned per_cu = 8) {
    uint64_t tiles = (words + 255) / 256;
    const uint64_t cap = 256ull * per_cu;
    return (unsigned)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

}  // namespace

namespace {
enum { E_SCAN0, E_SCAN1, E_SCAN_DONE, E_PREP_CNT, E_PREP_DONE, E_EXT0, E_EXT_DONE, E_RIGHT0, E_RIGHT_DONE, E_SORTED, E_PER_SHARD };
struct SubShard {
    uint64_t begin = 0, end = 0, w_lo = 0, w_count = 0, cap = 0;
    DevMem cnt, part, surv, sorted, bcnt, boff, tmp, nch, choff, chains, recs, crec, left, right, blob; // bcnt/boff: position buckets (ordered scan: survivors kept per batch / their offsets)
    DevMem slots, bcount;            // ordered scan: the batches' survivors and counts
    bool ordered = false;            // this sub-shard's scan ran in ordered mode
    uint64_t n_batches = 0;
    eb_cluster* d_recs = nullptr; // sections: inside `blob` (one shard) or the separate buffers above (sub-shards)
    eb_chain* d_crec = nullptr;
    uint64_t *d_left = nullptr, *d_right = nullptr;
    uint64_t ns_raw = 0, ns = 0, n_chains = 0, rec_base = 0, chain_base = 0;
    uint64_t n_buckets = 0;          // position buckets of the ordering step
    const uint32_t* d_ns = nullptr;  // survivors kept, in device memory (k_right's bound)
    hipEvent_t e[E_PER_SHARD]; // scan_t0, scan_t1, scan_done, prep_cnt, prep_done, ext_t0, ext_done, right_t0, right_done, sorted
    ScanCounters* h_cnt = nullptr;
    uint32_t* h_last = nullptr;
};
}  // namespace

// A phase-4 call in flight: everything is enqueued, the record copies to host may still be running.
struct eb_scan_job {
    eb_index* b1 = nullptr;
    std::vector<SubShard> sh;
    std::vector<hipEvent_t> events; // owned until the job ends, then returned to the context's pool
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    eb_calls* calls = nullptr;
    ScanCounters c;
    uint64_t begin = 0, end = 0, ns = 0, n_chains = 0;
    int S = 1;
    uint32_t options = 0;
    bool has_blob = false;      // sh[0].blob holds the contiguous device image
    uint64_t blob_bytes = 0;
    eb_calls_header header;     // first 64 bytes of the device image
    DevMem wire;                // compact wire image (packed by the job with EB_SCAN_WIRE, else on demand)
    unsigned long long* h_bad = nullptr; // pinned: records that do not fit the wire image
    uint64_t wire_bytes = 0;
    int32_t k_left = 0, k_right = 0;
    bool wire_full_width = false; // the tight wire image overflowed: full-width records were packed instead
    // what launch_contexts (part two of _begin) needs
    eb_dev_index x1, x2;
    eb_dev_flags df;
    ScanParams p;
    uint32_t lw = 0, rw = 0;
    // EB_SCAN_PIPELINED: the job's last part (finish_begin) is held back until the next _begin on the index, or until the
    // job is waited for / ended; rc_tail is what that part returned
    hipStream_t s_ext = nullptr, s_right = nullptr; // this job's pair of context streams
    hipStream_t s_order = nullptr;                  // stream of the ordering step and the image layout (the preparation stream, or the scan's in overlap mode)
    uint32_t reruns = 0;            // scans run a second time (ordered hand-over spilled / survivor list too short)
    bool half = false;
    int rc_tail = EB_OK;
    char err_tail[256] = "";
};

// friendly records -> compact wire records (eb_scan_job_device_wire); *bad counts fields that do not fit
__global__ void __launch_bounds__(256) k_pack_wire(const eb_cluster* __restrict__ recs, uint64_t count, const eb_chain* __restrict__ chains, uint64_t n_chains,
                                                  eb_wire_cluster* __restrict__ wc, eb_wire_chain* __restrict__ wk, unsigned long long* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        eb_cluster r = recs[i];
        eb_wire_cluster w;
        w.begin = r.begin;
        w.len = (uint32_t)(r.end - r.begin);
        w.n_slots = (uint8_t)r.n_slots;
        w.right_found = (uint8_t)((r.right_found ? 1u : 0u) | ((r.reserved & EB_RIGHT_SKIPPED) ? 2u : 0u));
        w.right_len = (uint16_t)r.right_len;
        if (r.end - r.begin > 0xFFFFFFFFull || r.right_len > 0xFFFFu || r.n_slots > 0xFFu) atomicAdd(bad, 1ull);
        wc[i] = w;
    }
    if (i < n_chains) {
        eb_chain k = chains[i];
        eb_wire_chain w;
        w.side = k.side;
        w.sym = k.sym;
        w.len = (uint16_t)k.len;
        w.support = k.support;
        if (k.len > 0xFFFFu) atomicAdd(bad, 1ull);
        wk[i] = w;
    }
}

__global__ void __launch_bounds__(256) k_pack_wire4(const eb_cluster* __restrict__ recs, uint64_t count, const eb_chain* __restrict__ chains, uint64_t n_chains,
                                                   eb_wire4_cluster* __restrict__ wc, eb_wire4_chain* __restrict__ wk, unsigned long long* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        eb_cluster r = recs[i];
        eb_wire4_cluster w;
        w.begin_lo = (uint32_t)r.begin;
        w.begin_hi = (uint8_t)(r.begin >> 32);
        w.slots_found = (uint8_t)((r.n_slots & 15u) | ((r.right_found ? 1u : 0u) << 4) | (((r.reserved & EB_RIGHT_SKIPPED) ? 1u : 0u) << 5));
        w.len = (uint16_t)(r.end - r.begin);
        w.right_len = (uint16_t)r.right_len;
        w.pad = 0;
        if ((r.begin >> 40) || r.end - r.begin > 0xFFFFull || r.right_len > 0xFFFFu || r.n_slots > 15u) atomicAdd(bad, 1ull);
        wc[i] = w;
    }
    if (i < n_chains) {
        eb_chain k = chains[i];
        uint32_t letter = k.sym == 'A' ? 0u : k.sym == 'C' ? 1u : k.sym == 'G' ? 2u : 3u;
        eb_wire4_chain w;
        w.side_sym_len = (uint16_t)((k.side & 1u) | (letter << 1) | (k.len << 3));
        w.support = (uint16_t)k.support;
        if (k.len > 8191u || k.support > 0xFFFFu) atomicAdd(bad, 1ull);
        wk[i] = w;
    }
}

// first 64 bytes of the device image of a call set
__global__ void k_blob_header(eb_calls_header h, uint64_t* dst) {
    const uint64_t v[8] = {h.magic, h.count, h.n_chains, h.k_left, h.k_right, 0, 0, 0};
    if (threadIdx.x < 8) dst[threadIdx.x] = v[threadIdx.x];
}

void eb_scan_ctx_destroy(eb_scan_ctx* c) {
    if (!c) return;
    for (hipStream_t st : {c->prep, c->ext, c->right, c->copy})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->last_ext) (void)hipEventDestroy(c->last_ext);
    if (c->last_right) (void)hipEventDestroy(c->last_right);
    for (unsigned long long* p : c->h_bad_pool) (void)hipHostFree(p);
    delete c;
}

static int wire_enqueue(eb_scan_job* job, hipStream_t st, unsigned long long* h_bad, int version);
static void begin_trace_at(const char* what); // EB_BEGIN_TRACE (see BeginTrace)

// Part two of a phase-4 job: the context kernels, the hand-over of the call set (host blob or device image, wire image)
// and the end event.  (Round 2 tried holding this part back until the NEXT job on the index had enqueued its scan, so that
// the small kernels and the host round trip between a scan and its own context kernels would lie under the previous job's
// context kernels: 3.80 ms per step against 3.61 -- the preparation kernels and the context kernels slow each other down
// by more than the gap they hide.)
static int launch_contexts(eb_scan_job* job) {
    eb_index* b1 = job->b1;
    eb_scan_ctx* ctx = b1->scan_ctx;
    EB_HIP(hipSetDevice(b1->device));
    const hipStream_t sD = ctx->copy;
    const eb_dev_index& x1 = job->x1;
    const eb_dev_index& x2 = job->x2;
    const eb_dev_flags& df = job->df;
    const ScanParams& p = job->p;
    const uint32_t lw = job->lw, rw = job->rw;
    const int S = job->S;
    const uint32_t options = job->options;
    std::vector<SubShard>& sh = job->sh;
    // one lane per chain (k_extend1) since round 4: 0.98 -> 0.80 ms alone at 7.55 G positions, 1.42 -> 0.87 beside the scan
    // (profiles/r04_extend_ab.txt); EB_EXTEND_LANES=4: a quad per chain (k_extend), kept under test
    // (two BWTs: the quad form -- with a lane's index chosen at run time the one-lane form loses, 1.49 against 0.96 ms at 2 x 3.8 G.
    // Large indexes: the quad form too -- a lane's four 16-byte loads are four address translations where the quad's coalesced
    // request is one, and once the index outgrows what the translation caches reach that is what counts: 0.72 / 0.82 ms (one lane /
    // quad) at 7.55 G symbols, 1.92 / 1.82 at 15.1 G, 45 / 23 ms at 151 G; the superblock counts in LDS changed nothing there.)
    const int extend_env = getenv("EB_EXTEND_LANES") ? atoi(getenv("EB_EXTEND_LANES")) : 0;
    const int extend_lanes = extend_env == 4 || extend_env == 1 ? extend_env : (p.mode == 2 || x1.n > (1ull << 33) ? 4 : 1);
    auto kern_extend = extend_lanes == 1 ? (p.mode == 2 ? k_extend1<true> : k_extend1<false>) : (p.mode == 2 ? k_extend<true> : k_extend<false>);
    // the late walk of a large call set: a quad per cluster (EB_RIGHT_LANES=1: one lane, the form the early walk keeps)
    // With the FL table (eb_psi_lookup) a step is one 16-byte read by one lane: one lane per cluster, 64 walks per wave.
    static const int right_lanes_env = getenv("EB_RIGHT_LANES") ? atoi(getenv("EB_RIGHT_LANES")) : 0;
    const bool have_psi = x1.psi && (p.mode != 2 || x2.psi);
    const int right_lanes = right_lanes_env == 1 || right_lanes_env == 4 ? right_lanes_env : (have_psi ? 1 : 4);
    auto kern_right = p.mode == 2 ? (right_lanes == 4 ? k_right<true, 4> : k_right<true, 1>) : (right_lanes == 4 ? k_right<false, 4> : k_right<false, 1>);
    int rc;
    for (int k = 0; k < S; ++k) {
        SubShard& x = sh[(size_t)k];
        if (x.ns == 0) continue;
        // The walk runs BEHIND the extension, so both simply follow the ordering step on its stream -- a hand-over between streams
        // costs 15-25 us of idle device each time.  (Rounds 1-2 launched the walk of a small call set early, with the ordering step,
        // into temporaries; round 3 measured that against the late walk and retired it: docs/EXPERIMENTS.md.)
        // (Round 4, with the FL table: the walk of pass i on its own stream beside the extension of pass i+1 -- own streams for a
        // pipelining caller's large call sets too -- 2.50-2.61 against 2.20-2.31 ms per step: not kept, profiles/r04_overlap_ab.txt.)
        const bool own_streams = S > 1; // (sub-shards overlap one shard's ordering with the previous one's kernels)
        const hipStream_t sB = own_streams ? job->s_ext : ctx->prep, sC = own_streams ? job->s_right : ctx->prep;
        EB_HIP(hipStreamWaitEvent(sB, x.e[E_PREP_DONE], 0));
        EB_HIP(hipEventRecord(x.e[E_EXT0], sB));
        if (x.n_chains)
            hipLaunchKernelGGL(kern_extend, dim3(grid_of(x.n_chains * (uint64_t)extend_lanes)), dim3(256), 0, sB, x1, x2, p, x.sorted.as<Survivor>(), x.chains.as<uint64_t>(), x.n_chains,
                               x.d_crec, x.d_left, lw, x.cnt.as<ScanCounters>()->req_ext);
        EB_HIP(hipEventRecord(x.e[E_EXT_DONE], sB));
        EB_HIP(hipEventRecord(ctx->last_ext, sB));
        EB_HIP(hipStreamWaitEvent(sC, x.e[E_PREP_DONE], 0));
        {
            // A large call set keeps the memory system busy with either kernel alone, so nothing is lost by running the
            // right-context walk BEHIND the left-context extension -- and it can then skip the clusters whose chains came
            // out too short to ever print (see the gate in k_right): straight into the image, one request in nine saved.
            // A small call set (up to 524288 clusters: a rank's share) walks every cluster: its records then equal the reference's
            // cluster for cluster, which is what the window checks against the oracle compare (gated: 0.47 instead of 0.48 ms per pass
            // at 98 k clusters, 0.84 / 0.86 at 197 k, 1.52 / 1.60 at 395 k; not waiting for the extension: no gain).
            const uint64_t gate_min = getenv("EB_RIGHT_GATE_MIN") ? strtoull(getenv("EB_RIGHT_GATE_MIN"), nullptr, 10) : 524288; // tests: the gate on small call sets
            const bool gated = x.ns_raw > gate_min;
            EB_HIP(hipStreamWaitEvent(sC, x.e[E_EXT_DONE], 0));
            EB_HIP(hipEventRecord(x.e[E_RIGHT0], sC));
            hipLaunchKernelGGL(kern_right, dim3(grid_of(x.ns * (uint64_t)right_lanes)), dim3(256), 0, sC, x1, x2, df, p, x.sorted.as<Survivor>(), x.d_ns,
                               x.choff.as<uint32_t>(), (uint32_t)x.chain_base, x.d_recs, x.d_right, rw, x.cnt.as<ScanCounters>()->req_right,
                               gated ? (const eb_chain*)x.d_crec : (const eb_chain*)nullptr);
        }
        EB_HIP(hipEventRecord(x.e[E_RIGHT_DONE], sC));
        EB_HIP(hipEventRecord(ctx->last_right, sC));
        EB_HIP(hipGetLastError());
    }
    begin_trace_at("context-kernels");
    const uint64_t ns = job->ns, n_chains = job->n_chains;
    struct { int32_t k_left, k_right; } prm_{job->k_left, job->k_right};
    auto* prm = &prm_;
    hipEvent_t ev_end = job->ev_end;
    // ---- the call set in host memory (unless the caller keeps it on the device): one page-locked blob; with one shard
    // the device image is copied in one piece, with sub-shards each shard's sections as soon as its kernels are done
    const bool device_only = (options & EB_SCAN_DEVICE_ONLY) != 0;
    if (ns == 0 && S == 1 && !job->has_blob) {
        // nothing survived: the device image is just a header
        SubShard& x = sh[0];
        if ((rc = x.blob.alloc(sizeof(eb_calls_header))) != EB_OK) return rc;
        job->has_blob = true;
        job->blob_bytes = sizeof(eb_calls_header);
        job->header = eb_calls_header{EB_CALLS_MAGIC, 0, 0, (uint64_t)(uint32_t)prm->k_left, (uint64_t)(uint32_t)prm->k_right, {0, 0, 0}};
        hipLaunchKernelGGL(k_blob_header, dim3(1), dim3(64), 0, sD, job->header, x.blob.as<uint64_t>());
    }
    eb_calls* calls = nullptr;
    struct CallGuard { eb_calls** c; ~CallGuard() { if (*c) eb_calls_free(*c); } } cg{&calls};
    EB_HIP(hipStreamWaitEvent(sD, sh[(size_t)S - 1].e[E_SCAN_DONE], 0));
    for (int k = 0; k < S; ++k) {
        SubShard& x = sh[(size_t)k];
        if (x.ns == 0) continue;
        EB_HIP(hipStreamWaitEvent(sD, x.e[E_EXT_DONE], 0));
        EB_HIP(hipStreamWaitEvent(sD, x.e[E_RIGHT_DONE], 0));
    }
    for (int k = 0; k < S; ++k) { // request counts of the context kernels -> the job's pinned words [1 + 2k, 2 + 2k]
        SubShard& x = sh[(size_t)k];
        if (x.ns == 0) continue;
        hipLaunchKernelGGL(k_req_sum, dim3(1), dim3(EB_REQ_SLOTS), 0, sD, x.cnt.as<ScanCounters>());
        EB_HIP(hipMemcpyAsync(job->h_bad + 1 + 2 * k, x.cnt.as<ScanCounters>()->req_sum, 16, hipMemcpyDeviceToHost, sD));
    }
    if ((options & EB_SCAN_PIPELINED) && ns && job->sh.size() && ns > ctx->scratch_reserved + ctx->scratch_reserved / 8) {
        // Several jobs of this size will be alive at once (begun, context kernels running, being handed over, being read) and each
        // holds its own set of device temporaries.  The cache grows by one set whenever the number of live jobs reaches a new
        // maximum -- a dozen hipMallocs, 16-19 ms, in the middle of the pipeline (round 3: one begin call of 19 ms among 2.9 ms
        // ones, at a different step of every run).  A pipelining caller pays for a ring of sets once, with its first call set.
        // Bounded: the ring may take an eighth of the HBM that is free right now, no more (a call set of tens of millions of
        // clusters on a device the index nearly fills reserves nothing and pays the mallocs as its pipeline fills), and a failed
        // allocation ends the reservation (eb_scratch_get empties the whole cache before it gives up).
        size_t set_bytes = 0, free_b = 0, total_b = 0;
        for (SubShard& x : job->sh)
            for (DevMem* m : {&x.cnt, &x.part, &x.surv, &x.sorted, &x.bcnt, &x.boff, &x.tmp, &x.nch, &x.choff, &x.chains, &x.recs, &x.crec, &x.left, &x.right, &x.blob,
                              &x.slots, &x.bcount})
                if (m->p) set_bytes += m->cap;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        std::vector<std::pair<void*, size_t>> held;
        bool failed = 5 * set_bytes > free_b / 8;
        for (int copy = 0; copy < 5 && !failed; ++copy)
            for (SubShard& x : job->sh)
                for (DevMem* m : {&x.cnt, &x.part, &x.surv, &x.sorted, &x.bcnt, &x.boff, &x.tmp, &x.nch, &x.choff, &x.chains, &x.recs, &x.crec, &x.left, &x.right, &x.blob,
                                  &x.slots, &x.bcount}) {
                    if (!m->p || failed) continue;
                    size_t cap = 0;
                    void* q = eb_scratch_get(b1->device, m->cap - m->cap / 16, &cap); // a block of this job's class (a cached one of that class does as well)
                    if (q) held.emplace_back(q, cap);
                    else failed = true;
                }
        for (const auto& h : held) eb_scratch_put(b1->device, h.first, h.second);
        ctx->scratch_reserved = ns; // (also after a refusal: asked once per call-set size, not once per job)
    }
    if (!device_only) {
        if ((options & EB_SCAN_PIPELINED) && ns) { // several call sets of this size will be alive at once: pin their blobs now, not in mid-pipeline
            const uint64_t need = eb_calls_blob_size(ns, n_chains, prm->k_left, prm->k_right);
            if (need > ctx->pinned_reserved) { // (at most EB_PINNED_RING_BYTES page-locked in all: larger call sets pin their blobs one by one)
                if (8 * (need + need / 16) <= EB_PINNED_RING_BYTES)
                    eb_pinned_reserve(need + need / 16, 8, job->has_blob ? sh[0].blob.p : nullptr, job->has_blob ? job->blob_bytes : 0);
                ctx->pinned_reserved = need + need / 16;
            }
        }
        begin_trace_at("reserves");
        calls = eb_calls_alloc(ns, n_chains, prm->k_left, prm->k_right, ns != 0);
        if (!calls) return EB_ERR_NOMEM;
        begin_trace_at("host-blob");
        if (job->has_blob && ns) {
            EB_HIP(hipMemcpyAsync(calls->blob, sh[0].blob.p, job->blob_bytes, hipMemcpyDeviceToHost, sD));
        } else {
            for (int k = 0; k < S; ++k) {
                SubShard& x = sh[(size_t)k];
                if (x.ns == 0) continue;
                EB_HIP(hipMemcpyAsync(calls->clusters + x.rec_base, x.d_recs, x.ns * sizeof(eb_cluster), hipMemcpyDeviceToHost, sD));
                EB_HIP(hipMemcpyAsync(calls->right + x.rec_base * rw, x.d_right, x.ns * (size_t)rw * 8, hipMemcpyDeviceToHost, sD));
                if (x.n_chains) {
                    EB_HIP(hipMemcpyAsync(calls->chains + x.chain_base, x.d_crec, x.n_chains * sizeof(eb_chain), hipMemcpyDeviceToHost, sD));
                    EB_HIP(hipMemcpyAsync(calls->left + x.chain_base * lw, x.d_left, x.n_chains * (size_t)lw * 8, hipMemcpyDeviceToHost, sD));
                }
            }
        }
    }
    job->ns = ns;
    job->n_chains = n_chains;
    if ((options & EB_SCAN_WIRE) && job->has_blob && (rc = wire_enqueue(job, sD, job->h_bad, 4)) != EB_OK) return rc;
    EB_HIP(hipEventRecord(ev_end, sD));
    job->calls = calls;
    calls = nullptr; // owned by the job from here on
    job->ns = ns;
    job->n_chains = n_chains;
    return EB_OK;
}

// Last part of _begin: once the ordering step's counts are on the host, lay out the call set's image, build the chain
// list, launch the context kernels and the hand-over (launch_contexts).
static int finish_begin(eb_scan_job* job) {
    eb_index* b1 = job->b1;
    EB_HIP(hipSetDevice(b1->device));
    const hipStream_t sP = job->s_order;
    const int S = job->S;
    const uint32_t lw = job->lw, rw = job->rw;
    uint64_t rec_total = 0, chain_total = 0;
    int rc;
    for (int k = 0; k < S; ++k) {
        SubShard& x = job->sh[(size_t)k];
        x.rec_base = rec_total;
        x.chain_base = chain_total;
        if (x.ns_raw == 0) continue;
        EB_HIP(hipEventSynchronize(x.e[E_PREP_CNT]));
        begin_trace_at("ordering-counts");
        x.ns = x.h_last[0];
        x.n_chains = x.h_last[1];
        if (x.ns == 0) continue;
        if (chain_total + x.n_chains > 0xFFFFFFFFull) { eb_set_error("too many chains in one call; shard the range"); return EB_ERR_STATE; }
        const size_t left_bytes = x.n_chains * (size_t)lw * 8, right_bytes = x.ns * (size_t)rw * 8;
        if ((rc = x.chains.alloc(x.n_chains * 8)) != EB_OK) return rc;
        if (S == 1) {
            // one shard: the kernels write straight into a contiguous image of the call set (the wire / host format)
            const uint64_t bytes = eb_calls_blob_size(x.ns, x.n_chains, job->k_left, job->k_right);
            if ((rc = x.blob.alloc(bytes)) != EB_OK) return rc;
            uint8_t* q = x.blob.as<uint8_t>() + sizeof(eb_calls_header);
            x.d_recs = reinterpret_cast<eb_cluster*>(q);
            q += x.ns * sizeof(eb_cluster);
            x.d_crec = reinterpret_cast<eb_chain*>(q);
            q += (x.n_chains * sizeof(eb_chain) + 7) & ~(size_t)7;
            x.d_left = reinterpret_cast<uint64_t*>(q);
            q += left_bytes;
            x.d_right = reinterpret_cast<uint64_t*>(q);
            job->has_blob = true;
            job->blob_bytes = bytes;
            job->header = eb_calls_header{EB_CALLS_MAGIC, x.ns, x.n_chains, (uint64_t)(uint32_t)job->k_left, (uint64_t)(uint32_t)job->k_right, {0, 0, 0}};
            EB_HIP(hipMemsetAsync(x.blob.p, 0, bytes, sP));
            hipLaunchKernelGGL(k_blob_header, dim3(1), dim3(64), 0, sP, job->header, x.blob.as<uint64_t>());
        } else {
            if ((rc = x.recs.alloc(x.ns * sizeof(eb_cluster))) != EB_OK || (rc = x.crec.alloc(x.n_chains * sizeof(eb_chain))) != EB_OK ||
                (rc = x.left.alloc(left_bytes)) != EB_OK || (rc = x.right.alloc(right_bytes)) != EB_OK)
                return rc;
            x.d_recs = x.recs.as<eb_cluster>();
            x.d_crec = x.crec.as<eb_chain>();
            x.d_left = x.left.as<uint64_t>();
            x.d_right = x.right.as<uint64_t>();
            EB_HIP(hipMemsetAsync(x.left.p, 0, left_bytes, sP));
            EB_HIP(hipMemsetAsync(x.right.p, 0, right_bytes, sP));
        }
        hipLaunchKernelGGL(k_chain_list, dim3(grid_of(x.ns)), dim3(256), 0, sP, x.sorted.as<Survivor>(), x.choff.as<uint32_t>(), x.ns, x.chains.as<uint64_t>());
        EB_HIP(hipEventRecord(x.e[E_PREP_DONE], sP));
        x.d_ns = x.boff.as<uint32_t>() + x.n_buckets;
        rec_total += x.ns;
        chain_total += x.n_chains;
    }
    job->ns = rec_total;
    job->n_chains = chain_total;
    begin_trace_at("image-laid-out");
    return launch_contexts(job);
}

// the held-back last part of `job` (EB_SCAN_PIPELINED), if it is still pending; ctx->half_mu is held by the caller
static void finish_held(eb_scan_ctx* ctx, eb_scan_job* job) {
    if (!job->half) return;
    job->half = false;
    if (ctx->half_job == job) ctx->half_job = nullptr;
    job->rc_tail = finish_begin(job);
    if (job->rc_tail != EB_OK) snprintf(job->err_tail, sizeof job->err_tail, "%s", eb_last_error());
}
static int finish_pending(eb_scan_ctx* ctx) { // called by the next _begin on the index; a failure surfaces when that job is ended
    std::lock_guard<std::mutex> lk(ctx->half_mu);
    if (ctx->half_job) finish_held(ctx, ctx->half_job);
    return EB_OK;
}
static int ensure_begun(eb_scan_job* job) { // every entry point that takes a job
    eb_scan_ctx* ctx = job->b1->scan_ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->half_mu);
        finish_held(ctx, job);
    }
    if (job->rc_tail != EB_OK) eb_set_error("%s", job->err_tail);
    return job->rc_tail;
}

namespace {
// EB_BEGIN_TRACE=<ms>: host-side timeline of every _begin call that takes longer than <ms> (stderr)
struct BeginTrace {
    double limit_ms = 0, t0 = 0;
    char line[512];
    size_t len = 0;
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    BeginTrace() { if (const char* e = getenv("EB_BEGIN_TRACE")) { limit_ms = atof(e); t0 = now(); } line[0] = 0; }
    void at(const char* what) {
        if (limit_ms <= 0 || len > sizeof line - 48) return;
        len += (size_t)snprintf(line + len, sizeof line - len, " %s %.2f", what, now() - t0);
    }
    ~BeginTrace() { if (limit_ms > 0 && now() - t0 > limit_ms) fprintf(stderr, "[begin] %.2f ms:%s\n", now() - t0, line); }
};
thread_local BeginTrace* g_begin_trace = nullptr;
}  // namespace
static void begin_trace_at(const char* what) { if (g_begin_trace) g_begin_trace->at(what); }

extern "C" int eb_scan_clusters_begin_ex(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* prm,
                                         uint32_t options, eb_scan_job** out_job) {
    if (!b1 || !f || !prm || !out_job) { eb_set_error("null argument"); return EB_ERR_ARG; }
    BeginTrace bt;
    struct TraceScope { TraceScope(BeginTrace* b) { g_begin_trace = b; } ~TraceScope() { g_begin_trace = nullptr; } } trace_scope(&bt);
    *out_job = nullptr;
    const uint64_t n = b1->n + (b2 ? b2->n : 0);
    if (f->n != n) { eb_set_error("flag vectors cover %llu positions, BWT has %llu", (unsigned long long)f->n, (unsigned long long)n); return EB_ERR_ARG; }
    if (b1->device != f->device || (b2 && b2->device != b1->device)) { eb_set_error("handles live on different devices"); return EB_ERR_ARG; }
    if (b2 && !f->d_da) { eb_set_error("two-BWT mode needs flags built from both indexes"); return EB_ERR_STATE; }
    // Two-BWT mode addresses the BWTs by document-array ranks (row i of the merge is row rank1(DA, i) of the second BWT,
    // ebwt2InDel.cpp:1431-1432): that stays inside them iff the array holds exactly |bwt2| ones.  Two files that are not the
    // eBWTs of two collections can leave the traversal short of that (the reference then reads past its vectors).
    if (b2 && f->da_ones != b2->n) {
        eb_set_error("the document array of the merge holds %llu ones but the second BWT %llu symbols: not the eBWTs of two collections",
                     (unsigned long long)f->da_ones, (unsigned long long)b2->n);
        return EB_ERR_STATE;
    }
    if (prm->k_left < 1 || prm->k_right < 1 || prm->mcov_out < 1) { eb_set_error("parameters not resolved (call eb_params_resolve)"); return EB_ERR_ARG; }
    if (!f->derived) { eb_set_error("flags of a sharded build must be combined and finalised first (eb_flags_finalize)"); return EB_ERR_STATE; }
    if (end > n) end = n;
    if (begin > end) begin = end;
    EB_HIP(hipSetDevice(b1->device));
    if (!b2) { // modes 1 / 3 read symbols in the scan: 2-bit planes, derived from the index on first use
        int rcp = eb_index_scan_planes(b1);
        if (rcp != EB_OK) return rcp;
    } else if (!f->d_mxy) { // mode 2: planes of the two BWTs merged by the document array, derived on first use
        const size_t bytes = eb_bit_alloc_words(f->n) * sizeof(uint4);
        hipError_t e = hipMalloc(&f->d_mxy, bytes);
        if (e != hipSuccess) { eb_set_error("merged scan plane allocation failed (%zu bytes): %s", bytes, hipGetErrorString(e)); return EB_ERR_NOMEM; }
        EB_HIP(hipMemsetAsync(f->d_mxy, 0, bytes, f->stream));
        hipLaunchKernelGGL(k_merge_planes, dim3(4096), dim3(256), 0, f->stream, b1->dev(), b2->dev(), f->dev(), f->d_mxy);
        EB_HIP(hipStreamSynchronize(f->stream));
        EB_HIP(hipGetLastError());
    }

    // a caller that keeps scanning an index gets the FL table for its right-context walks (third job; eb_index.hip)
    eb_index_note_phase4_job(b1);
    if (b2) eb_index_note_phase4_job(b2);

    ScanParams p;
    p.mode = b2 ? 2 : (f->d_da ? 3 : 1);
    p.mcov = (uint32_t)prm->mcov_out;
    p.min_len = 2u * (uint32_t)prm->mcov_out;
    p.q = prm->max_variants_per_position;
    p.k_left = prm->k_left;
    p.k_right = prm->k_right;
    p.begin = begin;
    p.end = end;
    p.n = n;
#ifdef EB_SCAN_EXPERIMENTS
    p.dbg = getenv("EB_SCAN_DEBUG") ? (uint32_t)atoi(getenv("EB_SCAN_DEBUG")) : 0u;
#endif
    eb_dev_index x1 = b1->dev(), x2 = b2 ? b2->dev() : b1->dev();
    eb_dev_flags df = f->dev();

    // ---- pipelined driver.  The range is cut into sub-shards; their scans are queued back to back on the index
    // stream while, for each finished scan, survivors are sorted on a preparation stream and the two context kernels
    // run on two more streams.  The streaming scan of shard k+1 thus overlaps the dependent-load chains of shard k,
    // and the record download of shard k overlaps the chains of shard k+1.
    eb_scan_ctx* ctx = b1->scan_ctx;
    if (!ctx) {
        ctx = new eb_scan_ctx();
        b1->scan_ctx = ctx;
        EB_HIP(hipStreamCreateWithFlags(&ctx->prep, hipStreamNonBlocking));
        EB_HIP(hipStreamCreateWithFlags(&ctx->ext, hipStreamNonBlocking));
        EB_HIP(hipStreamCreateWithFlags(&ctx->right, hipStreamNonBlocking));
        EB_HIP(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
        EB_HIP(hipEventCreateWithFlags(&ctx->last_ext, hipEventDisableTiming));
        EB_HIP(hipEventCreateWithFlags(&ctx->last_right, hipEventDisableTiming));
        EB_HIP(hipEventRecord(ctx->last_ext, ctx->ext));
        EB_HIP(hipEventRecord(ctx->last_right, ctx->right));
    }
    // THE SCAN BESIDE THE PREVIOUS JOB'S CONTEXT KERNELS (round 3).  The scan is bound by vector instructions (~480 per 64-position
    // word), the context kernels by the fabric's rate of random 128-byte lines with next to no arithmetic: run one after the
    // other they add up (1.05 + 2.07 ms at 7.55 G positions), run side by side on the same CUs they do not.  Round 2 tried it with
    // the scan's persistent grid of 8 workgroups per CU and lost (4.4 against 3.9 ms): whichever kernel arrives first owns every
    // wave slot and the other starves.  What works is a PARTITION of each CU: the scan holds overlap_wgs workgroups per CU (its
    // grid is 256 x overlap_wgs, and a dynamic LDS request of 160 KB / overlap_wgs keeps one more from fitting), issues the loads
    // of its next tile before it works on the current one (k_scan<.., true>: half the waves must cover the same latency), and
    // the context kernels -- no LDS, 16 waves per CU left -- keep the memory system busy.  Steady state at 7.55 G positions:
    // 2.85 ms per pass against 3.27 on the same box (profiles/r03_overlap_ab.txt).
    // Applied when the caller keeps jobs coming (EB_SCAN_PIPELINED) and the previous job's context kernels are still running;
    // a job with nothing to share the chip with gets the full-width scan.  EB_SCAN_OVERLAP=0: never; =k: always, k per CU.
    const unsigned overlap_wgs = [&]() -> unsigned {
        const char* e = getenv("EB_SCAN_OVERLAP");
        const int v = e ? atoi(e) : -1;
        if (v >= 0) return (unsigned)(v > 8 ? 8 : v);
        if (!(options & EB_SCAN_PIPELINED)) return 0u;
        // (4 per CU in round 3, when the walk took as long as the extension; with the FL table the context kernels are the shorter
        // side and the scan gets every slot it can use: 2.60 / 2.46 / 2.36 / 2.22 ms per step for 4 / 5 / 6 / 8, profiles/r04_overlap_ab.txt)
        const unsigned share = (b1->d_psi && (!b2 || b2->d_psi)) ? 8u : 4u;
        return hipEventQuery(ctx->last_right) == hipErrorNotReady ? share : 0u;
    }();
    // overlap mode: the ordering of the survivors follows the scan on ITS stream, so that it does not queue behind the previous
    // job's context kernels (which own the preparation stream) -- the context kernels of this job do
    const hipStream_t sA = b1->stream, sP = overlap_wgs ? sA : ctx->prep;
    uint64_t w_lo = begin >> 6, w_hi = (end + 63) >> 6;
    uint64_t w_total = w_hi > w_lo ? w_hi - w_lo : 0;
    // Measured at 7.55 G positions (EB_SCAN_SUBSHARDS=1/2/4/8): 6.5 / 6.1 / 6.4 / 11.0 ms per call -- the streaming scan and
    // the random-access chains compete for the same memory request rate, so overlap buys little and slows each kernel.
    // Default: one shard (kernels run alone, cleanest timings); the knob stays for experiments.
    const uint64_t small_set_max = 524288; // call sets up to this size: every cluster gets its right context, the job's last part may be held back
    int S = 1;
    if (const char* e = getenv("EB_SCAN_SUBSHARDS")) { int v = atoi(e); if (v >= 1 && v <= 16) S = v; }
    if ((uint64_t)S > w_total) S = w_total ? (int)w_total : 1;

    eb_scan_job* job = new eb_scan_job();
    struct JobGuard {
        eb_scan_job** j;
        eb_scan_ctx* ctx;
        ~JobGuard() {
            if (!*j) return;
            (void)hipDeviceSynchronize(); // nothing may still use the job's buffers
            {
                std::lock_guard<std::mutex> lk(ctx->pool_mu);
                for (hipEvent_t e : (*j)->events) ctx->events.push_back(e);
                if ((*j)->h_bad) ctx->h_bad_pool.push_back((*j)->h_bad);
            }
            if ((*j)->calls) eb_calls_free((*j)->calls);
            delete *j;
        }
    } job_guard{&job, ctx};
    job->b1 = b1;
    job->S = S;
    job->s_ext = ctx->ext;
    job->s_right = ctx->right;
    job->s_order = sP;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        if (!ctx->h_bad_pool.empty()) {
            job->h_bad = ctx->h_bad_pool.back();
            ctx->h_bad_pool.pop_back();
        }
    }
    if (!job->h_bad) {
        void* hb = nullptr;
        // [0] wire overflow count, [1 + 2k, 2 + 2k] request counts of sub-shard k (<= 16), [40, 41] ordering counts; behind the
        // first 512 bytes: the job's OWN read-back block of every sub-shard's ScanCounters (+ 64 bytes of ordering counts) -- a
        // block shared by all jobs of the index was safe only while _begin calls never overlapped (round-2 advisor finding)
        EB_HIP(hipHostMalloc(&hb, EB_JOB_PINNED_BYTES, hipHostMallocDefault));
        job->h_bad = static_cast<unsigned long long*>(hb);
    }
    memset(job->h_bad, 0, 512);
    job->k_left = prm->k_left;
    job->k_right = prm->k_right;
    job->options = options;
    job->begin = begin;
    job->end = end;
    job->sh.resize((size_t)S);
    std::vector<SubShard>& sh = job->sh;
    const size_t n_events = 2 + (size_t)E_PER_SHARD * (size_t)S;
    for (size_t i = 0; i < n_events; ++i) { // events are recycled through the context's pool
        hipEvent_t e = nullptr;
        {
            std::lock_guard<std::mutex> lk(ctx->pool_mu);
            if (!ctx->events.empty()) { e = ctx->events.back(); ctx->events.pop_back(); }
        }
        if (!e) EB_HIP(hipEventCreate(&e));
        job->events.push_back(e);
    }
    hipEvent_t ev_begin = job->events[0], ev_end = job->events[1];
    job->ev_begin = ev_begin;
    job->ev_end = ev_end;
    int rc;
    const int scan_prefetch = getenv("EB_SCAN_PREFETCH") ? atoi(getenv("EB_SCAN_PREFETCH")) : -1; // -1: with EB_SCAN_OVERLAP only
    const bool prefetch = scan_prefetch < 0 ? overlap_wgs != 0 : scan_prefetch != 0;
    auto kern = prefetch ? (p.mode == 1 ? k_scan<1, true> : (p.mode == 2 ? k_scan<2, true> : k_scan<3, true>))
                         : (p.mode == 1 ? k_scan<1, false> : (p.mode == 2 ? k_scan<2, false> : k_scan<3, false>));
    // The scan hands its survivors over in BWT-ordered batches (ScanParams.ordered) unless a batch overflows, which it
    // reports (ScanCounters.spilled): that scan is then run again the unordered way, into one list that position buckets put
    // in order.  EB_SCAN_ORDERED=0: always the unordered way.
    const bool ordered_ok = !(getenv("EB_SCAN_ORDERED") && atoi(getenv("EB_SCAN_ORDERED")) == 0);
    // static LDS of k_scan is ~17.3 KB; with a dynamic request that brings a workgroup to 160 KB / k (minus a little), k fit a CU
    // (a workgroup may ask for 64 KB at most: k = 1 and 2 both end at two per CU)
    unsigned scan_dyn_lds = 0;
    if (overlap_wgs) {
        unsigned total = (160u * 1024u) / overlap_wgs - 256u;
        if (total > 64u * 1024u) total = 64u * 1024u;
        scan_dyn_lds = total - 18u * 1024u;
    }
    auto launch_scan = [&](SubShard& x, bool ordered) -> int {
        ScanParams q = p;
        q.begin = x.begin;
        q.end = x.end;
        int r;
        x.ordered = ordered && x.w_count != 0;
        // (overlap_wgs != 0: the scan that shares every CU with the previous job's context kernels, see above)
        const unsigned grid = scan_grid(x.w_count, overlap_wgs ? overlap_wgs : 8);
        if (x.ordered) {
            const uint64_t tiles = (x.w_count + 255) / 256;
            q.ordered = 1;
            q.tiles_per_wg = (uint32_t)((tiles + grid - 1) / grid);
            x.n_batches = (uint64_t)grid * ((q.tiles_per_wg + EB_BATCH_TILES - 1) / EB_BATCH_TILES);
            if ((r = x.slots.alloc(x.n_batches * EB_BATCH_SLOTS * sizeof(Survivor))) != EB_OK || (r = x.bcount.alloc(x.n_batches * 4)) != EB_OK) return r;
        } else {
            q.ordered = 0;
            q.tiles_per_wg = 0;
            if ((r = x.surv.alloc(x.cap * sizeof(Survivor))) != EB_OK) return r;
        }
        EB_HIP(hipMemsetAsync(x.cnt.p, 0, sizeof(ScanCounters), sA));
        if (x.ordered) EB_HIP(hipMemsetAsync(x.bcount.p, 0, x.n_batches * 4, sA)); // the batches' cursors
        EB_HIP(hipEventRecord(x.e[E_SCAN0], sA));
        if (x.w_count) {
            if ((r = x.part.alloc((size_t)grid * EB_SCAN_PARTIAL * 8)) != EB_OK) return r;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), scan_dyn_lds, sA, x1, x2, df, q, x.w_lo, x.w_count, x.surv.as<Survivor>(), x.cap,
                               x.cnt.as<ScanCounters>(), x.part.as<unsigned long long>(), x.slots.as<Survivor>(), x.bcount.as<uint32_t>());
            hipLaunchKernelGGL(k_scan_reduce, dim3(EB_SCAN_PARTIAL), dim3(256), 0, sA, x.part.as<unsigned long long>(), grid, x.cnt.as<ScanCounters>(), x.ordered ? 1 : 0);
        }
        EB_HIP(hipEventRecord(x.e[E_SCAN1], sA));
        EB_HIP(hipMemcpyAsync(x.h_cnt, x.cnt.p, sizeof(ScanCounters), hipMemcpyDeviceToHost, sA));
        EB_HIP(hipEventRecord(x.e[E_SCAN_DONE], sA));
        EB_HIP(hipGetLastError());
        return EB_OK;
    };
    // A caller that pipelines jobs gets the record download of job i under the kernels of job i+1; the streaming scan
    // itself starts only after job i's context kernels (they would just slow each other down: same memory system).
    // (measured again in round 2 with the serialised context kernels: letting the scan start under the previous job's
    // right-context walk or under both context kernels: 4.41 / 4.37 ms per step against 3.89)
    if (!overlap_wgs) {
        EB_HIP(hipStreamWaitEvent(sA, ctx->last_ext, 0));
        EB_HIP(hipStreamWaitEvent(sA, ctx->last_right, 0));
    }
    EB_HIP(hipEventRecord(ev_begin, sA));
    for (int k = 0; k < S; ++k) {
        SubShard& x = sh[(size_t)k];
        for (DevMem* m : {&x.cnt, &x.part, &x.surv, &x.sorted, &x.bcnt, &x.boff, &x.tmp, &x.nch, &x.choff, &x.chains, &x.recs, &x.crec, &x.left, &x.right, &x.blob, &x.slots, &x.bcount})
            m->device = b1->device;
        for (int i = 0; i < E_PER_SHARD; ++i) x.e[i] = job->events[2 + (size_t)E_PER_SHARD * (size_t)k + (size_t)i];
        x.h_cnt = reinterpret_cast<ScanCounters*>(reinterpret_cast<uint8_t*>(job->h_bad) + 512 + (size_t)k * (sizeof(ScanCounters) + 64));
        x.h_last = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(x.h_cnt) + sizeof(ScanCounters));
        // cut on flag-word boundaries; clusters belong to the sub-shard that holds their closing position
        uint64_t wa = w_lo + w_total * (uint64_t)k / (uint64_t)S, wb = w_lo + w_total * (uint64_t)(k + 1) / (uint64_t)S;
        x.w_lo = wa;
        x.w_count = wb - wa;
        x.begin = k == 0 ? begin : wa << 6;
        x.end = k == S - 1 ? end : wb << 6;
        x.cap = x.w_count / 16 + 4096; // survivors are ~1e-4 per position; grown on demand
        if ((rc = x.cnt.alloc(sizeof(ScanCounters))) != EB_OK) return rc;
        if ((rc = launch_scan(x, ordered_ok && EB_DBG(p) == 0)) != EB_OK) return rc;
    }

    const uint32_t lw = (uint32_t)((prm->k_left + 31) / 32), rw = (uint32_t)((prm->k_right + 31) / 32);
    job->x1 = x1;
    job->x2 = x2;
    job->df = df;
    job->p = p;
    job->lw = lw;
    job->rw = rw;
    // EB_SCAN_PIPELINED: the previous job's last part goes out now, behind this job's scan: the host's wait for that job's
    // ordering counts and the wait for this scan below overlap instead of adding up
    bt.at("scan-enqueued");
    if ((rc = finish_pending(ctx)) != EB_OK) return rc;
    bt.at("pending-out");
    ScanCounters c;
    memset(&c, 0, sizeof c);
    bool defer = true; // every sub-shard's call set is a small one
    for (int k = 0; k < S; ++k) {
        SubShard& x = sh[(size_t)k];
        if (S == 1) x.h_last = reinterpret_cast<uint32_t*>(job->h_bad + 40); // the job's own pinned words: a held-back tail reads them later
        for (;;) {
            EB_HIP(hipEventSynchronize(x.e[E_SCAN_DONE]));
            bt.at("scan-done");
            if (x.ordered) {
                if (!x.h_cnt->spilled) break;
                if (getenv("EB_SCAN_TRACE")) fprintf(stderr, "[scan] a batch of the ordered hand-over overflowed: scanning again unordered\n");
                ++job->reruns;
                if ((rc = launch_scan(x, false)) != EB_OK) return rc; // rare: a batch held more than its slots
                continue;
            }
            if (x.h_cnt->nsurv <= x.cap) break;
            x.cap = x.h_cnt->nsurv + 1024; // rare: rerun this sub-shard's scan with room for every survivor
            ++job->reruns;
            if ((rc = launch_scan(x, false)) != EB_OK) return rc;
        }
        ScanCounters hc = *x.h_cnt;
        x.ns_raw = hc.nsurv;
        if (x.ns_raw > 0x1FFFFFFFull) { eb_set_error("too many surviving clusters in one call; shard the range"); return EB_ERR_STATE; }
        EB_HIP(hipStreamWaitEvent(sP, x.e[E_SCAN_DONE], 0));
        c.runs_closed += hc.runs_closed;
        c.clusters += hc.clusters;
        c.bases += hc.bases;
        for (int i = 0; i < 201; ++i) c.hist[i] += hc.hist[i];
        if (x.ns_raw == 0) { x.ns = 0; continue; }
        uint32_t* d_boff = nullptr;  // exclusive offsets of the buckets / batches; [n_buckets] = survivors kept
        uint64_t n_buckets = 0;
        size_t tmp_bytes = 0, scan_bytes = 0;
        if (x.ordered) {
            // the batches lie in BWT order already: resolve the long-run candidates, count what every batch keeps, scan, and
            // let every batch sort its few dozen members into place (k_batch_compact also writes the chain counts)
            n_buckets = x.n_batches;
            x.n_buckets = n_buckets;
            if ((rc = x.bcnt.alloc((n_buckets + 1) * 4)) != EB_OK || (rc = x.boff.alloc((n_buckets + 2) * 4)) != EB_OK ||
                (rc = x.sorted.alloc(x.ns_raw * sizeof(Survivor))) != EB_OK || (rc = x.nch.alloc(x.ns_raw * 4)) != EB_OK ||
                (rc = x.choff.alloc((x.ns_raw + 3) * 4)) != EB_OK)
                return rc;
            uint32_t* d_kept = x.bcnt.as<uint32_t>();
            d_boff = x.boff.as<uint32_t>();
            EB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_kept, d_boff, (int)(n_buckets + 1), sP));
            EB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, x.nch.as<uint32_t>(), x.choff.as<uint32_t>(), (int)x.ns_raw, sP));
            if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
            if ((rc = x.tmp.alloc(tmp_bytes)) != EB_OK) return rc;
            const unsigned wave_grid = grid_of(n_buckets * 64);
            EB_HIP(hipMemsetAsync(x.nch.p, 0, x.ns_raw * 4, sP));
            hipLaunchKernelGGL(k_batch_resolve, dim3(wave_grid), dim3(256), 0, sP, x1, x2, df, p, x.slots.as<Survivor>(), x.bcount.as<uint32_t>(), n_buckets,
                               d_kept, &x.cnt.as<ScanCounters>()->dropped);
            EB_HIP(hipcub::DeviceScan::ExclusiveSum(x.tmp.p, tmp_bytes, d_kept, d_boff, (int)(n_buckets + 1), sP)); // d_boff[n_batches] = survivors kept
            hipLaunchKernelGGL(k_batch_compact, dim3((unsigned)n_buckets), dim3(256), 0, sP, x.slots.as<Survivor>(), x.bcount.as<uint32_t>(), d_boff, n_buckets,
                               x.sorted.as<Survivor>(), x.nch.as<uint32_t>());
        } else {
        // BWT order: deal the survivors into position buckets (x.ns, the count without the dropped candidates, comes back
        // with the chain count)
        const uint64_t bucket_base = x.begin & ~((1ull << EB_BUCKET_SHIFT) - 1ull);
        const uint64_t range_end = x.end > n ? n : x.end;
        n_buckets = ((range_end > bucket_base ? range_end - bucket_base : 0) >> EB_BUCKET_SHIFT) + 2;
        if (n_buckets > 0x7FFFFFFEull) { eb_set_error("range too long for one call; shard it"); return EB_ERR_STATE; }
        x.n_buckets = n_buckets;
        // bcnt = [count | fill] (zeroed), boff = offsets (n_buckets + 1)
        if ((rc = x.bcnt.alloc((n_buckets + 1) * 8)) != EB_OK || (rc = x.boff.alloc((n_buckets + 2) * 4)) != EB_OK ||
            (rc = x.sorted.alloc(x.ns_raw * sizeof(Survivor))) != EB_OK || (rc = x.nch.alloc(x.ns_raw * 4)) != EB_OK ||
            (rc = x.choff.alloc((x.ns_raw + 3) * 4)) != EB_OK)
            return rc;
        uint32_t* d_bcnt = x.bcnt.as<uint32_t>();
        uint32_t* d_bfill = d_bcnt + (n_buckets + 1);
        d_boff = x.boff.as<uint32_t>();
        EB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_bcnt, d_boff, (int)(n_buckets + 1), sP));
        EB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, x.nch.as<uint32_t>(), x.choff.as<uint32_t>(), (int)x.ns_raw, sP));
        if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
        if ((rc = x.tmp.alloc(tmp_bytes)) != EB_OK) return rc;
        EB_HIP(hipMemsetAsync(d_bcnt, 0, (n_buckets + 1) * 8, sP));
        // candidates whose run is longer than the scan's two words get their histogram by rank differences (the ones
        // that fail: begin = ~0); every survivor is counted into its bucket
        hipLaunchKernelGGL(k_long_runs, dim3(grid_of(x.ns_raw)), dim3(256), 0, sP, x1, x2, df, p, x.surv.as<Survivor>(), x.ns_raw,
                           &x.cnt.as<ScanCounters>()->dropped, bucket_base, d_bcnt);
        EB_HIP(hipcub::DeviceScan::ExclusiveSum(x.tmp.p, tmp_bytes, d_bcnt, d_boff, (int)(n_buckets + 1), sP)); // d_boff[n_buckets] = survivors kept
        hipLaunchKernelGGL(k_bucket_place, dim3(grid_of(x.ns_raw)), dim3(256), 0, sP, x.surv.as<Survivor>(), x.ns_raw, bucket_base, d_boff, d_bfill,
                           x.sorted.as<Survivor>());
        hipLaunchKernelGGL(k_bucket_fix, dim3(grid_of(n_buckets)), dim3(256), 0, sP, d_boff, n_buckets, x.sorted.as<Survivor>());
        }
        defer = defer && x.ns_raw <= small_set_max;
        if (!x.ordered) hipLaunchKernelGGL(k_surv_chains, dim3(grid_of(x.ns_raw)), dim3(256), 0, sP, x.sorted.as<Survivor>(), d_boff + n_buckets, x.ns_raw, x.nch.as<uint32_t>());
        // dense chain list: exclusive scan of the per-cluster chain counts
        EB_HIP(hipcub::DeviceScan::ExclusiveSum(x.tmp.p, tmp_bytes, x.nch.as<uint32_t>(), x.choff.as<uint32_t>(), (int)x.ns_raw, sP));
        uint32_t* d_tot = x.choff.as<uint32_t>() + x.ns_raw + 1;
        hipLaunchKernelGGL(k_prep_totals, dim3(1), dim3(1), 0, sP, d_boff + n_buckets, x.choff.as<uint32_t>(), x.nch.as<uint32_t>(), d_tot);
        EB_HIP(hipMemcpyAsync(&x.h_last[0], d_tot, 8, hipMemcpyDeviceToHost, sP));
        EB_HIP(hipEventRecord(x.e[E_PREP_CNT], sP));
        EB_HIP(hipGetLastError());
    }
    bt.at("ordering-enqueued");
    job->c = c;
    // Last part: the call set's image, the chain list, the context kernels and the hand-over.  It needs the ordering
    // step's counts on the host.  A pipelining caller (EB_SCAN_PIPELINED, one shard, small call set) gets the job back
    // now; the part goes out with the next _begin on this index, or when the job is waited for / ended.
    if ((options & EB_SCAN_PIPELINED) && S == 1 && defer) {
        std::lock_guard<std::mutex> lk(ctx->half_mu);
        job->half = true;
        ctx->half_job = job;
    } else if ((rc = finish_begin(job)) != EB_OK)
        return rc;
    bt.at("contexts-enqueued");
    *out_job = job;
    job = nullptr;
    return EB_OK;
}

extern "C" int eb_scan_clusters_end(eb_scan_job* job, eb_calls** out, eb_scan_stats* stats) {
    if (!job) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (out) *out = nullptr;
    eb_index* b1 = job->b1;
    eb_scan_ctx* ctx = b1->scan_ctx;
    EB_HIP(hipSetDevice(b1->device));
    const int tail_rc = ensure_begun(job);
    if (tail_rc != EB_OK) (void)hipDeviceSynchronize(); // whatever part of the tail went out must be done with the job's buffers
    // ev_end lies behind both context kernels' end events (launch_contexts); the streams themselves may already hold the
    // next job's kernels, begun on another host thread, and are not waited for
    hipError_t werr = tail_rc == EB_OK ? hipEventSynchronize(job->ev_end) : hipSuccess;
    float ms_scan = 0, ms_ext = 0, ms_right = 0, ms_total = 0, ms_context = 0;
    uint64_t req_ext = 0, req_right = 0;
    if (werr == hipSuccess && tail_rc == EB_OK) {
        (void)hipEventElapsedTime(&ms_total, job->ev_begin, job->ev_end);
        for (int k = 0; k < job->S; ++k) {
            SubShard& x = job->sh[(size_t)k];
            float t = 0;
            (void)hipEventElapsedTime(&t, x.e[E_SCAN0], x.e[E_SCAN1]);
            ms_scan += t;
            if (x.ns == 0) continue;
            (void)hipEventElapsedTime(&t, x.e[E_EXT0], x.e[E_EXT_DONE]);
            ms_ext += t;
            (void)hipEventElapsedTime(&t, x.e[E_RIGHT0], x.e[E_RIGHT_DONE]);
            ms_right += t;
            // the two kernels run side by side on two streams: first start .. last end
            float span = 0;
            for (int a : {E_EXT0, E_RIGHT0})
                for (int b : {E_EXT_DONE, E_RIGHT_DONE}) {
                    (void)hipEventElapsedTime(&t, x.e[a], x.e[b]);
                    if (t > span) span = t;
                }
            ms_context += span;
            req_ext += job->h_bad[1 + 2 * k];
            req_right += job->h_bad[2 + 2 * k];
        }
    }
    if (werr == hipSuccess && tail_rc == EB_OK && getenv("EB_SCAN_TRACE") && job->S == 1) {
        SubShard& x = job->sh[0];
        auto at = [&](hipEvent_t e) { float t = 0; (void)hipEventElapsedTime(&t, job->ev_begin, e); return t; };
        if (x.ns)
            fprintf(stderr, "[scan] ms since begin: scan %.3f-%.3f  counters on host %.3f  sort+offsets %.3f  lists %.3f  extend %.3f-%.3f  right %.3f-%.3f  end %.3f\n",
                    at(x.e[E_SCAN0]), at(x.e[E_SCAN1]), at(x.e[E_SCAN_DONE]), at(x.e[E_PREP_CNT]), at(x.e[E_PREP_DONE]), at(x.e[E_EXT0]), at(x.e[E_EXT_DONE]),
                    at(x.e[E_RIGHT0]), at(x.e[E_RIGHT_DONE]), at(job->ev_end));
    }
    if (stats && werr == hipSuccess && tail_rc == EB_OK) {
        memset(stats, 0, sizeof *stats);
        stats->begin = job->begin;
        stats->end = job->end;
        stats->runs_closed = job->c.runs_closed;
        stats->clusters = job->c.clusters;
        stats->bases_in_runs = job->c.bases;
        stats->survivors = job->ns;
        stats->chains = job->n_chains; // one left-context chain per frequent symbol
        for (int k = 0; k < 201; ++k) stats->hist[k] = job->c.hist[k];
        stats->ms_scan = ms_scan;
        stats->ms_extend = ms_ext;
        stats->ms_right = ms_right;
        stats->ms_total = ms_total;
        stats->scan_launches = (uint32_t)job->S;
        stats->ms_context = ms_context;
        stats->requests_extend = req_ext;
        stats->requests_right = req_right;
        stats->scan_reruns = job->reruns;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        for (hipEvent_t e : job->events) ctx->events.push_back(e);
        if (job->h_bad) ctx->h_bad_pool.push_back(job->h_bad);
    }
    eb_calls* calls = job->calls;
    const std::string job_err = job->err_tail;
    delete job; // releases the device scratch to the cache
    if (tail_rc != EB_OK) { // the held-back last part failed: its message is the job's
        if (calls) eb_calls_free(calls);
        eb_set_error("%s", job_err.c_str());
        return tail_rc;
    }
    if (werr != hipSuccess) {
        eb_set_error("phase-4 job failed: %s", hipGetErrorString(werr));
        if (calls) eb_calls_free(calls);
        return EB_ERR_DEVICE;
    }
    if (out) *out = calls;
    else if (calls) eb_calls_free(calls);
    return EB_OK;
}

extern "C" int eb_scan_clusters_begin(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* prm,
                                      eb_scan_job** out_job) {
    return eb_scan_clusters_begin_ex(b1, b2, f, begin, end, prm, 0u, out_job);
}

extern "C" int eb_scan_job_wait(eb_scan_job* job) {
    if (!job) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(job->b1->device));
    int rc = ensure_begun(job);
    if (rc != EB_OK) return rc;
    EB_HIP(hipEventSynchronize(job->ev_end));
    return EB_OK;
}

extern "C" int eb_scan_job_device_blob(eb_scan_job* job, const void** d_data, uint64_t* len) {
    if (!job || !d_data || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(job->b1->device));
    { int rc = ensure_begun(job); if (rc != EB_OK) return rc; }
    if (!job->has_blob) { eb_set_error("the job was split into sub-shards: no contiguous device image"); return EB_ERR_STATE; }
    *d_data = job->sh[0].blob.p;
    *len = job->blob_bytes;
    return EB_OK;
}

// enqueues the packing of the compact wire image on `st` (after the records are complete there); h_bad: pinned word
// version 4: the tightest image; version 3: the fallback with full-width fields
static int wire_enqueue(eb_scan_job* job, hipStream_t st, unsigned long long* h_bad, int version) {
    const uint64_t count = job->ns, n_chains = job->n_chains;
    const uint64_t lw = ((uint64_t)job->k_left + 31) / 32, rw = ((uint64_t)job->k_right + 31) / 32;
    const uint64_t bytes = version == 4 ? eb_wire4_size(count, n_chains, job->k_left, job->k_right) : eb_wire_size(count, n_chains, job->k_left, job->k_right);
    job->wire.device = job->b1->device;
    int rc = job->wire.alloc(bytes + 16); // + the overflow counter
    if (rc != EB_OK) return rc;
    uint8_t* q = job->wire.as<uint8_t>();
    unsigned long long* d_bad = reinterpret_cast<unsigned long long*>(q + ((bytes + 7) & ~7ull));
    EB_HIP(hipMemsetAsync(d_bad, 0, 8, st));
    eb_calls_header h = job->header;
    h.magic = version == 4 ? EB_WIRE4_MAGIC : EB_WIRE_MAGIC;
    hipLaunchKernelGGL(k_blob_header, dim3(1), dim3(64), 0, st, h, reinterpret_cast<uint64_t*>(q));
    if (count) {
        SubShard& x = job->sh[0];
        const uint64_t most = count > n_chains ? count : n_chains;
        uint8_t* ctx_words;
        if (version == 4) {
            eb_wire4_cluster* wc = reinterpret_cast<eb_wire4_cluster*>(q + sizeof(eb_calls_header));
            eb_wire4_chain* wk = reinterpret_cast<eb_wire4_chain*>(q + sizeof(eb_calls_header) + count * sizeof(eb_wire4_cluster));
            const uint64_t recs = (count * sizeof(eb_wire4_cluster) + n_chains * sizeof(eb_wire4_chain) + 7) & ~7ull;
            ctx_words = q + sizeof(eb_calls_header) + recs;
            if (recs >= 8) EB_HIP(hipMemsetAsync(ctx_words - 8, 0, 8, st)); // the padding behind the chains
            hipLaunchKernelGGL(k_pack_wire4, dim3(grid_of(most)), dim3(256), 0, st, x.d_recs, count, x.d_crec, n_chains, wc, wk, d_bad);
        } else {
            eb_wire_cluster* wc = reinterpret_cast<eb_wire_cluster*>(q + sizeof(eb_calls_header));
            eb_wire_chain* wk = reinterpret_cast<eb_wire_chain*>(q + sizeof(eb_calls_header) + count * sizeof(eb_wire_cluster));
            ctx_words = reinterpret_cast<uint8_t*>(wk) + n_chains * sizeof(eb_wire_chain);
            hipLaunchKernelGGL(k_pack_wire, dim3(grid_of(most)), dim3(256), 0, st, x.d_recs, count, x.d_crec, n_chains, wc, wk, d_bad);
        }
        EB_HIP(hipMemcpyAsync(ctx_words, x.d_left, n_chains * lw * 8, hipMemcpyDeviceToDevice, st));
        EB_HIP(hipMemcpyAsync(ctx_words + n_chains * lw * 8, x.d_right, count * rw * 8, hipMemcpyDeviceToDevice, st));
    }
    EB_HIP(hipMemcpyAsync(h_bad, d_bad, 8, hipMemcpyDeviceToHost, st));
    EB_HIP(hipGetLastError());
    job->wire_bytes = bytes;
    return EB_OK;
}

extern "C" int eb_scan_job_device_wire(eb_scan_job* job, const void** d_data, uint64_t* len) {
    if (!job || !d_data || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(job->b1->device));
    { int rc = ensure_begun(job); if (rc != EB_OK) return rc; }
    if (!job->has_blob) { eb_set_error("the job was split into sub-shards: no contiguous device image"); return EB_ERR_STATE; }
    EB_HIP(hipEventSynchronize(job->ev_end)); // the records (and, with EB_SCAN_WIRE, the packing) are complete
    hipStream_t st = job->b1->scan_ctx->copy;
    if (!job->wire.p) {                       // not asked for at _begin: pack now
        int rc = wire_enqueue(job, st, job->h_bad, 4);
        if (rc != EB_OK) return rc;
        EB_HIP(hipStreamSynchronize(st));
    }
    if (*job->h_bad && !job->wire_full_width) { // a field does not fit the tight image: full-width records
        job->wire_full_width = true;
        int rc = wire_enqueue(job, st, job->h_bad, 3);
        if (rc != EB_OK) return rc;
        EB_HIP(hipStreamSynchronize(st));
    }
    if (*job->h_bad) { eb_set_error("%llu records do not fit the compact wire image", *job->h_bad); return EB_ERR_STATE; }
    *d_data = job->wire.p;
    *len = job->wire_bytes;
    return EB_OK;
}

extern "C" void eb_scan_clusters_cancel(eb_scan_job* job) {
    eb_calls* c = nullptr;
    if (job) (void)eb_scan_clusters_end(job, &c, nullptr);
    if (c) eb_calls_free(c);
}

extern "C" int eb_scan_clusters(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* prm,
                                eb_calls** out, eb_scan_stats* stats) {
    if (!out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *out = nullptr;
    eb_scan_job* job = nullptr;
    int rc = eb_scan_clusters_begin(b1, b2, f, begin, end, prm, &job);
    if (rc != EB_OK) return rc;
    return eb_scan_clusters_end(job, out, stats);
}
