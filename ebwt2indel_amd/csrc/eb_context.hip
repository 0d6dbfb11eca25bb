// eb_context.hip -- phase 4, second half: the per-cluster backward left-context extension and the right-context walk.
//
// Replaces extract_consensus / consensus_letter (ebwt2InDel.cpp:243-319) and extract_dna (:325-342) as find_variants calls
// them (:967-988; two BWTs :880-912; merged BWT + DA :1063-1075).
//
//   k_extend1  one lane per (cluster, frequent symbol) chain: k_left-1 dependent LF(range) steps, arg-max with ties A<C<G<T
//   k_extend   the same by a quad per chain (two BWTs, and indexes beyond 2^33 symbols)
//   k_right    one lane (FL table) or a quad (directory select) per cluster: first position with thrR, then k_right x (F, FL)
//   k_finish   the job's last kernel: request counts -> the job's report in page-locked host memory
// All of them read their bounds and the sections of the call set's image from the JobLayout the ordering step left in device
// memory (eb_scan_internal.h): grids are sized by capacity, lanes beyond the live counts leave at once.
#include "eb_scan_internal.h"

namespace {

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
                                               const uint64_t* __restrict__ chains, ScanCounters* __restrict__ cnt, uint8_t* __restrict__ blob,
                                               uint32_t left_words) {
    const uint32_t j = threadIdx.x & 3u;
    uint64_t ch = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    uint32_t n_req = 0; // 64-byte block requests issued by this quad (counted by its lane 0)
    const uint64_t n_chains = cnt->lay.n_chains;
    eb_chain* __restrict__ crec = reinterpret_cast<eb_chain*>(blob + cnt->lay.off_crec);
    uint64_t* __restrict__ left = reinterpret_cast<uint64_t*>(blob + cnt->lay.off_left);
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
            // a context shorter than k_left is dropped by the writer (ebwt2InDel.cpp:317): blank (every word is written: the image
            // is not cleared beforehand)
            for (uint32_t wi = 0; wi < left_words; ++wi) out[wi] = 0;
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
    count_requests(cnt->req_ext, j == 0 ? n_req : 0u);
}

// The same chains, ONE lane per chain (round 4).  The quad form above spends ~113 vector instructions per LF step and wave on 16
// chains -- every lane runs the whole instruction stream on a quarter of the block -- and beside the scan the pass is short of
// vector issue as much as of memory lines.  Here a lane loads the whole 64-byte block (four 16-byte loads from one line: one
// fabric request all the same) and counts all four letters at both ends of its range: ~170 instructions per step and wave for
// 64 chains.  Same results by construction (the counts are those of eb_rank4).
template <bool TWO>
__global__ void __launch_bounds__(256) k_extend1(eb_dev_index x1, eb_dev_index x2, ScanParams p, const Survivor* __restrict__ surv,
                                                const uint64_t* __restrict__ chains, ScanCounters* __restrict__ cnt, uint8_t* __restrict__ blob,
                                                uint32_t left_words) {
    const uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_req = 0;
    const uint64_t n_chains = cnt->lay.n_chains;
    eb_chain* __restrict__ crec = reinterpret_cast<eb_chain*>(blob + cnt->lay.off_crec);
    uint64_t* __restrict__ left = reinterpret_cast<uint64_t*>(blob + cnt->lay.off_left);
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
            for (uint32_t wi = 0; wi < left_words; ++wi) out[wi] = 0; // a context shorter than k_left is dropped by the writer (:317): blank
        }
        eb_chain k;
        k.side = (uint8_t)side;
        k.sym = (uint8_t)"ACGT"[c];
        k.reserved = 0;
        k.len = have;
        k.support = support;
        crec[ch] = k;
    }
    count_requests(cnt->req_ext, n_req);
}

// One lane per cluster: record header + right context (extract_dna, ebwt2InDel.cpp:325-342) at the first
// cluster position whose LCP >= k_right (:979-988; two BWTs :896-912); bases packed 2 bits each in text order.
// Launched behind the left-context extension, straight into the call set's image (ns_ptr: the number of survivors, in device
// memory; gate_chains: the extension's records when the clusters that cannot print are skipped).
template <bool TWO, int LANES> // LANES = 1: one lane per cluster; 4: a quad per cluster (eb_select_quad)
__global__ void __launch_bounds__(256) k_right(eb_dev_index x1, eb_dev_index x2, eb_dev_flags f, ScanParams p,
                                              const Survivor* __restrict__ surv, const uint32_t* __restrict__ chain_off,
                                              ScanCounters* __restrict__ cnt, uint8_t* __restrict__ blob, uint32_t right_words, uint64_t gate_min) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint32_t j = LANES == 4 ? threadIdx.x & 3u : 0u, qbase = (threadIdx.x & 63u) & ~3u;
    uint32_t n_req = 0, n_tab = 0; // memory requests of this cluster's FL steps (counted by its first lane): all of them / the FL-table chunks among them
    const JobLayout& L = cnt->lay;
    eb_cluster* __restrict__ recs = reinterpret_cast<eb_cluster*>(blob + sizeof(eb_calls_header));
    uint64_t* __restrict__ right = reinterpret_cast<uint64_t*>(blob + L.off_right);
    // large call sets skip the clusters that cannot print (the gate below reads the extension's records)
    const eb_chain* __restrict__ gate_chains = (uint64_t)L.ns > gate_min ? reinterpret_cast<const eb_chain*>(blob + L.off_crec) : nullptr;
    if (i < L.ns) {
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
    uint64_t* out = right + i * (uint64_t)right_words;
    if (pos != ~0ull && walk) {
        uint64_t row = pos;
        eb_dev_index x = x1; // by value, see k_extend
        if (TWO) {
            const bool second = eb_bit(f.da, pos);
            uint64_t i1 = eb_da_rank(f, pos);
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
        uint64_t word = 0;
        int c = eb_F_letter(x, row);
        while (c >= 0 && len < (uint32_t)p.k_right) {
            word |= (uint64_t)c << (2u * (len & 31u));
            ++len;
            if ((len & 31u) == 0) { if (j == 0) out[(len >> 5) - 1] = word; word = 0; }
            if (LANES == 4) row = eb_select_quad(x, row - eb_F_of(x, c), c, j, qbase, n_req);
            else row = eb_FL(x, row, c, n_req, n_tab);
            c = eb_F_letter(x, row);
        }
        if ((len & 31u) && j == 0) out[len >> 5] = word;
    }
    if (j == 0) // the words no base reached: zero (every word is written: the image is not cleared beforehand)
        for (uint32_t wi = (len + 31u) >> 5; wi < right_words; ++wi) out[wi] = 0;
    eb_cluster r;
    r.begin = s.b;
    r.end = s.e;
    r.first_chain = chain_off[i];
    r.n_slots = (uint16_t)(__popc(s.fmask0) + __popc(s.fmask1));
    r.right_found = pos != ~0ull ? 1 : 0;
    r.right_len = len;
    r.reserved = walk ? 0u : EB_RIGHT_SKIPPED;
    if (j == 0) recs[i] = r;
    }
    count_requests(cnt->req_right, j == 0 ? n_req : 0u);
    count_requests(cnt->req_tab, j == 0 ? n_tab : 0u);
}

// The job's last kernel (one wave): the context kernels' request counts and the wire image's overflow count go straight into
// the job's report in page-locked host memory.
__global__ void k_finish(const ScanCounters* __restrict__ c, JobReport* __restrict__ report) {
    unsigned long long a = c->req_ext[threadIdx.x], b = c->req_right[threadIdx.x], t = c->req_tab[threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); t += __shfl_xor(t, o); }
    if (threadIdx.x == 0) { report->req_ext = a; report->req_right = b; report->req_tab = t; report->wire_bad = c->wire_bad; report->n_hot = c->n_hot; report->n_todo = c->n_todo; }
}

}  // namespace

// ------------------------------------------------------------------ launchers (eb_scan_internal.h)

void eb_launch_extend(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const ScanParams& p, const Survivor* sorted, const uint64_t* chains,
                      ScanCounters* cnt, uint8_t* blob, uint32_t lw, uint64_t cap_chains, int extend_lanes) {
    // One lane per chain (k_extend1) since round 4: 0.98 -> 0.80 ms alone at 7.55 G positions, 1.42 -> 0.87 beside the scan
    // (profiles/r04_extend_ab.txt).  Two BWTs keep the quad form -- with a lane's index chosen at run time the one-lane form
    // loses, 1.49 against 0.96 ms at 2 x 3.8 G -- and so do large indexes: a lane's four 16-byte loads are four address
    // translations where the quad's coalesced request is one, and once the index outgrows what the translation caches reach that
    // is what counts: 0.72 / 0.82 ms (one lane / quad) at 7.55 G symbols, 1.92 / 1.82 at 15.1 G, 45 / 23 ms at 151 G.
    auto kern_extend = extend_lanes == 1 ? (p.mode == 2 ? k_extend1<true> : k_extend1<false>) : (p.mode == 2 ? k_extend<true> : k_extend<false>);
    hipLaunchKernelGGL(kern_extend, dim3(eb_grid_of(cap_chains * (uint64_t)extend_lanes)), dim3(256), 0, st, x1, x2, p, sorted, chains, cnt, blob, lw);
}

void eb_launch_right(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, const Survivor* sorted,
                     const uint32_t* choff, ScanCounters* cnt, uint8_t* blob, uint32_t rw, uint64_t cap_ns, int right_lanes, uint64_t gate_min) {
    // With the FL table (eb_psi_lookup) a step of the walk is one 16-byte read by one lane: one lane per cluster, 64 walks per
    // wave; with the directory select a quad per cluster (eb_select_quad).
    auto kern_right = p.mode == 2 ? (right_lanes == 4 ? k_right<true, 4> : k_right<true, 1>) : (right_lanes == 4 ? k_right<false, 4> : k_right<false, 1>);
    hipLaunchKernelGGL(kern_right, dim3(eb_grid_of(cap_ns * (uint64_t)right_lanes)), dim3(256), 0, st, x1, x2, f, p, sorted, choff, cnt, blob, rw, gate_min);
}

void eb_launch_finish(hipStream_t st, ScanCounters* cnt, JobReport* report) {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(EB_REQ_SLOTS), 0, st, cnt, report);
}

// loads this file's code object now (the runtime does it with the first launch otherwise: milliseconds in front of the first job)
void eb_context_preload() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_finish));
}
