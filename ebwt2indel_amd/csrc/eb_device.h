// eb_device.h -- device-side data layout and the rank primitive (gfx950 / wave64).
//
// HBM layout of the packed BWT (replaces dna_string's block, dna_string.hpp:19-41,554-585):
//   one 64-byte block per 128 symbols, made of four 16-byte chunks; chunk j holds
//     .x/.y/.z = code bit 0 / 1 / 2 of symbols 32j..32j+31 (symbol 32j+k at bit k),
//     .w      = number of symbol j (A,C,G,T for j=0..3) before this block inside its
//               2^32-symbol superblock.
//   Codes: A=0 C=1 G=2 T=3 TERM=4 (3 bits/char).  A query needs exactly one block:
//   four adjacent lanes each load one chunk (one 64-byte coalesced segment per query),
//   popcount their 32 symbols for all four letters, and combine with two lane swaps.
//   Absolute counts at superblock starts (<= 36 entries for 151 G symbols) sit in a
//   small table that stays cache resident.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EB_BLOCK_SYMS 128u
#define EB_SUPER_SHIFT 32 /* 2^32 symbols per superblock (dna_string.hpp:36) */
/* The select directory samples every 2^sel_shift-th occurrence of a letter (eb_dev_index::sel_shift, chosen when the index is
 * built: 5 up to 2^35 symbols -- n/8 bytes of directory, 1.24 block probes per FL by interpolation, ~1.03 with the ordinals of
 * eb_select_guess -- 7 beyond -- n/32 bytes, 1.52 probes). */
#define EB_SEL_SHIFT_DENSE 5
#define EB_SEL_SHIFT_SPARSE 7

struct eb_dev_index {
    const uint4* chunks;    // 4 * n_blocks chunks
    const uint64_t* super;  // 4 * n_super absolute counts (A,C,G,T)
    uint64_t n;             // symbols
    uint64_t n_blocks;      // ceil((n+1)/128)
    uint64_t F[4];          // F_A,F_C,F_G,F_T
    const uint4* xy;         // code planes x,y of every 64 positions {x lo, x hi, y lo, y hi}: the scan's 2-bit view (may be null)
    const uint32_t* sel_dir; // select directory: block of every 2^sel_shift-th occurrence of each letter
    uint64_t sel_off[4];     // first directory entry of each letter
    uint32_t term;          // ASCII terminator
    uint32_t sel_shift;     // log2 of the select directory's sampling distance
    uint32_t sel_blk_bits;  // low bits of a directory entry that hold the block (32: nothing else in the entry), see eb_select_guess
    const uint4* psi;        // FL table (eb_psi_lookup): one 16-byte chunk per 16 occurrences of a letter, or null
    uint64_t psi_off[4];     // first chunk of each letter
};

struct eb_dev_flags {
    uint64_t* thrK;   // LCP[i] >= K
    uint64_t* thrR;   // LCP[i] >= k_right
    uint64_t* minima; // local LCP minimum
    const uint64_t* flag; // thrK & ~minima, the only combination phase 4 reads (derived once, eb_flags_derive)
    const uint4* mxy;     // two-BWT mode: scan planes of the merged sequence (k_merge_planes), else null
    uint64_t* da;     // document array (may be null)
    const uint64_t* da_rank; // ones before each group of 8 words (two-BWT mode), may be null
    uint64_t n;
    uint64_t words;   // ceil(n/64)
};

// One 16-byte chunk of a block.  EB_NT_LOADS: with the nontemporal hint -- random block reads that nothing reads again soon
// (scripts/rank_probe.hip: 50.8 against 47.6 G random reads/s on a 4 GB table).
#ifndef EB_NT_LOADS
#define EB_NT_LOADS 0
#endif
typedef uint32_t eb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 eb_load_chunk_nt(const uint4* p) {
    eb_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const eb_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 eb_load_chunk(const uint4* p) {
#if EB_NT_LOADS
    return eb_load_chunk_nt(p);
#else
    return *p;
#endif
}
// the probes of a select (the right-context walk): blocks nobody looks at twice
#ifndef EB_NT_SELECT
#define EB_NT_SELECT 1
#endif
__device__ __forceinline__ uint4 eb_load_chunk_select(const uint4* p) {
#if EB_NT_SELECT
    return eb_load_chunk_nt(p);
#else
    return eb_load_chunk(p);
#endif
}

// A pointer that was LOADED from memory (launch arguments kept in device memory) is a generic pointer to the compiler, and
// accesses through it become flat_load / flat_store, which count in lgkmcnt as well as vmcnt: every wait for an LDS read
// then also waits for them.  These move whole 16-byte-multiple records through explicit global-memory (address space 1)
// accesses instead.
#define EB_GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T eb_load_global(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "records are moved as 16-byte pieces");
    T r;
    const EB_GLOBAL_AS eb_u32x4* g = (const EB_GLOBAL_AS eb_u32x4*)p;
    eb_u32x4* d = reinterpret_cast<eb_u32x4*>(&r);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); ++k) d[k] = g[k];
    return r;
}
__device__ __forceinline__ void eb_store_global(uint4* p, uint4 v) {
    eb_u32x4 t = {v.x, v.y, v.z, v.w};
    *(EB_GLOBAL_AS eb_u32x4*)p = t;
}

struct rank4_t {
    uint64_t v[4];
};

__device__ __forceinline__ uint32_t eb_prefix_mask32(uint32_t o, uint32_t j) {
    // mask of the symbols of chunk j that lie before in-block offset o (0..127)
    // the low min(max(o - 32 j, 0), 32) bits: a clamp, a 64-bit shift (32 must give all ones) and a decrement of its low half
    int t = (int)o - (int)(32u * j);
    t = t < 0 ? 0 : (t > 32 ? 32 : t);
    return (uint32_t)(1ull << t) - 1u;
}

// Packed per-letter popcounts (A | C<<8 | G<<16 | T<<24) of the symbols selected by `mask`.
__device__ __forceinline__ uint32_t eb_chunk_counts(uint4 c, uint32_t mask) {
    uint32_t d = ~c.z & mask;
    uint32_t lo = d & ~c.y, hi = d & c.y;
    return (uint32_t)__popc(lo & ~c.x) | ((uint32_t)__popc(lo & c.x) << 8) | ((uint32_t)__popc(hi & ~c.x) << 16) |
           ((uint32_t)__popc(hi & c.x) << 24);
}

// parallel_rank (dna_string.hpp:140-152) by ONE lane: loads the whole 64-byte block.
__device__ __forceinline__ rank4_t eb_rank4(const eb_dev_index& x, uint64_t i) {
    const uint4* b = x.chunks + ((i >> 7) << 2);
    const uint64_t* s = x.super + ((i >> EB_SUPER_SHIFT) << 2);
    uint32_t o = (uint32_t)(i & 127u);
    uint4 c0 = b[0], c1 = b[1], c2 = b[2], c3 = b[3];
    uint32_t p = eb_chunk_counts(c0, eb_prefix_mask32(o, 0)) + eb_chunk_counts(c1, eb_prefix_mask32(o, 1)) +
                 eb_chunk_counts(c2, eb_prefix_mask32(o, 2)) + eb_chunk_counts(c3, eb_prefix_mask32(o, 3));
    rank4_t r;
    r.v[0] = s[0] + c0.w + (p & 0xFFu);
    r.v[1] = s[1] + c1.w + ((p >> 8) & 0xFFu);
    r.v[2] = s[2] + c2.w + ((p >> 16) & 0xFFu);
    r.v[3] = s[3] + c3.w + (p >> 24);
    return r;
}

// ---- quad-local data movement with DPP (quad_perm) instead of __shfl / __shfl_xor: those compile to ds_bpermute_b32 -- an
// LDS round trip of ~100 cycles on the path of every dependent step -- whereas a quad permutation is a modifier of the vector
// instruction that consumes the value (v_add_u32_dpp, v_max_u32_dpp, v_or_b32_dpp).  All four lanes of a quad must be active.
#define EB_QUAD_XOR1 0xB1 /* quad_perm:[1,0,3,2] */
#define EB_QUAD_XOR2 0x4E /* quad_perm:[2,3,0,1] */
template <int CTRL>
__device__ __forceinline__ uint32_t eb_quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t eb_quad_sum(uint32_t v) {
    v += eb_quad_perm<EB_QUAD_XOR1>(v);
    v += eb_quad_perm<EB_QUAD_XOR2>(v);
    return v;
}
__device__ __forceinline__ uint32_t eb_quad_max(uint32_t v) {
    uint32_t o = eb_quad_perm<EB_QUAD_XOR1>(v);
    v = o > v ? o : v;
    o = eb_quad_perm<EB_QUAD_XOR2>(v);
    return o > v ? o : v;
}
__device__ __forceinline__ uint64_t eb_quad_max64(uint64_t v) {
    uint64_t o = ((uint64_t)eb_quad_perm<EB_QUAD_XOR1>((uint32_t)(v >> 32)) << 32) | eb_quad_perm<EB_QUAD_XOR1>((uint32_t)v);
    v = o > v ? o : v;
    o = ((uint64_t)eb_quad_perm<EB_QUAD_XOR2>((uint32_t)(v >> 32)) << 32) | eb_quad_perm<EB_QUAD_XOR2>((uint32_t)v);
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t eb_quad_or(uint32_t v) {
    v |= eb_quad_perm<EB_QUAD_XOR1>(v);
    v |= eb_quad_perm<EB_QUAD_XOR2>(v);
    return v;
}
// the value of the ONE lane of the quad whose `mine` is set, in all four
__device__ __forceinline__ uint64_t eb_quad_pick64(uint64_t v, bool mine) {
    const uint32_t lo = eb_quad_or(mine ? (uint32_t)v : 0u), hi = eb_quad_or(mine ? (uint32_t)(v >> 32) : 0u);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t eb_quad_pick32(uint32_t v, bool mine) { return eb_quad_or(mine ? v : 0u); }
// the value of lane (byte_addr >> 2) of the wave, through the LDS crossbar (no vector-ALU work; the lane read from must be active)
__device__ __forceinline__ uint64_t eb_bpermute64(int byte_addr, uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Letter j's count before the superblock of position i, with the last answer kept: consecutive queries of a walk nearly always
// stay in one superblock (2^32 symbols), and the table read is a global load on the path of every step otherwise.
struct eb_super_cache {
    uint32_t hi;   // superblock of `val` (0xFFFFFFFF: none yet)
    uint64_t val;
};
__device__ __forceinline__ uint64_t eb_super_cached(const eb_dev_index& x, eb_super_cache& sc, uint64_t i, uint32_t j) {
    const uint32_t hi = (uint32_t)(i >> EB_SUPER_SHIFT);
    if (hi != sc.hi) {
        sc.hi = hi;
        sc.val = x.super[((uint64_t)hi << 2) + j];
    }
    return sc.val;
}

// rank of letter j at both ends of [lo, hi] by one quad, as eb_rank4_quad_pair, with the quad sums through DPP and the
// superblock counts through the cache; sh8 = 8 * j.
__device__ __forceinline__ void eb_rank4_quad_pair_dpp(const eb_dev_index& x, eb_super_cache& sc, uint64_t lo, uint64_t hi, uint32_t j, uint32_t sh8,
                                                       uint64_t& a, uint64_t& b) {
    uint4 c = eb_load_chunk(x.chunks + ((lo >> 7) << 2) + j);
    uint4 d = c;
    if ((hi >> 7) != (lo >> 7)) d = eb_load_chunk(x.chunks + ((hi >> 7) << 2) + j); // uniform inside the quad
    const uint32_t p = eb_quad_sum(eb_chunk_counts(c, eb_prefix_mask32((uint32_t)(lo & 127u), j)));
    const uint32_t q = eb_quad_sum(eb_chunk_counts(d, eb_prefix_mask32((uint32_t)(hi & 127u), j)));
    a = eb_super_cached(x, sc, lo, j) + ((uint64_t)c.w + ((p >> sh8) & 0xFFu));
    b = eb_super_cached(x, sc, hi, j) + ((uint64_t)d.w + ((q >> sh8) & 0xFFu));
}

// parallel_rank by FOUR adjacent lanes (lane&3 = j): each loads chunk j; returns the rank of letter j.
// All four lanes of the quad must be active and pass the same i.
__device__ __forceinline__ uint64_t eb_rank4_quad(const eb_dev_index& x, uint64_t i, uint32_t j) {
    uint4 c = eb_load_chunk(x.chunks + ((i >> 7) << 2) + j);
    uint32_t p = eb_chunk_counts(c, eb_prefix_mask32((uint32_t)(i & 127u), j));
    p = eb_quad_sum(p);
    return x.super[((i >> EB_SUPER_SHIFT) << 2) + j] + c.w + ((p >> (8u * j)) & 0xFFu);
}

// The same in two steps, so that a kernel can put the loads of many queries in flight before it needs the first result.
__device__ __forceinline__ uint4 eb_rank4_quad_load(const eb_dev_index& x, uint64_t i, uint32_t j) { return eb_load_chunk(x.chunks + ((i >> 7) << 2) + j); }
__device__ __forceinline__ uint64_t eb_rank4_quad_finish(const eb_dev_index& x, uint4 c, uint64_t i, uint32_t j) {
    uint32_t p = eb_chunk_counts(c, eb_prefix_mask32((uint32_t)(i & 127u), j));
    p = eb_quad_sum(p);
    return x.super[((i >> EB_SUPER_SHIFT) << 2) + j] + c.w + ((p >> (8u * j)) & 0xFFu);
}

// Ranks at both ends of an interval [lo, hi] by one quad: a narrow interval (the usual case when following a
// cluster's left context, width ~ coverage) lies in ONE block, which is then loaded once.
__device__ __forceinline__ void eb_rank4_quad_pair(const eb_dev_index& x, uint64_t lo, uint64_t hi, uint32_t j, uint64_t& a, uint64_t& b) {
    uint4 c = eb_load_chunk(x.chunks + ((lo >> 7) << 2) + j);
    uint4 d = c;
    if ((hi >> 7) != (lo >> 7)) d = eb_load_chunk(x.chunks + ((hi >> 7) << 2) + j); // uniform inside the quad
    uint32_t p = eb_chunk_counts(c, eb_prefix_mask32((uint32_t)(lo & 127u), j));
    uint32_t q = eb_chunk_counts(d, eb_prefix_mask32((uint32_t)(hi & 127u), j));
    p = eb_quad_sum(p);
    q = eb_quad_sum(q);
    a = x.super[((lo >> EB_SUPER_SHIFT) << 2) + j] + c.w + ((p >> (8u * j)) & 0xFFu);
    b = x.super[((hi >> EB_SUPER_SHIFT) << 2) + j] + d.w + ((q >> (8u * j)) & 0xFFu);
}

// operator[] (dna_string.hpp:113-135): 3-bit code of symbol i.
__device__ __forceinline__ uint32_t eb_code(const eb_dev_index& x, uint64_t i) {
    uint4 c = x.chunks[i >> 5];
    uint32_t k = (uint32_t)(i & 31u);
    return ((c.x >> k) & 1u) | (((c.y >> k) & 1u) << 1) | (((c.z >> k) & 1u) << 2);
}

__device__ __forceinline__ uint32_t eb_code_to_ascii(uint32_t code, uint32_t term) {
    return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : code == 3 ? 'T' : term;
}

// a[c] for a run-time letter c of a four-entry table held in a by-value index struct, as four independent selects OR-ed together:
// the chain  c == 0 ? a[0] : c == 1 ? a[1] : ...  is turned back into a dynamically indexed load by the compiler, and a struct
// that is indexed dynamically lives in scratch memory (k_right<two BWTs, one lane>: 176 bytes per lane, round 4)
__device__ __forceinline__ uint64_t eb_pick_letter(const uint64_t (&a)[4], int c) {
    return (c == 0 ? a[0] : 0ull) | (c == 1 ? a[1] : 0ull) | (c == 2 ? a[2] : 0ull) | (c == 3 ? a[3] : 0ull);
}

// F(i) (dna_bwt.hpp:100-110) as a letter index: -1 for a TERM row.
__device__ __forceinline__ int eb_F_letter(const eb_dev_index& x, uint64_t i) {
    return i < x.F[0] ? -1 : i < x.F[1] ? 0 : i < x.F[2] ? 1 : i < x.F[3] ? 2 : 3;
}

// Absolute count of letter c before block b.
__device__ __forceinline__ uint64_t eb_block_count(const eb_dev_index& x, uint64_t b, int c) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(x.chunks + (b << 2) + c);
    return x.super[((b >> 25) << 2) + c] + w[3];
}

// A directory entry holds the block of a sampled occurrence in its low sel_blk_bits bits and, in the bits an index of this size
// leaves free, how many occurrences of the letter precede the sample INSIDE that block (0..127, in units of 2^os so that it
// fits: os = 7 - (32 - sel_blk_bits)).  With that the block of the r-th occurrence is known before any block is read, in the
// usual case: if the next sample (2^sel_shift occurrences on) sits in block hi with o' occurrences of its block before it,
// the last o' occurrences before that sample lie in hi, all earlier ones before hi -- so with hi = lo + 1 (consecutive samples
// are about a block apart at the dense sampling) the answer is exact up to the rounding of o'.  Round 3's directory held the
// blocks only and guessed by interpolation: right three times out of four (1.24 probes per select); every wrong guess costs a
// wave of 16 walks a second pass through the probe body and a second random line.  Returns a block in [lo, hi].
// (Written without early exits, as one 32-bit select: the first form -- `return hi;` from inside nested divergent ifs of a function
// returning uint64_t -- came out of hipcc 7.2 with the high half of the result left holding a temporary in the lanes that took
// that exit, i.e. a block index beyond 2^32 and a memory fault; found in the disassembly of k_right, round 3.)
__device__ __forceinline__ uint32_t eb_select_guess(const eb_dev_index& x, const uint32_t* __restrict__ dir, uint64_t r) {
    const uint32_t sh = x.sel_shift, bb = x.sel_blk_bits; // (wave-uniform)
    const uint64_t m = r >> sh;
    const uint32_t e0 = dir[m], e1 = dir[m + 1]; // (same line nearly always)
    const uint32_t frac = (uint32_t)r & ((1u << sh) - 1u);
    const bool ord = bb < 32u; // entries carry ordinals (up to 2^29 blocks)
    uint32_t lo = e0, hi = e1, o0 = 0, o1 = 0;
    if (ord) {
        const uint32_t bmask = (1u << bb) - 1u;
        lo = e0 & bmask;
        hi = e1 & bmask;
        o0 = (e0 >> bb) << (bb - 25u); // occurrences in block lo before this sample, rounded down to the stored unit
        o1 = (e1 >> bb) << (bb - 25u); // occurrences in block hi before the next sample
    }
    const uint32_t span = hi - lo;
    // with ordinals: the last o' occurrences before the next sample lie in hi, every earlier one before hi (samples beyond the
    // letter's last occurrence point at the last block with o' = 0)
    const bool at_hi = ord && (span == 0u || frac + o1 >= (1u << sh));
    // otherwise spread the guess over the blocks that can hold the answer: lo .. hi - 1 with ordinals, lo .. hi without
    const uint32_t sp0 = ord ? span : (span < 0xFFFFFFFFu ? span + 1u : span);
    const uint32_t sp = sp0 < 0xFFFFFFu ? sp0 : 0xFFFFFFu; // 24 bits x sh <= 7 bits: the product fits (a wider sh only blurs the guess)
    uint32_t gi = (frac * sp) >> sh;                         // < sp, were it not for a product that overflows with sh > 8
    gi = gi < sp ? gi : sp - 1u;
    // two blocks before hi (one select in twelve at the dense sampling): they share the 2^sh - o' + o occurrences from the start of
    // lo to the next sample's block about evenly, and the answer is number o + frac of those -- in lo + 1 iff past the middle
    // (CPU model of the whole rule on read-like letter frequencies: 98 % of first probes right, 95.5 % without this line)
    if (ord && span == 2u) gi = 2u * frac + o0 + o1 >= (1u << sh) ? 1u : 0u;
    return at_hi ? hi : lo + gi;
}

// select (dna_string.hpp:182-188,254-272): position of the r-th (0-based) letter c.  The reference
// bisects [0,n) with ~log2(n) full rank queries (33 at n = 7.55 G).  Here a directory (block of every 2^sel_shift-th
// occurrence of each letter: every 32nd up to 2^35 symbols, n/8 bytes in all; every 128th beyond) brackets the answer to a
// block or two, and eb_select_guess names the right one nearly always.  Each probe loads the WHOLE 64-byte block (one line): its
// counter says how many c precede it, its planes how many it holds, so the block that contains the answer is
// recognised without probing its neighbour, and the in-block select needs no further memory access.
// Same unique answer as the reference's bisection in ~2 dependent reads (directory line, block).
// `probes` counts the memory requests issued: 1 for the directory pair + 1 per 64-byte block probed (bench.py's
// context_roofline is built from these counts).
__device__ __forceinline__ uint64_t eb_select(const eb_dev_index& x, uint64_t r, int c, uint32_t& probes) {
    const uint32_t* dir = x.sel_dir + eb_pick_letter(x.sel_off, c);
    uint64_t g = (uint64_t)eb_select_guess(x, dir, r); // the answer lies in a block between the two samples around r
    if (g >= x.n_blocks) g = x.n_blocks - 1; // (never, by construction of the directory: a wrong guess may cost probes, it must not cost a fault)
    ++probes;
    for (;;) {
        ++probes;
        const uint4* b = x.chunks + (g << 2);
        uint4 ch0 = b[0], ch1 = b[1], ch2 = b[2], ch3 = b[3];
        uint32_t w = c == 0 ? ch0.w : c == 1 ? ch1.w : c == 2 ? ch2.w : ch3.w;
        uint64_t before = x.super[((g >> 25) << 2) + c] + w; // occurrences of c before block g
        uint32_t m0 = ~ch0.z & ((c & 2) ? ch0.y : ~ch0.y) & ((c & 1) ? ch0.x : ~ch0.x);
        uint32_t m1 = ~ch1.z & ((c & 2) ? ch1.y : ~ch1.y) & ((c & 1) ? ch1.x : ~ch1.x);
        uint32_t m2 = ~ch2.z & ((c & 2) ? ch2.y : ~ch2.y) & ((c & 1) ? ch2.x : ~ch2.x);
        uint32_t m3 = ~ch3.z & ((c & 2) ? ch3.y : ~ch3.y) & ((c & 1) ? ch3.x : ~ch3.x);
        uint32_t k0 = (uint32_t)__popc(m0), k1 = (uint32_t)__popc(m1), k2 = (uint32_t)__popc(m2), k3 = (uint32_t)__popc(m3);
        uint32_t k = k0 + k1 + k2 + k3;
        if (r < before) {                      // answer lies earlier: at least ceil((before - r) / 128) blocks back
            g -= ((before - r - 1) >> 7) + 1;
            continue;
        }
        if (r >= before + k) {                 // answer lies later: a block holds at most 128 occurrences
            g += ((r - before - k) >> 7) + 1;
            continue;
        }
        uint32_t need = (uint32_t)(r - before); // occurrences to skip inside block g
        uint32_t mk, base;
        if (need < k0) { mk = m0; base = 0; }
        else if (need < k0 + k1) { mk = m1; base = 32; need -= k0; }
        else if (need < k0 + k1 + k2) { mk = m2; base = 64; need -= k0 + k1; }
        else { mk = m3; base = 96; need -= k0 + k1 + k2; }
        for (uint32_t t = 0; t < need; ++t) mk &= mk - 1; // drop the `need` lowest set bits
        return (g << 7) + base + (uint32_t)__ffs((int)mk) - 1u;
    }
}

__device__ __forceinline__ uint64_t eb_select(const eb_dev_index& x, uint64_t r, int c) {
    uint32_t probes = 0;
    return eb_select(x, r, c, probes);
}

// The same select by FOUR adjacent lanes (lane & 3 = j, qbase = first lane of the quad inside the wave; all four pass the
// same r and c): each probe is one coalesced 64-byte request -- lane j loads chunk j -- instead of four 16-byte loads by one
// lane, and a wave holds 16 walks instead of 64, so fewer lanes wait for the walk that needs a third probe.
__device__ __forceinline__ uint64_t eb_select_quad(const eb_dev_index& x, uint64_t r, int c, uint32_t j, uint32_t qbase, uint32_t& probes) {
    const uint32_t* dir = x.sel_dir + eb_pick_letter(x.sel_off, c);
    uint64_t g = (uint64_t)eb_select_guess(x, dir, r); // same addresses in all four lanes: one request
    if (g >= x.n_blocks) g = x.n_blocks - 1; // (never, by construction of the directory: a wrong guess may cost probes, it must not cost a fault)
    ++probes;
    for (;;) {
        ++probes;
        const uint4 ch = eb_load_chunk_select(x.chunks + (g << 2) + j);
        const uint32_t mk = ~ch.z & ((c & 2) ? ch.y : ~ch.y) & ((c & 1) ? ch.x : ~ch.x);
        const uint32_t kj = (uint32_t)__popc(mk);
        const uint32_t k0 = (uint32_t)__shfl((int)kj, qbase), k1 = (uint32_t)__shfl((int)kj, qbase + 1), k2 = (uint32_t)__shfl((int)kj, qbase + 2),
                       k3 = (uint32_t)__shfl((int)kj, qbase + 3);
        const uint32_t w = (uint32_t)__shfl((int)ch.w, qbase + c); // chunk c carries letter c's counter
        const uint64_t before = x.super[((g >> 25) << 2) + c] + w;
        const uint32_t k = k0 + k1 + k2 + k3;
        if (r < before) { g -= ((before - r - 1) >> 7) + 1; continue; }   // uniform inside the quad
        if (r >= before + k) { g += ((r - before - k) >> 7) + 1; continue; }
        uint32_t need = (uint32_t)(r - before);
        uint32_t t;
        if (need < k0) t = 0;
        else if (need < k0 + k1) { t = 1; need -= k0; }
        else if (need < k0 + k1 + k2) { t = 2; need -= k0 + k1; }
        else { t = 3; need -= k0 + k1 + k2; }
        uint32_t mt = (uint32_t)__shfl((int)mk, qbase + t);
        for (uint32_t q = 0; q < need; ++q) mt &= mt - 1; // drop the `need` lowest set bits
        return (g << 7) + 32u * t + (uint32_t)__ffs((int)mt) - 1u;
    }
}

// F[letter] by selects: a dynamic index into a kernel-argument array would send the whole struct to scratch memory
__device__ __forceinline__ uint64_t eb_F_of(const eb_dev_index& x, int letter) {
    return eb_pick_letter(x.F, letter);
}

// ---- FL table ("psi", round 4).  An index that is scanned again and again (eb_index_build_psi; phase 4 builds it by itself with
// the third job on an index) keeps, for every letter c and every 16 consecutive occurrences of c, ONE 16-byte chunk that spells
// their positions out, so that select(r, c) is one 16-byte load by one lane and a few dozen integer instructions -- no directory,
// no block, no probe loop: a right-context walk (k_right x FL, dna_bwt.hpp:115-133) moves ONE line per step instead of 2.1.
// In the BWT of a read collection a letter comes in runs (~coverage rows long): 93 % of the occurrences of a letter sit right
// behind the previous one, the others a jump of some tens or hundreds of rows on.  Chunk = 128 bits:
//   bits   0..37   position of the chunk's first occurrence
//   bits  38..52   bit k-1 set iff occurrence k (1..15) does NOT sit right behind occurrence k-1 (a "jump")
//   bits  53..127  the jumps' lengths (rows skipped), in order, W bits each: W = min(15, 75 / jumps) -- few long jumps or many
//                  short ones (a heterozygous column alternates letters), eb_psi_width
// position(k) = first + k + the first popcount(bitmap & low k bits) jump lengths.  A chunk whose jumps do not fit (0.1 % on
// 30x read data) is VOID -- all ones; no position of an index this table is built for, n < 2^38 - 1, looks like that -- and the
// directory select answers.  n bytes in all (the index itself is n/2).
#define EB_PSI_SHIFT 4u
#define EB_PSI_MAX_N ((1ull << 38) - 2ull)
#define EB_PSI_MAX_JUMP 32767u
__device__ __forceinline__ uint4 eb_psi_void() { return make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); }
// width of a jump field in a chunk with nj (1..15) jumps: min(15, 75 / nj), as a table of nibbles
__host__ __device__ __forceinline__ uint32_t eb_psi_width(uint32_t nj) { return (uint32_t)((0x55566789ACFFFFF0ull >> (4u * nj)) & 15u); }
// W bits at bit offset `off` (38 <= off, off + W <= 128) of the chunk
__device__ __forceinline__ uint32_t eb_psi_bits(uint4 v, uint32_t off, uint32_t W) {
    const uint32_t idx = off >> 5, bit = off & 31u;
    const uint32_t lo = idx == 1u ? v.y : idx == 2u ? v.z : v.w, hi = idx == 1u ? v.z : idx == 2u ? v.w : 0u;
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> bit) & ((1u << W) - 1u);
}
// position of the r-th (0-based) letter c, from its chunk; false: the chunk is void
__device__ __forceinline__ bool eb_psi_lookup(const eb_dev_index& x, uint64_t r, int c, uint64_t& pos) {
    const uint4 v = eb_load_chunk_select(x.psi + eb_pick_letter(x.psi_off, c) + (r >> EB_PSI_SHIFT));
    if (v.x == 0xFFFFFFFFu && (v.y & 63u) == 63u) return false;
    const uint32_t k = (uint32_t)r & 15u;
    const uint32_t bitmap = (v.y >> 6) & 0x7FFFu;
    const uint32_t W = eb_psi_width((uint32_t)__popc(bitmap));
    const uint32_t j = (uint32_t)__popc(bitmap & ((1u << k) - 1u)); // jumps on the way from the first occurrence to this one
    uint32_t skipped = 0;
    for (uint32_t t = 0; t < j; ++t) skipped += eb_psi_bits(v, 53u + t * W, W);
    pos = (((uint64_t)(v.y & 63u) << 32) | v.x) + k + skipped;
    return true;
}

// FL(i) (dna_bwt.hpp:115-133); i must not be a TERM row.
__device__ __forceinline__ uint64_t eb_FL(const eb_dev_index& x, uint64_t i, int letter) {
    const uint64_t r = i - eb_F_of(x, letter);
    uint64_t pos;
    if (x.psi && eb_psi_lookup(x, r, letter, pos)) return pos;
    return eb_select(x, r, letter);
}
// `probes`: memory requests issued (one per table chunk; directory read + blocks probed otherwise); `table_reads`: the table
// chunks among them (16-byte requests; a void chunk costs its read AND the select's)
__device__ __forceinline__ uint64_t eb_FL(const eb_dev_index& x, uint64_t i, int letter, uint32_t& probes, uint32_t& table_reads) {
    const uint64_t r = i - eb_F_of(x, letter);
    uint64_t pos;
    if (x.psi) {
        ++probes;
        ++table_reads;
        if (eb_psi_lookup(x, r, letter, pos)) return pos;
    }
    return eb_select(x, r, letter, probes);
}

__device__ __forceinline__ bool eb_bit(const uint64_t* b, uint64_t i) { return (b[i >> 6] >> (i & 63u)) & 1ull; }
// ones in DA[0,p): row p of the merge of two BWTs is row eb_da_rank(p) of the second or p - eb_da_rank(p) of the first
// (ebwt2InDel.cpp:1431-1432); da_rank[] holds the ones before every 512 positions
__device__ __forceinline__ uint64_t eb_da_rank(const eb_dev_flags& f, uint64_t p) {
    uint64_t g = p >> 9, w = p >> 6;
    uint64_t s = f.da_rank[g];
    for (uint64_t k = g << 3; k < w; ++k) s += (uint64_t)__popcll(f.da[k]);
    uint32_t o = (uint32_t)(p & 63u);
    if (o) s += (uint64_t)__popcll(f.da[w] & ((1ull << o) - 1ull));
    return s;
}
__device__ __forceinline__ void eb_bit_set_atomic(uint64_t* b, uint64_t i) {
    atomicOr(reinterpret_cast<unsigned long long*>(b + (i >> 6)), 1ull << (i & 63u));
}
// set bits [lo,hi) (hi > lo) with one atomic per touched word
__device__ __forceinline__ void eb_bits_fill_atomic(uint64_t* b, uint64_t lo, uint64_t hi) {
    uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
    for (uint64_t w = w0; w <= w1; ++w) {
        uint64_t m = ~0ull;
        if (w == w0) m &= ~0ull << (lo & 63u);
        if (w == w1) m &= ~0ull >> (63u - ((hi - 1) & 63u));
        atomicOr(reinterpret_cast<unsigned long long*>(b + w), m);
    }
}

// Output records (suffix-tree children, surviving clusters) are staged in LDS and appended to their list in
// HBM with ONE global atomic per few hundred records: a returning atomic on a single word serialises at
// ~11 ns (MI355X_MICROARCH.md, 'dequeue'), which per wave or per record would dominate these kernels.
template <typename Item, int STAGE>
struct Stager {
    static constexpr int W = sizeof(Item) / 16; // uint4 per record
    uint4* buf;                                 // STAGE * W uint4 in LDS
    uint32_t* s_count;
    unsigned long long* s_base;

    // every thread of the block calls this once per tile; `keep` lanes hand over one record
    __device__ __forceinline__ void push(bool keep, const Item& rec) {
        unsigned long long mask = __ballot(keep);
        uint32_t lane = threadIdx.x & 63u;
        uint32_t wcount = (uint32_t)__popcll(mask);
        uint32_t wbase = 0;
        if (lane == 0 && wcount) wbase = atomicAdd(s_count, wcount);
        wbase = (uint32_t)__shfl((int)wbase, 0);
        if (keep) {
            uint32_t slot = wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            const uint4* src = reinterpret_cast<const uint4*>(&rec);
            for (int k = 0; k < W; ++k) buf[slot * W + k] = src[k];
        }
    }
    // block-uniform: append the staged records to out[] and reset
    __device__ __forceinline__ void flush(Item* out, uint64_t out_cap, unsigned long long* pushed) {
        __syncthreads();
        uint32_t n = *s_count;
        if (n == 0) return; // uniform
        if (threadIdx.x == 0) *s_base = atomicAdd(pushed, (unsigned long long)n);
        __syncthreads();
        unsigned long long base = *s_base;
        uint4* dst = reinterpret_cast<uint4*>(out);
        if (base + n <= out_cap)
            for (uint32_t k = threadIdx.x; k < n * W; k += blockDim.x) eb_store_global(dst + base * W + k, buf[k]); // `out` is global memory (it may come from a pointer the kernel loaded)
        __syncthreads();
        if (threadIdx.x == 0) *s_count = 0;
        __syncthreads();
    }
};

