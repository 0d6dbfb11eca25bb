// eb_index.hip -- phase 1 on the GPU: ASCII eBWT -> packed 3-bit planes + rank counters + F array.
// Replaces dna_string(path,TERM) (dna_string.hpp:55-110: read/validate/pack; :275-315: counters)
// and the F-array loop of dna_bwt(path,TERM) (dna_bwt.hpp:36-62).
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "eb_config.h"
#include "eb_internal.h"
#include "eb_pieces.h"

// ------------------------------------------------------------------ errors / params

static thread_local char g_err[512] = "";
static thread_local uint8_t g_bad_sym = 0;
static thread_local uint64_t g_bad_pos = 0;

void eb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* eb_last_error(void) { return g_err; }

extern "C" int eb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void eb_params_init(eb_params* p) {
    memset(p, 0, sizeof *p);
    p->term = '#';
    eb_params_resolve(p);
}

// 0 means "use the default" (ebwt2InDel.cpp:1740-1746); -q and -t are taken as given.
extern "C" void eb_params_resolve(eb_params* p) {
    if (p->complexity == 0) p->complexity = 20; // complexity_def is frozen at 30-10 (ebwt2InDel.cpp:64)
    if (p->K == 0) p->K = 16;
    if (p->max_gap == 0) p->max_gap = 10;
    if (p->k_left == 0) p->k_left = 31;
    if (p->k_right == 0) p->k_right = 30;
    if (p->max_snvs == 0) p->max_snvs = 2;
    if (p->mcov_out == 0) p->mcov_out = 3;
}

extern "C" void eb_index_bad_symbol(uint8_t* symbol, uint64_t* position) {
    if (symbol) *symbol = g_bad_sym;
    if (position) *position = g_bad_pos;
}

// ------------------------------------------------------------------ scratch cache

#include <mutex>
#include <thread>
#include <vector>

namespace {
struct ScratchBlock { int device; void* p; size_t cap; };
std::mutex g_scratch_mu;
std::vector<ScratchBlock> g_scratch;
}  // namespace

void* eb_scratch_get(int device, size_t bytes, size_t* capacity) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        size_t best = g_scratch.size();
        for (size_t i = 0; i < g_scratch.size(); ++i)
            if (g_scratch[i].device == device && g_scratch[i].cap >= bytes && g_scratch[i].cap <= 2 * bytes + (1u << 20) &&
                (best == g_scratch.size() || g_scratch[i].cap < g_scratch[best].cap))
                best = i;
        if (best != g_scratch.size()) {
            void* p = g_scratch[best].p;
            *capacity = g_scratch[best].cap;
            g_scratch.erase(g_scratch.begin() + best);
            return p;
        }
    }
    size_t slack = bytes / 8 < (64u << 20) ? bytes / 8 : (64u << 20); // room to be reused by a slightly larger request
    size_t cap = bytes + slack + 256;
    void* p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) {
        eb_scratch_release(device); // give cached blocks back and retry once
        if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    }
    if (eb_cfg().trace_alloc) fprintf(stderr, "[scratch] hipMalloc of %.1f MB (no cached block fits)\n", cap / 1e6);
    *capacity = cap;
    return p;
}

void eb_scratch_put(int device, void* p, size_t capacity) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch.size() < 512) { g_scratch.push_back({device, p, capacity}); return; }
    (void)hipFree(p);
}

void eb_scratch_release(int device) {
    std::vector<ScratchBlock> mine;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        for (size_t i = 0; i < g_scratch.size();)
            if (g_scratch[i].device == device) { mine.push_back(g_scratch[i]); g_scratch.erase(g_scratch.begin() + i); } else ++i;
    }
    for (auto& b : mine) (void)hipFree(b.p);
}

// ------------------------------------------------------------------ kernels

// One thread packs 32 symbols into one chunk (planes only; counter filled later).
// Positions >= n_total are padding: code bit 2 set, so they never count as A,C,G,T.
__global__ void __launch_bounds__(256) k_pack(const uint8_t* __restrict__ ascii, uint64_t piece_pos, uint64_t piece_len,
                                              uint64_t first_chunk, uint64_t n_chunks, uint32_t term, int aligned16,
                                              uint4* __restrict__ chunks, unsigned long long* __restrict__ bad_pos) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    uint64_t off = t * 32; // offset inside the piece
    uint8_t s[32];
    if (off + 32 <= piece_len && aligned16) {
        const uint4* p = reinterpret_cast<const uint4*>(ascii + off);
        *reinterpret_cast<uint4*>(s) = p[0];
        *reinterpret_cast<uint4*>(s + 16) = p[1];
    } else {
        for (int k = 0; k < 32; ++k) s[k] = off + k < piece_len ? ascii[off + k] : 0;
    }
    uint32_t p0 = 0, p1 = 0, p2 = 0;
    for (int k = 0; k < 32; ++k) {
        uint32_t code;
        if (off + k >= piece_len) {
            code = 4; // padding behind the last symbol
        } else {
            uint32_t c = s[k];
            code = c == term ? 4u : c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 8u;
            if (code == 8u) {
                atomicMin(bad_pos, (unsigned long long)(piece_pos + off + k));
                code = 4;
            }
        }
        p0 |= (code & 1u) << k;
        p1 |= ((code >> 1) & 1u) << k;
        p2 |= ((code >> 2) & 1u) << k;
    }
    chunks[first_chunk + t] = make_uint4(p0, p1, p2, 0u);
}

__device__ __forceinline__ uint32_t block_letter_counts(const uint4* b) {
    return eb_chunk_counts(b[0], ~0u) + eb_chunk_counts(b[1], ~0u) + eb_chunk_counts(b[2], ~0u) + eb_chunk_counts(b[3], ~0u);
}

#define TILE_BLOCKS 256 /* BWT blocks per workgroup tile; divides 2^25 so superblocks start on tile borders */

// Per tile: sum of the four letter counts (packed bytes can reach 128 per block -> widen before summing).
__global__ void __launch_bounds__(TILE_BLOCKS) k_tile_sums(const uint4* __restrict__ chunks, uint64_t n_blocks,
                                                           uint64_t* __restrict__ tile_sums /* 4 per tile */) {
    __shared__ uint32_t sh[4][TILE_BLOCKS / 64];
    uint64_t b = (uint64_t)blockIdx.x * TILE_BLOCKS + threadIdx.x;
    uint32_t p = b < n_blocks ? block_letter_counts(chunks + (b << 2)) : 0u;
    uint32_t v[4] = {p & 0xFFu, (p >> 8) & 0xFFu, (p >> 16) & 0xFFu, p >> 24};
    for (int c = 0; c < 4; ++c) {
        uint32_t x = v[c];
        for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_down((int)x, o);
        if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t s = 0;
        for (int w = 0; w < TILE_BLOCKS / 64; ++w) s += sh[threadIdx.x][w];
        tile_sums[(uint64_t)blockIdx.x * 4 + threadIdx.x] = s;
    }
}

// Per tile: exclusive scan of block counts + tile prefix -> counters relative to the superblock start
// (set_counters, dna_string.hpp:554-567); records superblock_ranks (dna_string.hpp:285-289).
__global__ void __launch_bounds__(TILE_BLOCKS) k_write_counters(uint4* __restrict__ chunks, uint64_t n_blocks,
                                                                const uint64_t* __restrict__ tile_prefix,
                                                                uint64_t* __restrict__ super) {
    __shared__ uint32_t sh[4][TILE_BLOCKS];
    uint64_t b = (uint64_t)blockIdx.x * TILE_BLOCKS + threadIdx.x;
    uint32_t p = b < n_blocks ? block_letter_counts(chunks + (b << 2)) : 0u;
    sh[0][threadIdx.x] = p & 0xFFu;
    sh[1][threadIdx.x] = (p >> 8) & 0xFFu;
    sh[2][threadIdx.x] = (p >> 16) & 0xFFu;
    sh[3][threadIdx.x] = p >> 24;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 entries, 4 letters
    for (int o = 1; o < TILE_BLOCKS; o <<= 1) {
        uint32_t a[4];
        for (int c = 0; c < 4; ++c) a[c] = threadIdx.x >= (unsigned)o ? sh[c][threadIdx.x - o] : 0u;
        __syncthreads();
        for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] += a[c];
        __syncthreads();
    }
    if (b >= n_blocks) return;
    uint64_t tile = blockIdx.x;
    uint64_t sb_tile = ((b >> 25) << 25) / TILE_BLOCKS; // tile at which this block's superblock starts
    uint32_t own[4] = {p & 0xFFu, (p >> 8) & 0xFFu, (p >> 16) & 0xFFu, p >> 24};
    for (int c = 0; c < 4; ++c) {
        uint64_t abs_before = tile_prefix[tile * 4 + c] + (sh[c][threadIdx.x] - own[c]);
        uint64_t sb_base = tile_prefix[sb_tile * 4 + c];
        reinterpret_cast<uint32_t*>(chunks + (b << 2) + c)[3] = (uint32_t)(abs_before - sb_base);
        if ((b & ((1ull << 25) - 1)) == 0) super[(b >> 25) * 4 + c] = abs_before;
    }
}

// Select directory: for each letter, the block that holds its (m * 2^sel_shift)-th occurrence (a block holds at most 128
// occurrences: up to four samples per letter fall into it at the dense sampling).
__global__ void __launch_bounds__(256) k_select_dir(eb_dev_index x, uint32_t* __restrict__ dir) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= x.n_blocks) return;
    uint32_t p = block_letter_counts(x.chunks + (b << 2));
    for (int c = 0; c < 4; ++c) {
        uint64_t before = eb_block_count(x, b, c);
        uint64_t in_block = (p >> (8 * c)) & 0xFFu;
        // samples m * 2^sel_shift that fall on an occurrence inside this block
        // (with the sample's ordinal inside the block in the entry's free bits, eb_select_guess)
        const uint32_t bb = x.sel_blk_bits, os = bb < 32u ? 7u - (32u - bb) : 0u;
        for (uint64_t m = (before + (1ull << x.sel_shift) - 1) >> x.sel_shift; (m << x.sel_shift) < before + in_block; ++m) {
            uint32_t e = (uint32_t)b;
            if (bb < 32u) e |= (((uint32_t)((m << x.sel_shift) - before)) >> os) << bb;
            dir[x.sel_off[c] + m] = e;
        }
    }
}
__global__ void __launch_bounds__(256) k_fill_u32(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

// scan planes: the x and y code planes of each 64 positions, side by side (16 bytes), without the TERM plane and the
// counters the streaming scan never looks at
__global__ void __launch_bounds__(256) k_scan_planes(const uint4* __restrict__ chunks, uint64_t n_words, uint4* __restrict__ xy) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint4 c0 = chunks[2 * w], c1 = chunks[2 * w + 1];
        xy[w] = make_uint4(c0.x, c1.x, c0.y, c1.y);
    }
}

// FL table (eb_psi_lookup in eb_device.h): one thread per chunk = 16 consecutive occurrences of one letter.  The first is
// found by the directory select; the fifteen that follow by walking the letter's bit mask forward through the blocks (threads
// of neighbouring chunks read neighbouring blocks, so the BWT is streamed once per letter).  tot[c] = occurrences of letter c.
__global__ void __launch_bounds__(256) k_build_psi(eb_dev_index x, uint4* __restrict__ psi, uint64_t tot0, uint64_t tot1, uint64_t tot2, uint64_t tot3,
                                                  uint64_t n_chunks, unsigned long long* __restrict__ n_void) {
    uint32_t voids = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_chunks; t += (uint64_t)gridDim.x * blockDim.x) {
        const int c = t >= x.psi_off[3] ? 3 : t >= x.psi_off[2] ? 2 : t >= x.psi_off[1] ? 1 : 0;
        const uint64_t tot = c == 0 ? tot0 : c == 1 ? tot1 : c == 2 ? tot2 : tot3;
        const uint64_t off = c == 0 ? x.psi_off[0] : c == 1 ? x.psi_off[1] : c == 2 ? x.psi_off[2] : x.psi_off[3];
        const uint64_t r0 = (t - off) << EB_PSI_SHIFT;
        if (r0 >= tot) { psi[t] = make_uint4(0, 0, 0, 0); continue; } // the spare chunk behind a letter's last one: never read
        const uint64_t p0 = eb_select(x, r0, c);
        uint64_t g = p0 >> 7;
        auto letter_masks = [&](uint64_t blk, uint64_t& lo, uint64_t& hi) {
            const uint4* b = x.chunks + (blk << 2);
            const uint4 ch0 = b[0], ch1 = b[1], ch2 = b[2], ch3 = b[3];
            const uint32_t m0 = ~ch0.z & ((c & 2) ? ch0.y : ~ch0.y) & ((c & 1) ? ch0.x : ~ch0.x);
            const uint32_t m1 = ~ch1.z & ((c & 2) ? ch1.y : ~ch1.y) & ((c & 1) ? ch1.x : ~ch1.x);
            const uint32_t m2 = ~ch2.z & ((c & 2) ? ch2.y : ~ch2.y) & ((c & 1) ? ch2.x : ~ch2.x);
            const uint32_t m3 = ~ch3.z & ((c & 2) ? ch3.y : ~ch3.y) & ((c & 1) ? ch3.x : ~ch3.x);
            lo = (uint64_t)m0 | ((uint64_t)m1 << 32);
            hi = (uint64_t)m2 | ((uint64_t)m3 << 32);
        };
        uint64_t lo, hi;
        letter_masks(g, lo, hi);
        { // drop the occurrences up to and including p0
            const uint32_t o = (uint32_t)(p0 & 127u);
            if (o < 63u) lo &= ~0ull << (o + 1u);
            else { lo = 0; if (o >= 64u) hi = o < 127u ? hi & (~0ull << (o - 63u)) : 0ull; }
        }
        uint32_t jump[15];   // rows skipped in front of occurrence k + 1 (0: it sits right behind its predecessor)
        uint32_t bitmap = 0, nj = 0, widest = 0;
        bool ok = true;
        uint64_t prev = p0;
        const uint64_t left = tot - r0 - 1; // occurrences behind the first
        const uint32_t want = left < 15 ? (uint32_t)left : 15u;
#pragma unroll
        for (uint32_t k = 0; k < 15; ++k) jump[k] = 0;
        for (uint32_t k = 1; k <= want && ok; ++k) {
            while (!(lo | hi)) {
                if (++g >= x.n_blocks || ((g << 7) - prev) > (uint64_t)EB_PSI_MAX_JUMP + 1u) { ok = false; break; } // (a longer jump does not fit anyway)
                letter_masks(g, lo, hi);
            }
            if (!ok) break;
            uint64_t q;
            if (lo) { q = (g << 7) + (uint32_t)__ffsll((long long)lo) - 1u; lo &= lo - 1; }
            else { q = (g << 7) + 64u + (uint32_t)__ffsll((long long)hi) - 1u; hi &= hi - 1; }
            const uint64_t skipped = q - prev - 1;
            if (skipped > EB_PSI_MAX_JUMP) { ok = false; break; }
            if (skipped) {
                bitmap |= 1u << (k - 1);
                // (the jumps are kept in a register array indexed by compile-time constants only: nj-th slot by a select chain)
#pragma unroll
                for (uint32_t slot = 0; slot < 15; ++slot) if (slot == nj) jump[slot] = (uint32_t)skipped;
                ++nj;
                widest = skipped > widest ? (uint32_t)skipped : widest;
            }
            prev = q;
        }
        const uint32_t W = nj ? eb_psi_width(nj) : 0u;
        if (!ok || (nj && (widest >> W) != 0u)) { psi[t] = eb_psi_void(); ++voids; continue; }
        // 128 bits: position (38) | bitmap (15) | nj fields of W bits
        uint64_t w_lo = (p0 & ((1ull << 38) - 1ull)) | ((uint64_t)bitmap << 38), w_hi = 0; // bits 0..52 used
        uint32_t at = 53;
#pragma unroll
        for (uint32_t t2 = 0; t2 < 15; ++t2) {
            if (t2 < nj) {
                const uint64_t v = jump[t2];
                if (at < 64u) { w_lo |= v << at; if (at + W > 64u) w_hi |= v >> (64u - at); }
                else w_hi |= v << (at - 64u);
                at += W;
            }
        }
        psi[t] = make_uint4((uint32_t)w_lo, (uint32_t)(w_lo >> 32), (uint32_t)w_hi, (uint32_t)(w_hi >> 32));
    }
    if (voids) atomicAdd(n_void, (unsigned long long)voids);
}

// ------------------------------------------------------------------ host side

// FL table: see eb_psi_lookup (eb_device.h).  n bytes of HBM; refused (EB_ERR_STATE / EB_ERR_NOMEM, the index works on without it)
// beyond 2^38 - 2 symbols or when the device cannot spare the memory.
extern "C" int eb_index_build_psi(eb_index* h) {
    if (!h) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (h->d_psi) return EB_OK;
    h->psi_tried = true;
    if (h->n > EB_PSI_MAX_N) { eb_set_error("the FL table holds 38-bit positions: %llu symbols are too many", (unsigned long long)h->n); return EB_ERR_STATE; }
    if (h->n == 0) { eb_set_error("empty index"); return EB_ERR_STATE; }
    EB_HIP(hipSetDevice(h->device));
    const uint64_t tot[4] = {h->F[1] - h->F[0], h->F[2] - h->F[1], h->F[3] - h->F[2], h->n - h->F[3]};
    uint64_t off[4], n_chunks = 0;
    for (int c = 0; c < 4; ++c) { off[c] = n_chunks; n_chunks += ((tot[c] + 15) >> EB_PSI_SHIFT) + 1; }
    const size_t bytes = n_chunks * sizeof(uint4);
    size_t free_b = 0, total_b = 0;
    EB_HIP(hipMemGetInfo(&free_b, &total_b));
    // leave room for what phase 4 itself allocates (call-set images, scratch): the table may take at most half of what is free
    if (bytes > free_b / 2) { eb_set_error("FL table of %zu bytes refused: %zu bytes of HBM free", bytes, free_b); return EB_ERR_NOMEM; }
    uint4* d = nullptr;
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { eb_set_error("FL table allocation failed (%zu bytes): %s", bytes, hipGetErrorString(e)); return EB_ERR_NOMEM; }
    // (the counter of void chunks comes from the scratch cache: a hipFree would synchronise the device in the middle of a caller's pipeline)
    size_t void_cap = 0;
    unsigned long long* d_void = static_cast<unsigned long long*>(eb_scratch_get(h->device, sizeof(unsigned long long), &void_cap));
    if (!d_void) { (void)hipFree(d); eb_set_error("FL table: no scratch memory"); return EB_ERR_NOMEM; }
    eb_dev_index x = h->dev();
    x.psi = nullptr; // (the kernel's own select goes through the directory)
    for (int c = 0; c < 4; ++c) x.psi_off[c] = off[c];
    const unsigned grid = (unsigned)((n_chunks + 255) / 256 < 65536u * 16u ? (n_chunks + 255) / 256 : 65536u * 16u);
    unsigned long long voids = 0;
    e = hipMemsetAsync(d_void, 0, sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_psi, dim3(grid), dim3(256), 0, h->stream, x, d, tot[0], tot[1], tot[2], tot[3], n_chunks, d_void);
        e = hipMemcpyAsync(&voids, d_void, sizeof voids, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    eb_scratch_put(h->device, d_void, void_cap);
    if (e != hipSuccess) { (void)hipFree(d); eb_set_error("FL table build failed: %s", hipGetErrorString(e)); return EB_ERR_DEVICE; }
    // a BWT in which a letter is absent from 64 rows in a row every few chunks gains nothing from the table: give the memory back
    if (voids > n_chunks / 4) { (void)hipFree(d); eb_set_error("FL table dropped: %llu of %llu chunks void (gaps of 64 rows or more)", voids, (unsigned long long)n_chunks); return EB_ERR_STATE; }
    for (int c = 0; c < 4; ++c) h->psi_off[c] = off[c];
    h->d_psi = d;
    h->device_bytes += bytes;
    h->psi_void_chunks = voids;
    return EB_OK;
}

extern "C" int eb_index_has_psi(const eb_index* h, uint64_t* bytes, uint64_t* void_chunks) {
    if (!h) return 0;
    if (bytes) {
        uint64_t n_chunks = 0;
        const uint64_t tot[4] = {h->F[1] - h->F[0], h->F[2] - h->F[1], h->F[3] - h->F[2], h->n - h->F[3]};
        for (int c = 0; c < 4; ++c) n_chunks += ((tot[c] + 15) >> EB_PSI_SHIFT) + 1;
        *bytes = h->d_psi ? n_chunks * sizeof(uint4) : 0;
    }
    if (void_chunks) *void_chunks = h->d_psi ? h->psi_void_chunks : 0;
    return h->d_psi != nullptr;
}

// Phase 4 calls this once per job begun on the index, with the number of positions the job scans.  The table costs about as much
// as five whole-range passes of the walk it speeds up (and n bytes): it is built once the jobs on the index have scanned three
// times its length -- the third pass of a caller that keeps scanning the whole range, the 24th job of a rank that scans an eighth
// -- so a one-shot run never builds it.  EB_PSI=0: never; EB_PSI=1: with the first job.  A build that fails or is refused is not an
// error of the job that triggered it: the thread's error text is left as it was.
void eb_index_note_phase4_job(eb_index* h, uint64_t positions) {
    h->phase4_positions += positions;
    if (h->d_psi || h->psi_tried) return;
    const int mode = eb_cfg().psi;
    if (mode == 0) return;
    if (mode == 1 || (h->n && h->phase4_positions / h->n >= 3)) {
        const std::string saved = eb_last_error();
        if (eb_index_build_psi(h) != EB_OK) {
            if (eb_cfg().trace_alloc) fprintf(stderr, "[index] %s\n", eb_last_error());
            eb_set_error("%s", saved.c_str());
        }
    }
}

int eb_index_scan_planes(eb_index* h) {
    if (h->d_xy) return EB_OK;
    EB_HIP(hipSetDevice(h->device));
    const uint64_t n_words = h->n_blocks * 2;
    hipError_t e = hipMalloc(&h->d_xy, n_words * sizeof(uint4));
    if (e != hipSuccess) { eb_set_error("scan plane allocation failed (%llu bytes): %s", (unsigned long long)(n_words * 16), hipGetErrorString(e)); return EB_ERR_NOMEM; }
    hipLaunchKernelGGL(k_scan_planes, dim3(4096), dim3(256), 0, h->stream, h->d_chunks, n_words, h->d_xy);
    EB_HIP(hipStreamSynchronize(h->stream));
    EB_HIP(hipGetLastError());
    h->device_bytes += n_words * sizeof(uint4);
    return EB_OK;
}

static double idx_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static int index_alloc(uint64_t n, uint8_t term, int device, eb_index** out, bool defer_scan_ctx = false) {
    int ndev = eb_device_count();
    if (ndev <= 0) {
        eb_set_error("no HIP device available: the ebwt2InDel hot path runs on the GPU only (no CPU fallback)");
        return EB_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        eb_set_error("device %d out of range (have %d)", device, ndev);
        return EB_ERR_ARG;
    }
    EB_HIP(hipSetDevice(device));
    eb_index* h = new eb_index();
    h->device = device;
    h->n = n;
    h->term = term;
    h->n_blocks = (n + 1 + 127) / 128;                 // dna_string.hpp:62
    h->n_super = (n + 1 + 0xFFFFFFFFull) >> 32;        // dna_string.hpp:61
    // The index's stream and, RIGHT BEHIND IT, the streams of its phase-4 jobs (eb_scan_ctx: context kernels, a small call set's
    // walk, record downloads).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says
    // otherwise): a new stream gets a new queue while the pool has room, the least-used one afterwards -- four streams created in a
    // row therefore sit on four different queues, the fourth (the copies') at worst beside an idle stream of the host program, and
    // the scan never shares a queue with the context kernels it runs beside.  (Rounds 2-4 asked for 16 queues instead: more than 4
    // slows every dispatch -- k_scan alone 171 against 143 us for an eighth of config 2, the ordering kernels 45 + 52 against
    // 12 + 15 us -- and which streams end up on one pipe is then left to chance: profiles/r05_hw_queues.txt.)
    // The index's stream carries the phase-4 scans: the chain every pass of a pipelining caller waits for (scan -> ordering -> next
    // scan), so its workgroups get their wave slots ahead of the context kernels' (streams of the lowest priority, eb_scan_ctx_create)
    // when both arrive at once: a rank's eighth 0.345 -> 0.326 ms per pass over 200 passes (the passes right behind a pipeline fill
    // otherwise fall into a phase where the contexts take the slots first and the scan lasts 0.35 instead of 0.25 ms); the whole range
    // is indifferent (profiles/r05_rank_share.txt).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // (hi: numerically smallest = highest priority)
    const double x0 = idx_now();
    hipError_t e = hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi);
    const double x1 = idx_now();
    // (the loader creates them itself a little later, still right behind this one: three streams are ~25 ms in a fresh process,
    // which its first uploads need not wait for)
    if (e == hipSuccess && !defer_scan_ctx && eb_scan_ctx_create(h) != EB_OK) e = hipErrorOutOfMemory;
    const double x2 = idx_now();
    if (e == hipSuccess) e = hipMalloc(&h->d_chunks, h->n_blocks * 64);
    if (e == hipSuccess) e = hipMalloc(&h->d_super, h->n_super * 4 * sizeof(uint64_t));
    if (eb_cfg().trace_index) fprintf(stderr, "[index] index_alloc: first stream %.3f s, phase-4 streams + events + report block %.3f, index memory %.3f\n", x1 - x0, x2 - x1, idx_now() - x2);
    if (e != hipSuccess) {
        eb_set_error("index allocation failed (%llu bytes): %s", (unsigned long long)(h->n_blocks * 64), hipGetErrorString(e));
        eb_index_destroy(h);
        return e == hipErrorOutOfMemory ? EB_ERR_NOMEM : EB_ERR_DEVICE;
    }
    h->device_bytes = h->n_blocks * 64 + h->n_super * 32;
    *out = h;
    return EB_OK;
}

// Packs one piece of text that is already on the device.
static int pack_piece(eb_index* h, const uint8_t* d_ascii, uint64_t piece_pos, uint64_t piece_len, bool last,
                      unsigned long long* d_bad, hipStream_t stream = nullptr) {
    if (!stream) stream = h->stream;
    uint64_t first_chunk = piece_pos / 32;
    uint64_t n_chunks = last ? h->n_blocks * 4 - first_chunk : piece_len / 32;
    if (n_chunks == 0) return EB_OK;
    int aligned = (reinterpret_cast<uintptr_t>(d_ascii) & 15u) == 0;
    // a dispatch holds fewer than 2^32 work-items: pack at most 2^30 chunks (2^35 symbols) per launch
    const uint64_t MAX_CHUNKS = 1ull << 30;
    for (uint64_t done = 0; done < n_chunks; done += MAX_CHUNKS) {
        uint64_t cnt = n_chunks - done < MAX_CHUNKS ? n_chunks - done : MAX_CHUNKS;
        uint64_t byte_off = done * 32;
        uint64_t sub_len = piece_len > byte_off ? piece_len - byte_off : 0;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d_ascii + byte_off, piece_pos + byte_off,
                           sub_len, first_chunk + done, cnt, (uint32_t)h->term, aligned, h->d_chunks, d_bad);
        EB_HIP(hipGetLastError());
    }
    return EB_OK;
}

// Counters, superblock table and F array from the packed planes.
static int finish_index(eb_index* h, unsigned long long* d_bad, const uint8_t* d_text_or_null, const uint8_t* h_text_or_null, int fd = -1) {
    unsigned long long bad = ~0ull;
    EB_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, h->stream));
    EB_HIP(hipStreamSynchronize(h->stream));
    if (bad != ~0ull) {
        uint8_t sym = 0;
        if (h_text_or_null) sym = h_text_or_null[bad];
        else if (fd >= 0) { if (pread(fd, &sym, 1, (off_t)bad) != 1) sym = 0; }
        else if (d_text_or_null) (void)hipMemcpy(&sym, d_text_or_null + bad, 1, hipMemcpyDeviceToHost);
        g_bad_sym = sym;
        g_bad_pos = bad;
        // same wording as dna_string.hpp:91-93
        eb_set_error("Error while reading file: read forbidden character '%c' (ASCII code %d) at position %llu. "
                     "Only A,C,G,T, and %c are admitted in the input BWT!", sym, (int)sym, bad, h->term);
        return EB_ERR_ALPHABET;
    }
    uint64_t n_tiles = (h->n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    uint64_t* d_tiles = nullptr;
    EB_HIP(hipMalloc(&d_tiles, (n_tiles + 1) * 4 * sizeof(uint64_t)));
    EB_HIP(hipMemsetAsync(d_tiles, 0, (n_tiles + 1) * 4 * sizeof(uint64_t), h->stream));
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)n_tiles), dim3(TILE_BLOCKS), 0, h->stream, h->d_chunks, h->n_blocks, d_tiles);
    int rc = eb_scan4_exclusive(d_tiles, n_tiles + 1, h->stream); // element n_tiles becomes the grand total
    if (rc != EB_OK) return rc;
    hipLaunchKernelGGL(k_write_counters, dim3((unsigned)n_tiles), dim3(TILE_BLOCKS), 0, h->stream, h->d_chunks, h->n_blocks,
                       d_tiles, h->d_super);
    uint64_t tot[4];
    EB_HIP(hipMemcpyAsync(tot, d_tiles + n_tiles * 4, sizeof tot, hipMemcpyDeviceToHost, h->stream));
    EB_HIP(hipStreamSynchronize(h->stream));
    EB_HIP(hipGetLastError());
    EB_HIP(hipFree(d_tiles));
    // select directory (eb_select): samples beyond a letter's last occurrence point at the last block
    if (h->n_blocks > 0xFFFFFFFFull) { eb_set_error("BWT too large for the 32-bit select directory"); return EB_ERR_ARG; }
    // Sampling distance of the directory: the right-context walk reads one directory entry and then 1.24 (every 32nd occurrence)
    // or 1.52 (every 128th) blocks per FL step; the dense form costs n/8 instead of n/32 bytes, which the largest inputs
    // (BASELINE configs[4]: 151 G symbols, 277 of 309 GB of HBM in use) cannot spare.  EB_SELECT=<5..10> overrides.
    h->sel_shift = h->n <= (1ull << 35) ? EB_SEL_SHIFT_DENSE : EB_SEL_SHIFT_SPARSE;
    if (eb_cfg().sel_shift) h->sel_shift = (uint32_t)eb_cfg().sel_shift;
    // bits of an entry the block index leaves free carry the sample's ordinal inside its block (at least three, else none)
    {
        uint32_t bb = 1;
        while (bb < 32u && (h->n_blocks >> bb) != 0) ++bb;
        h->sel_blk_bits = bb <= 29u ? (bb < 25u ? 25u : bb) : 32u;
        // EB_SELECT=<shift>,32: plain entries (round 3's interpolation); 25..29: ordinals in coarser units than the index needs
        // (what a larger index would use: tests run the layouts of 2^32 .. 2^36 symbols on small inputs)
        const int v = eb_cfg().sel_blk_bits;
        if (v == 32 || (v >= 25 && v <= 29 && (uint32_t)v >= bb)) h->sel_blk_bits = (uint32_t)v;
    }
    uint64_t sel_entries = 0;
    for (int c = 0; c < 4; ++c) {
        h->sel_off[c] = sel_entries;
        sel_entries += (tot[c] >> h->sel_shift) + 3;
    }
    EB_HIP(hipMalloc(&h->d_sel_dir, sel_entries * sizeof(uint32_t)));
    h->device_bytes += sel_entries * sizeof(uint32_t);
    hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, h->stream, h->d_sel_dir, sel_entries, (uint32_t)(h->n_blocks - 1));
    hipLaunchKernelGGL(k_select_dir, dim3((unsigned)((h->n_blocks + 255) / 256)), dim3(256), 0, h->stream, h->dev(), h->d_sel_dir);
    EB_HIP(hipStreamSynchronize(h->stream));
    EB_HIP(hipGetLastError());
    uint64_t n_term = h->n - (tot[0] + tot[1] + tot[2] + tot[3]);
    h->F[0] = n_term;            // F_A = #TERM            (dna_bwt.hpp:51)
    h->F[1] = h->F[0] + tot[0];  // F_C = F_A + #A         (dna_bwt.hpp:58)
    h->F[2] = h->F[1] + tot[1];
    h->F[3] = h->F[2] + tot[2];
    return EB_OK;
}

extern "C" int eb_index_create_from_device(const void* d_ascii, uint64_t n, uint8_t term, int device, eb_index** out) {
    if (!out || (!d_ascii && n)) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *out = nullptr;
    eb_index* h = nullptr;
    int rc = index_alloc(n, term, device, &h);
    if (rc != EB_OK) return rc;
    unsigned long long* d_bad = nullptr;
    unsigned long long init = ~0ull;
    EB_HIP(hipMalloc(&d_bad, sizeof init));
    EB_HIP(hipMemcpy(d_bad, &init, sizeof init, hipMemcpyHostToDevice));
    EB_HIP(hipDeviceSynchronize()); // the caller's text may have been produced on another stream
    rc = pack_piece(h, static_cast<const uint8_t*>(d_ascii), 0, n, true, d_bad);
    if (rc == EB_OK) rc = finish_index(h, d_bad, static_cast<const uint8_t*>(d_ascii), nullptr);
    (void)hipFree(d_bad);
    if (rc != EB_OK) { eb_index_destroy(h); return rc; }
    *out = h;
    return EB_OK;
}



// ascii: the symbols in host memory, or NULL with fd >= 0: the symbols are the file's bytes (read with pread)
static int index_create_from(const uint8_t* ascii, int fd, uint64_t n, uint8_t term, int device, eb_index** out) {
    *out = nullptr;
    const EbConfig& cfg = eb_cfg();
    const bool trace = cfg.trace_index;
    const double tr0 = idx_now();
    {
        int ndev = eb_device_count();
        if (ndev <= 0) { eb_set_error("no HIP device available: the ebwt2InDel hot path runs on the GPU only (no CPU fallback)"); return EB_ERR_DEVICE; }
        if (device < 0 || device >= ndev) { eb_set_error("device %d out of range (have %d)", device, ndev); return EB_ERR_ARG; }
        EB_HIP(hipSetDevice(device));
    }
    // Upload in pieces (multiple of 128 symbols) through staging buffers so that copies overlap packing.  Large inputs (a file, or
    // any pageable buffer) are first copied into page-locked pieces by a few host threads: the runtime's own staging of pageable
    // memory runs at ~16 GB/s, PCIe at ~50.  EB_INDEX_PINNED: tests exercise the path on small inputs.
    const bool pinned_path = n >= cfg.pinned_min || !ascii;
    const uint64_t PIECE = pinned_path ? cfg.pinned_piece : (256ull << 20);
    const int SLOTS = 4;
    const uint64_t stage_bytes = n < PIECE ? (n ? n : 1) : PIECE;
    const size_t n_pieces = n ? (size_t)((n + PIECE - 1) / PIECE) : 1;
    uint8_t* d_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t* h_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t h_cap[4] = {0, 0, 0, 0};
    size_t d_cap[4] = {0, 0, 0, 0};
    hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long* d_bad = nullptr;
    eb_index* h = nullptr;
    PieceReaders readers;
    readers.src = ascii;
    readers.fd = fd;
    auto release = [&]() { // staging buffers go back to the library's pools (a second index -- two-BWT mode -- takes them from there)
        for (int i = 0; i < 4; ++i) {
            if (d_stage[i]) eb_scratch_put(device, d_stage[i], d_cap[i]);
            if (h_stage[i]) eb_pinned_put(h_stage[i], h_cap[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
        if (d_bad) (void)hipFree(d_bad);
    };
    auto piece_len = [&](size_t k) -> uint64_t { const uint64_t pos = (uint64_t)k * PIECE; return n - pos < PIECE ? n - pos : PIECE; };
    // The index's own allocation (a few GB of hipMalloc: 40-80 ms in a fresh process) and the device staging buffers are made on a
    // side thread while this one pins the host pieces and starts the readers on them.
    unsigned long long init = ~0ull;
    int rc_alloc = EB_OK, rc_ctx = EB_OK;
    std::string err_alloc;
    double ta[6] = {0, 0, 0, 0, 0, 0};
    std::mutex ready_mu;
    std::condition_variable ready_cv;
    bool ready = false;
    std::atomic<hipStream_t> up2{nullptr}; // the second upload stream, once the side thread has created it
    std::thread alloc_thread([&] {
        ta[0] = idx_now();
        (void)hipSetDevice(device);
        ta[1] = idx_now();
        rc_alloc = index_alloc(n, term, device, &h, true);
        ta[2] = idx_now();
        hipError_t ea = rc_alloc == EB_OK ? hipMalloc(&d_bad, sizeof init) : hipSuccess;
        for (int i = 0; i < SLOTS && (size_t)i < n_pieces && ea == hipSuccess && rc_alloc == EB_OK; ++i) {
            d_stage[i] = static_cast<uint8_t*>(eb_scratch_get(device, stage_bytes, &d_cap[i]));
            if (!d_stage[i]) ea = hipErrorOutOfMemory;
            if (ea == hipSuccess) ea = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        }
        ta[3] = idx_now();
        if (ea == hipSuccess && rc_alloc == EB_OK) ea = hipMemcpy(d_bad, &init, sizeof init, hipMemcpyHostToDevice);
        ta[4] = idx_now();
        if (rc_alloc != EB_OK) err_alloc = eb_last_error(); // (the message is thread-local)
        else if (ea != hipSuccess) { rc_alloc = EB_ERR_NOMEM; err_alloc = std::string("staging allocation failed: ") + hipGetErrorString(ea); }
        { std::lock_guard<std::mutex> lk(ready_mu); ready = true; }
        ready_cv.notify_all();
        // The uploads run from here on.  What they do not need follows on this thread, in the order the streams' hardware queues
        // depend on (index_alloc): the index's phase-4 streams right behind its own stream, then the loader's second upload stream
        // (one copy engine moves ~33 GB/s in a fresh process, the link carries ~50: consecutive pieces alternate between two streams).
        if (rc_alloc == EB_OK) {
            rc_ctx = eb_scan_ctx_create(h);
            hipStream_t s2 = nullptr;
            if (n_pieces > 1 && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess) up2.store(s2, std::memory_order_release);
        }
        ta[5] = idx_now();
    });
    struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } join_guard{alloc_thread};
    size_t posted = 0;
    hipError_t e = hipSuccess;
    if (pinned_path) {
        readers.start(cfg.copy_threads, n_pieces);
        for (int i = 0; i < SLOTS && (size_t)i < n_pieces && e == hipSuccess; ++i) {
            h_stage[i] = static_cast<uint8_t*>(eb_pinned_get(stage_bytes, &h_cap[i]));
            if (!h_stage[i]) { e = hipErrorOutOfMemory; break; }
            if (n) readers.post((int)posted, h_stage[i], (uint64_t)posted * PIECE, piece_len(posted));
            ++posted;
        }
    }
    const double tr1 = idx_now();
    {
        std::unique_lock<std::mutex> lk(ready_mu);
        ready_cv.wait(lk, [&] { return ready; });
    }
    int rc = rc_alloc;
    if (rc == EB_OK && e != hipSuccess) { rc = EB_ERR_NOMEM; err_alloc = std::string("staging allocation failed: ") + hipGetErrorString(e); }
    if (rc != EB_OK) {
        alloc_thread.join();
        eb_set_error("%s", err_alloc.c_str());
        release();
        if (h) eb_index_destroy(h);
        return rc;
    }
    const double tr2 = idx_now();
    double tr_read = 0, tr_slot = 0;
    for (size_t k = 0; k < n_pieces; ++k) {
        hipStream_t st = h->stream;
        if (k & 1) { const hipStream_t s2 = up2.load(std::memory_order_acquire); if (s2) st = s2; }
        const int slot = (int)(k % SLOTS);
        const uint64_t pos = (uint64_t)k * PIECE, len = n ? piece_len(k) : 0;
        const bool last = k + 1 == n_pieces;
        const uint8_t* src = ascii ? ascii + pos : nullptr;
        if (pinned_path && len) {
            const double t0 = idx_now();
            if (!readers.wait((int)k)) { eb_set_error("read error in the input file"); rc = EB_ERR_IO; break; }
            tr_read += idx_now() - t0;
            src = h_stage[slot];
        } else if (k >= (size_t)SLOTS && (e = hipEventSynchronize(done[slot])) != hipSuccess) break; // (the device staging buffer is free again)
        if (len && (e = hipMemcpyAsync(d_stage[slot], src, len, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        rc = pack_piece(h, d_stage[slot], pos, len, last, d_bad, st);
        if (rc != EB_OK) break;
        if ((e = hipEventRecord(done[slot], st)) != hipSuccess) break;
        // the slot of the piece BEFORE this one is free once that piece's upload is done -- likely by now, while this one's upload
        // has only just been enqueued (waiting for this piece's own upload here would put a host round trip between any two
        // uploads: 33 GB/s instead of the link's 50) -- and takes the next piece that has no slot yet
        if (pinned_path && k >= 1 && posted < n_pieces) {
            const int free_slot = (int)((k - 1) % SLOTS);
            const double t0 = idx_now();
            if ((e = hipEventSynchronize(done[free_slot])) != hipSuccess) break;
            tr_slot += idx_now() - t0;
            readers.post((int)posted, h_stage[free_slot], (uint64_t)posted * PIECE, piece_len(posted));
            ++posted;
        }
    }
    if (e != hipSuccess) { eb_set_error("upload failed: %s", hipGetErrorString(e)); rc = EB_ERR_DEVICE; }
    alloc_thread.join();
    if (hipStream_t s2 = up2.load()) { // (the counters follow on the index's stream)
        if (hipStreamSynchronize(s2) != hipSuccess && rc == EB_OK) { eb_set_error("upload failed"); rc = EB_ERR_DEVICE; }
        (void)hipStreamDestroy(s2);
    }
    if (rc == EB_OK && rc_ctx != EB_OK) { eb_set_error("phase-4 streams could not be created"); rc = rc_ctx; }
    if (rc != EB_OK) (void)hipStreamSynchronize(h->stream);
    const double tr3 = idx_now();
    if (rc == EB_OK) rc = finish_index(h, d_bad, nullptr, ascii, fd);
    const double tr4 = idx_now();
    release();
    if (trace)
        fprintf(stderr, "[index] side thread: device set %.3f s, index_alloc (stream, index memory) %.3f, staging + events %.3f, first host-to-device copy %.3f; then, beside the uploads: phase-4 streams + second upload stream %.3f\n",
                ta[1] - ta[0], ta[2] - ta[1], ta[3] - ta[2], ta[4] - ta[3], ta[5] - ta[4]);
    if (trace)
        fprintf(stderr, "[index] n=%llu: page-locked pieces + readers started %.3f s, then waited for the side thread %.3f, piece loop %.3f (waits for the readers %.3f, "
                        "for an upload %.3f), counters + directory %.3f, release %.3f\n",
                (unsigned long long)n, tr1 - tr0, tr2 - tr1, tr3 - tr2, tr_read, tr_slot, tr4 - tr3, idx_now() - tr4);
    if (rc != EB_OK) { eb_index_destroy(h); return rc; }
    *out = h;
    return EB_OK;
}

extern "C" int eb_index_create(const uint8_t* ascii, uint64_t n, uint8_t term, int device, eb_index** out) {
    if (!out || (!ascii && n)) { eb_set_error("null argument"); return EB_ERR_ARG; }
    static const uint8_t empty = 0;
    return index_create_from(ascii ? ascii : &empty, -1, n, term, device, out);
}

extern "C" int eb_index_create_from_file(const char* path, uint8_t term, int device, eb_index** out) {
    if (!path || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { eb_set_error("Error: could not find file %s", path); return EB_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); eb_set_error("cannot stat %s", path); return EB_ERR_IO; }
    uint64_t n = (uint64_t)st.st_size; // n = file size (dna_string.hpp:59)
    int rc = index_create_from(nullptr, fd, n, term, device, out);
    close(fd);
    return rc;
}

extern "C" void eb_index_destroy(eb_index* h) {
    if (!h) return;
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->d_chunks) (void)hipFree(h->d_chunks);
    if (h->d_super) (void)hipFree(h->d_super);
    if (h->d_sel_dir) (void)hipFree(h->d_sel_dir);
    if (h->d_xy) (void)hipFree(h->d_xy);
    if (h->d_psi) (void)hipFree(h->d_psi);
    eb_scan_ctx_destroy(h->scan_ctx);
    eb_scratch_release(h->device);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int eb_index_info(const eb_index* h, eb_index_info_t* info) {
    if (!h || !info) { eb_set_error("null argument"); return EB_ERR_ARG; }
    info->n = h->n;
    for (int i = 0; i < 4; ++i) info->F[i] = h->F[i];
    info->n_blocks = h->n_blocks;
    info->device_bytes = h->device_bytes;
    info->device = h->device;
    info->term = h->term;
    return EB_OK;
}
