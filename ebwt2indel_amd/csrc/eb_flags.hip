// eb_flags.hip -- the flag vectors as an object: allocation, what phase 4 derives from them (flag = thrK & ~minima, the DA rank
// directory), the shard form and its reduction helpers, the document array of mode 3 (from a buffer, a file or HBM), import / export.
// The traversal that fills them is eb_navigate.hip.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "eb_config.h"
#include "eb_internal.h"
#include "eb_pieces.h"

namespace {

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
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
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
    if (reinterpret_cast<uintptr_t>(d_ascii01) & 15u) { eb_set_error("document array in device memory must be 16-byte aligned"); return EB_ERR_ARG; }
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
