// eb_flags.hip -- the flag vectors as an object: allocation, what phase 4 derives from them (flag = thrK & ~minima, the DA rank
// directory), the shard form and its reduction helpers, the document array of mode 3 (from a buffer, a file or HBM), import / export.
// The traversal that fills them is eb_navigate.hip.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include <hipcub/hipcub.hpp>

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

// ones in each group of 8 DA words, one u64 per group (the extra last group counts nothing: its exclusive prefix is the total)
__global__ void __launch_bounds__(256) k_da_group_ones(const uint64_t* __restrict__ da, uint64_t words, uint64_t groups, uint64_t* __restrict__ out) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint64_t w = g * 8 + k;
        if (w < words) s += (uint64_t)__popcll(da[w]);
    }
    out[g] = s;
}

// Rank directory of the document array: ones before every group of 512 positions.  A device-wide exclusive scan (hipcub) over one
// count per group -- 14.7 M of them at 7.55 G positions, which the one-workgroup scan of the letter-count tiles (eb_scanutil.hip: an
// array 64 times shorter) took 68 ms for; this takes well under a millisecond.
int eb_flags_build_da_rank(eb_flags* f) {
    if (!f->d_da) return EB_OK;
    EB_HIP(hipSetDevice(f->device));
    const uint64_t groups = (f->words + 7) / 8 + 1;
    if (groups > 0x7FFFFFFFull) { eb_set_error("document array too long for its rank directory"); return EB_ERR_ARG; }
    if (f->d_da_rank) { (void)hipFree(f->d_da_rank); f->d_da_rank = nullptr; }
    EB_HIP(hipMalloc(&f->d_da_rank, groups * sizeof(uint64_t)));
    const unsigned grid = (unsigned)((groups + 255) / 256);
    hipLaunchKernelGGL(k_da_group_ones, dim3(grid), dim3(256), 0, f->stream, f->d_da, f->words, groups, f->d_da_rank);
    size_t tmp_bytes = 0;
    EB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, f->d_da_rank, f->d_da_rank, (int)groups, f->stream));
    size_t tmp_cap = 0;
    void* d_tmp = eb_scratch_get(f->device, tmp_bytes ? tmp_bytes : 1, &tmp_cap);
    if (!d_tmp) { eb_set_error("out of device memory for the document array's rank directory"); return EB_ERR_NOMEM; }
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, f->d_da_rank, f->d_da_rank, (int)groups, f->stream); // (in place)
    if (e == hipSuccess) e = hipMemcpyAsync(&f->da_ones, f->d_da_rank + (groups - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, f->stream); // the extra last group: ones before the end
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    eb_scratch_put(f->device, d_tmp, tmp_cap);
    if (e != hipSuccess) { eb_set_error("document array rank directory failed: %s", hipGetErrorString(e)); return EB_ERR_DEVICE; }
    EB_HIP(hipGetLastError());
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
// d_raw != NULL: the pieces of a host / file source are only UPLOADED (copy engines: no kernel, so that this can run beside kernels that hold every
// CU for tens of milliseconds at a time), to d_raw + position; the caller packs them later with another call whose d_src is d_raw.
static int da_pack(int device, hipStream_t stream, uint64_t n_bwt, uint64_t* d_da, uint64_t positions, const uint8_t* host, int fd, const uint8_t* d_src, uint64_t n,
                   uint8_t* d_raw = nullptr) {
    EB_HIP(hipSetDevice(device));
    const EbConfig& cfg = eb_cfg();
    const size_t bytes = eb_bit_alloc_words(n_bwt) * 8;
    if (!d_raw) EB_HIP(hipMemsetAsync(d_da, 0, bytes, stream));
    const uint64_t lim = n < n_bwt ? n : n_bwt;
    bool sticky_one = false;
    if (lim > 0 && lim < n_bwt) {
        uint8_t last = 0;
        if (host) last = host[lim - 1];
        else if (d_src) EB_HIP(hipMemcpy(&last, d_src + lim - 1, 1, hipMemcpyDeviceToHost));
        else if (pread(fd, &last, 1, (off_t)(lim - 1)) != 1) { eb_set_error("read error in the document array"); return EB_ERR_IO; }
        sticky_one = last == '1';
    }
    // a document array means phase 4 in two-sample form (twice the chains per cluster): its first job's reservations are made now,
    // beside the upload (the traversal reserved for one sample: it could not know)
    std::thread reserve([dev = device, n_pos = positions ? positions : n_bwt] { eb_phase4_reserve(dev, n_pos, true); });
    struct ReserveGuard { std::thread& t; ~ReserveGuard() { if (t.joinable()) t.join(); } } reserve_guard{reserve};
    auto pack = [&](const uint8_t* d_piece, uint64_t pos, uint64_t len, hipStream_t st) {
        const uint64_t n_words = (len + 63) / 64;
        hipLaunchKernelGGL(k_pack_da, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_piece, len, pos / 64, n_words, d_da);
    };
    int rc = EB_OK;
    if (d_src) {
        const uint64_t PIECE = 1ull << 36; // (one launch's grid holds 2^32 workgroups of 256 words)
        for (uint64_t pos = 0; pos < lim; pos += PIECE) pack(d_src + pos, pos, lim - pos < PIECE ? lim - pos : PIECE, stream);
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
        hipStream_t up[2] = {stream, nullptr};
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
                if (!d_raw) {
                    d_stage[i] = static_cast<uint8_t*>(eb_scratch_get(device, stage_bytes, &d_cap[i]));
                    if (!d_stage[i]) { e = hipErrorOutOfMemory; break; }
                }
                e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
            }
            if (e == hipSuccess && n_pieces > 1 && hipStreamCreateWithFlags(&up[1], hipStreamNonBlocking) != hipSuccess) up[1] = nullptr;
            if (!up[1]) up[1] = stream;
            if (e == hipSuccess && up[1] != stream) {
                // the second stream starts behind the memset of the array
                hipEvent_t zeroed = nullptr;
                e = hipEventCreateWithFlags(&zeroed, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(zeroed, stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(up[1], zeroed, 0);
                if (zeroed) (void)hipEventDestroy(zeroed);
            }
            auto da_now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
            const double tl0 = da_now();
            double t_read = 0, t_slot = 0;
            for (size_t k = 0; k < n_pieces && e == hipSuccess && rc == EB_OK; ++k) {
                const hipStream_t st = up[k & 1];
                const int slot = (int)(k % SLOTS);
                const uint64_t pos = (uint64_t)k * PIECE, len = piece_len(k);
                const uint8_t* src = host ? host + pos : nullptr;
                if (pinned_path) {
                    const double t0 = da_now();
                    if (!readers.wait((int)k)) { eb_set_error("read error in the document array"); rc = EB_ERR_IO; break; }
                    t_read += da_now() - t0;
                    src = h_stage[slot];
                } else if (k >= (size_t)SLOTS && (e = hipEventSynchronize(done[slot])) != hipSuccess) break;
                if (d_raw) {
                    if ((e = hipMemcpyAsync(d_raw + pos, src, len, hipMemcpyHostToDevice, st)) != hipSuccess) break;
                } else {
                    if ((e = hipMemcpyAsync(d_stage[slot], src, len, hipMemcpyHostToDevice, st)) != hipSuccess) break;
                    pack(d_stage[slot], pos, len, st);
                }
                if ((e = hipEventRecord(done[slot], st)) != hipSuccess) break;
                if (pinned_path && k >= 1 && posted < n_pieces) { // the slot of the piece BEFORE this one takes the next piece (as in the index loader)
                    const int free_slot = (int)((k - 1) % SLOTS);
                    const double t0 = da_now();
                    if ((e = hipEventSynchronize(done[free_slot])) != hipSuccess) break;
                    t_slot += da_now() - t0;
                    readers.post((int)posted, h_stage[free_slot], (uint64_t)posted * PIECE, piece_len(posted));
                    ++posted;
                }
            }
            if (cfg.trace_index)
                fprintf(stderr, "[da] %zu pieces of %llu MB%s: loop %.3f s (waits for the readers %.3f, for an upload %.3f)\n", n_pieces, (unsigned long long)(PIECE >> 20),
                        d_raw ? ", uploaded only" : ", uploaded and packed", da_now() - tl0, t_read, t_slot);
            if (up[1] != stream) {
                if (hipStreamSynchronize(up[1]) != hipSuccess && e == hipSuccess) e = hipErrorUnknown;
                (void)hipStreamDestroy(up[1]);
            }
            if (hipStreamSynchronize(stream) != hipSuccess && e == hipSuccess) e = hipErrorUnknown;
        }
        for (int i = 0; i < 4; ++i) {
            if (d_stage[i]) eb_scratch_put(device, d_stage[i], d_cap[i]);
            if (h_stage[i]) eb_pinned_put(h_stage[i], h_cap[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
        if (rc == EB_OK && e != hipSuccess) { eb_set_error("document array upload failed: %s", hipGetErrorString(e)); rc = EB_ERR_DEVICE; }
        if (rc != EB_OK) return rc;
    }
    if (sticky_one && !d_raw) hipLaunchKernelGGL(k_fill_da_tail, dim3(2048), dim3(256), 0, stream, d_da, lim, n_bwt);
    EB_HIP(hipStreamSynchronize(stream));
    EB_HIP(hipGetLastError());
    return EB_OK;
}

static int set_da_from(eb_flags* f, const uint8_t* host, int fd, const uint8_t* d_src, uint64_t n) {
    EB_HIP(hipSetDevice(f->device));
    if (!f->d_da) EB_HIP(hipMalloc(&f->d_da, eb_bit_alloc_words(f->n) * 8));
    int rc = da_pack(f->device, f->stream, f->n, f->d_da, f->phase4_positions, host, fd, d_src, n);
    if (rc != EB_OK) return rc;
    return eb_flags_build_da_rank(f);
}

// The document array of a run that is still in its traversal: loaded from the file into device words of its own (a stream of its own,
// no flags object involved -- eb_main.cpp calls this on a second host thread beside eb_flags_build), then handed to the flags.
struct eb_da_staged {
    int device = -1;
    uint64_t n = 0;
    uint64_t* d_words = nullptr;
    uint8_t* d_raw = nullptr; // the file's bytes as uploaded, still to be packed (eb_flags_set_da_staged does it): see eb_da_stage_file
    uint64_t n_raw = 0;
};

extern "C" int eb_da_stage_file(const char* path, uint64_t n_bwt, int device, eb_da_staged** out) {
    if (!path || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *out = nullptr;
    int fd = open(path, O_RDONLY);
    if (fd < 0) { eb_set_error("Error: could not read file %s", path); return EB_ERR_IO; }
    struct FdGuard { int fd; ~FdGuard() { close(fd); } } fd_guard{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) { eb_set_error("cannot stat %s", path); return EB_ERR_IO; }
    EB_HIP(hipSetDevice(device));
    eb_da_staged* s = new eb_da_staged();
    s->device = device;
    s->n = n_bwt;
    hipStream_t stream = nullptr;
    hipError_t e = hipMalloc(&s->d_words, eb_bit_alloc_words(n_bwt) * 8);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e != hipSuccess) { eb_set_error("document array allocation failed: %s", hipGetErrorString(e)); eb_da_staged_free(s); return EB_ERR_NOMEM; }
    static const uint8_t empty = 0;
    const uint64_t n = (uint64_t)st.st_size;
    // Beside a traversal the packing kernel of a piece would wait for a launch of the node kernels to end (they hold every CU for tens
    // of milliseconds), and the pipeline of pieces with it: 0.48 s for 7.55 GB instead of 0.21.  So the file's bytes only go UP here
    // (copy engines; nothing else crosses PCIe during the traversal) and ONE kernel packs them when the flags take the array over --
    // when the device has room for the raw bytes beside everything else; piece by piece otherwise.
    size_t free_b = 0, total_b = 0;
    const uint64_t lim = n < n_bwt ? n : n_bwt;
    int rc;
    if (lim >= eb_cfg().pinned_min && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 3 * (size_t)lim + ((size_t)8 << 30) && hipMalloc(&s->d_raw, lim) == hipSuccess) {
        s->n_raw = n;
        rc = da_pack(device, stream, n_bwt, s->d_words, 0, nullptr, fd, nullptr, n, s->d_raw);
    } else {
        (void)hipGetLastError();
        s->d_raw = nullptr;
        rc = n ? da_pack(device, stream, n_bwt, s->d_words, 0, nullptr, fd, nullptr, n) : da_pack(device, stream, n_bwt, s->d_words, 0, &empty, -1, nullptr, 0);
    }
    (void)hipStreamDestroy(stream);
    if (rc != EB_OK) { eb_da_staged_free(s); return rc; }
    *out = s;
    return EB_OK;
}

extern "C" int eb_flags_set_da_staged(eb_flags* f, eb_da_staged* s) {
    if (!f || !s || !s->d_words) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (s->n != f->n || s->device != f->device) { eb_set_error("staged document array was made for %llu positions on device %d", (unsigned long long)s->n, s->device); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(f->device));
    auto st_now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    const double q0 = st_now();
    double q1 = q0, q2 = q0;
    if (s->d_raw) { // uploaded beside the traversal, packed now: one kernel over the raw bytes
        int rc = da_pack(f->device, f->stream, f->n, s->d_words, f->phase4_positions, nullptr, -1, s->d_raw, s->n_raw);
        q1 = st_now();
        (void)hipFree(s->d_raw);
        q2 = st_now();
        s->d_raw = nullptr;
        if (rc != EB_OK) return rc;
    }
    if (f->d_da) (void)hipFree(f->d_da);
    f->d_da = s->d_words; // (handed over, not copied)
    s->d_words = nullptr;
    eb_da_staged_free(s);
    // (staging reserved the first phase-4 job's buffers for the whole range; a rank that scans a share of it gets its own sizes here)
    if (f->phase4_positions && f->phase4_positions <= f->n / 2) eb_phase4_reserve(f->device, f->phase4_positions, true);
    const double q3 = st_now();
    const int rc_rank = eb_flags_build_da_rank(f);
    if (eb_cfg().trace_index) fprintf(stderr, "[da] hand-over: packed %.3f s, raw bytes released %.3f, reservations %.3f, rank directory %.3f\n", q1 - q0, q2 - q1, q3 - q2, st_now() - q3);
    return rc_rank;
}

extern "C" void eb_da_staged_free(eb_da_staged* s) {
    if (!s) return;
    if (s->d_words || s->d_raw) (void)hipSetDevice(s->device);
    if (s->d_words) (void)hipFree(s->d_words);
    if (s->d_raw) (void)hipFree(s->d_raw);
    delete s;
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
