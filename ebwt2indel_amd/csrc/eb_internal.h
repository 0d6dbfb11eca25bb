// eb_internal.h -- host-side definitions shared by the translation units of libebwt2indel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/ebwt2indel.h"
#include "eb_device.h"

void eb_set_error(const char* fmt, ...);

#define EB_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            eb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return EB_ERR_DEVICE;                                                                 \
        }                                                                                         \
    } while (0)

// streams / events / page-locked report blocks of the phase-4 driver (eb_scan_job.cpp), created with the first job on an index
struct eb_scan_ctx {
    hipStream_t ctx = nullptr, right = nullptr, copy = nullptr; // context kernels (right: the walk of a small call set, beside its extension); record downloads (the scan runs on the index's own stream)
    hipEvent_t last_ctx = nullptr;               // end of the most recent job's context kernels
    std::mutex pool_mu;                          // events / reports: a job may be ended on one host thread while the next is begun on another
    std::vector<hipEvent_t> events;              // recycled between jobs
    std::vector<void*> reports;                  // page-locked JobReport blocks, recycled between jobs
    uint32_t batch_shift = 0;                    // log2 of the tiles per batch of the ordered hand-over (0: not chosen yet); shrinks when a batch overflows
    uint64_t scratch_reserved = 0;               // job capacity (clusters) a ring of device-scratch sets was reserved for (EB_SCAN_PIPELINED callers)
    uint64_t pinned_reserved = 0;                // largest call-set size a ring of page-locked blobs was reserved for (likewise)
};
void eb_scan_ctx_destroy(eb_scan_ctx* c);
int eb_scan_ctx_create(struct eb_index* h); // (eb_scan_job.cpp; called by index_alloc right behind the index's own stream)

struct eb_index {
    int device = -1;
    hipStream_t stream = nullptr;
    uint4* d_chunks = nullptr;
    uint64_t* d_super = nullptr;
    uint4* d_xy = nullptr;         // scan planes (eb_index_scan_planes), built on first use by phase 4 in modes 1 / 3
    uint32_t* d_sel_dir = nullptr; // select directory, letter c at sel_off[c]
    uint64_t sel_off[4] = {0, 0, 0, 0};
    uint64_t n = 0, n_blocks = 0, n_super = 0;
    uint64_t F[4] = {0, 0, 0, 0};
    uint8_t term = '#';
    uint32_t sel_shift = EB_SEL_SHIFT_SPARSE; // log2 of the select directory's sampling distance
    uint32_t sel_blk_bits = 32;               // block bits of a directory entry (eb_select_guess)
    uint4* d_psi = nullptr;                   // FL table (eb_index_build_psi), or null
    uint64_t psi_off[4] = {0, 0, 0, 0};
    uint64_t phase4_positions = 0;            // positions scanned by the phase-4 jobs begun on this index so far (3 n of them: the FL table is built)
    bool psi_tried = false;                   // a build was attempted (and failed or was refused): not again
    uint64_t psi_void_chunks = 0;             // chunks of the table that fall back to the directory select
    uint64_t device_bytes = 0;
    eb_scan_ctx* scan_ctx = nullptr;
    eb_dev_index dev() const {
        eb_dev_index d;
        d.chunks = d_chunks;
        d.super = d_super;
        d.n = n;
        d.n_blocks = n_blocks;
        for (int i = 0; i < 4; ++i) d.F[i] = F[i];
        d.xy = d_xy;
        d.sel_dir = d_sel_dir;
        for (int i = 0; i < 4; ++i) d.sel_off[i] = sel_off[i];
        d.term = term;
        d.sel_shift = sel_shift;
        d.sel_blk_bits = sel_blk_bits;
        d.psi = d_psi;
        for (int i = 0; i < 4; ++i) d.psi_off[i] = psi_off[i];
        return d;
    }
};

struct eb_flags {
    int device = -1;
    hipStream_t stream = nullptr;
    uint64_t n = 0, words = 0;
    uint64_t *d_thrK = nullptr, *d_thrR = nullptr, *d_min = nullptr, *d_da = nullptr, *d_da_rank = nullptr;
    uint64_t* d_flag = nullptr; // thrK & ~minima
    uint4* d_mxy = nullptr;     // two-BWT mode: merged scan planes (eb_flags_merge_planes: behind the traversal, or with the first phase-4 job)
    uint4* d_mxy_spare = nullptr; // their memory, allocated BEFORE the traversal (an allocation right behind the traversal's frees waits for the driver's wipe)
    uint64_t phase4_positions = 0; // closing positions the phase-4 jobs on these flags are expected to scan (a shard's share; 0: all)
    uint64_t da_ones = 0;       // ones in d_da (set with the rank directory): two-BWT phase 4 indexes the second BWT by DA ranks
    bool derived = false;       // d_flag / d_da_rank match the vectors (false between a sharded build and eb_flags_finalize)
    bool complemented = false;  // d_thrK / d_thrR hold the complement (bit = LCP BELOW the threshold) until eb_flags_derive inverts them
    eb_dev_flags dev() const {
        eb_dev_flags f;
        f.thrK = d_thrK;
        f.thrR = d_thrR;
        f.minima = d_min;
        f.flag = d_flag;
        f.mxy = d_mxy;
        f.da = d_da;
        f.da_rank = d_da_rank;
        f.n = n;
        f.words = words;
        return f;
    }
};

// Call set = ONE contiguous blob (also its wire format for the gather between ranks):
//   [header 64 B][eb_cluster x count][eb_chain x n_chains (padded to 8 B)][left words][right words]
struct eb_calls_header {
    uint64_t magic, count, n_chains, k_left, k_right, reserved[3];
};
static const uint64_t EB_CALLS_MAGIC = 0x32534C4C41434245ull; // "EBCALLS2"
// compact wire image (eb_scan_job_device_wire): same header with this magic, then the two record arrays below, then the
// context words exactly as in the blob
static const uint64_t EB_WIRE_MAGIC = 0x33534C4C41434245ull;  // "EBCALLS3"
struct eb_wire_cluster { uint64_t begin; uint32_t len; uint8_t n_slots, right_found; uint16_t right_len; };
struct eb_wire_chain { uint8_t side, sym; uint16_t len; uint32_t support; };
static_assert(sizeof(eb_wire_cluster) == 16 && sizeof(eb_wire_chain) == 8, "wire records");
// tighter still, when every field fits (begin < 2^40, cluster and contexts shorter than 2^16 rows / 8192 bases, supports
// below 2^16 -- any real call set): 12-byte clusters, 4-byte chains
static const uint64_t EB_WIRE4_MAGIC = 0x34534C4C41434245ull; // "EBCALLS4"
struct eb_wire4_cluster { uint32_t begin_lo; uint8_t begin_hi, slots_found; uint16_t len; uint16_t right_len, pad; }; // slots_found: n_slots | right_found << 4
struct eb_wire4_chain { uint16_t side_sym_len; uint16_t support; };                                                   // side | letter << 1 | len << 3
static_assert(sizeof(eb_wire4_cluster) == 12 && sizeof(eb_wire4_chain) == 4, "wire records");
static inline uint64_t eb_wire4_size(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right) {
    uint64_t recs = (count * sizeof(eb_wire4_cluster) + n_chains * sizeof(eb_wire4_chain) + 7) & ~7ull;
    return sizeof(eb_calls_header) + recs + n_chains * (((uint64_t)k_left + 31) / 32) * 8 + count * (((uint64_t)k_right + 31) / 32) * 8;
}
static inline uint64_t eb_wire_size(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right) {
    return sizeof(eb_calls_header) + count * sizeof(eb_wire_cluster) + n_chains * sizeof(eb_wire_chain) +
           n_chains * (((uint64_t)k_left + 31) / 32) * 8 + count * (((uint64_t)k_right + 31) / 32) * 8;
}
struct eb_calls {
    uint8_t* blob = nullptr;
    uint64_t size = 0;
    bool pinned = false; // page-locked (device -> host copies) or plain malloc (unpacked on a host without GPU)
    uint64_t count = 0, n_chains = 0;
    int32_t k_left = 0, k_right = 0;
    eb_cluster* clusters = nullptr;
    eb_chain* chains = nullptr;
    uint64_t* left = nullptr;
    uint64_t* right = nullptr;
    uint64_t left_words() const { return ((uint64_t)k_left + 31) / 32; }
    uint64_t right_words() const { return ((uint64_t)k_right + 31) / 32; }
};
uint64_t eb_calls_blob_size(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right);
// the pool of page-locked host blocks (eb_writer.cpp): call-set blobs and the loader's staging pieces are recycled through it --
// pinning tens of MB per use would cost more than the copy it speeds up.  get: a block of at least `size` bytes, *got its capacity
void* eb_pinned_get(uint64_t size, uint64_t* got);
void eb_pinned_put(void* p, uint64_t cap);
// makes sure the pool holds `count` blocks of at least `size` bytes
void eb_pinned_reserve(uint64_t size, int count, const void* d_warm, uint64_t warm_bytes);
// allocates the blob (zeroed), fills the header and the section pointers; nullptr on failure (error set)
eb_calls* eb_calls_alloc(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right, bool pinned);

// words allocated for a bitvector of n bits (one spare word so neighbours can be read unguarded)
static inline uint64_t eb_bit_words(uint64_t n) { return (n + 63) / 64; }
static inline uint64_t eb_bit_alloc_words(uint64_t n) { return (n + 63) / 64 + 2; }

// eb_scanutil.hip: exclusive prefix sum of `count` vectors of 4 x u64 (in place), on `stream`.
int eb_scan4_exclusive(uint64_t* d_vec4, uint64_t count, hipStream_t stream);

// (re)builds d_da_rank from d_da
int eb_flags_build_da_rank(eb_flags* f);
// (re)builds everything derived from the vectors: d_flag and d_da_rank
int eb_flags_derive(eb_flags* f);
// builds the scan planes of an index if absent (eb_index.hip)
int eb_index_scan_planes(eb_index* h);
// phase 4 counts the positions it scans per index and builds the FL table when a caller keeps scanning (eb_index.hip)
void eb_index_note_phase4_job(eb_index* h, uint64_t positions);
int eb_flags_alloc(uint64_t n, int device, bool with_da, eb_flags** out);
// phase 4's first-job costs (scan planes, device temporaries, a page-locked blob) paid ahead of time (eb_scan_job.cpp); called by the
// traversal on a side thread
int eb_flags_merge_planes(eb_flags* f, eb_index* b1, eb_index* b2); // mode 2: merged scan planes (eb_scan_job.cpp)
void eb_phase4_warm(eb_index* b1, eb_index* b2, uint64_t n);
void eb_phase4_reserve(int device, uint64_t positions, bool two_samples); // the reservations alone (eb_flags_set_da*: mode 3 shows itself there)

// Caching device-scratch allocator (eb_index.hip).  hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises
// the device, so the per-call temporaries of the phase-4 driver are recycled instead of returned to the runtime.
void* eb_scratch_get(int device, size_t bytes, size_t* capacity);
void eb_scratch_put(int device, void* p, size_t capacity);
void eb_scratch_release(int device); // frees everything cached for the device
