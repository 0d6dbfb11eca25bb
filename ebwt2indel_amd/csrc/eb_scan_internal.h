// eb_scan_internal.h -- what the translation units of phase 4 share: the records that travel between its kernels, the
// device-resident description of a job's call set, and the launchers each kernel file exports.
//   eb_scan_kernels.hip   cluster scan + ordering of the survivors      (ebwt2InDel.cpp:1609-1655, :941-1005, :840-934, :1013-1096)
//   eb_context.hip        left-context extension, right-context walk    (:243-319, :325-342)
//   eb_wire.hip           compact wire image of a call set              (no reference counterpart: the inter-rank gather)
//   eb_scan_job.cpp       the job state machine behind eb_scan_clusters_begin / _end (host only)
#pragma once
#include "eb_internal.h"

struct ScanParams {
    int mode;           // 1 one sample, 2 two BWTs, 3 merged BWT + DA
    uint32_t mcov;      // mcov_out
    uint32_t min_len;   // 2 * mcov_out
    int q;              // max_variants_per_position (0 = unlimited)
    int k_left, k_right;
    uint64_t begin, end; // closing positions owned by this call
    uint64_t n;          // merged length
    // ordered != 0: every workgroup scans ONE contiguous run of tiles_per_wg tiles and the survivors are handed over in batches
    // of 2^batch_shift consecutive tiles of the range, batch b into slots [ b * EB_BATCH_SLOTS, ... ): the batches are in BWT
    // order as they lie, only their few dozen members need sorting (k_batch_compact).  ordered == 0: the workgroups stride over
    // the tiles and append to one list in arrival order (position buckets put it in order).
    uint32_t ordered, tiles_per_wg, batch_shift, reserved;
#ifdef EB_SCAN_EXPERIMENTS
    uint32_t dbg;        // EB_SCAN_DEBUG bits: 1 no length histogram, 2 no cluster analysis, 4 no run loop, 8 no queues, 16 no statistics
#endif
};

struct alignas(16) Survivor {
    uint64_t b, e;            // merged coordinates
    uint64_t b0, e0, b1, e1;  // per-BWT ranges (mode 2)
    uint32_t fmask0, fmask1;  // frequent letters per side, bit c = "ACGT"[c]
    uint64_t pad;
};
#define EB_NEEDS_HIST 0x80000000u /* fmask0 of a candidate whose run is longer than the scan's two words: histogram pending */

#define EB_SCAN_PARTIAL 205      /* per-workgroup statistics row: runs, clusters, bases, hist[201], survivors (ordered scan) */
#define EB_REQ_SLOTS 64
#define EB_BATCH_SHIFT_MAX 7u    /* log2 of the tiles per batch to start with (128 x 256 words x 64 = 2 097 152 positions: ~210 survivors on read data) */
#define EB_BATCH_SLOTS 512u      /* survivors a batch can hold; more than that: the job is done again with smaller batches */
#define EB_STAGE_SLOTS 128u      /* the unordered scan's staging buffer in LDS */
#define EB_MAX_BATCHES (1u << 20) /* the ordered hand-over's offsets are scanned by one workgroup */

// Where the sections of a job's call set lie, known on the DEVICE as soon as the survivors are ordered: the kernels behind the
// ordering step read it from device memory, so the host enqueues a whole job without waiting for a count.  Buffers are sized by
// capacity (slots of the batches); the image itself is dense, exactly eb_calls_blob_size(ns, n_chains, ...) bytes.
struct JobLayout {
    uint32_t ns, n_chains;       // surviving clusters, left-context chains
    uint32_t spilled;            // a batch of the ordered hand-over overflowed: ns = n_chains = 0, the job is done again
    uint32_t max_batch;          // most survivors the scan handed to one batch (ordered hand-over)
    uint64_t off_crec, off_left, off_right, blob_bytes;   // byte offsets in the image (clusters start behind the 64-byte header)
    uint64_t wire_off_chain, wire_off_ctx, wire_bytes;    // the same for the compact wire image (EBCALLS4)
};

struct ScanCounters {
    unsigned long long nsurv, runs_closed, clusters, bases, n_long, dropped, spilled;
    unsigned long long hist[201];
    unsigned long long req_ext[EB_REQ_SLOTS], req_right[EB_REQ_SLOTS], req_tab[EB_REQ_SLOTS]; // memory requests counted by k_extend / k_right (count_requests); req_tab: k_right's FL-table chunks
    unsigned long long wire_bad;                                       // records that do not fit the tight wire image
    unsigned int max_batch, pad;                                       // most survivors in one batch of the ordered hand-over
    unsigned long long n_hot, n_todo;                                  // profiling build only (EB_SCAN_EXPERIMENTS): clusters handed to k_scan's hot / todo queues
    JobLayout lay;
};

// What a job reports to the host, in page-locked memory the kernels write straight into (part 1 by the ordering step, part 2
// by k_finish at the very end); the host reads it after the event recorded behind the writing kernel.
struct JobReport {
    unsigned long long nsurv_raw, runs_closed, clusters, bases, dropped;
    unsigned long long hist[201];
    JobLayout lay;
    unsigned long long req_ext, req_right, req_tab, wire_bad; // part 2
    unsigned long long n_hot, n_todo;                         // (profiling build: 0 otherwise)
};

static inline unsigned eb_grid_of(uint64_t threads) { return (unsigned)((threads + 255) / 256); }

// the dense image of a call set with ns clusters and n_chains chains: [64-byte header][eb_cluster x ns][eb_chain x n_chains, padded
// to 8][left words][right words] (eb_calls_blob_size), and its tight wire image [header][12 B x ns][4 B x n_chains, padded][words]
__host__ __device__ static inline void eb_job_layout(JobLayout& L, uint32_t ns, uint32_t n_chains, uint32_t lw, uint32_t rw) {
    L.ns = ns;
    L.n_chains = n_chains;
    L.spilled = 0;
    L.max_batch = 0;
    uint64_t o = sizeof(eb_calls_header) + (uint64_t)ns * sizeof(eb_cluster);
    L.off_crec = o;
    o += ((uint64_t)n_chains * sizeof(eb_chain) + 7) & ~7ull;
    L.off_left = o;
    o += (uint64_t)n_chains * lw * 8;
    L.off_right = o;
    o += (uint64_t)ns * rw * 8;
    L.blob_bytes = o;
    L.wire_off_chain = sizeof(eb_calls_header) + (uint64_t)ns * sizeof(eb_wire4_cluster);
    L.wire_off_ctx = sizeof(eb_calls_header) + (((uint64_t)ns * sizeof(eb_wire4_cluster) + (uint64_t)n_chains * sizeof(eb_wire4_chain) + 7) & ~7ull);
    L.wire_bytes = L.wire_off_ctx + (uint64_t)n_chains * lw * 8 + (uint64_t)ns * rw * 8;
}

// ---- eb_scan_kernels.hip
void eb_launch_merge_planes(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, uint4* mxy);
// k_scan + k_scan_reduce; slots / bcount (ordered) or surv (unordered) receive the survivors
void eb_launch_scan(hipStream_t st, unsigned grid, unsigned dyn_lds, bool prefetch, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f,
                    const ScanParams& p, uint64_t w_lo, uint64_t w_count, Survivor* surv, uint64_t surv_cap, ScanCounters* cnt, unsigned long long* partial,
                    Survivor* slots, uint32_t* bcount, hipEvent_t scan_end);
// ordered hand-over: long-run candidates resolved, batches counted, offsets and the JobLayout written (image header too), batches
// sorted into `sorted`, chain offsets and the chain work list written.  kept: 4 x (n_batches + 1) u32 of scratch.
void eb_launch_order(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, Survivor* slots,
                     const uint32_t* bcount, uint64_t n_batches, uint32_t* kept, ScanCounters* cnt, uint32_t lw, uint32_t rw, uint8_t* blob, JobReport* report,
                     Survivor* sorted, uint32_t* choff, uint64_t* chains);
// unordered hand-over (the fallback; the host knows ns_raw): position buckets, chain counts and offsets, JobLayout, chain list.
// boff: (n_buckets + 2) u32, bcnt: 2 x (n_buckets + 1) u32, tmp: eb_order_tmp_bytes
size_t eb_order_tmp_bytes(uint64_t n_buckets, uint64_t ns_raw);
int eb_launch_order_unordered(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, Survivor* surv,
                              uint64_t ns_raw, uint64_t bucket_base, uint64_t n_buckets, uint32_t* bcnt, uint32_t* boff, void* tmp, size_t tmp_bytes,
                              uint32_t* nch, ScanCounters* cnt, uint32_t lw, uint32_t rw, JobReport* report, Survivor* sorted, uint32_t* choff);
void eb_launch_chain_list(hipStream_t st, const Survivor* sorted, const uint32_t* choff, uint64_t ns, uint64_t* chains, const ScanCounters* cnt, const ScanParams& p,
                          uint8_t* blob);
#define EB_BUCKET_SHIFT 11 /* position buckets of the unordered hand-over: 2048 positions each */

// ---- eb_context.hip
// left-context extension of every chain / right-context walk of every cluster, into the image `blob` laid out by cnt->lay;
// grids by capacity.  gate_min: call-set size from which the walk skips clusters that cannot print -- it must then run BEHIND the
// extension; below it the two are independent.
void eb_launch_extend(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const ScanParams& p, const Survivor* sorted, const uint64_t* chains,
                      ScanCounters* cnt, uint8_t* blob, uint32_t lw, uint64_t cap_chains, int extend_lanes);
void eb_launch_right(hipStream_t st, const eb_dev_index& x1, const eb_dev_index& x2, const eb_dev_flags& f, const ScanParams& p, const Survivor* sorted,
                     const uint32_t* choff, ScanCounters* cnt, uint8_t* blob, uint32_t rw, uint64_t cap_ns, int right_lanes, uint64_t gate_min);
// the job's last kernel: request counts and wire overflow count -> report
void eb_launch_finish(hipStream_t st, ScanCounters* cnt, JobReport* report);

// ---- eb_wire.hip
// tight wire image (EBCALLS4) of the image `blob`, sizes from cnt->lay; fields that do not fit are counted in cnt->wire_bad
void eb_launch_wire4(hipStream_t st, const uint8_t* blob, ScanCounters* cnt, uint8_t* wire, uint32_t lw, uint32_t rw, int32_t k_left, int32_t k_right,
                     uint64_t cap_ns, uint64_t cap_chains);
// full-width wire image (EBCALLS3) for a call set whose counts the host knows; *d_bad counts what does not fit even there
void eb_launch_wire3(hipStream_t st, const uint8_t* blob, uint64_t count, uint64_t n_chains, uint8_t* wire, uint32_t lw, uint32_t rw, int32_t k_left,
                     int32_t k_right, unsigned long long* d_bad);

// ---- each kernel file: load its code object ahead of the first launch (eb_phase4_warm)
void eb_scan_kernels_preload();
void eb_context_preload();
void eb_wire_preload();
