/*
 * ebwt2indel.h -- C ABI of the MI355X-native ebwt2InDel hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, opaque handles, no C++
 * or torch types.  Every entry point names the reference interface it replaces
 * (file:line relative to the upstream ebwt2InDel tree).  The reference has no
 * FFI of its own -- its host program calls the C++ classes dna_string / dna_bwt
 * and the free functions of ebwt2InDel.cpp directly -- so these are the calls a
 * maintainer would make from ebwt2InDel.cpp instead.
 *
 * All work behind this ABI runs on the GPU (hand-written HIP for gfx950).  There
 * is no CPU fallback: without a usable HIP device every call that needs one
 * returns EB_ERR_DEVICE and eb_last_error() says why.
 *
 * Ownership: the caller owns every buffer it passes in; the library owns the
 * device memory behind handles and the host memory behind eb_calls.
 * Threading: a handle is bound to one device and is thread-compatible (one
 * caller at a time); distinct handles may be driven from distinct threads.
 * Errors: int return (0 = ok, <0 = error), text via eb_last_error(); the
 * library never calls exit().
 */
#ifndef EBWT2INDEL_H_
#define EBWT2INDEL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EB_OK            0
#define EB_ERR_ARG      -1
#define EB_ERR_IO       -2
#define EB_ERR_ALPHABET -3 /* byte outside {A,C,G,T,TERM}: dna_string.hpp:90-96 */
#define EB_ERR_DEVICE   -4 /* no HIP device, or a HIP call failed */
#define EB_ERR_NOMEM    -5
#define EB_ERR_STATE    -6

#define EB_MAX_SLOTS 8 /* <=4 frequent symbols per side, 2 sides */

typedef struct eb_index eb_index; /* packed BWT + rank counters, resident on one GPU  (dna_bwt_t) */
typedef struct eb_flags eb_flags; /* LCP_threshold / LCP_minima / DA bitvectors on the GPU          */
typedef struct eb_da_staged eb_da_staged; /* a document array loaded ahead of the flags it belongs to (mode 3) */
typedef struct eb_calls eb_calls; /* phase-4 output: one record per surviving cluster (host memory) */

/* Options of the reference CLI (ebwt2InDel.cpp:20-74, getopt at :1684). */
typedef struct {
    int32_t k_left;                    /* -L, default 31 */
    int32_t k_right;                   /* -R, default 30 */
    int32_t K;                         /* -k, default 16 */
    int32_t max_gap;                   /* -g, default 10 */
    int32_t max_snvs;                  /* -v, default 2  */
    int32_t mcov_out;                  /* -m, default 3  */
    int32_t complexity;                /* -c, default 20 (frozen: ebwt2InDel.cpp:64) */
    int32_t max_variants_per_position; /* -q, default 0 = unlimited */
    int32_t term;                      /* -t, ASCII code, default 35 ('#') */
} eb_params;

/* All-zero params, then the reference's "0 means default" rule (ebwt2InDel.cpp:1740-1746). */
void eb_params_init(eb_params* p);
void eb_params_resolve(eb_params* p);

/* Message of the last failing call on this thread. */
const char* eb_last_error(void);
/* Number of visible HIP devices (0 when there is none). */
int eb_device_count(void);

/* ------------------------------------------------------------------ index (phase 1)
 * Replaces dna_bwt_t(path, TERM): dna_bwt.hpp:36-62 (F array) and
 * dna_string(path, TERM): dna_string.hpp:55-110 (packing), :275-315 (counters).
 * `ascii` is the raw eBWT over {A,C,G,T,TERM}, n bytes, host memory.
 * On EB_ERR_ALPHABET, eb_index_bad_symbol() reports the first offending byte. */
int eb_index_create(const uint8_t* ascii, uint64_t n, uint8_t term, int device, eb_index** out);
int eb_index_create_from_file(const char* path, uint8_t term, int device, eb_index** out);
/* Same, with the ASCII text already in device memory on `device` (synthetic benchmarks). */
int eb_index_create_from_device(const void* d_ascii, uint64_t n, uint8_t term, int device, eb_index** out);
void eb_index_destroy(eb_index* h);
void eb_index_bad_symbol(uint8_t* symbol, uint64_t* position);

typedef struct {
    uint64_t n;            /* dna_bwt::size() (dna_bwt.hpp:231) */
    uint64_t F[4];         /* F_A, F_C, F_G, F_T (dna_bwt.hpp:51-60) */
    uint64_t n_blocks;     /* 128-symbol blocks, ceil((n+1)/128) (dna_string.hpp:62) */
    uint64_t device_bytes; /* HBM held by the handle */
    int32_t device;
    int32_t term;
} eb_index_info_t;
int eb_index_info(const eb_index* h, eb_index_info_t* info);

/* FL table for indexes that are scanned repeatedly (no reference counterpart: dna_bwt::FL, dna_bwt.hpp:115-133, bisects with
 * rank every time).  One 16-byte chunk per 16 occurrences of a letter spells their positions out, so that FL -- the step of the
 * right-context walk extract_dna, ebwt2InDel.cpp:325-342 -- is one 16-byte read instead of a directory read plus a block probe.
 * Costs n bytes of HBM and about the time of five phase-4 walks to build; results are those of select either way.  Phase 4
 * builds it by itself once the jobs begun on an index have scanned three times its length -- the third whole-range pass, the
 * 24th job of a rank that scans an eighth (EB_PSI=0: never, EB_PSI=1: with the first job); this call builds it now.
 * EB_ERR_STATE beyond 2^38 - 2 symbols, EB_ERR_NOMEM if the device cannot spare the memory: the index works on without it. */
int eb_index_build_psi(eb_index* h);
/* 1 iff the table exists; *bytes its size, *void_chunks the chunks that fall back to the directory (a gap of 64 rows or more). */
int eb_index_has_psi(const eb_index* h, uint64_t* bytes, uint64_t* void_chunks);

/* ------------------------------------------------------------------ rank / LF primitives
 * Host-buffer batches.  m queries in, results out; positions are 0 <= i <= n. */
/* dna_string::parallel_rank (dna_string.hpp:140-152): out4[4q..4q+3] = #A,#C,#G,#T in [0,pos[q]). */
int eb_rank4_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint64_t* out4);
/* dna_string::operator[] (dna_string.hpp:113-135): ASCII symbol at pos[q] (< n). */
int eb_access_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint8_t* out);
/* dna_bwt::LF(range_t) (dna_bwt.hpp:138-166): lr[2q..2q+1] = [l,r) -> out8[8q..] = A.first,A.second,C..,G..,T.. */
int eb_lf_range_batch(eb_index* h, const uint64_t* lr, uint64_t m, uint64_t* out8);
/* dna_bwt::LF(range_t, char) (dna_bwt.hpp:168-192): sym[q] in {A,C,G,T}. */
int eb_lf_range_sym_batch(eb_index* h, const uint64_t* lr, const uint8_t* sym, uint64_t m, uint64_t* out2);
/* dna_bwt::FL (dna_bwt.hpp:115-133) = dna_string::select (dna_string.hpp:182-188, 254-272);
 * pos[q] must not be a TERM row (pos >= F_A). */
int eb_fl_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint64_t* out);
/* dna_bwt::LF(sa_node) (dna_bwt.hpp:323-356): nodes[7q..] = first_TERM,first_A,first_C,first_G,first_T,last,depth
 * -> out28[28q..] = the four sa_nodes for A,C,G,T. */
int eb_lf_node_batch(eb_index* h, const uint64_t* nodes7, uint64_t m, uint64_t* out28);

/* Device-buffer form of eb_rank4_batch for benchmarks: d_pos / d_out4 are device
 * pointers on the handle's device; the launch goes on the handle's stream and, when
 * kernel_ms != NULL, is bracketed by HIP events on that stream.  Positions are not
 * checked on the host; one past the end of the BWT counts as the end. */
int eb_rank4_batch_device(eb_index* h, const void* d_pos, uint64_t m, void* d_out4, float* kernel_ms);

/* ------------------------------------------------------------------ phases 2-3
 * Replaces navigate_one_bwt (ebwt2InDel.cpp:555-676) when b2 == NULL and
 * navigate_two_bwts (:679-831) when b2 != NULL: builds, on the GPU,
 *   thrK[i]  = LCP[i] >= K        (LCP_threshold[2i],   :344-355, include.hpp:826-860)
 *   thrR[i]  = LCP[i] >= k_right  (LCP_threshold[2i+1])
 *   minima[i]                     (LCP_minima, :357-391)
 *   da[i]    (two-BWT mode only: update_DA :394-449, find_leaves :474-527)
 * over the n (or n1+n2) merged positions. */
typedef struct {
    uint64_t n;          /* merged length */
    uint64_t leaves;     /* "Processed ... suffix-tree leaves"  */
    uint64_t nodes;      /* "Processed ... suffix-tree nodes"   */
    uint64_t lcp_values; /* positions whose LCP slot was written (reaches n) */
    uint64_t da_values;  /* two-BWT mode: DA slots written (reaches n) */
    uint64_t minima;     /* "Found ... LCP minima" */
    uint64_t rank_queries; /* parallel_rank evaluations issued */
    uint64_t batches;    /* frontier kernel launches */
    double   seconds_leaves, seconds_nodes; /* wall time of each traversal; they run side by side, so these overlap */
    double   seconds_traversal;             /* both traversals + packing of the vectors, first launch to last result */
    /* Sum of the positions whose LCP slot was written, and of their squares, modulo 2^64.  "Every position 1 .. n-1 exactly
     * once" (ebwt2InDel.cpp:575,617,670) means lcp_values == n AND these equal the closed forms -- eb_nav_covers_every_slot_once
     * checks all three; the count alone would let "one slot twice, one never" through (an input that is no eBWT). */
    uint64_t lcp_pos_sum, lcp_pos_sq_sum;
} eb_nav_stats;
/* 1 iff the statistics (of one build, or summed over the shards of a sharded one: all fields add, the sums modulo 2^64) say
 * that every LCP slot 1 .. n-1 was written exactly once.  eb_flags_build collects the COMPLEMENT of the thresholds, which is
 * exact only then, and walks again the reference's way otherwise; callers of eb_flags_build_shard must check it themselves. */
int eb_nav_covers_every_slot_once(const eb_nav_stats* stats);
int eb_flags_build(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, eb_flags** out, eb_nav_stats* stats);
/* Phases 2-3 split over GPUs (no reference counterpart; the reference is one process).  Shard `shard` of `n_shards`
 * walks its share of the Weiner-link subtrees: all shards expand the top of the tree breadth-first, the first level of
 * at least 4096 * n_shards records (EB_NAV_SPLIT_MIN) is cut by a hash of each record's interval start, and from there
 * on a shard follows only its own records.  Above the cut only shard 0 sets bits and counts, below it only the owner:
 * every bit and every counter has ONE writer over all shards, so the vectors (and eb_nav_stats' counters) of the shards
 * combine with an integer SUM (== OR) all-reduce -- RCCL has no bitwise OR.  Then call eb_flags_finalize on every rank. */
int eb_flags_build_shard(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                         eb_nav_stats* stats);
/* SHARD FORM.  Until eb_flags_finalize, the thrK / thrR vectors of a shard hold the COMPLEMENT: bit i set iff this shard
 * wrote position i with an LCP BELOW the threshold (with reads of 100+ bases nearly all writers lie deeper than K and
 * k_right, so the traversal marks the few shallow ones and skips ~95 % of its scattered stores); minima and da are plain.
 * Every position 1..n-1 has one writer over all shards when the input is the eBWT of a read collection -- callers check
 * eb_nav_covers_every_slot_once on the summed statistics -- so the shards' vectors still combine with SUM, and eb_flags_finalize turns
 * the sum into thrK / thrR (inverts it on positions 1..n-1).  eb_flags_device_vectors and eb_flags_export hand out the
 * vectors as they are stored; eb_flags_import_shard_sum takes the sum back when it was formed in host memory. */
/* Device pointers of thrK, thrR, minima, da (NULL in one-BWT mode), eb_flags_words() words each, for the reduction. */
int eb_flags_device_vectors(eb_flags* f, uint64_t* d_vectors[4], uint64_t* words);
/* Ones in thrK, thrR, minima, da (0 where absent).  The SUM reduction of a sharded build is an OR only while every bit
 * has a single writer; a doubly-set bit would carry into its neighbour.  Since popcount(a + b) = popcount(a) +
 * popcount(b) - carries, the ranks check  popcount(reduced) == sum of the ranks' popcounts  (distributed.py does). */
int eb_flags_popcounts(eb_flags* f, uint64_t ones[4]);
/* After the vectors of a sharded build were combined: rebuilds what is derived from them (the DA rank directory). */
int eb_flags_finalize(eb_flags* f);
/* Mode 3: ASCII '0'/'1' document array, '1' -> 1, anything else -> 0 (ebwt2InDel.cpp:1503-1508).  n < BWT length:
 * the remaining slots repeat the file's last byte, as the reference's failed read() leaves its buffer unchanged. */
int eb_flags_set_da(eb_flags* f, const uint8_t* ascii01, uint64_t n);
/* The same array read from a file (the -d argument, ebwt2InDel.cpp:1497-1508: one read() per byte there; here page-locked pieces
 * filled by the loader's reader threads while earlier pieces are uploaded and packed), or taken from ASCII bytes that are already
 * in HBM on the flags' device (16-byte aligned; a generator's output). */
int eb_flags_set_da_file(eb_flags* f, const char* path);
int eb_flags_set_da_device(eb_flags* f, const void* d_ascii01, uint64_t n);
/* The same in two steps, for a caller that loads the -d file WHILE the traversal runs (the reference reads it between phases 3 and 4,
 * ebwt2InDel.cpp:1497-1508; the array does not depend on the traversal): eb_da_stage_file loads and packs it into device words of its
 * own for a BWT of n_bwt positions -- callable on another host thread beside eb_flags_build --, eb_flags_set_da_staged hands those words
 * to the flags (no copy; the staged object is consumed when the call succeeds).  eb_da_staged_free: a staged array that was never handed over. */
int eb_da_stage_file(const char* path, uint64_t n_bwt, int device, eb_da_staged** out);
int eb_flags_set_da_staged(eb_flags* f, eb_da_staged* staged);
void eb_da_staged_free(eb_da_staged* staged);
/* Bit i of each vector is bit (i & 63) of word (i >> 6).  Any pointer may be NULL.
 * export: device -> host (tests, broadcast to other ranks); import: host -> device. */
uint64_t eb_flags_words(const eb_flags* f);
int eb_flags_has_da(const eb_flags* f);
int eb_flags_export(eb_flags* f, uint64_t* thrK, uint64_t* thrR, uint64_t* minima, uint64_t* da);
int eb_flags_import(uint64_t n, int device, const uint64_t* thrK, const uint64_t* thrR, const uint64_t* minima,
                    const uint64_t* da, eb_flags** out);
/* import of vectors in SHARD FORM (the SUM of the shards' eb_flags_export words); returns finalised flags */
int eb_flags_import_shard_sum(uint64_t n, int device, const uint64_t* thrK_below, const uint64_t* thrR_below, const uint64_t* minima,
                              const uint64_t* da, eb_flags** out);
void eb_flags_destroy(eb_flags* f);

/* ------------------------------------------------------------------ phase 4
 * One record per cluster that passed the length and frequent-symbol filters
 * (find_variants, ebwt2InDel.cpp:840-934, 941-1005, 1013-1096), in BWT order, plus one
 * record per left-context chain (one chain per frequent symbol: extract_consensus :303-319).
 * Contexts are packed 2 bits per base (A=0 C=1 G=2 T=3), base e at bits 2*(e&31) of word e>>5:
 *   left  -- ceil(k_left/32) words per chain, in EXTENSION order: base 0 is the frequent symbol
 *            itself (the last character of the context), base e the e-th base prepended to it;
 *   right -- ceil(k_right/32) words per cluster, in text order. */
typedef struct {
    uint64_t begin, end;           /* cluster [begin,end) in merged coordinates (:1618,1632) */
    uint32_t first_chain;          /* index of the cluster's first chain; side 0 first, A<C<G<T within a side */
    uint16_t n_slots;              /* chains of this cluster (<= EB_MAX_SLOTS) */
    uint16_t right_found;          /* 1 iff some cluster position has LCP >= k_right (:982-985) */
    uint32_t right_len;            /* bases of right context extracted there (k_right unless TERM was met) */
    uint32_t reserved;             /* EB_RIGHT_SKIPPED or 0 */
} eb_cluster;
/* eb_cluster.reserved: the right context was not extracted (right_len = 0, words zero) because the cluster can never
 * reach the output: fewer than two of its left contexts have full length (to_file returns at ebwt2InDel.cpp:1256
 * before it touches cluster_nr; pair mode: no pair of full contexts with different last characters, :1161-1163 with
 * :317).  Set by large call sets, whose right-context walk runs behind the left-context extension. */
#define EB_RIGHT_SKIPPED 1u

typedef struct {
    uint8_t  side;                 /* 0/1 (always 0 in one-sample mode) */
    uint8_t  sym;                  /* 'A','C','G','T' */
    uint16_t reserved;
    uint32_t len;                  /* bases obtained; the context is valid iff == k_left (:317) */
    uint32_t support;              /* |LF(range,c)| (:308-310) */
} eb_chain;

typedef struct {
    uint64_t begin, end;       /* scanned range */
    uint64_t runs_closed;      /* clusters closed inside the range (any length) */
    uint64_t clusters;         /* "Analyzed ... clusters" (len >= 2*mcov_out) */
    uint64_t bases_in_runs;    /* sum of closed-run lengths ("Average cluster length") */
    uint64_t survivors;        /* records emitted */
    uint64_t chains;           /* left-context chains extended */
    uint64_t hist[201];        /* CLUST_SIZES (:1604,1627) */
    float    ms_scan, ms_extend, ms_right; /* HIP-event kernel times on the streams the kernels run on */
    float    ms_total;                     /* first scan launch .. last record byte in host memory */
    uint32_t scan_launches;                /* k_scan launches behind ms_scan (1) */
    float    ms_context;                   /* start of the left-context extension .. end of the right-context walk behind it */
    /* memory requests the context kernels issued, counted on the device where they are issued: k_extend one per
     * 64-byte block of a rank pair (1 or 2 per LF step), k_right one per FL-table chunk, select-directory read and block probed */
    uint64_t requests_extend, requests_right;
    /* scans that ran a second time: a batch of the ordered hand-over held more survivors than its slots (the job is then
     * done again with smaller batches), or the unordered survivor list was too short (it runs again with room for all) */
    uint32_t scan_reruns;
    uint32_t batch_tiles;                  /* tiles (16 384 positions each) per batch of the ordered hand-over the job ended with; 0: the unordered hand-over */
    uint64_t requests_right_table;         /* of requests_right: reads of FL-table chunks (16 bytes each; the others are directory reads and 64-byte block probes) */
} eb_scan_stats;

/* Replaces the phase-4 loops run_one_dataset (:1609-1655), run_two_datasets (:1395-1445)
 * and run_two_datasets_da (:1510-1560) restricted to clusters whose closing position (the
 * first unflagged position after the run, :1623) lies in [begin,end); the whole BWT is
 * begin=0, end=n.  Shards that tile [0,n) therefore partition the clusters.  Mode: b2 != NULL -> two BWTs;
 * else flags carrying a DA -> merged BWT + DA; else one sample. */
int eb_scan_clusters(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* p,
                     eb_calls** out, eb_scan_stats* stats);

/* Split form for callers that pipeline calls (the record download of one call overlaps the kernels of the next).
 * _begin ENQUEUES the whole job -- scan, ordering of the survivors, context kernels, wire image -- and returns without waiting
 * for the device (round 5; about 0.1 ms): buffers are sized by capacity, the survivor and chain counts stay in device memory
 * where the kernels read them, and the host learns them from a report the kernels write into page-locked memory.  _wait / _end
 * and the device-image accessors look at that report first: should a batch of the scan's ordered hand-over have held more
 * survivors than its slots, they run the job again with smaller batches (eb_scan_stats.scan_reruns; the size is remembered per
 * index), so a caller never sees a partial call set.  _end then copies the dense image of the call set into a page-locked
 * blob (one copy of exactly its size), waits for it and hands the results over.
 * Several jobs may be in flight on the same handles; each must be ended (or cancelled) exactly once, in any order.
 * Threads: _begin calls on one index are made from one host thread at a time; _wait / _end / the device-image accessors
 * of a begun job may be called from another thread while the next job is being begun (bench.py hands records over
 * that way). */
typedef struct eb_scan_job eb_scan_job;
int eb_scan_clusters_begin(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* p,
                           eb_scan_job** job);
int eb_scan_clusters_end(eb_scan_job* job, eb_calls** out, eb_scan_stats* stats);
void eb_scan_clusters_cancel(eb_scan_job* job);
/* Options for _begin_ex.  EB_SCAN_DEVICE_ONLY: leave the call set in device memory only (no copy to the host) -- for
 * ranks that forward it to rank 0 with RCCL; _end then returns *out = NULL. */
#define EB_SCAN_DEVICE_ONLY 1u
/* EB_SCAN_WIRE: the job also packs the compact wire image (eb_scan_job_device_wire) as its last step, so that asking
 * for it later costs no extra launch or wait. */
#define EB_SCAN_WIRE 2u
/* EB_SCAN_PIPELINED: the caller keeps jobs coming on this index.  While the previous job's context kernels are still running,
 * the scan of this job shares every CU with them instead of waiting for them (round 3), and the library reserves, with the
 * caller's first job of a size, what several live jobs need -- page-locked blobs, device temporaries, copy-engine queues. */
#define EB_SCAN_PIPELINED 4u
int eb_scan_clusters_begin_ex(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* p,
                              uint32_t options, eb_scan_job** job);
/* Blocks until the job's kernels and copies are complete; the job stays valid. */
int eb_scan_job_wait(eb_scan_job* job);
/* The job's call set as one contiguous image in DEVICE memory (same format as eb_calls_blob), valid until _end; its kernels
 * may still be running (eb_scan_job_wait). */
int eb_scan_job_device_blob(eb_scan_job* job, const void** d_data, uint64_t* len);
/* The same call set as a COMPACT wire image in device memory (44 instead of 80 bytes per two-allele cluster), for the
 * gather to rank 0 and its download -- the PCIe link of rank 0 carries every record of an N-GPU job:
 *   [64-byte header, magic "EBCALLS4"]
 *   [{u32 begin_lo; u8 begin_hi; u8 n_slots | right_found << 4 | right_skipped << 5; u16 end-begin; u16 right_len; u16 0} x count]
 *   [{u16 side | letter << 1 | len << 3; u16 support} x n_chains][pad to 8][left words][right words]
 * (first_chain is the running sum of n_slots).  If a field does not fit (begin >= 2^40, a cluster of 2^16 rows or more,
 * a context of 8192 bases or more, a support of 2^16 or more) the image has full-width records instead (magic
 * "EBCALLS3": {u64 begin; u32 end-begin; u8 n_slots; u8 right_found | right_skipped << 1; u16 right_len}, {u8 side; u8 sym; u16 len; u32
 * support}; 56 bytes per two-allele cluster).  eb_calls_unpack reads all images.  Packed by the job with EB_SCAN_WIRE,
 * else on first call; valid until _end.  EB_ERR_STATE if not even the full-width records fit: use
 * eb_scan_job_device_blob then. */
int eb_scan_job_device_wire(eb_scan_job* job, const void** d_data, uint64_t* len);
/* The same wire image copied into the CALLER's host memory (`cap` bytes at `dst`; page-locked memory makes it a DMA at link
 * rate) over this GPU's own PCIe link; *len = its size, also when it does not fit (EB_ERR_ARG then).  What a rank of a one-node
 * multi-GPU run does with its records: the node's host memory is shared, so every rank downloads its own share over its own
 * link into one arena and no link carries another rank's records (the RCCL gather to GPU 0 put all of them on GPU 0's link). */
int eb_scan_job_wire_to_host(eb_scan_job* job, void* dst, uint64_t cap, uint64_t* len);
/* Page-locks `bytes` of the caller's host memory at `p` for such downloads (a shared-memory arena that several processes of one
 * node map: each rank registers the part it writes), and releases the lock again (before the memory is unmapped). */
int eb_host_register(void* p, uint64_t bytes);
int eb_host_unregister(void* p);

typedef struct {
    uint64_t count;                /* clusters */
    uint64_t n_chains;             /* chains over all clusters */
    int32_t k_left, k_right;
    const eb_cluster* clusters;    /* count records */
    const eb_chain* chains;        /* n_chains records */
    const uint64_t* left;          /* n_chains * ceil(k_left/32) words */
    const uint64_t* right;         /* count * ceil(k_right/32) words */
} eb_calls_view;
int eb_calls_get(const eb_calls* c, eb_calls_view* view);
void eb_calls_free(eb_calls* c);
/* Flat byte image of a call set, for the RCCL gather between ranks. */
uint64_t eb_calls_packed_size(const eb_calls* c);
int eb_calls_pack(const eb_calls* c, uint8_t* buf, uint64_t cap);
int eb_calls_unpack(const uint8_t* buf, uint64_t len, eb_calls** out);
/* The call set IS one contiguous blob in that same format: zero-copy access (valid until eb_calls_free). */
int eb_calls_blob(const eb_calls* c, const uint8_t** data, uint64_t* len);

/* ------------------------------------------------------------------ event logic + KisSNP2 writer (host)
 * Replaces to_file (ebwt2InDel.cpp:1254-1330 one sample, :1149-1252 pairs) with distance (:192-240),
 * event_type (:1102-1144), has_run (:144-152).  `two_samples` selects the pair writer.
 * cluster_nr is the running counter (:52), threaded through so shards can be written in order;
 * *events counts records written by the one-sample writer (:1320).  Appends to `path`
 * when `append` != 0, otherwise truncates it. */
int eb_write_snp(const eb_calls* c, const eb_params* p, int two_samples, const char* path, int append,
                 uint64_t* cluster_nr, uint64_t* events);
/* Same, into a malloc'd buffer the caller frees with eb_free(). */
int eb_format_snp(const eb_calls* c, const eb_params* p, int two_samples, uint64_t* cluster_nr, uint64_t* events,
                  char** text, uint64_t* len);
void eb_free(void* p);

/* ------------------------------------------------------------------ environment switches
 * The library reads its switches (EB_PSI, EB_SELECT, EB_NAV_CHAINS, EB_NAV_FORM, EB_NAV_BATCH, EB_SCAN_OVERLAP, EB_SCAN_ORDERED,
 * EB_CONTEXT, EB_INDEX_PINNED, EB_WRITER_THREADS, EB_TRACE: DESIGN.md section 10 -- every one result-neutral, none needed for
 * normal use) ONCE, when it is loaded.  Tests that change them inside one process call this afterwards. */
void eb_config_reload(void);

/* ------------------------------------------------------------------ self-test hooks (host only, no GPU needed)
 * Round trip of the traversal's packed stack records (eb_nav_common.h: 40-bit positions + saturating 16-bit depth;
 * n_fields = 6 one-BWT node, 12 two-BWT node, 4 two-BWT leaf) through the same code the kernels run.  fields_out gets
 * n_fields values, words_out (may be NULL) the packed words, *n_words their number. */
int eb_selftest_pack_fields(int32_t n_fields, const uint64_t* fields, uint64_t depth, uint64_t* fields_out, uint64_t* depth_out,
                            uint64_t* words_out, uint32_t* n_words);

#ifdef __cplusplus
}
#endif
#endif /* EBWT2INDEL_H_ */
