// eb_scan_job.cpp -- the host side of phase 4: a job = one pass of the cluster scan + context extraction over a range of
// closing positions (eb_scan_clusters_begin / _end and friends, include/ebwt2indel.h).  Replaces the driver loops of
// run_one_dataset (ebwt2InDel.cpp:1584-1674), run_two_datasets (:1344-1463) and run_two_datasets_da (:1471-1577) around
// find_variants.  Host code only: the kernels are launched through eb_scan_internal.h.
//
// A job is ENQUEUED WHOLE, without a host wait (round 5).  Until round 4 _begin waited twice for a few words from the device
// (the survivor count sized the ordering step's buffers, survivor and chain counts the call set's image): for a rank's share of
// a multi-GPU run -- an eighth of the positions, 0.23 ms of device work -- the step WAS that begin call (0.36 ms).  Now buffers
// are sized by capacity (the slots of the ordered hand-over's batches), the counts stay on the device (JobLayout: every kernel
// behind the ordering step reads its bounds and section offsets there), and the host learns them from a report the kernels
// write into its page-locked memory -- when it comes to hand the records over, on whatever thread does that.  What can go wrong
// with a speculative enqueue is a batch with more survivors than slots: the layout then says "spilled", the kernels behind it
// find an empty call set, and the job is done again with smaller batches (remembered per index) -- or, beyond that, the
// unordered way, which sizes its list from the scan's count and is the one path that still waits.
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "eb_config.h"
#include "eb_scan_internal.h"

namespace {

#define EB_PINNED_RING_BYTES (2ull << 30) /* most host memory a pipelining caller's ring of eight page-locked blobs may take */

struct DevMem { // per-job temporary from the caching scratch allocator
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

// persistent scan grid: per_cu workgroups per CU, each striding over (or owning a run of) 256-word tiles
unsigned scan_grid(uint64_t words, unsigned per_cu) {
    uint64_t tiles = (words + 255) / 256;
    const uint64_t cap = 256ull * per_cu;
    return (unsigned)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

enum { E_BEGIN, E_SCAN1, E_ORDER, E_EXT0, E_EXT1, E_RIGHT0, E_RIGHT1, E_CTX_DONE, E_END, E_PER_JOB };

// EB_TRACE=begin[:ms]: host-side timeline of every _begin call that takes longer than <ms> (stderr)
struct BeginTrace {
    double limit_ms = 0, t0 = 0;
    char line[512];
    size_t len = 0;
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    BeginTrace() { limit_ms = eb_cfg().trace_begin_ms; if (limit_ms > 0) t0 = now(); line[0] = 0; }
    void at(const char* what) {
        if (limit_ms <= 0 || len > sizeof line - 48) return;
        len += (size_t)snprintf(line + len, sizeof line - len, " %s %.3f", what, now() - t0);
    }
    ~BeginTrace() { if (limit_ms > 0 && now() - t0 > limit_ms) fprintf(stderr, "[begin] %.3f ms:%s\n", now() - t0, line); }
};

}  // namespace

// A phase-4 call in flight.
struct eb_scan_job {
    eb_index* b1 = nullptr;
    eb_scan_ctx* ctx = nullptr;
    uint32_t options = 0;
    uint64_t begin = 0, end = 0;
    int32_t k_left = 0, k_right = 0;
    uint32_t lw = 0, rw = 0;
    eb_dev_index x1, x2;
    eb_dev_flags df;
    ScanParams p;
    hipStream_t s_scan = nullptr, s_ctx = nullptr; // scan (+ ordering of the survivors); context kernels + wire image
    hipStream_t s_order = nullptr;                 // the stream the ordering step ran on
    unsigned overlap_wgs = 0;                      // != 0: the scan shares every CU with the previous job's context kernels
    int extend_lanes = 1, right_lanes = 1;
    bool split = false;                            // the right-context walk on its own stream, beside the extension (small call sets)
    // geometry of the scan and capacities of everything behind it
    uint64_t w_lo = 0, w_count = 0;
    bool ordered = true;
    uint32_t batch_shift = EB_BATCH_SHIFT_MAX;
    uint64_t n_batches = 0, cap_ns = 0, cap_chains = 0, surv_cap = 0;
    DevMem cnt, part, slots, kept, sorted, choff, chains, blob, wire;  // cnt: ScanCounters, then the batches' cursors
    DevMem surv, bcnt, boff, tmp, nch;                                 // the unordered hand-over's
    hipEvent_t e[E_PER_JOB] = {};
    JobReport* report = nullptr; // page-locked; written by the kernels
    eb_calls* calls = nullptr;
    bool validated = false, delivered = false, wire_full_width = false;
    int rc_valid = EB_OK;
    char err_valid[256] = "";
    uint32_t reruns = 0;
    uint64_t wire_bytes = 0;
};

void eb_scan_ctx_destroy(eb_scan_ctx* c) {
    if (!c) return;
    for (hipStream_t st : {c->ctx, c->right, c->copy})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->last_ctx) (void)hipEventDestroy(c->last_ctx);
    for (void* p : c->reports) (void)hipHostFree(p);
    delete c;
}

namespace {

void job_release(eb_scan_job* job) { // events and the report go back to the context's pools; the DevMems to the scratch cache
    eb_scan_ctx* ctx = job->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        for (hipEvent_t e : job->e)
            if (e) ctx->events.push_back(e);
        if (job->report) ctx->reports.push_back(job->report);
    }
    if (job->calls) eb_calls_free(job->calls);
    delete job;
}

}  // namespace

// streams of the index's phase-4 jobs (created with the index, right behind its own stream: see index_alloc), the "previous
// job's context kernels are done" event, a set of events and a report block for the first job
int eb_scan_ctx_create(eb_index* b1) {
    if (b1->scan_ctx) return EB_OK;
    eb_scan_ctx* ctx = new eb_scan_ctx();
    b1->scan_ctx = ctx;
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // lo: numerically greatest = lowest priority (the scan's stream has the highest: index_alloc)
    const double y0 = BeginTrace::now();
    EB_HIP(hipStreamCreateWithPriority(&ctx->ctx, hipStreamNonBlocking, prio_lo));
    EB_HIP(hipStreamCreateWithPriority(&ctx->right, hipStreamNonBlocking, prio_lo));
    EB_HIP(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    if (eb_cfg().trace_index) fprintf(stderr, "[index] three phase-4 streams: %.1f ms\n", BeginTrace::now() - y0);
    EB_HIP(hipEventCreateWithFlags(&ctx->last_ctx, hipEventDisableTiming));
    EB_HIP(hipEventRecord(ctx->last_ctx, ctx->ctx));
    for (int k = 0; k < E_PER_JOB; ++k) {
        hipEvent_t e = nullptr;
        EB_HIP(hipEventCreate(&e));
        ctx->events.push_back(e);
    }
    void* hb = nullptr;
    EB_HIP(hipHostMalloc(&hb, (sizeof(JobReport) + 255) & ~(size_t)255, hipHostMallocDefault));
    ctx->reports.push_back(hb);
    return EB_OK;
}

namespace {

// ---- the context kernels, the wire image and the job's last kernel, behind the ordering step (E_ORDER recorded on s_scan)
int enqueue_contexts(eb_scan_job* job) {
    eb_scan_ctx* ctx = job->ctx;
    const hipStream_t sE = job->s_ctx;
    ScanCounters* cnt = job->cnt.as<ScanCounters>();
    const uint64_t cap_ns = job->cap_ns ? job->cap_ns : 1, cap_chains = job->cap_chains ? job->cap_chains : 1;
    const uint64_t gate_min = eb_cfg().right_gate_min;
    // A LARGE call set keeps the memory system busy with either kernel alone, so nothing is lost by running the right-context
    // walk BEHIND the left-context extension on the same stream -- and it can then skip the clusters whose chains came out too
    // short to ever print (the gate in k_right): one request in nine saved.  A SMALL call set (a rank's share of a multi-GPU
    // run: no cluster is skipped, its records equal the reference's cluster for cluster) is another matter: 30 dependent steps
    // per lane and only a few waves per SIMD -- each kernel waits for memory LATENCY, not for lines -- so its walk runs on a
    // stream of its own, beside its own extension and beside the next job's scan (a rank's eighth at 7.55 G positions: 0.320-0.335 ms
    // per pass against 0.340-0.347 one behind the other; two alternating sets of such streams: 0.50-0.59, not kept;
    // profiles/r05_rank_share.txt).
    const bool split = job->split && job->cap_ns <= gate_min; // (never with a gated walk)
    const hipStream_t sR = split ? ctx->right : sE;
    if (sE != job->s_order) EB_HIP(hipStreamWaitEvent(sE, job->e[E_ORDER], 0));
    if (split) EB_HIP(hipStreamWaitEvent(sR, job->e[E_ORDER], 0));
    EB_HIP(hipEventRecord(job->e[E_EXT0], sE));
    eb_launch_extend(sE, job->x1, job->x2, job->p, job->sorted.as<Survivor>(), job->chains.as<uint64_t>(), cnt, job->blob.as<uint8_t>(), job->lw, cap_chains, job->extend_lanes);
    EB_HIP(hipEventRecord(job->e[E_EXT1], sE));
    EB_HIP(hipEventRecord(job->e[E_RIGHT0], sR));
    eb_launch_right(sR, job->x1, job->x2, job->df, job->p, job->sorted.as<Survivor>(), job->choff.as<uint32_t>(), cnt, job->blob.as<uint8_t>(), job->rw, cap_ns,
                    job->right_lanes, split ? ~0ull : gate_min);
    EB_HIP(hipEventRecord(job->e[E_RIGHT1], sR));
    if (split) EB_HIP(hipStreamWaitEvent(sR, job->e[E_EXT1], 0)); // what follows needs both kernels' records
    if (job->options & EB_SCAN_WIRE) {
        int rc = job->wire.alloc(eb_wire4_size(job->cap_ns, job->cap_chains, job->k_left, job->k_right));
        if (rc != EB_OK) return rc;
        eb_launch_wire4(sR, job->blob.as<uint8_t>(), cnt, job->wire.as<uint8_t>(), job->lw, job->rw, job->k_left, job->k_right, job->cap_ns, job->cap_chains);
    }
    eb_launch_finish(sR, cnt, job->report);
    EB_HIP(hipEventRecord(job->e[E_CTX_DONE], sR));
    EB_HIP(hipEventRecord(ctx->last_ctx, sR));
    EB_HIP(hipGetLastError());
    return EB_OK;
}

// the scan waits for the previous job's context kernels unless it is the one that shares the CUs with them
int scan_preamble(eb_scan_job* job) {
    // (A caller that pipelines jobs gets the record download of job i under the kernels of job i+1; a full-width scan itself
    // starts only after job i's context kernels -- whichever kernel arrives first owns every wave slot and the other starves:
    // 4.4 against 3.9 ms per step, round 2.)
    if (!job->overlap_wgs) EB_HIP(hipStreamWaitEvent(job->s_scan, job->ctx->last_ctx, 0));
    EB_HIP(hipEventRecord(job->e[E_BEGIN], job->s_scan));
    return EB_OK;
}

unsigned scan_dyn_lds(unsigned overlap_wgs) {
    // static LDS of k_scan is ~17.3 KB; with a dynamic request that brings a workgroup to 160 KB / k (minus a little), k fit a CU
    // (a workgroup may ask for 64 KB at most: k = 1 and 2 both end at two per CU)
    if (!overlap_wgs) return 0;
    unsigned total = (160u * 1024u) / overlap_wgs - 256u;
    if (total > 64u * 1024u) total = 64u * 1024u;
    return total - 18u * 1024u;
}

// Buffer sizes of an ordered job over `words` 64-position words -- one statement for the job itself and for the reservations made
// ahead of the first job (eb_phase4_reserve: a block reserved one batch short of what the job asks for is of no use to it).  The batch
// count does not depend on the grid: tile indexes run up to grid x tiles_per_wg < tiles + the largest grid.
struct OrderedSizes {
    uint64_t n_batches, cap_ns, cap_chains;
    size_t cnt, part, slots, kept, sorted, choff, chains, blob;
};
OrderedSizes ordered_sizes(uint64_t words, uint32_t batch_shift, bool two_samples, int k_left, int k_right) {
    OrderedSizes z;
    const uint64_t tiles = (words + 255) / 256;
    z.n_batches = ((tiles + 256u * 8u) >> batch_shift) + 1;
    uint64_t cap = z.n_batches * EB_BATCH_SLOTS;
    if (cap > 0x1FFFFFFFull) cap = 0x1FFFFFFFull; // (a call set has fewer than 2^29 clusters: more survivors than that is an error either way)
    z.cap_ns = cap;
    z.cap_chains = cap * (two_samples ? 8u : 4u);
    z.cnt = ((sizeof(ScanCounters) + 255) & ~(size_t)255) + z.n_batches * 4;
    z.part = (size_t)256 * 8 * EB_SCAN_PARTIAL * 8;
    z.slots = z.n_batches * EB_BATCH_SLOTS * sizeof(Survivor);
    z.kept = 4 * (z.n_batches + 1) * 4;
    z.sorted = z.cap_ns * sizeof(Survivor);
    z.choff = z.cap_ns * 4;
    z.chains = z.cap_chains * 8;
    z.blob = eb_calls_blob_size(z.cap_ns, z.cap_chains, (uint32_t)k_left, (uint32_t)k_right);
    return z;
}

// ---- the ordered hand-over: everything enqueued, nothing waited for.  EB_ERR_STATE + *fits = false: too many batches.
int enqueue_ordered(eb_scan_job* job, bool* fits) {
    *fits = true;
    const unsigned grid = scan_grid(job->w_count, job->overlap_wgs ? job->overlap_wgs : 8);
    const uint64_t tiles = (job->w_count + 255) / 256;
    ScanParams q = job->p;
    q.ordered = 1;
    q.tiles_per_wg = (uint32_t)((tiles + grid - 1) / grid);
    q.batch_shift = job->batch_shift;
    q.reserved = 0;
    const OrderedSizes z = ordered_sizes(job->w_count, q.batch_shift, job->p.mode != 1, (int)job->k_left, (int)job->k_right);
    job->n_batches = z.n_batches;
    if (job->n_batches > EB_MAX_BATCHES) { *fits = false; return EB_OK; }
    job->cap_ns = z.cap_ns;
    job->cap_chains = z.cap_chains;
    const size_t cnt_bytes = (sizeof(ScanCounters) + 255) & ~(size_t)255;
    int rc;
    if ((rc = job->cnt.alloc(z.cnt)) != EB_OK || (rc = job->part.alloc(z.part)) != EB_OK || (rc = job->slots.alloc(z.slots)) != EB_OK ||
        (rc = job->kept.alloc(z.kept)) != EB_OK || (rc = job->sorted.alloc(z.sorted)) != EB_OK || (rc = job->choff.alloc(z.choff)) != EB_OK ||
        (rc = job->chains.alloc(z.chains)) != EB_OK || (rc = job->blob.alloc(z.blob)) != EB_OK)
        return rc;
    ScanCounters* cnt = job->cnt.as<ScanCounters>();
    uint32_t* bcount = reinterpret_cast<uint32_t*>(job->cnt.as<uint8_t>() + cnt_bytes);
    const hipStream_t st = job->s_scan;
    EB_HIP(hipMemsetAsync(job->cnt.p, 0, cnt_bytes + job->n_batches * 4, st)); // counters, layout, the batches' cursors
    if ((rc = scan_preamble(job)) != EB_OK) return rc;
    eb_launch_scan(st, grid, scan_dyn_lds(job->overlap_wgs), job->overlap_wgs != 0, job->x1, job->x2, job->df, q, job->w_lo, job->w_count, nullptr, 0, cnt,
                   job->part.as<unsigned long long>(), job->slots.as<Survivor>(), bcount, job->e[E_SCAN1]);
    // The ordering step follows the scan on ITS stream, ahead of the previous job's context kernels still queued on the context
    // stream.  (Round 5, a rank's eighth: on the context stream -- the scan stream free for the next scan 45 us earlier -- the pass
    // takes 0.40 ms instead of 0.33: the ordering then queues behind the previous job's extension.)
    // (A stream of its own for the ordering step -- the next scan then no longer queues behind it -- or the copy stream: no gain,
    // 0.345 / 0.401 against 0.344 ms, profiles/r05_rank_share.txt.)
    const hipStream_t so = st;
    eb_launch_order(so, job->x1, job->x2, job->df, q, job->slots.as<Survivor>(), bcount, job->n_batches, job->kept.as<uint32_t>(), cnt, job->lw, job->rw,
                    job->blob.as<uint8_t>(), job->report, job->sorted.as<Survivor>(), job->choff.as<uint32_t>(), job->chains.as<uint64_t>());
    EB_HIP(hipEventRecord(job->e[E_ORDER], so));
    job->s_order = so;
    EB_HIP(hipGetLastError());
    return enqueue_contexts(job);
}

// ---- the unordered hand-over (EB_SCAN_ORDERED=0, or batches that overflow whatever their size): survivors appended to one list
// in arrival order, sized from the scan's own count -- this path waits for the device twice.
int run_unordered(eb_scan_job* job) {
    job->ordered = false;
    const unsigned grid = scan_grid(job->w_count, job->overlap_wgs ? job->overlap_wgs : 8);
    ScanParams q = job->p;
    q.ordered = 0;
    q.tiles_per_wg = 0;
    q.batch_shift = 0;
    q.reserved = 0;
    const hipStream_t st = job->s_scan;
    int rc;
    if ((rc = job->cnt.alloc(sizeof(ScanCounters))) != EB_OK || (rc = job->part.alloc((size_t)grid * EB_SCAN_PARTIAL * 8)) != EB_OK) return rc;
    ScanCounters* cnt = job->cnt.as<ScanCounters>();
    if (!job->surv_cap) job->surv_cap = job->w_count / 16 + 4096; // survivors are ~1e-4 per position; grown on demand
    uint64_t ns_raw = 0;
    for (;;) {
        if ((rc = job->surv.alloc(job->surv_cap * sizeof(Survivor))) != EB_OK) return rc;
        EB_HIP(hipMemsetAsync(job->cnt.p, 0, sizeof(ScanCounters), st));
        if ((rc = scan_preamble(job)) != EB_OK) return rc;
        eb_launch_scan(st, grid, scan_dyn_lds(job->overlap_wgs), job->overlap_wgs != 0, job->x1, job->x2, job->df, q, job->w_lo, job->w_count, job->surv.as<Survivor>(),
                       job->surv_cap, cnt, job->part.as<unsigned long long>(), nullptr, nullptr, job->e[E_SCAN1]);
        EB_HIP(hipMemcpyAsync(&job->report->nsurv_raw, &cnt->nsurv, 8, hipMemcpyDeviceToHost, st));
        EB_HIP(hipStreamSynchronize(st));
        ns_raw = job->report->nsurv_raw;
        if (ns_raw <= job->surv_cap) break;
        job->surv_cap = ns_raw + 1024; // rare: scan again with room for every survivor
        ++job->reruns;
    }
    if (ns_raw > 0x1FFFFFFFull) { eb_set_error("too many surviving clusters in one call; shard the range"); return EB_ERR_STATE; }
    const uint64_t n = job->p.n;
    const uint64_t bucket_base = job->begin & ~((1ull << EB_BUCKET_SHIFT) - 1ull);
    const uint64_t range_end = job->end > n ? n : job->end;
    const uint64_t n_buckets = ((range_end > bucket_base ? range_end - bucket_base : 0) >> EB_BUCKET_SHIFT) + 2;
    if (n_buckets > 0x7FFFFFFEull) { eb_set_error("range too long for one call; shard it"); return EB_ERR_STATE; }
    const size_t tmp_bytes = eb_order_tmp_bytes(n_buckets, ns_raw ? ns_raw : 1);
    if ((rc = job->bcnt.alloc((n_buckets + 1) * 8)) != EB_OK || (rc = job->boff.alloc((n_buckets + 2) * 4)) != EB_OK || (rc = job->tmp.alloc(tmp_bytes)) != EB_OK ||
        (rc = job->sorted.alloc((ns_raw ? ns_raw : 1) * sizeof(Survivor))) != EB_OK || (rc = job->nch.alloc((ns_raw ? ns_raw : 1) * 4)) != EB_OK ||
        (rc = job->choff.alloc((ns_raw + 3) * 4)) != EB_OK)
        return rc;
    if ((rc = eb_launch_order_unordered(st, job->x1, job->x2, job->df, q, job->surv.as<Survivor>(), ns_raw, bucket_base, n_buckets, job->bcnt.as<uint32_t>(),
                                        job->boff.as<uint32_t>(), job->tmp.p, tmp_bytes, job->nch.as<uint32_t>(), cnt, job->lw, job->rw, job->report,
                                        job->sorted.as<Survivor>(), job->choff.as<uint32_t>())) != EB_OK)
        return rc;
    EB_HIP(hipStreamSynchronize(st)); // the report holds the counts now
    const uint64_t ns = job->report->lay.ns, n_chains = job->report->lay.n_chains;
    job->cap_ns = ns;
    job->cap_chains = n_chains;
    if ((rc = job->chains.alloc((n_chains ? n_chains : 1) * 8)) != EB_OK || (rc = job->blob.alloc(eb_calls_blob_size(ns, n_chains, job->k_left, job->k_right))) != EB_OK)
        return rc;
    eb_launch_chain_list(st, job->sorted.as<Survivor>(), job->choff.as<uint32_t>(), ns, job->chains.as<uint64_t>(), cnt, q, job->blob.as<uint8_t>());
    EB_HIP(hipEventRecord(job->e[E_ORDER], st));
    job->s_order = st;
    EB_HIP(hipGetLastError());
    return enqueue_contexts(job);
}

// Several jobs of one size will be alive at once with a pipelining caller (begun, context kernels running, being handed over,
// being read) and each holds its own set of device temporaries.  The scratch cache would grow by one set whenever the number of
// live jobs reaches a new maximum -- a dozen hipMallocs, 16-19 ms, in the middle of the pipeline (round 3).  Such a caller pays
// for a ring of sets once, with its first job of a size.  Bounded: the ring may take an eighth of the HBM that is free right now,
// and a failed allocation ends the reservation (eb_scratch_get empties the whole cache before it gives up).
void reserve_scratch_ring(eb_scan_job* job) {
    eb_scan_ctx* ctx = job->ctx;
    if (!(job->options & EB_SCAN_PIPELINED) || !job->ordered || job->cap_ns == ctx->scratch_reserved) return;
    ctx->scratch_reserved = job->cap_ns; // (also after a refusal: asked once per job size, not once per job)
    DevMem* all[] = {&job->cnt, &job->part, &job->slots, &job->kept, &job->sorted, &job->choff, &job->chains, &job->blob, &job->wire};
    size_t set_bytes = 0, free_b = 0, total_b = 0;
    for (DevMem* m : all)
        if (m->p) set_bytes += m->cap;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    if (5 * set_bytes > free_b / 8) return;
    std::vector<std::pair<void*, size_t>> held;
    bool failed = false;
    for (int copy = 0; copy < 5 && !failed; ++copy)
        for (DevMem* m : all) {
            if (!m->p || failed) continue;
            size_t cap = 0;
            void* q = eb_scratch_get(job->b1->device, m->cap - m->cap / 16, &cap); // a block of this job's class (a cached one of that class does as well)
            if (q) held.emplace_back(q, cap);
            else failed = true;
        }
    for (const auto& h : held) eb_scratch_put(job->b1->device, h.first, h.second);
}

// ---- before anything of the job is looked at: the ordering step's report is on the host and says the call set is whole.  A batch
// that overflowed: the job is done again with batches small enough for the density the scan counted (remembered per index, so
// the next jobs start there), or the unordered way when no batch size helps.
int validate(eb_scan_job* job) {
    if (job->validated) {
        if (job->rc_valid != EB_OK) eb_set_error("%s", job->err_valid);
        return job->rc_valid;
    }
    job->validated = true;
    auto fail = [&](int rc) {
        job->rc_valid = rc;
        snprintf(job->err_valid, sizeof job->err_valid, "%s", eb_last_error());
        (void)hipDeviceSynchronize(); // whatever part of the job went out must be done with its buffers
        return rc;
    };
    hipError_t e = hipEventSynchronize(job->e[E_ORDER]);
    if (e != hipSuccess) { eb_set_error("phase-4 job failed: %s", hipGetErrorString(e)); return fail(EB_ERR_DEVICE); }
    while (job->ordered && job->report->lay.spilled) {
        ++job->reruns;
        if (eb_cfg().trace_scan) fprintf(stderr, "[scan] a batch of the ordered hand-over overflowed (%llu survivors in %llu batches of %u tiles, %u in the fullest): again with smaller batches\n",
                                         (unsigned long long)job->report->nsurv_raw, (unsigned long long)job->n_batches, 1u << job->batch_shift, job->report->lay.max_batch);
        if ((e = hipEventSynchronize(job->e[E_CTX_DONE])) != hipSuccess) { eb_set_error("phase-4 job failed: %s", hipGetErrorString(e)); return fail(EB_ERR_DEVICE); }
        // batches small enough that the fullest one would have come out half full (survivors spread evenly inside a batch, as they do)
        const uint64_t fullest = job->report->lay.max_batch, seen = job->report->nsurv_raw;
        uint32_t shrink = 1;
        while ((fullest >> shrink) > EB_BATCH_SLOTS / 2 && shrink < 8) ++shrink;
        int rc;
        bool fits = job->batch_shift > 0;
        if (fits) {
            job->batch_shift = job->batch_shift > shrink ? job->batch_shift - shrink : 0;
            job->ctx->batch_shift = job->batch_shift + 1; // (stored + 1: 0 means "not chosen yet")
            memset(job->report, 0, sizeof(JobReport));
            rc = enqueue_ordered(job, &fits);
            if (rc != EB_OK) return fail(rc);
        }
        if (!fits) {
            memset(job->report, 0, sizeof(JobReport));
            job->surv_cap = seen + 1024; // (the ordered scan counted every survivor, the ones it could not hand over too)
            if ((rc = run_unordered(job)) != EB_OK) return fail(rc);
            break;
        }
        if ((e = hipEventSynchronize(job->e[E_ORDER])) != hipSuccess) { eb_set_error("phase-4 job failed: %s", hipGetErrorString(e)); return fail(EB_ERR_DEVICE); }
    }
    if (job->report->nsurv_raw > 0x1FFFFFFFull) { eb_set_error("too many surviving clusters in one call; shard the range"); return fail(EB_ERR_STATE); }
    return EB_OK;
}

// ---- the call set in host memory (unless the caller keeps it on the device): one page-locked blob, one copy of the dense image
int deliver(eb_scan_job* job) {
    if (job->delivered) return EB_OK;
    eb_scan_ctx* ctx = job->ctx;
    const JobLayout& L = job->report->lay;
    if (job->options & EB_SCAN_DEVICE_ONLY) {
        job->delivered = true;
        return EB_OK;
    }
    const uint64_t ns = L.ns, n_chains = L.n_chains;
    if ((job->options & EB_SCAN_PIPELINED) && ns) { // several call sets of this size will be alive at once: pin their blobs now, not in mid-pipeline
        const uint64_t need = L.blob_bytes;
        bool reserve = false;
        {
            std::lock_guard<std::mutex> lk(ctx->pool_mu);
            if (need > ctx->pinned_reserved) { reserve = true; ctx->pinned_reserved = need + need / 16; }
        }
        // (at most EB_PINNED_RING_BYTES page-locked in all: larger call sets pin their blobs one by one)
        if (reserve && 8 * (need + need / 16) <= EB_PINNED_RING_BYTES) eb_pinned_reserve(need + need / 16, 8, job->blob.p, L.blob_bytes);
    }
    eb_calls* calls = eb_calls_alloc(ns, n_chains, job->k_left, job->k_right, ns != 0);
    if (!calls) return EB_ERR_NOMEM;
    job->calls = calls;
    if (ns) {
        const hipStream_t st = ctx->copy;
        uint8_t* dst = calls->blob;
        const uint8_t* src = job->blob.as<uint8_t>();
        if (L.blob_bytes >= (8u << 20) && L.off_crec < L.off_right && L.off_right <= L.blob_bytes) {
            // The chains' records and their left contexts -- half of the image -- are final when the extension ends: they travel while
            // the right-context walk still runs (a job by itself: first launch to records on the host 3.6 -> 3.15 ms at 7.55 G positions, the first
            // pass on an index 4.15 -> 3.6 ms;
            // a pipelining caller's downloads were hidden behind the next pass anyway); header, clusters and right contexts follow.
            EB_HIP(hipStreamWaitEvent(st, job->e[E_EXT1], 0));
            EB_HIP(hipMemcpyAsync(dst + L.off_crec, src + L.off_crec, L.off_right - L.off_crec, hipMemcpyDeviceToHost, st));
            EB_HIP(hipStreamWaitEvent(st, job->e[E_CTX_DONE], 0));
            EB_HIP(hipMemcpyAsync(dst, src, L.off_crec, hipMemcpyDeviceToHost, st));
            if (L.blob_bytes > L.off_right) EB_HIP(hipMemcpyAsync(dst + L.off_right, src + L.off_right, L.blob_bytes - L.off_right, hipMemcpyDeviceToHost, st));
        } else {
            EB_HIP(hipStreamWaitEvent(st, job->e[E_CTX_DONE], 0));
            EB_HIP(hipMemcpyAsync(dst, src, L.blob_bytes, hipMemcpyDeviceToHost, st));
        }
        EB_HIP(hipEventRecord(job->e[E_END], st));
    }
    job->delivered = true;
    return EB_OK;
}

hipEvent_t end_event(eb_scan_job* job) { return (job->calls && job->report->lay.ns) ? job->e[E_END] : job->e[E_CTX_DONE]; }

}  // namespace

extern "C" int eb_scan_clusters_begin_ex(eb_index* b1, eb_index* b2, eb_flags* f, uint64_t begin, uint64_t end, const eb_params* prm,
                                         uint32_t options, eb_scan_job** out_job) {
    if (!b1 || !f || !prm || !out_job) { eb_set_error("null argument"); return EB_ERR_ARG; }
    BeginTrace bt;
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
    } else if (!f->d_mxy) { // mode 2: planes of the two BWTs merged by the document array (the traversal leaves them behind; flags put together otherwise: now)
        int rcp = eb_flags_merge_planes(f, b1, b2);
        if (rcp != EB_OK) return rcp;
    }
    // a caller that keeps scanning an index gets the FL table for its right-context walks (eb_index.hip)
    // (two BWTs: each is credited its share of the merged positions)
    eb_index_note_phase4_job(b1, b2 ? (uint64_t)((double)(end - begin) * (double)b1->n / (double)(n ? n : 1)) : end - begin);
    if (b2) eb_index_note_phase4_job(b2, (uint64_t)((double)(end - begin) * (double)b2->n / (double)(n ? n : 1)));

    int rcc = eb_scan_ctx_create(b1);
    if (rcc != EB_OK) return rcc;
    eb_scan_ctx* ctx = b1->scan_ctx;
    const EbConfig& cfg = eb_cfg();
    // THE SCAN BESIDE THE PREVIOUS JOB'S CONTEXT KERNELS (round 3).  The scan is bound by vector instructions (~300 per 64-position
    // word), the context kernels by the fabric's rate of random 128-byte lines with next to no arithmetic: run one after the
    // other they add up, run side by side on the same CUs they do not -- if each CU is PARTITIONED: the scan holds overlap_wgs
    // workgroups per CU (its grid is 256 x overlap_wgs, and a dynamic LDS request of 160 KB / overlap_wgs keeps one more from
    // fitting), issues the loads of its next tile before it works on the current one (k_scan<.., true>: half the waves must cover
    // the same latency), and the context kernels -- no LDS -- keep the memory system busy.  Applied when the caller keeps jobs coming
    // (EB_SCAN_PIPELINED) and the previous job's context kernels have not finished; a job with nothing to share the chip with gets
    // the full-width scan.  EB_SCAN_OVERLAP=0: never; =k: always, k per CU.  (4 per CU in round 3, when the walk took as long as
    // the extension; with the FL table the context kernels are the shorter side and the scan gets every slot it can use: 2.60 /
    // 2.46 / 2.36 / 2.22 ms per step for 4 / 5 / 6 / 8, profiles/r04_overlap_ab.txt.)
    // A SMALL call set (a rank's share: at most 2^19 clusters by the capacity of its batches) is another regime: its context kernels
    // are a few waves per SIMD waiting for memory latency, and they must find wave slots whenever they arrive -- the scan then
    // holds HALF of each CU (0.373 / 0.362 / 0.335 / 0.426 ms per pass of a rank's eighth for 8 / 6 / 4 / 3 workgroups per CU).
    const uint64_t w_lo = begin >> 6, w_hi = (end + 63) >> 6;
    const uint64_t w_count = w_hi > w_lo ? w_hi - w_lo : 0;
    const bool small_set = ((((w_count + 255) / 256) >> EB_BATCH_SHIFT_MAX) + 2) * EB_BATCH_SLOTS <= cfg.right_gate_min;
    unsigned overlap_wgs = 0;
    if (cfg.scan_overlap >= 0) overlap_wgs = (unsigned)cfg.scan_overlap;
    else if (options & EB_SCAN_PIPELINED) {
        const unsigned share = small_set ? 4u : ((b1->d_psi && (!b2 || b2->d_psi)) ? 8u : 4u);
        overlap_wgs = hipEventQuery(ctx->last_ctx) == hipErrorNotReady ? share : 0u;
    }

    eb_scan_job* job = new eb_scan_job();
    struct JobGuard {
        eb_scan_job** j;
        ~JobGuard() {
            if (!*j) return;
            (void)hipDeviceSynchronize(); // nothing may still use the job's buffers
            job_release(*j);
        }
    } job_guard{&job};
    job->b1 = b1;
    job->ctx = ctx;
    job->options = options;
    job->begin = begin;
    job->end = end;
    job->k_left = prm->k_left;
    job->k_right = prm->k_right;
    job->lw = (uint32_t)((prm->k_left + 31) / 32);
    job->rw = (uint32_t)((prm->k_right + 31) / 32);
    job->x1 = b1->dev();
    job->x2 = b2 ? b2->dev() : b1->dev();
    job->df = f->dev();
    ScanParams& p = job->p;
    memset(&p, 0, sizeof p);
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
    job->s_scan = b1->stream;
    job->s_ctx = ctx->ctx;
    job->overlap_wgs = overlap_wgs;
    // lanes per chain / per cluster of the context kernels (eb_context.hip): one lane per chain for one BWT of up to 2^33 symbols,
    // one lane per cluster with the FL table
    job->extend_lanes = cfg.extend_lanes ? cfg.extend_lanes : (p.mode == 2 || job->x1.n > (1ull << 33) ? 4 : 1);
    const bool have_psi = job->x1.psi && (p.mode != 2 || job->x2.psi);
    job->right_lanes = cfg.right_lanes ? cfg.right_lanes : (have_psi ? 1 : 4);
    job->w_lo = w_lo;
    job->w_count = w_count;
    job->split = small_set;
    for (DevMem* m : {&job->cnt, &job->part, &job->slots, &job->kept, &job->sorted, &job->choff, &job->chains, &job->blob, &job->wire, &job->surv, &job->bcnt, &job->boff,
                      &job->tmp, &job->nch})
        m->device = b1->device;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        for (hipEvent_t& e : job->e)
            if (!ctx->events.empty()) { e = ctx->events.back(); ctx->events.pop_back(); }
        if (!ctx->reports.empty()) { job->report = static_cast<JobReport*>(ctx->reports.back()); ctx->reports.pop_back(); }
    }
    for (hipEvent_t& e : job->e)
        if (!e) EB_HIP(hipEventCreate(&e));
    if (!job->report) {
        void* hb = nullptr;
        EB_HIP(hipHostMalloc(&hb, (sizeof(JobReport) + 255) & ~(size_t)255, hipHostMallocDefault));
        job->report = static_cast<JobReport*>(hb);
    }
    memset(job->report, 0, sizeof(JobReport));
    bt.at("set-up");
    // The scan hands its survivors over in BWT-ordered batches unless told otherwise (EB_SCAN_ORDERED=0) or no batch size fits.
    int rc;
    bool ordered = cfg.scan_ordered != 0;
    if (ordered) {
        job->batch_shift = ctx->batch_shift ? ctx->batch_shift - 1 : EB_BATCH_SHIFT_MAX;
        if (!ctx->batch_shift && cfg.scan_ordered > 0) { // tests: the size the batches start with
            job->batch_shift = 0;
            while ((2u << job->batch_shift) <= (unsigned)cfg.scan_ordered) ++job->batch_shift;
        }
        if ((rc = enqueue_ordered(job, &ordered)) != EB_OK) return rc;
    }
    if (!ordered && (rc = run_unordered(job)) != EB_OK) return rc;
    bt.at("enqueued");
    reserve_scratch_ring(job);
    bt.at("reserves");
    *out_job = job;
    job = nullptr;
    return EB_OK;
}

extern "C" int eb_scan_clusters_end(eb_scan_job* job, eb_calls** out, eb_scan_stats* stats) {
    if (!job) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (out) *out = nullptr;
    eb_index* b1 = job->b1;
    EB_HIP(hipSetDevice(b1->device));
    int rc = validate(job);
    if (rc == EB_OK) rc = deliver(job);
    std::string err = rc != EB_OK ? eb_last_error() : "";
    hipError_t werr = hipSuccess;
    if (rc == EB_OK) werr = hipEventSynchronize(end_event(job));
    else (void)hipDeviceSynchronize();
    if (rc == EB_OK && werr == hipSuccess && stats) {
        const JobReport& R = *job->report;
        memset(stats, 0, sizeof *stats);
        stats->begin = job->begin;
        stats->end = job->end;
        stats->runs_closed = R.runs_closed;
        stats->clusters = R.clusters;
        stats->bases_in_runs = R.bases;
        stats->survivors = R.lay.ns;
        stats->chains = R.lay.n_chains; // one left-context chain per frequent symbol
        for (int k = 0; k < 201; ++k) stats->hist[k] = R.hist[k];
        (void)hipEventElapsedTime(&stats->ms_scan, job->e[E_BEGIN], job->e[E_SCAN1]);
        (void)hipEventElapsedTime(&stats->ms_extend, job->e[E_EXT0], job->e[E_EXT1]);
        (void)hipEventElapsedTime(&stats->ms_right, job->e[E_RIGHT0], job->e[E_RIGHT1]);
        for (int a : {E_EXT0, E_RIGHT0})        // first start .. last end of the two (side by side for a small call set)
            for (int b : {E_EXT1, E_RIGHT1}) {
                float t = 0;
                (void)hipEventElapsedTime(&t, job->e[a], job->e[b]);
                if (t > stats->ms_context) stats->ms_context = t;
            }
        (void)hipEventElapsedTime(&stats->ms_total, job->e[E_BEGIN], end_event(job));
        stats->scan_launches = 1;
        stats->requests_extend = R.req_ext;
        stats->requests_right = R.req_right;
        stats->requests_right_table = R.req_tab;
        stats->scan_reruns = job->reruns;
        stats->batch_tiles = job->ordered ? 1u << job->batch_shift : 0u;
        if (eb_cfg().trace_scan) {
            auto at = [&](int k) { float t = 0; (void)hipEventElapsedTime(&t, job->e[E_BEGIN], job->e[k]); return t; };
            fprintf(stderr, "[scan] ms since the scan's start: scan done %.3f  ordered %.3f  extend %.3f-%.3f  right -%.3f  last kernel %.3f  records on host %.3f  (%s, %u reruns)\n",
                    at(E_SCAN1), at(E_ORDER), at(E_EXT0), at(E_EXT1), at(E_RIGHT1), at(E_CTX_DONE), stats->ms_total, job->ordered ? "ordered" : "unordered", job->reruns);
            if (R.n_hot) // (profiling build)
                fprintf(stderr, "[scan] of %llu clusters (runs of at least 2 x mcov rows; %llu runs closed): %llu to the hot queue (1 in %.1f), %llu on to the histogram (1 in %.1f), %llu survivors\n",
                        (unsigned long long)R.clusters, (unsigned long long)R.runs_closed, (unsigned long long)R.n_hot, (double)R.clusters / (double)R.n_hot, (unsigned long long)R.n_todo,
                        (double)R.clusters / (double)(R.n_todo ? R.n_todo : 1), (unsigned long long)R.nsurv_raw);
        }
    }
    eb_calls* calls = job->calls;
    job->calls = nullptr;
    job_release(job); // the device scratch goes back to the cache
    if (rc != EB_OK) {
        if (calls) eb_calls_free(calls);
        eb_set_error("%s", err.c_str());
        return rc;
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
    int rc = validate(job);
    if (rc == EB_OK) rc = deliver(job);
    if (rc != EB_OK) return rc;
    EB_HIP(hipEventSynchronize(end_event(job)));
    return EB_OK;
}

extern "C" int eb_scan_job_device_blob(eb_scan_job* job, const void** d_data, uint64_t* len) {
    if (!job || !d_data || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(job->b1->device));
    int rc = validate(job);
    if (rc != EB_OK) return rc;
    *d_data = job->blob.p;
    *len = job->report->lay.blob_bytes;
    return EB_OK;
}

extern "C" int eb_scan_job_device_wire(eb_scan_job* job, const void** d_data, uint64_t* len) {
    if (!job || !d_data || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(job->b1->device));
    int rc = validate(job);
    if (rc != EB_OK) return rc;
    EB_HIP(hipEventSynchronize(job->e[E_CTX_DONE])); // the records (and, with EB_SCAN_WIRE, the packing) are complete
    const JobLayout& L = job->report->lay;
    const hipStream_t st = job->ctx->copy;
    if (!job->wire.p) { // not asked for at _begin: pack now
        if ((rc = job->wire.alloc(L.wire_bytes + 16)) != EB_OK) return rc;
        ScanCounters* cnt = job->cnt.as<ScanCounters>();
        eb_launch_wire4(st, job->blob.as<uint8_t>(), cnt, job->wire.as<uint8_t>(), job->lw, job->rw, job->k_left, job->k_right, L.ns, L.n_chains);
        EB_HIP(hipMemcpyAsync(&job->report->wire_bad, &cnt->wire_bad, 8, hipMemcpyDeviceToHost, st));
        EB_HIP(hipStreamSynchronize(st));
    }
    job->wire_bytes = L.wire_bytes;
    if (job->report->wire_bad && !job->wire_full_width) { // a field does not fit the tight image: full-width records
        job->wire_full_width = true;
        const uint64_t bytes = eb_wire_size(L.ns, L.n_chains, job->k_left, job->k_right);
        const uint64_t off_bad = (bytes + 7) & ~7ull;
        if ((rc = job->wire.alloc(off_bad + 8)) != EB_OK) return rc;
        unsigned long long* d_bad = reinterpret_cast<unsigned long long*>(job->wire.as<uint8_t>() + off_bad);
        EB_HIP(hipMemsetAsync(d_bad, 0, 8, st));
        eb_launch_wire3(st, job->blob.as<uint8_t>(), L.ns, L.n_chains, job->wire.as<uint8_t>(), job->lw, job->rw, job->k_left, job->k_right, d_bad);
        EB_HIP(hipMemcpyAsync(&job->report->wire_bad, d_bad, 8, hipMemcpyDeviceToHost, st));
        EB_HIP(hipStreamSynchronize(st));
        EB_HIP(hipGetLastError());
        if (job->report->wire_bad) { eb_set_error("%llu records do not fit the compact wire image", job->report->wire_bad); return EB_ERR_STATE; }
    }
    if (job->wire_full_width) job->wire_bytes = eb_wire_size(L.ns, L.n_chains, job->k_left, job->k_right);
    *d_data = job->wire.p;
    *len = job->wire_bytes;
    return EB_OK;
}

// The job's wire image into the CALLER's host memory, over this GPU's own PCIe link: what a rank of a one-node multi-GPU run
// does with its share of the records (the host's memory is shared: no rank's link carries another rank's records).
extern "C" int eb_scan_job_wire_to_host(eb_scan_job* job, void* dst, uint64_t cap, uint64_t* len) {
    if (!job || !dst || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    const void* d = nullptr;
    uint64_t n = 0;
    int rc = eb_scan_job_device_wire(job, &d, &n);
    if (rc != EB_OK) return rc;
    *len = n;
    if (n > cap) { eb_set_error("the wire image takes %llu bytes, the caller's buffer holds %llu", (unsigned long long)n, (unsigned long long)cap); return EB_ERR_ARG; }
    const hipStream_t st = job->ctx->copy;
    EB_HIP(hipMemcpyAsync(dst, d, n, hipMemcpyDeviceToHost, st));
    EB_HIP(hipStreamSynchronize(st));
    return EB_OK;
}

// What the FIRST phase-4 job on an index pays before its scan can start, paid beside the traversal instead (eb_flags_build calls
// this on a side thread once its own buffers are allocated): the scan planes of a one-BWT index, the device temporaries of a
// whole-range job (into the scratch cache) and a page-locked blob of about the size a call set of read data takes.  The
// executable's phase 4 on the 7.55 GB file: 57 ms in round 4, of which the kernels took 6.
// Mode 2: the scan planes of the two BWTs merged by the document array (what k_scan<2> streams instead of an index's own planes).
int eb_flags_merge_planes(eb_flags* f, eb_index* b1, eb_index* b2) {
    if (f->d_mxy) return EB_OK;
    EB_HIP(hipSetDevice(f->device));
    const size_t bytes = eb_bit_alloc_words(f->n) * sizeof(uint4);
    if (f->d_mxy_spare) { f->d_mxy = f->d_mxy_spare; f->d_mxy_spare = nullptr; }
    else {
        hipError_t e = hipMalloc(&f->d_mxy, bytes);
        if (e != hipSuccess) { f->d_mxy = nullptr; eb_set_error("merged scan plane allocation failed (%zu bytes): %s", bytes, hipGetErrorString(e)); return EB_ERR_NOMEM; }
    }
    EB_HIP(hipMemsetAsync(f->d_mxy, 0, bytes, f->stream));
    eb_launch_merge_planes(f->stream, b1->dev(), b2->dev(), f->dev(), f->d_mxy);
    EB_HIP(hipStreamSynchronize(f->stream));
    EB_HIP(hipGetLastError());
    return EB_OK;
}

// Reservations of the FIRST phase-4 job over `positions` positions on `device`, made ahead of it (beside the traversal, or beside
// the upload of the document array, which is when a mode-3 run turns out to be one): the device temporaries of an ordered job at the
// starting batch size, and a page-locked block for the call set -- ~1e-4 clusters per position x ~85 bytes; the pool hands a block
// out for requests between half its size and its size.
void eb_phase4_reserve(int device, uint64_t positions, bool two_samples) {
    if (hipSetDevice(device) != hipSuccess) return;
    const std::string saved = eb_last_error();
    const OrderedSizes z = ordered_sizes((positions + 63) / 64 + 1, EB_BATCH_SHIFT_MAX, two_samples, 31, 30);
    if (z.n_batches <= EB_MAX_BATCHES) {
        const size_t sizes[] = {z.cnt, z.part, z.slots, z.kept, z.sorted, z.choff, z.chains, z.blob};
        std::vector<std::pair<void*, size_t>> held;
        for (size_t bytes : sizes) {
            size_t got = 0;
            void* q = eb_scratch_get(device, bytes, &got);
            if (!q) break;
            held.emplace_back(q, got);
        }
        for (const auto& h : held) eb_scratch_put(device, h.first, h.second);
    }
    uint64_t pinned_cap = 0;
    if (void* h = eb_pinned_get(positions / 100 + 4096, &pinned_cap)) eb_pinned_put(h, pinned_cap);
    eb_set_error("%s", saved.c_str());
}

void eb_phase4_warm(eb_index* b1, eb_index* b2, uint64_t n) {
    if (hipSetDevice(b1->device) != hipSuccess) return;
    const std::string saved = eb_last_error();
    if (!b2) (void)eb_index_scan_planes(b1);
    eb_scan_kernels_preload(); // (3-4 ms of code-object loading in front of the first scan otherwise)
    eb_context_preload();
    eb_wire_preload();
    eb_phase4_reserve(b1->device, n, b2 != nullptr);
    // a throw-away copy into a page-locked block: the first device-to-host copy of a process waits 10-30 ms for the runtime to open
    // a copy-engine queue
    uint64_t pinned_cap = 0;
    if (void* h = eb_pinned_get(1u << 20, &pinned_cap)) {
        if (b1->scan_ctx && b1->d_chunks && b1->n_blocks * 64 >= (1u << 20)) {
            (void)hipMemcpyAsync(h, b1->d_chunks, 1u << 20, hipMemcpyDeviceToHost, b1->scan_ctx->copy);
            (void)hipStreamSynchronize(b1->scan_ctx->copy);
        }
        eb_pinned_put(h, pinned_cap);
    }
    eb_set_error("%s", saved.c_str());
}

// Page-locks caller-owned host memory (a shared-memory arena the ranks of one node download their records into), so that
// eb_scan_job_wire_to_host into it is a DMA at link rate; the inverse before the memory is unmapped.
extern "C" int eb_host_register(void* p, uint64_t bytes) {
    if (!p || !bytes) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return EB_OK;
}
extern "C" int eb_host_unregister(void* p) {
    if (!p) { eb_set_error("null argument"); return EB_ERR_ARG; }
    EB_HIP(hipHostUnregister(p));
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
