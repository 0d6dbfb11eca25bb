// eb_navigate.hip -- phases 2-3 on the GPU: LCP-threshold / LCP-minima / DA bitvectors.
//
// Replaces navigate_one_bwt (ebwt2InDel.cpp:555-676) and navigate_two_bwts (:679-831).
// The reference walks suffix-tree leaves and right-maximal nodes depth-first with an explicit
// stack, one record at a time.  Every LCP / DA slot has exactly one writer, so the visiting order
// is free; here the stack lives in HBM and is drained in batches: a launch pops up to `batch`
// records from the top, one lane expands one record (<= 6 parallel_rank queries per BWT), and the
// surviving children are pushed back with one wave-aggregated atomic reservation.  Popping from
// the top keeps the footprint of a depth-first walk; a batch of millions keeps HBM busy with
// independent 64-byte block reads.  Bit writes are atomicOr on 64-bit words and are skipped when
// the value is 0 (vectors start zeroed), i.e. for records shallower than K / k_right.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <type_traits>

#include "eb_config.h"
#include "eb_internal.h"

#include "eb_nav_common.h"
#include "eb_nav_one.h"
#include "eb_nav_two.h"

namespace {

// Sharded traversal: at the first level wide enough, every shard keeps the records whose interval start
// hashes to it -- content-addressed, so all shards agree whatever order their own launches left the level in.
// W2 >= 0: a second word of the record is added to the key (packed two-BWT records: more of the two interval starts).
template <typename Item, int W2>
__global__ void __launch_bounds__(256) k_keep_shard(const NavLaunch* __restrict__ la, uint32_t shard, uint32_t n_shards, NavCounters* cnt) {
    const Item* __restrict__ in = static_cast<const Item*>(la->in);
    Item* __restrict__ out = static_cast<Item*>(la->out);
    const uint64_t count = la->count; // 0 unless k_drain_plan found the level to cut
    const uint64_t total = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < count; base += total) {
        uint64_t i = base + threadIdx.x;
        bool keep = false;
        if (i < count) {
            const uint64_t* w = reinterpret_cast<const uint64_t*>(in + i);
            uint64_t key = w[0];
            if (W2 >= 0) key += w[W2 < 0 ? 0 : W2];
            uint64_t h = key * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            keep = (uint32_t)(h % n_shards) == shard;
        }
        unsigned long long mask = __ballot(keep);
        unsigned long long wbase = 0;
        if (lane == 0 && mask) wbase = atomicAdd(&cnt->pushed, (unsigned long long)__popcll(mask));
        wbase = __shfl(wbase, 0);
        if (keep) {
            uint64_t slot = wbase + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
            const Item v = eb_load_global(in + i);
            const uint4* src = reinterpret_cast<const uint4*>(&v);
            uint4* dst = reinterpret_cast<uint4*>(out + slot);
            for (int k = 0; k < (int)(sizeof(Item) / 16); ++k) eb_store_global(dst + k, src[k]);
        }
    }
}

// after navigation: kr[2w], kr[2w+1] -> thrK[w], thrR[w] (streaming, ~1 ms per 7.5 G positions)
__global__ void __launch_bounds__(256) k_split_kr(const uint4* __restrict__ kr, uint64_t words, uint64_t* __restrict__ thrK, uint64_t* __restrict__ thrR) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = kr[w];
        thrK[w] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        thrR[w] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
}

// after navigation: 64 bytes -> one word of thrK and one of thrR
__global__ void __launch_bounds__(256) k_pack_lcp(const uint4* __restrict__ lcpb, uint64_t words, uint64_t* __restrict__ thrK, uint64_t* __restrict__ thrR) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0, r = 0;
        for (int q = 0; q < 4; ++q) {
            uint4 v = lcpb[w * 4 + q];
            uint32_t d[4] = {v.x, v.y, v.z, v.w};
            for (int t = 0; t < 4; ++t)
                for (int b = 0; b < 4; ++b) {
                    uint32_t byte = (d[t] >> (8 * b)) & 0xFFu;
                    uint32_t pos = q * 16 + t * 4 + b;
                    k |= (uint64_t)(byte & 1u) << pos;
                    r |= (uint64_t)((byte >> 1) & 1u) << pos;
                }
        }
        thrK[w] = k;
        thrR[w] = r;
    }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// 64 * per_quad records per 256-thread workgroup and trip; the grid is capped and strides over the batch
unsigned nav_grid(uint64_t count, uint64_t per_quad = 1) {
    uint64_t tiles = (count + 64 * per_quad - 1) / (64 * per_quad);
    const uint64_t cap = 256ull * 8;
    return (unsigned)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}
unsigned nav_grid_for(uint64_t batch) { return nav_grid(batch, 1); } // per_quad > 1 only lowers the number of tiles: the cap is what large batches see

// ---- the drain's bookkeeping, on the DEVICE.  Round 2 ran it on the host: after every launch the host read back how many
// children it had pushed (8 bytes, a blocking round trip: ~700 of them at 7.55 G positions) and chose the next batch.  Now a
// one-thread kernel does exactly that between two expansions, so the host enqueues eight launches back to back, reads the
// state once per group, and always keeps one group queued ahead of the one it waits for: the device never idles on the host.
enum { DRAIN_NONE = 0, DRAIN_EXPAND = 1, DRAIN_SPLIT = 2 };
enum { DRAIN_ERR_CAPACITY = 1, DRAIN_ERR_OVERFLOW = 2 };
#define EB_NAV_OVERFLOW (-100) /* internal: a launch pushed more records than it was given room for (never leaves this file) */
struct DrainState {
    unsigned long long stack[2];          // device addresses of the two LIFO stacks
    unsigned long long top[2];            // records on each
    unsigned long long cap, batch;        // records a stack holds; most records popped per launch
    unsigned long long launches, max_top; // expansions that had work; most records pending at once
    unsigned long long split_min;
    uint32_t item_bytes;
    uint32_t recent;                      // stack that received records last
    uint32_t last, last_src;              // what the previous step was (DRAIN_*) and which stack it popped
    unsigned long long last_base, last_room;
    uint32_t shard, n_shards, split_done, error;
    unsigned long long records;           // records popped and expanded so far (= chains started by the chain-following kernels)
    unsigned long long room_hook;         // tests (EB_NAV_CHAINS=3): most room a launch is TOLD it has, so that it really overflows (0: the truth)
    NavLaunch expand;                     // what the next expansion kernel works on (count 0: nothing)
    NavLaunch split;                      // ... and the next k_keep_shard (count 0: nothing)
};

// books the children of the previous step
__device__ __forceinline__ void drain_account(DrainState* st, NavCounters* cnt) {
    const unsigned long long pushed = cnt->pushed;
    cnt->pushed = 0;
    cnt->taken = 0;
    if (st->last == DRAIN_EXPAND) {
        const uint32_t src = st->last_src, dst = 1u - src;
        if (pushed > st->last_room) st->error = DRAIN_ERR_OVERFLOW;
        st->top[src] = st->last_base;
        st->top[dst] += pushed;
        if (pushed) st->recent = dst;
    } else if (st->last == DRAIN_SPLIT) {
        const uint32_t src = st->last_src, dst = 1u - src;
        st->top[src] = 0;
        st->top[dst] = pushed;
        st->recent = dst;
        st->split_done = 1;
    }
    st->last = DRAIN_NONE;
    if (st->top[0] + st->top[1] > st->max_top) st->max_top = st->top[0] + st->top[1];
}
__global__ void k_drain_account(DrainState* st, NavCounters* cnt) { drain_account(st, cnt); }
// ... and chooses the next step: TWO stacks alternate; a batch is popped from the top of one and expanded in place (the
// kernel reads it where it lies -- no staging copy), its children are pushed onto the other, which becomes the one to pop from
// next.  Popping the most recently pushed records first keeps a depth-first footprint.
__global__ void k_drain_plan(DrainState* st, NavCounters* cnt) {
    drain_account(st, cnt);
    st->expand.count = 0;
    st->split.count = 0;
    if (st->error || st->top[0] + st->top[1] == 0) return;
    const uint32_t recent = st->recent;
    const uint32_t src = st->top[recent] > 0 ? recent : 1u - recent, dst = 1u - src;
    const unsigned long long ib = st->item_bytes;
    if (st->n_shards > 1 && !st->split_done && st->top[dst] == 0 && st->top[src] >= st->split_min * st->n_shards) {
        // so far every launch expanded one whole level: cut it (k_keep_shard)
        st->split.in = reinterpret_cast<const void*>(st->stack[src]);
        st->split.count = st->top[src];
        st->split.out = reinterpret_cast<void*>(st->stack[dst]);
        st->split.out_cap = st->cap;
        st->last = DRAIN_SPLIT;
        st->last_src = src;
        return;
    }
    const unsigned long long room = st->cap - st->top[dst];
    unsigned long long B = st->top[src] < st->batch ? st->top[src] : st->batch;
    if (4 * B > room) B = room / 4; // a record has at most four children
    if (B == 0) { st->error = DRAIN_ERR_CAPACITY; return; }
    const unsigned long long base = st->top[src] - B;
    st->expand.in = reinterpret_cast<const void*>(st->stack[src] + base * ib);
    st->expand.count = B;
    st->expand.out = reinterpret_cast<void*>(st->stack[dst] + st->top[dst] * ib);
    const unsigned long long handed = st->room_hook && st->room_hook < room ? st->room_hook : room;
    st->expand.out_cap = handed;
    st->expand.write = (st->n_shards == 1 || st->split_done || st->shard == 0) ? 1 : 0;
    st->expand.follow = (st->n_shards == 1 || st->split_done) ? 1 : 0;
    st->last = DRAIN_EXPAND;
    st->last_src = src;
    st->last_base = base;
    st->last_room = handed;
    st->records += B;
    ++st->launches;
}

template <typename Item>
struct Drain {
    // Drains LIFO work stacks in HBM with batched launches.  `launch(const NavLaunch* la, grid)` must append children
    // through NavCounters::pushed.
    int device;
    hipStream_t stream;
    Item* d_stack[2] = {nullptr, nullptr};
    uint64_t cap = 0, batch = 0, top[2] = {0, 0};
    uint64_t launches = 0, max_top = 0;
    // sharded traversal: the tree is walked breadth-first (identically on every shard, bits written by shard 0 only)
    // down to the first level of at least split_min * n_shards records, which is then cut by k_keep_shard
    uint32_t shard = 0, n_shards = 1;
    uint64_t split_min = 4096;
    uint64_t room_hook = 0; // see DrainState

    // The stacks live in one arena shared by both phases (HBM that was released shortly before -- by this process or the one before it -- is wiped by the driver before it is handed out again, ~20 ms per GB, so the arena is sized from
    // the batch, not from n).  A depth-first drain needs a few batches of room (observed: < 7 batches at 7.55 G
    // positions); if a stack ever fills, the planner shrinks the batch -- in the limit a plain depth-first walk that needs
    // 4 x depth records -- so capacity never affects results.
    int init(void* arena, uint64_t arena_bytes, uint64_t n_bound, uint64_t batch_max) {
        batch = batch_max;
        if (batch > n_bound / 2 + 64) batch = n_bound / 2 + 64;
        uint64_t records = arena_bytes / sizeof(Item);
        if (records < 16 * 1024) { eb_set_error("traversal arena too small"); return EB_ERR_NOMEM; }
        cap = records / 2;
        if (batch > cap / 8) batch = cap / 8;
        uint64_t need = n_bound + 4 * batch + 64;   // every pending record is a distinct suffix-tree node/leaf
        if (cap > need) cap = need;
        d_stack[0] = static_cast<Item*>(arena);
        d_stack[1] = d_stack[0] + cap;
        top[0] = top[1] = 0;
        if (eb_cfg().trace_nav) fprintf(stderr, "[nav] 2 stacks x %llu records (%.2f GB) batch=%llu\n", (unsigned long long)cap, 2 * cap * sizeof(Item) / 1e9, (unsigned long long)batch);
        return EB_OK;
    }
    int push_host(const Item& it) {
        EB_HIP(hipMemcpyAsync(d_stack[0] + top[0], &it, sizeof(Item), hipMemcpyHostToDevice, stream));
        EB_HIP(hipStreamSynchronize(stream));
        ++top[0];
        return EB_OK;
    }
    // d_state: sizeof(DrainState) bytes of device memory; h_state: two DrainState in page-locked host memory
    template <int W2, typename Launch>
    int run(NavCounters* d_cnt, DrainState* d_state, DrainState* h_state, Launch launch) {
        const bool trace = eb_cfg().trace_nav;
        const double t_start = now_s();
        const int group = 8; // launches per read-back
        DrainState init;
        memset(&init, 0, sizeof init);
        init.stack[0] = reinterpret_cast<unsigned long long>(d_stack[0]);
        init.stack[1] = reinterpret_cast<unsigned long long>(d_stack[1]);
        init.top[0] = top[0];
        init.top[1] = top[1];
        init.cap = cap;
        init.batch = batch;
        init.split_min = split_min;
        init.item_bytes = (uint32_t)sizeof(Item);
        init.shard = shard;
        init.n_shards = n_shards;
        init.room_hook = room_hook;
        init.max_top = top[0] + top[1];
        h_state[0] = init;
        EB_HIP(hipMemsetAsync(&d_cnt->pushed, 0, sizeof(unsigned long long), stream));
        EB_HIP(hipMemcpyAsync(d_state, &h_state[0], sizeof(DrainState), hipMemcpyHostToDevice, stream));
        EB_HIP(hipStreamSynchronize(stream)); // h_state[0] is reused as a snapshot buffer below
        hipEvent_t ev[2] = {nullptr, nullptr};
        struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } evg{ev};
        EB_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        EB_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        // full-batch grid: the launch does not know its batch (k_drain_plan decides it on the device); workgroups beyond it leave at once
        const unsigned full_grid = nav_grid_for(batch);
        bool split_pending = n_shards > 1;
        auto enqueue_group = [&](int slot) -> int {
            for (int k = 0; k < group; ++k) {
                hipLaunchKernelGGL(k_drain_plan, dim3(1), dim3(1), 0, stream, d_state, d_cnt);
                if (split_pending)
                    hipLaunchKernelGGL((k_keep_shard<Item, W2>), dim3(2048), dim3(256), 0, stream, &d_state->split, shard, n_shards, d_cnt);
                launch(&d_state->expand, full_grid);
            }
            hipLaunchKernelGGL(k_drain_account, dim3(1), dim3(1), 0, stream, d_state, d_cnt);
            EB_HIP(hipGetLastError());
            EB_HIP(hipMemcpyAsync(&h_state[slot], d_state, sizeof(DrainState), hipMemcpyDeviceToHost, stream));
            EB_HIP(hipEventRecord(ev[slot], stream));
            return EB_OK;
        };
        int rc = EB_OK, g = 0;
        if ((rc = enqueue_group(0)) != EB_OK) return rc;
        for (;;) {
            // one group queued ahead of the one waited for
            if ((rc = enqueue_group((g + 1) & 1)) != EB_OK) break;
            EB_HIP(hipEventSynchronize(ev[g & 1]));
            const DrainState& s = h_state[g & 1];
            if (trace)
                fprintf(stderr, "[nav] group %d: launches=%llu records=%llu pending=%llu max=%llu t=%.3fs\n", g, (unsigned long long)s.launches, (unsigned long long)s.records,
                        (unsigned long long)(s.top[0] + s.top[1]), (unsigned long long)s.max_top, now_s() - t_start);
            if (s.split_done) split_pending = false;
            if (s.error || s.top[0] + s.top[1] == 0) {
                // the group queued ahead finds nothing to do (or the same error): wait for it, it still uses the buffers
                EB_HIP(hipEventSynchronize(ev[(g + 1) & 1]));
                const DrainState& e = h_state[(g + 1) & 1];
                top[0] = e.top[0];
                top[1] = e.top[1];
                launches = e.launches;
                max_top = e.max_top;
                if (e.error == DRAIN_ERR_CAPACITY) { eb_set_error("traversal stack capacity exceeded (cap %llu records)", (unsigned long long)cap); return EB_ERR_NOMEM; }
                if (e.error) { eb_set_error("traversal stack overflow"); return EB_NAV_OVERFLOW; }
                return EB_OK;
            }
            ++g;
        }
        (void)hipStreamSynchronize(stream);
        return rc;
    }
};

}  // namespace

// derive: finish the vectors (inversion of the complement form, fused flag word, DA ranks) -- eb_flags_build; false: leave them in
// the form the traversal collected them in, for a caller that still has to combine shards (eb_flags_build_shard)
static int flags_build_impl(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                            eb_nav_stats* stats, bool complement, bool derive, bool chains) {
    if (!b1 || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (n_shards == 0 || shard >= n_shards) { eb_set_error("shard %u of %u", shard, n_shards); return EB_ERR_ARG; }
    if (K < 0 || K > 65534 || k_right < 0 || k_right > 65534) { eb_set_error("K and k_right must be below 65535"); return EB_ERR_ARG; }
    if (b1->n + (b2 ? b2->n : 0) >= (1ull << 40)) { eb_set_error("BWT of 2^40 symbols or more: node records hold 40-bit positions"); return EB_ERR_ARG; }
    if (b2 && b2->device != b1->device) { eb_set_error("both indexes must live on the same device"); return EB_ERR_ARG; }
    *out = nullptr;
    EB_HIP(hipSetDevice(b1->device));
    const uint64_t n = b1->n + (b2 ? b2->n : 0);
    const bool trace_alloc = eb_cfg().trace_nav;
    double ta = now_s();
    auto lap = [&](const char* what) {
        if (!trace_alloc) return;
        double t = now_s();
        fprintf(stderr, "[nav] %s: %.1f ms\n", what, (t - ta) * 1e3);
        ta = t;
    };
    eb_flags* f = nullptr;
    int rc = eb_flags_alloc(n, b1->device, b2 != nullptr, &f);
    if (rc != EB_OK) return rc;
    lap("flag vectors allocated + cleared");
    struct Guard { eb_flags* f; ~Guard() { if (f) eb_flags_destroy(f); } } guard{f};

    // per traversal: the hot counter, the drain's state (device) with two snapshots of it (page-locked host), statistics rows
    struct NavBufs {
        NavCounters* cnt = nullptr;
        DrainState* state = nullptr;
        DrainState* h_state = nullptr;
        unsigned long long* rows = nullptr; // EB_NAV_STATS counters per workgroup (nav_grid() launches at most 2048)
        ~NavBufs() {
            if (cnt) (void)hipFree(cnt);
            if (state) (void)hipFree(state);
            if (h_state) (void)hipHostFree(h_state);
            if (rows) (void)hipFree(rows);
        }
        int alloc() {
            EB_HIP(hipMalloc(&cnt, sizeof(NavCounters)));
            EB_HIP(hipMemset(cnt, 0, sizeof(NavCounters)));
            EB_HIP(hipMalloc(&state, sizeof(DrainState)));
            EB_HIP(hipHostMalloc(&h_state, 2 * sizeof(DrainState)));
            EB_HIP(hipMalloc(&rows, 2048 * EB_NAV_STATS * sizeof(unsigned long long)));
            EB_HIP(hipMemset(rows, 0, 2048 * EB_NAV_STATS * sizeof(unsigned long long)));
            return EB_OK;
        }
    };
    // The two traversals are independent (leaves write leaf interiors, nodes write child borders: disjoint positions), so
    // they run side by side -- the leaves on a second stream driven by a second host thread, with counters, statistics
    // rows and stacks of their own: the short leaf launches fill the gaps of the node launches (0.06 s hidden at config 2).
    NavBufs nb1, nb2;
    if ((rc = nb1.alloc()) != EB_OK || (rc = nb2.alloc()) != EB_OK) return rc;
    hipStream_t st2 = nullptr;
    {
        // The leaf traversal is the short one of the two (a twentieth of the node traversal's vector instructions) but all latency:
        // long chains of dependent block reads.  Beside the node kernels -- which keep the vector ALUs ~90 % busy -- its waves must not
        // also queue for wave slots: highest stream priority.  Two BWTs at 2 x 3.8 G positions: 0.69-0.71 -> 0.58-0.60 s per traversal
        // on the same box (without it the leaves end LAST, 0.06 s behind the nodes); one BWT: indifferent (profiles/r05_sq_nav.txt).
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        EB_HIP(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, prio_hi));
    }
    struct Guard5 { hipStream_t s; ~Guard5() { (void)hipStreamDestroy(s); } } guard5{st2};

    eb_dev_index x1 = b1->dev(), x2 = b2 ? b2->dev() : b1->dev();
    eb_dev_flags df = f->dev();
    NavParams p{K, k_right, 1, nullptr, complement ? 1 : 0};
#ifdef EB_SCAN_EXPERIMENTS
    p.dbg = getenv("EB_NAV_DEBUG") ? (uint32_t)atoi(getenv("EB_NAV_DEBUG")) : 0u; // (profiling build only)
#endif
    const uint64_t split_min = eb_cfg().nav_split_min;
    hipStream_t st = f->stream;
    const uint64_t batch_max = eb_cfg().nav_batch;
    // one arena for both traversals: two stacks of half the arena's batches (EB_NAV_BATCH) of the largest record each
    const uint64_t item_bytes = b2 ? sizeof(Node2P) : sizeof(Node1P); // the largest record of the two traversals
    uint64_t arena_batches = eb_cfg().nav_arena_batches + 1; // observed peak: 5 batches pending
    uint64_t eff_batch = batch_max < n / 2 + 64 ? batch_max : n / 2 + 64;
    uint64_t arena_records = arena_batches * eff_batch + (n < (1ull << 20) ? n + 4 * eff_batch + 64 : 0);
    if (arena_records < (1ull << 16)) arena_records = 1ull << 16;
    uint64_t arena_bytes = arena_records * item_bytes + 65536;
    {
        size_t free_b = 0, total_b = 0;
        EB_HIP(hipMemGetInfo(&free_b, &total_b));
        if (arena_bytes > free_b / 2) arena_bytes = free_b / 2;
    }
    // The traversal's temporaries come from the library's scratch cache and go back to it: a freed multi-GB block is wiped by
    // the driver, and the next large hipMalloc in the process then waits for that wipe (seconds -- measured 3.4 s for
    // the 7.55 GB below right after a previous call's hipFree); recycled blocks cost nothing and phase 4 reuses them.
    const int device = b1->device;
    size_t arena_cap = 0;
    void* d_arena = eb_scratch_get(device, arena_bytes, &arena_cap);
    if (!d_arena) { eb_set_error("traversal arena allocation failed (%llu bytes)", (unsigned long long)arena_bytes); return EB_ERR_NOMEM; }
    struct Guard3 { int dev; void* p; size_t cap; ~Guard3() { eb_scratch_put(dev, p, cap); } } guard3{device, d_arena, arena_cap};
    size_t arena2_cap = 0;
    uint64_t arena2_bytes = arena_bytes / 2; // the leaf frontier stays far smaller than the node frontier
    void* d_arena2 = eb_scratch_get(device, arena2_bytes, &arena2_cap);
    struct Guard6 { int dev; void* p; size_t cap; ~Guard6() { if (p) eb_scratch_put(dev, p, cap); } } guard6{device, d_arena2, arena2_cap};
    const bool side_by_side = d_arena2 != nullptr; // (one after the other in one arena when memory is that short)
    lap("arena allocated");
    // Phase 4 follows: what its first job would pay before its scan can start -- scan planes, device
    // temporaries, a page-locked blob: ~50 ms at 7.55 G positions -- is paid HERE, on a side thread, beside the traversal, when the
    // device has the memory to spare (the traversal's own buffers are allocated by now).
    std::thread warm;
    struct WarmGuard { std::thread& t; ~WarmGuard() { if (t.joinable()) t.join(); } } warm_guard{warm};
    {
        size_t free_b = 0, total_b = 0;
        const uint64_t positions = n / (n_shards ? n_shards : 1); // a shard's rank scans its share of the closing positions afterwards
        f->phase4_positions = positions;
        const bool have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        // (two BWTs, one shard: the merged scan planes are built right behind the traversal -- their memory is taken now)
        if (have_info && b2 && derive && n_shards <= 1 && free_b > 4 * (n / 4)) {
            if (hipMalloc(&f->d_mxy_spare, eb_bit_alloc_words(n) * sizeof(uint4)) != hipSuccess) { f->d_mxy_spare = nullptr; (void)hipGetLastError(); }
            else free_b -= eb_bit_alloc_words(n) * sizeof(uint4);
        }
        if (have_info && free_b > 3 * (n / 4 + positions / 8 + (512ull << 20)))
            warm = std::thread([b1, b2, positions] { eb_phase4_warm(b1, b2, positions); });
    }
    // Threshold bits are collected either as one byte per position (plain stores; needs n bytes of spare HBM) or, when
    // memory is short, as interleaved bit words set with atomics; both are converted to the two bit vectors afterwards.
    uint64_t* d_kr = nullptr;
    uint8_t* d_lcpb = nullptr;
    const size_t lcpb_bytes = eb_bit_alloc_words(n) * 64;
    size_t thr_cap = 0;
    {
        size_t free_b = 0, total_b = 0;
        EB_HIP(hipMemGetInfo(&free_b, &total_b));
        // Complement form (the default): writers are few, so atomicOr on the interleaved bit words costs nothing and the n-byte
        // array, its clearing and its packing pass are saved.  The reference's form (EB_NAV_FORM=bytes / words, and the fallback
        // for inputs that are no eBWT) writes every position: bytes with plain stores when n spare bytes of HBM exist.
        bool want_bytes = !complement && eb_cfg().nav_form != EbConfig::NAV_WORDS && lcpb_bytes + (4ull << 30) < free_b;
        if (want_bytes) d_lcpb = static_cast<uint8_t*>(eb_scratch_get(device, lcpb_bytes, &thr_cap));
        if (!d_lcpb) {
            d_kr = static_cast<uint64_t*>(eb_scratch_get(device, eb_bit_alloc_words(n) * 16, &thr_cap));
            if (!d_kr) { eb_set_error("threshold word allocation failed"); return EB_ERR_NOMEM; }
        }
    }
    struct Guard4 { int dev; void* p; size_t cap; ~Guard4() { eb_scratch_put(dev, p, cap); } } guard4{device, d_lcpb ? (void*)d_lcpb : (void*)d_kr, thr_cap};
    if (d_lcpb) EB_HIP(hipMemsetAsync(d_lcpb, 0, lcpb_bytes, f->stream));
    else EB_HIP(hipMemsetAsync(d_kr, 0, eb_bit_alloc_words(n) * 16, f->stream));
    p.lcpb = d_lcpb;
    if (trace_alloc) { EB_HIP(hipStreamSynchronize(f->stream)); lap("threshold bytes allocated + cleared"); }
    eb_nav_stats s;
    memset(&s, 0, sizeof s);
    s.n = n;
    EB_HIP(hipStreamSynchronize(f->stream)); // the cleared threshold words, before either traversal touches them
    uint64_t launches_l = 0, launches_n = 0;
    const uint64_t room_hook = chains && eb_cfg().nav_chains == 3 ? 64 : 0; // tests: the chain-following walk is told it has room for 64 records per launch
    // phase 2 (ebwt2InDel.cpp:577-615 / :708-756); first_leaf = {[0,F_A), 0} (dna_bwt.hpp:313-317)
    auto phase_leaves = [&](hipStream_t sx, NavBufs& nb, void* arena, uint64_t bytes) -> int {
        int r;
        NavCounters* dc = nb.cnt;
        unsigned long long* rows = nb.rows;
        if (!b2) {
            Drain<Leaf1> d;
            d.device = b1->device; d.stream = sx;
            d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
            if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
            if ((r = d.push_host(Leaf1{0, b1->F[0], 0, 0})) != EB_OK) return r;
            r = d.run<-1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
                if (chains) hipLaunchKernelGGL(k_leaves1c, dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else hipLaunchKernelGGL(k_leaves<1>, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            });
            launches_l = d.launches;
            return r;
        }
        Drain<Leaf2P> d;
        d.device = b1->device; d.stream = sx;
        d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
        if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
        if ((r = d.push_host(pack_rec(Leaf2{0, b1->F[0], 0, b2->F[0], 0, {0, 0, 0}}))) != EB_OK) return r;
        r = d.run<1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
            if (chains) hipLaunchKernelGGL(k_leaves2c, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else hipLaunchKernelGGL(k_leaves<2>, dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
        });
        launches_l = d.launches;
        return r;
    };
    // phase 3 (ebwt2InDel.cpp:622-673 / :764-823); root (dna_bwt.hpp:296-308)
    auto phase_nodes = [&](hipStream_t sx, NavBufs& nb, void* arena, uint64_t bytes) -> int {
        int r;
        NavCounters* dc = nb.cnt;
        unsigned long long* rows = nb.rows;
        if (!b2) {
            Drain<Node1P> d;
            d.device = b1->device; d.stream = sx;
            d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
            if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
            Node1 root{{0, b1->F[0], b1->F[1], b1->F[2], b1->F[3], b1->n}, 0, 0};
            if ((r = d.push_host(pack_rec(root))) != EB_OK) return r;
            r = d.run<-1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
                const bool super_lds = (b1->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS;
                if (chains && super_lds) hipLaunchKernelGGL((k_nodes1c<true>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else if (chains) hipLaunchKernelGGL((k_nodes1c<false>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else if (super_lds) hipLaunchKernelGGL((k_nodes1p<true>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
                else hipLaunchKernelGGL((k_nodes1p<false>), dim3(grid), dim3(256), 0, sx, x1, df, d_kr, p, la, dc, rows);
            });
            launches_n = d.launches;
            return r;
        }
        Drain<Node2P> d;
        d.device = b1->device; d.stream = sx;
        d.shard = shard; d.n_shards = n_shards; d.split_min = split_min; d.room_hook = room_hook;
        if ((r = d.init(arena, bytes, n, batch_max)) != EB_OK) return r;
        Node2 root{{0, b1->F[0], b1->F[1], b1->F[2], b1->F[3], b1->n}, {0, b2->F[0], b2->F[1], b2->F[2], b2->F[3], b2->n}, 0, {0, 0, 0}};
        if ((r = d.push_host(pack_rec(root))) != EB_OK) return r;
        const bool super_fits = (b1->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS && (b2->n >> EB_SUPER_SHIFT) + 1 <= EB_NAV_SUPER_LDS;
        r = d.run<1>(dc, nb.state, nb.h_state, [&](const NavLaunch* la, unsigned grid) {
            if (chains && super_fits) hipLaunchKernelGGL((k_nodes2c<true>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else if (chains) hipLaunchKernelGGL((k_nodes2c<false>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else if (super_fits) hipLaunchKernelGGL((k_nodes2p<true>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
            else hipLaunchKernelGGL((k_nodes2p<false>), dim3(grid), dim3(256), 0, sx, x1, x2, df, d_kr, p, la, dc, rows);
        });
        launches_n = d.launches;
        return r;
    };
    double t0 = now_s();
    if (side_by_side) {
        int rc_l = EB_OK;
        std::string err_l;
        double t_l = 0;
        std::thread leaf_thread([&] {
            (void)hipSetDevice(device);
            rc_l = phase_leaves(st2, nb2, d_arena2, arena2_bytes);
            if (rc_l != EB_OK) err_l = eb_last_error(); // the message is thread-local
            t_l = now_s() - t0;
        });
        rc = phase_nodes(st, nb1, d_arena, arena_bytes);
        s.seconds_nodes = now_s() - t0;
        leaf_thread.join();
        s.seconds_leaves = t_l;
        if (rc == EB_OK && rc_l != EB_OK) { eb_set_error("%s", err_l.c_str()); rc = rc_l; }
        if (rc != EB_OK) return rc;
    } else {
        if ((rc = phase_leaves(st, nb1, d_arena, arena_bytes)) != EB_OK) return rc;
        double t1 = now_s();
        s.seconds_leaves = t1 - t0;
        if ((rc = phase_nodes(st, nb1, d_arena, arena_bytes)) != EB_OK) return rc;
        s.seconds_nodes = now_s() - t1;
    }
    const uint64_t launches = launches_l + launches_n;
    if (d_lcpb) hipLaunchKernelGGL(k_pack_lcp, dim3(4096), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_lcpb), f->words, f->d_thrK, f->d_thrR);
    else hipLaunchKernelGGL(k_split_kr, dim3(2048), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_kr), f->words, f->d_thrK, f->d_thrR);
    lap("traversals");
    EB_HIP(hipStreamSynchronize(st));
    EB_HIP(hipGetLastError());
    lap("threshold bits packed");
    s.seconds_traversal = now_s() - t0;
    f->complemented = complement;
    if (derive && (rc = eb_flags_derive(f)) != EB_OK) return rc; // shards: eb_flags_finalize, after the reduction
    // two BWTs: phase 4 streams their scan planes merged by the document array, which is complete now (a sharded build's is not:
    // its first phase-4 job merges them)
    if (derive && b2 && n_shards <= 1 && f->d_mxy_spare && (rc = eb_flags_merge_planes(f, b1, b2)) != EB_OK) return rc;
    NavCounters c;
    memset(&c, 0, sizeof c);
    {
        std::vector<unsigned long long> rows(2048 * EB_NAV_STATS), rows2(2048 * EB_NAV_STATS);
        EB_HIP(hipMemcpy(rows.data(), nb1.rows, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        EB_HIP(hipMemcpy(rows2.data(), nb2.rows, rows2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < rows.size(); ++k) rows[k] += rows2[k];
        for (size_t r = 0; r < 2048; ++r) {
            c.leaves += rows[r * EB_NAV_STATS + 0];
            c.nodes += rows[r * EB_NAV_STATS + 1];
            c.lcp_values += rows[r * EB_NAV_STATS + 2];
            c.da_values += rows[r * EB_NAV_STATS + 3];
            c.minima += rows[r * EB_NAV_STATS + 4];
            c.ranks += rows[r * EB_NAV_STATS + 5];
            c.pos_sum += rows[r * EB_NAV_STATS + 6];
            c.pos_sq_sum += rows[r * EB_NAV_STATS + 7];
        }
    }
    s.leaves = c.leaves;
    s.nodes = c.nodes;
    s.lcp_values = c.lcp_values + (shard == 0 ? 1 : 0); // the reference starts its counter at 1 (ebwt2InDel.cpp:575)
    s.da_values = c.da_values;
    s.minima = c.minima;
    s.rank_queries = c.ranks;
    s.lcp_pos_sum = c.pos_sum;
    s.lcp_pos_sq_sum = c.pos_sq_sum;
    s.batches = launches;
    if (stats) *stats = s;
    guard.f = nullptr;
    *out = f;
    return EB_OK;
}

// Host-side round trip of the packed traversal records (the same __host__ __device__ code the kernels run): lets a
// test without a GPU exercise 40-bit positions beyond 2^32 and the saturating depth.
extern "C" int eb_selftest_pack_fields(int32_t n_fields, const uint64_t* fields, uint64_t depth, uint64_t* fields_out, uint64_t* depth_out,
                                       uint64_t* words_out, uint32_t* n_words) {
    if (!fields || !fields_out || !depth_out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    auto go = [&](auto packed, auto nf_tag) {
        constexpr int NF = decltype(nf_tag)::value;
        packed = pack_fields<NF>(fields, depth);
        *depth_out = unpack_fields<NF>(packed, fields_out);
        const uint32_t nw = (uint32_t)(sizeof(packed) / 8);
        if (n_words) *n_words = nw;
        if (words_out) memcpy(words_out, packed.w, sizeof(packed));
    };
    if (n_fields == 6) go(Node1P{}, std::integral_constant<int, 6>{});
    else if (n_fields == 12) go(Node2P{}, std::integral_constant<int, 12>{});
    else if (n_fields == 4) go(Leaf2P{}, std::integral_constant<int, 4>{});
    else { eb_set_error("record kinds have 4, 6 or 12 position fields"); return EB_ERR_ARG; }
    return EB_OK;
}

extern "C" int eb_nav_covers_every_slot_once(const eb_nav_stats* s) {
    if (!s) return 0;
    // positions 1 .. n-1, each exactly once: their number (the reference's counter starts at 1, ebwt2InDel.cpp:575), their
    // sum and the sum of their squares, modulo 2^64
    return s->lcp_values == s->n && (s->n == 0 || (s->lcp_pos_sum == eb_tri(s->n) && s->lcp_pos_sq_sum == eb_sq(s->n)));
}

// One-BWT node traversal with chains followed in registers (k_nodes1c; EB_NAV_CHAINS=0: k_nodes1p).  A launch of k_nodes1c may
// push more records than the planner left room for (it never has: see the kernel); the traversal then ends with an error and is
// walked again the round-3 way.
static int flags_build_chains(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                              eb_nav_stats* stats, bool complement, bool derive) {
    // EB_NAV_CHAINS (tests): 0 the fallback's kernels at once; 2 the first walk's result is thrown away as if it had overflowed; 3 the
    // first walk is told its stacks have room for 64 records per launch, so that it really overflows (records dropped in mid-launch,
    // error set on the device, the group queued ahead finding nothing to do, the leaf thread joined)
    const int mode = eb_cfg().nav_chains;
    const bool chains = mode != 0;
    int rc = flags_build_impl(b1, b2, K, k_right, shard, n_shards, out, stats, complement, derive, chains);
    if (mode == 2 && rc == EB_OK) { eb_flags_destroy(*out); *out = nullptr; rc = EB_NAV_OVERFLOW; }
    // only an overflow of the chain-following walk is answered by the other kernels; anything else (memory, a device error) would
    // fail the same way again
    if (chains && rc == EB_NAV_OVERFLOW) rc = flags_build_impl(b1, b2, K, k_right, shard, n_shards, out, stats, complement, derive, false);
    return rc == EB_NAV_OVERFLOW ? EB_ERR_STATE : rc;
}

extern "C" int eb_flags_build(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, eb_flags** out, eb_nav_stats* stats) {
    if (!out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    const bool complement = eb_cfg().nav_form == EbConfig::NAV_COMPLEMENT;
    eb_nav_stats s;
    int rc = flags_build_chains(b1, b2, K, k_right, 0, 1, out, &s, complement, true);
    if (rc != EB_OK) return rc;
    if (complement && !eb_nav_covers_every_slot_once(&s)) {
        // Not every LCP slot was written exactly once: the input is not the eBWT of a read collection.  The reference leaves
        // unwritten slots at 0 (vector<bool>(n,false), ebwt2InDel.cpp:571), which the complement form cannot know: walk again
        // the reference's way (writers at or above the thresholds set their bits).
        eb_flags_destroy(*out);
        *out = nullptr;
        rc = flags_build_chains(b1, b2, K, k_right, 0, 1, out, &s, false, true);
        if (rc != EB_OK) return rc;
    }
    if (stats) *stats = s;
    return EB_OK;
}

extern "C" int eb_flags_build_shard(eb_index* b1, eb_index* b2, int32_t K, int32_t k_right, uint32_t shard, uint32_t n_shards, eb_flags** out,
                                    eb_nav_stats* stats) {
    // shards collect the complement (see NavParams::below): their vectors still add up (one writer per marked bit), and
    // eb_flags_finalize inverts the sum.  Callers check that the shards' lcp_values add up to n (a valid eBWT).
    // ALWAYS shard form, never derived -- also for n_shards == 1: the callers' next step (all-reduce + eb_flags_finalize, or
    // eb_flags_export + eb_flags_import_shard_sum) is the same for any number of shards, and a one-shard build that came
    // back finished would be inverted a second time on the host-exchange paths (round-2 advisor finding).
    return flags_build_chains(b1, b2, K, k_right, shard, n_shards, out, stats, true, false);
}
