// eb_main.cpp -- the `ebwt2InDel` command line, same flags / defaults / messages as the reference
// (ebwt2InDel.cpp:76-103 help, :1677-1823 main), driving the GPU hot path through the C ABI only.
// Extra flags (not in the reference): -D <device> selects the GPU (default 0); -G <n> runs on n GPUs of the node:
// ONE process, one host thread per GPU.  The input files come off the disk once (the ranks read the same page-cache pages) and
// are uploaded to every GPU (BWT replicated); phases 2-3 are split by Weiner-link subtrees and combined by an in-place
// ncclAllReduce over xGMI (SUM == OR: one writer per bit, checked by popcounts); phase 4 is split by closing position and every
// rank downloads its own compact call-set image over its OWN PCIe link into one page-locked arena (the node's host memory is
// shared: no link carries another rank's records), from which GPU 0's thread writes the .snp -- the same bytes as -G 1 and as
// the reference.  The RCCL communicators come up FIRST (ncclCommInitAll holds the HIP runtime for seconds: 2.4 s before any
// other HIP work, 5-6 s beside it, scripts/rccl_init_probe.cpp) while the input files are read ahead into the page cache; one
// GPU needs none.  EB_MULTI_BACKEND=rccl-gather: the call sets travel GPU -> GPU (ncclSend / ncclRecv) to GPU 0, which
// downloads all of them (rounds 3-4); =host: both exchanges through host memory, ranks may share devices (rehearsal on a box
// with fewer GPUs).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ebwt2indel.h"

static void help() {
    printf("ebwt2InDel [options]\n"
           "Options:\n"
           "-h          Print this help.\n"
           "-1 <arg>    Input eBWT file (A,C,G,T,#) of first reads set (REQUIRED).\n"
           "-2 <arg>    Input eBWT file (A,C,G,T,#) of second reads set. If not specified, perform genotyping of first reads set.\n"
           "            If specified, find differences (SNPs/indels) between the two reads sets.\n"
           "-d <arg>    Input Document Array. If option -2 is not specified, this file specifies which characters from the input bwt\n"
           "            belong to the first (0) and which from the second (1) individual. Format: ASCII file filled with '0' and '1'.\n"
           "-o <arg>    Output .snp file (REQUIRED).\n"
           "-L <arg>    Length of left-context, SNP included. Default: 31.\n"
           "-R <arg>    Length of right context, SNP excluded. Default: 30.\n"
           "-k <arg>    Minimum LCP required in clusters. Default: 16.\n"
           "-g <arg>    Maximum allowed gap length in indel. Default: 10. If 0, indels are disabled.\n"
           "-v <arg>    Maximum number of non-isolated SNPs in left-contexts (excluding cntral SNP/indel). Default: 2.\n"
           "-m <arg>    Minimum coverage of output events. Default: 3.\n"
           "-c <arg>    Discard events with low-complexity right-context.  Here, low-complexity means that the context starts with a \n"
           "            run of <arg> equal characters. Default: length of right context (-R), minus 10.\n"
           "-q          Maximum number of allowed variants per genomic position in each sample. If 0, there is no limit. Default: 0.\n"
           "-t <arg>    ASCII value of terminator character. Default: 35 (#).\n"
           "-D <arg>    GPU to run on. Default: 0.\n"
           "-G <arg>    Number of GPUs of this node to run on (one host thread per GPU, RCCL). Default: 1.\n\n"
           "\nTo run ebwt2InDel, you must first build the extended Burrows-Wheeler Transform of the input sequences.\n\n"
           "Output format: A fasta file with DNA fragments containing the variations.\n");
    exit(0);
}

static bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int fail(const char* what) {
    // the reference reports problems on stdout and exits with 1 (dna_string.hpp:90-96)
    printf("%s: %s\n", what, eb_last_error());
    return 1;
}

// ------------------------------------------------------------------------------------------------ -G n: n GPUs, one process
namespace {

struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        uint64_t g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

// contiguous slice of closing positions for `rank`; cuts fall on 64-position flag words (ebwt2indel_amd/distributed.py: shard_range)
void shard_range(uint64_t n, int rank, int world, uint64_t* b, uint64_t* e) {
    auto cut = [&](int i) -> uint64_t {
        if (i <= 0) return 0;
        if (i >= world) return n;
        uint64_t c = (uint64_t)((unsigned __int128)n * (unsigned)i / (unsigned)world);
        c = (c + 63) / 64 * 64;
        return c < n ? c : n;
    };
    *b = cut(rank);
    *e = cut(rank + 1);
}

struct MultiJob {
    int world = 1;
    bool host_backend = false; // no RCCL at all: vectors and call sets through host memory
    bool rccl_gather = false;  // call sets GPU -> GPU to rank 0 (instead of every rank downloading its own)
    uint8_t* arena = nullptr;  // page-locked: the ranks' wire images one behind the other (rank 0 allocates it once the lengths are known)
    std::vector<uint64_t> arena_off;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    // inputs: mapped once, read-only; every rank uploads from the same pages (the page cache holds the files once)
    std::string path1, path2, path_da;
    bool two = false;
    double t_comm = 0;         // seconds ncclCommInitAll took (0: no communicator needed)
    eb_params p;
    Barrier* bar = nullptr;
    // per rank
    std::vector<int> rc;
    std::vector<std::string> err;
    std::vector<eb_nav_stats> ns;
    std::vector<eb_scan_stats> ss;
    std::vector<std::array<uint64_t, 4>> ones;
    std::vector<const void*> wire_ptr;
    std::vector<uint64_t> wire_len;
    // host backend: OR of the shards' vectors
    std::mutex mu;
    std::vector<uint64_t> acc[4];
    uint8_t bad_sym = 0;
    uint64_t bad_pos = 0;
    double t_index = 0, t_nav = 0, t_scan = 0;
};

bool any_failed(MultiJob& J) {
    for (int r : J.rc) if (r != EB_OK) return true;
    return false;
}

#define MJ_FAIL(code, ...)                                 \
    do {                                                   \
        char b_[512];                                      \
        snprintf(b_, sizeof b_, __VA_ARGS__);              \
        J.rc[(size_t)r] = (code);                          \
        J.err[(size_t)r] = b_;                             \
    } while (0)
#define MJ_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess && J.rc[(size_t)r] == EB_OK) MJ_FAIL(EB_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define MJ_NCCL(call)                                                                                     \
    do {                                                                                                  \
        ncclResult_t e_ = (call);                                                                         \
        if (e_ != ncclSuccess && J.rc[(size_t)r] == EB_OK) MJ_FAIL(EB_ERR_DEVICE, "%s: %s", #call, ncclGetErrorString(e_)); \
    } while (0)

// A failure INSIDE a collective (ncclAllReduce / ncclSend / ncclRecv or the stream synchronisation behind them) is different
// from every other failure: the peers may already be blocked in the same collective, with no timeout, and would never reach
// the barrier + common check below.  The failing rank therefore prints its error, aborts every communicator (a blocked
// peer's stream then returns) and ends the process non-zero -- no hang, no partial output mistaken for a result.
[[noreturn]] void collective_failure(MultiJob& J, int r) {
    printf("GPU %d: a collective failed: %s\n", J.dev[(size_t)r], J.err[(size_t)r].c_str());
    fflush(stdout);
    for (ncclComm_t c : J.comm)
        if (c) (void)ncclCommAbort(c);
    _exit(1);
}

// One rank = one host thread bound to one GPU.  Every step that can fail is followed by a barrier and a common check, so
// that no rank enters a collective its peers will not join.
void rank_main(MultiJob& J, int r) {
    const int device = J.dev[(size_t)r];
    const bool two = J.two;
    eb_index *b1 = nullptr, *b2 = nullptr;
    eb_flags* fl = nullptr;
    eb_scan_job* job = nullptr;
    eb_calls* calls = nullptr;
    hipStream_t st = nullptr;
    std::vector<void*> recv; // rank 0: device buffers of the other ranks' images
    auto cleanup = [&] {
        if (job) eb_scan_clusters_cancel(job);
        if (calls) eb_calls_free(calls);
        for (void* q : recv) if (q) (void)hipFree(q);
        if (fl) eb_flags_destroy(fl);
        eb_index_destroy(b1);
        eb_index_destroy(b2);
        if (st) (void)hipStreamDestroy(st);
    };
    auto sync_ok = [&]() -> bool { J.bar->wait(); return !any_failed(J); };

    // ---- phase 1: the text is in host memory once; every rank uploads and packs it on its own GPU (PCIe links in parallel)
    double t0 = now_s();
    (void)hipSetDevice(device);
    MJ_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // (through the library's loader: parallel preads of the file into page-locked pieces, uploaded and packed while the next
    // ones fill -- 2-4x faster than copying out of a mapping; the ranks all read the same page-cache pages)
    int rc = eb_index_create_from_file(J.path1.c_str(), (uint8_t)J.p.term, device, &b1);
    if (rc == EB_OK && two) rc = eb_index_create_from_file(J.path2.c_str(), (uint8_t)J.p.term, device, &b2);
    if (rc != EB_OK) {
        if (rc == EB_ERR_ALPHABET) eb_index_bad_symbol(&J.bad_sym, &J.bad_pos); // same symbol on every rank
        MJ_FAIL(rc, "%s", eb_last_error());
    }
    if (!sync_ok()) return cleanup();
    if (r == 0) J.t_index = now_s() - t0;

    // ---- phases 2-3: this rank's Weiner-link subtrees, then the vectors are combined
    t0 = now_s();
    // (the -d file is loaded and packed on a second host thread beside this rank's traversal)
    eb_da_staged* staged = nullptr;
    int rc_da = EB_OK;
    std::string err_da;
    std::thread da_loader;
    if (!J.path_da.empty()) {
        eb_index_info_t info;
        (void)eb_index_info(b1, &info);
        da_loader = std::thread([&, n_bwt = info.n] {
            rc_da = eb_da_stage_file(J.path_da.c_str(), n_bwt, device, &staged);
            if (rc_da != EB_OK) err_da = eb_last_error();
        });
    }
    rc = eb_flags_build_shard(b1, b2, J.p.K, J.p.k_right, (uint32_t)r, (uint32_t)J.world, &fl, &J.ns[(size_t)r]);
    if (da_loader.joinable()) da_loader.join();
    struct StagedGuard { eb_da_staged*& s; ~StagedGuard() { if (s) eb_da_staged_free(s); } } staged_guard{staged};
    if (rc == EB_OK) rc = eb_flags_popcounts(fl, J.ones[(size_t)r].data());
    if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
    else if (rc_da != EB_OK) MJ_FAIL(rc_da, "%s", err_da.c_str());
    if (!sync_ok()) return cleanup();
    const uint64_t n = J.ns[(size_t)r].n;
    uint64_t* v[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t words = 0;
    (void)eb_flags_device_vectors(fl, v, &words);
    if (!J.host_backend) {
        // every bit has ONE writer over all ranks: integer SUM == OR (RCCL has no bitwise OR); in place, in HBM, over xGMI
        // (one rank: the sum of one vector is the vector)
        if (J.world > 1) {
            for (int k = 0; k < 4; ++k)
                if (v[k]) MJ_NCCL(ncclAllReduce(v[k], v[k], words, ncclUint64, ncclSum, J.comm[(size_t)r], st));
            MJ_HIP(hipStreamSynchronize(st));
            if (J.rc[(size_t)r] != EB_OK) collective_failure(J, r);
        }
    } else {
        std::vector<uint64_t> h[4];
        for (int k = 0; k < 4; ++k) if (v[k]) h[k].resize(words);
        rc = eb_flags_export(fl, h[0].data(), h[1].data(), h[2].data(), v[3] ? h[3].data() : nullptr);
        if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
        {
            std::lock_guard<std::mutex> lk(J.mu);
            for (int k = 0; k < 4; ++k) {
                if (!v[k]) continue;
                if (J.acc[k].empty()) J.acc[k].assign(words, 0);
                for (uint64_t w = 0; w < words; ++w) J.acc[k][w] += h[k][w]; // the same SUM the all-reduce computes
            }
        }
        J.bar->wait();
        eb_flags_destroy(fl);
        fl = nullptr;
        if (J.rc[(size_t)r] == EB_OK) {
            rc = eb_flags_import_shard_sum(n, device, J.acc[0].data(), J.acc[1].data(), J.acc[2].data(), J.acc[3].empty() ? nullptr : J.acc[3].data(), &fl);
            if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
        }
    }
    if (J.rc[(size_t)r] == EB_OK) {
        // SUM stood in for OR: popcount(a + b) = popcount(a) + popcount(b) - carries, so the ones must add up exactly
        uint64_t got[4] = {0, 0, 0, 0}, want[4] = {0, 0, 0, 0};
        if (!J.host_backend) rc = eb_flags_popcounts(fl, got); // the reduced vectors, still in shard form
        else for (int k = 0; k < 4; ++k) for (uint64_t w : J.acc[k]) got[k] += (uint64_t)__builtin_popcountll(w);
        for (int q = 0; q < J.world; ++q) for (int k = 0; k < 4; ++k) want[k] += J.ones[(size_t)q][(size_t)k];
        if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
        else if (memcmp(got, want, sizeof got) != 0) MJ_FAIL(EB_ERR_STATE, "sharded traversal: a bit of the flag vectors was set by more than one rank");
        else if (!J.host_backend && (rc = eb_flags_finalize(fl)) != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
    }
    if (J.rc[(size_t)r] == EB_OK && !J.path_da.empty()) {
        eb_da_staged* hand = staged;
        staged = nullptr; // (consumed by the call below)
        rc = eb_flags_set_da_staged(fl, hand);
        if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
    }
    if (!sync_ok()) return cleanup();
    if (r == 0) J.t_nav = now_s() - t0;

    // ---- phase 4: this rank's slice of closing positions; the call sets meet on GPU 0
    t0 = now_s();
    uint64_t begin = 0, end = 0;
    shard_range(n, r, J.world, &begin, &end);
    {
        rc = eb_scan_clusters_begin_ex(b1, b2, fl, begin, end, &J.p, EB_SCAN_DEVICE_ONLY | EB_SCAN_WIRE, &job);
        if (rc == EB_OK) rc = eb_scan_job_device_wire(job, &J.wire_ptr[(size_t)r], &J.wire_len[(size_t)r]); // (waits for the job's kernels)
        if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
        if (!sync_ok()) return cleanup();
        // the lengths are known to everyone (one process): rank 0 lays the images out one behind the other
        if (r == 0) {
            uint64_t total = 0;
            J.arena_off.assign((size_t)J.world, 0);
            for (int q = 0; q < J.world; ++q) { J.arena_off[(size_t)q] = total; total += (J.wire_len[(size_t)q] + 63) & ~63ull; }
            // page-locked and visible to every device of the process: each GPU's DMA engine writes its own part
            MJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&J.arena), total ? total : 64, hipHostMallocPortable));
            if (J.rccl_gather) {
                recv.assign((size_t)J.world, nullptr);
                for (int q = 1; q < J.world; ++q) MJ_HIP(hipMalloc(&recv[(size_t)q], J.wire_len[(size_t)q] ? J.wire_len[(size_t)q] : 1));
            }
        }
        if (!sync_ok()) return cleanup();
        if (!J.rccl_gather) {
            // every rank's records over its own PCIe link (GPU r -> host), side by side
            uint64_t len = 0;
            rc = eb_scan_job_wire_to_host(job, J.arena + J.arena_off[(size_t)r], J.wire_len[(size_t)r], &len);
            if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
        } else {
            // variable-length gather to rank 0: the payloads go GPU -> GPU, rank 0's link then carries all of them
            if (J.world > 1) {
                if (r == 0) {
                    MJ_NCCL(ncclGroupStart());
                    for (int q = 1; q < J.world; ++q) MJ_NCCL(ncclRecv(recv[(size_t)q], J.wire_len[(size_t)q], ncclUint8, q, J.comm[0], st));
                    MJ_NCCL(ncclGroupEnd());
                } else {
                    MJ_NCCL(ncclSend(J.wire_ptr[(size_t)r], J.wire_len[(size_t)r], ncclUint8, 0, J.comm[(size_t)r], st));
                }
                MJ_HIP(hipStreamSynchronize(st));
                if (J.rc[(size_t)r] != EB_OK) collective_failure(J, r);
            }
            if (r == 0)
                for (int q = 0; q < J.world; ++q)
                    if (J.wire_len[(size_t)q])
                        MJ_HIP(hipMemcpy(J.arena + J.arena_off[(size_t)q], q == 0 ? J.wire_ptr[0] : recv[(size_t)q], J.wire_len[(size_t)q], hipMemcpyDeviceToHost));
        }
        J.bar->wait(); // every image is in the arena; the device images may be released
        rc = eb_scan_clusters_end(job, nullptr, &J.ss[(size_t)r]);
        job = nullptr;
        if (rc != EB_OK) MJ_FAIL(rc, "%s", eb_last_error());
    }
    J.bar->wait();
    if (r == 0) J.t_scan = now_s() - t0;
    cleanup();
}

}  // namespace

static void print_report(const eb_nav_stats& ns, const eb_scan_stats& ss, bool two_bwts, bool with_da, uint64_t events, uint64_t cluster_nr) {
    printf("\nDone.\nAnalyzed %llu clusters.\n", (unsigned long long)ss.clusters);
    printf("Average cluster length: %g.\n\n", ss.clusters ? (double)ss.bases_in_runs / (double)ss.clusters : 0.0);
    if (!two_bwts && !with_da) printf("Stored to file %llu events clustered in %llu clusters.\n\n", (unsigned long long)events, (unsigned long long)(cluster_nr - 1));
    printf("Distribution of bases inside clusters (cluster length / number of bases inside clusters of that length): \n\n");
    uint64_t scale = 0;
    for (int i = 0; i <= 200; ++i) scale = ss.hist[i] > scale ? ss.hist[i] : scale;
    // the reference divides by `scale` unguarded and dies with SIGFPE when no cluster closed (:1664-1669); we print an empty histogram
    for (int i = 0; i <= 200; ++i) {
        printf("%d%s", i, i < 10 ? "   " : (i < 100 ? "  " : " "));
        uint64_t bar = scale ? (100 * ss.hist[i]) / scale : 0;
        for (uint64_t j = 0; j < bar; ++j) putchar('-');
        printf(" %llu\n", (unsigned long long)ss.hist[i]);
    }
    if (with_da) printf("\nStored to file %llu sequences clustered in %llu clusters.\n", (unsigned long long)events, (unsigned long long)(cluster_nr - 1));
    (void)ns;
}

// -G n: the whole tool on n GPUs of this node (see the head of this file)
static int run_multi(int world, const std::string& input1, const std::string& input2, const std::string& input_da, const std::string& output,
                     const eb_params& p) {
    const bool two_bwts = !input2.empty(), with_da = !input_da.empty();
    const char* be = getenv("EB_MULTI_BACKEND");
    const bool host_backend = be && !strcmp(be, "host");
    const bool rccl_gather = be && !strcmp(be, "rccl-gather"); // (the vectors' all-reduce is RCCL's either way; =host: none of it)
    const int ndev = eb_device_count();
    if (world < 1 || world > 64) { printf("Error: -G %d: between 1 and 64 GPUs\n", world); return 1; }
    if (!host_backend && world > ndev) {
        printf("Error: -G %d but this node shows %d GPU(s): RCCL needs one GPU per rank (EB_MULTI_BACKEND=host lets ranks share devices)\n", world, ndev);
        return 1;
    }
    MultiJob J;
    J.world = world;
    J.host_backend = host_backend;
    J.rccl_gather = rccl_gather;
    J.p = p;
    for (int r = 0; r < world; ++r) J.dev.push_back(r % ndev);
    printf(two_bwts ? "Phase 1/4: loading and indexing eBWTs on %d GPU(s) ... " : "Phase 1/4: loading and indexing eBWT on %d GPU(s) ... ", world);
    fflush(stdout);
    double t0 = now_s();
    // the document array is read by every rank from the file (eb_flags_set_da_file): one set of pages in the page cache serves all
    if (with_da) {
        int fd = ::open(input_da.c_str(), O_RDONLY);
        if (fd < 0) { printf("Error: could not read file %s\n", input_da.c_str()); return 1; }
        ::close(fd);
    }
    const double t_read = now_s() - t0;
    J.path1 = input1;
    J.path2 = input2;
    J.two = two_bwts;
    J.path_da = input_da;
    if (!host_backend && world > 1) {
        // The communicators first, before this process does any other HIP work: ncclCommInitAll holds the runtime (a kernel launch
        // issued beside it waits until it is done) and takes twice as long when other HIP work runs beside it.  What the host can do
        // meanwhile is bring the input files into the page cache.
        J.comm.resize((size_t)world);
        std::thread readahead([&] {
            for (const std::string* path : {&input1, &input2, &input_da}) {
                if (path->empty()) continue;
                int fd = ::open(path->c_str(), O_RDONLY);
                if (fd < 0) continue;
                struct stat st;
                if (fstat(fd, &st) == 0) {
                    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
                    std::vector<char> buf(8u << 20);
                    while (read(fd, buf.data(), buf.size()) > 0) {} // (pages that are cached already cost a copy: ~5 GB/s)
                }
                ::close(fd);
            }
        });
        const double c0 = now_s();
        const ncclResult_t crc = ncclCommInitAll(J.comm.data(), world, J.dev.data());
        J.t_comm = now_s() - c0;
        readahead.join();
        if (crc != ncclSuccess) { printf("\nError: ncclCommInitAll failed: %s\n", ncclGetErrorString(crc)); return 1; }
    }
    Barrier bar(world);
    J.bar = &bar;
    J.rc.assign((size_t)world, EB_OK);
    J.err.resize((size_t)world);
    J.ns.resize((size_t)world);
    J.ss.resize((size_t)world);
    J.ones.resize((size_t)world);
    J.wire_ptr.assign((size_t)world, nullptr);
    J.wire_len.assign((size_t)world, 0);
    for (auto& s : J.ns) memset(&s, 0, sizeof s);
    for (auto& s : J.ss) memset(&s, 0, sizeof s);
    std::vector<std::thread> th;
    for (int r = 1; r < world; ++r) th.emplace_back(rank_main, std::ref(J), r);
    rank_main(J, 0);
    for (std::thread& t : th) t.join();
    // (the communicators are not torn down one by one: the process is about to end, and ncclCommDestroy takes seconds)
    for (int r = 0; r < world; ++r)
        if (J.rc[(size_t)r] != EB_OK) {
            if (J.rc[(size_t)r] == EB_ERR_ALPHABET) {
                printf("\nError while reading file: read forbidden character '%c' (ASCII code %d).\n"
                       "Only A,C,G,T, and %c are admitted in the input BWT!\n"
                       "If the unknown character is the terminator, you can solve the problem by adding option \"-t %d\".\n",
                       J.bad_sym, (int)(char)J.bad_sym, (char)p.term, (int)(char)J.bad_sym);
            } else {
                printf("\nError on GPU rank %d: %s\n", r, J.err[(size_t)r].c_str());
            }
            return 1;
        }
    printf("done.\n");
    // totals over the ranks: every counter has one writer per suffix-tree record / closing position
    eb_nav_stats ns;
    memset(&ns, 0, sizeof ns);
    eb_scan_stats ss;
    memset(&ss, 0, sizeof ss);
    for (int r = 0; r < world; ++r) {
        const eb_nav_stats& a = J.ns[(size_t)r];
        ns.n = a.n;
        ns.leaves += a.leaves; ns.nodes += a.nodes; ns.lcp_values += a.lcp_values; ns.da_values += a.da_values; ns.minima += a.minima;
        ns.lcp_pos_sum += a.lcp_pos_sum; ns.lcp_pos_sq_sum += a.lcp_pos_sq_sum; // modulo 2^64
        ns.rank_queries += a.rank_queries; ns.batches += a.batches;
        if (a.seconds_leaves > ns.seconds_leaves) ns.seconds_leaves = a.seconds_leaves;
        if (a.seconds_nodes > ns.seconds_nodes) ns.seconds_nodes = a.seconds_nodes;
        const eb_scan_stats& b = J.ss[(size_t)r];
        ss.runs_closed += b.runs_closed; ss.clusters += b.clusters; ss.bases_in_runs += b.bases_in_runs; ss.survivors += b.survivors; ss.chains += b.chains;
        for (int i = 0; i <= 200; ++i) ss.hist[i] += b.hist[i];
        if (b.ms_scan > ss.ms_scan) ss.ms_scan = b.ms_scan;
        if (b.ms_extend > ss.ms_extend) ss.ms_extend = b.ms_extend;
        if (b.ms_right > ss.ms_right) ss.ms_right = b.ms_right;
    }
    if (ns.n > 0 && !eb_nav_covers_every_slot_once(&ns)) {
        // the shards collect the complement of the LCP thresholds, exact only when every position has one writer
        printf("\nError: the traversal wrote %llu of %llu LCP slots: the input is not the eBWT of a read collection; run without -G "
               "(the single-GPU path handles such inputs the reference's way)\n", (unsigned long long)ns.lcp_values, (unsigned long long)ns.n);
        return 1;
    }
    printf(two_bwts ? "\nPhase 2/4: merging eBWTs.\n" : "\nPhase 2/4: navigating suffix tree leaves.\n");
    if (two_bwts) printf("Computed %llu/%llu DA values.\n", (unsigned long long)ns.da_values, (unsigned long long)ns.n);
    printf("Processed %llu suffix-tree leaves.\n\n", (unsigned long long)ns.leaves);
    printf("Phase 3/4: computing LCP minima.\n");
    printf("Computed %llu/%llu LCP values.\n", (unsigned long long)ns.lcp_values, (unsigned long long)ns.n);
    printf("Found %llu LCP minima.\n", (unsigned long long)ns.minima);
    printf("Processed %llu suffix-tree nodes.\n\n", (unsigned long long)ns.nodes);
    printf("Phase 4/4: detecting SNPs and indels.\n");
    printf("Output events will be stored in %s\n", output.c_str());
    // rank 0's host: the ranks' call sets in rank order = BWT order; cluster_nr runs through (ebwt2InDel.cpp:52)
    double t4 = now_s();
    uint64_t cluster_nr = 1, events = 0;
    for (int r = 0; r < world; ++r) {
        eb_calls* calls = nullptr;
        int rc = eb_calls_unpack(J.arena + J.arena_off[(size_t)r], J.wire_len[(size_t)r], &calls);
        if (rc == EB_OK) rc = eb_write_snp(calls, &p, two_bwts || with_da, output.c_str(), r > 0, &cluster_nr, &events);
        if (calls) eb_calls_free(calls);
        if (rc != EB_OK) return fail("Error while writing the output");
    }
    double t5 = now_s();
    print_report(ns, ss, two_bwts, with_da, events, cluster_nr);
    printf("\nTimings (s) on %d GPU(s)%s: communicators %.3f, upload+index %.3f, phases 2-3 %.3f (%llu rank queries, %llu launches over all ranks), "
           "phase 4 + records to host %.3f (slowest rank: scan %.3f ms, extend %.3f ms, right %.3f ms), write %.3f\n",
           world, host_backend ? " [host-memory exchange, records over every rank's own link]" : (world == 1 ? " [one rank: no communicator]" : (rccl_gather ? " [RCCL, records gathered to GPU 0]" : " [RCCL all-reduce, records over every GPU's own link]")),
           J.t_comm + t_read, J.t_index, J.t_nav, (unsigned long long)ns.rank_queries,
           (unsigned long long)ns.batches, J.t_scan, ss.ms_scan, ss.ms_extend, ss.ms_right, t5 - t4);
    if (J.arena) (void)hipHostFree(J.arena);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) help();
    eb_params p;
    memset(&p, 0, sizeof p);
    p.term = '#';
    std::string input1, input2, input_da, output;
    int device = 0, gpus = 0;
    int opt;
    while ((opt = getopt(argc, argv, "h1:2:v:L:R:m:g:k:t:o:d:c:q:D:G:")) != -1) {
        switch (opt) {
            case 'h': help(); break;
            case '1': input1 = optarg; break;
            case 'o': output = optarg; break;
            case '2': input2 = optarg; break;
            case 'd': input_da = optarg; break;
            case 'm': p.mcov_out = atoi(optarg); break;
            case 'k': p.K = atoi(optarg); break;
            case 'g': p.max_gap = atoi(optarg); break;
            case 'L': p.k_left = atoi(optarg); break;
            case 'R': p.k_right = atoi(optarg); break;
            case 'v': p.max_snvs = atoi(optarg); break;
            case 't': p.term = (int32_t)(char)atoi(optarg); break;
            case 'c': p.complexity = atoi(optarg); break;
            case 'q': p.max_variants_per_position = atoi(optarg); break;
            case 'D': device = atoi(optarg); break;
            case 'G': gpus = atoi(optarg); break;
            default: help(); return -1;
        }
    }
    eb_params_resolve(&p); // 0 -> default (ebwt2InDel.cpp:1740-1746)

    if (input1.empty() || output.empty()) help();
    if (!file_exists(input1)) { printf("Error: could not find file %s\n\n", input1.c_str()); help(); }
    if (!input2.empty() && !file_exists(input2)) { printf("Error: could not find file %s\n\n", input2.c_str()); help(); }
    if (!input2.empty() && !input_da.empty()) { printf("Error: Document array (-d) can only be used with one input BWT file (-1)\n\n"); help(); }

    printf("This is ebwt2InDel.\n");
    if (!input2.empty()) printf("Running on two samples. Input eBWT files : %s and %s\n", input1.c_str(), input2.c_str());
    else if (!input_da.empty()) printf("Running on one sample with input Document array. Input eBWT/DA files : %s and %s\n", input1.c_str(), input_da.c_str());
    else printf("Running on one sample (genotyping). Input eBWT file : %s\n", input1.c_str());
    printf("Left-extending eBWT ranges by %d bases.\nRight context length: %d bases.\nComplexity filter: %d\n"
           "Storing output events to file %s\nMinimum coverage of output events: %d\n",
           p.k_left, p.k_right, p.complexity, output.c_str(), p.mcov_out);
    if (p.max_variants_per_position > 0) printf("Maximum number of variants per genomic position per sample: %d\n", p.max_variants_per_position);
    else printf("Maximum number of variants per genomic position per sample: unlimited.\n");
    printf("\n");

    if (eb_device_count() <= 0) {
        printf("Error: no HIP device found. This build of ebwt2InDel runs its hot path on an AMD GPU only.\n");
        return 1;
    }

    const bool two_bwts = !input2.empty();
    const bool with_da = !input_da.empty();
    if (gpus != 0) return run_multi(gpus, input1, input2, input_da, output, p);
    double t0 = now_s();
    printf(two_bwts ? "Phase 1/4: loading and indexing eBWTs ... " : "Phase 1/4: loading and indexing eBWT ... ");
    fflush(stdout);
    eb_index *b1 = nullptr, *b2 = nullptr;
    int rc = eb_index_create_from_file(input1.c_str(), (uint8_t)p.term, device, &b1);
    if (rc == EB_OK && two_bwts) rc = eb_index_create_from_file(input2.c_str(), (uint8_t)p.term, device, &b2);
    if (rc == EB_ERR_ALPHABET) {
        uint8_t sym; uint64_t pos;
        eb_index_bad_symbol(&sym, &pos);
        // dna_string.hpp:91-93
        printf("Error while reading file: read forbidden character '%c' (ASCII code %d).\n"
               "Only A,C,G,T, and %c are admitted in the input BWT!\n"
               "If the unknown character is the terminator, you can solve the problem by adding option \"-t %d\".\n",
               sym, (int)(char)sym, (char)p.term, (int)(char)sym);
        return 1;
    }
    if (rc != EB_OK) return fail("Error while indexing the eBWT");
    printf("done.\n");
    double t1 = now_s();

    printf(two_bwts ? "\nPhase 2/4: merging eBWTs.\n" : "\nPhase 2/4: navigating suffix tree leaves.\n");
    eb_flags* fl = nullptr;
    eb_nav_stats ns;
    // the -d file does not depend on the traversal (the reference reads it between phases 3 and 4, :1497-1508): it is loaded and
    // packed on a second host thread meanwhile
    eb_da_staged* staged = nullptr;
    int rc_da = EB_OK;
    std::string err_da;
    double t_da = 0;
    std::thread da_loader;
    if (with_da) {
        eb_index_info_t info;
        (void)eb_index_info(b1, &info);
        da_loader = std::thread([&, n_bwt = info.n] {
            const double a = now_s();
            rc_da = eb_da_stage_file(input_da.c_str(), n_bwt, device, &staged);
            if (rc_da != EB_OK) err_da = eb_last_error(); // (the message is thread-local)
            t_da = now_s() - a;
        });
    }
    rc = eb_flags_build(b1, b2, p.K, p.k_right, &fl, &ns);
    if (da_loader.joinable()) da_loader.join();
    if (rc != EB_OK) { eb_da_staged_free(staged); return fail("Error while navigating the suffix tree"); }
    if (two_bwts) printf("Computed %llu/%llu DA values.\n", (unsigned long long)ns.da_values, (unsigned long long)ns.n);
    printf("Processed %llu suffix-tree leaves.\n\n", (unsigned long long)ns.leaves);
    printf("Phase 3/4: computing LCP minima.\n");
    printf("Computed %llu/%llu LCP values.\n", (unsigned long long)ns.lcp_values, (unsigned long long)ns.n);
    printf("Found %llu LCP minima.\n", (unsigned long long)ns.minima);
    printf("Processed %llu suffix-tree nodes.\n\n", (unsigned long long)ns.nodes);
    double t2 = now_s();

    if (with_da) {
        if (rc_da == EB_ERR_IO) { printf("Error: could not read file %s\n", input_da.c_str()); return 1; }
        if (rc_da != EB_OK) { printf("Error while loading the document array: %s\n", err_da.c_str()); return 1; }
        rc = eb_flags_set_da_staged(fl, staged);
        if (rc != EB_OK) return fail("Error while loading the document array");
    }
    double t2b = now_s();

    printf("Phase 4/4: detecting SNPs and indels.\n");
    printf("Output events will be stored in %s\n", output.c_str());
    eb_calls* calls = nullptr;
    eb_scan_stats ss;
    rc = eb_scan_clusters(b1, b2, fl, 0, ns.n, &p, &calls, &ss);
    if (rc != EB_OK) return fail("Error while scanning clusters");
    double t3 = now_s();
    uint64_t cluster_nr = 1, events = 0;
    rc = eb_write_snp(calls, &p, two_bwts || with_da, output.c_str(), 0, &cluster_nr, &events);
    if (rc != EB_OK) return fail("Error while writing the output");
    double t4 = now_s();

    print_report(ns, ss, two_bwts, with_da, events, cluster_nr);
    if (with_da) printf("\nDocument array loaded and packed in %.3f s beside the traversal; handed to the flags in %.3f s.", t_da, t2b - t2);
    printf("\nTimings (s): load+index %.3f, phases 2-3 %.3f (leaves %.3f and nodes %.3f side by side, %llu rank queries, %llu launches), "
           "phase 4 device %.3f (scan %.3f ms, extend %.3f ms, right %.3f ms), write %.3f\n",
           t1 - t0, t2 - t1, ns.seconds_leaves, ns.seconds_nodes, (unsigned long long)ns.rank_queries, (unsigned long long)ns.batches,
           t3 - t2b, ss.ms_scan, ss.ms_extend, ss.ms_right, t4 - t3);

    eb_calls_free(calls);
    eb_flags_destroy(fl);
    eb_index_destroy(b1);
    eb_index_destroy(b2);
    return 0;
}
