// eb_writer.cpp -- host side of the hot path: call-set container, event logic, KisSNP2 text.
//
// Replaces to_file (ebwt2InDel.cpp:1254-1330 one sample; :1149-1252 pairs), distance (:192-240),
// dH (:157-171), has_run (:144-152) and event_type (:1102-1144).  This part stays on the host in
// the reference's own language (C++): it is sequential by construction (running cluster_nr / id_nr
// counters, :52,:72) and touches ~1e-4 of the BWT positions.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "eb_config.h"
#include "eb_internal.h"

namespace {

struct Ctx {
    std::string s;
    int support;
};

// true iff s starts with a run of >= k equal characters (:144-152)
bool has_run(const std::string& s, int k) {
    if ((size_t)k > s.length()) return false; // int -> size_t like the reference: a negative k never matches
    for (int i = 1; i < k; ++i)
        if (s[i] != s[i - 1]) return false;
    return true;
}

// Hamming distance, strings aligned on the right, extra characters on the left ignored (:157-171)
int hamming_right(const std::string& a, const std::string& b) {
    size_t len = std::min(a.size(), b.size());
    int d = 0;
    for (size_t i = 0; i < len; ++i) d += a[a.size() - i - 1] != b[b.size() - i - 1];
    return d;
}

struct Align {
    int mism; // mismatches outside the gap
    int gap;  // > 0: insertion in a, < 0: insertion in b, 0: none
};

// distance (:192-240): best of {no indel, i = 1..max_gap characters cut from the right end of a, same for b};
// an indel wins ties against "no indel", "insert in b" wins ties against "insert in a", shortest gap first.
Align align_pair(const std::string& a, const std::string& b, int max_gap) {
    int none = hamming_right(a, b);
    if (max_gap == 0) return {none, 0};
    std::vector<int> cut_a(max_gap), cut_b(max_gap);
    for (int i = 1; i <= max_gap; ++i) {
        // substr(0, len - i): the count wraps for i > len and substr clamps it to the whole string
        std::string a1 = a.substr(0, a.length() - (size_t)i);
        std::string b1 = b.substr(0, b.length() - (size_t)i);
        cut_a[i - 1] = hamming_right(a1, b) + i;
        cut_b[i - 1] = hamming_right(a, b1) + i;
    }
    int ia = (int)(std::min_element(cut_a.begin(), cut_a.end()) - cut_a.begin());
    int ib = (int)(std::min_element(cut_b.begin(), cut_b.end()) - cut_b.begin());
    if (none < cut_a[ia] && none < cut_b[ib]) return {none, 0};
    if (cut_a[ia] < cut_b[ib]) return {cut_a[ia] - (ia + 1), ia + 1};
    return {cut_b[ib] - (ib + 1), -(ib + 1)};
}

// event_type (:1102-1144)
std::string event_label(const std::string& a, const std::string& b, int gap) {
    std::string out = "type:";
    out += gap != 0 ? "_INDEL_event:" : "_SNP_event:";
    if (gap == 0) {
        out += a[a.size() - 1];
        out += '/';
        out += b[b.size() - 1];
    } else if (gap > 0) {
        out += a.substr(a.size() - (size_t)gap);
        out += '/';
    } else {
        out += '/';
        out += b.substr(b.size() - (size_t)(-gap));
    }
    return out;
}

// Output of one contiguous piece of the call set.  The text lacks the cluster numbers: `marks` says where they go and
// which number, counted from the piece's first cluster, so that pieces can be formatted in parallel and numbered
// afterwards (cluster_nr runs over the whole file, ebwt2InDel.cpp:52).
struct Piece {
    std::string text;
    std::vector<std::pair<size_t, uint64_t>> marks; // (offset into text, cluster_nr relative to the piece)
    uint64_t clusters = 0, events = 0;               // cluster_nr increments / events of the piece
};

void header(Piece& out, uint64_t cluster_nr, uint64_t id_nr, size_t right_len, int cov, const std::string& type) {
    std::string& o = out.text;
    o += ">cluster:";
    out.marks.emplace_back(o.size(), cluster_nr);
    o += "_id:";
    o += std::to_string(id_nr);
    o += "_right:";
    o += std::to_string(right_len);
    o += "_cov:";
    o += std::to_string(cov);
    o += '_';
    o += type;
    o += '\n';
}

// one-sample writer (:1254-1330)
void write_single(Piece& out, const std::vector<Ctx>& v, const std::string& right, const eb_params& p) {
    std::string& o = out.text;
    uint64_t& cluster_nr = out.clusters;
    uint64_t& events = out.events;
    if (v.size() < 2) return; // :1256 -- cluster_nr untouched
    int max_dist = 0;
    std::vector<const Ctx*> good;
    for (size_t i = 0; i + 1 < v.size(); ++i) { // consecutive contexts in A,C,G,T order, low-support ones included (:1262-1270)
        Align d = align_pair(v[i].s, v[i + 1].s, p.max_gap);
        max_dist = std::max(max_dist, d.mism);
        if (v[i].support >= p.mcov_out) good.push_back(&v[i]);
    }
    if (v.back().support >= p.mcov_out) good.push_back(&v.back());
    if (max_dist <= p.max_snvs && good.size() >= 2) {
        uint64_t id_nr = 1;
        for (size_t i = 0; i < good.size(); ++i) {
            if (has_run(right, p.complexity)) continue;
            // allele 0 is labelled against allele 1; allele i >= 1 carries distance(good[i-1], good[1]) (:1299-1307)
            const std::string& a = i == 0 ? good[0]->s : good[i - 1]->s;
            Align d = align_pair(a, good[1]->s, p.max_gap);
            header(out, cluster_nr, id_nr++, right.size(), good[i]->support, event_label(a, good[1]->s, d.gap));
            o += good[i]->s;
            o += right;
            o += '\n';
            ++events;
        }
    }
    ++cluster_nr; // even when nothing was printed (:1328)
}

// pair writer (:1149-1252)
void write_pairs(Piece& out, const std::vector<Ctx>& c0, const std::vector<Ctx>& c1, const std::string& right, const eb_params& p) {
    std::string& o = out.text;
    uint64_t& cluster_nr = out.clusters;
    bool found = false;
    uint64_t id_nr = 1;
    for (const Ctx& a : c0)
        for (const Ctx& b : c1) {
            if (a.s.empty() || b.s.empty()) continue;
            if (a.s.back() == b.s.back()) continue; // only pairs whose last characters differ are variants (:921,:1083)
            Align d = align_pair(a.s, b.s, p.max_gap);
            if (has_run(right, p.complexity) || d.mism > p.max_snvs || a.support < p.mcov_out || b.support < p.mcov_out) continue;
            found = true;
            std::string type = event_label(a.s, b.s, d.gap);
            header(out, cluster_nr, id_nr, right.size(), a.support, type);
            o += d.gap < 0 ? a.s.substr((size_t)(-d.gap)) : a.s; // insertion in sample 1: sample 0 loses its first |gap| chars (:1197-1201)
            o += right;
            o += '\n';
            header(out, cluster_nr, id_nr, right.size(), b.support, type);
            o += d.gap > 0 ? b.s.substr((size_t)d.gap) : b.s; // (:1231-1233)
            o += right;
            o += '\n';
            ++id_nr;
        }
    cluster_nr += found ? 1 : 0; // :1250
}

std::string unpack_bases(const uint64_t* words, uint32_t len, bool reversed) {
    std::string s(len, 'A');
    for (uint32_t e = 0; e < len; ++e) {
        char ch = "ACGT"[(words[e >> 5] >> (2 * (e & 31))) & 3];
        s[reversed ? len - 1 - e : e] = ch;
    }
    return s;
}

// clusters [i0, i1) of the call set -> one piece
int format_range(const eb_calls* c, const eb_params* p, int two_samples, uint64_t i0, uint64_t i1, Piece& out) {
    const uint64_t lw = c->left_words(), rw = c->right_words();
    for (uint64_t i = i0; i < i1; ++i) {
        const eb_cluster& r = c->clusters[i];
        if (!r.right_found) continue; // no variants are collected without a right context (:985, :909, :1071)
        std::string right = unpack_bases(c->right + i * rw, r.right_len, false);
        std::vector<Ctx> side[2];
        for (uint32_t s = 0; s < r.n_slots; ++s) {
            uint64_t ch = (uint64_t)r.first_chain + s;
            if (ch >= c->n_chains) return EB_ERR_ARG;
            const eb_chain& k = c->chains[ch];
            if (k.len != (uint32_t)c->k_left) continue; // shorter contexts are dropped (:317)
            Ctx x;
            x.s = unpack_bases(c->left + ch * lw, k.len, true); // extension order -> text order (reverse, :315)
            x.support = (int)k.support;
            side[k.side ? 1 : 0].push_back(std::move(x));
        }
        const size_t before = out.text.size();
        if (two_samples) write_pairs(out, side[0], side[1], right, *p);
        else write_single(out, side[0], right, *p);
        if ((r.reserved & EB_RIGHT_SKIPPED) && out.text.size() != before) return EB_ERR_STATE; // a skipped right context was needed
    }
    return EB_OK;
}

// Event logic + text for the whole call set.  Large sets are cut into pieces formatted by host threads; a piece's text lacks only
// the running cluster numbers, which depend on the pieces before it: once all pieces exist their numbers and final sizes are known,
// and every thread puts its piece where it belongs -- into one buffer (eb_format_snp) or straight into the file at its own offset
// (eb_write_snp: no joined copy of the text at all).  The text is the same whatever the number of threads.
unsigned decimal_digits(uint64_t v) {
    unsigned d = 1;
    while (v >= 10) { v /= 10; ++d; }
    return d;
}

struct Formatted {
    std::vector<Piece> pieces;
    std::vector<uint64_t> base_nr;   // cluster number of a piece's first cluster
    std::vector<uint64_t> offset;    // where its final text starts; offset[pieces.size()] = total length
    unsigned threads = 1;
    uint64_t total() const { return offset.empty() ? 0 : offset.back(); }
    // final text of piece t -> dst (offset[t + 1] - offset[t] bytes)
    void emit(size_t t, char* dst) const {
        const Piece& q = pieces[t];
        size_t from = 0;
        char num[24];
        for (const auto& mk : q.marks) {
            memcpy(dst, q.text.data() + from, mk.first - from);
            dst += mk.first - from;
            int len = snprintf(num, sizeof num, "%llu", (unsigned long long)(base_nr[t] + mk.second));
            memcpy(dst, num, (size_t)len);
            dst += len;
            from = mk.first;
        }
        memcpy(dst, q.text.data() + from, q.text.size() - from);
    }
    template <typename F> void for_each_piece(F f) const {
        if (threads <= 1) { for (size_t t = 0; t < pieces.size(); ++t) f(t); return; }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < pieces.size(); ++t) pool.emplace_back(f, t);
        for (std::thread& th : pool) th.join();
    }
};

int format_calls(const eb_calls* c, const eb_params* p, int two_samples, uint64_t* cluster_nr, uint64_t* events, Formatted& F) {
    if (!c || !p || !cluster_nr) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (c->k_left != p->k_left || c->k_right != p->k_right) { eb_set_error("call set was produced with other -L/-R values"); return EB_ERR_ARG; }
    unsigned threads = 1;
    if (eb_cfg().writer_threads > 0) threads = (unsigned)eb_cfg().writer_threads;                  // EB_WRITER_THREADS (tests: any split)
    else if (c->count >= 50000) threads = std::min(std::thread::hardware_concurrency(), 32u);     // ~1 us per cluster: worth it from here
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > c->count) threads = c->count ? (unsigned)c->count : 1u;
    F.threads = threads;
    F.pieces.assign(threads, Piece());
    std::vector<int> rcs(threads, EB_OK);
    F.for_each_piece([&](size_t t) {
        uint64_t i0 = c->count * t / threads, i1 = c->count * (t + 1) / threads;
        rcs[t] = format_range(c, p, two_samples, i0, i1, F.pieces[t]);
    });
    for (int rc : rcs)
        if (rc != EB_OK) {
            eb_set_error(rc == EB_ERR_STATE ? "corrupt call set: a cluster marked EB_RIGHT_SKIPPED would print" : "corrupt call set: chain index out of range");
            return rc;
        }
    F.base_nr.assign(threads, 0);
    uint64_t ev = events ? *events : 0;
    for (unsigned t = 0; t < threads; ++t) {
        F.base_nr[t] = *cluster_nr;
        *cluster_nr += F.pieces[t].clusters;
        ev += F.pieces[t].events;
    }
    if (events) *events = ev;
    std::vector<uint64_t> size(threads, 0);
    F.for_each_piece([&](size_t t) {
        const Piece& q = F.pieces[t];
        uint64_t z = q.text.size();
        for (const auto& mk : q.marks) z += decimal_digits(F.base_nr[t] + mk.second);
        size[t] = z;
    });
    F.offset.assign(threads + 1, 0);
    for (unsigned t = 0; t < threads; ++t) F.offset[t + 1] = F.offset[t] + size[t];
    return EB_OK;
}

}  // namespace

// ------------------------------------------------------------------ call-set container

static const uint64_t kMagic = EB_CALLS_MAGIC;

namespace {
// page-locked blobs are recycled: pinning tens of MB per call would cost more than the copy it speeds up
std::mutex g_pool_mu;
std::vector<std::pair<void*, uint64_t>> g_pool;

}  // namespace
void* eb_pinned_get(uint64_t size, uint64_t* got) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].second >= size && g_pool[i].second <= 2 * size + (1u << 20)) {
                void* p = g_pool[i].first;
                *got = g_pool[i].second;
                g_pool.erase(g_pool.begin() + i);
                return p;
            }
    }
    void* p = nullptr;
    uint64_t cap = size + size / 4 + 4096;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (eb_cfg().trace_alloc) fprintf(stderr, "[calls] pinned a new block of %.1f MB (none of that size in the pool)\n", cap / 1e6);
    *got = cap;
    return p;
}
void eb_pinned_put(void* p, uint64_t cap) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 16) { g_pool.emplace_back(p, cap); return; }
    (void)hipHostFree(p);
}
namespace {
uint64_t align8(uint64_t v) { return (v + 7) & ~7ull; }
}  // namespace

// A caller that keeps phase-4 jobs coming (EB_SCAN_PIPELINED) has several call sets alive at once -- begun, being handed over,
// being read -- and each needs a page-locked blob.  Pinning one takes ~6 ms per 80 MB; and the first time a device-to-host copy is
// issued while another one is still running, the host sits 12-28 ms inside hipMemcpyAsync (EB_TRACE=begin, round 3: one begin
// call of 19-30 ms among 2.9 ms ones, at a different step of every run, not under the profiler where copies are kernels): the
// runtime opens a further copy-engine queue on first need.  Neither may land in the middle of the pipeline: makes sure `count`
// blobs of at least `size` bytes are in the pool, and runs throw-away copies from `d_warm` into the new ones, four at a time on
// four streams, so that the queues a pipelined caller will need exist before its second call.
void eb_pinned_reserve(uint64_t size, int count, const void* d_warm, uint64_t warm_bytes) {
    std::vector<std::pair<void*, uint64_t>> fresh;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (const auto& e : g_pool)
            if (e.second >= size && e.second <= 2 * size + (1u << 20)) --count;
    }
    for (; count > 0; --count) {
        void* p = nullptr;
        const uint64_t cap = size + size / 4 + 4096;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) break;
        fresh.emplace_back(p, cap);
    }
    if (d_warm && warm_bytes && !fresh.empty()) {
        hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
        bool ok = true;
        for (auto& s : st) ok = ok && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
        if (ok)
            for (size_t i = 0; i < fresh.size(); ++i)
                (void)hipMemcpyAsync(fresh[i].first, d_warm, warm_bytes < fresh[i].second ? warm_bytes : fresh[i].second, hipMemcpyDeviceToHost, st[i % 4]);
        for (auto& s : st)
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    }
    for (const auto& e : fresh) eb_pinned_put(e.first, e.second);
}

namespace {

void bind_sections(eb_calls* c) {
    uint8_t* q = c->blob + sizeof(eb_calls_header);
    c->clusters = reinterpret_cast<eb_cluster*>(q);
    q += c->count * sizeof(eb_cluster);
    c->chains = reinterpret_cast<eb_chain*>(q);
    q += align8(c->n_chains * sizeof(eb_chain));
    c->left = reinterpret_cast<uint64_t*>(q);
    q += c->n_chains * c->left_words() * 8;
    c->right = reinterpret_cast<uint64_t*>(q);
}
}  // namespace

struct eb_calls_storage { uint64_t cap; }; // capacity of a pinned blob is kept in front of the handle

uint64_t eb_calls_blob_size(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right) {
    uint64_t lw = ((uint64_t)k_left + 31) / 32, rw = ((uint64_t)k_right + 31) / 32;
    return sizeof(eb_calls_header) + count * sizeof(eb_cluster) + align8(n_chains * sizeof(eb_chain)) + n_chains * lw * 8 + count * rw * 8;
}

static std::vector<std::pair<eb_calls*, uint64_t>> g_caps; // pinned capacity per live handle
static std::mutex g_caps_mu;

eb_calls* eb_calls_alloc(uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right, bool pinned) {
    eb_calls* c = new eb_calls();
    c->count = count;
    c->n_chains = n_chains;
    c->k_left = k_left;
    c->k_right = k_right;
    c->size = eb_calls_blob_size(count, n_chains, k_left, k_right);
    c->pinned = pinned;
    if (pinned) {
        uint64_t cap = 0;
        c->blob = static_cast<uint8_t*>(eb_pinned_get(c->size, &cap));
        if (c->blob) {
            std::lock_guard<std::mutex> lk(g_caps_mu);
            g_caps.emplace_back(c, cap);
        }
    } else {
        c->blob = static_cast<uint8_t*>(malloc(c->size));
    }
    if (!c->blob) { eb_set_error("out of host memory for %llu call records", (unsigned long long)count); delete c; return nullptr; }
    if (!pinned) memset(c->blob, 0, c->size); // a page-locked blob is overwritten section by section by device copies
    eb_calls_header h = {kMagic, count, n_chains, (uint64_t)(uint32_t)k_left, (uint64_t)(uint32_t)k_right, {0, 0, 0}};
    memcpy(c->blob, &h, sizeof h);
    bind_sections(c);
    if (pinned && (n_chains * sizeof(eb_chain)) % 8) memset(reinterpret_cast<uint8_t*>(c->chains) + n_chains * sizeof(eb_chain), 0, 8 - (n_chains * sizeof(eb_chain)) % 8);
    return c;
}

extern "C" int eb_calls_get(const eb_calls* c, eb_calls_view* v) {
    if (!c || !v) { eb_set_error("null argument"); return EB_ERR_ARG; }
    v->count = c->count;
    v->n_chains = c->n_chains;
    v->k_left = c->k_left;
    v->k_right = c->k_right;
    v->clusters = c->clusters;
    v->chains = c->chains;
    v->left = c->left;
    v->right = c->right;
    return EB_OK;
}

extern "C" void eb_calls_free(eb_calls* c) {
    if (!c) return;
    if (c->blob) {
        if (c->pinned) {
            uint64_t cap = 0;
            {
                std::lock_guard<std::mutex> lk(g_caps_mu);
                for (size_t i = 0; i < g_caps.size(); ++i)
                    if (g_caps[i].first == c) { cap = g_caps[i].second; g_caps.erase(g_caps.begin() + i); break; }
            }
            eb_pinned_put(c->blob, cap);
        } else {
            free(c->blob);
        }
    }
    delete c;
}

extern "C" uint64_t eb_calls_packed_size(const eb_calls* c) { return c ? c->size : 0; }

extern "C" int eb_calls_pack(const eb_calls* c, uint8_t* buf, uint64_t cap) {
    if (!c || !buf) { eb_set_error("null argument"); return EB_ERR_ARG; }
    if (cap < c->size) { eb_set_error("buffer too small: need %llu bytes", (unsigned long long)c->size); return EB_ERR_ARG; }
    memcpy(buf, c->blob, c->size);
    return EB_OK;
}

extern "C" int eb_calls_blob(const eb_calls* c, const uint8_t** data, uint64_t* len) {
    if (!c || !data || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *data = c->blob;
    *len = c->size;
    return EB_OK;
}

extern "C" int eb_calls_unpack(const uint8_t* buf, uint64_t len, eb_calls** out) {
    if (!buf || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    *out = nullptr;
    eb_calls_header h;
    if (len < sizeof h) { eb_set_error("truncated call set"); return EB_ERR_ARG; }
    memcpy(&h, buf, sizeof h);
    if (h.magic != kMagic && h.magic != EB_WIRE_MAGIC && h.magic != EB_WIRE4_MAGIC) { eb_set_error("not a packed call set"); return EB_ERR_ARG; }
    if (h.k_left == 0 || h.k_left > (1u << 20) || h.k_right == 0 || h.k_right > (1u << 20) || h.count > (1ull << 40) || h.n_chains > h.count * EB_MAX_SLOTS) {
        eb_set_error("corrupt call set header");
        return EB_ERR_ARG;
    }
    const bool wire = h.magic == EB_WIRE_MAGIC, wire4 = h.magic == EB_WIRE4_MAGIC;
    uint64_t need = wire4 ? eb_wire4_size(h.count, h.n_chains, (int32_t)h.k_left, (int32_t)h.k_right)
                  : wire  ? eb_wire_size(h.count, h.n_chains, (int32_t)h.k_left, (int32_t)h.k_right)
                          : eb_calls_blob_size(h.count, h.n_chains, (int32_t)h.k_left, (int32_t)h.k_right);
    if (len < need) { eb_set_error("truncated call set"); return EB_ERR_ARG; }
    eb_calls* c = eb_calls_alloc(h.count, h.n_chains, (int32_t)h.k_left, (int32_t)h.k_right, false);
    if (!c) return EB_ERR_NOMEM;
    if (wire4) {
        const uint8_t* q = buf + sizeof h;
        uint64_t chain = 0;
        for (uint64_t i = 0; i < h.count; ++i) {
            eb_wire4_cluster w;
            memcpy(&w, q + i * sizeof w, sizeof w);
            eb_cluster& r = c->clusters[i];
            r.begin = (uint64_t)w.begin_lo | ((uint64_t)w.begin_hi << 32);
            r.end = r.begin + w.len;
            r.first_chain = (uint32_t)chain;
            r.n_slots = w.slots_found & 15u;
            r.right_found = (w.slots_found >> 4) & 1u;
            r.right_len = w.right_len;
            r.reserved = ((w.slots_found >> 5) & 1u) ? EB_RIGHT_SKIPPED : 0u;
            chain += r.n_slots;
        }
        if (chain != h.n_chains) { eb_set_error("corrupt call set: chain counts do not add up"); eb_calls_free(c); return EB_ERR_ARG; }
        q += h.count * sizeof(eb_wire4_cluster);
        for (uint64_t i = 0; i < h.n_chains; ++i) {
            eb_wire4_chain w;
            memcpy(&w, q + i * sizeof w, sizeof w);
            eb_chain& k = c->chains[i];
            k.side = w.side_sym_len & 1u;
            k.sym = (uint8_t)"ACGT"[(w.side_sym_len >> 1) & 3u];
            k.reserved = 0;
            k.len = w.side_sym_len >> 3;
            k.support = w.support;
        }
        q = buf + sizeof h + ((h.count * sizeof(eb_wire4_cluster) + h.n_chains * sizeof(eb_wire4_chain) + 7) & ~7ull);
        memcpy(c->left, q, h.n_chains * c->left_words() * 8);
        q += h.n_chains * c->left_words() * 8;
        memcpy(c->right, q, h.count * c->right_words() * 8);
    } else if (!wire) {
        memcpy(c->blob, buf, need);
    } else {
        // compact image -> records (first_chain = running sum of n_slots)
        const uint8_t* q = buf + sizeof h;
        uint64_t chain = 0;
        for (uint64_t i = 0; i < h.count; ++i) {
            eb_wire_cluster w;
            memcpy(&w, q + i * sizeof w, sizeof w);
            eb_cluster& r = c->clusters[i];
            r.begin = w.begin;
            r.end = w.begin + w.len;
            r.first_chain = (uint32_t)chain;
            r.n_slots = w.n_slots;
            r.right_found = w.right_found & 1u;
            r.right_len = w.right_len;
            r.reserved = (w.right_found & 2u) ? EB_RIGHT_SKIPPED : 0u;
            chain += w.n_slots;
        }
        if (chain != h.n_chains) { eb_set_error("corrupt call set: chain counts do not add up"); eb_calls_free(c); return EB_ERR_ARG; }
        q += h.count * sizeof(eb_wire_cluster);
        for (uint64_t i = 0; i < h.n_chains; ++i) {
            eb_wire_chain w;
            memcpy(&w, q + i * sizeof w, sizeof w);
            eb_chain& k = c->chains[i];
            k.side = w.side;
            k.sym = w.sym;
            k.reserved = 0;
            k.len = w.len;
            k.support = w.support;
        }
        q += h.n_chains * sizeof(eb_wire_chain);
        memcpy(c->left, q, h.n_chains * c->left_words() * 8);
        q += h.n_chains * c->left_words() * 8;
        memcpy(c->right, q, h.count * c->right_words() * 8);
    }
    *out = c;
    return EB_OK;
}

// ------------------------------------------------------------------ writer entry points

extern "C" int eb_format_snp(const eb_calls* c, const eb_params* p, int two_samples, uint64_t* cluster_nr, uint64_t* events,
                             char** text, uint64_t* len) {
    if (!text || !len) { eb_set_error("null argument"); return EB_ERR_ARG; }
    Formatted F;
    int rc = format_calls(c, p, two_samples, cluster_nr, events, F);
    if (rc != EB_OK) return rc;
    char* b = static_cast<char*>(malloc(F.total() + 1));
    if (!b) { eb_set_error("out of memory"); return EB_ERR_NOMEM; }
    F.for_each_piece([&](size_t t) { F.emit(t, b + F.offset[t]); });
    b[F.total()] = 0;
    *text = b;
    *len = F.total();
    return EB_OK;
}

extern "C" int eb_write_snp(const eb_calls* c, const eb_params* p, int two_samples, const char* path, int append,
                            uint64_t* cluster_nr, uint64_t* events) {
    if (!path) { eb_set_error("null argument"); return EB_ERR_ARG; }
    Formatted F;
    int rc = format_calls(c, p, two_samples, cluster_nr, events, F);
    if (rc != EB_OK) return rc;
    int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
    if (fd < 0) { eb_set_error("cannot open %s for writing", path); return EB_ERR_IO; }
    struct stat st;
    uint64_t base = 0;
    if (append) {
        if (fstat(fd, &st) != 0) { close(fd); eb_set_error("cannot stat %s", path); return EB_ERR_IO; }
        base = (uint64_t)st.st_size;
    }
    // every thread finishes its piece (cluster numbers filled in) and writes it at its own offset of the file
    std::vector<int> bad(F.pieces.size(), 0);
    F.for_each_piece([&](size_t t) {
        const uint64_t z = F.offset[t + 1] - F.offset[t];
        if (!z) return;
        std::string out((size_t)z, '\0');
        F.emit(t, &out[0]);
        for (uint64_t done = 0; done < z;) {
            ssize_t w = pwrite(fd, out.data() + done, (size_t)(z - done), (off_t)(base + F.offset[t] + done));
            if (w <= 0) { bad[t] = 1; return; }
            done += (uint64_t)w;
        }
    });
    int failed = close(fd);
    for (int b : bad) failed |= b;
    if (failed) { eb_set_error("short write to %s", path); return EB_ERR_IO; }
    return EB_OK;
}

extern "C" void eb_free(void* p) { free(p); }
