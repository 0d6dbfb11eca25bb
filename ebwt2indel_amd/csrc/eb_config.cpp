// eb_config.cpp -- reads the environment switches of libebwt2indel once (eb_config.h).
#include "eb_config.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

namespace {

std::vector<std::string> split(const char* s, char sep) {
    std::vector<std::string> out;
    if (!s) return out;
    std::string cur;
    for (; *s; ++s) {
        if (*s == sep) { out.push_back(cur); cur.clear(); }
        else cur.push_back(*s);
    }
    out.push_back(cur);
    return out;
}

EbConfig read_env() {
    EbConfig c;
    if (const char* e = getenv("EB_PSI")) c.psi = atoi(e);
    {
        auto v = split(getenv("EB_SELECT"), ',');
        if (v.size() > 0 && !v[0].empty()) { int s = atoi(v[0].c_str()); if (s >= 5 && s <= 10) c.sel_shift = s; }
        if (v.size() > 1 && !v[1].empty()) { int b = atoi(v[1].c_str()); if (b == 32 || (b >= 25 && b <= 29)) c.sel_blk_bits = b; }
    }
    if (const char* e = getenv("EB_NAV_CHAINS")) { int v = atoi(e); if (v >= 0 && v <= 3) c.nav_chains = v; }
    if (const char* e = getenv("EB_NAV_FORM")) {
        if (!strcmp(e, "bytes")) c.nav_form = EbConfig::NAV_BYTES;
        else if (!strcmp(e, "words")) c.nav_form = EbConfig::NAV_WORDS;
    }
    {
        auto v = split(getenv("EB_NAV_BATCH"), ',');
        if (v.size() > 0 && !v[0].empty()) c.nav_batch = strtoull(v[0].c_str(), nullptr, 10);
        if (v.size() > 1 && !v[1].empty()) c.nav_arena_batches = strtoull(v[1].c_str(), nullptr, 10);
        if (v.size() > 2 && !v[2].empty()) c.nav_split_min = strtoull(v[2].c_str(), nullptr, 10);
    }
    if (const char* e = getenv("EB_SCAN_OVERLAP")) { int v = atoi(e); c.scan_overlap = v < 0 ? -1 : (v > 8 ? 8 : v); }
    if (const char* e = getenv("EB_SCAN_ORDERED")) {
        int v = atoi(e);
        if (v <= 0) c.scan_ordered = 0;
        else { int t = 1; while (t * 2 <= v && t < 128) t *= 2; c.scan_ordered = t; }
    }
    for (const std::string& kv : split(getenv("EB_CONTEXT"), ',')) {
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        if (k == "extend") { int n = atoi(v.c_str()); if (n == 1 || n == 4) c.extend_lanes = n; }
        else if (k == "right") { int n = atoi(v.c_str()); if (n == 1 || n == 4) c.right_lanes = n; }
        else if (k == "gate") c.right_gate_min = strtoull(v.c_str(), nullptr, 10);
    }
    {
        auto v = split(getenv("EB_INDEX_PINNED"), ',');
        if (v.size() > 0 && !v[0].empty()) c.pinned_min = strtoull(v[0].c_str(), nullptr, 10);
        if (v.size() > 1 && !v[1].empty()) { c.pinned_piece = strtoull(v[1].c_str(), nullptr, 10) & ~127ull; if (c.pinned_piece < 128) c.pinned_piece = 128; }
        if (v.size() > 2 && !v[2].empty()) { int t = atoi(v[2].c_str()); if (t >= 1 && t <= 64) c.copy_threads = (unsigned)t; }
    }
    if (const char* e = getenv("EB_WRITER_THREADS")) c.writer_threads = atoi(e) < 1 ? 1 : atoi(e);
    for (const std::string& t : split(getenv("EB_TRACE"), ',')) {
        if (t == "alloc") c.trace_alloc = true;
        else if (t == "nav") c.trace_nav = true;
        else if (t == "scan") c.trace_scan = true;
        else if (t == "index") c.trace_index = true;
        else if (t.rfind("begin", 0) == 0) c.trace_begin_ms = t.size() > 6 ? atof(t.c_str() + 6) : 1e-9;
    }
    return c;
}

std::mutex g_mu;
EbConfig g_cfg = read_env(); // when the library is loaded

}  // namespace

const EbConfig& eb_cfg() { return g_cfg; }

// Tests change switches inside one process: they call this after every change (tests/conftest.py does it for them).
extern "C" void eb_config_reload(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cfg = read_env();
}
