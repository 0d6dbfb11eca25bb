// Synthetic-input tooling (NOT part of the hot path, no reference counterpart):
// builds the extended BWT (eBWT) of a read collection by direct suffix sorting.
//
// The reference consumes eBWTs made by external tools (BCR_LCP_GSA / egap,
// /root/reference/README.md:38); none of them exist offline, so fixtures and
// benchmark inputs are produced here.  Definition used:
// sort all suffixes  R[j][i:] + TERM  (0 <= i <= |R[j]|), TERM smaller than any
// base, equal strings ordered by read index j; the eBWT symbol of a suffix is
// R[j][i-1], or TERM when i == 0.  With two collections, collection-0 reads are
// numbered first and the document array (DA) records the collection of each row.
//
// C ABI (ctypes):  see ebwt_build() below.  Build: g++ -O3 -fopenmp -shared -fPIC.

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct SuffixLess {
    const uint8_t* t;   // concatenated text: read0 TERM read1 TERM ...
    uint8_t term;
    // Text positions grow with read index, so "p < q" is the read-index tie-break.
    bool operator()(uint64_t p, uint64_t q) const {
        const uint8_t* a = t + p;
        const uint8_t* b = t + q;
        for (;; ++a, ++b) {
            uint8_t x = *a, y = *b;
            if (x != y) {
                if (x == term) return true;
                if (y == term) return false;
                return x < y;
            }
            if (x == term) return p < q;
        }
    }
};

inline unsigned code5(uint8_t c, uint8_t term) {
    if (c == term) return 0;
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 3; default: return 4; }
}

}  // namespace

extern "C" {

// text      : nreads reads, each followed by one `term` byte; total length n.
// n_first   : text positions [0, n_first) belong to collection 0, the rest to 1.
// out_bwt   : n bytes (ASCII).   out_da : n bytes of '0'/'1', or NULL.
// returns 0 on success.
int ebwt_build(const uint8_t* text, uint64_t n, uint8_t term, uint64_t n_first,
               uint8_t* out_bwt, uint8_t* out_da, int threads) {
    if (!text || !out_bwt || n == 0 || text[n - 1] != term) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    // Bucket suffixes by their first PFX symbols (TERM ends the key), then sort buckets.
    const int PFX = 7;
    uint64_t nb = 1;
    for (int i = 0; i < PFX; ++i) nb *= 5;
    std::vector<uint64_t> key(n);
#pragma omp parallel for schedule(static)
    for (uint64_t p = 0; p < n; ++p) {
        uint64_t k = 0;
        bool ended = false;
        for (int i = 0; i < PFX; ++i) {
            unsigned c = 0;
            if (!ended) {
                c = code5(text[p + i], term);
                if (c == 0) ended = true;
            }
            k = k * 5 + c;
        }
        key[p] = k;
    }
    std::vector<uint64_t> start(nb + 1, 0);
    for (uint64_t p = 0; p < n; ++p) start[key[p] + 1]++;
    for (uint64_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    std::vector<uint64_t> sa(n);
    {
        std::vector<uint64_t> fill(start.begin(), start.end() - 1);
        for (uint64_t p = 0; p < n; ++p) sa[fill[key[p]]++] = p;  // stable: increasing p
    }
    key.clear();
    key.shrink_to_fit();
    SuffixLess less{text, term};
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t b = 0; b < nb; ++b) {
        if (start[b + 1] - start[b] > 1) std::sort(sa.begin() + start[b], sa.begin() + start[b + 1], less);
    }
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t p = sa[r];
        out_bwt[r] = (p == 0 || text[p - 1] == term) ? term : text[p - 1];
        if (out_da) out_da[r] = p < n_first ? '0' : '1';
    }
    return 0;
}

}  // extern "C"
