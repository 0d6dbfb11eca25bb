// eb_config.h -- the library's environment switches, read ONCE (when the library is loaded, or when a test calls
// eb_config_reload) into this struct; no other translation unit calls getenv.  Every switch is result-neutral: it selects
// between forms that the tests hold to the same bytes, and none is needed for normal use.
#pragma once
#include <stdint.h>

struct EbConfig {
    // EB_PSI: FL table of a repeatedly scanned index -- 0 never, 1 with the first phase-4 job, unset: once the index has been
    // scanned three times over
    int psi = -1;
    // EB_SELECT=<shift>[,<entry bits>]: sampling distance (log2, 5..10) and entry layout (25..29 or 32) of the select directory;
    // 0 = by index size (the layouts of indexes of 2^32 .. 2^38 symbols are reached on small inputs this way only)
    int sel_shift = 0, sel_blk_bits = 0;
    // EB_NAV_CHAINS: traversal kernels -- 1 chains in registers (default), 0 every node through the stack (the fallback's
    // kernels), 2 the chain-following walk thrown away and walked again, 3 chains in registers with stacks so small that the
    // walk really overflows and the fallback runs
    int nav_chains = 1;
    // EB_NAV_FORM: how the traversal collects the LCP thresholds -- complement (default: marks the few positions BELOW a
    // threshold), bytes / words (the reference's form, the fallback for inputs that are no eBWT)
    enum { NAV_COMPLEMENT = 0, NAV_BYTES = 1, NAV_WORDS = 2 };
    int nav_form = NAV_COMPLEMENT;
    // EB_NAV_BATCH=<records per batch>[,<arena batches>[,<records per shard at the cut>]]
    uint64_t nav_batch = 16ull << 20, nav_arena_batches = 12, nav_split_min = 4096;
    // EB_SCAN_OVERLAP: scan workgroups per CU beside the previous job's context kernels (-1: adaptive)
    int scan_overlap = -1;
    // EB_SCAN_ORDERED: 0 the unordered hand-over of the survivors (the fallback), k > 0: tiles per batch the ordered hand-over
    // starts with (power of two up to 128; it shrinks by itself when a batch overflows)
    int scan_ordered = -1;
    // EB_CONTEXT=extend=<1|4>,right=<1|4>,gate=<clusters>: lanes per chain / per cluster of the context kernels (0: by index
    // size and FL table), call-set size from which the right-context walk skips clusters that cannot print
    int extend_lanes = 0, right_lanes = 0;
    uint64_t right_gate_min = 524288;
    // EB_INDEX_PINNED=<min bytes>[,<piece bytes>[,<reader threads>]]: input size from which the loader goes through
    // page-locked pieces, piece size, host threads that fill them
    uint64_t pinned_min = 256ull << 20, pinned_piece = 32ull << 20;
    unsigned copy_threads = 12;
    // EB_WRITER_THREADS: host threads of the .snp writer (0: by call-set size)
    int writer_threads = 0;
    // EB_TRACE=<list of alloc, begin[:ms], nav, scan, index>: timelines on stderr
    bool trace_alloc = false, trace_nav = false, trace_scan = false, trace_index = false;
    double trace_begin_ms = 0; // > 0: every phase-4 begin call that takes longer
};

const EbConfig& eb_cfg();
