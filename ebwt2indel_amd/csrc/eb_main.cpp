// eb_main.cpp -- the `ebwt2InDel` command line, same flags / defaults / messages as the reference
// (ebwt2InDel.cpp:76-103 help, :1677-1823 main), driving the GPU hot path through the C ABI only.
// Extra flag (not in the reference): -D <device> selects the GPU (default 0).
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>
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
           "-D <arg>    GPU to run on. Default: 0.\n\n"
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

static int read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    size_t r = sz > 0 ? fread(out.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return r == out.size() ? 0 : -1;
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0); // the phase-4 driver overlaps five streams; the HIP default of 4 queues serialises them
    if (argc < 3) help();
    eb_params p;
    memset(&p, 0, sizeof p);
    p.term = '#';
    std::string input1, input2, input_da, output;
    int device = 0;
    int opt;
    while ((opt = getopt(argc, argv, "h1:2:v:L:R:m:g:k:t:o:d:c:q:D:")) != -1) {
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
    rc = eb_flags_build(b1, b2, p.K, p.k_right, &fl, &ns);
    if (rc != EB_OK) return fail("Error while navigating the suffix tree");
    if (two_bwts) printf("Computed %llu/%llu DA values.\n", (unsigned long long)ns.da_values, (unsigned long long)ns.n);
    printf("Processed %llu suffix-tree leaves.\n\n", (unsigned long long)ns.leaves);
    printf("Phase 3/4: computing LCP minima.\n");
    printf("Computed %llu/%llu LCP values.\n", (unsigned long long)ns.lcp_values, (unsigned long long)ns.n);
    printf("Found %llu LCP minima.\n", (unsigned long long)ns.minima);
    printf("Processed %llu suffix-tree nodes.\n\n", (unsigned long long)ns.nodes);
    double t2 = now_s();

    if (with_da) {
        std::vector<uint8_t> da;
        if (read_file(input_da, da) != 0) { printf("Error: could not read file %s\n", input_da.c_str()); return 1; }
        rc = eb_flags_set_da(fl, da.data(), da.size());
        if (rc != EB_OK) return fail("Error while loading the document array");
    }

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
    printf("\nTimings (s): load+index %.3f, phases 2-3 %.3f (leaves %.3f, nodes %.3f, %llu rank queries, %llu launches), "
           "phase 4 device %.3f (scan %.3f ms, extend %.3f ms, right %.3f ms), write %.3f\n",
           t1 - t0, t2 - t1, ns.seconds_leaves, ns.seconds_nodes, (unsigned long long)ns.rank_queries, (unsigned long long)ns.batches,
           t3 - t2, ss.ms_scan, ss.ms_extend, ss.ms_right, t4 - t3);

    eb_calls_free(calls);
    eb_flags_destroy(fl);
    eb_index_destroy(b1);
    eb_index_destroy(b2);
    return 0;
}
