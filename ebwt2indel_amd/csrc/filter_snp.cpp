// filter_snp -- keep the .snp records whose coverage lies in [m, M] (M = 0: no upper bound); output on stdout.
// Post-filter of the documented one-sample pipeline (reference: filter_snp.cpp:22-81; header grammar
// ">cluster:<c>_id:<i>_right:<r>_cov:<n>_type:..." written by ebwt2InDel.cpp:1286-1310).  Host-only tool.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

static int coverage_of(const std::string& header) {
    // value after the ':' of the fourth '_'-separated token
    size_t pos = 0;
    for (int skip = 0; skip < 3; ++skip) {
        pos = header.find('_', pos);
        if (pos == std::string::npos) return 0;
        ++pos;
    }
    size_t end = header.find('_', pos);
    std::string tok = header.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
    size_t colon = tok.find(':');
    if (colon == std::string::npos) return 0;
    size_t colon2 = tok.find(':', colon + 1);
    return atoi(tok.substr(colon + 1, colon2 == std::string::npos ? std::string::npos : colon2 - colon - 1).c_str());
}

int main(int argc, char** argv) {
    if (argc != 3 && argc != 4) {
        std::cout << "filter_snp calls.snp m [M]\n\n"
                     "Input: a .snp file. Keep only reads with at least coverage m and at most M. Output to stdout.\n";
        return 0;
    }
    const int m = atoi(argv[2]);
    const int M = argc == 4 ? atoi(argv[3]) : 0;
    std::ifstream in(argv[1]);
    std::string line, header;
    int cov = 0;
    for (unsigned long idx = 0; std::getline(in, line); ++idx) {
        if (idx % 2 == 0) {
            header = line;
            cov = coverage_of(line);
        } else {
            if (cov >= m && (M == 0 || cov <= M)) std::cout << header << '\n' << line << '\n';
            header.clear();
            cov = 0;
        }
    }
    return 0;
}
