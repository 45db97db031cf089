// Test helper (built by tests/test_gaf_reader.py with g++; no GPU, no scorer
// library): runs the record reader + PackedAlignments::add and the parallel
// path-only reader of gfalign_amd/csrc/graph_io.h on the same files and prints
// whether they agree.
//   check_gaf_reader <gfa> <gaf> <threads>
#include <cstdio>
#include <cstdlib>

#include "graph_io.h"
#include "search.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    gfal::Graph g;
    std::string err;
    if (!gfal::read_gfa(argv[1], g, err)) {
        printf("gfa error: %s\n", err.c_str());
        return 2;
    }
    std::vector<gfal::GafRecord> recs;
    std::string slow_err, fast_err;
    const bool slow_ok = gfal::read_gaf(argv[2], recs, slow_err);
    gfal::PackedAlignments slow;
    if (slow_ok)
        for (auto &r : recs) slow.add(r, g);
    std::vector<int32_t> off, steps;
    const bool fast_ok = gfal::read_gaf_paths(argv[2], g, off, steps, fast_err, (unsigned)atoi(argv[3]));
    if (slow_ok != fast_ok) {
        printf("DIFFER ok %d vs %d (%s | %s)\n", slow_ok, fast_ok, slow_err.c_str(), fast_err.c_str());
        return 1;
    }
    if (!slow_ok) {
        if (slow_err != fast_err) {
            printf("DIFFER errors: %s | %s\n", slow_err.c_str(), fast_err.c_str());
            return 1;
        }
        printf("SAME error: %s\n", slow_err.c_str());
        return 0;
    }
    if (slow.off != off || slow.steps != steps) {
        printf("DIFFER content: %zu/%zu records, %zu/%zu steps\n", slow.off.size() - 1, off.size() - 1,
               slow.steps.size(), steps.size());
        return 1;
    }
    printf("SAME %zu records %zu steps\n", off.size() - 1, steps.size());
    return 0;
}
