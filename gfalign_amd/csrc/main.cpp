// main.cpp -- `gfalign` command line for the tools that sit on the scoring hot
// path: search, evalPath, filter (and evalGFA's alignment summary).  Option
// names, messages and stdout follow reference src/main.cpp / src/eval.cpp /
// src/alignments.cpp; the scoring itself runs on the MI355X through
// include/gfalign_scorer.h.  `align` (shells out to GraphAligner) and
// `subgraph` (pure gfalibs) are out of scope.
#include <getopt.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "graph_io.h"
#include "search.h"

using namespace gfal;

namespace {

const char *VERSION = "0.0.1";   // reference src/main.cpp:49

struct Options {
    int mode = -1;
    std::string gfa, gaf, node_file, source, destination, path, out_file;
    uint32_t max_steps = 100000, min_nodes = 0;
    int return_all_paths = 0, cmd_flag = 0, stats_flag = 0, sort_alignment = 0;
    int terminal_alignments = 0;
    int device = 0, n_devices = 1;
    unsigned threads = 0;            // -j: host threads of the file readers (0 = default)
};

#ifndef GFAL_BUILD_ID
#define GFAL_BUILD_ID "unstamped"
#endif

// `-f -` / `-g -`: the reference reads that input from stdin (src/main.cpp:
// 425-429, gfalibs StreamObj).  The readers here map files, so the pipe is
// spooled to a temporary file first (removed at exit).
std::vector<std::string> g_spooled;
void remove_spooled()
{
    for (auto &f : g_spooled) unlink(f.c_str());
}
std::string spool_stdin()
{
    char name[] = "/tmp/gfalign-stdin-XXXXXX";
    int fd = mkstemp(name);
    if (fd < 0) {
        fprintf(stderr, "Error: cannot create a temporary file for the piped input\n");
        exit(EXIT_FAILURE);
    }
    if (g_spooled.empty()) atexit(remove_spooled);
    g_spooled.push_back(name);
    char buf[1 << 16];
    ssize_t got;
    while ((got = read(STDIN_FILENO, buf, sizeof buf)) > 0) {
        ssize_t off = 0;
        while (off < got) {
            ssize_t w = write(fd, buf + off, (size_t)(got - off));
            if (w <= 0) {
                fprintf(stderr, "Error: cannot spool the piped input\n");
                exit(EXIT_FAILURE);
            }
            off += w;
        }
    }
    close(fd);
    return name;
}

int verbose_flag = 0;

void require_file(const char *path)   // gfalibs ifFileExists
{
    if (access(path, F_OK) == -1) {
        fprintf(stderr, "Error - file does not exist: %s\n", path);
        exit(EXIT_FAILURE);
    }
}

// ---------------------------------------------------------------- stats ---
// reference src/alignments.cpp:237-276 (+ :304-351 for the duplicate marks)
struct AlignmentTotals {
    unsigned long long qlen = 0, algseq = 0, plus = 0, minus = 0, plen = 0, mapq = 0,
                       matches = 0, blocklen = 0;
    unsigned long long primary = 0, secondary = 0, supplementary = 0, terminal_supp = 0;
    void add(const GafRecord &r)
    {
        qlen += r.qlen;
        algseq += r.qend - r.qstart;
        (r.strand == '+' ? plus : minus)++;
        plen += r.plen;
        matches += r.matches;
        blocklen += r.blocklen;
        mapq += r.mapq;
    }
};

// gfalibs gfa_round: two decimals.  `fixed` reproduces the stream state the
// reference is in when the summary is printed from outputAlignments
// (validateFiles/test.7.tst: "80.00") as opposed to evalGFA (test.0: "37.5").
std::string rounded(double v, bool fixed)
{
    std::ostringstream os;
    double r = std::round(v * 100.0) / 100.0;
    if (fixed) os << std::fixed << std::setprecision(2);
    os << r;
    return os.str();
}

void print_stats(const AlignmentTotals &t, size_t n, bool fixed)
{
    auto avg = [&](unsigned long long v) { return rounded((double)v / (double)n, fixed); };
    std::cout << "+++Alignment summary+++: \n";
    std::cout << "# alignments: " << n << "\n";
    std::cout << "Average read length: " << avg(t.qlen) << "\n";
    std::cout << "Average aligned sequence: " << avg(t.algseq) << "\n";
    std::cout << "Alignment orientation (+/-): " << t.plus << "("
              << rounded((double)t.plus / (double)(t.plus + t.minus) * 100, fixed) << "%):"
              << t.minus << "("
              << rounded((double)t.minus / (double)(t.plus + t.minus) * 100, fixed) << "%)\n";
    std::cout << "Average path length: " << avg(t.plen) << "\n";
    std::cout << "Average alignment quality: " << avg(t.mapq) << "\n";
    std::cout << "Average matches #: " << avg(t.matches) << "\n";
    std::cout << "Average block length: " << avg(t.blocklen) << "\n";
    std::cout << "Primary alignments: " << t.primary << "\n";
    std::cout << "Secondary alignments: " << t.secondary << "\n";
    std::cout << "Supplementary alignments: " << t.supplementary << "\n";
    std::cout << "Terminal supplementary alignments: " << t.terminal_supp << "\n";
}

// ------------------------------------------------------------ evalPath ---
// Host restatement of the traceback rows, for PRINTING only: the counters and
// the per-alignment scores come from the GPU; this renders the B row of the
// winning orientation (reference include/alignments.h:113-121).
struct Rows {
    std::vector<Step> a, b;
    int score = 0;
};

Rows traceback_rows(const std::vector<Step> &A, const std::vector<Step> &B)
{
    const int n = (int)A.size(), m = (int)B.size();
    auto eq = [](const Step &x, const Step &y) {
        return x.id == y.id && x.orientation == y.orientation;
    };
    std::vector<int> dp((size_t)(n + 1) * (m + 1), 0);
    auto at = [&](int i, int j) -> int & { return dp[(size_t)i * (m + 1) + j]; };
    for (int j = 0; j <= m && j <= n; ++j) at(0, j) = -j;       // src/alignments.cpp:500
    for (int i = 1; i <= n; ++i)
        for (int j = 1; j <= m; ++j) {
            int s = eq(A[i - 1], B[j - 1]) ? 0 : -1;
            at(i, j) = std::max(at(i - 1, j - 1) + s,
                                std::max(at(i - 1, j) + (j < m ? -1 : 0), at(i, j - 1) - 1));
        }
    Rows r;
    const Step gap{-1, '0'};
    int i = n, j = m, taken = 0;
    while (i != 0 || j != 0) {                                   // :516-550
        if (i == 0) {
            r.a.push_back(gap); r.b.push_back(B[j - 1]); --j;
        } else if (j == 0) {
            r.a.push_back(A[i - 1]); r.b.push_back(gap); --i;
        } else {
            int s = eq(A[i - 1], B[j - 1]) ? 0 : -1;
            if (at(i, j) == at(i - 1, j - 1) + s) {
                r.a.push_back(A[i - 1]); r.b.push_back(B[j - 1]);
                ++taken; --i; --j; r.score += s;
            } else if (at(i - 1, j) >= at(i, j - 1)) {
                r.a.push_back(A[i - 1]); r.b.push_back(gap); --i;
                if (taken > 0) r.score -= 1;
            } else {
                r.a.push_back(gap); r.b.push_back(B[j - 1]);
                ++taken; --j; r.score -= 1;
            }
        }
    }
    std::reverse(r.a.begin(), r.a.end());
    std::reverse(r.b.begin(), r.b.end());
    return r;
}

std::string b_row(const Rows &r, const Graph &g)   // include/alignments.h:113-121
{
    std::string s;
    for (size_t i = 0; i < r.b.size(); ++i) {
        const Step &x = r.a[i], &y = r.b[i];
        if (y.id == -1)
            s += std::string(g.headers[(size_t)x.id].size() + 1, '-') + ",";
        else if (x.id != y.id || x.orientation != y.orientation)
            s += g.headers[(size_t)y.id] + y.orientation + ",";
        else
            s += std::string(g.headers[(size_t)y.id].size() + 1, '.') + ",";
    }
    return s;
}

int run_eval_path(const Options &o, const Graph &g, const std::vector<GafRecord> &recs)
{
    // src/eval.cpp:203-227: split on ',' / ';', last char = orientation
    std::vector<Step> path;
    std::string comp;
    std::vector<std::string> comps;
    for (char c : o.path) {
        if (c == ',' || c == ';') {
            comps.push_back(comp);
            comp.clear();
        } else {
            comp += c;
        }
    }
    comps.push_back(comp);
    for (size_t k = 0; k < comps.size(); ++k) {
        std::string c = comps[k];
        if (k == 0 && c.empty()) {
            fprintf(stderr, "Error: cannot handle starting gap. Terminating.\n");
            return 1;
        }
        if (c.empty()) {   // the reference would index an empty string here (UB)
            fprintf(stderr, "Error: cannot find node (). Terminating.\n");
            return 1;
        }
        char orient = c.back();
        c.pop_back();
        auto it = g.ids.find(c);
        if (it == g.ids.end()) {
            fprintf(stderr, "Error: cannot find node (%s). Terminating.\n", c.c_str());
            return 1;
        }
        path.push_back({(int32_t)it->second, orient});
    }
    if (path.size() > GFAL_MAX_STEPS) {
        fprintf(stderr, "Error: path longer than %d steps.\n", GFAL_MAX_STEPS);
        return 1;
    }
    const uint32_t uniques = count_uniques(path);

    PackedAlignments packed;
    for (auto &r : recs) packed.add(r, g);
    PathScorer scorer;
    std::vector<int32_t> universe;
    for (auto &s : path) universe.push_back(s.id);
    const bool share = getenv("GFALIGN_SHARE_DEVICE") != nullptr;
    if (!scorer.open(packed, (int32_t)g.headers.size(), o.device, universe, o.n_devices, share))
        return 1;

    std::cout << path_string(path, g) << std::endl;                       // :72-73
    std::vector<int32_t> pst;
    for (auto &s : path) pst.push_back(pack(s));
    std::vector<int32_t> off{0, (int32_t)pst.size()};
    std::vector<uint32_t> bad, good;
    if (!scorer.score(off, pst, false, bad, good)) return 1;              // :238
    std::vector<int32_t> fw, rc;
    if (!scorer.pair_scores(pst, fw, rc)) return 1;
    // One row per alignment (:100-102).  The scores come from the GPU; the gapped
    // row text needs the traceback itself, which is rendered on the host -- for a
    // 10 M-alignment GAF that is tens of GB of text, so blocks of alignments are
    // rendered on all cores and written in order.
    const size_t n_rec = recs.size();
    const size_t block = 2048;
    const size_t n_blocks = (n_rec + block - 1) / block;
    unsigned n_threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n_blocks < 4) n_threads = 1;
    std::vector<std::string> text(n_blocks);
    std::vector<char> ready(n_blocks, 0);
    std::string failure;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next_block{0};
    size_t written = 0;                       // guarded by mu: blocks already printed
    auto render = [&]() {
        std::vector<Step> B;
        for (size_t blk = next_block++; blk < n_blocks; blk = next_block++) {
            {   // stay at most 64 blocks ahead of the writer (memory)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return blk < written + 64 || !failure.empty(); });
                if (!failure.empty()) return;
            }
            std::string out;
            const size_t hi = std::min(n_rec, (blk + 1) * block);
            for (size_t k = blk * block; k < hi; ++k) {
                B.clear();
                for (int32_t t = packed.off[k]; t < packed.off[k + 1]; ++t)
                    B.push_back({packed.steps[(size_t)t] >> 1, (packed.steps[(size_t)t] & 1) ? '-' : '+'});
                const bool show_fw = fw[k] > rc[k];
                if (!show_fw) {   // include/alignments.h:64-70
                    std::reverse(B.begin(), B.end());
                    for (auto &st : B) st.orientation = (st.orientation == '+') ? '-' : '+';
                }
                Rows rows = traceback_rows(path, B);
                const int32_t best = std::max(fw[k], rc[k]);
                if (rows.score != best) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (failure.empty())
                        failure = "device score " + std::to_string(best) + " != rendered score " +
                                  std::to_string(rows.score) + " for " + recs[k].qname;
                    cv.notify_all();
                    return;
                }
                out += b_row(rows, g);
                out += '\t';
                out += recs[k].qname;
                out += '\t';
                out += std::to_string(best);
                out += '\n';
            }
            std::lock_guard<std::mutex> lk(mu);
            text[blk] = std::move(out);
            ready[blk] = 1;
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(render);
    for (size_t blk = 0; blk < n_blocks; ++blk) {
        std::string out;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[blk] || !failure.empty(); });
            if (!failure.empty()) break;
            out = std::move(text[blk]);
            text[blk].clear();
            text[blk].shrink_to_fit();
            written = blk + 1;
            cv.notify_all();
        }
        std::cout.write(out.data(), (std::streamsize)out.size());
    }
    for (auto &t : pool) t.join();
    if (!failure.empty()) {
        fprintf(stderr, "Error: %s\n", failure.c_str());
        return 1;
    }
    std::cout.flush();
    const int32_t alt = (int32_t)bad[0] - (int32_t)good[0] - (int32_t)uniques;
    std::cout << bad[0] << "\t" << good[0] << "\t" << alt << "\t" << path.size() << "\t"
              << uniques << std::endl;                                     // :241
    return 0;
}

// -------------------------------------------------------------- filter ---
int run_filter(const Options &o, std::vector<GafRecord> &recs, const AlignmentTotals &totals)
{
    std::unordered_set<std::string> allowed;      // src/input-gfalign.cpp:65-74
    {
        std::ifstream in(o.node_file);
        std::string line;
        while (std::getline(in, line)) allowed.insert(line);
    }
    std::vector<GafRecord> kept;                   // src/alignments.cpp:459-472
    std::vector<std::pair<std::string, char>> nodes;
    for (auto &r : recs) {
        gaf_path_nodes(r.path, nodes);
        bool inside = true;
        for (auto &nd : nodes) inside &= allowed.count(nd.first) != 0;
        if (inside && (int32_t)nodes.size() >= (int32_t)o.min_nodes) kept.push_back(std::move(r));
    }
    recs.swap(kept);
    if (o.out_file.empty()) return 0;              // src/input-gfalign.cpp:116-117
    // src/alignments.cpp:286-302: totals are the ones accumulated at load time
    print_stats(totals, recs.size(), true);
    std::ofstream out(o.out_file);
    for (auto &r : recs) out << r.print();
    return 0;
}

// ------------------------------------------------------------- evalGFA ---
// evalGFA -o: how many read alignments walk over every link of the graph,
// written as an RC:i tag on the L lines (reference src/eval.cpp:34-61 over the
// edge multigraph of src/alignments.cpp:353-403, restated step for step: the
// forward edge and its reverse twin are counted together, and a link that is its
// own reverse is counted twice from its second sighting on, as there).
// The reference hands the tagged graph to gfalibs' GFA writer, which is not in
// the tree: here the input GFA is written back line by line with the tag
// appended to every L line, to the file named by -o, or to stdout when -o names
// a format ("gfa") rather than a file.  The tag values are pinned by the
// in-tree code; the surrounding GFA text is "parity unpinned" (no fixture).
struct WeightedEdge {
    char from_orient;
    uint32_t to;
    char to_orient;
    unsigned weight;
    bool same(const WeightedEdge &e) const
    {
        return from_orient == e.from_orient && to == e.to && to_orient == e.to_orient;
    }
};

int tag_edges_and_write(const Options &o, const Graph &g, const std::vector<GafRecord> &recs)
{
    std::vector<std::vector<WeightedEdge>> adj(g.headers.size());
    auto find = [](std::vector<WeightedEdge> &lst, const WeightedEdge &e) -> WeightedEdge * {
        for (auto &x : lst)
            if (x.same(e)) return &x;
        return nullptr;
    };
    std::vector<std::pair<std::string, char>> nodes;
    for (const GafRecord &r : recs) {                         // src/alignments.cpp:361-399
        gaf_path_nodes(r.path, nodes);                        // :407-445 consecutive pairs
        for (size_t k = 0; k + 1 < nodes.size(); ++k) {
            if (g.headers.empty()) break;
            const uint32_t id1 = g.id_or_zero(nodes[k].first), id2 = g.id_or_zero(nodes[k + 1].first);
            const WeightedEdge fw{nodes[k].second, id2, nodes[k + 1].second, 1};
            const WeightedEdge rv{flip(nodes[k + 1].second), id1, flip(nodes[k].second), 1};
            if (WeightedEdge *hit = find(adj[id1], fw)) {
                ++hit->weight;
                if (WeightedEdge *twin = find(adj[id2], rv)) ++twin->weight;
            } else {
                adj[id1].push_back(fw);
                if (!find(adj[id2], rv)) adj[id2].push_back(rv);
            }
        }
    }
    std::ifstream in(o.gfa);
    if (!in) {
        fprintf(stderr, "Error: cannot open %s\n", o.gfa.c_str());
        return 1;
    }
    std::ofstream file;
    const bool to_file = o.out_file.find('.') != std::string::npos;
    if (to_file) {
        file.open(o.out_file);
        if (!file) {
            fprintf(stderr, "Error: cannot write %s\n", o.out_file.c_str());
            return 1;
        }
    }
    std::ostream &out = to_file ? static_cast<std::ostream &>(file) : std::cout;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == 'L') {
            auto cols = split(line, '\t');
            unsigned weight = 0;                               // src/eval.cpp:46-55
            if (cols.size() >= 5 && !cols[2].empty() && !cols[4].empty()) {
                auto a = g.ids.find(cols[1]), b = g.ids.find(cols[3]);
                if (a != g.ids.end() && b != g.ids.end())
                    if (WeightedEdge *hit = find(adj[a->second], WeightedEdge{cols[2][0], b->second, cols[4][0], 0}))
                        weight = hit->weight;
            }
            out << line << "\tRC:i:" << weight << "\n";        // :57-58
        } else {
            out << line << "\n";
        }
    }
    return 0;
}

int run_eval_gfa(const Options &o, const Graph &g, std::vector<GafRecord> &recs, AlignmentTotals totals)
{
    if (o.gaf.empty()) return 0;
    std::stable_sort(recs.begin(), recs.end(),
                     [](const GafRecord &a, const GafRecord &b) { return a.qname < b.qname; });
    // src/alignments.cpp:304-351
    std::string prev;
    std::vector<const GafRecord *> group;
    for (size_t k = 0; k < recs.size(); ++k) {
        group.push_back(&recs[k]);
        if (recs[k].qname == prev) {
            ++totals.secondary;
            if (k + 1 == recs.size() || recs[k + 1].qname != recs[k].qname) {
                std::vector<const GafRecord *> g2 = group;
                std::stable_sort(g2.begin(), g2.end(), [](const GafRecord *a, const GafRecord *b) {
                    return a->qstart < b->qstart;
                });
                unsigned pos = 0, count = 0;
                for (auto *r : g2) {
                    if (pos != 0 && r->qstart > pos) {
                        ++totals.supplementary;
                        ++count;
                    }
                    pos = r->qend;
                }
                // unsigned arithmetic as in the reference: pLen - 500 wraps for
                // paths shorter than 500 (src/alignments.cpp:343)
                if (g2.size() == 2 && count == 1 && g2[0]->pend >= g2[0]->plen - 500u &&
                    g2[1]->pstart <= 500) {
                    ++totals.terminal_supp;
                    if (o.terminal_alignments) std::cout << g2[0]->print() << g2[1]->print();
                }
                group.clear();
            }
        } else {
            ++totals.primary;
            prev = recs[k].qname;
        }
    }
    // -g always sets alignStats_flag (src/main.cpp:310), so the summary is
    // printed and the --sort-alignment output branch of
    // src/input-gfalign.cpp:88-91 is never taken (validateFiles/test.2.tst)
    print_stats(totals, recs.size(), false);
    if (!o.out_file.empty()) return tag_edges_and_write(o, g, recs);   // src/input-gfalign.cpp:93-97
    return 0;
}

void print_help()
{
    printf("gfalign [command]\n-h for additional help.\n");
    printf("\nModes:\n");
    printf("evalGFA\nsearch\nfilter\nevalPath\n");
    exit(0);
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc == 1) print_help();
    Options o;
    const std::string tool = argv[1];
    if (tool == "--build-id") {    // hash of the sources this binary was built from (build.py)
        printf("%s\n", GFAL_BUILD_ID);
        return 0;
    }
    // the scorer library is loaded at run time: refuse one with another ABI
    // (gfal_info layout) instead of letting it write past our structs
    if (gfal_abi_version() != GFAL_ABI_VERSION) {
        fprintf(stderr, "Error: libgfalign_scorer.so has ABI %d, this binary was built for %d\n",
                gfal_abi_version(), GFAL_ABI_VERSION);
        return EXIT_FAILURE;
    }
    if (tool == "evalGFA") o.mode = 1;
    else if (tool == "search") o.mode = 3;
    else if (tool == "filter") o.mode = 4;
    else if (tool == "evalPath") o.mode = 5;
    else if (tool == "align" || tool == "subgraph") {
        fprintf(stderr, "mode '%s' is not part of this build (see DESIGN.md).\n", argv[1]);
        return EXIT_FAILURE;
    } else if (tool == "-h" || tool == "--help") {
        print_help();
    } else if (tool == "-v" || tool == "--version") {
        printf("gfalign v%s\n", VERSION);
        return 0;
    } else {
        fprintf(stderr, "mode '%s' does not exist. Terminating.\n", argv[1]);   // main.cpp:116
        return EXIT_FAILURE;
    }

    static struct option long_options[] = {
        {"destination", required_argument, 0, 'd'},
        {"input-sequence", required_argument, 0, 'f'},
        {"input-alignment", required_argument, 0, 'g'},
        {"max-steps", required_argument, 0, 'm'},
        {"node-file", required_argument, 0, 'n'},
        {"node-list", required_argument, 0, 'n'},
        {"out-format", required_argument, 0, 'o'},
        {"source", required_argument, 0, 's'},
        {"path", required_argument, 0, 'p'},
        {"threads", required_argument, 0, 'j'},
        {"return-all-paths", no_argument, &o.return_all_paths, 1},
        {"graph-statistics", no_argument, &o.stats_flag, 1},
        {"sort-alignment", no_argument, &o.sort_alignment, 1},
        {"output-terminal-alignments", no_argument, &o.terminal_alignments, 1},
        {"min-nodes", required_argument, 0, 1},
        {"device", required_argument, 0, 2},
        {"devices", required_argument, 0, 3},
        {"cmd", no_argument, &o.cmd_flag, 1},
        {"verbose", no_argument, &verbose_flag, 1},
        {"version", no_argument, 0, 'v'},
        {"help", no_argument, 0, 'h'},
        {0, 0, 0, 0}};
    int c, idx = 0;
    while ((c = getopt_long(argc, argv, "-:d:f:g:j:m:n:o:p:s:vh", long_options, &idx)) != -1) {
        switch (c) {
        case 'd': o.destination = optarg; break;
        case 'f':
            if (!strcmp(optarg, "-")) o.gfa = spool_stdin();      // main.cpp:425-429, 443-449
            else { require_file(optarg); o.gfa = optarg; }
            break;
        case 'g':
            if (!strcmp(optarg, "-")) o.gaf = spool_stdin();      // main.cpp:450-458
            else { require_file(optarg); o.gaf = optarg; }
            break;
        case 'm': o.max_steps = (uint32_t)atoi(optarg); break;   // main.cpp:462-464 (atoi)
        case 'n': require_file(optarg); o.node_file = optarg; break;
        case 'o': o.out_file = optarg; break;
        case 'p': o.path = optarg; break;
        case 's': o.source = optarg; break;
        case 'j': o.threads = (unsigned)std::max(0, atoi(optarg)); break;   // main.cpp:472-474
        case 1: o.min_nodes = (uint32_t)atoi(optarg); break;
        case 2: o.device = atoi(optarg); break;
        case 3: o.n_devices = std::max(1, atoi(optarg)); break;
        case 'v':
            printf("gfalign v%s\n", VERSION);
            printf("Giulio Formenti giulio.formenti@gmail.com\n");
            return 0;
        case 'h':
            printf("gfalign %s [options]\nOptions:\n", tool.c_str());
            printf("-d --destination <string> destination node.\n");
            printf("-f --input-sequence <filename> sequence input file (GFA1).\n");
            printf("-g --input-alignment alignment input file (currently supports: GAF).\n");
            printf("-m --max-steps <int> limit graph exploration.\n");
            printf("-n --node-file <filename> list of nodes available to the search.\n");
            printf("-p --path path to evaluate (evalPath).\n");
            printf("-s --source <string> source node.\n");
            printf("--return-all-paths return all viable paths as they are discovered, not only "
                   "better ones (default: false).\n");
            printf("--min-nodes <int> do not report paths with less than int nodes (default: 0).\n");
            printf("--device <int> first HIP device to score on (default: 0).\n");
            printf("--devices <int> shard the alignments over this many devices (default: 1).\n");
            return 0;
        default: break;
        }
    }
    if (o.cmd_flag) {   // main.cpp:651-656
        for (int a = 0; a < argc; ++a) printf("%s ", argv[a]);
        printf("\n");
    }

    if (o.stats_flag) {
        // gfalibs' Report::reportStats (the gfastats summary): its source is not in
        // the reference tree, so the flag is refused, not ignored
        fprintf(stderr, "--graph-statistics is not part of this build (see DESIGN.md).\n");
        return EXIT_FAILURE;
    }
    const double t_start = gfal::now_s();
    Graph g;
    std::string err;
    if (!o.gfa.empty() && !read_gfa(o.gfa, g, err)) {
        fprintf(stderr, "Error: %s\n", err.c_str());
        return EXIT_FAILURE;
    }
    // HIP start-up (~0.2 s) runs beside the file parsing
    std::thread warm;
    if (o.mode == 3 || o.mode == 5)
        warm = std::thread([dev = o.device] {
            // an empty scorer brings up the runtime AND the device context
            if (gfal_device_count() <= dev) return;
            const int32_t off0 = 0;
            gfal_scorer *s = nullptr;
            if (gfal_scorer_create(&off0, nullptr, 0, 1, dev, &s) == GFAL_OK) gfal_scorer_destroy(s);
        });
    struct Joiner {
        std::thread &t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } joiner{warm};

    std::vector<GafRecord> recs;
    AlignmentTotals totals;
    PackedAlignments packed;
    if (!o.gaf.empty() && o.mode == 3) {
        // search only needs the path column (src/eval.cpp:123)
        if (!read_gaf_paths(o.gaf, g, packed.off, packed.steps, err, o.threads)) {
            fprintf(stderr, "Error: %s\n", err.c_str());
            return EXIT_FAILURE;
        }
    } else if (!o.gaf.empty()) {
        if (!read_gaf(o.gaf, recs, err, o.threads)) {
            fprintf(stderr, "Error: %s\n", err.c_str());
            return EXIT_FAILURE;
        }
        for (auto &r : recs) totals.add(r);
    }

    switch (o.mode) {
    case 1: return run_eval_gfa(o, g, recs, totals);
    case 3: {
        const double t_read = gfal::now_s();
        if (g.headers.empty()) {
            fprintf(stderr, "Error: the graph has no segments.\n");
            return EXIT_FAILURE;
        }
        // nodes the search can step on: the node list (include/nodetable.h:16-43)
        // plus source and destination (src/eval.cpp:127-128; a name that is not in
        // the graph becomes uId 0 there -- headersToIds[source] default-inserts --
        // and does so here)
        std::vector<int32_t> universe{(int32_t)g.id_or_zero(o.source),
                                      (int32_t)g.id_or_zero(o.destination)};
        {
            std::ifstream nf(o.node_file);
            std::string line;
            while (std::getline(nf, line)) {
                auto it = g.ids.find(split(line, '\t')[0]);
                if (it != g.ids.end()) universe.push_back((int32_t)it->second);
            }
        }
        const double t_pack = gfal::now_s();
        PathScorer scorer;
        const bool share = getenv("GFALIGN_SHARE_DEVICE") != nullptr;
        if (!scorer.open(packed, (int32_t)g.headers.size(), o.device, universe, o.n_devices,
                         share))
            return EXIT_FAILURE;
        SearchOptions so;
        so.node_file = o.node_file;
        so.source = o.source;
        so.destination = o.destination;
        so.max_steps = o.max_steps;
        so.min_nodes = o.min_nodes;
        so.return_all_paths = o.return_all_paths != 0;
        if (const char *k = getenv("GFALIGN_SPECULATE")) so.speculate = (size_t)std::max(1, atoi(k));
        if (const char *k = getenv("GFALIGN_PREFETCH")) so.prefetch = atoi(k) != 0;
        if (const char *k = getenv("GFALIGN_INCREMENTAL")) so.incremental = atoi(k) != 0;
        const double t_open = gfal::now_s();
        Search search(g, scorer, so, std::cout);
        search.set_alignments(packed);      // (GFALIGN_SPEC_POLICY=best: which extensions the speculation follows first)
        int rc = search.run();
        if (verbose_flag)
            fprintf(stderr,
                    "time: read %.3f s, pack %.3f s, scorer %.3f s, search %.3f s (candidates %.3f s, "
                    "scoring %.3f s, of which %.3f s making %llu extensions ahead)\n",
                    t_read - t_start, t_pack - t_read, t_open - t_pack, gfal::now_s() - t_open,
                    search.collect_seconds(), search.score_seconds(), search.ahead_seconds(),
                    (unsigned long long)search.made_ahead());
        if (verbose_flag && scorer.n_shards() > 1)
            fprintf(stderr, "%zu devices, per-path counters summed %s\n", scorer.n_shards(),
                    scorer.uses_rccl() ? "by an RCCL all-reduce" : "on the host");
        if (verbose_flag)
            fprintf(stderr, "scored %llu candidate paths in %llu batches, %llu of them in full and the rest from "
                            "their parents (%llu pairs took the exact DP); batch in "
                            "flight held what was needed next %llu times, not %llu times\n",
                    (unsigned long long)search.scored_paths(),
                    (unsigned long long)search.batches(), (unsigned long long)search.scored_in_full(),
                    (unsigned long long)scorer.dp_pairs(),
                    (unsigned long long)search.prefetch_hits(), (unsigned long long)search.prefetch_misses());
        if (verbose_flag)
            fprintf(stderr, "needed %llu of the scored candidates (extensions of popped entries: the reference's "
                            "evaluatePath calls)\n", (unsigned long long)search.needed_paths());
        return rc;
    }
    case 4: return run_filter(o, recs, totals);
    case 5: return run_eval_path(o, g, recs);
    }
    return EXIT_SUCCESS;
}
