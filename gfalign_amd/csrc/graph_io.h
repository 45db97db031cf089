// graph_io.h -- host-side readers for the inputs of `gfalign search`,
// `evalPath` and `filter`: GFA1 S/L lines, GAF records, node lists.
//
// The reference delegates GFA parsing to its gfalibs submodule, which is not
// in /root/reference; the conventions below are the documented ones of
// SURVEY.md Appendix C.3 (pinned by validateFiles/test.6.tst only):
//   * uIds are assigned to segments in S-line order, starting at 0;
//   * adjacency: for every L line in file order, the forward edge
//     {or1, id2, or2} is appended to id1's list, then the reverse edge
//     {flip(or2), id1, flip(or1)} to id2's list unless already present
//     (in-tree analogue: reference src/alignments.cpp:369-382).
#ifndef GFALIGN_GRAPH_IO_H
#define GFALIGN_GRAPH_IO_H

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gfal {

inline std::vector<std::string> split(const std::string &line, char delim)
{
    std::vector<std::string> out;
    size_t start = 0;
    while (true) {
        size_t pos = line.find(delim, start);
        if (pos == std::string::npos) {
            out.push_back(line.substr(start));
            break;
        }
        out.push_back(line.substr(start, pos - start));
        start = pos + 1;
    }
    return out;
}

// What the search needs of gfalibs' Edge (orientation0, id, orientation1).
struct Edge {
    char from_orient;
    uint32_t to;
    char to_orient;
    bool operator==(const Edge &o) const
    {
        return from_orient == o.from_orient && to == o.to && to_orient == o.to_orient;
    }
};

struct Graph {
    std::vector<std::string> headers;                    // uId -> header
    std::unordered_map<std::string, uint32_t> ids;       // header -> uId
    std::vector<std::vector<Edge>> adjacency;            // per uId
    bool loaded = false;

    // reference: headersToIds[...] is operator[] on a hash map, so an unknown
    // header silently becomes uId 0 (src/alignments.cpp:86, src/eval.cpp:127).
    uint32_t id_or_zero(const std::string &header) const
    {
        auto it = ids.find(header);
        return it == ids.end() ? 0u : it->second;
    }
};

inline char flip(char o) { return o == '+' ? '-' : '+'; }

inline bool read_gfa(const std::string &file, Graph &g, std::string &err)
{
    std::ifstream in(file);
    if (!in) {
        err = "cannot open " + file;
        return false;
    }
    std::string line;
    std::vector<std::vector<std::string>> links;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == 'S') {
            auto cols = split(line, '\t');
            if (cols.size() < 2) continue;
            if (g.ids.emplace(cols[1], (uint32_t)g.headers.size()).second)
                g.headers.push_back(cols[1]);
        } else if (line[0] == 'L') {
            auto cols = split(line, '\t');
            if (cols.size() >= 5) links.push_back(cols);
        }
    }
    g.adjacency.assign(g.headers.size(), {});
    for (auto &l : links) {
        auto a = g.ids.find(l[1]), b = g.ids.find(l[3]);
        if (a == g.ids.end() || b == g.ids.end() || l[2].empty() || l[4].empty()) continue;
        Edge fw{l[2][0], b->second, l[4][0]};
        g.adjacency[a->second].push_back(fw);
        Edge rv{flip(l[4][0]), a->second, flip(l[2][0])};
        auto &lst = g.adjacency[b->second];
        bool present = false;
        for (auto &e : lst) present |= e == rv;
        if (!present) lst.push_back(rv);
    }
    g.loaded = true;
    return true;
}

// One GAF record (reference include/alignments.h:125-158).
struct GafRecord {
    std::string qname;
    unsigned qlen = 0, qstart = 0, qend = 0;
    char strand = '+';
    std::string path;
    unsigned plen = 0, pstart = 0, pend = 0, matches = 0, blocklen = 0, mapq = 0;
    std::vector<std::string> tags;   // as "LB:T:content"

    // reference src/alignments.cpp:51-73
    std::string print() const
    {
        std::string s = qname + "\t" + std::to_string(qlen) + "\t" + std::to_string(qstart) +
                        "\t" + std::to_string(qend) + "\t" + std::string(1, strand) + "\t" +
                        path + "\t" + std::to_string(plen) + "\t" + std::to_string(pstart) +
                        "\t" + std::to_string(pend) + "\t" + std::to_string(matches) + "\t" +
                        std::to_string(blocklen) + "\t" + std::to_string(mapq);
        for (auto &t : tags) s += "\t" + t;
        return s + "\n";
    }
};

// Node names of a GAF path string (">a<b" -> a, b) with their orientation.
// Mirrors the scanning loop of reference src/alignments.cpp:75-94.
inline void gaf_path_nodes(const std::string &path,
                           std::vector<std::pair<std::string, char>> &out)
{
    out.clear();
    size_t i = 0;
    while (i < path.size()) {
        char mark = path[i];
        size_t j = i + 1;
        while (j < path.size() && path[j] != '>' && path[j] != '<') ++j;
        out.emplace_back(path.substr(i + 1, j - i - 1), mark == '>' ? '+' : '-');
        i = j;
    }
}

// ---------------------------------------------------------------------------
// `search` needs one thing from the GAF: the path column of every record as
// packed steps (reference src/eval.cpp:123 getPaths -> src/alignments.cpp:75-94).
// This reader maps the file, cuts it into line-aligned pieces and parses them on
// several threads straight into the CSR the scorer takes; no GafRecord is built.
// It accepts and rejects the same files as read_gaf (12 columns, the numeric
// columns must parse as std::stoi would) and yields exactly what
// PackedAlignments::add yields record by record.
// ---------------------------------------------------------------------------
class HeaderIndex {     // header -> uId without building std::string keys
public:
    explicit HeaderIndex(const Graph &g) : g_(g)
    {
        size_t cap = 16;
        while (cap < 2 * g.headers.size() + 1) cap <<= 1;
        slot_.assign(cap, -1);
        for (size_t u = 0; u < g.headers.size(); ++u) {
            size_t at = hash(g.headers[u].data(), g.headers[u].size()) & (cap - 1);
            while (slot_[at] >= 0) at = (at + 1) & (cap - 1);
            slot_[at] = (int32_t)u;
        }
    }
    // unknown names alias uId 0, as Graph::id_or_zero does
    uint32_t id_or_zero(const char *p, size_t n) const
    {
        size_t at = hash(p, n) & (slot_.size() - 1);
        while (slot_[at] >= 0) {
            const std::string &h = g_.headers[(size_t)slot_[at]];
            if (h.size() == n && memcmp(h.data(), p, n) == 0) return (uint32_t)slot_[at];
            at = (at + 1) & (slot_.size() - 1);
        }
        return 0;
    }

private:
    static size_t hash(const char *p, size_t n)
    {
        uint64_t h = 1469598103934665603ull;     // FNV-1a
        for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
        return (size_t)(h ^ (h >> 29));
    }
    const Graph &g_;
    std::vector<int32_t> slot_;
};

// would std::stoi(text) succeed?  (leading blanks, a sign, at least one digit,
// value inside int)
inline bool parses_as_int(const char *p, const char *end)
{
    while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
    bool neg = false;
    if (p < end && (*p == '+' || *p == '-')) neg = *p++ == '-';
    if (p >= end || *p < '0' || *p > '9') return false;
    long long v = 0;
    for (; p < end && *p >= '0' && *p <= '9'; ++p) {
        v = v * 10 + (*p - '0');
        if (v > 2147483648ll) return false;
    }
    return neg ? v <= 2147483648ll : v <= 2147483647ll;
}

inline bool read_gaf_paths(const std::string &file, const Graph &g, std::vector<int32_t> &off,
                           std::vector<int32_t> &steps, std::string &err, unsigned n_threads = 0)
{
    off.assign(1, 0);
    steps.clear();
    int fd = open(file.c_str(), O_RDONLY);
    if (fd < 0) {
        err = "cannot open " + file;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        err = "cannot stat " + file;
        return false;
    }
    const size_t size = (size_t)st.st_size;
    if (size == 0) {
        close(fd);
        return true;
    }
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        err = "cannot map " + file;
        return false;
    }
    const char *data = static_cast<const char *>(map);
    if (n_threads == 0) n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    n_threads = (unsigned)std::min<size_t>(n_threads, size / (1 << 20) + 1);
    // piece t = [cut[t], cut[t + 1]), every cut just after a newline
    std::vector<size_t> cut(n_threads + 1, size);
    cut[0] = 0;
    for (unsigned t = 1; t < n_threads; ++t) {
        size_t at = std::max(cut[t - 1], size / n_threads * t);
        const void *nl = at < size ? memchr(data + at, '\n', size - at) : nullptr;
        cut[t] = nl ? (size_t)(static_cast<const char *>(nl) - data) + 1 : size;
    }
    struct Piece {
        std::vector<int32_t> lens, steps;
        std::string err;
    };
    std::vector<Piece> pieces(n_threads);
    const HeaderIndex index(g);
    auto parse = [&](unsigned t) {
        Piece &out = pieces[t];
        const char *p = data + cut[t], *const end = data + cut[t + 1];
        while (p < end) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            const char *line_end = nl ? nl : end;
            const char *e = line_end;
            if (e > p && e[-1] == '\r') --e;
            // columns: starts of the first 12, end of the path column
            const char *col[13];
            int n_cols = 1;
            col[0] = p;
            for (const char *q = p; q < e && n_cols < 13; ++q)
                if (*q == '\t') col[n_cols++] = q + 1;
            if (n_cols < 12) {
                out.err = "GAF record with fewer than 12 columns: " + std::string(p, e);
                return;
            }
            if (n_cols == 12) col[12] = e + 1;
            bool numeric = true;
            for (int c : {1, 2, 3, 6, 7, 8, 9, 10, 11}) numeric &= parses_as_int(col[c], col[c + 1] - 1);
            if (!numeric) {
                out.err = "malformed GAF record: " + std::string(p, e);
                return;
            }
            const char *q = col[5], *const path_end = col[6] - 1;
            int32_t len = 0;
            while (q < path_end) {       // src/alignments.cpp:75-94: marker, then the name
                const char mark = *q++;
                const char *name = q;
                while (q < path_end && *q != '>' && *q != '<') ++q;
                const uint32_t id = index.id_or_zero(name, (size_t)(q - name));
                out.steps.push_back((int32_t)((id << 1) | (mark == '>' ? 0u : 1u)));
                ++len;
            }
            out.lens.push_back(len);
            p = nl ? nl + 1 : end;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(parse, t);
    parse(0);
    for (auto &th : pool) th.join();
    munmap(map, size);
    size_t n_rec = 0, n_steps = 0;
    for (const Piece &pc : pieces) {
        if (!pc.err.empty()) {
            err = pc.err;
            return false;
        }
        n_rec += pc.lens.size();
        n_steps += pc.steps.size();
    }
    if (n_steps > (size_t)INT32_MAX) {
        err = "more than 2^31 alignment steps";
        return false;
    }
    off.reserve(n_rec + 1);
    steps.reserve(n_steps);
    for (const Piece &pc : pieces) {
        for (int32_t len : pc.lens) off.push_back(off.back() + len);
        steps.insert(steps.end(), pc.steps.begin(), pc.steps.end());
    }
    return true;
}

// One line of a GAF -> record (reference src/alignments.cpp:143-235): 12 columns
// at least, the numeric ones through std::stoi, tags as label[2]:type:content.
inline bool parse_gaf_record(const std::string &line, GafRecord &r, std::string &err)
{
    auto cols = split(line, '\t');
    if (cols.size() < 12) {
        err = "GAF record with fewer than 12 columns: " + line;
        return false;
    }
    try {
        r.qname = cols[0];
        r.qlen = (unsigned)std::stoi(cols[1]);
        r.qstart = (unsigned)std::stoi(cols[2]);
        r.qend = (unsigned)std::stoi(cols[3]);
        r.strand = cols[4].empty() ? '+' : cols[4][0];
        r.path = cols[5];
        r.plen = (unsigned)std::stoi(cols[6]);
        r.pstart = (unsigned)std::stoi(cols[7]);
        r.pend = (unsigned)std::stoi(cols[8]);
        r.matches = (unsigned)std::stoi(cols[9]);
        r.blocklen = (unsigned)std::stoi(cols[10]);
        r.mapq = (unsigned)std::stoi(cols[11]);
    } catch (const std::exception &) {
        err = "malformed GAF record: " + line;
        return false;
    }
    for (size_t c = 12; c < cols.size(); ++c) {
        // reference keeps label[2], type, content (src/alignments.cpp:221-229)
        auto t = split(cols[c], ':');
        if (t.size() >= 3 && t[0].size() >= 2 && !t[1].empty())
            r.tags.push_back(t[0].substr(0, 2) + ":" + t[1].substr(0, 1) + ":" + t[2]);
    }
    return true;
}

// All records of a GAF, in file order (filter, evalGFA, evalPath).  The file is
// mapped and cut into line-aligned pieces that are parsed on several threads
// (the reference loads alignments through its thread pool too,
// src/alignments.cpp:182); the first bad line in file order is the error.
inline bool read_gaf(const std::string &file, std::vector<GafRecord> &out, std::string &err,
                     unsigned n_threads = 0)
{
    int fd = open(file.c_str(), O_RDONLY);
    if (fd < 0) {
        err = "cannot open " + file;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        err = "cannot stat " + file;
        return false;
    }
    const size_t size = (size_t)st.st_size;
    if (size == 0) {
        close(fd);
        return true;
    }
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        err = "cannot map " + file;
        return false;
    }
    const char *data = static_cast<const char *>(map);
    if (n_threads == 0) n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    n_threads = (unsigned)std::min<size_t>(n_threads, size / (1 << 20) + 1);
    std::vector<size_t> cut(n_threads + 1, size);
    cut[0] = 0;
    for (unsigned t = 1; t < n_threads; ++t) {
        size_t at = std::max(cut[t - 1], size / n_threads * t);
        const void *nl = at < size ? memchr(data + at, '\n', size - at) : nullptr;
        cut[t] = nl ? (size_t)(static_cast<const char *>(nl) - data) + 1 : size;
    }
    std::vector<std::vector<GafRecord>> pieces(n_threads);
    std::vector<std::string> errs(n_threads);
    auto parse = [&](unsigned t) {
        const char *p = data + cut[t], *const end = data + cut[t + 1];
        std::string line;
        while (p < end) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            const char *e = nl ? nl : end;
            line.assign(p, e);
            if (!line.empty() && line.back() == '\r') line.pop_back();
            GafRecord r;
            if (!parse_gaf_record(line, r, errs[t])) return;
            pieces[t].push_back(std::move(r));
            p = nl ? nl + 1 : end;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(parse, t);
    parse(0);
    for (auto &th : pool) th.join();
    munmap(map, size);
    size_t total = 0;
    for (unsigned t = 0; t < n_threads; ++t) {
        if (!errs[t].empty()) {
            err = errs[t];
            return false;
        }
        total += pieces[t].size();
    }
    out.reserve(out.size() + total);
    for (auto &pc : pieces)
        for (auto &r : pc) out.push_back(std::move(r));
    return true;
}

}  // namespace gfal
#endif
