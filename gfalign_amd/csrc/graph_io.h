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
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

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

inline bool read_gaf(const std::string &file, std::vector<GafRecord> &out, std::string &err)
{
    std::ifstream in(file);
    if (!in) {
        err = "cannot open " + file;
        return false;
    }
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        auto cols = split(line, '\t');
        if (cols.size() < 12) {
            err = "GAF record with fewer than 12 columns: " + line;
            return false;
        }
        GafRecord r;
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
        out.push_back(std::move(r));
    }
    return true;
}

}  // namespace gfal
#endif
