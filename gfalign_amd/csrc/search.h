// search.h -- host side of `gfalign search` and `gfalign evalPath`.
//
// The best-first expansion of reference src/eval.cpp:110-193 stays on the
// host; every call of evaluatePath (src/eval.cpp:162) goes through the C ABI
// of include/gfalign_scorer.h.  Scores are pure functions of the candidate
// path, so whole subtrees under the best queue entries are scored in one batch
// ahead of time (see class Search); entries are still popped, extended,
// enqueued and printed in exactly the reference's order.
#ifndef GFALIGN_SEARCH_H
#define GFALIGN_SEARCH_H

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gfalign_scorer.h"
#include "graph_io.h"

namespace gfal {

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

// reference include/alignments.h:11-21
struct Step {
    int32_t id;
    char orientation;
};

inline int32_t pack(const Step &s)
{
    if (s.orientation == '+') return s.id << 1;
    if (s.orientation == '-') return (s.id << 1) | 1;
    return GFAL_STEP_OTHER | (s.id << 1);
}

// reference include/nodetable.h: name -> {uId, count}; records keep insertion
// order here (lookups go through `index`), which nothing observable depends on.
struct NodeTable {
    struct Record {
        std::string name;
        uint32_t uid, count;
    };
    std::vector<Record> records;
    std::unordered_map<std::string, size_t> index;
    uint32_t node_count = 0;

    bool load(const std::string &file, const Graph &g)   // nodetable.h:16-43
    {
        std::ifstream in(file);
        std::string line;
        while (std::getline(in, line)) {
            auto cols = split(line, '\t');
            uint32_t count = 1;
            if (cols.size() > 1) {
                int c = std::atoi(cols[1].c_str());
                if (c < 1) continue;
                count = (uint32_t)c;
            }
            node_count += count;
            auto it = g.ids.find(cols[0]);
            if (it == g.ids.end()) {
                fprintf(stderr, "Error: node not in graph (pIUd: %s)\n", cols[0].c_str());
                return false;
            }
            insert(cols[0], it->second, count);
        }
        return true;
    }
    void insert(const std::string &name, uint32_t uid, uint32_t count)
    {
        if (index.count(name)) return;   // flat_hash_map::insert keeps the first
        index[name] = records.size();
        records.push_back({name, uid, count});
    }
    void add(const std::string &name, uint32_t uid, uint32_t count)   // :49-54
    {
        if (count < 1) return;
        insert(name, uid, count);
        node_count += count;
    }
    // nodetable.h:56-67
    bool hamiltonian(const std::vector<int32_t> &packed_path) const
    {
        if (packed_path.size() + 2 != node_count) return false;
        std::unordered_map<uint32_t, uint32_t> seen;
        for (int32_t s : packed_path) ++seen[(uint32_t)(s & ~GFAL_STEP_OTHER) >> 1];
        for (auto &r : records) {
            auto it = seen.find(r.uid);
            if (it == seen.end() || it->second != r.count) return false;
        }
        return true;
    }
};

inline std::string path_string(const std::vector<Step> &p, const Graph &g)
{
    std::string s;   // include/alignments.h:72-80
    for (size_t i = 0; i < p.size(); ++i) {
        s += g.headers[(size_t)p[i].id] + p[i].orientation;
        if (i + 1 < p.size()) s += ',';
    }
    return s;
}

// the same from packed steps: "utig4-1+,utig4-2-"
inline std::string path_string(const std::vector<int32_t> &p, const Graph &g)
{
    std::string s;
    for (size_t i = 0; i < p.size(); ++i) {
        s += g.headers[(size_t)((uint32_t)(p[i] & ~GFAL_STEP_OTHER) >> 1)];
        s += (p[i] & GFAL_STEP_OTHER) ? '0' : (p[i] & 1) ? '-' : '+';
        if (i + 1 < p.size()) s += ',';
    }
    return s;
}

inline uint32_t count_uniques(const std::vector<Step> &p)   // src/eval.cpp:153-160
{
    std::vector<int32_t> ids;
    ids.reserve(p.size());
    for (auto &s : p) ids.push_back(s.id);
    std::sort(ids.begin(), ids.end());
    return (uint32_t)(std::unique(ids.begin(), ids.end()) - ids.begin());
}

// Alignments as the packed CSR the scorer wants (src/eval.cpp:123 getPaths).
struct PackedAlignments {
    std::vector<int32_t> off{0};
    std::vector<int32_t> steps;
    void add(const GafRecord &r, const Graph &g)
    {
        std::vector<std::pair<std::string, char>> nodes;
        gaf_path_nodes(r.path, nodes);
        for (auto &nd : nodes)
            steps.push_back((int32_t)((g.id_or_zero(nd.first) << 1) | (nd.second == '-')));
        off.push_back((int32_t)steps.size());
    }
    int64_t size() const { return (int64_t)off.size() - 1; }
};

// The alignments, sharded over one or more MI355X: one
// gfal_scorer per device (each keeps its share of the set, see open()), grouped
// (gfal_group): every device scores the whole batch against its shard -- the
// kernels of all devices are enqueued before anything is waited for -- and the
// per-path integer counters are summed by one RCCL all-reduce over xGMI.
// Scoring with zero alignments never touches a device (the loop of
// src/eval.cpp:80 has no iterations: all counters are zero).
class PathScorer {
public:
    ~PathScorer()
    {
        if (group_) gfal_group_destroy(group_);
        for (gfal_scorer *h : shards_)
            if (h) gfal_scorer_destroy(h);
    }
    // universe: the nodes candidate paths may visit (node list + source +
    // destination in search; the path's nodes in evalPath).
    // devices: first device and how many to use; share_device puts every shard
    // on `first_device` (test rigs with one GPU).
    bool open(const PackedAlignments &a, int32_t n_nodes, int first_device,
              const std::vector<int32_t> &universe, int n_devices = 1,
              bool share_device = false)
    {
        n_aln_ = a.size();
        if (n_aln_ == 0) return true;
        n_devices = (int)std::max<int64_t>(1, std::min<int64_t>(n_devices, n_aln_));
        // every device gets the whole set and keeps its share of the groups of the
        // scorer's own sorted order (gfal_scorer_create_sharded): the shards
        // partition the set and balance by construction.  Creation (a host-side
        // sort each) runs on all devices at once.
        shards_.assign((size_t)n_devices, nullptr);
        std::vector<int> rcs((size_t)n_devices, GFAL_OK);
        std::vector<std::string> errs((size_t)n_devices);
        // identical alignments are collapsed into weighted lanes (same output; a GAF repeats
        // the same node paths: 6x fewer lanes at config 3, the search loop of config 5 takes
        // 0.085 s instead of 0.175 s).  GFALIGN_DEDUP=0: one lane per alignment.
        const bool dedup = getenv("GFALIGN_DEDUP") == nullptr || atoi(getenv("GFALIGN_DEDUP")) != 0;
        auto make = [&](int d) {
            rcs[(size_t)d] = (dedup ? gfal_scorer_create_dedup : gfal_scorer_create_sharded)(
                a.off.data(), a.steps.data(), n_aln_, n_nodes,
                share_device ? first_device : first_device + d, universe.data(),
                (int32_t)universe.size(), d, n_devices, &shards_[(size_t)d]);
            if (rcs[(size_t)d] != GFAL_OK) errs[(size_t)d] = gfal_last_error();
        };
        std::vector<std::thread> threads;
        for (int d = 1; d < n_devices; ++d) threads.emplace_back(make, d);
        make(0);
        for (auto &t : threads) t.join();
        for (int d = 0; d < n_devices; ++d)
            if (rcs[(size_t)d] != GFAL_OK) {
                fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rcs[(size_t)d]),
                        errs[(size_t)d].c_str());
                return false;
            }
        // the shards as one group: counters summed on the devices by one RCCL
        // all-reduce over xGMI (on the host if RCCL cannot serve the group, e.g. all
        // shards on one device; a group of one has nothing to sum)
        const int rc = gfal_group_create(shards_.data(), n_devices, &group_);
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer group: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        return true;
    }
    bool uses_rccl() const { return group_ && gfal_group_uses_rccl(group_); }
    // A batch in two halves: begin() hands it to the devices and returns (the
    // vectors may be reused at once), end() waits for the counters.
    bool begin(const std::vector<int32_t> &off, const std::vector<int32_t> &steps, bool filter)
    {
        pending_paths_ = off.size() - 1;
        if (n_aln_ == 0 || pending_paths_ == 0) return true;
        const int rc = gfal_group_score_begin(group_, off.data(), steps.data(), (int32_t)pending_paths_,
                                              filter ? 1 : 0);
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        return true;
    }
    bool end(std::vector<uint32_t> &bad, std::vector<uint32_t> &good)
    {
        const size_t P = pending_paths_;
        bad.assign(P, 0);
        good.assign(P, 0);
        if (n_aln_ == 0 || P == 0) return true;
        const int rc = gfal_group_score_end(group_, bad.data(), good.data(), nullptr);
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        for (gfal_scorer *h : shards_) {
            gfal_info info;
            if (gfal_scorer_get_info(h, &info) == GFAL_OK) dp_pairs_ += (uint64_t)info.dp_pairs;
        }
        return true;
    }
    bool score(const std::vector<int32_t> &off, const std::vector<int32_t> &steps, bool filter,
               std::vector<uint32_t> &bad, std::vector<uint32_t> &good)
    {
        return begin(off, steps, filter) && end(bad, good);
    }
    // Search mode (gfal_group_score_children): scored paths stay on the devices and a
    // candidate is scored from its parent.  The longest alignment of any shard bounds
    // the parents that may be used that way.
    int32_t max_aln_len() const
    {
        int32_t m = 0;
        for (gfal_scorer *h : shards_) {
            gfal_info info;
            if (gfal_scorer_get_info(h, &info) == GFAL_OK) m = std::max(m, info.max_aln_len);
        }
        return m;
    }
    bool store_reserve(int64_t n_slots)
    {
        if (n_aln_ == 0) return true;
        const int rc = gfal_group_store_reserve(group_, n_slots);
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        return true;
    }
    bool begin_store(const std::vector<int32_t> &off, const std::vector<int32_t> &steps,
                     const std::vector<int32_t> &slots)
    {
        pending_paths_ = off.size() - 1;
        if (n_aln_ == 0 || pending_paths_ == 0) return true;
        const int rc = gfal_group_score_store_begin(group_, off.data(), steps.data(), (int32_t)pending_paths_,
                                                    slots.data());
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        return true;
    }
    // the batch in flight has finished on the devices (never blocks)
    bool done() const { return n_aln_ == 0 || !group_ || gfal_group_score_poll(group_) != 0; }
    bool begin_children(const std::vector<int32_t> &parent, const std::vector<int32_t> &step,
                        const std::vector<int32_t> &slot, int32_t max_len)
    {
        pending_paths_ = parent.size();
        if (n_aln_ == 0 || pending_paths_ == 0) return true;
        const int rc = gfal_group_score_children_begin(group_, (int32_t)pending_paths_, parent.data(),
                                                       step.data(), slot.data(), max_len);
        if (rc != GFAL_OK) {
            fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(rc), gfal_last_error());
            return false;
        }
        return true;
    }
    uint64_t dp_pairs() const { return dp_pairs_; }   // pairs that needed the exact DP so far
    // fw / rc traceback scores of one path against every alignment, in input order
    bool pair_scores(const std::vector<int32_t> &path, std::vector<int32_t> &fw,
                     std::vector<int32_t> &rc)
    {
        fw.assign((size_t)n_aln_, 0);
        rc.assign((size_t)n_aln_, 0);
        for (gfal_scorer *h : shards_) {   // every shard writes the entries of its own alignments
            int err = gfal_scorer_pair_scores(h, path.data(), (int32_t)path.size(), fw.data(),
                                              rc.data());
            if (err != GFAL_OK) {
                fprintf(stderr, "Error: scorer: %s (%s)\n", gfal_strerror(err), gfal_last_error());
                return false;
            }
        }
        return true;
    }
    int64_t n_aln() const { return n_aln_; }
    size_t n_shards() const { return shards_.size(); }

private:
    std::vector<gfal_scorer *> shards_;
    gfal_group *group_ = nullptr;
    size_t pending_paths_ = 0;
    int64_t n_aln_ = 0;
    uint64_t dp_pairs_ = 0;
};

struct SearchOptions {
    std::string node_file, source, destination;
    uint32_t max_steps = 100000;   // include/input-gfalign.h:12
    uint32_t min_nodes = 0;
    bool return_all_paths = false;
    size_t speculate = 128;        // candidate paths scored per GPU batch (target)
    // GFALIGN_PREFETCH=1: keep one further batch in flight while the host pops.  Measured
    // on config 3 (-m 20000): the batch in flight holds what is needed next 88 % of the
    // time, but the search stalls every few pops, so there is little host work to hide and
    // the extra batches cost more than they save (0.24 s against 0.19 s): off by default.
    bool prefetch = false;
    // Candidates scored from their parents on the device (same counters, less work;
    // gfal_group_score_children).  GFALIGN_INCREMENTAL=0 scores every candidate in full.
    bool incremental = true;
};

// reference src/eval.cpp:110-193
//
// Batching.  The reference scores the extensions of one queue entry per step
// (src/eval.cpp:162), a handful of paths.  Scores are pure functions of the
// path, and so are the extensions themselves (orientation gate, budgets), so
// whenever the front entry has no scored extensions yet, its whole subtree is
// generated breadth-first -- and that of the next best queue entries -- until
// about `speculate` candidate paths are collected; they are scored in ONE call
// (gfal_group_score_begin / _end: optionally the next batch is handed to the
// devices before the host returns to popping, see SearchOptions::prefetch).
// Entries are still popped, extended, enqueued and printed in exactly the
// reference's order; speculation that is never popped is only wasted work.
class Search {
public:
    Search(const Graph &g, PathScorer &scorer, const SearchOptions &opt, std::ostream &out)
        : g_(g), scorer_(scorer), opt_(opt), out_(out)
    {
        if (const char *f = getenv("GFALIGN_DUMP_BATCHES")) dump_ = fopen(f, "wb");
        if (const char *k = getenv("GFALIGN_SPEC_POLICY")) best_first_ = std::string(k) == "best";
        if (const char *k = getenv("GFALIGN_SPEC_MIN_LIKE")) min_like_ = (float)atof(k);
        if (const char *k = getenv("GFALIGN_SPEC_DEPTH")) max_depth_ = std::max(1, std::min(62, atoi(k)));
    }
    // Which extensions the speculation follows first (collect_best): an edge that many
    // alignments take is the likelier continuation.  Counted over (a sample of) the
    // alignments' consecutive steps, both strands; has no influence on the output.
    void set_alignments(const PackedAlignments &a)
    {
        if (!best_first_) return;
        const size_t V = g_.adjacency.size();
        edge_w_.assign(V, {});
        size_t n_edges = 0;
        for (size_t u = 0; u < V; ++u) {
            edge_w_[u].assign(g_.adjacency[u].size(), 0u);
            n_edges += g_.adjacency[u].size();
        }
        if (n_edges == 0 || a.size() == 0) return;
        // (from step, to step) -> edge, open addressing
        size_t cap = 16;
        while (cap < 2 * n_edges) cap <<= 1;
        std::vector<uint64_t> keys(cap, ~0ull);
        std::vector<uint32_t *> vals(cap, nullptr);
        auto slot_of = [&](uint64_t k) {
            size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (keys[h] != ~0ull && keys[h] != k) h = (h + 1) & (cap - 1);
            return h;
        };
        for (size_t u = 0; u < V; ++u)
            for (size_t i = 0; i < g_.adjacency[u].size(); ++i) {
                const Edge &e = g_.adjacency[u][i];
                const uint64_t x = ((uint64_t)u << 1) | (e.from_orient == '-');
                const uint64_t y = ((uint64_t)e.to << 1) | (e.to_orient == '-');
                const size_t h = slot_of((x << 32) | y);
                if (keys[h] == ~0ull) {          // (a duplicate L line: the first edge counts)
                    keys[h] = (x << 32) | y;
                    vals[h] = &edge_w_[u][i];
                }
            }
        const int64_t n = a.size();
        const int64_t stride = std::max<int64_t>(1, (int64_t)a.steps.size() / 4000000);
        for (int64_t k = 0; k < n; k += stride)
            for (int32_t j = a.off[(size_t)k]; j + 1 < a.off[(size_t)k + 1]; ++j) {
                const uint64_t x = (uint32_t)a.steps[(size_t)j], y = (uint32_t)a.steps[(size_t)j + 1];
                size_t h = slot_of((x << 32) | y);
                if (vals[h]) ++*vals[h];
                h = slot_of(((y ^ 1u) << 32) | (x ^ 1u));          // the other strand
                if (vals[h]) ++*vals[h];
            }
    }
    ~Search()
    {
        if (dump_) fclose(dump_);
    }

    int run()
    {
        NodeTable table;
        if (!table.load(opt_.node_file, g_)) return EXIT_FAILURE;           // :126
        table.add(opt_.source, g_.id_or_zero(opt_.source), 1);             // :127
        table.add(opt_.destination, g_.id_or_zero(opt_.destination), 1);   // :128
        dest_uid_ = table.records[table.index.at(opt_.destination)].uid;
        const uint32_t src_uid = table.records[table.index.at(opt_.source)].uid;

        // per-uid record index for the budget gate (:142-143); -1 = not listed
        record_of_.assign(g_.headers.size(), -1);
        for (size_t r = 0; r < table.records.size(); ++r) {
            // the gate looks the header up by NAME; a record whose name is not
            // that uid's header (unknown source aliased to uid 0) never matches
            if (table.records[r].uid < g_.headers.size() &&
                g_.headers[table.records[r].uid] == table.records[r].name)
                record_of_[table.records[r].uid] = (int)r;
        }

        // extension budget per uid (:142-143): the record's count, 0 = not listed
        allowance_.assign(g_.headers.size(), 0);
        for (size_t u = 0; u < record_of_.size(); ++u)
            if (record_of_[u] >= 0) allowance_[u] = table.records[(size_t)record_of_[u]].count;

        incr_ = opt_.incremental && !opt_.prefetch && scorer_.n_aln() > 0;
        if (incr_) {
            min_parent_ = (size_t)std::max<int32_t>(1, scorer_.max_aln_len());
            store_cap_ = 1 << 14;
            if (!scorer_.store_reserve(store_cap_)) {
                // (no room for the device-side path store: every candidate is scored in full
                // instead -- same rows, more work)
                fprintf(stderr, "gfalign: no device memory for the path store, candidates are scored in full\n");
                incr_ = false;
            }
        }

        auto first = std::make_unique<Node>();
        first->set_path({GFAL_STEP_OTHER | (int32_t)(src_uid << 1)});       // :130, orientation '0'
        first->uniques = 1;
        first->scored = true;           // (never scored: its key is 0 by definition, :132)
        queue_.emplace(Key{0, seq_++}, std::move(first));                   // :132

        uint64_t path_counter = 0;
        uint32_t steps = 0, best_uniques = 0;
        int32_t best_alt = INT32_MAX;
        while (!queue_.empty() && steps < opt_.max_steps) {                 // :134
            if (!queue_.begin()->second->kids_scored && !score_front()) return EXIT_FAILURE;
            std::unique_ptr<Node> u = std::move(queue_.begin()->second);    // :135
            queue_.erase(queue_.begin());
            needed_ += u->kids.size();      // (the reference's evaluatePath calls, :162)
            for (std::unique_ptr<Node> &c : u->kids) {                      // :136-185
                const int32_t alt =
                    (int32_t)c->bad - (int32_t)c->good - (int32_t)c->uniques;   // :163
                if (!c->at_destination) {                                   // :165-169
                    queue_.emplace(Key{alt, seq_++}, std::move(c));
                } else {                                                    // :170-184
                    ++path_counter;
                    const bool ham = table.hamiltonian(c->path());
                    bool print = false;
                    if (c->uniques >= opt_.min_nodes &&
                        (best_uniques < c->uniques ||
                         (best_uniques == c->uniques && best_alt > alt))) {
                        best_alt = alt;
                        best_uniques = c->uniques;
                        print = true;
                    }
                    if (opt_.return_all_paths || print)
                        out_ << path_counter << '\t' << c->bad << '\t' << c->good << '\t' << alt
                             << '\t' << c->len << '\t' << c->uniques << '\t'
                             << (ham ? 'T' : 'F') << '\t' << path_string(c->path(), g_)
                             << std::endl;
                }
            }
            ++steps;                                                        // :187
        }
        if (steps >= opt_.max_steps)                                        // :190-191
            out_ << "Reached maximum number of steps (" << steps << ")" << std::endl;
        if (in_flight_ && !finish()) return EXIT_FAILURE;                   // (drain the device)
        return EXIT_SUCCESS;
    }

    uint64_t scored_paths() const { return scored_; }
    // candidates whose scores the search went on to use: the extensions of the entries it
    // popped (one evaluatePath call each in the reference, src/eval.cpp:146-162); the rest
    // of scored_paths() was speculation that was never popped
    uint64_t needed_paths() const { return needed_; }
    uint64_t scored_in_full() const { return incr_ ? full_scored_ : scored_; }
    uint64_t batches() const { return batches_; }
    uint64_t prefetch_hits() const { return prefetch_hits_; }
    uint64_t prefetch_misses() const { return prefetch_misses_; }
    double collect_seconds() const { return t_collect_; }
    double ahead_seconds() const { return t_ahead_; }
    uint64_t made_ahead() const { return made_ahead_; }
    double score_seconds() const { return t_score_; }

private:
    // A candidate path: a queue entry, or a pre-generated extension of one.
    // The per-path NodeTable copy of the reference (include/alignments.h:26) is not
    // stored: a record's remaining count is its initial count minus the node's
    // occurrences in path[1..] (every non-destination extension decrements it once,
    // :166-167; the source at path[0] was never an extension), see make_kids.
    struct Node {
        // The path (packed steps, as the scorer takes them): written out once somebody
        // needs all of it -- the entry is extended, printed, or scored in full -- and
        // until then the parent's path (shared by its extensions) plus one step.  Most
        // candidates are scored from their parents on the device and never popped.
        std::shared_ptr<std::vector<int32_t>> full, base;
        int32_t last = 0;
        uint32_t len = 0;
        void set_path(std::vector<int32_t> p)
        {
            len = (uint32_t)p.size();
            last = p.back();
            full = std::make_shared<std::vector<int32_t>>(std::move(p));
            base.reset();
        }
        const std::vector<int32_t> &path()
        {
            if (!full) {
                full = std::make_shared<std::vector<int32_t>>();
                full->reserve(len);
                full->assign(base->begin(), base->end());
                full->push_back(last);
                base.reset();
            }
            return *full;
        }
        uint32_t uniques = 0, bad = 0, good = 0;
        bool at_destination = false;
        bool scored = false;            // bad / good are known (queue entries, finished batches)
        bool kids_made = false, kids_scored = false;
        bool kids_batched = false;      // the extensions have been handed to a batch
        std::vector<std::unique_ptr<Node>> kids;   // extensions, adjacency order
        // search mode: where the path is kept on the devices (-1: nowhere), and where
        // the entry sits in the batch being collected (>= 0: the children sub-batch,
        // < 0: ~index in the sub-batch scored in full)
        int32_t slot = -1, batch_at = 0;
        int32_t batch_depth = 0;   // ancestors inside the children sub-batch being collected
        uint32_t edge_w = 0;       // alignments that take the edge this extension was made by (set_alignments)
        float like = 0.f;          // collect_best: how likely the search is to pop this entry soon
    };
    struct Key {
        int32_t alt;
        uint64_t seq;   // FIFO among equal keys
        bool operator<(const Key &o) const
        {
            return alt != o.alt ? alt < o.alt : seq < o.seq;
        }
    };

    // Extensions of one entry, in adjacency order (:136-151, :165-167).
    void make_kids(Node &e) const
    {
        e.kids_made = true;
        const std::vector<int32_t> &ep = e.path();
        const int32_t last = ep.back();
        const bool fresh = (last & GFAL_STEP_OTHER) != 0;          // orientation still '0'
        const uint32_t last_id = (uint32_t)(last & ~GFAL_STEP_OTHER) >> 1;
        const char last_orient = (last & 1) ? '-' : '+';
        const size_t n = ep.size();
        const std::vector<Edge> &adj = g_.adjacency[last_id];
        for (size_t vi = 0; vi < adj.size(); ++vi) {
            const Edge &v = adj[vi];
            if (!fresh && last_orient != v.from_orient) continue;                         // :137
            const uint32_t allowed = allowance_[v.to];
            if (allowed == 0) continue;                                                   // :142-143
            // times this node was stepped on so far; ids compare with the
            // orientation bit masked off (path[0] may still carry the '0' flag)
            uint32_t used = 0;
            {   // (steps after the first are plain (id << 1) | minus: a loop the compiler vectorises)
                const int32_t *pp = ep.data();
                const int32_t want = (int32_t)(v.to << 1);
                for (size_t i = 1; i < n; ++i) used += (uint32_t)((pp[i] & ~1) == want);
            }
            if (used >= allowed) continue;
            const bool seen = used > 0 || ((uint32_t)(ep[0] & ~GFAL_STEP_OTHER) >> 1) == v.to;
            auto c = std::make_unique<Node>();
            const int32_t step = (int32_t)(v.to << 1) | (v.to_orient == '-');
            if (fresh) {          // the source takes the orientation of the edge it leaves by (:148-149)
                std::vector<int32_t> p(ep);
                p.back() = (int32_t)(last_id << 1) | (v.from_orient == '-');
                p.push_back(step);
                c->set_path(std::move(p));
            } else {
                c->base = e.full;
                c->last = step;
                c->len = (uint32_t)n + 1;
            }
            c->uniques = e.uniques + (seen ? 0u : 1u);                                    // :153-160
            c->at_destination = v.to == dest_uid_;
            if (!edge_w_.empty()) c->edge_w = edge_w_[last_id][vi];
            e.kids.push_back(std::move(c));
        }
    }

    int32_t key_of(const Node &n) const
    {
        return (int32_t)n.bad - (int32_t)n.good - (int32_t)n.uniques;        // :163
    }

    // Collect the next batch: breadth-first under the best queue entries (front
    // first), through the nodes that already have their extensions -- scored or in
    // flight: extending a path needs no counters -- down to the first nodes without,
    // until about `budget` candidates are gathered.  The search mostly dives under
    // its front entry, so the levels below it are what it asks for next.
    void collect(size_t budget)
    {
        if (best_first_) return collect_best(budget);
        const double t0 = now_s();
        size_t n_paths = begin_collect();
        level_.clear();
        auto it = queue_.begin();
        // roots: the front entry always; further entries while there is room
        while (it != queue_.end() && (level_.empty() || n_paths + level_.size() < budget / 4)) {
            level_.push_back(it->second.get());
            ++it;
            if (level_.size() >= 64) break;
        }
        Node *front = queue_.begin()->second.get();
        while (!level_.empty() && n_paths < budget) {
            next_.clear();
            for (Node *e : level_) {
                if (n_paths >= budget && e != front) break;
                if (!e->kids_batched) n_paths += batch_kids(e);
                // (a chain of in-batch parents is walked by one device thread per child:
                // beyond 48 levels the rest of the dive waits for the next batch)
                for (auto &c : e->kids)
                    if (!c->at_destination && c->len < GFAL_MAX_STEPS && c->batch_depth < 48)
                        next_.push_back(c.get());
            }
            level_.swap(next_);
        }
        finish_collect(n_paths, t0);
    }

    // The extensions of e join the batch being collected; returns how many.
    size_t batch_kids(Node *e)
    {
        if (!e->kids_made) make_kids(*e);     // (often made already: see make_ahead)
        e->kids_batched = true;
        batch_parents_.push_back(e);
        // search mode: a kid is scored from its parent when the parent is long
        // enough (no alignment longer than it) and is on the devices -- kept
        // there by an earlier batch, or itself a child of this one
        const bool from_parent = incr_ && e->len >= min_parent_ &&
                                 !(e->last & GFAL_STEP_OTHER) &&
                                 (e->slot >= 0 || e->batch_at > 0);
        for (auto &c : e->kids) {
            if (incr_ && !c->at_destination && c->len < GFAL_MAX_STEPS && c->len >= min_parent_)
                c->slot = take_slot();
            if (from_parent) {
                // (batch_at > 0: the parent is entry batch_at - 1 of this very sub-batch)
                ch_parent_.push_back(e->batch_at > 0 ? ~(e->batch_at - 1) : e->slot);
                ch_step_.push_back(c->last);
                ch_slot_.push_back(c->slot);
                ch_max_len_ = std::max(ch_max_len_, (int32_t)c->len);
                c->batch_at = (int32_t)ch_parent_.size();
                c->batch_depth = e->batch_at > 0 ? e->batch_depth + 1 : 0;
            } else {
                const std::vector<int32_t> &cp = c->path();
                batch_steps_.insert(batch_steps_.end(), cp.begin(), cp.end());
                batch_off_.push_back((int32_t)batch_steps_.size());
                batch_slots_.push_back(c->slot);
                c->batch_at = -(int32_t)batch_slots_.size();
            }
        }
        return e->kids.size();
    }

    // Best-first speculation (GFALIGN_SPEC_POLICY=best; the default is the level-by-level walk
    // above).  Breadth-first spends most of a batch on subtrees the search never enters -- 65 %
    // of the candidates at config 3: every side branch of the dive is followed as deep as the
    // dive itself.  Here every entry carries an estimate of how likely the search is to pop it
    // soon: the front 1, an extension its parent's estimate times its share of the alignments
    // that leave the parent's last step (edge_w_, squared: the search takes the best one), known
    // keys instead where the extensions are scored already.  The likeliest entry whose extensions
    // are not in a batch yet is expanded next, until the budget is spent or only long shots are
    // left.  A candidate's counters do not depend on how it was chosen: same output.
    // Measured (scripts/spec_policy.sh, -m 20000): a third fewer candidates (config 3: 42 071
    // instead of 62 206, config 5: 27 810 instead of 40 233) but MORE batches (662 / 388 against
    // 486 / 315; 510 / 310 with GFALIGN_SPEC_MIN_LIKE=0.0005) -- one wrong guess at a branch
    // ends a batch's usefulness, where the breadth-first batch holds every branch -- and the
    // loop is bound by the ~85 us a scoring call takes, not by the candidates in it: no faster,
    // so it is not the default.
    void collect_best(size_t budget)
    {
        const double t0 = now_s();
        size_t n_paths = begin_collect();
        spec_heap_.clear();
        uint64_t seq = 0;
        Node *front = queue_.begin()->second.get();
        {
            float like = 1.f;
            size_t k = 0;
            for (auto it = queue_.begin(); it != queue_.end() && k < 32; ++it, ++k) {
                it->second->like = like;
                spec_heap_.push_back(Spec{like, seq++, it->second.get()});
                like = k == 0 ? 0.2f : like * 0.7f;       // (the other queue entries wait for the dive to fail)
            }
            std::make_heap(spec_heap_.begin(), spec_heap_.end());
        }
        while (!spec_heap_.empty()) {
            std::pop_heap(spec_heap_.begin(), spec_heap_.end());
            Node *e = spec_heap_.back().n;
            spec_heap_.pop_back();
            if (e != front && (n_paths >= budget || e->like < min_like_)) break;
            if (!e->kids_batched) n_paths += batch_kids(e);
            if (e->kids.empty()) continue;
            // the extensions' shares
            double total = 0;
            int32_t best_key = INT32_MAX;
            for (auto &c : e->kids) {
                if (e->kids_scored) best_key = std::min(best_key, key_of(*c));
                const double w = (double)c->edge_w + 1.0;
                total += w * w;
            }
            bool best_taken = false;
            for (auto &c : e->kids) {
                if (c->at_destination || c->len >= GFAL_MAX_STEPS || c->batch_depth >= max_depth_) continue;
                float share;
                if (e->kids_scored) {       // (keys known: the first of the smallest is popped first)
                    const bool best = !best_taken && key_of(*c) == best_key;
                    best_taken |= best;
                    share = best ? 0.9f : 0.1f / (float)e->kids.size();
                } else {
                    const double w = (double)c->edge_w + 1.0;
                    share = (float)(w * w / total);
                }
                c->like = e->like * share;
                spec_heap_.push_back(Spec{c->like, seq++, c.get()});
                std::push_heap(spec_heap_.begin(), spec_heap_.end());
            }
        }
        finish_collect(n_paths, t0);
    }

    size_t begin_collect()
    {
        batch_parents_.clear();
        batch_off_.assign(1, 0);
        batch_steps_.clear();
        batch_slots_.clear();
        ch_parent_.clear();
        ch_step_.clear();
        ch_slot_.clear();
        ch_max_len_ = 2;
        return 0;
    }

    void finish_collect(size_t n_paths, double t0)
    {
        if (dump_ && n_paths) {   // GFALIGN_DUMP_BATCHES: the candidate batches (full paths), for benchmarks
            std::vector<int32_t> off{0}, steps;
            for (Node *e : batch_parents_)
                for (auto &c : e->kids) {
                    steps.insert(steps.end(), c->path().begin(), c->path().end());
                    off.push_back((int32_t)steps.size());
                }
            const int32_t head[2] = {(int32_t)n_paths, (int32_t)steps.size()};
            fwrite(head, sizeof(int32_t), 2, dump_);
            fwrite(off.data(), sizeof(int32_t), off.size(), dump_);
            fwrite(steps.data(), sizeof(int32_t), steps.size(), dump_);
        }
        t_collect_ += now_s() - t0;
    }

    bool submit()
    {
        if (batch_off_.size() <= 1 && ch_parent_.empty()) return true;            // nothing to score
        const double t0 = now_s();
        if (incr_) {
            // the sub-batch scored in full first (its paths are kept: children of this
            // very collect may name them), then the children; both waited for here
            bad_.clear();
            good_.clear();
            ch_bad_.clear();
            ch_good_.clear();
            if (batch_off_.size() > 1 &&
                !(scorer_.begin_store(batch_off_, batch_steps_, batch_slots_) && scorer_.end(bad_, good_)))
                return false;
            full_scored_ += batch_off_.size() - 1;
            if (!ch_parent_.empty()) {
                if (!scorer_.begin_children(ch_parent_, ch_step_, ch_slot_, ch_max_len_)) return false;
                make_ahead();
                if (!scorer_.end(ch_bad_, ch_good_)) return false;
            }
            results_ready_ = true;
        } else if (!scorer_.begin(batch_off_, batch_steps_, true)) {     // :162
            return false;
        }
        flight_parents_.swap(batch_parents_);
        in_flight_ = true;
        t_score_ += now_s() - t0;
        return true;
    }

    // While the devices score the batch: the extensions of its candidates (the search
    // mostly dives, so these are what the next batch asks for) are made now -- an
    // extension needs its parent's path, not its counters -- until the devices are done.
    // Work that would otherwise sit between two scoring calls; what is never asked for
    // is only memory.
    void make_ahead()
    {
        const double t0 = now_s();
        unsigned since_poll = 0;
        for (Node *e : batch_parents_) {
            for (auto &c : e->kids) {
                if (c->kids_made || c->at_destination || c->len >= GFAL_MAX_STEPS) continue;
                if (since_poll++ % 8 == 0 && scorer_.done()) {      // (a poll costs about as much as an extension)
                    t_ahead_ += now_s() - t0;
                    return;
                }
                make_kids(*c);
                ++made_ahead_;
            }
        }
        t_ahead_ += now_s() - t0;
    }

    bool finish()
    {
        const double t0 = now_s();
        if (!results_ready_ && !scorer_.end(bad_, good_)) return false;
        results_ready_ = false;
        size_t k = 0;
        for (Node *e : flight_parents_) {
            for (auto &c : e->kids) {
                if (incr_) {
                    const bool ch = c->batch_at > 0;
                    const size_t at = ch ? (size_t)(c->batch_at - 1) : (size_t)(-c->batch_at - 1);
                    c->bad = ch ? ch_bad_[at] : bad_[at];
                    c->good = ch ? ch_good_[at] : good_[at];
                    c->batch_at = 0;
                    c->batch_depth = 0;
                } else {
                    c->bad = bad_[k];
                    c->good = good_[k];
                }
                c->scored = true;
                ++k;
            }
            e->kids_scored = true;
            if (e->slot >= 0) {          // its extensions exist: nothing names this path again
                free_slots_.push_back(e->slot);
                e->slot = -1;
            }
        }
        flight_parents_.clear();
        in_flight_ = false;
        scored_ += k;
        ++batches_;
        t_score_ += now_s() - t0;
        return true;
    }

    // The front entry needs its extensions' counters.  They are in the batch in
    // flight, or a batch is made for them now; either way the next batch goes out
    // before the host returns to popping.
    bool score_front()
    {
        Node *f = queue_.begin()->second.get();
        if (in_flight_) {
            ++(f->kids_made ? prefetch_hits_ : prefetch_misses_);
            if (!finish()) return false;
        }
        if (!f->kids_scored) {
            collect(opt_.speculate);
            if (!submit()) return false;
            if (in_flight_ && !finish()) return false;
            if (!f->kids_scored) f->kids_scored = true;     // (no extensions at all)
        }
        if (opt_.prefetch && opt_.speculate > 1) {
            collect(opt_.speculate);
            if (!submit()) return false;
        }
        return true;
    }

    const Graph &g_;
    PathScorer &scorer_;
    SearchOptions opt_;
    std::ostream &out_;
    uint32_t dest_uid_ = 0;
    std::vector<int> record_of_;
    std::vector<uint32_t> allowance_;
    std::map<Key, std::unique_ptr<Node>> queue_;
    uint64_t seq_ = 0, scored_ = 0, batches_ = 0, needed_ = 0;
    double t_collect_ = 0, t_score_ = 0, t_ahead_ = 0;
    uint64_t made_ahead_ = 0;
    // the batch being collected, and the parents of the one the devices are scoring
    std::vector<Node *> batch_parents_, flight_parents_, level_, next_;
    uint64_t prefetch_hits_ = 0, prefetch_misses_ = 0;
    std::vector<int32_t> batch_off_, batch_steps_;
    std::vector<uint32_t> bad_, good_;
    // search mode: the children sub-batch of the collect, the slots of the other one,
    // and the slots of the device-side path store
    bool incr_ = false, results_ready_ = false;
    size_t min_parent_ = 1;
    std::vector<int32_t> batch_slots_, ch_parent_, ch_step_, ch_slot_, free_slots_;
    std::vector<uint32_t> ch_bad_, ch_good_;
    int32_t ch_max_len_ = 2, next_slot_ = 0;
    int64_t store_cap_ = 0;
    uint64_t full_scored_ = 0;
    bool store_failed_ = false;
    int32_t take_slot()
    {
        if (!free_slots_.empty()) {
            const int32_t s = free_slots_.back();
            free_slots_.pop_back();
            return s;
        }
        if (next_slot_ >= store_cap_) {        // (no batch is in flight while one is collected)
            if (store_failed_ || !scorer_.store_reserve(store_cap_ * 2)) {
                store_failed_ = true;          // out of device memory: this path is not kept,
                return -1;                     // its extensions are scored in full
            }
            store_cap_ *= 2;
        }
        return next_slot_++;
    }
    bool in_flight_ = false;
    FILE *dump_ = nullptr;
    // speculation policy (collect / collect_best)
    std::vector<std::vector<uint32_t>> edge_w_;      // [uid][edge of g_.adjacency[uid]]
    bool best_first_ = false;
    float min_like_ = 0.004f;
    int max_depth_ = 48;
    struct Spec {
        float like;
        uint64_t seq;
        Node *n;
        bool operator<(const Spec &o) const { return like != o.like ? like < o.like : seq > o.seq; }
    };
    std::vector<Spec> spec_heap_;
};

}  // namespace gfal
#endif
