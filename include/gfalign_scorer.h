/*
 * gfalign_scorer.h -- C ABI of the MI355X path scorer (libgfalign_scorer.so).
 *
 * Drop-in boundary for the one seam the reference has on this path: the
 * internal C++ call
 *     PathAlignmentStats evaluatePath(const Path &path, InSequences &,
 *                                     std::vector<Path> alignmentPaths,
 *                                     bool filterAlignments, bool printAlignments)
 * (reference src/eval.cpp:67-108), called from the search loop
 * (src/eval.cpp:162, filter=true) and from evalPath (src/eval.cpp:238,
 * filter=false, print=true).  The reference has no FFI, so the entry points
 * below are the ones a binding for that call would need; INTEGRATION.md shows
 * the patch a reference maintainer would apply.
 *
 * Plain pointers and sizes only.  All functions return 0 on success or a
 * negative GFAL_E_* code; nothing throws or exits across this boundary.
 * A scorer is bound to one HIP device and is used from one host thread at a
 * time.  There is no CPU fallback: without a usable HIP device
 * gfal_scorer_create fails with GFAL_E_NO_DEVICE.
 *
 * Step encoding ("packed step", int32):
 *     (node_id << 1) | minus            node_id in [0, n_nodes), minus in {0,1}
 * reference Step{id, orientation} (include/alignments.h:11-21) with '+' -> 0,
 * '-' -> 1.  A candidate-path step whose orientation is neither '+' nor '-'
 * (the reference compares the raw char, so such a step equals no alignment
 * step but its node still counts for the filter; include/alignments.h:15-17,
 * src/eval.cpp:76-78) is passed as  GFAL_STEP_OTHER | (node_id << 1).
 * Alignment steps are always '+' / '-' (src/alignments.cpp:86).
 */
#ifndef GFALIGN_SCORER_H
#define GFALIGN_SCORER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFAL_ABI_VERSION 4

/* include/alignments.h:246 (MAX_N 1001): longest path / alignment accepted. */
#define GFAL_MAX_STEPS 1000

#define GFAL_STEP_OTHER 0x40000000

#define GFAL_OK            0
#define GFAL_E_ARG        -1  /* null pointer, negative size, bad offsets      */
#define GFAL_E_RANGE      -2  /* a path or alignment outside 1..GFAL_MAX_STEPS
                                 steps (UB in the reference), or a node id
                                 outside [0, n_nodes)                          */
#define GFAL_E_NO_DEVICE  -3  /* no HIP device / device index out of range     */
#define GFAL_E_HIP        -4  /* a HIP runtime call failed (see gfal_last_error)*/
#define GFAL_E_NOMEM      -5

typedef struct gfal_scorer gfal_scorer;

/* What src/eval.cpp:63-65 PathAlignmentStats holds, one per candidate path. */
typedef struct {
    uint32_t bad, good, unaligned;
} gfal_stats;

int         gfal_abi_version(void);
/* Hash of the sources and flags this library was compiled from (set by
   gfalign_amd/build.py; "unstamped" for a hand build): lets a caller or a test
   prove that the binary it loaded matches the source tree next to it. */
const char *gfal_build_id(void);
const char *gfal_strerror(int code);
/* Text of the last failure on this thread (HIP error string and call site). */
const char *gfal_last_error(void);
/* Number of HIP devices visible, or a negative GFAL_E_* code. */
int         gfal_device_count(void);

/*
 * Upload one shard of alignments (what src/eval.cpp:123 getPaths() yields,
 * done once per search) to `device` and lay it out for the kernels.
 *   aln_off   [n_aln + 1]  CSR offsets into aln_steps, aln_off[0] == 0
 *   aln_steps [aln_off[n_aln]] packed steps
 *   n_nodes   size of the node-id space (reference uId range)
 * Host buffers are only read during the call.  An empty shard (n_aln == 0) is
 * valid.  An alignment with zero steps is valid and scores "good" for every
 * path (src/alignments.cpp:516-524 with m == 0).
 */
int gfal_scorer_create(const int32_t *aln_off, const int32_t *aln_steps,
                       int64_t n_aln, int32_t n_nodes, int device,
                       gfal_scorer **out);

/*
 * Same, with the set of nodes candidate paths may step on ("universe": in
 * `gfalign search` the node list plus source and destination, src/eval.cpp:
 * 126-128; in evalPath the nodes of the path).  Alignment nodes outside it can
 * never be on a path, so they are folded into one id: results are identical,
 * and a GAF over a whole assembly graph needs no more device tables than the
 * tangle does.  universe == NULL means "any node" (gfal_scorer_create).  A
 * later path that steps on a node outside the universe is GFAL_E_RANGE.
 */
int gfal_scorer_create_ex(const int32_t *aln_off, const int32_t *aln_steps,
                          int64_t n_aln, int32_t n_nodes, int device,
                          const int32_t *universe, int32_t n_universe,
                          gfal_scorer **out);

/*
 * One of n_shards shards of the SAME alignment set (multi-GPU):
 * every shard is created from the complete arrays and keeps its share of the
 * groups of 64 alignments of the length- and content-sorted order the kernels
 * use -- a few alignment lengths per shard (the scan pays a fixed cost per
 * length it holds), the groups of a length spread evenly over the shards that
 * hold it -- so the shards of one set partition it, every shard derives the
 * same partition from the same input, and the work is balanced by what a
 * group costs the scan.  The counters of the shards add up to the unsharded
 * counters (integer sums: any reduction order).
 * gfal_scorer_create_ex is shard 0 of 1.
 */
int gfal_scorer_create_sharded(const int32_t *aln_off, const int32_t *aln_steps,
                               int64_t n_aln, int32_t n_nodes, int device,
                               const int32_t *universe, int32_t n_universe,
                               int32_t shard_index, int32_t n_shards, gfal_scorer **out);

/*
 * Same as gfal_scorer_create_sharded, with identical alignments collapsed: a
 * GAF of reads over a tangle repeats the same few-node paths many times, and a
 * lane of the kernels then stands for all copies with a weight.  Every result
 * (counters, pair scores, sharding) is identical to the uncollapsed scorer's;
 * only the work shrinks.  An algorithmic shortcut in the sense of SURVEY.md
 * 8(d): bench.py reports it next to, never inside, the headline figure.
 */
int gfal_scorer_create_dedup(const int32_t *aln_off, const int32_t *aln_steps,
                             int64_t n_aln, int32_t n_nodes, int device,
                             const int32_t *universe, int32_t n_universe,
                             int32_t shard_index, int32_t n_shards, gfal_scorer **out);

/*
 * Which shard of gfal_scorer_create_sharded(..., shard, n_shards) takes each
 * alignment: owner[k] in [0, n_shards).  Host code only (no HIP device needed):
 * the ranks of a multi-GPU run all derive the same partition from the same
 * input without communicating, and a test can check that on a CPU.
 */
int gfal_shard_owner(const int32_t *aln_off, const int32_t *aln_steps, int64_t n_aln,
                     int32_t n_nodes, const int32_t *universe, int32_t n_universe,
                     int32_t n_shards, int32_t *owner);

void gfal_scorer_destroy(gfal_scorer *s);

/*
 * evaluatePath for a batch of candidate paths (src/eval.cpp:67-108 once per
 * path).  Blocking; host buffers in, host counters out.
 *   path_off   [n_paths + 1], path_steps packed (GFAL_STEP_OTHER allowed)
 *   filter     src/eval.cpp:81-91 (true in search, false in evalPath)
 *   bad, good  [n_paths] required;  unaligned [n_paths] nullable
 */
int gfal_scorer_score(gfal_scorer *s,
                      const int32_t *path_off, const int32_t *path_steps,
                      int32_t n_paths, int filter,
                      uint32_t *bad, uint32_t *good, uint32_t *unaligned);

/*
 * Same work with everything resident in device memory and no host
 * synchronisation: the form bench.py and the multi-GPU path use (the counters
 * stay on the device for the RCCL all-reduce).  Enqueued on `hip_stream`
 * (a hipStream_t; NULL = the default stream).
 *   d_path_off   device int32 [n_paths + 1]
 *   d_path_steps device int32 [total_steps]   (total_steps == path_off[n_paths])
 *   max_path_len upper bound of any path length (<= GFAL_MAX_STEPS)
 *   d_counts     device uint32 [3 * n_paths]: bad[P] | good[P] | unaligned[P];
 *                overwritten, not accumulated
 * Lengths/ids are validated on the device; a violation is reported by
 * gfal_scorer_sync_status() after the stream has been synchronised and leaves
 * d_counts unspecified.
 */
int gfal_scorer_score_device(gfal_scorer *s,
                             const int32_t *d_path_off, const int32_t *d_path_steps,
                             int32_t n_paths, int64_t total_steps,
                             int32_t max_path_len, int filter,
                             uint32_t *d_counts, void *hip_stream);

/*
 * The shards of one alignment set on the GPUs of one node, in one process
 * (multi-GPU without torch / MPI: what `gfalign search --devices N` uses).
 * gfal_group_score is gfal_scorer_score for the whole set: every device scores
 * the batch against its shard, the per-path counters are summed on the devices
 * by one RCCL all-reduce over xGMI (uint32[3P]) and device 0's copy comes back.
 * RCCL is loaded on first use; if it cannot serve the group (library missing,
 * two shards on one device) the counters are added on the host instead --
 * gfal_group_uses_rccl tells which.  The scorers stay owned by the caller and
 * must outlive the group; one host thread at a time.
 */
typedef struct gfal_group gfal_group;
int  gfal_group_create(gfal_scorer *const *scorers, int n, gfal_group **out);
void gfal_group_destroy(gfal_group *g);
int  gfal_group_uses_rccl(const gfal_group *g);
int  gfal_group_score(gfal_group *g,
                      const int32_t *path_off, const int32_t *path_steps,
                      int32_t n_paths, int filter,
                      uint32_t *bad, uint32_t *good, uint32_t *unaligned);
/*
 * The same in two halves, so that the caller can work while the devices score:
 * _begin copies the batch into pinned staging (the host buffers are free again
 * on return) and enqueues everything -- copies in, kernels, all-reduce, copies
 * out; _end waits for it and hands the counters over.  One batch at a time.
 * `gfalign search` prepares its next candidate batch between the two.
 */
int  gfal_group_score_begin(gfal_group *g,
                            const int32_t *path_off, const int32_t *path_steps,
                            int32_t n_paths, int filter);
int  gfal_group_score_end(gfal_group *g, uint32_t *bad, uint32_t *good, uint32_t *unaligned);
/* 1 if the batch handed over by a _begin call has finished on every device (or none
   is in flight), 0 if not yet; never blocks.  `gfalign search` generates the next
   level of candidates until this says 1. */
int  gfal_group_score_poll(gfal_group *g);

/*
 * Search mode: a candidate scored from its parent.  `gfalign search` only ever
 * scores `parent + one step` (src/eval.cpp:146-162), and with the filter on the
 * counters of src/eval.cpp:92-98 follow from the parent's: the alignments that
 * newly pass the filter all carry the new node, the ones that newly become a
 * subpath all end at the new step, and the start-overhang pairs of the exact DP
 * all carry the path's first node (scorer.hip "Search mode", tests/incr_model.py).
 * The result is bit-identical to gfal_group_score on the full paths; only the work
 * shrinks from all alignments to two inverted lists.  An algorithmic shortcut in
 * the sense of SURVEY.md 8(d): bench.py reports it next to the headline figure.
 *
 *   gfal_group_store_reserve         room for n_slots scored paths on every shard
 *                                    (4 KB each, plus one bit per alignment of the
 *                                    longest inverted list: the exact DP's verdicts,
 *                                    which a child inherits where its tail avoids the
 *                                    alignment; grows, keeps what is there); the first
 *                                    call also builds the index on the devices
 *   gfal_group_score_store_begin     gfal_group_score_begin with filter = 1 that also
 *                                    keeps path p in slot slots[p] (-1: not kept)
 *   gfal_group_score_children_begin  child i = parent[i] + step[i]; parent[i] >= 0 is
 *                                    a store slot, parent[i] < 0 the child ~parent[i]
 *                                    of this batch (it must come earlier); the child
 *                                    is kept in slot[i] (-1: not kept).  Every parent
 *                                    must be at least as long as the longest
 *                                    alignment (gfal_info.max_aln_len) -- shorter
 *                                    paths go through score_store -- and no child
 *                                    longer than max_path_len.
 * Both are collected with gfal_group_score_end.  Slots are the caller's to manage:
 * a slot may be reused once no later batch names it as a parent.
 */
int  gfal_group_store_reserve(gfal_group *g, int64_t n_slots);
int  gfal_group_score_store_begin(gfal_group *g,
                                  const int32_t *path_off, const int32_t *path_steps,
                                  int32_t n_paths, const int32_t *slots);
int  gfal_group_score_children_begin(gfal_group *g, int32_t n,
                                     const int32_t *parent, const int32_t *step,
                                     const int32_t *slot, int32_t max_path_len);

/* Status word of the most recent score_device call (blocks on its stream). */
int gfal_scorer_sync_status(gfal_scorer *s);

/*
 * Traceback scores of ONE path against every alignment of the shard, both
 * orientations (src/eval.cpp:92-93): what evalPath prints per alignment
 * (src/eval.cpp:100-102).  fw, rc: host int32 [n_aln given to create], in the
 * order the alignments were given; a shard of gfal_scorer_create_sharded
 * writes the entries of its own alignments only, so calling every shard with
 * the same two arrays fills them.
 */
int gfal_scorer_pair_scores(gfal_scorer *s, const int32_t *path_steps, int32_t n,
                            int32_t *fw, int32_t *rc);

/* ---- introspection used by bench.py / tests; no effect on results ---- */

typedef struct {
    int64_t  n_aln;          /* alignments in the shard                        */
    int64_t  n_steps;        /* their total step count S                       */
    int32_t  n_nodes;        /* node-id space given at create                  */
    int32_t  n_local_nodes;  /* distinct nodes that occur in the shard         */
    int32_t  max_aln_len;
    int32_t  tile_paths;     /* candidate paths staged per workgroup (last run)*/
    int32_t  n_workgroups;   /* grid of the scan kernel (last run)             */
    int32_t  lds_bytes;      /* dynamic LDS per workgroup (last run)           */
    int64_t  dp_pairs;       /* pairs sent to the exact-DP kernel (last run,
                                valid after gfal_scorer_sync_status)           */
    float    scan_ms;        /* scan kernel, HIP events: mean over the calls   */
    float    dp_ms;          /* exact-DP kernel             made since         */
    float    total_ms;       /* whole score_device call     profiling went on  */
    int32_t  profiled_calls; /* calls in that mean (ring of 128)               */
    int64_t  n_lanes;        /* alignments resident on the device: n_aln minus
                                the zero-step ones, or the distinct ones of a
                                gfal_scorer_create_dedup scorer                */
    int64_t  n_score_calls;  /* gfal_scorer_score calls since create           */
    int64_t  n_device_passes;/* score passes enqueued since create (one per
                                score_device call; a blocking call that ran out
                                of worklist adds one per re-run piece)         */
    int64_t  n_overflow_reruns; /* blocking calls whose batch had to be run
                                again because the exact-DP worklist was too
                                short (the list grows to fit: once per scorer
                                and batch shape)                               */
    int64_t  wl_capacity;    /* exact-DP worklist entries (grows on overflow)  */
    float    scan_kernel_ms; /* of scan_ms: the dominant kernel alone (k_scan3:
                                the alignment walk itself, without the per-tile
                                window preparation); 0 when it was not timed
                                (k_scan / k_scan2 only, or a batch in slabs)   */
    int32_t  reserved_;
} gfal_info;

/* Record HIP events (on the caller's stream) around the kernels of each score
   call; switching it on resets the statistics.  Off by default.  enable == 1:
   every phase (scan_ms, scan_kernel_ms, dp_ms, total_ms; the events cost a
   10 000-path step about 3 %); enable == 2: the dominant scan kernel only
   (scan_kernel_ms; the other times stay 0). */
int gfal_scorer_set_profiling(gfal_scorer *s, int enable);
/* Blocks on the recorded events when profiling is on. */
int gfal_scorer_get_info(gfal_scorer *s, gfal_info *out);

#ifdef __cplusplus
}
#endif
#endif
