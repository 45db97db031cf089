/*
 * gfalign_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the gfalign path-scoring hot path
 * (reference: src/eval.cpp:63-108, src/alignments.cpp:499-561,
 * include/alignments.h:11-21,64-70,246).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library, and
 * only as the checker.  The product (gfalign_amd/csrc) never links it.
 *
 * Parity status: the reference cannot be built in this image (its gfalibs
 * submodule is empty; see DESIGN.md "Oracle").  This restatement is pinned by
 *   - validateFiles/test.6.tst (search rows; all counters zero), and
 *   - the evaluatePath / evalPath outputs recorded in SURVEY.md Appendix C.1
 *     and C.2 (non-zero good/bad counters and per-alignment scores),
 *   - the worked examples of SURVEY.md Appendix A.3.
 * No test shipped with the reference pins non-zero counters, so beyond those
 * vectors counter parity is "parity unpinned".
 */
#ifndef GFALIGN_ORACLE_H
#define GFALIGN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/alignments.h:246 -- both path lengths must stay below this. */
#define GFO_MAX_N 1001

/* include/alignments.h:11-21: one oriented node visit. */
typedef struct {
    int32_t id;
    char    orient; /* '+', '-', or anything else ('0' on a fresh source) */
} gfo_step;

/* src/eval.cpp:63-65 */
typedef struct {
    uint32_t bad, good, unaligned;
} gfo_stats;

/* Scratch the reference keeps on the stack (src/eval.cpp:79). */
typedef struct gfo_dp gfo_dp;
gfo_dp *gfo_dp_new(void);
void    gfo_dp_zero(gfo_dp *dp);
void    gfo_dp_free(gfo_dp *dp);

/* include/alignments.h:64-70 */
void gfo_reverse_complement(const gfo_step *in, uint32_t m, gfo_step *out);

/*
 * src/alignments.cpp:556-561 alignPaths(match, mismatch, gap, A, B, dp):
 * fill (499-509) then traceback (511-554).  Returns the traceback score.
 * If row_a/row_b are non-NULL they receive the gapped rows (id -1 / '0' marks
 * a gap) in left-to-right order, *row_len their common length; each must hold
 * n+m entries.
 */
int32_t gfo_align_paths(int match, int mismatch, int gap,
                        const gfo_step *a, uint32_t n,
                        const gfo_step *b, uint32_t m,
                        gfo_dp *dp,
                        gfo_step *row_a, gfo_step *row_b, uint32_t *row_len);

/*
 * src/eval.cpp:67-108 evaluatePath.  Alignments are CSR: alignment k is
 * aln[aln_off[k] .. aln_off[k+1]).  best_score (nullable, n_aln entries)
 * receives max(fw, rc) per alignment, or INT32_MIN for one the filter skipped.
 * rc_wins (nullable) receives 1 where the reference would print the rc row
 * (src/eval.cpp:101: fw is shown only when strictly greater).
 */
void gfo_evaluate_path(const gfo_step *path, uint32_t n,
                       const int64_t *aln_off, const gfo_step *aln,
                       int64_t n_aln, int filter,
                       gfo_stats *out, int32_t *best_score, uint8_t *rc_wins);

/* Same, on the packed int32 encoding of include/gfalign_scorer.h. */
int gfo_evaluate_paths_packed(const int32_t *aln_off, const int32_t *aln_steps,
                              int64_t n_aln,
                              const int32_t *path_off, const int32_t *path_steps,
                              int32_t n_paths, int filter,
                              uint32_t *bad, uint32_t *good, uint32_t *unaligned);

/* Per-alignment fw / rc traceback scores of one packed path (evalPath rows). */
int gfo_pair_scores_packed(const int32_t *aln_off, const int32_t *aln_steps,
                           int64_t n_aln,
                           const int32_t *path_steps, int32_t n,
                           int32_t *fw, int32_t *rc);

#ifdef __cplusplus
}
#endif
#endif
