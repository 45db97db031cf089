/*
 * gfalign_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see the header).
 *
 * Literal CPU restatement of the reference scoring path.  Every function
 * names the reference lines it follows (paths relative to the reference root).
 * Deliberately unoptimised: the full (n+1)x(m+1) table and the full traceback
 * are kept so the quirks of the original survive --
 *   - row 0 is initialised up to column n, not m       (src/alignments.cpp:500)
 *   - column 0 is never written and stays zero          (src/eval.cpp:79)
 *   - an "up" move is free in the last column           (src/alignments.cpp:504)
 *   - the traceback charges an "up" only after some B step was consumed and
 *     walks row 0 / column 0 for free                   (src/alignments.cpp:517-546)
 */
#include "gfalign_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

struct gfo_dp {
    int cell[GFO_MAX_N][GFO_MAX_N];
};

gfo_dp *gfo_dp_new(void) { return (gfo_dp *)calloc(1, sizeof(gfo_dp)); }
void gfo_dp_zero(gfo_dp *dp) { memset(dp, 0, sizeof(*dp)); }
void gfo_dp_free(gfo_dp *dp) { free(dp); }

/* include/alignments.h:15-17 */
static int same_step(gfo_step x, gfo_step y)
{
    return x.id == y.id && x.orient == y.orient;
}

static int max3(int x, int y, int z)
{
    int best = x > y ? x : y;
    return best > z ? best : z;
}

/* include/alignments.h:64-70: reverse, '+' -> '-', everything else -> '+'. */
void gfo_reverse_complement(const gfo_step *in, uint32_t m, gfo_step *out)
{
    for (uint32_t k = 0; k < m; ++k) {
        gfo_step s = in[m - 1 - k];
        s.orient = (s.orient == '+') ? '-' : '+';
        out[k] = s;
    }
}

/* src/alignments.cpp:499-509 */
static void fill_table(int match, int mismatch, int gap,
                       const gfo_step *a, uint32_t n,
                       const gfo_step *b, uint32_t m, gfo_dp *dp)
{
    for (uint32_t j = 0; j <= n; ++j)          /* bound is n on purpose (:500) */
        dp->cell[0][j] = (int)j * gap;
    for (uint32_t i = 1; i <= n; ++i) {
        for (uint32_t j = 1; j <= m; ++j) {
            int sub = same_step(a[i - 1], b[j - 1]) ? match : mismatch;
            int up_cost = (j < m) ? gap : 0;    /* free below the last column */
            dp->cell[i][j] = max3(dp->cell[i - 1][j - 1] + sub,
                                  dp->cell[i - 1][j] + up_cost,
                                  dp->cell[i][j - 1] + gap);
        }
    }
}

/* src/alignments.cpp:511-554 */
static int32_t trace_back(int match, int mismatch,
                          const gfo_step *a, uint32_t n,
                          const gfo_step *b, uint32_t m, const gfo_dp *dp,
                          gfo_step *row_a, gfo_step *row_b, uint32_t *row_len)
{
    static const gfo_step GAP = { -1, '0' };
    int32_t score = 0, b_taken = 0;
    uint32_t i = n, j = m, len = 0;

    while (i != 0 || j != 0) {
        gfo_step out_a, out_b;
        if (i == 0) {                               /* :517-520 free */
            out_a = GAP; out_b = b[j - 1]; --j;
        } else if (j == 0) {                        /* :521-524 free */
            out_a = a[i - 1]; out_b = GAP; --i;
        } else {
            int sub = same_step(a[i - 1], b[j - 1]) ? match : mismatch;
            if (dp->cell[i][j] == dp->cell[i - 1][j - 1] + sub) {   /* :527 */
                out_a = a[i - 1]; out_b = b[j - 1];
                ++b_taken; --i; --j; score += sub;
            } else if (dp->cell[i - 1][j] >= dp->cell[i][j - 1]) { /* :534 */
                out_a = a[i - 1]; out_b = GAP; --i;
                if (b_taken > 0) score -= 1;                       /* :538 */
            } else {                                               /* :541 */
                out_a = GAP; out_b = b[j - 1];
                ++b_taken; --j; score -= 1;
            }
        }
        if (row_a) { row_a[len] = out_a; row_b[len] = out_b; }
        ++len;
    }
    if (row_a) {                                    /* :551-552 reverse */
        for (uint32_t k = 0; k < len / 2; ++k) {
            gfo_step t = row_a[k]; row_a[k] = row_a[len - 1 - k]; row_a[len - 1 - k] = t;
            t = row_b[k]; row_b[k] = row_b[len - 1 - k]; row_b[len - 1 - k] = t;
        }
    }
    if (row_len) *row_len = len;
    return score;
}

/* src/alignments.cpp:556-561 */
int32_t gfo_align_paths(int match, int mismatch, int gap,
                        const gfo_step *a, uint32_t n,
                        const gfo_step *b, uint32_t m, gfo_dp *dp,
                        gfo_step *row_a, gfo_step *row_b, uint32_t *row_len)
{
    fill_table(match, mismatch, gap, a, n, b, m, dp);
    return trace_back(match, mismatch, a, n, b, m, dp, row_a, row_b, row_len);
}

static int cmp_i32(const void *x, const void *y)
{
    int32_t a = *(const int32_t *)x, b = *(const int32_t *)y;
    return (a > b) - (a < b);
}

/* src/eval.cpp:67-108 (printing left to the caller through best_score). */
void gfo_evaluate_path(const gfo_step *path, uint32_t n,
                       const int64_t *aln_off, const gfo_step *aln,
                       int64_t n_aln, int filter,
                       gfo_stats *out, int32_t *best_score, uint8_t *rc_wins)
{
    gfo_stats st = { 0, 0, 0 };
    /* :76-78 the set of node ids on the path (sorted array stands in for the
       hash set; membership is all that is observable). */
    int32_t *ids = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    for (uint32_t i = 0; i < n; ++i) ids[i] = path[i].id;
    qsort(ids, n, sizeof(int32_t), cmp_i32);

    gfo_dp *dp = gfo_dp_new();                      /* :79, zeroed once per call */
    gfo_step *rc = (gfo_step *)malloc(GFO_MAX_N * sizeof(gfo_step));

    for (int64_t k = 0; k < n_aln; ++k) {           /* :80 */
        const gfo_step *b = aln + aln_off[k];
        uint32_t m = (uint32_t)(aln_off[k + 1] - aln_off[k]);
        if (filter) {                               /* :81-91 */
            int skip = 0;
            for (uint32_t j = 0; j < m; ++j) {
                if (!bsearch(&b[j].id, ids, n, sizeof(int32_t), cmp_i32)) {
                    skip = 1;
                    ++st.unaligned;
                }
            }
            if (skip) {
                if (best_score) best_score[k] = INT32_MIN;
                if (rc_wins) rc_wins[k] = 0;
                continue;
            }
        }
        int32_t fw = gfo_align_paths(0, -1, -1, path, n, b, m, dp, NULL, NULL, NULL);   /* :92 */
        gfo_reverse_complement(b, m, rc);
        int32_t rv = gfo_align_paths(0, -1, -1, path, n, rc, m, dp, NULL, NULL, NULL);  /* :93 */
        int32_t best = fw > rv ? fw : rv;           /* :94 */
        if (best < 0) ++st.bad; else ++st.good;     /* :95-98 */
        if (best_score) best_score[k] = best;
        if (rc_wins) rc_wins[k] = !(fw > rv);       /* :101 */
    }
    free(rc);
    gfo_dp_free(dp);
    free(ids);
    *out = st;
}

/* Packed encoding of include/gfalign_scorer.h: (id << 1) | minus, with
   GFAL_STEP_OTHER (bit 30) marking an orientation that is neither. */
#define PACK_OTHER 0x40000000

static gfo_step unpack(int32_t s)
{
    gfo_step r;
    if (s & PACK_OTHER) {
        r.id = (s & ~PACK_OTHER) >> 1;
        r.orient = '0';
    } else {
        r.id = s >> 1;
        r.orient = (s & 1) ? '-' : '+';
    }
    return r;
}

static int unpack_alignments(const int32_t *aln_off, const int32_t *aln_steps,
                             int64_t n_aln, int64_t **off_out, gfo_step **steps_out)
{
    int64_t total = n_aln ? aln_off[n_aln] : 0;
    int64_t *off = (int64_t *)malloc((size_t)(n_aln + 1) * sizeof(int64_t));
    gfo_step *st = (gfo_step *)malloc((size_t)(total ? total : 1) * sizeof(gfo_step));
    if (!off || !st) { free(off); free(st); return -1; }
    off[0] = 0;
    for (int64_t k = 0; k < n_aln; ++k) {
        if (aln_off[k + 1] - aln_off[k] >= GFO_MAX_N) { free(off); free(st); return -2; }
        off[k + 1] = aln_off[k + 1];
    }
    for (int64_t t = 0; t < total; ++t) st[t] = unpack(aln_steps[t]);
    *off_out = off; *steps_out = st;
    return 0;
}

int gfo_evaluate_paths_packed(const int32_t *aln_off, const int32_t *aln_steps,
                              int64_t n_aln,
                              const int32_t *path_off, const int32_t *path_steps,
                              int32_t n_paths, int filter,
                              uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    int64_t *off; gfo_step *st;
    int rc = unpack_alignments(aln_off, aln_steps, n_aln, &off, &st);
    if (rc) return rc;
    gfo_step *p = (gfo_step *)malloc(GFO_MAX_N * sizeof(gfo_step));
    for (int32_t q = 0; q < n_paths; ++q) {
        int32_t n = path_off[q + 1] - path_off[q];
        if (n < 1 || n >= GFO_MAX_N) { free(p); free(off); free(st); return -2; }
        for (int32_t i = 0; i < n; ++i) p[i] = unpack(path_steps[path_off[q] + i]);
        gfo_stats s;
        gfo_evaluate_path(p, (uint32_t)n, off, st, n_aln, filter, &s, NULL, NULL);
        bad[q] = s.bad; good[q] = s.good;
        if (unaligned) unaligned[q] = s.unaligned;
    }
    free(p); free(off); free(st);
    return 0;
}

int gfo_pair_scores_packed(const int32_t *aln_off, const int32_t *aln_steps,
                           int64_t n_aln,
                           const int32_t *path_steps, int32_t n,
                           int32_t *fw, int32_t *rc)
{
    int64_t *off; gfo_step *st;
    int err = unpack_alignments(aln_off, aln_steps, n_aln, &off, &st);
    if (err) return err;
    if (n < 1 || n >= GFO_MAX_N) { free(off); free(st); return -2; }
    gfo_step *p = (gfo_step *)malloc(GFO_MAX_N * sizeof(gfo_step));
    gfo_step *r = (gfo_step *)malloc(GFO_MAX_N * sizeof(gfo_step));
    for (int32_t i = 0; i < n; ++i) p[i] = unpack(path_steps[i]);
    gfo_dp *dp = gfo_dp_new();
    for (int64_t k = 0; k < n_aln; ++k) {
        const gfo_step *b = st + off[k];
        uint32_t m = (uint32_t)(off[k + 1] - off[k]);
        fw[k] = gfo_align_paths(0, -1, -1, p, (uint32_t)n, b, m, dp, NULL, NULL, NULL);
        gfo_reverse_complement(b, m, r);
        rc[k] = gfo_align_paths(0, -1, -1, p, (uint32_t)n, r, m, dp, NULL, NULL, NULL);
    }
    gfo_dp_free(dp);
    free(p); free(r); free(off); free(st);
    return 0;
}
