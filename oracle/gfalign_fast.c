/*
 * gfalign_fast.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A second, independent CPU implementation of evaluatePath (reference
 * src/eval.cpp:67-108): not the reference's algorithm (that is
 * gfalign_oracle.c: a full table and a traceback per pair) but the decision
 * rule the HIP kernels use written the way one would
 * write it for a CPU -- filter through a node bitmap, subpath test through
 * per-node occurrence lists, exact start-overhang test, and the forward DP
 * with exit propagation only for the pairs that need it -- with OpenMP over
 * blocks of alignments.  Two uses:
 *   - tests: it must agree with gfalign_oracle.c bit for bit (a third
 *     implementation next to the oracle and the kernels);
 *   - bench.py: the "cpu_fast" line, i.e. what a good CPU implementation of
 *     the same rule does on the host cores, next to the reference-faithful
 *     cpu_baseline.
 * Packed steps as in include/gfalign_scorer.h: (id << 1) | minus; bit 30 on a
 * path step = orientation that equals nothing.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define STEP_OTHER 0x40000000
#define MAX_STEPS 1000

static inline int32_t node_of(int32_t s) { return (s & ~STEP_OTHER) >> 1; }

/* traceback score of A (n) against B (m) -- src/alignments.cpp:499-554 folded
 * into one forward pass: dp row + the dp value at the traceback's exit cell */
static int forward_score(const int32_t *a, int n, const int32_t *b, int m, int flip,
                         int *dp, int *ex)
{
    for (int j = 0; j <= m; ++j) {
        dp[j] = j <= n ? -j : 0; /* :500 row 0 reaches column n only */
        ex[j] = dp[j];
    }
    for (int i = 1; i <= n; ++i) {
        int diag_dp = dp[0], diag_x = ex[0];
        dp[0] = 0; /* column 0 is never written */
        ex[0] = 0;
        int left_dp = 0, left_x = 0;
        const int32_t ai = a[i - 1];
        for (int j = 1; j <= m; ++j) {
            const int32_t bj = flip ? (b[m - j] ^ 1) : b[j - 1];
            const int up_dp = dp[j], up_x = ex[j];
            const int d = diag_dp + (ai == bj ? 0 : -1);
            const int u = up_dp + (j < m ? -1 : 0); /* :504 */
            const int l = left_dp - 1;
            int v = d > u ? d : u;
            if (l > v) v = l;
            int x;
            if (v == d) x = diag_x;             /* :527 */
            else if (up_dp >= left_dp) x = up_x; /* :534 */
            else x = left_x;                    /* :541 */
            dp[j] = v;
            ex[j] = x;
            diag_dp = up_dp;
            diag_x = up_x;
            left_dp = v;
            left_x = x;
        }
    }
    return dp[m] - ex[m];
}

/* does a proper suffix of B' (B or rc(B)) equal a prefix of A? */
static int has_overhang(const int32_t *a, int n, const int32_t *b, int m, int flip)
{
    for (int len = 1; len <= m - 1 && len <= n; ++len) {
        int eq = 1;
        for (int k = 0; k < len && eq; ++k) {
            const int t = m - len + k; /* index in B' */
            const int32_t bt = flip ? (b[m - 1 - t] ^ 1) : b[t];
            eq = a[k] == bt;
        }
        if (eq) return 1;
    }
    return 0;
}

int gfo_fast_evaluate_paths_packed(const int32_t *aln_off, const int32_t *aln_steps, int64_t n_aln,
                                   const int32_t *path_off, const int32_t *path_steps,
                                   int32_t n_paths, int filter, int n_threads,
                                   uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    if (n_aln < 0 || n_paths < 0) return 1;
    int32_t max_node = 0;
    for (int64_t t = 0; t < (n_aln ? aln_off[n_aln] : 0); ++t) {
        if (aln_steps[t] < 0) return 2;
        if (node_of(aln_steps[t]) > max_node) max_node = node_of(aln_steps[t]);
    }
    for (int64_t t = 0; t < (n_paths ? path_off[n_paths] : 0); ++t) {
        if (path_steps[t] < 0) return 2;
        if (node_of(path_steps[t]) > max_node) max_node = node_of(path_steps[t]);
    }
    for (int32_t p = 0; p < n_paths; ++p)
        if (path_off[p + 1] - path_off[p] < 1 || path_off[p + 1] - path_off[p] > MAX_STEPS) return 3;
    for (int64_t k = 0; k < n_aln; ++k)
        if (aln_off[k + 1] - aln_off[k] < 0 || aln_off[k + 1] - aln_off[k] > MAX_STEPS) return 3;
    const int32_t V = max_node + 1;
    int failed = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    for (int32_t p = 0; p < n_paths; ++p) {
        bad[p] = 0;
        good[p] = 0;
        if (unaligned) unaligned[p] = 0;
    }
    /* occurrence lists of every path of a group (head[p][node] / nxt[step]): built
     * once, then the threads take blocks of alignments -- a block stays in cache
     * while it meets all the paths of the group */
    const int64_t group_cap = (int64_t)1 << 24; /* head entries per group */
    const int32_t group = (int32_t)(group_cap / (V > 0 ? V : 1) > 0 ? group_cap / (V > 0 ? V : 1) : 1);
    const int64_t total_path_steps = n_paths ? path_off[n_paths] : 0;
    int32_t *nxt = (int32_t *)malloc((size_t)(total_path_steps + 1) * sizeof(int32_t));
    int32_t *head = (int32_t *)malloc((size_t)(group < n_paths ? group : (n_paths ? n_paths : 1)) *
                                      (size_t)V * sizeof(int32_t));
    if (!nxt || !head) {
        free(nxt);
        free(head);
        return 4;
    }
    const int64_t block = 2048;
    const int64_t n_blocks = (n_aln + block - 1) / block;
    for (int32_t p0 = 0; p0 < n_paths; p0 += group) {
        const int32_t p1 = p0 + group < n_paths ? p0 + group : n_paths;
        for (int64_t t = 0; t < (int64_t)(p1 - p0) * V; ++t) head[t] = -2; /* -2: node not on the path */
        for (int32_t p = p0; p < p1; ++p) {
            const int32_t *a = path_steps + path_off[p];
            const int n = path_off[p + 1] - path_off[p];
            int32_t *hd = head + (size_t)(p - p0) * V;
            for (int i = n - 1; i >= 0; --i) { /* positions in ascending order */
                const int32_t v = node_of(a[i]);
                nxt[path_off[p] + i] = hd[v] == -2 ? -1 : hd[v];
                hd[v] = i;
            }
        }
#pragma omp parallel
        {
            int *dp = (int *)malloc((MAX_STEPS + 1) * sizeof(int));
            int *ex = (int *)malloc((MAX_STEPS + 1) * sizeof(int));
            uint32_t *cnt = (uint32_t *)calloc((size_t)3 * (size_t)(p1 - p0), sizeof(uint32_t));
            if (!dp || !ex || !cnt) {
#pragma omp atomic write
                failed = 1;
            } else {
#pragma omp for schedule(dynamic, 1)
                for (int64_t blk = 0; blk < n_blocks; ++blk) {
                    const int64_t k0 = blk * block, k1 = k0 + block < n_aln ? k0 + block : n_aln;
                    for (int32_t p = p0; p < p1; ++p) {
                        const int32_t *a = path_steps + path_off[p];
                        const int n = path_off[p + 1] - path_off[p];
                        const int32_t *hd = head + (size_t)(p - p0) * V;
                        const int32_t *nx = nxt + path_off[p];
                        uint32_t n_bad = 0, n_good = 0, n_una = 0;
                        for (int64_t k = k0; k < k1; ++k) {
                            const int32_t *b = aln_steps + aln_off[k];
                            const int m = aln_off[k + 1] - aln_off[k];
                            if (filter) { /* src/eval.cpp:81-91 */
                                int miss = 0;
                                for (int j = 0; j < m; ++j) miss += hd[node_of(b[j])] == -2;
                                n_una += (uint32_t)miss;
                                if (miss) continue;
                            }
                            if (m == 0 || m > n) { /* free traceback */
                                ++n_good;
                                continue;
                            }
                            /* (b) B, or rc(B), is a contiguous subpath of A */
                            int found = 0;
                            const int32_t h = hd[node_of(b[0])];
                            for (int pos = h < 0 ? -1 : h; pos >= 0 && !found; pos = nx[pos]) {
                                if (a[pos] == b[0] && pos + m <= n) {
                                    int eq = 1;
                                    for (int t = 1; t < m && eq; ++t) eq = a[pos + t] == b[t];
                                    found = eq;
                                }
                                if (!found && a[pos] == (b[0] ^ 1) && pos >= m - 1) {
                                    int eq = 1;
                                    for (int t = 1; t < m && eq; ++t) eq = a[pos - t] == (b[t] ^ 1);
                                    found = eq;
                                }
                            }
                            if (found) {
                                ++n_good;
                                continue;
                            }
                            /* (c) start-overhang, then the exact DP for that orientation */
                            int free_tb = 0;
                            for (int flip = 0; flip < 2 && !free_tb; ++flip)
                                if (has_overhang(a, n, b, m, flip))
                                    free_tb = forward_score(a, n, b, m, flip, dp, ex) == 0;
                            if (free_tb) ++n_good;
                            else ++n_bad;
                        }
                        cnt[3 * (p - p0)] += n_bad;
                        cnt[3 * (p - p0) + 1] += n_good;
                        cnt[3 * (p - p0) + 2] += n_una;
                    }
                }
#pragma omp critical
                for (int32_t p = p0; p < p1; ++p) {
                    bad[p] += cnt[3 * (p - p0)];
                    good[p] += cnt[3 * (p - p0) + 1];
                    if (unaligned) unaligned[p] += cnt[3 * (p - p0) + 2];
                }
            }
            free(dp);
            free(ex);
            free(cnt);
        }
    }
    free(nxt);
    free(head);
    return failed ? 4 : 0;
}
