// scorer.hip -- MI355X (gfx950) implementation of include/gfalign_scorer.h.
//
// What it computes: reference src/eval.cpp:67-108 (evaluatePath) for a batch
// of candidate paths against one shard of GAF alignments; the per-pair
// decision follows src/alignments.cpp:499-554 exactly (DESIGN.md "Exact
// decision rule").  How it computes it is unrelated to the reference:
//
//   create      alignments -> dense local node ids, bucketed by length,
//               sorted by content, stored as 64-alignment "items" in
//               step-major order (one coalesced 128-B load per step per wave);
//               k_ct_build numbers the distinct contents (content table) and
//               every lane learns the number of its own.
//   k_prep      one wave per candidate path: builds the path's lookup image
//               (first-occurrence table by node, occurrence chain, steps) in
//               LDS and writes it to HBM; also the `unaligned` counter.
//   k_tile_masks, k_overhang, k_tile   (round 3; batches of >= 320 paths) per tile of
//               31 paths: node masks, and every window of the tile's paths looked
//               up in the content table ONCE -> per alignment length a list of
//               {content number, paths that contain it}.
//   k_scan3     the dominant kernel: workgroup = (tile) x (one alignment length) x
//               (a chunk of its items).  The subpath pairs are counted from the
//               tile's list x the contents' multiplicities; a lane ANDs its nodes'
//               masks (the filter for 31 paths at once) and counts who passes;
//               only lanes that can have a start overhang ask the LDS table
//               (keyed by content number) and take the exact overhang test.
//   k_scan2     round 2's scan (windows of 8 paths hashed into an LDS table,
//               steps compared per item): batches of 96..319 paths.
//   k_scan      the same decisions by occurrence-chain walks: workgroup = T path
//               images staged in LDS x one chunk of items; the rare alignment
//               lengths and batches of a few dozen paths.  Pairs the cheap rules
//               of these kernels cannot decide go to a worklist.
//   k_wl_*      counting sort of the worklist by (length class, path).
//   k_dp_regs   exact Needleman-Wunsch + traceback-exit propagation for the
//               worklist ("start-overhang" pairs): one pair per lane, rows in
//               registers (alignments of up to 32 steps), only the table rows
//               that can change the state; k_dp_long for longer alignments.
//   k_dp_sys    the same for short worklists (search-sized batches), one table
//               column per lane: the latency of a single fill decides there.
//   k_pairs     the same exact DP for every alignment of one path, both
//               orientations (evalPath's per-alignment scores).
//   k_child     search mode: a candidate scored from its parent (stored on the
//               device) through two inverted lists, a content table and the
//               DP verdicts remembered per stored path; k_dp_small decides what
//               is left in one launch.
//
// No MFMA anywhere: the work is integer compares and LDS table lookups.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "gfalign_scorer.h"

namespace {

constexpr int WAVE = 64;
constexpr int SCAN_THREADS = 768;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;
#ifndef GFAL_SCAN2_THREADS
#define GFAL_SCAN2_THREADS 1024
#endif
constexpr int SCAN2_THREADS = GFAL_SCAN2_THREADS;      // k_scan2: two workgroups per CU = 8 waves per SIMD (64 VGPRs);
                                                       // 768 threads (6 waves, 80 VGPRs) measured 4 % slower
constexpr int SCAN2_WAVES = SCAN2_THREADS / WAVE;
constexpr int SCAN2_WAVES_PER_SIMD = 2 * SCAN2_THREADS / 256;
constexpr int MAX_REG_K = 16;          // alignments of up to 2K+1 = 33 steps are compared
                                       // from registers (K pair dwords per lane)
constexpr int MAX_REG_M = 16;          // smallest step-array capacity of an image
constexpr int MAX_TILE = 32;           // one bit per tile path in the node masks
constexpr int LDS_BUDGET = 80 * 1024;  // k_scan: two workgroups per CU (160 KiB LDS);
                                       // one 160 KiB workgroup (twice the tile, 4 waves
                                       // per SIMD) measured 19 % slower
constexpr int LDS_MAX = 160 * 1024;
constexpr int PREP_THREADS = 256;     // k_prep: four waves build one path image
constexpr int N_CLASSES = 5, LONG_CLASS = 4;   // length classes of the DP kernels, see length_class()
constexpr int DP_THREADS = 64;
constexpr int DP_BLOCKS = 512;        // row-scratch kernels (k_dp_long, k_pairs)
constexpr int DP_REG_BLOCKS = 4096, DP_SYS_BLOCKS = 2048;   // register-row kernels: 4 waves per SIMD

// Chain entries (first[] / next[]): position | ENT_NEG.  The two terminal
// values decode to positions 1023 / 1022, which no window test accepts.
constexpr uint32_t ENT_NONE = 0xFFFFu;     // no (further) occurrence
constexpr uint32_t ENT_PRESENT = 0xFFFEu;  // node is on the path, but only as
                                           // steps that equal nothing
constexpr uint32_t ENT_FOUND = 0xFFFDu;    // scan kernel: this lane's search ended in a
                                           // match (next[1021] points back at it)
constexpr uint32_t ENT_POS = 0x03FFu;      // position in the path (0..999)
constexpr uint32_t ENT_NEG = 0x8000u;      // step there is '-'
constexpr int NEXT_CAP = 1024;             // next[] is indexable by any ENT_POS value
constexpr int NEVER_LEN = 40;              // uint16 entries of the never-matching window
constexpr uint32_t STEP_NOMATCH = 0xFFFEu; // path step that equals nothing
constexpr uint32_t STEP_INVALID = 0xFFFFu; // padding lane of an item
constexpr int MAX_LOCAL_NODES = 32766;     // 2*V-1 must stay below STEP_NOMATCH

constexpr uint32_t ST_BAD_LEN = 1u, ST_BAD_ID = 2u, ST_DP_OVERFLOW = 4u;

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Nothing may throw across the C ABI (std::vector / std::thread on the host
// side of create and pair_scores can): every entry point that allocates runs
// inside this.
template <typename F>
int no_throw(F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_err("out of host memory");
        return GFAL_E_NOMEM;
    } catch (const std::exception &e) {
        set_err("host error: %s", e.what());
        return GFAL_E_NOMEM;
    } catch (...) {
        set_err("host error");
        return GFAL_E_NOMEM;
    }
}

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                    __FILE__, __LINE__);                                       \
            return GFAL_E_HIP;                                                 \
        }                                                                      \
    } while (0)

// Geometry of one path image, in uint16 units.  Shared by host and device.
// Every section starts on a 4-byte boundary: the scan kernel reads the image
// with 32-bit LDS loads only (a ds_read_u16 costs 12-16 LDS cycles per
// wave-instruction on gfx950, a ds_read_b32 2.4; tools/lds_rate.hip).
//   first[v2]    node -> chain entry of its first occurrence on the path
//   next[1024]   uint32 per position: chain entry of the node's next
//                occurrence; indexable by any 10-bit position, the tail beyond
//                the path holds ENT_NONE
//   step[nm]     the path's steps (local packed codes), 0xFFFF beyond n
//   rstep[nm]    the steps of the path's reverse complement
//   never[32]    0xFFFF: a window that equals no alignment steps
//   len, pad
struct ImageLayout {
    int v2;        // first-occurrence table entries (n_local rounded up to even)
    int nm;        // capacity of the step arrays (even, >= n + 2, >= 2 * MAX_REG_M)
    int total;     // whole image, multiple of 8 (16 bytes)
    __host__ __device__ int first_at() const { return 0; }
    __host__ __device__ int next_at() const { return v2; }
    __host__ __device__ int step_at() const { return v2 + 2 * NEXT_CAP; }
    __host__ __device__ int rstep_at() const { return v2 + 2 * NEXT_CAP + nm; }
    __host__ __device__ int never_at() const { return v2 + 2 * NEXT_CAP + 2 * nm; }
    __host__ __device__ int len_at() const { return v2 + 2 * NEXT_CAP + 2 * nm + NEVER_LEN; }
};

ImageLayout make_layout(int n_local, int max_len)
{
    ImageLayout L;
    L.v2 = (n_local + 1) & ~1;
    L.nm = std::max((max_len + 5) & ~1, 2 * MAX_REG_M);   // windows read up to n + 3
    L.total = (L.len_at() + 2 + 7) & ~7;
    return L;
}

// Items: 64 alignments of equal length m, step-major.  Step t of lane l of
// item k sits at steps[item_base[k] * 64 + t * 64 + l].
// pairs[item_pbase[k] * 64 + j * 64 + l] = step 2j+1 | step 2j+2 << 16 of lane l:
// the steps after the first, two to a dword, as the scan kernel compares them
// (m / 2 dwords per lane; the upper half of the last one is 0 when m is even).
struct Items {
    const uint16_t *steps;
    const uint32_t *base;   // per item, in units of 64 uint16
    const uint16_t *len;    // per item, m
    const uint32_t *pairs;
    const uint32_t *pbase;  // per item, in units of 64 uint32
    int n_items;
    // per item: two local node ids (16 bits each; possibly the same) that occur in
    // EVERY alignment of the item (content-sorted items nearly always have some:
    // their first nodes) -- or, with COMMON_EITHER, two nodes one of which every
    // alignment has -- or NO_COMMON_NODE.  If no path of a tile carries them, every lane fails the
    // filter for every tile path and the item is skipped for that tile without
    // loading a step -- a third of the (item, tile) visits of the config-3 batch
    const uint32_t *common;
    // {base, pbase, common, len} of every item in one 16-byte load (k_scan)
    const uint4 *hdr;
    // gfal_scorer_create_dedup only (else NULL): how many identical alignments of
    // the caller a lane stands for, [item * 64 + lane]
    const uint32_t *weight;
};
constexpr uint32_t NO_COMMON_NODE = 0xFFFFFFFFu;
constexpr uint32_t COMMON_EITHER = 0x8000u;   // Items::common: every lane has the one node OR the other

// A children batch (search mode, see k_child): the path of candidate p is the stored
// path `root[p]` plus the steps of `depth[p]` candidates of the batch, p last; k_prep
// reads it from there and keeps it in the candidate's own store slot.
constexpr int STORE_STRIDE = GFAL_MAX_STEPS;     // int32 steps per store slot
struct PrepChild {
    const int32_t *root = nullptr, *depth = nullptr, *parent = nullptr, *step = nullptr, *slot = nullptr;
    int32_t *st_steps = nullptr, *st_len = nullptr;
};

// --------------------------------------------------------------------------
// k_prep: candidate path -> lookup image (+ counter initialisation)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(PREP_THREADS) void k_prep(
    const int32_t *__restrict__ path_off, const int32_t *__restrict__ path_steps,
    int n_paths, int64_t total_steps, int max_len,
    const int32_t *__restrict__ node_local, int n_nodes,
    const uint32_t *__restrict__ node_hist, uint32_t hist_total,
    uint32_t n_empty, int filter, ImageLayout L,
    const int32_t *__restrict__ order, uint16_t *__restrict__ images,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ status,
    uint32_t *__restrict__ wl_hist, uint16_t *__restrict__ lids_out, PrepChild pc)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t *img = smem;
    uint16_t *lids = smem + L.total;   // nm entries: local node id per step
    uint32_t *head32 = reinterpret_cast<uint32_t *>(smem + L.total + L.nm);   // v2 chain heads
    // slot q of the image pool / of `counts` holds the q-th longest path
    const int q = blockIdx.x;
    const int lane = threadIdx.x;      // thread of the workgroup (four waves per path)
    if (q >= n_paths) return;
    const int p = order ? order[q] : q;
    // this slot's bins of the worklist histogram (k_scan counts into them)
    if (wl_hist && lane < N_CLASSES) wl_hist[lane * n_paths + q] = 0;

    int64_t off = path_off[p];
    int64_t end = path_off[p + 1];
    int n = (int)(end - off);
    bool len_ok = off >= 0 && end <= total_steps && end >= off && n >= 1 &&
                  n <= max_len && n <= GFAL_MAX_STEPS;
    if (!len_ok) {
        if (lane == 0) atomicOr(status, ST_BAD_LEN);
        n = 0;
    }

    uint16_t *first = img + L.first_at();
    uint32_t *next = reinterpret_cast<uint32_t *>(img + L.next_at());
    uint16_t *step = img + L.step_at();
    uint16_t *rstep = img + L.rstep_at();
    {   // 0xFFFF everywhere (32-bit stores), then the 32-bit chain entries
        uint32_t *img32 = reinterpret_cast<uint32_t *>(img);
        for (int i = lane; i < L.total / 2; i += PREP_THREADS) img32[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (int i = lane; i < NEXT_CAP; i += PREP_THREADS)
            next[i] = i == (int)(ENT_FOUND & ENT_POS) ? ENT_FOUND : ENT_NONE;
        __syncthreads();
    }

    bool id_ok = true;
    // (children batch: from the stored ancestor, the last `depth` steps from the batch)
    const int c_root = pc.root ? pc.root[p] : 0;
    const int c_l0 = pc.root ? n - pc.depth[p] : n;
    const int c_slot = (pc.root && c_root >= 0) ? pc.slot[p] : -1;
    for (int i = lane; i < n; i += PREP_THREADS) {
        int32_t s;
        if (!pc.root) {
            s = path_steps[off + i];
        } else if (c_root < 0) {
            s = 0;                                  // (rejected by k_child_len: a one-step dummy)
        } else if (i < c_l0) {
            s = pc.st_steps[(size_t)c_root * STORE_STRIDE + i];
        } else {
            int j = p;
            for (int k = n - 1 - i; k > 0; --k) j = ~pc.parent[j];
            s = pc.step[j];
        }
        if (c_slot >= 0) pc.st_steps[(size_t)c_slot * STORE_STRIDE + i] = s;
        bool other = (s & GFAL_STEP_OTHER) != 0;
        int32_t id = (s & ~GFAL_STEP_OTHER) >> 1;
        uint32_t neg = (uint32_t)s & 1u;
        int32_t lid = -1;
        if (s < 0 || id >= n_nodes) id_ok = false;
        else lid = node_local[id];
        if (lid == -2) {          // node outside the scorer's universe
            id_ok = false;
            lid = -1;
        }
        step[i] = (uint16_t)((lid < 0 || other) ? STEP_NOMATCH
                                                 : (((uint32_t)lid << 1) | neg));
        lids[i] = (uint16_t)(lid < 0 ? ENT_NONE : (uint32_t)lid);
    }
    if (!id_ok) atomicOr(status, ST_BAD_ID);
    __syncthreads();
    // reverse complement (include/alignments.h:64-70), so that "rc(B) is a
    // subpath of the path" is searched as "B is a subpath of rc(path)"
    for (int i = lane; i < n; i += PREP_THREADS) {
        uint32_t code = step[n - 1 - i];
        rstep[i] = (uint16_t)(code == STEP_NOMATCH ? STEP_NOMATCH : (code ^ 1u));
    }

    // Occurrence chains, built by all lanes at once: every matchable position
    // swaps itself into its node's chain head (LDS atomic exchange) and keeps
    // the previous head as its successor.  The order of a chain is whatever the
    // hardware made it; the search ORs over all occurrences, so any order gives
    // the same answer.  Steps that equal nothing (STEP_NOMATCH) stay out of the
    // chains; a node that only has such steps gets ENT_PRESENT (it still passes
    // the filter).  Exactly one position per node sees an empty head: it stands
    // for the node in the `unaligned` sum below.
    for (int i = lane; i < L.v2; i += PREP_THREADS) head32[i] = ENT_NONE;
    __syncthreads();
    uint32_t covered = 0;
    for (int base = 0; base < n; base += PREP_THREADS) {
        const int i = base + lane;
        const uint32_t lid = i < n ? (uint32_t)lids[i] : ENT_NONE;
        const uint32_t code = i < n ? (uint32_t)step[i] : STEP_NOMATCH;
        if (lid != ENT_NONE && code != STEP_NOMATCH) {
            const uint32_t ent = (uint32_t)i | ((code & 1u) ? ENT_NEG : 0u);
            const uint32_t old = atomicExch(&head32[lid], ent);
            next[i] = old;
            // unaligned (src/eval.cpp:83-88) = steps of all alignments whose
            // node is not on the path = total - sum over the distinct path
            // nodes of how many alignment steps carry them
            if (filter && old == ENT_NONE) covered += node_hist[lid];
        }
    }
    __syncthreads();
    for (int base = 0; base < n; base += PREP_THREADS) {
        const int i = base + lane;
        const uint32_t lid = i < n ? (uint32_t)lids[i] : ENT_NONE;
        if (lid != ENT_NONE && step[i] == STEP_NOMATCH) {
            const uint32_t old = atomicCAS(&head32[lid], ENT_NONE, ENT_PRESENT);
            if (filter && old == ENT_NONE) covered += node_hist[lid];
        }
    }
    __syncthreads();
    for (int i = lane; i < L.v2; i += PREP_THREADS) first[i] = (uint16_t)head32[i];
    if (lane == 0) {
        img[L.len_at()] = (uint16_t)n;
        img[L.len_at() + 1] = 0;
        if (c_slot >= 0) pc.st_len[c_slot] = n;
    }
    __syncthreads();

    uint16_t *dst = images + (size_t)q * L.total;
    for (int i = lane * 2; i < L.total; i += PREP_THREADS * 2)
        *reinterpret_cast<uint32_t *>(dst + i) =
            *reinterpret_cast<const uint32_t *>(img + i);
    // k_scan2 builds its node masks from the node ids along the path
    if (lids_out)
        for (int i = lane; i < L.nm; i += PREP_THREADS)
            lids_out[(size_t)q * L.nm + i] = i < n ? lids[i] : (uint16_t)0xFFFFu;

    for (int o = 32; o > 0; o >>= 1) covered += __shfl_down(covered, o, WAVE);
    // across the waves: the chain heads are dead by now, their first word adds up
    if (lane == 0) head32[0] = 0;
    __syncthreads();
    if ((lane & (WAVE - 1)) == 0 && covered) atomicAdd(&head32[0], covered);
    __syncthreads();
    covered = head32[0];
    if (lane == 0) {
        counts[q] = 0;
        counts[n_paths + q] = n_empty;
        counts[2 * n_paths + q] = filter ? hist_total - covered : 0u;
    }
}

// --------------------------------------------------------------------------
// Longest paths first.  The scan kernel's workgroups are dispatched in index
// order and a tile's cost grows with the length of its paths, so tiles are
// formed from the paths in order of decreasing length (longest-processing-time
// first: the expensive tiles never end up in the tail) and paths of similar
// length share a tile (an item is in range for all of them or for none).
// A counting sort on the device; the order among equal lengths is whatever the
// atomics made it and is not observable (counters are written back by slot).
// --------------------------------------------------------------------------
constexpr int LEN_BINS = 1024;

__device__ __forceinline__ int length_bin(const int32_t *path_off, int p)
{
    const long long n = (long long)path_off[p + 1] - path_off[p];
    return GFAL_MAX_STEPS - (int)max(0ll, min(n, (long long)GFAL_MAX_STEPS));
}

__global__ void k_len_hist(const int32_t *__restrict__ path_off, int n_paths,
                           uint32_t *__restrict__ bins)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_paths) atomicAdd(&bins[length_bin(path_off, p)], 1u);
}

__global__ __launch_bounds__(LEN_BINS) void k_len_offsets(uint32_t *__restrict__ bins)
{
    // bins[0..1023] counts -> bins[1024..2047] exclusive offsets, bins[2048..] cursors
    __shared__ uint32_t part[LEN_BINS];
    const int tid = threadIdx.x;
    const uint32_t mine = bins[tid];
    part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < LEN_BINS; o <<= 1) {
        uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    bins[LEN_BINS + tid] = part[tid] - mine;
    bins[2 * LEN_BINS + tid] = 0;
}

__global__ void k_len_scatter(const int32_t *__restrict__ path_off, int n_paths,
                              uint32_t *__restrict__ bins, int32_t *__restrict__ order)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    const int b = length_bin(path_off, p);
    order[bins[LEN_BINS + b] + atomicAdd(&bins[2 * LEN_BINS + b], 1u)] = p;
}

// The three steps above in one workgroup (bins in LDS): batches of up to a few
// ten thousand paths, where two extra launches cost more than the sort.
__global__ __launch_bounds__(LEN_BINS) void k_len_sort_block(
    const int32_t *__restrict__ path_off, int n_paths, int32_t *__restrict__ order,
    uint32_t *__restrict__ zero_a, int n_a, uint32_t *__restrict__ zero_b, int n_b)
{
    __shared__ uint32_t count[LEN_BINS], part[LEN_BINS], cursor[LEN_BINS];
    const int tid = threadIdx.x;
    // also clears the per-call accumulators (instead of two memsets)
    for (int i = tid; i < n_a; i += LEN_BINS) zero_a[i] = 0;
    for (int i = tid; i < n_b; i += LEN_BINS) zero_b[i] = 0;
    count[tid] = 0;
    cursor[tid] = 0;
    __syncthreads();
    for (int p = tid; p < n_paths; p += LEN_BINS) atomicAdd(&count[length_bin(path_off, p)], 1u);
    __syncthreads();
    const uint32_t mine = count[tid];
    part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < LEN_BINS; o <<= 1) {
        uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    part[tid] -= mine;
    __syncthreads();
    for (int p = tid; p < n_paths; p += LEN_BINS) {
        const int b = length_bin(path_off, p);
        order[part[b] + atomicAdd(&cursor[b], 1u)] = p;
    }
}

// counters by slot -> counters by caller's path index
__global__ void k_unpermute(const uint32_t *__restrict__ by_slot,
                            const int32_t *__restrict__ order, int n_paths,
                            uint32_t *__restrict__ out, const uint32_t *__restrict__ status,
                            uint32_t *__restrict__ status_copy)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    // blocking API: the status words travel with the counters (one copy out)
    if (status_copy && q < 4) status_copy[q] = status[q];
    if (q >= n_paths) return;
    const int p = order[q];
    out[p] = by_slot[q];
    out[n_paths + p] = by_slot[n_paths + q];
    out[2 * n_paths + p] = by_slot[2 * n_paths + q];
}

// --------------------------------------------------------------------------
// k_scan
// --------------------------------------------------------------------------
struct ScanArgs {
    Items items;
    const uint16_t *images;   // n_paths images
    ImageLayout L;
    int n_paths;
    int tile;                 // paths per workgroup (<= MAX_TILE)
    int n_tiles;
    int n_chunks;
    int item_lo;              // first item this kernel scans (the ones before it: k_scan2)
    int filter;
    uint32_t *counts;         // bad[P] | good[P] | unaligned[P]
    unsigned long long *worklist;
    unsigned long long *wl_count;   // 64-bit: attempted appends (may exceed the capacity)
    uint32_t wl_capacity;
    uint32_t *wl_hist;        // [N_CLASSES * n_paths] entries per (length class, path)
    uint32_t *status;
};

// Worklist entry: bit 63 = rc orientation has an overhang, bit 62 = fw has one,
// bits 32..61 = path, bits 0..31 = slot (item * 64 + lane).
constexpr unsigned long long WL_FW = 1ull << 62, WL_RC = 1ull << 63;
constexpr uint32_t WL_PATH_MASK = 0x3FFFFFFFu;

// Length classes of the DP kernels (rows held in 4 / 8 / 16 / 32 registers, or
// in LDS / HBM for longer alignments).
__host__ __device__ __forceinline__ int length_class(int m)
{
    return m <= 4 ? 0 : m <= 8 ? 1 : m <= 16 ? 2 : m <= 32 ? 3 : LONG_CLASS;
}

// What a wave keeps about the tile while it walks its items.
struct TileView {
    const uint16_t *lds;       // T images
    const uint32_t *nodemask;  // [v2] bit p: node occurs on tile path p
    int tile_paths;
    int path0;
    uint32_t all_paths;        // low tile_paths bits set
    int hdr_n;                 // lane p: length of tile path p
    uint32_t hdr_a0;           // lane p: first step of tile path p
};

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m, int lane)
{
    return __popcll(m & ((1ull << lane) - 1ull));
}

// Append the lanes in `want` to the worklist (fw / rc: which orientation has
// the overhang) and count them in the (path, length class) histogram that the
// counting sort in front of k_dp uses.
// Wave votes as 64-bit lane masks (SGPR pairs): most of the per-(wave, path)
// bookkeeping below is scalar mask arithmetic, not per-lane code.
using lanemask = unsigned long long;
#define WAVE_MASK(pred) (__builtin_amdgcn_ballot_w64(pred))
#define WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0ull)

__device__ __forceinline__ void push_pairs_to(unsigned long long *worklist,
                                              unsigned long long *wl_count, uint32_t wl_capacity,
                                              uint32_t *wl_hist, uint32_t *status, int n_paths,
                                              bool fw, bool rc, int lane, uint32_t path_global,
                                              uint32_t slot, int M)
{
    const bool want = fw || rc;
    unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    if (m == 0) return;
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 5      // timing probe: what do the appends cost?
    return;
#endif
    unsigned long long base = 0;
    int leader = __ffsll((long long)m) - 1;
    if (lane == leader) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        base = atomicAdd(wl_count, (unsigned long long)cnt);
        atomicAdd(&wl_hist[(uint32_t)length_class(M) * (uint32_t)n_paths + path_global], cnt);
    }
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader);
    if (want) {
        const unsigned long long idx = base + lanes_below(m, lane);
        if (idx < wl_capacity)
            worklist[idx] = (fw ? WL_FW : 0ull) | (rc ? WL_RC : 0ull) |
                            ((unsigned long long)path_global << 32) | slot;
        else
            atomicOr(status, ST_DP_OVERFLOW);
    }
}

// The same for all tile paths of one item at once: bit p of fw / rc = this lane's
// alignment is a candidate on tile path p.  One returning atomic per item (the
// list cursor), fire-and-forget adds for the per-path histogram; the entries of
// one path stay together (k_wl_scatter moves runs of equal (class, path) with one
// atomic per run).
__device__ __forceinline__ void push_item_pairs(unsigned long long *worklist,
                                                unsigned long long *wl_count, uint32_t wl_capacity,
                                                uint32_t *wl_hist, uint32_t *status, int n_paths,
                                                uint32_t fw, uint32_t rc, int lane, uint32_t path0,
                                                int tile_paths, uint32_t slot, int M)
{
    const uint32_t any = fw | rc;
    if (!WAVE_ANY(any != 0u)) return;
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 8      // timing probe: what do the appends cost?
    return;
#endif
    const uint32_t cls = (uint32_t)length_class(M);
    uint32_t total = 0;
    for (int p = 0; p < tile_paths; ++p) {
        const uint32_t c = (uint32_t)__popcll(WAVE_MASK(((any >> p) & 1u) != 0u));
        if (c != 0u && lane == 0) atomicAdd(&wl_hist[cls * (uint32_t)n_paths + path0 + (uint32_t)p], c);
        total += c;
    }
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(wl_count, (unsigned long long)total);
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), 0) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, 0);
    bool overflow = false;
    for (int p = 0; p < tile_paths; ++p) {
        const bool mine = ((any >> p) & 1u) != 0u;
        const lanemask m = WAVE_MASK(mine);
        if (m == 0) continue;
        if (mine) {
            const unsigned long long idx = base + lanes_below(m, lane);
            if (idx < wl_capacity)
                worklist[idx] = (((fw >> p) & 1u) ? WL_FW : 0ull) | (((rc >> p) & 1u) ? WL_RC : 0ull) |
                                ((unsigned long long)(path0 + (uint32_t)p) << 32) | slot;
            else
                overflow = true;
        }
        base += (unsigned long long)__popcll(m);
    }
    if (overflow) atomicOr(status, ST_DP_OVERFLOW);
}

__device__ __forceinline__ void push_pairs(const ScanArgs &a, bool fw, bool rc, int lane,
                                           uint32_t path_global, uint32_t slot, int M)
{
    push_pairs_to(a.worklist, a.wl_count, a.wl_capacity, a.wl_hist, a.status, a.n_paths, fw, rc, lane,
                  path_global, slot, M);
}

// uint16 entry i of a 4-byte aligned LDS array, fetched with a 32-bit load.
__device__ __forceinline__ uint32_t lds_u16(const uint32_t *base32, uint32_t i)
{
    return (base32[i >> 1] >> ((i & 1u) << 4)) & 0xFFFFu;
}

// live lanes: does B[start + dirn*k] ^ flipbit equal path step k for k in
// [1, len)?  (k = 0 was checked by the caller.)  len, start, dirn are
// wave-uniform; B is re-read from the item (coalesced, cache-hot).
__device__ __forceinline__ bool tail_equals(const uint16_t *__restrict__ bp, int start,
                                            int dirn, int len, uint32_t flipbit,
                                            const uint32_t *step32, bool live)
{
    bool eq = live;
    for (int k = 1; k < len; ++k)
        eq &= ((uint32_t)bp[(start + dirn * k) * WAVE] ^ flipbit) ==
              lds_u16(step32, (uint32_t)k);
    return eq;
}


// Per-(wave, tile path) counters: lane p of `packed` holds good | bad << 16 for
// tile path p; flushed into the 32-bit lanes before 16 bits can overflow.
// Sum of w over the lanes of `mask` (all lanes get the result).  Dedup scorers
// only: a lane then stands for w identical alignments.
__device__ __forceinline__ uint32_t wave_weight(lanemask mask, uint32_t w, int lane)
{
    if (mask == 0) return 0u;
    uint32_t v = ((mask >> lane) & 1ull) ? w : 0u;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v += (uint32_t)__shfl_xor((int)v, o, WAVE);
    return v;
}

struct WaveCounts {
    uint32_t packed = 0, good = 0, bad = 0;
    int items = 0;
    __device__ __forceinline__ void add(int p, int lane, lanemask good_m, lanemask bad_m)
    {
        const uint32_t inc = (uint32_t)__popcll(good_m) | ((uint32_t)__popcll(bad_m) << 16);
        if (lane == p) packed += inc;
    }
    // dedup scorers: weights instead of lane counts, straight into the 32-bit sums.
    // all_m / w_all: the item's lanes and their total weight -- the usual masks
    // (all lanes found, or none) need no reduction
    __device__ __forceinline__ void add_weighted(int p, int lane, lanemask good_m, lanemask bad_m,
                                                 uint32_t w, lanemask all_m, uint32_t w_all)
    {
        const uint32_t gw = good_m == all_m ? w_all : wave_weight(good_m, w, lane);
        const uint32_t bw = bad_m == all_m ? w_all : wave_weight(bad_m, w, lane);
        if (lane == p) {
            good += gw;
            bad += bw;
        }
    }
    __device__ __forceinline__ void flush()
    {
        good += packed & 0xFFFFu;
        bad += packed >> 16;
        packed = 0;
        items = 0;
    }
    __device__ __forceinline__ void item_done()
    {
        if (++items == 1000) flush();   // 64 per item and field
    }
};

// Occurrence-chain search (DESIGN.md "k_scan"): is B a contiguous subpath of
// the path, or of its reverse complement?  `e` = chain head for B[0]'s node
// (ENT_NONE on lanes that take no part).  Both directions are forward scans:
// step32 views, as dwords, the path's steps, then (nm entries further) the
// steps of its reverse complement, then the never-matching window.  Requires
// M <= n.  Branch-free body; the only branch is the wave-level loop test.
// Returns the lanes whose alignment was found.
struct ChainView {
    const uint32_t *next;     // next[NEXT_CAP]
    const uint32_t *step32;   // step[nm] | rstep[nm] | never[NEVER_LEN]
    uint32_t nm;
    int n;
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 4
    uint32_t *dbg;
#endif
};

// pairs[k] = b[1 + 2k] | b[2 + 2k] << 16: steps 1..M-1 two to a dword.
// K = M / 2 is a template constant; M itself (2K or 2K + 1) is the template
// constant MC when that is non-zero (short alignments: everything folds at
// compile time) and the wave-uniform run-time value otherwise.
template <int K, int MC>
__device__ __forceinline__ lanemask subpath_search(const uint32_t (&pairs)[K ? K : 1], int M_rt,
                                                   uint32_t o0, uint32_t e,
                                                   const ChainView &cv)
{
    const int M = MC ? MC : M_rt;
    // dwords of path steps that cover the K pairs at either alignment
    constexpr int KD = MC ? (MC + 1) / 2 : K + 1;
    const uint32_t last_start = (uint32_t)(cv.n - M);
    const uint32_t n1 = (uint32_t)(cv.n - 1);
    // M even: the last pair holds one step only
    const uint32_t last_mask = (M & 1) ? 0xFFFFFFFFu : 0x0000FFFFu;
    while (true) {
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 4
        {
            const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
            const bool uni = __builtin_amdgcn_ballot_w64(e == e1 && o0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)o0)) == __builtin_amdgcn_ballot_w64(true);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(cv.dbg, 1u);
                if (uni) atomicAdd(cv.dbg + 1, 1u);
            }
        }
#endif
        const uint32_t pos = e & ENT_POS;
        // same orientation as b0: B may start at pos of the path; opposite:
        // B may start at n-1-pos of its reverse complement
        const bool rc = (e & ENT_NEG) != o0;
        const uint32_t start = rc ? n1 - pos : pos;
        // terminal entries decode to pos >= 1021: start > n - M either way
        const bool fits = start <= last_start;
        // uint16 index of the first step to compare (step 1 of the window)
        uint32_t idx = start + (rc ? cv.nm + 1u : 1u);
        idx = fits ? idx : 2u * cv.nm;
        const uint32_t nx = cv.next[pos];
        bool ok = fits;
        if (K > 0) {
            const uint32_t *wd = cv.step32 + ((idx & ~1u) >> 1);
            const uint32_t sh = idx << 4;          // alignbit uses bits 4:0: 0 or 16
            uint32_t d[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) d[k] = wd[k];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t got =
                    __builtin_amdgcn_alignbit(d[(k + 1 < KD) ? k + 1 : k], d[k], sh);
                if (k + 1 < K)
                    ok &= got == pairs[k];
                else if (MC && !(MC & 1))   // compile-time even M: one step in the last pair
                    ok &= (uint16_t)got == (uint16_t)pairs[k];
                else if (MC)
                    ok &= got == pairs[k];
                else
                    ok &= ((got ^ pairs[k]) & last_mask) == 0u;
            }
        }
        // a matched lane parks on ENT_FOUND (next[1021] == ENT_FOUND keeps it
        // there); exhausted chains park on ENT_NONE (next[1022..1023])
        e = ok ? ENT_FOUND : nx;
        if (!WAVE_ANY(e < ENT_FOUND)) break;
    }
    return WAVE_MASK(e == ENT_FOUND);
}

__device__ __forceinline__ lanemask subpath_search_long(const uint16_t *__restrict__ bp,
                                                        int M, uint32_t b0, uint32_t e,
                                                        const ChainView &cv)
{
    const uint32_t o0 = (b0 & 1u) << 15;
    const uint32_t last_start = (uint32_t)(cv.n - M);
    const uint32_t n1 = (uint32_t)(cv.n - 1);
    while (true) {
        const uint32_t pos = e & ENT_POS;
        const bool rc = (e & ENT_NEG) != o0;
        const uint32_t start = rc ? n1 - pos : pos;
        const bool fits = start <= last_start;
        const uint32_t idx = fits ? start + (rc ? cv.nm : 0u) : 2u * cv.nm;
        const uint32_t nx = cv.next[pos];
        bool ok = fits;
        // wave-uniform trip count; a lane that does not fit re-reads the
        // never-matching window and stays false
        const int lim = WAVE_ANY(fits) ? M : 1;
        for (int t = 1; t < lim; ++t) {
            const uint32_t at = fits ? idx + (uint32_t)t : idx;
            ok &= lds_u16(cv.step32, at) == (uint32_t)bp[t * WAVE];
        }
        e = ok ? ENT_FOUND : nx;
        if (!WAVE_ANY(e < ENT_FOUND)) break;
    }
    return WAVE_MASK(e == ENT_FOUND);
}

// One item against the tile.  K = M / 2 pair dwords per lane live in
// registers; M (wave-uniform) is 2K or 2K + 1, a compile-time constant MC for
// short alignments (MC = 0: run-time).
// W: dedup scorer (lane weights `w`); a separate instantiation, so that the
// plain kernel carries nothing for it.
template <int K, int MC, bool W>
__device__ __forceinline__ void scan_item(const ScanArgs &a, const TileView &tv,
                                          const uint16_t *__restrict__ bp,
                                          const uint32_t *__restrict__ pp, int M_rt, int lane,
                                          uint32_t slot, WaveCounts &wc, uint32_t w)
{
    // dedup: the item's lanes and their total weight, once per item
    const lanemask all_m = W ? WAVE_MASK(bp[0] != STEP_INVALID) : 0ull;
    const uint32_t w_all = W ? wave_weight(all_m, w, lane) : 0u;
    auto count = [&](int p, lanemask good_m, lanemask bad_m) {
        if (W) wc.add_weighted(p, lane, good_m, bad_m, w, all_m, w_all);
        else wc.add(p, lane, good_m, bad_m);
    };
    const int M = MC ? MC : M_rt;
    uint32_t b0 = bp[0];
    uint32_t pairs[K ? K : 1];
    pairs[0] = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) pairs[k] = pp[k * WAVE];
    const bool valid = b0 != STEP_INVALID;
    if (!valid) {           // keep table indices in range on padding lanes
        b0 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) pairs[k] = 0;
    }
    // bit p of pass: every node of this alignment occurs on tile path p, i.e.
    // the alignment survives the filter of src/eval.cpp:81-91 for that path
    uint32_t pass = valid ? tv.all_paths : 0u;
    if (a.filter) {
        pass &= tv.nodemask[b0 >> 1];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            pass &= tv.nodemask[(pairs[k] & 0xFFFFu) >> 1];
            if (k + 1 < K || (M & 1)) pass &= tv.nodemask[pairs[k] >> 17];
        }
    }
    // the head lookup reads first[] (uint16 entries) through a dword
    const uint32_t node0_word = b0 >> 2, node0_shift = (b0 & 2u) << 3;
    const uint32_t o0 = (b0 & 1u) << 15;
    // Which lanes hold the tile's first step anywhere in B (either strand)?  Only
    // those can have a start-overhang.  The candidate paths of a batch share their
    // first step (the search's source), so this is computed once per item for the
    // first step of tile path 0 and is loop-invariant: nothing per-lane is carried
    // across the path loop (carried masks cost five register copies per path
    // iteration).  A tile path with another first step takes the exact test always.
    const uint32_t tile_a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, 0);
    lanemask a0_lanes;
    {
        bool has = ((b0 ^ tile_a0) & ~1u) == 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            has |= (((pairs[k] & 0xFFFFu) ^ tile_a0) & ~1u) == 0u;
            if (k + 1 < K || (M & 1)) has |= (((pairs[k] >> 16) ^ tile_a0) & ~1u) == 0u;
        }
        a0_lanes = WAVE_MASK(has);
    }

    for (int p = 0; p < tv.tile_paths; ++p) {
        const lanemask in_m = WAVE_MASK(((pass >> p) & 1u) != 0u);
        if (in_m == 0) continue;
        const int n = __builtin_amdgcn_readlane(tv.hdr_n, p);
        if (M > n) {                      // src/alignments.cpp:500 row-0 bound:
            count(p, in_m, 0);            // longer than the path -> good
            continue;
        }
        const uint16_t *img = tv.lds + p * a.L.total;
        const uint32_t *img32 = reinterpret_cast<const uint32_t *>(img);
        const uint32_t *step32 = img32 + a.L.step_at() / 2;
        ChainView cv;
        cv.next = img32 + a.L.next_at() / 2;
        cv.step32 = step32;
        cv.nm = (uint32_t)a.L.nm;
        cv.n = n;
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 4
        cv.dbg = a.status + 4;
#endif
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 2
        count(p, in_m, 0);
        continue;
#endif
        // every lane searches (a lane that is not `in` misses a node of the
        // path and cannot match; padding lanes are masked off right after)
        const uint32_t head = (img32[node0_word] >> node0_shift) & 0xFFFFu;
        const lanemask found_m = subpath_search<K, MC>(pairs, M, o0, head, cv) & in_m;
        const lanemask open_m = in_m & ~found_m;
        lanemask bad_m = 0;
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 6      // timing probe: no overhang triage
        bad_m = open_m;
        if (false) {
#else
        if (open_m != 0) {
#endif
            // B is not a subpath and m <= n: the traceback stays free only if a
            // proper suffix of B (or of rc(B)) equals a prefix of the path
            // ("start-overhang").  Exact test; survivors go to the DP kernels.
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, p);
            const lanemask maybe_m = a0 == tile_a0 ? (open_m & a0_lanes) : open_m;
            lanemask cand_m = 0;
            if (maybe_m != 0) {
                // bit t-1 of a0_fw: b[t] == a0 (t >= 1); bit t of a0_rc: b[t]^1 == a0 (t <= M-2)
                uint32_t a0_fw = 0;
                uint32_t a0_rc = (M >= 2 && (b0 ^ 1u) == a0) ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t lo = pairs[k] & 0xFFFFu, hi = pairs[k] >> 16;
                    const int t_lo = 2 * k + 1, t_hi = 2 * k + 2;
                    a0_fw |= (lo == a0) ? (1u << (t_lo - 1)) : 0u;
                    a0_rc |= (t_lo <= M - 2 && (lo ^ 1u) == a0) ? (1u << t_lo) : 0u;
                    if (t_hi < M) {
                        a0_fw |= (hi == a0) ? (1u << (t_hi - 1)) : 0u;
                        a0_rc |= (t_hi <= M - 2 && (hi ^ 1u) == a0) ? (1u << t_hi) : 0u;
                    }
                }
                const bool open = (open_m >> lane) & 1ull;
                bool cand_fw = false, cand_rc = false;
                for (int t = 1; t < M; ++t) {      // B[t..M) == path[0..M-t) ?
                    const bool live = open && ((a0_fw >> (t - 1)) & 1u);
                    if (WAVE_ANY(live))
                        cand_fw |= tail_equals(bp, t, 1, M - t, 0u, step32, live);
                }
                for (int t = 0; t < M - 1; ++t) {  // rc(B)[M-1-t..M) == path[0..t+1) ?
                    const bool live = open && ((a0_rc >> t) & 1u);
                    if (WAVE_ANY(live))
                        cand_rc |= tail_equals(bp, t, -1, t + 1, 1u, step32, live);
                }
                cand_m = WAVE_MASK(cand_fw || cand_rc);
                push_pairs(a, cand_fw, cand_rc, lane, (uint32_t)(tv.path0 + p), slot, M);
            }
            bad_m = open_m & ~cand_m;
        }
        count(p, found_m, bad_m);
    }
    wc.item_done();
}

// Same decision for alignments too long for registers.
template <bool W>
__device__ __forceinline__ void scan_item_long(const ScanArgs &a, const TileView &tv,
                                               const uint16_t *__restrict__ bp, int M,
                                               int lane, uint32_t slot, WaveCounts &wc, uint32_t w)
{
    // dedup: the item's lanes and their total weight, once per item
    const lanemask all_m = W ? WAVE_MASK(bp[0] != STEP_INVALID) : 0ull;
    const uint32_t w_all = W ? wave_weight(all_m, w, lane) : 0u;
    auto count = [&](int p, lanemask good_m, lanemask bad_m) {
        if (W) wc.add_weighted(p, lane, good_m, bad_m, w, all_m, w_all);
        else wc.add(p, lane, good_m, bad_m);
    };
    const uint32_t first_step = bp[0];
    const bool valid = first_step != STEP_INVALID;
    const uint32_t b0 = valid ? first_step : 0u;
    uint32_t pass = valid ? tv.all_paths : 0u;
    if (a.filter) {
        for (int t = 0; t < M; ++t) {
            const uint32_t bt = valid ? (uint32_t)bp[t * WAVE] : 0u;
            pass &= tv.nodemask[bt >> 1];
        }
    }
    for (int p = 0; p < tv.tile_paths; ++p) {
        const lanemask in_m = WAVE_MASK(((pass >> p) & 1u) != 0u);
        if (in_m == 0) continue;
        const int n = __builtin_amdgcn_readlane(tv.hdr_n, p);
        if (M > n) {
            count(p, in_m, 0);
            continue;
        }
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, p);
        const uint16_t *img = tv.lds + p * a.L.total;
        const uint32_t *img32 = reinterpret_cast<const uint32_t *>(img);
        const uint32_t *step32 = img32 + a.L.step_at() / 2;
        ChainView cv;
        cv.next = img32 + a.L.next_at() / 2;
        cv.step32 = step32;
        cv.nm = (uint32_t)a.L.nm;
        cv.n = n;
        const uint32_t head = lds_u16(img32, b0 >> 1);
        const lanemask found_m = subpath_search_long(bp, M, b0, head, cv) & in_m;
        const lanemask open_m = in_m & ~found_m;
        lanemask bad_m = 0;
        if (open_m != 0) {
            const bool open = (open_m >> lane) & 1ull;
            bool cand_fw = false, cand_rc = false;
            for (int t = 0; t < M; ++t) {
                const uint32_t bt = bp[t * WAVE];
                const bool live_fw = open && t >= 1 && bt == a0;
                if (WAVE_ANY(live_fw))
                    cand_fw |= tail_equals(bp, t, 1, M - t, 0u, step32, live_fw);
                const bool live_rc = open && t < M - 1 && (bt ^ 1u) == a0;
                if (WAVE_ANY(live_rc))
                    cand_rc |= tail_equals(bp, t, -1, t + 1, 1u, step32, live_rc);
            }
            bad_m = open_m & ~WAVE_MASK(cand_fw || cand_rc);
            push_pairs(a, cand_fw, cand_rc, lane, (uint32_t)(tv.path0 + p), slot, M);
        }
        count(p, found_m, bad_m);
    }
    wc.item_done();
}

template <bool W>
__global__ __launch_bounds__(SCAN_THREADS, 6) void k_scan(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = tid >> 6;

    // (tile, chunk) <- workgroup id, tile-major.  A chunk is a strided subset of
    // the items (chunk, chunk + n_chunks, ...): items are ordered by alignment
    // length, so contiguous ranges would give the workgroups of the last chunks
    // all the long, expensive alignments.  (Pinning chunks to XCDs for L2
    // locality was measured slower for the same reason: the XCD with the long
    // alignments finishes last.)
    const int bid = blockIdx.x;
    const int tile_id = bid % a.n_tiles;
    const int chunk = bid / a.n_tiles;

    TileView tv;
    tv.path0 = tile_id * a.tile;
    tv.tile_paths = min(a.tile, a.n_paths - tv.path0);
    tv.all_paths = tv.tile_paths >= 32 ? 0xFFFFFFFFu : ((1u << tv.tile_paths) - 1u);
    tv.lds = lds;
    uint32_t *nodemask = reinterpret_cast<uint32_t *>(lds + (size_t)a.tile * a.L.total);
    tv.nodemask = nodemask;

    // stage the tile's images: contiguous in HBM, 16 bytes per lane
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(
            a.images + (size_t)tv.path0 * a.L.total);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        const int n16 = tv.tile_paths * a.L.total / 8;
        for (int i = tid; i < n16; i += SCAN_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    // which tile paths carry each node (the filter of src/eval.cpp:81-91 as a
    // bit test): read off the first-occurrence tables just staged
    for (int v2 = tid; v2 < a.L.v2 / 2; v2 += SCAN_THREADS) {    // two nodes per dword
        uint32_t m_lo = 0, m_hi = 0;
        for (int p = 0; p < tv.tile_paths; ++p) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(
                lds + (size_t)p * a.L.total + a.L.first_at())[v2];
            m_lo |= ((w & 0xFFFFu) != ENT_NONE) ? (1u << p) : 0u;
            m_hi |= ((w >> 16) != ENT_NONE) ? (1u << p) : 0u;
        }
        nodemask[2 * v2] = m_lo;
        nodemask[2 * v2 + 1] = m_hi;
    }
    tv.hdr_n = 0;
    tv.hdr_a0 = STEP_NOMATCH;
    if (lane < tv.tile_paths) {
        tv.hdr_n = lds[lane * a.L.total + a.L.len_at()];
        tv.hdr_a0 = lds[lane * a.L.total + a.L.step_at()];
    }
    __syncthreads();

    WaveCounts wc;                        // lane p: path0 + p
    // The wave takes its items 64 at a time: every lane loads the header of one
    // of them and looks its common node up in the tile's node mask, one ballot
    // says which of the 64 can have a lane that passes the filter, and only those
    // are scanned (a rejected visit costs 1/64 of a load and of an LDS read).
    // (Headers through scalar loads were measured 7 % slower: s_load shares the
    // lgkm counter with the LDS reads.)
    const int item_stride = SCAN_WAVES * a.n_chunks;
    for (int it0 = a.item_lo + chunk + wave * a.n_chunks; it0 < a.items.n_items; it0 += WAVE * item_stride) {
        const int my_it = it0 + lane * item_stride;
        const bool mine = my_it < a.items.n_items;
        const uint4 hdr = mine ? a.items.hdr[my_it] : make_uint4(0u, 0u, NO_COMMON_NODE, 0u);
        bool keep = mine;
        if (a.filter && mine && hdr.z != NO_COMMON_NODE) {
            const uint32_t m1 = tv.nodemask[hdr.z & 0x7FFFu], m2 = tv.nodemask[hdr.z >> 16];
            // every lane has both nodes: a tile path must carry both; every lane has
            // one of the two (COMMON_EITHER): some tile path must carry one of them
            keep = ((hdr.z & COMMON_EITHER) ? (m1 | m2) : (m1 & m2)) != 0u;
        }
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 7      // probe: (item, tile) visits and how many the common node rejects
        {
            const lanemask seen = WAVE_MASK(mine), dropped = WAVE_MASK(mine && !keep);
            if (lane == 0) {
                atomicAdd(a.status + 5, (uint32_t)__builtin_popcountll(seen));
                atomicAdd(a.status + 4, (uint32_t)__builtin_popcountll(dropped));
            }
        }
#endif
        for (lanemask todo = WAVE_MASK(keep); todo != 0; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const int it = it0 + src * item_stride;
        const uint32_t base_v = (uint32_t)__builtin_amdgcn_readlane((int)hdr.x, src);
        const uint32_t pbase_v = (uint32_t)__builtin_amdgcn_readlane((int)hdr.y, src);
        const int M = __builtin_amdgcn_readlane((int)hdr.w, src);
        const uint16_t *bp = a.items.steps + (size_t)base_v * WAVE + lane;
        const uint32_t slot = (uint32_t)it * WAVE + lane;
        const uint32_t *pp = a.items.pairs + (size_t)pbase_v * WAVE + lane;
        const uint32_t w = W ? a.items.weight[slot] : 1u;
        if (M > 2 * MAX_REG_K + 1) {
            scan_item_long<W>(a, tv, bp, M, lane, slot, wc, w);
            continue;
        }
        switch (M) {
#define GFAL_CASE(MM)                                                          \
    case MM:                                                                   \
        scan_item<MM / 2, MM, W>(a, tv, bp, pp, M, lane, slot, wc, w);         \
        break;
            GFAL_CASE(1) GFAL_CASE(2) GFAL_CASE(3) GFAL_CASE(4)
            GFAL_CASE(5) GFAL_CASE(6) GFAL_CASE(7) GFAL_CASE(8)
            GFAL_CASE(9) GFAL_CASE(10) GFAL_CASE(11) GFAL_CASE(12)
            GFAL_CASE(13) GFAL_CASE(14) GFAL_CASE(15) GFAL_CASE(16)
#undef GFAL_CASE
#define GFAL_CASE(KK)                                                          \
    case 2 * KK:                                                               \
    case 2 * KK + 1:                                                           \
        scan_item<KK, 0, W>(a, tv, bp, pp, M, lane, slot, wc, w);              \
        break;
            case 17: scan_item<8, 0, W>(a, tv, bp, pp, M, lane, slot, wc, w); break;
            GFAL_CASE(9) GFAL_CASE(10) GFAL_CASE(11) GFAL_CASE(12)
            GFAL_CASE(13) GFAL_CASE(14) GFAL_CASE(15) GFAL_CASE(16)
#undef GFAL_CASE
        }
        }   // items of this group of 64
    }

    // workgroup reduction through LDS (images are dead now), then one atomic
    // per counter per workgroup
    wc.flush();
    const uint32_t cnt_good = wc.good, cnt_bad = wc.bad;
    __syncthreads();
    uint32_t *red = reinterpret_cast<uint32_t *>(lds);
    if (tid < 2 * MAX_TILE) red[tid] = 0;
    __syncthreads();
    if (lane < tv.tile_paths) {
        if (cnt_bad) atomicAdd(&red[lane], cnt_bad);
        if (cnt_good) atomicAdd(&red[MAX_TILE + lane], cnt_good);
    }
    __syncthreads();
    if (tid < tv.tile_paths) {
        uint32_t d = red[tid], g = red[MAX_TILE + tid];
        if (d) atomicAdd(&a.counts[tv.path0 + tid], d);
        if (g) atomicAdd(&a.counts[a.n_paths + tv.path0 + tid], g);
    }
}

// --------------------------------------------------------------------------
// k_scan2: the same decisions as k_scan, with the subpath test answered for ALL
// paths of the tile at once.
//
// k_scan walks, per (item, tile path), the occurrence chain of the alignment's
// first node and compares windows: ~31 VALU + 23 SALU per (item, path), and it
// is bound by instruction issue.  Here a workgroup is
// (tile of T <= 8 paths) x (ONE alignment length M) x (chunk of the items of
// that length).  Its prologue enters every M-step window of the tile's paths,
// both strands, into an open-addressing table in LDS:
//     slot -> { fingerprint | representative (path, strand, position) },
//     mask[slot] = the tile paths that contain exactly this window
// (identical windows of different paths share one entry: the insert compares
// the steps themselves, so an entry stands for one window CONTENT).  An
// alignment of M steps is a subpath of tile path p (either strand)  <=>  its
// own step sequence is in the table with bit p set.  The alignment's hash is
// path independent and precomputed by `create`; per item a lane does one
// probe sequence, one exact comparison against the representative window,
// and has the answer for all T paths.  The filter (src/eval.cpp:81-91) stays
// the AND of per-node path masks; `m > n` is a constant mask per workgroup;
// the start-overhang triage runs, as in k_scan, only for lanes whose
// alignment contains the tile's first step.  Counters are kept per lane as
// packed bytes (one byte per tile path) and reduced once per 255 items.
// About 9 VALU per (item, path) at T = 8 instead of 31.
//
// The table holds H_CAP distinct windows.  Paths are entered one after the
// other; similar paths -- prefixes of one walk, siblings of a search -- share
// their windows and all fit.  If a path overflows the table, the table is
// rebuilt without it and the workgroup makes a further pass over its items
// for the remaining paths of the tile (exact for any input: one path alone
// always fits).
// --------------------------------------------------------------------------
constexpr int H_LOG_S = 13;
constexpr int H_SLOTS = 1 << H_LOG_S;
constexpr uint32_t H_EMPTY = 0xFFFFFFFFu;
constexpr int H_CAP = H_SLOTS / 2;        // distinct windows per table: load <= 1/2 (one path: < 2000)
constexpr int TILE2_MAX = 8;              // one mask byte per slot
constexpr int MAX_SEGS = 64;              // length segments per launch (one lane each)
constexpr uint32_t H_FP_SHIFT = 14;       // entry: fp[31:14] | index of the window's first step in the staged steps [13:0]
constexpr uint32_t NOT_A0 = 0x80000000u;  // node mask bit 31: node is NOT the tile's first node

// (the length goes in at the end: the hashes of all the windows that start at one
// position are prefixes of one fold -- k_tile)
__host__ __device__ __forceinline__ uint32_t whash_init() { return 0x811C9DC5u; }
__host__ __device__ __forceinline__ uint32_t whash_step(uint32_t h, uint32_t code)
{
    return (h ^ code) * 0x01000193u;
}
__host__ __device__ __forceinline__ uint32_t whash_final(uint32_t h, int M)
{
    h ^= (uint32_t)M * 0x9E3779B1u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

struct LenSeg {            // the items of one alignment length M
    uint32_t item_lo, item_hi;
    uint32_t m;
    uint32_t n_chunks;     // (filled in per workgroup)
    // item `it` of the segment: steps at items.steps[(step_base + (it - item_lo) * M) * 64],
    // step pairs at pairs0[(p0_base + (it - item_lo) * ceil(M / 2)) * 64]
    uint32_t step_base, p0_base;
};

struct Scan2Args {
    Items items;
    const uint32_t *item_hash;   // [n_items * 64] whash of every lane's alignment
    // pairs0[.. * 64 + lane] = step 2j | step 2j+1 << 16 of the lane's alignment (ALL
    // steps, two to a dword from step 0; 0xFFFF beyond the last): what a lane loads
    // of its alignment, compared dword-wise against a window of the path
    const uint32_t *pairs0;
    const uint16_t *images;      // k_prep's path images (steps, reverse steps, length)
    ImageLayout L;
    const uint16_t *lids;        // [n_paths][L.nm] local node id per path position (0xFFFF none)
    int n_paths, tile, n_tiles, filter;
    int debug;                   // GFAL_DEBUG_SCAN2: 1 = prologue only (timing probe, wrong counters)
    const LenSeg *segs;          // this launch's segments (<= MAX_SEGS), static per scorer
    int n_segs;
    // chunks of a segment of n items = clamp(min(n * chunk_mult, n * chunk_inv_min) >> 24, 1, 65535):
    // the same integer arithmetic on the host (grid size) and in every workgroup
    unsigned long long chunk_mult, chunk_inv_min;
    uint32_t *counts;
    unsigned long long *worklist;
    unsigned long long *wl_count;
    uint32_t wl_capacity;
    uint32_t *wl_hist;
    uint32_t *status;
};

__host__ __device__ __forceinline__ uint32_t seg_chunks(uint32_t n_items, unsigned long long mult,
                                                        unsigned long long inv_min)
{
    const unsigned long long a = ((unsigned long long)n_items * mult) >> 24;
    const unsigned long long b = ((unsigned long long)n_items * inv_min) >> 24;
    const unsigned long long c = a < b ? a : b;
    return (uint32_t)(c < 1ull ? 1ull : c > 65535ull ? 65535ull : c);
}

struct Tile2 {
    const uint16_t *steps;       // [T][2][nm] forward | reverse-complement steps
    const uint32_t *nodemask;    // [v2] bits 0..7: node on tile path p; bit 31: NOT_A0
    const uint32_t *table;       // [H_SLOTS]
    const uint8_t *maskb;        // [H_SLOTS]
    int nm;
    int tile_paths, path0;
    int hdr_n;                   // lane p: length of tile path p
    uint32_t hdr_a0;             // lane p: first step of tile path p
    bool uniform_a0;             // all tile paths start with the same step
    uint32_t sub_mask;           // tile paths of the current pass
    uint32_t gt_mask;            // of those: shorter than M (every passing alignment is good)
};

// GFAL_STAMPS (diagnostic build only, scripts/stamp_probe.sh): where a wave's
// cycles go inside scan2_item -- s_memtime stamps between the phases, summed per
// wave and added to a debug buffer nothing else reads (status words 8..).
#ifdef GFAL_STAMPS
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
__device__ unsigned long long g_stamp_sum[8];
__device__ unsigned long long g_stamp_wg[8];      // per wave: prologue, item loop, end-barrier wait, waves
struct StampAcc {
    unsigned long long v[8] = {};
};
#define STAMP_ACC StampAcc stamps;
#else
#define STAMP_ACC
#endif

// Node masks are 32 bits per node (bits 0..7: the tile paths, bit 31: NOT_A0), or
// -- NM8, tangles of many nodes: 4 bytes per node would not leave room for the
// paths -- one byte per node (bits 0..6: at most 7 tile paths, bit 7: NOT_A0; a
// ds_read_u8 costs 6-8x a ds_read_b32 on gfx950, so only then).
template <bool NM8>
__device__ __forceinline__ uint32_t node_mask(const Tile2 &tv, uint32_t node)
{
    if constexpr (NM8) return reinterpret_cast<const uint8_t *>(tv.nodemask)[node];
    else return tv.nodemask[node];
}
constexpr uint32_t NOT_A0_8 = 0x80u;
constexpr int TILE2_MAX_NM8 = 7;

// lane p: totals of tile path p; all lanes: byte p of packed_* counts this
// lane's alignments for tile path p since the last flush (< 256 items)
struct WaveCounts2 {
    STAMP_ACC
    uint32_t good = 0, bad = 0;
    uint32_t g_lo = 0, g_hi = 0, b_lo = 0, b_hi = 0;
    int items = 0;
    __device__ __forceinline__ static uint32_t spread4(uint32_t m4)
    {
        // bit k of m4 -> bit 8k (k < 4): the partial products land on distinct bits
        return (m4 * 0x00204081u) & 0x01010101u;
    }
    __device__ __forceinline__ void add(uint32_t good_mask, uint32_t bad_mask, int tile_paths)
    {
        g_lo += spread4(good_mask & 0xFu);
        b_lo += spread4(bad_mask & 0xFu);
        if (tile_paths > 4) {
            g_hi += spread4((good_mask >> 4) & 0xFu);
            b_hi += spread4((bad_mask >> 4) & 0xFu);
        }
        if (++items == 255) flush();
    }
    __device__ __forceinline__ static uint32_t wave_sum(uint32_t v)
    {
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) v += (uint32_t)__shfl_xor((int)v, o, WAVE);
        return v;
    }
    __device__ __forceinline__ void flush()
    {
        const int lane = threadIdx.x & (WAVE - 1);
        // two byte columns per reduction (sums stay below 64 * 255 < 2^16)
        const uint32_t g02 = wave_sum(g_lo & 0x00FF00FFu), g13 = wave_sum((g_lo >> 8) & 0x00FF00FFu);
        const uint32_t g46 = wave_sum(g_hi & 0x00FF00FFu), g57 = wave_sum((g_hi >> 8) & 0x00FF00FFu);
        const uint32_t b02 = wave_sum(b_lo & 0x00FF00FFu), b13 = wave_sum((b_lo >> 8) & 0x00FF00FFu);
        const uint32_t b46 = wave_sum(b_hi & 0x00FF00FFu), b57 = wave_sum((b_hi >> 8) & 0x00FF00FFu);
        const uint32_t gsel = lane < 4 ? ((lane & 1) ? g13 : g02) : ((lane & 1) ? g57 : g46);
        const uint32_t bsel = lane < 4 ? ((lane & 1) ? b13 : b02) : ((lane & 1) ? b57 : b46);
        if (lane < TILE2_MAX) {
            good += (lane & 2) ? (gsel >> 16) : (gsel & 0xFFFFu);
            bad += (lane & 2) ? (bsel >> 16) : (bsel & 0xFFFFu);
        }
        g_lo = g_hi = b_lo = b_hi = 0;
        items = 0;
    }
};

// dedup scorers: a lane stands for w identical alignments
struct WaveCounts2W {
    STAMP_ACC
    uint32_t good = 0, bad = 0;
    uint32_t g[TILE2_MAX] = {}, b[TILE2_MAX] = {};
    __device__ __forceinline__ void add(uint32_t good_mask, uint32_t bad_mask, uint32_t w)
    {
#pragma unroll
        for (int p = 0; p < TILE2_MAX; ++p) {
            g[p] += ((good_mask >> p) & 1u) ? w : 0u;
            b[p] += ((bad_mask >> p) & 1u) ? w : 0u;
        }
    }
    __device__ __forceinline__ void flush()
    {
        const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
        for (int p = 0; p < TILE2_MAX; ++p) {
            const uint32_t gs = WaveCounts2::wave_sum(g[p]), bs = WaveCounts2::wave_sum(b[p]);
            if (lane == p) {
                good += gs;
                bad += bs;
            }
            g[p] = b[p] = 0;
        }
    }
};

// A wave-uniform pointer, pinned to an SGPR pair: the per-lane offset is then
// added by the load itself (global_load v, v_off, s[base]) instead of living as
// a hoisted 64-bit VGPR pair per array (k_scan spills exactly those).
// The result is typed as a GLOBAL pointer: after the asm the compiler no longer
// knows where the pointer came from, and a generic pointer would be loaded with
// flat_load (which counts on lgkmcnt too, so every LDS wait would also wait for
// the item loads issued ahead).
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const GLOBAL_AS T *sgpr_ptr(const T *p)
{
    asm volatile("" : "+s"(p));
    return (const GLOBAL_AS T *)p;
}

// step t of the window (strand, pos) of a tile path, from the staged steps
__device__ __forceinline__ uint32_t tile_step(const uint16_t *steps, int nm, uint32_t path_strand,
                                              uint32_t pos)
{
    return steps[path_strand * (uint32_t)nm + pos];
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v += (uint32_t)__shfl_xor((int)v, o, WAVE);
    return v;
}

// Steps (idx + 2k, idx + 2k + 1) of the staged steps, packed like an alignment's
// pair dword: two 32-bit LDS reads + v_alignbit (a ds_read_u16 costs 12-16 LDS
// cycles per wave-instruction on gfx950, a ds_read_b32 2).
__device__ __forceinline__ uint32_t window_pair(const uint32_t *step32, uint32_t idx, int k)
{
    const uint32_t *w = step32 + (idx >> 1) + k;
    return __builtin_amdgcn_alignbit(w[1], w[0], idx << 4);
}

// Prologue: enter the M-step windows of tile path p (both strands) into the table.
// Tile paths are usually prefixes or siblings of one another.  `lcp[t]` = how many
// leading steps path t shares with the pass's first path (`base`): a window of
// path t that lies inside that common prefix IS the base path's window over the
// same positions, so the base path enters it once with the bits of all the paths
// that share it, and the other paths enter only their windows beyond the prefix.
__device__ __forceinline__ void insert_windows(uint16_t *steps, int nm, uint32_t *table,
                                               uint32_t *maskw, uint32_t *used, int p, int n, int M,
                                               int tid, int base, int n_base, const uint32_t *lcp_all,
                                               int pass_lo, int pass_hi)
{
    const int lcp = p == base ? 0 : (int)lcp_all[p];
    const uint32_t *step32 = reinterpret_cast<const uint32_t *>(steps);
    const int n_win = n - M + 1;                 // windows per strand
    const int P0 = (M + 1) / 2;
    uint32_t fresh = 0;                          // entries this thread created
    for (int w = tid; w < 2 * n_win; w += SCAN2_THREADS) {
        const uint32_t strand = w >= n_win ? 1u : 0u;
        const uint32_t pos = (uint32_t)(strand ? w - n_win : w);
        const uint32_t idx = ((uint32_t)p * 2u + strand) * (uint32_t)nm + pos;
        const int x = strand ? n - (int)pos - M : (int)pos;          // first path position covered
        if (base >= 0 && p != base && x + M <= lcp) continue;        // the base path's window: entered there
        uint32_t bits = 1u << p;
        if (p == base)
            for (int t = pass_lo; t < pass_hi; ++t)
                bits |= (t != base && x + M <= (int)lcp_all[t]) ? (1u << t) : 0u;
        uint32_t h = whash_init();
        bool real = true;
        for (int k = 0; k < P0; ++k) {
            const uint32_t d = window_pair(step32, idx, k);
            const uint32_t lo = d & 0xFFFFu, hi = d >> 16;
            real &= lo < STEP_NOMATCH;           // a step that equals nothing: no alignment matches
            h = whash_step(h, lo);
            if (2 * k + 1 < M) {
                real &= hi < STEP_NOMATCH;
                h = whash_step(h, hi);
            }
        }
        if (!real) continue;
        h = whash_final(h, M);
        const uint32_t want = (h & ~((1u << H_FP_SHIFT) - 1u)) | idx;
        uint32_t slot = h & (H_SLOTS - 1u);
        const uint32_t stride = ((h >> H_LOG_S) & (H_SLOTS - 1u)) | 1u;
        // Entries are counted when the call ends; a table that fills up meanwhile
        // (unrelated paths entered together) shows in the probe length: at a load of
        // 1/2 a run of 48 occupied slots has probability 2^-48, so such a run means
        // overflow -- flag it and stop (the caller rebuilds the table path by path).
        int probes = 0;
        bool gave_up = used[1] != 0u;
        while (!gave_up) {
            const uint32_t old = atomicCAS(&table[slot], H_EMPTY, want);
            if (old == H_EMPTY) {
                ++fresh;
                break;
            }
            if (++probes > 48) {
                used[1] = 1u;
                gave_up = true;
                break;
            }
            if (((old ^ h) >> H_FP_SHIFT) == 0u) {      // same fingerprint: the same window?
                const uint32_t rep = old & ((1u << H_FP_SHIFT) - 1u);
                bool same = true;
                for (int k = 0; k < P0; ++k) {
                    const uint32_t mask = (2 * k + 1 < M) ? 0xFFFFFFFFu : 0xFFFFu;
                    same &= ((window_pair(step32, rep, k) ^ window_pair(step32, idx, k)) & mask) == 0u;
                }
                if (same) break;
            }
            slot = (slot + stride) & (H_SLOTS - 1u);
        }
        if (gave_up) break;
        atomicOr(&maskw[slot >> 2], bits << ((slot & 3u) * 8u));
    }
    // one add per wave (64 lanes on one LDS word serialise)
    fresh = wave_sum_u32(fresh);
    if ((tid & (WAVE - 1)) == 0 && fresh) {
        if (atomicAdd(used, fresh) + fresh > (uint32_t)H_CAP) used[1] = 1u;     // overflow: see k_scan2
    }
}

// How many leading steps tile paths p and q share (both strands staged: the
// forward one is compared), left in *out (LDS, preset to min(n_p, n_q)).
__device__ __forceinline__ void common_prefix(const uint16_t *steps, int nm, int p, int q, int n,
                                              uint32_t *out, int tid)
{
    const uint32_t *a = reinterpret_cast<const uint32_t *>(steps + (size_t)p * 2 * nm);
    const uint32_t *b = reinterpret_cast<const uint32_t *>(steps + (size_t)q * 2 * nm);
    for (int i = tid; 2 * i < n; i += SCAN2_THREADS) {
        const uint32_t x = a[i] ^ b[i];
        if (x) atomicMin(out, (uint32_t)(2 * i + ((x & 0xFFFFu) ? 0 : 1)));
    }
}

// One item against the tile (all lanes: one alignment each, M steps).
// What a lane holds of its alignment of one item (loaded one item ahead):
// P0 = ceil(M / 2) step pairs (from step 0), its hash, its weight.
template <int P0>
struct ItemRegs {
    uint32_t p0[P0 > 0 ? P0 : 1];
    uint32_t h, w;
    uint32_t it;             // (uniform) the item
};

template <int P0, int MC, bool W, bool NM8, typename Counts>
__device__ __forceinline__ void scan2_item(const Scan2Args &a, const Tile2 &tv, const LenSeg &sg,
                                           const ItemRegs<P0> &r, int M_rt, int lane, Counts &wc)
{
    const int M = MC ? MC : M_rt;
#ifdef GFAL_STAMPS
    unsigned long long st0, st1, st2, st3, st4;
    STAMP(st0);
#endif
    const uint32_t h = r.h, w = r.w;
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(tv.table);
    const uint32_t *step32 = reinterpret_cast<const uint32_t *>(tv.steps);
    // first probe and its path mask go out together with the node-mask reads: one
    // LDS round trip for all of them
    const uint32_t stride4 = (((h >> H_LOG_S) & (H_SLOTS - 1u)) | 1u) << 2;
    uint32_t slot4 = (h & (H_SLOTS - 1u)) << 2;
    uint32_t e = *reinterpret_cast<const uint32_t *>(tb + slot4);
    uint32_t mword = *reinterpret_cast<const uint32_t *>(tb + 4u * H_SLOTS + (slot4 >> 2 & ~3u));

    uint32_t p0[P0];
#pragma unroll
    for (int k = 0; k < P0; ++k) p0[k] = r.p0[k];
    const bool valid = (p0[0] & 0xFFFFu) != STEP_INVALID;
    if (!valid) {           // keep table indices in range on padding lanes
#pragma unroll
        for (int k = 0; k < P0; ++k) p0[k] = 0;
    }
    // bits 0..7: every node of this alignment is on tile path p (the filter of
    // src/eval.cpp:81-91); bit 31 survives iff NO step is on the tile's first node
    uint32_t pm = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < P0; ++k) {
        pm &= node_mask<NM8>(tv, (p0[k] & 0xFFFFu) >> 1);
        if (k + 1 < P0 || !(M & 1)) pm &= node_mask<NM8>(tv, p0[k] >> 17);
    }
    uint32_t pass = a.filter ? pm : 0xFFu;
    pass = valid ? (pass & tv.sub_mask) : 0u;
    const bool has_a0 = !tv.uniform_a0 || (pm & (NM8 ? NOT_A0_8 : NOT_A0)) == 0u;
    const uint32_t gt = pass & tv.gt_mask;          // src/alignments.cpp:500: m > n -> good
    const uint32_t todo = pass & ~tv.gt_mask;
#ifdef GFAL_STAMPS
    asm volatile("" ::"v"(todo), "v"(gt));
    STAMP(st1);
#endif

    // ---- window lookup: which of the tile's paths contain this step sequence ----
    uint32_t fmask = 0;
    {
        bool searching = todo != 0u;
        while (true) {
            // probe on until an empty slot or an entry with this fingerprint
            while (true) {
                const bool stop = !searching || e == H_EMPTY || ((e ^ h) >> H_FP_SHIFT) == 0u;
                if (!WAVE_ANY(!stop)) break;
                slot4 = stop ? slot4 : ((slot4 + stride4) & (4u * H_SLOTS - 1u));
                const uint32_t e2 = *reinterpret_cast<const uint32_t *>(tb + slot4);
                const uint32_t m2 = *reinterpret_cast<const uint32_t *>(tb + 4u * H_SLOTS + (slot4 >> 2 & ~3u));
                e = stop ? e : e2;
                mword = stop ? mword : m2;
#ifdef GFAL_STAMPS
                wc.stamps.v[5] += 1;      // extra probe rounds
#endif
            }
            const bool cand = searching && e != H_EMPTY;
            if (!WAVE_ANY(cand)) break;
            // exact comparison with the entry's window: P0 dwords of the path's steps
            // from the window's first step on (lanes without a candidate read the
            // start of the steps and are masked off)
            const uint32_t idx = cand ? (e & ((1u << H_FP_SHIFT) - 1u)) : 0u;
            const uint32_t *wd = step32 + (idx >> 1);
            const uint32_t sh = idx << 4;               // alignbit uses bits 4:0: 0 or 16
            uint32_t d[P0 + 1];
#pragma unroll
            for (int k = 0; k <= P0; ++k) d[k] = wd[k];
            bool ok = cand;
#pragma unroll
            for (int k = 0; k < P0; ++k) {
                const uint32_t got = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
                if (k + 1 < P0 || !(M & 1)) ok &= got == p0[k];
                else ok &= ((got ^ p0[k]) & 0xFFFFu) == 0u;       // M odd: one step in the last pair
            }
            if (ok) fmask = (mword >> ((slot4 << 1) & 24u)) & 0xFFu;
            // a fingerprint collision (cand && !ok) keeps probing; everything else is settled
            searching = cand && !ok;
            if (!WAVE_ANY(searching)) break;
            slot4 = searching ? ((slot4 + stride4) & (4u * H_SLOTS - 1u)) : slot4;
            e = *reinterpret_cast<const uint32_t *>(tb + slot4);
            mword = *reinterpret_cast<const uint32_t *>(tb + 4u * H_SLOTS + (slot4 >> 2 & ~3u));
        }
    }
    fmask &= todo;
    const uint32_t good_mask = gt | fmask;
    const uint32_t open = todo & ~fmask;
    uint32_t bad_mask = open;
#ifdef GFAL_STAMPS
    asm volatile("" ::"v"(open));
    STAMP(st2);
#endif

    // ---- start-overhang triage (as in k_scan): B is not a subpath and m <= n ----
    // The traceback stays free only if a proper suffix of B (or of rc(B)) equals a
    // prefix of the path; survivors of this exact test go to the DP kernels.  Few
    // items have such lanes (the alignment must touch the tile's first node), but
    // a wave that gets here holds up its whole workgroup: everything comes from
    // registers and LDS, and the item makes ONE returning atomic for all its pairs.
    if (WAVE_ANY(open != 0u && has_a0)) {
#ifdef GFAL_STAMPS
        wc.stamps.v[6] += 1;                         // items that enter the triage
        wc.stamps.v[7] += __popcll(WAVE_MASK(open != 0u && has_a0));    // lanes that ask for it
#endif
        uint32_t cfw = 0, crc = 0;                   // bit p: candidate on tile path p
        // step t of B, t a compile-time constant after unrolling
        auto stepB = [&](int t) -> uint32_t { return (p0[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu; };
        for (int p = 0; p < tv.tile_paths; ++p) {
            const bool mine = ((open >> p) & 1u) != 0u && has_a0;
            if (!WAVE_ANY(mine)) continue;
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, p);
            const uint32_t *pstep32 = step32 + (uint32_t)p * (uint32_t)tv.nm;   // 2 * nm u16 per path
            bool cand_fw = false, cand_rc = false;
            if constexpr (MC != 0) {
                // the path's first M steps (the same for every lane: broadcast reads)
                uint32_t pre[P0];
#pragma unroll
                for (int k = 0; k < P0; ++k) pre[k] = pstep32[k];
                auto stepA = [&](int k) -> uint32_t { return (pre[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu; };
#pragma unroll
                for (int t = 0; t < MC; ++t) {
                    const uint32_t bt = stepB(t);
                    if (t >= 1) {                     // B[t..M) == path[0..M-t) ?
                        bool eq = mine && bt == a0;
#pragma unroll
                        for (int k = 1; k < MC - t; ++k) eq &= stepB(t + k) == stepA(k);
                        cand_fw |= eq;
                    }
                    if (t < MC - 1) {                 // rc(B)[M-1-t..M) == path[0..t+1) ?
                        bool eq = mine && (bt ^ 1u) == a0;
#pragma unroll
                        for (int k = 1; k <= t; ++k) eq &= (stepB(t - k) ^ 1u) == stepA(k);
                        cand_rc |= eq;
                    }
                }
            } else {
                const uint16_t *bp = a.items.steps +
                                     ((size_t)sg.step_base + (size_t)(r.it - sg.item_lo) * (uint32_t)M) * WAVE + lane;
                for (int t = 0; t < M; ++t) {      // (rare lengths: steps re-read from the item)
                    const uint32_t bt = bp[t * WAVE];
                    const bool live_fw = mine && t >= 1 && bt == a0;
                    if (WAVE_ANY(live_fw)) cand_fw |= tail_equals(bp, t, 1, M - t, 0u, pstep32, live_fw);
                    const bool live_rc = mine && t < M - 1 && (bt ^ 1u) == a0;
                    if (WAVE_ANY(live_rc)) cand_rc |= tail_equals(bp, t, -1, t + 1, 1u, pstep32, live_rc);
                }
            }
            cfw |= cand_fw ? (1u << p) : 0u;
            crc |= cand_rc ? (1u << p) : 0u;
        }
        const uint32_t cany = cfw | crc;
        bad_mask &= ~cany;
        push_item_pairs(a.worklist, a.wl_count, a.wl_capacity, a.wl_hist, a.status, a.n_paths, cfw, crc,
                        lane, (uint32_t)tv.path0, tv.tile_paths, r.it * WAVE + (uint32_t)lane, M);
    }
#ifdef GFAL_STAMPS
    asm volatile("" ::"v"(bad_mask));
    STAMP(st3);
#endif
    if constexpr (W) wc.add(good_mask, bad_mask, w);
    else wc.add(good_mask, bad_mask, tv.tile_paths);
#ifdef GFAL_STAMPS
    STAMP(st4);
    wc.stamps.v[0] += st1 - st0;     // item regs ready + node masks + first probe
    wc.stamps.v[1] += st2 - st1;     // window lookup
    wc.stamps.v[2] += st3 - st2;     // triage
    wc.stamps.v[3] += st4 - st3;     // counting
    wc.stamps.v[4] += 1;             // items
#endif
}

// The same for alignments too long for registers (M > 2 * MAX_REG_K + 1): steps are
// re-read from the item where needed.
template <bool W, bool NM8, typename Counts>
__device__ __forceinline__ void scan2_item_long(const Scan2Args &a, const Tile2 &tv,
                                                const uint16_t *__restrict__ bp, uint32_t h, int M,
                                                int lane, uint32_t slot_id, Counts &wc, uint32_t w)
{
    const uint32_t first_step = bp[0];
    const bool valid = first_step != STEP_INVALID;
    uint32_t pm = 0xFFFFFFFFu;
    for (int t = 0; t < M; ++t) {
        const uint32_t bt = valid ? (uint32_t)bp[t * WAVE] : 0u;
        pm &= node_mask<NM8>(tv, bt >> 1);
    }
    uint32_t pass = a.filter ? pm : 0xFFu;
    pass = valid ? (pass & tv.sub_mask) : 0u;
    const bool has_a0 = !tv.uniform_a0 || (pm & (NM8 ? NOT_A0_8 : NOT_A0)) == 0u;
    const uint32_t gt = pass & tv.gt_mask;
    const uint32_t todo = pass & ~tv.gt_mask;
    uint32_t fmask = 0;
    {
        const uint32_t stride4 = (((h >> H_LOG_S) & (H_SLOTS - 1u)) | 1u) << 2;
        uint32_t slot4 = (h & (H_SLOTS - 1u)) << 2;
        const uint8_t *tb = reinterpret_cast<const uint8_t *>(tv.table);
        bool searching = todo != 0u;
        while (WAVE_ANY(searching)) {
            uint32_t e;
            while (true) {
                e = *reinterpret_cast<const uint32_t *>(tb + slot4);
                const bool stop = !searching || e == H_EMPTY || ((e ^ h) >> H_FP_SHIFT) == 0u;
                if (!WAVE_ANY(!stop)) break;
                slot4 = stop ? slot4 : ((slot4 + stride4) & (4u * H_SLOTS - 1u));
            }
            const bool cand = searching && e != H_EMPTY;
            const uint16_t *win = tv.steps + (cand ? (e & ((1u << H_FP_SHIFT) - 1u)) : 0u);
            bool ok = cand;
            for (int t = 0; t < M; ++t) ok &= (uint32_t)win[cand ? t : 0] == (uint32_t)bp[t * WAVE];
            if (ok) fmask = tb[4u * H_SLOTS + (slot4 >> 2)];
            searching = cand && !ok;
            slot4 = searching ? ((slot4 + stride4) & (4u * H_SLOTS - 1u)) : slot4;
        }
    }
    fmask &= todo;
    const uint32_t good_mask = gt | fmask;
    const uint32_t open = todo & ~fmask;
    uint32_t bad_mask = open;
    if (WAVE_ANY(open != 0u && has_a0)) {
        const uint32_t *step32 = reinterpret_cast<const uint32_t *>(tv.steps);
        for (int p = 0; p < tv.tile_paths; ++p) {
            const bool mine = ((open >> p) & 1u) != 0u && has_a0;
            if (!WAVE_ANY(mine)) continue;
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, p);
            const uint32_t *pstep32 = step32 + (uint32_t)p * (uint32_t)tv.nm;
            bool cand_fw = false, cand_rc = false;
            for (int t = 0; t < M; ++t) {
                const uint32_t bt = bp[t * WAVE];
                const bool live_fw = mine && t >= 1 && bt == a0;
                if (WAVE_ANY(live_fw)) cand_fw |= tail_equals(bp, t, 1, M - t, 0u, pstep32, live_fw);
                const bool live_rc = mine && t < M - 1 && (bt ^ 1u) == a0;
                if (WAVE_ANY(live_rc)) cand_rc |= tail_equals(bp, t, -1, t + 1, 1u, pstep32, live_rc);
            }
            if (cand_fw || cand_rc) bad_mask &= ~(1u << p);
            push_pairs_to(a.worklist, a.wl_count, a.wl_capacity, a.wl_hist, a.status, a.n_paths,
                          cand_fw, cand_rc, lane, (uint32_t)(tv.path0 + p), slot_id, M);
        }
    }
    if constexpr (W) wc.add(good_mask, bad_mask, w);
    else wc.add(good_mask, bad_mask, tv.tile_paths);
}

// The wave's items of one segment chunk, 64 at a time (as in k_scan: one ballot
// drops the items none of whose lanes can pass the filter).  P0 = ceil(M / 2)
// pair dwords per lane in registers, or -1: long alignments.
template <int P0, int MC, bool W, bool NM8, typename Counts>
__device__ __forceinline__ void scan2_items(const Scan2Args &a, const Tile2 &tv, const LenSeg &sg,
                                            int chunk, int wave, int lane, Counts &wc)
{
    const int n_chunks = (int)sg.n_chunks;
    const int item_stride = SCAN2_WAVES * n_chunks;
    const int M = (int)sg.m;
    const uint32_t ulane = (uint32_t)lane;
    for (int it0 = (int)sg.item_lo + chunk + wave * n_chunks; it0 < (int)sg.item_hi;
         it0 += WAVE * item_stride) {
        const int my_it = it0 + lane * item_stride;
        const bool mine = my_it < (int)sg.item_hi;
        bool keep = mine;
        if (a.filter && mine) {
            const uint32_t common = a.items.common[my_it];
            if (common != NO_COMMON_NODE) {
                const uint32_t m1 = node_mask<NM8>(tv, common & 0x7FFFu), m2 = node_mask<NM8>(tv, common >> 16);
                keep = (((common & COMMON_EITHER) ? (m1 | m2) : (m1 & m2)) & tv.sub_mask) != 0u;
            }
        }
        lanemask todo = WAVE_MASK(keep);
        if (todo == 0) continue;
        if constexpr (P0 > 0 && P0 <= 6) {
            // The loads of an item are issued while the previous one is decided.  Two
            // register sets, used in turn (no copies), and the load ahead is
            // UNCONDITIONAL (past the last item it re-reads that item): at every use
            // exactly one younger set of loads is in flight, so the compiler can wait
            // with a counted vmcnt instead of draining the loads issued ahead.
            auto load_item = [&](int src_in, ItemRegs<P0> &r) {
                const int src = __builtin_amdgcn_readfirstlane(src_in);     // (uniform: from a ballot)
                const uint32_t it = (uint32_t)(it0 + src * item_stride);
                r.it = it;
                r.h = sgpr_ptr(a.item_hash + (size_t)it * WAVE)[ulane];
                r.w = W ? sgpr_ptr(a.items.weight + (size_t)it * WAVE)[ulane] : 1u;
                const GLOBAL_AS uint32_t *pp =
                    sgpr_ptr(a.pairs0 + ((size_t)sg.p0_base + (size_t)(it - sg.item_lo) * P0) * WAVE) + ulane;
#pragma unroll
                for (int k = 0; k < P0; ++k) r.p0[k] = pp[k * WAVE];
            };
            ItemRegs<P0> ra, rb;
            load_item(__builtin_ctzll(todo), ra);
            while (true) {
                lanemask rest = todo & (todo - 1);
                load_item(__builtin_ctzll(rest ? rest : todo), rb);
                scan2_item<P0, MC, W, NM8>(a, tv, sg, ra, M, lane, wc);
                if (rest == 0) break;
                todo = rest;
                rest = todo & (todo - 1);
                load_item(__builtin_ctzll(rest ? rest : todo), ra);
                scan2_item<P0, MC, W, NM8>(a, tv, sg, rb, M, lane, wc);
                if (rest == 0) break;
                todo = rest;
            }
        } else {      // long alignments are rare: no second register set for them
            for (; todo != 0; todo &= todo - 1) {
                const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                const uint32_t it = (uint32_t)(it0 + src * item_stride);
                const uint32_t h = a.item_hash[(size_t)it * WAVE + ulane];
                const uint32_t w = W ? a.items.weight[(size_t)it * WAVE + ulane] : 1u;
                if constexpr (P0 < 0) {
                    const uint16_t *bp = a.items.steps +
                                         ((size_t)sg.step_base + (size_t)(it - sg.item_lo) * (uint32_t)M) * WAVE + ulane;
                    scan2_item_long<W, NM8>(a, tv, bp, h, M, lane, it * WAVE + ulane, wc, w);
                } else {
                    ItemRegs<P0> r;
                    r.it = it;
                    r.h = h;
                    r.w = w;
                    const uint32_t *pp = a.pairs0 + ((size_t)sg.p0_base + (size_t)(it - sg.item_lo) * P0) * WAVE + ulane;
#pragma unroll
                    for (int k = 0; k < P0; ++k) r.p0[k] = pp[k * WAVE];
                    scan2_item<P0, MC, W, NM8>(a, tv, sg, r, M, lane, wc);
                }
            }
        }
    }
#ifdef GFAL_STAMPS
    if (lane == 0)
        for (int k = 0; k < 8; ++k)
            if (wc.stamps.v[k]) atomicAdd(&g_stamp_sum[k], wc.stamps.v[k]);
#endif
}

// G: the alignment lengths an instantiation knows -- 0: 1..4 steps, 1: 5..8 (exact
// lengths), 2: 9..16, 3: 17..33 (pair count known, length at run time), 4: longer (steps
// re-read from the item).  One function with the item loops of all lengths (30 instantiations) spilled
// 93 VGPRs at the 64-register budget of 8 waves per SIMD (703 MB of scratch writes per
// launch at config 3); a function with the loops of four lengths does not spill.
// G == SCAN2_GROUPS is that one function: a batch of a few thousand paths is better off
// with one launch that spills than with several that each end in a tail of prologues
// (2048 paths: 1.36 against 1.52 ms per call); from 4096 paths on the split ones run.
constexpr int SCAN2_GROUPS = 5;
constexpr int scan2_group(int M)
{
    return M <= 4 ? 0 : M <= 8 ? 1 : M <= 16 ? 2 : M <= 2 * MAX_REG_K + 1 ? 3 : 4;
}

template <bool W, bool NM8, int G>
__global__ __launch_bounds__(SCAN2_THREADS, SCAN2_WAVES_PER_SIMD) void k_scan2(Scan2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    // wave-uniform by construction; saying so keeps item indices and base
    // pointers in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_id = blockIdx.x % a.n_tiles;
    // (segment, chunk) <- the workgroup's row: lane k sizes segment k, a prefix sum
    // over the lanes finds the row's segment
    LenSeg sg;
    int chunk;
    {
        const int y = blockIdx.x / a.n_tiles;
        LenSeg mine{0u, 0u, 0u, 0u, 0u, 0u};
        if (lane < a.n_segs) mine = a.segs[lane];
        const uint32_t c = lane < a.n_segs
                               ? seg_chunks(mine.item_hi - mine.item_lo, a.chunk_mult, a.chunk_inv_min)
                               : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, WAVE);
            if (lane >= o) incl += v;
        }
        const lanemask beyond = WAVE_MASK(incl > (uint32_t)y);    // first set lane: my segment
        const int seg = beyond ? __builtin_ctzll(beyond) : 0;
        sg.item_lo = (uint32_t)__builtin_amdgcn_readlane((int)mine.item_lo, seg);
        sg.item_hi = (uint32_t)__builtin_amdgcn_readlane((int)mine.item_hi, seg);
        sg.m = (uint32_t)__builtin_amdgcn_readlane((int)mine.m, seg);
        sg.n_chunks = (uint32_t)__builtin_amdgcn_readlane((int)c, seg);
        sg.step_base = (uint32_t)__builtin_amdgcn_readlane((int)mine.step_base, seg);
        sg.p0_base = (uint32_t)__builtin_amdgcn_readlane((int)mine.p0_base, seg);
        chunk = y - (int)__builtin_amdgcn_readlane((int)(incl - c), seg);
    }
    const int M = (int)sg.m;

#ifdef GFAL_STAMPS
    unsigned long long wg0, wg1, wg2, wg3, wg_items = 0;
    STAMP(wg0);
#endif
    Tile2 tv;
    tv.path0 = tile_id * a.tile;
    tv.tile_paths = min(a.tile, a.n_paths - tv.path0);
    tv.nm = a.L.nm;
    const int nm = a.L.nm;
    uint16_t *steps = lds;                                          // [tile][2][nm]
    uint32_t *nodemask = reinterpret_cast<uint32_t *>(lds + (size_t)a.tile * 2 * nm);
    uint32_t *table = nodemask + (NM8 ? (a.L.v2 + 3) / 4 : a.L.v2);
    uint32_t *maskw = table + H_SLOTS;                              // H_SLOTS bytes
    uint32_t *misc = maskw + H_SLOTS / 4;                           // [0]: entries in the table
    tv.steps = steps;
    tv.nodemask = nodemask;
    tv.table = table;
    tv.maskb = reinterpret_cast<const uint8_t *>(maskw);

    // Stage the steps of the tile's paths (forward | reverse complement: contiguous
    // in the image, 4-byte aligned) and build the node masks from the node ids along
    // the paths.  Every global load of the prologue is issued before the first one is
    // waited for (one exposed memory latency, not one per loop trip and path).
    // (a) the steps go straight from global memory into LDS (LDS-DMA: no registers
    // in between; lane l of a wave-instruction lands at its uniform LDS base + 4 l)
    {
        const int chunks = (nm + WAVE - 1) / WAVE;                        // 64 dwords each
        for (int c = wave; c < tv.tile_paths * chunks; c += SCAN2_WAVES) {
            const int p = c / chunks, o = (c % chunks) * WAVE;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(
                a.images + (size_t)(tv.path0 + p) * a.L.total + a.L.step_at()) + o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(steps + (size_t)p * 2 * nm) + o;
            if (o + lane < nm)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + lane),
                    (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
        }
    }
    // (b) the node ids along the paths: a few dwords per thread, in registers
    constexpr int PER_PATH = SCAN2_THREADS / TILE2_MAX;                  // threads per path (NM8: 7 paths use them)
    constexpr int LID_LOADS = ((GFAL_MAX_STEPS + 8) / 2 + PER_PATH - 1) / PER_PATH;     // nm / 2 dwords
    const int my_p = tid / PER_PATH, my_q = tid % PER_PATH;
    const bool have_p = my_p < tv.tile_paths;
    uint32_t lreg[LID_LOADS];
    {
        const uint32_t *lsrc = reinterpret_cast<const uint32_t *>(
            a.lids + (size_t)(tv.path0 + (have_p ? my_p : 0)) * nm);
#pragma unroll
        for (int k = 0; k < LID_LOADS; ++k) {
            const int o = my_q + k * PER_PATH;
            lreg[k] = (have_p && o < nm / 2) ? lsrc[o] : 0xFFFFFFFFu;
        }
    }
    tv.hdr_n = 0;
    if (lane < tv.tile_paths)
        tv.hdr_n = a.images[(size_t)(tv.path0 + lane) * a.L.total + a.L.len_at()];
    // meanwhile: node masks, table and counters start empty
    if constexpr (NM8) {
        for (int v = tid; v < (a.L.v2 + 3) / 4; v += SCAN2_THREADS) nodemask[v] = NOT_A0_8 * 0x01010101u;
    } else {
        for (int v = tid; v < a.L.v2; v += SCAN2_THREADS) nodemask[v] = NOT_A0;
    }
    for (int i = tid; i < H_SLOTS; i += SCAN2_THREADS) table[i] = H_EMPTY;
    for (int i = tid; i < H_SLOTS / 4; i += SCAN2_THREADS) maskw[i] = 0;
    if (tid < 16) misc[tid] = 0;               // [0] entries, [1] overflow flag, [2 + t] common prefixes
    __syncthreads();
    if (have_p) {
        // which tile paths carry each node (the filter of src/eval.cpp:81-91 as a bit test)
#pragma unroll
        for (int k = 0; k < LID_LOADS; ++k) {
            const uint32_t lo = lreg[k] & 0xFFFFu, hi = lreg[k] >> 16;
            if constexpr (NM8) {
                if (lo != 0xFFFFu) atomicOr(&nodemask[lo >> 2], (1u << my_p) << ((lo & 3u) * 8u));
                if (hi != 0xFFFFu) atomicOr(&nodemask[hi >> 2], (1u << my_p) << ((hi & 3u) * 8u));
            } else {
                if (lo != 0xFFFFu) atomicOr(&nodemask[lo], 1u << my_p);
                if (hi != 0xFFFFu) atomicOr(&nodemask[hi], 1u << my_p);
            }
        }
    }
    __syncthreads();
    // the tiles's first steps; the node of path 0's first step loses NOT_A0
    tv.hdr_a0 = STEP_NOMATCH;
    if (lane < tv.tile_paths) tv.hdr_a0 = steps[(size_t)lane * 2 * nm];
    const uint32_t tile_a0 = (uint32_t)__builtin_amdgcn_readlane((int)tv.hdr_a0, 0);
    tv.uniform_a0 = WAVE_MASK(lane < tv.tile_paths && tv.hdr_a0 != tile_a0) == 0ull;
    if (tid == 0 && tile_a0 < STEP_NOMATCH) {
        const uint32_t v = tile_a0 >> 1;
        if constexpr (NM8) atomicAnd(&nodemask[v >> 2], ~(NOT_A0_8 << ((v & 3u) * 8u)));
        else atomicAnd(&nodemask[v], ~NOT_A0);
    }
    __syncthreads();
#ifdef GFAL_STAMPS
    unsigned long long pg1, pg2, pg3;
    STAMP(pg1);
    if (lane == 0) atomicAdd(&g_stamp_wg[5], pg1 - wg0);      // staging + node masks
#endif

    uint32_t cnt_good = 0, cnt_bad = 0;      // lane p: totals of tile path p over all passes
    // passes over the tile's paths: as many paths per pass as fit the table
    int t0 = 0;
    bool fresh_table = true;                   // the prologue has just cleared it
    while (t0 < tv.tile_paths) {
        int t1 = t0, limit = tv.tile_paths;
        uint32_t gt_mask = 0;
        bool one_by_one = false;
        while (true) {                         // further rounds only after an overflow
            __syncthreads();
            if (!fresh_table) {
                for (int i = tid; i < H_SLOTS; i += SCAN2_THREADS) table[i] = H_EMPTY;
                for (int i = tid; i < H_SLOTS / 4; i += SCAN2_THREADS) maskw[i] = 0;
                if (tid < 2) misc[tid] = 0;    // [0] entries, [1] overflow flag
            }
            fresh_table = false;
            // the first path of the pass that has windows, and what the others share with it
            int base = -1, n_base = 0;
            for (int t = t0; t < limit; ++t) {
                const int n = __builtin_amdgcn_readlane(tv.hdr_n, t);
                if (n >= M && base < 0) {
                    base = t;
                    n_base = n;
                }
                if (tid == 0) misc[2 + t] = (uint32_t)min(n, n_base);     // [2 + t]: common prefix with base
            }
            __syncthreads();
            for (int t = base + 1; base >= 0 && t < limit; ++t) {
                const int n = __builtin_amdgcn_readlane(tv.hdr_n, t);
                if (n >= M) common_prefix(steps, nm, base, t, min(n, n_base), &misc[2 + t], tid);
            }
            __syncthreads();
#ifdef GFAL_STAMPS
            STAMP(pg2);
            if (lane == 0) atomicAdd(&g_stamp_wg[6], pg2 - pg1);      // table clear + common prefixes
#endif
            t1 = t0;
            gt_mask = 0;
            bool overflow = false;
            if (!one_by_one) {
                // all paths of the pass at once, no barrier between them (similar paths
                // share their windows: this nearly always fits); if it does not, the
                // table is rebuilt path by path to find out how many do fit
                for (; t1 < limit; ++t1) {
                    const int n = __builtin_amdgcn_readlane(tv.hdr_n, t1);
                    if (n < M) gt_mask |= 1u << t1;      // no windows: every passing alignment is good
                    else
                        insert_windows(steps, nm, table, maskw, misc, t1, n, M, tid, base, n_base,
                                       misc + 2, t0, limit);
                }
                __syncthreads();
                if (misc[1] == 0u) break;
                one_by_one = true;
                continue;
            }
            while (t1 < limit) {
                const int n = __builtin_amdgcn_readlane(tv.hdr_n, t1);
                if (n < M) {
                    gt_mask |= 1u << t1;
                } else {
                    insert_windows(steps, nm, table, maskw, misc, t1, n, M, tid, -1, 0, misc + 2, t0,
                                   t1 + 1);          // (no sharing: the pass's extent is not known yet)
                    __syncthreads();
                    overflow = misc[1] != 0u;
                    if (overflow) break;
                }
                ++t1;
            }
            if (!overflow) break;
            // path t1 did not fit beside [t0, t1): rebuild the table without it (the
            // masks already carry its bit); it starts the next pass.  One path alone
            // always fits (< 2000 windows), so t1 > t0 here.
            limit = t1;
        }
        if (t1 == t0) t1 = t0 + 1;            // (unreachable; never loop forever)
        tv.sub_mask = ((t1 >= 32 ? 0u : (1u << t1)) - 1u) & ~((1u << t0) - 1u);
        tv.gt_mask = gt_mask;

#ifdef GFAL_STAMPS
        STAMP(wg1);
        if (lane == 0) atomicAdd(&g_stamp_wg[7], wg1 - pg2);          // window inserts
#endif
        LenSeg sgl = sg;
        if (a.debug == 1) sgl.item_hi = sgl.item_lo;      // timing probe: no items
        if (a.debug >= 2 && M != a.debug) sgl.item_hi = sgl.item_lo;   // timing probe: one length only
#define GFAL_RUN(KK, MM)                                                        \
    do {                                                                        \
        if constexpr (W) {                                                      \
            WaveCounts2W wc;                                                    \
            scan2_items<KK, MM, true, NM8>(a, tv, sgl, chunk, wave, lane, wc);       \
            wc.flush();                                                         \
            cnt_good += wc.good;                                                \
            cnt_bad += wc.bad;                                                  \
        } else {                                                                \
            WaveCounts2 wc;                                                     \
            scan2_items<KK, MM, false, NM8>(a, tv, sgl, chunk, wave, lane, wc);      \
            wc.flush();                                                         \
            cnt_good += wc.good;                                                \
            cnt_bad += wc.bad;                                                  \
        }                                                                       \
    } while (0)
#ifdef GFAL_ONLY_M      // codegen experiment: a kernel that knows one length only
        if (M == GFAL_ONLY_M) GFAL_RUN((GFAL_ONLY_M + 1) / 2, GFAL_ONLY_M);
#else
        constexpr bool ALL = G == SCAN2_GROUPS;       // the instantiation that knows every length
        const int grp = scan2_group(M);
#define GFAL_CASE(MM) case MM: GFAL_RUN((MM + 1) / 2, MM); break;
        if constexpr (ALL || G == 0)
            if (grp == 0) switch (M) { GFAL_CASE(1) GFAL_CASE(2) GFAL_CASE(3) GFAL_CASE(4) }
        if constexpr (ALL || G == 1)
            if (grp == 1) switch (M) { GFAL_CASE(5) GFAL_CASE(6) GFAL_CASE(7) GFAL_CASE(8) }
#undef GFAL_CASE
#define GFAL_CASE(PP) case PP: GFAL_RUN(PP, 0); break;      // pair dwords; M itself at run time
        if constexpr (ALL || G == 2)
            if (grp == 2) switch ((M + 1) / 2) { GFAL_CASE(5) GFAL_CASE(6) GFAL_CASE(7) GFAL_CASE(8) }
        if constexpr (ALL || G == 3)
            if (grp == 3)
                switch ((M + 1) / 2) {
                    GFAL_CASE(9) GFAL_CASE(10) GFAL_CASE(11) GFAL_CASE(12) GFAL_CASE(13)
                    GFAL_CASE(14) GFAL_CASE(15) GFAL_CASE(16) GFAL_CASE(17)
                }
        if constexpr (ALL || G == 4)
            if (grp == 4) GFAL_RUN(-1, 0);
#undef GFAL_CASE
#endif
#undef GFAL_RUN
#ifdef GFAL_STAMPS
        STAMP(wg2);
        wg_items += wg2 - wg1;
#endif
        t0 = t1;
        __syncthreads();                       // the table is rebuilt for the next pass
#ifdef GFAL_STAMPS
        STAMP(wg3);
        if (lane == 0) {
            atomicAdd(&g_stamp_wg[2], wg3 - wg2);      // waiting for the slowest wave
            atomicAdd(&g_stamp_wg[4], 1ull);           // passes x waves
        }
#endif
    }
#ifdef GFAL_STAMPS
    if (lane == 0) {
        atomicAdd(&g_stamp_wg[1], wg_items);
        atomicAdd(&g_stamp_wg[0], wg3 - wg0 - wg_items);     // everything that is not the item loop
        atomicAdd(&g_stamp_wg[3], 1ull);
    }
#endif

    // workgroup reduction through LDS (the steps are dead now), then one atomic
    // per counter per workgroup
    __syncthreads();
    uint32_t *red = reinterpret_cast<uint32_t *>(lds);
    if (tid < 2 * MAX_TILE) red[tid] = 0;
    __syncthreads();
    if (lane < tv.tile_paths) {
        if (cnt_bad) atomicAdd(&red[lane], cnt_bad);
        if (cnt_good) atomicAdd(&red[MAX_TILE + lane], cnt_good);
    }
    __syncthreads();
    if (tid < tv.tile_paths) {
        uint32_t d = red[tid], g = red[MAX_TILE + tid];
        if (d) atomicAdd(&a.counts[tv.path0 + tid], d);
        if (g) atomicAdd(&a.counts[a.n_paths + tv.path0 + tid], g);
    }
}

// --------------------------------------------------------------------------
// Exact decision: Needleman-Wunsch fill (src/alignments.cpp:499-509) with the
// traceback (src/alignments.cpp:511-554) folded into the forward pass.
//
// The traceback's direction at a cell depends only on the table, and its cost
// equals the table's own move cost everywhere except on row 0 / column 0,
// which it walks for free.  Hence  score = dp[n][m] - dp[exit]  where `exit`
// is the first row-0/column-0 cell the traceback reaches.  exit-values are
// propagated forward with the table, one row at a time:  X[i][j] =
// X[predecessor chosen by the traceback at (i,j)].
//
// Row state lives in `row` (one packed {dp,X} int16 pair per column), strided
// so that neighbouring threads touch neighbouring words.
// --------------------------------------------------------------------------
struct StepsA {   // candidate path steps inside an image (HBM)
    const uint16_t *step;
    int n;
};
struct StepsB {   // one alignment inside an item
    const uint16_t *p;   // step 0 of this lane
    int m;
    uint32_t flip;       // 0: as stored, 1: reverse complement
    __device__ __forceinline__ uint32_t at(int j) const
    {
        return flip ? ((uint32_t)p[(m - 1 - j) * WAVE] ^ 1u) : (uint32_t)p[j * WAVE];
    }
};

__device__ __forceinline__ uint32_t pack_cell(int dp, int x)
{
    return ((uint32_t)dp & 0xFFFFu) | ((uint32_t)x << 16);
}
__device__ __forceinline__ int cell_dp(uint32_t c) { return (int)(int16_t)(c & 0xFFFFu); }
__device__ __forceinline__ int cell_x(uint32_t c) { return (int)(int16_t)(c >> 16); }

__device__ __forceinline__ int traceback_score(const StepsA &A, const StepsB &B,
                                               uint32_t *row, int stride)
{
    const int n = A.n, m = B.m;
    for (int j = 0; j <= m; ++j) {
        int v = (j <= n) ? -j : 0;           // :500, row 0 reaches column n only
        row[(size_t)j * stride] = pack_cell(v, v);
    }
    for (int i = 1; i <= n; ++i) {
        const uint32_t ai = A.step[i - 1];
        uint32_t c = row[0];
        int diag_dp = cell_dp(c), diag_x = cell_x(c);
        row[0] = pack_cell(0, 0);            // column 0 is never written: 0
        int left_dp = 0, left_x = 0;
        for (int j = 1; j <= m; ++j) {
            c = row[(size_t)j * stride];
            const int up_dp = cell_dp(c), up_x = cell_x(c);
            const int sub = (ai == B.at(j - 1)) ? 0 : -1;
            const int d = diag_dp + sub;
            const int u = up_dp + (j < m ? -1 : 0);   // :504 free in last column
            const int l = left_dp - 1;
            const int v = max(d, max(u, l));
            int x;
            if (v == d) x = diag_x;                    // :527
            else if (up_dp >= left_dp) x = up_x;       // :534
            else x = left_x;                           // :541
            row[(size_t)j * stride] = pack_cell(v, x);
            diag_dp = up_dp;
            diag_x = up_x;
            left_dp = v;
            left_x = x;
        }
    }
    uint32_t c = row[(size_t)m * stride];
    return cell_dp(c) - cell_x(c);
}

// Row state: one word per column per thread.  Short alignments keep it in LDS
// ([column][thread], conflict-free); long ones use the HBM scratch.
template <bool ROWS_IN_LDS>
__device__ __forceinline__ uint32_t *dp_row(uint32_t *row_scratch, int &stride)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dp_lds[];
    if (ROWS_IN_LDS) {
        stride = DP_THREADS;
        return dp_lds + threadIdx.x;
    }
    stride = gridDim.x * DP_THREADS;
    return row_scratch + blockIdx.x * DP_THREADS + threadIdx.x;
}

// The same fill with the row in registers: MC columns, fully unrolled, one
// table row per call.  b[] holds the (oriented) alignment, padded with
// STEP_INVALID; columns beyond the lane's own m are computed and ignored
// (nothing flows from a column to its left).  Column 0 is never written by
// the reference: it is 0 in every row.
//
// A cell is ONE word: dp * 2^14 + (-x), x = the exit value that travels with the
// traceback (0 .. -MC: it is set in row 0 / column 0 only, src/alignments.cpp:500),
// bits 13..12 free.  The three candidates get their rank in the reference's tie order
// in those two bits -- diagonal 3 (:527, `v == d` first), up 2 and left 1 in the columns
// before the last (:534, `up >= left`), up 0 in the lane's last column (there up is free
// and the same test reads `u > l`) -- so one signed max3 picks the value AND carries the
// winner's exit value: 7 VALU per cell (compare, select, three adds, max3, and) against
// 12 with dp and x in separate registers (two compares and two selects for x alone).
constexpr int DPK_SHIFT = 14;
constexpr int DPK_ONE = 1 << DPK_SHIFT;
constexpr int DPK_RANK = 1 << 12;
constexpr int DPK_X_MASK = DPK_RANK - 1;

__device__ __forceinline__ int dpk_pack(int dp, int x) { return dp * DPK_ONE - x; }
__device__ __forceinline__ int dpk_dp(int c) { return c >> DPK_SHIFT; }
__device__ __forceinline__ int dpk_x(int c) { return -(c & DPK_X_MASK); }

// what the `up` candidate adds in column j (1-based) of a lane whose alignment has m steps
__device__ __forceinline__ int dpk_up_add(int j, int m) { return j < m ? 2 * DPK_RANK - DPK_ONE : 0; }

template <int MC>
__device__ __forceinline__ void dp_row_regs(uint32_t ai, const uint32_t (&b)[MC], const int (&up_add)[MC],
                                            int (&c)[MC + 1])
{
    int diag = 0, left = 0;            // column 0: dp 0, exit value 0
#pragma unroll
    for (int j = 1; j <= MC; ++j) {
        const int up = c[j];
        const int d = diag + ((ai == b[j - 1]) ? 3 * DPK_RANK : 3 * DPK_RANK - DPK_ONE);
        const int u = up + up_add[j - 1];
        const int l = left + (DPK_RANK - DPK_ONE);
        const int v = max(d, max(u, l)) & ~(3 * DPK_RANK);
        c[j] = v;
        diag = up;
        left = v;
    }
}

// Row skipping.  A row whose path step equals no step of B only subtracts: after
// j such rows in a row the columns 1..j hold dp = -j with exit value 0, the last
// column keeps its value, and further such rows change nothing (DESIGN.md
// section 4, "steady state"; checked against the full fill in
// tests/test_kernel_model.py).  So only the first m rows and the rows r..r+m
// after every row r whose NODE occurs in B have to be computed -- a few dozen
// of a 900-step path.  (Doing more rows than that would be exact too.)
constexpr int ROW_WORDS = (GFAL_MAX_STEPS + 31) / 32;

template <int MC>
__device__ __forceinline__ unsigned long long dilate_rows(uint32_t bits, int m)
{
    // every marked row r -> rows r .. r + m, by doubling (span 1 -> m + 1); m is
    // the lane's own alignment length (<= MC), so `by` is per lane
    unsigned long long v = bits;
    int span = 1;
#pragma unroll
    for (int step = 1; step <= MC; step <<= 1) {
        const int by = max(0, min(step, m + 1 - span));
        v |= v << by;
        span += by;
    }
    return v;
}

// Marks, in this lane's column of rowbits[][lane], the path positions whose node
// occurs in the alignment: walks the occurrence chains of the m nodes (first[] /
// next[] of the path image in HBM), all chains side by side.
template <int MC>
__device__ __forceinline__ void mark_match_rows(const uint16_t *__restrict__ img,
                                                const ImageLayout &L,
                                                const uint32_t (&steps)[MC], int m,
                                                uint32_t (*rowbits)[DP_THREADS], int lane)
{
    const uint16_t *first = img + L.first_at();
    const uint32_t *next = reinterpret_cast<const uint32_t *>(img + L.next_at());
    // every load below is unconditional (terminal codes index next[1022..1023],
    // which hold ENT_NONE; steps[] beyond m repeat a real step): MC independent
    // loads in flight per round, not MC exposed latencies
    uint32_t cur[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        const uint32_t head = first[j < m ? steps[j] >> 1 : 0u];   // index 0: always inside
        cur[j] = j < m ? head : ENT_NONE;
    }
    while (true) {
        uint32_t nx[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) nx[j] = next[cur[j] & ENT_POS];
        bool any = false;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            const bool on = cur[j] < ENT_FOUND;        // a position, not a terminal
            const uint32_t pos = cur[j] & ENT_POS;
            if (on) atomicOr(&rowbits[pos >> 5][lane], 1u << (pos & 31u));
            any |= on;
            cur[j] = on ? nx[j] : ENT_NONE;
        }
        if (!WAVE_ANY(any)) break;
    }
}

// One orientation of one worklist entry per lane, rows in registers, only the
// rows that can change the state.  `path_lds` holds the steps of the path most
// lanes of the wave are on (`staged` lanes); the others read theirs from HBM.
template <int MC>
__device__ __forceinline__ int traceback_score_skip(const uint16_t *__restrict__ astep,
                                                    const uint16_t *path_lds, bool staged, int n,
                                                    const uint32_t (&b)[MC], int m,
                                                    uint32_t (*rowbits)[DP_THREADS], int lane)
{
    int c[MC + 1], up_add[MC];
#pragma unroll
    for (int j = 0; j <= MC; ++j) {
        const int dp0 = (j <= n) ? -j : 0;       // :500, row 0 reaches column n only
        c[j] = dpk_pack(dp0, dp0);
    }
#pragma unroll
    for (int j = 1; j <= MC; ++j) up_add[j - 1] = dpk_up_add(j, m);
    typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
    lds_cu16 *lds_path = (lds_cu16 *)path_lds;
    const bool any_unstaged = WAVE_ANY(!staged && n > 0);
    int block = -1;                      // current 32-row block of the path
    uint32_t todo = 0;                   // its rows still to compute
    unsigned long long carry = (1ull << m) - 1ull;    // rows 1..m: the state is not steady yet
    bool done = n == 0;
    while (WAVE_ANY(!done)) {
        if (todo == 0 && !done) {
            ++block;
            if (block * 32 >= n) {
                done = true;
            } else {
                const unsigned long long rows = carry | dilate_rows<MC>(rowbits[block][lane], m);
                todo = (uint32_t)rows;
                carry = rows >> 32;
                const int left = n - block * 32;
                if (left < 32) todo &= (1u << left) - 1u;
            }
        }
        if (todo != 0) {
            const int pos = block * 32 + __builtin_ctz(todo);
            todo &= todo - 1u;
            // (an LDS read, and an HBM read for the few lanes beyond the staged paths: one
            // load of a generic pointer would be a flat instruction for every lane)
            uint32_t ai = lds_path[staged ? pos : 0];
            if (any_unstaged && !staged) ai = astep[pos];
            dp_row_regs<MC>(ai, b, up_add, c);
        }
    }
    int r = 0;
#pragma unroll
    for (int j = 1; j <= MC; ++j) r = (j == m) ? dpk_dp(c[j]) - dpk_x(c[j]) : r;
    return r;
}

// One worklist entry per lane: the orientations that were flagged.  `second`
// lanes (both flagged) run a second fill; the others idle through it with n = 0.
// fwd[j] = B[min(j, m-1)], rev[j] = B[max(m-1-j, 0)], loaded by the caller.
template <int MC>
__device__ __forceinline__ bool dp_decide_regs(const uint16_t *__restrict__ astep,
                                               const uint16_t *path_lds, bool staged, int n,
                                               const uint32_t (&fwd)[MC], const uint32_t (&rev)[MC],
                                               int m, bool has_fw, bool has_rc,
                                               uint32_t (*rowbits)[DP_THREADS], int lane)
{
    uint32_t b[MC];
    // first orientation: fw if flagged, else rc
    const bool first_rc = !has_fw;
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        const uint32_t v = first_rc ? (rev[j] ^ 1u) : fwd[j];
        b[j] = j < m ? v : STEP_INVALID;
    }
    bool good = traceback_score_skip<MC>(astep, path_lds, staged, n, b, m, rowbits, lane) == 0 && n > 0;
    const bool second = has_fw && has_rc;
    if (WAVE_ANY(second)) {
#pragma unroll
        for (int j = 0; j < MC; ++j) b[j] = j < m ? (rev[j] ^ 1u) : STEP_INVALID;
        const int n2 = second ? n : 0;
        good |= traceback_score_skip<MC>(astep, path_lds, staged, n2, b, m, rowbits, lane) == 0 && n2 > 0;
    }
    return good;
}

// Counting sort of the worklist by (length class, path): each class becomes a
// contiguous range for its own k_dp launch, and consecutive lanes share the
// path (same n, same steps: broadcast loads).  Order only affects speed; any order gives the same counters.
// hist / offsets / cursor are [N_CLASSES][n_paths].  Inside a class the bins are
// laid out thread-strided (thread t owns paths t, t + 1024, ...: coalesced
// loads); class_lo[c] is where class c starts, class_lo[N_CLASSES] the total.
__global__ __launch_bounds__(1024) void k_wl_offsets(const uint32_t *__restrict__ hist,
                                                      uint32_t *__restrict__ offsets,
                                                      uint32_t *__restrict__ cursor,
                                                      uint32_t *__restrict__ class_total,
                                                      int n_paths)
{
    // one workgroup per length class: offsets relative to the start of the class
    // and the class total; k_wl_scatter turns the totals into class_lo[]
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x, c = blockIdx.x;
    const uint32_t *h = hist + (size_t)c * n_paths;
    uint32_t sum = 0;
    for (int p = tid; p < n_paths; p += 1024) sum += h[p];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {      // Hillis-Steele inclusive scan
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (int p = tid; p < n_paths; p += 1024) {
        offsets[(size_t)c * n_paths + p] = run;
        cursor[(size_t)c * n_paths + p] = 0;
        run += h[p];
    }
    if (tid == 1023) class_total[c] = part[1023];
}

__global__ __launch_bounds__(256) void k_wl_scatter(
    Items items, const unsigned long long *__restrict__ worklist,
    const unsigned long long *__restrict__ wl_count, uint32_t wl_capacity,
    const uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
    uint32_t n_paths, unsigned long long *__restrict__ sorted,
    const uint32_t *__restrict__ class_total, uint32_t *__restrict__ class_lo,
    const uint32_t *__restrict__ wl_pos, uint32_t *__restrict__ sorted_pos)
{
    // overflow (status word set by k_scan): the histogram counted pairs that
    // were never stored, so offsets do not describe the list; the call fails
    // with GFAL_E_NOMEM and nothing downstream may touch the list
    if (*wl_count > wl_capacity) return;
    const uint32_t total = (uint32_t)*wl_count;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // where the classes start: prefix of the class totals (k_wl_offsets); the
    // first thread publishes it for the DP kernels
    uint32_t class_base[N_CLASSES + 1];
    class_base[0] = 0;
#pragma unroll
    for (int c = 0; c < N_CLASSES; ++c) class_base[c + 1] = class_base[c] + class_total[c];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int c = 0; c <= N_CLASSES; ++c) class_lo[c] = class_base[c];
    // whole waves iterate together (the tail is padded with dead lanes)
    for (uint32_t w0 = blockIdx.x * blockDim.x + (threadIdx.x & ~(WAVE - 1)); w0 < total;
         w0 += stride) {
        const uint32_t w = w0 + lane;
        const bool live = w < total;
        const unsigned long long ent = live ? worklist[w] : 0ull;
        const uint32_t p = (uint32_t)(ent >> 32) & WL_PATH_MASK;
        const uint32_t it = (uint32_t)ent >> 6;
        const uint32_t cls = (uint32_t)length_class((int)items.len[it]);
        const uint32_t bin = live ? cls * n_paths + p : 0xFFFFFFFFu;
        uint32_t my_base = 0;
#pragma unroll
        for (int c = 0; c < N_CLASSES; ++c) my_base = cls == (uint32_t)c ? class_base[c] : my_base;
        // the scan appends runs of one (path, item): one atomic per run of equal
        // bins (all of a wave's runs in flight together), not one per entry --
        // on a search batch a whole wave often lands in a single bin
        const uint32_t prev_bin = (uint32_t)__shfl_up((int)bin, 1, WAVE);
        const lanemask heads = WAVE_MASK(live && (lane == 0 || bin != prev_bin));
        const lanemask live_mask = WAVE_MASK(live);
        const lanemask upto_me = heads & ((2ull << lane) - 1ull);
        const int head = 63 - __builtin_clzll(upto_me | 1ull);          // my run's first lane
        const lanemask after = heads & ~((2ull << head) - 1ull);         // heads of later runs
        const int run_end = after ? __builtin_ctzll(after) : 64;         // one past my run
        const lanemask run = (run_end == 64 ? ~0ull : ((1ull << run_end) - 1ull)) &
                             ~((1ull << head) - 1ull) & live_mask;
        uint32_t base = 0;
        if (live && (int)lane == head)
            base = atomicAdd(&cursor[bin], (uint32_t)__builtin_popcountll(run));
        base = (uint32_t)__shfl((int)base, head, WAVE);
        const uint32_t at = live ? my_base + offsets[bin] + base + (uint32_t)(lane - head) : 0u;
        if (live) {
            sorted[at] = ent;
            if (wl_pos) sorted_pos[at] = wl_pos[w];
        }
    }
}

// Range of the sorted worklist that holds length class `cls`.
__device__ __forceinline__ void class_range(const uint32_t *class_lo, int cls, uint32_t total,
                                            uint32_t &lo, uint32_t &hi)
{
    lo = min(class_lo[cls], total);
    hi = min(class_lo[cls + 1], total);
}

struct DpArgs {
    Items items;
    const uint16_t *images;
    ImageLayout L;
    int n_paths;
    const unsigned long long *sorted;
    const uint32_t *class_lo;   // [N_CLASSES + 1] ranges of the sorted list
    const unsigned long long *wl_count;
    uint32_t wl_capacity;
    uint32_t sys_limit;         // see wavefront_class()
    uint32_t *row_scratch;
    uint32_t *counts;
    // search mode (children batches; else NULL): where an entry sits in the inverted
    // list of its path's first node, and the per-path bitmaps over that list in the
    // store that remember which entries the DP accepted (k_child inherits them)
    const uint32_t *sorted_pos;
    uint32_t *status;           // (k_dp_small: asks for the call to be run again the long way)
    uint32_t *bits;
    uint32_t bits_words;
    const int32_t *q_slot;      // image slot -> store slot or -1
};

struct DpEntry {
    uint32_t pos;
    uint32_t p;
    const uint16_t *astep, *bp;
    int n, m;
    bool has_fw, has_rc;
    uint32_t w;     // identical alignments this entry stands for (1 unless a dedup scorer)
};

__device__ __forceinline__ DpEntry load_entry(const DpArgs &a, uint32_t w, bool live)
{
    const unsigned long long ent = live ? a.sorted[w] : 0ull;
    DpEntry e;
    e.p = (uint32_t)(ent >> 32) & WL_PATH_MASK;
    const uint32_t slot = (uint32_t)ent;
    const uint32_t it = slot >> 6, ln = slot & 63u;
    const uint16_t *img = a.images + (size_t)e.p * a.L.total;
    e.astep = img + a.L.step_at();
    e.n = live ? (int)img[a.L.len_at()] : 0;
    e.m = live ? (int)a.items.len[it] : 0;
    e.bp = a.items.steps + (size_t)(live ? a.items.base[it] : 0u) * WAVE + ln;
    e.has_fw = (ent & WL_FW) != 0;
    e.has_rc = (ent & WL_RC) != 0;
    e.w = (live && a.items.weight) ? a.items.weight[slot] : 1u;
    e.pos = (live && a.sorted_pos) ? a.sorted_pos[w] : 0u;
    return e;
}

// Which kernel family runs a length class (the classes from 3 up count as one):
// the wavefront kernels (k_dp_sys) while the class holds
// few entries -- then the latency of one fill decides, and a column per lane has
// the shorter one -- the one-pair-per-lane kernels (k_dp_regs / k_dp_long) when
// it holds many.  Every kernel of both families evaluates this on the device.
__device__ __forceinline__ bool wavefront_class(const DpArgs &a, uint32_t total, int cls)
{
    const uint32_t lo = min(a.class_lo[min(cls, 3)], total);
    const uint32_t hi = cls >= 3 ? total : min(a.class_lo[cls + 1], total);
    // measured crossovers (config 3, its 1/8 shard and search-sized batches): the
    // 16-column wavefront stays ahead up to ~32 k entries, the 4- and 8-column
    // ones up to ~8 k, the whole-wave one (an entry per wave) up to ~4 k
    const uint32_t limit = cls >= 3 ? a.sys_limit / 2 : cls == 2 ? min(a.sys_limit, 0x3FFFFFFFu) * 4u
                                                                 : a.sys_limit;
    return hi - lo <= limit;
}

// Adds a wave's decisions to the per-path counters: two atomics per distinct
// path of the wave (runs of equal p; the list is sorted by path), not one per
// lane -- all resident waves work on the same few paths at a time, and per-lane
// atomics on those few words were 75 % of the DP phase.
__device__ __forceinline__ void add_results_by_path(const DpArgs &a, uint32_t p, bool live,
                                                    bool good, int lane, uint32_t w = 1u, uint32_t pos = 0u)
{
    if (a.bits != nullptr && live && good) {       // search mode: remembered for the path's children
        const int slot = a.q_slot[p];
        // (a path whose first node has a longer list than the bitmaps hold keeps none:
        // k_child marks its slot, its children recompute)
        if (slot >= 0 && (pos >> 5) < a.bits_words)
            atomicOr(&a.bits[(size_t)slot * a.bits_words + (pos >> 5)], 1u << (pos & 31u));
    }
    const uint32_t prev_p = (uint32_t)__shfl_up((int)p, 1, WAVE);
    const lanemask live_mask = WAVE_MASK(live), good_mask = WAVE_MASK(good && live);
    // a run starts at a live lane whose left neighbour is dead or on another path
    const lanemask starts =
        WAVE_MASK(live && (lane == 0 || p != prev_p || !((live_mask >> (lane - 1)) & 1ull)));
    const bool weighted = a.items.weight != nullptr;      // dedup scorer: sums of weights
    lanemask left = starts;
    while (left) {
        const int leader = __builtin_ctzll(left);
        left &= left - 1;
        const lanemask upto = left ? ((1ull << __builtin_ctzll(left)) - 1ull) : ~0ull;
        const lanemask mine = upto & ~((1ull << leader) - 1ull) & live_mask;
        uint32_t n_good, n_bad;
        if (weighted) {
            n_good = wave_weight(mine & good_mask, w, lane);
            n_bad = wave_weight(mine & ~good_mask, w, lane);
        } else {
            n_good = (uint32_t)__builtin_popcountll(mine & good_mask);
            n_bad = (uint32_t)__builtin_popcountll(mine) - n_good;
        }
        if (lane == leader) {
            if (n_good) atomicAdd(&a.counts[a.n_paths + p], n_good);
            if (n_bad) atomicAdd(&a.counts[p], n_bad);
        }
    }
}

// Length classes 0..3: rows in MC registers.
template <int MC, int CLS>
__global__ __launch_bounds__(DP_THREADS) void k_dp_regs(DpArgs a)
{
    __shared__ uint32_t rowbits[ROW_WORDS][DP_THREADS];   // per lane: rows whose node is in B
    constexpr int STAGED_PATHS = 4;
    __shared__ uint16_t path_lds[STAGED_PATHS][GFAL_MAX_STEPS + 8];
    if (*a.wl_count > a.wl_capacity) return;     // overflow: see k_wl_scatter
    const uint32_t total = (uint32_t)*a.wl_count;
    if (wavefront_class(a, total, CLS)) return;   // few entries: k_dp_sys has the lower latency
    uint32_t lo, hi;
    class_range(a.class_lo, CLS, total, lo, hi);
    const uint32_t n_threads = gridDim.x * DP_THREADS;
    const int lane = threadIdx.x;
    for (uint32_t w0 = lo + blockIdx.x * DP_THREADS; w0 < hi; w0 += n_threads) {
        const uint32_t w = w0 + threadIdx.x;
        const bool live = w < hi;
        const DpEntry e = load_entry(a, w, live);
        // the list is sorted by path: a wave is on one path, or on a few where
        // the paths change; the first STAGED_PATHS of them go to LDS (a lane on
        // a later one reads its steps from HBM, one exposed latency per row)
        const uint32_t prev_p = (uint32_t)__shfl_up((int)e.p, 1, WAVE);
        const lanemask starts = WAVE_MASK(live && (lane == 0 || e.p != prev_p));
        const int slot = __builtin_popcountll(starts & ((2ull << lane) - 1ull)) - 1;
        __syncthreads();
        {
            lanemask left = starts;
            for (int k = 0; k < STAGED_PATHS && left; ++k) {
                const int leader = __builtin_ctzll(left);
                left &= left - 1;
                const uint32_t p_k = (uint32_t)__builtin_amdgcn_readlane((int)e.p, leader);
                const int n_k = __builtin_amdgcn_readlane(e.n, leader);
                const uint16_t *src = a.images + (size_t)p_k * a.L.total + a.L.step_at();
                // all loads of a path in flight before the first store
                constexpr int LOADS = (GFAL_MAX_STEPS + DP_THREADS - 1) / DP_THREADS;
                uint16_t t[LOADS];
#pragma unroll
                for (int q = 0; q < LOADS; ++q)
                    t[q] = src[min(lane + q * DP_THREADS, max(n_k - 1, 0))];
#pragma unroll
                for (int q = 0; q < LOADS; ++q)
                    if (lane + q * DP_THREADS < n_k) path_lds[k][lane + q * DP_THREADS] = t[q];
            }
        }
        // the alignment, forward and reversed, with unconditional loads (2 * MC in
        // flight; a load per `if (j < m)` would expose 2 * MC latencies)
        uint32_t fwd[MC], rev[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            fwd[j] = (uint32_t)e.bp[max(min(j, e.m - 1), 0) * WAVE];
            rev[j] = (uint32_t)e.bp[max(e.m - 1 - j, 0) * WAVE];
        }
        int n_max = e.n;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) n_max = max(n_max, __shfl_xor(n_max, o, WAVE));
        const int n_words = (__builtin_amdgcn_readfirstlane(n_max) + 31) >> 5;
        for (int k = 0; k < n_words; ++k) rowbits[k][lane] = 0;
        __syncthreads();
        const bool staged = live && slot < STAGED_PATHS;
        const uint16_t *my_path = path_lds[staged ? slot : 0];
        mark_match_rows<MC>(a.images + (size_t)e.p * a.L.total, a.L, fwd, e.m, rowbits, lane);
        bool good = dp_decide_regs<MC>(e.astep, my_path, staged, e.n, fwd, rev, e.m, e.has_fw,
                                       e.has_rc, rowbits, lane);
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 3
        good = false;
#endif
        add_results_by_path(a, e.p, live, good, lane, e.w, e.pos);
    }
}

// Short worklists (the batches a search submits: a few thousand pairs) are bound
// by the latency of the longest single fill, not by throughput: a 900-step path
// against a 16-step alignment is 14 400 dependent cells for one lane of
// k_dp_regs.  k_dp_sys turns the table sideways: one column per lane, the rows
// stream through as a wavefront.  At step s lane c fills cell (i = s - c,
// j = c + 1); the cell to its left (i, j-1) is its left neighbour's result of
// step s-1, fetched with one DPP shift (zero shifted in = column 0, which the
// reference never writes), the diagonal one (i-1, j-1) is what it fetched the
// step before.  A fill takes n + m steps of ~15 instructions instead of n * m
// cells of 12; throughput is lower (idle columns), so long lists keep
// k_dp_regs.  MC lanes per pair: 4, 8, 16 (shifts inside DPP rows) or 64.
constexpr int SYS_MAX_M = 64;

template <int MC>
__device__ __forceinline__ int from_left_column(int v, bool first_column)
{
    // row_shr:1 within 16-lane rows / wave_shr:1; lanes without a source read 0,
    // and so must the first lane of a group that starts inside a row
    const int r = MC <= 16 ? __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true)
                           : __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
    return (MC < 16 && first_column) ? 0 : r;
}

// One wavefront step of k_dp_sys for this lane's column j: (d_dp, d_x) is the
// diagonal cell; the left one is fetched into (r_dp, r_x) and is the diagonal
// of the next step.  (A lone wave issues one instruction every ~4.5 cycles
// whatever the dependencies: running both orientations in one lane doubled the
// step time, so they go to different lane groups instead.)
template <int MC>
__device__ __forceinline__ void sys_step(int s, int j, int n, uint32_t b, int up_add,
                                         const uint16_t *arow, uint32_t &a_next, int &l, int d, int &r)
{
    // cells are the packed words of dp_row_regs: one shift from the left lane, one max3
    r = from_left_column<MC>(l, j == 1);
    const uint32_t ai = a_next;
    a_next = arow[s + 1];
    // selects, not a branch: the exec-mask round trip through the scalar unit
    // costs a lone wave more than the two extra instructions
    const bool active = (unsigned)(s - j) < (unsigned)n;     // row i = s - c is in 1..n
    const int dd = d + ((ai == b) ? 3 * DPK_RANK : 3 * DPK_RANK - DPK_ONE);      // :527
    const int u = l + up_add;                                                    // :534 (:504 free in the last column)
    const int v = max(dd, max(u, r + (DPK_RANK - DPK_ONE))) & ~(3 * DPK_RANK);   // :541
    l = active ? v : l;
}

// LDS of one workgroup of the wavefront DP
template <int MC>
struct DpSysLds {
    static constexpr int ENTRIES = MC == 64 ? 1 : WAVE / MC / 2;
    static constexpr int ROWLEN = GFAL_MAX_STEPS + 2 * MC + 8;      // (reads of idle lanes stay inside)
    static constexpr int BYTES = ENTRIES * (2 * ROWLEN * 2 + 2 * (ROW_WORDS + 2) * 4);
};

// block `block` of `n_blocks` of the wavefront DP over its length class.  whole (k_dp_small):
// the list is short and NOT sorted -- every entry of it with min_m <= m <= MC is this
// function's, wherever it sits
template <int MC>
__device__ __forceinline__ void dp_sys_body(const DpArgs &a, uint32_t block, uint32_t n_blocks,
                                            unsigned char *smem, bool whole, int min_m)
{
    static_assert(MC == 4 || MC == 8 || MC == 16 || MC == 64, "group inside a DPP row, or the wave");
    // MC < 64: an entry takes two neighbouring groups, one per orientation;
    // MC = 64: one entry per wave, the orientations one after the other
    constexpr int ENTRIES = DpSysLds<MC>::ENTRIES;
    constexpr int ROWLEN = DpSysLds<MC>::ROWLEN;
    constexpr int PAD = MC;
    // per entry (its two orientation groups write the same values)
    uint16_t (*apath)[ROWLEN] = reinterpret_cast<uint16_t (*)[ROWLEN]>(smem);          // the entries' paths
    uint16_t (*alist)[ROWLEN] = apath + ENTRIES;                  // their steps at the rows to compute
    uint32_t (*rowmark)[ROW_WORDS + 2] =
        reinterpret_cast<uint32_t (*)[ROW_WORDS + 2]>(alist + ENTRIES);                // rows whose node occurs in B
    uint32_t (*rowsel)[ROW_WORDS + 2] = rowmark + ENTRIES;        // rows to compute
    if (*a.wl_count > a.wl_capacity) return;              // overflow: see k_wl_scatter
    const uint32_t total = (uint32_t)*a.wl_count;
    // MC = 4, 8, 16: classes 0, 1, 2; MC = 64: class 3 and the entries of the last
    // class that fit (k_dp_long skips those on a short list)
    constexpr int CLS = MC == 4 ? 0 : MC == 8 ? 1 : MC == 16 ? 2 : 3;
    if (!whole && !wavefront_class(a, total, CLS)) return;        // many entries: k_dp_regs / k_dp_long
    const uint32_t lo = whole ? 0u : min(a.class_lo[CLS], total);
    const uint32_t hi = (whole || MC == 64) ? total : min(a.class_lo[CLS + 1], total);
    const int lane = threadIdx.x;
    const int c = lane % MC, g = lane / MC;
    const int en = MC == 64 ? 0 : g / 2;                  // my entry of the wave
    const int j = c + 1;
    for (uint32_t w0 = lo + block * ENTRIES; w0 < hi; w0 += n_blocks * ENTRIES) {
        const uint32_t w = w0 + en;
        const DpEntry e = load_entry(a, w, w < hi);
        const bool live = w < hi && e.m <= MC && e.m >= min_m;
        // stage the groups' paths with the whole wave, all loads of a path in
        // flight at once; neighbouring groups usually share the path (the two
        // orientations of an entry always do) and then share the copy
        __syncthreads();
        int my_row = en;
        {
            constexpr int SPAN_LANES = MC == 64 ? 64 : 2 * MC;    // lanes of one entry
            uint32_t prev_p = 0xFFFFFFFFu;
            int prev_row = 0;
            for (int ee = 0; ee < ENTRIES; ++ee) {
                const uint32_t p_ee = (uint32_t)__builtin_amdgcn_readlane((int)e.p, ee * SPAN_LANES);
                const int n_ee = __builtin_amdgcn_readlane(live ? e.n : 0, ee * SPAN_LANES);
                int row = ee;
                if (n_ee > 0) {
                    if (p_ee == prev_p) {
                        row = prev_row;
                    } else {
                        const uint16_t *src = a.images + (size_t)p_ee * a.L.total + a.L.step_at();
                        constexpr int LOADS = (GFAL_MAX_STEPS + WAVE - 1) / WAVE;
                        uint16_t t[LOADS];
#pragma unroll
                        for (int k = 0; k < LOADS; ++k)
                            t[k] = (lane + k * WAVE < n_ee) ? src[lane + k * WAVE] : (uint16_t)0;
#pragma unroll
                        for (int k = 0; k < LOADS; ++k)
                            if (lane + k * WAVE < n_ee) apath[ee][PAD + lane + k * WAVE] = t[k];
                        prev_p = p_ee;
                        prev_row = ee;
                    }
                }
                if (en == ee) my_row = row;
            }
        }
        __syncthreads();
        // Row skipping as in traceback_score_skip: only the first m rows and the
        // rows r .. r + m after every row r whose node occurs in B change the
        // state; the group compacts the path steps of those rows into alist[] and
        // the wavefront runs over that list (a 900-step path and a 10-step
        // alignment: ~150 steps instead of 900).  Windows of more than 32 rows
        // (m > 31) do not fit the 64-bit dilation: those entries take every row.
        const int m = e.m;
        const bool skipping = m <= 31;
        for (int w = c; w < ROW_WORDS + 2; w += MC) {
            rowmark[en][w] = 0;
            rowsel[en][w] = 0;
        }
        __syncthreads();
        {   // mark: lane c walks the occurrence chain of B[c]'s node
            const uint16_t *img = a.images + (size_t)e.p * a.L.total;
            const uint16_t *first = img + a.L.first_at();
            const uint32_t *next = reinterpret_cast<const uint32_t *>(img + a.L.next_at());
            uint32_t cur = ENT_NONE;
            if (live && skipping && c < m) cur = first[(uint32_t)e.bp[c * WAVE] >> 1];
            while (WAVE_ANY(cur < ENT_FOUND)) {
                const bool on = cur < ENT_FOUND;
                const uint32_t pos = cur & ENT_POS;
                const uint32_t nx = next[pos];      // always inside next[]
                if (on) atomicOr(&rowmark[en][pos >> 5], 1u << (pos & 31u));
                cur = on ? nx : ENT_NONE;
            }
        }
        __syncthreads();
        if (live && skipping) {
            for (int w = c; w < ROW_WORDS; w += MC) {
                const unsigned long long d = dilate_rows<32>(rowmark[en][w], m);
                if ((uint32_t)d) atomicOr(&rowsel[en][w], (uint32_t)d);
                if (d >> 32) atomicOr(&rowsel[en][w + 1], (uint32_t)(d >> 32));
            }
            if (c == 0) atomicOr(&rowsel[en][0], (1u << m) - 1u);     // rows 1..m
        }
        __syncthreads();
        int n_c = 0;                                      // rows to compute (group-uniform)
        {
            const int n_full = live ? e.n : 0;
            for (int w0r = 0; w0r < ROW_WORDS; w0r += MC) {
                const int w = w0r + c;
                uint32_t bits = 0;
                if (w < ROW_WORDS) {
                    bits = skipping ? rowsel[en][w] : 0xFFFFFFFFu;
                    const int left = n_full - w * 32;
                    bits = left <= 0 ? 0u : left >= 32 ? bits : (bits & ((1u << left) - 1u));
                }
                const int cnt = __builtin_popcount(bits);
                int incl = cnt;                           // inclusive prefix inside the group
#pragma unroll
                for (int o = 1; o < MC; o <<= 1) {
                    const int v = __shfl_up(incl, o, MC);
                    if (c >= o) incl += v;
                }
                int off = n_c + incl - cnt;
                n_c += __shfl(incl, MC - 1, MC);
                while (bits) {
                    const int bpos = __builtin_ctz(bits);
                    bits &= bits - 1u;
                    alist[en][PAD + off++] = apath[my_row][PAD + w * 32 + bpos];
                }
            }
        }
        __syncthreads();
        const uint16_t *arow = alist[en] + PAD - 1 - c;        // arow[s] = step of listed row s - c

        bool good = false;
        for (int pass = 0; pass < (MC == 64 ? 2 : 1); ++pass) {
            const bool flip = MC == 64 ? pass == 1 : (g & 1) != 0;
            // orientations the scan did not flag cannot be free (DESIGN.md
            // section 2): their group idles
            const bool run = live && (flip ? e.has_rc : e.has_fw);
            if (!WAVE_ANY(run)) continue;
            const int n = run ? n_c : 0;                  // the listed rows stand for the path
            uint32_t b = STEP_INVALID;
            if (run && c < m)
                b = flip ? ((uint32_t)e.bp[(m - 1 - c) * WAVE] ^ 1u) : (uint32_t)e.bp[c * WAVE];
            const int up_add = dpk_up_add(j, m);          // :504 free in the last column
            // row 0 of my column (:500; m <= n here, so every column is inside)
            int l = dpk_pack(-j, -j);                     // my latest cell (i - 1, j)
            int g = 0;                                    // the diagonal one (i - 1, j - 1): dp 0, exit value 0
            int n_max = n;
#pragma unroll
            for (int o = MC; o < WAVE; o <<= 1) n_max = max(n_max, __shfl_xor(n_max, o, WAVE));
            n_max = __builtin_amdgcn_readfirstlane(n_max);
            // two steps per trip: what was fetched from the left in one step is
            // the diagonal of the next, so the roles of (g, h) alternate; the
            // path step of the next row is loaded one step ahead
            uint32_t a_next = arow[1];
            for (int s = 1; s < n_max + MC; s += 2) {
                int h;
                sys_step<MC>(s, j, n, b, up_add, arow, a_next, l, g, h);
                sys_step<MC>(s + 1, j, n, b, up_add, arow, a_next, l, h, g);
            }
            // lane m-1 of the group holds cell (n, m)
            good |= run && n > 0 && c == m - 1 && dpk_dp(l) == dpk_x(l);
        }
        // the entry's lanes: both of its groups
        constexpr int SPAN = MC == 64 ? 64 : 2 * MC;
        const lanemask gm = WAVE_MASK(good);
        const lanemask mine = (gm >> (lane / SPAN * SPAN)) & (SPAN == 64 ? ~0ull : ((1ull << SPAN) - 1ull));
        // one result per entry, carried by the first lane of its span
        add_results_by_path(a, e.p, live && lane % SPAN == 0, mine != 0, lane, e.w, e.pos);
    }
}

template <int MC>
__global__ __launch_bounds__(DP_THREADS) void k_dp_sys(DpArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[DpSysLds<MC>::BYTES];
    dp_sys_body<MC>(a, blockIdx.x, gridDim.x, smem, false, 0);
}

// The search's batches leave the exact DP a few hundred to a few thousand pairs in all:
// one launch for every length class (grid.y) instead of a fork over four streams with a
// wavefront and a register kernel each, most of which find an empty or short list --
// and straight from the list as k_child pushed it: every class walks the whole list and
// takes its own entries, so the counting sort (two more launches) is not needed.
// The host picks it when the previous call's list was short (and no alignment is
// longer than a wave); a longer list than expected is still decided exactly, only
// more slowly than the sort and the register kernels would.
__global__ __launch_bounds__(DP_THREADS) void k_dp_small(DpArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[DpSysLds<4>::BYTES];      // the largest of the four
    static_assert(DpSysLds<4>::BYTES >= DpSysLds<8>::BYTES && DpSysLds<4>::BYTES >= DpSysLds<16>::BYTES &&
                      DpSysLds<4>::BYTES >= DpSysLds<64>::BYTES, "LDS of k_dp_small");
    // the list turned out long after all (every class would walk all of it): report it
    // like a list that did not fit -- the blocking call runs the batch again, and with
    // this call's count on record it takes the sort and the register kernels
    if (*a.wl_count > 32768ull && *a.wl_count <= a.wl_capacity) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicOr(a.status, ST_DP_OVERFLOW);
        return;
    }
    switch (blockIdx.y) {
    case 0: dp_sys_body<4>(a, blockIdx.x, gridDim.x, smem, true, 1); break;
    case 1: dp_sys_body<8>(a, blockIdx.x, gridDim.x, smem, true, 5); break;
    case 2: dp_sys_body<16>(a, blockIdx.x, gridDim.x, smem, true, 9); break;
    default: dp_sys_body<64>(a, blockIdx.x, gridDim.x, smem, true, 17); break;
    }
}

// Last length class (more than 32 steps): rows in LDS or HBM.
template <bool ROWS_IN_LDS>
__global__ __launch_bounds__(DP_THREADS) void k_dp_long(DpArgs a)
{
    int stride;
    uint32_t *row = dp_row<ROWS_IN_LDS>(a.row_scratch, stride);
    if (*a.wl_count > a.wl_capacity) return;     // overflow: see k_wl_scatter
    const uint32_t total = (uint32_t)*a.wl_count;
    uint32_t lo, hi;
    class_range(a.class_lo, LONG_CLASS, total, lo, hi);
    const uint32_t n_threads = gridDim.x * DP_THREADS;
    for (uint32_t w0 = lo + blockIdx.x * DP_THREADS; w0 < hi; w0 += n_threads) {
        const uint32_t w = w0 + threadIdx.x;
        const bool live = w < hi;
        const DpEntry e = load_entry(a, w, live);
        // on a short list k_dp_sys<64> takes the entries of up to 64 steps
        const bool mine = live && !(wavefront_class(a, total, 3) && e.m <= SYS_MAX_M);
        // src/eval.cpp:92-98; both fills always run so the lanes of the wave
        // stay in step through the row loops (a dead lane has n = m = 0)
        StepsA A{e.astep, mine ? e.n : 0};
        StepsB B{e.bp, mine ? e.m : 0, 0u};
        const int fw = traceback_score(A, B, row, stride);
        B.flip = 1u;
        const int rc = traceback_score(A, B, row, stride);
        const bool good = fw == 0 || rc == 0;
        add_results_by_path(a, e.p, mine, good, (int)threadIdx.x, e.w, e.pos);
    }
}

template <bool ROWS_IN_LDS>
__global__ __launch_bounds__(DP_THREADS) void k_pairs(
    Items items, const int32_t *__restrict__ slot_orig,
    const uint16_t *__restrict__ image, ImageLayout L,
    uint32_t *__restrict__ row_scratch, int32_t *__restrict__ fw,
    int32_t *__restrict__ rc)
{
    int stride;
    uint32_t *row = dp_row<ROWS_IN_LDS>(row_scratch, stride);
    const long long n_threads = (long long)gridDim.x * DP_THREADS;
    const long long gtid = (long long)blockIdx.x * DP_THREADS + threadIdx.x;
    const long long n_slots = (long long)items.n_items * WAVE;
    StepsA A{image + L.step_at(), (int)image[L.len_at()]};
    for (long long s = gtid; s < n_slots; s += n_threads) {
        const int32_t orig = slot_orig[s];
        if (orig < 0) continue;
        const uint32_t it = (uint32_t)(s >> 6), ln = (uint32_t)(s & 63);
        StepsB B{items.steps + (size_t)items.base[it] * WAVE + ln,
                 (int)items.len[it], 0u};
        fw[orig] = traceback_score(A, B, row, stride);
        B.flip = 1u;
        rc[orig] = traceback_score(A, B, row, stride);
    }
}

__global__ void k_fill_i32(int32_t *p, long long n, int32_t v)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// --------------------------------------------------------------------------
// Search mode: a candidate scored from its parent (gfal_group_score_children).
//
// `gfalign search` only ever scores `parent + one step` (reference
// src/eval.cpp:146-162), and with the filter on (src/eval.cpp:81-91) the
// counters of src/eval.cpp:92-98 split into
//     Pass(A)  alignments all of whose nodes are on A                 (bad + good)
//     G1(A)    of those: zero-step ones and contiguous subpaths of A, either
//              strand                                     (good without the DP)
//     G2(A)    of the rest: start-overhang pairs the exact DP accepts
// For A' = A + [s] with len(A) >= the longest alignment (so "m > n" never holds):
//     Pass(A') = Pass(A) + #{B : node(s) in B, nodes(B) on A'}   if node(s) is new
//     G1(A')   = G1(A) + sum over M in (Lmax, Mmax] of mult(W_M) + mult(rc(W_M))
//                (once if W_M is its own reverse complement); W_M = the last M steps
//                of A', Lmax = the longest W_M that already occurs in A, either
//                strand, mult = how many alignments have exactly that content
//     G2(A')   recomputed: its candidates all contain the node of A'[0]
// (tests/incr_model.py fuzzes these identities against the plain rule).  So a
// child costs a walk over the alignments that contain its new node and its first
// node -- two inverted lists -- and at most Mmax table lookups, instead of a scan
// over all alignments.  The exact DP and everything around it (k_prep images,
// worklist sort, k_dp_*) are the batch pipeline's own.
//
// Scored paths live in a device-side store (caller-managed slots: steps, Pass,
// G1); a batch names every child as (parent, step), the parent being a slot or
// an earlier child of the same batch.
// --------------------------------------------------------------------------
namespace {
constexpr int CHILD_MAX_DEPTH = 64;              // in-batch ancestors of one child
constexpr int CHILD_THREADS = 256;
constexpr uint32_t CT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t ST_BAD_CHILD = 8u;            // status flag: bad parent reference / slot / length

struct ChildIndex {              // built once per scorer, on the device
    const uint32_t *inv_off;     // [n_local * N_CLASSES + 1] (node, DP length class) -> range of inv_slot
    // per list entry: {lane (item * 64 + lane), m | first step << 16, steps 1..2, steps 3..4} of an
    // alignment that has the node: alignments of up to five steps need no further load
    const uint4 *inv_ent;
    const uint32_t *ct_key;      // content table: representative lane or CT_EMPTY
    const uint32_t *ct_hash;     // its whash
    const uint32_t *ct_mult;     // alignments with exactly that content (weights summed)
    uint32_t ct_mask;
};

__device__ __forceinline__ bool lane_same_content(const Items &items, uint32_t x, uint32_t y)
{
    const uint32_t ix = x >> 6, iy = y >> 6;
    const int m = items.len[ix];
    if (m != (int)items.len[iy]) return false;
    const uint16_t *px = items.steps + (size_t)items.base[ix] * WAVE + (x & 63u);
    const uint16_t *py = items.steps + (size_t)items.base[iy] * WAVE + (y & 63u);
    for (int t = 0; t < m; ++t)
        if (px[(size_t)t * WAVE] != py[(size_t)t * WAVE]) return false;
    return true;
}

// one thread per lane: distinct nodes of its alignment -> cnt[node]++ (fill == NULL)
// or inv_slot[cursor[node]++] = lane
__global__ void k_inv_build(Items items, const int32_t *__restrict__ slot_orig, uint32_t n_slots,
                            uint32_t *__restrict__ cnt_or_cursor, uint4 *__restrict__ fill)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots || slot_orig[slot] < 0) return;
    const uint32_t it = slot >> 6;
    const int m = items.len[it];
    const uint16_t *bp = items.steps + (size_t)items.base[it] * WAVE + (slot & 63u);
    for (int t = 0; t < m; ++t) {
        const uint32_t node = (uint32_t)bp[(size_t)t * WAVE] >> 1;
        bool seen = false;
        for (int u = 0; u < t && !seen; ++u) seen = ((uint32_t)bp[(size_t)u * WAVE] >> 1) == node;
        if (seen) continue;
        const uint32_t at = atomicAdd(&cnt_or_cursor[node * N_CLASSES + (uint32_t)length_class(m)], 1u);
        if (fill) {
            auto st = [&](int k) -> uint32_t { return k < m ? (uint32_t)bp[(size_t)k * WAVE] : 0u; };
            fill[at] = make_uint4(slot, (uint32_t)m | (st(0) << 16), st(1) | (st(2) << 16), st(3) | (st(4) << 16));
        }
    }
}

// exclusive scan of cnt[0 .. n) -> off[0 .. n], cursor = off (one workgroup)
__global__ __launch_bounds__(1024) void k_inv_scan(const uint32_t *__restrict__ cnt, int n,
                                                   uint32_t *__restrict__ off, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x, per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        cursor[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) off[n] = part[1023];
}

// one thread per lane: enter its alignment into the content table
// rec[idx] = {representative lane, m | step 0 << 16, steps 1..2, steps 3..4}: what a lookup
// compares in one 16-byte load (k_tile); idx, the identity of the lane's content, goes into
// the lane's item record (k_scan3)
constexpr int PAIR_BITS_LOG2 = 22;            // 4 M bits = 512 KB (L2-resident)
__device__ __forceinline__ uint32_t pair_bit_index(uint32_t x, uint32_t y)
{
    return (((x << 16) | (y & 0xFFFFu)) * 0x9E3779B1u) >> (32 - PAIR_BITS_LOG2);
}
__device__ __forceinline__ bool pair_present(const uint32_t *bits, uint32_t x, uint32_t y)
{
    const uint32_t i = pair_bit_index(x, y);
    return (bits[i >> 5] >> (i & 31u)) & 1u;
}

__global__ void k_ct_build(Items items, const int32_t *__restrict__ slot_orig, uint32_t n_slots,
                           const uint32_t *__restrict__ item_hash, uint32_t *__restrict__ key,
                           uint32_t *__restrict__ hash, uint32_t *__restrict__ mult, uint32_t mask,
                           uint4 *__restrict__ rec, uint32_t *__restrict__ rec3,
                           const uint32_t *__restrict__ item_r3, uint32_t *__restrict__ pair_bits)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots || slot_orig[slot] < 0) return;
    {   // the lane's pairs of consecutive steps, both strands (ContentTable::pair_bits)
        const uint32_t it = slot >> 6;
        const int m = items.len[it];
        const uint16_t *bp = items.steps + (size_t)items.base[it] * WAVE + (slot & 63u);
        uint32_t x = m > 0 ? (uint32_t)bp[0] : 0u;
        for (int k = 1; k < m; ++k) {
            const uint32_t y = bp[(size_t)k * WAVE];
            const uint32_t i1 = pair_bit_index(x, y), i2 = pair_bit_index(y ^ 1u, x ^ 1u);
            atomicOr(&pair_bits[i1 >> 5], 1u << (i1 & 31u));
            atomicOr(&pair_bits[i2 >> 5], 1u << (i2 & 31u));
            x = y;
        }
    }
    // the lane's key: first dword of its item record (k_scan3)
    uint32_t *my_key = rec3 + (size_t)item_r3[slot >> 6] * WAVE + (slot & 63u);
    const uint32_t h = item_hash[slot];
    const uint32_t w = items.weight ? items.weight[slot] : 1u;
    uint32_t idx = h & mask;
    while (true) {
        const uint32_t prev = atomicCAS(&key[idx], CT_EMPTY, slot);
        if (prev == CT_EMPTY) {
            hash[idx] = h;
            atomicAdd(&mult[idx], w);
            const uint32_t it = slot >> 6;
            const int m = items.len[it];
            const uint16_t *bp = items.steps + (size_t)items.base[it] * WAVE + (slot & 63u);
            auto st = [&](int k) -> uint32_t { return k < m ? (uint32_t)bp[(size_t)k * WAVE] : 0u; };
            rec[idx] = make_uint4(slot, (uint32_t)m | (st(0) << 16), st(1) | (st(2) << 16), st(3) | (st(4) << 16));
            *my_key = idx;
            return;
        }
        if (item_hash[prev] == h && lane_same_content(items, prev, slot)) {
            atomicAdd(&mult[idx], w);
            *my_key = idx;
            return;
        }
        idx = (idx + 1u) & mask;
    }
}

struct ChildBatch {
    const int32_t *parent;   // [n] >= 0: store slot; < 0: ~index of an earlier child of the batch
    const int32_t *step;     // [n] the appended step (caller's packed code)
    const int32_t *slot;     // [n] where to keep the child, or -1
    int n;
    int32_t *st_steps;       // the store
    int32_t *st_len;
    uint32_t *st_pass, *st_g1;
    int64_t st_cap;
    int32_t *root, *depth;   // [n] scratch: stored ancestor, steps beyond it
    uint32_t *dpass, *dg1;   // [n] scratch: the child's own deltas
    uint32_t *st_bits;       // [cap][bits_words] entries of the first node's list the DP accepted
    int32_t *st_bitsok;      // [cap] the bitmap is valid (NULL / 0 words: nothing is remembered)
    uint32_t bits_words;
};

// lengths and offsets of the batch's paths (one workgroup)
__global__ __launch_bounds__(1024) void k_child_len(ChildBatch b, int max_len, int min_parent_len,
                                                    int32_t *__restrict__ path_off,
                                                    uint32_t *__restrict__ status,
                                                    const int32_t *__restrict__ host_in, int32_t *__restrict__ dev_in)
{
    __shared__ uint32_t part[1024];
    // a children batch is not sorted by length (k_child's cost does not follow it): this
    // kernel also does what k_len_sort_block does besides sorting, clearing the status words
    if (threadIdx.x < 8) status[threadIdx.x] = 0;
    // the batch itself ([parent | step | slot], a few KB) is read straight from the
    // caller's pinned staging buffer: one copy operation less in front of every call
    if (host_in)
        for (int i = threadIdx.x; i < 3 * b.n; i += 1024) dev_in[i] = host_in[i];
    __syncthreads();
    const int tid = threadIdx.x, per = (b.n + 1023) / 1024;
    const int lo = min(tid * per, b.n), hi = min(lo + per, b.n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) {
        int j = i, d = 1;
        bool good = true;
        while (b.parent[j] < 0) {
            const int up = ~b.parent[j];
            if (up >= j || d >= CHILD_MAX_DEPTH) {
                good = false;
                break;
            }
            j = up;
            ++d;
        }
        int r = good ? b.parent[j] : 0;
        if (r < 0 || r >= b.st_cap) {
            good = false;
            r = 0;
        }
        int L = 1;
        if (good) {
            const int l0 = b.st_len[r];
            L = l0 + d;
            if (l0 < 1 || l0 < min_parent_len || L > max_len || L > GFAL_MAX_STEPS) good = false;
        }
        if (b.slot[i] >= b.st_cap) good = false;
        if (!good) {       // root = -1 marks it for the kernels behind
            atomicOr(status, ST_BAD_CHILD);
            L = 1;
            d = 1;
        }
        b.root[i] = good ? r : -1;
        b.depth[i] = d;
        b.dpass[i] = 0;
        b.dg1[i] = 0;
        path_off[i + 1] = L;
        sum += (uint32_t)L;
    }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    if (tid == 0) path_off[0] = 0;
    for (int i = lo; i < hi; ++i) {
        run += (uint32_t)path_off[i + 1];
        path_off[i + 1] = (int32_t)run;
    }
}

struct ChildArgs {
    Items items;
    ChildIndex ix;
    const uint16_t *images;
    ImageLayout L;
    const uint16_t *lids;      // [n_paths][nm] local node id per path step (k_prep)
    const int32_t *order;      // image slot -> child index
    int n_paths, max_aln_len;
    uint32_t *dpass, *dg1;     // by child index
    ChildBatch b;              // roots, depths, slots, the store's bitmaps
    uint32_t *wl_pos;          // list position of every pushed entry (NULL: nothing is remembered)
    uint32_t *counts;          // by image slot: bad | good | unaligned
    unsigned long long *worklist;
    unsigned long long *wl_count;
    uint32_t wl_capacity;
    uint32_t *wl_hist;
    uint32_t *status;
};

// multiplicity of the content  W[t] = step[start + dir * t] ^ flip,  t < M
__device__ __forceinline__ uint32_t ct_lookup(const ChildArgs &a, uint32_t h, int M, const uint16_t *step,
                                              int start, int dir, uint32_t flip)
{
    uint32_t idx = h & a.ix.ct_mask;
    while (true) {
        const uint32_t key = a.ix.ct_key[idx];
        if (key == CT_EMPTY) return 0u;
        if (a.ix.ct_hash[idx] == h) {
            const uint32_t it = key >> 6;
            if ((int)a.items.len[it] == M) {
                const uint16_t *bp = a.items.steps + (size_t)a.items.base[it] * WAVE + (key & 63u);
                bool eq = true;          // (no early exit: the loads go out together)
#pragma unroll 8
                for (int t = 0; t < M; ++t)
                    eq &= (uint32_t)bp[(size_t)t * WAVE] == ((uint32_t)step[start + dir * t] ^ flip);
                if (eq) return a.ix.ct_mult[idx];
            }
        }
        idx = (idx + 1u) & a.ix.ct_mask;
    }
}

// grid (children, chunks): chunk 0 finds the child's new windows; every chunk takes
// its share of the two inverted lists.  A lane holds one alignment of a list at a
// time: its first step in a register, the others (up to 33 steps) as pair dwords in
// the lane's own LDS column, so the divergent per-lane loops below read LDS, not HBM.
constexpr int CHILD_WAVES = CHILD_THREADS / WAVE;
constexpr int CHILD_STAGE_PAIRS = 8;
constexpr int CHILD_WL_BUF = 512;              // worklist entries a workgroup gathers before it appends them
constexpr int CHILD_STATIC_LDS = CHILD_WAVES * CHILD_STAGE_PAIRS * WAVE * 4 + CHILD_WL_BUF * 12 + 256;

template <bool W>
__global__ __launch_bounds__(CHILD_THREADS) void k_child(ChildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    __shared__ uint32_t red[4];     // Lmax | node seen before
    __shared__ uint32_t stage[CHILD_WAVES][CHILD_STAGE_PAIRS][WAVE];
    __shared__ unsigned long long wl_buf[CHILD_WL_BUF];
    __shared__ uint32_t wl_at[CHILD_WL_BUF];
    __shared__ uint32_t wl_n, wl_cls[N_CLASSES], wl_base[2];
    uint16_t *img = smem;
    uint16_t *lid = smem + a.L.total;
    const int q = blockIdx.x, chunk = blockIdx.y, n_chunks = gridDim.y;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.images + (size_t)q * a.L.total);
        uint32_t *dst = reinterpret_cast<uint32_t *>(img);
        for (int i = tid; i < a.L.total / 2; i += CHILD_THREADS) dst[i] = src[i];
        const uint32_t *lsrc = reinterpret_cast<const uint32_t *>(a.lids + (size_t)q * a.L.nm);
        uint32_t *ldst = reinterpret_cast<uint32_t *>(lid);
        for (int i = tid; i < a.L.nm / 2; i += CHILD_THREADS) ldst[i] = lsrc[i];
        if (tid < 4) red[tid] = 0;
        if (tid < N_CLASSES) wl_cls[tid] = 0;
        if (tid == 0) wl_n = 0;
    }
    __syncthreads();
    const uint32_t *first32 = reinterpret_cast<const uint32_t *>(img + a.L.first_at());
    const uint32_t *next = reinterpret_cast<const uint32_t *>(img + a.L.next_at());
    const uint32_t *step32 = reinterpret_cast<const uint32_t *>(img + a.L.step_at());
    const uint32_t *lid32 = reinterpret_cast<const uint32_t *>(lid);
    const int n = img[a.L.len_at()];
    if (n < 2) return;                       // (rejected child: nothing to add)
    const int child = a.order ? a.order[q] : q;
    const uint32_t s = lds_u16(step32, (uint32_t)(n - 1)), s_lid = lds_u16(lid32, (uint32_t)(n - 1));
    const uint32_t a0 = lds_u16(step32, 0u), a0_lid = lds_u16(lid32, 0u);
    const int cap = min(a.max_aln_len, n);   // longest window that can be an alignment
    auto stepA = [&](int k) -> uint32_t { return lds_u16(step32, (uint32_t)k); };
    auto on_path = [&](uint32_t code) -> bool { return lds_u16(first32, code >> 1) != ENT_NONE; };

    // has the new node been on the path before; how long a suffix window is old
    {
        uint32_t my_old = 0, my_l = 0;
        for (int p = tid; p < n - 1; p += CHILD_THREADS) {
            if (lds_u16(lid32, (uint32_t)p) == s_lid) my_old = 1;
            if (chunk == 0 && s != STEP_NOMATCH) {
                const uint32_t c = stepA(p);
                if (c == s) {
                    int l = 1;
                    while (l < cap && p - l >= 0 && stepA(p - l) == stepA(n - 1 - l)) ++l;
                    my_l = max(my_l, (uint32_t)l);
                } else if (c == (s ^ 1u)) {
                    int l = 1;
                    while (l < cap && p + l < n - 1 && stepA(p + l) == (stepA(n - 1 - l) ^ 1u)) ++l;
                    my_l = max(my_l, (uint32_t)l);
                }
            }
        }
        if (my_l) atomicMax(&red[0], my_l);
        if (my_old) atomicOr(&red[1], 1u);
    }
    __syncthreads();
    const int lmax = (int)red[0];
    const bool is_new = red[1] == 0u && s_lid != ENT_NONE;

    uint32_t hits = 0, newpass = 0, ncand = 0;
    if (chunk == 0 && s != STEP_NOMATCH) {
        const uint16_t *step = img + a.L.step_at();
        for (int M = lmax + 1 + tid; M <= cap; M += CHILD_THREADS) {
            uint32_t hf = whash_init(), hr = whash_init();
            bool ok = true, pal = true;
            for (int t = 0; t < M; ++t) {
                const uint32_t cf = stepA(n - M + t);
                const uint32_t cr = stepA(n - 1 - t) ^ 1u;
                ok &= cf != STEP_NOMATCH;
                pal &= cf == cr;
                hf = whash_step(hf, cf);
                hr = whash_step(hr, cr);
            }
            if (!ok) continue;
            hits += ct_lookup(a, whash_final(hf, M), M, step, n - M, 1, 0u);
            if (!pal) hits += ct_lookup(a, whash_final(hr, M), M, step, n - 1, -1, 1u);
        }
    }

    // one alignment of an inverted list per lane
    uint32_t (*mine)[WAVE] = stage[tid >> 6];
    uint32_t b0 = 0;
    const uint16_t *bp = nullptr;
    bool staged = false;
    auto load_B = [&](const uint4 ent) -> int {
        const uint32_t slot = ent.x;
        const int m = (int)(ent.y & 0xFFFFu);
        b0 = ent.y >> 16;
        staged = m <= 2 * CHILD_STAGE_PAIRS + 1;
        mine[0][lane] = ent.z;
        mine[1][lane] = ent.w;
        if (m > 5) {                                         // the rest of a longer alignment
            const uint4 h = a.items.hdr[slot >> 6];          // base, pbase, common, len
            bp = a.items.steps + (size_t)h.x * WAVE + (slot & 63u);
            if (staged) {
                const uint32_t *pp = a.items.pairs + (size_t)h.y * WAVE + (slot & 63u);
                for (int j = 2; j < m / 2; ++j) mine[j][lane] = pp[(size_t)j * WAVE];
            }
        }
        return m;
    };
    auto B = [&](int t) -> uint32_t {
        if (t == 0) return b0;
        if (staged) return (mine[(t - 1) >> 1][lane] >> (((t - 1) & 1) * 16)) & 0xFFFFu;
        return bp[(size_t)t * WAVE];
    };
    const uint32_t g = (uint32_t)chunk * CHILD_THREADS + (uint32_t)tid;
    const uint32_t stride = (uint32_t)n_chunks * CHILD_THREADS;
    if (is_new) {      // the alignments that carry the new node: which of them pass the filter now
        const uint32_t lo = a.ix.inv_off[s_lid * N_CLASSES], hi = a.ix.inv_off[(s_lid + 1u) * N_CLASSES];
        for (uint32_t e = lo + g; e < hi; e += stride) {
            const uint4 ent = a.ix.inv_ent[e];
            const uint32_t slot = ent.x;
            const int m = load_B(ent);
            bool pass = true;
            for (int t = 0; t < m && pass; ++t) pass = on_path(B(t));
            if (pass) newpass += W ? a.items.weight[slot] : 1u;
        }
    }
    // Start-overhang candidates: the alignments that carry the path's first node.  What
    // the exact DP said about one of them for the stored ancestor still holds for this
    // path if none of its nodes is on the path's last Mmax + depth steps: the rows in
    // between only subtract, the table was in its steady state at the ancestor and stays
    // there, and no new node or window concerns the alignment (tests/incr_model.py,
    // inherits()).  The ancestor's verdicts sit in a bitmap over the list; this path's
    // own bitmap takes the inherited bits here and the rest from the DP kernels.
    // Candidates that are not inherited are gathered in LDS and leave the workgroup
    // with ONE atomic on the list cursor (an atomic per wave and round on that one word
    // was 90 % of the first version of this kernel).
    // The bitmaps are sized for the longest list of at most GFAL_BITS_MAX_LIST (4 M)
    // entries: a path that starts on a node with a longer list (a hub in more alignments
    // than that) keeps no bitmap and inherits nothing -- everything is recomputed for it,
    // and its slot says so (st_bitsok == 2) to its children.
    bool remember = a.wl_pos != nullptr;
    const int root = a.b.root[child], my_slot = a.b.slot[child];
    {
        if (remember && a0 != STEP_NOMATCH && a0_lid != ENT_NONE) {
            const uint32_t len0 = a.ix.inv_off[(a0_lid + 1u) * N_CLASSES] - a.ix.inv_off[a0_lid * N_CLASSES];
            if (len0 > a.b.bits_words * 32u) remember = false;
        }
        if (a.wl_pos != nullptr && my_slot >= 0 && chunk == 0 && tid == 0) a.b.st_bitsok[my_slot] = remember ? 0 : 2;
    }
    const bool can_inherit = remember && root >= 0 && a.b.st_bitsok[root] == 1;
    // one node mask of the path's tail per DP length class: an alignment of m steps looks at
    // the last m + depth steps, its class's longest member stands in for m (4, 8, 16, 32,
    // the longest alignment): short alignments, the bulk, look at a short tail
    uint32_t *tail = reinterpret_cast<uint32_t *>(lid + a.L.nm);      // [N_CLASSES][v2 / 32]
    const int tail_words = (a.L.v2 + 31) / 32;
    if (can_inherit) {
        for (int i = tid; i < N_CLASSES * tail_words; i += CHILD_THREADS) tail[i] = 0;
        __syncthreads();
        const int depth = a.b.depth[child];
        const int from = max(0, n - (a.max_aln_len + depth));
        for (int p = from + tid; p < n; p += CHILD_THREADS) {
            const uint32_t v = lds_u16(lid32, (uint32_t)p);
            if (v == ENT_NONE) continue;
            const int back = n - p;          // 1 = the path's last step
            for (int c = 0; c < N_CLASSES; ++c) {
                const int reach = (c == LONG_CLASS ? a.max_aln_len : (4 << c)) + depth;
                if (back <= reach) atomicOr(&tail[c * tail_words + (int)(v >> 5)], 1u << (v & 31u));
            }
        }
        __syncthreads();
    }
    auto flush = [&]() {            // (every thread of the workgroup calls it)
        __syncthreads();
        const uint32_t cnt = wl_n;
        if (cnt != 0u) {
            if (tid == 0) {
                const unsigned long long base = atomicAdd(a.wl_count, (unsigned long long)cnt);
                wl_base[0] = (uint32_t)base;
                wl_base[1] = (uint32_t)(base >> 32);
                for (int c = 0; c < N_CLASSES; ++c)
                    if (wl_cls[c]) {
                        atomicAdd(&a.wl_hist[(uint32_t)c * (uint32_t)a.n_paths + (uint32_t)q], wl_cls[c]);
                        wl_cls[c] = 0;
                    }
            }
            __syncthreads();
            const unsigned long long base = ((unsigned long long)wl_base[1] << 32) | wl_base[0];
            bool over = false;
            for (uint32_t i = (uint32_t)tid; i < cnt; i += CHILD_THREADS) {
                if (base + i < a.wl_capacity) {
                    a.worklist[base + i] = wl_buf[i];
                    if (remember) a.wl_pos[base + i] = wl_at[i];
                } else {
                    over = true;
                }
            }
            if (over) atomicOr(a.status, ST_DP_OVERFLOW);
            __syncthreads();
            if (tid == 0) wl_n = 0;
        }
        __syncthreads();
    };
    uint32_t inh_good = 0;
    if (a0 != STEP_NOMATCH && a0_lid != ENT_NONE) {
        const uint32_t list_lo = a.ix.inv_off[a0_lid * N_CLASSES], list_hi = a.ix.inv_off[(a0_lid + 1u) * N_CLASSES];
        const uint32_t *pbits = can_inherit ? a.b.st_bits + (size_t)root * a.b.bits_words : nullptr;
        uint32_t *cbits = (remember && my_slot >= 0) ? a.b.st_bits + (size_t)my_slot * a.b.bits_words : nullptr;
        for (uint32_t e0 = list_lo + (uint32_t)chunk * CHILD_THREADS; e0 < list_hi; e0 += stride) {
            const uint32_t ew = e0 + (uint32_t)(tid & ~(WAVE - 1));     // the wave's 64 list positions:
            const uint32_t e = ew + (uint32_t)lane;                     // (ew - list_lo) is a multiple of 64
            const uint32_t pos = e - list_lo;
            bool fw = false, rc = false, inh_bit = false;
            uint32_t slot = 0;
            int m = 0;
            if (e < list_hi) {
                const uint4 ent = a.ix.inv_ent[e];
                slot = ent.x;
                m = load_B(ent);
                bool inherit = can_inherit;
                const uint32_t *my_tail = tail + length_class(m) * tail_words;
                for (int t = 0; t < m && inherit; ++t) {
                    const uint32_t v = B(t) >> 1;
                    inherit = ((my_tail[v >> 5] >> (v & 31u)) & 1u) == 0u;
                }
                if (inherit) {
                    inh_bit = ((pbits[pos >> 5] >> (pos & 31u)) & 1u) != 0u;
                } else if (m >= 2 && m <= n) {
                    // a proper suffix of B (of rc(B)) equals a prefix of the path
                    for (int t = 1; t < m && !fw; ++t)
                        if (B(t) == a0) {
                            bool eq = true;
                            for (int k = 1; k < m - t && eq; ++k) eq = B(t + k) == stepA(k);
                            fw = eq;
                        }
                    for (int t = 0; t + 1 < m && !rc; ++t)
                        if ((B(t) ^ 1u) == a0) {
                            bool eq = true;
                            for (int k = 1; k <= t && eq; ++k) eq = (B(t - k) ^ 1u) == stepA(k);
                            rc = eq;
                        }
                }
            }
            if (fw || rc) {
                bool pass = true;      // the filter (src/eval.cpp:81-91)
                for (int t = 0; t < m && pass; ++t) pass = on_path(B(t));
                bool found = false;    // a subpath on either strand is good without the DP
                if (pass) {
                    uint32_t ent = lds_u16(first32, b0 >> 1);
                    while ((ent & 0x7C00u) == 0u && !found) {
                        const int pos_a = (int)(ent & ENT_POS);
                        const bool neg = (ent & ENT_NEG) != 0u;
                        bool eq = false;
                        if (neg == ((b0 & 1u) != 0u)) {
                            if (pos_a + m <= n) {
                                eq = true;
                                for (int t = 1; t < m && eq; ++t) eq = stepA(pos_a + t) == B(t);
                            }
                        } else if (pos_a >= m - 1) {
                            eq = true;
                            for (int t = 1; t < m && eq; ++t) eq = stepA(pos_a - t) == (B(t) ^ 1u);
                        }
                        found = eq;
                        ent = next[pos_a];
                    }
                }
                if (!pass || found) fw = rc = false;
            }
            // this path's bitmap: the wave's 64 positions are two whole words of it
            const lanemask im = WAVE_MASK(inh_bit);
            if (cbits && ew < list_hi && lane < 2) {
                const uint32_t word = ((ew - list_lo) >> 5) + (uint32_t)lane;
                if (word < a.b.bits_words) cbits[word] = (uint32_t)(im >> (32 * lane));
            }
            if (inh_bit) inh_good += W ? a.items.weight[slot] : 1u;
            const bool want = fw || rc;
            const lanemask wm = WAVE_MASK(want);
            if (wm != 0ull) {
                const int leader = __builtin_ctzll(wm);
                const int cls = length_class(m);
                uint32_t at = 0;
                if (lane == leader) at = atomicAdd(&wl_n, (uint32_t)__builtin_popcountll(wm));
                at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader);
                for (int c = 0; c < N_CLASSES; ++c) {      // (the list is ordered by class: one or two per wave)
                    const lanemask mc = WAVE_MASK(want && cls == c);
                    if (mc != 0ull && lane == __builtin_ctzll(mc))
                        atomicAdd(&wl_cls[c], (uint32_t)__builtin_popcountll(mc));
                }
                if (want) {
                    const uint32_t k = at + lanes_below(wm, lane);
                    wl_buf[k] = (fw ? WL_FW : 0ull) | (rc ? WL_RC : 0ull) | ((unsigned long long)(uint32_t)q << 32) | slot;
                    wl_at[k] = pos;
                    ncand += W ? a.items.weight[slot] : 1u;
                }
            }
            // every thread must take the same decision: wl_n is read between two barriers, so
            // no wave that runs ahead into the next round can have added to it meanwhile
            __syncthreads();
            const bool full = wl_n > (uint32_t)(CHILD_WL_BUF - CHILD_THREADS);
            __syncthreads();
            if (full) flush();
        }
    }
    flush();
    // the workgroup's sums: the DP kernels add every candidate to good or to bad, so
    // the candidates leave `bad` here (k_child_resolve put Pass' - G1' there)
    for (int o = 32; o > 0; o >>= 1) {
        hits += __shfl_down(hits, o, WAVE);
        newpass += __shfl_down(newpass, o, WAVE);
        ncand += __shfl_down(ncand, o, WAVE);
        inh_good += __shfl_down(inh_good, o, WAVE);
    }
    if (lane == 0) {
        if (hits) atomicAdd(&a.dg1[child], hits);
        if (newpass) atomicAdd(&a.dpass[child], newpass);
        // (inherited verdicts: the accepted ones move from bad to good right here)
        if (ncand + inh_good) atomicSub(&a.counts[q], ncand + inh_good);
        if (inh_good) atomicAdd(&a.counts[a.n_paths + q], inh_good);
    }
}

// Pass / G1 of every child from its stored ancestor's and the deltas on the way: into
// the child's store slot and, with what the DP kernels added, into the caller's
// counters (a children batch is not sorted: image slot = child index; this is its
// k_unpermute).  counts[q] holds  -#candidates + the DP's bad,  counts[n + q]  the
// zero-step alignments + the DP's good.
__global__ void k_child_resolve(ChildBatch b, int n_paths, uint32_t n_empty,
                                const uint32_t *__restrict__ counts, uint32_t *__restrict__ out,
                                const uint32_t *__restrict__ status, uint32_t *__restrict__ status_copy)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (status_copy && q < 4) status_copy[q] = status[q];
    if (q >= n_paths) return;
    const int r = b.root[q];
    uint32_t pass = 0, g1 = n_empty;
    if (r >= 0) {
        pass = b.st_pass[r];
        g1 = b.st_g1[r];
        int j = q;
        for (int k = b.depth[q]; k > 0; --k) {
            pass += b.dpass[j];
            g1 += b.dg1[j];
            j = ~b.parent[j];
        }
        const int slot = b.slot[q];
        if (slot >= 0) {
            b.st_pass[slot] = pass;
            b.st_g1[slot] = g1;
            if (b.bits_words && b.st_bitsok[slot] != 2) b.st_bitsok[slot] = 1;     // k_child + the DP kernels filled it
        }
    }
    out[q] = counts[q] + (pass - g1);
    out[n_paths + q] = counts[n_paths + q] + (g1 - n_empty);
    out[2 * n_paths + q] = counts[2 * n_paths + q];
}

// full scoring with the store: good-without-DP per image slot, taken between the
// scan and the DP kernels ...
__global__ void k_store_snapshot(const uint32_t *__restrict__ counts, int n_paths, uint32_t *__restrict__ g1_tmp)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_paths) g1_tmp[q] = counts[n_paths + q];
}

// ... and the paths with their Pass / G1 into their slots when the call is done
__global__ __launch_bounds__(256) void k_store_paths(const int32_t *__restrict__ path_off,
                                                     const int32_t *__restrict__ path_steps,
                                                     const int32_t *__restrict__ order, int n_paths,
                                                     const uint32_t *__restrict__ counts,
                                                     const uint32_t *__restrict__ g1_tmp,
                                                     const int32_t *__restrict__ slots, int32_t *st_steps,
                                                     int32_t *st_len, uint32_t *st_pass, uint32_t *st_g1,
                                                     int32_t *st_bitsok, int64_t st_cap,
                                                     uint32_t *__restrict__ status)
{
    const int q = blockIdx.x, tid = threadIdx.x;
    const int i = order[q];
    const int slot = slots[i];
    if (slot < 0) return;
    if (slot >= st_cap) {
        if (tid == 0) atomicOr(status, ST_BAD_CHILD);
        return;
    }
    const int off = path_off[i], L = path_off[i + 1] - off;
    if (L < 1 || L > STORE_STRIDE) return;
    int32_t *keep = st_steps + (size_t)slot * STORE_STRIDE;
    for (int k = tid; k < L; k += 256) keep[k] = path_steps[off + k];
    if (tid == 0) {
        st_len[slot] = L;
        st_pass[slot] = counts[q] + counts[n_paths + q];
        st_g1[slot] = g1_tmp[q];
        if (st_bitsok) st_bitsok[slot] = 0;     // (the scan kernels do not know list positions)
    }
}

// --------------------------------------------------------------------------
// k_tile + k_scan3 (round 3): the subpath test by content identity.
//
// k_scan2 finds an alignment among a tile's windows through a hash of its steps
// and then compares the steps themselves with the staged path -- which is why
// the paths' steps (both strands) live in LDS, why a tile holds 8 paths, and why
// every (item, tile) visit costs ~125 VALU instructions of which 8 paths share
// the bill.  Here the comparison of steps is done ONCE per distinct window,
// against the content table `create` builds over the alignments (one entry per
// distinct step sequence; the entry's index is the content's identity and every
// lane carries the index of its own alignment):
//
//   k_tile_masks   per tile of T <= 31 paths: the node masks (which tile paths
//                  carry each node), written to HBM once instead of being
//                  rebuilt by every workgroup of the tile.
//   k_tile         per (tile, alignment length M): every M-step window of the
//                  tile's paths, both strands, is looked up in the content
//                  table (exact: hash, length, steps); the windows that ARE some
//                  alignment's content end up as a list of {content index, mask
//                  of the tile paths that contain it} in HBM.  Windows a path
//                  shares with the tile's first path (prefixes, siblings) are
//                  looked up once.
//   k_scan3        workgroup = (tile, M, chunk of items) as in k_scan2, but its
//                  prologue only loads the node masks and enters the list into
//                  an LDS table keyed by content index; per item a lane does its
//                  node-mask reads (the filter for 31 paths at once), ONE
//                  probe sequence that ends in an exact key comparison, and
//                  has the answer for all T paths.  No steps in LDS, no window
//                  comparison, ~4x the paths per (item, tile) visit.
//
// Exactness: two alignments have the same index iff they have the same steps
// (k_ct_build compares steps), a window gets an index only after its steps were
// compared with the entry's, and the LDS probe compares full 32-bit indices.
// --------------------------------------------------------------------------
constexpr int T3_MAX = 31;                 // tile paths: bits 0..30 of a node mask (bit 31: NOT_A0)
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr int T3_THDR_WORDS = 384;         // per tile: 32 lengths | 32 first steps | 32 common prefixes with path 0 | 32 x the first 16 steps (8 dwords)
constexpr int T3_THDR_PRE = 96;
constexpr int T3_CT_INLINE = 5;            // steps of an alignment a content-table record carries

struct ContentTable {
    const uint4 *rec;       // {representative lane or CT_EMPTY, m | s0 << 16, s1 | s2 << 16, s3 | s4 << 16}
    uint32_t mask;
    // one bit per hashed pair of consecutive steps that occurs in some alignment, on either
    // strand (a Bloom filter with one hash: no false negatives).  A window that contains a
    // pair no alignment has cannot be an alignment's content: k_tile skips its lookups.
    const uint32_t *pair_bits;
};

struct TileArgs {
    Items items;
    ContentTable ct;
    const uint16_t *images;
    ImageLayout L;
    const uint16_t *lids;
    int n_paths, tile, tile0;      // tile0: first tile of this launch (slab)
    const LenSeg *segs;
    int n_segs;
    int debug;                     // GFAL_DEBUG_TILE: stop k_tile after phase 1 (A) / 2 (R); 3: first path's windows only; 4: up to 5 steps only (timing probes)
    int filter, v2p;               // v2p: mask words per tile = v2 + 2 (word v2 stays zero: padding lanes point at it)
    uint32_t *tile_hdr;            // [tiles of the slab][T3_THDR_WORDS]
    uint32_t *tile_masks;          // [tiles of the slab][v2p]
    // reference: the content indices of every window (<= T3_REG_M steps) of the batch's longest
    // path (image slot 0), [n_segs][2 strands][L.nm], KEY_EMPTY = no alignment has that content.
    // A tile whose first path agrees with the reference over a window copies the index
    // (coalesced) instead of probing the content table (one scattered line per probe).
    uint32_t *ref;
    int n_launch_tiles;            // k_tile_masks: blocks beyond this many fill `ref`
    uint32_t *list_count;          // [tiles of the slab][n_segs] entries of every list
    uint2 *list;                   // [tiles of the slab][n_segs][stride] {content index, tile paths}
    uint32_t stride;
};

// which tile paths carry each node (the filter of src/eval.cpp:81-91 as a bit test);
// bit 31 (NOT_A0) on every node but the one of the tile's first step.  What k_scan3 would
// otherwise test per item is folded into the masks: without the filter every node counts
// as being on every path; tile paths that do not start with the same step: no node keeps
// NOT_A0 (every open pair takes the overhang test).  Also the tile's header (lengths,
// first steps) and -- block 0 -- the cold arguments of k_scan3.
struct Scan3Cold;
__device__ void t3_ref_block(const TileArgs &a, int block, int tid);
__global__ __launch_bounds__(1024) void k_tile_masks(TileArgs a, const Scan3Cold *cold_src, Scan3Cold *cold_dst,
                                                     int cold_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.n_launch_tiles) {       // the reference blocks
        t3_ref_block(a, (int)blockIdx.x - a.n_launch_tiles, tid);
        return;
    }
    if (blockIdx.x == 0 && cold_dst)
        for (int i = tid; i < cold_bytes / 4; i += 1024)
            reinterpret_cast<uint32_t *>(cold_dst)[i] = reinterpret_cast<const uint32_t *>(cold_src)[i];
    const int tile = a.tile0 + (int)blockIdx.x;
    const int path0 = tile * a.tile;
    const int T = min(a.tile, a.n_paths - path0);
    const int v2 = a.L.v2, nm = a.L.nm;
    const uint32_t init = a.filter ? NOT_A0 : 0xFFFFFFFFu;
    for (int v = tid; v < v2; v += 1024) lds32[v] = init;
    if (tid < WAVE) {
        uint32_t n = 0, a0 = STEP_NOMATCH;
        if (tid < T) {
            const uint16_t *img = a.images + (size_t)(path0 + tid) * a.L.total;
            n = img[a.L.len_at()];
            a0 = img[a.L.step_at()];
        }
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)a0, 0);
        const bool uniform = WAVE_MASK(tid < T && a0 != first) == 0ull;
        if (tid < 32) {
            a.tile_hdr[(size_t)blockIdx.x * T3_THDR_WORDS + tid] = n;
            a.tile_hdr[(size_t)blockIdx.x * T3_THDR_WORDS + 32 + tid] = a0;
        }
        if (tid == 0) {
            lds32[v2] = uniform ? 1u : 0u;
            lds32[v2 + 1] = first;
        }
    }
    __syncthreads();
    // how many of its first 64 steps every path shares with the batch's longest path (image
    // slot 0): k_scan3's overhang test looks at M - 1 leading steps, and for the paths that
    // share them with that one path the answer was worked out per alignment (k_overhang)
    for (int t = tid >> 6; t < T; t += 1024 / WAVE) {
        const int l = tid & (WAVE - 1);
        const uint32_t *p0 = reinterpret_cast<const uint32_t *>(a.images + a.L.step_at());
        const uint32_t *pt = reinterpret_cast<const uint32_t *>(a.images + (size_t)(path0 + t) * a.L.total + a.L.step_at());
        uint32_t x = 0;
        if (l < 32 && l < nm / 2) x = p0[l] ^ pt[l];
        const lanemask lo = WAVE_MASK((x & 0xFFFFu) != 0u), hi = WAVE_MASK((x >> 16) != 0u);
        int lcp = 64;
        if (lo) lcp = min(lcp, 2 * __builtin_ctzll(lo));
        if (hi) lcp = min(lcp, 2 * __builtin_ctzll(hi) + 1);
        if (l == 0) a.tile_hdr[(size_t)blockIdx.x * T3_THDR_WORDS + 64 + t] = (uint32_t)lcp;
        // the path's first 16 steps, for k_scan3's overhang test
        if (l < 8) a.tile_hdr[(size_t)blockIdx.x * T3_THDR_WORDS + T3_THDR_PRE + t * 8 + l] = l < nm / 2 ? pt[l] : 0xFFFFFFFFu;
    }
    for (int t = 0; t < T; ++t) {
        const uint32_t *lsrc = reinterpret_cast<const uint32_t *>(a.lids + (size_t)(path0 + t) * nm);
        for (int o = tid; o < nm / 2; o += 1024) {
            const uint32_t d = lsrc[o];
            const uint32_t lo = d & 0xFFFFu, hi = d >> 16;
            if (lo != 0xFFFFu) atomicOr(&lds32[lo], 1u << t);
            if (hi != 0xFFFFu) atomicOr(&lds32[hi], 1u << t);
        }
    }
    __syncthreads();
    const bool uniform = lds32[v2] != 0u;
    const uint32_t a0 = lds32[v2 + 1];
    uint32_t *dst = a.tile_masks + (size_t)blockIdx.x * a.v2p;
    for (int v = tid; v < v2; v += 1024) {
        uint32_t m = lds32[v];
        if (!uniform || (a0 < STEP_NOMATCH && (uint32_t)v == (a0 >> 1))) m &= ~NOT_A0;
        dst[v] = m;
    }
    if (tid < 2) dst[v2 + tid] = 0;
}

// does the record r hold the content  W[k] = F[dir * k] ^ flip  (k < M)?  w: its first steps.
__device__ __forceinline__ bool ct_match(const TileArgs &a, const uint4 &r, int M, const uint16_t *F, int dir,
                                         uint32_t flip, const uint32_t (&w)[T3_CT_INLINE])
{
    if ((int)(r.y & 0xFFFFu) != M) return false;
    bool eq = (r.y >> 16) == w[0];
    if (M > 1) eq &= (r.z & 0xFFFFu) == w[1];
    if (M > 2) eq &= (r.z >> 16) == w[2];
    if (M > 3) eq &= (r.w & 0xFFFFu) == w[3];
    if (M > 4) eq &= (r.w >> 16) == w[4];
    if (eq && M > T3_CT_INLINE) {
        const uint32_t it = r.x >> 6;
        const uint16_t *bp = a.items.steps + (size_t)a.items.base[it] * WAVE + (r.x & 63u);
        for (int k = T3_CT_INLINE; k < M; ++k)
            eq &= (uint32_t)bp[(size_t)k * WAVE] == ((uint32_t)F[dir * k] ^ flip);
    }
    return eq;
}

// Append v to a window list for the lanes that `have` one: one LDS atomic per wave (1024
// threads adding to one counter lane by lane serialise).  May be called in divergent code.
__device__ __forceinline__ void t3_append(uint32_t *counter, uint2 *out, bool have, uint2 v)
{
    const lanemask m = WAVE_MASK(have);
    if (m == 0) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (have) out[base + lanes_below(m, lane)] = v;
}

// One window's two lookups (forward content, reverse complement) on their way through
// the content table: issue() sends the first probes, resolve() follows them up and
// appends {content index, tile paths} to the list of the window's length.
struct T3Pend {
    uint4 rf, rr;
    uint32_t xf, xr, bits;
    int seg, M;      // M == 0: nothing pending
};

__device__ __forceinline__ void t3_issue(const TileArgs &a, T3Pend &p, uint32_t hf, uint32_t hr)
{
    p.xf = hf & a.ct.mask;
    p.xr = hr & a.ct.mask;
    p.rf = a.ct.rec[p.xf];
    p.rr = a.ct.rec[p.xr];
}

__device__ __forceinline__ void t3_resolve(const TileArgs &a, T3Pend &p, const uint16_t *F,
                                           const uint32_t (&wf)[T3_CT_INLINE], const uint32_t (&wr)[T3_CT_INLINE],
                                           uint32_t *cnt, uint2 *lists)
{
    const int M = p.M;
    uint32_t kf = KEY_EMPTY, kr = KEY_EMPTY;
    while (p.rf.x != CT_EMPTY) {
        if (ct_match(a, p.rf, M, F, 1, 0u, wf)) {
            kf = p.xf;
            break;
        }
        p.xf = (p.xf + 1u) & a.ct.mask;
        p.rf = a.ct.rec[p.xf];
    }
    while (p.rr.x != CT_EMPTY) {
        if (ct_match(a, p.rr, M, F + (M - 1), -1, 1u, wr)) {
            kr = p.xr;
            break;
        }
        p.xr = (p.xr + 1u) & a.ct.mask;
        p.rr = a.ct.rec[p.xr];
    }
    // (p.seg is the same for every lane that is here with this slot?  Not necessarily:
    // lanes fill their slots at different segments -- so one append per distinct segment)
    lanemask todo = WAVE_MASK(true);
    while (todo) {
        const int seg = __builtin_amdgcn_readlane(p.seg, __builtin_ctzll(todo));
        const bool mine = p.seg == seg;
        uint2 *out = lists + (size_t)seg * a.stride;
        t3_append(&cnt[seg], out, mine && kf != KEY_EMPTY, make_uint2(kf, p.bits));
        t3_append(&cnt[seg], out, mine && kr != KEY_EMPTY && kr != kf, make_uint2(kr, p.bits));
        todo &= ~WAVE_MASK(mine);
    }
    p.M = 0;
}

constexpr int T3_REG_M = 16;         // windows of up to 16 steps are hashed from registers
constexpr int T3_RUN_CAP = 127;      // run[][]: positions two paths agree on from here, capped
constexpr int T3_PEND = 3;           // windows of one thread in flight (two lookups each)
constexpr int T3_WL_CAP = 4096;      // (path, position) pairs with windows of their own, gathered in LDS

// tile paths t with run[f][t] >= M, from the 32 run bytes of position f (eight dwords)
__device__ __forceinline__ uint32_t t3_share_bits(const uint32_t (&r)[8], int M)
{
    uint32_t bits = 0;
    const uint32_t m4 = (uint32_t)M * 0x01010101u;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const uint32_t ge = (((r[d] | 0x80808080u) - m4) & 0x80808080u) >> 7;     // bit 8 i: byte i >= M
        bits |= ((ge * 0x00204081u) >> 21 & 0xFu) << (4 * d);
    }
    return bits & 0x7FFFFFFFu;       // (byte 31 is the run against the reference path, not a tile path)
}

// the first T3_REG_M steps from F (path of n steps, position f) and how many of them can
// match something
__device__ __forceinline__ int t3_load_steps(const uint16_t *F, int f, int n, uint32_t (&c)[T3_REG_M])
{
#pragma unroll
    for (int k = 0; k < T3_REG_M; ++k) c[k] = f + k < n ? (uint32_t)F[k] : 0xFFFFu;
    int m_real = T3_REG_M;
#pragma unroll
    for (int k = T3_REG_M - 1; k >= 0; --k)
        if (c[k] >= STEP_NOMATCH) m_real = k;
    return m_real;
}

// hash of the reverse complement of the first M of the steps c
__device__ __forceinline__ uint32_t t3_hash_rc(const uint32_t (&c)[T3_REG_M], int M)
{
    uint32_t hr = whash_init();
#pragma unroll
    for (int k = T3_REG_M - 1; k >= 0; --k)
        if (k < M) hr = whash_step(hr, c[k] ^ 1u);
    return whash_final(hr, M);
}

// the first steps of the M-step window c and of its reverse complement (ct_match)
__device__ __forceinline__ void t3_first_steps(const uint32_t (&c)[T3_REG_M], int M,
                                               uint32_t (&wf)[T3_CT_INLINE], uint32_t (&wr)[T3_CT_INLINE])
{
#pragma unroll
    for (int j = 0; j < T3_CT_INLINE; ++j) {
        wf[j] = j < M ? c[j] : 0xFFFFFFFFu;
        wr[j] = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < T3_REG_M; ++k)
            if (k == M - 1 - j) wr[j] = c[k] ^ 1u;
    }
}

// All windows that start at position f of one tile path, every length of the scorer's
// segments: hashed, looked up (or copied from the reference), appended.  min_m: lengths up
// to min_m - 1 are someone else's (the window is shared with the tile's first path);
// own_bits: the bits to enter, or 0 = this IS the first path, enter every path that shares
// the window (run row of f).
__device__ __forceinline__ void t3_windows_at(const TileArgs &a, const uint16_t *Fp, int f, int n, int min_m,
                                              uint32_t own_bits, const uint8_t *run, const uint32_t *segm,
                                              uint32_t *cnt, uint2 *lists)
{
    const uint16_t *F = Fp + f;
    uint32_t c[T3_REG_M];
    const int m_real = t3_load_steps(F, f, n, c);
    uint32_t r8[8];
    int run_ref = 0;                     // the first path agrees with the reference this far from f on
    if (own_bits == 0u) {
        const uint4 *row = reinterpret_cast<const uint4 *>(run + (size_t)f * 32);
        const uint4 lo = row[0], hi = row[1];
        r8[0] = lo.x, r8[1] = lo.y, r8[2] = lo.z, r8[3] = lo.w;
        r8[4] = hi.x, r8[5] = hi.y, r8[6] = hi.z, r8[7] = hi.w;
        run_ref = (int)(hi.w >> 24);
    } else {
        // a position with windows of its own (this path and the tile's first path differ
        // nearby): where the path still agrees with the batch's longest path -- it is the
        // tile's FIRST path that carries the substituted step -- the indices are the reference's
        const uint16_t *Fr = a.images + a.L.step_at() + f;      // image slot 0
        const int n_ref = (int)a.images[a.L.len_at()];
        int agree = 0;
        bool same = true;
#pragma unroll
        for (int k = 0; k < T3_REG_M; ++k) {
            const uint32_t r = f + k < n_ref ? (uint32_t)Fr[k] : 0xFFFEu;
            same = same && r == c[k] && c[k] < STEP_NOMATCH;
            agree += same ? 1 : 0;
        }
        run_ref = agree;
    }
    // Lengths from skip_from on need no lookup: the window then holds a pair of consecutive
    // steps that no alignment has on either strand (ContentTable::pair_bits) -- what a
    // substituted step nearly always brings.  The pairs around the first step that is not
    // the reference's (position run_ref of the window; for the tile's first path: where it
    // leaves the reference) are asked once for all lengths.
    int skip_from = T3_REG_M + 1;
    if (run_ref < m_real && run_ref < T3_REG_M) {
        const int d = run_ref;
        bool have1 = true, have2 = true;      // (c[d - 1], c[d]) and (c[d], c[d + 1])
        uint32_t cd = 0, cm = 0, cp = 0;
#pragma unroll
        for (int k = 0; k < T3_REG_M; ++k) {
            cd = k == d ? c[k] : cd;
            cm = k == d - 1 ? c[k] : cm;
            cp = k == d + 1 ? c[k] : cp;
        }
        if (d > 0) have1 = pair_present(a.ct.pair_bits, cm, cd);
        if (d + 1 < m_real) have2 = pair_present(a.ct.pair_bits, cd, cp);
        if (!have1) skip_from = d + 1;                    // every window that reaches position d
        else if (!have2) skip_from = d + 2;               // every window that reaches position d + 1
    }
    T3Pend pend[T3_PEND];
#pragma unroll
    for (int g = 0; g < T3_PEND; ++g) pend[g].M = 0;
    uint32_t hf = whash_init();
    int k_done = 0, n_pend = 0;
    for (int s = 0; s < a.n_segs; ++s) {
        const int M = (int)segm[s];            // (uniform)
        if (M > T3_REG_M) continue;            // (the long ones: below)
        if (a.debug == 4 && M > T3_CT_INLINE) continue;
        if (M < k_done) {                      // (lengths ascend inside a run of segments)
            hf = whash_init();
            k_done = 0;
        }
#pragma unroll
        for (int k = 0; k < T3_REG_M; ++k)
            if (k >= k_done && k < M) hf = whash_step(hf, c[k]);
        k_done = M;
        if (M > m_real || M < min_m) continue;
        const uint32_t bits = own_bits ? own_bits : t3_share_bits(r8, M);
        if (run_ref >= M) {                    // the reference path's window: its indices
            const uint32_t kf = a.ref[((size_t)s * 2) * a.L.nm + f], kr = a.ref[((size_t)s * 2 + 1) * a.L.nm + f];
            uint2 *out = lists + (size_t)s * a.stride;
            t3_append(&cnt[s], out, kf != KEY_EMPTY, make_uint2(kf, bits));
            t3_append(&cnt[s], out, kr != KEY_EMPTY && kr != kf, make_uint2(kr, bits));
            continue;
        }
        if (M >= skip_from) continue;          // (holds a pair of steps no alignment has)
        // (a slot of the in-flight set; the set is resolved when it is full)
#pragma unroll
        for (int g = 0; g < T3_PEND; ++g)
            if (g == n_pend) {
                T3Pend &p = pend[g];
                p.M = M;
                p.seg = s;
                p.bits = bits;
                t3_issue(a, p, whash_final(hf, M), t3_hash_rc(c, M));
            }
        if (++n_pend == T3_PEND) {
#pragma unroll
            for (int g = 0; g < T3_PEND; ++g) {
                uint32_t wf[T3_CT_INLINE], wr[T3_CT_INLINE];
                t3_first_steps(c, pend[g].M, wf, wr);
                t3_resolve(a, pend[g], F, wf, wr, cnt, lists);
            }
            n_pend = 0;
        }
    }
#pragma unroll
    for (int g = 0; g < T3_PEND; ++g)
        if (g < n_pend) {
            uint32_t wf[T3_CT_INLINE], wr[T3_CT_INLINE];
            t3_first_steps(c, pend[g].M, wf, wr);
            t3_resolve(a, pend[g], F, wf, wr, cnt, lists);
        }
    // alignments of more than T3_REG_M steps: hashed from memory, one at a time
    for (int s = 0; s < a.n_segs; ++s) {
        const int M = (int)segm[s];
        if (M <= T3_REG_M || f + M > n || M < min_m) continue;
        uint32_t h1 = whash_init(), h2 = h1;
        bool real = true;
        for (int k = 0; k < M; ++k) {
            const uint32_t x = F[k], y = (uint32_t)F[M - 1 - k] ^ 1u;
            real &= x < STEP_NOMATCH;
            h1 = whash_step(h1, x);
            h2 = whash_step(h2, y);
        }
        if (!real) continue;
        T3Pend p;
        p.M = M;
        p.seg = s;
        if (own_bits) {
            p.bits = own_bits;
        } else {
            p.bits = 0;
            if (M <= T3_RUN_CAP) {
                for (int t = 0; t < T3_MAX; ++t) p.bits |= (int)run[(size_t)f * 32 + t] >= M ? (1u << t) : 0u;
            } else {
                p.bits = 1u;       // (longer than a run is ever counted: every path enters its own)
            }
        }
        uint32_t wf[T3_CT_INLINE], wr[T3_CT_INLINE];
#pragma unroll
        for (int j = 0; j < T3_CT_INLINE; ++j) {
            wf[j] = F[j];
            wr[j] = (uint32_t)F[M - 1 - j] ^ 1u;
        }
        t3_issue(a, p, whash_final(h1, M), whash_final(h2, M));
        t3_resolve(a, p, F, wf, wr, cnt, lists);
    }
}

// The reference (TileArgs::ref), filled by extra blocks of k_tile_masks: thread = (position
// f of the batch's longest path, one length): two lookups.
__device__ void t3_ref_block(const TileArgs &a, int block, int tid)
{
    const int n_r = a.images[a.L.len_at()];
    const uint16_t *Fr = a.images + a.L.step_at();
    const int w = block * 1024 + tid;
    const int s = w / a.L.nm, f = w % a.L.nm;
    if (s >= a.n_segs) return;
    const int M = (int)a.segs[s].m;
    uint32_t kf = KEY_EMPTY, kr = KEY_EMPTY;
    if (M <= T3_REG_M && f + M <= n_r) {
        uint32_t c[T3_REG_M];
        const int m_real = t3_load_steps(Fr + f, f, n_r, c);
        if (M <= m_real) {
            uint32_t hf = whash_init();
#pragma unroll
            for (int k = 0; k < T3_REG_M; ++k)
                if (k < M) hf = whash_step(hf, c[k]);
            T3Pend p;
            p.M = M;
            t3_issue(a, p, whash_final(hf, M), t3_hash_rc(c, M));
            uint32_t wf[T3_CT_INLINE], wr[T3_CT_INLINE];
            t3_first_steps(c, M, wf, wr);
            while (p.rf.x != CT_EMPTY) {
                if (ct_match(a, p.rf, M, Fr + f, 1, 0u, wf)) {
                    kf = p.xf;
                    break;
                }
                p.xf = (p.xf + 1u) & a.ct.mask;
                p.rf = a.ct.rec[p.xf];
            }
            while (p.rr.x != CT_EMPTY) {
                if (ct_match(a, p.rr, M, Fr + f + (M - 1), -1, 1u, wr)) {
                    kr = p.xr;
                    break;
                }
                p.xr = (p.xr + 1u) & a.ct.mask;
                p.rr = a.ct.rec[p.xr];
            }
        }
    }
    a.ref[((size_t)s * 2) * a.L.nm + f] = kf;
    a.ref[((size_t)s * 2 + 1) * a.L.nm + f] = kr;
}

__host__ __device__ inline size_t t3_tile_lds(int n_segs)
{
    return (size_t)(1024 + 8192 + T3_WL_CAP + 32 + 16 + 2 * ((n_segs + 3) & ~3)) * 4;
}

// One workgroup per tile.  Per (tile, length) the list of {content index, tile paths}
// k_scan3 enters into its table: one entry per window of the tile's first path (with the
// paths that share it) and per window another path has of its own; identical contents
// may appear more than once (k_scan3's insert merges them).
__global__ __launch_bounds__(1024) void k_tile(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    uint32_t *eqbm = lds32;                      // [32 words][32 paths] bit f of (f / 32, t): step f of path t == step f of path 0
    uint8_t *run = reinterpret_cast<uint8_t *>(eqbm + 1024);      // [1024][32] run[f][t]: positions from f on where path t agrees
    uint32_t *wl = eqbm + 1024 + 8192;           // [T3_WL_CAP] path | position << 8
    uint32_t *nlen = wl + T3_WL_CAP;             // [32] ([31]: the reference path)
    uint32_t *misc = nlen + 32;                  // [16] [0]: work-list entries
    uint32_t *cnt = misc + 16;                   // [n_segs] entries per list
    uint32_t *segm = cnt + ((a.n_segs + 3) & ~3);                 // [n_segs] the lengths
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_rel = (int)blockIdx.x;
    const int path0 = (a.tile0 + tile_rel) * a.tile;
    const int T = min(a.tile, a.n_paths - path0);
    uint2 *lists = a.list + (size_t)tile_rel * a.n_segs * a.stride;
    const size_t img_stride = (size_t)a.L.total;
    const uint16_t *img0 = a.images + (size_t)path0 * img_stride;

    if (tid < 32) {
        uint32_t n = 0;
        if (tid < T) n = img0[(size_t)tid * img_stride + a.L.len_at()];
        if (tid == 31) n = a.images[a.L.len_at()];       // the reference: image slot 0
        nlen[tid] = n;
    }
    if (tid < 16) misc[tid] = 0;
    for (int s = tid; s < a.n_segs; s += 1024) {
        cnt[s] = 0;
        segm[s] = a.segs[s].m;
    }
    __syncthreads();
    const int n_b = (int)nlen[0];
    const uint16_t *Fb = img0 + a.L.step_at();

    // (A) where every path -- and the reference -- agrees with the tile's first path, one bit
    // per position (a wave takes whole paths; eight blocks of 64 positions are loaded before
    // the first is compared)
    for (int half = 0; half < 2; ++half) {
        if (half * 512 >= n_b && half > 0) {
            for (int i = tid; i < 32 * 16; i += 1024) eqbm[(16 + (i & 15)) * 32 + (i >> 4)] = 0;
            break;
        }
        uint32_t vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = half * 512 + k * WAVE + lane;
            vb[k] = p < n_b ? (uint32_t)Fb[p] : 0x20000u;      // (fillers equal nothing)
        }
        for (int t = wave; t < 32; t += 1024 / WAVE) {
            const int lim = min((int)nlen[t], n_b);
            const uint16_t *Ft = t == 31 ? a.images + a.L.step_at()
                                         : img0 + (size_t)min(t, T - 1) * img_stride + a.L.step_at();
            uint32_t vt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = half * 512 + k * WAVE + lane;
                vt[k] = p < lim ? (uint32_t)Ft[p] : 0x10000u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const lanemask m = WAVE_MASK(vt[k] == vb[k]);
                if (lane == 0) {
                    eqbm[(half * 16 + k * 2) * 32 + t] = (uint32_t)m;
                    eqbm[(half * 16 + k * 2 + 1) * 32 + t] = (uint32_t)(m >> 32);
                }
            }
        }
    }
    __syncthreads();
    if (a.debug == 1) return;
    // (R) thread (j, t): for the positions 32 j .. 32 j + 31 of path t, how far it goes on
    // agreeing with the first path (a window of M steps that lies inside such a run IS the
    // first path's window: entered there, with this path's bit).  Positions of a path
    // that have windows of their own go to the work list.
    int max_m = 0;
    for (int s = 0; s < a.n_segs; ++s) max_m = max(max_m, (int)segm[s]);
    const int q = tid & 31, j = tid >> 5;
    int inline_from = 32;                        // (work list full: the rest is done by this thread itself)
    {
        int nz = 32 * (j + 1);                   // next position >= the dword's end where the paths differ
        for (int jj = j + 1; jj < 32; ++jj) {
            const uint32_t w = ~eqbm[jj * 32 + q];
            if (w) {
                nz = 32 * jj + __builtin_ctz(w);
                break;
            }
            nz = 32 * (jj + 1);
        }
        const uint32_t w = eqbm[j * 32 + q];
        const int n_t = (q >= 1 && q < T) ? (int)nlen[q] : 0;
        uint32_t own = 0;                        // positions of this dword with windows of their own
        for (int i = 31; i >= 0; --i) {
            if (!((w >> i) & 1u)) nz = 32 * j + i;
            const int r = min(nz - (32 * j + i), T3_RUN_CAP);
            run[(size_t)(32 * j + i) * 32 + q] = (uint8_t)r;
            // (a window of its own: longer than the run, and still inside the path)
            if (32 * j + i < n_t && r < max_m && r < n_t - (32 * j + i)) own |= 1u << i;
        }
        if (own && a.debug != 3) {
            const uint32_t k = (uint32_t)__builtin_popcount(own);
            uint32_t at = atomicAdd(&misc[0], k);
            for (int i = 0; i < 32; ++i)
                if ((own >> i) & 1u) {
                    if (at < (uint32_t)T3_WL_CAP) wl[at] = (uint32_t)q | ((uint32_t)(32 * j + i) << 8);
                    else inline_from = min(inline_from, i);
                    ++at;
                }
        }
    }
    __syncthreads();
    if (a.debug == 2) return;
    // (C) work items: the first path's positions, then the work list (and what did not fit it)
    const int n_wl = (int)min(misc[0], (uint32_t)T3_WL_CAP);
    for (int w = tid; w < n_b + n_wl + 1024; w += 1024) {
        const uint16_t *Fp;
        int f, n, min_m;
        uint32_t own;
        if (w < n_b) {
            Fp = Fb, f = w, n = n_b, min_m = 1, own = 0u;
        } else if (w < n_b + n_wl) {
            const uint32_t e = wl[w - n_b];
            const int t = (int)(e & 0xFFu);
            f = (int)(e >> 8);
            Fp = img0 + (size_t)t * img_stride + a.L.step_at();
            n = (int)nlen[t];
            min_m = (int)run[(size_t)f * 32 + t] + 1;
            own = 1u << t;
        } else {
            break;
        }
        t3_windows_at(a, Fp, f, n, min_m, own, run, segm, cnt, lists);
    }
    if (inline_from < 32 && q >= 1 && q < T) {   // (unrelated paths: more positions than the list holds)
        const int n_t = (int)nlen[q];
        const uint16_t *Fp = img0 + (size_t)q * img_stride + a.L.step_at();
        for (int i = inline_from; i < 32; ++i) {
            const int f = 32 * j + i;
            if (f >= n_t) break;
            const int r = run[(size_t)f * 32 + q];
            if (r >= max_m || r >= n_t - f) continue;
            t3_windows_at(a, Fp, f, n_t, r + 1, 1u << q, run, segm, cnt, lists);
        }
    }
    __syncthreads();
    for (int s = tid; s < a.n_segs; s += 1024) a.list_count[(size_t)tile_rel * a.n_segs + s] = cnt[s];
}

// What only the rare paths of k_scan3 read (the overhang triage, the worklist): kept in
// HBM behind one pointer, so that none of it occupies SGPRs across the item loop.
// The overhang test (see scan3_triage) of every alignment of up to 16 steps against the
// first steps of the batch's longest path, once per batch: a wave per item, a lane per
// alignment; two 64-bit masks per item (a proper suffix of B / of rc(B) equals a prefix of
// the path).  Candidate paths of a search all start at the source: nearly every tile path
// shares these steps, and k_scan3 then only reads the masks.
__global__ __launch_bounds__(256) void k_overhang(Items items, int n_items, const uint16_t *images, ImageLayout L,
                                                  uint32_t *__restrict__ ovh)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int it = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (it >= n_items) return;
    const int M = items.len[it];
    if (M > 16) return;                      // (k_scan3 tests those pair by pair)
    const uint16_t *bp = items.steps + (size_t)items.base[it] * WAVE + lane;
    const uint16_t *pre = images + L.step_at();           // image slot 0
    uint32_t b[16], a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        b[t] = t < M ? (uint32_t)bp[t * WAVE] : 0xFFFFu;
        a[t] = pre[t];                       // (L.nm >= 32: always there; 0xFFFF beyond the path)
    }
    bool cand_fw = false, cand_rc = false;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (t >= 1 && t < M) {               // B[t..M) == path[0..M-t) ?
            bool eq = true;
#pragma unroll
            for (int k = 0; k < 16 - t; ++k)
                if (t + k < M) eq &= b[t + k] == a[k];
            cand_fw |= eq;
        }
        if (t < M - 1) {                     // rc(B)[M-1-t..M) == path[0..t+1) ?
            bool eq = true;
#pragma unroll
            for (int k = 0; k <= t; ++k) eq &= (b[t - k] ^ 1u) == a[k];
            cand_rc |= eq;
        }
    }
    const lanemask fw = WAVE_MASK(cand_fw), rc = WAVE_MASK(cand_rc);
    if (lane == 0) {
        ovh[(size_t)it * 4] = (uint32_t)fw;
        ovh[(size_t)it * 4 + 1] = (uint32_t)(fw >> 32);
        ovh[(size_t)it * 4 + 2] = (uint32_t)rc;
        ovh[(size_t)it * 4 + 3] = (uint32_t)(rc >> 32);
    }
}

// k_scan3's chunks of a segment: seg_chunks, but never fewer than keep a wave at SCAN3_GROUPS
// groups of 64 items (the items that need the overhang test are remembered group by
// group and settled after the wave's last item)
constexpr int SCAN3_GROUPS = 8;
__host__ __device__ __forceinline__ uint32_t seg_chunks3(uint32_t n_items, unsigned long long mult,
                                                         unsigned long long inv_min)
{
    const uint32_t per = (uint32_t)(SCAN3_GROUPS * WAVE * (GFAL_SCAN2_THREADS / WAVE));
    const uint32_t lo = (n_items + per - 1u) / per;
    const uint32_t c = seg_chunks(n_items, mult, inv_min);
    return c < lo ? (lo > 65535u ? 65535u : lo) : c;
}

struct Scan3Cold {
    const uint16_t *item_steps;  // Items::steps
    const uint16_t *images;
    ImageLayout L;
    int n_paths;
    unsigned long long *worklist;
    unsigned long long *wl_count;
    uint32_t wl_capacity;
    uint32_t *wl_hist;
    uint32_t *status;
};

// One length segment as k_scan3 sees it.  Item `it` of the segment: its record at
// rec3[(r3_base + (it - item_lo) * R) * 64], R = 1 + ceil(m / 2) (+ 1: dedup weights) dwords
// per lane: the lane's content key, the byte offsets into the node masks of the nodes of
// its steps two to a dword (the last step twice when m is odd; padding lanes: the offset
// of the mask word that is always zero), the weight.
struct Seg3 {
    uint32_t item_lo, item_hi, m, n_chunks, step_base, r3_base;
};

struct Scan3Args {
    const uint32_t *rec3;        // item records
    const uint16_t *item_steps;  // Items::steps (the overhang test reads an alignment's steps)
    const uint32_t *ovh;         // [n_items][4] k_overhang's masks
    const uint32_t *common;      // Items::common
    const Scan3Cold *cold;
    const uint32_t *tile_masks;  // [tiles of the slab][v2p] node masks; word v2p - 2 stays zero
    const uint32_t *tile_hdr;    // [tiles of the slab][T3_THDR_WORDS]
    const uint32_t *list_count;  // [tiles of the slab][n_segs_total] entries of every window list
    const uint2 *list;           // [tiles of the slab][n_segs_total][stride] {content index, tile paths}
    uint32_t stride;
    int n_paths, tile, n_tiles, tile0, debug;
    int v2p;                     // mask words per tile
    int nm_shift;                // NMG: record offsets are node indices (2) or byte offsets (0)
    const Seg3 *segs;            // this launch's segments (<= MAX_SEGS): seg0 .. seg0 + n_segs of n_segs_total
    int n_segs, seg0, n_segs_total;
    unsigned long long chunk_mult, chunk_inv_min;
    uint32_t h_slots;            // table slots of a workgroup (power of two)
    uint32_t *counts;
    uint32_t *counts_dbg;        // the scorer's status words (diagnostic builds count into words 4..7)
    const uint32_t *ct_mult;     // alignments per content index (weights summed): the subpath pairs are counted from the lists
};

// GFAL_SCAN3_LG (default on): the good pairs that are subpaths are counted from the tile's
// lists -- {content, paths} x the content's multiplicity, once per (tile, length) -- and the
// item loop counts every pair that passes the filter as bad: the probe of the table is left
// to the few waves with a lane that can have a start overhang (its alignment touches the
// tile's first node).  0: every lane probes and counts its own good pairs (round 3's first form).
#ifndef GFAL_SCAN3_LG
#define GFAL_SCAN3_LG 1
#endif

typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
// table entry {key, mask} at an LDS byte address (one ds_read_b64)
__device__ __forceinline__ uint2 lds_entry(uint32_t byte_addr)
{
    const unsigned long long v = *(lds_cu64 *)(uintptr_t)byte_addr;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// a workgroup's pairs for the exact DP (LDS words): cursor, end of the valid entries, per-path
// histogram, then WL3_CAP entries of 8 bytes
constexpr int WL3_CUR = 0, WL3_END = 1, WL3_HIST = 2, WL3_BUF = 34, WL3_CAP = 512;
constexpr int WL3_WORDS = WL3_BUF + 2 * WL3_CAP;

struct Tile3 {
    uint32_t *wl_lds;            // [WL3_WORDS]
    uint32_t *tri_lds;           // [waves][SCAN3_GROUPS][2] per wave and group of 64 items: those that need the overhang test
    const char *gmask;           // NMG: the tile's node masks in HBM
    int nm_shift;
    uint32_t tab_base;           // LDS byte address of the table
    uint32_t h_mask;
    uint32_t sub_mask, gt_mask, half;   // half: popcount(sub_mask) / 2
    int tile_paths, path0;
};

template <bool NMG>
__device__ __forceinline__ uint32_t nm_read(const Tile3 &tv, uint32_t off)
{
    if constexpr (NMG) return *reinterpret_cast<const uint32_t *>(tv.gmask + ((size_t)off << tv.nm_shift));
    else return *(lds_cu32 *)(uintptr_t)off;     // the node masks start at LDS address 0
}

// OR / sum over the wave, all in registers (DPP inside a row of 16, four readlanes across)
__device__ __forceinline__ uint32_t wave_or_dpp(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);    // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ uint32_t wave_add_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// Counters of one pass.  Nearly every alignment that passes the filter is good for ALL
// paths of the pass or bad for all of them (tile paths are prefixes / siblings of one
// another).  A lane's alignment counts as "good for all" when it is good for more than
// half of the paths -- one wave-wide popcount per item -- and the paths where that is
// wrong are exceptions, settled bit by bit over the union of the wave's exception bits
// (few: a substituted step, a path that ends earlier).
template <bool W>
struct Acc3 {
    uint32_t good = 0, bad = 0;          // lane p: totals of tile path p (may go through negative partial sums)
    uint32_t fg = 0, fb = 0;             // (uniform, or per lane with weights) good / bad for all paths of the pass
    __device__ __forceinline__ void add(uint32_t good_mask, uint32_t bad_mask, const Tile3 &tv, uint32_t w, int lane)
    {
        const bool isg = (uint32_t)__builtin_popcount(good_mask) > tv.half;
        const bool isb = (uint32_t)__builtin_popcount(bad_mask) > tv.half;
        const lanemask mg = WAVE_MASK(isg), mb = WAVE_MASK(isb);
        if constexpr (W) {
            fg += isg ? w : 0u;
            fb += isb ? w : 0u;
        } else {
            fg += (uint32_t)__popcll(mg);
            fb += (uint32_t)__popcll(mb);
        }
        const uint32_t xg = good_mask ^ (isg ? tv.sub_mask : 0u), xb = bad_mask ^ (isb ? tv.sub_mask : 0u);
        lanemask any_x = WAVE_MASK((xg | xb) != 0u);
        asm volatile("" : "+s"(any_x));          // (keeps the reduction below out of the common path)
        if (__builtin_expect(any_x != 0ull, 0)) {
            uint32_t u = wave_or_dpp(xg | xb);
            while (u) {
                const int p = __builtin_ctz(u);
                u &= u - 1u;
                const bool eg = ((xg >> p) & 1u) != 0u, eb = ((xb >> p) & 1u) != 0u;
                uint32_t dg, db;
                if constexpr (W) {
                    dg = wave_add_dpp(eg ? (isg ? 0u - w : w) : 0u);
                    db = wave_add_dpp(eb ? (isb ? 0u - w : w) : 0u);
                } else {
                    const lanemask g = WAVE_MASK(eg), b = WAVE_MASK(eb);
                    dg = (uint32_t)__popcll(g & ~mg) - (uint32_t)__popcll(g & mg);
                    db = (uint32_t)__popcll(b & ~mb) - (uint32_t)__popcll(b & mb);
                }
                if (lane == p) {
                    good += dg;
                    bad += db;
                }
            }
        }
    }
    // the same for pairs that are all bad (no tile path is shorter than the alignments: the
    // usual case when the subpath pairs are counted from the lists)
    __device__ __forceinline__ void add_bad(uint32_t bad_mask, const Tile3 &tv, uint32_t w, int lane)
    {
        const bool isb = (uint32_t)__builtin_popcount(bad_mask) > tv.half;
        const lanemask mb = WAVE_MASK(isb);
        if constexpr (W) fb += isb ? w : 0u;
        else fb += (uint32_t)__popcll(mb);
        const uint32_t xb = bad_mask ^ (isb ? tv.sub_mask : 0u);
        lanemask any_x = WAVE_MASK(xb != 0u);
        asm volatile("" : "+s"(any_x));
        if (__builtin_expect(any_x != 0ull, 0)) {
            uint32_t u = wave_or_dpp(xb);
            while (u) {
                const int p = __builtin_ctz(u);
                u &= u - 1u;
                const bool eb = ((xb >> p) & 1u) != 0u;
                uint32_t db;
                if constexpr (W) {
                    db = wave_add_dpp(eb ? (isb ? 0u - w : w) : 0u);
                } else {
                    const lanemask b = WAVE_MASK(eb);
                    db = (uint32_t)__popcll(b & ~mb) - (uint32_t)__popcll(b & mb);
                }
                if (lane == p) bad += db;
            }
        }
    }
    // pairs counted as bad that the overhang test hands to the exact DP after all
    __device__ __forceinline__ void uncount_bad(uint32_t cand, uint32_t w, int lane)
    {
        for (uint32_t u = wave_or_dpp(cand); u; u &= u - 1u) {
            const int p = __builtin_ctz(u);
            const bool c = ((cand >> p) & 1u) != 0u;
            uint32_t d;
            if constexpr (W) d = wave_add_dpp(c ? w : 0u);
            else d = (uint32_t)__popcll(WAVE_MASK(c));
            if (lane == p) bad -= d;
        }
    }
    __device__ __forceinline__ void finish(const Tile3 &tv, int lane)
    {
        uint32_t sg = fg, sb = fb;
        if constexpr (W) {
            sg = wave_add_dpp(fg);
            sb = wave_add_dpp(fb);
        }
        if ((tv.sub_mask >> lane) & 1u) {
            good += sg;
            bad += sb;
        }
        fg = fb = 0;
    }
};

template <int P0, bool W>
struct Item3Regs {
    uint32_t key, w;
    uint32_t np[P0 > 0 ? P0 : 1];
    int src;                     // (uniform) the item's rank in its group of 64
};

// The rare part of an item: lanes whose alignment is not a subpath of some tile path but
// touches the tile's first node.  The traceback stays free only if a proper suffix of B
// (or of rc(B)) equals a prefix of the path; survivors of this exact test go to the DP
// kernels.  The alignment's steps are loaded in one go (P0 > 0: at most 2 P0 of them) and
// compared in registers with the path's first steps from the tile header (uniform:
// scalar loads); tile paths that share their first M - 1 steps with the tile's first
// path (nearly always all of them) are decided together.
template <int P0>
__device__ __forceinline__ uint32_t scan3_triage(const Scan3Args &a, const Tile3 &tv, const Seg3 &sg,
                                                 uint32_t it, uint32_t open, bool has_a0, int tile_rel, int lane)
{
    const int M = (int)sg.m;
    const uint32_t *thdr = a.tile_hdr + (size_t)tile_rel * T3_THDR_WORDS;
    uint32_t cfw = 0, crc = 0;
    const uint16_t *bp = a.item_steps + ((size_t)sg.step_base + (size_t)(it - sg.item_lo) * (uint32_t)M) * WAVE + lane;
    constexpr int NB = P0 > 0 ? 2 * P0 : 1;
    uint32_t b[NB];
    if constexpr (P0 > 0) {
#pragma unroll
        for (int t = 0; t < NB; ++t) b[t] = t < M ? (uint32_t)bp[t * WAVE] : 0xFFFFu;
    }
    uint32_t want = wave_or_dpp(has_a0 ? open : 0u);
    uint32_t my_lcp = 0;
    if (lane < tv.tile_paths) my_lcp = thdr[64 + lane];
    const uint32_t class0 = (uint32_t)WAVE_MASK(lane < tv.tile_paths && (int)my_lcp >= M - 1);
    if (M <= 16 && (want & class0) != 0u) {          // k_overhang has the answer for these paths
        const uint32_t *o = a.ovh + (size_t)it * 4;
        const lanemask fwm = (lanemask)o[0] | ((lanemask)o[1] << 32), rcm = (lanemask)o[2] | ((lanemask)o[3] << 32);
        const uint32_t mine_bits = has_a0 ? (open & class0) : 0u;
        cfw |= ((fwm >> lane) & 1ull) ? mine_bits : 0u;
        crc |= ((rcm >> lane) & 1ull) ? mine_bits : 0u;
        want &= ~class0;
    }
    while (want) {                                   // (paths that start differently: one by one)
        const int p = __builtin_ctz(want);
        const uint32_t grp = 1u << p;
        want &= ~grp;
        const uint32_t mine_bits = has_a0 ? (open & grp) : 0u;
        const bool mine = mine_bits != 0u;
        bool cand_fw = false, cand_rc = false;
        if constexpr (P0 > 0) {
            const uint32_t *pre = thdr + T3_THDR_PRE + p * 8;        // (uniform)
            auto stepA = [&](int k) -> uint32_t { return (pre[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu; };
            const uint32_t a0 = stepA(0);
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                if (t >= 1 && t < M) {                     // B[t..M) == path[0..M-t) ?
                    bool eq = mine && b[t] == a0;
#pragma unroll
                    for (int k = 1; k < NB - t; ++k)
                        if (t + k < M) eq &= b[t + k] == stepA(k);
                    cand_fw |= eq;
                }
                if (t < M - 1) {                           // rc(B)[M-1-t..M) == path[0..t+1) ?
                    bool eq = mine && (b[t] ^ 1u) == a0;
#pragma unroll
                    for (int k = 1; k <= t; ++k) eq &= (b[t - k] ^ 1u) == stepA(k);
                    cand_rc |= eq;
                }
            }
        } else {
            const Scan3Cold c = *a.cold;
            const uint32_t a0 = thdr[32 + p];
            const uint32_t *pstep32 = reinterpret_cast<const uint32_t *>(
                c.images + (size_t)(tv.path0 + p) * c.L.total + c.L.step_at());
            for (int t = 0; t < M; ++t) {
                const uint32_t bt = bp[t * WAVE];
                const bool live_fw = mine && t >= 1 && bt == a0;
                if (WAVE_ANY(live_fw)) cand_fw |= tail_equals(bp, t, 1, M - t, 0u, pstep32, live_fw);
                const bool live_rc = mine && t < M - 1 && (bt ^ 1u) == a0;
                if (WAVE_ANY(live_rc)) cand_rc |= tail_equals(bp, t, -1, t + 1, 1u, pstep32, live_rc);
            }
        }
        cfw |= cand_fw ? mine_bits : 0u;
        crc |= cand_rc ? mine_bits : 0u;
    }
    // The pairs for the exact DP are gathered per workgroup (LDS) and leave it with ONE
    // atomic on the list's cursor: every returning atomic on that one word costs a round
    // trip to L2 and they serialise chip-wide (67 k items take the test per config-3 step:
    // 0.29 of 0.63 ms went there).  What does not fit the buffer goes the direct way.
    const uint32_t any = cfw | crc;
    const uint32_t uany = wave_or_dpp(any);          // (uniform) the tile paths with a pair
    if (uany) {
        uint32_t total = 0;
        for (uint32_t u = uany; u; u &= u - 1u)
            total += (uint32_t)__popcll(WAVE_MASK(((any >> __builtin_ctz(u)) & 1u) != 0u));
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&tv.wl_lds[WL3_CUR], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base + total <= (uint32_t)WL3_CAP) {
            unsigned long long *buf = reinterpret_cast<unsigned long long *>(tv.wl_lds + WL3_BUF);
            for (uint32_t u = uany; u; u &= u - 1u) {
                const int p = __builtin_ctz(u);
                const bool mine = ((any >> p) & 1u) != 0u;
                const lanemask m = WAVE_MASK(mine);
                if (lane == 0) atomicAdd(&tv.wl_lds[WL3_HIST + p], (uint32_t)__popcll(m));
                if (mine)
                    buf[base + lanes_below(m, lane)] =
                        (((cfw >> p) & 1u) ? WL_FW : 0ull) | (((crc >> p) & 1u) ? WL_RC : 0ull) |
                        ((unsigned long long)(tv.path0 + p) << 32) | (it * WAVE + (uint32_t)lane);
                base += (uint32_t)__popcll(m);
            }
        } else {
            if (lane == 0) atomicMin(&tv.wl_lds[WL3_END], base);
            const Scan3Cold c = *a.cold;
            push_item_pairs(c.worklist, c.wl_count, c.wl_capacity, c.wl_hist, c.status, c.n_paths, cfw, crc, lane,
                            (uint32_t)tv.path0, tv.tile_paths, it * WAVE + (uint32_t)lane, M);
        }
    }
    return any;
}

// One item against the tile, from the lane's registers: key, pm = AND of the node masks of
// the lane's alignment (bits 0..30: every node is on tile path p -- the filter of
// src/eval.cpp:81-91; bit 31: none is the tile's first node), e = the table entry of the
// key's home slot.  Returns the tile paths the alignment passes the filter for but is no
// subpath of (`open`), good_mask = the paths it is good for without further tests.
__device__ __forceinline__ uint32_t scan3_resolve(const Tile3 &tv, uint32_t key, uint32_t pm, uint2 e,
                                                  uint32_t &good_mask)
{
    const uint32_t pass = pm & tv.sub_mask;
    const uint32_t todo = pass & ~tv.gt_mask;       // (src/alignments.cpp:500: m > n -> good)
    // ---- which of the tile's paths contain exactly this step sequence ----
#if !defined(GFAL_ABLATE3) || (GFAL_ABLATE3 != 5 && GFAL_ABLATE3 != 6)      // (timing probe: home slot only)
    if (__builtin_expect(WAVE_ANY(todo != 0u && e.x != key && e.x != KEY_EMPTY), 0)) {      // (the home slot is taken)
        uint32_t slot = key & tv.h_mask;
        bool more = todo != 0u && e.x != key && e.x != KEY_EMPTY;
        while (WAVE_ANY(more)) {
            slot = (slot + 1u) & tv.h_mask;
            const uint2 e2 = lds_entry(tv.tab_base + (slot << 3));
            e.x = more ? e2.x : e.x;
            e.y = more ? e2.y : e.y;
            more = more && e.x != key && e.x != KEY_EMPTY;
        }
    }
#endif
    const uint32_t have = (e.x == key ? e.y : 0u) | tv.gt_mask;
    good_mask = pass & have;
    return pass & ~have;
}

// The hot path of an item.  Lanes whose alignment is open for some path AND touches the
// tile's first node need the overhang test: the item is only MARKED here (bit `src` of
// `tri`) and its open pairs are counted as bad for now; scan3_fixup settles them after
// the group's items -- the test's registers stay out of this loop (with the test inline
// the loop spilled 28 VGPRs and ~150 SGPRs and took twice as long).
template <bool W>
__device__ __forceinline__ void scan3_decide(const Tile3 &tv, uint32_t key, uint32_t w, uint32_t pm, uint2 e,
                                             int src, lanemask &tri, int lane, Acc3<W> &acc)
{
    uint32_t good_mask;
    const uint32_t open = scan3_resolve(tv, key, pm, e, good_mask);
#if !defined(GFAL_ABLATE3) || (GFAL_ABLATE3 != 4 && GFAL_ABLATE3 != 6)      // (timing probe: no overhang test)
    if (WAVE_ANY(open != 0u && (int32_t)pm >= 0)) tri |= 1ull << src;
#endif
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 3      // timing probe: no counting
    acc.good += good_mask ^ open;
    return;
#endif
#if GFAL_SCAN3_LG
    // (the subpath pairs among `pass` are taken back out of `bad` by the list count)
    const uint32_t pass = pm & tv.sub_mask;
    acc.add(pass & tv.gt_mask, pass & ~tv.gt_mask, tv, w, lane);
#else
    acc.add(good_mask, open, tv, w, lane);
#endif
}

#if GFAL_SCAN3_LG
// The hot path of an item: filter and count; the table is asked only when a lane can have a
// start overhang at all.
template <bool W>
__device__ __forceinline__ void scan3_count(const Tile3 &tv, uint32_t key, uint32_t w, uint32_t pm,
                                            int src, lanemask &tri, int lane, Acc3<W> &acc)
{
    const uint32_t pass = pm & tv.sub_mask;
    const uint32_t todo = pass & ~tv.gt_mask;
    if (__builtin_expect(WAVE_ANY(todo != 0u && (int32_t)pm >= 0), 0)) {
        const uint2 e = lds_entry(tv.tab_base + ((key & tv.h_mask) << 3));
        scan3_decide<W>(tv, key, w, pm, e, src, tri, lane, acc);
        return;
    }
    if (tv.gt_mask == 0u) acc.add_bad(todo, tv, w, lane);
    else acc.add(pass & tv.gt_mask, todo, tv, w, lane);
}
#endif

template <int P0, bool W, bool NMG>
__device__ __forceinline__ void scan3_item(const Tile3 &tv, const Item3Regs<P0, W> &r, lanemask &tri, int lane,
                                           Acc3<W> &acc)
{
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 1      // timing probe: the loads only
    uint32_t x = r.key;
#pragma unroll
    for (int k = 0; k < P0; ++k) x ^= r.np[k];
    acc.good += x;
    return;
#endif
#if GFAL_SCAN3_LG
    uint32_t pm = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < P0; ++k) {
        pm &= nm_read<NMG>(tv, r.np[k] & 0xFFFFu);
        pm &= nm_read<NMG>(tv, r.np[k] >> 16);
    }
    scan3_count<W>(tv, r.key, r.w, pm, r.src, tri, lane, acc);
#else
    // the probe goes out together with the node-mask reads: one LDS round trip
    const uint2 e = lds_entry(tv.tab_base + ((r.key & tv.h_mask) << 3));
    uint32_t pm = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < P0; ++k) {
        pm &= nm_read<NMG>(tv, r.np[k] & 0xFFFFu);
        pm &= nm_read<NMG>(tv, r.np[k] >> 16);
    }
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 2      // timing probe: loads + LDS reads
    acc.good += pm ^ e.x ^ e.y;
    return;
#endif
    scan3_decide<W>(tv, r.key, r.w, pm, e, r.src, tri, lane, acc);
#endif
}

// A marked item again, now with the overhang test: the pairs it hands to the exact DP
// were counted as bad by the hot path.
template <int P0, bool W, bool NMG>
__device__ __forceinline__ void scan3_fixup(const Scan3Args &a, const Tile3 &tv, const Seg3 &sg, uint32_t it,
                                            int tile_rel, int lane, Acc3<W> &acc)
{
    const int P0rt = ((int)sg.m + 1) / 2;
    const uint32_t *pp = a.rec3 + ((size_t)sg.r3_base + (size_t)(it - sg.item_lo) * (1 + P0rt + (W ? 1 : 0))) * WAVE + lane;
    const uint32_t key = pp[0];
    const uint32_t w = W ? pp[(size_t)(P0rt + 1) * WAVE] : 1u;
    const uint2 e = lds_entry(tv.tab_base + ((key & tv.h_mask) << 3));
    uint32_t pm = 0xFFFFFFFFu;
    for (int k = 0; k < P0rt; ++k) {
        const uint32_t x = pp[(size_t)(k + 1) * WAVE];
        pm &= nm_read<NMG>(tv, x & 0xFFFFu);
        pm &= nm_read<NMG>(tv, x >> 16);
    }
    uint32_t good_mask;
    const uint32_t open = scan3_resolve(tv, key, pm, e, good_mask);
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 9      // timing probe: the item is re-derived, the test is skipped
    acc.good += open ^ good_mask;
    return;
#endif
    const uint32_t cand = scan3_triage<P0>(a, tv, sg, it, open, (int32_t)pm >= 0, tile_rel, lane);
    acc.uncount_bad(cand, w, lane);
}

// The wave's items of one segment chunk, 64 at a time: one ballot drops the items none
// of whose lanes can pass the filter.  P0 = ceil(M / 2) offset dwords per lane in registers
// (two sets, loaded one item ahead), or 0: any length, loaded where it is used.
template <int P0, bool W, bool NMG>
__device__ __forceinline__ void scan3_items(const Scan3Args &a, const Tile3 &tv, const Seg3 &sg,
                                            int chunk, int wave, int tile_rel, int lane, Acc3<W> &acc)
{
    const int n_chunks = (int)sg.n_chunks;
    const int item_stride = SCAN2_WAVES * n_chunks;
    const int P0rt = ((int)sg.m + 1) / 2;
    const int R = 1 + P0rt + (W ? 1 : 0);
    const uint32_t ulane = (uint32_t)lane;
    int grp = 0;
    for (int it0 = (int)sg.item_lo + chunk + wave * n_chunks; it0 < (int)sg.item_hi;
         it0 += WAVE * item_stride, ++grp) {
        const int my_it = it0 + lane * item_stride;
        bool keep = my_it < (int)sg.item_hi;
        if (keep) {
            const uint32_t common = a.common[my_it];
            if (common != NO_COMMON_NODE) {
                const uint32_t m1 = nm_read<NMG>(tv, (common & 0x7FFFu) << (NMG ? 2 - tv.nm_shift : 2)),
                               m2 = nm_read<NMG>(tv, (common >> 16) << (NMG ? 2 - tv.nm_shift : 2));
                keep = (((common & COMMON_EITHER) ? (m1 | m2) : (m1 & m2)) & tv.sub_mask) != 0u;
            }
        }
        lanemask todo = WAVE_MASK(keep);
        if (todo == 0) continue;
        lanemask tri = 0;                // items of this group that need the overhang test
        if constexpr (P0 > 0) {
            // the record of item `src` of this group: the group's first record + src * a fixed
            // stride (one scalar multiply-add per item instead of the address from scratch: the
            // scalar unit is as busy as the VALUs in this loop)
            constexpr uint32_t RW = (uint32_t)(P0 + 1 + (W ? 1 : 0));
            const uint32_t *gbase = a.rec3 + ((size_t)sg.r3_base + (size_t)(it0 - (int)sg.item_lo) * RW) * WAVE;
            asm volatile("" : "+s"(gbase));
            const uint32_t gstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)item_stride * RW * WAVE * 4u));      // bytes
            // (the rank of the mask's first item; an empty mask -- one item too far ahead --
            // loads the group's first record again: s_ff1 gives -1 for it)
            auto load_item = [&](lanemask m, Item3Regs<P0, W> &r) {
                int src;
                asm("s_ff1_i32_b64 %0, %1" : "=s"(src) : "s"(m));
                src = max(src, 0);
                r.src = src;
                const GLOBAL_AS uint32_t *pp =
                    sgpr_ptr(reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(gbase) + (uint32_t)src * gstep)) + ulane;
                r.key = pp[0];
#pragma unroll
                for (int k = 0; k < P0; ++k) r.np[k] = pp[(k + 1) * WAVE];
                r.w = W ? pp[(P0 + 1) * WAVE] : 1u;
            };
            Item3Regs<P0, W> ra, rb;
            load_item(todo, ra);
            while (true) {
                const lanemask rest = todo & (todo - 1);
                load_item(rest, rb);
                scan3_item<P0, W, NMG>(tv, ra, tri, lane, acc);
                if (rest == 0) break;
                todo = rest & (rest - 1);
                load_item(todo, ra);
                scan3_item<P0, W, NMG>(tv, rb, tri, lane, acc);
                if (todo == 0) break;
            }
        } else {      // the longer alignments are rare: no second register set for them
            for (; todo != 0; todo &= todo - 1) {
                const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                const uint32_t it = (uint32_t)(it0 + src * item_stride);
                const uint32_t *pp = a.rec3 + ((size_t)sg.r3_base + (size_t)(it - sg.item_lo) * R) * WAVE + ulane;
                const uint32_t key = pp[0];
                const uint32_t w = W ? pp[(size_t)(P0rt + 1) * WAVE] : 1u;
#if !GFAL_SCAN3_LG
                const uint2 e = lds_entry(tv.tab_base + ((key & tv.h_mask) << 3));
#endif
                uint32_t pm = 0xFFFFFFFFu;
                for (int k = 0; k < P0rt; ++k) {
                    const uint32_t x = pp[(size_t)(k + 1) * WAVE];
                    pm &= nm_read<NMG>(tv, x & 0xFFFFu);
                    pm &= nm_read<NMG>(tv, x >> 16);
                }
#if GFAL_SCAN3_LG
                scan3_count<W>(tv, key, w, pm, src, tri, lane, acc);
#else
                scan3_decide<W>(tv, key, w, pm, e, src, tri, lane, acc);
#endif
            }
        }
        if (tri != 0 && lane == 0) {      // (settled by scan3_fixups when the wave's items are done)
            tv.tri_lds[(wave * SCAN3_GROUPS + grp) * 2] = (uint32_t)tri;
            tv.tri_lds[(wave * SCAN3_GROUPS + grp) * 2 + 1] = (uint32_t)(tri >> 32);
        }
    }
}

// the marked items of a wave (Tile3::tri_lds), one after the other
template <bool W, bool NMG>
__device__ __forceinline__ void scan3_fixups(const Scan3Args &a, const Tile3 &tv, const Seg3 &sg, int chunk,
                                             int wave, int tile_rel, int lane, Acc3<W> &acc)
{
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 8      // timing probe: items are marked, nothing is settled
    return;
#endif
    const int n_chunks = (int)sg.n_chunks;
    const int item_stride = SCAN2_WAVES * n_chunks;
    for (int grp = 0; grp < SCAN3_GROUPS; ++grp) {
        lanemask tri = (lanemask)tv.tri_lds[(wave * SCAN3_GROUPS + grp) * 2] |
                       ((lanemask)tv.tri_lds[(wave * SCAN3_GROUPS + grp) * 2 + 1] << 32);
        tri = ((lanemask)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tri)) |
              ((lanemask)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tri >> 32)) << 32);
        const int it0 = (int)sg.item_lo + chunk + wave * n_chunks + grp * WAVE * item_stride;
        for (; tri != 0; tri &= tri - 1) {
            const int src = __builtin_ctzll(tri);
            scan3_fixup<0, W, NMG>(a, tv, sg, (uint32_t)(it0 + src * item_stride), tile_rel, lane, acc);
        }
    }
}

template <bool W, bool NMG>
__global__ __launch_bounds__(SCAN2_THREADS, SCAN2_WAVES_PER_SIMD) void k_scan3(Scan3Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_rel = blockIdx.x % a.n_tiles;
    // (segment, chunk) <- the workgroup's row, as in k_scan2
    Seg3 sg;
    int chunk, seg;
    {
        const int y = blockIdx.x / a.n_tiles;
        Seg3 mine{0u, 0u, 0u, 0u, 0u, 0u};
        if (lane < a.n_segs) mine = a.segs[lane];
        const uint32_t c = lane < a.n_segs
                               ? seg_chunks3(mine.item_hi - mine.item_lo, a.chunk_mult, a.chunk_inv_min)
                               : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, WAVE);
            if (lane >= o) incl += v;
        }
        const lanemask beyond = WAVE_MASK(incl > (uint32_t)y);
        seg = beyond ? __builtin_ctzll(beyond) : 0;
        sg.item_lo = (uint32_t)__builtin_amdgcn_readlane((int)mine.item_lo, seg);
        sg.item_hi = (uint32_t)__builtin_amdgcn_readlane((int)mine.item_hi, seg);
        sg.m = (uint32_t)__builtin_amdgcn_readlane((int)mine.m, seg);
        sg.n_chunks = (uint32_t)__builtin_amdgcn_readlane((int)c, seg);
        sg.step_base = (uint32_t)__builtin_amdgcn_readlane((int)mine.step_base, seg);
        sg.r3_base = (uint32_t)__builtin_amdgcn_readlane((int)mine.r3_base, seg);
        chunk = y - (int)__builtin_amdgcn_readlane((int)(incl - c), seg);
    }
    const int M = (int)sg.m;
    const int v2p = a.v2p;
    Tile3 tv;
    tv.path0 = (a.tile0 + tile_rel) * a.tile;
    tv.tile_paths = min(a.tile, a.n_paths - tv.path0);
    tv.nm_shift = a.nm_shift;
    // LDS: node masks first (address = the record's byte offset: no base to add), then the table
    uint32_t *tab = lds32 + (NMG ? 0 : v2p);
    uint32_t *misc = tab + 2 * a.h_slots;    // [64]
    tv.wl_lds = misc + 2 * MAX_TILE;
    tv.tri_lds = tv.wl_lds + WL3_WORDS;
    if (tid < WL3_BUF) tv.wl_lds[tid] = tid == WL3_END ? 0xFFFFFFFFu : 0u;
    const uint32_t *gmask = a.tile_masks + (size_t)tile_rel * v2p;
    tv.gmask = reinterpret_cast<const char *>(gmask);
    tv.tab_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tab;
    if constexpr (!NMG) {
        // straight from global memory into LDS (lane l of a wave-instruction lands at its
        // uniform LDS base + 4 l)
        for (int o = wave * WAVE; o < v2p; o += SCAN2_THREADS)
            if (o + lane < v2p)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(gmask + o + lane),
                    (__attribute__((address_space(3))) void *)(lds32 + o), 4, 0, 0);
    }
    int hdr_n = 0;
    if (lane < tv.tile_paths) hdr_n = (int)a.tile_hdr[(size_t)tile_rel * T3_THDR_WORDS + lane];
    const uint32_t count = a.list_count[(size_t)tile_rel * a.n_segs_total + a.seg0 + seg];
    const uint2 *list = a.list + ((size_t)tile_rel * a.n_segs_total + a.seg0 + seg) * a.stride;
    // the table: as many slots as four times the list's entries (a short tile, a rare length:
    // less to clear and to walk), at most what the launch has room for
    uint32_t h_slots = 1024;
    while (h_slots < 4u * count && h_slots < a.h_slots) h_slots <<= 1;
    tv.h_mask = h_slots - 1u;
    const uint32_t cap = h_slots / 2;

    uint32_t cnt_good = 0, cnt_bad = 0;      // lane p: totals of tile path p over all passes
    // Passes over the tile's paths: usually one.  The window list of unrelated paths may
    // hold more distinct contents than the table takes at load 1/2: then the range of
    // paths is halved until it fits (one path alone always does: < 2000 windows).
    int t0 = 0, t1 = tv.tile_paths;
    while (t0 < tv.tile_paths) {
        const uint32_t range = ((1u << t1) - 1u) & ~((1u << t0) - 1u);
        __syncthreads();                     // (the previous pass's probes are done)
        {
            uint4 *t4 = reinterpret_cast<uint4 *>(tab);
            for (uint32_t i = tid; i < h_slots / 2; i += SCAN2_THREADS)
                t4[i] = make_uint4(KEY_EMPTY, 0u, KEY_EMPTY, 0u);
            if (tid < 2) misc[tid] = 0;      // [0] entries [1] overflow
            if (tid < SCAN2_WAVES * SCAN3_GROUPS * 2) tv.tri_lds[tid] = 0;
        }
        __syncthreads();
        for (uint32_t i = tid; i < count; i += SCAN2_THREADS) {
            if (*(volatile uint32_t *)&misc[1]) break;
            const uint2 en = list[i];
            const uint32_t m = en.y & range;
            if (m == 0u) continue;
            uint32_t slot = en.x & tv.h_mask;
            while (true) {                   // (identical contents of several windows: one entry)
                const uint32_t old = atomicCAS(&tab[2u * slot], KEY_EMPTY, en.x);
                if (old == KEY_EMPTY) {
                    if (atomicAdd(&misc[0], 1u) + 1u > cap) misc[1] = 1u;
                    break;
                }
                if (old == en.x) break;
                slot = (slot + 1u) & tv.h_mask;
            }
            atomicOr(&tab[2u * slot + 1u], m);
        }
        if constexpr (!NMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the node masks have landed
        __syncthreads();
        if (misc[1] != 0u && t1 - t0 > 1) {
            t1 = t0 + (t1 - t0) / 2;
            continue;
        }
        tv.sub_mask = range;
        tv.half = (uint32_t)__builtin_popcount(tv.sub_mask) / 2u;
        tv.gt_mask = (uint32_t)WAVE_MASK(lane >= t0 && lane < t1 && hdr_n < M);
#if GFAL_SCAN3_LG
        // The subpath pairs of this (tile, length), counted from the table: every entry is one
        // content with the paths that contain it (windows merged), and ct_mult alignments carry
        // it.  The item loops of the segment's chunks count those pairs as bad; every chunk
        // settles the contents a hash of the index assigns to it (the same in every workgroup,
        // whatever slot the entry took there).  Lane p adds up path p: an entry's mask and
        // multiplicity are broadcast from the lane that read it -- no atomics.  (Paths shorter
        // than the alignments have no windows; their passing pairs are good, src/alignments.cpp:500.)
        {
            uint32_t lg = 0, all_sum = 0;
            const uint32_t n_chunks = sg.n_chunks;
            const uint32_t all = range & ~tv.gt_mask;       // (most windows are shared by every path of the tile)
            for (uint32_t sl0 = (uint32_t)wave * WAVE; sl0 < h_slots; sl0 += SCAN2_THREADS) {
                const uint32_t sl = sl0 + (uint32_t)lane;
                const uint2 e = lds_entry(tv.tab_base + (sl << 3));
                uint32_t bits = 0, mult = 0;
                if (e.x != KEY_EMPTY && ((((e.x * 0x9E3779B1u) >> 16) * n_chunks) >> 16) == (uint32_t)chunk) {
                    bits = e.y & all;
                    if (bits) mult = a.ct_mult[e.x];
                    if (bits == all) {       // every path: summed per lane, spread once at the end
                        all_sum += mult;
                        bits = 0;
                    }
                }
                for (lanemask todo = WAVE_MASK(bits != 0u); todo; todo &= todo - 1ull) {
                    const int src = __builtin_ctzll(todo);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bits, src);
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mult, src);
                    lg += ((b >> lane) & 1u) ? m : 0u;
                }
            }
            all_sum = wave_add_dpp(all_sum);
            lg += ((all >> lane) & 1u) ? all_sum : 0u;
            cnt_good += lg;
            cnt_bad -= lg;
        }
#endif
        Seg3 sgl = sg;
        if (a.debug == 1) sgl.item_hi = sgl.item_lo;      // timing probe: no items
        Acc3<W> acc;
#define GFAL_RUN3(PP) scan3_items<PP, W, NMG>(a, tv, sgl, chunk, wave, tile_rel, lane, acc)
        switch ((M + 1) / 2) {
        case 1: GFAL_RUN3(1); break;
        case 2: GFAL_RUN3(2); break;
        case 3: GFAL_RUN3(3); break;
        case 4: GFAL_RUN3(4); break;
        case 5: GFAL_RUN3(5); break;
        case 6: GFAL_RUN3(6); break;
        default: GFAL_RUN3(0); break;
        }
#undef GFAL_RUN3
        scan3_fixups<W, NMG>(a, tv, sgl, chunk, wave, tile_rel, lane, acc);
        acc.finish(tv, lane);
        cnt_good += acc.good;
        cnt_bad += acc.bad;
        t0 = t1;
        t1 = tv.tile_paths;
    }
    // workgroup reduction through LDS (the table is dead now), then one atomic per
    // counter per workgroup
    __syncthreads();
    if (tid < 2 * MAX_TILE) misc[tid] = 0;
    __syncthreads();
    if (lane < tv.tile_paths) {
        if (cnt_bad) atomicAdd(&misc[lane], cnt_bad);
        if (cnt_good) atomicAdd(&misc[MAX_TILE + lane], cnt_good);
    }
    // the workgroup's pairs for the exact DP: one atomic on the list's cursor
    const uint32_t n_wl = min(tv.wl_lds[WL3_CUR], tv.wl_lds[WL3_END]);
    __syncthreads();
    if (n_wl) {
        const Scan3Cold c = *a.cold;
        if (tid == 0) {
            const unsigned long long g = atomicAdd(c.wl_count, (unsigned long long)n_wl);
            tv.wl_lds[WL3_CUR] = (uint32_t)g;
            tv.wl_lds[WL3_END] = (uint32_t)(g >> 32);
        }
        __syncthreads();
        const unsigned long long g = (unsigned long long)tv.wl_lds[WL3_CUR] | ((unsigned long long)tv.wl_lds[WL3_END] << 32);
        const unsigned long long *buf = reinterpret_cast<const unsigned long long *>(tv.wl_lds + WL3_BUF);
        bool overflow = false;
        for (uint32_t i = tid; i < n_wl; i += SCAN2_THREADS) {
            if (g + i < c.wl_capacity) c.worklist[g + i] = buf[i];
            else overflow = true;
        }
        if (overflow) atomicOr(c.status, ST_DP_OVERFLOW);
        if (tid < tv.tile_paths) {
            const uint32_t h = tv.wl_lds[WL3_HIST + tid];
            if (h) atomicAdd(&c.wl_hist[(uint32_t)length_class(M) * (uint32_t)c.n_paths + tv.path0 + tid], h);
        }
    }
    if (tid < tv.tile_paths) {
        const uint32_t d = misc[tid], g = misc[MAX_TILE + tid];
        if (d) atomicAdd(&a.counts[tv.path0 + tid], d);
        if (g) atomicAdd(&a.counts[a.n_paths + tv.path0 + tid], g);
    }
}
}  // namespace

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
struct gfal_scorer {
    int device = 0;
    int n_cus = 256;
    int64_t n_aln = 0, n_steps = 0;      // this shard's alignments and their steps
    int64_t n_aln_in = 0;                 // alignments given to create (all shards)
    std::vector<uint8_t> owned;           // [n_aln_in] 1 = scored by this shard
    int32_t n_nodes = 0, n_local = 0, max_aln_len = 0;
    uint32_t n_empty = 0;
    int n_items = 0;
    int64_t n_item_u16 = 0;

    // device, owned
    int32_t *d_node_local = nullptr;   // [n_nodes] -> local id or -1
    uint32_t *d_node_hist = nullptr;   // [n_local] alignment steps per node
    uint16_t *d_item_steps = nullptr;
    uint32_t *d_item_base = nullptr;
    uint16_t *d_item_len = nullptr;
    uint32_t *d_item_pairs = nullptr;
    uint32_t *d_item_pbase = nullptr;
    uint32_t *d_item_common = nullptr;   // Items::common
    uint4 *d_item_hdr = nullptr;         // Items::hdr
    uint32_t *d_item_weight = nullptr;   // Items::weight (dedup scorers)
    uint32_t *d_item_hash = nullptr;     // [n_items * 64] whash of every lane (k_scan2)
    uint32_t *d_item_pairs0 = nullptr;   // Scan2Args::pairs0
    uint32_t *d_rec3 = nullptr;          // Scan3Args::rec3 (item records: key, node offsets, weight)
    uint32_t *d_item_r3 = nullptr;       // [n_items] where every item's record starts (units of 64 dwords)
    Seg3 *d_segs3 = nullptr;             // the segments as k_scan3 reads them
    bool np_scaled = true;               // record offsets are byte offsets into the node masks (else node indices)
    uint4 *d_ct_rec = nullptr;           // ContentTable::rec
    // k_tile / k_scan3 per-call buffers: node masks and header per tile, window lists per
    // (tile, length) and their lengths (d_t3_hdr), the cold arguments
    uint32_t *d_tile_masks = nullptr, *d_t3_hdr = nullptr, *d_tile_hdr = nullptr, *d_t3_ref = nullptr;
    size_t t3_ref_cap = 0;
    uint32_t *d_ovh = nullptr;           // [n_items][4] k_overhang
    uint2 *d_t3_list = nullptr;
    size_t tile_masks_cap = 0, t3_hdr_cap = 0, t3_list_cap = 0, tile_hdr_cap = 0;
    Scan3Cold *d_cold = nullptr, *h_cold = nullptr;      // h_cold: pinned ring of COLD_RING structs
    int cold_next = 0;
    // k_scan2 takes the items of the well-populated alignment lengths: they come
    // first in the item order, one contiguous segment per length; the items of
    // the rare lengths follow and are scanned by k_scan
    std::vector<LenSeg> segs;            // every length, in item order
    LenSeg *d_segs = nullptr;            // the same on the device
    int n_hash_items = 0, n_hash_segs = 0;   // the k_scan2 prefix of items / segs
    int scan_mode = 0;                   // GFAL_SCAN: 0 auto, 1 k_scan only, 2 k_scan2 only
    uint16_t *d_lids = nullptr;          // [n_paths][nm] node ids along the paths (per call)
    size_t lids_cap = 0;
    int64_t n_lanes = 0;                 // alignments resident on the device (distinct ones if dedup)
    std::vector<int32_t> rep_of;         // dedup: caller's alignment -> the identical one that is resident
    int32_t *d_slot_orig = nullptr;    // [n_items*64] original index or -1
    uint32_t *d_status = nullptr;      // [8]: status flags, pad, worklist count (64 bit), 4 debug words
    unsigned long long *d_worklist = nullptr;   // as pushed by k_scan
    unsigned long long *d_worklist_sorted = nullptr;
    uint32_t wl_capacity = 0;
    uint32_t dp_sys_limit = 8192;   // entries of class 2 (half of it: classes 3+), see wavefront_class()
    uint32_t *d_len_bins = nullptr;    // [3 * LEN_BINS] path-length counting sort
    int32_t *d_order = nullptr;        // [n_paths] slot -> caller's path index
    uint32_t *d_counts_slot = nullptr; // [3 * n_paths] counters by slot
    size_t order_cap = 0, counts_slot_cap = 0;
    uint32_t *d_wl_bins = nullptr;     // [3][N_CLASSES * n_paths]: hist | offsets | cursor
    size_t wl_bins_cap = 0;
    uint32_t *d_rows = nullptr;        // DP row scratch
    uint16_t *d_images = nullptr;
    size_t images_cap = 0;             // uint16 units

    // blocking-API staging
    int32_t *d_path_off = nullptr, *d_path_steps = nullptr;
    uint32_t *d_counts = nullptr;
    size_t path_off_cap = 0, path_steps_cap = 0, counts_cap = 0;
    // blocking API: pinned staging, one copy in ([offsets | steps] -> d_path_off)
    // and one out ([counters | status words] <- d_counts)
    int32_t *h_in = nullptr;
    uint32_t *h_out = nullptr;
    size_t h_in_cap = 0, h_out_cap = 0;
    // blocking API, small batches: the ~15 dependent launches of a call replayed as
    // one HIP graph (re-captured every call, the executable updated in place).
    // Measured: no gain over direct launches on ROCm 7.2 (8 paths 0.147 vs 0.134 ms:
    // capture + update cost what the tighter dispatch saves), so opt-in: GFAL_GRAPHS=1
    bool use_graphs = false;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t stream = nullptr;      // owned, for the blocking API
    // the DP kernels of the different length classes run side by side: each is
    // bound by the latency of its longest single fill, not by throughput
    hipStream_t dp_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t dp_fork = nullptr, dp_join[3] = {nullptr, nullptr, nullptr};

    // search mode (gfal_group_score_children): inverted lists by node and the content
    // table, built on the device at the first use; the path store; per-call scratch
    bool child_index = false;
    uint32_t *d_inv_off = nullptr;
    uint4 *d_inv_ent = nullptr;
    uint32_t *d_ct_key = nullptr, *d_ct_hash = nullptr, *d_ct_mult = nullptr, *d_pair_bits = nullptr;
    uint32_t ct_mask = 0;
    int32_t *d_st_steps = nullptr, *d_st_len = nullptr;
    uint32_t *d_st_pass = nullptr, *d_st_g1 = nullptr;
    // per stored path: a bitmap over the inverted list of its first node -- the entries
    // the exact DP accepted -- and whether it is valid (paths kept by a full evaluation
    // have none); the list position of every worklist entry, as pushed and sorted
    uint32_t *d_st_bits = nullptr;
    int32_t *d_st_bitsok = nullptr;
    uint32_t bits_words = 0;
    uint32_t *d_wl_pos = nullptr, *d_wl_pos_sorted = nullptr;
    int64_t st_cap = 0;
    int32_t *d_child_in = nullptr;     // [parent | step | slot] of a children batch, or the slots of a stored one
    int32_t *d_child_tmp = nullptr;    // root | depth | dpass | dg1   (or the G1 snapshot)
    size_t child_in_cap = 0, child_tmp_cap = 0;
    int64_t n_children_calls = 0;
    bool out_on_host = false;          // the last call's kernels wrote the counters into h_out themselves

    int64_t n_score_calls = 0, n_device_passes = 0, n_overflow_reruns = 0;
    // status words of the last blocking call (they came back with its counters):
    // gfal_scorer_get_info then needs no device round trip (a search asks after every batch)
    bool status_cached = false;
    uint32_t cached_status[8] = {};
    hipEvent_t order_ev = nullptr;     // orders a call behind the previous one on another stream

    // last call
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    int last_tile = 0, last_grid = 0, last_lds = 0;
    // profiling: one {start, prep done, scan done, end} event set per call,
    // kept in a ring and averaged by gfal_scorer_get_info
    static constexpr int EV_RING = 128;
    bool profiling = false;
    bool prof_all = true;                // false: only the dominant scan kernel is bracketed (enable == 2)
    hipEvent_t ev[EV_RING][6] = {};      // [4], [5]: around the dominant scan kernel (k_scan3)
    bool ev_scan3 = false;               // the last profiled call recorded them
    int ev_calls = 0;   // calls recorded since profiling was switched on
};

namespace {

template <typename T>
int dev_upload(T **dst, const std::vector<T> &src)
{
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (!src.empty())
        HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T),
                          hipMemcpyHostToDevice));
    return GFAL_OK;
}

template <typename T>
int dev_reserve(T **buf, size_t *cap, size_t want)
{
    if (want <= *cap && *buf) return GFAL_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    size_t grow = std::max<size_t>(want, 1);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(buf), grow * sizeof(T)));
    *cap = grow;
    return GFAL_OK;
}

template <typename T>
int pinned_reserve(T **buf, size_t *cap, size_t want)
{
    if (want <= *cap && *buf) return GFAL_OK;
    if (*buf) HIP_TRY(hipHostFree(*buf));
    *buf = nullptr;
    size_t grow = std::max<size_t>(want + want / 2, 1024);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(buf), grow * sizeof(T), hipHostMallocDefault));
    *cap = grow;
    return GFAL_OK;
}

void free_scorer(gfal_scorer *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    void *bufs[] = {s->d_item_pairs, s->d_item_pbase, s->d_item_common, s->d_item_hdr, s->d_item_weight,
                    s->d_item_hash, s->d_item_pairs0, s->d_lids, s->d_segs,
                    s->d_rec3, s->d_item_r3, s->d_segs3, s->d_ct_rec, s->d_tile_masks, s->d_t3_hdr, s->d_t3_list,
                    s->d_tile_hdr, s->d_cold, s->d_t3_ref, s->d_ovh,
                    s->d_len_bins, s->d_order,
                    s->d_counts_slot,
                    s->d_node_local, s->d_node_hist, s->d_item_steps, s->d_item_base,
                    s->d_item_len,   s->d_slot_orig, s->d_status,     s->d_worklist,
                    s->d_worklist_sorted, s->d_wl_bins,
                    s->d_rows,       s->d_images,    s->d_path_off,   s->d_path_steps,
                    s->d_counts,     s->d_inv_off,   s->d_inv_ent,    s->d_ct_key,
                    s->d_ct_hash,    s->d_ct_mult,   s->d_pair_bits, s->d_st_steps,   s->d_st_len,
                    s->d_st_pass,    s->d_st_g1,     s->d_child_in,   s->d_child_tmp,
                    s->d_st_bits,    s->d_st_bitsok, s->d_wl_pos,     s->d_wl_pos_sorted};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    if (s->h_cold) (void)hipHostFree(s->h_cold);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    for (auto &set : s->ev)
        for (hipEvent_t e : set)
            if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    for (hipStream_t x : s->dp_stream)
        if (x) (void)hipStreamDestroy(x);
    if (s->dp_fork) (void)hipEventDestroy(s->dp_fork);
    if (s->order_ev) (void)hipEventDestroy(s->order_ev);
    for (hipEvent_t e : s->dp_join)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

size_t dp_lds_bytes(int max_aln_len)
{
    return (size_t)(max_aln_len + 1) * DP_THREADS * sizeof(uint32_t);
}

bool dp_rows_fit_lds(int max_aln_len) { return dp_lds_bytes(max_aln_len) <= 32 * 1024; }

size_t row_scratch_words(int max_aln_len)
{
    if (dp_rows_fit_lds(max_aln_len)) return 1;
    return (size_t)(max_aln_len + 1) * DP_BLOCKS * DP_THREADS;
}

}  // namespace

extern "C" {

int gfal_abi_version(void) { return GFAL_ABI_VERSION; }

#ifndef GFAL_BUILD_ID
#define GFAL_BUILD_ID "unstamped"
#endif
const char *gfal_build_id(void) { return GFAL_BUILD_ID; }

const char *gfal_strerror(int code)
{
    switch (code) {
    case GFAL_OK: return "ok";
    case GFAL_E_ARG: return "invalid argument";
    case GFAL_E_RANGE: return "path/alignment length or node id out of range";
    case GFAL_E_NO_DEVICE: return "no usable HIP device";
    case GFAL_E_HIP: return "HIP runtime error";
    case GFAL_E_NOMEM: return "out of memory";
    default: return "unknown error";
    }
}

const char *gfal_last_error(void) { return g_err; }

int gfal_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_err("hipGetDeviceCount: %s", hipGetErrorString(e));
        return GFAL_E_NO_DEVICE;
    }
    return n;
}

int gfal_scorer_create(const int32_t *aln_off, const int32_t *aln_steps,
                       int64_t n_aln, int32_t n_nodes, int device,
                       gfal_scorer **out)
{
    return gfal_scorer_create_ex(aln_off, aln_steps, n_aln, n_nodes, device, nullptr, 0, out);
}

int gfal_scorer_create_ex(const int32_t *aln_off, const int32_t *aln_steps,
                          int64_t n_aln, int32_t n_nodes, int device,
                          const int32_t *universe, int32_t n_universe,
                          gfal_scorer **out)
{
    return gfal_scorer_create_sharded(aln_off, aln_steps, n_aln, n_nodes, device, universe,
                                      n_universe, 0, 1, out);
}

// k_scan2<dedup weights, byte-wide node masks, length group>
static const void *scan2_kernel(bool w, bool nm8, int grp)
{
    typedef void (*kern_t)(Scan2Args);
#define GFAL_ROW(W_, N_) {k_scan2<W_, N_, 0>, k_scan2<W_, N_, 1>, k_scan2<W_, N_, 2>, k_scan2<W_, N_, 3>, k_scan2<W_, N_, 4>, k_scan2<W_, N_, 5>}
    static const kern_t table[2][2][SCAN2_GROUPS + 1] = {{GFAL_ROW(false, false), GFAL_ROW(false, true)},
                                                         {GFAL_ROW(true, false), GFAL_ROW(true, true)}};
#undef GFAL_ROW
    return reinterpret_cast<const void *>(table[w ? 1 : 0][nm8 ? 1 : 0][grp]);
}

static const void *scan3_kernel(bool w, bool nmg)
{
    typedef void (*kern_t)(Scan3Args);
    static const kern_t table[2][2] = {{k_scan3<false, false>, k_scan3<false, true>},
                                       {k_scan3<true, false>, k_scan3<true, true>}};
    return reinterpret_cast<const void *>(table[w ? 1 : 0][nmg ? 1 : 0]);
}

static int create_impl(const int32_t *aln_off, const int32_t *aln_steps, int64_t n_aln,
                       int32_t n_nodes, int device, const int32_t *universe, int32_t n_universe,
                       int32_t shard_index, int32_t n_shards, bool dedup, gfal_scorer **out,
                       uint8_t *plan_owned = nullptr);
static int build_content_table(gfal_scorer *s);

int gfal_scorer_create_sharded(const int32_t *aln_off, const int32_t *aln_steps,
                               int64_t n_aln, int32_t n_nodes, int device,
                               const int32_t *universe, int32_t n_universe,
                               int32_t shard_index, int32_t n_shards, gfal_scorer **out)
{
    return no_throw([&] {
        return create_impl(aln_off, aln_steps, n_aln, n_nodes, device, universe, n_universe,
                           shard_index, n_shards, false, out);
    });
}

int gfal_scorer_create_dedup(const int32_t *aln_off, const int32_t *aln_steps,
                             int64_t n_aln, int32_t n_nodes, int device,
                             const int32_t *universe, int32_t n_universe,
                             int32_t shard_index, int32_t n_shards, gfal_scorer **out)
{
    return no_throw([&] {
        return create_impl(aln_off, aln_steps, n_aln, n_nodes, device, universe, n_universe,
                           shard_index, n_shards, true, out);
    });
}

// plan_owned != NULL: the host half only -- which of the alignments this shard
// would take ([n_aln] flags), no device touched, *out untouched (gfal_shard_owner).
static int create_impl(const int32_t *aln_off, const int32_t *aln_steps, int64_t n_aln,
                       int32_t n_nodes, int device, const int32_t *universe, int32_t n_universe,
                       int32_t shard_index, int32_t n_shards, bool dedup, gfal_scorer **out,
                       uint8_t *plan_owned)
{
    if (!out && !plan_owned) return GFAL_E_ARG;
    if (out) *out = nullptr;
    // GFAL_DEBUG_TIMING=1: where the host side of create spends its time (stderr)
    const bool timing = getenv("GFAL_DEBUG_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "create: %-28s %.3f s\n", what,
                std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    if (n_shards < 1 || shard_index < 0 || shard_index >= n_shards) return GFAL_E_ARG;
    if (n_aln < 0 || n_nodes < 0 || (n_aln > 0 && (!aln_off || aln_off[0] != 0)))
        return GFAL_E_ARG;
    if (n_universe < 0 || (n_universe > 0 && !universe)) return GFAL_E_ARG;
    if (n_aln >= (int64_t)1 << 31) return GFAL_E_RANGE;
    const int64_t S = n_aln ? aln_off[n_aln] : 0;
    if (S < 0 || (S > 0 && !aln_steps)) return GFAL_E_ARG;

    if (!plan_owned) {
        int ndev = gfal_device_count();
        if (ndev <= 0 || device < 0 || device >= ndev) {
            if (ndev >= 0) set_err("device %d requested, %d visible", device, ndev);
            return GFAL_E_NO_DEVICE;
        }
    }

    // ---- validate, find the nodes that occur, bucket by length ----
    // node_local: >= 0 local id; -1 may be on a path but occurs in no
    // alignment; -2 outside the universe (a path that steps on it is an error)
    const bool has_universe = universe != nullptr;
    std::vector<int32_t> node_local((size_t)n_nodes, has_universe ? -2 : -1);
    for (int32_t u = 0; u < n_universe; ++u) {
        if (universe[u] < 0 || universe[u] >= n_nodes) {
            set_err("universe entry %d: node id out of range", u);
            return GFAL_E_RANGE;
        }
        node_local[(size_t)universe[u]] = -1;
    }
    int max_len = 0;
    uint32_t n_empty = 0;
    bool any_outside = false;
    std::vector<uint32_t> visit_order;
    for (int64_t k = 0; k < n_aln; ++k) {
        int64_t m = (int64_t)aln_off[k + 1] - aln_off[k];
        if (m < 0 || aln_off[k + 1] > S) return GFAL_E_ARG;
        if (m > GFAL_MAX_STEPS) {
            set_err("alignment %lld has %lld steps (max %d)", (long long)k,
                    (long long)m, GFAL_MAX_STEPS);
            return GFAL_E_RANGE;
        }
        if (m == 0 && shard_index == 0) ++n_empty;   // zero-step alignments go to shard 0
        max_len = std::max(max_len, (int)m);
    }
    for (int64_t t = 0; t < S; ++t) {
        int32_t s = aln_steps[t];
        if (s < 0 || (s >> 1) >= n_nodes) {
            set_err("alignment step %lld: node id out of range", (long long)t);
            return GFAL_E_RANGE;
        }
        if (node_local[s >> 1] == -1) node_local[s >> 1] = 0;
        else if (node_local[s >> 1] == -2) any_outside = true;
    }
    mark("validate");
    // Local ids follow a walk over the graph the alignments themselves trace
    // out (consecutive steps = an edge, weight = how often): depth-first along
    // the heaviest unvisited edge.  Nodes that are neighbours on the tangle get
    // neighbouring ids, so the content-sorted items hold alignments from one
    // region of the tangle: the lanes of a wave then agree on which candidate
    // paths they touch.  Any numbering gives the same counters; this one is
    // 7-15 % faster than the caller's order on the synthetic tangles.
    int n_local = 0;
    {
        std::vector<uint64_t> edges;
        const int64_t max_edges = (int64_t)4 << 20;   // (1 M sampled edges: scan +2.4 % at config 3)
        const int64_t stride = std::max<int64_t>(1, S / max_edges);
        for (int64_t k = 0; k < n_aln; k += stride)
            for (int64_t t = aln_off[k]; t + 1 < aln_off[k + 1]; ++t) {
                const uint32_t u = (uint32_t)(aln_steps[t] >> 1), w = (uint32_t)(aln_steps[t + 1] >> 1);
                if (u != w && node_local[u] == 0 && node_local[w] == 0)
                    edges.push_back(((uint64_t)std::min(u, w) << 32) | std::max(u, w));
            }
        if (edges.size() < ((size_t)1 << 18) || std::thread::hardware_concurrency() < 4) {
            std::sort(edges.begin(), edges.end());
        } else {      // four sorted quarters, merged pairwise
            const size_t q = edges.size() / 4;
            auto part = [&](int k) { return edges.begin() + (ptrdiff_t)(k == 4 ? edges.size() : q * (size_t)k); };
            std::thread t1([&] { std::sort(part(1), part(2)); });
            std::thread t2([&] { std::sort(part(2), part(3)); });
            std::thread t3([&] { std::sort(part(3), part(4)); });
            std::sort(part(0), part(1));
            t1.join();
            t2.join();
            t3.join();
            std::thread tm([&] { std::inplace_merge(part(2), part(3), part(4)); });
            std::inplace_merge(part(0), part(1), part(2));
            tm.join();
            std::inplace_merge(part(0), part(2), part(4));
        }
        struct Nb { uint32_t to, weight; };
        std::vector<uint32_t> deg((size_t)n_nodes + 1, 0);
        std::vector<std::pair<uint64_t, uint32_t>> uniq;     // edge, multiplicity
        for (size_t i = 0; i < edges.size();) {
            size_t j = i;
            while (j < edges.size() && edges[j] == edges[i]) ++j;
            uniq.emplace_back(edges[i], (uint32_t)(j - i));
            ++deg[(size_t)(edges[i] >> 32)];
            ++deg[(size_t)(edges[i] & 0xFFFFFFFFu)];
            i = j;
        }
        std::vector<size_t> at((size_t)n_nodes + 1, 0);
        for (int32_t v = 0; v < n_nodes; ++v) at[(size_t)v + 1] = at[(size_t)v] + deg[(size_t)v];
        std::vector<Nb> nb(at[(size_t)n_nodes]);
        std::vector<size_t> fill(at.begin(), at.end() - 1);
        for (auto &e : uniq) {
            const uint32_t u = (uint32_t)(e.first >> 32), w = (uint32_t)(e.first & 0xFFFFFFFFu);
            nb[fill[u]++] = Nb{w, e.second};
            nb[fill[w]++] = Nb{u, e.second};
        }
        for (int32_t v = 0; v < n_nodes; ++v)
            std::sort(nb.begin() + (ptrdiff_t)at[(size_t)v], nb.begin() + (ptrdiff_t)at[(size_t)v + 1],
                      [](const Nb &x, const Nb &y) {
                          return x.weight != y.weight ? x.weight > y.weight : x.to < y.to;
                      });
        std::vector<size_t> cursor(at.begin(), at.end() - 1);   // next neighbour to try
        auto next_unvisited = [&](uint32_t v) -> int64_t {
            while (cursor[v] < at[(size_t)v + 1]) {
                const uint32_t w = nb[cursor[v]].to;
                if (node_local[w] == 0) return w;
                ++cursor[v];
            }
            return -1;
        };
        std::vector<uint32_t> stack;
        // first start: an end of the walk if there is one (fewest neighbours)
        int32_t first_start = -1;
        for (int32_t v = 0; v < n_nodes; ++v)
            if (node_local[v] == 0 && deg[(size_t)v] > 0 &&
                (first_start < 0 || deg[(size_t)v] < deg[(size_t)first_start]))
                first_start = v;
        auto walk_from = [&](uint32_t start) {
            uint32_t cur = start;
            while (true) {
                node_local[cur] = -3;            // visited, id assigned below
                visit_order.push_back(cur);
                int64_t nx = next_unvisited(cur);
                if (nx >= 0) {
                    stack.push_back(cur);
                    cur = (uint32_t)nx;
                    continue;
                }
                nx = -1;
                while (!stack.empty()) {
                    nx = next_unvisited(stack.back());
                    if (nx >= 0) break;
                    stack.pop_back();
                }
                if (nx < 0) break;
                cur = (uint32_t)nx;
            }
        };
        if (first_start >= 0) walk_from((uint32_t)first_start);
        for (int32_t v = 0; v < n_nodes; ++v)
            if (node_local[v] == 0) walk_from((uint32_t)v);
        for (uint32_t v : visit_order) node_local[v] = n_local++;
    }
    mark("node numbering");
    // every node outside the universe shares one local id: no path can carry
    // it, so such steps only ever fail the filter / a comparison, and their
    // histogram bin keeps `unaligned` exact
    const int32_t outside_lid = any_outside ? n_local++ : -1;
    if (n_local > MAX_LOCAL_NODES) {
        set_err("%d distinct nodes in the alignments; this build stages at most %d "
                "(pass the nodes candidate paths can visit to gfal_scorer_create_ex)",
                n_local, MAX_LOCAL_NODES);
        return GFAL_E_RANGE;
    }
    {
        // what binds first is k_prep's workgroup: the image, the node ids along the path
        // and a 32-bit chain head per node must fit 160 KiB of LDS (~25 600 nodes)
        const ImageLayout L = make_layout(n_local, GFAL_MAX_STEPS);
        const size_t prep_lds = (size_t)L.total * sizeof(uint16_t) + (size_t)L.nm * sizeof(uint16_t) +
                                (size_t)L.v2 * sizeof(uint32_t);
        if (prep_lds > (size_t)LDS_MAX) {
            set_err("%d distinct nodes in the alignments; the path preparation stages at most ~25 600 "
                    "(pass the nodes candidate paths can visit to gfal_scorer_create_ex)", n_local);
            return GFAL_E_RANGE;
        }
    }
    std::vector<uint32_t> hist((size_t)n_local, 0);
    std::vector<uint16_t> local_steps((size_t)S);
    for (int64_t t = 0; t < S; ++t) {
        int32_t s = aln_steps[t];
        int32_t mapped = node_local[s >> 1];
        uint32_t lid = (uint32_t)(mapped >= 0 ? mapped : outside_lid);
        local_steps[(size_t)t] = (uint16_t)((lid << 1) | ((uint32_t)s & 1u));
    }

    mark("local steps");
    // alignments of one length together, ordered by content so the lanes of a
    // wave look at neighbouring table entries and take the same branches
    std::vector<std::vector<int32_t>> by_len((size_t)max_len + 1);
    for (int64_t k = 0; k < n_aln; ++k) {
        int m = aln_off[k + 1] - aln_off[k];
        if (m > 0) by_len[(size_t)m].push_back((int32_t)k);
    }
    {
        // content order per length bucket: 64-bit key of the first four steps,
        // the rest only on equal keys; buckets are sorted side by side
        std::vector<int> lengths;
        for (int m = 1; m <= max_len; ++m)
            if (by_len[(size_t)m].size() > 1) lengths.push_back(m);
        std::sort(lengths.begin(), lengths.end(), [&](int x, int y) {
            return by_len[(size_t)x].size() > by_len[(size_t)y].size();
        });
        const uint16_t *ls = local_steps.data();
        auto sort_bucket = [&](int m) {
            struct Keyed {
                uint64_t key;
                int32_t idx;
            };
            std::vector<int32_t> &idx = by_len[(size_t)m];
            std::vector<Keyed> keyed(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) {
                const uint16_t *px = ls + aln_off[idx[i]];
                uint64_t key = 0;
                for (int t = 0; t < 4; ++t) key = (key << 16) | (t < m ? px[t] : 0u);
                keyed[i] = Keyed{key, idx[i]};
            }
            auto less = [&](const Keyed &x, const Keyed &y) {
                if (x.key != y.key) return x.key < y.key;
                const uint16_t *px = ls + aln_off[x.idx], *py = ls + aln_off[y.idx];
                for (int t = 4; t < m; ++t)
                    if (px[t] != py[t]) return px[t] < py[t];
                return x.idx < y.idx;
            };
            if (keyed.size() < ((size_t)1 << 18) || std::thread::hardware_concurrency() < 4) {
                std::sort(keyed.begin(), keyed.end(), less);
            } else {      // a big bucket: four sorted quarters, merged pairwise
                const size_t q = keyed.size() / 4;
                auto part = [&](int k) { return keyed.begin() + (ptrdiff_t)(k == 4 ? keyed.size() : q * (size_t)k); };
                std::thread t1([&] { std::sort(part(1), part(2), less); });
                std::thread t2([&] { std::sort(part(2), part(3), less); });
                std::thread t3([&] { std::sort(part(3), part(4), less); });
                std::sort(part(0), part(1), less);
                t1.join();
                t2.join();
                t3.join();
                std::thread tm([&] { std::inplace_merge(part(2), part(3), part(4), less); });
                std::inplace_merge(part(0), part(1), part(2), less);
                tm.join();
                std::inplace_merge(part(0), part(2), part(4), less);
            }
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = keyed[i].idx;
        };
        unsigned n_threads = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *env = getenv("GFAL_CREATE_THREADS")) n_threads = (unsigned)std::max(1, atoi(env));
        n_threads = (unsigned)std::min<size_t>(n_threads, std::max<size_t>(1, lengths.size()));
        if (S < (1 << 16)) n_threads = 1;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i = next++; i < lengths.size(); i = next++) sort_bucket(lengths[i]);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    mark("bucket + sort");
    // dedup: runs of identical alignments (neighbours in the content order) become
    // one lane with a weight; rep_of[] remembers who stands for whom
    std::vector<std::vector<uint32_t>> wt_len;
    std::vector<int32_t> rep_of;
    if (dedup) {
        wt_len.resize((size_t)max_len + 1);
        rep_of.resize((size_t)n_aln);
        for (int64_t k = 0; k < n_aln; ++k) rep_of[(size_t)k] = (int32_t)k;
        const uint16_t *ls = local_steps.data();
        for (int m = 1; m <= max_len; ++m) {
            std::vector<int32_t> &idx = by_len[(size_t)m];
            std::vector<int32_t> reps;
            std::vector<uint32_t> &wts = wt_len[(size_t)m];
            for (size_t i = 0; i < idx.size(); ++i) {
                const bool same = !reps.empty() &&
                                  memcmp(ls + aln_off[reps.back()], ls + aln_off[idx[i]],
                                         (size_t)m * sizeof(uint16_t)) == 0;
                if (same) {
                    ++wts.back();
                    rep_of[(size_t)idx[i]] = reps.back();
                } else {
                    reps.push_back(idx[i]);
                    wts.push_back(1u);
                }
            }
            idx.swap(reps);
        }
        mark("dedup");
    }
    // item directory first (sequential, cheap), then the lanes are filled in
    // parallel: every item writes its own ranges
    struct ItemSrc {
        const int32_t *idx;   // the item's alignments (original indices)
        const uint32_t *wt;   // their weights (dedup) or NULL
        int cnt, m;
    };
    std::vector<ItemSrc> src;
    std::vector<uint32_t> item_base, item_pbase;
    std::vector<uint16_t> item_len;
    uint64_t n_u16 = 0, n_pairs = 0, n_pairs0 = 0;
    int64_t own_aln = n_empty, own_steps = 0, n_lanes = 0;
    // Item order: the well-populated lengths first (k_scan2 builds one window
    // table per tile and length, which pays from a few dozen items up), then the
    // rare ones (k_scan).  The shards of one set agree on the order (same input)
    // and on who takes which group of 64 (below).
    // (k_scan2 built a window table per (tile, length): worth it from ~48 items on.  With
    // k_tile / k_scan3 a length costs a tile one light workgroup and k_scan -- chain walks for
    // 6 paths at a time -- costs ~2 us per item at 10 000 paths: from 4 items on a length is a
    // segment of its own, whatever the number of shards.  The shard that ended up with the rare
    // lengths of an 8-way split took 0.62 ms for 14 000 alignments before, 0.43 of it in k_scan.)
    int hash_min_items = 4;
    if (const char *env = getenv("GFAL_HASH_MIN_ITEMS")) hash_min_items = std::max(1, atoi(env));
    std::vector<int> len_order;
    std::vector<char> is_hash_len((size_t)max_len + 1, 0);
    for (int m = 1; m <= max_len; ++m)
        if ((int64_t)by_len[(size_t)m].size() >= (int64_t)hash_min_items * WAVE) {
            is_hash_len[(size_t)m] = 1;
            len_order.push_back(m);
        }
    for (int m = 1; m <= max_len; ++m)
        if (!is_hash_len[(size_t)m] && !by_len[(size_t)m].empty()) len_order.push_back(m);
    std::vector<LenSeg> segs;      // one per length, in item order
    int n_hash_items = 0, n_hash_segs = 0;
    // Which group of 64 goes to which shard.  k_scan2 pays one prologue per (tile, length)
    // whatever the shard holds of that length -- measured at config 3: one length costs a
    // shard as much as ~2000 groups of average length do, whatever the set size -- so a
    // shard should hold FEW lengths; but within a length its groups must be spread over
    // the whole content order (a contiguous piece would be one region of the tangle: busy
    // or idle with the batch's paths).  The lengths, in item order, are laid on a line:
    // a fixed stretch for holding the length at all, then its groups weighted by what one
    // costs (about proportional to the length); shard k owns the piece [k W / n,
    // (k + 1) W / n) of the line; group i of a length falls on its groups' stretch at the
    // fraction frac(i * golden ratio) (low discrepancy: every run of i's spreads evenly).
    // A shard then holds two or three lengths (the short tail of rare lengths, which run
    // in one k_scan launch, counts as one), and all shards of a set derive the same
    // partition from the same input.  (Round robin over all groups, the first policy,
    // made every shard pay every prologue: 35 % of ideal at 1/8 of config 3, now 45 %.)
    // (what a group costs depends on how many of the batch's paths its alignments can be on: about
    // proportional to the length at config 3, 7 + length at config 5; 3 + length is in between)
    auto group_weight = [](int m) { return (uint64_t)(36 + 12 * m); };
    // (~270 groups of average length: what a length costs a shard with k_tile / k_scan3 -- one
    // light workgroup per tile; k_scan2's table per (tile, length) cost ~2000; the k_scan
    // launch of the rare lengths: twice that)
    uint64_t shard_fixed = 24000;
    if (const char *env = getenv("GFAL_SHARD_FIXED")) shard_fixed = (uint64_t)std::max(0ll, atoll(env));
    std::vector<uint64_t> len_w_lo((size_t)max_len + 2, 0);      // where a length's groups start on the line
    uint64_t total_w = 0;
    {
        // (a small set: the fixed stretches must not crowd the groups off the line, or
        // shards end up empty -- at most a quarter of the groups' share per length)
        uint64_t items_w = 0, n_charged = 0;
        bool any_rare = false;
        for (int m : len_order) {
            items_w += (uint64_t)((by_len[(size_t)m].size() + WAVE - 1) / WAVE) * group_weight(m);
            if (is_hash_len[(size_t)m]) ++n_charged;
            else any_rare = true;
        }
        n_charged += any_rare ? 2 : 0;
        if (n_charged) shard_fixed = std::min(shard_fixed, items_w / (4 * n_charged) + 1);
        bool rare_seen = false;
        for (int m : len_order) {
            if (is_hash_len[(size_t)m]) total_w += shard_fixed;
            else if (!rare_seen) total_w += 2 * shard_fixed;
            if (!is_hash_len[(size_t)m]) rare_seen = true;
            len_w_lo[(size_t)m] = total_w;
            total_w += (uint64_t)((by_len[(size_t)m].size() + WAVE - 1) / WAVE) * group_weight(m);
        }
    }
    const bool by_length = getenv("GFAL_SHARD_ROUND_ROBIN") == nullptr;
    uint64_t global_item = 0;
    for (int m : len_order) {
        const std::vector<int32_t> &idx = by_len[(size_t)m];
        const size_t seg_lo = src.size();
        for (size_t at = 0; at < idx.size(); at += WAVE) {
            const size_t cnt = std::min<size_t>(WAVE, idx.size() - at);
            // shards are cut AFTER the global sort, item by item: a shard's items
            // are a subset of the unsharded ones (same lanes side by side), so the
            // scan kernel does on 1/n of the items exactly 1/n of the work
            {
                const uint64_t gi = global_item++;
                int32_t owner;
                if (n_shards == 1) {
                    owner = 0;
                } else if (by_length) {
                    const uint64_t n_groups = (idx.size() + WAVE - 1) / WAVE;
                    const uint64_t stretch = n_groups * group_weight(m);
                    const uint64_t u = ((uint64_t)(at / WAVE) * 2654435769ull) & 0xFFFFFFFFull;     // frac(i * phi) in 2^-32
                    const uint64_t pos = len_w_lo[(size_t)m] + ((stretch * u) >> 32);
                    owner = (int32_t)std::min<uint64_t>((uint64_t)n_shards - 1,
                                                        (unsigned __int128)pos * (uint64_t)n_shards / total_w);
                } else {
                    owner = (int32_t)(gi % (uint64_t)n_shards);      // (round 1 / first half of round 2)
                }
                if (owner != shard_index) continue;
            }
            const uint32_t *wt = dedup ? wt_len[(size_t)m].data() + at : nullptr;
            src.push_back(ItemSrc{idx.data() + at, wt, (int)cnt, m});
            item_base.push_back((uint32_t)(n_u16 / WAVE));
            item_pbase.push_back((uint32_t)(n_pairs / WAVE));
            item_len.push_back((uint16_t)m);
            n_u16 += (uint64_t)m * WAVE;
            n_pairs += (uint64_t)(m / 2) * WAVE;
            for (size_t l = 0; l < cnt; ++l) {
                const int64_t w = wt ? wt[l] : 1;
                own_aln += w;
                own_steps += w * m;
            }
            n_lanes += (int64_t)cnt;
        }
        if (src.size() > seg_lo) {
            segs.push_back(LenSeg{(uint32_t)seg_lo, (uint32_t)src.size(), (uint32_t)m, 1u,
                                  item_base[seg_lo], (uint32_t)(n_pairs0 / WAVE)});
            n_pairs0 += (uint64_t)(src.size() - seg_lo) * (uint64_t)((m + 1) / 2) * WAVE;
            if (is_hash_len[(size_t)m]) {
                n_hash_items = (int)src.size();
                n_hash_segs = (int)segs.size();
            }
        }
    }
    if (n_u16 / WAVE >= ((uint64_t)1 << 32) || n_pairs0 / WAVE >= ((uint64_t)1 << 32) ||
        src.size() >= ((size_t)1 << 25)) {
        set_err("shard too large for 32-bit item addressing");
        return GFAL_E_RANGE;
    }
    std::vector<uint16_t> item_steps((size_t)n_u16, (uint16_t)STEP_INVALID);
    std::vector<uint32_t> item_pairs((size_t)n_pairs, 0xFFFFFFFFu);
    std::vector<int32_t> slot_orig(src.size() * WAVE, -1);
    std::vector<uint32_t> item_common(src.size(), NO_COMMON_NODE);
    std::vector<uint32_t> item_weight(dedup ? src.size() * WAVE : 0, 0u);
    std::vector<uint32_t> item_hash(src.size() * WAVE, 0u);
    std::vector<uint32_t> item_pairs0((size_t)n_pairs0, 0xFFFFFFFFu);
    // k_scan3's item records (see Seg3): the key is filled in on the device (k_ct_build)
    const int v2_local = (n_local + 1) & ~1;
    const bool np_scaled = v2_local * 4 + 8 <= 0xFFFF;
    const uint32_t np_zero = (uint32_t)v2_local * (np_scaled ? 4u : 1u);      // the mask word that stays zero
    std::vector<uint32_t> item_r3(src.size(), 0u);
    std::vector<Seg3> segs3;
    uint64_t n_rec3 = 0;
    for (const LenSeg &sg : segs) {
        const uint32_t R = 1u + (sg.m + 1) / 2 + (dedup ? 1u : 0u);
        segs3.push_back(Seg3{sg.item_lo, sg.item_hi, sg.m, 1u, sg.step_base, (uint32_t)(n_rec3 / WAVE)});
        for (uint32_t it = sg.item_lo; it < sg.item_hi; ++it) {
            item_r3[it] = (uint32_t)(n_rec3 / WAVE);
            n_rec3 += (uint64_t)R * WAVE;
        }
    }
    if (n_rec3 / WAVE >= ((uint64_t)1 << 32)) {
        set_err("shard too large for 32-bit item addressing");
        return GFAL_E_RANGE;
    }
    std::vector<uint32_t> rec3((size_t)n_rec3, np_zero | (np_zero << 16));
    // where every item's pairs0 block starts (units of 64 dwords): its segment's base
    // plus its rank in the segment times ceil(m / 2)
    std::vector<uint32_t> item_p0base(src.size(), 0u);
    for (const LenSeg &sg : segs)
        for (uint32_t it = sg.item_lo; it < sg.item_hi; ++it)
            item_p0base[it] = sg.p0_base + (it - sg.item_lo) * ((sg.m + 1) / 2);
    {
        unsigned n_threads = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *env = getenv("GFAL_CREATE_THREADS")) n_threads = (unsigned)std::max(1, atoi(env));
        if (src.size() < 4096) n_threads = 1;
        std::vector<std::vector<uint32_t>> hists(n_threads, std::vector<uint32_t>((size_t)n_local, 0));
        const uint16_t *ls = local_steps.data();
        auto fill = [&](unsigned tid) {
            std::vector<uint32_t> &h = hists[tid];
            const size_t lo = src.size() * tid / n_threads, hi = src.size() * (tid + 1) / n_threads;
            for (size_t it = lo; it < hi; ++it) {
                const ItemSrc &is = src[it];
                const int m = is.m, K = m / 2;
                uint16_t *steps = item_steps.data() + (size_t)item_base[it] * WAVE;
                uint32_t *pairs = item_pairs.data() + (size_t)item_pbase[it] * WAVE;
                for (int l = 0; l < is.cnt; ++l) {
                    const uint16_t *px = ls + aln_off[is.idx[l]];
                    const uint32_t w = is.wt ? is.wt[l] : 1u;
                    if (is.wt) item_weight[it * WAVE + (size_t)l] = w;
                    uint32_t wh = whash_init();
                    for (int t = 0; t < m; ++t) {
                        steps[(size_t)t * WAVE + l] = px[t];
                        h[px[t] >> 1] += w;
                        wh = whash_step(wh, px[t]);
                    }
                    item_hash[it * WAVE + (size_t)l] = whash_final(wh, m);
                    for (int k = 0; k < K; ++k)
                        pairs[(size_t)k * WAVE + l] =
                            (uint32_t)px[2 * k + 1] |
                            ((2 * k + 2 < m) ? ((uint32_t)px[2 * k + 2] << 16) : 0u);
                    uint32_t *pairs0 = item_pairs0.data() + (size_t)item_p0base[it] * WAVE;
                    uint32_t *r3 = rec3.data() + (size_t)item_r3[it] * WAVE;
                    const uint32_t np_mul = np_scaled ? 4u : 1u;
                    for (int k = 0; k < (m + 1) / 2; ++k) {
                        pairs0[(size_t)k * WAVE + l] =
                            (uint32_t)px[2 * k] | ((2 * k + 1 < m) ? ((uint32_t)px[2 * k + 1] << 16) : 0xFFFF0000u);
                        const uint32_t n_lo = ((uint32_t)px[2 * k] >> 1) * np_mul;
                        r3[(size_t)(k + 1) * WAVE + l] =
                            n_lo | ((2 * k + 1 < m ? ((uint32_t)px[2 * k + 1] >> 1) * np_mul : n_lo) << 16);
                    }
                    if (dedup) r3[(size_t)((m + 1) / 2 + 1) * WAVE + l] = w;
                    slot_orig[it * WAVE + (size_t)l] = is.idx[l];
                }
                // nodes every lane has: the first and the last such node of lane 0
                // (two 16-bit halves; NO_COMMON_NODE if there is none)
                const uint16_t *p0 = ls + aln_off[is.idx[0]];
                uint32_t first_c = 0xFFFFu, last_c = 0xFFFFu;
                for (int t = 0; t < m; ++t) {
                    const uint32_t node = p0[t] >> 1;
                    if (node == first_c || node == last_c) continue;
                    bool everywhere = true;
                    for (int l = 1; l < is.cnt && everywhere; ++l) {
                        const uint16_t *px = ls + aln_off[is.idx[l]];
                        bool has = false;
                        for (int u = 0; u < m && !has; ++u) has = (uint32_t)(px[u] >> 1) == node;
                        everywhere = has;
                    }
                    if (everywhere) {
                        if (first_c == 0xFFFFu) first_c = node;
                        last_c = node;
                    }
                }
                if (first_c != 0xFFFFu) {
                    item_common[it] = first_c | (last_c << 16);          // every lane has both
                } else {
                    // no node in all lanes (the item straddles a change of the first
                    // node): if the lanes start with one of two nodes, a tile path
                    // must carry at least one of those
                    uint32_t n_a = p0[0] >> 1, n_b = 0xFFFFu;
                    bool two = true;
                    for (int l = 1; l < is.cnt && two; ++l) {
                        const uint32_t node = ls[aln_off[is.idx[l]]] >> 1;
                        if (node == n_a || node == n_b) continue;
                        if (n_b == 0xFFFFu) n_b = node;
                        else two = false;
                    }
                    if (two)
                        item_common[it] = n_a | COMMON_EITHER | ((n_b == 0xFFFFu ? n_a : n_b) << 16);
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(fill, t);
        fill(0);
        for (auto &t : pool) t.join();
        for (auto &h : hists)
            for (size_t v = 0; v < h.size(); ++v) hist[v] += h[v];
    }
    mark("items");
    if (plan_owned) {
        memset(plan_owned, 0, (size_t)n_aln);
        for (int32_t orig : slot_orig)
            if (orig >= 0) plan_owned[(size_t)orig] = 1;
        if (shard_index == 0)
            for (int64_t k = 0; k < n_aln; ++k)
                if (aln_off[k + 1] == aln_off[k]) plan_owned[(size_t)k] = 1;
        if (dedup)
            for (int64_t k = 0; k < n_aln; ++k)
                if (plan_owned[(size_t)rep_of[(size_t)k]]) plan_owned[(size_t)k] = 1;
        return GFAL_OK;
    }

    gfal_scorer *s = new (std::nothrow) gfal_scorer();
    if (!s) return GFAL_E_NOMEM;
    s->device = device;
    s->n_aln = own_aln;
    s->n_aln_in = n_aln;
    s->n_steps = own_steps;
    // which of the caller's alignments this shard scores (pair_scores)
    s->owned.assign((size_t)n_aln, 0);
    for (int32_t orig : slot_orig)
        if (orig >= 0) s->owned[(size_t)orig] = 1;
    if (shard_index == 0)
        for (int64_t k = 0; k < n_aln; ++k)
            if (aln_off[k + 1] == aln_off[k]) s->owned[(size_t)k] = 1;
    if (dedup) {      // the copies go where their representative goes
        for (int64_t k = 0; k < n_aln; ++k)
            if (s->owned[(size_t)rep_of[(size_t)k]]) s->owned[(size_t)k] = 1;
        s->rep_of.swap(rep_of);
    }
    s->n_lanes = n_lanes;
    s->n_nodes = n_nodes;
    s->n_local = n_local;
    s->max_aln_len = max_len;
    s->n_empty = n_empty;
    s->n_items = (int)item_base.size();
    s->n_item_u16 = (int64_t)item_steps.size();

    int rc = GFAL_OK;
    struct Guard {      // an exception below must not leak the device buffers
        gfal_scorer *s;
        ~Guard() { if (s) free_scorer(s); }
    } guard{s};
    auto fail = [&](int code) {
        guard.s = nullptr;
        free_scorer(s);
        return code;
    };
#define CREATE_TRY(expr)                                                       \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                    __FILE__, __LINE__);                                       \
            return fail(e_ == hipErrorOutOfMemory ? GFAL_E_NOMEM : GFAL_E_HIP); \
        }                                                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, device));
    CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&s->dp_fork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) {
        CREATE_TRY(hipStreamCreateWithFlags(&s->dp_stream[i], hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&s->dp_join[i], hipEventDisableTiming));
    }
    // both kernels may ask for more than the default 64 KiB of dynamic LDS
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LDS_BUDGET));
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LDS_BUDGET));
    for (int w = 0; w < 2; ++w)
        for (int nm8 = 0; nm8 < 2; ++nm8)
            for (int grp = 0; grp <= SCAN2_GROUPS; ++grp)
                CREATE_TRY(hipFuncSetAttribute(scan2_kernel(w != 0, nm8 != 0, grp),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    for (int w = 0; w < 2; ++w)
        for (int g = 0; g < 2; ++g)
            CREATE_TRY(hipFuncSetAttribute(scan3_kernel(w != 0, g != 0),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_masks),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    if ((rc = dev_upload(&s->d_node_local, node_local))) return fail(rc);
    if ((rc = dev_upload(&s->d_node_hist, hist))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_steps, item_steps))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_base, item_base))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_len, item_len))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_pairs, item_pairs))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_pbase, item_pbase))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_common, item_common))) return fail(rc);
    if (dedup && (rc = dev_upload(&s->d_item_weight, item_weight))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_hash, item_hash))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_pairs0, item_pairs0))) return fail(rc);
    for (size_t it = 0; it < src.size(); ++it)
        for (int l = 0; l < WAVE; ++l) rec3[(size_t)item_r3[it] * WAVE + l] = KEY_EMPTY;
    if ((rc = dev_upload(&s->d_rec3, rec3))) return fail(rc);
    if ((rc = dev_upload(&s->d_item_r3, item_r3))) return fail(rc);
    if ((rc = dev_upload(&s->d_segs3, segs3))) return fail(rc);
    s->np_scaled = np_scaled;
    if ((rc = dev_upload(&s->d_segs, segs))) return fail(rc);
    s->segs = segs;
    s->n_hash_items = n_hash_items;
    s->n_hash_segs = n_hash_segs;
    if (const char *env = getenv("GFAL_SCAN")) s->scan_mode = atoi(env);
    {
        std::vector<uint4> item_hdr(src.size());
        for (size_t it = 0; it < src.size(); ++it)
            item_hdr[it] = make_uint4(item_base[it], item_pbase[it], item_common[it], item_len[it]);
        if ((rc = dev_upload(&s->d_item_hdr, item_hdr))) return fail(rc);
    }
    if ((rc = dev_upload(&s->d_slot_orig, slot_orig))) return fail(rc);
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_len_bins),
                         3 * LEN_BINS * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_status), 8 * sizeof(uint32_t)));
    CREATE_TRY(hipMemset(s->d_status, 0, 8 * sizeof(uint32_t)));
    // worklist: at least one entry per alignment, so a single path always fits
    if (const char *env = getenv("GFAL_DP_SYS_LIMIT")) s->dp_sys_limit = (uint32_t)atoll(env);
    if (const char *env = getenv("GFAL_GRAPHS")) s->use_graphs = atoi(env) != 0;
    s->wl_capacity = (uint32_t)std::max<int64_t>(own_aln, (int64_t)1 << 22);
    if (const char *env = getenv("GFAL_DEBUG_WL_CAPACITY"))   // tests: force the overflow path
        s->wl_capacity = (uint32_t)std::max<int64_t>(own_aln, atoll(env));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_worklist),
                         (size_t)s->wl_capacity * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_worklist_sorted),
                         (size_t)s->wl_capacity * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_rows),
                         (size_t)row_scratch_words(max_len) * sizeof(uint32_t)));

#undef CREATE_TRY
    mark("device upload");
    if ((rc = build_content_table(s))) return fail(rc);
    mark("content table");
    guard.s = nullptr;
    *out = s;
    return GFAL_OK;
}

void gfal_scorer_destroy(gfal_scorer *s) { free_scorer(s); }

int gfal_shard_owner(const int32_t *aln_off, const int32_t *aln_steps, int64_t n_aln,
                     int32_t n_nodes, const int32_t *universe, int32_t n_universe,
                     int32_t n_shards, int32_t *owner)
{
    if (!owner || n_shards < 1) return GFAL_E_ARG;
    return no_throw([&] {
        std::vector<uint8_t> flags((size_t)std::max<int64_t>(n_aln, 1));
        for (int64_t k = 0; k < n_aln; ++k) owner[k] = -1;
        for (int32_t sh = 0; sh < n_shards; ++sh) {
            const int rc = create_impl(aln_off, aln_steps, n_aln, n_nodes, 0, universe, n_universe, sh,
                                       n_shards, false, nullptr, flags.data());
            if (rc != GFAL_OK) return rc;
            for (int64_t k = 0; k < n_aln; ++k)
                if (flags[(size_t)k]) {
                    if (owner[k] >= 0) {
                        set_err("alignment %lld is claimed by shards %d and %d", (long long)k, owner[k], sh);
                        return GFAL_E_ARG;
                    }
                    owner[k] = sh;
                }
        }
        return GFAL_OK;
    });
}

int gfal_scorer_set_profiling(gfal_scorer *s, int enable)
{
    if (!s) return GFAL_E_ARG;
    if (enable) {
        HIP_TRY(hipSetDevice(s->device));
        for (auto &set : s->ev)
            for (hipEvent_t &e : set)
                if (!e) HIP_TRY(hipEventCreate(&e));
    }
    s->profiling = enable != 0;
    s->prof_all = enable != 2;
    s->ev_calls = 0;
    return GFAL_OK;
}

// Worklist entries the blocking API may grow to on its own (two lists of 8 bytes
// per entry: 2 x 8 GiB); beyond that a batch is split instead.
constexpr unsigned long long WL_MAX_ENTRIES = 1ull << 30;

// Make room for `need` exact-DP pairs per call (+25 %).  The streams that used
// the old lists must be idle.
static int grow_worklist(gfal_scorer *s, unsigned long long need)
{
    unsigned long long want = need + need / 4 + 1024;
    if (want > WL_MAX_ENTRIES) want = WL_MAX_ENTRIES;
    if (want <= s->wl_capacity) return GFAL_OK;
    if (need > want) {
        set_err("exact-DP worklist: %llu pairs in one batch exceed the %llu-entry limit: split the batch",
                need, WL_MAX_ENTRIES);
        return GFAL_E_NOMEM;
    }
    HIP_TRY(hipSetDevice(s->device));
    if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
    unsigned long long *a = nullptr, *b = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&a), (size_t)want * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&b), (size_t)want * sizeof(unsigned long long)) != hipSuccess) {
        if (a) (void)hipFree(a);
        (void)hipGetLastError();
        set_err("exact-DP worklist: cannot allocate %llu entries", want);
        return GFAL_E_NOMEM;
    }
    (void)hipFree(s->d_worklist);
    (void)hipFree(s->d_worklist_sorted);
    s->d_worklist = a;
    s->d_worklist_sorted = b;
    s->wl_capacity = (uint32_t)want;
    if (s->d_wl_pos) {          // search mode keeps a list position beside every entry
        (void)hipFree(s->d_wl_pos);
        (void)hipFree(s->d_wl_pos_sorted);
        s->d_wl_pos = s->d_wl_pos_sorted = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_wl_pos), (size_t)want * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_wl_pos_sorted), (size_t)want * sizeof(uint32_t)));
    }
    return GFAL_OK;
}

// Per-call device buffers that grow with the batch (image pool, worklist bins,
// slot order, slot counters).  Growing frees and allocates: not inside a stream
// capture, and not while an earlier call may still use the old buffers.
static int ensure_call_buffers(gfal_scorer *s, int32_t n_paths, const ImageLayout &L)
{
    size_t want = (size_t)n_paths * L.total;
    if (want > s->images_cap) {
        if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
        int rc = dev_reserve(&s->d_images, &s->images_cap, want);
        if (rc) return rc;
    }
    size_t bins = (size_t)3 * N_CLASSES * n_paths + 16;   // + class_lo[6], class totals[5]
    if (bins > s->wl_bins_cap || (size_t)n_paths > s->order_cap) {
        if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
        int rc = dev_reserve(&s->d_wl_bins, &s->wl_bins_cap, bins);
        if (rc) return rc;
        if ((rc = dev_reserve(&s->d_order, &s->order_cap, (size_t)n_paths))) return rc;
        if ((rc = dev_reserve(&s->d_counts_slot, &s->counts_slot_cap, (size_t)3 * n_paths)))
            return rc;
    }
    want = (size_t)n_paths * L.nm;
    if (want > s->lids_cap) {
        if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
        int rc = dev_reserve(&s->d_lids, &s->lids_cap, want);
        if (rc) return rc;
    }
    return GFAL_OK;
}

// k_tile_masks + k_tile + k_scan3 over the segments [0, n_segs) of the scorer, in slabs of
// tiles that fit the window-list buffer.
constexpr int COLD_RING = 64;
static int launch_scan3(gfal_scorer *s, hipStream_t st, const Items &items, const ImageLayout &L,
                        int32_t n_paths, int32_t max_path_len, int filter, int n_segs, int n_items3,
                        int want_groups, int slots, uint32_t *d_counts, unsigned long long *wl_count,
                        uint32_t *d_hist, hipEvent_t *ev)
{
    const int tile = std::max(1, std::min(T3_MAX, (int)n_paths));
    const int n_tiles = (n_paths + tile - 1) / tile;
    // LDS of a k_scan3 workgroup: node masks (4 bytes per node) + table (8 bytes per slot).
    // The table holds at least one path's windows at load 1/2 (4096 slots); many nodes:
    // the masks stay in HBM (read through L1 / L2)
    const int v2p = L.v2 + 2;
    const size_t mask_bytes = (size_t)v2p * sizeof(uint32_t);
    uint32_t h_slots = 8192;
    bool nmg = !s->np_scaled;
    // reduction words + the workgroup's pairs for the DP + the marks of the items that need the overhang test
    const size_t fixed3 = 256 + (size_t)WL3_WORDS * 4 + (size_t)SCAN2_WAVES * SCAN3_GROUPS * 8;
    if (mask_bytes + (size_t)h_slots * 8 + fixed3 > (size_t)LDS_BUDGET) h_slots = 4096;
    if (mask_bytes + (size_t)h_slots * 8 + fixed3 > (size_t)LDS_BUDGET) nmg = true;
    if (const char *env = getenv("GFAL_SCAN3_NMG")) nmg = nmg || atoi(env) != 0;
    if (nmg) h_slots = 8192;
    if (const char *env = getenv("GFAL_SCAN3_SLOTS")) {
        const int v = atoi(env);
        if ((v == 4096 || v == 8192) && (nmg ? 0 : mask_bytes) + (size_t)v * 8 + fixed3 <= (size_t)LDS_BUDGET)
            h_slots = (uint32_t)v;
    }
    const size_t lds3 = (nmg ? 0 : mask_bytes) + (size_t)h_slots * 8 + fixed3;
    const uint32_t stride = (uint32_t)tile * 2u * (uint32_t)max_path_len;
    // slabs of tiles: the window lists of a slab stay below GFAL_SCAN3_LIST_MB (worst case:
    // every window of every path its own entry; what is touched is what exists)
    size_t list_limit = (size_t)6 << 30;
    if (const char *env = getenv("GFAL_SCAN3_LIST_MB")) list_limit = (size_t)std::max(1, atoi(env)) << 20;
    const size_t per_tile = (size_t)n_segs * stride * sizeof(uint2);
    const int slab_tiles = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_tiles, list_limit / std::max<size_t>(per_tile, 1)));
    {
        const size_t want_masks = (size_t)slab_tiles * v2p, want_hdr = (size_t)slab_tiles * n_segs,
                     want_list = (size_t)slab_tiles * n_segs * stride, want_thdr = (size_t)slab_tiles * T3_THDR_WORDS;
        const size_t want_ref = (size_t)n_segs * 2 * L.nm;
        if (want_masks > s->tile_masks_cap || want_hdr > s->t3_hdr_cap || want_list > s->t3_list_cap ||
            want_thdr > s->tile_hdr_cap || want_ref > s->t3_ref_cap || !s->d_cold) {
            if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
            int rc;
            if ((rc = dev_reserve(&s->d_tile_masks, &s->tile_masks_cap, want_masks))) return rc;
            if ((rc = dev_reserve(&s->d_t3_hdr, &s->t3_hdr_cap, want_hdr))) return rc;
            if ((rc = dev_reserve(&s->d_t3_list, &s->t3_list_cap, want_list))) return rc;
            if ((rc = dev_reserve(&s->d_tile_hdr, &s->tile_hdr_cap, want_thdr))) return rc;
            if ((rc = dev_reserve(&s->d_t3_ref, &s->t3_ref_cap, want_ref))) return rc;
            if (!s->d_cold) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_ovh), std::max<size_t>((size_t)s->n_items, 1) * 16));
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_cold), sizeof(Scan3Cold)));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->h_cold), COLD_RING * sizeof(Scan3Cold),
                                      hipHostMallocDefault));
            }
        }
    }
    // the cold arguments: through pinned memory (a ring: calls may be queued), copied on the
    // device by the first kernel
    Scan3Cold *cold = s->h_cold + (s->cold_next++ % COLD_RING);
    cold->item_steps = items.steps;
    cold->images = s->d_images;
    cold->L = L;
    cold->n_paths = n_paths;
    cold->worklist = s->d_worklist;
    cold->wl_count = wl_count;
    cold->wl_capacity = s->wl_capacity;
    cold->wl_hist = d_hist;
    cold->status = s->d_status;

    TileArgs ta;
    ta.items = items;
    ta.ct = ContentTable{s->d_ct_rec, s->ct_mask, s->d_pair_bits};
    ta.images = s->d_images;
    ta.L = L;
    ta.lids = s->d_lids;
    ta.n_paths = n_paths;
    ta.tile = tile;
    ta.segs = s->d_segs;
    ta.n_segs = n_segs;
    ta.filter = filter ? 1 : 0;
    ta.debug = getenv("GFAL_DEBUG_TILE") ? atoi(getenv("GFAL_DEBUG_TILE")) : 0;
    ta.v2p = v2p;
    ta.tile_hdr = s->d_tile_hdr;
    ta.tile_masks = s->d_tile_masks;
    ta.list_count = s->d_t3_hdr;
    ta.ref = s->d_t3_ref;
    ta.list = s->d_t3_list;
    ta.stride = stride;

    Scan3Args a3;
    a3.rec3 = s->d_rec3;
    a3.common = items.common;
    a3.item_steps = items.steps;
    a3.ovh = s->d_ovh;
    a3.cold = s->d_cold;
    a3.tile_masks = s->d_tile_masks;
    a3.tile_hdr = s->d_tile_hdr;
    a3.list_count = s->d_t3_hdr;
    a3.list = s->d_t3_list;
    a3.stride = stride;
    a3.n_paths = n_paths;
    a3.tile = tile;
    a3.debug = getenv("GFAL_DEBUG_SCAN2") ? atoi(getenv("GFAL_DEBUG_SCAN2")) : 0;
    a3.v2p = v2p;
    a3.nm_shift = s->np_scaled ? 0 : 2;
    a3.n_segs_total = n_segs;
    a3.h_slots = h_slots;
    a3.counts = d_counts;
    a3.ct_mult = s->d_ct_mult;
    a3.counts_dbg = s->d_status;
    // chunks per segment: in proportion to the segment's items (see k_scan2's launch)
    int y_want = (want_groups + n_tiles - 1) / n_tiles;
    int min_items = 24 * SCAN2_WAVES;
    if (n_tiles < slots) {
        int rounds4 = n_tiles <= 40 ? 4 : 8;
        if (const char *env = getenv("GFAL_SCAN2_ROUNDS4")) rounds4 = std::max(1, atoi(env));
        y_want = std::min(y_want, (rounds4 * slots / 4 + n_tiles - 1) / n_tiles);
        min_items = 12 * SCAN2_WAVES;
    }
    if (const char *env = getenv("GFAL_SCAN3_MIN_ITEMS")) min_items = std::max(1, atoi(env)) * SCAN2_WAVES;
    a3.chunk_mult = (((unsigned long long)y_want << 24) + (unsigned long long)n_items3 - 1) /
                    (unsigned long long)std::max<int64_t>(n_items3, 1);
    a3.chunk_inv_min = (1ull << 24) / (unsigned long long)min_items;

    s->last_tile = tile;
    s->last_lds = (int)lds3;
    hipLaunchKernelGGL(k_overhang, dim3((unsigned)((n_items3 + 3) / 4)), dim3(256), 0, st, items, n_items3,
                       (const uint16_t *)s->d_images, L, s->d_ovh);
    for (int t0 = 0; t0 < n_tiles; t0 += slab_tiles) {
        const int nt = std::min(slab_tiles, n_tiles - t0);
        ta.tile0 = t0;
        ta.n_launch_tiles = nt;
        // (the reference blocks ride along with the first slab's masks)
        const unsigned ref_blocks = t0 == 0 ? (unsigned)(((size_t)n_segs * L.nm + 1023) / 1024) : 0u;
        hipLaunchKernelGGL(k_tile_masks, dim3((unsigned)nt + ref_blocks), dim3(1024), mask_bytes, st, ta,
                           (const Scan3Cold *)cold, t0 == 0 ? s->d_cold : (Scan3Cold *)nullptr,
                           (int)sizeof(Scan3Cold));
        hipLaunchKernelGGL(k_tile, dim3((unsigned)nt), dim3(1024), t3_tile_lds(n_segs), st, ta);
        HIP_TRY(hipGetLastError());
        a3.tile0 = t0;
        a3.n_tiles = nt;
        // (profiling: the dominant kernel by itself, when the batch is one slab)
        const bool time_it = ev != nullptr && slab_tiles >= n_tiles;
        if (time_it) HIP_TRY(hipEventRecord(ev[4], st));
        for (int s0 = 0; s0 < n_segs; s0 += MAX_SEGS) {
            const int ns = std::min(MAX_SEGS, n_segs - s0);
            a3.segs = s->d_segs3 + s0;
            a3.n_segs = ns;
            a3.seg0 = s0;
            unsigned y_total = 0;
            for (int k = 0; k < ns; ++k) {
                const LenSeg &sg = s->segs[(size_t)(s0 + k)];
                y_total += seg_chunks3(sg.item_hi - sg.item_lo, a3.chunk_mult, a3.chunk_inv_min);
            }
            const unsigned grid3 = (unsigned)nt * y_total;
            void *kargs[] = {&a3};
            HIP_TRY(hipLaunchKernel(scan3_kernel(s->d_item_weight != nullptr, nmg), dim3(grid3),
                                    dim3(SCAN2_THREADS), kargs, lds3, st));
            s->last_grid += (int)grid3;
        }
        if (time_it) {
            HIP_TRY(hipEventRecord(ev[5], st));
            s->ev_scan3 = true;
        }
    }
    return GFAL_OK;
}

// What a call does besides scoring (search mode, see k_child).
struct ExtraCtx {
    int mode = 0;                    // 1: keep the batch's paths in the store; 2: children batch
    const int32_t *d_slots = nullptr;   // mode 1: [n_paths] store slot per path or -1
    uint32_t *d_g1_tmp = nullptr;       // mode 1: [n_paths] scratch
    ChildBatch batch;                // mode 2
};

static int score_device_impl(gfal_scorer *s, const int32_t *d_path_off,
                             const int32_t *d_path_steps, int32_t n_paths,
                             int64_t total_steps, int32_t max_path_len, int filter,
                             uint32_t *d_counts, void *hip_stream, uint32_t *status_copy,
                             const ExtraCtx *cx = nullptr)
{
    if (!s || n_paths < 0 || total_steps < 0) return GFAL_E_ARG;
    if (n_paths == 0) return GFAL_OK;
    if (!d_path_off || !d_path_steps || !d_counts) return GFAL_E_ARG;
    if (max_path_len < 1 || max_path_len > GFAL_MAX_STEPS) return GFAL_E_RANGE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipSetDevice(s->device));
    // the per-call buffers (images, worklists, bins, status) are shared by all
    // calls on this scorer: a call on another stream waits for the previous one
    if (s->have_last && s->last_stream != st) {
        HIP_TRY(hipEventRecord(s->order_ev, s->last_stream));
        HIP_TRY(hipStreamWaitEvent(st, s->order_ev, 0));
    }
    ++s->n_device_passes;
    // the previous blocking call's exact-DP list, if its status words came back with it
    const long long prev_dp_pairs =
        s->status_cached ? (long long)((unsigned long long)s->cached_status[2] | ((unsigned long long)s->cached_status[3] << 32))
                         : -1;
    s->status_cached = false;

    const ImageLayout L = make_layout(s->n_local, max_path_len);
    const size_t img_bytes = (size_t)L.total * sizeof(uint16_t);
    const size_t mask_bytes = (size_t)L.v2 * sizeof(uint32_t);
    // k_scan stages whole images (first-occurrence tables included: they grow with
    // the node count), k_scan2 only steps, node masks and its window table
    const bool chain_fits = img_bytes + mask_bytes <= (size_t)LDS_BUDGET;
    // (node masks of 32 bits per node, or -- many nodes -- of 8: see node_mask())
    const size_t scan2_per_path = (size_t)L.nm * 2 * sizeof(uint16_t);
    const size_t scan2_fixed32 = mask_bytes + (size_t)H_SLOTS * 5 + 64;
    const size_t scan2_fixed8 = (((size_t)L.v2 + 3) & ~(size_t)3) + (size_t)H_SLOTS * 5 + 64;
    auto tile_for = [&](size_t fixed, int cap) {
        return fixed + scan2_per_path > (size_t)LDS_BUDGET
                   ? 0
                   : (int)std::min<size_t>(((size_t)LDS_BUDGET - fixed) / scan2_per_path, (size_t)cap);
    };
    const int tile32 = tile_for(scan2_fixed32, TILE2_MAX), tile8 = tile_for(scan2_fixed8, TILE2_MAX_NM8);
    // byte reads from LDS are slow, a smaller tile is slower still (config 5, 5 000 nodes:
    // 7 paths per tile with byte masks 19.6 ms, 5 paths with 32-bit masks 26.7 ms)
    bool nm8 = tile8 > tile32;
    if (const char *env = getenv("GFAL_NM8")) nm8 = atoi(env) != 0 && tile8 > 0;
    const size_t scan2_fixed = nm8 ? scan2_fixed8 : scan2_fixed32;
    const bool hash_fits = (nm8 ? tile8 : tile32) > 0;
    if (!chain_fits && !hash_fits) {
        set_err("%d local nodes and paths of up to %d steps exceed the LDS budget of the scan kernels",
                s->n_local, (int)max_path_len);
        return GFAL_E_RANGE;
    }
    if (img_bytes + (size_t)L.nm * sizeof(uint16_t) + mask_bytes > (size_t)LDS_MAX) {
        set_err("%d local nodes exceed the LDS budget of the path preparation", s->n_local);
        return GFAL_E_RANGE;
    }
    {
        int rc = ensure_call_buffers(s, n_paths, L);
        if (rc) return rc;
    }
    const int n_bins = N_CLASSES * n_paths;
    uint32_t *d_hist = s->d_wl_bins, *d_offsets = s->d_wl_bins + n_bins,
             *d_cursor = s->d_wl_bins + 2 * (size_t)n_bins,
             *d_class_lo = s->d_wl_bins + 3 * (size_t)n_bins;
    // the worklist histogram is cleared by k_prep (every path its own bins), the
    // status words by k_len_sort_block
    const bool one_block_sort = n_paths <= 32768;
    if (!one_block_sort && !(cx && cx->mode == 2)) {
        HIP_TRY(hipMemsetAsync(s->d_status, 0, 8 * sizeof(uint32_t), st));
    }
    hipEvent_t *ev = s->ev[s->ev_calls % gfal_scorer::EV_RING];
    if (s->profiling) {
        if (s->prof_all) HIP_TRY(hipEventRecord(ev[0], st));
        s->ev_scan3 = false;
    }

    const size_t prep_lds = img_bytes + (size_t)L.nm * sizeof(uint16_t) +
                            (size_t)L.v2 * sizeof(uint32_t);
    // longest paths first (see k_len_*): everything below works on slots
    const unsigned p_blocks = (unsigned)((n_paths + 255) / 256);
    uint32_t *const d_user_counts = d_counts;
    d_counts = s->d_counts_slot;
    const bool children = cx && cx->mode == 2;
    const bool inherit = children && s->bits_words > 0;      // DP results remembered per stored path
    const int32_t *const d_order = children ? nullptr : s->d_order;      // (children: slot = index)
    if (children) {
        // (k_child_len cleared the status words)
    } else if (one_block_sort) {
        hipLaunchKernelGGL(k_len_sort_block, dim3(1), dim3(LEN_BINS), 0, st, d_path_off,
                           (int)n_paths, s->d_order, s->d_status, 8, nullptr, 0);
    } else {
        HIP_TRY(hipMemsetAsync(s->d_len_bins, 0, LEN_BINS * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_len_hist, dim3(p_blocks), dim3(256), 0, st, d_path_off,
                           (int)n_paths, s->d_len_bins);
        hipLaunchKernelGGL(k_len_offsets, dim3(1), dim3(LEN_BINS), 0, st, s->d_len_bins);
        hipLaunchKernelGGL(k_len_scatter, dim3(p_blocks), dim3(256), 0, st, d_path_off,
                           (int)n_paths, s->d_len_bins, s->d_order);
    }
    PrepChild pc;
    if (children) {
        pc.root = cx->batch.root;
        pc.depth = cx->batch.depth;
        pc.parent = cx->batch.parent;
        pc.step = cx->batch.step;
        pc.slot = cx->batch.slot;
        pc.st_steps = cx->batch.st_steps;
        pc.st_len = cx->batch.st_len;
    }
    hipLaunchKernelGGL(k_prep, dim3((unsigned)n_paths), dim3(PREP_THREADS), prep_lds, st,
                       d_path_off, d_path_steps, (int)n_paths, total_steps,
                       (int)max_path_len, s->d_node_local, (int)s->n_nodes,
                       s->d_node_hist, (uint32_t)s->n_steps, s->n_empty, filter, L,
                       d_order, s->d_images, d_counts, s->d_status, d_hist, s->d_lids, pc);
    HIP_TRY(hipGetLastError());
    if (s->profiling && s->prof_all) HIP_TRY(hipEventRecord(ev[1], st));

    s->last_tile = 0;
    s->last_grid = 0;
    s->last_lds = 0;
    if (s->n_items > 0) {
        const Items items{s->d_item_steps, s->d_item_base, s->d_item_len, s->d_item_pairs,
                          s->d_item_pbase, s->n_items, s->d_item_common, s->d_item_hdr, s->d_item_weight};
        unsigned long long *const wl_count = reinterpret_cast<unsigned long long *>(s->d_status + 2);
        int want_groups = 8192;
        if (const char *env = getenv("GFAL_SCAN_GROUPS")) want_groups = std::max(1, atoi(env));
        const int slots = 2 * s->n_cus;
        // Which kernel scans what.  k_scan2 (window tables): the segments of the
        // well-populated lengths -- or all of them when k_scan's images do not fit
        // (many nodes) or GFAL_SCAN=2 asks; k_scan (occurrence chains): the rest.
        int n_segs2 = s->n_hash_segs;
        // a batch of a few dozen paths is one or two tiles: k_scan2's per-workgroup
        // prologue (staging, node masks, the tile's windows into the table) is then
        // the whole latency (0.20 ms at 8 paths against 0.04 ms), above ~100 paths it
        // is ahead (scripts/small_batch_probe.py)
        if (n_paths < 96) n_segs2 = 0;
        if (!hash_fits || s->scan_mode == 1) n_segs2 = 0;
        if (!chain_fits || s->scan_mode == 2 || s->scan_mode == 3) n_segs2 = (int)s->segs.size();
        if (n_segs2 == 0 && !chain_fits) return GFAL_E_RANGE;
        const int item_lo_chain = n_segs2 > 0 ? (int)s->segs[(size_t)n_segs2 - 1].item_hi : 0;
        if (children) {
            // every child from its parent: the two inverted lists and the content table
            // instead of a scan over all alignments
            ChildArgs c;
            c.items = items;
            c.ix = ChildIndex{s->d_inv_off, s->d_inv_ent, s->d_ct_key, s->d_ct_hash, s->d_ct_mult, s->ct_mask};
            c.images = s->d_images;
            c.L = L;
            c.lids = s->d_lids;
            c.order = nullptr;
            c.n_paths = n_paths;
            c.max_aln_len = s->max_aln_len;
            c.dpass = cx->batch.dpass;
            c.dg1 = cx->batch.dg1;
            c.b = cx->batch;
            c.wl_pos = inherit ? s->d_wl_pos : nullptr;
            c.counts = d_counts;
            c.worklist = s->d_worklist;
            c.wl_count = wl_count;
            c.wl_capacity = s->wl_capacity;
            c.wl_hist = d_hist;
            c.status = s->d_status;
            int chunks = std::max(1, std::min(32, 8 * s->n_cus / (int)n_paths));
            if (const char *env = getenv("GFAL_CHILD_CHUNKS")) chunks = std::max(1, atoi(env));
            const size_t lds_c = img_bytes + (size_t)L.nm * sizeof(uint16_t) +
                                 (size_t)N_CLASSES * ((L.v2 + 31) / 32) * sizeof(uint32_t);
            if (lds_c > (size_t)LDS_MAX - CHILD_STATIC_LDS) {
                set_err("%d local nodes exceed the LDS budget of the children kernel", s->n_local);
                return GFAL_E_RANGE;
            }
            if (s->d_item_weight)
                hipLaunchKernelGGL(k_child<true>, dim3((unsigned)n_paths, (unsigned)chunks), dim3(CHILD_THREADS),
                                   lds_c, st, c);
            else
                hipLaunchKernelGGL(k_child<false>, dim3((unsigned)n_paths, (unsigned)chunks), dim3(CHILD_THREADS),
                                   lds_c, st, c);
            HIP_TRY(hipGetLastError());
            s->last_tile = 1;
            s->last_grid = (int)n_paths * chunks;
            s->last_lds = (int)lds_c;
        }

        int scan_forks = 0;          // side streams that run a scan launch of this call
        // k_scan3 (content identities, 31 paths per tile) takes what k_scan2 took unless
        // GFAL_SCAN=2 asks for the older kernel
        // or the batch is small: k_tile works a tile through in one workgroup, ~0.1 ms whatever
        // the batch, which a couple of hundred paths do not pay back (scripts/small_batch_probe.py,
        // scan phase with k_scan2 / with k_tile + k_scan3: 128 paths 0.15 / 0.22 ms, 256 paths
        // 0.22 / 0.24, 384 paths 0.30 / 0.24, 1024 paths k_scan3 0.29 ms)
        int scan3_min_paths = 320;
        if (const char *env = getenv("GFAL_SCAN3_MIN_PATHS")) scan3_min_paths = atoi(env);
        const bool use3 = n_segs2 > 0 && !children && s->scan_mode != 2 &&
                          (n_paths >= scan3_min_paths || s->scan_mode == 3);
        if (use3) {
            const int rc3 = launch_scan3(s, st, items, L, n_paths, max_path_len, filter, n_segs2, item_lo_chain,
                                         want_groups, slots, d_counts, wl_count, d_hist,
                                         s->profiling ? ev : nullptr);
            if (rc3) return rc3;
        }
        if (n_segs2 > 0 && !children && !use3) {
            Scan2Args a2;
            a2.items = items;
            a2.item_hash = s->d_item_hash;
            a2.pairs0 = s->d_item_pairs0;
            a2.images = s->d_images;
            a2.L = L;
            a2.lids = s->d_lids;
            a2.n_paths = n_paths;
            int tile = nm8 ? tile8 : tile32;
            tile = std::max(1, std::min(tile, (int)n_paths));
            a2.tile = tile;
            a2.n_tiles = (n_paths + tile - 1) / tile;
            a2.filter = filter ? 1 : 0;
            a2.debug = getenv("GFAL_DEBUG_SCAN2") ? atoi(getenv("GFAL_DEBUG_SCAN2")) : 0;
            a2.counts = d_counts;
            a2.worklist = s->d_worklist;
            a2.wl_count = wl_count;
            a2.wl_capacity = s->wl_capacity;
            a2.wl_hist = d_hist;
            a2.status = s->d_status;
            const size_t lds2 = scan2_fixed + (size_t)tile * scan2_per_path;
            // chunks per segment: in proportion to the segment's items, every
            // workgroup keeping enough items to pay for its prologue (staging, node
            // masks, one table insert per window of the tile); small batches trade
            // that for filling the GPU, as k_scan does
            const int64_t items2 = item_lo_chain;
            int y_want = (want_groups + a2.n_tiles - 1) / a2.n_tiles;
            int min_items = 100 * SCAN2_WAVES;
            if (a2.n_tiles < slots) {          // fewer tiles than resident workgroups: fill the GPU
                // quarter rounds of the resident workgroups to aim for: every workgroup
                // pays a prologue, so a search-sized batch (128 paths: 16 tiles) wants
                // ONE round (scan 0.14 ms against 0.24 ms with four), a few hundred
                // paths two (scripts/small_batch_probe.py)
                int rounds4 = a2.n_tiles <= 40 ? 4 : 8;
                if (const char *env = getenv("GFAL_SCAN2_ROUNDS4")) rounds4 = std::max(1, atoi(env));
                y_want = std::min(y_want, (rounds4 * slots / 4 + a2.n_tiles - 1) / a2.n_tiles);
                min_items = 12 * SCAN2_WAVES;
            }
            a2.chunk_mult = (((unsigned long long)y_want << 24) + (unsigned long long)items2 - 1) /
                            (unsigned long long)std::max<int64_t>(items2, 1);
            a2.chunk_inv_min = (1ull << 24) / (unsigned long long)min_items;
            // one launch per run of segments of one length group (the lengths ascend: as
            // many runs as groups that occur), side by side: the first on the caller's
            // stream, the others on the side streams the DP kernels use later
            int n_launch = 0;
            int &joined = scan_forks;
            bool split = n_paths >= 4096;       // (see scan2_group)
            if (const char *env = getenv("GFAL_SCAN2_SPLIT")) split = atoi(env) != 0;
            for (int s0 = 0; s0 < n_segs2;) {
                const int grp = split ? scan2_group((int)s->segs[(size_t)s0].m) : SCAN2_GROUPS;
                int ns = 1;
                while (s0 + ns < n_segs2 && ns < MAX_SEGS &&
                       (!split || scan2_group((int)s->segs[(size_t)(s0 + ns)].m) == grp))
                    ++ns;
                a2.segs = s->d_segs + s0;
                a2.n_segs = ns;
                unsigned y_total = 0;
                for (int k = 0; k < ns; ++k) {
                    const LenSeg &sg = s->segs[(size_t)(s0 + k)];
                    y_total += seg_chunks(sg.item_hi - sg.item_lo, a2.chunk_mult, a2.chunk_inv_min);
                }
                const unsigned grid2 = (unsigned)a2.n_tiles * y_total;
                hipStream_t on = st;
                // (10 000-path batches at config 3, its 1/8 shard and config 5: within 1-2 %
                // of one launch after the other, GFAL_SCAN2_SERIAL=1; a 128-path batch, whose
                // launches are mostly prologue: 0.26 against 0.35 ms)
                static const bool serial = getenv("GFAL_SCAN2_SERIAL") != nullptr;
                if (n_launch > 0 && !serial) {
                    const int side = (n_launch - 1) % 3;
                    if (n_launch == 1) HIP_TRY(hipEventRecord(s->dp_fork, st));
                    if (!(joined & (1 << side))) HIP_TRY(hipStreamWaitEvent(s->dp_stream[side], s->dp_fork, 0));
                    joined |= 1 << side;
                    on = s->dp_stream[side];
                }
                void *kargs[] = {&a2};
                HIP_TRY(hipLaunchKernel(scan2_kernel(s->d_item_weight != nullptr, nm8, grp), dim3(grid2),
                                        dim3(SCAN2_THREADS), kargs, lds2, on));
                s->last_grid += (int)grid2;
                ++n_launch;
                s0 += ns;
            }

            s->last_tile = tile;
            s->last_lds = (int)lds2;
        }

        ScanArgs a;
        a.items = items;
        a.images = s->d_images;
        a.L = L;
        a.n_paths = n_paths;
        a.filter = filter ? 1 : 0;
        a.counts = d_counts;
        a.worklist = s->d_worklist;
        a.wl_count = wl_count;
        a.wl_capacity = s->wl_capacity;
        a.wl_hist = d_hist;
        a.status = s->d_status;
        a.item_lo = item_lo_chain;
        const int n_items_chain = s->n_items - item_lo_chain;
        if (n_items_chain > 0 && !children) {
            int tile = (int)std::min<size_t>(((size_t)LDS_BUDGET - mask_bytes) / img_bytes,
                                             MAX_TILE);
            tile = std::max(1, std::min(tile, (int)n_paths));
            a.tile = tile;
            a.n_tiles = (n_paths + tile - 1) / tile;
            // enough workgroups to fill 256 CUs x 2 several times over, but every
            // chunk keeps a few items per wave
            int chunks = (want_groups + a.n_tiles - 1) / a.n_tiles;
            // at least ~100 items per wave and workgroup: every workgroup re-stages its
            // tile's images, which small shards cannot amortise otherwise, and the
            // item rejection works on 64 items of a wave at a time
            const int want_chunks = chunks;
            const int max_chunks = std::max(1, n_items_chain / (100 * SCAN_WAVES));
            chunks = std::max(1, std::min(chunks, max_chunks));
            // small batches (what a search submits) end up with about one round of the
            // 2-per-CU resident workgroups, and tiles of long paths cost more than
            // tiles of short ones: trade staging for balance down to ~12 items per
            // wave until there are four rounds, and always fill the first round
            if (a.n_tiles < slots && (long long)a.n_tiles * chunks < 4LL * slots) {
                const int balanced = std::min((4 * slots + a.n_tiles - 1) / a.n_tiles,
                                              std::max(1, n_items_chain / (12 * SCAN_WAVES)));
                chunks = std::max(chunks, std::min(want_chunks, balanced));
            }
            if ((long long)a.n_tiles * chunks < slots && n_segs2 == 0)
                chunks = std::max(chunks, std::min(slots / a.n_tiles,
                                                   std::max(1, n_items_chain / SCAN_WAVES)));
            a.n_chunks = chunks;
            const size_t lds = std::max((size_t)tile * img_bytes + mask_bytes,
                                        (size_t)2 * MAX_TILE * sizeof(uint32_t));
            const unsigned grid = (unsigned)a.n_tiles * (unsigned)a.n_chunks;
            if (s->d_item_weight)
                hipLaunchKernelGGL(k_scan<true>, dim3(grid), dim3(SCAN_THREADS), lds, st, a);
            else
                hipLaunchKernelGGL(k_scan<false>, dim3(grid), dim3(SCAN_THREADS), lds, st, a);
            HIP_TRY(hipGetLastError());
            if (n_segs2 == 0) {
                s->last_tile = tile;
                s->last_lds = (int)lds;
            }
            s->last_grid += (int)grid;
        }
        for (int i = 0; i < 3; ++i)      // the scan launches on the side streams are done
            if (scan_forks & (1 << i)) {
                HIP_TRY(hipEventRecord(s->dp_join[i], s->dp_stream[i]));
                HIP_TRY(hipStreamWaitEvent(st, s->dp_join[i], 0));
            }
        if (cx && cx->mode == 1)      // good-without-DP, before the DP kernels add theirs
            hipLaunchKernelGGL(k_store_snapshot, dim3(p_blocks), dim3(256), 0, st, d_counts, (int)n_paths,
                               cx->d_g1_tmp);
        if (s->profiling && s->prof_all) HIP_TRY(hipEventRecord(ev[2], st));

        // a search's batches: a short list last time (and no alignment longer than a wave)
        // -> one launch runs every DP class, on the list as it was pushed
        static const bool dp_small_off = getenv("GFAL_DP_SMALL") != nullptr && atoi(getenv("GFAL_DP_SMALL")) == 0;
        const bool dp_small = children && !dp_small_off && prev_dp_pairs >= 0 && prev_dp_pairs <= 16384 &&
                              s->max_aln_len <= 64;
        if (!dp_small) {
            hipLaunchKernelGGL(k_wl_offsets, dim3(N_CLASSES), dim3(1024), 0, st, d_hist, d_offsets,
                               d_cursor, d_class_lo + 8, (int)n_paths);
            hipLaunchKernelGGL(k_wl_scatter, dim3(256), dim3(256), 0, st, a.items,
                               s->d_worklist, a.wl_count, s->wl_capacity, d_offsets,
                               d_cursor, (uint32_t)n_paths, s->d_worklist_sorted, d_class_lo + 8, d_class_lo,
                               inherit ? s->d_wl_pos : nullptr, s->d_wl_pos_sorted);
        }
        DpArgs d;
        d.items = a.items;
        d.images = s->d_images;
        d.L = L;
        d.n_paths = n_paths;
        d.sorted = dp_small ? s->d_worklist : s->d_worklist_sorted;
        d.class_lo = d_class_lo;
        d.wl_count = a.wl_count;
        d.wl_capacity = s->wl_capacity;
        d.sys_limit = s->dp_sys_limit;
        d.row_scratch = s->d_rows;
        d.counts = d_counts;
        d.sorted_pos = inherit ? (dp_small ? s->d_wl_pos : s->d_wl_pos_sorted) : nullptr;
        d.status = s->d_status;
        d.bits = inherit ? s->d_st_bits : nullptr;
        d.bits_words = s->bits_words;
        d.q_slot = inherit ? cx->batch.slot : nullptr;
        if (dp_small)
            hipLaunchKernelGGL(k_dp_small, dim3(512, 4), dim3(DP_THREADS), 0, st, d);
        // fork: classes 8 / 16 / 32+ on side streams, class 4 on the caller's
        if (!dp_small) HIP_TRY(hipEventRecord(s->dp_fork, st));
        int forked = 0;
        auto side = [&](int i) -> hipStream_t {
            (void)hipStreamWaitEvent(s->dp_stream[i], s->dp_fork, 0);
            forked |= 1 << i;
            return s->dp_stream[i];
        };
        // every kernel of both families is launched; the list length (known on
        // the device only) decides which family returns at once
        if (!dp_small) {
        const bool big = n_paths >= 2048;
        // (a big batch: the throughput kernel first, here too -- the other family's workgroups
        // return at once, but a launch in front of the one with the work is 6-12 us of the step)
        if (!big) hipLaunchKernelGGL(k_dp_sys<4>, dim3(DP_SYS_BLOCKS), dim3(DP_THREADS), 0, st, d);
        hipLaunchKernelGGL((k_dp_regs<4, 0>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, st, d);
        if (big) hipLaunchKernelGGL(k_dp_sys<4>, dim3(DP_SYS_BLOCKS), dim3(DP_THREADS), 0, st, d);
        if (s->max_aln_len > 4) {
            hipStream_t s0 = side(0);
            // (a big batch: the throughput kernel first -- behind the other family's launch,
            // whose workgroups return at once but queue for CU slots behind k_dp_regs<4>'s, it
            // started 110 us late at config 3)
            if (big) hipLaunchKernelGGL((k_dp_regs<8, 1>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, s0, d);
            hipLaunchKernelGGL(k_dp_sys<8>, dim3(DP_SYS_BLOCKS), dim3(DP_THREADS), 0, s0, d);
            if (!big) hipLaunchKernelGGL((k_dp_regs<8, 1>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, s0, d);
        }
        if (s->max_aln_len > 8) {
            hipStream_t s1 = side(1);
            // up to 32 k entries, two per wave: enough blocks that none loops for long
            if (big) hipLaunchKernelGGL((k_dp_regs<16, 2>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, s1, d);
            hipLaunchKernelGGL(k_dp_sys<16>, dim3(4 * DP_SYS_BLOCKS), dim3(DP_THREADS), 0, s1, d);
            if (!big) hipLaunchKernelGGL((k_dp_regs<16, 2>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, s1, d);
        }
        if (s->max_aln_len > 16) {
            hipStream_t s2 = side(2);
            hipLaunchKernelGGL(k_dp_sys<64>, dim3(2 * DP_SYS_BLOCKS), dim3(DP_THREADS), 0, s2, d);
            hipLaunchKernelGGL((k_dp_regs<32, 3>), dim3(DP_REG_BLOCKS), dim3(DP_THREADS), 0, s2, d);
            if (s->max_aln_len > 32) {
                if (dp_rows_fit_lds(s->max_aln_len))
                    hipLaunchKernelGGL(k_dp_long<true>, dim3(DP_BLOCKS), dim3(DP_THREADS),
                                       dp_lds_bytes(s->max_aln_len), s2, d);
                else
                    hipLaunchKernelGGL(k_dp_long<false>, dim3(DP_BLOCKS), dim3(DP_THREADS), 0,
                                       s2, d);
            }
        }
        }
        for (int i = 0; i < 3; ++i)        // join
            if (forked & (1 << i)) {
                HIP_TRY(hipEventRecord(s->dp_join[i], s->dp_stream[i]));
                HIP_TRY(hipStreamWaitEvent(st, s->dp_join[i], 0));
            }
        HIP_TRY(hipGetLastError());
    } else {
        // a shard without alignments (zero-step ones at most): the store still follows
        if (cx && cx->mode == 1)
            hipLaunchKernelGGL(k_store_snapshot, dim3(p_blocks), dim3(256), 0, st, d_counts, (int)n_paths,
                               cx->d_g1_tmp);
        if (s->profiling && s->prof_all) HIP_TRY(hipEventRecord(ev[2], st));
    }
    if (cx && cx->mode == 1)
        hipLaunchKernelGGL(k_store_paths, dim3((unsigned)n_paths), dim3(256), 0, st, d_path_off, d_path_steps,
                           s->d_order, (int)n_paths, d_counts, cx->d_g1_tmp, cx->d_slots, s->d_st_steps,
                           s->d_st_len, s->d_st_pass, s->d_st_g1, s->d_st_bitsok, s->st_cap, s->d_status);
    if (children)
        hipLaunchKernelGGL(k_child_resolve, dim3(p_blocks), dim3(256), 0, st, cx->batch, (int)n_paths,
                           s->n_empty, d_counts, d_user_counts, s->d_status, status_copy);
    else
        hipLaunchKernelGGL(k_unpermute, dim3(p_blocks), dim3(256), 0, st, d_counts, s->d_order,
                           (int)n_paths, d_user_counts, s->d_status, status_copy);
    HIP_TRY(hipGetLastError());
    if (s->profiling) {
        if (s->prof_all) HIP_TRY(hipEventRecord(ev[3], st));
        ++s->ev_calls;
    }
    s->last_stream = st;
    s->have_last = true;
    return GFAL_OK;
}

int gfal_scorer_score_device(gfal_scorer *s, const int32_t *d_path_off,
                             const int32_t *d_path_steps, int32_t n_paths,
                             int64_t total_steps, int32_t max_path_len, int filter,
                             uint32_t *d_counts, void *hip_stream)
{
    return no_throw([&] {
        return score_device_impl(s, d_path_off, d_path_steps, n_paths, total_steps, max_path_len,
                                 filter, d_counts, hip_stream, nullptr);
    });
}

// status words of a finished run -> return code
static int status_to_code(const gfal_scorer *s, const uint32_t *host)
{
    if (host[0] & (ST_BAD_LEN | ST_BAD_ID)) {
        set_err("device-side validation failed (status 0x%x)", host[0]);
        return GFAL_E_RANGE;
    }
    if (host[0] & ST_BAD_CHILD) {
        set_err("children batch: a parent reference, a store slot or a path length is invalid "
                "(parents must be stored or earlier in the batch, at least as long as the longest "
                "alignment, and the child within the stated maximum)");
        return GFAL_E_ARG;
    }
    if (host[0] & ST_DP_OVERFLOW) {
        set_err("exact-DP worklist overflow (%llu pairs, capacity %u)",
                (unsigned long long)host[2] | ((unsigned long long)host[3] << 32), s->wl_capacity);
        return GFAL_E_NOMEM;
    }
    return GFAL_OK;
}

int gfal_scorer_sync_status(gfal_scorer *s)
{
    if (!s) return GFAL_E_ARG;
    if (!s->have_last) return GFAL_OK;
    HIP_TRY(hipSetDevice(s->device));
    uint32_t host[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(host, s->d_status, sizeof(host), hipMemcpyDeviceToHost,
                           s->last_stream));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    const int rc = status_to_code(s, host);
    if (rc == GFAL_E_NOMEM) {
        // the list was too short for this batch: make room, so that the caller's
        // next attempt with the same batch fits (the stream is idle here)
        const unsigned long long need = (unsigned long long)host[2] | ((unsigned long long)host[3] << 32);
        const int grown = grow_worklist(s, need);
        if (grown == GFAL_OK)
            set_err("exact-DP worklist overflow (%llu pairs): the list has been grown to %u entries, "
                    "submit the batch again", need, s->wl_capacity);
    }
    return rc;
}

// Blocking-API staging without the wait: paths into the pinned buffer and onto the
// device, the kernels enqueued on the scorer's stream, counters left in
// s->d_counts[0 .. 3P) and the status words behind them (gfal_group_score).
// slots != NULL: the paths are also kept in the store (search mode), slots[p] or -1;
// they travel behind the steps in the same staging buffer.
static int score_stage(gfal_scorer *s, const int32_t *path_off, const int32_t *path_steps, int32_t P,
                       int32_t max_len, int filter, const int32_t *slots = nullptr)
{
    const int64_t total = path_off[P];
    const size_t n_paths_in = (size_t)P + 1 + (size_t)total;
    const size_t n_in = n_paths_in + (slots ? (size_t)P : 0), n_out = (size_t)3 * P + 4;
    int rc;
    if (s->have_last && (n_in > s->path_off_cap || n_out > s->counts_cap))
        HIP_TRY(hipStreamSynchronize(s->last_stream));
    if ((rc = dev_reserve(&s->d_path_off, &s->path_off_cap, n_in))) return rc;
    if ((rc = dev_reserve(&s->d_counts, &s->counts_cap, n_out))) return rc;
    if ((rc = pinned_reserve(&s->h_in, &s->h_in_cap, n_in))) return rc;
    if ((rc = pinned_reserve(&s->h_out, &s->h_out_cap, n_out))) return rc;
    if (path_off != s->h_in) {      // (a re-run stages from this very buffer)
        memcpy(s->h_in, path_off, ((size_t)P + 1) * sizeof(int32_t));
        memcpy(s->h_in + P + 1, path_steps, (size_t)total * sizeof(int32_t));
        if (slots) memcpy(s->h_in + n_paths_in, slots, (size_t)P * sizeof(int32_t));
    }
    HIP_TRY(hipMemcpyAsync(s->d_path_off, s->h_in, n_in * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    ExtraCtx cx;
    if (slots) {
        if (s->have_last && (size_t)P > s->child_tmp_cap) HIP_TRY(hipStreamSynchronize(s->last_stream));
        if ((rc = dev_reserve(&s->d_child_tmp, &s->child_tmp_cap, (size_t)P))) return rc;
        cx.mode = 1;
        cx.d_slots = s->d_path_off + n_paths_in;
        cx.d_g1_tmp = reinterpret_cast<uint32_t *>(s->d_child_tmp);
    }
    rc = score_device_impl(s, s->d_path_off, s->d_path_off + P + 1, P, total, max_len, filter, s->d_counts,
                           s->stream, s->d_counts + (size_t)3 * P, slots ? &cx : nullptr);
    s->last_stream = s->stream;
    s->have_last = true;
    return rc;
}

// The content table of a scorer: one entry per distinct step sequence among the
// resident alignments (open addressing on the content hash), and every lane's entry.
// Built on the device when the scorer is created: k_tile looks the windows of the
// candidate paths up in it, k_scan3 compares entry indices, k_child reads multiplicities.
static int build_content_table(gfal_scorer *s)
{
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t n_slots = (uint32_t)s->n_items * WAVE;
    uint32_t slots_pow2 = 1024;
    while ((uint64_t)slots_pow2 < 2ull * std::max<uint32_t>(n_slots, 1u)) slots_pow2 <<= 1;
    s->ct_mask = slots_pow2 - 1u;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_ct_key), (size_t)slots_pow2 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_ct_hash), (size_t)slots_pow2 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_ct_mult), (size_t)slots_pow2 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_ct_rec), (size_t)slots_pow2 * sizeof(uint4)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_pair_bits), ((size_t)1 << PAIR_BITS_LOG2) / 8));
    hipStream_t st = s->stream;
    HIP_TRY(hipMemsetAsync(s->d_pair_bits, 0, ((size_t)1 << PAIR_BITS_LOG2) / 8, st));
    HIP_TRY(hipMemsetAsync(s->d_ct_key, 0xFF, (size_t)slots_pow2 * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(s->d_ct_hash, 0, (size_t)slots_pow2 * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(s->d_ct_mult, 0, (size_t)slots_pow2 * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(s->d_ct_rec, 0xFF, (size_t)slots_pow2 * sizeof(uint4), st));
    if (n_slots > 0) {
        const Items items{s->d_item_steps, s->d_item_base, s->d_item_len, s->d_item_pairs,
                          s->d_item_pbase, s->n_items, s->d_item_common, s->d_item_hdr, s->d_item_weight};
        const unsigned blocks = (n_slots + 255u) / 256u;
        hipLaunchKernelGGL(k_ct_build, dim3(blocks), dim3(256), 0, st, items, s->d_slot_orig, n_slots,
                           s->d_item_hash, s->d_ct_key, s->d_ct_hash, s->d_ct_mult, s->ct_mask, s->d_ct_rec,
                           s->d_rec3, s->d_item_r3, s->d_pair_bits);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    return GFAL_OK;
}

// The search-mode index of a scorer (inverted lists by node), built on the device
// from the resident items at the first use.
static int build_child_index(gfal_scorer *s)
{
    if (s->child_index) return GFAL_OK;
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t n_slots = (uint32_t)s->n_items * WAVE;
    const int n_loc = std::max(1, (int)s->n_local) * N_CLASSES;      // one list per (node, length class)
    uint32_t *cnt = nullptr, *cursor = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&cnt), (size_t)(n_loc + 1) * sizeof(uint32_t)));
    struct Tmp {
        uint32_t *&a, *&b;
        ~Tmp()
        {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
        }
    } tmp{cnt, cursor};
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&cursor), (size_t)(n_loc + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_inv_off), (size_t)(n_loc + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_inv_ent),
                      std::max<size_t>((size_t)s->n_item_u16, 1) * sizeof(uint4)));
    hipStream_t st = s->stream;
    HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)(n_loc + 1) * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(s->d_inv_off, 0, (size_t)(n_loc + 1) * sizeof(uint32_t), st));
    if (n_slots > 0) {
        const Items items{s->d_item_steps, s->d_item_base, s->d_item_len, s->d_item_pairs,
                          s->d_item_pbase, s->n_items, s->d_item_common, s->d_item_hdr, s->d_item_weight};
        const unsigned blocks = (n_slots + 255u) / 256u;
        hipLaunchKernelGGL(k_inv_build, dim3(blocks), dim3(256), 0, st, items, s->d_slot_orig, n_slots, cnt,
                           (uint4 *)nullptr);
        hipLaunchKernelGGL(k_inv_scan, dim3(1), dim3(1024), 0, st, cnt, n_loc, s->d_inv_off, cursor);
        hipLaunchKernelGGL(k_inv_build, dim3(blocks), dim3(256), 0, st, items, s->d_slot_orig, n_slots, cursor,
                           s->d_inv_ent);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    {
        // the longest inverted list of a node bounds the per-path bitmaps (a list of more
        // than 4 M alignments -- e.g. the one node that stands for everything outside the
        // universe, which no path can start on -- is not remembered: its paths recompute)
        std::vector<uint32_t> off((size_t)n_loc + 1);
        HIP_TRY(hipMemcpy(off.data(), s->d_inv_off, off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t longest = 0;
        for (int v = 0; v + N_CLASSES <= n_loc; v += N_CLASSES) {
            const uint32_t len = off[(size_t)v + N_CLASSES] - off[(size_t)v];
            uint32_t max_list = 4u << 20;
            if (const char *env = getenv("GFAL_BITS_MAX_LIST")) max_list = (uint32_t)std::max(1ll, atoll(env));   // (tests)
            if (len <= max_list) longest = std::max(longest, len);
        }
        s->bits_words = (longest + 31u) / 32u;
        if (getenv("GFAL_NO_INHERIT")) s->bits_words = 0;
        if (s->bits_words) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_wl_pos), (size_t)s->wl_capacity * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_wl_pos_sorted),
                              (size_t)s->wl_capacity * sizeof(uint32_t)));
        }
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_child<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - CHILD_STATIC_LDS));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_child<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - CHILD_STATIC_LDS));
    s->child_index = true;
    return GFAL_OK;
}

// Room for `cap` stored paths (grows, keeps what is there).
static int store_reserve(gfal_scorer *s, int64_t cap)
{
    if (cap <= s->st_cap) return GFAL_OK;
    HIP_TRY(hipSetDevice(s->device));
    if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
    int32_t *steps = nullptr, *len = nullptr;
    uint32_t *pass = nullptr, *g1 = nullptr;
    const size_t n = (size_t)cap;
    if (hipMalloc(reinterpret_cast<void **>(&steps), n * STORE_STRIDE * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&len), n * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&pass), n * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&g1), n * sizeof(uint32_t)) != hipSuccess) {
        for (void *b : {(void *)steps, (void *)len, (void *)pass, (void *)g1})
            if (b) (void)hipFree(b);
        (void)hipGetLastError();
        set_err("path store: cannot allocate %lld slots", (long long)cap);
        return GFAL_E_NOMEM;
    }
    HIP_TRY(hipMemset(len, 0, n * sizeof(int32_t)));
    if (s->bits_words) {
        uint32_t *bits = nullptr;
        int32_t *ok = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&bits), n * s->bits_words * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&ok), n * sizeof(int32_t)) != hipSuccess) {
            for (void *b : {(void *)steps, (void *)len, (void *)pass, (void *)g1, (void *)bits, (void *)ok})
                if (b) (void)hipFree(b);
            (void)hipGetLastError();
            set_err("path store: cannot allocate %lld slots", (long long)cap);
            return GFAL_E_NOMEM;
        }
        HIP_TRY(hipMemset(ok, 0, n * sizeof(int32_t)));
        if (s->st_cap > 0) {
            const size_t o = (size_t)s->st_cap;
            HIP_TRY(hipMemcpy(bits, s->d_st_bits, o * s->bits_words * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(ok, s->d_st_bitsok, o * sizeof(int32_t), hipMemcpyDeviceToDevice));
            (void)hipFree(s->d_st_bits);
            (void)hipFree(s->d_st_bitsok);
        }
        s->d_st_bits = bits;
        s->d_st_bitsok = ok;
    }
    if (s->st_cap > 0) {
        const size_t o = (size_t)s->st_cap;
        HIP_TRY(hipMemcpy(steps, s->d_st_steps, o * STORE_STRIDE * sizeof(int32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(len, s->d_st_len, o * sizeof(int32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(pass, s->d_st_pass, o * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(g1, s->d_st_g1, o * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        (void)hipFree(s->d_st_steps);
        (void)hipFree(s->d_st_len);
        (void)hipFree(s->d_st_pass);
        (void)hipFree(s->d_st_g1);
    }
    s->d_st_steps = steps;
    s->d_st_len = len;
    s->d_st_pass = pass;
    s->d_st_g1 = g1;
    s->st_cap = cap;
    return GFAL_OK;
}

// A children batch on one shard: [parent | step | slot] in, the paths written out on
// the device, then the usual pipeline with k_child in place of the scans.  Counters
// are left in s->d_counts like score_stage does.
static int children_stage(gfal_scorer *s, const int32_t *parent, const int32_t *step, const int32_t *slot,
                          int32_t n, int32_t max_len, bool direct_out)
{
    int rc;
    if ((rc = build_child_index(s))) return rc;
    if (s->st_cap < 1) {
        set_err("children batch without a path store (gfal_group_store_reserve)");
        return GFAL_E_ARG;
    }
    const size_t n_in = (size_t)3 * n, n_out = (size_t)3 * n + 4;
    const size_t n_paths_buf = (size_t)n + 1;      // offsets only: k_prep reads the paths from the store
    if (s->have_last && (n_in > s->child_in_cap || (size_t)4 * n > s->child_tmp_cap ||
                         n_paths_buf > s->path_off_cap || n_out > s->counts_cap))
        HIP_TRY(hipStreamSynchronize(s->last_stream));
    if ((rc = dev_reserve(&s->d_child_in, &s->child_in_cap, n_in))) return rc;
    if ((rc = dev_reserve(&s->d_child_tmp, &s->child_tmp_cap, (size_t)4 * n))) return rc;
    if ((rc = dev_reserve(&s->d_path_off, &s->path_off_cap, n_paths_buf))) return rc;
    if ((rc = dev_reserve(&s->d_counts, &s->counts_cap, n_out))) return rc;
    if ((rc = pinned_reserve(&s->h_in, &s->h_in_cap, n_in))) return rc;
    if ((rc = pinned_reserve(&s->h_out, &s->h_out_cap, n_out))) return rc;
    if (parent != s->h_in) {
        memcpy(s->h_in, parent, (size_t)n * sizeof(int32_t));
        memcpy(s->h_in + n, step, (size_t)n * sizeof(int32_t));
        memcpy(s->h_in + 2 * (size_t)n, slot, (size_t)n * sizeof(int32_t));
    }
    hipStream_t st = s->stream;
    if (s->have_last && s->last_stream != st) {
        HIP_TRY(hipEventRecord(s->order_ev, s->last_stream));
        HIP_TRY(hipStreamWaitEvent(st, s->order_ev, 0));
    }
    // pinned host memory is device-visible: the first kernel reads the batch from it and,
    // when this shard's counters are the result (no sum over shards), the last one writes
    // them into the pinned output -- no copy operations around a children call
    int32_t *h_in_dev = nullptr;
    uint32_t *h_out_dev = nullptr;
    if (getenv("GFAL_CHILD_COPIES") == nullptr) {
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h_in_dev), s->h_in, 0) != hipSuccess) h_in_dev = nullptr;
        if (!direct_out ||
            hipHostGetDevicePointer(reinterpret_cast<void **>(&h_out_dev), s->h_out, 0) != hipSuccess)
            h_out_dev = nullptr;
        (void)hipGetLastError();
    }
    if (!h_in_dev)
        HIP_TRY(hipMemcpyAsync(s->d_child_in, s->h_in, n_in * sizeof(int32_t), hipMemcpyHostToDevice, st));
    ExtraCtx cx;
    cx.mode = 2;
    ChildBatch &b = cx.batch;
    b.parent = s->d_child_in;
    b.step = s->d_child_in + n;
    b.slot = s->d_child_in + 2 * (size_t)n;
    b.n = n;
    b.st_steps = s->d_st_steps;
    b.st_len = s->d_st_len;
    b.st_pass = s->d_st_pass;
    b.st_g1 = s->d_st_g1;
    b.st_cap = s->st_cap;
    b.root = s->d_child_tmp;
    b.depth = s->d_child_tmp + n;
    b.dpass = reinterpret_cast<uint32_t *>(s->d_child_tmp + 2 * (size_t)n);
    b.dg1 = reinterpret_cast<uint32_t *>(s->d_child_tmp + 3 * (size_t)n);
    b.st_bits = s->d_st_bits;
    b.st_bitsok = s->d_st_bitsok;
    b.bits_words = s->bits_words;
    hipLaunchKernelGGL(k_child_len, dim3(1), dim3(1024), 0, st, b, (int)max_len, (int)s->max_aln_len,
                       s->d_path_off, s->d_status, (const int32_t *)h_in_dev, s->d_child_in);
    HIP_TRY(hipGetLastError());
    ++s->n_children_calls;
    uint32_t *const out = h_out_dev ? h_out_dev : s->d_counts;
    rc = score_device_impl(s, s->d_path_off, s->d_path_off + n + 1, n, (int64_t)n * max_len, max_len, 1,
                           out, st, out + (size_t)3 * n, &cx);
    s->last_stream = st;
    s->have_last = true;
    s->out_on_host = h_out_dev != nullptr;
    return rc;
}

static int score_range(gfal_scorer *s, const int32_t *path_off,
                       const int32_t *path_steps, int32_t lo, int32_t hi,
                       int32_t max_len, int filter, uint32_t *bad, uint32_t *good,
                       uint32_t *unaligned)
{
    const int32_t P = hi - lo;
    const int64_t step0 = path_off[lo];
    const int64_t total = path_off[hi] - step0;
    const size_t n_in = (size_t)P + 1 + (size_t)total, n_out = (size_t)3 * P + 4;
    int rc;
    if ((rc = dev_reserve(&s->d_path_off, &s->path_off_cap, n_in))) return rc;
    if ((rc = dev_reserve(&s->d_counts, &s->counts_cap, n_out))) return rc;
    if ((rc = pinned_reserve(&s->h_in, &s->h_in_cap, n_in))) return rc;
    if ((rc = pinned_reserve(&s->h_out, &s->h_out_cap, n_out))) return rc;
    for (int32_t i = 0; i <= P; ++i) s->h_in[i] = (int32_t)(path_off[lo + i] - step0);
    memcpy(s->h_in + P + 1, path_steps + step0, (size_t)total * sizeof(int32_t));
    // copy in, the kernels, copy out: enqueued directly, or -- for the small
    // batches a search submits, where the GPU otherwise waits for the host between
    // fifteen short dependent launches -- captured and launched as one graph
    auto enqueue = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(s->d_path_off, s->h_in, n_in * sizeof(int32_t),
                               hipMemcpyHostToDevice, s->stream));
        int r = score_device_impl(s, s->d_path_off, s->d_path_off + P + 1, P, total, max_len,
                                  filter, s->d_counts, s->stream, s->d_counts + (size_t)3 * P);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(s->h_out, s->d_counts, n_out * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, s->stream));
        return GFAL_OK;
    };
    bool launched = false;
    if (s->use_graphs && !s->profiling && P <= 4096) {
        const ImageLayout L = make_layout(s->n_local, max_len);
        if ((rc = ensure_call_buffers(s, P, L))) return rc;    // nothing may allocate while capturing
        if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int r = enqueue();
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(s->stream, &graph);
            if (r == GFAL_OK && e == hipSuccess && graph) {
                if (s->graph_exec) {
                    hipGraphNode_t err_node = nullptr;
                    hipGraphExecUpdateResult how;
                    if (hipGraphExecUpdate(s->graph_exec, graph, &err_node, &how) != hipSuccess) {
                        (void)hipGraphExecDestroy(s->graph_exec);    // topology changed
                        s->graph_exec = nullptr;
                    }
                }
                if (!s->graph_exec &&
                    hipGraphInstantiate(&s->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess)
                    s->graph_exec = nullptr;
                if (s->graph_exec && hipGraphLaunch(s->graph_exec, s->stream) == hipSuccess)
                    launched = true;
            }
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (!launched) {       // graphs do not work here: direct launches from now on
                s->use_graphs = false;
                if (r != GFAL_OK) return r;
            }
        } else {
            (void)hipGetLastError();
            s->use_graphs = false;
        }
    }
    if (!launched && (rc = enqueue())) return rc;
    s->last_stream = s->stream;
    s->have_last = true;
    HIP_TRY(hipStreamSynchronize(s->stream));
    memcpy(s->cached_status, s->h_out + (size_t)3 * P, 4 * sizeof(uint32_t));
    s->status_cached = true;
    rc = status_to_code(s, s->h_out + (size_t)3 * P);
    if (rc == GFAL_E_NOMEM) {
        // Worklist overflow.  The pass counted every pair it wanted to append, so
        // the need is known exactly: grow the lists ONCE to fit (they stay that
        // size for the scorer's lifetime, so a search pays this on the first deep
        // batch only) and run the batch again.  Only a batch that needs more than
        // WL_MAX_ENTRIES is split (a single path always fits: at most one entry
        // per alignment, and the lists hold at least that).
        const uint32_t *st4 = s->h_out + (size_t)3 * P;
        const unsigned long long need = (unsigned long long)st4[2] | ((unsigned long long)st4[3] << 32);
        ++s->n_overflow_reruns;
        if (need + need / 4 + 1024 <= WL_MAX_ENTRIES && getenv("GFAL_DEBUG_WL_NO_GROW") == nullptr) {
            if ((rc = grow_worklist(s, need))) return rc;
            return score_range(s, path_off, path_steps, lo, hi, max_len, filter, bad, good, unaligned);
        }
        if (P == 1) return rc;
        int32_t mid = lo + P / 2;
        rc = score_range(s, path_off, path_steps, lo, mid, max_len, filter, bad, good,
                         unaligned);
        if (rc) return rc;
        return score_range(s, path_off, path_steps, mid, hi, max_len, filter, bad, good,
                           unaligned);
    }
    if (rc) return rc;
    memcpy(bad + lo, s->h_out, (size_t)P * sizeof(uint32_t));
    memcpy(good + lo, s->h_out + P, (size_t)P * sizeof(uint32_t));
    if (unaligned) memcpy(unaligned + lo, s->h_out + (size_t)2 * P, (size_t)P * sizeof(uint32_t));
    return GFAL_OK;
}

int gfal_scorer_score(gfal_scorer *s, const int32_t *path_off,
                      const int32_t *path_steps, int32_t n_paths, int filter,
                      uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    if (!s || n_paths < 0) return GFAL_E_ARG;
    if (n_paths == 0) return GFAL_OK;
    if (!path_off || !path_steps || !bad || !good || path_off[0] != 0) return GFAL_E_ARG;
    int32_t max_len = 1;
    for (int32_t p = 0; p < n_paths; ++p) {
        int64_t n = (int64_t)path_off[p + 1] - path_off[p];
        if (n < 1 || n > GFAL_MAX_STEPS) {
            set_err("path %d has %lld steps (allowed 1..%d)", p, (long long)n,
                    GFAL_MAX_STEPS);
            return GFAL_E_RANGE;
        }
        max_len = std::max(max_len, (int32_t)n);
    }
    for (int64_t t = 0; t < path_off[n_paths]; ++t) {
        int32_t st = path_steps[t];
        if (st < 0 || ((st & ~GFAL_STEP_OTHER) >> 1) >= s->n_nodes) {
            set_err("path step %lld: node id out of range", (long long)t);
            return GFAL_E_RANGE;
        }
    }
    HIP_TRY(hipSetDevice(s->device));
    ++s->n_score_calls;
    return no_throw([&] {
        return score_range(s, path_off, path_steps, 0, n_paths, max_len, filter, bad, good,
                           unaligned);
    });
}

// --------------------------------------------------------------------------
// Scorer groups: the shards of one alignment set on the GPUs of one node, in one
// process.  Every device scores the whole batch against its shard; the per-path
// counters (uint32[3P]: 120 KB at 10 k paths, latency-bound) are summed where
// they are, by ONE RCCL all-reduce over xGMI (ncclAllReduce inside a group call,
// one communicator per device from ncclCommInitAll), and only device 0's copy
// travels to the host.  RCCL is loaded on first use (dlopen: a scorer that is
// never grouped does not pay for loading it).  When RCCL cannot serve the group
// -- library missing, or two shards share a device (test rigs) -- the counters
// are copied out per shard and added on the host: same integers either way.
// --------------------------------------------------------------------------
namespace {
struct Rccl {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *lib = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd;
    }
};
constexpr int NCCL_UINT32 = 3, NCCL_SUM = 0;      // ncclDataType_t / ncclRedOp_t (rccl.h)
}  // namespace

struct gfal_group {
    std::vector<gfal_scorer *> shards;
    Rccl rccl;
    std::vector<Rccl::comm_t> comms;          // empty: host sums
    std::vector<uint32_t> h_sum;              // host-sum scratch
    // the batch between gfal_group_score_begin and _end
    bool pending = false;
    int32_t pend_paths = 0, pend_max_len = 0;
    int pend_filter = 0;
    int pend_kind = 0;        // 0 plain, 1 paths kept in the store, 2 children batch
};

int gfal_group_create(gfal_scorer *const *scorers, int n, gfal_group **out)
{
    if (!out || !scorers || n < 1) return GFAL_E_ARG;
    *out = nullptr;
    return no_throw([&] {
        std::unique_ptr<gfal_group> g(new gfal_group());
        std::vector<int> devs;
        bool distinct = true;
        for (int i = 0; i < n; ++i) {
            if (!scorers[i]) return GFAL_E_ARG;
            for (int d : devs) distinct &= d != scorers[i]->device;
            devs.push_back(scorers[i]->device);
            g->shards.push_back(scorers[i]);
        }
        const char *off = getenv("GFAL_GROUP_HOST_SUM");
        // (a group of one has nothing to sum; GFAL_GROUP_RCCL_SINGLE=1 runs the one-rank
        // all-reduce anyway: the only RCCL a one-GPU test box can exercise)
        const char *single = getenv("GFAL_GROUP_RCCL_SINGLE");
        if (distinct && (n > 1 || (single && atoi(single))) && !(off && atoi(off)) && g->rccl.load()) {
            g->comms.assign((size_t)n, nullptr);
            const int rc = g->rccl.CommInitAll(g->comms.data(), n, devs.data());
            if (rc != 0) {
                (void)hipGetLastError();
                g->comms.clear();             // RCCL cannot serve this group: sums on the host
            }
        }
        *out = g.release();
        return GFAL_OK;
    });
}

void gfal_group_destroy(gfal_group *g)
{
    if (!g) return;
    for (Rccl::comm_t c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    delete g;
}

int gfal_group_uses_rccl(const gfal_group *g) { return g && !g->comms.empty() ? 1 : 0; }

// every device: paths in, the kernels, the all-reduce, counters and status words
// on their way out -- nothing waited for
// (kind 1: `slots` behind the paths; kind 2: path_off / path_steps / slots are the
// batch's parent / step / slot arrays)
static int group_enqueue(gfal_group *g, const int32_t *path_off, const int32_t *path_steps,
                         const int32_t *slots = nullptr)
{
    const size_t D = g->shards.size();
    const int32_t P = g->pend_paths;
    for (size_t d = 0; d < D; ++d) {
        gfal_scorer *s = g->shards[d];
        HIP_TRY(hipSetDevice(s->device));
        s->out_on_host = false;
        const int rc = g->pend_kind == 2
                           ? children_stage(s, path_off, path_steps, slots, P, g->pend_max_len, D == 1 && g->comms.empty())
                           : score_stage(s, path_off, path_steps, P, g->pend_max_len, g->pend_filter,
                                         g->pend_kind == 1 ? slots : nullptr);
        if (rc) return rc;
    }
    const size_t n_cnt = (size_t)3 * P;
    if (!g->comms.empty()) {
        // sum the counters where they are: one all-reduce over the group
        int rc = g->rccl.GroupStart();
        for (size_t d = 0; d < D && rc == 0; ++d) {
            gfal_scorer *s = g->shards[d];
            (void)hipSetDevice(s->device);
            rc = g->rccl.AllReduce(s->d_counts, s->d_counts, n_cnt, NCCL_UINT32, NCCL_SUM, g->comms[d],
                                   s->stream);
        }
        const int rc2 = g->rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) {
            set_err("RCCL all-reduce failed: %s",
                    g->rccl.GetErrorString ? g->rccl.GetErrorString(rc ? rc : rc2) : "?");
            return GFAL_E_HIP;
        }
    }
    // counters (device 0's after an all-reduce, every shard's otherwise) and status words out
    for (size_t d = 0; d < D; ++d) {
        gfal_scorer *s = g->shards[d];
        HIP_TRY(hipSetDevice(s->device));
        // (k_unpermute put the status words behind the counters BEFORE the all-reduce
        // touched the first 3P words: one copy takes both)
        if (s->out_on_host) {
            // (a children batch on the group's only shard: already there)
        } else if (g->comms.empty() || d == 0)
            HIP_TRY(hipMemcpyAsync(s->h_out, s->d_counts, (n_cnt + 4) * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   s->stream));
        else
            HIP_TRY(hipMemcpyAsync(s->h_out + n_cnt, s->d_counts + n_cnt, 4 * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, s->stream));
    }
    return GFAL_OK;
}

static int group_begin_impl(gfal_group *g, const int32_t *path_off, const int32_t *path_steps,
                            int32_t n_paths, int filter, const int32_t *slots = nullptr)
{
    if (g->pending) {
        set_err("gfal_group_score_begin: the previous batch has not been collected");
        return GFAL_E_ARG;
    }
    int32_t max_len = 1;
    for (int32_t p = 0; p < n_paths; ++p) {
        const int64_t n = (int64_t)path_off[p + 1] - path_off[p];
        if (n < 1 || n > GFAL_MAX_STEPS) {
            set_err("path %d has %lld steps (allowed 1..%d)", p, (long long)n, GFAL_MAX_STEPS);
            return GFAL_E_RANGE;
        }
        max_len = std::max(max_len, (int32_t)n);
    }
    for (int64_t t = 0; t < path_off[n_paths]; ++t) {
        const int32_t st = path_steps[t];
        if (st < 0 || ((st & ~GFAL_STEP_OTHER) >> 1) >= g->shards[0]->n_nodes) {
            set_err("path step %lld: node id out of range", (long long)t);
            return GFAL_E_RANGE;
        }
    }
    g->pend_paths = n_paths;
    g->pend_max_len = max_len;
    g->pend_filter = filter;
    g->pend_kind = slots ? 1 : 0;
    if (slots)
        for (gfal_scorer *s : g->shards)
            if (s->st_cap < 1) {
                set_err("gfal_group_score_store_begin without a path store (gfal_group_store_reserve)");
                return GFAL_E_ARG;
            }
    for (gfal_scorer *s : g->shards) ++s->n_score_calls;
    const int rc = group_enqueue(g, path_off, path_steps, slots);
    g->pending = rc == GFAL_OK;
    return rc;
}

static int group_children_impl(gfal_group *g, int32_t n, const int32_t *parent, const int32_t *step,
                               const int32_t *slot, int32_t max_len)
{
    if (g->pending) {
        set_err("gfal_group_score_children_begin: the previous batch has not been collected");
        return GFAL_E_ARG;
    }
    if (max_len < 2 || max_len > GFAL_MAX_STEPS) return GFAL_E_RANGE;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t st = step[i];
        if (st < 0 || ((st & ~GFAL_STEP_OTHER) >> 1) >= g->shards[0]->n_nodes) {
            set_err("child %d: node id out of range", i);
            return GFAL_E_RANGE;
        }
        if (parent[i] < 0 && ~parent[i] >= i) {
            set_err("child %d: an in-batch parent must come earlier in the batch", i);
            return GFAL_E_ARG;
        }
    }
    g->pend_paths = n;
    g->pend_max_len = max_len;
    g->pend_filter = 1;
    g->pend_kind = 2;
    for (gfal_scorer *s : g->shards) ++s->n_score_calls;
    const int rc = group_enqueue(g, parent, step, slot);
    g->pending = rc == GFAL_OK;
    return rc;
}

// The counters of the batch that just finished on every shard (P paths), summed
// over the shards unless the devices already did that.
static void group_gather(gfal_group *g, int32_t P, uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    const size_t D = g->shards.size();
    const size_t n_cnt = (size_t)3 * P;
    const uint32_t *h = g->shards[0]->h_out;
    if (g->comms.empty() && D > 1) {
        g->h_sum.assign(n_cnt, 0u);
        for (size_t d = 0; d < D; ++d)
            for (size_t k = 0; k < n_cnt; ++k) g->h_sum[k] += g->shards[d]->h_out[k];
        h = g->h_sum.data();
    }
    memcpy(bad, h, (size_t)P * sizeof(uint32_t));
    memcpy(good, h + P, (size_t)P * sizeof(uint32_t));
    if (unaligned) memcpy(unaligned, h + 2 * (size_t)P, (size_t)P * sizeof(uint32_t));
}

// May the lists grow to `need` pairs, or must the batch be split?  (GFAL_DEBUG_WL_NO_GROW
// forces the split path in tests.)
static bool worklist_may_grow(unsigned long long need)
{
    return need + need / 4 + 1024 <= WL_MAX_ENTRIES && getenv("GFAL_DEBUG_WL_NO_GROW") == nullptr;
}

// Paths [lo, hi) of a batch whose exact-DP need is beyond what the lists may grow to,
// run on their own (blocking), halved again while they still overflow -- what
// score_range does for the one-scorer call.  `in` is a host copy of the whole batch in
// the staging layout: [offsets | steps | slots], or [parent | step | slot] of a children
// batch (parents precede their children in one, so the halves are run in order: the first
// half is in the store when the second names it -- by slot, see below).
static int group_run_range(gfal_group *g, const std::vector<int32_t> &in, int32_t P0, int32_t lo, int32_t hi,
                           uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    const int32_t P = hi - lo;
    const size_t D = g->shards.size();
    g->pend_paths = P;
    int rc;
    std::vector<int32_t> off;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (g->pend_kind == 2) {
            // a parent inside the batch is named by its index there: by its index in this
            // piece, or -- scored by an earlier piece -- by the slot it was kept in
            const int32_t *slot = in.data() + 2 * (size_t)P0;
            off.resize((size_t)P);
            for (int32_t k = 0; k < P; ++k) {
                int32_t par = in[(size_t)(lo + k)];
                if (par < 0) {
                    const int32_t j = ~par;
                    if (j >= lo)
                        par = ~(j - lo);
                    else if (j >= 0 && slot[j] >= 0)
                        par = slot[j];
                    else {
                        set_err("exact-DP worklist: the children batch cannot be split (a parent inside it is not kept)");
                        return GFAL_E_NOMEM;
                    }
                }
                off[(size_t)k] = par;
            }
            rc = group_enqueue(g, off.data(), in.data() + P0 + lo, slot + lo);
        } else {
            off.resize((size_t)P + 1);
            for (int32_t p = 0; p <= P; ++p) off[(size_t)p] = in[(size_t)(lo + p)] - in[(size_t)lo];
            const int32_t *steps = in.data() + P0 + 1;
            rc = group_enqueue(g, off.data(), steps + in[(size_t)lo], steps + in[(size_t)P0] + lo);
        }
        if (rc) return rc;
        bool again = false, split = false;
        for (size_t d = 0; d < D; ++d) {
            gfal_scorer *s = g->shards[d];
            HIP_TRY(hipSetDevice(s->device));
            HIP_TRY(hipStreamSynchronize(s->stream));
            const uint32_t *st4 = s->h_out + (size_t)3 * P;
            memcpy(s->cached_status, st4, 4 * sizeof(uint32_t));
            s->status_cached = true;
            rc = status_to_code(s, st4);
            if (rc == GFAL_E_NOMEM) {
                const unsigned long long need = (unsigned long long)st4[2] | ((unsigned long long)st4[3] << 32);
                ++s->n_overflow_reruns;
                if (!worklist_may_grow(need))
                    split = true;
                else if ((rc = grow_worklist(s, need)))
                    return rc;
                again = true;
            } else if (rc) {
                return rc;
            }
        }
        if (!again) {
            group_gather(g, P, bad + lo, good + lo, unaligned ? unaligned + lo : nullptr);
            return GFAL_OK;
        }
        if (split) {
            if (P == 1) {
                set_err("exact-DP worklist: one path needs more pairs than the lists may hold");
                return GFAL_E_NOMEM;
            }
            const int32_t mid = lo + P / 2;
            if ((rc = group_run_range(g, in, P0, lo, mid, bad, good, unaligned))) return rc;
            return group_run_range(g, in, P0, mid, hi, bad, good, unaligned);
        }
    }
    set_err("exact-DP worklist kept overflowing");
    return GFAL_E_NOMEM;
}

static int group_end_impl(gfal_group *g, uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    if (!g->pending) {
        set_err("gfal_group_score_end without a batch");
        return GFAL_E_ARG;
    }
    g->pending = false;
    const size_t D = g->shards.size();
    const int32_t P = g->pend_paths;
    const size_t n_cnt = (size_t)3 * P;
    for (int attempt = 0; attempt < 4; ++attempt) {
        bool again = false, split = false;
        for (size_t d = 0; d < D; ++d) {
            gfal_scorer *s = g->shards[d];
            HIP_TRY(hipSetDevice(s->device));
            HIP_TRY(hipStreamSynchronize(s->stream));
            const uint32_t *st4 = s->h_out + n_cnt;
            memcpy(s->cached_status, st4, 4 * sizeof(uint32_t));
            s->status_cached = true;
            const int rc = status_to_code(s, st4);
            if (rc == GFAL_E_NOMEM) {        // worklist overflow on this shard: grow, run the batch again
                const unsigned long long need = (unsigned long long)st4[2] | ((unsigned long long)st4[3] << 32);
                ++s->n_overflow_reruns;
                if (!worklist_may_grow(need)) {
                    split = true;
                } else {
                    const int grown = grow_worklist(s, need);
                    if (grown) return grown;
                }
                again = true;
            } else if (rc) {
                return rc;
            }
        }
        if (!again) {
            group_gather(g, P, bad, good, unaligned);
            return GFAL_OK;
        }
        // the batch is still in shard 0's pinned staging buffer: [offsets | steps | slots],
        // or [parent | step | slot] of a children batch (re-running one is idempotent:
        // the store receives the same paths and counters again)
        const int32_t *h_in = g->shards[0]->h_in;
        if (split) {
            // more pairs than the lists may ever hold: the batch in halves (the sub-batches
            // reuse the staging buffer, so the batch is copied out of it first)
            if (P == 1) {
                set_err("exact-DP worklist: one path needs more pairs than the lists may hold");
                return GFAL_E_NOMEM;
            }
            const size_t n_in = g->pend_kind == 2 ? (size_t)3 * P
                                                  : (size_t)P + 1 + (size_t)h_in[P] + (g->pend_kind == 1 ? (size_t)P : 0);
            const std::vector<int32_t> in(h_in, h_in + n_in);
            const int32_t mid = P / 2;
            int rc = group_run_range(g, in, P, 0, mid, bad, good, unaligned);
            if (!rc) rc = group_run_range(g, in, P, mid, P, bad, good, unaligned);
            g->pend_paths = P;
            return rc;
        }
        const int rc = g->pend_kind == 2 ? group_enqueue(g, h_in, h_in + P, h_in + 2 * (size_t)P)
                                         : group_enqueue(g, h_in, h_in + P + 1, h_in + P + 1 + h_in[P]);
        if (rc) return rc;
    }
    set_err("exact-DP worklist kept overflowing");
    return GFAL_E_NOMEM;
}

int gfal_group_score_begin(gfal_group *g, const int32_t *path_off, const int32_t *path_steps,
                           int32_t n_paths, int filter)
{
    if (!g || n_paths < 1) return GFAL_E_ARG;
    if (!path_off || !path_steps || path_off[0] != 0) return GFAL_E_ARG;
    return no_throw([&] { return group_begin_impl(g, path_off, path_steps, n_paths, filter); });
}

int gfal_group_store_reserve(gfal_group *g, int64_t n_slots)
{
    if (!g || n_slots < 1 || n_slots > ((int64_t)1 << 30)) return GFAL_E_ARG;
    return no_throw([&] {
        if (g->pending) {
            set_err("gfal_group_store_reserve while a batch is in flight");
            return GFAL_E_ARG;
        }
        for (gfal_scorer *s : g->shards) {
            int rc = build_child_index(s);
            if (rc == GFAL_OK) rc = store_reserve(s, n_slots);
            if (rc) return rc;
        }
        return GFAL_OK;
    });
}

int gfal_group_score_store_begin(gfal_group *g, const int32_t *path_off, const int32_t *path_steps,
                                 int32_t n_paths, const int32_t *slots)
{
    if (!g || n_paths < 1 || !slots) return GFAL_E_ARG;
    if (!path_off || !path_steps || path_off[0] != 0) return GFAL_E_ARG;
    return no_throw([&] { return group_begin_impl(g, path_off, path_steps, n_paths, 1, slots); });
}

int gfal_group_score_children_begin(gfal_group *g, int32_t n, const int32_t *parent, const int32_t *step,
                                    const int32_t *slot, int32_t max_path_len)
{
    if (!g || n < 1 || !parent || !step || !slot) return GFAL_E_ARG;
    return no_throw([&] { return group_children_impl(g, n, parent, step, slot, max_path_len); });
}

int gfal_group_score_poll(gfal_group *g)
{
    if (!g) return GFAL_E_ARG;
    if (!g->pending) return 1;
    for (gfal_scorer *s : g->shards) {
        (void)hipSetDevice(s->device);
        const hipError_t e = hipStreamQuery(s->stream);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return 1;                   // (gfal_group_score_end reports it)
        }
    }
    return 1;
}

int gfal_group_score_end(gfal_group *g, uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    if (!g || !bad || !good) return GFAL_E_ARG;
    return no_throw([&] { return group_end_impl(g, bad, good, unaligned); });
}

int gfal_group_score(gfal_group *g, const int32_t *path_off, const int32_t *path_steps, int32_t n_paths,
                     int filter, uint32_t *bad, uint32_t *good, uint32_t *unaligned)
{
    if (!g || n_paths < 0) return GFAL_E_ARG;
    if (n_paths == 0) return GFAL_OK;
    const int rc = gfal_group_score_begin(g, path_off, path_steps, n_paths, filter);
    if (rc) return rc;
    return gfal_group_score_end(g, bad, good, unaligned);
}

static int pair_scores_impl(gfal_scorer *s, const int32_t *path_steps, int32_t n,
                            int32_t *fw, int32_t *rc_out);

int gfal_scorer_pair_scores(gfal_scorer *s, const int32_t *path_steps, int32_t n,
                            int32_t *fw, int32_t *rc_out)
{
    return no_throw([&] { return pair_scores_impl(s, path_steps, n, fw, rc_out); });
}

static int pair_scores_impl(gfal_scorer *s, const int32_t *path_steps, int32_t n,
                            int32_t *fw, int32_t *rc_out)
{
    if (!s || !path_steps || !fw || !rc_out) return GFAL_E_ARG;
    if (n < 1 || n > GFAL_MAX_STEPS) return GFAL_E_RANGE;
    for (int32_t i = 0; i < n; ++i) {
        int32_t st = path_steps[i];
        if (st < 0 || ((st & ~GFAL_STEP_OTHER) >> 1) >= s->n_nodes) return GFAL_E_RANGE;
    }
    HIP_TRY(hipSetDevice(s->device));
    if (s->n_aln == 0) return GFAL_OK;
    const ImageLayout L = make_layout(s->n_local, n);
    const size_t img_bytes = (size_t)L.total * sizeof(uint16_t);
    if (img_bytes > (size_t)LDS_BUDGET) return GFAL_E_RANGE;
    int rc;
    if (s->have_last) HIP_TRY(hipStreamSynchronize(s->last_stream));
    if ((rc = dev_reserve(&s->d_images, &s->images_cap, (size_t)L.total))) return rc;
    if ((rc = dev_reserve(&s->d_path_off, &s->path_off_cap, 2))) return rc;
    if ((rc = dev_reserve(&s->d_path_steps, &s->path_steps_cap, (size_t)n))) return rc;
    if ((rc = dev_reserve(&s->d_counts, &s->counts_cap, 3))) return rc;
    int32_t off[2] = {0, n};
    HIP_TRY(hipMemcpy(s->d_path_off, off, sizeof(off), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_path_steps, path_steps, (size_t)n * sizeof(int32_t),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(s->d_status, 0, 8 * sizeof(uint32_t), s->stream));
    const size_t prep_lds = img_bytes + (size_t)L.nm * sizeof(uint16_t) +
                            (size_t)L.v2 * sizeof(uint32_t);
    hipLaunchKernelGGL(k_prep, dim3(1), dim3(PREP_THREADS), prep_lds, s->stream, s->d_path_off,
                       s->d_path_steps, 1, (int64_t)n, (int)n, s->d_node_local,
                       (int)s->n_nodes, s->d_node_hist, (uint32_t)s->n_steps,
                       s->n_empty, 0, L, nullptr, s->d_images, s->d_counts, s->d_status, nullptr, nullptr,
                       PrepChild());
    HIP_TRY(hipGetLastError());

    // device results are indexed like the caller's alignments (all shards)
    const size_t n_in = (size_t)s->n_aln_in;
    int32_t *d_fw = nullptr, *d_rc = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_fw), n_in * sizeof(int32_t)));
    hipError_t e2 = hipMalloc(reinterpret_cast<void **>(&d_rc), n_in * sizeof(int32_t));
    if (e2 != hipSuccess) {
        (void)hipFree(d_fw);
        set_err("hipMalloc failed: %s", hipGetErrorString(e2));
        return GFAL_E_NOMEM;
    }
    // zero-step alignments: both tracebacks are free (score 0)
    unsigned fill_blocks = (unsigned)((n_in + 255) / 256);
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks), dim3(256), 0, s->stream, d_fw,
                       (long long)n_in, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks), dim3(256), 0, s->stream, d_rc,
                       (long long)n_in, 0);
    if (s->n_items > 0) {
        Items items{s->d_item_steps, s->d_item_base, s->d_item_len, s->d_item_pairs,
                    s->d_item_pbase, s->n_items, s->d_item_common, s->d_item_hdr, s->d_item_weight};
        if (dp_rows_fit_lds(s->max_aln_len))
            hipLaunchKernelGGL(k_pairs<true>, dim3(DP_BLOCKS), dim3(DP_THREADS),
                               dp_lds_bytes(s->max_aln_len), s->stream, items,
                               s->d_slot_orig, s->d_images, L, s->d_rows, d_fw, d_rc);
        else
            hipLaunchKernelGGL(k_pairs<false>, dim3(DP_BLOCKS), dim3(DP_THREADS), 0,
                               s->stream, items, s->d_slot_orig, s->d_images, L,
                               s->d_rows, d_fw, d_rc);
    }
    std::vector<int32_t> h_fw(n_in), h_rc(n_in);
    hipError_t e3 = hipGetLastError();
    if (e3 == hipSuccess)
        e3 = hipMemcpyAsync(h_fw.data(), d_fw, n_in * sizeof(int32_t), hipMemcpyDeviceToHost,
                            s->stream);
    if (e3 == hipSuccess)
        e3 = hipMemcpyAsync(h_rc.data(), d_rc, n_in * sizeof(int32_t), hipMemcpyDeviceToHost,
                            s->stream);
    if (e3 == hipSuccess) e3 = hipStreamSynchronize(s->stream);
    (void)hipFree(d_fw);
    (void)hipFree(d_rc);
    if (e3 != hipSuccess) {
        set_err("pair_scores: %s", hipGetErrorString(e3));
        return GFAL_E_HIP;
    }
    // only this shard's alignments are written: the shards of one alignment set
    // fill one pair of arrays between them
    const bool dedup = !s->rep_of.empty();       // a copy takes its representative's scores
    for (size_t k = 0; k < n_in; ++k)
        if (s->owned[k]) {
            const size_t from = dedup ? (size_t)s->rep_of[k] : k;
            fw[k] = h_fw[from];
            rc_out[k] = h_rc[from];
        }
    s->last_stream = s->stream;
    s->have_last = true;
    return gfal_scorer_sync_status(s);
}

int gfal_scorer_get_info(gfal_scorer *s, gfal_info *out)
{
    if (!s || !out) return GFAL_E_ARG;
    memset(out, 0, sizeof(*out));
    out->n_aln = s->n_aln;
    out->n_steps = s->n_steps;
    out->n_nodes = s->n_nodes;
    out->n_local_nodes = s->n_local;
    out->max_aln_len = s->max_aln_len;
    out->tile_paths = s->last_tile;
    out->n_workgroups = s->last_grid;
    out->lds_bytes = s->last_lds;
    out->n_lanes = s->n_lanes;
    out->n_score_calls = s->n_score_calls;
    out->n_device_passes = s->n_device_passes;
    out->n_overflow_reruns = s->n_overflow_reruns;
    out->wl_capacity = s->wl_capacity;
    if (s->have_last) {
        HIP_TRY(hipSetDevice(s->device));
        uint32_t host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (s->status_cached && !s->profiling) {
            memcpy(host, s->cached_status, sizeof(host));
        } else {
            HIP_TRY(hipMemcpyAsync(host, s->d_status, sizeof(host), hipMemcpyDeviceToHost,
                                   s->last_stream));
            HIP_TRY(hipStreamSynchronize(s->last_stream));
        }
        out->dp_pairs = (int64_t)((unsigned long long)host[2] | ((unsigned long long)host[3] << 32));
#ifdef GFAL_STAMPS
        {
            unsigned long long hs[8] = {};
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamp_sum), sizeof(hs));
            const double n = hs[4] ? (double)hs[4] : 1.0;
            fprintf(stderr, "stamps per item-wave (%llu items): masks %.0f  lookup %.0f (extra probe rounds %.2f)  "
                            "triage %.0f (entered by %.3f of the items, %.2f lanes each)  counting %.0f cycles\n",
                    hs[4], hs[0] / n, hs[1] / n, hs[5] / n, hs[2] / n, hs[6] / n, hs[6] ? (double)hs[7] / hs[6] : 0.0,
                    hs[3] / n);
            unsigned long long hw[8] = {};
            (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_stamp_wg), sizeof(hw));
            const double nw = hw[3] ? (double)hw[3] : 1.0;
            fprintf(stderr, "per wave (%llu waves, %.2f passes): item loop %.0f cycles, of the rest %.0f: waiting for the "
                            "workgroup's slowest wave %.0f, staging + node masks %.0f, clear + prefixes %.0f, inserts %.0f\n",
                    hw[3], hw[4] / nw, hw[1] / nw, hw[0] / nw, hw[2] / nw, hw[5] / nw, hw[6] / nw, hw[7] / nw);
            unsigned long long zero[8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), zero, sizeof(zero));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_wg), zero, sizeof(zero));
        }
#endif
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 4
        fprintf(stderr, "chain-loop iterations %u, with a wave-uniform entry %u\n", host[4], host[5]);
#endif
#if defined(GFAL_ABLATE) && GFAL_ABLATE == 7
        fprintf(stderr, "(item, tile) visits %u, rejected through the common node %u\n", host[5], host[4]);
#endif
#if defined(GFAL_ABLATE3) && GFAL_ABLATE3 == 7
        fprintf(stderr, "k_scan3: item visits %u, with an open pair %u, taking the overhang test %u (%u lanes)\n", host[4],
                host[7], host[5], host[6]);
#endif
        if (s->profiling && s->ev_calls > 0) {
            const int n = std::min(s->ev_calls, (int)gfal_scorer::EV_RING);
            double scan = 0, dp = 0, total = 0, scan3 = 0;
            for (int i = 0; i < n; ++i) {
                hipEvent_t *ev = s->ev[i];
                float t = 0.f;
                if (s->ev_scan3) {
                    HIP_TRY(hipEventSynchronize(ev[5]));
                    HIP_TRY(hipEventElapsedTime(&t, ev[4], ev[5]));
                    scan3 += t;
                }
                if (!s->prof_all) continue;
                HIP_TRY(hipEventSynchronize(ev[3]));
                HIP_TRY(hipEventElapsedTime(&t, ev[1], ev[2]));
                scan += t;
                HIP_TRY(hipEventElapsedTime(&t, ev[2], ev[3]));
                dp += t;
                HIP_TRY(hipEventElapsedTime(&t, ev[0], ev[3]));
                total += t;
            }
            out->scan_ms = (float)(scan / n);
            out->scan_kernel_ms = (float)(scan3 / n);
            out->dp_ms = (float)(dp / n);
            out->total_ms = (float)(total / n);
            out->profiled_calls = n;
        }
    }
    return GFAL_OK;
}

}  // extern "C"
