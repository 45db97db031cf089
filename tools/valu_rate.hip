// valu_rate.hip -- micro-benchmark: sustained ISSUE rate of the instructions
// k_scan is made of, on gfx950, at 1..8 waves per SIMD.
//
// Why: k_scan moves 0.5 % of its algorithmic bytes through HBM, so its
// roofline is instruction issue, not bandwidth.  /opt/skills/guides/
// MI355X_MICROARCH.md gives 2 cycles per wave64 VALU instruction per SIMD
// (4 for a lone wave) for v_fma_f32; this tool measures the same for the
// integer / mask instructions the kernel actually issues (v_alignbit_b32,
// v_cmp -> SGPR, v_cndmask, v_and/or, v_readlane, v_bfe), for the scalar unit
// (one per CU, shared by its 4 SIMDs), and for mixes of the two.
//
// Method: every wave runs ITERS iterations of an unrolled block of UNROLL
// instructions on 8 independent register chains (inline asm, so the stream is
// exactly what is written), stamps s_memtime before and after, and stores the
// difference.  Exactly W blocks of 256 threads (1 wave per SIMD each) are made
// resident per CU by giving each block 160 KiB / W of LDS; grid = CUs x W.
// Output (JSON lines): per instruction kind and W, the sustained issue rate in
// G wave-instructions per second per SIMD (VALU) or per CU (SALU, LDS), over the
// span of the launch, and the same in cycles at the clock the waves measured.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                               \
        }                                                                           \
    } while (0)

constexpr int ITERS = 20000;  // x UNROLL blocks per wave (launch ramp / tail << span)
constexpr int UNROLL = 4;

// eight independent chains: a..h; s = shift / second operand
#define V8(OP)                                                                      \
    OP(a) OP(b) OP(c) OP(d) OP(e) OP(f) OP(g) OP(h)

enum Kind {
    K_AND = 0,       // v_and_b32                       (VOP2)
    K_ADD,           // v_add_u32                       (VOP2)
    K_ALIGNBIT,      // v_alignbit_b32                  (VOP3, 3 sources)
    K_BFE,           // v_bfe_u32                       (VOP3)
    K_CNDMASK_VCC,   // v_cndmask_b32 (vcc)             (VOP2)
    K_CNDMASK_SGPR,  // v_cndmask_b32_e64 (sgpr pair)   (VOP3)
    K_CMP_CND,       // v_cmp -> vcc ; v_cndmask vcc pairs
    K_CMP_VCC,       // v_cmp_eq_u32 -> vcc             (VOPC)
    K_CMP_SGPR,      // v_cmp_eq_u32_e64 -> s[n:n+1]    (VOP3)
    K_READLANE,      // v_readlane_b32 -> sgpr
    K_SALU,          // s_and_b64 / s_bcnt1_i32_b64 / s_add_u32 mix
    K_MIX_VS,        // 1 VALU (alignbit) : 1 SALU (s_and_b64), alternating
    K_MIX_SCAN,      // the k_scan chain-walk mix: 4 VALU : 2 SALU : (1 ds_read_b32 per 8)
    K_DSREAD,        // ds_read_b32, 8 in flight, then wait
    N_KINDS
};

const char *kind_name[N_KINDS] = {"v_and_b32", "v_add_u32", "v_alignbit_b32", "v_bfe_u32",
                                  "v_cndmask_b32(vcc)", "v_cndmask_b32_e64(sgpr)", "v_cmp+v_cndmask pairs", "v_cmp_eq_u32->vcc", "v_cmp_eq_u32->sgpr",
                                  "v_readlane_b32", "salu(and_b64,bcnt1,add)", "mix 1 valu:1 salu",
                                  "mix scan(4v:2s:ds/8)", "ds_read_b32 x8"};
// instructions per unrolled block, [valu, salu, lds]
const int per_block[N_KINDS][3] = {{8, 0, 0}, {8, 0, 0}, {8, 0, 0}, {8, 0, 0}, {8, 0, 0}, {8, 0, 0}, {8, 0, 0},
                                   {8, 0, 0},                                    {8, 0, 0}, {8, 0, 0}, {0, 8, 0}, {8, 8, 0}, {8, 4, 1}, {0, 0, 8}};

template <int KIND>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long *out, int iters)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3, e = a * 11 + 4,
             f = a * 13 + 5, g = a * 17 + 6, h = a * 19 + 7;
    const uint32_t s = 16u + (threadIdx.x & 1u) * 0u;
    uint32_t addr = (threadIdx.x & 63u) * 4u;
    unsigned long long t0, t1, w0, w1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // one asm statement per iteration (UNROLL x 8 instructions): hipcc puts an
    // s_nop between asm statements it cannot see into
#define R4(x) x x x x
#define CHAINS "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
    unsigned long long m0 = t0, m1 = t0 + 1, m2 = t0 + 2, m3 = t0 + 3;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_AND) {
            asm volatile(R4("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n"
                            "v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n"
                            "v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                         : CHAINS : "v"(s));
        } else if (KIND == K_ADD) {
            asm volatile(R4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n"
                            "v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n"
                            "v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : CHAINS : "v"(s));
        } else if (KIND == K_ALIGNBIT) {
            asm volatile(R4("v_alignbit_b32 %0, %0, %1, %8\n v_alignbit_b32 %1, %1, %2, %8\n"
                            "v_alignbit_b32 %2, %2, %3, %8\n v_alignbit_b32 %3, %3, %4, %8\n"
                            "v_alignbit_b32 %4, %4, %5, %8\n v_alignbit_b32 %5, %5, %6, %8\n"
                            "v_alignbit_b32 %6, %6, %7, %8\n v_alignbit_b32 %7, %7, %0, %8\n")
                         : CHAINS : "v"(s));
        } else if (KIND == K_BFE) {
            asm volatile(R4("v_bfe_u32 %0, %0, %8, 11\n v_bfe_u32 %1, %1, %8, 11\n v_bfe_u32 %2, %2, %8, 11\n"
                            "v_bfe_u32 %3, %3, %8, 11\n v_bfe_u32 %4, %4, %8, 11\n v_bfe_u32 %5, %5, %8, 11\n"
                            "v_bfe_u32 %6, %6, %8, 11\n v_bfe_u32 %7, %7, %8, 11\n")
                         : CHAINS : "v"(s));
        } else if (KIND == K_CNDMASK_VCC) {
            asm volatile(R4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
                            "v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                            "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : CHAINS : "v"(s) : "vcc");
        } else if (KIND == K_CNDMASK_SGPR) {
            asm volatile(R4("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %10\n"
                            "v_cndmask_b32_e64 %2, %2, %8, %11\n v_cndmask_b32_e64 %3, %3, %8, %12\n"
                            "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %10\n"
                            "v_cndmask_b32_e64 %6, %6, %8, %11\n v_cndmask_b32_e64 %7, %7, %8, %12\n")
                         : CHAINS : "v"(s), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
        } else if (KIND == K_CMP_CND) {
            asm volatile(R4("v_cmp_eq_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                            "v_cmp_eq_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cmp_eq_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n"
                            "v_cmp_eq_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : CHAINS : "v"(s) : "vcc");
        } else if (KIND == K_CMP_VCC) {
            asm volatile(R4("v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n"
                            "v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n"
                            "v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8\n")
                         : CHAINS : "v"(s) : "vcc");
        } else if (KIND == K_CMP_SGPR) {
            asm volatile(R4("v_cmp_eq_u32_e64 %8, %0, %12\n v_cmp_eq_u32_e64 %9, %1, %12\n"
                            "v_cmp_eq_u32_e64 %10, %2, %12\n v_cmp_eq_u32_e64 %11, %3, %12\n"
                            "v_cmp_eq_u32_e64 %8, %4, %12\n v_cmp_eq_u32_e64 %9, %5, %12\n"
                            "v_cmp_eq_u32_e64 %10, %6, %12\n v_cmp_eq_u32_e64 %11, %7, %12\n")
                         : CHAINS, "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(s));
        } else if (KIND == K_READLANE) {
            asm volatile(R4("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %9, %1, 5\n"
                            "v_readlane_b32 %10, %2, 7\n v_readlane_b32 %11, %3, 9\n"
                            "v_readlane_b32 %8, %4, 3\n v_readlane_b32 %9, %5, 5\n"
                            "v_readlane_b32 %10, %6, 7\n v_readlane_b32 %11, %7, 9\n")
                         : CHAINS, "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));
        } else if (KIND == K_SALU) {
            asm volatile(R4("s_and_b64 %0, %0, %1\n s_bcnt1_i32_b64 %4, %2\n s_or_b64 %1, %1, %3\n"
                            "s_add_u32 %5, %6, 7\n s_andn2_b64 %2, %2, %3\n s_bcnt1_i32_b64 %6, %0\n"
                            "s_xor_b64 %3, %3, %1\n s_lshl_b32 %7, %4, 2\n")
                         : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3)
                         : : "scc");
        } else if (KIND == K_MIX_VS) {
            asm volatile(R4("v_alignbit_b32 %0, %0, %1, %12\n s_and_b64 %8, %8, %9\n"
                            "v_alignbit_b32 %1, %1, %2, %12\n s_or_b64 %9, %9, %10\n"
                            "v_alignbit_b32 %2, %2, %3, %12\n s_andn2_b64 %10, %10, %11\n"
                            "v_alignbit_b32 %3, %3, %4, %12\n s_xor_b64 %11, %11, %8\n"
                            "v_alignbit_b32 %4, %4, %5, %12\n s_and_b64 %8, %8, %9\n"
                            "v_alignbit_b32 %5, %5, %6, %12\n s_or_b64 %9, %9, %10\n"
                            "v_alignbit_b32 %6, %6, %7, %12\n s_andn2_b64 %10, %10, %11\n"
                            "v_alignbit_b32 %7, %7, %0, %12\n s_xor_b64 %11, %11, %8\n")
                         : CHAINS, "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(s) : "scc");
        } else if (KIND == K_MIX_SCAN) {
            // the shape of one chain-walk iteration of k_scan: an LDS read, then 8
            // VALU and 4 SALU that do not depend on it, then the wait
            uint32_t got = 0;
            asm volatile(R4("ds_read_b32 %12, %14\n"
                            "v_alignbit_b32 %0, %0, %1, %13\n v_and_b32 %2, %2, %13\n"
                            "s_and_b64 %8, %8, %9\n"
                            "v_cmp_eq_u32 vcc, %3, %13\n v_cndmask_b32 %4, %4, %13, vcc\n"
                            "s_bcnt1_i32_b64 %15, %8\n"
                            "v_add_u32 %5, %5, %13\n v_bfe_u32 %6, %6, %13, 11\n"
                            "s_or_b64 %9, %9, %10\n"
                            "v_lshlrev_b32 %7, 2, %7\n v_and_b32 %1, %1, %13\n"
                            "s_andn2_b64 %10, %10, %11\n"
                            "s_waitcnt lgkmcnt(0)\n")
                         : CHAINS, "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+v"(got)
                         : "v"(s), "v"(addr), "s"(r0) : "scc", "vcc", "memory");
        } else if (KIND == K_DSREAD) {
            uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile(R4("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n"
                            "ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                            "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n"
                            "ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                            "s_waitcnt lgkmcnt(0)\n")
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                         : "v"(addr) : "memory");
        }
    }
    asm volatile("" ::"s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(r0), "s"(r1), "s"(r2), "s"(r3));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1)::"memory");
    // keep everything live
    uint32_t sink = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (sink == 0x12345679u) out[0] = sink;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *rec = out + 1 + 3 * (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        rec[0] = t1 - t0;     // shader cycles
        rec[1] = w0;          // 100 MHz wall clock
        rec[2] = w1;
    }
}

// W waves per SIMD: one block of 256 x W threads per CU (W <= 4), or two blocks
// of 128 x W threads (W = 6, 8: the geometry k_scan runs in); the LDS request
// (all of the CU's 160 KiB, or half) keeps any further block away.
struct Result {
    double cycles;        // median shader cycles of a wave (s_memtime)
    double dur_ns;        // median wall time of a wave (s_memrealtime, 100 MHz)
    double span_ns;       // first start .. last end over all waves
    double concurrency;   // sum of wave durations / (span x SIMDs): waves per SIMD really co-resident
};

template <int KIND>
int run(int n_cus, int waves_per_simd, Result *res)
{
    const int blocks_per_cu = waves_per_simd > 4 ? 2 : 1;
    const int threads = 256 * waves_per_simd / blocks_per_cu;
    const int grid = n_cus * blocks_per_cu;
    const size_t lds = (size_t)(160 * 1024 / blocks_per_cu);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rate<KIND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned long long *d_out;
    const size_t n_waves = (size_t)grid * (threads / 64);
    const size_t n_out = 1 + 3 * n_waves;
    CHECK(hipMalloc((void **)&d_out, n_out * sizeof(unsigned long long)));
    CHECK(hipMemset(d_out, 0, n_out * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(threads), lds, 0, d_out, 50);   // warm-up
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(threads), lds, 0, d_out, ITERS);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(n_out);
    CHECK(hipMemcpy(h.data(), d_out, n_out * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> cyc(n_waves), dur(n_waves);
    unsigned long long first = ~0ull, last = 0;
    double sum = 0;
    for (size_t w = 0; w < n_waves; ++w) {
        const unsigned long long *rec = h.data() + 1 + 3 * w;
        cyc[w] = (double)rec[0];
        dur[w] = (double)(rec[2] - rec[1]) * 10.0;
        sum += dur[w];
        first = std::min(first, rec[1]);
        last = std::max(last, rec[2]);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(dur.begin(), dur.end());
    res->cycles = cyc[n_waves / 2];
    res->dur_ns = dur[n_waves / 2];
    res->span_ns = (double)(last - first) * 10.0;
    res->concurrency = sum / (res->span_ns * n_cus * 4);
    CHECK(hipFree(d_out));
    return 0;
}

template <int KIND>
int report(int n_cus)
{
    for (int w : {1, 2, 3, 4, 6, 8}) {
        Result r;
        if (run<KIND>(n_cus, w, &r)) return 1;
        const double nv = (double)per_block[KIND][0] * ITERS * UNROLL, ns = (double)per_block[KIND][1] * ITERS * UNROLL,
                     nl = (double)per_block[KIND][2] * ITERS * UNROLL;
        // Rates over the SPAN of the launch (first wave start .. last wave end, 100 MHz
        // wall clock): all instructions of a SIMD's W waves divided by the time the
        // SIMD needed for them.  (Per-wave durations mislead: the arbiter favours the
        // oldest waves, which then finish early and leave the rest of the span to the
        // others.)  G wave-instructions per second; cycles at the measured clock.
        const double clock = r.cycles / r.dur_ns;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"resident_waves_per_simd\": %.2f, "
               "\"span_us\": %.1f, \"clock_ghz\": %.3f", kind_name[KIND], w, r.concurrency, r.span_ns / 1e3, clock);
        if (nv > 0) printf(", \"valu_ginst_per_s_simd\": %.4f, \"cyc_per_valu_simd\": %.3f", nv * w / r.span_ns, r.span_ns * clock / (nv * w));
        if (ns > 0) printf(", \"salu_ginst_per_s_cu\": %.4f, \"cyc_per_salu_cu\": %.3f", ns * w * 4 / r.span_ns, r.span_ns * clock / (ns * w * 4));
        if (nl > 0) printf(", \"lds_ginst_per_s_cu\": %.4f, \"cyc_per_lds_cu\": %.3f", nl * w * 4 / r.span_ns, r.span_ns * clock / (nl * w * 4));
        printf("}\n");
    }
    return 0;
}

int main()
{
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    fprintf(stderr, "CUs: %d, %d iterations per wave\n", n_cus, ITERS);
    if (report<K_AND>(n_cus)) return 1;
    if (report<K_ADD>(n_cus)) return 1;
    if (report<K_ALIGNBIT>(n_cus)) return 1;
    if (report<K_BFE>(n_cus)) return 1;
    if (report<K_CNDMASK_VCC>(n_cus)) return 1;
    if (report<K_CNDMASK_SGPR>(n_cus)) return 1;
    if (report<K_CMP_CND>(n_cus)) return 1;
    if (report<K_CMP_VCC>(n_cus)) return 1;
    if (report<K_CMP_SGPR>(n_cus)) return 1;
    if (report<K_READLANE>(n_cus)) return 1;
    if (report<K_SALU>(n_cus)) return 1;
    if (report<K_MIX_VS>(n_cus)) return 1;
    if (report<K_MIX_SCAN>(n_cus)) return 1;
    if (report<K_DSREAD>(n_cus)) return 1;
    return 0;
}
