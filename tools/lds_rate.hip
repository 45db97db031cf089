// lds_rate.hip -- micro-benchmark: LDS read throughput per wave-instruction for
// the access shapes k_scan uses (gfx950).  Prints cycles per wave-instruction
// per CU, measured with all CUs busy, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters, int spread)
{
    __shared__ uint32_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // `spread` distinct addresses per wave (1 = broadcast, 64 = all different)
    uint32_t idx = ((lane % spread) * 37u + (threadIdx.x >> 6) * 101u) & 8191u;
    uint32_t acc = 0;
    const uint16_t *l16 = reinterpret_cast<const uint16_t *>(lds);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 4 x ds_read_u16, consecutive entries
            acc += l16[idx] + l16[idx + 1] + l16[idx + 2] + l16[idx + 3];
        } else if (MODE == 1) {   // 4 x ds_read_b32
            acc += lds[idx] + lds[idx + 1] + lds[idx + 2] + lds[idx + 3];
        } else {                  // 2 x ds_read_b64 (aligned)
            const uint2 *l64 = reinterpret_cast<const uint2 *>(lds);
            uint2 a = l64[idx >> 1], b = l64[(idx >> 1) + 1];
            acc += a.x + a.y + b.x + b.y;
        }
        idx = (idx + (acc & 1u) * 2u + 6u) & 8191u;   // dependent, stays even-ish
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
float run(int iters, int spread)
{
    unsigned *out;
    hipMalloc((void **)&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(1024), 0, 0, out, 10, spread);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(1024), 0, 0, out, iters, spread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    return ms;
}

int main()
{
    const int iters = 20000;
    const char *names[] = {"4x ds_read_u16", "4x ds_read_b32", "2x ds_read_b64"};
    for (int spread : {1, 8, 64}) {
        float t[3] = {run<0>(iters, spread), run<1>(iters, spread), run<2>(iters, spread)};
        for (int m = 0; m < 3; ++m) {
            // 512 blocks x 16 waves over 256 CUs = 32 waves per CU; each wave does iters x (4 or 2) reads
            double insts_per_cu = 32.0 * iters * (m == 2 ? 2 : 4);
            double cycles = t[m] * 1e-3 * 2.4e9;
            printf("spread %2d  %-16s %8.2f ms  %.2f cycles per wave-instruction per CU\n", spread,
                   names[m], t[m], cycles / insts_per_cu);
        }
    }
    return 0;
}
