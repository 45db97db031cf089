// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access shapes
// k_scan uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a wide
// coalesced stream on gfx950; other widths are uncalibrated).  Reads a buffer
// far larger than the 256 MiB Infinity Cache once per kernel:
//   read_u16   2 bytes per lane, 128 B per wave-instruction (the item loads)
//   read_b128  16 bytes per lane (the image staging loads)
// Run under `rocprofv3 --pmc FETCH_SIZE` and compare with the bytes printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read_u16(const uint16_t *p, size_t n, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

__global__ void read_b128(const uint4 *p, size_t n, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;   // 1 GiB
    void *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_u16, dim3(4096), dim3(256), 0, 0, (const uint16_t *)buf, bytes / 2, sink);
        hipLaunchKernelGGL(read_b128, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
    }
    hipDeviceSynchronize();
    printf("each kernel reads %zu bytes\n", bytes);
    return 0;
}
