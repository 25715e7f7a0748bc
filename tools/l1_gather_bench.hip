// Dev tool: how fast can a CU gather 128-byte rows with 16-byte-per-lane buffer loads (8 lanes per row, 8 rows per
// wave instruction, 16 loads in flight per lane) from a table that sits in L1 / L2?  Sets the ceiling of the MSDA
// forward's gather phase.   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_bench.hip -o tools/bin/l1_gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT> __global__ __launch_bounds__(256) void k(const float *table, int rows, int iters, float *sink, int waves_note)
{
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, rows * 128, 0x00020000);
    const int lane = threadIdx.x & 63, j = lane & 7, unit = (threadIdx.x >> 3);
    uint32_t s = (blockIdx.x * 32u + unit) * 2654435761u + 12345u;
    u4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u4 v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t row = (s >> 8) % (uint32_t)rows;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, row * 128u + j * 16u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int INFLIGHT> int run(const float *table, int rows, int wgs_per_cu, float *sink)
{
    const int iters = 4096 / INFLIGHT;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, table, rows, iters, sink, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, table, rows, iters, sink, 0);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 256.0 * wgs_per_cu * 32 * (double)iters * INFLIGHT * 128;
    printf("rows=%6d (%7.1f KiB) inflight=%2d wg/cu=%d: %7.3f ms  %6.1f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", rows, rows * 128 / 1024.0,
           INFLIGHT, wgs_per_cu, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    return 0;
}

int main()
{
    float *table, *sink;
    const int maxrows = 1 << 20;
    CHECK(hipMalloc(&table, (size_t)maxrows * 128));
    CHECK(hipMemset(table, 1, (size_t)maxrows * 128));
    CHECK(hipMalloc(&sink, 4));
    for (int rows : {64, 320, 1344, 5440, 43520, 348160}) {
        for (int wg : {2, 4, 5}) {
            run<16>(table, rows, wg, sink);
        }
        run<8>(table, rows, 5, sink);
        run<32>(table, rows, 4, sink);
    }
    return 0;
}
