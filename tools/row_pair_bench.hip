// Dev tool: would an x-pair-contiguous layout make 16-bit value rows fill their cache lines?
// The bilinear footprint of a sample is 2 x 2 pixels.  With 64-byte rows (D = 32, bf16):
//   A  [pixel][head][D] (the operator's layout): the four corner rows are four 64-byte pieces in four different 128-byte
//      lines; a unit = 4 lanes x 16 B, 16 units per wave, 4 load instructions per sample;
//   B  [head][pixel][D]: (x0, y) and (x0 + 1, y) are adjacent, so a unit = 8 lanes x 16 B fetches both x-corners of a
//      row with ONE instruction (128 contiguous bytes: one line when the pair is 128-byte aligned, two otherwise),
//      8 units per wave, 2 load instructions per sample.
// Both move 256 bytes per sample with 0.25 wave-level load instructions per sample; what differs is the lines touched
// per instruction: 16 half-used lines (A) against 8..16, ~12 on average, fully used (B).
// The benchmark gathers random footprints of one level (w x h pixels) of one plane, nothing else, 16 loads in flight per
// lane, and prints picoseconds per sample for both layouts.
//   hipcc -O3 --offload-arch=gfx950 tools/row_pair_bench.hip -o tools/bin/row_pair_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// LAYOUT 0: A (pixel stride = heads * 64 bytes), 1: B (pixel stride 64 bytes, pairs), 2: B with every pair forced onto
// an even x0 (always one line: the bound of what alignment could give)
template <int LAYOUT> __global__ __launch_bounds__(256) void k(const char *table, int w, int h, int heads, int iters, float *sink)
{
    const uint32_t pix_stride = LAYOUT == 0 ? (uint32_t)heads * 64u : 64u;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(table), 0, (int)((uint32_t)w * h * pix_stride + 256), 0x00020000);
    constexpr int G = LAYOUT == 0 ? 4 : 8;
    const int lane = threadIdx.x & 63, j = lane % G, unit = threadIdx.x / G;
    uint32_t s = (blockIdx.x * 64u + unit) * 2654435761u + 12345u;
    u4 acc = {0, 0, 0, 0};
    constexpr int SAMPLES = LAYOUT == 0 ? 4 : 8;  // 16 loads in flight per lane either way
    for (int it = 0; it < iters; ++it) {
        u4 v[16];
#pragma unroll
        for (int u = 0; u < SAMPLES; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t x0 = (s >> 8) % (uint32_t)(w - 1), y0 = (s >> 20) % (uint32_t)(h - 1);
            if (LAYOUT == 2) x0 &= ~1u;
            const uint32_t p00 = (y0 * (uint32_t)w + x0) * pix_stride, prow = (uint32_t)w * pix_stride;
            if (LAYOUT == 0) {
                v[4 * u + 0] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + j * 16u, 0, 0);
                v[4 * u + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + pix_stride + j * 16u, 0, 0);
                v[4 * u + 2] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + prow + j * 16u, 0, 0);
                v[4 * u + 3] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + prow + pix_stride + j * 16u, 0, 0);
            } else {
                v[2 * u + 0] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + j * 16u, 0, 0);
                v[2 * u + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, p00 + prow + j * 16u, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int LAYOUT> int run(const char *table, int w, int h, int heads, int wgs_per_cu, float *sink, const char *name)
{
    constexpr int SAMPLES = LAYOUT == 0 ? 4 : 8, UNITS = LAYOUT == 0 ? 64 : 32;
    const int iters = 512;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<LAYOUT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, table, w, h, heads, iters, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<LAYOUT>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, table, w, h, heads, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double samples = 256.0 * wgs_per_cu * UNITS * (double)iters * SAMPLES;
    printf("  %-34s wg/cu=%d: %7.3f ms  %6.2f ps/sample  %5.1f TB/s of row bytes\n", name, wgs_per_cu, ms, ms * 1e9 / samples,
           samples * 256 / ms / 1e9);
    return 0;
}

int main()
{
    char *table;
    float *sink;
    const size_t bytes = (size_t)256 << 20;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 1, bytes));
    CHECK(hipMalloc(&sink, 4));
    const int heads = 8;
    for (auto wh : {std::pair<int, int>{134, 100}, {67, 50}, {17, 13}, {64, 64}}) {
        printf("level %d x %d (plane %.1f KiB of 64-byte rows)\n", wh.first, wh.second, wh.first * wh.second * 64 / 1024.0);
        for (int wg : {4, 5}) {
            run<0>(table, wh.first, wh.second, heads, wg, sink, "A [pixel][head][D], 4 x 64 B");
            run<1>(table, wh.first, wh.second, heads, wg, sink, "B [head][pixel][D], 2 x 128 B");
            run<2>(table, wh.first, wh.second, heads, wg, sink, "B, every pair line-aligned (bound)");
        }
    }
    return 0;
}
