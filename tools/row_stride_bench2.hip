// Dev tool (round 5): the slow residue of tools/row_stride_bench.hip in the LDSL kernels' geometry — ONE 1024-thread
// workgroup per CU, eight workgroups per plane, the four planes of an XCD `plane_bytes` apart, every wave 16 loads in
// flight — per residue (address % 1024) of the gathered 128-byte rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int BLK> __global__ __launch_bounds__(BLK) void k(const char *table, int rows, uint32_t stride, uint32_t res, size_t plane_bytes, int wgs_per_plane,
                                         int iters, float *sink, unsigned long long *cyc)
{
    const int wg = blockIdx.x, x = wg & 7, t = wg >> 3, plane_in_xcd = t / wgs_per_plane;
    const char *plane = table + (size_t)plane_in_xcd * plane_bytes + (size_t)x * 0;  // (same head on every XCD, as xcd_map does)
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(plane), 0, (int)(rows * stride), 0x00020000);
    const int lane = threadIdx.x & 63, j = lane & 7, unit = (threadIdx.x >> 3);
    uint32_t s = (blockIdx.x * 128u + unit) * 2654435761u + 12345u;
    u4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t row = (s >> 8) % (uint32_t)rows;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, row * stride + res + j * 16u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
    if (threadIdx.x == 0) cyc[wg] = __builtin_amdgcn_s_memtime() - t0;
}

int main(int argc, char **argv)
{
    char *table;
    float *sink;
    unsigned long long *cyc, h[4096];
    const size_t plane_bytes = argc > 1 ? strtoull(argv[1], 0, 0) : (size_t)5440 * 4096;
    CHECK(hipMalloc(&table, plane_bytes * 4 + (8 << 20)));
    CHECK(hipMemset(table, 1, plane_bytes * 4 + (8 << 20)));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&cyc, sizeof(h)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int iters = 64, rows = 5120;
    printf("plane_bytes %zu (low 20 bits 0x%zx)\n", plane_bytes, plane_bytes & 0xfffff);
    for (int blk : {1024, 256}) {
        const int wgs = blk == 1024 ? 256 : 256 * 5, wpp = wgs / 8 / 4;
        printf("%4d-thread workgroups, %d per plane; ms by residue:", blk, wpp);
        for (uint32_t res = 0; res < 1024; res += 128) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(a));
                if (blk == 1024) hipLaunchKernelGGL(k<1024>, dim3(wgs), dim3(1024), 0, 0, table, rows, 1024u, res, plane_bytes, wpp, iters, sink, cyc);
                else hipLaunchKernelGGL(k<256>, dim3(wgs), dim3(256), 0, 0, table, rows, 1024u, res, plane_bytes, wpp, iters * 4 / 5, sink, cyc);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf(" %.4f", best);
        }
        printf("\n");
    }
    return 0;
}
