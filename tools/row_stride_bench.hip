// Dev tool: gather rate of 128-byte rows that sit `stride` bytes apart at residue `res` (what a (batch, head) plane of
// the value pyramid [B, I, H, D] is: stride = H * D * 4, res = h * D * 4), 8 lanes x 16 B per row, 16 loads in flight per
// lane.  Found in round 5: at stride 1024 the rows at ONE residue (address % 1024 == 384) gather ~20 % slower than the
// other seven, whatever tensor they belong to.   hipcc -O3 --offload-arch=gfx950 tools/row_stride_bench.hip -o tools/bin/row_stride_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const char *table, int rows, uint32_t stride, uint32_t res, int iters, float *sink)
{
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(table), 0, (int)(rows * stride), 0x00020000);
    const int lane = threadIdx.x & 63, j = lane & 7, unit = (threadIdx.x >> 3);
    uint32_t s = (blockIdx.x * 32u + unit) * 2654435761u + 12345u;
    u4 acc = {0, 0, 0, 0};
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
}

int main(int argc, char **argv)
{
    char *table;
    float *sink;
    const size_t bytes = (size_t)64 << 20;
    CHECK(hipMalloc(&table, bytes + 4096));
    CHECK(hipMemset(table, 1, bytes + 4096));
    CHECK(hipMalloc(&sink, 4));
    const int shift = argc > 1 ? atoi(argv[1]) : 0;  // move the table off its allocation's start (bytes, multiple of 16)
    printf("table %% 4096 = %zu\n", (size_t)(uintptr_t)(table + shift) % 4096);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int iters = 256;
    for (uint32_t stride : {1024u, 512u, 2048u, 256u, 128u}) {
        const int rows = 5440;
        printf("stride %4u rows %d:", stride, rows);
        for (uint32_t res = 0; res < stride && res < 2048; res += 128) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(k, dim3(256 * 5), dim3(256), 0, 0, table + shift, rows, stride, res, iters, sink);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            const double gathered = 256.0 * 5 * 32 * (double)iters * 16 * 128;
            printf(" %5.1f", gathered / best / 1e9);
        }
        printf("  TB/s by residue\n");
    }
    return 0;
}
