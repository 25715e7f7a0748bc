// dev probe: which lanes of a wave64 ds_read_b128 are served together — i.e. which row-gather patterns are free of LDS
// bank conflicts.  One workgroup per CU, W waves, each wave issues ITERS dependent-free ds_read_b128 with a per-pattern
// address function; cycles per instruction from s_memtime.  Patterns (rows of 128 bytes at pseudo-random row indices):
//   0  lane * 16 (one contiguous 1 KB block: the conflict-free reference)
//   1  8 consecutive lanes read one row (the gather kernels' unit = lanes 8u .. 8u+7)
//   2  unit = lanes {4u .. 4u+3} + {32+4u .. 32+4u+3}
//   3  unit = lanes {2u, 2u+1} + {16+..} + {32+..} + {48+..}   (two lanes of every 16)
//   4  unit = lanes u, u+8, u+16, ... (stride 8)
//   5  as 1, rows rotated by (row % 8) * 16 bytes inside their 128 bytes (lane j reads piece (j + row) % 8)
//   hipcc -O3 --offload-arch=gfx950 tools/lds_b128_bench.hip -o /tmp/lds_b128_bench && /tmp/lds_b128_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 256;

__global__ __launch_bounds__(1024) void k(int pattern, int rows, unsigned long long *cyc, uint32_t *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < rows * 32; i += blockDim.x) ((uint32_t *)smem)[i] = i;
    __syncthreads();
    int unit, j;
    switch (pattern) {
    case 2: unit = (lane & 31) >> 2, j = (lane & 3) | ((lane >> 5) << 2); break;
    case 3: unit = (lane & 15) >> 1, j = (lane & 1) | ((lane >> 4) << 1); break;
    case 4: unit = lane & 7, j = lane >> 3; break;
    default: unit = lane >> 3, j = lane & 7; break;
    }
    uint32_t s = (blockIdx.x * 16u + wave) * 2654435761u + unit * 40503u + 17u;
    u4 acc = {0, 0, 0, 0};
    uint32_t offs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        s = s * 1664525u + 1013904223u;
        const uint32_t row = (s >> 10) % (uint32_t)rows;
        const uint32_t piece = pattern == 5 ? ((uint32_t)j + row) & 7u : (uint32_t)j;
        offs[u] = pattern == 0 ? (uint32_t)lane * 16u + (row & ~7u) * 128u % ((uint32_t)rows * 128u - 1024u) : row * 128u + piece * 16u;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u4 v = *reinterpret_cast<const u4 *>(smem + offs[u]);
            acc ^= v;
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main()
{
    unsigned long long *cyc;
    uint32_t *sink;
    hipMalloc(&cyc, 256 * 16 * 8);
    hipMalloc(&sink, 4);
    const int rows = 320;  // 40 KB: the c2 coarse levels
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int waves : {1, 4, 16}) {
        printf("waves per CU %2d: ", waves);
        for (int pat = 0; pat < 6; ++pat) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), rows * 128, 0, pat, rows, cyc, sink);
                hipDeviceSynchronize();
                static unsigned long long h[256 * 16];
                hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                double sum = 0;
                for (int b = 0; b < 256; ++b)
                    for (int w = 0; w < waves; ++w) sum += (double)h[b * 16 + w];
                const double per = sum / (256.0 * waves) / ITERS;
                if (per < best) best = per;
            }
            printf(" p%d %.1f", pat, best);
        }
        printf("   (s_memtime ticks per ds_read_b128 per wave)\n");
    }
    return 0;
}
