// dev tool: does it matter that the finish pass (and the forward's `out`) writes 128-byte rows 1 KB apart, plane by plane,
// instead of contiguous memory?  Writes N rows of 128 bytes (a) contiguously, (b) as [pixel][H=8][32 floats] with a
// workgroup per (plane, 64-pixel tile), planes spread over the XCDs like the library's launches, (c) the same with a
// workgroup per (64-pixel tile, all 8 heads).   hipcc -O3 --offload-arch=gfx950 tools/stride_write_bench.hip -o /tmp/swb && /tmp/swb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void w_contig(v4f *dst, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, dst + i);
}
// grid (8, tiles, planes / 8): plane = z * 8 + x (same XCD for a plane's tiles), tile = y; 64 pixels per workgroup
__global__ __launch_bounds__(256) void w_plane(v4f *dst, int I, int H)
{
    const int plane = blockIdx.z * 8 + blockIdx.x, tile = blockIdx.y;
    const int b = plane / H, h = plane % H;
    const int j = threadIdx.x & 7, unit = threadIdx.x >> 3;
    for (int r = 0; r < 2; ++r) {
        const int pix = tile * 64 + r * 32 + unit;
        if (pix < I) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, dst + (((long long)b * I + pix) * H + h) * 8 + j);
    }
}
// grid (tiles, B): a workgroup writes 64 pixels x all heads = 64 KB contiguous (H = 8)
__global__ __launch_bounds__(256) void w_allheads(v4f *dst, int I, int H)
{
    const int b = blockIdx.y, tile = blockIdx.x;
    for (int r = 0; r < 64 * H * 8 / 256; ++r) {
        const int e = r * 256 + threadIdx.x;  // float4 index inside the tile
        const int pix = tile * 64 + e / (H * 8);
        if (pix < I) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, dst + ((long long)b * I + tile * 64) * H * 8 + e);
    }
}

int main()
{
    const int cfg[3][3] = {{8, 17821, 8}, {4, 5440, 8}, {2, 17821, 8}};  // (B, I, H): dec_coco, c2, c3 (fp32 rows here)
    for (auto &c : cfg) {
        const int B = c[0], I = c[1], H = c[2];
        const long long n4 = (long long)B * I * H * 8;
        v4f *d;
        hipMalloc(&d, n4 * 16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int tiles = (I + 63) / 64;
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) w_contig<<<dim3((unsigned)((n4 + 255) / 256)), 256>>>(d, n4);
                if (mode == 1) w_plane<<<dim3(8, tiles, B * H / 8), 256>>>(d, I, H);
                if (mode == 2) w_allheads<<<dim3(tiles, B), 256>>>(d, I, H);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 3 && ms < best) best = ms;
            }
            printf("B=%d I=%d H=%d  %-28s %7.1f us  %6.2f TB/s\n", B, I, H,
                   mode == 0 ? "contiguous" : mode == 1 ? "plane-major rows, 1 KB apart" : "tile x all heads (64 KB)", best * 1e3,
                   n4 * 16 / (best * 1e-3) / 1e12);
        }
        hipFree(d);
    }
    return 0;
}
