// dev probe: where global_load_lds_{dword,dwordx3} put a lane's data (measured: M0 + lane * 4; dwordx3: M0 + lane * 16), with a
// partial exec mask, issued from inline assembly (the compiler does not track it):  hipcc --offload-arch=gfx950 -O3
// tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ char smem[];
__device__ inline void dma4(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ inline void dma12(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx3 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__global__ void k(const float *src, float *out)
{
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    float *mine = (float *)smem + wave * 1024;
    for (int i = lane; i < 1024; i += 64) mine[i] = -1.0f;
    __syncthreads();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char *)(smem) + wave * 4096);
    if (lane % 3 != 1) dma4(src + 1000 + lane, base);            // floats [0, 64)
    if (lane < 40) dma12(src + 3 * lane, base + 1024);            // floats [256, 256 + 256): 16 bytes apart
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[wave * 1024 + i] = mine[i];
}
int main()
{
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 8192, 0, d, o);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w) {
        for (int l = 0; l < 64; ++l) {
            const float want = (l % 3 != 1) ? 1000.0f + l : -1.0f;
            if (r[w * 1024 + l] != want) ++bad;
        }
        for (int l = 0; l < 64; ++l)
            for (int c = 0; c < 3; ++c) {
                const float want = l < 40 ? 3.0f * l + c : -1.0f;
                if (r[w * 1024 + 256 + 4 * l + c] != want) ++bad;
            }
    }
    printf("lds dma probe: %d mismatches; wave 1 dword lanes 0..5: %g %g %g %g %g %g; x3 lane 2: %g %g %g\n", bad, r[1024], r[1025], r[1026],
           r[1027], r[1028], r[1029], r[1024 + 256 + 8], r[1024 + 256 + 9], r[1024 + 256 + 10]);
    return bad != 0;
}
