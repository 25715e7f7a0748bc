// Dev tool: LDS accumulate micro-benchmark on gfx950 (decides the grad_value tile design).
// hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: f32 atomic AoS [pix][4]; 1: f32 atomic SoA [4][npx]; 2: u32 atomic AoS; 3: u32 SoA; 4: u64 atomic SoA
// 5: non-atomic f32 RMW SoA; 9: u32 atomic with return SoA; 6: f32 atomic with return SoA; 7: f32 atomic, 1 channel only (SoA); 8: f64 atomic SoA
template <int MODE> __global__ __launch_bounds__(1024) void k(int npx, int iters, float *sink)
{
    float *f = reinterpret_cast<float *>(smem);
    uint32_t *u = reinterpret_cast<uint32_t *>(smem);
    unsigned long long *q = reinterpret_cast<unsigned long long *>(smem);
    double *d = reinterpret_cast<double *>(smem);
    const int words = (MODE == 4 || MODE == 8) ? npx * 8 : npx * 4;
    for (int i = threadIdx.x; i < words; i += blockDim.x) u[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t pix = lcg(s) % (uint32_t)npx;
        const float v = (float)(pix & 7) * 0.125f + 0.5f;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&f[pix * 4 + c], v);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&f[c * npx + pix], v);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&u[pix * 4 + c], (uint32_t)pix);
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&u[c * npx + pix], (uint32_t)pix);
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&q[c * npx + pix], (unsigned long long)pix);
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int c = 0; c < 4; ++c) f[c * npx + pix] += v;
        } else if constexpr (MODE == 6) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += atomicAdd(&f[c * npx + pix], v);
        } else if constexpr (MODE == 7) {
            atomicAdd(&f[pix], v);
        } else if constexpr (MODE == 9) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += (float)atomicAdd(&u[c * npx + pix], 1u);
        } else if constexpr (MODE == 8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&d[c * npx + pix], (double)v);
        }
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < words; i += blockDim.x) t += f[i];
    if (t == 123.456f) sink[0] = t;
}

template <int MODE> int run(const char *name, int npx, int iters, float *sink)
{
    const size_t lds = (size_t)npx * ((MODE == 4 || MODE == 8) ? 32 : 16);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, npx, iters, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, npx, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const int per = (MODE == 7) ? 1 : 4;
    const double wave_instr_per_cu = 16.0 * iters * per;  // 16 waves per CU
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s npx=%5d  %8.3f ms  ~%6.1f cycles per wave-instr (at 2.4 GHz)\n", name, npx, ms, cycles / wave_instr_per_cu);
    return 0;
}

int main()
{
    float *sink;
    CHECK(hipMalloc(&sink, 4));
    const int iters = 2000;
    for (int npx : {5440, 1344, 64}) {
        run<0>("f32 atomic  AoS [pix][4]", npx, iters, sink);
        run<1>("f32 atomic  SoA [4][npx]", npx, iters, sink);
        run<2>("u32 atomic  AoS", npx, iters, sink);
        run<3>("u32 atomic  SoA", npx, iters, sink);
        if (npx * 32 <= 160 * 1024) run<4>("u64 atomic  SoA", npx, iters, sink);
        run<5>("f32 plain RMW SoA (racy)", npx, iters, sink);
        run<6>("f32 atomic rtn SoA", npx, iters, sink);
        run<7>("f32 atomic 1 channel", npx, iters, sink);
        run<9>("u32 atomic rtn SoA", npx, iters, sink);
        if (npx * 32 <= 160 * 1024) run<8>("f64 atomic  SoA", npx, iters, sink);
    }
    return 0;
}
