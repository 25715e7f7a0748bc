// Dev tool: how fast does gfx950 dispatch workgroups?  (Is a kernel with ~15k short workgroups launch-bound?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_empty(int *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1; }
__global__ __launch_bounds__(256) void k_load(const int *p, int *o) { int v = p[blockIdx.x & 1023]; if (v == 123456789) o[0] = v; }
__global__ __launch_bounds__(256) void k_load2(const int4 *p, int *o) {
    int v = p[0].x;  // dependent chain of two loads
    int4 w = p[(blockIdx.x * 32 + (threadIdx.x >> 3)) * 3 + (v & 0)];
    if (w.x == 123456789) o[0] = w.y;
}
template <int VG> __global__ __launch_bounds__(256) void k_regs(const int *p, int *o) {
    int acc[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) acc[i] = p[(threadIdx.x + i) & 1023];
    int s = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += acc[i] * (i + 1);
    if (s == 123456789) o[0] = s;
}

template <typename F> float time_it(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main() {
    int *buf; int4 *buf4;
    CHECK(hipMalloc(&buf, 1 << 20)); CHECK(hipMemset(buf, 0, 1 << 20));
    CHECK(hipMalloc(&buf4, 64 << 20)); CHECK(hipMemset(buf4, 0, 64 << 20));
    for (int blocks : {1024, 4096, 10016, 15456, 40000}) {
        float t1 = time_it([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, buf); });
        float t3 = time_it([&] { hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, 0, buf, buf); });
        float t4 = time_it([&] { hipLaunchKernelGGL(k_load2, dim3(blocks), dim3(256), 0, 0, buf4, buf); });
        float t5 = time_it([&] { hipLaunchKernelGGL((k_regs<64>), dim3(blocks), dim3(256), 0, 0, buf, buf); });
        float t6 = time_it([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 48 * 1024, 0, buf); });
        float t7 = time_it([&] { hipLaunchKernelGGL(k_empty, dim3(8, blocks / 32, 4), dim3(256), 0, 0, buf); });
        printf("blocks %6d: empty %7.1f us | 1 load %7.1f | 2 dep loads %7.1f | 64 regs+loads %7.1f | empty+48KB LDS %7.1f | empty 3-D %7.1f\n",
               blocks, t1, t3, t4, t5, t6, t7);
    }
    return 0;
}
