// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this GPU / runtime?
// Build: hipcc -O2 --offload-arch=gfx950 tools/anyorder_probe.hip -o /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>

__global__ void spin(long long cycles, int *out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[blockIdx.x] = 1;
}

static double run(int flags_second, int reps, hipStream_t s, int *buf)
{
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, 5000LL, buf);                               // ~50 us at 100 MHz
        hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, flags_second, 5000LL, buf + 64);
    }
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *buf;
    hipMalloc(&buf, 4096);
    run(0, 5, s, buf);
    printf("two 50-us kernels, in order : %.1f us per pair\n", run(0, 50, s, buf));
    printf("second with AnyOrderLaunch  : %.1f us per pair\n", run(hipExtAnyOrderLaunch, 50, s, buf));
    printf("in order again              : %.1f us per pair\n", run(0, 50, s, buf));
    // host cost of the launch calls alone
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 0LL, (int *)nullptr);
    double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    hipStreamSynchronize(s);
    printf("hipExtLaunchKernel host cost: %.2f us\n", enq);
    // event fork/join host cost for comparison
    hipStream_t side; hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreateWithFlags(&a, hipEventDisableTiming); hipEventCreateWithFlags(&b, hipEventDisableTiming);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) {
        hipEventRecord(a, s); hipStreamWaitEvent(side, a, 0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, side, 0LL, (int *)nullptr);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 0LL, (int *)nullptr);
        hipEventRecord(b, side); hipStreamWaitEvent(s, b, 0);
    }
    enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    hipStreamSynchronize(s);
    double all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    printf("event fork/join of two empty kernels: enqueue %.2f us, complete %.2f us per pair\n", enq, all);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 0LL, (int *)nullptr);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 0LL, (int *)nullptr);
    }
    enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    hipStreamSynchronize(s);
    all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    printf("two empty kernels in one stream      : enqueue %.2f us, complete %.2f us per pair\n", enq, all);
    return 0;
}
