// scratch: cost of a software pipeline of dependent kernels over two HIP streams (events as cross-stream edges)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_stream(const double* __restrict__ a, double* out, size_t n, int iters)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (int it = 0; it < iters; ++it)
        for (size_t j = i; j < n; j += (size_t)gridDim.x * blockDim.x) s += a[j];
    if (s == 123.456) out[0] = s;
}
__global__ void k_spin(double* out, int cycles)
{
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] += 1;
}
int main()
{
    size_t n = 80 * 1024 * 1024 / 8; // 80 MB
    double *a, *out;
    hipMalloc(&a, n * 8);
    hipMalloc(&out, 64);
    hipMemset(a, 0, n * 8);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const int P = 64;
    std::vector<hipEvent_t> eb(P), ef(P);
    for (int p = 0; p < P; ++p) hipEventCreateWithFlags(&eb[p], hipEventDisableTiming), hipEventCreateWithFlags(&ef[p], hipEventDisableTiming);
    auto run = [&](int mode) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 10; ++rep)
            for (int p = 0; p < P; ++p) {
                if (mode == 0) { // serial: bulk then finish on one stream
                    hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, sa, a, out, n, 1);
                    hipLaunchKernelGGL(k_spin, dim3(575), dim3(512), 36 * 1024, sa, out, 12 * 2100);
                }
                else { // pipelined: bulk(p) on sa waits finish(p-2); finish(p) on sb waits bulk(p)
                    if (p >= 2) hipStreamWaitEvent(sa, ef[p - 2], 0);
                    hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, sa, a, out, n, 1);
                    hipEventRecord(eb[p], sa);
                    hipStreamWaitEvent(sb, eb[p], 0);
                    hipLaunchKernelGGL(k_spin, dim3(575), dim3(512), 36 * 1024, sb, out, 12 * 2100);
                    hipEventRecord(ef[p], sb);
                }
            }
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d: %.1f us per pass\n", mode, ms * 1e3 / (10 * P));
    };
    run(0), run(0), run(1), run(1);
    return 0;
}
