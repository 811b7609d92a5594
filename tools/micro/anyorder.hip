// (1) Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start before its predecessor in the SAME stream has finished (AQL barrier
//     bit cleared) on gfx950?  Eight one-workgroup kernels that spin ~100 us each: serial = 800 us, concurrent = ~100 us.
// (2) What does a dependency between two streams cost (hipEventRecord + hipStreamWaitEvent) against the launch boundary inside one stream?
//     A chain of 64 kernels of ~10 us, all in one stream / alternating between two streams.
// (3) Fork-join per link: A: k(20 us) ; B (after A's predecessor): k(20 us) concurrently ; join — the shape of "subst(c) || early(c+1)".
// hipcc --offload-arch=gfx950 -O3 tools/micro/anyorder.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long clocks, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(8);
    if (out) out[blockIdx.x] = 1;
}
int main()
{
    hipStream_t s, s2;
    hipStreamCreate(&s), hipStreamCreate(&s2);
    hipEvent_t e0, e1, ev[256];
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int* out;
    hipMalloc(&out, 4096);
    long long clocks = 10000; // wall_clock64: 100 MHz -> 100 us
    float ms;
    for (int flags = 0; flags < 2; ++flags)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            for (int k = 0; k < 8; ++k) {
                void* args[] = { &clocks, &out };
                hipError_t e = hipExtLaunchKernel((const void*)spin, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0);
                if (e != hipSuccess) printf("launch error %d\n", (int)e);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("(1) flags %d rep %d: 8 x 100 us kernels in %.3f ms\n", flags, rep, ms);
        }
    for (int two = 0; two < 2; ++two)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int k = 0; k < 64; ++k) {
                hipStream_t cur = (two && (k & 1)) ? s2 : s;
                if (two && k) hipStreamWaitEvent(cur, ev[k - 1], 0);
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, cur, 1000LL, out);
                if (two) hipEventRecord(ev[k], cur);
            }
            if (two) hipStreamWaitEvent(s, ev[63], 0);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("(2) %s rep %d: chain of 64 x 10 us kernels in %.3f ms (%.2f us per link beyond the kernel)\n", two ? "two streams" : "one stream ", rep, ms, (ms * 1e3 - 640) / 64);
        }
    for (int fork = 0; fork < 2; ++fork)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int k = 0; k < 32; ++k) {
                // late(c): 5 us ; then subst(c) 20 us || early(c+1) 20 us
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 500LL, out);
                if (fork) {
                    hipEventRecord(ev[2 * k], s);
                    hipStreamWaitEvent(s2, ev[2 * k], 0);
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, 2000LL, out);
                    hipEventRecord(ev[2 * k + 1], s2);
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 2000LL, out);
                    hipStreamWaitEvent(s, ev[2 * k + 1], 0);
                }
                else {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 2000LL, out);
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 2000LL, out);
                }
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("(3) %s rep %d: 32 x (5 us ; 20 us %s 20 us) in %.3f ms = %.2f us per link (ideal %d)\n", fork ? "fork-join" : "serial   ", rep, fork ? "||" : "; ", ms, ms * 1e3 / 32, fork ? 25 : 45);
        }
    return 0;
}
