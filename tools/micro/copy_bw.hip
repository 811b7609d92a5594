// Device-to-device copy kernels, 1 GiB: which shape streams fastest on this box (hot_copy_bandwidth uses the winner).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bw.hip -o tools/micro/copy_bw && tools/micro/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT)
                __builtin_nontemporal_store(v[k], dst + i + k * stride);
            else
                dst[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
template <int U, bool NT>
double run(const f4* a, f4* b, size_t n16, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, n16);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, n16);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * n16 * 16 * 20 / (ms * 1e-3) / 1e9;
}
int main()
{
    const size_t n16 = (1ull << 30) / 16;
    f4 *a, *b;
    hipMalloc(&a, n16 * 16), hipMalloc(&b, n16 * 16);
    hipMemset(a, 1, n16 * 16);
    for (int grid : { 1024, 2048, 4096, 8192, 16384, 65536, (int)(n16 / 256), (int)(n16 / 1024) }) {
        printf("grid %7d: U1 %.0f  U2 %.0f  U4 %.0f  U8 %.0f | nt U1 %.0f U4 %.0f U8 %.0f GB/s\n", grid, run<1, false>(a, b, n16, grid), run<2, false>(a, b, n16, grid), run<4, false>(a, b, n16, grid),
            run<8, false>(a, b, n16, grid), run<1, true>(a, b, n16, grid), run<4, true>(a, b, n16, grid), run<8, true>(a, b, n16, grid));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipMemcpyAsync(b, a, n16 * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyAsync D2D: %.0f GB/s\n", 2.0 * n16 * 16 * 20 / (ms * 1e-3) / 1e9);
    return 0;
}
