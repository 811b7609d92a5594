// Micro-benchmark (gfx950): issue rate of v_fmac_f64_dpp row_newbcast against plain v_fma_f64, v_fmac_f32_dpp, v_readlane,
// ds_read_b64 with one address for the whole wavefront.  hipcc --offload-arch=gfx950 -O3 dpp_fma.hip -o dpp_fma && ./dpp_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define DPP64(acc, k, g, n) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(k), "v"(g))
#define DPP32(acc, k, g, n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(k), "v"(g))

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double* out, const double* in, int iters, unsigned long long* clk)
{
    __shared__ double sh[512];
    const int tid = threadIdx.x;
    sh[tid] = in[tid], sh[tid + 256] = in[tid + 256];
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, k = in[tid], g0 = in[tid + 64], g1 = in[tid + 128], g2 = in[tid + 192];
    float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, kf = (float)k, h0 = (float)g0, h1 = (float)g1, h2 = (float)g2;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // 15 DPP fp64 FMAs, the pattern of the assembly kernel
            DPP64(a0, k, g0, 0); DPP64(a1, k, g0, 3); DPP64(a2, k, g0, 6); DPP64(a3, k, g0, 9); DPP64(a4, k, g0, 12);
            DPP64(a0, k, g1, 1); DPP64(a1, k, g1, 4); DPP64(a2, k, g1, 7); DPP64(a3, k, g1, 10); DPP64(a4, k, g1, 13);
            DPP64(a0, k, g2, 2); DPP64(a1, k, g2, 5); DPP64(a2, k, g2, 8); DPP64(a3, k, g2, 11); DPP64(a4, k, g2, 14);
        }
        else if (MODE == 1) { // 15 plain fp64 FMAs
            asm volatile("v_fma_f64 %0, %5, %6, %0\n v_fma_f64 %1, %5, %6, %1\n v_fma_f64 %2, %5, %6, %2\n v_fma_f64 %3, %5, %6, %3\n v_fma_f64 %4, %5, %6, %4\n"
                         "v_fma_f64 %0, %5, %7, %0\n v_fma_f64 %1, %5, %7, %1\n v_fma_f64 %2, %5, %7, %2\n v_fma_f64 %3, %5, %7, %3\n v_fma_f64 %4, %5, %7, %4\n"
                         "v_fma_f64 %0, %5, %8, %0\n v_fma_f64 %1, %5, %8, %1\n v_fma_f64 %2, %5, %8, %2\n v_fma_f64 %3, %5, %8, %3\n v_fma_f64 %4, %5, %8, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(k), "v"(g0), "v"(g1), "v"(g2));
        }
        else if (MODE == 2) { // 15 DPP fp32 FMAs
            DPP32(f0, kf, h0, 0); DPP32(f1, kf, h0, 3); DPP32(f2, kf, h0, 6); DPP32(f3, kf, h0, 9); DPP32(f4, kf, h0, 12);
            DPP32(f0, kf, h1, 1); DPP32(f1, kf, h1, 4); DPP32(f2, kf, h1, 7); DPP32(f3, kf, h1, 10); DPP32(f4, kf, h1, 13);
            DPP32(f0, kf, h2, 2); DPP32(f1, kf, h2, 5); DPP32(f2, kf, h2, 8); DPP32(f3, kf, h2, 11); DPP32(f4, kf, h2, 14);
        }
        else if (MODE == 3) { // 15 plain fp32 FMAs
            asm volatile("v_fmac_f32 %0, %5, %6\n v_fmac_f32 %1, %5, %6\n v_fmac_f32 %2, %5, %6\n v_fmac_f32 %3, %5, %6\n v_fmac_f32 %4, %5, %6\n"
                         "v_fmac_f32 %0, %5, %7\n v_fmac_f32 %1, %5, %7\n v_fmac_f32 %2, %5, %7\n v_fmac_f32 %3, %5, %7\n v_fmac_f32 %4, %5, %7\n"
                         "v_fmac_f32 %0, %5, %8\n v_fmac_f32 %1, %5, %8\n v_fmac_f32 %2, %5, %8\n v_fmac_f32 %3, %5, %8\n v_fmac_f32 %4, %5, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(kf), "v"(h0), "v"(h1), "v"(h2));
        }
        else if (MODE == 4) { // 15 DPP FMAs + 3 wave-uniform ds_read_b64 + 3 K FMAs (one row step of the assembly kernel)
            const double* s = sh + ((it * 3) & 255);
            double w0, w1, w2;
            asm volatile("ds_read_b64 %0, %3\n ds_read_b64 %1, %3 offset:8\n ds_read_b64 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(w0), "=v"(w1), "=v"(w2) : "v"((unsigned)(size_t)s) : "memory");
            double kk = g0 * w0;
            kk = fma(g1, w1, kk), kk = fma(g2, w2, kk);
            DPP64(a0, kk, g0, 0); DPP64(a1, kk, g0, 3); DPP64(a2, kk, g0, 6); DPP64(a3, kk, g0, 9); DPP64(a4, kk, g0, 12);
            DPP64(a0, kk, g1, 1); DPP64(a1, kk, g1, 4); DPP64(a2, kk, g1, 7); DPP64(a3, kk, g1, 10); DPP64(a4, kk, g1, 13);
            DPP64(a0, kk, g2, 2); DPP64(a1, kk, g2, 5); DPP64(a2, kk, g2, 8); DPP64(a3, kk, g2, 11); DPP64(a4, kk, g2, 14);
        }
        else if (MODE == 5) { // 15 v_readlane_b32 into SGPRs
            int lane = it & 31, r;
#pragma unroll
            for (int q = 0; q < 15; ++q) {
                asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(r) : "v"(tid + q), "s"(lane));
                lane ^= r & 1;
            }
            a0 += lane;
        }
        else if (MODE == 6) { // 15 ds_read_b64, one address for the wavefront
            const unsigned s = (unsigned)(size_t)(sh + ((it * 16) & 255));
            double w[15];
#pragma unroll
            for (int q = 0; q < 15; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[q]) : "v"(s), "n"(q * 8) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 15; ++q) a0 += w[q];
        }
        else if (MODE == 7) { // 15 ds_read_b64, lane addresses 24 bytes apart
            const unsigned s = (unsigned)(size_t)(sh + ((it * 16) & 63) + (tid & 63) * 3);
            double w[15];
#pragma unroll
            for (int q = 0; q < 15; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[q]) : "v"(s), "n"(q * 8) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 15; ++q) a0 += w[q];
        }
    }
    const unsigned long long t1 = clock64();
    out[(size_t)blockIdx.x * 256 + tid] = a0 + a1 + a2 + a3 + a4 + f0 + f1 + f2 + f3 + f4;
    if (tid == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int MODE>
int run(const char* name, int wgs, double* out, const double* in, unsigned long long* clk, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(wgs), dim3(256), 0, 0, out, in, 10, clk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(wgs), dim3(256), 0, 0, out, in, iters, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    // wavefront instructions per SIMD: wgs * 4 waves * iters * 15 / (256 CUs * 4 SIMDs)
    const double winst = (double)wgs * 4 * iters * 15 / 1024.0;
    printf("%-44s wgs %5d: %8.3f ms, %6.2f ns per wavefront instruction and SIMD, clock64 per instruction of wave 0: %.2f\n", name, wgs, ms, ms * 1e6 / winst, (double)c / (iters * 15.0));
    return 0;
}

int main()
{
    double *out, *in;
    unsigned long long* clk;
    CK(hipMalloc(&out, 8 * 256 * 8192)); CK(hipMalloc(&in, 8 * 512)); CK(hipMalloc(&clk, 8));
    std::vector<double> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 1e-3 * (i % 7);
    CK(hipMemcpy(in, h.data(), 8 * 512, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int wgs : { 256, 1024, 2048 }) { // 1, 4, 8 waves per SIMD
        run<0>("v_fmac_f64_dpp row_newbcast", wgs, out, in, clk, iters);
        run<1>("v_fma_f64", wgs, out, in, clk, iters);
        run<2>("v_fmac_f32_dpp row_newbcast", wgs, out, in, clk, iters);
        run<3>("v_fmac_f32", wgs, out, in, clk, iters);
        run<4>("row step: 3 uniform ds_read + 3 fma + 15 dpp (per 15)", wgs, out, in, clk, iters);
        run<5>("v_readlane_b32", wgs, out, in, clk, iters);
        run<6>("ds_read_b64 one address", wgs, out, in, clk, iters);
        run<7>("ds_read_b64 24-byte stride", wgs, out, in, clk, iters);
    }
    return 0;
}
