// Microbenchmark: what a 64-step substitution chain costs a lone wavefront (no memory): 6 v_readlane + 9 DP FMA per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <class T> __device__ __forceinline__ T bc(T v, int c);
template <> __device__ __forceinline__ float bc(float v, int c) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)); }
template <> __device__ __forceinline__ double bc(double v, int c)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), c), hi = __builtin_amdgcn_readlane((int)(b >> 32), c);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <class T, int MODE>
__global__ __launch_bounds__(64) void k_chain(const T* in, T* out, int reps)
{
    const int lane = threadIdx.x;
    T L[9];
    for (int e = 0; e < 9; ++e) L[e] = in[(blockIdx.x * 64 + lane) * 9 + e];
    T a0 = in[lane], a1 = in[lane + 64], a2 = in[lane + 128];
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            T b0, b1, b2;
            if (MODE == 0) b0 = bc(a0, s), b1 = bc(a1, s), b2 = bc(a2, s);
            if (MODE == 1) b0 = a0, b1 = a1, b2 = a2; // no cross-lane step: the FMA chain alone
            if (MODE == 2) b0 = bc(a0, s), b1 = b0, b2 = b0; // one third of the readlanes
            a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
            a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
            a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
        }
    }
    out[blockIdx.x * 192 + lane] = a0, out[blockIdx.x * 192 + 64 + lane] = a1, out[blockIdx.x * 192 + 128 + lane] = a2;
}
template <class T, int MODE>
void run(const char* name, int nblk)
{
    T *in, *out;
    hipMalloc(&in, sizeof(T) * nblk * 64 * 9 + 4096), hipMalloc(&out, sizeof(T) * nblk * 192);
    hipMemset(in, 0, sizeof(T) * nblk * 64 * 9 + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int reps : { 1, 11 }) {
        k_chain<T, MODE><<<nblk, 64>>>(in, out, reps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) k_chain<T, MODE><<<nblk, 64>>>(in, out, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s blocks %5d reps %2d : %.2f us per launch\n", name, nblk, reps, ms * 1e3 / 50);
    }
    hipFree(in), hipFree(out);
}
int main()
{
    for (int nblk : { 729, 4096 }) {
        run<double, 0>("f64 readlane x6 + 9 fma", nblk);
        run<double, 1>("f64 9 fma only", nblk);
        run<double, 2>("f64 readlane x2 + 9 fma", nblk);
        run<float, 0>("f32 readlane x3 + 9 fma", nblk);
        run<float, 1>("f32 9 fma only", nblk);
    }
    return 0;
}
