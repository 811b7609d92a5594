// Microbenchmark: what bounds a lone wavefront streaming 72-byte entries to a few of its lanes per step (k_gs_subst's access pattern):
//  mode 0  every lane loads the same (zero) entry                         (5 x 64 x 16 B returned per step)
//  mode 1  12 lanes load consecutive entries, 52 the zero entry           (the kernel's pattern)
//  mode 2  12 lanes load consecutive entries, 52 are out of range of a buffer resource (returns 0, no request)
//  mode 3  only 12 lanes are alive (static EXEC)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <int MODE, int D>
__global__ __launch_bounds__(64) void k_ret(const double* img, double* out, size_t per_block)
{
    const int lane = threadIdx.x;
    const double* ent = img + (size_t)blockIdx.x * per_block;
    if (MODE == 3 && lane >= 12) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ent, 0, (int)(per_block * 8), 0x00020000);
    double acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    double ring[D][9];
    auto issue = [&](int s, double (&L)[9]) __attribute__((always_inline)) {
        const bool mine = lane < 12;
        unsigned idx = mine ? 1u + 12u * s + lane : 0u;
        if (MODE == 0) idx = 0;
        if (MODE == 2) {
            const unsigned off = mine ? idx * 72u : 0x7fffff00u;
            v4i a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
            v4i c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
            v2i e = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 64, 0, 0);
            int w[18] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w, e.x, e.y };
            for (int q = 0; q < 9; ++q) L[q] = __longlong_as_double(((long long)w[2 * q + 1] << 32) | (unsigned)w[2 * q]);
        }
        else {
            const double* p = ent + (size_t)idx * 9;
            for (int q = 0; q < 9; ++q) L[q] = p[q];
        }
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int k = 0; k < D; ++k) issue(k, ring[k]);
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        double(&L)[9] = ring[s % D];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] += L[q];
        if (s + D < 64) issue(s + D, L);
    }
    double t = 0;
    for (int q = 0; q < 9; ++q) t += acc[q];
    out[blockIdx.x * 64 + lane] = t;
}
template <int MODE>
void run(const char* name, int nblk, const double* img, double* out, size_t per_block)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k_ret<MODE, 8><<<nblk, 64>>>(img, out, per_block);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k_ret<MODE, 8><<<nblk, 64>>>(img, out, per_block);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks %5d : %.2f us per launch\n", name, nblk, ms * 1e3 / 20);
}
int main()
{
    const size_t per_block = 37458; // doubles, as GsImg<double>::per_block
    const int maxb = 8192;
    double *img, *out;
    hipMalloc(&img, per_block * 8 * maxb), hipMalloc(&out, maxb * 64 * 8);
    hipMemset(img, 0, per_block * 8 * maxb);
    for (int nblk : { 729, 5832 }) {
        run<0>("all lanes the zero entry", nblk, img, out, per_block);
        run<1>("12 lanes entries + 52 lanes zero entry", nblk, img, out, per_block);
        run<2>("12 lanes entries + 52 lanes out of range", nblk, img, out, per_block);
        run<3>("12 lanes alive", nblk, img, out, per_block);
    }
    return 0;
}
