// Micro-benchmark of the block-GS substitution phase (mg_solve.hip gs_phase_b): lane = row, SB steps, per step one LDS column of 3x3
// blocks, a broadcast of the finished row and nine FP64 FMAs.  Variants of the broadcast / layout, timed in isolation on one wavefront per
// workgroup (what the kernel does while the other waves idle).   hipcc --offload-arch=gfx950 -O3 phaseb.hip -o phaseb && ./phaseb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int SB = 32, TRI = SB * (SB - 1) / 2 + 1;
__device__ __forceinline__ int tri_fwd(int row, int colm) { return (SB - 1) * colm - (colm * (colm - 1)) / 2 + (row - colm - 1); }
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// V0: the production loop (readlane broadcast, one column ahead)
template <int VAR>
__global__ __launch_bounds__(64) void k(const double* __restrict__ gtri, const double* __restrict__ gsv, double* out, int reps)
{
    __shared__ double tri[9 * TRI];
    __shared__ double sv[3 * SB];
    __shared__ double bb[3 * SB];
    __shared__ __attribute__((aligned(16))) double trp[10 * TRI]; // paired planes: [(e / 2)][idx][2] (plane 4 holds e = 8 and a pad)
    const int lane = threadIdx.x, me = lane;
    for (int e = lane; e < 9 * TRI; e += 64) tri[e] = gtri[e];
    for (int e = lane; e < 3 * SB; e += 64) sv[e] = gsv[e];
    for (int e = lane; e < 10 * TRI; e += 64) {
        const int pl = e / (2 * TRI), rem = e - pl * 2 * TRI, idx = rem >> 1, h = rem & 1, ee = 2 * pl + h;
        trp[e] = ee < 9 ? gtri[ee * TRI + idx] : 0.0;
    }
    __syncthreads();
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        double a0 = me < SB ? sv[me * 3] : 0, a1 = me < SB ? sv[me * 3 + 1] : 0, a2 = me < SB ? sv[me * 3 + 2] : 0;
        auto load_col = [&](int cidx, double(&L)[9]) {
            bool act = me > cidx && me < SB;
            int idx = act ? tri_fwd(me, cidx) : TRI - 1;
#pragma unroll
            for (int e = 0; e < 9; ++e) L[e] = tri[e * TRI + idx];
        };
        if (VAR == 0) {
            auto step = [&](int cidx, const double(&L)[9]) {
                double b0 = lane_bcast(a0, cidx), b1 = lane_bcast(a1, cidx), b2 = lane_bcast(a2, cidx);
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            };
            double LA[9], LB[9];
            load_col(0, LA);
            int s = 0;
            for (; s + 1 < SB; s += 2) {
                load_col(s + 1, LB);
                step(s, LA);
                load_col(min(s + 2, SB - 1), LA);
                step(s + 1, LB);
            }
            if (s < SB) step(s, LA);
        }
        else if (VAR == 1) { // broadcast through LDS: the finished row writes its three values, everybody reads them (uniform address)
            double LA[9], LB[9];
            load_col(0, LA);
            auto step = [&](int cidx, const double(&L)[9]) {
                if (me == cidx) bb[3 * cidx] = a0, bb[3 * cidx + 1] = a1, bb[3 * cidx + 2] = a2;
                __builtin_amdgcn_wave_barrier();
                double b0 = bb[3 * cidx], b1 = bb[3 * cidx + 1], b2 = bb[3 * cidx + 2];
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            };
            int s = 0;
            for (; s + 1 < SB; s += 2) {
                load_col(s + 1, LB);
                step(s, LA);
                load_col(min(s + 2, SB - 1), LA);
                step(s + 1, LB);
            }
            if (s < SB) step(s, LA);
        }
        else if (VAR == 2) { // no broadcast at all (wrong result): the cost of everything else
            double LA[9], LB[9];
            load_col(0, LA);
            auto step = [&](int cidx, const double(&L)[9]) {
                double b0 = a0, b1 = a1, b2 = a2;
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            };
            int s = 0;
            for (; s + 1 < SB; s += 2) {
                load_col(s + 1, LB);
                step(s, LA);
                load_col(min(s + 2, SB - 1), LA);
                step(s + 1, LB);
            }
            if (s < SB) step(s, LA);
        }
        else if (VAR == 3) { // readlane broadcast, no LDS (columns = constants): the arithmetic / readlane chain alone
            double L[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) L[e] = 1e-3 * (e + 1 + lane);
            for (int s = 0; s < SB; ++s) {
                double b0 = lane_bcast(a0, s), b1 = lane_bcast(a1, s), b2 = lane_bcast(a2, s);
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            }
        }
        else if (VAR == 4) { // two-level: 4-step diagonal blocks sequential (readlane), the rest of each 4-column panel applied as a batch
            // columns are applied to later rows only when needed: row r needs columns < r; within a panel of 4 the chain is sequential,
            // rows beyond the panel take the panel's 4 columns with independent FMAs (no dependency on each other)
            for (int p0 = 0; p0 < SB; p0 += 4) {
                double P[4][9];
#pragma unroll
                for (int q = 0; q < 4; ++q) load_col(p0 + q, P[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = p0 + q;
                    double b0 = lane_bcast(a0, c), b1 = lane_bcast(a1, c), b2 = lane_bcast(a2, c);
                    a0 = fma(P[q][0], b0, a0), a1 = fma(P[q][1], b0, a1), a2 = fma(P[q][2], b0, a2);
                    a0 = fma(P[q][3], b1, a0), a1 = fma(P[q][4], b1, a1), a2 = fma(P[q][5], b1, a2);
                    a0 = fma(P[q][6], b2, a0), a1 = fma(P[q][7], b2, a1), a2 = fma(P[q][8], b2, a2);
                }
            }
        }
        else if (VAR == 5 || VAR == 6) { // three buffers (two columns ahead); 5: readlane, 6: LDS broadcast
            auto step = [&](int cidx, const double(&L)[9]) {
                double b0, b1, b2;
                if (VAR == 5)
                    b0 = lane_bcast(a0, cidx), b1 = lane_bcast(a1, cidx), b2 = lane_bcast(a2, cidx);
                else {
                    if (me == cidx) bb[3 * cidx] = a0, bb[3 * cidx + 1] = a1, bb[3 * cidx + 2] = a2;
                    __builtin_amdgcn_wave_barrier();
                    b0 = bb[3 * cidx], b1 = bb[3 * cidx + 1], b2 = bb[3 * cidx + 2];
                }
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            };
            double LA[9], LB[9], LC[9];
            auto lc = [&](int st, double(&L)[9]) { load_col(min(st, SB - 1), L); };
            lc(0, LA), lc(1, LB);
            int s = 0;
            for (; s + 2 < SB; s += 3) {
                lc(s + 2, LC);
                step(s, LA);
                lc(s + 3, LA);
                step(s + 1, LB);
                lc(s + 4, LB);
                step(s + 2, LC);
            }
            if (s < SB) step(s, LA);
            if (s + 1 < SB) step(s + 1, LB);
        }
        else if (VAR == 7 || VAR == 8) { // paired planes: five loads per column (four 16-byte, one 8-byte); 7: LDS broadcast, 8: readlane
            auto load_colp = [&](int cidx, double(&L)[9]) {
                bool act = me > cidx && me < SB;
                int idx = act ? tri_fwd(me, cidx) : TRI - 1;
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) {
                    const double2 v = *(const double2*)(trp + (size_t)pl * 2 * TRI + 2 * idx);
                    L[2 * pl] = v.x, L[2 * pl + 1] = v.y;
                }
                L[8] = trp[(size_t)8 * TRI + 2 * idx];
            };
            auto step = [&](int cidx, const double(&L)[9]) {
                double b0, b1, b2;
                if (VAR == 8)
                    b0 = lane_bcast(a0, cidx), b1 = lane_bcast(a1, cidx), b2 = lane_bcast(a2, cidx);
                else {
                    if (me == cidx) bb[3 * cidx] = a0, bb[3 * cidx + 1] = a1, bb[3 * cidx + 2] = a2;
                    __builtin_amdgcn_wave_barrier();
                    b0 = bb[3 * cidx], b1 = bb[3 * cidx + 1], b2 = bb[3 * cidx + 2];
                }
                a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
                a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
                a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            };
            double LA[9], LB[9];
            load_colp(0, LA);
            int s = 0;
            for (; s + 1 < SB; s += 2) {
                load_colp(s + 1, LB);
                step(s, LA);
                load_colp(min(s + 2, SB - 1), LA);
                step(s + 1, LB);
            }
            if (s < SB) step(s, LA);
        }
        acc += a0 + a1 + a2;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int VAR>
float run(const double* t, const double* s, double* o, int grid, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k<VAR><<<grid, 64>>>(t, s, o, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<VAR><<<grid, 64>>>(t, s, o, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main()
{
    std::vector<double> t(9 * TRI), s(3 * SB);
    for (size_t i = 0; i < t.size(); ++i) t[i] = 1e-3 * ((i * 2654435761u) % 1000) / 1000.0;
    t[9 * TRI - 1] = 0; // (the zero entry of every plane is the last of plane 0..8: set all)
    for (int e = 0; e < 9; ++e) t[e * TRI + TRI - 1] = 0;
    for (size_t i = 0; i < s.size(); ++i) s[i] = 1.0 + 0.01 * i;
    double *dt, *ds, *dout;
    hipMalloc(&dt, t.size() * 8), hipMalloc(&ds, s.size() * 8), hipMalloc(&dout, 1024 * 64 * 8);
    hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice), hipMemcpy(ds, s.data(), s.size() * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int grid : { 1, 512 }) {
        float m0 = run<0>(dt, ds, dout, grid, reps), m1 = run<1>(dt, ds, dout, grid, reps), m2 = run<2>(dt, ds, dout, grid, reps), m3 = run<3>(dt, ds, dout, grid, reps), m4 = run<4>(dt, ds, dout, grid, reps);
        float m5 = run<5>(dt, ds, dout, grid, reps), m6 = run<6>(dt, ds, dout, grid, reps), m7 = run<7>(dt, ds, dout, grid, reps), m8 = run<8>(dt, ds, dout, grid, reps);
        printf("grid %4d: ns per step: production %.1f | LDS broadcast %.1f | no broadcast %.1f | readlane+fma only %.1f | 4-column panels %.1f\n", grid, 1e6 * m0 / reps / SB, 1e6 * m1 / reps / SB,
            1e6 * m2 / reps / SB, 1e6 * m3 / reps / SB, 1e6 * m4 / reps / SB);
        printf("           3 buffers readlane %.1f | 3 buffers LDS broadcast %.1f | paired planes LDS broadcast %.1f | paired planes readlane %.1f\n", 1e6 * m5 / reps / SB, 1e6 * m6 / reps / SB, 1e6 * m7 / reps / SB, 1e6 * m8 / reps / SB);
        double h[64];
        for (int v = 0; v < 9; ++v) {
            switch (v) { case 0: run<0>(dt, ds, dout, 1, 1); break; case 1: run<1>(dt, ds, dout, 1, 1); break; case 5: run<5>(dt, ds, dout, 1, 1); break; case 6: run<6>(dt, ds, dout, 1, 1); break; case 7: run<7>(dt, ds, dout, 1, 1); break; case 8: run<8>(dt, ds, dout, 1, 1); break; default: continue; }
            hipMemcpy(h, dout, 64 * 8, hipMemcpyDeviceToHost);
            printf("  variant %d lane 31 result %.15g\n", v, h[31]);
        }
    }
    double h[64];
    hipMemcpy(h, dout, 64 * 8, hipMemcpyDeviceToHost);
    printf("check %g\n", h[5]);
    return 0;
}
