// ORACLE tooling (test infrastructure only).
//
// Golden-vector generator that runs the REAL reference SPGrid code: it includes the reference headers
// where they lie under /root/reference/Lib/SPGrid/Core (never copied into this repo) and links the
// reference's SPGrid_Utilities.cpp / SPGrid_Geometry.cpp (see oracle/Makefile, target _ref/spgrid_ref).
//
// What it exercises, per struct size (64 B == GridState<float,3>, 128 B == GridState<double,3>,
// reference Lib/MPM/MpmGrid.h:15-34,108-115):
//   * SPGrid_Mask::Linear_Offset / LinearToCoord / Packed_Add  (SPGrid_Mask.h:150-157,182-189,237-245)
//   * SPGrid_Page_Map::Set_Page / Get_Blocks insertion order     (SPGrid_Page_Map.h:61-70,90-96)
//   * the integer half of sortParticlesAndPolluteGrid (reference Lib/MPM/MpmSimulationBase.cpp:1066-1137)
//     replayed with the reference's own Mask and Page_Map classes: sort keys, particle_order,
//     particle_base_offset, particle_group, block_offset, block list order.
//   * the getNumNodes numbering rule (reference Lib/MPM/MpmGrid.h:148-161) given a "mass != 0" predicate
//     (a node is massive iff some particle's 3x3x3 kernel with non-zero weight touches it; the driver
//     marks every kernel node, matching w>0 for all seeded positions which avoid exact cell faces).
//
// Output: JSON on stdout.  Usage: spgrid_ref <float|double> <n_particles> <seed>
#include <SPGrid/Core/SPGrid_Allocator.h>
#include <SPGrid/Core/SPGrid_Page_Map.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

using namespace SPGrid;

template <int BYTES>
struct Node {
    char pad[BYTES];
};

static uint64_t rng_state;
static inline uint64_t xorshift64star()
{
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1DULL;
}
static inline double uniform01() { return (double)(xorshift64star() >> 11) * (1.0 / 9007199254740992.0); }

static inline int int_floor(double x)
{
    int i = (int)x;
    return i - (i > x);
}
static inline int int_floor(float x)
{
    int i = (int)x;
    return i - (i > x);
}

template <class T, int BYTES>
void run(int np, uint64_t seed)
{
    using Alloc = SPGrid_Allocator<Node<BYTES>, 3, 12>;
    using Mask = typename Alloc::template Array_type<>::MASK;
    using PageMap = SPGrid_Page_Map<12>;
    Alloc alloc(4096, 4096, 4096);
    PageMap page_map(alloc);

    printf("{\n\"struct_bytes\": %d, \"data_bits\": %d, \"block_bits\": %d,\n", BYTES, (int)Mask::data_bits, (int)Mask::block_bits);
    printf("\"block_xbits\": %d, \"block_ybits\": %d, \"block_zbits\": %d,\n", (int)Mask::block_xbits, (int)Mask::block_ybits, (int)Mask::block_zbits);

    // ---- 1. coordinate -> offset table, round trip, packed add
    int coords[][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 4, 4, 4 }, { 3, 3, 3 }, { 2, 4, 4 }, { 500, 500, 500 },
        { 501, 502, 503 }, { 4095, 4095, 4095 }, { 1023, 2048, 77 }, { 511, 512, 513 }, { 7, 8, 9 }, { 1234, 5, 4000 } };
    int nc = sizeof(coords) / sizeof(coords[0]);
    printf("\"coords\": [");
    for (int c = 0; c < nc; ++c) printf("%s[%d,%d,%d]", c ? "," : "", coords[c][0], coords[c][1], coords[c][2]);
    printf("],\n\"offsets\": [");
    for (int c = 0; c < nc; ++c) printf("%s%llu", c ? "," : "", (unsigned long long)Mask::Linear_Offset(coords[c][0], coords[c][1], coords[c][2]));
    printf("],\n\"roundtrip\": [");
    for (int c = 0; c < nc; ++c) {
        auto r = Mask::LinearToCoord(Mask::Linear_Offset(coords[c][0], coords[c][1], coords[c][2]));
        printf("%s[%d,%d,%d]", c ? "," : "", r[0], r[1], r[2]);
    }
    // random coordinates + random small deltas: Packed_Add(Linear_Offset(a), Linear_Offset(d))
    rng_state = seed * 0x9E3779B97F4A7C15ULL + 1;
    printf("],\n\"rand_coords\": [");
    std::vector<std::array<int, 6>> rc(256);
    for (int c = 0; c < 256; ++c) {
        for (int d = 0; d < 3; ++d) rc[c][d] = (int)(xorshift64star() % 4090);
        for (int d = 0; d < 3; ++d) rc[c][3 + d] = (int)(xorshift64star() % 6);
        printf("%s[%d,%d,%d,%d,%d,%d]", c ? "," : "", rc[c][0], rc[c][1], rc[c][2], rc[c][3], rc[c][4], rc[c][5]);
    }
    printf("],\n\"rand_offsets\": [");
    for (int c = 0; c < 256; ++c) printf("%s%llu", c ? "," : "", (unsigned long long)Mask::Linear_Offset(rc[c][0], rc[c][1], rc[c][2]));
    printf("],\n\"rand_packed_add\": [");
    for (int c = 0; c < 256; ++c)
        printf("%s%llu", c ? "," : "",
            (unsigned long long)Mask::Packed_Add(Mask::Linear_Offset(rc[c][0], rc[c][1], rc[c][2]), Mask::Linear_Offset(rc[c][3], rc[c][4], rc[c][5])));
    printf("],\n");

    // ---- 2. seeded particle cloud (two blobs so that the Morton order is non-trivial) and the sort
    const T dx = (T)0.01;
    std::vector<std::array<T, 3>> X(np);
    rng_state = seed * 0xD1B54A32D192ED03ULL + 7;
    for (int p = 0; p < np; ++p) {
        double cx = (p & 1) ? 5.0 : 5.13, cy = (p & 1) ? 5.02 : 4.93, cz = (p & 1) ? 4.97 : 5.11;
        for (int d = 0; d < 3; ++d) {
            double c = d == 0 ? cx : (d == 1 ? cy : cz);
            X[p][d] = (T)(c + 0.085 * (uniform01() - 0.5));
        }
    }
    printf("\"dx\": %.17g,\n\"X\": [", (double)dx);
    for (int p = 0; p < np; ++p) printf("%s[%.17g,%.17g,%.17g]", p ? "," : "", (double)X[p][0], (double)X[p][1], (double)X[p][2]);
    printf("],\n");

    constexpr int index_bits = 32 - Mask::block_bits;
    std::vector<uint64_t> sorter(np), base_offset(np);
    std::vector<int> order(np);
    T one_over_dx = (T)1 / dx;
    for (int i = 0; i < np; ++i) {
        std::array<int, 3> base;
        for (int d = 0; d < 3; ++d) base[d] = int_floor(X[i][d] * one_over_dx - (T)0.5);
        uint64_t offset = Mask::Linear_Offset(base);
        sorter[i] = ((offset >> Mask::data_bits) << index_bits) + i;
    }
    std::sort(sorter.begin(), sorter.end());
    std::vector<std::pair<int, int>> groups;
    std::vector<uint64_t> block_offset;
    int last_index = 0;
    for (int i = 0; i < np; ++i)
        if (i == np - 1 || (sorter[i] >> 32) != (sorter[i + 1] >> 32)) {
            groups.push_back(std::make_pair(last_index, i));
            block_offset.push_back(sorter[i] >> 32);
            last_index = i + 1;
        }
    page_map.Clear();
    for (int i = 0; i < np; ++i) {
        order[i] = (int)(sorter[i] & ((1ll << index_bits) - 1));
        uint64_t offset = (sorter[i] >> index_bits) << Mask::data_bits;
        base_offset[order[i]] = offset;
        if (i == np - 1 || (sorter[i] >> 32) != (sorter[i + 1] >> 32)) {
            page_map.Set_Page(offset);
            auto x = 1 << Mask::block_xbits;
            auto y = 1 << Mask::block_ybits;
            auto z = 1 << Mask::block_zbits;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int c = 0; c < 2; ++c)
                        page_map.Set_Page(Mask::Packed_Add(offset, Mask::Linear_Offset(x * a, y * b, z * c)));
        }
    }
    page_map.Update_Block_Offsets();
    auto blocks = page_map.Get_Blocks();

    printf("\"particle_order\": [");
    for (int i = 0; i < np; ++i) printf("%s%d", i ? "," : "", order[i]);
    printf("],\n\"particle_base_offset\": [");
    for (int i = 0; i < np; ++i) printf("%s%llu", i ? "," : "", (unsigned long long)base_offset[i]);
    printf("],\n\"particle_group\": [");
    for (size_t g = 0; g < groups.size(); ++g) printf("%s[%d,%d]", g ? "," : "", groups[g].first, groups[g].second);
    printf("],\n\"block_offset\": [");
    for (size_t g = 0; g < block_offset.size(); ++g) printf("%s%llu", g ? "," : "", (unsigned long long)block_offset[g]);
    printf("],\n\"blocks\": [");
    for (unsigned b = 0; b < blocks.second; ++b) printf("%s%llu", b ? "," : "", (unsigned long long)blocks.first[b]);
    printf("],\n");

    // ---- 3. node numbering: mark every node of every particle's 3x3x3 kernel through the reference's
    //         own virtual-memory array (one byte flag in the struct), then number like getNumNodes.
    auto arr = alloc.Get_Array();
    for (unsigned b = 0; b < blocks.second; ++b) std::memset(&arr(blocks.first[b]), 0, 4096);
    for (int p = 0; p < np; ++p)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    uint64_t o = Mask::Packed_Add(base_offset[p], Mask::Linear_Offset(i, j, k));
                    arr(o).pad[0] = 1;
                }
    int counter = 0;
    std::vector<std::array<int, 3>> id2coord;
    std::vector<unsigned long long> id2offset;
    for (unsigned b = 0; b < blocks.second; ++b) {
        Node<BYTES>* g = &arr(blocks.first[b]);
        for (int e = 0; e < (int)Mask::elements_per_block; ++e)
            if (g[e].pad[0]) {
                uint64_t o = blocks.first[b] + ((uint64_t)e << Mask::data_bits);
                id2offset.push_back(o);
                id2coord.push_back(Mask::LinearToCoord(o));
                ++counter;
            }
    }
    printf("\"num_nodes\": %d,\n\"id2coord\": [", counter);
    for (int i = 0; i < counter; ++i) printf("%s[%d,%d,%d]", i ? "," : "", id2coord[i][0], id2coord[i][1], id2coord[i][2]);
    printf("],\n\"id2offset\": [");
    for (int i = 0; i < counter; ++i) printf("%s%llu", i ? "," : "", id2offset[i]);
    printf("]\n}\n");
}


// ---- tie points: particles whose index-space coordinate X / dx - 0.5 lies within a few ulp of an integer, where the base node — and with it the
// SPGrid sort key — depends on how the product X * (1 / dx) is rounded before 0.5 is subtracted.  The reference evaluates
//     baseNode<2>(Xarray[i] * one_over_dx)        (Lib/MPM/MpmSimulationBase.cpp:1080-1084, Lib/Ziran/Math/Splines/BSplines.h:16-20)
// i.e. the product goes into an Eigen temporary and int_floor(x(d) - 0.5) is taken component by component; whether the subtraction is fused
// with the multiply is the compiler's choice.  Emitted per point and axis-0 coordinate: the base node with the product ROUNDED first (kept from
// fusing by a volatile store), with ONE rounding (std::fma), and `as_compiled`: what the statement in the reference's shape — the three products
// stored in an array, then int_floor(t[d] - 0.5) in another function — yields under the flags THIS binary was built with.  oracle/Makefile
// builds the driver twice: -O2 (no FMA instructions: x86-64 baseline) and -O3 -march=native -fno-math-errno (the reference's Release flags,
// CMakeLists.txt:28, on a machine with FMA); tests/golden/spgrid_tie_*.json hold both.
template <class T>
struct Vec3 {
    T v[3];
    T operator()(int d) const { return v[d]; }
};
template <class T>
static inline int base_node_shape(const T& x) { return int_floor(x - (T)0.5 * (2 - 1)); } // BSplines.h:16-20 with interpolation_degree = 2
template <class T>
static inline Vec3<T> scaled(const std::array<T, 3>& X, T c) // stands for the Eigen temporary of Xarray[i] * one_over_dx
{
    Vec3<T> t;
    for (int d = 0; d < 3; ++d) t.v[d] = X[d] * c;
    return t;
}
template <class T>
static inline std::array<int, 3> base_as_compiled(const std::array<T, 3>& X, T c)
{
    const Vec3<T>& x = scaled(X, c);
    std::array<int, 3> b;
    for (int d = 0; d < 3; ++d) b[d] = base_node_shape<T>(x(d));
    return b;
}
template <class T, int BYTES>
void run_tie()
{
    using Alloc = SPGrid_Allocator<Node<BYTES>, 3, 12>;
    using Mask = typename Alloc::template Array_type<>::MASK;
    // candidates: the floating-point numbers next to (N + 0.5) dx for cell faces N + 0.5 around 500 and 3800 and around the powers of two 256,
    // 512, 1024, 2048, for several grid spacings; kept: every candidate at which the two roundings give different base nodes and, for contrast,
    // every 16th other.  RN(X c) - 0.5 and RN(X c - 0.5) are the same number whenever N + 0.5 and N share a binade (rounding commutes with
    // subtracting a multiple of the spacing), so the two roundings can only disagree where N is a power of two: just below N the spacing of
    // the floating-point numbers is half that of N + 0.5.
    std::vector<std::array<T, 3>> X;
    std::vector<T> DX;
    int others = 0;
    for (double dxd : { 0.01, 0.013, 0.007, 0.0093, 0.0117, 0.0101, 0.0087, 0.0123 }) {
        const T dx = (T)dxd, c = (T)1 / dx;
        auto rounded_base = [&](T x) {
            volatile T t = x * c;
            return int_floor((T)t - (T)0.5);
        };
        for (int lo : { 498, 3798, 254, 510, 1022, 2046 })
            for (int N = lo; N < lo + 4; ++N) {
                T x0 = (T)(((double)N + 0.5) * (double)dx);
                for (int j = -2; j <= 2; ++j) {
                    T x = x0;
                    for (int q = 0; q < (j < 0 ? -j : j); ++q) x = std::nextafter(x, j < 0 ? (T)0 : (T)1e30);
                    const bool differ = rounded_base(x) != int_floor(std::fma(x, c, -(T)0.5));
                    if (differ || (others++ % 16) == 0) X.push_back({ x, (T)(503.3 * dxd), (T)(499.7 * dxd) }), DX.push_back(dx);
                }
            }
    }
    printf("{\n\"struct_bytes\": %d,\n\"dx\": [", BYTES);
    for (size_t p = 0; p < X.size(); ++p) printf("%s%.17g", p ? "," : "", (double)DX[p]);
    printf("],\n\"X\": [");
    for (size_t p = 0; p < X.size(); ++p) printf("%s[%.17g,%.17g,%.17g]", p ? "," : "", (double)X[p][0], (double)X[p][1], (double)X[p][2]);
    const char* names[3] = { "base_rounded_product", "base_fma", "base_as_compiled" };
    for (int mode = 0; mode < 3; ++mode) {
        printf("],\n\"%s\": [", names[mode]);
        for (size_t p = 0; p < X.size(); ++p) {
            const T c = (T)1 / DX[p];
            std::array<int, 3> b;
            if (mode == 2)
                b = base_as_compiled<T>(X[p], c);
            else
                for (int d = 0; d < 3; ++d) {
                    if (mode == 0) {
                        volatile T t = X[p][d] * c; // the rounded product, whatever the flags
                        b[d] = int_floor((T)t - (T)0.5);
                    }
                    else
                        b[d] = int_floor(std::fma(X[p][d], c, -(T)0.5));
                }
            printf("%s[%d,%d,%d,%llu]", p ? "," : "", b[0], b[1], b[2], (unsigned long long)Mask::Linear_Offset(b));
        }
    }
    printf("]\n}\n");
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <float|double> <n_particles> <seed>   |   %s <float|double> tie 0\n", argv[0], argv[0]);
        return 2;
    }
    if (!strcmp(argv[2], "tie")) {
        if (!strcmp(argv[1], "float"))
            run_tie<float, 64>();
        else
            run_tie<double, 128>();
        return 0;
    }
    int np = atoi(argv[2]);
    uint64_t seed = strtoull(argv[3], 0, 10);
    if (!strcmp(argv[1], "float"))
        run<float, 64>(np, seed);
    else
        run<double, 128>(np, seed);
    return 0;
}
