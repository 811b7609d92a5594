// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the SPGrid 3-D address arithmetic used by HOT's MPM grid:
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:21-57   (bit budget: page/data/block bits, page masks)
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:76-80   (element masks inside a 4 KiB page)
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:119-123 (aggregate x/y/z masks)
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:150-157 (Linear_Offset = spread x | spread y | spread z)
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:182-189 (LinearToCoord = Bit_Pack per axis)
//   reference Lib/SPGrid/Core/SPGrid_Mask.h:237-245 (Packed_Add)
//   reference Lib/SPGrid/Core/SPGrid_Utilities.h:80-343 / SPGrid_Utilities.cpp:43-62 (Bit_Spread / Bit_Pack,
//             i.e. software pdep / pext)
// MpmGrid instantiates SPGrid_Mask<log2(sizeof GridState), log2(sizeof GridState), 3, 12>
// (reference Lib/MPM/MpmGrid.h:108-115, SPGrid_Allocator.h:26-29): GridState<float,3> is 64 B
// (data_bits 6, 4x4x4 block), GridState<double,3> is 128 B (data_bits 7, 2x4x4 block).
//
// Pinned against the real reference code: oracle/_ref/spgrid_ref (built from /root/reference by
// oracle/Makefile) emits tests/golden/spgrid_index_*.json which tests/test_oracle_indexing.py replays.
#pragma once
#include <cstdint>
#include <array>

namespace hot_oracle {

template <int LOG2_STRUCT>
struct SpMask {
    static constexpr int page_bits = 12;
    static constexpr int data_bits = LOG2_STRUCT;
    static constexpr int block_bits = page_bits - data_bits;
    static constexpr int block_zbits = block_bits / 3 + (block_bits % 3 > 0);
    static constexpr int block_ybits = block_bits / 3 + (block_bits % 3 > 1);
    static constexpr int block_xbits = block_bits / 3;
    static constexpr int elements_per_block = 1 << block_bits;
    static constexpr int log2_field = LOG2_STRUCT; // Array_type<> default field == the whole struct

    static constexpr uint64_t hi = 0xffffffffffffffffULL << page_bits;
    static constexpr uint64_t page_zmask = (0x9249249249249249ULL << (3 - block_bits % 3)) & hi;
    static constexpr uint64_t page_ymask = (0x2492492492492492ULL << (3 - block_bits % 3)) & hi;
    static constexpr uint64_t page_xmask = (0x4924924924924924ULL << (3 - block_bits % 3)) & hi;
    static constexpr uint64_t element_zmask = (uint64_t)((1 << block_zbits) - 1) << log2_field;
    static constexpr uint64_t element_ymask = (uint64_t)((1 << block_ybits) - 1) << (log2_field + block_zbits);
    static constexpr uint64_t element_xmask = (uint64_t)((1 << block_xbits) - 1) << (log2_field + block_zbits + block_ybits);
    static constexpr uint64_t zmask = page_zmask | element_zmask;
    static constexpr uint64_t ymask = page_ymask | element_ymask;
    static constexpr uint64_t xmask = page_xmask | element_xmask;

    // software pdep: deposit the low bits of `data` into the set bits of `mask`, low to high
    static inline uint64_t bit_spread(uint64_t data, uint64_t mask)
    {
        uint64_t result = 0;
        for (uint64_t bit = 1; bit; bit <<= 1) {
            if (bit & mask) {
                if (data & 1) result |= bit;
                data >>= 1;
            }
        }
        return result;
    }
    // software pext
    static inline int bit_pack(uint64_t data, uint64_t mask)
    {
        uint64_t result = 0;
        int out = 0;
        for (uint64_t bit = 1; bit; bit <<= 1) {
            if (bit & mask) {
                if (data & bit) result |= (1ULL << out);
                ++out;
            }
        }
        return (int)(int64_t)result;
    }
    static inline uint64_t linear_offset(int i, int j, int k)
    {
        return bit_spread((uint64_t)(int64_t)i, xmask) | bit_spread((uint64_t)(int64_t)j, ymask) | bit_spread((uint64_t)(int64_t)k, zmask);
    }
    static inline std::array<int, 3> linear_to_coord(uint64_t o)
    {
        return { bit_pack(o, xmask), bit_pack(o, ymask), bit_pack(o, zmask) };
    }
    static inline uint64_t packed_add(uint64_t i, uint64_t j)
    {
        const uint64_t mx = ~xmask, my = ~ymask, mz = ~zmask, mw = xmask | ymask | zmask;
        uint64_t x_result = ((i | mx) + (j & ~mx)) & ~mx;
        uint64_t y_result = ((i | my) + (j & ~my)) & ~my;
        uint64_t z_result = ((i | mz) + (j & ~mz)) & ~mz;
        uint64_t w_result = ((i | mw) + (j & ~mw)) & ~mw;
        return x_result | y_result | z_result | w_result;
    }
};

} // namespace hot_oracle
