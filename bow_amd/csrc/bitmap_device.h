// bitmap_device.h — Arrow validity-bitmap helpers shared by the kernels of interp_fill.hip and interpolate.hip:
// bit tests and previous / next valid-row lookups (Bow.GetPrevRowIndex / GetNextRowIndex semantics,
// bowgetters.go:125-151), unbounded (word walks) and bounded through the block neighbour index (common.h NbrIndex).
#pragma once

#include "agg_device.h"

namespace bowgpu {

__device__ __forceinline__ bool bit_at(const uint32_t *bits, int64_t bit0, int64_t row) {
    if (!bits) return true;
    const int64_t b = bit0 + row;
    return (bits[b >> 5] >> (b & 31)) & 1u;
}

// previous / next valid row of a column (Bow.GetPrevFloat64 / GetNextFloat64 index semantics,
// bowgetters.go:252-277), skipping 32 rows at a time over all-null words
__device__ inline int64_t prev_valid(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    while (b >= bit0) {
        const int64_t w = b >> 5;
        const int sh = (int)(b & 31);
        uint32_t x = bits[w];
        x = sh == 31 ? x : (x & ((2u << sh) - 1u));  // bits <= sh
        if (w == (bit0 >> 5)) x &= ~0u << (bit0 & 31);  // not before the column's first bit
        if (x) return (w << 5) + (31 - __clz((int)x)) - bit0;
        b = (w << 5) - 1;
    }
    return -1;
}
__device__ inline int64_t next_valid(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    const int64_t bend = bit0 + n;
    while (b < bend) {
        const int64_t w = b >> 5;
        uint32_t x = bits[w] & (~0u << (b & 31));
        if (x) {
            const int64_t r = (w << 5) + (__ffs((int)x) - 1) - bit0;
            return r < n ? r : -1;
        }
        b = (w + 1) << 5;
    }
    return -1;
}


// previous / next valid row at or before / after `row`, looking at most one block of words, then the index
__device__ __forceinline__ int64_t prev_valid_ix(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row, const NbrIndex &ix) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    const int64_t g = b / kNbrBlockBits;
    const int64_t stop = g * kNbrBlockBits > bit0 ? g * kNbrBlockBits : bit0;  // first bit of the block that belongs to the column
    while (b >= stop) {
        const int64_t w = b >> 5;
        const int sh = (int)(b & 31);
        uint32_t x = bits[w];
        x = sh == 31 ? x : (x & ((2u << sh) - 1u));
        if ((w << 5) < stop) x &= ~0u << (stop - (w << 5));
        if (x) return (w << 5) + (31 - __clz((int)x)) - bit0;
        b = (w << 5) - 1;
    }
    return ix.prev_before[g - ix.g0];
}
__device__ __forceinline__ int64_t next_valid_ix(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row, const NbrIndex &ix) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    const int64_t g = b / kNbrBlockBits;
    const int64_t bend = (g + 1) * kNbrBlockBits < bit0 + n ? (g + 1) * kNbrBlockBits : bit0 + n;
    while (b < bend) {
        const int64_t w = b >> 5;
        uint32_t x = bits[w] & (~0u << (b & 31));
        if (((w + 1) << 5) > bend) x &= (1u << (bend - (w << 5))) - 1u;
        if (x) return (w << 5) + (__ffs((int)x) - 1) - bit0;
        b = (w + 1) << 5;
    }
    return ix.next_after[g - ix.g0];
}

}  // namespace bowgpu
