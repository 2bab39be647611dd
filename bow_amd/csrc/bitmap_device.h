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

// ... and WITHOUT an index: at most kNbrNearWords words are looked at (2048 rows); *far is set when they ran out before a valid row
// or the column's first / last row was found - the caller then raises a status word and the host repeats the call with the index
// built (three small launches that the usual column - no run of thousands of nulls next to a window start - never needs).
constexpr int kNbrNearWords = 64;
__device__ inline int64_t prev_valid_near(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row, bool *far) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    for (int k = 0; k < kNbrNearWords && b >= bit0; k++) {
        const int64_t w = b >> 5;
        const int sh = (int)(b & 31);
        uint32_t x = bits[w];
        x = sh == 31 ? x : (x & ((2u << sh) - 1u));
        if (w == (bit0 >> 5)) x &= ~0u << (bit0 & 31);
        if (x) return (w << 5) + (31 - __clz((int)x)) - bit0;
        b = (w << 5) - 1;
    }
    if (b >= bit0) *far = true;
    return -1;
}
__device__ inline int64_t next_valid_near(const uint32_t *bits, int64_t bit0, int64_t n, int64_t row, bool *far) {
    if (row < 0 || row >= n) return -1;
    if (!bits) return row;
    int64_t b = bit0 + row;
    const int64_t bend = bit0 + n;
    for (int k = 0; k < kNbrNearWords && b < bend; k++) {
        const int64_t w = b >> 5;
        const uint32_t x = bits[w] & (~0u << (b & 31));
        if (x) {
            const int64_t r = (w << 5) + (__ffs((int)x) - 1) - bit0;
            return r < n ? r : -1;
        }
        b = (w + 1) << 5;
    }
    if (b < bend) *far = true;
    return -1;
}

// ---------------------------------------------------------------- wave-uniform bitmap access (fills, IsColSorted, Interpolate)
// 128 validity bits of rows [row0, row0 + 128) in row order (w0 = rows 0..63); rows at or beyond n read as 0 (kFull: the
// caller knows row0 + 128 <= n).  Every lane passes the same arguments and the bitmap is an input nobody writes during the
// kernel, so it is read through the constant address space: scalar loads, scalar arithmetic.
template <bool kFull>
__device__ __forceinline__ void load_bits128(const uint32_t *vbits, int64_t vbit0, int64_t row0, int64_t n, uint64_t *w0, uint64_t *w1) {
    uint64_t x0 = ~0ull, x1 = ~0ull;
    const int64_t left = n - row0;  // >= 1
    if (vbits) {
        typedef const uint32_t __attribute__((address_space(4))) *const_words;
        const_words q = (const_words)(uintptr_t)vbits;
        const int64_t bit = vbit0 + row0;
        const int64_t wi = bit >> 5;
        const int sh = (int)(bit & 31);
        uint32_t d0, d1, d2, d3, d4;
        if (kFull) {
            d0 = q[wi]; d1 = q[wi + 1]; d2 = q[wi + 2]; d3 = q[wi + 3];
            d4 = sh ? q[wi + 4] : 0u;  // (the chunk ends inside word wi + 3 when it starts on a word boundary)
        } else {
            const int64_t wl = (vbit0 + (left < 128 ? n : row0 + 128) - 1) >> 5;  // last word that holds a row of the chunk
            d0 = q[wi];
            d1 = wi + 1 <= wl ? q[wi + 1] : 0u; d2 = wi + 2 <= wl ? q[wi + 2] : 0u;
            d3 = wi + 3 <= wl ? q[wi + 3] : 0u; d4 = wi + 4 <= wl ? q[wi + 4] : 0u;
        }
        const uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32), mid = (uint64_t)d2 | ((uint64_t)d3 << 32);
        x0 = sh ? (lo >> sh) | (mid << (64 - sh)) : lo;
        x1 = sh ? (mid >> sh) | ((uint64_t)d4 << (64 - sh)) : mid;
    }
    if (!kFull && left < 128) {
        if (left <= 64) { x1 = 0; x0 = left == 64 ? x0 : (x0 & ((1ull << left) - 1ull)); }
        else x1 &= (1ull << (left - 64)) - 1ull;
    }
    *w0 = x0;
    *w1 = x1;
}

// bits of e on the even positions, bits of o on the odd ones
__device__ __forceinline__ uint64_t spread32(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__device__ __forceinline__ uint64_t interleave32(uint32_t e, uint32_t o) { return spread32(e) | (spread32(o) << 1); }

__device__ __forceinline__ uint64_t lane_value(uint64_t v, int src_lane) {  // src_lane is the same in every lane
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

}  // namespace bowgpu
