// rolling_tw.hip — the wave-tile kernel for the TIME-WEIGHTED reducers and INCLUSIVE windows: IntegralStep,
// IntegralTrapezoid, WeightedAverageStep, WeightedAverageLinear (reference rolling/aggregation/integral.go:8-69,
// weightedmean.go:8-34) next to WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows, with the
// windows built inclusive when a reducer asks for it (rolling/aggregation.go:183-185, :207-211; rolling.go:201-209).
// Same tile structure and limits as rolling_simple.hip: one wavefront per tile of 512 + 128 rows, 32-bit window ids, preset
// validity bitmaps, one lane per window.
//
// Round 4: the integrals are no longer computed BY the walk.  A term of integral.go:28 / :54 - (v0 + v1) / 2 * (t1 - t0), v0 * (t1 - t0)
// - depends on a row and on the valid point before it, not on the running sum, so every lane computes the terms of ITS ten rows
// (all 64 lanes busy, whatever the window length), the terms are staged in LDS in row order, and the lane that owns a window only
// adds them up, in row order: one LDS read and one addition per row, the same chain of additions integral.go performs, bit for
// bit.  (The round-1 walk did the conversions, the subtraction, the products and the first-point test per row on a chain that only
// 512 / w lanes walked: 14 dependent instructions per row - 0.33 of the HBM peak at 64-row windows, 0.19 with nulls.)
//   * a row that is not its window's first valid point holds its term against the previous valid point of the SAME window;
//   * a row that is the first valid point of its window holds +0.0 (adding +0.0 to a sum that started at +0.0 changes no bit);
//   * the slot of a window's FIRST ROW (a head: valid or not, its own term is never added) holds what CLOSES the window before
//     it: for the step integral the value v0 of that window's last valid point - its lane adds v0 * (float64(LastValue) - t0)
//     (integral.go:49-55) - and for the trapezoid the term that joins that point to the head row, added only when the head row is
//     the inclusive row of the window before (rolling.go:201-209);
//   * null rows hold +0.0.
// The previous valid point of a row: the row before it (a DPP move) in a column without nulls; with nulls the nearest set bit
// below it in the tile's validity words (count-leading-zeros), its value and time gathered from LDS.
// One LDS array serves every phase of a column in turn - the staged values (Sum / Mean / Min / Max / First / Last: the walk of
// rolling_simple.hip), then the step terms, then the trapezoid terms, written in place - next to the rows' times, which every term
// pass reads: nothing of a tile stays in registers between the phases.
// What this kernel does not take goes to rolling_agg.hip (64-bit window ids, tiles denser than the head list).
#include <type_traits>

#include <stddef.h>

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kWave = 64;
constexpr int kTileT = 512;
constexpr int kHaloT = 128;
constexpr int kRowsT = kTileT + kHaloT;
constexpr int kChunksT = kRowsT / 128;
// At most TwCap windows may start inside one tile (+ look-ahead); denser tiles send the call to the general kernel.  The head list holds
// 16-bit entries (row | on-window-start flag << 15; the window id of a head is recomputed from its staged time, so ids of any
// distance inside one tile stay on this kernel).  The kernel's rate follows its occupancy and that follows the LDS per wavefront
// (1 KB steps), so the list is sized to the byte: 32-bit times 8 KB (20 wavefronts per CU): 240 heads = windows of >= 2.7 rows, 208
// with nulls; 64-bit times 11 KB (14 per CU): 464 heads
constexpr int64_t kShortAvgRows = 14;   // calls whose windows average fewer rows: rolling_tw_kernel<.., kShort = true>

constexpr int kLeanCap = 170;           // heads of the lean form (6 KB with the padded term array): windows of >= 3.8 rows; the host sends averages below 5 to the other forms
template <bool kNulls, bool kTs32>
struct TwCap { static constexpr int value = kTs32 ? (kNulls ? 208 : 240) : 464; };

// kTs32: float64(ts) is rebuilt as float64(s0 of slot 0) + float64(offset), which is exact - and so equal to the reference's single
// conversion (integral.go:17) - when every |ts| of the call is below 2^53 (the host checks; nanosecond epochs take the 64-bit form).
// kLean: the commonest call - ONE value column without nulls, integrals only - keeps nothing but the term array and the head list in
// LDS (6 KB: 26 wavefronts per CU, the simple kernel's occupancy): no staged times - a head entry carries its window id (32-bit
// entries: row | on-window-start flag << 10 | (wid - wid of the tile's first row) << 11; ids more than 2^21 apart inside one tile
// send the call to the general kernel) and the slot of a head holds the closing PRODUCT v0 * (float64(LastValue) - t0) of the window
// before it, computed by the head's lane in the flag pass, where both are in registers.
template <bool kNulls, bool kTs32, bool kLean>
struct TwShared {
    uint64_t val[kLean ? swz_slots(kRowsT) : kRowsT];   // the column's staged values; then its step terms; then its trapezoid terms (kLean: padded - agg_device.h swz)
    // the time of every row, where the term pass reads a row's own and its previous point's: a 32-bit offset from slot 0 (kTs32) or the int64 itself
    typename std::conditional<kTs32, uint32_t, int64_t>::type tsx[kLean ? 1 : kRowsT];
    uint32_t vbits[kNulls ? kRowsT / 32 + 2 : 2];  // validity words of the value column for this tile
    // heads in row order: local row | on-window-start flag << 15 (kLean: see above)
    typename std::conditional<kLean, uint32_t, uint16_t>::type seg[(kLean ? kLeanCap : TwCap<kNulls, kTs32>::value) + 2];
};
static_assert(sizeof(TwShared<false, true, false>) <= 8192 && sizeof(TwShared<true, true, false>) <= 8192, "LDS of the 32-bit forms: 8 KB");
static_assert(sizeof(TwShared<false, false, false>) <= 11264 && sizeof(TwShared<true, false, false>) <= 11264, "LDS of the 64-bit forms: 11 KB");
static_assert(sizeof(TwShared<false, true, true>) <= 6144 && sizeof(TwShared<false, false, true>) <= 6144, "LDS of the lean forms: 6 KB");

__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// the value lane - 1 holds (lane 0: `lane0`)
__device__ __forceinline__ double left64(double x, double lane0) {
    const uint64_t b = (uint64_t)__double_as_longlong(x), f = (uint64_t)__double_as_longlong(lane0);
    const uint32_t lo = left32((uint32_t)b, (uint32_t)f), hi = left32((uint32_t)(b >> 32), (uint32_t)(f >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double lane63(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// non-temporal 16-byte loads / 8-byte stores: see rolling_simple.hip (rows 128..511 of a tile are read by this wavefront only;
// outputs are never read back)
typedef unsigned long long u64x2_tw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 load16_nt(const ulonglong2 *q) {
    const u64x2_tw v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_tw *>(q));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// kNulls: some column has nulls; kWide: see rolling_simple.hip (window ids relative to the tile's first window: rows may span more
// than 2^32 from slot 0); kTs32: times as 32-bit offsets from slot 0 (above)
// kBoth: the call has step AND trapezoid integrals: the terms of the second kind wait in registers while the first kind is walked
// kShort: a call whose windows average fewer than kShortAvgRows rows - every tile takes the one-walk form (walk_all below), and the
// term machinery is not even compiled in: its registers cost a wavefront per SIMD, which at that window length is what sets the rate
// kMulti: more than one value column - a real loop over the column passes.  One column (kMulti = false): straight-line code, and
// what LLVM would compute in front of that loop and park, out of scalar registers, in vector lanes is computed where it is used
// (round 5, found on interp_wave3_kernel: 75 - 90 parked scalars per wavefront here, 31 - 45 without the loop, 20 - 30 fewer vector
// registers).
template <bool kNulls, bool kWide, bool kTs32, bool kBoth, bool kShort, bool kLean, bool kMulti>
__global__ __launch_bounds__(kWave, (kShort || kLean || (!kNulls && kTs32 && (!kBoth || !kMulti))) ? 5 : 4) void rolling_tw_kernel(const SimpleParams p, const int64_t ntiles, const int64_t tiles_per_xcd) {
    static_assert(!(kWide && kTs32), "the wide form keeps 64-bit timestamps");
    static_assert(!kLean || (!kNulls && !kShort), "the lean form: one column without nulls, integrals only, terms from the flag pass");
    __shared__ TwShared<kNulls, kTs32, kLean> sh;
    constexpr int kSegCapT = kLean ? kLeanCap : TwCap<kNulls, kTs32>::value;
    // the padded layout of the term array (agg_device.h swz) where two walks per tile make LDS the busiest unit: both kinds of integral in the
    // lean form.  (One kind: the three single steps in front of a window's first aligned group cost more than the bank conflicts did - same-box
    // A/B at 1e8 rows, 16 .. 96 rows per window: +2 .. 4 % kernel time with the pads.)
#ifndef BOWGPU_TW_LEAN_PAD
#define BOWGPU_TW_LEAN_PAD 0   // A/B build (scratch/build_variant.sh twpad rolling_tw.hip -DBOWGPU_TW_LEAN_PAD=1): the pads for ONE kind of integral too - profiles/r06_stdout_tw_lean_pad_ab.txt
#endif
    constexpr bool kSwzT = kLean && (kBoth || BOWGPU_TW_LEAN_PAD != 0);
    constexpr uint32_t kRowMask = kLean ? 0x3FFu : 0x7FFFu, kStartBit = kLean ? 0x400u : 0x8000u;
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs (look-ahead rows hit the same L2)
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int64_t base = tile * kTileT;
    const int64_t n = p.n;
    const bool interior = base + kRowsT <= n;
    const int nloc = interior ? kRowsT : (int)(n - base);

    // ---- loads: ts, then the first value column right behind it (and its validity words: one memory round trip for all)
    uint64_t ta[kChunksT], tb[kChunksT], va[kChunksT], vb[kChunksT];
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    // a column whose first row is only 8-byte aligned (an Arrow slice with an odd offset) is read with two 8-byte loads per lane and
    // chunk instead of one 16-byte load: slower through the L1, but the call stays on this kernel
    auto load_col = [&](const uint64_t *__restrict__ src, uint64_t (&a)[kChunksT], uint64_t (&bb)[kChunksT], bool aligned) {
        if (interior && aligned) {
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(src + base) + lane;
#pragma unroll
            for (int j = 0; j < kChunksT; j++) {
                const ulonglong2 x = (j > 0 && j < kChunksT - 1) ? load16_nt(q + j * 64) : q[j * 64];
                a[j] = x.x; bb[j] = x.y;
            }
        } else if (interior) {
            const uint64_t *q = src + base + 2 * lane;
#pragma unroll
            for (int j = 0; j < kChunksT; j++) { a[j] = q[j * 128]; bb[j] = q[j * 128 + 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksT; j++) load_pair(src, base + j * 128 + 2 * lane, n, aligned, a[j], bb[j]);
        }
    };
    // validity words of one column's 640 rows, 32 per lane (lanes 0..19), any bit offset (Arrow slices)
    auto load_vword = [&](int c) -> uint32_t {
        uint32_t word = 0xFFFFFFFFu;
        if (lane < kRowsT / 32 && p.vbits[c] != nullptr) {
            const int64_t bit = p.vbit0[c] + base + 32 * (int64_t)lane;
            const int64_t wi = bit >> 5;
            const int shb = (int)(bit & 31);
            const uint32_t lo = wi < p.vwords[c] ? p.vbits[c][wi] : 0u;
            const uint32_t hi = (shb != 0 && wi + 1 < p.vwords[c]) ? p.vbits[c][wi + 1] : 0u;
            word = shb ? ((lo >> shb) | (hi << (32 - shb))) : lo;
        }
        return word;
    };
    // The usual tile - interior, both columns 16-byte aligned - issues its ten loads back to back in ONE block, and nothing in
    // front of them (rolling_simple.hip: two load_col() calls put a wait for all outstanding loads between the columns)
    if (interior && !(p.unaligned_mask & 0x80000001u)) {
        const ulonglong2 *qt = reinterpret_cast<const ulonglong2 *>(ts + base) + lane;
        const ulonglong2 *qv = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const uint64_t *>(p.values[0]) + base) + lane;
#pragma unroll
        for (int j = 0; j < kChunksT; j++) {
            const ulonglong2 x = (j > 0 && j < kChunksT - 1) ? load16_nt(qt + j * 64) : qt[j * 64];
            ta[j] = x.x; tb[j] = x.y;
        }
#pragma unroll
        for (int j = 0; j < kChunksT; j++) {
            const ulonglong2 x = (j > 0 && j < kChunksT - 1) ? load16_nt(qv + j * 64) : qv[j * 64];
            va[j] = x.x; vb[j] = x.y;
        }
    } else {
        load_col(ts, ta, tb, !(p.unaligned_mask >> 31));
        load_col(reinterpret_cast<const uint64_t *>(p.values[0]), va, vb, !(p.unaligned_mask & 1u));
    }
    uint32_t vword = kNulls ? load_vword(0) : 0u;
    // the row left of the tile (scalar load): first head flag + order check
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    // ids are 32-bit and relative to window w0, which starts at ws0: slot 0 of the call, or (kWide) the tile's first window
    uint64_t w0 = 0;
    int64_t ws0 = p.s0;
    bool unsorted = false, sat = false;  // rows out of order ; ids the 16-bit local fields / 32-bit arithmetic cannot hold
    const int64_t ts_first = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta[0] >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta[0]));
    // p.pre_rows: Go's truncating division put s0 above a negative first timestamp (rolling.go:96-99); the rows below s0 ride in
    // window 0 (rolling.go:194-196).  They are the frame's first rows, fewer than one interval's worth of time: ids below are
    // forced to 0 for them (wave-uniform test per tile: only tiles that start below s0 pay for the per-row comparison)
    const bool pre = p.pre_rows && ts_first < p.s0;
    if (kWide) {
        w0 = pre ? 0ull : magic_div((uint64_t)ts_first - (uint64_t)p.s0, p.magic);
        ws0 = p.s0 + (int64_t)(w0 * (uint64_t)p.interval);
        const int64_t ts_last = p.ts[base + nloc - 1];
        // ids come from (ts - ws0) >> k with k = trailing zero bits of the interval (floor(a / b) == floor((a >> k) / (b >> k)) when
        // 2^k divides b): the tile's rows must lie within 2^(32+k) of ws0 - 2199 s for 1 s windows of nanosecond timestamps
        sat = ts_last < ts_first || (((uint64_t)ts_last - (uint64_t)ws0) >> p.shift_k) >= 0xFFFFFFF0ull;  // (unsorted rows are caught below too)
    }
    const uint32_t s0_lo = (uint32_t)ws0;
    auto rel32 = [&](int64_t t) -> uint32_t {  // timestamp -> 32-bit numerator of the window id
        return kWide ? (uint32_t)(((uint64_t)t - (uint64_t)ws0) >> p.shift_k) : (uint32_t)t - s0_lo;
    };
    auto wid_of = [&](int64_t t) -> uint32_t { return (pre && t < ws0) ? 0u : mdiv32(rel32(t), p.m32, p.sh1, p.sh2); };

    // ---- window ids (32-bit), head flags, compaction with a running scalar count
    const uint32_t w_first = (kWide || pre) ? 0u : mdiv32((uint32_t)ts_first - s0_lo, p.m32, p.sh1, p.sh2);   // (kLean: head entries hold ids relative to it)
    const uint32_t left_w0 = base == 0 ? 0xFFFFFFFEu : (pre && left0 < ws0) ? 0u : (kWide && left0 < ws0) ? 0xFFFFFFFEu : mdiv32(rel32(left0), p.m32, p.sh1, p.sh2);
    uint32_t left_w = left_w0;
    int64_t left_ts = left0;
    typedef typename std::conditional<kTs32, uint32_t, uint64_t>::type tkey_t;   // a row's time as the term pass reads it back from LDS
    uint32_t hmask = 0;   // bit 2j: row a of chunk j starts a window, bit 2j + 1: row b
    // which running statistics the outputs of this call read (set by the host, wave-uniform): only those are staged, computed and walked
    const bool need_step = p.need & kNeedStep, need_trap = p.need & kNeedTrap, need_mm = p.need & kNeedMinMax, need_sum = p.need & kNeedSum,
               need_fl = p.need & kNeedFirstLast;
    const bool need_vals = need_mm || need_sum || need_fl;
    // The usual call - integrals only (next to WindowStart / Count / NumRows), first column without nulls: the terms of column 0 are
    // computed right here in the flag pass, where a row's timestamp, value, head flag and left neighbour are all in registers, and go
    // straight to LDS: no staging, no term pass, no value phase.
    const bool early_terms = !kShort && !kNulls && !need_vals;
    if (kLean && (p.ncols != 1 || need_vals)) {   // (the host launches the lean form for such calls only)
        if (lane == 0) atomicOr(&p.status[4], 1u);
        return;
    }
    constexpr bool kKeep = kBoth && !kShort;
    double keep_a[kKeep ? kChunksT : 1], keep_b[kKeep ? kChunksT : 1];   // kBoth: the trapezoid terms wait here while the step terms are walked
    double early_carry_x = 0.0;
    const bool cint0 = p.col_is_int[0] != 0;
    int nseg_total = 0, nseg_owned = 0;
#pragma unroll
    for (int j = 0; j < kChunksT; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
        const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
        const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
        unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
        const uint32_t ra = rel32(tsa), rb = rel32(tsb);
        uint32_t wa = mdiv32(ra, p.m32, p.sh1, p.sh2);
        uint32_t wb = mdiv32(rb, p.m32, p.sh1, p.sh2);
        if (pre) { if (tsa < ws0) wa = 0u; if (tsb < ws0) wb = 0u; }
        const uint32_t wprev = left32(wb, left_w);
        const bool ha = pa && (wa != wprev);
        const bool hb = pb && (wb != wa);
        hmask |= (ha ? 1u : 0u) << (2 * j) | (hb ? 2u : 0u) << (2 * j);
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        // bit 15: the row sits exactly on its window's start (what makes it the inclusive row of the window before)
        const uint32_t ik = (uint32_t)((uint64_t)p.interval >> (kWide ? p.shift_k : 0));
        const uint64_t lowmask = kWide ? ((1ull << p.shift_k) - 1ull) : 0ull;
        uint32_t sa = 0u, sb = 0u;
        if (p.inclusive) {   // (uniform: only inclusive calls look at the flag)
            sa = (ra == wa * ik && (((uint64_t)tsa - (uint64_t)ws0) & lowmask) == 0) ? kStartBit : 0u;
            sb = (rb == wb * ik && (((uint64_t)tsb - (uint64_t)ws0) & lowmask) == 0) ? kStartBit : 0u;
        }
        if (kLean) {
            const uint32_t la = wa - w_first, lb = wb - w_first;
            sat |= (ha && la >= (1u << 21)) || (hb && lb >= (1u << 21));
            if (ha && pos < kSegCapT) sh.seg[pos] = (uint32_t)l | sa | (la << 11);
            pos += ha ? 1 : 0;
            if (hb && pos < kSegCapT) sh.seg[pos] = (uint32_t)(l + 1) | sb | (lb << 11);
        } else {
            if (ha && pos < kSegCapT) sh.seg[pos] = (uint16_t)((uint32_t)l | sa);
            pos += ha ? 1 : 0;
            if (hb && pos < kSegCapT) sh.seg[pos] = (uint16_t)((uint32_t)(l + 1) | sb);
        }
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kChunksT - 2) nseg_owned = nseg_total;
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
        // the rows' times, for the term passes (float64(ts) and the window id are both recomputed from them there)
        if (!kLean) {
            if (kTs32) *reinterpret_cast<uint2 *>(&sh.tsx[l]) = make_uint2(ra, rb);   // (l is even: one 8-byte LDS write)
            else *reinterpret_cast<ulonglong2 *>(&sh.tsx[l]) = make_ulonglong2(ta[j], tb[j]);
        }
        if (early_terms) {   // (see the term pass below for what a slot holds)
            const double xa = cint0 ? (double)(int64_t)va[j] : __longlong_as_double((long long)va[j]);
            const double xb = cint0 ? (double)(int64_t)vb[j] : __longlong_as_double((long long)vb[j]);
            const double xp = left64(xb, early_carry_x);
            early_carry_x = lane63(xb);
            // float64(t1) - float64(t0): the 32-bit difference converted (exact: both times are, and they are less than 2^32 apart), else as written
            const double dta = kTs32 ? (double)(uint32_t)((uint32_t)tsa - (uint32_t)prev_ts) : (double)tsa - (double)prev_ts;
            const double dtb = kTs32 ? (double)(uint32_t)((uint32_t)tsb - (uint32_t)tsa) : (double)tsb - (double)tsa;
            double s1 = 0.0, s2 = 0.0, q1 = 0.0, q2 = 0.0;
            if (need_step) {
                if (kLean) {
                    // a head's slot: v0 * (float64(LastValue) - t0) of the window before (integral.go:49-55).  32-bit times: the
                    // difference as an integer, converted - it is below the interval, and equal to the difference of the two
                    // float64 values, which are exact themselves (every |ts| < 2^53); else the two conversions as written
                    const double ca = kTs32 ? (double)(uint32_t)((wprev + 1u) * (uint32_t)p.interval - ((uint32_t)prev_ts - s0_lo))
                                            : (double)(ws0 + (int64_t)(((uint64_t)wprev + 1ull) * (uint64_t)(uint32_t)p.interval)) - (double)prev_ts;
                    const double cb = kTs32 ? (double)(uint32_t)((wa + 1u) * (uint32_t)p.interval - ra)
                                            : (double)(ws0 + (int64_t)(((uint64_t)wa + 1ull) * (uint64_t)(uint32_t)p.interval)) - (double)tsa;
                    s1 = ha ? xp * ca : xp * dta; s2 = hb ? xa * cb : xa * dtb;
                } else { s1 = ha ? xp : xp * dta; s2 = hb ? xa : xa * dtb; }
            }
            if (need_trap) { q1 = (xp + xa) / 2 * dta; q2 = (xa + xb) / 2 * dtb; }
            const double oa = need_step ? s1 : q1, ob = need_step ? s2 : q2;
            *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzT>(l)]) = make_ulonglong2((uint64_t)__double_as_longlong(oa), (uint64_t)__double_as_longlong(ob));
            if (kKeep) { keep_a[kKeep ? j : 0] = q1; keep_b[kKeep ? j : 0] = q2; }
        }
    }
    if (__ballot(unsorted)) {  // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0) atomicOr(&p.status[0], 1u);
        return;
    }
    if (nseg_total > kSegCapT) sat = true;
    if (__ballot(sat)) {  // a tile the segment list / the 32-bit arithmetic cannot describe: the host redoes the call with the general kernel
        // (one atomic per call, not one per tile: 2e5 atomics on one address took 2 ms)
        if (lane == 0 && !__hip_atomic_load(&p.status[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[4], 1u);
        return;
    }

    const bool reaches_end = base + kRowsT >= n;
    // windows of the call, as an id relative to w0 (the last tile's successor id when the data ends in it)
    const uint64_t Wrel = (uint64_t)p.W - w0;
    const uint32_t W32 = Wrel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)Wrel;
    const double ws0_d = (double)ws0;
    // float64(LastValue) of window id w (integral.go:51: t1 = float64(w.LastValue)) ; the time of a row as the walk's float64: relative
    // to ws0 in the 32-bit form (differences of two such times equal the differences of the absolute ones bit for bit), else absolute
    auto last_value_d = [&](uint32_t w) -> double { return (double)(ws0 + (int64_t)(((uint64_t)w + 1ull) * (uint64_t)(uint32_t)p.interval)); };
    auto time_d = [&](tkey_t k) -> double { return kTs32 ? (double)(uint32_t)k : (double)(int64_t)k; };
    auto wid_k = [&](tkey_t k) -> uint32_t { return kTs32 ? mdiv32((uint32_t)k, p.m32, p.sh1, p.sh2) : wid_of((int64_t)k); };
    auto abs_d = [&](double t) -> double { return kTs32 ? ws0_d + t : t; };

    // ---- the geometry of a window: its rows [r0, r1), its id and its successor's.  It depends on neither the column nor the phase, so
    // this lane's first window (q = lane: every window of the tile unless more than 64 start in it) is worked out once, here
    auto geometry = [&](int q, int &r0, int &r1, uint32_t &wid, uint32_t &next_wid, uint32_t &e1) -> int {   // 0: a window; 1: it runs past the look-ahead
        const uint32_t e0 = sh.seg[q];
        e1 = sh.seg[q + 1];
        r0 = (int)(e0 & kRowMask);
        // (the id of a head row from its staged time - the division the flag pass did - or, kLean, out of its entry)
        if (kLean) wid = w_first + (e0 >> 11); else wid = wid_k((tkey_t)sh.tsx[r0]);
        if (q + 1 < nseg_total) {
            r1 = (int)(e1 & kRowMask);
            if (kLean) next_wid = w_first + (e1 >> 11); else next_wid = wid_k((tkey_t)sh.tsx[r1]);
            return 0;
        }
        r1 = nloc;
        next_wid = W32;
        return reaches_end ? 0 : 1;
    };
    lds_order();   // the head list and the staged times are complete
    int g_r0 = 0, g_r1 = 0, g_state = 2;
    uint32_t g_wid = 0, g_next = 0, g_e1 = 0;
    if (lane < nseg_owned) g_state = geometry(lane, g_r0, g_r1, g_wid, g_next, g_e1);

    // ---- one pass per value column
    const int ncols = kMulti ? p.ncols : 1;
    for (int c = 0; c < ncols; c++) {
        const bool cint = p.col_is_int[c] != 0;  // mixed column types: per pass (uniform)
        // ---- stage the values (converted to float64, nulls replaced: rolling_simple.hip) - the gather source of the term pass for a
        // nullable column, and what the value reducers walk
        lds_order();  // the previous pass is done with sh.val / sh.vbits
        const bool early = early_terms && c == 0;   // its first kind of terms is in sh.val already (flag pass)
        if (kNulls) {
            if (lane < kRowsT / 32) sh.vbits[lane] = vword;
            lds_order();
        }
        // ---- stage the column: float64(v) (Int64 columns are converted here: bowgetters.go:224-229), null rows replaced as in
        // rolling_simple.hip.  It is what the value reducers walk and what the term pass reads.
        bool snan = false;   // a signalling NaN among the staged values: the tile's extrema are walked by comparison (agg_device.h is_snan)
        if (!early) {
            const uint64_t fill = need_sum ? 0ull : kNullAsNaN;
#pragma unroll
            for (int j = 0; j < kChunksT; j++) {
                const int l = j * 128 + 2 * lane;
                uint64_t ra_ = va[j], rb_ = vb[j];
                if (cint) {
                    ra_ = (uint64_t)__double_as_longlong((double)(int64_t)ra_);
                    rb_ = (uint64_t)__double_as_longlong((double)(int64_t)rb_);
                }
                if (kNulls) {
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) ra_ = fill;
                    if (!(two & 2u)) rb_ = fill;
                }
                if (need_mm && !cint) snan = snan || is_snan(ra_) || is_snan(rb_);
                *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzT>(l)]) = make_ulonglong2(ra_, rb_);
            }
        }
        const bool exact_mm = need_mm && __ballot(snan) != 0ull;
        // the next column's loads go out now: its registers are free (this column lives in LDS)
        if (c + 1 < ncols) {
            load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
            if (kNulls) vword = load_vword(c + 1);
        }
        lds_order();
        // ---- the term pass: the terms of this lane's ten rows from the staged values and times.  `first` (step terms when the call
        // has them, else trapezoid terms) replaces the staged values in LDS - chunk by chunk for a column without nulls (a row's
        // previous point is the previous row: read before the chunk is written, the row in front of the NEXT chunk saved first), and
        // for a nullable column too, whose chunks are taken from the last to the first (previous points are gathered from below).
        // kBoth: the trapezoid terms wait in registers until the step terms have been walked.
        auto term_pass = [&]() __attribute__((always_inline)) {
            if constexpr (kShort) return;
            else {
            const bool first_is_step = need_step;
            double carry_x = 0.0;                   // staged value of row 128 j - 1, read before chunk j - 1 was overwritten
            // (the head flags and the tile's row count do not change from column to column and phase to phase: left visible, the compiler
            // hoists the twenty lane masks derived from them to the top of the kernel and spills them - 60 scalar registers, a
            // v_readlane per use)
            uint32_t hm = hmask;
            int nl = nloc;
            asm volatile("" : "+v"(hm), "+s"(nl));
            // (a nullable column: chunks from the LAST to the first.  The previous points of a chunk's rows lie in the chunk itself or below
            // it, so once every lane has gathered for chunk j nothing will read the staged values of chunk j again - the chunks still to
            // come are below it - and its terms go into LDS right away, as for a column without nulls.  In forward order the terms of
            // all five chunks had to wait in registers for the end of the pass: 20 of them, which put the instantiation with both
            // kinds of integral 26 registers over its 128 - spilled inside this loop: 0.82 ms for ANY call through it, 0.55 without)
#pragma unroll
            for (int jj = 0; jj < kChunksT; jj++) {
                const int j = kNulls ? kChunksT - 1 - jj : jj;
                const int l = j * 128 + 2 * lane;
                const ulonglong2 xv = *reinterpret_cast<const ulonglong2 *>(&sh.val[swz<kSwzT>(l)]);
                const double xa = __longlong_as_double((long long)xv.x), xb = __longlong_as_double((long long)xv.y);
                tkey_t ka, kb;
                if (kTs32) { const uint2 k2 = *reinterpret_cast<const uint2 *>(&sh.tsx[l]); ka = (tkey_t)k2.x; kb = (tkey_t)k2.y; }
                else { const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(&sh.tsx[l]); ka = (tkey_t)k2.x; kb = (tkey_t)k2.y; }
                // float64(t1) - float64(t0) (integral.go:28, :54): with 32-bit offsets the difference of the offsets, converted - exact, and
                // equal to the difference of the two float64 times, which are exact themselves (every |ts| < 2^53)
                auto dt_of = [&](tkey_t k1, tkey_t k0) -> double {
                    return kTs32 ? (double)(uint32_t)((uint32_t)k1 - (uint32_t)k0) : time_d(k1) - time_d(k0);
                };
                const bool ha = (hm >> (2 * j)) & 1u, hb = (hm >> (2 * j + 1)) & 1u;
                double sa = 0.0, sb = 0.0, qa = 0.0, qb = 0.0;
                if (!kNulls) {
                    // previous point of row a: the row before it; of row b: row a.  A head's slot: the value of the point before it (step:
                    // what the window's lane multiplies by the time left in the window before; trapezoid: the joining term)
                    const int lp = l > 0 ? l - 1 : 0;
                    const double xp = (lane == 0 && j > 0) ? carry_x : __longlong_as_double((long long)sh.val[swz<kSwzT>(lp)]);
                    const double dta = dt_of(ka, (tkey_t)sh.tsx[lp]), dtb = dt_of(kb, ka);
                    if (need_step) {
                        sa = ha ? xp : xp * dta;
                        sb = hb ? xa : xa * dtb;
                    }
                    if (need_trap) {
                        qa = (xp + xa) / 2 * dta;
                        qb = (xa + xb) / 2 * dtb;
                    }
                    if (j + 1 < kChunksT) carry_x = __longlong_as_double((long long)sh.val[swz<kSwzT>(j * 128 + 127)]);
                    lds_order();   // every lane has read this chunk's values
                    const double oa = first_is_step ? sa : qa, ob = first_is_step ? sb : qb;
                    *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzT>(l)]) = make_ulonglong2((uint64_t)__double_as_longlong(oa), (uint64_t)__double_as_longlong(ob));
                } else {
                    uint32_t two = (sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31)) & 3u;
                    if (l >= nl) two = 0u; else if (l + 1 >= nl) two &= 1u;
                    const bool a_ok = two & 1u, b_ok = two & 2u;
                    // previous valid point of row a: the nearest set bit below it in the tile's validity words
                    int wi = l >> 5;
                    uint32_t m = sh.vbits[wi] & ((1u << (l & 31)) - 1u);
                    while (m == 0u && wi > 0) { wi--; m = sh.vbits[wi]; }
                    const int pr = m ? wi * 32 + 31 - __clz((int)m) : -1;
                    const bool has_p = pr >= 0;
                    const int prc = has_p ? pr : 0;
                    const double xp = __longlong_as_double((long long)sh.val[swz<kSwzT>(prc)]);
                    const tkey_t kp = (tkey_t)sh.tsx[prc];
                    // the same window?  (a head's previous point never is; a row that follows a head in its lane pair may not be either)
                    const uint32_t wa = wid_k(ka), wb = wid_k(kb), wp = wid_k(kp);
                    const bool same_a = has_p && !ha && wp == wa;
                    // ... of row b: row a when that is a valid point
                    const double xp2 = a_ok ? xa : xp;
                    const tkey_t kp2 = a_ok ? ka : kp;
                    const bool has_p2 = a_ok || has_p;
                    const bool same_b = has_p2 && !hb && (a_ok ? wa : wp) == wb;
                    const double dta = dt_of(ka, kp), dtb = dt_of(kb, kp2);
                    if (need_step) {
                        if (ha) sa = xp;                       // (read only when the window before has a valid point: then it is that point)
                        else if (a_ok && same_a) sa = xp * dta;
                        if (hb) sb = xp2;
                        else if (b_ok && same_b) sb = xp2 * dtb;
                    }
                    if (need_trap) {
                        if (a_ok && has_p && (ha || same_a)) qa = (xp + xa) / 2 * dta;
                        if (b_ok && has_p2 && (hb || same_b)) qb = (xp2 + xb) / 2 * dtb;
                    }
                    lds_order();   // every lane has gathered for this chunk
                    const double oa = first_is_step ? sa : qa, ob = first_is_step ? sb : qb;
                    *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzT>(l)]) = make_ulonglong2((uint64_t)__double_as_longlong(oa), (uint64_t)__double_as_longlong(ob));
                }
                if (kKeep) { keep_a[kKeep ? j : 0] = qa; keep_b[kKeep ? j : 0] = qb; }
                // (one chunk at a time: left to itself the scheduler interleaves the five unrolled chunks to hide latencies, which costs
                // more registers than the kernel has)
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        };
        // ---- phases, each one walk over what sh.val holds:
        //   0 / 1 / 2  the staged values (rolling_simple.hip: 1 + 2 when a nullable column feeds sums AND extrema), every output that
        //              is not an integral; 3  the step terms; 4  the trapezoid terms
        // A tile of many SHORT windows (more than kWalkAllMaxHeads heads in its 640 rows: windows of fewer than ~13 rows) whose column
        // needs more than one walk - values next to integrals, both kinds of integral, or nulls (the term pass gathers) - is served
        // by ONE walk that does everything per row (walk_all below: the round-1 walk).  At that length the chain is short and a phase
        // per kind costs more than it saves: 10-row windows, IntegralStep + IntegralTrapezoid + Mean 0.56 ms per 1e8 rows in phases
        // against 0.37 in one walk; from ~16 rows on the phases win and keep winning (64-row windows with nulls: 0.59 against 1.04 ms).
        const bool walk_all = kShort || (!early && nseg_total > kWalkAllMaxHeads && (need_step || need_trap));
        const bool two_phase = !walk_all && kNulls && need_mm && need_sum && nseg_total <= kTwoWalksMaxHeads;
        const bool pred_walk = kNulls && need_mm && need_sum && !two_phase;   // one walk, extrema under the validity bit (agg_device.h)
        for (int phase = two_phase ? 1 : 0; phase <= (walk_all ? 0 : 4); phase++) {
            if (phase == 1 && !two_phase) continue;
            if (phase == 2 && !two_phase) continue;
            if (phase == 0 && !walk_all && (two_phase || (!need_vals && (need_step || need_trap)))) continue;   // (no value reducer: the first integral phase also writes WindowStart / Count / NumRows)
            if (phase == 3 && !need_step) continue;
            if (phase == 4 && !need_trap) continue;
            if (phase == 2) {
                lds_order();
#pragma unroll
                for (int j = 0; j < kChunksT; j++) {
                    const int l = j * 128 + 2 * lane;
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) sh.val[swz<kSwzT>(l)] = kNullAsNaN;
                    if (!(two & 2u)) sh.val[swz<kSwzT>(l + 1)] = kNullAsNaN;
                }
                lds_order();
            }
            if (phase >= 3) {
                lds_order();   // the walks of the phase before are done with sh.val
                if (phase == 3 || !need_step) { if (!early) term_pass(); }
                else if (kKeep) {
#pragma unroll
                    for (int j = 0; j < kChunksT; j++)
                        *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzT>(j * 128 + 2 * lane)]) = make_ulonglong2(
                            (uint64_t)__double_as_longlong(keep_a[kKeep ? j : 0]), (uint64_t)__double_as_longlong(keep_b[kKeep ? j : 0]));
                }
                lds_order();
            }
            const bool do_sum = need_sum && phase <= 1;
            const bool do_mm = need_mm && (phase == 0 || phase == 2);
            // the outputs of this phase: every output in exactly one
            uint32_t out_mask;
            {
                const uint32_t others = p.kind_mask[4];
                const int ph1 = walk_all ? 0 : two_phase ? 1 : (need_vals || !(need_step || need_trap)) ? 0 : need_step ? 3 : 4;   // the first phase that runs
                out_mask = phase == 0 ? (p.kind_mask[0] | p.kind_mask[1]) : phase == 1 ? p.kind_mask[0] : phase == 2 ? p.kind_mask[1]
                           : phase == 3 ? p.kind_mask[2] : p.kind_mask[3];
                if (phase == ph1) out_mask |= others;
                if (walk_all) out_mask = p.kind_mask[0] | p.kind_mask[1] | p.kind_mask[2] | p.kind_mask[3] | others;
                out_mask &= p.col_mask[c];
            }
            const int phase1 = walk_all ? 0 : two_phase ? 1 : (need_vals || !(need_step || need_trap)) ? 0 : need_step ? 3 : 4;   // the first phase that runs
            const bool first_phase = phase == phase1;

    for (int q = lane; q < nseg_owned; q += kWave) {
        int r0 = g_r0, r1 = g_r1, state = g_state;
        uint32_t wid = g_wid, next_wid = g_next, e1 = g_e1;
        if (q != lane) state = geometry(q, r0, r1, wid, next_wid, e1);
        if (state != 0) {
            // rows run past the look-ahead: hand the window (all its columns) to the cooperative path
            if (c == 0 && first_phase) {
                push_long_window(p.status, p.long_list, p.long_cap, tile, (uint64_t)p.wid_base + w0 + wid, base + r0);
            }
            continue;
        }
        const bool next_staged = q + 1 < nseg_total;   // row r1 is a head of this tile: its slot holds what closes this window
        // does the successor's first row sit exactly on this window's end?  (rolling.go:201-209; only looked at for inclusive calls)
        const bool incl_row = p.inclusive && next_staged && next_wid == wid + 1 && (e1 & kStartBit);
        // window 0 made only of rows below s0 is an EMPTY slice in the reference (rolling.go:194-196: lastRowIndex stays -1)
        const bool dead = pre && tile == 0 && q == 0 && !(p.ts[base + r1 - 1] >= p.s0 || incl_row);
        int count = r1 - r0, fv = r0, lv = r1 - 1;
        if (kNulls) window_valid_rows(sh.vbits, r0, r1, count, fv, lv);
        if (dead) count = 0;
        const bool has_value = count > 0;
        double sum = 0.0, mn = 0.0, mx = 0.0, integ = 0.0, integ_t = 0.0;
        uint64_t first_raw = 0, last_raw = 0;
        bool trap_nil = true;
        if (walk_all) {
            // every reducer of the column in one walk over the window's valid rows, in row order (sum.go:16-22, minmax.go:16-28,
            // integral.go:14-31 / :46-62), then the inclusive row for the trapezoid
            if (has_value) {
                first_raw = sh.val[swz<kSwzT>(fv)];
                double pt = 0.0, pv = 0.0, step = 0.0, trap = 0.0;
                int cnt = 0;
                for (int r = fv; r <= lv; r++) {
                    if (kNulls && !((sh.vbits[r >> 5] >> (r & 31)) & 1u)) continue;
                    const uint64_t raw = sh.val[swz<kSwzT>(r)];
                    const double x = __longlong_as_double((long long)raw), t = time_d((tkey_t)sh.tsx[r]);
                    sum += x;
                    if (cnt == 0) { mn = x; mx = x; }
                    else {
                        if (need_mm) { if (x < mn) mn = x; if (x > mx) mx = x; }
                        const double dt = t - pt;
                        if (need_trap) trap += (pv + x) / 2 * dt;
                        if (need_step) step += pv * dt;
                    }
                    pt = t; pv = x; last_raw = raw;
                    cnt++;
                }
                integ = step + pv * (last_value_d(wid) - abs_d(pt));
                integ_t = trap;
                if (incl_row && (!kNulls || ((sh.vbits[r1 >> 5] >> (r1 & 31)) & 1u))) {
                    const double x = __longlong_as_double((long long)sh.val[swz<kSwzT>(r1)]);
                    integ_t = trap + (pv + x) / 2 * (time_d((tkey_t)sh.tsx[r1]) - pt);
                    cnt++;
                }
                trap_nil = cnt < 2;
                if (need_fl && cint) {
                    const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(p.values[c]);
                    first_raw = src[base + fv];
                    last_raw = src[base + lv];
                }
            }
        } else if (phase <= 2) {
            if (has_value && need_vals) {
                first_raw = sh.val[swz<kSwzT>(fv)];
                if (kNulls && pred_walk) walk_values_pred<kSwzT>(sh.val, sh.vbits, fv, lv, sum, mn, mx);
                else walk_values<kSwzT, kLean || kShort>(sh.val, fv, lv, do_sum, do_mm, exact_mm, sum, mn, mx);
                if (need_fl) {
                    last_raw = sh.val[swz<kSwzT>(lv)];
                    if (cint) {   // the staged values are float64(v): First / Last return the Int64 itself (firstlast.go:17, :32)
                        const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(p.values[c]);
                        first_raw = src[base + fv];
                        last_raw = src[base + lv];
                    }
                }
            }
        } else if (has_value) {
            // the terms of the window's points after its first (the first point's own slot holds +0.0, or - when it is the window's
            // first row - what closes the window before), in row order
            integ = walk_terms<kSwzT, kLean || (!kNulls && kTs32 && !kBoth)>(sh.val, fv + 1, lv + 1);
            if (phase == 3) {
                // + v0 * (float64(LastValue) - t0) of the last valid point (integral.go:49-55): staged in the next window's head slot
                if (kLean && next_staged) integ = integ + __longlong_as_double((long long)sh.val[swz<kSwzT>(r1)]);   // the next window's head slot: the closing product
                else {
                    double pv, tlast;
                    if (next_staged) pv = __longlong_as_double((long long)sh.val[swz<kSwzT>(r1)]);   // the next window's head slot: the value of the point before it
                    else {   // the data ends inside this window (one window per call): from the column itself
                        const uint64_t raw = reinterpret_cast<const uint64_t *>(p.values[c])[base + lv];
                        pv = cint ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
                    }
                    if (kLean) tlast = (double)p.ts[base + lv]; else tlast = abs_d(time_d((tkey_t)sh.tsx[lv]));
                    integ = integ + pv * (last_value_d(wid) - tlast);
                }
            } else {
                // the inclusive row (the successor's first row, when it sits on this window's end and is a valid point) joins in
                int cnt = count;
                if (incl_row && (!kNulls || ((sh.vbits[r1 >> 5] >> (r1 & 31)) & 1u))) {
                    integ += __longlong_as_double((long long)sh.val[swz<kSwzT>(r1)]);
                    cnt++;
                }
                trap_nil = cnt < 2;
                integ_t = integ;
            }
        } else if (phase == 4) {
            trap_nil = true;   // (no point of its own: at most the inclusive row - fewer than two points, integral.go:33-35)
        }
        const int nrows = dead ? 0 : r1 - r0;
        const int64_t win_start = ws0 + (int64_t)((uint64_t)wid * (uint64_t)(uint32_t)p.interval);
        const int64_t slot = (int64_t)(w0 + wid);  // output slot
        if (wid >= W32) continue;  // (only on corrupt input)
        const uint32_t gap = next_wid - wid - 1;
        // ---- outputs of this column and phase: lane q -> slot wid
        // (the output's pointer through a SCALAR load from the kernel-argument segment, the index made wave-uniform explicitly:
        // rolling_simple.hip - a vector load of it made every output wait for the previous output's store)
        // (the outputs of this column and phase as a bit mask from the host: no iteration, no descriptor load for the others)
#pragma unroll 1
        for (uint32_t am = out_mask; am; am &= am - 1u) {
            const int a = __builtin_amdgcn_readfirstlane(__builtin_ctz(am));
            const int k = p.kind[a];
            typedef const uint64_t __attribute__((address_space(4))) *karg_u64;
            typedef uint64_t __attribute__((address_space(1))) *global_u64;
            const global_u64 out_a = (global_u64)((karg_u64)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(SimpleParams, out_values) / 8 + a];
            uint64_t bits;
            bool nil = false;
            switch (k) {
            case BOWGPU_AGG_WINDOW_START: bits = (uint64_t)win_start; break;
            case BOWGPU_AGG_SUM: bits = (uint64_t)__double_as_longlong(sum); break;
            case BOWGPU_AGG_MEAN: bits = (uint64_t)__double_as_longlong(sum / (double)(int64_t)count); nil = !has_value; break;
            case BOWGPU_AGG_MIN: bits = (uint64_t)__double_as_longlong(mn); nil = !has_value; break;
            case BOWGPU_AGG_MAX: bits = (uint64_t)__double_as_longlong(mx); nil = !has_value; break;
            case BOWGPU_AGG_COUNT: bits = (uint64_t)(int64_t)count; break;
            case BOWGPU_AGG_FIRST: bits = first_raw; nil = !has_value; break;
            case BOWGPU_AGG_LAST: bits = last_raw; nil = !has_value; break;
            case BOWGPU_AGG_INTEGRAL_STEP:                                                  // integral.go:43-68
            case BOWGPU_AGG_WAVG_STEP: {                                                    // weightedmean.go:11-19
                double r = integ;
                if (k == BOWGPU_AGG_WAVG_STEP) r = r / (double)((win_start + p.interval) - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = !has_value;
                break;
            }
            case BOWGPU_AGG_INTEGRAL_TRAPEZOID:                                             // integral.go:11-37 (inclusive window)
            case BOWGPU_AGG_WAVG_LINEAR: {                                                  // weightedmean.go:25-33
                double r = integ_t;
                if (k == BOWGPU_AGG_WAVG_LINEAR) r = r / (double)((win_start + p.interval) - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = trap_nil;
                break;
            }
            default: bits = (uint64_t)__double_as_longlong((double)nrows); break;  // NumRows
            }
            const bool int_result = k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_COUNT || (cint && (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST));
            const int nf = p.nfac[a];
            if (nf) bits = apply_factors(bits, int_result, nf, p.fac[a]);
            if (nil) {  // no value (all null; fewer than two points for the trapezoid): nil => slot 0, bit cleared
                bits = 0;
                atomicAnd(&p.out_valid[a][slot >> 5], ~(1u << (slot & 31)));
            }
            __builtin_nontemporal_store(bits, &out_a[slot]);
            // the empty windows right after this one (rare): values of an empty slice + cleared validity bits
            // (A.9 "Empty slice": WindowStart s_k ; Sum 0.0 ; Count 0 ; NumRows 0.0 ; the rest nil)
            for (uint32_t g = 1; g <= gap; g++) {
                if (wid + g >= W32) break;
                const int64_t gw = slot + g;
                const int64_t gstart = win_start + (int64_t)((uint64_t)g * (uint64_t)(uint32_t)p.interval);
                uint64_t gbits = k == BOWGPU_AGG_WINDOW_START ? (uint64_t)gstart : 0ull;
                // (Sum / NumRows of an empty slice are +0.0 and Count is 0: a negative factor still turns the floats into -0.0)
                if (nf && (k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_SUM || k == BOWGPU_AGG_NUM_ROWS || k == BOWGPU_AGG_COUNT))
                    gbits = apply_factors(gbits, int_result, nf, p.fac[a]);
                out_a[gw] = gbits;
                if (p.out_valid[a]) atomicAnd(&p.out_valid[a][gw >> 5], ~(1u << (gw & 31)));
            }
        }
    }
        }  // phases
        if (!kNulls && c + 1 < ncols) load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
    }  // columns
}

int launch_rolling_tw(Ctx *c, const SimpleParams &p, bool is_int, bool has_nulls, bool wide, bool ts32) {
    (void)is_int;   // (column types are read per pass from p.col_is_int: the conversion happens where the values are staged)
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileT - 1) / kTileT;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    const dim3 g((unsigned)grid), blk(kWave);
    const bool both = (p.need & kNeedStep) && (p.need & kNeedTrap);
    // the lean form: one value column without nulls, integrals only, windows of 5 rows and more on average (its head list: kLeanCap)
    const bool lean = !has_nulls && p.ncols == 1 && !(p.need & (kNeedSum | kNeedMinMax | kNeedFirstLast)) && p.W > 0 && p.n / p.W >= 5;
    // the one-walk instantiation up to the window length where the phases take over (scratch/midw_sweep.py, 1e8 rows): a single kind of
    // integral on a column without nulls 14 rows, with nulls or next to value reducers 20; both kinds 12 where the lean form takes
    // over (its padded term array: 0.385 against 0.413 ms at 12 rows per window, 0.436 against 0.562 at 48), else 64 (38 with nulls:
    // 0.642 against 0.698 ms at 32 rows per window, 0.710 against 0.685 at 40)
    const int64_t short_rows = both ? (has_nulls ? 38 : lean ? 12 : 64) : (has_nulls || (p.need & (kNeedSum | kNeedMinMax | kNeedFirstLast))) ? 20 : kShortAvgRows;
    const bool shrt = p.W > 0 && p.n / p.W < short_rows;
    const bool multi = p.ncols > 1;
#define BG_TW4(U, B, S, L)                                                                                                       \
    do {                                                                                                                          \
        if (multi) {                                                                                                                      \
            if (wide) hipLaunchKernelGGL((rolling_tw_kernel<U, true, false, B, S, L, true>), g, blk, 0, c->stream, p, ntiles, per_xcd);      \
            else if (ts32) hipLaunchKernelGGL((rolling_tw_kernel<U, false, true, B, S, L, true>), g, blk, 0, c->stream, p, ntiles, per_xcd); \
            else hipLaunchKernelGGL((rolling_tw_kernel<U, false, false, B, S, L, true>), g, blk, 0, c->stream, p, ntiles, per_xcd);         \
        } else {                                                                                                                          \
            if (wide) hipLaunchKernelGGL((rolling_tw_kernel<U, true, false, B, S, L, false>), g, blk, 0, c->stream, p, ntiles, per_xcd);      \
            else if (ts32) hipLaunchKernelGGL((rolling_tw_kernel<U, false, true, B, S, L, false>), g, blk, 0, c->stream, p, ntiles, per_xcd); \
            else hipLaunchKernelGGL((rolling_tw_kernel<U, false, false, B, S, L, false>), g, blk, 0, c->stream, p, ntiles, per_xcd);         \
        }                                                                                                                                 \
    } while (0)
#define BG_TW3(U, B, S) BG_TW4(U, B, S, false)
#define BG_TW2(U, B) do { if (shrt) BG_TW3(U, false, true); else BG_TW3(U, B, false); } while (0)
#define BG_TW(U) do { if (both) BG_TW2(U, true); else BG_TW2(U, false); } while (0)
    if (lean && !(both && shrt)) { if (both) BG_TW4(false, true, false, true); else BG_TW4(false, false, false, true); }   // (both kinds over short windows: the one-walk form is faster)
    else if (has_nulls) BG_TW(true); else BG_TW(false);
#undef BG_TW4
#undef BG_TW3
#undef BG_TW
#undef BG_TW2
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
