// rolling_tw.hip — the wave-tile kernel for the TIME-WEIGHTED reducers and INCLUSIVE windows: IntegralStep,
// IntegralTrapezoid, WeightedAverageStep, WeightedAverageLinear (reference rolling/aggregation/integral.go:8-69,
// weightedmean.go:8-34) next to WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows, with the
// windows built inclusive when a reducer asks for it (rolling/aggregation.go:183-185, :207-211; rolling.go:201-209).
// Same structure and limits as rolling_simple.hip (one wavefront per tile of 512 + 128 rows, 32-bit global window ids,
// preset validity bitmaps, one lane walks one window in row order => bit-exact sums and integrals); in addition the
// interval column is staged in LDS as float64 (the reducers read float64(ts): integral.go:17,:49) and a window whose
// successor starts exactly at its end also folds that first row of the successor in for the reducers that declared
// NeedInclusiveWindow, while the others see the window without it (Window.UnsetInclusive, window.go:23-31).
// What this kernel does not take goes to rolling_agg.hip (rows below s0, mixed column types, 64-bit window ids).
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
constexpr uint32_t kSatT = 0xFFFFu;
// At most kSegCap windows may start inside one tile (+ look-ahead); denser tiles send the call to the general kernel.
// The kernel's rate follows its OCCUPANCY (same wave lifetime as rolling_simple.hip, fewer resident wavefronts: measured), and the
// occupancy follows the LDS per wavefront, so the segment list is sized to the byte:
//   float64 timestamps (kTs32 = false): 5120 + 5120 + 88 + 4 * 274 = 11424 B  => 14 wavefronts per CU;   272 heads = windows of >= 2.4 rows
//   32-bit offsets     (kTs32 = true) : 5120 + 2560 (+ 88 with nulls) + 2 * cap <= 8192 B => 20 wavefronts per CU; the list holds 16-bit
//   entries (row | on-window-start flag << 15; the window id of a head is recomputed from its staged offset): 240 heads (200 with
//   nulls) = windows of >= 2.7 (3.2) rows on average
template <bool kTs32, bool kNulls>
struct TwCap { static constexpr int value = kTs32 ? (kNulls ? 200 : 240) : 272; };

// kTs32: the timestamps are staged as 32-bit offsets from the tile's base window start instead of float64 values.  float64(ts) is
// then rebuilt as float64(base) + float64(offset), which is exact - and so equal to the reference's single conversion
// (integral.go:17) - when every |ts| of the call is below 2^53 (the host checks; nanosecond epochs take the float64 form).
template <bool kTs32, bool kNulls>
struct TwShared {
    uint64_t val[kRowsT];
    typename std::conditional<kTs32, uint32_t, double>::type tsf[kRowsT];   // float64(ts) of every row of the tile (integral.go:17), or its 32-bit offset
    uint32_t vbits[kNulls ? kRowsT / 32 + 2 : 1];  // validity words of the value column for this tile (kNulls only)
    // heads in row order: local row | on-window-start flag << 15 (| (wid - wid of the tile's first row) << 16 in the 32-bit form)
    typename std::conditional<kTs32, uint16_t, uint32_t>::type seg[TwCap<kTs32, kNulls>::value + 2];
};

__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// non-temporal 16-byte loads / 8-byte stores: see rolling_simple.hip (rows 128..511 of a tile are read by this wavefront only;
// outputs are never read back)
typedef unsigned long long u64x2_tw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 load16_nt(const ulonglong2 *q) {
    const u64x2_tw v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_tw *>(q));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void store8_nt(uint64_t *p, uint64_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// kInt: Int64 value columns; kNulls: some column has nulls
// kWide: see rolling_simple.hip (window ids relative to the tile's first window: rows may span more than 2^32 from slot 0)
template <bool kInt, bool kNulls, bool kWide, bool kTs32>
__global__ __launch_bounds__(kWave, 3) void rolling_tw_kernel(const SimpleParams p, const int64_t ntiles, const int64_t tiles_per_xcd) {
    static_assert(!(kWide && kTs32), "the wide form keeps float64 timestamps");
    constexpr bool kMulti = true;  // one pass per value column, always in loop form
    __shared__ TwShared<kTs32, kNulls> sh;
    constexpr int kSegCapT = TwCap<kTs32, kNulls>::value;
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs (look-ahead rows hit the same L2)
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int64_t base = tile * kTileT;
    const int64_t n = p.n;
    const bool interior = base + kRowsT <= n;
    const int nloc = interior ? kRowsT : (int)(n - base);

    // ---- loads: ts, then the first value column right behind it
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
    // The usual tile - interior, both columns 16-byte aligned - issues its ten loads back to back in ONE block, and nothing in
    // front of them (rolling_simple.hip: two load_col() calls put a wait for all outstanding loads between the columns, and the
    // status word that used to be looked at here cost every tile a memory round trip before its real loads)
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

    // ---- window ids (32-bit), head flags, compaction with a running scalar count
    const uint32_t w_first = (kWide || pre) ? 0u : mdiv32((uint32_t)ts_first - s0_lo, p.m32, p.sh1, p.sh2);
    uint32_t left_w = base == 0 ? 0xFFFFFFFEu : (pre && left0 < ws0) ? 0u : (kWide && left0 < ws0) ? 0xFFFFFFFEu : mdiv32(rel32(left0), p.m32, p.sh1, p.sh2);
    int64_t left_ts = left0;
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
        const uint32_t la = wa - w_first, lb = wb - w_first;
        if (!kTs32) sat |= (ha && la >= kSatT) || (hb && lb >= kSatT);   // (the 16-bit id field of the 32-bit list entries)
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        // bit 15: the row sits exactly on its window's start (what makes it the inclusive row of the window before)
        const uint32_t ik = (uint32_t)((uint64_t)p.interval >> (kWide ? p.shift_k : 0));
        const uint64_t lowmask = kWide ? ((1ull << p.shift_k) - 1ull) : 0ull;
        uint32_t sa = 0u, sb = 0u;
        if (p.inclusive) {   // (uniform: only inclusive calls look at the flag)
            sa = (ra == wa * ik && (((uint64_t)tsa - (uint64_t)ws0) & lowmask) == 0) ? 0x8000u : 0u;
            sb = (rb == wb * ik && (((uint64_t)tsb - (uint64_t)ws0) & lowmask) == 0) ? 0x8000u : 0u;
        }
        if (ha && pos < kSegCapT) sh.seg[pos] = kTs32 ? ((uint32_t)l | sa) : ((uint32_t)l | sa | (la << 16));
        pos += ha ? 1 : 0;
        if (hb && pos < kSegCapT) sh.seg[pos] = kTs32 ? ((uint32_t)(l + 1) | sb) : ((uint32_t)(l + 1) | sb | (lb << 16));
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kChunksT - 2) nseg_owned = nseg_total;
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
        if (kTs32) *reinterpret_cast<uint2 *>(&sh.tsf[l]) = make_uint2(ra, rb);   // (l is even: one 8-byte LDS write)
        else { sh.tsf[l] = (double)tsa; sh.tsf[l + 1] = (double)tsb; }
    }
    if (__ballot(unsorted)) {  // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0) atomicOr(&p.status[0], 1u);
        return;
    }
    if (nseg_total > kSegCapT) sat = true;
    if (__ballot(sat)) {  // a tile the 16-bit local ids / the segment list cannot describe: the host redoes the call with the general lean kernel
        // (one atomic per call, not one per tile: 2e5 atomics on one address took 2 ms)
        if (lane == 0 && !__hip_atomic_load(&p.status[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[4], 1u);
        return;
    }

    const bool reaches_end = base + kRowsT >= n;
    // which running statistics the outputs of this call read (wave-uniform: the walk skips the others' arithmetic; a call with
    // WeightedAverageStep alone then does two float64 operations per row instead of seven)
    bool need_step = false, need_trap = false, need_mm = false;
    for (int a = 0; a < p.naggs; a++) {
        const int k = p.kind[a];
        need_step |= k == BOWGPU_AGG_INTEGRAL_STEP || k == BOWGPU_AGG_WAVG_STEP;
        need_trap |= k == BOWGPU_AGG_INTEGRAL_TRAPEZOID || k == BOWGPU_AGG_WAVG_LINEAR;
        need_mm |= k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX;
    }
    // windows of the call, as an id relative to w0 (the last tile's successor id when the data ends in it)
    const uint64_t Wrel = (uint64_t)p.W - w0;
    const uint32_t W32 = Wrel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)Wrel;
    // ---- one pass per value column: stage its values in LDS (the next column's loads go out first), walk, store
    const int ncols = kMulti ? p.ncols : 1;
    for (int c = 0; c < ncols; c++) {
        const bool cint = kMulti ? (p.col_is_int[c] != 0) : kInt;  // mixed column types: per pass (uniform)
        if (kMulti) {
            lds_order();  // the previous pass is done with sh.val / sh.vbits
#pragma unroll
            for (int j = 0; j < kChunksT; j++)
                *reinterpret_cast<ulonglong2 *>(&sh.val[j * 128 + 2 * lane]) = make_ulonglong2(va[j], vb[j]);
            if (c + 1 < ncols) load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
        }
        if (kNulls && lane < kRowsT / 32) {  // 32 validity bits per lane, any bit offset (Arrow slices)
            uint32_t word = 0xFFFFFFFFu;
            if (p.vbits[c] != nullptr) {
                const int64_t bit = p.vbit0[c] + base + 32 * (int64_t)lane;
                const int64_t wi = bit >> 5;
                const int shb = (int)(bit & 31);
                const uint32_t lo = wi < p.vwords[c] ? p.vbits[c][wi] : 0u;
                const uint32_t hi = (shb != 0 && wi + 1 < p.vwords[c]) ? p.vbits[c][wi + 1] : 0u;
                word = shb ? ((lo >> shb) | (hi << (32 - shb))) : lo;
            }
            sh.vbits[lane] = word;
        }
        lds_order();

    for (int q = lane; q < nseg_owned; q += kWave) {
        const uint32_t e0 = sh.seg[q], e1 = sh.seg[q + 1];
        const int r0 = (int)(e0 & 0x7FFFu);
        // (kTs32: the id of a head row from its staged offset - the same division the flag pass did)
        const uint32_t wid = kTs32 ? mdiv32((uint32_t)sh.tsf[r0], p.m32, p.sh1, p.sh2) : w_first + (e0 >> 16);
        int r1;
        uint32_t next_wid;
        if (q + 1 < nseg_total) {
            r1 = (int)(e1 & 0x7FFFu);
            next_wid = kTs32 ? mdiv32((uint32_t)sh.tsf[r1], p.m32, p.sh1, p.sh2) : w_first + (e1 >> 16);
        } else if (reaches_end) {
            r1 = nloc;
            next_wid = W32;
        } else {
            // rows run past the look-ahead: hand the window (all its columns) to the cooperative path
            if (c == 0) {
                push_long_window(p.status, p.long_list, p.long_cap, tile, (uint64_t)p.wid_base + w0 + wid, base + r0);
            }
            continue;
        }
        // does the successor's first row sit exactly on this window's end?  (rolling.go:201-209; only looked at for inclusive calls)
        const bool incl_row = p.inclusive && q + 1 < nseg_total && next_wid == wid + 1 && (e1 & 0x8000u);
        // window 0 made only of rows below s0 is an EMPTY slice in the reference (rolling.go:194-196: lastRowIndex stays -1)
        const bool dead = pre && tile == 0 && q == 0 && !(p.ts[base + r1 - 1] >= p.s0 || incl_row);
        // ---- the walk: rows r0 .. r1-1 in order (sum.go:16-22, arithmeticmean.go:17-24, minmax.go:16-28, count.go:12-18,
        // firstlast.go, integral.go:14-31 / :46-62), then the inclusive row for the reducers that want it
        double sum = 0.0, mn = 0.0, mx = 0.0;
        uint64_t first_raw = 0, last_raw = 0;
        int count = 0;
        double pt = 0.0, pv = 0.0, integ_step = 0.0, integ_trap = 0.0;
        const double ws0_d = (double)ws0;
        // kTs32: the walk runs on times RELATIVE to ws0 (one conversion per row): every |ts| is below 2^53, so ws0 + offset is exact
        // and differences of two such times equal the differences of their offsets bit for bit; the absolute time of the last
        // point is rebuilt once, after the walk
        auto ts_at = [&](int r) -> double { return kTs32 ? (double)(uint32_t)sh.tsf[r] : (double)sh.tsf[r]; };
        auto step = [&](uint64_t raw, double t) {   // one valid row, in row order
            const double x = cint ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
            sum += x;
            if (count == 0) { mn = x; mx = x; first_raw = raw; }
            else {
                if (need_mm) { if (x < mn) mn = x; if (x > mx) mx = x; }
                const double dt = t - pt;
                if (need_trap) integ_trap += (pv + x) / 2 * dt;
                if (need_step) integ_step += pv * dt;
            }
            pt = t; pv = x;
            last_raw = raw;
            count++;
        };
        if (dead) { }
        else if (!kNulls) {
            int r = r0;
            for (; r + 4 <= r1; r += 4) {   // four rows' LDS reads in flight; the arithmetic stays in row order
                const uint64_t q0 = sh.val[r], q1 = sh.val[r + 1], q2 = sh.val[r + 2], q3 = sh.val[r + 3];
                const double t0 = ts_at(r), t1 = ts_at(r + 1), t2 = ts_at(r + 2), t3 = ts_at(r + 3);
                step(q0, t0); step(q1, t1); step(q2, t2); step(q3, t3);
            }
            for (; r < r1; r++) step(sh.val[r], ts_at(r));
        } else {
            for (int r = r0; r < r1; r++) {
                if (!((sh.vbits[r >> 5] >> (r & 31)) & 1u)) continue;
                step(sh.val[r], ts_at(r));
            }
        }
        // the same state with the inclusive row folded in (only the trapezoid integral and its point count are read)
        double integ_trap_incl = integ_trap;
        int count_incl = count;
        if (incl_row && (!kNulls || ((sh.vbits[r1 >> 5] >> (r1 & 31)) & 1u))) {
            const uint64_t raw = sh.val[r1];
            const double x = cint ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
            if (count > 0) integ_trap_incl += (pv + x) / 2 * (ts_at(r1) - pt);
            count_incl++;
        }
        if (kTs32) pt = ws0_d + pt;   // (absolute again: integral.go:66 reads float64(last_value) - t of the last point)
        const int nrows = dead ? 0 : r1 - r0;
        const bool has_value = count > 0;
        const int64_t win_start = ws0 + (int64_t)((uint64_t)wid * (uint64_t)(uint32_t)p.interval);
        const int64_t slot = (int64_t)(w0 + wid);  // output slot
        if (wid >= W32) continue;  // (only on corrupt input)
        const uint32_t gap = next_wid - wid - 1;
        // ---- outputs of this column: lane q -> slot wid
        // (the output's pointer through a SCALAR load from the kernel-argument segment, the index made wave-uniform explicitly:
        // rolling_simple.hip - a vector load of it made every output wait for the previous output's store)
#pragma unroll 1
        for (int a_ = 0; a_ < p.naggs; a_++) {
            const int a = __builtin_amdgcn_readfirstlane(a_);
            if (kMulti && p.col[a] != c) continue;
            typedef const uint64_t __attribute__((address_space(4))) *karg_u64;
            typedef uint64_t __attribute__((address_space(1))) *global_u64;
            const global_u64 out_a = (global_u64)((karg_u64)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(SimpleParams, out_values) / 8 + a];
            uint64_t bits;
            bool nil = false;
            const int k = p.kind[a];
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
                const int64_t last_value = win_start + p.interval;
                double r = integ_step + pv * ((double)last_value - pt);
                if (k == BOWGPU_AGG_WAVG_STEP) r = r / (double)(last_value - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = !has_value;
                break;
            }
            case BOWGPU_AGG_INTEGRAL_TRAPEZOID:                                             // integral.go:11-37 (inclusive window)
            case BOWGPU_AGG_WAVG_LINEAR: {                                                  // weightedmean.go:25-33
                double r = integ_trap_incl;
                if (k == BOWGPU_AGG_WAVG_LINEAR) r = r / (double)((win_start + p.interval) - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = count_incl < 2;
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
    }  // columns
}

int launch_rolling_tw(Ctx *c, const SimpleParams &p, bool is_int, bool has_nulls, bool wide, bool ts32) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileT - 1) / kTileT;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    const dim3 g((unsigned)grid), blk(kWave);
#define BG_TW(I, U)                                                                                                      \
    do {                                                                                                                 \
        if (wide) hipLaunchKernelGGL((rolling_tw_kernel<I, U, true, false>), g, blk, 0, c->stream, p, ntiles, per_xcd);    \
        else if (ts32) hipLaunchKernelGGL((rolling_tw_kernel<I, U, false, true>), g, blk, 0, c->stream, p, ntiles, per_xcd); \
        else hipLaunchKernelGGL((rolling_tw_kernel<I, U, false, false>), g, blk, 0, c->stream, p, ntiles, per_xcd);       \
    } while (0)
    if (is_int) { if (has_nulls) BG_TW(true, true); else BG_TW(true, false); }
    else { if (has_nulls) BG_TW(false, true); else BG_TW(false, false); }
#undef BG_TW
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
