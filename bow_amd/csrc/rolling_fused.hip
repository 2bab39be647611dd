// rolling_fused.hip — Rolling.Interpolate followed by Rolling.Aggregate as ONE pass over the rows, without materialising the
// interpolated frame:  r.Interpolate(interps...).Aggregate(aggrs...)  (reference rolling/interpolation.go:30-69 returns a Rolling
// over the interpolated Bow with the same interval and options; rolling/aggregation.go:123-145 consumes it).
//
// For an ascending, non-null interval column and exclusive windows the interpolated frame has the SAME window grid as the original
// (its first row sits on s_0 - the synthetic row of window 0 or an original row that starts it - and its last row is the original's)
// and window k of it holds
//     [ one synthetic row at s_k, unless the window's first row sits exactly on s_k - empty windows included
//       (interpolation.go:98-161: firstColValue != window.FirstValue) ]  +  the window's original rows.
// The synthetic row's value in column c is what the column's interpolator gives for the window (linear.go:8-38: the expression
// ((v2 - v0) * ((s_k - t0) / (t2 - t0))) + v0 on the nearest both-valid rows before / from the window's FirstIndex; stepprevious.go,
// windowstart.go, none.go), stored in the column's type (interpolation.go:149-150: an Int64 column truncates), and the reducers
// then see it as the window's FIRST row: Sum = ((0 + x_s) + x_1) + ..., Min / Max seeded by it, First = it, Count / NumRows + 1.
// So the tile kernel of rolling_simple.hip - one wavefront per tile of 512 + 128 rows, one lane walks one window in row order -
// only needs each window's synthetic value fed to the same left-to-right walk: bit-identical to the two-call path by construction.
//
// What it does NOT take raises a status flag and the host makes the two calls instead (api.cpp fused_run -> the generic path):
// windows that run past a tile's look-ahead, tiles denser than the head list, a run of more than 2048 null rows between a window
// and its neighbour point (the two-call path has the neighbour index for those).  Inclusive windows, time-weighted reducers, wide
// frames, rows below s0 and interval columns with nulls never get here (the host declines them up front).
#include <stddef.h>

#include <type_traits>

#include "bitmap_device.h"
#include "interp_device.h"

namespace bowgpu {

namespace {

constexpr int kWave = 64;
constexpr int kTileF = 512;
constexpr int kHaloF = 128;
constexpr int kRowsF = kTileF + kHaloF;
constexpr int kChunksF = kRowsF / 128;
constexpr int kCapF = 208;          // heads per tile (+ look-ahead): windows of 3.1 rows and more on average; LDS exactly 8 KB
constexpr int kAlignF = 16;
#ifndef BOWGPU_FUSED_NT
#define BOWGPU_FUSED_NT 1
#endif
constexpr bool kNtF = BOWGPU_FUSED_NT != 0;   // non-temporal loads of a tile's interior chunks (rolling_simple.hip); A/B: -DBOWGPU_FUSED_NT=0
#ifndef BOWGPU_FUSED_SWZ
#define BOWGPU_FUSED_SWZ 0
#endif
#ifndef BOWGPU_FUSED_T32
#define BOWGPU_FUSED_T32 1
#endif
constexpr bool kT32Fused = BOWGPU_FUSED_T32 != 0;   // Linear between two rows of the frame on 32-bit time differences (the q-loop); A/B: -DBOWGPU_FUSED_T32=0
constexpr bool kSwzFused = BOWGPU_FUSED_SWZ != 0;   // the staged column padded by two slots per 32 rows (agg_device.h swz): A/B -DBOWGPU_FUSED_SWZ=1

// The rows' times stay in LDS as 32-bit offsets from the first window start: a synthetic row needs the times of its two neighbour
// points (linear.go:34), rows a lane other than the window's holds.  (Fetched from the column instead - two 8-byte gathers per window
// after the tile's own loads - the kernel read 2.19 GB for 1.61 GB of rows at 1e8 rows and took 0.43 ms; profiles/r05_stdout_fused_ab.txt.)
// A head entry is the head's local row alone: its window id and whether it sits exactly on the window's start are recomputed from
// its staged time.  8 KB per wavefront: 20 per CU.
struct FusedShared {
    uint64_t val[kSwzFused ? swz_slots(kRowsF) : kRowsF];   // the staged column
    uint32_t tsx[kRowsF];              // ts - s0 of every row of the tile
    uint32_t vbits[kRowsF / 32 + 2];   // validity words of the value column for this tile
    uint16_t seg[kCapF + 2];           // heads in row order: local row
};
static_assert(sizeof(FusedShared) <= (kSwzFused ? 8192 + 320 : 8192), "LDS of the fused kernel: 8 KB (20 wavefronts per CU; padded: 8.3 KB, 18)");

// (the high half as the instruction itself: __umulhi reaches the backend as a 64-bit product of two zero-extended values, and with the
// magic number's extension hoisted into another block every division carried a v_mad_u64_u32 with a ZERO multiplicand - a quarter-rate
// instruction per division, 16 per tile; round 6, read off the ISA)
__device__ __forceinline__ uint32_t mulhi32(uint32_t m, uint32_t n) {
    uint32_t t;
    asm("v_mul_hi_u32 %0, %1, %2" : "=v"(t) : "s"(m), "v"(n));
    return t;
}
__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = mulhi32(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
#ifndef BOWGPU_FUSED_FDIV
#define BOWGPU_FUSED_FDIV 1
#endif
// n / interval through float64: n < 2^32 is exact as a double, inv = (1 / interval)(1 + d) with d = 2^-40 (+- 2^-52: the host's two
// roundings), the product rounds by at most 2^-53 relative.  With q = n / interval: the computed value is q (1 + d)(1 + e) - never below q
// (d > 2^-52), and below floor(q) + 1 because the distance of q to the next integer is at least 1 / interval while the excess is
// q (d + e) < 2^32 / interval * 2^-39: the truncating conversion gives floor(q).  Three full-rate instructions where the magic-number
// form takes a quarter-rate multiplication and four more (A/B: -DBOWGPU_FUSED_FDIV=0; every n x interval pair of tests/test_gpu_fused.py's sweep)
__device__ __forceinline__ uint32_t fdiv32(uint32_t n, double inv) { return (uint32_t)((double)n * inv); }
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
typedef unsigned long long u64x2_f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 load16_nt(const ulonglong2 *q) {
    const u64x2_f v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_f *>(q));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// rows fv .. lv of the staged column behind an optional synthetic first row of value xs (seeded), in row order: sum.go:16-22,
// arithmeticmean.go:17-24, minmax.go:16-28 over [x_s, x_fv, ..., x_lv].  The same loops and the same settling of v_min_f64 / v_max_f64
// as agg_device.h walk_values; the seed is the synthetic value when there is one, else the first valid row.
template <bool kSwz, bool kEight>
__device__ __forceinline__ void walk_seeded(const uint64_t *val, int fv, int lv, bool do_sum, bool do_mm, bool exact_mm, bool seeded, double xs,
                                            double &sum, double &mn, double &mx) {
    const double seed = seeded ? xs : __longlong_as_double((long long)val[swz<kSwz>(fv)]);
    sum = seeded ? 0.0 + xs : 0.0;
    mn = seed; mx = seed;
    const int rend = lv + 1;
    if (fv < rend) {
        if (do_sum && do_mm) walk_rows<kSwz, kEight, true, true>(val, fv, rend, sum, mn, mx);
        else if (do_mm) walk_rows<kSwz, kEight, false, true>(val, fv, rend, sum, mn, mx);
        else if (do_sum) walk_rows<kSwz, kEight, true, false>(val, fv, rend, sum, mn, mx);
    }
    if (do_mm) {
        if (seed != seed) { mn = seed; mx = seed; }
        else if (exact_mm || mn == 0.0 || mx == 0.0 || mn != mn || mx != mx) {
            mn = seed; mx = seed;
            for (int rr = fv; rr < rend; rr++) {
                const double x = __longlong_as_double((long long)val[swz<kSwz>(rr)]);
                if (x < mn) mn = x;
                if (x > mx) mx = x;
            }
        }
    }
}

// ... and for a nullable column staged with +0.0 in its null rows whose outputs want sums AND extrema, in a tile of many short windows:
// one walk, the extrema under the row's validity bit (agg_device.h walk_values_pred; minmax.go:16-28 as written, so a NaN seed stays)
template <bool kSwz>
__device__ __forceinline__ void walk_pred_seeded(const uint64_t *val, const uint32_t *vbits, int fv, int lv, bool seeded, double xs, double &sum, double &mn, double &mx) {
    const double seed = seeded ? xs : __longlong_as_double((long long)val[swz<kSwz>(fv)]);
    sum = seeded ? 0.0 + xs : 0.0;
    mn = seed; mx = seed;
    for (int r = fv; r <= lv; r++) {
        const double x = __longlong_as_double((long long)val[swz<kSwz>(r)]);
        sum += x;
        if ((vbits[r >> 5] >> (r & 31)) & 1u) {
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
    }
}

}  // namespace

// kNeed: bit0 min/max wanted, bit1 first/last wanted; kNulls: some value column has nulls; kMulti: more than one value column
template <int kNeed, bool kNulls, bool kMulti>
__global__ __launch_bounds__(kWave, kMulti ? 4 : 6) void rolling_fused_kernel(const FusedParams fp, const int64_t ntiles, const int64_t tiles_per_xcd) {
    __shared__ FusedShared sh;
    constexpr bool kSwzF = kSwzFused;
    const SimpleParams &p = fp.s;
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs (look-ahead rows hit the same L2)
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int64_t base = tile * kTileF;
    const int64_t n = p.n;
    const bool interior = base + kRowsF <= n;
    const int nloc = interior ? kRowsF : (int)(n - base);

    // ---- loads: ts, then the first value column right behind it (rolling_simple.hip: one block of loads for the usual tile)
    uint64_t ta[kChunksF], tb[kChunksF], va[kChunksF], vb[kChunksF];
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    auto load_col = [&](const uint64_t *__restrict__ src, uint64_t (&a)[kChunksF], uint64_t (&bb)[kChunksF], bool aligned) {
        if (interior && aligned) {
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(src + base) + lane;
#pragma unroll
            for (int j = 0; j < kChunksF; j++) {
                const ulonglong2 x = (kNtF && j > 0 && j < kChunksF - 1) ? load16_nt(q + j * 64) : q[j * 64];
                a[j] = x.x; bb[j] = x.y;
            }
        } else if (interior) {
            const uint64_t *q = src + base + 2 * lane;
#pragma unroll
            for (int j = 0; j < kChunksF; j++) { a[j] = q[j * 128]; bb[j] = q[j * 128 + 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksF; j++) load_pair(src, base + j * 128 + 2 * lane, n, aligned, a[j], bb[j]);
        }
    };
    if (interior && !(p.unaligned_mask & 0x80000001u)) {
        const ulonglong2 *qt = reinterpret_cast<const ulonglong2 *>(ts + base) + lane;
        const ulonglong2 *qv = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const uint64_t *>(p.values[0]) + base) + lane;
#pragma unroll
        for (int j = 0; j < kChunksF; j++) {
            const ulonglong2 x = (kNtF && j > 0 && j < kChunksF - 1) ? load16_nt(qt + j * 64) : qt[j * 64];
            ta[j] = x.x; tb[j] = x.y;
        }
#pragma unroll
        for (int j = 0; j < kChunksF; j++) {
            const ulonglong2 x = (kNtF && j > 0 && j < kChunksF - 1) ? load16_nt(qv + j * 64) : qv[j * 64];
            va[j] = x.x; vb[j] = x.y;
        }
    } else {
        load_col(ts, ta, tb, !(p.unaligned_mask >> 31));
        load_col(reinterpret_cast<const uint64_t *>(p.values[0]), va, vb, !(p.unaligned_mask & 1u));
    }
    auto load_vword = [&](int c) -> uint32_t {
        uint32_t word = 0xFFFFFFFFu;
        if (lane < kRowsF / 32 && p.vbits[c] != nullptr) {
            const int64_t bit = p.vbit0[c] + base + 32 * (int64_t)lane;
            const int64_t wi = bit >> 5;
            const int shb = (int)(bit & 31);
            const uint32_t lo = wi < p.vwords[c] ? p.vbits[c][wi] : 0u;
            const uint32_t hi = (shb != 0 && wi + 1 < p.vwords[c]) ? p.vbits[c][wi + 1] : 0u;
            word = shb ? ((lo >> shb) | (hi << (32 - shb))) : lo;
        }
        return word;
    };
    uint32_t vword = kNulls ? load_vword(0) : 0u;
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    const int64_t ws0 = p.s0;
    bool unsorted = false, sat = false;
    const uint32_t s0_lo = (uint32_t)ws0;
    const uint32_t ik = (uint32_t)p.interval;
    const double inv_ik = fp.inv_interval;
    auto wdiv = [&](uint32_t x) -> uint32_t { return BOWGPU_FUSED_FDIV ? fdiv32(x, inv_ik) : mdiv32(x, p.m32, p.sh1, p.sh2); };

    // ---- window ids (32-bit, global: the host sends frames whose rows lie within 2^32 of s0), head flags, compaction with a running
    // scalar count; every row's time goes to LDS as its offset from s0
    uint32_t left_w = base == 0 ? 0xFFFFFFFEu : wdiv((uint32_t)left0 - s0_lo);
    int64_t left_ts = left0;
    int nseg_total = 0, nseg_owned = 0;
#ifndef BOWGPU_FUSED_INTERIOR
#define BOWGPU_FUSED_INTERIOR 1
#endif
    // (an interior tile holds all its 640 rows: told to the compiler, the per-row "is there such a row" tests and the divergent regions
    // they guard leave the usual tile's flag pass; A/B: -DBOWGPU_FUSED_INTERIOR=0)
    auto flag_pass = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
    for (int j = 0; j < kChunksF; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = kFull || l < nloc, pb = kFull || l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
        const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
        const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
        unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
        const uint32_t ra = (uint32_t)tsa - s0_lo, rb = (uint32_t)tsb - s0_lo;
        const uint32_t wa = wdiv(ra);
        const uint32_t wb = wdiv(rb);
        const uint32_t wprev = left32(wb, left_w);
        const bool ha = pa && (wa != wprev);
        const bool hb = pb && (wb != wa);
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        if (ha && pos < kCapF) sh.seg[pos] = (uint16_t)l;
        pos += ha ? 1 : 0;
        if (hb && pos < kCapF) sh.seg[pos] = (uint16_t)(l + 1);
        *reinterpret_cast<uint2 *>(&sh.tsx[l]) = make_uint2(ra, rb);   // (l is even: one 8-byte LDS write)
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kChunksF - 2) nseg_owned = nseg_total;
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
    }
    };
    if (BOWGPU_FUSED_INTERIOR && interior) flag_pass(std::true_type{});
    else flag_pass(std::false_type{});
    if (__ballot(unsorted)) {  // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0) atomicOr(&p.status[0], 1u);
        return;
    }
    if (nseg_total > kCapF) sat = true;
    if (__ballot(sat)) {  // a tile this kernel cannot describe: the host makes the two calls instead
        if (lane == 0 && !__hip_atomic_load(&p.status[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[4], 1u);
        return;
    }

    const bool reaches_end = base + kRowsF >= n;
    // ---- which windows this wavefront outputs: those that start in its 512 rows, with the slot-aligned hand-over of rolling_simple.hip
    int q_start = 0, q_end = nseg_owned;
    {
        lds_order();
        auto wid_at = [&](int q) -> uint32_t { return wdiv(sh.tsx[sh.seg[q]]); };
        auto handover = [&](int qf, int qlim) -> int {
            if (qf >= qlim) return qf;
            const uint32_t gf = wid_at(qf);
            const uint32_t A = (gf + (kAlignF - 1)) & ~(uint32_t)(kAlignF - 1);
            if (A == gf) return qf;
            const int qi = qf + lane;
            const bool below = lane < kAlignF && qi < qlim && wid_at(qi < qlim ? qi : qf) < A;
            const int nb = __popcll(__ballot(below));
            return qf + nb < qlim ? qf + nb : qf;
        };
        int n128 = 0;   // heads inside this tile's first 128 rows (at most 128 of them: two list entries per lane)
        {
            const int qa = lane, qb = lane + 64;
            const bool a = qa < nseg_total && (int)sh.seg[qa < nseg_total ? qa : 0] < 128;
            const bool bq = qb < nseg_total && (int)sh.seg[qb < nseg_total ? qb : 0] < 128;
            n128 = __popcll(__ballot(a)) + __popcll(__ballot(bq));
        }
        if (tile > 0) q_start = handover(0, n128);
        q_end = handover(nseg_owned, nseg_total);
    }
    const uint32_t W32 = (uint64_t)p.W > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)p.W;
    const bool need_sum = p.need & kNeedSum;
    bool gave_up = false;   // a neighbour point further away than the bounded search looks: the host makes the two calls instead

    // ---- one pass per value column: stage (rolling_simple.hip: float64(v), nulls replaced), synthetic values, walk, store
    const int ncols = kMulti ? p.ncols : 1;
    for (int c = 0; c < ncols; c++) {
        const bool cint = p.col_is_int[c] != 0;
        const FusedCol fc = fp.cols[c];
        const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(p.values[c]);
        const uint32_t *cbits = p.vbits[c];
        lds_order();  // the previous pass is done with sh.val / sh.vbits
        if (kNulls) {
            if (lane < kRowsF / 32) sh.vbits[lane] = vword;
            lds_order();
        }
        bool snan = false;
        {
            const uint64_t fill = need_sum ? 0ull : kNullAsNaN;
            if (cint) {
                // a BRANCH (the column's type is uniform): as a select the compiler converted every row of every Float64 column too -
                // eight float64 instructions and four selects per chunk, 60 vector instructions per tile for nothing (round 6, the ISA)
                asm volatile("");
#pragma unroll
                for (int j = 0; j < kChunksF; j++) {
                    va[j] = (uint64_t)__double_as_longlong((double)(int64_t)va[j]);
                    vb[j] = (uint64_t)__double_as_longlong((double)(int64_t)vb[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < kChunksF; j++) {
                uint64_t xa = va[j], xb = vb[j];
                if (kNulls) {
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) xa = fill;
                    if (!(two & 2u)) xb = fill;
                }
                if ((kNeed & 1) && !cint) snan = snan || is_snan(xa) || is_snan(xb);
                *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzF>(j * 128 + 2 * lane)]) = make_ulonglong2(xa, xb);
            }
            if (kMulti && c + 1 < ncols) {
                load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
                if (kNulls) vword = load_vword(c + 1);
            }
        }
        lds_order();
        const bool exact_mm = (kNeed & 1) && __ballot(snan) != 0ull;

        // the two neighbour points of a window whose FirstIndex is local row a (0 <= a <= nloc): the nearest valid row of the column
        // before it and from it on (linear.go:20-31, stepprevious.go:19; the interval column has no nulls: both-valid == value valid).
        // Inside the tile: the tile's validity words; outside: a bounded walk over the column's bitmap.
        auto neighbours = [&](int a, NbPoint &pp, NbPoint &np) {
            pp.has = 0; pp.t = 0; pp.bits = 0; np.has = 0; np.t = 0; np.bits = 0;
            const bool want_prev = fc.kind == BOWGPU_INTERP_LINEAR || fc.kind == BOWGPU_INTERP_STEP_PREVIOUS;
            const bool want_next = fc.kind == BOWGPU_INTERP_LINEAR;
            int64_t prow = -1, nrow = -1;
            if (want_prev) {
                if (!kNulls || cbits == nullptr) prow = base + a - 1;   // (-1 in front of the frame's first row)
                else {
                    int pr = -1;
                    if (a > 0) {
                        int wi = (a - 1) >> 5;
                        uint32_t m = sh.vbits[wi] & (0xFFFFFFFFu >> (31 - ((a - 1) & 31)));
                        while (m == 0u && wi > 0) { wi--; m = sh.vbits[wi]; }
                        if (m) pr = wi * 32 + 31 - __clz((int)m);
                    }
                    prow = pr >= 0 ? base + pr : prev_valid_near(cbits, p.vbit0[c], n, base - 1, &gave_up);
                }
            }
            if (want_next) {
                if (!kNulls || cbits == nullptr) nrow = base + a < n ? base + a : -1;
                else {
                    int nr = -1;
                    if (a < nloc) {
                        const int wl = (nloc - 1) >> 5;
                        int wi = a >> 5;
                        uint32_t m = sh.vbits[wi] & (0xFFFFFFFFu << (a & 31));
                        while (m == 0u && wi < wl) { wi++; m = sh.vbits[wi]; }
                        if (m) { nr = wi * 32 + __ffs((int)m) - 1; if (nr >= nloc) nr = -1; }
                    }
                    nrow = nr >= 0 ? base + nr : next_valid_near(cbits, p.vbit0[c], n, base + nloc, &gave_up);
                }
            }
            // the points themselves: time and (Float64) value out of the staged tile when the row lies in it; an Int64 column hands its own
            // bits on (linear.go reads float64(v), StepPrevious copies the Int64: synth_value_pt converts)
            // (q.t: the row's time as its 32-bit offset from the first window start - every row of the frame lies within 2^32 of it, the
            // host's precondition; abs_points() below makes them the timestamps synth_value_pt reads)
            auto point = [&](int64_t row, NbPoint &q) {
                if (row < 0) return;
                q.has = 1;
                const int64_t loc = row - base;
                const bool in_tile = loc >= 0 && loc < nloc;
                q.t = (int64_t)(uint64_t)(in_tile ? sh.tsx[in_tile ? (int)loc : 0] : (uint32_t)((uint64_t)p.ts[row] - (uint64_t)ws0));
                if (!cint && in_tile) q.bits = sh.val[swz<kSwzF>((int)loc)];
                else q.bits = src[row];
            };
            point(prow, pp);
            point(nrow, np);
        };

        auto abs_points = [&](NbPoint &pp, NbPoint &np) { pp.t += ws0; np.t += ws0; };

        // a nullable column whose outputs want sums AND extrema is walked twice in tiles of few long windows (rolling_simple.hip): phase 1
        // with +0.0 in the null rows, then the null rows are overwritten with NaN and phase 2 walks the extrema; in tiles of many short
        // windows once, the extrema under the validity bit.  Every other shape: phase 0, one walk.
        const bool two_phase = kNulls && (kNeed & 1) && need_sum && nseg_total <= kTwoWalksMaxHeads;
        const bool pred_walk = kNulls && (kNeed & 1) && need_sum && !two_phase;   // one walk, extrema under the validity bit
        for (int phase = two_phase ? 1 : 0; phase <= (two_phase ? 2 : 0); phase++) {
            if (phase == 2) {
                lds_order();
#pragma unroll
                for (int j = 0; j < kChunksF; j++) {
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) sh.val[swz<kSwzF>(j * 128 + 2 * lane)] = kNullAsNaN;
                    if (!(two & 2u)) sh.val[swz<kSwzF>(j * 128 + 2 * lane) + 1] = kNullAsNaN;
                }
                lds_order();
            }
            const bool do_sum = need_sum && phase != 2;
            const bool do_mm = (kNeed & 1) && phase != 1;

    for (int q = q_start + lane; q < q_end; q += kWave) {
        const int r0 = (int)sh.seg[q];
        const uint32_t t0x = sh.tsx[r0];
        const uint32_t wid = wdiv(t0x);
        const bool exact = t0x == wid * ik;     // the window's first row sits on its start: no synthetic row (interpolation.go:108-116)
        int r1;
        uint32_t next_wid;
        if (q + 1 < nseg_total) {
            r1 = (int)sh.seg[q + 1];
            next_wid = wdiv(sh.tsx[r1]);
        } else if (reaches_end) {
            r1 = nloc;
            next_wid = W32;
        } else {
            gave_up = true;   // rows run past the look-ahead: the two-call path has the long-window forms
            continue;
        }
        if (wid >= W32) continue;  // (only on corrupt input)
        const int64_t win_start = ws0 + (int64_t)((uint64_t)wid * (uint64_t)ik);
        // ---- the synthetic row in front of the window's rows (interpolation.go:118-160), as the reducers of this column see it
        uint64_t sbits = 0;
        int sv = 0;
        if (!exact) {
            NbPoint pp, np;
            neighbours(r0, pp, np);
            if (kT32Fused && fc.kind == BOWGPU_INTERP_LINEAR && pp.has && np.has) {
                // linear.go:34 between two rows of the frame: float64(s) - float64(t0) and float64(t2) - float64(t0) are differences of
                // exact values (every |time| < 2^53, the host's precondition) that are integers below 2^32 themselves (t0 < s <= t2: the
                // point before the window's first row lies in an earlier window, the point from it on in this one or later), so each
                // equals its 32-bit integer difference, converted: two subtractions and two conversions instead of three 64-bit
                // conversions with their range checks (synth_value_pt; bit-identical - tests/test_gpu_fused.py, the fuzz)
                const double a = (double)(uint32_t)(wid * ik - (uint32_t)pp.t), b = (double)(uint32_t)((uint32_t)np.t - (uint32_t)pp.t);
                const double v0 = bits_to_f64(pp.bits, fc.type), v2 = bits_to_f64(np.bits, fc.type);
                const double r = ((v2 - v0) * (a / b)) + v0;
                sbits = fc.type == BOWGPU_INT64 ? (uint64_t)go_f64_to_i64(r) : (uint64_t)__double_as_longlong(r);
                sv = 1;
            } else {
                abs_points(pp, np);
                synth_value_pt(fc, win_start, pp, np, &sbits, &sv);
            }
        }
        double xs = __longlong_as_double((long long)sbits);
        if (cint) { asm volatile(""); xs = (double)(int64_t)sbits; }   // bowgetters.go:224-229 (a branch: see the staging loop)
        // ---- valid rows of the window: how many, the first, the last; then the walk behind the synthetic row
        int count = r1 - r0, fv = r0, lv = r1 - 1;
        if (kNulls) window_valid_rows(sh.vbits, r0, r1, count, fv, lv);
        const bool own = count > 0;      // the window has a valid row of its own
        if (!own) { fv = r0; lv = r0 - 1; }
        count += sv;
        const bool has_value = count > 0;
        double sum = 0.0, mn = 0.0, mx = 0.0;
        uint64_t first_raw = 0, last_raw = 0;
        if (has_value) {
            if (kNulls && pred_walk) walk_pred_seeded<kSwzF>(sh.val, sh.vbits, fv, lv, sv != 0, xs, sum, mn, mx);
            else walk_seeded<kSwzF, !kMulti>(sh.val, fv, lv, do_sum, do_mm, exact_mm, sv != 0, xs, sum, mn, mx);
            if (kNeed & 2) {
                if (own) {
                    first_raw = sh.val[swz<kSwzF>(fv)];
                    last_raw = sh.val[swz<kSwzF>(lv)];
                    if (cint) { first_raw = src[base + fv]; last_raw = src[base + lv]; }   // First / Last return the Int64 itself (firstlast.go:17, :32)
                }
                if (sv) { first_raw = sbits; if (!own) last_raw = sbits; }
            }
        }
        const int nrows = (r1 - r0) + (exact ? 0 : 1);
        const int64_t slot = (int64_t)wid;
        const uint32_t gap = next_wid - wid - 1;
        typedef const uint64_t __attribute__((address_space(4))) *karg_u64;
        typedef uint64_t __attribute__((address_space(1))) *global_u64;
#pragma unroll 1
        for (int a_ = 0; a_ < p.naggs; a_++) {
            const int a = __builtin_amdgcn_readfirstlane(a_);
            if (kMulti && p.col[a] != c) continue;
            const int k = p.kind[a];
            if (phase != 0 && (phase == 2) != (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX)) continue;   // two walks: each output once
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
            default: bits = (uint64_t)__double_as_longlong((double)nrows); break;  // NumRows
            }
            const bool int_result = k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_COUNT || (cint && (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST));
            const int nf = p.nfac[a];
            if (nf) bits = apply_factors(bits, int_result, nf, p.fac[a]);
            if (nil) {
                bits = 0;
                atomicAnd(&p.out_valid[a][slot >> 5], ~(1u << (slot & 31)));
            }
            __builtin_nontemporal_store(bits, &out_a[slot]);
        }
        // the empty windows right after this one hold ONE row each, their synthetic row: every one of them has FirstIndex r1 (the next
        // window's first row: rolling.go:224-231 leaves currRowIndex there), so one pair of neighbour points serves the whole run.
        // Its own loop over the outputs, behind a branch: inside the loop above, what the run needs that does not depend on the output
        // (the neighbour points' times and values as float64, their differences) was computed in front of that loop for EVERY window,
        // gap or not - 45 vector instructions per pass over the heads (round 6, the ISA); frames without empty windows now skip all of it.
        if (gap == 0) continue;
        asm volatile("");
        NbPoint gp, gn;
        neighbours(r1, gp, gn);
        abs_points(gp, gn);
#pragma unroll 1
        for (int a_ = 0; a_ < p.naggs; a_++) {
            const int a = __builtin_amdgcn_readfirstlane(a_);
            if (kMulti && p.col[a] != c) continue;
            const int k = p.kind[a];
            if (phase != 0 && (phase == 2) != (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX)) continue;
            const global_u64 out_a = (global_u64)((karg_u64)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(SimpleParams, out_values) / 8 + a];
            const bool int_result = k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_COUNT || (cint && (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST));
            const int nf = p.nfac[a];
            for (uint32_t g = 1; g <= gap; g++) {
                if (wid + g >= W32) break;
                const int64_t gw = slot + g;
                const int64_t gstart = win_start + (int64_t)((uint64_t)g * (uint64_t)ik);
                uint64_t gs = 0;
                int gv = 0;
                synth_value_pt(fc, gstart, gp, gn, &gs, &gv);
                const double gx = cint ? (double)(int64_t)gs : __longlong_as_double((long long)gs);
                uint64_t gbits;
                bool gnil = false;
                switch (k) {
                case BOWGPU_AGG_WINDOW_START: gbits = (uint64_t)gstart; break;
                case BOWGPU_AGG_SUM: gbits = (uint64_t)__double_as_longlong(gv ? 0.0 + gx : 0.0); break;
                case BOWGPU_AGG_MEAN: gbits = (uint64_t)__double_as_longlong((0.0 + gx) / 1.0); gnil = !gv; break;
                case BOWGPU_AGG_MIN: case BOWGPU_AGG_MAX: gbits = (uint64_t)__double_as_longlong(gx); gnil = !gv; break;
                case BOWGPU_AGG_COUNT: gbits = (uint64_t)(int64_t)gv; break;
                case BOWGPU_AGG_FIRST: case BOWGPU_AGG_LAST: gbits = gs; gnil = !gv; break;
                default: gbits = (uint64_t)__double_as_longlong(1.0); break;  // NumRows: the synthetic row
                }
                if (nf) gbits = apply_factors(gbits, int_result, nf, p.fac[a]);
                if (gnil) {
                    gbits = 0;
                    atomicAnd(&p.out_valid[a][gw >> 5], ~(1u << (gw & 31)));
                }
                out_a[gw] = gbits;
            }
        }
    }
        }  // phases
    }  // columns
    if (__ballot(gave_up)) {
        if (lane == 0 && !__hip_atomic_load(&p.status[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[5], 1u);
    }
}

int launch_rolling_fused(Ctx *c, const FusedParams &fp, int need, bool has_nulls) {
    const SimpleParams &p = fp.s;
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileF - 1) / kTileF;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    const dim3 g((unsigned)grid), blk(kWave);
#define BG_FGO(N, U, M) hipLaunchKernelGGL((rolling_fused_kernel<N, U, M>), g, blk, 0, c->stream, fp, ntiles, per_xcd)
#define BG_FM(N, U) do { if (p.ncols > 1) BG_FGO(N, U, true); else BG_FGO(N, U, false); } while (0)
#define BG_FN(U) switch (need) { case 0: BG_FM(0, U); break; case 1: BG_FM(1, U); break; case 2: BG_FM(2, U); break; default: BG_FM(3, U); break; }
    if (has_nulls) { BG_FN(true) } else { BG_FN(false) }
#undef BG_FN
#undef BG_FM
#undef BG_FGO
    BG_HIP(hipGetLastError());
    c->last_kernel_name = "rolling_fused_kernel";
    return 0;
}

}  // namespace bowgpu
