// agg_device.h — device-side helpers shared by the tile kernels: exact division by the interval,
// Go numeric conversions, the running state of a window and the reducers' result rules
// (reference rolling/aggregation/*.go; see each function).
#pragma once

#include "common.h"

namespace bowgpu {

constexpr uint32_t kSat = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t magic_div(uint64_t n, const MagicDiv &d) {
    uint64_t t = __umul64hi(d.m, n);
    return (t + ((n - t) >> d.sh1)) >> d.sh2;
}

// Go's float64 -> int64 conversion on amd64 (CVTTSD2SI): NaN / out of range => INT64_MIN.
__device__ __forceinline__ int64_t go_f64_to_i64(double x) {
    if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)x;
}

__device__ __forceinline__ double bits_to_f64(uint64_t b, int type) {
    // Bow.GetFloat64: bowgetters.go:224-229 (Int64 columns convert per element)
    return type == BOWGPU_FLOAT64 ? __longlong_as_double((long long)b) : (double)(int64_t)b;
}

// ---------------------------------------------------------------- running state of a window
struct Stats {
    double sum;          // sum.go:16-22 / arithmeticmean.go:17-24
    double vmin, vmax;   // minmax.go:16-28 / :41-53 (seeded by the first valid value)
    double nn_min, nn_max;  // NaN-ignoring extrema, only used when partial states are merged
    uint64_t first_bits, last_bits;  // firstlast.go
    int64_t count;       // count.go / mean
    // integrals (integral.go): previous both-valid point and running sums
    double pt, pv, first_pt, first_pv;
    double integ_step, integ_trap;
    int has_value;
    int has_nn;
    int has_point;
    int has_pair;
};

__device__ __forceinline__ void stats_init(Stats &s) {
    s.sum = 0.0; s.vmin = 0.0; s.vmax = 0.0; s.nn_min = 0.0; s.nn_max = 0.0;
    s.first_bits = 0; s.last_bits = 0; s.count = 0;
    s.pt = 0.0; s.pv = 0.0; s.first_pt = 0.0; s.first_pv = 0.0; s.integ_step = 0.0; s.integ_trap = 0.0;
    s.has_value = 0; s.has_nn = 0; s.has_point = 0; s.has_pair = 0;
}

// one valid value, in row order
template <bool kMerge>
__device__ __forceinline__ void stats_value(Stats &s, double x, uint64_t raw) {
    s.sum += x;
    s.count++;
    if (s.has_value) {
        if (x < s.vmin) s.vmin = x;
        if (x > s.vmax) s.vmax = x;
    } else {
        s.vmin = x; s.vmax = x; s.first_bits = raw; s.has_value = 1;
    }
    s.last_bits = raw;
    if (kMerge) {
        if (x == x) {
            if (!s.has_nn) { s.nn_min = x; s.nn_max = x; s.has_nn = 1; }
            else { if (x < s.nn_min) s.nn_min = x; if (x > s.nn_max) s.nn_max = x; }
        }
    }
}

// one both-valid (t, v) point, in row order (integral.go:14-31, :46-62)
__device__ __forceinline__ void stats_point(Stats &s, double t, double v) {
    if (s.has_point) {
        s.integ_trap += (s.pv + v) / 2 * (t - s.pt);
        s.integ_step += s.pv * (t - s.pt);
        s.has_pair = 1;
    } else {
        s.first_pt = t; s.first_pv = v; s.has_point = 1;
    }
    s.pt = t; s.pv = v;
}

// merge R (later rows) into L (earlier rows): used by the cooperative long-window path only
__device__ __forceinline__ void stats_merge(Stats &L, const Stats &R) {
    if (R.has_point) {
        if (L.has_point) {
            L.integ_trap = L.integ_trap + (L.pv + R.first_pv) / 2 * (R.first_pt - L.pt) + R.integ_trap;
            L.integ_step = L.integ_step + L.pv * (R.first_pt - L.pt) + R.integ_step;
            L.has_pair = 1;
        } else {
            L.first_pt = R.first_pt; L.first_pv = R.first_pv; L.integ_trap = R.integ_trap;
            L.integ_step = R.integ_step; L.has_pair = R.has_pair; L.has_point = 1;
        }
        L.pt = R.pt; L.pv = R.pv;
    }
    if (!R.has_value) return;
    if (!L.has_value) {
        double isum = L.integ_step, itrap = L.integ_trap;  // keep merged integral fields
        double pt = L.pt, pv = L.pv, fpt = L.first_pt, fpv = L.first_pv;
        int hp = L.has_point, hpair = L.has_pair;
        L = R;
        L.integ_step = isum; L.integ_trap = itrap; L.pt = pt; L.pv = pv; L.first_pt = fpt; L.first_pv = fpv;
        L.has_point = hp; L.has_pair = hpair;
        return;
    }
    L.sum += R.sum;
    L.count += R.count;
    // minmax.go semantics on the concatenation: only R's non-NaN values can replace the seed
    if (R.has_nn) {
        if (R.nn_min < L.vmin) L.vmin = R.nn_min;
        if (R.nn_max > L.vmax) L.vmax = R.nn_max;
        if (!L.has_nn) { L.nn_min = R.nn_min; L.nn_max = R.nn_max; L.has_nn = 1; }
        else { if (R.nn_min < L.nn_min) L.nn_min = R.nn_min; if (R.nn_max > L.nn_max) L.nn_max = R.nn_max; }
    }
    L.last_bits = R.last_bits;
}

// ---------------------------------------------------------------- result of one reducer
struct Val {
    uint64_t bits;  // float64 or int64 payload
    int valid;      // 0 => nil
    int is_int;
};

// transformation.Factor chain (factor.go:7-20), then Buffer.SetOrDrop into a column of
// out_type (bowbuffer.go:60-80; bowconvert.go:24-29,:59-60)
__device__ __forceinline__ Val finish_val(Val v, const AggDesc &a) {
    if (!v.valid) { v.bits = 0; return v; }
    for (int f = 0; f < a.n_factors; f++) {
        if (v.is_int) v.bits = (uint64_t)go_f64_to_i64((double)(int64_t)v.bits * a.factors[f]);
        else v.bits = (uint64_t)__double_as_longlong(__longlong_as_double((long long)v.bits) * a.factors[f]);
    }
    if (a.out_type == BOWGPU_INT64 && !v.is_int) {
        v.bits = (uint64_t)go_f64_to_i64(__longlong_as_double((long long)v.bits));
        v.is_int = 1;
    } else if (a.out_type == BOWGPU_FLOAT64 && v.is_int) {
        v.bits = (uint64_t)__double_as_longlong((double)(int64_t)v.bits);
        v.is_int = 0;
    }
    return v;
}

// the same chain on a raw 8-byte result of the lean kernels (is_int: the result is an Int64)
__device__ __forceinline__ uint64_t apply_factors(uint64_t bits, bool is_int, int n, const double *f) {
    for (int k = 0; k < n; k++) {
        if (is_int) bits = (uint64_t)go_f64_to_i64((double)(int64_t)bits * f[k]);
        else bits = (uint64_t)__double_as_longlong(__longlong_as_double((long long)bits) * f[k]);
    }
    return bits;
}

__device__ __forceinline__ Val make_f64(double x) { Val v; v.bits = (uint64_t)__double_as_longlong(x); v.valid = 1; v.is_int = 0; return v; }
__device__ __forceinline__ Val make_i64(int64_t x) { Val v; v.bits = (uint64_t)x; v.valid = 1; v.is_int = 1; return v; }
__device__ __forceinline__ Val make_nil() { Val v; v.bits = 0; v.valid = 0; v.is_int = 0; return v; }

__device__ __forceinline__ bool kind_needs_inclusive(int kind) {
    return kind == BOWGPU_AGG_INTEGRAL_TRAPEZOID || kind == BOWGPU_AGG_WAVG_LINEAR;
}

// The value a reducer returns for a window, from the running state over its rows.
//   nrows      = w.Bow.NumRows() as THIS reducer sees it (after UnsetInclusive, aggregation.go:207-208)
//   s          = state over those rows
//   col_is_int = input column type (First/Last return the input type)
__device__ __forceinline__ Val reduce_val(int kind, const Stats &s, int64_t nrows, int64_t win_start,
                                          int64_t interval, int col_is_int) {
    switch (kind) {
    case BOWGPU_AGG_WINDOW_START: return make_i64(win_start);                       // windowstart.go:11
    case BOWGPU_AGG_NUM_ROWS: return make_f64((double)nrows);
    case BOWGPU_AGG_SUM: return make_f64(nrows == 0 ? 0.0 : s.sum);                 // sum.go:11-24
    case BOWGPU_AGG_MEAN:                                                          // arithmeticmean.go:11-29
        if (nrows == 0 || s.count == 0) return make_nil();
        return make_f64(s.sum / (double)s.count);
    case BOWGPU_AGG_MIN: return s.has_value ? make_f64(s.vmin) : make_nil();        // minmax.go:11-30
    case BOWGPU_AGG_MAX: return s.has_value ? make_f64(s.vmax) : make_nil();
    case BOWGPU_AGG_COUNT: return make_i64(s.count);                                // count.go:11-19
    case BOWGPU_AGG_FIRST:                                                         // firstlast.go:11-20
        if (!s.has_value) return make_nil();
        { Val v; v.bits = s.first_bits; v.valid = 1; v.is_int = col_is_int; return v; }
    case BOWGPU_AGG_LAST:
        if (!s.has_value) return make_nil();
        { Val v; v.bits = s.last_bits; v.valid = 1; v.is_int = col_is_int; return v; }
    case BOWGPU_AGG_INTEGRAL_STEP:                                                 // integral.go:43-68
    case BOWGPU_AGG_WAVG_STEP: {                                                   // weightedmean.go:11-19
        if (!s.has_point) return make_nil();
        int64_t last_value = win_start + interval;
        double r = s.integ_step + s.pv * ((double)last_value - s.pt);
        if (kind == BOWGPU_AGG_WAVG_STEP) r = r / (double)(last_value - win_start);
        return make_f64(r);
    }
    case BOWGPU_AGG_INTEGRAL_TRAPEZOID:                                            // integral.go:11-37
    case BOWGPU_AGG_WAVG_LINEAR: {                                                 // weightedmean.go:25-33
        if (!s.has_pair) return make_nil();
        double r = s.integ_trap;
        if (kind == BOWGPU_AGG_WAVG_LINEAR) r = r / (double)((win_start + interval) - win_start);
        return make_f64(r);
    }
    default: return make_nil();
    }
}

__device__ __forceinline__ bool kind_is_integral(int kind) {
    return kind >= BOWGPU_AGG_INTEGRAL_STEP && kind <= BOWGPU_AGG_WAVG_LINEAR;
}


// ---------------------------------------------------------------- the walk of the wave-tile kernels (rolling_simple.hip, rolling_tw.hip)
// A tile's values are staged in LDS as float64 with the NULL rows already replaced (see rolling_simple.hip "one pass per value
// column"), so the walk of one lane over one window is branch-free: one LDS read and one addition per row.

// v_min_f64 / v_max_f64 without the canonicalising copies LLVM puts in front of llvm.minnum / maxnum (IEEE mode: a signalling NaN
// would have to be quieted first).  What the instruction does with the inputs minmax.go never meets in this order - a NaN seed, a
// signalling NaN, zeros of both signs - is settled after the walk (walk_values), so the raw instruction is enough.
__device__ __forceinline__ double vmin64(double a, double b) { double d; asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ double vmax64(double a, double b) { double d; asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
constexpr uint64_t kNullAsNaN = 0x7FF8000000000000ull;   // what a null row holds in LDS while extrema are walked (x < mn and x > mx are false for it)
// a signalling NaN (v_cmp_class_f64, class bit 0).  The kernels run in IEEE mode: v_min_f64 / v_max_f64 return a signalling operand
// QUIETED, and the next step then drops that quiet NaN for the row after it - the running extremum is lost ([5, sNaN, 7]: 7, where
// minmax.go:22-27 keeps 5).  So the staging pass looks for one (one compare per row, all lanes busy) and a tile that holds one walks
// its extrema with the comparison itself (walk_values' `exact_mm`).
__device__ __forceinline__ bool is_snan(uint64_t bits) { return __builtin_amdgcn_class(__longlong_as_double((long long)bits), 1); }

// valid rows of the window [r0, r1) of a tile whose validity words (32 rows each, tile-relative) are vbits[]: how many, the first,
// the last (count.go:12-18, firstlast.go:11-35, the seed of minmax.go:16-21); fv = lv = -1 when there is none
__device__ __forceinline__ void window_valid_rows(const uint32_t *vbits, int r0, int r1, int &count, int &fv, int &lv) {
    count = 0; fv = -1; lv = -1;
    const int wl = (r1 - 1) >> 5;
    for (int w = r0 >> 5; w <= wl; w++) {
        uint32_t m = vbits[w];
        if (w == (r0 >> 5)) m &= 0xFFFFFFFFu << (r0 & 31);
        if (w == wl) m &= 0xFFFFFFFFu >> (31 - ((r1 - 1) & 31));
        count += __popc(m);
        if (m) {
            if (fv < 0) fv = w * 32 + __ffs((int)m) - 1;
            lv = w * 32 + 31 - __clz((int)m);
        }
    }
}

// Where row r of a tile lives in the staged array.  kSwz: two pad slots per 32 rows.  One lane walks one window, so the lanes of a
// wavefront read rows that lie one window length apart - and REGULAR windows of 16 / 32 / 64 / 128 rows put every lane on the same
// LDS bank (a stride of 256 B and its multiples): at 64 rows per window the bank-conflict cycles were 60 % of a wavefront's life
// (SQ_LDS_BANK_CONFLICT; Mean 0.67 of the HBM peak at 64 rows per window, 0.79 at 100).  With the pads a stride of 32 k rows becomes 34 k
// slots: 64-row windows land 8 banks apart.  A group of four rows that starts on a multiple of four never straddles a pad, so the
// walk reads its groups as before after at most three single steps; pairs (2 i, 2 i + 1) stay 16-byte aligned for the staging stores.
template <bool kSwz> __device__ __forceinline__ int swz(int r) { return kSwz ? r + 2 * (r >> 5) : r; }
constexpr int swz_slots(int rows) { return rows + 2 * (rows / 32); }

// rows r .. rend-1 of the staged column in order: eight LDS reads in flight (kEight: two groups of four, each on its own - a group that
// starts on a multiple of four never straddles a pad; the instantiations short of registers keep four), the additions in row order.  kSum / kMM: what is accumulated - one loop per
// combination, chosen by a uniform branch: a walk instruction issues for the whole wavefront however few lanes still have rows, and
// at 64 rows per window that made the walk half of a tile's instructions (sum + extrema in one loop with selects: 5.75 vector
// instructions per row; now 1.9 for a sum, 2.9 for extrema, 3.9 for both).
template <bool kSwz, bool kEight, bool kSum, bool kMM>
__device__ __forceinline__ void walk_rows(const uint64_t *val, int r, const int rend, double &sum, double &mn, double &mx) {
    auto one = [&](int rr) {
        const double x = __longlong_as_double((long long)val[swz<kSwz>(rr)]);
        if (kSum) sum += x;
        if (kMM) { mn = vmin64(mn, x); mx = vmax64(mx, x); }
    };
    auto four = [&](double x0, double x1, double x2, double x3) {
        if (kSum) { sum += x0; sum += x1; sum += x2; sum += x3; }
        if (kMM) {
            mn = vmin64(vmin64(vmin64(vmin64(mn, x0), x1), x2), x3);
            mx = vmax64(vmax64(vmax64(vmax64(mx, x0), x1), x2), x3);
        }
    };
    if (kSwz) for (; r < rend && (r & 3); r++) one(r);
    if (kEight) for (; r + 8 <= rend; r += 8) {
        const uint64_t *g = val + swz<kSwz>(r), *h = val + swz<kSwz>(r + 4);
        const double x0 = __longlong_as_double((long long)g[0]), x1 = __longlong_as_double((long long)g[1]);
        const double x2 = __longlong_as_double((long long)g[2]), x3 = __longlong_as_double((long long)g[3]);
        const double x4 = __longlong_as_double((long long)h[0]), x5 = __longlong_as_double((long long)h[1]);
        const double x6 = __longlong_as_double((long long)h[2]), x7 = __longlong_as_double((long long)h[3]);
        four(x0, x1, x2, x3);
        four(x4, x5, x6, x7);
    }
    for (; r + 4 <= rend; r += 4) {     // (kEight: at most once)
        const uint64_t *g = val + swz<kSwz>(r);
        four(__longlong_as_double((long long)g[0]), __longlong_as_double((long long)g[1]), __longlong_as_double((long long)g[2]),
             __longlong_as_double((long long)g[3]));
    }
    for (; r < rend; r++) one(r);
}

// rows fv .. lv of the staged column (sum.go:16-22, arithmeticmean.go:17-24, minmax.go:16-28).  Extrema: v_min_f64 / v_max_f64 from the
// seed (row fv) on.  The instruction differs from
// `if x < mn { mn = x }` in three cases only - the seed is a NaN (minmax.go keeps it: no value compares below a NaN; the instruction
// drops it), a signalling NaN among the values (the instruction returns it quieted and the step after it then drops the running
// extremum: the RESULT does not show it, so the staging pass reports it - `exact_mm`, see is_snan), and a result of zero (+0 and -0
// are equal for minmax.go, so the EARLIEST zero stays; the instruction orders them) - and in those the window is walked again with
// the comparison itself.  Null rows hold +0.0 when sums are walked and a quiet NaN when extrema are.
template <bool kSwz, bool kEight>
__device__ __forceinline__ void walk_values(const uint64_t *val, int fv, int lv, bool do_sum, bool do_mm, bool exact_mm, double &sum, double &mn, double &mx) {
    const double seed = __longlong_as_double((long long)val[swz<kSwz>(fv)]);
    sum = 0.0; mn = seed; mx = seed;
    const int rend = lv + 1;
    if (kEight) {
        if (do_sum && do_mm) walk_rows<kSwz, true, true, true>(val, fv, rend, sum, mn, mx);
        else if (do_mm) walk_rows<kSwz, true, false, true>(val, fv, rend, sum, mn, mx);
        else if (do_sum) walk_rows<kSwz, true, true, false>(val, fv, rend, sum, mn, mx);
    } else if (do_sum || do_mm) {
        // (the instantiations with several value columns sit at their register limit: one loop of four rows, the two kinds under a flag)
        int r = fv;
        auto one = [&](int rr) {
            const double x = __longlong_as_double((long long)val[swz<kSwz>(rr)]);
            if (do_sum) sum += x;
            if (do_mm) { mn = vmin64(mn, x); mx = vmax64(mx, x); }
        };
        if (kSwz) for (; r < rend && (r & 3); r++) one(r);
        for (; r + 4 <= rend; r += 4) {
            const uint64_t *g = val + swz<kSwz>(r);
            const double x0 = __longlong_as_double((long long)g[0]), x1 = __longlong_as_double((long long)g[1]);
            const double x2 = __longlong_as_double((long long)g[2]), x3 = __longlong_as_double((long long)g[3]);
            if (do_sum) { sum += x0; sum += x1; sum += x2; sum += x3; }
            if (do_mm) {
                mn = vmin64(vmin64(vmin64(vmin64(mn, x0), x1), x2), x3);
                mx = vmax64(vmax64(vmax64(vmax64(mx, x0), x1), x2), x3);
            }
        }
        for (; r < rend; r++) one(r);
    }
    if (do_mm) {
        if (seed != seed) { mn = seed; mx = seed; }
        else if (exact_mm || mn == 0.0 || mx == 0.0 || mn != mn || mx != mx) {
            mn = seed; mx = seed;
            for (int rr = fv + 1; rr < rend; rr++) {
                const double x = __longlong_as_double((long long)val[swz<kSwz>(rr)]);
                if (x < mn) mn = x;
                if (x > mx) mx = x;
            }
        }
    }
}

// the same for a nullable column staged with +0.0 in its null rows when sums AND extrema are wanted and the tile holds many short
// windows: one walk, the extrema under the row's validity bit (minmax.go:16-28 as written).  Tiles of few long windows walk twice
// instead - sums, then extrema over NaN-filled nulls with walk_values - which costs a second pass over the windows but 5 instead of
// 11 instructions per row.
constexpr int kTwoWalksMaxHeads = 24;   // (640 rows / 24: windows of ~27 rows and more)
constexpr int kWalkAllMaxHeads = 48;    // rolling_tw.hip: tiles with more heads (windows of < ~13 rows) walk every reducer in one pass
template <bool kSwz>
__device__ __forceinline__ void walk_values_pred(const uint64_t *val, const uint32_t *vbits, int fv, int lv, double &sum, double &mn, double &mx) {
    const double seed = __longlong_as_double((long long)val[swz<kSwz>(fv)]);
    sum = 0.0; mn = seed; mx = seed;
    for (int r = fv; r <= lv; r++) {
        const double x = __longlong_as_double((long long)val[swz<kSwz>(r)]);
        sum += x;
        if ((vbits[r >> 5] >> (r & 31)) & 1u) {
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
    }
}

// rows a .. b-1 of a staged array of terms added in order onto +0.0 (the integrals: integral.go:22-31, :48-62)
template <bool kSwz, bool kEight>
__device__ __forceinline__ double walk_terms(const uint64_t *t, int a, int b) {
    double acc = 0.0, unused0 = 0.0, unused1 = 0.0;
    walk_rows<kSwz, kEight, true, false>(t, a, b, acc, unused0, unused1);
    return acc;
}

// ---------------------------------------------------------------- the queue of long windows
// A tile queues at most one window (the one still open at the end of its look-ahead).  kLongLists sub-lists with their
// own counters (status[16 + s]) keep the appends off a single contended address: 1e5 appends to ONE counter cost ~1 ms.
__device__ __forceinline__ void push_long_window(uint32_t *status, int64_t *long_list, int64_t sub_cap, int64_t tile,
                                                 uint64_t wid, int64_t row) {
    const int64_t s = tile & (kLongLists - 1);
    const unsigned idx = atomicAdd(&status[kLongCountWord + s], 1u);
    if ((int64_t)idx < sub_cap) {
        int64_t *e = long_list + 2 * (s * sub_cap + (int64_t)idx);
        e[0] = (int64_t)wid;
        e[1] = row;
    } else {
        atomicOr(&status[2], 1u);
    }
}

// ---------------------------------------------------------------- loads
__device__ __forceinline__ void load_pair(const uint64_t *__restrict__ p, int64_t g, int64_t n, bool vec,
                                          uint64_t &a, uint64_t &b) {
    if (vec && g + 1 < n) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p + g);
        a = v.x; b = v.y;
    } else {
        a = g < n ? p[g] : 0;
        b = g + 1 < n ? p[g + 1] : 0;
    }
}

// 32 validity bits starting at logical row `row` of a column
__device__ __forceinline__ uint32_t load_vbits32(const ColDesc &c, int64_t row) {
    int64_t bit = c.vbit0 + row;
    int64_t wi = bit >> 5;
    int sh = (int)(bit & 31);
    uint32_t lo = wi < c.vwords ? c.vbits[wi] : 0u;
    if (sh == 0) return lo;
    uint32_t hi = wi + 1 < c.vwords ? c.vbits[wi + 1] : 0u;
    return (lo >> sh) | (hi << (32 - sh));
}


}  // namespace bowgpu
