// long_windows.hip — windows too long for one tile (more rows than a tile's look-ahead), down to ONE window over
// the whole frame.  The tile kernels queue such windows as (window id, first row); here they are reduced by a
// coalesced, multi-workgroup, ORDER-FREE formulation of the same reducers (reference rolling/aggregation/*.go):
//
//   Sum / Mean      sum and count of valid values (any association => Sum/Mean/Integral agree with the reference's
//                   left-to-right sum within 1e-12 relative; every other output is bit-exact)
//   Min / Max       minmax.go seeds with the FIRST valid value and replaces on strict < / >: equivalently
//                   "NaN iff the first valid value is NaN, else the extreme of the non-NaN values, ties (e.g. -0.0 /
//                   +0.0) resolved by the smallest row index"  -> (value, row index) pairs merge in any order
//   First / Last    smallest / largest valid row index
//   Integrals       integral.go walks consecutive both-valid points: each valid row contributes one term with its NEXT
//                   valid row (looked up forward, across chunk boundaries), so the terms sum in any order
//
// Launches: long_bounds (end row of each window by bisection + chunk counts) -> exclusive scan -> long_map ->
// long_partial (one wavefront per 4096-row chunk) -> long_final (merge in chunk order, outputs,
// empty windows after it).
#include <type_traits>
#include "agg_device.h"
#include "bitmap_device.h"

namespace bowgpu {

namespace {

constexpr int kChunkRows = kLongChunkRows;
constexpr int kStreamRows = kLongStreamRows;   // the streaming form's chunk (one wavefront)

struct LongEntry {
    uint64_t wid;
    int64_t r0, r1;
    uint64_t next_wid;
    int32_t incl_row;   // the row r1 sits exactly on the window's end and windows are inclusive
    int32_t dead;       // window 0 made only of rows below s0 (an empty slice in the reference)
};

struct Part {
    double sum, trap, step;
    double vmin, vmax;
    int64_t count;
    int64_t min_idx, max_idx;     // -1: no non-NaN value
    int64_t first_idx, last_idx;  // -1: no valid value
};

__device__ __forceinline__ void part_init(Part &p) {
    p.sum = 0.0; p.trap = 0.0; p.step = 0.0; p.vmin = 0.0; p.vmax = 0.0; p.count = 0;
    p.min_idx = -1; p.max_idx = -1; p.first_idx = -1; p.last_idx = -1;
}

__device__ __forceinline__ void part_merge(Part &a, const Part &b) {
    a.sum += b.sum; a.trap += b.trap; a.step += b.step; a.count += b.count;
    if (b.min_idx >= 0 && (a.min_idx < 0 || b.vmin < a.vmin || (b.vmin == a.vmin && b.min_idx < a.min_idx))) { a.vmin = b.vmin; a.min_idx = b.min_idx; }
    if (b.max_idx >= 0 && (a.max_idx < 0 || b.vmax > a.vmax || (b.vmax == a.vmax && b.max_idx < a.max_idx))) { a.vmax = b.vmax; a.max_idx = b.max_idx; }
    if (b.first_idx >= 0 && (a.first_idx < 0 || b.first_idx < a.first_idx)) a.first_idx = b.first_idx;
    if (b.last_idx > a.last_idx) a.last_idx = b.last_idx;
}

__device__ __forceinline__ bool col_valid(const ColDesc &cd, int64_t r) {
    if (!cd.vbits) return true;
    const int64_t bit = cd.vbit0 + r;
    return (cd.vbits[bit >> 5] >> (bit & 31)) & 1u;
}

// next valid row of the column in [r, lim), or -1
__device__ __forceinline__ int64_t col_next_valid(const ColDesc &cd, int64_t r, int64_t lim) {
    if (r >= lim) return -1;
    if (!cd.vbits) return r;
    int64_t b = cd.vbit0 + r;
    const int64_t bend = cd.vbit0 + lim;
    while (b < bend) {
        const int64_t w = b >> 5;
        const uint32_t x = cd.vbits[w] & (~0u << (b & 31));
        if (x) {
            const int64_t rr = (w << 5) + (__ffs((int)x) - 1) - cd.vbit0;
            return rr < lim ? rr : -1;
        }
        b = (w + 1) << 5;
    }
    return -1;
}

__device__ __forceinline__ bool slot_needs(const AggParams &p, int slot, bool *need_ts) {
    const unsigned m = p.pass_mask[slot + 1];
    *need_ts = false;
    for (unsigned mm = m; mm; mm &= mm - 1) {
        const int k = p.aggs[__ffs(mm) - 1].kind;
        if (k >= BOWGPU_AGG_INTEGRAL_STEP && k <= BOWGPU_AGG_WAVG_LINEAR) *need_ts = true;
    }
    return (p.pass_flags[slot + 1] & kPassNeedVals) != 0;
}

// workgroup (256 threads) reduction in a fixed shape: lanes by shuffle, then the four waves in order; result in thread 0
__device__ __forceinline__ void block_reduce(Part &acc, Part *red /* [4], LDS */, int tid) {
    for (int o = 32; o > 0; o >>= 1) {
        Part other;
        other.sum = __shfl_down(acc.sum, o); other.trap = __shfl_down(acc.trap, o); other.step = __shfl_down(acc.step, o);
        other.vmin = __shfl_down(acc.vmin, o); other.vmax = __shfl_down(acc.vmax, o);
        other.count = __shfl_down((long long)acc.count, o);
        other.min_idx = __shfl_down((long long)acc.min_idx, o); other.max_idx = __shfl_down((long long)acc.max_idx, o);
        other.first_idx = __shfl_down((long long)acc.first_idx, o); other.last_idx = __shfl_down((long long)acc.last_idx, o);
        if ((tid & 63) + o < 64) part_merge(acc, other);
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        acc = red[0];
        part_merge(acc, red[1]); part_merge(acc, red[2]); part_merge(acc, red[3]);
    }
}

// partial `i` of an array of Part, or - lite - of {sum, count} pairs (what Sum / Mean / Count need: the streaming form's
// storage when no reducer of the call wants more)
struct PartLite { double sum; int64_t count; };
__device__ __forceinline__ Part part_at(const Part *base, int64_t i, int lite) {
    if (!lite) return base[i];
    const PartLite l = reinterpret_cast<const PartLite *>(base)[i];
    Part q;
    part_init(q);
    q.sum = l.sum; q.count = l.count;
    return q;
}

// window id of a row's timestamp (rows below s0 ride in window 0: SURVEY A.5)
__device__ __forceinline__ uint64_t row_wid(const AggParams &p, int64_t t) {
    return (p.pre_rows && t < p.s0) ? 0ull : magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
}

// wid, r0 and r1 known: what follows the window (rolling.go:197-209) - the next non-empty window, whether the row at r1 sits
// exactly on the window's end (an inclusive window takes it), whether window 0 is made of rows below s0 only
__device__ __forceinline__ void entry_close(const AggParams &p, LongEntry &le) {
    uint64_t next_wid = (uint64_t)(p.wid_base + p.W);
    bool at_start = false;
    if (le.r1 < p.n) {
        const int64_t t = p.ts[le.r1];
        next_wid = magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
        at_start = (t == p.s0 + (int64_t)(next_wid * (uint64_t)p.interval));
    }
    le.next_wid = next_wid;
    le.incl_row = (p.inclusive && le.r1 < p.n && at_start && next_wid == le.wid + 1) ? 1 : 0;
    le.dead = (p.pre_rows && le.r0 == 0 && !(p.ts[le.r1 - 1] >= p.s0 || le.incl_row)) ? 1 : 0;
}

// first row >= lo whose timestamp reaches the end of window `wid` (p.n when none), by bisection in global memory
__device__ __forceinline__ int64_t window_end_row(const AggParams &p, uint64_t wid, int64_t lo) {
    const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
    const int64_t lim = win_start + p.interval;
    if (lim < win_start) return p.n;  // int64 overflow: no row can reach it
    int64_t hi = p.n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (p.ts[mid] >= lim) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ void emit_stats(const AggParams &p, int slot, const LongEntry &w, const Stats &st, uint32_t *valid_out);

// The outputs of one column pass (slot; -1: the reducers that need no column) for window `w`, from its merged order-free partial
// (valid_out != nullptr: the validity bitmaps are not touched; bit i of *valid_out is set when the output of aggregation i is valid -
// the caller assembles whole bitmap words from its lanes)
__device__ __forceinline__ void emit_window(const AggParams &p, int slot, const LongEntry &w, const Part &acc, uint32_t *valid_out = nullptr) {
    const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
    const int col_type = cd ? cd->type : BOWGPU_INT64;
    const uint64_t *vp = cd ? reinterpret_cast<const uint64_t *>(cd->values) : nullptr;
    const int64_t oslot = (int64_t)(w.wid - (uint64_t)p.wid_base);
    if ((uint64_t)oslot >= (uint64_t)p.W) return;
    // rebuild the reference's running state from the order-free partial
    Stats st;
    stats_init(st);
    st.sum = acc.sum; st.count = acc.count; st.has_value = acc.first_idx >= 0;
    if (st.has_value) {
        st.first_bits = vp[acc.first_idx]; st.last_bits = vp[acc.last_idx];
        const double f = bits_to_f64(st.first_bits, col_type);
        // minmax.go:16-28: seeded by the first valid value; a NaN seed is never replaced
        st.vmin = (f != f) ? f : (acc.min_idx >= 0 ? acc.vmin : f);
        st.vmax = (f != f) ? f : (acc.max_idx >= 0 ? acc.vmax : f);
        st.has_point = 1; st.pt = (double)p.ts[acc.last_idx]; st.pv = bits_to_f64(st.last_bits, col_type);
        st.integ_trap = acc.trap; st.integ_step = acc.step; st.has_pair = acc.count >= 2;
    }
    emit_stats(p, slot, w, st, valid_out);
}

// ... from the reference's running state itself (emit_window rebuilds it from an order-free partial; long_strict_kernel arrives
// with the state of a walk in row order)
__device__ __forceinline__ void emit_stats(const AggParams &p, int slot, const LongEntry &w, const Stats &st, uint32_t *valid_out) {
    const unsigned my_mask = p.pass_mask[slot + 1];
    const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
    const int col_type = cd ? cd->type : BOWGPU_INT64;
    const bool need_vals = cd && (p.pass_flags[slot + 1] & kPassNeedVals);
    const uint64_t *vp = cd ? reinterpret_cast<const uint64_t *>(cd->values) : nullptr;
    const int64_t win_start = p.s0 + (int64_t)(w.wid * (uint64_t)p.interval);
    const int64_t oslot = (int64_t)(w.wid - (uint64_t)p.wid_base);
    const int64_t len = w.dead ? 0 : w.r1 - w.r0;
    if ((uint64_t)oslot >= (uint64_t)p.W) return;
    // the state including the inclusive row, for the reducers that want it (aggregation.go:207-211)
    Stats st_incl = st;
    if (w.incl_row && need_vals && col_valid(*cd, w.r1)) {
        const uint64_t raw = vp[w.r1];
        const double x = bits_to_f64(raw, col_type);
        stats_value<false>(st_incl, x, raw);
        stats_point(st_incl, (double)p.ts[w.r1], x);
    }
    for (unsigned m = my_mask; m; m &= m - 1) {
        const AggDesc &a = p.aggs[__ffs(m) - 1];
        const bool inc = a.kind == BOWGPU_AGG_INTEGRAL_TRAPEZOID || a.kind == BOWGPU_AGG_WAVG_LINEAR;
        Val v = finish_val(reduce_val(a.kind, inc ? st_incl : st, inc ? len + w.incl_row : len, win_start, p.interval,
                                      col_type == BOWGPU_INT64), a);
        reinterpret_cast<uint64_t *>(a.out_values)[oslot] = v.bits;
        if (valid_out) { if (v.valid) *valid_out |= 1u << (__ffs(m) - 1); }
        else if (a.out_valid) {
            if (v.valid) atomicOr(&a.out_valid[oslot >> 5], 1u << (oslot & 31));
            else if (p.bits_preset) atomicAnd(&a.out_valid[oslot >> 5], ~(1u << (oslot & 31)));
        }
    }
}

// the empty windows gk = first, first + step, ... <= gap behind window `w`, same pass
__device__ __forceinline__ void emit_empties(const AggParams &p, int slot, const LongEntry &w, int64_t gap, int64_t first, int64_t step) {
    const unsigned my_mask = p.pass_mask[slot + 1];
    const int col_type = slot >= 0 ? p.cols[slot].type : BOWGPU_INT64;
    const int64_t win_start = p.s0 + (int64_t)(w.wid * (uint64_t)p.interval);
    const int64_t oslot = (int64_t)(w.wid - (uint64_t)p.wid_base);
    Stats em;
    stats_init(em);
    for (int64_t gk = first; gk <= gap; gk += step) {
        const int64_t gs = oslot + gk;
        if (gs < 0 || gs >= p.W) break;
        const int64_t gstart = win_start + gk * p.interval;
        for (unsigned m = my_mask; m; m &= m - 1) {
            const AggDesc &a = p.aggs[__ffs(m) - 1];
            Val v = finish_val(reduce_val(a.kind, em, 0, gstart, p.interval, col_type == BOWGPU_INT64), a);
            reinterpret_cast<uint64_t *>(a.out_values)[gs] = v.bits;
            if (p.bits_preset && a.out_valid && !v.valid) atomicAnd(&a.out_valid[gs >> 5], ~(1u << (gs & 31)));
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256) void long_bounds_kernel(const AggParams p, const LongListStarts starts, LongEntry *entries,
                                                          int32_t *nchunks) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= starts.start[kLongLists]) return;
    int sub = 0;
    while (starts.start[sub + 1] <= e) sub++;
    const int64_t *item = p.long_list + 2 * (sub * p.long_cap + (e - starts.start[sub]));
    const uint64_t wid = (uint64_t)item[0];
    const int64_t r0 = item[1];
    const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
    // first row >= r0 with ts >= win_start + interval (rolling.go:197-209), by bisection
    const int64_t lim = win_start + p.interval;
    const bool ovf = lim < win_start;  // int64 overflow: no row can reach it
    int64_t lo = r0 + 1, hi = p.n;
    while (lo < hi && !ovf) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (p.ts[mid] >= lim) hi = mid; else lo = mid + 1;
    }
    const int64_t r1 = ovf ? p.n : lo;
    LongEntry le;
    le.wid = wid; le.r0 = r0; le.r1 = r1;
    entry_close(p, le);
    entries[e] = le;
    nchunks[e] = (int32_t)((r1 - r0 + kChunkRows - 1) / kChunkRows);
}

// Every window of the call as an entry (the "long-only" pipeline: when windows average thousands of rows the tile kernels
// would read every row just to queue almost every window here, so the host skips them): r0 / r1 by bisection.
__global__ __launch_bounds__(256) void long_bounds_all_kernel(const AggParams p, LongEntry *entries, int32_t *nchunks) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.W) return;
    const uint64_t wid = (uint64_t)(p.wid_base + k);
    const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
    auto lower_bound = [&](int64_t v) {
        int64_t lo = 0, hi = p.n;
        while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (p.ts[mid] >= v) hi = mid; else lo = mid + 1; }
        return lo;
    };
    const int64_t lim = win_start + p.interval;
    const int64_t r0 = wid == 0 ? 0 : lower_bound(win_start);  // rows below s0 ride in window 0
    const int64_t r1 = lim < win_start ? p.n : lower_bound(lim);
    LongEntry le;
    le.wid = wid; le.r0 = r0; le.r1 = r1; le.next_wid = wid + 1;  // (empty windows are entries of their own: no run to fill behind)
    le.incl_row = (p.inclusive && r1 < p.n && p.ts[r1] == lim && lim > win_start) ? 1 : 0;
    le.dead = (p.pre_rows && wid == 0 && !((r1 > 0 && p.ts[r1 - 1] >= p.s0) || le.incl_row)) ? 1 : 0;
    entries[k] = le;
    nchunks[k] = (int32_t)((r1 - r0 + kChunkRows - 1) / kChunkRows);
}

// status[0] |= 1 when the interval column is not ascending (the tile kernels check this on their way; the long-only pipeline
// has no tile kernel)
__global__ __launch_bounds__(256) void ts_sorted_kernel(const int64_t *__restrict__ ts, int64_t n, uint32_t *status) {
    bool bad = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += stride) {
        const int64_t a = ts[i], b = i + 1 < n ? ts[i + 1] : a, l = i > 0 ? ts[i - 1] : a;
        bad |= l > a || a > b;
    }
    if (bad) atomicOr(&status[0], 1u);
}

// bowgpu_options.strict_order for the windows no tile holds: one LANE per window walks its rows in the reference's order - every
// reducer through the same running state the reference keeps (sum.go:16-22, minmax.go:16-28, integral.go:14-31, :46-62), so Sum /
// Mean / Integral* / WeightedAverage* come out bit for bit - and writes its outputs and the empty windows behind it.  Neighbouring
// lanes read neighbouring windows: a load touches 64 cache lines, but the next fifteen rows of each lane come out of the same lines
// (L1 / L2), so the rows are still fetched from HBM once.  Four rows of a lane are loaded before the first is consumed.  Windows
// longer than kStrictMaxRows are not walked (status[7]: the call is declined) - a single lane would take milliseconds per window.
constexpr int64_t kStrictMaxRows = 1ll << 20;
// one lane, one window: the walk in row order, the window's outputs, the empty windows behind it (fill_gaps)
// check_order (the long-only strict form, when some column pass reads the timestamps anyway): the walk of the first such pass also checks
// that the window's rows ascend - its own rows pair by pair and its first row against the row in front of it - and that the windows'
// row ranges are what bisection gives on an ascending column (r0 <= r1, the first window starts at row 0, the last one ends at row n):
// together, every adjacent pair of rows of the frame.  That replaces ts_sorted_kernel's pass over the whole interval column (0.16 ms per
// 1e8 rows in front of a 0.47 ms walk).
__device__ __forceinline__ void walk_entry(const AggParams &p, const LongEntry &le, const int fill_gaps, const int check_order = 0) {
    const int64_t gap = fill_gaps ? (int64_t)(le.next_wid - le.wid) - 1 : 0;
    bool order_pending = check_order != 0, bad = false;
    if (check_order) {
        bad = le.r1 < le.r0 || (le.wid == (uint64_t)p.wid_base && le.r0 != 0) || (le.wid == (uint64_t)(p.wid_base + p.W - 1) && le.r1 != p.n);
    }
    for (int slot = -1; slot < p.ncols; slot++) {
        if (p.pass_mask[slot + 1] == 0) continue;
        Stats st;
        stats_init(st);
        bool need_ts = false;
        if (slot >= 0 && slot_needs(p, slot, &need_ts) && !le.dead) {
            const ColDesc &cd = p.cols[slot];
            const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd.values);
            const uint64_t *tp = reinterpret_cast<const uint64_t *>(p.ts);
            const bool chk = order_pending && need_ts && !bad;
            if (chk) order_pending = false;
            int64_t last_t = (chk && le.r0 > 0) ? p.ts[le.r0 - 1] : INT64_MIN;
            auto one = [&](bool ok, uint64_t raw, uint64_t traw) {
                if (chk) { bad = bad || (int64_t)traw < last_t; last_t = (int64_t)traw; }
                if (!ok) return;
                const double x = bits_to_f64(raw, cd.type);
                stats_value<false>(st, x, raw);
                if (need_ts) stats_point(st, (double)(int64_t)traw, x);
            };
            // eight rows of the lane are loaded (values, timestamps when an integral wants them, the validity bits as one or two words)
            // before the first is consumed: the chain of additions is the only thing that has to wait
            int64_t r = le.r0;
            // 16-byte loads from an even row on (a lane's eight rows are one 64-byte line of each column: four load instructions per
            // column instead of eight - neighbouring lanes read lines far apart, so what a load instruction costs is its count of lines:
            // 1000-row windows, WeightedAverageStep, 0.706 -> 0.602 ms per 1e8 rows.  A second batch of eight rows in flight behind the
            // first - tried in round 6 - made it 0.652.  Also tried and not kept: the wavefront fetching its 64 windows' lines TOGETHER - four
            // neighbouring lanes per line, every line asked for once, 16 lines per load instruction - and handing each window's lane its
            // eight rows through LDS: every test green, 0.470 -> 0.757 ms for the same call, Mean 0.43 -> 0.61 (the load, the LDS round trip
            // and the chain of additions then run strictly one after the other in a kernel that has one or two wavefronts per SIMD))
            const bool vec = ((reinterpret_cast<uintptr_t>(vp) | (need_ts ? reinterpret_cast<uintptr_t>(tp) : 0)) & 15) == 0;
            if (vec && (r & 1) && r < le.r1) { one(col_valid(cd, r), vp[r], need_ts ? tp[r] : 0ull); r++; }
            for (; r + 8 <= le.r1; r += 8) {
                uint64_t q[8], t[8];
                if (vec) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(vp + r + 2 * i); q[2 * i] = x.x; q[2 * i + 1] = x.y; }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) q[i] = vp[r + i];
                }
                if (need_ts && vec) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(tp + r + 2 * i); t[2 * i] = x.x; t[2 * i + 1] = x.y; }
                } else if (need_ts) {
#pragma unroll
                    for (int i = 0; i < 8; i++) t[i] = tp[r + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) t[i] = 0;
                }
                uint32_t bits = 0xFFu;
                if (cd.vbits) {
                    const int64_t bit = cd.vbit0 + r;
                    const uint32_t lo = cd.vbits[bit >> 5];
                    const int sh = (int)(bit & 31);
                    bits = lo >> sh;
                    if (sh > 24) bits |= cd.vbits[(bit >> 5) + 1] << (32 - sh);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) one((bits >> i) & 1u, q[i], t[i]);
            }
            for (; r < le.r1; r++) one(col_valid(cd, r), vp[r], need_ts ? tp[r] : 0ull);
        }
        emit_stats(p, slot, le, st, nullptr);
        if (gap > 0) emit_empties(p, slot, le, gap, 1, 1);
    }
    if (check_order && (bad || order_pending)) {   // (order_pending: no pass of this window read the timestamps - the host only asks when one does)
        if (!__hip_atomic_load(&p.status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[0], 1u);
    }
}

__global__ __launch_bounds__(256) void long_strict_kernel(const AggParams p, const int64_t n_long, const LongEntry *entries, const int fill_gaps, const int check_order) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_long) return;
    const LongEntry le = entries[e];
    if (le.r1 - le.r0 > kStrictMaxRows) {
        if (!__hip_atomic_load(&p.status[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[7], 1u);
        return;
    }
    walk_entry(p, le, fill_gaps, check_order);
}

// The queued windows of a tile pass served WITHOUT the host in between (reference rolling/aggregation.go:190-238 is one loop): launched
// behind the tile kernel with a grid sized from the queue's CAPACITY; how many windows were queued is read here, from the sub-lists'
// counters in the status block.  One lane per queued window: its end row (galloping from its first row - these windows are a few
// hundred rows long, not the log2(n) probes of a bisection over the frame), then by its LENGTH either the walk in row order right
// here (exact, like long_strict_kernel: up to walk_max_rows rows) or an entry of the compact list `big` for the chunked order-free
// machinery, which the host runs only when status[kQueueBigWord] comes back non-zero.  strict != 0: nothing goes to the list - a window
// beyond walk_max_rows raises status[7] (the call is declined).
__global__ __launch_bounds__(256) void long_queue_kernel(const AggParams p, LongEntry *big, int32_t *big_nchunks, const int64_t walk_max_rows,
                                                         const int strict) {
    __shared__ int64_t s_start[kLongLists + 1];
    if (threadIdx.x < 64) {   // prefix sums of the 64 sub-list counters (every workgroup: 64 loads, one wave scan)
        const uint32_t cnt = p.status[kLongCountWord + threadIdx.x];
        uint32_t incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up;
        }
        s_start[threadIdx.x + 1] = (int64_t)incl;
        if (threadIdx.x == 0) s_start[0] = 0;
    }
    __syncthreads();
    const int64_t n_long = s_start[kLongLists];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_long) return;
    int sub = 0;
    while (s_start[sub + 1] <= e) sub++;
    const int64_t *item = p.long_list + 2 * (sub * p.long_cap + (e - s_start[sub]));
    LongEntry le;
    le.wid = (uint64_t)item[0];
    le.r0 = item[1];
    const int64_t win_start = p.s0 + (int64_t)(le.wid * (uint64_t)p.interval);
    const int64_t lim = win_start + p.interval;
    int64_t r1 = p.n;
    if (lim >= win_start) {   // (else int64 overflow: no row can reach the window's end)
        // gallop: the first probe at the look-ahead the window outgrew, doubling until a row beyond the window is found
        int64_t lo = le.r0 + 1, step = 128, hi = p.n;
        for (;;) {
            const int64_t probe = le.r0 + step;
            if (probe >= p.n) break;
            if (p.ts[probe] >= lim) { hi = probe; break; }
            lo = probe + 1;
            step <<= 1;
        }
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (p.ts[mid] >= lim) hi = mid; else lo = mid + 1;
        }
        r1 = lo;
    }
    le.r1 = r1;
    entry_close(p, le);
    if (le.r1 - le.r0 <= walk_max_rows) {
        walk_entry(p, le, 1);
    } else if (strict) {
        if (!__hip_atomic_load(&p.status[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[7], 1u);
    } else {
        const uint32_t at = atomicAdd(&p.status[kQueueBigWord], 1u);
        big[at] = le;
        big_nchunks[at] = (int32_t)((le.r1 - le.r0 + kChunkRows - 1) / kChunkRows);
    }
}

// chunk -> window map: work_entry[w] = e for the chunks [offsets[e], offsets[e+1]) of queued window e
__global__ __launch_bounds__(256) void long_map_kernel(const int64_t n_long, const int64_t *offsets, int32_t *work_entry) {
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wavefront per window
    if (e >= n_long) return;
    for (int64_t w = offsets[e] + (threadIdx.x & 63); w < offsets[e + 1]; w += 64) work_entry[w] = (int32_t)e;
}

// one workgroup per chunk of one long window; partials[(work * ncols) + slot]
// one WAVEFRONT per chunk of one window (four independent wavefronts per workgroup, no barrier): lanes stride the chunk's rows,
// four rows per lane in flight per trip; partials[(work * ncols) + slot]
__global__ __launch_bounds__(256) void long_partial_kernel(const AggParams p, const int64_t n_long, const LongEntry *entries,
                                                           const int64_t *offsets /* n_long + 1 */, const int32_t *work_entry,
                                                           Part *partials) {
    const int lane = threadIdx.x & 63;
    const int64_t work = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (work >= offsets[n_long]) return;
    const int64_t s_e = work_entry[work];
    const LongEntry le = entries[s_e];
    const int64_t c0 = le.r0 + (work - offsets[s_e]) * kChunkRows;
    const int64_t c1 = (c0 + kChunkRows < le.r1) ? c0 + kChunkRows : le.r1;

    for (int slot = 0; slot < p.ncols; slot++) {
        bool need_ts;
        const bool need_vals = slot_needs(p, slot, &need_ts);
        Part acc;
        part_init(acc);
        if (need_vals && !le.dead) {
            const ColDesc &cd = p.cols[slot];
            const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd.values);
            auto one = [&](int64_t r, uint64_t raw) {
                const double x = bits_to_f64(raw, cd.type);
                acc.sum += x;
                acc.count++;
                if (acc.first_idx < 0) acc.first_idx = r;
                acc.last_idx = r;
                if (x == x) {
                    if (acc.min_idx < 0 || x < acc.vmin) { acc.vmin = x; acc.min_idx = r; }
                    if (acc.max_idx < 0 || x > acc.vmax) { acc.vmax = x; acc.max_idx = r; }
                }
                if (need_ts) {  // the term of the pair (this point, next both-valid point inside the window)
                    const int64_t rn = col_next_valid(cd, r + 1, le.r1);
                    if (rn >= 0) {
                        const double t0 = (double)p.ts[r], t1 = (double)p.ts[rn];
                        const double x1 = bits_to_f64(vp[rn], cd.type);
                        acc.trap += (x + x1) / 2 * (t1 - t0);   // integral.go:24
                        acc.step += x * (t1 - t0);              // integral.go:55
                    }
                }
            };
            int64_t r = c0 + lane;
            for (; r + 192 < c1; r += 256) {  // four independent loads per lane before any of them is consumed
                const uint64_t q0 = vp[r], q1 = vp[r + 64], q2 = vp[r + 128], q3 = vp[r + 192];
                if (col_valid(cd, r)) one(r, q0);
                if (col_valid(cd, r + 64)) one(r + 64, q1);
                if (col_valid(cd, r + 128)) one(r + 128, q2);
                if (col_valid(cd, r + 192)) one(r + 192, q3);
            }
            for (; r < c1; r += 64)
                if (col_valid(cd, r)) one(r, vp[r]);
        }
        for (int o = 32; o > 0; o >>= 1) {  // wavefront reduction in a fixed shape
            Part other;
            other.sum = __shfl_down(acc.sum, o); other.trap = __shfl_down(acc.trap, o); other.step = __shfl_down(acc.step, o);
            other.vmin = __shfl_down(acc.vmin, o); other.vmax = __shfl_down(acc.vmax, o);
            other.count = __shfl_down((long long)acc.count, o);
            other.min_idx = __shfl_down((long long)acc.min_idx, o); other.max_idx = __shfl_down((long long)acc.max_idx, o);
            other.first_idx = __shfl_down((long long)acc.first_idx, o); other.last_idx = __shfl_down((long long)acc.last_idx, o);
            if (lane + o < 64) part_merge(acc, other);
        }
        if (lane == 0) partials[work * p.ncols + slot] = acc;
    }
}

// One LANE per long window: the usual long window has a handful of chunk partials and no run of empty windows behind it, and
// finishing it (rebuild the reference's running state from the order-free partial, evaluate the reducers, write the slot) is
// work for one lane - 64 windows per wavefront instead of one workgroup each.  Windows with more partials or a longer run of
// empty windows are left to long_final_block_kernel (lane_finishes is the split).
constexpr int kLaneParts = 8, kLaneGap = 8;
__device__ __forceinline__ bool lane_finishes(const LongEntry &le, int64_t nparts) {
    return nparts <= kLaneParts && (int64_t)(le.next_wid - le.wid) - 1 <= kLaneGap;
}

__global__ __launch_bounds__(256) void long_final_kernel(const AggParams p, const int64_t n_long, const LongEntry *entries,
                                                         const int64_t *offsets, const Part *partials, int32_t *leftover,
                                                         unsigned long long *n_leftover) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = e < n_long;
    LongEntry le;
    le.wid = 0; le.r0 = 0; le.r1 = 0; le.next_wid = 1; le.incl_row = 0; le.dead = 1;
    int64_t w0 = 0, w1 = 0;
    if (have) { le = entries[e]; w0 = offsets[e]; w1 = offsets[e + 1]; }
    const int64_t my_gap = have ? (int64_t)(le.next_wid - le.wid) - 1 : 0;
    const bool simple = have && lane_finishes(le, w1 - w0);
    if (have && !simple) leftover[atomicAdd(n_leftover, 1ull)] = (int32_t)e;  // (rare) for long_final_block_kernel
    (void)lane;

    for (int slot = -1; slot < p.ncols; slot++) {
        const unsigned my_mask = p.pass_mask[slot + 1];
        if (my_mask == 0) continue;
        const bool need_vals = slot >= 0 && (p.pass_flags[slot + 1] & kPassNeedVals);

        if (simple) {
            Part acc;
            part_init(acc);
            if (need_vals)
                for (int64_t w = w0; w < w1; w++) part_merge(acc, partials[w * p.ncols + slot]);  // (at most kLaneParts, in order)
            emit_window(p, slot, le, acc);
            emit_empties(p, slot, le, my_gap, 1, 1);
        }
    }
}

// The windows the lane-per-window kernels leave (many chunk partials, or a long run of empty windows behind them): one workgroup
// per window merges its chunk partials in a fixed shape, writes the outputs, then the empty windows that follow it.
// leftover: indices into entries / off0 / off1 (nullptr: the lists are compact, entry i).  stream_chunks > 0: the entries come
// from stream_final_kernel - windows that ran past its look-ahead; their end is found here by bisection, their partials are
// [off0, 2 * (chunk of the end) + 1) of the chunk grid, and the empty windows are not theirs to write.
__global__ __launch_bounds__(256) void long_final_block_kernel(const AggParams p, const LongEntry *entries, const int64_t *off0,
                                                               const int64_t *off1, const Part *partials, const int32_t *leftover,
                                                               const unsigned long long *n_leftover, const int64_t stream_chunks, const int lite) {
    __shared__ Part red[4];
    __shared__ LongEntry s_le;
    __shared__ int64_t s_w1;
    const int64_t nleft = (int64_t)*n_leftover;  // (usually 0: the launch then costs a few microseconds)
    for (int64_t i = blockIdx.x; i < nleft; i += gridDim.x) {
        const int64_t e = leftover ? leftover[i] : i;
        const int lane = threadIdx.x;
        if (lane == 0) {
            LongEntry le = entries[e];
            int64_t w1 = off1[e];
            if (stream_chunks > 0) {
                le.r1 = window_end_row(p, le.wid, le.r0 + 1);
                entry_close(p, le);
                le.next_wid = le.wid + 1;
                w1 = 2 * (le.r1 / kStreamRows) + 1;
                if (w1 > 2 * stream_chunks) w1 = 2 * stream_chunks;
            }
            s_le = le;
            s_w1 = w1;
        }
        __syncthreads();
        const LongEntry le = s_le;
        const int64_t w0 = off0[e], w1 = s_w1;
        const int64_t gap = (int64_t)(le.next_wid - le.wid) - 1;
        for (int slot = -1; slot < p.ncols; slot++) {
            if (p.pass_mask[slot + 1] == 0) continue;
            Part acc;
            part_init(acc);
            if (slot >= 0 && (p.pass_flags[slot + 1] & kPassNeedVals)) {
                if (w1 - w0 == 1) {  // the common medium-sized window: nothing to merge
                    if (lane == 0) acc = part_at(partials, w0 * p.ncols + slot, lite);
                } else {
                    for (int64_t w = w0 + lane; w < w1; w += 256) part_merge(acc, part_at(partials, w * p.ncols + slot, lite));
                    block_reduce(acc, red, lane);
                    __syncthreads();
                }
            }
            if (lane == 0) emit_window(p, slot, le, acc);
            emit_empties(p, slot, le, gap, 1 + lane, 256);
        }
        __syncthreads();  // (red[], s_le are reused by the next entry)
    }
}

// ---------------------------------------------------------------- the streaming form (every window of the call, one read of the rows)
//
// When the windows of a call average a hundred rows or more, the lane-per-window tile kernels would queue most of them for the
// pipeline above, which then reads the rows a second time.  Instead: ONE pass over the rows on a fixed grid of kStreamRows-row
// chunks, one wavefront per chunk, nothing through LDS: lane l holds rows 128 j + 2 l, + 1 of the chunk for trip j (16-byte loads).
// Every chunk cuts its rows into its windows' segments and reduces each segment into an order-free partial:
//   - a window that starts and ends inside the chunk: its partial goes to wparts[window], its rows to recs[window];
//   - the rows of a window that began in an earlier chunk go to the chunk's HEAD partial, the rows of one that runs on into the
//     next chunk to its TAIL partial: parts[2 g] / parts[2 g + 1] (identity partials when unused), so that the partials of a window
//     that spans chunks g .. k are the consecutive range [2 g + 1, 2 k + 1); recs[window] names the chunk it starts in.
// Two kernels do that.  long_short_kernel takes the chunks with at most one boundary behind their first row - all of them once
// the windows are longer than a chunk - with uniform (scalar) work for everything a boundary decides and one plain reduction per
// segment and field; it flags the others for long_stream_kernel, which handles any number of boundaries per chunk (a segmented
// DPP scan for the {sum, count} sets, a reduction per segment for the others) and is the only one launched when the windows are
// shorter than a chunk on average.  stream_final_kernel then finishes the windows, one LANE per window of the call: empty ones
// (no record), the ones inside a chunk, and the ones over a few chunks (it walks the heads that follow); windows over many
// chunks go to long_final_block_kernel.  Same order-free reducers and the same tolerance as above; every row is read once.
// 1e8 rows, 1000-row windows (bracket of all kernels of the call, dense / 30 % nulls): Mean 0.28 / 0.37 ms, First + Last 0.30 /
// 0.39, Sum + Mean + Min + Max 0.32 / 0.46, WeightedAverageStep 0.35 / 0.55 ms - against 0.44 - 0.98 ms for the bisection form.
struct ChunkMeta {
    int32_t head_rows;    // leading rows that belong to the window of the row before the chunk (the whole chunk: it runs through)
};
struct WinRec {
    int64_t r0;           // first row of the window; < 0: the window has no rows
    int32_t rows;         // its rows when it lies inside one chunk; -1: it runs on into the chunks after `chunk`
    int32_t chunk;
};

constexpr int kStreamScan = 64;   // chunks a lane of stream_final_kernel looks ahead for the end of its window

__device__ __forceinline__ uint32_t magic_div32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}

typedef unsigned long long u64x2s_t __attribute__((ext_vector_type(2)));
// rows i0, i0 + 1 of the chunk (i0 even) as one 16-byte streaming load when the column allows it
__device__ __forceinline__ void stream_load_pair(const uint64_t *base, int i0, int rows, bool vec, uint64_t &x, uint64_t &y) {
    if (vec && i0 + 1 < rows) {
        const u64x2s_t q = __builtin_nontemporal_load(reinterpret_cast<const u64x2s_t *>(base + i0));
        x = q.x; y = q.y;
    } else {
        x = i0 < rows ? __builtin_nontemporal_load(base + i0) : 0;
        y = i0 + 1 < rows ? __builtin_nontemporal_load(base + i0 + 1) : 0;
    }
}

// ---- one DPP move of a 32-bit / 64-bit lane value; lanes without a source receive 0
template <int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, kCtrl, kRowMask, 0xf, kBound);
}
template <int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), kCtrl, kRowMask, 0xf, kBound);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), kCtrl, kRowMask, 0xf, kBound);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
// a DPP move that leaves a lane without a source (or of a masked row) its own value: the identity of min / max
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_keep_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, kCtrl, kRowMask, 0xf, false);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_keep_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), kCtrl, kRowMask, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), kCtrl, kRowMask, 0xf, false);
    return __hiloint2double(hi, lo);
}
// reductions over the wavefront (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31): the total arrives in lane 63
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0x111, 0xf, true>(v); v += dpp_f64<0x112, 0xf, true>(v); v += dpp_f64<0x114, 0xf, true>(v);
    v += dpp_f64<0x118, 0xf, true>(v); v += dpp_f64<0x142, 0xa, false>(v); v += dpp_f64<0x143, 0xc, false>(v);
    return v;
}
// (fmin / fmax: a NaN operand - a lane without a value - yields the other one)
__device__ __forceinline__ double wave_min_f64(double v) {
    v = fmin(v, dpp_keep_f64<0x111, 0xf>(v)); v = fmin(v, dpp_keep_f64<0x112, 0xf>(v)); v = fmin(v, dpp_keep_f64<0x114, 0xf>(v));
    v = fmin(v, dpp_keep_f64<0x118, 0xf>(v)); v = fmin(v, dpp_keep_f64<0x142, 0xa>(v)); v = fmin(v, dpp_keep_f64<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_keep_f64<0x111, 0xf>(v)); v = fmax(v, dpp_keep_f64<0x112, 0xf>(v)); v = fmax(v, dpp_keep_f64<0x114, 0xf>(v));
    v = fmax(v, dpp_keep_f64<0x118, 0xf>(v)); v = fmax(v, dpp_keep_f64<0x142, 0xa>(v)); v = fmax(v, dpp_keep_f64<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    v = mn(v, dpp_keep_u32<0x111, 0xf>(v)); v = mn(v, dpp_keep_u32<0x112, 0xf>(v)); v = mn(v, dpp_keep_u32<0x114, 0xf>(v));
    v = mn(v, dpp_keep_u32<0x118, 0xf>(v)); v = mn(v, dpp_keep_u32<0x142, 0xa>(v)); v = mn(v, dpp_keep_u32<0x143, 0xc>(v));
    return v;
}

// The order-free partial of a run of rows as the segmented scan carries it: row indices are 1 + the row's index in the chunk
// (0: none), so that the zero a DPP move hands to a lane without a source is the identity.
template <int kNeed>
struct SegVal {
    double sum;
    uint32_t cnt;
    uint32_t first, last;       // kNeed != 0
    double vmin, vmax;          // kNeed & 1
    uint32_t imin, imax;
    double trap, step;          // kNeed & 4
};
template <int kNeed>
__device__ __forceinline__ SegVal<kNeed> seg_identity() {
    SegVal<kNeed> v;
    v.sum = 0.0; v.cnt = 0; v.first = 0; v.last = 0; v.vmin = 0.0; v.vmax = 0.0; v.imin = 0; v.imax = 0; v.trap = 0.0; v.step = 0.0;
    return v;
}
// a (earlier rows) followed by b (later rows)
template <int kNeed>
__device__ __forceinline__ SegVal<kNeed> seg_join(const SegVal<kNeed> &a, const SegVal<kNeed> &b) {
    SegVal<kNeed> r = b;
    r.sum = a.sum + b.sum;
    r.cnt = a.cnt + b.cnt;
    if (kNeed != 0) { r.first = a.first ? a.first : b.first; r.last = b.last ? b.last : a.last; }
    if (kNeed & 1) {   // ties keep the smaller row index: a's
        if (a.imin && (!b.imin || a.vmin <= b.vmin)) { r.vmin = a.vmin; r.imin = a.imin; }
        if (a.imax && (!b.imax || a.vmax >= b.vmax)) { r.vmax = a.vmax; r.imax = a.imax; }
    }
    if (kNeed & 4) { r.trap = a.trap + b.trap; r.step = a.step + b.step; }
    return r;
}
// the same for two partials in ANY order (rows identified by their indices): what the per-segment reduction below uses, whose
// lanes hold rows of several 128-row trips
template <int kNeed>
__device__ __forceinline__ SegVal<kNeed> seg_join_any(const SegVal<kNeed> &a, const SegVal<kNeed> &b) {
    SegVal<kNeed> r = b;
    r.sum = a.sum + b.sum;
    r.cnt = a.cnt + b.cnt;
    if (kNeed != 0) {
        r.first = (a.first && (!b.first || a.first < b.first)) ? a.first : b.first;
        r.last = a.last > b.last ? a.last : b.last;
    }
    if (kNeed & 1) {   // ties keep the smaller row index
        if (a.imin && (!b.imin || a.vmin < b.vmin || (a.vmin == b.vmin && a.imin < b.imin))) { r.vmin = a.vmin; r.imin = a.imin; }
        if (a.imax && (!b.imax || a.vmax > b.vmax || (a.vmax == b.vmax && a.imax < b.imax))) { r.vmax = a.vmax; r.imax = a.imax; }
    }
    if (kNeed & 4) { r.trap = a.trap + b.trap; r.step = a.step + b.step; }
    return r;
}
template <int kNeed, int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ SegVal<kNeed> seg_dpp(const SegVal<kNeed> &v) {
    SegVal<kNeed> r = seg_identity<kNeed>();
    r.sum = dpp_f64<kCtrl, kRowMask, kBound>(v.sum);
    r.cnt = dpp_u32<kCtrl, kRowMask, kBound>(v.cnt);
    if (kNeed != 0) { r.first = dpp_u32<kCtrl, kRowMask, kBound>(v.first); r.last = dpp_u32<kCtrl, kRowMask, kBound>(v.last); }
    if (kNeed & 1) {
        r.vmin = dpp_f64<kCtrl, kRowMask, kBound>(v.vmin); r.vmax = dpp_f64<kCtrl, kRowMask, kBound>(v.vmax);
        r.imin = dpp_u32<kCtrl, kRowMask, kBound>(v.imin); r.imax = dpp_u32<kCtrl, kRowMask, kBound>(v.imax);
    }
    if (kNeed & 4) { r.trap = dpp_f64<kCtrl, kRowMask, kBound>(v.trap); r.step = dpp_f64<kCtrl, kRowMask, kBound>(v.step); }
    return r;
}
template <int kNeed>
__device__ __forceinline__ SegVal<kNeed> seg_readlane(const SegVal<kNeed> &v, int l) {
    SegVal<kNeed> r = seg_identity<kNeed>();
    r.sum = readlane_f64(v.sum, l);
    r.cnt = (uint32_t)__builtin_amdgcn_readlane((int)v.cnt, l);
    if (kNeed != 0) { r.first = (uint32_t)__builtin_amdgcn_readlane((int)v.first, l); r.last = (uint32_t)__builtin_amdgcn_readlane((int)v.last, l); }
    if (kNeed & 1) {
        r.vmin = readlane_f64(v.vmin, l); r.vmax = readlane_f64(v.vmax, l);
        r.imin = (uint32_t)__builtin_amdgcn_readlane((int)v.imin, l); r.imax = (uint32_t)__builtin_amdgcn_readlane((int)v.imax, l);
    }
    if (kNeed & 4) { r.trap = readlane_f64(v.trap, l); r.step = readlane_f64(v.step, l); }
    return r;
}
// one step of the segmented inclusive scan: lanes whose run has not met a boundary yet take the incoming prefix
template <int kNeed, int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ void seg_step(SegVal<kNeed> &v, uint32_t &f) {
    const SegVal<kNeed> in = seg_dpp<kNeed, kCtrl, kRowMask, kBound>(v);
    const uint32_t fin = dpp_u32<kCtrl, kRowMask, kBound>(f);
    if (!f) v = seg_join<kNeed>(in, v);
    f |= fin;
}
// partial -> storage slot i: a Part, or (kNeed == 0) a PartLite
template <int kNeed>
__device__ __forceinline__ void seg_store(Part *base, int64_t i, const SegVal<kNeed> &v, int64_t c0) {
    if (kNeed == 0) {
        PartLite l;
        l.sum = v.sum; l.count = (int64_t)v.cnt;
        reinterpret_cast<PartLite *>(base)[i] = l;
    } else {
        Part q;
        part_init(q);
        q.sum = v.sum; q.count = (int64_t)v.cnt;
        q.first_idx = v.first ? c0 + v.first - 1 : -1; q.last_idx = v.last ? c0 + v.last - 1 : -1;
        if (kNeed & 1) {
            q.vmin = v.vmin; q.vmax = v.vmax;
            q.min_idx = v.imin ? c0 + v.imin - 1 : -1; q.max_idx = v.imax ? c0 + v.imax - 1 : -1;
        }
        if (kNeed & 4) { q.trap = v.trap; q.step = v.step; }
        base[i] = q;
    }
}

// ---- one lane's partial of a segment, for the chunks with at most one boundary (long_stream_kernel's short path)
__device__ __forceinline__ double asm_min_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double asm_max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int kNeed, bool kTrack>
struct LaneAcc {
    double s, tr, st, mn, mx;    // extrema: seeded with +inf / -inf and replaced on < / > only, so a NaN never enters and equal values
    uint32_t imn, imx;           // keep the earliest row; nothing below +inf found: index 0, emit_window falls back to the window's first value
    uint32_t vc, vf, vl;         // kTrack (nullable columns): the lane's valid rows of the segment - how many, the first and the last (1 + row)
    __device__ __forceinline__ void init() {
        s = 0.0; tr = 0.0; st = 0.0; imn = 0; imx = 0;
        mn = __longlong_as_double(0x7ff0000000000000ll); mx = __longlong_as_double((long long)0xfff0000000000000ull);
        vc = 0; vf = 0xFFFFFFFFu; vl = 0;
    }
    __device__ __forceinline__ void track(bool c, uint32_t idx1) {
        if (kTrack) {
            vc += c ? 1u : 0u;
            const uint32_t lo = c ? idx1 : 0xFFFFFFFFu, hi = c ? idx1 : 0u;
            vf = lo < vf ? lo : vf; vl = hi > vl ? hi : vl;
        }
    }
    __device__ __forceinline__ void add(double x, double trv, double stv, uint32_t idx1) {
        s += x;
        if (kNeed & 1) {
            if (x < mn) { mn = x; imn = idx1; }
            if (x > mx) { mx = x; imx = idx1; }
        }
        if (kNeed & 4) { tr += trv; st += stv; }
    }
    // (a row of a nullable column: xs = its value or 0, its terms are zero when it has none)
    __device__ __forceinline__ void add_v(bool c, double xs, double x, double trv, double stv, uint32_t idx1) {
        s += xs;
        if (kNeed & 1) {
            if (c && x < mn) { mn = x; imn = idx1; }
            if (c && x > mx) { mx = x; imx = idx1; }
        }
        if (kNeed & 4) { tr += trv; st += stv; }
        track(c, idx1);
    }
    __device__ __forceinline__ void add_if(bool c, double x, double trv, double stv, uint32_t idx1) {
        s += c ? x : 0.0;
        if (kNeed & 1) {
            if (c && x < mn) { mn = x; imn = idx1; }
            if (c && x > mx) { mx = x; imx = idx1; }
        }
        if (kNeed & 4) { tr += c ? trv : 0.0; st += c ? stv : 0.0; }
        track(c, idx1);
    }
};
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_u32<0x111, 0xf, true>(v); v += dpp_u32<0x112, 0xf, true>(v); v += dpp_u32<0x114, 0xf, true>(v);
    v += dpp_u32<0x118, 0xf, true>(v); v += dpp_u32<0x142, 0xa, false>(v); v += dpp_u32<0x143, 0xc, false>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, dpp_keep_u32<0x111, 0xf>(v)); v = mx(v, dpp_keep_u32<0x112, 0xf>(v)); v = mx(v, dpp_keep_u32<0x114, 0xf>(v));
    v = mx(v, dpp_keep_u32<0x118, 0xf>(v)); v = mx(v, dpp_keep_u32<0x142, 0xa>(v)); v = mx(v, dpp_keep_u32<0x143, 0xc>(v));
    return v;
}
// the lanes' partials joined: the segment's partial, uniform.  kTrack: count / first / last valid row from the lanes' own (cnt / first /
// last are ignored); else they are the caller's closed forms (rows, 0-based).  Extrema: the extreme over the wavefront, then the
// earliest row among the lanes that hold it - a lane holds the earliest of its own rows, lanes of one 128-row trip are in row
// order, trips are tried in order.
template <int kNeed, bool kTrack>
__device__ __forceinline__ SegVal<kNeed> lane_acc_finish(const LaneAcc<kNeed, kTrack> &a, uint32_t cnt, int first, int last) {
    SegVal<kNeed> t = seg_identity<kNeed>();
    if (kTrack) cnt = readlane_u32(wave_sum_u32(a.vc), 63);
    if (cnt == 0) return t;
    t.cnt = cnt;
    if (kTrack) { t.first = readlane_u32(wave_min_u32(a.vf), 63); t.last = readlane_u32(wave_max_u32(a.vl), 63); }
    else { t.first = (uint32_t)(first + 1); t.last = (uint32_t)(last + 1); }
    t.sum = readlane_f64(wave_sum_f64(a.s), 63);
    if (kNeed & 4) { t.trap = readlane_f64(wave_sum_f64(a.tr), 63); t.step = readlane_f64(wave_sum_f64(a.st), 63); }
    if (kNeed & 1) {
        double v = a.mn;
        v = asm_min_f64(v, dpp_keep_f64<0x111, 0xf>(v)); v = asm_min_f64(v, dpp_keep_f64<0x112, 0xf>(v)); v = asm_min_f64(v, dpp_keep_f64<0x114, 0xf>(v));
        v = asm_min_f64(v, dpp_keep_f64<0x118, 0xf>(v)); v = asm_min_f64(v, dpp_keep_f64<0x142, 0xa>(v)); v = asm_min_f64(v, dpp_keep_f64<0x143, 0xc>(v));
        const double m = readlane_f64(v, 63);
        const bool cmin = a.imn != 0 && a.mn == m;
        v = a.mx;
        v = asm_max_f64(v, dpp_keep_f64<0x111, 0xf>(v)); v = asm_max_f64(v, dpp_keep_f64<0x112, 0xf>(v)); v = asm_max_f64(v, dpp_keep_f64<0x114, 0xf>(v));
        v = asm_max_f64(v, dpp_keep_f64<0x118, 0xf>(v)); v = asm_max_f64(v, dpp_keep_f64<0x142, 0xa>(v)); v = asm_max_f64(v, dpp_keep_f64<0x143, 0xc>(v));
        const double M = readlane_f64(v, 63);
        const bool cmax = a.imx != 0 && a.mx == M;
        const uint32_t tmn = (a.imn - 1u) >> 7, tmx = (a.imx - 1u) >> 7;
        bool dmin = false, dmax = false;
#pragma unroll
        for (int j = 0; j < kStreamRows / 128; j++) {
            const uint64_t b1 = dmin ? 0ull : __ballot(cmin && tmn == (uint32_t)j), b2 = dmax ? 0ull : __ballot(cmax && tmx == (uint32_t)j);
            if (b1) { const int l = __ffsll((long long)b1) - 1; t.imin = readlane_u32(a.imn, l); t.vmin = readlane_f64(a.mn, l); dmin = true; }
            if (b2) { const int l = __ffsll((long long)b2) - 1; t.imax = readlane_u32(a.imx, l); t.vmax = readlane_f64(a.mx, l); dmax = true; }
        }
    }
    return t;
}

// The general form.  kNeed: 1 extrema, 2 first / last, 4 time-weighted terms (any of them: the row indices are tracked); 0: {sum,
// count} storage.  Per trip: the rows' window ids (relative to the chunk's first window: 32-bit arithmetic when the chunk allows
// it), a boundary flag per row (its id differs from the row before).  kNeed == 0: a SEGMENTED inclusive scan of the rows'
// partials over the 64 lanes (DPP: row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) with the running partial carried from trip to trip
// in scalar registers; every boundary row closes the window of the row before it: the lane that holds the boundary writes that
// window's partial and record.  kNeed != 0: the chunk's boundaries are walked in row order, one reduction per segment.
// kFlagged: the chunks long_short_kernel left to this kernel (only[g] != 0) - a wavefront looks at 64 flags and takes the chunks they
// name one after the other (a launch of one wavefront per chunk that finds nothing to do costs 0.02 ms per 1e8 rows all the same).
template <int kNeed, bool kFlagged>
__global__ __launch_bounds__(256) void long_stream_kernel(const AggParams p, const int64_t nchunks, ChunkMeta *meta, Part *parts, WinRec *recs,
                                                          Part *wparts, const uint8_t *only) {
    typedef SegVal<kNeed> SV;
    constexpr bool kMinMax = (kNeed & 1) != 0, kTw = (kNeed & 4) != 0, kIdx = kNeed != 0;
    constexpr int kTrips = kStreamRows / 128;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_outer = lane;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t g_base = kFlagged ? wave * 64 : wave;
    uint64_t todo = kFlagged ? __ballot(g_base + lane < nchunks && only[g_base + lane] != 0) : (wave < nchunks ? 1ull : 0ull);
    while (todo) {
    const int64_t g = g_base + __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int64_t c0 = g * kStreamRows;
    const int rows = (int)(p.n - c0 < kStreamRows ? p.n - c0 : kStreamRows);
    const bool has_prev = c0 > 0, has_next = c0 + rows < p.n;
    const uint64_t *tsp = reinterpret_cast<const uint64_t *>(p.ts);
    const bool tvec = (reinterpret_cast<uintptr_t>(p.ts) & 15) == 0;
    int slot0 = -1;   // the first column pass that reads values: its loads travel with the timestamps'
    for (int sl = p.ncols - 1; sl >= 0; sl--)
        if (p.pass_mask[sl + 1] && (p.pass_flags[sl + 1] & kPassNeedVals)) slot0 = sl;

    uint64_t tx[kTrips], ty[kTrips], vx[kTrips], vy[kTrips];
#pragma unroll
    for (int j = 0; j < kTrips; j++) stream_load_pair(tsp + c0, j * 128 + 2 * lane, rows, tvec, tx[j], ty[j]);
    if (slot0 >= 0) {
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(p.cols[slot0].values);
        const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
        for (int j = 0; j < kTrips; j++) stream_load_pair(vp + c0, j * 128 + 2 * lane, rows, vvec, vx[j], vy[j]);
    }
    // (uniform: scalar loads) the rows around the chunk and its first / last row
    const int64_t ts_prev = p.ts[has_prev ? c0 - 1 : c0], ts_next = p.ts[has_next ? c0 + rows : c0 + rows - 1];
    const int64_t ts_first = p.ts[c0], ts_last = p.ts[c0 + rows - 1];
    const uint64_t wid0 = row_wid(p, ts_first);
    const int64_t start0 = p.s0 + (int64_t)(wid0 * (uint64_t)p.interval);
    const bool fast32 = p.fits32 && ts_last >= ts_first && (uint64_t)ts_last - (uint64_t)start0 < 0xFFFFFFFFull;
    const uint32_t rel_last = (uint32_t)(row_wid(p, ts_last) - wid0);
    const bool prev_same = has_prev && (ts_prev >= start0 || wid0 == 0);      // (rows are ascending; rows below s0 ride in window 0)
    bool next_same = false;
    {
        const uint64_t wl = wid0 + rel_last;
        const int64_t ws = p.s0 + (int64_t)(wl * (uint64_t)p.interval), lim = ws + p.interval;
        next_same = has_next && (lim < ws || ts_next < lim);
    }
    auto rel_of = [&](int64_t t) -> uint32_t {
        if (fast32) return t < start0 ? 0u : magic_div32((uint32_t)((uint64_t)t - (uint64_t)start0), p.m32, p.sh1_32, p.sh2_32);
        return (uint32_t)(row_wid(p, t) - wid0);
    };

    // ---- window ids, boundary flags, order check, the start row of the run each row belongs to (1 + row; 0: before the chunk)
    uint32_t relx[kTrips], rely[kTrips], pst[kTrips], fmask = 0;
    int head_rows = 0;     // (one lane) the boundary that closes the window running in from the chunks before
    uint32_t carry_rel = 0, carry_st = 0;
    {
        bool bad = false;
        int64_t carry_ts = ts_prev;
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            const int i0 = j * 128 + 2 * lane;
            const bool inx = i0 < rows, iny = i0 + 1 < rows;
            const int64_t x = (int64_t)tx[j], y = (int64_t)ty[j];
            const int64_t py = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)carry_ts, (int)(uint32_t)(uint64_t)y, 0x138, 0xf, 0xf, false) |
                                         (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)carry_ts >> 32), (int)(uint32_t)((uint64_t)y >> 32), 0x138, 0xf, 0xf, false) << 32);
            bad |= (inx && py > x) || (iny && x > y);
            carry_ts = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)y, 63) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)y >> 32), 63) << 32);
            const uint32_t rx = inx ? rel_of(x) : rel_last, ry = iny ? rel_of(y) : rel_last;
            relx[j] = rx; rely[j] = ry;
            const uint32_t prel = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_rel, (int)ry, 0x138, 0xf, 0xf, false);
            const bool fx = inx && ((j == 0 && lane == 0) ? !prev_same : prel != rx);
            const bool fy = iny && ry != rx;
            fmask |= (fx ? 1u : 0u) << (2 * j) | (fy ? 2u : 0u) << (2 * j);
            carry_rel = (uint32_t)__builtin_amdgcn_readlane((int)ry, 63);
            // start of the run in front of row x: the latest boundary among the rows before it (plain max-scan + carry)
            uint32_t m = fy ? (uint32_t)(i0 + 2) : (fx ? (uint32_t)(i0 + 1) : 0u);
            auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
            m = mx(m, dpp_u32<0x111, 0xf, true>(m)); m = mx(m, dpp_u32<0x112, 0xf, true>(m));
            m = mx(m, dpp_u32<0x114, 0xf, true>(m)); m = mx(m, dpp_u32<0x118, 0xf, true>(m));
            m = mx(m, dpp_u32<0x142, 0xa, false>(m)); m = mx(m, dpp_u32<0x143, 0xc, false>(m));
            const uint32_t mprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x138, 0xf, 0xf, false);
            pst[j] = mx(carry_st, mprev);
            carry_st = mx(carry_st, (uint32_t)__builtin_amdgcn_readlane((int)m, 63));
            // the records of the windows these boundaries close
            const uint32_t row_x = (uint32_t)i0, row_y = (uint32_t)(i0 + 1);
            if (fx && row_x > 0) {
                if (pst[j] == 0) head_rows = (int)row_x;
                else {
                    const int64_t oslot = (int64_t)(wid0 + prel - (uint64_t)p.wid_base);
                    if ((uint64_t)oslot < (uint64_t)p.W) { WinRec r; r.r0 = c0 + pst[j] - 1; r.rows = (int32_t)(row_x - (pst[j] - 1)); r.chunk = (int32_t)g; recs[oslot] = r; }
                }
            }
            if (fy) {
                const uint32_t st = fx ? row_x + 1 : pst[j];
                if (st == 0) head_rows = (int)row_y;
                else {
                    const int64_t oslot = (int64_t)(wid0 + rx - (uint64_t)p.wid_base);
                    if ((uint64_t)oslot < (uint64_t)p.W) { WinRec r; r.r0 = c0 + st - 1; r.rows = (int32_t)(row_y - (st - 1)); r.chunk = (int32_t)g; recs[oslot] = r; }
                }
            }
        }
        if (lane == 0) bad |= ts_last > ts_next;
        if (__ballot(bad) && lane == 0) atomicOr(&p.status[0], 1u);
    }
    // the run that reaches the end of the chunk: through (no boundary at all), a whole window, or one that runs on
    const bool through = carry_st == 0;
    const bool tail_open = !through && next_same;
    const int64_t tail_slot = (int64_t)(wid0 + rel_last - (uint64_t)p.wid_base);
    const unsigned long long hm = __ballot(head_rows != 0);
    const int hr = through ? rows : (hm ? __shfl(head_rows, __ffsll((long long)hm) - 1) : 0);
    if (lane == 0) {
        meta[g].head_rows = hr;
        if (!through && (uint64_t)tail_slot < (uint64_t)p.W) {
            WinRec r;
            r.r0 = c0 + carry_st - 1; r.rows = tail_open ? -1 : (int32_t)(rows - (int)(carry_st - 1)); r.chunk = (int32_t)g;
            recs[tail_slot] = r;
        }
    }

    // ---- per column pass, the reducer sets beyond {sum, count} (kNeed != 0).  Windows here average hundreds of rows or more, so a
    // 512-row chunk holds a handful of boundaries at most - half of the chunks none at 1000-row windows: instead of a SEGMENTED
    // SCAN of ten fields over the lanes for each of the four trips (24 DPP steps of ~40 instructions), ONE reduction per segment
    // of the chunk, field by field with what each field needs:
    //   count / first / last valid row   scalar: population count, lowest and highest bit of the rows' ballots
    //   sum, the time-weighted terms     eight local additions + six DPP steps each
    //   extrema                          local strict compare in row order, v_min_f64 / v_max_f64 over the wavefront, then the
    //                                    smallest row index among the lanes that hold the extreme (the reference keeps the first
    //                                    of equal values - minmax.go:16-28 replaces on < / > only - and +0 / -0 are equal values)
    // The time-weighted terms of a row need its NEXT both-valid point: fetched from the lane that holds it (ballot of the lanes
    // with a valid row, lowest set bit above the lane, five ds_bpermute) instead of a bitmap walk + two gathers per row.
    if (kNeed != 0) {
        uint64_t bx[kTrips], by[kTrips];   // the chunk's boundaries as lane masks per trip (x rows / y rows)
        int nb = 0;
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            bx[j] = __ballot((fmask >> (2 * j)) & 1u);
            by[j] = __ballot((fmask >> (2 * j)) & 2u);
            nb += __popcll(bx[j]) + __popcll(by[j]);
        }
        if (bx[0] & 1ull) nb--;            // (a boundary at the chunk's first row closes nothing inside the chunk)
        const double kNaN = __longlong_as_double(0x7ff8000000000000ll);
        for (int slot = 0; slot < p.ncols; slot++) {
            if (p.pass_mask[slot + 1] == 0 || !(p.pass_flags[slot + 1] & kPassNeedVals)) continue;
            int lane_v = lane_outer;   // (opaque per column pass: see long_short_kernel)
            asm volatile("" : "+v"(lane_v));
            const int lane = lane_v;
            const ColDesc &cd = p.cols[slot];
            const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd.values);
            const bool need_ts = kTw && cd.need_ts;
            if (slot != slot0) {
                const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
                for (int j = 0; j < kTrips; j++) stream_load_pair(vp + c0, j * 128 + 2 * lane, rows, vvec, vx[j], vy[j]);
            }
            // validity of the lane's rows; the values as float64 (window.go: every reducer reads them through GetFloat64)
            uint32_t vmask = 0;
            double xv[kTrips], yv[kTrips];
#pragma unroll
            for (int j = 0; j < kTrips; j++) {
                const int i0 = j * 128 + 2 * lane;
                if (i0 < rows && col_valid(cd, c0 + i0)) vmask |= 1u << (2 * j);
                if (i0 + 1 < rows && col_valid(cd, c0 + i0 + 1)) vmask |= 2u << (2 * j);
                xv[j] = bits_to_f64(vx[j], cd.type); yv[j] = bits_to_f64(vy[j], cd.type);
            }
            // time-weighted terms per row: (this point, the next both-valid point of the same window)
            double trx[kTrips], try_[kTrips], stx[kTrips], sty[kTrips];
#pragma unroll
            for (int j = 0; j < kTrips; j++) { trx[j] = 0.0; try_[j] = 0.0; stx[j] = 0.0; sty[j] = 0.0; }
            if (need_ts) {
                // the first valid point behind the chunk (uniform)
                double cn_t = 0.0, cn_v = 0.0;
                uint32_t cn_rel = 0;
                bool cn_has = false;
                if (has_next) {
                    const int64_t rn = col_next_valid(cd, c0 + rows, p.n);
                    if (rn >= 0) {
                        const int64_t tn = p.ts[rn];
                        // (its window relative to the chunk's first: only "the same as the row's or not" is read; far-away rows saturate)
                        const uint64_t wn = row_wid(p, tn) - wid0;
                        cn_rel = wn > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)wn;
                        cn_t = (double)tn; cn_v = bits_to_f64(vp[rn], cd.type); cn_has = true;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < kTrips; jj++) {
                    const int j = kTrips - 1 - jj;     // last trip first: each trip hands its first valid point to the one before
                    const bool okx = (vmask >> (2 * j)) & 1u, oky = (vmask >> (2 * j)) & 2u;
                    const double xt = (double)(int64_t)tx[j], yt = (double)(int64_t)ty[j];
                    // the lane's own first valid point, and the lanes that have one
                    const double o_t = okx ? xt : yt, o_v = okx ? xv[j] : yv[j];
                    const uint32_t o_rel = okx ? relx[j] : rely[j];
                    const uint64_t hasm = __ballot(okx || oky);
                    // what follows lane l's row y: the own point of the next lane above l that has one, else the carry
                    const uint64_t above = (hasm >> lane) >> 1;
                    const int src = (lane + 1 + (above ? __ffsll((long long)above) - 1 : 0)) << 2;
                    double a_t = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(o_t)), __builtin_amdgcn_ds_bpermute(src, __double2loint(o_t)));
                    double a_v = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(o_v)), __builtin_amdgcn_ds_bpermute(src, __double2loint(o_v)));
                    uint32_t a_rel = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)o_rel);
                    bool a_has = true;
                    if (!above) { a_t = cn_t; a_v = cn_v; a_rel = cn_rel; a_has = cn_has; }
                    if (oky && a_has && a_rel == rely[j]) { try_[j] = (yv[j] + a_v) / 2 * (a_t - yt); sty[j] = yv[j] * (a_t - yt); }   // integral.go:24 / :55
                    if (okx) {
                        const double b_t = oky ? yt : a_t, b_v = oky ? yv[j] : a_v;
                        const uint32_t b_rel = oky ? rely[j] : a_rel;
                        const bool b_has = oky || a_has;
                        if (b_has && b_rel == relx[j]) { trx[j] = (xv[j] + b_v) / 2 * (b_t - xt); stx[j] = xv[j] * (b_t - xt); }
                    }
                    // the carry for the trip before: the own point of the lowest lane that has one
                    if (hasm) {
                        const int l0 = __ffsll((long long)hasm) - 1;
                        cn_t = readlane_f64(o_t, l0); cn_v = readlane_f64(o_v, l0);
                        cn_rel = (uint32_t)__builtin_amdgcn_readlane((int)o_rel, l0); cn_has = true;
                    }
                }
            }
            // one segment = the rows [ra, rb) of the chunk (kWhole: all of them): the total, uniform over the wavefront
            auto segment = [&](auto whole_t, int ra, int rb) -> SV {
                constexpr bool kWhole = decltype(whole_t)::value;
                double s = 0.0, tr = 0.0, st = 0.0, mn = kNaN, mx = kNaN;
                uint32_t imn = 0, imx = 0, cnt = 0;
                int first = 0x7fffffff, last = -1;
#pragma unroll
                for (int j = 0; j < kTrips; j++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int i = j * 128 + 2 * lane + h;
                        bool in = (vmask >> (2 * j + h)) & 1u;
                        if (!kWhole) in = in && i >= ra && i < rb;
                        const uint64_t bm = __ballot(in);
                        if (bm) {
                            cnt += (uint32_t)__popcll(bm);
                            const int f = j * 128 + 2 * (__ffsll((long long)bm) - 1) + h, l = j * 128 + 2 * (63 - __clzll((long long)bm)) + h;
                            first = f < first ? f : first; last = l > last ? l : last;
                        }
                        const double x = h ? yv[j] : xv[j];
                        s += in ? x : 0.0;
                        if (kMinMax) {
                            if (in && !(x >= mn) && x == x) { mn = x; imn = (uint32_t)(i + 1); }
                            if (in && !(x <= mx) && x == x) { mx = x; imx = (uint32_t)(i + 1); }
                        }
                        if (kTw) { tr += in ? (h ? try_[j] : trx[j]) : 0.0; st += in ? (h ? sty[j] : stx[j]) : 0.0; }
                    }
                }
                SV tot = seg_identity<kNeed>();
                tot.cnt = cnt;
                tot.first = cnt ? (uint32_t)(first + 1) : 0u; tot.last = cnt ? (uint32_t)(last + 1) : 0u;
                tot.sum = readlane_f64(wave_sum_f64(s), 63);
                if (kTw) { tot.trap = readlane_f64(wave_sum_f64(tr), 63); tot.step = readlane_f64(wave_sum_f64(st), 63); }
                if (kMinMax) {
                    const double m = readlane_f64(wave_min_f64(mn), 63), M = readlane_f64(wave_max_f64(mx), 63);
                    // (no lane has a value that is not NaN: m is NaN and equals nothing)
                    const uint32_t cn = readlane_u32(wave_min_u32((imn && mn == m) ? imn : 0xFFFFFFFFu), 63);
                    const uint32_t cx = readlane_u32(wave_min_u32((imx && mx == M) ? imx : 0xFFFFFFFFu), 63);
                    if (cn != 0xFFFFFFFFu) { tot.imin = cn; tot.vmin = readlane_f64(mn, __ffsll((long long)__ballot(imn == cn)) - 1); }
                    if (cx != 0xFFFFFFFFu) { tot.imax = cx; tot.vmax = readlane_f64(mx, __ffsll((long long)__ballot(imx == cx)) - 1); }
                }
                return tot;
            };
            SV tail;
            if (nb == 0) tail = segment(std::true_type(), 0, rows);
            else {
                // the boundaries in row order: each closes the segment in front of it
                int seg_start = 0;        // first row of the open segment
                bool first_seg = true;    // ... which is the one that runs in from the chunks before (when prev_same)
#pragma unroll 1
                for (int j = 0; j < kTrips; j++) {
                    uint64_t mx = bx[j], my = by[j];
                    while (mx | my) {
                        const int lx = mx ? __ffsll((long long)mx) - 1 : 64, ly = my ? __ffsll((long long)my) - 1 : 64;
                        const bool is_x = lx <= ly;      // (row x of a lane comes before its row y)
                        const int l = is_x ? lx : ly;
                        if (is_x) mx &= mx - 1; else my &= my - 1;
                        const int rb = j * 128 + 2 * l + (is_x ? 0 : 1);
                        if (rb == 0) { first_seg = false; continue; }
                        const SV tot = segment(std::false_type(), seg_start, rb);
                        // the window this segment belongs to: the one of row rb - 1
                        const int rp = rb - 1, jp = rp >> 7, lp = (rp & 127) >> 1;
                        uint32_t prel = 0;
#pragma unroll
                        for (int q = 0; q < kTrips; q++)
                            if (q == jp) prel = (uint32_t)__builtin_amdgcn_readlane((int)((rp & 1) ? rely[q] : relx[q]), lp);
                        if (lane == 63) {
                            if (first_seg && prev_same) seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, tot, c0);     // the HEAD partial
                            else {
                                const int64_t oslot = (int64_t)(wid0 + prel - (uint64_t)p.wid_base);
                                if ((uint64_t)oslot < (uint64_t)p.W) seg_store<kNeed>(wparts, oslot * p.ncols + slot, tot, c0);
                            }
                        }
                        first_seg = false;
                        seg_start = rb;
                    }
                }
                tail = segment(std::false_type(), seg_start, rows);   // the segment that reaches the end of the chunk
            }
            if (lane == 63) {
                const SV id = seg_identity<kNeed>();
                if (through) {
                    seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, tail, c0);
                    seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
                } else {
                    if (hr == 0) seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, id, c0);
                    if (tail_open) seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, tail, c0);
                    else {
                        seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
                        if ((uint64_t)tail_slot < (uint64_t)p.W) seg_store<kNeed>(wparts, tail_slot * p.ncols + slot, tail, c0);
                    }
                }
            }
        }
        continue;
    }

    // ---- per column pass: the rows' partials, the segmented scan, the partials of the closed windows
    for (int slot = 0; slot < p.ncols; slot++) {
        if (p.pass_mask[slot + 1] == 0 || !(p.pass_flags[slot + 1] & kPassNeedVals)) continue;
        int lane_v = lane_outer;   // (opaque per column pass: see long_short_kernel)
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        const ColDesc &cd = p.cols[slot];
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd.values);
        const bool need_ts = kTw && cd.need_ts;
        if (slot != slot0) {
            const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
            for (int j = 0; j < kTrips; j++) stream_load_pair(vp + c0, j * 128 + 2 * lane, rows, vvec, vx[j], vy[j]);
        }
        SV carry = seg_identity<kNeed>();   // (uniform) the run that reaches into the current trip
        bool head_done = false;
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            const int i0 = j * 128 + 2 * lane;
            const bool fx = (fmask >> (2 * j)) & 1u, fy = (fmask >> (2 * j)) & 2u;
            auto elem = [&](int i, uint64_t raw, int64_t t, uint32_t rel) -> SV {
                SV e = seg_identity<kNeed>();
                if (i >= rows || !col_valid(cd, c0 + i)) return e;
                const double x = bits_to_f64(raw, cd.type);
                e.sum = x; e.cnt = 1;
                if (kIdx) { e.first = (uint32_t)(i + 1); e.last = (uint32_t)(i + 1); }
                if (kMinMax && x == x) { e.vmin = x; e.vmax = x; e.imin = (uint32_t)(i + 1); e.imax = (uint32_t)(i + 1); }
                if (need_ts) {   // the term of the pair (this point, next both-valid point inside the window)
                    const int64_t rn = col_next_valid(cd, c0 + i + 1, p.n);
                    if (rn >= 0) {
                        const uint64_t wid = wid0 + rel;
                        const int64_t ws = p.s0 + (int64_t)(wid * (uint64_t)p.interval), lim = ws + p.interval;
                        const int64_t tn = p.ts[rn];
                        if (lim < ws || tn < lim) {
                            const double t0 = (double)t, t1 = (double)tn;
                            const double x1 = bits_to_f64(vp[rn], cd.type);
                            e.trap = (x + x1) / 2 * (t1 - t0);   // integral.go:24
                            e.step = x * (t1 - t0);              // integral.go:55
                        }
                    }
                }
                return e;
            };
            const SV ex = elem(i0, vx[j], (int64_t)tx[j], relx[j]), ey = elem(i0 + 1, vy[j], (int64_t)ty[j], rely[j]);
            // the lane's run since its last boundary, scanned over the lanes
            SV v = fy ? ey : seg_join<kNeed>(ex, ey);
            uint32_t f = (fx || fy) ? 1u : 0u;
            seg_step<kNeed, 0x111, 0xf, true>(v, f);
            seg_step<kNeed, 0x112, 0xf, true>(v, f);
            seg_step<kNeed, 0x114, 0xf, true>(v, f);
            seg_step<kNeed, 0x118, 0xf, true>(v, f);
            seg_step<kNeed, 0x142, 0xa, false>(v, f);
            seg_step<kNeed, 0x143, 0xc, false>(v, f);
            // exclusive: what lies in front of row x in its run (lane 0: the carry)
            SV before = seg_dpp<kNeed, 0x138, 0xf, false>(v);
            const uint32_t fbefore = dpp_u32<0x138, 0xf, false>(f);
            if (!fbefore) before = seg_join<kNeed>(carry, before);
            const SV upto_x = fx ? ex : seg_join<kNeed>(before, ex);
            // closed windows
            // (the id of the row before row x: a DPP move, so outside the divergent code below)
            const uint32_t prel0 = j > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)rely[j > 0 ? j - 1 : 0], 63) : 0u;
            const uint32_t prel = (uint32_t)__builtin_amdgcn_update_dpp((int)prel0, (int)rely[j], 0x138, 0xf, 0xf, false);
            if (fx && i0 > 0) {
                if (pst[j] == 0) { seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, before, c0); }
                else {
                    const int64_t oslot = (int64_t)(wid0 + prel - (uint64_t)p.wid_base);
                    if ((uint64_t)oslot < (uint64_t)p.W) seg_store<kNeed>(wparts, oslot * p.ncols + slot, before, c0);
                }
            }
            if (fy) {
                const uint32_t st = fx ? (uint32_t)(i0 + 1) : pst[j];
                if (st == 0) { seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, upto_x, c0); }
                else {
                    const int64_t oslot = (int64_t)(wid0 + relx[j] - (uint64_t)p.wid_base);
                    if ((uint64_t)oslot < (uint64_t)p.W) seg_store<kNeed>(wparts, oslot * p.ncols + slot, upto_x, c0);
                }
            }
            // the run that leaves the trip
            const SV last = seg_readlane<kNeed>(v, 63);
            const uint32_t flast = (uint32_t)__builtin_amdgcn_readlane((int)f, 63);
            carry = flast ? last : seg_join<kNeed>(carry, last);
        }
        (void)head_done;
        if (lane == 0) {
            const SV id = seg_identity<kNeed>();
            if (through) {
                seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, carry, c0);
                seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
            } else {
                if (hr == 0) seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, id, c0);
                if (tail_open) seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, carry, c0);
                else {
                    seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
                    if ((uint64_t)tail_slot < (uint64_t)p.W) seg_store<kNeed>(wparts, tail_slot * p.ncols + slot, carry, c0);
                }
            }
        }
    }
    }
}

// ---- Chunks with AT MOST ONE window boundary PER 128-ROW TRIP (and at most four in all) - every chunk once windows are longer
// than the chunk, nearly every chunk from 256-row windows on; the others are flagged (todo[g] = 1) for long_stream_kernel.  (A
// list with one atomic counter instead of the flags: 2.8 ms per 1e8 rows when every chunk is appended - 2e5 atomics on one
// address.)  Same outputs as long_stream_kernel (meta, recs, parts, wparts), but nothing is computed per row that a uniform value
// can say:
//   - no window ids: boundary k is the first row at or above the k-th window start behind the chunk's first window (one 64-bit
//     compare per row and boundary, none in a chunk without boundaries); no segmented scan: ONE running partial, closed by a plain
//     reduction at every boundary and at the chunk's end, range tests only in the trips that hold a boundary;
//   - count / first / last row of a segment: closed forms for a dense column; for a nullable one the lanes count their valid rows
//     and keep their first / last, three 32-bit reductions at the segment's end (the validity WORDS come by scalar loads; doing
//     the counts on them in scalar code cost more scalar instructions than the scalar unit had left);
//   - sum and the time-weighted terms: local additions + six DPP steps per field; extrema: see lane_acc_finish;
//   - the next point of a row (time-weighted terms): the same lane / the next lane (one DPP move) for a dense column, the lane
//     that holds it (ballot, lowest set bit above the lane, ds_bpermute) for a nullable one - no bitmap walk, no gathers.
// 1e8 rows, 1000-row windows, dense: ~440 vector instructions per 512-row chunk against 930 / 1370 (extrema / time-weighted) for the
// segment loop of long_stream_kernel and ~2000 for a segmented scan of all ten fields; a nullable column: 700 - 800 (and as many
// scalar ones - the bitmap words, the counts).  kDense: no column of the pass has a validity bitmap (the launcher's choice; one
// kernel with both paths needs 190 - 240 registers, the two apart 97 - 128).
template <int kNeed, bool kDense>
__global__ __launch_bounds__(256, 4) void long_short_kernel(const AggParams p, const int64_t nchunks, ChunkMeta *meta, Part *parts, WinRec *recs,
                                                            Part *wparts, uint8_t *todo) {
    typedef SegVal<kNeed> SV;
    constexpr bool kTw = (kNeed & 4) != 0;
    constexpr int kTrips = kStreamRows / 128;
    constexpr int kMaxB = 4;                 // boundaries per chunk (one per trip)
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (g >= nchunks) return;
    const int64_t c0 = g * kStreamRows;
    const int rows = (int)(p.n - c0 < kStreamRows ? p.n - c0 : kStreamRows);
    const bool has_prev = c0 > 0, has_next = c0 + rows < p.n;
    const uint64_t *tsp = reinterpret_cast<const uint64_t *>(p.ts);
    const bool tvec = (reinterpret_cast<uintptr_t>(p.ts) & 15) == 0;
    int slot0 = -1;   // the first column pass that reads values: its loads travel with the timestamps'
    for (int sl = p.ncols - 1; sl >= 0; sl--)
        if (p.pass_mask[sl + 1] && (p.pass_flags[sl + 1] & kPassNeedVals)) slot0 = sl;
    uint64_t tx[kTrips], ty[kTrips], vx[kTrips], vy[kTrips];
#pragma unroll
    for (int j = 0; j < kTrips; j++) stream_load_pair(tsp + c0, j * 128 + 2 * lane, rows, tvec, tx[j], ty[j]);
    if (slot0 >= 0) {
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(p.cols[slot0].values);
        const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
        for (int j = 0; j < kTrips; j++) stream_load_pair(vp + c0, j * 128 + 2 * lane, rows, vvec, vx[j], vy[j]);
    }
    // the validity words of a column's rows, in row order (bit k of W[q]: row 64 q + k of the chunk), and of the 128 rows behind the
    // chunk - ten scalar loads through the constant address space, issued with the column's value loads.  Only whole 64-bit words
    // inside the column: a bitmap at an odd bit offset and the column's last chunks go the general way.
    uint64_t W[2 * kTrips], Wn[2];
    auto load_validity = [&](const ColDesc &cd) -> bool {
        if (!cd.vbits) {
#pragma unroll
            for (int i = 0; i < 2 * kTrips; i++) W[i] = ~0ull;
            Wn[0] = ~0ull; Wn[1] = ~0ull;
            return true;
        }
        const int64_t bit = cd.vbit0 + c0;
        if (bit & 63) return false;
        typedef const uint64_t __attribute__((address_space(4))) *const_words64;
        const_words64 q = (const_words64)(uintptr_t)cd.vbits + (bit >> 6);
#pragma unroll
        for (int i = 0; i < 2 * kTrips; i++) W[i] = q[i];
        Wn[0] = q[2 * kTrips]; Wn[1] = q[2 * kTrips + 1];
        return true;
    };
    bool words_ok = true;
    if (!kDense && slot0 >= 0) words_ok = load_validity(p.cols[slot0]);
    // (uniform: scalar loads) the rows around the chunk and its first / last row
    const int64_t ts_prev = p.ts[has_prev ? c0 - 1 : c0], ts_next = p.ts[has_next ? c0 + rows : c0 + rows - 1];
    const int64_t ts_first = p.ts[c0], ts_last = p.ts[c0 + rows - 1];
    const uint64_t wid0 = row_wid(p, ts_first);
    const int64_t start0 = p.s0 + (int64_t)(wid0 * (uint64_t)p.interval);
    const uint32_t rel_last = (uint32_t)(row_wid(p, ts_last) - wid0);
    // float64(t1) - float64(t0) of two timestamps below 2^53 that are less than 2^32 apart is exactly their 32-bit difference: the
    // time-weighted terms are computed that way here, chunks outside these limits go the general way
    const bool small_t = ts_first > -(1ll << 53) && ts_next < (1ll << 53) && ts_next >= ts_first && (uint64_t)ts_next - (uint64_t)ts_first < 0xFFFFFFFFull;
    // (kDense: no column of the pass has a validity bitmap - the launcher's choice - and the one chunk that is not full goes the general way too)
    bool general = rel_last > (uint32_t)kMaxB || (kDense && rows != kStreamRows) || (kTw && !small_t) ||
                   (!kDense && (!words_ok || c0 + kStreamRows + 128 > p.n));   // (rel_last: also a chunk whose last timestamp lies below its first)
    const bool prev_same = has_prev && (ts_prev >= start0 || wid0 == 0);      // (rows are ascending; rows below s0 ride in window 0)
    bool next_same = false;
    {
        const uint64_t wl = wid0 + rel_last;
        const int64_t ws = p.s0 + (int64_t)(wl * (uint64_t)p.interval), lim = ws + p.interval;
        next_same = has_next && (lim < ws || ts_next < lim);
    }
    // ---- order check; boundary k = the first row at or above window start k behind the chunk's first window (k <= rel_last: such a row
    // exists, and start0 + k * interval <= ts_last cannot wrap); per trip: the boundary row it holds (-1: none) and the window,
    // relative to the chunk's first, that starts there (windows without rows in between: several k share a row, the last one counts)
    int tb_row[kTrips], tb_rel[kTrips];
#pragma unroll
    for (int j = 0; j < kTrips; j++) { tb_row[j] = -1; tb_rel[j] = 0; }
    if (!general) {
        bool bad = false;
        uint32_t kfound = 0;                 // boundaries 1 .. kfound are placed (rows ascend: boundary k + 1 never lies in front of boundary k)
        int64_t carry_ts = ts_prev;
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            const int i0 = j * 128 + 2 * lane;
            const bool inx = i0 < rows, iny = i0 + 1 < rows;
            const int64_t x = (int64_t)tx[j], y = (int64_t)ty[j];
            const int64_t py = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)carry_ts, (int)(uint32_t)(uint64_t)y, 0x138, 0xf, 0xf, false) |
                                         (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)((uint64_t)carry_ts >> 32), (int)(uint32_t)((uint64_t)y >> 32), 0x138, 0xf, 0xf, false) << 32);
            bad |= (inx && py > x) || (iny && x > y);
            carry_ts = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)y, 63) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)y >> 32), 63) << 32);
            // the boundaries that fall into this trip, in order (a trip with none - every trip of a chunk without boundaries - costs one
            // scalar compare)
#pragma unroll
            for (int it = 0; it < kMaxB; it++) {
                if (kfound >= rel_last) break;
                const int64_t lim = start0 + (int64_t)(kfound + 1) * p.interval;
                const uint64_t ax = __ballot(inx && x >= lim), ay = __ballot(iny && y >= lim);
                if (!(ax | ay)) break;
                const int lx = ax ? __ffsll((long long)ax) - 1 : 64, ly = ay ? __ffsll((long long)ay) - 1 : 64;
                const int r = j * 128 + (lx <= ly ? 2 * lx : 2 * ly + 1);
                if (tb_row[j] >= 0 && tb_row[j] != r) general = true;     // two boundaries in one trip
                tb_row[j] = r; tb_rel[j] = (int)(kfound + 1);
                kfound++;
            }
        }
        if (lane == 0) bad |= ts_last > ts_next;
        if (__ballot(bad) && lane == 0) atomicOr(&p.status[0], 1u);
        if (kfound != rel_last) general = true;     // (rows out of order: the general form raises the flag)
    }
    if (general) {
        if (lane == 0) todo[g] = 1;
        return;
    }
    if (lane == 0) todo[g] = 0;
    // (a chunk that stays here is full - kDense: rows != kStreamRows went the general way, else c0 + kStreamRows + 128 <= n: said to
    // the compiler, the row-bound predicates of everything below fold away)
    if (rows != kStreamRows) __builtin_unreachable();
    const bool b0 = !prev_same;      // a boundary at the chunk's first row: its first segment is a window of its own, not a head
    int nb = 0, first_rb = 0, last_rb = 0;
#pragma unroll
    for (int j = 0; j < kTrips; j++)
        if (tb_row[j] >= 0) { if (nb == 0) first_rb = tb_row[j]; last_rb = tb_row[j]; nb++; }
    const bool through = nb == 0 && !b0;
    const bool tail_open = !through && next_same;
    const int64_t tail_slot = (int64_t)(wid0 + rel_last - (uint64_t)p.wid_base);
    const int hr = through ? rows : ((nb > 0 && !b0) ? first_rb : 0);
    if (lane == 0) {
        meta[g].head_rows = hr;
        int seg_start = 0;
        uint32_t seg_rel = 0;
        bool first_seg = true;
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            if (tb_row[j] < 0) continue;
            if (!(first_seg && !b0)) {
                const int64_t sl = (int64_t)(wid0 + seg_rel - (uint64_t)p.wid_base);
                if ((uint64_t)sl < (uint64_t)p.W) { WinRec r; r.r0 = c0 + seg_start; r.rows = tb_row[j] - seg_start; r.chunk = (int32_t)g; recs[sl] = r; }
            }
            seg_start = tb_row[j]; seg_rel = (uint32_t)tb_rel[j]; first_seg = false;
        }
        if (!through && (uint64_t)tail_slot < (uint64_t)p.W) {
            WinRec r;
            r.r0 = c0 + last_rb; r.rows = tail_open ? -1 : (int32_t)(rows - last_rb); r.chunk = (int32_t)g;
            recs[tail_slot] = r;
        }
    }

    const int lane0 = lane;
    const int64_t g0 = g;
    for (int slot = 0; slot < p.ncols; slot++) {
        if (p.pass_mask[slot + 1] == 0 || !(p.pass_flags[slot + 1] & kPassNeedVals)) continue;
        // (the lane number, opaque per column pass: LLVM otherwise computes every lane predicate of the body - "row in front of this trip's
        // boundary", lane == 63, ... - in front of the loop and parks each, out of scalar registers, in two lanes of a vector register:
        // ~100 v_writelane there and as many v_readlane here for what one v_cmp each recomputes; interp_wave3_kernel has the same)
        int lane_v = lane0;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        // (the same for the chunk number and the trips' boundary rows: what hangs on them - the partials' addresses, "this trip has a
        // boundary" - is scalar arithmetic the loop can afford; hoisted, it is forty more scalars to park)
        int64_t g_v = g0;
        asm volatile("" : "+s"(g_v));
        const int64_t g = g_v;
#pragma unroll
        for (int j = 0; j < kTrips; j++) asm volatile("" : "+s"(tb_row[j]));
        const ColDesc &cd = p.cols[slot];
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd.values);
        const bool need_ts = kTw && cd.need_ts;
        if (slot != slot0) {
            const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
            for (int j = 0; j < kTrips; j++) stream_load_pair(vp + c0, j * 128 + 2 * lane, rows, vvec, vx[j], vy[j]);
            if (!kDense && !load_validity(cd)) {
                if (lane == 0) todo[g] = 1;
                return;
            }
        }
        double xv[kTrips], yv[kTrips];
        if (cd.type == BOWGPU_FLOAT64) {      // (a branch, not a select: Int64 columns convert per element - bowgetters.go:224-229 - and only they pay for it)
#pragma unroll
            for (int j = 0; j < kTrips; j++) { xv[j] = __longlong_as_double((long long)vx[j]); yv[j] = __longlong_as_double((long long)vy[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < kTrips; j++) { xv[j] = (double)(int64_t)vx[j]; yv[j] = (double)(int64_t)vy[j]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the lane's rows of trip j (nullable column): bits 0 / 1 = row x / row y has a value
        uint32_t vmask = 0;
        if (!kDense) {
#pragma unroll
            for (int j = 0; j < kTrips; j++) vmask |= (uint32_t)(((lane < 32 ? W[2 * j] : W[2 * j + 1]) >> ((2 * lane) & 63)) & 3ull) << (2 * j);
        }
        // (nullable, time-weighted) the points are the rows with a value.  For each trip, the first point BEHIND it (uniform; times: low
        // words, see small_t), found back to front; and the segment it belongs to (the tail segment = nb, another window = none)
        uint32_t cn_t[kTrips], cn_seg[kTrips];
        double cn_v[kTrips];
        if (!kDense && need_ts) {
            uint32_t c_t = 0, c_seg = 0xFFu;
            double c_v = 0.0;
            if (has_next) {
                // (nearly always among the 128 rows behind the chunk: their words are here already)
                int64_t rn = Wn[0] ? c0 + rows + __ffsll((long long)Wn[0]) - 1 : (Wn[1] ? c0 + rows + 64 + __ffsll((long long)Wn[1]) - 1 : -2);
                if (rn == -2) rn = col_next_valid(cd, c0 + rows, p.n);
                if (rn >= 0) {
                    const int64_t tn = p.ts[rn];
                    if (!(tn < (1ll << 53) && tn >= ts_first && (uint64_t)tn - (uint64_t)ts_first < 0xFFFFFFFFull)) {   // (see small_t)
                        if (lane == 0) todo[g] = 1;
                        return;
                    }
                    if (row_wid(p, tn) - wid0 == (uint64_t)rel_last) c_seg = (uint32_t)nb;
                    c_t = (uint32_t)(uint64_t)tn; c_v = bits_to_f64(vp[rn], cd.type);
                }
            }
            uint32_t seg_of_trip_end = (uint32_t)nb;     // the segment of the rows behind trip j's boundary, back to front
#pragma unroll
            for (int jj = 0; jj < kTrips; jj++) {
                const int j = kTrips - 1 - jj;
                cn_t[j] = c_t; cn_v[j] = c_v; cn_seg[j] = c_seg;
                if (W[2 * j] | W[2 * j + 1]) {
                    const int r0 = W[2 * j] ? __ffsll((long long)W[2 * j]) - 1 : 64 + __ffsll((long long)W[2 * j + 1]) - 1;   // the trip's first valid row
                    const int l0 = r0 >> 1;
                    const bool is_x = !(r0 & 1);
                    c_t = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(is_x ? tx[j] : ty[j]), l0);
                    c_v = readlane_f64(is_x ? xv[j] : yv[j], l0);
                    c_seg = (tb_row[j] >= 0 && j * 128 + r0 < tb_row[j]) ? seg_of_trip_end - 1u : seg_of_trip_end;
                }
                if (tb_row[j] >= 0) seg_of_trip_end--;
            }
        }
        // ---- the running partial over the trips, closed at every boundary
        LaneAcc<kNeed, !kDense> cur;        // (a nullable column: the lanes count their valid rows and keep the first / last themselves)
        cur.init();
        int seg_start = 0;
        uint32_t seg_rel = 0, seg_ix = 0;        // the open segment: its first row, its window relative to the chunk's first, its number
        bool first_seg = true;
        auto close = [&](int rb) {                 // the open segment ends in front of row rb (a boundary): its partial goes out
            const SV tot = lane_acc_finish<kNeed, !kDense>(cur, (uint32_t)(rb - seg_start), seg_start, rb - 1);
            if (lane == 63) {
                if (first_seg && prev_same) seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, tot, c0);      // the partial of the window that runs in
                else {
                    const int64_t sl = (int64_t)(wid0 + seg_rel - (uint64_t)p.wid_base);
                    if ((uint64_t)sl < (uint64_t)p.W) seg_store<kNeed>(wparts, sl * p.ncols + slot, tot, c0);
                }
            }
            cur.init();
            first_seg = false;
        };
#pragma unroll
        for (int j = 0; j < kTrips; j++) {
            const int i0 = j * 128 + 2 * lane;
            const uint32_t ix = (uint32_t)(i0 + 1);
            const int rbj = tb_row[j];                                  // (uniform) the boundary of this trip, -1: none
            const bool ax = rbj < 0 || i0 < rbj, ay = rbj < 0 || i0 + 1 < rbj;   // row in front of it
            double trx = 0.0, try_ = 0.0, stx = 0.0, sty = 0.0;
            bool okx = true, oky = true;
            if (kDense) {
                if (need_ts) {
                    // every row is a point: the next point of row x is row y of the lane, of row y row x of the next lane (lane 63: the first
                    // row of the next trip / the row behind the chunk)
                    uint32_t n_t;
                    double n_v;
                    if (j + 1 < kTrips) {
                        const int jn = j + 1 < kTrips ? j + 1 : j;
                        n_t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tx[jn]);
                        n_v = readlane_f64(xv[jn], 0);
                    } else {
                        n_t = (uint32_t)(uint64_t)ts_next;
                        n_v = has_next ? bits_to_f64(vp[c0 + rows], cd.type) : 0.0;
                    }
                    const uint32_t a_t = (uint32_t)__builtin_amdgcn_update_dpp((int)n_t, (int)(uint32_t)tx[j], 0x130, 0xf, 0xf, false);   // (low words: see small_t)
                    const double a_v = __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(n_v), __double2hiint(xv[j]), 0x130, 0xf, 0xf, false),
                                                        __builtin_amdgcn_update_dpp(__double2loint(n_v), __double2loint(xv[j]), 0x130, 0xf, 0xf, false));
                    const double dx = (double)((uint32_t)ty[j] - (uint32_t)tx[j]), dy = (double)(a_t - (uint32_t)ty[j]);
                    trx = (xv[j] + yv[j]) / 2 * dx; stx = xv[j] * dx;      // integral.go:24 / :55
                    try_ = (yv[j] + a_v) / 2 * dy; sty = yv[j] * dy;
                    // the pairs that straddle a window boundary contribute nothing: the chunk's last row unless its window runs on, and the
                    // row in front of this trip's / the next trip's boundary
                    if (j == kTrips - 1 && !(has_next && next_same) && lane == 63) { try_ = 0.0; sty = 0.0; }
                    const int rq0 = rbj - 1, rq1 = (j + 1 < kTrips ? tb_row[j + 1 < kTrips ? j + 1 : j] : -1) - 1;
                    if (rbj > 0 && (rq0 >> 7) == j) {
                        if (i0 == rq0) { trx = 0.0; stx = 0.0; }
                        if (i0 + 1 == rq0) { try_ = 0.0; sty = 0.0; }
                    }
                    if (rq1 >= 0 && (rq1 >> 7) == j && i0 + 1 == rq1) { try_ = 0.0; sty = 0.0; }     // (that boundary is the next trip's first row)
                }
            } else {
                okx = (vmask >> (2 * j)) & 1u; oky = (vmask >> (2 * j)) & 2u;
                if (need_ts) {
                    // the segment of the rows behind this trip's boundary = the number of boundaries up to and including this trip
                    uint32_t seg_hi = seg_ix + (rbj >= 0 ? 1u : 0u);
                    const uint32_t sgx = ax ? seg_ix : seg_hi, sgy = ay ? seg_ix : seg_hi;
                    const uint32_t xt = (uint32_t)tx[j], yt = (uint32_t)ty[j];
                    // the lane's own first point, the lanes that have one, the first of them above this lane
                    const uint32_t o_t = okx ? xt : yt, o_seg = okx ? sgx : sgy;
                    const double o_v = okx ? xv[j] : yv[j];
                    const uint64_t hasm = __ballot(okx || oky);
                    const uint64_t above = (hasm >> lane) >> 1;
                    const int src = (lane + 1 + (above ? __ffsll((long long)above) - 1 : 0)) << 2;
                    uint32_t a_t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)o_t), a_seg = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)o_seg);
                    double a_v = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(o_v)), __builtin_amdgcn_ds_bpermute(src, __double2loint(o_v)));
                    if (!above) { a_t = cn_t[j]; a_v = cn_v[j]; a_seg = cn_seg[j]; }
                    if (oky && a_seg == sgy) {
                        const double dy = (double)(a_t - yt);
                        try_ = (yv[j] + a_v) / 2 * dy; sty = yv[j] * dy;   // integral.go:24 / :55
                    }
                    if (okx) {
                        const uint32_t b_t = oky ? yt : a_t, b_seg = oky ? sgy : a_seg;
                        const double b_v = oky ? yv[j] : a_v;
                        if (b_seg == sgx) {
                            const double dx = (double)(b_t - xt);
                            trx = (xv[j] + b_v) / 2 * dx; stx = xv[j] * dx;
                        }
                    }
                }
            }
            if (rbj < 0) {
                if (kDense) { cur.add(xv[j], trx, stx, ix); cur.add(yv[j], try_, sty, ix + 1); }
                else {       // (a row without a value: its terms are zero already, its value counts as 0)
                    cur.add_v(okx, okx ? xv[j] : 0.0, xv[j], trx, stx, ix); cur.add_v(oky, oky ? yv[j] : 0.0, yv[j], try_, sty, ix + 1);
                }
            } else {
                cur.add_if(okx && ax, xv[j], trx, stx, ix); cur.add_if(oky && ay, yv[j], try_, sty, ix + 1);
                close(rbj);
                cur.add_if(okx && !ax, xv[j], trx, stx, ix); cur.add_if(oky && !ay, yv[j], try_, sty, ix + 1);
                seg_start = rbj; seg_rel = (uint32_t)tb_rel[j]; seg_ix++;
            }
            __builtin_amdgcn_sched_barrier(0);     // (one trip after the other: interleaved, the four trips' temporaries cost ~100 registers)
        }
        // the segment that reaches the chunk's end
        const SV tail = lane_acc_finish<kNeed, !kDense>(cur, (uint32_t)(rows - seg_start), seg_start, rows - 1);
        if (lane == 63) {
            const SV id = seg_identity<kNeed>();
            if (through) {
                seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, tail, c0);
                seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
            } else {
                if (hr == 0) seg_store<kNeed>(parts, (2 * g) * p.ncols + slot, id, c0);
                if (tail_open) seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, tail, c0);
                else {
                    seg_store<kNeed>(parts, (2 * g + 1) * p.ncols + slot, id, c0);
                    if ((uint64_t)tail_slot < (uint64_t)p.W) seg_store<kNeed>(wparts, tail_slot * p.ncols + slot, tail, c0);
                }
            }
        }
    }
}

// One lane per window of the call: the empty ones (no record), the ones inside one chunk (their partial is ready), and the ones
// over a few chunks (walk the heads that follow the chunk they start in, merge the range of partials).  Windows that run
// further than kStreamScan chunks are queued for long_final_block_kernel: entries / off0 / off1 [i], i < *n_leftover.
__global__ __launch_bounds__(256) void stream_final_kernel(const AggParams p, const int64_t nchunks, const ChunkMeta *meta,
                                                           const Part *parts, const WinRec *recs, const Part *wparts,
                                                           LongEntry *entries, int64_t *off0, int64_t *off1,
                                                           unsigned long long *n_leftover, const int lite) {
    // A lane's life here is a chain of dependent loads (record -> partials -> the first / last value of the window): what can be
    // loaded early is (the next two records with the lane's own; up to four partials at once), and the validity bits do not go
    // through atomics - lane k is window k, so the ballots of a wavefront ARE two words of every output's bitmap.
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = k < p.W;
    const bool defer_bits = !p.bits_preset;      // (bitmaps preset to all-null: whole words can be stored)
    uint32_t vbits = 0;                          // bit i: the output of aggregation i is valid for this lane's window
    if (in) {
        const WinRec rec = recs[k];
        const int64_t rn1 = k + 1 < p.W ? recs[k + 1].r0 : p.n, rn2 = k + 2 < p.W ? recs[k + 2].r0 : p.n;   // (behind the last window: the end of the rows)
        LongEntry le;
        le.wid = (uint64_t)(p.wid_base + k); le.r0 = rec.r0; le.r1 = -1; le.next_wid = le.wid + 1; le.incl_row = 0; le.dead = 0;
        bool done = false;
        if (rec.r0 < 0) {   // no rows: the reducers' values for an empty window (window k itself: "0 windows behind it")
            for (int slot = -1; slot < p.ncols; slot++)
                if (p.pass_mask[slot + 1]) emit_empties(p, slot, le, 0, 0, 1);
            done = true;
        }
        const int64_t g = rec.chunk;
        int64_t w0 = 2 * g + 1, w1 = w0;
        if (!done) {
            if (rec.rows >= 0) le.r1 = rec.r0 + rec.rows;
            else {
                // the window ends where the next window that has rows starts (rows are ascending, the windows partition them) ...
                if (rn1 >= 0) le.r1 = rn1;
                else if (rn2 >= 0) le.r1 = rn2;
                else
                    for (int64_t kk = k + 3; kk <= k + 8; kk++) {
                        if (kk >= p.W) { le.r1 = p.n; break; }
                        const int64_t rs = recs[kk].r0;
                        if (rs >= 0) { le.r1 = rs; break; }
                    }
                if (le.r1 >= 0) w1 = 2 * ((le.r1 - 1) / kStreamRows) + 1;    // (its last row's chunk: the head partial of that chunk is its last partial)
                // ... or, behind a run of empty windows, where the first chunk head ends that does not cover its chunk
                else for (int64_t c = g + 1; c <= g + kStreamScan; c++) {
                    if (c >= nchunks) { le.r1 = p.n; w1 = 2 * nchunks; break; }
                    const int64_t rows_c = p.n - c * kStreamRows < kStreamRows ? p.n - c * kStreamRows : kStreamRows;
                    const int64_t hr = meta[c].head_rows;
                    if (hr < rows_c) { le.r1 = c * kStreamRows + hr; w1 = 2 * c + 1; break; }
                }
                if (le.r1 < 0) {       // (long_final_block_kernel finishes it, validity bits included: this lane's stay 0 here)
                    const unsigned long long i = atomicAdd(n_leftover, 1ull);
                    entries[i] = le;
                    off0[i] = w0;
                    off1[i] = w1;
                    done = true;
                }
            }
        }
        if (!done) {
            if (p.inclusive || p.pre_rows) entry_close(p, le);   // (the row behind the window / rows below s0 matter only then)
            for (int slot = -1; slot < p.ncols; slot++) {
                if (p.pass_mask[slot + 1] == 0) continue;
                Part acc;
                part_init(acc);
                if (slot >= 0 && (p.pass_flags[slot + 1] & kPassNeedVals)) {
                    if (rec.rows >= 0) acc = part_at(wparts, k * p.ncols + slot, lite);
                    else {
                        // four partials at a time (a 1000-row window has three to five): their loads go out together
                        for (int64_t w = w0; w < w1; w += 4) {
                            const int64_t wl = w1 - 1;
                            const Part q0 = part_at(parts, w * p.ncols + slot, lite), q1 = part_at(parts, (w + 1 < wl ? w + 1 : wl) * p.ncols + slot, lite),
                                       q2 = part_at(parts, (w + 2 < wl ? w + 2 : wl) * p.ncols + slot, lite), q3 = part_at(parts, (w + 3 < wl ? w + 3 : wl) * p.ncols + slot, lite);
                            part_merge(acc, q0);
                            if (w + 1 < w1) part_merge(acc, q1);
                            if (w + 2 < w1) part_merge(acc, q2);
                            if (w + 3 < w1) part_merge(acc, q3);
                        }
                    }
                }
                emit_window(p, slot, le, acc, defer_bits ? &vbits : nullptr);
            }
        }
    }
    if (defer_bits) {
        // (every lane of the wavefront arrives here: windows 64 w .. 64 w + 63 = bitmap words 2 w, 2 w + 1 of every output)
        const int64_t k0 = k - lane;
        for (int a = 0; a < p.naggs; a++) {
            uint32_t *bits = p.aggs[a].out_valid;
            const uint64_t m = __ballot((vbits >> a) & 1u);
            if (!bits) continue;
            if (lane == 0 && k0 < p.W) bits[k0 >> 5] = (uint32_t)m;
            if (lane == 32 && k0 + 32 < p.W) bits[(k0 + 32) >> 5] = (uint32_t)(m >> 32);
        }
    }
}

// The same for windows of many chunks: kT (64 / 256) threads per window - the next window that has rows is looked for 64 records
// at a time, the window's partials are merged kT at a time, then over the lanes (and the workgroup's four wavefronts).
template <int kT>
__global__ __launch_bounds__(256) void stream_final_wide_kernel(const AggParams p, const int64_t nchunks, const ChunkMeta *meta, const Part *parts,
                                                                const WinRec *recs, const Part *wparts, const int lite) {
    __shared__ Part red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = kT == 64 ? lane : tid;
    const int64_t k = kT == 64 ? (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6) : (int64_t)blockIdx.x;
    if (k >= p.W) return;
    const WinRec rec = recs[k];
    LongEntry le;
    le.wid = (uint64_t)(p.wid_base + k); le.r0 = rec.r0; le.r1 = -1; le.next_wid = le.wid + 1; le.incl_row = 0; le.dead = 0;
    if (rec.r0 < 0) {
        if (t == 0)
            for (int slot = -1; slot < p.ncols; slot++)
                if (p.pass_mask[slot + 1]) emit_empties(p, slot, le, 0, 0, 1);
        return;
    }
    const int64_t w0 = 2 * (int64_t)rec.chunk + 1;
    int64_t w1 = w0;
    if (rec.rows >= 0) le.r1 = rec.r0 + rec.rows;
    else {
        for (int64_t kb = k + 1; le.r1 < 0; kb += 64) {
            const int64_t kk = kb + lane;
            const int64_t rs = kk < p.W ? recs[kk].r0 : p.n;       // (behind the call's last window: the end of the rows)
            const uint64_t m = __ballot(rs >= 0);
            if (m) {
                const int l = __ffsll((long long)m) - 1;
                le.r1 = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)rs, l) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)rs >> 32), l) << 32);
            }
        }
        w1 = 2 * ((le.r1 - 1) / kStreamRows) + 1;
    }
    if (p.inclusive || p.pre_rows) entry_close(p, le);
    for (int slot = -1; slot < p.ncols; slot++) {
        if (p.pass_mask[slot + 1] == 0) continue;
        Part acc;
        part_init(acc);
        if (slot >= 0 && (p.pass_flags[slot + 1] & kPassNeedVals)) {
            if (rec.rows >= 0) {
                if (t == 0) acc = part_at(wparts, k * p.ncols + slot, lite);
            } else {
                for (int64_t w = w0 + t; w < w1; w += kT) part_merge(acc, part_at(parts, w * p.ncols + slot, lite));
                if (kT == 256) block_reduce(acc, red, tid);
                else
                    for (int o = 32; o > 0; o >>= 1) {
                        Part other;
                        other.sum = __shfl_down(acc.sum, o); other.trap = __shfl_down(acc.trap, o); other.step = __shfl_down(acc.step, o);
                        other.vmin = __shfl_down(acc.vmin, o); other.vmax = __shfl_down(acc.vmax, o);
                        other.count = __shfl_down((long long)acc.count, o);
                        other.min_idx = __shfl_down((long long)acc.min_idx, o); other.max_idx = __shfl_down((long long)acc.max_idx, o);
                        other.first_idx = __shfl_down((long long)acc.first_idx, o); other.last_idx = __shfl_down((long long)acc.last_idx, o);
                        if (lane + o < 64) part_merge(acc, other);
                    }
            }
        }
        if (t == 0) emit_window(p, slot, le, acc);
    }
}

size_t long_entry_size() { return sizeof(LongEntry); }
size_t long_part_size() { return sizeof(Part); }

// starts == nullptr: every window of the call is an entry (long-only pipeline), preceded by the order check of the interval column
int launch_long_queue(Ctx *c, const AggParams &p, int64_t capacity, void *big_entries, int32_t *big_nchunks, int64_t walk_max_rows, bool strict) {
    if (capacity <= 0) return 0;
    hipLaunchKernelGGL(long_queue_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, c->stream, p,
                       reinterpret_cast<LongEntry *>(big_entries), big_nchunks, walk_max_rows, strict ? 1 : 0);
    BG_HIP(hipGetLastError());
    return 0;
}

// n_given > 0: entries / nchunks [0, n_given) are already there (long_queue_kernel's compact list of the windows it did not walk)
int launch_long_windows_v2(Ctx *c, const AggParams &p, const LongListStarts *starts, void *entries, int32_t *nchunks,
                           int64_t *offsets, int64_t *block_sums, int64_t *d_total, int32_t *work_entry, void *partials,
                           int64_t max_work, bool strict, int64_t n_given) {
    const int64_t n_long = n_given > 0 ? n_given : starts ? starts->start[kLongLists] : p.W;
    if (n_long <= 0) return 0;
    bool walk_checks = false;
    if (n_given > 0) {
    } else if (starts) {
        hipLaunchKernelGGL(long_bounds_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p, *starts,
                           reinterpret_cast<LongEntry *>(entries), nchunks);
    } else {
        // the order check of the interval column: its own pass - or, under strict_order when some column pass walks the timestamps anyway and
        // no row lies below s0 (no window is skipped as dead), the walks themselves (walk_entry check_order)
        for (int sl = 0; sl < p.ncols; sl++) walk_checks = walk_checks || (p.pass_mask[sl + 1] && p.cols[sl].need_ts);
        walk_checks = walk_checks && strict && !p.pre_rows;
        if (!walk_checks) {
            int64_t g = (p.n / 2 + 255) / 256;
            if (g > 256 * 32) g = 256 * 32;
            if (g < 1) g = 1;
            hipLaunchKernelGGL(ts_sorted_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, p.ts, p.n, p.status);
        }
        hipLaunchKernelGGL(long_bounds_all_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p,
                           reinterpret_cast<LongEntry *>(entries), nchunks);
    }
    if (strict) {   // every window in row order, one lane each (queued windows: + the empty windows behind them)
        hipLaunchKernelGGL(long_strict_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p, n_long,
                           reinterpret_cast<const LongEntry *>(entries), starts ? 1 : 0, walk_checks ? 1 : 0);
        BG_HIP(hipGetLastError());
        return 0;
    }
    BG_TRY(launch_exclusive_scan(c, nchunks, n_long, offsets, block_sums, d_total));
    hipLaunchKernelGGL(long_map_kernel, dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, c->stream, n_long, offsets, work_entry);
    hipLaunchKernelGGL(long_partial_kernel, dim3((unsigned)((max_work + 3) / 4)), dim3(256), 0, c->stream, p, n_long,
                       reinterpret_cast<const LongEntry *>(entries), offsets, work_entry, reinterpret_cast<Part *>(partials));
    // the chunk -> window map has served long_partial_kernel: its buffer now lists the windows left to the workgroup kernel
    unsigned long long *n_leftover = reinterpret_cast<unsigned long long *>(d_total);
    BG_HIP(hipMemsetAsync(n_leftover, 0, 8, c->stream));
    hipLaunchKernelGGL(long_final_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p, n_long,
                       reinterpret_cast<const LongEntry *>(entries), offsets, reinterpret_cast<const Part *>(partials), work_entry, n_leftover);
    hipLaunchKernelGGL(long_final_block_kernel, dim3((unsigned)(n_long < 2048 ? n_long : 2048)), dim3(256), 0, c->stream, p,
                       reinterpret_cast<const LongEntry *>(entries), offsets, offsets + 1, reinterpret_cast<const Part *>(partials),
                       (const int32_t *)work_entry, n_leftover, (int64_t)0, 0);
    BG_HIP(hipGetLastError());
    return 0;
}

// The streaming form: every window of the call in one read of the rows.  Workspace (bytes): long_stream_workspace(n, W, ncols).
static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t long_stream_workspace(int64_t n, int64_t W, int ncols) {
    const size_t nch = (size_t)((n + kStreamRows - 1) / kStreamRows), nc = (size_t)(ncols > 0 ? ncols : 1), w = (size_t)(W > 0 ? W : 1);
    return up256(nch * sizeof(ChunkMeta)) + up256(nch * 2 * nc * sizeof(Part)) + up256(w * sizeof(WinRec)) + up256(w * nc * sizeof(Part)) +
           up256(nch * sizeof(LongEntry)) + 2 * up256(nch * 8) + 256;
}

int launch_long_stream(Ctx *c, const AggParams &p, void *workspace) {
    if (p.n <= 0 || p.W <= 0) return 0;
    const size_t nch = (size_t)((p.n + kStreamRows - 1) / kStreamRows), nc = (size_t)(p.ncols > 0 ? p.ncols : 1), w = (size_t)p.W;
    char *q = reinterpret_cast<char *>(workspace);
    ChunkMeta *meta = reinterpret_cast<ChunkMeta *>(q); q += up256(nch * sizeof(ChunkMeta));
    Part *parts = reinterpret_cast<Part *>(q); q += up256(nch * 2 * nc * sizeof(Part));
    WinRec *recs = reinterpret_cast<WinRec *>(q); q += up256(w * sizeof(WinRec));
    Part *wparts = reinterpret_cast<Part *>(q); q += up256(w * nc * sizeof(Part));
    LongEntry *entries = reinterpret_cast<LongEntry *>(q); q += up256(nch * sizeof(LongEntry));
    int64_t *off0 = reinterpret_cast<int64_t *>(q); q += up256(nch * 8);
    int64_t *off1 = reinterpret_cast<int64_t *>(q); q += up256(nch * 8);
    unsigned long long *n_leftover = reinterpret_cast<unsigned long long *>(q);
    BG_HIP(hipMemsetAsync(n_leftover, 0, 8, c->stream));
    BG_HIP(hipMemsetAsync(recs, 0xFF, w * sizeof(WinRec), c->stream));   // r0 = -1: no rows
    int need = 0;
    bool all_dense = true;
    for (int sl = 0; sl < p.ncols; sl++) {
        if (p.pass_flags[sl + 1] & kPassMinMax) need |= 1;
        if (p.pass_flags[sl + 1] & kPassFirstLast) need |= 2;
        if (p.cols[sl].need_ts) need |= 4;
        if (p.pass_mask[sl + 1] && (p.pass_flags[sl + 1] & kPassNeedVals) && p.cols[sl].vbits) all_dense = false;
    }
    const dim3 grid((unsigned)((nch + 3) / 4)), block(256);
    const int64_t nchunks = (int64_t)nch;
    // Windows of at least a trip's rows (128) on average - every call that gets here: long_short_kernel first, then the general form
    // for the chunks it flagged (two boundaries inside one 128-row trip, ...).  Measured down to 128-row windows on regular and on
    // irregular data (1e8 rows, Min + Max: 0.55 / 0.62 ms against 0.72 / 0.74 for the general form alone; scratch/longw_sweep.py).
    const bool short_first = p.n / p.W >= 128;
    const dim3 fgrid((unsigned)((nch + 255) / 256));      // a wavefront per 64 chunk flags
    uint8_t *todo = reinterpret_cast<uint8_t *>(entries);   // (entries is written by stream_final_kernel, behind these launches: borrowed)
#define BG_LONG_STREAM(K)                                                                                                              \
    do {                                                                                                                               \
        if (short_first && all_dense) hipLaunchKernelGGL((long_short_kernel<K, true>), grid, block, 0, c->stream, p, nchunks, meta, parts, recs, wparts, todo); \
        else if (short_first) hipLaunchKernelGGL((long_short_kernel<K, false>), grid, block, 0, c->stream, p, nchunks, meta, parts, recs, wparts, todo); \
        if (short_first) hipLaunchKernelGGL((long_stream_kernel<K, true>), fgrid, block, 0, c->stream, p, nchunks, meta, parts, recs, wparts, (const uint8_t *)todo); \
        else hipLaunchKernelGGL((long_stream_kernel<K, false>), grid, block, 0, c->stream, p, nchunks, meta, parts, recs, wparts, (const uint8_t *)nullptr); \
    } while (0)
    if (need == 0) BG_LONG_STREAM(0);
    else if (!(need & 4) && !(need & 1)) BG_LONG_STREAM(2);
    else if (!(need & 4)) BG_LONG_STREAM(3);
    else if (!(need & 1)) BG_LONG_STREAM(6);
    else BG_LONG_STREAM(7);
#undef BG_LONG_STREAM
    // the windows' partials merged and the reducers finished: a lane per window while windows span a few chunks, a wavefront / a
    // workgroup per window beyond that
    const int64_t avg_rows = p.n / p.W;
#ifndef BOWGPU_WIDE_FINAL_ROWS
#define BOWGPU_WIDE_FINAL_ROWS (8 * kStreamRows)
#endif
    constexpr int64_t kStreamWideFinalRows = BOWGPU_WIDE_FINAL_ROWS;
    if (avg_rows < kStreamWideFinalRows) {
        hipLaunchKernelGGL(stream_final_kernel, dim3((unsigned)((p.W + 255) / 256)), dim3(256), 0, c->stream, p, nchunks, meta, parts, recs, wparts,
                           entries, off0, off1, n_leftover, need == 0 ? 1 : 0);
        hipLaunchKernelGGL(long_final_block_kernel, dim3((unsigned)(nch < 2048 ? nch : 2048)), dim3(256), 0, c->stream, p, entries, off0, off1,
                           parts, (const int32_t *)nullptr, n_leftover, nchunks, need == 0 ? 1 : 0);
    } else if (avg_rows < 128 * kStreamRows)
        hipLaunchKernelGGL(stream_final_wide_kernel<64>, dim3((unsigned)((p.W + 3) / 4)), dim3(256), 0, c->stream, p, nchunks, meta, parts, recs,
                           wparts, need == 0 ? 1 : 0);
    else
        hipLaunchKernelGGL(stream_final_wide_kernel<256>, dim3((unsigned)p.W), dim3(256), 0, c->stream, p, nchunks, meta, parts, recs, wparts,
                           need == 0 ? 1 : 0);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
