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
#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kChunkRows = kLongChunkRows;

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
    uint64_t next_wid = (uint64_t)(p.wid_base + p.W);
    bool at_start = false;
    if (r1 < p.n) {
        const int64_t t = p.ts[r1];
        next_wid = magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
        at_start = (t == p.s0 + (int64_t)(next_wid * (uint64_t)p.interval));
    }
    LongEntry le;
    le.wid = wid; le.r0 = r0; le.r1 = r1; le.next_wid = next_wid;
    le.incl_row = (p.inclusive && r1 < p.n && at_start && next_wid == wid + 1) ? 1 : 0;
    le.dead = (p.pre_rows && r0 == 0 && !(p.ts[r1 - 1] >= p.s0 || le.incl_row)) ? 1 : 0;
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
        const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;
        const bool need_vals = cd && (p.pass_flags[slot + 1] & kPassNeedVals);
        const uint64_t *vp = cd ? reinterpret_cast<const uint64_t *>(cd->values) : nullptr;

        // the outputs of window `w` (an entry) from its merged partial
        auto finish = [&](const LongEntry &w, const Part &acc) {
            const int64_t win_start = p.s0 + (int64_t)(w.wid * (uint64_t)p.interval);
            const int64_t oslot = (int64_t)(w.wid - (uint64_t)p.wid_base);
            const int64_t len = w.dead ? 0 : w.r1 - w.r0;
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
                if (a.out_valid) {
                    if (v.valid) atomicOr(&a.out_valid[oslot >> 5], 1u << (oslot & 31));
                    else if (p.bits_preset) atomicAnd(&a.out_valid[oslot >> 5], ~(1u << (oslot & 31)));
                }
            }
        };
        // the empty windows gk = first, first + step, ... <= gap behind window `w`
        auto empties = [&](const LongEntry &w, int64_t gap, int64_t first, int64_t step) {
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
        };

        if (simple) {
            Part acc;
            part_init(acc);
            if (need_vals)
                for (int64_t w = w0; w < w1; w++) part_merge(acc, partials[w * p.ncols + slot]);  // (at most kLaneParts, in order)
            finish(le, acc);
            empties(le, my_gap, 1, 1);
        }
    }
}

// The windows long_final_kernel leaves (many chunk partials, or a long run of empty windows behind them): one workgroup per
// window merges its chunk partials in a fixed shape, writes the outputs, then the empty windows that follow it
__global__ __launch_bounds__(256) void long_final_block_kernel(const AggParams p, const LongEntry *entries, const int64_t *offsets,
                                                               const Part *partials, const int32_t *leftover,
                                                               const unsigned long long *n_leftover) {
    __shared__ Part red[4];
    const int64_t nleft = (int64_t)*n_leftover;  // (usually 0: the launch then costs a few microseconds)
    for (int64_t i = blockIdx.x; i < nleft; i += gridDim.x) {
    const int64_t e = leftover[i];
    const LongEntry le = entries[e];
    const int lane = threadIdx.x;
    const int64_t win_start = p.s0 + (int64_t)(le.wid * (uint64_t)p.interval);
    const int64_t oslot = (int64_t)(le.wid - (uint64_t)p.wid_base);
    const int64_t len = le.dead ? 0 : le.r1 - le.r0;

    for (int slot = -1; slot < p.ncols; slot++) {
        const unsigned my_mask = p.pass_mask[slot + 1];
        if (my_mask == 0) continue;
        const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;
        Part acc;
        part_init(acc);
        if (cd && (p.pass_flags[slot + 1] & kPassNeedVals)) {
            const int64_t w0 = offsets[e], w1 = offsets[e + 1];
            if (w1 - w0 == 1) {  // the common medium-sized window: nothing to merge
                if (lane == 0) acc = partials[w0 * p.ncols + slot];
            } else {
                for (int64_t w = w0 + lane; w < w1; w += 256) part_merge(acc, partials[w * p.ncols + slot]);
                block_reduce(acc, red, lane);
                __syncthreads();
            }
        }
        if (lane == 0 && (uint64_t)oslot < (uint64_t)p.W) {
            // rebuild the reference's running state from the order-free partial
            const uint64_t *vp = cd ? reinterpret_cast<const uint64_t *>(cd->values) : nullptr;
            Stats st;
            stats_init(st);
            auto fill = [&](Stats &s, const Part &a) {
                s.sum = a.sum; s.count = a.count; s.has_value = a.first_idx >= 0;
                if (s.has_value) {
                    s.first_bits = vp[a.first_idx]; s.last_bits = vp[a.last_idx];
                    const double f = bits_to_f64(s.first_bits, col_type);
                    // minmax.go:16-28: seeded by the first valid value; a NaN seed is never replaced
                    s.vmin = (f != f) ? f : (a.min_idx >= 0 ? a.vmin : f);
                    s.vmax = (f != f) ? f : (a.max_idx >= 0 ? a.vmax : f);
                    s.has_point = 1; s.pt = (double)p.ts[a.last_idx]; s.pv = bits_to_f64(s.last_bits, col_type);
                    s.integ_trap = a.trap; s.integ_step = a.step; s.has_pair = a.count >= 2;
                }
            };
            fill(st, acc);
            // the state including the inclusive row, for the reducers that want it (aggregation.go:207-211)
            Stats st_incl = st;
            if (le.incl_row && cd && (p.pass_flags[slot + 1] & kPassNeedVals) && col_valid(*cd, le.r1)) {
                const uint64_t raw = vp[le.r1];
                const double x = bits_to_f64(raw, col_type);
                stats_value<false>(st_incl, x, raw);
                stats_point(st_incl, (double)p.ts[le.r1], x);
            }
            for (unsigned m = my_mask; m; m &= m - 1) {
                const AggDesc &a = p.aggs[__ffs(m) - 1];
                const bool inc = a.kind == BOWGPU_AGG_INTEGRAL_TRAPEZOID || a.kind == BOWGPU_AGG_WAVG_LINEAR;
                Val v = finish_val(reduce_val(a.kind, inc ? st_incl : st, inc ? len + le.incl_row : len, win_start, p.interval,
                                              col_type == BOWGPU_INT64), a);
                reinterpret_cast<uint64_t *>(a.out_values)[oslot] = v.bits;
                if (a.out_valid) {
                    if (v.valid) atomicOr(&a.out_valid[oslot >> 5], 1u << (oslot & 31));
                    else if (p.bits_preset) atomicAnd(&a.out_valid[oslot >> 5], ~(1u << (oslot & 31)));
                }
            }
        }
        // empty windows after this one
        const int64_t gap = (int64_t)(le.next_wid - le.wid) - 1;
        Stats em;
        stats_init(em);
        for (int64_t gk = 1 + lane; gk <= gap; gk += 256) {
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
    __syncthreads();  // (red[] is reused by the next entry)
    }
}

size_t long_entry_size() { return sizeof(LongEntry); }
size_t long_part_size() { return sizeof(Part); }

// starts == nullptr: every window of the call is an entry (long-only pipeline), preceded by the order check of the interval column
int launch_long_windows_v2(Ctx *c, const AggParams &p, const LongListStarts *starts, void *entries, int32_t *nchunks,
                           int64_t *offsets, int64_t *block_sums, int64_t *d_total, int32_t *work_entry, void *partials,
                           int64_t max_work) {
    const int64_t n_long = starts ? starts->start[kLongLists] : p.W;
    if (n_long <= 0) return 0;
    if (starts) {
        hipLaunchKernelGGL(long_bounds_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p, *starts,
                           reinterpret_cast<LongEntry *>(entries), nchunks);
    } else {
        int64_t g = (p.n / 2 + 255) / 256;
        if (g > 256 * 32) g = 256 * 32;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(ts_sorted_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, p.ts, p.n, p.status);
        hipLaunchKernelGGL(long_bounds_all_kernel, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, p,
                           reinterpret_cast<LongEntry *>(entries), nchunks);
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
                       reinterpret_cast<const LongEntry *>(entries), offsets, reinterpret_cast<const Part *>(partials), work_entry, n_leftover);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
