// ts_nulls.hip — the device path for an interval column WITH NULLS (Rolling.Aggregate; exclusive and inclusive windows).
//
// The reference's window walk (rolling/rolling.go:177-239) skips a row whose interval value is null (:190-193): the row neither
// ends a window nor extends it, but the window's slice is [first row, last taken row + 1) (:224-228), so a null row that lies
// BETWEEN two taken rows of one window sits inside the slice - the reducers see its value columns (sum.go:15-22 reads every row
// of w.Bow) - while the null rows behind a window's last taken row, in front of the next window's first row, belong to no slice.
// countWindows (rolling.go:143-154) measures from the last VALID timestamp, and HasNext (:162-173) ends the iteration at once
// when the physically last timestamp is null (api.cpp handles that case: every output slot stays nil).
//
// As a statement about rows (ascending valid timestamps; the tile kernels check that): with p(i) / q(i) the nearest row before /
// after a null row i whose timestamp is valid, i belongs to window w iff wid(ts[p]) == wid(ts[q]) == w.  So the tile kernels can
// run UNCHANGED on
//     ts_eff[i]  = ts[i] for a valid row, ts[p(i)] for a null one                  (dense, ascending, no validity)
//     keep       = 1 for a valid row, [wid(ts[p]) == wid(ts[q])] for a null one     (one bit per row)
// with every value column's validity ANDed with `keep` (a row outside every slice is never read) and, for the time-weighted
// reducers - their points are the rows where timestamp AND value are valid (bowgetters.go:299-311 GetNextFloat64s) - with the
// interval column's own validity too.  Cost: one more pass over the interval column (8 B read + 8 B written per row) and n / 8 bytes
// per bitmap; the reference's Go loop runs at ~1e7 rows/s.  NumRows (the test closure of aggregation_test.go:28-31 counts the rows of
// the slice, valid or not) is Count over the keep bits.
#include "bitmap_device.h"
#include "agg_device.h"
#include "interp_device.h"

namespace bowgpu {

namespace {

__device__ __forceinline__ uint64_t wid_of(int64_t t, int64_t s0, const MagicDiv &magic) {
    return t < s0 ? 0ull : magic_div((uint64_t)t - (uint64_t)s0, magic);   // rows below s0 ride in window 0 (rolling.go:194-196)
}

// INCLUSIVE iteration (Options.Inclusive, or a reducer of the call asks for it - IntegralTrapezoid / WeightedAverageLinear,
// aggregation.go:183-185 - and then for every reducer of the call: the others see each window through UnsetInclusive, window.go:23-31).
// A window w also takes the first valid row whose timestamp equals its end E (rolling.go:201-209), so
//   - the null rows between its last row below E and that row lie inside its slice:   keep |= ts[q] == start(wid(ts[p]) + 1)
//   - the next window starts at `rowIndex - 1` (:214-218), the row in front of the one that ended the scan.  That is the inclusive row
//     itself - unless null rows follow it: then it is the LAST of those null rows, and the next window, w + 1, begins there WITHOUT the
//     row that sits on its start.  Such a row i ("a row on a window start, the first with that timestamp, not in window 0,
//     with a null timestamp right behind it") belongs to window w only, and there only as the inclusive row; the null rows behind it
//     belong to no slice, except the last one, which opens the slice of w + 1 when w + 1 takes a row at all.
// In the rewritten call: such a row is invisible (validity 0) to every reducer that reads windows through UnsetInclusive - they see it
// neither in w (the inclusive row is dropped) nor in w + 1 - which is every reducer but the two that need inclusive windows; for
// those two the row must stay the end point of w, so the tile kernels compute w + 1 WITH it, and ts_quirk_fix_kernel below recomputes
// their outputs for those windows afterwards, walking the rows the way the reference does.
__device__ __forceinline__ bool on_later_start(int64_t t, int64_t s0, int64_t interval, const MagicDiv &magic) {
    if (t < s0) return false;
    const uint64_t d = (uint64_t)t - (uint64_t)s0, q = magic_div(d, magic);
    return q >= 1 && q * (uint64_t)interval == d;
}
// row i (timestamp valid) is such a row
__device__ __forceinline__ bool quirk_row(const int64_t *__restrict__ ts, const uint32_t *__restrict__ tbits, int64_t tbit0, int64_t n, const NbrIndex &ix,
                                          int64_t s0, int64_t interval, const MagicDiv &magic, int64_t i) {
    if (i + 1 >= n || bit_at(tbits, tbit0, i + 1)) return false;
    const int64_t t = ts[i];
    if (!on_later_start(t, s0, interval, magic)) return false;
    const int64_t pp = prev_valid_ix(tbits, tbit0, n, i - 1, ix);
    return pp < 0 || ts[pp] < t;
}

// one lane per row, 64 rows per wavefront: the keep bits of a wavefront's rows are one 64-bit word of the bitmap (bit 0 = row 0).
// plain (inclusive only, else nullptr): the interval column's validity without the rows described above.
__global__ __launch_bounds__(256) void ts_nullfill_kernel(const int64_t *__restrict__ ts, const uint32_t *__restrict__ tbits, const int64_t tbit0,
                                                          const int64_t n, const NbrIndex ix, const int64_t s0, const int64_t interval, const MagicDiv magic,
                                                          const int inclusive, int64_t *__restrict__ ts_eff, uint64_t *__restrict__ keep,
                                                          uint64_t *__restrict__ plain, unsigned long long *n_dropped) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool k = false, dropped = false, quirk = false, tv = false;
    if (i < n) {
        if (bit_at(tbits, tbit0, i)) {
            ts_eff[i] = ts[i];
            tv = true;
            quirk = inclusive && quirk_row(ts, tbits, tbit0, n, ix, s0, interval, magic, i);
            k = !quirk || inclusive == 2;     // (Interpolate keeps such a row: it is the last row of its window's slice)
            if (quirk && inclusive == 2) {
                // two shapes the compacted call cannot express (interp_null_ts declines them): the next valid timestamp EQUALS this row's - the
                // next window then has its start and adds no row, an ordinary call would copy this row a second time - and a row on -1,
                // the value interpolateWindow uses for "no first value" (interpolation.go:119-127)
                const int64_t b = next_valid_ix(tbits, tbit0, n, i + 1, ix);
                if (ts[i] == -1 || (b >= 0 && ts[b] == ts[i])) atomicAdd(n_dropped + 1, 1ull);
            }
        } else {
            const int64_t p = prev_valid_ix(tbits, tbit0, n, i - 1, ix);   // (row 0 is valid: the constructor checked it, rolling.go:89-93)
            const int64_t q = next_valid_ix(tbits, tbit0, n, i + 1, ix);
            const int64_t tp = p >= 0 ? ts[p] : s0;
            ts_eff[i] = tp;
            if (p >= 0 && q >= 0) {
                const int64_t tq = ts[q];
                const uint64_t wp = wid_of(tp, s0, magic);
                k = wp == wid_of(tq, s0, magic);
                if (inclusive) {
                    k = k || (tq >= s0 && (uint64_t)tq - (uint64_t)s0 == (wp + 1) * (uint64_t)interval);
                    if (quirk_row(ts, tbits, tbit0, n, ix, s0, interval, magic, p)) k = k && i == q - 1;
                }
            }
            dropped = !k;
        }
    }
    const unsigned long long m = __ballot(k), d = __ballot(dropped || (quirk && inclusive != 2)), pm = __ballot(tv && !quirk);
    if ((threadIdx.x & 63) == 0) {
        if (i < n) { keep[i >> 6] = m; if (plain) plain[i >> 6] = pm; }
        if (d) atomicAdd(n_dropped, (unsigned long long)__popcll(d));
    }
}

// The two reducers that need inclusive windows, for the windows described above: thread i finds out whether row i is such a row; if
// so it walks the window that starts at ts[i] as rolling.go:188-212 does from the row in front of the next valid timestamp - rows below
// the window's end, then the first row on it - and integral.go:14-31 over the points among them (timestamp and value valid:
// bowgetters.go:299-311), and overwrites the window's slot in the outputs of those reducers.
__global__ __launch_bounds__(256) void ts_quirk_fix_kernel(const int64_t *__restrict__ ts, const uint32_t *__restrict__ tbits, const int64_t tbit0,
                                                           const int64_t n, const NbrIndex ix, const int64_t s0, const int64_t interval, const MagicDiv magic,
                                                           const int64_t W, const QuirkFix fx, unsigned long long *n_fixed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !bit_at(tbits, tbit0, i)) return;
    if (!quirk_row(ts, tbits, tbit0, n, ix, s0, interval, magic, i)) return;
    const int64_t start = ts[i];
    const uint64_t slot = magic_div((uint64_t)start - (uint64_t)s0, magic);
    if (slot >= (uint64_t)W) return;
    const int64_t end = (int64_t)((uint64_t)start + (uint64_t)interval);
    const bool open_end = end < start;                 // (the end wraps: no timestamp reaches it)
    const int64_t b = next_valid_ix(tbits, tbit0, n, i + 1, ix);
    atomicAdd(n_fixed, 1ull);
    for (int a = 0; a < fx.naggs; a++) {
        const QuirkFixAgg &fa = fx.a[a];
        double sum = 0.0, t0 = 0.0, v0 = 0.0;
        bool have = false, ok = false, took_end = false;
        for (int64_t r = b; r >= 0 && r < n; r++) {
            if (!bit_at(tbits, tbit0, r)) continue;
            const int64_t t = ts[r];
            if (!open_end) {
                if (t > end) break;
                if (t == end) { if (took_end) break; took_end = true; }
            }
            if (!bit_at(fa.vbits, fa.vbit0, r)) continue;
            const uint64_t raw = reinterpret_cast<const uint64_t *>(fa.values)[r];
            const double v1 = fa.type == BOWGPU_INT64 ? (double)(int64_t)raw : __longlong_as_double((long long)raw), t1 = (double)t;
            if (have) { sum += (v0 + v1) / 2 * (t1 - t0); ok = true; }     // integral.go:24
            t0 = t1; v0 = v1; have = true;
        }
        uint64_t bits = 0;
        if (ok) {
            if (fa.kind == BOWGPU_AGG_WAVG_LINEAR) sum = sum / (double)interval;    // weightedmean.go:29-33: LastValue - FirstValue
            bits = apply_factors((uint64_t)__double_as_longlong(sum), false, fa.n_factors, fa.factors);
        }
        fa.out_values[slot] = bits;
        if (ok) atomicOr(&fa.out_valid[slot >> 5], 1u << (slot & 31));
        else atomicAnd(&fa.out_valid[slot >> 5], ~(1u << (slot & 31)));
    }
}

// NumRows of a call over an interval column with nulls is counted as Count over the keep bits (Int64): float64(count) in place
// (aggregation_test.go:28-31 returns float64(w.Bow.NumRows()))
__global__ __launch_bounds__(256) void count_to_f64_kernel(uint64_t *__restrict__ v, const int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint64_t)__double_as_longlong((double)(int64_t)v[i]);
}

// out (bit 0 = row 0, whole 64-bit words) = a AND b; a / b: Arrow bitmaps at any bit offset, nullptr = all ones
__global__ __launch_bounds__(256) void and_bits_kernel(const uint32_t *__restrict__ a, const int64_t abit0, const uint32_t *__restrict__ b,
                                                       const int64_t bbit0, const int64_t n, uint64_t *__restrict__ out) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nw = (n + 63) >> 6;
    if (w >= nw) return;
    auto word64 = [&](const uint32_t *bits, int64_t bit0) -> uint64_t {
        if (!bits) return ~0ull;
        const int64_t bit = bit0 + 64 * w, wi = bit >> 5, last = (bit0 + n - 1) >> 5;
        const int sh = (int)(bit & 31);
        const uint64_t d0 = bits[wi], d1 = wi + 1 <= last ? bits[wi + 1] : 0u, d2 = (sh && wi + 2 <= last) ? bits[wi + 2] : 0u;
        const uint64_t lo = d0 | (d1 << 32);
        return sh ? (lo >> sh) | (d2 << (64 - sh)) : lo;
    };
    uint64_t x = word64(a, abit0) & word64(b, bbit0);
    const int64_t left = n - 64 * w;
    if (left < 64) x &= (1ull << left) - 1ull;
    out[w] = x;
}

// ---- Rolling.Interpolate over an interval column with nulls (exclusive iteration).  Its output is the slices themselves
// (interpolation.go:98-161): rows in no slice vanish, null-timestamp rows inside a slice are copied as they are, and the
// interpolators look for neighbours among the rows where timestamp AND value are valid (linear.go:20-31, stepprevious.go:19).  So
// the call is made on the KEPT rows - compacted, timestamps forward-filled, every column's validity ANDed with the interval column's
// - plus one more Int64 column under interpolation.None whose only valid rows are the null-timestamp ones, holding their row
// number: wherever that column comes out valid, the output row is a copy of such a row, and interp_patch_kernel gives it its
// null timestamp and the values' own validity back.
__global__ __launch_bounds__(256) void keep_counts_kernel(const uint64_t *__restrict__ keep, const int64_t nw, int32_t *__restrict__ counts) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) counts[w] = __popcll(keep[w]);
}

constexpr uint32_t kFlagQuirk = 1u << 31;
// flags[dst]: bit 0 = the row's timestamp is valid, bit 1 + c = column c has a value in it (the interval column: its timestamp), bit 31: kFlagQuirk
__global__ __launch_bounds__(256) void compact_rows_kernel(const uint64_t *__restrict__ keep, const int64_t *__restrict__ base, const int64_t n,
                                                           const int64_t *__restrict__ ts_eff, const uint32_t *__restrict__ tbits, const int64_t tbit0,
                                                           const uint64_t *__restrict__ plain, const CompactCols cc, int64_t *__restrict__ marker,
                                                           uint32_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t kw = keep[i >> 6];
    const int lane = (int)(i & 63);
    if (!((kw >> lane) & 1ull)) return;
    const int64_t dst = base[i >> 6] + __popcll(kw & ((1ull << lane) - 1ull));
    const bool tv = bit_at(tbits, tbit0, i);
    uint32_t f = tv ? 1u : 0u;
    for (int c = 0; c < cc.ncols; c++) {
        if (c == cc.ts_col) { cc.out_values[c][dst] = (uint64_t)ts_eff[i]; if (tv) f |= 2u << c; }
        else { cc.out_values[c][dst] = cc.values[c][i]; if (bit_at(cc.vbits[c], cc.vbit0[c], i)) f |= 2u << c; }
    }
    if (plain && tv && !((plain[i >> 6] >> lane) & 1ull)) f |= kFlagQuirk;     // (inclusive iteration: ts_nulls.hip's "row on a window start ...")
    flags[dst] = f;
    marker[dst] = dst;
}

// the bitmaps of the compacted call, 64 rows per wavefront: lookup[c] = value AND timestamp valid; marker_bits = timestamp null
__global__ __launch_bounds__(256) void pack_flags_kernel(const uint32_t *__restrict__ flags, const int64_t m, const CompactCols cc, uint64_t *__restrict__ marker_bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = i < m ? flags[i] : 1u;
    const bool lane0 = (threadIdx.x & 63) == 0 && i < m;
    for (int c = 0; c < cc.ncols; c++) {
        const unsigned long long b = __ballot(i < m && (f & 1u) && ((f >> (1 + c)) & 1u));
        if (lane0 && cc.lookup_bits[c]) cc.lookup_bits[c][i >> 6] = b;
        if (lane0 && c == cc.ts_col && cc.ts_bits) cc.ts_bits[i >> 6] = b;     // (the patch pass looks neighbours of the interval column up too)
    }
    const unsigned long long mb = __ballot(!(f & 1u) || (f & kFlagQuirk));
    if (lane0) marker_bits[i >> 6] = mb;
}

__global__ __launch_bounds__(256) void interp_patch_kernel(const int64_t *__restrict__ marker_out, const uint32_t *__restrict__ marker_valid, const int64_t m_out,
                                                           const uint32_t *__restrict__ flags, const CompactCols cc, const PatchInterps px) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_out || !((marker_valid[j >> 5] >> (j & 31)) & 1u)) return;
    const int64_t r = marker_out[j];
    const uint32_t f = flags[r];
    if (f & 1u) {
        // a row with a timestamp under the marker: a row on a window start with a null timestamp behind it (inclusive iteration).  Its
        // FIRST copy closes the window it ends; the copy right behind it stands where the next window - which begins at the last of
        // those null rows, without this row - gets its synthetic start row (interpolation.go:137-160): the interpolators' values at the
        // window's start, neighbours looked up among the rows with timestamp AND value from this row down / from the row behind it up
        if (!(f & kFlagQuirk) || j == 0 || !((marker_valid[(j - 1) >> 5] >> ((j - 1) & 31)) & 1u) || marker_out[j - 1] != r) return;
        const int64_t sk = (int64_t)cc.out_values[cc.ts_col][r];
        for (int c = 0; c < cc.ncols; c++) {
            InterpCol ic;
            ic.type = px.type[c]; ic.kind = px.kind[c]; ic.const_value = px.const_value[c];
            ic.has_prev = px.has_prev[c]; ic.prev_t_valid = px.prev_t_valid[c]; ic.prev_v_valid = px.prev_v_valid[c];
            ic.prev_t = px.prev_t[c]; ic.prev_v = px.prev_v[c]; ic.prev_v_i64 = px.prev_v_i64[c];
            ic.next_valid = 0; ic.next_t = 0; ic.next_v = 0;
            const uint32_t *lb = reinterpret_cast<const uint32_t *>(px.both_bits[c]);
            const int64_t pi = prev_valid_ix(lb, 0, px.m, r, px.nbr[c]), ni = next_valid_ix(lb, 0, px.m, r + 1, px.nbr[c]);
            NbPoint pp, np;
            pp.has = pi >= 0; pp.t = pi >= 0 ? (int64_t)cc.out_values[cc.ts_col][pi] : 0; pp.bits = pi >= 0 ? cc.out_values[c][pi] : 0;
            np.has = ni >= 0; np.t = ni >= 0 ? (int64_t)cc.out_values[cc.ts_col][ni] : 0; np.bits = ni >= 0 ? cc.out_values[c][ni] : 0;
            uint64_t bits; int valid;
            synth_value_pt(ic, sk, pp, np, &bits, &valid);
            cc.patch_values[c][j] = bits;
            if (valid) atomicOr(&cc.patch_valid[c][j >> 5], 1u << (j & 31));
            else atomicAnd(&cc.patch_valid[c][j >> 5], ~(1u << (j & 31)));
        }
        return;
    }
    for (int c = 0; c < cc.ncols; c++) {
        if ((f >> (1 + c)) & 1u) {
            cc.patch_values[c][j] = cc.out_values[c][r];
            atomicOr(&cc.patch_valid[c][j >> 5], 1u << (j & 31));
        } else atomicAnd(&cc.patch_valid[c][j >> 5], ~(1u << (j & 31)));
    }
}

}  // namespace

int launch_ts_nullfill(Ctx *c, const int64_t *ts, const uint32_t *tbits, int64_t tbit0, int64_t n, const NbrIndex &ix, int64_t s0, int64_t interval,
                       const MagicDiv &magic, int inclusive, int64_t *ts_eff, uint64_t *keep, uint64_t *plain, unsigned long long *d_dropped) {
    if (n <= 0) return 0;
    BG_HIP(hipMemsetAsync(d_dropped, 0, 8, c->stream));
    hipLaunchKernelGGL(ts_nullfill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ts, tbits, tbit0, n, ix, s0, interval, magic, inclusive,
                       ts_eff, keep, plain, d_dropped);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_ts_quirk_fix(Ctx *c, const int64_t *ts, const uint32_t *tbits, int64_t tbit0, int64_t n, const NbrIndex &ix, int64_t s0, int64_t interval,
                        const MagicDiv &magic, int64_t W, const QuirkFix &fx, unsigned long long *d_fixed) {
    if (n <= 0 || fx.naggs == 0) return 0;
    hipLaunchKernelGGL(ts_quirk_fix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ts, tbits, tbit0, n, ix, s0, interval, magic, W, fx,
                       d_fixed);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_count_to_f64(Ctx *c, uint64_t *v, int64_t n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(count_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, v, n);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_and_bits(Ctx *c, const uint32_t *a, int64_t abit0, const uint32_t *b, int64_t bbit0, int64_t n, uint64_t *out) {
    if (n <= 0) return 0;
    const int64_t nw = (n + 63) >> 6;
    hipLaunchKernelGGL(and_bits_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->stream, a, abit0, b, bbit0, n, out);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_keep_counts(Ctx *c, const uint64_t *keep, int64_t nw, int32_t *counts) {
    if (nw <= 0) return 0;
    hipLaunchKernelGGL(keep_counts_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->stream, keep, nw, counts);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_compact_rows(Ctx *c, const uint64_t *keep, const int64_t *base, int64_t n, const int64_t *ts_eff, const uint32_t *tbits, int64_t tbit0,
                        const uint64_t *plain, const CompactCols &cc, int64_t *marker, uint32_t *flags) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, keep, base, n, ts_eff, tbits, tbit0, plain, cc, marker, flags);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_pack_flags(Ctx *c, const uint32_t *flags, int64_t m, const CompactCols &cc, uint64_t *marker_bits) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(pack_flags_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, flags, m, cc, marker_bits);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_interp_patch(Ctx *c, const int64_t *marker_out, const uint32_t *marker_valid, int64_t m_out, const uint32_t *flags, const CompactCols &cc,
                        const PatchInterps &px) {
    if (m_out <= 0) return 0;
    hipLaunchKernelGGL(interp_patch_kernel, dim3((unsigned)((m_out + 255) / 256)), dim3(256), 0, c->stream, marker_out, marker_valid, m_out, flags, cc, px);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu

