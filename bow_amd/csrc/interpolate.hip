// interpolate.hip — Rolling.Interpolate (reference rolling/interpolation.go:30-161 with the built-in interpolators of
// rolling/interpolation/{windowstart,linear,stepprevious,none}.go) as ONE row-tiled streaming pass.
//
// The reference walks the windows; window k gets one synthetic row {interpolated value per column} in front of its rows
// when its first valid timestamp is not s_k (interpolation.go:118-137) - empty windows included.  For an ascending,
// non-null interval column that is a statement about ROWS: call row j an "exact head" when it is the first row of its
// window and ts[j] == s_wid(j).  Windows 0..wid(i) each have either an exact head (among rows <= i) or a synthetic row, so
//
//     output position of row i  =  i + (wid(i) + 1) - E(i),        E(i) = number of exact heads among rows 0..i
//
// and the synthetic rows of the windows between row i-1 and row i sit right in front of row i.  Two launches:
//   interp_count_kernel   exact heads per tile of 1024 rows (reads ts once)  -> M = W - sum, and the scan gives E per tile
//   interp_tile_kernel    per tile: flags, workgroup scan, copy the rows of every column to their positions, synthesise
//                         the start rows, assemble the output validity bits in LDS (flushed as whole words)
// Algorithmic traffic: 8 B (ts, twice) + 8 B per column read, 8 B per column written, per row.
#include "bitmap_device.h"

namespace bowgpu {

namespace {

constexpr int kITile = 1024;        // rows per workgroup
constexpr int kIThreads = 256;
constexpr int kIRounds = kITile / kIThreads;
constexpr int kISpanWords = 128;    // output validity bits staged in LDS per column: 4096 bits (rows + synthetic rows of a tile)
constexpr int kSmallRun = 4;        // synthetic rows a lane writes itself; longer runs of empty windows go to the whole workgroup

struct RowFlags {
    uint64_t wid;
    int64_t synth;   // synthetic rows right in front of this row
    bool exact;
};

// wid / head / exact-head / synthetic rows in front of row i (i > 0 reads ts[i-1] too)
// kq >= 0: the window whose start is -1 has no row of its own; the reference then takes its "first value" -1
// (interpolation.go:119) for a timestamp equal to the window start and adds NO synthetic row for it.
__device__ __forceinline__ RowFlags row_flags(const int64_t *ts, int64_t i, int64_t s0, int64_t interval, const MagicDiv &magic,
                                              int64_t kq, bool *unsorted) {
    RowFlags f;
    const int64_t t = ts[i];
    f.wid = t < s0 ? 0 : magic_div((uint64_t)t - (uint64_t)s0, magic);   // rows below s0 ride in window 0
    bool head = true;
    uint64_t wprev = 0;
    int64_t before = 0;  // windows that end before this row's window and have no row: wid - wprev - 1
    if (i > 0) {
        const int64_t tp = ts[i - 1];
        if (tp > t) *unsorted = true;
        wprev = tp < s0 ? 0 : magic_div((uint64_t)tp - (uint64_t)s0, magic);
        head = f.wid != wprev;
        before = head ? (int64_t)(f.wid - wprev) - 1 : 0;
    }
    // first valid ts of the window, through float64 as the reference does (interpolation.go:121-123)
    f.exact = head && go_f64_to_i64((double)t) == s0 + (int64_t)(f.wid * (uint64_t)interval);
    f.synth = head ? before + (f.exact ? 0 : 1) : 0;
    if (head && kq >= 0 && (uint64_t)kq <= f.wid && (i == 0 || (uint64_t)kq > wprev) && !(f.exact && (uint64_t)kq == f.wid)) f.synth -= 1;
    return f;
}

// the value an interpolator gives the synthetic row of a window starting at sk whose FirstIndex is row a
// (interpolation/windowstart.go:10-12, linear.go:12-37, stepprevious.go:11-24, none.go); pi / ni = previous valid row
// before a / next valid row from a (looked up once per run of synthetic rows)
__device__ __forceinline__ void synth_value(const InterpCol &ic, const int64_t *ts, int64_t sk, int64_t pi, int64_t ni,
                                            uint64_t *bits_out, int *valid_out) {
    const bool is_int = ic.type == BOWGPU_INT64;
    uint64_t bits = 0;
    int valid = 0;
    switch (ic.kind) {
    case BOWGPU_INTERP_WINDOW_START:
        bits = is_int ? (uint64_t)sk : (uint64_t)__double_as_longlong((double)sk);
        valid = 1;
        break;
    case BOWGPU_INTERP_CONST:
        bits = is_int ? (uint64_t)go_f64_to_i64(ic.const_value) : (uint64_t)__double_as_longlong(ic.const_value);
        valid = 1;
        break;
    case BOWGPU_INTERP_LINEAR: {  // (ts has no nulls: both-valid == value valid)
        double t0, v0;
        if (pi >= 0) { t0 = (double)ts[pi]; v0 = bits_to_f64(ic.values[pi], ic.type); }
        else if (ic.has_prev && ic.prev_t_valid && ic.prev_v_valid) { t0 = ic.prev_t; v0 = ic.prev_v; }
        else break;
        if (ni < 0) break;
        const double t2 = (double)ts[ni], v2 = bits_to_f64(ic.values[ni], ic.type);
        const double coef = ((double)sk - t0) / (t2 - t0);
        const double r = ((v2 - v0) * coef) + v0;
        bits = is_int ? (uint64_t)go_f64_to_i64(r) : (uint64_t)__double_as_longlong(r);  // SetOrDrop: bowconvert.go:28-29
        valid = 1;
        break;
    }
    case BOWGPU_INTERP_STEP_PREVIOUS:
        if (pi >= 0) { bits = ic.values[pi]; valid = 1; }
        else if (ic.has_prev && ic.prev_v_valid) {
            bits = is_int ? (uint64_t)ic.prev_v_i64 : (uint64_t)__double_as_longlong(ic.prev_v);
            valid = 1;
        }
        break;
    default: break;  // None: nil
    }
    *bits_out = valid ? bits : 0;
    *valid_out = valid;
}

}  // namespace

__global__ __launch_bounds__(kIThreads) void interp_count_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval,
                                                                 MagicDiv magic, int32_t *tile_exact, uint32_t *status) {
    __shared__ int part[kIThreads / 64];
    const int64_t r0 = (int64_t)blockIdx.x * kITile;
    int cnt = 0;
    bool unsorted = false;
    for (int k = 0; k < kIRounds; k++) {
        const int64_t i = r0 + k * kIThreads + threadIdx.x;
        if (i < n) cnt += row_flags(ts, i, s0, interval, magic, -1, &unsorted).exact ? 1 : 0;
    }
    if (unsorted) atomicOr(&status[0], 1u);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_exact[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(kIThreads) void interp_tile_kernel(const InterpParams p) {
    struct LongRun { long long a, o_row, synth; unsigned long long k0; };
    __shared__ uint32_t lbits[kMaxCols][kISpanWords];
    __shared__ long long wave_tot[kIThreads / 64];
    __shared__ long long s_running;
    __shared__ int s_nlong;
    __shared__ LongRun runs[kIThreads];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kITile;
    for (int w = tid; w < kMaxCols * kISpanWords; w += kIThreads) (&lbits[0][0])[w] = 0;
    const int64_t kq = (p.kq >= 0 && p.status[1]) ? p.kq : -1;  // see row_flags

    // first output position of this tile: one past the position of row r0 - 1
    int64_t o_base = 0;
    if (r0 > 0) {
        const int64_t tp = p.ts[r0 - 1];
        const uint64_t wp = tp < p.s0 ? 0 : magic_div((uint64_t)tp - (uint64_t)p.s0, p.magic);
        o_base = (r0 - (p.drop < r0 ? p.drop : r0)) + (int64_t)wp + 1 - p.tile_exact_before[blockIdx.x];
        if (kq >= 0 && (uint64_t)kq <= wp) o_base -= 1;
    }
    const int64_t lbase = o_base & ~(int64_t)31;  // LDS bit 0
    if (tid == 0) s_running = 0;
    __syncthreads();

    auto set_bit = [&](int c, int64_t o) {
        const int64_t rel = o - lbase;
        if (rel < (int64_t)kISpanWords * 32) atomicOr(&lbits[c][rel >> 5], 1u << (rel & 31));
        else atomicOr(&p.cols[c].out_valid_words[o >> 5], 1u << (o & 31));  // a tile with very many synthetic rows
    };
    // synthetic rows j = first, first + step, ... < count in front of row a (j = 0 is the one next to the row; its window is
    // k0 = the row's own window when the row is not an exact head, else the window before; then the empty windows, latest first)
    auto emit_synth = [&](int64_t a, int64_t o_row, uint64_t k0, int64_t count, int64_t first, int64_t step) {
        for (int c = 0; c < p.ncols; c++) {
            const InterpCol &ic = p.cols[c];
            int64_t pi = -1, ni = -1;  // the same two neighbours for the whole run: FirstIndex of all these windows is row a
            if (ic.kind == BOWGPU_INTERP_LINEAR || ic.kind == BOWGPU_INTERP_STEP_PREVIOUS) pi = prev_valid_ix(ic.vbits, ic.vbit0, p.n, a - 1, ic.nbr);
            if (ic.kind == BOWGPU_INTERP_LINEAR) ni = next_valid_ix(ic.vbits, ic.vbit0, p.n, a, ic.nbr);
            for (int64_t j = first; j < count; j += step) {
                uint64_t k = k0 - (uint64_t)j;
                if (kq >= 0 && (uint64_t)kq <= k0 && (uint64_t)kq >= k) k -= 1;  // the run skips window kq
                const int64_t sk = p.s0 + (int64_t)(k * (uint64_t)p.interval);
                uint64_t bits;
                int valid;
                synth_value(ic, p.ts, sk, pi, ni, &bits, &valid);
                ic.out_values[o_row - 1 - j] = bits;
                if (valid) set_bit(c, o_row - 1 - j);
            }
        }
    };

    for (int round = 0; round < kIRounds; round++) {
        const int64_t i = r0 + round * kIThreads + tid;
        RowFlags f;
        f.wid = 0; f.synth = 0; f.exact = false;
        bool unsorted = false;
        const bool live = i < p.n;
        if (live) f = row_flags(p.ts, i, p.s0, p.interval, p.magic, kq, &unsorted);
        if (unsorted) atomicOr(&p.status[0], 1u);
        // workgroup inclusive scan of (1 + synthetic rows in front of the row)
        const int emitted = (live && i >= p.drop) ? 1 : 0;  // (rows of a window 0 without own rows are dropped: interp_quirk_kernel)
        long long inc = live ? emitted + f.synth : 0;
        for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(inc, o); if (lane >= o) inc += y; }
        if (lane == 63) wave_tot[wv] = inc;
        if (tid == 0) s_nlong = 0;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < wv; k++) woff += wave_tot[k];
        const long long round_total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const int64_t o_row = o_base + s_running + woff + inc - emitted;  // output position of the real row; synthetic rows end right before it
        if (live) {
            if (emitted)
                for (int c = 0; c < p.ncols; c++) {
                    const InterpCol &ic = p.cols[c];
                    ic.out_values[o_row] = ic.values[i];
                    if (bit_at(ic.vbits, ic.vbit0, i)) set_bit(c, o_row);
                }
            if (f.synth > 0) {
                const uint64_t k0 = f.exact ? f.wid - 1 : f.wid;
                if (f.synth <= kSmallRun) {
                    emit_synth(i, o_row, k0, f.synth, 0, 1);
                } else {  // a long run of empty windows: shared by the whole workgroup below
                    const int q = atomicAdd(&s_nlong, 1);
                    runs[q].a = i; runs[q].o_row = o_row; runs[q].synth = f.synth; runs[q].k0 = k0;
                }
            }
        }
        __syncthreads();
        const int nlong = s_nlong;
        for (int q = 0; q < nlong; q++) emit_synth(runs[q].a, runs[q].o_row, runs[q].k0, runs[q].synth, tid, kIThreads);
        __syncthreads();
        if (tid == 0) s_running += round_total;
        __syncthreads();
    }

    // flush the staged validity bits: whole words; the first and last word may be shared with the neighbouring tiles
    const int64_t span_bits = (o_base - lbase) + s_running;
    int64_t nwords = (span_bits + 31) >> 5;
    if (nwords > kISpanWords) nwords = kISpanWords;
    for (int c = 0; c < p.ncols; c++) {
        uint32_t *dst = p.cols[c].out_valid_words + (lbase >> 5);
        for (int64_t w = tid; w < nwords; w += kIThreads) {
            const uint32_t x = lbits[c][w];
            if (w == 0 || w == nwords - 1) { if (x) atomicOr(&dst[w], x); }
            else dst[w] = x;
        }
    }
}

// The two corner cases of the reference's window walk that are not statements about single rows:
//   status[1] = 1 when window kq (the one that starts at -1, if any) has no row of its own;
//   status[2..3] = number of leading rows below s0 when window 0 has no row of its own: Go's truncating division can put
//   s0 above a negative ts[0]; such rows ride in window 0's slice, and when that slice is empty (rolling.go:
//   "lastRowIndex stays -1") they belong to no window at all, so Interpolate's concatenation of window bows drops them.
__global__ void interp_quirk_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval, int64_t kq, uint32_t *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto lower_bound = [&](int64_t v) {
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (ts[mid] >= v) hi = mid; else lo = mid + 1; }
        return lo;
    };
    if (kq >= 0) {
        const int64_t a = lower_bound(-1);
        status[1] = (a == n || ts[a] >= -1 + interval) ? 1u : 0u;
    }
    if (ts[0] < s0) {
        const int64_t a = lower_bound(s0);
        const bool dead = a == n || ts[a] >= s0 + interval;
        const uint64_t drop = dead ? (uint64_t)a : 0;
        status[2] = (uint32_t)drop; status[3] = (uint32_t)(drop >> 32);
    }
}

int launch_interp_count(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t kq, int32_t *tile_exact, uint32_t *status) {
    const int64_t ntiles = (n + kITile - 1) / kITile;
    if (ntiles <= 0) return 0;
    if (kq >= 0 || plan.first_ts < plan.s0)
        hipLaunchKernelGGL(interp_quirk_kernel, dim3(1), dim3(64), 0, c->stream, ts, n, plan.s0, plan.interval, kq, status);
    hipLaunchKernelGGL(interp_count_kernel, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, ts, n, plan.s0, plan.interval, plan.magic,
                       tile_exact, status);
    BG_HIP(hipGetLastError());
    return 0;
}

int64_t interp_tiles(int64_t n) { return (n + kITile - 1) / kITile; }

int launch_interp_tiles(Ctx *c, const InterpParams &p) {
    const int64_t ntiles = (p.n + kITile - 1) / kITile;
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(interp_tile_kernel, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, p);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
