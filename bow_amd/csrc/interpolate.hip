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

constexpr int kIR = 2;              // consecutive rows per thread (one 16-B load per lane and column)
constexpr int kITile = 512;         // rows per workgroup (1 wavefront x 128 rows: 3.6 ms per call, 4 x 128: 2.75, 8 x 128: 2.75)
constexpr int kIThreads = 256;
constexpr int kCountThreads = 256;  // interp_count_kernel: four tiles (wavefronts) per workgroup
constexpr int kIStage = 1024;       // outputs of one column staged in LDS per tile (rows + synthetic rows)
constexpr int kISpanWords = 128;    // output validity bits staged in LDS per column: 4096 bits (rows + synthetic rows of a tile)
constexpr int kLongRuns = 32;       // long runs of empty windows a tile shares among its threads (more: their owners write them)
constexpr int kSmallRun = 4;        // synthetic rows a lane writes itself; longer runs of empty windows go to the whole workgroup

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
        double t2, v2;
        if (ni >= 0) { t2 = (double)ts[ni]; v2 = bits_to_f64(ic.values[ni], ic.type); }
        else if (ic.next_valid) { t2 = ic.next_t; v2 = ic.next_v; }  // the nearest valid point lies on a shard to the right
        else break;
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

// ---- kIR consecutive rows per thread: ts (and each column) arrive as one 16-B load per lane
struct RowsR {
    uint64_t wid[kIR];
    int64_t synth[kIR];   // synthetic rows right in front of row k
    bool exact[kIR];
};

__device__ __forceinline__ void loadR(const uint64_t *__restrict__ src, int64_t i, int64_t n, bool vec, uint64_t (&v)[kIR]) {
    if (vec && i + 1 < n) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(src + i);
        v[0] = a.x; v[1] = a.y;
    } else {
        v[0] = i < n ? src[i] : 0;
        v[1] = i + 1 < n ? src[i + 1] : 0;
    }
}

// flags of rows i..i+kIR-1 from their timestamps and the timestamp left of row i (t_left; ignored for i == 0).
// kFast: every row lies in [s0, s0 + 2^31), |ts| < 2^53 and no window starts at -1 - the usual case, established by the host:
// window ids come from one 32-bit multiply-high per row and "exact head" is an integer comparison (the float64 round trip
// of interpolation.go:121-123 is the identity below 2^53).
struct Magic32 { uint32_t m, sh1, sh2; };
__device__ __forceinline__ uint32_t mdiv32(uint32_t x, const Magic32 &d) {
    const uint32_t t = __umulhi(d.m, x);
    return (t + ((x - t) >> d.sh1)) >> d.sh2;
}

// has_left: the shard has rows to its left (sharded Interpolate): row 0 then has a left neighbour too, t_left = their last ts
template <bool kFast>
__device__ __forceinline__ RowsR rows_flags(const uint64_t (&t)[kIR], int64_t t_left, int64_t i, int64_t n, int64_t s0, int64_t interval,
                                            const MagicDiv &magic, const Magic32 &m32, int64_t kq, bool has_left, bool *unsorted) {
    RowsR f;
    uint64_t wprev = 0;
    int64_t tprev = t_left;
    if (i > 0 || has_left) {
        if (kFast) wprev = mdiv32((uint32_t)((uint64_t)t_left - (uint64_t)s0), m32);
        else wprev = t_left < s0 ? 0 : magic_div((uint64_t)t_left - (uint64_t)s0, magic);
    }
#pragma unroll
    for (int k = 0; k < kIR; k++) {
        f.wid[k] = 0; f.synth[k] = 0; f.exact[k] = false;
        if (i + k >= n) continue;
        const int64_t tk = (int64_t)t[k];
        const bool first = i + k == 0 && !has_left;
        if (!first && tprev > tk) *unsorted = true;
        uint64_t w;
        bool head, exact;
        int64_t synth;
        if (kFast) {
            const uint32_t rel = (uint32_t)((uint64_t)tk - (uint64_t)s0);
            const uint32_t w32 = mdiv32(rel, m32);
            head = first || w32 != (uint32_t)wprev;
            exact = head && rel == w32 * (uint32_t)interval;
            const uint32_t before = (head && !first) ? w32 - (uint32_t)wprev - 1u : 0u;
            synth = head ? (int64_t)(before + (exact ? 0u : 1u)) : 0;
            w = w32;
        } else {
            w = tk < s0 ? 0 : magic_div((uint64_t)tk - (uint64_t)s0, magic);  // rows below s0 ride in window 0
            head = first || w != wprev;
            const int64_t before = (head && !first) ? (int64_t)(w - wprev) - 1 : 0;        // empty windows in front of this row's window
            // first valid ts of the window, through float64 as the reference does (interpolation.go:121-123)
            exact = head && go_f64_to_i64((double)tk) == s0 + (int64_t)(w * (uint64_t)interval);
            synth = head ? before + (exact ? 0 : 1) : 0;
            // kq >= 0: the window that starts at -1 has no row of its own; the reference then takes its "first value" -1
            // (interpolation.go:119) for a timestamp equal to the window start and adds NO synthetic row for it
            if (head && kq >= 0 && (uint64_t)kq <= w && (first || (uint64_t)kq > wprev) && !(exact && (uint64_t)kq == w)) synth -= 1;
        }
        f.wid[k] = w; f.synth[k] = synth; f.exact[k] = exact;
        wprev = w; tprev = tk;
    }
    return f;
}

// timestamp left of a thread's first row: the neighbouring lane's last row, LDS across waves, global across tiles
__device__ __forceinline__ int64_t left_ts(const uint64_t (&t)[kIR], const int64_t *ts, int64_t i, int64_t n, long long *wave_last, int tid,
                                           int64_t shard_left_ts) {
    const int lane = tid & 63, wv = tid >> 6;
    const long long mine = (long long)t[kIR - 1];
    long long l = __shfl_up(mine, 1);
    if (lane == 63) wave_last[wv] = mine;
    __syncthreads();
    if (lane == 0) l = wv > 0 ? wave_last[wv - 1] : ((i > 0 && i < n) ? ts[i - 1] : shard_left_ts);
    return (int64_t)l;
}

// exact heads per tile of kITile rows.  One wavefront per tile, no barrier: lane l holds rows 2l, 2l+1 of each of the tile's four
// 128-row chunks (16-B loads, all in flight at once); the timestamp left of a lane's rows comes from its neighbour lane, from
// the previous chunk's last lane, or (first chunk) from the row before the tile; the count is two ballots per chunk.
template <bool kFast>
__global__ __launch_bounds__(kCountThreads) void interp_count_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval,
                                                                 MagicDiv magic, Magic32 m32, int has_left, int64_t shard_left_ts,
                                                                 int32_t *tile_exact, uint32_t *status) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * (kCountThreads / 64) + wv;
    const int64_t base = tile * kITile;
    if (base >= n) return;
    const bool vec = (reinterpret_cast<uintptr_t>(ts) & 15) == 0;
    uint64_t t[kITile / 128][kIR];
#pragma unroll
    for (int k = 0; k < kITile / 128; k++) loadR(reinterpret_cast<const uint64_t *>(ts), base + 128 * k + kIR * lane, n, vec, t[k]);
    int64_t t_before = base > 0 ? ts[base - 1] : shard_left_ts;
    bool unsorted = false;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kITile / 128; k++) {
        const int64_t i = base + 128 * k + kIR * lane;
        long long tl = __shfl_up((long long)t[k][kIR - 1], 1);
        if (lane == 0) tl = (long long)t_before;
        const RowsR f = rows_flags<kFast>(t[k], (int64_t)tl, i, n, s0, interval, magic, m32, -1, has_left != 0, &unsorted);
        cnt += __popcll(__ballot(f.exact[0])) + __popcll(__ballot(f.exact[1]));
        t_before = (int64_t)lane_value(t[k][kIR - 1], 63);
    }
    if (__ballot(unsorted) && lane == 0) atomicOr(&status[0], 1u);
    if (lane == 0) tile_exact[tile] = cnt;
}

template <bool kFast>
__global__ __launch_bounds__(kIThreads) void interp_tile_kernel(const InterpParams p) {
    struct LongRun { long long a, o_row, synth; unsigned long long k0; };
    __shared__ uint32_t lbits[kMaxCols][kISpanWords];
    __shared__ long long wave_tot[kIThreads / 64];
    __shared__ long long wave_last[kIThreads / 64];
    __shared__ int s_nlong;
    __shared__ LongRun runs[kLongRuns];
    __shared__ uint64_t sval2[2][kIStage];  // double-buffered by column parity: a column is staged while the previous one drains
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kITile;
    const int64_t i = r0 + kIR * (int64_t)tid;
    for (int w = tid; w < kMaxCols * kISpanWords; w += kIThreads) (&lbits[0][0])[w] = 0;
    const int64_t kq = (!kFast && p.kq >= 0 && p.status[1]) ? p.kq : -1;  // see rows_flags
    const Magic32 m32 = {p.m32, p.sh1_32, p.sh2_32};
    if (tid == 0) s_nlong = 0;

    // ---- loads: ts, then the first column right behind it
    const bool vec_ts = (reinterpret_cast<uintptr_t>(p.ts) & 15) == 0;
    uint64_t t[kIR], v[kIR];
    loadR(reinterpret_cast<const uint64_t *>(p.ts), i, p.n, vec_ts, t);
    loadR(p.cols[0].values, i, p.n, (reinterpret_cast<uintptr_t>(p.cols[0].values) & 15) == 0, v);

    // first output position of this tile: one past the position of row r0 - 1
    int64_t o_base = 0;
    if (r0 > 0) {
        const int64_t tp = p.ts[r0 - 1];
        const uint64_t wp = kFast ? (uint64_t)mdiv32((uint32_t)((uint64_t)tp - (uint64_t)p.s0), m32)
                                  : (tp < p.s0 ? 0 : magic_div((uint64_t)tp - (uint64_t)p.s0, p.magic));
        // (a shard with rows to its left only accounts for the windows after their last one: wbase = that window + 1)
        o_base = (r0 - (p.drop < r0 ? p.drop : r0)) + (int64_t)wp + 1 - p.wbase - p.tile_exact_before[blockIdx.x];
        if (kq >= 0 && (uint64_t)kq <= wp) o_base -= 1;
    }
    const int64_t lbase = o_base & ~(int64_t)31;  // LDS bit 0

    const int64_t tl = left_ts(t, p.ts, i, p.n, wave_last, tid, p.left_ts);  // (one __syncthreads inside: the LDS clears above are visible after it)
    bool unsorted = false;
    const RowsR f = rows_flags<kFast>(t, tl, i, p.n, p.s0, p.interval, p.magic, m32, kq, p.has_left != 0, &unsorted);
    if (unsorted) atomicOr(&p.status[0], 1u);

    // ---- output positions: thread totals -> wave scan -> workgroup
    int emitted[kIR];
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < kIR; k++) {
        emitted[k] = (i + k < p.n && i + k >= p.drop) ? 1 : 0;  // (rows of a window 0 without own rows are dropped: interp_quirk_kernel)
        mine += emitted[k] + f.synth[k];
    }
    long long inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    long long woff = 0;
    for (int k = 0; k < wv; k++) woff += wave_tot[k];
    long long tile_total = 0;
    for (int k = 0; k < kIThreads / 64; k++) tile_total += wave_tot[k];
    int64_t o_row[kIR];  // output position of real row k; its synthetic rows end right before it
    {
        int64_t o = o_base + woff + inc - mine;
#pragma unroll
        for (int k = 0; k < kIR; k++) { o += f.synth[k]; o_row[k] = o; o += emitted[k]; }
    }

    auto set_bits = [&](int c, int64_t o, uint32_t mask) {  // mask: up to 4 consecutive bits from output position o
        const int64_t rel = o - lbase;
        if (rel + 4 <= (int64_t)kISpanWords * 32) {
            const int sh = (int)(rel & 31);
            atomicOr(&lbits[c][rel >> 5], mask << sh);
            if (sh > 28 && (mask >> (32 - sh))) atomicOr(&lbits[c][(rel >> 5) + 1], mask >> (32 - sh));
        } else {  // a tile with very many synthetic rows
            for (int b = 0; b < 4; b++)
                if ((mask >> b) & 1u) atomicOr(&p.cols[c].out_valid_words[(o + b) >> 5], 1u << ((o + b) & 31));
        }
    };
    // Outputs are staged in LDS in output order and leave as contiguous 8-B-per-lane stores: written straight from the rows'
    // own lanes they would be 8-B pieces ~36 B apart, which the memory system turns into 2.4x the write traffic plus
    // read-modify-write fills (measured).  A tile with more outputs than the stage holds (long runs of empty windows)
    // writes directly.
    const bool staged = tile_total <= kIStage;
    uint64_t *sval = sval2[0];
    auto put = [&](const InterpCol &ic, int64_t o, uint64_t bits) {
        if (staged) sval[o - o_base] = bits;  // (sval: the buffer of the column being staged)
        else ic.out_values[o] = bits;
    };
    // synthetic rows j = first, first + step, ... < count of column c in front of row a (j = 0 is the one next to the row; its
    // window is k0 = the row's own window when the row is not an exact head, else the window before; then the empty windows,
    // latest first)
    auto emit_synth = [&](int c, int64_t a, int64_t o_a, uint64_t k0, int64_t count, int64_t first, int64_t step) {
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
            put(ic, o_a - 1 - j, bits);
            if (valid) set_bits(c, o_a - 1 - j, 1u);
        }
    };

    // long runs of empty windows are shared by the whole workgroup; the short ones stay with their row's lane
    bool own_run[kIR];
#pragma unroll
    for (int k = 0; k < kIR; k++) {
        own_run[k] = f.synth[k] > 0;
        if (f.synth[k] > kSmallRun) {
            const int q = atomicAdd(&s_nlong, 1);
            if (q < kLongRuns) {
                runs[q].a = i + k; runs[q].o_row = o_row[k]; runs[q].synth = f.synth[k];
                runs[q].k0 = f.exact[k] ? f.wid[k] - 1 : f.wid[k];
                own_run[k] = false;
            }  // (list full: the owner writes the run itself)
        }
    }
    __syncthreads();
    const int nlong = s_nlong < kLongRuns ? s_nlong : kLongRuns;

    // ---- one column at a time (the next column's loads go out before this one's stores)
    const bool contiguous = f.synth[1] == 0 && emitted[0] && i + 1 < p.n;
    for (int c = 0; c < p.ncols; c++) {
        const InterpCol &ic = p.cols[c];
        sval = sval2[c & 1];
        uint64_t cur[kIR] = {v[0], v[1]};
        if (c + 1 < p.ncols) loadR(p.cols[c + 1].values, i, p.n, (reinterpret_cast<uintptr_t>(p.cols[c + 1].values) & 15) == 0, v);
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < kIR; k++)
            if (emitted[k]) {
                put(ic, o_row[k], cur[k]);
                if (bit_at(ic.vbits, ic.vbit0, i + k)) m |= 1u << k;
            }
        if (contiguous) { if (m) set_bits(c, o_row[0], m); }
        else
            for (int k = 0; k < kIR; k++) if ((m >> k) & 1u) set_bits(c, o_row[k], 1u);
#pragma unroll
        for (int k = 0; k < kIR; k++)
            if (own_run[k]) emit_synth(c, i + k, o_row[k], f.exact[k] ? f.wid[k] - 1 : f.wid[k], f.synth[k], 0, 1);
        for (int q = 0; q < nlong; q++) emit_synth(c, runs[q].a, runs[q].o_row, runs[q].k0, runs[q].synth, tid, kIThreads);
        if (staged) {
            __syncthreads();  // staged -> flush; the next column stages into the other buffer, so no barrier after the flush
            for (int64_t rel = tid; rel < tile_total; rel += kIThreads) ic.out_values[o_base + rel] = sval[rel];
        }
    }
    __syncthreads();

    // ---- flush the staged validity bits: whole words; the first and last word may be shared with the neighbouring tiles
    const int64_t span_bits = (o_base - lbase) + tile_total;
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

static Magic32 magic32_make(int64_t interval) {  // Granlund & Montgomery fig. 4.1 with N = 32 (interval < 2^31)
    const uint64_t d = (uint64_t)interval;
    int l = 0;
    while (l < 32 && (1ull << l) < d) l++;
    Magic32 m;
    m.m = (uint32_t)((((1ull << l) - d) << 32) / d) + 1;
    m.sh1 = l < 1 ? (uint32_t)l : 1u;
    m.sh2 = l > 1 ? (uint32_t)(l - 1) : 0u;
    return m;
}

// the fast 32-bit form applies when every row lies in [s0, s0 + 2^31), timestamps are exact in float64 and no window starts at -1
bool interp_fast32(const Plan &plan, int64_t kq) {
    const int64_t lim53 = 1ll << 53;
    return kq < 0 && plan.first_ts >= plan.s0 && plan.interval < (1ll << 31) && (uint64_t)plan.last_ts - (uint64_t)plan.s0 < (1ull << 31) &&
           plan.first_ts > -lim53 && plan.last_ts < lim53;
}

int launch_interp_count(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t kq, int has_left, int64_t left_ts,
                        int32_t *tile_exact, uint32_t *status) {
    const int64_t ntiles = (n + kITile - 1) / kITile;
    if (ntiles <= 0) return 0;
    if (kq >= 0 || plan.first_ts < plan.s0)
        hipLaunchKernelGGL(interp_quirk_kernel, dim3(1), dim3(64), 0, c->stream, ts, n, plan.s0, plan.interval, kq, status);
    if (interp_fast32(plan, kq))
        hipLaunchKernelGGL(interp_count_kernel<true>, dim3((unsigned)((ntiles + 3) / 4)), dim3(kCountThreads), 0, c->stream, ts, n, plan.s0, plan.interval,
                           plan.magic, magic32_make(plan.interval), has_left, left_ts, tile_exact, status);
    else
        hipLaunchKernelGGL(interp_count_kernel<false>, dim3((unsigned)((ntiles + 3) / 4)), dim3(kCountThreads), 0, c->stream, ts, n, plan.s0, plan.interval,
                           plan.magic, Magic32{0, 0, 0}, has_left, left_ts, tile_exact, status);
    BG_HIP(hipGetLastError());
    return 0;
}

int64_t interp_tiles(int64_t n) { return (n + kITile - 1) / kITile; }
void interp_magic32(int64_t interval, uint32_t *m, uint32_t *sh1, uint32_t *sh2) {
    const Magic32 x = magic32_make(interval);
    *m = x.m; *sh1 = x.sh1; *sh2 = x.sh2;
}

int launch_interp_tiles(Ctx *c, const InterpParams &p) {
    const int64_t ntiles = (p.n + kITile - 1) / kITile;
    if (ntiles <= 0) return 0;
    if (p.fast32) hipLaunchKernelGGL(interp_tile_kernel<true>, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, p);
    else hipLaunchKernelGGL(interp_tile_kernel<false>, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, p);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
