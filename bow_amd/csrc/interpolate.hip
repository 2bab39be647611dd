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
#include <stdlib.h>

#include <type_traits>

#include "bitmap_device.h"
#include "interp_device.h"

namespace bowgpu {

namespace {

constexpr int kIR = 2;              // consecutive rows per thread (one 16-B load per lane and column)
constexpr int kITile = 512;         // rows per workgroup (1 wavefront x 128 rows: 3.6 ms per call, 4 x 128: 2.75, 8 x 128: 2.75)
constexpr int kIThreads = 256;
constexpr int kCountThreads = 1024; // interp_count_kernel: sixteen tiles (wavefronts) per workgroup = one super-tile of kInterpSuperRows rows
constexpr int kIStage = 1024;       // outputs of one column staged in LDS per tile (rows + synthetic rows)
constexpr int kISpanWords = 128;    // output validity bits staged in LDS per column: 4096 bits (rows + synthetic rows of a tile)
constexpr int kLongRuns = 32;       // long runs of empty windows a tile shares among its threads (more: their owners write them)
constexpr int kSmallRun = 4;        // synthetic rows a lane writes itself; longer runs of empty windows go to the whole workgroup

// the same with the neighbours given as row numbers (-1: none)
__device__ __forceinline__ void synth_value(const InterpCol &ic, const int64_t *ts, int64_t sk, int64_t pi, int64_t ni,
                                            uint64_t *bits_out, int *valid_out) {
    NbPoint pp, np;
    pp.has = pi >= 0 ? 1 : 0; pp.t = pi >= 0 ? ts[pi] : 0; pp.bits = pi >= 0 ? ic.values[pi] : 0;
    np.has = ni >= 0 ? 1 : 0; np.t = ni >= 0 ? ts[ni] : 0; np.bits = ni >= 0 ? ic.values[ni] : 0;
    synth_value_pt(ic, sk, pp, np, bits_out, valid_out);
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
// exact heads in front of 256-row entry e (pass 1's two-level prefix)
__device__ __forceinline__ int64_t interp_exact_before(const InterpParams &p, int64_t e) {
    return p.super_before[e / (kInterpSuperRows / 256)] + (int64_t)p.tile_local[e];
}

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

// exact heads per 256 rows, as an exclusive prefix WITHIN the super-tile of kSuperRows rows that holds them (tile_local), and per
// super-tile (super_sum; interp_super_scan_kernel turns those into the prefix over the super-tiles - two launches for pass 1
// where round 4 had four, and 1.5 MB of prefix written where the three-kernel scan wrote and re-read 3 + 1.5).  A workgroup per
// super-tile, one wavefront per tile of kITile rows: lane l holds rows 2l, 2l+1 of each of the tile's four 128-row chunks (16-B
// loads, all in flight at once); the timestamp left of a lane's rows comes from its neighbour lane, from the previous chunk's last
// lane, or (first chunk) from the row before the tile; the count is two ballots per chunk; ONE barrier, behind the loads.
template <bool kFast>
__global__ __launch_bounds__(kCountThreads) void interp_count_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval,
                                                                 MagicDiv magic, Magic32 m32, int has_left, int64_t shard_left_ts,
                                                                 int32_t *tile_local, int32_t *super_sum, uint32_t *status) {
    __shared__ int32_t sh[2 * (kCountThreads / 64)];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * (kCountThreads / 64) + wv;
    const int64_t base = tile * kITile;
    int cnt[2] = {0, 0};   // per 256 rows
    if (base < n) {
        const bool vec = (reinterpret_cast<uintptr_t>(ts) & 15) == 0;
        uint64_t t[kITile / 128][kIR];
#pragma unroll
        for (int k = 0; k < kITile / 128; k++) loadR(reinterpret_cast<const uint64_t *>(ts), base + 128 * k + kIR * lane, n, vec, t[k]);
        int64_t t_before = base > 0 ? ts[base - 1] : shard_left_ts;
        bool unsorted = false;
#pragma unroll
        for (int k = 0; k < kITile / 128; k++) {
            const int64_t i = base + 128 * k + kIR * lane;
            long long tl = __shfl_up((long long)t[k][kIR - 1], 1);
            if (lane == 0) tl = (long long)t_before;
            const RowsR f = rows_flags<kFast>(t[k], (int64_t)tl, i, n, s0, interval, magic, m32, -1, has_left != 0, &unsorted);
            cnt[k >> 1] += __popcll(__ballot(f.exact[0])) + __popcll(__ballot(f.exact[1]));
            t_before = (int64_t)lane_value(t[k][kIR - 1], 63);
        }
        if (__ballot(unsorted) && lane == 0) atomicOr(&status[0], 1u);
    }
    if (lane == 0) { sh[2 * wv] = cnt[0]; sh[2 * wv + 1] = cnt[1]; }
    __syncthreads();
    if (wv == 0) {
        constexpr int kE = 2 * (kCountThreads / 64);   // entries of a super-tile (<= 64)
        const int32_t mine = lane < kE ? sh[lane] : 0;
        int32_t incl = mine;
#pragma unroll
        for (int o = 1; o < kE; o <<= 1) {
            const int32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int64_t e = (int64_t)blockIdx.x * kE + lane;
        if (lane < kE && e * 256 < n) tile_local[e] = incl - mine;
        if (lane == kE - 1) super_sum[blockIdx.x] = incl;
    }
}

// exclusive prefix of the super-tile sums (one workgroup: n / 8192 sums - 12 k for 1e8 rows), the total behind them.  In passes of
// 16384 sums: a thread takes 16 consecutive ones (four 16-byte loads, all in flight at once), the workgroup scans the threads'
// totals, a running carry joins the passes (1e8 rows: one pass, 9 us; 1e9 rows: eight).
__global__ __launch_bounds__(1024) void interp_super_scan_kernel(const int32_t *super_sum, int64_t nsuper, int64_t *super_before, int64_t *total,
                                                                 const uint32_t *status, int64_t *host_back) {
    constexpr int kQ = 4;
    __shared__ long long sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    long long carry = 0;
    for (int64_t pass0 = 0; pass0 < nsuper; pass0 += 1024 * 4 * kQ) {
        const int64_t a = pass0 + (int64_t)tid * 4 * kQ;
        int32_t x[4 * kQ];
#pragma unroll
        for (int q = 0; q < kQ; q++) {
            const int64_t i = a + 4 * q;
            int4 v = make_int4(0, 0, 0, 0);
            if (i + 4 <= nsuper) v = *reinterpret_cast<const int4 *>(super_sum + i);   // (the pool block is 256-byte aligned)
            else { if (i < nsuper) v.x = super_sum[i]; if (i + 1 < nsuper) v.y = super_sum[i + 1]; if (i + 2 < nsuper) v.z = super_sum[i + 2]; }
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
        long long mine = 0;
#pragma unroll
        for (int j = 0; j < 4 * kQ; j++) mine += x[j];
        long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        __syncthreads();   // (the previous pass is done with sh)
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        long long before = carry, all = 0;
        for (int w = 0; w < 16; w++) { if (w < wv) before += sh[w]; all += sh[w]; }
        long long run = before + incl - mine;
#pragma unroll
        for (int j = 0; j < 4 * kQ; j++) { if (a + j < nsuper) super_before[a + j] = run; run += x[j]; }
        carry += all;
    }
    if (tid == 1023) { super_before[nsuper] = carry; *total = carry; }
    // pass 1's findings straight into the host's registered block (the stores go over the link; no copy command behind the launch):
    // [0] the total, [1] status words 0 | 1 << 32, [2] status words 2 | 3 << 32 - written by the count kernel in front of this one
    if (host_back) {
        if (tid == 1023) host_back[0] = carry;
        if (tid < 2) host_back[1 + tid] = (int64_t)((uint64_t)status[2 * tid] | ((uint64_t)status[2 * tid + 1] << 32));
    }
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
    const int64_t kq = (!kFast && p.kq >= 0 && p.kq_empty) ? p.kq : -1;  // see rows_flags
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
        o_base = (r0 - (p.drop < r0 ? p.drop : r0)) + (int64_t)wp + 1 - p.wbase - interp_exact_before(p, 2 * (int64_t)blockIdx.x);   // (one entry per 256 rows)
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

// (rounds 1 - 2 had two more kernels here, interp_wave_kernel and interp_wave2_kernel: interp_wave3_kernel below replaced both, 1.6 - 2.9x
// faster; they are gone from the library since round 4.  interp_tile_kernel above stays as the kernel for the shapes wave3 does not
// take - 64-bit window ids, dropped rows, the -1 sentinel window, a trip whose lists overflow - and as the second opinion of the tests.)
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive wave scan of one unsigned per lane: Hillis-Steele inside the 16-lane rows (row_shr 1, 2, 4, 8), then the row totals
// across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3): six DPP moves + adds, no LDS
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

// In-kernel stamps (diagnostic build -DBOWGPU_STAMPS only; cdna_hip_programming.md section 7): cycles per phase, summed over the
// wavefronts into status[32 + 2 * phase] (64-bit), read back with bowgpu_debug_status.  Never in the product build.
#ifdef BOWGPU_STAMPS
#define W2_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                         __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// interp_wave3_kernel (round 3): interp_wave2_kernel's pass rebuilt for OCCUPANCY.  wave2 holds 13 KB of LDS and 158 VGPRs per
// wavefront - 11 resident per CU - and its rate followed that (counters: 9.5 wavefronts per CU, 68 % of a wave's life spent waiting,
// LDS bank-conflict cycles 2.5x the LDS issue cycles: 64 lanes OR-ing their validity bits into two or three words).  Same trips
// (512 rows, one wavefront, no barrier, all of a trip's loads up front), but everything that does not have to be in LDS is not:
//   * the column's validity words of the trip come through the constant address space (load_bits128: scalar loads), as do the
//     128 rows in front of the trip, among which the last valid point before it nearly always lies (two scalar loads fetch it) -
//     no per-column bitmap and "previous rows" arrays, no LDS-DMA round;
//   * timestamps stay in REGISTERS as 32-bit offsets from s0; a run of synthetic rows gets its neighbours' timestamps with
//     ds_bpermute (the LDS crossbar, no LDS memory) and their values from the stage through a row -> position table;
//   * outputs are staged in LDS in output order, ALIGNED to the output bitmap's words (stage slot = position + (o_trip & 31)),
//     with one validity BYTE per staged output; the flush reads 16 bytes per lane for the values and, in a second lane mapping
//     (lane l = slot 64 r + l), the flags whose ballots ARE the bitmap words - no LDS atomics, no bit interleaving;
//   * 16-byte non-temporal stores onto 16-byte aligned addresses, whole validity words (atomic OR only for the two words a trip
//     may share with its neighbours).
// LDS 9.5 KB per wavefront (10 KB allocated: 16 per CU), <= 128 VGPRs.  Two designs that lost on the way (same data, 1e8 rows):
// 256-row trips at 20 wavefronts per CU, 1.74 ms against wave2's 1.26 - twice the per-trip instructions per row, the scalar unit
// saturated (5.1e8 scalar instructions per launch against 1.5e8); neighbour points gathered from global memory instead of
// LDS / registers, 1.95 ms - gfx950 counts loads and stores in ONE counter, so waiting for a gather waited for every store of the
// previous column's flush.  Trips whose outputs exceed the stage (long runs of empty windows) write directly; more runs than the
// list holds (windows of < 4 rows on average) raise status[5] and the host redoes the call with interp_tile_kernel.  Results are
// bit-identical to that kernel's (the tests run both).
constexpr int kT3Rows = 512;                  // rows per trip: 4 chunks of 128, lane l = rows 2l, 2l + 1 of each
constexpr int kT3Ch = kT3Rows / 128;
constexpr int kT3Stage = kT3Rows + kT3Rows / 2;   // outputs staged per trip and column (the rows + up to half as many synthetic rows)
constexpr int kT3Slots = kT3Stage + 32;       // ... + the bit offset of the trip's first output inside its bitmap word

template <int kRuns>
struct Wave3Lds {
    alignas(16) uint64_t val[kT3Slots];       // staged outputs
    alignas(8) uint8_t fl[kT3Slots + 8];      // their validity, one byte each
    uint32_t run_a[kRuns];                    // local row | copy flag << 9 | count << 10 (count saturates: such a trip is redone elsewhere)
    uint32_t run_k[kRuns];                    // window of the run's first synthetic row
    uint16_t run_o[kRuns];                    // output position (relative to the trip's first) of the row the run sits in front of
    uint16_t pos[kT3Rows];                    // row -> output position: a run's neighbour rows are read back from the stage
    uint64_t vw[2 * kT3Ch];                   // the current column's validity bits of the trip's rows: row r = bit r & 63 of word r >> 6
};

// One trip.  kFull: all of its 512 rows exist - every trip but the frame's last - and every input column is 16-byte aligned (the
// usual call: Arrow buffers are; a slice at an odd row offset is not); kFull and (below) `staged` are compile-time constants
// of the column loop so that what LLVM hoists out of that loop is what the USUAL trip needs: with the partial trip's masks and the
// unstaged form's addresses in the same loop it hoisted - and spilled, one v_writelane each - 233 scalars per wavefront, 114 without.
template <bool kIncl, bool kFull, int kRuns>
__device__ __forceinline__ void wave3_trip(const InterpParams &p, const int64_t ntrips, const int64_t trip, Wave3Lds<kRuns> &L) {
    const int lane = threadIdx.x;
#ifdef BOWGPU_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    const int64_t base = trip * kT3Rows;
    const int64_t left_trip = p.n - base;
    const bool full = kFull || left_trip >= kT3Rows;   // (kFull = false also takes the full trips of a call with a column off 16-byte alignment)
    const int nloc = full ? kT3Rows : (int)left_trip;
    const Magic32 m32 = {p.m32, p.sh1_32, p.sh2_32};
    const uint32_t i32 = (uint32_t)p.interval;
    const uint64_t *tsu = reinterpret_cast<const uint64_t *>(p.ts);
    typedef const uint64_t __attribute__((address_space(4))) *const_u64;   // inputs nobody writes during the kernel: scalar loads

    auto load2 = [&](const uint64_t *col, int k, uint64_t *a, uint64_t *bb) {
        const int r = 128 * k + 2 * lane;
        const uint64_t *src = col + base;
        if (full && (kFull || (reinterpret_cast<uintptr_t>(col) & 15) == 0)) {   // (kFull: the host found every column 16-byte aligned)
            typedef unsigned long long u64x2i_t __attribute__((ext_vector_type(2)));
            const u64x2i_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2i_t *>(src + r));   // (streamed once)
            *a = v.x; *bb = v.y;
        } else { *a = r < left_trip ? src[r] : 0; *bb = r + 1 < left_trip ? src[r + 1] : 0; }
    };

    // ---- round 1: the trip's timestamps and the values of its first column that is not the interval column itself
    uint64_t ta[kT3Ch], tb[kT3Ch], na[kT3Ch], nb[kT3Ch];
#pragma unroll
    for (int k = 0; k < kT3Ch; k++) load2(tsu, k, &ta[k], &tb[k]);
    const bool col0_is_ts = p.cols[0].values == tsu;
    const int first_loaded = col0_is_ts ? 1 : 0;     // the column na / nb hold when the column loop starts
    if (first_loaded < p.ncols) {
#pragma unroll
        for (int k = 0; k < kT3Ch; k++) load2(p.cols[first_loaded].values, k, &na[k], &nb[k]);
    }
    // Window ids and timestamps are 32-bit and RELATIVE TO THE TRIP: w0 = the window of its first row (one exact 64-bit division on
    // the scalar unit), ws0 = that window's start.  Only the trip's own span has to fit 32 bits - millisecond timestamps over months,
    // microseconds over hours per 512 rows - not the frame's (round 2's wave kernels needed every row within 2^31 of s0 and sent
    // everything else to the workgroup kernel at half the rate).  A trip that does not fit raises status[5] (redo elsewhere).
    const int64_t ts_first = (int64_t)((const_u64)(uintptr_t)tsu)[base], ts_lastrow = (int64_t)((const_u64)(uintptr_t)tsu)[base + nloc - 1];
    const uint64_t w0 = magic_div((uint64_t)ts_first - (uint64_t)p.s0, p.magic);      // (every row lies at or above s0: the host checked the first)
    const int64_t ws0 = p.s0 + (int64_t)(w0 * (uint64_t)p.interval);
    const uint32_t s0lo = (uint32_t)ws0;
    bool toolong = ts_lastrow < ts_first || (uint64_t)ts_lastrow - (uint64_t)ws0 >= 0x7FFFFFFFull;
    int64_t o_trip = 0, t_before = p.left_ts;
    uint64_t gap0 = 0;   // windows between the row before the trip and the trip's first row (0: the same window)
    if (base > 0 || p.has_left) {
        if (base > 0) t_before = (int64_t)((const_u64)(uintptr_t)tsu)[base - 1];
        const uint64_t wp = magic_div((uint64_t)t_before - (uint64_t)p.s0, p.magic);
        gap0 = w0 - wp;
        toolong |= t_before > ts_first || gap0 >= 0x3FFFFFull;
        // inclusive windows: one extra row in front of every window's first row (synthetic or the copy), but none for an exact row 0
        // (tile_exact_before holds one entry per 256 rows)
        if (base > 0) o_trip = kIncl ? base + (int64_t)wp + 1 - p.wbase - p.e0 : base + (int64_t)wp + 1 - p.wbase - interp_exact_before(p, trip * (kT3Rows / 256));
    }
    if (lane < p.ncols) p.edge_words[(int64_t)lane * ntrips + trip] = 0ull;   // (no entry unless a staged flush below leaves one)

    // ---- phase 1: output positions (relative to o_trip) of the lane's rows, the run list; the timestamps stay as 32-bit offsets
    uint32_t rr0[kT3Ch], rr1[kT3Ch];
    uint32_t tot = 0;
    int nrun = 0;
    bool unsorted = false;   // (the count pass checks the order too; an inclusive fill on its own - no count pass - has only this check)
    {
        uint32_t wb_prev = 0;
        uint64_t tb_prev = (uint64_t)t_before;
#pragma unroll
        for (int k = 0; k < kT3Ch; k++) {
            const int64_t i = base + 128 * k + 2 * lane;
            const uint32_t ra = (uint32_t)ta[k] - s0lo, rb = (uint32_t)tb[k] - s0lo;
            rr0[k] = ra; rr1[k] = rb;
            const bool in0 = i < p.n, in1 = i + 1 < p.n;
            const bool first = i == 0 && !p.has_left;  // the frame's first row has no left neighbour
            if (kIncl) {
                // (exclusive windows: every fill follows a count pass - its own or the _count call's, whose reuse is guarded by the
                // row-count check below - and interp_count_kernel checks the order; 14 vector instructions per chunk not spent twice)
                const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)tb[k], 0x138, 0xf, 0xf, false);
                const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(tb[k] >> 32), 0x138, 0xf, 0xf, false);
                const int64_t tl = lane == 0 ? (int64_t)tb_prev : (int64_t)(((uint64_t)phi << 32) | plo);
                unsorted |= (in0 && !first && tl > (int64_t)ta[k]) || (in1 && (int64_t)ta[k] > (int64_t)tb[k]);
                tb_prev = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[k] >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[k], 63);
            }
            const uint32_t wa = mdiv32(ra, m32), wb = mdiv32(rb, m32);
            // the left neighbour's window: the lane below computed it as its wb (a DPP move, not a third division); lane 0 takes the
            // previous chunk's last - or, the trip's first row: its window is local id 0 by construction, the row before it lies gap0
            // windows further back (ids are unsigned and wrap, the differences below come out right)
            uint32_t wl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            if (lane == 0) wl = k == 0 ? 0u - (uint32_t)gap0 : (uint32_t)__builtin_amdgcn_readlane((int)wb_prev, 63);
            wb_prev = wb;
            const bool head0 = in0 && (first || wa != wl), head1 = in1 && wb != wa;
            const bool exact0 = head0 && ra == wa * i32, exact1 = head1 && rb == wb * i32;
            // rows in front of a head: the empty windows before it + a synthetic row for its own window - or, when the row sits on
            // its window's start, nothing (exclusive windows) / the copy of itself that closes the window before (inclusive ones)
            const uint32_t sy0 = head0 ? ((first ? 0u : wa - wl - 1u) + (exact0 ? ((kIncl && !first) ? 1u : 0u) : 1u)) : 0u;
            const uint32_t sy1 = head1 ? (wb - wa - 1u + (exact1 ? (kIncl ? 1u : 0u) : 1u)) : 0u;
            const uint32_t e0 = in0 ? 1u : 0u, e1 = in1 ? 1u : 0u;
            const uint32_t mine = e0 + e1 + sy0 + sy1;
            const uint32_t inc = wave_scan_u32(mine);
            uint32_t o = tot + inc - mine;
            o += sy0; const uint32_t o0 = o; o += e0;
            o += sy1; const uint32_t o1 = o;
            tot += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            *reinterpret_cast<uint32_t *>(&L.pos[128 * k + 2 * lane]) = (o0 < 0xFFFFu ? o0 : 0xFFFFu) | ((o1 < 0xFFFFu ? o1 : 0xFFFFu) << 16);
            // the runs of this chunk, in row order
            const bool ha = sy0 > 0, hb = sy1 > 0;
            const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
            toolong |= sy0 >= 0x3FFFFFu || sy1 >= 0x3FFFFFu;
            if (ma | mb) {
                int pos = nrun;
                pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
                pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
                const uint32_t la = (uint32_t)(128 * k + 2 * lane);
                // (the run's first row is the copy of its row, not a synthetic row, when the windows are inclusive and the row sits on its window's start)
                if (ha && pos < kRuns) {
                    L.run_a[pos] = la | ((kIncl && exact0) ? 0x200u : 0u) | (sy0 << 10);
                    L.run_k[pos] = exact0 ? wa - 1u : wa;
                    L.run_o[pos] = (uint16_t)(o0 < 0xFFFFu ? o0 : 0xFFFFu);
                }
                pos += ha ? 1 : 0;
                if (hb && pos < kRuns) {
                    L.run_a[pos] = (la + 1u) | ((kIncl && exact1) ? 0x200u : 0u) | (sy1 << 10);
                    L.run_k[pos] = exact1 ? wb - 1u : wb;
                    L.run_o[pos] = (uint16_t)(o1 < 0xFFFFu ? o1 : 0xFFFFu);
                }
                nrun += __popcll(ma) + __popcll(mb);
            }
        }
    }
    const uint32_t sh_o = (uint32_t)(o_trip & 31);   // stage slot of the trip's first output = its bit inside its bitmap word
    if (__ballot(unsorted)) {   // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0 && !__hip_atomic_load(&p.status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[0], 1u);
        return;
    }
    // outside this kernel's list, or positions its 16-bit tables cannot hold (a trip with 65 535 or more synthetic rows): the host
    // redoes the call with another kernel
    if (nrun > kRuns || tot >= 0xFFFFu || __ballot(toolong)) {
        if (lane == 0) atomicOr(&p.status[5], 1u);
        return;
    }
    const bool staged_rt = tot <= (uint32_t)kT3Stage;
    // The last trip ends where the count pass said the outputs end, and every trip lies inside the outputs - or the interval
    // column is not the one that was counted (a _fill that reuses its _count's prefix: include/bowgpu.h, the contract between the
    // two calls).  Such a trip stores NOTHING: a stale prefix must not become a write outside the caller's buffers.
    if (o_trip < 0 || o_trip + (int64_t)tot > p.n_out || (trip == ntrips - 1 && o_trip + (int64_t)tot != p.n_out)) {
        if (lane == 0 && !__hip_atomic_load(&p.status[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[6], 1u);
        return;
    }
    // the timestamp offset of row r of the trip (r differs per lane): out of the registers of the lane that holds it
    auto rel_of = [&](int r) -> uint32_t {
        const int src = ((r & 127) >> 1) << 2;
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < kT3Ch; k++) {
            const uint32_t y0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rr0[k]);
            const uint32_t y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rr1[k]);
            if ((r >> 7) == k) x = (r & 1) ? y1 : y0;
        }
        return x;
    };
    W2_STAMP(0);   // round 1 (timestamps) + phase 1
    bool far = false;   // a neighbour point further away than the index-free walk looks: the host repeats the call with the index (status[7])

    // ---- phase 2: one column at a time (the next column's values in flight meanwhile)
    // (the loop twice, with `staged` a compile-time constant: the staged form then holds no global store inside a loop - LLVM
    // drains the memory counter in front of a loop that stores and uses registers loaded outside it, which on gfx950, where
    // stores and loads share the counter, is a wait for every store of the previous column's flush)
    const int lane0 = lane;
    auto column_loop = [&](auto staged_tag) {
    constexpr bool staged = decltype(staged_tag)::value;
#pragma unroll 1
    for (int c = 0; c < p.ncols; c++) {
        // (the lane number, opaque per column: LLVM otherwise computes every lane predicate of the body - lane == w, lane < 8, ... , a
        // v_cmp each - in front of the loop and, out of scalar registers, parks each in two lanes of a vector register: two
        // v_writelane there, two v_readlane per use here, for what one v_cmp recomputes)
        int lane_v = lane0;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        const InterpCol &ic = p.cols[c];
        const bool want_p = ic.kind == BOWGPU_INTERP_LINEAR || ic.kind == BOWGPU_INTERP_STEP_PREVIOUS, want_n = ic.kind == BOWGPU_INTERP_LINEAR;
        uint64_t a[kT3Ch], bq[kT3Ch];
        if (c == 0 && col0_is_ts) {   // the interval column's own values: s0 + offset (the loaded registers were given up after phase 1)
#pragma unroll
            for (int k = 0; k < kT3Ch; k++) { a[k] = (uint64_t)ws0 + (uint64_t)rr0[k]; bq[k] = (uint64_t)ws0 + (uint64_t)rr1[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < kT3Ch; k++) { a[k] = na[k]; bq[k] = nb[k]; }
            if (c + 1 < p.ncols) {
#pragma unroll
                for (int k = 0; k < kT3Ch; k++) load2(p.cols[c + 1].values, k, &na[k], &nb[k]);
            }
        }
        // the column's validity bits of the trip's rows: scalar 64-bit words
        uint64_t vs[2 * kT3Ch];
#pragma unroll
        for (int k = 0; k < kT3Ch; k++) {
            vs[2 * k] = 0; vs[2 * k + 1] = 0;
            if (full) load_bits128<true>(ic.vbits, ic.vbit0, base + 128 * k, p.n, &vs[2 * k], &vs[2 * k + 1]);
            else if (left_trip > 128 * k) load_bits128<false>(ic.vbits, ic.vbit0, base + 128 * k, p.n, &vs[2 * k], &vs[2 * k + 1]);
        }
        // the last valid point before the trip: among the 64 rows in front of it; further back only after a run of 64 nulls -
        // then through the bitmap and the neighbour index
        NbPoint carry; carry.has = 0; carry.t = 0; carry.bits = 0;
        if (want_p && base > 0 && nrun > 0) {
            uint64_t back = ~0ull, unused;
            if (ic.vbits) load_bits128<true>(ic.vbits, ic.vbit0, base - 128, p.n, &unused, &back);   // (base is a multiple of 512: the 128 rows exist)
            int64_t pi = -1;
            if (back) pi = base - 1 - __clzll((long long)back);
            else if (base > 64) pi = ic.nbr.prev_before ? prev_valid_ix(ic.vbits, ic.vbit0, p.n, base - 65, ic.nbr) : prev_valid_near(ic.vbits, ic.vbit0, p.n, base - 65, &far);
            if (pi >= 0) {
                carry.has = 1;
                carry.t = (int64_t)((const_u64)(uintptr_t)tsu)[pi];
                carry.bits = ((const_u64)(uintptr_t)ic.values)[pi];
            }
        }
        uint64_t *out = ic.out_values + o_trip;
        uint32_t nvalid = 0;   // valid outputs of this trip and column: staged - wave-uniform, off the flush's ballots; else per lane
        if (staged) {
#pragma unroll
            for (int i = 0; i < ((kT3Slots + 8) / 8 + 63) / 64; i++)
                if (lane + 64 * i < (kT3Slots + 8) / 8) *reinterpret_cast<uint64_t *>(&L.fl[8 * (lane + 64 * i)]) = 0ull;
        } else if (p.in_place) {
            // nobody zeroed the caller's bitmap: the words this trip owns (all it touches but a first one shared with the trip before,
            // which that trip stores) start as 0 - agent-scope stores, like the ORs that follow them
            const int64_t wl = (o_trip + (int64_t)tot - 1) >> 5;
            for (int64_t w = ((o_trip + 31) >> 5) + lane; w <= wl; w += 64) __hip_atomic_store(&ic.out_valid_words[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (lane < 2 * kT3Ch) {
            uint64_t x = 0;
#pragma unroll
            for (int w = 0; w < 2 * kT3Ch; w++) if (lane == w) x = vs[w];
            L.vw[lane] = x;
        }
        wave_lds_order();
        auto put = [&](uint32_t pos, uint64_t bits, int valid) {
            if (staged) {
                L.val[sh_o + pos] = bits;
                L.fl[sh_o + pos] = (uint8_t)valid;
            } else {
                out[pos] = bits;
                if (valid) {
                    nvalid++;
                    // (bits of the word shared with the trip before go to this trip's edge entry, like the staged form's: that trip may
                    // store the word plainly)
                    if (sh_o + pos < 32u && sh_o != 0u)
                        atomicOr(reinterpret_cast<unsigned long long *>(&p.edge_words[(int64_t)c * ntrips + trip]),
                                 (unsigned long long)(uint32_t)(o_trip >> 5) | ((unsigned long long)(1u << (sh_o + pos)) << 32));
                    else atomicOr(&ic.out_valid_words[(o_trip + pos) >> 5], 1u << ((o_trip + pos) & 31));
                }
            }
        };
        // the rows to their places
#pragma unroll
        for (int k = 0; k < kT3Ch; k++) {
            const int r = 128 * k + 2 * lane;
            const uint32_t pp = *reinterpret_cast<const uint32_t *>(&L.pos[r]);
            const int fl = (int)(((lane < 32 ? vs[2 * k] : vs[2 * k + 1]) >> ((2 * lane) & 63)) & 3ull);
            if (r < nloc) put(pp & 0xFFFFu, a[k], fl & 1);
            if (r + 1 < nloc) put(pp >> 16, bq[k], (fl >> 1) & 1);
        }
        wave_lds_order();
        W2_STAMP(1);   // column head: values arrive, validity words, carry, rows staged
        // the nearest valid rows around row al inside the trip (-1: none), from the column's validity words
        auto nearest = [&](int al, int *rp, int *rn) {
            *rp = -1; *rn = -1;
            if (want_p) {
                int r = al - 1;
                while (r >= 0) {
                    const int sh = r & 63;
                    uint64_t x = L.vw[r >> 6];
                    x = sh == 63 ? x : (x & ((2ull << sh) - 1ull));
                    if (x) { r = (r & ~63) + 63 - __clzll((long long)x); break; }
                    r = (r & ~63) - 1;
                }
                *rp = r;
            }
            if (want_n) {
                int r = al;
                bool found = false;
                while (r < nloc) {
                    const uint64_t x = L.vw[r >> 6] & (~0ull << (r & 63));
                    if (x) { r = (r & ~63) + __ffsll((long long)x) - 1; found = r < nloc; break; }
                    r = (r | 63) + 1;
                }
                *rn = found ? r : -1;
            }
        };
        // their (timestamp, value): the timestamps out of the registers of the lanes that hold them, the values from the stage
        // through the row -> position table (no global load in the run pass: on gfx950 a wait for a load is also a wait for every
        // store issued before it, i.e. for the previous column's flush); the point after the trip (rare) through the bitmap + index
        auto row_bits = [&](int r) -> uint64_t { return staged ? L.val[sh_o + (uint32_t)L.pos[r]] : ic.values[base + r]; };
        auto points = [&](int rp, int rn, NbPoint *qp, NbPoint *qn) {
            *qp = carry; qn->has = 0; qn->t = 0; qn->bits = 0;
            const uint32_t tp = want_p ? rel_of(rp < 0 ? 0 : rp) : 0u, tn = want_n ? rel_of(rn < 0 ? 0 : rn) : 0u;   // (bpermute: every lane takes part)
            if (rp >= 0) { qp->has = 1; qp->t = ws0 + (int64_t)(uint64_t)tp; qp->bits = row_bits(rp); }
            if (rn >= 0) { qn->has = 1; qn->t = ws0 + (int64_t)(uint64_t)tn; qn->bits = row_bits(rn); }
            else if (want_n && base + nloc < p.n) {
                const int64_t ni = ic.nbr.next_after ? next_valid_ix(ic.vbits, ic.vbit0, p.n, base + nloc, ic.nbr) : next_valid_near(ic.vbits, ic.vbit0, p.n, base + nloc, &far);
                if (ni >= 0) {
                    uint64_t xb = ic.values[ni], xt = tsu[ni];
                    // (the two loads are waited for HERE, inside the rare branch: left pending at the join, the compiler puts a wait for
                    // every outstanding load AND store in front of each later write to their registers - the flush's LDS reads among them)
                    asm volatile("" : "+v"(xb), "+v"(xt));
                    qn->has = 1; qn->bits = xb; qn->t = (int64_t)xt;
                }
            }
        };
        // ---- one lane per run (runs of up to kSmallRun rows; the long runs of empty windows follow, one at a time)
        bool any_long = false;
#ifdef BOWGPU_X_SKIP_RUNS   // (diagnostic build only, like BOWGPU_STAMPS: the kernel WITHOUT its run pass - wrong outputs, same bytes moved; the
        const int nrun_x = 0;   //  difference to the product build is what the pass costs: scratch/build_variant.sh xruns interpolate.hip -DBOWGPU_X_SKIP_RUNS)
#else
        const int nrun_x = nrun;
#endif
#pragma unroll 1
        for (int q0 = 0; q0 < nrun_x; q0 += 64) {
            const int q = q0 + lane;
            const bool act = q < nrun;
            const uint32_t e = act ? L.run_a[q] : 0u;
            const uint32_t cnt = e >> 10;
            const bool mine = act && cnt <= (uint32_t)kSmallRun;
            any_long |= __ballot(act && !mine) != 0ull;
            const uint32_t orow = act ? (uint32_t)L.run_o[q] : 0u;
            const uint32_t kfirst = act ? L.run_k[q] : 0u;
            const int al = (int)(e & 511u);
            const uint32_t jd = (kIncl && (e & 0x200u) && cnt > 0) ? 1u : 0u;   // the run's first row is the copy of row al
            int rp = -1, rn = -1;
            if (mine) nearest(al, &rp, &rn);
            NbPoint qp, qn;
            points(rp, rn, &qp, &qn);
            if (mine) {
                for (uint32_t j = 0; j < cnt; j++) {
                    uint64_t bits; int valid;
                    if (kIncl && j < jd) { bits = row_bits(al); valid = (int)((L.vw[al >> 6] >> (al & 63)) & 1ull); }
                    else {
                        // (local ids: -1 is the window before the trip's first; the offset of a window start inside the trip fits 32 bits
                        // like every row's - one 32-bit multiplication)
                        const int32_t soff = (int32_t)((kfirst - (j - jd)) * i32);
                        const int64_t sk = ws0 + (int64_t)soff;
                        synth_value_pt(ic, sk, qp, qn, &bits, &valid);
                    }
                    put(orow - 1 - j, bits, valid);
                }
            }
        }
        if (any_long) {   // long runs of empty windows: the whole wavefront on one run at a time, everything about the run wave-uniform
#pragma unroll 1
            for (int q = 0; q < nrun; q++) {
                const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.run_a[q]);
                const uint32_t cnt = e >> 10;
                if (cnt <= (uint32_t)kSmallRun) continue;
                const uint32_t orow = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)L.run_o[q]);
                const uint32_t kfirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.run_k[q]);
                const int al = (int)(e & 511u);
                const uint32_t jd = (kIncl && (e & 0x200u)) ? 1u : 0u;
                int rp, rn;
                nearest(al, &rp, &rn);
                NbPoint qp, qn;
                points(rp, rn, &qp, &qn);
                for (uint32_t j = (uint32_t)lane; j < cnt; j += 64) {
                    uint64_t bits; int valid;
                    if (kIncl && j < jd) { bits = row_bits(al); valid = (int)((L.vw[al >> 6] >> (al & 63)) & 1ull); }
                    else {
                        const int32_t soff = (int32_t)((kfirst - (j - jd)) * i32);
                        const int64_t sk = ws0 + (int64_t)soff;
                        synth_value_pt(ic, sk, qp, qn, &bits, &valid);
                    }
                    put(orow - 1 - j, bits, valid);
                }
            }
        }
        W2_STAMP(2);   // run pass
        // ---- the stage leaves: stage slot e <-> output position (o_trip - sh_o) + e, a multiple of 32 at e = 0, so every pair
        // (2i, 2i + 1) is 16-byte aligned in the output and every 64 slots are two whole words of the output bitmap
        if (staged) {
            wave_lds_order();
            const uint32_t e_lo = sh_o, e_hi = sh_o + tot;   // real slots
            uint64_t *outA = out - sh_o;
            uint32_t *wdst = ic.out_valid_words + ((o_trip - (int64_t)sh_o) >> 5);
            constexpr int kFlush = (kT3Slots + 127) / 128;
            // (every LDS read unconditional, at a clamped address: a conditional read would first zero its destination registers with
            // vector moves, and the compiler puts a wait for ALL outstanding stores in front of a vector write to registers an earlier
            // store took its data from - measured: the flush then waited for the previous batch's stores, 55 % of a wave's life)
#pragma unroll 1
            for (int k2 = 0; k2 < kFlush; k2 += 2) {   // two rounds of 128 slots at a time: the LDS reads of both, then the stores
                if (128u * (uint32_t)k2 >= e_hi) break;   // (wave-uniform)
                ulonglong2 fx[2];
                uint32_t ff[4];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t e0 = 128u * (uint32_t)(k2 + u) + 2u * (uint32_t)lane;
                    fx[u] = *reinterpret_cast<const ulonglong2 *>(&L.val[e0 < (uint32_t)(kT3Slots - 2) ? e0 : (uint32_t)(kT3Slots - 2)]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {   // the flags in bitmap order: lane l = slot 64 (2 k2 + u) + l
                    const uint32_t e = 64u * (uint32_t)(2 * k2 + u) + (uint32_t)lane;
                    ff[u] = (uint32_t)L.fl[e < (uint32_t)(kT3Slots + 7) ? e : (uint32_t)(kT3Slots + 7)];   // (slots outside [e_lo, e_hi) hold 0: zeroed per column)
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t e0 = 128u * (uint32_t)(k2 + u) + 2u * (uint32_t)lane;
                    const bool lo_ok = e0 - e_lo < tot, hi_ok = e0 + 1u - e_lo < tot;   // (unsigned: also false below e_lo)
                    const ulonglong2 x = fx[u];
                    if (lo_ok && hi_ok) { typedef unsigned long long u64x2o_t __attribute__((ext_vector_type(2))); u64x2o_t v; v.x = x.x; v.y = x.y; __builtin_nontemporal_store(v, reinterpret_cast<u64x2o_t *>(outA + e0)); }   // (never read back)
                    else if (lo_ok) outA[e0] = x.x;
                    else if (hi_ok) outA[e0 + 1] = x.y;
                }
                {
                // eight bitmap words: ballot u = slots 64 (2 k2 + u) .. + 63
                const uint64_t m0 = __ballot(ff[0] != 0u), m1 = __ballot(ff[1] != 0u), m2 = __ballot(ff[2] != 0u), m3 = __ballot(ff[3] != 0u);
                nvalid += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));   // (slots outside the trip's hold 0)
                if (lane < 8) {
                    const uint64_t mm = (lane >> 1) == 0 ? m0 : (lane >> 1) == 1 ? m1 : (lane >> 1) == 2 ? m2 : m3;
                    const uint32_t x32 = (lane & 1) ? (uint32_t)(mm >> 32) : (uint32_t)mm;
                    const uint32_t wi = 8u * (uint32_t)(k2 >> 1) + (uint32_t)lane;      // word of the stage
                    const uint32_t first_bit = 32u * wi;
                    if (first_bit < e_hi && first_bit + 32u > e_lo) {
                        // A trip's first word may also hold the last bits of the trip before it.  NO atomic: an agent-scope atomic OR is
                        // executed beyond the XCD's L2, and the two per trip and column this kernel (like interp_wave2_kernel) used to
                        // issue cost 0.7 of its 1.5 ms (same data, the stores left plain: 0.87 ms for the call against 1.57).  Instead the
                        // trip BEFORE stores that word plainly - its own bits, zeroes above them - and this trip's bits of it go to a
                        // list (one entry per trip and column) that interp_edge_fix_kernel ORs in afterwards.
                        if (first_bit < e_lo) p.edge_words[(int64_t)c * ntrips + trip] = (uint64_t)(uint32_t)((o_trip - (int64_t)sh_o) >> 5) | ((uint64_t)x32 << 32);
                        else wdst[wi] = x32;
                    }
                }
                }
            }
            wave_lds_order();
        }
        if (!staged) { for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_down(nvalid, o); }
        // the column's null count without a pass over its bitmap: one partial per trip, summed by interp_edge_fix_kernel
        if (p.trip_valid && lane == 0) p.trip_valid[(int64_t)c * ntrips + trip] = nvalid;
        W2_STAMP(3);   // flush
    }
    };   // column_loop
    if (staged_rt) column_loop(std::true_type{});
    else column_loop(std::false_type{});
    if (__ballot(far)) {
        if (lane == 0 && !__hip_atomic_load(&p.status[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[7], 1u);
    }
#ifdef BOWGPU_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(reinterpret_cast<unsigned long long *>(p.status + 32) + i, st_acc[i]);
        atomicAdd(reinterpret_cast<unsigned long long *>(p.status + 32) + 7, 1ull);
    }
#endif
}

template <bool kIncl>
__global__ __launch_bounds__(64, 4) void interp_wave3_kernel(const InterpParams p, const int64_t ntrips, const int64_t trips_per_xcd) {
    constexpr int kRuns = kIncl ? kT3Rows : kT3Rows / 4;
    __shared__ Wave3Lds<kRuns> L;
    const int64_t b = blockIdx.x;
    const int64_t trip = (b & 7) * trips_per_xcd + (b >> 3);   // XCD-contiguous runs of trips: the rows around a trip's ends are in that XCD's L2
    if (trip >= ntrips) return;
    if (p.aligned16 && p.n - trip * kT3Rows >= kT3Rows) wave3_trip<kIncl, true, kRuns>(p, ntrips, trip, L);
    else wave3_trip<kIncl, false, kRuns>(p, ntrips, trip, L);
}

// the bits the trips of interp_wave3_kernel left for the words they share with the trip before them (one entry per trip and column:
// word index | bits << 32; 0: nothing).  Two entries never name the same word: a trip in the middle of the frame holds 512 rows.
// Also the trips' valid-output counts when the kernel kept them, as kEdgeBlocks partial sums per column (plain stores - the host adds
// them up: 1564 agent-scope adds to two addresses cost 16 us here); blockIdx.y = column, grid-stride over the trips.
__global__ __launch_bounds__(256) void interp_edge_fix_kernel(const InterpParams p, const int64_t ntrips) {
    __shared__ unsigned long long sh[4];
    const int c = blockIdx.y;
    unsigned long long v = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntrips; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t e = p.edge_words[(int64_t)c * ntrips + t];
        const uint32_t bits = (uint32_t)(e >> 32);
        if (bits) p.cols[c].out_valid_words[(uint32_t)e] |= bits;
        if (p.trip_valid) v += p.trip_valid[(int64_t)c * ntrips + t];
    }
    if (!p.trip_valid) return;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    // (the partial counts and the status words - interp_wave3_kernel's, the launch in front of this one - go straight into the host's
    // registered block: no copy command behind the launch)
    if (threadIdx.x == 0) p.valid_counts[c * kInterpEdgeBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    if (blockIdx.x == 0 && c == 0 && threadIdx.x < 16 && p.host_status) p.host_status[threadIdx.x] = p.status[threadIdx.x];
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

// interp_wave3_kernel's shapes: as above, but only each 512-row TRIP has to span less than 2^31 (checked by the kernel itself)
bool interp_wide32(const Plan &plan, int64_t kq) {
    const int64_t lim53 = 1ll << 53;
    return kq < 0 && plan.first_ts >= plan.s0 && plan.interval < (1ll << 31) && plan.first_ts > -lim53 && plan.last_ts < lim53 && plan.last_ts >= plan.first_ts;
}

int launch_interp_count(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t kq, int has_left, int64_t left_ts,
                        int32_t *tile_local, int32_t *super_sum, int64_t *super_before, int64_t *d_total, uint32_t *status, int64_t *host_back) {
    static_assert(kCountThreads / 64 * kITile == kInterpSuperRows, "a workgroup of the count kernel is one super-tile");
    const int64_t nsuper = interp_supers(n);
    if (nsuper <= 0) return 0;
    if (kq >= 0 || plan.first_ts < plan.s0)
        hipLaunchKernelGGL(interp_quirk_kernel, dim3(1), dim3(64), 0, c->stream, ts, n, plan.s0, plan.interval, kq, status);
    if (interp_fast32(plan, kq))
        hipLaunchKernelGGL(interp_count_kernel<true>, dim3((unsigned)nsuper), dim3(kCountThreads), 0, c->stream, ts, n, plan.s0, plan.interval,
                           plan.magic, magic32_make(plan.interval), has_left, left_ts, tile_local, super_sum, status);
    else
        hipLaunchKernelGGL(interp_count_kernel<false>, dim3((unsigned)nsuper), dim3(kCountThreads), 0, c->stream, ts, n, plan.s0, plan.interval,
                           plan.magic, Magic32{0, 0, 0}, has_left, left_ts, tile_local, super_sum, status);
    hipLaunchKernelGGL(interp_super_scan_kernel, dim3(1), dim3(1024), 0, c->stream, super_sum, nsuper, super_before, d_total, status, host_back);
    BG_HIP(hipGetLastError());
    return 0;
}

int64_t interp_supers(int64_t n) { return (n + kInterpSuperRows - 1) / kInterpSuperRows; }
int64_t interp_tiles(int64_t n) { return (n + kITile - 1) / kITile; }
void interp_magic32(int64_t interval, uint32_t *m, uint32_t *sh1, uint32_t *sh2) {
    const Magic32 x = magic32_make(interval);
    *m = x.m; *sh1 = x.sh1; *sh2 = x.sh2;
}

// does interp_wave3_kernel take this call?  (It looks a synthetic row's neighbour points up without the neighbour index - a bounded
// walk, status[7] when it does not reach - so the host builds the index only for the workgroup kernel and for a repeat.)
bool interp_takes_wave3(const InterpParams &p) {
    const bool force_tile = (route_mask() & BOWGPU_ROUTE_INTERP_TILE) != 0 && !p.inclusive;
    return p.drop == 0 && p.kq < 0 && (p.fast32 || p.wide32) && p.allow_wave2 && !force_tile;
}

int launch_interp_tiles(Ctx *c, const InterpParams &p) {
    const int64_t ntiles = (p.n + kITile - 1) / kITile;
    if (ntiles <= 0) return 0;
    // interp_wave3_kernel for every shape it can take (its trip-relative form only needs each TRIP within 2^31: wide32); the workgroup
    // kernel for the rest - 64-bit window ids, dropped rows, the -1 sentinel window - and for the redo of a call one of whose trips
    // overflowed wave3's lists (allow_wave2 == 0).  BOWGPU_ROUTE_INTERP_TILE: the test switch that runs an exclusive call through the
    // workgroup kernel (the second opinion of the tests).  Inclusive windows are built by wave3 alone.
    static_assert(kT3Rows == kITile, "interp_wave3_kernel's trips are the count kernel's tiles");
    if (interp_takes_wave3(p)) {
        const int64_t ntrips = (p.n + kT3Rows - 1) / kT3Rows, per_xcd = (ntrips + 7) / 8;
        if (per_xcd * 8 > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
        if (!p.edge_words) return fail(BOWGPU_ERR_ARG, "internal: interp_wave3_kernel needs its edge list");
        if (p.inclusive) hipLaunchKernelGGL((interp_wave3_kernel<true>), dim3((unsigned)(per_xcd * 8)), dim3(64), 0, c->stream, p, ntrips, per_xcd);
        else hipLaunchKernelGGL((interp_wave3_kernel<false>), dim3((unsigned)(per_xcd * 8)), dim3(64), 0, c->stream, p, ntrips, per_xcd);
        // (kInterpEdgeBlocks workgroups per column whatever the size: the host reads that many partial counts)
        hipLaunchKernelGGL(interp_edge_fix_kernel, dim3(kInterpEdgeBlocks, (unsigned)p.ncols), dim3(256), 0, c->stream, p, ntrips);
    }
    else if (p.inclusive)   // no other kernel builds inclusive windows: never fall through to an exclusive one
        return fail(BOWGPU_ERR_UNSUPPORTED, "Interpolate on inclusive windows: this shape is outside the device path");
    else if (p.fast32) hipLaunchKernelGGL(interp_tile_kernel<true>, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, p);
    else hipLaunchKernelGGL(interp_tile_kernel<false>, dim3((unsigned)ntiles), dim3(kIThreads), 0, c->stream, p);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
