// rolling_fast.hip — the lean kernel for the headline family of Rolling.Aggregate:
// WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows over
// exclusive windows (reference rolling/rolling.go:177-239 + rolling/aggregation.go:190-238 +
// rolling/aggregation/{windowstart,sum,arithmeticmean,minmax,count,firstlast}.go).
// Inclusive windows, the time-weighted reducers and inputs with rows below s0 take the general
// kernel in rolling_agg.hip; results are identical where both apply.
//
// Same data-parallel restatement as rolling_agg.hip (row i is a head when its window id differs
// from row i-1's; heads -> LDS segment list -> ONE lane walks one window in row order, i.e. the
// reference's summation order, bit-exact) organised for the machine:
//   * one WAVEFRONT (64 lanes) owns a tile of 512 rows + 128 look-ahead rows and never meets
//     another wave: no s_barrier, no cross-wave prefix.  ~20 such waves are resident per CU at
//     different phases, which is what keeps HBM reads in flight while others reduce;
//   * coalesced 16-B/lane loads (2 rows per lane, 5 chunks of 128 rows);
//   * the window id is computed RELATIVE to the tile's first window in 32-bit arithmetic (one
//     v_mul_hi_u32 per row) whenever the tile's ts span and the interval fit 32 bits — decided per
//     tile on the scalar unit; otherwise the exact 64-bit multiply-high division is used;
//   * neighbour rows come from DPP wave shifts / v_readlane, not memory;
//   * heads are compacted with ballot + mbcnt and a running scalar count (chunks are consecutive);
//   * the walk is specialised at compile time on {has nulls, int64 values, min/max, first/last};
//   * outputs: lane q stores window slot wid(q) (8-B coalesced stores); validity bits are assembled in
//     LDS and flushed as whole words, atomicOr only for the boundary words shared with neighbours.
//
// HBM-bound: algorithmic bytes = 8 (ts) + 8 per value column (+1/8 per nullable column) per row.

#include <stdlib.h>

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kWave = 64;
constexpr int kTileW = 512;               // rows owned by a wavefront
constexpr int kHaloW = 128;               // look-ahead rows
constexpr int kRowsW = kTileW + kHaloW;   // 640
constexpr int kChunksW = kRowsW / 128;    // 5 (the last one is the halo)
constexpr int kSpanBitsW = kTileW + 64;
constexpr int kSpanWordsW = kSpanBitsW / 32;
constexpr int kMaxNullableW = 4;
constexpr int kGapInlineW = 8;
constexpr uint32_t kSat16 = 0xFFFFu;

struct WaveShared {
    uint64_t val[kRowsW];
    uint32_t seg[kRowsW + 2];       // heads in row order: local row | (window id - w0) << 16 (0xFFFF => recompute from ts)
    uint32_t vbits[kRowsW / 32 + 2];
    uint32_t obits[kMaxNullableW][kSpanWordsW + 1];
};

__device__ __forceinline__ uint32_t magic_div32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}

// value of lane-1 (wave shift right by one lane); lane 0 receives `lane0`
__device__ __forceinline__ uint32_t from_left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t from_left64(uint64_t x, uint64_t lane0) {
    const uint32_t lo = from_left32((uint32_t)x, (uint32_t)lane0);
    const uint32_t hi = from_left32((uint32_t)(x >> 32), (uint32_t)(lane0 >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// One wave only: LDS operations of a wave execute in order, so a write phase followed by reads from
// other lanes needs neither s_barrier nor a wait - only that the compiler keeps the program order.
// (__syncthreads() would also drain vmcnt(0), i.e. the next tile's prefetch.)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the walk of one window, specialised at compile time
template <bool kNulls, bool kInt, bool kMinMax, bool kFirstLast>
__device__ __forceinline__ void walk(const WaveShared &sh, int r0, int r1, Stats &s) {
    double sum = 0.0;
    if (!kNulls) {
        const uint64_t raw0 = sh.val[r0];
        double mn = kInt ? (double)(int64_t)raw0 : __longlong_as_double((long long)raw0);
        double mx = mn;
        int r = r0;
        for (; r + 1 < r1; r += 2) {  // two LDS values per round trip; the two adds keep the row order
            const uint64_t rawa = sh.val[r], rawb = sh.val[r + 1];
            const double xa = kInt ? (double)(int64_t)rawa : __longlong_as_double((long long)rawa);
            const double xb = kInt ? (double)(int64_t)rawb : __longlong_as_double((long long)rawb);
            sum += xa;
            sum += xb;
            if (kMinMax) {
                if (xa < mn) mn = xa;
                if (xa > mx) mx = xa;
                if (xb < mn) mn = xb;
                if (xb > mx) mx = xb;
            }
        }
        if (r < r1) {
            const uint64_t raw = sh.val[r];
            const double x = kInt ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
            sum += x;
            if (kMinMax) {
                if (x < mn) mn = x;
                if (x > mx) mx = x;
            }
        }
        s.sum = sum;
        s.count = r1 - r0;
        s.has_value = 1;
        s.vmin = mn; s.vmax = mx;
        if (kFirstLast) { s.first_bits = raw0; s.last_bits = sh.val[r1 - 1]; }
    } else {
        int64_t count = 0;
        double mn = 0.0, mx = 0.0;
        uint64_t first = 0, last = 0;
        for (int r = r0; r < r1; r++) {
            const bool ok = (sh.vbits[r >> 5] >> (r & 31)) & 1u;
            if (!ok) continue;
            const uint64_t raw = sh.val[r];
            const double x = kInt ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
            sum += x;
            if (kMinMax || kFirstLast) {
                if (count == 0) { mn = x; mx = x; first = raw; }
                else if (kMinMax) { if (x < mn) mn = x; if (x > mx) mx = x; }
                last = raw;
            }
            count++;
        }
        s.sum = sum;
        s.count = count;
        s.has_value = count > 0;
        s.vmin = mn; s.vmax = mx; s.first_bits = first; s.last_bits = last;
    }
}

__device__ __forceinline__ void walk_dispatch(int variant, const WaveShared &sh, int r0, int r1, Stats &s) {
    // variant: bit0 nulls, bit1 int64 values, bit2 min/max, bit3 first/last  (wave-uniform)
    switch (variant) {
#define BG_CASE(v) case v: walk<((v) & 1) != 0, ((v) & 2) != 0, ((v) & 4) != 0, ((v) & 8) != 0>(sh, r0, r1, s); break;
        BG_CASE(0) BG_CASE(1) BG_CASE(2) BG_CASE(3) BG_CASE(4) BG_CASE(5) BG_CASE(6) BG_CASE(7)
        BG_CASE(8) BG_CASE(9) BG_CASE(10) BG_CASE(11) BG_CASE(12) BG_CASE(13) BG_CASE(14) BG_CASE(15)
#undef BG_CASE
    }
}

}  // namespace

// kPersist: the wave walks tiles t0, t0 + stride, ... and keeps the NEXT tile's global loads in flight while it
// reduces the current one (the registers of the current tile's ts / values are dead by then), so it never sits
// idle on HBM latency while holding its LDS.
template <bool kPersist>
__global__ __launch_bounds__(kWave, kPersist ? 4 : 5) void rolling_wave_kernel(const AggParams p, const int64_t ntiles,
                                                                               const int64_t tiles_per_xcd) {
    __shared__ WaveShared sh;

    // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs, so give every XCD a
    // contiguous run of tiles (a tile's look-ahead rows are its right neighbour's first rows: same L2).
    const int64_t b = blockIdx.x;
    const int64_t xcd_lo = (b & 7) * tiles_per_xcd;
    int64_t xcd_hi = xcd_lo + tiles_per_xcd;
    if (xcd_hi > ntiles) xcd_hi = ntiles;
    const int64_t t_step = kPersist ? (int64_t)(gridDim.x >> 3) : tiles_per_xcd;
    int64_t tile = xcd_lo + (b >> 3);
    if (tile >= xcd_hi) return;

    const int lane = threadIdx.x;
    const int64_t n = p.n;
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    const bool ts_vec = (reinterpret_cast<uintptr_t>(ts) & 15) == 0;
    const uint64_t *__restrict__ vp0 = p.ncols > 0 ? reinterpret_cast<const uint64_t *>(p.cols[0].values) : nullptr;
    const bool v0_vec = (reinterpret_cast<uintptr_t>(vp0) & 15) == 0;
    const int last_val_slot = p.last_val_slot;  // the last column slot whose values get staged (the next tile's value prefetch follows it)

    uint64_t ta[kChunksW], tb[kChunksW];
    uint64_t va[kChunksW], vb[kChunksW];
    auto is_interior = [&](int64_t t) { return (t * kTileW + kRowsW <= n) && ts_vec && v0_vec; };
    auto load_ts = [&](int64_t t, bool inter) {
        const int64_t b0 = t * kTileW;
        if (inter) {
            const ulonglong2 *tp = reinterpret_cast<const ulonglong2 *>(ts + b0) + lane;
#pragma unroll
            for (int j = 0; j < kChunksW; j++) { const ulonglong2 x = tp[j * 64]; ta[j] = x.x; tb[j] = x.y; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksW; j++) load_pair(ts, b0 + j * 128 + 2 * lane, n, ts_vec, ta[j], tb[j]);
        }
    };
    auto load_v0 = [&](int64_t t, bool inter) {
        if (vp0 == nullptr) return;
        const int64_t b0 = t * kTileW;
        if (inter) {
            const ulonglong2 *vq = reinterpret_cast<const ulonglong2 *>(vp0 + b0) + lane;
#pragma unroll
            for (int j = 0; j < kChunksW; j++) { const ulonglong2 x = vq[j * 64]; va[j] = x.x; vb[j] = x.y; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksW; j++) load_pair(vp0, b0 + j * 128 + 2 * lane, n, v0_vec, va[j], vb[j]);
        }
    };

    // ---- 1. loads: ts, then the first value column right behind it.  Interior tiles (all rows exist,
    // 16-B aligned columns) take straight-line 16-B/lane loads; only the last tile / odd offsets are guarded.
    bool interior = is_interior(tile);
    load_ts(tile, interior);
    load_v0(tile, interior);
    bool ts_prefetched = true, v0_prefetched = true;

  for (; tile < xcd_hi; tile += t_step) {
    const int64_t base = tile * kTileW;
    const int nloc = (int)((n - base) < kRowsW ? (n - base) : kRowsW);  // rows of this tile(+halo) that exist
    interior = is_interior(tile);
    if (!ts_prefetched) load_ts(tile, interior);
    if (!v0_prefetched) load_v0(tile, interior);
    int staged_slot = vp0 != nullptr ? 0 : -1;
    // the next tile of this wave (persistent variant): prefetched only through the straight-line path
    const int64_t next_tile = tile + t_step;
    const bool prefetch_next = kPersist && next_tile < xcd_hi && is_interior(next_tile);
    ts_prefetched = prefetch_next;
    v0_prefetched = prefetch_next;
    // the row left of the tile and the tile's last row (scalar loads): first head flag, order check, ts span
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    const int64_t ts_last = p.ts[base + nloc - 1];

    // scalar: the tile's first window (one exact 64-bit division on the SALU)
    const int64_t ts_first = (int64_t)readlane64(ta[0], 0);
    const uint64_t w0 = magic_div((uint64_t)ts_first - (uint64_t)p.s0, p.magic);
    const int64_t ws0 = p.s0 + (int64_t)(w0 * (uint64_t)p.interval);
    const bool fast32 = p.fits32 && ts_last >= ts_first && (uint64_t)(ts_last - ws0) < 0xFFFFFFF0ull;

    // ---- 2. local window ids, head flags, compaction (chunks are consecutive: running scalar count)
    bool unsorted = false;
    int nseg_total = 0, nseg_owned = 0;
    int64_t left_ts = left0;
    // window of the row left of the current chunk: from the scalar unit for chunk 0, then lane 63 of the previous chunk
    const uint32_t base_lo = (uint32_t)ws0;
    uint32_t left_lw = 0xFFFFFFFEu;  // an id no row of the tile has: the first row of the frame, or a left row in an earlier window
    uint64_t left_w = ~0ull;
    if (left0 != INT64_MIN) {
        if (fast32) { if (left0 >= ws0) left_lw = magic_div32((uint32_t)left0 - base_lo, p.m32, p.sh1_32, p.sh2_32); }
        else left_w = magic_div((uint64_t)left0 - (uint64_t)p.s0, p.magic);
    }
    // no local window id saturates its 16-bit slot => wid = w0 + id without a per-lane fallback test
    const bool no_sat = fast32 && magic_div32((uint32_t)ts_last - base_lo, p.m32, p.sh1_32, p.sh2_32) < kSat16;
#pragma unroll
    for (int j = 0; j < kChunksW; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const int64_t prev_ts = (int64_t)from_left64((uint64_t)tb[j], (uint64_t)left_ts);
        if (pa && prev_ts > tsa) unsorted = true;
        if (pb && tsa > tsb) unsorted = true;
        bool ha, hb;
        uint32_t la, lb;
        if (fast32) {
            la = magic_div32((uint32_t)tsa - base_lo, p.m32, p.sh1_32, p.sh2_32);
            lb = magic_div32((uint32_t)tsb - base_lo, p.m32, p.sh1_32, p.sh2_32);
            const uint32_t lprev = from_left32(lb, left_lw);
            ha = pa && (la != lprev);
            hb = pb && (lb != la);
            left_lw = (uint32_t)__builtin_amdgcn_readlane((int)lb, 63);
        } else {
            const uint64_t wa = magic_div((uint64_t)tsa - (uint64_t)p.s0, p.magic);
            const uint64_t wb = magic_div((uint64_t)tsb - (uint64_t)p.s0, p.magic);
            const uint64_t wprev = from_left64(wb, left_w);
            ha = pa && (wa != wprev);
            hb = pb && (wb != wa);
            left_w = readlane64(wb, 63);
            const uint64_t da = wa - w0, db = wb - w0;
            la = da >= kSat16 ? kSat16 : (uint32_t)da;
            lb = db >= kSat16 ? kSat16 : (uint32_t)db;
        }
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        if (ha) { sh.seg[pos] = (uint32_t)l | ((la >= kSat16 ? kSat16 : la) << 16); pos++; }
        if (hb) { sh.seg[pos] = (uint32_t)(l + 1) | ((lb >= kSat16 ? kSat16 : lb) << 16); }
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kChunksW - 2) nseg_owned = nseg_total;  // heads inside the 512 owned rows
        left_ts = (int64_t)readlane64(tb[j], 63);         // last row of this chunk = left neighbour of the next
    }
    if (unsorted) atomicOr(&p.status[0], 1u);
    if (prefetch_next) load_ts(next_tile, true);  // this tile's ts registers are dead

    const bool reaches_end = base + kRowsW >= n;
    const int64_t wid_end = p.wid_base + p.W;

    auto wid_of = [&](uint32_t e) -> uint64_t {
        const uint32_t d = e >> 16;
        if (no_sat || d != kSat16) return w0 + d;
        const int64_t t = p.ts[base + (e & 0xFFFFu)];
        return magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
    };
    auto wid_at = [&](int q) -> uint64_t { return wid_of(sh.seg[q]); };

    // ---- 3./4. one pass per column slot
    bool first_pass = true;
    for (int slot = p.first_pass_slot; slot < p.ncols; slot++) {
        const unsigned my_mask = p.pass_mask[slot + 1];
        if (my_mask == 0) continue;
        const unsigned pfl = p.pass_flags[slot + 1];
        const bool any_nullable = pfl & kPassNullable, need_vals = pfl & kPassNeedVals;
        const bool need_mm = pfl & kPassMinMax, need_fl = pfl & kPassFirstLast;

        const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
        const bool has_nulls = cd && cd->vbits != nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;
        const int variant = (has_nulls ? 1 : 0) | (col_type == BOWGPU_INT64 ? 2 : 0) | (need_mm ? 4 : 0) | (need_fl ? 8 : 0);

        wave_lds_fence();  // previous pass finished with sh.val / sh.vbits / sh.obits
        if (cd && need_vals) {
            if (staged_slot != slot) {
                const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd->values);
                const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
                if (interior && vvec) {
                    const ulonglong2 *vq = reinterpret_cast<const ulonglong2 *>(vp + base) + lane;
#pragma unroll
                    for (int j = 0; j < kChunksW; j++) { const ulonglong2 t = vq[j * 64]; va[j] = t.x; vb[j] = t.y; }
                } else {
#pragma unroll
                    for (int j = 0; j < kChunksW; j++) load_pair(vp, base + j * 128 + 2 * lane, n, vvec, va[j], vb[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < kChunksW; j++)
                *reinterpret_cast<ulonglong2 *>(&sh.val[j * 128 + 2 * lane]) = make_ulonglong2(va[j], vb[j]);
            staged_slot = -2;
            if (has_nulls && lane < kRowsW / 32) sh.vbits[lane] = load_vbits32(*cd, base + 32 * (int64_t)lane);
            if (prefetch_next && slot == last_val_slot) load_v0(next_tile, true);  // the value registers are dead
        }
        if (any_nullable) {
            (&sh.obits[0][0])[lane] = 0u;  // 4 x 19 words: two stores per lane at most
            if (lane < kMaxNullableW * (kSpanWordsW + 1) - kWave) (&sh.obits[0][0])[kWave + lane] = 0u;
        }
        wave_lds_fence();

        const uint64_t wid_first = nseg_owned > 0 ? wid_at(0) : 0;
        const int64_t slot_first = (int64_t)(wid_first - (uint64_t)p.wid_base);
        const int64_t span0 = slot_first & ~(int64_t)31;
        bool any_big_gap = false;

        for (int q = lane; q < nseg_owned; q += kWave) {
            const uint32_t e0 = sh.seg[q], e1 = sh.seg[q + 1];  // (one ds_read2_b32; entry q+1 is only used when it exists)
            const int r0 = (int)(e0 & 0xFFFFu);
            const uint64_t wid = wid_of(e0);
            int r1;
            uint64_t next_wid;
            if (q + 1 < nseg_total) {
                r1 = (int)(e1 & 0xFFFFu);
                next_wid = wid_of(e1);
            } else if (reaches_end) {
                r1 = nloc;
                next_wid = (uint64_t)wid_end;
            } else {
                // rows run past the look-ahead: hand the window (all its column passes) to the cooperative path
                if (first_pass) {
                    push_long_window(p.status, p.long_list, p.long_cap, tile, wid, base + r0);
                }
                continue;
            }
            Stats st;
            stats_init(st);
            if (need_vals) walk_dispatch(variant, sh, r0, r1, st);
            const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
            const int64_t oslot = (int64_t)(wid - (uint64_t)p.wid_base);
            const int64_t nrows = r1 - r0;
            const int64_t gap = (int64_t)(next_wid - wid) - 1;
            const bool big_gap = gap > kGapInlineW;
            any_big_gap |= big_gap;
            if ((uint64_t)oslot < (uint64_t)p.W) {
                int nb = 0;
                for (unsigned m = my_mask; m; m &= m - 1) {
                    const AggDesc &a = p.aggs[__ffs(m) - 1];
                    Val v = finish_val(reduce_val(a.kind, st, nrows, win_start, p.interval, col_type == BOWGPU_INT64), a);
                    reinterpret_cast<uint64_t *>(a.out_values)[oslot] = v.bits;
                    if (a.out_valid) {
                        if (v.valid) {
                            const int64_t lb = oslot - span0;
                            if (lb >= 0 && lb < kSpanBitsW) atomicOr(&sh.obits[nb][lb >> 5], 1u << (lb & 31));
                            else atomicOr(&a.out_valid[oslot >> 5], 1u << (oslot & 31));
                        }
                        nb++;
                    }
                }
                if (gap > 0 && !big_gap) {  // the few empty windows right after this one (A.9 "Empty slice")
                    Stats e;
                    stats_init(e);
                    for (int64_t gk = 1; gk <= gap; gk++) {
                        const int64_t gs = oslot + gk;
                        if (gs >= p.W) break;
                        const int64_t gstart = win_start + gk * p.interval;
                        for (unsigned m = my_mask; m; m &= m - 1) {
                            const AggDesc &a = p.aggs[__ffs(m) - 1];
                            Val v = finish_val(reduce_val(a.kind, e, 0, gstart, p.interval, col_type == BOWGPU_INT64), a);
                            reinterpret_cast<uint64_t *>(a.out_values)[gs] = v.bits;
                        }
                    }
                }
            }
        }
        wave_lds_fence();

        // ---- big gaps (sparse data): the whole wave writes the empty windows, coalesced
        if (__ballot(any_big_gap)) {
            for (int q = 0; q < nseg_owned; q++) {
                if (q + 1 >= nseg_total && !reaches_end) continue;
                const uint64_t wid = wid_at(q);
                const uint64_t next_wid = (q + 1 < nseg_total) ? wid_at(q + 1) : (uint64_t)wid_end;
                const int64_t gap = (int64_t)(next_wid - wid) - 1;
                if (gap <= kGapInlineW) continue;
                const int64_t oslot = (int64_t)(wid - (uint64_t)p.wid_base);
                const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
                Stats e;
                stats_init(e);
                for (int64_t gk = 1 + lane; gk <= gap; gk += kWave) {
                    const int64_t gs = oslot + gk;
                    if (gs < 0 || gs >= p.W) break;
                    const int64_t gstart = win_start + gk * p.interval;
                    for (unsigned m = my_mask; m; m &= m - 1) {
                        const AggDesc &a = p.aggs[__ffs(m) - 1];
                        Val v = finish_val(reduce_val(a.kind, e, 0, gstart, p.interval, col_type == BOWGPU_INT64), a);
                        reinterpret_cast<uint64_t *>(a.out_values)[gs] = v.bits;
                    }
                }
            }
        }

        // ---- flush the validity bits assembled in LDS (whole words stored, boundary words OR-ed)
        if (any_nullable && nseg_owned > 0) {
            int64_t slot_end;
            {
                const int ql = nseg_owned - 1;
                if (ql + 1 < nseg_total) slot_end = (int64_t)(wid_at(ql + 1) - (uint64_t)p.wid_base);
                else if (reaches_end) slot_end = p.W;
                else slot_end = (int64_t)(wid_at(ql) - (uint64_t)p.wid_base);  // long window: not ours
            }
            if (slot_end > p.W) slot_end = p.W;
            int64_t lim = slot_end - span0;
            if (lim > kSpanBitsW) lim = kSpanBitsW;
            const int nwords = (int)((lim + 31) >> 5);
            int nb = 0;
            for (unsigned m = my_mask; m; m &= m - 1) {
                const AggDesc &a = p.aggs[__ffs(m) - 1];
                if (!a.out_valid) continue;
                for (int w = lane; w < nwords; w += kWave) {
                    const uint32_t bits = sh.obits[nb][w];
                    const int64_t gw = (span0 >> 5) + w;
                    const int64_t wlo = gw << 5, whi = wlo + 32;
                    if (gw < 0 || wlo >= p.W) continue;  // only reachable on unsorted input
                    const bool whole = wlo >= slot_first && whi <= slot_end;
                    if (whole) a.out_valid[gw] = bits;
                    else if (bits) atomicOr(&a.out_valid[gw], bits);
                }
                nb++;
            }
        }
        first_pass = false;
    }
    if (prefetch_next && last_val_slot < 0) load_v0(next_tile, true);
    wave_lds_fence();
    if (!kPersist) break;  // one tile per wave: no loop for the compiler to carry registers around
  }  // tiles of this wave
}

int launch_rolling_fast(Ctx *c, const AggParams &p) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileW - 1) / kTileW;
    const int64_t per_xcd = (ntiles + 7) / 8;
    // (a persistent variant - each wave prefetching its next tile - measured SLOWER on MI355X: 4.24 vs 3.78 ms at 1e9 rows, fewer
    // resident waves at 127 VGPRs; the kPersist form of the kernel is kept in the source for that record, never launched)
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    hipLaunchKernelGGL(rolling_wave_kernel<false>, dim3((unsigned)grid), dim3(kWave), 0, c->stream, p, ntiles, per_xcd);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
