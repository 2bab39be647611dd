// rolling_fast.hip — the lean tile kernel for the headline family of Rolling.Aggregate:
// WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows over
// exclusive windows (reference rolling/rolling.go:177-239 + rolling/aggregation.go:190-238 +
// rolling/aggregation/{windowstart,sum,arithmeticmean,minmax,count,firstlast}.go).
// Inclusive windows, the time-weighted reducers and inputs with rows below s0 take the general
// kernel in rolling_agg.hip; results are identical where both apply.
//
// Same data-parallel restatement as rolling_agg.hip (heads -> LDS segment list -> one lane walks
// one window in row order => reference summation order, bit-exact), but the per-row work is cut
// to the bone, because at 16 B/row the HBM roofline leaves ~1.7 wave-instructions per row:
//   * the window id is computed RELATIVE to the tile's first window, in 32-bit arithmetic
//     (one v_mul_hi_u32) whenever the tile's ts span and the interval fit 32 bits — decided per
//     tile from two scalar loads; otherwise the exact 64-bit multiply-high path is used;
//   * the per-tile base window is derived on the scalar unit (s_load + SALU), not per lane;
//   * neighbour rows come from DPP wave shifts, not LDS; lane 0 takes its neighbour from a
//     scalar load;
//   * the walk is specialised at compile time on {has nulls, int64 values, min/max, first/last}.

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 2048;
constexpr int kHalo = 128;
constexpr int kRows = kTile + kHalo;
constexpr int kChunks = 4;
constexpr int kCnt = kChunks * 4 + 1;
constexpr int kSpanBits = 2048 + 64;
constexpr int kSpanWords = kSpanBits / 32;
constexpr int kMaxNullable = 8;
constexpr int kGapInline = 4;
constexpr int kGapList = 64;

struct FastShared {
    uint64_t val[kRows];
    uint32_t seg_lw[kRows + 2];    // window id - w0 (kSat => recompute from ts)
    uint16_t seg_row[kRows + 2];   // local head row
    uint32_t vbits[kRows / 32 + 2];
    uint32_t obits[kMaxNullable][kSpanWords];
    int cnt[kCnt + 3];
    int pre[kCnt + 3];
    int gap_q[kGapList];
    int gap_n;
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

// ---- the walk of one window, specialised at compile time
template <bool kNulls, bool kInt, bool kMinMax, bool kFirstLast>
__device__ __forceinline__ void walk(const FastShared &sh, int r0, int r1, Stats &s) {
    double sum = 0.0;
    if (!kNulls) {
        const uint64_t raw0 = sh.val[r0];
        double mn = kInt ? (double)(int64_t)raw0 : __longlong_as_double((long long)raw0);
        double mx = mn;
        for (int r = r0; r < r1; r++) {
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

__device__ __forceinline__ void walk_dispatch(int variant, const FastShared &sh, int r0, int r1, Stats &s) {
    // variant: bit0 nulls, bit1 int64 values, bit2 min/max, bit3 first/last  (workgroup-uniform)
    switch (variant) {
#define BG_CASE(v) case v: walk<((v) & 1) != 0, ((v) & 2) != 0, ((v) & 4) != 0, ((v) & 8) != 0>(sh, r0, r1, s); break;
        BG_CASE(0) BG_CASE(1) BG_CASE(2) BG_CASE(3) BG_CASE(4) BG_CASE(5) BG_CASE(6) BG_CASE(7)
        BG_CASE(8) BG_CASE(9) BG_CASE(10) BG_CASE(11) BG_CASE(12) BG_CASE(13) BG_CASE(14) BG_CASE(15)
#undef BG_CASE
    }
}

}  // namespace

__global__ __launch_bounds__(kBlock) void rolling_fast_kernel(const AggParams p, const int64_t ntiles,
                                                              const int64_t tiles_per_xcd) {
    __shared__ FastShared sh;

    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs
    if (tile >= ntiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t base = tile * kTile;
    const int64_t n = p.n;
    const int nloc = (int)((n - base) < kRows ? (n - base) : kRows);  // rows of this tile(+halo) that exist
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    const bool ts_vec = (reinterpret_cast<uintptr_t>(ts) & 15) == 0;

    // ---- 1. vector loads first (latency), scalar prologue underneath them
    uint64_t ta[kChunks + 1], tb[kChunks + 1];
#pragma unroll
    for (int j = 0; j < kChunks; j++) load_pair(ts, base + j * 512 + 2 * tid, n, ts_vec, ta[j], tb[j]);
    ta[kChunks] = 0; tb[kChunks] = 0;
    if (wave == 0) load_pair(ts, base + kTile + 2 * tid, n, ts_vec, ta[kChunks], tb[kChunks]);

    uint64_t va[kChunks + 1], vb[kChunks + 1];
    int staged_slot = -1;
    if (p.ncols > 0 && p.cols[0].values != nullptr) {
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(p.cols[0].values);
        const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
        for (int j = 0; j < kChunks; j++) load_pair(vp, base + j * 512 + 2 * tid, n, vvec, va[j], vb[j]);
        va[kChunks] = 0; vb[kChunks] = 0;
        if (wave == 0) load_pair(vp, base + kTile + 2 * tid, n, vvec, va[kChunks], vb[kChunks]);
        staged_slot = 0;
    }

    // scalar: the tile's first window (one exact 64-bit division on the SALU) and its ts span
    const int64_t ts_first = p.ts[base];
    const int64_t ts_last = p.ts[base + nloc - 1];
    const uint64_t w0 = magic_div((uint64_t)ts_first - (uint64_t)p.s0, p.magic);
    const int64_t ws0 = p.s0 + (int64_t)(w0 * (uint64_t)p.interval);
    const bool fast32 = p.fits32 && ts_last >= ts_first && (uint64_t)(ts_last - ws0) < 0xFFFFFFF0ull;

    // ---- 2. local window ids, head flags, compaction
    unsigned long long mask_a[kChunks + 1], mask_b[kChunks + 1];
    uint32_t lwa[kChunks + 1], lwb[kChunks + 1];
    bool unsorted = false;
#pragma unroll
    for (int j = 0; j <= kChunks; j++) {
        if (j == kChunks && wave != 0) { mask_a[j] = 0; mask_b[j] = 0; lwa[j] = 0; lwb[j] = 0; continue; }
        const int l = (j < kChunks) ? (j * 512 + 2 * tid) : (kTile + 2 * tid);
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        // the row left of this wave's first row, from the scalar unit
        const int l0 = (j < kChunks) ? (j * 512 + 128 * wave) : kTile;  // wave-uniform
        const int64_t g0 = base + l0;
        const bool has_left = g0 > 0 && l0 < nloc;
        const int64_t left_ts = has_left ? p.ts[g0 - 1] : INT64_MIN;
        const int64_t prev_ts = (int64_t)from_left64((uint64_t)tb[j], (uint64_t)left_ts);
        if (pa && prev_ts > tsa) unsorted = true;
        if (pb && tsa > tsb) unsorted = true;
        bool ha, hb;
        if (fast32) {
            const uint32_t base_lo = (uint32_t)ws0;
            const uint32_t la = magic_div32((uint32_t)tsa - base_lo, p.m32, p.sh1_32, p.sh2_32);
            const uint32_t lb = magic_div32((uint32_t)tsb - base_lo, p.m32, p.sh1_32, p.sh2_32);
            // left neighbour's local window; rows left of the tile's first window wrap to a huge id (!= any la)
            const uint32_t left_lw = (has_left && left_ts >= ws0) ? magic_div32((uint32_t)left_ts - base_lo, p.m32, p.sh1_32, p.sh2_32)
                                                                   : 0xFFFFFFFEu;
            const uint32_t lprev = from_left32(lb, left_lw);
            ha = pa && (la != lprev);
            hb = pb && (lb != la);
            lwa[j] = la; lwb[j] = lb;
        } else {
            const uint64_t wa = magic_div((uint64_t)tsa - (uint64_t)p.s0, p.magic);
            const uint64_t wb = magic_div((uint64_t)tsb - (uint64_t)p.s0, p.magic);
            const uint64_t left_w = has_left ? magic_div((uint64_t)left_ts - (uint64_t)p.s0, p.magic) : ~0ull;
            const uint64_t wprev = from_left64(wb, left_w);
            ha = pa && (wa != wprev);
            hb = pb && (wb != wa);
            const uint64_t da = wa - w0, db = wb - w0;
            lwa[j] = da >= kSat ? kSat : (uint32_t)da;
            lwb[j] = db >= kSat ? kSat : (uint32_t)db;
        }
        mask_a[j] = __ballot(ha);
        mask_b[j] = __ballot(hb);
        if (lane == 0) sh.cnt[(j < kChunks) ? (j * 4 + wave) : (kChunks * 4)] = __popcll(mask_a[j]) + __popcll(mask_b[j]);
    }
    if (tid == 0) sh.gap_n = 0;
    if (unsorted) atomicOr(&p.status[0], 1u);
    __syncthreads();
    // exclusive prefix of the 17 counters, once
    if (tid < 32) {
        int v = tid < kCnt ? sh.cnt[tid] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (tid >= o) incl += up;
        }
        sh.pre[tid < kCnt + 2 ? tid : kCnt + 2] = incl - v;  // pre[k] = heads before counter k ; pre[kCnt] = total
    }
    __syncthreads();
    const int nseg_owned = sh.pre[kChunks * 4];
    const int nseg_total = sh.pre[kCnt];
#pragma unroll
    for (int j = 0; j <= kChunks; j++) {
        if (j == kChunks && wave != 0) continue;
        int pos = sh.pre[(j < kChunks) ? (j * 4 + wave) : (kChunks * 4)];
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mask_a[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask_a[j], 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mask_b[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask_b[j], 0));
        const int l = (j < kChunks) ? (j * 512 + 2 * tid) : (kTile + 2 * tid);
        const bool ha = (mask_a[j] >> lane) & 1, hb = (mask_b[j] >> lane) & 1;
        if (ha) { sh.seg_row[pos] = (uint16_t)l; sh.seg_lw[pos] = lwa[j]; pos++; }
        if (hb) { sh.seg_row[pos] = (uint16_t)(l + 1); sh.seg_lw[pos] = lwb[j]; }
    }

    const bool reaches_end = base + kRows >= n;
    const int64_t wid_end = p.wid_base + p.W;

    auto seg_wid_of = [&](int q) -> uint64_t {
        const uint32_t d = sh.seg_lw[q];
        if (d != kSat) return w0 + d;
        const int64_t t = p.ts[base + sh.seg_row[q]];
        return magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
    };

    // ---- 3./4. one pass per column slot
    bool first_pass = true;
    for (int slot = -1; slot < p.ncols; slot++) {
        unsigned my_mask = 0;
        bool any_nullable = false, need_vals = false, need_mm = false, need_fl = false;
        for (int a = 0; a < p.naggs; a++) {
            if (p.aggs[a].slot != slot) continue;
            my_mask |= 1u << a;
            any_nullable |= (p.aggs[a].out_valid != nullptr);
            const int k = p.aggs[a].kind;
            need_vals |= !(k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_NUM_ROWS);
            need_mm |= (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX);
            need_fl |= (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST);
        }
        if (my_mask == 0) continue;

        const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
        const bool has_nulls = cd && cd->vbits != nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;
        const int variant = (has_nulls ? 1 : 0) | (col_type == BOWGPU_INT64 ? 2 : 0) | (need_mm ? 4 : 0) | (need_fl ? 8 : 0);

        __syncthreads();  // previous pass finished with sh.val / sh.vbits / sh.obits / sh.gap_*
        if (tid == 0) sh.gap_n = 0;
        if (cd && need_vals) {
            if (staged_slot != slot) {
                const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd->values);
                const bool vvec = (reinterpret_cast<uintptr_t>(vp) & 15) == 0;
#pragma unroll
                for (int j = 0; j < kChunks; j++) load_pair(vp, base + j * 512 + 2 * tid, n, vvec, va[j], vb[j]);
                if (wave == 0) load_pair(vp, base + kTile + 2 * tid, n, vvec, va[kChunks], vb[kChunks]);
            }
#pragma unroll
            for (int j = 0; j < kChunks; j++)
                *reinterpret_cast<ulonglong2 *>(&sh.val[j * 512 + 2 * tid]) = make_ulonglong2(va[j], vb[j]);
            if (wave == 0) *reinterpret_cast<ulonglong2 *>(&sh.val[kTile + 2 * tid]) = make_ulonglong2(va[kChunks], vb[kChunks]);
            staged_slot = -2;
            if (has_nulls && tid < kRows / 32) sh.vbits[tid] = load_vbits32(*cd, base + 32 * (int64_t)tid);
        }
        if (any_nullable)
            for (int i = tid; i < kMaxNullable * kSpanWords; i += kBlock) (&sh.obits[0][0])[i] = 0u;
        __syncthreads();

        const uint64_t wid_first = nseg_owned > 0 ? seg_wid_of(0) : 0;
        const int64_t slot_first = (int64_t)(wid_first - (uint64_t)p.wid_base);
        const int64_t span0 = slot_first & ~(int64_t)31;

        for (int q = tid; q < nseg_owned; q += kBlock) {
            const int r0 = sh.seg_row[q];
            const uint64_t wid = seg_wid_of(q);
            int r1;
            uint64_t next_wid;
            if (q + 1 < nseg_total) {
                r1 = sh.seg_row[q + 1];
                next_wid = seg_wid_of(q + 1);
            } else if (reaches_end) {
                r1 = nloc;
                next_wid = (uint64_t)wid_end;
            } else {
                // rows run past the halo: hand the window (all its column passes) to the cooperative path
                if (first_pass) {
                    const unsigned idx = atomicAdd(&p.status[1], 1u);
                    if ((int64_t)idx < p.long_cap) {
                        p.long_list[2 * idx] = (int64_t)wid;
                        p.long_list[2 * idx + 1] = base + r0;
                    } else {
                        atomicOr(&p.status[2], 1u);
                    }
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
            const bool big_gap = gap > kGapInline;
            if (big_gap) {
                const int gi = atomicAdd(&sh.gap_n, 1);
                if (gi < kGapList) sh.gap_q[gi] = q;
            }
            if ((uint64_t)oslot < (uint64_t)p.W) {
                int nb = 0;
                for (unsigned m = my_mask; m; m &= m - 1) {
                    const AggDesc &a = p.aggs[__ffs(m) - 1];
                    Val v = finish_val(reduce_val(a.kind, st, nrows, win_start, p.interval, col_type == BOWGPU_INT64), a);
                    reinterpret_cast<uint64_t *>(a.out_values)[oslot] = v.bits;
                    if (a.out_valid) {
                        if (v.valid) {
                            const int64_t lb = oslot - span0;
                            if (lb >= 0 && lb < kSpanBits) atomicOr(&sh.obits[nb][lb >> 5], 1u << (lb & 31));
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
        __syncthreads();

        // ---- big gaps (sparse data): the whole workgroup writes the empty windows, coalesced
        const int n_gaps = sh.gap_n;
        if (n_gaps > 0) {
            const bool overflow = n_gaps > kGapList;
            const int n_iter = overflow ? nseg_owned : n_gaps;
            for (int gi = 0; gi < n_iter; gi++) {
                const int q = overflow ? gi : sh.gap_q[gi];
                if (q + 1 >= nseg_total && !reaches_end) continue;
                const uint64_t wid = seg_wid_of(q);
                const uint64_t next_wid = (q + 1 < nseg_total) ? seg_wid_of(q + 1) : (uint64_t)wid_end;
                const int64_t gap = (int64_t)(next_wid - wid) - 1;
                if (gap <= kGapInline) continue;
                const int64_t oslot = (int64_t)(wid - (uint64_t)p.wid_base);
                const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
                Stats e;
                stats_init(e);
                for (int64_t gk = 1 + tid; gk <= gap; gk += kBlock) {
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
                if (ql + 1 < nseg_total) slot_end = (int64_t)(seg_wid_of(ql + 1) - (uint64_t)p.wid_base);
                else if (reaches_end) slot_end = p.W;
                else slot_end = (int64_t)(seg_wid_of(ql) - (uint64_t)p.wid_base);  // long window: not ours
            }
            if (slot_end > p.W) slot_end = p.W;
            int64_t lim = slot_end - span0;
            if (lim > kSpanBits) lim = kSpanBits;
            const int nwords = (int)((lim + 31) >> 5);
            int nb = 0;
            for (unsigned m = my_mask; m; m &= m - 1) {
                const AggDesc &a = p.aggs[__ffs(m) - 1];
                if (!a.out_valid) continue;
                for (int w = tid; w < nwords; w += kBlock) {
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
}

int launch_rolling_fast(Ctx *c, const AggParams &p) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTile - 1) / kTile;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    hipLaunchKernelGGL(rolling_fast_kernel, dim3((unsigned)grid), dim3(kBlock), 0, c->stream, p, ntiles, per_xcd);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
