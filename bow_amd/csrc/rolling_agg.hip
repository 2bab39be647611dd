// rolling_agg.hip — the hot path: Rolling.Aggregate over an int64 interval column.
//
// Replaces, in one pass, reference rolling/rolling.go:177-239 (intervalRolling.Next: the
// sequential interval-bucketing scan), rolling/aggregation.go:190-238 (aggregateWindows: one
// full scan per aggregator) and the reducer closures of rolling/aggregation/*.go.
//
// Data-parallel restatement: for an ascending, non-null interval column the
// scan is  wid(i) = (ts[i] - s0) div interval ; window k = the run of rows with wid == k.
// Row i is a *head* when wid(i) != wid(i-1).  A workgroup owns the windows whose head lies in
// its TILE of rows and walks each of them with ONE lane, in row order, so Sum / Mean /
// Integral are accumulated in exactly the reference's left-to-right order (bit-exact).
//
// Per workgroup (256 threads = 4 wavefronts of 64):
//   1. coalesced 16-B/lane loads of ts (+ the first value column) for TILE+HALO rows,
//   2. wid by an exact multiply-high division (no 64-bit divide in the loop),
//      head flags via ballot / mbcnt, compaction of the heads into an LDS segment list,
//   3. per value column: values (+ validity words) staged in LDS, lane q walks segment q,
//   4. outputs: 8-B coalesced stores (lane q -> window slot wid(q)); validity bits are
//      assembled in an LDS bitmap and flushed as whole words (atomicOr only on the two
//      boundary words shared with the neighbouring tiles).
// Windows whose rows run past TILE+HALO are queued for long_windows.hip (cooperative,
// fixed-shape tree order: Sum/Mean/Integral within 1e-12 relative; the rest bit-exact).
//
// HBM-bound: algorithmic bytes = 8 (ts) + 8 per value column (+1/8 per nullable column) per row.
// Nothing here is a contraction, so no MFMA.

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 2048;                // rows owned by a workgroup
constexpr int kHalo = 128;                 // look-ahead rows (one 16-B/lane load of wave 0)
constexpr int kRows = kTile + kHalo;
constexpr int kChunks = kTile / (2 * kBlock);  // 4 chunks of 512 rows, 2 rows per lane each
constexpr int kCnt = kChunks * 4 + 1;      // (chunk, wave) head counters + halo
constexpr int kSpanBits = 2048 + 64;       // output-validity bits assembled in LDS per tile
constexpr int kSpanWords = kSpanBits / 32;
constexpr int kMaxNullable = 8;            // nullable reducers per column pass
constexpr int kGapInline = 4;
constexpr int kGapList = 64;

struct TileShared {
    uint64_t val[kRows];
    uint32_t seg_wid[kRows + 2];   // window id - wid0 (kSat => recompute from ts)
    uint16_t seg_row[kRows + 2];   // local head row | (at-window-start << 15)
    uint32_t vbits[kRows / 32 + 2];
    uint32_t obits[kMaxNullable][kSpanWords];
    int cnt[kCnt + 1];
    int gap_q[kGapList];
    int gap_n;
    unsigned long long wid0;
};

template <bool kWithTs>
struct TileSharedTs : TileShared {
    double tsf[kWithTs ? kRows : 1];
};

}  // namespace

// ---------------------------------------------------------------- the tile kernel
template <bool kWithTs>
__global__ __launch_bounds__(kBlock) void rolling_agg_kernel(const AggParams p, const int64_t ntiles,
                                                             const int64_t tiles_per_xcd) {
    __shared__ TileSharedTs<kWithTs> sh;

    // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs, so give every
    // XCD a contiguous run of tiles (a tile's halo is its right neighbour's first rows: same L2).
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);
    if (tile >= ntiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t base = tile * kTile;
    const int64_t n = p.n;
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    const bool ts_vec = (reinterpret_cast<uintptr_t>(ts) & 15) == 0;

    // ---- 1. loads: ts for 4 chunks (+halo), first value column right behind them
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

    // previous row's ts for lane 0 of each wave (one extra 8-B load per wave per chunk; L2/L1 hit)
    // ---- 2. window ids, head flags, compaction
    unsigned long long mask_a[kChunks + 1], mask_b[kChunks + 1];
    uint64_t wid_a[kChunks + 1], wid_b[kChunks + 1];
    int flag_a[kChunks + 1], flag_b[kChunks + 1];  // at-window-start
    bool unsorted = false;
#pragma unroll
    for (int j = 0; j <= kChunks; j++) {
        const bool active = (j < kChunks) || (wave == 0);
        const int64_t l = (j < kChunks) ? (j * 512 + 2 * tid) : (kTile + 2 * tid);
        const int64_t g = base + l;
        const bool pa = active && g < n, pb = active && g + 1 < n;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        // previous row
        int64_t prev = (int64_t)__shfl_up((unsigned long long)tb[j], 1);
        if (lane == 0 && pa && g > 0) prev = (int64_t)ts[g - 1];
        // Rows below s0 exist only when Go's truncating division put s0 above a negative ts[0]
        // (rolling.go:96-99); the reference's scan skips-but-spans them (rolling.go:194-196), so
        // they ride in window 0.
        const uint64_t wa = tsa < p.s0 ? 0 : magic_div((uint64_t)tsa - (uint64_t)p.s0, p.magic);
        const uint64_t wb = tsb < p.s0 ? 0 : magic_div((uint64_t)tsb - (uint64_t)p.s0, p.magic);
        const int64_t wsa = p.s0 + (int64_t)(wa * (uint64_t)p.interval);
        const int64_t wsb = p.s0 + (int64_t)(wb * (uint64_t)p.interval);
        const bool ha = pa && (g == 0 || (prev < wsa && (prev >= p.s0 || wa != 0)));
        const bool hb = pb && (tsa < wsb && (tsa >= p.s0 || wb != 0));
        if (pa && g > 0 && prev > tsa) unsorted = true;
        if (pb && tsa > tsb) unsorted = true;
        wid_a[j] = wa; wid_b[j] = wb;
        flag_a[j] = (tsa == wsa); flag_b[j] = (tsb == wsb);
        mask_a[j] = __ballot(ha);
        mask_b[j] = __ballot(hb);
        if (kWithTs) {
            if (pa) sh.tsf[l] = (double)tsa;        // float64(ts): lossy above 2^53, as in the reference
            if (pb) sh.tsf[l + 1] = (double)tsb;
        }
        if (lane == 0 && active) {
            const int slot = (j < kChunks) ? (j * 4 + wave) : (kChunks * 4);
            sh.cnt[slot] = __popcll(mask_a[j]) + __popcll(mask_b[j]);
        }
    }
    if (tid == 0) { sh.wid0 = wid_a[0]; sh.gap_n = 0; }
    if (unsorted) atomicOr(&p.status[0], 1u);
    __syncthreads();

    const uint64_t wid0 = sh.wid0;
    int nseg_owned = 0, nseg_total = 0;
    {
        int run = 0;
#pragma unroll
        for (int k = 0; k < kCnt; k++) {
            if (k == kChunks * 4) nseg_owned = run;
            run += sh.cnt[k];
        }
        nseg_total = run;
    }
#pragma unroll
    for (int j = 0; j <= kChunks; j++) {
        const bool active = (j < kChunks) || (wave == 0);
        if (!active) continue;
        const int slot = (j < kChunks) ? (j * 4 + wave) : (kChunks * 4);
        int pos = 0;
        for (int k = 0; k < slot; k++) pos += sh.cnt[k];
        const unsigned lo_a = (unsigned)mask_a[j], hi_a = (unsigned)(mask_a[j] >> 32);
        const unsigned lo_b = (unsigned)mask_b[j], hi_b = (unsigned)(mask_b[j] >> 32);
        pos += __builtin_amdgcn_mbcnt_hi(hi_a, __builtin_amdgcn_mbcnt_lo(lo_a, 0));
        pos += __builtin_amdgcn_mbcnt_hi(hi_b, __builtin_amdgcn_mbcnt_lo(lo_b, 0));
        const int l = (j < kChunks) ? (j * 512 + 2 * tid) : (kTile + 2 * tid);
        const bool ha = (mask_a[j] >> lane) & 1, hb = (mask_b[j] >> lane) & 1;
        if (ha) {
            const uint64_t d = wid_a[j] - wid0;
            sh.seg_row[pos] = (uint16_t)(l | (flag_a[j] << 15));
            sh.seg_wid[pos] = d >= kSat ? kSat : (uint32_t)d;
            pos++;
        }
        if (hb) {
            const uint64_t d = wid_b[j] - wid0;
            sh.seg_row[pos] = (uint16_t)((l + 1) | (flag_b[j] << 15));
            sh.seg_wid[pos] = d >= kSat ? kSat : (uint32_t)d;
        }
    }
    // (seg arrays are read after the staging barrier below)

    const int64_t rows_here = (n - base) < kRows ? (n - base) : kRows;  // local rows that exist
    const bool reaches_end = base + kRows >= n;
    const int64_t wid_end = p.wid_base + p.W;  // one past the last addressable window

    auto seg_wid_of = [&](int q) -> uint64_t {
        const uint32_t d = sh.seg_wid[q];
        if (d != kSat) return wid0 + d;
        const int64_t t = p.ts[base + (sh.seg_row[q] & 0x7FFF)];
        return magic_div((uint64_t)t - (uint64_t)p.s0, p.magic);
    };

    // ---- 3./4. one pass per column slot (-1 = reducers that read no column values)
    bool first_pass = true;
    for (int slot = -1; slot < p.ncols; slot++) {
        // which aggregators belong to this pass? (bitmask over p.aggs: no private arrays)
        unsigned my_mask = 0;
        bool any_nullable = false, any_incl = false, need_vals = false;
        for (int a = 0; a < p.naggs; a++) {
            if (p.aggs[a].slot != slot) continue;
            my_mask |= 1u << a;
            any_nullable |= (p.aggs[a].out_valid != nullptr);
            any_incl |= kind_needs_inclusive(p.aggs[a].kind);
            const int k = p.aggs[a].kind;
            need_vals |= !(k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_NUM_ROWS);
        }
        if (my_mask == 0) continue;

        const ColDesc *cd = slot >= 0 ? &p.cols[slot] : nullptr;
        const bool has_nulls = cd && cd->vbits != nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;

        __syncthreads();  // previous pass finished reading sh.val / sh.vbits / sh.obits / sh.gap_*
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
            for (int j = 0; j < kChunks; j++) {
                sh.val[j * 512 + 2 * tid] = va[j];
                sh.val[j * 512 + 2 * tid + 1] = vb[j];
            }
            if (wave == 0) { sh.val[kTile + 2 * tid] = va[kChunks]; sh.val[kTile + 2 * tid + 1] = vb[kChunks]; }
            staged_slot = -2;
            if (has_nulls && tid < kRows / 32) sh.vbits[tid] = load_vbits32(*cd, base + 32 * (int64_t)tid);
        }
        if (any_nullable)
            for (int i = tid; i < kMaxNullable * kSpanWords; i += kBlock) (&sh.obits[0][0])[i] = 0u;
        __syncthreads();

        // window slot range this tile flushes through LDS: [span0, span0 + kSpanBits)
        const uint64_t wid_first = nseg_owned > 0 ? seg_wid_of(0) : 0;
        const int64_t slot_first = (int64_t)(wid_first - (uint64_t)p.wid_base);
        const int64_t span0 = slot_first & ~(int64_t)31;

        for (int q = tid; q < nseg_owned; q += kBlock) {
            const int r0 = sh.seg_row[q] & 0x7FFF;
            const uint64_t wid = seg_wid_of(q);
            int r1;
            uint64_t next_wid;
            bool next_at_start = false;
            bool complete = true;
            if (q + 1 < nseg_total) {
                r1 = sh.seg_row[q + 1] & 0x7FFF;
                next_at_start = (sh.seg_row[q + 1] >> 15) & 1;
                next_wid = seg_wid_of(q + 1);
            } else if (reaches_end) {
                r1 = (int)rows_here;
                next_wid = (uint64_t)wid_end;
            } else {
                complete = false; r1 = r0; next_wid = wid + 1;
            }
            if (!complete) {
                // rows run past the halo: hand the window (all its column passes) to the cooperative path
                if (!first_pass) continue;
                push_long_window(p.status, p.long_list, p.long_cap, tile, wid, base + r0);
                continue;
            }
            const bool incl_row = p.inclusive && (q + 1 < nseg_total) && next_at_start && next_wid == wid + 1;
            // window 0 made only of rows below s0 is an empty slice in the reference (lastRowIndex stays -1)
            const bool dead = p.pre_rows && tile == 0 && q == 0 && !(p.ts[base + r1 - 1] >= p.s0 || incl_row);

            // ---- sequential walk of rows [r0, r1) (+ the inclusive row)
            Stats st;
            stats_init(st);
            Stats st_incl;  // state including the inclusive row, for reducers that need it
            if (need_vals && !dead) {
                for (int r = r0; r < r1; r++) {
                    const bool ok = !has_nulls || ((sh.vbits[r >> 5] >> (r & 31)) & 1u);
                    if (!ok) continue;
                    const uint64_t raw = sh.val[r];
                    const double x = bits_to_f64(raw, col_type);
                    stats_value<false>(st, x, raw);
                    if (kWithTs) stats_point(st, sh.tsf[r], x);
                }
                st_incl = st;
                if (any_incl && incl_row) {
                    const int r = r1;
                    const bool ok = !has_nulls || ((sh.vbits[r >> 5] >> (r & 31)) & 1u);
                    if (ok) {
                        const uint64_t raw = sh.val[r];
                        const double x = bits_to_f64(raw, col_type);
                        stats_value<false>(st_incl, x, raw);
                        if (kWithTs) stats_point(st_incl, sh.tsf[r], x);
                    }
                }
            } else {
                st_incl = st;
            }
            const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
            const int64_t oslot = (int64_t)(wid - (uint64_t)p.wid_base);
            const int64_t nrows = dead ? 0 : r1 - r0;
            const int64_t gap = (int64_t)(next_wid - wid) - 1;
            const bool big_gap = gap > kGapInline;
            if (big_gap) {
                const int gi = atomicAdd(&sh.gap_n, 1);
                if (gi < kGapList) sh.gap_q[gi] = q;
            }
            if (oslot >= 0 && oslot < p.W) {
                int nb = 0;
                for (unsigned m = my_mask; m; m &= m - 1) {
                    const AggDesc &a = p.aggs[__ffs(m) - 1];
                    const bool inc = kind_needs_inclusive(a.kind);
                    Val v = reduce_val(a.kind, inc ? st_incl : st, inc ? nrows + (incl_row ? 1 : 0) : nrows,
                                       win_start, p.interval, col_type == BOWGPU_INT64);
                    v = finish_val(v, a);
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
                // small gaps: the empty windows right after this one (A.9 "Empty slice" column)
                if (gap > 0 && (!big_gap)) {
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

        // ---- flush the validity bits assembled in LDS
        if (any_nullable && nseg_owned > 0) {
            // bits [slot_first, slot_end) are this tile's to define (later ones belong to other
            // tiles or to the long-window path); whole words inside are stored, the rest OR-ed.
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

// ---------------------------------------------------------------- small helpers
__global__ void fix_tail_bits_kernel(uint8_t *bitmap, int64_t nbits) {
    // Arrow/bow leave the padding bits of the last byte clear (bowbuffer.go:25, bitutil.SetBit)
    if (threadIdx.x == 0 && blockIdx.x == 0 && (nbits & 7)) bitmap[nbits >> 3] &= (uint8_t)((1u << (nbits & 7)) - 1u);
}

__global__ __launch_bounds__(256) void popcount_kernel(const uint32_t *words, int64_t bit0, int64_t nbits,
                                                       unsigned long long *out) {
    // counts set bits in [bit0, bit0+nbits)
    const int64_t w0 = bit0 >> 5, w1 = (bit0 + nbits + 31) >> 5;
    unsigned long long acc = 0;
    for (int64_t w = w0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < w1; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = words[w];
        const int64_t lo = w << 5;
        if (lo < bit0) x &= ~0u << (bit0 - lo);
        if (lo + 32 > bit0 + nbits) { const int keep = (int)(bit0 + nbits - lo); x &= keep >= 32 ? ~0u : ((1u << keep) - 1u); }
        acc += __popc(x);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(out, t);
    }
}

int launch_rolling_aggregate(Ctx *c, const AggParams &p) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTile - 1) / kTile;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    bool with_ts = false;
    for (int a = 0; a < p.naggs; a++) with_ts |= (p.aggs[a].kind >= BOWGPU_AGG_INTEGRAL_STEP && p.aggs[a].kind <= BOWGPU_AGG_WAVG_LINEAR);
    if (with_ts) hipLaunchKernelGGL(rolling_agg_kernel<true>, dim3((unsigned)grid), dim3(kBlock), 0, c->stream, p, ntiles, per_xcd);
    else hipLaunchKernelGGL(rolling_agg_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, c->stream, p, ntiles, per_xcd);
    BG_HIP(hipGetLastError());
    return 0;
}

// ---- the output bitmaps of one call, all in ONE launch each way (per-output memset + tail fix + popcount + copy-back used to be
// eleven stream operations around a kernel of a few microseconds: DESIGN.md section 6)
// preset: working bitmaps to all-ones / all-zeros with the padding bits of the last byte clear (bowbuffer.go:25), status words and
// the valid counters to zero.  finish: valid bits counted (nullable outputs) and the bitmap's ceil(W/8) bytes copied into the
// caller's device buffer (any alignment; the working copy is word-aligned because the kernels update validity as 32-bit words).
__global__ __launch_bounds__(256) void preset_bitmaps_kernel(const BitmapBatch b) {
    const int a = blockIdx.y;
    if (a == 0 && blockIdx.x == 0) {
        // (status[6], thread 6: a caller-supplied plan - bowgpu_rolling_aggregate_planned - against the column it is used on: the two
        // timestamps it was made from are the column's first and last row, or the call fails instead of taking wrong routes)
        for (int i = threadIdx.x; i < b.status_words; i += blockDim.x)
            b.status[i] = (i == 6 && b.check_ts && b.check_n > 0 && (b.check_ts[0] != b.check_first || b.check_ts[b.check_n - 1] != b.check_last)) ? 1u : 0u;
        if (threadIdx.x < kMaxAggs) b.counts[threadIdx.x] = 0ull;
    }
    if (a >= b.n) return;
    const int64_t nwords = (b.nbits + 31) >> 5;
    const uint32_t fill = b.ones[a] ? 0xFFFFFFFFu : 0u;
    uint32_t *w = b.work[a];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = fill;
        if (i == nwords - 1 && (b.nbits & 31)) x &= (1u << (b.nbits & 31)) - 1u;
        w[i] = x;
    }
}
__global__ __launch_bounds__(256) void finish_bitmaps_kernel(const BitmapBatch b) {
    const int a = blockIdx.y;
    if (a >= b.n) return;
    const int64_t nwords = (b.nbits + 31) >> 5, nbytes = (b.nbits + 7) >> 3;
    const uint32_t *w = b.work[a];
    uint8_t *u = b.user[a];
    const bool aligned = (reinterpret_cast<uintptr_t>(u) & 3) == 0;
    unsigned long long acc = 0;
    // four words per thread and trip where the buffers allow 16-byte accesses (a 4-byte load per thread leaves too few bytes in
    // flight: 1e8 bits took 38 us), the words behind the last full group and the partial last word one by one below
    const bool vec = (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (!u || (reinterpret_cast<uintptr_t>(u) & 15) == 0);
    const int64_t nvec = vec ? ((b.nbits >> 5) >> 2) : 0;   // groups of four FULL words
    // (four independent 16-byte loads in flight per thread: one per trip left this kernel latency-bound - 35 us for two bitmaps of 1.09e8
    // bits, 1.5 TB/s; the Interpolate call and every Aggregate call end with it)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; g + 3 * stride < nvec; g += 4 * stride) {
        const uint4 *q = reinterpret_cast<const uint4 *>(w) + g;
        const uint4 x0 = q[0], x1 = q[stride], x2 = q[2 * stride], x3 = q[3 * stride];
        if (b.count[a]) acc += __popc(x0.x) + __popc(x0.y) + __popc(x0.z) + __popc(x0.w) + __popc(x1.x) + __popc(x1.y) + __popc(x1.z) + __popc(x1.w) +
                               __popc(x2.x) + __popc(x2.y) + __popc(x2.z) + __popc(x2.w) + __popc(x3.x) + __popc(x3.y) + __popc(x3.z) + __popc(x3.w);
        if (u) {
            uint4 *d = reinterpret_cast<uint4 *>(u) + g;
            d[0] = x0; d[stride] = x1; d[2 * stride] = x2; d[3 * stride] = x3;
        }
    }
    for (; g < nvec; g += stride) {
        const uint4 x = reinterpret_cast<const uint4 *>(w)[g];
        if (b.count[a]) acc += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
        if (u) reinterpret_cast<uint4 *>(u)[g] = x;
    }
    for (int64_t i = 4 * nvec + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = w[i];
        if (i == nwords - 1 && (b.nbits & 31)) x &= (1u << (b.nbits & 31)) - 1u;
        if (b.count[a]) acc += __popc(x);
        if (u) {
            if (aligned && 4 * i + 4 <= nbytes) reinterpret_cast<uint32_t *>(u)[i] = x;
            else for (int k = 0; k < 4 && 4 * i + k < nbytes; k++) u[4 * i + k] = (uint8_t)(x >> (8 * k));
        }
    }
    if (b.count[a]) {   // one atomic per WORKGROUP: atomics on one address cost ~10 ns each (2048 of them were 20 of this kernel's 34 us)
        __shared__ unsigned long long wave_sum[4];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
            if (b.host_block) reinterpret_cast<unsigned long long *>(b.host_block + 1024)[a] = t;   // (the only workgroup of this bitmap)
            else if (t) atomicAdd(&b.counts[a], t);
        }
    } else if (b.host_block && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(b.host_block + 1024)[a] = 0;
    // the status words of the launches in front of this one, straight into the host's block: no copy command behind the call's last launch
    if (b.host_block && a == 0 && blockIdx.x == 0)
        for (int t = threadIdx.x; t < b.status_words; t += blockDim.x) reinterpret_cast<uint32_t *>(b.host_block)[t] = b.status[t];
}

static unsigned bitmap_grid(int64_t nbits) {
    const int64_t nwords = (nbits + 31) >> 5;
    int64_t g = (nwords + 255) / 256;
    return (unsigned)(g < 1 ? 1 : g > 1024 ? 1024 : g);
}
int launch_preset_bitmaps(Ctx *c, const BitmapBatch &b) {
    hipLaunchKernelGGL(preset_bitmaps_kernel, dim3(bitmap_grid(b.nbits), b.n > 0 ? b.n : 1), dim3(256), 0, c->stream, b);
    BG_HIP(hipGetLastError());
    return 0;
}
int launch_finish_bitmaps(Ctx *c, const BitmapBatch &b) {
    if (b.n <= 0 || b.nbits <= 0) return 0;
    hipLaunchKernelGGL(finish_bitmaps_kernel, dim3(b.host_block ? 1u : bitmap_grid(b.nbits), b.n), dim3(256), 0, c->stream, b);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_fix_tail_bits(Ctx *c, uint8_t *bitmap, int64_t nbits) {
    if ((nbits & 7) == 0) return 0;
    hipLaunchKernelGGL(fix_tail_bits_kernel, dim3(1), dim3(64), 0, c->stream, bitmap, nbits);
    BG_HIP(hipGetLastError());
    return 0;
}

// two rows of a device-resident column to the host's registered block: what a plan needs of the interval column (its first and last
// timestamp).  One launch whose stores go over the link, then the caller's synchronize - two copy commands (a blit kernel each on
// this runtime) cost ~5 us more per call.
__global__ void fetch_two_kernel(const int64_t *col, int64_t i0, int64_t i1, int64_t *host_out) {
    if (threadIdx.x == 0) host_out[0] = col[i0];
    if (threadIdx.x == 1) host_out[1] = col[i1];
}
int launch_fetch_two(Ctx *c, const int64_t *col, int64_t i0, int64_t i1, int64_t *host_out) {
    hipLaunchKernelGGL(fetch_two_kernel, dim3(1), dim3(64), 0, c->stream, col, i0, i1, host_out);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_popcount(Ctx *c, const uint32_t *words, int64_t bit0, int64_t nbits, uint64_t *d_count) {
    BG_HIP(hipMemsetAsync(d_count, 0, 8, c->stream));
    if (nbits <= 0) return 0;
    const int64_t nwords = (nbits + 63) / 32;
    int64_t grid = (nwords + 255) / 256;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(popcount_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, words, bit0, nbits,
                       reinterpret_cast<unsigned long long *>(d_count));
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
