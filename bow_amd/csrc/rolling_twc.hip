// rolling_twc.hip — the wave-tile kernel for NULLABLE columns under the time-weighted reducers (IntegralStep, IntegralTrapezoid,
// WeightedAverageStep, WeightedAverageLinear: reference rolling/aggregation/integral.go:8-69, weightedmean.go:8-34) and next to them
// every value reducer of rolling_simple.hip; inclusive windows where a reducer asks for them (rolling.go:201-209).
//
// integral.go walks the BOTH-VALID points of a window (bowgetters.go:299-311): nulls do not exist for it.  rolling_tw.hip keeps the
// tile's rows where they are and lets every row look for "the valid point before me" in the validity words - a count-leading-zeros
// search, two gathers from LDS and three window-id divisions per row pair, then two walks over NaN / +0.0 filled rows: 1322 vector
// instructions per tile for one integral (dense: 865), 0.26 - 0.32 of the HBM peak for all four at 32 - 96 rows per window
// (profiles/r04_pmc_mid_windows.txt, r05_stdout_midw_sweep.txt).  Here the nulls are taken OUT first:
//   1. COMPACTION.  A row's rank among the tile's valid rows = valid rows in front of its validity word (a wave scan over the 20
//      words' popcounts) + set bits below it in the word; every valid row writes its float64 value and its 32-bit time to slot
//      `rank` of two dense LDS arrays.  A window [r0, r1) is the slot range [rank(r0), rank(r1)).
//   2. In slot space "the point before me" is the slot before me: the term of slot k against slot k - 1 - v(k-1) * (t(k) - t(k-1)) and
//      (v(k-1) + v(k)) / 2 * (t(k) - t(k-1)) - is computed for EVERY slot by the dense neighbour logic (one LDS read of the slot in front
//      of a lane's pair, no search, no window test), in place; a window's lane adds the terms of slots ka + 1 .. kb - 1 in order - the
//      chain of additions integral.go:22-31 / :48-62 performs - then v(kb-1) * (float64(LastValue) - t(kb-1)) for the step integral
//      (:49-55), or term(kb) when the next window's first row sits on this window's end and is valid (the inclusive row) for the trapezoid.
//   3. The value reducers walk the same slots: no null rows left, so ONE walk serves sums and extrema (rolling_simple.hip needs two,
//      or a predicated one), over 70 % of the rows at 30 % nulls.
// Bit-exact like the forms it replaces (one lane adds one window's terms in row order).  32-bit times (every |ts| < 2^53, rows within
// 2^32 of slot 0): the host sends everything else - and calls of short windows - to rolling_tw.hip, and a tile with more window heads
// than this kernel's list holds raises status[4]: the call is redone there.
#include <stddef.h>

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kWave = 64;
constexpr int kTileC = 512;
constexpr uint32_t kRowMaskC = 0x3FFu, kStartBitC = 0x400u;   // a head entry: local row | sits on its window's start << 10 | (wid - wid of the tile's first row) << 11

// kHalo: the look-ahead behind a tile's 512 rows.  128 rows: windows of up to 128 rows always lie inside one tile, 8 KB of LDS (20
// wavefronts per CU), 92 heads (windows of 7 rows and more on average).  256 rows: the form for calls whose windows average 129 .. 224
// rows, which no 128-row look-ahead holds and the streaming form serves at 0.32 of the peak when the column has nulls: 10 KB (16 per CU).
template <int kHalo> struct TwcCap { static constexpr int value = kHalo == 128 ? 92 : 200; };
template <int kHalo>
struct TwcShared {
    static constexpr int kRows = kTileC + kHalo, kWords = kRows / 32;
    uint64_t val[kRows];            // the column's valid values, compacted; then its step terms; then its trapezoid terms
    uint32_t ctx[kRows];            // their times: 32-bit offsets from the start of slot 0
    uint32_t vbits[kWords + 2];     // validity words of the value column for this tile (rows beyond the tile's last row: 0)
    uint16_t wpre[kWords + 2];      // valid rows in front of each word; [kWords] = all of the tile's
    uint32_t seg[TwcCap<kHalo>::value + 2];   // heads in row order
};
static_assert(sizeof(TwcShared<128>) <= 8192, "LDS of the compacting form: 8 KB (20 wavefronts per CU)");
static_assert(sizeof(TwcShared<256>) <= 10240, "LDS of the compacting form with the long look-ahead: 10 KB (16 wavefronts per CU)");

__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
typedef unsigned long long u64x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 load16_nt(const ulonglong2 *q) {
    const u64x2_c v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_c *>(q));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// inclusive wave scan (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, row_bcast 15 / 31 across): six DPP moves + adds, no LDS
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

}  // namespace

// kBoth: the call has step AND trapezoid integrals: the terms of the second kind wait in registers while the first kind is walked
// kMulti: more than one value column.  The single-column form compacts its column right in the flag pass, where a row's time and
// value are in registers anyway - nothing of the tile then lives in registers behind it (no times kept for later columns, no next
// column in flight): 96 instead of 134 registers, a fifth more wavefronts per CU.
template <bool kBoth, bool kMulti, int kHalo>
__global__ __launch_bounds__(kWave, (kMulti && kBoth) ? 3 : (kMulti || kHalo != 128) ? 4 : 5) void rolling_twc_kernel(const SimpleParams p, const int64_t ntiles, const int64_t tiles_per_xcd) {
    constexpr int kRowsC = kTileC + kHalo, kChunksC = kRowsC / 128, kWordsC = kRowsC / 32, kCapC = TwcCap<kHalo>::value;
    __shared__ TwcShared<kHalo> sh;
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs (look-ahead rows hit the same L2)
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int64_t base = tile * kTileC;
    const int64_t n = p.n;
    const bool interior = base + kRowsC <= n;
    const int nloc = interior ? kRowsC : (int)(n - base);

    // ---- loads: ts, then the first value column right behind it and its validity words (rolling_tw.hip: one block for the usual tile)
    uint64_t ta[kChunksC], tb[kChunksC], va[kChunksC], vb[kChunksC];
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    auto load_col = [&](const uint64_t *__restrict__ src, uint64_t (&a)[kChunksC], uint64_t (&bb)[kChunksC], bool aligned) {
        if (interior && aligned) {
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(src + base) + lane;
#pragma unroll
            for (int j = 0; j < kChunksC; j++) {
                const ulonglong2 x = (j >= kHalo / 128 && j < kTileC / 128) ? load16_nt(q + j * 64) : q[j * 64];
                a[j] = x.x; bb[j] = x.y;
            }
        } else if (interior) {
            const uint64_t *q = src + base + 2 * lane;
#pragma unroll
            for (int j = 0; j < kChunksC; j++) { a[j] = q[j * 128]; bb[j] = q[j * 128 + 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksC; j++) load_pair(src, base + j * 128 + 2 * lane, n, aligned, a[j], bb[j]);
        }
    };
    // validity word `lane` of one column's 640 rows, any bit offset (Arrow slices); rows at or beyond the tile's last row read as null
    auto load_vword = [&](int c) -> uint32_t {
        uint32_t word = 0u;
        if (lane < kWordsC && 32 * lane < nloc) {
            word = 0xFFFFFFFFu;
            if (p.vbits[c] != nullptr) {
                const int64_t bit = p.vbit0[c] + base + 32 * (int64_t)lane;
                const int64_t wi = bit >> 5;
                const int shb = (int)(bit & 31);
                const uint32_t lo = wi < p.vwords[c] ? p.vbits[c][wi] : 0u;
                const uint32_t hi = (shb != 0 && wi + 1 < p.vwords[c]) ? p.vbits[c][wi + 1] : 0u;
                word = shb ? ((lo >> shb) | (hi << (32 - shb))) : lo;
            }
            if (32 * lane + 32 > nloc) word &= (1u << (nloc - 32 * lane)) - 1u;
        }
        return word;
    };
    if (interior && !(p.unaligned_mask & 0x80000001u)) {
        const ulonglong2 *qt = reinterpret_cast<const ulonglong2 *>(ts + base) + lane;
        const ulonglong2 *qv = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const uint64_t *>(p.values[0]) + base) + lane;
#pragma unroll
        for (int j = 0; j < kChunksC; j++) {
            const ulonglong2 x = (j >= kHalo / 128 && j < kTileC / 128) ? load16_nt(qt + j * 64) : qt[j * 64];
            ta[j] = x.x; tb[j] = x.y;
        }
#pragma unroll
        for (int j = 0; j < kChunksC; j++) {
            const ulonglong2 x = (j >= kHalo / 128 && j < kTileC / 128) ? load16_nt(qv + j * 64) : qv[j * 64];
            va[j] = x.x; vb[j] = x.y;
        }
    } else {
        load_col(ts, ta, tb, !(p.unaligned_mask >> 31));
        load_col(reinterpret_cast<const uint64_t *>(p.values[0]), va, vb, !(p.unaligned_mask & 1u));
    }
    uint32_t vword = load_vword(0);
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    const int64_t ws0 = p.s0;
    const uint32_t s0_lo = (uint32_t)ws0;
    const uint32_t ik = (uint32_t)p.interval;
    bool unsorted = false, sat = false;
    const int64_t ts_first = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta[0] >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta[0]));

    const bool need_step = p.need & kNeedStep, need_trap = p.need & kNeedTrap, need_mm = p.need & kNeedMinMax, need_sum = p.need & kNeedSum,
               need_fl = p.need & kNeedFirstLast;
    const bool need_vals = need_mm || need_sum || need_fl;
    // ---- a column's validity words and the valid rows in front of each word -> LDS
    auto stage_validity = [&](uint32_t vw) {
        const uint32_t pc = (uint32_t)__popc(vw);            // (lanes beyond the tile's words hold 0)
        const uint32_t incl = wave_scan_u32(pc);
        if (lane < kWordsC + 2) sh.vbits[lane] = vw;
        if (lane <= kWordsC) sh.wpre[lane] = (uint16_t)(incl - pc);   // lane kWordsC: every valid row of the tile
    };
    // ---- compaction of one chunk: every valid row's float64 value (Int64 columns convert here: bowgetters.go:224-229) and time to its slot
    auto compact_chunk = [&](int j, uint64_t xa, uint64_t xb, uint32_t ta32, uint32_t tb32, bool cint_, bool &snan_) {
        const int l = j * 128 + 2 * lane;
        const uint32_t word = sh.vbits[l >> 5];
        const int shb = l & 31;
        const uint32_t two = (word >> shb) & 3u;
        const int slot = (int)sh.wpre[l >> 5] + __popc(word & ((1u << shb) - 1u));
        if (cint_) {
            xa = (uint64_t)__double_as_longlong((double)(int64_t)xa);
            xb = (uint64_t)__double_as_longlong((double)(int64_t)xb);
        }
        if (need_mm && !cint_) snan_ = snan_ || ((two & 1u) && is_snan(xa)) || ((two & 2u) && is_snan(xb));
        if (two & 1u) { sh.val[slot] = xa; sh.ctx[slot] = ta32; }
        if (two & 2u) { const int s2 = slot + (int)(two & 1u); sh.val[s2] = xb; sh.ctx[s2] = tb32; }
    };
    stage_validity(vword);
    lds_order();
    bool snan0 = false;
    const bool cint0 = p.col_is_int[0] != 0;

    // ---- window ids (32-bit, global), head flags, compaction of the heads with a running scalar count; the first column's valid
    // rows go to their slots right here; kMulti: the rows' times stay in registers as 32-bit offsets for the later columns
    const uint32_t w_first = mdiv32((uint32_t)ts_first - s0_lo, p.m32, p.sh1, p.sh2);
    uint32_t left_w = base == 0 ? 0xFFFFFFFEu : mdiv32((uint32_t)left0 - s0_lo, p.m32, p.sh1, p.sh2);
    int64_t left_ts = left0;
    uint32_t ra[kMulti ? kChunksC : 1], rb[kMulti ? kChunksC : 1];
    int nseg_total = 0, nseg_owned = 0;
#pragma unroll
    for (int j = 0; j < kChunksC; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
        const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
        const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
        unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
        const uint32_t ra_j = (uint32_t)tsa - s0_lo, rb_j = (uint32_t)tsb - s0_lo;
        if (kMulti) { ra[kMulti ? j : 0] = ra_j; rb[kMulti ? j : 0] = rb_j; }
        compact_chunk(j, va[j], vb[j], ra_j, rb_j, cint0, snan0);
        const uint32_t wa = mdiv32(ra_j, p.m32, p.sh1, p.sh2);
        const uint32_t wb = mdiv32(rb_j, p.m32, p.sh1, p.sh2);
        const uint32_t wprev = left32(wb, left_w);
        const bool ha = pa && (wa != wprev);
        const bool hb = pb && (wb != wa);
        const uint32_t la = wa - w_first, lb = wb - w_first;
        sat |= (ha && la >= (1u << 21)) || (hb && lb >= (1u << 21));
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        // the row sits exactly on its window's start: what makes it the inclusive row of the window before (only inclusive calls look)
        const uint32_t sa = (p.inclusive && ra_j == wa * ik) ? kStartBitC : 0u, sb = (p.inclusive && rb_j == wb * ik) ? kStartBitC : 0u;
        if (ha && pos < kCapC) sh.seg[pos] = (uint32_t)l | sa | (la << 11);
        pos += ha ? 1 : 0;
        if (hb && pos < kCapC) sh.seg[pos] = (uint32_t)(l + 1) | sb | (lb << 11);
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kTileC / 128 - 1) nseg_owned = nseg_total;   // (the heads inside the tile's own 512 rows)
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
    }
    if (__ballot(unsorted)) {  // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0) atomicOr(&p.status[0], 1u);
        return;
    }
    if (nseg_total > kCapC) sat = true;
    if (__ballot(sat)) {  // more heads than the list holds / ids too far apart: the host redoes the call with rolling_tw.hip
        if (lane == 0 && !__hip_atomic_load(&p.status[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[4], 1u);
        return;
    }

    const bool reaches_end = base + kRowsC >= n;
    const uint32_t W32 = (uint64_t)p.W > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)p.W;
    const double ws0_d = (double)ws0;
    // float64(LastValue) of window id w (integral.go:51) and the time of a point, absolute: float64(s0) + float64(offset) is exact
    // (every |ts| < 2^53: the host's condition for this kernel), so it equals the reference's single conversion
    auto last_value_d = [&](uint32_t w) -> double { return (double)(ws0 + (int64_t)(((uint64_t)w + 1ull) * (uint64_t)ik)); };

    // ---- the geometry of a window: its rows [r0, r1), its id and its successor's (column- and phase-independent: this lane's first
    // window is worked out once)
    auto geometry = [&](int q, int &r0, int &r1, uint32_t &wid, uint32_t &next_wid, uint32_t &e1) -> int {   // 0: a window; 1: it runs past the look-ahead
        const uint32_t e0 = sh.seg[q];
        e1 = sh.seg[q + 1];
        r0 = (int)(e0 & kRowMaskC);
        wid = w_first + (e0 >> 11);
        if (q + 1 < nseg_total) {
            r1 = (int)(e1 & kRowMaskC);
            next_wid = w_first + (e1 >> 11);
            return 0;
        }
        r1 = nloc;
        next_wid = W32;
        return reaches_end ? 0 : 1;
    };
    lds_order();   // the head list is complete
    int g_r0 = 0, g_r1 = 0, g_state = 2;
    uint32_t g_wid = 0, g_next = 0, g_e1 = 0;
    if (lane < nseg_owned) g_state = geometry(lane, g_r0, g_r1, g_wid, g_next, g_e1);
    // valid rows of the tile in front of row r (r <= the tile's rows): the slot of row r when it is valid
    auto rank_of = [&](int r) -> int {
        const int w = r >> 5;
        return (int)sh.wpre[w] + __popc(sh.vbits[w] & ((1u << (r & 31)) - 1u));
    };

    // ---- one pass per value column
    const int ncols = kMulti ? p.ncols : 1;
    for (int c = 0; c < ncols; c++) {
        const bool cint = p.col_is_int[c] != 0;
        const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(p.values[c]);
        bool snan = snan0;
        if (kMulti && c > 0) {
            lds_order();  // the previous pass is done with sh.val / sh.ctx / sh.vbits
            stage_validity(vword);
            lds_order();
            snan = false;
#pragma unroll
            for (int j = 0; j < kChunksC; j++) compact_chunk(j, va[j], vb[j], ra[kMulti ? j : 0], rb[kMulti ? j : 0], cint, snan);
        }
        // the next column's loads go out now: its registers are free (this column lives in LDS)
        if (kMulti && c + 1 < ncols) {
            load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
            vword = load_vword(c + 1);
        }
        lds_order();
        const int total = __builtin_amdgcn_readfirstlane((int)sh.wpre[kWordsC]);   // (the same in every lane: a scalar)
        const bool exact_mm = need_mm && __ballot(snan) != 0ull;
        // ---- this lane's first window in slot space: [ka, kb); its last point, which the step integral's closing term needs, is read
        // now - the term pass overwrites the values
        int g_ka = 0, g_kb = 0;
        double g_pv = 0.0;
        uint32_t g_pt = 0;
        if (g_state == 0) {
            g_ka = rank_of(g_r0);
            g_kb = g_r1 >= kRowsC ? total : rank_of(g_r1);
            if (g_kb > g_ka) { g_pv = __longlong_as_double((long long)sh.val[g_kb - 1]); g_pt = sh.ctx[g_kb - 1]; }
        }
        // ---- the term pass over the slots (the dense neighbour logic of rolling_tw.hip, without window tests: a window's lane never
        // adds the term of its first slot).  `first` - step terms when the call has them, else trapezoid terms - replaces the values
        // in place, chunk by chunk: the slot in front of a lane's pair is read before the chunk is written, the one in front of the NEXT
        // chunk saved first.  kBoth: the trapezoid terms wait in registers until the step terms have been walked.
        double keep_a[kBoth ? kChunksC : 1], keep_b[kBoth ? kChunksC : 1];
        auto term_pass = [&]() __attribute__((always_inline)) {
            double carry_x = 0.0;
            int tot_ = total;
            asm volatile("" : "+s"(tot_));
#pragma unroll
            for (int j = 0; j < kChunksC; j++) {
                if (j * 128 < tot_) {
                    const int k = j * 128 + 2 * lane;
                    const ulonglong2 xv = *reinterpret_cast<const ulonglong2 *>(&sh.val[k]);
                    const uint2 t2 = *reinterpret_cast<const uint2 *>(&sh.ctx[k]);
                    const double xa = __longlong_as_double((long long)xv.x), xb = __longlong_as_double((long long)xv.y);
                    const int kp = k > 0 ? k - 1 : 0;
                    const double xp = (lane == 0 && j > 0) ? carry_x : __longlong_as_double((long long)sh.val[kp]);
                    // float64(t1) - float64(t0) (integral.go:28, :54): the difference of the 32-bit offsets, converted - exact, and equal to the
                    // difference of the two float64 times, which are exact themselves
                    const double dta = (double)(uint32_t)(t2.x - sh.ctx[kp]), dtb = (double)(uint32_t)(t2.y - t2.x);
                    double sa = 0.0, sb = 0.0, qa = 0.0, qb = 0.0;
                    if (need_step) { sa = xp * dta; sb = xa * dtb; }
                    if (need_trap) { qa = (xp + xa) / 2 * dta; qb = (xa + xb) / 2 * dtb; }
                    if (j + 1 < kChunksC) carry_x = __longlong_as_double((long long)sh.val[j * 128 + 127]);
                    lds_order();   // every lane has read this chunk's values
                    const double oa = need_step ? sa : qa, ob = need_step ? sb : qb;
                    *reinterpret_cast<ulonglong2 *>(&sh.val[k]) = make_ulonglong2((uint64_t)__double_as_longlong(oa), (uint64_t)__double_as_longlong(ob));
                    if (kBoth) { keep_a[kBoth ? j : 0] = qa; keep_b[kBoth ? j : 0] = qb; }
                }
                __builtin_amdgcn_sched_barrier(0);   // (one chunk at a time: interleaved by the scheduler the five chunks cost more registers than there are)
            }
        };
        // ---- phases, each one walk over what sh.val holds: 0 the values (every output that is not an integral), 3 the step terms,
        // 4 the trapezoid terms
        const int phase1 = need_vals || !(need_step || need_trap) ? 0 : need_step ? 3 : 4;   // the first phase that runs: it also writes WindowStart / Count / NumRows
        for (int phase = 0; phase <= 4; phase++) {
            if (phase == 1 || phase == 2) continue;
            if (phase == 0 && phase1 != 0) continue;
            if (phase == 3 && !need_step) continue;
            if (phase == 4 && !need_trap) continue;
            if (phase >= 3) {
                lds_order();   // the walks of the phase before are done with sh.val
                if (phase == 3 || !need_step) term_pass();
                else if (kBoth) {
#pragma unroll
                    for (int j = 0; j < kChunksC; j++)
                        if (j * 128 < total)
                            *reinterpret_cast<ulonglong2 *>(&sh.val[j * 128 + 2 * lane]) = make_ulonglong2(
                                (uint64_t)__double_as_longlong(keep_a[kBoth ? j : 0]), (uint64_t)__double_as_longlong(keep_b[kBoth ? j : 0]));
                }
                lds_order();
            }
            uint32_t out_mask = phase == 0 ? (p.kind_mask[0] | p.kind_mask[1]) : phase == 3 ? p.kind_mask[2] : p.kind_mask[3];
            if (phase == phase1) out_mask |= p.kind_mask[4];
            out_mask &= p.col_mask[c];
            const bool first_phase = phase == phase1;

    for (int q = lane; q < nseg_owned; q += kWave) {
        int r0 = g_r0, r1 = g_r1, state = g_state, ka = g_ka, kb = g_kb;
        uint32_t wid = g_wid, next_wid = g_next, e1 = g_e1;
        if (q != lane) {
            state = geometry(q, r0, r1, wid, next_wid, e1);
            if (state == 0) { ka = rank_of(r0); kb = r1 >= kRowsC ? total : rank_of(r1); }
        }
        if (state != 0) {
            // rows run past the look-ahead: hand the window (all its columns) to the cooperative path
            if (c == 0 && first_phase) push_long_window(p.status, p.long_list, p.long_cap, tile, (uint64_t)p.wid_base + wid, base + r0);
            continue;
        }
        const bool next_staged = q + 1 < nseg_total;   // row r1 is a head of this tile
        // does the successor's first row sit exactly on this window's end, and is it a valid point?  Then it is slot kb (rolling.go:201-209)
        const bool incl_pt = p.inclusive && next_staged && next_wid == wid + 1 && (e1 & kStartBitC) && ((sh.vbits[r1 >> 5] >> (r1 & 31)) & 1u);
        const int count = kb - ka;
        const bool has_value = count > 0;
        double sum = 0.0, mn = 0.0, mx = 0.0, integ = 0.0, integ_t = 0.0;
        uint64_t first_raw = 0, last_raw = 0;
        bool trap_nil = true;
        if (phase == 0) {
            if (has_value && need_vals) {
                walk_values<false, true>(sh.val, ka, kb - 1, need_sum, need_mm, exact_mm, sum, mn, mx);
                if (need_fl) {
                    first_raw = sh.val[ka];
                    last_raw = sh.val[kb - 1];
                    if (cint) {   // the staged values are float64(v): First / Last return the Int64 itself (firstlast.go:17, :32)
                        int cnt_, fv, lv;
                        window_valid_rows(sh.vbits, r0, r1, cnt_, fv, lv);
                        first_raw = src[base + fv];
                        last_raw = src[base + lv];
                    }
                }
            }
        } else if (has_value) {
            integ = walk_terms<false, true>(sh.val, ka + 1, kb);
            if (phase == 3) {
                // + v0 * (float64(LastValue) - t0) of the window's last point (integral.go:49-55)
                double pv;
                uint32_t pt;
                if (q == lane) { pv = g_pv; pt = g_pt; }
                else {   // (more than 64 windows in the tile: the value from the column itself)
                    int cnt_, fv, lv;
                    window_valid_rows(sh.vbits, r0, r1, cnt_, fv, lv);
                    const uint64_t raw = src[base + lv];
                    pv = cint ? (double)(int64_t)raw : __longlong_as_double((long long)raw);
                    pt = sh.ctx[kb - 1];
                }
                integ = integ + pv * (last_value_d(wid) - (ws0_d + (double)pt));
            } else {
                int cnt = count;
                if (incl_pt) { integ += __longlong_as_double((long long)sh.val[kb]); cnt++; }   // the term that joins the inclusive row
                trap_nil = cnt < 2;
                integ_t = integ;
            }
        } else if (phase == 4) {
            trap_nil = true;   // (no point of its own: at most the inclusive row - fewer than two points, integral.go:33-35)
        }
        const int nrows = r1 - r0;
        const int64_t win_start = ws0 + (int64_t)((uint64_t)wid * (uint64_t)ik);
        const int64_t slot = (int64_t)wid;
        if (wid >= W32) continue;  // (only on corrupt input)
        const uint32_t gap = next_wid - wid - 1;
#pragma unroll 1
        for (uint32_t am = out_mask; am; am &= am - 1u) {
            const int a = __builtin_amdgcn_readfirstlane(__builtin_ctz(am));
            const int k = p.kind[a];
            typedef const uint64_t __attribute__((address_space(4))) *karg_u64;
            typedef uint64_t __attribute__((address_space(1))) *global_u64;
            const global_u64 out_a = (global_u64)((karg_u64)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(SimpleParams, out_values) / 8 + a];
            uint64_t bits;
            bool nil = false;
            switch (k) {
            case BOWGPU_AGG_WINDOW_START: bits = (uint64_t)win_start; break;
            case BOWGPU_AGG_SUM: bits = (uint64_t)__double_as_longlong(sum); break;
            case BOWGPU_AGG_MEAN: bits = (uint64_t)__double_as_longlong(sum / (double)(int64_t)count); nil = !has_value; break;
            case BOWGPU_AGG_MIN: bits = (uint64_t)__double_as_longlong(mn); nil = !has_value; break;
            case BOWGPU_AGG_MAX: bits = (uint64_t)__double_as_longlong(mx); nil = !has_value; break;
            case BOWGPU_AGG_COUNT: bits = (uint64_t)(int64_t)count; break;
            case BOWGPU_AGG_FIRST: bits = first_raw; nil = !has_value; break;
            case BOWGPU_AGG_LAST: bits = last_raw; nil = !has_value; break;
            case BOWGPU_AGG_INTEGRAL_STEP:                                                  // integral.go:43-68
            case BOWGPU_AGG_WAVG_STEP: {                                                    // weightedmean.go:11-19
                double r = integ;
                if (k == BOWGPU_AGG_WAVG_STEP) r = r / (double)((win_start + p.interval) - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = !has_value;
                break;
            }
            case BOWGPU_AGG_INTEGRAL_TRAPEZOID:                                             // integral.go:11-37 (inclusive window)
            case BOWGPU_AGG_WAVG_LINEAR: {                                                  // weightedmean.go:25-33
                double r = integ_t;
                if (k == BOWGPU_AGG_WAVG_LINEAR) r = r / (double)((win_start + p.interval) - win_start);
                bits = (uint64_t)__double_as_longlong(r);
                nil = trap_nil;
                break;
            }
            default: bits = (uint64_t)__double_as_longlong((double)nrows); break;  // NumRows
            }
            const bool int_result = k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_COUNT || (cint && (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST));
            const int nf = p.nfac[a];
            if (nf) bits = apply_factors(bits, int_result, nf, p.fac[a]);
            if (nil) {  // no value (all null; fewer than two points for the trapezoid): nil => slot 0, bit cleared
                bits = 0;
                atomicAnd(&p.out_valid[a][slot >> 5], ~(1u << (slot & 31)));
            }
            __builtin_nontemporal_store(bits, &out_a[slot]);
            // the empty windows right after this one (rare): values of an empty slice + cleared validity bits
            for (uint32_t g = 1; g <= gap; g++) {
                if (wid + g >= W32) break;
                const int64_t gw = slot + g;
                const int64_t gstart = win_start + (int64_t)((uint64_t)g * (uint64_t)ik);
                uint64_t gbits = k == BOWGPU_AGG_WINDOW_START ? (uint64_t)gstart : 0ull;
                if (nf && (k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_SUM || k == BOWGPU_AGG_NUM_ROWS || k == BOWGPU_AGG_COUNT))
                    gbits = apply_factors(gbits, int_result, nf, p.fac[a]);
                out_a[gw] = gbits;
                if (p.out_valid[a]) atomicAnd(&p.out_valid[a][gw >> 5], ~(1u << (gw & 31)));
            }
        }
    }
        }  // phases
    }  // columns
}

int launch_rolling_twc(Ctx *c, const SimpleParams &p, bool long_halo) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileC - 1) / kTileC;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    const dim3 g((unsigned)grid), blk(kWave);
    const bool both = (p.need & kNeedStep) && (p.need & kNeedTrap);
#define BG_TWC(B, M)                                                                                                         \
    do {                                                                                                                      \
        if (long_halo) hipLaunchKernelGGL((rolling_twc_kernel<B, M, 256>), g, blk, 0, c->stream, p, ntiles, per_xcd);           \
        else hipLaunchKernelGGL((rolling_twc_kernel<B, M, 128>), g, blk, 0, c->stream, p, ntiles, per_xcd);                    \
    } while (0)
    if (p.ncols > 1) { if (both) BG_TWC(true, true); else BG_TWC(false, true); }
    else { if (both) BG_TWC(true, false); else BG_TWC(false, false); }
#undef BG_TWC
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
