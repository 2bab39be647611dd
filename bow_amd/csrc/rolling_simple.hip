// rolling_simple.hip — the tile kernel for the most common shape of Rolling.Aggregate, stripped of every
// descriptor-driven branch:
//   * up to 8 value columns (Float64 / Int64, with or without nulls) plus the interval column;
//   * up to 16 outputs among WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows,
//     transformation.Factor chains applied to the result, exclusive windows;
//   * the rows of this call (the whole frame, or one rank's shard of it) span less than 2^32 from the start of output
//     slot 0 and interval < 2^32, so window ids are 32-bit: wid = magic32((uint32)(ts - s0)) with no per-tile base.
// (reference rolling/rolling.go:177-239 + rolling/aggregation.go:190-238 + the reducer closures of
// rolling/aggregation/{windowstart,sum,arithmeticmean,minmax,count,firstlast}.go.)  Anything else takes
// rolling_fast.hip / rolling_agg.hip; results are identical where several apply (tests run all of them).
//
// Nil results are the exception (empty windows; windows whose values are all null), so the host presets all
// output validity bitmaps to ones and this kernel only CLEARS the bits of nil results, which removes the LDS
// bitmap assembly from the hot path.  Structure otherwise as rolling_fast.hip: one wavefront per tile of
// 512 rows + 128 look-ahead rows, no barriers, heads -> LDS segment list -> one lane walks one window in
// row order (reference summation order, bit-exact).
#include <stddef.h>

#include "agg_device.h"

namespace bowgpu {

namespace {

constexpr int kWave = 64;
constexpr int kTileS = 512;
constexpr int kHaloS = 128;
constexpr int kRowsS = kTileS + kHaloS;
constexpr int kChunksS = kRowsS / 128;
constexpr uint32_t kSatS = 0xFFFFu;
// At most SimpleCap windows may start inside one tile (+ look-ahead); denser tiles send the call to the general lean kernel.
// The kernel's rate follows its OCCUPANCY and the occupancy follows the LDS per wavefront, which is allocated in 1 KB steps
// (same-session A/B at 1e9 rows, three repeats: 6816 B and 6656 B 3.24 ms - 22 wavefronts per CU; 6144 B 3.10 ms - 26).  So the
// head list comes in two sizes, chosen by the host from the plan (n / W):
//   kDense = false: 174 heads (152 with nulls), LDS exactly 6144 B - calls whose windows average >= 5 rows
//   kDense = true : 400 heads (378 with nulls), LDS 6.9 KB          - the rest: windows of 1.6 .. 5 rows, and frames with many
//                   empty windows (n / W says nothing about their non-empty ones); the round-1 layout
// kPad: the staged column carries two pad slots per 32 rows (agg_device.h swz): regular windows of 16 k rows off one LDS bank.  Calls of
// SHORT windows (at most kPlainMaxAvgRows rows on average) over one Float64 column without nulls - the benched shape - take the
// instantiation WITHOUT the pads: lanes that walk windows a few rows apart never met on a bank, and the pads cost them 45 vector
// instructions per tile (432 -> 477: the address arithmetic of swz() and the single steps in front of a walk's first aligned group)
// plus 80 of the head list's entries (254 heads in 6144 B without the pads).
constexpr int64_t kPlainMaxAvgRows = 16;
template <bool kNulls, bool kDense, bool kPad> struct SimpleCap {
    static constexpr int value = kDense ? (kNulls ? 378 : 400) : kPad ? (kNulls ? 152 : 174) : 254;
};
constexpr int kAlignS = 16;   // output slots per 128-byte line: the granule of the slot-aligned hand-over between tiles

template <bool kNulls, bool kDense, bool kPad>
struct SimpleShared {
    uint64_t val[kPad ? swz_slots(kRowsS) : kRowsS];
    uint32_t vbits[kNulls ? kRowsS / 32 + 2 : 1];  // validity words of the value column for this tile (kNulls only)
    uint32_t seg[SimpleCap<kNulls, kDense, kPad>::value + (kNulls ? 2 : 1)];  // heads in row order: local row | (wid - wid of the tile's first row) << 16
};
static_assert(sizeof(SimpleShared<false, false, true>) <= 6144 && sizeof(SimpleShared<true, false, true>) <= 6144 &&
              sizeof(SimpleShared<false, false, false>) <= 6144, "LDS of the small-list forms: 6 KB (26 wavefronts per CU)");

__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// 16-byte loads / 8-byte stores with the non-temporal hint: rows 128..511 of a tile are read by this wavefront only, so their
// lines should be the first to leave the XCD's L2; the first and the last chunk are shared with the neighbouring tiles (the
// look-ahead) and stay plain.  Measured on the benched shape (scratch/headline_ab.hip, 1e9 rows): -2 % kernel time for the
// loads, another -1.5 % with the outputs stored non-temporal (they are never read back by this kernel).
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 load16_nt(const ulonglong2 *q) {
    const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(q));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void store8_nt(uint64_t *p, uint64_t v) { __builtin_nontemporal_store(v, p); }
// the double that the lane kCtrl names holds (row_shr:N inside a row of 16 lanes); a lane without such a neighbour keeps its own
template <int kCtrl>
__device__ __forceinline__ double row_dpp64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), kCtrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Tiles of FEW windows (at most kCoopMaxHeads heads in the tile's 640 rows: windows of ~92 rows and more; ~128 when the window's lane walks
// its rows for a sum anyway - the extrema then ride along at 2 instructions per row): the extrema of the tile's windows
// are found by ALL 64 lanes - 16 lanes per window, four windows per pass, each lane over a contiguous piece of its window - instead of
// by one lane per window (minmax.go:16-28 walks left to right; 192-row windows left 61 lanes idle for 192 steps).  Extrema are order-free
// apart from two rules of the reference's loop, both kept: a NaN first value stays (the window's lane applies it), and among values that
// compare equal - -0.0 and +0.0 - the EARLIER row stays: pieces are contiguous and ascend with the lane, a lane steps through its piece
// in row order with `<` / `>`, and the combine takes the LATER lane's value only when it is strictly better.
// Same-box A/B at 1e8 rows (profiles/r06_stdout_coop_extrema_ab.txt; kernel ms with / without, dense): Min + Max 96 rows 0.299 / 0.306, 128 rows
// 0.285 / 0.330, 160 rows 0.340 / 0.392, 192 rows 0.359 / 0.428; with nulls 128 rows 0.321 / 0.386; Sum + Min + Max 144 rows 0.398 / 0.416 - but
// 64 rows 0.338 / 0.305 and 96 rows 0.361 / 0.331 with twelve heads allowed, hence the two limits.
#ifndef BOWGPU_COOP_MAX_HEADS
#define BOWGPU_COOP_MAX_HEADS 7   // (A/B: scratch/build_variant.sh nocoop rolling_simple.hip -DBOWGPU_COOP_MAX_HEADS=0)
#endif
constexpr int kCoopMaxHeads = BOWGPU_COOP_MAX_HEADS;
constexpr int kCoopMaxHeadsSum = BOWGPU_COOP_MAX_HEADS < 5 ? BOWGPU_COOP_MAX_HEADS : 5;
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// kNeed: bit0 min/max wanted, bit1 first/last wanted; kInt: Int64 value column (several columns carry their types in
// p.col_is_int); kNulls: some column has nulls;
// kMulti: more than one value column (the single-column shape keeps its straight-line form)
// kWide: the rows of the call span 2^32 or more from slot 0 (nanosecond timestamps): window ids are taken relative to the
// tile's first window (one exact 64-bit division on the scalar unit per tile), which only needs each TILE's rows within 2^32
template <int kNeed, bool kInt, bool kNulls, bool kMulti, bool kWide, bool kDense, bool kPad>
__global__ __launch_bounds__(kWave, 6) void rolling_simple_kernel(const SimpleParams p, const int64_t ntiles,
                                                                  const int64_t tiles_per_xcd) {
    static_assert(kPad || (!kInt && !kNulls && !kMulti), "the unpadded form: one Float64 column without nulls");
    __shared__ SimpleShared<kNulls, kDense, kPad> sh;
    constexpr int kSegCapS = SimpleCap<kNulls, kDense, kPad>::value;
    constexpr bool kSwzS = kPad;
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);  // XCD-contiguous tile runs (look-ahead rows hit the same L2)
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int64_t base = tile * kTileS;
    const int64_t n = p.n;
    const bool interior = base + kRowsS <= n;
    const int nloc = interior ? kRowsS : (int)(n - base);

    // ---- loads: ts, then the first value column right behind it
    uint64_t ta[kChunksS], tb[kChunksS], va[kChunksS], vb[kChunksS];
    const uint64_t *__restrict__ ts = reinterpret_cast<const uint64_t *>(p.ts);
    // a column whose first row is only 8-byte aligned (an Arrow slice with an odd offset) is read with two 8-byte loads per lane and
    // chunk instead of one 16-byte load: slower through the L1, but the call stays on this kernel
    auto load_col = [&](const uint64_t *__restrict__ src, uint64_t (&a)[kChunksS], uint64_t (&bb)[kChunksS], bool aligned) {
        if (interior && aligned) {
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(src + base) + lane;
#pragma unroll
            for (int j = 0; j < kChunksS; j++) {
                const ulonglong2 x = (j > 0 && j < kChunksS - 1) ? load16_nt(q + j * 64) : q[j * 64];
                a[j] = x.x; bb[j] = x.y;
            }
        } else if (interior) {
            const uint64_t *q = src + base + 2 * lane;
#pragma unroll
            for (int j = 0; j < kChunksS; j++) { a[j] = q[j * 128]; bb[j] = q[j * 128 + 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < kChunksS; j++) load_pair(src, base + j * 128 + 2 * lane, n, aligned, a[j], bb[j]);
        }
    };
    // (Round 2 looked at status[4] here - "some tile has already found that the call needs the other kernel: stop early" - with a load
    // issued in front of the columns'.  The compiler made every wavefront WAIT for that load before it issued the columns' loads: a
    // whole memory round trip in front of the real ones, in every tile of every call, to shorten a call that is redone anyway.  Gone.)
    // The usual tile - interior, both columns 16-byte aligned - issues its ten loads back to back in ONE block.  Written as two
    // load_col() calls (three-way branches each) the compiler saw, at every join, loads pending into registers that another path
    // loads into as well, and put a wait for ALL outstanding loads in front of the next ones: the flag's load, the interval
    // column went out one memory round trip before the value column instead of together.
    if (interior && !(p.unaligned_mask & 0x80000001u)) {
        const ulonglong2 *qt = reinterpret_cast<const ulonglong2 *>(ts + base) + lane;
        const ulonglong2 *qv = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const uint64_t *>(p.values[0]) + base) + lane;
#pragma unroll
        for (int j = 0; j < kChunksS; j++) {
            const ulonglong2 x = (j > 0 && j < kChunksS - 1) ? load16_nt(qt + j * 64) : qt[j * 64];
            ta[j] = x.x; tb[j] = x.y;
        }
#pragma unroll
        for (int j = 0; j < kChunksS; j++) {
            const ulonglong2 x = (j > 0 && j < kChunksS - 1) ? load16_nt(qv + j * 64) : qv[j * 64];
            va[j] = x.x; vb[j] = x.y;
        }
    } else {
        load_col(ts, ta, tb, !(p.unaligned_mask >> 31));
        load_col(reinterpret_cast<const uint64_t *>(p.values[0]), va, vb, !(p.unaligned_mask & 1u));
    }
    // the row left of the tile (scalar load): first head flag + order check
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    // ids are 32-bit and relative to window w0, which starts at ws0: slot 0 of the call, or (kWide) the tile's first window
    uint64_t w0 = 0;
    int64_t ws0 = p.s0;
    bool unsorted = false, sat = false;  // rows out of order ; ids the 16-bit local fields / 32-bit arithmetic cannot hold
    const int64_t ts_first = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta[0] >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta[0]));
    // p.pre_rows: Go's truncating division put s0 above a negative first timestamp (rolling.go:96-99); the rows below s0 ride in
    // window 0 (rolling.go:194-196).  They are the frame's first rows, fewer than one interval's worth of time: ids below are
    // forced to 0 for them (wave-uniform test per tile: only tiles that start below s0 pay for the per-row comparison)
    const bool pre = p.pre_rows && ts_first < p.s0;
    if (kWide) {
        w0 = pre ? 0ull : magic_div((uint64_t)ts_first - (uint64_t)p.s0, p.magic);
        ws0 = p.s0 + (int64_t)(w0 * (uint64_t)p.interval);
        const int64_t ts_last = p.ts[base + nloc - 1];
        // ids come from (ts - ws0) >> k with k = trailing zero bits of the interval (floor(a / b) == floor((a >> k) / (b >> k)) when
        // 2^k divides b): the tile's rows must lie within 2^(32+k) of ws0 - 2199 s for 1 s windows of nanosecond timestamps
        sat = ts_last < ts_first || (((uint64_t)ts_last - (uint64_t)ws0) >> p.shift_k) >= 0xFFFFFFF0ull;  // (unsorted rows are caught below too)
    }
    const uint32_t s0_lo = (uint32_t)ws0;
    auto rel32 = [&](int64_t t) -> uint32_t {  // timestamp -> 32-bit numerator of the window id
        return kWide ? (uint32_t)(((uint64_t)t - (uint64_t)ws0) >> p.shift_k) : (uint32_t)t - s0_lo;
    };

    // ---- window ids (32-bit), head flags, compaction with a running scalar count
    const uint32_t w_first = (kWide || pre) ? 0u : mdiv32((uint32_t)ts_first - s0_lo, p.m32, p.sh1, p.sh2);
    uint32_t left_w = base == 0 ? 0xFFFFFFFEu : (pre && left0 < ws0) ? 0u : (kWide && left0 < ws0) ? 0xFFFFFFFEu : mdiv32(rel32(left0), p.m32, p.sh1, p.sh2);
    int64_t left_ts = left0;
    int nseg_total = 0, nseg_owned = 0;
    bool snan = false;   // (kNeed & 1) a signalling NaN among this lane's staged values: the tile's extrema are walked by comparison (agg_device.h is_snan)
#pragma unroll
    for (int j = 0; j < kChunksS; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
        const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
        const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
        unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
        const uint32_t ra = rel32(tsa), rb = rel32(tsb);
        uint32_t wa = mdiv32(ra, p.m32, p.sh1, p.sh2);
        uint32_t wb = mdiv32(rb, p.m32, p.sh1, p.sh2);
        if (pre) { if (tsa < ws0) wa = 0u; if (tsb < ws0) wb = 0u; }
        const uint32_t wprev = left32(wb, left_w);
        const bool ha = pa && (wa != wprev);
        const bool hb = pb && (wb != wa);
        const uint32_t la = wa - w_first, lb = wb - w_first;
        sat |= (ha && la >= kSatS) || (hb && lb >= kSatS);
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        if (ha && pos < kSegCapS) sh.seg[pos] = (uint32_t)l | (la << 16);
        pos += ha ? 1 : 0;
        if (hb && pos < kSegCapS) sh.seg[pos] = (uint32_t)(l + 1) | (lb << 16);
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == kChunksS - 2) nseg_owned = nseg_total;
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
        // single column: its values go to LDS now (their registers die here)
        if (!kMulti && !kNulls && !kInt) {
            *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzS>(l)]) = make_ulonglong2(va[j], vb[j]);
            if (kNeed & 1) snan = snan || is_snan(va[j]) || is_snan(vb[j]);
        }
    }
    if (__ballot(unsorted)) {  // the call fails with BOWGPU_ERR_TS_UNSORTED
        if (lane == 0) atomicOr(&p.status[0], 1u);
        return;
    }
    if (nseg_total > kSegCapS) sat = true;
    if (__ballot(sat)) {  // a tile the 16-bit local ids / the segment list cannot describe: the host redoes the call with the general lean kernel
        // (one atomic per call, not one per tile: 2e5 atomics on one address took 2 ms)
        if (lane == 0 && !__hip_atomic_load(&p.status[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&p.status[4], 1u);
        return;
    }

    const bool reaches_end = base + kRowsS >= n;
    // ---- which windows this wavefront outputs.  By default those that START in its 512 rows.  Slot-aligned hand-over: the
    // windows at the front of a tile whose slots lie below the next multiple of kAlignS belong to the tile on the LEFT, so that
    // every run of output slots a wavefront stores begins on a 128-byte line and no line is ever written by two wavefronts
    // (partial lines from two writers cost HBM bandwidth: -1.4 % kernel time on the benched shape).  Both neighbours decide
    // from the same 128 rows - the left tile's look-ahead is the right tile's first chunk - and the hand-over only happens
    // when the head that CLOSES the handed-over windows lies inside those rows; otherwise each tile keeps its own.
    int q_start = 0, q_end = nseg_owned;
    {
        lds_order();                                // the segment list is complete
        const uint32_t slot_lo = (uint32_t)w0;      // output slot = w0 + id: its low bits are the same whichever tile computes them
        // heads are in id order and ids grow by at least one per head: the heads below the aligned id are among the first kAlignS
        auto handover = [&](int qf, int qlim) -> int {   // qf: first head of a tile; qlim: heads inside the shared 128 rows end here
            if (qf >= qlim) return qf;
            const uint32_t gf = slot_lo + w_first + (sh.seg[qf] >> 16);            // low bits of the output slot
            const uint32_t A = (gf + (kAlignS - 1)) & ~(uint32_t)(kAlignS - 1);
            if (A == gf) return qf;
            const int qi = qf + lane;
            const bool below = lane < kAlignS && qi < qlim && (slot_lo + w_first + (sh.seg[qi < qlim ? qi : qf] >> 16)) < A;
            const int nb = __popcll(__ballot(below));
            return qf + nb < qlim ? qf + nb : qf;
        };
        int n128 = 0;   // heads inside this tile's first 128 rows (at most 128 of them: two list entries per lane)
        {
            const int qa = lane, qb = lane + 64;
            const bool a = qa < nseg_total && (int)(sh.seg[qa < nseg_total ? qa : 0] & 0xFFFFu) < 128;
            const bool bq = qb < nseg_total && (int)(sh.seg[qb < nseg_total ? qb : 0] & 0xFFFFu) < 128;
            n128 = __popcll(__ballot(a)) + __popcll(__ballot(bq));
        }
        if (tile > 0) q_start = handover(0, n128);
        q_end = handover(nseg_owned, nseg_total);
    }
    // windows of the call, as an id relative to w0 (the last tile's successor id when the data ends in it)
    const uint64_t Wrel = (uint64_t)p.W - w0;
    const uint32_t W32 = Wrel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)Wrel;
    // which running sums the outputs read (wave-uniform): a call without Sum / ArithmeticMean skips the additions, and a nullable
    // column then stages its nulls as NaN straight away (see below)
    const bool need_sum = p.need & kNeedSum;
    // validity words of one column's 640 rows, 32 per lane (lanes 0..19), any bit offset (Arrow slices); issued WITH the column's
    // value loads so that they share one memory round trip
    auto load_vword = [&](int c) -> uint32_t {
        uint32_t word = 0xFFFFFFFFu;
        if (lane < kRowsS / 32 && p.vbits[c] != nullptr) {
            const int64_t bit = p.vbit0[c] + base + 32 * (int64_t)lane;
            const int64_t wi = bit >> 5;
            const int shb = (int)(bit & 31);
            const uint32_t lo = wi < p.vwords[c] ? p.vbits[c][wi] : 0u;
            const uint32_t hi = (shb != 0 && wi + 1 < p.vwords[c]) ? p.vbits[c][wi + 1] : 0u;
            word = shb ? ((lo >> shb) | (hi << (32 - shb))) : lo;
        }
        return word;
    };
    uint32_t vword = kNulls ? load_vword(0) : 0u;
    // ---- one pass per value column: stage its values in LDS (the next column's loads go out first), walk, store.
    // What is staged is what the walk adds, so the walk itself is branch-free (round 4; the per-row validity test, type switch and
    // first-value test of the round-1 walk cost 14 instructions per row on a chain only 512 / w lanes deep):
    //   * Int64 columns are converted to float64 here, ten rows per lane in parallel, not one per step of the walk;
    //   * a NULL row holds +0.0 while sums are walked - exact: sum.go:16 starts at +0.0 and x + (+0.0) == x bit for bit for every
    //     x but -0.0, which a sum that started at +0.0 can never be - and a quiet NaN while extrema are walked (x < mn, x > mx are
    //     false for it: minmax.go:22-27 would have skipped the row);
    //   * count / first / last valid row of a window come from its validity words (popcount, count-leading / trailing-zeros).
    const int ncols = kMulti ? p.ncols : 1;
    for (int c = 0; c < ncols; c++) {
        const bool cint = kMulti ? (p.col_is_int[c] != 0) : kInt;  // mixed column types: per pass (uniform)
        constexpr bool kStageHere = kMulti || kNulls || kInt;
        if (kStageHere) {
            lds_order();  // the previous pass is done with sh.val / sh.vbits
            if (kNulls) {
                if (lane < kRowsS / 32) sh.vbits[lane] = vword;
                lds_order();
            }
            const uint64_t fill = need_sum ? 0ull : kNullAsNaN;
            snan = false;
#pragma unroll
            for (int j = 0; j < kChunksS; j++) {
                uint64_t xa = va[j], xb = vb[j];
                if (cint) {
                    xa = (uint64_t)__double_as_longlong((double)(int64_t)xa);
                    xb = (uint64_t)__double_as_longlong((double)(int64_t)xb);
                }
                if (kNulls) {
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) xa = fill;
                    if (!(two & 2u)) xb = fill;
                }
                if ((kNeed & 1) && !cint) snan = snan || is_snan(xa) || is_snan(xb);   // (what is staged: a null row's own bits are gone)
                *reinterpret_cast<ulonglong2 *>(&sh.val[swz<kSwzS>(j * 128 + 2 * lane)]) = make_ulonglong2(xa, xb);
            }
            if (kMulti && c + 1 < ncols) {
                load_col(reinterpret_cast<const uint64_t *>(p.values[c + 1]), va, vb, !((p.unaligned_mask >> (c + 1)) & 1u));
                if (kNulls) vword = load_vword(c + 1);
            }
        }
        lds_order();
        const bool exact_mm = (kNeed & 1) && __ballot(snan) != 0ull;
        // a nullable column whose outputs want sums AND extrema is walked twice: phase 1 with +0.0 in the null rows (everything but
        // Min / Max), then the null rows are overwritten with NaN and phase 2 walks the extrema.  Every other shape: phase 0, one walk.
        const bool two_phase = kNulls && (kNeed & 1) && need_sum && nseg_total <= kTwoWalksMaxHeads;
        const bool pred_walk = kNulls && (kNeed & 1) && need_sum && !two_phase;   // one walk, extrema under the validity bit
        for (int phase = two_phase ? 1 : 0; phase <= (two_phase ? 2 : 0); phase++) {
            if (phase == 2) {
                lds_order();
#pragma unroll
                for (int j = 0; j < kChunksS; j++) {
                    const uint32_t two = sh.vbits[j * 4 + (lane >> 4)] >> ((2 * lane) & 31);
                    if (!(two & 1u)) sh.val[swz<kSwzS>(j * 128 + 2 * lane)] = kNullAsNaN;
                    if (!(two & 2u)) sh.val[swz<kSwzS>(j * 128 + 2 * lane) + 1] = kNullAsNaN;
                }
                lds_order();
            }
            const bool do_sum = need_sum && phase != 2;
            const bool do_mm = (kNeed & 1) && phase != 1;

            // ---- the extrema of a tile of few windows, by all lanes (see kCoopMaxHeads)
            double coop_mn = 0.0, coop_mx = 0.0;
            const int nw_own = q_end - q_start;
            const bool coop = (kNeed & 1) && do_mm && !pred_walk && nw_own > 0 && nseg_total <= (do_sum ? kCoopMaxHeadsSum : kCoopMaxHeads);
            if ((kNeed & 1) && coop) {
                const int sub = lane & 15, grp = lane >> 4;
#pragma unroll 1
                for (int j0 = 0; j0 < nw_own; j0 += 4) {
                    const int wq = q_start + j0 + grp;
                    int a0 = 0, a1 = 0;
                    if (j0 + grp < nw_own) {
                        a0 = (int)(sh.seg[wq] & 0xFFFFu);
                        a1 = wq + 1 < nseg_total ? (int)(sh.seg[wq + 1] & 0xFFFFu) : (reaches_end ? nloc : a0);   // (a window that runs past the look-ahead is queued: nothing to do here)
                    }
                    const int per = (a1 - a0 + 15) >> 4;
                    int r = a0 + sub * per;
                    const int rend = r + per < a1 ? r + per : a1;
                    double mnl = __longlong_as_double(0x7FF0000000000000ll), mxl = __longlong_as_double((long long)0xFFF0000000000000ull);
                    for (; r < rend; r++) {
                        const double x = __longlong_as_double((long long)sh.val[swz<kSwzS>(r)]);
                        if (x < mnl) mnl = x;
                        if (x > mxl) mxl = x;
                    }
                    // ordered inclusive scan over the row's 16 lanes: what the earlier lanes hold stays unless this lane's is strictly better
#define BG_COOP_STEP(CTRL, DIST)                                            \
                    {                                                        \
                        const double en = row_dpp64<CTRL>(mnl), ex = row_dpp64<CTRL>(mxl);  \
                        if (sub >= DIST) { if (!(mnl < en)) mnl = en; if (!(mxl > ex)) mxl = ex; } \
                    }
                    BG_COOP_STEP(0x111, 1) BG_COOP_STEP(0x112, 2) BG_COOP_STEP(0x114, 4) BG_COOP_STEP(0x118, 8)
#undef BG_COOP_STEP
                    // lane 15 of row i holds window j0 + i's extrema: hand them to the window's own lane (lane j0 + i of this pass)
                    const int src = (16 * (lane - j0) + 15) & 63;
                    const double gmn = __shfl(mnl, src), gmx = __shfl(mxl, src);
                    if (lane >= j0 && lane < j0 + 4 && lane < nw_own) { coop_mn = gmn; coop_mx = gmx; }
                }
            }

    for (int q = q_start + lane; q < q_end; q += kWave) {
        const uint32_t e0 = sh.seg[q], e1 = sh.seg[q + 1];
        const int r0 = (int)(e0 & 0xFFFFu);
        const uint32_t wid = w_first + (e0 >> 16);
        int r1;
        uint32_t next_wid;
        if (q + 1 < nseg_total) {
            r1 = (int)(e1 & 0xFFFFu);
            next_wid = w_first + (e1 >> 16);
        } else if (reaches_end) {
            r1 = nloc;
            next_wid = W32;
        } else {
            // rows run past the look-ahead: hand the window (all its columns) to the cooperative path
            if (c == 0 && phase != 2) {
                push_long_window(p.status, p.long_list, p.long_cap, tile, (uint64_t)p.wid_base + w0 + wid, base + r0);
            }
            continue;
        }
        // window 0 made only of rows below s0 is an EMPTY slice in the reference (rolling.go:194-196: lastRowIndex stays -1)
        const bool dead = pre && tile == 0 && q == 0 && p.ts[base + r1 - 1] < p.s0;
        // ---- valid rows of the window: how many, the first, the last; then the walk (agg_device.h)
        int count = r1 - r0, fv = r0, lv = r1 - 1;
        if (kNulls) window_valid_rows(sh.vbits, r0, r1, count, fv, lv);
        if (dead) count = 0;
        double sum = 0.0, mn = 0.0, mx = 0.0;
        uint64_t first_raw = 0, last_raw = 0;
        if (count > 0) {
            first_raw = sh.val[swz<kSwzS>(fv)];
            if (kNulls && pred_walk) walk_values_pred<kSwzS>(sh.val, sh.vbits, fv, lv, sum, mn, mx);
            else if ((kNeed & 1) && coop) {
                // the window's extrema were found by all lanes above; what is left to the window's lane: the sum (in row order) and the
                // reference's seed rule - a NaN first value is never replaced (minmax.go:16-28)
                if (do_sum) walk_values<kSwzS, !kMulti>(sh.val, fv, lv, true, false, false, sum, mn, mx);
                const double seed = __longlong_as_double((long long)first_raw);
                if (seed != seed) { mn = seed; mx = seed; } else { mn = coop_mn; mx = coop_mx; }
            }
            else walk_values<kSwzS, !kMulti>(sh.val, fv, lv, do_sum, do_mm, exact_mm, sum, mn, mx);
            if (kNeed & 2) {
                last_raw = sh.val[swz<kSwzS>(lv)];
                if (cint) {   // the staged values are float64(v): First / Last return the Int64 itself (firstlast.go:17, :32)
                    const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(p.values[c]);
                    first_raw = src[base + fv];
                    last_raw = src[base + lv];
                }
            }
        }
        const int nrows = dead ? 0 : r1 - r0;
        const bool has_value = count > 0;  // (always true without nulls, but for the dead window 0)
        const int64_t win_start = ws0 + (int64_t)((uint64_t)wid * (uint64_t)(uint32_t)p.interval);
        const int64_t slot = (int64_t)(w0 + wid);  // output slot
        if (wid >= W32) continue;  // (only on corrupt input)
        const uint32_t gap = next_wid - wid - 1;
        // ---- outputs of this column: lane q -> slot wid
        // (the output's descriptor - kind, pointers, factors - through SCALAR loads: the index is made wave-uniform explicitly.  Left
        // to itself the compiler fetched p.out_values[a] with a VECTOR load from the kernel-argument block, and because gfx950 counts
        // loads and stores in one counter, the wait for that pointer was also a wait for the previous output's store to be
        // acknowledged: two memory round trips per output in every wavefront's epilogue)
#pragma unroll 1
        for (int a_ = 0; a_ < p.naggs; a_++) {
            const int a = __builtin_amdgcn_readfirstlane(a_);
            if (kMulti && p.col[a] != c) continue;
            const int k = p.kind[a];
            if (phase != 0 && (phase == 2) != (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX)) continue;   // two walks: each output once
            // (p is the kernel's first argument: its bytes start the kernel-argument segment, which is constant address space)
            typedef const uint64_t __attribute__((address_space(4))) *karg_u64;
            typedef uint64_t __attribute__((address_space(1))) *global_u64;   // (a pointer read as an integer has lost its address space: say it)
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
            default: bits = (uint64_t)__double_as_longlong((double)nrows); break;  // NumRows
            }
            const bool int_result = k == BOWGPU_AGG_WINDOW_START || k == BOWGPU_AGG_COUNT || (cint && (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST));
            const int nf = p.nfac[a];
            if (nf) bits = apply_factors(bits, int_result, nf, p.fac[a]);
            if ((kNulls || pre) && nil) {  // a window whose values are all null (rare) / the dead window 0: nil => slot 0, bit cleared
                bits = 0;
                atomicAnd(&p.out_valid[a][slot >> 5], ~(1u << (slot & 31)));
            }
            __builtin_nontemporal_store(bits, &out_a[slot]);
            // the empty windows right after this one (rare): values of an empty slice + cleared validity bits
            // (A.9 "Empty slice": WindowStart s_k ; Sum 0.0 ; Count 0 ; NumRows 0.0 ; the rest nil)
            for (uint32_t g = 1; g <= gap; g++) {
                if (wid + g >= W32) break;
                const int64_t gw = slot + g;
                const int64_t gstart = win_start + (int64_t)((uint64_t)g * (uint64_t)(uint32_t)p.interval);
                uint64_t gbits = k == BOWGPU_AGG_WINDOW_START ? (uint64_t)gstart : 0ull;
                // (Sum / NumRows of an empty slice are +0.0 and Count is 0: a negative factor still turns the floats into -0.0)
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

bool rolling_simple_plain(const SimpleParams &p, bool is_int, bool has_nulls) {
    return p.ncols <= 1 && !is_int && !has_nulls && p.W > 0 && p.n / p.W <= kPlainMaxAvgRows && !(route_mask() & BOWGPU_ROUTE_SIMPLE_PADDED);
}

int launch_rolling_simple(Ctx *c, const SimpleParams &p, int need, bool is_int, bool has_nulls, bool wide, bool dense) {
    if (p.n <= 0) return 0;
    const int64_t ntiles = (p.n + kTileS - 1) / kTileS;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t grid = per_xcd * 8;
    if (grid > 0x7FFFFFFFll) return fail(BOWGPU_ERR_UNSUPPORTED, "too many rows for one launch: %lld", (long long)p.n);
    const dim3 g((unsigned)grid), blk(kWave);
    const bool plain = rolling_simple_plain(p, is_int, has_nulls);
    // (the instantiation's name as rocprofv3 prints it: bowgpu_last_kernel_name() - what ties a committed counter file to this launch)
#define BG_GO(N, I, U, M, WD, D, PD)                                                                                          \
    do {                                                                                                                      \
        hipLaunchKernelGGL((rolling_simple_kernel<N, I, U, M, WD, D, PD>), g, blk, 0, c->stream, p, ntiles, per_xcd);          \
        c->last_kernel_name = "rolling_simple_kernel<" #N ", " #I ", " #U ", " #M ", " #WD ", " #D ", " #PD ">";               \
    } while (0)
#define BG_LAUNCH3(N, I, U, M, PD)                                       \
    do {                                                                 \
        if (wide && dense) BG_GO(N, I, U, M, true, true, PD);            \
        else if (wide) BG_GO(N, I, U, M, true, false, PD);               \
        else if (dense) BG_GO(N, I, U, M, false, true, PD);              \
        else BG_GO(N, I, U, M, false, false, PD);                        \
    } while (0)
#define BG_LAUNCH(N, I, U)                                                \
    do {                                                                  \
        if (p.ncols > 1) BG_LAUNCH3(N, I, U, true, true);                 \
        else BG_LAUNCH3(N, I, U, false, true);                            \
    } while (0)
#define BG_NEED(I, U)                                                                                   \
    switch (need) { case 0: BG_LAUNCH(0, I, U); break; case 1: BG_LAUNCH(1, I, U); break;              \
                    case 2: BG_LAUNCH(2, I, U); break; default: BG_LAUNCH(3, I, U); break; }
    if (plain) {
        switch (need) { case 0: BG_LAUNCH3(0, false, false, false, false); break; case 1: BG_LAUNCH3(1, false, false, false, false); break;
                        case 2: BG_LAUNCH3(2, false, false, false, false); break; default: BG_LAUNCH3(3, false, false, false, false); break; }
    } else if (is_int) { if (has_nulls) { BG_NEED(true, true) } else { BG_NEED(true, false) } }
    else { if (has_nulls) { BG_NEED(false, true) } else { BG_NEED(false, false) } }
#undef BG_NEED
#undef BG_LAUNCH
#undef BG_LAUNCH3
#undef BG_GO
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
