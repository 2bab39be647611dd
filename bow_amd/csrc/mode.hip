// mode.hip — aggregation.Mode on the device (reference rolling/aggregation/mode.go:8-32).
//
// The reference walks a window's rows in order, counts each non-nil value in a map[interface{}]int and keeps the value
// whose count first exceeds every count seen so far.  Restated without the map: with nb(i) = #{valid j <= i : v_j == v_i}
// (1 for a NaN: a NaN map key never matches) and M = max nb, the result is the value of the FIRST row with nb(i) == M -
// counts grow by one, so the running maximum reaches M exactly there.  Key equality is Go's == on the dynamic type:
// Int64 compares bits, Float64 compares numerically (-0 == +0; the value returned is the row's own).
//
// Mode is not a streaming reducer (no constant-size state), so it does not ride in the tile kernels; it runs over the
// window row ranges (first row of every window: interp_fill.hip window_first_rows_kernel) in size classes:
//   <= 32 rows     mode_small_kernel : one lane per window, validity as a 32-bit mask, the window copied to the lane's LDS
//                                      column, O(n^2) compares out of LDS
//   <= kModeWave   mode_wave_kernel  : one wavefront per window: a 512-slot LDS hash table of (key, count), no barrier
//   <= kModeHash   mode_mid_kernel   : one workgroup per window: an LDS hash table counts the keys, and the answer is the
//                                      smallest LAST row among the keys with the largest count (the first row at which a count
//                                      reaches M is the last row of a key that ends at M) - linear in the rows
//   <= kModeMid    mode_big_kernel   : the same with 16384 slots that hold the claiming ROW instead of the key (128 KB of LDS)
//   <= kModeGlobal mode_global_kernel: the same table in global memory, 2^lg >= 2 n slots per window, one workgroup per window
//   longer         mode_long_*       : valid rows selected, keyed (canonical bits; NaNs get unique keys), radix-sorted
//                                      stably with their row number (hipCUB), run lengths by binary search from each run head,
//                                      (length, row of the run's M-th element) reduced with a 64-bit atomic max
// HBM-bound / latency-bound integer work: no MFMA anywhere.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "agg_device.h"
#include "bitmap_device.h"

namespace bowgpu {

namespace {

constexpr int kModeSmall = 32;
constexpr int kModeMid = 7680;  // the largest window a workgroup takes (mode_big_kernel's table at load 0.47)
constexpr int kModeWave = 256;   // up to this many rows: one WAVEFRONT per window (a 512-slot table in its LDS slice, no barrier)
constexpr int kModeGlobal = 2000000;  // up to this many rows: a hash table in global memory, one workgroup per window; beyond: radix sort
constexpr int kModeHash = 2560;  // windows up to this many rows: an LDS hash table of kHashSlots (key, count) slots instead of the O(n^2) scan
constexpr int kHashSlots = 4096;

struct ModeParams {
    const uint64_t *values;
    const uint32_t *vbits;  // nullptr: no nulls
    int64_t vbit0;
    const int64_t *ts;
    const int64_t *first_idx;  // [W + 1] first row of every window
    int64_t s0, W, n, interval;
    int32_t pre_rows, is_int;
    int32_t nfac, inclusive;
    double fac[BOWGPU_MAX_FACTORS];
    uint64_t *out_values;
    uint32_t *out_valid;
    uint32_t *counters;  // [0] = mid-size windows queued, [1] = long windows queued, [2] = windows for one wavefront each
    int64_t *mid_queue, *long_queue, *wave_queue, *big_queue;  // counters[3] = windows for mode_big_kernel
};

__device__ __forceinline__ bool mode_eq(uint64_t a, uint64_t b, bool is_int) {
    return is_int ? a == b : __longlong_as_double((long long)a) == __longlong_as_double((long long)b);
}

__device__ __forceinline__ void window_rows(const ModeParams &p, int64_t k, int64_t *a, int64_t *b) {
    int64_t lo = p.first_idx[k], hi = p.first_idx[k + 1];
    // rows below s0 ride in window 0, but alone they do not make a window (rolling.go:177-239) - unless the call is inclusive
    // and the next window's first row sits exactly on its start: that row makes window 0 exist, and once it is dropped again
    // (Window.UnsetInclusive, window.go:23-31) the rows below s0 are what Mode sees
    if (k == 0 && p.pre_rows && !(hi > 0 && p.ts[hi - 1] >= p.s0) && !(p.inclusive && hi < p.n && p.ts[hi] == p.s0 + p.interval)) hi = lo;
    *a = lo;
    *b = hi;
}

__device__ __forceinline__ bool row_valid(const ModeParams &p, int64_t row) { return !p.vbits || bit_at(p.vbits, p.vbit0, row); }

__device__ __forceinline__ void store_result(const ModeParams &p, int64_t k, uint64_t v) {
    p.out_values[k] = apply_factors(v, p.is_int != 0, p.nfac, p.fac);
    atomicOr(&p.out_valid[k >> 5], 1u << (k & 31));
}

constexpr int kSmallThreads = 128;
__global__ __launch_bounds__(kSmallThreads) void mode_small_kernel(ModeParams p) {
    // a lane's window goes to its own LDS column first (sw[row][thread]: conflict-free): the O(n^2) compares then read LDS -
    // the same reads from global memory touch a different cache line per lane, 64 lines per wavefront instruction
    __shared__ uint64_t sw[kModeSmall * kSmallThreads];
    const int64_t k = (int64_t)blockIdx.x * kSmallThreads + threadIdx.x;
    bool have = false, to_mid = false, to_long = false, to_wave = false, to_big = false;
    uint64_t res = 0;
    if (k < p.W) {
        int64_t a, b;
        window_rows(p, k, &a, &b);
        const int64_t n = b - a;
        if (n > kModeSmall) {
            to_wave = n <= kModeWave;
            to_mid = !to_wave && n <= kModeHash;
            to_big = n > kModeHash && n <= kModeMid;
            to_long = n > kModeMid;
        } else if (n > 0) {
            uint32_t mask = 0;
            uint64_t *mine = sw + threadIdx.x;
            for (int i = 0; i < (int)n; i++) {
                mask |= row_valid(p, a + i) ? (1u << i) : 0u;
                mine[i * kSmallThreads] = p.values[a + i];
            }
            int best = 0;
            const bool is_int = p.is_int != 0;
            for (int i = 0; i < (int)n; i++) {
                if (!((mask >> i) & 1u)) continue;
                const uint64_t v = mine[i * kSmallThreads];
                int nb = 0;
                for (int j = 0; j < i; j++) nb += (((mask >> j) & 1u) && mode_eq(mine[j * kSmallThreads], v, is_int)) ? 1 : 0;
                nb += 1;  // this row (a NaN: its own fresh key)
                if (nb > best) { best = nb; res = v; }
            }
            have = best > 0;
        }
        if (!to_mid && !to_long && !to_wave && !to_big) p.out_values[k] = have ? apply_factors(res, p.is_int != 0, p.nfac, p.fac) : 0ull;  // nil slots hold 0 (bowbuffer.go:22-40)
    }
    const int lane = threadIdx.x & 63;
    // queue pushes, one atomic per wavefront and queue: atomics on a single address are serialised (~10 ns each: 2e6 windows of
    // 50 rows pushed one by one cost 19 ms)
    {
        const uint64_t below = (1ull << lane) - 1ull;
        const uint64_t bm = __ballot(to_mid), bl = __ballot(to_long), bw = __ballot(to_wave), bb = __ballot(to_big);
        if (bb) {
            uint32_t base = 0;
            if (lane == __ffsll((long long)bb) - 1) base = atomicAdd(&p.counters[3], (uint32_t)__popcll(bb));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)bb) - 1);
            if (to_big) p.big_queue[base + (uint32_t)__popcll(bb & below)] = k;
        }
        if (bw) {
            uint32_t base = 0;
            if (lane == __ffsll((long long)bw) - 1) base = atomicAdd(&p.counters[2], (uint32_t)__popcll(bw));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)bw) - 1);
            if (to_wave) p.wave_queue[base + (uint32_t)__popcll(bw & below)] = k;
        }
        if (bm) {
            uint32_t base = 0;
            if (lane == __ffsll((long long)bm) - 1) base = atomicAdd(&p.counters[0], (uint32_t)__popcll(bm));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)bm) - 1);
            if (to_mid) p.mid_queue[base + (uint32_t)__popcll(bm & below)] = k;
        }
        if (bl) {
            uint32_t base = 0;
            if (lane == __ffsll((long long)bl) - 1) base = atomicAdd(&p.counters[1], (uint32_t)__popcll(bl));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)bl) - 1);
            if (to_long) p.long_queue[base + (uint32_t)__popcll(bl & below)] = k;
        }
    }
    // validity words of this wavefront's 64 windows (queued windows set their bit later, atomically)
    const uint64_t m = __ballot(have);
    const int64_t k0 = k - lane;
    if (lane == 0 && k0 < p.W) p.out_valid[k0 >> 5] = (uint32_t)m;
    if (lane == 32 && k0 + 32 < p.W) p.out_valid[(k0 >> 5) + 1] = (uint32_t)(m >> 32);
}

// Windows of 33 .. kModeWave rows: one wavefront per window, four per workgroup, no barrier.  The same order-free form as the
// hash path of mode_mid_kernel (count per key; the smallest LAST row among the keys with the largest count) with a 512-slot
// table in the wavefront's own LDS slice and the scalars reduced by shuffles.  (One workgroup per such window - 60 KB of LDS,
// six barriers - ran 2e6 windows of 50 rows in 19 ms.)
constexpr int kWaveSlots = 512;
__global__ __launch_bounds__(256) void mode_wave_kernel(ModeParams p, const int64_t nq) {
    __shared__ uint64_t keys_all[4][kWaveSlots];
    __shared__ uint32_t cnt_all[4][kWaveSlots];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= nq) return;
    uint64_t *keys = keys_all[wv];
    uint32_t *cnt = cnt_all[wv];
    const int64_t k = p.wave_queue[q];
    int64_t a, b;
    window_rows(p, k, &a, &b);
    const int n = (int)(b - a);   // 33 .. kModeWave
    const bool is_int = p.is_int != 0;
    constexpr uint64_t kEmpty = ~0ull;
    constexpr uint32_t kFlag = 0x80000000u;
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int i = lane; i < kWaveSlots; i += 64) { keys[i] = kEmpty; cnt[i] = 0; }
    wave_sync();
    // the lane's rows: lane, lane + 64, ... (at most four)
    uint64_t key[4];
    bool use[4];
    uint32_t first = 0xFFFFFFFFu, spec = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = lane + 64 * j;
        use[j] = false; key[j] = 0;
        if (r < n && row_valid(p, a + r)) {
            first = first < (uint32_t)r ? first : (uint32_t)r;
            uint64_t v = p.values[a + r];
            bool skip = false;
            if (!is_int) {
                const double x = __longlong_as_double((long long)v);
                if (x != x) skip = true;          // every NaN is a key of its own: count 1
                else if (x == 0.0) v = 0ull;      // -0 == +0
            }
            if (!skip) {
                if (v == kEmpty) spec++;          // Int64 -1 is the table's empty marker: counted apart
                else { use[j] = true; key[j] = v; }
            }
        }
    }
    auto slot_of = [&](uint64_t kk, bool claim) -> int {
        int h = (int)((kk * 0x9E3779B97F4A7C15ull) >> 55);
        for (;;) {
            unsigned long long cur = keys[h];
            if (cur == kk) return h;
            if (cur == kEmpty) {
                if (!claim) return -1;
                cur = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[h]), (unsigned long long)kEmpty, (unsigned long long)kk);
                if (cur == kEmpty || cur == kk) return h;
            }
            h = (h + 1) & (kWaveSlots - 1);
        }
    };
    int slot[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (use[j]) { slot[j] = slot_of(key[j], true); atomicAdd(&cnt[slot[j]], 1u); }
    wave_sync();
    auto wave_max = [&](uint32_t x) { for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_xor((int)x, o); x = y > x ? y : x; } return x; };
    auto wave_min = [&](uint32_t x) { for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_xor((int)x, o); x = y < x ? y : x; } return x; };
    auto wave_sum = [&](uint32_t x) { for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o); return x; };
    uint32_t mx = 0;
    for (int i = lane; i < kWaveSlots; i += 64) mx = cnt[i] > mx ? cnt[i] : mx;
    const uint32_t spec_total = wave_sum(spec);
    uint32_t M = wave_max(mx);
    M = spec_total > M ? spec_total : M;
    first = wave_min(first);
    if (M <= 1) {   // all values distinct, or only NaNs: the first valid row
        if (lane == 0) {
            if (first != 0xFFFFFFFFu) store_result(p, k, p.values[a + first]);
            else p.out_values[k] = 0ull;
        }
        return;
    }
    wave_sync();
    for (int i = lane; i < kWaveSlots; i += 64) cnt[i] = cnt[i] == M ? kFlag : 0u;
    wave_sync();
    uint32_t spec_last = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t r = (uint32_t)(lane + 64 * j);
        if (use[j]) { if (cnt[slot[j]] & kFlag) atomicMax(&cnt[slot[j]], kFlag | r); }
        else if (spec_total == M && (int)r < n && is_int && row_valid(p, a + r) && p.values[a + r] == kEmpty) spec_last = r;  // (rows ascend with j)
    }
    wave_sync();
    uint32_t best = 0xFFFFFFFFu;
    for (int i = lane; i < kWaveSlots; i += 64)
        if (cnt[i] & kFlag) { const uint32_t last = cnt[i] & ~kFlag; best = last < best ? last : best; }
    best = wave_min(best);
    if (spec_total == M) {
        const uint32_t sl = wave_max(spec_last);
        best = sl < best ? sl : best;
    }
    if (lane == 0) store_result(p, k, p.values[a + best]);
}

__global__ __launch_bounds__(256) void mode_mid_kernel(ModeParams p) {
    __shared__ uint64_t sv[kHashSlots + kHashSlots / 2];   // keys, then the 32-bit counts
    const int64_t k = p.mid_queue[blockIdx.x];
    int64_t a, b;
    window_rows(p, k, &a, &b);
    const int n = (int)(b - a);
    const int tid = threadIdx.x;
    const bool is_int = p.is_int != 0;
    {
        // Order-free form of the map walk: with M = the largest count of a (non-NaN) key, the first row at which some count
        // reaches M is the LAST row of a key whose final count is M - the smallest such last row.  So: count every key in an
        // LDS hash table (phase A), flag the slots whose count is M, let every row of a flagged key raise its slot's "last row"
        // (phase B), take the smallest of those (phase C).  M <= 1 (all values distinct, or only NaNs): the first valid row.
        constexpr uint64_t kEmpty = ~0ull;        // a NaN pattern for Float64 keys; Int64 -1 is counted apart ("spec")
        constexpr uint32_t kFlag = 0x80000000u;
        uint64_t *keys = sv;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(sv + kHashSlots);
        int lg = 8;                                  // table size: the power of two >= 2 n, 256 .. kHashSlots (n <= kModeHash fits: load <= 0.63)
        while ((1 << lg) < 2 * n && (1 << lg) < kHashSlots) lg++;
        const int S = 1 << lg;
        __shared__ uint32_t s_first, s_M, s_spec, s_spec_last, s_best;
        for (int i = tid; i < S; i += 256) { keys[i] = kEmpty; cnt[i] = 0; }
        if (tid == 0) { s_first = 0xFFFFFFFFu; s_M = 0; s_spec = 0; s_spec_last = 0; s_best = 0xFFFFFFFFu; }
        __syncthreads();
        auto key_of = [&](uint64_t v, bool *skip) -> uint64_t {
            *skip = false;
            if (!is_int) {
                const double x = __longlong_as_double((long long)v);
                if (x != x) { *skip = true; return 0; }   // every NaN is a key of its own: count 1
                if (x == 0.0) return 0ull;                 // -0 == +0
            }
            return v;
        };
        auto slot_of = [&](uint64_t key, bool claim) -> int {
            int h = (int)((key * 0x9E3779B97F4A7C15ull) >> (64 - lg));
            for (;;) {
                unsigned long long cur = keys[h];
                if (cur == key) return h;
                if (cur == kEmpty) {
                    if (!claim) return -1;
                    cur = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[h]), (unsigned long long)kEmpty, (unsigned long long)key);
                    if (cur == kEmpty || cur == key) return h;
                }
                h = (h + 1) & (S - 1);
            }
        };
        for (int r = tid; r < n; r += 256) {   // phase A
            if (!row_valid(p, a + r)) continue;
            atomicMin(&s_first, (uint32_t)r);
            bool skip;
            const uint64_t key = key_of(p.values[a + r], &skip);
            if (skip) continue;
            if (key == kEmpty) atomicAdd(&s_spec, 1u);
            else atomicAdd(&cnt[slot_of(key, true)], 1u);
        }
        __syncthreads();
        uint32_t mx = 0;
        for (int i = tid; i < S; i += 256) mx = cnt[i] > mx ? cnt[i] : mx;
        if (mx) atomicMax(&s_M, mx);
        if (tid == 0 && s_spec) atomicMax(&s_M, s_spec);
        __syncthreads();
        const uint32_t M = s_M;
        if (M <= 1) {
            if (tid == 0) {
                if (s_first != 0xFFFFFFFFu) store_result(p, k, p.values[a + s_first]);
                else p.out_values[k] = 0ull;
            }
            return;
        }
        for (int i = tid; i < S; i += 256) cnt[i] = cnt[i] == M ? kFlag : 0u;
        const bool spec_flag = s_spec == M;
        __syncthreads();
        for (int r = tid; r < n; r += 256) {   // phase B
            if (!row_valid(p, a + r)) continue;
            bool skip;
            const uint64_t key = key_of(p.values[a + r], &skip);
            if (skip) continue;
            if (key == kEmpty) { if (spec_flag) atomicMax(&s_spec_last, (uint32_t)r); continue; }
            const int h = slot_of(key, false);
            if (cnt[h] & kFlag) atomicMax(&cnt[h], kFlag | (uint32_t)r);
        }
        __syncthreads();
        uint32_t best = 0xFFFFFFFFu;   // phase C
        for (int i = tid; i < S; i += 256)
            if (cnt[i] & kFlag) { const uint32_t last = cnt[i] & ~kFlag; best = last < best ? last : best; }
        if (best != 0xFFFFFFFFu) atomicMin(&s_best, best);
        if (tid == 0 && spec_flag) atomicMin(&s_best, s_spec_last);
        __syncthreads();
        if (tid == 0) store_result(p, k, p.values[a + s_best]);
    }
}

// Windows of kModeHash + 1 .. kModeMid rows (a queue of their own): the same counting with a table of 16384 slots
// that hold the ROW that claimed them instead of the key (4 + 4 bytes per slot: 128 KB of the CU's 160 KB LDS; keys are compared
// through the claiming row's value), so no key needs an "empty" marker.  One workgroup per CU, ~20 us per window - the O(n^2)
// scan it replaces took 2.5 ms.
constexpr int kBigSlots = 16384;

// The counting itself, for a table of 2^lg slots (owner[] = kNone, cnt[] = 0 on entry) that may live in LDS or in global
// memory; s3 = three words of LDS.  Called by all 256 threads of the workgroup.
__device__ __forceinline__ void mode_owner_table(const ModeParams &p, const int64_t k, const int64_t a, const int n, uint32_t *owner,
                                                 uint32_t *cnt, const int lg, uint32_t *s3) {
    constexpr uint32_t kNone = 0xFFFFFFFFu, kFlag = 0x80000000u;
    uint32_t &s_first = s3[0], &s_M = s3[1], &s_best = s3[2];
    const int tid = threadIdx.x, S = 1 << lg;
    const bool is_int = p.is_int != 0;
    if (tid == 0) { s_first = kNone; s_M = 0; s_best = kNone; }
    __syncthreads();
    // slot of the key of row r (value v, not a NaN); claim: take an empty slot for it
    auto slot_of = [&](uint32_t r, uint64_t v, bool claim) -> int {
        uint64_t key = v;
        if (!is_int && __longlong_as_double((long long)v) == 0.0) key = 0ull;   // -0 and +0 hash alike
        int h = (int)((key * 0x9E3779B97F4A7C15ull) >> (64 - lg));
        for (;;) {
            uint32_t o = owner[h];
            if (o == kNone) {
                if (!claim) return -1;
                o = atomicCAS(&owner[h], kNone, r);
                if (o == kNone) return h;
            }
            if (mode_eq(p.values[a + o], v, is_int)) return h;
            h = (h + 1) & (S - 1);
        }
    };
    for (int r = tid; r < n; r += 256) {   // phase A: counts
        if (!row_valid(p, a + r)) continue;
        atomicMin(&s_first, (uint32_t)r);
        const uint64_t v = p.values[a + r];
        if (!is_int) { const double x = __longlong_as_double((long long)v); if (x != x) continue; }   // a NaN: a key of its own, count 1
        atomicAdd(&cnt[slot_of((uint32_t)r, v, true)], 1u);
    }
    __syncthreads();
    uint32_t mx = 0;
    for (int i = tid; i < S; i += 256) mx = cnt[i] > mx ? cnt[i] : mx;
    if (mx) atomicMax(&s_M, mx);
    __syncthreads();
    const uint32_t M = s_M;
    if (M <= 1) {
        if (tid == 0) {
            if (s_first != kNone) store_result(p, k, p.values[a + s_first]);
            else p.out_values[k] = 0ull;
        }
        return;
    }
    for (int i = tid; i < S; i += 256) cnt[i] = cnt[i] == M ? kFlag : 0u;
    __syncthreads();
    for (int r = tid; r < n; r += 256) {   // phase B: the last row of every key that reached M
        if (!row_valid(p, a + r)) continue;
        const uint64_t v = p.values[a + r];
        if (!is_int) { const double x = __longlong_as_double((long long)v); if (x != x) continue; }
        const int h = slot_of((uint32_t)r, v, false);
        if (cnt[h] & kFlag) atomicMax(&cnt[h], kFlag | (uint32_t)r);
    }
    __syncthreads();
    uint32_t best = kNone;   // phase C: the smallest of those
    for (int i = tid; i < S; i += 256)
        if (cnt[i] & kFlag) { const uint32_t last = cnt[i] & ~kFlag; best = last < best ? last : best; }
    if (best != kNone) atomicMin(&s_best, best);
    __syncthreads();
    if (tid == 0) store_result(p, k, p.values[a + s_best]);
}

__global__ __launch_bounds__(256) void mode_big_kernel(ModeParams p) {
    __shared__ uint32_t owner[kBigSlots];
    __shared__ uint32_t cnt[kBigSlots];
    __shared__ uint32_t s3[3];
    const int64_t k = p.big_queue[blockIdx.x];
    int64_t a, b;
    window_rows(p, k, &a, &b);
    const int n = (int)(b - a);
    for (int i = threadIdx.x; i < kBigSlots; i += 256) { owner[i] = 0xFFFFFFFFu; cnt[i] = 0; }
    mode_owner_table(p, k, a, n, owner, cnt, 14, s3);
}

// Windows of kModeMid + 1 .. kModeGlobal rows: the same table in global memory (2^lg >= 2 n slots per window, carved out of
// one allocation by the host, which knows these windows' bounds), one workgroup per window.  Without it every such window went
// through the radix-sort path below, one after the other from the host: 1e4 windows of 1e4 rows took seconds.
struct ModeTable { int64_t k, a, n, off; int32_t lg, _pad; };
__global__ __launch_bounds__(256) void mode_global_kernel(ModeParams p, const ModeTable *tables, uint32_t *owner_all, uint32_t *cnt_all) {
    __shared__ uint32_t s3[3];
    const ModeTable t = tables[blockIdx.x];
    mode_owner_table(p, t.k, t.a, (int)t.n, owner_all + t.off, cnt_all + t.off, t.lg, s3);
}

// ---- long windows: sort path -----------------------------------------------------------------------------------------
struct RowValid {
    const uint32_t *vbits;
    int64_t bit0;  // bit of the window's first row
    __host__ __device__ __forceinline__ bool operator()(uint32_t r) const { return (vbits[(bit0 + r) >> 5] >> ((bit0 + r) & 31)) & 1u; }
};

// keys[q] of the q-th valid row of the window: equal keys <=> equal map keys
__global__ __launch_bounds__(256) void mode_long_keys_kernel(const uint64_t *values, int64_t a, const uint32_t *rows, int64_t m, int is_int,
                                                             uint64_t *keys, uint32_t *rows_out) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < m; q += (int64_t)gridDim.x * 256) {
        const uint32_t r = rows ? rows[q] : (uint32_t)q;
        uint64_t v = values[a + r];
        if (!is_int) {
            const double x = __longlong_as_double((long long)v);
            if (x != x) v = 0xFFF8000000000000ull | (uint64_t)q;  // every NaN is its own key (q < 2^32)
            else if (x == 0.0) v = 0ull;                           // -0 == +0
        }
        keys[q] = v;
        rows_out[q] = r;
    }
}

// from every run head: the run's length (binary search for its end) and the row that completes it
__global__ __launch_bounds__(256) void mode_long_runs_kernel(const uint64_t *keys, const uint32_t *rows, int64_t m, unsigned long long *best) {
    unsigned long long mine = 0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < m; q += (int64_t)gridDim.x * 256) {
        const uint64_t key = keys[q];
        if (q > 0 && keys[q - 1] == key) continue;
        int64_t lo = q + 1, hi = m;  // first index in (q, m] whose key differs
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] == key) lo = mid + 1;
            else hi = mid;
        }
        const unsigned long long cand = ((unsigned long long)(lo - q) << 32) | (unsigned long long)(0xFFFFFFFFu - rows[lo - 1]);
        mine = cand > mine ? cand : mine;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_down(mine, d, 64);
        mine = o > mine ? o : mine;
    }
    if ((threadIdx.x & 63) == 0 && mine) atomicMax(best, mine);
}

__global__ void mode_long_store_kernel(ModeParams p, int64_t k, int64_t a, const unsigned long long *best) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long bv = *best;
    if (bv) store_result(p, k, p.values[a + (int64_t)(0xFFFFFFFFu - (uint32_t)bv)]);
    else p.out_values[k] = 0ull;
}

__global__ void mode_fetch_bounds_kernel(ModeParams p, int64_t nq, int64_t *bounds) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = p.long_queue[q];
        int64_t a, b;
        window_rows(p, k, &a, &b);
        bounds[3 * q] = k; bounds[3 * q + 1] = a; bounds[3 * q + 2] = b;
    }
}

int mode_long_window(Ctx *c, const ModeParams &P, int64_t k, int64_t a, int64_t b) {
    const int64_t n = b - a;
    if (n >= 0xFFFFFFF0ll) return fail(BOWGPU_ERR_UNSUPPORTED, "Mode over a window of %lld rows (at most 2^32 rows per window on the device)", (long long)n);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    DevBuf d_small;
    BG_TRY(d_small.alloc(512));
    int64_t *d_m = reinterpret_cast<int64_t *>(d_small.p);
    unsigned long long *d_best = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(d_small.p) + 256);
    BG_HIP(hipMemsetAsync(d_small.p, 0, 512, c->stream));
    int64_t m = n;
    DevBuf d_rows;
    if (P.vbits) {  // the window's valid rows, ascending
        BG_TRY(d_rows.alloc(up((size_t)n * 4)));
        hipcub::CountingInputIterator<uint32_t> rows_in(0u);
        RowValid pred{P.vbits, P.vbit0 + a};
        size_t tmp_bytes = 0;
        BG_HIP(hipcub::DeviceSelect::If(nullptr, tmp_bytes, rows_in, reinterpret_cast<uint32_t *>(d_rows.p), d_m, n, pred, c->stream));
        DevBuf tmp;
        BG_TRY(tmp.alloc(tmp_bytes + 256));
        BG_HIP(hipcub::DeviceSelect::If(tmp.p, tmp_bytes, rows_in, reinterpret_cast<uint32_t *>(d_rows.p), d_m, n, pred, c->stream));
        BG_HIP(hipMemcpyAsync(&m, d_m, 8, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    if (m > 0) {
        DevBuf d_keys, d_keys2, d_r1, d_r2, tmp;
        BG_TRY(d_keys.alloc(up((size_t)m * 8)));
        BG_TRY(d_keys2.alloc(up((size_t)m * 8)));
        BG_TRY(d_r1.alloc(up((size_t)m * 4)));
        BG_TRY(d_r2.alloc(up((size_t)m * 4)));
        int64_t g = (m + 255) / 256;
        if (g > 256 * 32) g = 256 * 32;
        hipLaunchKernelGGL(mode_long_keys_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, P.values, a,
                           P.vbits ? reinterpret_cast<const uint32_t *>(d_rows.p) : nullptr, m, P.is_int,
                           reinterpret_cast<uint64_t *>(d_keys.p), reinterpret_cast<uint32_t *>(d_r1.p));
        size_t tmp_bytes = 0;
        BG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, reinterpret_cast<const uint64_t *>(d_keys.p), reinterpret_cast<uint64_t *>(d_keys2.p),
                                                  reinterpret_cast<const uint32_t *>(d_r1.p), reinterpret_cast<uint32_t *>(d_r2.p), m, 0, 64, c->stream));
        BG_TRY(tmp.alloc(tmp_bytes + 256));
        BG_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, reinterpret_cast<const uint64_t *>(d_keys.p), reinterpret_cast<uint64_t *>(d_keys2.p),
                                                  reinterpret_cast<const uint32_t *>(d_r1.p), reinterpret_cast<uint32_t *>(d_r2.p), m, 0, 64, c->stream));
        hipLaunchKernelGGL(mode_long_runs_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, reinterpret_cast<const uint64_t *>(d_keys2.p),
                           reinterpret_cast<const uint32_t *>(d_r2.p), m, d_best);
        hipLaunchKernelGGL(mode_long_store_kernel, dim3(1), dim3(64), 0, c->stream, P, k, a, d_best);
        BG_HIP(hipGetLastError());
        BG_HIP(hipStreamSynchronize(c->stream));  // the DevBufs above go back to the cache here
    } else {
        hipLaunchKernelGGL(mode_long_store_kernel, dim3(1), dim3(64), 0, c->stream, P, k, a, d_best);
        BG_HIP(hipGetLastError());
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

}  // namespace

// One Mode output over the windows whose first rows are first_idx[0 .. W] (device).  out_valid must be zeroed words.
int launch_mode(Ctx *c, const int64_t *ts, const int64_t *first_idx, int64_t n, int64_t s0, int64_t interval, int64_t W, int pre_rows,
                int inclusive, const void *values,
                const uint32_t *vbits, int64_t vbit0, int is_int, const bowgpu_agg *agg, void *out_values, uint32_t *out_valid,
                int64_t *n_mid, int64_t *n_long) {
    *n_mid = 0;
    *n_long = 0;
    if (W <= 0) return 0;
    ModeParams P;
    memset(&P, 0, sizeof P);
    P.values = reinterpret_cast<const uint64_t *>(values);
    P.vbits = vbits; P.vbit0 = vbit0;
    P.ts = ts; P.first_idx = first_idx; P.s0 = s0; P.W = W; P.n = n; P.interval = interval;
    P.pre_rows = pre_rows; P.is_int = is_int; P.inclusive = inclusive;
    P.nfac = agg->n_factors;
    for (int f = 0; f < agg->n_factors && f < BOWGPU_MAX_FACTORS; f++) P.fac[f] = agg->factors[f];
    P.out_values = reinterpret_cast<uint64_t *>(out_values);
    P.out_valid = out_valid;
    // queues: a queued window has > kModeSmall rows
    const int64_t qcap = n / (kModeSmall + 1) + 2;
    DevBuf dq;
    BG_TRY(dq.alloc(256 + (size_t)qcap * 32));
    P.counters = reinterpret_cast<uint32_t *>(dq.p);
    P.mid_queue = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(dq.p) + 256);
    P.long_queue = P.mid_queue + qcap;
    P.wave_queue = P.long_queue + qcap;
    P.big_queue = P.wave_queue + qcap;
    BG_HIP(hipMemsetAsync(P.counters, 0, 256, c->stream));
    hipLaunchKernelGGL(mode_small_kernel, dim3((unsigned)((W + kSmallThreads - 1) / kSmallThreads)), dim3(kSmallThreads), 0, c->stream, P);
    BG_HIP(hipGetLastError());
    uint32_t hcount[4] = {0, 0, 0, 0};
    BG_HIP(hipMemcpyAsync(hcount, P.counters, 16, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    *n_mid = (int64_t)hcount[0] + hcount[2] + hcount[3];
    *n_long = hcount[1];
    if (hcount[2] > 0) {
        hipLaunchKernelGGL(mode_wave_kernel, dim3((hcount[2] + 3) / 4), dim3(256), 0, c->stream, P, (int64_t)hcount[2]);
        BG_HIP(hipGetLastError());
    }
    if (hcount[0] > 0) {
        hipLaunchKernelGGL(mode_mid_kernel, dim3(hcount[0]), dim3(256), 0, c->stream, P);
        BG_HIP(hipGetLastError());
    }
    if (hcount[3] > 0) {
        hipLaunchKernelGGL(mode_big_kernel, dim3(hcount[3]), dim3(256), 0, c->stream, P);
        BG_HIP(hipGetLastError());
    }
    if (hcount[1] > 0) {
        DevBuf db;
        BG_TRY(db.alloc((size_t)hcount[1] * 24));
        hipLaunchKernelGGL(mode_fetch_bounds_kernel, dim3(64), dim3(256), 0, c->stream, P, (int64_t)hcount[1], reinterpret_cast<int64_t *>(db.p));
        BG_HIP(hipGetLastError());
        std::vector<int64_t> hb((size_t)hcount[1] * 3);   // (window, first row, end row) of every queued long window
        BG_HIP(hipMemcpyAsync(hb.data(), db.p, hb.size() * 8, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
        const int64_t *bounds = hb.data();
        // windows up to kModeGlobal rows: one launch, tables of 2^lg >= 2 n slots each out of one allocation
        std::vector<ModeTable> tabs;
        int64_t slots = 0;
        for (uint32_t q = 0; q < hcount[1]; q++) {
            const int64_t wn = bounds[3 * q + 2] - bounds[3 * q + 1];
            if (wn > kModeGlobal) continue;
            ModeTable t;
            t.k = bounds[3 * q]; t.a = bounds[3 * q + 1]; t.n = wn; t.off = slots; t._pad = 0;
            t.lg = 14;
            while ((1ll << t.lg) < 2 * wn) t.lg++;
            slots += 1ll << t.lg;
            tabs.push_back(t);
        }
        if (!tabs.empty()) {
            DevBuf d_tabs, d_owner, d_cnt;
            BG_TRY(d_tabs.alloc(tabs.size() * sizeof(ModeTable)));
            BG_TRY(d_owner.alloc((size_t)slots * 4));
            BG_TRY(d_cnt.alloc((size_t)slots * 4));
            BG_HIP(hipMemcpyAsync(d_tabs.p, tabs.data(), tabs.size() * sizeof(ModeTable), hipMemcpyHostToDevice, c->stream));
            BG_HIP(hipMemsetAsync(d_owner.p, 0xFF, (size_t)slots * 4, c->stream));
            BG_HIP(hipMemsetAsync(d_cnt.p, 0, (size_t)slots * 4, c->stream));
            hipLaunchKernelGGL(mode_global_kernel, dim3((unsigned)tabs.size()), dim3(256), 0, c->stream, P,
                               reinterpret_cast<const ModeTable *>(d_tabs.p), reinterpret_cast<uint32_t *>(d_owner.p),
                               reinterpret_cast<uint32_t *>(d_cnt.p));
            BG_HIP(hipGetLastError());
            BG_HIP(hipStreamSynchronize(c->stream));   // (tabs and the DevBufs go out of scope)
        }
        for (uint32_t q = 0; q < hcount[1]; q++)
            if (bounds[3 * q + 2] - bounds[3 * q + 1] > kModeGlobal) BG_TRY(mode_long_window(c, P, bounds[3 * q], bounds[3 * q + 1], bounds[3 * q + 2]));
    }
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace bowgpu
