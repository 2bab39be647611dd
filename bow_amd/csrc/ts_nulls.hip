// ts_nulls.hip — the device path for an interval column WITH NULLS (Rolling.Aggregate, exclusive windows).
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
// per bitmap; the reference's Go loop runs at ~1e7 rows/s.  The only reducer this cannot serve is NumRows (the test closure of
// aggregation_test.go:28-31 counts rows, valid or not): such a call is declined.
#include "bitmap_device.h"

namespace bowgpu {

namespace {

__device__ __forceinline__ uint64_t wid_of(int64_t t, int64_t s0, const MagicDiv &magic) {
    return t < s0 ? 0ull : magic_div((uint64_t)t - (uint64_t)s0, magic);   // rows below s0 ride in window 0 (rolling.go:194-196)
}

// one lane per row, 64 rows per wavefront: the keep bits of a wavefront's rows are one 64-bit word of the bitmap (bit 0 = row 0)
__global__ __launch_bounds__(256) void ts_nullfill_kernel(const int64_t *__restrict__ ts, const uint32_t *__restrict__ tbits, const int64_t tbit0,
                                                          const int64_t n, const NbrIndex ix, const int64_t s0, const MagicDiv magic,
                                                          int64_t *__restrict__ ts_eff, uint64_t *__restrict__ keep, unsigned long long *n_dropped) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool k = false, dropped = false;
    if (i < n) {
        if (bit_at(tbits, tbit0, i)) {
            ts_eff[i] = ts[i];
            k = true;
        } else {
            const int64_t p = prev_valid_ix(tbits, tbit0, n, i - 1, ix);   // (row 0 is valid: the constructor checked it, rolling.go:89-93)
            const int64_t q = next_valid_ix(tbits, tbit0, n, i + 1, ix);
            const int64_t tp = p >= 0 ? ts[p] : s0;
            ts_eff[i] = tp;
            k = p >= 0 && q >= 0 && wid_of(tp, s0, magic) == wid_of(ts[q], s0, magic);
            dropped = !k;
        }
    }
    const unsigned long long m = __ballot(k), d = __ballot(dropped);
    if ((threadIdx.x & 63) == 0) {
        if (i < n) keep[i >> 6] = m;
        if (d) atomicAdd(n_dropped, (unsigned long long)__popcll(d));
    }
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

}  // namespace

int launch_ts_nullfill(Ctx *c, const int64_t *ts, const uint32_t *tbits, int64_t tbit0, int64_t n, const NbrIndex &ix, int64_t s0,
                       const MagicDiv &magic, int64_t *ts_eff, uint64_t *keep, unsigned long long *d_dropped) {
    if (n <= 0) return 0;
    BG_HIP(hipMemsetAsync(d_dropped, 0, 8, c->stream));
    hipLaunchKernelGGL(ts_nullfill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ts, tbits, tbit0, n, ix, s0, magic, ts_eff, keep,
                       d_dropped);
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

}  // namespace bowgpu
