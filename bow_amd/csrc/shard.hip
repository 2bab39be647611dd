// shard.hip — the boundary-window stitch of the row-range sharded Rolling.Aggregate.
//
// A rank reduces its own rows with the ordinary tile kernels.  Only the window that straddles a
// shard boundary needs more: the left rank exports the RUNNING STATE of its last window
// (range_state_kernel, mode 0), the ranks exchange those fixed-size records (one RCCL all_gather of
// bytes, done by the caller), and the right rank re-walks its own rows of that window seeded with
// the left state (mode 1) - i.e. the straddling window is folded in the reference's row order
// (sum.go:16-22, arithmeticmean.go:17-24, minmax.go:16-28), bit-exact, with no collective on the data.
#include "agg_device.h"

namespace bowgpu {

namespace {

__device__ __forceinline__ void carry_to_stats(const bowgpu_carry_state &c, Stats &s) {
    stats_init(s);
    s.sum = c.sum; s.vmin = c.vmin; s.vmax = c.vmax; s.nn_min = c.nn_min; s.nn_max = c.nn_max;
    s.first_bits = c.first_bits; s.last_bits = c.last_bits; s.count = c.count;
    s.has_value = c.has_value; s.has_nn = c.has_nn;
    s.pt = c.pt; s.pv = c.pv; s.first_pt = c.first_pt; s.first_pv = c.first_pv;
    s.integ_step = c.integ_step; s.integ_trap = c.integ_trap; s.has_point = c.has_point; s.has_pair = c.has_pair;
}
__device__ __forceinline__ void stats_to_carry(const Stats &s, int64_t nrows, bowgpu_carry_state &c) {
    c.sum = s.sum; c.vmin = s.vmin; c.vmax = s.vmax; c.nn_min = s.nn_min; c.nn_max = s.nn_max;
    c.first_bits = s.first_bits; c.last_bits = s.last_bits; c.count = s.count; c.nrows = nrows;
    c.has_value = s.has_value; c.has_nn = s.has_nn;
    c.pt = s.pt; c.pv = s.pv; c.first_pt = s.first_pt; c.first_pv = s.first_pv;
    c.integ_step = s.integ_step; c.integ_trap = s.integ_trap; c.has_point = s.has_point; c.has_pair = s.has_pair;
}

constexpr int kSeqLimit = 8192;  // longer ranges are merged from 256 contiguous partials (order-changing for Sum)

}  // namespace

// mode 0: state of window `wid` over rows [lower_bound(ts, start(wid)), n)        -> states_out
// mode 1: seeds + rows [0, lower_bound(ts, start(wid+1))) of window `wid` -> outputs slot + states_out
// mode 2: like mode 0, and the outputs of slot `wid` are (re)written: the shard owns its last window and only now learns
//         whether the next shard's first row is that window's inclusive row
// next: the first row of the next non-empty shard to the right (nullable).  The window's inclusive row (rolling.go:201-209) is
// the local row right after its rows when that row sits exactly on the window's end, else `next` when ITS timestamp does.
// seed_alive: some row behind the seeds has ts >= s0.  Window 0 also spans the rows BELOW s0 (negative timestamps), but it is an
// empty slice unless one of its rows reaches s0 or it takes an inclusive row (rolling.go:194-228: lastRowIndex stays -1) - the
// "dead window 0" rule of the tile kernels, here for the window as stitched across shards.
// strict: bowgpu_options.strict_order - the rows are walked by ONE lane in row order whatever their number (the 256 contiguous partials
// of a long range change the order of a Sum's additions); a range of more than 2^20 rows raises status[7] and is left alone (the
// stated limit of strict_order: api.cpp turns it into BOWGPU_ERR_UNSUPPORTED)
__global__ __launch_bounds__(256) void range_state_kernel(const AggParams p, const int mode, const uint64_t wid,
                                                          const bowgpu_carry_state *seeds,
                                                          bowgpu_carry_state *states_out, const bowgpu_next_row *next, const int seed_alive,
                                                          const int strict) {
    __shared__ Stats part[256];
    __shared__ int64_t s_r0, s_r1;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
        const int64_t lim = mode != 1 ? win_start : win_start + p.interval;
        const bool ovf = mode == 1 && lim < win_start;
        int64_t lo = 0, hi = p.n;
        while (lo < hi && !ovf) {  // first row with ts >= lim
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (p.ts[mid] >= lim) hi = mid; else lo = mid + 1;
        }
        const int64_t b = ovf ? p.n : lo;
        // window 0 also holds the rows below s0 (Go's truncating division puts s0 above a negative first timestamp:
        // rolling.go:96-99, :194-196), so its state starts at row 0
        s_r0 = mode != 1 ? (wid == 0 ? 0 : b) : 0;
        s_r1 = mode != 1 ? p.n : b;
    }
    __syncthreads();
    const int64_t r0 = s_r0, r1 = s_r1;
    const int64_t len = r1 - r0;
    if (strict && len > (1ll << 20)) {
        if (tid == 0) atomicOr(&p.status[7], 1u);
        return;
    }
    const int64_t win_start = p.s0 + (int64_t)(wid * (uint64_t)p.interval);
    const int64_t win_end = win_start + p.interval;
    const int64_t oslot = (int64_t)(wid - (uint64_t)p.wid_base);
    // where the inclusive row comes from: 1 = local row r1, 2 = the next shard's first row, 0 = there is none
    int incl_src = 0;
    if (p.inclusive && mode != 0 && win_end > win_start) {
        if (r1 < p.n) incl_src = p.ts[r1] == win_end ? 1 : 0;
        else if (next && next->present && next->ts == win_end) incl_src = 2;
    }

    for (int a = 0; a < p.naggs; a++) {
        const AggDesc &ad = p.aggs[a];
        const bool with_ts = kind_is_integral(ad.kind);
        const bool reads = !(ad.kind == BOWGPU_AGG_WINDOW_START || ad.kind == BOWGPU_AGG_NUM_ROWS);
        const ColDesc *cd = (reads && ad.slot >= 0) ? &p.cols[ad.slot] : nullptr;
        const int col_type = cd ? cd->type : BOWGPU_INT64;
        Stats acc;
        stats_init(acc);
        int64_t seed_rows = 0;
        if (seeds) { carry_to_stats(seeds[a], acc); seed_rows = seeds[a].nrows; }
        if (cd) {
            const uint64_t *vp = reinterpret_cast<const uint64_t *>(cd->values);
            auto valid_at = [&](int64_t r) -> bool {
                if (!cd->vbits) return true;
                const int64_t bit = cd->vbit0 + r;
                return (cd->vbits[bit >> 5] >> (bit & 31)) & 1u;
            };
            if (len <= kSeqLimit || strict) {
                if (tid == 0)
                    for (int64_t r = r0; r < r1; r++)
                        if (valid_at(r)) {
                            const uint64_t raw = vp[r];
                            const double x = bits_to_f64(raw, col_type);
                            stats_value<true>(acc, x, raw);
                            if (with_ts) stats_point(acc, (double)p.ts[r], x);
                        }
            } else {
                Stats st;
                stats_init(st);
                const int64_t lo_r = r0 + (len * tid) / 256, hi_r = r0 + (len * (tid + 1)) / 256;
                for (int64_t r = lo_r; r < hi_r; r++)
                    if (valid_at(r)) {
                        const uint64_t raw = vp[r];
                        const double x = bits_to_f64(raw, col_type);
                        stats_value<true>(st, x, raw);
                        if (with_ts) stats_point(st, (double)p.ts[r], x);
                    }
                __syncthreads();
                part[tid] = st;
                __syncthreads();
                if (tid == 0)
                    for (int t = 0; t < 256; t++) stats_merge(acc, part[t]);
            }
        }
        if (tid == 0) {
            int64_t nrows = seed_rows + len;
            if (states_out) stats_to_carry(acc, nrows, states_out[a]);
            if (mode != 0 && (uint64_t)oslot < (uint64_t)p.W) {
                if (wid == 0 && !incl_src && !seed_alive && !(len > 0 && p.ts[r1 - 1] >= p.s0)) {   // dead window 0: an empty slice
                    stats_init(acc);
                    nrows = 0;
                }
                int64_t nrows_seen = nrows;
                if (incl_src && kind_needs_inclusive(ad.kind)) {  // the reducers that declared NeedInclusiveWindow see one more row
                    nrows_seen = nrows + 1;
                    bool ok;
                    uint64_t raw;
                    double t;
                    if (incl_src == 1) {
                        ok = cd != nullptr;
                        if (ok && cd->vbits) { const int64_t bit = cd->vbit0 + r1; ok = (cd->vbits[bit >> 5] >> (bit & 31)) & 1u; }
                        raw = ok ? reinterpret_cast<const uint64_t *>(cd->values)[r1] : 0;
                        t = (double)p.ts[r1];
                    } else {
                        ok = cd != nullptr && next->valid[a] != 0;
                        raw = next->bits[a];
                        t = (double)next->ts;
                    }
                    if (ok) {
                        const double x = bits_to_f64(raw, col_type);
                        stats_value<true>(acc, x, raw);
                        stats_point(acc, t, x);
                    }
                }
                Val v = finish_val(reduce_val(ad.kind, acc, nrows_seen, win_start, p.interval, col_type == BOWGPU_INT64), ad);
                reinterpret_cast<uint64_t *>(ad.out_values)[oslot] = v.bits;
                if (ad.out_valid) {
                    const uint32_t bit = 1u << (oslot & 31);
                    if (v.valid) atomicOr(&ad.out_valid[oslot >> 5], bit);
                    else atomicAnd(&ad.out_valid[oslot >> 5], ~bit);
                }
            }
        }
        __syncthreads();
    }
}

// the empty windows [slot0, slot1) of every reducer (A.9 "Empty slice"); nullable bits end up 0
__global__ __launch_bounds__(256) void fill_empty_kernel(const AggParams p, const int64_t slot0, const int64_t slot1) {
    Stats e;
    stats_init(e);
    for (int64_t s = slot0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < slot1; s += (int64_t)gridDim.x * blockDim.x) {
        if (s < 0 || s >= p.W) continue;
        const int64_t win_start = p.s0 + (int64_t)(((uint64_t)p.wid_base + (uint64_t)s) * (uint64_t)p.interval);
        for (int a = 0; a < p.naggs; a++) {
            const AggDesc &ad = p.aggs[a];
            const int ct = ad.slot >= 0 ? p.cols[ad.slot].type : BOWGPU_INT64;
            Val v = finish_val(reduce_val(ad.kind, e, 0, win_start, p.interval, ct == BOWGPU_INT64), ad);
            reinterpret_cast<uint64_t *>(ad.out_values)[s] = v.bits;
            if (p.bits_preset && ad.out_valid && !v.valid) atomicAnd(&ad.out_valid[s >> 5], ~(1u << (s & 31)));
        }
    }
}

int launch_fill_empty(Ctx *c, const AggParams &p, int64_t slot0, int64_t slot1) {
    if (slot1 <= slot0) return 0;
    int64_t grid = (slot1 - slot0 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, p, slot0, slot1);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_range_state(Ctx *c, const AggParams &p, int mode, uint64_t wid, const bowgpu_carry_state *d_seeds,
                       bowgpu_carry_state *d_states_out, const bowgpu_next_row *d_next, int seed_alive, int strict) {
    hipLaunchKernelGGL(range_state_kernel, dim3(1), dim3(256), 0, c->stream, p, mode, wid, d_seeds, d_states_out, d_next, seed_alive, strict);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu
