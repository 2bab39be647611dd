// interp_fill.hip — the callers either side of Rolling.Aggregate:
//   * Rolling.Interpolate with the built-in interpolators (reference rolling/interpolation.go:30-161,
//     rolling/interpolation/{windowstart,linear,stepprevious,none}.go),
//   * Bow.FillLinear (reference bowfill.go:14-103) and Bow.IsColSorted (bowassertion.go:15-81),
//   * the whole-frame aggregation.Aggregate (reference rolling/aggregation/whole.go:12-93).
// All are streaming, HBM-bound passes over Arrow value / validity buffers; none is a contraction.
#include "bitmap_device.h"

namespace bowgpu {


// ---- neighbour index (common.h NbrIndex)
// one wavefront per block of 4096 bits = 128 words: last / first valid row inside the block
__global__ __launch_bounds__(256) void nbr_block_kernel(const uint32_t *vbits, int64_t vbit0, int64_t n, int64_t g0, int64_t nblocks,
                                                        int64_t *last_in, int64_t *first_in) {
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nblocks) return;
    const int lane = threadIdx.x & 63;
    const int64_t wbase = (g0 + g) * (kNbrBlockBits / 32);
    const int64_t bit_lo = vbit0, bit_hi = vbit0 + n;  // the column's bits
    int64_t hi = -1, lo = INT64_MAX;
    for (int k = 0; k < 2; k++) {
        const int64_t w = wbase + lane + 64 * k;
        const int64_t b0 = w << 5;
        if (b0 + 32 <= bit_lo || b0 >= bit_hi) continue;
        uint32_t x = vbits[w];
        if (b0 < bit_lo) x &= ~0u << (bit_lo - b0);
        if (b0 + 32 > bit_hi) x &= (1u << (bit_hi - b0)) - 1u;
        if (!x) continue;
        const int64_t h = b0 + (31 - __clz((int)x)) - vbit0, l = b0 + (__ffs((int)x) - 1) - vbit0;
        if (h > hi) hi = h;
        if (l < lo) lo = l;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t h2 = __shfl_down((long long)hi, o), l2 = __shfl_down((long long)lo, o);
        if (h2 > hi) hi = h2;
        if (l2 < lo) lo = l2;
    }
    if (lane == 0) { last_in[g] = hi; first_in[g] = lo; }
}

// exclusive running max of last_in (-> prev_before) and exclusive reverse running min of first_in (-> next_after):
// one workgroup, each thread owns a contiguous run of blocks
__global__ __launch_bounds__(1024) void nbr_scan_kernel(const int64_t *last_in, const int64_t *first_in, int64_t nblocks,
                                                        int64_t *prev_before, int64_t *next_after) {
    __shared__ int64_t s_hi[1024], s_lo[1024];
    const int t = threadIdx.x;
    const int64_t per = (nblocks + 1023) / 1024;
    const int64_t a = t * per, b = a + per < nblocks ? a + per : nblocks;
    int64_t hi = -1, lo = INT64_MAX;
    for (int64_t g = a; g < b; g++) { if (last_in[g] > hi) hi = last_in[g]; if (first_in[g] < lo) lo = first_in[g]; }
    s_hi[t] = hi; s_lo[t] = lo;
    __syncthreads();
    if (t == 0) {  // 1024 entries: serial is cheap
        int64_t run = -1;
        for (int i = 0; i < 1024; i++) { const int64_t x = s_hi[i]; s_hi[i] = run; if (x > run) run = x; }
        run = INT64_MAX;
        for (int i = 1023; i >= 0; i--) { const int64_t x = s_lo[i]; s_lo[i] = run; if (x < run) run = x; }
    }
    __syncthreads();
    int64_t run = s_hi[t];
    for (int64_t g = a; g < b; g++) { prev_before[g] = run; if (last_in[g] > run) run = last_in[g]; }
    run = s_lo[t];
    for (int64_t g = b - 1; g >= a; g--) { next_after[g] = run == INT64_MAX ? -1 : run; if (first_in[g] < run) run = first_in[g]; }
}


// ------------------------------------------------------------------ Interpolate
// first row of every window: first_idx[k] = lower_bound(ts, s_k) for k in [0, W], first_idx[W] = n
__global__ __launch_bounds__(256) void window_first_rows_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval,
                                                                MagicDiv magic, int64_t W, int64_t *first_idx,
                                                                uint32_t *status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
        if (i == n) {  // windows after the last row's window do not exist (W = wid(last)+1); terminator
            first_idx[W] = n;
            continue;
        }
        const int64_t t = ts[i];
        const uint64_t w = t < s0 ? 0 : magic_div((uint64_t)t - (uint64_t)s0, magic);
        uint64_t wp;
        if (i == 0) {
            first_idx[0] = 0;
            wp = 0;
        } else {
            const int64_t tp = ts[i - 1];
            if (tp > t) atomicOr(&status[0], 1u);
            wp = tp < s0 ? 0 : magic_div((uint64_t)tp - (uint64_t)s0, magic);
        }
        for (uint64_t k = wp + 1; k <= w && (int64_t)k < W; k++) first_idx[k] = i;  // k's first row (and every empty window before it)
    }
}

// three-kernel exclusive scan of int32 flags into int64 positions
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t *in, int64_t n, int64_t *block_sums) {
    __shared__ long long sh[4];
    const int64_t base = (int64_t)blockIdx.x * 2048;
    long long acc = 0;
    for (int j = 0; j < 8; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) acc += in[i];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void scan_sums_kernel(int64_t *block_sums, int64_t nblocks, int64_t *total) {
    // single thread: nblocks = W/2048 (<= ~50 k for W = 1e8)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int64_t i = 0; i < nblocks; i++) { const int64_t v = block_sums[i]; block_sums[i] = run; run += v; }
        *total = run;
    }
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t *in, int64_t n, const int64_t *block_sums, int64_t *out) {
    __shared__ long long sh[256];
    const int64_t base = (int64_t)blockIdx.x * 2048 + (int64_t)threadIdx.x * 8;
    long long loc[8], acc = 0;
    for (int j = 0; j < 8; j++) { loc[j] = acc; if (base + j < n) acc += in[base + j]; }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { long long run = block_sums[blockIdx.x]; for (int t = 0; t < 256; t++) { const long long v = sh[t]; sh[t] = run; run += v; } }
    __syncthreads();
    const long long off = sh[threadIdx.x];
    for (int j = 0; j < 8; j++) if (base + j < n) out[base + j] = off + loc[j];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = off + acc;  // out has n+1 entries
}

// ------------------------------------------------------------------ IsColSorted / FillLinear
// flags[0] |= 1 if some consecutive valid pair increases, |= 2 if some decreases, |= 4 if any valid value exists
__global__ __launch_bounds__(256) void col_order_kernel(const uint64_t *values, const uint32_t *vbits, int64_t vbit0, int64_t n,
                                                        int32_t type, uint32_t *flags) {
    uint32_t f = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!bit_at(vbits, vbit0, i)) continue;
        f |= 4;
        const int64_t pi = prev_valid(vbits, vbit0, n, i - 1);
        if (pi < 0) continue;
        if (type == BOWGPU_INT64) {
            const int64_t c = (int64_t)values[pi], x = (int64_t)values[i];
            if (c < x) f |= 1; else if (c > x) f |= 2;
        } else {
            const double c = __longlong_as_double((long long)values[pi]), x = __longlong_as_double((long long)values[i]);
            if (c < x) f |= 1; else if (c > x) f |= 2;  // NaN compares false both ways (bowassertion.go:64-74)
        }
    }
    if (f) atomicOr(flags, f);
}

// the value a null row i receives (bits, valid) under p.method
__device__ __forceinline__ void fill_one(const FillParams &p, int64_t i, uint64_t *bits_io, int *valid_io) {
    uint64_t bits = *bits_io;
    int valid = 0;
    const int64_t rp = p.method == BOWGPU_FILL_NEXT ? -1 : prev_valid_ix(p.fill_vbits, p.fill_vbit0, p.n, i - 1, p.nbr);
    const int64_t rn = p.method == BOWGPU_FILL_PREVIOUS ? -1 : next_valid_ix(p.fill_vbits, p.fill_vbit0, p.n, i + 1, p.nbr);
    if (p.method == BOWGPU_FILL_PREVIOUS) {
        if (rp >= 0) { bits = p.fill_values[rp]; valid = 1; }          // arr.Value(fillRowIndex): bowfill.go:196-199
    } else if (p.method == BOWGPU_FILL_NEXT) {
        if (rn >= 0) { bits = p.fill_values[rn]; valid = 1; }
    } else if (p.method == BOWGPU_FILL_MEAN) {
        if (rp >= 0 && rn >= 0) {                                       // bowfill.go:145-154
            const double m = (bits_to_f64(p.fill_values[rp], p.fill_type) + bits_to_f64(p.fill_values[rn], p.fill_type)) / 2;
            bits = p.fill_type == BOWGPU_INT64 ? (uint64_t)go_f64_to_i64(round(m)) : (uint64_t)__double_as_longlong(m);
            valid = 1;
        }
    } else {                                                            // FillLinear: bowfill.go:65-97
        const bool v1 = bit_at(p.ref_vbits, p.ref_vbit0, i);
        const bool v2 = rp >= 0 && bit_at(p.ref_vbits, p.ref_vbit0, rp);   // GetFloat64(ref, -1) => (0,false) :72
        const bool v3 = rn >= 0 && bit_at(p.ref_vbits, p.ref_vbit0, rn);
        if (v1 && v2 && v3) {
            const double prev_fill = bits_to_f64(p.fill_values[rp], p.fill_type);
            const double next_fill = bits_to_f64(p.fill_values[rn], p.fill_type);
            const double row_ref = bits_to_f64(p.ref_values[i], p.ref_type);
            const double prev_ref = bits_to_f64(p.ref_values[rp], p.ref_type);
            const double next_ref = bits_to_f64(p.ref_values[rn], p.ref_type);
            // (the nextRef-prevRef == 0 branch of :78-85 is overwritten by the fall-through below)
            double tmp = row_ref - prev_ref;   // :87-90, four separate statements
            tmp /= next_ref - prev_ref;
            tmp *= next_fill - prev_fill;
            tmp += prev_fill;
            if (p.fill_type == BOWGPU_INT64) bits = (uint64_t)go_f64_to_i64(round(tmp));  // math.Round: half away from zero :93
            else bits = (uint64_t)__double_as_longlong(tmp);
            valid = 1;
        }
    }
    *bits_io = bits;
    *valid_io = valid;
}

// Bow.FillLinear (bowfill.go:14-103), FillPrevious / FillNext (:162-253), FillMean (:105-160).  A wavefront owns 512 consecutive
// rows per trip as four chunks of 128: lane l holds rows 2l, 2l+1 of each chunk (one 16-B load and one 16-B store per chunk,
// all four loads in flight at once); null rows look their neighbours up through the index; the validity bits of a chunk
// leave as two aligned 64-bit words (the flags hop to the lane whose number is the row number, then one ballot per word).
__global__ __launch_bounds__(256) void fill_kernel(const FillParams p) {
    unsigned long long nvalid = 0;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.fill_values) | reinterpret_cast<uintptr_t>(p.out_values)) & 15) == 0;
    for (int64_t base = wave * 512; base < p.n; base += nwaves * 512) {
        uint64_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) load_pair(p.fill_values, base + 128 * k + 2 * lane, p.n, vec, a[k], b[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t i = base + 128 * k + 2 * lane;
            int va = (i < p.n && bit_at(p.fill_vbits, p.fill_vbit0, i)) ? 1 : 0;
            int vb = (i + 1 < p.n && bit_at(p.fill_vbits, p.fill_vbit0, i + 1)) ? 1 : 0;
            if (i < p.n && !va) fill_one(p, i, &a[k], &va);
            if (i + 1 < p.n && !vb) fill_one(p, i + 1, &b[k], &vb);
            if (vec && i + 1 < p.n) *reinterpret_cast<ulonglong2 *>(p.out_values + i) = make_ulonglong2(a[k], b[k]);
            else {
                if (i < p.n) p.out_values[i] = a[k];
                if (i + 1 < p.n) p.out_values[i + 1] = b[k];
            }
            const int f = va | (vb << 1);
            const int lo = __shfl(f, lane >> 1), hi = __shfl(f, 32 + (lane >> 1));   // rows lane and 64 + lane of the chunk
            const unsigned long long w0 = __ballot((lo >> (lane & 1)) & 1), w1 = __ballot((hi >> (lane & 1)) & 1);
            if (lane == 0 && base + 128 * k < p.n) {
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.out_valid_words + ((base + 128 * k) >> 5));
                dst[0] = w0;
                if (base + 128 * k + 64 < p.n) dst[1] = w1;
                nvalid += __popcll(w0) + __popcll(w1);
            }
        }
    }
    if (lane == 0 && nvalid) atomicAdd(p.valid_count, nvalid);
}

// ------------------------------------------------------------------ whole-frame aggregation
// Level 1: workgroup b reduces rows [b*chunk, (b+1)*chunk) of one column into a partial state; each of its four wavefronts
// owns a contiguous quarter and steps through it 512 rows at a time - lane l holds rows 8l..8l+7 of the step (four 16-B loads) -
// merging the lanes' states with an ORDER-PRESERVING shuffle tree (stats_merge is concatenation: First / Last, the NaN-seed
// rule of Min / Max and the integrals' adjacency survive), then the step into the wavefront's running state.  Level 2: one
// thread merges the workgroup partials in order.  Fixed shape => deterministic; Sum / Mean / Integral are not in strict
// row order (1e-12 rel).
struct WholeParams {
    const int64_t *ts;
    const uint64_t *values;
    const uint32_t *vbits;
    int64_t vbit0;
    int64_t n;
    int32_t type;
    int32_t need_ts;
    Stats *partials;
    int64_t chunk;
};

__device__ __forceinline__ Stats stats_shfl_down(const Stats &s, int o) {
    Stats r;
    r.sum = __shfl_down(s.sum, o); r.vmin = __shfl_down(s.vmin, o); r.vmax = __shfl_down(s.vmax, o);
    r.nn_min = __shfl_down(s.nn_min, o); r.nn_max = __shfl_down(s.nn_max, o);
    r.first_bits = __shfl_down((unsigned long long)s.first_bits, o); r.last_bits = __shfl_down((unsigned long long)s.last_bits, o);
    r.count = __shfl_down((long long)s.count, o);
    r.pt = __shfl_down(s.pt, o); r.pv = __shfl_down(s.pv, o); r.first_pt = __shfl_down(s.first_pt, o); r.first_pv = __shfl_down(s.first_pv, o);
    r.integ_step = __shfl_down(s.integ_step, o); r.integ_trap = __shfl_down(s.integ_trap, o);
    r.has_value = __shfl_down(s.has_value, o); r.has_nn = __shfl_down(s.has_nn, o);
    r.has_point = __shfl_down(s.has_point, o); r.has_pair = __shfl_down(s.has_pair, o);
    return r;
}

__global__ __launch_bounds__(256) void whole_partial_kernel(const WholeParams p) {
    __shared__ Stats part[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.x * p.chunk;
    int64_t hi = lo + p.chunk;
    if (hi > p.n) hi = p.n;
    // the wavefront's quarter, in whole steps of 512 rows
    const int64_t steps = hi > lo ? (hi - lo + 511) / 512 : 0;
    const int64_t s_per = (steps + 3) / 4;
    const int64_t q_lo = lo + (int64_t)wv * s_per * 512;
    int64_t q_hi = q_lo + s_per * 512;
    if (q_hi > hi) q_hi = hi;
    const bool vec = (reinterpret_cast<uintptr_t>(p.values) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.ts) & 15) == 0;
    Stats running;
    stats_init(running);
    for (int64_t base = q_lo; base < q_hi; base += 512) {
        const int64_t r0 = base + 8 * lane;
        uint64_t v[8], t[8];
#pragma unroll
        for (int k = 0; k < 4; k++) load_pair(p.values, r0 + 2 * k, q_hi, vec && (r0 & 1) == 0, v[2 * k], v[2 * k + 1]);
        if (p.need_ts) {
#pragma unroll
            for (int k = 0; k < 4; k++) load_pair(reinterpret_cast<const uint64_t *>(p.ts), r0 + 2 * k, q_hi, vec && (r0 & 1) == 0, t[2 * k], t[2 * k + 1]);
        }
        Stats st;
        stats_init(st);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t r = r0 + k;
            if (r >= q_hi || !bit_at(p.vbits, p.vbit0, r)) continue;
            const double x = bits_to_f64(v[k], p.type);
            stats_value<true>(st, x, v[k]);
            if (p.need_ts) stats_point(st, (double)(int64_t)t[k], x);
        }
        for (int o = 1; o < 64; o <<= 1) {  // lane i <- merge(lane i, lane i + o): contiguous row ranges, left then right
            const Stats other = stats_shfl_down(st, o);
            if ((lane & (2 * o - 1)) == 0) stats_merge(st, other);
        }
        if (lane == 0) stats_merge(running, st);
    }
    if (lane == 0) part[wv] = running;
    __syncthreads();
    if (threadIdx.x == 0) {
        Stats acc = part[0];
        stats_merge(acc, part[1]); stats_merge(acc, part[2]); stats_merge(acc, part[3]);
        p.partials[blockIdx.x] = acc;
    }
}

// Level 2: the workgroup partials of one column -> one state, in order: thread t merges a contiguous run of them, then the
// same order-preserving tree across lanes and the four wavefronts.  merged[0] receives the result.
__global__ __launch_bounds__(256) void whole_merge_kernel(const Stats *partials, int64_t nblocks, Stats *merged) {
    __shared__ Stats part[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t per = (nblocks + 255) / 256;
    const int64_t a = (int64_t)threadIdx.x * per;
    int64_t b = a + per;
    if (b > nblocks) b = nblocks;
    Stats st;
    stats_init(st);
    for (int64_t i = a; i < b; i++) stats_merge(st, partials[i]);
    for (int o = 1; o < 64; o <<= 1) {
        const Stats other = stats_shfl_down(st, o);
        if ((lane & (2 * o - 1)) == 0) stats_merge(st, other);
    }
    if (lane == 0) part[wv] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        Stats acc = part[0];
        stats_merge(acc, part[1]); stats_merge(acc, part[2]); stats_merge(acc, part[3]);
        merged[0] = acc;
    }
}

struct WholeFinal {
    int32_t kind, out_type, col_is_int, n_factors;
    double factors[BOWGPU_MAX_FACTORS];
    uint64_t *out_value;
    uint8_t *out_valid_byte;
};

__global__ void whole_final_kernel(const Stats *partials, int64_t nblocks, int64_t nrows, int64_t first_value,
                                   int64_t last_value, WholeFinal f) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Stats acc;
    stats_init(acc);
    for (int64_t i = 0; i < nblocks; i++) stats_merge(acc, partials[i]);
    // reduce_val expects (win_start, interval) with LastValue = win_start + interval (whole.go:64-71)
    Val v = reduce_val(f.kind, acc, nrows, first_value, last_value - first_value, f.col_is_int);
    if (v.valid) {
        for (int k = 0; k < f.n_factors; k++) {
            if (v.is_int) v.bits = (uint64_t)go_f64_to_i64((double)(int64_t)v.bits * f.factors[k]);
            else v.bits = (uint64_t)__double_as_longlong(__longlong_as_double((long long)v.bits) * f.factors[k]);
        }
        // SetOrDropStrict (bowbuffer.go:84-104): a type assertion, no conversion
        if ((f.out_type == BOWGPU_INT64) != (v.is_int != 0)) v.valid = 0;
    }
    *f.out_value = v.valid ? v.bits : 0;
    *f.out_valid_byte = (uint8_t)(v.valid ? 1 : 0);
}

// Window.FirstIndex / Window.Bow row range / Window.IsInclusive of every window (rolling.go:177-239)
__global__ __launch_bounds__(256) void window_bounds_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval, int64_t W,
                                                            int inclusive, int pre_rows, const int64_t *first_idx,
                                                            int64_t *first_index, int64_t *slice_begin, int64_t *slice_end,
                                                            uint8_t *is_incl) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < W; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = first_idx[k], b = first_idx[k + 1];
        const bool incl = inclusive && b < n && ts[b] == s0 + (k + 1) * interval;       // rolling.go:201-209
        bool real = b > a;
        if (k == 0 && pre_rows) real = b > 0 && ts[b - 1] >= s0;                        // rows below s0 alone do not make a window
        const int64_t end = b + (incl ? 1 : 0);
        const bool empty = !(real || incl);
        if (first_index) first_index[k] = a;
        if (slice_begin) slice_begin[k] = empty ? 0 : a;                                // NewEmptySlice :225-226
        if (slice_end) slice_end[k] = empty ? 0 : end;
        if (is_incl) is_incl[k] = incl ? 1 : 0;
    }
}

// ------------------------------------------------------------------ launchers
static inline unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 256 * 16) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int launch_window_first_rows(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t *first_idx, uint32_t *status) {
    hipLaunchKernelGGL(window_first_rows_kernel, dim3(grid_for(n + 1)), dim3(256), 0, c->stream, ts, n, plan.s0, plan.interval,
                       plan.magic, plan.W, first_idx, status);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_exclusive_scan(Ctx *c, const int32_t *in, int64_t n, int64_t *out /* n+1 */, int64_t *block_sums, int64_t *d_total) {
    const int64_t nblocks = (n + 2047) / 2048;
    if (n == 0) { BG_HIP(hipMemsetAsync(out, 0, 8, c->stream)); BG_HIP(hipMemsetAsync(d_total, 0, 8, c->stream)); return 0; }
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, c->stream, in, n, block_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, c->stream, block_sums, nblocks, d_total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, c->stream, in, n, block_sums, out);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_col_order(Ctx *c, const uint64_t *values, const uint32_t *vbits, int64_t vbit0, int64_t n, int32_t type, uint32_t *d_flags) {
    BG_HIP(hipMemsetAsync(d_flags, 0, 4, c->stream));
    if (n == 0) return 0;
    hipLaunchKernelGGL(col_order_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, values, vbits, vbit0, n, type, d_flags);
    BG_HIP(hipGetLastError());
    return 0;
}

int launch_window_bounds(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int inclusive, int pre_rows,
                         const int64_t *first_idx, int64_t *first_index, int64_t *slice_begin, int64_t *slice_end, uint8_t *is_incl) {
    if (plan.W <= 0) return 0;
    hipLaunchKernelGGL(window_bounds_kernel, dim3(grid_for(plan.W)), dim3(256), 0, c->stream, ts, n, plan.s0, plan.interval, plan.W,
                       inclusive, pre_rows, first_idx, first_index, slice_begin, slice_end, is_incl);
    BG_HIP(hipGetLastError());
    return 0;
}

}  // namespace bowgpu

// launchers that take the kernel-parameter structs live here with them
namespace bowgpu {

static_assert(sizeof(WholeParams) == sizeof(WholeParamsH), "WholeParams layout");
static_assert(sizeof(WholeFinal) == sizeof(WholeFinalH), "WholeFinal layout");

// rows[0] = first valid row of the column (-1: none), rows[1] = last valid row; word walks by one thread (a handful of words
// unless the column starts / ends with a long run of nulls)
__global__ void first_last_valid_kernel(const uint32_t *vbits, int64_t vbit0, int64_t n, int64_t *rows) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    rows[0] = next_valid(vbits, vbit0, n, 0);
    rows[1] = prev_valid(vbits, vbit0, n, n - 1);
}

int launch_first_last_valid(Ctx *c, const uint32_t *vbits, int64_t vbit0, int64_t n, int64_t *d_rows) {
    hipLaunchKernelGGL(first_last_valid_kernel, dim3(1), dim3(64), 0, c->stream, vbits, vbit0, n, d_rows);
    BG_HIP(hipGetLastError());
    return 0;
}

size_t nbr_index_bytes(int64_t n, int64_t vbit0) {
    const int64_t nblocks = (vbit0 + (n > 0 ? n : 1) - 1) / kNbrBlockBits - vbit0 / kNbrBlockBits + 1;
    return (size_t)nblocks * 8 * 4;
}

int nbr_index_build(Ctx *c, const uint32_t *vbits, int64_t vbit0, int64_t n, void *work, NbrIndex *out) {
    out->prev_before = nullptr; out->next_after = nullptr; out->g0 = vbit0 / kNbrBlockBits;
    if (!vbits || n <= 0) return 0;  // no bitmap: every lookup answers without the index
    const int64_t nblocks = (vbit0 + n - 1) / kNbrBlockBits - out->g0 + 1;
    int64_t *w = reinterpret_cast<int64_t *>(work);
    int64_t *last_in = w, *first_in = w + nblocks, *prev_before = w + 2 * nblocks, *next_after = w + 3 * nblocks;
    hipLaunchKernelGGL(nbr_block_kernel, dim3((unsigned)((nblocks + 3) / 4)), dim3(256), 0, c->stream, vbits, vbit0, n, out->g0, nblocks,
                       last_in, first_in);
    hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, c->stream, last_in, first_in, nblocks, prev_before, next_after);
    BG_HIP(hipGetLastError());
    out->prev_before = prev_before; out->next_after = next_after;
    return 0;
}

int fill_run(Ctx *c, const FillParams &p) {
    if (p.n > 0) hipLaunchKernelGGL(fill_kernel, dim3(grid_for(p.n)), dim3(256), 0, c->stream, p);
    BG_HIP(hipGetLastError());
    return 0;
}

// partial states of the column's row chunks, then their ordered merge into partials[nblocks] (one extra slot)
int whole_run(Ctx *c, const void *params_blob, int64_t nblocks) {
    const WholeParams &p = *reinterpret_cast<const WholeParams *>(params_blob);
    hipLaunchKernelGGL(whole_partial_kernel, dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    hipLaunchKernelGGL(whole_merge_kernel, dim3(1), dim3(256), 0, c->stream, p.partials, nblocks, p.partials + nblocks);
    BG_HIP(hipGetLastError());
    return 0;
}

int whole_final_run(Ctx *c, const void *partials, int64_t nblocks, int64_t nrows, int64_t first_value, int64_t last_value,
                    const void *final_blob) {
    const WholeFinal &f = *reinterpret_cast<const WholeFinal *>(final_blob);
    hipLaunchKernelGGL(whole_final_kernel, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<const Stats *>(partials), nblocks, nrows,
                       first_value, last_value, f);
    BG_HIP(hipGetLastError());
    return 0;
}

size_t stats_size() { return sizeof(Stats); }

}  // namespace bowgpu
