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

// exclusive running max of last_in (-> prev_before) and exclusive reverse running min of first_in (-> next_after), in tiles of
// 1024 blocks: nbr_tile_kernel reduces every tile to (max, min); nbr_scan_kernel folds the tiles before / after its own into a
// carry and scans its tile with shuffles (coalesced reads and writes; the first version - one workgroup, every thread walking a
// contiguous run of entries - spent 80 us in dependent strided loads)
__device__ __forceinline__ void block_max_min(long long &hi, long long &lo, int64_t *s_hi, int64_t *s_lo) {  // 1024 threads; result in s_hi[0], s_lo[0]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        const long long y = __shfl_down(hi, o), z = __shfl_down(lo, o);
        if (y > hi) hi = y;
        if (z < lo) lo = z;
    }
    if (lane == 0) { s_hi[wv] = hi; s_lo[wv] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long h = s_hi[0], l = s_lo[0];
        for (int w = 1; w < 16; w++) { if (s_hi[w] > h) h = s_hi[w]; if (s_lo[w] < l) l = s_lo[w]; }
        s_hi[0] = h; s_lo[0] = l;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void nbr_tile_kernel(const int64_t *last_in, const int64_t *first_in, int64_t nblocks, int64_t *tile_hi,
                                                        int64_t *tile_lo) {
    __shared__ int64_t s_hi[16], s_lo[16];
    const int64_t g = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    long long hi = g < nblocks ? last_in[g] : -1, lo = g < nblocks ? first_in[g] : INT64_MAX;
    block_max_min(hi, lo, s_hi, s_lo);
    if (threadIdx.x == 0) { tile_hi[blockIdx.x] = s_hi[0]; tile_lo[blockIdx.x] = s_lo[0]; }
}

__global__ __launch_bounds__(1024) void nbr_scan_kernel(const int64_t *last_in, const int64_t *first_in, int64_t nblocks, const int64_t *tile_hi,
                                                        const int64_t *tile_lo, int64_t ntiles, int64_t *prev_before, int64_t *next_after) {
    __shared__ int64_t s_hi[16], s_lo[16], s_wh[16], s_wl[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // carries: everything in the tiles before / after this one
    long long ch = -1, cl = INT64_MAX;
    for (int64_t j = t; j < ntiles; j += 1024) {
        if (j < (int64_t)blockIdx.x && tile_hi[j] > ch) ch = tile_hi[j];
        if (j > (int64_t)blockIdx.x && tile_lo[j] < cl) cl = tile_lo[j];
    }
    block_max_min(ch, cl, s_hi, s_lo);
    const long long carry_hi = s_hi[0], carry_lo = s_lo[0];
    __syncthreads();
    // the tile: inclusive scans inside each wavefront, the 16 wavefront totals through LDS
    const int64_t g = (int64_t)blockIdx.x * 1024 + t;
    long long ih = g < nblocks ? last_in[g] : -1, il = g < nblocks ? first_in[g] : INT64_MAX;
    for (int o = 1; o < 64; o <<= 1) {
        const long long y = __shfl_up(ih, o), z = __shfl_down(il, o);
        if (lane >= o && y > ih) ih = y;
        if (lane + o < 64 && z < il) il = z;
    }
    if (lane == 63) s_wh[wv] = ih;
    if (lane == 0) s_wl[wv] = il;
    __syncthreads();
    long long before = __shfl_up(ih, 1), after = __shfl_down(il, 1);
    if (lane == 0) before = -1;
    if (lane == 63) after = INT64_MAX;
    for (int w = 0; w < 16; w++) {
        if (w < wv && s_wh[w] > before) before = s_wh[w];
        if (w > wv && s_wl[w] < after) after = s_wl[w];
    }
    if (carry_hi > before) before = carry_hi;
    if (carry_lo < after) after = carry_lo;
    if (g < nblocks) { prev_before[g] = before; next_after[g] = after == INT64_MAX ? -1 : after; }
}


// ------------------------------------------------------------------ Interpolate
// first row of every window: first_idx[k] = lower_bound(ts, s_k) for k in [0, W], first_idx[W] = n
// The row that opens window w also names the first row of every EMPTY window before it.  Short runs of empties are written by
// that row's thread; a long run (two rows 1e9 apart with interval 10 leave 1e8 empty windows: one lane storing them one by one
// took seconds) is queued and filled by the whole grid in window_gaps_kernel.
constexpr int kGapInline = 64;       // empties a thread writes itself
constexpr int kGapListCap = 4096;    // queued runs; beyond that the thread falls back to writing them itself
struct GapRun { int64_t k0, k1, row; };   // first_idx[k0 .. k1) = row
__global__ __launch_bounds__(256) void window_first_rows_kernel(const int64_t *ts, int64_t n, int64_t s0, int64_t interval,
                                                                MagicDiv magic, int64_t W, int64_t *first_idx,
                                                                uint32_t *status, GapRun *gaps, uint32_t *gap_count) {
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
        uint64_t k_end = w + 1;   // k's first row (and every empty window before it): windows wp + 1 .. w
        if ((int64_t)k_end > W) k_end = (uint64_t)W;
        if (w >= wp && k_end > wp + 1 && k_end - (wp + 1) > (uint64_t)kGapInline) {
            const uint32_t slot = atomicAdd(gap_count, 1u);
            if (slot < (uint32_t)kGapListCap) { gaps[slot] = GapRun{(int64_t)(wp + 1), (int64_t)k_end, i}; continue; }
        }
        for (uint64_t k = wp + 1; k < k_end; k++) first_idx[k] = i;
    }
}
__global__ __launch_bounds__(256) void window_gaps_kernel(const GapRun *gaps, const uint32_t *gap_count, int64_t *first_idx) {
    uint32_t ng = *gap_count;
    if (ng > (uint32_t)kGapListCap) ng = kGapListCap;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (uint32_t g = 0; g < ng; g++) {
        const GapRun r = gaps[g];
        for (int64_t k = r.k0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < r.k1; k += stride) first_idx[k] = r.row;
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
__global__ __launch_bounds__(64) void scan_sums_kernel(int64_t *block_sums, int64_t nblocks, int64_t *total) {
    // ONE wavefront: every lane sums its contiguous share of the block sums, the 64 shares are scanned over the lanes, then every lane
    // writes the exclusive prefixes of its share (one thread walking all of them took 22 us for 200 sums: a chain of dependent
    // loads and stores; nblocks = n / 2048 <= ~50 k for n = 1e8)
    const int lane = threadIdx.x;
    const int64_t per = (nblocks + 63) / 64, a = lane * per, b = a + per < nblocks ? a + per : nblocks;
    long long mine = 0;
    for (int64_t i = a; i < b; i++) mine += block_sums[i];
    long long incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    long long run = incl - mine;
    for (int64_t i = a; i < b; i++) { const int64_t v = block_sums[i]; block_sums[i] = run; run += v; }
    if (lane == 63) *total = incl;
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

// ------------------------------------------------------------------ FillLinear / FillMean / FillPrevious / FillNext
// The value a null row i receives (bits, valid) under p.method.  rp / rn: its nearest valid rows (-1: none), with their
// values pbits / nbits (and, FillLinear, the reference column's values there: pref / nref; own_ref: at row i itself).
// kFillInt / kRefInt: the filled / the reference column is Int64 (else Float64)
template <int kMethod, bool kFillInt, bool kRefInt>
__device__ __forceinline__ void fill_row(const FillParams &p, int64_t i, int64_t rp, int64_t rn, uint64_t pbits, uint64_t nbits,
                                         uint64_t pref, uint64_t nref, uint64_t own_ref, uint64_t *bits_io, int *valid_io) {
    uint64_t bits = *bits_io;
    int valid = 0;
    if (kMethod == BOWGPU_FILL_PREVIOUS) {
        if (rp >= 0) { bits = pbits; valid = 1; }                      // arr.Value(fillRowIndex): bowfill.go:196-199
    } else if (kMethod == BOWGPU_FILL_NEXT) {
        if (rn >= 0) { bits = nbits; valid = 1; }
    } else if (kMethod == BOWGPU_FILL_MEAN) {
        if (rp >= 0 && rn >= 0) {                                       // bowfill.go:145-154
            const double m = (bits_to_f64(pbits, kFillInt ? BOWGPU_INT64 : BOWGPU_FLOAT64) + bits_to_f64(nbits, kFillInt ? BOWGPU_INT64 : BOWGPU_FLOAT64)) / 2;
            bits = kFillInt ? (uint64_t)go_f64_to_i64(round(m)) : (uint64_t)__double_as_longlong(m);
            valid = 1;
        }
    } else {                                                            // FillLinear: bowfill.go:65-97
        const bool v1 = bit_at(p.ref_vbits, p.ref_vbit0, i);
        const bool v2 = rp >= 0 && bit_at(p.ref_vbits, p.ref_vbit0, rp);   // GetFloat64(ref, -1) => (0,false) :72
        const bool v3 = rn >= 0 && bit_at(p.ref_vbits, p.ref_vbit0, rn);
        if (v1 && v2 && v3) {
            const double prev_fill = bits_to_f64(pbits, kFillInt ? BOWGPU_INT64 : BOWGPU_FLOAT64);
            const double next_fill = bits_to_f64(nbits, kFillInt ? BOWGPU_INT64 : BOWGPU_FLOAT64);
            const double row_ref = bits_to_f64(own_ref, kRefInt ? BOWGPU_INT64 : BOWGPU_FLOAT64);
            const double prev_ref = bits_to_f64(pref, kRefInt ? BOWGPU_INT64 : BOWGPU_FLOAT64);
            const double next_ref = bits_to_f64(nref, kRefInt ? BOWGPU_INT64 : BOWGPU_FLOAT64);
            // (the nextRef-prevRef == 0 branch of :78-85 is overwritten by the fall-through below)
            double tmp = row_ref - prev_ref;   // :87-90, four separate statements
            tmp /= next_ref - prev_ref;
            tmp *= next_fill - prev_fill;
            tmp += prev_fill;
            if (kFillInt) bits = (uint64_t)go_f64_to_i64(round(tmp));  // math.Round: half away from zero :93
            else bits = (uint64_t)__double_as_longlong(tmp);
            valid = 1;
        }
    }
    *bits_io = bits;
    *valid_io = valid;
}

// a neighbour of a null row: its row (-1: none), its value and (FillLinear) the reference column's value there
struct FillNb { int64_t row; uint64_t bits, ref; };

// Bow.FillLinear (bowfill.go:14-103), FillPrevious / FillNext (:162-253), FillMean (:105-160).  A wavefront owns 512 consecutive
// rows per trip as four chunks of 128 (FillLinear: 256 rows, two chunks): lane l holds rows 2l, 2l+1 of each chunk (one 16-B load and one 16-B store per chunk,
// all loads of the trip in flight at once).  The kernel is bound by instruction issue unless the per-row work is a handful of
// operations, so everything that is the same for the whole wavefront stays scalar: a chunk's validity comes as two 64-bit
// words (scalar loads) and is split into an even-row and an odd-row lane mask by two ballots; a null row finds its nearest
// valid rows with two count-leading / trailing-zeros on those masks and takes their values from the lanes that hold them
// (shuffles); a run of nulls that reaches past its chunk gets the chunk's carry - the nearest valid row before / after the
// chunk, handed from chunk to chunk in scalar registers, and looked up in the bitmap + neighbour index once per trip, only
// when the trip starts / ends with a null.  The output's validity words are the two result ballots, bit-interleaved.
// kFull: all rows of the trip exist (every trip but the last): no range checks.
// chunks of 128 rows a wavefront takes per trip: FillLinear holds two columns in registers, so half as many
__host__ __device__ constexpr int fill_chunks(int method) { return method == kFillLinear ? 2 : 4; }

template <int kMethod, bool kFillInt, bool kRefInt, bool kFull>
__device__ __forceinline__ void fill_trip(const FillParams &p, const int64_t base, const int lane, const uint64_t lt, const uint64_t gt,
                                          const bool vec, const bool rvec, unsigned long long *nvalid_io) {
    constexpr bool linear = kMethod == kFillLinear;
    constexpr int kC = fill_chunks(kMethod), kTrip = 128 * kC;
    constexpr bool want_prev = kMethod != BOWGPU_FILL_NEXT, want_next = kMethod != BOWGPU_FILL_PREVIOUS;
    unsigned long long nvalid = *nvalid_io;
    const uint64_t *src = p.fill_values + base, *rsrc = linear ? p.ref_values + base : nullptr;
    uint64_t *dst = p.out_values + base;
    const int64_t left_trip = p.n - base;  // rows of the trip that exist (>= 1; >= kTrip when kFull)
    uint64_t a[kC], b[kC], ra[kC], rb[kC];
#pragma unroll
    for (int k = 0; k < kC; k++) {
        const int r = 128 * k + 2 * lane;
        if (kFull && vec) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + r); a[k] = v.x; b[k] = v.y; }
        else { a[k] = r < left_trip ? src[r] : 0; b[k] = r + 1 < left_trip ? src[r + 1] : 0; }
    }
#pragma unroll
    for (int k = 0; k < kC; k++) { ra[k] = 0; rb[k] = 0; }
    if (linear) {
#pragma unroll
        for (int k = 0; k < kC; k++) {
            const int r = 128 * k + 2 * lane;
            if (kFull && rvec) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rsrc + r); ra[k] = v.x; rb[k] = v.y; }
            else { ra[k] = r < left_trip ? rsrc[r] : 0; rb[k] = r + 1 < left_trip ? rsrc[r + 1] : 0; }
        }
    }
    // validity of the trip: per chunk the even-row / odd-row lane masks
    uint64_t me[kC], mo[kC];
    int fl[kC];
    bool any_null = false;
    const int sh = (2 * lane) & 63;
#pragma unroll
    for (int k = 0; k < kC; k++) {
        const int64_t cb = base + 128 * k;
        me[k] = 0; mo[k] = 0; fl[k] = 0;
        if (kFull || cb < p.n) {
            uint64_t w0, w1;
            load_bits128<kFull>(p.fill_vbits, p.fill_vbit0, cb, p.n, &w0, &w1);
            const uint64_t w = lane < 32 ? w0 : w1;
            fl[k] = (int)((w >> sh) & 3ull);
            me[k] = __ballot(fl[k] & 1);
            mo[k] = __ballot(fl[k] & 2);
            uint64_t full_e = ~0ull, full_o = ~0ull;
            if (!kFull) {
                const int64_t left = p.n - cb;
                if (left < 128) {
                    full_e = (left + 1) / 2 >= 64 ? ~0ull : ((1ull << ((left + 1) / 2)) - 1ull);
                    full_o = left / 2 >= 64 ? ~0ull : ((1ull << (left / 2)) - 1ull);
                }
            }
            any_null = any_null || me[k] != full_e || mo[k] != full_o;
        }
    }
    // carries: the nearest valid row before / after every chunk (the same for all lanes)
    FillNb cp[kC], cn[kC];
#pragma unroll
    for (int k = 0; k < kC; k++) { cp[k].row = -1; cp[k].bits = 0; cp[k].ref = 0; cn[k] = cp[k]; }
    if (any_null) {
        if (want_prev) {
            FillNb cur; cur.row = -1; cur.bits = 0; cur.ref = 0;
            if (!(me[0] & 1ull)) {  // the trip starts with a null: what lies before it
                bool far = false;
                cur.row = p.nbr.prev_before ? prev_valid_ix(p.fill_vbits, p.fill_vbit0, p.n, base - 1, p.nbr) : prev_valid_near(p.fill_vbits, p.fill_vbit0, p.n, base - 1, &far);
                if (far && lane == 0 && !__hip_atomic_load(p.far_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(p.far_flag, 1u);
                if (cur.row >= 0) { cur.bits = p.fill_values[cur.row]; if (linear) cur.ref = p.ref_values[cur.row]; }
            }
#pragma unroll
            for (int k = 0; k < kC; k++) {
                cp[k] = cur;
                if (me[k] | mo[k]) {
                    const int le = me[k] ? 63 - __clzll((long long)me[k]) : -1, lo = mo[k] ? 63 - __clzll((long long)mo[k]) : -1;
                    const bool odd = lo >= le;
                    const int sl = odd ? lo : le;
                    cur.row = base + 128 * k + 2 * sl + (odd ? 1 : 0);
                    cur.bits = odd ? lane_value(b[k], sl) : lane_value(a[k], sl);
                    if (linear) cur.ref = odd ? lane_value(rb[k], sl) : lane_value(ra[k], sl);
                }
            }
        }
        if (want_next) {
            FillNb cur; cur.row = -1; cur.bits = 0; cur.ref = 0;
            const int last = (int)(kFull ? kTrip - 1 : (left_trip < kTrip ? left_trip : kTrip) - 1);  // last row of the trip, relative
            const int kl = last >> 7, rl = last & 127;
            uint64_t mlast = 0;
#pragma unroll
            for (int k = 0; k < kC; k++) if (k == kl) mlast = (rl & 1) ? mo[k] : me[k];
            if (!((mlast >> (rl >> 1)) & 1ull)) {  // the trip ends with a null: what lies after it
                bool far = false;
                cur.row = p.nbr.next_after ? next_valid_ix(p.fill_vbits, p.fill_vbit0, p.n, base + kTrip, p.nbr) : next_valid_near(p.fill_vbits, p.fill_vbit0, p.n, base + kTrip, &far);
                if (far && lane == 0 && !__hip_atomic_load(p.far_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(p.far_flag, 1u);
                if (cur.row >= 0) { cur.bits = p.fill_values[cur.row]; if (linear) cur.ref = p.ref_values[cur.row]; }
            }
#pragma unroll
            for (int k = kC - 1; k >= 0; k--) {
                cn[k] = cur;
                if (me[k] | mo[k]) {
                    const int fe = me[k] ? __ffsll((long long)me[k]) - 1 : 64, fo = mo[k] ? __ffsll((long long)mo[k]) - 1 : 64;
                    const bool even = fe <= fo;
                    const int sl = even ? fe : fo;
                    cur.row = base + 128 * k + 2 * sl + (even ? 0 : 1);
                    cur.bits = even ? lane_value(a[k], sl) : lane_value(b[k], sl);
                    if (linear) cur.ref = even ? lane_value(ra[k], sl) : lane_value(rb[k], sl);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kC; k++) {
        const int64_t cb = base + 128 * k;
        const int r = 128 * k + 2 * lane;   // relative to the trip
        const int64_t i = base + r;
        int va = fl[k] & 1, vb = (fl[k] >> 1) & 1;
        uint64_t oa = a[k], ob = b[k];
        const bool in_a = kFull || r < left_trip, in_b = kFull || r + 1 < left_trip;
        const bool need_a = in_a && !va, need_b = in_b && !vb;
        if (any_null && __ballot(need_a || need_b)) {
            FillNb pa, nb2;  // previous of the lane's even row, next of its odd row
            pa = cp[k]; nb2 = cn[k];
            if (want_prev) {
                const uint64_t xe = me[k] & lt, xo = mo[k] & lt;
                const int le = xe ? 63 - __clzll((long long)xe) : -1, lo = xo ? 63 - __clzll((long long)xo) : -1;
                const bool odd = lo >= le;
                const int sl = (xe | xo) ? (odd ? lo : le) : lane;
                const uint64_t sa = __shfl((unsigned long long)a[k], sl), sb = __shfl((unsigned long long)b[k], sl);
                uint64_t sr = 0;
                if (linear) { const uint64_t x = __shfl((unsigned long long)ra[k], sl), y = __shfl((unsigned long long)rb[k], sl); sr = odd ? y : x; }
                if (xe | xo) { pa.row = cb + 2 * sl + (odd ? 1 : 0); pa.bits = odd ? sb : sa; pa.ref = sr; }
            }
            if (want_next) {
                const uint64_t xe = me[k] & gt, xo = mo[k] & gt;
                const int fe = xe ? __ffsll((long long)xe) - 1 : 64, fo = xo ? __ffsll((long long)xo) - 1 : 64;
                const bool even = fe <= fo;
                const int sl = (xe | xo) ? (even ? fe : fo) : lane;
                const uint64_t sa = __shfl((unsigned long long)a[k], sl), sb = __shfl((unsigned long long)b[k], sl);
                uint64_t sr = 0;
                if (linear) { const uint64_t x = __shfl((unsigned long long)ra[k], sl), y = __shfl((unsigned long long)rb[k], sl); sr = even ? x : y; }
                if (xe | xo) { nb2.row = cb + 2 * sl + (even ? 0 : 1); nb2.bits = even ? sa : sb; nb2.ref = sr; }
            }
            // the even row's next: its odd partner when that is valid; the odd row's previous: its even partner
            FillNb na = nb2, pb = pa;
            if (vb) { na.row = i + 1; na.bits = b[k]; na.ref = rb[k]; }
            if (va) { pb.row = i; pb.bits = a[k]; pb.ref = ra[k]; }
            if (need_a) fill_row<kMethod, kFillInt, kRefInt>(p, i, pa.row, na.row, pa.bits, na.bits, pa.ref, na.ref, ra[k], &oa, &va);
            if (need_b) fill_row<kMethod, kFillInt, kRefInt>(p, i + 1, pb.row, nb2.row, pb.bits, nb2.bits, pb.ref, nb2.ref, rb[k], &ob, &vb);
        }
        if (kFull && vec) *reinterpret_cast<ulonglong2 *>(dst + r) = make_ulonglong2(oa, ob);
        else {
            if (in_a) dst[r] = oa;
            if (in_b) dst[r + 1] = ob;
        }
        if (kFull || cb < p.n) {
            const uint64_t oe = __ballot(va), oo = __ballot(vb);
            const uint64_t w0 = interleave32((uint32_t)oe, (uint32_t)oo), w1 = interleave32((uint32_t)(oe >> 32), (uint32_t)(oo >> 32));
            if (lane == 0) {
                unsigned long long *wdst = reinterpret_cast<unsigned long long *>(p.out_valid_words + (cb >> 5));
                wdst[0] = w0;
                if (kFull || cb + 64 < p.n) wdst[1] = w1;
            }
            nvalid += __popcll(oe) + __popcll(oo);
        }
    }
    *nvalid_io = nvalid;
}

template <int kMethod, bool kFillInt, bool kRefInt>
__global__ __launch_bounds__(256) void fill_kernel(const FillParams p) {
    __shared__ unsigned long long block_valid[4];
    unsigned long long nvalid = 0;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.fill_values) | reinterpret_cast<uintptr_t>(p.out_values)) & 15) == 0;
    const bool rvec = kMethod == kFillLinear && (reinterpret_cast<uintptr_t>(p.ref_values) & 15) == 0;
    const uint64_t lt = (1ull << lane) - 1ull, gt = lane == 63 ? 0ull : (~0ull << (lane + 1));
    constexpr int kTrip = 128 * fill_chunks(kMethod);
    for (int64_t base = wave * kTrip; base < p.n; base += nwaves * kTrip) {
        if (base + kTrip <= p.n) fill_trip<kMethod, kFillInt, kRefInt, true>(p, base, lane, lt, gt, vec, rvec, &nvalid);
        else fill_trip<kMethod, kFillInt, kRefInt, false>(p, base, lane, lt, gt, vec, rvec, &nvalid);
    }
    if (lane == 0) block_valid[wv] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = block_valid[0] + block_valid[1] + block_valid[2] + block_valid[3];
        if (t) atomicAdd(p.valid_count, t);
    }
}

// ------------------------------------------------------------------ IsColSorted (bowassertion.go:15-86)
// flags[0] |= 1 if some consecutive valid pair increases, |= 2 if some decreases, |= 4 if any valid value exists.
// Same wave-trip shape as fill_kernel: 512 rows per trip, lane l holds rows 2l, 2l+1 of each 128-row chunk; a valid row finds
// the valid row before it through the chunk's even / odd lane masks and a shuffle, across chunks through a scalar carry.
// Pairs that straddle two trips are settled afterwards from one (first valid, last valid) record per trip
// (col_order_edges_kernel), so no lookup ever walks back through a long run of nulls.
struct TripEdge { uint64_t first, last; int32_t has, _pad; };

template <bool kInt>
__device__ __forceinline__ uint32_t order_cmp(uint64_t prev, uint64_t cur) {
    if (kInt) {
        const int64_t c = (int64_t)prev, x = (int64_t)cur;
        return c < x ? 1u : (c > x ? 2u : 0u);
    }
    const double c = __longlong_as_double((long long)prev), x = __longlong_as_double((long long)cur);
    return c < x ? 1u : (c > x ? 2u : 0u);  // NaN compares false both ways (bowassertion.go:64-74)
}

template <bool kInt>
__global__ __launch_bounds__(256) void col_order_kernel(const uint64_t *values, const uint32_t *vbits, int64_t vbit0, int64_t n,
                                                        TripEdge *edges, uint32_t *flags) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (reinterpret_cast<uintptr_t>(values) & 15) == 0;
    const uint64_t lt = (1ull << lane) - 1ull;
    const int sh = (2 * lane) & 63;
    uint32_t f = 0;
    bool any_valid = false;
    for (int64_t trip = wave; trip * 512 < n; trip += nwaves) {
        const int64_t base = trip * 512;
        const int64_t left_trip = n - base;
        const bool full = left_trip >= 512;
        const uint64_t *src = values + base;
        uint64_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = 128 * k + 2 * lane;
            if (full && vec) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + r); a[k] = v.x; b[k] = v.y; }
            else { a[k] = r < left_trip ? src[r] : 0; b[k] = r + 1 < left_trip ? src[r + 1] : 0; }
        }
        bool chas = false, fhas = false;  // carry = the last valid value so far in this trip; the trip's first valid value
        uint64_t cbits = 0, first = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t cb = base + 128 * k;
            if (cb >= n) break;
            uint64_t w0, w1;
            if (full) load_bits128<true>(vbits, vbit0, cb, n, &w0, &w1);
            else load_bits128<false>(vbits, vbit0, cb, n, &w0, &w1);
            const uint64_t w = lane < 32 ? w0 : w1;
            const int fl = (int)((w >> sh) & 3ull);
            const uint64_t me = __ballot(fl & 1), mo = __ballot(fl & 2);
            if (!(me | mo)) continue;
            const uint64_t xe = me & lt, xo = mo & lt;
            const int le = xe ? 63 - __clzll((long long)xe) : -1, lo = xo ? 63 - __clzll((long long)xo) : -1;
            const bool odd = lo >= le, inch = (xe | xo) != 0;
            const int sl = inch ? (odd ? lo : le) : lane;
            const uint64_t sa = __shfl((unsigned long long)a[k], sl), sb = __shfl((unsigned long long)b[k], sl);
            const uint64_t pbits = inch ? (odd ? sb : sa) : cbits;
            const bool phas = inch || chas;
            if ((fl & 1) && phas) f |= order_cmp<kInt>(pbits, a[k]);
            if (fl & 2) {
                const bool h2 = (fl & 1) || phas;
                if (h2) f |= order_cmp<kInt>((fl & 1) ? a[k] : pbits, b[k]);
            }
            if (!fhas) {
                const int fe = me ? __ffsll((long long)me) - 1 : 64, fo = mo ? __ffsll((long long)mo) - 1 : 64;
                first = fe <= fo ? lane_value(a[k], fe) : lane_value(b[k], fo);
                fhas = true;
            }
            {
                const int ge = me ? 63 - __clzll((long long)me) : -1, go = mo ? 63 - __clzll((long long)mo) : -1;
                cbits = go >= ge ? lane_value(b[k], go) : lane_value(a[k], ge);
                chas = true;
            }
        }
        any_valid = any_valid || fhas;
        if (lane == 0) { TripEdge e; e.first = first; e.last = cbits; e.has = fhas ? 1 : 0; e._pad = 0; edges[trip] = e; }
    }
    // one atomic per workgroup: atomics on a single address are serialised at ~10 ns each (one per wavefront of a 48 k-workgroup
    // launch was measured at 2.2 ms for the kernel)
    __shared__ uint32_t block_flags[4];
    const uint32_t F = (__ballot(f & 1) ? 1u : 0u) | (__ballot(f & 2) ? 2u : 0u) | (any_valid ? 4u : 0u);
    if (lane == 0) block_flags[wv] = F;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t B = block_flags[0] | block_flags[1] | block_flags[2] | block_flags[3];
        if (B) atomicOr(flags, B);
    }
}

// The same for a column WITHOUT nulls (bowassertion.go:15-81 then compares every row with the row before it): no masks, no carries through
// runs of nulls, no per-trip records and no join launches - a row's left neighbour is the lane to the left (DPP wave_shr:1), lane 0's is
// lane 63 of the 128-row group before, and the first row of a trip reads the row in front of the trip from memory (one scalar load per
// 512 rows).  One launch for the whole column: what remains is four 16-byte loads and eight compares per lane and trip.
// (Round 6 also tried to let the last workgroup hand the flags to the host - a ticket counter next to the flags word, no memset in front
// of the launch, no copy behind it: 4096 more atomics on one cache line, 0.32 ms instead of 0.18 per 1e8 rows.  What stays of it: a
// workgroup whose findings are already in the flags word - after the first few, every one of them on a sorted column - does not touch it.)
template <bool kInt>
__global__ __launch_bounds__(256) void col_order_dense_kernel(const uint64_t *values, int64_t n, uint32_t *flags) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (reinterpret_cast<uintptr_t>(values) & 15) == 0;
    uint32_t f = 0;
    for (int64_t trip = wave; trip * 512 < n; trip += nwaves) {
        const int64_t base = trip * 512;
        const int64_t left_trip = n - base;
        const bool full = left_trip >= 512;
        const uint64_t *src = values + base;
        uint64_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = 128 * k + 2 * lane;
            if (full && vec) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + r); a[k] = v.x; b[k] = v.y; }
            else { a[k] = r < left_trip ? src[r] : 0; b[k] = r + 1 < left_trip ? src[r + 1] : 0; }
        }
        uint64_t carry = base > 0 ? src[-1] : 0;   // (uniform address: the row in front of the trip)
        bool has_carry = base > 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = 128 * k + 2 * lane;
            const uint32_t llo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const uint32_t lhi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b[k] >> 32), 0x138, 0xf, 0xf, false);
            const uint64_t left = lane == 0 ? carry : ((uint64_t)lhi << 32) | llo;
            if ((lane > 0 || has_carry) && r < left_trip) f |= order_cmp<kInt>(left, a[k]);
            if (r + 1 < left_trip) f |= order_cmp<kInt>(a[k], b[k]);
            carry = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b[k], 63) |
                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b[k] >> 32), 63) << 32;
            has_carry = true;
        }
    }
    __shared__ uint32_t block_flags[4];
    const uint32_t F = (__ballot(f & 1) ? 1u : 0u) | (__ballot(f & 2) ? 2u : 0u);
    if (lane == 0) block_flags[wv] = F;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t B = block_flags[0] | block_flags[1] | block_flags[2] | block_flags[3] | (blockIdx.x == 0 && n > 0 ? 4u : 0u);
        if (B && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & B) != B) atomicOr(flags, B);
    }
}

// pairs whose rows lie in different trips: 256 consecutive records per workgroup, joined in order by one thread out of LDS;
// the workgroup's own (first valid, last valid) record goes to the next level (the host repeats until one record is left)
template <bool kInt>
__global__ __launch_bounds__(256) void col_order_join_kernel(const TripEdge *in, int64_t n_in, TripEdge *out, uint32_t *flags) {
    __shared__ TripEdge seg[256];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    TripEdge e; e.first = 0; e.last = 0; e.has = 0; e._pad = 0;
    if (q < n_in) e = in[q];
    seg[threadIdx.x] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t f = 0;
        TripEdge c; c.has = 0; c.last = 0; c.first = 0; c._pad = 0;
        for (int j = 0; j < 256; j++) {
            if (!seg[j].has) continue;
            if (c.has) f |= order_cmp<kInt>(c.last, seg[j].first);
            else c.first = seg[j].first;
            c.last = seg[j].last; c.has = 1;
        }
        out[blockIdx.x] = c;
        if (f) atomicOr(flags, f);
    }
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

// Level 1 for the VALUE reducers (Count / Sum / Mean / Min / Max / First / Last; no time-weighted one on the column), round 5: nothing
// is merged across lanes per step.  The kernel above spends ~600 of its ~850 instructions per 512-row step on the six-round shuffle tree
// over the 14-field state (0.56 ms per 1e8 rows, 0.18 of peak); none of these reducers needs the adjacency that tree preserves - what
// depends on ROW ORDER is positional and reduces by row index:
//   Sum, Count       per lane over all its steps, added up at the end (whole-frame sums never were in row order: 1e-12 rel, see above)
//   First / Last     the lane's lowest / highest valid row and its bits; the lowest / highest index wins
//   Min / Max        minmax.go:16-28 over the concatenation = the first valid value if that is a NaN (a NaN seed sticks), else the
//                    extremum of the non-NaN values, the EARLIEST of equal ones (+0 / -0): (value, row) pairs, ties to the lower row
// A step's validity bits are one byte load per lane, not a load per row.  The wavefront's
// result is written as the same Stats record, for the same contiguous quarter of the workgroup's rows: the merge levels do not change.
// kTs: a time-weighted reducer wants the column's integrals too (integral.go:14-31, :46-62 over the whole frame): every valid point's
// terms with its NEXT valid point.  Inside a 128-row chunk that point is the lane's own second row or the first point of the next lane
// that has one (ballot, lowest set bit above the lane, four ds_bpermute - long_short_kernel's neighbour search); a chunk's first
// point closes the pair with the last point so far (uniform: carried from chunk to chunk, step to step); the terms are summed per
// lane, their order is free like the sums' (1e-12 rel).  The range's first and last point go into the Stats record: stats_merge
// stitches the ranges.
// (launch bounds: four wavefronts per SIMD - the time-weighted form then fits 128 vector registers with three of them parked in scratch;
// with nulls as well it would park seventeen and ran 0.46 ms on one box, 0.58 on the next: three wavefronts, 141 registers, none parked)
template <bool kNulls, bool kTs>
__global__ __launch_bounds__(256, (kNulls && kTs) ? 3 : 4) void whole_value_kernel(const WholeParams p) {
    __shared__ Stats part[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.x * p.chunk;
    int64_t hi = lo + p.chunk;
    if (hi > p.n) hi = p.n;
    const int64_t steps = hi > lo ? (hi - lo + 511) / 512 : 0;
    const int64_t s_per = (steps + 3) / 4;
    const int64_t q_lo = lo + (int64_t)wv * s_per * 512;
    int64_t q_hi = q_lo + s_per * 512;
    if (q_hi > hi) q_hi = hi;
    const bool vec = (reinterpret_cast<uintptr_t>(p.values) & 15) == 0;
    double sum = 0.0, mn = 0.0, mx = 0.0;
    long long count = 0, mn_row = -1, mx_row = -1, first_row = -1, last_row = -1;
    uint64_t first_bits = 0, last_bits = 0;
    double trap = 0.0, step = 0.0;                            // kTs: this lane's share of the terms
    double c_t = 0.0, c_v = 0.0, f_t = 0.0, f_v = 0.0;        // (uniform) the last point so far; the range's first point
    bool c_has = false, any_pair = false;
    const bool tvec = kTs && (reinterpret_cast<uintptr_t>(p.ts) & 15) == 0;
    // Lane l holds rows 2l, 2l + 1 of each of the step's four 128-row chunks: every load instruction reads 1 KB of consecutive bytes
    // (eight consecutive rows per lane - the kernel above - make an instruction touch 32 cache lines for 16 bytes each, four times
    // over: 0.234 ms per 1e8 rows at four times the L2 traffic).  Rows still ascend within a lane.  The next step's loads are in flight
    // while this step's rows are consumed.
    // The step's 512 validity bits are 64 (65 at an odd bit offset) consecutive bytes: lane l loads byte l - one coalesced load per
    // step - and a lane's two bits of a chunk come out of the lane that holds them (ds_bpermute: the crossbar, no LDS memory).  (The
    // chunks' bits as wave-uniform scalar words, as the rolling kernels read them, cost this lean kernel 0.28 ms per 1e8 rows where
    // this costs 0.2: four dependent trips to the scalar cache per step and nothing else to do meanwhile.)
    const uint8_t *vbytes = reinterpret_cast<const uint8_t *>(p.vbits);
    const int64_t last_byte = kNulls ? (p.vbit0 + p.n - 1) >> 3 : 0;
    auto fetch = [&](int64_t base, uint64_t (&v)[8], uint64_t (&t)[8], uint32_t &vb, uint32_t &vb64) {
#pragma unroll
        for (int k = 0; k < 4; k++) load_pair(p.values, base + 128 * k + 2 * lane, q_hi, vec && (base & 1) == 0, v[2 * k], v[2 * k + 1]);
        if (kTs) {
#pragma unroll
            for (int k = 0; k < 4; k++) load_pair(reinterpret_cast<const uint64_t *>(p.ts), base + 128 * k + 2 * lane, q_hi, tvec && (base & 1) == 0, t[2 * k], t[2 * k + 1]);
        }
        vb = 0; vb64 = 0;
        if (kNulls) {
            const int64_t b0 = (p.vbit0 + base) >> 3;
            if (b0 + lane <= last_byte) vb = vbytes[b0 + lane];
            if (b0 + 64 <= last_byte) vb64 = vbytes[b0 + 64];
        }
    };
    // (kTs: no loads ahead - two more columns' worth of registers in flight cost the time-weighted form a third of its wavefronts)
    constexpr bool kAhead = !kTs;
    uint64_t vn[8], tn[8];
    uint32_t vbn = 0, vb64n = 0;
    if (kAhead && q_lo < q_hi) fetch(q_lo, vn, tn, vbn, vb64n);
    for (int64_t base = q_lo; base < q_hi; base += 512) {
        uint64_t v[8], tt[8];
        if (!kAhead) fetch(base, vn, tn, vbn, vb64n);
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = vn[k]; if (kTs) tt[k] = tn[k]; }
        uint32_t m = 0xFFu;
        if (kNulls) {
            // 16 bits from byte j on: the lane's own byte and its upper neighbour's (lane 63: the 65th byte)
            uint32_t up = (uint32_t)__shfl_down((int)vbn, 1);
            if (lane == 63) up = vb64n;
            const uint32_t pair = vbn | (up << 8);
            const uint32_t s_off = (uint32_t)((p.vbit0 + base) & 7);
            m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t bitpos = s_off + 128u * k + 2u * (uint32_t)lane;   // < 519
                const uint32_t j = bitpos >> 3;
                uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j < 64u ? j : 63u) << 2), (int)pair);
                if (j >= 64u) w = vb64n;                                           // (the 65th byte is in every lane)
                m |= ((w >> (bitpos & 7u)) & 3u) << (2 * k);
            }
        }
        if (kAhead && base + 512 < q_hi) fetch(base + 512, vn, tn, vbn, vb64n);
        if (kTs) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t rx = base + 128 * k + 2 * lane;
                const bool okx = rx < q_hi && ((m >> (2 * k)) & 1u), oky = rx + 1 < q_hi && ((m >> (2 * k + 1)) & 1u);
                const uint64_t hasm = __ballot(okx || oky);
                if (!hasm) continue;                                   // (uniform)
                const double xt = (double)(int64_t)tt[2 * k], yt = (double)(int64_t)tt[2 * k + 1];   // whole.go / integral.go: float64(ts)
                const double xv = bits_to_f64(v[2 * k], p.type), yv = bits_to_f64(v[2 * k + 1], p.type);
                const double o_t = okx ? xt : yt, o_v = okx ? xv : yv;      // the lane's first point
                const double l_t = oky ? yt : xt, l_v = oky ? yv : xv;      // ... and its last
                // the first point of the next lane that has one
                const uint64_t above = (hasm >> lane) >> 1;
                const int src = (lane + 1 + (above ? __ffsll((long long)above) - 1 : 0)) << 2;
                const double a_t = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(o_t)), __builtin_amdgcn_ds_bpermute(src, __double2loint(o_t)));
                const double a_v = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(o_v)), __builtin_amdgcn_ds_bpermute(src, __double2loint(o_v)));
                if (okx && oky) { trap += (xv + yv) / 2 * (yt - xt); step += xv * (yt - xt); }
                if ((okx || oky) && above) { trap += (l_v + a_v) / 2 * (a_t - l_t); step += l_v * (a_t - l_t); }
                // (uniform) the chunk's first point closes the pair with the last point so far; its last point is the new one
                const int lf = __ffsll((long long)hasm) - 1, ll = 63 - __clzll((long long)hasm);
                const double q_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(o_t), lf), __builtin_amdgcn_readlane(__double2loint(o_t), lf));
                const double q_v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(o_v), lf), __builtin_amdgcn_readlane(__double2loint(o_v), lf));
                if (c_has) {
                    if (lane == 0) { trap += (c_v + q_v) / 2 * (q_t - c_t); step += c_v * (q_t - c_t); }
                    any_pair = true;
                } else { f_t = q_t; f_v = q_v; }
                any_pair = any_pair || (hasm & (hasm - 1)) != 0 || __ballot(okx && oky) != 0;
                c_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(l_t), ll), __builtin_amdgcn_readlane(__double2loint(l_t), ll));
                c_v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(l_v), ll), __builtin_amdgcn_readlane(__double2loint(l_v), ll));
                c_has = true;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // (selects, not branches: 30 % nulls would split every row's step in two.  A row without a value adds +0.0 - exact: the sum
            // starts at +0.0 and x + (+0.0) == x for every x a sum that started there can be)
            const int64_t r = base + 128 * (k >> 1) + 2 * lane + (k & 1);
            const bool ok = r < q_hi && ((m >> k) & 1u);
            const double x = bits_to_f64(v[k], p.type);
            sum += ok ? x : 0.0;
            count += ok ? 1 : 0;
            const bool fst = ok && first_row < 0;
            first_row = fst ? r : first_row; first_bits = fst ? v[k] : first_bits;
            last_row = ok ? r : last_row; last_bits = ok ? v[k] : last_bits;
            // (rows come in ascending order within the lane: strict comparisons keep the earliest of equal values; a NaN never enters)
            const bool lo_ = ok && x == x && (mn_row < 0 || x < mn), hi_ = ok && x == x && (mx_row < 0 || x > mx);
            mn = lo_ ? x : mn; mn_row = lo_ ? r : mn_row;
            mx = hi_ ? x : mx; mx_row = hi_ ? r : mx_row;
        }
    }
    // ---- across the lanes, once per wavefront: by value, ties to the lower row; by row index
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_down(sum, o);
        count += __shfl_down(count, o);
        const double omn = __shfl_down(mn, o), omx = __shfl_down(mx, o);
        const long long omn_row = __shfl_down(mn_row, o), omx_row = __shfl_down(mx_row, o);
        if (omn_row >= 0 && (mn_row < 0 || omn < mn || (omn == mn && omn_row < mn_row))) { mn = omn; mn_row = omn_row; }
        if (omx_row >= 0 && (mx_row < 0 || omx > mx || (omx == mx && omx_row < mx_row))) { mx = omx; mx_row = omx_row; }
        const long long ofr = __shfl_down(first_row, o), olr = __shfl_down(last_row, o);
        const unsigned long long ofb = __shfl_down((unsigned long long)first_bits, o), olb = __shfl_down((unsigned long long)last_bits, o);
        if (ofr >= 0 && (first_row < 0 || ofr < first_row)) { first_row = ofr; first_bits = ofb; }
        if (olr > last_row) { last_row = olr; last_bits = olb; }
        if (kTs) { trap += __shfl_down(trap, o); step += __shfl_down(step, o); }
    }
    if (lane == 0) {
        Stats st;
        stats_init(st);
        if (kTs && c_has) {
            st.has_point = 1; st.has_pair = any_pair ? 1 : 0;
            st.first_pt = f_t; st.first_pv = f_v; st.pt = c_t; st.pv = c_v;
            st.integ_trap = trap; st.integ_step = step;
        }
        if (count > 0) {
            st.sum = sum; st.count = count; st.has_value = 1;
            st.first_bits = first_bits; st.last_bits = last_bits;
            const double f = bits_to_f64(first_bits, p.type);
            st.has_nn = mn_row >= 0 ? 1 : 0;
            st.nn_min = mn; st.nn_max = mx;
            // (stats_value's seed rule on this range: a NaN first value sticks, else the non-NaN extrema - f itself is one of them)
            st.vmin = (f != f || !st.has_nn) ? f : mn;
            st.vmax = (f != f || !st.has_nn) ? f : mx;
        }
        part[wv] = st;
    }
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

// The tail of a whole-frame call as ONE launch per column (round 6; rounds 1 - 5: a merge launch, one launch per reducer, a launch +
// synchronisation for the first / last timestamp in front, two copies behind): the workgroup partials merged in order (whole_merge_kernel's
// tree), FirstValue / LastValue read here - int64(float64(first / last ts)), whole.go:54-71 -, every reducer of the column evaluated and
// stored where the host reads it: the registered block (value j at host_values[j], validity at host_valid[j]).
struct WholeFinish {
    const int64_t *ts;
    int64_t nrows;
    int32_t n, _pad;
    int32_t slot[kMaxAggs];            // which output of the call
    WholeFinal f[kMaxAggs];
    uint64_t *host_values;
    uint8_t *host_valid;
};
__global__ __launch_bounds__(256) void whole_finish_kernel(const Stats *partials, int64_t nblocks, const WholeFinish fin) {
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
    if (threadIdx.x != 0) return;
    Stats acc = part[0];
    stats_merge(acc, part[1]); stats_merge(acc, part[2]); stats_merge(acc, part[3]);
    const int64_t first_value = go_f64_to_i64((double)fin.ts[0]), last_value = go_f64_to_i64((double)fin.ts[fin.nrows - 1]);
    for (int j = 0; j < fin.n; j++) {
        const WholeFinal &f = fin.f[j];
        // reduce_val expects (win_start, interval) with LastValue = win_start + interval (whole.go:64-71)
        Val v = reduce_val(f.kind, acc, fin.nrows, first_value, last_value - first_value, f.col_is_int);
        if (v.valid) {
            for (int k = 0; k < f.n_factors; k++) {
                if (v.is_int) v.bits = (uint64_t)go_f64_to_i64((double)(int64_t)v.bits * f.factors[k]);
                else v.bits = (uint64_t)__double_as_longlong(__longlong_as_double((long long)v.bits) * f.factors[k]);
            }
            // SetOrDropStrict (bowbuffer.go:84-104): a type assertion, no conversion
            if ((f.out_type == BOWGPU_INT64) != (v.is_int != 0)) v.valid = 0;
        }
        fin.host_values[fin.slot[j]] = v.valid ? v.bits : 0;
        fin.host_valid[fin.slot[j]] = (uint8_t)(v.valid ? 1 : 0);
    }
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
    void *w;
    BG_TRY(ctx_pool(c, kPoolGaps, 256 + sizeof(GapRun) * kGapListCap, &w));
    uint32_t *gap_count = reinterpret_cast<uint32_t *>(w);
    GapRun *gaps = reinterpret_cast<GapRun *>(reinterpret_cast<char *>(w) + 256);
    BG_HIP(hipMemsetAsync(gap_count, 0, 4, c->stream));
    hipLaunchKernelGGL(window_first_rows_kernel, dim3(grid_for(n + 1)), dim3(256), 0, c->stream, ts, n, plan.s0, plan.interval,
                       plan.magic, plan.W, first_idx, status, gaps, gap_count);
    // long runs of empty windows: the whole grid fills them (a few blocks when there are none: it reads the count and leaves)
    const int64_t spare = plan.W - n;   // an upper bound of the empties
    const unsigned gblocks = spare > (int64_t)kGapInline ? (unsigned)std::min<int64_t>(2048, (spare + 255) / 256) : 1u;
    hipLaunchKernelGGL(window_gaps_kernel, dim3(gblocks), dim3(256), 0, c->stream, gaps, gap_count, first_idx);
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
    const int64_t ntrips = (n + 511) / 512;
    const dim3 grid(grid_for(ntrips, 4, 4096)), block(256);   // (1024 workgroups left 16 wavefronts per CU waiting out one load each: 0.25 ms per 1e8 rows)
    const bool is_int = type == BOWGPU_INT64;
    if (!vbits && !(route_mask() & BOWGPU_ROUTE_FORCE_GENERAL)) {   // no nulls: one launch, no per-trip records to join (BOWGPU_ROUTE_FORCE_GENERAL: the tests' switch to the general form)
        if (is_int) hipLaunchKernelGGL(col_order_dense_kernel<true>, grid, block, 0, c->stream, values, n, d_flags);
        else hipLaunchKernelGGL(col_order_dense_kernel<false>, grid, block, 0, c->stream, values, n, d_flags);
        BG_HIP(hipGetLastError());
        return 0;
    }
    void *w;
    BG_TRY(ctx_pool(c, kPoolColOrder, (size_t)(ntrips + ntrips / 128 + 8) * sizeof(TripEdge), &w));  // every level of the join
    TripEdge *edges = reinterpret_cast<TripEdge *>(w);
    if (is_int) hipLaunchKernelGGL(col_order_kernel<true>, grid, block, 0, c->stream, values, vbits, vbit0, n, edges, d_flags);
    else hipLaunchKernelGGL(col_order_kernel<false>, grid, block, 0, c->stream, values, vbits, vbit0, n, edges, d_flags);
    int64_t n_in = ntrips;
    while (n_in > 1) {
        const int64_t n_out = (n_in + 255) / 256;
        TripEdge *out = edges + n_in;
        if (is_int) hipLaunchKernelGGL(col_order_join_kernel<true>, dim3((unsigned)n_out), block, 0, c->stream, edges, n_in, out, d_flags);
        else hipLaunchKernelGGL(col_order_join_kernel<false>, dim3((unsigned)n_out), block, 0, c->stream, edges, n_in, out, d_flags);
        edges = out;
        n_in = n_out;
    }
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
    return (size_t)nblocks * 8 * 4 + (size_t)((nblocks + 1023) / 1024) * 8 * 2;  // four per-block tables + two per-tile ones
}

int nbr_index_build(Ctx *c, const uint32_t *vbits, int64_t vbit0, int64_t n, void *work, NbrIndex *out) {
    out->prev_before = nullptr; out->next_after = nullptr; out->g0 = vbit0 / kNbrBlockBits;
    if (!vbits || n <= 0) return 0;  // no bitmap: every lookup answers without the index
    const int64_t nblocks = (vbit0 + n - 1) / kNbrBlockBits - out->g0 + 1;
    int64_t *w = reinterpret_cast<int64_t *>(work);
    int64_t *last_in = w, *first_in = w + nblocks, *prev_before = w + 2 * nblocks, *next_after = w + 3 * nblocks;
    const int64_t ntiles = (nblocks + 1023) / 1024;
    int64_t *tile_hi = w + 4 * nblocks, *tile_lo = tile_hi + ntiles;
    hipLaunchKernelGGL(nbr_block_kernel, dim3((unsigned)((nblocks + 3) / 4)), dim3(256), 0, c->stream, vbits, vbit0, n, out->g0, nblocks,
                       last_in, first_in);
    hipLaunchKernelGGL(nbr_tile_kernel, dim3((unsigned)ntiles), dim3(1024), 0, c->stream, last_in, first_in, nblocks, tile_hi, tile_lo);
    hipLaunchKernelGGL(nbr_scan_kernel, dim3((unsigned)ntiles), dim3(1024), 0, c->stream, last_in, first_in, nblocks, tile_hi, tile_lo, ntiles,
                       prev_before, next_after);
    BG_HIP(hipGetLastError());
    out->prev_before = prev_before; out->next_after = next_after;
    return 0;
}

int fill_run(Ctx *c, const FillParams &p) {
    if (p.n > 0) {
        const dim3 grid(grid_for(p.n, 256, 2048)), block(256);
        const bool fi = p.fill_type == BOWGPU_INT64, ri = p.ref_type == BOWGPU_INT64;
        // (FillPrevious / FillNext copy bits: one instantiation serves both column types)
        if (p.method == BOWGPU_FILL_PREVIOUS) hipLaunchKernelGGL((fill_kernel<BOWGPU_FILL_PREVIOUS, false, false>), grid, block, 0, c->stream, p);
        else if (p.method == BOWGPU_FILL_NEXT) hipLaunchKernelGGL((fill_kernel<BOWGPU_FILL_NEXT, false, false>), grid, block, 0, c->stream, p);
        else if (p.method == BOWGPU_FILL_MEAN) {
            if (fi) hipLaunchKernelGGL((fill_kernel<BOWGPU_FILL_MEAN, true, false>), grid, block, 0, c->stream, p);
            else hipLaunchKernelGGL((fill_kernel<BOWGPU_FILL_MEAN, false, false>), grid, block, 0, c->stream, p);
        } else if (fi) {
            if (ri) hipLaunchKernelGGL((fill_kernel<kFillLinear, true, true>), grid, block, 0, c->stream, p);
            else hipLaunchKernelGGL((fill_kernel<kFillLinear, true, false>), grid, block, 0, c->stream, p);
        } else {
            if (ri) hipLaunchKernelGGL((fill_kernel<kFillLinear, false, true>), grid, block, 0, c->stream, p);
            else hipLaunchKernelGGL((fill_kernel<kFillLinear, false, false>), grid, block, 0, c->stream, p);
        }
    }
    BG_HIP(hipGetLastError());
    return 0;
}

// partial states of the column's row chunks, then their ordered merge into partials[nblocks] (one extra slot)
int whole_run(Ctx *c, const void *params_blob, int64_t nblocks) {
    const WholeParams &p = *reinterpret_cast<const WholeParams *>(params_blob);
    // (whole_partial_kernel - the shuffle-tree form of rounds 1 - 4 - stays as the second implementation: BOWGPU_ROUTE_FORCE_GENERAL)
    if (route_mask() & BOWGPU_ROUTE_FORCE_GENERAL) hipLaunchKernelGGL(whole_partial_kernel, dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    else if (p.need_ts) {
        if (p.vbits) hipLaunchKernelGGL((whole_value_kernel<true, true>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((whole_value_kernel<false, true>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    } else if (p.vbits) hipLaunchKernelGGL((whole_value_kernel<true, false>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL((whole_value_kernel<false, false>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
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

// the value kernel of one column (no merge launch): whole_finish_run follows
int whole_value_run(Ctx *c, const void *params_blob, int64_t nblocks) {
    const WholeParams &p = *reinterpret_cast<const WholeParams *>(params_blob);
    if (p.need_ts) {
        if (p.vbits) hipLaunchKernelGGL((whole_value_kernel<true, true>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((whole_value_kernel<false, true>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    } else if (p.vbits) hipLaunchKernelGGL((whole_value_kernel<true, false>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL((whole_value_kernel<false, false>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, p);
    BG_HIP(hipGetLastError());
    return 0;
}

static_assert(sizeof(WholeFinish) == sizeof(WholeFinishH), "host / device layouts of the whole-frame tail");
int whole_finish_run(Ctx *c, const void *partials, int64_t nblocks, const WholeFinishH &fin) {
    hipLaunchKernelGGL(whole_finish_kernel, dim3(1), dim3(256), 0, c->stream, reinterpret_cast<const Stats *>(partials), nblocks,
                       *reinterpret_cast<const WholeFinish *>(&fin));
    BG_HIP(hipGetLastError());
    return 0;
}

size_t stats_size() { return sizeof(Stats); }

}  // namespace bowgpu
