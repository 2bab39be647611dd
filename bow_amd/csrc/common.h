// Internal declarations shared by the HIP translation units of libbowgpu.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, 160 KB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/bowgpu.h"
#include "debug_routes.h"

namespace bowgpu {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define BG_HIP(expr)                                         \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(_e, #expr);    \
    } while (0)

#define BG_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// Counts the calls that write to or free device memory THROUGH THE LIBRARY, in any thread (bowgpu_free, bowgpu_memcpy_h2d, bowgpu_memset,
// the generators): results cached from a caller's device buffer (the Interpolate count -> fill prefix) are dropped when it moves.
uint64_t device_write_epoch();
void device_write_epoch_bump();

// test / A-B routing of the calling thread (BOWGPU_ROUTE_* bits; bowgpu_debug_set_route): never the environment
uint32_t route_mask();

// ---------------------------------------------------------------- per-thread context
struct Ctx {
    int device = 0;
    bool inited = false;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in use (own or external)
    // small persistent device scratch (status words, long-window list) + pinned host mirror
    void *d_scratch = nullptr;
    size_t d_scratch_bytes = 0;
    void *h_pinned = nullptr;
    size_t h_pinned_bytes = 0;
    void *h_bounce = nullptr;   // two halves of pinned staging for copies of pageable caller buffers (copy_d2h / copy_h2d)
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    bool bounce_busy[2] = {false, false};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_kernel_ms = 0.0;
    const char *last_kernel_name = "";  // the tile kernel last_kernel_ms brackets
    int64_t last_slow_rows = 0;         // rows of the last Aggregate / Interpolate call served by a kernel kept for the shapes the fast ones decline (bowgpu_last_call_slow_rows)
    // grow-only pool of temporaries reused across calls (word-aligned validity working copies, ...):
    // hipMalloc / hipFree per call cost more than the kernels' fixed overhead
    static constexpr int kPoolSlots = 40;   // 0..15 output validity working copies, kPoolInterp.. Interpolate / fill scratch, last: long windows
    void *pool[kPoolSlots] = {};
    size_t pool_bytes[kPoolSlots] = {};
    uint64_t pool_gen[kPoolSlots] = {};     // bumped on every ctx_pool() of the slot: lets a cached result notice that its block was handed out again
    // what bowgpu_rolling_interpolate_count leaves for the _fill call that follows it on the same (unchanged) columns: the
    // per-trip prefix of exact window heads, so that the interval column is not scanned a second time (extras.cpp)
    struct InterpCache {
        bool valid = false;
        const void *ts_values = nullptr;
        int64_t ts_offset = 0, n = 0, interval = 0, raw_offset = 0;
        bool sharded = false;
        int64_t global_s0 = 0, left_ts = 0;
        int has_left = 0;
        uint64_t gen = 0, gen0 = 0;         // pool_gen of the two prefix blocks when they were written
        uint64_t epoch = 0;                 // device_write_epoch() when it was written: any write / free through the library since then drops it
        int64_t s0 = 0, W = 0, first_ts = 0, last_ts = 0, offset_norm = 0, kq = -1, drop = 0, M = 0, wbase = 0;
        int kq_empty = 0, inclusive = 0, e0 = 0;
    } interp_cache;
    void *d_params = nullptr;      // 4 KB device block holding the kernels' descriptor struct
    // what an Interpolate _count over an interval column WITH NULLS leaves for its _fill (extras.cpp NullTsState: device temporaries);
    // freed through null_ts_cache_free by the fill, bowgpu_trim, bowgpu_set_device and at thread exit
    void *null_ts_cache = nullptr;
    void (*null_ts_cache_free)(void *) = nullptr;
};
void ctx_drop_null_ts_cache(Ctx *c);
int ctx_get(Ctx **out);                       // initialises HIP on first use; fails loudly without a GPU
int ctx_scratch(Ctx *c, size_t bytes, void **dptr);
int ctx_pinned(Ctx *c, size_t bytes, void **hptr);
int ctx_params(Ctx *c, void **dptr);
int ctx_pool(Ctx *c, int slot, size_t bytes, void **dptr);

// ---------------------------------------------------------------- temp device buffers
// RAII device allocation (stream-ordered free at scope exit after a sync by the caller).
// Device scratch of one call.  Blocks come from / go back to a small per-thread cache (api.cpp devbuf_*), so a call pays
// neither hipMalloc nor hipFree (which also synchronises the device) once the sizes have been seen; every entry point
// synchronises its stream before its DevBufs go out of scope.
void devbuf_release(void *p, size_t cap);
void devbuf_cache_drop();
int devbuf_acquire(size_t n, void **p, size_t *cap);
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;  // requested
    size_t cap = 0;    // size of the underlying block
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap) { o.p = nullptr; o.bytes = 0; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { if (p) devbuf_release(p, cap); p = o.p; bytes = o.bytes; cap = o.cap; o.p = nullptr; o.bytes = 0; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { if (p) devbuf_release(p, cap); }
    int alloc(size_t n);
};

// A column made device-resident: aliases the caller's pointers (BOWGPU_DEVICE) or owns an
// uploaded copy (BOWGPU_HOST).  values_dev points at element `offset` already; validity is
// kept as (aligned dword pointer, bit offset).
struct DevCol {
    const void *values = nullptr;      // element 0 of the logical array
    const uint32_t *vbits = nullptr;   // 4-byte aligned word containing bit `vbit0`
    int64_t vbit0 = 0;                 // bit index (from vbits) of logical row 0
    int64_t vwords = 0;                // number of readable 32-bit words at vbits
    int64_t length = 0;
    int64_t null_count = 0;            // exact (counted on device if the caller said -1)
    int32_t type = 0;
    DevBuf own_values, own_validity;
};
int devcol_prepare(Ctx *c, const bowgpu_col *col, DevCol *out, bool need_values, bool need_validity);
int count_nulls_device(Ctx *c, DevCol *dc);

// An output column on the device: aliases the caller's buffers or owns temporaries that are
// copied back by finish().
struct DevOut {
    void *values = nullptr;
    uint8_t *validity = nullptr;
    int64_t capacity = 0;
    int pool_slot = -1;   // >= 0: the validity working copy comes from the context pool
    DevBuf own_values, own_validity;
    bowgpu_out *user = nullptr;
};
// copies of caller buffers: pageable ones through the context's pinned staging, registered ones (BOWGPU_HOST_PINNED) directly
int copy_d2h(Ctx *c, void *dst, const void *src, size_t bytes, bool registered = false);
int copy_h2d(Ctx *c, void *dst, const void *src, size_t bytes, bool registered = false);
int devout_prepare(Ctx *c, bowgpu_out *out, int64_t slots, DevOut *d, int pool_slot = -1);
int devout_finish(Ctx *c, DevOut *d, int64_t slots, int32_t type, int64_t null_count, bool copy_bitmap = true);

// ---------------------------------------------------------------- division by the interval
// Granlund–Montgomery round-up method (N = 64): exact floor(n / d) for every 0 <= n < 2^64.
struct MagicDiv {
    uint64_t m;
    uint32_t sh1, sh2;
};
MagicDiv magic_make(uint64_t d);

// ---------------------------------------------------------------- window plan (host side)
struct Plan {
    int64_t interval = 0;
    int64_t offset = 0;    // normalised
    int64_t s0 = 0;
    int64_t W = 0;
    int64_t first_ts = 0, last_ts = 0;
    MagicDiv magic{};
};
int plan_make(Ctx *c, const bowgpu_col *ts, int64_t interval, int64_t raw_offset, Plan *p);
bool kind_needs_inclusive(int kind);
int kind_type(int kind);
bool kind_never_nil(int kind);
bool kind_reads_values(int kind);

// multi.cpp: one Rolling.Aggregate over the devices of bowgpu_set_devices (*done = false: not a call for it, nothing was touched)
int multi_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive, bool strict,
                    const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, bowgpu_agg_info *info, bool *done);
int multi_interpolate_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive, bool strict,
                                const bowgpu_interp *interps, int32_t ninterps, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                                bowgpu_agg_info *info, bool *done);

// ---------------------------------------------------------------- kernels (rolling_agg.hip)
constexpr int kMaxCols = 8;    // value columns reduced per launch
constexpr int kMaxAggs = 16;   // output columns per launch

struct AggDesc {
    int32_t kind;
    int32_t slot;        // index into AggParams::cols (value column), -1 for ts-only reducers
    int32_t out_type;    // BOWGPU_FLOAT64 / BOWGPU_INT64
    int32_t n_factors;
    double factors[BOWGPU_MAX_FACTORS];
    void *out_values;
    uint32_t *out_valid; // nullptr when the reducer never yields nil (WindowStart/Sum/Count/NumRows)
};

struct ColDesc {
    const void *values;
    const uint32_t *vbits;  // nullptr => no nulls
    int64_t vbit0;
    int64_t vwords;
    int32_t type;
    int32_t need_ts;        // some reducer of this column integrates over time
};

enum : uint32_t { kPassNeedVals = 1, kPassMinMax = 2, kPassFirstLast = 4, kPassNullable = 8 };

struct AggParams {
    const int64_t *ts;
    int64_t n;              // rows in this launch's column slice
    int64_t row_base;       // global row index of ts[0] (sharding); 0 otherwise
    int64_t s0;
    int64_t interval;
    int64_t W;              // windows addressable in the outputs
    int64_t wid_base;       // global window id of output slot 0 (sharding); 0 otherwise
    MagicDiv magic;
    uint32_t m32, sh1_32, sh2_32;  // 32-bit form of the same division (valid when fits32)
    int32_t fits32;         // interval < 2^32: tiles whose ts span fits 32 bits use 32-bit window arithmetic
    int32_t inclusive;      // effective Options.Inclusive
    int32_t ncols;
    int32_t naggs;
    int32_t _pad0;
    int32_t pre_rows;       // s0 > ts[0] (negative ts + truncating division): rows below s0 ride in window 0
    ColDesc cols[kMaxCols];
    AggDesc aggs[kMaxAggs];
    // per column-slot pass summary (index = slot + 1; index 0 is the "no column" pass), precomputed on the host so
    // the kernels do not scan the aggregator list per tile
    uint32_t pass_mask[kMaxCols + 1];   // bit a set: aggregator a belongs to this pass
    uint32_t pass_flags[kMaxCols + 1];  // kPass* bits
    int32_t first_pass_slot, last_val_slot, n_nullable_max;
    int32_t bits_preset;    // output bitmaps start as all-ones (rolling_simple.hip): the long-window path clears empties instead of setting valids
    // status block in device memory
    uint32_t *status;       // [0]=unsorted flag, [1]=long-window count, [2]=overflow flag
    int64_t *long_list;     // kLongLists sub-lists of pairs (global window id, first row), long_cap pairs each
    int64_t long_cap;
};

constexpr int kSimpleMaxAggs = 16;
enum : uint32_t { kNeedStep = 1, kNeedTrap = 2, kNeedMinMax = 4, kNeedSum = 8, kNeedFirstLast = 16 };
// descriptor of rolling_simple.hip's kernel: value columns of one type, factor-free outputs, 32-bit window ids
struct SimpleParams {
    const int64_t *ts;
    int64_t n, s0, interval, W;            // s0 = start of output slot 0 (a shard: global s0 + wid_base * interval)
    int64_t wid_base;                      // global id of output slot 0 (only for the long-window queue)
    MagicDiv magic;                        // 64-bit magic of the interval: the per-tile base window of the kWide variants
    int32_t shift_k;                       // kWide: trailing zero bits of the interval; m32 / sh1 / sh2 then divide by interval >> shift_k
    uint32_t kind_mask[5];                 // outputs by class (bit a = output a): 0 Sum / Mean / First / Last, 1 Min / Max, 2 step integrals, 3 trapezoid integrals, 4 WindowStart / Count / NumRows
    uint32_t col_mask[kMaxCols];           // outputs of value column c
    uint32_t need;                         // kNeed* bits: which running statistics the outputs of the call read (set by the host: one scalar test per use in the kernels)
    uint32_t m32, sh1, sh2;
    int32_t naggs;
    int32_t ncols;                         // value columns (>= 1; reducers over the interval column use it as a column)
    int32_t inclusive;                     // rolling_tw.hip: windows are built inclusive (some reducer needs it)
    int32_t pre_rows;                      // s0 lies above the first timestamp: the rows below it ride in window 0
    uint32_t unaligned_mask;               // bit c: value column c starts on an 8-byte, not a 16-byte boundary; bit 31: the interval column
    const void *values[kMaxCols];
    const uint32_t *vbits[kMaxCols];       // nullptr: this column has no nulls
    int64_t vbit0[kMaxCols], vwords[kMaxCols];
    int32_t col_is_int[kMaxCols];          // Int64 column (read as float64(v), First / Last return Int64)
    int32_t kind[kSimpleMaxAggs];
    int32_t col[kSimpleMaxAggs];           // column slot each output reads (WindowStart / NumRows ride with slot 0)
    int32_t nfac[kSimpleMaxAggs];          // transformation.Factor chain of each output (factor.go:7-20), usually empty
    double fac[kSimpleMaxAggs][BOWGPU_MAX_FACTORS];
    uint64_t *out_values[kSimpleMaxAggs];
    uint32_t *out_valid[kSimpleMaxAggs];   // nullptr for never-nil reducers; all bitmaps are preset to ones by the host
    uint32_t *status;         // [0] unsorted, [2] list overflow, [4] redo with the general lean kernel, [16..79] long-window counts
    int64_t *long_list;
    int64_t long_cap;
};
int launch_rolling_simple(Ctx *c, const SimpleParams &p, int need, bool is_int, bool has_nulls, bool wide, bool dense);   // dense: the larger head list (windows of < 3 rows)
bool rolling_simple_plain(const SimpleParams &p, bool is_int, bool has_nulls);   // the call takes the unpadded instantiation (short windows over one Float64 column without nulls): its small list holds 254 heads
int launch_rolling_tw(Ctx *c, const SimpleParams &p, bool is_int, bool has_nulls, bool wide, bool ts32);  // time-weighted reducers / inclusive windows (rolling_tw.hip)

// rolling_fused.hip: Interpolate -> Aggregate in one pass.  Per value column pass: the column's interpolator (what interp_device.h
// synth_value_pt reads of an InterpCol, same field names)
struct FusedCol {
    int32_t type, kind;            // BOWGPU_FLOAT64 / BOWGPU_INT64 ; BOWGPU_INTERP_*
    int32_t has_prev, prev_t_valid, prev_v_valid, next_valid;   // Options.PrevRow (linear.go:14-18, stepprevious.go:13-15); next_valid: always 0 here
    double const_value, prev_t, prev_v, next_t, next_v;
    int64_t prev_v_i64;
};
struct FusedParams {
    SimpleParams s;                // FIRST: the kernel reads the output pointers through the kernel-argument segment at SimpleParams' offsets
    FusedCol cols[kMaxCols];       // by column pass (SimpleParams::values[c])
    double inv_interval;           // (1 / interval) * (1 + 2^-40): floor(x * inv_interval) == x / interval for every x < 2^32 (rolling_fused.hip fdiv32)
};
int launch_rolling_fused(Ctx *c, const FusedParams &fp, int need, bool has_nulls);
// Interpolate + validateInterpolation (extras.cpp; reference rolling/interpolation.go:30-96)
int interp_validate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_options *o, const bowgpu_interp *interps, int32_t ninterps);

int launch_rolling_twc(Ctx *c, const SimpleParams &p, bool long_halo = false);   // (long_halo: 256 rows of look-ahead) nullable columns under time-weighted reducers, 32-bit times: the valid points compacted first (rolling_twc.hip)
constexpr int64_t kCompactLongMaxAvgRows = 255;             // ... and, with 256 rows of look-ahead, up to this many (beyond: the streaming form) - one kind of integral, First / Last
constexpr int64_t kCompactLongBothMaxAvgRows = 176;         //     both kinds of integral, Min / Max: the streaming form is ahead from 192 rows on (api.cpp compact_long_max_rows)
constexpr int64_t kCompactValuesMinAvgRows = 48;            // value reducers alone: only sums AND extrema on a nullable column, from this window length on
constexpr int64_t kCompactMinAvgRows = 12;                  // ... for calls whose windows average at least this many rows (its head list: 92 per 640 rows)
int launch_rolling_aggregate(Ctx *c, const AggParams &p);   // general kernel (rolling_agg.hip)
int launch_rolling_fast(Ctx *c, const AggParams &p);        // lean kernel for exclusive windows without time-weighted reducers (rolling_fast.hip)
constexpr int kLongChunkRows = 4096;
constexpr int kLongStreamRows = 512;  // chunk of the streaming form of the long-window reduction (long_windows.hip)
constexpr int64_t kLongOnlyAvgRows = 128;   // calls with BOTH kinds of integral whose windows average at least this many rows skip the tile kernels: streaming form (api.cpp job_run)
constexpr int64_t kLongStreamAnyAvgRows = 129;   // ... the same for every other reducer set (256 until long_short_kernel took a boundary per 128-row trip; 128 until the tile kernels' walks became branch-free: windows of exactly 128 rows fit a tile's look-ahead and the tile kernels win there)
constexpr int64_t kLongBisectAvgRows = 512; // ... bisection form where the streaming form does not apply (BOWGPU_ROUTE_LONG_CLASSIC; W >= 2^32)
constexpr int64_t kLongClassicAvgRows = 1ll << 22;   // ... and from here on the handful of giant windows go by bisection + per-window chunks
constexpr int kLongLists = 64;      // sub-lists of the long-window queue (agg_device.h push_long_window)
constexpr int kLongCountWord = 16;  // status[kLongCountWord + s] = entries in sub-list s
constexpr int kStatusWords = kLongCountWord + kLongLists;
struct LongListStarts { int64_t start[kLongLists + 1]; };  // prefix sums of the sub-list counts (host side)
size_t long_entry_size();
size_t long_part_size();
int stream_rw_run(Ctx *c, const void *a, const void *b, int64_t bytes_each, void *o0, void *o1, int64_t rows_per_slot, int64_t nslots, bool nt,
                  int reps, float *ms);
int stream_sum_run(Ctx *c, const void *a, const void *b, int64_t bytes_each, int mode, int blocks_per_cu, int reps, uint64_t *d_out, float *ms);
int launch_long_windows_v2(Ctx *c, const AggParams &p, const LongListStarts *starts, void *entries, int32_t *nchunks, int64_t *offsets,
                           int64_t *block_sums, int64_t *d_total, int32_t *work_entry, void *partials, int64_t max_work, bool strict = false,
                           int64_t n_given = 0);
// the queued windows of a tile pass without the host in between: grid from the queue's capacity, counts read on the device; short windows
// walked in row order, the others listed in big_entries / big_nchunks (status[kQueueBigWord] of them) for launch_long_windows_v2(n_given)
int launch_long_queue(Ctx *c, const AggParams &p, int64_t capacity, void *big_entries, int32_t *big_nchunks, int64_t walk_max_rows, bool strict);
constexpr int kQueueBigWord = 8;          // status word: windows long_queue_kernel left to the chunked machinery
constexpr int64_t kQueueWalkMaxRows = 1024;   // ... those longer than this
size_t long_stream_workspace(int64_t n, int64_t W, int ncols);
int launch_long_stream(Ctx *c, const AggParams &p, void *workspace);   // every window of the call, one read of the rows
int launch_fix_tail_bits(Ctx *c, uint8_t *bitmap, int64_t nbits);
// every output bitmap of one call in one launch (rolling_agg.hip)
struct BitmapBatch {
    int32_t n, status_words;
    int64_t nbits;                 // W
    uint32_t *work[kMaxAggs];      // word-aligned working copies the kernels update
    uint8_t *user[kMaxAggs];       // finish: the caller's device buffer of ceil(W/8) bytes (nullptr: host-resident output, copied separately)
    int32_t ones[kMaxAggs];        // preset: start all-valid (else all-null)
    int32_t count[kMaxAggs];       // finish: count the valid bits into counts[a]
    uint32_t *status;              // preset zeroes status[0 .. status_words) and counts[0 .. kMaxAggs)
    unsigned long long *counts;
    char *host_block;              // finish, one workgroup per bitmap (nbits <= kFinishHostBits): registered host memory that receives the status
                                   // words [0, status_words) and, at byte 1024 + 8 a, bitmap a's count - by the kernel's own stores
    const int64_t *check_ts;       // preset: a caller-supplied plan is checked against this interval column (nullptr: no check);
    int64_t check_n, check_first, check_last;   // status[6] = 1 when its first / last row are not the plan's two timestamps
};
constexpr int64_t kFinishHostBits = 262144;   // up to here ONE workgroup finishes a bitmap (32 KB: a microsecond) and can hand its count to the host itself
int launch_preset_bitmaps(Ctx *c, const BitmapBatch &b);
int launch_finish_bitmaps(Ctx *c, const BitmapBatch &b);
int launch_popcount(Ctx *c, const uint32_t *words, int64_t bit0, int64_t nbits, uint64_t *d_count);
int launch_fetch_two(Ctx *c, const int64_t *col, int64_t i0, int64_t i1, int64_t *host_out);   // host_out: registered host memory

// mode.hip: one aggregation.Mode output over the windows whose first rows are first_idx[0 .. W]
int launch_mode(Ctx *c, const int64_t *ts, const int64_t *first_idx, int64_t n, int64_t s0, int64_t interval, int64_t W, int pre_rows,
                int inclusive, const void *values,
                const uint32_t *vbits, int64_t vbit0, int is_int, const bowgpu_agg *agg, void *out_values, uint32_t *out_valid,
                int64_t *n_mid, int64_t *n_long);

// neighbour index of a validity bitmap (interp_fill.hip nbr_index_build)
struct NbrIndex {
    const int64_t *prev_before;  // [nblocks] last valid row in any earlier block, -1 if none
    const int64_t *next_after;   // [nblocks] first valid row in any later block, -1 if none
    int64_t g0;                  // absolute block number of the column's first bit
};
// ts_nulls.hip: an interval column with nulls rewritten for the tile kernels (forward-filled timestamps, the rows that belong to a window)
// (inclusive: the keep rule of inclusive windows; plain - inclusive only - receives the interval column's validity without the rows
// that sit on a window start with a null timestamp right behind them: ts_nulls.hip)
int launch_ts_nullfill(Ctx *c, const int64_t *ts, const uint32_t *tbits, int64_t tbit0, int64_t n, const struct NbrIndex &ix, int64_t s0, int64_t interval,
                       const MagicDiv &magic, int inclusive, int64_t *ts_eff, uint64_t *keep, uint64_t *plain, unsigned long long *d_dropped);
// the outputs of IntegralTrapezoid / WeightedAverageLinear for the windows behind such rows, recomputed by a walk in the reference's order
struct QuirkFixAgg {
    const void *values; const uint32_t *vbits; int64_t vbit0;     // the reducer's input column (its own validity)
    uint64_t *out_values; uint32_t *out_valid;                    // its output: 8-byte slots, validity words (bit 0 = window 0)
    int32_t type, kind, n_factors, _pad;
    double factors[BOWGPU_MAX_FACTORS];
};
struct QuirkFix { int32_t naggs, _pad; QuirkFixAgg a[8]; };
int launch_ts_quirk_fix(Ctx *c, const int64_t *ts, const uint32_t *tbits, int64_t tbit0, int64_t n, const struct NbrIndex &ix, int64_t s0, int64_t interval,
                        const MagicDiv &magic, int64_t W, const QuirkFix &fx, unsigned long long *d_fixed);
int launch_count_to_f64(Ctx *c, uint64_t *v, int64_t n);
int fetch_valid(Ctx *c, const bowgpu_col *col, int64_t row, int *valid);   // (api.cpp) validity bit of one row of a column, wherever it lives
// Rolling.Interpolate over an interval column with nulls: the kept rows compacted (ts_nulls.hip)
constexpr int kMaxCompactCols = 16;
struct CompactCols {
    int32_t ncols, ts_col;
    const uint64_t *values[kMaxCompactCols]; const uint32_t *vbits[kMaxCompactCols]; int64_t vbit0[kMaxCompactCols];   // the call's columns
    uint64_t *out_values[kMaxCompactCols];      // their kept rows
    uint64_t *lookup_bits[kMaxCompactCols];     // validity of the compacted call's columns (nullptr: the interval column - dense)
    uint64_t *patch_values[kMaxCompactCols]; uint32_t *patch_valid[kMaxCompactCols];   // the outputs interp_patch_kernel corrects
    uint64_t *ts_bits;                          // inclusive iteration: which compacted rows have a timestamp (nullptr: not wanted)
};
// the interpolators of the call + the both-valid bitmaps of the compacted columns with their neighbour indices: what interp_patch_kernel
// needs to make the synthetic row of a window that begins behind null rows (inclusive iteration)
struct PatchInterps {
    int64_t m;
    const uint64_t *both_bits[kMaxCompactCols];
    NbrIndex nbr[kMaxCompactCols];
    int32_t type[kMaxCompactCols], kind[kMaxCompactCols], has_prev[kMaxCompactCols], prev_t_valid[kMaxCompactCols], prev_v_valid[kMaxCompactCols];
    double const_value[kMaxCompactCols], prev_t[kMaxCompactCols], prev_v[kMaxCompactCols];
    int64_t prev_v_i64[kMaxCompactCols];
};
int launch_keep_counts(Ctx *c, const uint64_t *keep, int64_t nw, int32_t *counts);
int launch_compact_rows(Ctx *c, const uint64_t *keep, const int64_t *base, int64_t n, const int64_t *ts_eff, const uint32_t *tbits, int64_t tbit0,
                        const uint64_t *plain, const CompactCols &cc, int64_t *marker, uint32_t *flags);
int launch_pack_flags(Ctx *c, const uint32_t *flags, int64_t m, const CompactCols &cc, uint64_t *marker_bits);
int launch_interp_patch(Ctx *c, const int64_t *marker_out, const uint32_t *marker_valid, int64_t m_out, const uint32_t *flags, const CompactCols &cc,
                        const PatchInterps &px);
int launch_and_bits(Ctx *c, const uint32_t *a, int64_t abit0, const uint32_t *b, int64_t bbit0, int64_t n, uint64_t *out);

// shard.hip
int launch_range_state(Ctx *c, const AggParams &p, int mode, uint64_t wid, const bowgpu_carry_state *d_seeds,
                       bowgpu_carry_state *d_states_out, const bowgpu_next_row *d_next = nullptr, int seed_alive = 1, int strict = 0);

int launch_fill_empty(Ctx *c, const AggParams &p, int64_t slot0, int64_t slot1);

// interp_fill.hip
int launch_window_first_rows(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t *first_idx, uint32_t *status);
int launch_exclusive_scan(Ctx *c, const int32_t *in, int64_t n, int64_t *out, int64_t *block_sums, int64_t *d_total);
int launch_col_order(Ctx *c, const uint64_t *values, const uint32_t *vbits, int64_t vbit0, int64_t n, int32_t type, uint32_t *d_flags);
int launch_window_bounds(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int inclusive, int pre_rows,
                         const int64_t *first_idx, int64_t *first_index, int64_t *slice_begin, int64_t *slice_end, uint8_t *is_incl);
int whole_run(Ctx *c, const void *params_blob, int64_t nblocks);
int whole_final_run(Ctx *c, const void *partials, int64_t nblocks, int64_t nrows, int64_t first_value, int64_t last_value,
                    const void *final_blob);
size_t stats_size();

// Neighbour index of one validity bitmap: for every block of kNbrBlockBits bits (absolute bit positions, so blocks are
// word-aligned whatever the Arrow offset) the nearest valid ROW before the block and after it.  Bounds every
// previous/next-valid lookup to one block of words + one table read, however long the runs of nulls are.
constexpr int kNbrBlockBits = 4096;
constexpr int kPoolMode = 17;    // aggregation.Mode: its output's validity working copy
constexpr int kPoolColOrder = 18; // IsColSorted: one (first valid, last valid) record per 512-row trip
constexpr int kPoolShard = 19;   // shard stitch: seed / merged states and the next shard's first row
constexpr int kPoolGaps = 32;    // window_first_rows: queued runs of empty windows
constexpr int kPoolWhole = 33;        // bowgpu_aggregate_whole: partial states, the reducers' values and validity bytes
constexpr int kPoolInterpEdge = 31;   // Interpolate: the trips' edge words (interp_wave3_kernel)
constexpr int kPoolInterp = 20;  // context pool slots 20..30: tile counts, their scan, scan sums, one neighbour index per column
// every user its own slot: 0..15 the outputs' validity working copies, 17..19, 20..30 (kPoolInterp + 0..10), 31, 32, 33, kPoolSlots - 1 the long windows
static_assert(kPoolMode != kPoolColOrder && kPoolColOrder != kPoolShard && kPoolMode != kPoolShard && kPoolMode > 15 && kPoolShard < kPoolInterp &&
              kPoolInterp + 10 < kPoolInterpEdge && kPoolInterpEdge < kPoolGaps && kPoolGaps < kPoolWhole && kPoolWhole < Ctx::kPoolSlots - 1,
              "context pool slots must be distinct");
// kernel parameter blocks of interp_fill.hip (filled by extras.cpp, passed by value)
struct InterpCol {
    const uint64_t *values;
    const uint32_t *vbits;
    int64_t vbit0;
    int32_t type;
    int32_t kind;
    double const_value;
    int32_t has_prev, prev_t_valid, prev_v_valid, _pad;
    double prev_t, prev_v;
    int64_t prev_v_i64;
    uint64_t *out_values;
    uint32_t *out_valid_words; // output validity bitmap (zeroed by the host before the launch)
    NbrIndex nbr;              // of this column's bitmap (Linear / StepPrevious look their neighbours up through it)
    int32_t next_valid, _pad2; // sharded Interpolate: the nearest valid point on the shards to the right (Linear)
    double next_t, next_v;
};
struct InterpParams {
    const int64_t *ts;
    int64_t n, s0, interval, W;
    MagicDiv magic;
    const int32_t *tile_local;         // pass 1 of interpolate.hip: per 256 rows, the exact heads in the earlier rows of the same super-tile
    const int64_t *super_before;       // ... and per super-tile of kInterpSuperRows rows, the exact heads in all earlier super-tiles
    uint32_t *status;                  // [0] |= 1: interval column not ascending; [1]: window kq has no row of its own
    int64_t kq;                        // index of the window that starts at -1 (the reference's "no first value" sentinel), else -1
    int64_t drop;                      // leading rows that belong to no window (interp_quirk_kernel), normally 0
    uint32_t m32, sh1_32, sh2_32;      // 32-bit magic of the interval (fast32 only)
    int32_t fast32;                    // interp_fast32(plan, kq): 32-bit window ids, integer exact-head test
    int32_t has_left, wide32;          // wide32: interp_wide32() - interp_wave3_kernel's trip-relative form applies (fast32 implies it); sharded Interpolate: rows exist on shards to the left, the last of them at left_ts;
    int64_t left_ts, wbase;            // the windows up to theirs (wbase = its id + 1) are not this shard's to account for
    int32_t ncols, ts_col;
    int32_t allow_wave2;               // the whole-trip wave kernel may take the call (0: the call is being redone after its run list overflowed)
    int32_t kq_empty;                  // window kq has no row of its own (pass 1's finding)
    int32_t inclusive, e0;             // Options.Inclusive (interp_wave2 / wave3 kernels only); e0: row 0 sits exactly on the first window's start
    int64_t n_out;                     // rows the call is to produce (n + what the count pass found): interp_wave3_kernel's last trip checks that it ends there
    uint64_t *edge_words;              // interp_wave3_kernel: [ncols][trips of 512 rows] - a trip's bits of the bitmap word it shares with the trip before it
    uint32_t *trip_valid;              // interp_wave3_kernel: [ncols][trips] valid outputs per trip (nullptr: not kept - a pass over the bitmaps counts them)
    unsigned long long *valid_counts;  // ... summed by interp_edge_fix_kernel into [ncols][kInterpEdgeBlocks] partial counts
    uint32_t *host_status;             // in place: registered host memory that interp_edge_fix_kernel copies the 16 status words into (valid_counts then lies there too)
    int32_t in_place, aligned16;       // aligned16: ts and every input column's values are 16-byte aligned (interp_wave3_kernel's vector loads unconditional); out_valid_words ARE the caller's bitmaps and nobody zeroed them: every word of [0, n_out) gets stored
    InterpCol cols[kMaxCols];
};
int64_t interp_tiles(int64_t n);
bool interp_fast32(const Plan &plan, int64_t kq);
bool interp_wide32(const Plan &plan, int64_t kq);   // ... without the bound on the frame's span: 32-bit arithmetic relative to each trip
void interp_magic32(int64_t interval, uint32_t *m, uint32_t *sh1, uint32_t *sh2);
constexpr int kInterpSuperRows = 8192;
constexpr int kInterpEdgeBlocks = 128;   // interp_edge_fix_kernel: workgroups (= partial valid-output counts) per column
int64_t interp_supers(int64_t n);
// pass 1: tile_local [ceil(n/256)] + super_sum [supers] (scratch) + super_before [supers + 1] + *d_total; two launches
int launch_interp_count(Ctx *c, const int64_t *ts, int64_t n, const Plan &plan, int64_t kq, int has_left, int64_t left_ts,
                        int32_t *tile_local, int32_t *super_sum, int64_t *super_before, int64_t *d_total, uint32_t *status,
                        int64_t *host_back /* registered host memory: [0] total, [1], [2] the four status words */);
int launch_interp_tiles(Ctx *c, const InterpParams &p);
bool interp_takes_wave3(const InterpParams &p);   // (with p.allow_wave2 set) - that kernel needs no neighbour index on its first attempt
enum { kFillLinear = -1 };  // FillParams::method; >= 0: BOWGPU_FILL_PREVIOUS / NEXT / MEAN
struct FillParams {
    const uint64_t *ref_values; const uint32_t *ref_vbits; int64_t ref_vbit0; int32_t ref_type;  // FillLinear only
    const uint64_t *fill_values; const uint32_t *fill_vbits; int64_t fill_vbit0; int32_t fill_type;
    int32_t method;
    int64_t n;
    uint64_t *out_values;
    uint32_t *out_valid_words;   // 8-byte aligned, a multiple of 64 bits long
    unsigned long long *valid_count;  // += valid output rows
    NbrIndex nbr;                // of the fill column's bitmap; prev_before == nullptr: not built - the kernel then looks at most 2048
                                 // rows beyond a trip's ends and raises *far_flag when that does not reach (the host repeats the call with the index)
    uint32_t *far_flag;
};
// builds the index of (vbits, vbit0, n) into `work` (nbr_index_bytes(n, vbit0) bytes of device memory)
int launch_first_last_valid(Ctx *c, const uint32_t *vbits, int64_t vbit0, int64_t n, int64_t *d_rows);
size_t nbr_index_bytes(int64_t n, int64_t vbit0);
int nbr_index_build(Ctx *c, const uint32_t *vbits, int64_t vbit0, int64_t n, void *work, NbrIndex *out);
int fill_run(Ctx *c, const FillParams &p);
struct WholeParamsH {
    const int64_t *ts;
    const uint64_t *values;
    const uint32_t *vbits;
    int64_t vbit0;
    int64_t n;
    int32_t type;
    int32_t need_ts;
    void *partials;
    int64_t chunk;
};
struct WholeFinalH {
    int32_t kind, out_type, col_is_int, n_factors;
    double factors[BOWGPU_MAX_FACTORS];
    uint64_t *out_value;
    uint8_t *out_valid_byte;
};

struct WholeFinishH {
    const int64_t *ts;
    int64_t nrows;
    int32_t n, _pad;
    int32_t slot[kMaxAggs];
    WholeFinalH f[kMaxAggs];
    uint64_t *host_values;    // registered host memory
    uint8_t *host_valid;
};
int whole_value_run(Ctx *c, const void *params_blob, int64_t nblocks);
int whole_finish_run(Ctx *c, const void *partials, int64_t nblocks, const WholeFinishH &fin);

// generate.hip
int launch_gen_dense(Ctx *c, int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val);
int launch_gen_sparse(Ctx *c, int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val, uint8_t *validity);
int launch_checksum64(Ctx *c, const void *dev, int64_t n, uint64_t *d_out2, uint64_t index_base = 0);

}  // namespace bowgpu
