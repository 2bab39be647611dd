/*
 * bowgpu.h — C ABI of the MI355X-native rolling-window aggregation path for Metronlab/bow.
 *
 * This is the drop-in boundary: the entry points a cgo shim inside the
 * reference's `rolling` package would bind (INTEGRATION.md shows that shim).  Plain C99,
 * plain pointers and sizes, no exceptions, no torch / HIP types.  Every function returns
 * 0 on success or a negative BOWGPU_ERR_* code; bowgpu_last_error() gives the message
 * (thread-local).  Nothing here ever falls back to a CPU implementation: if the HIP
 * runtime or a GPU is missing the call fails with BOWGPU_ERR_NO_DEVICE.
 *
 * Column layout is Arrow's, exactly as bow holds it (reference bowseries.go:59-83,
 * bowgetters.go:46-63): values = Data().Buffers()[1], validity = Data().Buffers()[0]
 * (LSB-first bits, bit i <-> buf[i>>3] & (1<<(i&7)); NULL => all valid), both indexed
 * from Data().Offset().  Buffers are borrowed for the duration of the call only.
 */
#ifndef BOWGPU_H
#define BOWGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 4): bowgpu_options' former padding word is `strict_order` (a caller that left it uninitialised now gets the row-order forms),
 * bowgpu_stream_rw_ceiling became bowgpu_stream_rw_probe, the BOWGPU_ROUTE_* bits moved.  A binding checks bowgpu_abi_version()
 * against the value it was written for when it loads the library (bow_amd/capi.py lib(); shim/go/rolling/gpu_cgo.go init()). */
/* 5 (round 5): bowgpu_rolling_interpolate_aggregate added; bowgpu_last_kernel_name() spells rolling_simple_kernel with its template
 * arguments; bowgpu_agg_info gained nothing (same layout). */
/* 6 (round 6): bowgpu_set_devices / bowgpu_get_devices / bowgpu_set_fanout_min_rows added (one call over several devices); no struct
 * changed. */
#define BOWGPU_ABI_VERSION 6

/* bow.Type (reference bowtypes.go:17-32) */
enum {
    BOWGPU_UNKNOWN = 0,
    BOWGPU_FLOAT64 = 1,
    BOWGPU_INT64 = 2,
    BOWGPU_BOOLEAN = 3,            /* not accepted by the device path */
    BOWGPU_STRING = 4,             /* not accepted by the device path */
    BOWGPU_INPUT_DEPENDENT = 5,
    BOWGPU_ITERATOR_DEPENDENT = 6
};

/* where a buffer lives */
enum {
    BOWGPU_HOST = 0,   /* ordinary (pageable) host memory - Go heap / malloc: staged through HBM by the call */
    BOWGPU_DEVICE = 1, /* HBM of the current device (bowgpu_malloc or any hipMalloc'd pointer) */
    BOWGPU_HOST_PINNED = 2  /* host memory page-locked and mapped for the device with bowgpu_host_register (or hipHostMalloc /
                               hipHostRegister): INPUT columns are read by the kernels where they lie - zero-copy over PCIe, no staging
                               copy, no HBM footprint; OUTPUT columns are produced in HBM and leave by one asynchronous DMA each */
};

/* error codes; the Go shim maps them back to the reference's error strings */
enum {
    BOWGPU_OK = 0,
    BOWGPU_ERR_INTERVAL = -1,        /* "strictly positive interval required"            rolling/rolling.go:115-117 */
    BOWGPU_ERR_TS_TYPE = -2,         /* "impossible to create a new intervalRolling ..."  rolling/rolling.go:70-73 */
    BOWGPU_ERR_FIRST_TS_NULL = -3,   /* "the first value of the column should be ..."     rolling/rolling.go:89-93 */
    BOWGPU_ERR_NO_AGG = -4,          /* "at least one column aggregation is required"    rolling/aggregation.go:148-150 */
    BOWGPU_ERR_KEEP_INTERVAL = -5,   /* "must keep interval column '%s'"                 rolling/aggregation.go:163-166 */
    BOWGPU_ERR_BAD_COL = -6,         /* "no column '%s'"                                 bowgetters.go:323 */
    BOWGPU_ERR_TYPE = -7,            /* type whitelist                                   rolling/interpolation.go:82-93 */
    BOWGPU_ERR_NOT_SORTED = -8,      /* FillLinear: "refColIndex '%d' is empty or not sorted" bowfill.go:39-42 */
    BOWGPU_ERR_UNSUPPORTED = -9,     /* input outside the device path's contract (see each function) */
    BOWGPU_ERR_ARG = -10,
    BOWGPU_ERR_NO_DEVICE = -11,      /* no HIP device / runtime: the product path has no CPU fallback */
    BOWGPU_ERR_HIP = -12,            /* a HIP call failed; message has the hipError string */
    BOWGPU_ERR_TS_NULLS = -13,       /* interval column has nulls AND the call has Mode or is sharded (or, Rolling.Interpolate on inclusive
                                        windows: a row on a window start has null timestamps behind it and then an EQUAL timestamp, or sits on
                                        -1): the device path declines (caller keeps the reference path); Aggregate and Interpolate are served
                                        otherwise. */
    BOWGPU_ERR_TS_UNSORTED = -14,    /* interval column not ascending: device path declines */
    BOWGPU_ERR_OOM = -15
};

/* One Arrow array as bow holds it.  Replaces per-element Bow.GetInt64/GetFloat64/GetValue
 * (reference bowgetters.go:155-247) with bulk buffer access. */
typedef struct bowgpu_col {
    const void *values;       /* int64 / float64 little-endian, 8-byte aligned */
    const uint8_t *validity;  /* may be NULL (all valid) */
    int64_t offset;           /* arrow Data.Offset(): slices share buffers (bow.go:279-283) */
    int64_t length;           /* Data.Len() */
    int64_t null_count;       /* Data.NullN(); -1 = unknown (the library counts) */
    int32_t type;             /* BOWGPU_FLOAT64 | BOWGPU_INT64 */
    int32_t residency;        /* BOWGPU_HOST | BOWGPU_DEVICE | BOWGPU_HOST_PINNED */
} bowgpu_col;

/* Output column: caller-owned storage for `length` slots — what bow.NewBuffer(W, typ)
 * (bowbuffer.go:22-40) would allocate: 8*length value bytes and ceil(length/8) validity
 * bytes.  The call fills both (null slots hold 0, as in the reference) and sets
 * null_count so the shim can wrap them with bow.NewSeries(name, typ, data, validity)
 * (bowseries.go:27-29). */
typedef struct bowgpu_out {
    void *values;
    uint8_t *validity;
    int64_t length;           /* in: capacity in slots (>= W); out: slots produced */
    int64_t null_count;       /* out */
    int32_t type;             /* out: resolved return type (rolling/aggregation.go:110-121) */
    int32_t residency;        /* in */
} bowgpu_out;

/* Built-in reducers: one per constructor of reference rolling/aggregation/<name>.go */
enum {
    BOWGPU_AGG_WINDOW_START = 0,       /* windowstart.go:8-13 */
    BOWGPU_AGG_SUM = 1,                /* sum.go:8-25 */
    BOWGPU_AGG_MEAN = 2,               /* arithmeticmean.go:8-30 */
    BOWGPU_AGG_MIN = 3,                /* minmax.go:8-31 */
    BOWGPU_AGG_MAX = 4,                /* minmax.go:33-56 */
    BOWGPU_AGG_COUNT = 5,              /* count.go:8-20 */
    BOWGPU_AGG_FIRST = 6,              /* firstlast.go:8-21 */
    BOWGPU_AGG_LAST = 7,               /* firstlast.go:23-36 */
    BOWGPU_AGG_INTEGRAL_STEP = 8,      /* integral.go:40-69 */
    BOWGPU_AGG_INTEGRAL_TRAPEZOID = 9, /* integral.go:8-38, NeedInclusiveWindow */
    BOWGPU_AGG_WAVG_STEP = 10,         /* weightedmean.go:8-20 */
    BOWGPU_AGG_WAVG_LINEAR = 11,       /* weightedmean.go:22-34, NeedInclusiveWindow */
    BOWGPU_AGG_NUM_ROWS = 12,          /* float64(w.Bow.NumRows()): the closure of aggregation_test.go:28-31 */
    BOWGPU_AGG_MODE = 13,              /* mode.go:8-32 (not mergeable: unsharded calls only) */
    BOWGPU_AGG__COUNT = 14
};

#define BOWGPU_MAX_FACTORS 4

/* One rolling.ColAggregation (rolling/aggregation.go:40-50) restricted to the built-in
 * kinds, with its transformation.Factor chain (rolling/transformation/factor.go:7-20). */
typedef struct bowgpu_agg {
    int32_t kind;
    int32_t col;                         /* input column index: what SetInputIndex receives (aggregation.go:181) */
    int32_t n_factors;                   /* 0..BOWGPU_MAX_FACTORS */
    int32_t _pad;
    double factors[BOWGPU_MAX_FACTORS];  /* applied in order to the reducer's result (aggregation.go:216-221) */
} bowgpu_agg;

/* rolling.Options (rolling/rolling.go:49-53); PrevRow travels with the interpolators. */
typedef struct bowgpu_options {
    int64_t offset;
    int32_t inclusive;
    int32_t strict_order;       /* not in the reference.  != 0: every window is reduced in the reference's left-to-right row order - no
                                   order-free form, Sum / ArithmeticMean / Integral* / WeightedAverage* bit for bit, and
                                   bowgpu_agg_info.long_windows is 0 on success.  Windows that a tile cannot hold (longer than its
                                   128-row look-ahead) are walked by one lane each (round 4; round 3 declined them): 0.25 - 0.48 of
                                   the HBM peak at 1000-row windows.  A window of more than 2^20 rows makes the call
                                   BOWGPU_ERR_UNSUPPORTED (one lane per window: beyond that it would take milliseconds each).
                                   Honoured by bowgpu_rolling_aggregate[_planned], by bowgpu_rolling_interpolate_aggregate and (round 5) by
                                   the shard record protocol - bowgpu_shard_begin / _pass_begin / _finish: a window shared by TWO ranks
                                   is re-walked by the right rank seeded with the left rank's running state, i.e. in row order across
                                   the boundary; a window spread over three or more ranks, or a boundary window of more than 2^20
                                   rows, is BOWGPU_ERR_UNSUPPORTED.  The round-1 building blocks (bowgpu_shard_aggregate,
                                   _carry_only) decline it.  0: see bowgpu_agg_info.long_windows for the bound that applies */
} bowgpu_options;

/* Diagnostics of one aggregate call */
typedef struct bowgpu_agg_info {
    int64_t s0;                 /* first window start (rolling.go:95-99) */
    int64_t num_windows;        /* W (rolling.go:143-154) */
    int32_t new_interval_col;   /* index of the LAST aggregator reading the interval column (aggregation.go:152-161) */
    int32_t inclusive;          /* effective Options.Inclusive after validateAggregation (aggregation.go:183-185) */
    int64_t long_windows;       /* windows reduced in an ORDER-FREE form instead of the reference's left-to-right walk: windows longer
                                   than a tile's look-ahead (128 rows), and every window of a call whose windows average >= 129 rows (128 for calls with both kinds of integral).  THE STATED TOLERANCE, for the float sums
                                   of those windows (u = 2^-53, n = the window's rows, x = its valid values, T = its time-weighted
                                   terms):   |Sum - ref| <= 2 (n + 2) u SUM|x_i|;   ArithmeticMean: that / count + 2 u |ref|;
                                   Integral*: 4 (n + 2) u SUM|T_i| (a Factor scales it);   WeightedAverage*: that / (t_last - t_first)
                                   + 2 u |ref|.  A bound on SUM|x|, not on |ref|: a window whose values cancel may differ from the
                                   reference in every digit of a result near zero (tests/tolerance.py states and asserts it, a
                                   cancelling window included).  The summation tree is fixed: equal inputs give equal bits.  Every
                                   other reducer, and every window when this is 0, is bit-exact.  bowgpu_options.strict_order
                                   walks those windows in row order instead.
                                   WHAT "BIT-EXACT" EXCLUDES, everywhere in this header: the sign and payload of a NaN that the
                                   arithmetic itself GENERATES (inf - inf, 0 * inf, 0 / 0 in a Sum, a mean, an integral, an
                                   interpolated value).  gfx950 produces the positive default NaN 0x7FF8000000000000 where x86 - the
                                   reference's Go on amd64 - produces the negative "real indefinite" 0xFFF8000000000000; Go prints
                                   both as NaN and no operation on this path tells them apart.  A NaN that comes FROM THE INPUT keeps
                                   its bits (payload and sign) through First / Last / Min / Max / fills / interpolation copies, and a
                                   NaN operand propagates through an addition as on x86 (quieted, payload kept). */
    double kernel_ms;           /* device time of the kernels of this call (HIP events on the library stream) */
} bowgpu_agg_info;

/* ---- library / device ------------------------------------------------------------- */
int bowgpu_abi_version(void);
const char *bowgpu_last_error(void);
int bowgpu_device_count(int *count);
int bowgpu_set_device(int device);          /* per calling thread; default device 0 */
int bowgpu_device_name(char *buf, int cap);
/* ONE call, N devices (reference rolling/aggregation.go:123-145 - the user makes one Aggregate call).  Process-wide:
 * after bowgpu_set_devices(ids, n >= 2) every bowgpu_rolling_aggregate / _planned / _interpolate_aggregate call of any thread whose interval
 * column holds at least 2 x min_rows rows is cut into up to n row ranges (multiples of 4096 rows; never fewer than min_rows rows each), one per
 * listed device, and runs as the shard record protocol below (bowgpu_shard_begin -> records -> bowgpu_shard_finish, the carry-in stitch in
 * row order) on one persistent library thread per list entry.  The records are exchanged in host memory - one process holds them all: no
 * collective, no transport.  Each rank puts the windows it owns at their places in the caller's buffers, so the outputs, null counts and
 * bowgpu_agg_info are what the one-device call gives, bit for bit (info.kernel_ms: the slowest rank's; info.long_windows: the ranks' sum).
 *   Served: HOST columns (each rank stages its own rows once, over its own device's host link) and HOST_PINNED columns (each device reads
 * its range in place) with host-resident outputs, on any list of devices; DEVICE-resident columns or outputs only when every listed device is
 * the calling thread's device (the same id may be listed several times - which is also how a one-GPU box exercises the path).  Calls the
 * record protocol declines - aggregation.Mode, more than 16 aggregators, an interval column with nulls, strict_order with a window over
 * three ranks - and calls with too few rows take the one-device path of the calling thread as before; so does everything else in this
 * header.  Fanned-out calls are serialised process-wide (the devices are busy with one anyway).
 *   n <= 1 (or ids == NULL, n == 0) switches the fan-out off.  min_rows: default 2^20 (bowgpu_set_fanout_min_rows).  A process that never
 * calls bowgpu_set_devices starts with the list BOWGPU_DEVICES="0,1,..." names (and BOWGPU_FANOUT_MIN_ROWS), read once - for running an
 * unmodified program through the fan-out; nothing on the call path reads the environment. */
int bowgpu_set_devices(const int *ids, int n);
int bowgpu_get_devices(int *ids, int cap, int *n);   /* the list in force (n = 0: off); ids may be NULL */
int bowgpu_set_fanout_min_rows(int64_t rows);
/* row ranges (= library threads, = entries of the device list) that served the calling thread's last bowgpu_rolling_aggregate / _planned /
 * _interpolate_aggregate call; 1: the one-device path */
int bowgpu_last_call_ranks(int *ranks);
/* process-wide since load: Rolling.Aggregate calls that arrived while a list of two or more devices was in force, and how many of them ran as
 * row ranges (the rest: what the fan-out declines, served by the one-device path) */
int bowgpu_fanout_counts(int64_t *calls, int64_t *served);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all work of this
 * thread; NULL restores the library's own stream. */
int bowgpu_set_stream(void *hip_stream);
int bowgpu_synchronize(void);
/* Device state is per calling thread (stream, pools, a cache of freed scratch blocks: at most 24 blocks / 24 GB per thread);
 * a thread that exits releases its own.  bowgpu_trim frees cached scratch blocks now: the calling thread's (all_threads = 0)
 * or every thread's.  A failed allocation inside the library trims every thread's cache itself before reporting BOWGPU_ERR_OOM. */
int bowgpu_trim(int32_t all_threads, int64_t *bytes_freed /* nullable */);
/* free / total HBM of the calling thread's device (hipMemGetInfo): lets a host size its shards (288 GB per MI355X) */
int bowgpu_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* device time (HIP events on the stream) of the tile kernel of this thread's last aggregate call */
int bowgpu_last_kernel_ms(double *ms);
/* Rows of this thread's last Rolling.Aggregate / Rolling.Interpolate (_fill) call that were served by one of the two kernels the
 * library keeps for the shapes its fast kernels decline - rolling_agg_kernel (window ids beyond 32 bits inside one tile, tiles denser
 * than a head list, intervals >= 2^32 with time-weighted reducers: about 0.3 of the HBM peak where the wave-tile kernels reach
 * 0.6 - 0.7) and interp_tile_kernel (64-bit window ids, dropped rows, the -1 sentinel window, a trip whose lists overflow: half the
 * rate of interp_wave3_kernel).  0 for the usual call; the results are the same bits either way.  A caller that finds this non-zero on
 * its hot path is on a route 2 - 3x slower than the figures the documentation quotes. */
int bowgpu_last_call_slow_rows(int64_t *rows);
/* ... and which tile kernel that was ("rolling_tw_kernel", "rolling_wave_kernel", "rolling_agg_kernel", "long_stream_kernel", ...; "" before
 * any call).  rolling_simple_kernel comes with its template arguments, spelled as rocprofv3 prints them
 * ("rolling_simple_kernel<0, false, false, false, false, false, false>"): the text up to '<' is the kernel, the whole string the
 * instantiation that ran - what bench.py matches against the committed counter files (profiles/) and the per-kernel code hashes
 * the build leaves next to the library (libbowgpu.kernel_sha.json) */
const char *bowgpu_last_kernel_name(void);

/* ---- HBM buffers (so callers can keep columns resident between calls) ------------- */
int bowgpu_malloc(void **ptr, int64_t bytes);
int bowgpu_free(void *ptr);
int bowgpu_memcpy_h2d(void *dst, const void *src, int64_t bytes);
int bowgpu_memcpy_d2h(void *dst, const void *src, int64_t bytes);
int bowgpu_memset(void *dst, int value, int64_t bytes);

/* Page-lock `bytes` of host memory at `ptr` and map them for the device, so that columns inside the range may be passed with
 * residency BOWGPU_HOST_PINNED (reference bowseries.go:59-83 hands Go-heap slices: a cgo shim registers the Arrow buffers of a Bow
 * it keeps using - Bows are immutable - once, e.g. from a finalizer-paired constructor).  Registration costs about a millisecond
 * per 4 MB (every page is touched and locked): do it per buffer, not per call.  The range must stay valid until unregistered. */
int bowgpu_host_register(void *ptr, int64_t bytes);
int bowgpu_host_unregister(void *ptr);

/* ---- stream timers (HIP events on the stream the kernels are launched on) --------- */
int bowgpu_timer_create(void **timer);
int bowgpu_timer_start(void *timer);
int bowgpu_timer_stop(void *timer);
int bowgpu_timer_elapsed_ms(void *timer, double *ms); /* synchronises on the stop event */
int bowgpu_timer_destroy(void *timer);

/* ---- rolling.IntervalRolling ------------------------------------------------------- */

/* enforceIntervalAndOffset — reference rolling/rolling.go:114-128 */
int bowgpu_enforce_interval_and_offset(int64_t interval, int64_t offset, int64_t *offset_out);

/* newIntervalRolling + countWindows — reference rolling/rolling.go:69-112, :143-154.
 * `offset` is the raw Options.Offset.  Outputs the first window start and numWindows. */
int bowgpu_plan_windows(const bowgpu_col *ts, int64_t interval, int64_t offset, int64_t *s0,
                        int64_t *num_windows);

/* The same plan as one record, for hosts that keep it: the reference computes it ONCE, in the constructor
 * (newIntervalRolling, rolling/rolling.go:69-112, stores the first window start and numWindows in the intervalRolling), and
 * every later Aggregate / Interpolate on that Rolling uses it.  first_ts / last_ts are the two scalars the plan was made from.
 * With device-resident columns the plan costs one round trip to the GPU; bowgpu_rolling_aggregate_planned skips it. */
typedef struct bowgpu_plan {
    int64_t s0, num_windows;
    int64_t first_ts, last_ts;
    int64_t interval, offset;      /* offset: normalised (enforceIntervalAndOffset) */
    int64_t nrows;                 /* of the interval column the plan was made for */
} bowgpu_plan;
int bowgpu_plan_windows_ex(const bowgpu_col *ts, int64_t interval, int64_t offset, bowgpu_plan *plan);

/* Rolling.Aggregate — reference rolling/aggregation.go:123-238 (indexedAggregations,
 * validateAggregation, aggregateWindows) fused with the reducers of rolling/aggregation/.
 * One pass over the interval column buckets rows into windows; every aggregator of every
 * value column is reduced in that same pass (any number of aggregators and columns: beyond 16 outputs /
 * 8 column passes the call is cut into launches internally; aggregation.Mode runs as a pass of its own).
 *   cols[ncols]   the Bow's columns (only those referenced by ts_col / aggs are touched)
 *   outs[naggs]   one output column per aggregator, in aggregator order (A.7)
 * Device-path contract: interval column Int64, its valid values ascending (else BOWGPU_ERR_TS_UNSORTED: the caller keeps the
 * reference's own path); value columns Float64 or Int64.  Nulls in the interval column (rolling.go:190-193: such a row neither
 * ends nor extends a window, yet lies inside its window's slice when rows of the same window surround it) are served for
 * exclusive and inclusive iterations alike (incl. rolling.go:214-218's `rowIndex - 1` when a null follows an inclusive row);
 * with Mode the call is BOWGPU_ERR_TS_NULLS.  Cost of such a call: one extra pass over the interval column (16 bytes per row) that
 * writes a forward-filled copy of it (8 bytes per row) and n / 8 bytes per rewritten validity class (up to three per value column on
 * an inclusive iteration); every output goes through a device temporary (W * 8 bytes each) before it reaches the caller's buffer;
 * NumRows is counted as Count over the rows that belong to a window and converted to float64 in a pass of its own. */
int bowgpu_rolling_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col,
                             int64_t interval, const bowgpu_options *opts,
                             const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                             bowgpu_agg_info *info);

/* ... with the plan the host already holds (bowgpu_plan_windows_ex on THIS interval column; a plan whose nrows differs from the
 * column's length is rejected).  opts->offset is ignored in favour of the plan's. */
int bowgpu_rolling_aggregate_planned(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_plan *plan,
                                     const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs,
                                     bowgpu_out *outs, bowgpu_agg_info *info);

/* intervalRolling.Next for every window at once — reference rolling/rolling.go:177-239.
 * For window k: first_index[k] = Window.FirstIndex, [slice_begin[k], slice_end[k]) = the rows
 * of Window.Bow (0,0 when empty), is_inclusive[k] = Window.IsInclusive.  Arrays hold W
 * entries and may be NULL; residency applies to all of them. */
int bowgpu_window_bounds(const bowgpu_col *ts, int64_t interval, const bowgpu_options *opts,
                         int64_t *first_index, int64_t *slice_begin, int64_t *slice_end,
                         uint8_t *is_inclusive, int32_t residency);

/* aggregation.Aggregate (whole frame) — reference rolling/aggregation/whole.go:12-93.
 * outs[i].length is 1 (0 for an empty Bow). */
int bowgpu_aggregate_whole(const bowgpu_col *cols, int32_t ncols, int32_t ts_col,
                           const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs);

/* ---- Rolling.Interpolate ----------------------------------------------------------- */
enum {
    BOWGPU_INTERP_WINDOW_START = 0,  /* interpolation/windowstart.go:8-14 */
    BOWGPU_INTERP_LINEAR = 1,        /* interpolation/linear.go:8-38 */
    BOWGPU_INTERP_STEP_PREVIOUS = 2, /* interpolation/stepprevious.go:8-26 */
    BOWGPU_INTERP_NONE = 3,          /* interpolation/none.go:7-13 */
    BOWGPU_INTERP_CONST = 4          /* constant-valued closure (interpolation_test.go:16-19) */
};

/* One rolling.ColInterpolation (rolling/interpolation.go:10-28) of a built-in kind plus what
 * it reads from Options.PrevRow (linear.go:14-18, stepprevious.go:13-15). */
typedef struct bowgpu_interp {
    int32_t kind;
    int32_t col;
    double const_value;
    int32_t has_prev_row;
    int32_t prev_t_valid;
    int32_t prev_v_valid;
    int32_t _pad;
    double prev_t;      /* prevRow.GetFloat64(intervalCol, last) */
    double prev_v;      /* prevRow.GetFloat64(col, last) */
    int64_t prev_v_i64; /* prevRow.GetValue(col, last) for Int64 columns (StepPrevious) */
} bowgpu_interp;

/* Rolling.Interpolate — reference rolling/interpolation.go:30-161.  Two calls: _count gives
 * the number of output rows (N + windows missing their start), _fill writes them.
 * interps must list the Bow's columns in order (bowappend.go:11-13 needs equal schemas).
 * CONTRACT between the two calls: a _fill that directly follows the _count of the same DEVICE-resident interval column (same
 * pointer, offset, length, interval, options, calling thread) reuses what the count pass learnt about it and does not scan it
 * again - the column must not be modified or freed in between (Bows are immutable in the reference; a cgo shim makes the two
 * calls back to back inside Rolling.Interpolate).  Writes and frees made THROUGH this library (bowgpu_free, bowgpu_memcpy_h2d,
 * bowgpu_memset, the generators) drop the reuse by themselves; a buffer rewritten by the caller's own kernels is the caller's to
 * keep unchanged.  As a last line the fill compares the number of rows it produces with the count: a mismatch is
 * BOWGPU_ERR_ARG and the outputs are to be discarded.
 * _fill does not NEED a preceding _count: called on its own it sizes the outputs itself against bowgpu_out.length (in: capacity;
 * rows + windows always suffice; too small is BOWGPU_ERR_ARG naming the size).  Inclusive windows then take one pass over the rows
 * (n_out = n + W - [row 0 on its window's start] needs no count); exclusive windows make their own count pass first.
 * DEVICE-resident outputs are written in place - no working copy of their bitmaps, no launch that zeroes or counts them - when each
 * validity pointer is 4-byte aligned and the capacity (bowgpu_out.length on entry) reaches the end of the 32-bit word that holds row
 * n_out - 1, i.e. ceil(length / 8) >= 4 * ceil(n_out / 32): allocate a multiple of 32 rows (Arrow's allocators pad to 64 bytes, 512
 * rows).  Outputs that do not qualify cost one more small launch and a copy of the bitmaps (0.03 ms per 1e8 rows), nothing else.
 * Input columns whose value pointers are not all 16-byte aligned (a slice at an odd row offset) take 8-byte loads in the kernel's
 * general instantiation.
 * An interval column WITH NULLS: the output is the windows' slices - rows that belong to no window vanish (rolling.go:190-193,
 * :224-228), null-timestamp rows inside a slice are copied as they are; inclusive windows too (incl. rolling.go:214-218's
 * `rowIndex - 1` after a null), except for two shapes that are BOWGPU_ERR_TS_NULLS (see the error code).  Any number
 * of columns (round 6: beyond 16 the value columns go in groups of 15, each with the interval column).  Cost: the call is made on the
 * KEPT rows, compacted into device temporaries (8 bytes per row and column + n / 8 bytes per bitmap); a _fill that follows its _count on
 * the same DEVICE-resident columns (the contract above) uses what the count built, any other _fill builds it itself.
 */
int bowgpu_rolling_interpolate_count(const bowgpu_col *cols, int32_t ncols, int32_t ts_col,
                                     int64_t interval, const bowgpu_options *opts,
                                     const bowgpu_interp *interps, int32_t ninterps,
                                     int64_t *n_out);
int bowgpu_rolling_interpolate_fill(const bowgpu_col *cols, int32_t ncols, int32_t ts_col,
                                    int64_t interval, const bowgpu_options *opts,
                                    const bowgpu_interp *interps, int32_t ninterps,
                                    bowgpu_out *outs);

/* Row-range sharded Rolling.Interpolate: a shard of the frame plus what lies beyond its two ends.
 * Synthetic rows sit in front of a window's first row, so the shard holding that row emits them - including the empty windows
 * since the last row on the shards to its left (has_left / left_last_ts).  A Linear / StepPrevious interpolator whose nearest
 * valid point lies on another shard gets it from the caller: the point to the LEFT through bowgpu_interp.prev_* (the reference's
 * own Options.PrevRow mechanism), the point to the RIGHT through next_*.  Shards concatenated in rank order = the unsharded
 * result, for exclusive and for inclusive windows (opts->inclusive: a row on its window's start is preceded by the copy of itself
 * that closes the window before - the copy travels with its row).  Negative timestamps: negative window starts are served (the
 * window that starts at -1, the reference's "no first value" sentinel, by the shard that accounts for it); rows below the first
 * window start (rolling.go:96-99) by the frame's first shard when the first row at or above s0 is its own - a first shard of nothing
 * but such rows is declined (BOWGPU_ERR_UNSUPPORTED). */
typedef struct bowgpu_interp_edge {
    int32_t has_left;         /* a shard to the left holds rows */
    int32_t _pad;
    int64_t left_last_ts;     /* the last of them */
    int32_t next_valid[8];    /* per column of the Bow: nearest valid point on the shards to the right (Linear) */
    double next_t[8], next_v[8];
} bowgpu_interp_edge;
/* a shard's own first / last valid point per column: what its neighbours fill their edges from (all_gather'ed as bytes) */
typedef struct bowgpu_interp_points {
    int64_t nrows, first_ts, last_ts;
    int32_t first_valid[8], last_valid[8];
    double first_t[8], first_v[8], last_t[8], last_v[8];
    int64_t last_v_i64[8];    /* raw payload of the last valid value of an Int64 column (StepPrevious keeps the integer) */
} bowgpu_interp_points;
int bowgpu_shard_interp_points(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, bowgpu_interp_points *out);
int bowgpu_shard_interpolate_count(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                                   const bowgpu_options *opts, int64_t global_s0, const bowgpu_interp *interps, int32_t ninterps,
                                   const bowgpu_interp_edge *edge, int64_t *n_out);
int bowgpu_shard_interpolate_fill(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                                  const bowgpu_options *opts, int64_t global_s0, const bowgpu_interp *interps, int32_t ninterps,
                                  const bowgpu_interp_edge *edge, bowgpu_out *outs);

/* ---- Rolling.Interpolate(...).Aggregate(...) without the interpolated frame ---------- */

/* r.Interpolate(interps...).Aggregate(aggs...) - reference rolling/interpolation.go:30-69 (returns a Rolling over the interpolated
 * Bow with the same interval and options) + rolling/aggregation.go:123-145 (consumes it) - when the caller wants only the aggregated
 * Bow: the interpolated frame (rows + one synthetic row per window that does not start on its first row, interpolation.go:98-161) is
 * never handed back and, where the shape allows it, never written.  interps: one per column of the Bow, in column order (as for
 * bowgpu_rolling_interpolate_*); aggs[i].col indexes those same columns - the interpolated Bow has the input's columns and types
 * (interpolation.go:139-155).  outs / info: as bowgpu_rolling_aggregate on the Rolling Interpolate returns.
 *   ONE pass over the rows (rolling_fused.hip: a window's synthetic value - linear.go:34-35's expression on the nearest both-valid
 * rows either side of the window's FirstIndex - is fed as the window's first row to the same left-to-right walk) for: exclusive
 * windows; WindowStart / Sum / ArithmeticMean / Min / Max / Count / First / Last / NumRows; an interval column without nulls under
 * interpolation.WindowStart; value columns under Linear / StepPrevious / None (/ WindowStart); a frame that starts at or above 0 and
 * spans less than 2^32 from its first window start; windows of 4 .. 128 rows on average, none longer than 128 rows.
 *   Everything else (and any call some tile of which the fused kernel cannot describe): bowgpu_rolling_interpolate_count + _fill into
 * device temporaries, then bowgpu_rolling_aggregate on them - what the two calls give, bit for bit, including their declines
 * (BOWGPU_ERR_TS_UNSORTED, ...).  Errors: newIntervalRolling's first, then Interpolate's (BOWGPU_ERR_TYPE "accepts types ...",
 * BOWGPU_ERR_KEEP_INTERVAL), then Aggregate's.  Both forms are bit-identical to the two calls made one after the other
 * (tests/test_gpu_fused.py: every window against oracle interpolate -> aggregate, and against the two-call path). */
int bowgpu_rolling_interpolate_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                                         const bowgpu_options *opts, const bowgpu_interp *interps, int32_t ninterps,
                                         const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, bowgpu_agg_info *info);

/* ---- Bow.FillLinear / IsColSorted -------------------------------------------------- */

/* Bow.FillLinear — reference bowfill.go:14-103.  out receives the filled copy of
 * cols[fill_col]; *unchanged = 1 where the reference returns the receiver itself
 * (bowfill.go:35-37, :53-55). */
int bowgpu_fill_linear(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col,
                       bowgpu_out *out, int32_t *unchanged);
/* The same for a caller whose own checks have run - reference bowfill.go:35-42: the ref column holds a value and IsColSorted says
 * yes (the cgo shim's hook sits behind them; its IsColSorted is bowgpu_is_col_sorted): the pass over the ref column that establishes
 * both (a quarter of the call at 1e8 rows) is not made a second time.  On a ref column that is NOT sorted the filled values are
 * whatever linear.go's expression gives between the neighbours found - no error, no out-of-bounds access. */
int bowgpu_fill_linear_sorted(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col,
                              bowgpu_out *out, int32_t *unchanged);

/* Bow.FillPrevious / Bow.FillNext (LOCF / NOCB) — reference bowfill.go:162-253 — and Bow.FillMean — reference
 * bowfill.go:105-160 — of ONE Int64 / Float64 column (the reference loops over the selected columns, one goroutine each;
 * call once per column).  out receives the filled copy; *unchanged = 1 when the column has no nulls (the reference
 * passes such a column through, bowfill.go:130-133, :176-179).  Boolean / String columns: BOWGPU_ERR_UNSUPPORTED. */
#define BOWGPU_FILL_PREVIOUS 0
#define BOWGPU_FILL_NEXT 1
#define BOWGPU_FILL_MEAN 2
int bowgpu_fill(const bowgpu_col *col, int32_t method, bowgpu_out *out, int32_t *unchanged);

/* Bow.IsColSorted — reference bowassertion.go:15-81 (ascending OR descending, nulls skipped,
 * empty => false). */
int bowgpu_is_col_sorted(const bowgpu_col *col, int32_t *sorted);

/* ---- Parquet column chunk -> device column --------------------------- */

/* Reads INT64 / DOUBLE columns of a Parquet file the way the reference's NewBowFromParquet does (bowparquet.go:44-153), but
 * decodes on the device: the column's compressed pages are uploaded as they lie in the file and decompressed (Snappy), their
 * definition levels turned into an Arrow validity bitmap and their PLAIN values scattered to row slots by HIP kernels.
 * Read: flat schemas, OPTIONAL / REQUIRED columns, UNCOMPRESSED / SNAPPY, PLAIN and dictionary-encoded values (PLAIN_DICTIONARY /
 * RLE_DICTIONARY, with PLAIN fall-back pages), RLE definition levels, data page v1 and v2 - what the reference writes
 * (bowparquet.go:326-338) and what pyarrow / pandas write by default.  Anything else: BOWGPU_ERR_UNSUPPORTED. */
typedef struct bowgpu_parquet bowgpu_parquet;
int bowgpu_parquet_open(const char *path, bowgpu_parquet **handle);
int bowgpu_parquet_close(bowgpu_parquet *handle);
int bowgpu_parquet_info(const bowgpu_parquet *handle, int64_t *num_rows, int32_t *num_columns);
/* name (NUL-terminated, truncated to name_cap), type (BOWGPU_INT64 / BOWGPU_FLOAT64 / BOWGPU_BOOLEAN / BOWGPU_STRING / -1) and
 * whether the column is OPTIONAL (may hold nulls) */
int bowgpu_parquet_column(const bowgpu_parquet *handle, int32_t i, char *name, int32_t name_cap, int32_t *type, int32_t *optional);
/* out: num_rows slots (HOST or DEVICE residency); values, validity, null_count and type are filled in */
int bowgpu_parquet_read_column(bowgpu_parquet *handle, int32_t i, bowgpu_out *out);

/* ---- row-range sharding across GPUs ----------------------------------- */

/* Running state of one reducer over the rows a rank holds of a window that straddles a shard
 * boundary: what the reference's loops carry from one row to the next (sum.go:15-22,
 * arithmeticmean.go:15-24, minmax.go:14-28, count.go:12-18, firstlast.go).  Fixed size so a
 * rank's record travels through one RCCL all_gather; nn_min / nn_max (NaN-ignoring extrema) are
 * only consulted when more than two ranks share one window. */
typedef struct bowgpu_carry_state {
    double sum;
    double vmin, vmax;
    double nn_min, nn_max;
    uint64_t first_bits, last_bits;   /* First / Last raw 64-bit payloads */
    int64_t count;                    /* valid values */
    int64_t nrows;                    /* rows (w.Bow.NumRows() contribution) */
    /* time-weighted reducers (integral.go:14-31, :46-62): last and first both-valid point, running sums */
    double pt, pv, first_pt, first_pv;
    double integ_step, integ_trap;
    int32_t has_value;
    int32_t has_nn;
    int32_t has_point;
    int32_t has_pair;
} bowgpu_carry_state;

/* The first row of the next non-empty shard to the right, per aggregator: what an INCLUSIVE window that ends exactly where
 * this shard's rows end needs from its successor (rolling.go:201-209).  Fixed size: it rides in the same all_gather as
 * (first_ts, last_ts, nrows). */
typedef struct bowgpu_next_row {
    int32_t present;                  /* 0: there is no row to the right */
    int32_t _pad;
    int64_t ts;
    uint64_t bits[16];                /* value of each aggregator's input column at that row */
    int32_t valid[16];
} bowgpu_next_row;

#define BOWGPU_CARRY_MAX_AGGS 16
typedef struct bowgpu_shard_carry {
    int64_t first_window_id;   /* global id of the FIRST window with a row in this shard; -1: empty shard */
    int64_t last_window_id;    /* global id of the LAST one */
    int64_t first_ts, last_ts;
    int64_t nrows;
    int32_t naggs;
    int32_t _pad;
    bowgpu_carry_state last[BOWGPU_CARRY_MAX_AGGS];  /* per aggregator: state of the LAST window over this shard's rows */
} bowgpu_shard_carry;

/* first / last timestamp and row count of a shard's interval column: what the ranks exchange first (one round trip to the device) */
int bowgpu_shard_span(const bowgpu_col *ts, int64_t *first_ts, int64_t *last_ts, int64_t *nrows);

/* Sharded Rolling.Aggregate, phase 1.  This rank holds rows [row0, row0+len) of every column
 * (device-resident; outs device-resident).  global_s0 comes from bowgpu_plan_windows on the rank
 * that holds global row 0.  Reduces every window that has a row in the shard - output slot k is
 * global window first_window_id + k - treating the shard's first row as a window start, and
 * exports in *carry the running state of its last window.  All reducers; when some reducer needs inclusive windows
 * (IntegralTrapezoid, WeightedAverageLinear) pass next_row = the first row of the next non-empty shard to the right
 * (bowgpu_shard_first_row there), and finish_last = 1 when this shard owns its last window and that window is not
 * also its first one shared with ranks to the left (then bowgpu_shard_fix_first folds the row in).
 * lead_empty_windows: number of EMPTY windows between the left neighbour's last window and this
 * shard's first one that this rank also outputs (known after the ranks exchanged first/last ts):
 * output slot k is then global window first_window_id - lead_empty_windows + k. */
int bowgpu_shard_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                           const bowgpu_options *opts, int64_t global_s0, int32_t holds_global_row0,
                           int64_t lead_empty_windows,
                           const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                           bowgpu_shard_carry *carry,
                           const bowgpu_next_row *next_row /* nullable */, int32_t finish_last);

/* The carry bowgpu_shard_aggregate would export, WITHOUT reducing the shard: only the rows of the shard's last window are read.
 * Lets a caller put the carry exchange in flight before the shard's main pass (the exchange is latency, the pass is milliseconds).
 * For calls without inclusive reducers (with them the last window's state depends on the neighbour's first row: use the carry
 * bowgpu_shard_aggregate returns). */
int bowgpu_shard_carry_only(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                            const bowgpu_options *opts, int64_t global_s0, int32_t holds_global_row0,
                            const bowgpu_agg *aggs, int32_t naggs, bowgpu_shard_carry *carry);

/* This shard's first row in the layout bowgpu_shard_aggregate / _fix_first of the LEFT neighbour expect. */
int bowgpu_shard_first_row(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_agg *aggs, int32_t naggs,
                           bowgpu_next_row *out);

/* Phase 2, after the ranks exchanged their carries.  seeds[i] is the state of this shard's first
 * window accumulated over the rows the LEFT ranks hold (one rank: its carry as is; several:
 * bowgpu_carry_merge in rank order).  Re-walks the shard's rows of that window seeded with it - so
 * the straddling window is reduced in the reference's row order - and rewrites output slot 0. */
int bowgpu_shard_fix_first(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                           const bowgpu_options *opts, int64_t global_s0,
                           int64_t lead_empty_windows,
                           const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                           int64_t first_window_id, const bowgpu_carry_state *seeds,
                           bowgpu_carry_state *merged_out /* nullable: state after this shard's rows */,
                           const bowgpu_next_row *next_row /* nullable: used when the window is also the shard's last */);

/* left (earlier rows) then right: the state of the concatenation.  Pure bookkeeping on two
 * records (no column data); Sum is left.sum + right.sum, i.e. NOT the row-order association -
 * only used when one window spans three or more ranks. */
int bowgpu_carry_merge(const bowgpu_carry_state *left, const bowgpu_carry_state *right,
                       bowgpu_carry_state *out);

/* ---- the shard protocol behind the ABI: begin -> ONE exchange -> finish ------------------------------------------
 * What a host (the cgo shim inside reference rolling/aggregation.go:123-145, or bow_amd/sharded.py here) does per
 * Rolling.Aggregate over R ranks:
 *     bowgpu_shard_begin(my columns)            -> my record             (fixed size)
 *     all_gather of the records as bytes        (RCCL over xGMI / any transport: the ONLY exchange of the call)
 *     bowgpu_shard_finish(my columns, records)  -> my output slots + which global windows they are
 * All ownership rules (which rank outputs a window that straddles a boundary, who emits the empty windows between two
 * shards, whose running state seeds whose first window, who needs whose first row for an inclusive window) are decided
 * inside the library from the gathered records; every rank derives the same decisions from the same bytes. */
typedef struct bowgpu_shard_record {
    int64_t nrows, first_ts, last_ts;      /* of this rank's interval column (0 rows: the rest is zero) */
    int64_t carry_from_ts;                 /* the running states below cover this rank's rows with ts >= carry_from_ts:
                                              the start, on the offset-aligned window grid, of the window holding last_ts
                                              (INT64_MIN: all of the rank's rows) */
    int32_t naggs;
    int32_t flags;                         /* bit 0: built with the global first window start known (second attempt) */
    bowgpu_next_row first_row;             /* this rank's first row (filled when some reducer needs inclusive windows) */
    bowgpu_carry_state last[BOWGPU_CARRY_MAX_AGGS];  /* per aggregator: running state of the rank's LAST window over its rows */
} bowgpu_shard_record;

/* What bowgpu_shard_plan decides for one rank. */
typedef struct bowgpu_shard_decision {
    int64_t s0;                     /* global first window start (rolling.go:95-99 on the first row of the first non-empty rank) */
    int64_t num_windows;            /* global W (rolling.go:143-154) */
    int64_t first_window_id;        /* global id of the first / last window with a row on this rank; -1: no rows */
    int64_t last_window_id;
    int64_t lead_empty_windows;     /* empty windows in front of first_window_id that this rank outputs too */
    int64_t first_slot_window_id;   /* global window id of output slot 0 (= first_window_id - lead_empty_windows); -1: none */
    int64_t windows_local;          /* output slots this rank writes */
    int64_t windows_owned;          /* ... of which it owns the first windows_owned (a last window that continues on a rank
                                       to the right belongs to that rank) */
    int32_t holds_global_row0;
    int32_t drops_last;
    int32_t seed_first_rank;        /* ranks seed_first_rank .. rank-1 hold earlier rows of this rank's first window; -1: none */
    int32_t next_rank;              /* next rank to the right that holds rows; -1: none */
    int32_t finish_last;            /* this rank folds next_rank's first row into its last window during the pass */
    int32_t retry_with_s0;          /* 1: the records cannot settle window 0 (rows below s0, rolling.go:96-99 with negative
                                       timestamps, split across ranks): run bowgpu_shard_begin again with &s0 and exchange again */
} bowgpu_shard_decision;

#define BOWGPU_SHARD_RETRY 1   /* bowgpu_shard_finish: nothing was computed; see retry_with_s0 */
#define BOWGPU_SHARD_PASS_DECLINED 1   /* bowgpu_shard_pass_begin: nothing was enqueued (not an error); bowgpu_shard_finish runs the pass */

/* Columns and outputs of the three calls below: any residency since ABI 5.  Host-resident ones are staged through HBM per call
 * the way bowgpu_rolling_aggregate stages them (so begin + pass_begin move each column over PCIe twice, once for the record and
 * once for the pass; a pass put in flight keeps its staged copies until bowgpu_shard_finish collects it).  The one-call-per-phase
 * building blocks above (bowgpu_shard_aggregate, _fix_first, _shard_span) keep taking device memory only.
 * global_s0: NULL on the first attempt. */
int bowgpu_shard_begin(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                       const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs,
                       const int64_t *global_s0, bowgpu_shard_record *record);
/* The exchange off the critical path (comms row): between bowgpu_shard_begin and bowgpu_shard_finish a host may call
 *     bowgpu_shard_pass_begin(my columns, my outputs, my record)
 * which ENQUEUES the rank's pass over its rows and returns at once - the all_gather of the records then travels while the kernel
 * runs.  The pass assumes what is true of every rank but the odd one: no empty windows between its left neighbour's rows and its
 * own, no row below the frame's first window start.  bowgpu_shard_finish on the same thread collects it when the gathered records
 * agree (same columns, outputs, options, aggregators) and discards it otherwise; results are the same bytes either way.  At most
 * one pass per thread is in flight; no other bowgpu call of this thread may come between the two except bowgpu_shard_plan.
 * Returns 0, BOWGPU_SHARD_PASS_DECLINED (empty shard, negative timestamps: nothing enqueued), or a negative error. */
int bowgpu_shard_pass_begin(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                            const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                            const bowgpu_shard_record *my_record);
/* Pure host arithmetic on the gathered records (no device, no column data): usable from any process. */
int bowgpu_shard_plan(const bowgpu_shard_record *records, int32_t world, int32_t rank, int64_t interval,
                      int64_t raw_offset, bowgpu_shard_decision *out);
/* The rank's pass + stitch.  outs: any residency (the ones given to bowgpu_shard_pass_begin), capacity >= (last_ts - first_ts) / interval + 2 + lead (or simply
 * global W).  Returns 0, BOWGPU_SHARD_RETRY, or a negative error. */
int bowgpu_shard_finish(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                        const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                        const bowgpu_shard_record *records, int32_t world, int32_t rank,
                        bowgpu_shard_decision *decision /* nullable */, bowgpu_agg_info *info /* nullable */);

/* ---- synthetic inputs generated in HBM -------------------------------- */

/* cfg-dense: ts[i] = row0+i, val[i] = u01(mix64(seed, row0+i)); device pointers. */
int bowgpu_gen_dense(int64_t row0, int64_t n, uint64_t seed, int64_t *ts_dev, double *val_dev);
/* cfg-sparse: ts = 10*i + U{0..9}, val = U{0..9}+0.5, P(valid) = 0.7; validity_dev holds
 * ceil(n/8) bytes, bit i = row row0+i (row0 must be a multiple of 8). */
int bowgpu_gen_sparse(int64_t row0, int64_t n, uint64_t seed, int64_t *ts_dev, double *val_dev,
                      uint8_t *validity_dev);
/* The "achievable" line next to the 8 TB/s peak: best rate (GB/s) of a trivial streaming sum over two device
 * buffers of bytes_each bytes (16-byte aligned), tried in a few launch shapes.  Measurement aid for bench.py. */
int bowgpu_stream_read_ceiling(const void *dev_a, const void *dev_b, int64_t bytes_each, double *gb_per_s);
/* The achievable line for the benched traffic MIX: the same two 8-byte-per-row input streams read by a trivial kernel that also
 * writes two output streams of 8 bytes per `rows_per_slot` rows (out_a / out_b: device buffers of ceil(rows / rows_per_slot)
 * 8-byte slots) in the tile kernels' store pattern.  Best of plain / non-temporal loads; the rate counts the bytes READ, like
 * roofline.achieved does.  Measurement aid for bench.py ("stream_rw_probe"). */
int bowgpu_stream_rw_probe(const void *dev_a, const void *dev_b, int64_t bytes_each, void *out_a, void *out_b,
                             int64_t rows_per_slot, double *read_gb_per_s, double *ms /* nullable */);
/* Per-thread route mask.  Which kernel serves a call follows from the call's shape alone; two bits are a caller's business:
 *   BOWGPU_ROUTE_STRICT_ORDER  bowgpu_options.strict_order for every call of the thread
 *   BOWGPU_ROUTE_PINNED_STAGE  BOWGPU_HOST_PINNED inputs staged through HBM instead of read in place over the host link
 * The remaining bits (bow_amd/csrc/debug_routes.h, not part of the ABI) let the parity tests push one call through every kernel
 * that can take it; all routes give the same results where several apply.  The library never reads the environment on the call
 * path; BOWGPU_ROUTE=<mask> is read once per process as every thread's initial mask (profiling an unmodified script). */
enum {
    BOWGPU_ROUTE_PINNED_STAGE = 1024,
    BOWGPU_ROUTE_STRICT_ORDER = 2048
};
int bowgpu_debug_set_route(uint32_t mask);
int bowgpu_debug_get_route(uint32_t *mask);

/* Diagnostic builds only (kernels compiled with in-kernel stamps): words [first, first + n) of the calling thread's device
 * status block; zero_after clears them.  The product build never writes those words. */
int bowgpu_debug_status(int32_t first, int32_t n, uint32_t *out, int32_t zero_after);
/* The CPU side of the staging of pageable buffers on its own (no device involved): a memcpy split over the library's helper threads
 * (BOWGPU_COPY_THREADS).  For the sanitizer run of the host code and for measuring the host's copy rate. */
int bowgpu_debug_host_copy(void *dst, const void *src, int64_t bytes);
/* order-independent 64-bit checksum of a device buffer of n 8-byte words (xor / sum of mix) */
int bowgpu_checksum64(const void *dev, int64_t n_words, uint64_t *xor_out, uint64_t *sum_out);
/* ... of words that are words [index_base, index_base + n_words) of a larger array: the checksums of the pieces of an array
 * combine (xor with xor, sum with sum, mod 2^64) into the checksum of the whole - sharded outputs against the unsharded ones
 * without bringing either to the host */
int bowgpu_checksum64_at(const void *dev, int64_t n_words, int64_t index_base, uint64_t *xor_out, uint64_t *sum_out);

#ifdef __cplusplus
}
#endif
#endif
