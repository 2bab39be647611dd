/*
 * bow_oracle.h — CPU ORACLE for the rolling-window aggregation path of Metronlab/bow.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (bow_amd/, include/)
 * never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference's Go algorithm (the reference cannot be
 * compiled here: no Go toolchain, un-vendored arrow/go/v8).  Every function cites the
 * reference file:line it follows (paths relative to the reference root).  Parity is
 * PINNED: tests/test_oracle_golden.py replays every golden vector the reference's own
 * tests hold for this path (tests/golden/reference_vectors.json, transcribed from
 * rolling/{,aggregation/,interpolation/}<name>_test.go and
 * bowfill_test.go).  What no reference test covers is listed in DESIGN.md as
 * "parity by code-reading only".
 */
#ifndef BOW_ORACLE_H
#define BOW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bowtypes.go:17-32 */
enum {
    ORC_UNKNOWN = 0,
    ORC_FLOAT64 = 1,
    ORC_INT64 = 2,
    ORC_BOOLEAN = 3, /* Arrow bit-packed values */
    ORC_STRING = 4,  /* not supported by the oracle */
    ORC_INPUT_DEPENDENT = 5,
    ORC_ITERATOR_DEPENDENT = 6
};

/* One Arrow array as bow sees it (bowgetters.go:46-63):
 * values = Buffers()[1], validity = Buffers()[0] (NULL => all valid), both indexed
 * from `offset` (array slices share buffers: bow.go:279-283). */
typedef struct {
    const void *values;
    const uint8_t *validity;
    int64_t offset;
    int64_t length;
    int32_t type;
    int32_t _pad;
} orc_col_t;

/* Output column: caller allocates `length` slots (8 B each, or ceil(n/8) B for BOOLEAN)
 * and ceil(length/8) validity bytes.  The oracle zero-fills both first
 * (bow.NewBuffer, bowbuffer.go:22-40) then sets value + bit (SetOrDrop, :60-80). */
typedef struct {
    void *values;
    uint8_t *validity;
    int64_t length;
    int32_t type; /* filled in by the oracle: resolved return type (aggregation.go:110-121) */
    int32_t _pad;
} orc_out_t;

/* Aggregator kinds — one per constructor in rolling/aggregation/ */
enum {
    ORC_AGG_WINDOW_START = 0,    /* windowstart.go:8-13 */
    ORC_AGG_SUM = 1,             /* sum.go:8-25 */
    ORC_AGG_MEAN = 2,            /* arithmeticmean.go:8-30 */
    ORC_AGG_MIN = 3,             /* minmax.go:8-31 */
    ORC_AGG_MAX = 4,             /* minmax.go:33-56 */
    ORC_AGG_COUNT = 5,           /* count.go:8-20 */
    ORC_AGG_FIRST = 6,           /* firstlast.go:8-21 */
    ORC_AGG_LAST = 7,            /* firstlast.go:23-36 */
    ORC_AGG_INTEGRAL_STEP = 8,   /* integral.go:40-69 */
    ORC_AGG_INTEGRAL_TRAPEZOID = 9, /* integral.go:8-38 (needs inclusive window) */
    ORC_AGG_WAVG_STEP = 10,      /* weightedmean.go:8-20 */
    ORC_AGG_WAVG_LINEAR = 11,    /* weightedmean.go:22-34 (needs inclusive window) */
    ORC_AGG_NUM_ROWS = 12,       /* test-only closure float64(w.Bow.NumRows()): aggregation_test.go:28-31 */
    ORC_AGG_MODE = 13            /* mode.go:8-32 */
};

typedef struct {
    int32_t kind;
    int32_t col;          /* input column index (aggregation.go:176-181) */
    int32_t n_factors;    /* transformation.Factor chain (factor.go:7-20), applied in order */
    int32_t _pad;
    const double *factors;
} orc_agg_t;

/* Interpolator kinds — rolling/interpolation/ */
enum {
    ORC_INTERP_WINDOW_START = 0,  /* interpolation/windowstart.go:8-14 */
    ORC_INTERP_LINEAR = 1,        /* interpolation/linear.go:8-38 */
    ORC_INTERP_STEP_PREVIOUS = 2, /* interpolation/stepprevious.go:8-26 */
    ORC_INTERP_NONE = 3,          /* interpolation/none.go:7-13 */
    ORC_INTERP_CONST = 4          /* test-only closure returning a constant: interpolation_test.go:16-19 */
};

typedef struct {
    int32_t kind;
    int32_t col;
    double const_value;   /* ORC_INTERP_CONST */
    /* Options.PrevRow (rolling.go:49-53) as seen by this interpolator:
     * prevRow.GetFloat64(intervalCol), prevRow.GetFloat64(col) (linear.go:14-18) /
     * prevRow.GetValue(col) (stepprevious.go:13-15). has_prev_row=0 => PrevRow nil. */
    int32_t has_prev_row;
    int32_t prev_t_valid;
    int32_t prev_v_valid;
    int32_t _pad;
    double prev_t;
    double prev_v;        /* for STEP_PREVIOUS on int64 columns the raw value is prev_v_i64 */
    int64_t prev_v_i64;
} orc_interp_t;

/* error codes */
enum {
    ORC_OK = 0,
    ORC_ERR_INTERVAL = -1,       /* "strictly positive interval required"  rolling.go:115-117 */
    ORC_ERR_TS_TYPE = -2,        /* "impossible to create a new intervalRolling on column of type %v" :70-73 */
    ORC_ERR_FIRST_TS_NULL = -3,  /* "the first value of the column should be convertible to int64" :89-93 */
    ORC_ERR_NO_AGG = -4,         /* "at least one column aggregation is required" aggregation.go:148-150 */
    ORC_ERR_KEEP_INTERVAL = -5,  /* "must keep interval column '%s'" :163-166 */
    ORC_ERR_BAD_COL = -6,        /* "no column '%s'" bowgetters.go:323 */
    ORC_ERR_TYPE = -7,           /* validateInterpolation type whitelist interpolation.go:82-93 */
    ORC_ERR_NOT_SORTED = -8,     /* FillLinear: "refColIndex '%d' is empty or not sorted" bowfill.go:39-42 */
    ORC_ERR_UNSUPPORTED = -9,
    ORC_ERR_ARG = -10
};

/* rolling.go:114-128 */
int orc_enforce_interval_and_offset(int64_t interval, int64_t offset, int64_t *offset_out);

/* newIntervalRolling: rolling.go:69-112 (+ countWindows :143-154).
 * offset is the RAW Options.Offset; s0 = first window start, W = numWindows. */
int orc_plan_windows(const orc_col_t *ts, int64_t interval, int64_t offset, int64_t *s0, int64_t *W);

/* Literal window iterator (rolling.go:162-239).  Fills one record per produced window;
 * returns the number of windows produced (<= W) or a negative error.
 * Arrays may be NULL. cap = capacity of the arrays. */
int64_t orc_iterate_windows(const orc_col_t *ts, int64_t interval, int64_t offset, int inclusive,
                            int64_t cap, int64_t *first_index, int64_t *slice_begin,
                            int64_t *slice_end, int64_t *first_value, int64_t *last_value,
                            uint8_t *is_inclusive);

/* Rolling.Aggregate: aggregation.go:123-238.  `inclusive` = Options.Inclusive of the
 * receiver; the effective flag is inclusive || any aggregator needs it (A.6).
 * outs[i] must have room for W slots.  On return *new_interval_col = index of the last
 * aggregator reading ts_col (aggregation.go:152-161). */
int orc_aggregate(const orc_col_t *cols, int ncols, int ts_col, int64_t interval, int64_t offset,
                  int inclusive, const orc_agg_t *aggs, int naggs, orc_out_t *outs,
                  int *new_interval_col);

/* Rolling.Interpolate: interpolation.go:30-161.  Two-phase: call with outs == NULL to get
 * *n_out (number of output rows), then with outs sized for it.  interps must list the
 * bow's columns in order (AppendBows needs equal schemas: bowappend.go:11-13). */
int orc_interpolate(const orc_col_t *cols, int ncols, int ts_col, int64_t interval, int64_t offset,
                    int inclusive, const orc_interp_t *interps, int ninterps, orc_out_t *outs,
                    int64_t *n_out);

/* Bow.FillLinear: bowfill.go:14-103.  out gets a copy of cols[fill_col] with nulls filled.
 * *unchanged = 1 when the reference returns the same Bow (bowfill.go:35-37, :53-55)
 * (out is still filled with a copy). */
int orc_fill_linear(const orc_col_t *cols, int ncols, int ref_col, int fill_col, orc_out_t *out,
                    int *unchanged);

/* Bow.FillPrevious / FillNext (bowfill.go:162-253) and Bow.FillMean (bowfill.go:105-160) of ONE Int64 / Float64
 * column: out gets the filled copy; *unchanged = 1 when the column has no nulls (passed through). */
enum { ORC_FILL_PREVIOUS = 0, ORC_FILL_NEXT = 1, ORC_FILL_MEAN = 2 };
int orc_fill(const orc_col_t *col, int method, orc_out_t *out, int *unchanged);

/* bowassertion.go:15-86 */
int orc_is_col_sorted(const orc_col_t *col);
int orc_is_col_empty(const orc_col_t *col);

/* whole-frame aggregation: rolling/aggregation/whole.go:12-93. outs[i] has 1 slot. */
int orc_aggregate_whole(const orc_col_t *cols, int ncols, int ts_col, const orc_agg_t *aggs,
                        int naggs, orc_out_t *outs);

/* ------------------------------------------------------------------------------------
 * Synthetic-input generators.  NOT reference semantics: these restate the
 * DEVICE generators of the product (bow_amd/csrc/generate.hip) so tests can check the
 * device-generated inputs and feed identical inputs to both sides.
 * ---------------------------------------------------------------------------------- */
uint64_t orc_mix64(uint64_t seed, uint64_t i);
/* cfg-dense: ts[i]=row0+i, val=u01(mix64(seed,row0+i)), all valid */
void orc_gen_dense(int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val);
/* cfg-sparse: ts=10*i+U{0..9}, val=U{0..9}+0.5, valid with p=0.7 (bit (row0+i) of an
 * absolute bitmap => validity is written at bit offset (row0+i) - 8*floor(row0/8); row0
 * must be a multiple of 8). */
void orc_gen_sparse(int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val,
                    uint8_t *validity);

#ifdef __cplusplus
}
#endif
#endif
