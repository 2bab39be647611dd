/*
 * bow_oracle.c — CPU ORACLE (test infrastructure only; see bow_oracle.h).
 *
 * Plain-C restatement of the reference's rolling-window path.  The control flow follows
 * the Go sources statement by statement; citations are reference file:line.
 * Build with -O2 -fwrapv -ffp-contract=off (Go wraps on int64 overflow and gc/amd64
 * never fuses a*b+c).
 *
 * Parity: pinned by tests/test_oracle_golden.py against the reference's own test vectors.
 */
#include "bow_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- Go numeric semantics */

/* Go float64->int64 conversion on amd64 is CVTTSD2SI: NaN / out of range => 0x8000...0. */
static inline int64_t go_f64_to_i64(double x) {
    if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)x;
}

/* ---------------------------------------------------------------- element access */

static inline int bit_is_set(const uint8_t *bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t *bm, int64_t i) { bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline void bit_clear(uint8_t *bm, int64_t i) { bm[i >> 3] &= (uint8_t)~(1u << (i & 7)); }

/* A view [begin, begin+len) over the full bow's columns == bow.NewSlice (bow.go:279-283). */
typedef struct {
    const orc_col_t *cols;
    int64_t begin;
    int64_t len;
} view_t;

static inline int col_is_valid(const orc_col_t *c, int64_t row) {
    if (!c->validity) return 1;
    return bit_is_set(c->validity, c->offset + row);
}

/* Bow.GetInt64 on an Int64 column: bowgetters.go:155-163 */
static inline int view_get_i64(const view_t *v, int col, int64_t i, int64_t *out) {
    if (i < 0 || i >= v->len) { *out = 0; return 0; }
    const orc_col_t *c = &v->cols[col];
    int64_t row = v->begin + i;
    switch (c->type) {
    case ORC_INT64: *out = ((const int64_t *)c->values)[c->offset + row]; break;
    case ORC_FLOAT64: *out = go_f64_to_i64(((const double *)c->values)[c->offset + row]); break; /* :164-166 */
    case ORC_BOOLEAN: *out = bit_is_set((const uint8_t *)c->values, c->offset + row); break;   /* :167-173 */
    default: *out = 0; return 0;
    }
    return col_is_valid(c, row);
}

/* Bow.GetFloat64: bowgetters.go:218-247 */
static inline int view_get_f64(const view_t *v, int col, int64_t i, double *out) {
    if (i < 0 || i >= v->len) { *out = 0.; return 0; }
    const orc_col_t *c = &v->cols[col];
    int64_t row = v->begin + i;
    switch (c->type) {
    case ORC_FLOAT64: *out = ((const double *)c->values)[c->offset + row]; break;
    case ORC_INT64: *out = (double)((const int64_t *)c->values)[c->offset + row]; break;        /* :227-229 */
    case ORC_BOOLEAN: *out = bit_is_set((const uint8_t *)c->values, c->offset + row) ? 1. : 0.; break;
    default: *out = 0.; return 0;
    }
    return col_is_valid(c, row);
}

/* GetPrevFloat64 / GetNextFloat64: bowgetters.go:252-277 */
static double view_prev_f64(const view_t *v, int col, int64_t i, int64_t *idx) {
    while (i >= 0 && i < v->len) {
        double x;
        if (view_get_f64(v, col, i, &x)) { *idx = i; return x; }
        i--;
    }
    *idx = -1;
    return 0.;
}
static double view_next_f64(const view_t *v, int col, int64_t i, int64_t *idx) {
    while (i >= 0 && i < v->len) {
        double x;
        if (view_get_f64(v, col, i, &x)) { *idx = i; return x; }
        i++;
    }
    *idx = -1;
    return 0.;
}

/* GetPrevFloat64s: bowgetters.go:282-294 */
static int64_t view_prev_f64s(const view_t *v, int c1, int c2, int64_t i, double *v1, double *v2) {
    while (i >= 0 && i < v->len) {
        int64_t i2;
        *v1 = view_prev_f64(v, c1, i, &i);
        *v2 = view_prev_f64(v, c2, i, &i2);
        if (i == i2) return i;
        i--;
    }
    *v1 = 0.; *v2 = 0.;
    return -1;
}
/* GetNextFloat64s: bowgetters.go:299-311 */
static int64_t view_next_f64s(const view_t *v, int c1, int c2, int64_t i, double *v1, double *v2) {
    while (i >= 0 && i < v->len) {
        int64_t i2;
        *v1 = view_next_f64(v, c1, i, &i);
        *v2 = view_next_f64(v, c2, i, &i2);
        if (i == i2) return i;
        i++;
    }
    *v1 = 0.; *v2 = 0.;
    return -1;
}

/* GetPrevInt64: bowgetters.go:189-199 */
static int64_t view_prev_i64(const view_t *v, int col, int64_t i, int64_t *idx) {
    while (i >= 0 && i < v->len) {
        int64_t x;
        if (view_get_i64(v, col, i, &x)) { *idx = i; return x; }
        i--;
    }
    *idx = -1;
    return 0;
}

/* GetNextRowIndex/GetPrevRowIndex via validity only: bowgetters.go:127-151 (== GetNext/PrevValue row) */
static int64_t view_next_valid(const view_t *v, int col, int64_t i) {
    while (i >= 0 && i < v->len) {
        if (col_is_valid(&v->cols[col], v->begin + i)) return i;
        i++;
    }
    return -1;
}
static int64_t view_prev_valid(const view_t *v, int col, int64_t i) {
    while (i >= 0 && i < v->len) {
        if (col_is_valid(&v->cols[col], v->begin + i)) return i;
        i--;
    }
    return -1;
}

/* ---------------------------------------------------------------- rolling.go */

int orc_enforce_interval_and_offset(int64_t interval, int64_t offset, int64_t *offset_out) {
    /* rolling.go:114-128 */
    if (interval <= 0) return ORC_ERR_INTERVAL;
    if (offset >= interval || offset <= -interval) offset = offset % interval;
    if (offset < 0) offset += interval;
    *offset_out = offset;
    return ORC_OK;
}

int orc_plan_windows(const orc_col_t *ts, int64_t interval, int64_t offset, int64_t *s0, int64_t *W) {
    /* newIntervalRolling: rolling.go:69-112 */
    if (ts->type != ORC_INT64) return ORC_ERR_TS_TYPE; /* :70-73 */
    int rc = orc_enforce_interval_and_offset(interval, offset, &offset);
    if (rc) return rc;
    view_t full = {ts, 0, ts->length};
    int64_t first = 0;
    if (ts->length > 0) { /* :87-100 */
        int64_t v;
        if (!view_get_i64(&full, 0, 0, &v)) return ORC_ERR_FIRST_TS_NULL;
        first = (v / interval) * interval + offset;
        if (first > v) first -= interval;
    }
    /* countWindows: :143-154 */
    int64_t n = 0;
    if (ts->length > 0) {
        int64_t idx;
        int64_t last = view_prev_i64(&full, 0, ts->length - 1, &idx);
        if (idx == -1 || first > last) n = 0;
        else n = (last - first) / interval + 1;
    }
    *s0 = first;
    *W = n;
    return ORC_OK;
}

/* iterator state: intervalRolling (rolling.go:31-43) */
typedef struct {
    const orc_col_t *ts; /* interval column only */
    int64_t interval;
    int inclusive; /* options.Inclusive */
    int64_t curr_first_value;
    int64_t curr_row;
    int64_t curr_win;
} iter_t;

typedef struct {
    int64_t begin, end;   /* Window.Bow = full[begin:end) ; empty => 0,0 (NewEmptySlice) */
    int64_t first_index;  /* Window.FirstIndex */
    int64_t first_value, last_value;
    int is_inclusive;
} win_t;

static int iter_has_next(const iter_t *r) {
    /* rolling.go:162-173 */
    if (r->curr_row >= r->ts->length) return 0;
    view_t full = {r->ts, 0, r->ts->length};
    int64_t last;
    if (!view_get_i64(&full, 0, r->ts->length - 1, &last)) return 0;
    return r->curr_first_value <= last;
}

static int iter_next(iter_t *r, win_t *w, int64_t *win_index) {
    /* rolling.go:177-239 */
    if (!iter_has_next(r)) { *win_index = r->curr_win; return 0; }
    view_t full = {r->ts, 0, r->ts->length};
    int64_t first_value = r->curr_first_value;
    int64_t last_value = r->curr_first_value + r->interval;
    int64_t row = 0;
    int is_inclusive = 0;
    int64_t first_row = r->curr_row;
    int64_t last_row = -1;
    for (row = first_row; row < r->ts->length; row++) {
        int64_t val;
        if (!view_get_i64(&full, 0, row, &val)) continue; /* :191-193 */
        if (val < first_value) continue;                  /* :194-196 */
        if (val > last_value) break;                      /* :197-199 */
        if (val == last_value) {                          /* :201-209 */
            if (is_inclusive) break;
            if (!r->inclusive) break;
            is_inclusive = 1;
        }
        last_row = row;
    }
    if (!is_inclusive) r->curr_row = row; else r->curr_row = row - 1; /* :214-218 */
    r->curr_first_value = last_value;
    *win_index = r->curr_win;
    r->curr_win++;
    if (last_row == -1) { w->begin = 0; w->end = 0; }     /* NewEmptySlice :225-226 */
    else { w->begin = first_row; w->end = last_row + 1; } /* :228 */
    w->first_index = first_row;
    w->first_value = first_value;
    w->last_value = last_value;
    w->is_inclusive = is_inclusive;
    return 1;
}

/* Window.UnsetInclusive: window.go:23-31 */
static win_t win_unset_inclusive(win_t w) {
    if (!w.is_inclusive) return w;
    w.is_inclusive = 0;
    w.end -= 1; /* NewSlice(0, NumRows()-1) */
    return w;
}

int64_t orc_iterate_windows(const orc_col_t *ts, int64_t interval, int64_t offset, int inclusive,
                            int64_t cap, int64_t *first_index, int64_t *slice_begin,
                            int64_t *slice_end, int64_t *first_value, int64_t *last_value,
                            uint8_t *is_inclusive) {
    int64_t s0, W;
    int rc = orc_plan_windows(ts, interval, offset, &s0, &W);
    if (rc) return rc;
    iter_t it = {ts, interval, inclusive, s0, 0, 0};
    int64_t n = 0;
    while (iter_has_next(&it)) {
        win_t w;
        int64_t wi;
        iter_next(&it, &w, &wi);
        if (wi < cap) {
            if (first_index) first_index[wi] = w.first_index;
            if (slice_begin) slice_begin[wi] = w.begin;
            if (slice_end) slice_end[wi] = w.end;
            if (first_value) first_value[wi] = w.first_value;
            if (last_value) last_value[wi] = w.last_value;
            if (is_inclusive) is_inclusive[wi] = (uint8_t)w.is_inclusive;
        }
        n++;
    }
    return n;
}

/* ---------------------------------------------------------------- reducers */

/* A reducer result: nil | float64 | int64 (interface{} in Go) */
enum { VAL_NIL = 0, VAL_F64 = 1, VAL_I64 = 2, VAL_BOOL = 3 };
typedef struct {
    int tag;
    double f;
    int64_t i;
} val_t;

static val_t val_nil(void) { val_t v = {VAL_NIL, 0., 0}; return v; }
static val_t val_f64(double x) { val_t v = {VAL_F64, x, 0}; return v; }
static val_t val_i64(int64_t x) { val_t v = {VAL_I64, 0., x}; return v; }
static val_t val_bool(int b) { val_t v = {VAL_BOOL, 0., b ? 1 : 0}; return v; }

static int agg_needs_inclusive(int kind) {
    /* NewColAggregation(col, true, ...) only in integral.go:9 and weightedmean.go:24 */
    return kind == ORC_AGG_INTEGRAL_TRAPEZOID || kind == ORC_AGG_WAVG_LINEAR;
}

static int agg_type(int kind) {
    switch (kind) {
    case ORC_AGG_WINDOW_START: return ORC_ITERATOR_DEPENDENT; /* windowstart.go:9 */
    case ORC_AGG_COUNT: return ORC_INT64;                     /* count.go:9 */
    case ORC_AGG_FIRST:
    case ORC_AGG_LAST: return ORC_INPUT_DEPENDENT;            /* firstlast.go:9,24 */
    case ORC_AGG_MODE: return ORC_INPUT_DEPENDENT;            /* mode.go:9 */
    default: return ORC_FLOAT64;
    }
}

/* GetValue (bowgetters.go:46-63) as a val_t */
static val_t view_get_value(const view_t *v, int col, int64_t i) {
    const orc_col_t *c = &v->cols[col];
    int64_t row = v->begin + i;
    if (!col_is_valid(c, row)) return val_nil();
    switch (c->type) {
    case ORC_FLOAT64: return val_f64(((const double *)c->values)[c->offset + row]);
    case ORC_INT64: return val_i64(((const int64_t *)c->values)[c->offset + row]);
    case ORC_BOOLEAN: return val_bool(bit_is_set((const uint8_t *)c->values, c->offset + row));
    default: return val_nil();
    }
}

static val_t integral_trapezoid(const view_t *wv, int ts_col, int col) {
    /* integral.go:8-38 */
    if (wv->len == 0) return val_nil();
    double sum = 0.;
    int ok = 0;
    double t0, v0;
    int64_t row = view_next_f64s(wv, ts_col, col, 0, &t0, &v0);
    if (row < 0) return val_nil();
    while (row >= 0) {
        double t1, v1;
        int64_t next = view_next_f64s(wv, ts_col, col, row + 1, &t1, &v1);
        if (next < 0) break;
        sum += (v0 + v1) / 2 * (t1 - t0);
        ok = 1;
        t0 = t1; v0 = v1; row = next;
    }
    if (!ok) return val_nil();
    return val_f64(sum);
}

static val_t integral_step(const view_t *wv, int ts_col, int col, int64_t last_value) {
    /* integral.go:40-69 */
    if (wv->len == 0) return val_nil();
    double sum = 0.;
    int ok = 0;
    double t0, v0;
    int64_t row = view_next_f64s(wv, ts_col, col, 0, &t0, &v0);
    while (row >= 0) {
        double t1, v1;
        int64_t next = view_next_f64s(wv, ts_col, col, row + 1, &t1, &v1);
        if (next < 0) t1 = (double)last_value;
        sum += v0 * (t1 - t0);
        ok = 1;
        if (next < 0) break;
        t0 = t1; v0 = v1; row = next;
    }
    if (!ok) return val_nil();
    return val_f64(sum);
}

static val_t apply_agg(int kind, const orc_col_t *cols, int ts_col, int col, const win_t *w) {
    view_t wv = {cols, w->begin, w->end - w->begin};
    int64_t n = wv.len; /* w.Bow.NumRows() */
    switch (kind) {
    case ORC_AGG_WINDOW_START: /* windowstart.go:8-13 */
        return val_i64(w->first_value);
    case ORC_AGG_NUM_ROWS: /* aggregation_test.go:28-31 */
        return val_f64((double)n);
    case ORC_AGG_SUM: { /* sum.go:8-25 */
        if (n == 0) return val_f64(0.);
        double sum = 0.;
        for (int64_t i = 0; i < n; i++) {
            double x;
            if (!view_get_f64(&wv, col, i, &x)) continue;
            sum += x;
        }
        return val_f64(sum);
    }
    case ORC_AGG_MEAN: { /* arithmeticmean.go:8-30 */
        if (n == 0) return val_nil();
        double sum = 0.;
        int64_t count = 0;
        for (int64_t i = 0; i < n; i++) {
            double x;
            if (!view_get_f64(&wv, col, i, &x)) continue;
            sum += x;
            count++;
        }
        if (count == 0) return val_nil();
        return val_f64(sum / (double)count);
    }
    case ORC_AGG_MIN:   /* minmax.go:8-31 */
    case ORC_AGG_MAX: { /* minmax.go:33-56 */
        if (n == 0) return val_nil();
        int have = 0;
        double m = 0.;
        for (int64_t i = 0; i < n; i++) {
            double x;
            if (!view_get_f64(&wv, col, i, &x)) continue;
            if (have) {
                if (kind == ORC_AGG_MIN ? (x < m) : (x > m)) m = x;
                continue;
            }
            m = x;
            have = 1;
        }
        return have ? val_f64(m) : val_nil();
    }
    case ORC_AGG_COUNT: { /* count.go:8-20 */
        int64_t count = 0;
        for (int64_t i = 0; i < n; i++)
            if (col_is_valid(&cols[col], wv.begin + i)) count++;
        return val_i64(count);
    }
    case ORC_AGG_FIRST: { /* firstlast.go:8-21 */
        if (n == 0) return val_nil();
        int64_t r = view_next_valid(&wv, col, 0);
        if (r == -1) return val_nil();
        return view_get_value(&wv, col, r);
    }
    case ORC_AGG_LAST: { /* firstlast.go:23-36 */
        if (n == 0) return val_nil();
        int64_t r = view_prev_valid(&wv, col, n - 1);
        if (r == -1) return val_nil();
        return view_get_value(&wv, col, r);
    }
    case ORC_AGG_MODE: { /* mode.go:8-32: occurrences[v]++ per non-nil value in row order; the result is the value whose count first
                          * exceeds every earlier count.  Map keys are interface{} values: == of the dynamic type, so for float64
                          * NaN never equals anything (each NaN counts once) and -0 == +0; the value kept is the row's own. */
        if (n == 0) return val_nil();
        int64_t max = 0;
        val_t res = val_nil();
        for (int64_t i = 0; i < n; i++) {
            val_t v = view_get_value(&wv, col, i);
            if (v.tag == VAL_NIL) continue;
            int64_t nb = 0;
            for (int64_t j = 0; j <= i; j++) { /* occurrences[v] after this row's increment */
                val_t u = view_get_value(&wv, col, j);
                if (u.tag == VAL_NIL) continue;
                if (u.tag == VAL_F64 ? (u.f == v.f) : (u.i == v.i)) nb++;
            }
            if (v.tag == VAL_F64 && v.f != v.f) nb = 1; /* a NaN key never matches an earlier one (nor itself above) */
            if (nb > max) { max = nb; res = v; }
        }
        return res;
    }
    case ORC_AGG_INTEGRAL_TRAPEZOID: return integral_trapezoid(&wv, ts_col, col);
    case ORC_AGG_INTEGRAL_STEP: return integral_step(&wv, ts_col, col, w->last_value);
    case ORC_AGG_WAVG_STEP: { /* weightedmean.go:8-20 */
        val_t v = integral_step(&wv, ts_col, col, w->last_value);
        if (v.tag == VAL_NIL) return v;
        double wide = (double)(w->last_value - w->first_value);
        return val_f64(v.f / wide);
    }
    case ORC_AGG_WAVG_LINEAR: { /* weightedmean.go:22-34 */
        val_t v = integral_trapezoid(&wv, ts_col, col);
        if (v.tag == VAL_NIL) return v;
        double wide = (double)(w->last_value - w->first_value);
        return val_f64(v.f / wide);
    }
    default: return val_nil();
    }
}

/* transformation.Factor: factor.go:7-20 */
static val_t apply_factor(val_t x, double n) {
    switch (x.tag) {
    case VAL_F64: return val_f64(x.f * n);
    case VAL_I64: return val_i64(go_f64_to_i64((double)x.i * n));
    default: return x; /* nil stays nil; bool => error in Go, unreachable here */
    }
}

/* Buffer.SetOrDrop: bowbuffer.go:60-80 with Type.Convert (bowtypes.go:64-81, bowconvert.go) */
static void out_set_or_drop(orc_out_t *o, int64_t i, val_t v) {
    int valid = 0;
    switch (o->type) {
    case ORC_INT64: {
        int64_t x = 0;
        if (v.tag == VAL_I64) { x = v.i; valid = 1; }
        else if (v.tag == VAL_F64) { x = go_f64_to_i64(v.f); valid = 1; } /* bowconvert.go:28-29 */
        else if (v.tag == VAL_BOOL) { x = v.i; valid = 1; }
        ((int64_t *)o->values)[i] = x;
        break;
    }
    case ORC_FLOAT64: {
        double x = 0.;
        if (v.tag == VAL_F64) { x = v.f; valid = 1; }
        else if (v.tag == VAL_I64) { x = (double)v.i; valid = 1; } /* bowconvert.go:59-60 */
        else if (v.tag == VAL_BOOL) { x = v.i ? 1. : 0.; valid = 1; }
        ((double *)o->values)[i] = x;
        break;
    }
    case ORC_BOOLEAN: {
        int x = 0;
        if (v.tag == VAL_BOOL) { x = (int)v.i; valid = 1; }
        else if (v.tag == VAL_I64) { x = v.i != 0; valid = 1; }
        else if (v.tag == VAL_F64) { x = v.f != 0.; valid = 1; }
        if (x) bit_set((uint8_t *)o->values, i); else bit_clear((uint8_t *)o->values, i);
        break;
    }
    default: break;
    }
    if (valid) bit_set(o->validity, i); else bit_clear(o->validity, i);
}

/* Buffer.SetOrDropStrict: bowbuffer.go:84-104 (type assertion, no conversion) */
static void out_set_or_drop_strict(orc_out_t *o, int64_t i, val_t v) {
    int want = o->type == ORC_INT64 ? VAL_I64 : o->type == ORC_FLOAT64 ? VAL_F64 : VAL_BOOL;
    if (v.tag != want) v = val_nil();
    if (v.tag == VAL_NIL) {
        /* Go: `b.Data[i], valid = value.(T)` stores the zero value on a failed assertion */
        if (o->type == ORC_INT64) ((int64_t *)o->values)[i] = 0;
        else if (o->type == ORC_FLOAT64) ((double *)o->values)[i] = 0.;
        else bit_clear((uint8_t *)o->values, i);
        bit_clear(o->validity, i);
        return;
    }
    out_set_or_drop(o, i, v);
}

static size_t out_value_bytes(int type, int64_t n) {
    return type == ORC_BOOLEAN ? (size_t)((n + 7) / 8) : (size_t)n * 8;
}

/* bow.NewBuffer: bowbuffer.go:22-40 — zero data, all-null bitmap */
static void out_init(orc_out_t *o, int64_t n, int type) {
    o->type = type;
    o->length = n;
    if (n > 0) {
        memset(o->values, 0, out_value_bytes(type, n));
        memset(o->validity, 0, (size_t)((n + 7) / 8));
    }
}

/* colAggregation.GetReturnType: aggregation.go:110-121 */
static int resolve_type(int t, int input_type, int iter_type) {
    if (t == ORC_INPUT_DEPENDENT) return input_type;
    if (t == ORC_ITERATOR_DEPENDENT) return iter_type;
    return t;
}

int orc_aggregate(const orc_col_t *cols, int ncols, int ts_col, int64_t interval, int64_t offset,
                  int inclusive, const orc_agg_t *aggs, int naggs, orc_out_t *outs,
                  int *new_interval_col) {
    if (ts_col < 0 || ts_col >= ncols) return ORC_ERR_BAD_COL;
    int64_t s0, W;
    int rc = orc_plan_windows(&cols[ts_col], interval, offset, &s0, &W);
    if (rc) return rc;

    /* indexedAggregations + validateAggregation: aggregation.go:147-188 */
    if (naggs == 0) return ORC_ERR_NO_AGG;
    int nic = -1;
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].col < 0 || aggs[i].col >= ncols) return ORC_ERR_BAD_COL;
        if (agg_needs_inclusive(aggs[i].kind)) inclusive = 1; /* :183-185 */
        if (aggs[i].col == ts_col) nic = i;                    /* :158-160, last one wins */
    }
    if (nic == -1) return ORC_ERR_KEEP_INTERVAL;
    if (new_interval_col) *new_interval_col = nic;

    /* aggregateWindows: aggregation.go:190-238 */
    for (int a = 0; a < naggs; a++) {
        const orc_agg_t *ag = &aggs[a];
        int typ = resolve_type(agg_type(ag->kind), cols[ag->col].type, cols[ts_col].type);
        out_init(&outs[a], W, typ); /* :198 */
        iter_t it = {&cols[ts_col], interval, inclusive, s0, 0, 0}; /* rCopy := *r  :194 */
        while (iter_has_next(&it)) {
            win_t w;
            int64_t wi;
            iter_next(&it, &w, &wi);
            if (!agg_needs_inclusive(ag->kind) && w.is_inclusive) w = win_unset_inclusive(w); /* :207-208 */
            val_t v = apply_agg(ag->kind, cols, ts_col, ag->col, &w);
            for (int t = 0; t < ag->n_factors; t++) v = apply_factor(v, ag->factors[t]); /* :216-221 */
            if (v.tag == VAL_NIL) continue; /* :223-225 */
            if (wi >= W) continue;          /* Go would panic (index out of range); unreachable for sorted ts */
            out_set_or_drop(&outs[a], wi, v); /* :227 */
        }
    }
    return ORC_OK;
}

int orc_aggregate_whole(const orc_col_t *cols, int ncols, int ts_col, const orc_agg_t *aggs,
                        int naggs, orc_out_t *outs) {
    /* whole.go:12-93 */
    if (naggs == 0) return ORC_ERR_NO_AGG;
    if (ts_col < 0 || ts_col >= ncols) return ORC_ERR_BAD_COL;
    int64_t n = cols[ts_col].length;
    view_t full = {cols, 0, n};
    for (int a = 0; a < naggs; a++) {
        const orc_agg_t *ag = &aggs[a];
        if (ag->col < 0 || ag->col >= ncols) return ORC_ERR_BAD_COL;
        int typ = resolve_type(agg_type(ag->kind), cols[ag->col].type, cols[ag->col].type); /* :44-46 */
        if (n == 0) { out_init(&outs[a], 0, typ); continue; }
        out_init(&outs[a], 1, typ);
        int64_t idx;
        double first = view_next_f64(&full, ts_col, 0, &idx); /* :54-57 */
        if (idx == -1) first = -1;
        double last = view_prev_f64(&full, ts_col, n - 1, &idx); /* :59-62 */
        if (idx == -1) last = -1;
        win_t w = {0, n, 0, go_f64_to_i64(first), go_f64_to_i64(last), 1};
        val_t v = apply_agg(ag->kind, cols, ts_col, ag->col, &w);
        for (int t = 0; t < ag->n_factors; t++) v = apply_factor(v, ag->factors[t]);
        out_set_or_drop_strict(&outs[a], 0, v); /* :86 */
    }
    return ORC_OK;
}

/* ---------------------------------------------------------------- interpolation.go */

typedef struct {
    /* closure state of interpolation.Linear (linear.go:9-10) / StepPrevious (stepprevious.go:9) */
    double prev_t0, prev_v0;
    int prev_valid;
    val_t prev_val;
} interp_state_t;

static int interp_type_ok(int kind, int col_type) {
    switch (kind) {
    case ORC_INTERP_WINDOW_START: return col_type == ORC_INT64;                             /* windowstart.go:9 */
    case ORC_INTERP_LINEAR: return col_type == ORC_INT64 || col_type == ORC_FLOAT64;        /* linear.go:11 */
    case ORC_INTERP_STEP_PREVIOUS: return col_type == ORC_INT64 || col_type == ORC_FLOAT64 || col_type == ORC_BOOLEAN; /* + String */
    case ORC_INTERP_NONE: return col_type == ORC_INT64 || col_type == ORC_FLOAT64 || col_type == ORC_BOOLEAN;
    case ORC_INTERP_CONST: return col_type == ORC_INT64 || col_type == ORC_FLOAT64;         /* interpolation_test.go:16 */
    default: return 0;
    }
}

static val_t apply_interp(const orc_interp_t *ip, interp_state_t *st, const orc_col_t *cols,
                          int64_t nrows, int ts_col, const win_t *w) {
    view_t full = {cols, 0, nrows};
    switch (ip->kind) {
    case ORC_INTERP_WINDOW_START: return val_i64(w->first_value);
    case ORC_INTERP_CONST: return val_f64(ip->const_value);
    case ORC_INTERP_NONE: return val_nil();
    case ORC_INTERP_LINEAR: { /* linear.go:12-37 */
        if (w->first_index == 0 && ip->has_prev_row) {
            st->prev_t0 = ip->prev_t;
            st->prev_v0 = ip->prev_v;
            st->prev_valid = ip->prev_t_valid && ip->prev_v_valid;
        }
        double t0, v0;
        int64_t prev = view_prev_f64s(&full, ts_col, ip->col, w->first_index - 1, &t0, &v0);
        if (prev == -1) {
            if (!st->prev_valid) return val_nil();
            t0 = st->prev_t0;
            v0 = st->prev_v0;
        }
        double t2, v2;
        int64_t next = view_next_f64s(&full, ts_col, ip->col, w->first_index, &t2, &v2);
        if (next == -1) return val_nil();
        double coef = ((double)w->first_value - t0) / (t2 - t0);
        return val_f64(((v2 - v0) * coef) + v0);
    }
    case ORC_INTERP_STEP_PREVIOUS: { /* stepprevious.go:11-24 */
        if (w->first_index == 0 && ip->has_prev_row) {
            if (!ip->prev_v_valid) st->prev_val = val_nil();
            else if (cols[ip->col].type == ORC_INT64) st->prev_val = val_i64(ip->prev_v_i64);
            else if (cols[ip->col].type == ORC_BOOLEAN) st->prev_val = val_bool(ip->prev_v != 0.);
            else st->prev_val = val_f64(ip->prev_v);
        }
        /* GetPrevValues(ts, col, FirstIndex-1): bowgetters.go:95-107 => previous row where both valid */
        int64_t i = w->first_index - 1;
        while (i >= 0 && i < nrows) {
            int64_t i1 = view_prev_valid(&full, ts_col, i);
            int64_t i2 = view_prev_valid(&full, ip->col, i1);
            if (i1 == i2) { i = i1; break; }
            i = i1 - 1;
            if (i1 < 0) { i = -1; break; }
        }
        if (i >= 0 && i < nrows) {
            val_t v = view_get_value(&full, ip->col, i);
            if (v.tag != VAL_NIL) st->prev_val = v;
        }
        return st->prev_val;
    }
    default: return val_nil();
    }
}

static void out_copy_row(orc_out_t *o, int64_t dst, const orc_col_t *c, int64_t row) {
    int valid = col_is_valid(c, row);
    switch (c->type) {
    case ORC_INT64: ((int64_t *)o->values)[dst] = ((const int64_t *)c->values)[c->offset + row]; break;
    case ORC_FLOAT64: ((double *)o->values)[dst] = ((const double *)c->values)[c->offset + row]; break;
    case ORC_BOOLEAN:
        if (bit_is_set((const uint8_t *)c->values, c->offset + row)) bit_set((uint8_t *)o->values, dst);
        else bit_clear((uint8_t *)o->values, dst);
        break;
    default: break;
    }
    if (valid) bit_set(o->validity, dst); else bit_clear(o->validity, dst);
}

int orc_interpolate(const orc_col_t *cols, int ncols, int ts_col, int64_t interval, int64_t offset,
                    int inclusive, const orc_interp_t *interps, int ninterps, orc_out_t *outs,
                    int64_t *n_out) {
    if (ts_col < 0 || ts_col >= ncols) return ORC_ERR_BAD_COL;
    int64_t s0, W;
    int rc = orc_plan_windows(&cols[ts_col], interval, offset, &s0, &W);
    if (rc) return rc;
    /* Interpolate: interpolation.go:30-69 */
    if (ninterps == 0) return ORC_ERR_ARG;
    int nic = -1;
    for (int i = 0; i < ninterps; i++) { /* validateInterpolation :71-96 */
        if (interps[i].col < 0 || interps[i].col >= ncols) return ORC_ERR_BAD_COL;
        if (!interp_type_ok(interps[i].kind, cols[interps[i].col].type)) return ORC_ERR_TYPE;
        if (interps[i].col == ts_col) nic = i;
    }
    if (nic == -1) return ORC_ERR_KEEP_INTERVAL;
    /* AppendBows needs window.Bow (all columns, bow order) and the synthetic row (interps order)
     * to share one schema (bowappend.go:11-13): interps must be the identity column list. */
    if (ninterps != ncols) return ORC_ERR_UNSUPPORTED;
    for (int i = 0; i < ninterps; i++) if (interps[i].col != i) return ORC_ERR_UNSUPPORTED;

    int64_t nrows = cols[ts_col].length;
    interp_state_t *st = (interp_state_t *)calloc((size_t)ninterps, sizeof(interp_state_t));
    int filling = outs != NULL;
    if (filling)
        for (int i = 0; i < ninterps; i++) out_init(&outs[i], *n_out, cols[interps[i].col].type);

    /* interpolateWindows: interpolation.go:98-116 */
    iter_t it = {&cols[ts_col], interval, inclusive, s0, 0, 0};
    int64_t pos = 0;
    while (iter_has_next(&it)) {
        win_t w;
        int64_t wi;
        iter_next(&it, &w, &wi);
        /* interpolateWindow: :118-161 */
        view_t wv = {cols, w.begin, w.end - w.begin};
        int64_t first_col_value = -1;
        if (wv.len > 0) {
            int64_t i;
            double f = view_next_f64(&wv, ts_col, 0, &i); /* :121 */
            if (i > -1) first_col_value = go_f64_to_i64(f);
        }
        if (first_col_value == w.first_value) {
            for (int k = 0; k < ninterps; k++) (void)apply_interp(&interps[k], &st[k], cols, nrows, ts_col, &w); /* :129-134 */
        } else {
            for (int k = 0; k < ninterps; k++) {
                val_t v = apply_interp(&interps[k], &st[k], cols, nrows, ts_col, &w); /* :144 */
                if (filling) out_set_or_drop(&outs[k], pos, v);                            /* :149-150 */
            }
            pos++;
        }
        for (int64_t r = 0; r < wv.len; r++) {
            if (filling)
                for (int k = 0; k < ninterps; k++) out_copy_row(&outs[k], pos, &cols[interps[k].col], wv.begin + r);
            pos++;
        }
    }
    free(st);
    if (!filling) *n_out = pos;
    return ORC_OK;
}

/* ---------------------------------------------------------------- bowassertion.go / bowfill.go */

int orc_is_col_empty(const orc_col_t *col) {
    /* bowassertion.go:84-86: NullN() == Len() */
    for (int64_t i = 0; i < col->length; i++)
        if (col_is_valid(col, i)) return 0;
    return 1;
}

int orc_is_col_sorted(const orc_col_t *col) {
    /* bowassertion.go:15-81 */
    if (orc_is_col_empty(col)) return 0;
    enum { UNDEF, ASC, DESC } order = UNDEF;
    int64_t row = 0;
    if (col->type == ORC_INT64) {
        const int64_t *values = (const int64_t *)col->values + col->offset;
        while (!col_is_valid(col, row)) row++;
        int64_t curr = values[row], next;
        row++;
        for (; row < col->length; row++) {
            if (!col_is_valid(col, row)) continue;
            next = values[row];
            if (order == UNDEF) { if (curr < next) order = ASC; else if (curr > next) order = DESC; }
            if ((order == ASC && next < curr) || (order == DESC && next > curr)) return 0;
            curr = next;
        }
    } else if (col->type == ORC_FLOAT64) {
        const double *values = (const double *)col->values + col->offset;
        while (!col_is_valid(col, row)) row++;
        double curr = values[row], next;
        row++;
        for (; row < col->length; row++) {
            if (!col_is_valid(col, row)) continue;
            next = values[row];
            if (order == UNDEF) { if (curr < next) order = ASC; else if (curr > next) order = DESC; }
            if ((order == ASC && next < curr) || (order == DESC && next > curr)) return 0;
            curr = next;
        }
    } else {
        return 0;
    }
    return 1;
}

int orc_fill_linear(const orc_col_t *cols, int ncols, int ref_col, int fill_col, orc_out_t *out,
                    int *unchanged) {
    /* bowfill.go:14-103 */
    if (ref_col < 0 || ref_col > ncols - 1) return ORC_ERR_BAD_COL;
    if (fill_col < 0 || fill_col > ncols - 1) return ORC_ERR_BAD_COL;
    if (ref_col == fill_col) return ORC_ERR_ARG;
    if (cols[ref_col].type != ORC_INT64 && cols[ref_col].type != ORC_FLOAT64) return ORC_ERR_TYPE;
    const orc_col_t *fc = &cols[fill_col];
    int64_t n = fc->length;
    view_t b = {cols, 0, n};
    *unchanged = 0;
    /* NewBufferFromCol: copy of the column */
    out_init(out, n, fc->type);
    for (int64_t i = 0; i < n; i++) out_copy_row(out, i, fc, i);

    if (orc_is_col_empty(&cols[ref_col])) { *unchanged = 1; return ORC_OK; } /* :35-37 */
    if (!orc_is_col_sorted(&cols[ref_col])) return ORC_ERR_NOT_SORTED;        /* :39-42 */
    if (fc->type != ORC_INT64 && fc->type != ORC_FLOAT64) return ORC_ERR_TYPE; /* :44-51 */
    int64_t nulls = 0;
    for (int64_t i = 0; i < n; i++) nulls += !col_is_valid(fc, i);
    if (nulls == 0) { *unchanged = 1; return ORC_OK; } /* :53-55 */

    for (int64_t row = 0; row < n; row++) { /* :65-97 */
        if (col_is_valid(fc, row)) continue; /* buf.IsValid: the buffer's bitmap is only ever set for null rows we skip past */
        int64_t row_prev, row_next;
        double prev_fill = view_prev_f64(&b, fill_col, row - 1, &row_prev);
        double next_fill = view_next_f64(&b, fill_col, row + 1, &row_next);
        double row_ref, prev_ref, next_ref;
        int v1 = view_get_f64(&b, ref_col, row, &row_ref);
        int v2 = view_get_f64(&b, ref_col, row_prev, &prev_ref);
        int v3 = view_get_f64(&b, ref_col, row_next, &next_ref);
        if (!v1 || !v2 || !v3) continue;
        if (next_ref - prev_ref == 0) { /* :78-85 — then falls through */
            if (fc->type == ORC_INT64) out_set_or_drop_strict(out, row, val_i64(go_f64_to_i64(prev_fill)));
            else out_set_or_drop_strict(out, row, val_f64(prev_fill));
        }
        double tmp = row_ref - prev_ref; /* :87-90 */
        tmp /= next_ref - prev_ref;
        tmp *= next_fill - prev_fill;
        tmp += prev_fill;
        if (fc->type == ORC_INT64) out_set_or_drop_strict(out, row, val_i64(go_f64_to_i64(round(tmp)))); /* math.Round :93 */
        else out_set_or_drop_strict(out, row, val_f64(tmp));
    }
    return ORC_OK;
}

int orc_fill(const orc_col_t *col, int method, orc_out_t *out, int *unchanged) {
    /* FillPrevious / FillNext: bowfill.go:166-253 (fill + getFillRowIndex); FillMean: bowfill.go:108-160.
     * One column per call (the reference runs one goroutine per selected column). */
    if (method != ORC_FILL_PREVIOUS && method != ORC_FILL_NEXT && method != ORC_FILL_MEAN) return ORC_ERR_ARG;
    if (col->type != ORC_INT64 && col->type != ORC_FLOAT64) return ORC_ERR_TYPE; /* :115-123 (Mean); Bool/String: not restated */
    int64_t n = col->length;
    view_t b = {col, 0, n};
    out_init(out, n, col->type);
    for (int64_t i = 0; i < n; i++) out_copy_row(out, i, col, i); /* NewBufferFromCol */
    int64_t nulls = 0;
    for (int64_t i = 0; i < n; i++) nulls += !col_is_valid(col, i);
    *unchanged = nulls == 0; /* NullN() == 0: the column is passed through (:130-133, :176-179) */
    if (nulls == 0) return ORC_OK;
    for (int64_t row = 0; row < n; row++) {
        if (col_is_valid(col, row)) continue;
        if (method == ORC_FILL_MEAN) {
            int64_t prev_row, next_row;
            double prev_val = view_prev_f64(&b, 0, row - 1, &prev_row); /* :145 */
            double next_val = view_next_f64(&b, 0, row + 1, &next_row); /* :146 */
            if (prev_row > -1 && next_row > -1) {
                if (col->type == ORC_INT64) out_set_or_drop_strict(out, row, val_i64(go_f64_to_i64(round((prev_val + next_val) / 2)))); /* :150 */
                else out_set_or_drop_strict(out, row, val_f64((prev_val + next_val) / 2)); /* :152 */
            }
        } else {
            int64_t src = method == ORC_FILL_PREVIOUS ? view_prev_valid(&b, 0, row - 1) : view_next_valid(&b, 0, row + 1); /* :247-253 */
            if (src > -1) out_copy_row(out, row, col, src); /* arr.Value(fillRowIndex), valid (:196-199, :207-210) */
        }
    }
    return ORC_OK;
}

/* ---------------------------------------------------------------- synthetic generators */

uint64_t orc_mix64(uint64_t seed, uint64_t i) {
    /* splitmix64 finaliser over a counter; restates bow_amd/csrc/generate.hip */
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

void orc_gen_dense(int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val) {
    for (int64_t i = 0; i < n; i++) {
        uint64_t r = (uint64_t)(row0 + i);
        ts[i] = row0 + i;
        val[i] = (double)(orc_mix64(seed, r) >> 11) * 0x1.0p-53;
    }
}

void orc_gen_sparse(int64_t row0, int64_t n, uint64_t seed, int64_t *ts, double *val,
                    uint8_t *validity) {
    for (int64_t i = 0; i < n; i++) {
        uint64_t r = (uint64_t)(row0 + i);
        uint64_t h = orc_mix64(seed, r);
        ts[i] = 10 * (row0 + i) + (int64_t)(h % 10);
        val[i] = (double)((h >> 16) % 10) + 0.5;
        int valid = ((h >> 32) % 10) > 2;
        int64_t bit = (row0 + i) - ((row0 >> 3) << 3);
        if (valid) bit_set(validity, bit); else bit_clear(validity, bit);
    }
}
