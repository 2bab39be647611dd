// extras.cpp — C-ABI entry points either side of Rolling.Aggregate: window bounds (the iterator),
// Rolling.Interpolate, Bow.FillLinear / IsColSorted and the whole-frame aggregation.Aggregate.
// Host code here only validates (mirroring the reference's drivers), prepares residency and
// orchestrates the HIP kernels of interp_fill.hip; no column data is reduced on the CPU.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

using namespace bowgpu;

namespace {

// a device-resident output of `slots` rows whose validity is produced as one byte per row and packed
struct ByteOut {
    DevOut out;
    DevBuf bytes;
};

int copy_or_alias(Ctx *c, void *user, int residency, const void *dev, size_t nbytes) {
    if (nbytes == 0 || !user) return 0;
    if (residency == BOWGPU_DEVICE) BG_HIP(hipMemcpyAsync(user, dev, nbytes, hipMemcpyDeviceToDevice, c->stream));
    else BG_TRY(copy_d2h(c, user, dev, nbytes, residency == BOWGPU_HOST_PINNED));
    return 0;
}

int ts_contract(Ctx *c, const bowgpu_col *tsc) {
    if (tsc->validity && tsc->null_count != 0) {
        DevCol probe;
        BG_TRY(devcol_prepare(c, tsc, &probe, false, true));
        if (probe.null_count > 0) return fail(BOWGPU_ERR_TS_NULLS, "interval column has %lld nulls: outside the device path", (long long)probe.null_count);
    }
    return 0;
}

int ts_device(Ctx *c, const bowgpu_col *tsc, DevCol *dts) {
    bowgpu_col t = *tsc;
    t.validity = nullptr;
    t.null_count = 0;
    return devcol_prepare(c, &t, dts, true, false);
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------- window bounds
int bowgpu_window_bounds(const bowgpu_col *ts, int64_t interval, const bowgpu_options *opts, int64_t *first_index,
                         int64_t *slice_begin, int64_t *slice_end, uint8_t *is_inclusive, int32_t residency) {
    if (!ts) return fail(BOWGPU_ERR_ARG, "null argument");
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    Plan plan;
    BG_TRY(plan_make(nullptr, ts, interval, o.offset, &plan));
    if (plan.W == 0) return 0;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_TRY(ts_contract(c, ts));
    DevCol dts;
    BG_TRY(ts_device(c, ts, &dts));
    const int64_t n = ts->length, W = plan.W;
    DevBuf first_idx, d_fi, d_sb, d_se, d_in;
    BG_TRY(first_idx.alloc((size_t)(W + 1) * 8));
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint32_t *status = reinterpret_cast<uint32_t *>(dscr);
    BG_HIP(hipMemsetAsync(status, 0, 64, c->stream));
    BG_TRY(launch_window_first_rows(c, reinterpret_cast<const int64_t *>(dts.values), n, plan, reinterpret_cast<int64_t *>(first_idx.p), status));
    const bool dev = residency == BOWGPU_DEVICE;
    int64_t *pfi = first_index, *psb = slice_begin, *pse = slice_end;
    uint8_t *pin = is_inclusive;
    if (!dev) {
        if (first_index) { BG_TRY(d_fi.alloc((size_t)W * 8)); pfi = reinterpret_cast<int64_t *>(d_fi.p); }
        if (slice_begin) { BG_TRY(d_sb.alloc((size_t)W * 8)); psb = reinterpret_cast<int64_t *>(d_sb.p); }
        if (slice_end) { BG_TRY(d_se.alloc((size_t)W * 8)); pse = reinterpret_cast<int64_t *>(d_se.p); }
        if (is_inclusive) { BG_TRY(d_in.alloc((size_t)W)); pin = reinterpret_cast<uint8_t *>(d_in.p); }
    }
    BG_TRY(launch_window_bounds(c, reinterpret_cast<const int64_t *>(dts.values), n, plan, o.inclusive ? 1 : 0,
                                plan.s0 > plan.first_ts ? 1 : 0, reinterpret_cast<const int64_t *>(first_idx.p), pfi, psb, pse, pin));
    uint32_t hstat[4] = {0, 0, 0, 0};
    BG_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, c->stream));
    if (!dev) {
        if (first_index) BG_TRY(copy_d2h(c, first_index, pfi, (size_t)W * 8));
        if (slice_begin) BG_TRY(copy_d2h(c, slice_begin, psb, (size_t)W * 8));
        if (slice_end) BG_TRY(copy_d2h(c, slice_end, pse, (size_t)W * 8));
        if (is_inclusive) BG_TRY(copy_d2h(c, is_inclusive, pin, (size_t)W));
    }
    BG_HIP(hipStreamSynchronize(c->stream));
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    return 0;
}

// ---------------------------------------------------------------------------- Interpolate
static bool interp_type_ok(int kind, int t) {
    switch (kind) {
    case BOWGPU_INTERP_WINDOW_START: return t == BOWGPU_INT64;                         // interpolation/windowstart.go:9
    case BOWGPU_INTERP_LINEAR: return t == BOWGPU_INT64 || t == BOWGPU_FLOAT64;        // interpolation/linear.go:11
    case BOWGPU_INTERP_STEP_PREVIOUS:                                                  // stepprevious.go:10 (+ Boolean, String)
    case BOWGPU_INTERP_NONE: return t == BOWGPU_INT64 || t == BOWGPU_FLOAT64 || t == BOWGPU_BOOLEAN || (kind == BOWGPU_INTERP_STEP_PREVIOUS && t == BOWGPU_STRING);
    case BOWGPU_INTERP_CONST: return t == BOWGPU_INT64 || t == BOWGPU_FLOAT64;
    default: return false;
    }
}

static const char *type_name(int t) {
    return t == BOWGPU_FLOAT64 ? "float64" : t == BOWGPU_INT64 ? "int64" : t == BOWGPU_BOOLEAN ? "bool" : t == BOWGPU_STRING ? "utf8" : "undefined";
}

extern "C++" int bowgpu::interp_validate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_options *o,
                    const bowgpu_interp *interps, int32_t ninterps) {
    // Interpolate + validateInterpolation: reference rolling/interpolation.go:30-96
    if (ninterps <= 0) return fail(BOWGPU_ERR_ARG, "at least one column interpolation is required");
    int nic = -1;
    for (int i = 0; i < ninterps; i++) {
        if (interps[i].col < 0 || interps[i].col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "interpolation %d: no column with index %d", i, interps[i].col);
        const int t = cols[interps[i].col].type;
        if (!interp_type_ok(interps[i].kind, t)) {
            const char *acc = interps[i].kind == BOWGPU_INTERP_WINDOW_START ? "[int64]"
                              : interps[i].kind == BOWGPU_INTERP_LINEAR ? "[int64 float64]"
                              : interps[i].kind == BOWGPU_INTERP_STEP_PREVIOUS ? "[int64 float64 bool utf8]"
                              : interps[i].kind == BOWGPU_INTERP_NONE ? "[int64 float64 bool]" : "[int64 float64]";
            return fail(BOWGPU_ERR_TYPE, "accepts types %s, got type %s", acc, type_name(t));
        }
        if (interps[i].col == ts_col) nic = i;
    }
    if (nic == -1) return fail(BOWGPU_ERR_KEEP_INTERVAL, "must keep interval column");
    // AppendBows(startBow, window.Bow) needs one schema (bowappend.go:11-13): the interpolators must list the columns in order
    if (ninterps != ncols) return fail(BOWGPU_ERR_UNSUPPORTED, "interpolators must cover every column of the Bow, in order (bowappend.go:11-13)");
    for (int i = 0; i < ninterps; i++) {
        if (interps[i].col != i) return fail(BOWGPU_ERR_UNSUPPORTED, "interpolators must cover every column of the Bow, in order (bowappend.go:11-13)");
        const int t = cols[i].type;
        if (t != BOWGPU_INT64 && t != BOWGPU_FLOAT64) return fail(BOWGPU_ERR_UNSUPPORTED, "column type %s is outside the device path", type_name(t));
    }
    return 0;
}

struct InterpJob {
    Plan plan;
    DevCol dts;
    std::vector<DevCol> dcols;
    void *tile_local = nullptr, *super_before = nullptr, *super_sum = nullptr;  // context pool (no hipMalloc / hipFree per call)
    int64_t kq = -1;  // window whose start is -1 (InterpParams::kq)
    int64_t drop = 0; // leading rows that belong to no window (InterpParams::drop)
    int64_t M = 0;    // output rows - input rows
    int kq_empty = 0; // window kq has no row of its own
    int e0 = 0;       // inclusive windows: row 0 sits exactly on the first window's start
    int has_left = 0; // sharded Interpolate: rows exist to the left, the last of them at left_ts, in window wbase - 1
    int64_t left_ts = 0, wbase = 0;
    bool from_cache = false;   // pass 1 was not run by this call: its results are the preceding _count's
};

// pass 1 of interpolate.hip: exact heads per tile, their exclusive scan, M = synthetic rows
// shard: global_s0 + edge of a row-range shard (nullptr: the whole frame)
static bool interp_cache_hit(Ctx *c, const bowgpu_col *ts, int64_t interval, int64_t raw_offset, int inclusive, const int64_t *global_s0,
                             const bowgpu_interp_edge *edge) {
    const Ctx::InterpCache &k = c->interp_cache;
    if (!k.valid || ts->residency != BOWGPU_DEVICE) return false;   // (a host column is staged into a fresh device copy per call)
    if (k.ts_values != ts->values || k.ts_offset != ts->offset || k.n != ts->length || k.interval != interval || k.raw_offset != raw_offset) return false;
    if (k.sharded != (global_s0 != nullptr) || k.inclusive != (inclusive ? 1 : 0)) return false;
    if (global_s0 && (k.global_s0 != *global_s0 || k.has_left != ((edge && edge->has_left) ? 1 : 0) || (k.has_left && k.left_ts != edge->left_last_ts))) return false;
    if (k.epoch != device_write_epoch()) return false;   // something was written to / freed from device memory through the library since the count
    return k.gen == c->pool_gen[kPoolInterp + 1] && c->pool[kPoolInterp + 1] != nullptr && k.gen0 == c->pool_gen[kPoolInterp + 0] && c->pool[kPoolInterp + 0] != nullptr;
}

static int interp_prepare(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                          const bowgpu_options *o, InterpJob *job, const int64_t *global_s0 = nullptr, const bowgpu_interp_edge *edge = nullptr,
                          bool use_cache = false, const Plan *probe = nullptr) {
    // The _fill call right after _count on the same device-resident, unchanged interval column (Bows are immutable in the
    // reference; include/bowgpu.h, "Rolling.Interpolate", states the contract): pass 1's prefix is still in the context pool.  A
    // write or free THROUGH the library in between drops it (device_write_epoch); the fill kernel checks the output count it
    // arrives at against the cached one and the call fails with BOWGPU_ERR_ARG when they differ (status[6])
    if (use_cache && interp_cache_hit(c, &cols[ts_col], interval, o->offset, o->inclusive, global_s0, edge)) {
        const Ctx::InterpCache &k = c->interp_cache;
        for (int i = 0; i < ncols; i++)
            if (cols[i].length != k.n) return fail(BOWGPU_ERR_ARG, "column %d has a different length", i);
        job->plan.interval = interval; job->plan.offset = k.offset_norm; job->plan.s0 = k.s0; job->plan.W = k.W;
        job->plan.first_ts = k.first_ts; job->plan.last_ts = k.last_ts; job->plan.magic = magic_make((uint64_t)interval);
        job->kq = k.kq; job->drop = k.drop; job->M = k.M; job->wbase = k.wbase; job->has_left = k.has_left; job->left_ts = k.left_ts;
        job->kq_empty = k.kq_empty; job->e0 = k.e0;
        job->tile_local = c->pool[kPoolInterp + 0];
        job->super_before = c->pool[kPoolInterp + 1];
        job->from_cache = true;
        BG_TRY(ts_device(c, &cols[ts_col], &job->dts));
        c->interp_cache.valid = false;   // one use: the outputs of this fill may be what the next call reads
        return 0;
    }
    c->interp_cache.valid = false;
    if (probe) job->plan = *probe;   // (the caller's constructor checks already fetched the column's first and last timestamp: one round trip, not two)
    else BG_TRY(plan_make(c, &cols[ts_col], interval, o->offset, &job->plan));
    if (global_s0) {
        // a shard: windows are counted from the frame's s0; the shard accounts for the windows after its left neighbours' last one
        Plan &pl = job->plan;
        pl.s0 = *global_s0;
        if (cols[ts_col].length > 0) {
            // rows below the first window start (Go's truncating division on a negative first timestamp: rolling.go:96-99) ride in window 0
            // or belong to no window at all - which of the two depends on the first row AT OR ABOVE s0 (interp_quirk_kernel).  They are the
            // frame's first rows: the frame's first shard serves them as the unsharded call does when that row is its own; the corner that
            // is left - a first shard of nothing but such rows, fewer than one interval's worth of time - is declined.
            if (pl.first_ts < pl.s0 && ((edge && edge->has_left) || pl.last_ts < pl.s0))
                return fail(BOWGPU_ERR_UNSUPPORTED, "sharded interpolate: a shard of nothing but rows below the first window start is outside the sharded path");
            const int64_t wl = (int64_t)(((uint64_t)pl.last_ts - (uint64_t)pl.s0) / (uint64_t)interval);
            job->has_left = edge && edge->has_left ? 1 : 0;
            if (job->has_left) {
                if (edge->left_last_ts < pl.s0 || edge->left_last_ts > pl.first_ts) return fail(BOWGPU_ERR_ARG, "sharded interpolate: the left neighbour's last timestamp does not precede this shard");
                job->left_ts = edge->left_last_ts;
                job->wbase = (int64_t)(((uint64_t)edge->left_last_ts - (uint64_t)pl.s0) / (uint64_t)interval) + 1;
            }
            pl.W = wl + 1 - job->wbase;  // windows this shard accounts for
        }
    }
    const int64_t n = cols[ts_col].length, W = job->plan.W;
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has a different length", i);
    job->M = 0;
    if (n == 0) return 0;
    BG_TRY(ts_contract(c, &cols[ts_col]));
    BG_TRY(ts_device(c, &cols[ts_col], &job->dts));
    if (W == 0 && !global_s0) { job->M = -n; return 0; }  // no window at all (every row lies below s0): the concatenation of zero window bows is empty
    const Plan &pl = job->plan;
    job->kq = -1;
    if (pl.s0 <= -1 && (uint64_t)(-1 - pl.s0) % (uint64_t)pl.interval == 0) {
        // (a shard accounts for windows wbase .. wbase + W - 1 of the frame; the one that starts at -1 concerns the shard that accounts for it:
        // its rows, if it has any, can only be that shard's own - the left neighbours end in an earlier window, and when this shard ends
        // in a later one nothing of the window lies to its right)
        const int64_t k = (int64_t)((uint64_t)(-1 - pl.s0) / (uint64_t)pl.interval);
        if (k >= job->wbase && k - job->wbase < W) job->kq = k;
    }
    const int64_t ntiles = 2 * interp_tiles(n);   // exact heads are counted per 256 rows: two entries per tile of the count kernel
    const int64_t nsuper = interp_supers(n);
    BG_TRY(ctx_pool(c, kPoolInterp + 0, (size_t)ntiles * 4 + 16, &job->tile_local));
    BG_TRY(ctx_pool(c, kPoolInterp + 1, (size_t)(nsuper + 1) * 8, &job->super_before));
    BG_TRY(ctx_pool(c, kPoolInterp + 2, (size_t)nsuper * 4 + 16, &job->super_sum));
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint32_t *status = reinterpret_cast<uint32_t *>(dscr);
    BG_HIP(hipMemsetAsync(status, 0, 64, c->stream));
    const int64_t *ts = reinterpret_cast<const int64_t *>(job->dts.values);
    int64_t *d_total = reinterpret_cast<int64_t *>(status + 4);
    int64_t *hback;   // the pass' findings arrive in the registered block by the scan kernel's own stores
    BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&hback)));
    hback += 384;     // bytes 3072.. of it
    BG_TRY(launch_interp_count(c, ts, n, pl, job->kq, job->has_left, job->left_ts, reinterpret_cast<int32_t *>(job->tile_local),
                               reinterpret_cast<int32_t *>(job->super_sum), reinterpret_cast<int64_t *>(job->super_before), d_total, status, hback));
    BG_HIP(hipStreamSynchronize(c->stream));
    const uint32_t hstat[4] = {(uint32_t)(uint64_t)hback[1], (uint32_t)((uint64_t)hback[1] >> 32), (uint32_t)(uint64_t)hback[2], (uint32_t)((uint64_t)hback[2] >> 32)};
    const int64_t total = hback[0];
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    job->drop = (int64_t)(((uint64_t)hstat[3] << 32) | hstat[2]);
    job->kq_empty = hstat[1] ? 1 : 0;
    if (o->inclusive) {
        // Inclusive windows (rolling.go:201-209 + interpolation.go:98-116): a row on a window's end is also the last row of that
        // window's bow, so it appears twice in the concatenation - every window then contributes exactly one row in front of
        // its first row (a synthetic row, or that copy), except window 0 when row 0 sits on its start: n + W - e0 rows.  The
        // whole-trip wave kernel takes them; its preconditions (interp_fast32, no dropped rows, no -1 sentinel window) bound the shape.
        // A shard (global_s0): the same statement over the windows the shard accounts for (those behind its left neighbours' last
        // one) - the copy of a row that sits on its window's start travels with the row itself, and only the FRAME's first row can
        // be the exact row 0 that gets nothing in front.
        if (!(interp_fast32(pl, job->kq) || interp_wide32(pl, job->kq)) || job->drop != 0)
            return fail(BOWGPU_ERR_UNSUPPORTED, "Interpolate on inclusive windows: rows below s0, intervals of 2^31 and more or timestamps beyond 2^53 are outside the device path");
        job->e0 = (!job->has_left && pl.first_ts == pl.s0) ? 1 : 0;
        job->M = W - job->e0;
    } else
    job->M = W - total - ((job->kq >= 0 && hstat[1]) ? 1 : 0) - job->drop;  // rows added (synthetic) minus rows dropped
    {
        Ctx::InterpCache &k = c->interp_cache;
        const bowgpu_col *tsc = &cols[ts_col];
        k.ts_values = tsc->values; k.ts_offset = tsc->offset; k.n = n; k.interval = interval; k.raw_offset = o->offset;
        k.sharded = global_s0 != nullptr; k.global_s0 = global_s0 ? *global_s0 : 0;
        k.has_left = job->has_left; k.left_ts = job->left_ts;
        k.gen = c->pool_gen[kPoolInterp + 1];
        k.gen0 = c->pool_gen[kPoolInterp + 0];
        k.epoch = device_write_epoch();
        k.s0 = pl.s0; k.W = pl.W; k.first_ts = pl.first_ts; k.last_ts = pl.last_ts; k.offset_norm = pl.offset;
        k.kq = job->kq; k.drop = job->drop; k.M = job->M; k.wbase = job->wbase; k.kq_empty = job->kq_empty;
        k.inclusive = o->inclusive ? 1 : 0; k.e0 = job->e0;
        k.valid = tsc->residency == BOWGPU_DEVICE;
    }
    return 0;
}

// Rolling.Interpolate on INCLUSIVE windows as ONE pass over the rows (a _fill that does not follow its _count): every window then
// contributes exactly one row in front of its first - n_out = n + W - e0, and a trip's output position depends on nothing a count
// pass would tell - so there is no count pass; the fill kernel checks the order of the interval column itself.  (Exclusive windows
// need the number of rows sitting exactly on a window start before each trip.  A decoupled look-back inside the fill kernel was
// built and measured in round 4: 1.67 ms against 1.12 ms for count pass + fill at 1e8 rows - an uncached word per trip each way
// costs more than the 0.16 ms count pass it replaces; not kept.)  Taken for the shapes interp_wave3_kernel serves on its own: the
// whole frame (no shard), no -1 sentinel window, no rows below s0.  *applies = false: the two-pass path (interp_prepare).
static int interp_onepass_prepare(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *o,
                                  InterpJob *job, bool *applies, const Plan *probe = nullptr) {
    *applies = false;
    if (!o->inclusive) return 0;
    const int64_t n = cols[ts_col].length;
    if (n <= 0) return 0;
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has a different length", i);
    if (probe) job->plan = *probe;
    else BG_TRY(plan_make(c, &cols[ts_col], interval, o->offset, &job->plan));
    const Plan &pl = job->plan;
    if (pl.W <= 0 || pl.first_ts < pl.s0 || pl.s0 <= -1) return 0;   // (s0 <= -1: a window may start at -1 - the reference's sentinel)
    if (!(interp_fast32(pl, -1) || interp_wide32(pl, -1))) return 0;
    BG_TRY(ts_contract(c, &cols[ts_col]));
    BG_TRY(ts_device(c, &cols[ts_col], &job->dts));
    job->kq = -1; job->drop = 0; job->kq_empty = 0; job->has_left = 0; job->wbase = 0;
    job->e0 = (o->inclusive && pl.first_ts == pl.s0) ? 1 : 0;
    job->M = pl.W - job->e0;
    c->interp_cache.valid = false;
    *applies = true;
    return 0;
}

static int interp_count_impl(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                             const bowgpu_interp *interps, int32_t ninterps, int64_t *n_out, const int64_t *global_s0,
                             const bowgpu_interp_edge *edge) {
    if (!cols || ncols <= 0 || !n_out) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    if (edge && ncols > kMaxCols) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded Interpolate: at most %d columns (bowgpu_interp_edge)", kMaxCols);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    Plan probe;
    BG_TRY(plan_make(nullptr, &cols[ts_col], interval, o.offset, &probe));  // ctor errors first (rolling.go:69-112)
    BG_TRY(interp_validate(cols, ncols, ts_col, &o, interps, ninterps));
    if (cols[ts_col].length == 0) { *n_out = 0; return 0; }
    Ctx *c;
    BG_TRY(ctx_get(&c));
    InterpJob job;
    BG_TRY(interp_prepare(c, cols, ncols, ts_col, interval, &o, &job, global_s0, edge, false, &probe));
    *n_out = cols[ts_col].length + job.M;
    return 0;
}

static int interp_fill_impl(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                            const bowgpu_interp *interps, int32_t ninterps, bowgpu_out *outs, const int64_t *global_s0,
                            const bowgpu_interp_edge *edge) {
    if (!cols || ncols <= 0 || !outs) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    if (edge && ncols > kMaxCols) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded Interpolate: at most %d columns (bowgpu_interp_edge)", kMaxCols);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    Ctx *c = nullptr;
    const bool cached = cols[ts_col].length > 0 && ctx_get(&c) == 0 && interp_cache_hit(c, &cols[ts_col], interval, o.offset, o.inclusive, global_s0, edge);
    Plan probe;
    if (!cached) BG_TRY(plan_make(nullptr, &cols[ts_col], interval, o.offset, &probe));   // (the _count call that filled the cache ran these checks on the same arguments)
    BG_TRY(interp_validate(cols, ncols, ts_col, &o, interps, ninterps));
    const int64_t n = cols[ts_col].length;
    if (n == 0) {
        for (int i = 0; i < ninterps; i++) { outs[i].length = 0; outs[i].null_count = 0; outs[i].type = cols[i].type; }
        return 0;
    }
    BG_TRY(ctx_get(&c));
    c->last_slow_rows = 0;
    InterpJob job;
    bool onepass = false;
    if (!cached && !global_s0 && !edge) BG_TRY(interp_onepass_prepare(c, cols, ncols, ts_col, interval, &o, &job, &onepass, &probe));
    if (!onepass) BG_TRY(interp_prepare(c, cols, ncols, ts_col, interval, &o, &job, global_s0, edge, true, cached ? nullptr : &probe));
    const int64_t n_out = n + job.M;
    if (n_out == 0) {
        for (int i = 0; i < ninterps; i++) { outs[i].length = 0; outs[i].null_count = 0; outs[i].type = cols[i].type; }
        return 0;
    }
    job.dcols.resize(ncols);
    std::vector<DevOut> douts(ninterps);
    InterpParams P;
    memset(&P, 0, sizeof P);
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    P.ts = reinterpret_cast<const int64_t *>(job.dts.values);
    P.n = n; P.s0 = job.plan.s0; P.interval = job.plan.interval; P.W = job.plan.W; P.magic = job.plan.magic;
    P.tile_local = reinterpret_cast<const int32_t *>(job.tile_local);
    P.super_before = reinterpret_cast<const int64_t *>(job.super_before);
    P.status = reinterpret_cast<uint32_t *>(dscr);
    P.kq = job.kq;
    P.kq_empty = job.kq_empty;
    P.inclusive = o.inclusive ? 1 : 0; P.e0 = job.e0;
    P.drop = job.drop;
    P.has_left = job.has_left; P.left_ts = job.left_ts; P.wbase = job.wbase;
    P.fast32 = interp_fast32(job.plan, job.kq) ? 1 : 0;
    P.wide32 = interp_wide32(job.plan, job.kq) ? 1 : 0;
    if (P.fast32 || P.wide32) interp_magic32(job.plan.interval, &P.m32, &P.sh1_32, &P.sh2_32);
    P.ts_col = ts_col;
    P.n_out = n_out;
    const int64_t ntrips512 = (n + 511) / 512;
    {   // one word per 512-row trip and column of a launch (at most kMaxCols columns per launch), and one valid-output count
        void *ew;
        BG_TRY(ctx_pool(c, kPoolInterpEdge, (size_t)ntrips512 * kMaxCols * 12 + 64, &ew));
        P.edge_words = reinterpret_cast<uint64_t *>(ew);
    }
    uint32_t *trip_valid = reinterpret_cast<uint32_t *>(P.edge_words + ntrips512 * kMaxCols);
    // Device-resident outputs whose bitmaps can take whole 32-bit words - a 4-byte aligned pointer and a capacity that reaches the end
    // of the word holding row n_out - 1 - are WRITTEN IN PLACE by interp_wave3_kernel, which stores every word of [0, n_out) itself
    // and counts the valid outputs per trip: no launch that zeroes the bitmaps in front of it and no pass over them behind it (round 4:
    // preset + finish, 41 us of 1.0 ms at 1e8 rows).  Anything else - a host-resident output, an odd pointer, a capacity of exactly
    // n_out rows that ends mid-word, the workgroup kernel - keeps the working copies from the context pool and those two launches.
    bool can_place = true;
    for (int i = 0; i < ncols; i++) {
        DevCol &dc = job.dcols[i];
        if (i == ts_col) { dc.values = job.dts.values; dc.length = n; dc.type = BOWGPU_INT64; }
        else BG_TRY(devcol_prepare(c, &cols[i], &dc, true, true));
        BG_TRY(devout_prepare(c, &outs[i], n_out, &douts[i], i < 16 ? i : -1));  // validity working copy from the context pool
        can_place = can_place && outs[i].residency == BOWGPU_DEVICE && (reinterpret_cast<uintptr_t>(outs[i].validity) & 3) == 0 &&
                    ((outs[i].length + 7) >> 3) >= 4 * ((n_out + 31) >> 5);
    }
    std::vector<uint64_t> hcnt(ninterps, 0);
    uint32_t hstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long *dcnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dscr) + 1024);
    // The output positions depend on the interval column alone, so the columns go through the kernel kMaxCols at a time (a Bow
    // of any width: interpolation.go:98-161 loops over the interpolators).  Per batch, in place: the kernel + the edge fix (which
    // also sums the counts); through working copies: ONE launch zeroes the batch's bitmaps (and, first batch, the status words), the
    // kernel(s), ONE launch counts the valid bits and copies the bitmaps into the caller's buffers.
    // (with_index: the neighbour index of each nullable column under Linear / StepPrevious - three small launches per column.
    // interp_wave3_kernel finds a synthetic row's neighbour points by a bounded walk and says so when one lies further away
    // (status[7]): the index is built for the workgroup kernel and for that repeat only.)
    auto run_all = [&](int allow_wave2, bool with_index) -> int {
        P.allow_wave2 = allow_wave2;
        const bool place = can_place && interp_takes_wave3(P) && !(route_mask() & BOWGPU_ROUTE_INTERP_COPIES);
        P.in_place = place ? 1 : 0;
        P.trip_valid = place ? trip_valid : nullptr;
        char *hp;
        BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&hp)));
        // (in place: kInterpEdgeBlocks partial counts per column of the batch, every one of them stored - by interp_edge_fix_kernel, straight
        // into the host's registered block, the status words in front of them: no copy command behind the launches)
        P.valid_counts = place ? reinterpret_cast<unsigned long long *>(hp + 1024) : dcnt;
        P.host_status = place ? reinterpret_cast<uint32_t *>(hp) : nullptr;
        static_assert(1024 + 8 * (size_t)kMaxCols * kInterpEdgeBlocks <= 16384, "the registered block holds the status words and the partial counts");
        for (int b0 = 0; b0 < ncols; b0 += kMaxCols) {
            const int nb = ncols - b0 < kMaxCols ? ncols - b0 : kMaxCols;
            const bool last_batch = b0 + nb == ncols;
            P.ncols = nb;
            BitmapBatch bb;
            memset(&bb, 0, sizeof bb);
            bb.n = nb; bb.nbits = n_out; bb.status = P.status; bb.counts = dcnt; bb.status_words = b0 == 0 ? 16 : 0;
            for (int j = 0; j < nb; j++) {
                const int i = b0 + j;
                const DevCol &dc = job.dcols[i];
                InterpCol &ic = P.cols[j];
                memset(&ic, 0, sizeof ic);
                ic.values = reinterpret_cast<const uint64_t *>(dc.values);
                ic.vbits = dc.vbits; ic.vbit0 = dc.vbit0; ic.type = cols[i].type; ic.kind = interps[i].kind;
                ic.const_value = interps[i].const_value;
                ic.has_prev = interps[i].has_prev_row; ic.prev_t_valid = interps[i].prev_t_valid; ic.prev_v_valid = interps[i].prev_v_valid;
                ic.prev_t = interps[i].prev_t; ic.prev_v = interps[i].prev_v; ic.prev_v_i64 = interps[i].prev_v_i64;
                ic.out_values = reinterpret_cast<uint64_t *>(douts[i].values);
                ic.out_valid_words = reinterpret_cast<uint32_t *>(place ? outs[i].validity : douts[i].validity);
                if (edge && edge->next_valid[i]) { ic.next_valid = 1; ic.next_t = edge->next_t[i]; ic.next_v = edge->next_v[i]; }
                if (with_index && dc.vbits && (ic.kind == BOWGPU_INTERP_LINEAR || ic.kind == BOWGPU_INTERP_STEP_PREVIOUS)) {
                    void *ix;
                    BG_TRY(ctx_pool(c, kPoolInterp + 3 + j, nbr_index_bytes(n, dc.vbit0), &ix));  // (reused by the next batch: stream order)
                    BG_TRY(nbr_index_build(c, dc.vbits, dc.vbit0, n, ix, &ic.nbr));
                }
                bb.work[j] = reinterpret_cast<uint32_t *>(douts[i].validity);
                bb.user[j] = outs[i].residency == BOWGPU_DEVICE ? outs[i].validity : nullptr;
                bb.ones[j] = 0;
                bb.count[j] = 1;
            }
            if (place) { if (b0 == 0) BG_HIP(hipMemsetAsync(P.status, 0, 64, c->stream)); }   // (the status words; the partial counts are all stored)
            else BG_TRY(launch_preset_bitmaps(c, bb));
            P.aligned16 = (reinterpret_cast<uintptr_t>(P.ts) & 15) == 0 ? 1 : 0;
            for (int j = 0; j < nb; j++) if (reinterpret_cast<uintptr_t>(P.cols[j].values) & 15) P.aligned16 = 0;
            BG_TRY(launch_interp_tiles(c, P));
            if (!interp_takes_wave3(P) && b0 == 0) c->last_slow_rows += n;   // (interp_tile_kernel: 2.2 ms per 1e8 rows where interp_wave3_kernel takes 1.15)
            if (!place) BG_TRY(launch_finish_bitmaps(c, bb));
            // the batch's counts - and, behind the last batch, the status words in front of them: one copy
            if (!place) {
                if (last_batch) BG_HIP(hipMemcpyAsync(hp, P.status, 1024 + 8 * (size_t)nb, hipMemcpyDeviceToHost, c->stream));
                else BG_HIP(hipMemcpyAsync(hp + 1024, dcnt, 8 * (size_t)nb, hipMemcpyDeviceToHost, c->stream));
            }
            BG_HIP(hipStreamSynchronize(c->stream));   // (per batch: the pinned block is read before the next batch's copy lands in it)
            const unsigned long long *hc = reinterpret_cast<const unsigned long long *>(hp + 1024);
            for (int j = 0; j < nb; j++) {
                unsigned long long v = 0;
                if (place) for (int k = 0; k < kInterpEdgeBlocks; k++) v += hc[j * kInterpEdgeBlocks + k];
                else v = hc[j];
                hcnt[b0 + j] = v;
            }
            if (last_batch) memcpy(hstat, hp, sizeof hstat);
        }
        return 0;
    };
    P.allow_wave2 = 1;
    const bool first_with_index = !interp_takes_wave3(P);
    BG_TRY(run_all(1, first_with_index));
    if (hstat[7] && !first_with_index && !hstat[0] && !hstat[6]) BG_TRY(run_all(1, true));   // a run of thousands of nulls next to a window start
    // An interval column out of order comes FIRST: a one-pass call derives its row count from the first and the last timestamp alone,
    // so on an unsorted column most trips also fail the "fits the counted range" test below - and the documented answer for such a
    // column is the decline (BOWGPU_ERR_TS_UNSORTED: the caller keeps the reference's own path), not an argument error.
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    // (a reused count whose column has changed since: interp_wave3_kernel stored nothing for the trips that did not fit and said
    // so; the kernels a redo would use do not check, so the error comes first)
    if (hstat[6]) return fail(BOWGPU_ERR_ARG, "Interpolate: the rows produced do not add up to the count - the interval column changed between "
                                              "bowgpu_rolling_interpolate_count and _fill (include/bowgpu.h: the contract between the two calls)");
    if (hstat[5] && (job.from_cache || onepass)) {
        // the redo runs a kernel that trusts pass 1 blindly (and needs one): make pass 1 this call's own before handing it over
        InterpJob fresh;
        BG_TRY(interp_prepare(c, cols, ncols, ts_col, interval, &o, &fresh, global_s0, edge, false));
        if (n + fresh.M != n_out) return fail(BOWGPU_ERR_ARG, "Interpolate: the interval column changed between bowgpu_rolling_interpolate_count and _fill");
        P.tile_local = reinterpret_cast<const int32_t *>(fresh.tile_local);
        P.super_before = reinterpret_cast<const int64_t *>(fresh.super_before);
        P.kq = fresh.kq; P.kq_empty = fresh.kq_empty; P.drop = fresh.drop; P.e0 = fresh.e0;
        P.s0 = fresh.plan.s0; P.W = fresh.plan.W;
    }
    // some trip has more runs of synthetic rows than interp_wave3_kernel lists, spans 2^31 or more, or holds a gap of millions of empty
    // windows: the workgroup kernel takes the call.  It builds exclusive windows only: an inclusive call is declined instead of being
    // answered with the exclusive layout.
    if (hstat[5] && o.inclusive)
        return fail(BOWGPU_ERR_UNSUPPORTED, "Interpolate on inclusive windows: a 512-row trip that spans 2^31 or more, or holds a gap of "
                                            "millions of empty windows, is outside the device path");
    if (hstat[5]) BG_TRY(run_all(0, true));
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    for (int i = 0; i < ninterps; i++) BG_TRY(devout_finish(c, &douts[i], n_out, cols[i].type, n_out - (int64_t)hcnt[i], false));
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// Rolling.Interpolate over an interval column WITH NULLS (ts_nulls.hip has the semantics and the kernels): the
// call is made on the kept rows - those inside some window's slice - compacted into device temporaries, plus a marker column that
// tells afterwards which output rows are copies of null-timestamp rows; those get their null timestamp and their values' own
// validity back; after an inclusive iteration the second copy of a row on a window start with null timestamps behind it becomes the
// synthetic start row of the window that begins behind it.  outs == nullptr: the row count only.  *applies = false: the interval
// column has no nulls.
// What bowgpu_rolling_interpolate_count builds for an interval column with nulls and the bowgpu_rolling_interpolate_fill that follows
// it needs again: the kept rows compacted (one 8-byte column and one bitmap per column of the Bow, the marker column, the flags) and the
// compacted call's descriptors.  Kept per thread between the two calls, keyed like the count -> fill prefix of the ordinary path
// (Ctx::InterpCache): DEVICE-resident columns, the same pointers / offsets / lengths / null counts / interpolators / options, nothing
// written to or freed from device memory through the library in between (device_write_epoch) - include/bowgpu.h states the contract
// for the ordinary path, this is the same one.  Dropped by the fill that uses it, by any other Interpolate of the thread over an interval
// column with nulls, by bowgpu_trim, bowgpu_set_device and at thread exit (round 5 ran the whole compaction twice: ADVICE r04).
namespace {
struct NullTsState {
    std::vector<bowgpu_col> key_cols;
    std::vector<bowgpu_interp> key_interps;
    int32_t ts_col = 0;
    int64_t interval = 0;
    bowgpu_options o = {0, 0, 0};
    uint64_t epoch = 0;
    DevBuf flags, marker, marker_bits, ts_bits;
    std::vector<DevBuf> cvals, cbits;
    CompactCols cc;
    std::vector<bowgpu_col> cols2;
    std::vector<bowgpu_interp> interps2;
    int64_t m = 0, m_out = 0;
};
void null_ts_state_free(void *p) { delete reinterpret_cast<NullTsState *>(p); }
void null_ts_cache_drop(Ctx *c) {
    if (c->null_ts_cache) { void *p = c->null_ts_cache; c->null_ts_cache = nullptr; null_ts_state_free(p); }
}
bool null_ts_cache_hit(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options &o,
                       const bowgpu_interp *interps, int32_t ninterps) {
    const NullTsState *st = reinterpret_cast<const NullTsState *>(c->null_ts_cache);
    if (!st || st->ts_col != ts_col || st->interval != interval || (int32_t)st->key_cols.size() != ncols || (int32_t)st->key_interps.size() != ninterps) return false;
    if (st->o.offset != o.offset || st->o.inclusive != o.inclusive || st->o.strict_order != o.strict_order) return false;
    if (st->epoch != device_write_epoch()) return false;
    for (int i = 0; i < ncols; i++) {
        const bowgpu_col &a = st->key_cols[i], &b = cols[i];
        if (b.residency != BOWGPU_DEVICE || a.values != b.values || a.validity != b.validity || a.offset != b.offset || a.length != b.length ||
            a.null_count != b.null_count || a.type != b.type) return false;
    }
    return memcmp(st->key_interps.data(), interps, sizeof(bowgpu_interp) * (size_t)ninterps) == 0;
}
}  // namespace

static int interp_null_ts(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                          const bowgpu_interp *interps, int32_t ninterps, int64_t *n_out, bowgpu_out *outs, bool *applies) {
    *applies = false;
    if (!cols || ncols <= 0 || ts_col < 0 || ts_col >= ncols) return 0;       // (the ordinary path reports these)
    const bowgpu_col *tsc = &cols[ts_col];
    if (!tsc->validity || tsc->null_count == 0 || tsc->length == 0) return 0;
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    Plan plan;
    BG_TRY(plan_make(nullptr, tsc, interval, o.offset, &plan));                // ctor errors first (rolling.go:69-112)
    BG_TRY(interp_validate(cols, ncols, ts_col, &o, interps, ninterps));
    Ctx *c;
    BG_TRY(ctx_get(&c));
    // the _fill right behind its _count: the compaction is there (NullTsState above)
    const bool reuse = outs && null_ts_cache_hit(c, cols, ncols, ts_col, interval, o, interps, ninterps);
    if (!reuse) null_ts_cache_drop(c);
    DevCol dts;
    if (!reuse) {
        BG_TRY(devcol_prepare(c, tsc, &dts, true, true));
        if (dts.null_count <= 0) return 0;
    }
    *applies = true;
    const int64_t n = tsc->length;
    if (ncols > kMaxCompactCols) return fail(BOWGPU_ERR_UNSUPPORTED, "Interpolate over an interval column with nulls: at most %d columns", kMaxCompactCols);
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has a different length", i);
    auto nothing = [&]() -> int {
        if (n_out) *n_out = 0;
        if (outs) for (int i = 0; i < ninterps; i++) { outs[i].length = 0; outs[i].null_count = 0; outs[i].type = cols[i].type; }
        return 0;
    };
    const int incl = o.inclusive ? 1 : 0;
    // (the state of this call: a fresh one, or the one the count left; a fresh one is kept for the fill only when the count succeeded on
    // DEVICE-resident columns)
    struct StateOwner { NullTsState *p = nullptr; bool keep = false; ~StateOwner() { if (p && !keep) delete p; } } owner;
    if (reuse) { owner.p = reinterpret_cast<NullTsState *>(c->null_ts_cache); c->null_ts_cache = nullptr; }   // (one use: this fill owns it now)
    else owner.p = new NullTsState();
    NullTsState &S = *owner.p;
    DevBuf &flags = S.flags, &marker = S.marker, &marker_bits = S.marker_bits, &ts_bits = S.ts_bits;
    std::vector<DevBuf> &cvals = S.cvals, &cbits = S.cbits;
    CompactCols &cc = S.cc;
    std::vector<bowgpu_col> &cols2 = S.cols2;
    std::vector<bowgpu_interp> &interps2 = S.interps2;
    int64_t m = S.m, m_out = S.m_out;
    if (!reuse) {
    int last_valid = 1;
    BG_TRY(fetch_valid(c, tsc, n - 1, &last_valid));
    if (!last_valid || plan.W == 0) return nothing();      // HasNext (rolling.go:162-173) is false from the start: no window, no rows
    // the rows that belong to a window, compacted
    DevBuf ixbuf, ts_eff, keep, plain_ts, dropped, counts, base, sums;
    NbrIndex ix;
    BG_TRY(ixbuf.alloc(nbr_index_bytes(n, dts.vbit0)));
    BG_TRY(nbr_index_build(c, dts.vbits, dts.vbit0, n, ixbuf.p, &ix));
    BG_TRY(ts_eff.alloc((size_t)n * 8 + 16));
    const int64_t nw = (n + 63) >> 6;
    BG_TRY(keep.alloc((size_t)nw * 8 + 8));
    BG_TRY(dropped.alloc(32));
    BG_HIP(hipMemsetAsync(dropped.p, 0, 32, c->stream));
    if (incl) BG_TRY(plain_ts.alloc((size_t)nw * 8 + 8));
    // (inclusive iteration: mode 2 - the rows on a window start with a null timestamp behind them stay, ts_nulls.hip; slot 1 of `dropped`
    // counts the two shapes of them this path cannot express)
    BG_TRY(launch_ts_nullfill(c, reinterpret_cast<const int64_t *>(dts.values), dts.vbits, dts.vbit0, n, ix, plan.s0, plan.interval, plan.magic, incl ? 2 : 0,
                              reinterpret_cast<int64_t *>(ts_eff.p), reinterpret_cast<uint64_t *>(keep.p), reinterpret_cast<uint64_t *>(plain_ts.p),
                              reinterpret_cast<unsigned long long *>(dropped.p)));
    if (incl) {
        unsigned long long h_out = 0;
        BG_HIP(hipMemcpyAsync(&h_out, reinterpret_cast<unsigned long long *>(dropped.p) + 1, 8, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (h_out)
            return fail(BOWGPU_ERR_TS_NULLS, "inclusive Interpolate over an interval column with nulls: %llu row(s) on a window start with null timestamps behind them are "
                        "followed by an equal timestamp or sit on -1: outside the device path", h_out);
    }
    BG_TRY(counts.alloc((size_t)nw * 4 + 16));
    BG_TRY(base.alloc((size_t)(nw + 1) * 8 + 16));
    BG_TRY(sums.alloc((size_t)((nw + 2047) / 2048 + 2) * 8));
    BG_TRY(launch_keep_counts(c, reinterpret_cast<const uint64_t *>(keep.p), nw, reinterpret_cast<int32_t *>(counts.p)));
    int64_t *d_total = reinterpret_cast<int64_t *>(dropped.p) + 2;
    BG_TRY(launch_exclusive_scan(c, reinterpret_cast<const int32_t *>(counts.p), nw, reinterpret_cast<int64_t *>(base.p), reinterpret_cast<int64_t *>(sums.p), d_total));
    BG_HIP(hipMemcpyAsync(&m, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (m <= 0) return nothing();
    const size_t mbm = (size_t)((m + 63) >> 6) * 8 + 8;
    std::vector<DevCol> dcs(ncols);
    cvals.resize(ncols); cbits.resize(ncols);
    memset(&cc, 0, sizeof cc);
    cc.ncols = ncols; cc.ts_col = ts_col;
    for (int i = 0; i < ncols; i++) {
        BG_TRY(cvals[i].alloc((size_t)m * 8 + 16));
        cc.out_values[i] = reinterpret_cast<uint64_t *>(cvals[i].p);
        if (i == ts_col) continue;
        BG_TRY(devcol_prepare(c, &cols[i], &dcs[i], true, true));
        cc.values[i] = reinterpret_cast<const uint64_t *>(dcs[i].values); cc.vbits[i] = dcs[i].vbits; cc.vbit0[i] = dcs[i].vbit0;
        BG_TRY(cbits[i].alloc(mbm));
        cc.lookup_bits[i] = reinterpret_cast<uint64_t *>(cbits[i].p);
    }
    BG_TRY(flags.alloc((size_t)m * 4 + 16));
    BG_TRY(marker.alloc((size_t)m * 8 + 16));
    BG_TRY(marker_bits.alloc(mbm));
    if (incl) { BG_TRY(ts_bits.alloc(mbm)); cc.ts_bits = reinterpret_cast<uint64_t *>(ts_bits.p); }
    BG_TRY(launch_compact_rows(c, reinterpret_cast<const uint64_t *>(keep.p), reinterpret_cast<const int64_t *>(base.p), n, reinterpret_cast<const int64_t *>(ts_eff.p),
                               dts.vbits, dts.vbit0, reinterpret_cast<const uint64_t *>(plain_ts.p), cc, reinterpret_cast<int64_t *>(marker.p),
                               reinterpret_cast<uint32_t *>(flags.p)));
    BG_TRY(launch_pack_flags(c, reinterpret_cast<const uint32_t *>(flags.p), m, cc, reinterpret_cast<uint64_t *>(marker_bits.p)));
    device_write_epoch_bump();      // (temporaries may sit where an earlier call's columns sat: no count -> fill reuse across this point)
    // the compacted call: the Bow's columns + the marker under interpolation.None
    cols2.assign(ncols + 1, bowgpu_col());
    interps2.assign(interps, interps + ninterps);
    for (int i = 0; i <= ncols; i++) {
        bowgpu_col &t = cols2[i];
        memset(&t, 0, sizeof t);
        t.offset = 0; t.length = m; t.residency = BOWGPU_DEVICE;
        if (i == ncols) { t.values = marker.p; t.validity = reinterpret_cast<const uint8_t *>(marker_bits.p); t.type = BOWGPU_INT64; t.null_count = -1; }
        else if (i == ts_col) { t.values = cvals[i].p; t.validity = nullptr; t.type = BOWGPU_INT64; t.null_count = 0; }
        else { t.values = cvals[i].p; t.validity = reinterpret_cast<const uint8_t *>(cbits[i].p); t.type = cols[i].type; t.null_count = -1; }
    }
    {
        bowgpu_interp mk;
        memset(&mk, 0, sizeof mk);
        mk.kind = BOWGPU_INTERP_NONE; mk.col = ncols;
        interps2.push_back(mk);
    }
    BG_TRY(interp_count_impl(cols2.data(), ncols + 1, ts_col, interval, &o, interps2.data(), ninterps + 1, &m_out, nullptr, nullptr));
    S.m = m; S.m_out = m_out;
    }   // (!reuse)
    if (n_out) *n_out = m_out;
    if (!outs) {
        // the count alone: what it built stays for the fill that follows (DEVICE-resident columns: the contract of the ordinary path) -
        // the compacted call's own count -> fill prefix (c->interp_cache) with it, its interval column lives in the kept state
        bool all_device = true;
        for (int i = 0; i < ncols; i++) all_device = all_device && cols[i].residency == BOWGPU_DEVICE;
        if (all_device && m_out > 0) {
            S.key_cols.assign(cols, cols + ncols); S.key_interps.assign(interps, interps + ninterps);
            S.ts_col = ts_col; S.interval = interval; S.o = o; S.epoch = device_write_epoch();
            BG_HIP(hipStreamSynchronize(c->stream));
            c->null_ts_cache = owner.p; c->null_ts_cache_free = null_ts_state_free;
            owner.keep = true;
        } else c->interp_cache.valid = false;
        return 0;
    }
    if (m_out == 0) return nothing();
    const size_t vb = (size_t)((m_out + 7) >> 3);
    std::vector<DevBuf> tvals(ncols + 1), tbits(ncols + 1);
    std::vector<bowgpu_out> touts(ncols + 1);
    for (int i = 0; i <= ncols; i++) {
        BG_TRY(tvals[i].alloc((size_t)m_out * 8 + 16));
        BG_TRY(tbits[i].alloc(((vb + 3) & ~(size_t)3) + 8));
        BG_HIP(hipMemsetAsync(tbits[i].p, 0, ((vb + 3) & ~(size_t)3) + 8, c->stream));
        bowgpu_out &t = touts[i];
        t.values = tvals[i].p; t.validity = reinterpret_cast<uint8_t *>(tbits[i].p); t.length = m_out; t.null_count = 0; t.type = 0; t.residency = BOWGPU_DEVICE;
        if (i < ncols) { cc.patch_values[i] = reinterpret_cast<uint64_t *>(tvals[i].p); cc.patch_valid[i] = reinterpret_cast<uint32_t *>(tbits[i].p); }
    }
    BG_TRY(interp_fill_impl(cols2.data(), ncols + 1, ts_col, interval, &o, interps2.data(), ninterps + 1, touts.data(), nullptr, nullptr));
    c->interp_cache.valid = false;
    if (touts[0].length != m_out) return fail(BOWGPU_ERR_ARG, "internal: Interpolate over an interval column with nulls produced %lld rows, counted %lld", (long long)touts[0].length, (long long)m_out);
    PatchInterps px;
    memset(&px, 0, sizeof px);
    px.m = m;
    std::vector<DevBuf> nbrbufs(ncols);
    if (incl) {
        for (int i = 0; i < ncols; i++) {
            px.both_bits[i] = i == ts_col ? reinterpret_cast<const uint64_t *>(ts_bits.p) : reinterpret_cast<const uint64_t *>(cbits[i].p);
            BG_TRY(nbrbufs[i].alloc(nbr_index_bytes(m, 0)));
            BG_TRY(nbr_index_build(c, reinterpret_cast<const uint32_t *>(px.both_bits[i]), 0, m, nbrbufs[i].p, &px.nbr[i]));
            px.type[i] = cols[i].type; px.kind[i] = interps[i].kind; px.const_value[i] = interps[i].const_value;
            px.has_prev[i] = interps[i].has_prev_row; px.prev_t_valid[i] = interps[i].prev_t_valid; px.prev_v_valid[i] = interps[i].prev_v_valid;
            px.prev_t[i] = interps[i].prev_t; px.prev_v[i] = interps[i].prev_v; px.prev_v_i64[i] = interps[i].prev_v_i64;
        }
    }
    BG_TRY(launch_interp_patch(c, reinterpret_cast<const int64_t *>(tvals[ncols].p), reinterpret_cast<const uint32_t *>(tbits[ncols].p), m_out,
                               reinterpret_cast<const uint32_t *>(flags.p), cc, px));
    BG_HIP(hipStreamSynchronize(c->stream));       // (nbrbufs are released at the end of this scope)
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint64_t *dcnt = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(dscr) + 1024);
    for (int i = 0; i < ncols; i++) {
        uint64_t hcnt = 0;
        BG_TRY(launch_popcount(c, reinterpret_cast<const uint32_t *>(tbits[i].p), 0, m_out, dcnt));
        BG_HIP(hipMemcpyAsync(&hcnt, dcnt, 8, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
        DevOut d;
        BG_TRY(devout_prepare(c, &outs[i], m_out, &d, i < 16 ? i : -1));
        BG_HIP(hipMemcpyAsync(d.values, tvals[i].p, (size_t)m_out * 8, hipMemcpyDeviceToDevice, c->stream));
        BG_HIP(hipMemcpyAsync(d.validity, tbits[i].p, vb, hipMemcpyDeviceToDevice, c->stream));
        BG_TRY(devout_finish(c, &d, m_out, cols[i].type, m_out - (int64_t)hcnt, true));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    device_write_epoch_bump();
    return 0;
}

// ... for Bows wider than the compaction's 16 columns (round 6; BOWGPU_ERR_UNSUPPORTED before): the value columns in groups of 15, each
// group with the interval column in front as a frame of its own.  A column's output depends on the interval column and on itself only
// (the interpolators look at one column each: interpolation.go:139-155), so the groups' outputs side by side are the call's; the row count
// comes from the first group, the interval column's output is written by every group (the same bytes).  Errors are the whole frame's:
// it is validated before anything runs.
static int interp_null_ts_any(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                              const bowgpu_interp *interps, int32_t ninterps, int64_t *n_out, bowgpu_out *outs, bool *applies) {
    *applies = false;
    if (!cols || ncols <= kMaxCompactCols || ts_col < 0 || ts_col >= ncols || !interps || ninterps != ncols)
        return interp_null_ts(cols, ncols, ts_col, interval, opts, interps, ninterps, n_out, outs, applies);
    const bowgpu_col *tsc = &cols[ts_col];
    if (!tsc->validity || tsc->null_count == 0 || tsc->length == 0) return 0;
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    Plan plan;
    BG_TRY(plan_make(nullptr, tsc, interval, o.offset, &plan));
    BG_TRY(interp_validate(cols, ncols, ts_col, &o, interps, ninterps));
    for (int i = 0; i < ncols; i++) if (interps[i].col != i) return interp_null_ts(cols, ncols, ts_col, interval, opts, interps, ninterps, n_out, outs, applies);
    std::vector<int> values;
    for (int i = 0; i < ncols; i++) if (i != ts_col) values.push_back(i);
    bool first = true;
    for (size_t at = 0; at < values.size(); at += kMaxCompactCols - 1) {
        const size_t m = std::min<size_t>(kMaxCompactCols - 1, values.size() - at);
        std::vector<bowgpu_col> sc(m + 1);
        std::vector<bowgpu_interp> si(m + 1);
        std::vector<bowgpu_out> so(m + 1);
        sc[0] = cols[ts_col]; si[0] = interps[ts_col]; si[0].col = 0;
        if (outs) so[0] = outs[ts_col];
        for (size_t k = 0; k < m; k++) {
            sc[k + 1] = cols[values[at + k]];
            si[k + 1] = interps[values[at + k]]; si[k + 1].col = (int32_t)k + 1;
            if (outs) so[k + 1] = outs[values[at + k]];
        }
        bool ap = false;
        int64_t cnt = 0;
        BG_TRY(interp_null_ts(sc.data(), (int32_t)m + 1, 0, interval, opts, si.data(), (int32_t)m + 1, (n_out && first) ? &cnt : nullptr, outs ? so.data() : nullptr, &ap));
        if (!ap) return 0;   // (the interval column turned out to have no nulls: the ordinary path takes the call)
        *applies = true;
        if (n_out && first) *n_out = cnt;
        if (outs) {
            outs[ts_col] = so[0];
            for (size_t k = 0; k < m; k++) outs[values[at + k]] = so[k + 1];
        }
        first = false;
        if (!outs) break;   // (the count: one group tells it)
    }
    return 0;
}

int bowgpu_rolling_interpolate_count(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                                     const bowgpu_options *opts, const bowgpu_interp *interps, int32_t ninterps, int64_t *n_out) {
    bool null_ts = false;
    if (n_out) BG_TRY(interp_null_ts_any(cols, ncols, ts_col, interval, opts, interps, ninterps, n_out, nullptr, &null_ts));
    if (null_ts) return 0;
    return interp_count_impl(cols, ncols, ts_col, interval, opts, interps, ninterps, n_out, nullptr, nullptr);
}

int bowgpu_rolling_interpolate_fill(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                                    const bowgpu_options *opts, const bowgpu_interp *interps, int32_t ninterps, bowgpu_out *outs) {
    bool null_ts = false;
    if (outs) BG_TRY(interp_null_ts_any(cols, ncols, ts_col, interval, opts, interps, ninterps, nullptr, outs, &null_ts));
    if (null_ts) return 0;
    return interp_fill_impl(cols, ncols, ts_col, interval, opts, interps, ninterps, outs, nullptr, nullptr);
}

// ---- row-range sharded Interpolate: a shard + what lies beyond its two ends
int bowgpu_shard_interpolate_count(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                                   int64_t global_s0, const bowgpu_interp *interps, int32_t ninterps, const bowgpu_interp_edge *edge,
                                   int64_t *n_out) {
    return interp_count_impl(cols, ncols, ts_col, interval, opts, interps, ninterps, n_out, &global_s0, edge);
}

int bowgpu_shard_interpolate_fill(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                                  int64_t global_s0, const bowgpu_interp *interps, int32_t ninterps, const bowgpu_interp_edge *edge,
                                  bowgpu_out *outs) {
    return interp_fill_impl(cols, ncols, ts_col, interval, opts, interps, ninterps, outs, &global_s0, edge);
}

// first / last valid point of every column of a shard: what its neighbours' Linear / StepPrevious interpolators may need
int bowgpu_shard_interp_points(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, bowgpu_interp_points *out) {
    if (!cols || !out || ncols <= 0 || ncols > kMaxCols) return fail(BOWGPU_ERR_ARG, "bad argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    memset(out, 0, sizeof *out);
    const int64_t n = cols[ts_col].length;
    out->nrows = n;
    if (n == 0) return 0;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    DevCol dts;
    BG_TRY(ts_contract(c, &cols[ts_col]));
    BG_TRY(ts_device(c, &cols[ts_col], &dts));
    void *pool;
    BG_TRY(ctx_pool(c, kPoolShard, 16384, &pool));
    int64_t *drows = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(pool) + 12288);  // [col][first, last]
    std::vector<DevCol> dcs(ncols);
    for (int i = 0; i < ncols; i++) {
        if (cols[i].type != BOWGPU_INT64 && cols[i].type != BOWGPU_FLOAT64) return fail(BOWGPU_ERR_UNSUPPORTED, "column type outside the device path");
        BG_TRY(devcol_prepare(c, &cols[i], &dcs[i], true, true));
        BG_TRY(launch_first_last_valid(c, dcs[i].vbits, dcs[i].vbit0, n, drows + 2 * i));
    }
    int64_t hrows[2 * kMaxCols];
    BG_HIP(hipMemcpyAsync(hrows, drows, sizeof(int64_t) * 2 * ncols, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    auto at = [&](const void *base, int64_t row, uint64_t *v) -> int {
        BG_HIP(hipMemcpyAsync(v, reinterpret_cast<const char *>(base) + 8 * row, 8, hipMemcpyDeviceToHost, c->stream));
        return 0;
    };
    uint64_t tf = 0, tl = 0;
    BG_TRY(at(dts.values, 0, &tf));
    BG_TRY(at(dts.values, n - 1, &tl));
    uint64_t tv[4 * kMaxCols] = {0};
    for (int i = 0; i < ncols; i++) {
        const int64_t fr = hrows[2 * i], lr = hrows[2 * i + 1];
        out->first_valid[i] = fr >= 0; out->last_valid[i] = lr >= 0;
        if (fr >= 0) { BG_TRY(at(dts.values, fr, &tv[4 * i])); BG_TRY(at(dcs[i].values, fr, &tv[4 * i + 1])); }
        if (lr >= 0) { BG_TRY(at(dts.values, lr, &tv[4 * i + 2])); BG_TRY(at(dcs[i].values, lr, &tv[4 * i + 3])); }
    }
    BG_HIP(hipStreamSynchronize(c->stream));
    out->first_ts = (int64_t)tf; out->last_ts = (int64_t)tl;
    for (int i = 0; i < ncols; i++) {
        auto as_f64 = [&](uint64_t b) { double d; if (cols[i].type == BOWGPU_FLOAT64) memcpy(&d, &b, 8); else d = (double)(int64_t)b; return d; };
        out->first_t[i] = (double)(int64_t)tv[4 * i]; out->first_v[i] = as_f64(tv[4 * i + 1]);
        out->last_t[i] = (double)(int64_t)tv[4 * i + 2]; out->last_v[i] = as_f64(tv[4 * i + 3]);
        out->last_v_i64[i] = (int64_t)tv[4 * i + 3];
    }
    return 0;
}

// ---------------------------------------------------------------------------- IsColSorted / FillLinear
static int col_order_flags(Ctx *c, const DevCol &dc, int32_t type, uint32_t *flags) {
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint32_t *dflags = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dscr) + 256);
    BG_TRY(launch_col_order(c, reinterpret_cast<const uint64_t *>(dc.values), dc.vbits, dc.vbit0, dc.length, type, dflags));
    BG_HIP(hipMemcpyAsync(flags, dflags, 4, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bowgpu_is_col_sorted(const bowgpu_col *col, int32_t *sorted) {
    // reference bowassertion.go:15-81: empty column or unsupported type => false
    if (!col || !sorted) return fail(BOWGPU_ERR_ARG, "null argument");
    *sorted = 0;
    if (col->type != BOWGPU_INT64 && col->type != BOWGPU_FLOAT64) return 0;
    if (col->length == 0) return 0;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    DevCol dc;
    BG_TRY(devcol_prepare(c, col, &dc, true, true));
    uint32_t f = 0;
    BG_TRY(col_order_flags(c, dc, col->type, &f));
    *sorted = (f & 4) && !((f & 1) && (f & 2));
    return 0;
}

static int fill_finish(Ctx *c, FillParams &P, int64_t n, const DevCol &dfill, DevOut *dout, int type);

static int fill_linear_impl(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col, bowgpu_out *out, int32_t *unchanged,
                            bool ref_checked);
int bowgpu_fill_linear(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col, bowgpu_out *out, int32_t *unchanged) {
    return fill_linear_impl(cols, ncols, ref_col, fill_col, out, unchanged, false);
}
int bowgpu_fill_linear_sorted(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col, bowgpu_out *out, int32_t *unchanged) {
    return fill_linear_impl(cols, ncols, ref_col, fill_col, out, unchanged, true);
}
static int fill_linear_impl(const bowgpu_col *cols, int32_t ncols, int32_t ref_col, int32_t fill_col, bowgpu_out *out, int32_t *unchanged,
                            bool ref_checked) {
    // reference bowfill.go:14-103
    if (!cols || !out || !unchanged) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ref_col < 0 || ref_col > ncols - 1) return fail(BOWGPU_ERR_BAD_COL, "refColIndex is out of range");
    if (fill_col < 0 || fill_col > ncols - 1) return fail(BOWGPU_ERR_BAD_COL, "toFillColIndex is out of range");
    if (ref_col == fill_col) return fail(BOWGPU_ERR_ARG, "refColIndex and toFillColIndex are equal");
    const int rt = cols[ref_col].type, ft = cols[fill_col].type;
    if (rt != BOWGPU_INT64 && rt != BOWGPU_FLOAT64) return fail(BOWGPU_ERR_TYPE, "refColIndex '%d' is of type '%s'", ref_col, type_name(rt));
    const int64_t n = cols[fill_col].length;
    if (cols[ref_col].length != n) return fail(BOWGPU_ERR_ARG, "columns differ in length");
    *unchanged = 0;
    if (ft != BOWGPU_INT64 && ft != BOWGPU_FLOAT64) {
        // the reference checks the fill column's type after the ref column's emptiness / order (bowfill.go:35-51)
        return fail(BOWGPU_ERR_UNSUPPORTED, "toFillColIndex '%d' is of unsupported type '%s'", fill_col, type_name(ft));
    }
    Ctx *c;
    BG_TRY(ctx_get(&c));
    DevCol dref, dfill;
    BG_TRY(devcol_prepare(c, &cols[ref_col], &dref, true, true));
    BG_TRY(devcol_prepare(c, &cols[fill_col], &dfill, true, true));
    DevOut dout;
    BG_TRY(devout_prepare(c, out, n, &dout, 0));
    // (ref_checked: the caller's own bowfill.go:35-42 - the ref column holds a value and IsColSorted - has run; the pass over the ref
    // column that establishes both, a quarter of the call at 1e8 rows, is not made a second time)
    uint32_t f = ref_checked ? 4u : 0u;
    if (n > 0 && !ref_checked) BG_TRY(col_order_flags(c, dref, rt, &f));
    const bool ref_empty = !(f & 4);                                  // IsColEmpty: bowassertion.go:84-86
    const bool ref_sorted = (f & 4) && !((f & 1) && (f & 2));
    if (!ref_empty && !ref_sorted) return fail(BOWGPU_ERR_NOT_SORTED, "refColIndex '%d' is empty or not sorted", ref_col);
    const bool nothing = ref_empty || dfill.null_count == 0;          // bowfill.go:35-37, :53-55 return the receiver
    *unchanged = nothing ? 1 : 0;
    FillParams P;
    memset(&P, 0, sizeof P);
    P.ref_values = reinterpret_cast<const uint64_t *>(dref.values); P.ref_vbits = dref.vbits; P.ref_vbit0 = dref.vbit0; P.ref_type = rt;
    P.fill_values = reinterpret_cast<const uint64_t *>(dfill.values); P.fill_vbits = dfill.vbits; P.fill_vbit0 = dfill.vbit0; P.fill_type = ft;
    P.method = kFillLinear;
    // (no special casing is needed for the two "return the receiver" cases: without nulls every row is copied,
    //  and with an all-null reference column no null row passes the valid1 test of bowfill.go:74)
    return fill_finish(c, P, n, dfill, &dout, ft);
}

// shared tail of the fill entry points: the kernel, the count of valid outputs, copy-back.
// The neighbour index of the fill column's bitmap (three small launches, ~25 us at 1e8 rows) is built only for a REPEAT: the kernel
// finds the valid row beyond a trip's ends by a bounded walk (2048 rows) and says so when a run of nulls is longer - the usual column
// never needs the index (Interpolate does the same since round 5).  A device-resident output bitmap that can take the kernel's whole
// 64-bit words - 8-byte aligned, capacity to the end of the word holding row n - 1 - is written in place: no copy behind the kernel.
static int fill_finish(Ctx *c, FillParams &P, int64_t n, const DevCol &dfill, DevOut *dout, int type) {
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint64_t *dcnt = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(dscr) + 8);
    P.n = n;
    P.out_values = reinterpret_cast<uint64_t *>(dout->values);
    // the kernel stores one whole 64-bit validity word per wavefront trip: ceil(n/64)*8 bytes, which the word-aligned working
    // copy of devout_prepare (((ceil(n/8)+3)&~3)+4 bytes) always holds
    const bowgpu_out *u = dout->user;
    const bool place = n > 0 && u->residency == BOWGPU_DEVICE && (reinterpret_cast<uintptr_t>(u->validity) & 7) == 0 &&
                       ((u->length + 7) >> 3) >= 8 * ((n + 63) >> 6);   // (u->length: still the capacity the caller handed in)
    P.out_valid_words = reinterpret_cast<uint32_t *>(place ? u->validity : dout->validity);
    P.valid_count = reinterpret_cast<unsigned long long *>(dcnt);
    P.far_flag = reinterpret_cast<uint32_t *>(dscr);
    memset(&P.nbr, 0, sizeof P.nbr);
    struct { uint32_t far, pad; uint64_t cnt; } back = {0, 0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        if (attempt == 1) {
            void *ix;
            BG_TRY(ctx_pool(c, kPoolInterp + 3, nbr_index_bytes(n, dfill.vbit0), &ix));
            BG_TRY(nbr_index_build(c, dfill.vbits, dfill.vbit0, n, ix, &P.nbr));
        }
        BG_HIP(hipMemsetAsync(dscr, 0, 16, c->stream));
        BG_HIP(hipEventRecord(c->ev0, c->stream));
        BG_TRY(fill_run(c, P));
        BG_HIP(hipEventRecord(c->ev1, c->stream));
        BG_HIP(hipMemcpyAsync(&back, dscr, 16, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (!back.far) break;
    }
    {   // (bowgpu_last_kernel_ms / _name: the fill kernel of this call)
        float ms = 0;
        BG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->last_kernel_ms = ms;
        c->last_kernel_name = "fill_kernel";
    }
    BG_TRY(devout_finish(c, dout, n, type, n - (int64_t)back.cnt, !place));
    if (!place) BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bowgpu_fill(const bowgpu_col *col, int32_t method, bowgpu_out *out, int32_t *unchanged) {
    // reference bowfill.go:105-160 (FillMean), :162-253 (FillNext / FillPrevious)
    if (!col || !out || !unchanged) return fail(BOWGPU_ERR_ARG, "null argument");
    if (method != BOWGPU_FILL_PREVIOUS && method != BOWGPU_FILL_NEXT && method != BOWGPU_FILL_MEAN) return fail(BOWGPU_ERR_ARG, "unknown fill method %d", method);
    const int ft = col->type;
    if (ft != BOWGPU_INT64 && ft != BOWGPU_FLOAT64)  // FillMean's own error (bowfill.go:119-122); Previous/Next on Boolean/String: declined
        return fail(BOWGPU_ERR_UNSUPPORTED, "column is of unsupported type '%s'", type_name(ft));
    const int64_t n = col->length;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    DevCol dfill;
    BG_TRY(devcol_prepare(c, col, &dfill, true, true));
    DevOut dout;
    BG_TRY(devout_prepare(c, out, n, &dout, 0));
    *unchanged = dfill.null_count == 0 ? 1 : 0;
    FillParams P;
    memset(&P, 0, sizeof P);
    P.fill_values = reinterpret_cast<const uint64_t *>(dfill.values); P.fill_vbits = dfill.vbits; P.fill_vbit0 = dfill.vbit0; P.fill_type = ft;
    P.method = method;
    return fill_finish(c, P, n, dfill, &dout, ft);
}

// ---------------------------------------------------------------------------- whole-frame Aggregate
int bowgpu_aggregate_whole(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs) {
    // reference rolling/aggregation/whole.go:12-93
    if (!cols || ncols <= 0) return fail(BOWGPU_ERR_ARG, "nil bow");
    if (naggs <= 0) return fail(BOWGPU_ERR_NO_AGG, "at least one column aggregation is required");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    if (!outs) return fail(BOWGPU_ERR_ARG, "no output columns");
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].col < 0 || aggs[i].col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "column aggregation %d: no column with index %d", i, aggs[i].col);
        if (aggs[i].kind < 0 || aggs[i].kind >= BOWGPU_AGG__COUNT) return fail(BOWGPU_ERR_ARG, "column aggregation %d: unknown kind", i);
        const int t = cols[aggs[i].col].type;
        if (t != BOWGPU_INT64 && t != BOWGPU_FLOAT64) return fail(BOWGPU_ERR_UNSUPPORTED, "column aggregation %d: column type outside the device path", i);
    }
    const int64_t n = cols[ts_col].length;
    auto out_type_of = [&](int i) {
        int t = kind_type(aggs[i].kind);
        // both InputDependent and IteratorDependent resolve to the INPUT column's type here (whole.go:44-46)
        if (t == BOWGPU_INPUT_DEPENDENT || t == BOWGPU_ITERATOR_DEPENDENT) t = cols[aggs[i].col].type;
        return t;
    };
    if (n == 0) {  // whole.go:49-50
        for (int i = 0; i < naggs; i++) { outs[i].length = 0; outs[i].null_count = 0; outs[i].type = out_type_of(i); }
        return 0;
    }
    if (cols[ts_col].type != BOWGPU_INT64 && cols[ts_col].type != BOWGPU_FLOAT64) return fail(BOWGPU_ERR_UNSUPPORTED, "interval column type outside the device path");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_TRY(ts_contract(c, &cols[ts_col]));  // (the reference tolerates null ts here; the device path declines them)
    if (cols[ts_col].type != BOWGPU_INT64) return fail(BOWGPU_ERR_UNSUPPORTED, "whole-frame aggregation on the device path needs an Int64 interval column");
    DevCol dts;
    BG_TRY(ts_device(c, &cols[ts_col], &dts));
    // FirstValue / LastValue: int64(float64(first / last valid ts)) (whole.go:54-71).  The product path reads them in its finish
    // launch; the second implementation (BOWGPU_ROUTE_FORCE_GENERAL: whole_partial_kernel and the launches of rounds 1 - 5) on the host
    const bool general = (route_mask() & BOWGPU_ROUTE_FORCE_GENERAL) != 0;
    bool any_mode = false;
    for (int i = 0; i < naggs; i++) any_mode |= aggs[i].kind == BOWGPU_AGG_MODE;
    int64_t tfirst = 0, tlast = 0;
    if (general) {
        int64_t *hp;   // (one small kernel that stores both into the registered block: api.cpp plan_make does the same)
        BG_TRY(ctx_pinned(c, 4096, reinterpret_cast<void **>(&hp)));
        hp += 256;
        BG_TRY(launch_fetch_two(c, reinterpret_cast<const int64_t *>(dts.values), 0, n - 1, hp));
        BG_HIP(hipStreamSynchronize(c->stream));
        tfirst = hp[0]; tlast = hp[1];
    }
    auto go_i64 = [](double x) -> int64_t { return (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) ? INT64_MIN : (int64_t)x; };
    const int64_t first_value = go_i64((double)tfirst), last_value = go_i64((double)tlast);
    // where the finish launches store what the host reads back: the context's registered block (bytes 8192.. : values, then validity bytes)
    char *hblock;
    BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&hblock)));
    uint64_t *host_values = reinterpret_cast<uint64_t *>(hblock + 8192);
    uint8_t *host_valid = reinterpret_cast<uint8_t *>(hblock + 8192 + 8 * 64);
    if (naggs > 64) return fail(BOWGPU_ERR_UNSUPPORTED, "whole-frame aggregation: at most 64 aggregators per call");

    int64_t nblocks = (n + 65535) / 65536;
    if (nblocks > 2048) nblocks = 2048;
    // (a multiple of the kernels' 512-row step: every workgroup's rows then start on a 16-byte boundary of the columns and on a byte of
    // the bitmap's rows; the workgroups this leaves without rows write the identity state)
    const int64_t chunk = ((n + nblocks - 1) / nblocks + 511) & ~(int64_t)511;
    // (one block of the context pool - no hipMalloc / hipFree per call: the partial states + the merged state of the column
    // (whole_run), one value and one validity byte per reducer)
    struct Part { void *p; } partials, onev, oneb;
    {
        const size_t pb = ((size_t)(nblocks + 1) * stats_size() + 255) & ~(size_t)255, vb = (8 * (size_t)naggs + 255) & ~(size_t)255;
        void *blk;
        BG_TRY(ctx_pool(c, kPoolWhole, pb + vb + 64 + (size_t)naggs, &blk));
        partials.p = blk; onev.p = reinterpret_cast<char *>(blk) + pb; oneb.p = reinterpret_cast<char *>(blk) + pb + vb;
    }
    std::vector<DevCol> dcols(ncols);
    std::vector<char> done(naggs, 0);
    // Mode (mode.go:8-32) over the one window [0, n): mode.hip, not the streaming partial states
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].kind != BOWGPU_AGG_MODE) continue;
        done[i] = 1;
        const int col = aggs[i].col;
        DevCol &dc = dcols[col];
        if (dc.values == nullptr) {
            if (col == ts_col) { dc.values = dts.values; dc.length = n; dc.type = BOWGPU_INT64; }
            else BG_TRY(devcol_prepare(c, &cols[col], &dc, true, true));
        }
        DevBuf fi;
        BG_TRY(fi.alloc(64));
        const int64_t bounds[2] = {0, n};
        BG_HIP(hipMemcpyAsync(fi.p, bounds, 16, hipMemcpyHostToDevice, c->stream));
        BG_HIP(hipMemsetAsync(reinterpret_cast<char *>(fi.p) + 16, 0, 8, c->stream));
        uint32_t *word = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(fi.p) + 16);
        int64_t n_mid = 0, n_long = 0;
        BG_TRY(launch_mode(c, reinterpret_cast<const int64_t *>(dts.values), reinterpret_cast<const int64_t *>(fi.p), n, 0, 1, 1, 0, 0, dc.values, dc.vbits,
                           dc.vbit0, cols[col].type == BOWGPU_INT64, &aggs[i], reinterpret_cast<uint64_t *>(onev.p) + i, word, &n_mid, &n_long));
        BG_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t *>(oneb.p) + i, word, 1, hipMemcpyDeviceToDevice, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < naggs; i++) {
        if (done[i]) continue;
        // one pass per distinct input column: the partial state holds what every reducer needs
        const int col = aggs[i].col;
        DevCol &dc = dcols[col];
        if (dc.values == nullptr) {
            if (col == ts_col) { dc.values = dts.values; dc.length = n; dc.type = BOWGPU_INT64; }
            else BG_TRY(devcol_prepare(c, &cols[col], &dc, true, true));
        }
        WholeParamsH P;
        memset(&P, 0, sizeof P);
        P.ts = reinterpret_cast<const int64_t *>(dts.values);
        P.values = reinterpret_cast<const uint64_t *>(dc.values);
        P.vbits = dc.vbits; P.vbit0 = dc.vbit0; P.n = n; P.type = cols[col].type;
        P.need_ts = 0;
        for (int j = i; j < naggs; j++)
            if (aggs[j].col == col && aggs[j].kind >= BOWGPU_AGG_INTEGRAL_STEP && aggs[j].kind <= BOWGPU_AGG_WAVG_LINEAR) P.need_ts = 1;
        P.partials = partials.p; P.chunk = chunk;
        if (!general) {
            // the value kernel + ONE launch that merges its partials, evaluates every reducer of the column and stores the results where
            // the host reads them (kMaxAggs reducers per launch)
            BG_TRY(whole_value_run(c, &P, nblocks));
            std::vector<int> mine;
            for (int j = i; j < naggs; j++) if (aggs[j].col == col && !done[j]) { mine.push_back(j); done[j] = 1; }
            for (size_t at = 0; at < mine.size(); at += kMaxAggs) {
                WholeFinishH fin;
                memset(&fin, 0, sizeof fin);
                fin.ts = reinterpret_cast<const int64_t *>(dts.values); fin.nrows = n;
                fin.n = (int32_t)std::min<size_t>(kMaxAggs, mine.size() - at);
                fin.host_values = host_values; fin.host_valid = host_valid;
                for (int q = 0; q < fin.n; q++) {
                    const int j = mine[at + q];
                    WholeFinalH &F = fin.f[q];
                    F.kind = aggs[j].kind; F.out_type = out_type_of(j); F.col_is_int = cols[col].type == BOWGPU_INT64;
                    F.n_factors = aggs[j].n_factors;
                    for (int k = 0; k < F.n_factors && k < BOWGPU_MAX_FACTORS; k++) F.factors[k] = aggs[j].factors[k];
                    fin.slot[q] = j;
                }
                BG_TRY(whole_finish_run(c, partials.p, nblocks, fin));
            }
            continue;
        }
        BG_TRY(whole_run(c, &P, nblocks));
        for (int j = i; j < naggs; j++) {
            if (aggs[j].col != col || done[j]) continue;
            done[j] = 1;
            WholeFinalH F;
            memset(&F, 0, sizeof F);
            F.kind = aggs[j].kind; F.out_type = out_type_of(j); F.col_is_int = cols[col].type == BOWGPU_INT64;
            F.n_factors = aggs[j].n_factors;
            for (int k = 0; k < F.n_factors && k < BOWGPU_MAX_FACTORS; k++) F.factors[k] = aggs[j].factors[k];
            F.out_value = reinterpret_cast<uint64_t *>(onev.p) + j;
            F.out_valid_byte = reinterpret_cast<uint8_t *>(oneb.p) + j;
            BG_TRY(whole_final_run(c, reinterpret_cast<const char *>(partials.p) + (size_t)nblocks * stats_size(), 1, n, first_value, last_value, &F));
        }
    }
    std::vector<uint64_t> hv(naggs);
    std::vector<uint8_t> hb(naggs);
    if (general || any_mode) {   // (Mode's outputs, and everything of the second implementation, leave through the device block)
        BG_HIP(hipMemcpyAsync(hv.data(), onev.p, 8 * (size_t)naggs, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipMemcpyAsync(hb.data(), oneb.p, (size_t)naggs, hipMemcpyDeviceToHost, c->stream));
    }
    BG_HIP(hipStreamSynchronize(c->stream));   // THE synchronisation of the call
    if (!general)
        for (int i = 0; i < naggs; i++)
            if (aggs[i].kind != BOWGPU_AGG_MODE) { hv[i] = host_values[i]; hb[i] = host_valid[i]; }
    bool any_device = false;
    std::vector<uint8_t> vbs(naggs);
    for (int i = 0; i < naggs; i++) {
        if (outs[i].length < 1 || !outs[i].values || !outs[i].validity) return fail(BOWGPU_ERR_ARG, "output column %d needs one slot", i);
        vbs[i] = hb[i] ? 1 : 0;
        if (outs[i].residency == BOWGPU_DEVICE) {   // (one synchronisation behind all of them; host-resident slots are plain stores)
            BG_HIP(hipMemcpyAsync(outs[i].values, &hv[i], 8, hipMemcpyHostToDevice, c->stream));
            BG_HIP(hipMemcpyAsync(outs[i].validity, &vbs[i], 1, hipMemcpyHostToDevice, c->stream));
            any_device = true;
        } else {
            memcpy(outs[i].values, &hv[i], 8);
            *outs[i].validity = vbs[i];
        }
        outs[i].length = 1;
        outs[i].null_count = vbs[i] ? 0 : 1;
        outs[i].type = out_type_of(i);
    }
    if (any_device) BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
