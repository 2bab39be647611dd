// multi.cpp — ONE Rolling.Aggregate call over SEVERAL devices, inside one process and behind the C ABI (SURVEY.md §8b
// `bowgpu_set_devices`, §8e; reference rolling/aggregation.go:123-145: the user makes one call).
//
// bowgpu_set_devices(ids, n) names the devices; from then on bowgpu_rolling_aggregate / _planned / _interpolate_aggregate cut the
// rows of a call into row ranges, one per listed device, and run the shard record protocol of this same library over them
// (bowgpu_shard_begin -> records -> bowgpu_shard_finish: the carry-in stitch in row order) on one persistent host thread per
// device.  The "exchange" of the protocol is a vector in host memory - one process holds every record - so there is no collective,
// no transport, no torch.  Each rank reduces into device temporaries and puts the windows it owns at their places in the CALLER's
// buffers: values by one copy per output, validity bits through a host-side bit placement (a rank's first window rarely starts on
// a byte of the frame's bitmap).  The result is bit-identical to the one-device call (the protocol's own guarantee, asserted by
// tests/test_gpu_multi.py against the oracle and against the one-device call).
//
// Who is served: host-resident columns and outputs (pageable: each rank stages ITS rows once, over ITS device's PCIe link; registered:
// each device reads its range in place) on any device list; device-resident buffers only when every listed device is the calling
// thread's device (how a one-GPU box exercises the path: the same id listed N times).  Everything the record protocol declines
// (Mode, more than 16 aggregators, interval columns with nulls, strict_order windows over three ranks) is served by the one-device
// path as before.  No CPU implementation of anything here: the ranks run the HIP kernels.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace bowgpu {

int current_device_of_thread();   // api.cpp: the device the calling thread's context is (or will be) on

namespace {

// ---------------------------------------------------------------- the workers: one persistent thread per listed device
struct Worker {
    int device = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, quit = false;
    bool device_set = false;
};

struct Fanout {
    std::mutex call_mu;            // one fanned-out call at a time (the devices are busy with it anyway)
    std::vector<Worker *> workers;
    std::vector<int> ids;          // what the workers were started for
    std::mutex done_mu;
    std::condition_variable done_cv;
    int pending = 0;
};

std::atomic<int64_t> g_calls_listed{0}, g_calls_served{0};   // bowgpu_fanout_counts
thread_local int g_last_ranks = 1;            // ranks that served the calling thread's last Rolling.Aggregate (bowgpu_last_call_ranks)
std::mutex g_cfg_mu;
std::vector<int> g_ids;                      // guarded by g_cfg_mu
int64_t g_min_rows = (int64_t)1 << 20;       // rows per rank below which a call is not cut further
Fanout *g_fan = nullptr;                     // never destroyed (a worker may sit in its condition variable at process exit)

// BOWGPU_DEVICES="0,1,2,3" / BOWGPU_FANOUT_MIN_ROWS=<rows>, read ONCE per process (like BOWGPU_ROUTE): the initial device list of a process
// that never calls bowgpu_set_devices - how an unmodified program (the C++ mirror's replay of the reference's test tables,
// tests/test_gpu_host_mirror.py) runs through the fan-out.  Nothing on the call path reads the environment.
std::once_flag g_env_once;
void env_defaults() {
    std::call_once(g_env_once, [] {
        const char *d = getenv("BOWGPU_DEVICES");
        const char *m = getenv("BOWGPU_FANOUT_MIN_ROWS");
        std::lock_guard<std::mutex> g(g_cfg_mu);
        if (d && *d && g_ids.empty()) {
            std::vector<int> ids;
            for (const char *q = d; *q;) {
                char *end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                if (v >= 0 && v < 1024) ids.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
                if (*end != ',' ) break;
            }
            if (ids.size() >= 2 && ids.size() <= 64) g_ids = ids;
        }
        if (m && *m) { const long long v = strtoll(m, nullptr, 10); if (v >= 1) g_min_rows = v; }
    });
}

void worker_main(Worker *w) {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->has_job || w->quit; });
            if (w->quit && !w->has_job) return;
            job = std::move(w->job);
            w->has_job = false;
        }
        job();
    }
}

void fan_stop_locked(Fanout *f) {   // f->call_mu held
    for (Worker *w : f->workers) {
        { std::lock_guard<std::mutex> g(w->mu); w->quit = true; }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();   // (the thread's context, block cache and stream are released by its thread-local destructors)
        delete w;
    }
    f->workers.clear();
    f->ids.clear();
}

void fan_start_locked(Fanout *f, const std::vector<int> &ids) {
    for (int id : ids) {
        Worker *w = new Worker();
        w->device = id;
        w->th = std::thread(worker_main, w);
        f->workers.push_back(w);
    }
    f->ids = ids;
}

// runs fn(rank) on worker `rank` for rank in [0, world) and waits for all of them
void fan_run(Fanout *f, int world, const std::function<void(int)> &fn) {
    { std::lock_guard<std::mutex> g(f->done_mu); f->pending = world; }
    for (int r = 0; r < world; r++) {
        Worker *w = f->workers[r];
        {
            std::lock_guard<std::mutex> g(w->mu);
            w->job = [f, r, &fn] {
                fn(r);
                std::lock_guard<std::mutex> g2(f->done_mu);
                if (--f->pending == 0) f->done_cv.notify_all();
            };
            w->has_job = true;
        }
        w->cv.notify_one();
    }
    std::unique_lock<std::mutex> lk(f->done_mu);
    f->done_cv.wait(lk, [&] { return f->pending == 0; });
}

// ---------------------------------------------------------------- bit placement on the host
// A rank's bitmap (slot k at bit k) goes to bits [d0, d0 + nbits) of the frame's bitmap.  Whole bytes in the middle are stored by the
// rank itself; the first and last byte it touches may be shared with its neighbours and are handed to the calling thread as (index,
// mask, value) - applied after every rank is done, so no two threads ever store to one byte.
struct EdgeByte { int64_t idx; uint8_t mask, val; };

inline uint64_t load_bits64(const uint8_t *src, int64_t bit) {   // 64 bits starting at source bit `bit` (src is padded: 16 readable bytes past the end)
    uint64_t lo, hi;
    memcpy(&lo, src + (bit >> 3), 8);
    memcpy(&hi, src + (bit >> 3) + 8, 8);
    const int r = (int)(bit & 7);
    return r ? (lo >> r) | (hi << (64 - r)) : lo;
}

int64_t place_bits(const uint8_t *src, int64_t nbits, uint8_t *dst, int64_t d0, std::vector<EdgeByte> *edges) {
    if (nbits <= 0) return 0;
    int64_t set = 0;
    const int64_t B0 = d0 >> 3, B1 = (d0 + nbits - 1) >> 3;
    auto byte_at = [&](int64_t B, uint8_t *mask) -> uint8_t {   // dest byte B: its bits that belong to this rank, and their values
        const int64_t lo = std::max<int64_t>(B * 8, d0), hi = std::min<int64_t>(B * 8 + 8, d0 + nbits);   // [lo, hi) global bits
        const uint64_t v = load_bits64(src, lo - d0);
        const int cnt = (int)(hi - lo), sh = (int)(lo - B * 8);
        const uint8_t m = (uint8_t)(((1u << cnt) - 1u) << sh);
        *mask = m;
        return (uint8_t)(((uint32_t)(v & ((1u << cnt) - 1u))) << sh);
    };
    uint8_t m, v = byte_at(B0, &m);
    edges->push_back({B0, m, v});
    set += __builtin_popcount(v);
    if (B1 > B0) {
        v = byte_at(B1, &m);
        edges->push_back({B1, m, v});
        set += __builtin_popcount(v);
    }
    int64_t B = B0 + 1;
    for (; B + 8 <= B1; B += 8) {   // eight whole destination bytes per step
        const uint64_t w = load_bits64(src, B * 8 - d0);
        memcpy(dst + B, &w, 8);
        set += __builtin_popcountll(w);
    }
    for (; B < B1; B++) {
        const uint8_t b = (uint8_t)load_bits64(src, B * 8 - d0);
        dst[B] = b;
        set += __builtin_popcount(b);
    }
    return set;
}

// ---------------------------------------------------------------- one rank of one call
struct Rank {
    // filled by the calling thread
    int64_t row0 = 0, nrows = 0;
    // the rank's columns as the record protocol sees them; pageable columns are staged ONCE (the record pass and the rank's pass both
    // read the staged copy: each row crosses the host link once)
    std::vector<bowgpu_col> cols;
    std::vector<DevBuf> staged_values, staged_bits;
    bool staged = false;
    // its outputs: device temporaries
    std::vector<DevBuf> out_values, out_bits;
    std::vector<bowgpu_out> outs;
    std::vector<std::vector<uint8_t>> host_bits;     // the rank's bitmaps on the host (padded)
    std::vector<std::vector<EdgeByte>> edges;         // per output
    std::vector<int64_t> valid;                       // per output: valid slots among the owned ones
    bowgpu_shard_record record;
    bowgpu_shard_decision decision;
    bowgpu_agg_info info;
    // Interpolate -> Aggregate: the rank's points, what lies beyond its two ends, its interpolated rows (device temporaries)
    bowgpu_interp_points points;
    bowgpu_interp_edge edge;
    std::vector<bowgpu_interp> interps;
    std::vector<DevBuf> mid_values, mid_bits;
    std::vector<bowgpu_col> raw_cols;   // the rank's rows as given (cols becomes the interpolated frame once it exists)
    bool interpolated = false;
    int rc = 0;
    std::string err;
    void fail_from_thread(int code) { rc = code; err = bowgpu_last_error(); }
    void release() {   // on the rank's own thread: the blocks go back to THAT thread's (device's) cache
        staged_values.clear(); staged_bits.clear(); out_values.clear(); out_bits.clear(); mid_values.clear(); mid_bits.clear();
        staged = false; interpolated = false;
    }
};

struct Call {
    const bowgpu_col *cols; int32_t ncols, ts_col;
    int64_t interval; bowgpu_options opts;
    const bowgpu_agg *aggs; int32_t naggs;
    bowgpu_out *outs;
    uint32_t route;
    int world;
    std::vector<uint8_t *> frame_bits;   // per output: where the frame's bitmap is assembled (the caller's buffer when it lies on the host)
    std::vector<Rank> ranks;
    std::vector<bowgpu_shard_record> records;
    const bowgpu_interp *interps = nullptr;   // Rolling.Interpolate(...).Aggregate(...): one interpolator per column; nullptr: Aggregate alone
    int32_t ninterps = 0;
    int64_t global_s0 = 0;
};

int rank_enter(Worker *w, const Call &call, Ctx **c) {
    if (!w->device_set) {
        BG_TRY(bowgpu_set_device(w->device));
        w->device_set = true;
    }
    BG_TRY(bowgpu_debug_set_route(call.route));
    return ctx_get(c);
}

// stage the rank's rows of every referenced pageable column into HBM (values from an 8-row boundary so that one offset serves values and bits)
int rank_stage(Ctx *c, const Call &call, Rank *rk) {
    if (rk->staged) return 0;
    std::vector<char> used(call.ncols, call.interps ? 1 : 0);   // (Interpolate rewrites every column: interpolation.go:139-155)
    used[call.ts_col] = 1;
    for (int a = 0; a < call.naggs; a++) used[call.aggs[a].col] = 1;
    rk->cols.resize(call.ncols);
    rk->staged_values.resize(call.ncols);
    rk->staged_bits.resize(call.ncols);
    for (int i = 0; i < call.ncols; i++) {
        bowgpu_col s = call.cols[i];
        const int64_t off = call.cols[i].offset + rk->row0;
        s.offset = off;
        s.length = rk->nrows;
        s.null_count = (call.cols[i].validity && call.cols[i].null_count != 0) ? -1 : 0;
        if (used[i] && call.cols[i].residency == BOWGPU_HOST && rk->nrows > 0) {
            const int64_t a = off & ~(int64_t)7, rows = off + rk->nrows - a;
            BG_TRY(rk->staged_values[i].alloc((size_t)rows * 8 + 16));
            BG_TRY(copy_h2d(c, rk->staged_values[i].p, reinterpret_cast<const char *>(call.cols[i].values) + 8 * a, (size_t)rows * 8));
            s.values = rk->staged_values[i].p;
            if (s.null_count != 0) {
                const size_t nb = (size_t)((rows + 7) >> 3);
                BG_TRY(rk->staged_bits[i].alloc(((nb + 3) & ~(size_t)3) + 8));
                BG_HIP(hipMemsetAsync(rk->staged_bits[i].p, 0, rk->staged_bits[i].bytes, c->stream));
                BG_TRY(copy_h2d(c, rk->staged_bits[i].p, call.cols[i].validity + (a >> 3), nb));
                s.validity = reinterpret_cast<const uint8_t *>(rk->staged_bits[i].p);
            } else {
                s.validity = nullptr;
            }
            s.offset = off - a;
            s.residency = BOWGPU_DEVICE;
        }
        rk->cols[i] = s;
    }
    BG_HIP(hipStreamSynchronize(c->stream));   // (the staging halves are free again; the copies are in HBM)
    rk->staged = true;
    return 0;
}

// Interpolate -> Aggregate, phase 1: the rank's first / last valid point per column (what its neighbours' Linear / StepPrevious need)
void rank_points(Worker *w, Call *call, int r) {
    Rank *rk = &call->ranks[r];
    Ctx *c;
    int rc = rank_enter(w, *call, &c);
    if (rc == 0) rc = rank_stage(c, *call, rk);
    if (rc == 0) rc = bowgpu_shard_interp_points(rk->cols.data(), call->ncols, call->ts_col, &rk->points);
    if (rc != 0) rk->fail_from_thread(rc);
}

// ... phase 2: the rank's interpolated rows into device temporaries (bowgpu_shard_interpolate_count + _fill: concatenated in rank order they
// are the unsharded Interpolate, bit for bit); from here on the rank's columns ARE that frame
int rank_interpolate_impl(Ctx *c, Call *call, int r) {
    Rank *rk = &call->ranks[r];
    const int nc = call->ncols;
    int64_t n_out = 0;
    BG_TRY(bowgpu_shard_interpolate_count(rk->cols.data(), nc, call->ts_col, call->interval, &call->opts, call->global_s0, rk->interps.data(),
                                          call->ninterps, &rk->edge, &n_out));
    const int64_t cap = (n_out + 31) & ~(int64_t)31;   // (whole bitmap words: the fill writes device bitmaps in place)
    rk->mid_values.resize(nc); rk->mid_bits.resize(nc);
    std::vector<bowgpu_out> mid(nc);
    for (int i = 0; i < nc; i++) {
        BG_TRY(rk->mid_values[i].alloc((size_t)cap * 8 + 16));
        BG_TRY(rk->mid_bits[i].alloc((size_t)(cap >> 3) + 16));
        memset(&mid[i], 0, sizeof mid[i]);
        mid[i].values = rk->mid_values[i].p; mid[i].validity = reinterpret_cast<uint8_t *>(rk->mid_bits[i].p);
        mid[i].length = cap; mid[i].residency = BOWGPU_DEVICE;
    }
    if (n_out > 0)
        BG_TRY(bowgpu_shard_interpolate_fill(rk->cols.data(), nc, call->ts_col, call->interval, &call->opts, call->global_s0, rk->interps.data(),
                                             call->ninterps, &rk->edge, mid.data()));
    rk->raw_cols = rk->cols;
    for (int i = 0; i < nc; i++) {
        bowgpu_col ic;
        memset(&ic, 0, sizeof ic);
        ic.values = mid[i].values; ic.validity = mid[i].validity; ic.offset = 0; ic.length = n_out;
        ic.null_count = n_out > 0 ? mid[i].null_count : 0;
        ic.type = call->cols[i].type; ic.residency = BOWGPU_DEVICE;
        rk->cols[i] = ic;
    }
    // the staged copies of the input rows have served
    BG_HIP(hipStreamSynchronize(c->stream));
    rk->staged_values.clear(); rk->staged_bits.clear();
    rk->interpolated = true;
    return 0;
}

void rank_interpolate(Worker *w, Call *call, int r) {
    Rank *rk = &call->ranks[r];
    Ctx *c = nullptr;
    int rc = rank_enter(w, *call, &c);
    if (rc == 0) rc = rank_interpolate_impl(c, call, r);
    if (rc != 0) rk->fail_from_thread(rc);
}

void rank_begin(Worker *w, Call *call, int r, const int64_t *global_s0) {
    Rank *rk = &call->ranks[r];
    Ctx *c;
    int rc = rank_enter(w, *call, &c);
    if (rc == 0 && !rk->interpolated) rc = rank_stage(c, *call, rk);
    if (rc == 0)
        rc = bowgpu_shard_begin(rk->cols.data(), call->ncols, call->ts_col, call->interval, &call->opts, call->aggs, call->naggs, global_s0,
                                &rk->record);
    if (rc != 0) rk->fail_from_thread(rc);
}

// the rank's pass + stitch into device temporaries, then its owned windows into the caller's buffers
int rank_finish_impl(Ctx *c, Call *call, int r) {
    Rank *rk = &call->ranks[r];
    const bowgpu_shard_decision &d = rk->decision;
    const int na = call->naggs;
    const int64_t cap = std::max<int64_t>(d.windows_local, 1);
    rk->out_values.resize(na); rk->out_bits.resize(na); rk->outs.resize(na);
    const size_t vb = (size_t)((cap + 7) >> 3);
    for (int i = 0; i < na; i++) {
        BG_TRY(rk->out_values[i].alloc((size_t)cap * 8));
        BG_TRY(rk->out_bits[i].alloc(((vb + 3) & ~(size_t)3) + 8));
        memset(&rk->outs[i], 0, sizeof(bowgpu_out));
        rk->outs[i].values = rk->out_values[i].p;
        rk->outs[i].validity = reinterpret_cast<uint8_t *>(rk->out_bits[i].p);
        rk->outs[i].length = cap;
        rk->outs[i].residency = BOWGPU_DEVICE;
    }
    const int rc = bowgpu_shard_finish(rk->cols.data(), call->ncols, call->ts_col, call->interval, &call->opts, call->aggs, na, rk->outs.data(),
                                       call->records.data(), call->world, r, &rk->decision, &rk->info);
    if (rc != 0) return rc;   // (BOWGPU_SHARD_RETRY included: the calling thread decides)
    const int64_t owned = d.windows_owned, slot0 = d.first_slot_window_id;
    rk->host_bits.assign(na, std::vector<uint8_t>());
    rk->edges.assign(na, std::vector<EdgeByte>());
    rk->valid.assign(na, 0);
    if (owned <= 0) return 0;
    BG_TRY(ctx_get(&c));
    for (int i = 0; i < na; i++) {
        bowgpu_out *u = &call->outs[i];
        char *dst = reinterpret_cast<char *>(u->values) + 8 * slot0;
        if (u->residency == BOWGPU_DEVICE) BG_HIP(hipMemcpyAsync(dst, rk->outs[i].values, (size_t)owned * 8, hipMemcpyDeviceToDevice, c->stream));
        else BG_TRY(copy_d2h(c, dst, rk->outs[i].values, (size_t)owned * 8, u->residency == BOWGPU_HOST_PINNED));
        rk->host_bits[i].assign(((size_t)(owned + 7) >> 3) + 24, 0);
        BG_TRY(copy_d2h(c, rk->host_bits[i].data(), rk->outs[i].validity, (size_t)((owned + 7) >> 3)));
    }
    BG_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < na; i++)
        rk->valid[i] = place_bits(rk->host_bits[i].data(), owned, call->frame_bits[i], slot0, &rk->edges[i]);
    return 0;
}

void rank_finish(Worker *w, Call *call, int r) {
    Rank *rk = &call->ranks[r];
    Ctx *c = nullptr;
    int rc = rank_enter(w, *call, &c);
    if (rc == 0) rc = rank_finish_impl(c, call, r);
    if (rc < 0) rk->fail_from_thread(rc); else rk->rc = rc;
    if (rc != BOWGPU_SHARD_RETRY) {
        if (c) (void)hipStreamSynchronize(c->stream);
        rk->release();
    }
}

void rank_cleanup(Worker *w, Call *call, int r) {
    Ctx *c;
    if (rank_enter(w, *call, &c) == 0) (void)hipStreamSynchronize(c->stream);
    call->ranks[r].release();
}

bool is_decline(int rc) { return rc == BOWGPU_ERR_UNSUPPORTED || rc == BOWGPU_ERR_TS_NULLS; }

}  // namespace

// One fanned-out call.  *done = false (and 0 returned): the call is not one for the fan-out - the caller goes on with the one-device
// path.  plan: the frame's (newIntervalRolling's s0 / numWindows), against which the ranks' decisions are checked.  interps != nullptr:
// Rolling.Interpolate(interps...).Aggregate(aggs...) - every rank interpolates its rows first (its neighbours' edge points reach it
// through the same host-memory exchange), then the ranks aggregate their interpolated rows like any other frame.
static int fan_call(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive, bool strict,
                    const bowgpu_interp *interps, int32_t ninterps, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                    bowgpu_agg_info *info, bool *done) {
    *done = false;
    g_last_ranks = 1;
    env_defaults();
    std::vector<int> ids;
    int64_t min_rows;
    {
        std::lock_guard<std::mutex> g(g_cfg_mu);
        ids = g_ids;
        min_rows = g_min_rows;
    }
    if (ids.size() >= 2) g_calls_listed.fetch_add(1, std::memory_order_relaxed);
    if (ids.size() < 2) return 0;
    const int64_t n = cols[ts_col].length;
    const int64_t W = plan.W;
    if (n <= 0 || W <= 0) return 0;
    int world = (int)std::min<int64_t>((int64_t)ids.size(), n / std::max<int64_t>(min_rows, 1));
    if (world < 2) return 0;
    if (naggs > BOWGPU_CARRY_MAX_AGGS) return 0;
    for (int i = 0; i < naggs; i++) if (aggs[i].kind == BOWGPU_AGG_MODE) return 0;
    if (cols[ts_col].validity && cols[ts_col].null_count != 0) return 0;   // (nulls in the interval column: the one-device path serves them)
    if (interps) {
        // the sharded Interpolate's own limit (include/bowgpu.h): at most 8 columns (bowgpu_interp_edge).  Rows below the first window start
        // (a negative first timestamp under Go's truncating division) make the interpolated frame start with its synthetic row at s0 and go
        // on BELOW it: what Aggregate then does with that frame is the one-device path's business
        if (ncols > 8 || ninterps != ncols || plan.first_ts < plan.s0) return 0;
        for (int i = 0; i < ncols; i++) if (interps[i].col != i) return 0;
    }
    // device-resident buffers belong to ONE device: only a list that names the calling thread's device throughout can share them
    bool any_device = false;
    for (int i = 0; i < ncols; i++) any_device |= cols[i].residency == BOWGPU_DEVICE;
    for (int i = 0; i < naggs; i++) {
        any_device |= outs[i].residency == BOWGPU_DEVICE;
        if (outs[i].length < W) return fail(BOWGPU_ERR_ARG, "output column has %lld slots, %lld needed", (long long)outs[i].length, (long long)W);
        if (!outs[i].values || !outs[i].validity) return fail(BOWGPU_ERR_ARG, "output column lacks a values or validity buffer");
    }
    if (any_device) {
        const int mine = current_device_of_thread();
        for (int id : ids) if (id != mine) return 0;
        // the ranks work on streams of their own: what the calling thread's stream (bowgpu_set_stream: possibly the application's) still
        // has in flight on these buffers is done first
        Ctx *c;
        BG_TRY(ctx_get(&c));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    Fanout *f;
    {
        std::lock_guard<std::mutex> g(g_cfg_mu);
        if (!g_fan) g_fan = new Fanout();
        f = g_fan;
    }
    std::lock_guard<std::mutex> call_lock(f->call_mu);
    if (f->ids != ids) { fan_stop_locked(f); fan_start_locked(f, ids); }

    Call call;
    call.cols = cols; call.ncols = ncols; call.ts_col = ts_col; call.interval = plan.interval;
    call.opts.offset = plan.offset; call.opts.inclusive = opt_inclusive ? 1 : 0; call.opts.strict_order = strict ? 1 : 0;
    call.aggs = aggs; call.naggs = naggs; call.outs = outs; call.route = route_mask(); call.world = world;
    call.interps = interps; call.ninterps = ninterps; call.global_s0 = plan.s0;
    call.ranks.resize(world);
    call.records.resize(world);
    // row ranges: multiples of 4096 rows (16-byte aligned value pointers, whole bitmap words) except the last one's end
    {
        const int64_t per = ((n / world) + 4095) & ~(int64_t)4095;
        int64_t at = 0;
        for (int r = 0; r < world; r++) {
            const int64_t end = r + 1 == world ? n : std::min<int64_t>(n, at + per);
            call.ranks[r].row0 = at;
            call.ranks[r].nrows = end - at;
            at = end;
        }
    }
    // where the frame's bitmaps are assembled
    std::vector<std::vector<uint8_t>> frame_tmp(naggs);
    const size_t frame_bytes = (size_t)((W + 7) >> 3);
    call.frame_bits.resize(naggs);
    for (int i = 0; i < naggs; i++) {
        if (outs[i].residency == BOWGPU_DEVICE) { frame_tmp[i].assign(frame_bytes, 0); call.frame_bits[i] = frame_tmp[i].data(); }
        else call.frame_bits[i] = outs[i].validity;
    }
    auto first_error = [&](bool *declined) -> int {
        for (int r = 0; r < world; r++)
            if (call.ranks[r].rc < 0) {
                if (is_decline(call.ranks[r].rc)) { *declined = true; return 0; }
                return fail(call.ranks[r].rc, "%s", call.ranks[r].err.c_str());
            }
        return 0;
    };
    auto cleanup = [&] { fan_run(f, world, [&](int r) { rank_cleanup(f->workers[r], &call, r); }); };

    if (interps) {
        fan_run(f, world, [&](int r) { call.ranks[r].rc = 0; rank_points(f->workers[r], &call, r); });
        bool declined = false;
        int rc = first_error(&declined);
        if (rc != 0 || declined) { cleanup(); return rc; }
        // what lies beyond each rank's two ends, from every rank's points (host arithmetic: the "exchange")
        for (int r = 0; r < world; r++) {
            Rank &rk = call.ranks[r];
            memset(&rk.edge, 0, sizeof rk.edge);
            rk.interps.assign(interps, interps + ninterps);
            int left_last = -1;
            for (int q = 0; q < r; q++) if (call.ranks[q].points.nrows > 0) left_last = q;
            if (left_last >= 0) { rk.edge.has_left = 1; rk.edge.left_last_ts = call.ranks[left_last].points.last_ts; }
            for (int i = 0; i < ncols; i++) {
                for (int q = r - 1; q >= 0; q--) {      // nearest valid point on the left: the reference's own PrevRow mechanism carries it (linear.go:14-18)
                    const bowgpu_interp_points &pq = call.ranks[q].points;
                    if (pq.nrows > 0 && pq.last_valid[i]) {
                        bowgpu_interp &ip = rk.interps[i];
                        ip.has_prev_row = 1; ip.prev_t_valid = 1; ip.prev_v_valid = 1;
                        ip.prev_t = pq.last_t[i]; ip.prev_v = pq.last_v[i]; ip.prev_v_i64 = pq.last_v_i64[i];
                        break;
                    }
                }
                for (int q = r + 1; q < world; q++) {   // nearest valid point on the right
                    const bowgpu_interp_points &pq = call.ranks[q].points;
                    if (pq.nrows > 0 && pq.first_valid[i]) {
                        rk.edge.next_valid[i] = 1; rk.edge.next_t[i] = pq.first_t[i]; rk.edge.next_v[i] = pq.first_v[i];
                        break;
                    }
                }
            }
        }
        fan_run(f, world, [&](int r) { rank_interpolate(f->workers[r], &call, r); });
        rc = first_error(&declined);
        if (rc != 0 || declined) { cleanup(); return rc; }
    }

    int64_t s0_known = 0;
    const int64_t *s0_ptr = nullptr;
    for (int attempt = 0; attempt < 2; attempt++) {
        fan_run(f, world, [&](int r) { call.ranks[r].rc = 0; rank_begin(f->workers[r], &call, r, s0_ptr); });
        bool declined = false;
        int rc = first_error(&declined);
        if (rc != 0 || declined) { cleanup(); return rc; }
        for (int r = 0; r < world; r++) call.records[r] = call.ranks[r].record;
        // every rank's decision from the same records (host arithmetic)
        for (int r = 0; r < world; r++) {
            rc = bowgpu_shard_plan(call.records.data(), world, r, call.interval, call.opts.offset, &call.ranks[r].decision);
            if (rc != 0) { cleanup(); return rc; }
        }
        const bowgpu_shard_decision &d0 = call.ranks[0].decision;
        if (!d0.retry_with_s0 && (d0.s0 != plan.s0 || d0.num_windows != W)) {
            cleanup();
            if (interps) return 0;   // (an interpolated frame whose windows are not the input's: the two calls on one device word whatever there is to say)
            return fail(BOWGPU_ERR_ARG, "the plan was not made for this interval column (its first / last timestamp differ)");
        }
        fan_run(f, world, [&](int r) { rank_finish(f->workers[r], &call, r); });
        rc = first_error(&declined);
        if (rc != 0 || declined) { cleanup(); return rc; }
        bool retry = false;
        for (int r = 0; r < world; r++) retry |= call.ranks[r].rc == BOWGPU_SHARD_RETRY;
        if (!retry) break;
        if (attempt == 1) { cleanup(); return fail(BOWGPU_ERR_ARG, "the shard protocol did not settle after the second exchange"); }
        s0_known = call.ranks[0].decision.s0;   // rows below the first window start split across ranks (rolling.go:96-99): once more, s0 known
        s0_ptr = &s0_known;
    }
    // the bytes ranks share, padding bits, counts
    int64_t owned_total = 0;
    for (int r = 0; r < world; r++) owned_total += std::max<int64_t>(call.ranks[r].decision.windows_owned, 0);
    if (owned_total != W) return fail(BOWGPU_ERR_ARG, "internal: the ranks own %lld of %lld windows", (long long)owned_total, (long long)W);
    for (int i = 0; i < naggs; i++) {
        uint8_t *fb = call.frame_bits[i];
        // (edge bytes: first clear every bit some rank owns, then set - a byte may hold the bits of several ranks)
        for (int r = 0; r < world; r++) for (const EdgeByte &e : call.ranks[r].edges[i]) fb[e.idx] &= (uint8_t)~e.mask;
        for (int r = 0; r < world; r++) for (const EdgeByte &e : call.ranks[r].edges[i]) fb[e.idx] |= e.val;
        if (W & 7) fb[frame_bytes - 1] &= (uint8_t)((1u << (W & 7)) - 1u);   // padding bits stay clear (bowbuffer.go:22-40 zero-initialises)
        int64_t valid = 0;
        for (int r = 0; r < world; r++) valid += call.ranks[r].valid[i];
        outs[i].length = W;
        outs[i].null_count = W - valid;
        for (int r = 0; r < world; r++)   // (the types were resolved by every rank alike)
            if (!call.ranks[r].outs.empty()) { outs[i].type = call.ranks[r].outs[i].type; break; }
    }
    bool any_dev_out = false;
    for (int i = 0; i < naggs; i++) any_dev_out |= outs[i].residency == BOWGPU_DEVICE;
    if (any_dev_out) {
        Ctx *c;
        BG_TRY(ctx_get(&c));
        for (int i = 0; i < naggs; i++)
            if (outs[i].residency == BOWGPU_DEVICE) BG_TRY(copy_h2d(c, outs[i].validity, frame_tmp[i].data(), frame_bytes));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    if (info) {
        const bowgpu_agg_info &i0 = call.ranks[0].info;
        info->s0 = plan.s0; info->num_windows = W; info->new_interval_col = i0.new_interval_col; info->inclusive = i0.inclusive;
        info->long_windows = 0; info->kernel_ms = 0;
        for (int r = 0; r < world; r++) {
            info->long_windows += call.ranks[r].info.long_windows;
            info->kernel_ms = std::max(info->kernel_ms, call.ranks[r].info.kernel_ms);
        }
    }
    g_last_ranks = world;
    g_calls_served.fetch_add(1, std::memory_order_relaxed);
    *done = true;
    return 0;
}

int multi_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive, bool strict,
                    const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, bowgpu_agg_info *info, bool *done) {
    return fan_call(cols, ncols, ts_col, plan, opt_inclusive, strict, nullptr, 0, aggs, naggs, outs, info, done);
}

int multi_interpolate_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive, bool strict,
                                const bowgpu_interp *interps, int32_t ninterps, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                                bowgpu_agg_info *info, bool *done) {
    return fan_call(cols, ncols, ts_col, plan, opt_inclusive, strict, interps, ninterps, aggs, naggs, outs, info, done);
}

}  // namespace bowgpu

using namespace bowgpu;

extern "C" {

int bowgpu_set_devices(const int *ids, int n) {
    if (n < 0 || (n > 0 && !ids)) return fail(BOWGPU_ERR_ARG, "bowgpu_set_devices: bad arguments");
    env_defaults();   // (so that a later first call does not overwrite what this call sets)
    if (n > 64) return fail(BOWGPU_ERR_ARG, "bowgpu_set_devices: at most 64 ranks");
    int count = 0;
    if (n > 0) {
        BG_TRY(bowgpu_device_count(&count));
        if (count <= 0) return fail(BOWGPU_ERR_NO_DEVICE, "no HIP device available; the bowgpu path has no CPU fallback");
        for (int i = 0; i < n; i++)
            if (ids[i] < 0 || ids[i] >= count) return fail(BOWGPU_ERR_NO_DEVICE, "device %d out of range (%d devices)", ids[i], count);
    }
    std::lock_guard<std::mutex> g(g_cfg_mu);
    g_ids.assign(ids, ids + n);
    return 0;
}

int bowgpu_get_devices(int *ids, int cap, int *n) {
    if (!n) return fail(BOWGPU_ERR_ARG, "null argument");
    env_defaults();
    std::lock_guard<std::mutex> g(g_cfg_mu);
    *n = (int)g_ids.size();
    for (int i = 0; ids && i < cap && i < (int)g_ids.size(); i++) ids[i] = g_ids[i];
    return 0;
}

int bowgpu_last_call_ranks(int *ranks) {
    if (!ranks) return fail(BOWGPU_ERR_ARG, "null argument");
    *ranks = g_last_ranks;
    return 0;
}

int bowgpu_fanout_counts(int64_t *calls, int64_t *served) {
    if (!calls || !served) return fail(BOWGPU_ERR_ARG, "null argument");
    *calls = g_calls_listed.load(std::memory_order_relaxed);
    *served = g_calls_served.load(std::memory_order_relaxed);
    return 0;
}

int bowgpu_set_fanout_min_rows(int64_t rows) {
    if (rows < 1) return fail(BOWGPU_ERR_ARG, "bowgpu_set_fanout_min_rows: at least one row per rank");
    env_defaults();
    std::lock_guard<std::mutex> g(g_cfg_mu);
    g_min_rows = rows;
    return 0;
}

}  // extern "C"
