// api.cpp — the C ABI of libbowgpu.so (include/bowgpu.h): argument validation mirroring the
// reference's drivers, residency handling, window planning, and kernel orchestration.
// No CPU implementation of any reducer lives here: everything that touches column data runs
// in the HIP kernels; without a GPU every entry point fails with BOWGPU_ERR_NO_DEVICE.

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace bowgpu {

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    if (e == hipErrorOutOfMemory) return BOWGPU_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return BOWGPU_ERR_NO_DEVICE;
    return BOWGPU_ERR_HIP;
}

// ---------------------------------------------------------------- test / A-B routing
// Which kernel serves a call is decided from the call's shape alone.  The tests (and A/B measurements) need to push one call
// through every kernel that can take it: bowgpu_debug_set_route() sets a PER-THREAD mask of BOWGPU_ROUTE_* bits.  Nothing on the
// call path reads the environment; BOWGPU_ROUTE=<mask> is read ONCE per process as the initial mask of every thread (so that a
// profiler run of an unmodified script can pick a kernel).
static uint32_t route_env_default() {
    static const uint32_t v = [] { const char *e = getenv("BOWGPU_ROUTE"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; }();
    return v;
}
static thread_local uint32_t g_route = route_env_default();
uint32_t route_mask() { return g_route; }

static std::atomic<uint64_t> g_write_epoch{1};
uint64_t device_write_epoch() { return g_write_epoch.load(std::memory_order_relaxed); }
void device_write_epoch_bump() { g_write_epoch.fetch_add(1, std::memory_order_relaxed); }

// ---------------------------------------------------------------- context
// All device state is per calling thread (cgo calls arrive on arbitrary OS threads).  A thread that exits gives its state
// back: stream, events, pools, pinned block and its cache of scratch blocks - unless the process itself is exiting (the HIP
// runtime may already be going down then; the sentinel below is constructed after the runtime's own statics, so it is
// destroyed before them).
static bool g_process_exiting = false;
namespace {
struct ExitSentinel { ~ExitSentinel() { g_process_exiting = true; } };
void ctx_release(Ctx *c) {
    if (!c->inited) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->null_ts_cache && c->null_ts_cache_free) { void *q = c->null_ts_cache; c->null_ts_cache = nullptr; c->null_ts_cache_free(q); }
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_params) (void)hipFree(c->d_params);
    for (int i = 0; i < Ctx::kPoolSlots; i++) if (c->pool[i]) (void)hipFree(c->pool[i]);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_bounce) { (void)hipHostFree(c->h_bounce); (void)hipEventDestroy(c->bounce_ev[0]); (void)hipEventDestroy(c->bounce_ev[1]); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    *c = Ctx();
}
struct CtxHolder {
    Ctx c;
    ~CtxHolder() { if (!g_process_exiting) ctx_release(&c); }
};
}  // namespace
static thread_local CtxHolder g_ctx_holder;
#define g_ctx (g_ctx_holder.c)

// BOWGPU_ABORT_TRACE=1 (diagnostic): the C stack of whoever calls abort() - the HIP runtime does so on a queue error, with no
// message when stderr is not a terminal - before the process dies
static void abort_trace(int) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "bowgpu: SIGABRT, C stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

// bowgpu_shard_pass_begin leaves a pass in flight on the thread's stream ("PendingPass" below): its kernels use the context's
// scratch, pools and pinned read-back block.  Its two owners (bowgpu_shard_pass_begin, bowgpu_shard_finish) mark themselves; EVERY
// other entry point that takes the context settles and drops the pass first, so nothing reuses that memory under running kernels
// (the header states the rule; this enforces it).  A thread that exits with a pass in flight: PendingHolder's destructor.
static thread_local bool g_pending_owner = false;
struct PendingOwnerScope { bool was; PendingOwnerScope() : was(g_pending_owner) { g_pending_owner = true; } ~PendingOwnerScope() { g_pending_owner = was; } };
static void pending_drop(Ctx *c);
static bool pending_exists();

int ctx_get(Ctx **out) {
    static ExitSentinel sentinel;   // (constructed on the first call of any thread)
    (void)sentinel;
    static const bool traced = [] {
        const char *t = getenv("BOWGPU_ABORT_TRACE");
        if (t && t[0] == '1') signal(SIGABRT, abort_trace);
        return true;
    }();
    (void)traced;
    Ctx *c = &g_ctx;
    if (!c->inited) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0)
            return fail(BOWGPU_ERR_NO_DEVICE, "no HIP device available (hipGetDeviceCount: %s); the bowgpu path has no CPU fallback",
                        e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        if (c->device >= n) return fail(BOWGPU_ERR_NO_DEVICE, "device %d out of range (%d devices)", c->device, n);
        BG_HIP(hipSetDevice(c->device));
        BG_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        BG_HIP(hipEventCreate(&c->ev0));
        BG_HIP(hipEventCreate(&c->ev1));
        if (!c->stream) c->stream = c->own_stream;
        c->inited = true;
    } else {
        BG_HIP(hipSetDevice(c->device));  // cgo calls may hop OS threads: no thread-affine HIP state assumed
    }
    if (!g_pending_owner && pending_exists()) pending_drop(c);
    *out = c;
    return 0;
}

int current_device_of_thread() { return g_ctx.device; }

void ctx_drop_null_ts_cache(Ctx *c) {
    if (c->null_ts_cache && c->null_ts_cache_free) { void *p = c->null_ts_cache; c->null_ts_cache = nullptr; c->null_ts_cache_free(p); }
}

int ctx_scratch(Ctx *c, size_t bytes, void **dptr) {
    if (c->d_scratch_bytes < bytes) {
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->d_scratch_bytes = 0;
        BG_HIP(hipMalloc(&c->d_scratch, bytes));
        c->d_scratch_bytes = bytes;
    }
    *dptr = c->d_scratch;
    return 0;
}

int ctx_pinned(Ctx *c, size_t bytes, void **hptr) {
    if (bytes < 16384) bytes = 16384;  // one block for every small user (status words, null counts, seeds): it never moves under them
    if (c->h_pinned_bytes < bytes) {
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        c->h_pinned = nullptr;
        c->h_pinned_bytes = 0;
        BG_HIP(hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
        c->h_pinned_bytes = bytes;
    }
    *hptr = c->h_pinned;
    return 0;
}

int ctx_params(Ctx *c, void **dptr) {
    if (!c->d_params) BG_HIP(hipMalloc(&c->d_params, 4096));
    *dptr = c->d_params;
    return 0;
}

int ctx_pool(Ctx *c, int slot, size_t bytes, void **dptr) {
    if (slot < 0 || slot >= Ctx::kPoolSlots) return fail(BOWGPU_ERR_ARG, "bad pool slot");
    c->pool_gen[slot]++;
    if (c->pool_bytes[slot] < bytes) {
        if (c->pool[slot]) (void)hipFree(c->pool[slot]);
        c->pool[slot] = nullptr;
        c->pool_bytes[slot] = 0;
        BG_HIP(hipMalloc(&c->pool[slot], bytes));
        c->pool_bytes[slot] = bytes;
    }
    *dptr = c->pool[slot];
    return 0;
}

// ---------------------------------------------------------------- per-thread cache of device scratch blocks
// Each thread keeps the blocks its calls freed (steady-state calls issue no hipMalloc / hipFree).  The caches are registered
// process-wide so that memory one thread hoards does not starve another: a failed allocation trims EVERY thread's cache before
// it gives up, bowgpu_trim() does the same on request, and a thread that exits frees its own.
namespace {
struct BufCache;
std::mutex g_caches_mu;
std::vector<BufCache *> g_caches;   // guarded by g_caches_mu
struct BufCache {
    struct E { void *p; size_t cap; };
    std::mutex mu;                  // the owner works under it; another thread takes it only to trim
    std::vector<E> v;
    size_t total = 0;
    BufCache() { std::lock_guard<std::mutex> g(g_caches_mu); g_caches.push_back(this); }
    size_t drop_all_locked() {
        size_t freed = 0;
        for (const E &e : v) { (void)hipFree(e.p); freed += e.cap; }
        v.clear();
        total = 0;
        return freed;
    }
    size_t drop_all() { std::lock_guard<std::mutex> g(mu); return drop_all_locked(); }
    ~BufCache() {
        { std::lock_guard<std::mutex> g(g_caches_mu); g_caches.erase(std::remove(g_caches.begin(), g_caches.end(), this), g_caches.end()); }
        if (!g_process_exiting) drop_all();   // (process exit: the runtime may already be gone; leave the blocks to it)
    }
};
thread_local BufCache g_bufs;
constexpr size_t kCacheEntries = 24;
constexpr size_t kCacheBytes = (size_t)24 << 30;   // per thread
size_t trim_all_threads() {
    std::lock_guard<std::mutex> g(g_caches_mu);
    size_t freed = 0;
    for (BufCache *c : g_caches) freed += c->drop_all();
    return freed;
}
}  // namespace

void devbuf_cache_drop() { g_bufs.drop_all(); }

int devbuf_acquire(size_t n, void **p, size_t *cap) {
    {
        std::lock_guard<std::mutex> g(g_bufs.mu);
        // smallest cached block that holds n without wasting more than half of itself (or 1 MiB)
        int best = -1;
        for (size_t i = 0; i < g_bufs.v.size(); i++) {
            const size_t c = g_bufs.v[i].cap;
            if (c >= n && (c <= 2 * n || c <= n + (1u << 20)) && (best < 0 || c < g_bufs.v[best].cap)) best = (int)i;
        }
        if (best >= 0) {
            *p = g_bufs.v[best].p;
            *cap = g_bufs.v[best].cap;
            g_bufs.total -= *cap;
            g_bufs.v.erase(g_bufs.v.begin() + best);
            return 0;
        }
    }
    hipError_t e = hipMalloc(p, n);
    if (e == hipErrorOutOfMemory) {  // give every thread's cached blocks back and try once more
        (void)hipGetLastError();
        if (trim_all_threads() > 0) e = hipMalloc(p, n);
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    *cap = n;
    return 0;
}

void devbuf_release(void *p, size_t cap) {
    std::lock_guard<std::mutex> g(g_bufs.mu);
    g_bufs.v.push_back({p, cap});
    g_bufs.total += cap;
    while (g_bufs.v.size() > kCacheEntries || g_bufs.total > kCacheBytes) {  // evict the largest
        size_t big = 0;
        for (size_t i = 1; i < g_bufs.v.size(); i++) if (g_bufs.v[i].cap > g_bufs.v[big].cap) big = i;
        (void)hipFree(g_bufs.v[big].p);
        g_bufs.total -= g_bufs.v[big].cap;
        g_bufs.v.erase(g_bufs.v.begin() + big);
    }
}

int DevBuf::alloc(size_t n) {
    if (p) { devbuf_release(p, cap); p = nullptr; cap = 0; }
    bytes = n;
    if (n == 0) return 0;
    return devbuf_acquire(n, &p, &cap);
}

// ---------------------------------------------------------------- magic division
MagicDiv magic_make(uint64_t d) {
    // Granlund & Montgomery, "Division by invariant integers using multiplication", fig. 4.1, N = 64
    MagicDiv r;
    int l = 0;
    while (l < 64 && ((unsigned __int128)1 << l) < d) l++;
    const unsigned __int128 two_l = (unsigned __int128)1 << l;
    const unsigned __int128 num = ((two_l - d) << 64);
    r.m = (uint64_t)(num / d) + 1;
    r.sh1 = l < 1 ? (uint32_t)l : 1u;
    r.sh2 = l > 1 ? (uint32_t)(l - 1) : 0u;
    return r;
}

// ---------------------------------------------------------------- columns on the device
static int fetch_i64(Ctx *c, const bowgpu_col *col, int64_t row, int64_t *out) {
    const int64_t *p = reinterpret_cast<const int64_t *>(col->values) + col->offset + row;
    if (col->residency == BOWGPU_DEVICE) {
        if (!c) BG_TRY(ctx_get(&c));
        BG_HIP(hipMemcpyAsync(out, p, 8, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    } else {
        *out = *p;
    }
    return 0;
}

int fetch_valid(Ctx *c, const bowgpu_col *col, int64_t row, int *valid) {
    if (!col->validity) { *valid = 1; return 0; }
    const int64_t bit = col->offset + row;
    uint8_t byte = 0;
    if (col->residency == BOWGPU_DEVICE) {
        if (!c) BG_TRY(ctx_get(&c));
        BG_HIP(hipMemcpyAsync(&byte, col->validity + (bit >> 3), 1, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    } else {
        byte = col->validity[bit >> 3];
    }
    *valid = (byte >> (bit & 7)) & 1;
    return 0;
}

int count_nulls_device(Ctx *c, DevCol *dc) {
    if (!dc->vbits || dc->length == 0) { dc->null_count = 0; return 0; }
    void *d;
    BG_TRY(ctx_scratch(c, 4096, &d));
    uint64_t *dcount = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(d) + 2048);
    BG_TRY(launch_popcount(c, dc->vbits, dc->vbit0, dc->length, dcount));
    uint64_t set = 0;
    BG_HIP(hipMemcpyAsync(&set, dcount, 8, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    dc->null_count = dc->length - (int64_t)set;
    return 0;
}

// BOWGPU_ROUTE_PINNED_STAGE (A/B switch): pinned input columns are staged through HBM by DMA like pageable ones instead of being read in place
static bool pinned_as_host() { return (route_mask() & BOWGPU_ROUTE_PINNED_STAGE) != 0; }

int devcol_prepare(Ctx *c, const bowgpu_col *col, DevCol *out, bool need_values, bool need_validity) {
    if (col->length < 0 || col->offset < 0) return fail(BOWGPU_ERR_ARG, "negative column length/offset");
    if (col->length > 0 && !col->values) return fail(BOWGPU_ERR_ARG, "column has no values buffer");
    out->length = col->length;
    out->type = col->type;
    out->null_count = col->null_count;
    const int64_t n = col->length;
    if (n == 0) { out->null_count = 0; return 0; }
    const bool has_bitmap = col->validity != nullptr && col->null_count != 0;
    if (col->residency == BOWGPU_DEVICE) {
        if (reinterpret_cast<uintptr_t>(col->values) & 7) return fail(BOWGPU_ERR_ARG, "values buffer must be 8-byte aligned");
        out->values = reinterpret_cast<const char *>(col->values) + 8 * col->offset;
        if (has_bitmap) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(col->validity);
            out->vbits = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
            out->vbit0 = (int64_t)(a & 3) * 8 + col->offset;
            out->vwords = (out->vbit0 + n + 31) >> 5;
        }
    } else if (col->residency == BOWGPU_HOST_PINNED && !pinned_as_host()) {
        // zero-copy: the kernels read the registered host buffers where they lie (coalesced 16-byte loads over PCIe)
        if (reinterpret_cast<uintptr_t>(col->values) & 7) return fail(BOWGPU_ERR_ARG, "values buffer must be 8-byte aligned");
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, const_cast<void *>(col->values), 0) != hipSuccess || !dv) {
            (void)hipGetLastError();
            return fail(BOWGPU_ERR_ARG, "BOWGPU_HOST_PINNED: the values buffer is not registered (bowgpu_host_register)");
        }
        out->values = reinterpret_cast<const char *>(dv) + 8 * col->offset;
        if (has_bitmap) {
            void *db = nullptr;
            if (hipHostGetDevicePointer(&db, const_cast<uint8_t *>(col->validity), 0) != hipSuccess || !db) {
                (void)hipGetLastError();
                return fail(BOWGPU_ERR_ARG, "BOWGPU_HOST_PINNED: the validity buffer is not registered (bowgpu_host_register)");
            }
            const uintptr_t a = reinterpret_cast<uintptr_t>(db);
            out->vbits = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
            out->vbit0 = (int64_t)(a & 3) * 8 + col->offset;
            out->vwords = (out->vbit0 + n + 31) >> 5;
        }
    } else if (col->residency == BOWGPU_HOST || col->residency == BOWGPU_HOST_PINNED) {
        if (need_values) {
            BG_TRY(out->own_values.alloc((size_t)n * 8 + 16));
            BG_TRY(copy_h2d(c, out->own_values.p, reinterpret_cast<const char *>(col->values) + 8 * col->offset, (size_t)n * 8,
                            col->residency == BOWGPU_HOST_PINNED));
            out->values = out->own_values.p;
        }
        if (has_bitmap && need_validity) {
            const int64_t b0 = col->offset >> 3, b1 = (col->offset + n + 7) >> 3;
            const size_t nb = (size_t)(b1 - b0);
            BG_TRY(out->own_validity.alloc(((nb + 3) & ~(size_t)3) + 8));
            BG_HIP(hipMemsetAsync(out->own_validity.p, 0, out->own_validity.bytes, c->stream));
            BG_TRY(copy_h2d(c, out->own_validity.p, col->validity + b0, nb, col->residency == BOWGPU_HOST_PINNED));
            out->vbits = reinterpret_cast<const uint32_t *>(out->own_validity.p);
            out->vbit0 = col->offset & 7;
            out->vwords = (out->vbit0 + n + 31) >> 5;
        }
    } else {
        return fail(BOWGPU_ERR_ARG, "unknown residency %d", col->residency);
    }
    if (has_bitmap && out->vbits && col->null_count < 0) BG_TRY(count_nulls_device(c, out));
    if (!has_bitmap) out->null_count = 0;
    if (out->null_count == 0) { out->vbits = nullptr; }
    return 0;
}

int devout_prepare(Ctx *c, bowgpu_out *out, int64_t slots, DevOut *d, int pool_slot) {
    d->user = out;
    d->capacity = slots;
    if (out->length < slots) return fail(BOWGPU_ERR_ARG, "output column has %lld slots, %lld needed", (long long)out->length, (long long)slots);
    if (slots == 0) return 0;
    if (!out->values || !out->validity) return fail(BOWGPU_ERR_ARG, "output column lacks a values or validity buffer");
    const size_t vb = (size_t)((slots + 7) >> 3);
    if (out->residency == BOWGPU_DEVICE) {
        if (reinterpret_cast<uintptr_t>(out->values) & 7) return fail(BOWGPU_ERR_ARG, "output values must be 8-byte aligned");
        d->values = out->values;
        // The kernels update validity as 32-bit words; a caller buffer of exactly ceil(W/8) bytes may end
        // mid-word, so always assemble in an aligned temporary and copy the exact byte count back.
        if (pool_slot >= 0) {
            void *pp;
            BG_TRY(ctx_pool(c, pool_slot, ((vb + 3) & ~(size_t)3) + 4, &pp));
            d->validity = reinterpret_cast<uint8_t *>(pp);
            d->pool_slot = pool_slot;
        } else {
            BG_TRY(d->own_validity.alloc(((vb + 3) & ~(size_t)3) + 4));
            d->validity = reinterpret_cast<uint8_t *>(d->own_validity.p);
        }
    } else {
        BG_TRY(d->own_values.alloc((size_t)slots * 8));
        BG_TRY(d->own_validity.alloc(((vb + 3) & ~(size_t)3) + 4));
        d->values = d->own_values.p;
        d->validity = reinterpret_cast<uint8_t *>(d->own_validity.p);
    }
    return 0;
}

// Device <-> host copies of PAGEABLE caller buffers go through the context's own pinned staging (two kBouncePiece halves: the
// DMA of one overlaps the CPU copy of the other) and never hand the caller's pointer to the HIP runtime.  The runtime would pin
// the caller's pages itself (a userptr mapping, kept in a cache), and that goes wrong in ways the library cannot see: it refuses
// a range that lies only partly inside a registered one (a small malloc'ed buffer sharing a page with a buffer somebody
// registered), and on this pool GPU writes through such mappings were seen to die with "write access to a read-only page" on
// fresh machines.  Small copies (the runtime stages those itself) and BOWGPU_RUNTIME_PINS=1 (A/B switch) keep the direct form.
constexpr size_t kBouncePiece = 4u << 20;
constexpr size_t kBounceDirect = 16384;
static bool runtime_pins() {
    static const bool on = [] { const char *e = getenv("BOWGPU_RUNTIME_PINS"); return e && e[0] == '1'; }();
    return on;
}
static int bounce_get(Ctx *c, char **half0, char **half1) {
    if (!c->h_bounce) {
        BG_HIP(hipHostMalloc(&c->h_bounce, 2 * kBouncePiece, hipHostMallocDefault));   // (its own block: ctx_pinned's must not move)
        BG_HIP(hipEventCreateWithFlags(&c->bounce_ev[0], hipEventDisableTiming));
        BG_HIP(hipEventCreateWithFlags(&c->bounce_ev[1], hipEventDisableTiming));
    }
    *half0 = reinterpret_cast<char *>(c->h_bounce);
    *half1 = *half0 + kBouncePiece;
    return 0;
}
// The CPU side of the staging: one core copies at ~12 GB/s, the link takes 55.  Pieces of a megabyte or more are split over a small
// process-wide pool of helper threads (they only ever call memcpy; created on first use, BOWGPU_COPY_THREADS = 0 .. 8, default 3;
// the pool object is never destroyed, so no thread can wake up into a torn-down condition variable at exit).
namespace {
struct CopyPool {
    struct Job { char *d; const char *s; size_t n; std::atomic<int> *left; };
    std::mutex m;
    std::condition_variable work, done;
    std::vector<Job> q;
    int nthreads = 0;
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m);
                work.wait(lk, [&] { return !q.empty(); });
                j = q.back();
                q.pop_back();
            }
            memcpy(j.d, j.s, j.n);
            if (j.left->fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(m);
                done.notify_all();
            }
        }
    }
};
CopyPool *copy_pool() {
    static CopyPool *pool = [] {
        CopyPool *p = new CopyPool();   // (leaked on purpose)
        const char *e = getenv("BOWGPU_COPY_THREADS");
        int n = e ? atoi(e) : 3;
        if (n < 0) n = 0;
        if (n > 8) n = 8;
        for (int i = 0; i < n; i++) {
            try { std::thread([p] { p->run(); }).detach(); p->nthreads++; } catch (...) { break; }
        }
        return p;
    }();
    return pool;
}
void staged_memcpy(void *dst, const void *src, size_t n) {
    constexpr size_t kMinPart = 256u << 10;
    CopyPool *p = n >= 4 * kMinPart ? copy_pool() : nullptr;
    if (!p || p->nthreads == 0) { memcpy(dst, src, n); return; }
    const int parts = p->nthreads + 1;
    const size_t part = ((n / parts) + 63) & ~(size_t)63;
    std::atomic<int> left(0);
    char *d = reinterpret_cast<char *>(dst);
    const char *s = reinterpret_cast<const char *>(src);
    size_t off = part;   // [0, part) is the caller's own share
    {
        std::lock_guard<std::mutex> lk(p->m);
        for (int i = 1; i < parts && off < n; i++, off += part) {
            const size_t len = off + part < n && i + 1 < parts ? part : n - off;
            left.fetch_add(1);
            p->q.push_back({d + off, s + off, len, &left});
            if (len == n - off) { off = n; break; }
        }
    }
    p->work.notify_all();
    memcpy(d, s, part < n ? part : n);
    std::unique_lock<std::mutex> lk(p->m);
    p->done.wait(lk, [&] { return left.load() == 0; });
}
}  // namespace

int copy_d2h(Ctx *c, void *dst, const void *src, size_t bytes, bool registered) {
    if (bytes == 0) return 0;
    if (bytes <= kBounceDirect || registered || runtime_pins()) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) return 0;
        (void)hipGetLastError();
        if (e != hipErrorInvalidValue) return hip_fail(e, "hipMemcpyAsync (device to host)");
    }
    char *h[2];
    BG_TRY(bounce_get(c, &h[0], &h[1]));
    const size_t pieces = (bytes + kBouncePiece - 1) / kBouncePiece;
    auto len = [&](size_t k) { return k + 1 < pieces ? kBouncePiece : bytes - k * kBouncePiece; };
    for (size_t k = 0; k <= pieces; k++) {
        if (k < pieces) {   // piece k on its way into half k & 1 ...  (behind an earlier upload out of that half: stream order would do, but
                            // the caller may have switched streams since - bowgpu_set_stream - so wait for the upload's event)
            if (c->bounce_busy[k & 1]) BG_HIP(hipEventSynchronize(c->bounce_ev[k & 1]));
            c->bounce_busy[k & 1] = false;
            BG_HIP(hipMemcpyAsync(h[k & 1], reinterpret_cast<const char *>(src) + k * kBouncePiece, len(k), hipMemcpyDeviceToHost, c->stream));
            BG_HIP(hipEventRecord(c->bounce_ev[k & 1], c->stream));
        }
        if (k > 0) {        // ... while piece k - 1 leaves the other half
            BG_HIP(hipEventSynchronize(c->bounce_ev[(k - 1) & 1]));
            staged_memcpy(reinterpret_cast<char *>(dst) + (k - 1) * kBouncePiece, h[(k - 1) & 1], len(k - 1));
        }
    }
    return 0;
}
int copy_h2d(Ctx *c, void *dst, const void *src, size_t bytes, bool registered) {
    if (bytes == 0) return 0;
    if (bytes <= kBounceDirect || registered || runtime_pins()) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) return 0;
        (void)hipGetLastError();
        if (e != hipErrorInvalidValue) return hip_fail(e, "hipMemcpyAsync (host to device)");
    }
    char *h[2];
    BG_TRY(bounce_get(c, &h[0], &h[1]));
    const size_t pieces = (bytes + kBouncePiece - 1) / kBouncePiece;
    for (size_t k = 0; k < pieces; k++) {
        const size_t m = k + 1 < pieces ? kBouncePiece : bytes - k * kBouncePiece;
        if (c->bounce_busy[k & 1]) BG_HIP(hipEventSynchronize(c->bounce_ev[k & 1]));   // the DMA that last read this half is done
        staged_memcpy(h[k & 1], reinterpret_cast<const char *>(src) + k * kBouncePiece, m);
        BG_HIP(hipMemcpyAsync(reinterpret_cast<char *>(dst) + k * kBouncePiece, h[k & 1], m, hipMemcpyHostToDevice, c->stream));
        BG_HIP(hipEventRecord(c->bounce_ev[k & 1], c->stream));
        c->bounce_busy[k & 1] = true;
    }
    return 0;
}

int devout_finish(Ctx *c, DevOut *d, int64_t slots, int32_t type, int64_t null_count, bool copy_bitmap) {
    bowgpu_out *out = d->user;
    out->length = slots;
    out->type = type;
    out->null_count = null_count;
    if (slots == 0) return 0;
    const size_t vb = (size_t)((slots + 7) >> 3);
    if (out->residency == BOWGPU_DEVICE) {
        if (copy_bitmap) BG_HIP(hipMemcpyAsync(out->validity, d->validity, vb, hipMemcpyDeviceToDevice, c->stream));
    } else {
        // (registered buffers take the DMA directly; pageable ones go through the staging halves)
        BG_TRY(copy_d2h(c, out->values, d->values, (size_t)slots * 8, out->residency == BOWGPU_HOST_PINNED));
        BG_TRY(copy_d2h(c, out->validity, d->validity, vb, out->residency == BOWGPU_HOST_PINNED));
    }
    return 0;
}

// ---------------------------------------------------------------- plan
static int enforce_interval_and_offset(int64_t interval, int64_t offset, int64_t *out) {
    // reference rolling/rolling.go:114-128
    if (interval <= 0) return fail(BOWGPU_ERR_INTERVAL, "strictly positive interval required");
    if (offset >= interval || offset <= -interval) offset = offset % interval;
    if (offset < 0) offset += interval;
    *out = offset;
    return 0;
}

// s0 of newIntervalRolling (rolling.go:95-99) from the first timestamp: Go's (first/interval)*interval + offset with
// truncating division and wrapping int64 arithmetic
static int64_t first_window_start(int64_t first, int64_t interval, int64_t offset_norm) {
    int64_t s0 = (int64_t)((uint64_t)((first / interval) * interval) + (uint64_t)offset_norm);
    if (s0 > first) s0 = (int64_t)((uint64_t)s0 - (uint64_t)interval);
    return s0;
}

int plan_make(Ctx *c, const bowgpu_col *ts, int64_t interval, int64_t raw_offset, Plan *p) {
    // reference rolling/rolling.go:69-112 and :143-154
    if (ts->type != BOWGPU_INT64)
        return fail(BOWGPU_ERR_TS_TYPE, "impossible to create a new intervalRolling on column of type %s",
                    ts->type == BOWGPU_FLOAT64 ? "float64" : ts->type == BOWGPU_BOOLEAN ? "bool" : ts->type == BOWGPU_STRING ? "utf8" : "undefined");
    BG_TRY(enforce_interval_and_offset(interval, raw_offset, &p->offset));
    p->interval = interval;
    p->magic = magic_make((uint64_t)interval);
    p->s0 = 0;
    p->W = 0;
    const int64_t n = ts->length;
    if (n == 0) return 0;
    int64_t first = 0, last = 0;
    int64_t row = n - 1;
    if (!ts->validity) {
        // common case (no validity buffer): both scalars with one round trip
        const int64_t *base = reinterpret_cast<const int64_t *>(ts->values) + ts->offset;
        if (ts->residency == BOWGPU_DEVICE) {
            if (!c) BG_TRY(ctx_get(&c));
            int64_t *hp;
            BG_TRY(ctx_pinned(c, 4096, reinterpret_cast<void **>(&hp)));
            hp += 256;  // bytes 2048.. of the pinned block
            BG_TRY(launch_fetch_two(c, base, 0, n - 1, hp));   // (the kernel stores into the registered block)
            BG_HIP(hipStreamSynchronize(c->stream));
            first = hp[0]; last = hp[1];
        } else {
            first = base[0]; last = base[n - 1];
        }
    } else {
        int valid = 1;
        BG_TRY(fetch_valid(c, ts, 0, &valid));
        if (!valid) return fail(BOWGPU_ERR_FIRST_TS_NULL, "the first value of the column should be convertible to int64, got <nil>");
        BG_TRY(fetch_i64(c, ts, 0, &first));
    }
    p->first_ts = first;
    const int64_t s0 = first_window_start(first, interval, p->offset);
    p->s0 = s0;
    if (ts->validity) {
        // countWindows: last VALID ts scanning backwards (GetPrevInt64, bowgetters.go:189-199)
        int lv = 1;
        while (row >= 0) {
            BG_TRY(fetch_valid(c, ts, row, &lv));
            if (lv) break;
            row--;
        }
        if (row < 0) { p->W = 0; return 0; }
        BG_TRY(fetch_i64(c, ts, row, &last));
    }
    p->last_ts = last;
    if (s0 > last) { p->W = 0; return 0; }
    p->W = (int64_t)(((uint64_t)last - (uint64_t)s0) / (uint64_t)interval) + 1;
    // the reference computes (last - s0)/interval in int64: decline ranges where that wraps
    if ((last > 0 && s0 < 0 && (uint64_t)last - (uint64_t)s0 > (uint64_t)INT64_MAX))
        return fail(BOWGPU_ERR_UNSUPPORTED, "interval column spans more than 2^63: int64 overflow in the reference's countWindows");
    return 0;
}

// ---------------------------------------------------------------- aggregate
bool kind_needs_inclusive(int kind) {
    // NewColAggregation(col, true, ...) only at integral.go:9 and weightedmean.go:24
    return kind == BOWGPU_AGG_INTEGRAL_TRAPEZOID || kind == BOWGPU_AGG_WAVG_LINEAR;
}

int kind_type(int kind) {
    switch (kind) {
    case BOWGPU_AGG_WINDOW_START: return BOWGPU_ITERATOR_DEPENDENT;  // windowstart.go:9
    case BOWGPU_AGG_COUNT: return BOWGPU_INT64;                      // count.go:9
    case BOWGPU_AGG_FIRST:
    case BOWGPU_AGG_LAST: return BOWGPU_INPUT_DEPENDENT;             // firstlast.go:9,:24
    case BOWGPU_AGG_MODE: return BOWGPU_INPUT_DEPENDENT;             // mode.go:9
    default: return BOWGPU_FLOAT64;
    }
}

bool kind_never_nil(int kind) {
    return kind == BOWGPU_AGG_WINDOW_START || kind == BOWGPU_AGG_SUM || kind == BOWGPU_AGG_COUNT || kind == BOWGPU_AGG_NUM_ROWS;
}

bool kind_reads_values(int kind) { return !(kind == BOWGPU_AGG_WINDOW_START || kind == BOWGPU_AGG_NUM_ROWS); }

struct AggRun {
    Plan plan;
    std::vector<DevCol> dcols;   // indexed by user column
    std::vector<DevOut> douts;
    AggParams params;
    int inclusive = 0;
    int new_interval_col = -1;
};

// validation shared by the single-GPU and the sharded entry points
// (indexedAggregations + validateAggregation: reference rolling/aggregation.go:147-188)
static int validate_aggs(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_agg *aggs, int32_t naggs,
                         int *inclusive, int *new_interval_col) {
    if (naggs <= 0) return fail(BOWGPU_ERR_NO_AGG, "at least one column aggregation is required");
    int nic = -1;
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].col < 0 || aggs[i].col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "aggregation %d: no column with index %d", i, aggs[i].col);
        if (aggs[i].kind < 0 || aggs[i].kind >= BOWGPU_AGG__COUNT) return fail(BOWGPU_ERR_ARG, "aggregation %d: unknown kind %d", i, aggs[i].kind);
        if (aggs[i].n_factors < 0 || aggs[i].n_factors > BOWGPU_MAX_FACTORS) return fail(BOWGPU_ERR_ARG, "aggregation %d: bad factor count", i);
        const int t = cols[aggs[i].col].type;
        if (t != BOWGPU_FLOAT64 && t != BOWGPU_INT64)
            return fail(BOWGPU_ERR_UNSUPPORTED, "aggregation %d: column type %d is outside the device path (Float64/Int64 only)", i, t);
        if (kind_needs_inclusive(aggs[i].kind)) *inclusive = 1;  // aggregation.go:183-185
        if (aggs[i].col == ts_col) nic = i;                      // aggregation.go:158-160 (last one wins)
    }
    if (nic == -1) return fail(BOWGPU_ERR_KEEP_INTERVAL, "must keep interval column");
    *new_interval_col = nic;
    return 0;
}

// One aggregate call in three stages so that the sharded entry points can reuse them:
//   job_build  - device residency of columns / outputs + the kernels' descriptor block
//   job_run    - bitmap initialisation, tile kernel, long-window kernel
//   job_finish - null counts of the nullable outputs, copy-back of host-resident outputs
struct AggJob {
    std::vector<DevCol> dcols;
    std::vector<DevOut> douts;
    DevCol dts;
    AggParams P;
    int64_t W = 0;
    size_t scratch_bytes = 0;
    int inclusive = 0;
    bool counts_used = false;   // the valid counters hold a previous count (they accumulate): zero them before counting again
    bool tail_wrote_host = false;   // the finish launch stores the status words and the counts into the registered host block itself
    bool check_plan = false;    // the plan came from the caller (bowgpu_rolling_aggregate_planned): the pass checks it against the column
    bool band_rows = false;     // job_run: a nullable column under extrema / First + Last alone, 129 .. 200 rows per window - rolling_simple.hip + the queue launch, not rolling_twc.hip
};
static thread_local bool g_plan_from_caller = false;   // set around run_aggregate by the planned entry point
// Up to which window length (rows on average) a call on a NULLABLE column stays on rolling_twc_kernel with its 256 rows of look-ahead
// instead of taking the streaming form (0: it does not) - from the 1e8-row sweeps of profiles/r05_stdout_midw_sweep.txt, 30 % nulls, kernel
// ms compacting / streaming at 144 and 192 rows per window: one kind of integral 0.45 / 0.70 and 0.47 / 0.64, First + Last 0.38 / 0.46 and
// 0.37 / 0.43 (both up to 255 rows); Min + Max 0.47 / 0.61 and 0.49 / 0.52, both kinds of integral 0.63 / 0.73 and 0.69 / 0.65 (up to 176);
// sums and counts alone: the streaming form (0.41 / 0.37).  Columns without nulls: the streaming form throughout (it has its dense
// instantiations; the compacting kernel is 10 - 40 % behind there).
static int64_t compact_long_max_rows(const bowgpu_agg *aggs, int32_t naggs) {
    bool step = false, trap = false, mm = false, fl = false;
    for (int i = 0; i < naggs; i++) {
        const int k = aggs[i].kind;
        step |= k == BOWGPU_AGG_INTEGRAL_STEP || k == BOWGPU_AGG_WAVG_STEP;
        trap |= k == BOWGPU_AGG_INTEGRAL_TRAPEZOID || k == BOWGPU_AGG_WAVG_LINEAR;
        mm |= k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX;
        fl |= k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST;
    }
    if ((step && trap) || mm) return kCompactLongBothMaxAvgRows;
    if (step || trap || fl) return kCompactLongMaxAvgRows;
    return 0;
}
static thread_local bool g_strict_order = false;   // bowgpu_options.strict_order of the call in progress (or BOWGPU_ROUTE_STRICT_ORDER)

static void pending_drop(Ctx *c);   // a pass put in flight by bowgpu_shard_pass_begin and not collected: settled before the scratch is reused

static int job_build(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int inclusive,
                     const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, int64_t wid_base, int64_t W,
                     bool holds_row0, AggJob *job, int nullable_per_pass = 4) {
    pending_drop(c);
    const bowgpu_col *tsc = &cols[ts_col];
    const int64_t n = tsc->length;
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has %lld rows, interval column has %lld", i, (long long)cols[i].length, (long long)n);

    // device path contract: no null timestamps
    if (tsc->validity && tsc->null_count != 0) {
        DevCol probe;
        BG_TRY(devcol_prepare(c, tsc, &probe, false, true));
        if (probe.null_count > 0)
            return fail(BOWGPU_ERR_TS_NULLS, "interval column has %lld nulls: outside the device path", (long long)probe.null_count);
    }

    job->dcols.clear();
    job->dcols.resize(ncols);
    job->douts.clear();
    job->douts.resize(naggs);
    job->W = W;
    job->inclusive = inclusive;
    job->check_plan = g_plan_from_caller;
    std::vector<DevCol> &dcols = job->dcols;
    std::vector<int> slot_of(ncols, -1);
    AggParams &P = job->P;
    memset(&P, 0, sizeof P);
    {
        bowgpu_col t = *tsc;
        t.validity = nullptr;
        t.null_count = 0;
        BG_TRY(devcol_prepare(c, &t, &job->dts, true, false));
    }
    P.ts = reinterpret_cast<const int64_t *>(job->dts.values);
    P.n = n;
    P.row_base = 0;
    P.s0 = plan.s0;
    P.interval = plan.interval;
    P.W = W;
    P.wid_base = wid_base;
    P.magic = plan.magic;
    P.fits32 = ((uint64_t)plan.interval >> 32) == 0;
    if (P.fits32) {
        // Granlund & Montgomery fig. 4.1 with N = 32
        const uint64_t d = (uint64_t)plan.interval;
        int l = 0;
        while (l < 32 && (1ull << l) < d) l++;
        P.m32 = (uint32_t)((((1ull << l) - d) << 32) / d) + 1;
        P.sh1_32 = l < 1 ? (uint32_t)l : 1u;
        P.sh2_32 = l > 1 ? (uint32_t)(l - 1) : 0u;
    }
    P.inclusive = inclusive;
    P.naggs = naggs;
    P.pre_rows = (n > 0 && holds_row0 && plan.s0 > plan.first_ts) ? 1 : 0;

    // column slots: each distinct input column whose values some reducer reads
    std::vector<int> nullable_in_slot;
    for (int i = 0; i < naggs; i++) {
        if (!kind_reads_values(aggs[i].kind)) continue;
        const int col = aggs[i].col;
        int s = slot_of[col];
        const bool nullable = !kind_never_nil(aggs[i].kind);
        if (s >= 0 && nullable && nullable_in_slot[s] >= nullable_per_pass) s = -1;  // at most 4 nullable reducers per pass of the general kernel: open another slot (rolling_fused.hip has no such limit)
        if (s < 0) {
            if (P.ncols >= kMaxCols) return fail(BOWGPU_ERR_UNSUPPORTED, "at most %d value columns per call", kMaxCols);
            s = P.ncols++;
            slot_of[col] = s;
            nullable_in_slot.push_back(0);
            DevCol &dc = dcols[col];
            if (dc.values == nullptr && n > 0) {
                if (col == ts_col) {
                    dc.values = job->dts.values; dc.length = n; dc.type = BOWGPU_INT64; dc.null_count = 0;
                } else {
                    BG_TRY(devcol_prepare(c, &cols[col], &dc, true, true));
                }
            }
            ColDesc &cd = P.cols[s];
            cd.values = dc.values;
            cd.vbits = dc.vbits;
            cd.vbit0 = dc.vbit0;
            cd.vwords = dc.vwords;
            cd.type = cols[col].type;
        }
        if (nullable) nullable_in_slot[s]++;
        P.aggs[i].slot = s;
    }

    for (int i = 0; i < naggs; i++) {
        AggDesc &a = P.aggs[i];
        a.kind = aggs[i].kind;
        if (!kind_reads_values(aggs[i].kind)) a.slot = P.ncols > 0 ? 0 : -1;  // rides along with the first column pass
        int t = kind_type(aggs[i].kind);
        if (t == BOWGPU_INPUT_DEPENDENT) t = cols[aggs[i].col].type;      // aggregation.go:114-115
        if (t == BOWGPU_ITERATOR_DEPENDENT) t = tsc->type;                // aggregation.go:116-117
        a.out_type = t;
        a.n_factors = aggs[i].n_factors;
        for (int f = 0; f < a.n_factors; f++) a.factors[f] = aggs[i].factors[f];
        BG_TRY(devout_prepare(c, &outs[i], W, &job->douts[i], i));
        a.out_values = job->douts[i].values;
        a.out_valid = kind_never_nil(aggs[i].kind) ? nullptr : reinterpret_cast<uint32_t *>(job->douts[i].validity);
    }

    // per-pass summaries
    P.first_pass_slot = kMaxCols;
    P.last_val_slot = -1;
    for (int i = 0; i < naggs; i++) {
        const AggDesc &d = P.aggs[i];
        const int k = d.kind;
        const int idx = d.slot + 1;
        P.pass_mask[idx] |= 1u << i;
        uint32_t fl = 0;
        if (kind_reads_values(k)) fl |= kPassNeedVals;
        if (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX) fl |= kPassMinMax;
        if (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST) fl |= kPassFirstLast;
        if (d.out_valid) fl |= kPassNullable;
        if (k >= BOWGPU_AGG_INTEGRAL_STEP && k <= BOWGPU_AGG_WAVG_LINEAR && d.slot >= 0) P.cols[d.slot].need_ts = 1;
        P.pass_flags[idx] |= fl;
        if (d.slot < P.first_pass_slot) P.first_pass_slot = d.slot;
        if (kind_reads_values(k) && d.slot > P.last_val_slot) P.last_val_slot = d.slot;
    }
    for (int s = 0; s <= kMaxCols; s++) {
        int nn = 0;
        for (int i = 0; i < naggs; i++) if ((P.pass_mask[s] >> i) & 1u) nn += P.aggs[i].out_valid != nullptr;
        if (nn > P.n_nullable_max) P.n_nullable_max = nn;
    }

    // status words + long-window list
    // at most one long window per tile (the lean kernels' tile = 512 rows), spread over kLongLists sub-lists
    const int64_t ntiles = (n + 511) / 512;
    const int64_t sub_cap = ntiles / kLongLists + 2;
    job->scratch_bytes = 8192 + (size_t)sub_cap * kLongLists * 16;
    void *dscr;
    BG_TRY(ctx_scratch(c, job->scratch_bytes, &dscr));
    P.status = reinterpret_cast<uint32_t *>(dscr);
    P.long_list = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(dscr) + 8192);
    P.long_cap = sub_cap;
    return 0;
}

// status words and valid counts live side by side in the device scratch block and come back in ONE copy
constexpr size_t kCountsOffset = 1024;                       // bytes: counts[kMaxAggs] behind the status words
constexpr size_t kReadbackBytes = kCountsOffset + 8 * kMaxAggs;
static_assert(kStatusWords * 4 <= kCountsOffset, "status words overlap the counters");

static void job_bitmaps(AggJob *job, const bowgpu_agg *aggs, int32_t naggs, bool all_ones, BitmapBatch *b) {
    memset(b, 0, sizeof *b);
    b->n = naggs;
    b->status_words = kStatusWords;
    b->nbits = job->W > 0 ? job->W : 0;
    b->status = job->P.status;
    b->counts = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(job->P.status) + kCountsOffset);
    for (int i = 0; i < naggs; i++) {
        b->work[i] = reinterpret_cast<uint32_t *>(job->douts[i].validity);
        const bowgpu_out *u = job->douts[i].user;
        b->user[i] = (u && u->residency == BOWGPU_DEVICE) ? u->validity : nullptr;
        // never-nil reducers: all-ones bitmap; nullable ones start all-null (bowbuffer.go:25) - except under the simple kernels,
        // where every bitmap starts as ones and only the bits of nil results are cleared
        b->ones[i] = all_ones || kind_never_nil(aggs[i].kind);
        b->count[i] = !kind_never_nil(aggs[i].kind);
    }
}

// valid counts of the nullable outputs + bitmaps into the caller's buffers (one launch) + copy-back of host-resident outputs,
// enqueued only (no sync); the counts come back with the status words (job_readback)
static int job_enqueue_tail(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs) {
    const int64_t W = job->W;
    if (W > 0) {
        BitmapBatch b;
        job_bitmaps(job, aggs, naggs, false, &b);
        if (W <= kFinishHostBits) {
            // a small call is a chain of dependent launches and little else: its last launch hands the status words and the counts to
            // the host itself instead of a copy command doing so behind it
            BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&b.host_block)));
            job->tail_wrote_host = true;
        } else if (job->counts_used) BG_HIP(hipMemsetAsync(b.counts, 0, 8 * kMaxAggs, c->stream));
        BG_TRY(launch_finish_bitmaps(c, b));
        job->counts_used = true;
    }
    for (int i = 0; i < naggs; i++) BG_TRY(devout_finish(c, &job->douts[i], W, job->P.aggs[i].out_type, 0, false));
    return 0;
}

static int job_readback(Ctx *c, AggJob *job, uint32_t **hstat, uint64_t **hcnt) {
    char *hp;
    BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&hp)));
    if (job->tail_wrote_host) job->tail_wrote_host = false;   // (finish_bitmaps_kernel is storing both there)
    else BG_HIP(hipMemcpyAsync(hp, job->P.status, kReadbackBytes, hipMemcpyDeviceToHost, c->stream));
    *hstat = reinterpret_cast<uint32_t *>(hp);
    *hcnt = reinterpret_cast<uint64_t *>(hp + kCountsOffset);
    return 0;
}

// Can the wave-tile kernels (rolling_simple.hip; time_weighted: rolling_tw.hip) take this call?  16-B aligned columns, no rows
// below s0, interval < 2^32, at most kSimpleMaxAggs outputs; rolling_simple.hip: exclusive windows without time-weighted
// reducers.  *wide: the rows reach 2^32 or more past output slot 0 (nanosecond timestamps): ids relative to each tile's window.
static bool simple_applies(const AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan &plan, bool time_weighted, int *need,
                           bool *is_int, bool *has_nulls, bool *wide) {
    const AggParams &P = job->P;
    if ((job->inclusive && !time_weighted) || !P.fits32 || naggs > kSimpleMaxAggs) return false;
    if (P.W <= 0) return false;

    // output slot 0 starts at s0 + wid_base * interval (wid_base != 0: a shard; it never lies above the shard's first row).  Rows
    // below it exist only as the frame's rows below s0 (P.pre_rows: they ride in window 0; the kernels force their ids)
    const int64_t slot0_start = P.s0 + (int64_t)((uint64_t)P.wid_base * (uint64_t)P.interval);
    if (plan.first_ts < slot0_start && !(P.pre_rows && P.wid_base == 0)) return false;
    // rows within 2^32 of slot 0: global 32-bit window ids; else (nanosecond timestamps) ids relative to each tile's first window
    *wide = (uint64_t)plan.last_ts - (uint64_t)slot0_start >= 0xFFFFFFF0ull || P.W >= 0xFFFFFFF0ll;
    // (columns that start on an 8-byte but not a 16-byte boundary - Arrow slices with odd offsets - stay here: 8-byte loads)
    *need = 0;
    *is_int = true;  // (reducers over the interval column itself)
    *has_nulls = false;
    for (int s = 0; s < P.ncols; s++) {
        *has_nulls = *has_nulls || P.cols[s].vbits != nullptr;
        *is_int = P.cols[0].type == BOWGPU_INT64;
    }
    for (int i = 0; i < naggs; i++) {
        const int k = aggs[i].kind;
        if (k >= BOWGPU_AGG_INTEGRAL_STEP && k <= BOWGPU_AGG_WAVG_LINEAR && !time_weighted) return false;
        if (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX) *need |= 1;
        if (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST) *need |= 2;
    }
    if (route_mask() & BOWGPU_ROUTE_NO_SIMPLE) return false;
    return true;
}

// the descriptor of the wave-tile kernels (rolling_simple.hip, rolling_tw.hip, rolling_fused.hip) from a built job
static void simple_params_build(const AggJob *job, const bowgpu_agg *aggs, int32_t naggs, bool wide, SimpleParams *out) {
    const AggParams &P = job->P;
    SimpleParams &S = *out;
    memset(&S, 0, sizeof S);
    S.ts = P.ts;
    S.n = P.n; S.interval = P.interval; S.W = P.W;
    S.wid_base = P.wid_base;
    S.magic = P.magic;
    S.s0 = P.s0 + (int64_t)((uint64_t)P.wid_base * (uint64_t)P.interval);
    S.m32 = P.m32; S.sh1 = P.sh1_32; S.sh2 = P.sh2_32;
    S.shift_k = 0;
    if (wide) {  // ids from (ts - tile base) >> k divided by interval >> k, k = trailing zero bits of the interval
        int k = 0;
        while (k < 31 && !(((uint64_t)P.interval >> k) & 1ull)) k++;
        S.shift_k = k;
        const uint64_t d = (uint64_t)P.interval >> k;
        int l = 0;
        while (l < 32 && (1ull << l) < d) l++;
        S.m32 = (uint32_t)((((1ull << l) - d) << 32) / d) + 1;
        S.sh1 = l < 1 ? (uint32_t)l : 1u;
        S.sh2 = l > 1 ? (uint32_t)(l - 1) : 0u;
    }
    S.naggs = naggs;
    S.ncols = P.ncols > 0 ? P.ncols : 1;
    S.values[0] = P.ts;  // only WindowStart / NumRows: any column serves as "the" column
    for (int s = 0; s < P.ncols; s++) {
        S.values[s] = P.cols[s].values;
        S.vbits[s] = P.cols[s].vbits; S.vbit0[s] = P.cols[s].vbit0; S.vwords[s] = P.cols[s].vwords;
        S.col_is_int[s] = P.cols[s].type == BOWGPU_INT64;
    }
    for (int i = 0; i < naggs; i++) {
        S.kind[i] = aggs[i].kind;
        S.nfac[i] = aggs[i].n_factors;
        for (int f = 0; f < aggs[i].n_factors && f < BOWGPU_MAX_FACTORS; f++) S.fac[i][f] = aggs[i].factors[f];
        S.col[i] = P.aggs[i].slot < 0 ? 0 : P.aggs[i].slot;
        S.out_values[i] = reinterpret_cast<uint64_t *>(P.aggs[i].out_values);
        S.out_valid[i] = P.aggs[i].out_valid;
    }
    for (int i = 0; i < naggs; i++) {
        const int k = aggs[i].kind;
        if (k == BOWGPU_AGG_INTEGRAL_STEP || k == BOWGPU_AGG_WAVG_STEP) S.need |= kNeedStep;
        if (k == BOWGPU_AGG_INTEGRAL_TRAPEZOID || k == BOWGPU_AGG_WAVG_LINEAR) S.need |= kNeedTrap;
        if (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX) S.need |= kNeedMinMax;
        if (k == BOWGPU_AGG_SUM || k == BOWGPU_AGG_MEAN) S.need |= kNeedSum;
        if (k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST) S.need |= kNeedFirstLast;
        const int cls = (k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX) ? 1 : (k == BOWGPU_AGG_INTEGRAL_STEP || k == BOWGPU_AGG_WAVG_STEP) ? 2
                        : (k == BOWGPU_AGG_INTEGRAL_TRAPEZOID || k == BOWGPU_AGG_WAVG_LINEAR) ? 3
                        : (k == BOWGPU_AGG_SUM || k == BOWGPU_AGG_MEAN || k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST) ? 0 : 4;
        S.kind_mask[cls] |= 1u << i;
        S.col_mask[S.col[i]] |= 1u << i;
    }
    S.status = P.status; S.long_list = P.long_list; S.long_cap = P.long_cap;
    S.inclusive = job->inclusive ? 1 : 0;
    S.pre_rows = P.pre_rows;
    S.unaligned_mask = (reinterpret_cast<uintptr_t>(P.ts) & 15) ? 0x80000000u : 0u;
    for (int s = 0; s < S.ncols; s++)
        if (reinterpret_cast<uintptr_t>(S.values[s]) & 15) S.unaligned_mask |= 1u << s;
}

static int job_launch_tiles(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan *plan, bool allow_simple,
                            bool *used_simple, bool force_large_list = false, bool *used_small_list = nullptr) {
    AggParams &P = job->P;
    const int64_t W = job->W;
    *used_simple = false;
    bool small_dummy = false;
    if (!used_small_list) used_small_list = &small_dummy;
    *used_small_list = false;
    if (W <= 0) { BG_HIP(hipMemsetAsync(P.status, 0, kReadbackBytes, c->stream)); return 0; }
    // The lean kernels cover exclusive windows without time-weighted reducers and without rows below s0; everything
    // else (and BOWGPU_ROUTE_FORCE_GENERAL, used by the tests to cover all of them) takes the general kernel.
    bool lean = !job->inclusive;   // (rows below s0 - P.pre_rows - are taken by the wave-tile kernels; the lean wave kernel declines them below)
    for (int i = 0; i < naggs; i++)
        if (aggs[i].kind >= BOWGPU_AGG_INTEGRAL_STEP && aggs[i].kind <= BOWGPU_AGG_WAVG_LINEAR) lean = false;
    const uint32_t route = route_mask();
    const bool force = (route & BOWGPU_ROUTE_FORCE_GENERAL) != 0;
    if (force) lean = false;
    int need = 0;
    bool is_int = false, has_nulls = false, wide = false;
    bool simple = lean && allow_simple && plan && simple_applies(job, aggs, naggs, *plan, false, &need, &is_int, &has_nulls, &wide);
    // time-weighted reducers / inclusive windows: the same wave-tile structure with ts staged as float64 (rolling_tw.hip)
    const bool tw = !lean && !force && allow_simple && plan &&
                    simple_applies(job, aggs, naggs, *plan, true, &need, &is_int, &has_nulls, &wide);
    simple = simple || tw;
    P.bits_preset = simple ? 1 : 0;
    {
        BitmapBatch b;
        job_bitmaps(job, aggs, naggs, simple, &b);
        if (job->check_plan && plan) { b.check_ts = P.ts; b.check_n = P.n; b.check_first = plan->first_ts; b.check_last = plan->last_ts; }
        BG_TRY(launch_preset_bitmaps(c, b));   // + status words and counters to zero (+ the check of a caller-supplied plan)
        job->counts_used = false;
    }
    // kernel_ms brackets the dominant kernel only, on the stream it runs on
    BG_HIP(hipEventRecord(c->ev0, c->stream));
    if (simple) {
        SimpleParams S;
        simple_params_build(job, aggs, naggs, wide, &S);
        if (tw) {
            // 32-bit staged timestamps: exact when float64(s0 of slot 0) + float64(offset) needs no rounding, i.e. every |ts| < 2^53
            // (BOWGPU_ROUTE_TW_F64: test / A-B switch that keeps the float64 form)
            const int64_t lim53 = 1ll << 53;
            const bool ts32 = !wide && !P.pre_rows && plan->first_ts > -lim53 && plan->last_ts < lim53 && S.s0 > -lim53 && !(route & BOWGPU_ROUTE_TW_F64);
            // a nullable column: the compacting form (rolling_twc.hip) where its 32-bit times and its head list allow; a tile that overflows
            // the list raises status[4] and the call is redone here with force_large_list (job_pass_complete), i.e. by rolling_tw.hip
            const bool compact = has_nulls && ts32 && !force_large_list && !(route & BOWGPU_ROUTE_TW_ROWS) && P.n / P.W >= kCompactMinAvgRows;
            if (compact) {
                BG_TRY(launch_rolling_twc(c, S, P.n / P.W > 128));   // (windows of 129 .. kCompactLongMaxAvgRows rows on average: 256 rows of look-ahead)
                *used_small_list = true;
                c->last_kernel_name = "rolling_twc_kernel";
            } else {
                BG_TRY(launch_rolling_tw(c, S, is_int, has_nulls, wide, ts32));
                c->last_kernel_name = "rolling_tw_kernel";
            }
        } else if ([&] {
                       // a nullable column whose outputs want sums AND extrema, windows of kCompactValuesMinAvgRows rows and more: the compacting
                       // form walks the valid points once where rolling_simple.hip walks the rows twice (or once under the validity bit) - 1e8
                       // rows, 30 % nulls, Sum + Min + Max: 0.412 against 0.440 ms at 64 rows per window, 0.451 against 0.558 at 128; below
                       // that length, and for every other value set, rolling_simple.hip stays ahead (the A/B of round 5, timing only; the sweep that shows the routed
                       // kernels side by side is profiles/r05_stdout_midw_sweep.txt)
                       const int64_t lim53 = 1ll << 53;
                       const bool ts32 = !wide && !P.pre_rows && plan->first_ts > -lim53 && plan->last_ts < lim53 && S.s0 > -lim53;
                       const bool sums_and_extrema = (S.need & kNeedSum) && (S.need & kNeedMinMax);
                       // (beyond 128 rows per window the call is here only because job_run kept it off the streaming form: compact_long_max_rows)
                       if (job->band_rows) return false;   // (job_run: extrema / First + Last alone on a nullable column in the 129 .. 200 band - this kernel + the queue launch)
                       if (has_nulls && ts32 && !force_large_list && !(route & BOWGPU_ROUTE_TW_ROWS) && P.n / P.W > 128 && P.n / P.W <= compact_long_max_rows(aggs, naggs)) return true;
                       return has_nulls && ts32 && sums_and_extrema && !force_large_list && !(route & BOWGPU_ROUTE_TW_ROWS) && P.n / P.W >= kCompactValuesMinAvgRows;
                   }()) {
            BG_TRY(launch_rolling_twc(c, S, P.n / P.W > 128));
            *used_small_list = true;
            c->last_kernel_name = "rolling_twc_kernel";
        } else {
            // (the head list of a tile comes in two sizes - rolling_simple.hip SimpleCap: the small one buys four more resident
            // wavefronts per CU and serves calls whose windows average >= 5 rows; BOWGPU_ROUTE_SIMPLE_LARGE_LIST / _SMALL_LIST force
            // either, for tests; a call whose tiles overflow the small list is redone with the large one - job_run)
            // (the unpadded instantiation's small list holds 254 heads: windows of 3 rows and more, as before the pads)
            const int64_t small_list_rows = rolling_simple_plain(S, is_int, has_nulls) ? 3 : 5;
            const bool dense = force_large_list || ((route & BOWGPU_ROUTE_SIMPLE_LARGE_LIST) ? true : (route & BOWGPU_ROUTE_SIMPLE_SMALL_LIST) ? false : P.n / P.W < small_list_rows);
            *used_small_list = !dense;
            BG_TRY(launch_rolling_simple(c, S, need, is_int, has_nulls, wide, dense));   // (sets c->last_kernel_name: the instantiation)
        }
        *used_simple = true;
    } else if (lean && !P.pre_rows) {
        BG_TRY(launch_rolling_fast(c, P));
        c->last_kernel_name = "rolling_wave_kernel";
    } else {
        static const bool trace = [] { const char *t = getenv("BOWGPU_TRACE_ROUTE"); return t && t[0] == '1'; }();
        if (trace && !force && !(route & BOWGPU_ROUTE_NO_SIMPLE) && allow_simple)
            fprintf(stderr, "bowgpu route: general kernel (n=%lld W=%lld interval=%lld inclusive=%d naggs=%d pre_rows=%lld wid_base=%lld "
                            "fits32=%d allow_simple=%d plan=%d first_ts=%lld s0=%lld)\n", (long long)P.n, (long long)P.W,
                    (long long)P.interval, (int)job->inclusive, naggs, (long long)P.pre_rows, (long long)P.wid_base, (int)P.fits32,
                    (int)allow_simple, plan ? 1 : 0, plan ? (long long)plan->first_ts : 0ll, (long long)P.s0);
        BG_TRY(launch_rolling_aggregate(c, P));
        c->last_kernel_name = "rolling_agg_kernel";
        c->last_slow_rows += P.n;   // (the general kernel: 0.32 of the HBM peak where the wave-tile kernels reach 0.6 - 0.7)
    }
    BG_HIP(hipEventRecord(c->ev1, c->stream));
    return 0;
}

// Windows longer than a tile's look-ahead: workspace from the context pool, then long_windows.hip
struct LongWork {
    void *entries; int32_t *nchunks; int64_t *offsets, *sums, *total; int32_t *work_entry; void *parts;
    int64_t max_work;
};
static int long_workspace(Ctx *c, const AggParams &P, int64_t n_entries, LongWork *w) {
    const int64_t max_work = n_entries + (P.n + kLongChunkRows - 1) / kLongChunkRows;
    const int64_t scan_blocks = (n_entries + 2047) / 2048;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_entries = up((size_t)n_entries * long_entry_size());
    const size_t b_nchunks = up((size_t)n_entries * 4);
    const size_t b_offsets = up((size_t)(n_entries + 1) * 8);
    const size_t b_sums = up((size_t)(scan_blocks + 1) * 8);
    const size_t b_parts = up((size_t)max_work * (size_t)(P.ncols > 0 ? P.ncols : 1) * long_part_size());
    const size_t b_map = up((size_t)max_work * 4);
    void *blk;
    BG_TRY(ctx_pool(c, Ctx::kPoolSlots - 1, b_entries + b_nchunks + b_offsets + b_sums + 256 + b_map + b_parts, &blk));
    char *q = reinterpret_cast<char *>(blk);
    w->entries = q; q += b_entries;
    w->nchunks = reinterpret_cast<int32_t *>(q); q += b_nchunks;
    w->offsets = reinterpret_cast<int64_t *>(q); q += b_offsets;
    w->sums = reinterpret_cast<int64_t *>(q); q += b_sums;
    w->total = reinterpret_cast<int64_t *>(q); q += 256;
    w->work_entry = reinterpret_cast<int32_t *>(q); q += b_map;
    w->parts = q;
    w->max_work = max_work;
    return 0;
}
// hstat == nullptr: the long-only pipeline (every window of the call)
static int run_long_windows(Ctx *c, const AggParams &P, const uint32_t *hstat, int64_t *n_long_out, bool strict = false) {
    LongListStarts starts;
    starts.start[0] = 0;
    if (hstat)
        for (int s = 0; s < kLongLists; s++) starts.start[s + 1] = starts.start[s] + hstat[kLongCountWord + s];
    const int64_t n_long = hstat ? starts.start[kLongLists] : P.W;
    *n_long_out = n_long;
    if (n_long == 0) return 0;
    LongWork w;
    BG_TRY(long_workspace(c, P, n_long, &w));
    return launch_long_windows_v2(c, P, hstat ? &starts : nullptr, w.entries, w.nchunks, w.offsets, w.sums, w.total, w.work_entry, w.parts, w.max_work, strict);
}

// BOWGPU_CALL_PROFILE=1 (diagnostic): where a call's wall time goes on the host - until the synchronisation, inside it, after it
static thread_local double g_prof_sync_begin = 0, g_prof_sync_end = 0;
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

struct PassState {
    bool used_simple = false, used_small_list = false;
    uint32_t *hstat = nullptr;
    uint64_t *hcnt = nullptr;
    bool queue = false;     // long_queue_kernel rides behind the tile kernel: the queued windows are served without the host in between
    LongWork qw{};          // ... its workspace (sized for the queue's capacity)
};
// From how many rows per window (on average) on a tile pass is followed by long_queue_kernel unconditionally: below, windows longer than a
// tile's look-ahead are the exception and a call pays nothing for them until one shows up (the host then reads the counts and launches the
// machinery: rounds 1 - 5); from here on they are expected, and the launch (a few microseconds when the queue is empty) replaces a host
// round trip and eight launches
constexpr int64_t kQueueMinAvgRows = 64;
static int job_queue_enqueue(Ctx *c, AggJob *job, PassState *ps) {
    if (!ps->queue) return 0;
    const AggParams &P = job->P;
    BG_TRY(launch_long_queue(c, P, P.long_cap * kLongLists, ps->qw.entries, ps->qw.nchunks, g_strict_order ? ((int64_t)1 << 20) : kQueueWalkMaxRows, g_strict_order));
    BG_HIP(hipEventRecord(c->ev1, c->stream));   // (the bracket of kernel_ms: tile kernel + queue)
    return 0;
}
static int job_pass_enqueue(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan *plan, bool finish, PassState *ps);
static int job_pass_complete(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan *plan, bool finish, PassState *ps,
                             int64_t *long_windows, double *kernel_ms);

static int job_run(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, int64_t *long_windows, double *kernel_ms,
                   bool finish, const Plan *plan = nullptr, bool allow_long_only = false) {
    AggParams &P = job->P;
    const int64_t W = job->W;
    // Long-only pipeline: when the windows of an unsharded call average thousands of rows, nearly all of them would be queued
    // for long_windows.hip by a tile kernel that reads every row just to find that out.  Skip it: order check of the interval
    // column + every window as an entry of the multi-workgroup reduction.  (BOWGPU_ROUTE_NO_LONG_ONLY: test switch.)
    const uint32_t route = route_mask();
    const bool nlo = (route & BOWGPU_ROUTE_NO_LONG_ONLY) != 0 || g_strict_order;
    const bool cls = (route & BOWGPU_ROUTE_LONG_CLASSIC) != 0;      // test / A-B switches: only the bisection + per-window chunks form ...
    const bool sall = (route & BOWGPU_ROUTE_LONG_STREAM_ALL) != 0;  // ... / the streaming form for every reducer set
    // Which form?  The streaming form - one read of the rows, long_windows.hip long_short_kernel / long_stream_kernel - for every
    // reducer set from 128 rows per window on average (1e8 rows: 1000-row windows 0.26 - 0.33 ms against 0.44 - 0.72 ms for the
    // bisection form; 128 .. 256-row windows 0.35 - 0.55 ms dense, 0.40 - 0.76 ms on irregular data with nulls, against 0.35 - 1.0 /
    // 0.7 - 1.8 ms for the tile kernels, whose lane-per-window walk idles most lanes at that length); the bisection form keeps the
    // calls of a handful of giant windows (from 4M rows per window: the streaming form's final merge is one workgroup per window).
    // (round 4: the tile kernels' walks are branch-free and their staged columns padded against LDS bank conflicts - at 128 rows per
    // window they beat the streaming form for every set but the calls with both kinds of integral (Mean: 0.283 against 0.319 ms dense,
    // 0.327 against 0.341 with nulls), scratch/midw_sweep.py; from 129 on some window of the call no longer fits a tile's look-ahead and
    // the cooperative path would have to run as well: the streaming form takes over)
    bool step_k = false, trap_k = false;
    for (int i = 0; i < naggs; i++) {
        step_k |= aggs[i].kind == BOWGPU_AGG_INTEGRAL_STEP || aggs[i].kind == BOWGPU_AGG_WAVG_STEP;
        trap_k |= aggs[i].kind == BOWGPU_AGG_INTEGRAL_TRAPEZOID || aggs[i].kind == BOWGPU_AGG_WAVG_LINEAR;
    }
    const int64_t avg_rows = W > 0 ? P.n / W : 0;
    const bool classic_only = cls;
    // (round 5: with a nullable column the tile kernels compact the valid points first - rolling_twc.hip - and beat the streaming form at 128
    // rows per window for both kinds of integral too: 0.53 against 0.72 ms per 1e8 rows)
    bool any_nulls = false;
    for (int s = 0; s < P.ncols; s++) any_nulls = any_nulls || P.cols[s].vbits != nullptr;
    // ... and with 256 rows of look-ahead the same kernel keeps the calls of 129 .. 176 / 255 rows per window on a nullable column whose
    // reducer set the streaming form serves worst (compact_long_max_rows)
    const int64_t lim53 = 1ll << 53;
    // Round 6: the windows a tile pass queues are served behind it without the host (long_queue_kernel: a lane per queued window, in row
    // order), which moves the hand-over to the streaming form for columns WITHOUT nulls from 129 rows per window to where the streaming
    // form wins on WALL (1e8 rows, dense, wall ms tile route / streaming form at 144, 160, 192, 224 rows per window -
    // profiles/r06_stdout_midw_band_first.txt): Min + Max 0.407 / 0.547, 0.420 / 0.510, 0.455 / 0.472, 0.504 / 0.448; Sum + Min + Max
    // 0.445 / 0.549, 0.463 / 0.507, 0.508 / 0.472; First + Last 0.363 / 0.412, 0.368 / 0.396, 0.408 / 0.379; one kind of integral
    // 0.458 / 0.504, 0.470 / 0.483, 0.511 / 0.454; sums and counts alone and both kinds of integral: the streaming form throughout.
    // ... and for NULLABLE columns when the set has neither sums nor integrals (the null rows are staged as NaN and never win; no second walk):
    // 1e8 rows, 30 % nulls, wall ms rolling_simple.hip + queue / what round 5 routed (rolling_twc.hip, from 192 rows the streaming form) at 144,
    // 160, 192 rows per window: Min + Max 0.448 / 0.575, 0.454 / 0.581, 0.468 / 0.568 (224 rows: 0.562 / 0.543); First + Last 0.419 / 0.466,
    // 0.422 / 0.466, 0.450 / 0.460; Sum + Min + Max gains nothing (0.594 / 0.608) and keeps round 5's rule (profiles/r06_stdout_nullable_band_ab.txt)
    int64_t tile_band_rows = 0;
    bool no_sum_set = false;
    {
        bool mm = false, fl = false, sums = false;
        for (int i = 0; i < naggs; i++) {
            const int k = aggs[i].kind;
            mm |= k == BOWGPU_AGG_MIN || k == BOWGPU_AGG_MAX;
            fl |= k == BOWGPU_AGG_FIRST || k == BOWGPU_AGG_LAST;
            sums |= k == BOWGPU_AGG_SUM || k == BOWGPU_AGG_MEAN;
        }
        no_sum_set = (mm || fl) && !sums && !step_k && !trap_k;
        if (step_k && trap_k) tile_band_rows = 0;
        else if (mm && !sums && !step_k && !trap_k) tile_band_rows = 200;   // (240 was tried once the tile kernel found the extrema of long windows with all its lanes: at 224 rows the queue walk costs more than that saves - 0.498 against 0.453 ms wall, profiles/r06_stdout_midw_band.txt)
        else if (mm || fl || step_k || trap_k) tile_band_rows = 176;
    }
    const bool tile_band = !sall && !cls && (!any_nulls || (no_sum_set && !(route & BOWGPU_ROUTE_TW_ROWS))) && plan && avg_rows > 128 && avg_rows <= tile_band_rows && P.fits32 && !P.pre_rows &&
                           plan->first_ts > -lim53 && plan->last_ts < lim53 && (uint64_t)plan->last_ts - (uint64_t)P.s0 < 0xFFFFFFF0ull &&
                           W < 0xFFFFFFF0ll && naggs <= kSimpleMaxAggs &&
                           !(route & (BOWGPU_ROUTE_NO_SIMPLE | BOWGPU_ROUTE_FORCE_GENERAL | BOWGPU_ROUTE_QUEUE_HOST)) && !g_strict_order;
    const bool compact_long = !sall && !cls && !tile_band && any_nulls && plan && avg_rows > 128 && avg_rows <= compact_long_max_rows(aggs, naggs) && P.fits32 &&
                              !P.pre_rows && plan->first_ts > -lim53 && plan->last_ts < lim53 &&
                              (uint64_t)plan->last_ts - (uint64_t)P.s0 < 0xFFFFFFF0ull && W < 0xFFFFFFF0ll && naggs <= kSimpleMaxAggs &&
                              !(route & (BOWGPU_ROUTE_TW_ROWS | BOWGPU_ROUTE_TW_F64 | BOWGPU_ROUTE_NO_SIMPLE | BOWGPU_ROUTE_FORCE_GENERAL)) && !g_strict_order;
    job->band_rows = tile_band && any_nulls;
    const bool stream_ok = !classic_only && !compact_long && !tile_band &&
                           avg_rows >= ((sall || (step_k && trap_k && !any_nulls)) ? kLongOnlyAvgRows : kLongStreamAnyAvgRows) &&
                           avg_rows < kLongClassicAvgRows && W < (1ll << 32);
    const bool classic_ok = avg_rows >= kLongBisectAvgRows;
    // bowgpu_options.strict_order on a call of long windows: every window by one lane in row order (long_windows.hip
    // long_strict_kernel) instead of a tile kernel that reads every row only to queue nearly every window
    const bool strict_long = g_strict_order && avg_rows >= kLongStreamAnyAvgRows && !(route & BOWGPU_ROUTE_NO_LONG_ONLY);
    if (allow_long_only && plan && W > 0 && P.wid_base == 0 && (((stream_ok || classic_ok) && !nlo) || strict_long)) {
        P.bits_preset = 0;
        {
            BitmapBatch b;
            job_bitmaps(job, aggs, naggs, false, &b);
            if (job->check_plan) { b.check_ts = P.ts; b.check_n = P.n; b.check_first = plan->first_ts; b.check_last = plan->last_ts; }
            BG_TRY(launch_preset_bitmaps(c, b));
        }
        BG_HIP(hipEventRecord(c->ev0, c->stream));
        int64_t n_all = W;
        if (strict_long) {
            BG_TRY(run_long_windows(c, P, nullptr, &n_all, true));
            n_all = 0;   // (bowgpu_agg_info.long_windows counts the windows reduced order-free: none)
            c->last_kernel_name = "long_strict_kernel";
        } else if (!stream_ok) {
            BG_TRY(run_long_windows(c, P, nullptr, &n_all));
            c->last_kernel_name = "long_partial_kernel";
        } else {
            void *w;
            BG_TRY(ctx_pool(c, Ctx::kPoolSlots - 1, long_stream_workspace(P.n, P.W, P.ncols), &w));
            BG_TRY(launch_long_stream(c, P, w));
            c->last_kernel_name = "long_stream_kernel";
        }
        BG_HIP(hipEventRecord(c->ev1, c->stream));
        uint32_t *hs;
        uint64_t *hc = nullptr;
        if (finish) BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
        BG_TRY(job_readback(c, job, &hs, &hc));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (hs[6]) return fail(BOWGPU_ERR_ARG, "the plan was not made for this interval column (its first / last timestamp differ)");
        if (hs[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
        if (hs[7]) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: some window holds more than 2^20 rows (one lane walks a window in row order: beyond that the call is declined)");
        if (finish)
            for (int i = 0; i < naggs; i++)
                job->douts[i].user->null_count = !kind_never_nil(aggs[i].kind) ? W - (int64_t)hc[i] : 0;
        if (long_windows) *long_windows = n_all;
        float ms = 0;
        BG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->last_kernel_ms = ms;
        if (kernel_ms) *kernel_ms = ms;
        return 0;
    }
    PassState ps;
    BG_TRY(job_pass_enqueue(c, job, aggs, naggs, plan, finish, &ps));
    return job_pass_complete(c, job, aggs, naggs, plan, finish, &ps, long_windows, kernel_ms);
}

// the tile pass of a call, enqueued only: bitmap preset, tile kernel, (tail), read-back of the status words - no synchronisation
static int job_pass_enqueue(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan *plan, bool finish, PassState *ps) {
    {
        const uint32_t route = route_mask();
        const int64_t avg_rows = job->W > 0 ? job->P.n / job->W : 0;
        ps->queue = job->W > 0 && !(route & BOWGPU_ROUTE_QUEUE_HOST) && (avg_rows >= kQueueMinAvgRows || (route & BOWGPU_ROUTE_QUEUE_DEVICE));
        if (ps->queue) BG_TRY(long_workspace(c, job->P, job->P.long_cap * kLongLists, &ps->qw));
    }
    BG_TRY(job_launch_tiles(c, job, aggs, naggs, plan, true, &ps->used_simple, false, &ps->used_small_list));
    BG_TRY(job_queue_enqueue(c, job, ps));
    // status -> host (pinned).  Optimistically enqueue the tail (null counts, copy-back) behind the tile kernel so
    // the common case needs ONE synchronisation; if windows were queued for the cooperative path, run it and redo the tail.
    if (finish) BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
    BG_TRY(job_readback(c, job, &ps->hstat, &ps->hcnt));
    return 0;
}

// ... and its completion: the one synchronisation, the redo of a call the wave-tile kernels could not describe, the queued long windows
static int job_pass_complete(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, const Plan *plan, bool finish, PassState *ps,
                             int64_t *long_windows, double *kernel_ms) {
    AggParams &P = job->P;
    const int64_t W = job->W;
    bool used_simple = ps->used_simple, used_small_list = ps->used_small_list;
    uint32_t *hstat = ps->hstat;
    uint64_t *hcnt = ps->hcnt;
    g_prof_sync_begin = now_us();
    BG_HIP(hipStreamSynchronize(c->stream));
    g_prof_sync_end = now_us();
    if (hstat[6]) return fail(BOWGPU_ERR_ARG, "the plan was not made for this interval column (its first / last timestamp differ)");
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    if (used_simple && hstat[4] && used_small_list) {
        // clumpy data: a high average of rows per window, but some tile holds more window heads than the small list takes - the
        // same kernel with its large list (400 heads per 640 rows) before anything slower is tried
        BG_TRY(job_launch_tiles(c, job, aggs, naggs, plan, true, &used_simple, true, &used_small_list));
        BG_TRY(job_queue_enqueue(c, job, ps));
        if (finish) BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
        BG_TRY(job_readback(c, job, &hstat, &hcnt));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    }
    if (used_simple && hstat[4]) {
        // some tile needs window ids the simple kernel cannot encode: redo the call with the general lean kernel
        BG_TRY(job_launch_tiles(c, job, aggs, naggs, plan, false, &used_simple));
        BG_TRY(job_queue_enqueue(c, job, ps));
        if (finish) BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
        BG_TRY(job_readback(c, job, &hstat, &hcnt));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    }
    if (hstat[2]) return fail(BOWGPU_ERR_HIP, "internal: long-window list overflow");
    int64_t n_long = 0;
    if (ps->queue) {
        // long_queue_kernel has walked the queued windows of up to kQueueWalkMaxRows rows in row order (exact) behind the tile kernel; the
        // longer ones are listed for the chunked order-free machinery - usually none, and the call is done with its one synchronisation
        if (hstat[7]) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: some window holds more than 2^20 rows (one lane walks a window in row order: beyond that the call is declined)");
        const int64_t n_big = (int64_t)hstat[kQueueBigWord];
        if (n_big > 0) {
            const LongWork &w = ps->qw;
            BG_TRY(launch_long_windows_v2(c, P, nullptr, w.entries, w.nchunks, w.offsets, w.sums, w.total, w.work_entry, w.parts, w.max_work, false, n_big));
            if (finish) {
                BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
                BG_TRY(job_readback(c, job, &hstat, &hcnt));
                BG_HIP(hipStreamSynchronize(c->stream));
            }
        }
        n_long = n_big;
    } else {
        // (strict_order: the windows a tile could not hold are walked in row order by one lane each - long_strict_kernel - instead of
        // being reduced as a tree; a window beyond 2^20 rows declines the call)
        BG_TRY(run_long_windows(c, P, hstat, &n_long, g_strict_order));
        if (n_long > 0 && (finish || g_strict_order)) {
            if (finish) BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
            BG_TRY(job_readback(c, job, &hstat, &hcnt));
            BG_HIP(hipStreamSynchronize(c->stream));
            if (hstat[7]) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: some window holds more than 2^20 rows (one lane walks a window in row order: beyond that the call is declined)");
        }
        if (g_strict_order) n_long = 0;   // (none of them was reduced order-free)
    }
    if (finish)
        for (int i = 0; i < naggs; i++)
            job->douts[i].user->null_count = (W > 0 && !kind_never_nil(aggs[i].kind)) ? W - (int64_t)hcnt[i] : 0;
    if (long_windows) *long_windows = n_long;
    {
        float ms = 0;
        if (W > 0) BG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->last_kernel_ms = ms;
        if (kernel_ms) *kernel_ms = ms;
    }
    return 0;
}

static int job_finish(Ctx *c, AggJob *job, const bowgpu_agg *aggs, int32_t naggs, uint32_t *strict_limit_hit = nullptr) {
    uint32_t *hstat;
    uint64_t *hcnt = nullptr;
    BG_TRY(job_enqueue_tail(c, job, aggs, naggs));
    BG_TRY(job_readback(c, job, &hstat, &hcnt));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (strict_limit_hit) *strict_limit_hit = hstat[7];
    for (int i = 0; i < naggs; i++)
        job->douts[i].user->null_count = (job->W > 0 && !kind_never_nil(aggs[i].kind)) ? job->W - (int64_t)hcnt[i] : 0;
    return 0;
}

// aggregation.Mode outputs (mode.go:8-32): not a streaming reducer, so they run apart from the tile kernels, over the
// windows' row ranges (an inclusive window reaches them without its extra row: window.go:23-31, aggregation.go:207-211)
static int run_modes(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, const bowgpu_agg *aggs,
                     int32_t naggs, bowgpu_out *outs, int inclusive, int64_t *long_windows) {
    const bowgpu_col *tsc = &cols[ts_col];
    const int64_t n = tsc->length, W = plan.W;
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has %lld rows, interval column has %lld", i, (long long)cols[i].length, (long long)n);
    if (tsc->validity && tsc->null_count != 0) {
        DevCol probe;
        BG_TRY(devcol_prepare(c, tsc, &probe, false, true));
        if (probe.null_count > 0)
            return fail(BOWGPU_ERR_TS_NULLS, "interval column has %lld nulls: outside the device path", (long long)probe.null_count);
    }
    DevCol dts;
    DevBuf first_idx;
    if (W > 0) {
        bowgpu_col t = *tsc;
        t.validity = nullptr;
        t.null_count = 0;
        BG_TRY(devcol_prepare(c, &t, &dts, true, false));
        BG_TRY(first_idx.alloc((size_t)(W + 1) * 8));
        void *dscr;
        BG_TRY(ctx_scratch(c, 8192, &dscr));
        uint32_t *status = reinterpret_cast<uint32_t *>(dscr);
        BG_HIP(hipMemsetAsync(status, 0, 64, c->stream));
        BG_TRY(launch_window_first_rows(c, reinterpret_cast<const int64_t *>(dts.values), n, plan, reinterpret_cast<int64_t *>(first_idx.p), status));
        uint32_t hstat[4] = {0, 0, 0, 0};
        BG_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
        if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    }
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].kind != BOWGPU_AGG_MODE) continue;
        const int col = aggs[i].col;
        DevCol dc;
        const void *values = dts.values;
        const uint32_t *vbits = nullptr;
        int64_t vbit0 = 0;
        if (col != ts_col && W > 0) {
            BG_TRY(devcol_prepare(c, &cols[col], &dc, true, true));
            values = dc.values; vbits = dc.vbits; vbit0 = dc.vbit0;
        }
        DevOut d;
        BG_TRY(devout_prepare(c, &outs[i], W, &d, kPoolMode));
        int64_t nulls = 0;
        if (W > 0) {
            const size_t vb = (size_t)((W + 7) >> 3);
            BG_HIP(hipMemsetAsync(d.validity, 0, ((vb + 3) & ~(size_t)3) + 4, c->stream));
            int64_t n_mid = 0, n_long = 0;
            BG_TRY(launch_mode(c, reinterpret_cast<const int64_t *>(dts.values), reinterpret_cast<const int64_t *>(first_idx.p), n, plan.s0, plan.interval, W,
                               plan.s0 > plan.first_ts ? 1 : 0, inclusive, values, vbits, vbit0, cols[col].type == BOWGPU_INT64, &aggs[i], d.values,
                               reinterpret_cast<uint32_t *>(d.validity), &n_mid, &n_long));
            // (Mode is bit-exact in every size class; the classes beyond a lane's are reported for the tests - not under strict_order,
            // whose contract is long_windows == 0)
            if (long_windows && !g_strict_order) *long_windows += n_mid + n_long;
            void *dscr;
            BG_TRY(ctx_scratch(c, 8192, &dscr));
            uint64_t *dcnt = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(dscr) + 1024);
            uint64_t hcnt = 0;
            BG_TRY(launch_popcount(c, reinterpret_cast<const uint32_t *>(d.validity), 0, W, dcnt));
            BG_HIP(hipMemcpyAsync(&hcnt, dcnt, 8, hipMemcpyDeviceToHost, c->stream));
            BG_HIP(hipStreamSynchronize(c->stream));
            nulls = W - (int64_t)hcnt;
        }
        BG_TRY(devout_finish(c, &d, W, cols[col].type, nulls));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

static int run_aggregate(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan,
                         int inclusive, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                         int64_t wid_base, int64_t W, int64_t *long_windows, double *kernel_ms);

// An interval column WITH NULLS (ts_nulls.hip has the semantics: rolling.go:177-239 skips such rows, :143-154 counts windows from the
// last valid timestamp, :162-173 ends the iteration at once when the physically last timestamp is null).  The call is rewritten
// onto a dense interval column - nulls forward-filled - with every value column's validity ANDed with the rows that belong to a
// window (and, for the time-weighted reducers, with the interval column's own validity), and then takes the ordinary path into
// device temporaries; NumRows is counted as Count over the rows that belong to a window; after an inclusive iteration the windows
// behind a row on a window start with a null timestamp right behind it get the outputs of IntegralTrapezoid / WeightedAverageLinear
// from ts_quirk_fix_kernel.  Unsharded calls; Mode is declined.
static int run_aggregate_null_ts(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int inclusive,
                                 const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, DevCol &dts, int64_t *long_windows, double *kernel_ms) {
    const bowgpu_col *tsc = &cols[ts_col];
    const int64_t n = tsc->length, W = plan.W;
    for (int i = 0; i < naggs; i++)
        if (aggs[i].kind == BOWGPU_AGG_MODE)
            return fail(BOWGPU_ERR_TS_NULLS, "interval column has %lld nulls: Mode over it is outside the device path", (long long)dts.null_count);
    for (int i = 0; i < ncols; i++)
        if (cols[i].length != n) return fail(BOWGPU_ERR_ARG, "column %d has %lld rows, interval column has %lld", i, (long long)cols[i].length, (long long)n);
    if (long_windows) *long_windows = 0;
    if (kernel_ms) *kernel_ms = 0;
    int last_valid = 1;
    BG_TRY(fetch_valid(c, tsc, n - 1, &last_valid));
    if (!last_valid || W == 0) {
        // HasNext (rolling.go:162-173) is false from the start when the physically last timestamp is null: no window is ever
        // produced and every slot of the numWindows-long output buffers stays nil (bowbuffer.go:22-40: zero data, all-null bitmap)
        for (int i = 0; i < naggs; i++) {
            DevOut d;
            BG_TRY(devout_prepare(c, &outs[i], W, &d, i));
            int t = kind_type(aggs[i].kind);
            if (t == BOWGPU_INPUT_DEPENDENT) t = cols[aggs[i].col].type;
            if (t == BOWGPU_ITERATOR_DEPENDENT) t = tsc->type;
            if (W > 0) {
                BG_HIP(hipMemsetAsync(d.values, 0, (size_t)W * 8, c->stream));
                BG_HIP(hipMemsetAsync(d.validity, 0, (size_t)((W + 7) >> 3), c->stream));
            }
            BG_TRY(devout_finish(c, &d, W, t, W));
            BG_HIP(hipStreamSynchronize(c->stream));
        }
        return 0;
    }
    // the interval column forward-filled + which rows belong to a window (+ inclusive: its validity without the rows ts_nulls.hip describes)
    DevBuf ixbuf, ts_eff, keep, plain_ts, dropped;
    NbrIndex ix;
    BG_TRY(ixbuf.alloc(nbr_index_bytes(n, dts.vbit0)));
    BG_TRY(nbr_index_build(c, dts.vbits, dts.vbit0, n, ixbuf.p, &ix));
    BG_TRY(ts_eff.alloc((size_t)n * 8 + 16));
    const size_t bm_bytes = (size_t)((n + 63) >> 6) * 8 + 8;
    BG_TRY(keep.alloc(bm_bytes));
    if (inclusive) BG_TRY(plain_ts.alloc(bm_bytes));
    BG_TRY(dropped.alloc(16));
    BG_TRY(launch_ts_nullfill(c, reinterpret_cast<const int64_t *>(dts.values), dts.vbits, dts.vbit0, n, ix, plan.s0, plan.interval, plan.magic, inclusive,
                              reinterpret_cast<int64_t *>(ts_eff.p), reinterpret_cast<uint64_t *>(keep.p), reinterpret_cast<uint64_t *>(plain_ts.p),
                              reinterpret_cast<unsigned long long *>(dropped.p)));
    // one rewritten column per (input column, class of reducer) that some reducer reads:
    //   plain       value validity AND keep
    //   step        IntegralStep / WeightedAverageStep: their points are the rows with timestamp AND value (bowgetters.go:299-311); they read
    //               windows through UnsetInclusive, so after an inclusive iteration the rows ts_nulls.hip describes are not among them
    //   linear      IntegralTrapezoid / WeightedAverageLinear: value validity AND the interval column's
    //   rows        NumRows: Count over keep itself
    enum { kPlain = 0, kStep = 1, kLinear = 2, kClasses = 3 };
    std::vector<bowgpu_col> cols2(cols, cols + ncols);
    std::vector<bowgpu_agg> aggs2(aggs, aggs + naggs);
    std::vector<DevCol> dcs(ncols);
    std::vector<DevBuf> bitmaps;
    bitmaps.reserve(kClasses * (size_t)ncols + 1);
    std::vector<int> slot_of(kClasses * (size_t)ncols, -1);
    int rows_slot = -1;
    {
        bowgpu_col &t = cols2[ts_col];
        t.values = ts_eff.p; t.validity = nullptr; t.offset = 0; t.length = n; t.null_count = 0; t.type = BOWGPU_INT64; t.residency = BOWGPU_DEVICE;
    }
    auto device_col = [&](int col, const void **values, const uint32_t **vbits, int64_t *vbit0) -> int {
        if (col == ts_col) { *values = dts.values; *vbits = dts.vbits; *vbit0 = dts.vbit0; return 0; }
        DevCol &dc = dcs[col];
        if (dc.values == nullptr) BG_TRY(devcol_prepare(c, &cols[col], &dc, true, true));
        *values = dc.values; *vbits = dc.vbits; *vbit0 = dc.vbit0;
        return 0;
    };
    bool any_linear = false;
    for (int i = 0; i < naggs; i++) {
        const int kind = aggs[i].kind;
        if (kind == BOWGPU_AGG_NUM_ROWS) {
            if (rows_slot < 0) {
                bowgpu_col nc;
                nc.values = ts_eff.p; nc.validity = reinterpret_cast<const uint8_t *>(keep.p); nc.offset = 0; nc.length = n; nc.null_count = -1;
                nc.type = BOWGPU_INT64; nc.residency = BOWGPU_DEVICE;
                rows_slot = (int)cols2.size();
                cols2.push_back(nc);
            }
            aggs2[i].kind = BOWGPU_AGG_COUNT;
            aggs2[i].col = rows_slot;
            continue;
        }
        if (!kind_reads_values(kind)) continue;
        const int col = aggs[i].col;
        const bool linear = kind_needs_inclusive(kind);
        const bool tw = kind >= BOWGPU_AGG_INTEGRAL_STEP && kind <= BOWGPU_AGG_WAVG_LINEAR;
        const int cls = linear ? kLinear : tw ? kStep : kPlain;
        any_linear |= linear;
        int &slot = slot_of[(size_t)cls * ncols + col];
        if (slot < 0) {
            const uint32_t *vbits; int64_t vbit0; const void *values;
            BG_TRY(device_col(col, &values, &vbits, &vbit0));
            bitmaps.emplace_back();
            DevBuf &bm = bitmaps.back();
            BG_TRY(bm.alloc(bm_bytes));
            if (cls == kPlain) BG_TRY(launch_and_bits(c, vbits, vbit0, reinterpret_cast<const uint32_t *>(keep.p), 0, n, reinterpret_cast<uint64_t *>(bm.p)));
            else if (cls == kStep && inclusive) BG_TRY(launch_and_bits(c, vbits, vbit0, reinterpret_cast<const uint32_t *>(plain_ts.p), 0, n, reinterpret_cast<uint64_t *>(bm.p)));
            else BG_TRY(launch_and_bits(c, vbits, vbit0, dts.vbits, dts.vbit0, n, reinterpret_cast<uint64_t *>(bm.p)));
            bowgpu_col nc;
            nc.values = values; nc.validity = reinterpret_cast<const uint8_t *>(bm.p); nc.offset = 0; nc.length = n; nc.null_count = -1;
            nc.type = cols[col].type; nc.residency = BOWGPU_DEVICE;
            slot = (int)cols2.size();
            cols2.push_back(nc);
        }
        aggs2[i].col = slot;
    }
    // the ordinary path, into device temporaries
    const size_t vb = (size_t)((W + 7) >> 3);
    std::vector<DevBuf> tvals(naggs), tbits(naggs);
    std::vector<bowgpu_out> touts(naggs);
    for (int i = 0; i < naggs; i++) {
        BG_TRY(tvals[i].alloc((size_t)W * 8 + 8));
        BG_TRY(tbits[i].alloc(((vb + 3) & ~(size_t)3) + 8));
        bowgpu_out &t = touts[i];
        t.values = tvals[i].p; t.validity = reinterpret_cast<uint8_t *>(tbits[i].p); t.length = W; t.null_count = 0; t.type = 0; t.residency = BOWGPU_DEVICE;
    }
    BG_TRY(run_aggregate(c, cols2.data(), (int32_t)cols2.size(), ts_col, plan, inclusive, aggs2.data(), naggs, touts.data(), 0, W, long_windows, kernel_ms));
    for (int i = 0; i < naggs; i++)
        if (aggs[i].kind == BOWGPU_AGG_NUM_ROWS) {
            BG_TRY(launch_count_to_f64(c, reinterpret_cast<uint64_t *>(tvals[i].p), W));
            touts[i].type = BOWGPU_FLOAT64;
        }
    if (inclusive && any_linear) {
        unsigned long long *d_fixed = reinterpret_cast<unsigned long long *>(dropped.p) + 1;
        BG_HIP(hipMemsetAsync(d_fixed, 0, 8, c->stream));
        std::vector<int> fixed;
        QuirkFix fx;
        fx.naggs = 0; fx._pad = 0;
        auto flush = [&]() -> int {
            if (fx.naggs) BG_TRY(launch_ts_quirk_fix(c, reinterpret_cast<const int64_t *>(dts.values), dts.vbits, dts.vbit0, n, ix, plan.s0, plan.interval, plan.magic,
                                                    W, fx, d_fixed));
            fx.naggs = 0;
            return 0;
        };
        for (int i = 0; i < naggs; i++) {
            if (!kind_needs_inclusive(aggs[i].kind)) continue;
            QuirkFixAgg &fa = fx.a[fx.naggs++];
            BG_TRY(device_col(aggs[i].col, &fa.values, &fa.vbits, &fa.vbit0));
            fa.out_values = reinterpret_cast<uint64_t *>(tvals[i].p); fa.out_valid = reinterpret_cast<uint32_t *>(tbits[i].p);
            fa.type = cols[aggs[i].col].type; fa.kind = aggs[i].kind; fa.n_factors = aggs[i].n_factors; fa._pad = 0;
            for (int f = 0; f < BOWGPU_MAX_FACTORS; f++) fa.factors[f] = aggs[i].factors[f];
            fixed.push_back(i);
            if (fx.naggs == 8) BG_TRY(flush());
        }
        BG_TRY(flush());
        // their null counts again
        void *dscr;
        BG_TRY(ctx_scratch(c, 8192, &dscr));
        uint64_t *dcnt = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(dscr) + 1024);
        for (int i : fixed) {
            uint64_t hcnt = 0;
            BG_TRY(launch_popcount(c, reinterpret_cast<const uint32_t *>(tbits[i].p), 0, W, dcnt));
            BG_HIP(hipMemcpyAsync(&hcnt, dcnt, 8, hipMemcpyDeviceToHost, c->stream));
            BG_HIP(hipStreamSynchronize(c->stream));
            touts[i].null_count = W - (int64_t)hcnt;
        }
    }
    // ... and on to the caller's columns
    for (int i = 0; i < naggs; i++) {
        DevOut d;
        BG_TRY(devout_prepare(c, &outs[i], W, &d, i));
        BG_HIP(hipMemcpyAsync(d.values, tvals[i].p, (size_t)W * 8, hipMemcpyDeviceToDevice, c->stream));
        BG_HIP(hipMemcpyAsync(d.validity, tbits[i].p, vb, hipMemcpyDeviceToDevice, c->stream));
        BG_TRY(devout_finish(c, &d, W, touts[i].type, touts[i].null_count, true));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// An unsharded Aggregate call: the streaming reducers in batches that one launch of the tile kernels takes (at most kMaxAggs
// outputs over at most kMaxCols column passes - the reference has no such limits, aggregation.go:190-238 simply loops), then the
// Mode outputs.
static int run_aggregate(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan,
                         int inclusive, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                         int64_t wid_base, int64_t W, int64_t *long_windows, double *kernel_ms) {
    int n_mode = 0;
    for (int i = 0; i < naggs; i++) n_mode += aggs[i].kind == BOWGPU_AGG_MODE;
    if (n_mode > 0 && (wid_base != 0 || W != plan.W)) return fail(BOWGPU_ERR_UNSUPPORTED, "Mode runs on unsharded calls only");
    if (wid_base == 0 && W == plan.W && cols[ts_col].validity && cols[ts_col].null_count != 0 && cols[ts_col].length > 0) {
        DevCol dts;   // (values + validity on the device; counts the nulls when the caller did not)
        BG_TRY(devcol_prepare(c, &cols[ts_col], &dts, true, true));
        if (dts.null_count > 0) return run_aggregate_null_ts(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, dts, long_windows, kernel_ms);
    }
    if (long_windows) *long_windows = 0;
    if (kernel_ms) *kernel_ms = 0;
    // greedy batches in output order; the column-pass count follows job_build's rule (a pass serves at most 4 nullable reducers)
    std::vector<std::vector<int>> batches;
    {
        std::vector<int> cur, slot_of(ncols, -1), nullable_in_slot;
        auto flush = [&]() {
            if (!cur.empty()) batches.push_back(cur);
            cur.clear(); nullable_in_slot.clear();
            std::fill(slot_of.begin(), slot_of.end(), -1);
        };
        for (int i = 0; i < naggs; i++) {
            if (aggs[i].kind == BOWGPU_AGG_MODE) continue;
            for (int attempt = 0; attempt < 2; attempt++) {
                bool fits = (int)cur.size() < kMaxAggs;
                int s = -1;
                bool new_slot = false;
                const bool reads = kind_reads_values(aggs[i].kind), nullable = !kind_never_nil(aggs[i].kind);
                if (fits && reads) {
                    s = slot_of[aggs[i].col];
                    if (s >= 0 && nullable && nullable_in_slot[s] >= 4) s = -1;
                    if (s < 0) { new_slot = true; fits = (int)nullable_in_slot.size() < kMaxCols; }
                }
                if (!fits) { flush(); continue; }  // (an empty batch always takes one reducer)
                if (reads) {
                    if (new_slot) { s = (int)nullable_in_slot.size(); nullable_in_slot.push_back(0); slot_of[aggs[i].col] = s; }
                    if (nullable) nullable_in_slot[s]++;
                }
                cur.push_back(i);
                break;
            }
        }
        flush();
    }
    for (const std::vector<int> &at : batches) {
        if ((int)at.size() == naggs) {  // the usual case: the whole call in one launch
            AggJob job;
            BG_TRY(job_build(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, wid_base, W, wid_base == 0, &job));
            BG_TRY(job_run(c, &job, aggs, naggs, long_windows, kernel_ms, true, &plan, true));
            continue;
        }
        std::vector<bowgpu_agg> part;
        std::vector<bowgpu_out> part_outs;
        for (int i : at) { part.push_back(aggs[i]); part_outs.push_back(outs[i]); }
        int64_t lw = 0;
        double ms = 0;
        AggJob job;
        BG_TRY(job_build(c, cols, ncols, ts_col, plan, inclusive, part.data(), (int32_t)part.size(), part_outs.data(), wid_base, W, wid_base == 0, &job));
        BG_TRY(job_run(c, &job, part.data(), (int32_t)part.size(), &lw, &ms, true, &plan, true));
        for (size_t j = 0; j < at.size(); j++) outs[at[j]] = part_outs[j];
        if (long_windows && lw > *long_windows) *long_windows = lw;   // (the same windows in every batch)
        if (kernel_ms) *kernel_ms += ms;
    }
    if (n_mode > 0) return run_modes(c, cols, ncols, ts_col, plan, aggs, naggs, outs, inclusive, long_windows);
    return 0;
}

// ---- Rolling.Interpolate(...).Aggregate(...) in ONE pass over the rows (rolling_fused.hip), where the shape allows it.
// *done = false: the call is outside the fused kernel's domain (or some tile of it was - the kernel said so): the caller makes the two
// calls through device temporaries instead (bowgpu_rolling_interpolate_aggregate).  The domain: exclusive windows, the reducers of
// rolling_simple.hip, an interval column without nulls interpolated by WindowStart, value columns under Linear / StepPrevious / None /
// WindowStart, a frame that starts at or above 0 and spans less than 2^32 from its first window, windows of 4 .. 128 rows on average.
static int fused_try(Ctx *c, const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, const bowgpu_options &o, int inclusive,
                     const bowgpu_interp *interps, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, double *kernel_ms, bool *done) {
    *done = false;
    const bowgpu_col *tsc = &cols[ts_col];
    const int64_t n = tsc->length, W = plan.W;
    if (route_mask() & (BOWGPU_ROUTE_NO_FUSED | BOWGPU_ROUTE_NO_SIMPLE | BOWGPU_ROUTE_FORCE_GENERAL)) return 0;
    if (n <= 0 || W <= 0 || inclusive || o.inclusive || naggs > kSimpleMaxAggs) return 0;
    if (tsc->validity && tsc->null_count != 0) return 0;                      // (an interval column with nulls: extras.cpp interp_null_ts)
    const int64_t lim53 = 1ll << 53;
    if (plan.first_ts < plan.s0 || plan.last_ts >= lim53 || plan.s0 <= -lim53) return 0;   // rows below s0; float64(ts) inexact
    // a window that starts at -1, the reference's "no first value" sentinel (interpolation.go:99-105): such a window never gets a synthetic row
    if (plan.s0 <= -1 && (uint64_t)(-1 - plan.s0) % (uint64_t)plan.interval == 0 && (int64_t)((uint64_t)(-1 - plan.s0) / (uint64_t)plan.interval) < W) return 0;
    if (((uint64_t)plan.interval >> 32) != 0 || (uint64_t)plan.last_ts - (uint64_t)plan.s0 >= 0xFFFFFFF0ull || W >= 0xFFFFFFF0ll) return 0;
    if (n / W < 4 || n / W > 128) return 0;
    for (int i = 0; i < ncols; i++) {
        const int k = interps[i].kind;
        if (k != BOWGPU_INTERP_WINDOW_START && k != BOWGPU_INTERP_LINEAR && k != BOWGPU_INTERP_STEP_PREVIOUS && k != BOWGPU_INTERP_NONE) return 0;
    }
    if (interps[ts_col].kind != BOWGPU_INTERP_WINDOW_START) return 0;           // anything else does not keep the window grid
    {
        std::vector<int> used(ncols, 0);
        int distinct = 0;
        for (int i = 0; i < naggs; i++) {
            const int k = aggs[i].kind;
            if ((k >= BOWGPU_AGG_INTEGRAL_STEP && k <= BOWGPU_AGG_WAVG_LINEAR) || k == BOWGPU_AGG_MODE) return 0;
            if (kind_reads_values(k) && !used[aggs[i].col]++) distinct++;
        }
        if (distinct > kMaxCols) return 0;
    }
    AggJob job;
    BG_TRY(job_build(c, cols, ncols, ts_col, plan, 0, aggs, naggs, outs, 0, W, true, &job, kMaxAggs));   // (one pass per column whatever its reducers)
    AggParams &P = job.P;
    int need = 0;
    bool is_int = false, has_nulls = false, wide = false;
    if (P.pre_rows || !simple_applies(&job, aggs, naggs, plan, false, &need, &is_int, &has_nulls, &wide) || wide) return 0;
    P.bits_preset = 1;
    {
        BitmapBatch b;
        job_bitmaps(&job, aggs, naggs, true, &b);
        BG_TRY(launch_preset_bitmaps(c, b));
        job.counts_used = false;
    }
    FusedParams F;
    memset(&F, 0, sizeof F);
    simple_params_build(&job, aggs, naggs, false, &F.s);
    for (int s = 0; s < kMaxCols; s++) { F.cols[s].kind = BOWGPU_INTERP_NONE; F.cols[s].type = BOWGPU_INT64; }
    for (int i = 0; i < naggs; i++) {
        if (!kind_reads_values(aggs[i].kind) || P.aggs[i].slot < 0) continue;
        const bowgpu_interp &ip = interps[aggs[i].col];
        FusedCol &fc = F.cols[P.aggs[i].slot];
        fc.type = cols[aggs[i].col].type; fc.kind = ip.kind;
        fc.has_prev = ip.has_prev_row; fc.prev_t_valid = ip.prev_t_valid; fc.prev_v_valid = ip.prev_v_valid;
        fc.const_value = ip.const_value; fc.prev_t = ip.prev_t; fc.prev_v = ip.prev_v; fc.prev_v_i64 = ip.prev_v_i64;
    }
    F.inv_interval = (1.0 / (double)plan.interval) * (1.0 + 0x1.0p-40);
    BG_HIP(hipEventRecord(c->ev0, c->stream));
    BG_TRY(launch_rolling_fused(c, F, need, has_nulls));
    BG_HIP(hipEventRecord(c->ev1, c->stream));
    uint32_t *hstat;
    uint64_t *hcnt = nullptr;
    BG_TRY(job_enqueue_tail(c, &job, aggs, naggs));
    BG_TRY(job_readback(c, &job, &hstat, &hcnt));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (hstat[0]) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending: outside the device path");
    if (hstat[4] || hstat[5]) return 0;   // a tile the fused kernel cannot describe (too many heads, a long window, a far neighbour point)
    for (int i = 0; i < naggs; i++)
        job.douts[i].user->null_count = !kind_never_nil(aggs[i].kind) ? W - (int64_t)hcnt[i] : 0;
    float ms = 0;
    BG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_kernel_ms = ms;
    if (kernel_ms) *kernel_ms = ms;
    *done = true;
    return 0;
}

// ---- the pass of a sharded call put in flight BEFORE the exchange (bowgpu_shard_pass_begin): one per thread
struct PendingPass {
    AggJob job;
    Plan plan;
    PassState ps;
    std::vector<bowgpu_agg> aggs;
    std::vector<const void *> col_values, out_values;
    int64_t n = 0, interval = 0, raw_offset = 0, base = 0;
    int32_t ts_col = 0, inclusive = 0;
};
// (declared after g_bufs and g_ctx_holder: thread-local objects of one translation unit are destroyed in reverse declaration order, so
// at thread exit this runs while the thread's block cache and context are still alive; at process exit the pass is left to the runtime)
struct PendingHolder {
    PendingPass *p = nullptr;
    ~PendingHolder();
};
static thread_local PendingHolder g_pending_holder;
#define g_pending (g_pending_holder.p)

static bool pending_exists() { return g_pending != nullptr; }
static void pending_drop(Ctx *c) {
    if (!g_pending) return;
    if (c && c->inited) (void)hipStreamSynchronize(c->stream);   // its kernels are done with the scratch blocks before anyone reuses them
    delete g_pending;
    g_pending = nullptr;
}
PendingHolder::~PendingHolder() {
    // (this object is destroyed before the thread's block cache and context: what holds device blocks is released here)
    if (!g_process_exiting && g_ctx.inited) { (void)hipSetDevice(g_ctx.device); ctx_drop_null_ts_cache(&g_ctx); }
    if (!p || g_process_exiting) return;
    Ctx *c = &g_ctx;
    if (c->inited) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    delete p;
    p = nullptr;
}

}  // namespace bowgpu

using namespace bowgpu;

// ====================================================================================== C ABI
extern "C" {

int bowgpu_abi_version(void) { return BOWGPU_ABI_VERSION; }

int bowgpu_debug_set_route(uint32_t mask) {
    if (mask & ~(uint32_t)BOWGPU_ROUTE__ALL) return fail(BOWGPU_ERR_ARG, "unknown route bits 0x%x", mask & ~(uint32_t)BOWGPU_ROUTE__ALL);
    g_route = mask;
    return 0;
}
int bowgpu_debug_get_route(uint32_t *mask) {
    if (!mask) return fail(BOWGPU_ERR_ARG, "null argument");
    *mask = g_route;
    return 0;
}

const char *bowgpu_last_error(void) { return g_err; }

int bowgpu_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return 0;
}

int bowgpu_set_device(int device) {
    Ctx *c = &g_ctx;
    if (c->inited && c->device != device) {
        // drop per-device state of the old device
        (void)hipSetDevice(c->device);
        ctx_drop_null_ts_cache(c);   // (its blocks go back to the cache that is dropped next)
        devbuf_cache_drop();
        ctx_release(c);
    }
    c->device = device;
    Ctx *cc;
    return ctx_get(&cc);
}

int bowgpu_trim(int32_t all_threads, int64_t *bytes_freed) {
    // cached scratch blocks back to the device: the calling thread's, or every thread's (an idle service about to hand the GPU
    // to someone else; a host that saw BOWGPU_ERR_OOM from another library)
    Ctx *c;
    BG_TRY(ctx_get(&c));
    ctx_drop_null_ts_cache(c);
    const size_t freed = all_threads ? trim_all_threads() : g_bufs.drop_all();
    if (bytes_freed) *bytes_freed = (int64_t)freed;
    return 0;
}

// The range is registered as given.  A range that an earlier registration already covers is fine; one it covers only partly is
// refused (two small buffers inside one page can do that: register the enclosing allocation instead).
static bool host_range_mapped(const void *p) {
    void *d = nullptr;
    const bool ok = hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) == hipSuccess && d;
    if (!ok) (void)hipGetLastError();
    return ok;
}

int bowgpu_host_register(void *ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(BOWGPU_ERR_ARG, "null buffer / non-positive size");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();
    if (e == hipErrorHostMemoryAlreadyRegistered) {
        if (host_range_mapped(ptr) && host_range_mapped(reinterpret_cast<char *>(ptr) + bytes - 1)) return 0;
        return fail(BOWGPU_ERR_ARG, "bowgpu_host_register: the range overlaps an earlier registration only partly");
    }
    return hip_fail(e, "hipHostRegister");
}

int bowgpu_host_unregister(void *ptr) {
    if (!ptr) return fail(BOWGPU_ERR_ARG, "null buffer");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_HIP(hipStreamSynchronize(c->stream));   // nothing of this thread is still reading it
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {   // (a range that an earlier, still live registration covers stays with its owner)
        (void)hipGetLastError();
        if (e != hipErrorHostMemoryNotRegistered && e != hipErrorInvalidValue) return hip_fail(e, "hipHostUnregister");
    }
    return 0;
}

int bowgpu_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    size_t f = 0, t = 0;
    BG_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}

int bowgpu_device_name(char *buf, int cap) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    hipDeviceProp_t prop;
    BG_HIP(hipGetDeviceProperties(&prop, c->device));
    snprintf(buf, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int bowgpu_set_stream(void *hip_stream) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return 0;
}

int bowgpu_synchronize(void) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

const char *bowgpu_last_kernel_name(void) {
    Ctx *c;
    if (ctx_get(&c) != 0) return "";
    return c->last_kernel_name;
}

int bowgpu_last_call_slow_rows(int64_t *rows) {
    if (!rows) return fail(BOWGPU_ERR_ARG, "null argument");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    *rows = c->last_slow_rows;
    return 0;
}

int bowgpu_last_kernel_ms(double *ms) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    *ms = c->last_kernel_ms;
    return 0;
}

int bowgpu_malloc(void **ptr, int64_t bytes) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (bytes < 0) return fail(BOWGPU_ERR_ARG, "negative size");
    *ptr = nullptr;
    BG_HIP(hipMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1)));
    return 0;
}

int bowgpu_free(void *ptr) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    device_write_epoch_bump();   // (the address may come back from the allocator holding other data)
    if (ptr) BG_HIP(hipFree(ptr));
    return 0;
}

int bowgpu_memcpy_h2d(void *dst, const void *src, int64_t bytes) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (bytes > 0) {
        device_write_epoch_bump();
        BG_TRY(copy_h2d(c, dst, src, (size_t)bytes));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int bowgpu_memcpy_d2h(void *dst, const void *src, int64_t bytes) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (bytes > 0) {
        BG_TRY(copy_d2h(c, dst, src, (size_t)bytes));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int bowgpu_memset(void *dst, int value, int64_t bytes) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (bytes > 0) { device_write_epoch_bump(); BG_HIP(hipMemsetAsync(dst, value, (size_t)bytes, c->stream)); }
    return 0;
}

struct Timer {
    hipEvent_t a, b;
};

int bowgpu_timer_create(void **timer) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    Timer *t = new Timer();
    BG_HIP(hipEventCreate(&t->a));
    BG_HIP(hipEventCreate(&t->b));
    *timer = t;
    return 0;
}
int bowgpu_timer_start(void *timer) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_HIP(hipEventRecord(reinterpret_cast<Timer *>(timer)->a, c->stream));
    return 0;
}
int bowgpu_timer_stop(void *timer) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    BG_HIP(hipEventRecord(reinterpret_cast<Timer *>(timer)->b, c->stream));
    return 0;
}
int bowgpu_timer_elapsed_ms(void *timer, double *ms) {
    Timer *t = reinterpret_cast<Timer *>(timer);
    BG_HIP(hipEventSynchronize(t->b));
    float f = 0;
    BG_HIP(hipEventElapsedTime(&f, t->a, t->b));
    *ms = f;
    return 0;
}
int bowgpu_timer_destroy(void *timer) {
    Timer *t = reinterpret_cast<Timer *>(timer);
    if (!t) return 0;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return 0;
}

int bowgpu_enforce_interval_and_offset(int64_t interval, int64_t offset, int64_t *offset_out) {
    return enforce_interval_and_offset(interval, offset, offset_out);
}

int bowgpu_plan_windows(const bowgpu_col *ts, int64_t interval, int64_t offset, int64_t *s0, int64_t *num_windows) {
    if (!ts || !s0 || !num_windows) return fail(BOWGPU_ERR_ARG, "null argument");
    // O(1) host arithmetic on ts[0] and the last valid ts; the GPU is touched only to fetch
    // those scalars when the column lives in HBM (plan_make acquires the context lazily).
    Plan p;
    BG_TRY(plan_make(nullptr, ts, interval, offset, &p));
    *s0 = p.s0;
    *num_windows = p.W;
    return 0;
}

static int aggregate_with_plan(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const Plan &plan, int opt_inclusive,
                               const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, bowgpu_agg_info *info) {
    int inclusive = opt_inclusive ? 1 : 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    if (!outs) return fail(BOWGPU_ERR_ARG, "no output columns");
    {   // bowgpu_set_devices: the call cut into row ranges over the listed devices (multi.cpp); not taken -> the one-device path below
        bool fanned = false;
        BG_TRY(multi_aggregate(cols, ncols, ts_col, plan, opt_inclusive, g_strict_order, aggs, naggs, outs, info, &fanned));
        if (fanned) return 0;
    }
    Ctx *c;
    BG_TRY(ctx_get(&c));
    c->last_slow_rows = 0;
    int64_t n_long = 0;
    double ms = 0;
    static const bool prof = [] { const char *e = getenv("BOWGPU_CALL_PROFILE"); return e && e[0] == '1'; }();
    const double t_in = prof ? now_us() : 0;
    BG_TRY(run_aggregate(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, 0, plan.W, &n_long, &ms));
    if (prof) {
        static thread_local double acc[3] = {0, 0, 0};
        static thread_local int calls = 0;
        const double t_out = now_us();
        // (a route with no single synchronisation point of its own - the long-window forms - leaves the stamps of an EARLIER call behind:
        // such a call is booked whole under "enqueue" instead of producing negative parts)
        if (g_prof_sync_begin >= t_in) { acc[0] += g_prof_sync_begin - t_in; acc[1] += g_prof_sync_end - g_prof_sync_begin; acc[2] += t_out - g_prof_sync_end; }
        else acc[0] += t_out - t_in;
        if (++calls == 200) {
            fprintf(stderr, "bowgpu call profile (200 calls): enqueue %.1f us, synchronise %.1f us, after %.1f us\n", acc[0] / 200, acc[1] / 200, acc[2] / 200);
            acc[0] = acc[1] = acc[2] = 0; calls = 0;
        }
    }
    if (info) {
        info->s0 = plan.s0;
        info->num_windows = plan.W;
        info->new_interval_col = nic;
        info->inclusive = inclusive;
        info->long_windows = n_long;
        info->kernel_ms = ms;
    }
    return 0;
}

int bowgpu_rolling_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                             const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs,
                             bowgpu_out *outs, bowgpu_agg_info *info) {
    if (!cols || ncols <= 0) return fail(BOWGPU_ERR_ARG, "no columns");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    // reference order: the Rolling exists first (newIntervalRolling errors), then Aggregate validates
    Plan plan;
    BG_TRY(plan_make(nullptr, &cols[ts_col], interval, o.offset, &plan));
    g_strict_order = o.strict_order != 0 || (route_mask() & BOWGPU_ROUTE_STRICT_ORDER) != 0;
    const int rc = aggregate_with_plan(cols, ncols, ts_col, plan, o.inclusive, aggs, naggs, outs, info);
    g_strict_order = false;
    return rc;
}

/* Rolling.Interpolate(interps...) followed by Rolling.Aggregate(aggs...) on the Rolling it returns (reference
 * rolling/interpolation.go:30-69, rolling/aggregation.go:123-145), without handing the interpolated frame back: one pass over the
 * rows where the shape allows it (rolling_fused.hip), else the two calls through device temporaries - the same bits either way. */
int bowgpu_rolling_interpolate_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                                         const bowgpu_interp *interps, int32_t ninterps, const bowgpu_agg *aggs, int32_t naggs,
                                         bowgpu_out *outs, bowgpu_agg_info *info) {
    if (!cols || ncols <= 0) return fail(BOWGPU_ERR_ARG, "no columns");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    if (!interps) return fail(BOWGPU_ERR_ARG, "null argument");
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    // the reference's order: the Rolling exists first (newIntervalRolling), Interpolate validates its interpolators, then Aggregate
    // validates its aggregators against the interpolated Bow - which has the input's columns and types (interpolation.go:139-155)
    Plan plan;
    BG_TRY(plan_make(nullptr, &cols[ts_col], interval, o.offset, &plan));
    BG_TRY(interp_validate(cols, ncols, ts_col, &o, interps, ninterps));
    int inclusive = o.inclusive ? 1 : 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    if (!outs) return fail(BOWGPU_ERR_ARG, "no output columns");
    {   // bowgpu_set_devices: every rank interpolates and aggregates its own row range (multi.cpp); not taken -> one device, below
        bool fanned = false;
        BG_TRY(multi_interpolate_aggregate(cols, ncols, ts_col, plan, o.inclusive, o.strict_order != 0 || (route_mask() & BOWGPU_ROUTE_STRICT_ORDER) != 0, interps, ninterps, aggs, naggs, outs, info, &fanned));
        if (fanned) return 0;
    }
    Ctx *c;
    BG_TRY(ctx_get(&c));
    c->last_slow_rows = 0;   // (bowgpu_last_call_slow_rows speaks of THIS call; the two-call form below resets it again in its own entry points)
    bool done = false;
    double ms = 0;
    BG_TRY(fused_try(c, cols, ncols, ts_col, plan, o, inclusive, interps, aggs, naggs, outs, &ms, &done));
    if (done) {
        if (info) {
            info->s0 = plan.s0; info->num_windows = plan.W; info->new_interval_col = nic; info->inclusive = inclusive;
            info->long_windows = 0; info->kernel_ms = ms;
        }
        return 0;
    }
    // the two calls.  The interpolated frame lives in device temporaries and is aggregated where it lies.
    int64_t n_out = 0;
    BG_TRY(bowgpu_rolling_interpolate_count(cols, ncols, ts_col, interval, &o, interps, ninterps, &n_out));
    std::vector<DevBuf> vals(ncols), bits(ncols);
    std::vector<bowgpu_out> mid(ncols);
    std::vector<bowgpu_col> icols(ncols);
    for (int i = 0; i < ncols; i++) {
        BG_TRY(vals[i].alloc((size_t)n_out * 8 + 16));
        BG_TRY(bits[i].alloc((size_t)((n_out + 7) >> 3) + 16));
        memset(&mid[i], 0, sizeof mid[i]);
        mid[i].values = vals[i].p; mid[i].validity = reinterpret_cast<uint8_t *>(bits[i].p);
        mid[i].length = n_out; mid[i].residency = BOWGPU_DEVICE;
    }
    if (n_out > 0) BG_TRY(bowgpu_rolling_interpolate_fill(cols, ncols, ts_col, interval, &o, interps, ninterps, mid.data()));
    for (int i = 0; i < ncols; i++) {
        memset(&icols[i], 0, sizeof icols[i]);
        icols[i].values = mid[i].values; icols[i].validity = mid[i].validity;
        icols[i].offset = 0; icols[i].length = n_out; icols[i].null_count = n_out > 0 ? mid[i].null_count : 0;
        icols[i].type = cols[i].type; icols[i].residency = BOWGPU_DEVICE;
    }
    const int rc = bowgpu_rolling_aggregate(icols.data(), ncols, ts_col, interval, &o, aggs, naggs, outs, info);
    // (the temporaries go back to the block cache: every entry point above has synchronised the stream)
    return rc;
}

int bowgpu_plan_windows_ex(const bowgpu_col *ts, int64_t interval, int64_t offset, bowgpu_plan *out) {
    if (!ts || !out) return fail(BOWGPU_ERR_ARG, "null argument");
    Plan p;
    BG_TRY(plan_make(nullptr, ts, interval, offset, &p));
    out->s0 = p.s0; out->num_windows = p.W; out->first_ts = p.first_ts; out->last_ts = p.last_ts;
    out->interval = p.interval; out->offset = p.offset; out->nrows = ts->length;
    return 0;
}

int bowgpu_rolling_aggregate_planned(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_plan *pl,
                                     const bowgpu_options *opts, const bowgpu_agg *aggs, int32_t naggs,
                                     bowgpu_out *outs, bowgpu_agg_info *info) {
    if (!cols || ncols <= 0 || !pl) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    const bowgpu_col *ts = &cols[ts_col];
    if (ts->type != BOWGPU_INT64) return fail(BOWGPU_ERR_TS_TYPE, "impossible to create a new intervalRolling on column of type float64");
    if (pl->interval <= 0) return fail(BOWGPU_ERR_INTERVAL, "strictly positive interval required");
    if (pl->nrows != ts->length || pl->offset < 0 || pl->offset >= pl->interval || pl->num_windows < 0)
        return fail(BOWGPU_ERR_ARG, "the plan was not made for this interval column");
    // the plan against itself (host arithmetic) ... and against the column: the pass compares the two timestamps it was made from with
    // the column's first and last row (one tiny launch in front of the tile kernel, no round trip) - a plan made for another
    // column of the same length, or for this buffer before it was refilled, is BOWGPU_ERR_ARG instead of silently wrong routes
    if (ts->length > 0) {
        const int64_t s0 = first_window_start(pl->first_ts, pl->interval, pl->offset);
        const int64_t W = s0 > pl->last_ts ? 0 : (int64_t)(((uint64_t)pl->last_ts - (uint64_t)s0) / (uint64_t)pl->interval) + 1;
        if (pl->last_ts < pl->first_ts || s0 != pl->s0 || W != pl->num_windows)
            return fail(BOWGPU_ERR_ARG, "the plan is not consistent (s0 / num_windows do not follow from its first / last timestamp)");
    }
    Plan plan;
    plan.interval = pl->interval; plan.offset = pl->offset; plan.s0 = pl->s0; plan.W = pl->num_windows;
    plan.first_ts = pl->first_ts; plan.last_ts = pl->last_ts;
    plan.magic = magic_make((uint64_t)pl->interval);
    g_plan_from_caller = ts->length > 0;
    g_strict_order = (opts && opts->strict_order != 0) || (route_mask() & BOWGPU_ROUTE_STRICT_ORDER) != 0;
    const int rc = aggregate_with_plan(cols, ncols, ts_col, plan, opts ? opts->inclusive : 0, aggs, naggs, outs, info);
    g_plan_from_caller = false;
    g_strict_order = false;
    return rc;
}

// ---- entry points implemented in extras.cpp: window_bounds, aggregate_whole, interpolate,
// fill_linear, is_col_sorted, shard_* ----


// ---- row-range sharding -------------------------------------------------------------------

static bool strict_wanted(const bowgpu_options *o) { return (o && o->strict_order) || (route_mask() & BOWGPU_ROUTE_STRICT_ORDER); }
struct StrictScope { bool was; explicit StrictScope(bool on) : was(g_strict_order) { g_strict_order = on; } ~StrictScope() { g_strict_order = was; } };
// allow_strict: the record protocol (bowgpu_shard_begin / _pass_begin / _finish) serves bowgpu_options.strict_order since round 5 - a
// rank's own windows by the unsharded forms, a window shared by TWO ranks by the right rank's re-walk of its rows seeded with the left
// rank's running state (row order across the boundary); a window spread over three or more ranks merges partial sums and is declined
// by _finish.  The building blocks of round 1 (bowgpu_shard_aggregate, _carry_only) still decline it.
static int shard_check(const bowgpu_col *cols, int32_t ncols, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                       const bowgpu_options *o, bool allow_strict = false) {
    if (!allow_strict && strict_wanted(o)) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: strict_order is offered by the record protocol only (bowgpu_shard_begin / _pass_begin / _finish)");
    // (allow_strict: the record protocol.  It also takes HOST-resident columns and outputs since round 5 - staged through HBM per call the way
    // the unsharded entry points stage them: the pass put in flight by _pass_begin keeps its staged copies until _finish collects it)
    for (int i = 0; i < naggs; i++) {
        // a window cut by a shard boundary needs all its rows in one place: Mode has no constant-size partial state
        if (aggs[i].kind == BOWGPU_AGG_MODE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: Mode is not a mergeable reducer");
        if (!allow_strict && outs[i].residency != BOWGPU_DEVICE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: outputs must be device-resident");
    }
    for (int i = 0; i < ncols; i++)
        if (!allow_strict && cols[i].residency != BOWGPU_DEVICE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: columns must be device-resident");
    if (naggs > BOWGPU_CARRY_MAX_AGGS) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: too many aggregations");
    return 0;
}

// the plan of a shard: global s0, local first/last ts -> local window range
static int shard_plan(Ctx *c, const bowgpu_col *ts, int64_t interval, int64_t raw_offset, int64_t global_s0, Plan *p,
                      int64_t *wid_first, int64_t *wid_last) {
    BG_TRY(plan_make(c, ts, interval, raw_offset, p));  // validates type / interval / first ts; local s0, W are replaced below
    p->s0 = global_s0;
    *wid_first = -1;
    *wid_last = -1;
    p->W = 0;
    if (ts->length == 0) return 0;
    if (p->last_ts < global_s0) return 0;
    const int64_t f = p->first_ts < global_s0 ? global_s0 : p->first_ts;
    *wid_first = (int64_t)(((uint64_t)f - (uint64_t)global_s0) / (uint64_t)interval);
    *wid_last = (int64_t)(((uint64_t)p->last_ts - (uint64_t)global_s0) / (uint64_t)interval);
    p->W = *wid_last - *wid_first + 1;
    return 0;
}

int bowgpu_shard_aggregate(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                           const bowgpu_options *opts, int64_t global_s0, int32_t holds_global_row0,
                           int64_t lead, const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs,
                           bowgpu_shard_carry *carry, const bowgpu_next_row *next_row, int32_t finish_last) {
    if (!cols || ncols <= 0 || !carry || !outs) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    BG_TRY(shard_check(cols, ncols, aggs, naggs, outs, &o));
    Ctx *c;
    BG_TRY(ctx_get(&c));
    Plan plan;
    int64_t wf, wl;
    BG_TRY(shard_plan(c, &cols[ts_col], interval, o.offset, global_s0, &plan, &wf, &wl));
    memset(carry, 0, sizeof *carry);
    carry->first_window_id = wf;
    carry->last_window_id = wl;
    carry->first_ts = plan.first_ts;
    carry->last_ts = plan.last_ts;
    carry->nrows = cols[ts_col].length;
    carry->naggs = naggs;
    AggJob job;
    if (lead < 0 || (wf < 0 && lead != 0) || (wf >= 0 && lead > wf)) return fail(BOWGPU_ERR_ARG, "bad lead_empty_windows %lld", (long long)lead);
    const int64_t Wtot = plan.W + lead;
    BG_TRY(job_build(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, wf < 0 ? 0 : wf - lead, Wtot, holds_global_row0 != 0, &job));
    BG_TRY(job_run(c, &job, aggs, naggs, nullptr, nullptr, false, &plan));
    if (lead > 0) BG_TRY(launch_fill_empty(c, job.P, 0, lead));
    if (plan.W > 0) {
        // running state of the last window over this shard's rows; when the shard owns that window and the windows are
        // inclusive, its outputs are rewritten with the next shard's first row folded in where it is the inclusive row
        void *pool;
        BG_TRY(ctx_pool(c, kPoolShard, 16384, &pool));
        bowgpu_carry_state *dst = reinterpret_cast<bowgpu_carry_state *>(pool);
        bowgpu_next_row *dnext = nullptr;
        const bool finish = inclusive && finish_last && next_row && next_row->present;
        if (finish) {
            dnext = reinterpret_cast<bowgpu_next_row *>(reinterpret_cast<char *>(pool) + 8192);
            BG_HIP(hipMemcpyAsync(dnext, next_row, sizeof *next_row, hipMemcpyHostToDevice, c->stream));
        }
        BG_TRY(launch_range_state(c, job.P, finish ? 2 : 0, (uint64_t)wl, nullptr, dst, dnext, 0));
        BG_HIP(hipMemcpyAsync(carry->last, dst, sizeof(bowgpu_carry_state) * naggs, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    BG_TRY(job_finish(c, &job, aggs, naggs));
    return 0;
}

int bowgpu_shard_carry_only(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                            int64_t global_s0, int32_t holds_global_row0, const bowgpu_agg *aggs, int32_t naggs,
                            bowgpu_shard_carry *carry) {
    if (!cols || ncols <= 0 || !carry) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    if (inclusive) return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_shard_carry_only: inclusive windows take their carry from bowgpu_shard_aggregate");
    if (naggs > BOWGPU_CARRY_MAX_AGGS) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: too many aggregations");
    for (int i = 0; i < naggs; i++)
        if (aggs[i].kind == BOWGPU_AGG_MODE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: Mode is not a mergeable reducer");
    for (int i = 0; i < ncols; i++)
        if (cols[i].residency != BOWGPU_DEVICE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: columns must be device-resident");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    Plan plan;
    int64_t wf, wl;
    BG_TRY(shard_plan(c, &cols[ts_col], interval, o.offset, global_s0, &plan, &wf, &wl));
    memset(carry, 0, sizeof *carry);
    carry->first_window_id = wf;
    carry->last_window_id = wl;
    carry->first_ts = plan.first_ts;
    carry->last_ts = plan.last_ts;
    carry->nrows = cols[ts_col].length;
    carry->naggs = naggs;
    if (plan.W <= 0) return 0;
    // the descriptor block of the shard's call, with output columns nobody writes (range_state mode 0 only reads rows)
    std::vector<bowgpu_out> no_outs(naggs);
    void *dummy;
    BG_TRY(ctx_pool(c, kPoolShard, 16384, &dummy));
    for (int i = 0; i < naggs; i++) {
        memset(&no_outs[i], 0, sizeof(bowgpu_out));
        no_outs[i].values = dummy; no_outs[i].validity = reinterpret_cast<uint8_t *>(dummy);
        no_outs[i].length = 0; no_outs[i].residency = BOWGPU_DEVICE;
    }
    AggJob job;
    BG_TRY(job_build(c, cols, ncols, ts_col, plan, 0, aggs, naggs, no_outs.data(), wf, 0, holds_global_row0 != 0, &job));
    bowgpu_carry_state *dst = reinterpret_cast<bowgpu_carry_state *>(dummy);
    BG_TRY(launch_range_state(c, job.P, 0, (uint64_t)wl, nullptr, dst, nullptr));
    BG_HIP(hipMemcpyAsync(carry->last, dst, sizeof(bowgpu_carry_state) * naggs, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bowgpu_shard_fix_first(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval,
                           const bowgpu_options *opts, int64_t global_s0, int64_t lead, const bowgpu_agg *aggs,
                           int32_t naggs, bowgpu_out *outs, int64_t first_window_id, const bowgpu_carry_state *seeds,
                           bowgpu_carry_state *merged_out, const bowgpu_next_row *next_row) {
    if (!cols || ncols <= 0 || !seeds || !outs) return fail(BOWGPU_ERR_ARG, "null argument");
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    BG_TRY(shard_check(cols, ncols, aggs, naggs, outs, &o));
    Ctx *c;
    BG_TRY(ctx_get(&c));
    Plan plan;
    int64_t wf, wl;
    BG_TRY(shard_plan(c, &cols[ts_col], interval, o.offset, global_s0, &plan, &wf, &wl));
    if (wf != first_window_id) return fail(BOWGPU_ERR_ARG, "first_window_id %lld does not match the shard (%lld)", (long long)first_window_id, (long long)wf);
    AggJob job;
    if (lead < 0 || lead > wf) return fail(BOWGPU_ERR_ARG, "bad lead_empty_windows %lld", (long long)lead);
    const int64_t Wtot = plan.W + lead;
    BG_TRY(job_build(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, wf - lead, Wtot, false, &job));
    // the caller's validity bytes are the truth for this second phase: bring them into the word-aligned working copy
    for (int i = 0; i < naggs; i++)
        BG_HIP(hipMemcpyAsync(job.douts[i].validity, outs[i].validity, (size_t)((Wtot + 7) >> 3), hipMemcpyDeviceToDevice, c->stream));
    void *pool;
    BG_TRY(ctx_pool(c, kPoolShard, 16384, &pool));
    bowgpu_carry_state *dseed = reinterpret_cast<bowgpu_carry_state *>(pool);
    bowgpu_carry_state *dout = reinterpret_cast<bowgpu_carry_state *>(reinterpret_cast<char *>(pool) + 4096);
    bowgpu_next_row *dnext = nullptr;
    if (inclusive && next_row && next_row->present) {
        dnext = reinterpret_cast<bowgpu_next_row *>(reinterpret_cast<char *>(pool) + 8192);
        BG_HIP(hipMemcpyAsync(dnext, next_row, sizeof *next_row, hipMemcpyHostToDevice, c->stream));
    }
    BG_HIP(hipMemcpyAsync(dseed, seeds, sizeof(bowgpu_carry_state) * naggs, hipMemcpyHostToDevice, c->stream));
    BG_TRY(launch_range_state(c, job.P, 1, (uint64_t)wf, dseed, dout, dnext));
    if (merged_out) {
        BG_HIP(hipMemcpyAsync(merged_out, dout, sizeof(bowgpu_carry_state) * naggs, hipMemcpyDeviceToHost, c->stream));
        BG_HIP(hipStreamSynchronize(c->stream));
    }
    BG_TRY(job_finish(c, &job, aggs, naggs));
    return 0;
}

int bowgpu_carry_merge(const bowgpu_carry_state *L, const bowgpu_carry_state *R, bowgpu_carry_state *out) {
    // concatenation of two row ranges of one window, left then right (same rules as the device stats_merge)
    if (!L || !R || !out) return fail(BOWGPU_ERR_ARG, "null argument");
    bowgpu_carry_state m = *L;
    m.nrows = L->nrows + R->nrows;
    if (R->has_value) {
        if (!L->has_value) {
            const int64_t nrows = m.nrows;
            m = *R;
            m.nrows = nrows;
        } else {
            m.sum = L->sum + R->sum;
            m.count = L->count + R->count;
            if (R->has_nn) {
                if (R->nn_min < m.vmin) m.vmin = R->nn_min;
                if (R->nn_max > m.vmax) m.vmax = R->nn_max;
                if (!m.has_nn) { m.nn_min = R->nn_min; m.nn_max = R->nn_max; m.has_nn = 1; }
                else { if (R->nn_min < m.nn_min) m.nn_min = R->nn_min; if (R->nn_max > m.nn_max) m.nn_max = R->nn_max; }
            }
            m.last_bits = R->last_bits;
        }
    }
    // the both-valid points of the time-weighted reducers (device stats_merge)
    m.pt = L->pt; m.pv = L->pv; m.first_pt = L->first_pt; m.first_pv = L->first_pv;
    m.integ_step = L->integ_step; m.integ_trap = L->integ_trap; m.has_point = L->has_point; m.has_pair = L->has_pair;
    if (R->has_point) {
        if (L->has_point) {
            m.integ_trap = L->integ_trap + (L->pv + R->first_pv) / 2 * (R->first_pt - L->pt) + R->integ_trap;
            m.integ_step = L->integ_step + L->pv * (R->first_pt - L->pt) + R->integ_step;
            m.has_pair = 1;
        } else {
            m.first_pt = R->first_pt; m.first_pv = R->first_pv; m.integ_trap = R->integ_trap; m.integ_step = R->integ_step;
            m.has_pair = R->has_pair; m.has_point = 1;
        }
        m.pt = R->pt; m.pv = R->pv;
    }
    *out = m;
    return 0;
}

int bowgpu_shard_span(const bowgpu_col *ts, int64_t *first_ts, int64_t *last_ts, int64_t *nrows) {
    if (!ts || !first_ts || !last_ts || !nrows) return fail(BOWGPU_ERR_ARG, "null argument");
    *first_ts = 0; *last_ts = 0; *nrows = ts->length;
    if (ts->length == 0) return 0;
    Plan p;
    BG_TRY(plan_make(nullptr, ts, 1, 0, &p));  // (interval 1: only the two scalars matter; one synchronisation)
    *first_ts = p.first_ts;
    *last_ts = p.last_ts;
    return 0;
}

int bowgpu_shard_first_row(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, const bowgpu_agg *aggs, int32_t naggs,
                           bowgpu_next_row *out) {
    if (!cols || !aggs || !out || ncols <= 0) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    if (naggs > BOWGPU_CARRY_MAX_AGGS) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: too many aggregations");
    memset(out, 0, sizeof *out);
    if (cols[ts_col].length == 0) return 0;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    auto fetch = [&](const bowgpu_col &col, uint64_t *bits, int32_t *valid) -> int {
        if (col.residency != BOWGPU_DEVICE) {   // host memory (pageable or registered): read where it lies
            memcpy(bits, reinterpret_cast<const char *>(col.values) + 8 * col.offset, 8);
            *valid = (col.validity && col.null_count != 0) ? ((col.validity[col.offset >> 3] >> (col.offset & 7)) & 1) : 1;
            return 0;
        }
        BG_HIP(hipMemcpyAsync(bits, reinterpret_cast<const char *>(col.values) + 8 * col.offset, 8, hipMemcpyDeviceToHost, c->stream));
        *valid = 1;
        if (col.validity && col.null_count != 0) {
            uint8_t byte = 0;
            BG_HIP(hipMemcpyAsync(&byte, col.validity + (col.offset >> 3), 1, hipMemcpyDeviceToHost, c->stream));
            BG_HIP(hipStreamSynchronize(c->stream));
            *valid = (byte >> (col.offset & 7)) & 1;
        }
        return 0;
    };
    uint64_t tsb = 0;
    int32_t tv = 0;
    BG_TRY(fetch(cols[ts_col], &tsb, &tv));
    for (int i = 0; i < naggs; i++) {
        if (aggs[i].col < 0 || aggs[i].col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "aggregation %d: no column with index %d", i, aggs[i].col);
        BG_TRY(fetch(cols[aggs[i].col], &out->bits[i], &out->valid[i]));
    }
    BG_HIP(hipStreamSynchronize(c->stream));
    out->ts = (int64_t)tsb;
    out->present = 1;
    return 0;
}

// ---- the shard protocol: begin -> one exchange -> finish --------------------------------------------------------------

// largest point of the window grid {offset + k * interval} that is <= t; false when it is not an int64
static bool grid_floor(int64_t t, int64_t interval, int64_t offset_norm, int64_t *out) {
    const __int128 d = (__int128)t - offset_norm;
    const __int128 k = d >= 0 ? d / interval : -((-d + interval - 1) / interval);
    const __int128 b = k * interval + offset_norm;
    if (b < (__int128)INT64_MIN || b > (__int128)INT64_MAX) return false;
    *out = (int64_t)b;
    return true;
}

static int shard_cols_check(const bowgpu_col *cols, int32_t ncols, const bowgpu_agg *aggs, int32_t naggs) {
    (void)cols; (void)ncols;   // (any residency: bowgpu_shard_begin stages host-resident columns like every other entry point)
    for (int i = 0; i < naggs; i++)
        if (aggs[i].kind == BOWGPU_AGG_MODE) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: Mode is not a mergeable reducer");
    if (naggs > BOWGPU_CARRY_MAX_AGGS) return fail(BOWGPU_ERR_UNSUPPORTED, "sharded aggregate: too many aggregations");
    return 0;
}

int bowgpu_shard_begin(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                       const bowgpu_agg *aggs, int32_t naggs, const int64_t *global_s0, bowgpu_shard_record *rec) {
    if (!cols || ncols <= 0 || !rec || !aggs) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = o.inclusive ? 1 : 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    BG_TRY(shard_cols_check(cols, ncols, aggs, naggs));
    memset(rec, 0, sizeof *rec);
    rec->naggs = naggs;
    rec->flags = global_s0 ? 1 : 0;
    rec->nrows = cols[ts_col].length;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    Plan plan;
    BG_TRY(plan_make(c, &cols[ts_col], interval, o.offset, &plan));  // type / interval / first-ts checks; first and last ts
    if (rec->nrows == 0) return 0;
    rec->first_ts = plan.first_ts;
    rec->last_ts = plan.last_ts;
    // the window grid is {offset + k * interval} whatever the frame's first row is (rolling.go:95-99): the window holding this
    // rank's last row starts on it.  Ids here are relative to `base`, a grid point at or below the rank's first row.
    int64_t base;
    if (global_s0) base = *global_s0;
    else if (!grid_floor(plan.first_ts, interval, plan.offset, &base))
        return fail(BOWGPU_ERR_UNSUPPORTED, "interval column reaches below the int64 window grid");
    plan.s0 = base;
    const int64_t wl = plan.last_ts < base ? 0 : (int64_t)(((uint64_t)plan.last_ts - (uint64_t)base) / (uint64_t)interval);
    // (window 0 of the real grid also takes the rows below s0: range_state_kernel starts at row 0 for it)
    rec->carry_from_ts = wl == 0 ? INT64_MIN : (int64_t)((uint64_t)base + (uint64_t)wl * (uint64_t)interval);
    if (inclusive)
        BG_TRY(bowgpu_shard_first_row(cols, ncols, ts_col, aggs, naggs, &rec->first_row));
    std::vector<bowgpu_out> no_outs(naggs);
    void *dummy;
    BG_TRY(ctx_pool(c, kPoolShard, 16384, &dummy));
    for (int i = 0; i < naggs; i++) {
        memset(&no_outs[i], 0, sizeof(bowgpu_out));
        no_outs[i].values = dummy; no_outs[i].validity = reinterpret_cast<uint8_t *>(dummy);
        no_outs[i].length = 0; no_outs[i].residency = BOWGPU_DEVICE;
    }
    AggJob job;
    BG_TRY(job_build(c, cols, ncols, ts_col, plan, 0, aggs, naggs, no_outs.data(), wl, 0, false, &job));
    bowgpu_carry_state *dst = reinterpret_cast<bowgpu_carry_state *>(dummy);
    const bool strict = strict_wanted(&o);
    uint32_t far = 0;
    if (strict) BG_HIP(hipMemsetAsync(job.P.status + 7, 0, 4, c->stream));
    BG_TRY(launch_range_state(c, job.P, 0, (uint64_t)wl, nullptr, dst, nullptr, 1, strict ? 1 : 0));
    BG_HIP(hipMemcpyAsync(rec->last, dst, sizeof(bowgpu_carry_state) * naggs, hipMemcpyDeviceToHost, c->stream));
    if (strict) BG_HIP(hipMemcpyAsync(&far, job.P.status + 7, 4, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (far) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: the shard's last window holds more than 2^20 rows (one lane walks a window in row order: beyond that the call is declined)");
    return 0;
}

// The rank's pass put in flight BEFORE the exchange, from its own record alone: output slot 0 = the window of the rank's first
// row on the offset-aligned grid, i.e. what bowgpu_shard_finish will decide whenever no empty windows lie in front of the rank
// (lead_empty_windows == 0) and no row lies below the frame's first window start.  Enqueued only - the call returns while the
// kernel runs - and collected by the bowgpu_shard_finish that follows on this thread; if the gathered records decide otherwise
// (a gap to the left neighbour, the negative-timestamp corner) finish discards it and runs the pass the serial way.
int bowgpu_shard_pass_begin(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                            const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, const bowgpu_shard_record *me) {
    if (!cols || ncols <= 0 || !outs || !me || !aggs) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = o.inclusive ? 1 : 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    BG_TRY(shard_check(cols, ncols, aggs, naggs, outs, &o, true));
    if (cols[ts_col].type != BOWGPU_INT64) return fail(BOWGPU_ERR_TS_TYPE, "impossible to create a new intervalRolling on column of type float64");
    if (me->nrows != cols[ts_col].length) return fail(BOWGPU_ERR_ARG, "the record says %lld rows, the interval column has %lld",
                                                      (long long)me->nrows, (long long)cols[ts_col].length);
    StrictScope strict_scope(strict_wanted(&o));
    PendingOwnerScope owner;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    pending_drop(c);
    c->last_slow_rows = 0;
    // declined (not an error): nothing to reduce, or timestamps below zero, where the frame's first window start may lie ABOVE
    // a rank's rows (rolling.go:96-99) - that needs the records
    if (me->nrows == 0 || me->first_ts < 0 || me->last_ts < me->first_ts) return BOWGPU_SHARD_PASS_DECLINED;
    PendingPass *pp = new PendingPass();
    Plan &plan = pp->plan;
    plan.interval = interval;
    int rc = enforce_interval_and_offset(interval, o.offset, &plan.offset);
    int64_t base = 0;
    if (rc == 0 && !grid_floor(me->first_ts, interval, plan.offset, &base)) rc = BOWGPU_SHARD_PASS_DECLINED;
    if (rc != 0) { delete pp; return rc; }
    plan.magic = magic_make((uint64_t)interval);
    plan.s0 = base;                       // local numbering: slot k = window [base + k * interval, ...)
    plan.first_ts = me->first_ts;
    plan.last_ts = me->last_ts;
    plan.W = (int64_t)(((uint64_t)me->last_ts - (uint64_t)base) / (uint64_t)interval) + 1;
    pp->base = base;
    pp->n = me->nrows; pp->interval = interval; pp->raw_offset = o.offset; pp->ts_col = ts_col; pp->inclusive = inclusive;
    pp->aggs.assign(aggs, aggs + naggs);
    for (int i = 0; i < ncols; i++) pp->col_values.push_back(reinterpret_cast<const char *>(cols[i].values) + 8 * cols[i].offset);
    for (int i = 0; i < naggs; i++) pp->out_values.push_back(outs[i].values);
    rc = job_build(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, 0, plan.W, false, &pp->job);
    if (rc == 0) rc = job_pass_enqueue(c, &pp->job, aggs, naggs, &plan, false, &pp->ps);
    if (rc != 0) { (void)hipStreamSynchronize(c->stream); delete pp; return rc; }
    g_pending = pp;
    return 0;
}

int bowgpu_shard_plan(const bowgpu_shard_record *recs, int32_t world, int32_t rank, int64_t interval, int64_t raw_offset,
                      bowgpu_shard_decision *d) {
    if (!recs || !d || world <= 0 || rank < 0 || rank >= world) return fail(BOWGPU_ERR_ARG, "bad shard plan arguments");
    int64_t off;
    BG_TRY(enforce_interval_and_offset(interval, raw_offset, &off));
    memset(d, 0, sizeof *d);
    d->first_window_id = d->last_window_id = d->first_slot_window_id = -1;
    d->seed_first_rank = d->next_rank = -1;
    int g0 = -1, gl = -1;
    for (int q = 0; q < world; q++) {
        if (recs[q].nrows < 0) return fail(BOWGPU_ERR_ARG, "rank %d: negative row count", q);
        if (recs[q].nrows == 0) continue;
        if (recs[q].last_ts < recs[q].first_ts) return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending on rank %d", q);
        if (gl >= 0 && recs[gl].last_ts > recs[q].first_ts)
            return fail(BOWGPU_ERR_TS_UNSORTED, "interval column is not ascending across ranks %d and %d", gl, q);
        if (g0 < 0) g0 = q;
        gl = q;
    }
    if (g0 < 0) return 0;  // no rows anywhere: no windows (rolling.go:143-146)
    const int64_t s0 = first_window_start(recs[g0].first_ts, interval, off);
    d->s0 = s0;
    d->holds_global_row0 = rank == g0;
    const int64_t last = recs[gl].last_ts;
    if (s0 > last) return 0;   // countWindows: s0 beyond the last row (only rows below s0) => no windows (rolling.go:150-152)
    if (last > 0 && s0 < 0 && (uint64_t)last - (uint64_t)s0 > (uint64_t)INT64_MAX)
        return fail(BOWGPU_ERR_UNSUPPORTED, "interval column spans more than 2^63: int64 overflow in the reference's countWindows");
    d->num_windows = (int64_t)(((uint64_t)last - (uint64_t)s0) / (uint64_t)interval) + 1;
    // first / last window of a rank; rows below s0 (Go's truncating division on negative timestamps) ride in window 0
    auto wid_of = [&](int64_t t) -> int64_t { return t < s0 ? 0 : (int64_t)(((uint64_t)t - (uint64_t)s0) / (uint64_t)interval); };
    auto wf = [&](int q) { return recs[q].nrows == 0 ? (int64_t)-1 : wid_of(recs[q].first_ts); };
    auto wl = [&](int q) { return recs[q].nrows == 0 ? (int64_t)-1 : wid_of(recs[q].last_ts); };
    auto left_of = [&](int q) { int r = q - 1; while (r >= 0 && recs[r].nrows == 0) r--; return r; };
    auto right_of = [&](int q) { int r = q + 1; while (r < world && recs[r].nrows == 0) r++; return r < world ? r : -1; };
    // window 0 with rows below s0 split over ranks: the first-attempt states were cut on the local grid
    if (s0 > recs[g0].first_ts)
        for (int q = 0; q < world; q++) {
            if (recs[q].nrows == 0 || (recs[q].flags & 1)) continue;
            const int r = right_of(q);
            if (wl(q) == 0 && r >= 0 && wf(r) == 0 && recs[q].carry_from_ts > recs[q].first_ts) d->retry_with_s0 = 1;
        }
    if (recs[rank].nrows == 0) return 0;
    const int64_t f = wf(rank), l = wl(rank);
    d->first_window_id = f;
    d->last_window_id = l;
    const int lq = left_of(rank), rq = right_of(rank);
    d->next_rank = rq;
    // empty windows between the left neighbour's last window and this rank's first one are this rank's to output; with nothing
    // to the left there are none (global row 0 lies in window 0)
    d->lead_empty_windows = lq < 0 ? 0 : std::max<int64_t>(0, f - wl(lq) - 1);
    // ranks (ascending) holding earlier rows of this rank's FIRST window
    int sf = -1;
    for (int q = lq; q >= 0 && wl(q) == f; q = left_of(q)) {
        sf = q;
        if (wf(q) != f) break;   // q only contributes its tail
    }
    d->seed_first_rank = sf;
    d->drops_last = rq >= 0 && wf(rq) == l;
    d->first_slot_window_id = f - d->lead_empty_windows;
    d->windows_local = l - f + 1 + d->lead_empty_windows;
    d->windows_owned = d->windows_local - (d->drops_last ? 1 : 0);
    d->finish_last = !d->drops_last && !(f == l && sf >= 0);
    return 0;
}

int bowgpu_shard_finish(const bowgpu_col *cols, int32_t ncols, int32_t ts_col, int64_t interval, const bowgpu_options *opts,
                        const bowgpu_agg *aggs, int32_t naggs, bowgpu_out *outs, const bowgpu_shard_record *recs, int32_t world,
                        int32_t rank, bowgpu_shard_decision *decision, bowgpu_agg_info *info) {
    if (!cols || ncols <= 0 || !outs || !recs || !aggs) return fail(BOWGPU_ERR_ARG, "null argument");
    if (ts_col < 0 || ts_col >= ncols) return fail(BOWGPU_ERR_BAD_COL, "no interval column with index %d", ts_col);
    PendingOwnerScope owner;   // (the pass in flight is this call's to collect or drop)
    bowgpu_options o = {0, 0, 0};
    if (opts) o = *opts;
    int inclusive = o.inclusive ? 1 : 0, nic = -1;
    BG_TRY(validate_aggs(cols, ncols, ts_col, aggs, naggs, &inclusive, &nic));
    BG_TRY(shard_check(cols, ncols, aggs, naggs, outs, &o, true));
    const bool strict = strict_wanted(&o);
    StrictScope strict_scope(strict);
    bowgpu_shard_decision d;
    BG_TRY(bowgpu_shard_plan(recs, world, rank, interval, o.offset, &d));
    if (decision) *decision = d;
    if (info) {
        memset(info, 0, sizeof *info);
        info->s0 = d.s0; info->num_windows = d.num_windows; info->new_interval_col = nic; info->inclusive = inclusive;
    }
    if (d.retry_with_s0) {
        Ctx *cc;
        if (g_pending && ctx_get(&cc) == 0) pending_drop(cc);
        return BOWGPU_SHARD_RETRY;
    }
    const bowgpu_shard_record &me = recs[rank];
    if (me.nrows != cols[ts_col].length) return fail(BOWGPU_ERR_ARG, "record of rank %d says %lld rows, the interval column has %lld", rank,
                                                     (long long)me.nrows, (long long)cols[ts_col].length);
    if (cols[ts_col].type != BOWGPU_INT64) return fail(BOWGPU_ERR_TS_TYPE, "impossible to create a new intervalRolling on column of type float64");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (!g_pending) c->last_slow_rows = 0;   // (a pass in flight has booked its rows already: bowgpu_shard_pass_begin reset the counter)
    // the rank's plan from its own record: no round trip to the device for first / last ts
    Plan plan;
    plan.interval = interval;
    BG_TRY(enforce_interval_and_offset(interval, o.offset, &plan.offset));
    plan.magic = magic_make((uint64_t)interval);
    plan.s0 = d.s0;
    plan.first_ts = me.first_ts;
    plan.last_ts = me.last_ts;
    const int64_t wf = d.first_window_id, wl = d.last_window_id, lead = d.lead_empty_windows;
    plan.W = wf < 0 ? 0 : wl - wf + 1;
    const int64_t Wtot = plan.W + lead;
    AggJob job;
    const bool pre_rows_here = me.nrows > 0 && me.first_ts < d.s0;   // rows below s0 ride in window 0 (rolling.go:194-196)
    int64_t n_long = 0;
    double ms = 0;
    // the pass bowgpu_shard_pass_begin put in flight before the exchange, if the records decide what it assumed
    bool collected = false;
    if (g_pending) {
        PendingPass *pp = g_pending;
        bool same = pp->n == me.nrows && pp->interval == interval && pp->raw_offset == o.offset && pp->ts_col == ts_col &&
                    pp->inclusive == inclusive && (int32_t)pp->aggs.size() == naggs && (int32_t)pp->col_values.size() == ncols &&
                    memcmp(pp->aggs.data(), aggs, sizeof(bowgpu_agg) * (size_t)naggs) == 0;
        for (int i = 0; same && i < ncols; i++) same = pp->col_values[i] == reinterpret_cast<const char *>(cols[i].values) + 8 * cols[i].offset;
        for (int i = 0; same && i < naggs; i++) same = pp->out_values[i] == outs[i].values;
        const bool holds = same && wf >= 0 && lead == 0 && !pre_rows_here && pp->plan.W == plan.W &&
                           pp->base == (int64_t)((uint64_t)d.s0 + (uint64_t)wf * (uint64_t)interval);
        if (holds) {
            g_pending = nullptr;
            job = std::move(pp->job);
            const int rc = job_pass_complete(c, &job, aggs, naggs, &pp->plan, false, &pp->ps, &n_long, &ms);
            delete pp;
            if (rc != 0) return rc;
            // from the pass's local numbering (slot 0 starts at base) to the frame's: the stitch below speaks global window ids
            job.P.s0 = d.s0;
            job.P.wid_base = wf;
            for (int i = 0; i < naggs; i++) job.douts[i].user = &outs[i];   // (the descriptor array of THIS call receives length / null_count)
            collected = true;
        } else {
            pending_drop(c);
        }
    }
    if (!collected) {
        BG_TRY(job_build(c, cols, ncols, ts_col, plan, inclusive, aggs, naggs, outs, wf < 0 ? 0 : wf - lead, Wtot, pre_rows_here, &job));
        BG_TRY(job_run(c, &job, aggs, naggs, &n_long, &ms, false, &plan));
    }
    if (lead > 0) BG_TRY(launch_fill_empty(c, job.P, 0, lead));
    if (plan.W > 0) {
        void *pool;
        BG_TRY(ctx_pool(c, kPoolShard, 16384, &pool));
        bowgpu_carry_state *dseed = reinterpret_cast<bowgpu_carry_state *>(pool);
        bowgpu_next_row *dnext = nullptr;
        const bowgpu_next_row *next_row = d.next_rank >= 0 ? &recs[d.next_rank].first_row : nullptr;
        if (inclusive && next_row && next_row->present) {
            dnext = reinterpret_cast<bowgpu_next_row *>(reinterpret_cast<char *>(pool) + 8192);
            BG_HIP(hipMemcpyAsync(dnext, next_row, sizeof *next_row, hipMemcpyHostToDevice, c->stream));
        }
        // the rank owns its last window and only now knows the row that may close it (rolling.go:201-209)
        if (dnext && d.finish_last) BG_TRY(launch_range_state(c, job.P, 2, (uint64_t)wl, nullptr, nullptr, dnext, 0, strict ? 1 : 0));
        if (d.seed_first_rank >= 0) {
            // running state of this rank's first window over the rows the ranks to the left hold: one rank's state as is,
            // several merged in rank order (empty ranks in between hold zero states: identity)
            bowgpu_carry_state seeds[BOWGPU_CARRY_MAX_AGGS];
            // (window 0 stitched across shards is an empty slice unless one of its rows reaches s0: range_state_kernel's rule)
            int seed_alive = 0;
            for (int q = d.seed_first_rank; q < rank; q++)
                if (recs[q].nrows > 0 && recs[q].last_ts >= d.s0) seed_alive = 1;
            if (strict) {   // (a window spread over three or more ranks merges the middle ranks' partial sums: not row order)
                int with_rows = 0;
                for (int q = d.seed_first_rank; q < rank; q++) with_rows += recs[q].nrows > 0;
                if (with_rows > 1) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: a window is spread over three or more shards (its middle shards contribute partial sums)");
            }
            for (int a = 0; a < naggs; a++) {
                seeds[a] = recs[d.seed_first_rank].last[a];
                for (int q = d.seed_first_rank + 1; q < rank; q++) {
                    if (recs[q].nrows == 0) continue;
                    bowgpu_carry_state m;
                    BG_TRY(bowgpu_carry_merge(&seeds[a], &recs[q].last[a], &m));
                    seeds[a] = m;
                }
            }
            // (the records were built by the caller's all_gather: pageable memory, so stage through the pinned block)
            bowgpu_carry_state *hseed;
            static_assert(4096 + sizeof seeds <= 16384, "pinned block");
            BG_TRY(ctx_pinned(c, 16384, reinterpret_cast<void **>(&hseed)));
            hseed = reinterpret_cast<bowgpu_carry_state *>(reinterpret_cast<char *>(hseed) + 4096);
            memcpy(hseed, seeds, sizeof(bowgpu_carry_state) * naggs);
            BG_HIP(hipMemcpyAsync(dseed, hseed, sizeof(bowgpu_carry_state) * naggs, hipMemcpyHostToDevice, c->stream));
            // the window may also be the rank's last one: then the next rank's first row can be its inclusive row
            const bool also_last = wf == wl && !d.drops_last;
            BG_TRY(launch_range_state(c, job.P, 1, (uint64_t)wf, dseed, nullptr, also_last ? dnext : nullptr, seed_alive, strict ? 1 : 0));
        }
    }
    uint32_t too_long = 0;
    BG_TRY(job_finish(c, &job, aggs, naggs, strict ? &too_long : nullptr));
    if (too_long) return fail(BOWGPU_ERR_UNSUPPORTED, "strict_order: some window holds more than 2^20 rows (one lane walks a window in row order: beyond that the call is declined)");
    if (info) { info->long_windows = n_long; info->kernel_ms = ms; }
    return 0;
}

int bowgpu_gen_dense(int64_t row0, int64_t n, uint64_t seed, int64_t *ts_dev, double *val_dev) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    device_write_epoch_bump();
    return launch_gen_dense(c, row0, n, seed, ts_dev, val_dev);
}

int bowgpu_gen_sparse(int64_t row0, int64_t n, uint64_t seed, int64_t *ts_dev, double *val_dev, uint8_t *validity_dev) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (row0 & 7) return fail(BOWGPU_ERR_ARG, "row0 must be a multiple of 8");
    device_write_epoch_bump();
    return launch_gen_sparse(c, row0, n, seed, ts_dev, val_dev, validity_dev);
}

int bowgpu_stream_read_ceiling(const void *dev_a, const void *dev_b, int64_t bytes_each, double *gb_per_s) {
    if (!dev_a || !dev_b || !gb_per_s || bytes_each < (1 << 20)) return fail(BOWGPU_ERR_ARG, "two device buffers of at least 1 MiB each are needed");
    if ((reinterpret_cast<uintptr_t>(dev_a) | reinterpret_cast<uintptr_t>(dev_b)) & 15) return fail(BOWGPU_ERR_ARG, "buffers must be 16-byte aligned");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    void *d;
    BG_TRY(ctx_scratch(c, 4096, &d));
    uint64_t *dout = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(d) + 3072);
    double best = 0.0;
    const int shapes[][2] = {{0, 8}, {0, 16}, {0, 32}, {1, 0}};  // {mode, workgroups per CU}
    for (const auto &sh : shapes) {
        float ms = 0;
        BG_TRY(stream_sum_run(c, dev_a, dev_b, bytes_each, sh[0], sh[1], 5, dout, &ms));
        const double gbs = 2.0 * (double)(bytes_each / 16 * 16) / (ms * 1e-3) / 1e9;
        if (gbs > best) best = gbs;
    }
    *gb_per_s = best;
    return 0;
}

int bowgpu_stream_rw_probe(const void *dev_a, const void *dev_b, int64_t bytes_each, void *out_a, void *out_b, int64_t rows_per_slot,
                             double *read_gb_per_s, double *ms_out) {
    if (!dev_a || !dev_b || !out_a || !out_b || !read_gb_per_s || bytes_each < (1 << 20) || rows_per_slot <= 0)
        return fail(BOWGPU_ERR_ARG, "two device input buffers of at least 1 MiB each and two output buffers are needed");
    if ((reinterpret_cast<uintptr_t>(dev_a) | reinterpret_cast<uintptr_t>(dev_b)) & 15) return fail(BOWGPU_ERR_ARG, "buffers must be 16-byte aligned");
    Ctx *c;
    BG_TRY(ctx_get(&c));
    const int64_t rows = bytes_each / 4096 * 512;
    const int64_t nslots = (rows + rows_per_slot - 1) / rows_per_slot;
    double best = 0.0, best_ms = 0.0;
    for (int nt = 0; nt < 2; nt++) {
        float ms = 0;
        BG_TRY(stream_rw_run(c, dev_a, dev_b, bytes_each, out_a, out_b, rows_per_slot, nslots, nt != 0, 5, &ms));
        const double gbs = 2.0 * (double)rows * 8.0 / (ms * 1e-3) / 1e9;
        if (gbs > best) { best = gbs; best_ms = ms; }
    }
    *read_gb_per_s = best;
    if (ms_out) *ms_out = best_ms;
    return 0;
}

// diagnostic builds only (in-kernel stamps): words [first, first + n) of the calling thread's device status block
int bowgpu_debug_host_copy(void *dst, const void *src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(BOWGPU_ERR_ARG, "null buffer / negative size");
    if (bytes > 0) staged_memcpy(dst, src, (size_t)bytes);
    return 0;
}

int bowgpu_debug_status(int32_t first, int32_t n, uint32_t *out, int32_t zero_after) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    if (!out || first < 0 || n <= 0 || (size_t)(first + n) * 4 > 8192 || !c->d_scratch) return fail(BOWGPU_ERR_ARG, "bad status range");
    BG_TRY(copy_d2h(c, out, reinterpret_cast<uint32_t *>(c->d_scratch) + first, (size_t)n * 4));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (zero_after) BG_HIP(hipMemsetAsync(reinterpret_cast<uint32_t *>(c->d_scratch) + first, 0, (size_t)n * 4, c->stream));
    return 0;
}

int bowgpu_checksum64(const void *dev, int64_t n_words, uint64_t *xor_out, uint64_t *sum_out) {
    return bowgpu_checksum64_at(dev, n_words, 0, xor_out, sum_out);
}

int bowgpu_checksum64_at(const void *dev, int64_t n_words, int64_t index_base, uint64_t *xor_out, uint64_t *sum_out) {
    Ctx *c;
    BG_TRY(ctx_get(&c));
    void *d;
    BG_TRY(ctx_scratch(c, 4096, &d));
    uint64_t *dout = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(d) + 3072);
    BG_TRY(launch_checksum64(c, dev, n_words, dout, (uint64_t)index_base));
    uint64_t h[2] = {0, 0};
    BG_HIP(hipMemcpyAsync(h, dout, 16, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    *xor_out = h[0];
    *sum_out = h[1];
    return 0;
}

}  // extern "C"
