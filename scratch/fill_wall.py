"""Wall time of the Fill* / FillLinear / Mean ABI calls at 1e8 rows with the outputs allocated ONCE outside the timed region
(scratch/interp_bench.py times the Python helper, which allocates 0.8 GB of HBM per call) next to the kernel's own time."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
from bow_amd import capi
n = 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
L = capi.lib()
out = capi.OutColumn(n, capi.DEVICE)
def timeit(fn, reps=8):
    fn(); capi.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return t[len(t) // 2]
for method in ("Previous", "Next", "Mean"):
    def call():
        o = out.c(); c = val.c(); u = C.c_int32(0)
        capi.check(L.bowgpu_fill(C.byref(c), capi.FILL[method], C.byref(o), C.byref(u)))
    w = timeit(call)
    print("Fill%-9s wall %.3f ms per call (kernel %s %.3f ms)  %.1f G rows/s" % (method, w, capi.last_kernel_name(), capi.last_kernel_ms(), n / w / 1e6))
aggs = [("WindowStart", 0), ("ArithmeticMean", 1)]
s0, W = capi.plan_windows(ts, 100, 0)
outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
plan = capi.plan_windows_ex(ts, 100, 0)
info = [None]
def mean():
    info[0] = capi.rolling_aggregate([ts, val], 0, 100, aggs, outs=outs, plan=plan)[1]
w = timeit(mean)
print("Mean (30%% nulls, interval 100, planned) wall %.3f ms per call (kernel %.3f ms)  ratio %.2f" % (w, info[0].kernel_ms, w / info[0].kernel_ms))
tsd, vald = capi.gen_dense(0, n, seed=42)
s0, W = capi.plan_windows(tsd, 10, 0)
outs2 = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
plan2 = capi.plan_windows_ex(tsd, 10, 0)
def mean2():
    info[0] = capi.rolling_aggregate([tsd, vald], 0, 10, aggs, outs=outs2, plan=plan2)[1]
w = timeit(mean2)
print("Mean (dense, interval 10, planned)        wall %.3f ms per call (kernel %.3f ms)  ratio %.2f" % (w, info[0].kernel_ms, w / info[0].kernel_ms))
