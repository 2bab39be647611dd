"""BASELINE configs[2] as the pipeline it names - Interpolate(WindowStart, Linear) then Mean - on 1e8 irregular rows with 30 % nulls,
interval 100, offsets 0 and 7: the two calls (count + fill, then Aggregate on the filled frame: what a caller who wants the filled Bow
makes) against the ONE call bowgpu_rolling_interpolate_aggregate (rolling_fused.hip: the interpolated frame is never written), and the
same entry point pushed through its two-call form.  Wall per call (outputs allocated once), kernel bracket, bytes."""
import os, sys, time
sys.path.insert(0, '.')
import ctypes as C
from bow_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ts, val = capi.gen_sparse(0, n, seed=42)
# the column's null count as a Bow knows it (Data().NullN(): what the cgo shim passes; -1 would make every call count the bits first)
valid = capi.aggregate_whole([ts, val], 0, [("Count", 1)])[0].to_list()[0]
val = capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, n - valid)
cols = [ts, val]
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
L = capi.lib()
def med(fn, reps=9):
    fn(); capi.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); capi.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]
SETS = ([("WindowStart", 0), ("ArithmeticMean", 1)], [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1)],
        [("WindowStart", 0), ("Sum", 1), ("ArithmeticMean", 1), ("Min", 1), ("Max", 1), ("First", 1), ("Last", 1)])
if len(sys.argv) > 2 and sys.argv[2] == "quick":
    SETS = SETS[:1]
# steady clocks before the first measurement: round 6 found the first-measured configuration 6 - 10 % slower whichever it was (CFG2_ORDER=rev);
# one second of the same call in front takes that out of the table (CFG2_WARM=0: as before)
if os.environ.get('CFG2_WARM', '1') != '0':
    s0_, W_ = capi.plan_windows(ts, 100, 0)
    outs_ = [capi.OutColumn(W_, capi.DEVICE) for _ in SETS[0]]
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        capi.rolling_interpolate_aggregate(cols, 0, 100, ip, SETS[0], offset=0, outs=outs_)
    capi.synchronize()
    del outs_
for aggs in SETS:
    for offset in ((7, 0) if os.environ.get('CFG2_ORDER') == 'rev' else (0, 7)):   # (CFG2_ORDER=rev: which offset is measured first - round 6: the 10 % spread of round 5 follows the ORDER, not the offset)
        s0, W = capi.plan_windows(ts, 100, offset)
        outs = [capi.OutColumn(W, capi.DEVICE) for _ in aggs]
        kms = []
        def fused():
            _, info = capi.rolling_interpolate_aggregate(cols, 0, 100, ip, aggs, offset=offset, outs=outs)
            kms.append(info.kernel_ms)
        w_f = med(fused)
        took = capi.last_kernel_name()
        k_f = sorted(kms)[len(kms) // 2]
        chk = [capi.checksum64(o.values, W) for o in outs]
        with capi.route(capi.ROUTE_NO_FUSED):
            w_2 = med(fused)
            chk2 = [capi.checksum64(o.values, W) for o in outs]
        # the two public calls, outputs of the first allocated once
        filled = capi.rolling_interpolate(cols, 0, 100, ip, offset=offset, out_residency=capi.DEVICE)
        m = filled[0].length
        carr, iarr, opts = capi._cols(cols), capi._interps(ip), capi.Options(offset, 0, 0)
        oarr = (capi.Out * 2)()
        def two_calls():
            m_ = C.c_int64(0)
            capi.check(L.bowgpu_rolling_interpolate_count(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, C.byref(m_)))
            for i, o in enumerate(filled):
                oarr[i] = o.c()
            capi.check(L.bowgpu_rolling_interpolate_fill(carr, 2, 0, C.c_int64(100), C.byref(opts), iarr, 2, oarr))
            c2 = [capi.Column(filled[0].values, None, capi.INT64, 0, m, 0), capi.Column(filled[1].values, filled[1].validity, capi.FLOAT64, 0, m, -1)]
            capi.rolling_aggregate(c2, 0, 100, aggs, offset=offset, outs=outs)
        w_p = med(two_calls)
        chk3 = [capi.checksum64(o.values, W) for o in outs]
        rb = n * 16.125
        print("%-52s off=%d W=%d: ONE call %.3f ms wall, kernel %.3f ms (%s; %.1f G rows/s, %.2f of 8 TB/s on %.2f GB read) | entry point, two-call form %.3f ms | "
              "Interpolate (count + fill) then Aggregate %.3f ms (%d rows materialised) | same bits: %s" %
              ("+".join(a[0] for a in aggs), offset, W, w_f, k_f, took, n / w_f / 1e6, rb / k_f / 1e6 / 8000, rb / 1e9, w_2, w_p, m, chk == chk2 == chk3))
        del filled
