import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from bow_amd import capi
from oracle import pyoracle as orc
from test_gpu_aggregate import make_vals, ALL_AGGS, TIME_AGGS
rng = np.random.default_rng(2024)
n = 400_000
step = rng.integers(1, 5, n)
step[rng.random(n) < 0.002] = 4000
ts = np.cumsum(step).astype(np.int64) - 3000
burst = rng.random(n) < 0.0005
for i in np.flatnonzero(burst)[:40]:
    ts[i:i + int(rng.integers(200, 3000))] = ts[i]
ts = np.sort(ts)
f, fm = make_vals(rng, n, "f64", 0.3)
g, gm = make_vals(rng, n, "i64", 0.1)
import itertools
print("ts[:12]", ts[:12])
for aggs in ([('WindowStart',0),('WeightedAverageLinear',2)], [('WindowStart',0),('Sum',2),('WeightedAverageLinear',2)], [('WindowStart',0),('IntegralTrapezoid',1),('WeightedAverageLinear',2)], [('WindowStart',0),('WeightedAverageLinear',1),('WeightedAverageLinear',2)]):
  print(aggs)
  for strict in (False,):
      cols = [capi.Column(ts), capi.Column(f, np.packbits(fm, bitorder="little"), capi.FLOAT64, 0, n, -1), capi.Column(g, np.packbits(gm, bitorder="little"), capi.INT64, 0, n, -1)]
      ocols = [orc.Column(ts, None, orc.INT64), orc.Column(f, np.packbits(fm, bitorder="little"), orc.FLOAT64), orc.Column(g, np.packbits(gm, bitorder="little"), orc.INT64)]
      exp, _ = orc.aggregate(ocols, 0, 50, aggs)
      outs, info = capi.rolling_aggregate(cols, 0, 50, aggs, strict_order=strict)
      print("strict", strict, capi.last_kernel_name(), "W", info.num_windows, "avg", n / info.num_windows, "long", info.long_windows)
      for (k, c), got, want in zip(aggs, outs, exp):
          gmk, wmk = got.valid_mask(), want.valid_mask()
          bad = np.flatnonzero(gmk != wmk)
          gv = got.host_arrays()[0]; wv = want.values[:want.length]
          badv = np.flatnonzero((gv.view(np.uint64) != wv.view(np.uint64)) & gmk & wmk)
          if len(bad) or len(badv):
              print("  ", k, c, "validity mismatches", len(bad), bad[:8], "got", gmk[bad[:4]], "| value mismatches", len(badv), badv[:6], gv[badv[:3]], wv[badv[:3]])
