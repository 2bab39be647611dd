import time, sys
sys.path.insert(0, '.')
from oracle import pyoracle as orc
n = 20_000_000
t = time.time(); ts, val, bm = orc.gen_sparse(0, n, seed=5); print("gen_sparse", n, time.time() - t)
cols = [orc.Column(ts, None, orc.INT64), orc.Column(val, bm, orc.FLOAT64)]
ip = [{"kind": "WindowStart", "col": 0}, {"kind": "Linear", "col": 1}]
t = time.time(); w = orc.interpolate(cols, 0, 100, ip, offset=7); print("interp", time.time() - t, w[0].length)
