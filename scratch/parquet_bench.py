"""Parquet loader throughput: a file like the reference writes (SNAPPY, PLAIN, 8 KB pages, OPTIONAL columns, 30 % nulls)."""
import sys, time, os, tempfile
sys.path.insert(0, '.')
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from bow_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(1)
cols = {"ts": pa.array(np.cumsum(rng.integers(1, 20, n)).astype(np.int64)),
        "val": pa.array(rng.integers(0, 10, n) + 0.5, mask=rng.random(n) < 0.3),
        "rnd": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.3)}
table = pa.table(cols)
d = tempfile.mkdtemp()
for page in (8192, 1 << 20):
    path = os.path.join(d, "f%d.parquet" % page)
    pq.write_table(table, path, compression="snappy", use_dictionary=False, data_page_size=page, data_page_version="1.0")
    sz = os.path.getsize(path)
    t0 = time.perf_counter(); t = pq.read_table(path); t1 = time.perf_counter()
    print("page %7d B: file %.1f MB; pyarrow read_table (all cores) %.1f ms" % (page, sz / 1e6, (t1 - t0) * 1e3))
    f = capi.ParquetFile(path)
    for rep in range(2):
        for i, (name, typ, opt) in enumerate(f.columns):
            t0 = time.perf_counter()
            out = f.read_column(i, out_residency=capi.DEVICE)
            capi.synchronize()
            t1 = time.perf_counter()
            if rep == 1:
                print("   column %-4s -> device: %.1f ms  (%.2f G rows/s, %.2f GB/s of decoded values)" % (name, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, n * 8 / (t1 - t0) / 1e9))
            del out
    f.close()
