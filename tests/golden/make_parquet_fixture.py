"""Dump the columns of the reference's benchmark input (benchmarks/bow1-100000-rows.parquet, the file its
BenchmarkBow_Fill / BenchmarkBow_IsColSorted read: bowfill_test.go:550-553, bowassertion_test.go:93-96, and
BASELINE.json configs[0]) into tests/golden/bow1_100000_rows.npz: raw little-endian values + Arrow validity
bitmaps (LSB first).  The fixture is DATA only; every expectation the tests derive from it comes from the oracle or
from an independent numpy restatement ("restatement-derived, not reference-executed": the Go reference cannot
run in this image).

Run in the build container (pyarrow is available; /root/reference is mounted):
    python tests/golden/make_parquet_fixture.py
"""
import os

import numpy as np
import pyarrow.parquet as pq

SRC = "/root/reference/benchmarks/bow1-100000-rows.parquet"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bow1_100000_rows.npz")
COLS = ("Int64_ref", "Int64_no_nils_bow1", "Int64_bow1", "Float64_bow1")


def main():
    t = pq.read_table(SRC)
    out = {}
    for name in COLS:
        c = t.column(name).combine_chunks()
        valid = ~np.asarray(c.is_null())
        out[name] = np.asarray(c.fill_null(0))  # null slots hold 0, as bow.NewBuffer leaves them (bowbuffer.go:22-40)
        out[name + "_valid"] = np.packbits(valid, bitorder="little")
        print("%-20s %-8s rows=%d nulls=%d" % (name, out[name].dtype, len(valid), int((~valid).sum())))
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    main()
