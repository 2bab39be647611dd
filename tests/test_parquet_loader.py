"""Parquet column chunk -> device column (bowgpu_parquet_*; reference bowparquet.go:44-153).  The expectation is
pyarrow's own decode of the same file - an independent implementation.  Files: the reference's benchmark inputs (data files its
tests hold: benchmarks/bow1-{100,1000,10000}-rows.parquet, written by parquet-go with SNAPPY / PLAIN / 8 KB pages, copied to
tests/golden/) and files pyarrow writes here with the same options over the shapes that matter (null patterns, many small pages,
several row groups, REQUIRED columns, compressible data that makes Snappy emit overlapping back-references)."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from bow_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILES = [os.path.join(HERE, "golden", "bow1-%d-rows.parquet" % n) for n in (100, 1000, 10000)]


def expect(table, name):
    c = table.column(name).combine_chunks()
    valid = ~np.asarray(c.is_null())
    return np.asarray(c.fill_null(0)), valid


def check_file(path, table=None):
    table = table if table is not None else pq.read_table(path)
    f = capi.ParquetFile(path)
    assert f.num_rows == table.num_rows
    assert [c[0] for c in f.columns] == table.schema.names
    n = 0
    for i, (name, typ, optional) in enumerate(f.columns):
        if typ not in (capi.INT64, capi.FLOAT64):
            with pytest.raises(capi.BowGpuError) as e:
                f.read_column(i)
            assert e.value.code == -9
            continue
        for res in (capi.HOST, capi.DEVICE):
            out = f.read_column(i, out_residency=res)
            vals, valid = expect(table, name)
            assert out.type == typ and out.length == len(vals), name
            gm = out.valid_mask()
            assert np.array_equal(gm, valid), (name, np.flatnonzero(gm != valid)[:10])
            gv = out.host_arrays()[0]
            assert np.array_equal(gv.view(np.uint64)[gm], vals.view(np.uint64)[valid]), name
            assert not gv.view(np.uint64)[~gm].any(), name          # null slots hold 0 (bow.NewBuffer)
            assert out.null_count == int((~valid).sum()), name
        n += 1
    f.close()
    return n


def test_footer_of_the_reference_files_without_a_gpu():
    for path in REF_FILES:
        f = capi.ParquetFile(path)
        md = pq.ParquetFile(path)
        assert f.num_rows == md.metadata.num_rows
        assert [c[0] for c in f.columns] == md.schema_arrow.names
        kinds = {"int64": capi.INT64, "double": capi.FLOAT64}
        for (name, typ, optional), field in zip(f.columns, md.schema_arrow):
            if str(field.type) in kinds:
                assert typ == kinds[str(field.type)] and optional
        f.close()
    with pytest.raises(capi.BowGpuError):
        capi.ParquetFile(os.path.join(HERE, "golden", "reference_vectors.json"))  # not a parquet file
    with pytest.raises(capi.BowGpuError):
        capi.ParquetFile("/nonexistent/file.parquet")


@pytest.mark.gpu
def test_reference_benchmark_files_decode_like_pyarrow():
    for path in REF_FILES:
        assert check_file(path) == 4  # Int64_ref, Int64_no_nils_bow1, Int64_bow1, Float64_bow1


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["snappy", "none"])
def test_files_written_here(tmp_path, compression):
    rng = np.random.default_rng(5)
    for case, (n, null_frac, page, rg) in enumerate([(1, 0.0, 8192, None), (1000, 0.3, 8192, None), (100_000, 0.3, 8192, None),
                                                     (100_000, 0.0, 1024, 30_000), (250_000, 0.9, 4096, 100_000), (70_000, 1.0, 8192, None),
                                                     (1_000_000, 0.05, 65536, 300_000)]):
        def mask():
            if null_frac == 0.0:
                return None
            m = rng.random(n) < null_frac
            runs = rng.random(n) < 0.001   # long runs of nulls / of valid rows => RLE runs next to bit-packed ones
            m[np.repeat(runs[::64], 64)[:n]] = True
            return m
        cols = {
            "ts": pa.array(np.cumsum(rng.integers(1, 20, n)).astype(np.int64)),                        # ascending: compressible
            "f_rand": pa.array(rng.standard_normal(n), mask=mask()),                                    # incompressible
            "i_small": pa.array(rng.integers(0, 10, n).astype(np.int64), mask=mask()),                 # many back-references
            "f_const": pa.array(np.full(n, 2.5), mask=mask()),                                         # overlapping copies
            "i_req": pa.array(np.arange(n, dtype=np.int64) * 3),
        }
        schema = pa.schema([pa.field(k, v.type, nullable=(k != "i_req")) for k, v in cols.items()])
        table = pa.table(cols, schema=schema)
        path = str(tmp_path / ("case%d_%s.parquet" % (case, compression)))
        pq.write_table(table, path, compression=compression, use_dictionary=False, data_page_size=page, data_page_version="1.0",
                       row_group_size=rg, write_statistics=bool(case % 2))
        assert check_file(path, table) == 5


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["snappy", "none"])
def test_dictionary_encoded_files(tmp_path, compression):
    """what pyarrow / pandas write by default: a dictionary page per column chunk, data pages of RLE / bit-packed indices (bit widths
    1 .. 17 here), and PLAIN fall-back pages once a dictionary has grown too large"""
    rng = np.random.default_rng(8)
    for case, (n, card, null_frac, page, rg) in enumerate([(1000, 3, 0.0, 8192, None), (100_000, 7, 0.3, 8192, None), (100_000, 1000, 0.3, 4096, 30_000),
                                                           (300_000, 100_000, 0.1, 65536, None), (50_000, 1, 0.5, 8192, None),
                                                           (200_000, 2, 1.0, 8192, 50_000), (400_000, 300_000, 0.0, 1 << 20, None)]):
        def mask():
            return None if null_frac == 0.0 else rng.random(n) < null_frac
        pool_f = rng.standard_normal(card)
        pool_i = rng.integers(-2 ** 40, 2 ** 40, card).astype(np.int64)
        cols = {"f": pa.array(pool_f[rng.integers(0, card, n)], mask=mask()),
                "i": pa.array(pool_i[rng.integers(0, card, n)], mask=mask()),
                "runs": pa.array(np.repeat(pool_i[:max(card // 2, 1)], -(-n // max(card // 2, 1)))[:n], mask=mask())}   # long RLE runs
        table = pa.table(cols)
        path = str(tmp_path / ("dict%d_%s.parquet" % (case, compression)))
        pq.write_table(table, path, compression=compression, use_dictionary=True, data_page_size=page, data_page_version="1.0",
                       row_group_size=rg, dictionary_pagesize_limit=64 * 1024 if case in (3, 6) else 1 << 20)
        assert check_file(path, table) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["snappy", "none"])
@pytest.mark.parametrize("use_dictionary", [False, True])
def test_data_page_v2(tmp_path, compression, use_dictionary):
    """data page v2: the definition levels sit uncompressed in front of the (possibly compressed) values, without a length prefix"""
    rng = np.random.default_rng(13)
    for case, (n, null_frac, page, rg) in enumerate([(1000, 0.3, 8192, None), (200_000, 0.3, 4096, 60_000), (100_000, 0.0, 8192, None),
                                                     (50_000, 1.0, 8192, None)]):
        def mask():
            return None if null_frac == 0.0 else rng.random(n) < null_frac
        cols = {"ts": pa.array(np.cumsum(rng.integers(1, 20, n)).astype(np.int64)),
                "f": pa.array(np.round(rng.standard_normal(n), 1), mask=mask()),
                "i": pa.array(rng.integers(0, 50, n).astype(np.int64), mask=mask())}
        table = pa.table(cols)
        path = str(tmp_path / ("v2_%d.parquet" % case))
        pq.write_table(table, path, compression=compression, use_dictionary=use_dictionary, data_page_size=page, data_page_version="2.0",
                       row_group_size=rg)
        assert check_file(path, table) == 3


@pytest.mark.gpu
def test_declines_what_it_does_not_read(tmp_path):
    t = pa.table({"a": pa.array(np.arange(1000, dtype=np.int64) % 7)})
    p2 = str(tmp_path / "zstd.parquet")
    pq.write_table(t, p2, use_dictionary=False, compression="zstd")
    with pytest.raises(capi.BowGpuError) as e:
        capi.ParquetFile(p2).read_column(0)
    assert e.value.code == -9



@pytest.mark.gpu
def test_corrupted_files_fail_cleanly(tmp_path):
    # flipped bytes anywhere in the file (page payloads, page headers, footer): every call returns - with an error or with
    # some decode of the damaged data - and the library keeps working afterwards
    rng = np.random.default_rng(9)
    good = open(REF_FILES[1], "rb").read()
    for k in range(40):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(4, len(b) - 4))
            b[pos] = int(rng.integers(0, 256))
        if rng.random() < 0.2:
            b = b[:int(rng.integers(12, len(b)))] + b[-8:]  # truncated in the middle, tail kept
        path = str(tmp_path / ("bad%d.parquet" % k))
        open(path, "wb").write(bytes(b))
        try:
            f = capi.ParquetFile(path)
        except capi.BowGpuError:
            continue
        for i, (name, typ, opt) in enumerate(f.columns):
            if typ in (capi.INT64, capi.FLOAT64) and f.num_rows < 10_000_000:
                try:
                    f.read_column(i)
                except capi.BowGpuError:
                    pass
        f.close()
    assert check_file(REF_FILES[1]) == 4


@pytest.mark.gpu
def test_config0_parquet_to_rolling_mean_without_leaving_the_device():
    """BASELINE.json configs[0]: benchmarks/bow1-*-rows.parquet -> IntervalRolling + ArithmeticMean.  Columns are decoded into HBM
    and aggregated there; the expectation is the oracle over pyarrow's decode of the same file."""
    from oracle import pyoracle as orc
    from test_gpu_aggregate import compare
    path = REF_FILES[2]
    f = capi.ParquetFile(path)
    names = [c[0] for c in f.columns]
    ts = f.read_column(names.index("Int64_ref"), out_residency=capi.DEVICE)
    val = f.read_column(names.index("Float64_bow1"), out_residency=capi.DEVICE)
    n = f.num_rows
    assert ts.null_count == 0
    cols = [capi.Column(ts.values, None, capi.INT64, 0, n, 0), capi.Column(val.values, val.validity, capi.FLOAT64, 0, n, val.null_count)]
    aggs = [("WindowStart", 0), ("ArithmeticMean", 1), ("Count", 1), ("Min", 1), ("Max", 1)]
    table = pq.read_table(path)
    tv, _ = expect(table, "Int64_ref")
    vv, vm = expect(table, "Float64_bow1")
    ocols = [orc.Column(tv, None, orc.INT64), orc.Column(vv, np.packbits(vm, bitorder="little"), orc.FLOAT64)]
    for interval, offset in [(10, 0), (100, 7), (1000, 0)]:
        got, info = capi.rolling_aggregate(cols, 0, interval, aggs, offset=offset, out_residency=capi.DEVICE)
        want, _ = orc.aggregate(ocols, 0, interval, aggs, offset=offset)
        for (k, _), g, w in zip(aggs, got, want):
            compare("config0 %s I=%d" % (k, interval), g, w)


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["snappy", "none"])
def test_dictionary_pages_larger_than_one_mebibyte(tmp_path, compression):
    """High-cardinality but repeating INT64 / DOUBLE columns written with pyarrow defaults carry a dictionary page slightly over
    1 MiB (parquet-cpp falls back to PLAIN only after a batch pushed the dictionary past that size): ~140k distinct 8-byte values."""
    rng = np.random.default_rng(17)
    n = 600_000
    keys = rng.integers(0, 140_000, n)
    pool_i = rng.integers(-1 << 62, 1 << 62, 140_000)
    pool_f = rng.standard_normal(140_000) * 1e6
    mask = rng.random(n) < 0.1
    table = pa.table({"i": pa.array(pool_i[keys], mask=mask), "f": pa.array(pool_f[keys]), "k": pa.array(keys.astype(np.int64))})
    path = str(tmp_path / "bigdict.parquet")
    pq.write_table(table, path, compression=compression, use_dictionary=True)
    md = pq.ParquetFile(path).metadata
    enc = {md.row_group(0).column(j).path_in_schema: md.row_group(0).column(j) for j in range(3)}
    assert any(c.dictionary_page_offset is not None and c.total_uncompressed_size > (1 << 20) for c in enc.values())
    assert check_file(path, table) == 3


def _footer_file(tmp_path, name, footer):
    """PAR1 | footer | len | PAR1 : enough of a file for bowgpu_parquet_open to reach the Thrift parser"""
    p = tmp_path / name
    p.write_bytes(b"PAR1" + footer + len(footer).to_bytes(4, "little") + b"PAR1")
    return str(p)


def test_hostile_thrift_footers_fail_cleanly(tmp_path):
    """bowgpu_parquet_open is host-only: footers built to wrap a length, blow the stack or spin must come back as an error
    code - no exception across the C ABI, no crash, no seconds-long loop."""
    import time
    hostile = {
        # field 1 (i32 version) then field 2..: a binary (type 8) whose varint length is 2^64 - 1
        "huge_binary": bytes([0x15, 0x02, 0x18]) + b"\xff" * 9 + b"\x01",
        # a list (type 9) of lists of lists ... 200 deep, each with one element
        "deep_lists": bytes([0x19]) + bytes([0x19]) * 200 + b"\x00",
        # nested structs 100 000 deep: field id delta 1, type 12
        "deep_structs": bytes([0x1c]) * 100_000,
        # a map (type 11) with 2^62 bool -> bool entries (no bytes per entry)
        "huge_map": bytes([0x1b]) + b"\xff" * 8 + b"\x3f" + bytes([0x11]),
        # a list with a 2^60 element count of i32
        "huge_list": bytes([0x19, 0xf5]) + b"\xff" * 8 + b"\x0f",
    }
    for name, footer in hostile.items():
        path = _footer_file(tmp_path, name + ".parquet", footer)
        t0 = time.perf_counter()
        with pytest.raises(capi.BowGpuError):
            capi.ParquetFile(path)
        assert time.perf_counter() - t0 < 2.0, name
    # and the real files still open
    f = capi.ParquetFile(REF_FILES[0])
    assert f.num_rows == 100
    f.close()
