"""ctypes binding of the CPU ORACLE (oracle/libbow_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under bow_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FLOAT64, INT64, BOOLEAN = 1, 2, 3
TYPE_NAMES = {"float64": FLOAT64, "int64": INT64, "bool": BOOLEAN}

AGG = {
    "WindowStart": 0, "Sum": 1, "ArithmeticMean": 2, "Min": 3, "Max": 4, "Count": 5, "First": 6,
    "Last": 7, "IntegralStep": 8, "IntegralTrapezoid": 9, "WeightedAverageStep": 10,
    "WeightedAverageLinear": 11, "NumRows": 12, "Mode": 13,
}
INTERP = {"WindowStart": 0, "Linear": 1, "StepPrevious": 2, "None": 3, "Const": 4}

ERR = {
    -1: "interval", -2: "ts_type", -3: "first_ts_null", -4: "no_agg", -5: "keep_interval",
    -6: "bad_col", -7: "type", -8: "not_sorted", -9: "unsupported", -10: "arg",
}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d (%s)" % (code, ERR.get(code, "?")))
        self.code = code


class _Col(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("offset", C.c_int64),
                ("length", C.c_int64), ("type", C.c_int32), ("_pad", C.c_int32)]


class _Out(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64),
                ("type", C.c_int32), ("_pad", C.c_int32)]


class _Agg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("n_factors", C.c_int32), ("_pad", C.c_int32),
                ("factors", C.POINTER(C.c_double))]


class _Interp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("const_value", C.c_double),
                ("has_prev_row", C.c_int32), ("prev_t_valid", C.c_int32), ("prev_v_valid", C.c_int32),
                ("_pad", C.c_int32), ("prev_t", C.c_double), ("prev_v", C.c_double), ("prev_v_i64", C.c_int64)]


def build(force=False):
    # BOW_ORACLE_SANITIZED=1 (tests/test_oracle_sanitized.py): the AddressSanitizer + UBSan build of the same source
    if os.environ.get("BOW_ORACLE_SANITIZED") == "1":
        subprocess.check_call(["make", "-C", _HERE, "-s", "libbow_oracle_asan.so"])
        return os.path.join(_HERE, "libbow_oracle_asan.so")
    so = os.path.join(_HERE, "libbow_oracle.so")
    src = os.path.join(_HERE, "bow_oracle.c")
    hdr = os.path.join(_HERE, "bow_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libbow_oracle.so"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_iterate_windows.restype = C.c_int64
        _lib.orc_mix64.restype = C.c_uint64
        _lib.orc_mix64.argtypes = [C.c_uint64, C.c_uint64]
    return _lib


def pack_validity(valid_bools, bit_offset=0):
    """list/array of bools -> Arrow LSB-first bitmap bytes (np.uint8)."""
    v = np.asarray(valid_bools, dtype=bool)
    if bit_offset:
        v = np.concatenate([np.zeros(bit_offset, dtype=bool), v])
    return np.packbits(v, bitorder="little")


def unpack_validity(bitmap, n, bit_offset=0):
    if n == 0:
        return np.zeros(0, dtype=bool)
    bits = np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")
    return bits[bit_offset:bit_offset + n].astype(bool)


class Column:
    """One Arrow array: values (np.int64 / np.float64 / packed bool bits), packed validity or
    None (all valid), array offset, logical length."""

    def __init__(self, values, validity=None, typ=None, offset=0, length=None):
        if typ is None:
            typ = INT64 if np.asarray(values).dtype == np.int64 else FLOAT64
        self.type = typ
        if typ == BOOLEAN:
            self.values = np.ascontiguousarray(values, dtype=np.uint8)  # packed bits
            assert length is not None
        elif typ == INT64:
            self.values = np.ascontiguousarray(values, dtype=np.int64)
        else:
            self.values = np.ascontiguousarray(values, dtype=np.float64)
        self.validity = None if validity is None else np.ascontiguousarray(validity, dtype=np.uint8)
        self.offset = offset
        self.length = (len(self.values) - offset) if length is None else length

    @classmethod
    def from_list(cls, data, typ):
        """data: python list with None for nulls (like bow.NewBowFromColBasedInterfaces)."""
        if isinstance(typ, str):
            typ = TYPE_NAMES[typ]
        n = len(data)
        valid = [x is not None for x in data]
        validity = pack_validity(valid) if n else np.zeros(0, dtype=np.uint8)
        if typ == BOOLEAN:
            vals = pack_validity([bool(x) if x is not None else False for x in data]) if n else np.zeros(0, np.uint8)
            return cls(vals, validity, BOOLEAN, 0, n)
        dt = np.int64 if typ == INT64 else np.float64
        vals = np.array([x if x is not None else 0 for x in data], dtype=dt)
        return cls(vals, validity, typ, 0, n)

    def c(self):
        s = _Col()
        s.values = self.values.ctypes.data if self.values.size else None
        s.validity = None if self.validity is None else (self.validity.ctypes.data if self.validity.size else None)
        s.offset, s.length, s.type = self.offset, self.length, self.type
        return s

    def valid_mask(self):
        if self.validity is None:
            return np.ones(self.length, dtype=bool)
        return unpack_validity(self.validity, self.length, self.offset)

    def to_list(self):
        m = self.valid_mask()
        if self.type == BOOLEAN:
            vals = unpack_validity(self.values, self.length, self.offset)
            return [bool(v) if ok else None for v, ok in zip(vals, m)]
        vals = self.values[self.offset:self.offset + self.length]
        conv = int if self.type == INT64 else float
        return [conv(v) if ok else None for v, ok in zip(vals, m)]


class OutBuf:
    def __init__(self, n, width_type=FLOAT64):
        self.n = n
        self.values = np.full(max(n, 1), 0x5A5A5A5A5A5A5A5A, dtype=np.uint64)  # poisoned: oracle must zero-init
        self.validity = np.full((n + 7) // 8 + 1, 0xA5, dtype=np.uint8)
        self.type = 0

    def c(self):
        o = _Out()
        o.values = self.values.ctypes.data
        o.validity = self.validity.ctypes.data
        o.length = self.n
        o.type = 0
        return o

    def column(self, typ, n=None):
        n = self.n if n is None else n
        if typ == INT64:
            vals = self.values[:n].view(np.int64).copy()
        elif typ == FLOAT64:
            vals = self.values[:n].view(np.float64).copy()
        else:
            vals = self.values.view(np.uint8)[:(n + 7) // 8].copy()
        return Column(vals, self.validity[:(n + 7) // 8].copy(), typ, 0, n)


def _cols_array(cols):
    arr = (_Col * len(cols))()
    for i, c in enumerate(cols):
        arr[i] = c.c()
    return arr


def enforce_interval_and_offset(interval, offset):
    out = C.c_int64()
    rc = lib().orc_enforce_interval_and_offset(C.c_int64(interval), C.c_int64(offset), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out.value


def plan_windows(ts, interval, offset=0):
    s0, W = C.c_int64(), C.c_int64()
    c = ts.c()
    rc = lib().orc_plan_windows(C.byref(c), C.c_int64(interval), C.c_int64(offset), C.byref(s0), C.byref(W))
    if rc:
        raise OracleError(rc)
    return s0.value, W.value


def iterate_windows(ts, interval, offset=0, inclusive=False):
    s0, W = plan_windows(ts, interval, offset)
    cap = max(W, 1)
    arrs = [np.zeros(cap, dtype=np.int64) for _ in range(5)]
    inc = np.zeros(cap, dtype=np.uint8)
    c = ts.c()
    n = lib().orc_iterate_windows(C.byref(c), C.c_int64(interval), C.c_int64(offset), int(bool(inclusive)),
                                  C.c_int64(cap), *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                  inc.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise OracleError(n)
    keys = ["first_index", "slice_begin", "slice_end", "first_value", "last_value"]
    return [dict({k: int(a[i]) for k, a in zip(keys, arrs)}, is_inclusive=bool(inc[i])) for i in range(n)]


def aggregate(cols, ts_col, interval, aggs, offset=0, inclusive=False):
    """aggs: list of (kind_name, col_index[, factors]).  Returns (list[Column], new_interval_col)."""
    s0, W = plan_windows(cols[ts_col], interval, offset)
    carr = _cols_array(cols)
    aarr = (_Agg * max(len(aggs), 1))()
    keep = []
    for i, a in enumerate(aggs):
        factors = list(a[2]) if len(a) > 2 and a[2] else []
        fa = (C.c_double * max(len(factors), 1))(*factors)
        keep.append(fa)
        aarr[i].kind, aarr[i].col, aarr[i].n_factors = AGG[a[0]], a[1], len(factors)
        aarr[i].factors = C.cast(fa, C.POINTER(C.c_double))
    outs = [OutBuf(W) for _ in aggs]
    oarr = (_Out * max(len(aggs), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    nic = C.c_int(-1)
    rc = lib().orc_aggregate(carr, len(cols), ts_col, C.c_int64(interval), C.c_int64(offset),
                             int(bool(inclusive)), aarr, len(aggs), oarr, C.byref(nic))
    if rc:
        raise OracleError(rc)
    return [o.column(oarr[i].type) for i, o in enumerate(outs)], nic.value


def aggregate_whole(cols, ts_col, aggs):
    carr = _cols_array(cols)
    aarr = (_Agg * max(len(aggs), 1))()
    keep = []
    for i, a in enumerate(aggs):
        factors = list(a[2]) if len(a) > 2 and a[2] else []
        fa = (C.c_double * max(len(factors), 1))(*factors)
        keep.append(fa)
        aarr[i].kind, aarr[i].col, aarr[i].n_factors = AGG[a[0]], a[1], len(factors)
        aarr[i].factors = C.cast(fa, C.POINTER(C.c_double))
    outs = [OutBuf(1) for _ in aggs]
    oarr = (_Out * max(len(aggs), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    rc = lib().orc_aggregate_whole(carr, len(cols), ts_col, aarr, len(aggs), oarr)
    if rc:
        raise OracleError(rc)
    return [o.column(oarr[i].type, oarr[i].length) for i, o in enumerate(outs)]


def interpolate(cols, ts_col, interval, interps, offset=0, inclusive=False):
    """interps: list of dicts {kind, col, const, prev:(t, t_valid, v, v_valid, v_i64)}"""
    carr = _cols_array(cols)
    iarr = (_Interp * max(len(interps), 1))()
    for i, ip in enumerate(interps):
        iarr[i].kind, iarr[i].col = INTERP[ip["kind"]], ip["col"]
        iarr[i].const_value = ip.get("const", 0.0)
        prev = ip.get("prev")
        if prev is not None:
            iarr[i].has_prev_row = 1
            iarr[i].prev_t, iarr[i].prev_t_valid = prev[0], int(prev[1])
            iarr[i].prev_v, iarr[i].prev_v_valid = prev[2], int(prev[3])
            iarr[i].prev_v_i64 = prev[4] if len(prev) > 4 else 0
    n_out = C.c_int64(0)
    rc = lib().orc_interpolate(carr, len(cols), ts_col, C.c_int64(interval), C.c_int64(offset),
                               int(bool(inclusive)), iarr, len(interps), None, C.byref(n_out))
    if rc:
        raise OracleError(rc)
    n = n_out.value
    outs = [OutBuf(n) for _ in interps]
    oarr = (_Out * max(len(interps), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    rc = lib().orc_interpolate(carr, len(cols), ts_col, C.c_int64(interval), C.c_int64(offset),
                               int(bool(inclusive)), iarr, len(interps), oarr, C.byref(n_out))
    if rc:
        raise OracleError(rc)
    return [o.column(oarr[i].type) for i, o in enumerate(outs)]


def fill_linear(cols, ref_col, fill_col):
    carr = _cols_array(cols)
    n = cols[fill_col].length
    out = OutBuf(n)
    o = out.c()
    unchanged = C.c_int(0)
    rc = lib().orc_fill_linear(carr, len(cols), ref_col, fill_col, C.byref(o), C.byref(unchanged))
    if rc:
        raise OracleError(rc)
    return out.column(o.type), bool(unchanged.value)


FILL = {"Previous": 0, "Next": 1, "Mean": 2}


def fill(col, method):
    """Bow.FillPrevious / FillNext / FillMean of one column -> (Column, unchanged)"""
    c = col.c()
    out = OutBuf(col.length)
    o = out.c()
    unchanged = C.c_int(0)
    rc = lib().orc_fill(C.byref(c), FILL[method], C.byref(o), C.byref(unchanged))
    if rc:
        raise OracleError(rc)
    return out.column(o.type), bool(unchanged.value)


def is_col_sorted(col):
    c = col.c()
    return bool(lib().orc_is_col_sorted(C.byref(c)))


def gen_dense(row0, n, seed=42):
    ts = np.empty(n, dtype=np.int64)
    val = np.empty(n, dtype=np.float64)
    lib().orc_gen_dense(C.c_int64(row0), C.c_int64(n), C.c_uint64(seed), ts.ctypes.data_as(C.c_void_p),
                        val.ctypes.data_as(C.c_void_p))
    return ts, val


def gen_sparse(row0, n, seed=42):
    assert row0 % 8 == 0
    ts = np.empty(n, dtype=np.int64)
    val = np.empty(n, dtype=np.float64)
    validity = np.zeros((n + 7) // 8, dtype=np.uint8)
    lib().orc_gen_sparse(C.c_int64(row0), C.c_int64(n), C.c_uint64(seed), ts.ctypes.data_as(C.c_void_p),
                         val.ctypes.data_as(C.c_void_p), validity.ctypes.data_as(C.c_void_p))
    return ts, val, validity
