"""ctypes binding of libbowgpu.so (include/bowgpu.h) — plumbing for the tests and bench.py.

This module is NOT a CPU implementation of anything: every function forwards to the HIP
library.  If the library is missing or no GPU is present the call raises (BowGpuError /
OSError); there is no fallback.
"""
import contextlib
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOWGPU_LIB") or os.path.join(_HERE, "libbowgpu.so")

FLOAT64, INT64, BOOLEAN, STRING = 1, 2, 3, 4
HOST, DEVICE, HOST_PINNED = 0, 1, 2
TYPE_NAMES = {"float64": FLOAT64, "int64": INT64, "bool": BOOLEAN, "utf8": STRING}

AGG = {
    "WindowStart": 0, "Sum": 1, "ArithmeticMean": 2, "Min": 3, "Max": 4, "Count": 5, "First": 6,
    "Last": 7, "IntegralStep": 8, "IntegralTrapezoid": 9, "WeightedAverageStep": 10,
    "WeightedAverageLinear": 11, "NumRows": 12, "Mode": 13,
}
INTERP = {"WindowStart": 0, "Linear": 1, "StepPrevious": 2, "None": 3, "Const": 4}

MAX_FACTORS = 4
CARRY_MAX_AGGS = 16
ABI_VERSION = 6   # include/bowgpu.h BOWGPU_ABI_VERSION (asserted when the library is loaded)

ERR_NAMES = {
    -1: "INTERVAL", -2: "TS_TYPE", -3: "FIRST_TS_NULL", -4: "NO_AGG", -5: "KEEP_INTERVAL", -6: "BAD_COL",
    -7: "TYPE", -8: "NOT_SORTED", -9: "UNSUPPORTED", -10: "ARG", -11: "NO_DEVICE", -12: "HIP",
    -13: "TS_NULLS", -14: "TS_UNSORTED", -15: "OOM",
}


class BowGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bowgpu error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code
        self.message = msg


class Col(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("offset", C.c_int64), ("length", C.c_int64),
                ("null_count", C.c_int64), ("type", C.c_int32), ("residency", C.c_int32)]


class Out(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64), ("null_count", C.c_int64),
                ("type", C.c_int32), ("residency", C.c_int32)]


class Agg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("n_factors", C.c_int32), ("_pad", C.c_int32),
                ("factors", C.c_double * MAX_FACTORS)]


class Options(C.Structure):
    _fields_ = [("offset", C.c_int64), ("inclusive", C.c_int32), ("strict_order", C.c_int32)]


class AggInfo(C.Structure):
    _fields_ = [("s0", C.c_int64), ("num_windows", C.c_int64), ("new_interval_col", C.c_int32),
                ("inclusive", C.c_int32), ("long_windows", C.c_int64), ("kernel_ms", C.c_double)]


class Interp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("const_value", C.c_double), ("has_prev_row", C.c_int32),
                ("prev_t_valid", C.c_int32), ("prev_v_valid", C.c_int32), ("_pad", C.c_int32),
                ("prev_t", C.c_double), ("prev_v", C.c_double), ("prev_v_i64", C.c_int64)]


class InterpEdge(C.Structure):
    _fields_ = [("has_left", C.c_int32), ("_pad", C.c_int32), ("left_last_ts", C.c_int64), ("next_valid", C.c_int32 * 8),
                ("next_t", C.c_double * 8), ("next_v", C.c_double * 8)]


class InterpPoints(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("first_ts", C.c_int64), ("last_ts", C.c_int64), ("first_valid", C.c_int32 * 8),
                ("last_valid", C.c_int32 * 8), ("first_t", C.c_double * 8), ("first_v", C.c_double * 8), ("last_t", C.c_double * 8),
                ("last_v", C.c_double * 8), ("last_v_i64", C.c_int64 * 8)]


class CarryState(C.Structure):
    _fields_ = [("sum", C.c_double), ("vmin", C.c_double), ("vmax", C.c_double), ("nn_min", C.c_double),
                ("nn_max", C.c_double), ("first_bits", C.c_uint64), ("last_bits", C.c_uint64),
                ("count", C.c_int64), ("nrows", C.c_int64),
                ("pt", C.c_double), ("pv", C.c_double), ("first_pt", C.c_double), ("first_pv", C.c_double),
                ("integ_step", C.c_double), ("integ_trap", C.c_double),
                ("has_value", C.c_int32), ("has_nn", C.c_int32), ("has_point", C.c_int32), ("has_pair", C.c_int32)]


class NextRow(C.Structure):
    """bowgpu_next_row: the first row of a shard, as its left neighbour needs it for inclusive windows"""
    _fields_ = [("present", C.c_int32), ("_pad", C.c_int32), ("ts", C.c_int64), ("bits", C.c_uint64 * 16), ("valid", C.c_int32 * 16)]


class ShardCarry(C.Structure):
    _fields_ = [("first_window_id", C.c_int64), ("last_window_id", C.c_int64), ("first_ts", C.c_int64),
                ("last_ts", C.c_int64), ("nrows", C.c_int64), ("naggs", C.c_int32), ("_pad", C.c_int32),
                ("last", CarryState * CARRY_MAX_AGGS)]


class Plan(C.Structure):
    """bowgpu_plan: what newIntervalRolling computes once per Rolling"""
    _fields_ = [("s0", C.c_int64), ("num_windows", C.c_int64), ("first_ts", C.c_int64), ("last_ts", C.c_int64),
                ("interval", C.c_int64), ("offset", C.c_int64), ("nrows", C.c_int64)]


class ShardRecord(C.Structure):
    """bowgpu_shard_record: what one rank contributes to the exchange of a sharded Aggregate"""
    _fields_ = [("nrows", C.c_int64), ("first_ts", C.c_int64), ("last_ts", C.c_int64), ("carry_from_ts", C.c_int64),
                ("naggs", C.c_int32), ("flags", C.c_int32), ("first_row", NextRow), ("last", CarryState * CARRY_MAX_AGGS)]


class ShardDecision(C.Structure):
    """bowgpu_shard_decision: what bowgpu_shard_plan settles for one rank"""
    _fields_ = [("s0", C.c_int64), ("num_windows", C.c_int64), ("first_window_id", C.c_int64), ("last_window_id", C.c_int64),
                ("lead_empty_windows", C.c_int64), ("first_slot_window_id", C.c_int64), ("windows_local", C.c_int64),
                ("windows_owned", C.c_int64), ("holds_global_row0", C.c_int32), ("drops_last", C.c_int32),
                ("seed_first_rank", C.c_int32), ("next_rank", C.c_int32), ("finish_last", C.c_int32), ("retry_with_s0", C.c_int32)]


# every symbol include/bowgpu.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "bowgpu_abi_version", "bowgpu_rolling_interpolate_aggregate", "bowgpu_last_error", "bowgpu_device_count", "bowgpu_set_device", "bowgpu_device_name",
    "bowgpu_set_stream", "bowgpu_synchronize", "bowgpu_trim", "bowgpu_mem_info", "bowgpu_host_register", "bowgpu_host_unregister", "bowgpu_last_kernel_ms", "bowgpu_last_call_slow_rows", "bowgpu_last_kernel_name", "bowgpu_malloc", "bowgpu_free", "bowgpu_memcpy_h2d",
    "bowgpu_memcpy_d2h", "bowgpu_memset", "bowgpu_timer_create", "bowgpu_timer_start", "bowgpu_timer_stop",
    "bowgpu_timer_elapsed_ms", "bowgpu_timer_destroy", "bowgpu_enforce_interval_and_offset", "bowgpu_plan_windows",
    "bowgpu_rolling_aggregate", "bowgpu_plan_windows_ex", "bowgpu_rolling_aggregate_planned", "bowgpu_window_bounds", "bowgpu_aggregate_whole",
    "bowgpu_rolling_interpolate_count", "bowgpu_rolling_interpolate_fill", "bowgpu_shard_interp_points",
    "bowgpu_shard_interpolate_count", "bowgpu_shard_interpolate_fill", "bowgpu_fill_linear", "bowgpu_fill_linear_sorted", "bowgpu_fill",
    "bowgpu_is_col_sorted", "bowgpu_shard_span", "bowgpu_shard_aggregate", "bowgpu_shard_carry_only", "bowgpu_shard_first_row", "bowgpu_shard_fix_first", "bowgpu_carry_merge",
    "bowgpu_shard_begin", "bowgpu_shard_pass_begin", "bowgpu_shard_plan", "bowgpu_shard_finish", "bowgpu_gen_dense",
    "bowgpu_gen_sparse", "bowgpu_stream_read_ceiling", "bowgpu_stream_rw_probe", "bowgpu_debug_status", "bowgpu_debug_host_copy", "bowgpu_checksum64", "bowgpu_parquet_open", "bowgpu_parquet_close",
    "bowgpu_parquet_info", "bowgpu_parquet_column", "bowgpu_parquet_read_column",
    "bowgpu_debug_set_route", "bowgpu_debug_get_route", "bowgpu_checksum64_at",
    "bowgpu_set_devices", "bowgpu_get_devices", "bowgpu_set_fanout_min_rows", "bowgpu_last_call_ranks", "bowgpu_fanout_counts",
]

_lib = None


def lib():
    """Loads libbowgpu.so; raises OSError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C bow_amd/csrc). The bowgpu path has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.bowgpu_last_error.restype = C.c_char_p
        L.bowgpu_last_kernel_name.restype = C.c_char_p
        got = L.bowgpu_abi_version()
        if got != ABI_VERSION:   # struct layouts / option meanings of include/bowgpu.h this binding was written against
            raise OSError("%s has BOWGPU_ABI_VERSION %d, this binding expects %d: rebuild (make -C bow_amd/csrc)" % (LIB_PATH, got, ABI_VERSION))
        _lib = L
    return _lib


def page_aligned(count, dtype, fill=None):
    """a numpy array on pages of its own (registered buffers should not share pages with anything else)"""
    nbytes = max(count, 1) * np.dtype(dtype).itemsize
    raw = np.empty(((nbytes + 4095) // 4096 + 1) * 4096, dtype=np.uint8)
    off = (-raw.ctypes.data) % 4096
    arr = raw[off:off + nbytes].view(dtype)
    if fill is not None:
        arr[:] = fill
    return arr


def host_register(arr):
    """page-lock + map a numpy array for the device (columns over it may then be passed with residency HOST_PINNED)"""
    check(lib().bowgpu_host_register(arr.ctypes.data_as(C.c_void_p), C.c_int64(arr.nbytes)))


def host_unregister(arr):
    check(lib().bowgpu_host_unregister(arr.ctypes.data_as(C.c_void_p)))


def set_devices(ids, min_rows=None):
    """bowgpu_set_devices: Rolling.Aggregate calls are cut into row ranges over these devices from now on (process-wide; [] = off)"""
    ids = list(ids)
    arr = (C.c_int * max(len(ids), 1))(*ids)
    check(lib().bowgpu_set_devices(arr, len(ids)))
    if min_rows is not None:
        check(lib().bowgpu_set_fanout_min_rows(C.c_int64(min_rows)))


def get_devices():
    n = C.c_int(0)
    arr = (C.c_int * 64)()
    check(lib().bowgpu_get_devices(arr, 64, C.byref(n)))
    return [arr[i] for i in range(n.value)]


def fanout_counts():
    """(calls that arrived with a device list in force, calls that ran as row ranges), process-wide"""
    a, b = C.c_int64(0), C.c_int64(0)
    check(lib().bowgpu_fanout_counts(C.byref(a), C.byref(b)))
    return a.value, b.value


def last_call_ranks():
    """row ranges that served this thread's last Rolling.Aggregate call (1: the one-device path)"""
    n = C.c_int(0)
    check(lib().bowgpu_last_call_ranks(C.byref(n)))
    return n.value


class devices:
    """with capi.devices([0, 0, 0, 0], min_rows=1000): ...  - the fan-out in force inside the block, the previous setting after"""

    def __init__(self, ids, min_rows=None):
        self.ids, self.min_rows = list(ids), min_rows

    def __enter__(self):
        self.prev = get_devices()
        set_devices(self.ids, self.min_rows)
        return self

    def __exit__(self, *exc):
        set_devices(self.prev, (1 << 20) if self.min_rows is not None else None)
        return False


def mem_info():
    f, t = C.c_int64(0), C.c_int64(0)
    check(lib().bowgpu_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def trim(all_threads=True):
    n = C.c_int64(0)
    check(lib().bowgpu_trim(1 if all_threads else 0, C.byref(n)))
    return n.value


def check(rc):
    if rc != 0:
        raise BowGpuError(rc, lib().bowgpu_last_error().decode("utf-8", "replace"))


# ------------------------------------------------------------------ test / A-B routing (bowgpu_debug_set_route: per calling thread)
ROUTE_NO_SIMPLE, ROUTE_FORCE_GENERAL, ROUTE_NO_LONG_ONLY, ROUTE_LONG_CLASSIC, ROUTE_LONG_STREAM_ALL = 1, 2, 4, 8, 16
ROUTE_SIMPLE_SMALL_LIST, ROUTE_SIMPLE_LARGE_LIST, ROUTE_TW_F64, ROUTE_SIMPLE_PADDED, ROUTE_INTERP_TILE = 32, 64, 128, 256, 512
ROUTE_PINNED_STAGE, ROUTE_STRICT_ORDER, ROUTE_NO_FUSED, ROUTE_TW_ROWS, ROUTE_INTERP_COPIES = 1024, 2048, 4096, 8192, 16384
ROUTE_QUEUE_HOST, ROUTE_QUEUE_DEVICE = 32768, 65536


def set_route(mask):
    check(lib().bowgpu_debug_set_route(C.c_uint32(mask)))


def get_route():
    m = C.c_uint32(0)
    check(lib().bowgpu_debug_get_route(C.byref(m)))
    return m.value


class route:
    """with capi.route(mask): ...  - the calling thread's calls take the kernels `mask` selects; the previous mask comes back after"""

    def __init__(self, mask):
        self.mask = mask

    def __enter__(self):
        self.prev = get_route()
        set_route(self.mask)
        return self

    def __exit__(self, *exc):
        set_route(self.prev)
        return False


# every kernel / form a Rolling.Aggregate call can be pushed through (the tests run each case through all of them)
AGG_ROUTES = (("auto", 0), ("classic-long", ROUTE_LONG_CLASSIC), ("stream-all", ROUTE_LONG_STREAM_ALL),
              ("small-list", ROUTE_SIMPLE_SMALL_LIST), ("large-list", ROUTE_SIMPLE_LARGE_LIST), ("padded", ROUTE_SIMPLE_PADDED), ("tw-rows", ROUTE_TW_ROWS),
              ("queue-device", ROUTE_NO_LONG_ONLY | ROUTE_QUEUE_DEVICE), ("queue-host", ROUTE_NO_LONG_ONLY | ROUTE_QUEUE_HOST),
              ("lean", ROUTE_NO_SIMPLE | ROUTE_NO_LONG_ONLY), ("general", ROUTE_FORCE_GENERAL | ROUTE_NO_LONG_ONLY))   # ("general" stays last: tests read the last kernel's name)
INTERP_ROUTES = (("wave3", 0), ("tile", ROUTE_INTERP_TILE))   # the product kernel and the one kept second implementation


def agg_routes():
    """yields a label per route with that route in force for the calling thread"""
    for label, mask in AGG_ROUTES:
        with route(mask):
            yield label


# ------------------------------------------------------------------ device buffers
class DeviceBuffer:
    """A hipMalloc'd buffer owned by Python."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().bowgpu_malloc(C.byref(p), C.c_int64(self.nbytes)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(max(arr.nbytes, 8))
        if arr.nbytes:
            check(lib().bowgpu_memcpy_h2d(C.c_void_p(b.ptr), arr.ctypes.data_as(C.c_void_p), C.c_int64(arr.nbytes)))
        return b

    def to_numpy(self, dtype, count, first=0):
        """`count` elements of `dtype` starting at element `first`"""
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            check(lib().bowgpu_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + first * out.itemsize), C.c_int64(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib().bowgpu_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Column:
    """One Arrow array handed to the C ABI: numpy buffers (HOST) or DeviceBuffers (DEVICE)."""

    def __init__(self, values, validity=None, typ=None, offset=0, length=None, null_count=-1):
        if isinstance(values, DeviceBuffer):
            self.residency = DEVICE
            assert typ is not None and length is not None
            self.values, self.validity = values, validity
        else:
            self.residency = HOST
            if typ is None:
                typ = INT64 if np.asarray(values).dtype == np.int64 else FLOAT64
            dt = {INT64: np.int64, FLOAT64: np.float64}.get(typ, np.uint8)
            self.values = np.ascontiguousarray(values, dtype=dt)
            self.validity = None if validity is None else np.ascontiguousarray(validity, dtype=np.uint8)
            if length is None:
                length = len(self.values) - offset
        self.type, self.offset, self.length = typ, offset, length
        self.null_count = 0 if validity is None else null_count

    @classmethod
    def from_list(cls, data, typ):
        """python list with None for nulls (bow.NewBowFromColBasedInterfaces)."""
        if isinstance(typ, str):
            typ = TYPE_NAMES[typ]
        n = len(data)
        valid = np.array([x is not None for x in data], dtype=bool)
        validity = np.packbits(valid, bitorder="little") if n else np.zeros(0, np.uint8)
        if typ == BOOLEAN:
            vals = np.packbits(np.array([bool(x) for x in data], dtype=bool), bitorder="little") if n else np.zeros(0, np.uint8)
            return cls(vals, validity, BOOLEAN, 0, n, int(n - valid.sum()))
        dt = np.int64 if typ == INT64 else np.float64
        vals = np.array([x if x is not None else 0 for x in data], dtype=dt)
        return cls(vals, validity, typ, 0, n, int(n - valid.sum()))

    def pin(self):
        """register the numpy buffers (bowgpu_host_register) and pass the column as HOST_PINNED from now on"""
        assert self.residency != DEVICE
        if self.residency == HOST:
            if self.values.size:
                host_register(self.values)
            if self.validity is not None and self.validity.size:
                host_register(self.validity)
            self.residency = HOST_PINNED
        return self

    def unpin(self):
        if self.residency == HOST_PINNED:
            if self.values.size:
                host_unregister(self.values)
            if self.validity is not None and self.validity.size:
                host_unregister(self.validity)
            self.residency = HOST
        return self

    def to_device(self):
        if self.residency == DEVICE:
            return self
        v = DeviceBuffer.from_numpy(self.values.view(np.uint8) if self.values.size else np.zeros(8, np.uint8))
        b = None if self.validity is None else DeviceBuffer.from_numpy(
            np.concatenate([self.validity, np.zeros(8, np.uint8)]))
        return Column(v, b, self.type, self.offset, self.length, self.null_count)

    def c(self):
        s = Col()
        if self.residency == DEVICE:
            s.values = self.values.ptr
            s.validity = None if self.validity is None else self.validity.ptr
        else:
            s.values = self.values.ctypes.data if self.values.size else None
            s.validity = None if self.validity is None else (self.validity.ctypes.data if self.validity.size else None)
        s.offset, s.length, s.null_count = self.offset, self.length, self.null_count
        s.type, s.residency = self.type, self.residency
        return s


class OutColumn:
    """Caller-owned output storage (what bow.NewBuffer(W, typ) allocates)."""

    def __init__(self, slots, residency=HOST):
        self.slots = slots
        self.residency = residency
        nb = (slots + 7) // 8
        if residency != DEVICE:
            if residency == HOST_PINNED:
                self.values = page_aligned(slots, np.uint64, 0x5A5A5A5A5A5A5A5A)
                self.validity = page_aligned(nb, np.uint8, 0xA5)
                host_register(self.values)
                host_register(self.validity)
            else:
                self.values = np.full(max(slots, 1), 0x5A5A5A5A5A5A5A5A, dtype=np.uint64)  # poisoned
                self.validity = np.full(max(nb, 1), 0xA5, dtype=np.uint8)
        else:
            self.values = DeviceBuffer(max(slots, 1) * 8)
            self.validity = DeviceBuffer(max(nb, 1))
        self.type = 0
        self.null_count = -1
        self.length = slots

    def __del__(self):
        # a registered buffer must be unregistered before its pages go back to the allocator (a later mapping at the same address
        # would otherwise look registered and is not)
        if getattr(self, "residency", None) == HOST_PINNED and _lib is not None:
            try:
                host_unregister(self.values)
                host_unregister(self.validity)
            except Exception:
                pass
            self.residency = HOST

    def c(self):
        o = Out()
        if self.residency != DEVICE:
            o.values, o.validity = self.values.ctypes.data, self.validity.ctypes.data
        else:
            o.values, o.validity = self.values.ptr, self.validity.ptr
        o.length, o.residency = self.slots, self.residency
        return o

    def absorb(self, o):
        self.type, self.null_count, self.length = o.type, o.null_count, o.length

    def host_arrays(self):
        n = self.length
        nb = (n + 7) // 8
        if self.residency != DEVICE:
            vals, bm = self.values[:n], self.validity[:nb]
        else:
            vals, bm = self.values.to_numpy(np.uint64, n), self.validity.to_numpy(np.uint8, nb)
        dt = np.int64 if self.type == INT64 else np.float64
        return vals.view(dt), bm

    def valid_mask(self):
        _, bm = self.host_arrays()
        n = self.length
        if n == 0:
            return np.zeros(0, bool)
        return np.unpackbits(bm, bitorder="little")[:n].astype(bool)

    def to_list(self):
        vals, _ = self.host_arrays()
        m = self.valid_mask()
        conv = int if self.type == INT64 else float
        return [conv(v) if ok else None for v, ok in zip(vals, m)]


def _cols(cols):
    arr = (Col * max(len(cols), 1))()
    for i, c in enumerate(cols):
        arr[i] = c.c()
    return arr


def _aggs(aggs):
    arr = (Agg * max(len(aggs), 1))()
    for i, a in enumerate(aggs):
        arr[i].kind = AGG[a[0]] if isinstance(a[0], str) else a[0]
        arr[i].col = a[1]
        factors = list(a[2]) if len(a) > 2 and a[2] else []
        arr[i].n_factors = len(factors)
        for k, f in enumerate(factors[:MAX_FACTORS]):
            arr[i].factors[k] = f
    return arr


# ------------------------------------------------------------------ entry points
def device_count():
    n = C.c_int(0)
    check(lib().bowgpu_device_count(C.byref(n)))
    return n.value


def set_device(d):
    check(lib().bowgpu_set_device(int(d)))


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().bowgpu_device_name(buf, 256))
    return buf.value.decode()


def set_stream(ptr):
    check(lib().bowgpu_set_stream(C.c_void_p(ptr)))


def synchronize():
    check(lib().bowgpu_synchronize())


def last_kernel_name():
    """the tile kernel of this thread's last aggregate call (without template arguments)"""
    return lib().bowgpu_last_kernel_name().decode().split("<")[0]


def last_kernel_instance():
    """... with them, where the library reports them (rolling_simple_kernel): the instantiation that ran, as rocprofv3 spells it"""
    return lib().bowgpu_last_kernel_name().decode()


def last_call_slow_rows():
    """rows of the thread's last Aggregate / Interpolate call served by rolling_agg_kernel / interp_tile_kernel (0: the fast kernels took it)"""
    r = C.c_int64(0)
    check(lib().bowgpu_last_call_slow_rows(C.byref(r)))
    return r.value


def last_kernel_ms():
    ms = C.c_double(0)
    check(lib().bowgpu_last_kernel_ms(C.byref(ms)))
    return ms.value


def enforce_interval_and_offset(interval, offset):
    out = C.c_int64()
    check(lib().bowgpu_enforce_interval_and_offset(C.c_int64(interval), C.c_int64(offset), C.byref(out)))
    return out.value


def plan_windows(ts, interval, offset=0):
    s0, W = C.c_int64(), C.c_int64()
    c = ts.c()
    check(lib().bowgpu_plan_windows(C.byref(c), C.c_int64(interval), C.c_int64(offset), C.byref(s0), C.byref(W)))
    return s0.value, W.value


def plan_windows_ex(ts, interval, offset=0):
    """bowgpu_plan_windows_ex -> Plan (what newIntervalRolling keeps in the Rolling: one round trip to the device, once)"""
    p = Plan()
    c = ts.c()
    check(lib().bowgpu_plan_windows_ex(C.byref(c), C.c_int64(interval), C.c_int64(offset), C.byref(p)))
    return p


def rolling_aggregate(cols, ts_col, interval, aggs, offset=0, inclusive=False, out_residency=HOST, outs=None, plan=None, strict_order=False):
    """Returns (list[OutColumn], AggInfo).  aggs: [(kind, col[, factors])].  plan: a Plan from plan_windows_ex on this
    interval column (the call then skips its own round trip for the first / last timestamp).  strict_order: every window in the
    reference's row order or BOWGPU_ERR_UNSUPPORTED (bowgpu_options.strict_order)."""
    if outs is None:
        W = plan.num_windows if plan is not None else plan_windows(cols[ts_col], interval, offset)[1]
        outs = [OutColumn(W, out_residency) for _ in aggs]
    oarr = (Out * max(len(aggs), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    opts = Options(offset, int(bool(inclusive)), int(bool(strict_order)))
    info = AggInfo()
    if plan is not None:
        check(lib().bowgpu_rolling_aggregate_planned(_cols(cols), len(cols), ts_col, C.byref(plan), C.byref(opts),
                                                     _aggs(aggs), len(aggs), oarr, C.byref(info)))
    else:
        check(lib().bowgpu_rolling_aggregate(_cols(cols), len(cols), ts_col, C.c_int64(interval), C.byref(opts),
                                             _aggs(aggs), len(aggs), oarr, C.byref(info)))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs, info


def window_bounds(ts, interval, offset=0, inclusive=False):
    s0, W = plan_windows(ts, interval, offset)
    arrs = [np.zeros(max(W, 1), dtype=np.int64) for _ in range(3)]
    inc = np.zeros(max(W, 1), dtype=np.uint8)
    opts = Options(offset, int(bool(inclusive)), 0)
    c = ts.c()
    check(lib().bowgpu_window_bounds(C.byref(c), C.c_int64(interval), C.byref(opts),
                                     *[a.ctypes.data_as(C.c_void_p) for a in arrs], inc.ctypes.data_as(C.c_void_p), HOST))
    return s0, W, arrs[0][:W], arrs[1][:W], arrs[2][:W], inc[:W].astype(bool)


def aggregate_whole(cols, ts_col, aggs):
    outs = [OutColumn(1) for _ in aggs]
    oarr = (Out * max(len(aggs), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    check(lib().bowgpu_aggregate_whole(_cols(cols), len(cols), ts_col, _aggs(aggs), len(aggs), oarr))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs


def _interps(interps):
    arr = (Interp * max(len(interps), 1))()
    for i, ip in enumerate(interps):
        arr[i].kind, arr[i].col = INTERP[ip["kind"]], ip["col"]
        arr[i].const_value = ip.get("const", 0.0)
        prev = ip.get("prev")
        if prev is not None:
            arr[i].has_prev_row = 1
            arr[i].prev_t, arr[i].prev_t_valid = prev[0], int(prev[1])
            arr[i].prev_v, arr[i].prev_v_valid = prev[2], int(prev[3])
            arr[i].prev_v_i64 = prev[4] if len(prev) > 4 else 0
    return arr


def rolling_interpolate_aggregate(cols, ts_col, interval, interps, aggs, offset=0, inclusive=False, out_residency=HOST, outs=None, strict_order=False):
    """r.Interpolate(interps...).Aggregate(aggs...) without the interpolated frame (bowgpu_rolling_interpolate_aggregate): returns
    (list[OutColumn], AggInfo) as rolling_aggregate does.  The window grid of the interpolated frame is the input's, so the outputs hold
    plan_windows(...)[1] slots."""
    if outs is None:
        W = plan_windows(cols[ts_col], interval, offset)[1]
        outs = [OutColumn(W, out_residency) for _ in aggs]
    oarr = (Out * max(len(aggs), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    opts = Options(offset, int(bool(inclusive)), int(bool(strict_order)))
    info = AggInfo()
    check(lib().bowgpu_rolling_interpolate_aggregate(_cols(cols), len(cols), ts_col, C.c_int64(interval), C.byref(opts),
                                                     _interps(interps), len(interps), _aggs(aggs), len(aggs), oarr, C.byref(info)))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs, info


_interp_out_override = None


@contextlib.contextmanager
def interp_outputs(residency, padded):
    """tests: every rolling_interpolate() inside allocates its outputs with this residency / capacity rule, whatever it asks for"""
    global _interp_out_override
    old, _interp_out_override = _interp_out_override, (residency, padded)
    try:
        yield
    finally:
        _interp_out_override = old


def rolling_interpolate(cols, ts_col, interval, interps, offset=0, inclusive=False, out_residency=HOST, padded=True):
    """count -> allocate -> fill.  padded: the outputs' capacity is the row count rounded up to 512 rows, i.e. bitmaps of whole
    64-byte blocks like the Arrow allocator's (memory.NewGoAllocator pads to 64 bytes) - device-resident bitmaps that reach the end of
    their last 32-bit word are written in place (include/bowgpu.h at bowgpu_rolling_interpolate_fill); False: exactly the rows."""
    if _interp_out_override is not None:
        out_residency, padded = _interp_out_override
    opts = Options(offset, int(bool(inclusive)), 0)
    n_out = C.c_int64(0)
    carr, iarr = _cols(cols), _interps(interps)
    check(lib().bowgpu_rolling_interpolate_count(carr, len(cols), ts_col, C.c_int64(interval), C.byref(opts),
                                                 iarr, len(interps), C.byref(n_out)))
    outs = [OutColumn((n_out.value + 511) // 512 * 512 if padded else n_out.value, out_residency) for _ in interps]
    oarr = (Out * max(len(interps), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    check(lib().bowgpu_rolling_interpolate_fill(carr, len(cols), ts_col, C.c_int64(interval), C.byref(opts),
                                                iarr, len(interps), oarr))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs


def rolling_interpolate_onepass(cols, ts_col, interval, interps, offset=0, inclusive=False, out_residency=HOST, capacity=None, outs=None):
    """Rolling.Interpolate as ONE call: bowgpu_rolling_interpolate_fill without a preceding _count.  The output buffers are sized by
    the caller (default: rows + windows, the most any call adds); the call sets their length."""
    opts = Options(offset, int(bool(inclusive)), 0)
    if capacity is None:
        capacity = cols[ts_col].length + (plan_windows(cols[ts_col], interval, offset)[1] if cols[ts_col].length else 0)
    carr, iarr = _cols(cols), _interps(interps)
    if outs is None:
        outs = [OutColumn(capacity, out_residency) for _ in interps]
    oarr = (Out * max(len(interps), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    check(lib().bowgpu_rolling_interpolate_fill(carr, len(cols), ts_col, C.c_int64(interval), C.byref(opts), iarr, len(interps), oarr))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs


def shard_interp_points(cols, ts_col):
    """this shard's first / last valid point per column (bytes of bowgpu_interp_points: travels through an all_gather)"""
    pts = InterpPoints()
    check(lib().bowgpu_shard_interp_points(_cols(cols), len(cols), ts_col, C.byref(pts)))
    return bytes(pts)


def shard_interpolate(cols, ts_col, interval, interps, global_s0, rank, all_points, offset=0, out_residency=HOST, inclusive=False):
    """Rolling.Interpolate of one row-range shard.  all_points: every rank's shard_interp_points bytes, in rank order.  The
    shards' outputs concatenated in rank order equal the unsharded result."""
    pts = [InterpPoints.from_buffer_copy(b) for b in all_points]
    edge = InterpEdge()
    left = [q for q in range(rank) if pts[q].nrows > 0]
    if left:
        edge.has_left, edge.left_last_ts = 1, pts[left[-1]].last_ts
    ips = [dict(ip) for ip in interps]
    for i in range(len(cols)):
        for q in reversed(left):              # nearest valid point on the left: the reference's PrevRow mechanism carries it
            if pts[q].last_valid[i]:
                ips[i]["prev"] = (pts[q].last_t[i], True, pts[q].last_v[i], True, pts[q].last_v_i64[i])
                break
        for q in range(rank + 1, len(pts)):   # nearest valid point on the right
            if pts[q].nrows > 0 and pts[q].first_valid[i]:
                edge.next_valid[i], edge.next_t[i], edge.next_v[i] = 1, pts[q].first_t[i], pts[q].first_v[i]
                break
    opts = Options(offset, int(bool(inclusive)), 0)
    carr, iarr = _cols(cols), _interps(ips)
    n_out = C.c_int64(0)
    check(lib().bowgpu_shard_interpolate_count(carr, len(cols), ts_col, C.c_int64(interval), C.byref(opts), C.c_int64(global_s0),
                                               iarr, len(ips), C.byref(edge), C.byref(n_out)))
    outs = [OutColumn(n_out.value, out_residency) for _ in ips]
    oarr = (Out * max(len(ips), 1))()
    for i, o in enumerate(outs):
        oarr[i] = o.c()
    check(lib().bowgpu_shard_interpolate_fill(carr, len(cols), ts_col, C.c_int64(interval), C.byref(opts), C.c_int64(global_s0),
                                              iarr, len(ips), C.byref(edge), oarr))
    for i, o in enumerate(outs):
        o.absorb(oarr[i])
    return outs


def fill_linear(cols, ref_col, fill_col, out_residency=HOST, capacity=None, ref_checked=False):
    """capacity: rows the output buffers can hold (default: exactly the column's) - a device-resident bitmap that reaches the end of its
    last 64-bit word and is 8-byte aligned is written in place (bow_amd/csrc/extras.cpp fill_finish)"""
    out = OutColumn(cols[fill_col].length if capacity is None else capacity, out_residency)
    o = out.c()
    unchanged = C.c_int32(0)
    fn = lib().bowgpu_fill_linear_sorted if ref_checked else lib().bowgpu_fill_linear   # (ref_checked: the caller ran bowfill.go:35-42 itself)
    check(fn(_cols(cols), len(cols), ref_col, fill_col, C.byref(o), C.byref(unchanged)))
    out.absorb(o)
    return out, bool(unchanged.value)


FILL = {"Previous": 0, "Next": 1, "Mean": 2}


def fill(col, method, out_residency=HOST, capacity=None):
    """Bow.FillPrevious / FillNext / FillMean of one column -> (OutColumn, unchanged); capacity: see fill_linear"""
    out = OutColumn(col.length if capacity is None else capacity, out_residency)
    o = out.c()
    c = col.c()
    unchanged = C.c_int32(0)
    check(lib().bowgpu_fill(C.byref(c), FILL[method], C.byref(o), C.byref(unchanged)))
    out.absorb(o)
    return out, bool(unchanged.value)


def is_col_sorted(col):
    c = col.c()
    s = C.c_int32(0)
    check(lib().bowgpu_is_col_sorted(C.byref(c), C.byref(s)))
    return bool(s.value)


def gen_dense(row0, n, seed=42):
    ts, val = DeviceBuffer(max(n, 1) * 8), DeviceBuffer(max(n, 1) * 8)
    check(lib().bowgpu_gen_dense(C.c_int64(row0), C.c_int64(n), C.c_uint64(seed), C.c_void_p(ts.ptr), C.c_void_p(val.ptr)))
    return Column(ts, None, INT64, 0, n, 0), Column(val, None, FLOAT64, 0, n, 0)


def gen_sparse(row0, n, seed=42):
    ts, val = DeviceBuffer(max(n, 1) * 8), DeviceBuffer(max(n, 1) * 8)
    bm = DeviceBuffer((n + 7) // 8 + 8)
    check(lib().bowgpu_memset(C.c_void_p(bm.ptr), 0, C.c_int64(bm.nbytes)))
    check(lib().bowgpu_gen_sparse(C.c_int64(row0), C.c_int64(n), C.c_uint64(seed), C.c_void_p(ts.ptr),
                                  C.c_void_p(val.ptr), C.c_void_p(bm.ptr)))
    return Column(ts, None, INT64, 0, n, 0), Column(val, bm, FLOAT64, 0, n, -1)


def stream_read_ceiling(buf_a, buf_b, bytes_each):
    """best GB/s of a trivial streaming sum over two device buffers (the achievable line of the roofline)"""
    g = C.c_double(0)
    check(lib().bowgpu_stream_read_ceiling(C.c_void_p(buf_a.ptr), C.c_void_p(buf_b.ptr), C.c_int64(bytes_each), C.byref(g)))
    return g.value


def stream_rw_probe(buf_a, buf_b, bytes_each, out_a, out_b, rows_per_slot):
    """(read GB/s, ms) of a trivial kernel with the benched kernel's traffic mix: two input streams read, two output streams of
    one 8-byte slot per rows_per_slot rows written (the achievable line for reads AND writes together)"""
    g, ms = C.c_double(0), C.c_double(0)
    check(lib().bowgpu_stream_rw_probe(C.c_void_p(buf_a.ptr), C.c_void_p(buf_b.ptr), C.c_int64(bytes_each), C.c_void_p(out_a.ptr),
                                         C.c_void_p(out_b.ptr), C.c_int64(rows_per_slot), C.byref(g), C.byref(ms)))
    return g.value, ms.value


class ParquetFile:
    """Parquet column chunks decoded on the device (bowgpu_parquet_*): the reference's NewBowFromParquet for INT64 / DOUBLE columns"""

    def __init__(self, path):
        self.h = C.c_void_p()
        check(lib().bowgpu_parquet_open(path.encode(), C.byref(self.h)))
        n, k = C.c_int64(0), C.c_int32(0)
        check(lib().bowgpu_parquet_info(self.h, C.byref(n), C.byref(k)))
        self.num_rows, self.num_columns = n.value, k.value
        self.columns = []
        for i in range(k.value):
            name = C.create_string_buffer(256)
            t, opt = C.c_int32(0), C.c_int32(0)
            check(lib().bowgpu_parquet_column(self.h, i, name, 256, C.byref(t), C.byref(opt)))
            self.columns.append((name.value.decode(), t.value, bool(opt.value)))

    def read_column(self, i, out_residency=HOST):
        out = OutColumn(self.num_rows, out_residency)
        o = out.c()
        check(lib().bowgpu_parquet_read_column(self.h, i, C.byref(o)))
        out.absorb(o)
        return out

    def close(self):
        if self.h:
            lib().bowgpu_parquet_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def checksum64(devbuf, n_words, index_base=0, word_offset=0):
    """(xor, sum) of n_words 8-byte words starting word_offset words into devbuf, hashed as words index_base.. of a larger array"""
    x, s = C.c_uint64(0), C.c_uint64(0)
    check(lib().bowgpu_checksum64_at(C.c_void_p(devbuf.ptr + 8 * word_offset), C.c_int64(n_words), C.c_int64(index_base), C.byref(x), C.byref(s)))
    return x.value, s.value


class Timer:
    def __init__(self):
        p = C.c_void_p()
        check(lib().bowgpu_timer_create(C.byref(p)))
        self.p = p

    def start(self):
        check(lib().bowgpu_timer_start(self.p))

    def stop(self):
        check(lib().bowgpu_timer_stop(self.p))

    def elapsed_ms(self):
        ms = C.c_double(0)
        check(lib().bowgpu_timer_elapsed_ms(self.p, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().bowgpu_timer_destroy(self.p)
        except Exception:
            pass
