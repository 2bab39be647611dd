// bow_rolling.hpp — C++ mirror of the reference's Go interface for the rolling path, above the C ABI.
//
// The reference is Go (no Go toolchain in this image), so the host side above include/bowgpu.h is
// written in C++ with the SAME names, argument meaning and error strings as the reference:
//   bow::Bow / Series / Type                   <- bow.go, bowseries.go, bowtypes.go
//   bow::rolling::IntervalRolling, Rolling, Options, Window, ColAggregation, NewColAggregation,
//        ColInterpolation, NewColInterpolation  <- rolling/rolling.go, window.go, aggregation.go, interpolation.go
//   bow::rolling::aggregation::{WindowStart,Sum,ArithmeticMean,Min,Max,Count,First,Last,Mode,IntegralStep,
//        IntegralTrapezoid,WeightedAverageStep,WeightedAverageLinear}   <- rolling/aggregation/*.go
//   bow::rolling::interpolation::{WindowStart,Linear,StepPrevious,None} <- rolling/interpolation/*.go
//   bow::rolling::transformation::Factor                                 <- rolling/transformation/factor.go
// Go's (value, error) returns become std::pair<T, Error>; interface{} becomes bow::Value.
//
// Everything that touches column data goes through libbowgpu.so (HIP kernels).  The only host loops are
// the user's OWN closures (NewColAggregation with a custom func), which are called per window with
// Window slices whose row ranges come from the device (bowgpu_window_bounds).
#pragma once

#include <stdint.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/bowgpu.h"

namespace bow {

// ----------------------------------------------------------------------------- errors, values, types
struct Error {
    bool set = false;
    std::string msg;
    Error() = default;
    explicit Error(std::string m) : set(true), msg(std::move(m)) {}
    explicit operator bool() const { return set; }
    const std::string &Error_() const { return msg; }
};
inline Error Errorf(const std::string &m) { return Error(m); }
inline Error Wrap(const std::string &prefix, const Error &e) { return Error(prefix + ": " + e.msg); }

using Scalar = std::variant<int64_t, double, bool>;
using Value = std::optional<Scalar>;  // interface{}: nil | int64 | float64 | bool
inline Value Nil() { return std::nullopt; }

enum class Type : int32_t { Unknown = 0, Float64 = 1, Int64 = 2, Boolean = 3, String = 4, InputDependent = 5, IteratorDependent = 6 };
constexpr Type Float64 = Type::Float64, Int64 = Type::Int64, Boolean = Type::Boolean, String = Type::String,
               InputDependent = Type::InputDependent, IteratorDependent = Type::IteratorDependent;

inline std::string TypeString(Type t) {  // bowtypes.go:89-96 (arrow type names)
    switch (t) {
    case Type::Float64: return "float64";
    case Type::Int64: return "int64";
    case Type::Boolean: return "bool";
    case Type::String: return "utf8";
    default: return "undefined";
    }
}

inline int64_t GoInt64(double x) {  // float64 -> int64 on amd64
    if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)x;
}

// ----------------------------------------------------------------------------- Series / Bow
struct Series {
    std::string Name;
    Type typ = Type::Unknown;
    std::vector<uint64_t> data;     // 8-byte slots: int64 or float64 bit patterns (Arrow values buffer)
    std::vector<uint8_t> validity;  // Arrow LSB-first bitmap, ceil(n/8) bytes (bowseries.go:191-220)
    int64_t length = 0;

    bool IsValid(int64_t i) const { return (validity[i >> 3] >> (i & 7)) & 1; }
    int64_t NullN() const {
        int64_t c = 0;
        for (int64_t i = 0; i < length; i++) c += !IsValid(i);
        return c;
    }
};

inline Series NewSeriesRaw(const std::string &name, Type typ, std::vector<uint64_t> data, std::vector<uint8_t> validity) {
    Series s;
    s.Name = name; s.typ = typ; s.length = (int64_t)data.size(); s.data = std::move(data); s.validity = std::move(validity);
    s.validity.resize((size_t)((s.length + 7) / 8));
    return s;
}

// NewSeries(name, typ, dataArray, validityArray): validity empty => all valid (bowseries.go:27-29, :196-200)
template <typename T>
inline Series NewSeries(const std::string &name, Type typ, const std::vector<T> &values, const std::vector<bool> &valid = {}) {
    std::vector<uint64_t> d(values.size());
    for (size_t i = 0; i < values.size(); i++) {
        if (typ == Type::Int64) { int64_t v = (int64_t)values[i]; memcpy(&d[i], &v, 8); }
        else { double v = (double)values[i]; memcpy(&d[i], &v, 8); }
    }
    std::vector<uint8_t> bm((values.size() + 7) / 8, 0);
    for (size_t i = 0; i < values.size(); i++)
        if (valid.empty() || valid[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
    return NewSeriesRaw(name, typ, std::move(d), std::move(bm));
}

class Bow;
using BowPtr = std::shared_ptr<const Bow>;

class Bow {
  public:
    std::vector<Series> cols;
    // a slice shares nothing here (copies): slices are only made for the user's closures
    int NumCols() const { return (int)cols.size(); }
    int NumRows() const { return cols.empty() ? 0 : (int)cols[0].length; }
    const std::string &ColumnName(int i) const { return cols[i].Name; }
    Type ColumnType(int i) const { return cols[i].typ; }

    // ColumnIndex: bowgetters.go:320-331
    std::pair<int, Error> ColumnIndex(const std::string &name) const {
        int found = -1, n = 0;
        for (int i = 0; i < NumCols(); i++)
            if (cols[i].Name == name) { if (found < 0) found = i; n++; }
        if (n == 0) return {-1, Errorf("no column '" + name + "'")};
        if (n > 1) return {-1, Errorf("several columns '" + name + "'")};
        return {found, Error()};
    }

    // GetValue: bowgetters.go:46-63
    Value GetValue(int col, int row) const {
        const Series &s = cols[col];
        if (row < 0 || row >= s.length || !s.IsValid(row)) return Nil();
        if (s.typ == Type::Int64) { int64_t v; memcpy(&v, &s.data[row], 8); return Scalar(v); }
        double v; memcpy(&v, &s.data[row], 8); return Scalar(v);
    }
    // GetFloat64: bowgetters.go:218-247
    std::pair<double, bool> GetFloat64(int col, int row) const {
        const Series &s = cols[col];
        if (row < 0 || row >= s.length) return {0., false};
        if (s.typ == Type::Int64) { int64_t v; memcpy(&v, &s.data[row], 8); return {(double)v, s.IsValid(row)}; }
        double v; memcpy(&v, &s.data[row], 8); return {v, s.IsValid(row)};
    }
    std::pair<int64_t, bool> GetInt64(int col, int row) const {
        const Series &s = cols[col];
        if (row < 0 || row >= s.length) return {0, false};
        if (s.typ == Type::Int64) { int64_t v; memcpy(&v, &s.data[row], 8); return {v, s.IsValid(row)}; }
        double v; memcpy(&v, &s.data[row], 8); return {GoInt64(v), s.IsValid(row)};
    }

    // NewSlice: bow.go:279-283
    BowPtr NewSlice(int i, int j) const {
        auto b = std::make_shared<Bow>();
        for (const Series &s : cols) {
            Series t;
            t.Name = s.Name; t.typ = s.typ; t.length = j - i;
            t.data.assign(s.data.begin() + i, s.data.begin() + j);
            t.validity.assign((size_t)((t.length + 7) / 8), 0);
            for (int r = i; r < j; r++)
                if (s.IsValid(r)) t.validity[(r - i) >> 3] |= (uint8_t)(1u << ((r - i) & 7));
            b->cols.push_back(std::move(t));
        }
        return b;
    }
    BowPtr NewEmptySlice() const { return NewSlice(0, 0); }

    // Equal: bow.go:227-275 (schema + per-row values; floats compared exactly)
    bool Equal(const Bow &o) const {
        if (NumCols() != o.NumCols() || NumRows() != o.NumRows()) return false;
        for (int c = 0; c < NumCols(); c++) {
            if (cols[c].Name != o.cols[c].Name || cols[c].typ != o.cols[c].typ) return false;
            for (int r = 0; r < NumRows(); r++) {
                Value a = GetValue(c, r), b = o.GetValue(c, r);
                if (a.has_value() != b.has_value()) return false;
                if (a && *a != *b) return false;
            }
        }
        return true;
    }

    std::string String() const {
        std::ostringstream os;
        for (int c = 0; c < NumCols(); c++) {
            os << cols[c].Name << ":" << TypeString(cols[c].typ) << " [";
            for (int r = 0; r < NumRows(); r++) {
                Value v = GetValue(c, r);
                if (r) os << " ";
                if (!v) os << "<nil>";
                else if (std::holds_alternative<int64_t>(*v)) os << std::get<int64_t>(*v);
                else if (std::holds_alternative<double>(*v)) os << std::get<double>(*v);
                else os << (std::get<bool>(*v) ? "true" : "false");
            }
            os << "]\n";
        }
        return os.str();
    }

    // Arrow view of a column for the C ABI (what b.ArrowRecord().Column(i).Data() exposes: bow.go:183-186)
    bowgpu_col ArrowCol(int i) const {
        const Series &s = cols[i];
        bowgpu_col c;
        memset(&c, 0, sizeof c);
        c.values = s.data.empty() ? nullptr : s.data.data();
        c.validity = s.validity.empty() ? nullptr : s.validity.data();
        c.offset = 0; c.length = s.length; c.null_count = s.NullN();
        c.type = (int32_t)s.typ; c.residency = BOWGPU_HOST;
        return c;
    }

    // FillLinear: bowfill.go:14-103 (device) ; IsColSorted: bowassertion.go:15-81 (device)
    std::pair<BowPtr, Error> FillLinear(int refColIndex, int toFillColIndex) const;
    bool IsColSorted(int colIndex) const;
    // FillPrevious / FillNext: bowfill.go:162-253 ; FillMean: bowfill.go:105-160 (device; colIndices defaults to all columns)
    std::pair<BowPtr, Error> FillPrevious(std::vector<int> colIndices = {}) const { return fill(BOWGPU_FILL_PREVIOUS, colIndices); }
    std::pair<BowPtr, Error> FillNext(std::vector<int> colIndices = {}) const { return fill(BOWGPU_FILL_NEXT, colIndices); }
    std::pair<BowPtr, Error> FillMean(std::vector<int> colIndices = {}) const { return fill(BOWGPU_FILL_MEAN, colIndices); }

private:
    std::pair<BowPtr, Error> fill(int method, const std::vector<int> &colIndices) const;
};

// NewBow: bow.go:109-116 (all series must have the same length)
inline std::pair<BowPtr, Error> NewBow(std::vector<Series> series) {
    auto b = std::make_shared<Bow>();
    for (size_t i = 1; i < series.size(); i++)
        if (series[i].length != series[0].length) return {nullptr, Errorf("bow.NewBow: Series have different lengths")};
    b->cols = std::move(series);
    return {b, Error()};
}

// NewBowFromColBasedInterfaces: one vector<Value> per column, nil = null
inline std::pair<BowPtr, Error> NewBowFromColBasedInterfaces(const std::vector<std::string> &names, const std::vector<Type> &types,
                                                              const std::vector<std::vector<Value>> &colsv) {
    std::vector<Series> ss;
    for (size_t c = 0; c < names.size(); c++) {
        std::vector<uint64_t> d(colsv[c].size(), 0);
        std::vector<uint8_t> bm((colsv[c].size() + 7) / 8, 0);
        for (size_t r = 0; r < colsv[c].size(); r++) {
            const Value &v = colsv[c][r];
            if (!v) continue;
            if (types[c] == Type::Int64) {
                int64_t x = std::holds_alternative<int64_t>(*v) ? std::get<int64_t>(*v) : GoInt64(std::get<double>(*v));
                memcpy(&d[r], &x, 8);
            } else {
                double x = std::holds_alternative<double>(*v) ? std::get<double>(*v) : (double)std::get<int64_t>(*v);
                memcpy(&d[r], &x, 8);
            }
            bm[r >> 3] |= (uint8_t)(1u << (r & 7));
        }
        ss.push_back(NewSeriesRaw(names[c], types[c], std::move(d), std::move(bm)));
    }
    return NewBow(std::move(ss));
}

inline std::pair<BowPtr, Error> NewBowFromRowBasedInterfaces(const std::vector<std::string> &names, const std::vector<Type> &types,
                                                              const std::vector<std::vector<Value>> &rows) {
    std::vector<std::vector<Value>> colsv(names.size());
    for (const auto &row : rows)
        for (size_t c = 0; c < names.size(); c++) colsv[c].push_back(row[c]);
    return NewBowFromColBasedInterfaces(names, types, colsv);
}

// convenience literals for the table-driven tests: I(10), F(1.5), N
inline Value I(int64_t v) { return Scalar(v); }
inline Value F(double v) { return Scalar(v); }
static const Value N = std::nullopt;

namespace detail {
inline Error AbiError(int rc) { (void)rc; return Error(bowgpu_last_error()); }

// caller-owned output storage: what bow.NewBuffer(W, typ) allocates (bowbuffer.go:22-40)
struct OutStore {
    std::vector<uint64_t> data;
    std::vector<uint8_t> validity;
    bowgpu_out Make(int64_t slots) {
        data.assign((size_t)slots + 1, 0);
        validity.assign((size_t)((slots + 7) / 8) + 1, 0);
        bowgpu_out o;
        memset(&o, 0, sizeof o);
        o.values = data.data(); o.validity = validity.data(); o.length = slots; o.residency = BOWGPU_HOST;
        return o;
    }
    Series ToSeries(const std::string &name, const bowgpu_out &o) {
        data.resize((size_t)o.length);
        validity.resize((size_t)((o.length + 7) / 8));
        return NewSeriesRaw(name, (Type)o.type, data, validity);
    }
};
}  // namespace detail

inline std::pair<BowPtr, Error> Bow::FillLinear(int refColIndex, int toFillColIndex) const {
    if (refColIndex < 0 || refColIndex > NumCols() - 1) return {nullptr, Errorf("refColIndex is out of range")};
    if (toFillColIndex < 0 || toFillColIndex > NumCols() - 1) return {nullptr, Errorf("toFillColIndex is out of range")};
    std::vector<bowgpu_col> c;
    for (int i = 0; i < NumCols(); i++) c.push_back(ArrowCol(i));
    detail::OutStore st;
    bowgpu_out o = st.Make(NumRows());
    int32_t unchanged = 0;
    int rc = bowgpu_fill_linear(c.data(), NumCols(), refColIndex, toFillColIndex, &o, &unchanged);
    if (rc) return {nullptr, detail::AbiError(rc)};
    auto out = std::make_shared<Bow>(*this);
    if (!unchanged) out->cols[toFillColIndex] = st.ToSeries(cols[toFillColIndex].Name, o);
    return {out, Error()};
}

// NewBowFromParquet: bowparquet.go:44-153, with the columns decoded on the device (bowgpu_parquet_*).  This mirror holds
// Int64 / Float64 columns; the reference's Boolean / String columns are outside the device path and are skipped.
inline std::pair<BowPtr, Error> NewBowFromParquet(const std::string &path) {
    bowgpu_parquet *h = nullptr;
    int rc = bowgpu_parquet_open(path.c_str(), &h);
    if (rc) return {nullptr, Wrap("bow.NewBowFromParquet", detail::AbiError(rc))};
    int64_t rows = 0;
    int32_t ncols = 0;
    bowgpu_parquet_info(h, &rows, &ncols);
    std::vector<Series> series;
    for (int32_t i = 0; i < ncols && !rc; i++) {
        char name[256];
        int32_t type = -1, optional = 0;
        bowgpu_parquet_column(h, i, name, (int32_t)sizeof name, &type, &optional);
        if (type != BOWGPU_INT64 && type != BOWGPU_FLOAT64) continue;
        detail::OutStore st;
        bowgpu_out o = st.Make(rows);
        rc = bowgpu_parquet_read_column(h, i, &o);
        if (!rc) series.push_back(st.ToSeries(name, o));
    }
    const Error err = rc ? Wrap("bow.NewBowFromParquet", detail::AbiError(rc)) : Error();
    bowgpu_parquet_close(h);
    if (rc) return {nullptr, err};
    return NewBow(std::move(series));
}

inline std::pair<BowPtr, Error> Bow::fill(int method, const std::vector<int> &colIndices) const {
    std::vector<bool> selected(NumCols(), colIndices.empty());  // selectCols: bowfill.go:268-288
    for (int ci : colIndices) {
        if (ci < 0 || ci > NumCols() - 1) return {nullptr, Errorf("selectCols: colIndex '" + std::to_string(ci) + "' out of range")};
        selected[ci] = true;
    }
    for (int ci = 0; ci < NumCols(); ci++)  // FillMean checks types first (bowfill.go:114-125); this mirror holds Int64 / Float64 only
        if (selected[ci] && cols[ci].typ != Type::Int64 && cols[ci].typ != Type::Float64)
            return {nullptr, Errorf("column '" + cols[ci].Name + "' is of unsupported type '" + TypeString(cols[ci].typ) + "'")};
    auto out = std::make_shared<Bow>(*this);
    for (int ci = 0; ci < NumCols(); ci++) {
        if (!selected[ci] || cols[ci].NullN() == 0) continue;  // passed through: bowfill.go:130-133, :186-189
        bowgpu_col c = ArrowCol(ci);
        detail::OutStore st;
        bowgpu_out o = st.Make(NumRows());
        int32_t unchanged = 0;
        int rc = bowgpu_fill(&c, method, &o, &unchanged);
        if (rc) return {nullptr, detail::AbiError(rc)};
        out->cols[ci] = st.ToSeries(cols[ci].Name, o);
    }
    return {out, Error()};
}

inline bool Bow::IsColSorted(int colIndex) const {
    bowgpu_col c = ArrowCol(colIndex);
    int32_t s = 0;
    if (bowgpu_is_col_sorted(&c, &s)) return false;
    return s != 0;
}

// ============================================================================ rolling
namespace rolling {

namespace transformation {
// Func: rolling/transformation/factor.go:5 ; Factor: :7-20.  A Factor carries its multiplier so the device
// can fuse it; any other Func is applied to the reducer's results on the host afterwards.
struct Func {
    std::function<std::pair<Value, Error>(Value)> fn;
    bool is_factor = false;
    double factor = 1.0;
    std::pair<Value, Error> operator()(Value x) const { return fn(std::move(x)); }
};
inline Func Factor(double n) {
    Func f;
    f.is_factor = true;
    f.factor = n;
    f.fn = [n](Value x) -> std::pair<Value, Error> {
        if (!x) return {x, Error()};
        if (std::holds_alternative<double>(*x)) return {Scalar(std::get<double>(*x) * n), Error()};
        if (std::holds_alternative<int64_t>(*x)) return {Scalar(GoInt64((double)std::get<int64_t>(*x) * n)), Error()};
        return {Nil(), Errorf("factor: invalid type bool")};
    };
    return f;
}
}  // namespace transformation

// Options: rolling.go:49-53
struct Options {
    int64_t Offset = 0;
    bool Inclusive = false;
    BowPtr PrevRow;
};

// Window: window.go:12-19
struct Window {
    BowPtr Bow;
    int FirstIndex = 0;
    int IntervalColIndex = 0;
    int64_t FirstValue = 0;
    int64_t LastValue = 0;
    bool IsInclusive = false;
    // UnsetInclusive: window.go:23-31
    Window UnsetInclusive() const {
        if (!IsInclusive) return *this;
        Window w = *this;
        w.IsInclusive = false;
        w.Bow = w.Bow->NewSlice(0, w.Bow->NumRows() - 1);
        return w;
    }
};

using ColAggregationFunc = std::function<std::pair<Value, Error>(int colIndex, const Window &w)>;

// ColAggregation: aggregation.go:11-61 (value type with shared state, like the Go pointer receiver)
class ColAggregationImpl {
  public:
    std::string inputName;
    int inputIndex = -1;
    bool needInclusiveWindow = false;
    ColAggregationFunc aggregationFn;
    std::vector<transformation::Func> transformationFns;
    std::string outputName;
    Type typ = Type::Unknown;
    int32_t gpuKind = -1;  // BOWGPU_AGG_* for the built-in constructors, -1 for user closures
};

class ColAggregation {
  public:
    std::shared_ptr<ColAggregationImpl> p;
    ColAggregation() : p(std::make_shared<ColAggregationImpl>()) {}
    const std::string &InputName() const { return p->inputName; }
    int InputIndex() const { return p->inputIndex; }
    void SetInputIndex(int i) const { p->inputIndex = i; }  // mutates the caller's aggregator (aggregation.go:181)
    const std::string &OutputName() const { return p->outputName; }
    ColAggregation RenameOutput(const std::string &name) const {  // returns a copy: aggregation.go:82-86
        ColAggregation c;
        *c.p = *p;
        c.p->outputName = name;
        return c;
    }
    bool NeedInclusiveWindow() const { return p->needInclusiveWindow; }
    Type GetType() const { return p->typ; }
    Type GetReturnType(Type inputType, Type iteratorType) const {  // aggregation.go:110-121
        if (p->typ == Type::InputDependent) return inputType;
        if (p->typ == Type::IteratorDependent) return iteratorType;
        return p->typ;
    }
    const ColAggregationFunc &Func() const { return p->aggregationFn; }
    const std::vector<transformation::Func> &Transformations() const { return p->transformationFns; }
    ColAggregation SetTransformations(std::vector<transformation::Func> t) const {  // copy: aggregation.go:104-108
        ColAggregation c;
        *c.p = *p;
        c.p->transformationFns = std::move(t);
        return c;
    }
    int32_t GPUKind() const { return p->gpuKind; }
};

// NewColAggregation: aggregation.go:53-61
inline ColAggregation NewColAggregation(const std::string &inputName, bool needInclusiveWindow, Type typ, ColAggregationFunc fn) {
    ColAggregation a;
    a.p->inputName = inputName;
    a.p->needInclusiveWindow = needInclusiveWindow;
    a.p->typ = typ;
    a.p->aggregationFn = std::move(fn);
    return a;
}
using ColAggregationConstruct = std::function<ColAggregation(const std::string &)>;

// ColInterpolation: interpolation.go:10-28
using ColInterpolationFunc = std::function<std::pair<Value, Error>(int colIndex, const Window &w, const Bow &fullBow, BowPtr prevRow)>;
struct ColInterpolation {
    std::string colName;
    std::vector<Type> inputTypes;
    ColInterpolationFunc fn;
    int colIndex = -1;
    int32_t gpuKind = -1;
    double constValue = 0;
};
inline ColInterpolation NewColInterpolation(const std::string &colName, std::vector<Type> inputTypes, ColInterpolationFunc fn) {
    ColInterpolation c;
    c.colName = colName; c.inputTypes = std::move(inputTypes); c.fn = std::move(fn);
    return c;
}

class Rolling;
using RollingPtr = std::shared_ptr<Rolling>;

// Rolling: rolling.go:14-29 ; intervalRolling: rolling.go:31-43
class Rolling {
  public:
    BowPtr bow;
    int intervalColIndex = 0;
    int64_t interval = 0;
    Options options;
    int numWindows = 0;
    int64_t currWindowFirstValue = 0;
    int currWindowIndex = 0;
    Error err;
    // iterator state served from the device (all windows' bounds at once)
    bool boundsReady = false;
    std::vector<int64_t> firstIndex, sliceBegin, sliceEnd;
    std::vector<uint8_t> isIncl;
    int64_t s0 = 0;

    // A Rolling that Interpolate returned and whose interpolated Bow nobody has asked for yet (the Go shim: rolling/gpu_lazy.go,
    // patches/0006).  Aggregate on it is ONE call to the library (bowgpu_rolling_interpolate_aggregate); every other use first makes
    // the Rolling the reference's Interpolate returns (interpolation.go:57-68) and then IS that Rolling.
    struct Lazy {
        std::shared_ptr<const Rolling> base;   // the Rolling Interpolate was called on
        std::vector<ColInterpolation> interps; // validated: every colIndex is set
        int newIntervalCol = -1;
    };
    std::shared_ptr<Lazy> lazy;

    std::pair<int, Error> NumWindows() const { materialise(); return {numWindows, err}; }
    std::pair<BowPtr, Error> Bow() const { materialise(); return {bow, err}; }

    Error loadBounds() {
        materialise();
        if (boundsReady) return Error();
        firstIndex.assign((size_t)numWindows + 1, 0);
        sliceBegin = firstIndex; sliceEnd = firstIndex;
        isIncl.assign((size_t)numWindows + 1, 0);
        if (numWindows > 0) {
            bowgpu_col ts = bow->ArrowCol(intervalColIndex);
            bowgpu_options o = {options.Offset, options.Inclusive ? 1 : 0, 0};
            int rc = bowgpu_window_bounds(&ts, interval, &o, firstIndex.data(), sliceBegin.data(), sliceEnd.data(), isIncl.data(), BOWGPU_HOST);
            if (rc) return detail::AbiError(rc);
        }
        boundsReady = true;
        return Error();
    }

    // HasNext / Next: rolling.go:162-239, served from the device-computed bounds
    bool HasNext() {
        if (loadBounds()) return false;
        return currWindowIndex < numWindows;
    }
    struct NextResult { int windowIndex; std::optional<Window> window; Error err; };
    NextResult Next() {
        if (!HasNext()) return {currWindowIndex, std::nullopt, Error()};
        const int k = currWindowIndex++;
        Window w;
        w.FirstIndex = (int)firstIndex[k];
        w.IntervalColIndex = intervalColIndex;
        w.FirstValue = s0 + (int64_t)k * interval;
        w.LastValue = w.FirstValue + interval;
        w.IsInclusive = isIncl[k] != 0;
        w.Bow = sliceEnd[k] > sliceBegin[k] ? bow->NewSlice((int)sliceBegin[k], (int)sliceEnd[k]) : bow->NewEmptySlice();
        return {k, w, Error()};
    }

    RollingPtr Aggregate(const std::vector<ColAggregation> &aggrs) const;
    RollingPtr Interpolate(std::vector<ColInterpolation> interps) const;

  private:
    RollingPtr interpolated(const std::vector<ColInterpolation> &interps, int newIntervalCol) const;   // the tail of Interpolate (interpolation.go:57-68)
    RollingPtr interpolateAggregateGPU(const std::vector<ColAggregation> &aggrs) const;                // nullptr: make the two steps
    std::vector<bowgpu_interp> describeInterps(const std::vector<ColInterpolation> &interps, bool *all_builtin) const;
    void materialise() const {
        if (!lazy) return;
        Rolling *self = const_cast<Rolling *>(this);
        std::shared_ptr<Lazy> l = self->lazy;
        self->lazy = nullptr;
        RollingPtr real = l->base->interpolated(l->interps, l->newIntervalCol);
        *self = *real;
    }
    RollingPtr withError(const Error &e) const {
        auto r = std::make_shared<Rolling>(*this);
        r->err = e;
        return r;
    }
    friend std::pair<RollingPtr, Error> newIntervalRolling(BowPtr, int, int64_t, Options);
};

// newIntervalRolling: rolling.go:69-112
inline std::pair<RollingPtr, Error> newIntervalRolling(BowPtr b, int intervalColIndex, int64_t interval, Options options) {
    if (b->ColumnType(intervalColIndex) != Type::Int64)
        return {nullptr, Errorf("impossible to create a new intervalRolling on column of type " + TypeString(b->ColumnType(intervalColIndex)))};
    int64_t off = 0;
    if (bowgpu_enforce_interval_and_offset(interval, options.Offset, &off))
        return {nullptr, Wrap("enforceIntervalAndOffset", Error(bowgpu_last_error()))};
    options.Offset = off;
    if (options.PrevRow && options.PrevRow->NumRows() == 0) options.PrevRow = nullptr;  // enforcePrevRow: rolling.go:130-141
    if (options.PrevRow && options.PrevRow->NumRows() != 1)
        return {nullptr, Wrap("enforcePrevRow", Errorf("prevRow must have only one row, have " + std::to_string(options.PrevRow->NumRows())))};
    bowgpu_col ts = b->ArrowCol(intervalColIndex);
    int64_t s0 = 0, W = 0;
    if (bowgpu_plan_windows(&ts, interval, options.Offset, &s0, &W)) return {nullptr, Error(bowgpu_last_error())};
    auto r = std::make_shared<Rolling>();
    r->bow = b; r->intervalColIndex = intervalColIndex; r->interval = interval; r->options = options;
    r->numWindows = (int)W; r->currWindowFirstValue = s0; r->s0 = s0;
    return {r, Error()};
}

// IntervalRolling: rolling.go:60-67
inline std::pair<RollingPtr, Error> IntervalRolling(BowPtr b, const std::string &colName, int64_t interval, Options options) {
    auto [colIndex, err] = b->ColumnIndex(colName);
    if (err) return {nullptr, err};
    return newIntervalRolling(b, colIndex, interval, options);
}

// SetGPUDevices: the devices ONE Aggregate / Interpolate(...).Aggregate(...) call is spread over (process-wide; none or one id: one device) -
// the C++ twin of shim/go/rolling/gpu_cgo.go's SetGPUDevices (bowgpu_set_devices; a Go application gets the node's devices from the shim's
// init(), a C++ one calls this once).  Nothing else changes: r->Aggregate(...) is the same call with the same result.
inline Error SetGPUDevices(const std::vector<int> &ids) {
    if (bowgpu_set_devices(ids.empty() ? nullptr : ids.data(), (int)ids.size())) return Error(bowgpu_last_error());
    return Error();
}

// ----------------------------------------------------------------------------- Aggregate
inline RollingPtr Rolling::Aggregate(const std::vector<ColAggregation> &aggrs) const {
    if (lazy) {   // r.Interpolate(...).Aggregate(...): both steps in one call to the library when it takes them
        if (RollingPtr r = interpolateAggregateGPU(aggrs)) return r;
        materialise();
    }
    if (err) return std::make_shared<Rolling>(*this);  // aggregation.go:124-126
    // indexedAggregations + validateAggregation: aggregation.go:147-188
    auto fail = [&](const std::string &prefix, const Error &e) { return withError(Wrap(prefix, e)); };
    if (aggrs.empty()) return fail("intervalRolling.indexedAggregations", Errorf("at least one column aggregation is required"));
    Options opts = options;
    int newIntervalCol = -1;
    for (size_t i = 0; i < aggrs.size(); i++) {
        if (aggrs[i].InputName().empty())
            return fail("intervalRolling.indexedAggregations", Errorf("aggregation " + std::to_string(i) + " has no column name"));
        auto [readIndex, e] = bow->ColumnIndex(aggrs[i].InputName());
        if (e) return fail("intervalRolling.indexedAggregations", e);
        aggrs[i].SetInputIndex(readIndex);
        if (aggrs[i].NeedInclusiveWindow()) opts.Inclusive = true;
        if (readIndex == intervalColIndex) newIntervalCol = (int)i;
    }
    if (newIntervalCol == -1)
        return fail("intervalRolling.indexedAggregations", Errorf("must keep interval column '" + bow->ColumnName(intervalColIndex) + "'"));

    // aggregateWindows: aggregation.go:190-238.  Built-in reducers (with Factor chains) run fused on the device;
    // user closures are called per window with slices whose bounds come from the device.
    const size_t A = aggrs.size();
    std::vector<Series> series(A);
    std::vector<int> gpu_idx;
    for (size_t i = 0; i < A; i++) {
        bool fusable = aggrs[i].GPUKind() >= 0;
        for (const auto &t : aggrs[i].Transformations()) fusable = fusable && t.is_factor;
        fusable = fusable && aggrs[i].Transformations().size() <= BOWGPU_MAX_FACTORS;
        if (aggrs[i].GPUKind() >= 0) gpu_idx.push_back((int)i);
        (void)fusable;
    }
    auto outName = [&](size_t i) { return aggrs[i].OutputName().empty() ? bow->ColumnName(aggrs[i].InputIndex()) : aggrs[i].OutputName(); };

    if (!gpu_idx.empty()) {
        std::vector<bowgpu_col> cols;
        for (int i = 0; i < bow->NumCols(); i++) cols.push_back(bow->ArrowCol(i));
        std::vector<bowgpu_agg> ga;
        std::vector<bool> host_transform;
        for (int i : gpu_idx) {
            bowgpu_agg g;
            memset(&g, 0, sizeof g);
            g.kind = aggrs[i].GPUKind();
            g.col = aggrs[i].InputIndex();
            bool all_factor = aggrs[i].Transformations().size() <= BOWGPU_MAX_FACTORS;
            for (const auto &t : aggrs[i].Transformations()) all_factor = all_factor && t.is_factor;
            if (all_factor)
                for (const auto &t : aggrs[i].Transformations()) g.factors[g.n_factors++] = t.factor;
            host_transform.push_back(!all_factor);
            ga.push_back(g);
        }
        bool hidden_key = true;  // the ABI wants one aggregator on the interval column (aggregation.go:163-166)
        for (const auto &g : ga) hidden_key = hidden_key && g.col != intervalColIndex;
        if (hidden_key) {
            bowgpu_agg g;
            memset(&g, 0, sizeof g);
            g.kind = BOWGPU_AGG_WINDOW_START; g.col = intervalColIndex;
            ga.push_back(g);
        }
        std::vector<detail::OutStore> stores(ga.size());
        std::vector<bowgpu_out> outs;
        for (auto &s : stores) outs.push_back(s.Make(numWindows));
        bowgpu_options o = {opts.Offset, opts.Inclusive ? 1 : 0, 0};
        bowgpu_agg_info info;
        int rc = bowgpu_rolling_aggregate(cols.data(), bow->NumCols(), intervalColIndex, interval, &o, ga.data(), (int32_t)ga.size(), outs.data(), &info);
        if (rc) return fail("intervalRolling.aggregateWindows", detail::AbiError(rc));
        for (size_t k = 0; k < gpu_idx.size(); k++) {
            const int i = gpu_idx[k];
            Series s = stores[k].ToSeries(outName((size_t)i), outs[k]);
            if (host_transform[k]) {  // arbitrary transformation.Func: applied to each window's result (aggregation.go:216-227)
                const Type typ = s.typ;
                for (int64_t w = 0; w < s.length; w++) {
                    Value v;
                    if (s.IsValid(w)) {
                        if (typ == Type::Int64) { int64_t x; memcpy(&x, &s.data[w], 8); v = Scalar(x); }
                        else { double x; memcpy(&x, &s.data[w], 8); v = Scalar(x); }
                    }
                    for (const auto &t : aggrs[i].Transformations()) {
                        auto [nv, e] = t(v);
                        if (e) return fail("intervalRolling.aggregateWindows", e);
                        v = nv;
                    }
                    uint64_t bits = 0;
                    bool valid = v.has_value();
                    if (valid) {
                        if (typ == Type::Int64) { int64_t x = std::holds_alternative<int64_t>(*v) ? std::get<int64_t>(*v) : GoInt64(std::get<double>(*v)); memcpy(&bits, &x, 8); }
                        else { double x = std::holds_alternative<double>(*v) ? std::get<double>(*v) : (double)std::get<int64_t>(*v); memcpy(&bits, &x, 8); }
                    }
                    s.data[w] = bits;
                    if (valid) s.validity[w >> 3] |= (uint8_t)(1u << (w & 7)); else s.validity[w >> 3] &= (uint8_t)~(1u << (w & 7));
                }
            }
            series[(size_t)i] = std::move(s);
        }
    }

    // user closures
    bool any_closure = false;
    for (size_t i = 0; i < A; i++) any_closure = any_closure || aggrs[i].GPUKind() < 0;
    if (any_closure) {
        Rolling it = *this;
        it.options = opts;
        it.currWindowIndex = 0;
        it.boundsReady = false;
        std::vector<std::vector<Value>> results(A, std::vector<Value>((size_t)numWindows));
        while (it.HasNext()) {
            auto nx = it.Next();
            if (nx.err) return fail("intervalRolling.aggregateWindows", nx.err);
            for (size_t i = 0; i < A; i++) {
                if (aggrs[i].GPUKind() >= 0) continue;
                const Window &w = *nx.window;
                auto [val, e] = (!aggrs[i].NeedInclusiveWindow() && w.IsInclusive) ? aggrs[i].Func()(aggrs[i].InputIndex(), w.UnsetInclusive())
                                                                                   : aggrs[i].Func()(aggrs[i].InputIndex(), w);
                if (e) return fail("intervalRolling.aggregateWindows", e);
                Value v = val;
                for (const auto &t : aggrs[i].Transformations()) {
                    auto [nv, e2] = t(v);
                    if (e2) return fail("intervalRolling.aggregateWindows", e2);
                    v = nv;
                }
                results[i][(size_t)nx.windowIndex] = v;
            }
        }
        if (it.err) return fail("intervalRolling.aggregateWindows", it.err);
        for (size_t i = 0; i < A; i++) {
            if (aggrs[i].GPUKind() >= 0) continue;
            const Type typ = aggrs[i].GetReturnType(bow->ColumnType(aggrs[i].InputIndex()), bow->ColumnType(intervalColIndex));
            auto [b1, e] = NewBowFromColBasedInterfaces({outName(i)}, {typ}, {results[i]});  // SetOrDrop conversions
            if (e) return fail("intervalRolling.aggregateWindows", e);
            series[i] = b1->cols[0];
        }
    }

    auto [b, e] = NewBow(std::move(series));
    if (e) return fail("intervalRolling.aggregateWindows", e);
    auto [newR, e2] = newIntervalRolling(b, newIntervalCol, interval, opts);  // aggregation.go:139
    if (e2) return fail("newIntervalRolling", e2);
    return newR;
}

// ----------------------------------------------------------------------------- Interpolate
inline RollingPtr Rolling::Interpolate(std::vector<ColInterpolation> interps) const {
    materialise();
    if (err) return std::make_shared<Rolling>(*this);  // interpolation.go:31-33
    if (interps.empty()) return withError(Errorf("at least one column interpolation is required"));
    int newIntervalCol = -1;
    for (size_t i = 0; i < interps.size(); i++) {  // validateInterpolation: interpolation.go:71-96
        if (interps[i].colName.empty())
            return withError(Wrap("intervalRolling.validateInterpolation", Errorf("interpolation " + std::to_string(i) + " has no column name")));
        auto [idx, e] = bow->ColumnIndex(interps[i].colName);
        if (e) return withError(Wrap("intervalRolling.validateInterpolation", e));
        interps[i].colIndex = idx;
        bool ok = false;
        for (Type t : interps[i].inputTypes) ok = ok || t == bow->ColumnType(idx);
        if (!ok) {
            std::string acc = "[";
            for (size_t k = 0; k < interps[i].inputTypes.size(); k++) acc += (k ? " " : "") + TypeString(interps[i].inputTypes[k]);
            acc += "]";
            return withError(Wrap("intervalRolling.validateInterpolation", Errorf("accepts types " + acc + ", got type " + TypeString(bow->ColumnType(idx)))));
        }
        if (idx == intervalColIndex) newIntervalCol = (int)i;
    }
    if (newIntervalCol == -1) return withError(Errorf("must keep interval column '" + bow->ColumnName(intervalColIndex) + "'"));

    // every interpolator a built-in, a Rolling nobody has stepped, a Bow with windows: the interpolated Bow is made when first asked for
    bool all_builtin = true;
    for (const auto &ip : interps) all_builtin = all_builtin && ip.gpuKind >= 0;
    if (all_builtin && currWindowIndex == 0 && numWindows > 0) {
        materialise();
        auto r = std::make_shared<Rolling>(*this);
        r->lazy = std::make_shared<Lazy>();
        r->lazy->base = std::make_shared<const Rolling>(*this);
        r->lazy->interps = interps;
        r->lazy->newIntervalCol = newIntervalCol;
        return r;
    }
    return interpolated(interps, newIntervalCol);
}

inline std::vector<bowgpu_interp> Rolling::describeInterps(const std::vector<ColInterpolation> &interps, bool *all_builtin) const {
    std::vector<bowgpu_interp> gi;
    *all_builtin = true;
    for (const auto &ip : interps) {
        if (ip.gpuKind < 0) { *all_builtin = false; return gi; }
        bowgpu_interp g;
        memset(&g, 0, sizeof g);
        g.kind = ip.gpuKind; g.col = ip.colIndex; g.const_value = ip.constValue;
        if (options.PrevRow) {  // linear.go:14-18, stepprevious.go:13-15
            const int last = options.PrevRow->NumRows() - 1;
            auto [pt, tv] = options.PrevRow->GetFloat64(intervalColIndex, last);
            auto [pv, vv] = options.PrevRow->GetFloat64(ip.colIndex, last);
            g.has_prev_row = 1; g.prev_t = pt; g.prev_t_valid = tv; g.prev_v = pv; g.prev_v_valid = vv;
            g.prev_v_i64 = options.PrevRow->GetInt64(ip.colIndex, last).first;
        }
        gi.push_back(g);
    }
    return gi;
}

// interpolateWindows + newIntervalRolling on the result: interpolation.go:57-68, :98-161
inline RollingPtr Rolling::interpolated(const std::vector<ColInterpolation> &interps, int newIntervalCol) const {
    bool all_builtin = true;
    std::vector<bowgpu_interp> gi = describeInterps(interps, &all_builtin);
    if (!all_builtin)
        return withError(Wrap("intervalRolling.interpolateWindows", Errorf("custom ColInterpolation closures are outside the device path")));
    std::vector<bowgpu_col> cols;
    for (int i = 0; i < bow->NumCols(); i++) cols.push_back(bow->ArrowCol(i));
    bowgpu_options o = {options.Offset, options.Inclusive ? 1 : 0, 0};
    int64_t n_out = 0;
    int rc = bowgpu_rolling_interpolate_count(cols.data(), bow->NumCols(), intervalColIndex, interval, &o, gi.data(), (int32_t)gi.size(), &n_out);
    if (rc) return withError(Wrap("intervalRolling.interpolateWindows", detail::AbiError(rc)));
    std::vector<detail::OutStore> stores(gi.size());
    std::vector<bowgpu_out> outs;
    for (auto &s : stores) outs.push_back(s.Make(n_out));
    rc = bowgpu_rolling_interpolate_fill(cols.data(), bow->NumCols(), intervalColIndex, interval, &o, gi.data(), (int32_t)gi.size(), outs.data());
    if (rc) return withError(Wrap("intervalRolling.interpolateWindows", detail::AbiError(rc)));
    std::vector<Series> series;
    for (size_t i = 0; i < gi.size(); i++) {
        if (outs[i].length == 0) outs[i].type = (int32_t)bow->ColumnType(interps[i].colIndex);
        series.push_back(stores[i].ToSeries(bow->ColumnName(interps[i].colIndex), outs[i]));
    }
    auto [b, e] = NewBow(std::move(series));
    if (e) return withError(Wrap("intervalRolling.interpolateWindows", e));
    auto [newR, e2] = newIntervalRolling(b, newIntervalCol, interval, options);  // interpolation.go:63
    if (e2) return withError(Wrap("newIntervalRolling", e2));
    return newR;
}

// Interpolate(...).Aggregate(...) as one call (Go shim: interpolateAggregateGPU in rolling/gpu_cgo.go).  Mirrors Aggregate
// (aggregation.go:123-145) on the interpolated Rolling, whose Bow has the input's columns and types: indexedAggregations, the call,
// newIntervalRolling on the result.  nullptr = anything that is not a plain success: the caller makes the reference's two steps, which
// word every error.
inline RollingPtr Rolling::interpolateAggregateGPU(const std::vector<ColAggregation> &aggrs) const {
    const Rolling &b0 = *lazy->base;
    const std::vector<ColInterpolation> &interps = lazy->interps;
    if ((int)interps.size() != b0.bow->NumCols() || aggrs.empty()) return nullptr;
    for (size_t i = 0; i < interps.size(); i++)
        if (interps[i].colIndex != (int)i) return nullptr;
    Options opts = b0.options;
    int newIntervalCol = -1;
    std::vector<bowgpu_agg> ga;
    for (size_t i = 0; i < aggrs.size(); i++) {
        if (aggrs[i].InputName().empty()) return nullptr;
        auto [readIndex, e] = b0.bow->ColumnIndex(aggrs[i].InputName());
        if (e) return nullptr;
        aggrs[i].SetInputIndex(readIndex);
        if (aggrs[i].NeedInclusiveWindow()) opts.Inclusive = true;
        if (readIndex == b0.intervalColIndex) newIntervalCol = (int)i;
        if (aggrs[i].GPUKind() < 0 || aggrs[i].Transformations().size() > BOWGPU_MAX_FACTORS) return nullptr;
        bowgpu_agg g;
        memset(&g, 0, sizeof g);
        g.kind = aggrs[i].GPUKind(); g.col = readIndex;
        for (const auto &t : aggrs[i].Transformations()) {
            if (!t.is_factor) return nullptr;
            g.factors[g.n_factors++] = t.factor;
        }
        ga.push_back(g);
    }
    if (newIntervalCol == -1) return nullptr;
    bool all_builtin = true;
    std::vector<bowgpu_interp> gi = b0.describeInterps(interps, &all_builtin);
    if (!all_builtin) return nullptr;
    std::vector<bowgpu_col> cols;
    for (int i = 0; i < b0.bow->NumCols(); i++) cols.push_back(b0.bow->ArrowCol(i));
    std::vector<detail::OutStore> stores(ga.size());
    std::vector<bowgpu_out> outs;
    for (auto &s : stores) outs.push_back(s.Make(b0.numWindows));   // the interpolated frame keeps the window grid; the library checks the capacity
    bowgpu_options o = {opts.Offset, opts.Inclusive ? 1 : 0, 0};
    bowgpu_agg_info info;
    if (bowgpu_rolling_interpolate_aggregate(cols.data(), b0.bow->NumCols(), b0.intervalColIndex, b0.interval, &o, gi.data(), (int32_t)gi.size(),
                                             ga.data(), (int32_t)ga.size(), outs.data(), &info) != 0)
        return nullptr;
    std::vector<Series> series;
    for (size_t i = 0; i < ga.size(); i++) {
        const std::string name = aggrs[i].OutputName().empty() ? b0.bow->ColumnName(aggrs[i].InputIndex()) : aggrs[i].OutputName();   // aggregation.go:230-234
        series.push_back(stores[i].ToSeries(name, outs[i]));
    }
    auto [b, e] = NewBow(std::move(series));
    if (e) return nullptr;
    auto [newR, e2] = newIntervalRolling(b, newIntervalCol, b0.interval, opts);  // aggregation.go:139
    if (e2) return b0.withError(Wrap("newIntervalRolling", e2));
    return newR;
}

// ----------------------------------------------------------------------------- rolling/aggregation
namespace aggregation {
namespace detail2 {
inline ColAggregation builtin(const std::string &col, bool incl, Type typ, int32_t kind) {
    // the closure is never called for built-ins (the device computes them); it reports misuse loudly
    ColAggregation a = NewColAggregation(col, incl, typ, [](int, const Window &) -> std::pair<Value, Error> {
        return {Nil(), Errorf("built-in aggregation closures run on the device")};
    });
    a.p->gpuKind = kind;
    return a;
}
}  // namespace detail2
inline ColAggregation WindowStart(const std::string &col) { return detail2::builtin(col, false, IteratorDependent, BOWGPU_AGG_WINDOW_START); }
inline ColAggregation Sum(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_SUM); }
inline ColAggregation ArithmeticMean(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_MEAN); }
inline ColAggregation Min(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_MIN); }
inline ColAggregation Max(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_MAX); }
inline ColAggregation Count(const std::string &col) { return detail2::builtin(col, false, Int64, BOWGPU_AGG_COUNT); }
inline ColAggregation First(const std::string &col) { return detail2::builtin(col, false, InputDependent, BOWGPU_AGG_FIRST); }
inline ColAggregation Last(const std::string &col) { return detail2::builtin(col, false, InputDependent, BOWGPU_AGG_LAST); }
inline ColAggregation Mode(const std::string &col) { return detail2::builtin(col, false, InputDependent, BOWGPU_AGG_MODE); }  // mode.go:8-32
inline ColAggregation IntegralStep(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_INTEGRAL_STEP); }
inline ColAggregation IntegralTrapezoid(const std::string &col) { return detail2::builtin(col, true, Float64, BOWGPU_AGG_INTEGRAL_TRAPEZOID); }
inline ColAggregation WeightedAverageStep(const std::string &col) { return detail2::builtin(col, false, Float64, BOWGPU_AGG_WAVG_STEP); }
inline ColAggregation WeightedAverageLinear(const std::string &col) { return detail2::builtin(col, true, Float64, BOWGPU_AGG_WAVG_LINEAR); }

// Aggregate (whole frame): rolling/aggregation/whole.go:12-93
inline std::pair<BowPtr, Error> Aggregate(BowPtr b, const std::string &intervalColName, const std::vector<ColAggregation> &aggrs) {
    if (!b) return {nullptr, Errorf("nil bow")};
    if (aggrs.empty()) return {nullptr, Errorf("at least one column aggregation is required")};
    auto [intervalColIndex, err] = b->ColumnIndex(intervalColName);
    if (err) return {nullptr, err};
    std::vector<bowgpu_agg> ga;
    std::vector<std::string> names;
    for (size_t i = 0; i < aggrs.size(); i++) {
        if (aggrs[i].InputName().empty()) return {nullptr, Errorf("column aggregation " + std::to_string(i) + ": no input name")};
        auto [idx, e] = b->ColumnIndex(aggrs[i].InputName());
        if (e) return {nullptr, Wrap("column aggregation " + std::to_string(i), e)};
        aggrs[i].SetInputIndex(idx);
        if (aggrs[i].GPUKind() < 0) return {nullptr, Errorf("column aggregation " + std::to_string(i) + ": custom closures are outside the device path")};
        bowgpu_agg g;
        memset(&g, 0, sizeof g);
        g.kind = aggrs[i].GPUKind(); g.col = idx;
        for (const auto &t : aggrs[i].Transformations())
            if (t.is_factor && g.n_factors < BOWGPU_MAX_FACTORS) g.factors[g.n_factors++] = t.factor;
        ga.push_back(g);
        names.push_back(aggrs[i].OutputName().empty() ? b->ColumnName(idx) : aggrs[i].OutputName());
    }
    std::vector<bowgpu_col> cols;
    for (int i = 0; i < b->NumCols(); i++) cols.push_back(b->ArrowCol(i));
    std::vector<bow::detail::OutStore> stores(ga.size());
    std::vector<bowgpu_out> outs;
    for (auto &s : stores) outs.push_back(s.Make(1));
    int rc = bowgpu_aggregate_whole(cols.data(), b->NumCols(), intervalColIndex, ga.data(), (int32_t)ga.size(), outs.data());
    if (rc) return {nullptr, bow::detail::AbiError(rc)};
    std::vector<Series> series;
    for (size_t i = 0; i < ga.size(); i++) series.push_back(stores[i].ToSeries(names[i], outs[i]));
    return NewBow(std::move(series));
}
}  // namespace aggregation

// ----------------------------------------------------------------------------- rolling/interpolation
namespace interpolation {
namespace detail3 {
inline ColInterpolation builtin(const std::string &col, std::vector<Type> types, int32_t kind) {
    ColInterpolation c = NewColInterpolation(col, std::move(types), nullptr);
    c.gpuKind = kind;
    return c;
}
}  // namespace detail3
inline ColInterpolation WindowStart(const std::string &col) { return detail3::builtin(col, {Int64}, BOWGPU_INTERP_WINDOW_START); }               // windowstart.go:8-14
inline ColInterpolation Linear(const std::string &col) { return detail3::builtin(col, {Int64, Float64}, BOWGPU_INTERP_LINEAR); }                  // linear.go:8-38
inline ColInterpolation StepPrevious(const std::string &col) { return detail3::builtin(col, {Int64, Float64, Boolean, String}, BOWGPU_INTERP_STEP_PREVIOUS); }
inline ColInterpolation None(const std::string &col) { return detail3::builtin(col, {Int64, Float64, Boolean}, BOWGPU_INTERP_NONE); }
// the constant-valued closure of rolling/interpolation_test.go:16-19, as a tagged interpolator
inline ColInterpolation Const(const std::string &col, std::vector<Type> types, double value) {
    ColInterpolation c = detail3::builtin(col, std::move(types), BOWGPU_INTERP_CONST);
    c.constValue = value;
    return c;
}
}  // namespace interpolation

}  // namespace rolling
}  // namespace bow
