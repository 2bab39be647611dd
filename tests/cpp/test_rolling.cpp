// test_rolling.cpp — the reference's own table-driven tests for the rolling path, replayed through the C++
// mirror of its interface (bow_amd/host/bow_rolling.hpp) and therefore through the C ABI and the HIP kernels.
// Each test names the reference test it follows (file:line).  Needs a GPU (run by tests/test_gpu_host_mirror.py).
#include <cstdio>
#include <string>
#include <vector>

#include "../../bow_amd/host/bow_rolling.hpp"

using namespace bow;
namespace rl = bow::rolling;
namespace ag = bow::rolling::aggregation;
namespace ip = bow::rolling::interpolation;
namespace tr = bow::rolling::transformation;

static int g_fail = 0, g_checks = 0;
static std::string g_test;
#define CHECK(cond)                                                                     \
    do {                                                                                \
        g_checks++;                                                                     \
        if (!(cond)) { g_fail++; printf("FAIL %s:%d [%s] %s\n", __FILE__, __LINE__, g_test.c_str(), #cond); } \
    } while (0)
#define CHECK_EQ_STR(a, b)                                                              \
    do {                                                                                \
        g_checks++;                                                                     \
        std::string _a = (a), _b = (b);                                                 \
        if (_a != _b) { g_fail++; printf("FAIL %s:%d [%s]\n   got:  %s\n   want: %s\n", __FILE__, __LINE__, g_test.c_str(), _a.c_str(), _b.c_str()); } \
    } while (0)
#define TEST(name) g_test = name;

static const std::string timeCol = "time", valueCol = "value", badCol = "badcol";

static BowPtr tv(const std::vector<Value> &t, const std::vector<Value> &v, Type vt = Float64) {
    return NewBowFromColBasedInterfaces({timeCol, valueCol}, {Int64, vt}, {t, v}).first;
}
static BowPtr rows(const std::vector<std::vector<Value>> &r, Type vt = Float64) {
    return NewBowFromRowBasedInterfaces({timeCol, valueCol}, {Int64, vt}, r).first;
}
static void expectEqual(const BowPtr &got, const BowPtr &want) {
    g_checks++;
    if (!got || !want || !got->Equal(*want)) {
        g_fail++;
        printf("FAIL [%s]\n expect:\n%s have:\n%s", g_test.c_str(), want ? want->String().c_str() : "<nil>\n", got ? got->String().c_str() : "<nil>\n");
    }
}

// ---- rolling/rolling_test.go:20-68 TestIntervalRolling_NumWindows
static void TestNumWindows() {
    struct C { const char *name; std::vector<Value> t, v; int64_t interval; rl::Options o; int want; };
    std::vector<C> cases = {
        {"empty bow", {}, {}, 1, {}, 0},
        {"one liner bow", {I(0)}, {F(1.)}, 1, {}, 1},
        {"points in same window", {I(0), I(9)}, {F(1.), F(1.)}, 10, {}, 1},
        {"excluded point goes in next window", {I(0), I(10)}, {F(1.), F(1.)}, 10, {}, 2},
        {"offset puts first value in preceding window", {I(0), I(9)}, {F(1.), F(1.)}, 10, {1, false, nullptr}, 2},
    };
    for (auto &c : cases) {
        TEST(std::string("NumWindows/") + c.name);
        auto [r, err] = rl::IntervalRolling(tv(c.t, c.v), timeCol, c.interval, c.o);
        CHECK(!err);
        auto [n, e2] = r->NumWindows();
        CHECK(!e2);
        CHECK(n == c.want);
    }
}

// ---- rolling/rolling_test.go:70-109 TestIntervalRolling_iterator_init
static void TestIteratorInit() {
    TEST("iterator_init/interval == 0");
    { auto [r, err] = rl::IntervalRolling(tv({I(0)}, {F(1.)}), timeCol, 0, {}); CHECK(!r); CHECK_EQ_STR(err.msg, "enforceIntervalAndOffset: strictly positive interval required"); }
    TEST("iterator_init/non existing index");
    { auto [r, err] = rl::IntervalRolling(tv({I(0)}, {F(1.)}), badCol, 1, {}); CHECK_EQ_STR(err.msg, "no column 'badcol'"); }
    TEST("iterator_init/invalid interval type");
    {
        auto b = NewBowFromColBasedInterfaces({timeCol}, {Float64}, {{F(0.)}}).first;
        auto [r, err] = rl::IntervalRolling(b, timeCol, 1, {});
        CHECK_EQ_STR(err.msg, "impossible to create a new intervalRolling on column of type float64");
    }
    TEST("iterator_init/empty bow gives valid finished iterator");
    {
        auto [r, err] = rl::IntervalRolling(tv({}, {}), timeCol, 1, {});
        CHECK(!err);
        auto nx = r->Next();
        CHECK(!nx.window.has_value());
        CHECK(!nx.err);
    }
}

// ---- rolling/rolling_test.go:111-297 TestIntervalRolling_iterate
struct TW { int idx; int64_t start, end; int firstIndex; std::vector<Value> t, v; };
static void TestIterate() {
    auto b = tv({I(12), I(15), I(16), I(25), I(25), I(29)}, {F(1.2), F(1.5), F(1.6), F(2.5), F(3.5), F(2.9)});
    std::vector<TW> w0 = {{0, 10, 15, 0, {I(12)}, {F(1.2)}}, {1, 15, 20, 1, {I(15), I(16)}, {F(1.5), F(1.6)}}, {2, 20, 25, 3, {}, {}},
                          {3, 25, 30, 3, {I(25), I(25), I(29)}, {F(2.5), F(3.5), F(2.9)}}};
    std::vector<TW> w3 = {{0, 8, 13, 0, {I(12)}, {F(1.2)}}, {1, 13, 18, 1, {I(15), I(16)}, {F(1.5), F(1.6)}}, {2, 18, 23, 3, {}, {}},
                          {3, 23, 28, 3, {I(25), I(25)}, {F(2.5), F(3.5)}}, {4, 28, 33, 5, {I(29)}, {F(2.9)}}};
    struct C { const char *name; rl::Options o; std::vector<TW> w; };
    std::vector<C> cases = {
        {"no option", {}, w0},
        {"with inclusive windows", {0, true, nullptr}, {{0, 10, 15, 0, {I(12), I(15)}, {F(1.2), F(1.5)}}, {1, 15, 20, 1, {I(15), I(16)}, {F(1.5), F(1.6)}},
                                                        {2, 20, 25, 3, {I(25)}, {F(2.5)}}, {3, 25, 30, 3, {I(25), I(25), I(29)}, {F(2.5), F(3.5), F(2.9)}}}},
        {"with offset falling before first point", {1, false, nullptr}, {{0, 11, 16, 0, {I(12), I(15)}, {F(1.2), F(1.5)}}, {1, 16, 21, 2, {I(16)}, {F(1.6)}},
                                                                         {2, 21, 26, 3, {I(25), I(25)}, {F(2.5), F(3.5)}}, {3, 26, 31, 5, {I(29)}, {F(2.9)}}}},
        {"with offset falling at first point", {2, false, nullptr}, {{0, 12, 17, 0, {I(12), I(15), I(16)}, {F(1.2), F(1.5), F(1.6)}}, {1, 17, 22, 3, {}, {}},
                                                                     {2, 22, 27, 3, {I(25), I(25)}, {F(2.5), F(3.5)}}, {3, 27, 32, 5, {I(29)}, {F(2.9)}}}},
        {"with offset falling after first point", {3, false, nullptr}, w3},
        {"offset > interval", {8, false, nullptr}, w3},
        {"offset == interval", {5, false, nullptr}, w0},
        {"offset < 0", {-2, false, nullptr}, w3},
    };
    for (auto &c : cases) {
        TEST(std::string("iterate/") + c.name);
        auto [r, err] = rl::IntervalRolling(b, timeCol, 5, c.o);
        CHECK(!err);
        size_t i = 0;
        for (; r->HasNext(); i++) {
            auto nx = r->Next();  // checkTestWindow: rolling_test.go:307-318
            CHECK(i < c.w.size());
            if (i >= c.w.size()) break;
            CHECK(nx.windowIndex == c.w[i].idx);
            CHECK(nx.window.has_value());
            CHECK(nx.window->FirstValue == c.w[i].start);
            CHECK(nx.window->LastValue == c.w[i].end);
            CHECK(nx.window->FirstIndex == c.w[i].firstIndex);
            expectEqual(nx.window->Bow, tv(c.w[i].t, c.w[i].v));
        }
        CHECK(i == c.w.size());
        auto nx = r->Next();
        CHECK(!nx.window.has_value());
    }
}

// ---- rolling/aggregation_test.go:12-123 TestIntervalRolling_Aggregate (custom closures)
static void TestAggregateDriver() {
    auto b = tv({I(10), I(15), I(16), I(25), I(29)}, {F(1.0), F(1.5), F(1.6), F(2.5), F(2.9)});
    auto [r, err] = rl::IntervalRolling(b, timeCol, 10, {});
    CHECK(!err);
    auto timeAggr = rl::NewColAggregation(timeCol, false, Int64, [](int, const rl::Window &w) -> std::pair<Value, Error> { return {I(w.FirstValue), Error()}; });
    auto valueAggr = rl::NewColAggregation(valueCol, false, Float64, [](int, const rl::Window &w) -> std::pair<Value, Error> { return {F((double)w.Bow->NumRows()), Error()}; });
    auto doubleAggr = rl::NewColAggregation(valueCol, false, Float64, [](int, const rl::Window &w) -> std::pair<Value, Error> { return {F((double)w.Bow->NumRows() * 2), Error()}; });

    TEST("Aggregate/keep columns");
    { auto [a, e] = r->Aggregate({timeAggr, valueAggr})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({timeCol, valueCol}, {Int64, Float64}, {{I(10), I(20)}, {F(3.), F(2.)}}).first); }
    TEST("Aggregate/swap columns");
    { auto [a, e] = r->Aggregate({valueAggr, timeAggr})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({valueCol, timeCol}, {Float64, Int64}, {{F(3.), F(2.)}, {I(10), I(20)}}).first); }
    TEST("Aggregate/rename columns");
    { auto [a, e] = r->Aggregate({timeAggr.RenameOutput("a"), valueAggr.RenameOutput("b")})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({"a", "b"}, {Int64, Float64}, {{I(10), I(20)}, {F(3.), F(2.)}}).first); }
    TEST("Aggregate/less than in original");
    { auto [a, e] = r->Aggregate({timeAggr})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({timeCol}, {Int64}, {{I(10), I(20)}}).first); }
    TEST("Aggregate/more than in original");
    { auto [a, e] = r->Aggregate({timeAggr, doubleAggr.RenameOutput("double"), valueAggr})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({timeCol, "double", valueCol}, {Int64, Float64, Float64}, {{I(10), I(20)}, {F(6.), F(4.)}, {F(3.), F(2.)}}).first); }
    TEST("Aggregate/missing interval colIndex");
    { auto [a, e] = r->Aggregate({valueAggr})->Bow(); CHECK_EQ_STR(e.msg, "intervalRolling.indexedAggregations: must keep interval column 'time'"); }
    TEST("Aggregate/invalid colIndex");
    { auto bad = rl::NewColAggregation("-", false, Int64, [](int, const rl::Window &) -> std::pair<Value, Error> { return {Nil(), Error()}; });
      auto [a, e] = r->Aggregate({timeAggr, bad})->Bow(); CHECK_EQ_STR(e.msg, "intervalRolling.indexedAggregations: no column '-'"); }
    TEST("Aggregate/built-in and closure mixed, device + host");
    { auto [a, e] = r->Aggregate({ag::WindowStart(timeCol), valueAggr, ag::Sum(valueCol).RenameOutput("sum")})->Bow(); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({timeCol, valueCol, "sum"}, {Int64, Float64, Float64}, {{I(10), I(20)}, {F(3.), F(2.)}, {F(1.0 + 1.5 + 1.6), F(2.5 + 2.9)}}).first); }
}

// ---- rolling/aggregation_test.go:125-171 TestWindow_UnsetInclusive
static void TestUnsetInclusive() {
    TEST("Window.UnsetInclusive");
    auto inclusiveBow = NewBowFromColBasedInterfaces({timeCol, valueCol}, {Int64, Int64}, {{I(1), I(2)}, {I(1), I(2)}}).first;
    auto exclusiveBow = NewBowFromColBasedInterfaces({timeCol, valueCol}, {Int64, Int64}, {{I(1)}, {I(1)}}).first;
    rl::Window w{inclusiveBow, 0, 0, 0, 2, true};
    rl::Window x = w.UnsetInclusive();
    expectEqual(x.Bow, exclusiveBow);
    CHECK(!x.IsInclusive && x.FirstValue == 0 && x.LastValue == 2 && x.FirstIndex == 0);
    CHECK(w.IsInclusive && w.Bow->NumRows() == 2);
}

// ---- rolling/aggregation/core_test.go:24-108 fixtures + runTestCases; one case table per reducer test file
static BowPtr emptyBow() { return tv({}, {}); }
static BowPtr nilBow() { return rows({{I(10), N}, {I(11), N}, {I(20), N}}); }
static BowPtr sparseFloatBow() {
    return rows({{I(10), F(10.)}, {I(11), N}, {I(20), N}, {I(40), N}, {I(41), F(10.)}, {I(50), F(10.)}, {I(51), F(20.)}, {I(61), F(10.)}, {I(69), F(20.)}});
}
static void runTestCase(const std::string &name, rl::ColAggregationConstruct construct, const std::vector<tr::Func> &transforms,
                        BowPtr tested, BowPtr expected) {
    TEST(name);
    auto [r, err] = rl::IntervalRolling(tested, timeCol, 10, {});
    CHECK(!err);
    auto [aggregated, e] = r->Aggregate({ag::WindowStart(timeCol), construct(valueCol).SetTransformations(transforms)})->Bow();
    CHECK(!e);
    expectEqual(aggregated, expected);
}
static BowPtr win6(const std::vector<Value> &v, Type t = Float64) { return tv({I(10), I(20), I(30), I(40), I(50), I(60)}, v, t); }
static void TestReducers() {
    const double factor = 0.1;
    runTestCase("Sum/empty", ag::Sum, {}, emptyBow(), emptyBow());                                                      // sum_test.go:13-23
    runTestCase("Sum/sparse float", ag::Sum, {}, sparseFloatBow(), win6({F(10.), F(0.), F(0.), F(10.), F(30.), F(30.)}));  // :25-42
    runTestCase("ArithmeticMean/empty", ag::ArithmeticMean, {}, emptyBow(), emptyBow());
    runTestCase("ArithmeticMean/sparse", ag::ArithmeticMean, {}, sparseFloatBow(), win6({F(10.), N, N, F(10.), F(15.), F(15.)}));  // arithmeticmean_test.go:24-42
    runTestCase("Min/sparse float", ag::Min, {}, sparseFloatBow(), win6({F(10.), N, N, F(10.), F(10.), F(10.)}));       // minmax_test.go:25-42
    runTestCase("Max/sparse float", ag::Max, {}, sparseFloatBow(), win6({F(10.), N, N, F(10.), F(20.), F(20.)}));       // :99-116
    runTestCase("Count/empty", ag::Count, {}, emptyBow(), tv({}, {}, Int64));                                           // count_test.go:13-23
    runTestCase("Count/sparse", ag::Count, {}, sparseFloatBow(), win6({I(1), I(0), I(0), I(1), I(2), I(2)}, Int64));    // :24-42
    runTestCase("First/sparse", ag::First, {}, sparseFloatBow(), win6({F(10.), N, N, F(10.), F(10.), F(10.)}));         // firstlast_test.go:25-42
    runTestCase("Last/sparse float", ag::Last, {}, sparseFloatBow(), win6({F(10.), N, N, F(10.), F(20.), F(20.)}));     // :99-116
    runTestCase("Mode/empty", ag::Mode, {}, emptyBow(), emptyBow());                                                    // mode_test.go:33-44
    runTestCase("Mode/mode float", ag::Mode, {},                                                                        // mode_test.go:11-31, :45-62
                tv({I(10), I(11), I(20), I(21), I(22), I(30), I(31), I(32), I(50), I(51)}, {F(10.), F(10.), F(42.), F(42.), F(10.), N, N, F(10.), N, N}),
                tv({I(10), I(20), I(30), I(40), I(50)}, {F(10.), F(42.), F(10.), N, N}));
    runTestCase("IntegralStep/sparse float", ag::IntegralStep, {}, sparseFloatBow(), win6({F(100.), N, N, F(90.), F(190.), F(100.)}));  // integral_test.go:26-43
    runTestCase("IntegralStep_scaled/sparse (custom transform func)", ag::IntegralStep,                                  // integral_test.go:85-128
                {tr::Func{[factor](Value x) -> std::pair<Value, Error> { if (!x) return {Nil(), Error()}; return {F(std::get<double>(*x) * factor), Error()}; }}},
                sparseFloatBow(), win6({F(factor * 100.), N, N, F(factor * 90.), F(factor * 190.), F(factor * 100.)}));
    runTestCase("IntegralStep_scaled/Factor fused on the device", ag::IntegralStep, {tr::Factor(0.1)}, sparseFloatBow(),
                win6({F(factor * 100.), N, N, F(factor * 90.), F(factor * 190.), F(factor * 100.)}));
    runTestCase("IntegralTrapezoid/sparse float", ag::IntegralTrapezoid, {}, sparseFloatBow(), win6({N, N, N, F(90.), F(15.), F(120.)}));  // :145-162
    runTestCase("WeightedAverageStep/sparse float", ag::WeightedAverageStep, {}, sparseFloatBow(), win6({F(10.), N, N, F(9.), F(19.), F(10.)}));  // weightedmean_test.go:24-42
    runTestCase("WeightedAverageStep/float only nil", ag::WeightedAverageStep, {}, nilBow(), tv({I(10), I(20)}, {N, N}));   // :43-57
    runTestCase("WeightedAverageLinear/sparse float", ag::WeightedAverageLinear, {}, sparseFloatBow(), win6({N, N, N, F(9.), F(1.5), F(12.)}));  // :113-131
}

// ---- rolling/transformation/factor_test.go:10-34
static void TestFactor() {
    TEST("Factor");
    auto f = tr::Factor(0.1);
    { auto [v, e] = f(Nil()); CHECK(!e && !v); }
    { auto [v, e] = f(I(11)); CHECK(!e && v && std::get<int64_t>(*v) == 1); }
    { auto [v, e] = f(F(11.)); CHECK(!e && v && std::get<double>(*v) == 1.1); }
}

// ---- rolling/aggregation/whole_test.go:12-104
static void TestWhole() {
    TEST("whole/empty bow");
    { auto [a, e] = ag::Aggregate(tv({}, {}), timeCol, {ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol)}); CHECK(!e); expectEqual(a, tv({}, {})); }
    auto b = tv({I(10), I(20), I(30)}, {F(1.), F(2.), F(3.)});
    TEST("whole/keep columns");
    { auto [a, e] = ag::Aggregate(b, timeCol, {ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol)}); CHECK(!e); expectEqual(a, tv({I(10)}, {F(2.)})); }
    TEST("whole/rename columns");
    { auto [a, e] = ag::Aggregate(b, timeCol, {ag::WindowStart(timeCol).RenameOutput("a"), ag::ArithmeticMean(valueCol).RenameOutput("b")}); CHECK(!e);
      expectEqual(a, NewBowFromColBasedInterfaces({"a", "b"}, {Int64, Float64}, {{I(10)}, {F(2.)}}).first); }
    TEST("whole/invalid column");
    { auto [a, e] = ag::Aggregate(b, timeCol, {ag::WindowStart("-")}); CHECK_EQ_STR(e.msg, "column aggregation 0: no column '-'"); }
}

// ---- rolling/interpolation_test.go:12-101 (driver) + rolling/interpolation/{linear,stepprevious,none}_test.go
static void TestInterpolate() {
    auto b = tv({I(10), I(13)}, {F(1.0), F(1.3)});
    auto timeInterp = ip::WindowStart(timeCol);
    auto valueInterp = ip::Const(valueCol, {Int64, Float64}, 9.9);  // the closure of interpolation_test.go:16-19
    TEST("Interpolate/invalid input type");
    { auto [r, e0] = rl::IntervalRolling(b, timeCol, 2, {});
      auto [f, e] = r->Interpolate({timeInterp, ip::Const(valueCol, {Int64, Boolean}, 1.0)})->Bow();
      CHECK_EQ_STR(e.msg, "intervalRolling.validateInterpolation: accepts types [int64 bool], got type float64"); }
    TEST("Interpolate/missing interval column");
    { auto [r, e0] = rl::IntervalRolling(b, timeCol, 2, {});
      auto [f, e] = r->Interpolate({valueInterp})->Bow();
      CHECK_EQ_STR(e.msg, "must keep interval column 'time'"); }
    TEST("Interpolate/empty bow");
    { auto eb = tv({}, {}); auto [r, e0] = rl::IntervalRolling(eb, timeCol, 2, {});
      auto [f, e] = r->Interpolate({timeInterp, valueInterp})->Bow(); CHECK(!e); expectEqual(f, eb); }
    TEST("Interpolate/no options");
    { auto [r, e0] = rl::IntervalRolling(b, timeCol, 2, {});
      auto [f, e] = r->Interpolate({timeInterp, valueInterp})->Bow(); CHECK(!e);
      expectEqual(f, tv({I(10), I(12), I(13)}, {F(1.0), F(9.9), F(1.3)})); }
    TEST("Interpolate/with offset");
    { auto [r, e0] = rl::IntervalRolling(b, timeCol, 2, {1, false, nullptr});
      auto [f, e] = r->Interpolate({timeInterp, valueInterp})->Bow(); CHECK(!e);
      expectEqual(f, tv({I(9), I(10), I(11), I(13)}, {F(9.9), F(1.0), F(9.9), F(1.3)})); }

    auto asc = rows({{I(10), F(10.)}, {I(15), F(15.)}, {I(17), F(17.)}});   // linear_test.go:16-24
    auto desc = rows({{I(10), F(30.)}, {I(15), F(25.)}, {I(17), F(24.)}});  // :72-80
    struct C { const char *name; BowPtr in; rl::Options o; BowPtr want; };
    std::vector<C> lin = {
        {"Linear/asc no options", asc, {}, rows({{I(10), F(10.)}, {I(12), F(12.)}, {I(14), F(14.)}, {I(15), F(15.)}, {I(16), F(16.)}, {I(17), F(17.)}})},
        {"Linear/asc with offset", asc, {3, false, nullptr}, rows({{I(9), N}, {I(10), F(10.)}, {I(11), F(11.)}, {I(13), F(13.)}, {I(15), F(15.)}, {I(17), F(17.)}})},
        {"Linear/desc no options", desc, {}, rows({{I(10), F(30.)}, {I(12), F(28.)}, {I(14), F(26.)}, {I(15), F(25.)}, {I(16), F(24.5)}, {I(17), F(24.)}})},
        {"Linear/desc with offset", desc, {3, false, nullptr}, rows({{I(9), N}, {I(10), F(30.)}, {I(11), F(29.)}, {I(13), F(27.)}, {I(15), F(25.)}, {I(17), F(24.)}})},
    };
    for (auto &c : lin) {
        TEST(c.name);
        auto [r, e0] = rl::IntervalRolling(c.in, timeCol, 2, c.o);
        auto [f, e] = r->Interpolate({ip::WindowStart(timeCol), ip::Linear(valueCol)})->Bow();
        CHECK(!e);
        expectEqual(f, c.want);
    }
    TEST("StepPrevious/with nils");  // stepprevious_test.go:136-165
    { auto in = rows({{I(10), F(1.0)}, {I(11), N}, {I(13), N}, {I(15), F(1.5)}});
      auto [r, e0] = rl::IntervalRolling(in, timeCol, 2, {});
      auto [f, e] = r->Interpolate({ip::WindowStart(timeCol), ip::StepPrevious(valueCol)})->Bow(); CHECK(!e);
      expectEqual(f, rows({{I(10), F(1.0)}, {I(11), N}, {I(12), F(1.0)}, {I(13), N}, {I(14), F(1.0)}, {I(15), F(1.5)}})); }
    TEST("None/with offset");  // none_test.go:45-64
    { auto [r, e0] = rl::IntervalRolling(b, timeCol, 2, {1, false, nullptr});
      auto [f, e] = r->Interpolate({ip::WindowStart(timeCol), ip::None(valueCol)})->Bow(); CHECK(!e);
      expectEqual(f, rows({{I(9), N}, {I(10), F(1.0)}, {I(11), N}, {I(13), F(1.3)}})); }
    TEST("Linear/bool error");  // linear_test.go:146-162 (type whitelist)
    { auto bb = NewBowFromColBasedInterfaces({timeCol, valueCol}, {Int64, Boolean}, {{I(10), I(15)}, {I(1), I(0)}}).first;
      auto [r, e0] = rl::IntervalRolling(bb, timeCol, 2, {});
      auto [f, e] = r->Interpolate({ip::WindowStart(timeCol), ip::Linear(valueCol)})->Bow();
      CHECK_EQ_STR(e.msg, "intervalRolling.validateInterpolation: accepts types [int64 float64], got type bool"); }
    TEST("Interpolate then Aggregate chained on the device");
    { auto [r, e0] = rl::IntervalRolling(asc, timeCol, 2, {});
      auto [f, e] = r->Interpolate({ip::WindowStart(timeCol), ip::Linear(valueCol)})->Aggregate({ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol)})->Bow();
      CHECK(!e);
      expectEqual(f, rows({{I(10), F(10.)}, {I(12), F(12.)}, {I(14), F(14.5)}, {I(16), F(16.5)}})); }
    TEST("Interpolate then Aggregate: the lazy Rolling (one library call) equals the two steps");
    { // 6000 irregular rows, a third of the values null: inside the fused kernel's domain (bowgpu_rolling_interpolate_aggregate)
      std::vector<Value> t, v;
      uint64_t x = 88172645463325252ull;
      int64_t ts = 1000;
      for (int i = 0; i < 6000; i++) {
          x ^= x << 13; x ^= x >> 7; x ^= x << 17;
          ts += (int64_t)(x % 19);
          t.push_back(I(ts));
          if ((x >> 20) % 3 == 0) v.push_back(N); else v.push_back(F((double)((x >> 8) % 100000) / 7.0));
      }
      auto big = tv(t, v);
      for (int64_t off : {0, 7}) {
          auto [r, e0] = rl::IntervalRolling(big, timeCol, 100, {off, false, nullptr}); CHECK(!e0);
          std::vector<rl::ColAggregation> aggs = {ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol).RenameOutput("mean"), ag::Min(valueCol).RenameOutput("min"),
                                                  ag::Count(valueCol).RenameOutput("n"), ag::First(valueCol).RenameOutput("first").SetTransformations({tr::Factor(0.5)})};
          auto lazyR = r->Interpolate({ip::WindowStart(timeCol), ip::Linear(valueCol)});
          CHECK(lazyR->lazy != nullptr);
          auto [one, e1] = lazyR->Aggregate(aggs)->Bow(); CHECK(!e1);
          int ranks = 1; bowgpu_last_call_ranks(&ranks);
          if (ranks == 1) CHECK_EQ_STR(std::string(bowgpu_last_kernel_name()), "rolling_fused_kernel");   // (row ranges interpolate, then aggregate: BOWGPU_DEVICES runs)
          auto eagerR = r->Interpolate({ip::WindowStart(timeCol), ip::Linear(valueCol)});
          auto [mid, e2] = eagerR->Bow(); CHECK(!e2); CHECK(eagerR->lazy == nullptr && mid->NumRows() > 6000);   // asking for the Bow made it
          auto [two, e3] = eagerR->Aggregate(aggs)->Bow(); CHECK(!e3);
          expectEqual(one, two);
      }
    }
}

// ---- bowfill_test.go:11-26, :156-203, :332-380, :533-546
static BowPtr fresh(Type t) {
    std::vector<std::vector<Value>> r = {{I(20), I(6), I(30), I(400), I(-10)}, {I(13), N, N, N, N}, {I(10), I(4), I(10), I(10), I(-5)},
                                         {I(0), N, I(3), I(4), I(0)}, {N, N, N, N, N}, {I(-2), I(1), N, N, I(-8)}};
    return NewBowFromRowBasedInterfaces({"a", "b", "c", "d", "e"}, {t, t, t, t, t}, r).first;
}
static void TestFillLinear() {
    auto col = [](const BowPtr &b, int c) { std::vector<Value> v; for (int r = 0; r < b->NumRows(); r++) v.push_back(b->GetValue(c, r)); return v; };
    auto eq = [](const std::vector<Value> &a, const std::vector<Value> &b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); i++) if (a[i].has_value() != b[i].has_value() || (a[i] && *a[i] != *b[i])) return false;
        return true;
    };
    TEST("FillLinear/int64 ref a fill b (desc)");
    { auto [res, e] = fresh(Int64)->FillLinear(0, 1); CHECK(!e); CHECK(eq(col(res, 1), {I(6), I(5), I(4), I(2), N, I(1)})); }
    TEST("FillLinear/int64 ref a fill e (asc)");
    { auto [res, e] = fresh(Int64)->FillLinear(0, 4); CHECK(!e); CHECK(eq(col(res, 4), {I(-10), I(-7), I(-5), I(0), N, I(-8)})); }
    TEST("FillLinear/int64 ref not sorted");
    { auto [res, e] = fresh(Int64)->FillLinear(4, 1); CHECK((bool)e); }
    TEST("FillLinear/float64 ref a fill b (desc)");
    { auto [res, e] = fresh(Float64)->FillLinear(0, 1); CHECK(!e); CHECK(eq(col(res, 1), {F(6.0), F(4.6), F(4.0), F(1.5), N, F(1.0)})); }
    TEST("FillLinear/float64 ref a fill e (asc)");
    { auto [res, e] = fresh(Float64)->FillLinear(0, 4); CHECK(!e); CHECK(eq(col(res, 4), {F(-10.0), F(-6.5), F(-5.0), F(0.0), N, F(-8.0)})); }
    TEST("FillLinear/ref null at the row stays null");
    { auto b = NewBow({NewSeries<int64_t>("int", Int64, {1, 0, 3}, {true, false, true}), NewSeries<double>("float", Float64, {1., 0., 3.}, {true, false, true})}).first;
      auto [res, e] = b->FillLinear(0, 1); CHECK(!e); CHECK(eq(col(res, 1), {F(1.), N, F(3.)})); }
    TEST("IsColSorted");
    { auto b = fresh(Int64); CHECK(b->IsColSorted(0)); CHECK(!b->IsColSorted(4)); }
    // bowfill_test.go:29-154 (int64), :204-330 (float64): Mean / Next / Previous, one column then all columns
    auto all = [&](const BowPtr &b, std::vector<std::vector<Value>> want) {
        bool ok = true;
        for (int c = 0; c < 5; c++) ok = ok && eq(col(b, c), want[c]);
        return ok;
    };
    TEST("Fill/int64 Mean one column");
    { auto [res, e] = fresh(Int64)->FillMean({1}); CHECK(!e); CHECK(eq(col(res, 1), {I(6), I(5), I(4), I(3), I(3), I(1)})); CHECK(eq(col(res, 4), {I(-10), N, I(-5), I(0), N, I(-8)})); }
    TEST("Fill/int64 Mean all columns");
    { auto [res, e] = fresh(Int64)->FillMean(); CHECK(!e);
      CHECK(all(res, {{I(20), I(13), I(10), I(0), I(-1), I(-2)}, {I(6), I(5), I(4), I(3), I(3), I(1)}, {I(30), I(20), I(10), I(3), N, N},
                      {I(400), I(205), I(10), I(4), N, N}, {I(-10), I(-8), I(-5), I(0), I(-4), I(-8)}})); }
    TEST("Fill/int64 Next all columns");
    { auto [res, e] = fresh(Int64)->FillNext(); CHECK(!e);
      CHECK(all(res, {{I(20), I(13), I(10), I(0), I(-2), I(-2)}, {I(6), I(4), I(4), I(1), I(1), I(1)}, {I(30), I(10), I(10), I(3), N, N},
                      {I(400), I(10), I(10), I(4), N, N}, {I(-10), I(-5), I(-5), I(0), I(-8), I(-8)}})); }
    TEST("Fill/int64 Previous all columns");
    { auto [res, e] = fresh(Int64)->FillPrevious(); CHECK(!e);
      CHECK(all(res, {{I(20), I(13), I(10), I(0), I(0), I(-2)}, {I(6), I(6), I(4), I(4), I(4), I(1)}, {I(30), I(30), I(10), I(3), I(3), I(3)},
                      {I(400), I(400), I(10), I(4), I(4), I(4)}, {I(-10), I(-10), I(-5), I(0), I(0), I(-8)}})); }
    TEST("Fill/float64 Mean all columns");
    { auto [res, e] = fresh(Float64)->FillMean(); CHECK(!e);
      CHECK(all(res, {{F(20), F(13), F(10), F(0), F(-1), F(-2)}, {F(6), F(5), F(4), F(2.5), F(2.5), F(1)}, {F(30), F(20), F(10), F(3), N, N},
                      {F(400), F(205), F(10), F(4), N, N}, {F(-10), F(-7.5), F(-5), F(0), F(-4), F(-8)}})); }
    TEST("Fill/float64 Next one column");
    { auto [res, e] = fresh(Float64)->FillNext({1}); CHECK(!e); CHECK(eq(col(res, 1), {F(6), F(4), F(4), F(1), F(1), F(1)})); CHECK(eq(col(res, 0), {F(20), F(13), F(10), F(0), N, F(-2)})); }
    TEST("Fill/float64 Previous all columns");
    { auto [res, e] = fresh(Float64)->FillPrevious(); CHECK(!e);
      CHECK(all(res, {{F(20), F(13), F(10), F(0), F(0), F(-2)}, {F(6), F(6), F(4), F(4), F(4), F(1)}, {F(30), F(30), F(10), F(3), F(3), F(3)},
                      {F(400), F(400), F(10), F(4), F(4), F(4)}, {F(-10), F(-10), F(-5), F(0), F(0), F(-8)}})); }
    TEST("Fill/selectCols out of range");
    { auto [res, e] = fresh(Int64)->FillNext({7}); CHECK((bool)e); CHECK(e.msg == "selectCols: colIndex '7' out of range"); }
}

// ---- bowparquet.go:44 NewBowFromParquet on the reference's own benchmark input (benchmarks/bow1-100-rows.parquet, a data file of its
// tests, copied to tests/golden/), then the calls its benchmarks make on it (bowfill_test.go:550-585, bowassertion_test.go:93-111)
static void TestParquet(const std::string &dir) {
    TEST("NewBowFromParquet/bow1-100-rows");
    auto [b, e] = NewBowFromParquet(dir + "/bow1-100-rows.parquet");
    CHECK(!e);
    if (e) { printf("   %s\n", e.msg.c_str()); return; }
    CHECK(b->NumRows() == 100);
    CHECK(b->NumCols() == 4);  // Int64_ref, Int64_no_nils_bow1, Int64_bow1, Float64_bow1 (Boolean / String columns are skipped)
    CHECK(b->ColumnName(0) == "Int64_ref" && b->ColumnName(3) == "Float64_bow1");
    CHECK(b->ColumnType(0) == Int64 && b->ColumnType(3) == Float64);
    CHECK(b->IsColSorted(0));
    CHECK(!b->IsColSorted(1));
    int nulls = 0;
    for (int r = 0; r < 100; r++) nulls += !b->GetValue(3, r).has_value();
    CHECK(nulls > 0 && nulls < 100);
    { auto [f, e2] = b->FillLinear(0, 3); CHECK(!e2); }
    { auto [f, e2] = b->FillPrevious({3}); CHECK(!e2);
      bool ok = true, seen = false;
      for (int r = 0; r < 100; r++) { if (b->GetValue(3, r).has_value()) seen = true; if (seen && !f->GetValue(3, r).has_value()) ok = false; }
      CHECK(ok); }
    auto [r, e3] = rl::IntervalRolling(b, "Int64_ref", 100, {});
    CHECK(!e3);
    auto [a, e4] = r->Aggregate({ag::WindowStart("Int64_ref"), ag::ArithmeticMean("Float64_bow1")})->Bow();
    CHECK(!e4);
    CHECK(a->NumRows() > 0);
    TEST("NewBowFromParquet/missing file");
    { auto [x, e5] = NewBowFromParquet(dir + "/nope.parquet"); CHECK((bool)e5); }
}

// ---- rolling/aggregation/XXXbenchmarks_test.go:125-138 shape, small: IntervalRolling + Aggregate(WindowStart, ArithmeticMean)
static void TestBenchShape() {
    TEST("bench shape 1e5 rows");
    const int n = 100000;
    std::vector<int64_t> t(n);
    std::vector<double> v(n);
    for (int i = 0; i < n; i++) { t[i] = i; v[i] = (double)((i * 2654435761u) % 1000) / 1000.0; }
    auto b = NewBow({NewSeries<int64_t>(timeCol, Int64, t), NewSeries<double>(valueCol, Float64, v)}).first;
    auto [r, e0] = rl::IntervalRolling(b, timeCol, 10, {});
    CHECK(!e0);
    auto [a, e] = r->Aggregate({ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol)})->Bow();
    CHECK(!e);
    CHECK(a->NumRows() == n / 10);
    bool ok = true;
    for (int w = 0; w < n / 10 && ok; w++) {
        double s = 0;
        for (int k = 0; k < 10; k++) s += v[w * 10 + k];
        ok = std::get<int64_t>(*a->GetValue(0, w)) == w * 10 && std::get<double>(*a->GetValue(1, w)) == s / 10.0;
    }
    CHECK(ok);
}

// rl::SetGPUDevices: the bench-shaped call spread over this box's device listed three times - the same Bow, bit for bit - and the list off again
static void TestSetGPUDevices() {
    TEST("SetGPUDevices");
    const int n = 300000;
    std::vector<int64_t> t(n);
    std::vector<Value> v(n);
    uint64_t x = 99;
    int64_t ts = 0;
    for (int i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        ts += 1 + (int64_t)(x % 7);
        t[i] = ts;
        if ((x >> 20) % 4 == 0) v[i] = N; else v[i] = F((double)((x >> 8) % 100000) / 7.0);
    }
    std::vector<Value> tc(n);
    for (int i = 0; i < n; i++) tc[i] = I(t[i]);
    auto b = tv(tc, v);
    std::vector<rl::ColAggregation> aggs = {ag::WindowStart(timeCol), ag::ArithmeticMean(valueCol).RenameOutput("mean"), ag::Min(valueCol).RenameOutput("min"),
                                            ag::Count(valueCol).RenameOutput("n"), ag::Last(valueCol).RenameOutput("last")};
    auto [r, e0] = rl::IntervalRolling(b, timeCol, 50, {3, false, nullptr}); CHECK(!e0);
    auto [one, e1] = r->Aggregate(aggs)->Bow(); CHECK(!e1);
    int ranks = 0; bowgpu_last_call_ranks(&ranks); CHECK(ranks == 1 || getenv("BOWGPU_DEVICES") != nullptr);
    std::vector<int> before(64); int nb = 0; bowgpu_get_devices(before.data(), 64, &nb); before.resize(nb);
    CHECK(!rl::SetGPUDevices({0, 0, 0}));
    bowgpu_set_fanout_min_rows(50000);
    auto [many, e2] = r->Aggregate(aggs)->Bow(); CHECK(!e2);
    bowgpu_last_call_ranks(&ranks); CHECK(ranks == 3);
    expectEqual(one, many);
    CHECK((bool)rl::SetGPUDevices({0, 99}));   // no such device: the list stays as it was
    CHECK(!rl::SetGPUDevices(before));
    bowgpu_set_fanout_min_rows(getenv("BOWGPU_FANOUT_MIN_ROWS") ? atoll(getenv("BOWGPU_FANOUT_MIN_ROWS")) : (1ll << 20));
}

int main(int argc, char **argv) {
    int ndev = 0;
    if (bowgpu_device_count(&ndev) != 0 || ndev == 0) {
        printf("no GPU: %s\n", bowgpu_last_error());
        return 77;
    }
    TestNumWindows();
    TestIteratorInit();
    TestIterate();
    TestAggregateDriver();
    TestUnsetInclusive();
    TestReducers();
    TestFactor();
    TestWhole();
    TestInterpolate();
    TestFillLinear();
    TestParquet(argc > 1 ? argv[1] : "tests/golden");
    TestBenchShape();
    TestSetGPUDevices();
    printf("%d checks, %d failures\n", g_checks, g_fail);
    int64_t listed = 0, served = 0;
    bowgpu_fanout_counts(&listed, &served);
    printf("fan-out: %lld of %lld Rolling.Aggregate calls ran as row ranges\n", (long long)served, (long long)listed);
    return g_fail ? 1 : 0;
}
