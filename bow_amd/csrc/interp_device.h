// interp_device.h — what an interpolator gives the synthetic row of a window (shared by interpolate.hip and ts_nulls.hip)
#pragma once
#include "agg_device.h"

namespace bowgpu {

// a neighbour of a synthetic row: the nearest valid row of the column before its window's FirstIndex / from it on
struct NbPoint { int64_t t; uint64_t bits; int has; };

// (double)t for a 64-bit integer: exact below 2^52 in magnitude as high half * 2^32 + low half (two 32-bit conversions and one fused
// multiply-add whose result is representable, so it rounds nothing) - the general conversion, a dozen vector instructions, only for
// the lanes beyond that
__device__ __forceinline__ double i64_to_f64(int64_t t) {
    const int32_t hi = (int32_t)(t >> 32);
    if (hi >= -(1 << 20) && hi < (1 << 20)) return __builtin_fma((double)hi, 4294967296.0, (double)(uint32_t)t);
    return (double)t;
}

// the value an interpolator gives the synthetic row of a window starting at sk whose FirstIndex is row a
// (interpolation/windowstart.go:10-12, linear.go:12-37, stepprevious.go:11-24, none.go); pp / np = the valid row before a /
// the valid row from a on, found once per run of synthetic rows (ts has no nulls: both-valid == value valid)
// (IC: InterpCol, or rolling_fused.hip's FusedCol - the same fields)
template <class IC>
__device__ __forceinline__ void synth_value_pt(const IC &ic, int64_t sk, const NbPoint &pp, const NbPoint &np, uint64_t *bits_out,
                                               int *valid_out) {
    const bool is_int = ic.type == BOWGPU_INT64;
    uint64_t bits = 0;
    int valid = 0;
    switch (ic.kind) {
    case BOWGPU_INTERP_WINDOW_START:
        bits = is_int ? (uint64_t)sk : (uint64_t)__double_as_longlong(i64_to_f64(sk));
        valid = 1;
        break;
    case BOWGPU_INTERP_CONST:
        bits = is_int ? (uint64_t)go_f64_to_i64(ic.const_value) : (uint64_t)__double_as_longlong(ic.const_value);
        valid = 1;
        break;
    case BOWGPU_INTERP_LINEAR: {
        double t0, v0;
        if (pp.has) { t0 = i64_to_f64(pp.t); v0 = bits_to_f64(pp.bits, ic.type); }
        else if (ic.has_prev && ic.prev_t_valid && ic.prev_v_valid) { t0 = ic.prev_t; v0 = ic.prev_v; }
        else break;
        double t2, v2;
        if (np.has) { t2 = i64_to_f64(np.t); v2 = bits_to_f64(np.bits, ic.type); }
        else if (ic.next_valid) { t2 = ic.next_t; v2 = ic.next_v; }
        else break;
        const double coef = (i64_to_f64(sk) - t0) / (t2 - t0);
        const double r = ((v2 - v0) * coef) + v0;
        bits = is_int ? (uint64_t)go_f64_to_i64(r) : (uint64_t)__double_as_longlong(r);
        valid = 1;
        break;
    }
    case BOWGPU_INTERP_STEP_PREVIOUS:
        if (pp.has) { bits = pp.bits; valid = 1; }
        else if (ic.has_prev && ic.prev_v_valid) {
            bits = is_int ? (uint64_t)ic.prev_v_i64 : (uint64_t)__double_as_longlong(ic.prev_v);
            valid = 1;
        }
        break;
    default: break;
    }
    *bits_out = valid ? bits : 0;
    *valid_out = valid;
}

}  // namespace bowgpu
