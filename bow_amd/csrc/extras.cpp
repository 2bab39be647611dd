// extras.cpp — C-ABI entry points beyond Rolling.Aggregate (placeholders until their kernels land).
#include "common.h"

using namespace bowgpu;

extern "C" {

int bowgpu_window_bounds(const bowgpu_col *, int64_t, const bowgpu_options *, int64_t *, int64_t *, int64_t *, uint8_t *, int32_t) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_window_bounds: not implemented yet");
}
int bowgpu_aggregate_whole(const bowgpu_col *, int32_t, int32_t, const bowgpu_agg *, int32_t, bowgpu_out *) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_aggregate_whole: not implemented yet");
}
int bowgpu_rolling_interpolate_count(const bowgpu_col *, int32_t, int32_t, int64_t, const bowgpu_options *, const bowgpu_interp *, int32_t, int64_t *) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_rolling_interpolate_count: not implemented yet");
}
int bowgpu_rolling_interpolate_fill(const bowgpu_col *, int32_t, int32_t, int64_t, const bowgpu_options *, const bowgpu_interp *, int32_t, bowgpu_out *) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_rolling_interpolate_fill: not implemented yet");
}
int bowgpu_fill_linear(const bowgpu_col *, int32_t, int32_t, int32_t, bowgpu_out *, int32_t *) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_fill_linear: not implemented yet");
}
int bowgpu_is_col_sorted(const bowgpu_col *, int32_t *) {
    return fail(BOWGPU_ERR_UNSUPPORTED, "bowgpu_is_col_sorted: not implemented yet");
}
}
