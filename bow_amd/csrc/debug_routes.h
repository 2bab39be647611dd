// debug_routes.h - the test / A-B bits of the per-thread route mask (bowgpu_debug_set_route; the two bits a caller may want are in
// include/bowgpu.h).  NOT part of the ABI: the parity tests and the scratch/ scripts use them (bow_amd/capi.py holds the same
// values, tests/test_abi_symbols.py compares) to run one call through every kernel that can take it.
#pragma once
enum {
    BOWGPU_ROUTE_NO_SIMPLE = 1,          // wave-tile kernels (rolling_simple / rolling_tw) off: rolling_wave_kernel / the general kernel
    BOWGPU_ROUTE_FORCE_GENERAL = 2,      // everything through rolling_agg_kernel
    BOWGPU_ROUTE_NO_LONG_ONLY = 4,       // long windows through the tile kernels' queue instead of the long-only forms
    BOWGPU_ROUTE_LONG_CLASSIC = 8,       // long-only calls: bisection form only
    BOWGPU_ROUTE_LONG_STREAM_ALL = 16,   // long-only calls: streaming form for every reducer set
    BOWGPU_ROUTE_SIMPLE_SMALL_LIST = 32, // rolling_simple_kernel: the 254-head list whatever the plan says
    BOWGPU_ROUTE_SIMPLE_LARGE_LIST = 64, // ... the 400-head list
    BOWGPU_ROUTE_TW_F64 = 128,           // rolling_tw_kernel: 64-bit timestamps in LDS even where 32-bit offsets are exact
    BOWGPU_ROUTE_SIMPLE_PADDED = 256,    // rolling_simple_kernel: the padded staging also for the calls that would take the unpadded instantiation (short windows, one Float64 column without nulls)
    BOWGPU_ROUTE_INTERP_TILE = 512,      // Interpolate (exclusive windows): interp_tile_kernel instead of interp_wave3_kernel
    BOWGPU_ROUTE_NO_FUSED = 4096,        // bowgpu_rolling_interpolate_aggregate: the two calls through device temporaries even where rolling_fused_kernel applies
    BOWGPU_ROUTE_TW_ROWS = 8192,         // time-weighted reducers on a nullable column: rolling_tw_kernel's row-space form even where rolling_twc_kernel (valid points compacted) applies
    BOWGPU_ROUTE_INTERP_COPIES = 16384,  // Interpolate: output bitmaps through the zeroed working copies + the counting pass even where interp_wave3_kernel could write them in place
    BOWGPU_ROUTE_QUEUE_HOST = 32768,     // windows a tile pass queues: the host reads the counts and launches the long-window machinery (rounds 1 - 5) even where long_queue_kernel would be enqueued behind the tile kernel
    BOWGPU_ROUTE_QUEUE_DEVICE = 65536,   // ... long_queue_kernel behind every tile pass, whatever the call's average window length
    BOWGPU_ROUTE__ALL = 131071           // every defined bit, the two public ones (1024, 2048) included
};
