// render_common.h — device helpers shared by the composite kernels (render.hip, render_blocks.hip).
#pragma once
#include <cstdlib>
#include "dvs_device.h"

#define RB 256

// Timing-only ablation knobs (DVS_TR_DEBUG, DVS_MM_DEBUG, DVS_A9V_NOHOIST, DVS_BWD_EXTRA_LDS) exist only in experiment builds
// (-DDVS_EXPERIMENT, tools/xbuild.sh): the release library neither reads those variables nor carries the code they select
// (tests/test_abi.py::test_release_build_has_no_ablation_knobs). DVS_EXPERIMENT_ON folds the knob tests away at compile time.
#ifdef DVS_EXPERIMENT
#define DVS_EXPERIMENT_ON 1
#define DVS_DBG_PARAM , int dbg_arg /*experiment builds: ablation bits (timing only)*/
#define DVS_DBG_PASS(x) , (x)
#define DVS_DBG_VALUE dbg_arg
#else
#define DVS_EXPERIMENT_ON 0
#define DVS_DBG_PARAM              /* release kernels and helpers have no ablation argument at all */
#define DVS_DBG_PASS(x)
#define DVS_DBG_VALUE 0
#endif

// Experiment knob of the occupancy measurements (tools/bwd_probe.py): DVS_BWD_EXTRA_LDS = bytes of dynamic LDS added to the composite
// backward launches so that fewer workgroups fit a CU. Read ONCE per process and clamped to what a launch can carry.
static inline size_t dvs_experiment_extra_lds() {
#ifndef DVS_EXPERIMENT
    return 0;
#else
    static const size_t v = [] {
        const char* e = getenv("DVS_BWD_EXTRA_LDS");
        long x = e ? atol(e) : 0;
        if (x < 0) x = 0;
        if (x > 64 * 1024) x = 64 * 1024;
        return (size_t)x;
    }();
    return v;
#endif
}
// An integer knob of an experiment build (0 in release builds, where the variable is not read at all).
static inline int dvs_experiment_int(const char* name) {
#ifdef DVS_EXPERIMENT
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

// Per-view backgrounds of a multi-view batch: FIRST kernel parameter of the composite kernels, read through the kernarg segment
// pointer (see DvsCams in dvs_device.h). One launch covers the tiles of all views: global tile t = view * tiles_per_view + tile.
struct ViewBg { float bg[DVS_MAX_VIEWS][4]; };
__device__ __forceinline__ float3 dvs_load_bg(int v) {
    typedef const __attribute__((address_space(4))) float* KF;
    const KF f = (KF)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)v * 4;
    return make_float3(f[0], f[1], f[2]);
}

static inline ViewBg make_view_bg(int n_views, const float* bgs /*[n_views][3]*/) {
    ViewBg b{};
    for (int v = 0; v < n_views && v < DVS_MAX_VIEWS; ++v) for (int k = 0; k < 3; ++k) b.bg[v][k] = bgs[3 * v + k];
    return b;
}

// blockIdx -> tile: consecutive workgroups land on different XCDs (b % 8), so give each XCD a
// contiguous band of tiles; neighbouring tiles share splats and therefore L2 lines.
__device__ __forceinline__ int tile_of_block(int b, int num_tiles) {
    const int chunk = (num_tiles + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// 12 per-lane partials -> 12 wave totals in 9 VALU swaps + 12 LDS-crossbar swizzles (a plain DPP tree needs 6 DPP adds per value).
// Two halving steps with the gfx950 swap instructions fold the 64 lanes to 16 while packing 4 values per
// register (v_permlane32_swap: lanes 32-63 of A <-> lanes 0-31 of B; v_permlane16_swap: odd 16-lane rows of
// A <-> even rows of B), then a 4-step butterfly (ds_swizzle, see row_sum) finishes inside each 16-lane row.
// Result: q[k] holds, in every lane of row r, the total of value index kRowValue[k][r]:
//   q[0] rows -> v0,v2,v1,v3   q[1] rows -> v4,v6,v5,v7   q[2] rows -> v8,v10,v9,v11
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// 16-lane butterfly through ds_swizzle_b32 (the LDS crossbar; no LDS memory is touched) + a plain v_add per stage: the
// lane exchange leaves the VALU, which is the unit the backward kernel is bound by (a DPP add costs two VALU slots;
// measured -9 % kernel time against the DPP butterfly). Every lane of a row ends with the row total.
__device__ __forceinline__ float row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (1 << 10) | 0x1f));     // xor 1
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (2 << 10) | 0x1f));     // xor 2
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (4 << 10) | 0x1f));     // xor 4
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (8 << 10) | 0x1f));     // xor 8
    return v;
}
// NV = number of live values (11 with abs-grad, 9 without). The odd value out has no partner to be packed with: instead of a
// swap against a zero register (v_mov + v_permlane32_swap + v_add = 5 issue slots) it is folded across the two wave halves with
// ds_bpermute_b32 (lane ^ 32; LDS crossbar) + one v_add; both halves then hold the folded value, which only puts a duplicate into
// a row no lane publishes. `xaddr` = (lane ^ 32) * 4.
template <int NV>
__device__ __forceinline__ void wave_reduce12(const float v[12], float q[3], int xaddr) {
    const float h0 = swap32_add(v[0], v[1]), h1 = swap32_add(v[2], v[3]), h2 = swap32_add(v[4], v[5]);
    const float h3 = swap32_add(v[6], v[7]);
    float h4, h5;
    if (NV == 11) {
        h4 = swap32_add(v[8], v[9]);
        h5 = v[10] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[10])));
    } else {                                                   // 9 values: v[8] is the odd one, the sixth register is empty
        h4 = v[8] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[8])));
        h5 = 0.f;
    }
    q[0] = row_sum(swap16_add(h0, h1));
    q[1] = row_sum(swap16_add(h2, h3));
    q[2] = row_sum(swap16_add(h4, h5));
}

