// render.hip — A7 alpha-composite forward and A8 alpha-composite backward for gfx950.
//
// One 16x16 tile per 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel quadrant so a
// splat's footprint can be rejected per wave. The tile's depth-ordered splat list is streamed
// through LDS in batches of 256 (one gathered splat per lane, then broadcast reads).
// Backward replays the list back-to-front, reduces the 9 (11 with abs-grad) per-splat partials over
// the 64 lanes with DPP row operations (no LDS traffic) and issues one hardware fp32 atomic per value
// per wave (global_atomic_add_f32; built with -munsafe-fp-atomics, no CAS loop).
//
// Reference anchors (fenghuayumo/DIVSHOT): alpha rule and thresholds gsplat_ps.hlsl:60-65 (the viewer
// caps alpha at 0.999; the trainer constant fixed by this build is 0.99, SURVEY.md §8(a) A-notes),
// blend order renderer/gaussian.cpp:440, 16x16 groups gaussian_common.hlsl:162-163, abs-grad flag
// application/diverseshot-cli/source/main.cpp:44.
#include "dvs_device.h"
#include "dvs_kernels.h"

#define RB 256

// blockIdx -> tile: consecutive workgroups land on different XCDs (b % 8), so give each XCD a
// contiguous band of tiles; neighbouring tiles share splats and therefore L2 lines.
__device__ __forceinline__ int tile_of_block(int b, int num_tiles) {
    const int chunk = (num_tiles + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

__device__ __forceinline__ float dpp_add(float v, const int ctrl, const int row_mask) {
    // old = 0, bound_ctrl = true: lanes without a source (or masked rows) add 0
    switch (ctrl) {   // ctrl / row_mask must be immediates
        case 0xB1:  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
        case 0x4E:  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
        case 0x141: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
        case 0x140: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
        case 0x142: return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true));
        default:    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true));
    }
    (void)row_mask;
}
// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    v = dpp_add(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    v = dpp_add(v, 0x141, 0xF);   // row_half_mirror
    v = dpp_add(v, 0x140, 0xF);   // row_mirror
    v = dpp_add(v, 0x142, 0xA);   // row_bcast:15 into rows 1,3
    v = dpp_add(v, 0x143, 0xC);   // row_bcast:31 into rows 2,3
    return v;
}

// ---- A7 -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RB)
k_render_fwd(int W, int H, int tiles_x, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ sorted_splat, const float2* __restrict__ mean2d,
             const float4* __restrict__ conic_opacity, const float* __restrict__ rgb, float bg0, float bg1, float bg2,
             float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ float2 s_xy[RB];
    __shared__ float4 s_co[RB];
    __shared__ float4 s_rgb[RB];
    const int tile = tile_of_block(blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tx * DVS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DVS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;

    for (int base = 0; base < total; base += RB) {
        if (__syncthreads_and(done)) break;
        const int cnt = min(RB, total - base);
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = sorted_splat[range.x + base + threadIdx.x];
            s_xy[threadIdx.x] = mean2d[id];
            s_co[threadIdx.x] = conic_opacity[id];
            s_rgb[threadIdx.x] = make_float4(rgb[3 * (size_t)id], rgb[3 * (size_t)id + 1], rgb[3 * (size_t)id + 2], 0.f);
        }
        __syncthreads();
        if (__all(done)) continue;               // this wave's quadrant is finished
        for (int j = 0; j < cnt; ++j) {
            if (__all(done)) break;              // evaluated with the whole wave active (wave-uniform exit)
            if (done) continue;
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            if (power > 0.f) continue;
            const float alpha = fminf(DVS_ALPHA_MAX, co.w * __expf(power));
            if (alpha < DVS_ALPHA_MIN) continue;
            const float test_T = T * (1.f - alpha);
            if (test_T < DVS_T_STOP) { done = true; continue; }
            const float4 c = s_rgb[j];
            const float w = alpha * T;
            C0 += c.x * w; C1 += c.y * w; C2 += c.z * w;
            T = test_T;
            last = (uint32_t)(base + j + 1);
        }
    }
    if (inside) {
        const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg0;
        out_color[P + pix] = C1 + T * bg1;
        out_color[2 * P + pix] = C2 + T * bg2;
    }
}

// ---- A8 -------------------------------------------------------------------------------------------
template <bool ABSGRAD>
__global__ void __launch_bounds__(RB)
k_render_bwd(int W, int H, int tiles_x, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ sorted_splat, const float2* __restrict__ mean2d,
             const float4* __restrict__ conic_opacity, const float* __restrict__ rgb, float bg0, float bg1, float bg2,
             const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
             float* __restrict__ dL_dmean2d, float* __restrict__ dL_dconic_opacity, float* __restrict__ dL_drgb,
             float* __restrict__ absgrad) {
    __shared__ float2 s_xy[RB];
    __shared__ float4 s_co[RB];
    __shared__ float4 s_rgb[RB];
    __shared__ uint32_t s_id[RB];
    __shared__ uint32_t s_max[RB / 64];
    const int tile = tile_of_block(blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tx * DVS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DVS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t P = (size_t)W * H, pix = (size_t)py * W + px;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) { dLp0 = dL_dout[pix]; dLp1 = dL_dout[P + pix]; dLp2 = dL_dout[2 * P + pix]; }
    const float bg_dot = (bg0 * dLp0 + bg1 * dLp1) + bg2 * dLp2;

    // entries beyond the deepest contributor of any pixel of this tile are never touched
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (lane == 0) s_max[wave] = wmax;
    __syncthreads();
    const uint32_t wave_last = wmax;
    uint32_t todo = 0;
#pragma unroll
    for (int w = 0; w < RB / 64; ++w) todo = max(todo, s_max[w]);
    if (todo == 0) return;

    float T = T_final;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;

    const int nbatch = (int)((todo + RB - 1) / RB);
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * RB;
        const int cnt = min(RB, (int)todo - base);
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = sorted_splat[range.x + base + threadIdx.x];
            s_id[threadIdx.x] = id;
            s_xy[threadIdx.x] = mean2d[id];
            s_co[threadIdx.x] = conic_opacity[id];
            s_rgb[threadIdx.x] = make_float4(rgb[3 * (size_t)id], rgb[3 * (size_t)id + 1], rgb[3 * (size_t)id + 2], 0.f);
        }
        __syncthreads();
        if ((uint32_t)base >= wave_last) continue;      // nothing in this batch reaches this wave's pixels
        for (int j = cnt - 1; j >= 0; --j) {
            const uint32_t k = (uint32_t)(base + j);     // 0-based list position; contributor index k+1
            if (k >= wave_last) continue;
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            const float G = __expf(power);
            const float oa = co.w * G;
            const float alpha = fminf(DVS_ALPHA_MAX, oa);
            const bool contrib = (k < last) && !(power > 0.f) && !(alpha < DVS_ALPHA_MIN);
            if (!__any(contrib)) continue;
            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
            float a_mx = 0.f, a_my = 0.f;
            if (contrib) {
                const float4 c = s_rgb[j];
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                lc0 = c.x; lc1 = c.y; lc2 = c.z;
                float dL_dalpha = ((c.x - acc0) * dLp0 + (c.y - acc1) * dLp1) + (c.z - acc2) * dLp2;
                g_r = dchannel_dcolor * dLp0; g_g = dchannel_dcolor * dLp1; g_b = dchannel_dcolor * dLp2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                if (!(oa > DVS_ALPHA_MAX)) {
                    const float dL_dG = co.w * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co.x - gdy * co.y;
                    const float dG_ddely = -gdy * co.z - gdx * co.y;
                    g_mx = dL_dG * dG_ddelx; g_my = dL_dG * dG_ddely;
                    if (ABSGRAD) { a_mx = fabsf(g_mx); a_my = fabsf(g_my); }
                    g_ca = -0.5f * gdx * dx * dL_dG;
                    g_cb = -gdx * dy * dL_dG;
                    g_cc = -0.5f * gdy * dy * dL_dG;
                    g_op = G * dL_dalpha;
                }
            }
            g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
            g_ca = wave_sum_to_lane63(g_ca); g_cb = wave_sum_to_lane63(g_cb); g_cc = wave_sum_to_lane63(g_cc);
            g_op = wave_sum_to_lane63(g_op);
            g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
            if (ABSGRAD) { a_mx = wave_sum_to_lane63(a_mx); a_my = wave_sum_to_lane63(a_my); }
            if (lane == 63) {
                const size_t id = s_id[j];
                atomicAdd(&dL_dmean2d[2 * id], g_mx); atomicAdd(&dL_dmean2d[2 * id + 1], g_my);
                atomicAdd(&dL_dconic_opacity[4 * id], g_ca); atomicAdd(&dL_dconic_opacity[4 * id + 1], g_cb);
                atomicAdd(&dL_dconic_opacity[4 * id + 2], g_cc); atomicAdd(&dL_dconic_opacity[4 * id + 3], g_op);
                atomicAdd(&dL_drgb[3 * id], g_r); atomicAdd(&dL_drgb[3 * id + 1], g_g); atomicAdd(&dL_drgb[3 * id + 2], g_b);
                if (ABSGRAD) { atomicAdd(&absgrad[2 * id], a_mx); atomicAdd(&absgrad[2 * id + 1], a_my); }
            }
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
hipError_t dvs_launch_render_fwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* mean2d, const float* conic_opacity,
                                 const float* rgb, const float bg[3], float* out_color, float* final_T,
                                 uint32_t* n_contrib) {
    const int num_tiles = tiles_x * tiles_y;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges, sorted_splat,
                       (const float2*)mean2d, (const float4*)conic_opacity, rgb, bg[0], bg[1], bg[2], out_color, final_T,
                       n_contrib);
    return hipGetLastError();
}

hipError_t dvs_launch_render_bwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* mean2d, const float* conic_opacity,
                                 const float* rgb, const float bg[3], const float* final_T, const uint32_t* n_contrib,
                                 const float* dL_dout, float* dL_dmean2d, float* dL_dconic_opacity, float* dL_drgb,
                                 float* absgrad) {
    const int num_tiles = tiles_x * tiles_y;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    if (absgrad)
        hipLaunchKernelGGL(k_render_bwd<true>, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges,
                           sorted_splat, (const float2*)mean2d, (const float4*)conic_opacity, rgb, bg[0], bg[1], bg[2], final_T,
                           n_contrib, dL_dout, dL_dmean2d, dL_dconic_opacity, dL_drgb, absgrad);
    else
        hipLaunchKernelGGL(k_render_bwd<false>, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges,
                           sorted_splat, (const float2*)mean2d, (const float4*)conic_opacity, rgb, bg[0], bg[1], bg[2], final_T,
                           n_contrib, dL_dout, dL_dmean2d, dL_dconic_opacity, dL_drgb, absgrad);
    return hipGetLastError();
}
