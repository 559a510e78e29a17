// render.hip — A7 alpha-composite forward and A8 alpha-composite backward for gfx950.
//
// One 16x16 tile per 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel quadrant so a
// splat's footprint can be rejected per wave. The tile's depth-ordered splat list is streamed
// through LDS in batches of 256 (one gathered splat per lane, then broadcast reads).
// Backward replays the list back-to-front, reduces the 9 (11 with abs-grad) per-splat partials over
// the 64 lanes with v_permlane32/16_swap + ds_swizzle butterflies (LDS crossbar, no LDS memory) and publishes them with ONE
// hardware fp32 atomic instruction per (wave, splat): 11 lanes add 11 consecutive floats of the splat's
// 48-byte gradient row (global_atomic_add_f32; built with -munsafe-fp-atomics, no CAS loop).
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

// ---- batch staging + per-quadrant culling masks -------------------------------------------------------
// Lane t of the workgroup gathers splat t of the batch into LDS and tests the ellipse on which the splat reaches
// alpha = 1/255 ( d^T Q d = 2 ln(255 o), Q = conic ) against the four 8x8 quadrants of the tile (exact
// ellipse-rectangle test). One ballot per quadrant turns the tests into 64-bit wave masks: the wave that
// owns quadrant q later walks only the set bits (s_ff1 / s_flbit, scalar unit) and never spends a vector
// instruction on a splat that cannot touch its pixels. The bound is inflated (1e-4 rel + 1e-3) so the
// exact per-pixel alpha test — unchanged — decides every contribution: results are identical to a full walk.
struct __attribute__((aligned(16))) BatchLds {
    // Four 16-B-stride arrays (one address VGPR serves every read of a visit), packed so that each read is a full
    // ds_read_b128 or a ds_read_b32 (a 12-byte ds_read_b96 costs twice the LDS cycles of a b128):
    // the alpha test of a visit needs x, y, cs.xyz, o = one b128 + one b64; the rest is read only by contributing visits.
    // cs = (-0.5 log2e a, -log2e b, -0.5 log2e c): exp(power) = exp2(cs.x dx^2 + cs.y dx dy + cs.z dy^2)
    float4 xyc[RB];     // mean x, mean y, cs.x, cs.y
    float4 zoir[RB];    // cs.z, opacity | splat id bits, colour r
    float4 cog[RB];     // conic a, b, c as preprocess wrote them (gradient formulas), colour g
    float4 bl[RB];      // colour b in .x
    uint64_t qmask[RB / 64][4];
};

__device__ __forceinline__ void stage_batch(BatchLds& L, const uint32_t* __restrict__ sorted_splat, uint32_t first, int cnt,
                                            const float4* __restrict__ splat2d, float tile_x0, float tile_y0) {
    const int t = threadIdx.x;
    uint32_t qm = 0;
    if (t < cnt) {
        const uint32_t id = sorted_splat[first + t];
        // the splat's 64-B record (dvs_fwd_state.splat2d): three 16-B loads from ONE cache line
        const float4 r0 = splat2d[4 * (size_t)id], r1 = splat2d[4 * (size_t)id + 1];
        const float bl = splat2d[4 * (size_t)id + 2].x;
        const float2 xy = make_float2(r0.x, r0.y);
        const float4 co = make_float4(r0.z, r0.w, r1.x, r1.y);
        const float3 col = make_float3(r1.z, r1.w, bl);
        L.xyc[t] = make_float4(xy.x, xy.y, -0.72134752044448170f * co.x, -1.4426950408889634f * co.y);
        L.zoir[t] = make_float4(-0.72134752044448170f * co.z, co.w, __uint_as_float(id), col.x);
        L.cog[t] = make_float4(co.x, co.y, co.z, col.y);
        L.bl[t].x = col.z;
        // The splat can reach alpha >= 1/255 only where q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 o), d = pixel - mean.
        // Minimise the convex form q over each quadrant's pixel rectangle (exact: origin inside -> 0, otherwise the
        // minimum lies on one of the four edges) and keep the splat for that quadrant iff the minimum is within the bound.
        const float bound = 2.0f * __logf(255.0f * co.w) * 1.0001f + 1e-3f;
        const float a = co.x, b = co.y, c = co.z;
        const float nb_c = -b / c, nb_a = -b / a;
        const float ox = tile_x0 - xy.x, oy = tile_y0 - xy.y;                 // tile origin relative to the mean
        auto qmin = [&](float x0, float x1, float y0, float y1) -> float {
            if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return 0.f;
            auto qf = [&](float x, float y) { return a * x * x + 2.f * b * x * y + c * y * y; };
            const float e0 = qf(x0, fminf(fmaxf(nb_c * x0, y0), y1)), e1 = qf(x1, fminf(fmaxf(nb_c * x1, y0), y1));
            const float e2 = qf(fminf(fmaxf(nb_a * y0, x0), x1), y0), e3 = qf(fminf(fmaxf(nb_a * y1, x0), x1), y1);
            return fminf(fminf(e0, e1), fminf(e2, e3));
        };
        // NaN-safe: a failed comparison keeps the splat
        qm = (!(qmin(ox, ox + 7.f, oy, oy + 7.f) > bound) ? 1u : 0u) | (!(qmin(ox + 8.f, ox + 15.f, oy, oy + 7.f) > bound) ? 2u : 0u) |
             (!(qmin(ox, ox + 7.f, oy + 8.f, oy + 15.f) > bound) ? 4u : 0u) | (!(qmin(ox + 8.f, ox + 15.f, oy + 8.f, oy + 15.f) > bound) ? 8u : 0u);
    }
    const uint64_t m0 = __ballot(qm & 1u), m1 = __ballot(qm & 2u), m2 = __ballot(qm & 4u), m3 = __ballot(qm & 8u);
    if ((t & 63) == 0) {
        uint64_t* dst = L.qmask[t >> 6];
        dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3;
    }
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// ---- A7 -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RB)
k_render_fwd(int W, int H, int tiles_x, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d, float bg0, float bg1, float bg2,
             float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ BatchLds L;
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
        stage_batch(L, sorted_splat, range.x + base, cnt, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE));
        __syncthreads();
        if (__all(done)) continue;               // this wave's quadrant is finished
        // Predicated body (no per-lane branches: the scalar unit is shared by the CU's four SIMDs and a branchy
        // body made this kernel scalar-bound); the wave-uniform "everyone finished" exit is checked per 64-splat word.
#pragma unroll 1
        for (int lw = 0; lw < RB / 64; ++lw) {
            uint64_t m = uniform_u64(L.qmask[lw][wave]);
            if (m == 0) continue;
            if (__all(done)) break;
            while (m) {
                const int j = lw * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const float4 xy = L.xyc[j];
                const float4 zo = L.zoir[j];
                const float4 cs = make_float4(xy.z, xy.w, zo.x, zo.y);
                const float3 c = make_float3(zo.w, L.cog[j].w, L.bl[j].x);
                const float dx = xy.x - pxf, dy = xy.y - pyf;
                // log2 of the Gaussian falloff: p2 = log2e * power (sign unchanged), one v_exp_f32, no extra multiply
                const float p2 = __builtin_fmaf(cs.z * dy, dy, __builtin_fmaf(cs.y, dy, cs.x * dx) * dx);
                const float alpha = fminf(DVS_ALPHA_MAX, cs.w * __builtin_amdgcn_exp2f(p2));
                const bool valid = !done && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
                const float aT = alpha * T;
                const float test_T = T - aT;                       // = T (1 - alpha)
                const bool stop = valid && (test_T < DVS_T_STOP);
                const bool take = valid && !stop;
                done = done || stop;
                const float w = take ? aT : 0.f;
                C0 = __builtin_fmaf(c.x, w, C0); C1 = __builtin_fmaf(c.y, w, C1); C2 = __builtin_fmaf(c.z, w, C2);
                T = T - w;
                last = take ? (uint32_t)(base + j + 1) : last;
            }
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
             const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d, float bg0, float bg1, float bg2,
             const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
             float* __restrict__ grow /*[n,12]: Sx Sy Sxx Sxy Syy So r g b |mx| |my| pad (moments, see the loop body)*/) {
    __shared__ BatchLds L;
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
    // which reduced value this lane publishes: lane c (< 3) of row r carries q[c] = value kv (see wave_reduce12)
    const int lrow = lane >> 4, lcol = lane & 15;
    const int kv = lcol * 4 + ((lrow == 1) ? 2 : (lrow == 2) ? 1 : lrow);
    const bool publisher = lcol < 3 && kv < (ABSGRAD ? 11 : 9);
    const int xaddr = (lane ^ 32) << 2;                       // ds_bpermute address of the partner lane in the other wave half

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
    const int wave_last = (int)__builtin_amdgcn_readfirstlane(wmax);
    uint32_t todo = 0;
#pragma unroll
    for (int w = 0; w < RB / 64; ++w) todo = max(todo, s_max[w]);
    if (todo == 0) return;

    // Back-to-front state per pixel: T = transmittance in front of the current splat, and ONE scalar
    // D = sum over the splats behind of (c_k . dL/dC) alpha_k T_k + T_final (bg . dL/dC): because the upstream pixel
    // gradient is constant along the list, dL/dalpha_j = (c_j . dL/dC) T_j - D / (1 - alpha_j) needs no per-channel state.
    float T = T_final;
    float D = T_final * bg_dot;
    const int nbatch = (int)((todo + RB - 1) / RB);
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * RB;
        const int cnt = min(RB, (int)todo - base);
        __syncthreads();
        stage_batch(L, sorted_splat, range.x + base, cnt, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE));
        __syncthreads();
#pragma unroll 1
        for (int lw = RB / 64 - 1; lw >= 0; --lw) {
            const int lim = wave_last - base - lw * 64;          // list positions >= wave_last never reach this wave
            if (lim <= 0) continue;
            uint64_t m = uniform_u64(L.qmask[lw][wave]);
            if (lim < 64) m &= (1ull << lim) - 1ull;
            while (m) {
                const int bit = 63 - __builtin_clzll(m);
                m &= ~(1ull << bit);
                const int j = lw * 64 + bit;
                const uint32_t k = (uint32_t)(base + j);     // 0-based list position; contributor index k+1
                const float4 xy = L.xyc[j];
                const float2 zo2 = *reinterpret_cast<const float2*>(&L.zoir[j]);
                const float4 cs = make_float4(xy.z, xy.w, zo2.x, zo2.y);
                const float dx = xy.x - pxf, dy = xy.y - pyf;
                const float p2 = __builtin_fmaf(cs.z * dy, dy, __builtin_fmaf(cs.y, dy, cs.x * dx) * dx);   // same expression as the forward
                const float G = __builtin_amdgcn_exp2f(p2);
                const float oa = cs.w * G;
                const float alpha = fminf(DVS_ALPHA_MAX, oa);
                const bool contrib = (k < last) && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
                if (!__any(contrib)) continue;
                // predicated: a non-contributing lane runs with alpha = 0 (T, D unchanged, every partial exactly 0)
                const float4 cg = L.cog[j];
                const float2 ir = *(reinterpret_cast<const float2*>(&L.zoir[j]) + 1);       // splat id bits (publisher lanes), colour r
                const float3 c = make_float3(ir.y, cg.w, L.bl[j].x);
                const float4 co = make_float4(cg.x, cg.y, cg.z, cs.w);
                const float al = contrib ? alpha : 0.f;
                const float inv_1ma = __builtin_amdgcn_rcpf(1.f - al);        // 1-alpha >= 0.01: v_rcp_f32 (1 ulp) is ample
                T = T * inv_1ma;
                const float w = al * T;
                const float cd = (c.x * dLp0 + c.y * dLp1) + c.z * dLp2;
                float dL_dalpha = cd * T - D * inv_1ma;
                D = D + cd * w;
                dL_dalpha = (contrib && !(oa > DVS_ALPHA_MAX)) ? dL_dalpha : 0.f;   // the 0.99 clamp blocks the gradient
                // Per-splat sums are published as MOMENTS of the weight s = dL/dG * G about the splat's mean:
                //   S_x = sum s dx, S_y = sum s dy, S_xx = sum s dx^2, S_xy = sum s dx dy, S_yy = sum s dy^2, S_o = sum G dL/dalpha
                // k_preprocess_bwd turns them into dL/dmean2D = -(a S_x + b S_y, c S_y + b S_x) and dL/dconic = (-S_xx/2, -S_xy, -S_yy/2):
                // the conic multiplications happen once per splat instead of once per (pixel, splat) pair (7 VALU fewer per visit).
                const float v5 = G * dL_dalpha;
                const float sw = co.w * v5;
                const float su = sw * dx, st = sw * dy;
                float v[12];
                v[0] = su; v[1] = st;
                v[2] = su * dx; v[3] = su * dy; v[4] = st * dy;
                v[5] = v5;
                v[6] = w * dLp0; v[7] = w * dLp1; v[8] = w * dLp2;
                // abs-grad needs the per-pixel |dL/dmean2D| itself
                v[9] = ABSGRAD ? fabsf(__builtin_fmaf(co.x, su, co.y * st)) : 0.f;
                v[10] = ABSGRAD ? fabsf(__builtin_fmaf(co.z, st, co.y * su)) : 0.f;
                v[11] = 0.f;
                float q[3];
                wave_reduce12<ABSGRAD ? 11 : 9>(v, q, xaddr);
                if (publisher) {
                    // 11 lanes, 11 consecutive floats of the splat's row: one global_atomic_add_f32 instruction
                    const float val = lcol == 0 ? q[0] : (lcol == 1 ? q[1] : q[2]);
                    atomicAdd(&grow[(size_t)__float_as_uint(ir.x) * 12 + kv], val);
                }
            }
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
hipError_t dvs_launch_render_fwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* splat2d, const float bg[3], float* out_color, float* final_T,
                                 uint32_t* n_contrib) {
    const int num_tiles = tiles_x * tiles_y;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges, sorted_splat,
                       (const float4*)splat2d, bg[0], bg[1], bg[2], out_color, final_T,
                       n_contrib);
    return hipGetLastError();
}

hipError_t dvs_launch_render_bwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* splat2d, const float bg[3], const float* final_T, const uint32_t* n_contrib,
                                 const float* dL_dout, float* grad_rows, int absgrad) {
    const int num_tiles = tiles_x * tiles_y;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    if (absgrad)
        hipLaunchKernelGGL(k_render_bwd<true>, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges,
                           sorted_splat, (const float4*)splat2d, bg[0], bg[1], bg[2], final_T,
                           n_contrib, dL_dout, grad_rows);
    else
        hipLaunchKernelGGL(k_render_bwd<false>, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges,
                           sorted_splat, (const float4*)splat2d, bg[0], bg[1], bg[2], final_T,
                           n_contrib, dL_dout, grad_rows);
    return hipGetLastError();
}
