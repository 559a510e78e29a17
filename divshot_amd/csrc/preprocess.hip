// preprocess.hip — A2 (project / preprocess forward) and A9 (preprocess backward) for gfx950.
// Built with -ffp-contract=off: every expression that feeds an integer decision (radius, tile rect,
// sort key) is evaluated in a fixed IEEE order so bins are bit-exact against the CPU oracle.
//
// Reference anchors (fenghuayumo/DIVSHOT; the trainer itself is closed source):
//   EWA projection gsplat_vs.hlsl:74-110 · cov3D gsplat_vs.hlsl:171-209 · low-pass/eigen gsplat_vs.hlsl:304-311 ·
//   AA factor gsplat_vs.hlsl:296-301 · ndc2Pix gsplat_vs.hlsl:211-214 · opacity cut gsplat_vs.hlsl:269 ·
//   SH gsplat_sh.hlsl:42-124 · activations gaussian_model.cpp:137-159.
//
// Both kernels are HBM-streaming (236 B of parameters per splat, ~400 flop): one lane per splat.
// The 45-float shN row (76 % of the bytes) is moved through LDS so that global traffic is
// 16-B-per-lane coalesced while each lane still consumes / produces its own row.
#include <cstdlib>
#include "dvs_device.h"
#include "dvs_kernels.h"

#ifndef PP_BLOCK
#define PP_BLOCK 128
#endif

// ---- cooperative row staging: rows of RW floats per splat, block of PP_BLOCK splats ---------------
// global [n, RW] -> lds[PP_BLOCK * RW]; the row of lane t starts at lds + t*RW (RW odd => conflict-free).
template <int RW>
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ g, float* lds, int64_t base_splat, int n) {
    const int64_t first = base_splat * RW;
    const int64_t total = (int64_t)n * RW;
    int64_t cnt = (int64_t)PP_BLOCK * RW;
    if (first + cnt > total) cnt = total - first;
    // 16-byte vector path when the block's first float is 16-B aligned (base_splat multiple of 4 and RW*4 bytes...)
    const float* src = g + first;
    if (((uintptr_t)src & 15) == 0) {
        const int nvec = (int)(cnt >> 2);
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(lds);
        for (int v = threadIdx.x; v < nvec; v += PP_BLOCK) d4[v] = s4[v];
        for (int e = (nvec << 2) + threadIdx.x; e < cnt; e += PP_BLOCK) lds[e] = src[e];
    } else {
        for (int e = threadIdx.x; e < cnt; e += PP_BLOCK) lds[e] = src[e];
    }
}
template <int RW, bool ACCUM>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ g, const float* lds, int64_t base_splat, int n) {
    const int64_t first = base_splat * RW;
    const int64_t total = (int64_t)n * RW;
    int64_t cnt = (int64_t)PP_BLOCK * RW;
    if (first + cnt > total) cnt = total - first;
    float* dst = g + first;
    if (((uintptr_t)dst & 15) == 0) {
        const int nvec = (int)(cnt >> 2);
        float4* d4 = reinterpret_cast<float4*>(dst);
        const float4* s4 = reinterpret_cast<const float4*>(lds);
        for (int v = threadIdx.x; v < nvec; v += PP_BLOCK) {
            float4 x = s4[v];
            if (ACCUM) { const float4 o = d4[v]; x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w; }
            d4[v] = x;
        }
        for (int e = (nvec << 2) + threadIdx.x; e < cnt; e += PP_BLOCK) dst[e] = ACCUM ? dst[e] + lds[e] : lds[e];
    } else {
        for (int e = threadIdx.x; e < cnt; e += PP_BLOCK) dst[e] = ACCUM ? dst[e] + lds[e] : lds[e];
    }
}

// shN layouts. ROWS: [n][45] (the reference's hand-off layout, gaussian_model.cpp:163-167) — rows are moved through LDS.
// TILED: [ceil(n/64)][12][64][4] — a splat's 45 floats padded to 48 = twelve float4 chunks; chunk c of the 64 splats of a
// tile is one contiguous 1-KiB run, so a wave reads/writes it with ONE 16-B-per-lane instruction, no LDS round trip
// (and no LDS-bound occupancy). Element e of splat i: float4 index ((i>>6)*12 + e/4)*64 + (i&63), component e%4.
__device__ __forceinline__ int64_t shn_tiled_f4(int i, int c) { return ((int64_t)(i >> 6) * 12 + c) * 64 + (i & 63); }

// ---- A2 ---------------------------------------------------------------------------------------
template <bool TILED, bool MULTI /*more than one view: the parameter registers stay live across the view loop*/>
__global__ void __launch_bounds__(PP_BLOCK)
k_preprocess_fwd(DvsCams cams_arg /* MUST stay the first parameter: read through dvs_load_cam() */, int n_views, int n,
                 const float* __restrict__ pos, const float* __restrict__ sh0, const float* __restrict__ shN,
                 const float* __restrict__ opacity, const float* __restrict__ scale, const float* __restrict__ rot,
                 int deg, int antialias, int tiles_x, int tiles_y,
                 int* __restrict__ radii, float4* __restrict__ splat2d /*[n] 64-B records, DVS_S2D_* */, float* __restrict__ depth,
                 uint32_t* __restrict__ flags,
                 uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_key, uint32_t* __restrict__ ids,
                 uint2* __restrict__ rect /*tile rectangle [minx | maxx << 16, miny | maxy << 16]; empty for culled splats*/,
                 uint4* __restrict__ rect16 /*DVS_TILES_TIGHT: {rectangle, tile mask lo, hi} instead of `rect` (null: canonical rectangles)*/,
                 uint32_t* __restrict__ rect8 /*DVS_FE_RECT_U8: minx | miny << 8 | width << 16 | height << 24 instead of `rect` (null: 16-bit fields)*/,
                 uint32_t* __restrict__ kred /*segmented front end (frontend.hip): [view][64 slots][16 words], word 0 = max(~key), word 1 = max(key)
                 over the view's visible splats, zeroed by the caller; null = the batch-wide sort of rounds 1-4, which wants `ids`*/,
                 int i0, int i1 /*this launch covers the splats [i0, i1): all of them, or one chunk of a data-parallel step that projects the
                 next iteration's splats chunk by chunk behind the optimizer (dvs_raster_forward_views_prepare); n stays the view-major stride*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PP_BLOCK*45] when deg>0
    const int64_t base = (int64_t)i0 + (int64_t)blockIdx.x * PP_BLOCK;
    const int i = (int)(base + threadIdx.x);
    // issue this lane's own parameter loads first so they are in flight together with the cooperative shN staging
    const int il = i < i1 ? i : (i1 - 1);
    const float in_px = pos[3 * (int64_t)il], in_py = pos[3 * (int64_t)il + 1], in_pz = pos[3 * (int64_t)il + 2];
    const float in_s0 = scale[3 * (int64_t)il], in_s1 = scale[3 * (int64_t)il + 1], in_s2 = scale[3 * (int64_t)il + 2];
    const float4 in_q = reinterpret_cast<const float4*>(rot)[il];
    const float in_op = opacity[il];
    const float in_dc0 = sh0[3 * (int64_t)il], in_dc1 = sh0[3 * (int64_t)il + 1], in_dc2 = sh0[3 * (int64_t)il + 2];
    float4 in_q4[12];                         // TILED: the splat's twelve float4 chunks, requested up front with everything else
    if (TILED) {
        // One straight run of loads per SH degree (12 / 6 / 3 / 0 chunks), nothing conditional between them. (Through round 5 every chunk had
        // its own `c < nchunk` test: the compiler turned that into a branch per chunk with a register copy — and an s_waitcnt vmcnt(0) — behind
        // the second one, i.e. every wave waited for its first seven loads before it requested the other ten chunks. Removing that did NOT
        // change the kernel's time — 85 us for one view at C3 either way, same-box A/B — so the dependent round trip was not what bounds it;
        // the straight form stays because it is the simpler code.)
        const float4* t4 = reinterpret_cast<const float4*>(shN);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 12; ++c) in_q4[c] = z4;
        if (deg >= 3) {
#pragma unroll
            for (int c = 0; c < 12; ++c) in_q4[c] = t4[shn_tiled_f4(il, c)];
        } else if (deg == 2) {
#pragma unroll
            for (int c = 0; c < 6; ++c) in_q4[c] = t4[shn_tiled_f4(il, c)];
        } else if (deg == 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) in_q4[c] = t4[shn_tiled_f4(il, c)];
        }
    }
    if (!TILED && deg > 0) {
        stage_rows_in<45>(shN, lds, base, i1);
        __syncthreads();
    }
    if (i >= i1) return;
    (void)cams_arg;
    // One lane per splat, all views of the batch: the 236 B of parameters are read once per iteration instead of once per view;
    // the per-view outputs are view-major ([view][splat]: index o), the sort value is the global index.
    for (int view = 0; view < (MULTI ? n_views : 1); ++view) {
    const DvsCam cam = dvs_load_cam(view);
    const int64_t o = (int64_t)view * n + i;

    // Straight-line and predicated: every cull test only clears `ok`, and the outputs are selected once at the end. (Rounds 1-5b left the
    // body through seven early exits; each exit level re-materialised the seventeen zero defaults of the outputs and kept a saved EXEC mask
    // alive — 190 v_mov and 78 SGPR-spill v_readlane per view in a kernel that is 80 % vector-ALU-busy, profiles/r05b_pmc_sq.txt — while a
    // wave practically never has all 64 splats culled, so the exits saved nothing.) Same expressions in the same order for the lanes that
    // survive: bit-identical outputs; culled lanes compute on whatever they hold (NaN / inf are harmless, nothing is stored from them).
    const float px = in_px, py = in_py, pz = in_pz;
    const float tx = dvs_xform(cam.view, px, py, pz, 0);
    const float ty = dvs_xform(cam.view, px, py, pz, 1);
    const float tz = dvs_xform(cam.view, px, py, pz, 2);
    // a NaN log-scale or opacity logit culls the splat (the clamps inside dvs_exp_det would otherwise turn it into a number)
    bool ok = (tz > DVS_NEAR) && (in_s0 == in_s0) && (in_s1 == in_s1) && (in_s2 == in_s2) && (in_op == in_op);
    const float hx = dvs_xform(cam.proj, px, py, pz, 0);
    const float hy = dvs_xform(cam.proj, px, py, pz, 1);
    const float hw = dvs_xform(cam.proj, px, py, pz, 3);
    const float pw = 1.0f / (hw + 0.0000001f);
    const float ndc_x = hx * pw, ndc_y = hy * pw;

    const float s[3] = {dvs_exp_det(in_s0), dvs_exp_det(in_s1), dvs_exp_det(in_s2)};
    const float4 q4 = in_q;
    const float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
    const float qn = dvs_sqrt_rn(((qr * qr + qx * qx) + qy * qy) + qz * qz);
    ok = ok && (qn > 0.f);
    const float inv_qn = 1.0f / qn;
    float R[9];
    dvs_quat_to_rot(qr * inv_qn, qx * inv_qn, qy * inv_qn, qz * inv_qn, R);
    float c3[6];
    dvs_cov3d(s, R, c3);

    const float limx = DVS_FOV_GUARD * cam.tan_fovx, limy = DVS_FOV_GUARD * cam.tan_fovy;
    const float txtz = tx / tz, tytz = ty / tz;
    uint32_t fl = 0;
    if (txtz < -limx || txtz > limx) fl |= DVS_FLAG_CLAMP_X;
    if (tytz < -limy || tytz > limy) fl |= DVS_FLAG_CLAMP_Y;
    const float txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
    const float tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float fx = cam.focal_x, fy = cam.focal_y;
    const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
    const float J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
    float T0[3], T1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T0[k] = J00 * cam.view[k * 4 + 0] + J02 * cam.view[k * 4 + 2];
        T1[k] = J11 * cam.view[k * 4 + 1] + J12 * cam.view[k * 4 + 2];
    }
    const float v0x = (c3[0] * T0[0] + c3[1] * T0[1]) + c3[2] * T0[2];
    const float v0y = (c3[1] * T0[0] + c3[3] * T0[1]) + c3[4] * T0[2];
    const float v0z = (c3[2] * T0[0] + c3[4] * T0[1]) + c3[5] * T0[2];
    const float v1x = (c3[0] * T1[0] + c3[1] * T1[1]) + c3[2] * T1[2];
    const float v1y = (c3[1] * T1[0] + c3[3] * T1[1]) + c3[4] * T1[2];
    const float v1z = (c3[2] * T1[0] + c3[4] * T1[1]) + c3[5] * T1[2];
    const float cxx = (T0[0] * v0x + T0[1] * v0y) + T0[2] * v0z;
    const float cxy = (T0[0] * v1x + T0[1] * v1y) + T0[2] * v1z;
    const float cyy = (T1[0] * v1x + T1[1] * v1y) + T1[2] * v1z;

    const float a = cxx + DVS_LOWPASS, b = cxy, c = cyy + DVS_LOWPASS;
    const float det = a * c - b * b;
    ok = ok && (det > 0.f);
    float opac = dvs_sigmoid_det(in_op);
    if (antialias) {
        const float det_orig = cxx * cyy - b * b;
        const float aa = dvs_sqrt_rn(fmaxf(0.f, det_orig / det));
        opac = opac * aa;
    }
    ok = ok && (opac > DVS_ALPHA_MIN);
    const float det_inv = 1.0f / det;
    const float mid = 0.5f * (a + c);
    const float lam = mid + dvs_sqrt_rn(fmaxf(0.1f, mid * mid - det));
    const float radf = ceilf(3.0f * dvs_sqrt_rn(lam));
    const float m2x = ((ndc_x + 1.0f) * (float)cam.width - 1.0f) * 0.5f;
    const float m2y = ((ndc_y + 1.0f) * (float)cam.height - 1.0f) * 0.5f;
    const float gx = (float)tiles_x, gy = (float)tiles_y, inv_tile = 1.0f / DVS_TILE;
    const int rminx = (int)fminf(gx, fmaxf(0.f, (m2x - radf) * inv_tile));
    const int rminy = (int)fminf(gy, fmaxf(0.f, (m2y - radf) * inv_tile));
    const int rmaxx = (int)fminf(gx, fmaxf(0.f, (m2x + radf + (float)(DVS_TILE - 1)) * inv_tile));
    const int rmaxy = (int)fminf(gy, fmaxf(0.f, (m2y + radf + (float)(DVS_TILE - 1)) * inv_tile));
    const int touched = (rmaxx - rminx) * (rmaxy - rminy);
    ok = ok && (touched > 0);

    const float dx = px - cam.campos[0], dy = py - cam.campos[1], dz = pz - cam.campos[2];
    const float dl = dvs_sqrt_rn((dx * dx + dy * dy) + dz * dz);
    const float inv_dl = 1.0f / dl;
    float bas[16];
    dvs_sh_basis(deg, dx * inv_dl, dy * inv_dl, dz * inv_dl, bas);
    const int ncoef = (deg + 1) * (deg + 1);
    const float in_dc[3] = {in_dc0, in_dc1, in_dc2};
    float colr[3] = {bas[0] * in_dc[0], bas[0] * in_dc[1], bas[0] * in_dc[2]};
    if (TILED) {
        const int nchunk = ((ncoef - 1) * 3 + 3) >> 2;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            if (c < nchunk) {
                const float4 q = in_q4[c];
                const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = c * 4 + u;                     // compile-time: coefficient e/3 + 1, channel e%3
                    if (e < 45 && e / 3 + 1 < ncoef) colr[e % 3] = colr[e % 3] + bas[e / 3 + 1] * qv[u];
                }
            }
        }
    } else {
        const float* row = lds + threadIdx.x * 45;
        for (int k = 1; k < ncoef; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) colr[ch] = colr[ch] + bas[k] * row[(k - 1) * 3 + ch];
    }
    float rgb_c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float col = colr[ch] + 0.5f;
        if (col < 0.f) { fl |= (1u << ch); col = 0.f; }
        rgb_c[ch] = col;
    }
    const int out_radius = ok ? (int)fminf(radf, (float)(1 << 30)) : 0;
    const float2 out_mean = ok ? make_float2(m2x, m2y) : make_float2(0.f, 0.f);
    const float out_depth = ok ? tz : 0.f;
    const uint32_t out_key = ok ? __float_as_uint(tz) : 0xFFFFFFFFu;
    const float4 out_co = ok ? make_float4(c * det_inv, -b * det_inv, a * det_inv, opac) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float out_rgb[3] = {ok ? rgb_c[0] : 0.f, ok ? rgb_c[1] : 0.f, ok ? rgb_c[2] : 0.f};
    const uint32_t out_flags = ok ? fl : 0u;
    uint32_t out_tiles = ok ? (uint32_t)touched : 0u;
    const uint2 out_rect = ok ? make_uint2((uint32_t)rminx | ((uint32_t)rmaxx << 16), (uint32_t)rminy | ((uint32_t)rmaxy << 16)) : make_uint2(0u, 0u);
    unsigned long long out_mask = 0ull;
    if (rect16 && ok) {    // DVS_TILES_TIGHT: only the tiles the alpha >= 1/255 ellipse reaches (rectangles of more than 64 tiles stay whole)
        out_mask = ~0ull;
        if (touched <= 64) {
            out_mask = dvs_tight_tile_mask(out_co.x, out_co.y, out_co.z, opac, m2x, m2y, rminx, rminy, rmaxx, rmaxy);
            out_tiles = (uint32_t)__popcll(out_mask);
        }
    }

    radii[o] = out_radius;
    depth[o] = out_depth;
    // The projected splat as one 64-B record: the tile kernels gather it per instance, one cache line instead of three.
    // Whole lines are written (a partly written line costs a read-modify-write), and a full wave transposes its 64 records
    // through LDS so that each store instruction covers 1 KB of consecutive addresses instead of 64 quarter lines.
    const float4 rc0 = make_float4(out_mean.x, out_mean.y, out_co.x, out_co.y);
    const float4 rc1 = make_float4(out_co.z, out_co.w, out_rgb[0], out_rgb[1]);
    // DVS_S2D_CULL: the constants of the composite kernels' ellipse-vs-rectangle tests (render.hip stage_batch), which depend on the
    // splat alone — computed here once per splat and view instead of once per instance in A7 and four times per instance in A8
    float cull_bound = 0.f, cull_det_c = 0.f, cull_det_a = 0.f, cull_nb_c = 0.f, cull_nb_a = 0.f;
    if (out_radius > 0) {
        const float a = out_co.x, b = out_co.y, c = out_co.z;
        cull_bound = 1.3862943611f * __builtin_amdgcn_logf(255.0f * out_co.w) * 1.0001f + 1e-3f;      // 2 ln(255 o), inflated
        const float det = fmaxf(0.f, __builtin_fmaf(-2.4e-7f, a * c, a * c - b * b));                  // lowered by its rounding bound
        const float rc = __builtin_amdgcn_rcpf(c), ra = __builtin_amdgcn_rcpf(a);
        cull_det_c = det * rc; cull_det_a = det * ra; cull_nb_c = -b * rc; cull_nb_a = -b * ra;
    }
    const float4 rc2 = make_float4(out_rgb[2], out_depth, __int_as_float(out_radius), cull_bound);
    const float4 rc3 = make_float4(cull_det_c, cull_det_a, cull_nb_c, cull_nb_a);
    const int wave_base = (int)base + (int)(threadIdx.x & ~63u);
    if (wave_base + 64 <= i1) {                // (a whole wave of valid splats: uniform per wave)
        __shared__ float4 s_rec[PP_BLOCK * 4];
        float4* w = s_rec + (threadIdx.x & ~63u) * 4;           // this wave's 4 KB; wave-local, LDS ops of a wave execute in order
        const int lane = threadIdx.x & 63;
        w[lane * 4 + 0] = rc0; w[lane * 4 + 1] = rc1; w[lane * 4 + 2] = rc2; w[lane * 4 + 3] = rc3;
        float4* dst = splat2d + 4 * ((size_t)view * n + (size_t)wave_base);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 64 + lane] = w[q * 64 + lane];
    } else {
        float4* rec = splat2d + 4 * (size_t)o;
        rec[0] = rc0; rec[1] = rc1; rec[2] = rc2; rec[3] = rc3;
    }
    flags[o] = out_flags;
    tiles_touched[o] = out_tiles;
    if (rect16) rect16[o] = make_uint4(out_rect.x, out_rect.y, (uint32_t)out_mask, (uint32_t)(out_mask >> 32));
    else if (rect8) {
        const uint32_t x0 = out_rect.x & 0xFFFFu, y0 = out_rect.y & 0xFFFFu;
        rect8[o] = x0 | (y0 << 8) | (((out_rect.x >> 16) - x0) << 16) | (((out_rect.y >> 16) - y0) << 24);
    } else rect[o] = out_rect;
    depth_key[o] = out_key;
    if (ids) ids[o] = (uint32_t)o;
    if (kred) {
        // the view's key range for the range-adaptive depth sort: one pair of atomics per wave into one of 64 slots (64 B apart)
        uint32_t knm = ~out_key, kmx = out_radius > 0 ? out_key : 0u;         // both are max reductions with identity 0 (culled: ~0xFFFFFFFF = 0)
        uint32_t* slot = kred + ((size_t)view * 64 + (blockIdx.x & 63u)) * 16;
        if (wave_base + 64 <= i1) {
            // wave64 max in six DPP steps (row_shr 1, 2, 4, 8, row_bcast 15, 31): lane 63 ends with the wave's maximum
#define A2_DPP_MAX(v, ctrl, rmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false); v = o_ > v ? o_ : v; }
            A2_DPP_MAX(knm, 0x111, 0xF) A2_DPP_MAX(kmx, 0x111, 0xF) A2_DPP_MAX(knm, 0x112, 0xF) A2_DPP_MAX(kmx, 0x112, 0xF)
            A2_DPP_MAX(knm, 0x114, 0xF) A2_DPP_MAX(kmx, 0x114, 0xF) A2_DPP_MAX(knm, 0x118, 0xF) A2_DPP_MAX(kmx, 0x118, 0xF)
            A2_DPP_MAX(knm, 0x142, 0xA) A2_DPP_MAX(kmx, 0x142, 0xA) A2_DPP_MAX(knm, 0x143, 0xC) A2_DPP_MAX(kmx, 0x143, 0xC)
#undef A2_DPP_MAX
            if ((threadIdx.x & 63) == 63 && knm != 0u) { atomicMax(slot, knm); atomicMax(slot + 1, kmx); }
        } else if (out_radius > 0) { atomicMax(slot, knm); atomicMax(slot + 1, kmx); }
    }
    }   // views
}

// The geometry part of A9. Round 6 splits it where the chain stops depending on the view:
//   a9_geometry   per (splat, view): recomputes the forward intermediates (same expressions as k_preprocess_fwd: these feed nothing but
//                 floats here, but the conic must be the one the forward composited with), turns the A8 moments into dL/dmean2D and
//                 dL/dconic, walks back through conic -> cov2D -> (J, W) and through the projection to the position, and ADDS the
//                 view's symmetrised dL/dSigma (Gm + Gm^T, six values) to `Gs`;
//   a9_sigma_to_params   once per splat: Sigma = M M^T, M = R S  ->  dL/dscale, dL/drot. The 3D covariance does not depend on the view
//                 and this step is linear in dL/dSigma, so a multi-view pass runs it ONCE on the sum over its views instead of once
//                 per view (27 + 9 + 32 multiply-adds and the quaternion normalisation leave the view loop; k_preprocess_bwd_views is
//                 bound by its vector instructions once its loads are pipelined, profiles/r06_a9_ab.txt).
// The backward-only blocks allow FMA contraction (this file is compiled with -ffp-contract=off for A2's bit-exact bins; nothing below
// the conic feeds an integer decision). `gp` arrives holding the SH view-direction part and leaves holding the complete dL/dpos.
__device__ __forceinline__ void a9_geometry(const DvsCam& cam, float px, float py, float pz, float in_s0, float in_s1, float in_s2, float4 in_q,
                                            float in_op, uint32_t fl, float4 r0, float4 r1, int antialias, int grad_mode, float gp[3],
                                            float Gs[6] /* += : 00 01 02 11 12 22 of Gm + Gm^T */, float& g_op, float2& dm_out) {
        const float tx = dvs_xform(cam.view, px, py, pz, 0);
        const float ty = dvs_xform(cam.view, px, py, pz, 1);
        const float tz = dvs_xform(cam.view, px, py, pz, 2);
        const float hx = dvs_xform(cam.proj, px, py, pz, 0);
        const float hy = dvs_xform(cam.proj, px, py, pz, 1);
        const float hw = dvs_xform(cam.proj, px, py, pz, 3);
        const float pw = 1.0f / (hw + 0.0000001f);
        const float s[3] = {dvs_exp_det(in_s0), dvs_exp_det(in_s1), dvs_exp_det(in_s2)};
        const float4 q4 = in_q;
        const float qn = dvs_sqrt_rn(((q4.x * q4.x + q4.y * q4.y) + q4.z * q4.z) + q4.w * q4.w);
        const float inv_qn = 1.0f / qn;
        const float qr = q4.x * inv_qn, qx = q4.y * inv_qn, qy = q4.z * inv_qn, qz = q4.w * inv_qn;
        float R[9];
        dvs_quat_to_rot(qr, qx, qy, qz, R);
        float c3[6];
        dvs_cov3d(s, R, c3);
        const float limx = DVS_FOV_GUARD * cam.tan_fovx, limy = DVS_FOV_GUARD * cam.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        const float cl_x = fminf(limx, fmaxf(-limx, txtz)), cl_y = fminf(limy, fmaxf(-limy, tytz));
        const float txc = cl_x * tz, tyc = cl_y * tz;
        const float fx = cam.focal_x, fy = cam.focal_y;
        const float J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
        const float J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
        float T0[3], T1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T0[k] = J00 * cam.view[k * 4 + 0] + J02 * cam.view[k * 4 + 2];
            T1[k] = J11 * cam.view[k * 4 + 1] + J12 * cam.view[k * 4 + 2];
        }
        const float v0[3] = {(c3[0] * T0[0] + c3[1] * T0[1]) + c3[2] * T0[2], (c3[1] * T0[0] + c3[3] * T0[1]) + c3[4] * T0[2],
                             (c3[2] * T0[0] + c3[4] * T0[1]) + c3[5] * T0[2]};
        const float v1[3] = {(c3[0] * T1[0] + c3[1] * T1[1]) + c3[2] * T1[2], (c3[1] * T1[0] + c3[3] * T1[1]) + c3[4] * T1[2],
                             (c3[2] * T1[0] + c3[4] * T1[1]) + c3[5] * T1[2]};
        const float cxx = (T0[0] * v0[0] + T0[1] * v0[1]) + T0[2] * v0[2];
        const float cxy = (T0[0] * v1[0] + T0[1] * v1[1]) + T0[2] * v1[2];
        const float cyy = (T1[0] * v1[0] + T1[1] * v1[1]) + T1[2] * v1[2];
        const float a = cxx + DVS_LOWPASS, b = cxy, c = cyy + DVS_LOWPASS;
        const float det = a * c - b * b;
        const float det_inv = 1.0f / det;
        // A8 publishes moments of s = dL/dG * G about the mean (S_x S_y | S_xx S_xy S_yy | S_o); with the conic (A, B, C) — the same
        // expressions, hence the same bits, as the forward wrote — they become the gradients of the 2D mean and of the conic
        const float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
        const float sig = dvs_sigmoid_det(in_op);
        (void)ty;
    {
#pragma clang fp contract(fast)
        const float2 dL_dm = make_float2(-(cA * r0.x + cB * r0.y), -(cC * r0.y + cB * r0.x));
        const float4 gco = make_float4(-0.5f * r0.z, -r0.w, -0.5f * r1.x, r1.y);
        dm_out = dL_dm;

        // 2. opacity (+ AA)
        float g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f;
        float g_sig = gco.w;
        if (antialias) {
            const float det_orig = cxx * cyy - b * b;
            const float ratio = det_orig / det;
            const float aa = dvs_sqrt_rn(fmaxf(0.f, ratio));
            g_sig = gco.w * aa;
            if (ratio > 0.f) {
                const float g_aa = gco.w * sig;
                const float g_ratio = g_aa * 0.5f / aa;
                const float g_do = g_ratio * det_inv;
                const float g_db = -g_ratio * det_orig * det_inv * det_inv;
                g_cxx += g_do * cyy + g_db * c;
                g_cyy += g_do * cxx + g_db * a;
                g_cxy += -2.f * b * (g_do + g_db);
            }
        }
        g_op = g_sig * sig * (1.f - sig);

        // 3. conic
        {
            const float Ssum = (gco.x * c - gco.y * b) + gco.z * a;
            const float g_det = -Ssum * det_inv * det_inv;
            g_cxx += gco.z * det_inv + g_det * c;
            g_cyy += gco.x * det_inv + g_det * a;
            g_cxy += -gco.y * det_inv + g_det * (-2.f * b);
        }
        // 4. cov2D = T Sigma T^T: dL/dSigma = Gm with Gm[r][q] = g_cxx T0r T0q + g_cxy T0r T1q + g_cyy T1r T1q; only its symmetric part
        //    acts on Sigma = M M^T:  (Gm + Gm^T)[r][q] = T0q U_r + T1q W_r  with  U = 2 g_cxx T0 + g_cxy T1,  W = g_cxy T0 + 2 g_cyy T1
        {
            const float a2 = 2.f * g_cxx, c2 = 2.f * g_cyy;
            float U[3], Wv[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { U[r] = a2 * T0[r] + g_cxy * T1[r]; Wv[r] = g_cxy * T0[r] + c2 * T1[r]; }
            Gs[0] += T0[0] * U[0] + T1[0] * Wv[0];
            Gs[1] += T0[1] * U[0] + T1[1] * Wv[0];
            Gs[2] += T0[2] * U[0] + T1[2] * Wv[0];
            Gs[3] += T0[1] * U[1] + T1[1] * Wv[1];
            Gs[4] += T0[2] * U[1] + T1[2] * Wv[1];
            Gs[5] += T0[2] * U[2] + T1[2] * Wv[2];
        }
        float gT0[3], gT1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gT0[k] = 2.f * g_cxx * v0[k] + g_cxy * v1[k];
            gT1[k] = 2.f * g_cyy * v1[k] + g_cxy * v0[k];
        }
        // 5. Tm = J Wv
        float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gJ00 += gT0[k] * cam.view[k * 4 + 0]; gJ02 += gT0[k] * cam.view[k * 4 + 2];
            gJ11 += gT1[k] * cam.view[k * 4 + 1]; gJ12 += gT1[k] * cam.view[k * 4 + 2];
        }
        const float tz2 = 1.0f / (tz * tz), tz3 = tz2 / tz;
        float g_tx = 0.f, g_ty = 0.f, g_tz = 0.f;
        g_tz += -fx * tz2 * gJ00 - fy * tz2 * gJ11;
        g_tz += 2.f * fx * txc * tz3 * gJ02 + 2.f * fy * tyc * tz3 * gJ12;
        const float g_txc = -fx * tz2 * gJ02, g_tyc = -fy * tz2 * gJ12;
        // clamped branch (txc = cl_x * tz): the true gradient goes through tz; the credited lineage (README.md:95) drops it
        const float clamp_w = grad_mode == 1 ? 0.f : 1.f;
        if (fl & DVS_FLAG_CLAMP_X) g_tz += clamp_w * (g_txc * cl_x); else g_tx += g_txc;
        if (fl & DVS_FLAG_CLAMP_Y) g_tz += clamp_w * (g_tyc * cl_y); else g_ty += g_tyc;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            gp[k] += (cam.view[k * 4 + 0] * g_tx + cam.view[k * 4 + 1] * g_ty) + cam.view[k * 4 + 2] * g_tz;
        // 6. mean2D
        {
            const float2 gm = dL_dm;
            const float g_hx = gm.x * 0.5f * (float)cam.width * pw, g_hy = gm.y * 0.5f * (float)cam.height * pw;
            const float g_hw = -(gm.x * 0.5f * (float)cam.width * hx + gm.y * 0.5f * (float)cam.height * hy) * pw * pw;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                gp[k] += (cam.proj[k * 4 + 0] * g_hx + cam.proj[k * 4 + 1] * g_hy) + cam.proj[k * 4 + 3] * g_hw;
        }
    }
}

// 7. Sigma = M M^T with M = R S: dL/dM = (Gm + Gm^T) M, then scale and rotation (the quaternion is normalised in the forward, so its
// gradient is projected off the quaternion and divided by its norm). `Gs` = the symmetrised dL/dSigma summed over the views.
__device__ __forceinline__ void a9_sigma_to_params(float in_s0, float in_s1, float in_s2, float4 in_q, const float Gs[6], float gsc[3],
                                                   float gq_out[4]) {
    const float s[3] = {dvs_exp_det(in_s0), dvs_exp_det(in_s1), dvs_exp_det(in_s2)};
    const float qn = dvs_sqrt_rn(((in_q.x * in_q.x + in_q.y * in_q.y) + in_q.z * in_q.z) + in_q.w * in_q.w);
    const float inv_qn = 1.0f / qn;
    const float qr = in_q.x * inv_qn, qx = in_q.y * inv_qn, qy = in_q.z * inv_qn, qz = in_q.w * inv_qn;
    float R[9];
    dvs_quat_to_rot(qr, qx, qy, qz, R);
    {
#pragma clang fp contract(fast)
        const float G[9] = {Gs[0], Gs[1], Gs[2], Gs[1], Gs[3], Gs[4], Gs[2], Gs[4], Gs[5]};
        float M[9], gM[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) M[r * 3 + k] = R[r * 3 + k] * s[k];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 3; ++q) acc += G[r * 3 + q] * M[q * 3 + k];
                gM[r * 3 + k] = acc;
            }
        float gR[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float gs = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) { gs += gM[r * 3 + k] * R[r * 3 + k]; gR[r * 3 + k] = gM[r * 3 + k] * s[k]; }
            gsc[k] = gs * s[k];
        }
        float gq[4];
        gq[0] = 2.f * (-qz * gR[1] + qy * gR[2] + qz * gR[3] - qx * gR[5] - qy * gR[6] + qx * gR[7]);
        gq[1] = 2.f * (qy * gR[1] + qz * gR[2] + qy * gR[3] - 2.f * qx * gR[4] - qr * gR[5] + qz * gR[6] + qr * gR[7] - 2.f * qx * gR[8]);
        gq[2] = 2.f * (-2.f * qy * gR[0] + qx * gR[1] + qr * gR[2] + qx * gR[3] + qz * gR[5] - qr * gR[6] + qz * gR[7] - 2.f * qy * gR[8]);
        gq[3] = 2.f * (-2.f * qz * gR[0] - qr * gR[1] + qx * gR[2] + qr * gR[3] - 2.f * qz * gR[4] + qy * gR[5] + qx * gR[6] + qy * gR[7]);
        const float qg = ((qr * gq[0] + qx * gq[1]) + qy * gq[2]) + qz * gq[3];
        gq_out[0] = (gq[0] - qr * qg) * inv_qn; gq_out[1] = (gq[1] - qx * qg) * inv_qn;
        gq_out[2] = (gq[2] - qy * qg) * inv_qn; gq_out[3] = (gq[3] - qz * qg) * inv_qn;
    }
}

// ---- A9 ---------------------------------------------------------------------------------------
// dL/d(unit direction) of a view's colour: gdir[d] = sum_k d(basis_k)/d(dir_d) * s_k with s_k = sum_ch shN[k][ch] * gc[ch] (the colour
// gradient folded into the coefficients FIRST: 45 + 33 fused multiply-adds per (view, splat) instead of the 45 x 3 x 3 multiply-multiply-add
// triples of rounds 1-5a — k_preprocess_bwd_views is 60 % vector-ALU-busy by the counters, profiles/r05b_pmc_sq.txt). Only the structurally
// non-zero derivative entries are read (dvs_sh_basis_grad, dvs_device.h:211). Both A9 kernels use this one form, so a batch stays
// bit-identical to its views run one by one.
__device__ __forceinline__ void a9_dir_grad(int deg, float x, float y, float z, const float s[16], float g[3]) {
    float db[16][3];
    dvs_sh_basis_grad(deg, x, y, z, db);
#define A9_T(k, d) g[d] = __builtin_fmaf(db[k][d], s[k], g[d])
    if (deg >= 1) { A9_T(1, 1); A9_T(2, 2); A9_T(3, 0); }
    if (deg >= 2) {
        A9_T(4, 0); A9_T(4, 1); A9_T(5, 1); A9_T(5, 2); A9_T(6, 0); A9_T(6, 1); A9_T(6, 2); A9_T(7, 0); A9_T(7, 2); A9_T(8, 0); A9_T(8, 1);
    }
    if (deg >= 3) {
        A9_T(9, 0); A9_T(9, 1); A9_T(10, 0); A9_T(10, 1); A9_T(10, 2); A9_T(11, 0); A9_T(11, 1); A9_T(11, 2); A9_T(12, 0); A9_T(12, 1); A9_T(12, 2);
        A9_T(13, 0); A9_T(13, 1); A9_T(13, 2); A9_T(14, 0); A9_T(14, 1); A9_T(14, 2); A9_T(15, 0); A9_T(15, 1);
    }
#undef A9_T
}

template <bool ACCUM, bool TILED>
__global__ void __launch_bounds__(PP_BLOCK)
k_preprocess_bwd(int n, const float* __restrict__ pos, const float* __restrict__ shN, const float* __restrict__ opacity,
                 const float* __restrict__ scale, const float* __restrict__ rot, DvsCam cam, int deg, int antialias,
                 const int* __restrict__ radii, const uint32_t* __restrict__ flags,
                 float4* __restrict__ grad_rows /*[n,3] float4: Sx Sy Sxx Sxy | Syy So r g | b |mx| |my| pad (A8 moments); re-zeroed here*/,
                 float* __restrict__ g_pos, float* __restrict__ g_sh0, float* __restrict__ g_shN,
                 float* __restrict__ g_opacity, float* __restrict__ g_scale, float* __restrict__ g_rot,
                 float2* __restrict__ out_absgrad2d, float2* __restrict__ out_mean2d, float* __restrict__ out_dcolor, int rezero,
                 int grad_mode /*dvs_opts.grad_mode: 1 = DVS_GRAD_LINEAGE holds the clamped Jacobian coordinate constant*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PP_BLOCK*45]
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    const int i = (int)(base + threadIdx.x);
    const bool valid = i < n;
    // this lane's own loads go out before the cooperative shN staging so both are in flight together
    const int il = valid ? i : (n - 1);
    const int radius = valid ? radii[il] : 0;
    const float4 in_r0 = grad_rows[3 * (int64_t)il], in_r1 = grad_rows[3 * (int64_t)il + 1], in_r2 = grad_rows[3 * (int64_t)il + 2];
    const float in_px = pos[3 * (int64_t)il], in_py = pos[3 * (int64_t)il + 1], in_pz = pos[3 * (int64_t)il + 2];
    const float in_s0 = scale[3 * (int64_t)il], in_s1 = scale[3 * (int64_t)il + 1], in_s2 = scale[3 * (int64_t)il + 2];
    const float4 in_q = reinterpret_cast<const float4*>(rot)[il];
    const float in_op = opacity[il];
    const uint32_t in_fl = flags[il];
    if (!TILED && deg > 0) {
        stage_rows_in<45>(shN, lds, base, n);
        __syncthreads();
    }
    float gp[3] = {0.f, 0.f, 0.f}, gs0[3] = {0.f, 0.f, 0.f}, gsc[3] = {0.f, 0.f, 0.f}, gq_out[4] = {0.f, 0.f, 0.f, 0.f};
    float gcol[3] = {0.f, 0.f, 0.f};     // dL/d(colour), zeroed where the colour was clamped: all a peer needs to rebuild the SH rows
    float g_op = 0.f;
    // ROWS: the staged parameter row is overwritten in place by its gradient and leaves through LDS.
    // TILED: parameters are read from, and gradients written to, the tiled arrays directly.
    float* row = lds + threadIdx.x * 45;
    const float4* p4 = reinterpret_cast<const float4*>(shN);
    float4* g4 = reinterpret_cast<float4*>(g_shN);

    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    float2 dm_out = make_float2(0.f, 0.f);          // dL/dmean2D (optional output)
    if (radius > 0) {
        r0 = in_r0; r1 = in_r1; r2 = in_r2;
        // leave the accumulation row zeroed for the next backward (saves a 48 B/splat memset pass per view)
        if (rezero) {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            grad_rows[3 * (int64_t)i] = z4; grad_rows[3 * (int64_t)i + 1] = z4; grad_rows[3 * (int64_t)i + 2] = z4;
        }
        const float dL_dcol[3] = {r1.z, r1.w, r2.x};
        const float px = in_px, py = in_py, pz = in_pz;
        const uint32_t fl = in_fl;
        // 1. colour / SH
        const float dxw = px - cam.campos[0], dyw = py - cam.campos[1], dzw = pz - cam.campos[2];
        const float dl = dvs_sqrt_rn((dxw * dxw + dyw * dyw) + dzw * dzw);
        const float inv_dl = 1.0f / dl;
        const float ux = dxw * inv_dl, uy = dyw * inv_dl, uz = dzw * inv_dl;
        float bas[16];
        dvs_sh_basis(deg, ux, uy, uz, bas);
        const int ncoef = (deg + 1) * (deg + 1);
        float gdir[3] = {0.f, 0.f, 0.f};
        float sk[16];                                        // s_k = sum_ch shN[k][ch] * gc[ch] (a9_dir_grad)
#pragma unroll
        for (int k = 0; k < 16; ++k) sk[k] = 0.f;
        float gc[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            gc[ch] = (fl & (1u << ch)) ? 0.f : dL_dcol[ch];
            gcol[ch] = gc[ch];
            gs0[ch] = bas[0] * gc[ch];
        }
        if (TILED) {
            // twelve float4 chunks: load the parameters of chunk c, emit its four gradient elements, store — nothing is staged
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const int64_t idx = shn_tiled_f4(i, c);
                float gv[4] = {0.f, 0.f, 0.f, 0.f};
                if (c * 4 < (ncoef - 1) * 3) {
                    const float4 q = p4[idx];
                    const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = c * 4 + u;                     // compile-time: coefficient k = e/3 + 1, channel e%3
                        if (e < 45 && e / 3 + 1 < ncoef) {
                            const int k = e / 3 + 1, ch = e % 3;
                            gv[u] = bas[k] * gc[ch];
                            sk[k] = __builtin_fmaf(qv[u], gc[ch], sk[k]);
                        }
                    }
                }
                if (g4) {
                    float4 o = make_float4(gv[0], gv[1], gv[2], gv[3]);
                    if (ACCUM) { const float4 p = g4[idx]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                    g4[idx] = o;
                }
            }
        } else {
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                if (k < ncoef) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float coef = row[(k - 1) * 3 + ch];
                        row[(k - 1) * 3 + ch] = bas[k] * gc[ch];   // overwrite the staged parameter with its gradient
                        sk[k] = __builtin_fmaf(coef, gc[ch], sk[k]);
                    }
                }
            }
            for (int e = (ncoef - 1) * 3; e < 45; ++e) row[e] = 0.f;
        }
        {
            a9_dir_grad(deg, ux, uy, uz, sk, gdir);
            const float ug = (ux * gdir[0] + uy * gdir[1]) + uz * gdir[2];
            gp[0] += (gdir[0] - ux * ug) * inv_dl; gp[1] += (gdir[1] - uy * ug) * inv_dl; gp[2] += (gdir[2] - uz * ug) * inv_dl;
        }

        float Gs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        a9_geometry(cam, px, py, pz, in_s0, in_s1, in_s2, in_q, in_op, fl, r0, r1, antialias, grad_mode, gp, Gs, g_op, dm_out);
        a9_sigma_to_params(in_s0, in_s1, in_s2, in_q, Gs, gsc, gq_out);
    } else if (valid) {
        if (TILED) {
            if (g4 && !ACCUM) {
#pragma unroll
                for (int c = 0; c < 12; ++c) g4[shn_tiled_f4(i, c)] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int e = 0; e < 45; ++e) row[e] = 0.f;
        }
    }

    if (valid) {
        if (out_absgrad2d) {
            float2 a = make_float2(r2.y, r2.z);
            if (ACCUM) { const float2 o = out_absgrad2d[i]; a.x += o.x; a.y += o.y; }
            out_absgrad2d[i] = a;
        }
        if (out_mean2d) {
            float2 mm = dm_out;
            if (ACCUM) { const float2 o = out_mean2d[i]; mm.x += o.x; mm.y += o.y; }
            out_mean2d[i] = mm;
        }
        if (ACCUM) {
            g_opacity[i] += g_op;
            float4 o = reinterpret_cast<float4*>(g_rot)[i];
            o.x += gq_out[0]; o.y += gq_out[1]; o.z += gq_out[2]; o.w += gq_out[3];
            reinterpret_cast<float4*>(g_rot)[i] = o;
        } else {
            g_opacity[i] = g_op;
            reinterpret_cast<float4*>(g_rot)[i] = make_float4(gq_out[0], gq_out[1], gq_out[2], gq_out[3]);
        }
    }
    // shN gradient rows leave through LDS as coalesced 16-B stores (zero rows when deg == 0); skipped entirely in the
    // factorised multi-GPU mode (g_shN == nullptr: peers rebuild the rows from dcolor, dvs_sh_grad_combine)
    if (!TILED && g_shN) {
        if (deg == 0) {
            if (valid) for (int e = 0; e < 45; ++e) row[e] = 0.f;
        }
        __syncthreads();
        stage_rows_out<45, ACCUM>(g_shN, lds, base, n);
    }
    // the 3-float groups take the same route (a lane-strided 4-byte store would touch every line three times)
    __syncthreads();
    float* l_pos = lds, *l_sh0 = lds + PP_BLOCK * 3, *l_scl = lds + PP_BLOCK * 6, *l_col = lds + PP_BLOCK * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        l_pos[threadIdx.x * 3 + k] = gp[k]; l_sh0[threadIdx.x * 3 + k] = gs0[k]; l_scl[threadIdx.x * 3 + k] = gsc[k];
        l_col[threadIdx.x * 3 + k] = gcol[k];
    }
    __syncthreads();
    stage_rows_out<3, ACCUM>(g_pos, l_pos, base, n);
    if (g_sh0) stage_rows_out<3, ACCUM>(g_sh0, l_sh0, base, n);
    stage_rows_out<3, ACCUM>(g_scale, l_scl, base, n);
    if (out_dcolor) stage_rows_out<3, false>(out_dcolor, l_col, base, n);
}

// ---- A9 over the views of a batch (DVS_SHN_TILED layout) -----------------------------------------------------------------
// One lane per splat, all views of the iteration: the parameters are read once, every view adds its contribution to register
// accumulators, and the geometry gradients (pos, scale, rot, opacity: 44 B) are written once — instead of a read-modify-write of the
// gradient rows per view. The SH rows are rank-1 in the per-view colour gradient, so this kernel only emits that (dcolor, 12 B per
// splat and view) and k_sh_grad_combine builds sh0 / shN from it afterwards — on one GPU right away, in data-parallel runs after the
// all-gather of dcolor (the factorised exchange), with the same kernel. Per view the same expressions in the same order as
// k_preprocess_bwd, so the geometry gradients of a batch are bit-identical to its views run one by one with opts.accumulate.
template <bool ACCUM, bool NOHOIST, bool FUSE_SH /*build the SH rows here instead of emitting per-view colour gradients (below)*/>
__global__ void __launch_bounds__(PP_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_preprocess_bwd_views(DvsCams cams_arg /* MUST stay the first parameter: read through dvs_load_cam() */, int n_views, int n,
                       const float* __restrict__ pos, const float* __restrict__ shN, const float* __restrict__ opacity,
                       const float* __restrict__ scale, const float* __restrict__ rot, int deg, int antialias,
                       const int* __restrict__ radii /*[V,n]*/, const uint32_t* __restrict__ flags /*[V,n]*/,
                       float4* __restrict__ grad_rows /*[V,n,3]: A8 moments; re-zeroed here*/,
                       float* __restrict__ g_pos, float* __restrict__ g_opacity, float* __restrict__ g_scale, float* __restrict__ g_rot,
                       float2* __restrict__ out_absgrad2d, float2* __restrict__ out_mean2d, float* __restrict__ out_dcolor /*[V,n,3]; FUSE_SH: unused*/,
                       float* __restrict__ g_sh0 /*FUSE_SH: [n,3]*/, float* __restrict__ g_shN /*FUSE_SH: tiled rows*/,
                       int rezero, int grad_mode, int i0 /*this launch covers the splats [i0, i1): all of them, or one chunk of a*/,
                       int i1 /*data-parallel step that sends each chunk's gradients off while the next chunk computes*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PP_BLOCK*6] (+ FUSE_SH: [n_views][6][PP_BLOCK] colour gradient | unit direction)
    (void)cams_arg;
    float* const l_view = lds + PP_BLOCK * 6;
    const int64_t base = (int64_t)i0 + (int64_t)blockIdx.x * PP_BLOCK;
    const int i = (int)(base + threadIdx.x);
    const bool valid = i < i1;
    const int il = valid ? i : (i1 - 1);
    const float px = pos[3 * (int64_t)il], py = pos[3 * (int64_t)il + 1], pz = pos[3 * (int64_t)il + 2];
    const float in_s0 = scale[3 * (int64_t)il], in_s1 = scale[3 * (int64_t)il + 1], in_s2 = scale[3 * (int64_t)il + 2];
    const float4 in_q = reinterpret_cast<const float4*>(rot)[il];
    const float in_op = opacity[il];
    const float4* p4 = reinterpret_cast<const float4*>(shN);
    float gp[3] = {0.f, 0.f, 0.f}, gsc[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    float Gs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // dL/dSigma (symmetrised), summed over the views
    float g_op = 0.f;
    float2 ag = make_float2(0.f, 0.f), dm = make_float2(0.f, 0.f);
    // which views see the splat: all radii are requested up front, so that a view costs ONE dependent memory round trip (its rows
    // and flags) instead of two (radius, then rows)
    uint32_t vis = 0;
    for (int view = 0; view < n_views; ++view) vis |= (valid && radii[(int64_t)view * n + il] > 0) ? (1u << view) : 0u;
    // Round 6: the kernel was latency-bound (two waves per SIMD, every view one dependent HBM round trip plus twelve dependent L2 reads
    // of the SH coefficients; 57 % vector-ALU-busy at 4.2 TB/s, profiles/r05b_pmc_sq.txt). Two changes, neither touches an expression:
    //  (i) the 45 higher-order coefficients are read ONCE, before the view loop, and stay in 48 registers (the loop used to re-read
    //      them from L2 for every view: 1.26 GB of L2 traffic per 8-view step, and — vector memory returns in order — every wait for
    //      them was also a wait for anything requested earlier);
    //  (ii) the view loop is software-pipelined: view v+1's three 16-B rows and flags are requested before view v's arithmetic,
    //      so the round trip runs under ~850 vector instructions instead of in front of them.
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = z4, n1 = z4, n2 = z4;
    uint32_t nfl = 0;
    {   // (unconditional loads: a lane without work reads element 0 — one shared line — so that no branch, and with it no register
        // copy behind the load, i.e. no wait, sits between a request and its use one iteration later.
        // The first view's rows are requested whether or not the splat is visible there: they then do not wait for the radii.)
        const int64_t o = (int64_t)il;
        n0 = grad_rows[3 * o]; n1 = grad_rows[3 * o + 1]; n2 = grad_rows[3 * o + 2];
        nfl = flags[o];
    }
    // the coefficients are requested BEHIND the first view's rows (vector memory returns in order) and the loop body takes the geometry
    // chain, which does not need them, first: in a one-view launch the 192 B per splat arrive under that arithmetic
    float4 q4[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) q4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < 12; ++c) q4[c] = p4[shn_tiled_f4(il, c)];
    }

    for (int view = 0; view < n_views; ++view) {
        const DvsCam cam = dvs_load_cam(view);
        const int64_t o = (int64_t)view * n + il;
        const float4 r0 = n0, r1 = n1, r2 = n2;
        const uint32_t fl = nfl;
        {
            const int64_t on = (view + 1 < n_views && ((vis >> (view + 1)) & 1u)) ? o + n : 0;
            n0 = grad_rows[3 * on]; n1 = grad_rows[3 * on + 1]; n2 = grad_rows[3 * on + 2];
            nfl = flags[on];
        }
        // NOHOIST: keep the view-independent intermediates (exp of the scales, rotation, 3D covariance) from being hoisted out of the
        // loop — they are cheap to recompute and would otherwise stay live across it
        float s0_ = in_s0, s1_ = in_s1, s2_ = in_s2, q0_ = in_q.x, q1_ = in_q.y, q2_ = in_q.z, q3_ = in_q.w, op_ = in_op;
        if (NOHOIST) asm volatile("" : "+v"(s0_), "+v"(s1_), "+v"(s2_), "+v"(q0_), "+v"(q1_), "+v"(q2_), "+v"(q3_), "+v"(op_));
        float gcol[3] = {0.f, 0.f, 0.f};
        if ((vis >> view) & 1u) {
            if (rezero) { grad_rows[3 * o] = z4; grad_rows[3 * o + 1] = z4; grad_rows[3 * o + 2] = z4; }
            float gpv[3] = {0.f, 0.f, 0.f}, g_opv;
            float2 dmv;
            a9_geometry(cam, px, py, pz, s0_, s1_, s2_, make_float4(q0_, q1_, q2_, q3_), op_, fl, r0, r1, antialias, grad_mode, gpv, Gs, g_opv, dmv);
            const float dL_dcol[3] = {r1.z, r1.w, r2.x};
            // colour -> view direction (the SH rows themselves: k_sh_grad_combine)
            const float dxw = px - cam.campos[0], dyw = py - cam.campos[1], dzw = pz - cam.campos[2];
            const float dl = dvs_sqrt_rn((dxw * dxw + dyw * dyw) + dzw * dzw);
            const float inv_dl = 1.0f / dl;
            const float ux = dxw * inv_dl, uy = dyw * inv_dl, uz = dzw * inv_dl;
            float gdir[3] = {0.f, 0.f, 0.f};
            float sk[16];                                    // s_k = sum_ch shN[k][ch] * gc[ch] (a9_dir_grad)
#pragma unroll
            for (int k = 0; k < 16; ++k) sk[k] = 0.f;
            float gc[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { gc[ch] = (fl & (1u << ch)) ? 0.f : dL_dcol[ch]; gcol[ch] = gc[ch]; }
            // (a9_dir_grad does not read the sums of the bands above `deg`; with deg == 0 the coefficient registers hold zeros)
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const float qv[4] = {q4[c].x, q4[c].y, q4[c].z, q4[c].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = c * 4 + u;                                 // compile-time: coefficient k = e/3 + 1, channel e%3
                    if (e < 45) {
                        const int k = e / 3 + 1, ch = e % 3;
                        sk[k] = __builtin_fmaf(qv[u], gc[ch], sk[k]);
                    }
                }
            }
            if (FUSE_SH) {              // the epilogue builds the SH rows from these: what k_sh_grad_combine would re-read and recompute
                float* lv = l_view + (size_t)view * (6 * PP_BLOCK) + threadIdx.x;
                lv[0] = gc[0]; lv[PP_BLOCK] = gc[1]; lv[2 * PP_BLOCK] = gc[2]; lv[3 * PP_BLOCK] = ux; lv[4 * PP_BLOCK] = uy; lv[5 * PP_BLOCK] = uz;
            }
            {
                a9_dir_grad(deg, ux, uy, uz, sk, gdir);
                const float ug = (ux * gdir[0] + uy * gdir[1]) + uz * gdir[2];
                gpv[0] += (gdir[0] - ux * ug) * inv_dl; gpv[1] += (gdir[1] - uy * ug) * inv_dl; gpv[2] += (gdir[2] - uz * ug) * inv_dl;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] += gpv[k];
            g_op += g_opv;
            ag.x += r2.y; ag.y += r2.z;
            dm.x += dmv.x; dm.y += dmv.y;
        }
        // per-view colour gradient, always overwritten (zero for culled splats and clamped channels). Stored straight from the lane
        // (three 4-B stores at stride 12; the L2 merges them): staging it through LDS for full-line stores cost two workgroup barriers
        // per view, which this latency-bound kernel feels more than the partial-line writes
        if (!FUSE_SH && valid) {
            float* dc = out_dcolor + ((size_t)view * n + (size_t)i) * 3;
            dc[0] = gcol[0]; dc[1] = gcol[1]; dc[2] = gcol[2];
        }
    }

    if (vis) a9_sigma_to_params(in_s0, in_s1, in_s2, in_q, Gs, gsc, gq);       // once per splat: linear in the summed dL/dSigma
    // FUSE_SH (round 6; the one-GPU path, where nobody else needs the per-view colour gradients): dL/dsh0 = sum_v SH_C0 gc_v and
    // dL/dshN[k] = sum_v basis_k(dir_v) gc_v are built HERE, after the view loop — its registers are free by now — from the (gc, dir)
    // pairs the loop left in LDS, with the expressions and the view order of k_sh_grad_combine (bit-identical rows). Saved against the
    // two-kernel form: 12 B per (view, splat) written and read back, the positions read again, a launch.
    float acc0[3] = {0.f, 0.f, 0.f};
    if (FUSE_SH) {
        float acc[48];
#pragma unroll
        for (int e = 0; e < 48; ++e) acc[e] = 0.f;
        for (int view = 0; view < n_views; ++view) {
            if (!((vis >> view) & 1u)) continue;
            const float* lv = l_view + (size_t)view * (6 * PP_BLOCK) + threadIdx.x;
            const float gc[3] = {lv[0], lv[PP_BLOCK], lv[2 * PP_BLOCK]};
            if (gc[0] == 0.f && gc[1] == 0.f && gc[2] == 0.f) continue;              // fully clamped in this view
            float bas[16];
            dvs_sh_basis(deg, lv[3 * PP_BLOCK], lv[4 * PP_BLOCK], lv[5 * PP_BLOCK], bas);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) acc0[ch] = __builtin_fmaf(bas[0], gc[ch], acc0[ch]);
#pragma unroll
            for (int k = 1; k < 16; ++k)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) acc[(k - 1) * 3 + ch] = __builtin_fmaf(bas[k], gc[ch], acc[(k - 1) * 3 + ch]);
        }
        if (valid) {
            float4* d4 = reinterpret_cast<float4*>(g_shN);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                float4 o = c == 11 ? make_float4(acc[44], 0.f, 0.f, 0.f) : make_float4(acc[c * 4], acc[c * 4 + 1], acc[c * 4 + 2], acc[c * 4 + 3]);
                const int64_t idx = shn_tiled_f4(i, c);
                if (ACCUM) { const float4 p = d4[idx]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                d4[idx] = o;
            }
        }
    }
    if (valid) {
        if (out_absgrad2d) {
            float2 a = ag;
            if (ACCUM) { const float2 q = out_absgrad2d[i]; a.x += q.x; a.y += q.y; }
            out_absgrad2d[i] = a;
        }
        if (out_mean2d) {
            float2 mm = dm;
            if (ACCUM) { const float2 q = out_mean2d[i]; mm.x += q.x; mm.y += q.y; }
            out_mean2d[i] = mm;
        }
        if (ACCUM) {
            g_opacity[i] += g_op;
            float4 q = reinterpret_cast<float4*>(g_rot)[i];
            q.x += gq[0]; q.y += gq[1]; q.z += gq[2]; q.w += gq[3];
            reinterpret_cast<float4*>(g_rot)[i] = q;
        } else {
            g_opacity[i] = g_op;
            reinterpret_cast<float4*>(g_rot)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
        }
    }
    __syncthreads();
    float* l_pos = lds, *l_scl = lds + PP_BLOCK * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) { l_pos[threadIdx.x * 3 + k] = gp[k]; l_scl[threadIdx.x * 3 + k] = gsc[k]; }
    __syncthreads();
    stage_rows_out<3, ACCUM>(g_pos, l_pos, base, i1);
    stage_rows_out<3, ACCUM>(g_scale, l_scl, base, i1);
    if (FUSE_SH) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) l_pos[threadIdx.x * 3 + k] = acc0[k];
        __syncthreads();
        stage_rows_out<3, ACCUM>(g_sh0, l_pos, base, i1);
    }
}

// ---- factorised SH gradient: rows from per-view colour gradients -----------------------------------------------------
// dL/dsh0 = SH_C0 * gc and dL/dshN[k] = basis_k(dir_v) * gc are rank-1 in the 3-float colour gradient gc of a view, so in
// data-parallel training each GPU only has to all-gather gc (12 B/splat/view) instead of all-reducing the 192-B SH rows;
// every replica then rebuilds the summed rows locally from its own copy of the positions and the views' camera centres.
#define COMBINE_MAX_VIEWS 64
struct CombineViews { float campos[COMBINE_MAX_VIEWS][3]; };

template <bool ACCUM, bool TILED>
__global__ void __launch_bounds__(PP_BLOCK)
k_sh_grad_combine(CombineViews views_arg /* MUST stay the first parameter: read through the kernarg pointer below */, int n,
                  const float* __restrict__ pos, int deg, int n_views,
                  const float* __restrict__ dcolor /*[n_views, n, 3]*/, float* __restrict__ g_sh0, float* __restrict__ g_shN) {
    // Indexing a by-value struct with a runtime view number makes the compiler spill the whole table to scratch (48 x 16-B
    // scratch stores per lane, a scratch load per use). The table sits at offset 0 of the kernarg segment, which is uniform,
    // read-only memory: address it directly (scalar loads with a dynamic offset).
    (void)views_arg;
    typedef const __attribute__((address_space(4))) CombineViews* KernargViews;
    const KernargViews vp = (KernargViews)__builtin_amdgcn_kernarg_segment_ptr();
#define views (*vp)
    extern __shared__ __attribute__((aligned(16))) float lds[];   // ROWS: [PP_BLOCK*45] + [PP_BLOCK*3]; TILED: [PP_BLOCK*3]
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    const int i = (int)(base + threadIdx.x);
    float* l_sh0 = TILED ? lds : lds + PP_BLOCK * 45;
    float acc0[3] = {0.f, 0.f, 0.f};
    if (TILED) {
        // the 45 sums stay in registers (an LDS row per lane made this kernel LDS-bound: one read-modify-write per FMA), the
        // k loop is fully unrolled (basis entries above `deg` are zero), and the twelve float4 chunks leave straight from registers
        float acc[48];
#pragma unroll
        for (int e = 0; e < 48; ++e) acc[e] = 0.f;
        if (i < n) {
            const float px = pos[3 * (int64_t)i], py = pos[3 * (int64_t)i + 1], pz = pos[3 * (int64_t)i + 2];
            for (int v = 0; v < n_views; ++v) {
                const float* gcp = dcolor + ((int64_t)v * n + i) * 3;
                const float gc[3] = {gcp[0], gcp[1], gcp[2]};
                if (gc[0] == 0.f && gc[1] == 0.f && gc[2] == 0.f) continue;          // culled / fully clamped in this view
                const float dxw = px - views.campos[v][0], dyw = py - views.campos[v][1], dzw = pz - views.campos[v][2];
                const float dl = dvs_sqrt_rn((dxw * dxw + dyw * dyw) + dzw * dzw);
                const float inv_dl = 1.0f / dl;
                float bas[16];
                dvs_sh_basis(deg, dxw * inv_dl, dyw * inv_dl, dzw * inv_dl, bas);
                // (explicit FMAs: this file is compiled without contraction for A2's sake, and the kernel is 60 % vector-ALU-busy)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) acc0[ch] = __builtin_fmaf(bas[0], gc[ch], acc0[ch]);
#pragma unroll
                for (int k = 1; k < 16; ++k)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[(k - 1) * 3 + ch] = __builtin_fmaf(bas[k], gc[ch], acc[(k - 1) * 3 + ch]);
            }
            float4* d4 = reinterpret_cast<float4*>(g_shN);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                float4 o = c == 11 ? make_float4(acc[44], 0.f, 0.f, 0.f)          // chunk 11 holds element 44 and three pads
                                   : make_float4(acc[c * 4], acc[c * 4 + 1], acc[c * 4 + 2], acc[c * 4 + 3]);
                const int64_t idx = shn_tiled_f4(i, c);
                if (ACCUM) { const float4 p = d4[idx]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                d4[idx] = o;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) l_sh0[threadIdx.x * 3 + k] = acc0[k];
        __syncthreads();
    } else {
        float* row = lds + threadIdx.x * 45;
        for (int e = 0; e < 45; ++e) row[e] = 0.f;
        if (i < n) {
            const float px = pos[3 * (int64_t)i], py = pos[3 * (int64_t)i + 1], pz = pos[3 * (int64_t)i + 2];
            const int ncoef = (deg + 1) * (deg + 1);
            for (int v = 0; v < n_views; ++v) {
                const float* gcp = dcolor + ((int64_t)v * n + i) * 3;
                const float gc[3] = {gcp[0], gcp[1], gcp[2]};
                if (gc[0] == 0.f && gc[1] == 0.f && gc[2] == 0.f) continue;
                const float dxw = px - views.campos[v][0], dyw = py - views.campos[v][1], dzw = pz - views.campos[v][2];
                const float dl = dvs_sqrt_rn((dxw * dxw + dyw * dyw) + dzw * dzw);
                const float inv_dl = 1.0f / dl;
                float bas[16];
                dvs_sh_basis(deg, dxw * inv_dl, dyw * inv_dl, dzw * inv_dl, bas);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) acc0[ch] += bas[0] * gc[ch];
                for (int k = 1; k < ncoef; ++k)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) row[(k - 1) * 3 + ch] += bas[k] * gc[ch];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) l_sh0[threadIdx.x * 3 + k] = acc0[k];
        __syncthreads();
        stage_rows_out<45, ACCUM>(g_shN, lds, base, n);
    }
    stage_rows_out<3, ACCUM>(g_sh0, l_sh0, base, n);
#undef views
}

// ---- relayout between the reference rows [n][45] and the tiled layout --------------------------------------------------
__global__ void __launch_bounds__(PP_BLOCK)
k_shn_relayout(int n, const float* __restrict__ src, float* __restrict__ dst, int to_tiled) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PP_BLOCK*45]
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    const int i = (int)(base + threadIdx.x);
    float* row = lds + threadIdx.x * 45;
    if (to_tiled) {
        stage_rows_in<45>(src, lds, base, n);
        __syncthreads();
        // pad lanes of the last tile and the 3 pad floats of every splat are written as zero: the tiled array is fully defined
        if ((i >> 6) < ((n + 63) >> 6)) {
            float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n) o = c == 11 ? make_float4(row[44], 0.f, 0.f, 0.f)
                                       : make_float4(row[c * 4], row[c * 4 + 1], row[c * 4 + 2], row[c * 4 + 3]);
                d4[shn_tiled_f4(i, c)] = o;
            }
        }
    } else {
        if (i < n) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const float4 q = s4[shn_tiled_f4(i, c)];
                row[c * 4] = q.x;
                if (c < 11) { row[c * 4 + 1] = q.y; row[c * 4 + 2] = q.z; row[c * 4 + 3] = q.w; }
            }
        }
        __syncthreads();
        stage_rows_out<45, false>(dst, lds, base, n);
    }
}

// ---- launchers -----------------------------------------------------------------------------------
hipError_t dvs_launch_preprocess_fwd(hipStream_t st, int n, const float* pos, const float* sh0, const float* shN,
                                     const float* opacity, const float* scale, const float* rot, const DvsCams& cams, int n_views,
                                     int deg, int antialias, int tiles_x, int tiles_y, int* radii, float* splat2d,
                                     float* depth, uint32_t* flags,
                                     uint32_t* tiles_touched, uint32_t* depth_key, uint32_t* ids, int shn_tiled, uint32_t* rect, uint32_t* rect16,
                                     uint32_t* rect8, uint32_t* kred, int first, int count) {
    if (n <= 0) return hipSuccess;
    const int i0 = first < 0 ? 0 : first, i1 = count < 0 ? n : (first + count < n ? first + count : n);
    if (i1 <= i0) return hipSuccess;
    const int grid = (i1 - i0 + PP_BLOCK - 1) / PP_BLOCK;
    if (shn_tiled) {
#define DVS_A2(T, M, LDS) hipLaunchKernelGGL((k_preprocess_fwd<T, M>), dim3(grid), dim3(PP_BLOCK), LDS, st, cams, n_views, n, pos, sh0, shN, opacity, \
                                             scale, rot, deg, antialias, tiles_x, tiles_y, radii, (float4*)splat2d, depth, flags,          \
                                             tiles_touched, depth_key, ids, (uint2*)rect, (uint4*)rect16, rect8, kred, i0, i1)
        if (n_views > 1) DVS_A2(true, true, 0); else DVS_A2(true, false, 0);
    } else {
        const size_t lds = deg > 0 ? (size_t)PP_BLOCK * 45 * sizeof(float) : 0;
        if (n_views > 1) DVS_A2(false, true, lds); else DVS_A2(false, false, lds);
#undef DVS_A2
    }
    return hipGetLastError();
}

hipError_t dvs_launch_preprocess_bwd(hipStream_t st, int n, const float* pos, const float* shN, const float* opacity,
                                     const float* scale, const float* rot, const DvsCam& cam, int deg, int antialias,
                                     const int* radii, const uint32_t* flags, float* grad_rows, float* g_pos,
                                     float* g_sh0, float* g_shN, float* g_opacity, float* g_scale, float* g_rot,
                                     float* out_absgrad2d, float* out_mean2d, float* out_dcolor, int accumulate, int rezero,
                                     int shn_tiled, int grad_mode) {
    if (n <= 0) return hipSuccess;
    const int grid = (n + PP_BLOCK - 1) / PP_BLOCK;
    // ROWS: 45 floats per lane of staging; TILED: only the four 3-float groups go through LDS
    const size_t lds = (size_t)PP_BLOCK * (shn_tiled ? 12 : 45) * sizeof(float);
#define DVS_PPB(A, T)                                                                                                        \
    hipLaunchKernelGGL((k_preprocess_bwd<A, T>), dim3(grid), dim3(PP_BLOCK), lds, st, n, pos, shN, opacity, scale, rot, cam, deg, \
                       antialias, radii, flags, (float4*)grad_rows, g_pos, g_sh0, g_shN, g_opacity, g_scale, g_rot,           \
                       (float2*)out_absgrad2d, (float2*)out_mean2d, out_dcolor, rezero, grad_mode)
    if (accumulate) { if (shn_tiled) DVS_PPB(true, true); else DVS_PPB(true, false); }
    else { if (shn_tiled) DVS_PPB(false, true); else DVS_PPB(false, false); }
#undef DVS_PPB
    return hipGetLastError();
}

hipError_t dvs_launch_preprocess_bwd_views(hipStream_t st, int n, int n_views, const float* pos, const float* shN, const float* opacity,
                                           const float* scale, const float* rot, const DvsCams& cams, int deg, int antialias,
                                           const int* radii, const uint32_t* flags, float* grad_rows, float* g_pos, float* g_opacity,
                                           float* g_scale, float* g_rot, float* out_absgrad2d, float* out_mean2d, float* out_dcolor,
                                           int accumulate, int rezero, int grad_mode, int first, int count, float* g_sh0, float* g_shN) {
    if (n <= 0 || n_views <= 0) return hipSuccess;
    const int i0 = first < 0 ? 0 : first, i1 = count < 0 ? n : (first + count < n ? first + count : n);
    if (i1 <= i0) return hipSuccess;
    const int grid = (i1 - i0 + PP_BLOCK - 1) / PP_BLOCK;
    const bool fuse = g_sh0 && g_shN;               // the SH rows are built in the kernel's epilogue; out_dcolor is not written
    const size_t lds = (size_t)PP_BLOCK * (6 + (fuse ? 6 * n_views : 0)) * sizeof(float);
#ifdef DVS_EXPERIMENT
    static const bool nohoist = getenv("DVS_A9V_NOHOIST") && getenv("DVS_A9V_NOHOIST")[0] == '1';      // experiment builds only
#else
    constexpr bool nohoist = false;
#endif
#define DVS_PPV1(A, N, F)                                                                                                        \
    hipLaunchKernelGGL((k_preprocess_bwd_views<A, N, F>), dim3(grid), dim3(PP_BLOCK), lds, st, cams, n_views, n, pos, shN, opacity, scale, rot, \
                       deg, antialias, radii, flags, (float4*)grad_rows, g_pos, g_opacity, g_scale, g_rot,                         \
                       (float2*)out_absgrad2d, (float2*)out_mean2d, out_dcolor, g_sh0, g_shN, rezero, grad_mode, i0, i1)
#define DVS_PPV(A) do { if (fuse) DVS_PPV1(A, false, true); else if (nohoist) DVS_PPV1(A, true, false); else DVS_PPV1(A, false, false); } while (0)
    if (accumulate) DVS_PPV(true); else DVS_PPV(false);
#undef DVS_PPV
#undef DVS_PPV1
    return hipGetLastError();
}

hipError_t dvs_launch_sh_grad_combine(hipStream_t st, int n, const float* pos, int deg, int n_views, const float* campos_host,
                                      const float* dcolor, float* g_sh0, float* g_shN, int accumulate, int shn_tiled) {
    if (n <= 0 || n_views <= 0) return hipSuccess;
    const int grid = (n + PP_BLOCK - 1) / PP_BLOCK;
    const size_t lds = (size_t)PP_BLOCK * (shn_tiled ? 3 : 48) * sizeof(float);
    int acc = accumulate;
    for (int v0 = 0; v0 < n_views; v0 += COMBINE_MAX_VIEWS) {           // more than 64 views: chunks, accumulating
        const int nv = n_views - v0 < COMBINE_MAX_VIEWS ? n_views - v0 : COMBINE_MAX_VIEWS;
        CombineViews cv;
        for (int v = 0; v < nv; ++v) for (int k = 0; k < 3; ++k) cv.campos[v][k] = campos_host[(size_t)(v0 + v) * 3 + k];
        const float* dc = dcolor + (size_t)v0 * n * 3;
#define DVS_CMB(A, T) hipLaunchKernelGGL((k_sh_grad_combine<A, T>), dim3(grid), dim3(PP_BLOCK), lds, st, cv, n, pos, deg, nv, dc, g_sh0, g_shN)
        if (acc) { if (shn_tiled) DVS_CMB(true, true); else DVS_CMB(true, false); }
        else { if (shn_tiled) DVS_CMB(false, true); else DVS_CMB(false, false); }
#undef DVS_CMB
        acc = 1;
    }
    return hipGetLastError();
}

// ---- colour gradients straight from the A8 rows (before A9) ----------------------------------------------------------------
// dcolor[o] = dL/d(colour) of (view, splat) o with the clamped channels masked — exactly what A9 emits — so that a data-parallel
// trainer can start the all-gather of the colour gradients while A9 is still running.
__global__ void __launch_bounds__(PP_BLOCK)
k_dcolor_from_rows(int64_t total, const int* __restrict__ radii, const uint32_t* __restrict__ flags, const float4* __restrict__ rows,
                   float* __restrict__ dcolor) {
    __shared__ float l_col[PP_BLOCK * 3];
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    const int64_t o = base + threadIdx.x;
    float g[3] = {0.f, 0.f, 0.f};
    if (o < total && radii[o] > 0) {
        const float4 r1 = rows[3 * o + 1];
        const float r2x = rows[3 * o + 2].x;
        const uint32_t fl = flags[o];
        g[0] = (fl & 1u) ? 0.f : r1.z; g[1] = (fl & 2u) ? 0.f : r1.w; g[2] = (fl & 4u) ? 0.f : r2x;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) l_col[threadIdx.x * 3 + k] = g[k];
    __syncthreads();
    const int64_t lim = (total - base) * 3;                 // consecutive threads write consecutive floats
    for (int e = threadIdx.x; e < PP_BLOCK * 3 && e < lim; e += PP_BLOCK) dcolor[base * 3 + e] = l_col[e];
}

hipError_t dvs_launch_dcolor_from_rows(hipStream_t st, int64_t total, const int* radii, const uint32_t* flags, const float* rows, float* dcolor) {
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((total + PP_BLOCK - 1) / PP_BLOCK);
    hipLaunchKernelGGL(k_dcolor_from_rows, dim3(grid), dim3(PP_BLOCK), 0, st, total, radii, flags, (const float4*)rows, dcolor);
    return hipGetLastError();
}

hipError_t dvs_launch_shn_relayout(hipStream_t st, int n, const float* src, float* dst, int to_tiled) {
    if (n <= 0) return hipSuccess;
    const int grid = (((n + 63) / 64) * 64 + PP_BLOCK - 1) / PP_BLOCK;      // cover the pad lanes of the last tile
    hipLaunchKernelGGL(k_shn_relayout, dim3(grid), dim3(PP_BLOCK), (size_t)PP_BLOCK * 45 * sizeof(float), st, n, src, dst, to_tiled);
    return hipGetLastError();
}
