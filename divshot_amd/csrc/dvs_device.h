// dvs_device.h — device-side constants and math shared by the gfx950 kernels.
// Conventions follow the in-tree viewer of the reference (fenghuayumo/DIVSHOT):
//   SH constants/basis gsplat_sh.hlsl:42-61,65-103 · rotation matrix gsplat_vs.hlsl:196-200 ·
//   thresholds SURVEY.md §8(a) A-notes (alpha_min 1/255 gsplat_ps.hlsl:65, low-pass 0.3 gsplat_vs.hlsl:304-306,
//   1.3*tan_fov guard gsplat_vs.hlsl:81-82).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DVS_TILE 16
#define DVS_ALPHA_MIN (1.0f / 255.0f)
#define DVS_ALPHA_MAX 0.99f
#define DVS_T_STOP 1e-4f
#define DVS_LOWPASS 0.3f
#define DVS_NEAR 0.2f
#define DVS_FOV_GUARD 1.3f

#define DVS_SH_C0 0.28209479177387814f
#define DVS_SH_C1 0.4886025119029199f
#define DVS_SH_C2_0 1.0925484305920792f
#define DVS_SH_C2_1 (-1.0925484305920792f)
#define DVS_SH_C2_2 0.31539156525252005f
#define DVS_SH_C2_3 (-1.0925484305920792f)
#define DVS_SH_C2_4 0.5462742152960396f
#define DVS_SH_C3_0 (-0.5900435899266435f)
#define DVS_SH_C3_1 2.890611442640554f
#define DVS_SH_C3_2 (-0.4570457994644658f)
#define DVS_SH_C3_3 0.3731763325901154f
#define DVS_SH_C3_4 (-0.4570457994644658f)
#define DVS_SH_C3_5 1.445305721320277f
#define DVS_SH_C3_6 (-0.5900435899266435f)

// flag bits written by preprocess forward
#define DVS_FLAG_CLAMP_R 1u
#define DVS_FLAG_CLAMP_G 2u
#define DVS_FLAG_CLAMP_B 4u
#define DVS_FLAG_CLAMP_X 8u
#define DVS_FLAG_CLAMP_Y 16u

// Camera block as passed to kernels by value (mirrors dvs_camera in include/dvs_raster.h).
struct DvsCam {
    float view[16];
    float proj[16];
    float tan_fovx, tan_fovy;
    float focal_x, focal_y;
    float campos[3];
    int width, height;
    float bg[3];
};

// The cameras of one multi-view batch (one training iteration renders several views: BASELINE config C4). Passed by value as the
// FIRST kernel parameter and read through the kernarg segment pointer: indexing a by-value struct with a runtime view number would
// make the compiler spill it to scratch, while the kernarg segment is uniform read-only memory (scalar loads with a dynamic offset).
#define DVS_MAX_VIEWS 16
struct DvsCams { DvsCam c[DVS_MAX_VIEWS]; };
__device__ __forceinline__ DvsCam dvs_load_cam(int v) {
    typedef const __attribute__((address_space(4))) uint32_t* KW;
    const KW w = (KW)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)v * (sizeof(DvsCam) / 4);
    DvsCam cam;
    uint32_t* d = reinterpret_cast<uint32_t*>(&cam);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(DvsCam) / 4); ++k) d[k] = w[k];
    return cam;
}

// Deterministic exp: exp feeds integer decisions (scale -> cov -> radius -> tile rect), so it is a
// fixed sequence of IEEE-exact operations (v_rndne, v_fma, v_mul, v_add, exponent insert), not
// v_exp_f32. Cephes-style range reduction + degree-5 polynomial; < 2 ulp on [-87, 88].
__device__ __forceinline__ float dvs_exp_det(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    const int e = (int)n;
    return y * __uint_as_float((uint32_t)(e + 127) << 23);
}
// IEEE correctly-rounded sqrt: sqrtf() lowers to v_sqrt_f32 + the +-1 ulp fix-up; __fsqrt_rn() on
// ROCm 7.2 is the raw 1-ulp v_sqrt_f32 and breaks bit-exactness against the CPU oracle.
__device__ __forceinline__ float dvs_sqrt_rn(float x) { return sqrtf(x); }
__device__ __forceinline__ float dvs_sigmoid_det(float x) { return 1.0f / (1.0f + dvs_exp_det(-x)); }

// Deterministic natural logarithm for x > 0 (normal numbers): a fixed sequence of IEEE-exact operations, like dvs_exp_det — it feeds an
// integer decision (which tiles a splat is binned into with DVS_TILES_TIGHT), so the CPU oracle must reproduce it bit for bit.
// x = m 2^e with m in [sqrt(1/2), sqrt(2)); ln m = 2 t (1 + t^2/3 + t^4/5 + t^6/7 + t^8/9), t = (m - 1) / (m + 1), |t| < 0.172:
// truncation below 1e-9.
__device__ __forceinline__ float dvs_log_det(float x) {
    const uint32_t bits = __float_as_uint(x);
    int e = (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    float p = 0.111111111f;
    p = __builtin_fmaf(p, t2, 0.142857143f);
    p = __builtin_fmaf(p, t2, 0.2f);
    p = __builtin_fmaf(p, t2, 0.333333333f);
    p = __builtin_fmaf(p, t2, 1.0f);
    return __builtin_fmaf((float)e, 0.693147181f, (2.0f * t) * p);
}

// DVS_TILES_TIGHT (dvs_raster.h): which tiles of a splat's 3-sigma rectangle its alpha >= 1/255 ellipse can reach. Bit t of the result =
// tile t of the rectangle in row-major order (area <= 64). The minimum of the conic form q(d) = a dx^2 + 2 b dx dy + c dy^2 over the
// tile's pixel-centre rectangle [16 tx, 16 tx + 15] x [16 ty, 16 ty + 15] is 0 if the mean lies inside, otherwise it lies on an edge
// facing the mean; on the vertical line x: q = c (y - y*)^2 + x^2 det / c with y* = -b x / c — non-negative terms only (see
// render.hip stage_batch), det lowered by its rounding bound. The tile stays unless q_min > 2 ln(o / ((1/255)(1 - 1e-3))), i.e. unless
// even the nearest point of the tile stays below alpha = (1/255)(1 - 1e-3): the margin (2e-3 in q, plus 1e-6 of the size the form's TERMS reach on
// the tile: the per-pixel evaluation cancels for thin diagonal splats) covers the rounding of q_min, of
// dvs_log_det and of the per-pixel alpha in the composite kernels, so no contributing pixel can lose its splat.
// EVERY operation is IEEE-exact and in a fixed order (this file is compiled with -ffp-contract=off): the CPU oracle
// (oracle/dvs_oracle.hpp tight_tile_mask) evaluates the same sequence and reproduces the mask bit for bit.
#define DVS_TIGHT_LOG_INV_ALPHA 5.5422648f       /* -ln((1/255)(1 - 1e-3)) */
__device__ __forceinline__ unsigned long long dvs_tight_tile_mask(float a, float b, float c, float o, float mx, float my, int rminx, int rminy,
                                                                  int rmaxx, int rmaxy) {
    const float ac = a * c;
    const float det = fmaxf(0.f, __builtin_fmaf(-2.4e-7f, ac, ac - b * b));
    const float inv_c = 1.0f / c, inv_a = 1.0f / a;
    const float det_c = det * inv_c, det_a = det * inv_a, nb_c = -b * inv_c, nb_a = -b * inv_a;
    const float kappa = 2.0f * (dvs_log_det(o) + DVS_TIGHT_LOG_INV_ALPHA);
    unsigned long long mask = 0ull;
    int t = 0;
    for (int ty = rminy; ty < rmaxy; ++ty) {
        const float y0 = (float)(ty * DVS_TILE) - my, y1 = y0 + (float)(DVS_TILE - 1);
        const bool hin = y0 <= 0.f && y1 >= 0.f;
        const float ye = y0 > 0.f ? y0 : y1;
        const float hx = nb_a * ye, hbase = (ye * ye) * det_a;
        const float Yf = fmaxf(fabsf(y0), fabsf(y1));
        for (int tx = rminx; tx < rmaxx; ++tx, ++t) {
            const float x0 = (float)(tx * DVS_TILE) - mx, x1 = x0 + (float)(DVS_TILE - 1);
            const bool vin = x0 <= 0.f && x1 >= 0.f;
            // the composite kernels evaluate the form per pixel as a cancelling sum (a dx^2 + 2 b dx dy + c dy^2, contracted): its rounding
            // error grows with the size of the TERMS, which for a thin diagonal splat is thousands of times the value. The
            // margin therefore scales with the largest the terms get on this tile: 1e-6 (17 ulp) of a X^2 + 2 |b| X Y + c Y^2 at its far corner.
            const float Xf = fmaxf(fabsf(x0), fabsf(x1));
            const float mag = ((a * Xf) * Xf + ((2.0f * fabsf(b)) * Xf) * Yf) + (c * Yf) * Yf;
            float qmin = 0.f;
            if (!(vin && hin)) {
                qmin = __builtin_inff();
                if (!vin) {
                    const float xe = x0 > 0.f ? x0 : x1;
                    const float vy = nb_c * xe;
                    const float d = fminf(fmaxf(vy, y0), y1) - vy;
                    qmin = (c * d) * d + (xe * xe) * det_c;
                }
                if (!hin) {
                    const float d = fminf(fmaxf(hx, x0), x1) - hx;
                    qmin = fminf(qmin, (a * d) * d + hbase);
                }
            }
            mask |= !(qmin > kappa + 1e-6f * mag) ? (1ull << t) : 0ull;         // NaN-safe: a failed comparison keeps the tile
        }
    }
    return mask;
}

__device__ __forceinline__ float dvs_xform(const float* m, float x, float y, float z, int r) {
    return ((m[0 * 4 + r] * x + m[1 * 4 + r] * y) + m[2 * 4 + r] * z) + m[3 * 4 + r];
}

__device__ __forceinline__ void dvs_quat_to_rot(float r, float x, float y, float z, float R[9]) {
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void dvs_cov3d(const float s[3], const float R[9], float cov[6]) {
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[i * 3 + k] = R[i * 3 + k] * s[k];
    cov[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
    cov[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
    cov[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
    cov[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
    cov[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
    cov[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
}

// SH basis b[0..15] for unit direction (x,y,z); entries above the active degree are zero.
__device__ __forceinline__ void dvs_sh_basis(int deg, float x, float y, float z, float b[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 0.f;
    b[0] = DVS_SH_C0;
    if (deg < 1) return;
    b[1] = -DVS_SH_C1 * y; b[2] = DVS_SH_C1 * z; b[3] = -DVS_SH_C1 * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = DVS_SH_C2_0 * xy;
    b[5] = DVS_SH_C2_1 * yz;
    b[6] = DVS_SH_C2_2 * (2.f * zz - xx - yy);
    b[7] = DVS_SH_C2_3 * xz;
    b[8] = DVS_SH_C2_4 * (xx - yy);
    if (deg < 3) return;
    b[9]  = DVS_SH_C3_0 * y * (3.f * xx - yy);
    b[10] = DVS_SH_C3_1 * xy * z;
    b[11] = DVS_SH_C3_2 * y * (4.f * zz - xx - yy);
    b[12] = DVS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = DVS_SH_C3_4 * x * (4.f * zz - xx - yy);
    b[14] = DVS_SH_C3_5 * z * (xx - yy);
    b[15] = DVS_SH_C3_6 * x * (xx - 3.f * yy);
}

__device__ __forceinline__ void dvs_sh_basis_grad(int deg, float x, float y, float z, float db[16][3]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) db[i][0] = db[i][1] = db[i][2] = 0.f;
    if (deg < 1) return;
    db[1][1] = -DVS_SH_C1; db[2][2] = DVS_SH_C1; db[3][0] = -DVS_SH_C1;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    db[4][0] = DVS_SH_C2_0 * y;           db[4][1] = DVS_SH_C2_0 * x;
    db[5][1] = DVS_SH_C2_1 * z;           db[5][2] = DVS_SH_C2_1 * y;
    db[6][0] = DVS_SH_C2_2 * (-2.f * x);  db[6][1] = DVS_SH_C2_2 * (-2.f * y); db[6][2] = DVS_SH_C2_2 * (4.f * z);
    db[7][0] = DVS_SH_C2_3 * z;           db[7][2] = DVS_SH_C2_3 * x;
    db[8][0] = DVS_SH_C2_4 * (2.f * x);   db[8][1] = DVS_SH_C2_4 * (-2.f * y);
    if (deg < 3) return;
    db[9][0]  = DVS_SH_C3_0 * (6.f * xy);                 db[9][1]  = DVS_SH_C3_0 * (3.f * xx - 3.f * yy);
    db[10][0] = DVS_SH_C3_1 * yz; db[10][1] = DVS_SH_C3_1 * xz; db[10][2] = DVS_SH_C3_1 * xy;
    db[11][0] = DVS_SH_C3_2 * (-2.f * xy); db[11][1] = DVS_SH_C3_2 * (4.f * zz - xx - 3.f * yy); db[11][2] = DVS_SH_C3_2 * (8.f * yz);
    db[12][0] = DVS_SH_C3_3 * (-6.f * xz); db[12][1] = DVS_SH_C3_3 * (-6.f * yz); db[12][2] = DVS_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
    db[13][0] = DVS_SH_C3_4 * (4.f * zz - 3.f * xx - yy); db[13][1] = DVS_SH_C3_4 * (-2.f * xy); db[13][2] = DVS_SH_C3_4 * (8.f * xz);
    db[14][0] = DVS_SH_C3_5 * (2.f * xz); db[14][1] = DVS_SH_C3_5 * (-2.f * yz); db[14][2] = DVS_SH_C3_5 * (xx - yy);
    db[15][0] = DVS_SH_C3_6 * (3.f * xx - 3.f * yy); db[15][1] = DVS_SH_C3_6 * (-6.f * xy);
}
