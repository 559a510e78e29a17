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
