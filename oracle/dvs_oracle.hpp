// dvs_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// PARITY UNPINNED: the reference's rasterizer (`gsplatrast`) and trainer (`gstrain`) are closed
// source and absent from /root/reference (README.md:32,46; diverse_utils/CMakeLists.txt:1-3), and
// the reference holds no test or golden vector for this path.  This file is a
// restatement of the tile-rasterizer algorithm of the lineage the reference credits (README.md:95)
// using the conventions that ARE pinned by in-tree code:
//   cov3D  R*S*S^T*R^T, quaternion (w,x,y,z)        diverse/assets/shaders/gaussian/gsplat_vs.hlsl:171-209
//   EWA cov2D with 1.3*tan_fov clamp                gsplat_vs.hlsl:74-110
//   +0.3 low-pass, eigenvalues                      gsplat_vs.hlsl:304-311
//   anti-alias factor sqrt(max(det/det_blur,0))     gsplat_vs.hlsl:296-301
//   ndc2Pix                                         gsplat_vs.hlsl:211-214
//   opacity cut 1/255                               gsplat_vs.hlsl:269
//   SH basis, constants, signs, max(colour,0)       gsplat_sh.hlsl:42-61,65-103,124
//   activations exp/sigmoid/normalise, SH_C0, +0.5  diverse/source/assets/gaussian_model.cpp:14-22,128,137-159
//   parameter layout (59 fp32 / splat)              gaussian_model.cpp:60-65, editor.cpp:1578
//   16x16 tiles                                     gaussian_common.hlsl:162-163
// It is validated by fp64 central finite differences, by an independent dense PyTorch-autograd
// formulation (tests/golden/make_golden.py) and by the known-answer tests in tests/test_kat.py.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
//
// Templated on the scalar type: float = the fp32 specification the HIP path must match
// (fixed operation order, built with -ffp-contract=off); double = ground truth for the gradients.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include "../include/dvs_raster.h"

namespace dvso {

// ---- constants fixed by the build (SURVEY.md §8(a) "A-notes") --------------------------------
constexpr int   kTile = DVS_TILE;
constexpr float kAlphaMin = 1.0f / 255.0f;   // gsplat_ps.hlsl:65 minAlpha
constexpr float kAlphaMax = 0.99f;           // canonical trainer cap (viewer uses 0.999, gsplat_ps.hlsl:63)
constexpr float kTStop = 1e-4f;
constexpr float kLowPass = 0.3f;             // gsplat_vs.hlsl:304-306
constexpr float kNear = 0.2f;
constexpr float kFovGuard = 1.3f;            // gsplat_vs.hlsl:81-82

constexpr float SH_C0 = 0.28209479177387814f;  // gaussian_model.cpp:128
constexpr float SH_C1 = 0.4886025119029199f;   // gsplat_sh.hlsl:42
constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                            -1.0925484305920792f, 0.5462742152960396f};   // gsplat_sh.hlsl:46-50
constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                            -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};  // gsplat_sh.hlsl:54-60

// ---- deterministic exp -------------------------------------------------------------------------
// exp feeds integer decisions (scale -> cov -> radius -> tile rect), so the fp32 specification uses
// a fixed sequence of IEEE-exact operations (rint, fma, mul, add, exponent insertion) that the HIP
// kernel restates operation for operation.  Cephes-style: n = rint(x*log2e); r = x - n*ln2 (hi/lo);
// degree-5 polynomial in r (Horner, fma); result scaled by 2^n.  |rel err| < 2 ulp on [-87, 88].
inline float det_expf(float x) {
    x = std::fmin(std::fmax(x, -87.0f), 88.0f);
    const float n = std::nearbyint(x * 1.44269504088896341f);
    float r = std::fma(n, -0.693359375f, x);
    r = std::fma(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = std::fma(p, r, 1.3981999507e-3f);
    p = std::fma(p, r, 8.3334519073e-3f);
    p = std::fma(p, r, 4.1665795894e-2f);
    p = std::fma(p, r, 1.6666665459e-1f);
    p = std::fma(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = std::fma(p, r2, r) + 1.0f;
    const int32_t e = (int32_t)n;                       // in [-126, 127]
    uint32_t bits = (uint32_t)(e + 127) << 23;
    float s; std::memcpy(&s, &bits, 4);
    return y * s;
}
template <class T> inline T det_exp(T x);
template <> inline float  det_exp<float>(float x)   { return det_expf(x); }
template <> inline double det_exp<double>(double x) { return std::exp(x); }

template <class T> inline T sigmoid(T x) { return T(1) / (T(1) + det_exp<T>(-x)); }   // gaussian_common.hlsl:127-129

// ---- SH basis (gsplat_sh.hlsl:65-103) -----------------------------------------------------------
// b[0..15]; b[0] is the dc basis SH_C0; b[k], k>=1 multiplies shN[(k-1)*3 + c].
template <class T> inline void sh_basis(int deg, T x, T y, T z, T b[16]) {
    for (int i = 0; i < 16; ++i) b[i] = T(0);
    b[0] = T(SH_C0);
    if (deg < 1) return;
    b[1] = -T(SH_C1) * y; b[2] = T(SH_C1) * z; b[3] = -T(SH_C1) * x;
    if (deg < 2) return;
    const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = T(SH_C2[0]) * xy;
    b[5] = T(SH_C2[1]) * yz;
    b[6] = T(SH_C2[2]) * (T(2) * zz - xx - yy);
    b[7] = T(SH_C2[3]) * xz;
    b[8] = T(SH_C2[4]) * (xx - yy);
    if (deg < 3) return;
    b[9]  = T(SH_C3[0]) * y * (T(3) * xx - yy);
    b[10] = T(SH_C3[1]) * xy * z;
    b[11] = T(SH_C3[2]) * y * (T(4) * zz - xx - yy);
    b[12] = T(SH_C3[3]) * z * (T(2) * zz - T(3) * xx - T(3) * yy);
    b[13] = T(SH_C3[4]) * x * (T(4) * zz - xx - yy);
    b[14] = T(SH_C3[5]) * z * (xx - yy);
    b[15] = T(SH_C3[6]) * x * (xx - T(3) * yy);
}
// d b[k] / d(x,y,z)
template <class T> inline void sh_basis_grad(int deg, T x, T y, T z, T db[16][3]) {
    for (int i = 0; i < 16; ++i) db[i][0] = db[i][1] = db[i][2] = T(0);
    if (deg < 1) return;
    db[1][1] = -T(SH_C1); db[2][2] = T(SH_C1); db[3][0] = -T(SH_C1);
    if (deg < 2) return;
    const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    db[4][0] = T(SH_C2[0]) * y;          db[4][1] = T(SH_C2[0]) * x;
    db[5][1] = T(SH_C2[1]) * z;          db[5][2] = T(SH_C2[1]) * y;
    db[6][0] = T(SH_C2[2]) * (-T(2) * x); db[6][1] = T(SH_C2[2]) * (-T(2) * y); db[6][2] = T(SH_C2[2]) * (T(4) * z);
    db[7][0] = T(SH_C2[3]) * z;          db[7][2] = T(SH_C2[3]) * x;
    db[8][0] = T(SH_C2[4]) * (T(2) * x); db[8][1] = T(SH_C2[4]) * (-T(2) * y);
    if (deg < 3) return;
    db[9][0]  = T(SH_C3[0]) * (T(6) * xy);                 db[9][1]  = T(SH_C3[0]) * (T(3) * xx - T(3) * yy);
    db[10][0] = T(SH_C3[1]) * yz;  db[10][1] = T(SH_C3[1]) * xz;  db[10][2] = T(SH_C3[1]) * xy;
    db[11][0] = T(SH_C3[2]) * (-T(2) * xy); db[11][1] = T(SH_C3[2]) * (T(4) * zz - xx - T(3) * yy); db[11][2] = T(SH_C3[2]) * (T(8) * yz);
    db[12][0] = T(SH_C3[3]) * (-T(6) * xz); db[12][1] = T(SH_C3[3]) * (-T(6) * yz); db[12][2] = T(SH_C3[3]) * (T(6) * zz - T(3) * xx - T(3) * yy);
    db[13][0] = T(SH_C3[4]) * (T(4) * zz - T(3) * xx - yy); db[13][1] = T(SH_C3[4]) * (-T(2) * xy); db[13][2] = T(SH_C3[4]) * (T(8) * xz);
    db[14][0] = T(SH_C3[5]) * (T(2) * xz); db[14][1] = T(SH_C3[5]) * (-T(2) * yz); db[14][2] = T(SH_C3[5]) * (xx - yy);
    db[15][0] = T(SH_C3[6]) * (T(3) * xx - T(3) * yy); db[15][1] = T(SH_C3[6]) * (-T(6) * xy);
}

// rotation matrix of a UNIT quaternion (w,x,y,z), row-major R[i*3+k]  (gsplat_vs.hlsl:196-200)
template <class T> inline void quat_to_rot(T r, T x, T y, T z, T R[9]) {
    R[0] = T(1) - T(2) * (y * y + z * z); R[1] = T(2) * (x * y - r * z);        R[2] = T(2) * (x * z + r * y);
    R[3] = T(2) * (x * y + r * z);        R[4] = T(1) - T(2) * (x * x + z * z); R[5] = T(2) * (y * z - r * x);
    R[6] = T(2) * (x * z - r * y);        R[7] = T(2) * (y * z + r * x);        R[8] = T(1) - T(2) * (x * x + y * y);
}
// cov3D (xx,xy,xz,yy,yz,zz) = R diag(s^2) R^T   (gsplat_vs.hlsl:203-208)
template <class T> inline void cov3d_from_scale_rot(const T s[3], const T R[9], T cov[6]) {
    T M[9];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) M[i * 3 + k] = R[i * 3 + k] * s[k];
    cov[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
    cov[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
    cov[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
    cov[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
    cov[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
    cov[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
}

// deterministic ln(x), x > 0 normal: the CPU twin of dvs_log_det (divshot_amd/csrc/dvs_device.h), same operations in the same order
inline float det_logf(float x) {
    uint32_t bits; std::memcpy(&bits, &x, 4);
    int e = (int)(bits >> 23) - 127;
    uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m; std::memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    float p = 0.111111111f;
    p = std::fma(p, t2, 0.142857143f);
    p = std::fma(p, t2, 0.2f);
    p = std::fma(p, t2, 0.333333333f);
    p = std::fma(p, t2, 1.0f);
    return std::fma((float)e, 0.693147181f, (2.0f * t) * p);
}
template <class T> inline T det_log(T x);
template <> inline float  det_log<float>(float x)   { return det_logf(x); }
template <> inline double det_log<double>(double x) { return std::log(x); }

// DVS_TILES_TIGHT (include/dvs_raster.h): the tiles of a splat's rectangle its alpha >= 1/255 ellipse can reach — the CPU twin of
// dvs_tight_tile_mask (divshot_amd/csrc/dvs_device.h), the same IEEE operations in the same order, so that the fp32 instantiation
// reproduces the HIP path's instance list bit for bit. Anchor of the rule: the viewer's per-pixel alpha cull, gsplat_ps.hlsl:60-65.
template <class T> inline uint64_t tight_tile_mask(T a, T b, T c, T o, T mx, T my, int rminx, int rminy, int rmaxx, int rmaxy) {
    const T ac = a * c;
    const T det = std::fmax(T(0), std::fma(T(-2.4e-7f), ac, ac - b * b));
    const T inv_c = T(1) / c, inv_a = T(1) / a;
    const T det_c = det * inv_c, det_a = det * inv_a, nb_c = -b * inv_c, nb_a = -b * inv_a;
    const T kappa = T(2) * (det_log<T>(o) + T(5.5422648f));
    const T inf = std::numeric_limits<T>::infinity();
    uint64_t mask = 0;
    int t = 0;
    for (int ty = rminy; ty < rmaxy; ++ty) {
        const T y0 = T(ty * kTile) - my, y1 = y0 + T(kTile - 1);
        const bool hin = y0 <= T(0) && y1 >= T(0);
        const T ye = y0 > T(0) ? y0 : y1;
        const T hx = nb_a * ye, hbase = (ye * ye) * det_a;
        const T Yf = std::fmax(std::fabs(y0), std::fabs(y1));
        for (int tx = rminx; tx < rmaxx; ++tx, ++t) {
            const T x0 = T(tx * kTile) - mx, x1 = x0 + T(kTile - 1);
            const bool vin = x0 <= T(0) && x1 >= T(0);
            // margin scaled with the size the form's terms reach on this tile (the composite kernels' per-pixel evaluation cancels)
            const T Xf = std::fmax(std::fabs(x0), std::fabs(x1));
            const T mag = ((a * Xf) * Xf + ((T(2) * std::fabs(b)) * Xf) * Yf) + (c * Yf) * Yf;
            T qmin = T(0);
            if (!(vin && hin)) {
                qmin = inf;
                if (!vin) {
                    const T xe = x0 > T(0) ? x0 : x1;
                    const T vy = nb_c * xe;
                    const T d = std::fmin(std::fmax(vy, y0), y1) - vy;
                    qmin = (c * d) * d + (xe * xe) * det_c;
                }
                if (!hin) {
                    const T d = std::fmin(std::fmax(hx, x0), x1) - hx;
                    qmin = std::fmin(qmin, (a * d) * d + hbase);
                }
            }
            if (!(qmin > kappa + T(1e-6f) * mag)) mask |= (1ull << t);
        }
    }
    return mask;
}

// ---- state -------------------------------------------------------------------------------------
template <class T> struct State {
    int n = 0, W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    dvs_camera cam{};
    dvs_opts opts{};
    // inputs (kept for backward)
    std::vector<T> pos, sh0, shN, opacity, scale, rot;
    // A2 outputs
    std::vector<int32_t> radii;
    std::vector<T> mean2d, depth, conic_opacity, rgb;
    std::vector<uint32_t> flags, tiles_touched;
    std::vector<int32_t> rect;           // [n,4] minx,miny,maxx,maxy
    std::vector<uint64_t> take_masks;    // [T][4]: the decisions THIS forward took, in the layout dvs_debug_record_decisions uses (when record_masks is set)
    bool record_masks = false;
    std::vector<uint64_t> replay;        // decision replay (tests): [T][4] — bit l of word q of list position j = pixel lane l of 8x8 quadrant q of
                                         // the tile takes that entry (as the HIP forward decided: dvs_debug_record_decisions); empty = own decisions
    std::vector<uint64_t> tile_mask;     // DVS_TILES_TIGHT: surviving tiles of the rectangle, row-major (all ones: the whole rectangle)
    std::vector<uint32_t> depth_bits;    // fp32 bit pattern of depth (0 for culled)
    // A3-A6
    std::vector<uint32_t> offsets;       // inclusive scan of tiles_touched
    std::vector<uint64_t> keys;          // sorted
    std::vector<uint32_t> vals;          // sorted
    std::vector<uint32_t> ranges;        // [tiles,2]
    std::vector<uint32_t> forced_vals, forced_ranges;   // tests (dvso_set_lists): composite over THESE tile lists instead of binning — the fp64
                                         // instantiation over the lists of the fp32 run, when a radius on an integer boundary makes fp64 bin differently
    // A7
    std::vector<T> out_color, final_T;   // [3,H,W], [H,W]
    std::vector<uint32_t> n_contrib;
    std::vector<uint8_t> fragile;        // [H,W] a threshold decision at this pixel had < 1e-5 relative margin
    std::vector<uint8_t> cap_fragile;    // [H,W] an alpha at this pixel lies within 1e-5 (relative) of the 0.99 cap (a decision of the DVS_GRAD_TRUE backward)
    // A8
    std::vector<T> dL_dmean2d, dL_dconic_opacity, dL_drgb, absgrad;   // [n,2],[n,4],[n,3],[n,2]
    // A9
    std::vector<T> g_pos, g_sh0, g_shN, g_opacity, g_scale, g_rot;
    // compute-side counter: pixel-splat pairs evaluated forward (before per-pixel termination)
    uint64_t interactions = 0;
};

template <class T> inline T xform_x(const float* m, T x, T y, T z, int r) {   // row r of "m * (x,y,z,1)"
    return ((T(m[0 * 4 + r]) * x + T(m[1 * 4 + r]) * y) + T(m[2 * 4 + r]) * z) + T(m[3 * 4 + r]);
}

// ---- A2: preprocess forward --------------------------------------------------------------------
template <class T> void preprocess_forward(State<T>& S) {
    const int n = S.n;
    const dvs_camera& cam = S.cam;
    const int W = cam.width, H = cam.height;
    S.W = W; S.H = H;
    S.tiles_x = (W + kTile - 1) / kTile; S.tiles_y = (H + kTile - 1) / kTile;
    S.radii.assign(n, 0); S.mean2d.assign(2 * (size_t)n, T(0)); S.depth.assign(n, T(0));
    S.conic_opacity.assign(4 * (size_t)n, T(0)); S.rgb.assign(3 * (size_t)n, T(0));
    S.flags.assign(n, 0); S.tiles_touched.assign(n, 0); S.rect.assign(4 * (size_t)n, 0); S.tile_mask.assign(n, ~0ull);
    S.depth_bits.assign(n, 0);
    const int deg = S.opts.sh_degree;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const T px = S.pos[3 * i], py = S.pos[3 * i + 1], pz = S.pos[3 * i + 2];
        const T tx = xform_x<T>(cam.view, px, py, pz, 0);
        const T ty = xform_x<T>(cam.view, px, py, pz, 1);
        const T tz = xform_x<T>(cam.view, px, py, pz, 2);
        if (!(tz > T(kNear))) continue;                                   // near cull (NaN-safe)
        {   // a NaN log-scale or opacity logit culls the splat (the clamps inside det_exp would otherwise turn it into a number)
            const T l0 = S.scale[3 * i], l1 = S.scale[3 * i + 1], l2 = S.scale[3 * i + 2], lo = S.opacity[i];
            if (!(l0 == l0) || !(l1 == l1) || !(l2 == l2) || !(lo == lo)) continue;
        }
        const T hx = xform_x<T>(cam.proj, px, py, pz, 0);
        const T hy = xform_x<T>(cam.proj, px, py, pz, 1);
        const T hw = xform_x<T>(cam.proj, px, py, pz, 3);
        const T pw = T(1) / (hw + T(0.0000001f));                          // gsplat_vs.hlsl:253
        const T ndc_x = hx * pw, ndc_y = hy * pw;

        // activations (gaussian_model.cpp:137-152)
        T s[3];
        for (int k = 0; k < 3; ++k) s[k] = det_exp<T>(S.scale[3 * i + k]);
        const T qr = S.rot[4 * i], qx = S.rot[4 * i + 1], qy = S.rot[4 * i + 2], qz = S.rot[4 * i + 3];
        const T qn = std::sqrt(((qr * qr + qx * qx) + qy * qy) + qz * qz);
        if (!(qn > T(0))) continue;
        const T inv_qn = T(1) / qn;
        T R[9];
        quat_to_rot<T>(qr * inv_qn, qx * inv_qn, qy * inv_qn, qz * inv_qn, R);
        T c3[6];
        cov3d_from_scale_rot<T>(s, R, c3);

        // EWA projection (gsplat_vs.hlsl:74-110)
        const T limx = T(kFovGuard) * T(cam.tan_fovx), limy = T(kFovGuard) * T(cam.tan_fovy);
        const T txtz = tx / tz, tytz = ty / tz;
        uint32_t fl = 0;
        if (txtz < -limx || txtz > limx) fl |= 8u;
        if (tytz < -limy || tytz > limy) fl |= 16u;
        const T txc = std::fmin(limx, std::fmax(-limx, txtz)) * tz;
        const T tyc = std::fmin(limy, std::fmax(-limy, tytz)) * tz;
        const T fx = T(cam.focal_x), fy = T(cam.focal_y);
        const T J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
        const T J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
        // Wv[k][i] = d t_k / d p_i = view[i*4+k]
        T T0[3], T1[3];
        for (int k = 0; k < 3; ++k) {
            T0[k] = J00 * T(cam.view[k * 4 + 0]) + J02 * T(cam.view[k * 4 + 2]);
            T1[k] = J11 * T(cam.view[k * 4 + 1]) + J12 * T(cam.view[k * 4 + 2]);
        }
        // v0 = Sigma * T0, v1 = Sigma * T1
        const T v0x = (c3[0] * T0[0] + c3[1] * T0[1]) + c3[2] * T0[2];
        const T v0y = (c3[1] * T0[0] + c3[3] * T0[1]) + c3[4] * T0[2];
        const T v0z = (c3[2] * T0[0] + c3[4] * T0[1]) + c3[5] * T0[2];
        const T v1x = (c3[0] * T1[0] + c3[1] * T1[1]) + c3[2] * T1[2];
        const T v1y = (c3[1] * T1[0] + c3[3] * T1[1]) + c3[4] * T1[2];
        const T v1z = (c3[2] * T1[0] + c3[4] * T1[1]) + c3[5] * T1[2];
        const T cxx = (T0[0] * v0x + T0[1] * v0y) + T0[2] * v0z;
        const T cxy = (T0[0] * v1x + T0[1] * v1y) + T0[2] * v1z;
        const T cyy = (T1[0] * v1x + T1[1] * v1y) + T1[2] * v1z;

        const T a = cxx + T(kLowPass), b = cxy, c = cyy + T(kLowPass);
        const T det = a * c - b * b;
        if (!(det > T(0))) continue;
        T opac = sigmoid<T>(S.opacity[i]);
        if (S.opts.antialias) {
            const T det_orig = cxx * cyy - b * b;
            const T aa = std::sqrt(std::fmax(T(0), det_orig / det));        // gsplat_vs.hlsl:298-300
            opac = opac * aa;
        }
        if (!(opac > T(kAlphaMin))) continue;                              // gsplat_vs.hlsl:269
        const T det_inv = T(1) / det;
        const T mid = T(0.5f) * (a + c);
        const T lam = mid + std::sqrt(std::fmax(T(0.1f), mid * mid - det));
        const T radf = std::ceil(T(3) * std::sqrt(lam));
        const T m2x = ((ndc_x + T(1)) * T(W) - T(1)) * T(0.5f);            // ndc2Pix gsplat_vs.hlsl:211-214
        const T m2y = ((ndc_y + T(1)) * T(H) - T(1)) * T(0.5f);
        // tile rect [min,max) clipped to the grid; float clamp first so the int cast is always defined
        const T gx = T(S.tiles_x), gy = T(S.tiles_y), inv_tile = T(1.0f / kTile);
        const int rminx = (int)std::fmin(gx, std::fmax(T(0), (m2x - radf) * inv_tile));
        const int rminy = (int)std::fmin(gy, std::fmax(T(0), (m2y - radf) * inv_tile));
        const int rmaxx = (int)std::fmin(gx, std::fmax(T(0), (m2x + radf + T(kTile - 1)) * inv_tile));
        const int rmaxy = (int)std::fmin(gy, std::fmax(T(0), (m2y + radf + T(kTile - 1)) * inv_tile));
        const int touched = (rmaxx - rminx) * (rmaxy - rminy);
        if (touched <= 0) continue;
        // radius is stored as int; a radius beyond int range is clamped (the rect is already grid-clipped)
        const int radius = (int)std::fmin(radf, T(1 << 30));

        // colour from SH (gsplat_sh.hlsl:65-124), view direction = normalize(pos - campos)
        const T dx = px - T(cam.campos[0]), dy = py - T(cam.campos[1]), dz = pz - T(cam.campos[2]);
        const T dl = std::sqrt((dx * dx + dy * dy) + dz * dz);
        const T inv_dl = T(1) / dl;
        T bas[16];
        sh_basis<T>(deg, dx * inv_dl, dy * inv_dl, dz * inv_dl, bas);
        const int ncoef = (deg + 1) * (deg + 1);
        for (int ch = 0; ch < 3; ++ch) {
            T col = bas[0] * S.sh0[3 * i + ch];
            for (int k = 1; k < ncoef; ++k) col = col + bas[k] * S.shN[45 * (size_t)i + (k - 1) * 3 + ch];
            col = col + T(0.5f);
            if (col < T(0)) { fl |= (1u << ch); col = T(0); }
            S.rgb[3 * i + ch] = col;
        }

        S.radii[i] = radius;
        S.mean2d[2 * i] = m2x; S.mean2d[2 * i + 1] = m2y;
        S.depth[i] = tz;
        float dz32 = (float)tz; uint32_t db; std::memcpy(&db, &dz32, 4);
        S.depth_bits[i] = db;
        S.conic_opacity[4 * i + 0] = c * det_inv;
        S.conic_opacity[4 * i + 1] = -b * det_inv;
        S.conic_opacity[4 * i + 2] = a * det_inv;
        S.conic_opacity[4 * i + 3] = opac;
        S.flags[i] = fl;
        S.tiles_touched[i] = (uint32_t)touched;
        if (S.opts.tile_bounds == DVS_TILES_TIGHT && touched <= 64) {      // opt-in: only the tiles the alpha >= 1/255 ellipse reaches
            S.tile_mask[i] = tight_tile_mask<T>(c * det_inv, -b * det_inv, a * det_inv, opac, m2x, m2y, rminx, rminy, rmaxx, rmaxy);
            S.tiles_touched[i] = (uint32_t)__builtin_popcountll(S.tile_mask[i]);
        }
        S.rect[4 * i + 0] = rminx; S.rect[4 * i + 1] = rminy; S.rect[4 * i + 2] = rmaxx; S.rect[4 * i + 3] = rmaxy;
    }
}

// ---- A3-A6: scan, duplicate with keys, stable sort by (tile|depth), tile ranges -----------------
template <class T> void bin(State<T>& S) {
    const int n = S.n;
    S.offsets.resize(n);
    uint64_t run = 0;
    for (int i = 0; i < n; ++i) { run += S.tiles_touched[i]; S.offsets[i] = (uint32_t)run; }
    const uint64_t Tn = run;
    std::vector<uint64_t> keys(Tn);
    std::vector<uint32_t> vals(Tn);
    for (int i = 0; i < n; ++i) {
        if (S.radii[i] <= 0) continue;
        uint64_t off = (i == 0) ? 0 : S.offsets[i - 1];
        const int32_t* r = &S.rect[4 * (size_t)i];
        const uint64_t mask = S.tile_mask[i];
        int t = 0;
        for (int y = r[1]; y < r[3]; ++y)
            for (int x = r[0]; x < r[2]; ++x, ++t) {
                if (mask != ~0ull && !((mask >> t) & 1ull)) continue;           // DVS_TILES_TIGHT: this tile is out of the ellipse's reach
                const uint64_t tile = (uint64_t)y * S.tiles_x + x;
                keys[off] = (tile << 32) | S.depth_bits[i];      // depth > 0: raw IEEE bits order correctly
                vals[off] = (uint32_t)i;
                ++off;
            }
    }
    // stable sort by key: ties (same tile, same depth bits) keep emission order = ascending splat id
    std::vector<uint32_t> perm(Tn);
    for (uint64_t j = 0; j < Tn; ++j) perm[j] = (uint32_t)j;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t l, uint32_t r) { return keys[l] < keys[r]; });
    S.keys.resize(Tn); S.vals.resize(Tn);
    for (uint64_t j = 0; j < Tn; ++j) { S.keys[j] = keys[perm[j]]; S.vals[j] = vals[perm[j]]; }
    const int tiles = S.tiles_x * S.tiles_y;
    S.ranges.assign(2 * (size_t)tiles, 0);
    for (uint64_t j = 0; j < Tn; ++j) {
        const uint32_t t = (uint32_t)(S.keys[j] >> 32);
        if (j == 0 || t != (uint32_t)(S.keys[j - 1] >> 32)) S.ranges[2 * t] = (uint32_t)j;
        if (j + 1 == Tn || t != (uint32_t)(S.keys[j + 1] >> 32)) S.ranges[2 * t + 1] = (uint32_t)(j + 1);
    }
}

// ---- A7: alpha-composite forward -----------------------------------------------------------------
template <class T> inline bool near_rel(T v, T thr) { return std::fabs(v - thr) <= T(1e-5) * std::fabs(thr); }

template <class T> void render_forward(State<T>& S) {
    const int W = S.W, H = S.H;
    const size_t P = (size_t)W * H;
    S.out_color.assign(3 * P, T(0)); S.final_T.assign(P, T(1)); S.n_contrib.assign(P, 0); S.fragile.assign(P, 0); S.cap_fragile.assign(P, 0);
    S.take_masks.assign(S.record_masks ? 4 * S.vals.size() : 0, 0ull);     // (a tile is handled by one thread: no two threads share a word)
    uint64_t inter = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : inter)
    for (int tile = 0; tile < S.tiles_x * S.tiles_y; ++tile) {
        const int tx = tile % S.tiles_x, ty = tile / S.tiles_x;
        const uint32_t beg = S.ranges[2 * tile], end = S.ranges[2 * tile + 1];
        for (int ly = 0; ly < kTile; ++ly)
            for (int lx = 0; lx < kTile; ++lx) {
                const int x = tx * kTile + lx, y = ty * kTile + ly;
                if (x >= W || y >= H) continue;
                const T pxf = T(x), pyf = T(y);
                T Tr = T(1), C0 = T(0), C1 = T(0), C2 = T(0);
                uint32_t contributor = 0, last = 0;
                uint8_t frag = 0;
                const bool rp = !S.replay.empty();                    // decisions given (k_render_fwd's): no threshold is evaluated here
                const int rq = (ly >> 3) * 2 + (lx >> 3), rl = (ly & 7) * 8 + (lx & 7);
                for (uint32_t j = beg; j < end; ++j) {
                    ++contributor; ++inter;
                    if (rp && !((S.replay[4 * (size_t)j + rq] >> rl) & 1ull)) continue;
                    const uint32_t id = S.vals[j];
                    const T dx = S.mean2d[2 * id] - pxf, dy = S.mean2d[2 * id + 1] - pyf;
                    const T* co = &S.conic_opacity[4 * (size_t)id];
                    const T power = T(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (!rp && power > T(0)) continue;
                    const T oa = co[3] * det_exp<T>(power);
                    const T alpha = std::fmin(T(kAlphaMax), oa);
                    if (near_rel<T>(oa, T(kAlphaMin))) frag = 1;
                    if (near_rel<T>(oa, T(kAlphaMax))) frag |= 2;         // (bit 1: the 0.99 cap — a decision of the DVS_GRAD_TRUE backward)
                    if (!rp && alpha < T(kAlphaMin)) continue;
                    const T test_T = Tr * (T(1) - alpha);
                    if (near_rel<T>(test_T, T(kTStop))) frag |= 1;
                    if (!rp && test_T < T(kTStop)) break;
                    const T w = alpha * Tr;
                    C0 = C0 + S.rgb[3 * id] * w; C1 = C1 + S.rgb[3 * id + 1] * w; C2 = C2 + S.rgb[3 * id + 2] * w;
                    Tr = test_T;
                    last = contributor;
                    if (S.record_masks) S.take_masks[4 * (size_t)j + rq] |= 1ull << rl;
                }
                const size_t pix = (size_t)y * W + x;
                S.final_T[pix] = Tr; S.n_contrib[pix] = last; S.fragile[pix] = frag & 1; S.cap_fragile[pix] = (frag >> 1) & 1;
                S.out_color[0 * P + pix] = C0 + Tr * T(S.cam.bg[0]);
                S.out_color[1 * P + pix] = C1 + Tr * T(S.cam.bg[1]);
                S.out_color[2 * P + pix] = C2 + Tr * T(S.cam.bg[2]);
            }
    }
    S.interactions = inter;
}

// ---- A8: alpha-composite backward ----------------------------------------------------------------
// opts.grad_mode = DVS_GRAD_TRUE: true gradient of the forward above (the 0.99 clamp blocks the gradient to
// conic/mean/opacity, as autograd of min() does; verified against fp64 finite differences).
// opts.grad_mode = DVS_GRAD_LINEAGE: the backward of the lineage the reference credits (README.md:95), see dvs_raster.h.
// Per-instance partials are written at their sorted position and summed per splat in sorted order, so the result is
// deterministic regardless of thread count.
template <class T> void render_backward(State<T>& S, const T* dL_dout /*[3,H,W]*/) {
    const int W = S.W, H = S.H, n = S.n;
    const size_t P = (size_t)W * H;
    const size_t Tn = S.vals.size();
    std::vector<T> part(Tn * 12, T(0));   // per instance: mean2d(2) conic(3) opac(1) rgb(3) abs(2) pad
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < S.tiles_x * S.tiles_y; ++tile) {
        const int tx = tile % S.tiles_x, ty = tile / S.tiles_x;
        const uint32_t beg = S.ranges[2 * tile], end = S.ranges[2 * tile + 1];
        for (int ly = 0; ly < kTile; ++ly)
            for (int lx = 0; lx < kTile; ++lx) {
                const int x = tx * kTile + lx, y = ty * kTile + ly;
                if (x >= W || y >= H) continue;
                const size_t pix = (size_t)y * W + x;
                const T pxf = T(x), pyf = T(y);
                const T T_final = S.final_T[pix];
                T Tr = T_final;
                const uint32_t last = S.n_contrib[pix];
                const T dLp[3] = {dL_dout[pix], dL_dout[P + pix], dL_dout[2 * P + pix]};
                const T bg_dot = (T(S.cam.bg[0]) * dLp[0] + T(S.cam.bg[1]) * dLp[1]) + T(S.cam.bg[2]) * dLp[2];
                T accum[3] = {T(0), T(0), T(0)}, last_alpha = T(0), last_col[3] = {T(0), T(0), T(0)};
                const bool rp = !S.replay.empty();
                const int rq = (ly >> 3) * 2 + (lx >> 3), rl = (ly & 7) * 8 + (lx & 7);
                for (uint32_t k = last; k-- > 0;) {          // contributor index k+1, list position beg+k
                    const uint32_t j = beg + k;
                    (void)end;
                    if (rp && !((S.replay[4 * (size_t)j + rq] >> rl) & 1ull)) continue;
                    const uint32_t id = S.vals[j];
                    const T dx = S.mean2d[2 * id] - pxf, dy = S.mean2d[2 * id + 1] - pyf;
                    const T* co = &S.conic_opacity[4 * (size_t)id];
                    const T power = T(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (!rp && power > T(0)) continue;
                    const T G = det_exp<T>(power);
                    const T oa = co[3] * G;
                    const T alpha = std::fmin(T(kAlphaMax), oa);
                    if (!rp && alpha < T(kAlphaMin)) continue;
                    Tr = Tr / (T(1) - alpha);
                    const T dchannel_dcolor = alpha * Tr;
                    T dL_dalpha = T(0);
                    T* pp = &part[(size_t)j * 12];
                    for (int ch = 0; ch < 3; ++ch) {
                        const T cch = S.rgb[3 * id + ch];
                        accum[ch] = last_alpha * last_col[ch] + (T(1) - last_alpha) * accum[ch];
                        last_col[ch] = cch;
                        dL_dalpha = dL_dalpha + (cch - accum[ch]) * dLp[ch];
                        pp[6 + ch] += dchannel_dcolor * dLp[ch];
                    }
                    dL_dalpha = dL_dalpha * Tr;
                    last_alpha = alpha;
                    dL_dalpha = dL_dalpha + (-T_final / (T(1) - alpha)) * bg_dot;
                    // clamped: alpha is constant w.r.t. G and opacity (DVS_GRAD_TRUE). The credited lineage (README.md:95)
                    // lets the gradient through the cap as if alpha = opacity*G (DVS_GRAD_LINEAGE).
                    if (oa > T(kAlphaMax) && S.opts.grad_mode == DVS_GRAD_TRUE) continue;
                    const T dL_dG = co[3] * dL_dalpha;
                    const T gdx = G * dx, gdy = G * dy;
                    const T dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const T dG_ddely = -gdy * co[2] - gdx * co[1];
                    // d = mean2d - pix  =>  d(d)/d(mean2d) = +1
                    pp[0] += dL_dG * dG_ddelx;
                    pp[1] += dL_dG * dG_ddely;
                    pp[9] += std::fabs(dL_dG * dG_ddelx);
                    pp[10] += std::fabs(dL_dG * dG_ddely);
                    pp[2] += T(-0.5f) * gdx * dx * dL_dG;      // d/d conic.a
                    pp[3] += -gdx * dy * dL_dG;                // d/d conic.b (full derivative)
                    pp[4] += T(-0.5f) * gdy * dy * dL_dG;      // d/d conic.c
                    pp[5] += G * dL_dalpha;                    // d/d opacity
                }
            }
    }
    S.dL_dmean2d.assign(2 * (size_t)n, T(0)); S.dL_dconic_opacity.assign(4 * (size_t)n, T(0));
    S.dL_drgb.assign(3 * (size_t)n, T(0)); S.absgrad.assign(2 * (size_t)n, T(0));
    for (size_t j = 0; j < Tn; ++j) {
        const uint32_t id = S.vals[j];
        const T* pp = &part[j * 12];
        S.dL_dmean2d[2 * id] += pp[0]; S.dL_dmean2d[2 * id + 1] += pp[1];
        for (int k = 0; k < 4; ++k) S.dL_dconic_opacity[4 * (size_t)id + k] += pp[2 + k];
        for (int k = 0; k < 3; ++k) S.dL_drgb[3 * (size_t)id + k] += pp[6 + k];
        S.absgrad[2 * id] += pp[9]; S.absgrad[2 * id + 1] += pp[10];
    }
}

// ---- A9: preprocess backward ---------------------------------------------------------------------
template <class T> void preprocess_backward(State<T>& S) {
    const int n = S.n;
    const dvs_camera& cam = S.cam;
    const int W = cam.width, H = cam.height;
    const int deg = S.opts.sh_degree;
    S.g_pos.assign(3 * (size_t)n, T(0)); S.g_sh0.assign(3 * (size_t)n, T(0)); S.g_shN.assign(45 * (size_t)n, T(0));
    S.g_opacity.assign(n, T(0)); S.g_scale.assign(3 * (size_t)n, T(0)); S.g_rot.assign(4 * (size_t)n, T(0));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        if (S.radii[i] <= 0) continue;
        const T px = S.pos[3 * i], py = S.pos[3 * i + 1], pz = S.pos[3 * i + 2];
        const uint32_t fl = S.flags[i];
        T gp[3] = {T(0), T(0), T(0)};    // dL/dpos

        // --- recompute forward intermediates (same expressions as preprocess_forward) ---
        const T tx = xform_x<T>(cam.view, px, py, pz, 0);
        const T ty = xform_x<T>(cam.view, px, py, pz, 1);
        const T tz = xform_x<T>(cam.view, px, py, pz, 2);
        const T hx = xform_x<T>(cam.proj, px, py, pz, 0);
        const T hy = xform_x<T>(cam.proj, px, py, pz, 1);
        const T hw = xform_x<T>(cam.proj, px, py, pz, 3);
        const T pw = T(1) / (hw + T(0.0000001f));
        T s[3];
        for (int k = 0; k < 3; ++k) s[k] = det_exp<T>(S.scale[3 * i + k]);
        const T qr0 = S.rot[4 * i], qx0 = S.rot[4 * i + 1], qy0 = S.rot[4 * i + 2], qz0 = S.rot[4 * i + 3];
        const T qn = std::sqrt(((qr0 * qr0 + qx0 * qx0) + qy0 * qy0) + qz0 * qz0);
        const T inv_qn = T(1) / qn;
        const T qr = qr0 * inv_qn, qx = qx0 * inv_qn, qy = qy0 * inv_qn, qz = qz0 * inv_qn;
        T R[9];
        quat_to_rot<T>(qr, qx, qy, qz, R);
        T c3[6];
        cov3d_from_scale_rot<T>(s, R, c3);
        const T limx = T(kFovGuard) * T(cam.tan_fovx), limy = T(kFovGuard) * T(cam.tan_fovy);
        const T txtz = tx / tz, tytz = ty / tz;
        const T cl_x = std::fmin(limx, std::fmax(-limx, txtz)), cl_y = std::fmin(limy, std::fmax(-limy, tytz));
        const T txc = cl_x * tz, tyc = cl_y * tz;
        const T fx = T(cam.focal_x), fy = T(cam.focal_y);
        const T J00 = fx / tz, J02 = -(fx * txc) / (tz * tz);
        const T J11 = fy / tz, J12 = -(fy * tyc) / (tz * tz);
        T T0[3], T1[3];
        for (int k = 0; k < 3; ++k) {
            T0[k] = J00 * T(cam.view[k * 4 + 0]) + J02 * T(cam.view[k * 4 + 2]);
            T1[k] = J11 * T(cam.view[k * 4 + 1]) + J12 * T(cam.view[k * 4 + 2]);
        }
        const T v0[3] = {(c3[0] * T0[0] + c3[1] * T0[1]) + c3[2] * T0[2], (c3[1] * T0[0] + c3[3] * T0[1]) + c3[4] * T0[2],
                         (c3[2] * T0[0] + c3[4] * T0[1]) + c3[5] * T0[2]};
        const T v1[3] = {(c3[0] * T1[0] + c3[1] * T1[1]) + c3[2] * T1[2], (c3[1] * T1[0] + c3[3] * T1[1]) + c3[4] * T1[2],
                         (c3[2] * T1[0] + c3[4] * T1[1]) + c3[5] * T1[2]};
        const T cxx = (T0[0] * v0[0] + T0[1] * v0[1]) + T0[2] * v0[2];
        const T cxy = (T0[0] * v1[0] + T0[1] * v1[1]) + T0[2] * v1[2];
        const T cyy = (T1[0] * v1[0] + T1[1] * v1[1]) + T1[2] * v1[2];
        const T a = cxx + T(kLowPass), b = cxy, c = cyy + T(kLowPass);
        const T det = a * c - b * b;
        const T det_inv = T(1) / det;

        // --- 1. colour: SH backward ---
        const T dxw = px - T(cam.campos[0]), dyw = py - T(cam.campos[1]), dzw = pz - T(cam.campos[2]);
        const T dl = std::sqrt((dxw * dxw + dyw * dyw) + dzw * dzw);
        const T inv_dl = T(1) / dl;
        const T ux = dxw * inv_dl, uy = dyw * inv_dl, uz = dzw * inv_dl;
        T bas[16], dbas[16][3];
        sh_basis<T>(deg, ux, uy, uz, bas);
        sh_basis_grad<T>(deg, ux, uy, uz, dbas);
        const int ncoef = (deg + 1) * (deg + 1);
        T gdir[3] = {T(0), T(0), T(0)};
        for (int ch = 0; ch < 3; ++ch) {
            const T gc = (fl & (1u << ch)) ? T(0) : S.dL_drgb[3 * i + ch];
            S.g_sh0[3 * i + ch] = bas[0] * gc;
            for (int k = 1; k < ncoef; ++k) {
                const T coef = S.shN[45 * (size_t)i + (k - 1) * 3 + ch];
                S.g_shN[45 * (size_t)i + (k - 1) * 3 + ch] = bas[k] * gc;
                gdir[0] += dbas[k][0] * coef * gc; gdir[1] += dbas[k][1] * coef * gc; gdir[2] += dbas[k][2] * coef * gc;
            }
        }
        // u = d/|d|: dL/dd = (g - u (u.g)) / |d|
        {
            const T ug = (ux * gdir[0] + uy * gdir[1]) + uz * gdir[2];
            gp[0] += (gdir[0] - ux * ug) * inv_dl; gp[1] += (gdir[1] - uy * ug) * inv_dl; gp[2] += (gdir[2] - uz * ug) * inv_dl;
        }

        // --- 2. opacity (+ anti-alias factor) ---
        T g_cxx = T(0), g_cxy = T(0), g_cyy = T(0);
        const T g_opac = S.dL_dconic_opacity[4 * (size_t)i + 3];
        const T sig = sigmoid<T>(S.opacity[i]);
        T g_sig = g_opac;
        if (S.opts.antialias) {
            const T det_orig = cxx * cyy - b * b;
            const T ratio = det_orig / det;
            const T aa = std::sqrt(std::fmax(T(0), ratio));
            g_sig = g_opac * aa;
            if (ratio > T(0)) {
                const T g_aa = g_opac * sig;
                const T g_ratio = g_aa * T(0.5f) / aa;
                const T g_do = g_ratio * det_inv;
                const T g_db = -g_ratio * det_orig * det_inv * det_inv;
                g_cxx += g_do * cyy + g_db * c;
                g_cyy += g_do * cxx + g_db * a;
                g_cxy += T(-2) * b * (g_do + g_db);
            }
        }
        S.g_opacity[i] = g_sig * sig * (T(1) - sig);

        // --- 3. conic = (c, -b, a)/det ---
        {
            const T gka = S.dL_dconic_opacity[4 * (size_t)i + 0], gkb = S.dL_dconic_opacity[4 * (size_t)i + 1],
                    gkc = S.dL_dconic_opacity[4 * (size_t)i + 2];
            const T Ssum = (gka * c - gkb * b) + gkc * a;
            const T g_det = -Ssum * det_inv * det_inv;
            g_cxx += gkc * det_inv + g_det * c;
            g_cyy += gka * det_inv + g_det * a;
            g_cxy += -gkb * det_inv + g_det * (T(-2) * b);
        }

        // --- 4. cov2D = Tm Sigma Tm^T ---
        // dL/dSigma (full, unsymmetrised) G[i][j] = g_cxx T0i T0j + g_cxy T0i T1j + g_cyy T1i T1j
        T Gm[9];
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
            Gm[r * 3 + q] = (g_cxx * T0[r] * T0[q] + g_cxy * T0[r] * T1[q]) + g_cyy * T1[r] * T1[q];
        T gT0[3], gT1[3];
        for (int k = 0; k < 3; ++k) {
            gT0[k] = T(2) * g_cxx * v0[k] + g_cxy * v1[k];
            gT1[k] = T(2) * g_cyy * v1[k] + g_cxy * v0[k];
        }
        // --- 5. Tm = J Wv ---
        T gJ00 = T(0), gJ02 = T(0), gJ11 = T(0), gJ12 = T(0);
        for (int k = 0; k < 3; ++k) {
            gJ00 += gT0[k] * T(cam.view[k * 4 + 0]); gJ02 += gT0[k] * T(cam.view[k * 4 + 2]);
            gJ11 += gT1[k] * T(cam.view[k * 4 + 1]); gJ12 += gT1[k] * T(cam.view[k * 4 + 2]);
        }
        const T tz2 = T(1) / (tz * tz), tz3 = tz2 / tz;
        T g_tx = T(0), g_ty = T(0), g_tz = T(0);
        g_tz += -fx * tz2 * gJ00 - fy * tz2 * gJ11;
        g_tz += T(2) * fx * txc * tz3 * gJ02 + T(2) * fy * tyc * tz3 * gJ12;     // at fixed txc, tyc
        const T g_txc = -fx * tz2 * gJ02, g_tyc = -fy * tz2 * gJ12;
        // clamped branch: txc = cl_x * tz. DVS_GRAD_TRUE differentiates it through tz; DVS_GRAD_LINEAGE holds the clamped
        // coordinate constant (the credited lineage multiplies dL/dt.x by 0 there and adds nothing to dL/dt.z).
        const bool lineage = S.opts.grad_mode == DVS_GRAD_LINEAGE;
        if (fl & 8u) { if (!lineage) g_tz += g_txc * cl_x; } else g_tx += g_txc;
        if (fl & 16u) { if (!lineage) g_tz += g_tyc * cl_y; } else g_ty += g_tyc;
        for (int k = 0; k < 3; ++k)
            gp[k] += (T(cam.view[k * 4 + 0]) * g_tx + T(cam.view[k * 4 + 1]) * g_ty) + T(cam.view[k * 4 + 2]) * g_tz;

        // --- 6. mean2D ---
        {
            const T gmx = S.dL_dmean2d[2 * i], gmy = S.dL_dmean2d[2 * i + 1];
            const T g_hx = gmx * T(0.5f) * T(W) * pw, g_hy = gmy * T(0.5f) * T(H) * pw;
            const T g_hw = -(gmx * T(0.5f) * T(W) * hx + gmy * T(0.5f) * T(H) * hy) * pw * pw;
            for (int k = 0; k < 3; ++k)
                gp[k] += (T(cam.proj[k * 4 + 0]) * g_hx + T(cam.proj[k * 4 + 1]) * g_hy) + T(cam.proj[k * 4 + 3]) * g_hw;
        }
        (void)hy; (void)ty;

        // --- 7. Sigma = M M^T, M = R diag(s) ---
        T M[9], gM[9];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) M[r * 3 + k] = R[r * 3 + k] * s[k];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) {
            T acc = T(0);
            for (int q = 0; q < 3; ++q) acc += (Gm[r * 3 + q] + Gm[q * 3 + r]) * M[q * 3 + k];
            gM[r * 3 + k] = acc;
        }
        T gR[9];
        for (int k = 0; k < 3; ++k) {
            T gs = T(0);
            for (int r = 0; r < 3; ++r) { gs += gM[r * 3 + k] * R[r * 3 + k]; gR[r * 3 + k] = gM[r * 3 + k] * s[k]; }
            S.g_scale[3 * i + k] = gs * s[k];                                     // through exp
        }
        // R(q) backward (unit quaternion components treated as independent, then normalisation)
        T gq[4];
        gq[0] = T(2) * (-qz * gR[1] + qy * gR[2] + qz * gR[3] - qx * gR[5] - qy * gR[6] + qx * gR[7]);
        gq[1] = T(2) * (qy * gR[1] + qz * gR[2] + qy * gR[3] - T(2) * qx * gR[4] - qr * gR[5] + qz * gR[6] + qr * gR[7] - T(2) * qx * gR[8]);
        gq[2] = T(2) * (-T(2) * qy * gR[0] + qx * gR[1] + qr * gR[2] + qx * gR[3] + qz * gR[5] - qr * gR[6] + qz * gR[7] - T(2) * qy * gR[8]);
        gq[3] = T(2) * (-T(2) * qz * gR[0] - qr * gR[1] + qx * gR[2] + qr * gR[3] - T(2) * qz * gR[4] + qy * gR[5] + qx * gR[6] + qy * gR[7]);
        const T qg = ((qr * gq[0] + qx * gq[1]) + qy * gq[2]) + qz * gq[3];
        S.g_rot[4 * i + 0] = (gq[0] - qr * qg) * inv_qn;
        S.g_rot[4 * i + 1] = (gq[1] - qx * qg) * inv_qn;
        S.g_rot[4 * i + 2] = (gq[2] - qy * qg) * inv_qn;
        S.g_rot[4 * i + 3] = (gq[3] - qz * qg) * inv_qn;

        S.g_pos[3 * i] = gp[0]; S.g_pos[3 * i + 1] = gp[1]; S.g_pos[3 * i + 2] = gp[2];
    }
}

}  // namespace dvso
