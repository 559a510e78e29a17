// synth_scene.cpp — deterministic synthetic scenes (include/dvs_scene.h), host only.
// Distributions follow SURVEY.md §8(d): positions uniform in the frustum slab z in [2,12] of the
// reference camera (cone widened 10 %), log-scales N(ln(1.5 z / focal), 0.5^2), quaternions N(0,1)^4,
// opacity logits N(0,1.5^2), SH dc N(0,1), higher bands N(0,0.1^2), pinhole fov_x 60 deg.
#include <cmath>
#include <cstring>
#include "../../include/dvs_scene.h"

namespace {
struct Pcg32 {
    uint64_t state, inc;
    Pcg32(uint64_t seed, uint64_t stream) {
        state = 0; inc = (stream << 1u) | 1u;
        next(); state += seed; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    double uniform() { return (next() >> 8) * (1.0 / 16777216.0); }          // [0,1)
    double normal() {                                                        // Box-Muller, one value per call
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-12) u1 = 1e-12;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};
}  // namespace

extern "C" {

int dvs_make_camera(const float* R, const float* t, float fov_x_deg, int width, int height, dvs_camera* c) {
    if (!R || !t || !c || width <= 0 || height <= 0) return DVS_ERR_INVALID;
    memset(c, 0, sizeof *c);
    const double tanx = std::tan(0.5 * (double)fov_x_deg * 3.14159265358979323846 / 180.0);
    const double tany = tanx * (double)height / (double)width;
    c->tan_fovx = (float)tanx; c->tan_fovy = (float)tany;
    c->focal_x = (float)(width / (2.0 * tanx)); c->focal_y = (float)(height / (2.0 * tany));
    c->width = width; c->height = height;
    // view[c*4+r]: out.r = sum_c view[c*4+r] * in.c   (world -> camera)
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) c->view[k * 4 + r] = R[r * 3 + k];
        c->view[3 * 4 + r] = t[r];
    }
    c->view[15] = 1.f;
    // pinhole clip matrix P (camera -> clip): x' = x / tanx, y' = y / tany, w' = z  (only x, y, w are used)
    const double zn = 0.01, zf = 100.0;
    double P[16] = {0};  // P[c*4+r]
    P[0 * 4 + 0] = 1.0 / tanx; P[1 * 4 + 1] = 1.0 / tany;
    P[2 * 4 + 2] = zf / (zf - zn); P[3 * 4 + 2] = -(zf * zn) / (zf - zn); P[2 * 4 + 3] = 1.0;
    // proj = P * view
    for (int cc = 0; cc < 4; ++cc)
        for (int r = 0; r < 4; ++r) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += P[k * 4 + r] * (double)c->view[cc * 4 + k];
            c->proj[cc * 4 + r] = (float)acc;
        }
    // camera centre = -R^T t
    for (int k = 0; k < 3; ++k) c->campos[k] = -(R[0 * 3 + k] * t[0] + R[1 * 3 + k] * t[1] + R[2 * 3 + k] * t[2]);
    return DVS_OK;
}

int dvs_synth_camera(const dvs_scene_spec* s, int index, dvs_camera* out) {
    if (!s || !out || index < 0 || index >= (s->n_cams > 0 ? s->n_cams : 1)) return DVS_ERR_INVALID;
    double eye[3] = {0, 0, 0};
    if (index > 0) {
        const int ring = (s->n_cams > 1) ? s->n_cams - 1 : 1;
        const double ang = 6.283185307179586 * (double)(index - 1) / (double)ring;
        eye[0] = 0.5 * std::cos(ang); eye[1] = 0.5 * std::sin(ang);
    }
    const double at[3] = {0, 0, 7};
    double f[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
    const double fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (double& v : f) v /= fl;
    const double down[3] = {0, 1, 0};                       // +Y is down in the camera frame
    double r[3] = {down[1] * f[2] - down[2] * f[1], down[2] * f[0] - down[0] * f[2], down[0] * f[1] - down[1] * f[0]};
    const double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (double& v : r) v /= rl;
    const double d[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};
    float R[9], t[3];
    for (int k = 0; k < 3; ++k) { R[0 * 3 + k] = (float)r[k]; R[1 * 3 + k] = (float)d[k]; R[2 * 3 + k] = (float)f[k]; }
    for (int row = 0; row < 3; ++row)
        t[row] = -(float)((double)R[row * 3 + 0] * eye[0] + (double)R[row * 3 + 1] * eye[1] + (double)R[row * 3 + 2] * eye[2]);
    return dvs_make_camera(R, t, s->fov_x_deg > 0 ? s->fov_x_deg : 60.f, s->width, s->height, out);
}

int dvs_synth_splats(const dvs_scene_spec* s, float* pos, float* sh0, float* shN, float* opacity, float* scale, float* rot) {
    if (!s || s->n < 0 || !pos || !sh0 || !shN || !opacity || !scale || !rot) return DVS_ERR_INVALID;
    const double fov = (s->fov_x_deg > 0 ? s->fov_x_deg : 60.0) * 3.14159265358979323846 / 180.0;
    const double tanx = std::tan(0.5 * fov), tany = tanx * (double)s->height / (double)s->width;
    const double focal = s->width / (2.0 * tanx);
    Pcg32 rp(s->seed, 1), rs(s->seed, 2), rq(s->seed, 3), ro(s->seed, 4), rc(s->seed, 5), rh(s->seed, 6);
    const int ncoef_rest = ((s->sh_degree + 1) * (s->sh_degree + 1)) - 1;
    for (int i = 0; i < s->n; ++i) {
        const double z = 2.0 + 10.0 * rp.uniform();
        const double x = (2.0 * rp.uniform() - 1.0) * 1.1 * tanx * z;
        const double y = (2.0 * rp.uniform() - 1.0) * 1.1 * tany * z;
        pos[3 * i] = (float)x; pos[3 * i + 1] = (float)y; pos[3 * i + 2] = (float)z;
        const double mu = std::log(1.5 * z / focal) + (double)s->scale_log_offset;
        for (int k = 0; k < 3; ++k) scale[3 * i + k] = (float)(mu + 0.5 * rs.normal());
        for (int k = 0; k < 4; ++k) rot[4 * i + k] = (float)rq.normal();
        opacity[i] = (float)(1.5 * ro.normal());
        for (int k = 0; k < 3; ++k) sh0[3 * i + k] = (float)rc.normal();
        for (int j = 0; j < 15; ++j)
            for (int ch = 0; ch < 3; ++ch) {
                const double v = 0.1 * rh.normal();       // always drawn so the stream does not depend on the degree
                shN[45 * (size_t)i + j * 3 + ch] = j < ncoef_rest ? (float)v : 0.f;
            }
    }
    return DVS_OK;
}

int dvs_synth_target(const dvs_scene_spec* s, int index, float* target) {
    if (!s || !target) return DVS_ERR_INVALID;
    Pcg32 r(s->seed + 1, 100 + (uint64_t)index);
    const size_t cnt = 3 * (size_t)s->width * s->height;
    for (size_t k = 0; k < cnt; ++k) target[k] = (float)r.uniform();
    return DVS_OK;
}

}  // extern "C"
