// dvs_oracle_capi.cpp — C entry points of the CPU ORACLE for ctypes (test infrastructure only;
// see the header of dvs_oracle.hpp: PARITY UNPINNED, conventions cited there).
// Build: make -C oracle   ->  oracle/_build/libdvs_oracle.so
#include "dvs_oracle.hpp"
#include <string>

namespace {
struct Handle {
    int is_double;
    dvso::State<float> f;
    dvso::State<double> d;
};

template <class T>
void load_inputs(dvso::State<T>& S, int n, const void* pos, const void* sh0, const void* shN, const void* opacity,
                 const void* scale, const void* rot, const dvs_camera* cam, const dvs_opts* opts) {
    S.n = n; S.cam = *cam; S.opts = *opts;
    auto cp = [&](std::vector<T>& v, const void* p, size_t cnt) { v.assign((const T*)p, (const T*)p + cnt); };
    cp(S.pos, pos, 3 * (size_t)n); cp(S.sh0, sh0, 3 * (size_t)n); cp(S.shN, shN, 45 * (size_t)n);
    cp(S.opacity, opacity, n); cp(S.scale, scale, 3 * (size_t)n); cp(S.rot, rot, 4 * (size_t)n);
}

template <class T>
const void* get_array(dvso::State<T>& S, const std::string& name, uint64_t* count, int* elem_bytes) {
#define DVSO_ARR(nm, vec)                                                         \
    if (name == nm) { *count = (vec).size(); *elem_bytes = (int)sizeof((vec)[0]); return (vec).data(); }
    DVSO_ARR("radii", S.radii) DVSO_ARR("mean2d", S.mean2d) DVSO_ARR("depth", S.depth)
    DVSO_ARR("conic_opacity", S.conic_opacity) DVSO_ARR("rgb", S.rgb) DVSO_ARR("flags", S.flags)
    DVSO_ARR("tiles_touched", S.tiles_touched) DVSO_ARR("rect", S.rect) DVSO_ARR("depth_bits", S.depth_bits)
    DVSO_ARR("offsets", S.offsets) DVSO_ARR("keys", S.keys) DVSO_ARR("vals", S.vals) DVSO_ARR("ranges", S.ranges)
    DVSO_ARR("out_color", S.out_color) DVSO_ARR("final_T", S.final_T) DVSO_ARR("n_contrib", S.n_contrib)
    DVSO_ARR("fragile", S.fragile) DVSO_ARR("cap_fragile", S.cap_fragile) DVSO_ARR("take_masks", S.take_masks)
    DVSO_ARR("dL_dmean2d", S.dL_dmean2d) DVSO_ARR("dL_dconic_opacity", S.dL_dconic_opacity)
    DVSO_ARR("dL_drgb", S.dL_drgb) DVSO_ARR("absgrad", S.absgrad)
    DVSO_ARR("g_pos", S.g_pos) DVSO_ARR("g_sh0", S.g_sh0) DVSO_ARR("g_shN", S.g_shN)
    DVSO_ARR("g_opacity", S.g_opacity) DVSO_ARR("g_scale", S.g_scale) DVSO_ARR("g_rot", S.g_rot)
#undef DVSO_ARR
    *count = 0; *elem_bytes = 0;
    return nullptr;
}
}  // namespace

#include <omp.h>
extern "C" {

// number of OpenMP threads the oracle uses from now on (0 = all cores); used by the bench's cpu_baseline leg
void dvso_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }

void* dvso_create(int use_double) {
    Handle* h = new Handle();
    h->is_double = use_double;
    return h;
}
void dvso_destroy(void* hp) { delete (Handle*)hp; }

// Arrays are float (handle created with use_double=0) or double (use_double=1), host memory, A0 layout.
int dvso_forward(void* hp, int n, const void* pos, const void* sh0, const void* shN, const void* opacity,
                 const void* scale, const void* rot, const dvs_camera* cam, const dvs_opts* opts) {
    Handle* h = (Handle*)hp;
    auto run = [&](auto& S) -> int {
        load_inputs(S, n, pos, sh0, shN, opacity, scale, rot, cam, opts);
        dvso::preprocess_forward(S);
        if (S.forced_ranges.empty()) dvso::bin(S);
        else {
            // the lists of another run of the same scene (dvso_set_lists): no binning. Every listed splat must be visible here too
            // (a splat this precision culls has no projected record to composite): 3 = it is not, 4 = the ranges are not this image's
            if (S.forced_ranges.size() != 2 * (size_t)S.tiles_x * S.tiles_y) return 4;
            for (uint32_t id : S.forced_vals) if (id >= (uint32_t)S.n || S.radii[id] <= 0) return 3;
            S.vals = S.forced_vals; S.ranges = S.forced_ranges; S.keys.clear();
        }
        if (!S.replay.empty() && S.replay.size() != 4 * S.vals.size()) return 2;     // the recorded lists are not this scene's
        dvso::render_forward(S);
        return 0;
    };
    return h->is_double ? run(h->d) : run(h->f);
}

// From now on dvso_forward composites over the given tile lists (vals[count]: splat ids in list order; ranges[2 * tiles]: (start, end) per
// tile — the layout of dvso_array("vals") / ("ranges")) instead of binning the scene itself. count = n_ranges = 0 clears. Used by the parity
// tests to run the fp64 instantiation over the lists of the fp32 / HIP run when fp64 bins a splat differently.
int dvso_set_lists(void* hp, const uint32_t* vals, uint64_t count, const uint32_t* ranges, uint64_t n_ranges) {
    Handle* h = (Handle*)hp;
    h->f.forced_vals.assign(vals, vals + count); h->d.forced_vals.assign(vals, vals + count);
    h->f.forced_ranges.assign(ranges, ranges + n_ranges); h->d.forced_ranges.assign(ranges, ranges + n_ranges);
    return 0;
}

// Decision replay (parity tests): from now on render_forward / render_backward take the per-(list position, 8x8 quadrant) 64-bit masks
// of WHICH pixel takes WHICH entry instead of evaluating the thresholds — the masks the HIP forward recorded
// (dvs_debug_record_decisions). count = 4 * number of instances (0 clears). Returns 0.
int dvso_set_replay(void* hp, const uint64_t* masks, uint64_t count) {
    Handle* h = (Handle*)hp;
    h->f.replay.assign(masks, masks + count); h->d.replay.assign(masks, masks + count);
    return 0;
}

// From now on every forward also records the decisions it took itself ("take_masks", [T][4] uint64: same layout as the HIP hook).
void dvso_record_masks(void* hp, int on) {
    Handle* h = (Handle*)hp;
    h->f.record_masks = on != 0; h->d.record_masks = on != 0;
}

// dL_dout: [3,H,W] planar, float or double per the handle.
int dvso_backward(void* hp, const void* dL_dout) {
    Handle* h = (Handle*)hp;
    if (h->is_double) { dvso::render_backward(h->d, (const double*)dL_dout); dvso::preprocess_backward(h->d); }
    else { dvso::render_backward(h->f, (const float*)dL_dout); dvso::preprocess_backward(h->f); }
    return 0;
}

// which backward the two non-smooth points of the forward get from now on (dvs_opts.grad_mode: DVS_GRAD_TRUE / DVS_GRAD_LINEAGE)
void dvso_set_grad_mode(void* hp, int mode) {
    Handle* h = (Handle*)hp;
    h->f.opts.grad_mode = mode; h->d.opts.grad_mode = mode;
}

const void* dvso_array(void* hp, const char* name, uint64_t* count, int* elem_bytes) {
    Handle* h = (Handle*)hp;
    return h->is_double ? get_array(h->d, name, count, elem_bytes) : get_array(h->f, name, count, elem_bytes);
}

uint64_t dvso_interactions(void* hp) {
    Handle* h = (Handle*)hp;
    return h->is_double ? h->d.interactions : h->f.interactions;
}

// scalar helpers for the known-answer tests
float dvso_expf(float x) { return dvso::det_expf(x); }
float dvso_sigmoidf(float x) { return dvso::sigmoid<float>(x); }
void dvso_sh_basis(int deg, double x, double y, double z, double* b16) { dvso::sh_basis<double>(deg, x, y, z, b16); }
void dvso_cov3d(const double* scale3 /*already activated*/, const double* quat_wxyz_unit, double* cov6) {
    double R[9];
    dvso::quat_to_rot<double>(quat_wxyz_unit[0], quat_wxyz_unit[1], quat_wxyz_unit[2], quat_wxyz_unit[3], R);
    dvso::cov3d_from_scale_rot<double>(scale3, R, cov6);
}

}  // extern "C"
