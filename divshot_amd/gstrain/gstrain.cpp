// gstrain.cpp — MI355X-native `libgstrain.so`: the plugin DIVSHOT's hosts dlopen (PluginManager::ensure_plugin_loaded
// ("gstrain"), application/diverseshot-cli/source/gs_train.cpp:16-23 -> diverse/diverse_base/source/core/plugin.cpp:74,89)
// and drive through nine C symbols (gs_train.cpp:24,105-109,144-150,178) plus the two every plugin exports
// (plugin.cpp:89-111). The reference's implementation is closed source (README.md:46); this one keeps its call
// sequence and ownership rules (scene allocated/freed by the plugin, config copied, bool from load_train_data) and
// puts the MI355X rasterizer (include/dvs_raster.h) at the centre of train_step():
//     sample camera -> dvs_raster_forward -> (1-w) L1 + w (1-SSIM) loss gradient -> dvs_raster_backward -> fused Adam -> step++
// -> every refineEvery steps clone / split / prune (ADC) and every resetAlphaEvery steps the opacity reset.
// Out of scope this round: MCMC relocation, COLMAP / image ingestion,
// mesh export. load_train_data accepts a synthetic-scene spec instead of a dataset path.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <stdexcept>
#include "../../include/gaussian_trainer_scene.hpp"
#include "../../include/dvs_raster.h"
#include "../../include/dvs_scene.h"
#include "../../include/dvs_train.h"
#include "ply_io.hpp"

namespace {
const int kWidth[6] = {3, 3, 45, 1, 3, 4};          // pos sh0 shN opacity scale rot
enum { P_POS = 0, P_SH0, P_SHN, P_OPA, P_SCALE, P_ROT };

void logf_(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void logf_(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    fputs("[gstrain] ", stderr); vfprintf(stderr, fmt, ap); fputc('\n', stderr);
    va_end(ap);
}
#define HIP_OR_THROW(expr)                                                                          \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr ": ") + hipGetErrorString(e_)); } while (0)
#define DVS_OR_THROW(expr)                                                                          \
    do { int r_ = (expr); if (r_ != DVS_OK) throw std::runtime_error(std::string(#expr ": ") + dvs_last_error()); } while (0)

struct Lcg {       // tiny deterministic noise source for the synthetic initialisation
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ULL + 1442695040888963407ULL) {}
    float uni() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((s >> 40) * (1.0 / 16777216.0)); }
    float sym() { return 2.f * uni() - 1.f; }
};
}  // namespace

struct GaussianTrainerScene::Impl {
    GaussianTrainConfig cfg;
    int loadItr = -1;
    TrainingStatus status = TrainingStatus::Loading_Prepare;
    bool training = true;
    int device = 0;
    hipStream_t stream = nullptr;
    dvs_ctx* ctx = nullptr;
    int n = 0, W = 0, H = 0, sh_max = 3;
    float* d_param[6] = {}; float* d_grad[6] = {}; float* d_m[6] = {}; float* d_v[6] = {};
    float* d_param2[6] = {}; float* d_m2[6] = {}; float* d_v2[6] = {};       // densification writes old -> new, then the sets swap
    float* d_absgrad = nullptr;
    int cap = 0;                                                             // array capacity in splats (cfg.capMax)
    float* d_grad_accum = nullptr; float* d_denom = nullptr; int* d_max_radii = nullptr;
    uint8_t* d_action = nullptr; uint32_t* d_offsets = nullptr; uint32_t* d_dscratch = nullptr; uint64_t* d_newcount = nullptr;
    void* d_mcmc = nullptr;                                                  // dvs_mcmc_* scratch (densifyStrategy 1)
    float extent = 1.f;                                                      // scene extent (camera spread), sets the split/clone scale
    dvs_fwd_state fwd{};
    std::vector<dvs_camera> cams;
    std::vector<float*> d_targets;
    float* d_out = nullptr; float* d_dL = nullptr; float* d_loss = nullptr;     // d_loss[0..63] = (1-w) L1 partial sums, d_loss[64..127] = SSIM partial sums
    float* d_ssim_maps[3] = {nullptr, nullptr, nullptr};
    float last_loss = 0.f;
    int step = 0;
    uint64_t cam_rng = 88172645463325252ULL;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<float> host[6];
    bool host_valid = false;

    ~Impl() { release(); }
    void release() {
        if (device >= 0) (void)hipSetDevice(device);
        for (int g = 0; g < 6; ++g) {
            for (float** p : {&d_param[g], &d_grad[g], &d_m[g], &d_v[g], &d_param2[g], &d_m2[g], &d_v2[g]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        }
        for (void** p : {(void**)&d_grad_accum, (void**)&d_denom, (void**)&d_max_radii, (void**)&d_action, (void**)&d_offsets,
                         (void**)&d_dscratch, (void**)&d_newcount, &d_mcmc}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        for (float* t : d_targets) (void)hipFree(t);
        d_targets.clear();
        for (float** p : {&d_absgrad, &d_out, &d_dL, &d_loss, &d_ssim_maps[0], &d_ssim_maps[1], &d_ssim_maps[2]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (ctx) { dvs_destroy(ctx); ctx = nullptr; }
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    }
    // floats of group g on the device: the 45 higher-order SH floats live in the DVS_SHN_TILED layout (48 per splat,
    // whole 64-splat tiles); parameters, gradients and Adam moments share it — the optimizer is element-wise.
    size_t dev_floats_for(int g, int count) const { return g == P_SHN ? (size_t)((count + 63) / 64) * 64 * 48 : (size_t)count * kWidth[g]; }
    size_t dev_floats(int g) const { return dev_floats_for(g, n); }
    void upload(int g, const std::vector<float>& host_rows) {
        if (g != P_SHN) { HIP_OR_THROW(hipMemcpy(d_param[g], host_rows.data(), host_rows.size() * sizeof(float), hipMemcpyHostToDevice)); return; }
        float* tmp = nullptr;
        HIP_OR_THROW(hipMalloc((void**)&tmp, host_rows.size() * sizeof(float) + 4));
        HIP_OR_THROW(hipMemcpy(tmp, host_rows.data(), host_rows.size() * sizeof(float), hipMemcpyHostToDevice));
        DVS_OR_THROW(dvs_shn_relayout(ctx, stream, n, tmp, d_param[g], 1));
        HIP_OR_THROW(hipStreamSynchronize(stream));
        (void)hipFree(tmp);
    }
    void alloc_params(int count, int capacity, const std::vector<float> init[6]) {
        n = count; cap = std::max(capacity, count);
        for (int g = 0; g < 6; ++g) {
            const size_t bytes = dev_floats_for(g, cap) * sizeof(float);
            for (float** p : {&d_param[g], &d_grad[g], &d_m[g], &d_v[g], &d_param2[g], &d_m2[g], &d_v2[g]}) {
                HIP_OR_THROW(hipMalloc((void**)p, bytes ? bytes : 4));
                HIP_OR_THROW(hipMemset(*p, 0, bytes));         // pad lanes of the last tile are never written: keep them zero
            }
            upload(g, init[g]);
        }
        HIP_OR_THROW(hipMalloc((void**)&d_absgrad, (size_t)cap * 2 * sizeof(float) + 4));
        HIP_OR_THROW(hipMalloc((void**)&d_grad_accum, (size_t)cap * 4 + 4)); HIP_OR_THROW(hipMalloc((void**)&d_denom, (size_t)cap * 4 + 4));
        HIP_OR_THROW(hipMalloc((void**)&d_max_radii, (size_t)cap * 4 + 4)); HIP_OR_THROW(hipMalloc((void**)&d_action, (size_t)cap + 4));
        HIP_OR_THROW(hipMalloc((void**)&d_offsets, (size_t)cap * 4 + 4)); HIP_OR_THROW(hipMalloc((void**)&d_dscratch, ((size_t)cap / 256 + 8) * 4));
        HIP_OR_THROW(hipMalloc((void**)&d_newcount, 8));
        HIP_OR_THROW(hipMalloc(&d_mcmc, dvs_mcmc_scratch_bytes(cap)));
        DVS_OR_THROW(dvs_mcmc_init_scratch(stream, d_mcmc, cap));
        reset_stats();
    }
    void reset_stats() {
        HIP_OR_THROW(hipMemsetAsync(d_grad_accum, 0, (size_t)cap * 4, stream));
        HIP_OR_THROW(hipMemsetAsync(d_denom, 0, (size_t)cap * 4, stream));
        HIP_OR_THROW(hipMemsetAsync(d_max_radii, 0, (size_t)cap * 4, stream));
    }
    void densify(int it);
    void densify_mcmc(int it);
    bool mcmc() const { return cfg.densifyStrategy == 1; }
    dvs_splats splats() const {
        dvs_splats s{};
        s.pos = d_param[P_POS]; s.sh0 = d_param[P_SH0]; s.shN = d_param[P_SHN]; s.opacity = d_param[P_OPA];
        s.scale = d_param[P_SCALE]; s.rot = d_param[P_ROT]; s.n = n;
        return s;
    }
    std::string model_file(int it) const { return cfg.modelPath + "_" + std::to_string(it) + ".ply"; }
    void fetch_host() {
        if (host_valid) return;
        HIP_OR_THROW(hipStreamSynchronize(stream));
        for (int g = 0; g < 6; ++g) {
            host[g].resize((size_t)n * kWidth[g]);             // host copies are always in the reference layout (update_from_cpu)
            const float* src = d_param[g];
            float* tmp = nullptr;
            if (g == P_SHN) {
                HIP_OR_THROW(hipMalloc((void**)&tmp, host[g].size() * sizeof(float) + 4));
                DVS_OR_THROW(dvs_shn_relayout(ctx, stream, n, d_param[g], tmp, 0));
                HIP_OR_THROW(hipStreamSynchronize(stream));
                src = tmp;
            }
            HIP_OR_THROW(hipMemcpy(host[g].data(), src, host[g].size() * sizeof(float), hipMemcpyDeviceToHost));
            if (tmp) (void)hipFree(tmp);
        }
        host_valid = true;
    }
    bool load_synthetic(const std::string& spec_str);
};

bool GaussianTrainerScene::Impl::load_synthetic(const std::string& spec_str) {
    // "synthetic:N=100000,W=800,H=800,cams=8,sh=3,seed=1"
    std::map<std::string, double> kv = {{"N", 100000}, {"W", 800}, {"H", 800}, {"cams", 8}, {"sh", 3}, {"seed", 1}};
    size_t p = spec_str.find(':');
    std::string rest = p == std::string::npos ? "" : spec_str.substr(p + 1);
    while (!rest.empty()) {
        size_t c = rest.find(',');
        std::string item = rest.substr(0, c);
        rest = c == std::string::npos ? "" : rest.substr(c + 1);
        size_t e = item.find('=');
        if (e == std::string::npos) continue;
        kv[item.substr(0, e)] = atof(item.substr(e + 1).c_str());
    }
    dvs_scene_spec spec{};
    spec.n = (int)kv["N"]; spec.width = (int)kv["W"]; spec.height = (int)kv["H"]; spec.sh_degree = (int)kv["sh"];
    spec.n_cams = (int)kv["cams"]; spec.seed = (uint64_t)kv["seed"]; spec.fov_x_deg = 60.f; spec.scale_log_offset = 0.f;
    if (spec.n <= 0 || spec.width <= 0 || spec.height <= 0 || spec.n_cams <= 0 || spec.sh_degree < 0 || spec.sh_degree > 3) return false;
    if (spec.width > cfg.maxImageWidth || spec.height > cfg.maxImageHeight)
        logf_("note: synthetic image %dx%d exceeds maxImageWidth/Height %dx%d (kept as is)", spec.width, spec.height, cfg.maxImageWidth, cfg.maxImageHeight);
    W = spec.width; H = spec.height; sh_max = spec.sh_degree;
    std::vector<float> gt[6];
    for (int g = 0; g < 6; ++g) gt[g].resize((size_t)spec.n * kWidth[g]);
    DVS_OR_THROW(dvs_synth_splats(&spec, gt[0].data(), gt[1].data(), gt[2].data(), gt[3].data(), gt[4].data(), gt[5].data()));
    const int capacity = std::max(spec.n, cfg.capMax > 0 ? std::min(cfg.capMax, std::max(spec.n * 3, 4096)) : spec.n);
    ctx = dvs_create(device, (size_t)capacity, W, H);
    if (!ctx) throw std::runtime_error(std::string("dvs_create: ") + dvs_last_error());
    // ground-truth views: render the generating scene once per camera
    alloc_params(spec.n, capacity, gt);
    const size_t img = 3 * (size_t)W * H;
    HIP_OR_THROW(hipMalloc((void**)&d_out, img * sizeof(float)));
    HIP_OR_THROW(hipMalloc((void**)&d_dL, img * sizeof(float)));
    HIP_OR_THROW(hipMalloc((void**)&d_loss, 2 * DVS_SSIM_SLOTS * sizeof(float)));
    HIP_OR_THROW(hipMemset(d_loss, 0, 2 * DVS_SSIM_SLOTS * sizeof(float)));
    if (cfg.ssimWeight > 0.f)
        for (int k = 0; k < 3; ++k) HIP_OR_THROW(hipMalloc((void**)&d_ssim_maps[k], img * sizeof(float)));
    dvs_opts opts{sh_max, cfg.mipAntiliased ? 1 : 0, 0, 0, DVS_SHN_TILED};
    const dvs_splats sp = splats();
    for (int c = 0; c < spec.n_cams; ++c) {
        dvs_camera cam;
        DVS_OR_THROW(dvs_synth_camera(&spec, c, &cam));
        float* t = nullptr;
        HIP_OR_THROW(hipMalloc((void**)&t, img * sizeof(float)));
        DVS_OR_THROW(dvs_raster_forward(ctx, stream, &sp, &cam, &opts, t, nullptr, nullptr));
        cams.push_back(cam); d_targets.push_back(t);
    }
    HIP_OR_THROW(hipStreamSynchronize(stream));
    {   // scene extent = 1.1 x the largest distance of a camera centre from their mean (the usual "cameras_extent"); a single
        // camera or a tiny rig falls back to half the depth range of the synthetic slab
        double mean[3] = {0, 0, 0};
        for (auto& c : cams) for (int k = 0; k < 3; ++k) mean[k] += c.campos[k] / cams.size();
        double far = 0;
        for (auto& c : cams) { double d = 0; for (int k = 0; k < 3; ++k) d += (c.campos[k] - mean[k]) * (c.campos[k] - mean[k]); far = std::max(far, std::sqrt(d)); }
        extent = far > 1e-3 ? (float)(1.1 * far) : 5.0f;
    }
    // trainable initialisation = perturbed ground truth (or the checkpoint when --load_itr is given)
    std::vector<float> init[6];
    bool resumed = false;
    if (loadItr >= 0) {
        std::string err;
        resumed = gsply::read_ply(model_file(loadItr), init[0], init[1], init[2], init[3], init[4], init[5], &err) &&
                  !init[3].empty() && (int)init[3].size() <= cap;        // the count may differ from the spec after densification
        if (resumed) n = (int)init[3].size();
        if (!resumed) logf_("could not resume from %s (%s): starting from the synthetic initialisation", model_file(loadItr).c_str(), err.c_str());
        else step = loadItr;
    }
    if (!resumed) {
        Lcg r(spec.seed + 17);
        for (int g = 0; g < 6; ++g) init[g] = gt[g];
        for (int i = 0; i < spec.n; ++i) {
            const float z = gt[0][3 * i + 2];
            for (int k = 0; k < 3; ++k) init[P_POS][3 * i + k] += 0.002f * z * r.sym();
            for (int k = 0; k < 3; ++k) init[P_SH0][3 * i + k] += 0.5f * r.sym();
            for (int k = 0; k < 45; ++k) init[P_SHN][45 * (size_t)i + k] = 0.f;
            init[P_OPA][i] -= 1.0f;
            for (int k = 0; k < 3; ++k) init[P_SCALE][3 * i + k] += 0.15f * r.sym();
        }
    }
    for (int g = 0; g < 6; ++g) upload(g, init[g]);
    if (cfg.verbose) logf_("synthetic scene: %d splats, %d cameras @ %dx%d, SH degree %d%s", spec.n, spec.n_cams, W, H, sh_max, resumed ? " (resumed)" : "");
    return true;
}

// densifyStrategy 1 (MCMC): dead splats are relocated onto live ones drawn ~ opacity, then the model grows by 5 % up to the cap.
// In place: no second buffer set, no host round trip.
void GaussianTrainerScene::Impl::densify_mcmc(int it) {
    dvs_mcmc_sets sets{};
    for (int g = 0; g < 6; ++g) { sets.param[g] = d_param[g]; sets.m[g] = d_m[g]; sets.v[g] = d_v[g]; }
    DVS_OR_THROW(dvs_mcmc_relocate(stream, n, &sets, cfg.min_opacity, 2u * (uint32_t)it, DVS_SHN_TILED, d_mcmc, cap, nullptr));
    const int target = std::min(cap, (int)(1.05 * (double)n));
    const int n_new = target - n;
    if (n_new > 0) {
        DVS_OR_THROW(dvs_mcmc_grow(stream, n, n_new, &sets, cfg.min_opacity, 2u * (uint32_t)it + 1u, DVS_SHN_TILED, d_mcmc, cap));
        if (cfg.verbose) logf_("mcmc @%d: %d -> %d splats", it, n, n + n_new);
        n += n_new;
        HIP_OR_THROW(hipMemsetAsync(d_grad[P_SHN], 0, dev_floats_for(P_SHN, cap) * sizeof(float), stream));   // pad lanes of the new last tile
    }
    host_valid = false;
}

// clone / split / prune between two iterations (densifyStrategy 0 ADC; 2 "ADC+" is served by the same rule)
void GaussianTrainerScene::Impl::densify(int it) {
    dvs_densify_params prm{};
    prm.grad_threshold = cfg.growGrad2d;
    prm.scale_threshold = 0.01f * extent;                       // percent_dense x extent
    prm.min_opacity = cfg.min_opacity;
    const bool after_reset = it > cfg.resetAlphaEvery;
    prm.max_world_scale = after_reset ? cfg.pruneScale3d * extent : 0.f;                     // `pruneScale3d` (fraction of the scene extent)
    prm.max_screen_radius = after_reset && it < cfg.refineScale2dStopIter                    // `pruneScale2d` (fraction of the image size)
                                ? std::max(1, (int)(cfg.pruneScale2d * (float)std::max(W, H))) : 0;
    prm.cap_max = cap; prm.seed = (uint32_t)it; prm.shn_layout = DVS_SHN_TILED;
    prm.revised_opacity = cfg.revisedOpacity ? 1 : 0;
    uint64_t new_n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        DVS_OR_THROW(dvs_densify_plan(stream, n, d_param[P_OPA], d_param[P_SCALE], d_grad_accum, d_denom, d_max_radii, &prm, d_action,
                                      d_offsets, d_dscratch, d_newcount));
        HIP_OR_THROW(hipMemcpyAsync(&new_n, d_newcount, 8, hipMemcpyDeviceToHost, stream));
        HIP_OR_THROW(hipStreamSynchronize(stream));
        if (new_n <= (uint64_t)cap) break;
        prm.grad_threshold = 3.0e38f;                            // at the cap: prune only, no growth this round
    }
    if (new_n == 0 || new_n > (uint64_t)cap) { reset_stats(); return; }
    for (int set = 0; set < 3; ++set) {
        float** src = set == 0 ? d_param : (set == 1 ? d_m : d_v);
        float** dst = set == 0 ? d_param2 : (set == 1 ? d_m2 : d_v2);
        const float* s6[6] = {src[0], src[1], src[2], src[3], src[4], src[5]};
        if (set > 0)
            for (int g = 0; g < 6; ++g) HIP_OR_THROW(hipMemsetAsync(dst[g], 0, dev_floats_for(g, (int)new_n) * sizeof(float), stream));
        else HIP_OR_THROW(hipMemsetAsync(dst[P_SHN], 0, dev_floats_for(P_SHN, (int)new_n) * sizeof(float), stream));   // tile pads
        DVS_OR_THROW(dvs_densify_apply(stream, n, d_action, d_offsets, &prm, set == 0 ? 0 : 1, s6, dst, (int)new_n));
        for (int g = 0; g < 6; ++g) std::swap(src[g], dst[g]);
    }
    if (cfg.verbose) logf_("densify @%d: %d -> %llu splats", it, n, (unsigned long long)new_n);
    n = (int)new_n;
    HIP_OR_THROW(hipMemsetAsync(d_grad[P_SHN], 0, dev_floats_for(P_SHN, cap) * sizeof(float), stream));   // pad lanes of the new last tile
    reset_stats();
    host_valid = false;
}

GaussianTrainerScene::GaussianTrainerScene(const GaussianTrainConfig& cfg, int loadItr) : impl_(new Impl()) {
    impl_->cfg = cfg;
    impl_->loadItr = loadItr;
    const char* lr = getenv("LOCAL_RANK");
    impl_->device = lr ? atoi(lr) : 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw std::runtime_error("gstrain: no HIP device visible (this plugin has no CPU fallback)");
    impl_->device %= count;
    HIP_OR_THROW(hipSetDevice(impl_->device));
    HIP_OR_THROW(hipStreamCreate(&impl_->stream));
}
GaussianTrainerScene::~GaussianTrainerScene() = default;

bool GaussianTrainerScene::loadTrainData(const std::string& path) {
    Impl& m = *impl_;
    try {
        HIP_OR_THROW(hipSetDevice(m.device));
        if (path.rfind("synthetic", 0) == 0) {
            if (!m.load_synthetic(path)) { m.status = TrainingStatus::Loading_Failed; return false; }
            m.status = TrainingStatus::Preprocess_Done;
            trainSetup();
            return true;
        }
        logf_("load_train_data('%s'): dataset ingestion (COLMAP / images) is outside this build's scope; use a 'synthetic:N=..,W=..,H=..,cams=..,sh=..,seed=..' spec", path.c_str());
    } catch (const std::exception& e) {
        logf_("load_train_data failed: %s", e.what());
    }
    m.status = TrainingStatus::Loading_Failed;
    return false;
}

void GaussianTrainerScene::trainSetup() {
    impl_->t0 = std::chrono::steady_clock::now();
    impl_->status = TrainingStatus::Training;
    curIteration = impl_->step;
}

void GaussianTrainerScene::trainStep() {
    Impl& m = *impl_;
    if (!m.ctx || m.cams.empty()) throw std::runtime_error("trainStep before loadTrainData");
    HIP_OR_THROW(hipSetDevice(m.device));
    // camera: xorshift over the view list (one view per iteration, as the reference's trainStep renders one camera)
    m.cam_rng ^= m.cam_rng << 13; m.cam_rng ^= m.cam_rng >> 7; m.cam_rng ^= m.cam_rng << 17;
    const int ci = m.cfg.singleCamera ? 0 : (int)(m.cam_rng % m.cams.size());
    const int it = m.step + 1;
    const int deg = m.cfg.progressiveTrain ? std::min(m.sh_max, m.step / 1000) : m.sh_max;   // SH bands unlocked every 1000 steps
    dvs_opts opts{deg, m.cfg.mipAntiliased ? 1 : 0, m.cfg.useAbsGrad ? 1 : 0, 0, DVS_SHN_TILED};
    const dvs_splats sp = m.splats();
    DVS_OR_THROW(dvs_raster_forward(m.ctx, m.stream, &sp, &m.cams[ci], &opts, m.d_out, &m.fwd, nullptr));
    // photometric loss (1-w) L1 + w (1 - SSIM), w = --ssim (main.cpp:24-25); its gradient goes straight into d_dL
    const float w_ssim = m.d_ssim_maps[0] ? m.cfg.ssimWeight : 0.f;
    HIP_OR_THROW(hipMemsetAsync(m.d_loss, 0, 2 * DVS_SSIM_SLOTS * sizeof(float), m.stream));
    if (w_ssim > 0.f) {     // SSIM maps, then the L1 and SSIM gradients in one pass over the image
        DVS_OR_THROW(dvs_ssim_forward(m.stream, m.d_out, m.d_targets[ci], m.W, m.H, m.d_ssim_maps[0], m.d_ssim_maps[1], m.d_ssim_maps[2],
                                      m.d_loss + DVS_SSIM_SLOTS));
        DVS_OR_THROW(dvs_loss_l1_ssim_backward(m.stream, m.d_out, m.d_targets[ci], m.W, m.H, m.d_ssim_maps[0], m.d_ssim_maps[1],
                                               m.d_ssim_maps[2], w_ssim, m.d_dL, m.d_loss));
    } else {
        DVS_OR_THROW(dvs_l1_loss_grad_w(m.stream, m.d_out, m.d_targets[ci], 3 * (size_t)m.W * m.H, 1.f, m.d_dL, m.d_loss));
    }
    dvs_splat_grads g{};
    g.pos = m.d_grad[P_POS]; g.sh0 = m.d_grad[P_SH0]; g.shN = m.d_grad[P_SHN]; g.opacity = m.d_grad[P_OPA];
    g.scale = m.d_grad[P_SCALE]; g.rot = m.d_grad[P_ROT]; g.absgrad2d = m.cfg.useAbsGrad ? m.d_absgrad : nullptr; g.mean2d = nullptr;
    DVS_OR_THROW(dvs_raster_backward(m.ctx, m.stream, &sp, &m.cams[ci], &opts, m.d_dL, &g));
    const bool mcmc = m.mcmc();
    const bool refining = (mcmc || m.cfg.useAbsGrad) && it < m.cfg.refineStopIter;
    if (mcmc)          // opacity and scale regularisers of the MCMC strategy (0.01 each in the published rule)
        DVS_OR_THROW(dvs_mcmc_regularize(m.stream, m.n, m.d_param[P_OPA], m.d_param[P_SCALE], m.d_grad[P_OPA], m.d_grad[P_SCALE], 0.01f, 0.01f));
    if (refining && !mcmc)      // densification statistics of this view
        DVS_OR_THROW(dvs_densify_accumulate(m.stream, m.n, m.fwd.radii, m.d_absgrad, m.W, m.H, m.d_grad_accum, m.d_denom, m.d_max_radii));
    // Adam, per-group learning rates (names gs_train.cpp:52-57; position lr decays exponentially init -> final)
    const float t = std::min(1.0f, (float)m.step / (float)std::max(1, m.cfg.numIters));
    const float lr_pos = std::exp((1.f - t) * std::log(m.cfg.poslrInit) + t * std::log(m.cfg.poslrFinal));
    const float lr[6] = {lr_pos, m.cfg.featurelr, m.cfg.featurelr / 20.f, m.cfg.opacitylr, m.cfg.scalinglr, m.cfg.rotationlr};
    // one launch for the six groups; shN chunks above the active SH degree have g = m = v = 0 (Adam is the identity there)
    static const int width[6] = {3, 3, 45, 1, 3, 4};
    dvs_adam_group ag[6];
    for (int k = 0; k < 6; ++k) {
        ag[k] = dvs_adam_group{m.d_param[k], m.d_grad[k], m.d_m[k], m.d_v[k], (uint64_t)m.dev_floats(k), lr[k], width[k],
                               k == P_SHN ? DVS_SHN_TILED : DVS_SHN_ROWS, 0};
        if (k == P_SHN) ag[k].active_chunks = deg >= 3 ? 0 : (3 * ((deg + 1) * (deg + 1) - 1) + 3) / 4;
    }
    if (deg == 0) ag[P_SHN].count = 0;
    DVS_OR_THROW(dvs_adam_step_groups(m.stream, ag, 6, 0.9f, 0.999f, 1e-15f, it, m.cfg.visibleAdam ? m.fwd.radii : nullptr, m.n));
    if (mcmc && m.cfg.noiselr > 0.f)      // exploration noise, scaled by the position learning rate (`noiselr`, gs_train.cpp:97)
        DVS_OR_THROW(dvs_mcmc_add_noise(m.stream, m.n, m.d_param[P_POS], m.d_param[P_SCALE], m.d_param[P_ROT], m.d_param[P_OPA],
                                        m.cfg.noiselr * lr_pos, (uint32_t)it));
    if (refining && it > m.cfg.warmupLength && m.cfg.refineEvery > 0 && it % m.cfg.refineEvery == 0) { if (mcmc) m.densify_mcmc(it); else m.densify(it); }
    if (refining && !mcmc && m.cfg.resetAlphaEvery > 0 && it % m.cfg.resetAlphaEvery == 0)
        DVS_OR_THROW(dvs_reset_opacity(m.stream, m.n, m.d_param[P_OPA], 0.01f, m.d_m[P_OPA], m.d_v[P_OPA]));
    if (m.cfg.verbose && (m.step % 100 == 0))          // same line the editor logs (application/editor/source/editor.cpp:1554)
        logf_("Iteraions %d, loss : %f", m.step, (double)getCurrentLoss());
    m.step = it;
    curIteration = it;
    m.host_valid = false;
    if (m.step >= m.cfg.numIters) m.status = TrainingStatus::Training_Done;
}

void GaussianTrainerScene::saveGaussianModel() {
    Impl& m = *impl_;
    m.fetch_host();
    const std::string file = m.model_file(m.step);
    std::error_code ec;
    const auto parent = std::filesystem::path(file).parent_path();
    if (!parent.empty()) std::filesystem::create_directories(parent, ec);
    std::string err;
    if (!gsply::write_ply(file, (size_t)m.n, m.host[0].data(), m.host[1].data(), m.host[2].data(), m.host[3].data(), m.host[4].data(),
                          m.host[5].data(), m.cfg.mipAntiliased, &err))
        logf_("save_splat_model: %s", err.c_str());
    else if (m.cfg.verbose) logf_("saved %d splats to %s", m.n, file.c_str());
}
void GaussianTrainerScene::exportMesh(const std::string&) { logf_("export_mesh: mesh extraction is outside this build's scope"); }
bool GaussianTrainerScene::isTrain() const { return impl_->training; }
void GaussianTrainerScene::startTrain() { impl_->training = true; }
void GaussianTrainerScene::pauseTrain() { impl_->training = false; }
int GaussianTrainerScene::getCurrentIterations() const { return impl_->step; }
float GaussianTrainerScene::getCurrentLoss() {
    Impl& m = *impl_;
    if (m.d_loss && m.stream) {
        float h[2 * DVS_SSIM_SLOTS] = {0.f};
        (void)hipStreamSynchronize(m.stream);
        (void)hipMemcpy(h, m.d_loss, sizeof h, hipMemcpyDeviceToHost);
        const float w = m.d_ssim_maps[0] ? m.cfg.ssimWeight : 0.f;
        double l1 = 0, ssim_sum = 0;
        for (int k = 0; k < DVS_SSIM_SLOTS; ++k) { l1 += h[k]; ssim_sum += h[DVS_SSIM_SLOTS + k]; }
        m.last_loss = (float)l1 + (w > 0.f ? w * (1.f - (float)(ssim_sum / (3.0 * m.W * m.H))) : 0.f);
    }
    return m.last_loss;
}
int& GaussianTrainerScene::maxIteriaons() { return impl_->cfg.numIters; }
GaussianTrainConfig& GaussianTrainerScene::getTrainConfig() { return impl_->cfg; }
GaussianTrainerScene::TrainingStatus GaussianTrainerScene::getCurrentTrainingStatus() const { return impl_->status; }
void GaussianTrainerScene::setTrainingStatus(TrainingStatus s) { impl_->status = s; }
double GaussianTrainerScene::getTrainingElpasedTime() const {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - impl_->t0).count();
}
int GaussianTrainerScene::getNumGaussians() const { return impl_->n; }
int GaussianTrainerScene::getNumCameras() const { return (int)impl_->cams.size(); }
const std::vector<float>& GaussianTrainerScene::getGaussianPositionCpu() { impl_->fetch_host(); return impl_->host[P_POS]; }
const std::vector<float>& GaussianTrainerScene::getGaussianSH0Cpu() { impl_->fetch_host(); return impl_->host[P_SH0]; }
const std::vector<float>& GaussianTrainerScene::getGaussianSHNCpu() { impl_->fetch_host(); return impl_->host[P_SHN]; }
const std::vector<float>& GaussianTrainerScene::getGaussianOpcaitiesCpu() { impl_->fetch_host(); return impl_->host[P_OPA]; }
const std::vector<float>& GaussianTrainerScene::getGaussianScalingsCpu() { impl_->fetch_host(); return impl_->host[P_SCALE]; }
const std::vector<float>& GaussianTrainerScene::getGaussianRotationsCpu() { impl_->fetch_host(); return impl_->host[P_ROT]; }

// ---- C symbols -------------------------------------------------------------------------------------------
extern "C" {
__attribute__((visibility("default"))) void gstrain_init() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) logf_("gstrain_init: no HIP device visible");
    else logf_("gstrain_init: %d MI355X-class device(s), rasterizer '%s'", count, dvs_version());
}
__attribute__((visibility("default"))) void* create_splat(const GaussianTrainConfig& config, int loadItr) {
    try { return new GaussianTrainerScene(config, loadItr); }
    catch (const std::exception& e) { logf_("create_splat: %s", e.what()); return nullptr; }
}
__attribute__((visibility("default"))) bool load_train_data(GaussianTrainerScene* scene, const std::string& path) {
    return scene && scene->loadTrainData(path);
}
__attribute__((visibility("default"))) void train_step(GaussianTrainerScene* scene) { if (scene) scene->trainStep(); }
__attribute__((visibility("default"))) int get_cur_step(GaussianTrainerScene* scene) { return scene ? scene->getCurrentIterations() : 0; }
__attribute__((visibility("default"))) void save_splat_model(GaussianTrainerScene* scene) { if (scene) scene->saveGaussianModel(); }
__attribute__((visibility("default"))) void export_mesh(GaussianTrainerScene* scene) { if (scene) scene->exportMesh(""); }
__attribute__((visibility("default"))) void delete_splat(GaussianTrainerScene* scene) { delete scene; }
__attribute__((visibility("default"))) void gstrain_destroy() {}
__attribute__((visibility("default"))) const char* get_description() { return "gstrain: MI355X-native Gaussian-splat trainer (divshot_amd)"; }
__attribute__((visibility("default"))) void* create_instance() { return nullptr; }

// plain-C helpers so the PLY wire format can be tested from Python without a GPU (tests/test_ply.py)
__attribute__((visibility("default"))) int gstrain_write_ply(const char* path, uint64_t n, const float* pos, const float* sh0,
                                                              const float* shN, const float* opacity, const float* scale,
                                                              const float* rot, int antialiased) {
    std::string err;
    return gsply::write_ply(path, (size_t)n, pos, sh0, shN, opacity, scale, rot, antialiased != 0, &err) ? 0 : 1;
}
__attribute__((visibility("default"))) int64_t gstrain_read_ply(const char* path, float* pos, float* sh0, float* shN, float* opacity,
                                                                float* scale, float* rot, uint64_t capacity) {
    std::vector<float> a, b, c, d, e, f;
    std::string err;
    if (!gsply::read_ply(path, a, b, c, d, e, f, &err)) return -1;
    const uint64_t n = d.size();
    if (pos && n <= capacity) {
        memcpy(pos, a.data(), a.size() * 4); memcpy(sh0, b.data(), b.size() * 4); memcpy(shN, c.data(), c.size() * 4);
        memcpy(opacity, d.data(), d.size() * 4); memcpy(scale, e.data(), e.size() * 4); memcpy(rot, f.data(), f.size() * 4);
    }
    return (int64_t)n;
}
}
