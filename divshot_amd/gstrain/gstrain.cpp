// gstrain.cpp — MI355X-native `libgstrain.so`: the plugin DIVSHOT's hosts dlopen (PluginManager::ensure_plugin_loaded
// ("gstrain"), application/diverseshot-cli/source/gs_train.cpp:16-23 -> diverse/diverse_base/source/core/plugin.cpp:74,89)
// and drive through nine C symbols (gs_train.cpp:24,105-109,144-150,178) plus the two every plugin exports
// (plugin.cpp:89-111). The reference's implementation is closed source (README.md:46); this one keeps its call
// sequence and ownership rules (scene allocated/freed by the plugin, config copied, bool from load_train_data) and
// puts the MI355X rasterizer (include/dvs_raster.h) at the centre of train_step():
//     sample camera -> dvs_raster_forward -> (1-w) L1 + w (1-SSIM) loss gradient -> dvs_raster_backward -> fused Adam -> step++
// -> every refineEvery steps the densification strategy (0 ADC clone / split / prune, 1 MCMC relocation + growth, 2 ADC+), the
// opacity reset every resetAlphaEvery steps, a light prune pass every pruneInterval steps after refinement has stopped.
// With WORLD_SIZE > 1 (one process per GPU, launched with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) the step is
// data parallel: every rank holds a replica, renders its own camera of the iteration, the gradient rows are summed
// with ONE RCCL all-reduce over xGMI (include/dvs_comm.h), the densification statistics likewise before each refinement, and the
// optimizer / densifier run replicated and deterministic so that the replicas stay bit-identical.
// Out of scope: COLMAP / image ingestion, mesh export, the 2DGS model type. load_train_data accepts a
// synthetic-scene spec instead of a dataset path. Every GaussianTrainConfig field the hosts set is either honoured
// or named in the one-time "ignored" line of report_config().
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
#include "../../include/dvs_comm.h"
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

// training images kept as 8 bits per channel when packLevel has PackF32ToU8 (gs_train.cpp:91-96: the reference's VRAM saver):
// a quarter of the HBM footprint per view, expanded into one fp32 staging image right before the loss
__global__ void k_pack_u8(const float* __restrict__ src, uint8_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint8_t)fminf(255.f, fmaxf(0.f, rintf(src[i] * 255.f)));
}
__global__ void k_unpack_u8(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i] * (1.0f / 255.0f);
}
// useMask (main.cpp:69-70): pixels outside the mask carry no loss gradient; mask [H*W] in {0,1}, dL planar [3,H,W]
__global__ void k_mask_mul(float* __restrict__ dL, const float* __restrict__ mask, size_t P) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * P) dL[i] *= mask[i % P];
}
__global__ void k_norm2(const float* __restrict__ v2, float* __restrict__ out2, int n) {      // (x, y) -> (|v|, 0)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float x = v2[2 * i], y = v2[2 * i + 1]; out2[2 * i] = sqrtf(x * x + y * y); out2[2 * i + 1] = 0.f; }
}

struct Lcg {       // tiny deterministic noise source for the synthetic initialisation
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ULL + 1442695040888963407ULL) {}
    float uni() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((s >> 40) * (1.0 / 16777216.0)); }
    float sym() { return 2.f * uni() - 1.f; }
};
}  // namespace

struct __attribute__((visibility("hidden"))) GaussianTrainerScene::Impl {     // the class is exported (GSTRAIN_API), its implementation is not
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
    float* d_grad_flat = nullptr; size_t grad_floats = 0;                    // the six gradient groups live in ONE buffer: one all-reduce
    float* d_mean2d = nullptr;                                               // dL/dmean2D (ADC statistic when useAbsGrad is off)
    dvs_comm* comm = nullptr; int rank = 0, world = 1;                       // data-parallel replicas (include/dvs_comm.h)
    // Gradient exchange of the replicas. factorised (default): the geometry groups (pos, opacity, scale, rot: 44 B/splat) lead the
    // flat buffer and are all-reduced; of the SH rows only each view's 3-float colour gradient is all-gathered (12 B/splat/view) and
    // every replica rebuilds the summed rows with dvs_sh_grad_combine — at 8 GPUs 161 MB instead of 413 MB through each GPU's links
    // per million splats. DVS_EXCHANGE=allreduce: one all-reduce of all 59-float rows.
    bool factorised = true; size_t geom_floats = 0;
    float* d_dcolor_local = nullptr; float* d_dcolor_all = nullptr;          // [cap,3] / [world,n,3]
    float* d_dcolor_scratch = nullptr;                                       // A9's own copy of the colour gradient (the all-gather reads the early one)
    hipStream_t comm_stream = nullptr; hipEvent_t ev_dcolor = nullptr, ev_bwd = nullptr, ev_comm = nullptr;   // collectives run beside A9
    hipEvent_t ev_gather = nullptr; std::vector<hipEvent_t> ev_chunk;        // all-gather done / A9 chunk k queued (chunked geometry all-reduce)
    bool pipeline = false;                                                   // DVS_EXCHANGE_PIPELINE=1 (with DVS_A9_CHUNKS > 1): see trainStep
    std::vector<hipEvent_t> ev_ar;                                           // chunk k's geometry all-reduce has landed (communication stream)
    std::vector<int> next_ci;                                                // the NEXT iteration's cameras, drawn early for the pipelined step
    int a9_chunks = 1;                                                       // DVS_A9_CHUNKS: splat chunks of A9 whose geometry gradients leave one by one (default 1 until
                                                                             // the chunked exchange has run on real multi-GPU hardware: ADVICE r03; equality with the unchunked
                                                                             // exchange is asserted by tests/test_gpu_multirank.py::test_plugin_two_ranks_exchanges_agree)
    std::vector<uint8_t*> d_targets_u8; float* d_target_f32 = nullptr;       // packLevel & PackF32ToU8
    std::vector<float*> d_masks;                                             // useMask
    std::vector<float> init_host[6];                                         // initial splats (resetGaussian, getPoints3D)
    bool terminate = false, pruning = false;
    int cap = 0;                                                             // array capacity in splats (cfg.capMax)
    float* d_grad_accum = nullptr; float* d_denom = nullptr; int* d_max_radii = nullptr;
    uint8_t* d_action = nullptr; uint32_t* d_offsets = nullptr; uint32_t* d_dscratch = nullptr; uint64_t* d_newcount = nullptr;
    void* d_mcmc = nullptr;                                                  // dvs_mcmc_* scratch (densifyStrategy 1)
    float extent = 1.f;                                                      // scene extent (camera spread), sets the split/clone scale
    dvs_fwd_state fwd{};
    int vpi = 1;                                                             // views per trainStep and GPU, one multi-view pass (cfg.viewsPerIter / DVS_VIEWS_PER_ITER)
    bool sequential_views = false;                                           // DVS_VIEWS_MODE=sequential: the same views one pass at a time, accumulating (the reference shape)
    int* d_vis_radius = nullptr;                                             // visibleAdam with vpi > 1: max radius over the step's views
    int loss_views = 1;                                                      // views whose loss sums d_loss holds
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
            for (float** p : {&d_param[g], &d_m[g], &d_v[g], &d_param2[g], &d_m2[g], &d_v2[g]}) { if (*p) (void)hipFree(*p); *p = nullptr; }
            d_grad[g] = nullptr;                                             // slices of d_grad_flat
        }
        for (uint8_t* t : d_targets_u8) (void)hipFree(t);
        d_targets_u8.clear();
        for (float* t : d_masks) (void)hipFree(t);
        d_masks.clear();
        for (float** p : {&d_grad_flat, &d_mean2d, &d_target_f32, &d_dcolor_local, &d_dcolor_all, &d_dcolor_scratch}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (d_vis_radius) { (void)hipFree(d_vis_radius); d_vis_radius = nullptr; }
        for (hipEvent_t* e : {&ev_dcolor, &ev_bwd, &ev_comm, &ev_gather}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        for (hipEvent_t e : ev_chunk) (void)hipEventDestroy(e);
        ev_chunk.clear();
        for (hipEvent_t e : ev_ar) (void)hipEventDestroy(e);
        ev_ar.clear();
        if (comm_stream) { (void)hipStreamDestroy(comm_stream); comm_stream = nullptr; }
        if (comm) { dvs_comm_destroy(comm); comm = nullptr; }
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
        grad_floats = 0;
        size_t goff[6];
        static const int order[6] = {P_POS, P_OPA, P_SCALE, P_ROT, P_SH0, P_SHN};       // geometry first: one contiguous all-reduce
        for (int k = 0; k < 6; ++k) {
            const int g = order[k];
            if (g == P_SH0) geom_floats = grad_floats;
            goff[g] = grad_floats; grad_floats += (dev_floats_for(g, cap) + 3) & ~(size_t)3;                            // 16-B aligned groups
        }
        HIP_OR_THROW(hipMalloc((void**)&d_grad_flat, grad_floats * sizeof(float) + 16));
        HIP_OR_THROW(hipMemset(d_grad_flat, 0, grad_floats * sizeof(float)));
        for (int g = 0; g < 6; ++g) d_grad[g] = d_grad_flat + goff[g];
        HIP_OR_THROW(hipMalloc((void**)&d_mean2d, (size_t)cap * 2 * sizeof(float) + 4));
        for (int g = 0; g < 6; ++g) {
            const size_t bytes = dev_floats_for(g, cap) * sizeof(float);
            for (float** p : {&d_param[g], &d_m[g], &d_v[g], &d_param2[g], &d_m2[g], &d_v2[g]}) {
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
    void prune_light(int it);
    void report_config() const;
    void sync_stats();
    const float* target_for(int ci);
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

// the fp32 target image of camera ci on the device (expands the 8-bit copy when packLevel has PackF32ToU8)
const float* GaussianTrainerScene::Impl::target_for(int ci) {
    if (!(cfg.packLevel & PackF32ToU8)) return d_targets[ci];
    const size_t img = 3 * (size_t)W * H;
    hipLaunchKernelGGL(k_unpack_u8, dim3((unsigned)((img + 255) / 256)), dim3(256), 0, stream, d_targets_u8[ci], d_target_f32, img);
    return d_target_f32;
}

// One line per decision: which GaussianTrainConfig fields this build honours and which it ignores (gs_train.cpp:50-103 sets them all).
void GaussianTrainerScene::Impl::report_config() const {
    if (rank != 0) return;
    static const char* strat[3] = {"ADC (clone / split / prune)", "MCMC (relocation + growth)", "ADC+ (ADC on abs-grad statistics with revised opacity)"};
    logf_("config: densifyStrategy %d = %s; pruneStrategy %d (%s) every %d steps after refineStopIter %d; capMax %d; packLevel %d (%s%s); "
          "useMask %d; useAbsGrad %d; mipAntiliased %d; visibleAdam %d; singleCamera %d; progressiveTrain %d; world %d",
          cfg.densifyStrategy, strat[std::min(2, std::max(0, cfg.densifyStrategy))], cfg.pruneStrategy,
          cfg.pruneStrategy > 0 ? "light prune: opacity < pruneOpacity or scale > pruneScale3d" : "off", cfg.pruneInterval, cfg.refineStopIter,
          cfg.capMax, cfg.packLevel, (cfg.packLevel & PackF32ToU8) ? "PackF32ToU8: 8-bit training views" : "fp32 training views",
          (cfg.packLevel & PackTileID) ? ", PackTileID: always on here (the tile sort's keys are tile ids inside a view, written as 16-bit words while a view has <= 65536 tiles)" : "",
          (int)cfg.useMask, (int)cfg.useAbsGrad, (int)cfg.mipAntiliased, (int)cfg.visibleAdam, (int)cfg.singleCamera, (int)cfg.progressiveTrain, world);
    std::string ign;
    if (cfg.modelType != 0) ign += " modelType(only 3DGS)";
    if (cfg.cullSH) ign += " cullSH";
    if (cfg.pixelGradScale) ign += " pixelGradScale";
    if (cfg.bestQuality) ign += " bestQuality";
    if (cfg.normalConsistencyLoss) ign += " normalConsistencyLoss(2DGS)";
    if (cfg.enableBg) ign += " enableBg";
    if (cfg.enableFocusRegion) ign += " enableFocusRegion";
    if (cfg.exportMesh) ign += " exportMesh";
    if (cfg.outputSparsePoints) ign += " outputSparsePoints";
    if (cfg.resolutionSchedule) ign += " resolutionSchedule";
    if (cfg.maxImageCount) ign += " maxImageCount";
    if (!cfg.cameraPosePath.empty() || !cfg.pointCloudPath.empty()) ign += " cameraPosePath/pointCloudPath(dataset ingestion)";
    if (cfg.visibleAdam && world > 1) ign += " visibleAdam(off with WORLD_SIZE > 1: the visible set differs per rank)";
    if (!ign.empty()) logf_("config: IGNORED by this build:%s", ign.c_str());
}

// data parallel: the densification statistics are per-view sums / maxima — make them global before a refinement decision
void GaussianTrainerScene::Impl::sync_stats() {
    if (!comm) return;
    DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(comm, stream, d_grad_accum, (size_t)n));
    DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(comm, stream, d_denom, (size_t)n));
    DVS_OR_THROW(dvs_comm_all_reduce_max_i32(comm, stream, d_max_radii, (size_t)n));
}

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
    const int capacity = std::max(spec.n, cfg.capMax);          // --capMax is the array capacity (gs_train.cpp:89); 1.9 KB of HBM per splat
    vpi = cfg.viewsPerIter;
    if (const char* e = getenv("DVS_VIEWS_PER_ITER")) vpi = atoi(e);
    vpi = std::max(1, std::min(vpi, 16));
    if (const char* e = getenv("DVS_VIEWS_MODE")) sequential_views = std::string(e) == "sequential";
    if (vpi > 1 && rank == 0)
        logf_("config: %d views per trainStep and GPU, %s", vpi, sequential_views ? "one pass per view, gradients accumulated (DVS_VIEWS_MODE=sequential)"
                                                                                : "ONE multi-view pass (dvs_raster_forward_views / _backward_views), gradients summed");
    ctx = dvs_create_views(device, (size_t)capacity, W, H, sequential_views ? 1 : vpi);
    if (!ctx) throw std::runtime_error(std::string("dvs_create_views: ") + dvs_last_error());
    // ground-truth views: render the generating scene once per camera
    alloc_params(spec.n, capacity, gt);
    const size_t img = 3 * (size_t)W * H;
    HIP_OR_THROW(hipMalloc((void**)&d_out, (size_t)vpi * img * sizeof(float)));
    HIP_OR_THROW(hipMalloc((void**)&d_dL, (size_t)vpi * img * sizeof(float)));
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
        cams.push_back(cam);
        if (cfg.packLevel & PackF32ToU8) {                     // keep the view as 8 bits per channel, expand per step (target_for)
            uint8_t* t8 = nullptr;
            HIP_OR_THROW(hipMalloc((void**)&t8, img));
            hipLaunchKernelGGL(k_pack_u8, dim3((unsigned)((img + 255) / 256)), dim3(256), 0, stream, t, t8, img);
            HIP_OR_THROW(hipStreamSynchronize(stream));
            (void)hipFree(t);
            d_targets_u8.push_back(t8); d_targets.push_back(nullptr);
        } else {
            d_targets.push_back(t);
        }
        if (cfg.useMask) {                                     // synthetic mask: an ellipse inscribed in the image (no dataset masks here)
            std::vector<float> mk((size_t)W * H);
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
                const float u = (x + 0.5f) / W * 2.f - 1.f, v = (y + 0.5f) / H * 2.f - 1.f;
                mk[(size_t)y * W + x] = (u * u + v * v <= 1.f) ? 1.f : 0.f;
            }
            float* dm = nullptr;
            HIP_OR_THROW(hipMalloc((void**)&dm, mk.size() * sizeof(float)));
            HIP_OR_THROW(hipMemcpy(dm, mk.data(), mk.size() * sizeof(float), hipMemcpyHostToDevice));
            d_masks.push_back(dm);
        }
    }
    if (cfg.packLevel & PackF32ToU8) HIP_OR_THROW(hipMalloc((void**)&d_target_f32, img * sizeof(float)));
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
    for (int g = 0; g < 6; ++g) { upload(g, init[g]); init_host[g] = init[g]; }
    report_config();
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
    sync_stats();
    dvs_densify_params prm{};
    prm.grad_threshold = cfg.growGrad2d;
    prm.scale_threshold = 0.01f * extent;                       // percent_dense x extent
    prm.min_opacity = cfg.min_opacity;
    const bool after_reset = it > cfg.resetAlphaEvery;
    prm.max_world_scale = after_reset ? cfg.pruneScale3d * extent : 0.f;                     // `pruneScale3d` (fraction of the scene extent)
    prm.max_screen_radius = after_reset && it < cfg.refineScale2dStopIter                    // `pruneScale2d` (fraction of the image size)
                                ? std::max(1, (int)(cfg.pruneScale2d * (float)std::max(W, H))) : 0;
    prm.cap_max = cap; prm.seed = (uint32_t)it; prm.shn_layout = DVS_SHN_TILED;
    prm.revised_opacity = (cfg.revisedOpacity || cfg.densifyStrategy == 2) ? 1 : 0;      // ADC+ always uses the revised opacity of the copies
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
    if (cfg.verbose) {
        logf_("densify @%d: %d -> %llu splats", it, n, (unsigned long long)new_n);
        uint64_t cap_ = 0, grows_ = 0, lastT_ = 0, over_ = 0;        // the rasterizer's instance arena at this point (HBM pressure of big scenes)
        if (dvs_get_arena_info(ctx, &cap_, &grows_, &lastT_, &over_) == DVS_OK)
            logf_("raster @%d: T = %llu tile instances in the last pass, instance arena %llu (enlarged %llu times), overflowed forwards %llu",
                  it, (unsigned long long)lastT_, (unsigned long long)cap_, (unsigned long long)grows_, (unsigned long long)over_);
    }
    n = (int)new_n;
    HIP_OR_THROW(hipMemsetAsync(d_grad[P_SHN], 0, dev_floats_for(P_SHN, cap) * sizeof(float), stream));   // pad lanes of the new last tile
    reset_stats();
    host_valid = false;
}

// pruneStrategy > 0 ("Light Gaussian Prune" in the reference's log, screenshots/cli_example.png): after refinement has stopped, every
// pruneInterval steps splats that became transparent (opacity < pruneOpacity) or oversized (scale > pruneScale3d x extent) are removed
// and the arrays compacted; no growth. Runs replicated (deterministic) on every rank.
void GaussianTrainerScene::Impl::prune_light(int it) {
    pruning = true;
    dvs_densify_params prm{};
    prm.grad_threshold = 3.0e38f;                               // never clone / split
    prm.scale_threshold = 0.01f * extent;
    prm.min_opacity = std::max(cfg.pruneOpacity, cfg.min_opacity);       // --minOpacity is the only opacity threshold the CLI exposes
    prm.max_world_scale = cfg.pruneScale3d * extent;
    prm.max_screen_radius = 0;
    prm.cap_max = cap; prm.seed = (uint32_t)it; prm.shn_layout = DVS_SHN_TILED; prm.revised_opacity = 0;
    HIP_OR_THROW(hipMemsetAsync(d_grad_accum, 0, (size_t)cap * 4, stream));
    HIP_OR_THROW(hipMemsetAsync(d_denom, 0, (size_t)cap * 4, stream));
    HIP_OR_THROW(hipMemsetAsync(d_max_radii, 0, (size_t)cap * 4, stream));
    uint64_t new_n = 0;
    DVS_OR_THROW(dvs_densify_plan(stream, n, d_param[P_OPA], d_param[P_SCALE], d_grad_accum, d_denom, d_max_radii, &prm, d_action,
                                  d_offsets, d_dscratch, d_newcount));
    HIP_OR_THROW(hipMemcpyAsync(&new_n, d_newcount, 8, hipMemcpyDeviceToHost, stream));
    HIP_OR_THROW(hipStreamSynchronize(stream));
    if (new_n > 0 && new_n < (uint64_t)n) {
        for (int set = 0; set < 3; ++set) {
            float** src = set == 0 ? d_param : (set == 1 ? d_m : d_v);
            float** dst = set == 0 ? d_param2 : (set == 1 ? d_m2 : d_v2);
            const float* s6[6] = {src[0], src[1], src[2], src[3], src[4], src[5]};
            HIP_OR_THROW(hipMemsetAsync(dst[P_SHN], 0, dev_floats_for(P_SHN, (int)new_n) * sizeof(float), stream));
            DVS_OR_THROW(dvs_densify_apply(stream, n, d_action, d_offsets, &prm, set == 0 ? 0 : 1, s6, dst, (int)new_n));
            for (int g = 0; g < 6; ++g) std::swap(src[g], dst[g]);
        }
        if (cfg.verbose && rank == 0) logf_("light prune @%d: %d -> %llu splats", it, n, (unsigned long long)new_n);
        n = (int)new_n;
        HIP_OR_THROW(hipMemsetAsync(d_grad_flat, 0, grad_floats * sizeof(float), stream));
        host_valid = false;
    }
    pruning = false;
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
    // data parallel when launched as one process per GPU (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); DVS_FORCE_COMM=1 runs the
    // collectives on a 1-rank communicator too (single-GPU proof of the RCCL path)
    const char* ws = getenv("WORLD_SIZE"); const char* rk = getenv("RANK"); const char* force = getenv("DVS_FORCE_COMM");
    const int world = ws ? atoi(ws) : 1;
    if (world > 1 || (force && force[0] == '1')) {
        impl_->comm = dvs_comm_create(impl_->device, rk ? atoi(rk) : 0, std::max(1, world), nullptr, 0);
        if (!impl_->comm) throw std::runtime_error(std::string("gstrain: dvs_comm_create failed: ") + dvs_last_error());
        impl_->rank = dvs_comm_rank(impl_->comm); impl_->world = dvs_comm_world(impl_->comm);
        if (const char* ex = getenv("DVS_EXCHANGE")) impl_->factorised = std::string(ex) != "allreduce";
        const char* be = getenv("DVS_COMM_BACKEND");
        logf_("rank %d of %d on device %d: %s communicator up, gradient exchange: %s", impl_->rank, impl_->world, impl_->device,
              be && std::string(be) == "tcp" ? "host-staged TCP (TEST backend)" : "RCCL",
              impl_->factorised ? "factorised (all-gather of colour gradients + all-reduce of 44 B/splat)" : "all-reduce of all rows");
    }
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
    // cameras: one xorshift stream shared by all ranks; an iteration draws world x vpi views and rank r renders views r vpi .. r vpi + vpi - 1
    // (vpi = 1: one view per GPU and iteration, as the reference's trainStep renders one camera; vpi = 8 on one GPU: BASELINE config C4)
    const int V = m.vpi;
    std::vector<int> ci_all((size_t)m.world * V, 0);                // every rank knows every rank's cameras: the SH rows are rebuilt from them
    auto draw_cameras = [&](std::vector<int>& ci) {
        for (size_t k = 0; k < ci.size(); ++k) {
            m.cam_rng ^= m.cam_rng << 13; m.cam_rng ^= m.cam_rng >> 7; m.cam_rng ^= m.cam_rng << 17;
            ci[k] = m.cfg.singleCamera ? 0 : (int)(m.cam_rng % m.cams.size());
        }
    };
    if (m.next_ci.size() == ci_all.size()) { ci_all = m.next_ci; m.next_ci.clear(); }      // (drawn by the previous, pipelined step: the same stream)
    else draw_cameras(ci_all);
    const int* ci_mine = &ci_all[(size_t)m.rank * V];
    std::vector<dvs_camera> vcams((size_t)V);
    for (int v = 0; v < V; ++v) vcams[(size_t)v] = m.cams[(size_t)ci_mine[v]];
    const int it = m.step + 1;
    const int deg = m.cfg.progressiveTrain ? std::min(m.sh_max, m.step / 1000) : m.sh_max;   // SH bands unlocked every 1000 steps
    const bool mcmc = m.mcmc();
    const bool adc_plus = m.cfg.densifyStrategy == 2;
    const bool absgrad = m.cfg.useAbsGrad || adc_plus;                                        // ADC+ always splits on the abs-grad statistic
    const bool refining = it < m.cfg.refineStopIter;
    dvs_opts opts{};
    opts.sh_degree = deg; opts.antialias = m.cfg.mipAntiliased ? 1 : 0; opts.absgrad = absgrad ? 1 : 0; opts.accumulate = 0;
    opts.shn_layout = DVS_SHN_TILED;
    opts.grad_mode = DVS_GRAD_LINEAGE;          // the backward of the lineage the reference credits (README.md:95)
    static const bool tight_tiles = [] { const char* e = getenv("DVS_TIGHT_TILES"); return e && e[0] == '1'; }();
    opts.tile_bounds = tight_tiles ? DVS_TILES_TIGHT : DVS_TILES_CANONICAL;     // opt-in: same images and gradients, shorter tile lists (dvs_raster.h)
    const dvs_splats sp = m.splats();
    const size_t img = 3 * (size_t)m.W * m.H;
    const float w_ssim = m.d_ssim_maps[0] ? m.cfg.ssimWeight : 0.f;
    // photometric loss (1-w) L1 + w (1 - SSIM), w = --ssim (main.cpp:24-25), of view v: its gradient goes straight into d_dL[v]; the loss
    // sums of the step's views add up in d_loss (getCurrentLoss reports their mean)
    auto loss_of_view = [&](int v) {
        const float* target = m.target_for(ci_mine[v]);
        const float* out = m.d_out + (size_t)v * img;
        float* dL = m.d_dL + (size_t)v * img;
        if (w_ssim > 0.f) {     // SSIM maps, then the L1 and SSIM gradients in one pass over the image
            DVS_OR_THROW(dvs_ssim_forward(m.stream, out, target, m.W, m.H, m.d_ssim_maps[0], m.d_ssim_maps[1], m.d_ssim_maps[2],
                                          m.d_loss + DVS_SSIM_SLOTS));
            DVS_OR_THROW(dvs_loss_l1_ssim_backward(m.stream, out, target, m.W, m.H, m.d_ssim_maps[0], m.d_ssim_maps[1],
                                                   m.d_ssim_maps[2], w_ssim, dL, m.d_loss));
        } else {
            DVS_OR_THROW(dvs_l1_loss_grad_w(m.stream, out, target, img, 1.f, dL, m.d_loss));
        }
        if (m.cfg.useMask && !m.d_masks.empty()) {
            const size_t P = (size_t)m.W * m.H;
            hipLaunchKernelGGL(k_mask_mul, dim3((unsigned)((3 * P + 255) / 256)), dim3(256), 0, m.stream, dL, m.d_masks[(size_t)ci_mine[v]], P);
        }
    };
    HIP_OR_THROW(hipMemsetAsync(m.d_loss, 0, 2 * DVS_SSIM_SLOTS * sizeof(float), m.stream));
    m.loss_views = V;
    dvs_splat_grads g{};
    g.pos = m.d_grad[P_POS]; g.sh0 = m.d_grad[P_SH0]; g.shN = m.d_grad[P_SHN]; g.opacity = m.d_grad[P_OPA];
    g.scale = m.d_grad[P_SCALE]; g.rot = m.d_grad[P_ROT]; g.absgrad2d = absgrad ? m.d_absgrad : nullptr;
    g.mean2d = (!absgrad && !mcmc && refining) ? m.d_mean2d : nullptr;      // ADC without abs-grad: the norm of dL/dmean2D is the statistic
    const bool fact = m.comm && m.factorised;
    if (fact && !m.d_dcolor_local) {
        // factorised exchange: the SH rows are not written by the backward, only each view's colour gradient, which leaves right after
        // the composite backward (dvs_raster_backward_dcolor) so that its all-gather runs on the communication stream while A9 computes
        HIP_OR_THROW(hipMalloc((void**)&m.d_dcolor_local, (size_t)V * m.cap * 3 * sizeof(float) + 16));
        HIP_OR_THROW(hipMalloc((void**)&m.d_dcolor_scratch, (size_t)V * m.cap * 3 * sizeof(float) + 16));
        HIP_OR_THROW(hipMalloc((void**)&m.d_dcolor_all, (size_t)m.world * V * m.cap * 3 * sizeof(float) + 16));
        HIP_OR_THROW(hipStreamCreateWithFlags(&m.comm_stream, hipStreamNonBlocking));
        for (hipEvent_t* e : {&m.ev_dcolor, &m.ev_bwd, &m.ev_comm, &m.ev_gather}) HIP_OR_THROW(hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (const char* e = getenv("DVS_A9_CHUNKS")) m.a9_chunks = std::max(1, std::min(64, atoi(e)));
        if (const char* e = getenv("DVS_EXCHANGE_PIPELINE")) m.pipeline = e[0] == '1' && m.a9_chunks > 1;
        m.ev_chunk.resize((size_t)m.a9_chunks); m.ev_ar.resize((size_t)m.a9_chunks);
        for (hipEvent_t& e : m.ev_chunk) HIP_OR_THROW(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : m.ev_ar) HIP_OR_THROW(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (m.rank == 0 && m.a9_chunks > 1)
            logf_("gradient exchange: A9 in %d splat chunks, each chunk's geometry all-reduce behind it%s", m.a9_chunks,
                  m.pipeline ? "; PIPELINED across the iteration boundary: Adam and the next iteration's projection run chunk by chunk as the all-reduces land "
                               "(DVS_EXCHANGE_PIPELINE=1)" : "");
    }
    bool geom_reduced = false;                  // the geometry gradients already left chunk by chunk behind A9
    int chunk_per = 0, n_chunks = 0;            // ... in n_chunks chunks of chunk_per splats (the last one shorter)
    // densification statistics of the step's views, summed over the ranks in densify(): per view and visible
    // splat  grad_accum += |abs-grad|, denom += 1, max_radii = max
    const bool want_stats = refining && !mcmc;
    const int* vis_radii = nullptr;                                   // visible-only Adam: radius > 0 in any of the step's views
    if (m.sequential_views) {
        // one pass per view, gradients accumulating over the views (DVS_VIEWS_MODE=sequential: the reference shape, kept as the check of the multi-view pass)
        for (int v = 0; v < V; ++v) {
            opts.accumulate = v > 0 ? 1 : 0;
            DVS_OR_THROW(dvs_raster_forward(m.ctx, m.stream, &sp, &vcams[(size_t)v], &opts, m.d_out + (size_t)v * img, &m.fwd, nullptr));
            loss_of_view(v);
            const float* dLv = m.d_dL + (size_t)v * img;
            if (fact) {
                g.sh0 = nullptr; g.shN = nullptr; g.dcolor = m.d_dcolor_scratch + (size_t)v * m.n * 3;
                DVS_OR_THROW(dvs_raster_backward_composite(m.ctx, m.stream, &vcams[(size_t)v], &opts, dLv));
                DVS_OR_THROW(dvs_raster_backward_dcolor(m.ctx, m.stream, m.d_dcolor_local + (size_t)v * m.n * 3));
            } else {
                DVS_OR_THROW(dvs_raster_backward_composite(m.ctx, m.stream, &vcams[(size_t)v], &opts, dLv));
            }
            if (want_stats && absgrad) {
                const float* rows = nullptr; int rf = 0;
                DVS_OR_THROW(dvs_get_bwd_intermediates(m.ctx, &rows, &rf));
                DVS_OR_THROW(dvs_densify_accumulate_rows(m.stream, m.n, 1, m.fwd.radii, rows, m.W, m.H, m.d_grad_accum, m.d_denom, m.d_max_radii));
            }
            if (fact && v == V - 1) {           // all local views' colour gradients leave in ONE all-gather, under the last view's A9
                HIP_OR_THROW(hipEventRecord(m.ev_dcolor, m.stream));
                HIP_OR_THROW(hipStreamWaitEvent(m.comm_stream, m.ev_dcolor, 0));
                DVS_OR_THROW(dvs_comm_all_gather_f32(m.comm, m.comm_stream, m.d_dcolor_local, m.d_dcolor_all, (size_t)V * m.n * 3));
            }
            DVS_OR_THROW(dvs_raster_backward_project(m.ctx, m.stream, &sp, &vcams[(size_t)v], &opts, &g));
            if (want_stats && !absgrad) {       // the standard rule: |dL/dmean2D| of the view, threshold growGrad2d (0.0002)
                if (v > 0) throw std::runtime_error("gstrain: DVS_VIEWS_MODE=sequential with useAbsGrad off needs per-view mean2d rows (use the multi-view pass)");
                hipLaunchKernelGGL(k_norm2, dim3((unsigned)((m.n + 255) / 256)), dim3(256), 0, m.stream, m.d_mean2d, m.d_mean2d, m.n);
                DVS_OR_THROW(dvs_densify_accumulate(m.stream, m.n, m.fwd.radii, m.d_mean2d, m.W, m.H, m.d_grad_accum, m.d_denom, m.d_max_radii));
            }
        }
        vis_radii = m.fwd.radii;                // (the last view's — visibleAdam is a single-view notion in this mode)
    } else {
        // ONE multi-view pass: parameters read once, one depth sort / scan / (view, tile) sort / composite launch for all V views, the
        // gradient rows written once (their sum over the views)
        DVS_OR_THROW(dvs_raster_forward_views(m.ctx, m.stream, &sp, vcams.data(), V, &opts, m.d_out));
        DVS_OR_THROW(dvs_get_view_state(m.ctx, 0, &m.fwd));                       // (view-major arrays: fwd.radii = [V][n])
        for (int v = 0; v < V; ++v) loss_of_view(v);
        if (fact) { g.sh0 = nullptr; g.shN = nullptr; g.dcolor = m.d_dcolor_scratch; }
        DVS_OR_THROW(dvs_raster_backward_composite(m.ctx, m.stream, vcams.data(), &opts, m.d_dL));
        if (fact) {
            DVS_OR_THROW(dvs_raster_backward_dcolor(m.ctx, m.stream, m.d_dcolor_local));
            HIP_OR_THROW(hipEventRecord(m.ev_dcolor, m.stream));
            HIP_OR_THROW(hipStreamWaitEvent(m.comm_stream, m.ev_dcolor, 0));
            DVS_OR_THROW(dvs_comm_all_gather_f32(m.comm, m.comm_stream, m.d_dcolor_local, m.d_dcolor_all, (size_t)V * m.n * 3));
        }
        if (want_stats && absgrad) {            // per view, from the composite backward's rows (before A9 consumes them): the exact single-view rule
            const float* rows = nullptr; int rf = 0;
            DVS_OR_THROW(dvs_get_bwd_intermediates(m.ctx, &rows, &rf));
            DVS_OR_THROW(dvs_densify_accumulate_rows(m.stream, m.n, V, m.fwd.radii, rows, m.W, m.H, m.d_grad_accum, m.d_denom, m.d_max_radii));
        }
        if (fact && m.a9_chunks > 1 && m.n >= 256 * m.a9_chunks) {
            // A9 in splat chunks: as soon as chunk k is queued its 44 B/splat of geometry gradients (four ranges of the flat buffer, one
            // grouped collective) start their all-reduce on the communication stream, under the A9 of the chunks behind it
            HIP_OR_THROW(hipEventRecord(m.ev_gather, m.comm_stream));            // (behind the colour all-gather queued above)
            const int per = ((m.n + m.a9_chunks - 1) / m.a9_chunks + 255) / 256 * 256;
            int k = 0;
            for (int first = 0; first < m.n; first += per, ++k) {
                const int count = std::min(per, m.n - first);
                DVS_OR_THROW(dvs_raster_backward_project_chunk(m.ctx, m.stream, &sp, vcams.data(), &opts, &g, first, count));
                HIP_OR_THROW(hipEventRecord(m.ev_chunk[(size_t)k], m.stream));
                HIP_OR_THROW(hipStreamWaitEvent(m.comm_stream, m.ev_chunk[(size_t)k], 0));
                DVS_OR_THROW(dvs_comm_group_start(m.comm));
                DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.comm_stream, m.d_grad[P_POS] + 3 * (size_t)first, 3 * (size_t)count));
                DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.comm_stream, m.d_grad[P_SCALE] + 3 * (size_t)first, 3 * (size_t)count));
                DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.comm_stream, m.d_grad[P_ROT] + 4 * (size_t)first, 4 * (size_t)count));
                DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.comm_stream, m.d_grad[P_OPA] + (size_t)first, (size_t)count));
                DVS_OR_THROW(dvs_comm_group_end(m.comm));
                HIP_OR_THROW(hipEventRecord(m.ev_ar[(size_t)k], m.comm_stream));
            }
            geom_reduced = true;
            chunk_per = per; n_chunks = k;
        } else {
            DVS_OR_THROW(dvs_raster_backward_project(m.ctx, m.stream, &sp, vcams.data(), &opts, &g));
        }
        if (want_stats && !absgrad) {
            // without abs-grad the multi-view pass hands out the SUM over the views of dL/dmean2D: its norm is accumulated once per step
            // for splats visible in at least one view (for V = 1 the reference rule; documented difference for V > 1)
            if (!m.d_vis_radius) HIP_OR_THROW(hipMalloc((void**)&m.d_vis_radius, (size_t)m.cap * sizeof(int) + 16));
            DVS_OR_THROW(dvs_any_view_radius(m.stream, m.n, V, m.fwd.radii, m.d_vis_radius));
            hipLaunchKernelGGL(k_norm2, dim3((unsigned)((m.n + 255) / 256)), dim3(256), 0, m.stream, m.d_mean2d, m.d_mean2d, m.n);
            DVS_OR_THROW(dvs_densify_accumulate(m.stream, m.n, m.d_vis_radius, m.d_mean2d, m.W, m.H, m.d_grad_accum, m.d_denom, m.d_max_radii));
        }
        if (V == 1) {
            vis_radii = m.fwd.radii;
        } else if (m.cfg.visibleAdam && m.world == 1) {
            if (!m.d_vis_radius) HIP_OR_THROW(hipMalloc((void**)&m.d_vis_radius, (size_t)m.cap * sizeof(int) + 16));
            DVS_OR_THROW(dvs_any_view_radius(m.stream, m.n, V, m.fwd.radii, m.d_vis_radius));
            vis_radii = m.d_vis_radius;
        }
    }
    // Adam, per-group learning rates (names gs_train.cpp:52-57; position lr decays exponentially init -> final, scaled by the scene extent)
    const float t = std::min(1.0f, (float)m.step / (float)std::max(1, m.cfg.numIters));
    const float lr_pos = m.extent * std::exp((1.f - t) * std::log(m.cfg.poslrInit) + t * std::log(m.cfg.poslrFinal));
    const float lr[6] = {lr_pos, m.cfg.featurelr, m.cfg.featurelr / 20.f, m.cfg.opacitylr, m.cfg.scalinglr, m.cfg.rotationlr};
    // one launch per set of groups; shN chunks above the active SH degree have g = m = v = 0 (Adam is the identity there)
    static const int width[6] = {3, 3, 45, 1, 3, 4};
    dvs_adam_group ag[6];
    for (int k = 0; k < 6; ++k) {
        ag[k] = dvs_adam_group{m.d_param[k], m.d_grad[k], m.d_m[k], m.d_v[k], (uint64_t)m.dev_floats(k), lr[k], width[k],
                               k == P_SHN ? DVS_SHN_TILED : DVS_SHN_ROWS, 0};
        if (k == P_SHN) ag[k].active_chunks = deg >= 3 ? 0 : (3 * ((deg + 1) * (deg + 1) - 1) + 3) / 4;
    }
    if (deg == 0) ag[P_SHN].count = 0;
    const bool visible_only = m.cfg.visibleAdam && m.world == 1 && vis_radii;       // per-rank visibility would let the replicas drift apart
    const int* adam_gate = visible_only ? vis_radii : nullptr;
    bool sh_adam_done = false;
    // data parallel, over RCCL / xGMI: all-gather of the views' colour gradients + all-reduce of the geometry groups, then every
    // replica rebuilds the summed SH rows from all views (factorised) — or ONE sum-all-reduce of all six groups (they share a buffer)
    if (fact) {             // (all collectives on the communication stream, in the same order on every rank)
        if (!geom_reduced) {
            HIP_OR_THROW(hipEventRecord(m.ev_gather, m.comm_stream));            // (behind the colour all-gather)
            HIP_OR_THROW(hipEventRecord(m.ev_bwd, m.stream));
            HIP_OR_THROW(hipStreamWaitEvent(m.comm_stream, m.ev_bwd, 0));
            DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.comm_stream, m.d_grad_flat, m.geom_floats));
        }
        HIP_OR_THROW(hipEventRecord(m.ev_comm, m.comm_stream));
        // The SH rows need only the colour all-gather: they are rebuilt and their Adam step (sh0 + shN: 192 of the 236 B per splat) runs
        // WHILE the geometry all-reduce is still on the links; the geometry groups follow when it has landed.
        HIP_OR_THROW(hipStreamWaitEvent(m.stream, m.ev_gather, 0));
        std::vector<float> campos(ci_all.size() * 3);               // slot order of the all-gather: [rank][local view]
        for (size_t q = 0; q < ci_all.size(); ++q) for (int k = 0; k < 3; ++k) campos[q * 3 + k] = m.cams[(size_t)ci_all[q]].campos[k];
        DVS_OR_THROW(dvs_sh_grad_combine(m.ctx, m.stream, m.n, m.d_param[P_POS], deg, (int)ci_all.size(), campos.data(), m.d_dcolor_all,
                                         m.d_grad[P_SH0], m.d_grad[P_SHN], 0, DVS_SHN_TILED));
        dvs_adam_group sh_groups[2] = {ag[P_SH0], ag[P_SHN]};
        DVS_OR_THROW(dvs_adam_step_groups(m.stream, sh_groups, 2, 0.9f, 0.999f, 1e-15f, it, adam_gate, m.n));
        sh_adam_done = true;
        if (!(m.pipeline && geom_reduced)) HIP_OR_THROW(hipStreamWaitEvent(m.stream, m.ev_comm, 0));
    } else if (m.comm) {
        DVS_OR_THROW(dvs_comm_all_reduce_sum_f32(m.comm, m.stream, m.d_grad_flat, m.grad_floats));
    }
    const bool refine_now = refining && it > m.cfg.warmupLength && m.cfg.refineEvery > 0 && it % m.cfg.refineEvery == 0;
    const bool reset_now = refining && !mcmc && m.cfg.resetAlphaEvery > 0 && it % m.cfg.resetAlphaEvery == 0;
    const bool prune_now = !refining && m.cfg.pruneStrategy > 0 && m.cfg.pruneInterval > 0 && it % m.cfg.pruneInterval == 0;
    const bool pipelined_tail = m.pipeline && geom_reduced && sh_adam_done;
    if (pipelined_tail) {
        // PIPELINED exchange (round 6; SURVEY 8(e) "Overlap"; DVS_EXCHANGE_PIPELINE=1, off by default until it has run on real links): the
        // geometry gradients left in chunks behind A9. Here every chunk is finished as soon as ITS all-reduce has landed — regulariser,
        // Adam on the four geometry groups, exploration noise, each on the chunk's splat range (all element-wise: bit-identical to the
        // whole-array calls) — and then the NEXT iteration's projection (A2) of that chunk is queued: A2 is per splat, and the chunk's
        // parameters are final (the SH groups were stepped above, under the all-reduces). Only the last chunk's all-reduce is exposed;
        // the all-reduces of the chunks before it run under the Adam / A2 of their predecessors. Iterations that refine, reset or prune
        // change the parameters after Adam: no early projection there (the next forward projects everything itself).
        const bool early = !refine_now && !reset_now && !prune_now && it < m.cfg.numIters && !m.sequential_views;
        std::vector<dvs_camera> ncams;
        dvs_opts nopts = opts;
        if (early) {
            m.next_ci.assign(ci_all.size(), 0);
            draw_cameras(m.next_ci);
            ncams.resize((size_t)V);
            for (int v = 0; v < V; ++v) ncams[(size_t)v] = m.cams[(size_t)m.next_ci[(size_t)m.rank * V + v]];
            nopts.sh_degree = m.cfg.progressiveTrain ? std::min(m.sh_max, it / 1000) : m.sh_max;       // (what the next trainStep will compute from m.step = it)
            nopts.accumulate = 0;
        }
        for (int k = 0; k < n_chunks; ++k) {
            const int first = k * chunk_per, count = std::min(chunk_per, m.n - first);
            HIP_OR_THROW(hipStreamWaitEvent(m.stream, m.ev_ar[(size_t)k], 0));
            if (mcmc)
                DVS_OR_THROW(dvs_mcmc_regularize_range(m.stream, m.n, first, count, m.d_param[P_OPA], m.d_param[P_SCALE], m.d_grad[P_OPA], m.d_grad[P_SCALE], 0.01f, 0.01f));
            dvs_adam_group geo[4] = {ag[P_POS], ag[P_OPA], ag[P_SCALE], ag[P_ROT]};
            for (dvs_adam_group& q : geo) {
                const size_t off = (size_t)q.width * (size_t)first;
                q.param += off; q.grad += off; q.m += off; q.v += off; q.count = (uint64_t)q.width * (uint64_t)count;
            }
            DVS_OR_THROW(dvs_adam_step_groups(m.stream, geo, 4, 0.9f, 0.999f, 1e-15f, it, nullptr, count));
            if (mcmc && m.cfg.noiselr > 0.f)
                DVS_OR_THROW(dvs_mcmc_add_noise_range(m.stream, m.n, first, count, m.d_param[P_POS], m.d_param[P_SCALE], m.d_param[P_ROT], m.d_param[P_OPA],
                                                      m.cfg.noiselr * lr_pos, (uint32_t)it));
            if (early) DVS_OR_THROW(dvs_raster_forward_views_prepare(m.ctx, m.stream, &sp, ncams.data(), V, &nopts, first, count));
        }
    } else {
        if (mcmc)      // opacity and scale regularisers of the MCMC strategy (0.01 each in the published rule): a function of the replicated
                       // parameters, added once (after the exchange) on every rank
            DVS_OR_THROW(dvs_mcmc_regularize(m.stream, m.n, m.d_param[P_OPA], m.d_param[P_SCALE], m.d_grad[P_OPA], m.d_grad[P_SCALE], 0.01f, 0.01f));
        if (sh_adam_done) {
            dvs_adam_group geo_groups[4] = {ag[P_POS], ag[P_OPA], ag[P_SCALE], ag[P_ROT]};
            DVS_OR_THROW(dvs_adam_step_groups(m.stream, geo_groups, 4, 0.9f, 0.999f, 1e-15f, it, adam_gate, m.n));
        } else {
            DVS_OR_THROW(dvs_adam_step_groups(m.stream, ag, 6, 0.9f, 0.999f, 1e-15f, it, adam_gate, m.n));
        }
        if (mcmc && m.cfg.noiselr > 0.f)      // exploration noise, scaled by the position learning rate (`noiselr`, gs_train.cpp:97)
            DVS_OR_THROW(dvs_mcmc_add_noise(m.stream, m.n, m.d_param[P_POS], m.d_param[P_SCALE], m.d_param[P_ROT], m.d_param[P_OPA],
                                            m.cfg.noiselr * lr_pos, (uint32_t)it));
    }
    if (refine_now) { if (mcmc) m.densify_mcmc(it); else m.densify(it); }
    if (reset_now)
        DVS_OR_THROW(dvs_reset_opacity(m.stream, m.n, m.d_param[P_OPA], 0.01f, m.d_m[P_OPA], m.d_v[P_OPA]));
    if (prune_now) {
        m.prune_light(it);
        pruenIteraions.push_back(it);
    }
    if (m.cfg.verbose && m.rank == 0 && (m.step % 100 == 0))          // same line the editor logs (application/editor/source/editor.cpp:1554)
        logf_("Iteraions %d, loss : %f", m.step, (double)getCurrentLoss());
    m.step = it;
    curIteration = it;
    m.host_valid = false;
    if (m.step >= m.cfg.numIters) m.status = TrainingStatus::Training_Done;
}

void GaussianTrainerScene::saveGaussianModel() {
    Impl& m = *impl_;
    // the replicas are identical: one file — unless DVS_SAVE_ALL_RANKS=1 asks every rank for its own (<model>_<it>.ply.rank<r>), which
    // is how the two-rank test checks that they ARE identical, bit for bit
    static const bool all_ranks = [] { const char* e = getenv("DVS_SAVE_ALL_RANKS"); return e && e[0] == '1'; }();
    if (m.rank != 0 && !all_ranks) return;
    m.fetch_host();
    const std::string file = m.model_file(m.step) + (m.rank != 0 ? ".rank" + std::to_string(m.rank) : std::string());
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
void GaussianTrainerScene::exportSparsePointCloud(const std::string& path) {
    Impl& m = *impl_;
    m.fetch_host();
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { logf_("exportSparsePointCloud: cannot open %s", path.c_str()); return; }
    fprintf(f, "ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n", m.n);
    for (int i = 0; i < m.n; ++i) fprintf(f, "%g %g %g\n", m.host[P_POS][3 * i], m.host[P_POS][3 * i + 1], m.host[P_POS][3 * i + 2]);
    fclose(f);
}
void GaussianTrainerScene::saveCameraDatas(const std::string& path) {
    Impl& m = *impl_;
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { logf_("saveCameraDatas: cannot open %s", path.c_str()); return; }
    for (size_t c = 0; c < m.cams.size(); ++c) {
        const dvs_camera& k = m.cams[c];
        fprintf(f, "%zu %d %d %g %g %g %g %g", c, k.width, k.height, k.focal_x, k.focal_y, k.campos[0], k.campos[1], k.campos[2]);
        for (int e = 0; e < 16; ++e) fprintf(f, " %g", k.view[e]);
        fputc('\n', f);
    }
    fclose(f);
}
bool GaussianTrainerScene::isTerminate() const { return impl_->terminate; }
void GaussianTrainerScene::terminate() { impl_->terminate = true; }
bool GaussianTrainerScene::isPruningSplat() const { return impl_->pruning; }
void GaussianTrainerScene::resetGaussian() {
    Impl& m = *impl_;
    if (!m.ctx || m.init_host[P_OPA].empty()) return;
    HIP_OR_THROW(hipSetDevice(m.device));
    HIP_OR_THROW(hipStreamSynchronize(m.stream));
    m.n = (int)m.init_host[P_OPA].size();
    for (int g = 0; g < 6; ++g) {
        const size_t bytes = m.dev_floats_for(g, m.cap) * sizeof(float);
        for (float** p : {&m.d_param[g], &m.d_m[g], &m.d_v[g]}) HIP_OR_THROW(hipMemset(*p, 0, bytes));
        m.upload(g, m.init_host[g]);
    }
    HIP_OR_THROW(hipMemset(m.d_grad_flat, 0, m.grad_floats * sizeof(float)));
    m.reset_stats();
    m.step = 0; curIteration = 0; pruenIteraions.clear();
    m.host_valid = false;
    (void)dvs_raster_forward_cancel_prepared(m.ctx);          // (a pipelined step may have projected the next iteration's splats already)
    m.status = TrainingStatus::Training;
    m.t0 = std::chrono::steady_clock::now();
}
void GaussianTrainerScene::setDensifyStrategy(int strategy) { impl_->cfg.densifyStrategy = std::min(2, std::max(0, strategy)); impl_->report_config(); }
void GaussianTrainerScene::setModelPath(const std::string& path) { impl_->cfg.modelPath = path; }
void GaussianTrainerScene::updateFocusRegion(const dvs_types::Vec3& position, const dvs_types::Vec3& rotation, const dvs_types::Vec3& scale) {
    focus_region_position = position; focus_region_rotation = rotation; focus_region_scale = scale;     // stored; enableFocusRegion is not used by the trainer
}
std::string GaussianTrainerScene::getCurrentTrainingPhaseName() const {
    switch (impl_->status) {
        case TrainingStatus::Loading_Prepare: return "Loading";
        case TrainingStatus::Colmap_Sfm: return "SfM";
        case TrainingStatus::Preprocess_Done: return "Preprocess done";
        case TrainingStatus::Training: return impl_->pruning ? "Pruning" : "Training";
        case TrainingStatus::Training_Done: return "Done";
        case TrainingStatus::Loading_Failed: return "Failed";
        default: return "GS2Mesh";
    }
}
float GaussianTrainerScene::getProgressOnCurrentPhase() const {
    return impl_->cfg.numIters > 0 ? std::min(1.f, (float)impl_->step / (float)impl_->cfg.numIters) : 0.f;
}
double GaussianTrainerScene::getEstimateTrainingTime() const {
    const double el = getTrainingElpasedTime();
    return impl_->step > 0 ? el / impl_->step * std::max(0, impl_->cfg.numIters - impl_->step) : 0.0;
}
dvs_types::Mat4 GaussianTrainerScene::getCameraProjection(int i) const {
    // perspective projection of the training camera (column-major, OpenGL clip conventions as the editor's Frustum expects)
    dvs_types::Mat4 P{};
    float* o = reinterpret_cast<float*>(&P);
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    if (i < 0 || i >= (int)impl_->cams.size()) return P;
    const dvs_camera& k = impl_->cams[i];
    const float zn = 0.2f, zf = 100.f;
    o[0] = 1.f / k.tan_fovx; o[5] = 1.f / k.tan_fovy; o[10] = -(zf + zn) / (zf - zn); o[11] = -1.f; o[14] = -2.f * zf * zn / (zf - zn);
    return P;
}
dvs_types::Quat GaussianTrainerScene::getCameraRotation(int i) const {
    dvs_types::Quat q{};
    if (i < 0 || i >= (int)impl_->cams.size()) return q;
    const float* v = impl_->cams[i].view;          // view[c*4+r]: world -> camera; rotation R[r][c] = v[c*4+r]; camera -> world = R^T
    const float R[3][3] = {{v[0], v[1], v[2]}, {v[4], v[5], v[6]}, {v[8], v[9], v[10]}};      // R^T rows
    const float tr = R[0][0] + R[1][1] + R[2][2];
    float w, x, y, z;
    if (tr > 0.f) { const float s = std::sqrt(tr + 1.f) * 2.f; w = 0.25f * s; x = (R[2][1] - R[1][2]) / s; y = (R[0][2] - R[2][0]) / s; z = (R[1][0] - R[0][1]) / s; }
    else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) { const float s = std::sqrt(1.f + R[0][0] - R[1][1] - R[2][2]) * 2.f; w = (R[2][1] - R[1][2]) / s; x = 0.25f * s; y = (R[0][1] + R[1][0]) / s; z = (R[0][2] + R[2][0]) / s; }
    else if (R[1][1] > R[2][2]) { const float s = std::sqrt(1.f + R[1][1] - R[0][0] - R[2][2]) * 2.f; w = (R[0][2] - R[2][0]) / s; x = (R[0][1] + R[1][0]) / s; y = 0.25f * s; z = (R[1][2] + R[2][1]) / s; }
    else { const float s = std::sqrt(1.f + R[2][2] - R[0][0] - R[1][1]) * 2.f; w = (R[1][0] - R[0][1]) / s; x = (R[0][2] + R[2][0]) / s; y = (R[1][2] + R[2][1]) / s; z = 0.25f * s; }
    float* o = reinterpret_cast<float*>(&q);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
    return q;
}
dvs_types::Vec3 GaussianTrainerScene::getCameraPos(int i) const {
    dvs_types::Vec3 p{};
    if (i < 0 || i >= (int)impl_->cams.size()) return p;
    float* o = reinterpret_cast<float*>(&p);
    for (int k = 0; k < 3; ++k) o[k] = impl_->cams[i].campos[k];
    return p;
}
const std::vector<float>& GaussianTrainerScene::getPoints3D(int) { return impl_->init_host[P_POS]; }
bool is_device_support_gstrain() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return false;
    hipDeviceProp_t pr;
    return hipGetDeviceProperties(&pr, 0) == hipSuccess && strstr(pr.gcnArchName, "gfx95") != nullptr;
}
bool is_driver_support() { int v = 0; return hipRuntimeGetVersion(&v) == hipSuccess && v > 0; }
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
        const double nv = (double)std::max(1, m.loss_views);             // the sums cover the step's views: report their mean
        m.last_loss = (float)(l1 / nv) + (w > 0.f ? w * (1.f - (float)(ssim_sum / nv / (3.0 * m.W * m.H))) : 0.f);
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
// The hosts have no try/catch around these dlsym'd calls (gs_train.cpp:152-179): nothing may escape across the C boundary. A
// failure is logged, the scene is marked Loading_Failed / finished (so that the host's loop `get_cur_step >= maxIteration` ends).
#define GSTRAIN_GUARD(name_, scene, body)                                                                        \
    try { body; } catch (const std::exception& e) {                                                               \
        logf_("%s failed: %s", name_, e.what());                                                                    \
        if (scene) { scene->setTrainingStatus(GaussianTrainerScene::TrainingStatus::Loading_Failed); scene->terminate(); } \
    } catch (...) { logf_("%s failed: unknown exception", name_); if (scene) scene->terminate(); }
__attribute__((visibility("default"))) bool load_train_data(GaussianTrainerScene* scene, const std::string& path) {
    bool ok = false;
    GSTRAIN_GUARD("load_train_data", scene, ok = scene && scene->loadTrainData(path));
    return ok;
}
__attribute__((visibility("default"))) void train_step(GaussianTrainerScene* scene) {
    if (!scene || scene->isTerminate()) return;
    GSTRAIN_GUARD("train_step", scene, scene->trainStep());
}
__attribute__((visibility("default"))) int get_cur_step(GaussianTrainerScene* scene) {
    if (!scene) return 0;
    // after a fatal error the step counter reports "done" so that the host's `while (get_cur_step < maxIteration)` loop terminates
    return scene->isTerminate() ? std::max(scene->getCurrentIterations(), scene->maxIteriaons()) : scene->getCurrentIterations();
}
__attribute__((visibility("default"))) void save_splat_model(GaussianTrainerScene* scene) {
    if (!scene) return;
    GSTRAIN_GUARD("save_splat_model", scene, scene->saveGaussianModel());
}
__attribute__((visibility("default"))) void export_mesh(GaussianTrainerScene* scene) { if (scene) { GSTRAIN_GUARD("export_mesh", scene, scene->exportMesh("")); } }
__attribute__((visibility("default"))) void delete_splat(GaussianTrainerScene* scene) { try { delete scene; } catch (...) { logf_("delete_splat failed"); } }
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
