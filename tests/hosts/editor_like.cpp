// editor_like.cpp — a host of the SECOND kind the reference has: it does not dlsym() the plugin's C symbols (that is diverseshot-cli,
// gs_train.cpp:105-109) but links libgstrain directly, constructs the class itself and pulls the trained splats through the CPU
// getters, as the editor does:
//   probes                  application/editor/source/editor.cpp:1534,1539   is_device_support_gstrain() / is_driver_support()
//   construction            editor.cpp:2023   add_component<GaussianTrainerScene>(trainConfig, -1); :2024 setModelPath
//   load + setup            editor.cpp:2033-2035  loadTrainData(path) -> trainSetup()
//   step loop               editor.cpp:1603 (isTerminate) around trainStep()
//   trainer -> viewer       editor.cpp:1459-1473  six getGaussian*Cpu() + getNumGaussians() -> GaussianModel::update_from_cpu
//   the copy itself         diverse/source/assets/gaussian_model.cpp:43-68: memcpy of n x {12, 16, 12, 4, 12, 180} bytes
// Built and run by tests/test_plugin.py::test_editor_like_host_pulls_the_model_through_the_getters (-m gpu); it writes what the
// "viewer model" received to a raw dump that the test compares, bit for bit, with the PLY the same scene saved.
//
// usage: editor_like <synthetic spec> <model path prefix> <iterations before the first pull> <iterations after it> <dump prefix>
#include <gaussian_trainer_scene.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
// the viewer-side container and its update_from_cpu, restated with gaussian_model.cpp:52-65's sizes
struct ViewerModel {
    std::vector<vec3> pos, scales;
    std::vector<vec4> rot;
    std::vector<float> opacities;
    std::vector<float> shs_0, shs_n;          // 3 and 45 floats per splat; shs_n is [j*3+c] (gaussian_model.cpp:163-167)
    int n = 0;
    void update_from_cpu(const float* pos_d, const float* shs0_d, const float* shsn_d, const float* opacities_d, const float* scales_d,
                         const float* rots_d, int num_gaussians) {
        n = num_gaussians;
        pos.resize(n); rot.resize(n); scales.resize(n); opacities.resize(n);
        shs_0.resize((size_t)n * 3); shs_n.resize((size_t)n * 45);
        memcpy(pos.data(), pos_d, n * sizeof(vec3));
        memcpy(rot.data(), rots_d, n * sizeof(vec4));
        memcpy(scales.data(), scales_d, n * sizeof(vec3));
        memcpy(opacities.data(), opacities_d, n * sizeof(float));
        memcpy(shs_0.data(), shs0_d, n * sizeof(float) * 3);
        memcpy(shs_n.data(), shsn_d, n * sizeof(float) * 45);
    }
    bool dump(const char* path) const {
        FILE* f = fopen(path, "wb");
        if (!f) return false;
        const long long hdr = n;
        fwrite(&hdr, sizeof hdr, 1, f);
        fwrite(pos.data(), sizeof(vec3), n, f);
        fwrite(shs_0.data(), 12, n, f);
        fwrite(shs_n.data(), 180, n, f);
        fwrite(opacities.data(), 4, n, f);
        fwrite(scales.data(), sizeof(vec3), n, f);
        fwrite(rot.data(), sizeof(vec4), n, f);
        return fclose(f) == 0;
    }
};

void pull(GaussianTrainerScene& gs_train, ViewerModel& model) {
    // editor.cpp:1459-1473 verbatim in structure: copies of the six vectors, then the raw pointers and the count
    auto pos_cpu = gs_train.getGaussianPositionCpu();
    auto sh0_cpu = gs_train.getGaussianSH0Cpu();
    auto shn_cpu = gs_train.getGaussianSHNCpu();
    auto opacity_cpu = gs_train.getGaussianOpcaitiesCpu();
    auto scale_cpu = gs_train.getGaussianScalingsCpu();
    auto rot_cpu = gs_train.getGaussianRotationsCpu();
    model.update_from_cpu(pos_cpu.data(), sh0_cpu.data(), shn_cpu.data(), opacity_cpu.data(), scale_cpu.data(), rot_cpu.data(),
                          gs_train.getNumGaussians());
}
}   // namespace

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s <synthetic spec> <model prefix> <iters a> <iters b> <dump prefix>\n", argv[0]); return 2; }
    const std::string spec = argv[1], model_prefix = argv[2], dump_prefix = argv[5];
    const int iters_a = atoi(argv[3]), iters_b = atoi(argv[4]);
    if (!is_driver_support() || !is_device_support_gstrain()) { fprintf(stderr, "editor_like: no supported device\n"); return 3; }

    GaussianTrainConfig trainConfig;                       // the fields the editor's training dialog sets (editor.cpp:1750-1961), test-sized
    trainConfig.numIters = iters_a + iters_b;
    trainConfig.densifyStrategy = 0;                       // ADC: a refinement changes the splat count
    trainConfig.warmupLength = 5; trainConfig.refineEvery = 10; trainConfig.refineStopIter = 100000;
    trainConfig.progressiveTrain = false; trainConfig.ssimWeight = 0.2f; trainConfig.capMax = 200000;
    trainConfig.verbose = false;
    GaussianTrainerScene gs_train(trainConfig, -1);        // editor.cpp:2023
    gs_train.setModelPath(model_prefix);                   // editor.cpp:2024
    if (!gs_train.loadTrainData(spec)) { fprintf(stderr, "editor_like: loadTrainData failed\n"); return 4; }
    gs_train.trainSetup();
    gs_train.startTrain();

    ViewerModel model;
    const int n0 = gs_train.getNumGaussians();
    for (int i = 0; i < iters_a && !gs_train.isTerminate(); ++i) gs_train.trainStep();
    pull(gs_train, model);
    gs_train.saveGaussianModel();                          // <prefix>_<iters_a>.ply
    if (!model.dump((dump_prefix + "_a.bin").c_str())) return 5;
    const int n1 = model.n;
    for (int i = 0; i < iters_b && !gs_train.isTerminate(); ++i) gs_train.trainStep();
    pull(gs_train, model);                                  // the getters must notice that the model moved on (and that N changed)
    gs_train.saveGaussianModel();
    if (!model.dump((dump_prefix + "_b.bin").c_str())) return 5;
    printf("editor_like: cameras %d, splats %d -> %d -> %d, iterations %d, loss %g, status %d\n", gs_train.getNumCameras(), n0, n1, model.n,
           gs_train.getCurrentIterations(), gs_train.getCurrentLoss(), (int)gs_train.getCurrentTrainingStatus());
    return gs_train.isTerminate() ? 6 : 0;
}
