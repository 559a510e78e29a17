// diverseshot_cli.cpp — this build's counterpart of DIVSHOT's `diverseshot-cli` host: same flags and defaults
// (application/diverseshot-cli/source/main.cpp:9-70), same plugin loading (dlopen of "libgstrain.so" next to the
// executable, RTLD_LAZY|RTLD_LOCAL — diverse/diverse_base/source/core/plugin.cpp:74,89; symbols via dlsym, plugin.cpp:153-167)
// and the same call order as train_gaussian() (application/diverseshot-cli/source/gs_train.cpp:7-181):
//   gstrain_init -> create_splat -> load_train_data (false => exit(-1)) -> { get_cur_step, train_step }* with a progress line
//   every 500 steps and save_splat_model every 10000 steps once i > resetAlphaEvery -> [export_mesh] -> save_splat_model ->
//   delete_splat -> gstrain_destroy.
// The reference's own gs_train.cpp cannot be compiled here (it needs the closed header and <format>);
// CLI11 / indicators / spdlog are replaced by a few lines of standard C++.
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <map>
#include <string>
#include "../../include/gaussian_trainer_scene.hpp"

static std::map<std::string, std::string> g_opts = {
    // flag -> default (main.cpp:12-70)
    {"maxImageWidth", "2048"}, {"maxImageHeight", "2048"}, {"inputPath", ""}, {"ssim", "0.2"}, {"outputPath", "../out_put/iteration"},
    {"modelType", "0"}, {"densifyStrategy", "1"}, {"maxIteration", "30000"}, {"progressTrain", "1"}, {"load_itr", "-1"},
    {"absgrad", "1"}, {"growGrad2d", "0.0002"}, {"warmupLength", "500"}, {"refineEvery", "100"}, {"resetAlphaEvery", "3000"},
    {"refineStopIter", "15000"}, {"refineScale2dStopIter", "15000"}, {"minOpacity", "0.005"}, {"pruneEvery", "70000"},
    {"pruneStrategy", "1"}, {"revisedOpacity", "1"}, {"mipAntiliased", "0"}, {"packLevel", "1"}, {"exportMesh", "0"},
    {"noiselr", "100000"}, {"useMask", "0"}, {"eval", "0"},
    {"viewsPerIter", "1"}};       // extension of this build: cameras per train_step as one multi-view pass (BASELINE config C4: 8)

static bool as_bool(const std::string& v) { return v == "1" || v == "true" || v == "True" || v == "on"; }

template <class F> static F sym(void* h, const char* name, bool required = true) {
    void* p = dlsym(h, name);
    if (!p && required) { std::cerr << "can't find " << name << " symbol\n"; exit(-1); }
    return reinterpret_cast<F>(p);
}

int main(int argc, const char* argv[]) {
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);      // dmabuf IPC for RCCL between the ranks' processes; read when the HSA runtime starts (first HIP call)
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--help" || a == "-h") {
            std::cout << "gaussian_train (divshot_amd counterpart of diverseshot-cli)\nflags:";
            for (auto& kv : g_opts) std::cout << " --" << kv.first << " [" << kv.second << "]";
            std::cout << "\n--inputPath accepts synthetic:N=..,W=..,H=..,cams=..,sh=..,seed=..\n";
            return 0;
        }
        if (a == "--version") { std::cout << "1.0.0\n"; return 0; }
        if (a.rfind("--", 0) != 0) { std::cout << "Command Line Error: unexpected argument " << a << "\n"; return 1; }
        a = a.substr(2);
        std::string val;
        size_t eq = a.find('=');
        if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); }
        else if (a == "eval") val = "1";
        else if (i + 1 < argc) val = argv[++i];
        if (!g_opts.count(a)) { std::cout << "Command Line Error: unknown flag --" << a << "\n"; return 1; }   // config_extras_mode::error
        g_opts[a] = val;
    }
    // plugin: "libgstrain.so" next to the executable (plugin.cpp:74 builds parent_path / "lib<name>.so")
    std::error_code ec;
    auto exe = std::filesystem::read_symlink("/proc/self/exe", ec);
    const std::string plugin_path = (exe.parent_path() / "libgstrain.so").string();
    void* h = dlopen(plugin_path.c_str(), RTLD_LAZY | RTLD_LOCAL);
    if (!h) { std::cerr << "Failed to load library or its dependencies : " << plugin_path << " : " << dlerror() << "\ncreate plugin failed\n"; exit(-1); }
    (void)sym<const char* (*)()>(h, "get_description", false);
    (void)sym<void* (*)()>(h, "create_instance", false);
    typedef void (*voidFunc)();
    auto gsplat_init = sym<voidFunc>(h, "gstrain_init");
    gsplat_init();

    const std::string source_path = g_opts["inputPath"];
    const int max_iteraion = atoi(g_opts["maxIteration"].c_str());
    const int load_itr = atoi(g_opts["load_itr"].c_str());
    auto start = std::chrono::high_resolution_clock::now();
    GaussianTrainConfig train_config;                       // field-by-field, as gs_train.cpp:50-103
    train_config.sourcePath = source_path;
    train_config.growGrad2d = (float)atof(g_opts["growGrad2d"].c_str());
    train_config.warmupLength = atoi(g_opts["warmupLength"].c_str());
    train_config.refineEvery = atoi(g_opts["refineEvery"].c_str());
    train_config.resetAlphaEvery = atoi(g_opts["resetAlphaEvery"].c_str());
    train_config.refineStopIter = atoi(g_opts["refineStopIter"].c_str());
    train_config.revisedOpacity = as_bool(g_opts["revisedOpacity"]);
    train_config.refineScale2dStopIter = atoi(g_opts["refineScale2dStopIter"].c_str());
    train_config.ssimWeight = (float)atof(g_opts["ssim"].c_str());
    train_config.modelPath = g_opts["outputPath"];
    train_config.modelType = atoi(g_opts["modelType"].c_str());
    train_config.densifyStrategy = atoi(g_opts["densifyStrategy"].c_str());
    train_config.progressiveTrain = as_bool(g_opts["progressTrain"]);
    train_config.useAbsGrad = as_bool(g_opts["absgrad"]);
    train_config.pruneInterval = atoi(g_opts["pruneEvery"].c_str());
    train_config.pruneStrategy = atoi(g_opts["pruneStrategy"].c_str());
    train_config.mipAntiliased = as_bool(g_opts["mipAntiliased"]);
    train_config.maxImageHeight = atoi(g_opts["maxImageHeight"].c_str());
    train_config.maxImageWidth = atoi(g_opts["maxImageWidth"].c_str());
    train_config.exportMesh = as_bool(g_opts["exportMesh"]);
    train_config.useMask = as_bool(g_opts["useMask"]);
    train_config.numIters = max_iteraion;
    if (g_opts.count("viewsPerIter")) train_config.viewsPerIter = atoi(g_opts["viewsPerIter"].c_str());   // extension of this build (config C4)
    train_config.normalConsistencyLoss = false;
    if (train_config.exportMesh) { train_config.normalConsistencyLoss = true; train_config.useMask = true; }
    train_config.verbose = true;
    train_config.capMax = 3000000;
    const int packLevel = atoi(g_opts["packLevel"].c_str());
    train_config.packLevel = packLevel == 0 ? 0 : (packLevel == 1 ? (int)GSPackLevel::PackF32ToU8 : (int)(GSPackLevel::PackF32ToU8 | GSPackLevel::PackTileID));
    train_config.noiselr = (float)atof(g_opts["noiselr"].c_str());
    train_config.bestQuality = true;

    typedef void* (*createFunc)(const GaussianTrainConfig& config, int loadItr);
    auto create_splat = sym<createFunc>(h, "create_splat");
    auto scene = (GaussianTrainerScene*)create_splat(train_config, load_itr);
    if (!scene) { std::cout << "create_splat failed\n"; exit(-1); }
    typedef bool (*loadDataFunc)(GaussianTrainerScene* scene, const std::string& path);
    auto loadData = sym<loadDataFunc>(h, "load_train_data");
    if (!loadData(scene, source_path)) { std::cout << "load data failed!, please check source_path \n"; exit(-1); }
    std::cout << "Training [Preprocess done]\n";

    typedef void (*SceneParamFunc)(GaussianTrainerScene* scene);
    auto train_step = sym<SceneParamFunc>(h, "train_step");
    auto save_splat_model = sym<SceneParamFunc>(h, "save_splat_model");
    auto export_mesh = sym<SceneParamFunc>(h, "export_mesh");
    auto delete_splat = sym<SceneParamFunc>(h, "delete_splat");
    typedef int (*IntReturnFunc)(GaussianTrainerScene* scene);
    auto get_cur_step = sym<IntReturnFunc>(h, "get_cur_step");
    auto loop_start = std::chrono::high_resolution_clock::now();
    const int first = get_cur_step(scene);
    while (true) {
        auto i = get_cur_step(scene);
        if (i >= max_iteraion) break;
        train_step(scene);
        if (i % 500 == 0) {
            const double el = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - loop_start).count();
            printf("Training [%3d%%] train step : %d/%d  (%.1f it/s)\n", (int)((i / (float)max_iteraion) * 100), i, max_iteraion,
                   i > first ? (i - first) / el : 0.0);
            fflush(stdout);
        }
        if (i > train_config.resetAlphaEvery && i % 10000 == 0) save_splat_model(scene);
    }
    std::cout << "Training [100%] Train Done\n";
    if (train_config.exportMesh) export_mesh(scene);
    auto end = std::chrono::high_resolution_clock::now();
    std::cout << "train cost time : " << std::chrono::duration_cast<std::chrono::seconds>(end - start).count() << " seconds\n";
    save_splat_model(scene);
    delete_splat(scene);
    auto gstrain_destroy = sym<voidFunc>(h, "gstrain_destroy");
    gstrain_destroy();
    return 0;
}
