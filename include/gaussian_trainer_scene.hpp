// gaussian_trainer_scene.hpp — reconstruction of the header DIVSHOT's hosts compile against
// (`#include <gaussian_trainer_scene.hpp>` at application/diverseshot-cli/source/gs_train.cpp:3 and the editor;
// the original ships only with the closed `gstrain` plugin). Field names, types and defaults are
// the ones the in-tree callers use:
//   GaussianTrainConfig fields  gs_train.cpp:50-103, editor.cpp:1750-1961,2000-2020, inspector_panel.cpp:778-925
//   defaults                    application/diverseshot-cli/source/main.cpp:12-70 (CLI defaults)
//   GSPackLevel bit flags       gs_train.cpp:89-96
//   GaussianTrainerScene surface editor.cpp:1413-1654,2023-2035; inspector_panel.cpp:765-1000
// Because the config crosses dlsym() by const reference and carries std::string members, host and plugin must be
// built against THIS header with the same C++ standard library (SURVEY.md §7 "ABI of GaussianTrainConfig").
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

enum GSPackLevel : int { PackF32ToU8 = 1, PackTileID = 2 };   // bit flags (gs_train.cpp:91-96)

struct GaussianTrainConfig {
    std::string sourcePath, modelPath = "../out_put/iteration", cameraPosePath, pointCloudPath;
    int numIters = 30000;                 // --maxIteration (main.cpp:19)
    int modelType = 0;                    // 0 = 3DGS (2DGS etc. are out of scope)
    int densifyStrategy = 1;              // 0 ADC / 1 MCMC / 2 ADC+ (main.cpp:20,29)
    int warmupLength = 500, refineEvery = 100, resetAlphaEvery = 3000, refineStopIter = 15000;   // main.cpp:47-48
    int refineScale2dStopIter = 15000, pruneInterval = 70000, pruneStrategy = 1;
    int maxImageWidth = 2048, maxImageHeight = 2048, maxImageCount = 0;
    int capMax = 3000000;                 // gs_train.cpp:89
    int packLevel = PackF32ToU8;          // main.cpp:52 (--packLevel 1)
    int meshResolution = 0, resolutionSchedule = 0, cameraModel = 0, datasetType = 0, quality = 0, mapperType = 0;
    int videoStrategy = 0, videoFps = 0;
    float growGrad2d = 0.0002f;           // main.cpp:46
    float ssimWeight = 0.2f;              // main.cpp:24
    float noiselr = 1e5f;                 // main.cpp:67
    float poslrInit = 0.00016f, poslrFinal = 0.0000016f, rotationlr = 0.001f, scalinglr = 0.005f;   // main.cpp:31 (commented defaults)
    float featurelr = 0.0025f, opacitylr = 0.05f;
    float min_opacity = 0.005f, pruneOpacity = 0.005f, pruneScale3d = 0.1f, pruneScale2d = 0.15f;
    bool progressiveTrain = true, useAbsGrad = true, revisedOpacity = true, mipAntiliased = false;
    bool exportMesh = false, normalConsistencyLoss = false, useMask = false, verbose = true, bestQuality = false;
    bool enableBg = false, enableFocusRegion = false, cullSH = false, singleCamera = false, outputSparsePoints = false;
    bool visibleAdam = false, pixelGradScale = false;
};

class GaussianTrainerScene {
public:
    enum class TrainingStatus { Loading_Prepare, Colmap_Sfm, Preprocess_Done, Training, Training_Done, Loading_Failed, GS2Mesh };

    GaussianTrainerScene(const GaussianTrainConfig& cfg, int loadItr);
    ~GaussianTrainerScene();
    GaussianTrainerScene(const GaussianTrainerScene&) = delete;
    GaussianTrainerScene& operator=(const GaussianTrainerScene&) = delete;

    bool loadTrainData(const std::string& path);     // accepts "synthetic:N=..,W=..,H=..,cams=..,sh=..,seed=.." 
    void trainSetup();
    void trainStep();                                 // one iteration: sample camera -> raster fwd -> loss -> raster bwd -> Adam
    void saveGaussianModel();                         // PLY at modelPath (external/tinygsplat/tiny_gsplat.cpp:168-241 layout)
    void exportMesh(const std::string& path);         // out of scope: logs and returns
    bool isTrain() const;
    void startTrain();
    void pauseTrain();
    int  getCurrentIterations() const;
    float getCurrentLoss();                           // synchronises the training stream
    int& maxIteriaons();
    GaussianTrainConfig& getTrainConfig();
    TrainingStatus getCurrentTrainingStatus() const;
    void setTrainingStatus(TrainingStatus s);
    double getTrainingElpasedTime() const;
    int getNumGaussians() const;
    int getNumCameras() const;
    // trainer -> viewer hand-off (editor.cpp:1459-1473 -> GaussianModel::update_from_cpu, gaussian_model.cpp:43-68)
    const std::vector<float>& getGaussianPositionCpu();
    const std::vector<float>& getGaussianSH0Cpu();
    const std::vector<float>& getGaussianSHNCpu();
    const std::vector<float>& getGaussianOpcaitiesCpu();
    const std::vector<float>& getGaussianScalingsCpu();
    const std::vector<float>& getGaussianRotationsCpu();

    // public fields the editor touches
    bool ShowTrainView = false;
    int curIteration = 0;
    std::vector<int> pruenIteraions;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// C symbols the hosts resolve with dlsym (gs_train.cpp:24,105-109,144-150,178; plugin.cpp:89-111)
extern "C" {
void  gstrain_init();
void* create_splat(const GaussianTrainConfig& config, int loadItr);
bool  load_train_data(GaussianTrainerScene* scene, const std::string& path);
void  train_step(GaussianTrainerScene* scene);
int   get_cur_step(GaussianTrainerScene* scene);
void  save_splat_model(GaussianTrainerScene* scene);
void  export_mesh(GaussianTrainerScene* scene);
void  delete_splat(GaussianTrainerScene* scene);
void  gstrain_destroy();
const char* get_description();
void* create_instance();
}
