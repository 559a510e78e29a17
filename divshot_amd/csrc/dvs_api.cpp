// dvs_api.cpp — the C-ABI of libdvsraster.so (include/dvs_raster.h): context, HBM arenas, and the
// host-side orchestration of one forward / backward pass of the rasterizer on one MI355X.
//
// Replaces, behind a plain-C surface, what the reference's closed `gsplatrast` CUDA library does for
// `gstrain`'s train_step() (call sites application/diverseshot-cli/source/gs_train.cpp:156,
// application/editor/source/editor.cpp:1620).
//
// HBM layout (all arenas are ctx-owned, grow-only, sized for max_splats / max image at create):
//   per splat   : radii i32 | splat2d 64-B record (mean, conic, opacity, rgb, depth, radius) | depth f32 | flags u32 |
//                 tiles_touched u32 | depth_key u32 x2 (ping-pong) | ids u32 x2 (ping-pong)          = 96 B/splat
//   per instance: tile u32 x2 | splat u32 x2 (ping-pong)                                              = 16 B/instance
//   per tile    : range u32x2          per pixel: final_T f32, n_contrib u32
//   backward    : one 48-B row per splat: moments S_x S_y S_xx S_xy S_yy of dL/dG*G about the mean (A9 turns them into
//                 dL/dmean2D and dL/dconic), dL/dopacity, dL/drgb x3, |dL/dmean2D| x2, pad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/dvs_raster.h"
#include "dvs_kernels.h"

static thread_local std::string g_last_error;
static void set_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
}
void dvs_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }      // shared with dvs_comm.cpp
#define HIPCHECK(expr)                                                   \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) { set_error(#expr, _e, __FILE__, __LINE__); return DVS_ERR_HIP; } \
    } while (0)

namespace {
struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return DVS_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4;                 // 25 % headroom so a growing scene does not realloc every step
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, need); want = need; }
        if (e != hipSuccess) { set_error("hipMalloc", e, __FILE__, __LINE__); p = nullptr; return DVS_ERR_HIP; }
        bytes = want;
#ifdef DVS_EXPERIMENT
        // timing-only ablation builds (tools/xbuild.sh) may skip stores: zeroed arenas keep every index a later kernel gathers through valid
        (void)hipMemset(p, 0, want);
#endif
        return DVS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

DvsCam to_dev_cam(const dvs_camera& c) {
    DvsCam d;
    memcpy(d.view, c.view, sizeof d.view); memcpy(d.proj, c.proj, sizeof d.proj);
    d.tan_fovx = c.tan_fovx; d.tan_fovy = c.tan_fovy; d.focal_x = c.focal_x; d.focal_y = c.focal_y;
    memcpy(d.campos, c.campos, sizeof d.campos);
    d.width = c.width; d.height = c.height; memcpy(d.bg, c.bg, sizeof d.bg);
    return d;
}
int bits_for(uint32_t max_value) { int b = 0; while (max_value) { ++b; max_value >>= 1; } return b; }
}  // namespace

struct dvs_ctx {
    int device = 0;
    size_t max_splats = 0;
    int max_w = 0, max_h = 0;
    int max_views = 1;                   // views one forward can batch (dvs_create_views)
    int n_views = 1;                     // views of the last forward
    Buf radii, splat2d, depth, flags, tiles_touched, rect, rect_sorted, key[2], ids[2];
    Buf inst_tile[2], inst_splat[2];
    uint32_t* live_splat = nullptr;      // A7's compacted per-tile lists of the last forward (in the sort's spare instance arrays), or null
    uint32_t* live_pos = nullptr;
    bool live_lists = true;              // env DVS_LIVE_LISTS=0: A8 walks the full lists (A/B and parity of the two routes)
    Buf tmp_keys, tmp_vals;
    bool export_tiles = false;           // dvs_set_export_sorted_tiles: asynchronous forwards materialise the sorted tile ids too
    int fe_rank_atomic = 0;              // the sorts' scatters rank by returning LDS adds (frontend.hip): the device passed dvs_fe_probe_rank_atomic
    Buf ranges_canon;                    // (start, end) per tile as k_render_fwd decodes them when A6 rides on the tile sort (frontend.hip)
    Buf ranges, final_T, n_contrib;      // `ranges` starts with the front end's zeroed words (fe_zero_bytes), the tile ranges follow: ONE memset per forward
    // the binning front end (frontend.hip): every view is a segment of the sorts
    size_t fe_zero_bytes = 0;            // [kred: DVS_FE_KRED_WORDS u32][super sums: max_views x fe_nsb u64]
    uint32_t fe_nbv = 0, fe_nsb = 0;     // A3 workgroups / super sums per view at max_splats
    Buf fe_state;                        // [seg_all 16][seg_vis 16][seg_tile 16][superexcl V x nsb][totals V x 2048][block sums V x nbv]
    Buf fe_hist;                         // histogram table of the segmented sorts
    int fe_seg_n = -1, fe_seg_V = -1;    // what seg_all currently describes
    uint32_t fe_seg_rows = 0;
    Buf g_rows;
    Buf dcolor;                          // [views, n, 3] per-view colour gradients of a batched A9 when the caller gives no buffer
    uint64_t* total_dev = nullptr;       // [0] = T of the last forward, [1] = number of forwards whose T exceeded the instance capacity
    uint64_t* total_host = nullptr;      // pinned copy of both words (async: refreshed by every forward, read by the next one)
    uint64_t inst_cap = 0;               // instances the instance arenas can hold
    uint64_t inst_grow_events = 0;       // how often they have been enlarged since dvs_create (dvs_get_arena_info)
    bool async_T = false;                // dvs_set_async: no host synchronisation inside dvs_raster_forward
    uint64_t* rec_masks = nullptr; uint64_t rec_cap = 0;      // dvs_debug_record_decisions
    uint64_t overflow_seen = 0;          // value of total_host[1] already reported
    dvs_fwd_state st{};
    bool have_fwd = false;
    bool keep_rows = false;              // parity tests: leave the A8 rows in place after the backward
    bool rows_clean = false;             // gradient rows are all-zero (k_preprocess_bwd re-zeroes what it reads)
    bool rows_pending = false;           // dvs_raster_backward_composite ran, dvs_raster_backward_project has not yet
    int64_t proj_next = 0;               // dvs_raster_backward_project_chunk: the next splat a chunk must start at
    // dvs_raster_forward_views_prepare: A2 of the NEXT forward already ran for the splats [0, prep_next) with exactly these inputs
    struct Prep {
        bool valid = false;
        int64_t next = 0;
        int n = 0, V = 0;
        dvs_splats p{};
        dvs_opts opts{};
        dvs_camera cams[DVS_MAX_VIEWS];
    } prep;
    int bwd_variant = DVS_BWD_TR;        // which A8 kernel (dvs_set_backward_variant; env DVS_BWD_VARIANT at create): the measured winner
    int fwd_variant = DVS_FWD_QUADRANT;  // which A7 kernel (dvs_set_forward_variant; env DVS_FWD_VARIANT at create)
    // stage timing: `timing` = every stage, synchronising per call (profiling iterations); `probe` = hipEvent pairs around the
    // composite kernels only, never synchronising — they are read back once, so the kernels are timed under the concurrency of
    // the real (pipelined) step
    bool timing = false;
    bool probe = false;
    std::vector<hipEvent_t> probe_ev;    // pairs: [2k] before, [2k+1] after
    std::vector<int> probe_kind;         // per pair: 0 = k_render_fwd, 1 = k_render_bwd
    size_t probe_used = 0;
    std::vector<hipEvent_t> events;
    std::vector<const char*> ev_names;
    size_t ev_used = 0;
    std::vector<const char*> out_names;
    std::vector<float> out_ms;
    std::vector<std::pair<const char*, std::pair<size_t, size_t>>> spans;   // name -> (event idx begin, end)
};

namespace {
struct StageTimer {
    dvs_ctx* c; hipStream_t st;
    StageTimer(dvs_ctx* c_, hipStream_t s) : c(c_), st(s) {}
    size_t mark() {
        if (!c->timing) return 0;
        if (c->ev_used == c->events.size()) { hipEvent_t e; (void)hipEventCreate(&e); c->events.push_back(e); }
        (void)hipEventRecord(c->events[c->ev_used], st);
        return c->ev_used++;
    }
    void span(const char* name, size_t a, size_t b) { if (c->timing) c->spans.push_back({name, {a, b}}); }
};
void timing_reset(dvs_ctx* c) { c->ev_used = 0; c->spans.clear(); }
hipEvent_t probe_event(dvs_ctx* c, hipStream_t st) {
    if (c->probe_used == c->probe_ev.size()) { hipEvent_t e; (void)hipEventCreate(&e); c->probe_ev.push_back(e); }
    hipEvent_t e = c->probe_ev[c->probe_used++];
    (void)hipEventRecord(e, st);
    return e;
}
void timing_collect(dvs_ctx* c, bool append) {
    if (!c->timing) return;
    if (!append) { c->out_names.clear(); c->out_ms.clear(); }
    for (auto& s : c->spans) {
        float ms = 0.f;
        (void)hipEventSynchronize(c->events[s.second.second]);
        (void)hipEventElapsedTime(&ms, c->events[s.second.first], c->events[s.second.second]);
        c->out_names.push_back(s.first); c->out_ms.push_back(ms);
    }
}

bool fe_no_key16() { static const bool v = [] { const char* e = getenv("DVS_FE_NO_KEY16"); return e && e[0] == '1'; }(); return v; }      // cross-check switch (tests): 32-bit tile ids
bool fe_no_fuse() { static const bool v = [] { const char* e = getenv("DVS_FE_NO_FUSE_A6"); return e && e[0] == '1'; }(); return v; }      // cross-check switch (tests): A6 as its own kernel
uint32_t* ranges_ptr(dvs_ctx* c) { return (uint32_t*)((char*)c->ranges.p + c->fe_zero_bytes); }
uint32_t* fe_kred(dvs_ctx* c) { return (uint32_t*)c->ranges.p; }
unsigned long long* fe_super(dvs_ctx* c) { return (unsigned long long*)((char*)c->ranges.p + (size_t)DVS_FE_KRED_WORDS * 4); }
DvsSeg* fe_seg_all(dvs_ctx* c) { return c->fe_state.as<DvsSeg>(); }
DvsSeg* fe_seg_vis(dvs_ctx* c) { return c->fe_state.as<DvsSeg>() + DVS_MAX_VIEWS; }
DvsSeg* fe_seg_tile(dvs_ctx* c) { return c->fe_state.as<DvsSeg>() + 2 * DVS_MAX_VIEWS; }
uint32_t* fe_superexcl(dvs_ctx* c) { return (uint32_t*)(c->fe_state.as<DvsSeg>() + 3 * DVS_MAX_VIEWS); }
uint32_t* fe_totals(dvs_ctx* c) { return fe_superexcl(c) + (size_t)c->max_views * c->fe_nsb; }
uint32_t* fe_block_sums(dvs_ctx* c) { return fe_totals(c) + (size_t)c->max_views * DVS_FE_MAXBINS; }
int ensure_frontend_arenas(dvs_ctx* c) {
    dvs_fe_block_counts((int)c->max_splats, &c->fe_nbv, &c->fe_nsb);
    c->fe_zero_bytes = (size_t)DVS_FE_KRED_WORDS * 4 + (size_t)c->max_views * c->fe_nsb * 8 * DVS_FE_SUPER_STRIDE;
    c->fe_zero_bytes = (c->fe_zero_bytes + 255) & ~(size_t)255;
    int r;
    const size_t state = 3 * DVS_MAX_VIEWS * sizeof(DvsSeg) + ((size_t)c->max_views * c->fe_nsb + (size_t)c->max_views * DVS_FE_MAXBINS + (size_t)c->max_views * c->fe_nbv) * 4;
    if ((r = c->fe_state.ensure(state)) != DVS_OK) return r;
    if ((r = c->fe_hist.ensure(dvs_fe_hist_words((uint64_t)c->max_splats * c->max_views, c->max_views, DVS_FE_MAXBINS) * 4)) != DVS_OK) return r;
    return DVS_OK;
}

int ensure_splat_arenas(dvs_ctx* c, size_t n) {
    int r;
#define ENS(buf, bytes) if ((r = c->buf.ensure(bytes)) != DVS_OK) return r;
    ENS(radii, n * 4) ENS(splat2d, n * 64) ENS(depth, n * 4)
    ENS(flags, n * 4) ENS(tiles_touched, n * 4) ENS(rect, n * 16) ENS(rect_sorted, n * 16)     /* 8 B per (view, splat) canonically; 16 B with DVS_TILES_TIGHT (rectangle + tile mask) */ ENS(key[0], n * 4) ENS(key[1], n * 4) ENS(ids[0], n * 4) ENS(ids[1], n * 4)
    ENS(g_rows, n * 48)
#undef ENS
    return DVS_OK;
}
int ensure_image_arenas(dvs_ctx* c, int w, int h, int views) {
    const size_t P = (size_t)w * h * views;
    const size_t tiles = (size_t)((w + DVS_TILE - 1) / DVS_TILE) * ((h + DVS_TILE - 1) / DVS_TILE) * views;
    int r;
    if ((r = c->ranges.ensure(c->fe_zero_bytes + tiles * 8)) != DVS_OK) return r;
    if ((r = c->ranges_canon.ensure(tiles * 8)) != DVS_OK) return r;
    if ((r = c->final_T.ensure(P * 4)) != DVS_OK) return r;
    if ((r = c->n_contrib.ensure(P * 4)) != DVS_OK) return r;
    return DVS_OK;
}
int ensure_instance_arenas(dvs_ctx* c, uint64_t T) {
    int r;
    if (c->inst_cap && T > c->inst_cap) ++c->inst_grow_events;
    for (int k = 0; k < 2; ++k) {
        if ((r = c->inst_tile[k].ensure(T * 4)) != DVS_OK) return r;
        if ((r = c->inst_splat[k].ensure(T * 4)) != DVS_OK) return r;
    }
    c->inst_cap = c->inst_tile[0].bytes / 4;
    for (int k = 0; k < 2; ++k) {
        if (c->inst_tile[k].bytes / 4 < c->inst_cap) c->inst_cap = c->inst_tile[k].bytes / 4;
        if (c->inst_splat[k].bytes / 4 < c->inst_cap) c->inst_cap = c->inst_splat[k].bytes / 4;
    }
    if (c->inst_cap >= (1ull << 32)) c->inst_cap = (1ull << 32) - 1;       // instance offsets are 32-bit
    if ((r = c->fe_hist.ensure(dvs_fe_hist_words(c->inst_cap, c->max_views, 512) * 4)) != DVS_OK) return r;
    return DVS_OK;
}
}  // namespace

extern "C" {

const char* dvs_last_error(void) { return g_last_error.c_str(); }
const char* dvs_version(void) { return "divshot_amd raster 0.1 (gfx950)"; }

dvs_ctx* dvs_create_views(int device, size_t max_splats, int max_w, int max_h, int max_views) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count) {
        g_last_error = "dvs_create: no such HIP device (the rasterizer has no CPU fallback)";
        return nullptr;
    }
    if (max_views < 1 || max_views > DVS_MAX_VIEWS) { g_last_error = "dvs_create_views: max_views must be in [1, 16]"; return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { set_error("hipSetDevice", e, __FILE__, __LINE__); return nullptr; }
    dvs_ctx* c = new dvs_ctx();
    c->device = device; c->max_splats = max_splats; c->max_w = max_w; c->max_h = max_h; c->max_views = max_views;
    // (A/B runs; the retired kernels exist in experiment builds only — elsewhere the request falls back to the default)
    if (const char* v = getenv("DVS_BWD_VARIANT")) {
        c->bwd_variant = v[0] == '0' ? DVS_BWD_BLOCKS : v[0] == '1' ? DVS_BWD_REDUCE : v[0] == '2' ? DVS_BWD_MM : DVS_BWD_TR;
        if ((c->bwd_variant == DVS_BWD_REDUCE || c->bwd_variant == DVS_BWD_MM) && !dvs_launch_render_bwd) c->bwd_variant = DVS_BWD_TR;
    }
    if (const char* v = getenv("DVS_FWD_VARIANT")) c->fwd_variant = (v[0] == '0' && dvs_launch_render_fwd_blocks) ? DVS_FWD_BLOCKS : DVS_FWD_QUADRANT;
    if (const char* v = getenv("DVS_LIVE_LISTS")) c->live_lists = v[0] != '0';
    if (hipMalloc((void**)&c->total_dev, 32) != hipSuccess || hipHostMalloc((void**)&c->total_host, 32, hipHostMallocDefault) != hipSuccess ||
        hipMemset(c->total_dev, 0, 32) != hipSuccess) {
        g_last_error = "dvs_create: hipMalloc failed";
        delete c;
        return nullptr;
    }
    c->total_host[0] = c->total_host[1] = c->total_host[2] = c->total_host[3] = 0;   // [0] T  [1] arena overflows
    {   // how the sorts' scatters rank inside a wave (frontend.hip): returning LDS adds, if this device serves the lanes of one address in
        // lane order — probed once per process and device, on the device; DVS_FE_RANK=ballot keeps the multisplit of rounds 2-5
        static int probed[64];                    // 0 unknown, 1 lane-ordered, 2 not
        static std::mutex probe_mutex;            // (hosts create scenes from worker threads: editor.cpp:2030)
        std::lock_guard<std::mutex> probe_lock(probe_mutex);
        const char* mode = getenv("DVS_FE_RANK");
        if (mode && mode[0] == 'b') c->fe_rank_atomic = 0;
        else {
            if (device < 64 && probed[device] == 0) {
                uint32_t bad = 1u;
                uint32_t* word = nullptr;
                if (hipMalloc((void**)&word, 4) == hipSuccess && hipMemset(word, 0, 4) == hipSuccess && dvs_fe_probe_rank_atomic(nullptr, word) == hipSuccess &&
                    hipMemcpy(&bad, word, 4, hipMemcpyDeviceToHost) == hipSuccess)
                    probed[device] = bad == 0u ? 1 : 2;
                if (word) (void)hipFree(word);
            }
            c->fe_rank_atomic = (device < 64 && probed[device] == 1) ? 1 : 0;
        }
    }
    if (const char* v = getenv("DVS_ASYNC")) c->async_T = v[0] == '1';
    if (ensure_frontend_arenas(c) != DVS_OK ||
        ensure_splat_arenas(c, max_splats * (size_t)max_views) != DVS_OK || ensure_image_arenas(c, max_w, max_h, max_views) != DVS_OK ||
        ensure_instance_arenas(c, (uint64_t)max_splats * 4 * (uint64_t)max_views) != DVS_OK) {
        dvs_destroy(c);
        return nullptr;
    }
    return c;
}
dvs_ctx* dvs_create(int device, size_t max_splats, int max_w, int max_h) { return dvs_create_views(device, max_splats, max_w, max_h, 1); }

void dvs_destroy(dvs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    Buf* all[] = {&c->radii, &c->splat2d, &c->depth, &c->flags, &c->tiles_touched, &c->rect, &c->rect_sorted, &c->key[0], &c->key[1],
                  &c->ids[0], &c->ids[1], &c->inst_tile[0], &c->inst_tile[1], &c->inst_splat[0], &c->inst_splat[1],
                  &c->tmp_keys, &c->tmp_vals, &c->ranges, &c->final_T, &c->n_contrib, &c->g_rows, &c->dcolor, &c->fe_state, &c->fe_hist, &c->ranges_canon};
    for (Buf* b : all) b->release();
    if (c->total_dev) (void)hipFree(c->total_dev);
    if (c->total_host) (void)hipHostFree(c->total_host);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->probe_ev) (void)hipEventDestroy(e);
    delete c;
}

// A2 for the splats [first, first + count) of all views (begin: the front end's zeroed state first — the key-range slots and super sums of
// frontend.hip and the tile ranges behind them, ONE memset)
static int run_a2(dvs_ctx* c, hipStream_t st, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts, int64_t first, int64_t count, bool begin) {
    const int n = p->n, W = cams[0].width, H = cams[0].height;
    const int tiles_x = (W + DVS_TILE - 1) / DVS_TILE, tiles_y = (H + DVS_TILE - 1) / DVS_TILE, tiles = tiles_x * tiles_y;
    const int tight = opts->tile_bounds == DVS_TILES_TIGHT ? 1 : 0;
    const int rect_fmt = tight ? DVS_FE_RECT_TIGHT : (tiles_x <= 255 && tiles_y <= 255) ? DVS_FE_RECT_U8 : DVS_FE_RECT_U16;
    DvsCams dcams;
    for (int v = 0; v < V; ++v) dcams.c[v] = to_dev_cam(cams[v]);
    if (begin) HIPCHECK(hipMemsetAsync(c->ranges.p, 0, c->fe_zero_bytes + (size_t)tiles * V * 8, st));
    HIPCHECK(dvs_launch_preprocess_fwd(st, n, p->pos, p->sh0, p->shN, p->opacity, p->scale, p->rot, dcams, V, opts->sh_degree,
                                       opts->antialias, tiles_x, tiles_y, c->radii.as<int>(), c->splat2d.as<float>(),
                                       c->depth.as<float>(), c->flags.as<uint32_t>(), c->tiles_touched.as<uint32_t>(), c->key[0].as<uint32_t>(),
                                       nullptr, opts->shn_layout, rect_fmt == DVS_FE_RECT_U16 ? c->rect.as<uint32_t>() : nullptr,
                                       rect_fmt == DVS_FE_RECT_TIGHT ? c->rect.as<uint32_t>() : nullptr,
                                       rect_fmt == DVS_FE_RECT_U8 ? c->rect.as<uint32_t>() : nullptr, fe_kred(c), (int)first, (int)count));
    return DVS_OK;
}
static bool prep_matches(const dvs_ctx* c, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts) {
    const dvs_ctx::Prep& q = c->prep;
    return q.valid && q.n == p->n && q.V == V && memcmp(&q.p, p, sizeof *p) == 0 && memcmp(&q.opts, opts, sizeof *opts) == 0 &&
           memcmp(q.cams, cams, sizeof(dvs_camera) * (size_t)V) == 0;
}

// A2..A7 for the n_views views of a batch (n_views = 1: the single-view API). Per-splat arrays are view-major [view][splat]; the
// sort value of an element is its global index view * n + splat; tile ids are view * tiles + tile: ONE depth sort, ONE scan, ONE
// duplication, ONE tile sort and ONE composite launch cover the whole iteration.
static int forward_views(dvs_ctx* c, hipStream_t st, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts,
                         float* out_rgb, dvs_fwd_state* saved, uint64_t* num_rendered) {
    const dvs_camera* cam = &cams[0];
    const int n = p->n, W = cam->width, H = cam->height;
    const int tiles_x = (W + DVS_TILE - 1) / DVS_TILE, tiles_y = (H + DVS_TILE - 1) / DVS_TILE, tiles = tiles_x * tiles_y;
    const size_t nV = (size_t)n * V;
    DvsCams dcams;
    float bgs[DVS_MAX_VIEWS * 3];
    for (int v = 0; v < V; ++v) { dcams.c[v] = to_dev_cam(cams[v]); for (int k = 0; k < 3; ++k) bgs[3 * v + k] = cams[v].bg[k]; }
    c->have_fwd = false;
    c->rows_pending = false;
    timing_reset(c);
    StageTimer tm(c, st);
    const int tight = opts->tile_bounds == DVS_TILES_TIGHT ? 1 : 0;      // opt-in: only the tiles the alpha >= 1/255 ellipse reaches (dvs_raster.h)

    uint64_t T = 0, T_expected = 0;
    const uint64_t* T_dev = nullptr;           // async: the kernels over instances read T on the device, grids sized for the capacity
    int icur = 0;
    size_t e7 = 0;
    int ranges_encoded = 0;                    // A6 fused into the tile sort's last pass: k_render_fwd decodes the ranges
    bool keys_written = true;                  // the sorted tile ids exist (an asynchronous forward with the fused A6 does not write them)
    if (c->async_T) {
        // No host synchronisation: the arenas are over-allocated, T stays on the device. What the host knows is the T of EARLIER
        // forwards on this context (pinned copy, refreshed asynchronously): it grows the arenas ahead of need and reports an overflow
        // (T beyond the capacity: that view's outputs are invalid, nothing was written out of bounds) as DVS_ERR_CAPACITY, once.
        const uint64_t lastT = c->total_host[0], overflow = c->total_host[1];
        if (overflow != c->overflow_seen) {
            c->overflow_seen = overflow;
            HIPCHECK(hipStreamSynchronize(st));
            int r = ensure_instance_arenas(c, lastT + lastT / 2);
            if (r != DVS_OK) return r;
            g_last_error = "dvs_raster_forward (async): an earlier forward on this context produced more tile instances than the instance "
                           "arena held; its outputs are invalid. The arena has been enlarged — repeat that view.";
            return DVS_ERR_CAPACITY;
        }
        if (lastT + lastT / 8 > c->inst_cap) {            // getting close: grow before it can overflow (rare; needs the stream idle)
            HIPCHECK(hipStreamSynchronize(st));
            int r = ensure_instance_arenas(c, lastT + lastT / 2);
            if (r != DVS_OK) return r;
        }
        T = c->inst_cap;
        T_dev = c->total_dev;
        T_expected = lastT > 0 ? lastT + lastT / 16 + 4096 : 0;      // grid size only: the kernels stride over whatever T turns out to be
    }
    {
        // ---- the segmented front end (frontend.hip): every view is a segment of the sorts, workgroup b works for view b % V ----
        const int rect_fmt = tight ? DVS_FE_RECT_TIGHT : (tiles_x <= 255 && tiles_y <= 255) ? DVS_FE_RECT_U8 : DVS_FE_RECT_U16;
        // A2 — unless dvs_raster_forward_views_prepare already ran it, chunk by chunk, for exactly these inputs (a data-parallel step
        // projects the next iteration's splats behind the optimizer's chunks while the gradient exchange is still on the links)
        const bool prepared = prep_matches(c, p, cams, V, opts) && c->prep.next == (int64_t)n;
        c->prep.valid = false;
        size_t e0 = tm.mark();
        if (!prepared) { int r = run_a2(c, st, p, cams, V, opts, 0, n, true); if (r != DVS_OK) return r; }
        size_t e1 = tm.mark(); tm.span("preprocess_fwd", e0, e1);
        if (n > 0) {
            // A5 (low 32 key bits): three range-adaptive passes per view; the culled splats leave in the first
            const uint32_t rows = dvs_depth_sort_rows_per_view(n, V);
            if (c->fe_seg_n != n || c->fe_seg_V != V || c->fe_seg_rows != rows) {
                HIPCHECK(dvs_launch_seg_init(st, n, V, rows, fe_seg_all(c)));
                c->fe_seg_n = n; c->fe_seg_V = V; c->fe_seg_rows = rows;
            }
            HIPCHECK(dvs_launch_depth_sort(st, n, V, c->key[0].as<uint32_t>(), c->ids[0].as<uint32_t>(), c->key[1].as<uint32_t>(), c->ids[1].as<uint32_t>(),
                                           fe_seg_all(c), fe_seg_vis(c), fe_kred(c), c->fe_hist.as<uint32_t>(), fe_totals(c), c->fe_rank_atomic));
        }
        size_t e2 = tm.mark(); tm.span("depth_sort", e1, e2);
        // A3: tile counts in depth order (the one random gather: the tile rectangles), the views' instance ranges; T stays on the device
        const uint64_t lastT = c->total_host[0];
        const uint64_t t_hint = T_expected ? T_expected : (lastT ? lastT : (uint64_t)nV * 3);
        const uint32_t tile_part = dvs_fe_part_for(t_hint);
        if (n > 0)
            HIPCHECK(dvs_launch_seg_binning(st, n, V, rect_fmt, fe_seg_vis(c), fe_seg_tile(c), c->ids[1].as<uint32_t>(), c->rect.as<uint32_t>(),
                                            c->rect_sorted.as<uint32_t>(), fe_block_sums(c), fe_super(c), fe_superexcl(c), tile_part,
                                            (unsigned long long*)c->total_dev, c->async_T ? c->inst_cap : ~0ull, 0, tiles_x, nullptr, nullptr));
        else
            HIPCHECK(hipMemsetAsync(c->total_dev, 0, 8, st));
        HIPCHECK(hipMemcpyAsync(c->total_host, c->total_dev, 32, hipMemcpyDeviceToHost, st));
        size_t e3 = tm.mark(); tm.span("tile_scan", e2, e3);
        if (!c->async_T) {
            HIPCHECK(hipStreamSynchronize(st));
            T = c->total_host[0];
            if (T >= (1ull << 32)) { g_last_error = "dvs_raster_forward: more than 2^32 tile instances"; return DVS_ERR_CAPACITY; }
            { int r = ensure_instance_arenas(c, T ? T : 1); if (r != DVS_OK) return r; }
        }
        // A4 duplicate, view by view. Tile ids inside a view fit 16 bits up to 65 536 tiles (4096 x 4096 pixels): A4, the first
        // histogram and the first tile pass then move 2 B per instance for them instead of 4.
        const int key16 = (tiles <= 65536 && !fe_no_key16()) ? 1 : 0;
        size_t e4 = tm.mark();
        if (n > 0)
            HIPCHECK(dvs_launch_seg_binning(st, n, V, rect_fmt, fe_seg_vis(c), fe_seg_tile(c), c->ids[1].as<uint32_t>(), c->rect.as<uint32_t>(),
                                            c->rect_sorted.as<uint32_t>(), fe_block_sums(c), fe_super(c), fe_superexcl(c), tile_part,
                                            (unsigned long long*)c->total_dev, c->inst_cap, 1, tiles_x, c->inst_tile[0].as<uint32_t>(),
                                            c->inst_splat[0].as<uint32_t>(), key16));
        size_t e5 = tm.mark(); tm.span("duplicate", e4, e5);
        // A5 (high key bits): every view's instances by tile id; the last pass hands out view * tiles + tile
        // A6 rides on the last pass (a range boundary is where the scattered tile id changes) whenever k_render_fwd composites (it decodes
        // the ranges). An asynchronous forward does not even write the sorted tile ids: nothing on the device reads them, and
        // dvs_fwd_state.sorted_tile is then NULL (dvs_raster.h).
        const bool fuse_a6 = (c->fwd_variant == DVS_FWD_QUADRANT || V > 1) && !fe_no_fuse();
        const bool write_keys = !(fuse_a6 && c->async_T) || c->export_tiles;
        if (n > 0) {
            const uint64_t cap = c->async_T ? c->inst_cap : T;
            const uint32_t nbtot = (uint32_t)(cap / tile_part) + (uint32_t)V + 2u;
            HIPCHECK(dvs_launch_seg_sort(st, V, c->inst_tile[0].as<uint32_t>(), c->inst_splat[0].as<uint32_t>(), c->inst_tile[1].as<uint32_t>(),
                                         c->inst_splat[1].as<uint32_t>(), fe_seg_tile(c), 0, tiles > 1 ? bits_for((uint32_t)(tiles - 1)) : 1,
                                         c->async_T ? (T_expected ? T_expected : c->inst_cap) : T, tile_part, nbtot, c->fe_hist.as<uint32_t>(), fe_totals(c),
                                         (uint32_t)tiles, &icur, fuse_a6 ? ranges_ptr(c) : nullptr, write_keys ? 1 : 0, key16, c->fe_rank_atomic));
        }
        size_t e6 = tm.mark(); tm.span("tile_sort", e5, e6);
        if (fuse_a6) ranges_encoded = n > 0 ? 1 : 0;
        else HIPCHECK(dvs_launch_tile_ranges(st, T, c->inst_tile[icur].as<uint32_t>(), ranges_ptr(c), tiles * V, T_dev, T_expected, false));     // (cleared by the memset above)
        keys_written = write_keys;
        e7 = tm.mark(); tm.span("tile_ranges", e6, e7);
    }
    // A7 composite
    if (c->probe) { (void)probe_event(c, st); c->probe_kind.push_back(0); }
    // the live lists of A7 (entries that reach their tile, compacted) go into the sort's other pair of instance arrays, free by now
    c->live_splat = nullptr; c->live_pos = nullptr;
    uint64_t* rec_masks = nullptr; uint64_t rec_cap = 0;          // dvs_debug_record_decisions (parity tests): one view, synchronous T
    if (c->rec_masks && V == 1 && !c->async_T && (c->fwd_variant == DVS_FWD_QUADRANT)) {
        rec_masks = c->rec_masks; rec_cap = c->rec_cap;
        HIPCHECK(hipMemsetAsync(rec_masks, 0, (size_t)(T < rec_cap ? T : rec_cap) * 32, st));
    }
    if (c->fwd_variant == DVS_FWD_QUADRANT || V > 1) {
        // Only the "tr" composite backward walks them (the other variants ignore them), and they cost the forward two 4-B stores per
        // instance: none for an inference-only context (dvs_set_live_lists(ctx, 0)) or another backward. The spare pair of the tile
        // sort's ping-pong buffers is RESERVED for them until the next forward on this context: nothing after the sort may reuse
        // inst_*[icur ^ 1].
        if (c->live_lists && c->bwd_variant == DVS_BWD_TR) { c->live_splat = c->inst_splat[icur ^ 1].as<uint32_t>(); c->live_pos = c->inst_tile[icur ^ 1].as<uint32_t>(); }
        HIPCHECK(dvs_launch_render_fwd(st, W, H, tiles_x, tiles_y, V, ranges_ptr(c), c->inst_splat[icur].as<uint32_t>(),
                                       c->splat2d.as<float>(), bgs, out_rgb, c->final_T.as<float>(), c->n_contrib.as<uint32_t>(),
                                       c->live_splat, c->live_pos, rec_masks, rec_cap, ranges_encoded ? c->ranges_canon.as<uint32_t>() : nullptr));
    } else
        HIPCHECK(dvs_launch_render_fwd_blocks(st, W, H, tiles_x, tiles_y, ranges_ptr(c), c->inst_splat[icur].as<uint32_t>(),
                                              c->splat2d.as<float>(), cam->bg, out_rgb,
                                              c->final_T.as<float>(), c->n_contrib.as<uint32_t>()));
    if (c->probe) (void)probe_event(c, st);
    size_t e8 = tm.mark(); tm.span("render_fwd", e7, e8);

    dvs_fwd_state& s = c->st;
    s.radii = c->radii.as<int32_t>(); s.splat2d = c->splat2d.as<float>(); s.depth = c->depth.as<float>();
    s.flags = c->flags.as<uint32_t>();
    s.tiles_touched = c->tiles_touched.as<uint32_t>();
    s.sorted_tile = keys_written ? c->inst_tile[icur].as<uint32_t>() : nullptr; s.sorted_splat = c->inst_splat[icur].as<uint32_t>();
    s.ranges = ranges_encoded ? c->ranges_canon.as<uint32_t>() : ranges_ptr(c); s.final_T = c->final_T.as<float>(); s.n_contrib = c->n_contrib.as<uint32_t>();
    s.num_rendered = c->async_T ? DVS_T_UNKNOWN : T; s.n = n; s.width = W; s.height = H; s.tiles_x = tiles_x; s.tiles_y = tiles_y;
    s._pad = 0;
    c->n_views = V;
    c->have_fwd = true;
    if (saved) *saved = s;
    if (num_rendered) *num_rendered = c->async_T ? DVS_T_UNKNOWN : T;
    if (c->timing) { HIPCHECK(hipStreamSynchronize(st)); timing_collect(c, false); }
    return DVS_OK;
}

static int check_fwd_args(dvs_ctx* c, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts, const float* out_rgb) {
    if (!c || !p || !cams || !opts || !out_rgb) { g_last_error = "dvs_raster_forward: null argument"; return DVS_ERR_INVALID; }
    if (V < 1 || V > c->max_views) { g_last_error = "dvs_raster_forward: n_views exceeds the max_views given to dvs_create_views"; return DVS_ERR_CAPACITY; }
    const dvs_camera* cam = &cams[0];
    if (p->n < 0 || cam->width <= 0 || cam->height <= 0 || opts->sh_degree < 0 || opts->sh_degree > 3 ||
        (opts->shn_layout != DVS_SHN_ROWS && opts->shn_layout != DVS_SHN_TILED) ||
        (opts->tile_bounds != DVS_TILES_CANONICAL && opts->tile_bounds != DVS_TILES_TIGHT)) {
        g_last_error = "dvs_raster_forward: bad n / image size / sh_degree / shn_layout / tile_bounds"; return DVS_ERR_INVALID;
    }
    for (int v = 1; v < V; ++v)
        if (cams[v].width != cam->width || cams[v].height != cam->height) { g_last_error = "dvs_raster_forward: the views of a batch must share one image size"; return DVS_ERR_INVALID; }
    if (p->n > 0 && (((uintptr_t)p->pos | (uintptr_t)p->sh0 | (uintptr_t)p->shN | (uintptr_t)p->opacity | (uintptr_t)p->scale | (uintptr_t)p->rot) & 15u)) {
        g_last_error = "dvs_raster_forward: parameter arrays must be 16-byte aligned"; return DVS_ERR_INVALID;
    }
    if ((size_t)p->n > c->max_splats || cam->width > c->max_w || cam->height > c->max_h) {
        g_last_error = "dvs_raster_forward: exceeds the capacity given to dvs_create"; return DVS_ERR_CAPACITY;
    }
    // (the scan / duplication launchers and kernels index the (view, splat) elements with int)
    if ((uint64_t)p->n * (uint64_t)V >= (1ull << 31)) { g_last_error = "dvs_raster_forward: n * n_views must stay below 2^31"; return DVS_ERR_CAPACITY; }
    return DVS_OK;
}

int dvs_raster_forward(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cam, const dvs_opts* opts,
                       float* out_rgb, dvs_fwd_state* saved, uint64_t* num_rendered) {
    int r = check_fwd_args(c, p, cam, 1, opts, out_rgb);
    if (r != DVS_OK) return r;
    HIPCHECK(hipSetDevice(c->device));
    return forward_views(c, (hipStream_t)stream, p, cam, 1, opts, out_rgb, saved, num_rendered);
}

int dvs_raster_forward_views(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cams, int n_views, const dvs_opts* opts,
                             float* out_rgb) {
    int r = check_fwd_args(c, p, cams, n_views, opts, out_rgb);
    if (r != DVS_OK) return r;
    HIPCHECK(hipSetDevice(c->device));
    return forward_views(c, (hipStream_t)stream, p, cams, n_views, opts, out_rgb, nullptr, nullptr);
}

int dvs_raster_forward_cancel_prepared(dvs_ctx* c) {
    if (!c) { g_last_error = "dvs_raster_forward_cancel_prepared: null context"; return DVS_ERR_INVALID; }
    c->prep.valid = false;
    return DVS_OK;
}
int dvs_raster_forward_views_prepare(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cams, int n_views, const dvs_opts* opts,
                                     int64_t first, int64_t count) {
    static float dummy_rgb;                                       // (the argument check wants a non-null image pointer)
    int r = check_fwd_args(c, p, cams, n_views, opts, &dummy_rgb);
    if (r != DVS_OK) return r;
    if (opts->shn_layout != DVS_SHN_TILED) { g_last_error = "dvs_raster_forward_views_prepare: needs the DVS_SHN_TILED layout"; return DVS_ERR_INVALID; }
    const bool begin = first == 0;
    if (!begin && !(prep_matches(c, p, cams, n_views, opts) && c->prep.next == first)) {
        c->prep.valid = false;
        g_last_error = "dvs_raster_forward_views_prepare: chunks must cover [0, n) in ascending order with the same parameters, cameras and options";
        return DVS_ERR_STATE;
    }
    if (count <= 0 || first < 0 || first + count > p->n || (first % 256) != 0) {
        c->prep.valid = false;
        g_last_error = "dvs_raster_forward_views_prepare: bad chunk (each starts at a multiple of 256 and lies inside [0, n))"; return DVS_ERR_INVALID;
    }
    HIPCHECK(hipSetDevice(c->device));
    if (begin) {
        // the state of the previous forward is overwritten from here on: its backward must have been queued already (no backward after this)
        c->have_fwd = false; c->rows_pending = false;
        c->prep.valid = true; c->prep.next = 0; c->prep.n = p->n; c->prep.V = n_views; c->prep.p = *p; c->prep.opts = *opts;
        memcpy(c->prep.cams, cams, sizeof(dvs_camera) * (size_t)n_views);
    }
    if ((r = run_a2(c, (hipStream_t)stream, p, cams, n_views, opts, first, count, begin)) != DVS_OK) { c->prep.valid = false; return r; }
    c->prep.next = first + count;
    return DVS_OK;
}

int dvs_get_view_state(dvs_ctx* c, int view, dvs_fwd_state* out) {
    if (!c || !out) { g_last_error = "dvs_get_view_state: null argument"; return DVS_ERR_INVALID; }
    if (!c->have_fwd || view < 0 || view >= c->n_views) { g_last_error = "dvs_get_view_state: no such view in the last forward"; return DVS_ERR_STATE; }
    dvs_fwd_state s = c->st;
    const size_t on = (size_t)view * s.n, op = (size_t)view * s.width * s.height, ot = (size_t)view * s.tiles_x * s.tiles_y;
    s.radii += on; s.splat2d += on * DVS_S2D_FLOATS; s.depth += on; s.flags += on; s.tiles_touched += on;
    s.ranges += 2 * ot; s.final_T += op; s.n_contrib += op;
    *out = s;              // sorted_tile / sorted_splat stay the batch-wide lists: ranges index into them, values are view * n + splat
    return DVS_OK;
}

// A8: zero the 48-B rows if needed, then the alpha-composite backward into them (all views of the last forward in one launch)
static int bwd_composite(dvs_ctx* c, hipStream_t st, const dvs_camera* cams, const dvs_opts* opts, const float* dL_drgb, StageTimer* tm) {
    const dvs_fwd_state& s = c->st;
    const int V = c->n_views;
    size_t e0 = tm ? tm->mark() : 0;
    if (!c->rows_clean) HIPCHECK(hipMemsetAsync(c->g_rows.p, 0, c->g_rows.bytes, st));
    c->rows_clean = false;
    size_t e1 = tm ? tm->mark() : 0;
    if (tm) tm->span("bwd_zero", e0, e1);
    float bgs[DVS_MAX_VIEWS * 3];
    for (int v = 0; v < V; ++v) for (int k = 0; k < 3; ++k) bgs[3 * v + k] = cams[v].bg[k];
    if (c->probe) { (void)probe_event(c, st); c->probe_kind.push_back(1); }
    if (c->bwd_variant == DVS_BWD_TR)
        HIPCHECK(dvs_launch_render_bwd_tr(st, s.width, s.height, s.tiles_x, s.tiles_y, V, s.ranges, s.sorted_splat, s.splat2d, bgs, s.final_T,
                                          s.n_contrib, dL_drgb, c->g_rows.as<float>(), opts->absgrad, opts->grad_mode, c->live_splat, c->live_pos));
    else if (c->bwd_variant == DVS_BWD_BLOCKS)
        HIPCHECK(dvs_launch_render_bwd_blocks(st, s.width, s.height, s.tiles_x, s.tiles_y, V, s.ranges, s.sorted_splat, s.splat2d,
                                              bgs, s.final_T, s.n_contrib, dL_drgb, c->g_rows.as<float>(), opts->absgrad, opts->grad_mode));
    else {                                                 // experiment builds: the retired kernels (the mm experiment renders one view)
        if (!dvs_launch_render_bwd) { g_last_error = "dvs_raster_backward: this build does not contain the selected composite-backward variant"; return DVS_ERR_INVALID; }
        HIPCHECK(dvs_launch_render_bwd(st, s.width, s.height, s.tiles_x, s.tiles_y, V, s.ranges, s.sorted_splat, s.splat2d,
                                       bgs, s.final_T, s.n_contrib, dL_drgb, c->g_rows.as<float>(), opts->absgrad, opts->grad_mode,
                                       V > 1 ? DVS_BWD_REDUCE : c->bwd_variant));
    }
    if (c->probe) (void)probe_event(c, st);
    if (tm) { size_t e2 = tm->mark(); tm->span("render_bwd", e1, e2); }
    c->rows_pending = true;
    c->proj_next = 0;
    return DVS_OK;
}
// A9: rows -> parameter gradients (re-zeroes the rows it reads). The tiled layout — one view or a batch — goes through ONE pass that
// reads the parameters once and writes the geometry gradients once (+ the per-view colour gradients), the SH rows are then built from
// those (also for a single view: that pass plus the row rebuild is faster than the fused per-view kernel, 0.09 vs 0.15 ms at C3);
// the reference's row layout goes through the per-view kernel, once per view, accumulating.
static int bwd_project(dvs_ctx* c, hipStream_t st, const dvs_splats* p, const dvs_camera* cams, const dvs_opts* opts,
                       const dvs_splat_grads* out, StageTimer* tm) {
    const dvs_fwd_state& s = c->st;
    const int V = c->n_views, n = p->n;
    size_t e2 = tm ? tm->mark() : 0;
    if (opts->shn_layout == DVS_SHN_TILED) {
        DvsCams dcams;
        float campos[DVS_MAX_VIEWS * 3];
        for (int v = 0; v < V; ++v) { dcams.c[v] = to_dev_cam(cams[v]); for (int k = 0; k < 3; ++k) campos[3 * v + k] = cams[v].campos[k]; }
        // Nobody asked for the per-view colour gradients (one GPU, or the plain all-reduce exchange): the kernel builds the SH rows in its
        // epilogue (k_preprocess_bwd_views<.., FUSE_SH>). With out->dcolor (the factorised exchange, dvs_raster_backward_dcolor's twin) it
        // emits them and dvs_launch_sh_grad_combine builds the rows — here right away if sh0 / shN are given too, else after the all-gather.
        const char* nf = getenv("DVS_A9_NO_FUSE_SH");                      // cross-check switch (tests; read per call so that one process can run both)
        const bool no_fuse = nf && nf[0] == '1';
        const bool fuse = !out->dcolor && out->sh0 && out->shN && !no_fuse;
        float* dcol = out->dcolor;
        if (!dcol && !fuse) { int r = c->dcolor.ensure((size_t)V * n * 3 * sizeof(float)); if (r != DVS_OK) return r; dcol = c->dcolor.as<float>(); }
        HIPCHECK(dvs_launch_preprocess_bwd_views(st, n, V, p->pos, p->shN, p->opacity, p->scale, p->rot, dcams, opts->sh_degree,
                                                 opts->antialias, s.radii, s.flags, c->g_rows.as<float>(), out->pos, out->opacity, out->scale,
                                                 out->rot, opts->absgrad ? out->absgrad2d : nullptr, out->mean2d, dcol, opts->accumulate,
                                                 c->keep_rows ? 0 : 1, opts->grad_mode, 0, -1, fuse ? out->sh0 : nullptr, fuse ? out->shN : nullptr));
        if (!fuse && out->sh0 && out->shN)          // (NULL: the factorised exchange builds them after its all-gather of dcolor)
            HIPCHECK(dvs_launch_sh_grad_combine(st, n, p->pos, opts->sh_degree, V, campos, dcol, out->sh0, out->shN, opts->accumulate, 1));
    } else {
        for (int v = 0; v < V; ++v) {
            const DvsCam dcam = to_dev_cam(cams[v]);
            const size_t on = (size_t)v * n;
            HIPCHECK(dvs_launch_preprocess_bwd(st, n, p->pos, p->shN, p->opacity, p->scale, p->rot, dcam, opts->sh_degree, opts->antialias,
                                               s.radii + on, s.flags + on, c->g_rows.as<float>() + on * 12, out->pos, out->sh0, out->shN,
                                               out->opacity, out->scale, out->rot, opts->absgrad ? out->absgrad2d : nullptr, out->mean2d,
                                               out->dcolor ? out->dcolor + on * 3 : nullptr, (opts->accumulate || v > 0) ? 1 : 0,
                                               c->keep_rows ? 0 : 1, opts->shn_layout, opts->grad_mode));
        }
    }
    c->rows_clean = !c->keep_rows;        // every row render_bwd can have touched (radius > 0) was read and re-zeroed
    c->rows_pending = c->keep_rows;       // (kept rows can be projected again: the parity tests compare chunked and unchunked A9 on the same rows)
    c->proj_next = 0;
    if (tm) { size_t e3 = tm->mark(); tm->span("preprocess_bwd", e2, e3); }
    return DVS_OK;
}
static int check_bwd_args(dvs_ctx* c, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts, const char* who) {
    static thread_local std::string msg;
    if (!c || !cams || !opts) { msg = std::string(who) + ": null argument"; g_last_error = msg.c_str(); return DVS_ERR_INVALID; }
    if (opts->grad_mode != DVS_GRAD_TRUE && opts->grad_mode != DVS_GRAD_LINEAGE) { msg = std::string(who) + ": bad grad_mode"; g_last_error = msg.c_str(); return DVS_ERR_INVALID; }
    if (!c->have_fwd || V != c->n_views || (p && c->st.n != p->n) || c->st.width != cams[0].width || c->st.height != cams[0].height) {
        msg = std::string(who) + ": no matching forward on this context"; g_last_error = msg.c_str(); return DVS_ERR_STATE;
    }
    return DVS_OK;
}
static int check_grads(const dvs_splats* p, const dvs_splat_grads* out, const char* who) {
    static thread_local std::string msg;
    if (!p || !out) { msg = std::string(who) + ": null argument"; g_last_error = msg.c_str(); return DVS_ERR_INVALID; }
    if (p->n > 0 && (!out->pos || !out->opacity || !out->scale || !out->rot || ((!out->sh0 || !out->shN) && !out->dcolor))) {
        msg = std::string(who) + ": null gradient row pointer (sh0/shN may be NULL only when dcolor is given)";
        g_last_error = msg.c_str(); return DVS_ERR_INVALID;
    }
    if (p->n > 0 && (((uintptr_t)out->pos | (uintptr_t)out->sh0 | (uintptr_t)out->shN | (uintptr_t)out->opacity | (uintptr_t)out->scale |
                      (uintptr_t)out->rot | (uintptr_t)out->absgrad2d | (uintptr_t)out->mean2d | (uintptr_t)out->dcolor) & 15u)) {
        msg = std::string(who) + ": gradient arrays must be 16-byte aligned"; g_last_error = msg.c_str(); return DVS_ERR_INVALID;
    }
    return DVS_OK;
}

static int backward_views(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cams, int V, const dvs_opts* opts,
                          const float* dL_drgb, const dvs_splat_grads* out, const char* who) {
    int r;
    if (!dL_drgb) { g_last_error = "dvs_raster_backward: null argument"; return DVS_ERR_INVALID; }
    if ((uintptr_t)dL_drgb & 15u) { g_last_error = "dvs_raster_backward: dL_drgb must be 16-byte aligned (the composite backward reads it as 16-byte words)"; return DVS_ERR_INVALID; }
    if ((r = check_grads(p, out, who)) != DVS_OK) return r;
    if ((r = check_bwd_args(c, p, cams, V, opts, who)) != DVS_OK) return r;
    HIPCHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    timing_reset(c);
    StageTimer tm(c, st);
    if ((r = bwd_composite(c, st, cams, opts, dL_drgb, &tm)) != DVS_OK) return r;
    if ((r = bwd_project(c, st, p, cams, opts, out, &tm)) != DVS_OK) return r;
    if (c->timing) { HIPCHECK(hipStreamSynchronize(st)); timing_collect(c, true); }
    return DVS_OK;
}
int dvs_raster_backward(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cam, const dvs_opts* opts,
                        const float* dL_drgb, const dvs_splat_grads* out) {
    return backward_views(c, stream, p, cam, 1, opts, dL_drgb, out, "dvs_raster_backward");
}
int dvs_raster_backward_views(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cams, int n_views, const dvs_opts* opts,
                              const float* dL_drgb, const dvs_splat_grads* out) {
    return backward_views(c, stream, p, cams, n_views, opts, dL_drgb, out, "dvs_raster_backward_views");
}

int dvs_raster_backward_composite(dvs_ctx* c, void* stream, const dvs_camera* cam, const dvs_opts* opts, const float* dL_drgb) {
    int r;
    if (!dL_drgb) { g_last_error = "dvs_raster_backward_composite: null argument"; return DVS_ERR_INVALID; }
    if ((uintptr_t)dL_drgb & 15u) { g_last_error = "dvs_raster_backward_composite: dL_drgb must be 16-byte aligned (the composite backward reads it as 16-byte words)"; return DVS_ERR_INVALID; }
    if (!c) { g_last_error = "dvs_raster_backward_composite: null argument"; return DVS_ERR_INVALID; }
    if ((r = check_bwd_args(c, nullptr, cam, c->n_views, opts, "dvs_raster_backward_composite")) != DVS_OK) return r;
    HIPCHECK(hipSetDevice(c->device));
    timing_reset(c);
    StageTimer tm(c, (hipStream_t)stream);
    if ((r = bwd_composite(c, (hipStream_t)stream, cam, opts, dL_drgb, &tm)) != DVS_OK) return r;
    if (c->timing) { HIPCHECK(hipStreamSynchronize((hipStream_t)stream)); timing_collect(c, true); }
    return DVS_OK;
}

int dvs_raster_backward_project(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cam, const dvs_opts* opts,
                                const dvs_splat_grads* out) {
    int r;
    if ((r = check_grads(p, out, "dvs_raster_backward_project")) != DVS_OK) return r;
    if (!c) { g_last_error = "dvs_raster_backward_project: null argument"; return DVS_ERR_INVALID; }
    if ((r = check_bwd_args(c, p, cam, c->n_views, opts, "dvs_raster_backward_project")) != DVS_OK) return r;
    if (!c->rows_pending) { g_last_error = "dvs_raster_backward_project: no dvs_raster_backward_composite on this context"; return DVS_ERR_STATE; }
    if (c->proj_next != 0) {       // earlier chunks already consumed (and re-zeroed) their rows: projecting all of [0, n) now would overwrite their gradients with zeros
        g_last_error = "dvs_raster_backward_project: a dvs_raster_backward_project_chunk sequence is in progress on this context (finish it up to n)";
        return DVS_ERR_STATE;
    }
    HIPCHECK(hipSetDevice(c->device));
    timing_reset(c);
    StageTimer tm(c, (hipStream_t)stream);
    if ((r = bwd_project(c, (hipStream_t)stream, p, cam, opts, out, &tm)) != DVS_OK) return r;
    if (c->timing) { HIPCHECK(hipStreamSynchronize((hipStream_t)stream)); timing_collect(c, true); }
    return DVS_OK;
}

int dvs_raster_backward_project_chunk(dvs_ctx* c, void* stream, const dvs_splats* p, const dvs_camera* cam, const dvs_opts* opts,
                                      const dvs_splat_grads* out, int64_t first, int64_t count) {
    int r;
    if ((r = check_grads(p, out, "dvs_raster_backward_project_chunk")) != DVS_OK) return r;
    if (!c) { g_last_error = "dvs_raster_backward_project_chunk: null argument"; return DVS_ERR_INVALID; }
    if ((r = check_bwd_args(c, p, cam, c->n_views, opts, "dvs_raster_backward_project_chunk")) != DVS_OK) return r;
    if (!c->rows_pending) { g_last_error = "dvs_raster_backward_project_chunk: no dvs_raster_backward_composite pending on this context"; return DVS_ERR_STATE; }
    if (opts->shn_layout != DVS_SHN_TILED || !out->dcolor || out->sh0 || out->shN) {
        g_last_error = "dvs_raster_backward_project_chunk: needs the DVS_SHN_TILED layout and the factorised form (out->dcolor given, sh0 / shN NULL)";
        return DVS_ERR_INVALID;
    }
    if (first != c->proj_next || count <= 0 || first + count > p->n || (first % 256) != 0) {
        g_last_error = "dvs_raster_backward_project_chunk: chunks must cover [0, n) in ascending order, each starting at a multiple of 256";
        return DVS_ERR_INVALID;
    }
    HIPCHECK(hipSetDevice(c->device));
    const dvs_fwd_state& s = c->st;
    const int V = c->n_views, n = p->n;
    DvsCams dcams;
    for (int v = 0; v < V; ++v) dcams.c[v] = to_dev_cam(cam[v]);
    timing_reset(c);
    StageTimer tm(c, (hipStream_t)stream);                                  // (every chunk is a "preprocess_bwd" row of the timing table)
    const size_t t0 = tm.mark();
    HIPCHECK(dvs_launch_preprocess_bwd_views((hipStream_t)stream, n, V, p->pos, p->shN, p->opacity, p->scale, p->rot, dcams, opts->sh_degree,
                                             opts->antialias, s.radii, s.flags, c->g_rows.as<float>(), out->pos, out->opacity, out->scale,
                                             out->rot, opts->absgrad ? out->absgrad2d : nullptr, out->mean2d, out->dcolor, opts->accumulate,
                                             c->keep_rows ? 0 : 1, opts->grad_mode, (int)first, (int)count));
    { const size_t t1 = tm.mark(); tm.span("preprocess_bwd", t0, t1); }
    if (c->timing) { HIPCHECK(hipStreamSynchronize((hipStream_t)stream)); timing_collect(c, true); }
    c->proj_next = first + count;
    if (c->proj_next == n) { c->rows_clean = !c->keep_rows; c->rows_pending = c->keep_rows; c->proj_next = 0; }
    return DVS_OK;
}

int dvs_sort_pairs_u32(dvs_ctx* c, void* stream, uint32_t* keys, uint32_t* vals, uint64_t n, int bit_lo, int bit_hi) {
    if (!c || (n > 0 && (!keys || !vals)) || bit_lo < 0 || bit_hi > 32 || bit_lo > bit_hi) { g_last_error = "dvs_sort_pairs_u32: bad argument"; return DVS_ERR_INVALID; }
    HIPCHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return DVS_OK;
    int r;
    if ((r = c->tmp_keys.ensure(n * 4)) != DVS_OK) return r;
    if ((r = c->tmp_vals.ensure(n * 4)) != DVS_OK) return r;
    uint32_t* k[2] = {keys, c->tmp_keys.as<uint32_t>()};
    uint32_t* v[2] = {vals, c->tmp_vals.as<uint32_t>()};
    int cur = 0;
    {
        // (partition offsets are 32-bit: the last partition's element indices must not wrap — ADVICE r05)
        if (n > (1ull << 32) - 4097ull) { g_last_error = "dvs_sort_pairs_u32: n must be at most 2^32 - 4097"; return DVS_ERR_CAPACITY; }
        if ((r = c->fe_hist.ensure(dvs_fe_hist_words(n, 1, 512) * 4)) != DVS_OK) return r;
        const uint32_t part = dvs_fe_part_for(n);
        HIPCHECK(dvs_launch_seg_init(st, (int)n, 1, 0, fe_seg_all(c)));      // one segment: [0, n)
        c->fe_seg_n = -1;                                                    // (the forward's descriptors are gone)
        HIPCHECK(dvs_launch_seg_sort(st, 1, k[0], v[0], k[1], v[1], fe_seg_all(c), bit_lo, bit_hi - bit_lo, n, part, (uint32_t)(n / part) + 3u,
                                     c->fe_hist.as<uint32_t>(), fe_totals(c), 0u, &cur, nullptr, 1, 0, c->fe_rank_atomic));
    }
    if (cur == 1) {
        HIPCHECK(hipMemcpyAsync(keys, k[1], n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(hipMemcpyAsync(vals, v[1], n * 4, hipMemcpyDeviceToDevice, st));
    }
    return DVS_OK;
}

// TEST HOOK (dvs_raster.h): the forward's range-adaptive depth sort (A5, low 32 key bits) on caller-supplied keys, one view.
int dvs_debug_sort_depth_keys(dvs_ctx* c, void* stream, const uint32_t* keys, uint64_t n, uint32_t* sorted_ids, uint32_t* n_sorted, uint32_t* digit_bits) {
    if (!c || !keys || !sorted_ids || !n_sorted || n == 0 || n > c->max_splats * (uint64_t)c->max_views || n >= (1ull << 31)) {
        g_last_error = "dvs_debug_sort_depth_keys: bad argument (n must be in [1, max_splats * max_views])"; return DVS_ERR_INVALID;
    }
    HIPCHECK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    // the key range a forward's A2 would have left in the view's 64 slots: here reduced on the host from a copy of the keys
    std::vector<uint32_t> h(n);
    HIPCHECK(hipMemcpyAsync(h.data(), keys, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    uint32_t mn = 0xFFFFFFFFu, mx = 0u; bool any = false;
    for (uint32_t k : h) if (k != 0xFFFFFFFFu) { mn = k < mn ? k : mn; mx = k > mx ? k : mx; any = true; }
    uint32_t slots[64 * 16] = {0};
    if (any) { slots[0] = ~mn; slots[1] = mx; }
    HIPCHECK(hipMemcpyAsync(fe_kred(c), slots, sizeof slots, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(c->key[0].p, keys, n * 4, hipMemcpyDeviceToDevice, st));
    const uint32_t rows = dvs_depth_sort_rows_per_view((int)n, 1);
    HIPCHECK(dvs_launch_seg_init(st, (int)n, 1, rows, fe_seg_all(c)));
    c->fe_seg_n = -1;                                                       // (the forward's descriptors are gone)
    HIPCHECK(dvs_launch_depth_sort(st, (int)n, 1, c->key[0].as<uint32_t>(), c->ids[0].as<uint32_t>(), c->key[1].as<uint32_t>(), c->ids[1].as<uint32_t>(),
                                   fe_seg_all(c), fe_seg_vis(c), fe_kred(c), c->fe_hist.as<uint32_t>(), fe_totals(c), c->fe_rank_atomic));
    DvsSeg sv;
    HIPCHECK(hipMemcpyAsync(&sv, fe_seg_vis(c), sizeof sv, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    *n_sorted = sv.count;
    if (digit_bits) *digit_bits = sv.bits;
    HIPCHECK(hipMemcpyAsync(sorted_ids, c->ids[1].p, (size_t)sv.count * 4, hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipMemsetAsync(fe_kred(c), 0, sizeof slots, st));
    c->have_fwd = false;                                                    // the forward state of the context is gone
    return DVS_OK;
}

int dvs_export_sorted_keys(dvs_ctx* c, void* stream, uint64_t* out_keys) {
    if (!c || !out_keys) { g_last_error = "dvs_export_sorted_keys: null argument"; return DVS_ERR_INVALID; }
    if (!c->have_fwd) { g_last_error = "dvs_export_sorted_keys: no forward state"; return DVS_ERR_STATE; }
    if (c->st.num_rendered == DVS_T_UNKNOWN) { uint64_t t; int r = dvs_get_num_rendered(c, stream, &t); if (r != DVS_OK) return r; }
    if (!c->st.sorted_tile) {
        g_last_error = "dvs_export_sorted_keys: the last forward was asynchronous (dvs_set_async) and did not materialise the sorted tile ids; "
                       "export the keys of a synchronous forward";
        return DVS_ERR_STATE;
    }
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(dvs_launch_export_keys((hipStream_t)stream, c->st.num_rendered, c->st.sorted_tile, c->st.sorted_splat, c->st.depth, out_keys));
    return DVS_OK;
}

int dvs_shn_relayout(dvs_ctx* c, void* stream, int n, const float* src, float* dst, int to_tiled) {
    if (!c || n < 0 || (n > 0 && (!src || !dst))) { g_last_error = "dvs_shn_relayout: bad argument"; return DVS_ERR_INVALID; }
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(dvs_launch_shn_relayout((hipStream_t)stream, n, src, dst, to_tiled));
    return DVS_OK;
}

int dvs_sh_grad_combine(dvs_ctx* c, void* stream, int n, const float* pos, int sh_degree, int n_views, const float* campos,
                        const float* dcolor, float* g_sh0, float* g_shN, int accumulate, int shn_layout) {
    if (!c || n < 0 || n_views < 0 || sh_degree < 0 || sh_degree > 3 || (n > 0 && n_views > 0 && (!pos || !campos || !dcolor || !g_sh0 || !g_shN))) {
        g_last_error = "dvs_sh_grad_combine: bad argument"; return DVS_ERR_INVALID;
    }
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(dvs_launch_sh_grad_combine((hipStream_t)stream, n, pos, sh_degree, n_views, campos, dcolor, g_sh0, g_shN, accumulate, shn_layout));
    return DVS_OK;
}

int dvs_raster_backward_dcolor(dvs_ctx* c, void* stream, float* dcolor) {
    if (!c || !dcolor || ((uintptr_t)dcolor & 15u)) { g_last_error = "dvs_raster_backward_dcolor: bad argument (dcolor: 16-byte aligned device array)"; return DVS_ERR_INVALID; }
    if (!c->rows_pending) { g_last_error = "dvs_raster_backward_dcolor: no dvs_raster_backward_composite pending on this context"; return DVS_ERR_STATE; }
    HIPCHECK(hipSetDevice(c->device));
    const dvs_fwd_state& s = c->st;
    HIPCHECK(dvs_launch_dcolor_from_rows((hipStream_t)stream, (int64_t)c->n_views * s.n, s.radii, s.flags, c->g_rows.as<float>(), dcolor));
    return DVS_OK;
}

int dvs_set_async(dvs_ctx* c, int enable) {
    if (!c) { g_last_error = "dvs_set_async: null context"; return DVS_ERR_INVALID; }
    c->async_T = enable != 0;
    return DVS_OK;
}
int dvs_get_num_rendered(dvs_ctx* c, void* stream, uint64_t* T) {
    if (!c || !T) { g_last_error = "dvs_get_num_rendered: null argument"; return DVS_ERR_INVALID; }
    if (!c->have_fwd) { g_last_error = "dvs_get_num_rendered: no forward state"; return DVS_ERR_STATE; }
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    *T = c->total_host[0];
    c->st.num_rendered = *T;
    if (c->total_host[1] != c->overflow_seen) {
        c->overflow_seen = c->total_host[1];
        (void)ensure_instance_arenas(c, *T + *T / 2);
        g_last_error = "dvs_get_num_rendered: the last forward produced more tile instances than the instance arena held; its outputs are "
                       "invalid. The arena has been enlarged — repeat that view.";
        return DVS_ERR_CAPACITY;
    }
    return DVS_OK;
}

int dvs_get_sort_rank_mode(dvs_ctx* c) { return c ? c->fe_rank_atomic : -1; }
int dvs_set_export_sorted_tiles(dvs_ctx* c, int enable) {
    if (!c) { g_last_error = "dvs_set_export_sorted_tiles: null context"; return DVS_ERR_INVALID; }
    c->export_tiles = enable != 0;
    return DVS_OK;
}

int dvs_get_arena_info(dvs_ctx* c, uint64_t* cap, uint64_t* grows, uint64_t* last_T, uint64_t* overflows) {
    if (!c) { g_last_error = "dvs_get_arena_info: null context"; return DVS_ERR_INVALID; }
    if (cap) *cap = c->inst_cap;
    if (grows) *grows = c->inst_grow_events;
    if (last_T) *last_T = c->total_host[0];
    if (overflows) *overflows = c->total_host[1];
    return DVS_OK;
}

int dvs_debug_record_decisions(dvs_ctx* c, uint64_t* take_masks, uint64_t capacity_instances) {
    if (!c || (take_masks && (((uintptr_t)take_masks & 7u) || capacity_instances == 0))) { g_last_error = "dvs_debug_record_decisions: bad argument"; return DVS_ERR_INVALID; }
    c->rec_masks = take_masks; c->rec_cap = take_masks ? capacity_instances : 0;
    return DVS_OK;
}

int dvs_set_backward_variant(dvs_ctx* c, int variant) {
    if (!c || variant < DVS_BWD_BLOCKS || variant > DVS_BWD_TR) { g_last_error = "dvs_set_backward_variant: bad argument"; return DVS_ERR_INVALID; }
    if ((variant == DVS_BWD_REDUCE || variant == DVS_BWD_MM) && !dvs_launch_render_bwd) {
        g_last_error = "dvs_set_backward_variant: the 'reduce' and 'mm' kernels are retired — this library contains 'tr' (default) and 'blocks' only "
                       "(an experiment build, tools/xbuild.sh, brings them back)";
        return DVS_ERR_UNSUPPORTED;
    }
    c->bwd_variant = variant;
    return DVS_OK;
}
int dvs_set_live_lists(dvs_ctx* c, int enable) {
    if (!c) { g_last_error = "dvs_set_live_lists: null context"; return DVS_ERR_INVALID; }
    c->live_lists = enable != 0;
    return DVS_OK;
}
int dvs_set_forward_variant(dvs_ctx* c, int variant) {
    if (!c || (variant != DVS_FWD_BLOCKS && variant != DVS_FWD_QUADRANT)) { g_last_error = "dvs_set_forward_variant: bad argument"; return DVS_ERR_INVALID; }
    if (variant == DVS_FWD_BLOCKS && !dvs_launch_render_fwd_blocks) {
        g_last_error = "dvs_set_forward_variant: the per-block forward is retired — this library contains the quadrant forward only (tools/xbuild.sh brings it back)";
        return DVS_ERR_UNSUPPORTED;
    }
    c->fwd_variant = variant;
    return DVS_OK;
}

int dvs_keep_bwd_intermediates(dvs_ctx* c, int keep) { if (!c) return DVS_ERR_INVALID; c->keep_rows = keep != 0; return DVS_OK; }

int dvs_get_bwd_intermediates(dvs_ctx* c, const float** rows, int* row_floats) {
    if (!c) return DVS_ERR_INVALID;
    if (rows) *rows = c->g_rows.as<float>();
    if (row_floats) *row_floats = 12;
    return DVS_OK;
}

int dvs_enable_stage_timing(dvs_ctx* c, int enable) { if (!c) return DVS_ERR_INVALID; c->timing = enable != 0; return DVS_OK; }
int dvs_enable_kernel_probe(dvs_ctx* c, int enable) {
    if (!c) return DVS_ERR_INVALID;
    c->probe = enable != 0;
    c->probe_used = 0; c->probe_kind.clear();
    return DVS_OK;
}
int dvs_read_kernel_probe(dvs_ctx* c, float mean_ms[2], int count[2]) {
    if (!c || !mean_ms || !count) return DVS_ERR_INVALID;
    HIPCHECK(hipSetDevice(c->device));
    double sum[2] = {0, 0};
    count[0] = count[1] = 0;
    for (size_t k = 0; k < c->probe_kind.size() && 2 * k + 1 < c->probe_used; ++k) {
        float ms = 0.f;
        HIPCHECK(hipEventSynchronize(c->probe_ev[2 * k + 1]));
        HIPCHECK(hipEventElapsedTime(&ms, c->probe_ev[2 * k], c->probe_ev[2 * k + 1]));
        sum[c->probe_kind[k]] += ms; count[c->probe_kind[k]]++;
    }
    for (int i = 0; i < 2; ++i) mean_ms[i] = count[i] ? (float)(sum[i] / count[i]) : 0.f;
    c->probe_used = 0; c->probe_kind.clear();
    return DVS_OK;
}
int dvs_get_stage_timing(dvs_ctx* c, const char*** names, const float** ms) {
    if (!c) return 0;
    if (names) *names = c->out_names.data();
    if (ms) *ms = c->out_ms.data();
    return (int)c->out_names.size();
}

int dvs_memcpy_d2h(dvs_ctx* c, void* host_dst, const void* dev_src, size_t bytes) {
    if (!c) return DVS_ERR_INVALID;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return DVS_OK;
}
int dvs_memcpy_h2d(dvs_ctx* c, void* dev_dst, const void* host_src, size_t bytes) {
    if (!c) return DVS_ERR_INVALID;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return DVS_OK;
}
void* dvs_device_malloc(dvs_ctx* c, size_t bytes) {
    if (!c) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}
void dvs_device_free(dvs_ctx* c, void* p) { if (c && p) { (void)hipSetDevice(c->device); (void)hipFree(p); } }

}  // extern "C"
