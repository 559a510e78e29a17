/*
 * dvs_raster.h — the thin C-ABI between DIVSHOT-side host C++ and the MI355X (gfx950) HIP
 * rasterizer. Plain C: POD structs, raw device pointers, int status codes. No torch / STL types.
 *
 * What this boundary replaces in the reference (fenghuayumo/DIVSHOT):
 *   The reference's trainer plugin `gstrain` (closed source; see README.md:32,46 and
 *   diverse_utils/CMakeLists.txt:1-3, which add_subdirectory()s the absent `gsplatrast`,
 *   `gstrain_utils`, `gstrain`) calls its CUDA rasterizer `gsplatrast` from inside
 *   train_step() (call site application/diverseshot-cli/source/gs_train.cpp:156 and
 *   application/editor/source/editor.cpp:1620).  The include dir for that rasterizer is
 *   declared at CMakeLists.txt:103 / premake-dependencies.lua:38-42 but the directory is not
 *   in the tree, so the FFI below is specified from SURVEY.md §8(b) "B2": the two-op surface
 *   (forward returns image + saved state, backward returns the five per-splat gradient groups)
 *   of the rasterizer lineage credited at README.md:95.
 *
 *   Data layout (splat parameter block, "A0") follows the reference's trainer→viewer hand-off:
 *   getGaussian{Position,SH0,SHN,Opcaities,Scalings,Rotations}Cpu() (editor.cpp:1459-1473)
 *   → GaussianModel::update_from_cpu memcpy sizes (diverse/source/assets/gaussian_model.cpp:60-65):
 *   pos[N,3] sh0[N,3] shN[N,15,3] opacity[N] scale[N,3] rot[N,4] — 59 fp32 = 236 B per splat
 *   (editor.cpp:1578), raw (pre-activation) values; activations as gaussian_model.cpp:137-159.
 *
 * Alignment: every DEVICE array handed over (dvs_splats, dvs_splat_grads, images) must start on a 16-byte boundary: the kernels move
 * rot, the tiled shN chunks and the staged 3-float groups as 16-byte vectors. Misaligned pointers are rejected (DVS_ERR_INVALID).
 *
 * Threading: a dvs_ctx is single-caller (one in-flight view); use one ctx per concurrent view.
 * All work is enqueued on the hipStream_t passed in (as void*). Functions that return counts to
 * the host synchronise that stream once (documented per function).
 */
#ifndef DVS_RASTER_H
#define DVS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVS_TILE 16            /* tile edge in pixels (gaussian_common.hlsl:162-163 uses 16x16 groups) */
#define DVS_SH_REST 15         /* higher-order SH coefficients per channel (degree 3) */

/* status codes */
enum {
    DVS_OK = 0,
    DVS_ERR_INVALID = 1,       /* bad argument */
    DVS_ERR_HIP = 2,           /* a HIP runtime call failed; see dvs_last_error() */
    DVS_ERR_CAPACITY = 3,      /* n > max_splats or image > max_w x max_h given to dvs_create */
    DVS_ERR_STATE = 4,         /* backward called without a matching forward */
    DVS_ERR_UNSUPPORTED = 5    /* the library was built without what was asked for (a retired kernel variant) */
};

/* A0 — splat parameter block. DEVICE pointers, SoA, raw (pre-activation) fp32.
 *   pos[n*3]; sh0[n*3] (dc, rgb); shN[n*45] coefficient-major / channel-minor [j*3+c]
 *   (gaussian_model.cpp:163-167); opacity[n] logit; scale[n*3] log; rot[n*4] (w,x,y,z) unnormalised
 *   (quaternion order gsplat_vs.hlsl:189-199). */
typedef struct dvs_splats {
    const float* pos;
    const float* sh0;
    const float* shN;
    const float* opacity;
    const float* scale;
    const float* rot;
    int32_t n;
    int32_t _pad;
} dvs_splats;

/* A1 — camera block (HOST struct; copied into kernel arguments).
 *   view / proj are 4x4, stored so that element [c*4+r] multiplies input component c into
 *   output component r:  out.r = m[0*4+r]*x + m[1*4+r]*y + m[2*4+r]*z + m[3*4+r]
 *   (the convention of transformPoint4x4 at gsplat_vs.hlsl:63-72).
 *   view: world -> camera, +Z forward, Y down (COLMAP-like; editor.cpp:2028-2029 rotates trained
 *   models 180deg about X for display).  proj: world -> clip (full view-projection).
 *   focal_x = width / (2 tan_fovx) (gsplat_vs.hlsl:292-293). */
typedef struct dvs_camera {
    float view[16];
    float proj[16];
    float tan_fovx, tan_fovy;
    float focal_x, focal_y;
    float campos[3];
    int32_t width, height;
    float bg[3];
} dvs_camera;

/* layout of the shN array in dvs_splats AND of the shN rows in dvs_splat_grads */
enum {
    DVS_SHN_ROWS = 0,      /* [n][45]: the reference's hand-off layout (gaussian_model.cpp:163-167), element e of splat i at i*45+e */
    DVS_SHN_TILED = 1      /* [ceil(n/64)][12][64][4]: the 45 floats of a splat padded to 48 = twelve float4 chunks; element e of
                              splat i at (((i>>6)*12 + e/4)*64 + (i&63))*4 + e%4; ceil(n/64)*64*48 floats (pads are zero).
                              Chunk c of 64 consecutive splats = one contiguous 1-KiB run: a wave moves it with one 16-B-per-lane
                              instruction, no LDS round trip. Element-wise consumers (optimizer, all-reduce) are
                              layout-agnostic; dvs_shn_relayout converts. */
};
#define DVS_SHN_TILE_FLOATS 3072   /* floats per 64-splat tile of the DVS_SHN_TILED layout (12 * 64 * 4) */

typedef struct dvs_opts {
    int32_t sh_degree;     /* active SH degree 0..3 */
    int32_t antialias;     /* mip-splatting opacity compensation (main.cpp:63 --mipAntiliased; gsplat_vs.hlsl:296-301) */
    int32_t absgrad;       /* also accumulate |dL/dmean2D| (main.cpp:44 --absgrad) */
    int32_t accumulate;    /* backward: 0 = overwrite gradient rows, 1 = add into them (multi-view batches) */
    int32_t shn_layout;    /* DVS_SHN_ROWS (default 0) or DVS_SHN_TILED */
    int32_t grad_mode;     /* DVS_GRAD_TRUE (default 0) or DVS_GRAD_LINEAGE: which backward the two non-smooth points of the forward get */
    int32_t tile_bounds;   /* DVS_TILES_CANONICAL (default 0) or DVS_TILES_TIGHT: which (splat, tile) instances the binning stage emits */
    int32_t _reserved[1];
} dvs_opts;

/* dvs_opts.tile_bounds.
 *   DVS_TILES_CANONICAL  every tile of the ceil(3 sigma) rectangle of the splat (the rule of the rasterizer lineage the reference
 *                        credits, README.md:95): tiles_touched, the (tile | depth) keys, the per-tile lists, n_contrib and
 *                        num_rendered are the canonical ones every parity test pins. The default, and what the bench headline runs.
 *   DVS_TILES_TIGHT      OPT-IN: of that rectangle, only the tiles the splat's alpha >= 1/255 ellipse can reach (the cull the in-tree
 *                        viewer applies per pixel, gsplat_ps.hlsl:60-65; exact ellipse-vs-tile-rectangle test with a 1e-3 safety
 *                        margin on alpha, evaluated in preprocess from IEEE-exact operations so that the CPU oracle reproduces the
 *                        instance list bit for bit). Rectangles of more than 64 tiles are kept whole. Images, final_T and every
 *                        gradient are those of the canonical lists (an instance that is dropped never contributes to a pixel);
 *                        tiles_touched, num_rendered, the exported lists and n_contrib (a list position) refer to the shorter
 *                        lists. At BASELINE config C3, 28 % of the canonical instances go. */
enum { DVS_TILES_CANONICAL = 0, DVS_TILES_TIGHT = 1 };

/* dvs_opts.grad_mode. The forward is identical in both modes; they differ only where the forward is not smooth:
 *   DVS_GRAD_TRUE    the exact derivative of the forward: a pixel whose alpha hit the 0.99 cap passes no gradient to the
 *                    conic / mean / opacity of that splat (d min(0.99, x)/dx = 0 there), and on the clamped branch of the EWA
 *                    Jacobian (|t.x/t.z| > 1.3 tan_fov) the clamped coordinate t.x = +-lim * t.z is differentiated through t.z.
 *                    This is the mode the fp64 finite-difference checks validate.
 *   DVS_GRAD_LINEAGE the backward of the rasterizer lineage the reference credits (README.md:95): the gradient flows through
 *                    the alpha cap as if alpha were opacity*G, and on the clamped Jacobian branch the clamped coordinate is held
 *                    constant (its gradient to t.x is dropped, none is added to t.z). libgstrain.so uses this mode. */
enum { DVS_GRAD_TRUE = 0, DVS_GRAD_LINEAGE = 1 };

/* Saved forward state. DEVICE pointers into ctx-owned arenas; valid until the next
 * dvs_raster_forward on the same ctx. Exposed so the parity tests can diff every stage. */
enum { DVS_S2D_X = 0, DVS_S2D_Y = 1, DVS_S2D_CONIC = 2, DVS_S2D_OPACITY = 5, DVS_S2D_RGB = 6, DVS_S2D_DEPTH = 9, DVS_S2D_RADIUS = 10,
       DVS_S2D_CULL = 11, DVS_S2D_FLOATS = 16 };
typedef struct dvs_fwd_state {
    /* per splat (n) */
    const int32_t*  radii;          /* 0 = culled */
    const float*    splat2d;        /* [n,16] the projected splat as ONE 64-byte record (one cache line per gather in A4/A7/A8):
                                       DVS_S2D_X, _Y pixel coords of the mean (pixel i centre = i) | _CONIC a,b,c | _OPACITY final opacity |
                                       _RGB clamped colour r,g,b | _DEPTH view-space z | _RADIUS (int32 bits) | _CULL five constants of the composite
                                       kernels' conservative ellipse-vs-rectangle tests (internal: 2 ln(255 o) inflated, det/c, det/a, -b/c, -b/a) */
    const float*    depth;          /* [n] view-space z (also the sort key) */
    const uint32_t* flags;          /* [n] bit0..2 = SH clamp (colour channel <0), bit3 = fx clamped, bit4 = fy clamped */
    const uint32_t* tiles_touched;  /* [n] */
    /* per instance (num_rendered), sorted by (tile, depth, splat id) */
    const uint32_t* sorted_tile;    /* [T] tile id of each sorted instance (a batch: view * tiles + tile). NULL after an ASYNCHRONOUS forward
                                       (dvs_set_async): the ids are then not written at all — the list is grouped by tile, `ranges` says where —
                                       unless dvs_set_export_sorted_tiles(ctx, 1) asked for them. (ABI note, round 5: before that round the pointer
                                       was always valid; a consumer of this struct that runs asynchronous forwards must check it or set the switch.) */
    const uint32_t* sorted_splat;   /* [T] splat id ("value") of each sorted instance */
    /* per tile */
    const uint32_t* ranges;         /* [tiles,2] [start,end) into the sorted lists */
    /* per pixel */
    const float*    final_T;        /* [H,W] */
    const uint32_t* n_contrib;      /* [H,W] index (1-based, within tile list) of last contributor */
    uint64_t num_rendered;          /* T */
    int32_t n, width, height, tiles_x, tiles_y, _pad;
} dvs_fwd_state;

/* Per-splat gradient rows. DEVICE pointers, caller-owned, same shapes as dvs_splats
 * (59 floats per splat; rows of culled splats are written as zero unless opts.accumulate). */
typedef struct dvs_splat_grads {
    float* pos;
    float* sh0;
    float* shN;
    float* opacity;
    float* scale;
    float* rot;
    float* absgrad2d;   /* [n,2] optional (may be NULL): sum over pixels of |dL/dmean2D| per axis, pixel units */
    float* mean2d;      /* [n,2] optional (may be NULL): dL/dmean2D, pixel units (densification statistic) */
    float* dcolor;      /* [n,3] optional (may be NULL): dL/d(view-dependent colour), zero where the colour was clamped or the
                           splat culled. When given, sh0 and shN may be NULL: the SH rows are then NOT written and are rebuilt
                           later by dvs_sh_grad_combine (factorised data-parallel exchange). Always overwritten. */
} dvs_splat_grads;

typedef struct dvs_ctx dvs_ctx;

/* Create a rasterizer context on HIP device `device`. Arenas are sized for max_splats and
 * max_w x max_h; the instance arena grows on demand. Returns NULL on failure (dvs_last_error). */
dvs_ctx* dvs_create(int device, size_t max_splats, int max_w, int max_h);
void     dvs_destroy(dvs_ctx* ctx);

/* Forward: A2 preprocess -> A3 scan -> A4 duplicate -> A5 radix sort -> A6 ranges -> A7 composite.
 *   out_rgb: DEVICE [3,H,W] planar fp32.  saved: filled with pointers into ctx arenas (may be NULL).
 *   num_rendered: host pointer, may be NULL.
 * Synchronises `stream` once internally (reads the instance count T to size the sort) unless dvs_set_async(ctx, 1). */
int dvs_raster_forward(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cam,
                       const dvs_opts* opts, float* out_rgb, dvs_fwd_state* saved, uint64_t* num_rendered);

/* Backward: A8 composite backward -> A9 preprocess backward, using the state of the last forward on ctx.
 *   dL_drgb: DEVICE [3,H,W] planar fp32, 16-byte aligned like every device array of this interface (DVS_ERR_INVALID otherwise: the
 *   composite backward reads it as 16-byte words).  out: gradient rows (see dvs_splat_grads). Asynchronous. */
int dvs_raster_backward(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cam,
                        const dvs_opts* opts, const float* dL_drgb, const dvs_splat_grads* out);

/* The two halves of dvs_raster_backward as separate calls, for callers that batch several
 * views per optimizer step on several contexts/streams:
 *   _composite : A8 only — needs just the forward state and dL_drgb; writes the context's own 48-B intermediate rows, so the
 *                composite backward of view v+1 may run concurrently with anything of view v;
 *   _project   : A9 — turns those rows into parameter gradients. With opts->accumulate it adds into `out` non-atomically, so
 *                the _project calls that share gradient arrays must be ordered (an event between streams); nothing else must.
 * dvs_raster_backward == _composite followed by _project on the same stream (bit-identical results).
 * After dvs_raster_forward_views both calls cover all views of that forward: `cam` then points to its n_views cameras, dL_drgb is
 * [n_views,3,H,W] and out follows dvs_raster_backward_views. */
int dvs_raster_backward_composite(dvs_ctx* ctx, void* stream, const dvs_camera* cam, const dvs_opts* opts, const float* dL_drgb);
int dvs_raster_backward_project(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cam,
                                const dvs_opts* opts, const dvs_splat_grads* out);

/* dvs_raster_backward_project in CHUNKS of splats, for a data-parallel step that overlaps its gradient exchange with A9 (SURVEY.md
 * §8(e): "launch the reduce for splat-chunk k as soon as A9 has finished chunk k"): each call turns the rows of the splats
 * [first, first + count) into gradients; the chunks must cover [0, n) in ascending order, every `first` a multiple of 256; after the
 * last one the pending rows are consumed as after dvs_raster_backward_project. Factorised form only (DVS_SHN_TILED, out->dcolor given,
 * out->sh0 / out->shN NULL: the SH rows follow from dvs_sh_grad_combine). Results are bit-identical to the unchunked call. */
int dvs_raster_backward_project_chunk(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cam,
                                      const dvs_opts* opts, const dvs_splat_grads* out, int64_t first, int64_t count);

/* Between dvs_raster_backward_composite and dvs_raster_backward_project: dcolor [n_views,n,3] (DEVICE, 16-byte aligned) receives
 * the per-view colour gradients — the values dvs_raster_backward_project later writes to out->dcolor, bit for bit — so that a
 * data-parallel trainer can start their all-gather while A9 runs. Does not consume the pending rows. */
int dvs_raster_backward_dcolor(dvs_ctx* ctx, void* stream, float* dcolor);

/* Rebuild SH gradient rows from per-view colour gradients: for every view v and splat i with dir = normalize(pos_i - campos_v):
 *   g_sh0[i] (+)= SH_C0 * dcolor[v,i],   g_shN[i,k] (+)= basis_k(dir) * dcolor[v,i].
 * This is exactly what dvs_raster_backward writes into sh0/shN for one view, summed over views; it lets data-parallel ranks
 * all-gather 12 B/splat/view (dcolor) instead of all-reducing the 192-B SH rows. pos, dcolor [n_views,n,3], g_* are DEVICE
 * pointers; campos [n_views,3] is a HOST array (camera centres, dvs_camera.campos). accumulate = 0 overwrites the rows. */
int dvs_sh_grad_combine(dvs_ctx* ctx, void* stream, int n, const float* pos, int sh_degree, int n_views, const float* campos,
                        const float* dcolor, float* g_sh0, float* g_shN, int accumulate, int shn_layout);
/* Convert an shN array (DEVICE, src != dst) between DVS_SHN_ROWS [n*45] and DVS_SHN_TILED [ceil(n/64)*64*48]. */
int dvs_shn_relayout(dvs_ctx* ctx, void* stream, int n, const float* src, float* dst, int to_tiled);

/* Multi-view batches (BASELINE config C4: several cameras per training iteration). A context created with dvs_create_views renders
 * up to max_views (<= 16) views of ONE parameter block per call: the parameters are read once for all views (A2, A9), one depth sort,
 * one scan, one duplication, one (view, tile) sort and one composite launch cover the whole batch, and the backward reads the
 * parameters and writes the gradient rows ONCE (the sum over the views) instead of a read-modify-write per view.
 *   cams      HOST array [n_views], all views with the same image size
 *   out_rgb   DEVICE [n_views,3,H,W];   dL_drgb DEVICE [n_views,3,H,W]
 *   out       gradient rows = the sum over the views (opts->accumulate adds to what is there); out->dcolor, when given, is
 *             [n_views,n,3] (one colour gradient per view: what the factorised data-parallel exchange all-gathers); absgrad2d / mean2d
 *             are the sums over the views of the per-view statistics
 * Results equal the views run one by one with opts.accumulate: image / saved state bit for bit, geometry gradients bit for bit, SH rows
 * to fp32 roundoff. dvs_get_view_state returns view v's slice of the saved state (per-splat, per-pixel and per-tile arrays; the sorted
 * instance lists stay batch-wide: ranges index into them and their values are v * n + splat). The single-view calls are the
 * n_views = 1 case of the same code path. */
dvs_ctx* dvs_create_views(int device, size_t max_splats, int max_w, int max_h, int max_views);
int dvs_raster_forward_views(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cams, int n_views,
                             const dvs_opts* opts, float* out_rgb);
int dvs_raster_backward_views(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cams, int n_views,
                              const dvs_opts* opts, const float* dL_drgb, const dvs_splat_grads* out);

/* A2 (project / preprocess) of the NEXT dvs_raster_forward_views on this context, ahead of it and by splat range — for a data-parallel
 * trainer that updates its parameters chunk by chunk as the chunks' gradient all-reduces land (SURVEY.md 8(e) "Overlap"): as soon as the
 * optimizer has stepped the splats [first, first + count) their projection for the next iteration's cameras can run, under the all-reduces
 * of the chunks behind them. Chunks must cover [0, n) in ascending order, each starting at a multiple of 256, with the same parameters
 * pointers, cameras and options (DVS_SHN_TILED layout); the first chunk (first = 0) invalidates the previous forward's state — its backward
 * must have been queued before. A following dvs_raster_forward_views with exactly these arguments then skips its own A2 (bit-identical
 * outputs: A2 is per splat); with anything else, or after an incomplete sequence, it ignores the preparation and projects everything
 * itself. Asynchronous, on `stream`. */
int dvs_raster_forward_views_prepare(dvs_ctx* ctx, void* stream, const dvs_splats* params, const dvs_camera* cams, int n_views,
                                     const dvs_opts* opts, int64_t first, int64_t count);
/* Forget a preparation (the caller changed the parameters' CONTENTS behind the same pointers — a reset, a refinement): the next forward
 * projects everything itself. */
int dvs_raster_forward_cancel_prepared(dvs_ctx* ctx);
int dvs_get_view_state(dvs_ctx* ctx, int view, dvs_fwd_state* state);

/* Asynchronous forward. By default dvs_raster_forward synchronises `stream` once (it reads the instance count T to size the sort).
 * With dvs_set_async(ctx, 1) it never synchronises: the instance arena is over-allocated, T stays on the device and every kernel over
 * instances reads it there; *num_rendered and saved->num_rendered are DVS_T_UNKNOWN (dvs_get_num_rendered synchronises on demand).
 * The host learns the T of EARLIER forwards on the context from an asynchronously refreshed pinned copy and enlarges the arena ahead of
 * need. If a forward nevertheless produces more instances than the arena holds, nothing is written out of bounds, that view's outputs
 * are invalid, and the next dvs_raster_forward / dvs_get_num_rendered on the context returns DVS_ERR_CAPACITY once (after enlarging
 * the arena) — a hard error, never a silent truncation (SURVEY.md §7 "tile-list overflow"). DVS_ASYNC=1 sets the default. */
#define DVS_T_UNKNOWN (~0ull)
int dvs_set_async(dvs_ctx* ctx, int enable);
int dvs_get_num_rendered(dvs_ctx* ctx, void* stream, uint64_t* num_rendered);
/* The instance arena of the context (host bookkeeping, no synchronisation): how many tile instances it holds now, how often it has been
 * enlarged since dvs_create, the T of the last forward the host knows (synchronous forward: that forward's; asynchronous: an earlier
 * one's) and how many forwards overflowed it (each of those was reported once as DVS_ERR_CAPACITY). Any pointer may be NULL. */
int dvs_get_arena_info(dvs_ctx* ctx, uint64_t* instance_capacity, uint64_t* grow_events, uint64_t* last_num_rendered, uint64_t* overflows);
/* How the radix sorts of the context rank keys inside a wavefront: 1 = returning LDS adds (selected at dvs_create when the device serves
 * the lanes that add to one LDS address in lane order — probed on the device, csrc/frontend.hip), 0 = ballot multisplit (the fallback,
 * and what DVS_FE_RANK=ballot forces). Both give the same stable order; the parity suite runs both. -1 for a NULL context. */
int dvs_get_sort_rank_mode(dvs_ctx* ctx);
/* Asynchronous forwards (dvs_set_async) normally leave dvs_fwd_state.sorted_tile NULL: the tile sort's last pass builds the tile ranges
 * itself and skips the 4 B per instance nobody reads. enable = 1 makes them write the sorted tile ids again (for a caller that exports the
 * lists, e.g. dvs_export_sorted_keys, without giving up the asynchronous mode); costs ~10 us per 8-view step at C3. Default 0. */
int dvs_set_export_sorted_tiles(dvs_ctx* ctx, int enable);

/* TEST HOOK of the parity suite. While take_masks is non-NULL, every single-view synchronous forward on the context (default A7 kernel)
 * also records ITS OWN threshold decisions: take_masks[4 * j + q] (device memory, 8-byte aligned, capacity_instances * 4 words, zeroed by
 * the forward) = 64-bit mask of the pixels of 8x8 quadrant q of the tile that took list entry j (alpha >= 1/255, power <= 0, not yet
 * saturated). The fp64 oracle replays exactly these decisions (oracle dvso_set_replay), which removes the only legitimate source of
 * disagreement between an fp32 and an fp64 rasterizer — a threshold passed on one side and missed on the other — so that EVERY splat
 * is held to the 1e-4 bar (tests/test_gpu_parity.py). Same arithmetic and images as without it; NULL switches it off. */
int dvs_debug_record_decisions(dvs_ctx* ctx, uint64_t* take_masks, uint64_t capacity_instances);
/* TEST HOOK. Runs the forward's depth sort (A5, low 32 key bits: three passes whose digit width follows the key range — frontend.hip) on
 * caller-supplied DEVICE keys of ONE view: keys[n] as the projection would leave them (float bits of the depth; 0xFFFFFFFF = culled, those
 * leave in the first pass). sorted_ids[n_sorted] receives the indices of the surviving keys in stable ascending key order, *digit_bits the
 * digit width the range selected (9 up to a ratio of 2^12.5 between the keys' extremes, 11 for the full 31 bits — a range the projection
 * itself cannot produce, which is why the widest digit is reached through this hook only). Synchronises; invalidates the forward state. */
int dvs_debug_sort_depth_keys(dvs_ctx* ctx, void* stream, const uint32_t* keys, uint64_t n, uint32_t* sorted_ids, uint32_t* n_sorted, uint32_t* digit_bits);

/* The library ships ONE composite forward (A7, "quadrant") and ONE composite backward (A8, "tr"), plus the round-2 backward "blocks" as
 * the independent-summation-order cross-check of the parity tests. The other measured alternatives of DESIGN.md §5 — backward "reduce"
 * (round 1) and "mm" (matrix-pipe experiment), forward "blocks" — are retired: their source stays in csrc/ behind -DDVS_EXPERIMENT
 * (tools/xbuild.sh builds such a library for A/B runs) and the release library answers DVS_ERR_UNSUPPORTED when they are asked for.
 * Backward (A8), results equal to fp32 roundoff:
 *   3 "tr"      (default since round 3) per-4x4-pixel-block splat lists; a list step ends when the pair's two per-pixel scalars
 *               (G dL/dalpha, alpha T) are known: they cross an LDS transposition buffer, and every four steps each lane sums one
 *               pixel row of one (block, step) pair serially in registers (moments about the row origin moved to the mean
 *               algebraically) — no cross-lane reduction per step; per-wave LDS tables, one global atomic per (tile, splat, value)
 *   0 "blocks"  (round 2) the same lists; a 12-value reduction over the block's 16 lanes per step, group totals into the per-wave table
 *   1 "reduce"  (retired) per-8x8-quadrant cull masks, a 12-value wave-wide reduction tree and one atomic row update per (wave, splat) visit
 *   2 "mm"      (retired) per-quadrant masks, the per-splat sums contracted on the fp32 matrix pipe; one view per launch only
 * Forward (A7), bit-identical results:  1 "quadrant" (default) / 0 "blocks" (retired).
 * The environment variables DVS_BWD_VARIANT / DVS_FWD_VARIANT (digits) set the defaults of new contexts (a retired variant is ignored). */
int dvs_set_backward_variant(dvs_ctx* ctx, int variant);
int dvs_set_forward_variant(dvs_ctx* ctx, int variant);
/* Live lists (default on; DVS_LIVE_LISTS=0 sets the default of new contexts off): the "quadrant" forward writes, per tile, the
 * entries of its sorted list whose alpha >= 1/255 ellipse reaches the tile at all — 72 % of them at BASELINE config C3, the 3-sigma
 * rectangles of A4 being wider — compacted in list order, and the map list position -> live position; the "tr" backward walks those
 * instead of the full lists (28 % fewer entries staged, tabulated and published; the contributing entries are all on them, so the
 * gradients are the same sums). Internal arrays in the sort's spare buffers; the exported dvs_fwd_state lists stay the canonical ones.
 * Takes effect at the next forward. */
int dvs_set_live_lists(dvs_ctx* ctx, int enable);

/* Stage-level entry points (used by the parity tests and the profiler harness). */
/* radix sort of (u32 key, u32 value) pairs over key bits [bit_lo, bit_hi), stable, LSD, 8-bit digits.
 * keys/vals are DEVICE arrays of length n, sorted in place (ctx scratch is used as the ping-pong buffer). */
int dvs_sort_pairs_u32(dvs_ctx* ctx, void* stream, uint32_t* keys, uint32_t* vals, uint64_t n, int bit_lo, int bit_hi);
/* Reconstruct the canonical 64-bit keys ((tile<<32)|depth_bits) of the sorted instance list into
 * DEVICE out_keys[T] (parity tests compare them bit-exactly with the oracle's stable_sort). */
int dvs_export_sorted_keys(dvs_ctx* ctx, void* stream, uint64_t* out_keys);

/* Intermediate sums of the last backward (A8 output; DEVICE, ctx-owned): one row of *row_floats (=12) fp32 per splat:
 *   S_x, S_y | S_xx, S_xy, S_yy | dL/dopacity | dL/drgb r,g,b | sum|dL/dmean2D| x,y | pad
 * where S_* are the moments of s = dL/dG * G about the splat's 2D mean (d = mean - pixel). With the conic (a, b, c):
 *   dL/dmean2D = -(a S_x + b S_y, c S_y + b S_x),   dL/dconic (a, b, c) = (-S_xx / 2, -S_xy, -S_yy / 2)
 * (A9 applies this once per splat). For stage-level parity. */
int dvs_get_bwd_intermediates(dvs_ctx* ctx, const float** rows, int* row_floats);
/* By default the backward re-zeroes each row as A9 consumes it (no separate memset pass per view); keep = 1 leaves the rows
 * in place so dvs_get_bwd_intermediates() can be read after dvs_raster_backward (parity tests). */
int dvs_keep_bwd_intermediates(dvs_ctx* ctx, int keep);

/* Per-stage GPU time (ms) of the last forward/backward, measured with hipEvents on the caller's stream
 * when profiling is enabled. names/ms arrays are ctx-owned; returns the number of stages. */
int dvs_enable_stage_timing(dvs_ctx* ctx, int enable);
int dvs_get_stage_timing(dvs_ctx* ctx, const char*** names, const float** ms);
/* Kernel probe: hipEvent pairs around the two composite kernels (A7, A8) only, recorded on the caller's stream and never
 * synchronised inside forward/backward, so the kernels are timed under the concurrency of the caller's real (multi-stream) step.
 * dvs_read_kernel_probe synchronises on the recorded events and returns the mean duration (ms) and launch count of
 * [0] k_render_fwd, [1] k_render_bwd since the probe was enabled / last read. */
int dvs_enable_kernel_probe(dvs_ctx* ctx, int enable);
int dvs_read_kernel_probe(dvs_ctx* ctx, float mean_ms[2], int count[2]);

/* Synchronous copies between host and device on the ctx's device (test plumbing; no torch needed). */
int dvs_memcpy_d2h(dvs_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int dvs_memcpy_h2d(dvs_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
void* dvs_device_malloc(dvs_ctx* ctx, size_t bytes);
void  dvs_device_free(dvs_ctx* ctx, void* p);

const char* dvs_last_error(void);
const char* dvs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DVS_RASTER_H */
