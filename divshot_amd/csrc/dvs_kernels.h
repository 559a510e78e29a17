// dvs_kernels.h — host-callable launchers of the gfx950 kernels (internal to libdvsraster.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dvs_device.h"

// preprocess.hip
hipError_t dvs_launch_preprocess_fwd(hipStream_t st, int n, const float* pos, const float* sh0, const float* shN,
                                     const float* opacity, const float* scale, const float* rot, const DvsCams& cams, int n_views,
                                     int deg, int antialias, int tiles_x, int tiles_y, int* radii /*per-view outputs are [n_views][n]*/, float* splat2d,
                                     float* depth, uint32_t* flags,
                                     uint32_t* tiles_touched, uint32_t* depth_key, uint32_t* ids, int shn_tiled, uint32_t* rect /*[n,2]: 4 x u16*/,
                                     uint32_t* rect16 /*DVS_TILES_TIGHT: [n,4] = rectangle + 64-bit tile mask, written instead of rect; null = canonical*/,
                                     uint32_t* rect8 = nullptr /*DVS_FE_RECT_U8: [n] 4 x u8 (minx, miny, width, height) instead of rect*/,
                                     uint32_t* kred = nullptr /*segmented front end: the views' key ranges [n_views][64][16] (zeroed by the caller); ids may then be null*/,
                                     int first = 0, int count = -1 /*splat range of this launch: [first, first + count), count < 0 = up to n*/);
hipError_t dvs_launch_preprocess_bwd(hipStream_t st, int n, const float* pos, const float* shN, const float* opacity,
                                     const float* scale, const float* rot, const DvsCam& cam, int deg, int antialias,
                                     const int* radii, const uint32_t* flags, float* grad_rows /*[n,12], read then re-zeroed*/,
                                     float* g_pos, float* g_sh0, float* g_shN, float* g_opacity, float* g_scale,
                                     float* g_rot, float* out_absgrad2d /*nullable*/, float* out_mean2d /*nullable*/,
                                     float* out_dcolor /*nullable*/, int accumulate, int rezero_rows, int shn_tiled, int grad_mode);
// A9 for all views of a batch in one pass (DVS_SHN_TILED layout only): radii / flags / grad_rows are [n_views][n]; writes the
// geometry gradients (sums over the views) and out_dcolor [n_views][n][3]; the SH rows follow from dcolor (dvs_launch_sh_grad_combine)
hipError_t dvs_launch_preprocess_bwd_views(hipStream_t st, int n, int n_views, const float* pos, const float* shN, const float* opacity,
                                           const float* scale, const float* rot, const DvsCams& cams, int deg, int antialias,
                                           const int* radii, const uint32_t* flags, float* grad_rows, float* g_pos, float* g_opacity,
                                           float* g_scale, float* g_rot, float* out_absgrad2d, float* out_mean2d, float* out_dcolor,
                                           int accumulate, int rezero_rows, int grad_mode, int first = 0, int count = -1 /*splat range of this
                                           launch: [first, first + count), count < 0 = up to n*/,
                                           float* g_sh0 = nullptr, float* g_shN = nullptr /*both given: the kernel builds the SH rows itself
                                           (tiled layout) from the views' colour gradients and does NOT write out_dcolor — the one-GPU path*/);
// g_sh0 / g_shN may be nullptr in dvs_launch_preprocess_bwd (factorised exchange); this rebuilds them from dcolor[n_views,n,3].
hipError_t dvs_launch_sh_grad_combine(hipStream_t st, int n, const float* pos, int deg, int n_views, const float* campos_host,
                                      const float* dcolor, float* g_sh0, float* g_shN, int accumulate, int shn_tiled);
// reference rows [n][45] <-> tiled [ceil(n/64)][45][64]
hipError_t dvs_launch_dcolor_from_rows(hipStream_t st, int64_t total, const int* radii, const uint32_t* flags, const float* rows, float* dcolor);
hipError_t dvs_launch_shn_relayout(hipStream_t st, int n, const float* src, float* dst, int to_tiled);

// frontend.hip — the segmented (per-view) binning stage of round 5
// One segment of a segmented sort / scan (device memory, 32 B): a view's run of elements and its rows of the histogram table.
struct DvsSeg {
    uint32_t base;      // first element of the segment in the key / value arrays
    uint32_t count;     // its elements
    uint32_t pstart;    // its first row (partition) in the histogram table
    uint32_t sub;       // depth sort: the view's smallest key (digits are taken from key - sub)
    uint32_t bits;      // depth sort: digit width b = ceil(bits(max - min) / 3), 1..11
    uint32_t _r0, _r1, _r2;
};
// tile-rectangle record formats (A2 writes, A3 gathers, A4 streams)
enum { DVS_FE_RECT_U8 = 0 /*4 B: minx | miny << 8 | width << 16 | height << 24 (tiles_x, tiles_y <= 255)*/, DVS_FE_RECT_U16 = 1 /*8 B: 4 x u16*/,
       DVS_FE_RECT_TIGHT = 2 /*16 B: 4 x u16 + the 64-bit tile mask of DVS_TILES_TIGHT*/ };
#define DVS_FE_KRED_WORDS (DVS_MAX_VIEWS * 64 * 16)        /* key-range slots: [view][64][16 words] */
#define DVS_FE_MAXBINS 2048
#define DVS_FE_SUPER_STRIDE 32                                  /* u64 words between two super-sum counters (256 B: own memory channel) */
size_t dvs_fe_hist_words(uint64_t max_elems, int n_views, int max_bins);       // histogram table words for sorts of up to max_elems elements in n_views segments
uint32_t dvs_depth_sort_rows_per_view(int n, int V);
uint32_t dvs_fe_part_for(uint64_t grid_elems);                                 // keys per partition for a sort of about that many elements
void dvs_fe_block_counts(int n, uint32_t* nbv, uint32_t* nsb);                 // A3 workgroups per view, 64-bit super sums per view
hipError_t dvs_launch_seg_init(hipStream_t st, int n, int V, uint32_t rows_per_view, DvsSeg* seg_all);
// A5 (depth): keys0 [V][n] (culled = 0xFFFFFFFF) -> vals1[v * n + j] = index of the j-th nearest visible splat of view v, j < seg_vis[v].count
hipError_t dvs_launch_depth_sort(hipStream_t st, int n, int V, uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1,
                                 DvsSeg* seg_all, DvsSeg* seg_vis, const uint32_t* kred, uint32_t* hist, uint32_t* totals /*[V][DVS_FE_MAXBINS]*/,
                                 int rank_atomic = 0 /*scatter ranks by returning LDS adds: only after dvs_fe_probe_rank_atomic passed*/);
// stable LSD sort of every segment's pairs over the key bits [bit_lo, bit_lo + bits) (digits <= 9 bits); result in buffers *result_in
hipError_t dvs_launch_seg_sort(hipStream_t st, int V, uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, DvsSeg* seg, int bit_lo, int bits,
                               uint64_t grid_elems, uint32_t part, uint32_t nbtot, uint32_t* hist, uint32_t* totals, uint32_t key_add_per_view,
                               int* result_in, uint32_t* ranges_enc = nullptr /*A6 fused into the last pass: tile ranges as (~start, end), see k_seg_scatter*/,
                               int write_last_keys = 1, int first_keys16 = 0 /*keys0 holds 16-bit keys (A4's tile ids)*/, int rank_atomic = 0);
// runs the two in-wave rankings of k_seg_scatter side by side on synthetic digits; *bad_dev (zeroed) counts disagreements (frontend.hip)
hipError_t dvs_fe_probe_rank_atomic(hipStream_t st, uint32_t* bad_dev);
// stage 0 = A3 (tile counts in depth order, block / super sums) + the views' instance ranges (seg_tile, total_dev); stage 1 = A4
hipError_t dvs_launch_seg_binning(hipStream_t st, int n, int V, int rect_fmt, const DvsSeg* seg_vis, DvsSeg* seg_tile, const uint32_t* sorted_ids,
                                  const uint32_t* rect, uint32_t* rect_sorted, uint32_t* block_sums, unsigned long long* super, uint32_t* superexcl,
                                  uint32_t tile_part, unsigned long long* total_dev, unsigned long long capacity, int stage, int tiles_x,
                                  uint32_t* inst_tile, uint32_t* inst_splat, int key16 = 0 /*A4 writes the tile ids as 16-bit words*/);

// frontend.hip, continued
// A6: per-tile [start,end) from the sorted tile ids. T_dev (nullable): device-side count, T sizes the grid.
hipError_t dvs_launch_tile_ranges(hipStream_t st, uint64_t T, const uint32_t* sorted_tile, uint32_t* ranges, int tiles,
                                  const uint64_t* T_dev = nullptr, uint64_t T_expected = 0, bool clear = true /*false: the caller has zeroed `ranges`*/);
// canonical 64-bit keys of the sorted list (parity export)
hipError_t dvs_launch_export_keys(hipStream_t st, uint64_t T, const uint32_t* sorted_tile, const uint32_t* sorted_splat,
                                  const float* depth, uint64_t* out_keys);

// render.hip — one launch covers the tiles of all n_views views of a batch (view-major ranges / pixel arrays; bgs = [n_views][3])
hipError_t dvs_launch_render_fwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges /*ranges_out != null: the
                                 encoded form (~start, end) of the tile sort's last pass, (0, 0) = empty*/,
                                 const uint32_t* sorted_splat, const float* splat2d, const float* bgs, float* out_color, float* final_T,
                                 uint32_t* n_contrib, uint32_t* live_splat /*[T] out (or null): per tile, the entries whose alpha >= 1/255 ellipse
                                 reaches the tile, compacted in list order from ranges[tile].x*/, uint32_t* live_pos /*[T] out (or null): list
                                 position -> number of such entries before it in its tile*/,
                                 uint64_t* take_masks /*test hook (or null): [take_cap][4] zeroed by the caller — per list position and 8x8 quadrant, the pixels that took the entry*/,
                                 uint64_t take_cap, uint32_t* ranges_out = nullptr /*where k_render_fwd leaves (start, end) when `ranges` is encoded*/);
// EXPERIMENT BUILDS ONLY (-DDVS_EXPERIMENT): the retired A8 kernels "reduce" and "mm". Declared weak: the release library does not
// define it, dvs_set_backward_variant refuses those variants there.
hipError_t dvs_launch_render_bwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* splat2d, const float* bgs, const float* final_T, const uint32_t* n_contrib,
                                 const float* dL_dout, float* grad_rows /*[n,12] zero-initialised*/, int absgrad, int grad_mode,
                                 int variant /*DVS_BWD_*: which A8 kernel (the mm experiment renders one view)*/) __attribute__((weak));
// A8 kernel variants (dvs_set_backward_variant): same inputs, same 48-B row contract, results equal to fp32 roundoff
enum { DVS_BWD_BLOCKS = 0 /*per-4x4-block lists, four cursors per wave, 12-value group reduction per step (round 2): kept in the release library as
       the independent-summation-order cross-check of the parity tests*/, DVS_BWD_REDUCE = 1 /*per-quadrant masks, wave-wide reduction tree per visit
       (round 1; experiment builds only)*/, DVS_BWD_MM = 2 /*per-quadrant masks, sums contracted on the fp32 matrix pipe (experiment builds only)*/,
       DVS_BWD_TR = 3 /*per-4x4-block lists; (v5, w) pairs transposed through LDS and accumulated serially per (block, slot, pixel row):
       render_tr.hip (default since round 3)*/ };
// A7 kernel variants (dvs_set_forward_variant): bit-identical results
enum { DVS_FWD_BLOCKS = 0 /*per-4x4-block lists (experiment builds only)*/, DVS_FWD_QUADRANT = 1 /*per-quadrant masks walked by the scalar unit (default)*/ };

// render_tr.hip
hipError_t dvs_launch_render_bwd_tr(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                    const uint32_t* sorted_splat, const float* splat2d, const float* bgs /*[n_views][3]*/, const float* final_T,
                                    const uint32_t* n_contrib, const float* dL_dout, float* grad_rows, int absgrad, int grad_mode,
                                    const uint32_t* live_splat /*the forward's live lists (dvs_launch_render_fwd), or null: walk sorted_splat*/,
                                    const uint32_t* live_pos);

// render_blocks.hip
hipError_t dvs_launch_render_fwd_blocks(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                        const uint32_t* sorted_splat, const float* splat2d, const float bg[3], float* out_color,
                                        float* final_T, uint32_t* n_contrib) __attribute__((weak));      // experiment builds only
hipError_t dvs_launch_render_bwd_blocks(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                        const uint32_t* sorted_splat, const float* splat2d, const float* bgs /*[n_views][3]*/, const float* final_T,
                                        const uint32_t* n_contrib, const float* dL_dout, float* grad_rows, int absgrad, int grad_mode);
