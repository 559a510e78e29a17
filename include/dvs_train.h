/*
 * dvs_train.h — C-ABI of the two streaming ops that sit immediately before / after the rasterizer in
 * train_step(): image loss + its gradient, and the fused Adam update of
 * the 59-float splat rows. They exist so that libgstrain's train_step() is a real training iteration;
 * the reference's own implementations live in the closed `gstrain` plugin (README.md:46). Flags that
 * select them in the reference: --ssim (application/diverseshot-cli/source/main.cpp:24-25), learning
 * rates (main.cpp:31, gs_train.cpp:52-57).
 */
#ifndef DVS_TRAIN_H
#define DVS_TRAIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Mean-L1 image loss over count floats: dL[i] = sign(rgb[i]-target[i]) / count; *loss_accum += sum|rgb-target| / count.
 * rgb, target, dL, loss_accum are DEVICE pointers; loss_accum (1 float) must be zeroed by the caller. Asynchronous. */
int dvs_l1_loss_grad(void* stream, const float* rgb, const float* target, size_t count, float* dL, float* loss_accum);
/* Weighted form for the mixed loss (1-w) L1 + w (1-SSIM): dL[i] = weight * sign(..)/count; *loss_accum += weight * mean|..|. */
int dvs_l1_loss_grad_w(void* stream, const float* rgb, const float* target, size_t count, float weight, float* dL, float* loss_accum);
/* Mean-squared-error variant (the upstream gradient bench.py uses): dL = (rgb-target) * (2/count)... scaled by `scale`. */
int dvs_l2_loss_grad(void* stream, const float* rgb, const float* target, size_t count, float scale, float* dL, float* loss_accum);

/* D-SSIM term of the photometric loss (reference flag --ssim, default weight 0.2: main.cpp:24-25; loss = (1-w) L1 + w (1 - SSIM)).
 * SSIM with the standard 11x11 Gaussian window (sigma 1.5), zero padding, C1 = 0.01^2, C2 = 0.03^2, per channel, mean over
 * all 3*H*W values. Two fused passes over LDS-staged 16x16 tiles with a 5-pixel halo:
 *   dvs_ssim_forward : img, target [3,H,W] -> per-pixel partial derivative maps (3 x [3,H,W], caller-provided scratch) and
 *                      ssim_sum[0..DVS_SSIM_SLOTS) += partial sums of the SSIM map (add the slots, divide by 3*H*W for the mean;
 *                      several slots because tens of thousands of same-address atomics would serialise);
 *   dvs_ssim_backward: dL_dimg[3,H,W] (+)= scale * d(mean SSIM)/d(img)   (scale = -w to minimise 1 - SSIM; accumulate = add).
 * All pointers DEVICE; asynchronous on `stream`. */
#define DVS_SSIM_SLOTS 4096          /* (64 slots made the 24 k per-workgroup atomics of a 1080p image the bottleneck of both loss kernels: 0.17 vs 0.06 ms) */
int dvs_ssim_forward(void* stream, const float* img, const float* target, int width, int height, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, float* ssim_sum);
int dvs_ssim_backward(void* stream, const float* img, const float* target, int width, int height, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, float scale, float* dL_dimg, int accumulate);
/* The whole photometric gradient of L = (1-w) mean|x-y| + w (1 - mean SSIM) in one pass after dvs_ssim_forward:
 *   dL_dimg = (1-w)/(3HW) sign(x-y) - w/(3HW) dSSIM/dx   (overwritten),   l1_sum[0..DVS_SSIM_SLOTS) += (1-w)/(3HW) sum|x-y| (nullable).
 * Equivalent to dvs_l1_loss_grad_w(weight 1-w) followed by dvs_ssim_backward(scale -w, accumulate). */
int dvs_loss_l1_ssim_backward(void* stream, const float* img, const float* target, int width, int height, const float* dm_dmu1,
                              const float* dm_dsigma1_sq, const float* dm_dsigma12, float ssim_weight, float* dL_dimg, float* l1_sum);

/* Fused Adam over one parameter array (count floats): m, v are the moment arrays (same size, DEVICE).
 * step is 1-based. Asynchronous. */
int dvs_adam_step(void* stream, float* param, const float* grad, float* m, float* v, size_t count, float lr, float beta1,
                  float beta2, float eps, int step);

/* All parameter groups of one optimizer step in ONE launch (the trainer has six: pos, sh0, shN, opacity, scale, rot).
 *   width        floats per splat of the group (3, 45, 1, 4 ...); only used to find the splat of an element when `visible` is given.
 *   layout       DVS_SHN_ROWS: element e belongs to splat e / width.  DVS_SHN_TILED (width 45 only): the [ceil(n/64)][12][64][4] layout.
 *   active_chunks  DVS_SHN_TILED only: update just the first active_chunks (1..12) float4 chunks of every splat, 0 = all. While the
 *                progressive SH degree is below 3 the higher coefficients have g = m = v = 0 and Adam is the identity on them, so
 *                skipping them is exact (chunks needed for degree d: ceil(3*((d+1)^2-1)/4)).
 *   visible      nullable int32[n_splats] (dvs_fwd_state.radii of this view): when given, only splats with visible[i] > 0 are updated and
 *                the moments of the others do not decay — the reference's `visibleAdam` option (gs_train.cpp:87; the "sparse Adam"
 *                of Taming-3DGS). NULL = dense Adam, identical to dvs_adam_step per group.
 *   alignment    param / grad / m / v of every group must be 16-byte aligned (moved as float4); DVS_ERR_INVALID otherwise. */
typedef struct dvs_adam_group {
    float* param;
    const float* grad;
    float* m;
    float* v;
    uint64_t count;       /* floats in the arrays (tiled: padded to whole 64-splat tiles) */
    float lr;
    int32_t width;
    int32_t layout;
    int32_t active_chunks;
} dvs_adam_group;
#define DVS_ADAM_MAX_GROUPS 8
int dvs_adam_step_groups(void* stream, const dvs_adam_group* groups, int n_groups, float beta1, float beta2, float eps, int step,
                         const int32_t* visible, int32_t n_splats);

/* ---- adaptive density control: clone / split / prune, the "ADC" strategy of --densifyStrategy -------------
 * (flags application/diverseshot-cli/source/main.cpp:20,29,46-65: growGrad2d 2e-4, warmupLength 500, refineEvery 100, resetAlphaEvery
 * 3000, refineStopIter 15000, minOpacity 0.005, capMax gs_train.cpp:89). The reference's own densifier is in the closed plugin; this
 * is the public ADC rule it names, operating in place on HBM-resident arrays with no host round trip except the new count.
 *
 * Per refinement interval:
 *   dvs_densify_accumulate after every backward: for visible splats  grad_accum += |g| (g = abs-grad of the 2D mean in NDC units:
 *                          pixel-unit gradient x 0.5*(W,H)),  denom += 1,  max_radii = max(max_radii, radius).
 *   dvs_densify_plan       action per splat from avg = grad_accum/denom:   PRUNE  sigmoid(opacity) < min_opacity, or world/screen size
 *                          above the limits;  SPLIT  avg >= grad_threshold and max exp(scale) > scale_threshold (replaced by 2
 *                          samples, scale / 1.6);  CLONE  avg >= grad_threshold otherwise (kept + 1 copy);  KEEP.
 *                          Writes action[n], the exclusive scan of the output counts offsets[n] and the new count (device + pinned host
 *                          copy is the caller's business). Growth is cut off deterministically (by splat index) at cap_max.
 *   dvs_densify_apply      scatters ONE attribute set old -> new at the planned offsets. mode 0 = parameters (split samples drawn
 *                          from the splat's own Gaussian with a counter-based hash RNG), mode 1 = optimizer moments (kept rows copied,
 *                          rows of new splats zero). shN arrays may be in either layout (shn_layout).
 */
enum { DVS_DENSIFY_KEEP = 0, DVS_DENSIFY_CLONE = 1, DVS_DENSIFY_SPLIT = 2, DVS_DENSIFY_PRUNE = 3 };
typedef struct dvs_densify_params {
    float grad_threshold;      /* growGrad2d */
    float scale_threshold;     /* world units: percent_dense * scene extent */
    float min_opacity;         /* prune below (activated opacity) */
    float max_world_scale;     /* prune if max exp(scale) exceeds it; 0 = off */
    int32_t max_screen_radius; /* prune if max_radii exceeds it; 0 = off */
    int32_t cap_max;           /* hard cap on the new count */
    uint32_t seed;             /* RNG stream for split samples (use the step number) */
    int32_t shn_layout;        /* DVS_SHN_ROWS / DVS_SHN_TILED for the shN arrays passed to dvs_densify_apply */
    int32_t revised_opacity;   /* config `revisedOpacity`: both results of a clone / split take opacity 1 - sqrt(1 - o), so that
                                  the pair composites to the opacity of the splat it replaces ("Revising Densification in GS") */
} dvs_densify_params;

int dvs_densify_accumulate(void* stream, int n, const int32_t* radii, const float* absgrad2d, int width, int height,
                           float* grad_accum, float* denom, int32_t* max_radii);
/* The same rule for every view of a multi-view pass (dvs_raster_forward_views), exactly as if the views had been accumulated one by
 * one: radii [n_views][n] (dvs_fwd_state.radii of the batch), rows = the composite backward's intermediate rows [n_views][n][12]
 * (dvs_get_bwd_intermediates) — call it between dvs_raster_backward_composite and dvs_raster_backward_project (which consumes them). */
int dvs_densify_accumulate_rows(void* stream, int n, int n_views, const int32_t* radii, const float* rows, int width, int height,
                                float* grad_accum, float* denom, int32_t* max_radii);
/* out[i] = max over the views of radii[v][i] (> 0: visible in at least one view of the pass; the visible-only Adam step's gate) */
int dvs_any_view_radius(void* stream, int n, int n_views, const int32_t* radii, int32_t* out);
/* scratch: at least (n/256 + 2) uint32; new_count: DEVICE uint64. */
int dvs_densify_plan(void* stream, int n, const float* opacity, const float* scale, const float* grad_accum, const float* denom,
                     const int32_t* max_radii, const dvs_densify_params* prm, uint8_t* action, uint32_t* offsets, uint32_t* scratch,
                     uint64_t* new_count);
/* src/dst: six DEVICE arrays in A0 order (pos, sh0, shN, opacity, scale, rot); dst sized for the new count. */
int dvs_densify_apply(void* stream, int n, const uint8_t* action, const uint32_t* offsets, const dvs_densify_params* prm, int mode,
                      const float* const src[6], float* const dst[6], int new_n);
/* opacity reset (--resetAlphaEvery): opacity = min(opacity, logit(max_opacity)); zeroes the matching Adam moments if given. */
int dvs_reset_opacity(void* stream, int n, float* opacity, float max_opacity, float* adam_m, float* adam_v);

/* ---- MCMC densification strategy (--densifyStrategy 1, main.cpp:20,29; `noiselr` gs_train.cpp:97) ---------------------------------------
 * The published rule the reference names ("3D Gaussian Splatting as Markov Chain Monte Carlo"); its own code is in the closed plugin.
 * All arrays DEVICE, in A0 order (pos, sh0, shN, opacity, scale, rot), sized for `capacity` splats; m / v = Adam moments (entries may
 * be NULL). Asynchronous on `stream`, no host round trip (the dead count is optional, read back asynchronously).
 *   dvs_mcmc_relocate : every dead splat (sigmoid(opacity) <= min_opacity) becomes a copy of a live splat drawn with probability
 *                       ~ opacity; a splat drawn c times and its c copies take opacity 1-(1-o)^(1/(c+1)) and the scale factor that
 *                       preserves the rendered contribution; moments of everything touched are zeroed.
 *   dvs_mcmc_grow     : n_new more copies drawn the same way are appended at [n, n+n_new) (caller: 5 % growth up to capMax).
 *   dvs_mcmc_add_noise: after each optimizer step  pos += Sigma z gate(o) lr,  z ~ N(0,I), gate = sigmoid(-100 (o - 0.005));
 *                       lr = noiselr * position learning rate.
 *   dvs_mcmc_regularize: adds the gradients of  opacity_reg * mean(sigmoid(opacity)) + scale_reg * mean(exp(scale)).
 * scratch: dvs_mcmc_scratch_bytes(capacity) bytes, zeroed once with dvs_mcmc_init_scratch (the draw counters live there). */
typedef struct dvs_mcmc_sets { float* param[6]; float* m[6]; float* v[6]; } dvs_mcmc_sets;
size_t dvs_mcmc_scratch_bytes(int capacity);
int dvs_mcmc_init_scratch(void* stream, void* scratch, int capacity);
int dvs_mcmc_relocate(void* stream, int n, const dvs_mcmc_sets* sets, float min_opacity, uint32_t seed, int shn_layout, void* scratch,
                      int capacity, uint32_t* n_dead_out /* host (pinned), may be NULL */);
int dvs_mcmc_grow(void* stream, int n, int n_new, const dvs_mcmc_sets* sets, float min_opacity, uint32_t seed, int shn_layout, void* scratch,
                  int capacity);
int dvs_mcmc_add_noise(void* stream, int n, float* pos, const float* scale, const float* rot, const float* opacity, float lr, uint32_t seed);
int dvs_mcmc_regularize(void* stream, int n, const float* opacity, const float* scale, float* g_opacity, float* g_scale, float opacity_reg,
                        float scale_reg);
/* the same two for the splats [first, first + count) of the n (full-array pointers): bit-identical to the whole-array calls — the noise is a
 * function of the global splat index, the regularisers' means are over n. For a data-parallel step that updates its parameters chunk by chunk. */
int dvs_mcmc_add_noise_range(void* stream, int n, int first, int count, float* pos, const float* scale, const float* rot, const float* opacity, float lr,
                             uint32_t seed);
int dvs_mcmc_regularize_range(void* stream, int n, int first, int count, const float* opacity, const float* scale, float* g_opacity, float* g_scale,
                              float opacity_reg, float scale_reg);

#ifdef __cplusplus
}
#endif
#endif
