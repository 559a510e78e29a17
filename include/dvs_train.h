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
 *                      *ssim_sum += sum of the SSIM map (divide by 3*H*W for the mean);
 *   dvs_ssim_backward: dL_dimg[3,H,W] (+)= scale * d(mean SSIM)/d(img)   (scale = -w to minimise 1 - SSIM; accumulate = add).
 * All pointers DEVICE; asynchronous on `stream`. */
int dvs_ssim_forward(void* stream, const float* img, const float* target, int width, int height, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, float* ssim_sum);
int dvs_ssim_backward(void* stream, const float* img, const float* target, int width, int height, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, float scale, float* dL_dimg, int accumulate);

/* Fused Adam over one parameter array (count floats): m, v are the moment arrays (same size, DEVICE).
 * step is 1-based. Asynchronous. */
int dvs_adam_step(void* stream, float* param, const float* grad, float* m, float* v, size_t count, float lr, float beta1,
                  float beta2, float eps, int step);

#ifdef __cplusplus
}
#endif
#endif
