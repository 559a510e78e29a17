// ssim.hip — fused SSIM forward / backward (include/dvs_train.h), the "next" row 3 of SURVEY.md §8(f): the term the
// reference weights with --ssim 0.2 (application/diverseshot-cli/source/main.cpp:24-25; its implementation is in the closed
// plugin). Image-space stencil, HBM-bound: one 16x16 output tile per workgroup, the 26x26 input patch (5-pixel halo) of both
// images staged in LDS, separable 11-tap Gaussian done as a horizontal pass into LDS followed by a vertical pass.
//
//   mu1 = G*x, mu2 = G*y, s1 = G*x^2 - mu1^2, s2 = G*y^2 - mu2^2, s12 = G*xy - mu1 mu2
//   ssim = ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2))
// backward (x = rendered image, y = target, constant):
//   dL/dx = G*(dm_dmu1) + 2 x G*(dm_ds1) + y G*(dm_ds12)     with the three per-pixel maps produced by the forward.
#include <hip/hip_runtime.h>
#include <climits>
#include "../../include/dvs_train.h"
#include "../../include/dvs_raster.h"

#define ST 16
#define HALO 5
#define SP (ST + 2 * HALO)        // 26
#define SSIM_C1 0.0001f
#define SSIM_C2 0.0009f

__constant__ float c_gauss[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                                  0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
                                  0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};

__device__ __forceinline__ float block_sum256(float v, float* tmp) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    return tmp[0] + tmp[1] + tmp[2] + tmp[3];
}

// load the 26x26 patches of NP planes (zero outside the image) into LDS. All 3*NP global loads are issued before the first
// LDS write (clamped address + select instead of a branch): with a branch per element the loads serialise and the kernel
// is bound by 3*NP dependent HBM round trips per workgroup.
template <int NP>
__device__ __forceinline__ void load_patches(const float* const (&src)[NP], int W, int H, int x0, int y0, float (*const (&dst)[NP])[SP]) {
    constexpr int IT = (SP * SP + ST * ST - 1) / (ST * ST);
    float r[NP][IT];
    int at[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = threadIdx.x + it * ST * ST;
        const int py = e / SP, px = e - py * SP;
        const int gx = x0 + px - HALO, gy = y0 + py - HALO;
        const bool inb = e < SP * SP && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t idx = inb ? (size_t)gy * W + gx : 0;
        at[it] = e < SP * SP ? (inb ? e : ~e) : INT_MIN;
#pragma unroll
        for (int p = 0; p < NP; ++p) r[p][it] = src[p][idx];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (at[it] == INT_MIN) continue;
        const int e = at[it] >= 0 ? at[it] : ~at[it];
#pragma unroll
        for (int p = 0; p < NP; ++p) (&dst[p][0][0])[e] = at[it] >= 0 ? r[p][it] : 0.f;
    }
}

__global__ void __launch_bounds__(ST * ST)
k_ssim_fwd(const float* __restrict__ img, const float* __restrict__ tgt, int W, int H, float* __restrict__ dm_dmu1,
           float* __restrict__ dm_ds1, float* __restrict__ dm_ds12, float* __restrict__ ssim_sum) {
    __shared__ float sx[SP][SP], sy[SP][SP];
    __shared__ float hx[SP][ST], hy[SP][ST], hxx[SP][ST], hyy[SP][ST], hxy[SP][ST];     // after the horizontal pass
    __shared__ float tmp[4];
    const int ch = blockIdx.z;
    const size_t plane = (size_t)ch * W * H;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    {
        const float* const src[2] = {img + plane, tgt + plane};
        float (*const dst[2])[SP] = {sx, sy};
        load_patches<2>(src, W, H, x0, y0, dst);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SP * ST; e += ST * ST) {
        const int py = e / ST, px = e % ST;
        float ax = 0.f, ay = 0.f, axx = 0.f, ayy = 0.f, axy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = c_gauss[k], a = sx[py][px + k], b = sy[py][px + k];
            ax += g * a; ay += g * b; axx += g * a * a; ayy += g * b * b; axy += g * a * b;
        }
        hx[py][px] = ax; hy[py][px] = ay; hxx[py][px] = axx; hyy[py][px] = ayy; hxy[py][px] = axy;
    }
    __syncthreads();
    const int lx = threadIdx.x % ST, ly = threadIdx.x / ST;
    float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float g = c_gauss[k];
        mu1 += g * hx[ly + k][lx]; mu2 += g * hy[ly + k][lx]; exx += g * hxx[ly + k][lx]; eyy += g * hyy[ly + k][lx]; exy += g * hxy[ly + k][lx];
    }
    const int gx = x0 + lx, gy = y0 + ly;
    float local = 0.f;
    if (gx < W && gy < H) {
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = exx - mu1s, s2 = eyy - mu2s, s12 = exy - mu12;
        const float A = mu1s + mu2s + SSIM_C1, B = s1 + s2 + SSIM_C2, Cn = 2.f * mu12 + SSIM_C1, Dn = 2.f * s12 + SSIM_C2;
        const float iAB = 1.f / (A * B), iA = iAB * B, iB = iAB * A;       // one division per pixel
        const float m = Cn * Dn * iAB;
        local = m;
        const size_t o = plane + (size_t)gy * W + gx;
        dm_dmu1[o] = 2.f * iAB * (mu2 * (Dn - Cn) + mu1 * Cn * Dn * (iB - iA));
        dm_ds1[o] = -m * iB;
        dm_ds12[o] = 2.f * Cn * iAB;
    }
    const float s = block_sum256(local, tmp);
    // one atomic per workgroup, spread over DVS_SSIM_SLOTS addresses by the linear workgroup id (same-address atomics serialise at
    // ~0.3 us each on the memory side: with 64 slots they were 0.1 ms of this kernel)
    if (threadIdx.x == 0 && ssim_sum) atomicAdd(ssim_sum + (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & (DVS_SSIM_SLOTS - 1)), s);
}

// L1 = true: the kernel also adds the L1 term of the photometric loss, dL = l1_scale * sign(x - y) + scale * dSSIM/dx, and accumulates
// l1_scale * |x - y| into l1_sum[DVS_SSIM_SLOTS] — one pass over the image instead of two kernels and a read-modify-write of dL.
template <bool ACCUM, bool L1>
__global__ void __launch_bounds__(ST * ST)
k_ssim_bwd(const float* __restrict__ img, const float* __restrict__ tgt, int W, int H, const float* __restrict__ dm_dmu1,
           const float* __restrict__ dm_ds1, const float* __restrict__ dm_ds12, float scale, float* __restrict__ dL, float l1_scale,
           float* __restrict__ l1_sum) {
    __shared__ float sa[SP][SP], sb[SP][SP], sc[SP][SP];
    __shared__ float ha[SP][ST], hb[SP][ST], hc[SP][ST];
    const int ch = blockIdx.z;
    const size_t plane = (size_t)ch * W * H;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    {
        const float* const src[3] = {dm_dmu1 + plane, dm_ds1 + plane, dm_ds12 + plane};
        float (*const dst[3])[SP] = {sa, sb, sc};
        load_patches<3>(src, W, H, x0, y0, dst);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SP * ST; e += ST * ST) {
        const int py = e / ST, px = e % ST;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) { const float g = c_gauss[k]; a += g * sa[py][px + k]; b += g * sb[py][px + k]; c += g * sc[py][px + k]; }
        ha[py][px] = a; hb[py][px] = b; hc[py][px] = c;
    }
    __syncthreads();
    const int lx = threadIdx.x % ST, ly = threadIdx.x / ST;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) { const float g = c_gauss[k]; a += g * ha[ly + k][lx]; b += g * hb[ly + k][lx]; c += g * hc[ly + k][lx]; }
    const int gx = x0 + lx, gy = y0 + ly;
    float l1_local = 0.f;
    if (gx < W && gy < H) {
        const size_t o = plane + (size_t)gy * W + gx;
        const float x = img[o], y = tgt[o];
        float g = scale * (a + 2.f * x * b + y * c);
        if (L1) {
            const float d = x - y;
            g += d > 0.f ? l1_scale : (d < 0.f ? -l1_scale : 0.f);
            l1_local = fabsf(d) * l1_scale;
        }
        dL[o] = ACCUM ? dL[o] + g : g;
    }
    if (L1) {
        __shared__ float tmp[4];
        const float t = block_sum256(l1_local, tmp);
        if (threadIdx.x == 0 && l1_sum) atomicAdd(l1_sum + (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & (DVS_SSIM_SLOTS - 1)), t);
    }
}

extern "C" {
int dvs_ssim_forward(void* stream, const float* img, const float* target, int width, int height, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, float* ssim_sum) {
    if (!img || !target || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || width <= 0 || height <= 0) return DVS_ERR_INVALID;
    const dim3 grid((width + ST - 1) / ST, (height + ST - 1) / ST, 3);
    hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(ST * ST), 0, (hipStream_t)stream, img, target, width, height, dm_dmu1, dm_dsigma1_sq,
                       dm_dsigma12, ssim_sum);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_ssim_backward(void* stream, const float* img, const float* target, int width, int height, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, float scale, float* dL_dimg, int accumulate) {
    if (!img || !target || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg || width <= 0 || height <= 0) return DVS_ERR_INVALID;
    const dim3 grid((width + ST - 1) / ST, (height + ST - 1) / ST, 3);
    // d(mean SSIM)/dx: the mean is over 3*W*H values
    const float s = scale / (3.0f * (float)width * (float)height);
    if (accumulate)
        hipLaunchKernelGGL((k_ssim_bwd<true, false>), grid, dim3(ST * ST), 0, (hipStream_t)stream, img, target, width, height, dm_dmu1,
                           dm_dsigma1_sq, dm_dsigma12, s, dL_dimg, 0.f, (float*)nullptr);
    else
        hipLaunchKernelGGL((k_ssim_bwd<false, false>), grid, dim3(ST * ST), 0, (hipStream_t)stream, img, target, width, height, dm_dmu1,
                           dm_dsigma1_sq, dm_dsigma12, s, dL_dimg, 0.f, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_loss_l1_ssim_backward(void* stream, const float* img, const float* target, int width, int height, const float* dm_dmu1,
                              const float* dm_dsigma1_sq, const float* dm_dsigma12, float ssim_weight, float* dL_dimg, float* l1_sum) {
    if (!img || !target || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg || width <= 0 || height <= 0 ||
        !(ssim_weight >= 0.f && ssim_weight <= 1.f))
        return DVS_ERR_INVALID;
    const dim3 grid((width + ST - 1) / ST, (height + ST - 1) / ST, 3);
    const float count = 3.0f * (float)width * (float)height;
    // L = (1-w) mean|x-y| + w (1 - mean SSIM)  ->  dL/dx = (1-w)/count sign(x-y) - w/count dSSIM/dx
    hipLaunchKernelGGL((k_ssim_bwd<false, true>), grid, dim3(ST * ST), 0, (hipStream_t)stream, img, target, width, height, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, -ssim_weight / count, dL_dimg, (1.0f - ssim_weight) / count, l1_sum);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
}
