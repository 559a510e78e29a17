// train_ops.hip — image loss gradient and fused Adam (include/dvs_train.h): HBM-streaming, 16 B per lane.
// These are SURVEY.md §8(f) "next" rows 2-3, kept minimal; the reference's versions are in the closed plugin.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/dvs_train.h"
#include "../../include/dvs_raster.h"

#define TB 256

__device__ __forceinline__ float block_sum(float v, float* tmp) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    return tmp[0] + tmp[1] + tmp[2] + tmp[3];
}

template <bool L1>
__global__ void __launch_bounds__(TB)
k_loss_grad(const float* __restrict__ rgb, const float* __restrict__ target, size_t count, float scale, float* __restrict__ dL,
            float* __restrict__ loss_accum) {
    __shared__ float tmp[4];
    float local = 0.f;
    const size_t nvec = count >> 2;
    for (size_t v = (size_t)blockIdx.x * TB + threadIdx.x; v < nvec; v += (size_t)gridDim.x * TB) {
        const float4 a = reinterpret_cast<const float4*>(rgb)[v], b = reinterpret_cast<const float4*>(target)[v];
        float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w), g;
        if (L1) {
            local += fabsf(d.x) + fabsf(d.y) + fabsf(d.z) + fabsf(d.w);
            g = make_float4(d.x > 0.f ? scale : (d.x < 0.f ? -scale : 0.f), d.y > 0.f ? scale : (d.y < 0.f ? -scale : 0.f),
                            d.z > 0.f ? scale : (d.z < 0.f ? -scale : 0.f), d.w > 0.f ? scale : (d.w < 0.f ? -scale : 0.f));
        } else {
            local += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            g = make_float4(d.x * scale, d.y * scale, d.z * scale, d.w * scale);
        }
        reinterpret_cast<float4*>(dL)[v] = g;
    }
    if (blockIdx.x == 0)
        for (size_t e = (nvec << 2) + threadIdx.x; e < count; e += TB) {
            const float d = rgb[e] - target[e];
            local += L1 ? fabsf(d) : d * d;
            dL[e] = L1 ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : d * scale;
        }
    const float s = block_sum(local, tmp);
    if (threadIdx.x == 0 && loss_accum) atomicAdd(loss_accum, L1 ? s * scale : 0.5f * s * scale);
}

__global__ void __launch_bounds__(TB)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t count, float lr,
       float b1, float b2, float eps, float bc1, float bc2) {
    const size_t nvec = count >> 2;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < nvec; i += (size_t)gridDim.x * TB) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
#define DVS_ADAM1(c)                                                   \
        mm.c = b1 * mm.c + (1.f - b1) * gg.c;                          \
        vv.c = b2 * vv.c + (1.f - b2) * gg.c * gg.c;                   \
        pp.c -= lr * (mm.c * bc1) / (sqrtf(vv.c * bc2) + eps);
        DVS_ADAM1(x) DVS_ADAM1(y) DVS_ADAM1(z) DVS_ADAM1(w)
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (size_t e = (nvec << 2) + threadIdx.x; e < count; e += TB) {
            const float ge = g[e];
            const float me = b1 * m[e] + (1.f - b1) * ge, ve = b2 * v[e] + (1.f - b2) * ge * ge;
            m[e] = me; v[e] = ve;
            p[e] -= lr * (me * bc1) / (sqrtf(ve * bc2) + eps);
        }
}

// ---- grouped Adam: one launch for every parameter group --------------------------------------------------------------------
struct AdamGroupDev {
    float* p; const float* g; float* m; float* v;
    unsigned long long nvec;        // float4 items this group contributes (after the active-chunk compaction)
    unsigned long long tail0, tail1;// scalar tail [tail0, tail1)
    float lr; int width; int tiled; int active_vec;      // active_vec: float4 per 64-splat tile that are processed (tiled only)
    unsigned block0;                // first workgroup of this group
};
struct AdamGroupsDev { AdamGroupDev g[DVS_ADAM_MAX_GROUPS]; int n; };

#define ADAM_VPT 2                  // float4 per thread: two independent load batches in flight

__device__ __forceinline__ bool adam_vis(const int* __restrict__ vis, int n_splats, const AdamGroupDev& G, unsigned long long e) {
    if (!vis) return true;
    unsigned long long splat;
    if (G.tiled) splat = (e / DVS_SHN_TILE_FLOATS) * 64ull + ((e % DVS_SHN_TILE_FLOATS) >> 2 & 63ull);
    else splat = e / (unsigned)G.width;
    return splat < (unsigned long long)n_splats && vis[splat] > 0;
}

__global__ void __launch_bounds__(TB)
k_adam_groups(const AdamGroupsDev G, const int* __restrict__ vis, int n_splats, float b1, float b2, float eps, float bc1, float bc2) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < DVS_ADAM_MAX_GROUPS; ++k) gi += (k < G.n && blockIdx.x >= G.g[k].block0) ? 1 : 0;
    const AdamGroupDev& A = G.g[gi];
    const unsigned long long base = (unsigned long long)(blockIdx.x - A.block0) * (TB * ADAM_VPT) + threadIdx.x;
    unsigned long long idx[ADAM_VPT];
    bool on[ADAM_VPT];
    float4 pp[ADAM_VPT], mm[ADAM_VPT], vv[ADAM_VPT], gg[ADAM_VPT];
#pragma unroll
    for (int u = 0; u < ADAM_VPT; ++u) {
        const unsigned long long j = base + (unsigned long long)u * TB;
        on[u] = j < A.nvec;
        unsigned long long i = j;
        if (A.tiled && A.active_vec != DVS_SHN_TILE_FLOATS / 4) {
            const unsigned long long tile = j / (unsigned)A.active_vec;
            i = tile * (DVS_SHN_TILE_FLOATS / 4) + (j - tile * (unsigned)A.active_vec);
        }
        idx[u] = on[u] ? i : 0;
    }
#pragma unroll
    for (int u = 0; u < ADAM_VPT; ++u)
        if (on[u]) {
            gg[u] = reinterpret_cast<const float4*>(A.g)[idx[u]];
            pp[u] = reinterpret_cast<float4*>(A.p)[idx[u]];
            mm[u] = reinterpret_cast<float4*>(A.m)[idx[u]];
            vv[u] = reinterpret_cast<float4*>(A.v)[idx[u]];
        }
#pragma unroll
    for (int u = 0; u < ADAM_VPT; ++u)
        if (on[u]) {
            const unsigned long long e0 = idx[u] << 2;
            bool any = false;
#define DVS_ADAM1V(c, k)                                                                         \
            if (adam_vis(vis, n_splats, A, e0 + k)) {                                                      \
                mm[u].c = b1 * mm[u].c + (1.f - b1) * gg[u].c;                                   \
                vv[u].c = b2 * vv[u].c + (1.f - b2) * gg[u].c * gg[u].c;                         \
                pp[u].c -= A.lr * (mm[u].c * bc1) / (sqrtf(vv[u].c * bc2) + eps);                \
                any = true;                                                                      \
            }
            DVS_ADAM1V(x, 0) DVS_ADAM1V(y, 1) DVS_ADAM1V(z, 2) DVS_ADAM1V(w, 3)
            if (any) {
                reinterpret_cast<float4*>(A.p)[idx[u]] = pp[u];
                reinterpret_cast<float4*>(A.m)[idx[u]] = mm[u];
                reinterpret_cast<float4*>(A.v)[idx[u]] = vv[u];
            }
        }
    if (blockIdx.x == A.block0)
        for (unsigned long long e = A.tail0 + threadIdx.x; e < A.tail1; e += TB)
            if (adam_vis(vis, n_splats, A, e)) {
                const float ge = A.g[e];
                const float me = b1 * A.m[e] + (1.f - b1) * ge, ve = b2 * A.v[e] + (1.f - b2) * ge * ge;
                A.m[e] = me; A.v[e] = ve;
                A.p[e] -= A.lr * (me * bc1) / (sqrtf(ve * bc2) + eps);
            }
}

static int grid_for(size_t count) {
    size_t b = (count / 4 + TB - 1) / TB;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;          // 256 CUs x 8 workgroups, grid-stride the rest
    return (int)b;
}

extern "C" {
int dvs_l1_loss_grad_w(void* stream, const float* rgb, const float* target, size_t count, float weight, float* dL, float* loss_accum) {
    if (!rgb || !target || !dL) return DVS_ERR_INVALID;
    if (count == 0) return DVS_OK;
    hipLaunchKernelGGL(k_loss_grad<true>, dim3(grid_for(count)), dim3(TB), 0, (hipStream_t)stream, rgb, target, count,
                       weight / (float)count, dL, loss_accum);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_l1_loss_grad(void* stream, const float* rgb, const float* target, size_t count, float* dL, float* loss_accum) {
    return dvs_l1_loss_grad_w(stream, rgb, target, count, 1.0f, dL, loss_accum);
}
int dvs_l2_loss_grad(void* stream, const float* rgb, const float* target, size_t count, float scale, float* dL, float* loss_accum) {
    if (!rgb || !target || !dL) return DVS_ERR_INVALID;
    if (count == 0) return DVS_OK;
    hipLaunchKernelGGL(k_loss_grad<false>, dim3(grid_for(count)), dim3(TB), 0, (hipStream_t)stream, rgb, target, count, scale, dL,
                       loss_accum);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_adam_step(void* stream, float* param, const float* grad, float* m, float* v, size_t count, float lr, float beta1,
                  float beta2, float eps, int step) {
    if (!param || !grad || !m || !v || step < 1) return DVS_ERR_INVALID;
    if (count == 0) return DVS_OK;
    const float bc1 = 1.0f / (1.0f - powf(beta1, (float)step)), bc2 = 1.0f / (1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(k_adam, dim3(grid_for(count)), dim3(TB), 0, (hipStream_t)stream, param, grad, m, v, count, lr, beta1, beta2,
                       eps, bc1, bc2);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_adam_step_groups(void* stream, const dvs_adam_group* groups, int n_groups, float beta1, float beta2, float eps, int step,
                         const int32_t* visible, int32_t n_splats) {
    if (!groups || n_groups < 1 || n_groups > DVS_ADAM_MAX_GROUPS || step < 1) return DVS_ERR_INVALID;
    AdamGroupsDev G{};
    unsigned long long blocks = 0;
    for (int k = 0; k < n_groups; ++k) {
        const dvs_adam_group& a = groups[k];
        AdamGroupDev& d = G.g[G.n];
        if (a.count == 0) continue;
        if (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.m | (uintptr_t)a.v) & 15u) return DVS_ERR_INVALID;      // moved as float4
        if (!a.param || !a.grad || !a.m || !a.v || a.width < 1) return DVS_ERR_INVALID;
        const bool tiled = a.layout == DVS_SHN_TILED;
        if (tiled && (a.width != 45 || a.count % DVS_SHN_TILE_FLOATS != 0 || a.active_chunks < 0 || a.active_chunks > 12)) return DVS_ERR_INVALID;
        if (!tiled && a.layout != DVS_SHN_ROWS) return DVS_ERR_INVALID;
        d.p = a.param; d.g = a.grad; d.m = a.m; d.v = a.v; d.lr = a.lr; d.width = a.width; d.tiled = tiled ? 1 : 0;
        if (tiled) {
            d.active_vec = (a.active_chunks ? a.active_chunks : 12) * 64;
            d.nvec = (a.count / DVS_SHN_TILE_FLOATS) * (unsigned long long)d.active_vec;
            d.tail0 = d.tail1 = 0;
        } else {
            d.active_vec = 0;
            d.nvec = a.count >> 2; d.tail0 = d.nvec << 2; d.tail1 = a.count;
        }
        d.block0 = (unsigned)blocks;
        unsigned long long b = (d.nvec + TB * ADAM_VPT - 1) / (TB * ADAM_VPT);
        if (b < 1) b = 1;                   // the tail loop runs in the group's first workgroup
        blocks += b;
        ++G.n;
    }
    if (G.n == 0) return DVS_OK;
    if (blocks > 0x7fffffffull) return DVS_ERR_INVALID;
    const float bc1 = 1.0f / (1.0f - powf(beta1, (float)step)), bc2 = 1.0f / (1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(k_adam_groups, dim3((unsigned)blocks), dim3(TB), 0, (hipStream_t)stream, G, visible, n_splats, beta1, beta2, eps, bc1, bc2);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
}
