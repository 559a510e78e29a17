// densify.hip — adaptive density control (include/dvs_train.h), HBM-streaming kernels:
// statistics accumulation, plan (action + exclusive scan of output counts) and apply (scatter of parameters / optimizer moments).
#include <hip/hip_runtime.h>
#include "../../include/dvs_train.h"
#include "../../include/dvs_raster.h"
#include "dvs_device.h"

#define DB 256

__device__ __forceinline__ uint32_t d_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t d_block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const uint32_t lane = d_lane(), wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < DB / 64; ++w) { const uint32_t t = tmp[w]; if ((uint32_t)w < wave) wbase += t; tot += t; }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}
__device__ __forceinline__ int64_t d_shn_index(int layout, int i, int e) {
    return layout == DVS_SHN_TILED ? ((((int64_t)(i >> 6) * 12 + (e >> 2)) * 64 + (i & 63)) * 4 + (e & 3)) : ((int64_t)i * 45 + e);
}
// counter-based RNG: one 32-bit hash per (seed, splat, draw)
__device__ __forceinline__ uint32_t d_hash(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float d_uniform(uint32_t h) { return ((h >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float d_normal(uint32_t seed, uint32_t id, uint32_t k) {
    const float u1 = d_uniform(d_hash(seed, id, 2 * k)), u2 = d_uniform(d_hash(seed, id, 2 * k + 1));
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}

__global__ void __launch_bounds__(DB)
k_densify_accumulate(int n, const int* __restrict__ radii, const float2* __restrict__ absgrad, float half_w, float half_h,
                     float* __restrict__ grad_accum, float* __restrict__ denom, int* __restrict__ max_radii) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float2 g = absgrad[i];
    const float gx = g.x * half_w, gy = g.y * half_h;
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
    max_radii[i] = max(max_radii[i], r);
}

// the same rule for the n_views views of a multi-view pass, read from the composite backward's rows BEFORE the preprocess backward
// consumes them (row floats 9, 10 = sum over the view's pixels of |dL/dmean2D| x, y: exactly what A9 hands out as absgrad2d)
__global__ void __launch_bounds__(DB)
k_densify_accumulate_rows(int n, int n_views, const int* __restrict__ radii, const float* __restrict__ rows, float half_w, float half_h,
                          float* __restrict__ grad_accum, float* __restrict__ denom, int* __restrict__ max_radii) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f, cnt = 0.f;
    int rmax = 0;
    for (int v = 0; v < n_views; ++v) {
        const int r = radii[(size_t)v * n + i];
        if (r <= 0) continue;
        const float* row = rows + ((size_t)v * n + i) * 12;
        const float gx = row[9] * half_w, gy = row[10] * half_h;
        acc += sqrtf(gx * gx + gy * gy);
        cnt += 1.0f;
        rmax = max(rmax, r);
    }
    if (cnt > 0.f) {
        grad_accum[i] += acc;
        denom[i] += cnt;
        max_radii[i] = max(max_radii[i], rmax);
    }
}
// radius > 0 in any of the views (visibleAdam with several views per step)
__global__ void __launch_bounds__(DB) k_any_view_radius(int n, int n_views, const int* __restrict__ radii, int* __restrict__ out) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= n) return;
    int r = 0;
    for (int v = 0; v < n_views; ++v) r = max(r, radii[(size_t)v * n + i]);
    out[i] = r;
}

__device__ __forceinline__ int d_action(int i, const float* opacity, const float* scale, const float* grad_accum, const float* denom,
                                        const int* max_radii, const dvs_densify_params& p) {
    const float op = 1.0f / (1.0f + __expf(-opacity[i]));
    const float smax = __expf(fmaxf(scale[3 * (int64_t)i], fmaxf(scale[3 * (int64_t)i + 1], scale[3 * (int64_t)i + 2])));
    if (op < p.min_opacity) return DVS_DENSIFY_PRUNE;
    if (p.max_world_scale > 0.f && smax > p.max_world_scale) return DVS_DENSIFY_PRUNE;
    if (p.max_screen_radius > 0 && max_radii[i] > p.max_screen_radius) return DVS_DENSIFY_PRUNE;
    const float d = denom[i];
    const float avg = d > 0.f ? grad_accum[i] / d : 0.f;
    if (!(avg >= p.grad_threshold)) return DVS_DENSIFY_KEEP;
    return smax > p.scale_threshold ? DVS_DENSIFY_SPLIT : DVS_DENSIFY_CLONE;
}
__device__ __forceinline__ uint32_t d_count(int a) { return a == DVS_DENSIFY_PRUNE ? 0u : (a == DVS_DENSIFY_KEEP ? 1u : 2u); }

__global__ void __launch_bounds__(DB)
k_densify_blocksum(int n, const float* __restrict__ opacity, const float* __restrict__ scale, const float* __restrict__ grad_accum,
                   const float* __restrict__ denom, const int* __restrict__ max_radii, dvs_densify_params p, uint8_t* __restrict__ action,
                   uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t tmp[DB / 64 + 1];
    const int i = blockIdx.x * DB + threadIdx.x;
    int a = DVS_DENSIFY_PRUNE;
    if (i < n) { a = d_action(i, opacity, scale, grad_accum, denom, max_radii, p); action[i] = (uint8_t)a; }
    uint32_t tot;
    (void)d_block_excl_scan(i < n ? d_count(a) : 0u, tmp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// single workgroup: exclusive scan of the block sums; growth beyond cap_max is cut by demoting CLONE/SPLIT to KEEP later (k_offsets)
__global__ void __launch_bounds__(DB)
k_densify_scan_blocks(uint32_t* __restrict__ block_sums, uint32_t nb, uint64_t* __restrict__ total) {
    __shared__ uint32_t tmp[DB / 64 + 1];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nb; base += DB) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < nb ? block_sums[idx] : 0u;
        uint32_t tot;
        const uint32_t ex = d_block_excl_scan(v, tmp, &tot);
        if (idx < nb) block_sums[idx] = (uint32_t)(carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(DB)
k_densify_offsets(int n, uint8_t* __restrict__ action, const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ offsets,
                  int cap_max, uint64_t* __restrict__ total) {
    __shared__ uint32_t tmp[DB / 64 + 1];
    const int i = blockIdx.x * DB + threadIdx.x;
    const int a = i < n ? action[i] : DVS_DENSIFY_PRUNE;
    uint32_t tot;
    const uint32_t off = d_block_excl_scan(i < n ? d_count(a) : 0u, tmp, &tot) + block_offsets[blockIdx.x];
    if (i < n) offsets[i] = off;
    (void)cap_max; (void)total;
}

// mode 0: parameters; mode 1: optimizer moments
__global__ void __launch_bounds__(DB)
k_densify_apply(int n, const uint8_t* __restrict__ action, const uint32_t* __restrict__ offsets, dvs_densify_params p, int mode,
                const float* __restrict__ s_pos, const float* __restrict__ s_sh0, const float* __restrict__ s_shn,
                const float* __restrict__ s_op, const float* __restrict__ s_sc, const float* __restrict__ s_rot,
                float* __restrict__ d_pos, float* __restrict__ d_sh0, float* __restrict__ d_shn, float* __restrict__ d_op,
                float* __restrict__ d_sc, float* __restrict__ d_rot, int new_n) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= n) return;
    const int a = action[i];
    if (a == DVS_DENSIFY_PRUNE) return;
    const uint32_t o0 = offsets[i];
    const int copies = a == DVS_DENSIFY_KEEP ? 1 : 2;
    float pos[3], sc[3], q[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) { pos[k] = s_pos[3 * (int64_t)i + k]; sc[k] = s_sc[3 * (int64_t)i + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = s_rot[4 * (int64_t)i + k];
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (mode == 0 && a == DVS_DENSIFY_SPLIT) {
        const float qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float iq = qn > 0.f ? 1.0f / qn : 0.f;
        dvs_quat_to_rot(q[0] * iq, q[1] * iq, q[2] * iq, q[3] * iq, R);
    }
    for (int c = 0; c < copies; ++c) {
        const int j = (int)o0 + c;
        if (j >= new_n) break;
        const bool fresh = mode == 1 && !(a == DVS_DENSIFY_KEEP || (a == DVS_DENSIFY_CLONE && c == 0));   // moments of new splats = 0
        float np[3] = {pos[0], pos[1], pos[2]}, ns[3] = {sc[0], sc[1], sc[2]};
        if (mode == 0 && a == DVS_DENSIFY_SPLIT) {
            // sample from the splat's own Gaussian: pos + R diag(s) z, z ~ N(0, I); children shrink by 1.6
            float z[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) z[k] = d_normal(p.seed, (uint32_t)i, (uint32_t)(c * 3 + k)) * __expf(sc[k]);
#pragma unroll
            for (int r = 0; r < 3; ++r) np[r] = pos[r] + R[r * 3] * z[0] + R[r * 3 + 1] * z[1] + R[r * 3 + 2] * z[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) ns[k] = sc[k] - 0.47000363f;     // log(1.6)
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d_pos[3 * (int64_t)j + k] = fresh ? 0.f : np[k];
            d_sc[3 * (int64_t)j + k] = fresh ? 0.f : ns[k];
            d_sh0[3 * (int64_t)j + k] = fresh ? 0.f : s_sh0[3 * (int64_t)i + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) d_rot[4 * (int64_t)j + k] = fresh ? 0.f : q[k];
        float op = s_op[i];
        if (mode == 0 && p.revised_opacity && a != DVS_DENSIFY_KEEP) {        // o' = 1 - sqrt(1 - o) for both results
            const float o = 1.0f / (1.0f + __expf(-op));
            const float no = fminf(fmaxf(1.0f - sqrtf(1.0f - o), 1e-6f), 1.0f - 1e-6f);
            op = __logf(no / (1.0f - no));
        }
        d_op[j] = fresh ? 0.f : op;
        for (int e = 0; e < 45; ++e) d_shn[d_shn_index(p.shn_layout, j, e)] = fresh ? 0.f : s_shn[d_shn_index(p.shn_layout, i, e)];
    }
}

__global__ void __launch_bounds__(DB)
k_reset_opacity(int n, float* __restrict__ opacity, float logit_max, float* __restrict__ m, float* __restrict__ v) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i >= n) return;
    opacity[i] = fminf(opacity[i], logit_max);
    if (m) m[i] = 0.f;
    if (v) v[i] = 0.f;
}

extern "C" {
int dvs_densify_accumulate(void* stream, int n, const int32_t* radii, const float* absgrad2d, int width, int height, float* grad_accum,
                           float* denom, int32_t* max_radii) {
    if (n < 0 || (n > 0 && (!radii || !absgrad2d || !grad_accum || !denom || !max_radii))) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    hipLaunchKernelGGL(k_densify_accumulate, dim3((n + DB - 1) / DB), dim3(DB), 0, (hipStream_t)stream, n, radii, (const float2*)absgrad2d,
                       0.5f * (float)width, 0.5f * (float)height, grad_accum, denom, max_radii);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_densify_accumulate_rows(void* stream, int n, int n_views, const int32_t* radii, const float* rows, int width, int height,
                                float* grad_accum, float* denom, int32_t* max_radii) {
    if (n < 0 || n_views < 1 || (n > 0 && (!radii || !rows || !grad_accum || !denom || !max_radii))) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    hipLaunchKernelGGL(k_densify_accumulate_rows, dim3((n + DB - 1) / DB), dim3(DB), 0, (hipStream_t)stream, n, n_views, radii, rows,
                       0.5f * (float)width, 0.5f * (float)height, grad_accum, denom, max_radii);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_any_view_radius(void* stream, int n, int n_views, const int32_t* radii, int32_t* out) {
    if (n < 0 || n_views < 1 || (n > 0 && (!radii || !out))) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    hipLaunchKernelGGL(k_any_view_radius, dim3((n + DB - 1) / DB), dim3(DB), 0, (hipStream_t)stream, n, n_views, radii, out);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_densify_plan(void* stream, int n, const float* opacity, const float* scale, const float* grad_accum, const float* denom,
                     const int32_t* max_radii, const dvs_densify_params* prm, uint8_t* action, uint32_t* offsets, uint32_t* scratch,
                     uint64_t* new_count) {
    if (n < 0 || !prm || !new_count || (n > 0 && (!opacity || !scale || !grad_accum || !denom || !max_radii || !action || !offsets || !scratch)))
        return DVS_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nb = (uint32_t)((n + DB - 1) / DB);
    if (nb) hipLaunchKernelGGL(k_densify_blocksum, dim3(nb), dim3(DB), 0, st, n, opacity, scale, grad_accum, denom, max_radii, *prm, action, scratch);
    hipLaunchKernelGGL(k_densify_scan_blocks, dim3(1), dim3(DB), 0, st, scratch, nb, new_count);
    if (nb) hipLaunchKernelGGL(k_densify_offsets, dim3(nb), dim3(DB), 0, st, n, action, scratch, offsets, prm->cap_max, new_count);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_densify_apply(void* stream, int n, const uint8_t* action, const uint32_t* offsets, const dvs_densify_params* prm, int mode,
                      const float* const src[6], float* const dst[6], int new_n) {
    if (n < 0 || !prm || (n > 0 && (!action || !offsets || !src || !dst))) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    hipLaunchKernelGGL(k_densify_apply, dim3((n + DB - 1) / DB), dim3(DB), 0, (hipStream_t)stream, n, action, offsets, *prm, mode, src[0], src[1],
                       src[2], src[3], src[4], src[5], dst[0], dst[1], dst[2], dst[3], dst[4], dst[5], new_n);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_reset_opacity(void* stream, int n, float* opacity, float max_opacity, float* adam_m, float* adam_v) {
    if (n < 0 || (n > 0 && !opacity) || !(max_opacity > 0.f && max_opacity < 1.f)) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    hipLaunchKernelGGL(k_reset_opacity, dim3((n + DB - 1) / DB), dim3(DB), 0, (hipStream_t)stream, n, opacity,
                       logf(max_opacity / (1.f - max_opacity)), adam_m, adam_v);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
}
