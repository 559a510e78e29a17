// mcmc.hip — the MCMC densification strategy (include/dvs_train.h; reference flag --densifyStrategy 1 "MCMC",
// application/diverseshot-cli/source/main.cpp:20,29 and `noiselr`, gs_train.cpp:97).
// The reference's implementation is in the closed plugin; this follows the published algorithm it names
// ("3D Gaussian Splatting as Markov Chain Monte Carlo", Kheradmand et al. 2024):
//   relocate : dead splats (opacity <= min_opacity) are moved onto live ones drawn with probability ~ opacity;
//   grow     : n_new further copies drawn the same way (the caller grows by 5 % per interval up to capMax);
//   both     : a splat drawn c times ends as c+1 identical copies whose opacity and scale are shrunk so that the rendered
//              contribution is preserved:  o' = 1 - (1-o)^(1/(c+1)),  s' = s * o / sum_{i=1..c+1} sum_{k<i} C(i-1,k) (-1)^k o'^(k+1) / sqrt(k+1);
//   noise    : after every optimizer step  pos += Sigma * z * gate(o) * lr,  z ~ N(0, I),  gate = sigmoid(-100 (o - 0.005));
//   regularise: loss += lambda_o mean(o) + lambda_s mean(exp(scale)).
// Everything stays in HBM; sampling = fp64 prefix sum of the weights + one binary search per draw (counter-based hash RNG),
// so a refinement step costs a few passes over the splat arrays and no host round trip.
#include <hip/hip_runtime.h>
#include "../../include/dvs_train.h"
#include "../../include/dvs_raster.h"
#include "dvs_device.h"

#define MB 256
#define MCMC_NMAX 51

__device__ __forceinline__ int64_t m_shn_index(int layout, int i, int e) {
    return layout == DVS_SHN_TILED ? ((((int64_t)(i >> 6) * 12 + (e >> 2)) * 64 + (i & 63)) * 4 + (e & 3)) : ((int64_t)i * 45 + e);
}
__device__ __forceinline__ uint32_t m_hash(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float m_uniform(uint32_t h) { return ((h >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float m_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// scratch layout (one allocation, see dvs_mcmc_scratch_bytes)
struct McmcScratch {
    double* cdf;          // [cap]  inclusive prefix sum of the sampling weights
    double* blk_w;        // [nb]
    uint32_t* dead_pos;   // [cap]  exclusive prefix count of dead splats
    uint32_t* blk_d;      // [nb]
    uint32_t* dead_list;  // [cap]
    uint32_t* src;        // [cap]  drawn source per destination
    uint32_t* count;      // [cap]  times each splat was drawn (zero between calls)
    double* total_w;      // [1]
    uint32_t* n_dead;     // [1]
};
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static McmcScratch carve(void* base, size_t cap) {
    const size_t nb = (cap + MB - 1) / MB + 1;
    char* p = (char*)base;
    McmcScratch s;
    s.cdf = (double*)p; p += align256(cap * 8);
    s.blk_w = (double*)p; p += align256(nb * 8);
    s.dead_pos = (uint32_t*)p; p += align256(cap * 4);
    s.blk_d = (uint32_t*)p; p += align256(nb * 4);
    s.dead_list = (uint32_t*)p; p += align256(cap * 4);
    s.src = (uint32_t*)p; p += align256(cap * 4);
    s.count = (uint32_t*)p; p += align256(cap * 4);
    s.total_w = (double*)p; p += 256;
    s.n_dead = (uint32_t*)p; p += 256;
    return s;
}
static size_t scratch_bytes(size_t cap) {
    const size_t nb = (cap + MB - 1) / MB + 1;
    return align256(cap * 8) + align256(nb * 8) + 4 * align256(cap * 4) + align256(nb * 4) + 512;
}

// block-level inclusive scans (fp64 weights, u32 dead flags) ------------------------------------------------------------
__device__ __forceinline__ void m_block_scan(double w, uint32_t d, double* tw, uint32_t* td, double* inc_w, uint32_t* exc_d,
                                             double* tot_w, uint32_t* tot_d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double iw = w; uint32_t id = d;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const double ow = __shfl_up(iw, s, 64); const uint32_t od = __shfl_up(id, s, 64);
        if (lane >= s) { iw += ow; id += od; }
    }
    if (lane == 63) { tw[wave] = iw; td[wave] = id; }
    __syncthreads();
    double bw = 0, sw = 0; uint32_t bd = 0, sd = 0;
#pragma unroll
    for (int k = 0; k < MB / 64; ++k) { if (k < wave) { bw += tw[k]; bd += td[k]; } sw += tw[k]; sd += td[k]; }
    __syncthreads();
    *inc_w = bw + iw; *exc_d = bd + id - d; *tot_w = sw; *tot_d = sd;
}
// weight = activated opacity of live splats (dead ones: 0); dead = opacity <= min_opacity
__device__ __forceinline__ void m_weight(int i, int n, const float* opacity, float min_opacity, double* w, uint32_t* dead) {
    *w = 0.0; *dead = 0u;
    if (i < n) {
        const float o = m_sigmoid(opacity[i]);
        if (o <= min_opacity) *dead = 1u; else *w = (double)o;
    }
}
__global__ void __launch_bounds__(MB)
k_mcmc_blocksum(int n, const float* __restrict__ opacity, float min_opacity, double* __restrict__ blk_w, uint32_t* __restrict__ blk_d) {
    __shared__ double tw[MB / 64]; __shared__ uint32_t td[MB / 64];
    double w, iw, sw; uint32_t d, ed, sd;
    m_weight(blockIdx.x * MB + threadIdx.x, n, opacity, min_opacity, &w, &d);
    m_block_scan(w, d, tw, td, &iw, &ed, &sw, &sd);
    if (threadIdx.x == 0) { blk_w[blockIdx.x] = sw; blk_d[blockIdx.x] = sd; }
}
__global__ void __launch_bounds__(MB)
k_mcmc_scan_blocks(uint32_t nb, double* __restrict__ blk_w, uint32_t* __restrict__ blk_d, double* __restrict__ total_w, uint32_t* __restrict__ n_dead) {
    __shared__ double tw[MB / 64]; __shared__ uint32_t td[MB / 64];
    double cw = 0; uint32_t cd = 0;
    for (uint32_t base = 0; base < nb; base += MB) {
        const uint32_t idx = base + threadIdx.x;
        const double w = idx < nb ? blk_w[idx] : 0.0; const uint32_t d = idx < nb ? blk_d[idx] : 0u;
        double iw, sw; uint32_t ed, sd;
        m_block_scan(w, d, tw, td, &iw, &ed, &sw, &sd);
        if (idx < nb) { blk_w[idx] = cw + iw - w; blk_d[idx] = cd + ed; }      // exclusive block offsets
        cw += sw; cd += sd;
    }
    if (threadIdx.x == 0) { *total_w = cw; *n_dead = cd; }
}
__global__ void __launch_bounds__(MB)
k_mcmc_cdf(int n, const float* __restrict__ opacity, float min_opacity, const double* __restrict__ blk_w, const uint32_t* __restrict__ blk_d,
           double* __restrict__ cdf, uint32_t* __restrict__ dead_list) {
    __shared__ double tw[MB / 64]; __shared__ uint32_t td[MB / 64];
    const int i = blockIdx.x * MB + threadIdx.x;
    double w, iw, sw; uint32_t d, ed, sd;
    m_weight(i, n, opacity, min_opacity, &w, &d);
    m_block_scan(w, d, tw, td, &iw, &ed, &sw, &sd);
    if (i < n) {
        cdf[i] = blk_w[blockIdx.x] + iw;
        if (d) dead_list[blk_d[blockIdx.x] + ed] = (uint32_t)i;
    }
}
// draw j: u ~ U(0, total) -> first i with cdf[i] > u (a live splat: dead ones add nothing to the cdf)
__global__ void __launch_bounds__(MB)
k_mcmc_sample(int n, int k_fixed, const uint32_t* __restrict__ k_dev, const double* __restrict__ cdf, const double* __restrict__ total_w,
              uint32_t seed, uint32_t* __restrict__ src, uint32_t* __restrict__ count) {
    const int j = blockIdx.x * MB + threadIdx.x;
    const int k = k_dev ? (int)*k_dev : k_fixed;
    if (j >= k) return;
    const double tot = *total_w;
    if (!(tot > 0.0)) { src[j] = 0xffffffffu; return; }             // nothing alive: leave the destinations alone
    const uint32_t h0 = m_hash(seed, (uint32_t)j, 0u), h1 = m_hash(seed, (uint32_t)j, 1u);
    const double u = (((double)h0 * 4294967296.0 + (double)h1) + 0.5) * (1.0 / 18446744073709551616.0) * tot;
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    src[j] = (uint32_t)lo;
    atomicAdd(&count[lo], 1u);
}
// opacity / scale of the c+1 copies that replace one splat drawn c times
__device__ __forceinline__ void m_relocation(float o, int ratio, float min_opacity, float* new_o, float* log_coeff) {
    ratio = min(max(ratio, 1), MCMC_NMAX);
    const double od = (double)o;
    const double no = 1.0 - pow(1.0 - od, 1.0 / (double)ratio);
    double denom = 0.0;
    for (int i = 1; i <= ratio; ++i) {
        double binom = 1.0, p = no;             // C(i-1, 0), no^(k+1)
        for (int k = 0; k < i; ++k) {
            denom += binom * ((k & 1) ? -1.0 : 1.0) * p / sqrt((double)(k + 1));
            binom = binom * (double)(i - 1 - k) / (double)(k + 1);
            p *= no;
        }
    }
    *log_coeff = (float)log(od / denom);
    *new_o = fminf(fmaxf((float)no, min_opacity), 1.0f - 1.1920929e-7f);
}
struct McmcSets { float* p[6]; float* m[6]; float* v[6]; };
// destination j <- copy of src[j] with the relocated opacity/scale; its optimizer moments are cleared
__global__ void __launch_bounds__(MB)
k_mcmc_copy(int n, int k_fixed, const uint32_t* __restrict__ k_dev, const uint32_t* __restrict__ dead_list, int dst_base,
            const uint32_t* __restrict__ src, const uint32_t* __restrict__ count, float min_opacity, int layout, McmcSets S) {
    const int j = blockIdx.x * MB + threadIdx.x;
    const int k = k_dev ? (int)*k_dev : k_fixed;
    if (j >= k) return;
    const uint32_t s = src[j];
    if (s == 0xffffffffu) return;
    const int d = dead_list ? (int)dead_list[j] : dst_base + j;
    float no, lc;
    m_relocation(m_sigmoid(S.p[3][s]), (int)count[s] + 1, min_opacity, &no, &lc);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S.p[0][3 * (int64_t)d + c] = S.p[0][3 * (int64_t)s + c];
        S.p[1][3 * (int64_t)d + c] = S.p[1][3 * (int64_t)s + c];
        S.p[4][3 * (int64_t)d + c] = S.p[4][3 * (int64_t)s + c] + lc;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) S.p[5][4 * (int64_t)d + c] = S.p[5][4 * (int64_t)s + c];
    S.p[3][d] = __logf(no / (1.0f - no));
    for (int e = 0; e < 45; ++e) S.p[2][m_shn_index(layout, d, e)] = S.p[2][m_shn_index(layout, (int)s, e)];
    for (int g = 0; g < 6; ++g) {
        const int w = g == 2 ? 45 : (g == 3 ? 1 : (g == 5 ? 4 : 3));
        for (int e = 0; e < w; ++e) {
            const int64_t at = g == 2 ? m_shn_index(layout, d, e) : (int64_t)d * w + e;
            if (S.m[g]) S.m[g][at] = 0.f;
            if (S.v[g]) S.v[g][at] = 0.f;
        }
    }
}
// every splat that was drawn takes the same relocated opacity/scale; its moments are cleared; the draw counters are reset
__global__ void __launch_bounds__(MB)
k_mcmc_update_src(int n, uint32_t* __restrict__ count, float min_opacity, int layout, McmcSets S) {
    const int i = blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = count[i];
    if (c == 0) return;
    count[i] = 0;
    float no, lc;
    m_relocation(m_sigmoid(S.p[3][i]), (int)c + 1, min_opacity, &no, &lc);
    S.p[3][i] = __logf(no / (1.0f - no));
#pragma unroll
    for (int k = 0; k < 3; ++k) S.p[4][3 * (int64_t)i + k] += lc;
    for (int g = 0; g < 6; ++g) {
        const int w = g == 2 ? 45 : (g == 3 ? 1 : (g == 5 ? 4 : 3));
        for (int e = 0; e < w; ++e) {
            const int64_t at = g == 2 ? m_shn_index(layout, i, e) : (int64_t)i * w + e;
            if (S.m[g]) S.m[g][at] = 0.f;
            if (S.v[g]) S.v[g][at] = 0.f;
        }
    }
}

__global__ void __launch_bounds__(MB)
k_mcmc_noise(int i0, int n /*the splats [i0, n)*/, float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ rot,
             const float* __restrict__ opacity, float lr, uint32_t seed) {
    const int i = i0 + blockIdx.x * MB + threadIdx.x;              // (the random numbers are a function of the GLOBAL index: a range launch draws the same)
    if (i >= n) return;
    const float o = m_sigmoid(opacity[i]);
    const float gate = 1.0f / (1.0f + __expf(100.0f * (o - 0.005f)));          // ~1 for nearly dead splats, ~0 for solid ones
    float q[4], s2[3], z[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = rot[4 * (int64_t)i + k];
    const float qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(qn > 0.f)) return;
    float R[9];
    dvs_quat_to_rot(q[0] / qn, q[1] / qn, q[2] / qn, q[3] / qn, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = __expf(scale[3 * (int64_t)i + k]);
        s2[k] = e * e;
        const float u1 = m_uniform(m_hash(seed, (uint32_t)i, 2u * k)), u2 = m_uniform(m_hash(seed, (uint32_t)i, 2u * k + 1u));
        z[k] = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2) * gate * lr;
    }
    // Sigma z = R diag(s^2) R^T z
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = (R[k] * z[0] + R[3 + k] * z[1] + R[6 + k] * z[2]) * s2[k];        // diag(s^2) R^T z
#pragma unroll
    for (int r = 0; r < 3; ++r) pos[3 * (int64_t)i + r] += R[r * 3] * t[0] + R[r * 3 + 1] * t[1] + R[r * 3 + 2] * t[2];
}

__global__ void __launch_bounds__(MB)
k_mcmc_regularize(int i0, int n /*the splats [i0, n)*/, const float* __restrict__ opacity, const float* __restrict__ scale, float* __restrict__ g_opacity,
                  float* __restrict__ g_scale, float wo, float ws) {
    const int i = i0 + blockIdx.x * MB + threadIdx.x;
    if (i >= n) return;
    const float o = m_sigmoid(opacity[i]);
    g_opacity[i] += wo * o * (1.0f - o);                              // d/dlogit of wo * sigmoid(logit)
#pragma unroll
    for (int k = 0; k < 3; ++k) g_scale[3 * (int64_t)i + k] += ws * __expf(scale[3 * (int64_t)i + k]);   // d/dlog s of ws * exp(log s)
}

static int run_draws(hipStream_t st, int n, int k_fixed, bool relocate, int dst_base, const dvs_mcmc_sets* sets, float min_opacity,
                     uint32_t seed, int layout, void* scratch, size_t cap) {
    McmcScratch S = carve(scratch, cap);
    McmcSets D;
    for (int g = 0; g < 6; ++g) { D.p[g] = sets->param[g]; D.m[g] = sets->m[g]; D.v[g] = sets->v[g]; }
    const uint32_t nb = (uint32_t)((n + MB - 1) / MB);
    hipLaunchKernelGGL(k_mcmc_blocksum, dim3(nb), dim3(MB), 0, st, n, D.p[3], min_opacity, S.blk_w, S.blk_d);
    hipLaunchKernelGGL(k_mcmc_scan_blocks, dim3(1), dim3(MB), 0, st, nb, S.blk_w, S.blk_d, S.total_w, S.n_dead);
    hipLaunchKernelGGL(k_mcmc_cdf, dim3(nb), dim3(MB), 0, st, n, D.p[3], min_opacity, S.blk_w, S.blk_d, S.cdf, S.dead_list);
    const int kmax = relocate ? n : k_fixed;                         // relocate: the dead count is only known on the device
    const uint32_t kb = (uint32_t)((kmax + MB - 1) / MB);
    if (kb == 0) return DVS_OK;
    hipLaunchKernelGGL(k_mcmc_sample, dim3(kb), dim3(MB), 0, st, n, k_fixed, relocate ? S.n_dead : nullptr, S.cdf, S.total_w, seed, S.src, S.count);
    hipLaunchKernelGGL(k_mcmc_copy, dim3(kb), dim3(MB), 0, st, n, k_fixed, relocate ? S.n_dead : nullptr, relocate ? S.dead_list : nullptr,
                       dst_base, S.src, S.count, min_opacity, layout, D);
    hipLaunchKernelGGL(k_mcmc_update_src, dim3(nb), dim3(MB), 0, st, n, S.count, min_opacity, layout, D);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}

extern "C" {
size_t dvs_mcmc_scratch_bytes(int capacity) { return capacity > 0 ? scratch_bytes((size_t)capacity) : 0; }

int dvs_mcmc_init_scratch(void* stream, void* scratch, int capacity) {
    if (!scratch || capacity <= 0) return DVS_ERR_INVALID;
    return hipMemsetAsync(scratch, 0, scratch_bytes((size_t)capacity), (hipStream_t)stream) == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}

int dvs_mcmc_relocate(void* stream, int n, const dvs_mcmc_sets* sets, float min_opacity, uint32_t seed, int shn_layout, void* scratch,
                      int capacity, uint32_t* n_dead_out) {
    if (n < 0 || !sets || !scratch || capacity < n) return DVS_ERR_INVALID;
    for (int g = 0; g < 6; ++g) if (n > 0 && !sets->param[g]) return DVS_ERR_INVALID;
    if (n == 0) return DVS_OK;
    const int r = run_draws((hipStream_t)stream, n, 0, true, 0, sets, min_opacity, seed, shn_layout, scratch, (size_t)capacity);
    if (r == DVS_OK && n_dead_out)
        (void)hipMemcpyAsync(n_dead_out, carve(scratch, (size_t)capacity).n_dead, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return r;
}

int dvs_mcmc_grow(void* stream, int n, int n_new, const dvs_mcmc_sets* sets, float min_opacity, uint32_t seed, int shn_layout, void* scratch,
                  int capacity) {
    if (n < 0 || n_new < 0 || !sets || !scratch || (int64_t)n + n_new > capacity) return DVS_ERR_INVALID;
    for (int g = 0; g < 6; ++g) if (n > 0 && !sets->param[g]) return DVS_ERR_INVALID;
    if (n == 0 || n_new == 0) return DVS_OK;
    return run_draws((hipStream_t)stream, n, n_new, false, n, sets, min_opacity, seed, shn_layout, scratch, (size_t)capacity);
}

int dvs_mcmc_add_noise_range(void* stream, int n, int first, int count, float* pos, const float* scale, const float* rot, const float* opacity, float lr,
                             uint32_t seed) {
    if (n < 0 || first < 0 || count < 0 || first + count > n || (n > 0 && (!pos || !scale || !rot || !opacity))) return DVS_ERR_INVALID;
    if (count == 0 || lr == 0.f) return DVS_OK;
    hipLaunchKernelGGL(k_mcmc_noise, dim3((count + MB - 1) / MB), dim3(MB), 0, (hipStream_t)stream, first, first + count, pos, scale, rot, opacity, lr, seed);
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_mcmc_add_noise(void* stream, int n, float* pos, const float* scale, const float* rot, const float* opacity, float lr, uint32_t seed) {
    return dvs_mcmc_add_noise_range(stream, n, 0, n, pos, scale, rot, opacity, lr, seed);
}

int dvs_mcmc_regularize_range(void* stream, int n, int first, int count, const float* opacity, const float* scale, float* g_opacity, float* g_scale,
                              float opacity_reg, float scale_reg) {
    if (n < 0 || first < 0 || count < 0 || first + count > n || (n > 0 && (!opacity || !scale || !g_opacity || !g_scale))) return DVS_ERR_INVALID;
    if (count == 0) return DVS_OK;
    hipLaunchKernelGGL(k_mcmc_regularize, dim3((count + MB - 1) / MB), dim3(MB), 0, (hipStream_t)stream, first, first + count, opacity, scale, g_opacity, g_scale,
                       opacity_reg / (float)n, scale_reg / (3.0f * (float)n));
    return hipGetLastError() == hipSuccess ? DVS_OK : DVS_ERR_HIP;
}
int dvs_mcmc_regularize(void* stream, int n, const float* opacity, const float* scale, float* g_opacity, float* g_scale, float opacity_reg,
                        float scale_reg) {
    return dvs_mcmc_regularize_range(stream, n, 0, n, opacity, scale, g_opacity, g_scale, opacity_reg, scale_reg);
}
}
