// binning.hip — A3 tile-offset scan, A4 duplicate-with-keys, A5 radix sort, A6 tile ranges (gfx950).
//
// The (tileID|depth) order is produced by an LSD radix sort whose low 32 key bits (the depth) are
// sorted BEFORE duplication: every instance of a splat carries the same depth bits, so the four
// depth passes run over the N splats instead of the T instances, instances are then emitted in
// depth order, and only the tile-id bits (ceil(log2 tiles) of them) are sorted over T.  The result
// is bit-identical to a stable sort of ((tile << 32) | depth_bits) over the instance list emitted in
// splat order (proof in DESIGN.md §4), at about a third of the HBM traffic.
//
// Reference anchors: key idea gsplat_viewz_cs.hlsl:250-253, sortable float gaussian_common.hlsl:115-120,
// the viewer's own 8-bit-digit LSD sort renderer/gpu_sort.cpp:16-25,54-91 (32-bit keys, Vulkan; not reused).
//
// Wave64 idioms: in the sweep, digits are ranked with 8 ballots + mbcnt (a stable 64-wide multisplit) and per-wave digit counters in
// LDS; the global histograms, which need no ranks, count with native integer LDS atomics (ds_add_u32).
#include <cstdlib>
#include "dvs_device.h"
#include "dvs_kernels.h"

#define SORT_BLOCK 256
#define SORT_WAVES (SORT_BLOCK / 64)
// keys per workgroup = SORT_BLOCK * ITEMS; ITEMS = 8 (small inputs: more workgroups) or 16 (large inputs: longer runs per digit)
#define RADIX 256

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// peers = lanes of this wave holding the same 8-bit digit (invalid lanes never match valid ones)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return valid ? peers : 0ull;
}

// block-wide exclusive scan of one uint32 per thread (256 threads); tmp = LDS[SORT_WAVES+1]
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) { const uint32_t t = tmp[w]; if ((uint32_t)w < wave) wbase += t; tot += t; }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}

// ---- A5: one LSD pass = histogram, row scan, scatter -----------------------------------------------
// wave w of block b owns the contiguous items [b*PART + w*64*ITEMS, +64*ITEMS), read in ITEMS rounds of 64.
template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_hist(const uint32_t* __restrict__ keys, uint64_t n_host, const uint64_t* __restrict__ n_dev, int shift, uint32_t dmask,
            uint32_t* __restrict__ hist, uint32_t num_blocks) {
    __shared__ uint32_t cnt[SORT_WAVES][RADIX];
    // device-side count (no host round trip for T): `num_blocks` partitions cover the capacity n_host, the grid may be smaller (sized
    // for the expected count) — every workgroup strides over the partitions; partitions beyond n only publish zero counts
    const uint64_t n = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    constexpr int SORT_PART = SORT_BLOCK * SORT_ITEMS;
    for (uint32_t part = blockIdx.x; part < num_blocks; part += gridDim.x) {
        if ((uint64_t)part * SORT_PART >= n) { hist[(uint64_t)threadIdx.x * num_blocks + part] = 0u; continue; }
        for (int e = threadIdx.x; e < SORT_WAVES * RADIX; e += SORT_BLOCK) (&cnt[0][0])[e] = 0;
        __syncthreads();
        const uint64_t wbase = (uint64_t)part * SORT_PART + (uint64_t)wave * (64 * SORT_ITEMS);
        // all of the lane's keys are requested before the first one is used: with one load per round the kernel is bound by
        // SORT_ITEMS dependent HBM round trips at 3 waves per SIMD
        uint32_t kreg[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            kreg[r] = idx < n ? keys[idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            const bool valid = idx < n;
            const uint32_t d = valid ? ((kreg[r] >> shift) & dmask) : 0u;
            // integer LDS atomics are native (unlike ds_add_f32): one ds_add_u32 per key into the wave's own counters
            if (valid) atomicAdd(&cnt[wave][d], 1u);
        }
        __syncthreads();
        const uint32_t d = threadIdx.x;
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) s += cnt[w][d];
        hist[(uint64_t)d * num_blocks + part] = s;
        __syncthreads();
    }
}

// one workgroup (16 waves) per digit: exclusive scan of its row of per-block counts, row total -> totals[d]. Wave w owns a contiguous
// segment of the row and reads it in rows of 64 consecutive counts (coalesced); with 1024 threads a wave has 6 such rows at 24 M keys.
#define ROWSCAN_THREADS 1024
__global__ void __launch_bounds__(ROWSCAN_THREADS)
k_sort_rowscan(uint32_t* __restrict__ hist, uint32_t num_blocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[ROWSCAN_THREADS / 64];
    uint32_t* row = hist + (uint64_t)blockIdx.x * num_blocks;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t rows_per_wave = (num_blocks + ROWSCAN_THREADS - 1) / ROWSCAN_THREADS;
    const uint32_t seg = wave * rows_per_wave * 64u;
    uint32_t s = 0;
    for (uint32_t r = 0; r < rows_per_wave; ++r) {
        const uint32_t i = seg + r * 64u + lane;
        s += i < num_blocks ? row[i] : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    uint32_t carry = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ROWSCAN_THREADS / 64; ++w) { const uint32_t t = wsum[w]; if ((uint32_t)w < wave) carry += t; tot += t; }
    for (uint32_t r = 0; r < rows_per_wave; ++r) {
        const uint32_t i = seg + r * 64u + lane;
        const uint32_t v = i < num_blocks ? row[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
        if (i < num_blocks) row[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

// Scatter: rank the partition's 4096 keys (stable, wave64 multisplit), order them by digit in LDS, then write them out
// slot by slot: consecutive lanes hold consecutive keys of the same digit, so the global stores are runs of full
// cache lines instead of 64 scattered dwords per instruction.
template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
               uint32_t* __restrict__ vals_out, uint64_t n_host, const uint64_t* __restrict__ n_dev, int shift, uint32_t dmask,
               const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, uint32_t num_blocks) {
    __shared__ uint32_t cnt[SORT_WAVES][RADIX];     // per-wave digit counts, then per-wave local bases
    __shared__ uint32_t gdelta[RADIX];              // global destination of LDS slot s with digit d = gdelta[d] + s
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    constexpr int SORT_PART = SORT_BLOCK * SORT_ITEMS;
    __shared__ uint32_t stage_k[SORT_PART];
    __shared__ uint32_t stage_v[SORT_PART];
    const uint64_t n = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t part = blockIdx.x; (uint64_t)part * SORT_PART < n; part += gridDim.x) {       // (uniform per workgroup)
        for (int e = threadIdx.x; e < SORT_WAVES * RADIX; e += SORT_BLOCK) (&cnt[0][0])[e] = 0;
        __syncthreads();
        const uint64_t pbase = (uint64_t)part * SORT_PART;
        const uint64_t wbase = pbase + (uint64_t)wave * (64 * SORT_ITEMS);
        uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            const bool valid = idx < n;
            key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
            val[r] = valid ? vals_in[idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            const bool valid = idx < n;
            const uint32_t d = (key[r] >> shift) & dmask;
            const uint64_t peers = match_digit(d, valid);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            uint32_t prev = 0;
            if (valid) {
                prev = cnt[wave][d];                                   // in-order LDS: all peers read before the leader writes
                if (below == 0) cnt[wave][d] = prev + (uint32_t)__popcll(peers);
            }
            rank[r] = prev + below;
        }
        __syncthreads();
        {
            const uint32_t d = threadIdx.x;
            uint32_t c[SORT_WAVES], bc = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) { c[w] = cnt[w][d]; bc += c[w]; }
            uint32_t tot;
            const uint32_t loff = block_excl_scan(bc, tmp, &tot);                 // where digit d starts in the LDS stage
            const uint32_t digit_excl = block_excl_scan(totals[d], tmp, &tot);    // where digit d starts globally
            gdelta[d] = digit_excl + hist[(uint64_t)d * num_blocks + part] - loff;
            uint32_t run = loff;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) { cnt[w][d] = run; run += c[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            if (idx < n) {
                const uint32_t pos = cnt[wave][(key[r] >> shift) & dmask] + rank[r];
                stage_k[pos] = key[r];
                stage_v[pos] = val[r];
            }
        }
        __syncthreads();
        const uint32_t nvalid = (uint32_t)((n - pbase) < (uint64_t)SORT_PART ? (n - pbase) : (uint64_t)SORT_PART);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const uint32_t slot = (uint32_t)i * SORT_BLOCK + threadIdx.x;
            if (slot < nvalid) {
                const uint32_t k = stage_k[slot];
                const uint32_t dst = gdelta[(k >> shift) & dmask] + slot;
                keys_out[dst] = k;
                vals_out[dst] = stage_v[slot];
            }
        }
        __syncthreads();
    }
}

static inline int sort_items_for(uint64_t n) { return n <= 1500000ull ? 8 : 16; }

static hipError_t launch_sort_pass(hipStream_t st, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                uint32_t* vals_out, uint64_t n, int shift, int bits, uint32_t* scratch, const uint64_t* n_dev,
                                uint64_t n_expected) {
    if (n == 0) return hipSuccess;
    const uint64_t n_grid = (n_dev && n_expected > 0 && n_expected < n) ? n_expected : n;      // the kernels stride over the partitions
    const int items = sort_items_for(n_grid);
    const uint32_t nb = (uint32_t)((n + (uint64_t)SORT_BLOCK * items - 1) / ((uint64_t)SORT_BLOCK * items));
    uint32_t ng = (uint32_t)((n_grid + (uint64_t)SORT_BLOCK * items - 1) / ((uint64_t)SORT_BLOCK * items));
    if (ng > nb) ng = nb;
    uint32_t* hist = scratch;
    uint32_t* totals = scratch + (size_t)nb * RADIX;
    const uint32_t dmask = bits >= 8 ? 0xFFu : ((1u << bits) - 1u);
    if (items == 8) hipLaunchKernelGGL(k_sort_hist<8>, dim3(ng), dim3(SORT_BLOCK), 0, st, keys_in, n, n_dev, shift, dmask, hist, nb);
    else hipLaunchKernelGGL(k_sort_hist<16>, dim3(ng), dim3(SORT_BLOCK), 0, st, keys_in, n, n_dev, shift, dmask, hist, nb);
    hipLaunchKernelGGL(k_sort_rowscan, dim3(RADIX), dim3(ROWSCAN_THREADS), 0, st, hist, nb, totals);
    if (items == 8)
        hipLaunchKernelGGL(k_sort_scatter<8>, dim3(ng), dim3(SORT_BLOCK), 0, st, keys_in, vals_in, keys_out, vals_out, n, n_dev, shift,
                           dmask, hist, totals, nb);
    else
        hipLaunchKernelGGL(k_sort_scatter<16>, dim3(ng), dim3(SORT_BLOCK), 0, st, keys_in, vals_in, keys_out, vals_out, n, n_dev, shift,
                           dmask, hist, totals, nb);
    return hipGetLastError();
}

// ---- A5, measured alternative (DVS_SORT_ONESWEEP=1): one "sweep" kernel per 8-bit pass with chained-scan partition prefixes ---------
// The default above runs three kernels per pass (per-partition histogram, row scan, scatter: the keys are read twice per pass, 18
// launches per forward). SURVEY.md §8 A5 names the alternative ("onesweep"), built here and measured SLOWER on this chip (C3, per
// 8-view step 5.97 vs 5.77 ms; single view: depth sort 0.124 vs 0.079 ms, tile sort 0.102 vs 0.065 ms): with ~1000 workgroups resident,
// all of them publish their counts at about the same time, so a partition looks back over hundreds of "count only" predecessors, one
// dependent cross-XCD load (~1 us) per step and digit thread — the chain the reduce-then-scan form replaces by one 7-us row scan.
// A wave-parallel look-back (64 predecessors per step from a [digit][partition] status layout) would cost about as many vector
// instructions per partition as the ranking itself. Kept selectable, bit-exact like the default (tests/test_gpu_parity.py::test_sort_pairs).
//   k_sort_ghist   ONE read of the keys per SORT: the global digit histograms of all its passes (they do not depend on the order
//                  the earlier passes leave the keys in)
//   k_sort_sweep   one launch per pass: a workgroup takes the next partition (a global ticket, so partitions start in order), ranks
//                  its keys (stable wave64 multisplit, as before), publishes its per-digit counts in a status word, LOOKS BACK over
//                  its predecessors' status words until it meets an inclusive prefix, publishes its own inclusive prefix, and
//                  scatters — keys and values are read once and written once per pass.
// A status word carries its value and its state in ONE 32-bit word (bits 31..30: 0 not ready / 1 this partition's count / 2 inclusive
// prefix up to and including it), written and polled with relaxed agent-scope atomics: the per-XCD L2s are not coherent with each
// other, but a single word written with an agent-scope store is seen whole by an agent-scope load on any XCD, and nothing else has to
// be ordered against it (the scattered keys are read by the NEXT kernel). A partition only ever waits for partitions with smaller
// tickets, whose workgroups already run: no deadlock by construction; a poll budget turns a broken chain into a reported error
// (the instance-overflow counter: "outputs invalid") instead of a hang.
#define OS_FLAG_AGG 1u
#define OS_FLAG_INC 2u
#define OS_VALUE_MASK 0x3FFFFFFFu
#define OS_MAX_PASSES 4
#define OS_POLL_BUDGET (1u << 22)

__device__ __forceinline__ uint32_t os_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct OsPasses { int npass; int shift[OS_MAX_PASSES]; uint32_t dmask[OS_MAX_PASSES]; };

// global digit histograms of all passes: ghist[pass][256]
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_ghist(const uint32_t* __restrict__ keys, uint64_t n_host, const uint64_t* __restrict__ n_dev, OsPasses ps, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[OS_MAX_PASSES][RADIX];
    const uint64_t n = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    for (int e = threadIdx.x; e < OS_MAX_PASSES * RADIX; e += SORT_BLOCK) (&h[0][0])[e] = 0;
    __syncthreads();
    // 16 keys per thread and round: four 16-B loads in flight
    const uint64_t per_round = (uint64_t)SORT_BLOCK * 16;
    for (uint64_t base = (uint64_t)blockIdx.x * per_round; base < n; base += (uint64_t)gridDim.x * per_round) {
        uint4 k4[4];
        bool full[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t i0 = base + ((uint64_t)r * SORT_BLOCK + threadIdx.x) * 4;
            full[r] = i0 + 4 <= n;
            if (full[r]) k4[r] = reinterpret_cast<const uint4*>(keys)[i0 >> 2];
            else {
                k4[r].x = i0 < n ? keys[i0] : 0u; k4[r].y = i0 + 1 < n ? keys[i0 + 1] : 0u;
                k4[r].z = i0 + 2 < n ? keys[i0 + 2] : 0u; k4[r].w = 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t i0 = base + ((uint64_t)r * SORT_BLOCK + threadIdx.x) * 4;
            const uint32_t kk[4] = {k4[r].x, k4[r].y, k4[r].z, k4[r].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool valid = i0 + u < n;
                for (int p = 0; p < ps.npass; ++p) {
                    // the high bytes of a depth or tile key take few values: when the whole wave holds one digit, one lane adds the
                    // count (64 same-address LDS atomics would serialise)
                    const uint32_t d = (kk[u] >> ps.shift[p]) & ps.dmask[p];
                    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                    const uint64_t vm = __ballot(valid);
                    if (__ballot(valid && d != d0) == 0ull) {
                        if (vm != 0ull && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(vm)) atomicAdd(&h[p][valid ? d : d0], (uint32_t)__popcll(vm));
                    } else if (valid) {
                        atomicAdd(&h[p][d], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ps.npass * RADIX; e += SORT_BLOCK) {
        const uint32_t v = (&h[0][0])[e];
        if (v) atomicAdd(&ghist[e], v);
    }
}

template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_sweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
             uint32_t* __restrict__ vals_out, uint64_t n_host, const uint64_t* __restrict__ n_dev, int shift, uint32_t dmask,
             const uint32_t* __restrict__ ghist /*[256] of this pass*/, uint32_t* __restrict__ status /*[parts][256] of this pass, zeroed*/,
             uint32_t* __restrict__ ticket /*zeroed*/, unsigned long long* __restrict__ err /*+1 when the look-back ran out of polls*/) {
    __shared__ uint32_t cnt[SORT_WAVES][RADIX];     // per-wave digit counts, then per-wave local bases
    __shared__ uint32_t gdelta[RADIX];              // global destination of LDS slot s with digit d = gdelta[d] + s
    __shared__ uint32_t gbase[RADIX];               // where digit d starts in the output (exclusive scan of the global histogram)
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    __shared__ uint32_t s_part;
    constexpr int SORT_PART = SORT_BLOCK * SORT_ITEMS;
    __shared__ uint32_t stage_k[SORT_PART];
    __shared__ uint32_t stage_v[SORT_PART];
    const uint64_t n = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    {
        uint32_t tot;
        gbase[threadIdx.x] = block_excl_scan(ghist[threadIdx.x], tmp, &tot);
    }
    for (;;) {
        __syncthreads();                                 // (previous partition fully written; gbase visible)
        if (threadIdx.x == 0) s_part = atomicAdd(ticket, 1u);
        for (int e = threadIdx.x; e < SORT_WAVES * RADIX; e += SORT_BLOCK) (&cnt[0][0])[e] = 0;
        __syncthreads();
        const uint32_t part = s_part;
        const uint64_t pbase = (uint64_t)part * SORT_PART;
        if (pbase >= n) break;                           // (uniform per workgroup)
        const uint64_t wbase = pbase + (uint64_t)wave * (64 * SORT_ITEMS);
        uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            const bool valid = idx < n;
            key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
            val[r] = valid ? vals_in[idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            const bool valid = idx < n;
            const uint32_t d = (key[r] >> shift) & dmask;
            const uint64_t peers = match_digit(d, valid);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            uint32_t prev = 0;
            if (valid) {
                prev = cnt[wave][d];                                   // in-order LDS: all peers read before the leader writes
                if (below == 0) cnt[wave][d] = prev + (uint32_t)__popcll(peers);
            }
            rank[r] = prev + below;
        }
        __syncthreads();
        {
            const uint32_t d = threadIdx.x;
            uint32_t c[SORT_WAVES], bc = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) { c[w] = cnt[w][d]; bc += c[w]; }
            // chained scan over the partitions, one digit per thread: publish this partition's count, look back, publish the prefix
            uint32_t* const mine = status + (uint64_t)part * RADIX + d;
            uint32_t excl = 0;
            if (part == 0) {
                os_store(mine, (OS_FLAG_INC << 30) | bc);
            } else {
                os_store(mine, (OS_FLAG_AGG << 30) | bc);
                uint32_t polls = 0;
                for (int64_t q = (int64_t)part - 1; q >= 0;) {
                    const uint32_t sw = os_load(status + (uint64_t)q * RADIX + d);
                    const uint32_t flag = sw >> 30;
                    if (flag == 0u) {
                        if (++polls > OS_POLL_BUDGET) { atomicAdd(err, 1ull); break; }
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    excl += sw & OS_VALUE_MASK;
                    if (flag == OS_FLAG_INC) break;
                    --q;
                }
                os_store(mine, (OS_FLAG_INC << 30) | ((excl + bc) & OS_VALUE_MASK));
            }
            uint32_t tot;
            const uint32_t loff = block_excl_scan(bc, tmp, &tot);                 // where digit d starts in the LDS stage
            gdelta[d] = gbase[d] + excl - loff;
            uint32_t run = loff;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) { cnt[w][d] = run; run += c[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const uint64_t idx = wbase + (uint64_t)r * 64 + lane;
            if (idx < n) {
                const uint32_t pos = cnt[wave][(key[r] >> shift) & dmask] + rank[r];
                stage_k[pos] = key[r];
                stage_v[pos] = val[r];
            }
        }
        __syncthreads();
        const uint32_t nvalid = (uint32_t)((n - pbase) < (uint64_t)SORT_PART ? (n - pbase) : (uint64_t)SORT_PART);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const uint32_t slot = (uint32_t)i * SORT_BLOCK + threadIdx.x;
            if (slot < nvalid) {
                const uint32_t k = stage_k[slot];
                const uint32_t dst = gdelta[(k >> shift) & dmask] + slot;
                keys_out[dst] = k;
                vals_out[dst] = stage_v[slot];
            }
        }
    }
}

// scratch words of one sort of up to n items: [global histograms: 4 x 256][tickets: 64][status: 4 passes x partitions x 256]
static inline uint64_t sort_parts(uint64_t n) { return (n + SORT_BLOCK * 8 - 1) / (SORT_BLOCK * 8); }      // sized for the smaller partition
size_t dvs_sort_scratch_words(uint64_t n) { return (size_t)(OS_MAX_PASSES * RADIX + 64 + OS_MAX_PASSES * sort_parts(n) * RADIX); }   // (covers the default's nb * 256 + 256)

// Stable LSD sort of (key, value) pairs over key bits [bit_lo, bit_hi): buffers 0 hold the input, the result is in buffers
// (*result_in & 1). n sizes the scratch use and the grids; n_dev (nullable) is the device-side count, n_expected a grid hint for it.
hipError_t dvs_launch_sort(hipStream_t st, uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, uint64_t n, int bit_lo, int bit_hi,
                           uint32_t* scratch, const uint64_t* n_dev, uint64_t n_expected, unsigned long long* err_counter, int* result_in) {
    if (result_in) *result_in = 0;
    if (n == 0 || bit_hi <= bit_lo) return hipSuccess;
    static const bool onesweep = [] { const char* e = getenv("DVS_SORT_ONESWEEP"); return e && e[0] == '1'; }();
    // The opt-in chained-scan form packs (flag, 30-bit prefix) into its status words: an inclusive prefix of 2^30 or more items in one
    // digit bucket would spill into the flag bits, so sorts that large always take the default path.
    if (!onesweep || n >= (1ull << 30)) {                // default: histogram + row scan + scatter per pass
        uint32_t* kk[2] = {keys0, keys1};
        uint32_t* vv[2] = {vals0, vals1};
        int c = 0;
        for (int shift = bit_lo; shift < bit_hi; shift += 8) {
            hipError_t e = launch_sort_pass(st, kk[c], vv[c], kk[c ^ 1], vv[c ^ 1], n, shift, bit_hi - shift, scratch, n_dev, n_expected);
            if (e != hipSuccess) return e;
            c ^= 1;
        }
        if (result_in) *result_in = c;
        return hipSuccess;
    }
    OsPasses ps{};
    for (int shift = bit_lo; shift < bit_hi && ps.npass < OS_MAX_PASSES; shift += 8) {
        const int bits = bit_hi - shift;
        ps.shift[ps.npass] = shift; ps.dmask[ps.npass] = bits >= 8 ? 0xFFu : ((1u << bits) - 1u);
        ++ps.npass;
    }
    const uint64_t n_grid = (n_dev && n_expected > 0 && n_expected < n) ? n_expected : n;
    const int items = sort_items_for(n_grid);
    const uint64_t part = (uint64_t)SORT_BLOCK * items;
    const uint64_t nparts = (n + part - 1) / part;                               // status rows the kernels may touch (capacity)
    uint32_t* ghist = scratch;
    uint32_t* tickets = scratch + OS_MAX_PASSES * RADIX;
    uint32_t* status = tickets + 64;
    const size_t zero_words = (size_t)OS_MAX_PASSES * RADIX + 64 + (size_t)ps.npass * nparts * RADIX;
    hipError_t e = hipMemsetAsync(scratch, 0, zero_words * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    uint64_t gparts = (n_grid + part - 1) / part;
    uint32_t hgrid = (uint32_t)((n_grid + (uint64_t)SORT_BLOCK * 16 - 1) / ((uint64_t)SORT_BLOCK * 16));
    if (hgrid > 512u) hgrid = 512u;
    if (hgrid < 1u) hgrid = 1u;
    hipLaunchKernelGGL(k_sort_ghist, dim3(hgrid), dim3(SORT_BLOCK), 0, st, keys0, n, n_dev, ps, ghist);
    uint32_t sgrid = (uint32_t)(gparts + 1 < 4096 ? gparts + 1 : 4096);          // workgroups draw partitions until none are left
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    int cur = 0;
    for (int p = 0; p < ps.npass; ++p) {
        uint32_t* st_p = status + (size_t)p * nparts * RADIX;
        if (items == 8)
            hipLaunchKernelGGL(k_sort_sweep<8>, dim3(sgrid), dim3(SORT_BLOCK), 0, st, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], n, n_dev, ps.shift[p],
                               ps.dmask[p], ghist + p * RADIX, st_p, tickets + p, err_counter);
        else
            hipLaunchKernelGGL(k_sort_sweep<16>, dim3(sgrid), dim3(SORT_BLOCK), 0, st, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], n, n_dev, ps.shift[p],
                               ps.dmask[p], ghist + p * RADIX, st_p, tickets + p, err_counter);
        cur ^= 1;
    }
    if (result_in) *result_in = cur;
    return hipGetLastError();
}

// ---- A3: scan of tiles_touched in depth-sorted order ------------------------------------------------
// k_preprocess_fwd leaves each splat's tile rectangle as four u16 (8 B; an empty rectangle for culled splats). This kernel does the ONE
// random gather of the binning stage — rect[sorted_ids[j]] — and re-emits the rectangles in depth order, so that the scan and the
// duplication stream (round 1 gathered tiles_touched twice and the 64-B projected record once: 3.7x / 4.5x the algorithmic bytes).
__device__ __forceinline__ uint32_t rect_tiles(uint2 r) {
    return ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu));
}
// DVS_TILES_TIGHT: a splat's instances = the set bits of its 64-bit tile mask (all ones: the whole rectangle — rectangles of more than 64
// tiles are never tightened)
__device__ __forceinline__ uint32_t rect16_tiles(uint4 r) {
    const uint32_t both = r.z & r.w;
    return both == 0xFFFFFFFFu ? rect_tiles(make_uint2(r.x, r.y)) : (uint32_t)(__popc(r.z) + __popc(r.w));
}
template <bool TIGHT>
__global__ void __launch_bounds__(SORT_BLOCK)
k_tile_blocksum(int n, const uint32_t* __restrict__ sorted_ids, const uint2* __restrict__ rect, uint2* __restrict__ rect_sorted,
                uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    const int j = blockIdx.x * SORT_BLOCK + threadIdx.x;
    uint32_t v = 0;
    if (j < n) {
        if (TIGHT) {
            const uint4 r = reinterpret_cast<const uint4*>(rect)[sorted_ids[j]];
            reinterpret_cast<uint4*>(rect_sorted)[j] = r;
            v = rect16_tiles(r);
        } else {
            const uint2 r = rect[sorted_ids[j]];
            rect_sorted[j] = r;
            v = rect_tiles(r);
        }
    }
    uint32_t tot;
    (void)block_excl_scan(v, tmp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

#define SCANB_THREADS 1024
__global__ void __launch_bounds__(SCANB_THREADS)
k_tile_scan_blocks(uint32_t* __restrict__ block_sums, uint32_t num_blocks, uint64_t* __restrict__ total /*[0] = T, [1] += (T > capacity)*/,
                   uint64_t capacity) {
    // One workgroup of 16 waves (a multi-view batch has 31 k block sums). Wave w owns a contiguous segment, read in rows of 64
    // consecutive values (coalesced): first the segment totals, scanned over the waves, then every row is scanned across the lanes
    // and rewritten as exclusive offsets with a running carry. (Round 2a gave each THREAD a contiguous chunk: stride-31 accesses, 47 us.)
    __shared__ unsigned long long wsum[SCANB_THREADS / 64];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t rows_per_wave = (num_blocks + SCANB_THREADS - 1) / SCANB_THREADS;      // rows of 64 per wave
    const uint32_t seg = wave * rows_per_wave * 64u;
    unsigned long long s64 = 0ull;
    for (uint32_t r = 0; r < rows_per_wave; ++r) {
        const uint32_t i = seg + r * 64u + lane;
        s64 += i < num_blocks ? block_sums[i] : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s64 += __shfl_xor(s64, d, 64);
    if (lane == 0) wsum[wave] = s64;
    __syncthreads();
    unsigned long long base64 = 0ull, wide = 0ull;        // the true 64-bit total: T >= 2^32 or > capacity is an error (offsets are 32-bit)
#pragma unroll
    for (int w = 0; w < SCANB_THREADS / 64; ++w) { const unsigned long long t = wsum[w]; if ((uint32_t)w < wave) base64 += t; wide += t; }
    uint32_t carry = (uint32_t)base64;
    for (uint32_t r = 0; r < rows_per_wave; ++r) {
        const uint32_t i = seg + r * 64u + lane;
        const uint32_t v = i < num_blocks ? block_sums[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += o; }
        if (i < num_blocks) block_sums[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) { total[0] = wide; if (wide > capacity) total[1] += 1ull; }
}

size_t dvs_scan_scratch_words(int n) { return (size_t)((n + SORT_BLOCK - 1) / SORT_BLOCK) + 1; }

hipError_t dvs_launch_tile_scan(hipStream_t st, int n, const uint32_t* sorted_ids, const uint32_t* rect, uint32_t* rect_sorted,
                                uint32_t* block_offsets, uint64_t* total_dev, uint64_t capacity, int tight) {
    const uint32_t nb = (uint32_t)((n + SORT_BLOCK - 1) / SORT_BLOCK);
    if (nb > 0 && tight) hipLaunchKernelGGL(k_tile_blocksum<true>, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect, (uint2*)rect_sorted, block_offsets);
    else if (nb > 0) hipLaunchKernelGGL(k_tile_blocksum<false>, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect, (uint2*)rect_sorted, block_offsets);
    hipLaunchKernelGGL(k_tile_scan_blocks, dim3(1), dim3(SCANB_THREADS), 0, st, block_offsets, nb, total_dev, capacity);
    return hipGetLastError();
}

// ---- A4: duplicate with keys, in depth-sorted order ---------------------------------------------------
// Streams (splat id, rectangle) in depth order; instances beyond `capacity` are not written (k_tile_scan_blocks has raised the
// overflow counter; the host reports DVS_ERR_CAPACITY — never a silent truncation).
// Every wave emits the instances of its own 64 splats cooperatively: output slot k of the wave finds its splat by a binary search over
// the 64 exclusive offsets (LDS) and its tile from the slot's index inside the splat's rectangle (row-major), so that a store
// instruction covers 64 consecutive instances whatever the rectangle sizes are. (Rounds 1-2 let the owning lane loop over a small
// rectangle: the 64 lanes then wrote 64 different runs per instruction, 1.9 TB/s of stores.)
// position of the (t + 1)-th set bit of a 64-bit mask (t < popcount): the tile index of instance t of a tightened rectangle
__device__ __forceinline__ uint32_t nth_set_bit64(uint32_t lo, uint32_t hi, uint32_t t) {
    uint32_t w = lo, base = 0;
    const uint32_t c = (uint32_t)__popc(lo);
    if (t >= c) { t -= c; w = hi; base = 32u; }
#pragma unroll
    for (uint32_t h = 16u; h >= 1u; h >>= 1) {
        const uint32_t cl = (uint32_t)__popc(w & ((1u << h) - 1u));
        if (t >= cl) { t -= cl; w >>= h; base += h; }
    }
    return base;
}
template <bool TIGHT>
__global__ void __launch_bounds__(SORT_BLOCK)
k_duplicate(int n, const uint32_t* __restrict__ sorted_ids, const uint2* __restrict__ rect_sorted,
            const uint32_t* __restrict__ block_offsets, int tiles_x, uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_splat,
            uint64_t capacity, int n_per_view, int n_views, int tiles_per_view) {
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    __shared__ uint32_t s_pre[SORT_BLOCK + 1];      // exclusive offset of the thread's first instance inside the block; [SORT_BLOCK] = block total
    __shared__ uint32_t s_id[SORT_BLOCK];
    __shared__ uint32_t s_tile[SORT_BLOCK];         // tile id of the rectangle's first tile (view offset included)
    __shared__ uint32_t s_w[SORT_BLOCK];            // rectangle width in tiles
    __shared__ uint2 s_mask[TIGHT ? SORT_BLOCK : 1]; // DVS_TILES_TIGHT: the splat's tile mask
    const int j = blockIdx.x * SORT_BLOCK + threadIdx.x;
    uint32_t id = 0, touched = 0;
    uint2 r = make_uint2(0u, 0u);
    if (TIGHT) {
        uint4 r4 = make_uint4(0u, 0u, 0u, 0u);
        if (j < n) { id = sorted_ids[j]; r4 = reinterpret_cast<const uint4*>(rect_sorted)[j]; touched = rect16_tiles(r4); }
        r = make_uint2(r4.x, r4.y);
        s_mask[threadIdx.x] = make_uint2(r4.z, r4.w);
    } else if (j < n) { id = sorted_ids[j]; r = rect_sorted[j]; touched = rect_tiles(r); }
    // multi-view batch: the sort value is the global index view * n_per_view + splat; the tile ids of view v start at v * tiles_per_view
    uint32_t view = 0;
    for (int k = 1; k < n_views; ++k) view += (id >= (uint32_t)k * (uint32_t)n_per_view) ? 1u : 0u;
    const uint32_t minx = r.x & 0xFFFFu, miny = r.y & 0xFFFFu, maxx = r.x >> 16;
    uint32_t tot;
    const uint32_t pre = block_excl_scan(touched, tmp, &tot);
    s_pre[threadIdx.x] = pre;
    s_id[threadIdx.x] = id;
    s_tile[threadIdx.x] = view * (uint32_t)tiles_per_view + miny * (uint32_t)tiles_x + minx;
    s_w[threadIdx.x] = maxx - minx;
    if (threadIdx.x == 0) s_pre[SORT_BLOCK] = tot;
    __syncthreads();
    const uint32_t lane = lane_id(), w0 = (threadIdx.x >> 6) * 64u;
    const uint32_t k_end = s_pre[w0 + 64u];                  // (the next wave's first offset, or the block total)
    const uint64_t gbase = block_offsets[blockIdx.x];
    for (uint32_t k = s_pre[w0] + lane; k < k_end; k += 64u) {
        uint32_t lo = 0;                                      // largest i in [0, 64) with s_pre[w0 + i] <= k: its range is not empty and holds k
#pragma unroll
        for (uint32_t step = 32u; step >= 1u; step >>= 1)
            if (s_pre[w0 + lo + step] <= k) lo += step;
        const uint32_t src = w0 + lo, w = s_w[src];
        uint32_t t = k - s_pre[src];
        if (TIGHT) {                                          // instance t of the splat = its (t + 1)-th surviving tile
            const uint2 m = s_mask[src];
            if ((m.x & m.y) != 0xFFFFFFFFu) t = nth_set_bit64(m.x, m.y, t);
        }
        // row = t / w by a float quotient and one correction step (t < 2^24: a rectangle has fewer tiles than the screen)
        uint32_t q = (uint32_t)((float)t * __builtin_amdgcn_rcpf((float)w));
        int32_t rem = (int32_t)(t - q * w);
        if (rem < 0) { --q; rem += (int32_t)w; } else if (rem >= (int32_t)w) { ++q; rem -= (int32_t)w; }
        const uint64_t g = gbase + k;
        if (g < capacity) {
            inst_tile[g] = s_tile[src] + q * (uint32_t)tiles_x + (uint32_t)rem;
            inst_splat[g] = s_id[src];
        }
    }
}

hipError_t dvs_launch_duplicate(hipStream_t st, int n, const uint32_t* sorted_ids, const uint32_t* rect_sorted,
                                const uint32_t* block_offsets, int tiles_x, uint32_t* inst_tile, uint32_t* inst_splat, uint64_t capacity,
                                int n_per_view, int n_views, int tiles_per_view, int tight) {
    const uint32_t nb = (uint32_t)((n + SORT_BLOCK - 1) / SORT_BLOCK);
    if (nb == 0) return hipSuccess;
    if (tight) hipLaunchKernelGGL(k_duplicate<true>, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect_sorted, block_offsets,
                                  tiles_x, inst_tile, inst_splat, capacity, n_per_view, n_views, tiles_per_view);
    else hipLaunchKernelGGL(k_duplicate<false>, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect_sorted, block_offsets,
                            tiles_x, inst_tile, inst_splat, capacity, n_per_view, n_views, tiles_per_view);
    return hipGetLastError();
}

// ---- A6: tile ranges -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SORT_BLOCK)
k_tile_ranges(uint64_t T_host, const uint64_t* __restrict__ T_dev, const uint32_t* __restrict__ sorted_tile, uint2* __restrict__ ranges) {
    const uint64_t T = T_dev ? (*T_dev < T_host ? *T_dev : T_host) : T_host;
    // four consecutive instances per thread (one 16-B load + the two neighbours)
    for (uint64_t q = (uint64_t)blockIdx.x * SORT_BLOCK + threadIdx.x; q * 4 < T; q += (uint64_t)gridDim.x * SORT_BLOCK) {
        const uint64_t j0 = q * 4;
        uint32_t t[6];                                  // t[0] = element j0 - 1, t[1..4] = j0 .. j0 + 3, t[5] = j0 + 4
        if (j0 + 4 <= T) {
            const uint4 v = reinterpret_cast<const uint4*>(sorted_tile)[q];
            t[1] = v.x; t[2] = v.y; t[3] = v.z; t[4] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[1 + k] = j0 + k < T ? sorted_tile[j0 + k] : 0xFFFFFFFFu;
        }
        t[0] = j0 > 0 ? sorted_tile[j0 - 1] : 0xFFFFFFFFu;
        t[5] = j0 + 4 < T ? sorted_tile[j0 + 4] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t j = j0 + k;
            if (j < T) {
                if (j == 0 || t[k] != t[k + 1]) ranges[t[k + 1]].x = (uint32_t)j;
                if (j + 1 == T || t[k + 2] != t[k + 1]) ranges[t[k + 1]].y = (uint32_t)(j + 1);
            }
        }
    }
}

hipError_t dvs_launch_tile_ranges(hipStream_t st, uint64_t T, const uint32_t* sorted_tile, uint32_t* ranges, int tiles, const uint64_t* T_dev,
                                  uint64_t T_expected, bool clear) {
    hipError_t e = clear ? hipMemsetAsync(ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t), st) : hipSuccess;
    if (e != hipSuccess) return e;
    if (T == 0) return hipSuccess;
    const uint64_t T_grid = (T_dev && T_expected > 0 && T_expected < T) ? T_expected : T;
    const uint32_t nb = (uint32_t)((T_grid + 4 * SORT_BLOCK - 1) / (4 * SORT_BLOCK));
    hipLaunchKernelGGL(k_tile_ranges, dim3(nb), dim3(SORT_BLOCK), 0, st, T, T_dev, sorted_tile, (uint2*)ranges);
    return hipGetLastError();
}

// ---- parity export: canonical 64-bit keys -------------------------------------------------------------------
__global__ void __launch_bounds__(SORT_BLOCK)
k_export_keys(uint64_t T, const uint32_t* __restrict__ sorted_tile, const uint32_t* __restrict__ sorted_splat,
              const float* __restrict__ depth, uint64_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (j >= T) return;
    out[j] = ((uint64_t)sorted_tile[j] << 32) | (uint64_t)__float_as_uint(depth[sorted_splat[j]]);
}

hipError_t dvs_launch_export_keys(hipStream_t st, uint64_t T, const uint32_t* sorted_tile, const uint32_t* sorted_splat,
                                  const float* depth, uint64_t* out_keys) {
    if (T == 0) return hipSuccess;
    const uint32_t nb = (uint32_t)((T + SORT_BLOCK - 1) / SORT_BLOCK);
    hipLaunchKernelGGL(k_export_keys, dim3(nb), dim3(SORT_BLOCK), 0, st, T, sorted_tile, sorted_splat, depth, out_keys);
    return hipGetLastError();
}
