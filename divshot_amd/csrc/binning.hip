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
// Wave64 idioms: in the scatter, digits are ranked with 8 ballots + mbcnt (a stable 64-wide multisplit) and per-wave digit
// counters in LDS; the histogram pass, which needs no ranks, counts with native integer LDS atomics (ds_add_u32).
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

size_t dvs_sort_scratch_words(uint64_t n) {
    const uint64_t nb = (n + SORT_BLOCK * 8 - 1) / (SORT_BLOCK * 8);      // sized for the smaller partition
    return (size_t)(nb * RADIX + RADIX);
}

hipError_t dvs_launch_sort_pass(hipStream_t st, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
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

// ---- A3: scan of tiles_touched in depth-sorted order ------------------------------------------------
// k_preprocess_fwd leaves each splat's tile rectangle as four u16 (8 B; an empty rectangle for culled splats). This kernel does the ONE
// random gather of the binning stage — rect[sorted_ids[j]] — and re-emits the rectangles in depth order, so that the scan and the
// duplication stream (round 1 gathered tiles_touched twice and the 64-B projected record once: 3.7x / 4.5x the algorithmic bytes).
__device__ __forceinline__ uint32_t rect_tiles(uint2 r) {
    return ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu));
}
__global__ void __launch_bounds__(SORT_BLOCK)
k_tile_blocksum(int n, const uint32_t* __restrict__ sorted_ids, const uint2* __restrict__ rect, uint2* __restrict__ rect_sorted,
                uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    const int j = blockIdx.x * SORT_BLOCK + threadIdx.x;
    uint32_t v = 0;
    if (j < n) {
        const uint2 r = rect[sorted_ids[j]];
        rect_sorted[j] = r;
        v = rect_tiles(r);
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
                                uint32_t* block_offsets, uint64_t* total_dev, uint64_t capacity) {
    const uint32_t nb = (uint32_t)((n + SORT_BLOCK - 1) / SORT_BLOCK);
    if (nb > 0) hipLaunchKernelGGL(k_tile_blocksum, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect, (uint2*)rect_sorted, block_offsets);
    hipLaunchKernelGGL(k_tile_scan_blocks, dim3(1), dim3(SCANB_THREADS), 0, st, block_offsets, nb, total_dev, capacity);
    return hipGetLastError();
}

// ---- A4: duplicate with keys, in depth-sorted order ---------------------------------------------------
// Streams (splat id, rectangle) in depth order; instances beyond `capacity` are not written (k_tile_scan_blocks has raised the
// overflow counter; the host reports DVS_ERR_CAPACITY — never a silent truncation).
#define DUP_COOP_THRESHOLD 16
__global__ void __launch_bounds__(SORT_BLOCK)
k_duplicate(int n, const uint32_t* __restrict__ sorted_ids, const uint2* __restrict__ rect_sorted,
            const uint32_t* __restrict__ block_offsets, int tiles_x, uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_splat,
            uint64_t capacity, int n_per_view, int n_views, int tiles_per_view) {
    __shared__ uint32_t tmp[SORT_WAVES + 1];
    const int j = blockIdx.x * SORT_BLOCK + threadIdx.x;
    uint32_t id = 0, touched = 0;
    uint2 r = make_uint2(0u, 0u);
    if (j < n) { id = sorted_ids[j]; r = rect_sorted[j]; touched = rect_tiles(r); }
    // multi-view batch: the sort value is the global index view * n_per_view + splat; the tile ids of view v start at v * tiles_per_view
    uint32_t view = 0;
    for (int k = 1; k < n_views; ++k) view += (id >= (uint32_t)k * (uint32_t)n_per_view) ? 1u : 0u;
    const uint32_t tile0 = view * (uint32_t)tiles_per_view;
    uint32_t tot;
    uint32_t off = block_excl_scan(touched, tmp, &tot) + block_offsets[blockIdx.x];
    const int minx = (int)(r.x & 0xFFFFu), miny = (int)(r.y & 0xFFFFu), maxx = (int)(r.x >> 16), maxy = (int)(r.y >> 16);
    const int w = maxx - minx;
    // small rects: the owning lane emits its tiles (row-major inside the rect)
    if (touched > 0 && touched <= DUP_COOP_THRESHOLD) {
        for (int y = miny; y < maxy; ++y)
            for (int x = minx; x < maxx; ++x) {
                if (off < capacity) {
                    inst_tile[off] = tile0 + (uint32_t)(y * tiles_x + x);
                    inst_splat[off] = id;
                }
                ++off;
            }
    }
    // large rects: the whole wave emits one splat's tiles, 64 per step
    uint64_t big = __ballot(touched > DUP_COOP_THRESHOLD);
    const uint32_t lane = lane_id();
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t b_id = __shfl(id, src, 64), b_touched = __shfl(touched, src, 64), b_off = __shfl(off, src, 64);
        const int b_minx = __shfl(minx, src, 64), b_miny = __shfl(miny, src, 64), b_w = __shfl(w, src, 64);
        const uint32_t b_tile0 = __shfl(tile0, src, 64);
        for (uint32_t k = lane; k < b_touched; k += 64) {
            const int y = b_miny + (int)(k / (uint32_t)b_w), x = b_minx + (int)(k % (uint32_t)b_w);
            if ((uint64_t)b_off + k < capacity) {
                inst_tile[b_off + k] = b_tile0 + (uint32_t)(y * tiles_x + x);
                inst_splat[b_off + k] = b_id;
            }
        }
    }
}

hipError_t dvs_launch_duplicate(hipStream_t st, int n, const uint32_t* sorted_ids, const uint32_t* rect_sorted,
                                const uint32_t* block_offsets, int tiles_x, uint32_t* inst_tile, uint32_t* inst_splat, uint64_t capacity,
                                int n_per_view, int n_views, int tiles_per_view) {
    const uint32_t nb = (uint32_t)((n + SORT_BLOCK - 1) / SORT_BLOCK);
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_duplicate, dim3(nb), dim3(SORT_BLOCK), 0, st, n, sorted_ids, (const uint2*)rect_sorted, block_offsets,
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
                                  uint64_t T_expected) {
    hipError_t e = hipMemsetAsync(ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t), st);
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
