// frontend.hip — the binning stage of round 5: A5 depth sort, A3 tile-offset scan, A4 duplicate-with-keys, A5 tile sort, A6 tile
// ranges, SEGMENTED BY VIEW and laid out for the eight XCDs of an MI355X (gfx950).
//
// What is sorted is unchanged: the canonical per-view order is the stable sort of (tile << 32 | depth bits) over the
// instance list emitted splat-major; its low 32 bits (the depth) are sorted BEFORE duplication over the splats, the tile bits after
// it over the instances. What changed against rounds 1-4 (binning.hip, one sort over the (view, splat) elements of the whole batch):
//
//  * every view is its own SEGMENT: its splats are sorted among themselves, its instances form one contiguous run of the instance
//    arrays, its tile ids need ceil(log2 tiles) bits instead of ceil(log2 (views * tiles)). Workgroup b works for view b % n_views, and
//    since workgroup b runs on XCD b % 8 (observed placement, used for speed only) a view's keys, histogram rows and scattered output
//    stay inside ONE XCD's 4-MiB L2 when the batch has 8 (or 2, 4, 16) views: the gather of the tile rectangles hits lines its own XCD
//    fetched, the partial lines of a scatter merge in one L2 instead of meeting in memory.
//  * the depth sort is THREE passes whatever the scene: A2 leaves each view's smallest and largest depth key, a pass digit is
//    b = ceil(bits(max - min) / 3) bits of (key - min) (b = 9 for a depth ratio up to 2^16; 11 covers every positive float), chosen on
//    the device per view — no host round trip. Up to 9 bits a partition's keys are re-ordered by digit in LDS so that the global stores
//    are runs (as before); wider digits (far/near > 65536) scatter straight from registers.
//  * culled splats (key 0xFFFFFFFF) leave in the first pass: it neither counts nor scatters them, the later passes, A3 and A4 run over
//    the visible splats only; the first pass also makes up its values (the splat index) instead of reading an id array.
//  * A3 sums the tile counts of 256 splats per workgroup and adds the block sums into one 64-bit counter per 65536 splats; a 16-wave
//    kernel turns those few counters into the views' instance ranges (T stays on the device), A4 finds its output offset from them and
//    at most 255 block sums. The 31 k-element scan kernel of rounds 2-4 is gone.
//
// Reference anchors: key idea gsplat_viewz_cs.hlsl:250-253, sortable float gaussian_common.hlsl:115-120, the viewer's own 8-bit-digit
// LSD sort renderer/gpu_sort.cpp:16-25,54-91 / gpu_sort/sort_common.hlsl:2-17 (32-bit keys, Vulkan; not reused).
#include <cstdlib>
#include "dvs_device.h"
#include "dvs_kernels.h"

#define FE_BLOCK 256
#define FE_WAVES (FE_BLOCK / 64)
#define FE_MAXBINS 2048           // 11-bit digits: 3 x 11 >= 31 bits, every positive float's range
#define FE_REORDER_BITS 9         // digits up to this width are re-ordered in LDS before the global stores
#define FE_CULLED 0xFFFFFFFFu

__device__ __forceinline__ uint32_t fe_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// peers = lanes of this wave holding the same b-bit digit (invalid lanes never match valid ones). Per bit: the lane's bit as a 0 / ~0 word
// (v_bfe_i32), its ballot, and ONE v_bitop3_b32 per mask half — peers & ~(ballot ^ word) keeps the lanes whose bit equals this lane's —
// four vector instructions (the select-based form of rounds 2-5b took eight; the scatters are ~65 % vector-ALU-busy).
__device__ __forceinline__ uint64_t fe_match(uint32_t d, bool valid, uint32_t b) {
    const uint64_t v0 = __ballot(valid);
    uint32_t lo = (uint32_t)v0, hi = (uint32_t)(v0 >> 32);
    for (uint32_t k = 0; k < b; ++k) {
        const uint32_t m = (uint32_t)(((int32_t)(d << (31u - k))) >> 31);
        const uint64_t bal = __ballot(m != 0u);
        lo = lo & ~((uint32_t)bal ^ m);
        hi = hi & ~((uint32_t)(bal >> 32) ^ m);
    }
    const uint64_t peers = ((uint64_t)hi << 32) | lo;
    return valid ? peers : 0ull;
}

// Wave64 inclusive prefix sum in six DPP additions (gfx9 row operations: shift right by 1, 2, 4, 8 inside the 16-lane rows, then lane 15
// of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 / 3). No LDS crossbar round trip per step (a __shfl_up step is a ds_bpermute:
// ~60 cycles of latency on a dependent chain).
__device__ __forceinline__ uint32_t fe_wave_incl_scan(uint32_t v) {
#define FE_DPP_ADD(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false)
    FE_DPP_ADD(0x111, 0xF); FE_DPP_ADD(0x112, 0xF); FE_DPP_ADD(0x114, 0xF); FE_DPP_ADD(0x118, 0xF);   // row_shr:1, 2, 4, 8
    FE_DPP_ADD(0x142, 0xA);                                                                              // row_bcast:15 -> rows 1, 3
    FE_DPP_ADD(0x143, 0xC);                                                                              // row_bcast:31 -> rows 2, 3
#undef FE_DPP_ADD
    return v;
}
__device__ __forceinline__ uint32_t fe_wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)fe_wave_incl_scan(v), 63); }

// block-wide exclusive scan of one uint32 per thread (256 threads); tmp = LDS[FE_WAVES]
__device__ __forceinline__ uint32_t fe_block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6;
    const uint32_t inc = fe_wave_incl_scan(v);
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FE_WAVES; ++w) { const uint32_t t = tmp[w]; if ((uint32_t)w < wave) wbase += t; tot += t; }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}

// ---- segmented LSD pass = per-partition digit histogram, row scan, scatter --------------------------------------------------------
// A pass is described by (adaptive, pass, shift, bits): adaptive = the depth sort (digit width and key offset per view from the
// segment descriptor, shift = pass * width), otherwise the given shift / width (tile sort, dvs_sort_pairs_u32).
// Workgroup b: view b % V, then the partitions w, w + G, ... of that view (w = b / V, G = gridDim / V): the grids may be smaller than
// the partition count (the tile sort's instance count lives on the device).
// Histogram table: hist[digit][row], row = seg.pstart + partition; rows of different views never overlap.

template <int ITEMS>
__global__ void __launch_bounds__(FE_BLOCK)
k_seg_hist(const uint32_t* __restrict__ keys, DvsSeg* __restrict__ seg, int V, int adaptive, int pass, int shift_s, int bits_s, int cull,
           uint32_t* __restrict__ hist, uint32_t nbtot, const uint32_t* __restrict__ kred, int key16 /*the keys are 16-bit (A4's tile ids)*/) {
    __shared__ uint32_t h[FE_MAXBINS];
    constexpr uint32_t PART = FE_BLOCK * ITEMS;
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6, tid = threadIdx.x;
    const uint32_t view = blockIdx.x % (uint32_t)V, w = blockIdx.x / (uint32_t)V, gv = gridDim.x / (uint32_t)V;
    const uint32_t s_base = seg[view].base, s_count = seg[view].count, s_pstart = seg[view].pstart;
    uint32_t b = (uint32_t)bits_s, shift = (uint32_t)shift_s, sub = 0u;
    if (adaptive) {
        if (pass == 0) {
            // the view's key range: A2 left min / max in 64 slots (as max(~key), max(key): zero-initialised); every wave reduces them
            uint32_t mn = ~kred[((size_t)view * 64 + lane) * 16], mx = kred[((size_t)view * 64 + lane) * 16 + 1];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
                mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
            }
            sub = mn; b = 1u;
            if (mx >= mn) { const uint32_t range = mx - mn; const uint32_t rb = range ? 32u - (uint32_t)__builtin_clz(range) : 0u; b = (rb + 2u) / 3u; if (b < 1u) b = 1u; }
            else sub = 0u;                                         // no visible splat in this view
            if (w == 0 && tid == 0) { seg[view].sub = sub; seg[view].bits = b; }     // (the later kernels of the sort read them here)
        } else { sub = seg[view].sub; b = seg[view].bits; }
        shift = (uint32_t)pass * b;
    }
    const uint32_t nbins = 1u << b, dmask = nbins - 1u;
    const uint32_t nparts = (s_count + PART - 1) / PART;
    for (uint32_t p = w; p < nparts; p += gv) {
        for (uint32_t e = tid; e < nbins; e += FE_BLOCK) h[e] = 0u;
        __syncthreads();
        const uint32_t wb = p * PART + wave * (64 * ITEMS);
        uint32_t kreg[ITEMS];
        const uint32_t* const kin = keys + s_base;
        const uint16_t* const kin16 = reinterpret_cast<const uint16_t*>(keys) + s_base;
        const uint32_t i0 = wb + lane, ilast = s_count - 1u;
        if (key16) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) { const uint32_t idx = i0 + (uint32_t)r * 64; kreg[r] = kin16[idx < ilast ? idx : ilast]; }
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) { const uint32_t idx = i0 + (uint32_t)r * 64; kreg[r] = kin[idx < ilast ? idx : ilast]; }
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const bool valid = i0 + (uint32_t)r * 64 <= ilast && !(cull && kreg[r] == FE_CULLED);
            if (valid) atomicAdd(&h[((kreg[r] - sub) >> shift) & dmask], 1u);       // native integer LDS atomic (ds_add_u32)
        }
        __syncthreads();
        const size_t row = (size_t)s_pstart + p;
        for (uint32_t d = tid; d < nbins; d += FE_BLOCK) hist[(size_t)d * nbtot + row] = h[d];
        __syncthreads();
    }
}

// one wave per (digit, view): exclusive scan of that view's slice of the digit's row, slice total -> totals[view][digit]. The wave
// requests up to 16 chunks of 64 counts at once (one memory latency per 1024 partitions, not one per chunk) and scans them with DPP.
__global__ void __launch_bounds__(FE_BLOCK)
k_seg_rowscan(uint32_t* __restrict__ hist, uint32_t nbtot, const DvsSeg* __restrict__ seg, int V, int adaptive, int bits_s, uint32_t part,
              uint32_t* __restrict__ totals) {
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6;
    const uint32_t view = blockIdx.x % (uint32_t)V, d = (blockIdx.x / (uint32_t)V) * FE_WAVES + wave;
    const uint32_t b = adaptive ? seg[view].bits : (uint32_t)bits_s;
    if (d >= (1u << b)) return;
    const uint32_t nparts = (seg[view].count + part - 1) / part;
    uint32_t* row = hist + (size_t)d * nbtot + seg[view].pstart;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nparts; c0 += 1024) {
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t i = c0 + (uint32_t)k * 64 + lane; v[k] = i < nparts ? row[i] : 0u; }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (c0 + (uint32_t)k * 64 < nparts) {                  // (uniform)
                const uint32_t i = c0 + (uint32_t)k * 64 + lane;
                const uint32_t inc = fe_wave_incl_scan(v[k]);
                if (i < nparts) row[i] = carry + inc - v[k];
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
    }
    if (lane == 0) totals[(size_t)view * FE_MAXBINS + d] = carry;
}

// Scatter. Ranks are stable: within a wave by a ballot multisplit (fe_match + mbcnt), across the rounds of a wave by per-wave digit
// counters in LDS (a wave's LDS operations execute in order: all peers read the counter before their leader rewrites it), across the
// waves by the counters' prefix, across the partitions by the scanned histogram row.
// Round 6, RANK_ATOMIC: the ablations of profiles/r06_scatter_ablation.txt put the tile scatter's 93 us at 23 (loads) + 55 (ranking,
// scans, LDS staging) + 15 (stores): the multisplit's ~40 vector instructions per key and round were the largest single part. One
// returning LDS add per key (ds_add_rtn_u32 on the key's per-wave counter) yields the same rank — old value = keys of the digit in
// earlier rounds + lower lanes of this round — PROVIDED lanes that hit one address are served in ascending lane order. The ISA manual
// does not promise that, so it is not assumed: dvs_fe_probe_rank_atomic() runs both rankings on adversarial digit patterns on the
// device at context creation and the atomic form is used only when every rank agrees (DVS_FE_RANK=ballot forces the multisplit;
// the parity suite — bit-exact against std::stable_sort — runs both). Tile scatter 93 -> 75 us, depth scatter 39 -> 29 us per pass.
//   digits of <= 9 bits ("reorder"): the partition's keys are written to LDS in digit order first, then stored slot by slot —
//       consecutive lanes hold consecutive keys of one digit, so a store instruction covers a few runs instead of 64 scattered dwords;
//   wider digits: straight from the registers (a run would be two keys long).
// LDS (words): reorder  [cnt16: 4 x 512 u16 = 1024][gdelta 512][stage_k PART][stage_v PART][tmp 8]
//              wide     [cnt16: 4 x 2048 u16 = 4096][gdelta 2048]                             [tmp 8]
template <int ITEMS> struct FeScatterLds {
    static constexpr int PART = FE_BLOCK * ITEMS;
    static constexpr int REORDER_WORDS = 1024 + 512 + 2 * PART;
    static constexpr int WIDE_WORDS = 4096 + 2048 + 2048;          // cnt16, gdelta, digit bases
    static constexpr int WORDS = (REORDER_WORDS > WIDE_WORDS ? REORDER_WORDS : WIDE_WORDS) + 8;
};

template <int ITEMS, bool RANK_ATOMIC /*rank by returning LDS adds instead of the ballot multisplit (round 6)*/>
__global__ void __launch_bounds__(FE_BLOCK) __attribute__((amdgpu_waves_per_eu(4)))      // (<= 128 VGPRs: the LDS allows four workgroups per CU)
k_seg_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in /*null: the value is the element's index in its segment*/,
              uint32_t* __restrict__ keys_out /*null: the keys are not needed any more*/, uint32_t* __restrict__ vals_out,
              const DvsSeg* __restrict__ seg_in, DvsSeg* __restrict__ seg_out /*null, or where the segments of the OUTPUT are published
              (same base / pstart / sub / bits, count = the elements that survived the culling)*/,
              int V, int adaptive, int pass, int shift_s, int bits_s, int cull, uint32_t key_add_per_view,
              const uint32_t* __restrict__ hist, uint32_t nbtot, const uint32_t* __restrict__ totals,
              int key16 /*keys_in holds 16-bit keys (the tile ids A4 wrote: 2 B per instance instead of 4 through A4, the histogram and this pass)*/,
              uint32_t* __restrict__ ranges_enc /*null, or (last pass of the tile sort: A6 fused) the tile ranges [V * tiles][2], zeroed: every run of equal
              keys in a partition's output raises word 0 to ~(first position) and word 1 to (last position + 1) with atomic max — a tile's runs
              from all partitions leave (~start, end); k_render_fwd turns that into (start, end)*/) {
    constexpr uint32_t PART = FE_BLOCK * ITEMS;
    __shared__ __attribute__((aligned(16))) uint32_t lds[FeScatterLds<ITEMS>::WORDS];
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6, tid = threadIdx.x;
    const uint32_t view = blockIdx.x % (uint32_t)V, w = blockIdx.x / (uint32_t)V, gv = gridDim.x / (uint32_t)V;
    const DvsSeg S = seg_in[view];
    const uint32_t b = adaptive ? S.bits : (uint32_t)bits_s;
    const uint32_t shift = adaptive ? (uint32_t)pass * b : (uint32_t)shift_s;
    const uint32_t sub = adaptive ? S.sub : 0u;
    const uint32_t nbins = 1u << b, dmask = nbins - 1u;
    const bool wide = b > FE_REORDER_BITS;
    const uint32_t cstride = wide ? 2048u : 512u;
    uint16_t* const cnt16 = reinterpret_cast<uint16_t*>(lds);
    uint32_t* const gdelta = lds + (wide ? 4096 : 1024);
    uint32_t* const dexl = lds + 6144;                            // wide: where each digit starts in the view's output
    uint32_t* const stage_k = lds + 1536;
    uint32_t* const stage_v = stage_k + PART;
    uint32_t* const tmp = lds + (FeScatterLds<ITEMS>::WORDS - 8);
    const uint32_t key_add = key_add_per_view * view;
    const uint32_t* const vtot = totals + (size_t)view * FE_MAXBINS;
    // uniform segment bases + 32-bit lane offsets: scalar base / vector offset addressing, no 64-bit address per load
    const uint32_t* const kin = keys_in + S.base;
    const uint16_t* const kin16 = reinterpret_cast<const uint16_t*>(keys_in) + S.base;
    const uint32_t* const vin = vals_in ? vals_in + S.base : nullptr;
    uint32_t* const kout = keys_out ? keys_out + S.base : nullptr;
    uint32_t* const vout = vals_out + S.base;

    const uint32_t nparts = (S.count + PART - 1) / PART;
    const uint32_t ilast = S.count - 1u;
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    // Small sorts (ITEMS = 8: fewer workgroups than the chip holds, every kernel one latency chain) request the keys of the workgroup's
    // first partition BEFORE the digit bases are scanned: one memory round trip less in front of the ranking. Large sorts (ITEMS = 16)
    // request them at the top of the loop: held across the scan they would cost the fourth wave per SIMD.
    // Unconditional loads from clamped indices (a partition is never empty): no branch per load; lanes past the end are masked later.
    constexpr bool EARLY = ITEMS <= 8;
    auto request = [&](uint32_t p) {
        const uint32_t i0 = p * PART + wave * (64 * ITEMS) + lane;
        if (key16) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) { const uint32_t idx = i0 + (uint32_t)r * 64; key[r] = kin16[idx < ilast ? idx : ilast]; }
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) { const uint32_t idx = i0 + (uint32_t)r * 64; key[r] = kin[idx < ilast ? idx : ilast]; }
        }
        if (vin) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) { const uint32_t idx = i0 + (uint32_t)r * 64; val[r] = vin[idx < ilast ? idx : ilast]; }
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) val[r] = i0 + (uint32_t)r * 64;
        }
    };
    if (EARLY && w < nparts) request(w);

    // where each digit starts in the view's output: exclusive scan of the view's digit totals. Reorder: thread t owns bins 2t, 2t + 1 and
    // keeps their bases in registers; wide: bins 8t .. 8t + 7, bases in LDS.
    uint32_t dex0 = 0, dex1 = 0;
    {
        uint32_t s = 0;
        if (!wide) {
            const uint32_t t0 = 2u * tid < nbins ? vtot[2u * tid] : 0u, t1 = 2u * tid + 1u < nbins ? vtot[2u * tid + 1u] : 0u;
            dex1 = t0; s = t0 + t1;
        } else {
#pragma unroll 1
            for (uint32_t k = 0; k < 8; ++k) { const uint32_t d = tid * 8u + k; s += d < nbins ? vtot[d] : 0u; }
        }
        uint32_t tot;
        const uint32_t off = fe_block_excl_scan(s, tmp, &tot);
        if (!wide) { dex0 = off; dex1 += off; }
        else {
            uint32_t run = off;
#pragma unroll 1
            for (uint32_t k = 0; k < 8; ++k) { const uint32_t d = tid * 8u + k; dexl[d] = run; run += d < nbins ? vtot[d] : 0u; }
        }
        if (seg_out && w == 0 && tid == 0) { seg_out[view].base = S.base; seg_out[view].count = tot; seg_out[view].pstart = S.pstart; seg_out[view].sub = S.sub; seg_out[view].bits = S.bits; }
    }
    for (uint32_t p = w; p < nparts; p += gv) {
        const size_t row = (size_t)S.pstart + p;
        __syncthreads();                                                                 // (dexl written; the previous partition's stores read their LDS)
        for (uint32_t e = tid; e < cstride * 2u; e += FE_BLOCK) lds[e] = 0u;            // the four waves' u16 counters
        uint32_t hp0 = 0, hp1 = 0;                                                       // reorder: this partition's prefixes of the thread's two bins
        if (!wide) {
            if (2u * tid < nbins) hp0 = hist[(size_t)(2u * tid) * nbtot + row];
            if (2u * tid + 1u < nbins) hp1 = hist[(size_t)(2u * tid + 1u) * nbtot + row];
        }
        if (!EARLY) request(p);
        __syncthreads();
        const uint32_t i0 = p * PART + wave * (64 * ITEMS) + lane;
        uint32_t vmask = 0u;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const bool valid = i0 + (uint32_t)r * 64 <= ilast && !(cull && key[r] == FE_CULLED);
            const uint32_t d = ((key[r] - sub) >> shift) & dmask;
            uint32_t prev = 0;
            if constexpr (RANK_ATOMIC) {
                // rank inside the wave = what the returning LDS add hands back: lanes that add to one address are served in lane order
                // (checked against the ballot ranking by dvs_fe_probe_rank_atomic before a context selects this path), rounds in
                // program order, so the old value IS (earlier rounds' count) + (lower lanes with the same digit). Two u16 counters
                // share a word: a wave's counter stays below 64 * ITEMS, so the low half never carries into the high one.
                if (valid) {
                    const uint32_t old = atomicAdd(&lds[wave * (cstride >> 1) + (d >> 1)], (d & 1u) ? 0x10000u : 1u);
                    prev = (d & 1u) ? (old >> 16) : (old & 0xFFFFu);
                    vmask |= 1u << r;
                }
                rank[r] = prev;
            } else {
                const uint64_t peers = fe_match(d, valid, b);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
                if (valid) {
                    prev = cnt16[wave * cstride + d];                       // in-order LDS: all peers read before the leader writes
                    if (below == 0) cnt16[wave * cstride + d] = (uint16_t)(prev + (uint32_t)__popcll(peers));
                    vmask |= 1u << r;
                }
                rank[r] = prev + below;
            }
            asm volatile("" : "+v"(rank[r]));          // materialise the sum here: otherwise prev, below and the counter's address stay live per key
        }
        __syncthreads();
        if (!wide) {
            uint32_t c0[FE_WAVES], c1[FE_WAVES], bc0 = 0, bc1 = 0;
#pragma unroll
            for (int wv = 0; wv < FE_WAVES; ++wv) {
                const uint32_t two = *reinterpret_cast<const uint32_t*>(&cnt16[(uint32_t)wv * 512u + tid * 2u]);
                c0[wv] = two & 0xFFFFu; c1[wv] = two >> 16; bc0 += c0[wv]; bc1 += c1[wv];
            }
            uint32_t nvalid;
            uint32_t run = fe_block_excl_scan(bc0 + bc1, tmp, &nvalid);               // where the thread's first bin starts in the stage
            gdelta[tid * 2u] = dex0 + hp0 - run;                                       // global position of stage slot s with digit d = gdelta[d] + s
            gdelta[tid * 2u + 1u] = dex1 + hp1 - (run + bc0);
            uint32_t run1 = run + bc0;
#pragma unroll
            for (int wv = 0; wv < FE_WAVES; ++wv) {
                *reinterpret_cast<uint32_t*>(&cnt16[(uint32_t)wv * 512u + tid * 2u]) = run | (run1 << 16);
                run += c0[wv]; run1 += c1[wv];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                if (vmask & (1u << r)) {
                    uint32_t kk = key[r];
                    asm volatile("" : "+v"(kk));           // (recompute the digit: keeps the ranking loop's sixteen LDS addresses from living across the barrier)
                    const uint32_t pos = cnt16[wave * 512u + (((kk - sub) >> shift) & dmask)] + rank[r];
                    stage_k[pos] = key[r];
                    stage_v[pos] = val[r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t slot = (uint32_t)i * FE_BLOCK + tid;
                if (slot < nvalid) {
                    const uint32_t k = stage_k[slot];
                    const uint32_t dst = gdelta[((k - sub) >> shift) & dmask] + slot;
                    if (kout) kout[dst] = k + key_add;
                    vout[dst] = stage_v[slot];
                    if (ranges_enc) {
                        // A6: neighbours in the stage with the same key are neighbours in the output (same digit, consecutive slots), so a
                        // run's ends are where the stage's key changes; the same tile's runs of other partitions merge through the max
                        const uint32_t kp = slot > 0 ? stage_k[slot - 1] : ~k, kn = slot + 1 < nvalid ? stage_k[slot + 1] : ~k;
                        uint32_t* const e = ranges_enc + 2 * (size_t)(key_add + k);
                        const uint32_t at = S.base + dst;
                        if (kp != k) atomicMax(e, ~at);
                        if (kn != k) atomicMax(e + 1, at + 1u);
                    }
                }
            }
            if (EARLY && p + gv < nparts) request(p + gv);        // (a workgroup rarely has a second partition: the grids cover the expected counts)
        } else {
#pragma unroll 1
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t d = tid * 8u + k;
                uint32_t acc = 0;
#pragma unroll
                for (int wv = 0; wv < FE_WAVES; ++wv) { const uint32_t cw = cnt16[(uint32_t)wv * 2048u + d]; cnt16[(uint32_t)wv * 2048u + d] = (uint16_t)acc; acc += cw; }
                gdelta[d] = dexl[d] + (d < nbins ? hist[(size_t)d * nbtot + row] : 0u);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                if (vmask & (1u << r)) {
                    uint32_t kk = key[r];
                    asm volatile("" : "+v"(kk));
                    const uint32_t d = ((kk - sub) >> shift) & dmask;
                    const uint32_t dst = gdelta[d] + cnt16[wave * 2048u + d] + rank[r];
                    if (kout) kout[dst] = key[r] + key_add;
                    vout[dst] = val[r];
                }
            }
            if (EARLY && p + gv < nparts) request(p + gv);
        }
    }
}

static inline int fe_items_for(uint64_t n_total) {
    static const uint64_t thr = [] { const char* e = getenv("DVS_FE_ITEMS8_BELOW"); return e ? (uint64_t)atoll(e) : 1500000ull; }();
    return n_total <= thr ? 8 : 16;
}

size_t dvs_fe_hist_words(uint64_t max_elems, int n_views, int max_bins) {
    const uint64_t rows = (max_elems + (uint64_t)FE_BLOCK * 8 - 1) / ((uint64_t)FE_BLOCK * 8) + (uint64_t)n_views + 6;     // sized for the smaller partition
    return (size_t)(rows * (uint64_t)max_bins);
}

struct FeSortLaunch {
    hipStream_t st;
    int V;
    int items;                 // 8 / 16 keys per thread
    uint32_t grid_per_view;    // workgroups per view (the kernels stride over the partitions)
    uint32_t* hist; uint32_t nbtot; uint32_t* totals;
    int rank_atomic;           // k_seg_scatter<.., true>: only after dvs_fe_probe_rank_atomic() said yes on this device
};

static hipError_t fe_launch_pass(const FeSortLaunch& L, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, DvsSeg* seg_in,
                                 DvsSeg* seg_out, int adaptive, int pass, int shift, int bits, int cull, uint32_t key_add_per_view,
                                 const uint32_t* kred, uint32_t* ranges_enc = nullptr, int key16 = 0) {
    const dim3 grid(L.grid_per_view * (uint32_t)L.V), blk(FE_BLOCK);
    const int maxbins = adaptive ? FE_MAXBINS : (1 << bits);
    const dim3 rgrid((uint32_t)((maxbins + FE_WAVES - 1) / FE_WAVES) * (uint32_t)L.V);
#define FE_PASS(I, A)                                                                                                                          \
    do {                                                                                                                                       \
        hipLaunchKernelGGL(k_seg_hist<I>, grid, blk, 0, L.st, kin, seg_in, L.V, adaptive, pass, shift, bits, cull, L.hist, L.nbtot, kred, key16); \
        hipLaunchKernelGGL(k_seg_rowscan, rgrid, blk, 0, L.st, L.hist, L.nbtot, (const DvsSeg*)seg_in, L.V, adaptive, bits, (uint32_t)(FE_BLOCK * I), L.totals); \
        hipLaunchKernelGGL((k_seg_scatter<I, A>), grid, blk, 0, L.st, kin, vin, kout, vout, (const DvsSeg*)seg_in, seg_out, L.V, adaptive, pass, shift, bits, \
                           cull, key_add_per_view, (const uint32_t*)L.hist, L.nbtot, (const uint32_t*)L.totals, key16, ranges_enc);               \
    } while (0)
    if (L.items == 8) { if (L.rank_atomic) FE_PASS(8, true); else FE_PASS(8, false); }
    else { if (L.rank_atomic) FE_PASS(16, true); else FE_PASS(16, false); }
#undef FE_PASS
    return hipGetLastError();
}

// A5, low 32 key bits: every view's depth keys (keys0, view-major [V][n], culled = 0xFFFFFFFF) -> the visible splats' indices in
// depth order in vals1[view * n + j], j < seg_vis[view].count. Three passes: keys0 -> (keys1, vals1) -> (keys0, vals0) -> vals1.
hipError_t dvs_launch_depth_sort(hipStream_t st, int n, int V, uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1,
                                 DvsSeg* seg_all, DvsSeg* seg_vis, const uint32_t* kred, uint32_t* hist, uint32_t* totals, int rank_atomic) {
    if (n <= 0 || V <= 0) return hipSuccess;
    FeSortLaunch L;
    L.st = st; L.V = V; L.items = fe_items_for((uint64_t)n * V); L.rank_atomic = rank_atomic;
    const uint32_t part = (uint32_t)FE_BLOCK * L.items;
    L.grid_per_view = ((uint32_t)n + part - 1) / part;
    L.hist = hist; L.nbtot = L.grid_per_view * (uint32_t)V + 1; L.totals = totals;
    hipError_t e;
    if ((e = fe_launch_pass(L, keys0, nullptr, keys1, vals1, seg_all, seg_vis, 1, 0, 0, 0, 1, 0, kred)) != hipSuccess) return e;
    if ((e = fe_launch_pass(L, keys1, vals1, keys0, vals0, seg_vis, nullptr, 1, 1, 0, 0, 0, 0, nullptr)) != hipSuccess) return e;
    return fe_launch_pass(L, keys0, vals0, nullptr, vals1, seg_vis, nullptr, 1, 2, 0, 0, 0, 0, nullptr);
}
uint32_t dvs_depth_sort_rows_per_view(int n, int V) {        // partitions (histogram rows) per view: the pstart stride of the depth segments
    const uint32_t part = (uint32_t)FE_BLOCK * fe_items_for((uint64_t)n * V);
    return ((uint32_t)n + part - 1) / part;
}

// Stable LSD sort of the segments' (key, value) pairs over the key bits [bit_lo, bit_lo + bits) in ceil(bits / 9) passes (digits as equal as
// possible, widest first). Buffers 0 hold the input; the result is in buffers (*result_in). grid_elems sizes the grids (an upper bound
// or an estimate of the total element count: the kernels stride), cap_elems the histogram table (rows = cap / partition + V + 1 were
// assumed when the segments' pstart were assigned: see dvs_fe_tile_part). key_add_per_view: added to the keys of view v (times v) when
// the LAST pass writes them (the tile sort hands out view * tiles + tile). ranges_enc (nullable): the last pass also leaves the tile
// ranges [V * tiles][2] (zeroed by the caller) as (~start, end) — A6 fused; write_last_keys = 0: it does not write the sorted keys.
static void fe_split_bits(int bits, int* npass, int widths[4]) {
    int np = (bits + FE_REORDER_BITS - 1) / FE_REORDER_BITS;
    if (np < 1) np = 1;
    if (np > 4) np = 4;
    int left = bits;
    for (int k = 0; k < np; ++k) { const int wdt = (left + (np - k) - 1) / (np - k); widths[k] = wdt > 0 ? wdt : 1; left -= wdt; }
    *npass = np;
}
uint32_t dvs_fe_part_for(uint64_t grid_elems) { return (uint32_t)FE_BLOCK * (uint32_t)fe_items_for(grid_elems); }

hipError_t dvs_launch_seg_sort(hipStream_t st, int V, uint32_t* keys0, uint32_t* vals0, uint32_t* keys1, uint32_t* vals1, DvsSeg* seg, int bit_lo, int bits,
                               uint64_t grid_elems, uint32_t part, uint32_t nbtot, uint32_t* hist, uint32_t* totals, uint32_t key_add_per_view,
                               int* result_in, uint32_t* ranges_enc, int write_last_keys, int first_keys16, int rank_atomic) {
    if (result_in) *result_in = 0;
    if (V <= 0 || bits <= 0) return hipSuccess;
    int npass, widths[4];
    fe_split_bits(bits, &npass, widths);
    FeSortLaunch L;
    L.st = st; L.V = V; L.items = (int)(part / FE_BLOCK); L.rank_atomic = rank_atomic;
    uint64_t per_view = (grid_elems + (uint64_t)V - 1) / (uint64_t)V;
    uint64_t g = (per_view + part - 1) / part + 1;
    if (g > 65535u * 16u) g = 65535u * 16u;
    L.grid_per_view = (uint32_t)(g < 1 ? 1 : g);
    L.hist = hist; L.nbtot = nbtot; L.totals = totals;
    uint32_t* kk[2] = {keys0, keys1};
    uint32_t* vv[2] = {vals0, vals1};
    int c = 0, shift = bit_lo;
    for (int k = 0; k < npass; ++k) {
        const bool last = k == npass - 1;
        // the last pass of the tile sort builds the tile ranges (A6) and may skip the keys: nothing reads them but the exported state
        hipError_t e = fe_launch_pass(L, kk[c], vv[c], (last && !write_last_keys) ? nullptr : kk[c ^ 1], vv[c ^ 1], seg, nullptr, 0, k, shift, widths[k], 0,
                                      last ? key_add_per_view : 0u, nullptr, last ? ranges_enc : nullptr, (k == 0 && first_keys16) ? 1 : 0);
        if (e != hipSuccess) return e;
        shift += widths[k];
        c ^= 1;
    }
    if (result_in) *result_in = c;
    return hipSuccess;
}

// ---- is the returning LDS add lane-ordered on this device? ---------------------------------------------------------------------------
// Every wave of the launch ranks 64 x ROUNDS synthetic digits per pattern both ways — the ballot multisplit with leader-written counters
// (the form rounds 2-5 shipped, correct by construction) and ds_add_rtn_u32 on packed u16 counters exactly as k_seg_scatter<.., true>
// does — and raises *bad on the first disagreement. Patterns: one digit for all lanes (64-way conflict), two digits sharing a counter
// word, digits by lane pairs / triples / halves, stripes, and hashed digits of 1 .. 9 bits, with the four waves of a workgroup hammering
// their own counters at the same time as in the scatter. 0 disagreements over ~10^6 ranks -> the atomic ranking is selected.
__global__ void __launch_bounds__(FE_BLOCK)
k_fe_probe_rank_atomic(uint32_t* __restrict__ bad, uint32_t seed) {
    __shared__ uint32_t cnt_a[FE_WAVES * 256];             // packed u16 counters, atomic form: 512 digits per wave
    __shared__ uint16_t cnt_b[FE_WAVES * 512];             // ballot form
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6;
    uint32_t errs = 0;
    for (uint32_t pat = 0; pat < 24; ++pat) {
        for (uint32_t e = threadIdx.x; e < FE_WAVES * 256; e += FE_BLOCK) cnt_a[e] = 0u;
        for (uint32_t e = threadIdx.x; e < FE_WAVES * 512; e += FE_BLOCK) cnt_b[e] = 0;
        __syncthreads();
        const uint32_t bits = pat < 8 ? 9u : 1u + (pat - 8u) % 9u;
        for (uint32_t r = 0; r < 16; ++r) {
            uint32_t h = (lane * 0x9E3779B1u) ^ ((blockIdx.x * FE_WAVES + wave) * 0x85EBCA6Bu) ^ ((pat * 16u + r) * 0xC2B2AE35u) ^ seed;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
            uint32_t d;
            switch (pat) {
                case 0: d = 5u; break;                                   // every lane the same counter
                case 1: d = 6u + (lane & 1u); break;                     // the two halves of one word, alternating
                case 2: d = lane >> 1; break;                            // pairs
                case 3: d = lane / 3u; break;
                case 4: d = lane >> 5; break;                            // the two wave halves
                case 5: d = lane & 7u; break;                            // stripes
                case 6: d = (lane & 1u) ? 511u : (h & 3u); break;
                case 7: d = 63u - lane; break;                           // no conflicts, descending
                default: d = h & ((1u << bits) - 1u); break;
            }
            const bool valid = (h >> 20 & 15u) != 0u || pat < 8;         // a few idle lanes in the hashed patterns
            uint32_t ra = 0, rb = 0;
            if (valid) {
                const uint32_t old = atomicAdd(&cnt_a[wave * 256u + (d >> 1)], (d & 1u) ? 0x10000u : 1u);
                ra = (d & 1u) ? (old >> 16) : (old & 0xFFFFu);
            }
            const uint64_t peers = fe_match(d, valid, 9u);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            if (valid) {
                const uint32_t prev = cnt_b[wave * 512u + d];
                if (below == 0) cnt_b[wave * 512u + d] = (uint16_t)(prev + (uint32_t)__popcll(peers));
                rb = prev + below;
            }
            errs += (valid && ra != rb) ? 1u : 0u;
        }
        __syncthreads();
    }
    if (errs) atomicAdd(bad, errs);
}
hipError_t dvs_fe_probe_rank_atomic(hipStream_t st, uint32_t* bad_dev /*one zeroed word*/) {
    hipLaunchKernelGGL(k_fe_probe_rank_atomic, dim3(512), dim3(FE_BLOCK), 0, st, bad_dev, 0x1234567u);
    hipLaunchKernelGGL(k_fe_probe_rank_atomic, dim3(512), dim3(FE_BLOCK), 0, st, bad_dev, 0x89ABCDEu);
    return hipGetLastError();
}

// ---- tile rectangles: three record formats -----------------------------------------------------------------------------------------
//   FE_RECT_U8   4 B  minx | miny << 8 | width << 16 | height << 24            (tiles_x, tiles_y <= 255: up to 4080 x 4080 pixels)
//   FE_RECT_U16  8 B  [minx | maxx << 16, miny | maxy << 16]                    (larger images)
//   FE_RECT_TIGHT 16 B the 8-B rectangle + the 64-bit tile mask of DVS_TILES_TIGHT
template <int FMT> struct FeRect;
template <> struct FeRect<DVS_FE_RECT_U8> {
    typedef uint32_t T;
    static __device__ __forceinline__ uint32_t count(T r) { return ((r >> 16) & 0xFFu) * (r >> 24); }
    static __device__ __forceinline__ uint32_t minx(T r) { return r & 0xFFu; }
    static __device__ __forceinline__ uint32_t miny(T r) { return (r >> 8) & 0xFFu; }
    static __device__ __forceinline__ uint32_t width(T r) { return (r >> 16) & 0xFFu; }
    static __device__ __forceinline__ T zero() { return 0u; }
};
template <> struct FeRect<DVS_FE_RECT_U16> {
    typedef uint2 T;
    static __device__ __forceinline__ uint32_t count(T r) { return ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu)); }
    static __device__ __forceinline__ uint32_t minx(T r) { return r.x & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t miny(T r) { return r.y & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t width(T r) { return (r.x >> 16) - (r.x & 0xFFFFu); }
    static __device__ __forceinline__ T zero() { return make_uint2(0u, 0u); }
};
template <> struct FeRect<DVS_FE_RECT_TIGHT> {
    typedef uint4 T;
    static __device__ __forceinline__ uint32_t count(T r) {
        const uint32_t both = r.z & r.w;
        return both == 0xFFFFFFFFu ? ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu)) : (uint32_t)(__popc(r.z) + __popc(r.w));
    }
    static __device__ __forceinline__ uint32_t minx(T r) { return r.x & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t miny(T r) { return r.y & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t width(T r) { return (r.x >> 16) - (r.x & 0xFFFFu); }
    static __device__ __forceinline__ T zero() { return make_uint4(0u, 0u, 0u, 0u); }
};

// ---- A3: tile counts in depth order ---------------------------------------------------------------------------------------------------
// Workgroup (view, block): 256 consecutive elements of the view's depth-sorted list. The ONE random gather of the stage —
// rect[view][sorted id] — is done here and the rectangles are re-emitted in depth order, so that A4 streams. With FE_RECT_U8 a view's
// rectangle array is 4 B x n (4 MB at 10^6 splats): it stays in the L2 of the XCD that works for the view.
// block_sums[view][block] = instances of the block; super[view][block / 256] += them (one 64-bit atomic per workgroup).
template <int FMT>
__global__ void __launch_bounds__(FE_BLOCK)
k_seg_blocksum(int n, int V, uint32_t nbv, uint32_t nsb, const DvsSeg* __restrict__ seg_vis, const uint32_t* __restrict__ sorted_ids,
               const typename FeRect<FMT>::T* __restrict__ rect, typename FeRect<FMT>::T* __restrict__ rect_sorted,
               uint32_t* __restrict__ block_sums, unsigned long long* __restrict__ super) {
    // a workgroup takes FOUR consecutive blocks of 256 elements (four independent id -> rectangle gathers in flight per thread)
    __shared__ uint32_t wsum[4][FE_WAVES];
    const uint32_t view = blockIdx.x % (uint32_t)V, blk0 = (blockIdx.x / (uint32_t)V) * 4u;
    const uint32_t nvis = seg_vis[view].count;
    if (blk0 * FE_BLOCK >= nvis) return;                      // (uniform; A4 never reads the sums of blocks behind the visible splats)
    const size_t o = (size_t)view * n;
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6;
    uint32_t id[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t j = (blk0 + (uint32_t)k) * FE_BLOCK + threadIdx.x; id[k] = j < nvis ? sorted_ids[o + j] : 0u; }
    typename FeRect<FMT>::T r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = rect[o + id[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t j = (blk0 + (uint32_t)k) * FE_BLOCK + threadIdx.x;
        uint32_t v = 0;
        if (j < nvis) { rect_sorted[o + j] = r[k]; v = FeRect<FMT>::count(r[k]); }
        const uint32_t s = fe_wave_sum(v);
        if (lane == 0) wsum[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x, blk = blk0 + k;
        uint32_t tot = 0;
#pragma unroll
        for (int wv = 0; wv < FE_WAVES; ++wv) tot += wsum[k][wv];
        if (blk < nbv) block_sums[(size_t)view * nbv + blk] = tot;             // (blocks behind the visible splats get 0)
        // the four blocks lie in one super block (256 blocks): one 64-bit atomic per workgroup; one counter per 256 B, so that the
        // counters of a view are served in parallel
        const uint32_t t4 = tot + (uint32_t)__shfl_down(tot, 1, 64) + (uint32_t)__shfl_down(tot, 2, 64) + (uint32_t)__shfl_down(tot, 3, 64);
        if (k == 0 && t4) atomicAdd(&super[((size_t)view * nsb + (blk0 >> 8)) * DVS_FE_SUPER_STRIDE], (unsigned long long)t4);
    }
}

// One workgroup of 16 waves, wave v = view v: exclusive scan of the view's super sums (-> superexcl), the views' instance ranges
// (-> seg_tile: base = first instance, count, pstart = first histogram row of the tile sort), T -> total[0], total[1] += (T > capacity).
// Ranges are clamped to the capacity (an overflowing forward is reported and its outputs are invalid, but nothing is written out of
// bounds). tile_part = keys per partition of the tile sort.
__global__ void __launch_bounds__(1024)
k_seg_totals(int V, uint32_t nsb, const unsigned long long* __restrict__ super, uint32_t* __restrict__ superexcl, DvsSeg* __restrict__ seg_tile,
             uint32_t tile_part, unsigned long long* __restrict__ total, unsigned long long capacity) {
    __shared__ unsigned long long vsum[DVS_MAX_VIEWS];
    const uint32_t lane = fe_lane(), wave = threadIdx.x >> 6;
    if (wave < (uint32_t)V) {
        unsigned long long carry = 0ull;
        for (uint32_t c0 = 0; c0 < nsb; c0 += 64) {
            const uint32_t i = c0 + lane;
            const unsigned long long v = i < nsb ? super[((size_t)wave * nsb + i) * DVS_FE_SUPER_STRIDE] : 0ull;
            unsigned long long inc = v;
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) { const unsigned long long o = __shfl_up(inc, k, 64); if (lane >= (uint32_t)k) inc += o; }
            const unsigned long long ex = carry + inc - v;
            if (i < nsb) superexcl[(size_t)wave * nsb + i] = ex > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ex;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) vsum[wave] = carry;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long start = 0ull;
        uint32_t prow = 0;
        for (int v = 0; v < V; ++v) {
            const unsigned long long end = start + vsum[v];
            const unsigned long long cs = start < capacity ? start : capacity, ce = end < capacity ? end : capacity;
            DvsSeg s;
            s.base = (uint32_t)cs; s.count = (uint32_t)(ce - cs); s.pstart = prow; s.sub = 0u; s.bits = 0u; s._r0 = s._r1 = s._r2 = 0u;
            seg_tile[v] = s;
            prow += (s.count + tile_part - 1) / tile_part;
            start = end;
        }
        total[0] = start;
        if (start > capacity) total[1] += 1ull;
    }
}

// position of the (t + 1)-th set bit of a 64-bit mask (t < popcount): the tile index of instance t of a tightened rectangle
__device__ __forceinline__ uint32_t fe_nth_set_bit64(uint32_t lo, uint32_t hi, uint32_t t) {
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

// ---- A4: duplicate with keys, in depth order, view by view ------------------------------------------------------------------------------
// Workgroup (view, block pair) streams (splat index, rectangle) of its 2 x 256 elements and emits their instances at
// seg_tile[view].base + superexcl[view][block / 256] + sum of the block sums since that super block + the in-block offset.
// Every wave emits the instances of its own 64 splats cooperatively, 64 consecutive output slots at a time: the splats whose first
// instance falls into the 64 slots leave their lane number there (LDS, one word per slot), a DPP max-scan carries it to the slots behind,
// and the slot's tile follows from its index inside the rectangle (row-major) — a store instruction covers 64 consecutive instances
// whatever the rectangle sizes are. (Rounds 3-5a found a slot's splat by a binary search over the exclusive offsets: six dependent LDS
// round trips per 64 instances in a kernel that is one latency chain per workgroup.) Everything the chain needs from global memory — both
// blocks' elements, the block sums, the view's start, the super block's prefix — is requested before the first scan.
// Keys are tile ids INSIDE the view; values are view * n + splat. Instances beyond `capacity` are not written (k_seg_totals has raised
// the overflow counter).
__device__ __forceinline__ uint32_t fe_wave_incl_max(uint32_t v) {
#define FE_DPP_MAX(ctrl, rmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false); v = v > o_ ? v : o_; }
    FE_DPP_MAX(0x111, 0xF); FE_DPP_MAX(0x112, 0xF); FE_DPP_MAX(0x114, 0xF); FE_DPP_MAX(0x118, 0xF);   // row_shr:1, 2, 4, 8
    FE_DPP_MAX(0x142, 0xA);                                                                              // row_bcast:15 -> rows 1, 3
    FE_DPP_MAX(0x143, 0xC);                                                                              // row_bcast:31 -> rows 2, 3
#undef FE_DPP_MAX
    return v;
}
template <int FMT>
__global__ void __launch_bounds__(FE_BLOCK)
k_seg_duplicate(int n, int V, uint32_t nbv, uint32_t nsb, const DvsSeg* __restrict__ seg_vis, const DvsSeg* __restrict__ seg_tile,
                const uint32_t* __restrict__ sorted_ids, const typename FeRect<FMT>::T* __restrict__ rect_sorted,
                const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ superexcl, int tiles_x,
                uint32_t* __restrict__ inst_tile, uint32_t* __restrict__ inst_splat, unsigned long long capacity, int key16 /*tile ids as 16-bit words*/) {
    __shared__ uint32_t tmp[FE_WAVES];
    __shared__ uint32_t s_pre[FE_BLOCK];          // exclusive offset of the thread's first instance inside the block
    __shared__ uint32_t s_id[FE_BLOCK];
    __shared__ uint32_t s_tile[FE_BLOCK];         // tile id of the rectangle's first tile
    __shared__ uint32_t s_w[FE_BLOCK];            // rectangle width in tiles
    __shared__ uint32_t s_mark[FE_BLOCK];         // per wave: 64 output slots -> (lane of the splat that starts there) + 1
    __shared__ uint2 s_mask[FMT == DVS_FE_RECT_TIGHT ? FE_BLOCK : 1];
    const uint32_t view = blockIdx.x % (uint32_t)V, blk0 = (blockIdx.x / (uint32_t)V) * 2u;
    const uint32_t nvis = seg_vis[view].count;
    if (blk0 * FE_BLOCK >= nvis) return;
    const size_t o = (size_t)view * n;
    const uint32_t lane = fe_lane(), w0 = (threadIdx.x >> 6) * 64u;
    uint32_t id[2] = {0u, 0u};
    typename FeRect<FMT>::T r[2] = {FeRect<FMT>::zero(), FeRect<FMT>::zero()};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t j = (blk0 + (uint32_t)h) * FE_BLOCK + threadIdx.x;
        if (j < nvis) { id[h] = sorted_ids[o + j]; r[h] = rect_sorted[o + j]; }
    }
    // instances before the first block: the super block's offset + the block sums since (both blocks lie in one super block: 256 is even)
    const uint32_t sb = blk0 >> 8, nprev = blk0 & 255u;
    const uint32_t part = threadIdx.x < nprev ? block_sums[(size_t)view * nbv + (size_t)sb * 256 + threadIdx.x] : 0u;
    const unsigned long long vbase = (unsigned long long)seg_tile[view].base + superexcl[(size_t)view * nsb + sb];
    // (seg_tile.base is the clamped start: when it was clamped the forward has overflowed and every store below is dropped)
    uint32_t before, tot;
    (void)fe_block_excl_scan(part, tmp, &before);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((blk0 + (uint32_t)h) * FE_BLOCK >= nvis) break;        // (uniform)
        const uint32_t touched = FeRect<FMT>::count(r[h]);          // (0 for the lanes behind the visible splats: zero rectangle)
        const uint32_t pre = fe_block_excl_scan(touched, tmp, &tot);
        s_pre[threadIdx.x] = pre;
        s_id[threadIdx.x] = (uint32_t)o + id[h];
        s_tile[threadIdx.x] = FeRect<FMT>::miny(r[h]) * (uint32_t)tiles_x + FeRect<FMT>::minx(r[h]);
        s_w[threadIdx.x] = FeRect<FMT>::width(r[h]);
        if constexpr (FMT == DVS_FE_RECT_TIGHT) s_mask[threadIdx.x] = make_uint2(r[h].z, r[h].w);
        // (every wave reads only what its own lanes wrote: no workgroup barrier; a wave's LDS operations execute in order)
        const unsigned long long gbase = vbase + before;
        const uint32_t k_beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre);
        const uint32_t k_end = (uint32_t)__builtin_amdgcn_readlane((int)(pre + touched), 63);
        uint32_t cur1 = 0;                                         // (lane + 1) of the splat that holds the slot before the current 64
        for (uint32_t k0 = k_beg; k0 < k_end; k0 += 64u) {
            s_mark[w0 + lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            const uint32_t rel = pre - k0;                         // (wraps for splats that start before k0)
            if (touched != 0u && rel < 64u) s_mark[w0 + rel] = lane + 1u;      // distinct slots: non-empty splats start at distinct offsets
            __builtin_amdgcn_wave_barrier();
            uint32_t m = s_mark[w0 + lane];
            __builtin_amdgcn_wave_barrier();
            m = fe_wave_incl_max(m);
            m = m > cur1 ? m : cur1;                               // slots in front of the first start belong to the carried splat
            cur1 = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
            const uint32_t k = k0 + lane;
            if (k < k_end) {
                const uint32_t src = w0 + m - 1u, w = s_w[src];
                uint32_t t = k - s_pre[src];
                if constexpr (FMT == DVS_FE_RECT_TIGHT) {         // instance t of the splat = its (t + 1)-th surviving tile
                    const uint2 mk = s_mask[src];
                    if ((mk.x & mk.y) != 0xFFFFFFFFu) t = fe_nth_set_bit64(mk.x, mk.y, t);
                }
                // row = t / w by a float quotient and one correction step (t < 2^24: a rectangle has fewer tiles than the screen)
                uint32_t q = (uint32_t)((float)t * __builtin_amdgcn_rcpf((float)w));
                int32_t rem = (int32_t)(t - q * w);
                if (rem < 0) { --q; rem += (int32_t)w; } else if (rem >= (int32_t)w) { ++q; rem -= (int32_t)w; }
                const unsigned long long g = gbase + k;
                if (g < capacity) {
                    const uint32_t tile = s_tile[src] + q * (uint32_t)tiles_x + (uint32_t)rem;
                    if (key16) reinterpret_cast<uint16_t*>(inst_tile)[g] = (uint16_t)tile; else inst_tile[g] = tile;
                    inst_splat[g] = s_id[src];
                }
            }
        }
        before += tot;
        __builtin_amdgcn_wave_barrier();                           // (the next block's LDS writes stay behind this block's reads)
    }
}

hipError_t dvs_launch_seg_binning(hipStream_t st, int n, int V, int rect_fmt, const DvsSeg* seg_vis, DvsSeg* seg_tile, const uint32_t* sorted_ids,
                                  const uint32_t* rect, uint32_t* rect_sorted, uint32_t* block_sums, unsigned long long* super, uint32_t* superexcl,
                                  uint32_t tile_part, unsigned long long* total_dev, unsigned long long capacity, int stage /*0: A3 + totals, 1: A4*/,
                                  int tiles_x, uint32_t* inst_tile, uint32_t* inst_splat, int key16) {
    if (n <= 0 || V <= 0) return hipSuccess;
    const uint32_t nbv = (uint32_t)((n + FE_BLOCK - 1) / FE_BLOCK), nsb = (nbv + 255u) / 256u;
    const dim3 grid(((nbv + 1u) / 2u) * (uint32_t)V), grid3(((nbv + 3u) / 4u) * (uint32_t)V), blk(FE_BLOCK);
    if (stage == 0) {
#define FE_A3(F) hipLaunchKernelGGL(k_seg_blocksum<F>, grid3, blk, 0, st, n, V, nbv, nsb, seg_vis, sorted_ids, (const FeRect<F>::T*)rect, (FeRect<F>::T*)rect_sorted, block_sums, super)
        if (rect_fmt == DVS_FE_RECT_U8) FE_A3(DVS_FE_RECT_U8); else if (rect_fmt == DVS_FE_RECT_U16) FE_A3(DVS_FE_RECT_U16); else FE_A3(DVS_FE_RECT_TIGHT);
#undef FE_A3
        hipLaunchKernelGGL(k_seg_totals, dim3(1), dim3(1024), 0, st, V, nsb, (const unsigned long long*)super, superexcl, seg_tile, tile_part, total_dev, capacity);
    } else {
#define FE_A4(F) hipLaunchKernelGGL(k_seg_duplicate<F>, grid, blk, 0, st, n, V, nbv, nsb, seg_vis, (const DvsSeg*)seg_tile, sorted_ids, (const FeRect<F>::T*)rect_sorted, \
                                    (const uint32_t*)block_sums, (const uint32_t*)superexcl, tiles_x, inst_tile, inst_splat, capacity, key16)
        if (rect_fmt == DVS_FE_RECT_U8) FE_A4(DVS_FE_RECT_U8); else if (rect_fmt == DVS_FE_RECT_U16) FE_A4(DVS_FE_RECT_U16); else FE_A4(DVS_FE_RECT_TIGHT);
#undef FE_A4
    }
    return hipGetLastError();
}
void dvs_fe_block_counts(int n, uint32_t* nbv, uint32_t* nsb) { *nbv = (uint32_t)((n + FE_BLOCK - 1) / FE_BLOCK); *nsb = (*nbv + 255u) / 256u; }

// segment descriptors of the depth sort's input: view v = elements [v * n, (v + 1) * n), histogram rows from v * rows_per_view
__global__ void k_seg_init(int n, int V, uint32_t rows_per_view, DvsSeg* __restrict__ seg_all) {
    const int v = threadIdx.x;
    if (v < V) { DvsSeg s; s.base = (uint32_t)v * (uint32_t)n; s.count = (uint32_t)n; s.pstart = (uint32_t)v * rows_per_view; s.sub = 0; s.bits = 8; s._r0 = s._r1 = s._r2 = 0; seg_all[v] = s; }
}
hipError_t dvs_launch_seg_init(hipStream_t st, int n, int V, uint32_t rows_per_view, DvsSeg* seg_all) {
    hipLaunchKernelGGL(k_seg_init, dim3(1), dim3(64), 0, st, n, V, rows_per_view, seg_all);
    return hipGetLastError();
}

// ---- A6 as its own kernel: only when k_render_fwd does not composite (experiment builds' per-block forward) or DVS_FE_NO_FUSE_A6=1;
// normally the tile sort's last pass builds the ranges (k_seg_scatter) ---------------------------------------------------------------
__global__ void __launch_bounds__(FE_BLOCK)
k_tile_ranges(uint64_t T_host, const uint64_t* __restrict__ T_dev, const uint32_t* __restrict__ sorted_tile, uint2* __restrict__ ranges) {
    const uint64_t T = T_dev ? (*T_dev < T_host ? *T_dev : T_host) : T_host;
    // four consecutive instances per thread (one 16-B load + the two neighbours)
    for (uint64_t q = (uint64_t)blockIdx.x * FE_BLOCK + threadIdx.x; q * 4 < T; q += (uint64_t)gridDim.x * FE_BLOCK) {
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
    const uint32_t nb = (uint32_t)((T_grid + 4 * FE_BLOCK - 1) / (4 * FE_BLOCK));
    hipLaunchKernelGGL(k_tile_ranges, dim3(nb), dim3(FE_BLOCK), 0, st, T, T_dev, sorted_tile, (uint2*)ranges);
    return hipGetLastError();
}

// ---- parity export: canonical 64-bit keys -------------------------------------------------------------------
__global__ void __launch_bounds__(FE_BLOCK)
k_export_keys(uint64_t T, const uint32_t* __restrict__ sorted_tile, const uint32_t* __restrict__ sorted_splat,
              const float* __restrict__ depth, uint64_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * FE_BLOCK + threadIdx.x;
    if (j >= T) return;
    out[j] = ((uint64_t)sorted_tile[j] << 32) | (uint64_t)__float_as_uint(depth[sorted_splat[j]]);
}

hipError_t dvs_launch_export_keys(hipStream_t st, uint64_t T, const uint32_t* sorted_tile, const uint32_t* sorted_splat,
                                  const float* depth, uint64_t* out_keys) {
    if (T == 0) return hipSuccess;
    const uint32_t nb = (uint32_t)((T + FE_BLOCK - 1) / FE_BLOCK);
    hipLaunchKernelGGL(k_export_keys, dim3(nb), dim3(FE_BLOCK), 0, st, T, sorted_tile, sorted_splat, depth, out_keys);
    return hipGetLastError();
}
