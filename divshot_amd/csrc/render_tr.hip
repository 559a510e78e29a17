// render_tr.hip — A8 (alpha-composite backward), variant "tr": per-pixel recurrence and per-splat accumulation are split, and the
// (pixel, splat) pairs change hands through a small LDS transposition buffer instead of a cross-lane reduction per visit.
//
// The round-2 kernel (render_blocks.hip) spends 41 of its 81 vector instructions per list step on forming the 11 per-pixel
// partials of a (splat, 4x4 block) pair, folding them over the block's 16 lanes (9 v_permlane*_swap + 15 v_add + 6 ds_swizzle) and
// merging the group totals into the per-wave table. Every one of those sums is linear in just TWO per-pixel scalars of the pair,
//     v5 = G dL/dalpha   and   w = alpha T,
// against factors that depend on the pixel alone (its coordinates, its upstream gradient). So:
//   phase 1 (lane = pixel; the sequential part): the same per-4x4-block lists and back-to-front recurrence as before, but a step
//            ends when (v5, w) are known: they go into slot s of the wave's transposition buffer TB[slot][v5 | w][64 lanes].
//   phase 2 (every 4 steps; lane = (block g, slot s, pixel row r)): each lane reads the four pixels of ITS row of ITS (block, slot)
//            pair as two ds_read_b128, and accumulates serially in registers: row moments A = sum v5, B = sum x v5, C = sum x^2 v5
//            (x = 0..3: two FMAs each), moved to the splat's mean algebraically (S_x = Dx A - B, S_xx = Dx S_x - (Dx B - C), ...:
//            exact algebra, fp32 roundoff of the same size as the direct sums), the colour sums sum w dL/dC (the lane keeps its four
//            pixels' upstream gradients in registers), and the abs-grad sums with the per-pixel linear forms stepped by one
//            subtraction per pixel. The four rows of a pair sit in the four 16-lane rows of the wave, so the only cross-lane work
//            left is the two packing swaps (v_permlane32_swap / v_permlane16_swap: 8 swaps + 9 adds per FOUR steps instead of 9 + 15
//            + 6 swizzles per step), after which each of the four lanes owns three of the pair's twelve totals and adds them to the
//            per-wave table (plain read-add-write, one block group at a time, as in render_blocks.hip).
// Per list step: 32 + 80 / 4 = 52 vector instructions instead of 81 (ISA count, abs-grad on). The cursor is one saturating subtract:
// the sixteen lists are stored interleaved behind a row of sentinels that point at an all-zero dummy entry (opacity 0 -> alpha 0 ->
// "does not contribute"), so an exhausted group needs no predicate. The pair is written with ds_write_addtid_b32 (LDS address = M0 +
// offset + 4 lane: the scalar unit moves the slot base into M0). The kernel walks the forward's live lists (k_render_fwd: the entries whose
// alpha >= 1/255 ellipse reaches the tile, 72 % of a C3 list) when they exist: 2.48 ms. Measured before those: C3, 8-view launch 2.87-2.90 ms against 3.25-3.28 ms for the
// round-2 kernel; SQ_INSTS_VALU -23 %, SQ_ACTIVE_INST_VALU -26 % (DESIGN.md section 5.1 with the ablations and the variants dropped
// on the way: batches of 32, upstream gradients re-read per round, records prefetched in registers, conflict-free fast path, deferred
// table steps — each stopped by the register count (78 of 80) or the LDS footprint (26.5 of 26.6 KB) at six workgroups per CU).
//
// Same inputs and the same 48-B row contract (moments about the mean, see dvs_get_bwd_intermediates) as the other A8 kernels; the
// opacity factor of the moment / abs-grad sums is applied once per (tile, splat) when the tables are published.
// Reference anchors as in render.hip (alpha rule gsplat_ps.hlsl:60-65, 16x16 groups gaussian_common.hlsl:162-163, abs-grad main.cpp:44).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "dvs_device.h"
#include "dvs_kernels.h"
#include "render_common.h"

#define TR_SLOTS 4
#define TR_SS 136          // floats per slot of the transposition buffer: two planes of 64 + 8 pad — with the phase-2 lane map below the
                           // ds_read_b128 of a wave hit 16 distinct 16-B bank groups per service group (brute-forced over the gfx950 lane groups)
#ifndef TR_SKIP_EMPTY_STEPS
#define TR_SKIP_EMPTY_STEPS 0      // a step in which no lane contributes could skip its second half — measured: 450 of 2.78 M steps per C3 view
#endif                             // (the block test + per-block deepest contributor leave no empty steps); the branch only splits the schedule
// (Round 4 measured two more variants of this kernel and dropped them — the next batch's records touched ahead of their gather, and an
// owner byte per table row that lets conflict-free pairs update in one pass (code in the history at commit cb18e80) — and a wave-autonomous
// form of the whole kernel, one wave per workgroup without any workgroup barrier (commit 7484d16; 25 % slower): DESIGN.md section 5.3,
// profiles/r04_a8_variants_ab.txt, r04_a8_autonomous_ab.txt.)
#ifndef TR_MINW
#define TR_MINW 6          // waves per SIMD the kernel is compiled for (register cap 80; the LDS footprint allows six workgroups per CU)
#endif

template <int BK>
struct __attribute__((aligned(16))) TrLds {
    float4 ea[BK + 1];            // mean x, mean y, cs.x, cs.y          (cs = exponent constants, see render.hip); [BK] = all-zero dummy
    float4 eb[BK + 1];            // cs.z, opacity, colour r, colour g
    float4 ec[BK + 1];            // colour b, conic a, b, c
    uint2 idop[2][BK];            // splat id (row of the gradient table), opacity bits — by batch parity: the publish of batch b reads them
                                  // while batch b - 1 is being staged
    uint8_t list[(BK + 1) * 16];  // element k (1-based, list order) of block b at k * 16 + b; row 0 = sentinels (BK)
    uint32_t cnt[16];             // list lengths
    uint32_t blast[16];           // per block: deepest contributor of any of its pixels
#ifdef TR_STATS
    uint32_t live[BK];            // entry reaches at least one block
#endif
};

// The gather of a batch: splat ids (requested before the previous batch is published, so that round trip runs under the publish), then
// the 64-B records. (Measured: requesting the records a batch ahead too and carrying them across the list loop costs more registers
// than the kernel has at six waves per SIMD — the spills then serialise the loads.)
template <int BK>
__device__ __forceinline__ uint32_t tr_load_id(const uint32_t* __restrict__ sorted_splat, uint32_t first, int cnt) {
    const int e = threadIdx.x % BK;
    return e < cnt ? sorted_splat[first + e] : 0u;
}
// Stage the `cnt` entries of a batch (list positions base ..) and build the sixteen block lists. Thread t handles entry t % BK and the
// GPT = 16 BK / 256 consecutive blocks starting at GPT * (t / BK) (one block row for BK = 64, half a row for BK = 32). All threads
// holding one block sit in one wave (BK = 64) or one half wave (BK = 32), so ranks and lengths come straight from that wave's ballots.
template <int BK>
__device__ __forceinline__ void tr_stage(TrLds<BK>& L, const float4* __restrict__ splat2d, uint32_t id, int cnt, int base, int parity,
                                         float tile_x0, float tile_y0 DVS_DBG_PARAM) {
    const int dbg = DVS_DBG_VALUE;
    constexpr int GPT = 16 * BK / RB;
    static_assert(GPT == 4 || GPT == 2, "BK must be 64 or 32");
    const int t = threadIdx.x, e = t % BK, sub = t / BK, lane = t & 63;
    const int b0 = GPT * sub, row = b0 >> 2, col0 = b0 & 3;
    uint32_t hits = 0;
    if (e < cnt) {
        // the 64-B record (dvs_fwd_state.splat2d): mean, conic | conic c, opacity, colour r g | colour b, -, -, cull bound | cull constants
        const float4 r0 = splat2d[4 * (size_t)id], r1 = splat2d[4 * (size_t)id + 1], r2 = splat2d[4 * (size_t)id + 2], r3 = splat2d[4 * (size_t)id + 3];
        const float a = r0.z, b = r0.w, c = r1.x, op = r1.y;
        if (sub == 0) {
            L.ea[e] = make_float4(r0.x, r0.y, -0.72134752044448170f * a, -1.4426950408889634f * b);
            L.eb[e] = make_float4(-0.72134752044448170f * c, op, r1.z, r1.w);
            L.ec[e] = make_float4(r2.x, a, b, c);
            L.idop[parity][e] = make_uint2(id, __float_as_uint(op));
        }
        // exact ellipse-vs-block test from non-negative terms (derivation: render_blocks.hip stage_blocks / render.hip stage_batch)
        const float bound = r2.w, det_c = r3.x, det_a = r3.y, nb_c = r3.z, nb_a = r3.w;                // per splat, from A2 (DVS_S2D_CULL)
        const float ox = tile_x0 - r0.x + 4.f * (float)col0, oy = tile_y0 - r0.y + 4.f * (float)row;
        const float y0 = oy, y1 = oy + 3.f;
        const bool hin = y0 <= 0.f && y1 >= 0.f;
        const float ye = y0 > 0.f ? y0 : y1;
        const float hx = nb_a * ye, hbase = hin ? __builtin_inff() : ye * ye * det_a;
        if (dbg & 16384) hits = 0xFu;                       // timing only (with dbg 8: no list loop): staging without the ellipse-vs-block tests
        else
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const float x0 = ox + 4.f * (float)i, x1 = x0 + 3.f;
            const bool vin = x0 <= 0.f && x1 >= 0.f;
            const float xe = x0 > 0.f ? x0 : x1;
            const float vy = nb_c * xe, vbase = vin ? __builtin_inff() : xe * xe * det_c;
            const float ty_ = fminf(fmaxf(vy, y0), y1) - vy, tx_ = fminf(fmaxf(hx, x0), x1) - hx;
            const float ev = __builtin_fmaf(c * ty_, ty_, vbase), eh = __builtin_fmaf(a * tx_, tx_, hbase);
            const float qm = (vin && hin) ? 0.f : fminf(ev, eh);
            bool h = !(qm > bound);                                       // NaN-safe: a failed comparison keeps the entry
            h = h && ((uint32_t)(base + e) < L.blast[b0 + i]);            // entries at or beyond the block's deepest contributor never contribute there
            hits |= h ? (1u << i) : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const bool hit = (hits & (1u << i)) != 0u;
        const uint64_t m = __ballot(hit);
        uint32_t rk, len;
        if (BK == 64) {
            rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            len = (uint32_t)__popcll(m);
        } else {                               // two 32-entry halves per wave, each a different block pair
            const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
            const bool up = lane >= 32;
            rk = up ? __builtin_amdgcn_mbcnt_hi(hi, 0u) : __builtin_amdgcn_mbcnt_lo(lo, 0u);
            len = (uint32_t)__popc(up ? hi : lo);
        }
#ifdef TR_STATS
        if (i == 0 && hits) atomicOr(&L.live[e], 1u);
#endif
        if (hit) L.list[(rk + 1u) * 16u + (uint32_t)(b0 + i)] = (uint8_t)e;
        if (e == 0) L.cnt[b0 + i] = len;
    }
}

#ifdef TR_STATS
__device__ unsigned long long tr_stats[8];
#endif
template <bool ABSGRAD, bool LINEAGE, int BK>
__global__ void __launch_bounds__(RB, TR_MINW)
k_render_bwd_tr(ViewBg bg_arg /* MUST stay the first parameter: read through dvs_load_bg() */, int W, int H, int tiles_x, int tiles_per_view,
                int num_tiles /* = views * tiles_per_view */, const uint2* __restrict__ ranges,
                const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d,
                const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout /*[views,3,H,W]*/,
                float* __restrict__ grow /*[n,12], same row contract as k_render_bwd*/ DVS_DBG_PARAM,
                const uint32_t* __restrict__ live_splat /*k_render_fwd's compacted lists (entries that reach the tile), or null*/,
                const uint32_t* __restrict__ live_pos /*list position -> position in the compacted list*/) {
    __shared__ TrLds<BK> L;
    __shared__ __attribute__((aligned(16))) float s_tab[4][(BK + 1) * 12];   // per wave and batch entry: the 12-float row; row BK = sink of the dummy entry
    __shared__ __attribute__((aligned(16))) float s_tb[4][TR_SLOTS * TR_SS];  // per wave: slot, plane (v5 | w), phase-1 lane
    (void)bg_arg;
    const int dbg = DVS_DBG_VALUE;                          // release builds: 0, every `dbg &` test below folds away
    const int tile_g = tile_of_block(blockIdx.x, num_tiles);
    if (tile_g >= num_tiles) return;
    const int view = tile_g / tiles_per_view, tile = tile_g - view * tiles_per_view;
    const float3 bgv = dvs_load_bg(view);
    final_T += (size_t)view * W * H; n_contrib += (size_t)view * W * H; dL_dout += (size_t)view * 3 * W * H;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t P = (size_t)W * H;
    // phase-1 role: group g1 = lane >> 4 (one 16-lane row = one 4x4 block of the wave's 8x8 quadrant), pixel i = lane & 15
    const int g1 = lane >> 4, pi = lane & 15;
    const int bx1 = 2 * (wave & 1) + (g1 & 1), by1 = 2 * (wave >> 1) + (g1 >> 1), blk1 = by1 * 4 + bx1;
    const int px = tx * DVS_TILE + 4 * bx1 + (pi & 3), py = ty * DVS_TILE + 4 * by1 + (pi >> 2);
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile_g];
    // phase-2 role: pixel row r = lane >> 4, slot s = (lane >> 2) & 3, block g2 = lane & 3
    const int r2 = lane >> 4, s2 = (lane >> 2) & 3, g2 = lane & 3;
    const int bx2 = 2 * (wave & 1) + (g2 & 1), by2 = 2 * (wave >> 1) + (g2 >> 1);
    const int X0 = tx * DVS_TILE + 4 * bx2, Y0 = ty * DVS_TILE + 4 * by2 + r2;
    const float X0f = (float)X0, Y0f = (float)Y0;
    float d2[4][3];                                           // upstream gradient of the four pixels (X0 + x, Y0) this lane sums in phase 2
    // (measured: fetching them per round instead — three global loads, or twelve ds_bpermute from the phase-1 lanes — doubles the cost of a round)
    if ((W & 3) == 0) {                                       // the four pixels are one aligned 16-B word per channel: 3 loads instead of 12
        const bool in2 = X0 < W && Y0 < H;                    // (X0 and W are multiples of 4: all four pixels are inside, or none)
        const size_t q = (size_t)Y0 * W + X0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 v = in2 ? *reinterpret_cast<const float4*>(dL_dout + c * P + q) : make_float4(0.f, 0.f, 0.f, 0.f);
            d2[0][c] = v.x; d2[1][c] = v.y; d2[2][c] = v.z; d2[3][c] = v.w;
        }
    } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const bool in2 = X0 + x < W && Y0 < H;
            const size_t q = (size_t)Y0 * W + X0 + x;
#pragma unroll
            for (int c = 0; c < 3; ++c) d2[x][c] = in2 ? dL_dout[c * P + q] : 0.f;
        }
    }
    const int perm_r = (r2 == 1) ? 2 : (r2 == 2) ? 1 : r2;    // value index inside each packed register after the two swaps
    const int xaddr = (lane ^ 32) << 2;
    const int jaddr = (16 * g2) << 2;                         // any lane of phase-1 row g2 holds that group's slot -> entry map
    const uint32_t jshift = 8u * (uint32_t)(TR_SLOTS - 1 - s2);

    const float T_final = inside ? final_T[pix] : 0.f;
    uint32_t last = inside ? n_contrib[pix] : 0u;
    if (live_pos) {                                           // walk the forward's live list: same entries in the same order minus those that
        if (last > 0u) last = live_pos[range.x + last - 1u] + 1u;     // cannot reach the tile; a contributor is always on it
        sorted_splat = live_splat;
    }
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) { dLp0 = dL_dout[pix]; dLp1 = dL_dout[P + pix]; dLp2 = dL_dout[2 * P + pix]; }
    const float bg_dot = (bgv.x * dLp0 + bgv.y * dLp1) + bgv.z * dLp2;

    // deepest contributor per block (= per 16-lane row) and of the tile
    uint32_t bmax = last;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) bmax = max(bmax, (uint32_t)__shfl_xor((int)bmax, d, 64));
    if (pi == 0) L.blast[blk1] = bmax;
    for (int e = threadIdx.x; e < (BK + 1) * 12; e += RB) reinterpret_cast<float4*>(&s_tab[0][0])[e] = make_float4(0.f, 0.f, 0.f, 0.f);      // (4 tables x (BK + 1) x 12 floats = (BK + 1) x 12 16-B words)
    if (threadIdx.x < 16) L.list[threadIdx.x] = (uint8_t)BK;
#ifdef TR_STATS
    if (threadIdx.x < BK) L.live[threadIdx.x] = 0u;
#endif
    if (threadIdx.x == 0) { L.ea[BK] = make_float4(0.f, 0.f, 0.f, 0.f); L.eb[BK] = make_float4(0.f, 0.f, 0.f, 0.f); L.ec[BK] = make_float4(0.f, 0.f, 0.f, 0.f); }
    __syncthreads();
    uint32_t todo = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) todo = max(todo, L.blast[g]);
    if (todo == 0 || (dbg & 64)) return;

    float T = T_final;
    float D = T_final * bg_dot;          // see k_render_bwd: one scalar of "colour behind" state suffices
    const uint32_t tb_m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&s_tb[wave][0]);   // LDS byte offset of the wave's buffer (low half of the flat address), wave-uniform:
                                                                          // phase 1 writes (v5, w) of slot s, lane l at floats [TR_SS s + l], [TR_SS s + 64 + l]
    // LDS byte addresses kept as opaque 32-bit values (the low half of a flat LDS address is the LDS offset): the compiler otherwise
    // rebuilds them from two registers in every round
    typedef __attribute__((address_space(3))) float lds_float;
    typedef float tr_v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const tr_v4f lds_cv4f;
    uint32_t tbr_a = (uint32_t)(uintptr_t)&s_tb[wave][TR_SS * s2 + 16 * g2 + 4 * r2];   // phase 2 reads its row of four pixels: v5 at floats 0..3, w at 64..67
    uint32_t tabw_a = (uint32_t)(uintptr_t)&s_tab[wave][perm_r];
    asm("" : "+v"(tbr_a), "+v"(tabw_a));
    const uint8_t* const lbase = &L.list[0];

    // phase 2: the wave's TR_SLOTS x 4 (slot, block) pairs, four lanes (pixel rows) per pair
    auto flush = [&](uint32_t jpack) {
        if (dbg & 4) return;
        const uint32_t jp = (uint32_t)__builtin_amdgcn_ds_bpermute(jaddr, (int)jpack);
        const int j = (int)((jp >> jshift) & 0xffu);
        const tr_v4f V_ = *(lds_cv4f*)tbr_a, W_ = *(lds_cv4f*)(tbr_a + 256u);
        const float4 V = make_float4(V_.x, V_.y, V_.z, V_.w), Wv = make_float4(W_.x, W_.y, W_.z, W_.w);
        const float2 mean = *reinterpret_cast<const float2*>(&L.ea[j]);
        const float4 cq = L.ec[j];                                        // colour b | conic a, b, c
        asm("" : : "v"(cq.x));                                            // (keeps the read ONE ds_read_b128: without a use of .x it is split into
                                                                          //  two reads whose offsets need an extra address add)
        const float Dx = mean.x - X0f, Dy = mean.y - Y0f;                 // d = mean - pixel for the row's first pixel; pixel x: Dx - x
        if (dbg & 1024) {                                                 // timing only: a round without its arithmetic (what a matrix-pipe version could at best save)
            const float h0 = swap32_add(V.x, V.y), h1 = swap32_add(V.z, V.w), h2 = swap32_add(Wv.x, Wv.y), h3 = swap32_add(Wv.z, Wv.w);
            const float h4 = swap32_add(Dx, Dy), h5 = Dx + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(Dy)));
            const float q0 = swap16_add(h0, h1), q1 = swap16_add(h2, h3), q2 = swap16_add(h4, h5);
            uint32_t slot_a;
            asm("v_mad_u32_u24 %0, %1, 48, %2" : "=v"(slot_a) : "v"(j), "v"(tabw_a));
            lds_float* const slot = (lds_float*)slot_a;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g2 == g) { slot[0] += q0; slot[4] += q1; slot[8] += q2; }
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            }
            return;
        }
        const float A0 = (V.x + V.y) + (V.z + V.w);
        const float B0 = __builtin_fmaf(3.f, V.w, __builtin_fmaf(2.f, V.z, V.y));
        const float C0 = __builtin_fmaf(9.f, V.w, __builtin_fmaf(4.f, V.z, V.y));
        float v[12];
        v[0] = __builtin_fmaf(Dx, A0, -B0);                               // S_x  = sum v5 (Dx - x)
        v[1] = Dy * A0;                                                   // S_y
        v[2] = __builtin_fmaf(Dx, v[0], -__builtin_fmaf(Dx, B0, -C0));    // S_xx = sum v5 (Dx - x)^2 = Dx S_x - (Dx B - C)
        v[3] = Dy * v[0];                                                 // S_xy
        v[4] = Dy * v[1];                                                 // S_yy
        v[5] = A0;                                                        // S_o
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v[6 + c] = __builtin_fmaf(Wv.w, d2[3][c], __builtin_fmaf(Wv.z, d2[2][c], __builtin_fmaf(Wv.y, d2[1][c], Wv.x * d2[0][c])));
        if (ABSGRAD) {
            // |dL/dmean2D| per pixel = |v5| |(a dx + b dy, b dx + c dy)| (times the opacity, applied at publish); linear in x
            const float gx0 = __builtin_fmaf(cq.y, Dx, cq.z * Dy), gy0 = __builtin_fmaf(cq.z, Dx, cq.w * Dy);
            const float gx1 = gx0 - cq.y, gx2 = gx1 - cq.y, gx3 = gx2 - cq.y;
            const float gy1 = gy0 - cq.z, gy2 = gy1 - cq.z, gy3 = gy2 - cq.z;
            v[9] = __builtin_fmaf(fabsf(V.w), fabsf(gx3), __builtin_fmaf(fabsf(V.z), fabsf(gx2), __builtin_fmaf(fabsf(V.y), fabsf(gx1), fabsf(V.x) * fabsf(gx0))));
            v[10] = __builtin_fmaf(fabsf(V.w), fabsf(gy3), __builtin_fmaf(fabsf(V.z), fabsf(gy2), __builtin_fmaf(fabsf(V.y), fabsf(gy1), fabsf(V.x) * fabsf(gy0))));
        } else { v[9] = 0.f; v[10] = 0.f; }
        v[11] = 0.f;
        // fold the four pixel rows (the four 16-lane rows of the wave) with the packing swaps: afterwards row r holds, in q[k], the
        // total of value 4 k + perm(r) of its (block, slot) pair
        float q0, q1, q2;
        if (dbg & 4096) {                                                 // timing only: the round without its packing swaps
            q0 = (v[0] + v[1]) + (v[2] + v[3]); q1 = (v[4] + v[5]) + (v[6] + v[7]); q2 = (v[8] + v[9]) + v[10];
        } else {
        const float h0 = swap32_add(v[0], v[1]), h1 = swap32_add(v[2], v[3]), h2 = swap32_add(v[4], v[5]);
        const float h3 = swap32_add(v[6], v[7]);
        float h4, h5;
        if (ABSGRAD) {
            h4 = swap32_add(v[8], v[9]);
            h5 = v[10] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[10])));
        } else {
            h4 = v[8] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[8])));
            h5 = 0.f;
        }
        q0 = swap16_add(h0, h1); q1 = swap16_add(h2, h3); q2 = swap16_add(h4, h5);
        }
        // one block group at a time: two groups may hold the same entry (measured: in nearly every round); LDS operations of one wave
        // execute in order, so a later group sees an earlier group's write
        uint32_t slot_a;                                                  // = tabw_a + 48 j
        asm("v_mad_u32_u24 %0, %1, 48, %2" : "=v"(slot_a) : "v"(j), "v"(tabw_a));
        lds_float* const slot = (lds_float*)slot_a;
        if (dbg & 2048) { asm volatile("" : : "v"(q0), "v"(q1), "v"(q2), "v"(slot_a)); return; }     // timing only: no table update
        if (dbg & 8192) { slot[0] += q0; slot[4] += q1; slot[8] += q2; return; }                      // timing only: ONE read-add-write for all four groups (conflicts ignored)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g2 == g) { slot[0] += q0; slot[4] += q1; slot[8] += q2; }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
    };

    const int nbatch = (int)((todo + BK - 1) / BK);
    const float tile_x0 = (float)(tx * DVS_TILE), tile_y0 = (float)(ty * DVS_TILE);
    uint32_t id_stage = tr_load_id<BK>(sorted_splat, range.x + (nbatch - 1) * BK, min(BK, (int)todo - (nbatch - 1) * BK));
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * BK;
        const int cnt = min(BK, (int)todo - base);
        // (no barrier here: the publish of batch b + 1, which other threads may still be in, reads the tables and idop[(b + 1) & 1] only)
        tr_stage<BK>(L, splat2d, id_stage, cnt, base, b & 1, tile_x0, tile_y0 DVS_DBG_PASS(dbg));
        if (!(dbg & 256)) __syncthreads();                  // batch staged; the tables are zero again        (dbg 256: timing of a barrier-free batch loop — wrong results)
        const int len = (int)L.cnt[blk1];
        int nmax = len;
        nmax = max(nmax, __shfl_xor(nmax, 16, 64));
        nmax = max(nmax, __shfl_xor(nmax, 32, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
#ifdef TR_STATS                                             // tools/xbuild.sh stats -DTR_STATS: step / imbalance counters of a launch on stderr
        if (threadIdx.x == 0) {
            uint32_t sum = 0, wmx = 0, wsum = 0;
            for (int w = 0; w < 4; ++w) {                   // wave w owns the blocks (2 (w & 1) + {0, 1}, 2 (w >> 1) + {0, 1})
                uint32_t m = 0;
                for (int g = 0; g < 4; ++g) { const uint32_t c = L.cnt[(2 * (w >> 1) + (g >> 1)) * 4 + 2 * (w & 1) + (g & 1)]; m = max(m, c); sum += c; }
                wmx = max(wmx, m); wsum += m;
            }
            atomicAdd(&tr_stats[0], (unsigned long long)wsum);          // list steps summed over the waves
            atomicAdd(&tr_stats[1], (unsigned long long)(4 * wmx));     // the same if every wave took as many as the batch's slowest (what the barrier costs)
            atomicAdd(&tr_stats[2], (unsigned long long)sum);           // (block, entry) pairs
            atomicAdd(&tr_stats[3], 1ull);                              // batches
            atomicAdd(&tr_stats[4], (unsigned long long)cnt);           // staged entries
            uint32_t lv = 0;
            for (int e2 = 0; e2 < cnt; ++e2) lv += L.live[e2];
            atomicAdd(&tr_stats[5], (unsigned long long)lv);            // ... that reach at least one block of the tile
        }
        __syncthreads();
        if (threadIdx.x < BK) L.live[threadIdx.x] = 0u;
#endif
        const int lastb = (int)min(last, (uint32_t)(base + BK)) - base;          // entries of this batch below the pixel's last contributor
        uint32_t p = (uint32_t)len * 16u + (uint32_t)blk1;                         // byte offset of the list's last element
        uint32_t jn = lbase[p];                                                    // (32 bits behind an opaque copy: the compiler narrows the loop
        asm("" : "+v"(jn));                                                      //  variable to a byte otherwise and re-masks it every step)
        uint32_t jpack = 0;
        int nslot = 0;
#pragma unroll 1
        for (int it = 0; it < ((dbg & 8) ? 0 : nmax); ++it) {
            const int j = (int)jn;
            jpack = (jpack << 8) | (uint32_t)j;                                    // (here: every use of j ahead of the next element's load)
            const bool below_last = j < lastb;
            const float4 ea = L.ea[j];
            const float4 eb = L.eb[j];
            p = __builtin_elementwise_sub_sat(p, 16u);                             // an exhausted list parks on the sentinel row
            jn = lbase[p];
            asm("" : "+v"(jn));
            const float dx = ea.x - pxf, dy = ea.y - pyf;
            const float p2 = __builtin_fmaf(eb.x * dy, dy, __builtin_fmaf(ea.w, dy, ea.z * dx) * dx);   // same expression as the forward
            const float G = __builtin_amdgcn_exp2f(p2);
            const float oa = eb.y * G;
            const float alpha = fminf(DVS_ALPHA_MAX, oa);
            const bool contrib = below_last && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
#if TR_SKIP_EMPTY_STEPS
            if (__builtin_amdgcn_ballot_w64(contrib) == 0) continue;
#endif
            const float2 rg = make_float2(eb.z, eb.w);
            const float cb = L.ec[j].x;
            const float al = contrib ? alpha : 0.f;
            const float inv_1ma = __builtin_amdgcn_rcpf(1.f - al);
            T = T * inv_1ma;
            const float w = al * T;
            const float cd = (rg.x * dLp0 + rg.y * dLp1) + cb * dLp2;
            const float dL_dalpha = cd * T - D * inv_1ma;
            D = D + cd * w;
            // DVS_GRAD_TRUE: the 0.99 clamp blocks the gradient; DVS_GRAD_LINEAGE: it passes as if alpha = opacity * G
            const bool gate = LINEAGE ? contrib : (contrib && !(oa > DVS_ALPHA_MAX));
            const float v5 = gate ? G * dL_dalpha : 0.f;
            // the pair goes to slot nslot of the wave's buffer, lane-indexed: ds_write_addtid_b32 (address = M0 + offset + 4 lane) needs no
            // address register and no vector instruction for the slot offset (the scalar unit moves the slot base into M0)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256"
                         : : "v"(v5), "v"(w), "s"(tb_m0 + (uint32_t)(TR_SS * 4) * (uint32_t)nslot) : "memory", "m0");
            if (++nslot == TR_SLOTS) { flush(jpack); nslot = 0; }
        }
        if (nslot > 0 && !(dbg & 512)) {                    // pad the unfinished round with the dummy entry (zero pairs, sink row)   (dbg 512: timing without the padded rounds — wrong results)
            for (; nslot < TR_SLOTS; ++nslot) {
                float* const tbw = &s_tb[wave][lane];
                tbw[TR_SS * nslot] = 0.f; tbw[TR_SS * nslot + 64] = 0.f;
                jpack = (jpack << 8) | (uint32_t)BK;
            }
            flush(jpack);
        }
        if (!(dbg & 256)) __syncthreads();                  // tables complete; nobody reads the staged entries or lists any more
        if (b > 0) id_stage = tr_load_id<BK>(sorted_splat, range.x + (b - 1) * BK, BK);     // the next batch's ids arrive under the publish
        // the tile's total per touched (entry, value): ONE global atomic each — consecutive threads add consecutive floats of a row.
        // The moment and abs-grad sums were taken over v5 = G dL/dalpha; the row contract wants them over opacity * v5.
        for (int e = threadIdx.x; e < ((dbg & 128) ? 0 : cnt * 12); e += RB) {
            const float val = (s_tab[0][e] + s_tab[1][e]) + (s_tab[2][e] + s_tab[3][e]);
            if (val != 0.f) {
                const int ent = e / 12, comp = e - 12 * ent;
                const uint2 io = L.idop[b & 1][ent];
                const float sc = (comp == 5 || (comp >= 6 && comp <= 8)) ? 1.f : __uint_as_float(io.y);
                if (comp < 11 && !(dbg & 1)) atomicAdd(&grow[(size_t)io.x * 12 + comp], val * sc);
            }
            s_tab[0][e] = 0.f; s_tab[1][e] = 0.f; s_tab[2][e] = 0.f; s_tab[3][e] = 0.f;
        }
    }
}

// ---- launcher -------------------------------------------------------------------------------------------
hipError_t dvs_launch_render_bwd_tr(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                    const uint32_t* sorted_splat, const float* splat2d, const float* bgs, const float* final_T,
                                    const uint32_t* n_contrib, const float* dL_dout, float* grad_rows, int absgrad, int grad_mode,
                                    const uint32_t* live_splat, const uint32_t* live_pos) {
    const int tiles_pv = tiles_x * tiles_y, num_tiles = tiles_pv * n_views;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    const int lineage = grad_mode == 1 ? 1 : 0;
#ifdef DVS_EXPERIMENT
    static const int dbg = dvs_experiment_int("DVS_TR_DEBUG");      // ablation bits of tools/bwd_probe.py (timing only; experiment builds)
#endif
    const size_t extra_lds = dvs_experiment_extra_lds();
#define DVS_TR(A, LN, BKV)                                                                                                          \
    hipLaunchKernelGGL((k_render_bwd_tr<A, LN, BKV>), dim3(grid), dim3(RB), extra_lds, st, make_view_bg(n_views, bgs), W, H, tiles_x, tiles_pv, \
                       num_tiles, (const uint2*)ranges, sorted_splat, (const float4*)splat2d, final_T, n_contrib, dL_dout, grad_rows DVS_DBG_PASS(dbg), live_splat, live_pos)
#define DVS_TR_B(BKV)                                                                                   \
    do {                                                                                                \
        if (absgrad) { if (lineage) DVS_TR(true, true, BKV); else DVS_TR(true, false, BKV); }           \
        else { if (lineage) DVS_TR(false, true, BKV); else DVS_TR(false, false, BKV); }                 \
    } while (0)
    DVS_TR_B(64);
#ifdef TR_STATS
    {
        unsigned long long h[8];
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(tr_stats), sizeof(h));
        fprintf(stderr, "TR_STATS views %d: wave_steps %llu  if_lockstep %llu  block_pairs %llu  batches %llu  entries %llu  live_entries %llu\n", n_views, h[0], h[1], h[2], h[3], h[4], h[5]);
        memset(h, 0, sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(tr_stats), h, sizeof(h));
    }
#endif
#undef DVS_TR_B
#undef DVS_TR
    return hipGetLastError();
}
