// render_blocks.hip — A7 / A8 with per-4x4-block splat lists. The A8 kernel here was the default composite backward of round 2 (variant
// "blocks"; round 3's default, render_tr.hip, keeps its lists and tables and replaces its per-step reduction); the A7 kernel is a
// measured experiment (not faster than render.hip's, see the end of this header).
//
// The round-1 composite kernels walk, per 8x8 quadrant (= one wave), every splat whose alpha >= 1/255 ellipse reaches the quadrant;
// on the bench scene a visit has 21 of 64 lanes contributing (a footprint of ~40 px inside the tile against a 64-px quadrant), and
// the kernels are bound by vector-instruction issue, so two thirds of the issued lanes do nothing. Here the unit of culling is a
// 4x4-pixel block: while a batch of the tile's depth-ordered list is staged into LDS, every entry is tested (exactly) against the
// tile's sixteen blocks and the survivors are compacted, by ballot + mbcnt, into sixteen per-block index lists. A wave still owns an
// 8x8 quadrant, but its 64 lanes form FOUR interleaved groups of 16 (group = lane & 3 = one 4x4 block) and every group walks its own
// list with its own cursor: in one iteration the four groups evaluate four different splats. A wave needs max(list length of its
// four blocks) iterations instead of one per quadrant visit (0.74x the iterations at 44 % lane utilisation with batches of 64), and
// the scalar bit-walk of the round-1 forward is gone (the cursor is a vector register).
//
// Backward: the 12 per-pixel partials are still reduced across lanes, but only over a group (lanes with equal lane & 3): the two
// packing swaps (v_permlane32_swap, v_permlane16_swap) fold the four 16-lane rows, two ds_swizzle steps (xor 4, xor 8) finish —
// two butterfly levels fewer than the wave-wide tree, and one tree serves four (splat, block) pairs. The group totals then have to
// be merged per splat — across the groups of a wave, which may or may not hold the same entry in an iteration, and across the four
// waves. gfx950 has no cheap float merge primitive (measured below), so the merge stays inside the wave: each wave owns a table
// [batch entry][12 floats] in LDS and the four groups add their totals with plain ds_read / v_add / ds_write, one group after the
// other — the LDS executes the operations of a wave in program order, so two groups holding the same entry are safe without
// atomics. After the batch the four tables are summed and each touched (entry, value) is published with ONE global fp32 atomic:
// T x 11 atomics per view instead of (quadrant visits) x 11, about half the cross-XCD atomic traffic of the round-1 kernel.
// Batches of 64 entries keep the tables at 12 KB (8 workgroups per CU; with 128 the kernel loses 30 % to occupancy).
//
// Same inputs, same 48-B row contract (moments about the mean), same alpha rule and thresholds as render.hip; selected by
// dvs_set_backward_variant / dvs_set_forward_variant (DVS_*_BLOCKS). Reference anchors as in render.hip.
//
// MEASURED (C3, 1 MI355X, profiles/r02_variants.md):
//   backward 0.46-0.48 ms per view against 0.51 ms (round-1 kernel, same box); inside the 8-view launch 3.35 vs 3.90 ms.
//            Earlier forms of the merge: ds_add_f32 into one table per tile 0.83 ms (the LDS float atomic costs ~12 cycles per
//            active lane, tools/ubench/lds_atomic.hip: the LDS was busy 100 % of the kernel); global atomics per group 1.60 ms.
//            Ranking the sixteen lists by length per batch and dealing them to the waves in that order: -1.3 % instructions only.
//   forward  0.22 ms vs 0.18 ms: fewer iterations, but 39 instead of 28 vector instructions per iteration (per-lane cursor and
//            addresses) plus the sixteen block tests and the list compaction per staged entry (+25 M instructions): same total.
#include <cstdlib>
#include "dvs_device.h"
#include "dvs_kernels.h"
#include "render_common.h"

#ifndef BK_RB
#define BK_RB 64                        // list entries staged per batch (the four per-wave tables of A8 take 4 x BK_RB x 48 B)
#endif
#ifndef BK_STAGE_NT
// A8 staging threads. RB: four threads per entry, four blocks each. BK_RB: one thread per entry (the per-entry set-up is not repeated:
// 40 % fewer staging instructions), but the staging phase then runs in ONE wave — measured 1 % slower end to end (1185 vs 1195 views/s).
#define BK_STAGE_NT RB
#endif
#define BK_SW (BK_RB / 64)              // staging waves per block subset

struct __attribute__((aligned(16))) BlockLds {
    float4 xyc[BK_RB];            // mean x, mean y, cs.x, cs.y        (cs = exponent constants, see render.hip)
    float4 zoir[BK_RB];           // cs.z, opacity, colour b, colour r
    float4 cog[BK_RB];            // conic a, b, c, colour g
    uint32_t id[BK_RB];           // splat id (row of the gradient table)
    uint8_t list[16][BK_RB];      // per block: batch indices of the entries that can reach it, in list order
    uint32_t cnt[16];             // list lengths
    uint32_t wcnt[BK_SW][16];     // per staging wave
};

// Stage entries [first, first + cnt) and build the sixteen block lists. Thread t handles entry t % BK_RB and the blocks
// [BK_GPT * (t / BK_RB), +BK_GPT). `blast` (backward only): per block the deepest contributor of any of its pixels — entries at or
// beyond it can never contribute there and are left out of that block's list.
template <bool USE_LAST, int NT /*staging threads: RB, or BK_RB = one per entry (no per-entry set-up repeated by several threads)*/>
__device__ __forceinline__ void stage_blocks(BlockLds& L, const uint32_t* __restrict__ sorted_splat, uint32_t first, int cnt, int base,
                                             const float4* __restrict__ splat2d, float tile_x0, float tile_y0, const uint32_t* blast) {
    constexpr int GPT = 16 * BK_RB / NT, ROWS = GPT / 4;        // blocks / block rows per staging thread
    const int t = threadIdx.x, e = t % BK_RB, sub = t / BK_RB, lane = t & 63, ws = e >> 6;
    const bool stager = t < NT;
    uint32_t hits = 0;                    // bit i: block GPT * sub + i
    if (stager && e < cnt) {
        const uint32_t id = sorted_splat[first + e];
        const float4 r0 = splat2d[4 * (size_t)id], r1 = splat2d[4 * (size_t)id + 1];
        const float bl = splat2d[4 * (size_t)id + 2].x;
        const float a = r0.z, b = r0.w, c = r1.x, op = r1.y;
        if (sub == 0) {
            L.xyc[e] = make_float4(r0.x, r0.y, -0.72134752044448170f * a, -1.4426950408889634f * b);
            L.zoir[e] = make_float4(-0.72134752044448170f * c, op, bl, r1.z);
            L.cog[e] = make_float4(a, b, c, r1.w);
            L.id[e] = id;
        }
        // alpha >= 1/255 only where q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 o). The minimum of the convex form over a block's
        // pixel rectangle is 0 if the mean lies inside, otherwise it lies on an edge FACING the mean: at most one vertical edge (the
        // nearer one, when the mean is left or right of the block's column) and one horizontal edge. On the vertical line x,
        //   q(x, y) = c (y - y*)^2 + x^2 det / c,  y* = -b x / c,   so the edge minimum is at y = clamp(y*, y0, y1);
        // the line terms depend only on the block column (row for horizontal edges) and are computed once per entry.
        // The bound is inflated so that the exact per-pixel test — unchanged — decides every contribution.
        // (v_log_f32 / v_rcp_f32: 1-ulp errors are far inside the slack. det = a c - b^2 cancels for thin diagonal splats, so it is
        // lowered by its own rounding bound: every term below is non-negative and the minimum is never over-estimated.)
        const float bound = 1.3862943611f * __builtin_amdgcn_logf(255.0f * op) * 1.0001f + 1e-3f;       // 2 ln 2 log2(255 o)
        const float det = fmaxf(0.f, __builtin_fmaf(-2.4e-7f, a * c, a * c - b * b));
        const float rc = __builtin_amdgcn_rcpf(c), ra = __builtin_amdgcn_rcpf(a);
        const float det_c = det * rc, det_a = det * ra, nb_c = -b * rc, nb_a = -b * ra;
        const float ox = tile_x0 - r0.x, oy = tile_y0 - r0.y;
        // this thread's blocks: rows ROWS * sub + r (r < ROWS), all four columns — compile-time indices only
        float vy[4], vbase[4], hx[ROWS], hbase[ROWS];      // facing vertical edge per column / horizontal edge per row: 1-D
        bool vin[4], hin[ROWS];                               // minimiser and x^2 det/c (+inf: the mean lies inside that column/row)
        const float oyr = oy + (float)(4 * ROWS * sub);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0 = ox + 4.f * (float)i, x1 = x0 + 3.f;
            vin[i] = x0 <= 0.f && x1 >= 0.f;
            const float xe = x0 > 0.f ? x0 : x1;
            vy[i] = nb_c * xe; vbase[i] = vin[i] ? __builtin_inff() : xe * xe * det_c;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float y0 = oyr + 4.f * (float)r, y1 = y0 + 3.f;
            hin[r] = y0 <= 0.f && y1 >= 0.f;
            const float ye = y0 > 0.f ? y0 : y1;
            hx[r] = nb_a * ye; hbase[r] = hin[r] ? __builtin_inff() : ye * ye * det_a;
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int col = i & 3, r = i >> 2;
            const float x0 = ox + 4.f * (float)col, x1 = x0 + 3.f, y0 = oyr + 4.f * (float)r, y1 = y0 + 3.f;
            const float ty_ = fminf(fmaxf(vy[col], y0), y1) - vy[col], tx_ = fminf(fmaxf(hx[r], x0), x1) - hx[r];
            const float ev = __builtin_fmaf(c * ty_, ty_, vbase[col]), eh = __builtin_fmaf(a * tx_, tx_, hbase[r]);
            const float qm = (vin[col] && hin[r]) ? 0.f : fminf(ev, eh);
            bool h = !(qm > bound);                                       // NaN-safe: a failed comparison keeps the entry
            if (USE_LAST) h = h && ((uint32_t)(base + e) < blast[GPT * sub + i]);
            hits |= h ? (1u << i) : 0u;
        }
    }
    if (NT == 64 && BK_RB == 64) {        // one staging wave: ranks and list lengths come straight from its ballots, no second phase
        if (stager) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint64_t m = __ballot((hits >> i) & 1u);
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((hits >> i) & 1u) L.list[i][rk] = (uint8_t)e;
                if (lane == 0) L.cnt[i] = (uint32_t)__popcll(m);
            }
        }
        return;                           // (the caller's barrier publishes the lists)
    }
    uint32_t rank[GPT];
    if (stager) {                         // (whole waves: NT is a multiple of 64)
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const uint64_t m = __ballot((hits >> i) & 1u);
            rank[i] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0) L.wcnt[ws][GPT * sub + i] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    if (stager) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int g = GPT * sub + i;
            uint32_t off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < BK_SW; ++w) { const uint32_t v = L.wcnt[w][g]; if (w < ws) off += v; tot += v; }
            if ((hits >> i) & 1u) L.list[g][off + rank[i]] = (uint8_t)e;
            if (e == 0) L.cnt[g] = tot;
        }
    }
}

// pixel of lane l of wave w: group = l & 3 = one of the quadrant's four 4x4 blocks, i = l >> 2 = pixel inside the block
__device__ __forceinline__ void lane_pixel(int wave, int lane, int& block, int& lx, int& ly) {
    const int gl = lane & 3, i = lane >> 2;
    const int bx = 2 * (wave & 1) + (gl & 1), by = 2 * (wave >> 1) + (gl >> 1);
    block = by * 4 + bx;
    lx = 4 * bx + (i & 3);
    ly = 4 * by + (i >> 2);
}
// longest list among the wave's four blocks (lanes 0..3 hold one block each)
__device__ __forceinline__ int wave_max4(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 1), c = __builtin_amdgcn_readlane(v, 2),
              d = __builtin_amdgcn_readlane(v, 3);
    return max(max(a, b), max(c, d));
}

// (the per-block forward is an experiment of round 2 — same image bits as k_render_fwd, not faster: experiment builds only)
#ifdef DVS_EXPERIMENT
// ---- A7 -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RB)
k_render_fwd_blocks(int W, int H, int tiles_x, int num_tiles, const uint2* __restrict__ ranges,
                    const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d, float bg0, float bg1, float bg2,
                    float* __restrict__ out_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ BlockLds L;
    const int tile = tile_of_block(blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int block, lx, ly;
    lane_pixel(wave, lane, block, lx, ly);
    const int px = tx * DVS_TILE + lx, py = ty * DVS_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;
    const uint8_t* lp = &L.list[block][0];

    for (int base = 0; base < total; base += BK_RB) {
        if (__syncthreads_and(done)) break;
        const int cnt = min(BK_RB, total - base);
        stage_blocks<false, RB>(L, sorted_splat, range.x + base, cnt, base, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE), nullptr);
        __syncthreads();
        if (__all(done)) continue;
        const int len = (int)L.cnt[block];
        const int nmax = wave_max4(len);
        int jn = len > 0 ? (int)lp[0] : 0;
#pragma unroll 1
        for (int it = 0; it < nmax; ++it) {
            const bool act = it < len;
            const int j = jn;
            jn = it + 1 < len ? (int)lp[it + 1] : 0;                       // next cursor value, requested one iteration ahead
            const float4 xy = L.xyc[j];
            const float4 zo = L.zoir[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float p2 = __builtin_fmaf(zo.x * dy, dy, __builtin_fmaf(xy.w, dy, xy.z * dx) * dx);
            const float alpha = fminf(DVS_ALPHA_MAX, zo.y * __builtin_amdgcn_exp2f(p2));
            const bool valid = act && !done && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
            const float aT = alpha * T;
            const float test_T = T - aT;
            const bool stop = valid && (test_T < DVS_T_STOP);
            const bool take = valid && !stop;
            done = done || stop;
            const float w = take ? aT : 0.f;
            C0 = __builtin_fmaf(zo.w, w, C0); C1 = __builtin_fmaf(L.cog[j].w, w, C1); C2 = __builtin_fmaf(zo.z, w, C2);
            T = T - w;
            last = take ? (uint32_t)(base + j + 1) : last;
            if ((it & 15) == 15 && __all(done)) break;
        }
    }
    if (inside) {
        const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg0;
        out_color[P + pix] = C1 + T * bg1;
        out_color[2 * P + pix] = C2 + T * bg2;
    }
}

// ---- A8 -------------------------------------------------------------------------------------------
// 12 per-lane partials -> per-GROUP totals (group = lanes with equal lane & 3). As wave_reduce12 (render_common.h) but the
// butterfly stops after the column bits 2 and 3: q[k] holds, in lane (row r, column c), the total over the lanes of group c & 3 of
// value index  q[0]: v0,v2,v1,v3   q[1]: v4,v6,v5,v7   q[2]: v8,v10,v9,v11  (by row r).
#endif  // DVS_EXPERIMENT

template <int NV>
__device__ __forceinline__ void group_reduce12(const float v[12], float q[3], int xaddr) {
    const float h0 = swap32_add(v[0], v[1]), h1 = swap32_add(v[2], v[3]), h2 = swap32_add(v[4], v[5]);
    const float h3 = swap32_add(v[6], v[7]);
    float h4, h5;
    if (NV == 11) {
        h4 = swap32_add(v[8], v[9]);
        h5 = v[10] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[10])));
    } else {
        h4 = v[8] + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v[8])));
        h5 = 0.f;
    }
    float r[3] = {swap16_add(h0, h1), swap16_add(h2, h3), swap16_add(h4, h5)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = r[k];
        x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (4 << 10) | 0x1f));     // xor 4
        x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (8 << 10) | 0x1f));     // xor 8
        q[k] = x;
    }
}

template <bool ABSGRAD>
__global__ void __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_render_bwd_blocks(ViewBg bg_arg /* MUST stay the first parameter: read through dvs_load_bg() */, int W, int H, int tiles_x, int tiles_per_view,
                    int num_tiles /* = views * tiles_per_view */, const uint2* __restrict__ ranges,
                    const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d,
                    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout /*[views,3,H,W]*/,
                    float* __restrict__ grow /*[n,12], same row contract as k_render_bwd*/, int lineage) {
    __shared__ BlockLds L;
    __shared__ float s_tab[4][BK_RB * 12];                  // per wave and batch entry: the 12-float row, summed over the wave's blocks
    __shared__ uint32_t s_blast[16];
    (void)bg_arg;
    const int tile_g = tile_of_block(blockIdx.x, num_tiles);
    if (tile_g >= num_tiles) return;
    const int view = tile_g / tiles_per_view, tile = tile_g - view * tiles_per_view;
    const float3 bgv = dvs_load_bg(view);
    final_T += (size_t)view * W * H; n_contrib += (size_t)view * W * H; dL_dout += (size_t)view * 3 * W * H;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int block, lx, ly;
    lane_pixel(wave, lane, block, lx, ly);
    const int px = tx * DVS_TILE + lx, py = ty * DVS_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile_g];
    const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
    // which group total this lane publishes: column c = lane & 15 carries group c & 3; its four copies (c >> 2 = 0..3) take one
    // register each (the fourth idles); row r selects the value index inside the register (see group_reduce12)
    const int lrow = lane >> 4, lsel = (lane & 15) >> 2;
    const int kv = lsel * 4 + ((lrow == 1) ? 2 : (lrow == 2) ? 1 : lrow);
    const bool publisher = lsel < 3 && kv < (ABSGRAD ? 11 : 9);
    const int xaddr = (lane ^ 32) << 2;
    const int mygroup = lane & 3;
    float* const tab = &s_tab[wave][kv];

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) { dLp0 = dL_dout[pix]; dLp1 = dL_dout[P + pix]; dLp2 = dL_dout[2 * P + pix]; }
    const float bg_dot = (bgv.x * dLp0 + bgv.y * dLp1) + bgv.z * dLp2;

    // deepest contributor per block (lanes with equal lane & 3) and of the tile
    uint32_t bmax = last;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) bmax = max(bmax, (uint32_t)__shfl_xor((int)bmax, d, 64));
    if (lane < 4) s_blast[block] = bmax;
    for (int e = threadIdx.x; e < 4 * BK_RB * 12; e += RB) (&s_tab[0][0])[e] = 0.f;
    __syncthreads();
    uint32_t todo = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) todo = max(todo, s_blast[g]);
    if (todo == 0) return;

    float T = T_final;
    float D = T_final * bg_dot;          // see k_render_bwd: one scalar of "colour behind" state suffices
    const uint8_t* lp = &L.list[block][0];
    const int nbatch = (int)((todo + BK_RB - 1) / BK_RB);
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * BK_RB;
        const int cnt = min(BK_RB, (int)todo - base);
        __syncthreads();                                    // previous batch published and consumed
        stage_blocks<true, BK_STAGE_NT>(L, sorted_splat, range.x + base, cnt, base, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE), s_blast);
        __syncthreads();
        const int len = (int)L.cnt[block];
        const int nmax = __builtin_amdgcn_readfirstlane(wave_max4(len));
        const int lastb = (int)min(last, (uint32_t)(base + BK_RB)) - base;         // entries of this batch below the pixel's last contributor
        int idx = len - 1;
        int jn = idx >= 0 ? (int)lp[idx] : 0;
#pragma unroll 1
        for (int it = 0; it < nmax; ++it, --idx) {
            const bool act = idx >= 0;
            const int j = jn;
            jn = idx >= 1 ? (int)lp[idx - 1] : 0;
            const float4 xy = L.xyc[j];
            const float4 zo4 = L.zoir[j];
            const float2 zo2 = make_float2(zo4.x, zo4.y);
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float p2 = __builtin_fmaf(zo2.x * dy, dy, __builtin_fmaf(xy.w, dy, xy.z * dx) * dx);   // same expression as the forward
            const float G = __builtin_amdgcn_exp2f(p2);
            const float oa = zo2.y * G;
            const float alpha = fminf(DVS_ALPHA_MAX, oa);
            const bool contrib = act && (j < lastb) && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
            if (__builtin_amdgcn_ballot_w64(contrib) == 0) continue;
            const float4 cg = L.cog[j];
            const float3 c = make_float3(zo4.w, cg.w, zo4.z);
            const float al = contrib ? alpha : 0.f;
            const float inv_1ma = __builtin_amdgcn_rcpf(1.f - al);
            T = T * inv_1ma;
            const float w = al * T;
            const float cd = (c.x * dLp0 + c.y * dLp1) + c.z * dLp2;
            float dL_dalpha = cd * T - D * inv_1ma;
            D = D + cd * w;
            // DVS_GRAD_TRUE: the 0.99 clamp blocks the gradient; DVS_GRAD_LINEAGE: it passes as if alpha = opacity * G
            const bool capped = !lineage && (oa > DVS_ALPHA_MAX);
            dL_dalpha = (contrib && !capped) ? dL_dalpha : 0.f;
            const float v5 = G * dL_dalpha;
            const float sw = zo2.y * v5;
            const float su = sw * dx, st = sw * dy;
            float v[12];
            v[0] = su; v[1] = st;
            v[2] = su * dx; v[3] = su * dy; v[4] = st * dy;
            v[5] = v5;
            v[6] = w * dLp0; v[7] = w * dLp1; v[8] = w * dLp2;
            v[9] = ABSGRAD ? fabsf(__builtin_fmaf(cg.x, su, cg.y * st)) : 0.f;
            v[10] = ABSGRAD ? fabsf(__builtin_fmaf(cg.z, st, cg.y * su)) : 0.f;
            v[11] = 0.f;
            float q[3];
            group_reduce12<ABSGRAD ? 11 : 9>(v, q, xaddr);
            const float val = lsel == 0 ? q[0] : (lsel == 1 ? q[1] : q[2]);
            // The four groups hold (possibly equal) entries j: one group at a time reads, adds and writes its row of the wave's
            // table. LDS operations of one wave execute in order, so a later group sees an earlier group's write — no float atomics
            // (ds_add_f32 costs ~12 cycles per lane on gfx950). A group whose list is exhausted holds a stale entry: it stays out.
            float* const slot = reinterpret_cast<float*>(reinterpret_cast<char*>(tab) + __umul24((unsigned)j, 48u));
            const bool pub = publisher && act;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (pub && mygroup == g) *slot += val;
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        // the tile's total per touched (entry, value): ONE global atomic each — consecutive threads add consecutive floats of a row
        for (int e = threadIdx.x; e < cnt * 12; e += RB) {
            const float val = (s_tab[0][e] + s_tab[1][e]) + (s_tab[2][e] + s_tab[3][e]);
            if (val != 0.f) {
                const int ent = e / 12, comp = e - 12 * ent;
                atomicAdd(&grow[(size_t)L.id[ent] * 12 + comp], val);
            }
            s_tab[0][e] = 0.f; s_tab[1][e] = 0.f; s_tab[2][e] = 0.f; s_tab[3][e] = 0.f;
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
#ifdef DVS_EXPERIMENT
hipError_t dvs_launch_render_fwd_blocks(hipStream_t st, int W, int H, int tiles_x, int tiles_y, const uint32_t* ranges,
                                        const uint32_t* sorted_splat, const float* splat2d, const float bg[3], float* out_color,
                                        float* final_T, uint32_t* n_contrib) {
    const int num_tiles = tiles_x * tiles_y;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    hipLaunchKernelGGL(k_render_fwd_blocks, dim3(grid), dim3(RB), 0, st, W, H, tiles_x, num_tiles, (const uint2*)ranges, sorted_splat,
                       (const float4*)splat2d, bg[0], bg[1], bg[2], out_color, final_T, n_contrib);
    return hipGetLastError();
}
#endif  // DVS_EXPERIMENT

hipError_t dvs_launch_render_bwd_blocks(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                        const uint32_t* sorted_splat, const float* splat2d, const float* bgs, const float* final_T,
                                        const uint32_t* n_contrib, const float* dL_dout, float* grad_rows, int absgrad, int grad_mode) {
    const int tiles_pv = tiles_x * tiles_y, num_tiles = tiles_pv * n_views;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    const int lineage = grad_mode == 1 ? 1 : 0;
    const size_t extra_lds = dvs_experiment_extra_lds();
#define DVS_RBB(KERNEL)                                                                                                             \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(RB), extra_lds, st, make_view_bg(n_views, bgs), W, H, tiles_x, tiles_pv, num_tiles, \
                       (const uint2*)ranges, sorted_splat, (const float4*)splat2d, final_T, n_contrib, dL_dout, grad_rows, lineage)
    if (absgrad) DVS_RBB(k_render_bwd_blocks<true>); else DVS_RBB(k_render_bwd_blocks<false>);
#undef DVS_RBB
    return hipGetLastError();
}
