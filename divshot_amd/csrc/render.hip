// render.hip — A7 alpha-composite forward and A8 alpha-composite backward for gfx950.
//
// One 16x16 tile per 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel quadrant so a
// splat's footprint can be rejected per wave. The tile's depth-ordered splat list is streamed
// through LDS in batches of 256 (one gathered splat per lane, then broadcast reads). The forward (k_render_fwd) keeps the pixel predicates
// in scalar wave masks and runs the updates of a contributing pixel under EXEC = take (21 vector + 12 scalar instructions per visit), and
// writes per-tile live lists — the entries that reach the tile at all — for the default backward (render_tr.hip).
// The backward in THIS file (variant "reduce", round 1) replays the list back-to-front, reduces the 9 (11 with abs-grad) per-splat partials over
// the 64 lanes with v_permlane32/16_swap + ds_swizzle butterflies (LDS crossbar, no LDS memory) and publishes them with ONE
// hardware fp32 atomic instruction per (wave, splat): 11 lanes add 11 consecutive floats of the splat's
// 48-byte gradient row (global_atomic_add_f32; built with -munsafe-fp-atomics, no CAS loop).
//
// Reference anchors (fenghuayumo/DIVSHOT): alpha rule and thresholds gsplat_ps.hlsl:60-65 (the viewer
// caps alpha at 0.999; the trainer constant fixed by this build is 0.99, SURVEY.md §8(a) A-notes),
// blend order renderer/gaussian.cpp:440, 16x16 groups gaussian_common.hlsl:162-163, abs-grad flag
// application/diverseshot-cli/source/main.cpp:44.
#include <cstdlib>
#include "dvs_device.h"
#include "dvs_kernels.h"
#include "render_common.h"

// ---- batch staging + per-quadrant culling masks -------------------------------------------------------
// Lane t of the workgroup gathers splat t of the batch into LDS and tests the ellipse on which the splat reaches
// alpha = 1/255 ( d^T Q d = 2 ln(255 o), Q = conic ) against the four 8x8 quadrants of the tile (exact
// ellipse-rectangle test). One ballot per quadrant turns the tests into 64-bit wave masks: the wave that
// owns quadrant q later walks only the set bits (s_ff1 / s_flbit, scalar unit) and never spends a vector
// instruction on a splat that cannot touch its pixels. The bound is inflated (1e-4 rel + 1e-3) and the rectangle minimum is
// computed from non-negative terms only (see stage_batch), so the exact per-pixel alpha test — unchanged — decides every
// contribution: results are identical to a full walk.
struct __attribute__((aligned(16))) BatchLds {
    // Four 16-B-stride arrays (one address VGPR serves every read of a visit), packed so that each read is a full
    // ds_read_b128 or a ds_read_b32 (a 12-byte ds_read_b96 costs twice the LDS cycles of a b128):
    // the alpha test of a visit needs x, y, cs.xyz, o = one b128 + one b64; the rest is read only by contributing visits.
    // cs = (-0.5 log2e a, -log2e b, -0.5 log2e c): exp(power) = exp2(cs.x dx^2 + cs.y dx dy + cs.z dy^2)
    float4 xyc[RB];     // mean x, mean y, cs.x, cs.y
    float4 zoir[RB];    // cs.z, opacity | splat id bits, colour r
    float4 cog[RB];     // conic a, b, c as preprocess wrote them (gradient formulas), colour g
    float4 bl[RB];      // colour b in .x
    uint64_t qmask[RB / 64][4];
};

// A7's own layout: everything a visit reads sits at ONE address in three arrays — two ds_read_b128 and one ds_read_b32.
struct __attribute__((aligned(16))) FwdLds {
    float4 a[RB];       // mean x, mean y, cs.x, cs.y
    float4 b[RB];       // cs.z, opacity, colour r, colour g
    float4 c[RB];       // colour b in .x (16-B stride as the other two: one address register serves the three reads)
    uint64_t qmask[RB / 64][4];
    uint32_t wlive[RB / 64];    // entries of the batch that reach the tile at all, per wave (see the live lists in k_render_fwd)
};
__device__ __forceinline__ void lds_put(BatchLds& L, int t, float2 xy, float4 co, float3 col, uint32_t id) {
    L.xyc[t] = make_float4(xy.x, xy.y, -0.72134752044448170f * co.x, -1.4426950408889634f * co.y);
    L.zoir[t] = make_float4(-0.72134752044448170f * co.z, co.w, __uint_as_float(id), col.x);
    L.cog[t] = make_float4(co.x, co.y, co.z, col.y);
    L.bl[t].x = col.z;
}
__device__ __forceinline__ void lds_put(FwdLds& L, int t, float2 xy, float4 co, float3 col, uint32_t) {
    L.a[t] = make_float4(xy.x, xy.y, -0.72134752044448170f * co.x, -1.4426950408889634f * co.y);
    L.b[t] = make_float4(-0.72134752044448170f * co.z, co.w, col.x, col.y);
    L.c[t].x = col.z;
}

// Returns the lane's entry: its splat id in `id_out` and whether it reaches any quadrant of the tile.
template <class LDS>
__device__ __forceinline__ bool stage_batch(LDS& L, const uint32_t* __restrict__ sorted_splat, uint32_t first, int cnt,
                                            const float4* __restrict__ splat2d, float tile_x0, float tile_y0, uint32_t& id_out) {
    const int t = threadIdx.x;
    uint32_t qm = 0;
    id_out = 0u;
    if (t < cnt) {
        const uint32_t id = sorted_splat[first + t];
        id_out = id;
        // the splat's 64-B record (dvs_fwd_state.splat2d): four 16-B loads from ONE cache line
        const float4 r0 = splat2d[4 * (size_t)id], r1 = splat2d[4 * (size_t)id + 1];
        const float4 r2 = splat2d[4 * (size_t)id + 2], r3 = splat2d[4 * (size_t)id + 3];     // colour b | .. | DVS_S2D_CULL constants
        const float bl = r2.x;
        const float2 xy = make_float2(r0.x, r0.y);
        const float4 co = make_float4(r0.z, r0.w, r1.x, r1.y);
        const float3 col = make_float3(r1.z, r1.w, bl);
        lds_put(L, t, xy, co, col, id);
        // The splat can reach alpha >= 1/255 only where q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 o), d = pixel - mean.
        // The minimum of the convex form over a quadrant's pixel rectangle is 0 if the mean lies inside, otherwise it lies on an edge
        // FACING the mean: the nearer vertical edge when the mean is left / right of the rectangle's columns, the nearer horizontal
        // edge when it is above / below. On the vertical line x:  q(x, y) = c (y - y*)^2 + x^2 det / c,  y* = -b x / c  (minimum at
        // y = clamp(y*, y0, y1)); every term is non-negative, so — unlike a x^2 + 2 b x y + c y^2 evaluated directly — nothing cancels
        // for thin, far-away splats, and det = a c - b^2 (which does cancel for them) is lowered by its own rounding bound: the
        // minimum is never over-estimated and the exact per-pixel test decides every contribution.
        // bound = 2 ln(255 o) inflated, det = a c - b^2 lowered by its rounding bound, det / c, det / a, -b / c, -b / a: per splat, from A2
        const float bound = r2.w;
        const float a = co.x, c = co.z;
        const float det_c = r3.x, det_a = r3.y, nb_c = r3.z, nb_a = r3.w;
        const float ox = tile_x0 - xy.x, oy = tile_y0 - xy.y;                 // tile origin relative to the mean
        float vy[2], vbase[2], hx[2], hbase[2];
        bool vin[2], hin[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x0 = ox + 8.f * (float)i, x1 = x0 + 7.f, y0 = oy + 8.f * (float)i, y1 = y0 + 7.f;
            vin[i] = x0 <= 0.f && x1 >= 0.f; hin[i] = y0 <= 0.f && y1 >= 0.f;
            const float xe = x0 > 0.f ? x0 : x1, ye = y0 > 0.f ? y0 : y1;
            vy[i] = nb_c * xe; vbase[i] = vin[i] ? __builtin_inff() : xe * xe * det_c;
            hx[i] = nb_a * ye; hbase[i] = hin[i] ? __builtin_inff() : ye * ye * det_a;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = q & 1, row = q >> 1;
            const float x0 = ox + 8.f * (float)col, x1 = x0 + 7.f, y0 = oy + 8.f * (float)row, y1 = y0 + 7.f;
            const float ty_ = fminf(fmaxf(vy[col], y0), y1) - vy[col], tx_ = fminf(fmaxf(hx[row], x0), x1) - hx[row];
            const float ev = __builtin_fmaf(c * ty_, ty_, vbase[col]), eh = __builtin_fmaf(a * tx_, tx_, hbase[row]);
            const float qmn = (vin[col] && hin[row]) ? 0.f : fminf(ev, eh);
            qm |= !(qmn > bound) ? (1u << q) : 0u;                            // NaN-safe: a failed comparison keeps the splat
        }
    }
    const uint64_t m0 = __ballot(qm & 1u), m1 = __ballot(qm & 2u), m2 = __ballot(qm & 4u), m3 = __ballot(qm & 8u);
    if ((t & 63) == 0) {
        uint64_t* dst = L.qmask[t >> 6];
        dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3;
    }
    return qm != 0u;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

#ifndef FWD_CMPX
#define FWD_CMPX 1          // the visit's predicates as a v_cmpx chain: 9 instead of 12 scalar instructions per visit (round 4, same-box A/B: step
                            // 5.216 -> 5.190 ms, +0.5 % views/s; bit-identical images); 0 = the round-3 form with scalar mask arithmetic
#endif
// ---- A7 -------------------------------------------------------------------------------------------
// RECORD (dvs_debug_record_decisions, parity tests only): additionally stores, per (list position, 8x8 quadrant), the 64-bit mask of the
// pixels that TOOK the entry — the kernel's own threshold decisions — so that the fp64 oracle can be run on exactly these decisions
// (tests/test_gpu_parity.py: no "tainted" carve-out). Same arithmetic, same images.
template <bool RECORD>
__global__ void __launch_bounds__(RB)
k_render_fwd(ViewBg bg_arg /* MUST stay the first parameter: read through dvs_load_bg() */, int W, int H, int tiles_x, int tiles_per_view,
             int num_tiles /* = views * tiles_per_view */, const uint2* __restrict__ ranges, uint2* __restrict__ ranges_out /*null, or: `ranges` is the
             encoded form the tile sort's last pass leaves, and the (start, end) pairs go here*/,
             const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d,
             float* __restrict__ out_color /*[views,3,H,W]*/, float* __restrict__ final_T /*[views,H,W]*/, uint32_t* __restrict__ n_contrib,
             uint32_t* __restrict__ live_splat /*[T] per tile, from ranges[tile].x: the entries that reach the tile, in list order*/,
             uint32_t* __restrict__ live_pos /*[T] per list position: how many entries before it (in its tile) reach the tile*/,
             uint64_t* __restrict__ take_masks /*RECORD: [capacity][4], zeroed by the caller*/, uint64_t take_cap
             DVS_DBG_PARAM /*experiment builds: 1 = stage the batches but skip the walk*/) {
    __shared__ FwdLds L;
    (void)bg_arg;
    const int dbg = DVS_DBG_VALUE;
    const int tile_g = tile_of_block(blockIdx.x, num_tiles);
    if (tile_g >= num_tiles) return;
    const int view = tile_g / tiles_per_view, tile = tile_g - view * tiles_per_view;
    const float3 bgv = dvs_load_bg(view);
    const float bg0 = bgv.x, bg1 = bgv.y, bg2 = bgv.z;
    out_color += (size_t)view * 3 * W * H; final_T += (size_t)view * W * H; n_contrib += (size_t)view * W * H;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tx * DVS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DVS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    uint2 range = ranges[tile_g];
    if (ranges_out) {
        // A6 is fused into the tile sort's last pass (frontend.hip k_seg_scatter): the entry arrives as (~start, end), (0, 0) = no
        // instance. Decode it and leave the canonical pair — in its own array: no barrier between the waves' reads and this store — for
        // A8 and the exported state.
        range.x = range.y ? ~range.x : 0u;
        if (threadIdx.x == 0) ranges_out[tile_g] = range;
    }
    const int total = (int)(range.y - range.x);
    // Pixel state that decides control flow lives in wave masks (scalar registers): `notdone` = pixels still compositing. A visit forms
    // its predicates with three compares into masks, the scalar unit combines them, and the five updates of a contributing pixel
    // (three colour FMAs, T, last contributor) run under EXEC = `take` — no v_cndmask per visit, no branch either.
    uint64_t notdone = __builtin_amdgcn_ballot_w64(inside);
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;
    // Live lists for the composite backward: 28 % of a C3 tile list are entries whose alpha >= 1/255 ellipse misses the tile (the 3-sigma
    // rectangle of A4 is wider). The staging test below already knows them; the entries that do reach the tile are written out compacted
    // (same order), with the map list position -> live position, so that A8 stages, tabulates and publishes 28 % fewer entries.
    uint32_t live_base = 0;

    for (int base = 0; base < total; base += RB) {
        if (__syncthreads_and(notdone == 0ull)) break;
        const int cnt = min(RB, total - base);
        uint32_t my_id;
        const bool live = stage_batch(L, sorted_splat, range.x + base, cnt, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE), my_id);
        const uint64_t lm = __ballot(live);
        const uint32_t lrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
        if (lane == 0) L.wlive[wave] = (uint32_t)__popcll(lm);
        __syncthreads();
        if (live_pos) {
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < RB / 64; ++w) { const uint32_t c = L.wlive[w]; pre += w < wave ? c : 0u; tot += c; }
            const uint32_t lp = live_base + pre + lrank;
            if ((int)threadIdx.x < cnt) live_pos[range.x + base + threadIdx.x] = lp;
            if (live) live_splat[range.x + lp] = my_id;
            live_base += tot;
        }
        if (notdone == 0ull || (dbg & 1)) continue;           // this wave's quadrant is finished
        // No per-lane branches (the scalar unit is shared by the CU's four SIMDs; a branchy body made this kernel scalar-bound); the
        // wave-uniform "everyone finished" exit is checked per 64-splat word.
#pragma unroll 1
        for (int lw = 0; lw < RB / 64; ++lw) {
            uint64_t m = uniform_u64(L.qmask[lw][wave]);
            if (m == 0) continue;
            if (notdone == 0ull) break;
            // LDS byte address of the word's first entry in a vector register (wave-uniform): a visit forms its address with ONE
            // v_lshl_add from the scalar bit index instead of s_or + s_lshl + v_mov
            typedef float fwd_v4f __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) const fwd_v4f lds_cv4f;
            uint32_t word_a = (uint32_t)(uintptr_t)&L.a[lw * 64];
            asm("" : "+v"(word_a));
            const int idx0 = base + lw * 64 + 1;
#if FWD_CMPX
            // The visit below narrows EXEC and must leave it as it found it (not "-1" — a later per-lane branch around this
            // loop would otherwise get its dead lanes back): the entry mask is read once per word of the walk, not per visit.
            uint64_t exec_entry;
            asm volatile("s_mov_b64 %0, exec" : "=s"(exec_entry));
#endif
            while (m) {
                const int bit = __builtin_ctzll(m);
                asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));          // m &= m - 1 in one scalar instruction instead of three
                uint32_t ent_a;
                asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(ent_a) : "s"(bit), "v"(word_a));
                const fwd_v4f A = *(lds_cv4f*)ent_a;
                const fwd_v4f B = *(lds_cv4f*)(ent_a + (uint32_t)sizeof(L.a));
                const float cb = *(const __attribute__((address_space(3))) float*)(ent_a + (uint32_t)(sizeof(L.a) + sizeof(L.b)));
                const float dx = A.x - pxf, dy = A.y - pyf;
                // log2 of the Gaussian falloff: p2 = log2e * power (sign unchanged), one v_exp_f32, no extra multiply
                const float p2 = __builtin_fmaf(B.x * dy, dy, __builtin_fmaf(A.w, dy, A.z * dx) * dx);
                const float alpha = fminf(DVS_ALPHA_MAX, B.y * __builtin_amdgcn_exp2f(p2));
                const float aT = alpha * T;
                const float test_T = T - aT;                       // = T (1 - alpha)
                const uint64_t nd_before = notdone;
#if FWD_CMPX
                // The predicates as a v_cmpx chain: EXEC = notdone, narrowed by the two alpha-rule compares to the pixels the splat can
                // contribute to; the stop test runs under that mask (VCC = pixels that stop here, WITHOUT this splat), leaves `notdone`
                // and EXEC, and the five updates run on what is left. No mask ever travels through a scalar AND: 9 scalar instructions
                // per visit instead of 12 (the scalar unit is shared by the CU's four SIMDs).
                const uint32_t idx = (uint32_t)(idx0 + bit);
                asm volatile("s_mov_b64 exec, %[nd]\n\t"
                             "v_cmpx_nlt_f32_e32 vcc, 0, %[p2]\n\t"            // !(p2 > 0)
                             "v_cmpx_ngt_f32_e32 vcc, %[amin], %[al]\n\t"      // !(alpha < 1/255)
                             "v_cmp_gt_f32_e32 vcc, %[tstop], %[tt]\n\t"       // T (1 - alpha) < 1e-4: the pixel is done
                             "s_andn2_b64 %[nd], %[nd], vcc\n\t"
                             "s_andn2_b64 exec, exec, vcc\n\t"
                             "v_fmac_f32 %[c0], %[cr], %[at]\n\t"
                             "v_fmac_f32 %[c1], %[cg], %[at]\n\t"
                             "v_fmac_f32 %[c2], %[cbl], %[at]\n\t"
                             "v_sub_f32 %[T], %[T], %[at]\n\t"                 // (not a move of test_T: that one is a fused fma; same bits as the other A7 kernels)
                             "v_mov_b32 %[last], %[idx]\n\t"
                             "s_mov_b64 exec, %[ee]"
                             : [c0] "+v"(C0), [c1] "+v"(C1), [c2] "+v"(C2), [T] "+v"(T), [last] "+v"(last), [nd] "+s"(notdone)
                             : [p2] "v"(p2), [al] "v"(alpha), [tt] "v"(test_T), [cr] "v"(B.z), [cg] "v"(B.w), [cbl] "v"(cb), [at] "v"(aT), [idx] "s"(idx),
                               [amin] "s"(DVS_ALPHA_MIN), [tstop] "s"(DVS_T_STOP), [ee] "s"(exec_entry)
                             : "vcc", "scc");
#else
                const uint64_t m_ok = notdone & __builtin_amdgcn_ballot_w64(!(p2 > 0.f)) & __builtin_amdgcn_ballot_w64(!(alpha < DVS_ALPHA_MIN));
                const uint64_t m_lt = __builtin_amdgcn_ballot_w64(test_T < DVS_T_STOP);
                const uint64_t m_take = m_ok & ~m_lt;              // contributes; (m_ok & m_lt: the pixel stops here, without this splat)
                notdone &= ~(m_ok & m_lt);
                const uint32_t idx = (uint32_t)(idx0 + bit);
                uint64_t saved_exec;
                asm volatile("s_and_saveexec_b64 %[sv], %[tk]\n\t"
                             "v_fmac_f32 %[c0], %[cr], %[at]\n\t"
                             "v_fmac_f32 %[c1], %[cg], %[at]\n\t"
                             "v_fmac_f32 %[c2], %[cbl], %[at]\n\t"
                             "v_sub_f32 %[T], %[T], %[at]\n\t"
                             "v_mov_b32 %[last], %[idx]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [c0] "+v"(C0), [c1] "+v"(C1), [c2] "+v"(C2), [T] "+v"(T), [last] "+v"(last), [sv] "=&s"(saved_exec)
                             : [tk] "s"(m_take), [cr] "v"(B.z), [cg] "v"(B.w), [cbl] "v"(cb), [at] "v"(aT), [idx] "s"(idx)
                             : "scc");                                  // (s_and_saveexec writes SCC)
#endif
                if (RECORD) {       // the same three comparisons on the same registers as the predicates above
                    const uint64_t took = nd_before & __builtin_amdgcn_ballot_w64(!(p2 > 0.f)) & __builtin_amdgcn_ballot_w64(!(alpha < DVS_ALPHA_MIN)) &
                                          ~__builtin_amdgcn_ballot_w64(test_T < DVS_T_STOP);
                    const uint64_t pos = (uint64_t)range.x + (uint64_t)(idx - 1u);
                    if (lane == 0 && pos < take_cap) take_masks[pos * 4 + wave] = took;
                }
            }
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

// ---- A8, retired kernels: experiment builds only (-DDVS_EXPERIMENT, tools/xbuild.sh) ---------------------------------------------
// "reduce" (round 1) and "mm" (the matrix-pipe experiment of round 2) are 40-90 % slower than the shipped kernel (render_tr.hip) and are
// not part of libdvsraster.so: dvs_set_backward_variant refuses them there. They stay in the source as the measured alternatives of
// DESIGN.md §5; an experiment build brings them back for A/B runs (DVS_TEST_ALL_VARIANTS=1 with DVS_RASTER_LIB=tools/xlib/...).
#ifdef DVS_EXPERIMENT
// ---- A8 -------------------------------------------------------------------------------------------
template <bool ABSGRAD>
__global__ void __launch_bounds__(RB)
k_render_bwd(ViewBg bg_arg /* MUST stay the first parameter: read through dvs_load_bg() */, int W, int H, int tiles_x, int tiles_per_view,
             int num_tiles /* = views * tiles_per_view */, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d,
             const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout /*[views,3,H,W]*/,
             float* __restrict__ grow /*[n,12]: Sx Sy Sxx Sxy Syy So r g b |mx| |my| pad (moments, see the loop body)*/,
             int lineage /*dvs_opts.grad_mode == DVS_GRAD_LINEAGE: the gradient passes the 0.99 alpha cap*/) {
    __shared__ BatchLds L;
    __shared__ uint32_t s_max[RB / 64];
    (void)bg_arg;
    const int tile_g = tile_of_block(blockIdx.x, num_tiles);
    if (tile_g >= num_tiles) return;
    const int view = tile_g / tiles_per_view, tile = tile_g - view * tiles_per_view;
    const float3 bgv = dvs_load_bg(view);
    const float bg0 = bgv.x, bg1 = bgv.y, bg2 = bgv.z;
    final_T += (size_t)view * W * H; n_contrib += (size_t)view * W * H; dL_dout += (size_t)view * 3 * W * H;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tx * DVS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * DVS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile_g];
    const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
    // which reduced value this lane publishes: lane c (< 3) of row r carries q[c] = value kv (see wave_reduce12)
    const int lrow = lane >> 4, lcol = lane & 15;
    const int kv = lcol * 4 + ((lrow == 1) ? 2 : (lrow == 2) ? 1 : lrow);
    const bool publisher = lcol < 3 && kv < (ABSGRAD ? 11 : 9);
    const int xaddr = (lane ^ 32) << 2;                       // ds_bpermute address of the partner lane in the other wave half

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) { dLp0 = dL_dout[pix]; dLp1 = dL_dout[P + pix]; dLp2 = dL_dout[2 * P + pix]; }
    const float bg_dot = (bg0 * dLp0 + bg1 * dLp1) + bg2 * dLp2;

    // entries beyond the deepest contributor of any pixel of this tile are never touched
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (lane == 0) s_max[wave] = wmax;
    __syncthreads();
    const int wave_last = (int)__builtin_amdgcn_readfirstlane(wmax);
    uint32_t todo = 0;
#pragma unroll
    for (int w = 0; w < RB / 64; ++w) todo = max(todo, s_max[w]);
    if (todo == 0) return;

    // Back-to-front state per pixel: T = transmittance in front of the current splat, and ONE scalar
    // D = sum over the splats behind of (c_k . dL/dC) alpha_k T_k + T_final (bg . dL/dC): because the upstream pixel
    // gradient is constant along the list, dL/dalpha_j = (c_j . dL/dC) T_j - D / (1 - alpha_j) needs no per-channel state.
    float T = T_final;
    float D = T_final * bg_dot;
    const int nbatch = (int)((todo + RB - 1) / RB);
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * RB;
        const int cnt = min(RB, (int)todo - base);
        __syncthreads();
        { uint32_t id_unused; (void)stage_batch(L, sorted_splat, range.x + base, cnt, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE), id_unused); }
        __syncthreads();
#pragma unroll 1
        for (int lw = RB / 64 - 1; lw >= 0; --lw) {
            const int lim = wave_last - base - lw * 64;          // list positions >= wave_last never reach this wave
            if (lim <= 0) continue;
            uint64_t m = uniform_u64(L.qmask[lw][wave]);
            if (lim < 64) m &= (1ull << lim) - 1ull;
            while (m) {
                const int bit = 63 - __builtin_clzll(m);
                m &= ~(1ull << bit);
                const int j = lw * 64 + bit;
                const uint32_t k = (uint32_t)(base + j);     // 0-based list position; contributor index k+1
                const float4 xy = L.xyc[j];
                const float2 zo2 = *reinterpret_cast<const float2*>(&L.zoir[j]);
                const float4 cs = make_float4(xy.z, xy.w, zo2.x, zo2.y);
                const float dx = xy.x - pxf, dy = xy.y - pyf;
                const float p2 = __builtin_fmaf(cs.z * dy, dy, __builtin_fmaf(cs.y, dy, cs.x * dx) * dx);   // same expression as the forward
                const float G = __builtin_amdgcn_exp2f(p2);
                const float oa = cs.w * G;
                const float alpha = fminf(DVS_ALPHA_MAX, oa);
                const bool contrib = (k < last) && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
                if (!__any(contrib)) continue;
                // predicated: a non-contributing lane runs with alpha = 0 (T, D unchanged, every partial exactly 0)
                const float4 cg = L.cog[j];
                const float2 ir = *(reinterpret_cast<const float2*>(&L.zoir[j]) + 1);       // splat id bits (publisher lanes), colour r
                const float3 c = make_float3(ir.y, cg.w, L.bl[j].x);
                const float4 co = make_float4(cg.x, cg.y, cg.z, cs.w);
                const float al = contrib ? alpha : 0.f;
                const float inv_1ma = __builtin_amdgcn_rcpf(1.f - al);        // 1-alpha >= 0.01: v_rcp_f32 (1 ulp) is ample
                T = T * inv_1ma;
                const float w = al * T;
                const float cd = (c.x * dLp0 + c.y * dLp1) + c.z * dLp2;
                float dL_dalpha = cd * T - D * inv_1ma;
                D = D + cd * w;
                // DVS_GRAD_TRUE: the 0.99 clamp blocks the gradient; DVS_GRAD_LINEAGE (README.md:95 lineage): it passes as if alpha = opacity * G
                dL_dalpha = (contrib && (lineage || !(oa > DVS_ALPHA_MAX))) ? dL_dalpha : 0.f;
                // Per-splat sums are published as MOMENTS of the weight s = dL/dG * G about the splat's mean:
                //   S_x = sum s dx, S_y = sum s dy, S_xx = sum s dx^2, S_xy = sum s dx dy, S_yy = sum s dy^2, S_o = sum G dL/dalpha
                // k_preprocess_bwd turns them into dL/dmean2D = -(a S_x + b S_y, c S_y + b S_x) and dL/dconic = (-S_xx/2, -S_xy, -S_yy/2):
                // the conic multiplications happen once per splat instead of once per (pixel, splat) pair (7 VALU fewer per visit).
                const float v5 = G * dL_dalpha;
                const float sw = co.w * v5;
                const float su = sw * dx, st = sw * dy;
                float v[12];
                v[0] = su; v[1] = st;
                v[2] = su * dx; v[3] = su * dy; v[4] = st * dy;
                v[5] = v5;
                v[6] = w * dLp0; v[7] = w * dLp1; v[8] = w * dLp2;
                // abs-grad needs the per-pixel |dL/dmean2D| itself
                v[9] = ABSGRAD ? fabsf(__builtin_fmaf(co.x, su, co.y * st)) : 0.f;
                v[10] = ABSGRAD ? fabsf(__builtin_fmaf(co.z, st, co.y * su)) : 0.f;
                v[11] = 0.f;
                float q[3];
                wave_reduce12<ABSGRAD ? 11 : 9>(v, q, xaddr);
                if (publisher) {
                    // 11 lanes, 11 consecutive floats of the splat's row: one global_atomic_add_f32 instruction
                    const float val = lcol == 0 ? q[0] : (lcol == 1 ? q[1] : q[2]);
                    atomicAdd(&grow[(size_t)__float_as_uint(ir.x) * 12 + kv], val);
                }
            }
        }
    }
}

// ---- A8, variant "mm": the per-splat sums as a small dense contraction on the fp32 matrix pipe ---------------------------
// Same tiling, staging, culling and back-to-front recurrence as k_render_bwd, but the 12-value cross-lane reduction per
// (wave, splat) visit — 40 % of that kernel's vector instructions — is gone. Every per-splat sum of a visit is linear in just TWO
// per-pixel scalars, v5 = G dL/dalpha and w = alpha T, against factors that depend on the pixel alone:
//     sum v5 {1, dx, dy, dx^2, dx dy, dy^2}  (S_o and the five moments; d = mean - pixel is a polynomial in the pixel coordinates)
//     sum w  {dL/dC_r, dL/dC_g, dL/dC_b}     (colour gradient)
//     sum |v5| |a dx + b dy| = sum z1 (L1 - a xi - b eta),  z1 = |v5| sgn(l1)   (abs-grad; l1 is linear in the pixel coordinates)
// so phase A (lane = pixel, the sequential part) only writes (v5, w) of the visit into a wave-private LDS row, and once 16 rows
// are full phase B contracts the [16 splats x 64 pixels] block against the per-pixel factor matrices with v_mfma_f32_16x16x4_f32
// (exact fp32, a k-ordered fma chain): A = the (v5 | w | z1 | z2) block read back in MFMA operand layout, B = the monomials
// {1, xi, eta, xi^2, xi eta, eta^2} of the pixel about the quadrant centre / the pixel's upstream gradient, both constant per
// wave and held in registers. The epilogue moves the raw moments to the splat's mean (exact algebra, fp32 roundoff of the same
// size as the direct sums) and publishes the 11 values of the 16 splats with three fully coalesced atomic instructions.
// The matrix pipe is used as a reduction engine only; nothing here is reshaped into a GEMM that was not one.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MM_SLOTS 16
#define MM_STRIDE 66      // float2 per slot row: 64 pixels + 2 pad = 528 B, so the 16 rows of a phase-B read hit distinct banks

template <bool ABSGRAD>
__global__ void __launch_bounds__(RB)
k_render_bwd_mm(int W, int H, int tiles_x, int num_tiles, const uint2* __restrict__ ranges,
                const uint32_t* __restrict__ sorted_splat, const float4* __restrict__ splat2d, float bg0, float bg1, float bg2,
                const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dout,
                float* __restrict__ grow /*[n,12], same row contract as k_render_bwd*/, int lineage, int dbg_arg /*experiment builds: ablation bits*/) {
    const int dbg = dbg_arg;
    __shared__ BatchLds L;
    __shared__ uint32_t s_max[RB / 64];
    __shared__ __attribute__((aligned(16))) float2 s_pair[RB / 64][MM_SLOTS * MM_STRIDE];   // per wave: [slot][pixel] (v5, w); epilogue scratch
    const int tile = tile_of_block(blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qx0 = tx * DVS_TILE + (wave & 1) * 8, qy0 = ty * DVS_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
    float2* const pairp = s_pair[wave];
    float* const sc = reinterpret_cast<float*>(pairp);       // epilogue scratch: [16][4][16] floats, then [16][12] at float 1024
    const int sl = lane & 15, kq = lane >> 4;                // MFMA operand coordinates of this lane: row/column sl, k index kq

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) { dLp0 = dL_dout[pix]; dLp1 = dL_dout[P + pix]; dLp2 = dL_dout[2 * P + pix]; }
    const float bg_dot = (bg0 * dLp0 + bg1 * dLp1) + bg2 * dLp2;

    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (lane == 0) s_max[wave] = wmax;
    // B operands (constant per wave). Step s of phase B covers the four pixels 4s .. 4s+3 of the quadrant (row s >> 1,
    // columns 4 (s & 1) ..): lane (sl, kq) holds column sl of pixel 4s + kq.
    //   ball[s]: monomials of (xi, eta) = pixel - quadrant centre in columns 0..5;  bgr[s]: the pixel's dL/dC in columns 0..2
    reinterpret_cast<float4*>(sc)[lane] = make_float4(dLp0, dLp1, dLp2, 0.f);
    __syncthreads();
    float ball[16], bgr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float xi = (float)(4 * (s & 1) + kq) - 3.5f, eta = (float)(s >> 1) - 3.5f;
        ball[s] = sl == 0 ? 1.f : sl == 1 ? xi : sl == 2 ? eta : sl == 3 ? xi * xi : sl == 4 ? xi * eta : sl == 5 ? eta * eta : 0.f;
        bgr[s] = sc[(4 * s + kq) * 4 + min(sl, 3)];
    }
    const int wave_last = (int)__builtin_amdgcn_readfirstlane(wmax);
    uint32_t todo = 0;
#pragma unroll
    for (int w = 0; w < RB / 64; ++w) todo = max(todo, s_max[w]);
    if (todo == 0) return;
    const float cx = (float)qx0 + 3.5f, cy = (float)qy0 + 3.5f;

    // phase B + epilogue for the `count` filled rows; jv: lane s (< 16) holds the batch index of row s
    auto flush = [&](int count, uint32_t jv) {
        if (dbg & 1) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float P1a = 0.f, P1b = 0.f, P2a = 0.f, P2b = 0.f, nb = 0.f, nc = 0.f;
        if (ABSGRAD) {
            const int js = __shfl((int)jv, sl, 64);
            const float4 xy = L.xyc[js];
            const float4 cg = L.cog[js];
            const float mu = xy.x - cx, nu = xy.y - cy, xi0 = (float)kq - 3.5f;
            const float L1 = cg.x * mu + cg.y * nu, L2 = cg.y * mu + cg.z * nu;
            P1a = L1 - cg.x * xi0; P1b = P1a - 4.f * cg.x; P2a = L2 - cg.y * xi0; P2b = P2a - 4.f * cg.y;
            nb = -cg.y; nc = -cg.z;
        }
        f32x4 accv = {0.f, 0.f, 0.f, 0.f}, accw = accv, acc1 = accv, acc2 = accv;
        const float2* rp = pairp + sl * MM_STRIDE + kq;
        if (!(dbg & 2))
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float2 vw = rp[4 * s];
            accv = __builtin_amdgcn_mfma_f32_16x16x4f32(vw.x, ball[s], accv, 0, 0, 0);
            accw = __builtin_amdgcn_mfma_f32_16x16x4f32(vw.y, bgr[s], accw, 0, 0, 0);
            if (ABSGRAD) {
                const float eta = (float)(s >> 1) - 3.5f;
                const float l1 = __builtin_fmaf(nb, eta, (s & 1) ? P1b : P1a), l2 = __builtin_fmaf(nc, eta, (s & 1) ? P2b : P2a);
                const float z1 = __builtin_copysignf(vw.x, l1), z2 = __builtin_copysignf(vw.x, l2);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(z1, ball[s], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(z2, ball[s], acc2, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // D[row = 4 kq + r][column sl] -> scratch [row][type][column]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* o = sc + (4 * kq + r) * 64 + sl;
            o[0] = accv[r]; o[16] = accw[r];
            if (ABSGRAD) { o[32] = acc1[r]; o[48] = acc2[r]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* sc2 = sc + 1024;
        if (lane < count) {                       // lane = row: raw moments about the quadrant centre -> moments about the mean
            const int j = (int)jv;
            const float4 m0 = *reinterpret_cast<const float4*>(sc + lane * 64);          // R_1 R_xi R_eta R_xixi
            const float2 m1 = *reinterpret_cast<const float2*>(sc + lane * 64 + 4);      // R_xieta R_etaeta
            const float4 cw = *reinterpret_cast<const float4*>(sc + lane * 64 + 16);     // colour sums
            const float4 xy = L.xyc[j];
            const float4 cg = L.cog[j];
            const float4 zo = L.zoir[j];
            const float mu = xy.x - cx, nu = xy.y - cy, op = zo.y;
            const float sx = mu * m0.x - m0.y, sy = nu * m0.x - m0.z;              // sum v5 dx, sum v5 dy
            const float sxx = __builtin_fmaf(mu, sx, m0.w - mu * m0.y);              // mu^2 R1 - 2 mu Rxi + Rxixi
            const float sxy = __builtin_fmaf(mu, sy, m1.x - nu * m0.y);              // mu nu R1 - mu Reta - nu Rxi + Rxieta
            const float syy = __builtin_fmaf(nu, sy, m1.y - nu * m0.z);              // nu^2 R1 - 2 nu Reta + Retaeta
            float ax = 0.f, ay = 0.f;
            if (ABSGRAD) {
                const float4 z1 = *reinterpret_cast<const float4*>(sc + lane * 64 + 32);
                const float4 z2 = *reinterpret_cast<const float4*>(sc + lane * 64 + 48);
                const float L1 = cg.x * mu + cg.y * nu, L2 = cg.y * mu + cg.z * nu;
                ax = op * ((L1 * z1.x - cg.x * z1.y) - cg.y * z1.z);
                ay = op * ((L2 * z2.x - cg.y * z2.y) - cg.z * z2.z);
            }
            float4* dst = reinterpret_cast<float4*>(sc2 + lane * 12);
            dst[0] = make_float4(op * sx, op * sy, op * sxx, op * sxy);
            dst[1] = make_float4(op * syy, m0.x, cw.x, cw.y);
            dst[2] = make_float4(cw.z, ax, ay, zo.z /* splat id bits */);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 16 rows x 12 floats = 3 x 64 lanes: consecutive lanes add consecutive floats of a splat's 48-B row
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = 64 * i + lane;
            const int row = (e * 43691) >> 19;            // e / 12 for e < 192
            const int comp = e - row * 12;
            if (row < count && comp < (ABSGRAD ? 11 : 9)) {
                const uint32_t id = __float_as_uint(sc2[row * 12 + 11]);
                atomicAdd(&grow[(size_t)id * 12 + comp], sc2[e]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    float T = T_final;
    float D = T_final * bg_dot;
    int slot = 0;                 // rows filled so far (wave-uniform)
    uint32_t jvec = 0;
    const int nbatch = (int)((todo + RB - 1) / RB);
    for (int b = nbatch - 1; b >= 0; --b) {
        const int base = b * RB;
        const int cnt = min(RB, (int)todo - base);
        __syncthreads();
        { uint32_t id_unused; (void)stage_batch(L, sorted_splat, range.x + base, cnt, splat2d, (float)(tx * DVS_TILE), (float)(ty * DVS_TILE), id_unused); }
        __syncthreads();
#pragma unroll 1
        for (int lw = RB / 64 - 1; lw >= 0; --lw) {
            const int lim = wave_last - base - lw * 64;
            if (lim <= 0) continue;
            uint64_t m = uniform_u64(L.qmask[lw][wave]);
            if (lim < 64) m &= (1ull << lim) - 1ull;
            while (m) {
                const int bit = 63 - __builtin_clzll(m);
                m &= ~(1ull << bit);
                const int j = lw * 64 + bit;
                const uint32_t k = (uint32_t)(base + j);
                const float4 xy = L.xyc[j];
                const float2 zo2 = *reinterpret_cast<const float2*>(&L.zoir[j]);
                const float dx = xy.x - pxf, dy = xy.y - pyf;
                const float p2 = __builtin_fmaf(zo2.x * dy, dy, __builtin_fmaf(xy.w, dy, xy.z * dx) * dx);   // same expression as the forward
                const float G = __builtin_amdgcn_exp2f(p2);
                const float oa = zo2.y * G;
                const float alpha = fminf(DVS_ALPHA_MAX, oa);
                const bool contrib = (k < last) && !(p2 > 0.f) && !(alpha < DVS_ALPHA_MIN);
                if (!__any(contrib)) continue;
                const float3 c = make_float3(L.zoir[j].w, L.cog[j].w, L.bl[j].x);
                const float al = contrib ? alpha : 0.f;
                const float inv_1ma = __builtin_amdgcn_rcpf(1.f - al);
                T = T * inv_1ma;
                const float w = al * T;
                const float cd = (c.x * dLp0 + c.y * dLp1) + c.z * dLp2;
                const float dL_dalpha = cd * T - D * inv_1ma;
                D = D + cd * w;
                // DVS_GRAD_TRUE: the 0.99 clamp blocks the gradient; DVS_GRAD_LINEAGE: it passes as if alpha = opacity * G
                const bool gate = contrib && (lineage || !(oa > DVS_ALPHA_MAX));
                const float v5 = gate ? G * dL_dalpha : 0.f;
                if (!(dbg & 4)) pairp[slot * MM_STRIDE + lane] = make_float2(v5, w);
                // lane `slot` of jvec = j (both wave-uniform; one SGPR per VALU instruction on gfx9, so the lane select goes through m0)
                asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(jvec) : "s"(j), "s"(slot) : "m0");
                if (++slot == MM_SLOTS) { flush(MM_SLOTS, jvec); slot = 0; }
            }
        }
        if (slot > 0) { flush(slot, jvec); slot = 0; }        // the rows refer to L by batch index: publish before L is restaged
    }
}

#endif  // DVS_EXPERIMENT

// ---- launchers -----------------------------------------------------------------------------------------

hipError_t dvs_launch_render_fwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* splat2d, const float* bgs, float* out_color, float* final_T,
                                 uint32_t* n_contrib, uint32_t* live_splat, uint32_t* live_pos, uint64_t* take_masks, uint64_t take_cap, uint32_t* ranges_out) {
    const int tiles_pv = tiles_x * tiles_y, num_tiles = tiles_pv * n_views;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
#ifdef DVS_EXPERIMENT
    static const int dbg = dvs_experiment_int("DVS_FWD_DEBUG");
#endif
    if (take_masks)
        hipLaunchKernelGGL(k_render_fwd<true>, dim3(grid), dim3(RB), 0, st, make_view_bg(n_views, bgs), W, H, tiles_x, tiles_pv, num_tiles,
                           (const uint2*)ranges, (uint2*)ranges_out, sorted_splat, (const float4*)splat2d, out_color, final_T, n_contrib, live_splat, live_pos, take_masks, take_cap DVS_DBG_PASS(dbg));
    else
        hipLaunchKernelGGL(k_render_fwd<false>, dim3(grid), dim3(RB), 0, st, make_view_bg(n_views, bgs), W, H, tiles_x, tiles_pv, num_tiles,
                           (const uint2*)ranges, (uint2*)ranges_out, sorted_splat, (const float4*)splat2d, out_color, final_T, n_contrib, live_splat, live_pos, (uint64_t*)nullptr, (uint64_t)0 DVS_DBG_PASS(dbg));
    return hipGetLastError();
}

#ifdef DVS_EXPERIMENT
hipError_t dvs_launch_render_bwd(hipStream_t st, int W, int H, int tiles_x, int tiles_y, int n_views, const uint32_t* ranges,
                                 const uint32_t* sorted_splat, const float* splat2d, const float* bgs, const float* final_T, const uint32_t* n_contrib,
                                 const float* dL_dout, float* grad_rows, int absgrad, int grad_mode, int variant) {
    const int tiles_pv = tiles_x * tiles_y, num_tiles = tiles_pv * n_views;
    if (num_tiles <= 0) return hipSuccess;
    const int grid = ((num_tiles + 7) >> 3) << 3;
    const int lineage = grad_mode == 1 ? 1 : 0;
    // experiment knobs (tools/bwd_probe.py): extra dynamic LDS to lower the occupancy, debug bits that drop parts of the mm kernel
    static const int dbg = dvs_experiment_int("DVS_MM_DEBUG");
    const size_t extra_lds = dvs_experiment_extra_lds();
    if (variant == DVS_BWD_REDUCE) {
#define DVS_RB(KERNEL)                                                                                                             \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(RB), extra_lds, st, make_view_bg(n_views, bgs), W, H, tiles_x, tiles_pv, num_tiles, \
                       (const uint2*)ranges, sorted_splat, (const float4*)splat2d, final_T, n_contrib, dL_dout, grad_rows, lineage)
        if (absgrad) DVS_RB(k_render_bwd<true>); else DVS_RB(k_render_bwd<false>);
#undef DVS_RB
    } else {                                             // the mm experiment: one view
#define DVS_RM(KERNEL)                                                                                                          \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(RB), extra_lds, st, W, H, tiles_x, num_tiles, (const uint2*)ranges, sorted_splat, \
                       (const float4*)splat2d, bgs[0], bgs[1], bgs[2], final_T, n_contrib, dL_dout, grad_rows, lineage, dbg)
        if (absgrad) DVS_RM(k_render_bwd_mm<true>); else DVS_RM(k_render_bwd_mm<false>);
#undef DVS_RM
    }
    return hipGetLastError();
}
#endif  // DVS_EXPERIMENT
