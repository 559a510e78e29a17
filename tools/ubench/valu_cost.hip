// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the cross-lane / transcendental / packed ops the
// composite kernels lean on. 2048 waves (8 per SIMD on 256 CUs); each wave runs REP x UNROLL copies of one instruction on
// independent registers; cost = elapsed shader cycles * SIMDs-worth / instructions. Development aid (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef REP
#define REP 2000        /* long enough (0.3 - 1 ms per kernel) that launch ramp-up is < 2 % of the measurement (round 6; was 200) */
#endif
#define UN 16
#define BODY(ASM)                                                                                   \
    for (int r = 0; r < REP; ++r) {                                                                 \
        _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile(ASM : "+v"(x[u]), "+v"(y[u])); } \
    }
template <int OP> __global__ void __launch_bounds__(256) k(float* out, long long* cyc) {
    float x[UN], y[UN];
    for (int u = 0; u < UN; ++u) { x[u] = threadIdx.x * 0.001f + u; y[u] = 1.0f + u * 0.01f; }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { BODY("v_fma_f32 %0, %0, %1, %1") }
    if (OP == 1) { BODY("v_permlane32_swap_b32 %0, %1") }
    if (OP == 2) { BODY("v_permlane16_swap_b32 %0, %1") }
    if (OP == 3) { BODY("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1") }
    if (OP == 4) { BODY("v_add_f32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1") }
    if (OP == 5) { BODY("v_add_f32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf") }
    if (OP == 6) { BODY("v_exp_f32 %0, %1") }
    if (OP == 7) { BODY("v_rcp_f32 %0, %1") }
    if (OP == 8) { BODY("v_cndmask_b32 %0, %0, %1, vcc") }
    if (OP == 9) { BODY("v_mul_f32 %0, %0, %1") }
    if (OP == 10) { BODY("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1") }
    if (OP == 11) { BODY("v_log_f32 %0, %1") }
    if (OP == 12) { BODY("v_sqrt_f32 %0, %1") }
    if (OP == 13) { BODY("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf") }
    if (OP == 14) { unsigned long long m = __ballot(threadIdx.x & 1);
        for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(y[u]), "s"(m)); } } }
    if (OP == 15) { BODY("v_add_f32 %0, %0, %1") }
    if (OP == 16) { BODY("v_fmac_f32 %0, %1, %1") }
    if (OP == 17) { BODY("v_max_f32 %0, %0, %1") }
    if (OP == 18) { BODY("v_cmp_lt_f32 vcc, %0, %1") }
    if (OP == 19) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; u += 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[u]) : "v"(y[u]), "v"(x[u + 1])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[u + 1]) : "v"(y[u]), "v"(x[u + 1])); } } }
    if (OP == 25) { BODY("v_add_u32 %0, %0, %1") }
    if (OP == 26) { BODY("v_mov_b32 %0, %1") }
    if (OP == 27) { BODY("v_lshlrev_b32 %0, 3, %1") }
    if (OP == 28) { BODY("v_mad_u32_u24 %0, %0, %1, %1") }
    if (OP == 29) { BODY("v_cvt_f32_u32 %0, %1") }
    if (OP == 30) { BODY("v_and_b32 %0, %0, %1") }
    unsigned long long sm[UN];
    if (OP >= 20) for (int u = 0; u < UN; ++u) sm[u] = __builtin_amdgcn_readfirstlane(blockIdx.x + u) | 0x100000000ull;
    if (OP == 20) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm[u]) : "s"(sm[(u + 1) % UN]) : "scc"); } } }
    if (OP == 21) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm[u]) : "s"(sm[(u + 1) % UN]) : "scc"); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[u]) : "v"(y[u])); } } }
    if (OP == 22) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(x[u]) : "v"(y[u])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (OP == 23) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(x[u]) : "v"(y[u])); asm volatile("v_add_f32 %0, %0, %0" : "+v"(y[u])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (OP == 24) { for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int u = 0; u < UN; ++u) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm[u]) : "s"(sm[(u + 1) % UN]) : "scc"); asm volatile("s_or_b64 %0, %0, %1" : "+s"(sm[u]) : "s"(sm[(u + 2) % UN]) : "scc"); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[u]) : "v"(y[u])); } } }
    if (OP >= 20) { for (int u = 0; u < UN; ++u) x[0] += (float)(unsigned)(sm[u] & 3); }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int u = 0; u < UN; ++u) s += x[u] + y[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> __global__ void __launch_bounds__(256) kpk(float* out, long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x[UN], y[UN];
    for (int u = 0; u < UN; ++u) { x[u] = f2{threadIdx.x * 0.001f + u, 1.f}; y[u] = f2{1.0f + u * 0.01f, 2.f}; }
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[u]), "+v"(y[u]));
            else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[u]), "+v"(y[u]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int u = 0; u < UN; ++u) s += x[u][0] + y[u][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K> void run(const char* name, K kern, float* out, long long* cyc, int blocks) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // per SIMD: waves_per_simd * REP*UN instructions issued during avg cycles (counter ticks at 100 MHz on some parts: also report wall)
    const double instr_per_simd = (double)blocks * 4 / 1024.0 * REP * UN;
    printf("%-28s wall %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz); s_memtime delta avg %.0f\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, avg);
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;      // >= 0: run just that SALU/LDS test (20..24)
    const int blocks = 2048;     // 8 workgroups per CU, 8 waves per SIMD
    float* out; long long* cyc; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    if (only == 20) { run("s_and_b64", k<20>, out, cyc, blocks); return 0; }
    if (only == 21) { run("s_and_b64 + v_add_f32 (per pair)", k<21>, out, cyc, blocks); return 0; }
    if (only == 22) { run("ds_swizzle_b32", k<22>, out, cyc, blocks); return 0; }
    if (only == 23) { run("ds_swizzle_b32 + v_add_f32 (per pair)", k<23>, out, cyc, blocks); return 0; }
    if (only == 24) { run("2 SALU + v_add_f32 (per triple)", k<24>, out, cyc, blocks); return 0; }
    if (only == 15) { run("v_add_f32", k<15>, out, cyc, blocks); return 0; }
    run("v_fma_f32", k<0>, out, cyc, blocks); run("v_permlane32_swap", k<1>, out, cyc, blocks); run("v_permlane16_swap", k<2>, out, cyc, blocks);
    run("v_add_f32_dpp quad_perm", k<3>, out, cyc, blocks); run("v_add_f32_dpp row_half_mirror", k<4>, out, cyc, blocks);
    run("v_add_f32_dpp row_bcast15", k<5>, out, cyc, blocks); run("v_exp_f32", k<6>, out, cyc, blocks); run("v_rcp_f32", k<7>, out, cyc, blocks);
    run("v_cndmask_b32", k<8>, out, cyc, blocks); run("v_mul_f32", k<9>, out, cyc, blocks); run("v_mov_b32_dpp row_mirror", k<10>, out, cyc, blocks);
    run("v_log_f32", k<11>, out, cyc, blocks); run("v_sqrt_f32", k<12>, out, cyc, blocks); run("v_mov_b32_dpp wave_ror1", k<13>, out, cyc, blocks);
    run("v_cndmask_b32 sgpr mask", k<14>, out, cyc, blocks); run("v_add_f32", k<15>, out, cyc, blocks); run("v_fmac_f32", k<16>, out, cyc, blocks);
    run("v_max_f32", k<17>, out, cyc, blocks); run("v_cmp_lt_f32 vcc", k<18>, out, cyc, blocks); run("v_fma_f32 3 distinct src", k<19>, out, cyc, blocks);
    run("v_pk_fma_f32", kpk<0>, out, cyc, blocks); run("v_pk_mul_f32", kpk<1>, out, cyc, blocks);
    run("v_add_u32", k<25>, out, cyc, blocks); run("v_mov_b32", k<26>, out, cyc, blocks); run("v_lshlrev_b32", k<27>, out, cyc, blocks);
    run("v_mad_u32_u24", k<28>, out, cyc, blocks); run("v_cvt_f32_u32", k<29>, out, cyc, blocks); run("v_and_b32", k<30>, out, cyc, blocks);
    return 0;
}
