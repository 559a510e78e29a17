// lds_atomic.hip — cost of ds_add_f32 on gfx950 as a function of active lanes and address pattern (GPU box):
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench/lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* out, int mode, int iters, long long* cyc) {
    __shared__ float acc[4096];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    int addr;
    bool on = true;
    switch (mode) {
        case 0: addr = l; break;                                   // 64 lanes, consecutive
        case 1: addr = l; on = l < 44; break;                      // 44 lanes, consecutive
        case 2: addr = (l >> 4) * 100 * 12 + (l & 15); on = (l & 15) < 11; break;     // 4 rows of 11, random-ish rows (stride 1200 floats)
        case 3: addr = (l >> 4) * 17 * 16 + (l & 15); on = (l & 15) < 11; break;      // 4 rows of 11, 16-float stride rows
        case 4: addr = l & 15; on = (l & 15) < 11; break;          // 4 groups on the SAME row (same-address conflicts x4)
        case 5: addr = l * 16; break;                              // 64 lanes, all on bank 0 (stride 16 floats -> with 64 banks: 4 banks)
        case 6: addr = l; on = l < 11; break;                      // 11 lanes
        default: addr = l; break;
    }
    addr += w * 1024;
    const float v = 1.0f + l;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (on) atomicAdd(&acc[(addr + (mode == 2 ? (i * 48) & 511 : 0)) & 4095], v);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 4096;
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int mode = 0; mode <= 6; ++mode) {
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(64 * waves), 0, 0, out, mode, iters, cyc);
            hipDeviceSynchronize();
            std::vector<long long> h(1024);
            hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
            printf("waves/WG %d mode %d: %.1f cycles per ds_add_f32 wave-instruction (per wave)\n", waves, mode, s / 1024 / iters);
        }
    return 0;
}
