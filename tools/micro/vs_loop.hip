// Micro-benchmark: the V-stationary Winograd inner loop on gfx950 -- per "tile" 8 chunks x (4 global U loads + 16 MFMA 32x32x2)
// with 128 V registers.  MODE 0: no loads (U constant), 1: loads issued but U constant in MFMAs, 2: full (double-buffered U).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* __restrict__ u, int tiles)
{
    const int tid = threadIdx.x, lane = tid & 63;
    v4f V[8][4];
    for (int c = 0; c < 8; ++c) for (int n = 0; n < 4; ++n) V[c][n] = (v4f){1.f, 2.f, 3.f, 4.f} * (float)(c * 4 + n + lane);
    const float* up = u + (tid >> 6) * 65536 + lane * 4;
    v4f bq[2][4];
    for (int n = 0; n < 4; ++n) bq[0][n] = *(const __attribute__((address_space(1))) v4f*)(up + n * 256);
    for (int n = 0; n < 4; ++n) bq[1][n] = bq[0][n];
    float s = 0.f;
    int lin = 0;
#pragma unroll 1
    for (int t = 0; t < tiles; ++t) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            ++lin;
            if (MODE >= 1) {
                const float* un = up + (lin & 63) * 1024;
                v4f tmp[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) tmp[n] = *(const __attribute__((address_space(1))) v4f*)(un + n * 256);
                if (MODE == 2) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bq[(kc + 1) & 1][n] = tmp[n];
                } else {
                    asm volatile("" :: "v"(tmp[0]), "v"(tmp[1]), "v"(tmp[2]), "v"(tmp[3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const v4f bb = bq[MODE == 2 ? (kc & 1) : 0][n];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, V[kc][n].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, V[kc][n].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, V[kc][n].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, V[kc][n].w, acc[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][5];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> void run(const char* name, int blocks_per_cu, const float* u)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int tiles = 400, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, u, 10);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, u, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * tiles * 128 * 2.0 * 32 * 32 * 2;
    printf("%-40s blocks/CU=%d  %.3f ms  %.1f TFLOP/s (MFMA rate)\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    float* u; hipMalloc(&u, 4 * 65536 * 4 + 4096); hipMemset(u, 0, 4 * 65536 * 4 + 4096);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("128 V regs, U constant, no loads", bpc, u);
        run<1>("+ 4 loads per 16 MFMA (unused)", bpc, u);
        run<2>("+ U double-buffered from the loads", bpc, u);
    }
    return 0;
}
