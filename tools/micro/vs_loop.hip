// Micro-benchmark: the V-stationary Winograd inner loop on gfx950 -- per "tile" 8 chunks x (4 global U loads + 16 MFMA 32x32x2)
// with 128 V registers.  MODE 0: no loads (U constant), 1: loads issued but U constant in MFMAs, 2: full (double-buffered U).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int PRIO, int SKEW, int LATE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* __restrict__ u, int tiles)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    v4f V[8][4];
    for (int c = 0; c < 8; ++c) for (int n = 0; n < 4; ++n) V[c][n] = (v4f){1.f, 2.f, 3.f, 4.f} * (float)(c * 4 + n + lane);
    const float* up = u + (tid >> 6) * (65536 + SKEW) + (SKEW ? (blockIdx.x & 7) * 4096 : 0) + lane * 4;
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
            if (MODE == 3) {            // scalar base + 32-bit lane offset (global_load ... saddr)
                const float* ub_ = u + (tid >> 6) * 65536 + (lin & 63) * 1024;      // wave-uniform
                const unsigned vo = (unsigned)lane * 16u;
                v4f tmp[4];
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    tmp[n] = *(const __attribute__((address_space(1))) v4f*)((const char*)ub_ + n * 1024 + vo);
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[(kc + 1) & 1][n] = tmp[n];
            } else if (MODE == 6) {     // non-temporal loads (bypass L1)
                const float* un = up + (lin & 63) * 1024;
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[(kc + 1) & 1][n] = __builtin_nontemporal_load((const __attribute__((address_space(1))) v4f*)(un + n * 256));
            } else if (MODE == 4) {     // same bytes as 8 x dwordx2
                const float* un = up + (lin & 63) * 1024 - lane * 2;
                typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const v2f a0 = *(const __attribute__((address_space(1))) v2f*)(un + n * 256);
                    const v2f a1 = *(const __attribute__((address_space(1))) v2f*)(un + n * 256 + 128);
                    bq[(kc + 1) & 1][n] = (v4f){a0.x, a0.y, a1.x, a1.y};
                }
            } else if (MODE == 5) {     // LDS reads instead of global loads
                const float* un = lds + ((lin & 3) * 4096 + lane * 4) % 12288;
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[(kc + 1) & 1][n] = *(const v4f*)(un + n * 256);
            } else if (MODE == 2 && LATE > 0) {
                // loads are issued inside the MFMA block below
            } else if (MODE >= 1) {
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
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (MODE == 2 && LATE > 0 && n == LATE) {
                    const float* un = up + (lin & 63) * 1024;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[(kc + 1) & 1][q] = *(const __attribute__((address_space(1))) v4f*)(un + q * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const v4f bb = bq[MODE >= 2 ? (kc & 1) : 0][n];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, V[kc][n].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, V[kc][n].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, V[kc][n].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, V[kc][n].w, acc[n], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][5];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int PRIO = 0, int SKEW = 0, int LATE = 0> void run(const char* name, int blocks_per_cu, const float* u)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int tiles = 400, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, PRIO, SKEW, LATE><<<grid, 256>>>(out, u, 10);
    hipEventRecord(e0);
    k<MODE, PRIO, SKEW, LATE><<<grid, 256>>>(out, u, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * tiles * 128 * 2.0 * 32 * 32 * 2;
    printf("%-40s blocks/CU=%d  %.3f ms  %.1f TFLOP/s (MFMA rate)\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    float* u; hipMalloc(&u, 8 * 65536 * 4); hipMemset(u, 0, 8 * 65536 * 4);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("128 V regs, U constant, no loads", bpc, u);
        run<1>("+ 4 loads per 16 MFMA (unused)", bpc, u);
        run<2>("+ U double-buffered from the loads", bpc, u);
        run<2, 1>("U double-buffered, setprio 1 around MFMAs", bpc, u);
        run<2, 3>("U double-buffered, setprio 3 around MFMAs", bpc, u);
        run<2, 0, 1024>("U double-buffered, wave bases skewed 4 KB", bpc, u);
        run<2, 0, 256>("U double-buffered, wave bases skewed 1 KB", bpc, u);
        run<3>("U via saddr + 32-bit voffset", bpc, u);
        run<4>("U via 8 x dwordx2", bpc, u);
        run<2, 0, 0, 1>("U loads issued after 4 MFMAs", bpc, u);
        run<2, 0, 0, 2>("U loads issued after 8 MFMAs", bpc, u);
        run<2, 0, 0, 3>("U loads issued after 12 MFMAs", bpc, u);
        run<5>("U via 4 x ds_read_b128", bpc, u);
        run<6>("U via nontemporal loads", bpc, u);
    }
    return 0;
}
