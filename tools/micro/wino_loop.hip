// Micro-benchmark: the Winograd kernel's per-chunk instruction mix on gfx950 --
// 8 ds_read_b128 + 32 VALU + 16 v_mfma_f32_32x32x2f32 (4 accumulators) per iteration, optional U loads from global.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only; 1: + VALU transform on registers; 2: + LDS reads; 3: + global U loads
__global__ __launch_bounds__(256, 2) void k(float* out, const float* __restrict__ u, int iters, float sg)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float* P = lds + lane * 4;
    v4f dA[4], dB[4], b[4], vp[4];
    for (int j = 0; j < 4; ++j) vp[j] = (v4f){1.f, 1.f, 1.f, 1.f};
    for (int j = 0; j < 4; ++j) { dA[j] = (v4f){1.f, 2.f, 3.f, 4.f} * (float)(j + lane); dB[j] = (v4f){.5f, .25f, .125f, 1.f}; b[j] = dB[j] + (float)j; }
    const float* up = u + (tid >> 6) * 65536 + lane * 4;
    for (int it = 0; it < iters; ++it) {
        v4f bn[4];
        if (MODE >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[j] = *(const __attribute__((address_space(1))) v4f*)(up + ((it + 1) & 63) * 1024 + j * 256);
        }
        if (MODE >= 2) {
            const int o = (it & 3) * 2048;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dA[j] = *(const v4f*)(P + o + j * 256); dB[j] = *(const v4f*)(P + o + 1024 + j * 256); }
        }
        v4f v[4];
        if (MODE == 4) {
            // software pipelined: v for this iteration was produced during the previous one
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = vp[j];
            v4f t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = dA[j] + sg * dB[j];
            vp[0] = t[0] - t[2]; vp[1] = t[1] + t[2]; vp[2] = t[2] - t[1]; vp[3] = t[1] - t[3];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].x, b[nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].y, b[nu].y, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].z, b[nu].z, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].w, b[nu].w, acc[nu], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dA[j] = dA[j] + acc[0][j];
            continue;
        }
        if (MODE >= 1) {
            v4f t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = dA[j] + sg * dB[j];
            v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dA[j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].x, b[nu].x, acc[nu], 0, 0, 0);
            acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].y, b[nu].y, acc[nu], 0, 0, 0);
            acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].z, b[nu].z, acc[nu], 0, 0, 0);
            acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu].w, b[nu].w, acc[nu], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = bn[j];
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dA[j] = dA[j] + acc[0][j];      // keep the transform live in the loop
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> void run(const char* name, int blocks_per_cu, const float* u)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, u, 100, -1.f);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, u, iters, -1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-36s blocks/CU=%d  %.3f ms  %.1f TFLOP/s (MFMA rate)\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    float* u; hipMalloc(&u, 4 * 65536 * 4 + 4096); hipMemset(u, 0, 4 * 65536 * 4 + 4096);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("MFMA only (4 acc)", bpc, u);
        run<1>("+ 32 VALU transform", bpc, u);
        run<2>("+ 8 ds_read_b128", bpc, u);
        run<3>("+ 4 global U loads (L2)", bpc, u);
        run<4>("32 VALU interleaved between MFMAs", bpc, u);
    }
    return 0;
}
