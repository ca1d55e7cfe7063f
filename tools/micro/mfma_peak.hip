// Micro-benchmark: f32 MFMA issue ceilings on gfx950 (register-fed vs LDS-fed like the conv inner loops).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 fa = make_float4(1.f, 2.f, 3.f, 4.f), fb = make_float4(.5f, .25f, .125f, 1.f);
    const float* P = lds + (lane & 31) * 20 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {   // LDS-fed: 2 b128 (A) + 1 b128 (B) per 8 MFMAs, like TM=2,TN=1 (one h)
            const int o = (it & 7) * 640;
            float4 a0 = *reinterpret_cast<const float4*>(P + o);
            float4 a1 = *reinterpret_cast<const float4*>(P + o + 32 * 20);
            float4 b0 = *reinterpret_cast<const float4*>(P + o + 64 * 20);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0], 0, 0, 0);
            acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1 % NACC], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0], 0, 0, 0);
            acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1 % NACC], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0], 0, 0, 0);
            acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1 % NACC], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0], 0, 0, 0);
            acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1 % NACC], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.y, acc[u % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC> void run(const char* name, int blocks_per_cu)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC><<<grid, 256>>>(out, 100);
    hipEventRecord(e0);
    k<MODE, NACC><<<grid, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 2;
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<0, 1>("reg-fed, 1 acc (dependent)", 1);
    run<0, 2>("reg-fed, 2 acc", 1);
    run<0, 4>("reg-fed, 4 acc", 1);
    run<0, 2>("reg-fed, 2 acc", 2);
    run<0, 2>("reg-fed, 2 acc", 4);
    run<1, 2>("LDS-fed, 2 acc", 1);
    run<1, 2>("LDS-fed, 2 acc", 2);
    run<1, 2>("LDS-fed, 2 acc", 4);
    return 0;
}
