// Micro-benchmark: L1 (TCP) hit bandwidth of gather-style global_load_dwordx4 on gfx950.
// Pattern G lanes per segment: a wave instruction touches 64/G segments of G*16 bytes (G=4: 64 B half lines as in the
// DCNv2 gather with 16-channel k-steps; G=8: full 128 B lines; G=64: one contiguous 1 KiB run).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int G>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* out, int iters, int footprint_bytes)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nseg = footprint_bytes / (G * 16);
    // pseudo-random segment per lane group, changes every iteration
    unsigned s = (lane / G) * 2654435761u + w * 40503u + blockIdx.x * 9176u;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const int seg = (s >> 8) % nseg;
            const v4f v = *(const __attribute__((address_space(1))) v4f*)(x + (size_t)seg * (G * 4) + (lane % G) * 4);
            acc += v;
        }
    }
    out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int G> void run(const float* x, float* out, int footprint)
{
    const int iters = 2000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<G><<<grid, 256>>>(x, out, 10, footprint);
    hipEventRecord(e0);
    k<G><<<grid, 256>>>(x, out, iters, footprint);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 256 * iters * 8 * 16;
    printf("G=%2d lanes/segment (%4d B)  footprint %6d B  %.3f ms  %.1f TB/s aggregate  %.1f B/clk/CU @2.4GHz\n", G, G * 16, footprint, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4e3);
}
int main()
{
    float *x, *out; hipMalloc(&x, 1 << 24); hipMemset(x, 0, 1 << 24); hipMalloc(&out, 256 * 4 * 256 * 4);
    for (int fp : {8192, 16384, 262144}) {
        run<4>(x, out, fp); run<8>(x, out, fp); run<16>(x, out, fp); run<64>(x, out, fp);
    }
    return 0;
}
