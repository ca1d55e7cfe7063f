// 1x1 / stride 1 / pad 0 convolution of ONE NHWC source with a SHORT reduction (K = Cin = 64) + folded BN + residual + ReLU on the
// exact-f32 matrix cores (gfx950): the HBM-bound "expand" convolutions of the bottleneck blocks at high resolution
// (msra_resnet.py:82-102 Bottleneck conv3 64 -> 256 + residual and `downsample` @128x128; pose_higher_hrnet.py layer1: the same block).
//
// Why a separate kernel (round 6, measured first: tools/pointwise_probe.py, profiles/r6_pointwise_probe.txt).  These launches move
// 168-302 MB for 4.3 GFLOP; an elementwise kernel with the same traffic streams at 5.8-6.9 TB/s on this chip while igemm_conv_kernel
// reaches 2.8-3.0 TB/s on them.  The generic kernel walks K in 16-wide steps with one barrier and one register-prefetched slice per
// step and loads the residual in its epilogue: a block of a K = 64 layer lives for four k-steps and has one 12 KB slice (later its
// residual tile) in flight -- with 2-3 resident blocks per CU far less than the ~60 KB per CU that 8 TB/s x 2 us of latency need.
// Here a block requests EVERYTHING it will ever read up front -- its whole A tile (64 x 64), its whole weight tile (64 x 64) and its
// residual tile (64 x 64, parked in registers): 48 KB in flight per block, four blocks per CU (34.8 KB of LDS, 105 VGPRs) -- then
// one barrier, 32 MFMAs per wave without further synchronisation, one LDS transpose, float4 stores.
// Measured (B = 8, 128 x 128, in-graph): conv3 64 -> 256 + residual + ReLU 99.6-102.9 us -> 67.1-69.1 us (1.5x; a 128-row tile at three
// blocks per CU: 71.0-74.3), downsample 64 -> 256 59.1-60.4 -> 52.4-53.0, 64 -> 64 20.2 -> 20.0, 64 -> 256 + residual @32x32 (256 blocks)
// 8.6 -> 6.6.  Ablation of the kept kernel: without its MFMAs the downsample launch takes 23.1 us (7.3 TB/s: the memory phases alone),
// without its stores 44.6 us -- it is now bound by its LDS / MFMA phase, not by HBM; conv3 + residual without MFMAs 58.2 us (5.2 TB/s),
// without stores 50.9: within 15 % of its memory floor.  A K = 128 variant (128 -> 512 @64x64: 53.5 vs 52.2 us) did not pay and was
// not kept.
//
// Arithmetic: the SAME accumulation order as igemm_conv_kernel (k-steps of 16 in order, inside a step the two b128 halves and their
// four lanes-of-k in the same sequence) and the same epilogue expression -> BIT-IDENTICAL results; the choice between the two kernels
// is a pure performance rule (cp_conv2d_f32: tile 0 = auto, tile 1 = this kernel, 64064 / 128064 = the generic one; CP_POINTWISE=0
// keeps the generic kernel).
#include "igemm.h"

template <int BM, int KK>
struct PwTile {
    static constexpr int BN = 64;
    static constexpr int LDK = KK + 4;                           // LDS row stride in floats: 16-lane b128 groups hit 64 distinct banks
    static constexpr int TM = BM / 64;                           // 2 x 2 waves, wave tile (BM / 2) x 32
    static constexpr int A_F4 = BM * KK / 4, B_F4 = BN * KK / 4, R_F4 = BM * BN / 4;
    static constexpr int A_SLOTS = A_F4 / IG_THREADS, B_SLOTS = B_F4 / IG_THREADS, R_SLOTS = R_F4 / IG_THREADS;
    static constexpr int LDC = BN + 4;
    static constexpr int MAIN_BYTES = (BM + BN) * LDK * 4, EPI_BYTES = BM * LDC * 4;
    static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(A_F4 % IG_THREADS == 0 && B_F4 % IG_THREADS == 0 && R_F4 % IG_THREADS == 0, "whole slots");
    static_assert(BM % 64 == 0 && KK % IG_BK == 0, "tile");
};

template <int BM, int KK, int OCC>
__global__ __launch_bounds__(IG_THREADS, OCC) void pw_conv_kernel(const ConvArgs a)
{
    using T = PwTile<BM, KK>;
    constexpr int BN = T::BN, LDK = T::LDK, KQ = KK / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [BM][LDK]
    float* Bs = smem + BM * LDK;         // [BN][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NT = a.ldw / BN;
    const int tile = ig_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % NT, mt = tile / NT;                     // the channel tiles of one pixel tile are neighbours: its A rows stay in one L2
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * 32;

    // ---- every byte this block reads, requested at once
    const float* __restrict__ x = a.src[0];
    const int ld = a.srcLd[0];
    float4 ar[T::A_SLOTS], br[T::B_SLOTS], rr[T::R_SLOTS];
#pragma unroll
    for (int s = 0; s < T::A_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS, row = idx / KQ, c4 = idx - row * KQ;
        const int m = m0 + row;
        ar[s] = ig_ldg4(x + (size_t)(m < a.M ? m : 0) * ld + c4 * 4);      // rows past M: a valid address, never stored
    }
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS, row = idx / KQ, c4 = idx - row * KQ;
        br[s] = ig_ldg4(a.w + (size_t)(n0 + row) * KK + c4 * 4);
    }
    const int rc4 = tid & 15, n = n0 + rc4 * 4;                   // a thread's epilogue items share their four channels
    const bool nok = n < a.Cout;
    const float* const a_res = a.res;
    if (a_res) {
#pragma unroll
        for (int s = 0; s < T::R_SLOTS; ++s) {
            const int m = m0 + (tid >> 4) + s * 16;
            rr[s] = ig_ldg4(a_res + (size_t)((m < a.M && nok) ? m : 0) * a.resLd + (nok ? n : 0));
        }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nok) { sc = ig_ldg4(a.scale + n); sh = ig_ldg4(a.shift + n); }
#pragma unroll
    for (int s = 0; s < T::A_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS, row = idx / KQ, c4 = idx - row * KQ;
        *reinterpret_cast<float4*>(As + row * LDK + c4 * 4) = ar[s];
    }
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS, row = idx / KQ, c4 = idx - row * KQ;
        *reinterpret_cast<float4*>(Bs + row * LDK + c4 * 4) = br[s];
    }
    __syncthreads();

    // ---- K / 2 MFMAs per accumulator, no barrier: the order of igemm.h's ig_compute (k-step, b128 half, x y z w)
    f32x16 acc[T::TM];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int g = lane >> 5, il = lane & 31;
    const float* ap = As + (wm0 + il) * LDK + g * 4;
    const float* bp = Bs + (wn0 + il) * LDK + g * 4;
#pragma unroll
    for (int ks = 0; ks < KK / IG_BK; ++ks) {
        float4 af[2][T::TM], bf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < T::TM; ++i) af[h][i] = *reinterpret_cast<const float4*>(ap + i * 32 * LDK + ks * IG_BK + h * 8);
            bf[h] = *reinterpret_cast<const float4*>(bp + ks * IG_BK + h * 8);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < T::TM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i].x, bf[h].x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i].y, bf[h].y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i].z, bf[h].z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i].w, bf[h].w, acc[i], 0, 0, 0);
            }
    }

    // ---- epilogue: C tile through LDS (Cs[m][n]) so that every thread stores float4 runs along n; residual from the registers
    float* Cs = smem;
    constexpr int LDC = T::LDC;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[(wm0 + i * 32 + ig_row<32>(r, lane)) * LDC + wn0 + il] = acc[i][r];
    __syncthreads();
    const bool relu = a.act == CP_ACT_RELU;
#pragma unroll
    for (int s = 0; s < T::R_SLOTS; ++s) {
        const int row = (tid >> 4) + s * 16, m = m0 + row;
        if (m < a.M && nok) {
            float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + rc4 * 4);
            v = cp_scale_shift4(v, sc, sh);
            if (a_res) { v.x += rr[s].x; v.y += rr[s].y; v.z += rr[s].z; v.w += rr[s].w; }
            if (relu) v = cp_relu4(v);
            *reinterpret_cast<float4*>(a.out + (size_t)m * a.outLd + n) = v;
        }
    }
}

template <int BM, int KK, int OCC>
static int launch_pw(const ConvArgs& a, hipStream_t s)
{
    using T = PwTile<BM, KK>;
    auto kern = pw_conv_kernel<BM, KK, OCC>;
    static CpLdsGuard guard;
    if (T::SMEM > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, T::SMEM);
        if (e != hipSuccess) { cp_set_error("conv2d (pointwise): cannot reserve %d B LDS: %s", T::SMEM, hipGetErrorString(e)); return 2; }
    }
    const int grid = cp_cdiv(a.M, BM) * (a.ldw / T::BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(IG_THREADS), T::SMEM, s, a);
    cp_note_kernel("pw_conv_kernel<%d, %d, %d>", BM, KK, OCC);
    return 0;
}

// -1 = not this kernel's shape: one NHWC source, 1x1 / stride 1 / pad 0, K = 64, dense NHWC output with Cout % 4 == 0 and 16-byte aligned
// tensors, no activation or ReLU, no split.  Every eligible launch takes it (also small ones: 64 -> 256 + residual @32x32, 256 blocks,
// 6.6 vs 8.6 us; K = 64 is too short for the generic kernel's split-K anyway).
int cp_launch_conv_pointwise(const ConvArgs& a, hipStream_t s)
{
    const bool ok = a.nsrc == 1 && a.kh == 1 && a.kw == 1 && a.sy == 1 && a.sx == 1 && a.py == 0 && a.px == 0 && a.H == a.Ho && a.W == a.Wo &&
                    a.K == 64 && a.srcC[0] == a.K && a.srcLd[0] % 4 == 0 && a.ldw % 64 == 0 && !a.outNCHW && a.osy == 1 &&
                    a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.OH == a.Ho && a.OW == a.Wo && a.ksplit == 1 && a.nsub == 1 &&
                    (a.act == CP_ACT_NONE || a.act == CP_ACT_RELU) && ((a.outLd | a.Cout) & 3) == 0 &&
                    (((size_t)a.src[0] | (size_t)a.w | (size_t)a.out | (size_t)a.scale | (size_t)a.shift) & 15) == 0 &&
                    (!a.res || ((a.resLd & 3) == 0 && (((size_t)a.res) & 15) == 0)) &&
                    (long long)a.M * a.srcLd[0] * 4 < (1ll << 32);
    if (!ok) return -1;
    return launch_pw<64, 64, 4>(a, s);
}
