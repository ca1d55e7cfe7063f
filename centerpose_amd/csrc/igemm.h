// Implicit-GEMM convolution core for gfx950, fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s dense peak).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wp[k][n]
//   m = (b, oy, ox) output pixel, n = output channel, k = (tap, input channel) with the input
//   channel fastest (activations are NHWC, so a k-chunk of one pixel is contiguous in HBM).
// Block tile BM x BN, k-step BK = 16, 256 threads = 4 wave64.  Per k-step:
//   * A-producer (plain im2col gather, 3-channel NCHW stem gather, or DCNv2 bilinear sampler)
//     loads its slice into registers while the previous slice is being multiplied,
//   * the slice is written k-major into LDS (As[k][m], Bs[k][n]; row pads chosen so the
//     transposing ds_write_b32 and the row-contiguous ds_read_b32 are both conflict-free),
//   * every wave reads MFMA fragments with one ds_read_b32 per operand register.
// Two LDS buffers, one barrier per k-step.  Epilogue fuses scale/shift (folded BN or bias),
// residual add, ReLU / sigmoid and writes NHWC (coalesced along n) or NCHW (transposed through
// LDS so stores are coalesced along the pixel index).
#pragma once
#include "common.h"

#define IG_BK 16
#define IG_THREADS 256
#define IG_MAX_SRC 4

enum { CP_ACT_NONE = 0, CP_ACT_RELU = 1, CP_ACT_SIGMOID = 2 };

struct ConvArgs {
    // inputs: up to 4 NHWC tensors concatenated along channels (DLA Root: cat -> 1x1 conv)
    const float* src[IG_MAX_SRC];
    int srcC[IG_MAX_SRC];     // channels taken from each source (multiple of 16 unless stem)
    int srcLd[IG_MAX_SRC];    // floats between consecutive pixels
    int nsrc, Ctot;
    int B, H, W;              // input spatial size
    int Ho, Wo, M;            // output positions computed by this launch; M = B*Ho*Wo
    int kh, kw, sy, sx, py, px;
    int K;                    // kh*kw*Ctot rounded up to a multiple of 16 (zero weight rows)
    const float* w;           // packed weights [K][ldw]
    int ldw;
    const float* scale;       // [ldw]  y = acc*scale + shift
    const float* shift;
    const float* res;         // optional residual, NHWC
    int resLd;
    float* out;
    int outLd;                // NHWC: floats between pixels;  NCHW: unused
    int Cout;                 // channels actually stored
    int outNCHW;              // 0: NHWC, 1: NCHW [B, Cout, OH, OW]
    int OH, OW;               // full output tensor spatial size
    int osy, osx, ooy, oox;   // output pixel = (oy*osy+ooy, ox*osx+oox)  (sub-pixel deconv)
    int act;
    // DCNv2 only: offset/mask tensor [B,H,W,omLd]: ch 2k = dy_k, 2k+1 = dx_k, 18+k = mask logit
    const float* om;
    int omLd;
    int omMaskOff;            // first mask channel (2*kh*kw)
    int omSigmoid;            // 1: mask channel holds logits (apply sigmoid), 0: mask given directly
    int dily, dilx;           // dilation (DCNv2 drop-in only; plain convs are dilation 1)
};

// XCD-aware, bijective remap of the linear block id: blocks that are consecutive in the
// remapped order (and share the same activation rows) land on the same XCD / L2.
__device__ __forceinline__ int ig_xcd_remap(int id, int n)
{
    const int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

template <int MF> struct IgAcc;
template <> struct IgAcc<32> { typedef f32x16 type; static constexpr int N = 16; static constexpr int KS = 2; };
template <> struct IgAcc<16> { typedef f32x4 type; static constexpr int N = 4; static constexpr int KS = 4; };

template <int MF>
__device__ __forceinline__ typename IgAcc<MF>::type ig_mfma(float a, float b, typename IgAcc<MF>::type c)
{
    if constexpr (MF == 32) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row of accumulator register r held by `lane`
template <int MF> __device__ __forceinline__ int ig_row(int r, int lane)
{
    if constexpr (MF == 32) return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    else return (lane >> 4) * 4 + r;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MF>
struct IgTile {
    static constexpr int LDA = BM + 2;           // LDA % 8 == 2: transposing b32 writes conflict-free
    static constexpr int LDB = BN + 4;           // 16 B aligned rows for ds_write_b128
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / MF, TN = WN / MF;
    static constexpr int A_SLOTS = BM * 4 / IG_THREADS;                  // float4 per thread per k-step
    static constexpr int B_F4 = IG_BK * BN / 4;
    static constexpr int B_SLOTS = (B_F4 + IG_THREADS - 1) / IG_THREADS;
    static constexpr int A_BYTES = IG_BK * LDA * 4, B_BYTES = IG_BK * LDB * 4;
    static constexpr int MAIN_BYTES = 2 * (A_BYTES + B_BYTES);
    static constexpr int EPI_BYTES = BN * (BM + 1) * 4;                  // NCHW transposed epilogue
    static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(WAVES_M * WAVES_N * 64 == IG_THREADS, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    static_assert(A_SLOTS >= 1, "BM >= 64");
};

// ---- B (weights) tile: global -> regs -> LDS ---------------------------------------------
template <class T, int BN>
__device__ __forceinline__ void ig_load_b(const ConvArgs& a, int k0, int n0, int tid, float4 (&br)[T::B_SLOTS])
{
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        if (idx < T::B_F4) {
            const int kr = idx / (BN / 4), n4 = idx % (BN / 4);
            br[s] = *reinterpret_cast<const float4*>(a.w + (size_t)(k0 + kr) * a.ldw + n0 + n4 * 4);
        }
    }
}
template <class T, int BN>
__device__ __forceinline__ void ig_store_b(float* Bs, int tid, const float4 (&br)[T::B_SLOTS])
{
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        if (idx < T::B_F4) {
            const int kr = idx / (BN / 4), n4 = idx % (BN / 4);
            *reinterpret_cast<float4*>(Bs + kr * T::LDB + n4 * 4) = br[s];
        }
    }
}

// ---- MFMA over one k-step held in LDS --------------------------------------------------------
template <class T, int MF>
__device__ __forceinline__ void ig_compute(const float* As, const float* Bs, int wm0, int wn0, int lane,
                                           typename IgAcc<MF>::type (&acc)[T::TM][T::TN])
{
    constexpr int KS = IgAcc<MF>::KS;
    const int kl = (MF == 32) ? (lane >> 5) : (lane >> 4);
    const int il = lane & (MF - 1);
#pragma unroll
    for (int kk = 0; kk < IG_BK / KS; ++kk) {
        float af[T::TM], bf[T::TN];
        const int kr = kk * KS + kl;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) af[i] = As[kr * T::LDA + wm0 + i * MF + il];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) bf[j] = Bs[kr * T::LDB + wn0 + j * MF + il];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) acc[i][j] = ig_mfma<MF>(af[i], bf[j], acc[i][j]);
    }
}

__device__ __forceinline__ float ig_act(float v, int act)
{
    if (act == CP_ACT_RELU) return fmaxf(v, 0.f);
    if (act == CP_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// ---- epilogue --------------------------------------------------------------------------------
template <class T, int BM, int BN, int MF>
__device__ __forceinline__ void ig_epilogue(const ConvArgs& a, float* smem, int m0, int n0, int wm0, int wn0,
                                            int lane, int tid, typename IgAcc<MF>::type (&acc)[T::TM][T::TN])
{
    const int HoWo = a.Ho * a.Wo;
    const int cl = lane & (MF - 1);
    if (!a.outNCHW) {
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int n = n0 + wn0 + j * MF + cl;
            const bool nok = n < a.Cout;
            const float sc = a.scale[n], sh = a.shift[n];     // arrays are padded to ldw
#pragma unroll
            for (int i = 0; i < T::TM; ++i) {
#pragma unroll
                for (int r = 0; r < IgAcc<MF>::N; ++r) {
                    const int m = m0 + wm0 + i * MF + ig_row<MF>(r, lane);
                    if (m < a.M && nok) {
                        size_t opix;
                        if (a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.OH == a.Ho && a.OW == a.Wo) {
                            opix = (size_t)m;
                        } else {
                            const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
                            opix = ((size_t)b * a.OH + (oy * a.osy + a.ooy)) * a.OW + (ox * a.osx + a.oox);
                        }
                        float v = acc[i][j][r] * sc + sh;
                        if (a.res) v += a.res[opix * a.resLd + n];
                        a.out[opix * a.outLd + n] = ig_act(v, a.act);
                    }
                }
            }
        }
    } else {
        // transpose through LDS: Cs[n][m] so that global stores run along the pixel index
        float* Cs = smem;
        constexpr int LDC = BM + 1;
        __syncthreads();   // main-loop buffers are dead
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int nl = wn0 + j * MF + cl;
            const float sc = a.scale[n0 + nl], sh = a.shift[n0 + nl];
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int r = 0; r < IgAcc<MF>::N; ++r)
                    Cs[nl * LDC + wm0 + i * MF + ig_row<MF>(r, lane)] = ig_act(acc[i][j][r] * sc + sh, a.act);
        }
        __syncthreads();
        for (int idx = tid; idx < BM * BN; idx += IG_THREADS) {
            const int nl = idx / BM, ml = idx - nl * BM;
            const int m = m0 + ml, n = n0 + nl;
            if (m < a.M && n < a.Cout) {
                const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
                a.out[(((size_t)b * a.Cout + n) * a.OH + (oy * a.osy + a.ooy)) * a.OW + (ox * a.osx + a.oox)] =
                    Cs[nl * LDC + ml];
            }
        }
    }
}
