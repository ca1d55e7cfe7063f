// Implicit-GEMM convolution core for gfx950, fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s dense peak).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * W[n][k]
//   m = (b, oy, ox) output pixel, n = output channel, k = (tap, input channel) with the input
//   channel fastest (activations are NHWC, so a k-chunk of one pixel is contiguous in HBM).
// Block tile BM x BN, k-step BK = 16, 256 threads = 4 wave64.  Per k-step:
//   * the A-producer (plain im2col gather, 3-channel NCHW stem gather, or DCNv2 bilinear sampler)
//     loads its slice into registers while the previous slice is being multiplied,
//   * both operands are staged K-CONTIGUOUS in LDS (As[m][k], Bs[n][k], row stride 20 floats):
//     the global float4 goes to LDS with one ds_write_b128, no transpose;
//   * fragments are read with ds_read_b128: lane (row i, k-group g) takes 4 consecutive k and
//     feeds them to 4 successive MFMAs.  MFMA number s therefore multiplies the k-set
//     {4g+s : g} -- a permutation of the k order that A and B share, so the sum is unchanged.
//     One LDS instruction per operand per 4 MFMAs; the 20-float stride makes the 16-lane b128
//     groups hit 64 distinct banks.
// Two LDS buffers, one barrier per k-step.  Epilogue fuses scale/shift (folded BN or bias),
// residual add, ReLU / sigmoid and writes NHWC (coalesced along n) or NCHW (transposed through
// LDS so stores are coalesced along the pixel index).
#pragma once
#include "common.h"

#define IG_BK 16
#define IG_LDK 20            // LDS row stride in floats (16 B aligned; conflict-free b128 reads)
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
    int K;                    // kh*kw*Ctot rounded up to a multiple of 16 (zero weight columns)
    const float* w;           // packed weights [ldw][K]  (n-major, k contiguous)
    int ldw;                  // Cout padded to the N tile (zero rows)
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
    int ksplit;               // > 1 = split-K: block (tile, split s) accumulates its share of the reduction (DCNv2: taps [s*T/S, (s+1)*T/S); generic
                              // kernels: k-steps [s*nk/S, (s+1)*nk/S)) and stores the RAW partial sums to out + s*M*outLd
                              // (cp_splitk_reduce_f32 sums them in a fixed order)
    int dg;                   // DCNv2 only: deformable groups (>= 1)
    int nsub;                 // 1, or 4: the four sub-pixel 2x2 convs of a k4/s2/p1 ConvTranspose2d in ONE launch (generic kernel only):
                              // sub g = py*2+px uses weights w + g*ldw*K, pad (py0 - py, px0 - px) and output phase (ooy + py, oox + px)
};

// conv3x3_patch.hip: returns -1 when the shape is not eligible, else 0 / error code
int cp_launch_conv3x3_patch(const ConvArgs& a, int in_nchw, hipStream_t s, int variant = 0);
// stem7x7.hip: 3x3 / stride 2 / pad 1 stem on the 3-channel NCHW network input, 64 outputs (persistent weights-in-registers kernel): same convention
int cp_launch_stem3x3(const ConvArgs& a, hipStream_t s);
// conv3x3_c16.hip (16 input channels, stride 1 or 2, 16 / 32 outputs): same convention
int cp_launch_conv3x3_c16(const ConvArgs& a, int in_nchw, hipStream_t s);
// conv3x3_wino.hip (a.w = Winograd-domain weights): same convention
int cp_launch_conv3x3_wino(const ConvArgs& a, hipStream_t s, int variant = 0);
int cp_launch_conv3x3_wino24_group(const ConvArgs* a, int n, hipStream_t s);      // up to four independent convolutions in one launch
// head_wino24.hip: the same head branch on the F(2x4,3x3) transform (a.w from cp_winograd24_pack_f32), eight waves per block; -1 = not eligible
int cp_launch_head3x3_1x1_w24(const ConvArgs& a, const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, hipStream_t s);
// conv3x3_wino24.hip (a.w = F(2x4,3x3) Winograd-domain weights from cp_winograd24_pack_f32): same convention
int cp_launch_conv3x3_wino24(const ConvArgs& a, hipStream_t s);
// conv3x3_wino.hip: [3x3 + bias + ReLU] + 1x1 (n2 <= 2 outputs, NCHW) of a head branch in one launch; -1 = shape not eligible
int cp_launch_head3x3_1x1(const ConvArgs& a, const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, hipStream_t s);

// conv_pointwise.hip: 1x1 / stride 1 with K = 64 from one NHWC source, whole operand tiles + residual requested up front (bit-identical to the
// generic kernel).  -1 = not eligible
int cp_launch_conv_pointwise(const ConvArgs& a, hipStream_t s);
// conv_igemm_bf16x3.hip: the generic implicit GEMM as an fp32-equivalent 3-term split on the bf16 matrix pipe (opt-in; a.w = pre-split weights)
int cp_launch_conv_bf16x3(const ConvArgs& a, int tile, hipStream_t s);

// 16-byte load through an explicit GLOBAL address-space pointer.  A generic (flat) load also increments
// lgkmcnt, so every `s_waitcnt lgkmcnt` guarding an LDS fragment read would wait for the prefetch as well.
typedef float ig_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ig_ldg4(const float* p)
{
    const ig_v4f v = *(const __attribute__((address_space(1))) ig_v4f*)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// XCD-aware, bijective remap of the linear block id: blocks that are consecutive in the
// remapped order (and share the same activation rows) land on the same XCD / L2.
__device__ __forceinline__ int ig_xcd_remap(int id, int n)
{
    const int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

template <int MF> struct IgAcc;
template <> struct IgAcc<32> { typedef f32x16 type; static constexpr int N = 16; };
template <> struct IgAcc<16> { typedef f32x4 type; static constexpr int N = 4; };

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
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / MF, TN = WN / MF;
    static constexpr int A_SLOTS = BM * 4 / IG_THREADS;                  // float4 per thread per k-step
    static constexpr int B_F4 = BN * 4;                                  // float4 in one B slice
    static constexpr int B_SLOTS = (B_F4 + IG_THREADS - 1) / IG_THREADS;
    static constexpr int A_FLOATS = BM * IG_LDK, B_FLOATS = BN * IG_LDK;
    static constexpr int MAIN_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int EPI_BYTES = BN * (BM + 1) * 4;                  // NCHW transposed epilogue Cs[n][m]
    static constexpr int EPV_BYTES = BM * (BN + 4) * 4;                  // NHWC vectorised epilogue Cs[m][n]
    static constexpr int NHWC_BYTES = MAIN_BYTES > EPV_BYTES ? MAIN_BYTES : EPV_BYTES;
    static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(WAVES_M * WAVES_N * 64 == IG_THREADS, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    static_assert(A_SLOTS >= 1, "BM >= 64");
};

// ---- B (weights) slice: global [n][K] -> regs -> LDS Bs[n][k] --------------------------------
template <class T>
__device__ __forceinline__ void ig_load_b(const ConvArgs& a, int k0, int n0, int tid, float4 (&br)[T::B_SLOTS], const float* w = nullptr)
{
    if (!w) w = a.w;
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        if constexpr (T::B_F4 % IG_THREADS == 0) br[s] = ig_ldg4(w + (size_t)(n0 + (idx >> 2)) * a.K + k0 + (idx & 3) * 4);
        else if (idx < T::B_F4) br[s] = ig_ldg4(w + (size_t)(n0 + (idx >> 2)) * a.K + k0 + (idx & 3) * 4);
    }
}
template <class T>
__device__ __forceinline__ void ig_store_b(float* Bs, int tid, const float4 (&br)[T::B_SLOTS])
{
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        if (T::B_F4 % IG_THREADS == 0 || idx < T::B_F4) *reinterpret_cast<float4*>(Bs + (idx >> 2) * IG_LDK + (idx & 3) * 4) = br[s];
    }
}

// ---- MFMA over one k-step held in LDS --------------------------------------------------------
// `mid` is called once after half of the k-step's MFMAs have been issued: the place for the next slice's global loads.
// Issued in FRONT of an MFMA block they cost up to 20 % of the matrix rate at two waves per SIMD, from inside it nothing
// (tools/micro/vs_loop.hip).
template <class T, int MF, class F>
__device__ __forceinline__ void ig_compute(const float* As, const float* Bs, int wm0, int wn0, int lane,
                                           typename IgAcc<MF>::type (&acc)[T::TM][T::TN], F&& mid)
{
    constexpr int G = 64 / MF;           // k-groups across the wave: 2 (32x32x2) or 4 (16x16x4)
    constexpr int NH = IG_BK / (4 * G);  // b128 reads per operand row per k-step: 2 or 1
    const int g = lane / MF, il = lane % MF;
    float4 af[NH][T::TM], bf[NH][T::TN];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
            af[h][i] = *reinterpret_cast<const float4*>(As + (wm0 + i * MF + il) * IG_LDK + h * 4 * G + g * 4);
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
            bf[h][j] = *reinterpret_cast<const float4*>(Bs + (wn0 + j * MF + il) * IG_LDK + h * 4 * G + g * 4);
    }
    constexpr int HALF = NH * T::TM * T::TN / 2;       // (h, i, j) tiles before the hook
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) {
                if ((h * T::TM + i) * T::TN + j == HALF) {
                    __builtin_amdgcn_sched_barrier(0);
                    mid();
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[i][j] = ig_mfma<MF>(af[h][i].x, bf[h][j].x, acc[i][j]);
                acc[i][j] = ig_mfma<MF>(af[h][i].y, bf[h][j].y, acc[i][j]);
                acc[i][j] = ig_mfma<MF>(af[h][i].z, bf[h][j].z, acc[i][j]);
                acc[i][j] = ig_mfma<MF>(af[h][i].w, bf[h][j].w, acc[i][j]);
            }
    }
}

template <class T, int MF>
__device__ __forceinline__ void ig_compute(const float* As, const float* Bs, int wm0, int wn0, int lane,
                                           typename IgAcc<MF>::type (&acc)[T::TM][T::TN])
{
    ig_compute<T, MF>(As, Bs, wm0, wn0, lane, acc, [] {});
}

__device__ __forceinline__ float ig_act(float v, int act) { return cp_act(v, act); }

// ---- epilogue --------------------------------------------------------------------------------
template <class T, int BM, int BN, int MF>
__device__ __forceinline__ void ig_epilogue(const ConvArgs& a, float* smem, int m0, int n0, int wm0, int wn0,
                                            int lane, int tid, typename IgAcc<MF>::type (&acc)[T::TM][T::TN],
                                            int ooy_add = 0, int oox_add = 0)      // extra output phase (fused sub-pixel deconv)
{
    const int HoWo = a.Ho * a.Wo;
    const int ooy = a.ooy + ooy_add, oox = a.oox + oox_add;
    const int cl = lane & (MF - 1);
    const bool vec_ok = !a.outNCHW && ((a.outLd | a.Cout) & 3) == 0 && (((size_t)a.out) & 15) == 0 &&
                        (!a.res || ((a.resLd & 3) == 0 && (((size_t)a.res) & 15) == 0));
    if (vec_ok) {
        // NHWC, vectorised: the C tile goes through LDS (Cs[m][n]) so that every thread handles float4
        // runs along n: 16-byte residual loads and stores instead of 4-byte ones (4x fewer VMEM
        // instructions; memory-bound 1x1 layers are epilogue-dominated)
        const bool dense = (a.osy == 1 && a.osx == 1 && ooy == 0 && oox == 0 && a.OH == a.Ho && a.OW == a.Wo);
        const bool relu = a.act == CP_ACT_RELU, sigm = a.act == CP_ACT_SIGMOID;
        float* Cs = smem;
        constexpr int LDC = BN + 4;
        __syncthreads();   // main-loop buffers are dead
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < IgAcc<MF>::N; ++r)
                    Cs[(wm0 + i * MF + ig_row<MF>(r, lane)) * LDC + wn0 + j * MF + cl] = acc[i][j][r];
        __syncthreads();
        constexpr int NV = BN / 4;
#pragma unroll
        for (int s = 0; s < BM * NV / IG_THREADS; ++s) {
            const int idx = tid + s * IG_THREADS;
            const int row = idx / NV, c4 = idx - row * NV;
            const int m = m0 + row, n = n0 + c4 * 4;
            if (m < a.M && n < a.Cout) {
                size_t opix = (size_t)m;
                if (!dense) {
                    const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
                    opix = ((size_t)b * a.OH + (oy * a.osy + ooy)) * a.OW + (ox * a.osx + oox);
                }
                float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4);
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + n);
                const float4 sh = *reinterpret_cast<const float4*>(a.shift + n);
                v = cp_scale_shift4(v, sc, sh);
                if (a.res) {
                    const float4 rr = *reinterpret_cast<const float4*>(a.res + opix * a.resLd + n);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (relu) v = cp_relu4(v);
                else if (sigm) {
                    v.x = 1.0f / (1.0f + __expf(-v.x)); v.y = 1.0f / (1.0f + __expf(-v.y));
                    v.z = 1.0f / (1.0f + __expf(-v.z)); v.w = 1.0f / (1.0f + __expf(-v.w));
                } else if (a.act != CP_ACT_NONE) { v.x = cp_act(v.x, a.act); v.y = cp_act(v.y, a.act); v.z = cp_act(v.z, a.act); v.w = cp_act(v.w, a.act); }
                *reinterpret_cast<float4*>(a.out + opix * a.outLd + n) = v;
            }
        }
    } else if (!a.outNCHW) {
        const bool dense = (a.osy == 1 && a.osx == 1 && ooy == 0 && oox == 0 && a.OH == a.Ho && a.OW == a.Wo);
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < IgAcc<MF>::N; ++r) {
                const int m = m0 + wm0 + i * MF + ig_row<MF>(r, lane);
                if (m >= a.M) continue;
                size_t opix = (size_t)m;
                if (!dense) {
                    const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
                    opix = ((size_t)b * a.OH + (oy * a.osy + ooy)) * a.OW + (ox * a.osx + oox);
                }
                float* orow = a.out + opix * a.outLd;
                const float* rrow = a.res ? a.res + opix * a.resLd : nullptr;
#pragma unroll
                for (int j = 0; j < T::TN; ++j) {
                    const int n = n0 + wn0 + j * MF + cl;
                    if (n < a.Cout) {
                        float v = acc[i][j][r] * a.scale[n] + a.shift[n];
                        if (rrow) v += rrow[n];
                        v = cp_act(v, a.act);
                        orow[n] = v;
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
                a.out[(((size_t)b * a.Cout + n) * a.OH + (oy * a.osy + ooy)) * a.OW + (ox * a.osx + oox)] =
                    Cs[nl * LDC + ml];
            }
        }
    }
}
