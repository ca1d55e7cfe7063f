// OPT-IN fp32-equivalent implicit-GEMM convolution on the bf16 matrix pipe of gfx950 (CP_SPLIT_BF16=1; never the default, never
// the headline: the timed configuration stays on the exact-f32 MFMA of igemm.h).
//
// Every fp32 operand is represented EXACTLY as three bf16 terms, x = x1 + x2 + x3 (8 + 8 + 8 significand bits, round-to-nearest at
// each level: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) -- the residuals are exact in fp32), and the six significant
// cross products  x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1)  are issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulate.  A
// bf16 x bf16 product is exact in fp32; the three dropped products (x2y3, x3y2, x3y3) are <= 2^-24 + 2^-24 + 2^-32 relative to
// |xy| -- the size of ONE fp32 rounding of the product, which the f32 MFMA / an fmaf chain commits on every term anyway.  Measured
// error against an fp64 convolution: tools/bf16x3_ab.py (profiles/r5_bf16x3_ab_*.txt).
// DOMAIN: FINITE operands with |x| < 2^127.  bf16 has fp32's exponent range but rounds to nearest: the top binade's upper
// half rounds to +-inf, and for x = +-inf (or such an x) the residual x - bf16(x) is inf - inf = NaN, so the split kernel returns NaN
// where the f32 kernel propagates +-inf or a finite value; below 2^-110 the lower terms fall into bf16's subnormal range and the
// result carries fewer than 24 significand bits (absolute error < 1e-40).  Post-BN activations and weights of a network are ~2^-20 ..
// 2^20; tests/test_conv_hip.py::test_conv_split_bf16_magnitude_range pins 1e+-30.  Nothing selects this kernel by default.
// Rate: 6 bf16 MFMAs (32 cycles each, K = 16) replace 8 f32 MFMAs (64 cycles each, K = 2): 192 vs 512 matrix-pipe cycles per
// 32 x 32 x 16 tile step = 2.67x, ceiling 2516 / 6 = 419 TFLOP/s fp32-equivalent; and the split's VALU work runs BESIDE the
// bf16 matrix pipe (the f32 MFMA shares the fp32 FMA lanes with the VALU: DESIGN 7.2).
//
// Same GEMM view, producers (NHWC im2col gather of any kh / kw / stride / pad over up to 4 concatenated sources, fused sub-pixel
// deconvolution, split-K) and epilogue as igemm_conv_kernel (conv_igemm.hip).  What differs:
//   * weights arrive PRE-SPLIT (cp_split_bf16_weights_f32, once at plan time): wb [nsub][K/16][ldw][3 terms][16 k] bf16, i.e. one
//     k-step of a block's BN rows is one contiguous, fully coalesced slab that is copied to LDS as it is;
//   * activations are split in registers on the way into LDS (thread = (pixel, 8-channel half): 2 float4 loads ->
//     4 x [v_cvt_pk_bf16_f32, shift / and, v_pk_add_f32] x 2 levels + 4 v_cvt_pk -> three ds_write_b128);
//   * LDS rows are [3 terms][16 k] bf16 = 96 B + 16 B pad (stride 28 dwords: the 16-lane groups of a ds_read_b128 hit 64
//     distinct banks); a lane's MFMA operand (row i = lane & 31, k-group g = lane >> 5: 8 consecutive k) is one ds_read_b128
//     per term; A and B use the same k <-> position map, so the products pair up whatever the hardware's k order inside the
//     instruction is;
//   * a wave owns a (BM/2) x (BN/2) tile: 64 x 64 -> 12 fragment reads feed 24 MFMAs; the 64-row tiles (32 x 64 / 32 x 32 per wave, a thread
//     stages a 4-channel quarter of a pixel) exist for the small-M layers, where 128-row tiles leave CUs without a block.
#include "igemm.h"

typedef __bf16 sb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sb_v4u __attribute__((ext_vector_type(4)));

#define SB_LDR 28            // LDS row stride in dwords (3 x 8 dwords of bf16 pairs + 4 pad)
#define SB_ROW 24            // dwords of one (row, k-step) in the pre-split weights

__device__ __forceinline__ unsigned sb_cvt2(float a, float b)       // (bf16(a) | bf16(b) << 16), round-to-nearest-even: v_cvt_pk_bf16_f32
{
    const sb_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sb_bf16x2));
}
// two fp32 values -> their three bf16 terms, packed pairwise (low half = first value)
__device__ __forceinline__ void sb_split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3)
{
    p1 = sb_cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);     // exact
    p2 = sb_cvt2(r0, r1);
    p3 = sb_cvt2(r0 - __builtin_bit_cast(float, p2 << 16), r1 - __builtin_bit_cast(float, p2 & 0xffff0000u));         // exact, <= 8 bits left
}
__device__ __forceinline__ sb_v4u sb_ldg4u(const unsigned* p)
{
    return *(const __attribute__((address_space(1))) sb_v4u*)(p);
}
__device__ __forceinline__ f32x16 sb_mfma(sb_v4u a, sb_v4u b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sb_bf16x8, a), __builtin_bit_cast(sb_bf16x8, b), c, 0, 0, 0);
}

template <int BM, int BN>
struct SbTile {
    using T = IgTile<BM, BN, 2, 2, 32>;
    // A staging: BM >= 128: thread = (pixel, 8-channel HALF), BM / 128 pixels per thread, two float4 each;
    //            BM == 64:  thread = (pixel, 4-channel QUARTER), one float4 (the small-M layers: 32x32 / 16x16 maps at B = 8)
    static constexpr bool QUARTER = BM == 64;
    static constexpr int A_SLOTS = QUARTER ? 1 : BM * 2 / IG_THREADS;
    static constexpr int B_CH = BN * 6;                                              // 16-byte chunks of one B slice
    static constexpr int B_SLOTS = (B_CH + IG_THREADS - 1) / IG_THREADS;
    static constexpr int A_DW = BM * SB_LDR, B_DW = BN * SB_LDR;
    static constexpr int MAIN_BYTES = 2 * (A_DW + B_DW) * 4;
    static constexpr int NHWC_BYTES = MAIN_BYTES > T::EPV_BYTES ? MAIN_BYTES : T::EPV_BYTES;
    static constexpr int NCHW_BYTES = MAIN_BYTES > T::EPI_BYTES ? MAIN_BYTES : T::EPI_BYTES;
    static_assert(BM == 64 || BM % 128 == 0, "BM = 64 or a multiple of 128");
};

// MFMAs of one k-step held in LDS; `mid` (the next slice's global loads) is called after half of the output tiles
template <class S, class F>
__device__ __forceinline__ void sb_compute(const unsigned* As, const unsigned* Bs, int wm0, int wn0, int lane,
                                           f32x16 (&acc)[S::T::TM][S::T::TN], F&& mid)
{
    constexpr int TM = S::T::TM, TN = S::T::TN;
    const int g = lane >> 5, il = lane & 31;
    sb_v4u af[TM][3], bf[TN][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][t] = *reinterpret_cast<const sb_v4u*>(As + (wm0 + i * 32 + il) * SB_LDR + t * 8 + g * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j][t] = *reinterpret_cast<const sb_v4u*>(Bs + (wn0 + j * 32 + il) * SB_LDR + t * 8 + g * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (i * TN + j == (TM * TN) / 2) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            // smallest products first (the order inside a k-step is free; this one keeps the low-order sums together)
            acc[i][j] = sb_mfma(af[i][2], bf[j][0], acc[i][j]);
            acc[i][j] = sb_mfma(af[i][1], bf[j][1], acc[i][j]);
            acc[i][j] = sb_mfma(af[i][0], bf[j][2], acc[i][j]);
            acc[i][j] = sb_mfma(af[i][1], bf[j][0], acc[i][j]);
            acc[i][j] = sb_mfma(af[i][0], bf[j][1], acc[i][j]);
            acc[i][j] = sb_mfma(af[i][0], bf[j][0], acc[i][j]);
        }
}

struct SbPix { int boff, iy0, ix0; };

template <int BM, int BN>
__global__ __launch_bounds__(IG_THREADS, 2) void igemm_bf16x3_kernel(const ConvArgs a)
{
    using S = SbTile<BM, BN>;
    using T = typename S::T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* As0 = reinterpret_cast<unsigned*>(smem);          // [2][BM][SB_LDR]
    unsigned* Bs0 = As0 + 2 * S::A_DW;                           // [2][BN][SB_LDR]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int tile = ig_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned* wsub = reinterpret_cast<const unsigned*>(a.w);
    int pys = a.py, pxs = a.px, ooys = 0, ooxs = 0;
    const int nk_all = a.K / IG_BK;
    if (a.nsub > 1) {                                            // fused sub-pixel deconvolution: see igemm_conv_kernel
        const int per = gridDim.x / a.nsub, sub = tile / per;
        tile -= sub * per;
        wsub += (size_t)sub * nk_all * a.ldw * SB_ROW;
        pys -= sub >> 1; pxs -= sub & 1;
        ooys = sub >> 1; ooxs = sub & 1;
    }
    const int SK = a.ksplit, ksp = SK > 1 ? tile % SK : 0;       // split-K: raw partial sums to out + ksp*M*outLd
    if (SK > 1) tile /= SK;
    const int NT = a.ldw / BN;
    const int nt = tile % NT, mt = tile / NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wid >> 1) * T::WM, wn0 = (wid & 1) * T::WN;
    const int HoWo = a.Ho * a.Wo;

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ks0 = ksp * nk_all / SK, nk = (ksp + 1) * nk_all / SK;

    // ---- A producer: thread -> (pixel, 8-channel half), or (pixel, 4-channel quarter) for the 64-row tile ------------------
    constexpr int QS = S::QUARTER ? 4 : 2;                      // threads per pixel
    const int q = tid & (QS - 1);
    SbPix ps[S::A_SLOTS];
#pragma unroll
    for (int s = 0; s < S::A_SLOTS; ++s) {
        const int m = m0 + tid / QS + s * 128;
        if (m < a.M) {
            const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
            ps[s].boff = b * a.H * a.W;
            ps[s].iy0 = oy * a.sy - pys;
            ps[s].ix0 = ox * a.sx - pxs;
        } else { ps[s].boff = -1; ps[s].iy0 = 0; ps[s].ix0 = 0; }
    }
    float4 ar[S::A_SLOTS][2];
    bool aok[S::A_SLOTS];
    unsigned aoff[S::A_SLOTS];
    sb_v4u br[S::B_SLOTS];
    int ky = 0, kx = 0, si = 0, cl = 0;                          // wave-uniform k-walk: tap, source, channel offset in the source
    if (SK > 1) {
        const int per_tap = a.srcC[0] / IG_BK, tap = ks0 / per_tap;
        cl = (ks0 - tap * per_tap) * IG_BK;
        ky = tap / a.kw; kx = tap - ky * a.kw;
    }
    const float* sp = a.src[0];
    int ld = a.srcLd[0], sC = a.srcC[0];
    bool fresh = true;
    auto load_a = [&]() __attribute__((always_inline)) {
        if (cl == 0 || fresh) {
            fresh = false;
#pragma unroll
            for (int s = 0; s < S::A_SLOTS; ++s) {
                const int iy = ps[s].iy0 + ky, ix = ps[s].ix0 + kx;
                const bool ok = ps[s].boff >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                aoff[s] = ok ? ((unsigned)(ps[s].boff + iy * a.W + ix) * (unsigned)ld + (unsigned)q * (S::QUARTER ? 4u : 8u)) * 4u : 0u;
                aok[s] = ok;
            }
        }
        const char* xs = reinterpret_cast<const char*>(sp) + (size_t)cl * 4;      // uniform
#pragma unroll
        for (int s = 0; s < S::A_SLOTS; ++s) {
            ar[s][0] = ig_ldg4(reinterpret_cast<const float*>(xs + aoff[s]));
            if constexpr (!S::QUARTER) ar[s][1] = ig_ldg4(reinterpret_cast<const float*>(xs + aoff[s]) + 4);
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        cl += IG_BK;
        if (cl >= sC) {
            cl = 0;
            if (++si >= a.nsrc) { si = 0; if (++kx >= a.kw) { kx = 0; ++ky; } }
            if (a.nsrc > 1) { sp = a.src[si]; ld = a.srcLd[si]; sC = a.srcC[si]; }
        }
    };
    auto load_b = [&](int ks) __attribute__((always_inline)) {
        const unsigned* wk = wsub + ((size_t)ks * a.ldw + n0) * SB_ROW;           // BN rows of this k-step: one contiguous slab
#pragma unroll
        for (int s = 0; s < S::B_SLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            if (S::B_CH % IG_THREADS == 0 || idx < S::B_CH) br[s] = sb_ldg4u(wk + idx * 4);
        }
    };
    auto store_ab = [&](unsigned* As, unsigned* Bs) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S::A_SLOTS; ++s) {
            const float4 v0 = aok[s] ? ar[s][0] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (S::QUARTER) {
                unsigned t1[2], t2[2], t3[2];
                sb_split2(v0.x, v0.y, t1[0], t2[0], t3[0]); sb_split2(v0.z, v0.w, t1[1], t2[1], t3[1]);
                unsigned* row = As + (tid >> 2) * SB_LDR + q * 2;
                *reinterpret_cast<uint2*>(row) = make_uint2(t1[0], t1[1]);
                *reinterpret_cast<uint2*>(row + 8) = make_uint2(t2[0], t2[1]);
                *reinterpret_cast<uint2*>(row + 16) = make_uint2(t3[0], t3[1]);
            } else {
                const float4 v1 = aok[s] ? ar[s][1] : make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned t1[4], t2[4], t3[4];
                sb_split2(v0.x, v0.y, t1[0], t2[0], t3[0]); sb_split2(v0.z, v0.w, t1[1], t2[1], t3[1]);
                sb_split2(v1.x, v1.y, t1[2], t2[2], t3[2]); sb_split2(v1.z, v1.w, t1[3], t2[3], t3[3]);
                unsigned* row = As + ((tid >> 1) + s * 128) * SB_LDR + q * 4;
                *reinterpret_cast<sb_v4u*>(row) = (sb_v4u){t1[0], t1[1], t1[2], t1[3]};
                *reinterpret_cast<sb_v4u*>(row + 8) = (sb_v4u){t2[0], t2[1], t2[2], t2[3]};
                *reinterpret_cast<sb_v4u*>(row + 16) = (sb_v4u){t3[0], t3[1], t3[2], t3[3]};
            }
        }
#pragma unroll
        for (int s = 0; s < S::B_SLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            if (S::B_CH % IG_THREADS == 0 || idx < S::B_CH) *reinterpret_cast<sb_v4u*>(Bs + (idx / 6) * SB_LDR + (idx % 6) * 4) = br[s];
        }
    };

    load_a(); advance();
    load_b(ks0);
    store_ab(As0, Bs0);
    __syncthreads();
    int cur = 0;
    for (int ks = ks0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        sb_compute<S>(As0 + cur * S::A_DW, Bs0 + cur * S::B_DW, wm0, wn0, lane, acc, [&]() __attribute__((always_inline)) {
            if (more) { load_a(); advance(); load_b(ks + 1); }
        });
        __builtin_amdgcn_sched_barrier(0);      // nothing that consumes the prefetched registers above the MFMAs (vmcnt wait)
        if (more) store_ab(As0 + (cur ^ 1) * S::A_DW, Bs0 + (cur ^ 1) * S::B_DW);
        __syncthreads();
        cur ^= 1;
    }
    if (SK > 1) {
        ConvArgs e = a;
        e.out = a.out + (size_t)ksp * a.M * a.outLd;
        ig_epilogue<T, BM, BN, 32>(e, smem, m0, n0, wm0, wn0, lane, tid, acc, ooys, ooxs);
    } else ig_epilogue<T, BM, BN, 32>(a, smem, m0, n0, wm0, wn0, lane, tid, acc, ooys, ooxs);
}

template <int BM, int BN>
static int launch_sb(const ConvArgs& a, hipStream_t s)
{
    using S = SbTile<BM, BN>;
    auto kern = igemm_bf16x3_kernel<BM, BN>;
    if (a.ldw % BN != 0) { cp_set_error("conv2d (split-bf16): ldw=%d is not a multiple of the N tile %d", a.ldw, BN); return 1; }
    const int smem = a.outNCHW ? S::NCHW_BYTES : S::NHWC_BYTES;
    static CpLdsGuard guard;
    constexpr int smem_max = S::NCHW_BYTES > S::NHWC_BYTES ? S::NCHW_BYTES : S::NHWC_BYTES;
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem_max);
        if (e != hipSuccess) { cp_set_error("conv2d (split-bf16): cannot reserve %d B LDS: %s", smem_max, hipGetErrorString(e)); return 2; }
    }
    const int grid = cp_cdiv(a.M, BM) * (a.ldw / BN) * (a.nsub > 1 ? a.nsub : 1) * a.ksplit;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(IG_THREADS), smem, s, a);
    cp_note_kernel("igemm_bf16x3_kernel<%d, %d>", BM, BN);
    return 0;
}

// conv_igemm.hip dispatches here for cp_conv_desc.tile = 3000000 + BM*1000 + BN (a.w = the pre-split weights)
int cp_launch_conv_bf16x3(const ConvArgs& a, int tile, hipStream_t s)
{
    if (a.ksplit > 1 && (a.nsub > 1 || a.nsrc != 1 || a.ksplit > a.K / IG_BK)) {
        cp_set_error("conv2d (split-bf16): ksplit=%d needs one NHWC source, nsub = 1 and at most K/16 = %d splits", a.ksplit, a.K / IG_BK);
        return 1;
    }
    for (int i = 0; i < a.nsrc; ++i)
        if (a.srcLd[i] % 4 != 0 || (((size_t)a.src[i]) & 15) != 0) { cp_set_error("conv2d (split-bf16): source %d is not 16-byte aligned", i); return 1; }
    switch (tile) {
        case 3128128: return launch_sb<128, 128>(a, s);
        case 3128064: return launch_sb<128, 64>(a, s);
        case 3064128: return launch_sb<64, 128>(a, s);
        case 3064064: return launch_sb<64, 64>(a, s);
        default: cp_set_error("conv2d (split-bf16): unknown tile %d", tile); return 1;
    }
}

// ---- weights: fp32 [rows][K] (rows = nsub * ldw, n-major, k contiguous: what ops.pack_conv_weight makes) -> pre-split bf16
// [nsub][K/16][ldw][3][16], stored as dwords (pairs of consecutive k, low half = even k)
__global__ void split_bf16_weights_kernel(const float* __restrict__ w, unsigned* __restrict__ wb, int ldw, int K, long long total)
{
    // one thread = one output dword = (sub, kc, n, term, pair)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pair = (int)(i & 7), term = (int)((i >> 3) % 3);
        const long long rowi = i / SB_ROW;                           // (sub * nk + kc) * ldw + n
        const int n = (int)(rowi % ldw);
        const long long skc = rowi / ldw;
        const int nk = K / 16, kc = (int)(skc % nk), sub = (int)(skc / nk);
        const float* src = w + ((size_t)sub * ldw + n) * K + kc * 16 + pair * 2;
        unsigned p1, p2, p3;
        sb_split2(src[0], src[1], p1, p2, p3);
        wb[i] = term == 0 ? p1 : term == 1 ? p2 : p3;
    }
}

extern "C" size_t cp_split_bf16_weight_floats(int rows, int K) { return (size_t)rows * (K / 16) * SB_ROW; }

extern "C" int cp_split_bf16_weights_f32(const float* w, int rows, int ldw, int K, float* wb, void* stream)
{
    CP_CHECK_ARG(w && wb && ldw > 0 && rows % ldw == 0 && K > 0 && K % 16 == 0, "split_bf16_weights: rows=%d ldw=%d K=%d", rows, ldw, K);
    const long long total = (long long)rows * (K / 16) * SB_ROW;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(split_bf16_weights_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, reinterpret_cast<unsigned*>(wb), ldw, K, total);
    CP_CHECK_LAUNCH("split_bf16_weights_kernel");
    return 0;
}
