// 7x7 stem convolution on the 3-channel NCHW network input (gfx950, fp32 MFMA 16x16x4).
//
//   DLA-34   base_layer: Conv2d(3,16,k7,s1,p3)+BN+ReLU  (pose_dla_dcn.py:228-232)  @512x512
//   ResNet-50 conv1    : Conv2d(3,64,k7,s2,p3)+BN+ReLU  (msra_resnet.py:118-121)
// K = 147 is tiny and the input has 3 planes, so the generic im2col path (16 scalar gathers per
// thread per k-step) ran at 19 TFLOP/s.  Here the block stages the 3 x (rows+6) x (cols+6) input
// window once (coalesced row reads straight from NCHW), and MFMA A-fragments are read from it:
// the k axis is re-ordered as k = ((c*7 + ky)*8 + kx) (kx padded 7 -> 8 with zero weights), so the
// 4 consecutive k a lane feeds to 4 successive 16x16x4 MFMAs are 4 consecutive floats of one patch
// row.  Weights for all 11 k16-steps sit in LDS ([11][NOUT][20], b128 reads).  Output is NHWC.
#include "igemm.h"

#define S7_STEPS 11          // 21 (c,ky) rows, 2 per k16-step, last half-step zero
#define S7_K (S7_STEPS * 16)

template <int NOUT, int S, int TH, int TW>
__global__ __launch_bounds__(IG_THREADS) void stem7x7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ out, int B, int H, int W, int Ho, int Wo,
                                                              int outLd, int relu, int tilesX, int tilesY)
{
    constexpr int PH = (TH - 1) * S + 7, PWr = (TW - 1) * S + 7, PW = (PWr + 3) & ~3;   // patch rows / padded pitch
    constexpr int TN = NOUT / 16;
    constexpr int ROWS_W = TH / 4;                 // output rows per wave
    constexpr int TM = ROWS_W * (TW / 16);         // 16-pixel MFMA tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                           // [3][PH][PW]
    float* Bs = smem + ((3 * PH * PW + 3) & ~3);   // [S7_STEPS][NOUT][IG_LDK]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int t_ = blockIdx.x;
    const int tx = t_ % tilesX; t_ /= tilesX;
    const int ty = t_ % tilesY;
    const int b = t_ / tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 3, ix0 = ox0 * S - 3;

    // ---- stage the input window (zero padded) and the packed weights
    for (int idx = tid; idx < 3 * PH * PW; idx += IG_THREADS) {
        const int px = idx % PW, r = idx / PW, py = r % PH, c = r / PH;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (px < PWr && iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((size_t)(b * 3 + c) * H + iy) * W + ix];
        patch[idx] = v;
    }
    for (int idx = tid; idx < S7_STEPS * NOUT * 4; idx += IG_THREADS) {
        const int q = idx & 3, n = (idx >> 2) % NOUT, st = idx / (NOUT * 4);
        *reinterpret_cast<float4*>(Bs + (st * NOUT + n) * IG_LDK + q * 4) =
            *reinterpret_cast<const float4*>(w + (size_t)n * S7_K + st * 16 + q * 4);
    }
    __syncthreads();

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, il = lane & 15;
    // lane's A base inside a (c,ky) patch row: column of pixel il plus its 4-wide kx group
    const int a_col = il * S + (g & 1) * 4;
#pragma unroll          // 11 steps, fully unrolled: (c, ky) of a step and the row offsets become compile-time constants per k-half
    for (int st = 0; st < S7_STEPS; ++st) {
        const int row = 2 * st + (g >> 1);           // (c*7 + ky); row 21 (last half-step) has zero weights
        const int c = row / 7, ky = row - c * 7;
        const bool rok = row < 21;
        float4 bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs + (st * NOUT + j * 16 + il) * IG_LDK + g * 4);
        // all A fragments of the step are read first (TM x 4 floats), the MFMAs follow behind a scheduling
        // barrier: one exposed LDS round trip per step instead of one per 16-pixel tile
        float av[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int orow = wid * ROWS_W + i / (TW / 16), ocol = (i % (TW / 16)) * 16;
            const float* p = patch + ((rok ? c : 0) * PH + orow * S + (rok ? ky : 0)) * PW + ocol * S + a_col;
            av[i][0] = p[0]; av[i][1] = p[1]; av[i][2] = p[2]; av[i][3] = p[3];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][0], bf[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][1], bf[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][2], bf[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][3], bf[j].w, acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: folded BN + ReLU, NHWC
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oy = oy0 + wid * ROWS_W + i / (TW / 16);
        const int oxb = ox0 + (i % (TW / 16)) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ox = oxb + g * 4 + r;
            if (oy >= Ho || ox >= Wo) continue;
            float* orow = out + ((size_t)(b * Ho + oy) * Wo + ox) * outLd;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = j * 16 + il;
                float v = acc[i][j][r] * scale[n] + shift[n];
                if (relu) v = fmaxf(v, 0.f);
                orow[n] = v;
            }
        }
    }
}

// ---- 16 outputs / stride 1 (DLA-34 base_layer @512x512): weights-in-registers, persistent kernel ---------------------------------
// Same organisation as conv3x3_c16.hip.  The weights are the MFMA A operand, loaded once per block into 37 VGPRs: k is the plain
// (c, ky, kx) index, 147 -> 148 = 37 MFMAs per 16 pixels (the kernel above pads kx to 8: 44 steps of 4).  A lane's B operand for MFMA m
// is ONE float of the LDS window, patch[c][y + ky][x + kx] with (c, ky, kx) = k = 4m + (lane >> 4): its address is a per-lane
// constant (37 more VGPRs, computed once per block) + an immediate for the tile.  D = W x P leaves a lane with four consecutive channels
// of one pixel: folded BN + ReLU + one float4 NHWC store, no LDS epilogue.  Three blocks per CU walk the block tiles; the next
// tile's window (3 x 14 x 72 floats, three float4 per thread, for the DLA shape) is requested before the current tile's MFMAs and parked in registers.
// NOUT = 16 / stride 1 (DLA): 8x64-pixel block tiles, a wave owns two output rows.  NOUT = 64 / stride 2 (ResNet conv1, msra_resnet.py:118-121 /
// resnet_dcn.py): 4x64-pixel tiles, a wave owns 16 of the 64 output CHANNELS for all rows (37 weight registers per wave either way);
// the window's pixels are then read at stride 2 (even / odd LDS banks: conflict-free).
typedef float s7_v4 __attribute__((ext_vector_type(4)));
template <int NOUT, int S, int KS>
struct S7C {
    static constexpr int TH = NOUT == 16 ? 8 : 4, TW = 64;
    static constexpr int PAD = KS / 2, CO = 4 - PAD;                // window column of input column S * ox0 - PAD (the window starts 4 columns left of S * ox0)
    static constexpr int PH = (TH - 1) * S + KS;                    // window rows per plane
    static constexpr int PW = ((TW - 1) * S + KS + CO + 3) & ~3;    // floats per window row, first column S * ox0 - 4 (16-byte aligned in the image): 72 / 136 / 132
    static constexpr int F4 = 3 * PH * (PW / 4);
    static constexpr int SLOTS = (F4 + IG_THREADS - 1) / IG_THREADS;
    static constexpr int RW = NOUT == 16 ? TH / 4 : TH;             // output rows a wave computes
    static constexpr int OCC = NOUT == 16 ? 3 : (KS == 3 ? 4 : 2);  // resident blocks per CU (7x7: 168 / 256 VGPRs, no spills with six prefetch slots; 3x3: 96 VGPRs)
    static constexpr int KTOT = 3 * KS * KS, NM = (KTOT + 3) / 4;   // 147 -> 37 MFMAs per 16 pixels, 27 -> 7
    static constexpr int KSP = KS == 7 ? 8 : KS;                    // kx pitch of the weight pack: [n][(c*7+ky)*8 + kx] (pack_stem7) / [n][(c*KS+ky)*KS + kx] (generic)
    static_assert((NOUT == 16 || NOUT == 64) && (S == 1 || S == 2) && (KS == 7 || KS == 3), "stem shapes");
};

template <int NOUT, int S, int KS>
__global__ __launch_bounds__(IG_THREADS, (S7C<NOUT, S, KS>::OCC)) void stem7x7_c16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      float* __restrict__ out, int B, int H, int W, int Ho, int Wo, int outLd, int relu,
                                                                      int tilesX, int tilesY, int ntiles, int kpack)
{
    typedef S7C<NOUT, S, KS> G;
    __shared__ __attribute__((aligned(16))) float patch[3 * G::PH * G::PW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nset = NOUT == 16 ? 0 : wid;                    // this wave's 16 output channels
    const int wrow = NOUT == 16 ? wid * G::RW : 0;            // this wave's first output row inside the tile
    const int slot = ig_xcd_remap(blockIdx.x, gridDim.x);

    // window staging: float4 `idx` = (plane row pr = c * PH + py, quad f): per-thread constants, per tile only scalars change
    // (two registers per slot: the global offset and (py, f) packed; the LDS offset is (tid + 256 s) * 16 bytes -- the prefetch registers
    // share the file with 74 weight / address registers, and a spill reload in the loop would wait for the whole prefetch)
    int s_g[G::SLOTS], s_pf[G::SLOTS];
#pragma unroll
    for (int s = 0; s < G::SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        const int pr = idx / (G::PW / 4), f = idx - pr * (G::PW / 4);
        const int c = pr / G::PH, py = pr - c * G::PH;
        s_g[s] = (c * H + py) * W + f * 4;
        s_pf[s] = py | (f << 10);
    }
    float4 v[G::SLOTS];
    auto load_window = [&](int tl) {
        const int tx = tl % tilesX, r_ = tl / tilesX;
        const int ty = r_ % tilesY, b = r_ / tilesY;
        const int iy0 = ty * G::TH * S - G::PAD, ix0 = tx * G::TW * S - 4;
        const float* xb = x + ((long long)b * 3 * H + iy0) * W + ix0;            // scalar; may point before the image (never loaded from)
#pragma unroll
        for (int s = 0; s < G::SLOTS; ++s) {
            const int yy = iy0 + (s_pf[s] & 1023), xx = ix0 + (s_pf[s] >> 10) * 4;
            const bool in = (s + 1) * IG_THREADS <= G::F4 || tid + s * IG_THREADS < G::F4;
            v[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in && yy >= 0 && yy < H && xx >= 0 && xx < W) v[s] = ig_ldg4(xb + s_g[s]);      // W % 4 == 0: a quad is in or out
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int s = 0; s < G::SLOTS; ++s)
            if ((s + 1) * IG_THREADS <= G::F4 || tid + s * IG_THREADS < G::F4) *reinterpret_cast<float4*>(patch + (tid + s * IG_THREADS) * 4) = v[s];
    };

    int tl = slot;
    if (tl < ntiles) load_window(tl);
    // A operand (weights): row n = 16 nset + j, k = 4m + g = (c*7 + ky)*7 + kx, read from the [NOUT][176] pack (k' = (c*7 + ky)*8 + kx); zero for
    // k >= 147.  B operand address of MFMA m: window float (c*PH + ky) * PW + kx + 1 (+ S * (this wave's first row), + S * pixel j)
    float wr[G::NM];
    int ba[G::NM];
#pragma unroll
    for (int m = 0; m < G::NM; ++m) {
        const int k = 4 * m + g;
        const int row = k / KS, kx = k - row * KS;         // row = c*KS + ky
        const int c = row / KS, ky = row - c * KS;
        const bool kok = k < G::KTOT;
        wr[m] = kok ? w[(nset * 16 + j) * kpack + row * G::KSP + kx] : 0.f;
        ba[m] = kok ? ((c * G::PH + ky + S * wrow) * G::PW + kx + G::CO + S * j) : S * j;
    }
    const s7_v4 sc = *reinterpret_cast<const s7_v4*>(scale + nset * 16 + g * 4), sh = *reinterpret_cast<const s7_v4*>(shift + nset * 16 + g * 4);
    const long long orow = (long long)Wo * outLd;
    if (tl < ntiles) store_window();
    __syncthreads();

    for (; tl < ntiles; tl += gridDim.x) {
        const int nxt = tl + gridDim.x;
        if (nxt < ntiles) load_window(nxt);
        const int tx = tl % tilesX, r_ = tl / tilesX;
        const int ty = r_ % tilesY, b = r_ / tilesY;
        const int oyw = ty * G::TH + wrow, oxl = tx * G::TW + j;
        float* const obase = out + ((long long)(b * Ho + oyw) * Wo + oxl) * outLd + nset * 16 + g * 4;
#pragma unroll
        for (int r = 0; r < G::RW; ++r) {                  // one output row = four 16-pixel tiles multiplied together
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float fr[2][4];
#define S7C_READ(buf, m) _Pragma("unroll") for (int i = 0; i < 4; ++i) fr[buf][i] = patch[ba[m] + S * r * G::PW + S * i * 16];
            S7C_READ(0, 0)
#pragma unroll
            for (int m = 0; m < G::NM; ++m) {
                const int cb = m & 1;
                if (m + 1 < G::NM) { S7C_READ(cb ^ 1, m + 1) }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m], fr[cb][i], acc[i], 0, 0, 0);
            }
#undef S7C_READ
            if (oyw + r < Ho) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (oxl + i * 16 >= Wo) continue;
                    s7_v4 o = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    o = __builtin_elementwise_fma(o, sc, sh);
                    if (relu) o = (s7_v4){fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
                    *reinterpret_cast<s7_v4*>(obase + r * orow + i * 16 * outLd) = o;
                }
            }
        }
        if (nxt < ntiles) {            // block-uniform
            __syncthreads();
            store_window();
            __syncthreads();
        }
    }
}

template <int NOUT, int S, int KS>
static int launch_stem_c16(const float* x, const float* w, int kpack, const float* scale, const float* shift, float* out, int B, int H, int W,
                           int outLd, int relu, hipStream_t s)
{
    typedef S7C<NOUT, S, KS> G;
    const int Ho = (H + 2 * G::PAD - KS) / S + 1, Wo = (W + 2 * G::PAD - KS) / S + 1;
    const int tilesX = cp_cdiv(Wo, G::TW), tilesY = cp_cdiv(Ho, G::TH);
    const long long ntiles = (long long)B * tilesX * tilesY;
    if (ntiles >= (1ll << 31)) { cp_set_error("stem: %lld tiles", ntiles); return 1; }
    const long long cap = (long long)cp_num_cus() * G::OCC;
    hipLaunchKernelGGL((stem7x7_c16_kernel<NOUT, S, KS>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(IG_THREADS), 0, s, x, w, scale, shift, out,
                       B, H, W, Ho, Wo, outLd, relu, tilesX, tilesY, (int)ntiles, kpack);
    cp_note_kernel("stem7x7_c16_kernel<%d, %d, %d>", NOUT, S, KS);
    return 0;
}

// 3x3 / stride 2 / pad 1 stem on the NCHW network input with 64 outputs (HRNet conv1, pose_higher_hrnet.py:283-285): the same persistent kernel
// with 27 -> 28 = 7 MFMAs per 16 pixels; a.w is the generic stem pack [ldw][K] with k = (c*3 + ky)*3 + kx.  -1 = not this kernel's shape.
int cp_launch_stem3x3(const ConvArgs& a, hipStream_t s)
{
    const bool ok = a.nsrc == 1 && a.srcC[0] == 3 && a.kh == 3 && a.kw == 3 && a.sy == 2 && a.sx == 2 && a.py == 1 && a.px == 1 && a.Cout == 64 &&
                    a.ldw >= 64 && !a.outNCHW && !a.res && (a.act == CP_ACT_NONE || a.act == CP_ACT_RELU) && a.ksplit == 1 && a.nsub == 1 &&
                    a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.OH == a.Ho && a.OW == a.Wo &&
                    a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1 && a.W % 4 == 0 && a.outLd % 4 == 0 &&
                    (long long)a.B * 3 * a.H * a.W < (1ll << 31) && (((size_t)a.src[0] | (size_t)a.out | (size_t)a.scale | (size_t)a.shift) & 15) == 0;
    if (!ok) return -1;
    return launch_stem_c16<64, 2, 3>(a.src[0], a.w, a.K, a.scale, a.shift, a.out, a.B, a.H, a.W, a.outLd, a.act == CP_ACT_RELU ? 1 : 0, s);
}

template <int NOUT, int S, int TH, int TW>
static int launch_stem(const float* x, const float* w, const float* scale, const float* shift, float* out, int B, int H, int W,
                       int outLd, int relu, hipStream_t s)
{
    constexpr int PH = (TH - 1) * S + 7, PWr = (TW - 1) * S + 7, PW = (PWr + 3) & ~3;
    const int Ho = (H + 6 - 7) / S + 1, Wo = (W + 6 - 7) / S + 1;
    const int smem = (((3 * PH * PW + 3) & ~3) + S7_STEPS * NOUT * IG_LDK) * 4;
    auto kern = stem7x7_kernel<NOUT, S, TH, TW>;
    static CpLdsGuard guard;
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem);
        if (e != hipSuccess) { cp_set_error("stem7x7: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    const int tilesX = cp_cdiv(Wo, TW), tilesY = cp_cdiv(Ho, TH);
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * tilesX * tilesY)), dim3(IG_THREADS), smem, s, x, w, scale, shift, out, B, H, W, Ho,
                       Wo, outLd, relu, tilesX, tilesY);
    cp_note_kernel("stem7x7_kernel<%d, %d, %d, %d>", NOUT, S, TH, TW);
    return 0;
}

// x: NCHW [B,3,H,W]; w: packed [NOUT][176] with k = ((c*7+ky)*8 + kx) (see ops.pack_stem7_weight);
// out: NHWC [B,Ho,Wo,outLd].  Cout in {16, 64}, stride in {1, 2}, pad 3.
extern "C" int cp_stem7x7_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int B, int H,
                              int W, int Cout, int stride, int outLd, int relu, void* stream)
{
    CP_CHECK_ARG(x && w && scale && shift && out, "stem7x7: null pointer");
    CP_CHECK_ARG(outLd >= Cout, "stem7x7: outLd < Cout");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // the persistent kernel needs whole float4 quads inside / outside the image and aligned rows / stores
    const bool pers = W % 4 == 0 && outLd % 4 == 0 && (long long)B * 3 * H * W < (1ll << 31) &&
                      (((size_t)x | (size_t)out | (size_t)scale | (size_t)shift) & 15) == 0;
    if (pers && Cout == 16 && stride == 1) rc = launch_stem_c16<16, 1, 7>(x, w, S7_K, scale, shift, out, B, H, W, outLd, relu, s);
    else if (pers && Cout == 64 && stride == 2) rc = launch_stem_c16<64, 2, 7>(x, w, S7_K, scale, shift, out, B, H, W, outLd, relu, s);
    else if (Cout == 16 && stride == 1) rc = launch_stem<16, 1, 8, 64>(x, w, scale, shift, out, B, H, W, outLd, relu, s);
    else if (Cout == 64 && stride == 2) rc = launch_stem<64, 2, 8, 32>(x, w, scale, shift, out, B, H, W, outLd, relu, s);
    else if (Cout == 64 && stride == 1) rc = launch_stem<64, 1, 8, 32>(x, w, scale, shift, out, B, H, W, outLd, relu, s);
    else if (Cout == 16 && stride == 2) rc = launch_stem<16, 2, 8, 64>(x, w, scale, shift, out, B, H, W, outLd, relu, s);
    else { cp_set_error("stem7x7: unsupported Cout=%d stride=%d", Cout, stride); return 1; }
    if (rc) return rc;
    CP_CHECK_LAUNCH("stem7x7_kernel");
    return 0;
}
