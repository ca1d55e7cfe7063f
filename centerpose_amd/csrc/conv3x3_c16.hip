// 3x3 / pad 1 convolution of a 16-CHANNEL NHWC map: stride 1 -> 16 outputs, stride 2 -> 32 outputs (gfx950, fp32 MFMA 16x16x4).
//
//   DLA-34  level0: Conv2d(16,16,k3,s1,p1)+BN+ReLU @512x512,  level1: Conv2d(16,32,k3,s2,p1)+BN+ReLU -> 256x256
//   (pose_dla_dcn.py:234-246,272-282: `_make_conv_level`)
// These two layers ran on kernels built for wide layers (conv3x3_patch_kernel<16,...>: 72 MFMAs per wave and block under a
// prologue / LDS epilogue nothing overlaps, 3.1 other VALU per MFMA; igemm_conv_kernel<128,32>: 9 VALU per MFMA of im2col
// addressing) at 0.57 / 0.46 of the fp32 MFMA peak.  With K = 144 and N = 16 / 32 everything that is per-block constant fits in
// registers, so this kernel is organised the other way round:
//   * the WEIGHTS are the MFMA A operand and stay in registers for the whole block (36 VGPRs per 16 outputs): no weight
//     staging, no weight ds_reads;
//   * the pixels are the B operand: one ds_read_b128 of the LDS halo patch (a lane's pixel, channels 4g .. 4g+3) feeds four MFMAs
//     (the k order inside a tap is permuted accordingly: MFMA m takes channel 4g + m from k-group g);
//   * D = W x P leaves a lane with FOUR CONSECUTIVE CHANNELS of one pixel: folded BN + ReLU and one float4 NHWC store per
//     16-pixel tile, no LDS epilogue, no barrier after the prologue's;
//   * every LDS address is lane base + immediate; block tiles of 8x32 (stride 1) / 8x16 (stride 2) output pixels;
//   * PERSISTENT: three or four blocks per CU walk the block tiles and request the next tile's patch before the current tile's MFMAs
//     (see the loop below: without it the kernel measured exactly what the kernels it replaces did).
//   stride 2: the patch is stored de-interleaved by column parity, so that the 16 pixels of a tile (input columns 2x + kx) are
//   consecutive LDS rows again (stride-2 reads of a 20-float pitch would be two-way bank conflicts).
#include "igemm.h"

typedef float c16_v4 __attribute__((ext_vector_type(4)));

template <int S, int TH, int TW>
struct C16Geo {
    static constexpr int PH = (TH - 1) * S + 3;               // patch rows
    static constexpr int PC = (TW - 1) * S + 3;               // patch columns (input pixels)
    static constexpr int PWP = S == 1 ? TW + 2 : TW + 1;      // pixels per LDS row (stride 2: per parity plane)
    static constexpr int ROW = S * PWP * IG_LDK;              // floats per patch row (stride 2: even plane, odd plane)
    static constexpr int PATCH = PH * ROW;
    // staging: the first 32 pixels of a patch row are 128 float4 -- two rows per pass of the 256 threads, rows 2s + (tid >> 7), with
    // addresses that are one per-thread offset + a scalar per pass; the PC - 32 halo pixels of all rows make one more pass
    static constexpr int MSLOTS = (PH + 1) / 2;
    static constexpr int HF4 = (PC - 32) * 4;                 // float4 per row in the halo pass: 8 (stride 1) / 4 (stride 2)
    static_assert(TW * S == 32 && PH * HF4 <= IG_THREADS, "staging layout");
};

template <int NT, int S, int TH, int TW, int OCC, int TBW>
__global__ __launch_bounds__(IG_THREADS, OCC) void conv3x3_c16_kernel(const ConvArgs a, int tilesX, int tilesY, int ntiles)
{
    typedef C16Geo<S, TH, TW> G;
    constexpr int RW = TH / 4, CT = TW / 16;                  // output rows per wave, 16-pixel tiles per row
    constexpr int TILES = RW * CT;
    constexpr int TB = TILES >= TBW ? TBW : TILES;            // tiles multiplied together (independent accumulators)
    static_assert(TH % 4 == 0 && TW % 16 == 0 && TILES % TB == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];
    // PERSISTENT: the grid is OCC blocks per CU; round `it` works on the block tiles [it * gridDim, (it + 1) * gridDim), inside which the
    // XCD remap hands every XCD (block ids are dealt round-robin over the eight) one contiguous range -- neighbouring halos meet in one L2
    const int slot = ig_xcd_remap(blockIdx.x, gridDim.x);

    // patch staging of one block tile: global -> registers (`v`), later registers -> LDS.  Out-of-image pixels are zero.
    // Everything per-thread is computed ONCE (three offsets); per tile and pass only scalars change -- the prefetch registers have to
    // coexist with the weights, and a spill reload inside the loop would wait for the whole prefetch (scratch shares vmcnt).
    float4 v[G::MSLOTS + 1];
    const int mf = tid & 127, mrow = __builtin_amdgcn_readfirstlane(tid >> 7);       // main pass: float4 mf of row 2s + mrow
    const int mpc = mf >> 2;
    const unsigned m_g = (unsigned)((mrow * a.W + mpc) * ld + (mf & 3) * 4);
    const int m_l = mrow * G::ROW + (S == 1 ? mpc : ((mpc & 1) * G::PWP + (mpc >> 1))) * IG_LDK + (mf & 3) * 4;
    const int hrow = tid / G::HF4, hf = 128 + tid % G::HF4, hpc = hf >> 2;           // halo pass: pixel column 32 (+1), row hrow
    const unsigned h_g = (unsigned)((hrow * a.W + hpc) * ld + (hf & 3) * 4);
    const int h_l = hrow * G::ROW + (S == 1 ? hpc : ((hpc & 1) * G::PWP + (hpc >> 1))) * IG_LDK + (hf & 3) * 4;
    auto load_patch = [&](int tl) {
        const int tx = tl % tilesX, r_ = tl / tilesX;
        const int ty = r_ % tilesY, b = r_ / tilesY;
        const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
        const float* xb = x + ((long long)(b * a.H + iy0) * a.W + ix0) * ld;      // scalar; may point before the image (never loaded from)
        const bool mcol = ix0 + mpc >= 0 && ix0 + mpc < a.W;
#pragma unroll
        for (int s = 0; s < G::MSLOTS; ++s) {
            const int yy = iy0 + 2 * s + mrow;                                       // scalar
            v[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (2 * s + mrow < G::PH && yy >= 0 && yy < a.H && mcol) v[s] = ig_ldg4(xb + (long long)(2 * s) * a.W * ld + m_g);
        }
        v[G::MSLOTS] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hrow < G::PH && iy0 + hrow >= 0 && iy0 + hrow < a.H && ix0 + hpc < a.W) v[G::MSLOTS] = ig_ldg4(xb + h_g);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int s = 0; s < G::MSLOTS; ++s)
            if (2 * s + mrow < G::PH) *reinterpret_cast<float4*>(smem + 2 * s * G::ROW + m_l) = v[s];
        if (hrow < G::PH) *reinterpret_cast<float4*>(smem + h_l) = v[G::MSLOTS];
    };

    int tl = slot;
    if (tl < ntiles) load_patch(tl);
    // this lane's weights, MFMA A operand: row n = j (+16 nt), k-group g; tap t, MFMA m <-> channel 4g + m.  Loaded once per block.
    float4 wr[NT][9];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[nt][t] = ig_ldg4(a.w + (size_t)(nt * 16 + j) * a.K + t * 16 + g * 4);
    c16_v4 sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        sc[nt] = *reinterpret_cast<const c16_v4*>(a.scale + nt * 16 + g * 4);
        sh[nt] = *reinterpret_cast<const c16_v4*>(a.shift + nt * 16 + g * 4);
    }
    const float* P = smem + (wid * RW * S) * G::ROW + j * IG_LDK + g * 4;      // lane base inside the patch: pixel j, channels 4g .. 4g + 3
    const bool relu = a.act == CP_ACT_RELU;
    const float* const a_res = a.res;
    const int orow = a.Wo * a.outLd, rrow = a.Wo * a.resLd;
    if (tl < ntiles) store_patch();
    __syncthreads();

    for (; tl < ntiles; tl += gridDim.x) {
        // the NEXT tile's patch is requested before this tile's MFMAs and parked in registers: it lands while the matrix pipe works
        // (with three resident blocks per CU alone, a block in its MFMA phase has nothing in flight and memory-bound phases of
        // ~10 us per block were only partly covered: 0.203 ms for level0 at B = 16 whatever the tile shape, against 0.156 ms of MFMA
        // phases and 0.105 ms of memory phases measured separately)
        const int nxt = tl + gridDim.x;
        if (nxt < ntiles) load_patch(nxt);
        const int tx = tl % tilesX, r_ = tl / tilesX;
        const int ty = r_ % tilesY, b = r_ / tilesY;
        const int oyw = ty * TH + wid * RW, oxl = tx * TW + j;
        float* const obase = a.out + ((size_t)(b * a.Ho + oyw) * a.Wo + oxl) * a.outLd + g * 4;
        const float* const rbase = a_res ? a_res + ((size_t)(b * a.Ho + oyw) * a.Wo + oxl) * a.resLd + g * 4 : nullptr;
#pragma unroll
        for (int bt = 0; bt < TILES / TB; ++bt) {
            f32x4 acc[TB][NT];
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float4 fr[2][TB];
            // tile i of the batch: output row r, column tile ct of this wave's strip; tap (ky, kx) of it is one float4 of the patch
#define C16_OFF(i, t)                                                                                                            \
            ((((bt * TB + (i)) / CT) * S + (t) / 3) * G::ROW +                                                                    \
             (S == 1 ? (((bt * TB + (i)) % CT) * 16 + (t) % 3) : ((((t) % 3) & 1) * G::PWP + ((bt * TB + (i)) % CT) * 16 + (((t) % 3) >> 1))) * IG_LDK)
#define C16_READ(buf, t)                                                                                                         \
            _Pragma("unroll") for (int i = 0; i < TB; ++i) fr[buf][i] = *reinterpret_cast<const float4*>(P + C16_OFF(i, t));
            C16_READ(0, 0)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cb = t & 1;
                if (t + 1 < 9) { C16_READ(cb ^ 1, t + 1) }
                __builtin_amdgcn_sched_barrier(0);         // tap t+1's reads are issued before tap t's MFMAs
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][t].x, fr[cb][i].x, acc[i][nt], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][t].y, fr[cb][i].y, acc[i][nt], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][t].z, fr[cb][i].z, acc[i][nt], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TB; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][t].w, fr[cb][i].w, acc[i][nt], 0, 0, 0);
            }
#undef C16_READ
#undef C16_OFF
            // ---- epilogue: lane (j, g) holds channels 4g .. 4g+3 (+16 nt) of pixel j: folded BN (+ residual) + ReLU, one float4 store
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const int r = (bt * TB + i) / CT, ct = (bt * TB + i) % CT;
                if (oyw + r >= a.Ho || oxl + ct * 16 >= a.Wo) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    c16_v4 o = {acc[i][nt][0], acc[i][nt][1], acc[i][nt][2], acc[i][nt][3]};
                    o = __builtin_elementwise_fma(o, sc[nt], sh[nt]);
                    if (a_res) o += *reinterpret_cast<const c16_v4*>(rbase + (size_t)r * rrow + ct * 16 * a.resLd + nt * 16);
                    if (relu) o = (c16_v4){cp_relu(o.x), cp_relu(o.y), cp_relu(o.z), cp_relu(o.w)};
                    *reinterpret_cast<c16_v4*>(obase + (size_t)r * orow + ct * 16 * a.outLd + nt * 16) = o;
                }
            }
        }
        if (nxt < ntiles) {            // block-uniform
            __syncthreads();           // every wave is done with this tile's patch
            store_patch();
            __syncthreads();
        }
    }
}

template <int NT, int S, int TH, int TW, int OCC, int TBW>
static int launch_c16(const ConvArgs& a, hipStream_t s)
{
    typedef C16Geo<S, TH, TW> G;
    const int smem = G::PATCH * 4;
    static_assert(G::PATCH * 4 <= 64 * 1024, "patch fits the default LDS limit");
    const int tilesX = cp_cdiv(a.Wo, TW), tilesY = cp_cdiv(a.Ho, TH);
    const long long ntiles = (long long)a.B * tilesX * tilesY;
    if (ntiles >= (1ll << 31)) { cp_set_error("conv3x3_c16: %lld tiles", ntiles); return 1; }
    const int ncu = cp_num_cus();
    const long long cap = (long long)ncu * OCC;             // persistent: OCC resident blocks per CU walk the tiles
    const long long grid = ntiles < cap ? ntiles : cap;
    hipLaunchKernelGGL((conv3x3_c16_kernel<NT, S, TH, TW, OCC, TBW>), dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, tilesX, tilesY, (int)ntiles);
    cp_note_kernel("conv3x3_c16_kernel<%d, %d, %d, %d, %d, %d>", NT, S, TH, TW, OCC, TBW);      // as rocprofv3 prints the instantiation
    return 0;
}

// eligibility + dispatch; -1 = not this kernel's shape (16 NHWC input channels, 3x3 / pad 1, stride 1 or 2, exactly 16 or 32 outputs,
// no activation or ReLU, 16-byte aligned NHWC output / residual)
int cp_launch_conv3x3_c16(const ConvArgs& a, int in_nchw, hipStream_t s)
{
    const bool ok = !in_nchw && a.nsrc == 1 && a.srcC[0] == 16 && a.kh == 3 && a.kw == 3 && a.sy == a.sx && (a.sy == 1 || a.sy == 2) &&
                    a.py == 1 && a.px == 1 && a.K == 144 && !a.outNCHW && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 &&
                    a.OH == a.Ho && a.OW == a.Wo && a.Ho == (a.H - 1) / a.sy + 1 && a.Wo == (a.W - 1) / a.sx + 1 &&
                    (a.Cout == 16 || a.Cout == 32) && a.ldw >= a.Cout && a.ksplit == 1 && a.nsub == 1 &&
                    (a.act == CP_ACT_NONE || a.act == CP_ACT_RELU) && a.srcLd[0] % 4 == 0 && a.outLd % 4 == 0 &&
                    (((size_t)a.src[0] | (size_t)a.w | (size_t)a.out | (size_t)a.scale | (size_t)a.shift) & 15) == 0 &&
                    (!a.res || (a.resLd % 4 == 0 && (((size_t)a.res) & 15) == 0)) &&
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31) && (long long)a.Wo * a.outLd * 16 < (1ll << 31) &&
                    (long long)a.Wo * a.resLd * 16 < (1ll << 31);
    if (!ok) return -1;
    // measured at B = 16, 512x512 (tools/c16_ab.py, one box): stride 1 / 16 outputs: 8x32 tiles, 4 blocks per CU, two 16-pixel tiles
    // multiplied together 0.171 ms; 16x32 / 3 per CU / four together 0.179; stride 2 / 32 outputs: 8x16 tiles 0.106, 4x16 0.110
    if (a.sy == 1) return a.Cout == 16 ? launch_c16<1, 1, 8, 32, 4, 2>(a, s) : launch_c16<2, 1, 8, 32, 2, 2>(a, s);
    return a.Cout == 16 ? launch_c16<1, 2, 8, 16, 3, 2>(a, s) : launch_c16<2, 2, 8, 16, 3, 1>(a, s);
}
