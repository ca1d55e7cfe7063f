// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch (gfx950, fp32 MFMA).
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every input pixel once per tap:
// 9x the global loads, 9x the LDS writes and one barrier per 16-deep k-step.  3x3/s1 layers are
// ~75 % of the network's FLOPs (DLA base blocks, DCN offset convs, the six head convs), so this
// kernel stages, per 16-channel chunk, the (8+2) x (16+2) pixel halo patch of an 8x16 output
// tile ONCE and lets all nine taps read their MFMA A-fragments straight out of it:
//   A fragment of output pixel (y,x), tap (ky,kx)  =  patch[(y+ky)][(x+kx)][16 ch]   (one ds_read_b128)
// Patch rows are padded to 384 floats so that the two pixel rows a 32-row MFMA tile spans fall on
// the same bank pattern (384 = 0 mod 64) and the 20-float pixel stride keeps b128 reads
// conflict-free.  The chunk's weights for all nine taps sit next to it (Bs[9][BN][20]).
// Per chunk and wave (BN=64): 54 ds_read_b128, 144 MFMAs, two barriers -- versus 9 barriers and
// 9x the staging traffic before.  Fragments are double-buffered in registers (tap t+1 is read
// while tap t multiplies); global loads for chunk c+1 are in flight during chunk c.
// MF = 32 uses v_mfma_f32_32x32x2 (BN = 64 / 32); MF = 16 uses 16x16x4 for 16-channel layers.
#include "igemm.h"

#define P3_TH 8
#define P3_TW 16
#define P3_PH (P3_TH + 2)
#define P3_PW (P3_TW + 2)
#define P3_PITCH 384                       // floats per patch row (18*20 = 360, padded: = 0 mod 64)
#define P3_PATCH (P3_PH * P3_PITCH)        // floats per patch buffer
#define P3_PIX4 (P3_PH * P3_PW * 4)        // float4 per patch chunk (720)
#define P3_ASLOTS ((P3_PIX4 + IG_THREADS - 1) / IG_THREADS)

template <int BN, int WAVES_M, int WAVES_N, int MF>
__global__ __launch_bounds__(IG_THREADS, 2) void conv3x3_patch_kernel(const ConvArgs a, int tilesX, int tilesY)
{
    constexpr int G = 64 / MF;                        // k-groups across the wave (2 or 4)
    constexpr int NH = IG_BK / (4 * G);               // b128 reads per fragment row per chunk (2 or 1)
    constexpr int RPT = MF / 16;                      // output rows per MFMA tile (2 or 1)
    constexpr int WROWS = P3_TH / WAVES_M;            // output rows per wave
    constexpr int TM = WROWS / RPT;
    constexpr int WN = BN / WAVES_N, TN = WN / MF;
    constexpr int NACC = IgAcc<MF>::N;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch0 = smem;                             // [P3_PATCH]
    float* Bs = smem + P3_PATCH;                      // [9][BN][IG_LDK]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NT = a.ldw / BN;
    const int tile = ig_xcd_remap(blockIdx.x, gridDim.x);
    // tile order: (n-tile group of <= 8) outermost, then spatial tile, then n-tile inside the group.
    // An XCD works on a contiguous tile range, so its 4 MiB L2 has to hold the weights of only 8
    // n-tiles (head: 1.2 MB instead of all 3.5 MB) while the 8 blocks sharing a patch run back to back.
    int nt, sp_;
    if (NT <= 8) { nt = tile % NT; sp_ = tile / NT; }
    else {
        const int SP = gridDim.x / NT;                 // spatial tiles
        const int full = (NT / 8) * 8 * SP;            // tiles covered by complete groups
        if (tile < full) { const int grp = tile / (8 * SP), rem = tile - grp * 8 * SP; sp_ = rem >> 3; nt = grp * 8 + (rem & 7); }
        else { const int gs = NT - (NT / 8) * 8, rem = tile - full; sp_ = rem / gs; nt = (NT / 8) * 8 + rem % gs; }
    }
    const int tx = sp_ % tilesX; sp_ /= tilesX;
    const int ty = sp_ % tilesY;
    const int b = sp_ / tilesY;
    const int y0 = ty * P3_TH, x0 = tx * P3_TW, n0 = nt * BN;
    const int wy0 = (wid / WAVES_N) * WROWS, wn0 = (wid % WAVES_N) * WN;
    const int C = a.srcC[0], ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];
    const int g = lane / MF, il = lane % MF;

    typename IgAcc<MF>::type acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.f;

    // ---- staging assignments (fixed per thread; element offsets fit 32 bits: checked on the host)
    int a_goff0, a_goff1, a_goff2, a_lds0, a_lds1, a_lds2;
    {
        int go[P3_ASLOTS], lo[P3_ASLOTS];
#pragma unroll
        for (int s = 0; s < P3_ASLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            const int pp = idx >> 2, q = idx & 3;
            const int pr = pp / P3_PW, pc = pp - pr * P3_PW;
            const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
            const bool ok = idx < P3_PIX4 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            go[s] = ok ? ((b * a.H + yy) * a.W + xx) * ld + q * 4 : -1;
            lo[s] = idx < P3_PIX4 ? pr * P3_PITCH + pc * IG_LDK + q * 4 : -1;
        }
        a_goff0 = go[0]; a_goff1 = go[1]; a_goff2 = go[2];
        a_lds0 = lo[0]; a_lds1 = lo[1]; a_lds2 = lo[2];
    }
    static_assert(P3_ASLOTS == 3, "three patch slots per thread");
    // weight staging: thread -> fixed (n, q); slot s covers taps TPS*s + b_t0
    constexpr int TPS = IG_THREADS / (BN * 4);        // taps covered by one slot: 1 (BN=64), 2 (32), 4 (16)
    constexpr int NBS = (9 + TPS - 1) / TPS;          // slots in use: 9, 5, 3
    const int b_n = (tid >> 2) % BN, b_q = tid & 3, b_t0 = tid / (BN * 4);
    const float* wrow = a.w + (size_t)(n0 + b_n) * a.K + b_q * 4;
    float* bdst = Bs + (b_t0 * BN + b_n) * IG_LDK + b_q * 4;
    float4 ar0, ar1, ar2, b0, b1, b2, b3, b4, b5, b6, b7, b8;
#define P3_LB(s, c0) ig_ldg4(wrow + ((TPS * (s) + b_t0 < 9) ? (TPS * (s) + b_t0) : 8) * C + (c0))   /* unconditional, tap clamped; P3_SB guards the store */
#define P3_SB(s, v) if (TPS * (s) + b_t0 < 9) *reinterpret_cast<float4*>(bdst + TPS * (s) * BN * IG_LDK) = (v)
#define P3_LOAD(c0)                                                                                          \
    {                                                                                                        \
        /* raw loads only: zeroing of out-of-image pixels happens at store time -- a select here would make   \
           the wave wait for the prefetch (s_waitcnt vmcnt) before the chunk's MFMAs even start */           \
        ar0 = ig_ldg4(x + (a_goff0 >= 0 ? a_goff0 + (c0) : 0));                                              \
        ar1 = ig_ldg4(x + (a_goff1 >= 0 ? a_goff1 + (c0) : 0));                                              \
        ar2 = ig_ldg4(x + (a_goff2 >= 0 ? a_goff2 + (c0) : 0));                                              \
        b0 = P3_LB(0, c0); b1 = P3_LB(1, c0); b2 = P3_LB(2, c0);                                             \
        if (NBS > 3) { b3 = P3_LB(3, c0); b4 = P3_LB(4, c0); }                                               \
        if (NBS > 5) { b5 = P3_LB(5, c0); b6 = P3_LB(6, c0); b7 = P3_LB(7, c0); b8 = P3_LB(8, c0); }         \
    }
#define P3_STORE(patch)                                                                                      \
    {                                                                                                        \
        if (a_lds0 >= 0) *reinterpret_cast<float4*>((patch) + a_lds0) = a_goff0 >= 0 ? ar0 : make_float4(0.f, 0.f, 0.f, 0.f);         \
        if (a_lds1 >= 0) *reinterpret_cast<float4*>((patch) + a_lds1) = a_goff1 >= 0 ? ar1 : make_float4(0.f, 0.f, 0.f, 0.f);         \
        if (a_lds2 >= 0) *reinterpret_cast<float4*>((patch) + a_lds2) = a_goff2 >= 0 ? ar2 : make_float4(0.f, 0.f, 0.f, 0.f);         \
        P3_SB(0, b0); P3_SB(1, b1); P3_SB(2, b2);                                                            \
        if (NBS > 3) { P3_SB(3, b3); P3_SB(4, b4); }                                                         \
        if (NBS > 5) { P3_SB(5, b5); P3_SB(6, b6); P3_SB(7, b7); P3_SB(8, b8); }                             \
    }

    // lane's fragment bases: A row il -> pixel (il>>4, il&15) of MFMA tile i; B row il -> channel
    const int a_lane = (wy0 + (il >> 4)) * P3_PITCH + (il & 15) * IG_LDK + g * 4;
    const int b_lane = (wn0 + il) * IG_LDK + g * 4;

    const int nchunk = C / IG_BK;
    P3_LOAD(0);
    P3_STORE(patch0);
    __syncthreads();
    const float* P = patch0 + a_lane;
    const float* Q = Bs + b_lane;
    for (int ci = 0; ci < nchunk; ++ci) {
        const bool more = ci + 1 < nchunk;
        if (more) P3_LOAD((ci + 1) * IG_BK);
        // fragments double-buffered in registers: tap t+1 is read from LDS while tap t multiplies
        float4 af[2][NH][TM], bf[2][NH][TN];
#define P3_FRAG(buf, t)                                                                                      \
        _Pragma("unroll") for (int h = 0; h < NH; ++h) {                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
                af[buf][h][i] = *reinterpret_cast<const float4*>(P + (i * RPT + (t) / 3) * P3_PITCH + ((t) % 3) * IG_LDK + h * 4 * G); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
                bf[buf][h][j] = *reinterpret_cast<const float4*>(Q + ((t) * BN + j * MF) * IG_LDK + h * 4 * G); \
        }
        P3_FRAG(0, 0)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int cb = t & 1;
            if (t + 1 < 9) { P3_FRAG(cb ^ 1, t + 1) }
            // pin the order: tap t+1's LDS reads are ISSUED before tap t's MFMAs (otherwise the scheduler sinks
            // them next to their use, re-using one register set, and every tap waits a full LDS round trip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = ig_mfma<MF>(af[cb][h][i].x, bf[cb][h][j].x, acc[i][j]);
                        acc[i][j] = ig_mfma<MF>(af[cb][h][i].y, bf[cb][h][j].y, acc[i][j]);
                        acc[i][j] = ig_mfma<MF>(af[cb][h][i].z, bf[cb][h][j].z, acc[i][j]);
                        acc[i][j] = ig_mfma<MF>(af[cb][h][i].w, bf[cb][h][j].w, acc[i][j]);
                    }
        }
        __syncthreads();                // all waves are done with Bs and the patch
        if (more) P3_STORE(patch0);
        __syncthreads();
    }

    // ---- epilogue: scale/shift (+residual) + activation ----
    const bool relu = a.act == CP_ACT_RELU, sigm = a.act == CP_ACT_SIGMOID;
    const bool vec_ok = ((a.outLd | a.Cout) & 3) == 0 && (((size_t)a.out) & 15) == 0 &&
                        (!a.res || ((a.resLd & 3) == 0 && (((size_t)a.res) & 15) == 0));
    if (vec_ok) {
        // vectorised: C tile through LDS (Cs[pixel][n], the main-loop buffers are dead after the last
        // barrier), then float4 residual loads / stores along n
        float* Cs = smem;
        constexpr int LDC = BN + 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r)
                    Cs[((wy0 + i * RPT) * P3_TW + ig_row<MF>(r, lane)) * LDC + wn0 + j * MF + il] = acc[i][j][r];
        __syncthreads();
        constexpr int NV = BN / 4;
#pragma unroll
        for (int s2 = 0; s2 < P3_TH * P3_TW * NV / IG_THREADS; ++s2) {
            const int idx = tid + s2 * IG_THREADS;
            const int pl = idx / NV, c4 = idx - pl * NV;
            const int yy = y0 + (pl >> 4), xx = x0 + (pl & 15), n = n0 + c4 * 4;
            if (yy < a.H && xx < a.W && n < a.Cout) {
                const size_t opix = ((size_t)b * a.H + yy) * a.W + xx;
                float4 v = *reinterpret_cast<const float4*>(Cs + pl * LDC + c4 * 4);
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
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int rowl = ig_row<MF>(r, lane);     // row of the MFMA tile: pixel (rowl>>4, rowl&15)
            const int yy = y0 + wy0 + i * RPT + (rowl >> 4), xx = x0 + (rowl & 15);
            if (yy >= a.H || xx >= a.W) continue;
            const size_t opix = ((size_t)b * a.H + yy) * a.W + xx;
            float* orow = a.out + opix * a.outLd;
            const float* rrow = a.res ? a.res + opix * a.resLd : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * MF + il;
                if (n < a.Cout) {
                    float v = acc[i][j][r] * a.scale[n] + a.shift[n];
                    if (rrow) v += rrow[n];
                    v = cp_act(v, a.act);
                    orow[n] = v;
                }
            }
        }
    }
}

template <int BN, int WAVES_M, int WAVES_N, int MF>
static int launch_patch(const ConvArgs& a, hipStream_t s)
{
    auto kern = conv3x3_patch_kernel<BN, WAVES_M, WAVES_N, MF>;
    const int smem = (P3_PATCH + 9 * BN * IG_LDK) * 4;
    static CpLdsGuard guard;
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem);
        if (e != hipSuccess) { cp_set_error("conv3x3_patch: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    const int tilesX = cp_cdiv(a.W, P3_TW), tilesY = cp_cdiv(a.H, P3_TH);
    const long long grid = (long long)a.B * tilesX * tilesY * (a.ldw / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, tilesX, tilesY);
    cp_note_kernel("conv3x3_patch_kernel<%d, %d, %d, %d>", BN, WAVES_M, WAVES_N, MF);
    return 0;
}

// eligibility + dispatch; returns -1 if the shape is not handled here (caller falls back to the generic kernel)
int cp_launch_conv3x3_patch(const ConvArgs& a, int in_nchw, hipStream_t s, int variant)
{
    const bool ok = !in_nchw && a.nsrc == 1 && a.kh == 3 && a.kw == 3 && a.sy == 1 && a.sx == 1 && a.py == 1 && a.px == 1 &&
                    !a.outNCHW && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.Ho == a.H && a.Wo == a.W &&
                    a.OH == a.H && a.OW == a.W && a.srcC[0] % 16 == 0 && (a.ldw % 64 == 0 || a.ldw == 32 || a.ldw == 16) &&
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31);
    if (!ok) return -1;
    if (a.ldw == 16) return launch_patch<16, 4, 1, 16>(a, s);
    // measured (MI355X): with <= 4 channel chunks per tile the per-tile prologue/epilogue dominates and the
    // 32-channel variant (38 KB LDS -> 4 blocks/CU instead of 2) wins: head 3x3 106 -> 117 TFLOP/s;
    // from 128 input channels on, the 64-channel variant's better LDS-read/MFMA ratio wins.
    if (a.ldw == 32 || variant == 32 || (variant == 0 && a.srcC[0] <= 64)) return launch_patch<32, 4, 1, 32>(a, s);
    return launch_patch<64, 2, 2, 32>(a, s);
}
