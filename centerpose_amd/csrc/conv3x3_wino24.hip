// 3x3 / stride 1 / pad 1 convolution as a fused Winograd F(2x4,3x3) on the fp32 matrix cores (gfx950).
//
// conv3x3_wino.hip computes 2x2 output tiles with 16 multiplies per (cin,cout) -- 4 per output; this variant computes 2x4
// output tiles from 4x6 input tiles with 24 multiplies -- 3 per output, i.e. 0.75x the MFMAs ("measure one
// wider Winograd, don't estimate it").  Rows use the F(2,3) transform, columns the F(4,3) transform (Lavin & Gray):
//     Y = A2^T [ sum_c (G2 g G4^T) .* (B2^T d B4) ] A4
//     B4^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G4   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Same organisation as the F(2x2) kernel:
//   * block = 16 x 16 output pixels = 8 x 4 Winograd tiles (one 32-wide MFMA tile) x 32 output channels, 4 waves;
//   * wave xi owns transform ROW xi (frequencies (xi, nu = 0..5)): row xi of B2^T has two non-zeros, so a lane (tile m, k-half h)
//     reads two rows x six columns of its 4x6 input tile = 12 ds_read_b128 per 8-channel chunk, forms the row combination with 12
//     packed fmas and the six column frequencies with 24 packed ops, straight into the MFMA operand registers;
//   * the raw (16+2) x (16+2) halo patch is staged 16 channels at a time (double buffered, one barrier per stage); layout
//     [cg][py][px ^ s(py)][4] with s(py) = (py >> 1) & 3 on the low two bits of px and a row pitch of 20 float4: the 16 lanes of a
//     ds_read_b128 group (4 tile columns x 4 tile rows) hit slot 4 * ((2 tyy + txx) & 3) + (j ^ tyy) mod 16 -- all distinct;
//   * U = G2 g G4^T (cp_winograd24_pack_f32, fp64, rounded once) streams global -> registers in fragment order
//     [xi][ntile][kc][nu 6][lane][4], one 8-channel chunk ahead, issued from inside the MFMA block;
//   * per chunk and wave: 12 ds_read_b128 + 36 packed VALU + 6 global loads feed 24 v_mfma_f32_32x32x2f32;
//   * epilogue: the nu -> 4 output columns transform (A4) in registers, the xi-sum (A2) across the four waves through LDS in one
//     pass, then scale / shift + residual + activation.
// fp32 throughout.  F(4,3) has larger transform constants than F(2,3): the measured error against an fp64 convolution is in
// tools/wino_check.py / DESIGN.md 7.1.
#include <type_traits>
#include "igemm.h"

#define W24_PH 18
#define W24_PW 18
#define W24_PWP 20                                 // patch row pitch in float4 slots
#define W24_KS 16
#define W24_CGS (W24_KS / 4)
#define W24_CG (W24_PH * W24_PWP * 4 + 16)         // floats per 4-channel plane: 1440 + 16 -- with 1440 (= 32 mod 64 banks) the four channel
                                                   // quads q = 0..3 that one 8-lane ds_write_b128 group stores for a patch pixel landed on banks
                                                   // {0, 32, 0, 32}: a 2-way conflict on EVERY patch store (SQ_LDS_BANK_CONFLICT 43 % of the LDS-active
                                                   // cycles, profiles/r4_pmc_wino_kernels.txt); 1456 puts them on {0, 48, 32, 16}.  Reads stay inside one plane
#define W24_STAGE (W24_CGS * W24_CG)               // floats per stage buffer (23 KB)
#define W24_F4 (W24_PH * W24_PW * W24_CGS)         // 1296 float4 per stage
#define W24_SLOTS ((W24_F4 + IG_THREADS - 1) / IG_THREADS)
#define W24_LDR 36
#define W24_RED (16 * 32 * W24_LDR)                // [xi 4][col 4][tile 32][36] floats (73.7 KB): the epilogue's reduction buffer
#define W24_MAIN (2 * W24_STAGE > W24_RED ? 2 * W24_STAGE : W24_RED)
#define W24_SMEM_FLOATS (W24_MAIN + 16)

typedef float w24_v2 __attribute__((ext_vector_type(2)));
typedef float w24_v4 __attribute__((ext_vector_type(4)));

struct W24Grid {
    int tilesX, tilesY, ntb;
    unsigned mNtb, mTx, mTy;
};
__device__ __forceinline__ int w24_div(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }
__device__ __forceinline__ w24_v4 w24_lds4(const float* p) { return *reinterpret_cast<const w24_v4*>(p); }

// One accumulator group (32 tiles x 32 channels, the wave's six frequencies) -> output pixels, all four output columns in ONE pass:
// A4 in registers (the s / d sums once), A2 across the waves through a [xi 4][col 4][tile 32][36] buffer (73.7 KB: still two blocks
// per CU), four items per thread -- every item of a thread has the same four channels, so scale / shift are loaded once -- then
// scale / shift (+ residual) + activation and float4 NHWC stores.  (Round 4 first used two passes of two columns through the F(2x2)
// kernel's 36.9 KB buffer: three barriers and the s / d sums twice; one pass is 2-3 % faster on the 64- / 128-channel layers, same bits.)
__device__ __forceinline__ void w24_output_generic(const ConvArgs& a, float* red, int tid, int b, int y0, int x0, int tile)
{
    const float* const a_res = a.res;
    const bool relu = a.act == CP_ACT_RELU;
    const bool vec_ok = (a.outLd & 3) == 0 && (((size_t)a.out) & 15) == 0 &&
                        (!a_res || ((a.resLd & 3) == 0 && (((size_t)a_res) & 15) == 0));
    const int n4 = tid & 7, n = tile * 32 + n4 * 4;
    const bool nvec = vec_ok && n + 3 < a.Cout;
    w24_v4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nvec) {
        sc = *reinterpret_cast<const w24_v4*>(a.scale + n);
        sh = *reinterpret_cast<const w24_v4*>(a.shift + n);
    }
    int itox[4], itoy[4], itrd[4];
    bool itok[4];
    size_t itpix[4];
    w24_v4 rr[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = tid + it * IG_THREADS;
        const int col = (item >> 3) & 3, mi = item >> 5;
        itox[it] = x0 + 4 * (mi & 3) + col;
        itoy[it] = y0 + 2 * (mi >> 2);
        itrd[it] = (col * 32 + mi) * W24_LDR + n4 * 4;
        itok[it] = itox[it] < a.W && n < a.Cout && itoy[it] < a.H;
        itpix[it] = ((size_t)b * a.H + itoy[it]) * a.W + itox[it];
        if (itok[it] && nvec && a_res) {
            rr[it][0] = *reinterpret_cast<const w24_v4*>(a_res + itpix[it] * a.resLd + n);
            if (itoy[it] + 1 < a.H) rr[it][1] = *reinterpret_cast<const w24_v4*>(a_res + (itpix[it] + a.W) * a.resLd + n);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (!itok[it]) continue;
        const float* rp = red + itrd[it];
        const w24_v4 q0 = w24_lds4(rp), q1 = w24_lds4(rp + 128 * W24_LDR), q2 = w24_lds4(rp + 256 * W24_LDR), q3 = w24_lds4(rp + 384 * W24_LDR);
        w24_v4 yv[2];
        yv[0] = (q0 + q1) + q2;
        yv[1] = (q1 - q2) - q3;
        if (nvec) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                if (itoy[it] + aa >= a.H) continue;
                const size_t opix = itpix[it] + (size_t)aa * a.W;
                w24_v4 v = __builtin_elementwise_fma(yv[aa], sc, sh);
                if (a_res) v += rr[it][aa];
                if (relu) v = (w24_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};
                else if (a.act != CP_ACT_NONE) { v.x = cp_act(v.x, a.act); v.y = cp_act(v.y, a.act); v.z = cp_act(v.z, a.act); v.w = cp_act(v.w, a.act); }
                *reinterpret_cast<w24_v4*>(a.out + opix * a.outLd + n) = v;
            }
        } else {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                if (itoy[it] + aa >= a.H) continue;
                const size_t opix = itpix[it] + (size_t)aa * a.W;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (n + k >= a.Cout) break;
                    float w_ = yv[aa][k] * a.scale[n + k] + a.shift[n + k];
                    if (a_res) w_ += a_res[opix * a.resLd + n + k];
                    w_ = cp_act(w_, a.act);
                    a.out[opix * a.outLd + n + k] = w_;
                }
            }
        }
    }
}

// The epilogue proper.  The common case -- 16-byte aligned NHWC output / residual, channel count a multiple of four, no activation or
// ReLU -- is written for instruction count: an ablation (tools/w24_ablation.py, round 4) put the epilogue at 18 % / 11 % / 7 % of the
// kernel for 64 / 128 / 256 input channels, ~700 instructions per wave with a quarter of them 64-bit or quarter-rate integer address
// arithmetic (a v_mul_lo_u32 / v_mad_u64_u32 group per item and row) and thousands of lines of rarely taken paths (scalar tail, sigmoid)
// around them.  The four items of a thread differ only by four output rows: one 64-bit base per thread, scalar strides, immediate LDS
// offsets.  Everything else takes the generic function above (same arithmetic, same bits) in the kernels' FAST = false instantiation,
// chosen on the host (w24_fast_epilogue): the generic code stays out of the common kernel instead of behind a call and its stack frame.
template <bool FAST>
__device__ __forceinline__ void w24_output_all(const ConvArgs& a, float* red, const f32x16 (&acc)[6], int xi, int h, int m, int tid,
                                               int b, int y0, int x0, int tile)
{
    float* wp = red + (xi * 128 + m) * W24_LDR + 4 * h;           // [xi][col][tile][36]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w24_v4 q[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) q[nu] = (w24_v4){acc[nu][4 * j], acc[nu][4 * j + 1], acc[nu][4 * j + 2], acc[nu][4 * j + 3]};
        const w24_v4 s1 = q[1] + q[2], d1 = q[1] - q[2], s2 = q[3] + q[4], d2 = q[3] - q[4];
        *reinterpret_cast<w24_v4*>(wp + 8 * j) = (q[0] + s1) + s2;                               // column 0
        *reinterpret_cast<w24_v4*>(wp + 32 * W24_LDR + 8 * j) = d1 + 2.f * d2;                   // column 1
        *reinterpret_cast<w24_v4*>(wp + 64 * W24_LDR + 8 * j) = s1 + 4.f * s2;                   // column 2
        *reinterpret_cast<w24_v4*>(wp + 96 * W24_LDR + 8 * j) = (d1 + 8.f * d2) + q[5];          // column 3
    }
    if (!FAST) {
        w24_output_generic(a, red, tid, b, y0, x0, tile);
        return;
    }
    const float* const a_res = a.res;
    // item `it` of this thread is tile (tid >> 5) + 8 it -- tile row (tid >> 7) + 2 it, tile column (tid >> 5) & 3 -- output column
    // (tid >> 3) & 3, channels 4 (tid & 7) .. + 3: only the tile row changes with `it`, four output rows further down per item
    const int n4 = tid & 7, col = (tid >> 3) & 3, n = tile * 32 + n4 * 4;
    const int ox = x0 + 4 * ((tid >> 5) & 3) + col, oyb = y0 + 2 * (tid >> 7);
    const bool okx = ox < a.W && n < a.Cout;
    const size_t p0 = ((size_t)b * a.H + oyb) * a.W + ox;
    float* const op = a.out + p0 * a.outLd + n;
    const float* const rp0 = a_res ? a_res + p0 * a.resLd + n : nullptr;
    const int ostep = a.W * a.outLd, rstep = a.W * a.resLd;        // one output row, in floats (scalar registers)
    w24_v4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (okx) {
        sc = *reinterpret_cast<const w24_v4*>(a.scale + n);
        sh = *reinterpret_cast<const w24_v4*>(a.shift + n);
    }
    w24_v4 rr[4][2];
    if (a_res) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
                if (okx && oyb + 4 * it + aa < a.H) rr[it][aa] = *reinterpret_cast<const w24_v4*>(rp0 + (4 * it + aa) * rstep);
    }
    const float* const rdp = red + (col * 32 + (tid >> 5)) * W24_LDR + n4 * 4;
    const bool relu = a.act == CP_ACT_RELU;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float* rp = rdp + it * 8 * W24_LDR;
        const w24_v4 q0 = w24_lds4(rp), q1 = w24_lds4(rp + 128 * W24_LDR), q2 = w24_lds4(rp + 256 * W24_LDR), q3 = w24_lds4(rp + 384 * W24_LDR);
        w24_v4 yv[2];
        yv[0] = (q0 + q1) + q2;
        yv[1] = (q1 - q2) - q3;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            if (!(okx && oyb + 4 * it + aa < a.H)) continue;
            w24_v4 v = __builtin_elementwise_fma(yv[aa], sc, sh);
            if (a_res) v += rr[it][aa];
            if (relu) v = (w24_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};
            *reinterpret_cast<w24_v4*>(op + (4 * it + aa) * ostep) = v;
        }
    }
}

// Measured and NOT kept (same box, same process, tools/bench_conv.py): pinning the transform in front of the MFMA block with empty
// asm statements, the literal -5 form of the column transform (two scalar v_fma_f32 per packed one), a run-time stage-buffer
// offset (13 more address VALU per stage) -- all within the run-to-run noise of +-2 %: the kernel is not bound by its VALU count
// (PMC: 63 % of a wave's cycles wait for issue behind the other resident waves / barriers, 20 % sit in s_waitcnt).
// one block: tile t_ (already XCD-remapped) of the launch / group member described by (a, gd)
template <bool FAST>
__device__ __forceinline__ void w24_block(const ConvArgs& a, const W24Grid& gd, int t_, float* smem)
{
    constexpr int CPS = W24_KS / 8;                    // chunks per stage
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int NTILES = (a.Cout + 31) >> 5;
    int q_;
    q_ = w24_div(t_, gd.ntb, gd.mNtb); const int nb = t_ - q_ * gd.ntb; t_ = q_;
    q_ = w24_div(t_, gd.tilesX, gd.mTx); const int tx = t_ - q_ * gd.tilesX; t_ = q_;
    q_ = w24_div(t_, gd.tilesY, gd.mTy); const int ty = t_ - q_ * gd.tilesY;
    const int b = q_;
    const int y0 = ty * 16, x0 = tx * 16;
    const int C = a.srcC[0], ld = a.srcLd[0];
    const int KC = C >> 3;
    const float* __restrict__ x = a.src[0];
    const bool interior = y0 >= 1 && y0 + 17 <= a.H && x0 >= 1 && x0 + 17 <= a.W;

    // ---- staging slots (fixed per thread): patch pixel pp, channel group q
    int go[W24_SLOTS], lo[W24_SLOTS];
    bool ok[W24_SLOTS];
#pragma unroll
    for (int s = 0; s < W24_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        const int pp = idx / W24_CGS, q = idx % W24_CGS;
        const int py = pp / W24_PW, px = pp - py * W24_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inl = idx < W24_F4;
        ok[s] = inl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        go[s] = ok[s] ? ((b * a.H + gy) * a.W + gx) * ld + q * 4 : 0;
        lo[s] = inl ? q * W24_CG + (py * W24_PWP + (px ^ ((py >> 1) & 3))) * 4 : -1;
    }
    float4 rg[W24_SLOTS];
    auto stage_load = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < W24_SLOTS; ++s) rg[s] = ig_ldg4(x + go[s] + c0);
    };
    auto stage_store = [&](float* buf) __attribute__((always_inline)) {
        // slots 0 .. SLOTS-2 are inside the patch for every thread ((s + 1) * 256 <= F4): only the last one needs the check
        if (interior) {
#pragma unroll
            for (int s = 0; s < W24_SLOTS; ++s)
                if ((s + 1) * IG_THREADS <= W24_F4 || lo[s] >= 0) *reinterpret_cast<float4*>(buf + lo[s]) = rg[s];
        } else {
#pragma unroll
            for (int s = 0; s < W24_SLOTS; ++s)
                if ((s + 1) * IG_THREADS <= W24_F4 || lo[s] >= 0) {
                    float4 v = rg[s];
                    if (!ok[s]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(buf + lo[s]) = v;
                }
        }
    };

    // ---- wave's transform row (F(2,3)): t[j] = d[rA][j] + sg * d[rB][j]
    const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rB = xi == 2 ? 1 : (xi == 3 ? 3 : 2);
    const float sg1 = xi == 1 ? 1.f : -1.f;
    const w24_v2 sg = {sg1, sg1};
    const int tyy = m >> 2, txx = m & 3;
    const int sA = (tyy + (rA >> 1)) & 3, sB = (tyy + (rB >> 1)) & 3;     // swizzle of the lane's two patch rows
    const int baseA = ((2 * tyy + rA) * W24_PWP + 4 * txx) * 4 + h * W24_CG;
    const int baseB = ((2 * tyy + rB) * W24_PWP + 4 * txx) * 4 + h * W24_CG;
    int offA[4], offB[4];                      // column jj of an aligned group of four lives in slot jj ^ s
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { offA[jj] = baseA + ((jj ^ sA) << 2); offB[jj] = baseB + ((jj ^ sB) << 2); }

    // ---- U fragments: [xi][ntile][kc][nu 6][lane][4]
    int tile = nb;
    if (tile >= NTILES) tile = NTILES - 1;
    const float* ub = a.w + ((size_t)(xi * NTILES + tile) * KC) * 1536 + lane * 4;
    float4 bq[2][6];
    auto load_u = [&](int kc, float4 (&dst)[6]) __attribute__((always_inline)) {
        const int kk = kc < KC ? kc : KC - 1;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) dst[nu] = ig_ldg4(ub + (size_t)kk * 1536 + nu * 256);
    };

    const int nstage = C / W24_KS;
    stage_load(0);
    load_u(0, bq[0]);
    stage_store(smem);
    if (tid < 4) *reinterpret_cast<float4*>(smem + W24_MAIN + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    f32x16 acc[6];
    {
        const w24_v4* zp = reinterpret_cast<const w24_v4*>(smem + W24_MAIN);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                w24_v4 z = zp[r4];
                asm volatile("" : "+v"(z));
                acc[nu][4 * r4] = z.x; acc[nu][4 * r4 + 1] = z.y; acc[nu][4 * r4 + 2] = z.z; acc[nu][4 * r4 + 3] = z.w;
            }
    }

    // One stage = 16 channels = two 8-channel chunks out of stage buffer BUF (a compile-time constant: the stage loop is unrolled
    // by two so that every LDS address is lane base + immediate -- as a run-time buffer offset the compiler spent 13 VALU per
    // stage on ds_read / ds_write addresses, and every VALU instruction takes issue cycles the matrix pipe cannot use).
    auto stage = [&](int st, auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const float* buf = smem + BUF * W24_STAGE;
        const bool more = st + 1 < nstage;
#pragma unroll
        for (int ch = 0; ch < CPS; ++ch) {
            const float* pc = buf + ch * 2 * W24_CG;
            w24_v2 tl[6], th[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const w24_v4 da = w24_lds4(pc + offA[j & 3] + (j >> 2) * 16);
                const w24_v4 db = w24_lds4(pc + offB[j & 3] + (j >> 2) * 16);
                tl[j] = __builtin_elementwise_fma(sg, db.xy, da.xy);
                th[j] = __builtin_elementwise_fma(sg, db.zw, da.zw);
            }
            // V = t B4: six column frequencies per channel pair in 12 packed ops, written so that every constant is an inline
            // operand of v_pk_fma_f32 (4, -4, 2, -2): 4 t0 - 5 t2 + t4 = 4 (t0 - t2) + (t4 - t2), 4 t1 - 5 t3 + t5 = (t5 - t3) - 4 (t3 - t1)
            // (with the literal -5 the compiler split each of those into two scalar v_fma_f32)
            w24_v2 vl[6], vh[6];
            {
                const w24_v2 k4 = {4.f, 4.f}, kn4 = {-4.f, -4.f}, k2 = {2.f, 2.f}, kn2 = {-2.f, -2.f};
                const w24_v2 al = __builtin_elementwise_fma(kn4, tl[2], tl[4]), ah = __builtin_elementwise_fma(kn4, th[2], th[4]);
                const w24_v2 bl = __builtin_elementwise_fma(kn4, tl[1], tl[3]), bh = __builtin_elementwise_fma(kn4, th[1], th[3]);
                const w24_v2 cl = tl[4] - tl[2], chh = th[4] - th[2];
                const w24_v2 fl = tl[3] - tl[1], fh = th[3] - th[1];
                vl[0] = __builtin_elementwise_fma(k4, tl[0] - tl[2], cl); vh[0] = __builtin_elementwise_fma(k4, th[0] - th[2], chh);
                vl[5] = __builtin_elementwise_fma(kn4, fl, tl[5] - tl[3]); vh[5] = __builtin_elementwise_fma(kn4, fh, th[5] - th[3]);
                vl[1] = al + bl; vh[1] = ah + bh;
                vl[2] = al - bl; vh[2] = ah - bh;
                vl[3] = __builtin_elementwise_fma(k2, fl, cl); vh[3] = __builtin_elementwise_fma(k2, fh, chh);
                vl[4] = __builtin_elementwise_fma(kn2, fl, cl); vh[4] = __builtin_elementwise_fma(kn2, fh, chh);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                if (nu == 1) {                                            // next chunk's U: after 4 of the 24 MFMAs
                    load_u(st * CPS + ch + 1, bq[(ch + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nu == 3 && ch == 0 && more) {                         // next stage's patch: younger than every U load used in this stage
                    stage_load((st + 1) * W24_KS);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nu == 4 && ch == CPS - 1 && more) {                   // hand-over into the other stage buffer from inside the MFMA block
                    stage_store(smem + (BUF ^ 1) * W24_STAGE);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float4 bb = bq[ch & 1][nu];
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, vl[nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, vl[nu].y, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, vh[nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, vh[nu].y, acc[nu], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int st = 0; st < nstage; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        if (st + 1 < nstage) stage(st + 1, std::integral_constant<int, 1>{});
    }

    if (nb < NTILES) {            // block-uniform (ragged last channel block computes a duplicate that is never stored)
        w24_output_all<FAST>(a, smem, acc, xi, h, m, tid, b, y0, x0, nb);
    }
}

template <bool FAST>
__global__ __launch_bounds__(IG_THREADS, 2) void conv3x3_wino24_kernel(const ConvArgs a, const W24Grid gd)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    w24_block<FAST>(a, gd, ig_xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Up to four INDEPENDENT convolutions in one launch (HRNet: the same conv of every parallel branch, pose_higher_hrnet.py:217-235 --
// 32 ch @128x128, 64 @64x64, 128 @32x32, 256 @16x16 at B = 8 are 512 / 256 / 128 / 64 blocks: launched one by one the small ones leave
// most of the 256 CUs idle, and ROCm 7.2 captures at most two streams).  Member k owns the tiles [first[k], first[k + 1]); members are
// ordered longest block first (most input channels), so the short blocks fill the tail.
struct W24Group {
    ConvArgs a[4];
    W24Grid gd[4];
    int first[5];
    int n;
};
template <bool FAST>
__global__ __launch_bounds__(IG_THREADS, 2) void conv3x3_wino24_group_kernel(const W24Group g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // NO XCD remap here: it hands each XCD a contiguous tile range, i.e. all of the longest member's blocks to XCD 0 (measured: 83 us per
    // group against 75 us for the four launches one by one); consecutive block ids go round-robin over the XCDs, which spreads every member
    const int t = blockIdx.x;
    int k = 0;
    while (k + 1 < g.n && t >= g.first[k + 1]) ++k;            // scalar
    w24_block<FAST>(g.a[k], g.gd[k], t - g.first[k], smem);
}

// the lean epilogue's preconditions (w24_output_all<true>)
static bool w24_fast_epilogue(const ConvArgs& a)
{
    return (a.outLd & 3) == 0 && (a.Cout & 3) == 0 && (((size_t)a.out) & 15) == 0 &&
           (!a.res || ((a.resLd & 3) == 0 && (((size_t)a.res) & 15) == 0)) && (a.act == CP_ACT_NONE || a.act == CP_ACT_RELU) &&
           (long long)a.W * a.outLd * 8 < (1ll << 31) && (long long)a.W * a.resLd * 8 < (1ll << 31);
}

static unsigned w24_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

// a.w = F(2x4) Winograd-domain weights from cp_winograd24_pack_f32.  Returns -1 when the shape is not eligible.
int cp_launch_conv3x3_wino24(const ConvArgs& a, hipStream_t s)
{
    const bool ok = a.nsrc == 1 && a.kh == 3 && a.kw == 3 && a.sy == 1 && a.sx == 1 && a.py == 1 && a.px == 1 &&
                    !a.outNCHW && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.Ho == a.H && a.Wo == a.W &&
                    a.OH == a.H && a.OW == a.W && a.srcC[0] % 16 == 0 && a.srcLd[0] % 4 == 0 && a.ksplit == 1 &&
                    (((size_t)a.src[0] | (size_t)a.w) & 15) == 0 &&
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31);
    if (!ok) return -1;
    const int smem = W24_SMEM_FLOATS * 4;
    const bool fast = w24_fast_epilogue(a);
    static CpLdsGuard guard[2];
    {
        const hipError_t e = guard[fast].ensure(fast ? (const void*)conv3x3_wino24_kernel<true> : (const void*)conv3x3_wino24_kernel<false>, smem);
        if (e != hipSuccess) { cp_set_error("conv3x3_winograd24: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    W24Grid gd;
    gd.tilesX = cp_cdiv(a.W, 16); gd.tilesY = cp_cdiv(a.H, 16);
    gd.ntb = (a.Cout + 31) / 32;
    gd.mNtb = w24_magic(gd.ntb); gd.mTx = w24_magic(gd.tilesX); gd.mTy = w24_magic(gd.tilesY);
    const long long grid = (long long)a.B * gd.tilesX * gd.tilesY * gd.ntb;
    const long long dmax = gd.ntb > gd.tilesX ? (gd.ntb > gd.tilesY ? gd.ntb : gd.tilesY) : (gd.tilesX > gd.tilesY ? gd.tilesX : gd.tilesY);
    if (grid * dmax >= (1ll << 32)) { cp_set_error("conv3x3_winograd24: grid %lld too large", grid); return 1; }
    if (fast) hipLaunchKernelGGL(conv3x3_wino24_kernel<true>, dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, gd);
    else hipLaunchKernelGGL(conv3x3_wino24_kernel<false>, dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, gd);
    cp_note_kernel(fast ? "conv3x3_wino24_kernel<true>" : "conv3x3_wino24_kernel<false>");      // as rocprofv3 prints the instantiation
    return 0;
}

// n <= 4 independent eligible convolutions (each a[i].w = its own cp_winograd24_pack_f32 weights) as ONE launch.
int cp_launch_conv3x3_wino24_group(const ConvArgs* a, int n, hipStream_t s)
{
    if (n < 1 || n > 4) { cp_set_error("conv3x3_winograd24_group: 1..4 members (got %d)", n); return 1; }
    W24Group g;
    g.n = n;
    bool fast = true;
    long long total = 0, dmax = 1;
    for (int i = 0; i < n; ++i) {
        const ConvArgs& c = a[i];
        const bool ok = c.nsrc == 1 && c.kh == 3 && c.kw == 3 && c.sy == 1 && c.sx == 1 && c.py == 1 && c.px == 1 &&
                        !c.outNCHW && c.osy == 1 && c.osx == 1 && c.ooy == 0 && c.oox == 0 && c.Ho == c.H && c.Wo == c.W &&
                        c.OH == c.H && c.OW == c.W && c.srcC[0] % 16 == 0 && c.srcLd[0] % 4 == 0 && c.ksplit == 1 &&
                        (((size_t)c.src[0] | (size_t)c.w) & 15) == 0 && (long long)c.B * c.H * c.W * c.srcLd[0] < (1ll << 31);
        if (!ok) { cp_set_error("conv3x3_winograd24_group: member %d is not a 3x3 / stride 1 / pad 1 NHWC convolution", i); return 1; }
        g.a[i] = c;
        fast = fast && w24_fast_epilogue(c);
        W24Grid& gd = g.gd[i];
        gd.tilesX = cp_cdiv(c.W, 16); gd.tilesY = cp_cdiv(c.H, 16);
        gd.ntb = (c.Cout + 31) / 32;
        gd.mNtb = w24_magic(gd.ntb); gd.mTx = w24_magic(gd.tilesX); gd.mTy = w24_magic(gd.tilesY);
        g.first[i] = (int)total;
        const long long blocks = (long long)c.B * gd.tilesX * gd.tilesY * gd.ntb;
        total += blocks;
        const long long d = gd.ntb > gd.tilesX ? (gd.ntb > gd.tilesY ? gd.ntb : gd.tilesY) : (gd.tilesX > gd.tilesY ? gd.tilesX : gd.tilesY);
        if (blocks * d >= (1ll << 32)) { cp_set_error("conv3x3_winograd24_group: member %d too large", i); return 1; }
        if (d > dmax) dmax = d;
    }
    for (int i = n; i < 5; ++i) g.first[i] = (int)total;
    for (int i = n; i < 4; ++i) { g.a[i] = g.a[0]; g.gd[i] = g.gd[0]; }
    if (total >= (1ll << 31)) { cp_set_error("conv3x3_winograd24_group: grid %lld too large", total); return 1; }
    const int smem = W24_SMEM_FLOATS * 4;
    static CpLdsGuard guard[2];
    {
        const hipError_t e = guard[fast].ensure(fast ? (const void*)conv3x3_wino24_group_kernel<true> : (const void*)conv3x3_wino24_group_kernel<false>, smem);
        if (e != hipSuccess) { cp_set_error("conv3x3_winograd24_group: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    if (fast) hipLaunchKernelGGL(conv3x3_wino24_group_kernel<true>, dim3((unsigned)total), dim3(IG_THREADS), smem, s, g);
    else hipLaunchKernelGGL(conv3x3_wino24_group_kernel<false>, dim3((unsigned)total), dim3(IG_THREADS), smem, s, g);
    cp_note_kernel(fast ? "conv3x3_wino24_group_kernel<true>" : "conv3x3_wino24_group_kernel<false>");
    return 0;
}

// ---- weight transform: packed direct weights [rows >= Cout][9*C] (k = (ky*3+kx)*C + c)  ->  U = G2 g G4^T in fragment order
//   value(lane, j) = U[xi][nu][n = ntile*32 + lane%32][c = kc*8 + (lane/32)*4 + j]      (zero for n >= Cout)
__global__ void wino24_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int C, int Cout, int ntiles, long long total)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int KC = C >> 3;
    long long t = idx;
    const int j = (int)(t & 3); t >>= 2;
    const int lane = (int)(t & 63); t >>= 6;
    const int nu = (int)(t % 6); t /= 6;
    const int kc = (int)(t % KC); t /= KC;
    const int nt = (int)(t % ntiles);
    const int xi = (int)(t / ntiles);
    const int n = nt * 32 + (lane & 31), c = kc * 8 + (lane >> 5) * 4 + j;
    float r = 0.f;
    if (n < Cout) {
        const double G2[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        const double G4[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
        const float* g = w + (size_t)n * 9 * C + c;
        double accd = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) accd += G2[xi][ky] * (double)g[(ky * 3 + kx) * C] * G4[nu][kx];
        r = (float)accd;
    }
    u[idx] = r;
}

extern "C" size_t cp_winograd24_weight_floats(int C, int Cout)
{
    if (C <= 0 || Cout <= 0 || C % 8) return 0;
    return (size_t)24 * ((Cout + 31) / 32) * 32 * C;
}

extern "C" int cp_winograd24_pack_f32(const float* w, float* u, int C, int Cout, void* stream)
{
    CP_CHECK_ARG(w && u, "winograd24_pack: null pointer");
    CP_CHECK_ARG(C > 0 && C % 16 == 0 && Cout > 0, "winograd24_pack: C=%d must be a positive multiple of 16 (Cout=%d)", C, Cout);
    const int ntiles = (Cout + 31) / 32;
    const long long total = (long long)cp_winograd24_weight_floats(C, Cout);
    hipLaunchKernelGGL(wino24_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, u, C, Cout,
                       ntiles, total);
    CP_CHECK_LAUNCH("wino24_pack_kernel");
    return 0;
}
