// KeypointHead branch  conv3x3(64 -> head_conv) + bias + ReLU -> conv1x1(head_conv -> n) (+ bias, + sigmoid)  as ONE launch on the
// Winograd F(2x4,3x3) transform, V-STATIONARY (lib/models/heads/keypoint.py:14-37; multi_pose.py:35-37 for the sigmoid).
//
// conv3x3_wino_vs64_kernel (conv3x3_wino.hip) does this with F(2x2,3x3): the wave's transformed input V for all 64 channels stays in
// registers (128 VGPRs) while the block loops over the branch's output-channel tiles, so the inner loop is nothing but U loads and
// MFMAs.  F(2x4) executes 0.75x the MFMAs, but a row-owning wave would need V for SIX column frequencies (192 VGPRs).  Here a block has
// EIGHT waves: wave (xi, vh) owns transform row xi and the column frequencies 3 vh .. 3 vh + 2 -- 96 VGPRs of V, 48 accumulators:
//   * block = 16 x 16 output pixels = 8 x 4 tiles of 2 x 4 (one 32-wide MFMA tile), 512 threads, one block per CU (8 waves = 2 per SIMD);
//   * the whole 64-channel 18 x 18 halo patch is staged ONCE (92 KB, conv3x3_wino24.hip's swizzled layout); a lane reads two rows x five
//     columns per 8-channel chunk (frequencies 0..2 need columns 0..4, frequencies 3..5 columns 1..5) and forms its three frequencies
//     with 10 + 12 packed VALU ops; after that the patch region becomes the reduction buffers;
//   * per output-channel tile: 8 chunks x 3 frequencies x 4 = 96 v_mfma_f32_32x32x2f32 per wave, U (cp_winograd24_pack_f32 order
//     [xi][n-tile][chunk][nu 6][lane][4]) streaming global -> registers one chunk ahead from the middle of the MFMA block;
//   * epilogue per tile, two passes of two output columns: every wave adds ITS frequencies' share of A4 in registers and writes it to
//     red[wave 8][col 2][tile 32][36] (73.7 KB); after one barrier each thread sums the eight shares of one (tile, column, 4 channels)
//     item with A2 across xi, applies bias + ReLU and
//       - n <= 2 outputs (hm, wh, reg, hp_offset): multiplies with the 1x1 weights into per-thread partial sums (lane-reduced at the end);
//         the two passes use two buffers, so a channel tile costs two barriers;
//       - more outputs (hps 34, hm_hp 17): writes the tile to `mid` [256 px][32 ch] and the 1x1 runs as a second MFMA phase,
//         D[out j][pixel] += w2[j][c] * mid[pixel][c], 16 MFMAs per wave and tile into ONE accumulator that lives in registers across the
//         tiles (no parking: this kernel has the registers); one reduction buffer + mid = 110.6 KB, four barriers per tile.
// fp32 throughout; sum orders differ from the two-launch form as in the F(2x2) head kernel (<= 1e-5 relative, inside the 1e-3 bar).
#include "igemm.h"

#define HW_THREADS 512
#define HW_PH 18
#define HW_PW 18
#define HW_PWP 20
#define HW_C 64
#define HW_CGS (HW_C / 4)
#define HW_CG (HW_PH * HW_PWP * 4)                       // floats per 4-channel plane (1440)
#define HW_PATCH (HW_CGS * HW_CG)                        // 23040 floats = 92.2 KB
#define HW_F4 (HW_PH * HW_PW * HW_CGS)                   // 5184 float4
#define HW_SLOTS ((HW_F4 + HW_THREADS - 1) / HW_THREADS) // 11
#define HW_LDR 36
#define HW_RED (16 * 32 * HW_LDR)                        // [wave 8][col 2][tile 32][36] floats = 73.7 KB
#define HW_LDM 36
#define HW_MID (256 * HW_LDM)                            // [pixel 256][32 ch] floats = 36.9 KB
#define HW_SMEM_FLOATS (2 * HW_RED)                      // two reduction buffers (147.5 KB) >= the patch
#define HW_SMEM_FLOATS_MM (HW_RED + HW_MID > HW_PATCH ? HW_RED + HW_MID : HW_PATCH)

typedef float hw_v2 __attribute__((ext_vector_type(2)));
typedef float hw_v4 __attribute__((ext_vector_type(4)));

struct HwHead {
    const float* w2;     // [n2][ld2] 1x1 weights, row-major over the mid channels
    const float* b2;     // [n2]
    float* out2;         // NCHW [B][n2][H][W]
    int n2, ld2, act2;
};
struct HwGrid {
    int tilesX, tilesY;
    unsigned mTx, mTy;
};
__device__ __forceinline__ hw_v4 hw_lds4(const float* p) { return *reinterpret_cast<const hw_v4*>(p); }
__device__ __forceinline__ int hw_div(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }

// N2: outputs of the 1x1 on the register path (MM: outputs 32 .. 32 + N2 - 1); MM = 1: second MFMA phase for outputs 0..31
template <int N2, int MM>
__global__ __launch_bounds__(HW_THREADS, 1) void head_wino24_kernel(const ConvArgs a, const HwGrid gd, const HwHead hd)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wv & 3, vh = wv >> 2;
    const int h = lane >> 5, m = lane & 31;
    const int NTILES = (a.Cout + 31) >> 5;
    int t_ = ig_xcd_remap(blockIdx.x, gridDim.x), q_;
    q_ = hw_div(t_, gd.tilesX, gd.mTx); const int tx = t_ - q_ * gd.tilesX; t_ = q_;
    q_ = hw_div(t_, gd.tilesY, gd.mTy); const int ty = t_ - q_ * gd.tilesY;
    const int b = q_;
    const int y0 = ty * 16, x0 = tx * 16;
    const int ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];

    // ---- stage the whole 64-channel halo patch: [cg 16][py 18][px ^ s(py)][4], s(py) = (py >> 1) & 3 on the low two bits of px
    {
        float4 rg[HW_SLOTS];
        int lo[HW_SLOTS];
        bool ok[HW_SLOTS];
#pragma unroll
        for (int s = 0; s < HW_SLOTS; ++s) {
            const int idx = tid + s * HW_THREADS;
            const int pp = idx / HW_CGS, q = idx % HW_CGS;
            const int py = pp / HW_PW, px = pp - py * HW_PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool inl = idx < HW_F4;
            ok[s] = inl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int go = ok[s] ? ((b * a.H + gy) * a.W + gx) * ld + q * 4 : 0;
            lo[s] = inl ? ((q * HW_PH + py) * HW_PWP + (px ^ ((py >> 1) & 3))) * 4 : -1;
            rg[s] = ig_ldg4(x + go);
        }
#pragma unroll
        for (int s = 0; s < HW_SLOTS; ++s)
            if (lo[s] >= 0) {
                float4 v = rg[s];
                if (!ok[s]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(smem + lo[s]) = v;
            }
    }
    // first U chunk in flight under the transform: [xi][ntile][kc][nu 6][lane][4]; the wave's three frequencies start at nu = 3 vh
    const float* up = a.w + ((size_t)xi * NTILES * (HW_C / 8)) * 1536 + (size_t)(3 * vh) * 256 + lane * 4;
    const int nlin = NTILES * (HW_C / 8);                    // chunks this wave walks through, 1536 floats apart
    float4 bq[2][3];
#pragma unroll
    for (int nu = 0; nu < 3; ++nu) bq[0][nu] = ig_ldg4(up + nu * 256);
    __syncthreads();

    // ---- V for the wave's row xi and frequencies 3 vh .. 3 vh + 2, all 8 chunks, kept in registers
    const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rB = xi == 2 ? 1 : (xi == 3 ? 3 : 2);
    const float sg1 = xi == 1 ? 1.f : -1.f;
    const hw_v2 sg = {sg1, sg1};
    const int tyy = m >> 2, txx = m & 3;
    const int sA = (tyy + (rA >> 1)) & 3, sB = (tyy + (rB >> 1)) & 3;
    const int baseA = ((2 * tyy + rA) * HW_PWP + 4 * txx) * 4 + h * HW_CG;
    const int baseB = ((2 * tyy + rB) * HW_PWP + 4 * txx) * 4 + h * HW_CG;
    int offA[4], offB[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { offA[jj] = baseA + ((jj ^ sA) << 2); offB[jj] = baseB + ((jj ^ sB) << 2); }
    hw_v2 vl[HW_C / 8][3], vhh[HW_C / 8][3];
#pragma unroll
    for (int kc = 0; kc < HW_C / 8; ++kc) {
        const float* pc = smem + kc * 2 * HW_CG;
        hw_v2 tl[5], th[5];                                   // columns c0 .. c0 + 4 of the row combination, c0 = vh
        if (vh == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const hw_v4 da = hw_lds4(pc + offA[j & 3] + (j >> 2) * 16), db = hw_lds4(pc + offB[j & 3] + (j >> 2) * 16);
                tl[j] = __builtin_elementwise_fma(sg, db.xy, da.xy);
                th[j] = __builtin_elementwise_fma(sg, db.zw, da.zw);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int jc = j + 1;
                const hw_v4 da = hw_lds4(pc + offA[jc & 3] + (jc >> 2) * 16), db = hw_lds4(pc + offB[jc & 3] + (jc >> 2) * 16);
                tl[j] = __builtin_elementwise_fma(sg, db.xy, da.xy);
                th[j] = __builtin_elementwise_fma(sg, db.zw, da.zw);
            }
        }
        const hw_v2 k4 = {4.f, 4.f}, kn4 = {-4.f, -4.f}, k2 = {2.f, 2.f}, kn2 = {-2.f, -2.f};
        if (vh == 0) {                  // t[0..4] = columns 0..4:  v0 = 4 (t0 - t2) + (t4 - t2);  v1 / v2 = (t4 - 4 t2) +- (t3 - 4 t1)
            const hw_v2 al = __builtin_elementwise_fma(kn4, tl[2], tl[4]), ah = __builtin_elementwise_fma(kn4, th[2], th[4]);
            const hw_v2 bl = __builtin_elementwise_fma(kn4, tl[1], tl[3]), bh = __builtin_elementwise_fma(kn4, th[1], th[3]);
            vl[kc][0] = __builtin_elementwise_fma(k4, tl[0] - tl[2], tl[4] - tl[2]); vhh[kc][0] = __builtin_elementwise_fma(k4, th[0] - th[2], th[4] - th[2]);
            vl[kc][1] = al + bl; vhh[kc][1] = ah + bh;
            vl[kc][2] = al - bl; vhh[kc][2] = ah - bh;
        } else {                        // t[0..4] = columns 1..5:  v3 / v4 = (t4 - t2) +- 2 (t3 - t1);  v5 = (t5 - t3) - 4 (t3 - t1)
            const hw_v2 cl = tl[3] - tl[1], ch = th[3] - th[1];          // t4 - t2
            const hw_v2 fl = tl[2] - tl[0], fh = th[2] - th[0];          // t3 - t1
            vl[kc][0] = __builtin_elementwise_fma(k2, fl, cl); vhh[kc][0] = __builtin_elementwise_fma(k2, fh, ch);
            vl[kc][1] = __builtin_elementwise_fma(kn2, fl, cl); vhh[kc][1] = __builtin_elementwise_fma(kn2, fh, ch);
            vl[kc][2] = __builtin_elementwise_fma(kn4, fl, tl[4] - tl[2]); vhh[kc][2] = __builtin_elementwise_fma(kn4, fh, th[4] - th[2]);
        }
#pragma unroll
        for (int nu = 0; nu < 3; ++nu) asm volatile("" : "+v"(vl[kc][nu]), "+v"(vhh[kc][nu]));     // keep the chunk's transform here (cf. conv3x3_wino_vs64_kernel)
        __builtin_amdgcn_sched_barrier(0);                    // ... and the next chunk's patch reads behind it (all 80 hoisted = 320 VGPRs in flight)
    }
    __syncthreads();                // every wave has read its rows of the patch: the region becomes the reduction buffers / mid

    float acc2[2][2][N2 > 0 ? N2 : 1];                      // [pass][row aa][output] partial sums of the register-path 1x1
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < (N2 > 0 ? N2 : 1); ++j) acc2[i][k][j] = 0.f;
    f32x16 acc2m;                                            // MM: D[out j][pixel] of the wave's 32 pixels
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2m[r] = 0.f;
    constexpr int J0 = MM ? 32 : 0;
    float* const mid = smem + HW_RED;                        // MM only
    const int n4 = tid & 7, bb = (tid >> 3) & 1, mi = tid >> 4;          // the thread's item per pass: (tile mi, column 2 P + bb, 4 channels n4)
    const int ity = mi >> 2, itx = mi & 3;
    const int itrd = (bb * 32 + mi) * HW_LDR + n4 * 4;
    const int wpo = (wv * 64 + m) * HW_LDR + 4 * h;                     // the wave's share: red[wave][col][tile][36]

    int lin = 0;
    if constexpr (!MM) {
        // ---- n <= 2 outputs: the reduction of tile nt - 1 is DEFERRED into the MFMA loop of tile nt.  With one 8-wave block per CU
        // nobody else feeds the matrix pipe while all eight waves sit in an epilogue (first version: 6 100 of a tile's 18 400 cycles);
        // here a tile boundary costs two barriers around 16 ds_write_b128, and the 16 ds_read_b128 + ~100 VALU of the reduction are
        // issued between the MFMAs of the next tile (reads after chunks 0 / 4, arithmetic after chunks 2 / 6).  Both passes' shares
        // must survive until then: two reduction buffers (147.5 KB).
        hw_v4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, w2r[N2];          // of the tile being reduced (nt - 1)
#pragma unroll
        for (int j = 0; j < N2; ++j) w2r[j] = sc;
        hw_v4 rq[8];                                             // the item's eight shares, read two chunks before they are summed
        auto red_read = [&](int P) __attribute__((always_inline)) {
            int rbase = itrd + P * HW_RED;
            asm volatile("" : "+v"(rbase));                      // opaque base: the second buffer is beyond the 16-bit ds immediate
            const float* rp = smem + rbase;
#pragma unroll
            for (int w = 0; w < 8; ++w) rq[w] = hw_lds4(rp + w * 64 * HW_LDR);
        };
        auto red_sum = [&](int P) __attribute__((always_inline)) {
            const hw_v4 r0 = rq[0] + rq[4], r1 = rq[1] + rq[5], r2 = rq[2] + rq[6], r3 = rq[3] + rq[7];     // xi = 0..3, both frequency halves
            hw_v4 yv[2];
            yv[0] = (r0 + r1) + r2;
            yv[1] = (r1 - r2) - r3;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                hw_v4 v = __builtin_elementwise_fma(yv[aa], sc, sh);
                v = (hw_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};          // the head's ReLU (keypoint.py:17,21,...)
#pragma unroll
                for (int j = 0; j < N2; ++j) {
                    const hw_v4 w = w2r[j];
                    acc2[P][aa][j] = fmaf(v.w, w.w, fmaf(v.z, w.z, fmaf(v.y, w.y, fmaf(v.x, w.x, acc2[P][aa][j]))));
                }
            }
        };
#pragma unroll 1
        for (int nt = 0; nt < NTILES; ++nt) {
            const bool prev = nt > 0;
            f32x16 acc[3];
#pragma unroll
            for (int nu = 0; nu < 3; ++nu)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < HW_C / 8; ++kc) {
                ++lin;
                const float* un = up + (size_t)(lin < nlin ? lin : nlin - 1) * 1536;      // next chunk (next tile's first after the 8th)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nu = 0; nu < 3; ++nu) {
                    if (nu == 1) {                                   // next chunk's U after 4 of the chunk's 12 MFMAs
#pragma unroll
                        for (int q = 0; q < 3; ++q) bq[(kc + 1) & 1][q] = ig_ldg4(un + q * 256);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (nu == 2 && prev) {                           // the previous tile's reduction rides between this tile's MFMAs
                        if (kc == 0) red_read(0);
                        if (kc == 2) red_sum(0);
                        if (kc == 4) red_read(1);
                        if (kc == 6) red_sum(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float4 bbv = bq[kc & 1][nu];
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.x, vl[kc][nu].x, acc[nu], 0, 0, 0);
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.y, vl[kc][nu].y, acc[nu], 0, 0, 0);
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.z, vhh[kc][nu].x, acc[nu], 0, 0, 0);
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.w, vhh[kc][nu].y, acc[nu], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this tile's scale / shift / 1x1 weights, consumed inside the next tile's loop
            const int ncol = nt * 32 + n4 * 4;
            sc = *reinterpret_cast<const hw_v4*>(a.scale + ncol);
            sh = *reinterpret_cast<const hw_v4*>(a.shift + ncol);
#pragma unroll
            for (int j = 0; j < N2; ++j) w2r[j] = *reinterpret_cast<const hw_v4*>(hd.w2 + (size_t)j * hd.ld2 + ncol);
            if (prev) __syncthreads();                               // every thread has read the previous tile's shares
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                int wbase = wpo + P * HW_RED;
                asm volatile("" : "+v"(wbase));
                float* wq = smem + wbase;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hw_v4 q[3];
#pragma unroll
                    for (int nu = 0; nu < 3; ++nu) q[nu] = (hw_v4){acc[nu][4 * j], acc[nu][4 * j + 1], acc[nu][4 * j + 2], acc[nu][4 * j + 3]};
                    hw_v4 c0, c1;
                    if (vh == 0) {                                   // m0, m1, m2
                        const hw_v4 s1 = q[1] + q[2], d1 = q[1] - q[2];
                        c0 = P == 0 ? q[0] + s1 : s1;
                        c1 = d1;
                    } else {                                         // m3, m4, m5
                        const hw_v4 s2 = q[0] + q[1], d2 = q[0] - q[1];
                        c0 = P == 0 ? s2 : 4.f * s2;
                        c1 = P == 0 ? 2.f * d2 : 8.f * d2 + q[2];
                    }
                    *reinterpret_cast<hw_v4*>(wq + 8 * j) = c0;
                    *reinterpret_cast<hw_v4*>(wq + 32 * HW_LDR + 8 * j) = c1;
                }
            }
            __syncthreads();                                         // shares of tile nt complete
        }
        red_read(0); red_sum(0);                                     // the last tile's reduction has no MFMA loop to hide in
        red_read(1); red_sum(1);
    } else {
#pragma unroll 1
    for (int nt = 0; nt < NTILES; ++nt) {
        f32x16 acc[3];
#pragma unroll
        for (int nu = 0; nu < 3; ++nu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < HW_C / 8; ++kc) {
            ++lin;
            const float* un = up + (size_t)(lin < nlin ? lin : nlin - 1) * 1536;      // next chunk (next tile's first after the 8th)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) {
                if (nu == 1) {                                   // next chunk's U after 4 of the chunk's 12 MFMAs
#pragma unroll
                    for (int q = 0; q < 3; ++q) bq[(kc + 1) & 1][q] = ig_ldg4(un + q * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float4 bbv = bq[kc & 1][nu];
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.x, vl[kc][nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.y, vl[kc][nu].y, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.z, vhh[kc][nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbv.w, vhh[kc][nu].y, acc[nu], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- per-tile epilogue.  The thread's scale / shift / 1x1 weights (same channels in both passes) first: in flight under the barrier
        const int ncol = nt * 32 + n4 * 4;
        const hw_v4 sc = *reinterpret_cast<const hw_v4*>(a.scale + ncol), sh = *reinterpret_cast<const hw_v4*>(a.shift + ncol);
        hw_v4 w2r[N2 > 0 ? N2 : 1];
#pragma unroll
        for (int j = 0; j < N2; ++j) w2r[j] = *reinterpret_cast<const hw_v4*>(hd.w2 + (size_t)(J0 + j) * hd.ld2 + ncol);
        hw_v4 w2f[MM ? 4 : 1];
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            if (MM && P == 1) __syncthreads();                   // MM: one buffer for both passes -- pass 0's reads are done
            // the second buffer starts 73.7 KB into LDS, beyond the 16-bit immediate of ds_read / ds_write: an OPAQUE per-pass base
            // register keeps every access "base + immediate" (as compile-time constants the compiler materialised 16 address VGPRs,
            // and the kernel spilled)
            int wbase = wpo + (MM ? 0 : P * HW_RED), rbase = itrd + (MM ? 0 : P * HW_RED);
            asm volatile("" : "+v"(wbase), "+v"(rbase));
            float* wq = smem + wbase;
            // the wave's share of the two output columns 2P, 2P + 1 (A4 restricted to its three frequencies)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hw_v4 q[3];
#pragma unroll
                for (int nu = 0; nu < 3; ++nu) q[nu] = (hw_v4){acc[nu][4 * j], acc[nu][4 * j + 1], acc[nu][4 * j + 2], acc[nu][4 * j + 3]};
                hw_v4 c0, c1;
                if (vh == 0) {                                   // m0, m1, m2
                    const hw_v4 s1 = q[1] + q[2], d1 = q[1] - q[2];
                    c0 = P == 0 ? q[0] + s1 : s1;                // col 0: m0 + m1 + m2      col 2: m1 + m2
                    c1 = d1;                                     // col 1: m1 - m2           col 3: m1 - m2
                } else {                                         // m3, m4, m5
                    const hw_v4 s2 = q[0] + q[1], d2 = q[0] - q[1];
                    c0 = P == 0 ? s2 : 4.f * s2;                 // col 0: m3 + m4           col 2: 4 (m3 + m4)
                    c1 = P == 0 ? 2.f * d2 : 8.f * d2 + q[2];    // col 1: 2 (m3 - m4)       col 3: 8 (m3 - m4) + m5
                }
                *reinterpret_cast<hw_v4*>(wq + 8 * j) = c0;
                *reinterpret_cast<hw_v4*>(wq + 32 * HW_LDR + 8 * j) = c1;
            }
            if constexpr (MM) {
                if (P == 1) {       // the 48 accumulator registers are dead now: this wave's 1x1 weight fragments for the MFMA phase go out
                                    // here, in flight under the barrier and the reduction.  Lane (j = lane % 32, g = lane / 32) takes channels
                                    // nt*32 + 8s + 4g .. +3 of output row j (rows >= n2 read row n2 - 1 and are zeroed)
                    const int nmm = hd.n2 < 32 ? hd.n2 : 32;
                    const bool live = m < nmm;
                    const float* wrow = hd.w2 + (size_t)(live ? m : nmm - 1) * hd.ld2 + nt * 32 + 4 * h;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const float4 t4 = ig_ldg4(wrow + 8 * s4);
                        w2f[s4] = live ? (hw_v4){t4.x, t4.y, t4.z, t4.w} : (hw_v4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            __syncthreads();
            // the thread's item: eight shares -> two output rows (A2 over xi, both frequency halves)
            const float* rp = smem + rbase;
            hw_v4 r0 = hw_lds4(rp) + hw_lds4(rp + 4 * 64 * HW_LDR);                                   // xi = 0 (waves 0 and 4)
            const hw_v4 r1 = hw_lds4(rp + 64 * HW_LDR) + hw_lds4(rp + 5 * 64 * HW_LDR);               // xi = 1
            const hw_v4 r2 = hw_lds4(rp + 2 * 64 * HW_LDR) + hw_lds4(rp + 6 * 64 * HW_LDR);           // xi = 2
            const hw_v4 r3 = hw_lds4(rp + 3 * 64 * HW_LDR) + hw_lds4(rp + 7 * 64 * HW_LDR);           // xi = 3
            hw_v4 yv[2];
            yv[0] = (r0 + r1) + r2;
            yv[1] = (r1 - r2) - r3;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                hw_v4 v = __builtin_elementwise_fma(yv[aa], sc, sh);
                v = (hw_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};          // the head's ReLU (keypoint.py:17,21,...)
#pragma unroll
                for (int j = 0; j < N2; ++j) {
                    const hw_v4 w = w2r[j];
                    acc2[P][aa][j] = fmaf(v.w, w.w, fmaf(v.z, w.z, fmaf(v.y, w.y, fmaf(v.x, w.x, acc2[P][aa][j]))));
                }
                if constexpr (MM) {      // pixel (row 2 ity + aa, column 4 itx + 2 P + bb) of the 16 x 16 tile, row-major
                    *reinterpret_cast<hw_v4*>(mid + ((2 * ity + aa) * 16 + 4 * itx + 2 * P + bb) * HW_LDM + n4 * 4) = v;
                }
            }
        }
        if constexpr (MM) {
            __syncthreads();             // mid complete (and pass 1's reads of the reduction buffer done)
            // second MFMA phase: wave wv owns pixels 32 wv .. + 31 (two rows of 16): D[out j][pixel] += w2[j][c] * mid[pixel][c]
            const float* mp = mid + (32 * wv + m) * HW_LDM + 4 * h;
            hw_v4 mf[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) mf[s4] = hw_lds4(mp + 8 * s4);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].x, mf[s4].x, acc2m, 0, 0, 0);
                acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].y, mf[s4].y, acc2m, 0, 0, 0);
                acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].z, mf[s4].z, acc2m, 0, 0, 0);
                acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].w, mf[s4].w, acc2m, 0, 0, 0);
            }
            __syncthreads();             // the next tile's pass 0 writes the reduction buffer and (later) mid: all reads above are done
        }
    }

    }
    const size_t HW = (size_t)a.H * a.W;
    if constexpr (MM) {
        // D[out j][pixel]: lane (pixel 32 wv + m, row group h) holds outputs j = (r & 3) + 8 (r >> 2) + 4 h
        const int p = 32 * wv + m, ox = x0 + (p & 15), oy = y0 + (p >> 4);
        const int nmm = hd.n2 < 32 ? hd.n2 : 32;
        if (ox < a.W && oy < a.H) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j < nmm) hd.out2[((size_t)b * hd.n2 + j) * HW + (size_t)oy * a.W + ox] = cp_act(acc2m[r] + hd.b2[j], hd.act2);
            }
        }
    }
    if constexpr (N2 > 0) {
        // the 8 lanes tid & 7 = 0..7 hold the partial sums of one (tile, column) over 32 channels each: fixed-order lane sum
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            const int ox = x0 + 4 * itx + 2 * P + bb, oy = y0 + 2 * ity;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int j = 0; j < N2; ++j) {
                    float v = acc2[P][aa][j];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    if ((tid & 7) == 0 && J0 + j < hd.n2 && ox < a.W && oy + aa < a.H)
                        hd.out2[((size_t)b * hd.n2 + J0 + j) * HW + (size_t)(oy + aa) * a.W + ox] = cp_act(v + hd.b2[J0 + j], hd.act2);
                }
        }
    }
}

static unsigned hw_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int N2, int MM>
static int launch_head_wino24(const ConvArgs& a, hipStream_t s, const HwHead& hd)
{
    const int smem = (MM ? HW_SMEM_FLOATS_MM : HW_SMEM_FLOATS) * 4;
    static CpLdsGuard guard;
    {
        const hipError_t e = guard.ensure((const void*)head_wino24_kernel<N2, MM>, smem);
        if (e != hipSuccess) { cp_set_error("head_wino24: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    HwGrid gd;
    gd.tilesX = cp_cdiv(a.W, 16); gd.tilesY = cp_cdiv(a.H, 16);
    gd.mTx = hw_magic(gd.tilesX); gd.mTy = hw_magic(gd.tilesY);
    const long long grid = (long long)a.B * gd.tilesX * gd.tilesY;
    const long long dmax = gd.tilesX > gd.tilesY ? gd.tilesX : gd.tilesY;
    if (grid * dmax >= (1ll << 32)) { cp_set_error("head_wino24: grid %lld too large", grid); return 1; }
    hipLaunchKernelGGL((head_wino24_kernel<N2, MM>), dim3((unsigned)grid), dim3(HW_THREADS), smem, s, a, gd, hd);
    cp_note_kernel("head_wino24_kernel<%d, %d>", N2, MM);
    return 0;
}

// a.w = F(2x4) Winograd-domain weights (cp_winograd24_pack_f32) of the 3x3 conv.  Returns -1 when the shape is not this kernel's
// (64 input channels, mid channels a multiple of 32, ReLU, n2 <= 34, NHWC, 16-byte aligned operands).
int cp_launch_head3x3_1x1_w24(const ConvArgs& a, const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, hipStream_t s)
{
    const bool ok = a.nsrc == 1 && a.kh == 3 && a.kw == 3 && a.sy == 1 && a.sx == 1 && a.py == 1 && a.px == 1 && a.Ho == a.H &&
                    a.Wo == a.W && a.srcC[0] == 64 && a.srcLd[0] % 4 == 0 && a.Cout % 32 == 0 && a.act == CP_ACT_RELU && !a.res &&
                    n2 >= 1 && n2 <= 34 && ld2 % 4 == 0 && ld2 >= a.Cout &&
                    (((size_t)a.src[0] | (size_t)a.w | (size_t)w2 | (size_t)a.scale | (size_t)a.shift) & 15) == 0 &&
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31);
    if (!ok) return -1;
    HwHead hd;
    hd.w2 = w2; hd.b2 = b2; hd.out2 = out2; hd.n2 = n2; hd.ld2 = ld2; hd.act2 = act2;
    if (n2 == 1) return launch_head_wino24<1, 0>(a, s, hd);
    if (n2 == 2) return launch_head_wino24<2, 0>(a, s, hd);
    if (n2 <= 32) return launch_head_wino24<0, 1>(a, s, hd);
    if (n2 == 33) return launch_head_wino24<1, 1>(a, s, hd);
    return launch_head_wino24<2, 1>(a, s, hd);
}
