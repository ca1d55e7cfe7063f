// 3x3 / stride 1 / pad 1 convolution as a fused Winograd F(2x2,3x3) on the fp32 matrix cores (gfx950).
//
// 3x3/s1 layers are ~75 % of the network's FLOPs (DLA base blocks, DCN offset convs, the six head
// convs; pose_dla_dcn.py:29-57, keypoint.py:14-42).  F(2x2,3x3) computes a 2x2 output tile from a 4x4
// input tile with 16 multiplies per (cin,cout) instead of 36:
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A         (Lavin & Gray; B, G, A have entries 0, +-1, +-1/2)
// i.e. 16 independent GEMMs  M[xi][nu][tile][n] = sum_c V[xi][nu][tile][c] * U[xi][nu][n][c].
// Everything is fused in one kernel; nothing of the Winograd domain touches HBM:
//   * block = 8*MT x 16 output pixels = MT sets of 4x8 Winograd tiles (one 32-wide MFMA tile each), times 32*NT output
//     channels; (MT,NT) = (1,2) is the general shape, (1,1) for a single channel tile;
//   * wave xi (0..3) owns transform row xi: the four frequencies (xi, nu=0..3).  Row xi of B^T has two
//     non-zeros, so the wave needs only two of the four rows of every 4x4 input tile;
//   * the raw (8+2)x(16+2) input patch is staged in LDS 16 channels at a time (double buffered, one
//     barrier per stage).  A lane (tile m, k-half h) reads 8 ds_read_b128 (2 rows x 4 cols x 4 channels),
//     forms V with 16 packed-fp32 ops and feeds it STRAIGHT into the MFMA operand registers - V never exists in memory;
//   * U (transformed weights, cp_winograd_pack_f32) is wave-private (each wave has its own frequencies),
//     so its fragments go global -> registers as coalesced 1 KiB wave loads, prefetched one 8-channel chunk ahead
//     and issued from INSIDE the MFMA block (in front of it they cost up to 20 % of the matrix rate); no LDS, no barrier;
//   * per 8-channel chunk and wave: 8 ds_read_b128 + 16 packed VALU + 4*NT global loads feed 16*NT
//     v_mfma_f32_32x32x2f32 (64 cycles each).  Measured on MI355X (tools/micro/wino_loop.hip): every VALU
//     instruction costs ~4 cycles of matrix-pipe time (no co-issue, same or other wave), so the transform and
//     the epilogue use v_pk_{add,fma}_f32 and block-index math is scalar (host-side magic division);
//   * the MFMA is issued as D[channel][tile] (A = U fragment, B = V fragment) so a lane ends up with 4 consecutive
//     channels; epilogue: the nu-sum of A is done in registers, the xi-sum across the four waves through LDS
//     (ds_write_b128), then scale/shift (folded BN) + residual + activation and float4 NHWC stores.
// Arithmetic is fp32 throughout; the result differs from the direct convolution by ordinary fp32
// rounding (~1e-6 relative, measured in tests/test_conv_hip.py), far inside the path's 1e-3 bar.
#include "igemm.h"

#define WG_TW 16
#define WG_PW (WG_TW + 2)
#define WG_PWP 20                                  // padded patch row, in float4 slots (even: see the swizzle)
#define WG_LDR 36                                  // reduction-buffer row pitch (floats)
#define WG_RED (8 * 32 * WG_LDR)                   // [xi][b][tile][n] floats

typedef float wg_v2 __attribute__((ext_vector_type(2)));
typedef float wg_v4 __attribute__((ext_vector_type(4)));

template <int MT, int KS> struct WgGeo {
    static constexpr int TH = 8 * MT, PH = TH + 2;
    static constexpr int CGS = KS / 4;                   // 4-channel groups per stage
    static constexpr int CG = PH * WG_PWP * 4;          // floats per 4-channel group plane
    static constexpr int STAGE = CGS * CG;               // floats per stage buffer
    static constexpr int F4 = PH * WG_PW * CGS;          // float4 per stage
    static constexpr int SLOTS = (F4 + IG_THREADS - 1) / IG_THREADS;
    static constexpr int MAIN_FLOATS = 2 * STAGE > WG_RED ? 2 * STAGE : WG_RED;
    static constexpr int SMEM_FLOATS = MAIN_FLOATS + 16;      // + 64 bytes of zeros: the accumulators are initialised by LDS reads
};

// launch geometry; block-index decomposition uses host-made magic numbers: q = mulhi(n, ceil(2^32/d)) is exact
// for n*d < 2^32 (checked on the host), and stays on the scalar unit
struct WgGrid {
    int tilesX, tilesY, ntb;
    unsigned mNtb, mTx, mTy;
};
__device__ __forceinline__ int wg_div(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }

__device__ __forceinline__ wg_v4 wg_lds4(const float* p) { return *reinterpret_cast<const wg_v4*>(p); }

// One accumulator group (32 Winograd tiles x 32 channels, the wave's four frequencies) -> output pixels.
// Y = A^T M A: the nu-sum is done in registers, the xi-sum across the four waves through `red` (the caller
// guarantees nobody still uses that LDS region), then scale/shift (+residual) + activation, float4 NHWC stores.
__device__ __forceinline__ void wg_output_tile(const ConvArgs& a, float* red, const f32x16 (&acc)[4], int xi, int h, int m, int tid,
                                               int b, int yb, int x0, int tile, bool store)
{
    const float* const a_res = a.res;
    const bool relu = a.act == CP_ACT_RELU, sigm = a.act == CP_ACT_SIGMOID;
    const bool vec_ok = (a.outLd & 3) == 0 && (((size_t)a.out) & 15) == 0 &&
                        (!a_res || ((a.resLd & 3) == 0 && (((size_t)a_res) & 15) == 0));
    // The MFMA is issued as D[channel][tile] (A = U fragment, B = V fragment), so a lane holds 4 x 4 consecutive
    // channels (8j + 4h .. +3) of tile m: the reduction buffer red[(xi*2+bcol)*32 + tile][channel] is written with
    // ds_write_b128 (pitch 36 floats: the 16 tiles of a b128 lane group land on 16 distinct 4-bank groups).
    float* wp = red + (xi * 64 + m) * WG_LDR + 4 * h;
    // four accumulator elements at a time (no 2 x 16-register temporaries next to the 64 accumulator and 128 V registers)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wg_v4 q[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) q[nu] = (wg_v4){acc[nu][4 * j], acc[nu][4 * j + 1], acc[nu][4 * j + 2], acc[nu][4 * j + 3]};
        *reinterpret_cast<wg_v4*>(wp + 8 * j) = (q[0] + q[1]) + q[2];
        *reinterpret_cast<wg_v4*>(wp + 32 * WG_LDR + 8 * j) = (q[1] - q[2]) - q[3];
    }
    // per-thread output items (tile mi, column bb, 4 channels n4); their scale / shift / residual loads are issued
    // BEFORE the barrier so that the global-memory latency overlaps the cross-wave hand-over
    int itn[2], itox[2], itoy[2], itrd[2];
    bool itok[2], itvec[2];
    size_t itpix[2];
    wg_v4 sc[2], sh[2], rr[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * IG_THREADS;
        const int n4 = item & 7, bb = (item >> 3) & 1, mi = item >> 4;
        itn[it] = tile * 32 + n4 * 4;
        itox[it] = x0 + 2 * (mi & 7) + bb;
        itoy[it] = yb + 2 * (mi >> 3);
        itrd[it] = (bb * 32 + mi) * WG_LDR + n4 * 4;
        itok[it] = store && itox[it] < a.W && itn[it] < a.Cout && itoy[it] < a.H;
        itvec[it] = itok[it] && vec_ok && itn[it] + 3 < a.Cout;
        itpix[it] = ((size_t)b * a.H + itoy[it]) * a.W + itox[it];
        if (itvec[it]) {
            sc[it] = *reinterpret_cast<const wg_v4*>(a.scale + itn[it]);
            sh[it] = *reinterpret_cast<const wg_v4*>(a.shift + itn[it]);
            if (a_res) {
                rr[it][0] = *reinterpret_cast<const wg_v4*>(a_res + itpix[it] * a.resLd + itn[it]);
                if (itoy[it] + 1 < a.H) rr[it][1] = *reinterpret_cast<const wg_v4*>(a_res + (itpix[it] + a.W) * a.resLd + itn[it]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (!itok[it]) continue;
        const float* rp = red + itrd[it];
        const wg_v4 q0 = wg_lds4(rp), q1 = wg_lds4(rp + 64 * WG_LDR), q2 = wg_lds4(rp + 128 * WG_LDR), q3 = wg_lds4(rp + 192 * WG_LDR);
        wg_v4 yv[2];
        yv[0] = (q0 + q1) + q2;
        yv[1] = (q1 - q2) - q3;
        const int n = itn[it];
        if (itvec[it]) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                if (itoy[it] + aa >= a.H) continue;
                const size_t opix = itpix[it] + (size_t)aa * a.W;
                wg_v4 v = __builtin_elementwise_fma(yv[aa], sc[it], sh[it]);
                if (a_res) v += rr[it][aa];
                if (relu) v = (wg_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};
                else if (sigm) {
                    v.x = 1.0f / (1.0f + __expf(-v.x)); v.y = 1.0f / (1.0f + __expf(-v.y));
                    v.z = 1.0f / (1.0f + __expf(-v.z)); v.w = 1.0f / (1.0f + __expf(-v.w));
                } else if (a.act != CP_ACT_NONE) { v.x = cp_act(v.x, a.act); v.y = cp_act(v.y, a.act); v.z = cp_act(v.z, a.act); v.w = cp_act(v.w, a.act); }
                *reinterpret_cast<wg_v4*>(a.out + opix * a.outLd + n) = v;
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

// ---- fused head: [3x3 conv + bias + ReLU] -> 1x1 conv (+ bias, + sigmoid) without the intermediate ever leaving the CU ----
// lib/models/heads/keypoint.py:14-37: every branch is conv3x3(C -> head_conv) -> ReLU -> conv1x1(head_conv -> n).  The 268 MB mid
// tensor of a B = 16 head (written by the 3x3 launch, read back by the 1x1 launch: 2 x 268 MB per head) never exists:
// * n <= 2 outputs (hm, wh, reg, hp_offset): the 1x1 is applied on the channel tile that has just been reduced: each thread
//   multiplies its 4 channels of 4 output pixels with the matching 1x1 weights and keeps 4 * N2 partial sums in registers across
//   the block's channel tiles; at the end the 8 lanes that share a pixel are summed with three xor-shuffles (fixed order:
//   deterministic) and lane 0 writes the reference's NCHW output.
// * more outputs (hps 34, hm_hp 17; round 3, MM = 1): the ReLU'd 128-pixel x 32-channel tile goes to LDS (`mid`, 18 KB) and the 1x1
//   over it runs as a second MFMA phase, D[out j][pixel] += w2[j][c] * mid[pixel][c]: per channel tile and wave 16
//   v_mfma_f32_32x32x2f32 (+12.5 % on the 128 of the 3x3) into ONE 32x32 accumulator that lives across the block's channel
//   tiles.  Outputs 0..31 come from the MFMA phase; outputs 32, 33 (hps) ride on the n <= 2 register path above.  One
//   reduction buffer + `mid` (55 KB, two blocks per CU) instead of two reduction buffers, two barriers per channel tile.
struct WgHead {
    const float* w2;     // [N2][ld2] 1x1 weights, row-major over the mid channels
    const float* b2;     // [N2]
    float* out2;         // NCHW [B][n2][H][W]
    int n2, ld2, act2;
};

#define WGH_LDM 36                                  // `mid` row pitch in floats: 16-lane b128 groups on distinct 4-bank groups
#define WGH_MID (128 * WGH_LDM)                      // [pixel 128][32 channels] floats

template <int N2, int MM>
__device__ __forceinline__ void wg_output_tile_head(const ConvArgs& a, const WgHead& hd, float* red, const f32x16 (&acc)[4], int xi, int h,
                                                    int m, int tid, int yb, int x0, int tile, float (&acc2)[2][2][N2 > 0 ? N2 : 1],
                                                    float* mid)
{
    constexpr int J0 = MM ? 32 : 0;                  // first output channel of the register (VALU) path
    float* wp = red + (xi * 64 + m) * WG_LDR + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wg_v4 q[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) q[nu] = (wg_v4){acc[nu][4 * j], acc[nu][4 * j + 1], acc[nu][4 * j + 2], acc[nu][4 * j + 3]};
        *reinterpret_cast<wg_v4*>(wp + 8 * j) = (q[0] + q[1]) + q[2];
        *reinterpret_cast<wg_v4*>(wp + 32 * WG_LDR + 8 * j) = (q[1] - q[2]) - q[3];
    }
    int itn[2], itrd[2], itmid[2];
    wg_v4 sc[2], sh[2], w2r[2][N2 > 0 ? N2 : 1];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * IG_THREADS;
        const int n4 = item & 7, bb = (item >> 3) & 1, mi = item >> 4;
        itn[it] = tile * 32 + n4 * 4;
        itrd[it] = (bb * 32 + mi) * WG_LDR + n4 * 4;
        itmid[it] = (32 * (mi >> 3) + 2 * (mi & 7) + bb) * WGH_LDM + n4 * 4;      // pixel (row 2*(mi>>3), column 2*(mi&7)+bb) of the 8x16 tile, row-major
        sc[it] = *reinterpret_cast<const wg_v4*>(a.scale + itn[it]);
        sh[it] = *reinterpret_cast<const wg_v4*>(a.shift + itn[it]);
#pragma unroll
        for (int j = 0; j < N2; ++j) w2r[it][j] = *reinterpret_cast<const wg_v4*>(hd.w2 + (size_t)(J0 + j) * hd.ld2 + itn[it]);
    }
    // MM: this wave's 1x1 weight fragments for the MFMA phase, straight from the row-major w2: lane (j = lane % 32, g = lane / 32)
    // takes channels tile*32 + 8s + 4g .. +3 of output row j (rows >= n2 read row n2 - 1 and are zeroed) -- the four loads of a lane
    // cover one 128-byte line; issued here so that they land under the barrier and the reduction
    wg_v4 w2f[MM ? 4 : 1];
    if constexpr (MM) {
        const int nmm = hd.n2 < 32 ? hd.n2 : 32;
        const bool live = m < nmm;
        const float* wrow = hd.w2 + (size_t)(live ? m : nmm - 1) * hd.ld2 + tile * 32 + 4 * h;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 t4 = ig_ldg4(wrow + 8 * s4);
            w2f[s4] = live ? (wg_v4){t4.x, t4.y, t4.z, t4.w} : (wg_v4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const float* rp = red + itrd[it];
        const wg_v4 q0 = wg_lds4(rp), q1 = wg_lds4(rp + 64 * WG_LDR), q2 = wg_lds4(rp + 128 * WG_LDR), q3 = wg_lds4(rp + 192 * WG_LDR);
        wg_v4 yv[2];
        yv[0] = (q0 + q1) + q2;
        yv[1] = (q1 - q2) - q3;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            wg_v4 v = __builtin_elementwise_fma(yv[aa], sc[it], sh[it]);
            v = (wg_v4){cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)};  // the head's ReLU (keypoint.py:17,21,...)
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                const wg_v4 w = w2r[it][j];
                acc2[it][aa][j] = fmaf(v.w, w.w, fmaf(v.z, w.z, fmaf(v.y, w.y, fmaf(v.x, w.x, acc2[it][aa][j]))));
            }
            if constexpr (MM) *reinterpret_cast<wg_v4*>(mid + itmid[it] + aa * 16 * WGH_LDM) = v;      // row oy + aa: 16 pixels further
        }
    }
    if constexpr (MM) {
        // second MFMA phase: wave xi owns pixels 32*xi .. +31 of the tile (two rows of 16); D[out j][pixel].  The 32x32
        // accumulator lives across the block's channel tiles in a thread-private LDS slot (`park`, [4][256 threads] float4 behind
        // `mid`): next to V (128) + the 3x3 accumulators (64) + two U sets (32) sixteen more live registers spill to scratch
        float* pk = mid + WGH_MID + tid * 4;
        f32x16 acc2m;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const wg_v4 t4 = wg_lds4(pk + r4 * 4 * IG_THREADS);
            acc2m[4 * r4] = t4.x; acc2m[4 * r4 + 1] = t4.y; acc2m[4 * r4 + 2] = t4.z; acc2m[4 * r4 + 3] = t4.w;
        }
        __syncthreads();
        const float* mp = mid + (32 * xi + m) * WGH_LDM + 4 * h;
        wg_v4 mf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) mf[s4] = wg_lds4(mp + 8 * s4);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].x, mf[s4].x, acc2m, 0, 0, 0);
            acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].y, mf[s4].y, acc2m, 0, 0, 0);
            acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].z, mf[s4].z, acc2m, 0, 0, 0);
            acc2m = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[s4].w, mf[s4].w, acc2m, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<wg_v4*>(pk + r4 * 4 * IG_THREADS) = (wg_v4){acc2m[4 * r4], acc2m[4 * r4 + 1], acc2m[4 * r4 + 2], acc2m[4 * r4 + 3]};
    }
}

// MT = 32-tile M sets per block (block = 8*MT x 16 output pixels), NT = 32-channel N tiles per block,
// KS = channels per LDS stage, NB = U-fragment register sets (prefetch distance NB-1 chunks).
// The U fragment of a chunk is used by MT MFMAs, the V fragment by NT: U traffic per flop ~ 1/MT,
// LDS reads + transform VALU per flop ~ 1/NT.
template <int MT, int NT, int KS, int NB>
__global__ __launch_bounds__(IG_THREADS, (MT * NT == 1 && NB == 2) ? 3 : (MT * NT >= 4 ? 1 : 2)) void conv3x3_wino_kernel(const ConvArgs a, const WgGrid gd)
{
    using Geo = WgGeo<MT, KS>;
    constexpr int CPS = KS / 8, U = CPS * MT;          // chunks / units per stage
    static_assert(CPS % NB == 0 && NB >= 2, "static U-fragment register rotation");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int NTILES = (a.Cout + 31) >> 5;             // 32-channel tiles in the packed U
    int t_ = ig_xcd_remap(blockIdx.x, gridDim.x), q_;
    // split-C (a.ksplit = S > 1; small maps: a 512-channel 16x16 layer is 32 blocks of 32 stages on 256 CUs): block (tile, split)
    // accumulates the stages [sp*n/S, (sp+1)*n/S) and stores RAW partial outputs to out + sp*M*outLd; cp_splitk_reduce_f32 sums the
    // splits in a fixed order and applies scale / shift / activation (the output transform is linear, so partial sums add)
    const int S = a.ksplit, sp = S > 1 ? t_ % S : 0;
    if (S > 1) t_ /= S;
    q_ = wg_div(t_, gd.ntb, gd.mNtb); const int nb = t_ - q_ * gd.ntb; t_ = q_;
    q_ = wg_div(t_, gd.tilesX, gd.mTx); const int tx = t_ - q_ * gd.tilesX; t_ = q_;
    q_ = wg_div(t_, gd.tilesY, gd.mTy); const int ty = t_ - q_ * gd.tilesY;
    const int b = q_;
    const int y0 = ty * Geo::TH, x0 = tx * WG_TW;
    const int C = a.srcC[0], ld = a.srcLd[0];
    const int KC = C >> 3;                             // 8-channel chunks
    const float* __restrict__ x = a.src[0];
    // whole halo patch inside the image: no zero-select at LDS-store time (block-uniform)
    const bool interior = y0 >= 1 && y0 + Geo::TH + 1 <= a.H && x0 >= 1 && x0 + WG_TW + 1 <= a.W;

    // ---- staging slots (fixed per thread): patch pixel pp, channel group q.  LDS layout [q][py][px ^ f(py)][4],
    // f(py) = (py >> 1) & 1: the 16 lanes of a ds_read_b128 group are 8 tile columns (float4 slots 2*txx + j,
    // all of one parity) x 2 tile rows (2 patch rows apart = 40 slots, even); flipping the slot parity on
    // every other tile row makes the 16 slots distinct mod 16 -> conflict-free fragment reads.
    int go[Geo::SLOTS], lo[Geo::SLOTS];
    bool ok[Geo::SLOTS];
#pragma unroll
    for (int s = 0; s < Geo::SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        const int pp = idx / Geo::CGS, q = idx % Geo::CGS;
        const int py = pp / WG_PW, px = pp - py * WG_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inl = idx < Geo::F4;
        ok[s] = inl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        go[s] = ok[s] ? ((b * a.H + gy) * a.W + gx) * ld + q * 4 : 0;
        lo[s] = inl ? ((q * Geo::PH + py) * WG_PWP + (px ^ ((py >> 1) & 1))) * 4 : -1;
    }
    float4 rg[Geo::SLOTS];
    auto stage_load = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < Geo::SLOTS; ++s) rg[s] = ig_ldg4(x + go[s] + c0);     // raw; zeroed at store time
    };
    auto stage_store = [&](float* buf) __attribute__((always_inline)) {
        if (interior) {
#pragma unroll
            for (int s = 0; s < Geo::SLOTS; ++s)
                if (lo[s] >= 0) *reinterpret_cast<float4*>(buf + lo[s]) = rg[s];
        } else {
#pragma unroll
            for (int s = 0; s < Geo::SLOTS; ++s)
                if (lo[s] >= 0) {
                    float4 v = rg[s];
                    if (!ok[s]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(buf + lo[s]) = v;
                }
        }
    };

    // ---- wave's transform row: t[j] = d[rA][j] + sg * d[rB][j]
    const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rB = xi == 2 ? 1 : (xi == 3 ? 3 : 2);
    const float sg1 = xi == 1 ? 1.f : -1.f;
    const wg_v2 sg = {sg1, sg1};
    const int tyy = m >> 3, txx = m & 7;
    const int fA = (tyy + (rA >> 1)) & 1, fB = (tyy + (rB >> 1)) & 1;      // slot-parity flip of the lane's rows
    const int baseA = ((2 * tyy + rA) * WG_PWP + 2 * txx) * 4 + h * Geo::CG;
    const int baseB = ((2 * tyy + rB) * WG_PWP + 2 * txx) * 4 + h * Geo::CG;
    // column j lives in slot 2*txx + (j ^ f): even j -> +f, odd j -> -f
    const int offAe = baseA + fA * 4, offAo = baseA - fA * 4, offBe = baseB + fB * 4, offBo = baseB - fB * 4;

    // ---- U fragments: [xi][ntile][kc][nu][lane][4]
    const float* ub[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int tile = nb * NT + nt;
        if (tile >= NTILES) tile = NTILES - 1;          // ragged last block: compute a duplicate, never stored
        ub[nt] = a.w + ((size_t)(xi * NTILES + tile) * KC) * 1024 + lane * 4;
    }
    float4 bq[NB][NT][4];
    auto load_u = [&](int kc, float4 (&dst)[NT][4]) __attribute__((always_inline)) {
        const int kk = kc < KC ? kc : KC - 1;              // clamped: prefetches past the end re-read the last chunk
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) dst[nt][nu] = ig_ldg4(ub[nt] + (size_t)kk * 1024 + nu * 256);
    };

    const int nstage_all = C / KS;
    const int st0 = sp * nstage_all / S, nstage = (sp + 1) * nstage_all / S;      // this block's stages [st0, nstage)
    __builtin_assume(st0 < nstage);            // (the launcher checks ksplit <= C / KS) no second copy of the prologue for an empty loop
    stage_load(st0 * KS);
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) load_u(st0 * CPS + s, bq[s]);
    stage_store(smem + (st0 & 1) * Geo::STAGE);
    if (tid < 4) *reinterpret_cast<float4*>(smem + Geo::MAIN_FLOATS + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // 64 * MT * NT accumulator registers start at zero.  As v_mov_b32 that is 128 VALU instructions per wave (the compiler even emitted
    // them twice, once for the path around the loop), a quarter of what a 128-channel layer's main loop issues -- and every VALU
    // instruction takes issue cycles the matrix pipe cannot use.  Broadcast ds_read_b128 of a zeroed LDS line cost none.
    f32x16 acc[MT][NT][4];
    {
        const wg_v4* zp = reinterpret_cast<const wg_v4*>(smem + Geo::MAIN_FLOATS);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        wg_v4 z = zp[r4];
                        asm volatile("" : "+v"(z));          // four distinct reads: the compiler must not fold them into one + copies
                        acc[mt][nt][nu][4 * r4] = z.x; acc[mt][nt][nu][4 * r4 + 1] = z.y;
                        acc[mt][nt][nu][4 * r4 + 2] = z.z; acc[mt][nt][nu][4 * r4 + 3] = z.w;
                    }
    }

    wg_v4 dAp[4], dBp[4];                     // raw rows carried across the MFMAs (PIPE only)
    auto read_d = [&](const float* buf, int ch, int mt, wg_v4 (&dA)[4], wg_v4 (&dB)[4]) __attribute__((always_inline)) {
        const float* pc = buf + ch * 2 * Geo::CG + mt * (8 * WG_PWP * 4);       // tile set mt: 4 tile rows down
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dA[j] = wg_lds4(pc + ((j & 1) ? offAo : offAe) + j * 4);
            dB[j] = wg_lds4(pc + ((j & 1) ? offBo : offBe) + j * 4);
        }
    };
    constexpr bool PIPE = MT * NT >= 4;       // one wave per SIMD: nobody else hides the LDS round trip
    constexpr bool STORE_IN_BLOCK = U >= 2;   // patch hand-over (ds_write) from inside the last unit's MFMA block
#pragma unroll 1
    for (int st = st0; st < nstage; ++st) {
        const float* buf = smem + (st & 1) * Geo::STAGE;
        const bool more = st + 1 < nstage;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = u / MT, mt = u % MT;
            // Global loads are issued from INSIDE the unit's MFMA block (after 4 resp. 12 of its MFMAs), never in front
            // of it: in front they cost ~20 % of the matrix rate at two waves per SIMD (tools/micro/vs_loop.hip).
            // vmcnt retires in order: the next stage's patch loads (HBM latency) must be YOUNGER than every U
            // fragment load that is consumed inside this stage, or each such wait would also wait for the patch.
            // V = B^T d B for the wave's row xi: packed fp32 math, 16 VALU instructions per unit.  The raw rows of
            // unit u+1 (same stage buffer) are read right after unit u's transform has consumed the registers, i.e.
            // BEFORE unit u's MFMAs: the LDS latency hides behind them (matters at one wave per SIMD).
            wg_v2 tl[4], th[4];
            if constexpr (PIPE) {
                if (u == 0) read_d(buf, 0, 0, dAp, dBp);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tl[j] = __builtin_elementwise_fma(sg, dBp[j].xy, dAp[j].xy);
                    th[j] = __builtin_elementwise_fma(sg, dBp[j].zw, dAp[j].zw);
                }
                if (u + 1 < U) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_d(buf, (u + 1) / MT, (u + 1) % MT, dAp, dBp);
                }
            } else {
                const float* pc = buf + ch * 2 * Geo::CG + mt * (8 * WG_PWP * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const wg_v4 da = wg_lds4(pc + ((j & 1) ? offAo : offAe) + j * 4);
                    const wg_v4 db = wg_lds4(pc + ((j & 1) ? offBo : offBe) + j * 4);
                    tl[j] = __builtin_elementwise_fma(sg, db.xy, da.xy);
                    th[j] = __builtin_elementwise_fma(sg, db.zw, da.zw);
                }
            }
            wg_v2 vl[4], vh[4];
            vl[0] = tl[0] - tl[2]; vh[0] = th[0] - th[2];
            vl[1] = tl[1] + tl[2]; vh[1] = th[1] + th[2];
            vl[2] = tl[2] - tl[1]; vh[2] = th[2] - th[1];
            vl[3] = tl[1] - tl[3]; vh[3] = th[1] - th[3];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                if (nu == 1 && mt == 0) {
                    load_u(st * CPS + ch + NB - 1, bq[(ch + NB - 1) % NB]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nu == 3 && u == (NB - 2) * MT && more) {
                    stage_load((st + 1) * KS);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (STORE_IN_BLOCK && nu == 2 && u == U - 1 && more) {      // the other stage buffer: nobody reads it now
                    stage_store(smem + ((st + 1) & 1) * Geo::STAGE);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 bb = bq[ch % NB][nt][nu];
                    acc[mt][nt][nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, vl[nu].x, acc[mt][nt][nu], 0, 0, 0);
                    acc[mt][nt][nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, vl[nu].y, acc[mt][nt][nu], 0, 0, 0);
                    acc[mt][nt][nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, vh[nu].x, acc[mt][nt][nu], 0, 0, 0);
                    acc[mt][nt][nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, vh[nu].y, acc[mt][nt][nu], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!STORE_IN_BLOCK && more) stage_store(smem + ((st + 1) & 1) * Geo::STAGE);
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A.  nu-sum in registers, xi-sum across waves through LDS.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (mt + nt) __syncthreads();
            if (S > 1) {
                ConvArgs e = a;
                e.out = a.out + (size_t)sp * ((size_t)a.B * a.H * a.W) * a.outLd;
                wg_output_tile(e, smem, acc[mt][nt], xi, h, m, tid, b, y0 + mt * 8, x0, nb * NT + nt, nb * NT + nt < NTILES);
            } else wg_output_tile(a, smem, acc[mt][nt], xi, h, m, tid, b, y0 + mt * 8, x0, nb * NT + nt, nb * NT + nt < NTILES);
        }
}

// ---- V-stationary variant for 64 input channels (the six head 3x3 convs, 64 -> 256 each, 28 % of DLA-34's FLOPs) ----
// With C = 64 the wave's whole transformed input V[4 nu][32 tiles][64 ch] is 128 VGPRs: it is formed ONCE per
// block (whole 64-channel halo patch staged in LDS in one go), then the block loops over its output-channel tiles
// with an inner loop that is nothing but U-fragment loads and MFMAs - no LDS reads, no transform VALU, no barrier,
// no patch re-staging per channel tile (the generic kernel re-stages the patch for each of the 8 tiles of a head).
// Every VALU instruction costs ~4 cycles of matrix-pipe time and global loads issued in front of an MFMA block up to 20 %
// of the matrix rate (tools/micro/wino_loop.hip, vs_loop.hip), so this removes most of the non-MFMA issue slots.
// LDS: 51 KB patch, re-used as two 37 KB reduction buffers.
#define WGV_C 64
#define WGV_LOAD_AT 2                                    // U loads after 8 of a chunk's 16 MFMAs
#define WGV_CGS (WGV_C / 4)
#define WGV_F4 (10 * WG_PW * WGV_CGS)                       // 2880 float4
#define WGV_SLOTS ((WGV_F4 + IG_THREADS - 1) / IG_THREADS)  // 12
#define WGV_CG (10 * WG_PWP * 4)                            // floats per 4-channel plane
#define WGV_SMEM_FLOATS (2 * WG_RED)                        // two reduction buffers (73.7 KB) >= the 51.2 KB patch
#define WGV_SMEM_FLOATS_MM (WG_RED + WGH_MID + 16 * IG_THREADS)   // MM: one reduction buffer + the mid tile + the parked 1x1 accumulators (71.7 KB)

// N2: output channels of the head's 1x1 on the register path (0: none); MM = 1: second MFMA phase for outputs 0..31 of the 1x1
// (then the register path handles outputs 32 .. 32 + N2 - 1).  <0, 0> = plain convolution.
template <int N2, int MM>
__global__ __launch_bounds__(IG_THREADS, 2) void conv3x3_wino_vs64_kernel(const ConvArgs a, const WgGrid gd, int NL, const WgHead hd)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int NTILES = (a.Cout + 31) >> 5;
    int t_ = ig_xcd_remap(blockIdx.x, gridDim.x), q_;
    q_ = wg_div(t_, gd.ntb, gd.mNtb); const int ng = t_ - q_ * gd.ntb; t_ = q_;
    q_ = wg_div(t_, gd.tilesX, gd.mTx); const int tx = t_ - q_ * gd.tilesX; t_ = q_;
    q_ = wg_div(t_, gd.tilesY, gd.mTy); const int ty = t_ - q_ * gd.tilesY;
    const int b = q_;
    const int y0 = ty * 8, x0 = tx * WG_TW;
    const int ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];
    const int nt0 = ng * NL, nt1 = nt0 + NL < NTILES ? nt0 + NL : NTILES;

    // ---- stage the whole 64-channel halo patch: [cg 16][py 10][px ^ f(py)][4]
    {
        float4 rg[WGV_SLOTS];
        int lo[WGV_SLOTS];
        bool ok[WGV_SLOTS];
#pragma unroll
        for (int s = 0; s < WGV_SLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            const int pp = idx / WGV_CGS, q = idx % WGV_CGS;
            const int py = pp / WG_PW, px = pp - py * WG_PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool inl = idx < WGV_F4;
            ok[s] = inl && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int go = ok[s] ? ((b * a.H + gy) * a.W + gx) * ld + q * 4 : 0;
            lo[s] = inl ? ((q * 10 + py) * WG_PWP + (px ^ ((py >> 1) & 1))) * 4 : -1;
            rg[s] = ig_ldg4(x + go);
        }
#pragma unroll
        for (int s = 0; s < WGV_SLOTS; ++s)
            if (lo[s] >= 0) {
                float4 v = rg[s];
                if (!ok[s]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(smem + lo[s]) = v;
            }
    }
    // first U chunk in flight under the transform
    const float* up = a.w + ((size_t)(xi * NTILES + nt0) * (WGV_C / 8)) * 1024 + lane * 4;
    const int nlin = (nt1 - nt0) * (WGV_C / 8);              // chunks this block walks through, contiguous in U
    float4 bq[2][4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) bq[0][nu] = ig_ldg4(up + nu * 256);
    __syncthreads();

    // ---- V = B^T d B for the wave's row xi, all 8 chunks, kept in registers
    const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rB = xi == 2 ? 1 : (xi == 3 ? 3 : 2);
    const float sg1 = xi == 1 ? 1.f : -1.f;
    const wg_v2 sg = {sg1, sg1};
    const int tyy = m >> 3, txx = m & 7;
    const int fA = (tyy + (rA >> 1)) & 1, fB = (tyy + (rB >> 1)) & 1;
    const int baseA = ((2 * tyy + rA) * WG_PWP + 2 * txx) * 4 + h * WGV_CG;
    const int baseB = ((2 * tyy + rB) * WG_PWP + 2 * txx) * 4 + h * WGV_CG;
    const int offAe = baseA + fA * 4, offAo = baseA - fA * 4, offBe = baseB + fB * 4, offBo = baseB - fB * 4;
    wg_v2 vl[WGV_C / 8][4], vh[WGV_C / 8][4];
    wg_v4 da[2][4], db[2][4];                               // raw patch rows, chunk kc + 1 in flight while chunk kc is transformed
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        da[0][j] = wg_lds4(smem + ((j & 1) ? offAo : offAe) + j * 4);
        db[0][j] = wg_lds4(smem + ((j & 1) ? offBo : offBe) + j * 4);
    }
#pragma unroll
    for (int kc = 0; kc < WGV_C / 8; ++kc) {
        if (kc + 1 < WGV_C / 8) {
            const float* pc = smem + (kc + 1) * 2 * WGV_CG;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                da[(kc + 1) & 1][j] = wg_lds4(pc + ((j & 1) ? offAo : offAe) + j * 4);
                db[(kc + 1) & 1][j] = wg_lds4(pc + ((j & 1) ? offBo : offBe) + j * 4);
            }
        }
        wg_v2 tl[4], th[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tl[j] = __builtin_elementwise_fma(sg, db[kc & 1][j].xy, da[kc & 1][j].xy);
            th[j] = __builtin_elementwise_fma(sg, db[kc & 1][j].zw, da[kc & 1][j].zw);
        }
        vl[kc][0] = tl[0] - tl[2]; vh[kc][0] = th[0] - th[2];
        vl[kc][1] = tl[1] + tl[2]; vh[kc][1] = th[1] + th[2];
        vl[kc][2] = tl[2] - tl[1]; vh[kc][2] = th[2] - th[1];
        vl[kc][3] = tl[1] - tl[3]; vh[kc][3] = th[1] - th[3];
        // pin the chunk's transform HERE: the results are first used after the barrier below, so the compiler sank all 128
        // transform instructions behind it and hoisted all 64 ds_read_b128 in front of it -- 256 VGPRs of raw patch in flight,
        // 24 of them spilled to scratch (96 B per lane = 50 MB of HBM writes per launch, PMC WRITE_SIZE)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(vl[kc][j]), "+v"(vh[kc][j]));
    }

    __syncthreads();                // every wave has read its rows of the patch: the region becomes the reduction buffers
    // ---- output-channel tiles: U streams through two register sets, V stays put
    int lin = 0;
    float acc2[2][2][N2 > 0 ? N2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < (N2 > 0 ? N2 : 1); ++j) acc2[i][k][j] = 0.f;
    if constexpr (MM) {                                    // parked 1x1 accumulators: thread-private slots beyond the patch region
        static_assert(WG_RED + WGH_MID >= WGV_CGS * WGV_CG, "park must not overlap the patch");
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) *reinterpret_cast<wg_v4*>(smem + WG_RED + WGH_MID + tid * 4 + r4 * 4 * IG_THREADS) = (wg_v4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int nt = nt0; nt < nt1; ++nt) {
        f32x16 acc[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < WGV_C / 8; ++kc) {
            ++lin;
            const float* un = up + (size_t)(lin < nlin ? lin : nlin - 1) * 1024;      // next chunk (next tile's first after the 8th)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                // the next chunk's U loads go out in the MIDDLE of this chunk's MFMA block: issued in front of it they
                // cost 20 % of the matrix rate at two waves per SIMD, after 8 of the 16 MFMAs nothing (tools/micro/vs_loop.hip)
                if (nu == WGV_LOAD_AT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[(kc + 1) & 1][q] = ig_ldg4(un + q * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float4 bb = bq[kc & 1][nu];
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, vl[kc][nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, vl[kc][nu].y, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, vh[kc][nu].x, acc[nu], 0, 0, 0);
                acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, vh[kc][nu].y, acc[nu], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // two reduction buffers: tile i+2 overwrites buffer i&1 only after the barrier inside tile i+1's output stage.
        // MM: ONE reduction buffer + `mid`; the second barrier of the output stage (between the mid writes and the MFMA phase) is
        // what separates this tile's reads of the reduction buffer from the next tile's writes, and the first barrier of the
        // next tile separates this tile's mid reads from the next tile's mid writes.
        if constexpr (MM) wg_output_tile_head<N2, 1>(a, hd, smem, acc, xi, h, m, tid, y0, x0, nt, acc2, smem + WG_RED);
        else if constexpr (N2 > 0) wg_output_tile_head<N2, 0>(a, hd, smem + ((nt - nt0) & 1) * WG_RED, acc, xi, h, m, tid, y0, x0, nt, acc2, nullptr);
        else wg_output_tile(a, smem + ((nt - nt0) & 1) * WG_RED, acc, xi, h, m, tid, b, y0, x0, nt, true);
    }
    if constexpr (MM) {
        // D[out j][pixel]: lane (pixel 32*xi + m of the 8x16 tile, row group h) holds outputs j = (r & 3) + 8 * (r >> 2) + 4 * h
        const size_t HW = (size_t)a.H * a.W;
        const int p = 32 * xi + m, ox = x0 + (p & 15), oy = y0 + (p >> 4);
        const int nmm = hd.n2 < 32 ? hd.n2 : 32;
        if (ox < a.W && oy < a.H) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const wg_v4 t4 = wg_lds4(smem + WG_RED + WGH_MID + tid * 4 + r4 * 4 * IG_THREADS);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = e + 8 * r4 + 4 * h;
                    if (j < nmm) hd.out2[((size_t)b * hd.n2 + j) * HW + (size_t)oy * a.W + ox] = cp_act(t4[e] + hd.b2[j], hd.act2);
                }
            }
        }
    }
    if constexpr (N2 > 0) {
        constexpr int J0 = MM ? 32 : 0;
        // the 8 lanes tid & 7 = 0..7 hold the partial sums of one (tile, column) over 32 channels each: sum them in a fixed
        // order, add the bias, apply the output activation and store the reference's NCHW planes
        const size_t HW = (size_t)a.H * a.W;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * IG_THREADS;
            const int bb = (item >> 3) & 1, mi = item >> 4;
            const int ox = x0 + 2 * (mi & 7) + bb, oy = y0 + 2 * (mi >> 3);
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int j = 0; j < N2; ++j) {
                    float v = acc2[it][aa][j];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    if ((tid & 7) == 0 && J0 + j < hd.n2 && ox < a.W && oy + aa < a.H)
                        hd.out2[((size_t)b * hd.n2 + J0 + j) * HW + (size_t)(oy + aa) * a.W + ox] = cp_act(v + hd.b2[J0 + j], hd.act2);
                }
        }
    }
}

static unsigned wg_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int MT, int NT, int KS, int NB>
static int launch_wino(const ConvArgs& a, hipStream_t s)
{
    auto kern = conv3x3_wino_kernel<MT, NT, KS, NB>;
    using Geo = WgGeo<MT, KS>;
    const int smem = Geo::SMEM_FLOATS * 4;
    static CpLdsGuard guard;           // once per (instantiation, device), on the first (warm-up) launch: not legal inside a stream capture
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem);
        if (e != hipSuccess) { cp_set_error("conv3x3_winograd: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    WgGrid gd;
    gd.tilesX = cp_cdiv(a.W, WG_TW); gd.tilesY = cp_cdiv(a.H, Geo::TH);
    const int ntiles = (a.Cout + 31) / 32;
    gd.ntb = (ntiles + NT - 1) / NT;
    gd.mNtb = wg_magic(gd.ntb); gd.mTx = wg_magic(gd.tilesX); gd.mTy = wg_magic(gd.tilesY);
    const long long grid = (long long)a.B * gd.tilesX * gd.tilesY * gd.ntb * a.ksplit;
    const long long dmax = gd.ntb > gd.tilesX ? (gd.ntb > gd.tilesY ? gd.ntb : gd.tilesY) : (gd.tilesX > gd.tilesY ? gd.tilesX : gd.tilesY);
    if (grid * dmax >= (1ll << 32)) { cp_set_error("conv3x3_winograd: grid %lld too large", grid); return 1; }
    if (a.ksplit > a.srcC[0] / KS) { cp_set_error("conv3x3_winograd: ksplit=%d exceeds the %d channel stages", a.ksplit, a.srcC[0] / KS); return 1; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, gd);
    cp_note_kernel("conv3x3_wino_kernel<%d, %d, %d, %d>", MT, NT, KS, NB);
    return 0;
}

template <int N2, int MM = 0>
static int launch_wino_vs64(const ConvArgs& a, hipStream_t s, int ngroups, const WgHead& hd)
{
    const int smem = (MM ? WGV_SMEM_FLOATS_MM : WGV_SMEM_FLOATS) * 4;
    static_assert(WGV_SMEM_FLOATS_MM >= WGV_CGS * WGV_CG, "the 64-channel patch must fit the MM layout");
    static CpLdsGuard guard;
    {
        const hipError_t e = guard.ensure((const void*)conv3x3_wino_vs64_kernel<N2, MM>, smem);
        if (e != hipSuccess) { cp_set_error("conv3x3_winograd: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    WgGrid gd;
    gd.tilesX = cp_cdiv(a.W, WG_TW); gd.tilesY = cp_cdiv(a.H, 8);
    const int ntiles = (a.Cout + 31) / 32;
    if (ngroups < 1) ngroups = 1;
    if (ngroups > ntiles) ngroups = ntiles;
    const int NL = (ntiles + ngroups - 1) / ngroups;
    gd.ntb = (ntiles + NL - 1) / NL;
    gd.mNtb = wg_magic(gd.ntb); gd.mTx = wg_magic(gd.tilesX); gd.mTy = wg_magic(gd.tilesY);
    const long long grid = (long long)a.B * gd.tilesX * gd.tilesY * gd.ntb;
    const long long dmax = gd.ntb > gd.tilesX ? (gd.ntb > gd.tilesY ? gd.ntb : gd.tilesY) : (gd.tilesX > gd.tilesY ? gd.tilesX : gd.tilesY);
    if (grid * dmax >= (1ll << 32)) { cp_set_error("conv3x3_winograd: grid %lld too large", grid); return 1; }
    hipLaunchKernelGGL((conv3x3_wino_vs64_kernel<N2, MM>), dim3((unsigned)grid), dim3(IG_THREADS), smem, s, a, gd, NL, hd);
    cp_note_kernel("conv3x3_wino_vs64_kernel<%d, %d>", N2, MM);
    return 0;
}

// [3x3 conv + bias + ReLU] + 1x1 conv of a KeypointHead branch with n2 <= 34 outputs in one launch.  Returns -1 when the shape is
// not the fused kernel's (64 input channels, mid channels a multiple of 32, NHWC, 16-byte aligned operands).
int cp_launch_head3x3_1x1(const ConvArgs& a, const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, hipStream_t s)
{
    const bool ok = a.nsrc == 1 && a.kh == 3 && a.kw == 3 && a.sy == 1 && a.sx == 1 && a.py == 1 && a.px == 1 && a.Ho == a.H &&
                    a.Wo == a.W && a.srcC[0] == 64 && a.srcLd[0] % 4 == 0 && a.Cout % 32 == 0 && a.act == CP_ACT_RELU && !a.res &&
                    n2 >= 1 && n2 <= 34 && ld2 % 4 == 0 && ld2 >= a.Cout &&
                    (((size_t)a.src[0] | (size_t)a.w | (size_t)w2 | (size_t)a.scale | (size_t)a.shift) & 15) == 0 &&
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31);
    if (!ok) return -1;
    WgHead hd;
    hd.w2 = w2; hd.b2 = b2; hd.out2 = out2; hd.n2 = n2; hd.ld2 = ld2; hd.act2 = act2;
    if (n2 == 1) return launch_wino_vs64<1>(a, s, 1, hd);
    if (n2 == 2) return launch_wino_vs64<2>(a, s, 1, hd);
    if (n2 <= 32) return launch_wino_vs64<0, 1>(a, s, 1, hd);          // MFMA phase for all outputs (hm_hp: 17)
    if (n2 == 33) return launch_wino_vs64<1, 1>(a, s, 1, hd);
    return launch_wino_vs64<2, 1>(a, s, 1, hd);                         // hps: 32 through the MFMA phase + 2 on the register path
}

// a.w = Winograd-domain weights from cp_winograd_pack_f32.  Returns -1 when the shape is not eligible.
int cp_launch_conv3x3_wino(const ConvArgs& a, hipStream_t s, int variant)
{
    const bool ok = a.nsrc == 1 && a.kh == 3 && a.kw == 3 && a.sy == 1 && a.sx == 1 && a.py == 1 && a.px == 1 &&
                    !a.outNCHW && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.Ho == a.H && a.Wo == a.W &&
                    a.OH == a.H && a.OW == a.W && a.srcC[0] % 16 == 0 && a.srcLd[0] % 4 == 0 &&
                    (((size_t)a.src[0] | (size_t)a.w) & 15) == 0 &&            // 16-byte vector loads of the patch and of U
                    (long long)a.B * a.H * a.W * a.srcLd[0] < (1ll << 31);
    if (!ok) return -1;
    if (variant == 24) return cp_launch_conv3x3_wino24(a, s);      // F(2x4,3x3): `a.w` is cp_winograd24_pack_f32's layout
    const int ntiles = (a.Cout + 31) / 32;
    // variant = MT*10 + NT (tuning / tests); 0 = auto.  Measured on MI355X (tools/bench_conv.py, B = 16): the 8x16-pixel x
    // 64-channel block (NT = 2: every V fragment feeds two MFMAs) is the best general shape on every DLA-34 / ResNet-50
    // layer; a single 32-channel tile (DCN offset convs) takes the 32-channel block.  (21 stays selectable for tests; the 512-VGPR
    // one-wave-per-SIMD shapes 22 / 41 measured 10-40 % slower and are no longer instantiated.)
    if (variant == 0) {
        const long long blocks_vs = (long long)a.B * cp_cdiv(a.H, 8) * cp_cdiv(a.W, WG_TW);
        // 64 input channels and >= 4 channel tiles (the head convs): V-stationary kernel, one block per spatial tile
        // (round 3, same-box A/B: for 64 -> 64 (two tiles) the V-stationary kernel runs 0.305 vs 0.300 ms over DLA-34's three launches,
        // for the 27-channel offset convs (one tile) 0.059 vs 0.051 ms: below four tiles the generic kernel stays)
        if (a.srcC[0] == 64 && ntiles >= 4 && blocks_vs >= 512 && a.ksplit == 1) variant = 6401;
        else {
            variant = ntiles == 1 ? 11 : 12;
            // a launch that cannot give every CU a block takes the 32-channel block: twice the blocks (the V transform is
            // done twice as often, on CUs that would otherwise idle).  hrnet B=8 1 254 -> 1 405 img/s (its 128-/256-channel
            // branches at 32x32 / 16x16 are 128 / 64 blocks of the 64-channel shape), dla_34 B=16 +0.9 %, B=1 +14 %.
            if (variant == 12 && blocks_vs * ((ntiles + 1) / 2) < 300) variant = 11;
        }
    }
    // 64xx: V-stationary kernel (C == 64), xx = number of channel-tile groups per spatial tile (0 -> 1)
    if (variant >= 6400 && variant < 6500) return (a.srcC[0] == 64 && a.ksplit == 1) ? launch_wino_vs64<0>(a, s, variant - 6400, WgHead{}) : -1;
    switch (variant) {
        case 11: return launch_wino<1, 1, 16, 2>(a, s);
        case 12: return launch_wino<1, 2, 16, 2>(a, s);
        case 21: return launch_wino<2, 1, 16, 2>(a, s);
        default: cp_set_error("conv3x3_winograd: unknown variant %d", variant); return 1;
    }
}

// ---- weight transform: packed direct weights [rows >= Cout][9*C] (k = (ky*3+kx)*C + c)  ->  U = G g G^T,
// laid out as the kernel's B fragments [xi][ntile][kc][nu][lane][4]:
//   value(lane, j) = U[xi][nu][n = ntile*32 + lane%32][c = kc*8 + (lane/32)*4 + j]      (zero for n >= Cout)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int C, int Cout, int ntiles, long long total)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int KC = C >> 3;
    long long t = idx;
    const int j = (int)(t & 3); t >>= 2;
    const int lane = (int)(t & 63); t >>= 6;
    const int nu = (int)(t & 3); t >>= 2;
    const int kc = (int)(t % KC); t /= KC;
    const int nt = (int)(t % ntiles);
    const int xi = (int)(t / ntiles);
    const int n = nt * 32 + (lane & 31), c = kc * 8 + (lane >> 5) * 4 + j;
    float r = 0.f;
    if (n < Cout) {
        const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        const float* g = w + (size_t)n * 9 * C + c;
        double accd = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) accd += G[xi][ky] * (double)g[(ky * 3 + kx) * C] * G[nu][kx];
        r = (float)accd;
    }
    u[idx] = r;
}

extern "C" size_t cp_winograd_weight_floats(int C, int Cout)
{
    if (C <= 0 || Cout <= 0 || C % 8) return 0;
    return (size_t)16 * ((Cout + 31) / 32) * 32 * C;
}

extern "C" int cp_winograd_pack_f32(const float* w, float* u, int C, int Cout, void* stream)
{
    CP_CHECK_ARG(w && u, "winograd_pack: null pointer");
    CP_CHECK_ARG(C > 0 && C % 16 == 0 && Cout > 0, "winograd_pack: C=%d must be a positive multiple of 16 (Cout=%d)", C, Cout);
    const int ntiles = (Cout + 31) / 32;
    const long long total = (long long)cp_winograd_weight_floats(C, Cout);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, u, C, Cout,
                       ntiles, total);
    CP_CHECK_LAUNCH("wino_pack_kernel");
    return 0;
}
