// Fused convolution kernels (implicit GEMM on fp32 MFMA) -- see igemm.h for the core.
//
// Replaces, for the inference hot path, what the reference gets from cuDNN + separate
// BatchNorm / ReLU / add / cat kernels:
//   nn.Conv2d + BatchNorm2d(eval) + ReLU     pose_dla_dcn.py:272-282, msra_resnet.py:82-102, ...
//   BasicBlock / Bottleneck residual add      pose_dla_dcn.py:43-57, msra_resnet.py:82-102
//   Root: torch.cat -> 1x1 conv -> BN -> ReLU pose_dla_dcn.py:155-163  (concat-free: up to 4 sources)
//   dense ConvTranspose2d(k4,s2,p1)           msra_resnet.py:168-193   (4 sub-pixel 2x2 convs)
//   KeypointHead convs (+bias, +sigmoid)      lib/models/heads/keypoint.py:14-42, multi_pose.py:35-37
// A-producers: NHWC im2col gather (any kh,kw,stride,pad) and a scalar gather for the 3-channel
// NCHW network input (7x7 / 3x3 stems).
#include <cstdlib>
#include "igemm.h"

// per-thread description of the output pixels whose A rows this thread stages
struct PixSlot {
    int boff;   // b*H*W (input pixel index of the image origin), -1 if m >= M
    int iy0, ix0;
};

// one block of the launch described by `a`: tile `tile` (already XCD-remapped) of `nblocks`
template <int BM, int BN, int WAVES_M, int WAVES_N, int MF, bool STEM>
__device__ __forceinline__ void igemm_conv_block(const ConvArgs& a, int tile, int nblocks, float* smem)
{
    using T = IgTile<BM, BN, WAVES_M, WAVES_N, MF>;
    float* As0 = smem;                       // [2][BM][IG_LDK]
    float* Bs0 = smem + 2 * T::A_FLOATS;     // [2][BN][IG_LDK]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // per-block view of the fields that differ between the sub-convolutions of a fused sub-pixel deconvolution (nsub = 4):
    // a 2048 -> 256 deconv at 16x16 is 4 x 128 blocks with 512 k-steps each - half the CUs idle per launch when run one by one
    const float* wsub = a.w;
    int pys = a.py, pxs = a.px, ooys = 0, ooxs = 0;
    if (a.nsub > 1) {
        const int per = nblocks / a.nsub, sub = tile / per;
        tile -= sub * per;
        wsub += (size_t)sub * a.ldw * a.K;
        pys -= sub >> 1; pxs -= sub & 1;
        ooys = sub >> 1; ooxs = sub & 1;
    }
    // split-K (a.ksplit = S > 1; single NHWC source, small-M layers: res_50 layer4's 512 -> 512 stride-2 3x3 at 16x16, B = 8, is 256
    // blocks of 288 k-steps on 256 CUs): block (tile, split) accumulates the k-steps [sp*nk/S, (sp+1)*nk/S) and stores RAW partial sums
    // to out + sp*M*outLd (scale = 1, shift = 0, no activation); cp_splitk_reduce_f32 adds the splits in a fixed order and finishes
    const int S = a.ksplit, ksp = S > 1 ? tile % S : 0;
    if (S > 1) tile /= S;
    const int NT = a.ldw / BN;
    const int nt = tile % NT, mt = tile / NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wid / WAVES_N) * T::WM, wn0 = (wid % WAVES_N) * T::WN;
    const int HoWo = a.Ho * a.Wo;

    typename IgAcc<MF>::type acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < IgAcc<MF>::N; ++r) acc[i][j][r] = 0.f;

    const int nk_all = a.K / IG_BK;
    const int ks0 = ksp * nk_all / S, nk = (ksp + 1) * nk_all / S;      // this block's k-steps [ks0, nk)
    float4 br[T::B_SLOTS];

    if constexpr (!STEM) {
        // ---------------- NHWC im2col producer: thread -> (pixel, 4-channel quad) -------------
        const int q = tid & 3;
        PixSlot ps[T::A_SLOTS];
#pragma unroll
        for (int s = 0; s < T::A_SLOTS; ++s) {
            const int m = m0 + (tid >> 2) + s * 64;
            if (m < a.M) {
                const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
                ps[s].boff = b * a.H * a.W;
                ps[s].iy0 = oy * a.sy - pys;
                ps[s].ix0 = ox * a.sx - pxs;
            } else { ps[s].boff = -1; ps[s].iy0 = 0; ps[s].ix0 = 0; }
        }
        float4 ar[T::A_SLOTS];
        bool aok[T::A_SLOTS];
        // k-walk state (wave-uniform): tap (ky,kx), source index, channel offset inside the source
        int ky = 0, kx = 0, si = 0, cl = 0;
        if (S > 1) {                                   // (single source) start the walk at k-step ks0
            const int per_tap = a.srcC[0] / IG_BK, tap = ks0 / per_tap;
            cl = (ks0 - tap * per_tap) * IG_BK;
            ky = tap / a.kw; kx = tap - ky * a.kw;
        }

        // current source (pointer / pixel stride / channels) lives in registers and is re-read from the
        // kernel arguments only when the k-walk crosses into the next concatenated source: indexing
        // a.src[si] in the loop costs 3 dependent scalar loads + s_waitcnt lgkmcnt(0) per k-step
        const float* sp = a.src[0];
        int ld = a.srcLd[0], sC = a.srcC[0];
        // Per-slot byte offset and validity depend on (tap, source) only: computed at the first k-step of each segment
        // (cl == 0) and reused by its C/16 k-steps; the channel offset rides in the scalar base address, so the steady
        // state issues its gathers with no address VALU (for 1x1 layers that is the whole loop).
        unsigned aoff[T::A_SLOTS];
        bool fresh = true;                             // the first k-step of a split may start inside a tap (cl != 0)
        auto load_a = [&]() __attribute__((always_inline)) {
            if (cl == 0 || fresh) {
                fresh = false;
#pragma unroll
                for (int s = 0; s < T::A_SLOTS; ++s) {
                    const int iy = ps[s].iy0 + ky, ix = ps[s].ix0 + kx;
                    const bool ok = ps[s].boff >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    // branch-free: out-of-range taps read the (always valid) tensor base and are zeroed at store time
                    aoff[s] = ok ? ((unsigned)(ps[s].boff + iy * a.W + ix) * (unsigned)ld + (unsigned)q * 4u) * 4u : 0u;
                    aok[s] = ok;
                }
            }
            const char* xs = reinterpret_cast<const char*>(sp) + (size_t)cl * 4;      // uniform
#pragma unroll
            for (int s = 0; s < T::A_SLOTS; ++s)
                ar[s] = ig_ldg4(reinterpret_cast<const float*>(xs + aoff[s]));      // raw; a select here would wait on vmcnt
        };
        auto advance = [&]() __attribute__((always_inline)) {
            cl += IG_BK;
            if (cl >= sC) {
                cl = 0;
                if (++si >= a.nsrc) { si = 0; if (++kx >= a.kw) { kx = 0; ++ky; } }
                if (a.nsrc > 1) { sp = a.src[si]; ld = a.srcLd[si]; sC = a.srcC[si]; }
            }
        };
        auto store_a = [&](float* As) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < T::A_SLOTS; ++s) {
                const int pl = (tid >> 2) + s * 64;
                *reinterpret_cast<float4*>(As + pl * IG_LDK + q * 4) = aok[s] ? ar[s] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };

        load_a(); advance();
        ig_load_b<T>(a, ks0 * IG_BK, n0, tid, br, wsub);
        store_a(As0);
        ig_store_b<T>(Bs0, tid, br);
        __syncthreads();
        int cur = 0;
        for (int ks = ks0; ks < nk; ++ks) {
            const bool more = ks + 1 < nk;
            ig_compute<T, MF>(As0 + cur * T::A_FLOATS, Bs0 + cur * T::B_FLOATS, wm0, wn0, lane, acc, [&]() __attribute__((always_inline)) {
                if (more) { load_a(); advance(); ig_load_b<T>(a, (ks + 1) * IG_BK, n0, tid, br, wsub); }
            });
            // nothing that consumes the prefetched registers may be scheduled above the MFMAs (the blend /
            // zero-select would drag an s_waitcnt vmcnt in front of them and expose the whole load latency)
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                store_a(As0 + (cur ^ 1) * T::A_FLOATS);
                ig_store_b<T>(Bs0 + (cur ^ 1) * T::B_FLOATS, tid, br);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // ---------------- stem producer: NCHW input, tiny C (3): k = (c*kh + ky)*kw + kx ------
        // thread -> one pixel column of the tile (consecutive lanes = consecutive x: coalesced),
        // 16*BM/256 scalar gathers per k-step, written straight into As[m][k].
        const float* sp = a.src[0];
        const int C = a.srcC[0], khw = a.kh * a.kw, Kreal = C * khw;
        // thread handles pixel pl = tid % BM and k rows kq, kq + (256/BM), ...
        const int pl = tid % BM, kq = tid / BM;
        constexpr int KSTRIDE = IG_THREADS / BM;           // 1 (BM=256), 2 (BM=128), 4 (BM=64)
        constexpr int NPER = IG_BK / KSTRIDE;
        const int m = m0 + pl;
        int boff = -1, iy0 = 0, ix0 = 0;
        if (m < a.M) {
            const int b = m / HoWo, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
            boff = b * C * a.H * a.W;
            iy0 = oy * a.sy - a.py;
            ix0 = ox * a.sx - a.px;
        }
        float ar[NPER];
        auto load_a = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < NPER; ++e) {
                const int k = k0 + kq + e * KSTRIDE;
                const int c = k / khw, t = k - c * khw, ky = t / a.kw, kx = t - ky * a.kw;
                const int iy = iy0 + ky, ix = ix0 + kx;
                const bool ok = boff >= 0 && k < Kreal && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                ar[e] = ok ? sp[(size_t)boff + ((size_t)c * a.H + iy) * a.W + ix] : 0.f;
            }
        };
        auto store_a = [&](float* As) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < NPER; ++e) As[pl * IG_LDK + kq + e * KSTRIDE] = ar[e];
        };
        load_a(0);
        ig_load_b<T>(a, 0, n0, tid, br);
        store_a(As0);
        ig_store_b<T>(Bs0, tid, br);
        __syncthreads();
        int cur = 0;
        for (int ks = 0; ks < nk; ++ks) {
            const bool more = ks + 1 < nk;
            if (more) { load_a((ks + 1) * IG_BK); ig_load_b<T>(a, (ks + 1) * IG_BK, n0, tid, br); }
            ig_compute<T, MF>(As0 + cur * T::A_FLOATS, Bs0 + cur * T::B_FLOATS, wm0, wn0, lane, acc);
            // nothing that consumes the prefetched registers may be scheduled above the MFMAs (the blend /
            // zero-select would drag an s_waitcnt vmcnt in front of them and expose the whole load latency)
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                store_a(As0 + (cur ^ 1) * T::A_FLOATS);
                ig_store_b<T>(Bs0 + (cur ^ 1) * T::B_FLOATS, tid, br);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    if (S > 1) {
        ConvArgs e = a;
        e.out = a.out + (size_t)ksp * a.M * a.outLd;
        ig_epilogue<T, BM, BN, MF>(e, smem, m0, n0, wm0, wn0, lane, tid, acc, ooys, ooxs);
    } else ig_epilogue<T, BM, BN, MF>(a, smem, m0, n0, wm0, wn0, lane, tid, acc, ooys, ooxs);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MF, bool STEM>
__global__ __launch_bounds__(IG_THREADS) void igemm_conv_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    igemm_conv_block<BM, BN, WAVES_M, WAVES_N, MF, STEM>(a, ig_xcd_remap(blockIdx.x, gridDim.x), gridDim.x, smem);
}

// Up to IG_GROUP_MAX INDEPENDENT single-source NHWC convolutions in ONE launch on the 64 x 64 tile: HRNet's fuse layers
// (pose_higher_hrnet.py:169-212: per module up to six 1x1 convs at 16x16 .. 64x64 and six stride-2 3x3 convs, 13-29 us each one by one
// on 16-256 blocks -- the chip is mostly idle, and the two capture streams cannot run more than two of them side by side).  Member k owns
// the blocks [first[k], first[k + 1]); members are ordered longest block (most k-steps) first so the short ones fill the tail.
#define IG_GROUP_MAX 8
struct IgMember {
    const float* src; const float* w; const float* scale; const float* shift; const float* res; float* out;
    int C, srcLd, B, H, W, Ho, Wo, kh, kw, sy, sx, py, px, K, ldw, resLd, outLd, Cout, act, pad_;
};
struct IgGroup {
    IgMember m[IG_GROUP_MAX];
    int first[IG_GROUP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(IG_THREADS) void igemm_conv_group_kernel(const IgGroup g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = blockIdx.x;
    int k = 0;
    while (k + 1 < g.n && t >= g.first[k + 1]) ++k;            // scalar
    const IgMember& mm = g.m[k];
    ConvArgs a;
    a.src[0] = mm.src; a.src[1] = a.src[2] = a.src[3] = nullptr;
    a.srcC[0] = mm.C; a.srcC[1] = a.srcC[2] = a.srcC[3] = 0;
    a.srcLd[0] = mm.srcLd; a.srcLd[1] = a.srcLd[2] = a.srcLd[3] = 0;
    a.nsrc = 1; a.Ctot = mm.C;
    a.B = mm.B; a.H = mm.H; a.W = mm.W; a.Ho = mm.Ho; a.Wo = mm.Wo; a.M = mm.B * mm.Ho * mm.Wo;
    a.kh = mm.kh; a.kw = mm.kw; a.sy = mm.sy; a.sx = mm.sx; a.py = mm.py; a.px = mm.px;
    a.K = mm.K; a.w = mm.w; a.ldw = mm.ldw; a.scale = mm.scale; a.shift = mm.shift; a.res = mm.res; a.resLd = mm.resLd;
    a.out = mm.out; a.outLd = mm.outLd; a.Cout = mm.Cout; a.outNCHW = 0; a.OH = mm.Ho; a.OW = mm.Wo;
    a.osy = a.osx = 1; a.ooy = a.oox = 0; a.act = mm.act;
    a.om = nullptr; a.omLd = 0; a.omMaskOff = 0; a.omSigmoid = 0; a.dily = a.dilx = 1; a.ksplit = 1; a.dg = 1; a.nsub = 1;
    // no XCD remap across members (it would hand each XCD one member's blocks, cf. conv3x3_wino24_group_kernel); inside a member
    // consecutive blocks share activation rows and go round-robin over the XCDs anyway
    igemm_conv_block<64, 64, 2, 2, 32, false>(a, t - g.first[k], g.first[k + 1] - g.first[k], smem);
}

// `occ` (tile code digit 8, tuning / experiments): at most that many blocks per CU, enforced by asking for 160 KiB / occ of dynamic
// LDS -- a launch whose blocks all fit on the chip at once runs its prologue, MFMA and epilogue phases in lock-step; fewer resident
// blocks stagger them (tools/igemm_ab.py)
template <int BM, int BN, int WAVES_M, int WAVES_N, int MF, bool STEM>
static int launch_conv(const ConvArgs& a, hipStream_t s, int occ = 0)
{
    using T = IgTile<BM, BN, WAVES_M, WAVES_N, MF>;
    auto kern = igemm_conv_kernel<BM, BN, WAVES_M, WAVES_N, MF, STEM>;
    if (a.ldw % BN != 0) { cp_set_error("conv2d: ldw=%d is not a multiple of the N tile %d", a.ldw, BN); return 1; }
    int smem = a.outNCHW ? T::SMEM : T::NHWC_BYTES;
    static CpLdsGuard guard;
    int smem_max = T::SMEM > T::NHWC_BYTES ? T::SMEM : T::NHWC_BYTES;
    if (occ > 0) {
        const int want = (160 * 1024 / occ) & ~255;
        if (want > smem) smem = want;
        if (want > smem_max) smem_max = want;
    }
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem_max);
        if (e != hipSuccess) { cp_set_error("conv2d: cannot reserve %d B LDS: %s", smem_max, hipGetErrorString(e)); return 2; }
    }
    if (a.ksplit > 1 && (STEM || a.nsub > 1 || a.nsrc != 1 || a.ksplit > a.K / IG_BK)) {
        cp_set_error("conv2d: ksplit=%d needs one NHWC source, nsub = 1 and at most K/16 = %d splits", a.ksplit, a.K / IG_BK);
        return 1;
    }
    const int grid = cp_cdiv(a.M, BM) * (a.ldw / BN) * (a.nsub > 1 ? a.nsub : 1) * a.ksplit;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(IG_THREADS), smem, s, a);
    cp_note_kernel("igemm_conv_kernel<%d, %d, %d, %d, %d, %s>", BM, BN, WAVES_M, WAVES_N, MF, STEM ? "true" : "false");
    return 0;
}

// Public descriptor (mirrors ConvArgs with plain ints; see include/centerpose_hip.h)
struct cp_conv_desc {
    int nsrc;
    int srcC[4];
    int srcLd[4];
    int B, H, W;
    int Ho, Wo;
    int kh, kw, sy, sx, py, px;
    int K, ldw;
    int Cout;
    int resLd;
    int outLd;
    int outNCHW;
    int OH, OW, osy, osx, ooy, oox;
    int act;
    int inNCHW;   // 1: src[0] is the NCHW network input with srcC[0] (<16) channels (stem path)
    int tile;     // 0 = auto; otherwise BM*1000+BN of a specific instantiation (tuning / tests)
    int nsub;     // 0 / 1: one conv; 4: fused sub-pixel deconvolution (see ConvArgs::nsub)
    int ksplit;   // cp_conv3x3_winograd_f32 only: S > 1 = split over the input channels, raw partial outputs to out[S][B*H*W][outLd]
};

extern "C" int cp_sizeof_conv_desc(void) { return (int)sizeof(cp_conv_desc); }

static int conv_args_from_desc(const cp_conv_desc* d, const float* const* src, const float* w, const float* scale,
                               const float* shift, const float* res, float* out, ConvArgs& a)
{
    CP_CHECK_ARG(d && src && w && scale && shift && out, "conv2d: null pointer");
    CP_CHECK_ARG(d->nsrc >= 1 && d->nsrc <= IG_MAX_SRC, "conv2d: nsrc=%d", d->nsrc);
    CP_CHECK_ARG(d->K % IG_BK == 0 && d->K > 0, "conv2d: K=%d must be a positive multiple of 16", d->K);
    CP_CHECK_ARG(d->ldw % 16 == 0 && d->ldw >= d->Cout, "conv2d: ldw=%d (Cout=%d)", d->ldw, d->Cout);
    int ctot = 0;
    for (int i = 0; i < IG_MAX_SRC; ++i) {
        a.src[i] = i < d->nsrc ? src[i] : nullptr;
        a.srcC[i] = i < d->nsrc ? d->srcC[i] : 0;
        a.srcLd[i] = i < d->nsrc ? d->srcLd[i] : 0;
        if (i < d->nsrc) {
            CP_CHECK_ARG(src[i] != nullptr, "conv2d: src[%d] is null", i);
            ctot += d->srcC[i];
            if (!d->inNCHW)
                CP_CHECK_ARG(d->srcC[i] % 16 == 0 && d->srcLd[i] % 4 == 0 && d->srcLd[i] >= d->srcC[i],
                             "conv2d: NHWC source %d needs C%%16==0, ld%%4==0 (C=%d ld=%d)", i, d->srcC[i], d->srcLd[i]);
        }
    }
    if (d->inNCHW) CP_CHECK_ARG(d->nsrc == 1 && d->K >= ctot * d->kh * d->kw, "conv2d: bad stem descriptor");
    else CP_CHECK_ARG(d->K == ctot * d->kh * d->kw, "conv2d: K=%d != kh*kw*Ctot=%d", d->K, ctot * d->kh * d->kw);
    a.nsrc = d->nsrc; a.Ctot = ctot;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.M = d->B * d->Ho * d->Wo;
    a.kh = d->kh; a.kw = d->kw; a.sy = d->sy; a.sx = d->sx; a.py = d->py; a.px = d->px;
    a.K = d->K; a.w = w; a.ldw = d->ldw; a.scale = scale; a.shift = shift;
    a.res = res; a.resLd = d->resLd; a.out = out; a.outLd = d->outLd; a.Cout = d->Cout;
    a.outNCHW = d->outNCHW; a.OH = d->OH; a.OW = d->OW; a.osy = d->osy; a.osx = d->osx; a.ooy = d->ooy; a.oox = d->oox;
    a.act = d->act; a.om = nullptr; a.omLd = 0; a.omMaskOff = 0; a.omSigmoid = 0; a.dg = 1; a.dily = a.dilx = 1; a.ksplit = d->ksplit > 1 ? d->ksplit : 1; a.nsub = d->nsub > 1 ? d->nsub : 1;
    CP_CHECK_ARG(a.M > 0 && (long long)d->B * d->H * d->W < (1ll << 31), "conv2d: bad problem size");
    for (int i = 0; i < d->nsrc && !d->inNCHW; ++i)
        CP_CHECK_ARG((long long)d->B * d->H * d->W * d->srcLd[i] * 4 < (1ll << 32), "conv2d: source %d exceeds 32-bit byte offsets", i);
    CP_CHECK_ARG(!(res && d->outNCHW), "conv2d: residual with NCHW output is not supported");
    return 0;
}

extern "C" int cp_conv2d_f32(const cp_conv_desc* d, const float* const* src, const float* w, const float* scale,
                             const float* shift, const float* res, float* out, void* stream)
{
    ConvArgs a;
    if (int rc = conv_args_from_desc(d, src, w, scale, shift, res, out, a)) return rc;
    CP_CHECK_ARG(a.ksplit == 1 || (!res && d->act == CP_ACT_NONE && !d->outNCHW && !d->inNCHW && d->nsrc == 1 && a.nsub == 1 &&
                                   d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->OH == d->Ho && d->OW == d->Wo),
                 "conv2d: split-K writes raw NHWC partial sums (one NHWC source, dense output, no residual, no activation; scale = ones, shift = zeros)");
    hipStream_t s = (hipStream_t)stream;

    int tile = d->tile;
    if (tile >= 3000000 && tile < 4000000) {                                             // opt-in split-bf16 kernel (conv_igemm_bf16x3.hip): w = pre-split weights
        CP_CHECK_ARG(!d->inNCHW, "conv2d: the split-bf16 kernel takes NHWC sources");
        CP_CHECK_ARG(a.nsub == 1 || (a.nsub == 4 && d->osy == 2 && d->osx == 2 && d->ooy == 0 && d->oox == 0 && d->kh == 2 && d->kw == 2),
                     "conv2d: nsub=4 is the fused k4/s2/p1 deconvolution (2x2 taps, output stride 2)");
        if (int rc = cp_launch_conv_bf16x3(a, tile, s)) return rc;
        CP_CHECK_LAUNCH("igemm_bf16x3_kernel");
        return 0;
    }
    if (a.nsub > 1) {
        CP_CHECK_ARG(a.nsub == 4 && !d->inNCHW && d->osy == 2 && d->osx == 2 && d->ooy == 0 && d->oox == 0 && d->kh == 2 && d->kw == 2,
                     "conv2d: nsub=4 is the fused k4/s2/p1 deconvolution (2x2 taps, output stride 2)");
        CP_CHECK_ARG(tile == 0 || tile > 1000, "conv2d: nsub needs the generic kernel");
    }
    if (d->inNCHW && tile == 0) {                                      // 3-channel 3x3 / stride 2 stem with 64 outputs (stem7x7.hip)
        const int prc = cp_launch_stem3x3(a, s);
        if (prc >= 0) {
            if (prc) return prc;
            CP_CHECK_LAUNCH("stem7x7_c16_kernel");
            return 0;
        }
    }
    if (a.nsub == 1 && a.ksplit == 1 && (tile == 0 || tile == 16)) {   // 16-channel 3x3 (stride 1 / 2): weights-in-registers kernel (conv3x3_c16.hip)
        const int prc = cp_launch_conv3x3_c16(a, d->inNCHW, s);
        if (prc >= 0) {
            if (prc) return prc;
            CP_CHECK_LAUNCH("conv3x3_c16_kernel");
            return 0;
        }
        CP_CHECK_ARG(tile == 0, "conv2d: 16-channel kernel requested for an ineligible shape");
    }
    if (a.nsub == 1 && a.ksplit == 1 && (tile == 0 || tile == 3 || tile == 332)) {   // 3x3/s1/p1 NHWC: LDS-resident halo patch kernel (conv3x3_patch.hip)
        const int prc = cp_launch_conv3x3_patch(a, d->inNCHW, s, tile == 332 ? 32 : 0);
        if (prc >= 0) {
            if (prc) return prc;
            CP_CHECK_LAUNCH("conv3x3_patch_kernel");
            return 0;
        }
        CP_CHECK_ARG(tile == 0, "conv2d: patch kernel requested for an ineligible shape");
    }
    if (!d->inNCHW && (tile == 0 || tile == 1)) {   // short-K 1x1 layers that are HBM-bound: everything requested up front (conv_pointwise.hip)
        const char* sw = getenv("CP_POINTWISE");     // "0": the generic kernel for these too (A/B switch; read when a launch is recorded)
        const int prc = (tile == 0 && sw && sw[0] == '0') ? -1 : cp_launch_conv_pointwise(a, s);
        if (prc >= 0) {
            if (prc) return prc;
            CP_CHECK_LAUNCH("pw_conv_kernel");
            return 0;
        }
        CP_CHECK_ARG(tile == 0, "conv2d: pointwise kernel requested for an ineligible shape");
    }
    if (tile == 0) {
        // heuristic: N tile from the padded channel count, M tile from how many blocks fill 256 CUs
        if (d->ldw % 64 != 0) tile = (d->ldw % 32 == 0) ? 128032 : 256016;
        else {
            // measured on MI355X: 128x64 beats 128x128 (register pressure, occupancy 2) on every shape;
            // 64x64 wins once a launch has fewer than ~3 blocks per CU
            const long long blocks128 = (long long)cp_cdiv(a.M, 128) * (d->ldw / 64);
            tile = blocks128 >= 4096 ? 128064 : 64064;
        }
    }
    int rc = 0;
    const int occ = tile / 10000000;           // 0 = no limit
    tile %= 10000000;
    if (d->inNCHW) {
        switch (tile) {
            case 256016: rc = launch_conv<256, 16, 4, 1, 16, true>(a, s); break;
            case 128032: rc = launch_conv<128, 32, 4, 1, 32, true>(a, s); break;
            case 128064: case 64064: case 128128: rc = launch_conv<128, 64, 2, 2, 32, true>(a, s); break;
            default: CP_CHECK_ARG(false, "conv2d: unknown stem tile %d", tile);
        }
    } else {
        switch (tile) {
            case 256016: rc = launch_conv<256, 16, 4, 1, 16, false>(a, s); break;
            case 128032: rc = launch_conv<128, 32, 4, 1, 32, false>(a, s); break;
            case 128064: rc = launch_conv<128, 64, 2, 2, 32, false>(a, s, occ); break;
            case 64064: rc = launch_conv<64, 64, 2, 2, 32, false>(a, s, occ); break;
            case 128128: rc = launch_conv<128, 128, 2, 2, 32, false>(a, s, occ); break;
            default: CP_CHECK_ARG(false, "conv2d: unknown tile %d", tile);
        }
    }
    if (rc) return rc;
    CP_CHECK_LAUNCH("igemm_conv_kernel");
    return 0;
}

// Winograd F(2x2,3x3) path (conv3x3_wino.hip): same descriptor, `u` = cp_winograd_pack_f32 output.
// d->tile: 0 = auto, 1 / 2 = 32 / 64 output channels per block.
// KeypointHead branch (lib/models/heads/keypoint.py:14-37) with n2 <= 34 outputs as ONE launch: d / src / u / scale / shift describe the
// 3x3 conv (C = 64 -> Cmid, bias in `shift`, act = ReLU) exactly as for cp_conv3x3_winograd_f32; w2 [n2][ld2] / b2 [n2] are the
// 1x1 conv; out2 is the reference's NCHW output [B, n2, H, W]; act2 = CP_ACT_SIGMOID for hm (multi_pose.py:35-37).
extern "C" int cp_head3x3_1x1_f32(const cp_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                                  const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, void* stream)
{
    ConvArgs a;
    const float* srcs[1] = {src};
    CP_CHECK_ARG(d && d->nsrc == 1 && !d->inNCHW && w2 && b2 && out2, "head3x3_1x1: one NHWC source and the 1x1 operands expected");
    if (int rc = conv_args_from_desc(d, srcs, u, scale, shift, nullptr, out2, a)) return rc;
    CP_CHECK_ARG(a.ksplit == 1, "head3x3_1x1: no split-C");
    // d->tile == 24: `u` is cp_winograd24_pack_f32's layout and the launch goes to the F(2x4,3x3) head kernel (head_wino24.hip)
    const int rc = d->tile == 24 ? cp_launch_head3x3_1x1_w24(a, w2, b2, out2, n2, ld2, act2, (hipStream_t)stream)
                                 : cp_launch_head3x3_1x1(a, w2, b2, out2, n2, ld2, act2, (hipStream_t)stream);
    CP_CHECK_ARG(rc >= 0, "head3x3_1x1: shape not eligible (C = 64, Cmid %% 32 == 0, ReLU, n2 <= 34, 16-byte aligned operands)");
    if (rc) return rc;
    CP_CHECK_LAUNCH("head3x3_1x1 kernel");
    return 0;
}

// Up to 8 INDEPENDENT generic convolutions in ONE launch (HRNet's fuse layers): d[i], src[i], w[i], scale[i], shift[i], res[i] (may be
// NULL), out[i] describe member i exactly as for cp_conv2d_f32 with ONE NHWC source, NHWC dense output, ldw % 64 == 0, no split-K, no
// sub-pixel deconvolution; the members must not alias each other's outputs.
extern "C" int cp_conv2d_group_f32(const cp_conv_desc* d, int n, const float* const* src, const float* const* w, const float* const* scale,
                                   const float* const* shift, const float* const* res, float* const* out, void* stream)
{
    CP_CHECK_ARG(d && src && w && scale && shift && res && out && n >= 1 && n <= IG_GROUP_MAX, "conv2d_group: 1..%d members, no null arrays", IG_GROUP_MAX);
    using T = IgTile<64, 64, 2, 2, 32>;
    IgGroup g;
    g.n = n;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        ConvArgs a;
        const float* srcs[1] = {src[i]};
        CP_CHECK_ARG(d[i].nsrc == 1 && !d[i].inNCHW && !d[i].outNCHW && d[i].nsub <= 1 && d[i].ksplit <= 1 && d[i].ldw % 64 == 0,
                     "conv2d_group: member %d: one NHWC source, NHWC output, ldw %% 64 == 0, no nsub / ksplit", i);
        CP_CHECK_ARG(d[i].osy == 1 && d[i].osx == 1 && d[i].ooy == 0 && d[i].oox == 0 && d[i].OH == d[i].Ho && d[i].OW == d[i].Wo,
                     "conv2d_group: member %d: dense output expected", i);
        if (int rc = conv_args_from_desc(&d[i], srcs, w[i], scale[i], shift[i], res[i], out[i], a)) return rc;
        IgMember& m = g.m[i];
        m.src = a.src[0]; m.w = a.w; m.scale = a.scale; m.shift = a.shift; m.res = a.res; m.out = a.out;
        m.C = a.srcC[0]; m.srcLd = a.srcLd[0]; m.B = a.B; m.H = a.H; m.W = a.W; m.Ho = a.Ho; m.Wo = a.Wo;
        m.kh = a.kh; m.kw = a.kw; m.sy = a.sy; m.sx = a.sx; m.py = a.py; m.px = a.px; m.K = a.K; m.ldw = a.ldw; m.resLd = a.resLd;
        m.outLd = a.outLd; m.Cout = a.Cout; m.act = a.act; m.pad_ = 0;
        g.first[i] = (int)total;
        total += (long long)cp_cdiv(a.M, 64) * (a.ldw / 64);
    }
    for (int i = n; i <= IG_GROUP_MAX; ++i) g.first[i] = (int)total;
    for (int i = n; i < IG_GROUP_MAX; ++i) g.m[i] = g.m[0];
    CP_CHECK_ARG(total < (1ll << 31), "conv2d_group: grid %lld too large", total);
    hipLaunchKernelGGL(igemm_conv_group_kernel, dim3((unsigned)total), dim3(IG_THREADS), T::NHWC_BYTES, (hipStream_t)stream, g);
    cp_note_kernel("igemm_conv_group_kernel");
    CP_CHECK_LAUNCH("igemm_conv_group_kernel");
    return 0;
}

// Up to four INDEPENDENT 3x3 / stride-1 convolutions on the F(2x4,3x3) kernel in ONE launch (HRNet's parallel branches): d[i], src[i],
// u[i] (cp_winograd24_pack_f32), scale[i], shift[i], res[i] (may be NULL), out[i] describe member i exactly as for
// cp_conv3x3_winograd_f32 with tile = 24; the members must not alias each other's outputs.
extern "C" int cp_conv3x3_winograd24_group_f32(const cp_conv_desc* d, int n, const float* const* src, const float* const* u,
                                               const float* const* scale, const float* const* shift, const float* const* res,
                                               float* const* out, void* stream)
{
    CP_CHECK_ARG(d && src && u && scale && shift && res && out && n >= 1 && n <= 4, "conv3x3_winograd24_group: 1..4 members, no null arrays");
    ConvArgs a[4];
    for (int i = 0; i < n; ++i) {
        const float* srcs[1] = {src[i]};
        CP_CHECK_ARG(d[i].nsrc == 1 && !d[i].inNCHW, "conv3x3_winograd24_group: member %d: one NHWC source expected", i);
        if (int rc = conv_args_from_desc(&d[i], srcs, u[i], scale[i], shift[i], res[i], out[i], a[i])) return rc;
        CP_CHECK_ARG(a[i].ksplit == 1, "conv3x3_winograd24_group: no split launches inside a group");
    }
    if (int rc = cp_launch_conv3x3_wino24_group(a, n, (hipStream_t)stream)) return rc;
    CP_CHECK_LAUNCH("conv3x3_wino24_group_kernel");
    return 0;
}

extern "C" int cp_conv3x3_winograd_f32(const cp_conv_desc* d, const float* src, const float* u, const float* scale,
                                       const float* shift, const float* res, float* out, void* stream)
{
    ConvArgs a;
    const float* srcs[1] = {src};
    CP_CHECK_ARG(d && d->nsrc == 1 && !d->inNCHW, "conv3x3_winograd: one NHWC source expected");
    if (int rc = conv_args_from_desc(d, srcs, u, scale, shift, res, out, a)) return rc;
    CP_CHECK_ARG(a.ksplit == 1 || (!res && d->act == CP_ACT_NONE && !d->outNCHW),
                 "conv3x3_winograd: split-C writes raw NHWC partial outputs (no residual, no activation; scale = ones, shift = zeros)");
    const int rc = cp_launch_conv3x3_wino(a, (hipStream_t)stream, d->tile);
    CP_CHECK_ARG(rc >= 0, "conv3x3_winograd: shape not eligible (3x3, stride 1, pad 1, NHWC, C %% 16 == 0)");
    if (rc) return rc;
    CP_CHECK_LAUNCH("conv3x3_wino_kernel");
    return 0;
}
