// Fused modulated deformable convolution (DCNv2) forward for gfx950.
//
// Reference: lib/models/backbones/DCNv2/src/cuda/dcn_v2_cuda.cu:42-172 materialises
// columns[B, C*9, Ho*Wo] in HBM with modulated_deformable_im2col_gpu_kernel
// (dcn_v2_im2col_cuda.cu:125-195, bilinear :25-54) and then runs two batched cuBLAS SGEMMs
// (bias broadcast :123-137, out += W * columns :149-163).  Here the sampled, mask-modulated
// A-tile goes straight from registers into LDS and is consumed by fp32 MFMA: the `columns`
// round trip (37.7 MB per image for a 64-channel 128x128 layer) never exists, and the
// DeformConv wrapper's BatchNorm + ReLU (pose_dla_dcn.py:345-348) fold into the epilogue.
//
// Semantics reproduced exactly (Appendix B.3 of SURVEY.md):
//   h_im = oy*sy - py + ky*dil + dy_k ; sample valid iff h_im > -1 && w_im > -1 && h_im < H && w_im < W
//   4-corner bilinear with per-corner zeroing (h_low >= 0, w_high <= W-1, ...), value * mask_k.
// Per block: the (pixel, tap) sampling records {clamped corner base, dx, dy bits, 4 weights*mask}
// are computed once into LDS (9 * BM * 20 B); each k-step then issues 4 corner float4 loads per
// (pixel, 4-channel quad), blends in registers and stages the result k-major like a plain conv.
#include <cstdlib>
#include "igemm.h"

#define DCN_MAX_TAPS 121         // 11 x 11: the record rows live in LDS (20 B per (tap, pixel, deformable group)); the launcher checks the total
typedef float dcn_v2 __attribute__((ext_vector_type(2)));

// r = w.x * c00 + w.y * c01 + w.z * c10 + w.w * c11 on two channels (wxy = (w.x, w.y), wzw = (w.z, w.w)): one v_pk_mul_f32 and three
// v_pk_fma_f32, the scalar weight broadcast through op_sel (low element) / op_sel + op_sel_hi (high element).  Same products, same order
// of accumulation as the elementwise form (fma(w.w, c11, fma(w.z, c10, fma(w.y, c01, w.x * c00)))).
__device__ __forceinline__ void dcn_blend(dcn_v2& r, dcn_v2 wxy, dcn_v2 wzw, dcn_v2 c00, dcn_v2 c01, dcn_v2 c10, dcn_v2 c11)
{
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(wxy), "v"(c00));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(r) : "v"(wxy), "v"(c01));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r) : "v"(wzw), "v"(c10));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(r) : "v"(wzw), "v"(c11));
}

// One k-step = 16 input channels of one tap: 4 lanes (float4 quads) per pixel, 64 pixels per pass.  The gathers (and the
// weight slice) of k-step s+1 are issued from inside the MFMA block of k-step s.  Variants measured and removed in round 3
// (two-deep prefetch, full-line 32-channel gathers, 32-channel k-steps): DESIGN.md 7.3.
template <int BM, int BN, int WAVES_M, int WAVES_N, int MF>
__global__ __launch_bounds__(IG_THREADS, 3) void dcn_igemm_kernel(const ConvArgs a)
{
    using T = IgTile<BM, BN, WAVES_M, WAVES_N, MF>;
    constexpr int QL = 4;                           // lanes (float4 quads) per pixel
    constexpr int PPP = IG_THREADS / QL;            // pixels per pass
    constexpr int ASL = BM / PPP;                   // gather slots per thread per step
    static_assert(ASL >= 1, "BM too small for this gather width");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // records first, GEMM staging after; the NCHW epilogue reuses the whole region from the base
    // dg = deformable groups (dcn_v2_im2col_cuda.cu:153,162-164: input channel c samples with the offsets / mask of group
    // c / (C / dg)): one record set per group, record index (g * ntap + tap) * BM + pixel (ntap = kh * kw: any kernel size the
    // LDS holds -- round 6; rounds 1-5 had a compile-time row count of 9)
    const int dg = a.dg;
    const int ntap = a.kh * a.kw;
    float4* s_w = reinterpret_cast<float4*>(smem);                      // [dg][taps][BM] corner weights * mask
    int* s_code = reinterpret_cast<int*>(s_w + dg * ntap * BM);         // [dg][taps][BM] base | dx<<29 | dy<<30
    float* As0 = smem + dg * ntap * BM * 5;         // [2 buffers][A_FLOATS]
    float* Bs0 = As0 + 2 * T::A_FLOATS;             // [2 buffers][B_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NT = a.ldw / BN;
    int tile = ig_xcd_remap(blockIdx.x, gridDim.x);
    // split-K over the taps (small-M layers: 512 -> 256 @16x16 is 256 blocks of 288 k-steps on 256 CUs): consecutive blocks are
    // the splits of one tile, so they gather from the same neighbourhood
    const int S = a.ksplit, sp = tile % S;
    tile /= S;
    const int nt = tile % NT, mt = tile / NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wid / WAVES_N) * T::WM, wn0 = (wid % WAVES_N) * T::WN;
    const int HoWo = a.Ho * a.Wo;
    const int tap0 = sp * ntap / S, tap1 = (sp + 1) * ntap / S;       // this block's taps
    const int C = a.srcC[0], ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];

    // ---- sampling records for every (tap, pixel) of this tile.  A thread keeps ONE pixel (tid % BM) and walks the taps
    // tid / BM, + 256 / BM, ...: the pixel's (b, oy, ox) decomposition (two integer divisions by run-time values, ~70 VALU) is
    // done once instead of once per record.  Round 3: PMC counts 1 434 VALU instructions per wave for a 64 -> 64 layer against
    // 288 MFMAs -- 612 of them in the main loop and ~500 HERE (the disassembly had 170 per record: a vector integer division for
    // the tap's (ky, kx), a full IEEE division in the sigmoid, twelve selects), and every VALU instruction costs matrix-pipe
    // time.  Now: wave-uniform tap walk on the scalar unit, v_rcp_f32 for the sigmoid (1 ulp; the mask multiplies a value the
    // path holds to 1e-4), row / column weights selected BEFORE the four products (4 selects instead of 12).
    {
        static_assert(IG_THREADS % BM == 0, "one pixel per thread");
        constexpr int TS = IG_THREADS / BM;                               // taps advanced per pass
        const int pl = tid % BM, m = m0 + pl;
        const bool live = m < a.M;
        const int b = live ? m / HoWo : 0, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
        const float* omp = a.om + (size_t)(live ? m : 0) * a.omLd;
        const int by = oy * a.sy - a.py, bx = ox * a.sx - a.px, bpix = b * a.H * a.W;
        const float fH = (float)a.H, fW = (float)a.W;
        const unsigned ld4 = (unsigned)ld >> 2;
        for (int g = 0; g < dg; ++g) {                                     // (scalar loop; dg = 1 for every network layer)
        const int og = g * ntap;                                           // group g: offsets at 2 * (og + t), mask at omMaskOff + og + t
        int t = tap0 + __builtin_amdgcn_readfirstlane(tid / BM);           // wave-uniform: BM is a multiple of the wave size
        int ky = t / a.kw, kx = t - ky * a.kw;                             // scalar
        for (; t < tap1; t += TS) {
            const float offh = omp[2 * (og + t)], offw = omp[2 * (og + t) + 1];
            float mk = omp[a.omMaskOff + og + t];
            if (a.omSigmoid) mk = __builtin_amdgcn_rcpf(1.0f + __expf(-mk));
            const float h_im = (float)(by + ky * a.dily) + offh;
            const float w_im = (float)(bx + kx * a.dilx) + offw;
            const bool valid = live && h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const int h_low = (int)hf, w_low = (int)wf;
            const bool t_ok = h_low >= 0, b_ok = h_low < a.H - 1, l_ok = w_low >= 0, r_ok = w_low < a.W - 1;
            // the record's base is the top-left corner clamped into the image: when the top (left) corner is out of range the base
            // already IS the bottom (right) corner and takes its weight; the second row (column) is used only when both are in range
            const bool dyb = t_ok && b_ok, dxb = l_ok && r_ok;
            const float mv = valid ? mk : 0.f;
            const float top = (t_ok ? hh : lh) * mv, bot = (dyb ? lh : 0.f) * mv;
            const float lft = l_ok ? hw : lw, rgt = dxb ? lw : 0.f;
            const int yl = min(max(h_low, 0), a.H - 1), xl = min(max(w_low, 0), a.W - 1);      // in range even for an invalid sample
            s_w[(g * ntap + t) * BM + pl] = make_float4(top * lft, top * rgt, bot * lft, bot * rgt);
            // base = byte offset of the clamped top-left corner in 16-byte units (< 2^28: the launcher checks 32-bit byte offsets): the
            // multiply by the pixel stride is done HERE, once per record, not once per (tap, k-walk) in the main loop
            s_code[(g * ntap + t) * BM + pl] = (int)((unsigned)(bpix + yl * a.W + xl) * ld4) | ((dxb ? 1 : 0) << 29) | ((dyb ? 1 : 0) << 30);
            kx += TS;
            while (kx >= a.kw) { kx -= a.kw; ++ky; }
        }
        }
    }

    typename IgAcc<MF>::type acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < IgAcc<MF>::N; ++r) acc[i][j][r] = 0.f;

    const int ks0 = tap0 * (C / IG_BK), nk = tap1 * (C / IG_BK);       // k-steps [ks0, nk)
    const int q = tid % QL;
    float4 br[T::B_SLOTS];
    float4 c00[ASL], c01[ASL], c10[ASL], c11[ASL], wq[ASL];
    int tap = tap0, cl = 0;
    const int cpg = C / dg;                         // channels per deformable group (a multiple of the 16-channel k-step)
    int gnext = 0, grec = 0;                        // first channel of the next group inside the tap; record row of the current group
    __syncthreads();

    // Corner byte offsets and blend weights are per (tap, pixel): computed at the first k-step of a tap and reused by its
    // C/16 k-steps - the channel offset of the step goes into the (scalar) base address, so the steady state issues the
    // four gathers with no address VALU at all (every VALU instruction costs ~4 cycles of matrix-pipe time).
    const unsigned pixb = (unsigned)ld * 4u, rowb = (unsigned)a.W * pixb;
    unsigned o00[ASL], o01[ASL], o10[ASL], o11[ASL];
    auto load_a = [&]() __attribute__((always_inline)) {
        if (cl == gnext) {                          // first k-step of a (tap, group): dg = 1 -> once per tap
            grec = (cl == 0 ? 0 : grec + ntap);
            gnext = cl + cpg;
#pragma unroll
            for (int s = 0; s < ASL; ++s) {
                const int pl = tid / QL + s * PPP;
                const int code = s_code[(grec + tap) * BM + pl];
                wq[s] = s_w[(grec + tap) * BM + pl];
                o00[s] = (((unsigned)code & 0x0FFFFFFFu) << 4) + (unsigned)q * 16u;
                o01[s] = o00[s] + (((unsigned)code >> 29) & 1u) * pixb;
                o10[s] = o00[s] + (((unsigned)code >> 30) & 1u) * rowb;
                o11[s] = o10[s] + (o01[s] - o00[s]);
            }
        }
        const char* xs = reinterpret_cast<const char*>(x) + (size_t)cl * 4;      // uniform
#pragma unroll
        for (int s = 0; s < ASL; ++s) {
            c00[s] = ig_ldg4(reinterpret_cast<const float*>(xs + o00[s]));
            c01[s] = ig_ldg4(reinterpret_cast<const float*>(xs + o01[s]));
            c10[s] = ig_ldg4(reinterpret_cast<const float*>(xs + o10[s]));
            c11[s] = ig_ldg4(reinterpret_cast<const float*>(xs + o11[s]));
        }
    };
    auto advance = [&]() __attribute__((always_inline)) { cl += IG_BK; if (cl >= C) { cl = 0; gnext = 0; ++tap; } };
    auto store_a = [&](float* As) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < ASL; ++s) {
            const int pl = tid / QL + s * PPP;
            // packed fp32 (v_pk_mul / v_pk_fma, weight broadcast through op_sel): 8 VALU instructions per float4
            // instead of 16 -- every VALU instruction costs ~4 cycles of matrix-pipe time (tools/micro/wino_loop.hip)
            // the four blend weights sit in two register pairs; op_sel / op_sel_hi broadcast the low or the high element of a pair to
            // both halves of the packed operation.  Written out because the compiler copies the odd elements into fresh even
            // registers first (2 v_mov_b32 per k-step in the main loop's 14 VALU instructions).
            const float4 w = wq[s];
            const dcn_v2 wxy = {w.x, w.y}, wzw = {w.z, w.w};
            dcn_v2 lo, hi;
            dcn_blend(lo, wxy, wzw, (dcn_v2){c00[s].x, c00[s].y}, (dcn_v2){c01[s].x, c01[s].y}, (dcn_v2){c10[s].x, c10[s].y}, (dcn_v2){c11[s].x, c11[s].y});
            dcn_blend(hi, wxy, wzw, (dcn_v2){c00[s].z, c00[s].w}, (dcn_v2){c01[s].z, c01[s].w}, (dcn_v2){c10[s].z, c10[s].w}, (dcn_v2){c11[s].z, c11[s].w});
            *reinterpret_cast<float4*>(As + pl * IG_LDK + q * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
    };

    // weight slice of k-step ks: the thread's byte offset inside the packed matrix is fixed, the k offset rides in the SCALAR
    // base (global_load v, v_off, s[base]: no 64-bit address VALU per k-step)
    unsigned boff[T::B_SLOTS];
#pragma unroll
    for (int s = 0; s < T::B_SLOTS; ++s) {
        const int idx = tid + s * IG_THREADS;
        boff[s] = ((unsigned)(n0 + (idx >> 2)) * (unsigned)a.K + (unsigned)(idx & 3) * 4u) * 4u;
    }
    auto load_b = [&](int ks) __attribute__((always_inline)) {
        const char* wb = reinterpret_cast<const char*>(a.w) + (size_t)ks * (IG_BK * 4);      // uniform
#pragma unroll
        for (int s = 0; s < T::B_SLOTS; ++s)
            if (T::B_F4 % IG_THREADS == 0 || tid + s * IG_THREADS < T::B_F4) br[s] = ig_ldg4(reinterpret_cast<const float*>(wb + boff[s]));
    };
    load_a(); advance();
    load_b(ks0);
    store_a(As0);
    ig_store_b<T>(Bs0, tid, br);
    __syncthreads();
    int cur = 0;
    for (int ks = ks0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        // the next step's gathers are issued from inside the MFMA block (half-way through the k-step)
        ig_compute<T, MF>(As0 + cur * T::A_FLOATS, Bs0 + cur * T::B_FLOATS, wm0, wn0, lane, acc, [&]() __attribute__((always_inline)) {
            if (more) { load_a(); advance(); load_b(ks + 1); }
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
    if (S > 1) {
        // raw partial sums of this split (the launcher passes scale = 1, shift = 0, no activation, NHWC with outLd = ldw)
        ConvArgs e = a;
        e.out = a.out + (size_t)sp * a.M * a.outLd;
        ig_epilogue<T, BM, BN, MF>(e, smem, m0, n0, wm0, wn0, lane, tid, acc);
    } else ig_epilogue<T, BM, BN, MF>(a, smem, m0, n0, wm0, wn0, lane, tid, acc);
}


template <int BM, int BN, int WAVES_M, int WAVES_N, int MF>
static int launch_dcn(const ConvArgs& a, hipStream_t s)
{
    using T = IgTile<BM, BN, WAVES_M, WAVES_N, MF>;
    auto kern = dcn_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MF>;
    if (a.srcC[0] % IG_BK != 0) { cp_set_error("dcn: C=%d is not a multiple of %d", a.srcC[0], IG_BK); return 1; }
    if (a.ldw % BN != 0) { cp_set_error("dcn: ldw=%d is not a multiple of the N tile %d", a.ldw, BN); return 1; }
    if ((a.srcC[0] / a.dg) % IG_BK != 0) { cp_set_error("dcn: C / deformable_group = %d is not a multiple of %d", a.srcC[0] / a.dg, IG_BK); return 1; }
    const int main_bytes = T::MAIN_BYTES + a.dg * a.kh * a.kw * BM * 20;
    if (main_bytes > 160 * 1024) { cp_set_error("dcn: %d x %d taps x %d deformable group(s) need %d B of LDS for the sampling records", a.kh, a.kw, a.dg, main_bytes); return 1; }
    const int epi = a.outNCHW ? T::EPI_BYTES : T::EPV_BYTES;
    const int smem = epi > main_bytes ? epi : main_bytes;
    static CpLdsGuard guard;
    if (smem > 64 * 1024) {
        const hipError_t e = guard.ensure((const void*)kern, smem);
        if (e != hipSuccess) { cp_set_error("dcn_v2: cannot reserve %d B LDS: %s", smem, hipGetErrorString(e)); return 2; }
    }
    const int grid = cp_cdiv(a.M, BM) * (a.ldw / BN) * a.ksplit;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(IG_THREADS), smem, s, a);
    cp_note_kernel("dcn_igemm_kernel<%d, %d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N, MF);
    return 0;
}

struct cp_dcn_desc {
    int B, H, W, C, srcLd;     // input NHWC
    int Ho, Wo;
    int kh, kw, sy, sx, py, px, dily, dilx;
    int K, ldw, Cout;
    int omLd, omSigmoid;       // om[B,Ho,Wo,omLd]: 2k = dy, 2k+1 = dx, 2*kh*kw + k = mask (logit if omSigmoid)
    int outLd, outNCHW, act;
    int tile;
    int ksplit;                // 0 / 1: whole K in one block.  S > 1: split-K over the taps -- `out` is a workspace [S][M][outLd] of raw
                               // partial sums (pass scale = 1, shift = 0, act = none, NHWC, outLd = ldw); cp_splitk_reduce_f32 finishes
    int dg;                    // 0 / 1: one deformable group.  G > 1: input channel c uses the offsets / mask of group c / (C / G);
                               // om channels 2 * (g * kh*kw + k) = dy, + 1 = dx, 2 * G * kh*kw + g * kh*kw + k = mask; (C / G) % 16 == 0
};

extern "C" int cp_sizeof_dcn_desc(void) { return (int)sizeof(cp_dcn_desc); }

extern "C" int cp_dcn_v2_f32(const cp_dcn_desc* d, const float* x, const float* om, const float* w, const float* scale,
                             const float* shift, float* out, void* stream)
{
    CP_CHECK_ARG(d && x && om && w && scale && shift && out, "dcn_v2: null pointer");
    CP_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= DCN_MAX_TAPS, "dcn_v2: 1..%d taps (kh=%d kw=%d)", DCN_MAX_TAPS, d->kh, d->kw);
    CP_CHECK_ARG(d->sy > 0 && d->sx > 0 && d->dily > 0 && d->dilx > 0 && d->py >= 0 && d->px >= 0, "dcn_v2: stride / dilation > 0, pad >= 0");
    CP_CHECK_ARG(d->C % 16 == 0 && d->srcLd % 4 == 0 && d->srcLd >= d->C, "dcn_v2: C%%16==0, ld%%4==0 required");
    CP_CHECK_ARG(d->K == d->kh * d->kw * d->C, "dcn_v2: K=%d != kh*kw*C", d->K);
    CP_CHECK_ARG(d->ldw % 16 == 0 && d->ldw >= d->Cout, "dcn_v2: ldw=%d Cout=%d", d->ldw, d->Cout);
    const int dgv = d->dg > 1 ? d->dg : 1;
    CP_CHECK_ARG(d->C % dgv == 0 && (d->C / dgv) % 16 == 0, "dcn_v2: C=%d must split into %d deformable groups of a multiple of 16 channels", d->C, dgv);
    CP_CHECK_ARG(d->omLd >= 3 * dgv * d->kh * d->kw, "dcn_v2: omLd=%d too small", d->omLd);
    CP_CHECK_ARG((long long)d->B * d->H * d->W < (1ll << 29) && (long long)d->B * d->H * d->W * d->srcLd * 4 < (1ll << 32),
                 "dcn_v2: input too large (pixel index 29 bits, byte offsets 32 bits)");
    ConvArgs a;
    for (int i = 0; i < IG_MAX_SRC; ++i) { a.src[i] = nullptr; a.srcC[i] = 0; a.srcLd[i] = 0; }
    a.src[0] = x; a.srcC[0] = d->C; a.srcLd[0] = d->srcLd; a.nsrc = 1; a.Ctot = d->C;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.M = d->B * d->Ho * d->Wo;
    a.kh = d->kh; a.kw = d->kw; a.sy = d->sy; a.sx = d->sx; a.py = d->py; a.px = d->px;
    a.K = d->K; a.w = w; a.ldw = d->ldw; a.scale = scale; a.shift = shift; a.res = nullptr; a.resLd = 0;
    a.out = out; a.outLd = d->outLd; a.Cout = d->Cout; a.outNCHW = d->outNCHW;
    a.OH = d->Ho; a.OW = d->Wo; a.osy = a.osx = 1; a.ooy = a.oox = 0; a.act = d->act;
    a.om = om; a.omLd = d->omLd; a.omMaskOff = 2 * dgv * d->kh * d->kw; a.omSigmoid = d->omSigmoid; a.dg = dgv;
    a.dily = d->dily; a.dilx = d->dilx; a.nsub = 1;
    a.ksplit = d->ksplit > 1 ? d->ksplit : 1;
    CP_CHECK_ARG(a.ksplit <= d->kh * d->kw, "dcn_v2: ksplit=%d exceeds the %d taps", a.ksplit, d->kh * d->kw);
    CP_CHECK_ARG(a.ksplit == 1 || (!d->outNCHW && d->act == CP_ACT_NONE && d->outLd == d->ldw && d->Cout == d->ldw),
                 "dcn_v2: split-K writes raw NHWC partial sums (outLd == Cout == ldw, no activation)");
    hipStream_t s = (hipStream_t)stream;
    int tile = d->tile;
    if (tile == 0) {
        static const int force = getenv("CP_DCN_TILE") ? atoi(getenv("CP_DCN_TILE")) : 0;      // A/B switch for profiling
        if (force && a.ksplit == 1) tile = force;
    }
    if (tile == 0) {
        if (d->ldw % 64 != 0) tile = (d->ldw % 32 == 0) ? 128032 : 0;
        else {
            tile = 64064;
            // 128 output channels per block (one gathered + blended A tile feeds twice the MFMAs) where that still leaves two
            // blocks per CU: the 128-channel layers at 64x64 and the 256-channel one at 32x32 of DLA-34, B = 16 (+0.5 % end to end)
            if (d->ldw % 128 == 0 && (long long)cp_cdiv(a.M, 64) * (d->ldw / 128) >= 500) tile = 64128;

        }      // measured (MI355X, B = 16): 64x64 beats 128x64 by ~11 % on every DLA-34 shape (more, smaller blocks
                                // interleave gather and MFMA phases better; the kernel is L1-gather-bound, not tile-reuse-bound)
    }
    int rc = 0;
    switch (tile) {
        case 128032: rc = launch_dcn<128, 32, 4, 1, 32>(a, s); break;
        case 128064: rc = launch_dcn<128, 64, 2, 2, 32>(a, s); break;
        case 64064: rc = launch_dcn<64, 64, 2, 2, 32>(a, s); break;
        case 64032: rc = launch_dcn<64, 32, 4, 1, 16>(a, s); break;
        case 64128: rc = launch_dcn<64, 128, 2, 2, 32>(a, s); break;
        case 6064128: rc = (d->ldw % 128 == 0) ? launch_dcn<64, 128, 2, 2, 32>(a, s) : launch_dcn<64, 64, 2, 2, 32>(a, s); break;
        default: CP_CHECK_ARG(false, "dcn_v2: unsupported tile %d (ldw=%d)", tile, d->ldw);
    }
    if (rc) return rc;
    CP_CHECK_LAUNCH("dcn_igemm_kernel");
    return 0;
}
