// DCNv2 forward, "wave-private A" structure (round 3 experiment -> see DESIGN.md 7.3 for what was measured).
//
// Same arithmetic as dcn.hip (reference: dcn_v2_im2col_cuda.cu:25-54,125-195 -- boundary rule, per-corner zeroing, value * mask;
// dcn_v2_cuda.cu:123-163 -- out = bias + W * columns), different loop structure.  dcn_igemm_kernel shares ONE A tile (64 sampled
// pixels x 16 channels) between four waves, so every 16-channel k-step costs a block barrier and a weight-slice round trip
// (global -> registers -> LDS) in front of 8 MFMAs per wave; PMC showed the matrix pipe busy 58 % with the gathers exonerated
// (the same kernel without them: 0.66-0.72 of peak).  Here
//   * a wave owns 32 output pixels and ALL BN output channels of the block (block = 128 pixels x BN): it gathers, blends and
//     stages its own A rows in a wave-private LDS slab and reads them back as MFMA fragments -- LDS executes a wave's
//     instructions in order, so no barrier is needed around that round trip;
//   * the weight tile of a whole phase (one tap x 64 input channels x BN outputs, 17 KB) is staged once, double buffered:
//     ONE block barrier per 4 k-steps (64 MFMAs per wave) instead of one per k-step;
//   * sampling records (clamped corner base, dx/dy bits, 4 blend weights x mask) are wave-private as well.
#include "igemm.h"

#define WP_KC 64                         // input channels per phase
#define WP_LDB (WP_KC + 4)               // weight-tile row pitch (floats): 16-lane b128 groups on distinct 4-bank groups
#define WP_TAPS 9
typedef float wp_v2 __attribute__((ext_vector_type(2)));

template <int BN> struct WpGeo {
    static constexpr int BM = 128;
    static constexpr int B_FLOATS = BN * WP_LDB;                 // one weight buffer
    static constexpr int A_FLOATS = 2 * 32 * IG_LDK;             // per wave: two slabs (k-step parity)
    static constexpr int REC = WP_TAPS * 32;                     // records per wave
    static constexpr int OFF_A = 2 * B_FLOATS;
    static constexpr int OFF_RW = OFF_A + 4 * A_FLOATS;          // float4 [4 waves][REC]
    static constexpr int OFF_RC = OFF_RW + 4 * REC * 4;          // int    [4 waves][REC]
    static constexpr int MAIN_FLOATS = OFF_RC + 4 * REC;
    static constexpr int B_SLOTS = BN * (WP_KC / 4) / IG_THREADS; // float4 per thread per weight tile
};

template <int BN>
__global__ __launch_bounds__(IG_THREADS, 2) void dcn_wp_kernel(const ConvArgs a)
{
    using T = IgTile<128, BN, 4, 1, 32>;
    using G = WpGeo<BN>;
    constexpr int TN = BN / 32;
    static_assert(T::TM == 1 && T::TN == TN, "one 32-pixel M tile per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = a.ldw / BN;
    const int tile = ig_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % NT, mt = tile / NT;
    const int m0 = mt * G::BM, n0 = nt * BN, wm0 = wid * 32;
    const int HoWo = a.Ho * a.Wo, ntap = a.kh * a.kw;
    const int C = a.srcC[0], ld = a.srcLd[0];
    const float* __restrict__ x = a.src[0];
    float* As = smem + G::OFF_A + wid * G::A_FLOATS;
    float4* rw = reinterpret_cast<float4*>(smem + G::OFF_RW) + wid * G::REC;
    int* rc = reinterpret_cast<int*>(smem + G::OFF_RC) + wid * G::REC;

    // ---- sampling records of the wave's 32 pixels: lane = (pixel, tap parity)
    {
        const int pl = lane & 31, m = m0 + wm0 + pl;
        const bool live = m < a.M;
        const int b = live ? m / HoWo : 0, p = m - b * HoWo, oy = p / a.Wo, ox = p - oy * a.Wo;
        const float* omp = a.om + (size_t)(live ? m : 0) * a.omLd;
        const int by = oy * a.sy - a.py, bx = ox * a.sx - a.px, bpix = b * a.H * a.W;
        const float fH = (float)a.H, fW = (float)a.W;
        for (int t = lane >> 5; t < ntap; t += 2) {
            const int ky = t / a.kw, kx = t - ky * a.kw;
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            int code = 0;
            if (live) {
                const float offh = omp[2 * t], offw = omp[2 * t + 1];
                float mk = omp[a.omMaskOff + t];
                if (a.omSigmoid) mk = 1.0f / (1.0f + __expf(-mk));
                const float h_im = (float)(by + ky * a.dily) + offh;
                const float w_im = (float)(bx + kx * a.dilx) + offw;
                if (h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW) {
                    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const bool t_ok = h_low >= 0, b_ok = h_high <= a.H - 1, l_ok = w_low >= 0, r_ok = w_high <= a.W - 1;
                    w4.x = (t_ok && l_ok) ? hh * hw * mk : 0.f;
                    w4.y = (t_ok && r_ok) ? hh * lw * mk : 0.f;
                    w4.z = (b_ok && l_ok) ? lh * hw * mk : 0.f;
                    w4.w = (b_ok && r_ok) ? lh * lw * mk : 0.f;
                    const int yl = t_ok ? h_low : 0, xl = l_ok ? w_low : 0;     // clamped, always in range
                    const int dy = (t_ok && b_ok) ? 1 : 0, dx = (l_ok && r_ok) ? 1 : 0;
                    code = (bpix + yl * a.W + xl) | (dx << 29) | (dy << 30);
                    // top / left corner out of range: the record's base already IS the bottom / right corner
                    if (!t_ok) { w4.x = w4.z; w4.y = w4.w; w4.z = 0.f; w4.w = 0.f; }
                    if (!l_ok) { w4.x = w4.y; w4.z = w4.w; w4.y = 0.f; w4.w = 0.f; }
                }
            }
            rw[t * 32 + pl] = w4;
            rc[t * 32 + pl] = code;
        }
    }

    typename IgAcc<32>::type acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    const int CPT = C / WP_KC;                       // phases per tap
    const int NP = ntap * CPT;                       // phases
    const int SPT = C / IG_BK;                       // k-steps per tap
    const int nk = a.K / IG_BK;
    const int q = lane & 3, pq = lane >> 2;          // channel quad, pixel of a pass (16 pixels per pass, two passes)
    const int il = lane & 31, g = lane >> 5;
    const unsigned pixb = (unsigned)ld * 4u, rowb = (unsigned)a.W * pixb;
    float4 wcur[2];                                  // blend weights of the tap being gathered
    // two gather register sets (k-step parity): the corner loads of k-step s+2 are issued while k-step s multiplies and are
    // consumed one k-step later -- a whole MFMA block (16 x 64 cycles) to land, with only two waves per SIMD to hide latency
    float4 c00[2][2], c01[2][2], c10[2][2], c11[2][2], wq[2][2];
    unsigned o00[2], o01[2], o10[2], o11[2];
    float4 br[G::B_SLOTS];
    __syncthreads();                                 // records visible (they are wave-private, but other lanes wrote them)

    // gather of k-step ks (clamped by the caller) into register set `set`: per-tap corner offsets at the first k-step of a tap,
    // the channel offset of the step rides in the scalar base address
    auto gather = [&](int ks, int set) __attribute__((always_inline)) {
        const int tap = ks / SPT, cl = (ks - tap * SPT) * IG_BK;      // uniform
        if (cl == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pl = pq + 16 * s;
                const int code = rc[tap * 32 + pl];
                wcur[s] = rw[tap * 32 + pl];
                o00[s] = (unsigned)(code & 0x1FFFFFFF) * pixb + (unsigned)q * 16u;
                o01[s] = o00[s] + (((unsigned)code >> 29) & 1u) * pixb;
                o10[s] = o00[s] + (((unsigned)code >> 30) & 1u) * rowb;
                o11[s] = o10[s] + (o01[s] - o00[s]);
            }
        }
        const char* xs = reinterpret_cast<const char*>(x) + (size_t)cl * 4;      // uniform
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            wq[set][s] = wcur[s];
            c00[set][s] = ig_ldg4(reinterpret_cast<const float*>(xs + o00[s]));
            c01[set][s] = ig_ldg4(reinterpret_cast<const float*>(xs + o01[s]));
            c10[set][s] = ig_ldg4(reinterpret_cast<const float*>(xs + o10[s]));
            c11[set][s] = ig_ldg4(reinterpret_cast<const float*>(xs + o11[s]));
        }
    };
    auto blend_store = [&](int set) __attribute__((always_inline)) {
        float* Aw = As + set * 32 * IG_LDK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 w = wq[set][s];
            const wp_v2 wx = {w.x, w.x}, wy = {w.y, w.y}, wz = {w.z, w.z}, ww = {w.w, w.w};
            const wp_v2 lo = __builtin_elementwise_fma(ww, (wp_v2){c11[set][s].x, c11[set][s].y},
                             __builtin_elementwise_fma(wz, (wp_v2){c10[set][s].x, c10[set][s].y},
                             __builtin_elementwise_fma(wy, (wp_v2){c01[set][s].x, c01[set][s].y}, wx * (wp_v2){c00[set][s].x, c00[set][s].y})));
            const wp_v2 hi = __builtin_elementwise_fma(ww, (wp_v2){c11[set][s].z, c11[set][s].w},
                             __builtin_elementwise_fma(wz, (wp_v2){c10[set][s].z, c10[set][s].w},
                             __builtin_elementwise_fma(wy, (wp_v2){c01[set][s].z, c01[set][s].w}, wx * (wp_v2){c00[set][s].z, c00[set][s].w})));
            *reinterpret_cast<float4*>(Aw + (pq + 16 * s) * IG_LDK + q * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    // weight tile of phase p: rows n0 .. n0+BN, k = tap*C + c0 .. +63
    auto load_bt = [&](int p) __attribute__((always_inline)) {
        const int tap = p / CPT, kb = tap * C + (p - tap * CPT) * WP_KC;
#pragma unroll
        for (int s = 0; s < G::B_SLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            br[s] = ig_ldg4(a.w + (size_t)(n0 + (idx >> 4)) * a.K + kb + (idx & 15) * 4);
        }
    };
    auto store_bt = [&](float* Bs) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < G::B_SLOTS; ++s) {
            const int idx = tid + s * IG_THREADS;
            *reinterpret_cast<float4*>(Bs + (idx >> 4) * WP_LDB + (idx & 15) * 4) = br[s];
        }
    };
    // MFMA fragments, two sets (k-step parity): set (s+1)&1 is read from LDS during the MFMA block of k-step s
    float4 af[2][2], bf[2][2][TN];
    auto read_frags = [&](int set, const float* Bt, int kk) __attribute__((always_inline)) {
        const float* Ar = As + set * 32 * IG_LDK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            af[set][h] = *reinterpret_cast<const float4*>(Ar + il * IG_LDK + h * 8 + g * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[set][h][j] = *reinterpret_cast<const float4*>(Bt + (j * 32 + il) * WP_LDB + kk * IG_BK + h * 8 + g * 4);
        }
    };

    gather(0, 0);
    gather(nk > 1 ? 1 : 0, 1);
    load_bt(0);
    blend_store(0);
    store_bt(smem);
    __syncthreads();
    read_frags(0, smem, 0);

    constexpr int KPP = WP_KC / IG_BK;               // k-steps per phase (4)
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        const float* Bc = smem + (p & 1) * G::B_FLOATS;
        float* Bn = smem + ((p + 1) & 1) * G::B_FLOATS;
        const int pn = p + 1 < NP ? p + 1 : NP - 1;
#pragma unroll
        for (int kk = 0; kk < KPP; ++kk) {
            const int ks = p * KPP + kk;
            const int cur = kk & 1, nxt = cur ^ 1;               // KPP is even: the k-step's parity is kk's
            const int ks2 = ks + 2 < nk ? ks + 2 : nk - 1;       // past the end: re-gather the last k-step (never used)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int grp = h * TN + j;                  // TN = 2: four groups of four MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp == 1) {                              // after 4 MFMAs: the gathers of k-step s+2 (+ once per phase the next weight tile)
                        gather(ks2, cur);
                        if (kk == 0) load_bt(pn);
                    }
                    if (grp == 2) blend_store(nxt);              // after 8: k-step s+1's gathers (issued one k-step ago) -> blended A rows
                    if (grp == 3) {                              // after 12: k-step s+1's fragments; at a phase end the next weight tile first
                        if (kk == KPP - 1) {
                            store_bt(Bn);
                            __syncthreads();
                            read_frags(nxt, Bn, 0);
                        } else read_frags(nxt, Bc, kk + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0][j] = ig_mfma<32>(af[cur][h].x, bf[cur][h][j].x, acc[0][j]);
                    acc[0][j] = ig_mfma<32>(af[cur][h].y, bf[cur][h][j].y, acc[0][j]);
                    acc[0][j] = ig_mfma<32>(af[cur][h].z, bf[cur][h][j].z, acc[0][j]);
                    acc[0][j] = ig_mfma<32>(af[cur][h].w, bf[cur][h][j].w, acc[0][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    ig_epilogue<T, 128, BN, 32>(a, smem, m0, n0, wm0, 0, lane, tid, acc);
}

template <int BN>
static int launch_dcn_wp(const ConvArgs& a, hipStream_t s)
{
    using T = IgTile<128, BN, 4, 1, 32>;
    using G = WpGeo<BN>;
    auto kern = dcn_wp_kernel<BN>;
    const int main_bytes = G::MAIN_FLOATS * 4;
    const int epi = a.outNCHW ? T::EPI_BYTES : T::EPV_BYTES;
    const int smem = epi > main_bytes ? epi : main_bytes;
    static CpLdsGuard guard;
    if (smem > 64 * 1024 && guard.need(smem))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int grid = cp_cdiv(a.M, G::BM) * (a.ldw / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(IG_THREADS), smem, s, a);
    cp_note_kernel("dcn_wp_kernel<%d>", BN);
    return 0;
}

// -1: shape not eligible (C % 64, ldw % BN, taps)
int cp_launch_dcn_wp(const ConvArgs& a, hipStream_t s, int bn)
{
    if (a.srcC[0] % WP_KC != 0 || a.kh * a.kw > WP_TAPS || a.ldw % bn != 0) return -1;
    if (bn == 64) return launch_dcn_wp<64>(a, s);
    return -1;
}
