// Heatmap decode for CenterNet multi-person pose on gfx950.
//
// Replaces the reference's ~60-launch torch expression graph
//   lib/models/decode.py:10-16   (_nms: 3x3 max-pool equality)
//   lib/models/decode.py:87-115  (_topk_channel / _topk)
//   lib/models/utils.py:11-25    (_transpose_and_gather_feat: whole-map permute + gather)
//   lib/models/decode.py:235-308 (multi_pose_decode)
// with two kernels:
//   K6 nms_topk_kernel   : one 1024-thread workgroup per (image, heat-map plane).  NMS'd values
//                          become order-preserving uint32 keys in LDS; a 4-pass 8-bit radix
//                          select finds the K-th largest key; survivors are compacted and
//                          bitonic-sorted.  Order = (value desc, flat index asc).
//                          Planes above 32768 keys (FIX_RES = false inputs, TEST_SCALES > 1: the reference's
//                          torch.topk has no size limit, decode.py:87-115) are streamed through LDS in equal
//                          chunks: each chunk's top-K (same order rule, global flat indices) is merged into the
//                          running top-K by a 2P-wide bitonic sort -- the union of per-chunk top-Ks contains the
//                          plane's top-K, so the result is the same (value desc, index asc) list.
//   K7 pose_assign_kernel: one workgroup per (image, joint): gathers hps/reg/wh/hp_offset
//                          straight from the NCHW maps at the K peaks (no whole-map transpose),
//                          does the K x K nearest-candidate search from LDS, applies the
//                          reference's acceptance rules and packs dets[B,K,5+3J].
// All float arithmetic is done in the reference's operation order with FP contraction disabled
// (this file is compiled with -ffp-contract=off), so on tie-free inputs the output is
// bit-identical to the reference CPU path.  HBM-bound integer/compare work: no MFMA here.
#include "common.h"

#define TK_THREADS 1024
#define TK_WAVES (TK_THREADS / CP_WAVE)
#define TK_MAX_ELEMS 32768   // keys live in LDS: 128 KiB of the CU's 160 KiB
#define TK_MAX_K 256
#define TK_EPT 16           // elements per thread of the LDS-resident NMS (16 * 1024 = a 128 x 128 plane)

__device__ __forceinline__ uint32_t f2key(float f)
{
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0;  // -0.0 == +0.0 for torch.topk
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------
// K6: NMS + top-K of one plane group (centre: cat planes of H*W flattened; joints: one plane)
// ------------------------------------------------------------------------------------------
// CHUNKED = false is the LDS-resident case (the 128 x 128 maps of the 512 x 512 configurations): one pass, the chunk loop and
// its extra barrier compile away.  __launch_bounds__(1024, 8): TWO 16-wave blocks per CU (8 waves per SIMD) need <= 64 VGPRs
// AND <= 80 SGPRs per wave; at 91 SGPRs only seven waves per SIMD are admitted, i.e. one block per CU, and the 288 blocks of
// a 16-image batch take two rounds on 256 CUs (157 us instead of 99 us, rocprofv3).
template <bool CHUNKED>
__global__ __launch_bounds__(TK_THREADS, 8) void nms_topk_kernel(
    const float* __restrict__ heat, const float* __restrict__ hm_hp, int cat, int J, int H, int W,
    int K, int P /* pow2 >= K */, int wsh /* log2 W when W and H*W are powers of two, else -1 */, int nmax /* LDS key slots, multiple of 4 */, int chunk /* keys per pass through LDS, <= nmax */,
    float* __restrict__ out_scores, int* __restrict__ out_inds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ppb = 1 + J;
    const int b = blockIdx.x / ppb, pl = blockIdx.x % ppb;
    const int HW = H * W;
    const float* src;
    int n;
    if (pl == 0) { src = heat + (size_t)b * cat * HW; n = cat * HW; }
    else { src = hm_hp + ((size_t)b * J + (pl - 1)) * HW; n = HW; }

    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                    // [nmax]
    unsigned long long* cand0 = reinterpret_cast<unsigned long long*>(smem + (size_t)nmax * 4);  // [2 * TK_MAX_K]: running top-P | this chunk's
    uint32_t* hist = reinterpret_cast<uint32_t*>(cand0 + 2 * TK_MAX_K);    // [256]
    uint32_t* wsum = hist + 256;                                           // [TK_WAVES]
    uint32_t* sctl = wsum + TK_WAVES;                                      // [4] digit, remaining, cnt_gt

    const int ntot = n;
    for (int c0 = 0; c0 < (CHUNKED ? ntot : 1); c0 += chunk) {
    // keys of flat indices [c0, c0 + n) live in LDS; candidates carry the global index c0 + e
    if (CHUNKED) n = min(chunk, ntot - c0);
    unsigned long long* cand = (!CHUNKED || c0 == 0) ? cand0 : cand0 + P;
    if (CHUNKED) __syncthreads();      // previous chunk's merge has finished with keys / cand
    // ---- phase 1: 3x3 NMS (decode.py:10-16; -inf padding == skip out-of-range) -> keys
    const bool fast = !CHUNKED && n <= TK_EPT * TK_THREADS;       // block-uniform: the plane group's keys also fit the threads' registers
    uint32_t kreg[TK_EPT];
    if (fast) {
        // LDS-resident plane group: the raw floats are staged in the key buffer once (coalesced), every thread takes the 3x3 maxima of its
        // <= 16 elements from LDS into registers, and the keys replace the floats after a barrier.  (The direct form below does nine
        // dependent global loads and two run-time integer divisions per element: ~a quarter of this kernel's 99 us at 128x128.)
        float* raw = reinterpret_cast<float*>(keys);
        if ((n & 3) == 0 && ((reinterpret_cast<size_t>(src)) & 15) == 0) {
            for (int e = tid * 4; e < n; e += TK_THREADS * 4) *reinterpret_cast<float4*>(raw + e) = *reinterpret_cast<const float4*>(src + e);
        } else {
            for (int e = tid; e < n; e += TK_THREADS) raw[e] = src[e];
        }
        __syncthreads();
        const bool pow2 = wsh >= 0;
#pragma unroll
        for (int q = 0; q < TK_EPT; ++q) {
            const int e = tid + q * TK_THREADS;
            kreg[q] = 0;
            if (e < n) {
                int p, y, x;
                if (pow2) { p = e & (HW - 1); y = p >> wsh; x = p & (W - 1); }       // H*W and W powers of two (host-checked)
                else { const int c = e / HW; p = e - c * HW; y = p / W; x = p - y * W; }
                const float* pp = raw + (e - p);
                const float v = pp[p];
                float m = v;
                const int y0 = y > 0 ? y - 1 : y, y1 = y < H - 1 ? y + 1 : y;
                const int x0 = x > 0 ? x - 1 : x, x1 = x < W - 1 ? x + 1 : x;
                // clamped neighbours: re-reading the centre row / column never changes a maximum
                const float* r0 = pp + y0 * W; const float* r1 = pp + y * W; const float* r2 = pp + y1 * W;
                m = fmaxf(m, fmaxf(fmaxf(r0[x0], r0[x]), r0[x1]));
                m = fmaxf(m, fmaxf(r1[x0], r1[x1]));
                m = fmaxf(m, fmaxf(fmaxf(r2[x0], r2[x]), r2[x1]));
                const float o = (m == v) ? v : v * 0.0f;   // heat * keep
                kreg[q] = f2key(o);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TK_EPT; ++q) {
            const int e = tid + q * TK_THREADS;
            if (e < n) keys[e] = kreg[q];
        }
    } else
    for (int e = tid; e < n; e += TK_THREADS) {
        const int c = (c0 + e) / HW, p = (c0 + e) - c * HW;
        const int y = p / W, x = p - y * W;
        const float* pp = src + (size_t)c * HW;
        const float v = pp[p];
        float m = v;
        const int y0 = y > 0 ? y - 1 : y, y1 = y < H - 1 ? y + 1 : y;
        const int x0 = x > 0 ? x - 1 : x, x1 = x < W - 1 ? x + 1 : x;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) m = fmaxf(m, pp[yy * W + xx]);
        const float o = (m == v) ? v : v * 0.0f;   // heat * keep
        keys[e] = f2key(o);
    }
    for (int i = tid; i < P; i += TK_THREADS) cand[i] = 0ull;
    __syncthreads();

    // ---- phase 2: radix select of the K-th largest key (MSB first, 8 bits per pass)
    uint32_t prefix = 0, pmask = 0;
    int remaining = min(K, n);
    const int Kc = remaining;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        // wave-aggregated histogram update: the digit of the first active lane is counted with ONE LDS atomic for all lanes that share it
        // (NMS zeroes 8/9 of a map and post-sigmoid values share their top byte), the other lanes add theirs one by one
        auto hist_add = [&](bool act, uint32_t d) {
            const unsigned long long am = __ballot(act);
            if (am) {                                                         // wave-uniform
                const int first = __ffsll((long long)am) - 1;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
                const unsigned long long same = __ballot(act && d == d0);
                if (lane == first) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
                if (act && d != d0) atomicAdd(&hist[d], 1u);
            }
        };
        if (fast) {            // keys from registers (round 4: the four passes re-read all keys from LDS: 34 of the kernel's 56 us)
#pragma unroll
            for (int q = 0; q < TK_EPT; ++q) {
                const uint32_t k = kreg[q];
                hist_add(tid + q * TK_THREADS < n && (k & pmask) == prefix, (k >> shift) & 255u);
            }
        } else
        for (int base = 0; base < n; base += TK_THREADS) {
            const int e = base + tid;
            bool act = false;
            uint32_t d = 0;
            if (e < n) {
                const uint32_t k = keys[e];
                act = ((k & pmask) == prefix);
                d = (k >> shift) & 255u;
            }
            hist_add(act, d);
        }
        __syncthreads();
        if (wid == 0) {
            // lane l owns bins 255-4l .. 252-4l (descending order)
            uint32_t h[4], s = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { h[i] = hist[255 - 4 * lane - i]; s += h[i]; }
            uint32_t inc = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
            const uint32_t exc = inc - s;
            if (exc < (uint32_t)remaining && (uint32_t)remaining <= inc) {
                uint32_t acc = exc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (acc < (uint32_t)remaining && (uint32_t)remaining <= acc + h[i]) {
                        sctl[0] = 255 - 4 * lane - i;
                        sctl[1] = remaining - acc;
                    }
                    acc += h[i];
                }
            }
        }
        __syncthreads();
        prefix |= sctl[0] << shift;
        pmask |= 255u << shift;
        remaining = (int)sctl[1];
        __syncthreads();
    }
    const uint32_t T = prefix;          // K-th largest key
    const int need_eq = remaining;      // how many keys == T to take (lowest indices first)
    const int cnt_gt = Kc - need_eq;

    // ---- phase 3: compaction.  key > T: any slot in [0,cnt_gt).  key == T: index-ordered rank.
    if (tid == 0) sctl[2] = 0;
    // each wave owns a contiguous index range so that equal keys are ranked by index
    const int per_wave = ((n + TK_WAVES - 1) / TK_WAVES + 63) & ~63;
    const int wbeg = wid * per_wave, wend = min(n, wbeg + per_wave);
    uint32_t my_eq = 0;
    for (int base = wbeg; base < wend; base += 64) {
        const int e = base + lane;
        const bool eq = (e < wend) && (keys[e] == T);
        my_eq += (uint32_t)__popcll(__ballot(eq));
    }
    if (lane == 0) wsum[wid] = my_eq;
    __syncthreads();
    uint32_t eq_base = 0;
    for (int w = 0; w < wid; ++w) eq_base += wsum[w];
    for (int base = wbeg; base < wend; base += 64) {
        const int e = base + lane;
        uint32_t k = 0;
        bool gt = false, eq = false;
        if (e < wend) { k = keys[e]; gt = k > T; eq = (k == T); }
        const unsigned long long em = __ballot(eq);
        if (eq) {
            const uint32_t rank = eq_base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
            if (rank < (uint32_t)need_eq)
                cand[cnt_gt + rank] = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c0 + e));
        }
        eq_base += (uint32_t)__popcll(em);
        if (gt) {
            const uint32_t slot = atomicAdd(&sctl[2], 1u);
            cand[slot] = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c0 + e));
        }
    }
    __syncthreads();

    // ---- phase 4: bitonic sort, descending (value desc, index asc): the first chunk's P candidates, afterwards the
    // running top-P together with this chunk's P candidates (the better half stays in cand0[0, P))
    if (!CHUNKED) {
        // P <= 256 candidates: ONE wave sorts them in registers (element i = r * 64 + lane; partners at distance < 64 through
        // ds_bpermute, at 64 / 128 in the lane's other registers) -- no block barrier in any of the 28 (P = 128) compare-exchange steps
        if (wid == 0) {
            unsigned long long v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (r * 64 + lane < P) ? cand0[r * 64 + lane] : 0ull;
            const int S = P < 64 ? 64 : P;
            for (int k2 = 2; k2 <= S; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    if (j >= 64) {
                        const int jr = j >> 6;                                 // 1 or 2 (uniform)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rp = r ^ jr;
                            if (rp < r) continue;
                            // pair (r, rp), r < rp: element r*64+lane is the lower index
                            const unsigned long long a = v[r], c = jr == 1 ? v[r ^ 1] : v[r ^ 2];
                            const bool desc = (((r * 64 + lane) & k2) == 0);
                            const bool sw = desc ? (a < c) : (a > c);
                            if (sw) { v[r] = c; if (jr == 1) v[r ^ 1] = a; else v[r ^ 2] = a; }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (r * 64 >= S) continue;
                            const int i = r * 64 + lane;
                            const uint32_t lo = __shfl_xor((uint32_t)v[r], j), hi = __shfl_xor((uint32_t)(v[r] >> 32), j);
                            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                            const bool want_max = (((i & k2) == 0) == ((i & j) == 0));
                            v[r] = want_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = r * 64 + lane;
                if (i < K) {
                    out_scores[(size_t)blockIdx.x * K + i] = key2f((uint32_t)(v[r] >> 32));
                    out_inds[(size_t)blockIdx.x * K + i] = (int)(0xFFFFFFFFu - (uint32_t)(v[r] & 0xFFFFFFFFull));
                }
            }
        }
        return;
    }
    const int S = (c0 == 0) ? P : 2 * P;
    for (int k2 = 2; k2 <= S; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < S; i += TK_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = cand0[i], c = cand0[ixj];
                    const bool desc = ((i & k2) == 0);
                    if (desc ? (a < c) : (a > c)) { cand0[i] = c; cand0[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    }   // chunk loop
    for (int i = tid; i < K; i += TK_THREADS) {
        const unsigned long long c = cand0[i];
        out_scores[(size_t)blockIdx.x * K + i] = key2f((uint32_t)(c >> 32));
        out_inds[(size_t)blockIdx.x * K + i] = (int)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull));
    }
}

// ------------------------------------------------------------------------------------------
// K7: gather + keypoint-to-person assignment + pack  (decode.py:244-307)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_assign_kernel(
    const float* __restrict__ wh, const float* __restrict__ kps, const float* __restrict__ reg,
    const float* __restrict__ hp_offset, const float* __restrict__ scores,
    const int* __restrict__ inds, int J, int H, int W, int K, float* __restrict__ dets)
{
    __shared__ float s_hx[TK_MAX_K], s_hy[TK_MAX_K], s_hs[TK_MAX_K];
    const int b = blockIdx.x / J, j = blockIdx.x % J;
    const int HW = H * W, ppb = 1 + J, D = 5 + 3 * J;
    const float* jsc = scores + ((size_t)b * ppb + 1 + j) * K;
    const int* jin = inds + ((size_t)b * ppb + 1 + j) * K;

    for (int m = threadIdx.x; m < K; m += blockDim.x) {
        const float s = jsc[m];
        const int ind = jin[m];
        float hy = (float)(ind / W), hx = (float)(ind % W);       // decode.py:92-93
        if (hp_offset) {                                          // :272-277 (2-ch map shared by joints)
            hx = hx + hp_offset[((size_t)b * 2 + 0) * HW + ind];
            hy = hy + hp_offset[((size_t)b * 2 + 1) * HW + ind];
        } else { hx = hx + 0.5f; hy = hy + 0.5f; }
        const float mk = (s > 0.1f) ? 1.0f : 0.0f;               // :282
        s_hs[m] = (1.0f - mk) * -1.0f + mk * s;                   // :283
        s_hy[m] = (1.0f - mk) * -10000.0f + mk * hy;              // :284
        s_hx[m] = (1.0f - mk) * -10000.0f + mk * hx;              // :285
    }
    __syncthreads();

    const float* csc = scores + (size_t)b * ppb * K;
    const int* cin = inds + (size_t)b * ppb * K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int p = cin[k] % HW;                                // decode.py:104 (class plane dropped)
        const float ys = (float)(p / W), xs = (float)(p % W);
        const float kx = kps[((size_t)b * 2 * J + 2 * j) * HW + p] + xs;       // :246 integer peak coords
        const float ky = kps[((size_t)b * 2 * J + 2 * j + 1) * HW + p] + ys;   // :247
        float cx, cy;
        if (reg) { cx = xs + reg[((size_t)b * 2 + 0) * HW + p]; cy = ys + reg[((size_t)b * 2 + 1) * HW + p]; }
        else { cx = xs + 0.5f; cy = ys + 0.5f; }
        const float w = wh[((size_t)b * 2 + 0) * HW + p], h = wh[((size_t)b * 2 + 1) * HW + p];
        const float l = cx - w / 2.0f, t = cy - h / 2.0f, r = cx + w / 2.0f, bt = cy + h / 2.0f;  // :261-264

        float best = 0.f;
        int bi = 0;
        for (int m = 0; m < K; ++m) {                             // :286-289, first minimum
            const float dx = kx - s_hx[m], dy = ky - s_hy[m];
            const float d = __fsqrt_rn(dx * dx + dy * dy);
            if (m == 0 || d < best) { best = d; bi = m; }
        }
        const float ss = s_hs[bi], sx = s_hx[bi], sy = s_hy[bi];
        const bool rej = (sx < l) || (sx > r) || (sy < t) || (sy > bt) || (ss < 0.1f) ||
                         (best > fmaxf(bt - t, r - l) * 0.3f);    // :300-302
        const float mk = rej ? 1.0f : 0.0f;
        float* row = dets + ((size_t)b * K + k) * D;
        row[5 + 2 * j] = (1.0f - mk) * sx + mk * kx;              // :304
        row[5 + 2 * j + 1] = (1.0f - mk) * sy + mk * ky;
        row[5 + 2 * J + j] = ss;                                  // :307 (score emitted even if rejected)
        if (j == 0) { row[0] = l; row[1] = t; row[2] = r; row[3] = bt; row[4] = csc[k]; }
    }
}

extern "C" int cp_decode_workspace_bytes(int B, int J, int K, size_t* scores_bytes, size_t* inds_bytes)
{
    if (scores_bytes) *scores_bytes = (size_t)B * (1 + J) * K * sizeof(float);
    if (inds_bytes) *inds_bytes = (size_t)B * (1 + J) * K * sizeof(int);
    return 0;
}

// The decode in two halves, so that a launch schedule (engine.Engine / the plan runtime) can put them into its graph as two
// nodes: the peak extraction needs only hm and hm_hp and overlaps the remaining head convolutions on the other capture stream;
// the assignment needs all six heads.  cp_multi_pose_decode_f32 = both, back to back.
extern "C" int cp_decode_topk_f32(const float* heat, const float* hm_hp, int B, int cat, int J, int H, int W, int K,
                                  float* ws_scores, int* ws_inds, void* stream)
{
    CP_CHECK_ARG(heat && ws_scores && ws_inds, "multi_pose_decode: null pointer");
    CP_CHECK_ARG(hm_hp != nullptr,
                 "multi_pose_decode: hm_hp is required (the reference raises NameError without it, decode.py:307)");
    CP_CHECK_ARG(B > 0 && cat > 0 && J > 0 && H > 0 && W > 0, "multi_pose_decode: bad shape");
    CP_CHECK_ARG(K > 0 && K <= TK_MAX_K, "multi_pose_decode: K=%d out of range (1..%d)", K, TK_MAX_K);
    CP_CHECK_ARG((long long)cat * H * W < (1ll << 31), "multi_pose_decode: cat*H*W=%lld does not fit a 32-bit flat index",
                 (long long)cat * H * W);
    CP_CHECK_ARG(H * W >= K, "multi_pose_decode: K=%d larger than the map (%d)", K, H * W);
    int P = 1;
    while (P < K) P <<= 1;
    // planes up to TK_MAX_ELEMS keys are LDS-resident in one piece; larger ones stream through LDS in equal chunks
    const int nbig = cat * H * W;
    const int nchunks = cp_cdiv(nbig, TK_MAX_ELEMS);
    const int chunk = (cp_cdiv(nbig, nchunks) + 3) & ~3;
    const int nmax = chunk;
    const size_t lds = (size_t)nmax * 4 + 2 * TK_MAX_K * 8 + 256 * 4 + TK_WAVES * 4 + 16;
    static CpLdsGuard lds_reserved[2];        // per (instantiation, device)
    const int ck = nchunks > 1;
    {
        const hipError_t e = lds_reserved[ck].ensure(ck ? (const void*)nms_topk_kernel<true> : (const void*)nms_topk_kernel<false>, (int)lds);
        if (e != hipSuccess) { cp_set_error("decode: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return 2; }
    }
    hipStream_t s = (hipStream_t)stream;
    int wsh = -1;
    if ((W & (W - 1)) == 0 && ((H * W) & (H * W - 1)) == 0) { wsh = 0; while ((1 << wsh) < W) ++wsh; }
    if (ck) hipLaunchKernelGGL(nms_topk_kernel<true>, dim3(B * (1 + J)), dim3(TK_THREADS), lds, s, heat, hm_hp, cat, J, H, W, K, P, wsh,
                               nmax, chunk, ws_scores, ws_inds);
    else hipLaunchKernelGGL(nms_topk_kernel<false>, dim3(B * (1 + J)), dim3(TK_THREADS), lds, s, heat, hm_hp, cat, J, H, W, K, P, wsh,
                            nmax, chunk, ws_scores, ws_inds);
    CP_CHECK_LAUNCH("nms_topk_kernel");
    cp_note_kernel(ck ? "nms_topk_kernel<true>" : "nms_topk_kernel<false>");
    return 0;
}

extern "C" int cp_decode_assign_f32(const float* wh, const float* kps, const float* reg, const float* hp_offset,
                                    const float* ws_scores, const int* ws_inds, int B, int J, int H, int W, int K, float* dets,
                                    void* stream)
{
    CP_CHECK_ARG(wh && kps && dets && ws_scores && ws_inds, "multi_pose_decode: null pointer");
    CP_CHECK_ARG(B > 0 && J > 0 && H > 0 && W > 0 && K > 0 && K <= TK_MAX_K, "multi_pose_decode: bad shape");
    hipLaunchKernelGGL(pose_assign_kernel, dim3(B * J), dim3(128), 0, (hipStream_t)stream, wh, kps, reg, hp_offset, ws_scores, ws_inds, J,
                       H, W, K, dets);
    CP_CHECK_LAUNCH("pose_assign_kernel");
    cp_note_kernel("pose_assign_kernel");
    return 0;
}

extern "C" int cp_multi_pose_decode_f32(const float* heat, const float* wh, const float* kps,
                                        const float* reg, const float* hm_hp, const float* hp_offset,
                                        int B, int cat, int J, int H, int W, int K, float* dets,
                                        float* ws_scores, int* ws_inds, void* stream)
{
    CP_CHECK_ARG(heat && wh && kps && dets && ws_scores && ws_inds, "multi_pose_decode: null pointer");
    if (int rc = cp_decode_topk_f32(heat, hm_hp, B, cat, J, H, W, K, ws_scores, ws_inds, stream)) return rc;
    return cp_decode_assign_f32(wh, kps, reg, hp_offset, ws_scores, ws_inds, B, J, H, W, K, dets, stream);
}
