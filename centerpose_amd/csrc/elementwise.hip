// Bandwidth-bound NHWC helpers around the GEMM kernels (float4 over channels, grid-stride).
//   max-pool                        nn.MaxPool2d       pose_dla_dcn.py:197-198, msra_resnet.py:123
//   depthwise bilinear deconv + add IDAUp up_* + add   pose_dla_dcn.py:360-377 (groups=o, k=2f, s=f, p=f/2)
//   nearest-upsample sum (+ReLU)    HRNet fuse layers  pose_higher_hrnet.py:217-235
//   layout transforms               NCHW <-> NHWC (drop-in dcn_v2_forward, tests)
//   flip-test merge                 multi_pose.py:45-53 + models/utils.py:27-47 (device side, no numpy bounce)
//   depthwise conv + BN + act       MobileNetV3 Block.conv2 / ShuffleNetV2 banch*  mobilenetv3.py:119-121, shufflenetv2_dcn.py:67-88
//   squeeze-excite                  SeModule: global average pool, x * se(x) (+ shortcut)  mobilenetv3.py:99-113,141-143
//   channel shuffle of a concat     ShuffleNetV2 InvertedResidual.forward            shufflenetv2_dcn.py:28-42,94-104
#include <cstdlib>
#include "common.h"

#define EW_THREADS 256
static inline int ew_grid(long long n) { long long g = (n + EW_THREADS - 1) / EW_THREADS; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

// Row-major launch geometry for the pixel kernels: blockIdx.y walks the (image, output row) pairs, threads walk the
// (column, channel quad) pairs of one row.  All index math is 32-bit; the one division per element (by the uniform
// C/4) is a multiply-high with a host-made magic number (exact for e * C4 < 2^32).  The grid-stride 64-bit % and /
// this replaces cost ~200 instructions per element and made these "bandwidth-bound" helpers ALU-bound.
struct EwRow { int rows, rowElems, C4; unsigned mC4; };
static inline EwRow ew_row(int rows, int Wo, int C4)
{
    EwRow r; r.rows = rows; r.rowElems = Wo * C4; r.C4 = C4;
    r.mC4 = C4 <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)C4 - 1) / (unsigned)C4);
    return r;
}
static inline dim3 ew_row_grid(const EwRow& r)
{
    return dim3((unsigned)((r.rowElems + EW_THREADS - 1) / EW_THREADS), (unsigned)(r.rows > 65535 ? 65535 : r.rows));
}
__device__ __forceinline__ void ew_split(const EwRow& r, int e, int& x, int& c4)
{
    x = r.C4 == 1 ? e : (int)__umulhi((unsigned)e, r.mC4);
    c4 = e - x * r.C4;
}

__global__ void maxpool_nhwc_kernel(const float* __restrict__ in, int inLd, float* __restrict__ out, int outLd, EwRow r,
                                    int H, int W, int Ho, int Wo, int k, int s, int p)
{
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int ox, c4;
    ew_split(r, e, ox, c4);
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {
        const int b = row / Ho, oy = row - b * Ho;                     // uniform
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - p + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - p + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(b * H + iy) * W + ix) * inLd + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(out + ((size_t)row * Wo + ox) * outLd + c4 * 4) = m;
    }
}

extern "C" int cp_maxpool2d_nhwc_f32(const float* in, int inLd, float* out, int outLd, int B, int H, int W, int C, int k,
                                     int s, int p, void* stream)
{
    CP_CHECK_ARG(in && out && C % 4 == 0 && inLd % 4 == 0 && outLd % 4 == 0, "maxpool: C, ld must be multiples of 4");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    CP_CHECK_ARG((long long)Wo * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * Ho < (1ll << 31), "maxpool: row too large");
    const EwRow r = ew_row(B * Ho, Wo, C / 4);
    hipLaunchKernelGGL(maxpool_nhwc_kernel, ew_row_grid(r), dim3(EW_THREADS), 0, (hipStream_t)stream, in, inLd, out, outLd, r,
                       H, W, Ho, Wo, k, s, p);
    cp_note_kernel("maxpool_nhwc_kernel");
    CP_CHECK_LAUNCH("maxpool_nhwc_kernel");
    return 0;
}

// out[b,oy,ox,c] = add[b,oy,ox,c] + sum_{ky,kx} in[b,iy,ix,c] * w[(ky*k+kx)][c],  oy = iy*f - p + ky
// (`add` and `out` are NOT __restrict__: the engine may pass the same storage for both -- every thread reads exactly the `add`
// elements it then writes, so in-place is well defined)
__global__ void dw_deconv_add_kernel(const float* __restrict__ in, int inLd, const float* __restrict__ w,
                                     const float* add, int addLd, float* out, int outLd, EwRow r,
                                     int H, int W, int f, int p)
{
    const int k = 2 * f, Ho = H * f, Wo = W * f, C = r.C4 * 4;
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int ox, c4;
    ew_split(r, e, ox, c4);
    const int rx = (ox + p) % f;
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {
        const int b = row / Ho, oy = row - b * Ho;                     // uniform
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // ky = oy + p - iy*f must lie in [0, k): two candidate input rows / columns
        const int ry = (oy + p) % f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            const int ky = ry + ty * f, iy = (oy + p - ky) / f;
            if (oy + p - ky < 0 || iy >= H) continue;
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int kx = rx + tx * f, ix = (ox + p - kx) / f;
                if (ox + p - kx < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(b * H + iy) * W + ix) * inLd + c4 * 4);
                const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)(ky * k + kx) * C + c4 * 4);
                acc.x += v.x * ww.x; acc.y += v.y * ww.y; acc.z += v.z * ww.z; acc.w += v.w * ww.w;
            }
        }
        const size_t opix = (size_t)row * Wo + ox;
        if (add) {
            const float4 rr = *reinterpret_cast<const float4*>(add + opix * addLd + c4 * 4);
            acc.x += rr.x; acc.y += rr.y; acc.z += rr.z; acc.w += rr.w;
        }
        *reinterpret_cast<float4*>(out + opix * outLd + c4 * 4) = acc;
    }
}

// f = 2 (k = 4, s = 2, p = 1: seven of DLA-34's eight IDAUp up-samplings): one thread per INPUT pixel column and channel quad computes the
// 2 x 2 output patch of every input pixel it walks -- the 3 x 3 input neighbourhood is loaded once for four outputs (2.25 loads per output
// instead of 4) and the 16 tap weights stay in registers over the row loop (the generic kernel re-loads 4 per output): 4.25 memory
// instructions per output float4 instead of 10 (round 5: the generic kernel ran at 3.5 TB/s, its 9 L1 loads per 16 stored bytes, not HBM,
// being the limit).  Same taps in the same order per output as the generic kernel (ty, tx = 0, 1: ky = ry + 2 ty, kx = rx + 2 tx), out-of-range
// taps contribute fma(0, w, acc) = acc: bit-identical for finite weights.
__global__ __launch_bounds__(EW_THREADS) void dw_deconv2_add_kernel(const float* __restrict__ in, int inLd, const float* __restrict__ w,
                                                                   const float* add, int addLd, float* out,
                                                                   int outLd, EwRow r, int H, int W)
{
    const int C = r.C4 * 4, Wo = 2 * W;
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int ix, c4;
    ew_split(r, e, ix, c4);
    float4 wt[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) wt[t] = *reinterpret_cast<const float4*>(w + (size_t)t * C + c4 * 4);
    const bool lok = ix > 0, rok = ix + 1 < W;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {           // row = (b, iy): uniform
        const int b = row / H, iy = row - b * H;
        const float* base = in + ((size_t)row * W + ix) * inLd + c4 * 4;
        const bool tok = iy > 0, bok = iy + 1 < H;
        const size_t rs = (size_t)W * inLd;
        float4 v[3][3];                                                     // v[dy + 1][dx + 1] = in[iy + dy][ix + dx]
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const bool yok = dy == 0 || (dy < 0 ? tok : bok);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const bool ok = yok && (dx == 0 || (dx < 0 ? lok : rok));
                v[dy + 1][dx + 1] = ok ? *reinterpret_cast<const float4*>(base + dy * (long long)rs + dx * inLd) : z;
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                // output (2 iy + a, 2 ix + bq): taps (ky, input row) = a == 0 ? (1, iy), (3, iy - 1) : (0, iy + 1), (2, iy); columns alike
                float4 acc = z;
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) {
                    const int ky = (a == 0 ? 1 : 0) + 2 * ty, dy = a == 0 ? -ty : 1 - ty;
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) {
                        const int kx = (bq == 0 ? 1 : 0) + 2 * tx, dx = bq == 0 ? -tx : 1 - tx;
                        const float4 vv = v[dy + 1][dx + 1], ww = wt[ky * 4 + kx];
                        acc.x += vv.x * ww.x; acc.y += vv.y * ww.y; acc.z += vv.z * ww.z; acc.w += vv.w * ww.w;
                    }
                }
                const size_t opix = ((size_t)(b * 2 * H + 2 * iy + a)) * Wo + 2 * ix + bq;
                if (add) {
                    const float4 rr = *reinterpret_cast<const float4*>(add + opix * addLd + c4 * 4);
                    acc.x += rr.x; acc.y += rr.y; acc.z += rr.z; acc.w += rr.w;
                }
                *reinterpret_cast<float4*>(out + opix * outLd + c4 * 4) = acc;
            }
    }
}

extern "C" int cp_dw_deconv_add_nhwc_f32(const float* in, int inLd, const float* w, const float* add, int addLd, float* out,
                                         int outLd, int B, int H, int W, int C, int f, void* stream)
{
    CP_CHECK_ARG(in && w && out && C % 4 == 0 && inLd % 4 == 0 && outLd % 4 == 0 && (!add || addLd % 4 == 0),
                 "dw_deconv_add: C, ld must be multiples of 4");
    CP_CHECK_ARG(f >= 1, "dw_deconv_add: f=%d", f);
    CP_CHECK_ARG((long long)W * f * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * H * f < (1ll << 31), "dw_deconv_add: row too large");
    // CP_DWDECONV2=0: the generic kernel for f = 2 as well (A/B switch: tests compare the two bit for bit; read per call -- launches
    // are recorded once per plan, not per step)
    const char* sw = getenv("CP_DWDECONV2");
    if (f == 2 && !(sw && sw[0] == '0') && (long long)W * (C / 4) * (C / 4) < (1ll << 32)) {
        const EwRow r2 = ew_row(B * H, W, C / 4);
        hipLaunchKernelGGL(dw_deconv2_add_kernel, ew_row_grid(r2), dim3(EW_THREADS), 0, (hipStream_t)stream, in, inLd, w, add, addLd, out,
                           outLd, r2, H, W);
        cp_note_kernel("dw_deconv2_add_kernel");
        CP_CHECK_LAUNCH("dw_deconv2_add_kernel");
        return 0;
    }
    const EwRow r = ew_row(B * H * f, W * f, C / 4);
    hipLaunchKernelGGL(dw_deconv_add_kernel, ew_row_grid(r), dim3(EW_THREADS), 0, (hipStream_t)stream, in, inLd, w, add,
                       addLd, out, outLd, r, H, W, f, f / 2);
    cp_note_kernel("dw_deconv_add_kernel");
    CP_CHECK_LAUNCH("dw_deconv_add_kernel");
    return 0;
}

// ---- second half of a split-K DCNv2 launch (dcn.hip, cp_dcn_desc.ksplit): sum the S raw partial-sum slices in a FIXED order
// (deterministic, unlike float atomics), then scale / shift (folded BN + bias) + activation, NHWC store of the first Cout channels
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long slice, int ldw4, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int act, float* __restrict__ out, int outLd, int Cout, long long total4)
{
    const long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x;          // float4 index into one slice [M][ldw]
    if (i >= total4) return;
    const long long m = i / ldw4;
    const int n = (int)(i - m * ldw4) * 4;
    if (n >= Cout) return;
    float4 acc = *reinterpret_cast<const float4*>(ws + i * 4);
    for (int s = 1; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)s * slice + i * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 sc = *reinterpret_cast<const float4*>(scale + n), sh = *reinterpret_cast<const float4*>(shift + n);
    acc.x = cp_act(acc.x * sc.x + sh.x, act); acc.y = cp_act(acc.y * sc.y + sh.y, act);
    acc.z = cp_act(acc.z * sc.z + sh.z, act); acc.w = cp_act(acc.w * sc.w + sh.w, act);
    *reinterpret_cast<float4*>(out + (size_t)m * outLd + n) = acc;
}

extern "C" int cp_splitk_reduce_f32(const float* ws, int splits, int M, int ldw, const float* scale, const float* shift, int act, float* out,
                                    int outLd, int Cout, void* stream)
{
    CP_CHECK_ARG(ws && scale && shift && out && splits >= 1 && M > 0, "splitk_reduce: bad arguments");
    CP_CHECK_ARG(ldw % 4 == 0 && Cout % 4 == 0 && outLd % 4 == 0 && Cout <= ldw && Cout <= outLd, "splitk_reduce: ldw, Cout, outLd must be multiples of 4 (ldw=%d Cout=%d outLd=%d)", ldw, Cout, outLd);
    const long long total4 = (long long)M * (ldw / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ws, splits, (long long)M * ldw, ldw / 4, scale, shift, act, out, outLd, Cout, total4);
    cp_note_kernel("splitk_reduce_kernel");
    CP_CHECK_LAUNCH("splitk_reduce_kernel");
    return 0;
}

// out = act( sum_i nearest_up(src_i, 2^sh_i) ), all NHWC with C channels; out is [B,H,W,C]
struct SumUpArgs { const float* src[4]; int ld[4]; int sh[4]; int n; };
__global__ void sum_up_kernel(SumUpArgs a, float* __restrict__ out, int outLd, EwRow r, int H, int W, int relu)
{
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int x, c4;
    ew_split(r, e, x, c4);
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {
        const int b = row / H, y = row - b * H;                        // uniform
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < a.n; ++j) {
            const int sh = a.sh[j], hs = H >> sh, ws = W >> sh;
            const float4 v = *reinterpret_cast<const float4*>(a.src[j] + ((size_t)(b * hs + (y >> sh)) * ws + (x >> sh)) * a.ld[j] + c4 * 4);
            if (j == 0) acc = v;
            else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        *reinterpret_cast<float4*>(out + ((size_t)row * W + x) * outLd + c4 * 4) = acc;
    }
}

extern "C" int cp_sum_up_nhwc_f32(int n, const float* const* src, const int* ld, const int* shift, float* out, int outLd,
                                  int B, int H, int W, int C, int relu, void* stream)
{
    CP_CHECK_ARG(n >= 1 && n <= 4 && src && ld && shift && out && C % 4 == 0, "sum_up: bad arguments");
    SumUpArgs a;
    for (int i = 0; i < 4; ++i) { a.src[i] = i < n ? src[i] : nullptr; a.ld[i] = i < n ? ld[i] : 0; a.sh[i] = i < n ? shift[i] : 0; }
    a.n = n;
    CP_CHECK_ARG((long long)W * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * H < (1ll << 31), "sum_up: row too large");
    const EwRow r = ew_row(B * H, W, C / 4);
    hipLaunchKernelGGL(sum_up_kernel, ew_row_grid(r), dim3(EW_THREADS), 0, (hipStream_t)stream, a, out, outLd, r, H, W, relu);
    cp_note_kernel("sum_up_kernel");
    CP_CHECK_LAUNCH("sum_up_kernel");
    return 0;
}

// Up to four INDEPENDENT up-sampling sums in one launch (round 6): an HRNet module ends with one such sum per output branch
// (pose_higher_hrnet.py:224-235: y_i = relu(sum_j fuse_ij(x_j))), 7-12 us each for 10-39 MB -- launch-bound one by one, and every one of
// them sits on the critical path between two modules.  Member k owns the blocks [first[k], first[k + 1]); a block is one 256-element
// segment of one output row; per element the same loads and the same order of additions as sum_up_kernel (bit-identical).
struct SumUpMember { SumUpArgs a; float* out; int outLd; EwRow r; int H, W, xblocks; };
struct SumUpGroup { SumUpMember m[4]; int first[5]; int n, relu; };
__global__ void sum_up_group_kernel(const SumUpGroup g)
{
    int t = blockIdx.x, k = 0;
    while (k + 1 < g.n && t >= g.first[k + 1]) ++k;                    // scalar
    const SumUpMember& m = g.m[k];
    t -= g.first[k];
    const int row = t / m.xblocks, e = (t - row * m.xblocks) * EW_THREADS + threadIdx.x;
    if (e >= m.r.rowElems) return;
    int x, c4;
    ew_split(m.r, e, x, c4);
    const int H = m.H, W = m.W;
    const int b = row / H, y = row - b * H;                              // uniform
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < m.a.n; ++j) {
        const int sh = m.a.sh[j], hs = H >> sh, ws = W >> sh;
        const float4 v = *reinterpret_cast<const float4*>(m.a.src[j] + ((size_t)(b * hs + (y >> sh)) * ws + (x >> sh)) * m.a.ld[j] + c4 * 4);
        if (j == 0) acc = v;
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (g.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<float4*>(m.out + ((size_t)row * W + x) * m.outLd + c4 * 4) = acc;
}

// n <= 4 members; src: 4 pointers per member (NULL beyond its nsrc); meta: 14 ints per member = nsrc, ld[4], shift[4], outLd, B, H, W, C
extern "C" int cp_sum_up_group_nhwc_f32(int n, const float* const* src, const int* meta, float* const* out, int relu, void* stream)
{
    CP_CHECK_ARG(n >= 1 && n <= 4 && src && meta && out, "sum_up_group: 1..4 members (got %d)", n);
    SumUpGroup g;
    g.n = n; g.relu = relu;
    long long total = 0;
    for (int k = 0; k < 4; ++k) {
        const int kk = k < n ? k : 0;                                    // unused slots repeat member 0 (never indexed)
        const int* mt = meta + 14 * kk;
        SumUpMember& m = g.m[k];
        const int ns = mt[0], B = mt[10], H = mt[11], W = mt[12], C = mt[13];
        if (k < n) {
            CP_CHECK_ARG(ns >= 1 && ns <= 4 && out[k] && C > 0 && C % 4 == 0 && mt[9] % 4 == 0, "sum_up_group: member %d: bad arguments", k);
            CP_CHECK_ARG((long long)W * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * H < (1ll << 31), "sum_up_group: member %d: row too large", k);
        }
        for (int i = 0; i < 4; ++i) {
            m.a.src[i] = i < ns ? src[4 * kk + i] : nullptr; m.a.ld[i] = i < ns ? mt[1 + i] : 0; m.a.sh[i] = i < ns ? mt[5 + i] : 0;
            if (k < n && i < ns) CP_CHECK_ARG(m.a.src[i] && m.a.ld[i] % 4 == 0, "sum_up_group: member %d source %d", k, i);
        }
        m.a.n = ns;
        m.out = out[kk]; m.outLd = mt[9]; m.H = H; m.W = W;
        m.r = ew_row(B * H, W, C / 4);
        m.xblocks = (m.r.rowElems + EW_THREADS - 1) / EW_THREADS;
        g.first[k] = (int)total;
        if (k < n) total += (long long)m.r.rows * m.xblocks;
    }
    g.first[4] = (int)total;
    for (int k = n; k < 4; ++k) g.first[k] = (int)total;
    CP_CHECK_ARG(total > 0 && total < (1ll << 31), "sum_up_group: grid %lld", total);
    hipLaunchKernelGGL(sum_up_group_kernel, dim3((unsigned)total), dim3(EW_THREADS), 0, (hipStream_t)stream, g);
    cp_note_kernel("sum_up_group_kernel");
    CP_CHECK_LAUNCH("sum_up_group_kernel");
    return 0;
}

// ---- layout transforms (32x32 LDS tile transpose over (C, HW)) -------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int outLd, int cOff)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        t[r][tx] = (c < C && p < HW) ? in[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (c < C && p < HW) out[((size_t)b * HW + p) * outLd + cOff + c] = t[tx][r];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int inLd, int cOff, float* __restrict__ out, int C, int HW)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        t[r][tx] = (c < C && p < HW) ? in[((size_t)b * HW + p) * inLd + cOff + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((size_t)b * C + c) * HW + p] = t[tx][r];
    }
}

extern "C" int cp_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, int outLd, int cOff, void* stream)
{
    CP_CHECK_ARG(in && out && outLd >= cOff + C, "nchw_to_nhwc: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cp_cdiv(HW, 32), cp_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, out,
                       C, HW, outLd, cOff);
    CP_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return 0;
}
extern "C" int cp_nhwc_to_nchw_f32(const float* in, int inLd, int cOff, float* out, int B, int C, int H, int W, void* stream)
{
    CP_CHECK_ARG(in && out && inLd >= cOff + C, "nhwc_to_nchw: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cp_cdiv(HW, 32), cp_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, inLd,
                       cOff, out, C, HW);
    CP_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return 0;
}

// Kernel-side constants of a DCNv2 (weight, bias) pair in ONE launch: w [Co][C][kh*kw] (the reference's nn.Parameter layout,
// DCNv2/dcn_v2.py:99-103) -> wp [ldw][kh*kw*Cp] with k = (tap * dg + g) * cpgp + c (every deformable group padded from cpg to cpgp
// channels), zero rows / columns for the padding; scale = 1, shift = bias for n < Co, 0 above.  A few microseconds for the
// reference's layers, so the drop-in dcn_v2_forward faces pack on EVERY call instead of caching packed weights on tensor identity
// (`.data` edits do not bump a parameter's version counter; a cache keyed on it silently convolves with stale weights).
__global__ void dcn_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wp, float* __restrict__ scale,
                                float* __restrict__ shift, int Co, int C, int kk, int dg, int cpg, int cpgp, int ldw)
{
    const int Kp = kk * dg * cpgp;
    const long long total = (long long)ldw * Kp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kp), k = (int)(i - (long long)n * Kp);
        const int c = k % cpgp, tg = k / cpgp, g = tg % dg, tap = tg / dg;
        wp[i] = (n < Co && c < cpg) ? w[((size_t)n * C + g * cpg + c) * kk + tap] : 0.f;
        if (k == 0) { scale[n] = n < Co ? 1.f : 0.f; shift[n] = n < Co ? bias[n] : 0.f; }
    }
}
extern "C" int cp_dcn_pack_weights_f32(const float* w, const float* bias, int Co, int C, int kh, int kw, int dg, int Cp, int ldw, float* wp,
                                       float* scale, float* shift, void* stream)
{
    CP_CHECK_ARG(w && bias && wp && scale && shift, "dcn_pack_weights: null pointer");
    CP_CHECK_ARG(dg >= 1 && C % dg == 0 && Cp % dg == 0 && Cp / dg >= C / dg && (Cp / dg) % 16 == 0 && ldw >= Co && kh * kw >= 1,
                 "dcn_pack_weights: C=%d dg=%d Cp=%d Co=%d ldw=%d", C, dg, Cp, Co, ldw);
    const long long total = (long long)ldw * kh * kw * Cp;
    hipLaunchKernelGGL(dcn_pack_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream, w, bias, wp, scale, shift, Co, C,
                       kh * kw, dg, C / dg, Cp / dg, ldw);
    CP_CHECK_LAUNCH("dcn_pack_kernel");
    return 0;
}

extern "C" int cp_fill_f32(float* p, float v, long long n, void* stream);
__global__ void fill_kernel(float* p, float v, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int cp_fill_f32(float* p, float v, long long n, void* stream)
{
    CP_CHECK_ARG(p && n >= 0, "fill: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, v, n);
    CP_CHECK_LAUNCH("fill_kernel");
    return 0;
}

// ---- flip-test merge (multi_pose.py:45-53; models/utils.py:27-47), all on the device ------------
// in[2,C,H,W] NCHW: image 0 and its mirrored twin.  out[1,C,H,W] = (in[0] + flip_w(in[1]') ) / 2 where
// in[1]' optionally has channels permuted (left/right joint swap) and selected channels negated.
// mode 0: plain W-flip (hm, wh).  mode 1: hm_hp: joint swap.  mode 2: hps: joint swap on (x,y) pairs,
// x components negated (flip_lr_off).
__global__ void flip_merge_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int mode,
                                  const int* __restrict__ perm /* [J] joint permutation or NULL */)
{
    const long long total = (long long)C * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)(i / ((long long)W * H));
        int cs = c;
        float sign = 1.f;
        if (mode == 1) cs = perm[c];
        else if (mode == 2) { cs = 2 * perm[c >> 1] + (c & 1); if ((c & 1) == 0) sign = -1.f; }
        const float a = in[i];
        const float b = in[total + ((long long)cs * H + y) * W + (W - 1 - x)] * sign;
        out[i] = (a + b) / 2.0f;
    }
}

extern "C" int cp_flip_merge_f32(const float* in, float* out, int C, int H, int W, int mode, const int* perm, void* stream)
{
    CP_CHECK_ARG(in && out && (mode == 0 || perm), "flip_merge: bad arguments");
    const long long total = (long long)C * H * W;
    hipLaunchKernelGGL(flip_merge_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream, in, out, C, H, W, mode,
                       perm);
    CP_CHECK_LAUNCH("flip_merge_kernel");
    return 0;
}

// ---- depthwise k x k convolution + folded BN + activation (groups == channels) ---------------------------------------
// w: [k*k][C] (tap-major, channel contiguous); scale / shift: [C].  Bandwidth-bound: float4 over channels.
__global__ void dwconv_nhwc_kernel(const float* __restrict__ in, int inLd, const float* __restrict__ w, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ out, int outLd, EwRow r, int H, int W, int Ho,
                                   int Wo, int C, int k, int s, int p, int act)
{
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int ox, c4;
    ew_split(r, e, ox, c4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4), sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {
        const int b = row / Ho, oy = row - b * Ho;                     // uniform
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - p + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - p + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(b * H + iy) * W + ix) * inLd + c4 * 4);
                const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)(ky * k + kx) * C + c4 * 4);
                acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y); acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
            }
        }
        float4 o;
        o.x = cp_act(acc.x * sc.x + sh.x, act); o.y = cp_act(acc.y * sc.y + sh.y, act);
        o.z = cp_act(acc.z * sc.z + sh.z, act); o.w = cp_act(acc.w * sc.w + sh.w, act);
        *reinterpret_cast<float4*>(out + ((size_t)row * Wo + ox) * outLd + c4 * 4) = o;
    }
}

extern "C" int cp_dwconv2d_nhwc_f32(const float* in, int inLd, const float* w, const float* scale, const float* shift, float* out,
                                    int outLd, int B, int H, int W, int C, int k, int s, int p, int act, void* stream)
{
    CP_CHECK_ARG(in && w && scale && shift && out && C % 4 == 0 && inLd % 4 == 0 && outLd % 4 == 0, "dwconv: C, ld must be multiples of 4");
    CP_CHECK_ARG(k >= 1 && s >= 1 && p >= 0 && act >= 0 && act <= CP_ACT_HSIGMOID_, "dwconv: bad k / stride / pad / act");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    CP_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)Wo * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * Ho < (1ll << 31), "dwconv: bad shape");
    const EwRow r = ew_row(B * Ho, Wo, C / 4);
    hipLaunchKernelGGL(dwconv_nhwc_kernel, ew_row_grid(r), dim3(EW_THREADS), 0, (hipStream_t)stream, in, inLd, w, scale, shift, out,
                       outLd, r, H, W, Ho, Wo, C, k, s, p, act);
    cp_note_kernel("dwconv_nhwc_kernel");
    CP_CHECK_LAUNCH("dwconv_nhwc_kernel");
    return 0;
}

// ---- global average pool: in [B, HW, C] -> out [B, C] (nn.AdaptiveAvgPool2d(1)) -----------------------------------------
// block (32 channel quads x 8 pixel partitions); grid (channel-quad groups, B)
__global__ void global_avgpool_kernel(const float* __restrict__ in, int inLd, float* __restrict__ out, int outLd, int HW, int C4)
{
    __shared__ float4 part[8][32];
    const int cq = threadIdx.x & 31, pp = threadIdx.x >> 5, c4 = blockIdx.x * 32 + cq, b = blockIdx.y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4)
        for (int p = pp; p < HW; p += 8) {
            const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)b * HW + p) * inLd + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    part[pp][cq] = acc;
    __syncthreads();
    if (pp == 0 && c4 < C4) {
#pragma unroll
        for (int j = 1; j < 8; ++j) { const float4 v = part[j][cq]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        const float n = (float)HW;
        *reinterpret_cast<float4*>(out + (size_t)b * outLd + c4 * 4) = make_float4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
    }
}

extern "C" int cp_global_avgpool_nhwc_f32(const float* in, int inLd, float* out, int outLd, int B, int HW, int C, void* stream)
{
    CP_CHECK_ARG(in && out && C % 4 == 0 && inLd % 4 == 0 && outLd % 4 == 0 && B > 0 && HW > 0 && B <= 65535, "global_avgpool: bad arguments");
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)cp_cdiv(C / 4, 32), (unsigned)B), dim3(256), 0, (hipStream_t)stream, in, inLd,
                       out, outLd, HW, C / 4);
    cp_note_kernel("global_avgpool_kernel");
    CP_CHECK_LAUNCH("global_avgpool_kernel");
    return 0;
}

// ---- squeeze-excite apply: out[b,p,c] = x[b,p,c] * se[b,c] (+ add[b,p,c])   (mobilenetv3.py:112-113,141-143) ------------
__global__ void scale_add_kernel(const float* __restrict__ x, int xLd, const float* __restrict__ se, int seLd, const float* __restrict__ add,
                                 int addLd, float* __restrict__ out, int outLd, EwRow r, int H)
{
    const int e = blockIdx.x * EW_THREADS + threadIdx.x;
    if (e >= r.rowElems) return;
    int px, c4;
    ew_split(r, e, px, c4);
    for (int row = blockIdx.y; row < r.rows; row += gridDim.y) {
        const int b = row / H;                                          // uniform
        const size_t pix = (size_t)row * (r.rowElems / r.C4) + px;
        const float4 v = *reinterpret_cast<const float4*>(x + pix * xLd + c4 * 4);
        const float4 g = *reinterpret_cast<const float4*>(se + (size_t)b * seLd + c4 * 4);
        float4 o = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
        if (add) {
            const float4 a4 = *reinterpret_cast<const float4*>(add + pix * addLd + c4 * 4);
            o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
        }
        *reinterpret_cast<float4*>(out + pix * outLd + c4 * 4) = o;
    }
}

extern "C" int cp_scale_add_nhwc_f32(const float* x, int xLd, const float* se, int seLd, const float* add, int addLd, float* out, int outLd,
                                     int B, int H, int W, int C, void* stream)
{
    CP_CHECK_ARG(x && se && out && C % 4 == 0 && xLd % 4 == 0 && seLd % 4 == 0 && outLd % 4 == 0 && (!add || addLd % 4 == 0),
                 "scale_add: C, ld must be multiples of 4");
    CP_CHECK_ARG((long long)W * (C / 4) * (C / 4) < (1ll << 32) && (long long)B * H < (1ll << 31), "scale_add: row too large");
    const EwRow r = ew_row(B * H, W, C / 4);
    hipLaunchKernelGGL(scale_add_kernel, ew_row_grid(r), dim3(EW_THREADS), 0, (hipStream_t)stream, x, xLd, se, seLd, add, addLd, out, outLd,
                       r, H);
    cp_note_kernel("scale_add_kernel");
    CP_CHECK_LAUNCH("scale_add_kernel");
    return 0;
}

// ---- channel_shuffle(cat(x1, x2), groups = 2)   (shufflenetv2_dcn.py:28-42,103-104) --------------------------------------
// x1, x2: h channels each.  The shuffled tensor has 2h logical channels, logical 2i = x1[i], 2i+1 = x2[i].  It is stored
// as TWO halves of hp >= h physical channels each (hp % 16 == 0: the next block splits it down the middle, x[:, :h] /
// x[:, h:], and every kernel wants 16-aligned channel runs): logical c < h -> physical c, c >= h -> hp + (c - h); the
// padding channels are written as zeros.
__global__ void shuffle_concat_kernel(const float* __restrict__ x1, int ld1, const float* __restrict__ x2, int ld2, float* __restrict__ out,
                                      int outLd, long long npix, int h, int hp)
{
    const long long total = npix * 2 * hp;
    for (long long e = (long long)blockIdx.x * EW_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * EW_THREADS) {
        const long long pix = e / (2 * hp);
        const int pc = (int)(e - pix * 2 * hp);
        const int half = pc >= hp, q = pc - half * hp;
        float v = 0.f;
        if (q < h) {
            const int L = half * h + q;                                  // logical channel
            v = (L & 1) ? x2[pix * ld2 + (L >> 1)] : x1[pix * ld1 + (L >> 1)];
        }
        out[pix * outLd + pc] = v;
    }
}

extern "C" int cp_shuffle_concat_nhwc_f32(const float* x1, int ld1, const float* x2, int ld2, float* out, int outLd, long long npix, int h,
                                          int hp, void* stream)
{
    CP_CHECK_ARG(x1 && x2 && out && h > 0 && hp >= h && outLd >= 2 * hp && npix > 0, "shuffle_concat: bad arguments");
    hipLaunchKernelGGL(shuffle_concat_kernel, dim3(ew_grid(npix * 2 * hp)), dim3(EW_THREADS), 0, (hipStream_t)stream, x1, ld1, x2, ld2, out,
                       outLd, npix, h, hp);
    cp_note_kernel("shuffle_concat_kernel");
    CP_CHECK_LAUNCH("shuffle_concat_kernel");
    return 0;
}
