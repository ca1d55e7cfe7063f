/* TEST INFRASTRUCTURE ONLY -- scalar C restatement of the reference DCNv2 forward.
 *
 * The reference has no CPU implementation of this op
 * (lib/models/backbones/DCNv2/src/cpu/dcn_v2_cpu.cpp:7-24 raises "Not implemented on the
 * CPU"), its extension cannot be built here (THC headers, CUDA only), and its own test
 * (DCNv2/test.py) stores no expected values.  DCN parity is therefore pinned only through
 * the reference's known-answer properties (zero-offset identity, test.py:31-66; integer
 * offsets == shifted taps; boundary rule) -- see tests/test_oracle_dcn.py -- and this
 * transliteration of the algorithm:
 *
 *   bilinear sample   : src/cuda/dcn_v2_im2col_cuda.cu:25-54   (dmcn_im2col_bilinear)
 *   im2col + bounds   : src/cuda/dcn_v2_im2col_cuda.cu:125-195 (one (b,c,h,w) per thread,
 *                       offset channel 2k = dy, 2k+1 = dx, mask channel k; valid iff
 *                       h_im > -1 && w_im > -1 && h_im < H && w_im < W, line 180)
 *   bias + GEMM       : src/cuda/dcn_v2_cuda.cu:123-163 (out = bias; out += W[Co, C*kh*kw] . col)
 *
 * Layouts are the reference's: NCHW float32, weight [Co, C, kh, kw], offset
 * [B, 2*dg*kh*kw, Ho, Wo], mask [B, dg*kh*kw, Ho, Wo].  Accumulation is double so the
 * oracle is a tighter reference than either float32 implementation.
 */
#include <math.h>
#include <stddef.h>

static float bilinear(const float *im, int H, int W, float h, float w)
{
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - h_low, lw = w - w_low;
    float hh = 1 - lh, hw = 1 - lw;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* returns 0 on success */
int dcn_v2_forward_ref(const float *input, const float *weight, const float *bias,
                       const float *offset, const float *mask, float *output,
                       int B, int C, int H, int W, int Co,
                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                       int dg)
{
    const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    const int cpg = C / dg;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Co; ++o)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    double acc = bias ? bias[o] : 0.0;
                    for (int c = 0; c < C; ++c) {
                        const int g = c / cpg;
                        const float *im = input + ((size_t)b * C + c) * H * W;
                        const float *offp = offset + ((size_t)b * dg + g) * 2 * kh * kw * Ho * Wo;
                        const float *mp = mask + ((size_t)b * dg + g) * kh * kw * Ho * Wo;
                        for (int i = 0; i < kh; ++i)
                            for (int j = 0; j < kw; ++j) {
                                const int k = i * kw + j;
                                const float oh = offp[((size_t)(2 * k) * Ho + y) * Wo + x];
                                const float ow = offp[((size_t)(2 * k + 1) * Ho + y) * Wo + x];
                                const float m = mp[((size_t)k * Ho + y) * Wo + x];
                                const float h_im = (float)(y * sh - ph + i * dh) + oh;
                                const float w_im = (float)(x * sw - pw + j * dw) + ow;
                                float val = 0.f;
                                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                                    val = bilinear(im, H, W, h_im, w_im);
                                const float col = val * m;
                                acc += (double)weight[(((size_t)o * C + c) * kh + i) * kw + j] * col;
                            }
                    }
                    output[(((size_t)b * Co + o) * Ho + y) * Wo + x] = (float)acc;
                }
    return 0;
}
