// Device pre- and post-processing around the hot path (SURVEY 8(f1) "next" row).
//
//   cp_resize_u8         : cv2.resize(image, (new_w, new_h)) (INTER_LINEAR, 8-bit)    lib/detectors/base_detector.py:47
//   cp_preprocess_u8_f32 : cv2.warpAffine(INTER_LINEAR, border 0) + (x/255 - mean)/std + HWC->CHW
//                          (+ mirrored twin for FLIP_TEST)                            lib/detectors/base_detector.py:48-58
//   cp_transform_dets_f32: inverse affine of the 2 box corners + 17 keypoints of every detection
//                          lib/utils/post_process.py:8-19, lib/utils/image.py:19-24 (per-point Python loop)
// Once the network runs at >1000 img/s these host stages (cv2 + numpy in the reference) dominate
// BaseDetector.run; both are trivially bandwidth-bound on the device.
//
// The pixel arithmetic is OpenCV's 8-bit fixed-point algorithm (opencv-python, unpinned in requirements.txt; absent from the
// reference tree and from this image): resize.cpp HResizeLinear / VResizeLinear (11-bit weights) and imgwarp.cpp warpAffine /
// remapBilinear (coordinates in 1/32 px, 15-bit weights) -- restated in oracle/prepost_np.py, which these kernels match
// bit for bit.  "Parity unpinned" against cv2 itself (cannot be imported here); pinned by hand-computed cases.
// Compiled with -ffp-contract=off: the coordinate arithmetic must round like the host C++ it restates.
#include "common.h"

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// one (tap index pair, 11-bit weight pair) of cv::resize's linear interpolation along one axis
__device__ __forceinline__ void resize_taps(int d, double scale, int n, bool vertical, int& i0, int& i1, int& w0, int& w1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (!vertical) {                       // columns: fx = 0 at the clamped ends
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n - 1) { f = 0.f; s = n - 1; }
    }
    w1 = sat_short(__float2int_rn(f * 2048.f));
    w0 = sat_short(__float2int_rn((1.f - f) * 2048.f));
    i0 = min(max(s, 0), n - 1);            // rows: weights kept, indices clipped
    i1 = min(max(s + 1, 0), n - 1);
}

__global__ void resize_u8_kernel(const unsigned char* __restrict__ src, int H, int W, unsigned char* __restrict__ dst, int NH,
                                 int NW, double scale_x, double scale_y)
{
    const int total = NH * NW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int dy = i / NW, dx = i - dy * NW;
        int x0, x1, a0, a1, y0, y1, b0, b1;
        resize_taps(dx, scale_x, W, false, x0, x1, a0, a1);
        resize_taps(dy, scale_y, H, true, y0, y1, b0, b1);
        const unsigned char* r0 = src + (size_t)y0 * W * 3;
        const unsigned char* r1 = src + (size_t)y1 * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
            const int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
            const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)i * 3 + c] = (unsigned char)min(max(v, 0), 255);
        }
    }
}

extern "C" int cp_resize_u8(const unsigned char* src, int H, int W, unsigned char* dst, int NH, int NW, void* stream)
{
    CP_CHECK_ARG(src && dst && H > 0 && W > 0 && NH > 0 && NW > 0, "resize_u8: bad arguments");
    const int total = NH * NW;
    int grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, H, W, dst, NH, NW, (double)W / NW,
                       (double)H / NH);
    CP_CHECK_LAUNCH("resize_u8_kernel");
    return 0;
}

struct WarpMat { double m[6]; };     // INVERTED matrix: destination pixel -> source coordinates

// img: uint8 HWC (C = 3, BGR as cv2.imread gives).
__global__ void preprocess_kernel(const unsigned char* __restrict__ img, int H, int W, WarpMat Mi, float* __restrict__ out, int OH,
                                  int OW, float mean0, float mean1, float mean2, float std0, float std1, float std2, int flip)
{
    const int total = OH * OW;
    const float mean[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int oy = i / OW, ox = i - oy * OW;
        // imgwarp.cpp: AB_BITS = 10, INTER_BITS = 5, round_delta = 1024 / 32 / 2
        const int adelta = (int)__double2ll_rn(Mi.m[0] * (double)ox * 1024.0);
        const int bdelta = (int)__double2ll_rn(Mi.m[3] * (double)ox * 1024.0);
        const int X0 = (int)__double2ll_rn((Mi.m[1] * (double)oy + Mi.m[2]) * 1024.0) + 16;
        const int Y0 = (int)__double2ll_rn((Mi.m[4] * (double)oy + Mi.m[5]) * 1024.0) + 16;
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
        const int fx = X & 31, fy = Y & 31;
        int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
        if (w[0] > 32767) w[0] = 32767;      // saturate_cast<short>(32768) for the exact-pixel phase
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = sy + (t >> 1), xx = sx + (t & 1);
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;      // constant border, value 0
            const unsigned char* p = img + ((size_t)yy * W + xx) * 3;
            acc[0] += w[t] * p[0]; acc[1] += w[t] * p[1]; acc[2] += w[t] * p[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int u8 = min(max((acc[c] + (1 << 14)) >> 15, 0), 255);
            // numpy: (uint8 / 255. - float32 mean) / float32 std evaluates in float64, then .astype(float32)  (base_detector.py:53)
            const float r = (float)((((double)u8 / 255.0) - (double)mean[c]) / (double)sd[c]);
            out[(size_t)c * total + i] = r;
            if (flip) out[(size_t)(3 + c) * total + oy * OW + (OW - 1 - ox)] = r;      // images[:, :, :, ::-1]
        }
    }
}

extern "C" int cp_preprocess_u8_f32(const unsigned char* img, int H, int W, const double* M /* host 2x3, source -> destination */,
                                    float* out, int OH, int OW, const float* mean /* host 3 */, const float* std_ /* host 3 */,
                                    int flip, void* stream)
{
    CP_CHECK_ARG(img && M && out && mean && std_ && H > 0 && W > 0 && OH > 0 && OW > 0, "preprocess: bad arguments");
    // the inversion cv::warpAffine applies to its matrix argument (imgwarp.cpp), double precision
    WarpMat mi;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    mi.m[0] = A11; mi.m[1] = M[1] * -D;
    mi.m[3] = M[3] * -D; mi.m[4] = A22;
    mi.m[2] = -mi.m[0] * M[2] - mi.m[1] * M[5];
    mi.m[5] = -mi.m[3] * M[2] - mi.m[4] * M[5];
    const int total = OH * OW;
    int grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, H, W, mi, out, OH, OW, mean[0], mean[1],
                       mean[2], std_[0], std_[1], std_[2], flip);
    CP_CHECK_LAUNCH("preprocess_kernel");
    return 0;
}

// dets[B,K,D] (D = 5 + 3J): columns 0..3 and 5..5+2J are (x,y) pairs in feature-map pixels.
// trans: per image 2x3 (double, device) mapping feature-map -> image coordinates; result / scale.
__global__ void transform_dets_kernel(const float* __restrict__ dets, float* __restrict__ out, const double* __restrict__ trans,
                                      int B, int K, int J, float scale)
{
    const int D = 5 + 3 * J, total = B * K * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % D, b = i / (K * D);
        const bool coord = c < 4 || (c >= 5 && c < 5 + 2 * J);
        float v = dets[i];
        if (coord) {
            const int base = i - c;
            const int cc = c < 4 ? c : c - 5;              // index inside its (x,y) run
            const int xi = base + (c < 4 ? 0 : 5) + (cc & ~1);
            const double x = (double)dets[xi], y = (double)dets[xi + 1];
            const double* t = trans + b * 6 + ((cc & 1) ? 3 : 0);
            v = (float)(t[0] * x + t[1] * y + t[2]) / scale;   // float32(np.dot(t, pt)) / scale (multi_pose.py:68-70)
        }
        out[i] = v;
    }
}

extern "C" int cp_transform_dets_f32(const float* dets, float* out, const double* trans, int B, int K, int J, float scale,
                                     void* stream)
{
    CP_CHECK_ARG(dets && out && trans && B > 0 && K > 0 && J > 0 && scale > 0.f, "transform_dets: bad arguments");
    const int total = B * K * (5 + 3 * J);
    hipLaunchKernelGGL(transform_dets_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dets, out, trans, B, K,
                       J, scale);
    CP_CHECK_LAUNCH("transform_dets_kernel");
    return 0;
}
