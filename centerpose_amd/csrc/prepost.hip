// Device pre- and post-processing around the hot path (SURVEY 8(f1) "next" row).
//
//   cp_preprocess_u8_f32 : cv2.warpAffine(INTER_LINEAR, border 0) + (x/255 - mean)/std + HWC->CHW
//                          (+ mirrored twin for FLIP_TEST)      lib/detectors/base_detector.py:46-56
//   cp_transform_dets_f32: inverse affine of the 2 box corners + 17 keypoints of every detection
//                          lib/utils/post_process.py:8-19, lib/utils/image.py:19-24 (per-point Python loop)
// Once the network runs at >1000 img/s these host stages (cv2 + numpy in the reference) dominate
// BaseDetector.run; both are trivially bandwidth-bound on the device.
// Parity: the reference's cv2 path uses 5-bit fixed-point interpolation weights on uint8 images and
// cannot run here (cv2 absent): "parity unpinned"; the kernel is checked against a float restatement.
#include "common.h"

// M: 2x3 matrix mapping OUTPUT pixel (x,y) -> SOURCE image coordinates (i.e. the inverse of the
// matrix passed to cv2.warpAffine).  img: uint8 HWC (C = 3, BGR as cv2.imread gives).
__global__ void preprocess_kernel(const unsigned char* __restrict__ img, int H, int W, float m00, float m01, float m02,
                                  float m10, float m11, float m12, float* __restrict__ out, int OH, int OW, float mean0,
                                  float mean1, float mean2, float is0, float is1, float is2, int flip)
{
    const int total = OH * OW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int oy = i / OW, ox = i - oy * OW;
        const float sx = m00 * ox + m01 * oy + m02, sy = m10 * ox + m11 * oy + m12;
        const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
        const float fx = sx - x0, fy = sy - y0;
        float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float wgt = (dy ? fy : 1.f - fy) * (dx ? fx : 1.f - fx);
                const unsigned char* p = img + ((size_t)yy * W + xx) * 3;
                v[0] += wgt * p[0]; v[1] += wgt * p[1]; v[2] += wgt * p[2];
            }
        const float r0 = (v[0] / 255.f - mean0) * is0, r1 = (v[1] / 255.f - mean1) * is1, r2 = (v[2] / 255.f - mean2) * is2;
        out[(size_t)0 * total + i] = r0;
        out[(size_t)1 * total + i] = r1;
        out[(size_t)2 * total + i] = r2;
        if (flip) {   // second batch entry = images[:, :, :, ::-1]
            const int j = oy * OW + (OW - 1 - ox);
            out[(size_t)3 * total + j] = r0;
            out[(size_t)4 * total + j] = r1;
            out[(size_t)5 * total + j] = r2;
        }
    }
}

extern "C" int cp_preprocess_u8_f32(const unsigned char* img, int H, int W, const float* M /* host 2x3 */, float* out, int OH,
                                    int OW, const float* mean /* host 3 */, const float* std_ /* host 3 */, int flip, void* stream)
{
    CP_CHECK_ARG(img && M && out && mean && std_ && H > 0 && W > 0 && OH > 0 && OW > 0, "preprocess: bad arguments");
    const int total = OH * OW;
    int grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, H, W, M[0], M[1], M[2], M[3], M[4],
                       M[5], out, OH, OW, mean[0], mean[1], mean[2], 1.f / std_[0], 1.f / std_[1], 1.f / std_[2], flip);
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
