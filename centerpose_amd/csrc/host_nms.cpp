// Host-side soft-NMS for merged multi-scale / NMS-enabled results.
// Replaces the Cython soft_nms_39 (lib/external/nms.pyx:172-275), the only variant the detector
// calls (lib/detectors/multi_pose.py:76-77: soft_nms_39(results, Nt=0.5, method=2)).  O(N^2) on
// <= K * #scales rows of 56 floats: sequential, host memory, in place -- quirks kept on purpose:
//  * rows are re-ordered by swapping columns 0..38 only; the 17 keypoint scores (39..55) stay in
//    their row slot (nms.pyx:214-217,265-268);
//  * a suppressed row receives a COPY of the last row's box+score (cols 0..4) and a SWAP of cols 5..38;
//  * the gaussian weight goes through double precision (np.exp) and back to float;
//  * "+1" pixel area convention; the outer loop runs over the ORIGINAL N (Cython evaluates
//    range(N) once).
#include <cmath>
#include "common.h"

extern "C" int cp_soft_nms_39(float* boxes /* HOST [N,56] */, int N, float sigma, float Nt, float threshold, int method,
                              int* keep /* HOST [N] or NULL */, int* n_keep)
{
    CP_CHECK_ARG(boxes && N >= 0 && n_keep, "soft_nms_39: bad arguments");
    const int S = 56, N0 = N;
    for (int i = 0; i < N0; ++i) {
        float maxscore = boxes[i * S + 4];
        int maxpos = i;
        float t[39];
        for (int c = 0; c < 39; ++c) t[c] = boxes[i * S + c];
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < boxes[pos * S + 4]) { maxscore = boxes[pos * S + 4]; maxpos = pos; }
        for (int c = 0; c < 39; ++c) boxes[i * S + c] = boxes[maxpos * S + c];
        for (int c = 0; c < 39; ++c) boxes[maxpos * S + c] = t[c];
        const float tx1 = boxes[i * S], ty1 = boxes[i * S + 1], tx2 = boxes[i * S + 2], ty2 = boxes[i * S + 3];
        int pos = i + 1;
        while (pos < N) {
            float* r = boxes + pos * S;
            const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
            const float area = (x2 - x1 + 1) * (y2 - y1 + 1);
            const float iw = std::fmin(tx2, x2) - std::fmax(tx1, x1) + 1;
            if (iw > 0) {
                const float ih = std::fmin(ty2, y2) - std::fmax(ty1, y1) + 1;
                if (ih > 0) {
                    const float ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih;
                    const float ov = iw * ih / ua;
                    float weight;
                    if (method == 1) weight = ov > Nt ? 1 - ov : 1;
                    else if (method == 2) weight = (float)std::exp((double)(-(ov * ov) / sigma));
                    else weight = ov > Nt ? 0 : 1;
                    r[4] = weight * r[4];
                    if (r[4] < threshold) {
                        float* last = boxes + (N - 1) * S;
                        for (int c = 0; c < 5; ++c) r[c] = last[c];
                        for (int c = 5; c < 39; ++c) { const float q = r[c]; r[c] = last[c]; last[c] = q; }
                        --N; --pos;
                    }
                }
            }
            ++pos;
        }
    }
    if (keep) for (int i = 0; i < N; ++i) keep[i] = i;
    *n_keep = N;
    return 0;
}
