/* TEST INFRASTRUCTURE ONLY -- C restatement of the reference soft_nms_39
 * (lib/external/nms.pyx:172-275), the only variant the detector calls
 * (lib/detectors/multi_pose.py:76-77: soft_nms_39(results, Nt=0.5, method=2)).
 * The Cython source does not compile with Cython 3 / numpy 2 (np.int_t, nms.pyx:32).  Pinned since round 3 against vectors
 * produced by the reference's OWN source: tests/golden/make_golden_nms.py executes the body of soft_nms_39 as Python (the
 * `cdef` declarations stripped in memory, float32 numpy scalars standing for the C floats) -> tests/golden/soft_nms_39.npz;
 * tests/test_host_logic.py::test_soft_nms_39_matches_reference_source_golden: box moves, the 0:39-only swap, discards and `keep`
 * bit for bit, hard / linear scores bit for bit, Gaussian scores to 1 ulp per decay (exp evaluated in another precision).
 * Plus hand-worked cases (test_soft_nms_39_hand_worked).
 *
 * boxes: [N,56] float32 rows, modified IN PLACE exactly like the reference: the max-score row is
 * swapped into position i (columns 0..38 only, nms.pyx:214-235 -- the 17 keypoint scores 39..55
 * stay in their row slot), overlapping rows are re-weighted (linear / gaussian / hard) and rows
 * whose score drops below `threshold` are swapped to the end and N shrinks.  Returns the number
 * of kept rows; keep[] receives 0..N-1 like the reference's `keep = [i for i in range(N)]`.
 */
#include <math.h>
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static float fminf_(float a, float b) { return a < b ? a : b; }

int soft_nms_39_ref(float *boxes, int N, float sigma, float Nt, float threshold, int method, int *keep)
{
    const int S = 56;
    const int N0 = N;                          /* Cython evaluates range(N) once (nms.pyx:182) */
    for (int i = 0; i < N0; ++i) {
        float maxscore = boxes[i * S + 4];
        int maxpos = i;
        float tmp[39];
        for (int c = 0; c < 39; ++c) tmp[c] = boxes[i * S + c];
        int pos = i + 1;
        while (pos < N) {                      /* nms.pyx:203-207 */
            if (maxscore < boxes[pos * S + 4]) { maxscore = boxes[pos * S + 4]; maxpos = pos; }
            pos++;
        }
        for (int c = 0; c < 39; ++c) boxes[i * S + c] = boxes[maxpos * S + c];   /* :210-217 */
        for (int c = 0; c < 39; ++c) boxes[maxpos * S + c] = tmp[c];             /* :220-227 */
        for (int c = 0; c < 39; ++c) tmp[c] = boxes[i * S + c];                  /* :229-236 */
        const float tx1 = tmp[0], ty1 = tmp[1], tx2 = tmp[2], ty2 = tmp[3];
        pos = i + 1;
        while (pos < N) {                      /* :240-273 */
            float x1 = boxes[pos * S + 0], y1 = boxes[pos * S + 1];
            float x2 = boxes[pos * S + 2], y2 = boxes[pos * S + 3];
            float area = (x2 - x1 + 1) * (y2 - y1 + 1);
            float iw = fminf_(tx2, x2) - fmaxf_(tx1, x1) + 1;
            if (iw > 0) {
                float ih = fminf_(ty2, y2) - fmaxf_(ty1, y1) + 1;
                if (ih > 0) {
                    float ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih;
                    float ov = iw * ih / ua;
                    float weight;
                    if (method == 1) weight = ov > Nt ? 1 - ov : 1;
                    else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma)); /* np.exp -> double, :252 */
                    else weight = ov > Nt ? 0 : 1;
                    boxes[pos * S + 4] = weight * boxes[pos * S + 4];
                    if (boxes[pos * S + 4] < threshold) {          /* :263-270 */
                        for (int c = 0; c < 5; ++c) boxes[pos * S + c] = boxes[(N - 1) * S + c];   /* copy */
                        for (int c = 5; c < 39; ++c) {                                           /* swap */
                            float t2 = boxes[pos * S + c];
                            boxes[pos * S + c] = boxes[(N - 1) * S + c];
                            boxes[(N - 1) * S + c] = t2;
                        }
                        N--; pos--;
                    }
                }
            }
            pos++;
        }
    }
    for (int i = 0; i < N; ++i) keep[i] = i;
    return N;
}
