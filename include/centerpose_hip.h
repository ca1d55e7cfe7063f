/* centerpose_hip.h -- C ABI of libcenterpose_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the inference hot path of tensorboy/centerpose.  Plain pointers and
 * sizes, no torch types.  All pointers are DEVICE pointers unless stated; `stream` is a
 * hipStream_t (NULL = default stream).  Every function returns 0 on success, non-zero on error
 * (1 = bad argument, 2 = launch/runtime failure); cp_last_error() gives the message
 * (thread-local).  Nothing synchronises the host; nothing allocates device memory.
 *
 * File:line citations are relative to the reference checkout (/root/reference).
 */
#ifndef CENTERPOSE_HIP_H
#define CENTERPOSE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int cp_abi_version(void);
const char* cp_target_arch(void);            /* "gfx950" */
const char* cp_last_error(void);

/* ---- heat-map decode -------------------------------------------------------------------------
 * Replaces multi_pose_decode(heat, wh, kps, reg, hm_hp, hp_offset, K)
 *   lib/models/decode.py:235-308 (with _nms :10-16, _topk :99-115, _topk_channel :87-96,
 *   _transpose_and_gather_feat lib/models/utils.py:11-25), called from
 *   lib/detectors/multi_pose.py:55.
 * heat[B,cat,H,W], hm_hp[B,J,H,W] are post-sigmoid NCHW float32; wh[B,2,H,W]; kps[B,2J,H,W];
 * reg / hp_offset [B,2,H,W] or NULL (then +0.5, decode.py:253-255,278-280).  hm_hp is mandatory
 * (the reference raises NameError without it).  dets[B,K,5+3J] float32 =
 * [x1,y1,x2,y2, score, (x,y)*J, joint_score*J] in feature-map pixels.
 * ws_scores[B,1+J,K] float32 / ws_inds[B,1+J,K] int32 are caller-owned scratch that also
 * returns the top-K peaks: plane 0 = person centres, plane 1+j = joint j candidates
 * (flat index y*W+x; order = value desc, index asc).
 * Limits: K <= 256, cat*H*W <= 32768 (the plane's sort keys stay resident in LDS). */
int cp_decode_workspace_bytes(int B, int J, int K, size_t* scores_bytes, size_t* inds_bytes);
int cp_multi_pose_decode_f32(const float* heat, const float* wh, const float* kps, const float* reg,
                             const float* hm_hp, const float* hp_offset, int B, int cat, int J, int H,
                             int W, int K, float* dets, float* ws_scores, int* ws_inds, void* stream);

#ifdef __cplusplus
}
#endif
#endif
