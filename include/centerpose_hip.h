/* centerpose_hip.h -- C ABI of libcenterpose_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the inference hot path of tensorboy/centerpose.  Plain pointers and
 * sizes, no torch types.  All pointers are DEVICE pointers unless stated; `stream` is a
 * hipStream_t (NULL = default stream).  Every function returns 0 on success, non-zero on error
 * (1 = bad argument, 2 = launch/runtime failure); cp_last_error() gives the message
 * (thread-local).  The kernel entry points never synchronise the host and never allocate device memory; kernels
 * are enqueued on `stream` only (so a caller may capture a sequence of calls into a hipGraph).  The plan handle
 * (cp_plan_*) is the one exception and says so.
 *
 * File:line citations are relative to the reference checkout (/root/reference).
 * Activation layout: NHWC float32, `ld` = floats between consecutive pixels (>= C).
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
const char* cp_last_kernel(void);            /* device kernel (template instantiation) the calling thread launched last */

enum { CP_ACT_NONE = 0, CP_ACT_RELU = 1, CP_ACT_SIGMOID = 2,
       CP_ACT_HSWISH = 3, CP_ACT_HSIGMOID = 4 };   /* x*relu6(x+3)/6, relu6(x+3)/6: lib/models/backbones/mobilenet/mobilenetv3.py:87-96 */

/* ---- fused convolution (implicit GEMM, fp32 MFMA) ------------------------------------------------
 * Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU + residual add + torch.cat (+ sigmoid) of
 *   lib/models/backbones/pose_dla_dcn.py:29-57,145-163,272-282 (BasicBlock, Root, conv levels),
 *   lib/models/backbones/msra_resnet.py:64-102,168-193 (Bottleneck, dense ConvTranspose2d k4 s2 p1 as
 *   4 sub-pixel 2x2 convs via the osy/osx/ooy/oox output scatter),
 *   lib/models/backbones/pose_higher_hrnet.py:98-235, lib/models/heads/keypoint.py:14-42,
 *   and the in-place sigmoid of lib/detectors/multi_pose.py:35-37.
 * out = act( (sum over up to 4 channel-concatenated sources of conv(src)) * scale + shift [+ res] ).
 * w: packed weights [ldw][K] (n-major, k contiguous), k = (ky*kw + kx)*Ctot + c  (inNCHW stem:
 * k = (c*kh + ky)*kw + kx, K padded to a multiple of 16 with zeros), ldw = Cout padded to 16 / 32 / a
 * multiple of 64 with zero rows;  3x3/s1/p1 single-source NHWC launches use the LDS halo-patch kernel;
 * scale/shift: [ldw] (folded BN or 1/bias).  res: NHWC residual with pixel stride resLd, or NULL. */
typedef struct cp_conv_desc {
    int nsrc;                 /* 1..4 sources */
    int srcC[4], srcLd[4];    /* channels taken from / pixel stride of each source (C % 16 == 0 unless inNCHW) */
    int B, H, W;              /* input spatial size */
    int Ho, Wo;               /* output positions computed by this launch */
    int kh, kw, sy, sx, py, px;
    int K, ldw;
    int Cout;                 /* channels stored */
    int resLd, outLd;
    int outNCHW;              /* 0: out is NHWC [B,OH,OW,outLd]; 1: out is NCHW [B,Cout,OH,OW] */
    int OH, OW, osy, osx, ooy, oox;   /* output pixel = (oy*osy+ooy, ox*osx+oox) inside [OH,OW] */
    int act;
    int inNCHW;               /* 1: src[0] is the NCHW network input [B,srcC[0],H,W] (base_detector.py:53-58) */
    int tile;                 /* 0 = auto; BM*1000+BN to force a kernel instantiation; 3000000 + BM*1000 + BN (BM, BN in {128, 64}): the
                                 OPT-IN split-bf16 kernel below -- `w` is then cp_split_bf16_weights_f32's output, NHWC sources only */
    int nsub;                 /* 0 / 1: one convolution.  4: the four sub-pixel 2x2 convolutions of a dense ConvTranspose2d(k4,s2,p1)
                                 (msra_resnet.py:168-193) in ONE launch: w = [4][ldw][K] (sub g = py*2+px), py = px = 1, osy = osx = 2,
                                 ooy = oox = 0; sub g uses pad (1-py, 1-px) and output phase (py, px) */
    int ksplit;               /* 0 / 1: none.  S > 1: split over the reduction axis for launches too small to fill the chip -- over the
                                 input channels (cp_conv3x3_winograd_f32) or the k-steps (cp_conv2d_f32: one NHWC source, nsub = 1, dense
                                 NHWC output): `out` is a workspace [S][B*Ho*Wo][outLd] of raw partial outputs (scale = ones, shift =
                                 zeros, act = CP_ACT_NONE, no residual); cp_splitk_reduce_f32 (ldw = outLd) finishes the layer */
} cp_conv_desc;
int cp_conv2d_f32(const cp_conv_desc* d, const float* const* src, const float* w, const float* scale, const float* shift,
                  const float* res, float* out, void* stream);

/* Up to 8 INDEPENDENT convolutions of the cp_conv2d_f32 kind in ONE launch: the fuse layers of an HRNet module
 * (pose_higher_hrnet.py:169-212: the 1x1 convs of the up paths and the stride-2 3x3 convs of the down paths read only the branch
 * outputs, 16-256 blocks each).  d[i], src[i], w[i], scale[i], shift[i], res[i] (may be NULL), out[i]: member i as for
 * cp_conv2d_f32 with nsrc = 1, NHWC in / dense NHWC out, ldw % 64 == 0, nsub <= 1, ksplit <= 1; outputs must not alias. */
int cp_conv2d_group_f32(const cp_conv_desc* d, int n, const float* const* src, const float* const* w, const float* const* scale,
                        const float* const* shift, const float* const* res, float* const* out, void* stream);

/* Opt-in fp32-EQUIVALENT mode of cp_conv2d_f32 on the bf16 matrix pipe (conv_igemm_bf16x3.hip; never the default, the reference's
 * arithmetic is fp32 end to end: DCNv2/src/cuda/dcn_v2_cuda.cu:58): every fp32 operand is three bf16 terms (8+8+8 significand bits,
 * exact), six bf16 MFMAs with fp32 accumulate per product tile, dropped terms <= 2^-24 relative.  Activations are split in the
 * kernel; weights once, here: w [rows][K] (rows = nsub * ldw, the layout cp_conv2d_f32 takes) -> wb, cp_split_bf16_weight_floats
 * (rows, K) floats holding bf16 [nsub][K/16][ldw][3][16].  Pass wb as `w` together with tile = 3128128 / 3064128 (ldw % 128 == 0) / 3128064 / 3064064. */
size_t cp_split_bf16_weight_floats(int rows, int K);
int cp_split_bf16_weights_f32(const float* w, int rows, int ldw, int K, float* wb, void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolution as fused Winograd F(2x2,3x3) ---------------------------------
 * Same layers and epilogue as cp_conv2d_f32 (the reference gets these from cuDNN, which chooses the algorithm itself --
 * heuristics, or autotuned with CUDNN.BENCHMARK, lib/config/default.py:29, tools/train.py:32); 16 multiplies per
 * 2x2 output tile and (cin, cout) instead of 36, fp32 throughout (error ~3e-6 relative, below the direct kernel's).
 * cp_winograd_pack_f32: packed direct weights w [rows >= Cout][9*C] (k = (ky*3+kx)*C + c, as for cp_conv2d_f32)
 *   -> u [cp_winograd_weight_floats(C, Cout)] = G g G^T in the kernel's MFMA B-fragment order; C % 16 == 0.
 * cp_conv3x3_winograd_f32: d as for cp_conv2d_f32 with nsrc = 1, kh = kw = 3, stride 1, pad 1, NHWC in/out
 *   (returns 1 for any other shape); d->tile: 0 = auto, MT*10+NT in {11, 12, 21} forces a block shape, 64xx the
 *   V-stationary kernel for 64 input channels with xx channel-tile groups per spatial tile; 24 = the F(2x4,3x3) kernel
 *   (conv3x3_wino24.hip: 2x4 output tiles, 24 multiplies per tile and (cin, cout) = 3 per output instead of 4; `u` must then
 *   come from cp_winograd24_pack_f32 -- same argument meaning, [cp_winograd24_weight_floats(C, Cout)] floats; no split-C). */
size_t cp_winograd_weight_floats(int C, int Cout);
int cp_winograd_pack_f32(const float* w, float* u, int C, int Cout, void* stream);
size_t cp_winograd24_weight_floats(int C, int Cout);
int cp_winograd24_pack_f32(const float* w, float* u, int C, int Cout, void* stream);
int cp_conv3x3_winograd_f32(const cp_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                            const float* res, float* out, void* stream);

/* Up to four INDEPENDENT 3x3 / stride-1 / pad-1 convolutions on the F(2x4,3x3) kernel in ONE launch -- the same convolution of every parallel
 * HRNet branch (pose_higher_hrnet.py:217-235): launched one by one, the low-resolution branches leave most CUs idle.  Member i = (d[i], src[i],
 * u[i] from cp_winograd24_pack_f32, scale[i], shift[i], res[i] or NULL, out[i]) as for cp_conv3x3_winograd_f32; outputs must not alias. */
int cp_conv3x3_winograd24_group_f32(const cp_conv_desc* d, int n, const float* const* src, const float* const* u,
                                    const float* const* scale, const float* const* shift, const float* const* res,
                                    float* const* out, void* stream);

/* One KeypointHead branch (lib/models/heads/keypoint.py:14-37: conv3x3(C -> head_conv, bias) -> ReLU -> conv1x1(head_conv -> n, bias))
 * with n <= 34 outputs -- all six branches of the reference head -- as ONE launch: the 1x1 is applied to every channel tile of the
 * Winograd kernel while it is on the CU (n <= 2: in the epilogue registers; more: a second MFMA phase over the LDS-resident tile),
 * the [B,H,W,head_conv] intermediate is never written.  d / src / u / scale / shift: the 3x3 conv as for
 * cp_conv3x3_winograd_f32 (C = 64, act = CP_ACT_RELU); w2: [n2][ld2] (ld2 >= head_conv, % 4 == 0), b2: [n2]; out2: NCHW
 * [B,n2,H,W]; act2: CP_ACT_SIGMOID for hm (lib/detectors/multi_pose.py:35-37), else CP_ACT_NONE.  Returns 1 for other shapes. */
int cp_head3x3_1x1_f32(const cp_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                       const float* w2, const float* b2, float* out2, int n2, int ld2, int act2, void* stream);

/* ---- fused DCNv2 forward -------------------------------------------------------------------------
 * Replaces dcn_v2_forward / dcn_v2_cuda_forward (DCNv2/src/dcn_v2.h:9-39, src/cuda/dcn_v2_cuda.cu:42-172,
 * src/cuda/dcn_v2_im2col_cuda.cu:25-54,125-195) plus the BN + ReLU of DeformConv (pose_dla_dcn.py:345-348).
 * x: NHWC [B,H,W,srcLd]; om: NHWC [B,Ho,Wo,omLd] with ch 2k = dy_k, 2k+1 = dx_k, 2*kh*kw + k = mask_k
 * (logits if omSigmoid, as produced by conv_offset_mask, dcn_v2.py:117-121; already-sigmoided otherwise);
 * w: [ldw][kh*kw*C]; deformable groups: `dg` below (the reference's models only use 1, pose_dla_dcn.py:343; the interface
 * implements any, dcn_v2_im2col_cuda.cu:153,162-164). */
typedef struct cp_dcn_desc {
    int B, H, W, C, srcLd;
    int Ho, Wo;
    int kh, kw, sy, sx, py, px, dily, dilx;
    int K, ldw, Cout;
    int omLd, omSigmoid;
    int outLd, outNCHW, act;
    int tile;
    int ksplit;     /* 0 / 1: none.  S > 1: split-K over the taps for small-M layers: `out` is a workspace [S][B*Ho*Wo][ldw] that receives
                     * the raw partial sums (scale = ones, shift = zeros, act = CP_ACT_NONE, NHWC, outLd = Cout = ldw);
                     * cp_splitk_reduce_f32 sums the S slices in a fixed order (deterministic) and applies scale / shift / act */
    int dg;         /* 0 / 1: one deformable group.  G > 1: input channel c samples with the offsets / mask of group g = c / (C / G)
                     * ((C / G) % 16 == 0): om channel 2 * (g * kh*kw + k) = dy, + 1 = dx, 2 * G * kh*kw + g * kh*kw + k = mask;
                     * omLd >= 3 * G * kh*kw */
} cp_dcn_desc;
int cp_dcn_v2_f32(const cp_dcn_desc* d, const float* x, const float* om, const float* w, const float* scale,
                  const float* shift, float* out, void* stream);
/* Kernel-side constants of a DCNv2 layer from the reference's parameters, one launch: w [Co][C][kh][kw] and bias [Co]
 * (nn.Parameter layout, DCNv2/dcn_v2.py:99-103) -> wp [ldw][kh*kw*Cp] (k = (tap*dg + g)*(Cp/dg) + c, zero padding), scale [ldw] = 1,
 * shift [ldw] = bias -- the `w`, `scale`, `shift` of cp_dcn_v2_f32.  The drop-in dcn_v2_forward (DCNv2/src/dcn_v2.h:9-23) calls it
 * on every forward: no packed-weight cache to go stale. */
int cp_dcn_pack_weights_f32(const float* w, const float* bias, int Co, int C, int kh, int kw, int dg, int Cp, int ldw, float* wp,
                            float* scale, float* shift, void* stream);
/* sizeof the two descriptor structs as THIS library was compiled: an FFI binder (ctypes, cgo, JNI) asserts its own layout against
 * them, and the plan runtime rejects plan files whose descriptor blobs have another size */
int cp_sizeof_conv_desc(void);
int cp_sizeof_dcn_desc(void);

/* second half of a split-K DCNv2 launch: out[m][n] = act((sum_s ws[s][m][n]) * scale[n] + shift[n]), n < Cout; NHWC out */
int cp_splitk_reduce_f32(const float* ws, int splits, int M, int ldw, const float* scale, const float* shift, int act, float* out,
                         int outLd, int Cout, void* stream);

/* ---- 7x7 / pad 3 stem on the NCHW 3-channel network input ----------------------------------------
 * Replaces base_layer Conv2d(3,16,k7,s1,p3)+BN+ReLU (pose_dla_dcn.py:228-232) and conv1 Conv2d(3,64,k7,s2,p3)
 * +BN+ReLU (msra_resnet.py:118-121).  x: NCHW [B,3,H,W]; w: [Cout][176] with k = ((c*7+ky)*8 + kx) (kx padded
 * to 8, zeros); out: NHWC [B,Ho,Wo,outLd].  (Cout, stride) in {16,64} x {1,2}. */
int cp_stem7x7_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int B, int H, int W,
                   int Cout, int stride, int outLd, int relu, void* stream);

/* ---- bandwidth-bound NHWC helpers --------------------------------------------------------------- */
/* nn.MaxPool2d (pose_dla_dcn.py:197-198 k2 s2; msra_resnet.py:123 k3 s2 p1) */
int cp_maxpool2d_nhwc_f32(const float* in, int inLd, float* out, int outLd, int B, int H, int W, int C, int k, int s, int p,
                          void* stream);
/* IDAUp: depthwise ConvTranspose2d(k=2f, s=f, p=f/2, groups=C) + add (pose_dla_dcn.py:360-377); w: [k*k][C] */
int cp_dw_deconv_add_nhwc_f32(const float* in, int inLd, const float* w, const float* add, int addLd, float* out, int outLd,
                              int B, int H, int W, int C, int f, void* stream);
/* HRNet fuse: out = relu?( sum_i nearest_upsample(src_i, 2^shift_i) ) (pose_higher_hrnet.py:217-235) */
int cp_sum_up_nhwc_f32(int n, const float* const* src, const int* ld, const int* shift, float* out, int outLd, int B, int H,
                       int W, int C, int relu, void* stream);
/* Up to four INDEPENDENT sums of that kind in ONE launch (the y_i = relu(sum_j fuse_ij(x_j)) of every output branch of an HRNet module,
 * pose_higher_hrnet.py:224-235); bit-identical to the single launches.  src: 4 pointers per member (NULL beyond its source count);
 * meta: 14 ints per member = nsrc, ld[4], shift[4], outLd, B, H, W, C; out: one NHWC output per member (they must not alias). */
int cp_sum_up_group_nhwc_f32(int n, const float* const* src, const int* meta, float* const* out, int relu, void* stream);
/* depthwise k x k conv + folded BN + activation (groups == channels): MobileNetV3 Block.conv2 (mobilenetv3.py:119-121, k 3 / 5,
 * stride 1 / 2) and ShuffleNetV2 banch1 / banch2 (shufflenetv2_dcn.py:67-88); w: [k*k][C], scale / shift: [C] */
int cp_dwconv2d_nhwc_f32(const float* in, int inLd, const float* w, const float* scale, const float* shift, float* out, int outLd,
                         int B, int H, int W, int C, int k, int s, int p, int act, void* stream);
/* SeModule (mobilenetv3.py:99-113): nn.AdaptiveAvgPool2d(1) of in [B,HW,C] -> out [B,C]; then out = x * se[b,c] (+ add:
 * the block's shortcut, mobilenetv3.py:141-143) */
int cp_global_avgpool_nhwc_f32(const float* in, int inLd, float* out, int outLd, int B, int HW, int C, void* stream);
int cp_scale_add_nhwc_f32(const float* x, int xLd, const float* se, int seLd, const float* add, int addLd, float* out, int outLd,
                          int B, int H, int W, int C, void* stream);
/* channel_shuffle(torch.cat((x1, x2), 1), 2) of ShuffleNetV2 (shufflenetv2_dcn.py:28-42,94-104); x1, x2: h channels each;
 * out: two halves of hp (>= h, multiple of 16) physical channels, logical channel c -> c (c < h) or hp + c - h; pads = 0 */
int cp_shuffle_concat_nhwc_f32(const float* x1, int ld1, const float* x2, int ld2, float* out, int outLd, long long npix, int h, int hp,
                               void* stream);
int cp_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, int outLd, int cOff, void* stream);
int cp_nhwc_to_nchw_f32(const float* in, int inLd, int cOff, float* out, int B, int C, int H, int W, void* stream);
int cp_fill_f32(float* p, float v, long long n, void* stream);
/* flip-test merge (multi_pose.py:45-53; models/utils.py:27-47): in [2,C,H,W] NCHW -> out [1,C,H,W];
 * mode 0 = W-flip average (hm, wh); 1 = + left/right joint swap (hm_hp); 2 = swap on (x,y) pairs + negate x (hps).
 * perm: device int[J] joint permutation (NULL for mode 0). */
int cp_flip_merge_f32(const float* in, float* out, int C, int H, int W, int mode, const int* perm, void* stream);

/* ---- device pre-/post-processing ---------------------------------------------------
 * The 8-bit pixel arithmetic is OpenCV's (opencv-python, unpinned, requirements.txt): resize.cpp / imgwarp.cpp fixed-point
 * bilinear paths, restated in oracle/prepost_np.py; parity against cv2 itself is unpinned (cv2 is not installable here).
 * cp_resize_u8: cv2.resize(image, (NW, NH)) with the default INTER_LINEAR (base_detector.py:47); DEVICE uint8 [H,W,3] ->
 *   DEVICE uint8 [NH,NW,3].
 * cp_preprocess_u8_f32: cv2.warpAffine(INTER_LINEAR, border 0) + (x/255 - mean)/std + HWC->CHW of
 *   lib/detectors/base_detector.py:48-56.  img: DEVICE uint8 [H,W,3]; M: HOST double[6], the 2x3 matrix given to
 *   warpAffine (source -> destination, `trans_input`); mean/std_: HOST float[3];
 *   out: DEVICE float32 NCHW [1 or 2,3,OH,OW]; flip != 0 also writes the mirrored twin as batch entry 1 (:57-58).
 * cp_transform_dets_f32: lib/utils/post_process.py:8-19 + lib/utils/image.py:19-24 on the device: the 2 box corners
 *   and J keypoints of dets[B,K,5+3J] through the per-image 2x3 double matrix trans[B][6], then / scale. */
int cp_resize_u8(const unsigned char* src, int H, int W, unsigned char* dst, int NH, int NW, void* stream);
int cp_preprocess_u8_f32(const unsigned char* img, int H, int W, const double* M, float* out, int OH, int OW, const float* mean,
                         const float* std_, int flip, void* stream);
int cp_transform_dets_f32(const float* dets, float* out, const double* trans, int B, int K, int J, float scale, void* stream);

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
 * Limits: K <= 256 (the reference's TEST.TOPK is 100), K <= H*W.  Any map size: planes up to 32768 keys are
 * LDS-resident in one piece, larger ones (FIX_RES = false inputs, TEST_SCALES > 1) stream through LDS in chunks. */
int cp_decode_workspace_bytes(int B, int J, int K, size_t* scores_bytes, size_t* inds_bytes);
int cp_multi_pose_decode_f32(const float* heat, const float* wh, const float* kps, const float* reg,
                             const float* hm_hp, const float* hp_offset, int B, int cat, int J, int H,
                             int W, int K, float* dets, float* ws_scores, int* ws_inds, void* stream);
/* The same decode as two launches (cp_multi_pose_decode_f32 = both, back to back), for schedules that want them as two graph
 * nodes: cp_decode_topk_f32 = _nms + _topk + _topk_channel (decode.py:10-16,87-115) over hm / hm_hp -> ws_scores / ws_inds; it needs
 * only those two heads, so a two-stream schedule runs it beside the remaining head convolutions.  cp_decode_assign_f32 = the gathers
 * and the keypoint-to-person assignment (decode.py:240-308) -> dets. */
int cp_decode_topk_f32(const float* heat, const float* hm_hp, int B, int cat, int J, int H, int W, int K, float* ws_scores,
                       int* ws_inds, void* stream);
int cp_decode_assign_f32(const float* wh, const float* kps, const float* reg, const float* hp_offset, const float* ws_scores,
                         const int* ws_inds, int B, int J, int H, int W, int K, float* dets, void* stream);

/* ---- plan handle: a whole network behind three calls -----------------------------
 * Replaces BackBoneWithHead.forward (lib/models/model.py:57-59: head_model(backbone_model(x))) for one compiled
 * (arch, B, H, W) and, with cp_plan_process, MultiPoseDetector.process (lib/detectors/multi_pose.py:29-60; flip test
 * excluded) -- no Python involved.  A plan file / blob (layout: centerpose_amd/plan.py; written by Engine.save_plan) holds
 * the packed, BN-folded, Winograd-transformed constants and the schedule of the launches above.
 * cp_plan_load / cp_plan_create are the ONLY entry points of this library that allocate device memory (activation
 * buffers, constants; owned by the handle, released by cp_plan_destroy) and they synchronise the device once.
 * use_graph != 0: the first cp_plan_forward runs the schedule eagerly, captures it into a hipGraph, later calls replay it.
 * cp_plan_forward: images = DEVICE float32 NCHW [B,3,H,W] (mean/std-normalised, base_detector.py:53-58) or NULL when the
 *   caller wrote cp_plan_input() itself; enqueues on `stream`, no host synchronisation after the first call.
 * cp_plan_output: device pointer + NCHW shape of head i of [hm, wh, hps, reg, hm_hp, hp_offset] (keypoint.py:40-42);
 *   hm and hm_hp are already sigmoided (multi_pose.py:35-37 fused into the head epilogue).
 * cp_plan_process: forward + cp_multi_pose_decode_f32 -> dets DEVICE float32 [B,K,5+3J]. */
typedef struct cp_plan cp_plan;
/* FNV-1a (32 bit) of a host buffer: the checksum a CPPLAN04 file carries over everything behind its 48-byte header */
unsigned int cp_fnv1a32(const void* data, size_t bytes);
int cp_plan_load(const char* path, int use_graph, cp_plan** out);
int cp_plan_create(const void* blob, size_t bytes, int use_graph, cp_plan** out);
int cp_plan_info(const cp_plan* plan, int* B, int* H, int* W, int* n_outputs, int* n_launches);
float* cp_plan_input(const cp_plan* plan);
int cp_plan_output(const cp_plan* plan, int i, float** dev_ptr, int shape[4]);
int cp_plan_forward(cp_plan* plan, const float* images, void* stream);
int cp_plan_process(cp_plan* plan, const float* images, int K, float* dets, void* stream);
int cp_plan_destroy(cp_plan* plan);

/* ---- steps in flight (round 6): the C-ABI form of MultiPoseDetector.process_stream / engine.EnginePipeline ----------------------
 * Replaces nothing in the reference (it runs one image at a time, synchronously: lib/detectors/base_detector.py:79-140,
 * multi_pose.py:29-60); it is how the timed arrangement of bench.py is reached without Python.
 * cp_plan_clone: another INSTANCE of a loaded plan -- own activation buffers / static input / outputs / detections, the same
 *   constants (shared; freed with the last of the plan and its clones).  Destroy clones with cp_plan_destroy like any plan.
 * cp_pipeline_create: `depth` (1..8) instances of ONE plan (the plan and its clones, compiled with the decode inside the schedule:
 *   Engine(decode_k = K)) -> a handle that captures all their launch lists into ONE hipGraph on first use (both capture streams carry
 *   launches of several instances).  Does not take ownership of the plans; destroy the pipeline before its plans.
 * cp_pipeline_process: one replay = one step of EVERY instance.  images[k]: DEVICE float32 NCHW [B,3,H,W] (NULL = the caller filled
 *   cp_plan_input(plans[k]) itself); dets[k]: DEVICE float32 [B,K,5+3J] out; per instance bit-identical to cp_plan_process.
 *   Enqueues on `stream`; the results of all instances are complete when the stream reaches the end of this call's work
 *   (throughput up, a batch's latency ~ depth x: INTEGRATION.md 3c). */
typedef struct cp_pipeline cp_pipeline;
int cp_plan_clone(const cp_plan* plan, cp_plan** out);
int cp_pipeline_create(cp_plan* const* plans, int depth, cp_pipeline** out);
int cp_pipeline_process(cp_pipeline* pipe, const float* const* images, int K, float* const* dets, void* stream);
int cp_pipeline_destroy(cp_pipeline* pipe);
/* stream-ordered device-to-device copy (for callers that keep a plan's outputs beyond the next forward) */
int cp_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);

/* ---- host: soft-NMS of merged results --------------------------------------------------------
 * Replaces soft_nms_39 (lib/external/nms.pyx:172-275; called from multi_pose.py:76-77).
 * boxes: HOST float32 [N,56], modified in place with the reference's quirks; keep: HOST int[N] or NULL. */
int cp_soft_nms_39(float* boxes, int N, float sigma, float Nt, float threshold, int method, int* keep, int* n_keep);

#ifdef __cplusplus
}
#endif
#endif
