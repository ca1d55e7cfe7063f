"""Detector entry points with the reference's names, arguments, return values and timing keys:
``BaseDetector`` (lib/detectors/base_detector.py:15-140) and ``MultiPoseDetector``
(lib/detectors/multi_pose.py:24-79), so the class drops in behind tools/demo.py:47-49,77 and
tools/evaluate.py:48-67 (``detector_factory['multi_pose'](cfg).run(path)``).

Differences, all documented in INTEGRATION.md:
* the model is the fused HIP engine (model.BackBoneWithHead); hm / hm_hp come out of the head
  kernel already sigmoided, so ``process`` has no separate sigmoid pass;
* the flip-test merge runs on the device (the reference bounces through numpy, models/utils.py:30-47);
* ``process`` accepts any batch size when FLIP_TEST is off (the reference is batch-1 by
  construction, multi_pose.py:63) -- that is the batched throughput path of BASELINE.json;
* image decoding / warpAffine need cv2 in the reference; here ``pre_process`` uses a small numpy
  bilinear warp (cv2 is not a dependency).
"""
import time

import numpy as np
import torch

from . import _lib, ops
from .decode import multi_pose_decode
from .model import create_model, load_model
from .post_process import get_affine_transform, multi_pose_post_process

FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # multi_pose.py:27


def _warp_affine_bilinear(img, M, out_w, out_h):
    """cv2.warpAffine(img, M, (out_w,out_h), flags=INTER_LINEAR) stand-in (border = 0), float math."""
    Mi = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.float64)
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    H, W = img.shape[:2]
    out = np.zeros((out_h, out_w, img.shape[2]), np.float64)
    for dy, dx, wgt in ((0, 0, (1 - fy) * (1 - fx)), (0, 1, (1 - fy) * fx), (1, 0, fy * (1 - fx)), (1, 1, fy * fx)):
        yy, xx = y0 + dy, x0 + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64)
        out += v * (wgt * ok)[..., None]
    return out


class BaseDetector(object):
    def __init__(self, cfg):
        print("Creating model...")
        self.model = create_model(cfg.MODEL.NAME, cfg.MODEL.HEAD_CONV, cfg)
        if cfg.TEST.MODEL_PATH:
            self.model = load_model(self.model, cfg.TEST.MODEL_PATH)
        self.model = self.model.to(torch.device("cuda"))
        self.model.eval()
        self.mean = np.array(cfg.DATASET.MEAN, dtype=np.float32).reshape(1, 1, 3)
        self.std = np.array(cfg.DATASET.STD, dtype=np.float32).reshape(1, 1, 3)
        self.max_per_image = 100
        self.num_classes = cfg.MODEL.NUM_CLASSES
        self.scales = cfg.TEST.TEST_SCALES
        self.cfg = cfg
        self.pause = True
        self.device_preprocess = True     # HIP warp/normalise kernel for uint8 HxWx3 inputs
        self.device_postprocess = True    # HIP inverse-affine kernel, one D2H copy of the mapped dets

    def pre_process(self, image, scale, meta=None):
        """base_detector.py:32-62 (image: HxWx3 uint8/float BGR array)."""
        height, width = image.shape[0:2]
        new_height, new_width = int(height * scale), int(width * scale)
        if self.cfg.TEST.FIX_RES:
            inp_height, inp_width = self.cfg.MODEL.INPUT_H, self.cfg.MODEL.INPUT_W
            c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
            s = max(height, width) * 1.0
        else:
            inp_height = (new_height | self.cfg.MODEL.PAD) + 1
            inp_width = (new_width | self.cfg.MODEL.PAD) + 1
            c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
            s = np.array([inp_width, inp_height], dtype=np.float32)
        trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
        if (new_height, new_width) != (height, width):     # cv2.resize stand-in: fold the resize into the warp
            trans_input = trans_input @ np.array([[width / new_width, 0, 0], [0, height / new_height, 0], [0, 0, 1]])
        if self.device_preprocess and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3:
            # warp + normalise + HWC->CHW (+ mirrored twin) in one HIP kernel
            import ctypes
            Mi = np.ascontiguousarray(np.linalg.inv(np.vstack([trans_input, [0, 0, 1]]))[:2], np.float32)   # out px -> src px
            img_d = torch.from_numpy(np.ascontiguousarray(image)).cuda()
            nb = 2 if self.cfg.TEST.FLIP_TEST else 1
            images = torch.empty((nb, 3, inp_height, inp_width), dtype=torch.float32, device="cuda")
            mean = np.ascontiguousarray(self.mean.reshape(3), np.float32)
            std = np.ascontiguousarray(self.std.reshape(3), np.float32)
            rc = _lib.lib().cp_preprocess_u8_f32(ctypes.c_void_p(img_d.data_ptr()), height, width,
                                                  Mi.ctypes.data_as(ctypes.c_void_p), _lib.ptr(images), inp_height, inp_width,
                                                  mean.ctypes.data_as(ctypes.c_void_p), std.ctypes.data_as(ctypes.c_void_p),
                                                  1 if nb == 2 else 0, _lib.stream())
            _lib.check(rc, "cp_preprocess_u8_f32")
        else:
            inp_image = _warp_affine_bilinear(image, trans_input, inp_width, inp_height)
            inp_image = ((inp_image / 255. - self.mean) / self.std).astype(np.float32)
            images = inp_image.transpose(2, 0, 1).reshape(1, 3, inp_height, inp_width)
            if self.cfg.TEST.FLIP_TEST:
                images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
            images = torch.from_numpy(np.ascontiguousarray(images))
        meta = {"c": c, "s": s, "out_height": inp_height // self.cfg.MODEL.DOWN_RATIO,
                "out_width": inp_width // self.cfg.MODEL.DOWN_RATIO}
        return images, meta

    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1):
        raise NotImplementedError

    def merge_outputs(self, detections):
        raise NotImplementedError

    def run(self, image_or_path_or_tensor, meta=None):
        """base_detector.py:79-140 -- same 7 timing buckets and result dict."""
        load_time, pre_time, net_time, dec_time, post_time = 0, 0, 0, 0, 0
        merge_time, tot_time = 0, 0
        start_time = time.time()
        pre_processed = False
        if isinstance(image_or_path_or_tensor, np.ndarray):
            image = image_or_path_or_tensor
        elif isinstance(image_or_path_or_tensor, str):
            image = _imread(image_or_path_or_tensor)
        else:
            image = image_or_path_or_tensor["image"][0].numpy()
            pre_processed_images = image_or_path_or_tensor
            pre_processed = True
        loaded_time = time.time()
        load_time += loaded_time - start_time
        detections = []
        for scale in self.scales:
            scale_start_time = time.time()
            if not pre_processed:
                images, meta = self.pre_process(image, scale, meta)
            else:
                images = pre_processed_images["images"][scale][0]
                meta = pre_processed_images["meta"][scale]
                meta = {k: v.numpy()[0] for k, v in meta.items()}
            images = images.to(torch.device("cuda"))
            torch.cuda.synchronize()
            pre_process_time = time.time()
            pre_time += pre_process_time - scale_start_time
            output, dets, forward_time = self.process(images, return_time=True)
            torch.cuda.synchronize()
            net_time += forward_time - pre_process_time
            decode_time = time.time()
            dec_time += decode_time - forward_time
            dets = self.post_process(dets, meta, scale)
            torch.cuda.synchronize()
            post_process_time = time.time()
            post_time += post_process_time - decode_time
            detections.append(dets)
        results = self.merge_outputs(detections)
        torch.cuda.synchronize()
        end_time = time.time()
        merge_time += end_time - post_process_time
        tot_time += end_time - start_time
        return {"results": {1: results}, "tot": tot_time, "load": load_time, "pre": pre_time, "net": net_time,
                "dec": dec_time, "post": post_time, "merge": merge_time}


def _imread(path):
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        if path.endswith(".npy"):
            return np.load(path)
        raise RuntimeError("cv2 is not available here: pass an HxWx3 numpy array (or a .npy path) to run()")


class MultiPoseDetector(BaseDetector):
    def __init__(self, cfg):
        super(MultiPoseDetector, self).__init__(cfg)
        self.flip_idx = FLIP_IDX
        perm = list(range(17))
        for a, b in FLIP_IDX:
            perm[a], perm[b] = b, a
        self._perm = torch.tensor(perm, dtype=torch.int32, device="cuda")

    def _flip_merge(self, t, mode):
        out = torch.empty((1,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
        rc = _lib.lib().cp_flip_merge_f32(_lib.ptr(t.contiguous()), _lib.ptr(out), t.shape[1], t.shape[2], t.shape[3], mode,
                                          _lib.c_void_p(self._perm.data_ptr()), _lib.stream())
        _lib.check(rc, "cp_flip_merge_f32")
        return out

    def process(self, images, return_time=False):
        """multi_pose.py:29-60.  images: float32 NCHW, mean/std-normalised, on the HIP device."""
        with torch.no_grad():
            torch.cuda.synchronize()
            outputs = self.model(images)            # hm (and hm_hp) already sigmoided (fused epilogue)
            hm, wh, hps, reg, hm_hp, hp_offset = outputs
            reg = reg if self.cfg.LOSS.REG_OFFSET else None
            hm_hp = hm_hp if self.cfg.LOSS.HM_HP else None
            hp_offset = hp_offset if self.cfg.LOSS.REG_HP_OFFSET else None
            torch.cuda.synchronize()
            forward_time = time.time()
            if self.cfg.TEST.FLIP_TEST:             # batch of exactly 2: image + mirrored twin
                hm = self._flip_merge(hm, 0)
                wh = self._flip_merge(wh, 0)
                hps = self._flip_merge(hps, 2)
                hm_hp = self._flip_merge(hm_hp, 1) if hm_hp is not None else None
                reg = reg[0:1] if reg is not None else None
                hp_offset = hp_offset[0:1] if hp_offset is not None else None
            dets = multi_pose_decode(hm, wh, hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset, K=self.cfg.TEST.TOPK)
        if return_time:
            return outputs, dets, forward_time
        return outputs, dets

    def post_process(self, dets, meta, scale=1):
        """multi_pose.py:62-71 (batch-1 by construction, like the reference)."""
        if self.device_postprocess and dets.is_cuda and dets.shape[2] == 56 and self.num_classes == 1:
            flat = dets.detach().reshape(1, -1, 56).contiguous()
            trans = get_affine_transform(meta["c"], meta["s"], 0, (meta["out_width"], meta["out_height"]), inv=1)
            td = torch.from_numpy(np.ascontiguousarray(trans, np.float64)).cuda()
            out = torch.empty_like(flat)
            rc = _lib.lib().cp_transform_dets_f32(_lib.ptr(flat), _lib.ptr(out), _lib.c_void_p(td.data_ptr()), 1, flat.shape[1], 17,
                                                  _lib.c_float(float(scale)), _lib.stream())
            _lib.check(rc, "cp_transform_dets_f32")
            return {1: out[0].cpu().numpy()}
        dets = dets.detach().cpu().numpy().reshape(1, -1, dets.shape[2])
        dets = multi_pose_post_process(dets.copy(), [meta["c"]], [meta["s"]], meta["out_height"], meta["out_width"])
        for j in range(1, self.num_classes + 1):
            dets[0][j] = np.array(dets[0][j], dtype=np.float32).reshape(-1, 56)
            dets[0][j][:, :4] /= scale
            dets[0][j][:, 5:39] /= scale
        return dets[0]

    def merge_outputs(self, detections):
        """multi_pose.py:73-79"""
        results = np.concatenate([detection[1] for detection in detections], axis=0).astype(np.float32)
        if self.cfg.TEST.NMS or len(self.cfg.TEST.TEST_SCALES) > 1:
            soft_nms_39(results, Nt=0.5, method=2)
        return results.tolist()


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """lib/external/nms.pyx:172-275 -- in place on a float32 [N,56] host array; returns keep list."""
    import ctypes
    assert boxes.dtype == np.float32 and boxes.flags["C_CONTIGUOUS"] and boxes.ndim == 2 and boxes.shape[1] == 56
    n = ctypes.c_int(0)
    keep = np.zeros(boxes.shape[0], np.int32)
    rc = _lib.lib().cp_soft_nms_39(boxes.ctypes.data_as(ctypes.c_void_p), int(boxes.shape[0]), ctypes.c_float(sigma),
                                   ctypes.c_float(Nt), ctypes.c_float(threshold), int(method),
                                   keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    _lib.check(rc, "cp_soft_nms_39")
    return keep[: n.value].tolist()


detector_factory = {"multi_pose": MultiPoseDetector}     # lib/detectors/detector_factory.py
