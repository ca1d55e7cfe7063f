"""Detector entry points with the reference's names, arguments, return values and timing keys:
``BaseDetector`` (lib/detectors/base_detector.py:15-140) and ``MultiPoseDetector``
(lib/detectors/multi_pose.py:24-79), so the class drops in behind tools/demo.py:47-49,77 and
tools/evaluate.py:48-67 (``detector_factory['multi_pose'](cfg).run(path)``).

What is kept is the CONTRACT -- method names and signatures, the meta dict, the seven timing buckets and the result
dict of ``run`` -- not the reference's statement order: the stages are organised around a small stage clock and an input
source object, every pixel / tensor stage runs on the device:

* ``pre_process``: upload of the uint8 image, then cv2.resize / cv2.warpAffine / normalise / HWC->CHW (+ mirrored twin) as
  two HIP kernels that restate OpenCV's 8-bit fixed-point arithmetic (csrc/prepost.hip) -- cv2 is not a dependency;
* ``process``: fused HIP engine (hm / hm_hp leave the head kernel already sigmoided), device flip-test merge (the reference
  bounces through numpy, models/utils.py:30-47), HIP decode; any batch size when FLIP_TEST is off (the reference is batch-1
  by construction, multi_pose.py:63) -- that is the batched throughput path of BASELINE.json;
* ``post_process``: inverse affine of boxes / keypoints as one HIP kernel, one D2H copy;
* ``merge_outputs``: soft-NMS on the host (C++, csrc/host_nms.cpp).
"""
import ctypes
import time

import numpy as np
import torch

from . import _lib
from .decode import multi_pose_decode
from .model import create_model, load_model
from .post_process import get_affine_transform

FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # multi_pose.py:27


class StageClock:
    """The seven wall-clock buckets of BaseDetector.run (base_detector.py:80-81,138-140).  `lap(key)` synchronises the
    device and charges the time since the previous mark to `key`; `split(key, at)` charges up to an externally taken
    time stamp (process() reports when the forward finished)."""
    KEYS = ("tot", "load", "pre", "net", "dec", "post", "merge")

    def __init__(self, sync):
        self._sync = sync
        self.t = dict.fromkeys(self.KEYS, 0)
        self._start = self._mark = time.time()

    def lap(self, key, sync=True):
        if sync:
            self._sync()
        now = time.time()
        self.t[key] += now - self._mark
        self._mark = now

    def split(self, key, at):
        self.t[key] += at - self._mark
        self._mark = at

    def report(self, results):
        self.t["tot"] += self._mark - self._start
        out = {"results": {1: results}}
        out.update(self.t)
        return out


class _ArraySource:
    """run() input that still needs pre-processing: an HxWx3 uint8 BGR array (what cv2.imread returns)."""

    def __init__(self, image):
        self.image = image

    def at_scale(self, det, scale, meta):
        return det.pre_process(self.image, scale, meta)


class _PreparedSource:
    """run() input from the evaluation data loader: {'image', 'images': {scale: tensor}, 'meta': {scale: dict of tensors}}
    (base_detector.py:91-93,103-106)."""

    def __init__(self, item):
        self.item = item

    def at_scale(self, det, scale, meta):
        images = self.item["images"][scale][0]
        meta = {k: v.numpy()[0] for k, v in self.item["meta"][scale].items()}
        return images, meta


def _imread(path):
    try:
        import cv2
    except ImportError:
        if path.endswith(".npy"):
            return np.load(path)
        raise RuntimeError("cv2 is not available here: pass an HxWx3 uint8 numpy array (or a .npy path) to run()")
    return cv2.imread(path)


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

    # -- geometry of one (image, scale) ----------------------------------------------------------------------------------
    def input_geometry(self, height, width, scale):
        """(new_h, new_w, inp_h, inp_w, c, s) of base_detector.py:33-46: the resized image size, the network input size and
        the centre / extent the affine maps onto it.  FIX_RES: fixed INPUT_H x INPUT_W, longer image side fills it;
        otherwise the resized image is padded up to the next multiple of PAD + 1 and centred."""
        cfg = self.cfg
        new_h, new_w = int(height * scale), int(width * scale)
        if cfg.TEST.FIX_RES:
            return (new_h, new_w, cfg.MODEL.INPUT_H, cfg.MODEL.INPUT_W,
                    np.array([new_w / 2., new_h / 2.], dtype=np.float32), max(height, width) * 1.0)
        inp_h, inp_w = (new_h | cfg.MODEL.PAD) + 1, (new_w | cfg.MODEL.PAD) + 1
        return (new_h, new_w, inp_h, inp_w, np.array([new_w // 2, new_h // 2], dtype=np.float32),
                np.array([inp_w, inp_h], dtype=np.float32))

    def pre_process(self, image, scale, meta=None):
        """base_detector.py:32-62.  image: HxWx3 uint8 BGR host array -> (images float32 [1 or 2,3,inp_h,inp_w] ON THE
        DEVICE, meta).  cv2.resize + cv2.warpAffine + normalise + transpose (+ flipped twin) run as HIP kernels."""
        if not (isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3):
            raise _lib.CenterposeHipError("pre_process expects an HxWx3 uint8 array (cv2.imread layout); there is no host fallback")
        height, width = image.shape[0:2]
        new_h, new_w, inp_h, inp_w, c, s = self.input_geometry(height, width, scale)
        L = _lib.lib()
        src = torch.from_numpy(np.ascontiguousarray(image)).cuda()
        if (new_h, new_w) != (height, width):                  # base_detector.py:47 (cv2.resize copies when the size is unchanged)
            resized = torch.empty((new_h, new_w, 3), dtype=torch.uint8, device="cuda")
            _lib.check(L.cp_resize_u8(ctypes.c_void_p(src.data_ptr()), height, width, ctypes.c_void_p(resized.data_ptr()),
                                      new_h, new_w, _lib.stream()), "cp_resize_u8")
            src = resized
        trans_input = np.ascontiguousarray(get_affine_transform(c, s, 0, [inp_w, inp_h]), np.float64)
        nb = 2 if self.cfg.TEST.FLIP_TEST else 1
        images = torch.empty((nb, 3, inp_h, inp_w), dtype=torch.float32, device="cuda")
        mean = np.ascontiguousarray(self.mean.reshape(3), np.float32)
        std = np.ascontiguousarray(self.std.reshape(3), np.float32)
        _lib.check(L.cp_preprocess_u8_f32(ctypes.c_void_p(src.data_ptr()), new_h, new_w, trans_input.ctypes.data_as(ctypes.c_void_p),
                                          _lib.ptr(images), inp_h, inp_w, mean.ctypes.data_as(ctypes.c_void_p),
                                          std.ctypes.data_as(ctypes.c_void_p), 1 if nb == 2 else 0, _lib.stream()),
                   "cp_preprocess_u8_f32")
        down = self.cfg.MODEL.DOWN_RATIO
        return images, {"c": c, "s": s, "out_height": inp_h // down, "out_width": inp_w // down}

    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1):
        raise NotImplementedError

    def merge_outputs(self, detections):
        raise NotImplementedError

    def _source(self, x):
        if isinstance(x, np.ndarray):
            return _ArraySource(x)
        if isinstance(x, str):
            return _ArraySource(_imread(x))
        return _PreparedSource(x)

    def run(self, image_or_path_or_tensor, meta=None):
        """base_detector.py:79-140: every TEST_SCALES entry through pre_process -> process -> post_process, then
        merge_outputs; returns {'results': {1: rows}, 'tot', 'load', 'pre', 'net', 'dec', 'post', 'merge'}."""
        clock = StageClock(torch.cuda.synchronize)
        source = self._source(image_or_path_or_tensor)
        clock.lap("load", sync=False)
        per_scale = []
        for scale in self.scales:
            images, meta = source.at_scale(self, scale, meta)
            images = images.to(torch.device("cuda"))
            clock.lap("pre")
            _, dets, forward_done = self.process(images, return_time=True)
            clock.split("net", forward_done)
            clock.lap("dec")
            per_scale.append(self.post_process(dets, meta, scale))
            clock.lap("post")
        results = self.merge_outputs(per_scale)
        clock.lap("merge")
        return clock.report(results)


class MultiPoseDetector(BaseDetector):
    def __init__(self, cfg):
        super(MultiPoseDetector, self).__init__(cfg)
        self.flip_idx = FLIP_IDX
        perm = list(range(17))
        for a, b in FLIP_IDX:
            perm[a], perm[b] = b, a
        self._perm = torch.tensor(perm, dtype=torch.int32, device="cuda")

    def _flip_merge(self, t, mode):
        if t.shape[0] != 2:      # the reference's hm[0:1] + flip(hm[1:2]) (multi_pose.py:45-53) is only defined for image + mirrored twin
            raise ValueError("FLIP_TEST needs a batch of exactly 2 (image + mirrored twin), got %d" % t.shape[0])
        out = torch.empty((1,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
        rc = _lib.lib().cp_flip_merge_f32(_lib.ptr(t.contiguous()), _lib.ptr(out), t.shape[1], t.shape[2], t.shape[3], mode,
                                          _lib.c_void_p(self._perm.data_ptr()), _lib.stream())
        _lib.check(rc, "cp_flip_merge_f32")
        return out

    def _one_replay_path(self):
        """forward + decode can be ONE graph replay: nothing to do between them (no flip merge, no head gated off)."""
        loss = self.cfg.LOSS
        return (not self.cfg.TEST.FLIP_TEST and loss.REG_OFFSET and loss.HM_HP and loss.REG_HP_OFFSET and not loss.MSE_LOSS
                and hasattr(self.model, "process"))

    def process_stream(self, batches, depth=2):
        """`process` over an iterable of image batches with `depth` steps in flight (model.BackBoneWithHead.process_many): a
        generator of `(outputs, dets)` per batch, in order, each bit-identical to `process(batch)`.  Two consecutive batches are
        captured into one hipGraph so that one step's kernels fill the other's launch gaps: throughput up, a batch's result
        available only with its group (latency ~ depth x).  Configurations that need host logic between forward and decode
        (FLIP_TEST, a head gated off by cfg.LOSS) run batch by batch through `process`.  The batched-throughput entry point of
        BASELINE.json's metric; the reference has one image at a time (multi_pose.py:29-60, base_detector.py:79-140)."""
        if self._one_replay_path() and hasattr(self.model, "process_many"):
            # (no torch.no_grad() around the yields: a grad mode entered inside a generator leaks into the consumer between
            # yields; nothing on this path records autograd history anyway -- raw HIP launches and a clone of a plain tensor)
            for r in self.model.process_many(batches, self.cfg.TEST.TOPK, depth):
                yield r
            return
        for images in batches:
            yield self.process(images)

    def process(self, images, return_time=False):
        """multi_pose.py:29-60.  images: float32 NCHW, mean/std-normalised, on the HIP device."""
        if not return_time and self._one_replay_path():
            # no stage timing asked for and nothing to do between forward and decode: both in ONE graph replay (the peak
            # extraction overlaps the last head convolutions).  `run()` keeps the two-stage form for its 'net' / 'dec' timers.
            # `dets` is a fresh tensor on both paths (the reference returns one); `outputs` are the plan's static buffers.
            with torch.no_grad():
                outputs, dets = self.model.process(images, self.cfg.TEST.TOPK)
            return outputs, dets
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
        """multi_pose.py:62-71 (batch-1 by construction, like the reference): feature-map pixels -> image pixels / scale,
        {1: float32 [N,56]}.  The per-point Python loop of utils/image.py:19-24 is one HIP kernel."""
        if self.num_classes != 1 or dets.shape[2] != 56:
            raise _lib.CenterposeHipError("multi_pose post_process handles one class and 17 joints (dets [..,56])")
        flat = dets.detach().reshape(1, -1, 56).contiguous()
        inv = get_affine_transform(meta["c"], meta["s"], 0, (meta["out_width"], meta["out_height"]), inv=1)
        inv_d = torch.from_numpy(np.ascontiguousarray(inv, np.float64)).cuda()
        mapped = torch.empty_like(flat)
        rc = _lib.lib().cp_transform_dets_f32(_lib.ptr(flat), _lib.ptr(mapped), _lib.c_void_p(inv_d.data_ptr()), 1, flat.shape[1], 17,
                                              _lib.c_float(float(scale)), _lib.stream())
        _lib.check(rc, "cp_transform_dets_f32")
        return {1: mapped[0].cpu().numpy()}

    def merge_outputs(self, detections):
        """multi_pose.py:73-79: rows of every scale stacked; soft-NMS when configured or when several scales were run."""
        rows = np.ascontiguousarray(np.concatenate([d[1] for d in detections], axis=0), dtype=np.float32)
        if self.cfg.TEST.NMS or len(self.cfg.TEST.TEST_SCALES) > 1:
            soft_nms_39(rows, Nt=0.5, method=2)
        return rows.tolist()


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """lib/external/nms.pyx:172-275 -- in place on a float32 [N,56] host array; returns keep list."""
    assert boxes.dtype == np.float32 and boxes.flags["C_CONTIGUOUS"] and boxes.ndim == 2 and boxes.shape[1] == 56
    n = ctypes.c_int(0)
    keep = np.zeros(boxes.shape[0], np.int32)
    rc = _lib.lib().cp_soft_nms_39(boxes.ctypes.data_as(ctypes.c_void_p), int(boxes.shape[0]), ctypes.c_float(sigma),
                                   ctypes.c_float(Nt), ctypes.c_float(threshold), int(method),
                                   keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    _lib.check(rc, "cp_soft_nms_39")
    return keep[: n.value].tolist()


detector_factory = {"multi_pose": MultiPoseDetector}     # lib/detectors/detector_factory.py
