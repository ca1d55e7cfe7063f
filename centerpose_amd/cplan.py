"""ctypes binding of the C plan handle (`cp_plan_*`, include/centerpose_hip.h): run a compiled network -- and the whole
of MultiPoseDetector.process -- through the C ABI alone.  This module imports neither `engine` nor `ops`: what it drives is
exactly what a C/C++ caller of libcenterpose_hip.so gets.  torch is used only to hand device memory in and out.

    plan = CPlan("dla34_b16.cpplan")          # written by Engine.save_plan(...)
    heads = plan.forward(images)              # [hm, wh, hps, reg, hm_hp, hp_offset], NCHW views of the plan's buffers
    dets = plan.process(images, K=100)        # [B, K, 56]
    pipe = CPipeline(plan, depth=2)           # two instances (the plan + a clone sharing its constants) in ONE hipGraph
    dets_a, dets_b = pipe.process([images_a, images_b], K=100)
"""
import ctypes

import torch

from . import _lib


class CPlan:
    def __init__(self, path_or_bytes, use_graph=True, _clone_of=None):
        if not torch.cuda.is_available():
            raise _lib.CenterposeHipError("CPlan needs a HIP device; there is no CPU fallback")
        L = _lib.lib()
        L.cp_plan_input.restype = ctypes.c_void_p
        self._L = L
        self._h = ctypes.c_void_p()
        if _clone_of is not None:
            rc = L.cp_plan_clone(_clone_of._h, ctypes.byref(self._h))
        elif isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            blob = bytes(path_or_bytes)
            rc = L.cp_plan_create(blob, ctypes.c_size_t(len(blob)), int(use_graph), ctypes.byref(self._h))
        else:
            rc = L.cp_plan_load(str(path_or_bytes).encode(), int(use_graph), ctypes.byref(self._h))
        _lib.check(rc, "cp_plan_load")
        B, H, W, no, nl = (ctypes.c_int() for _ in range(5))
        _lib.check(L.cp_plan_info(self._h, ctypes.byref(B), ctypes.byref(H), ctypes.byref(W), ctypes.byref(no), ctypes.byref(nl)),
                   "cp_plan_info")
        self.B, self.H, self.W, self.n_outputs, self.n_launches = B.value, H.value, W.value, no.value, nl.value
        self.output_ptrs, self.output_shapes = [], []
        for i in range(self.n_outputs):
            p, shp = ctypes.c_void_p(), (ctypes.c_int * 4)()
            _lib.check(L.cp_plan_output(self._h, i, ctypes.byref(p), shp), "cp_plan_output")
            self.output_ptrs.append(p.value)
            self.output_shapes.append(tuple(shp))

    def _copy_out(self, i):
        """Head i as a fresh torch tensor (device-to-device copy out of the handle's buffer)."""
        shape = self.output_shapes[i]
        t = torch.empty(shape, dtype=torch.float32, device="cuda")
        _lib.check(self._L.cp_memcpy_d2d(_lib.ptr(t), ctypes.c_void_p(self.output_ptrs[i]), ctypes.c_size_t(t.numel() * 4), _lib.stream()),
                   "cp_memcpy_d2d")
        return t

    def forward(self, images):
        """images: float32 NCHW [B,3,H,W] on the device -> list of the six head tensors (copies)."""
        if tuple(images.shape) != (self.B, 3, self.H, self.W):
            raise ValueError("plan was compiled for input %s, got %s" % ((self.B, 3, self.H, self.W), tuple(images.shape)))
        _lib.check(self._L.cp_plan_forward(self._h, _lib.ptr(_lib.f32(images)), _lib.stream()), "cp_plan_forward")
        return [self._copy_out(i) for i in range(self.n_outputs)]

    def process(self, images, K=100):
        """forward + decode in one C call -> dets [B, K, 5 + 3J] (multi_pose.py:29-60 without the flip test)."""
        if tuple(images.shape) != (self.B, 3, self.H, self.W):
            raise ValueError("plan was compiled for input %s, got %s" % ((self.B, 3, self.H, self.W), tuple(images.shape)))
        J = self.output_shapes[4][1]
        dets = torch.empty((self.B, K, 5 + 3 * J), dtype=torch.float32, device="cuda")
        _lib.check(self._L.cp_plan_process(self._h, _lib.ptr(_lib.f32(images)), int(K), _lib.ptr(dets), _lib.stream()), "cp_plan_process")
        return dets

    def clone(self):
        """Another instance of this plan (`cp_plan_clone`): own activations and static buffers, the same constants."""
        return CPlan(None, _clone_of=self)

    def close(self):
        if self._h:
            self._L.cp_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CPipeline:
    """Steps in flight through the C ABI alone (`cp_pipeline_*`): `depth` instances of one plan -- `plan` and depth - 1 clones --
    captured into ONE hipGraph; `process([images_0, ...], K)` = one replay = one step of every instance, per instance bit-identical
    to `CPlan.process`.  The plan must have been compiled with the decode inside its schedule (`Engine(..., decode_k=K)`)."""

    def __init__(self, plan, depth=2):
        self._L = plan._L
        self.plans = [plan] + [plan.clone() for _ in range(depth - 1)]
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_void_p * depth)(*[q._h.value for q in self.plans])
        _lib.check(self._L.cp_pipeline_create(arr, depth, ctypes.byref(self._h)), "cp_pipeline_create")
        self.depth = depth

    def process(self, images, K=100):
        if len(images) != self.depth:
            raise ValueError("expected %d image batches" % self.depth)
        p0 = self.plans[0]
        for x in images:
            if tuple(x.shape) != (p0.B, 3, p0.H, p0.W):
                raise ValueError("plan was compiled for input %s, got %s" % ((p0.B, 3, p0.H, p0.W), tuple(x.shape)))
        J = p0.output_shapes[4][1]
        imgs = [_lib.f32(x) for x in images]
        dets = [torch.empty((p0.B, K, 5 + 3 * J), dtype=torch.float32, device="cuda") for _ in range(self.depth)]
        ia = (ctypes.c_void_p * self.depth)(*[x.data_ptr() for x in imgs])
        da = (ctypes.c_void_p * self.depth)(*[d.data_ptr() for d in dets])
        _lib.check(self._L.cp_pipeline_process(self._h, ia, int(K), da, _lib.stream()), "cp_pipeline_process")
        return dets

    def close(self):
        if self._h:
            self._L.cp_pipeline_destroy(self._h)
            self._h = ctypes.c_void_p()
        for q in self.plans[1:]:                      # the clones are this object's; plans[0] belongs to the caller
            q.close()
        self.plans = self.plans[:1]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
