"""centerpose_amd -- MI355X (gfx950) native inference hot path for CenterNet multi-person pose.

Python mirrors of the reference's entry points (tensorboy/centerpose) over a C-ABI HIP library:
  multi_pose_decode            <- lib/models/decode.py:235-308
  MultiPoseDetector.process    <- lib/detectors/multi_pose.py:29-60
  create_model / load_model    <- lib/models/model.py:63-120
  dcn_v2_forward (module _ext) <- lib/models/backbones/DCNv2/src/dcn_v2.h:9-39
"""
from . import decode  # noqa: F401
from .decode import multi_pose_decode  # noqa: F401

__all__ = ["multi_pose_decode", "decode"]
