"""Latency of the whole detector.run() (pre_process -> process -> post_process -> merge) per image on the GPU box, per stage, as the
reference's own timers report it (base_detector.py:138-140).  usage: run_latency.py [arch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from centerpose_amd import config, detector

archs = sys.argv[1:] or ["dla_34", "res_50", "hrnet"]
img = (np.random.RandomState(0).rand(480, 640, 3) * 255).astype(np.uint8)
for arch in archs:
    cfg = config.get_cfg(arch)
    det = detector.MultiPoseDetector(cfg)
    for _ in range(5):
        det.run(img)
    acc, n = {}, 30
    t0 = time.perf_counter()
    for _ in range(n):
        r = det.run(img)
        for k in ("tot", "load", "pre", "net", "dec", "post", "merge"):
            acc[k] = acc.get(k, 0.0) + r[k]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print("%-8s FLIP_TEST=%s FIX_RES=%s scales=%s  wall %.2f ms/image:  %s" % (
        arch, cfg.TEST.FLIP_TEST, cfg.TEST.FIX_RES, cfg.TEST.TEST_SCALES, wall,
        "  ".join("%s %.2f" % (k, acc[k] / n * 1e3) for k in ("tot", "load", "pre", "net", "dec", "post", "merge"))), flush=True)
