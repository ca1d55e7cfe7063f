"""GPU: the fused HIP engine / detector against (a) golden outputs of the imported reference modules,
(b) the torch-CPU oracle on the same seeded checkpoint and inputs.

Tolerances (north_star: "heatmap/offset floats within 1e-3", indices bit-exact on identical decode
inputs): post-sigmoid hm / hm_hp, reg, hp_offset: |err| <= 1e-3 absolute; wh / hps (tens of pixels):
|err| <= 1e-3 * max|ref|.  fp32-in / fp32-accumulate MFMA actually lands ~1e-5."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import decode_np, nets_torch

pytestmark = pytest.mark.gpu
ARCHS = ["dla_34", "res_50", "hrnet", "mobilenetv3", "shufflenetV2", "resdcn_18"]


def _check_heads(outs, refs):
    names = ["hm", "wh", "hps", "reg", "hm_hp", "hp_offset"]
    for n, o, r in zip(names, outs, refs):
        o, r = o.detach().cpu().double(), r.detach().cpu().double() if torch.is_tensor(r) else torch.from_numpy(r).double()
        if n in ("hm", "hm_hp"):
            o, r = torch.sigmoid(o), torch.sigmoid(r)
        tol = 1e-3 if n in ("hm", "hm_hp", "reg", "hp_offset") else 1e-3 * r.abs().max().item()
        err = (o - r).abs().max().item()
        assert err <= tol, "%s: max err %.3e > %.3e" % (n, err, tol)


@pytest.mark.parametrize("arch", ARCHS + ["resdcn_50"])
def test_forward_matches_reference_golden(arch, golden_dir):
    from centerpose_amd import engine, synth
    g = np.load(os.path.join(golden_dir, "net_%s_128.npz" % arch))
    refs = [g["out%d" % i] for i in range(6)]
    eng = engine.Engine(arch, synth.make_state_dict(arch), 1, 128, 128, sigmoid_heads=False, use_graph=False)
    outs = eng(synth.make_images(1, 128, 128, seed=7).cuda())
    torch.cuda.synchronize()
    _check_heads(outs, refs)


@pytest.mark.parametrize("arch,B,hw", [("dla_34", 3, (160, 96)), ("res_50", 2, (96, 160)), ("hrnet", 2, (64, 128)),
                                       ("mobilenetv3", 2, (96, 160)), ("shufflenetV2", 3, (160, 96)), ("resdcn_18", 3, (96, 160)),
                                       ("resdcn_50", 2, (160, 96)), ("resdcn_34", 2, (96, 160)), ("resdcn_101", 2, (160, 96))])
def test_forward_matches_oracle_ragged_shapes(arch, B, hw):
    """non-square inputs, batch > 1, hipGraph replay (twice: static buffers must be reusable)."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict(arch, seed=11)
    x = synth.make_images(B, hw[0], hw[1], seed=3)
    refs = nets_torch.forward(arch, sd, x)
    eng = engine.Engine(arch, sd, B, hw[0], hw[1], sigmoid_heads=False, use_graph=True)
    eng(x.cuda())
    outs = eng(x.cuda())
    torch.cuda.synchronize()
    _check_heads(outs, refs)


def test_dla34_full_resolution_vs_oracle():
    """BASELINE config shape (512x512), one image: every head within tolerance of the CPU oracle."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict("dla_34")
    x = synth.make_images(1)
    refs = nets_torch.forward("dla_34", sd, x)
    eng = engine.Engine("dla_34", sd, 1, 512, 512, sigmoid_heads=False, use_graph=False)
    outs = eng(x.cuda())
    torch.cuda.synchronize()
    _check_heads(outs, refs)


@pytest.mark.parametrize("arch", ["dla_34", "res_50"])
def test_process_end_to_end(arch):
    """MultiPoseDetector.process: (1) decode of the engine's own maps is bit-exact vs the oracle decode,
    (2) detections agree with the all-CPU oracle path wherever its score gaps exceed the tolerance."""
    from centerpose_amd import config, detector, synth
    cfg = config.get_cfg(arch, TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    B = 2
    x = synth.make_images(B, 256, 256, seed=5)
    outputs, dets = det.process(x.cuda())
    torch.cuda.synchronize()
    o = [t.cpu().numpy() for t in outputs]
    dets = dets.cpu().numpy()
    assert dets.shape == (B, 100, 56)
    # (1) identical decode inputs -> bit-exact
    ref_same = decode_np.multi_pose_decode(o[0], o[1], o[2], o[3], o[4], o[5], K=100)
    assert np.array_equal(dets, ref_same)
    # (2) full CPU oracle path
    _, ref = nets_torch.process(arch, det.model.state_dict(), x, K=100)
    sc = ref[..., 4].astype(np.float64)
    gap = np.minimum(np.abs(np.diff(sc, axis=1, prepend=np.inf)), np.abs(np.diff(sc, axis=1, append=-np.inf)))
    stable = gap > 2e-4          # 2 x the observed fp32 forward error bound (1e-4)
    close = np.isclose(dets[..., 5:39][stable], ref[..., 5:39][stable], atol=2e-2)
    # shown with `pytest -s` / on failure: a regression from 99 % to 51 % stable must be visible, not just still-passing
    print("%s: %d / %d detections compared (stable fraction %.3f); max |d score| %.2e, max |d box| %.2e, keypoint coords "
          "within 2e-2: %.4f" % (arch, int(stable.sum()), stable.size, stable.mean(),
                                 np.abs(dets[..., 4][stable] - ref[..., 4][stable]).max(),
                                 np.abs(dets[..., :4][stable] - ref[..., :4][stable]).max(), close.mean()))
    # the stable fraction is a property of the ORACLE's scores (seeded checkpoint + images): 0.965 for dla_34, 0.870 for res_50.
    # Floor = that minus a margin: a change that leaves only half of the detections compared must fail.
    floor = {"dla_34": 0.93, "res_50": 0.83}[arch]
    assert stable.mean() > floor, "only %.3f of the detections have a score gap > 2e-4 (expected > %.2f)" % (stable.mean(), floor)
    assert np.allclose(dets[..., 4][stable], ref[..., 4][stable], atol=1e-3)
    assert np.allclose(dets[..., :4][stable], ref[..., :4][stable], atol=2e-2)
    # keypoint coordinates: allow rare accept/reject flips at a threshold
    assert close.mean() > 0.995


def test_resdcn_through_the_detector():
    """`resdcn_18` (resnet_dcn.py: the reference's own factory cannot construct it, model.py:52 vs resnet_dcn.py:284) behind the
    same entry points: create_model -> MultiPoseDetector.process; the dict-of-heads of PoseResNet.forward arrives as the six-tensor
    list, dets bit-equal to the oracle decode of the engine's maps, heads within the 1e-3 bar of the oracle network."""
    from centerpose_amd import config, detector, synth
    cfg = config.get_cfg("resdcn_18", TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    assert "conv1.weight" in det.model.state_dict() and "hm.2.bias" in det.model.state_dict()
    x = synth.make_images(2, 256, 192, seed=8)
    outputs, dets = det.process(x.cuda())
    torch.cuda.synchronize()
    o = [t.cpu().numpy() for t in outputs]
    assert np.array_equal(dets.cpu().numpy(), decode_np.multi_pose_decode(o[0], o[1], o[2], o[3], o[4], o[5], K=100))
    refs = nets_torch.forward("resdcn_18", det.model.state_dict(), x)
    for n, got, r in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset"), outputs, refs):
        got, r = got.cpu().double(), r.double()
        if n in ("hm", "hm_hp"):
            r = torch.sigmoid(r)
        tol = 1e-3 if n in ("hm", "hm_hp", "reg", "hp_offset") else 1e-3 * r.abs().max().item()
        assert (got - r).abs().max().item() <= tol, n


def test_process_flip_test_matches_oracle():
    """FLIP_TEST path (multi_pose.py:45-53) with the merge on the device."""
    from centerpose_amd import config, detector, synth
    cfg = config.get_cfg("res_50", TEST__FLIP_TEST=True)
    det = detector.MultiPoseDetector(cfg)
    img = synth.make_images(1, 128, 128, seed=9)
    x = torch.cat([img, torch.flip(img, [3])], 0)
    outputs, dets = det.process(x.cuda())
    torch.cuda.synchronize()
    o = [t.cpu().numpy() for t in outputs]
    merged = decode_np.flip_merge(*o)
    ref = decode_np.multi_pose_decode(*merged, K=100)
    assert dets.shape == (1, 100, 56)
    assert np.array_equal(dets.cpu().numpy(), ref)


def test_detector_run_dict():
    """run(): same result dict / timing keys as base_detector.py:138-140."""
    from centerpose_amd import config, detector
    cfg = config.get_cfg("res_50", TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    img = (np.random.RandomState(0).rand(300, 400, 3) * 255).astype(np.uint8)
    ret = det.run(img)
    assert set(ret) == {"results", "tot", "load", "pre", "net", "dec", "post", "merge"}
    res = np.array(ret["results"][1])
    assert res.shape == (100, 56)


def test_engine_determinism():
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict("dla_34")
    x = synth.make_images(2, 128, 128).cuda()
    eng = engine.Engine("dla_34", sd, 2, 128, 128)
    a = [t.clone() for t in eng(x)]
    b = eng(x)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(a, b))


@pytest.mark.parametrize("arch", ARCHS)
def test_plan_roundtrip(arch, tmp_path):
    """On-disk plan: save the compiled plan, reload it without checkpoint / packing, same bits out,
    eagerly and through a captured hipGraph."""
    from centerpose_amd import engine, synth
    x = synth.make_images(2, 128, 128).cuda()
    eng = engine.Engine(arch, synth.make_state_dict(arch), 2, 128, 128, use_graph=False)
    ref = [t.clone() for t in eng(x)]
    path = str(tmp_path / "plan.cpplan")
    eng.save_plan(path)
    assert all(torch.equal(p, q) for p, q in zip(ref, eng(x)))          # recording did not disturb the engine
    del eng
    for use_graph in (False, True):
        e2 = engine.Engine.from_plan(path, use_graph=use_graph)
        assert (e2.arch, e2.B, e2.H, e2.W) == (engine.nets.canonical_arch(arch), 2, 128, 128)
        out = e2(x)
        torch.cuda.synchronize()
        assert len(out) == 6 and all(torch.equal(p, q) for p, q in zip(ref, out))
        assert len(e2.profile(iters=1)) == len(e2.launches)


def test_deterministic_plan_same_bytes_before_and_after_capture(tmp_path, monkeypatch):
    """save_plan(deterministic=True): scheduled from the engine's EMISSION order with model durations, so the bytes do
    not depend on whether -- and in which measured order -- the engine has captured its graph; two engines of the same checkpoint
    write the same file; the flag raises when it cannot be honoured (CP_SCHED=0); the plan still runs to the engine's bits."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict("dla_34")
    x = synth.make_images(2, 128, 128).cuda()
    eng = engine.Engine("dla_34", sd, 2, 128, 128, use_graph=True)
    pa, pb, pc = (str(tmp_path / n) for n in ("a.cpplan", "b.cpplan", "c.cpplan"))
    eng.save_plan(pa, deterministic=True)                      # before capture
    ref = [t.clone() for t in eng(x)]                          # captures: measured critical-path order
    torch.cuda.synchronize()
    assert eng.graph is not None and eng.launches is not eng.emission
    eng.save_plan(pb, deterministic=True)                      # after capture
    engine.Engine("dla_34", sd, 2, 128, 128, use_graph=False).save_plan(pc, deterministic=True)
    a, b, c = (open(q, "rb").read() for q in (pa, pb, pc))
    assert a == b and a == c
    out = engine.Engine.from_plan(pb, use_graph=False)(x)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(ref, out))
    monkeypatch.setenv("CP_SCHED", "0")
    with pytest.raises(ValueError):
        eng.save_plan(pa, deterministic=True)


C_PLAN_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from centerpose_amd import cplan
assert "centerpose_amd.engine" not in sys.modules and "centerpose_amd.ops" not in sys.modules
x = torch.from_numpy(np.load(sys.argv[3])).cuda()
for use_graph in (0, 1):
    p = cplan.CPlan(sys.argv[2], use_graph=use_graph)
    for rep in range(3):                      # eager / capture / replay
        heads = p.forward(x)
        dets = p.process(x, K=100)
    torch.cuda.synchronize()
    np.savez(sys.argv[4] + str(use_graph), dets=dets.cpu().numpy(), **{"h%d" % i: h.cpu().numpy() for i, h in enumerate(heads)})
    p.close()
assert "centerpose_amd.engine" not in sys.modules and "centerpose_amd.ops" not in sys.modules
print("child ok", p.n_launches)
"""


@pytest.mark.parametrize("arch,B,hw", [("dla_34", 2, 128), ("hrnet", 2, 128), ("mobilenetv3", 2, 128), ("shufflenetV2", 2, 128),
                                       ("dla_34", 4, 512)])      # 4 x 512 x 512: the fused head launches (cp_head3x3_1x1_f32) are in the plan
def test_c_plan_handle_runs_network_without_engine(arch, B, hw, tmp_path):
    """cp_plan_load / cp_plan_forward / cp_plan_process / cp_plan_destroy.  A fresh interpreter that imports
    neither engine.py nor ops.py runs the plan file through the C ABI alone; heads and dets equal the Python engine's bits."""
    import subprocess
    import sys
    from centerpose_amd import engine, synth
    from centerpose_amd.decode import multi_pose_decode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    x = synth.make_images(B, hw, hw, seed=21)
    eng = engine.Engine(arch, synth.make_state_dict(arch), B, hw, hw, use_graph=False)
    assert (hw == 512) == any(l.fn == "cp_head3x3_1x1_f32" for _, _, _, l in eng.launches)
    ref = [t.clone() for t in eng(x.cuda())]
    ref_dets = multi_pose_decode(ref[0], ref[1], ref[2], reg=ref[3], hm_hp=ref[4], hp_offset=ref[5], K=100)
    path, xin, outp = str(tmp_path / "p.cpplan"), str(tmp_path / "x.npy"), str(tmp_path / "out")
    eng.save_plan(path)
    assert eng.stream_plan is not None and 1 in eng.stream_plan          # the file carries the two-stream schedule
    ref = [t.clone() for t in eng(x.cuda())]                              # ... and the engine still computes the same bits in that order
    np.save(xin, x.numpy())
    r = subprocess.run([sys.executable, "-c", C_PLAN_CHILD, root, path, xin, outp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for g in (0, 1):
        got = np.load(outp + "%d.npz" % g)
        for i in range(6):
            assert np.array_equal(got["h%d" % i], ref[i].cpu().numpy()), (g, i)
        assert np.array_equal(got["dets"], ref_dets.cpu().numpy())


@pytest.mark.parametrize("arch,depth", [("dla_34", 2), ("hrnet", 3)])
def test_c_pipeline_steps_in_flight_equal_single_plan(arch, depth, tmp_path):
    """Steps in flight through the C ABI alone (round 6: cp_plan_clone / cp_pipeline_create / cp_pipeline_process / cp_pipeline_destroy,
    include/centerpose_hip.h) -- the C form of MultiPoseDetector.process_stream: `depth` instances of one plan file (clones share the
    constants) captured into ONE hipGraph by the C runtime.  Per instance the detections equal cp_plan_process of the same batch bit for
    bit, over several replays with changing inputs; the clones outlive the plan they were cloned from (shared constants)."""
    from centerpose_amd import cplan, engine, synth
    B, H, W = 2, 128, 96
    eng = engine.Engine(arch, synth.make_state_dict(arch, seed=9), B, H, W, use_graph=True, decode_k=100)
    path = str(tmp_path / "p.cpplan")
    eng.save_plan(path)
    imgs = [synth.make_images(B, H, W, seed=80 + i).cuda() for i in range(2 * depth)]
    want = [eng.process(x)[1].clone() for x in imgs]                      # the Python engine ...
    plan = cplan.CPlan(path)
    single = [plan.process(x, 100).clone() for x in imgs]                 # ... and the single C plan agree
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(single, want))
    pipe = cplan.CPipeline(plan, depth=depth)
    got = []
    for r in range(2):                                                    # first call: warm-up + capture; second: graph replay
        got += pipe.process(imgs[r * depth:(r + 1) * depth], 100)
    again = pipe.process(imgs[:depth], 100)                               # replay with the first inputs again
    torch.cuda.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g, w), "instance %d of replay %d differs from the single plan" % (i % depth, i // depth)
    assert all(torch.equal(a, w) for a, w in zip(again, want[:depth]))
    with pytest.raises(Exception):
        pipe.process(imgs[:depth], 50)                                    # not the K the plan was compiled with
    other = cplan.CPlan(path)                                             # an independently loaded plan is NOT a clone: own constants
    h = ctypes.c_void_p()
    arr = (ctypes.c_void_p * 2)(plan._h.value, other._h.value)
    assert plan._L.cp_pipeline_create(arr, 2, ctypes.byref(h)) != 0 and b"not a clone" in plan._L.cp_last_error()
    other.close()
    # the clones keep the shared constants alive when the source plan goes first
    clone = plan.clone()
    first = clone.process(imgs[0], 100).clone()
    pipe.close()
    plan.close()
    second = clone.process(imgs[0], 100)
    torch.cuda.synchronize()
    assert torch.equal(first, want[0]) and torch.equal(second, want[0])
    clone.close()


def test_pybind_ext_plan_and_decode(tmp_path, golden_dir):
    """`_ext.plan_create / plan_forward / plan_process / plan_destroy` and `_ext.multi_pose_decode`."""
    import cases
    from centerpose_amd import _ext, engine, synth
    from centerpose_amd.decode import multi_pose_decode
    x = synth.make_images(2, 128, 128, seed=4).cuda()
    eng = engine.Engine("res_50", synth.make_state_dict("res_50"), 2, 128, 128, use_graph=False)
    ref = [t.clone() for t in eng(x)]
    ref_dets = multi_pose_decode(ref[0], ref[1], ref[2], reg=ref[3], hm_hp=ref[4], hp_offset=ref[5], K=100)
    path = str(tmp_path / "p.cpplan")
    eng.save_plan(path)
    h = _ext.plan_create(path, True)
    for _ in range(3):
        outs = _ext.plan_forward(h, x)
        dets = _ext.plan_process(h, x, 100)
    torch.cuda.synchronize()
    assert len(outs) == 6 and all(torch.equal(a, b) for a, b in zip(outs, ref)) and torch.equal(dets, ref_dets)
    _ext.plan_destroy(h)
    gen, kw, K, use_reg, use_off = cases.DECODE_CASES["rand_b2"]
    g = np.load(os.path.join(golden_dir, "decode_rand_b2.npz"))
    t = {k: torch.from_numpy(v).cuda() for k, v in gen(**kw).items()}
    d = _ext.multi_pose_decode(t["hm"], t["wh"], t["hps"], t["reg"], t["hm_hp"], t["hp_offset"], K)
    assert np.array_equal(d.cpu().numpy(), g["dets"])
    # optional index outputs: the same tensors the Python face returns, bit-equal to the reference's own indices
    d2, inds, hm_inds, scores = _ext.multi_pose_decode(t["hm"], t["wh"], t["hps"], t["reg"], t["hm_hp"], t["hp_offset"], K, return_indices=True)
    p2, pinds, phm, pscores = multi_pose_decode(t["hm"], t["wh"], t["hps"], reg=t["reg"], hm_hp=t["hm_hp"], hp_offset=t["hp_offset"], K=K,
                                                 return_indices=True)
    assert torch.equal(d2, d) and torch.equal(inds, pinds) and torch.equal(hm_inds, phm) and torch.equal(scores, pscores)
    assert np.array_equal(inds.cpu().numpy(), g["inds"]) and np.array_equal(hm_inds.cpu().numpy(), g["hm_inds"])


def test_pybind_ext_pipeline_from_state_dict():
    """`_ext.plan_create_from_state_dict(..., decode_k=K)` + `_ext.pipeline_create / pipeline_process / pipeline_destroy` (round 6): the
    reference-shaped pybind module reaches the steps-in-flight arrangement from a checkpoint in three calls, no plan file; per batch the
    detections equal `_ext.plan_process` of the same handle and the Python detector's `process`."""
    from centerpose_amd import _ext, config, detector, synth
    cfg = config.get_cfg("res_50", TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    B, H, W = 2, 128, 128
    imgs = [synth.make_images(B, H, W, seed=90 + i).cuda() for i in range(4)]
    want = [det.process(x)[1].clone() for x in imgs]
    h = _ext.plan_create_from_state_dict("res_50", det.model.state_dict(), B, H, W, head_conv=cfg.MODEL.HEAD_CONV, use_graph=True, decode_k=100)
    single = [_ext.plan_process(h, x, 100) for x in imgs]
    pipe = _ext.pipeline_create(h, 2)
    got = _ext.pipeline_process(pipe, imgs[0:2], 100) + _ext.pipeline_process(pipe, imgs[2:4], 100) + _ext.pipeline_process(pipe, imgs[0:2], 100)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(single, want))
    assert all(torch.equal(a, b) for a, b in zip(got, want + want[:2]))
    with pytest.raises(RuntimeError):
        _ext.pipeline_process(pipe, imgs[0:1], 100)                      # one batch for a depth-2 pipeline
    _ext.pipeline_destroy(pipe)
    h2 = _ext.plan_create_from_state_dict("res_50", det.model.state_dict(), B, H, W, head_conv=cfg.MODEL.HEAD_CONV)     # no decode inside
    with pytest.raises(RuntimeError, match="decode"):
        _ext.pipeline_create(h2, 2)
    _ext.plan_destroy(h2)
    _ext.plan_destroy(h)


@pytest.mark.parametrize("arch,head_conv", [("dla_34", 256), ("res_50", 64)])
def test_pybind_ext_plan_create_from_state_dict(arch, head_conv):
    """SURVEY 8b item 3 as written: plan_create(arch, state_dict tensors, B, H, W) -- `_ext.plan_create_from_state_dict` compiles the
    reference-format checkpoint (with a 'module.' prefix, as DataParallel checkpoints have: model.py:76-80) straight into a C plan
    handle, no plan file; heads / dets equal the Python engine's bits."""
    from centerpose_amd import _ext, engine, synth
    from centerpose_amd.decode import multi_pose_decode
    sd = synth.make_state_dict(arch, head_conv=head_conv)
    x = synth.make_images(2, 128, 160, seed=9).cuda()
    ref = [t.clone() for t in engine.Engine(arch, sd, 2, 128, 160, head_conv=head_conv, use_graph=False)(x)]
    ref_dets = multi_pose_decode(ref[0], ref[1], ref[2], reg=ref[3], hm_hp=ref[4], hp_offset=ref[5], K=100)
    h = _ext.plan_create_from_state_dict(arch, {"module." + k: v for k, v in sd.items()}, 2, 128, 160, head_conv, True)
    for _ in range(3):
        outs = _ext.plan_forward(h, x)
        dets = _ext.plan_process(h, x, 100)
    torch.cuda.synchronize()
    assert len(outs) == 6 and all(torch.equal(a, b) for a, b in zip(outs, ref)) and torch.equal(dets, ref_dets)
    _ext.plan_destroy(h)
    with pytest.raises(Exception):
        _ext.plan_create_from_state_dict(arch, {k: v for k, v in sd.items() if "hm" not in k}, 2, 128, 160, head_conv, True)


def test_c_plan_rejects_bad_files(tmp_path):
    from centerpose_amd import cplan
    from centerpose_amd._lib import CenterposeHipError
    bad = tmp_path / "bad.cpplan"
    bad.write_bytes(b"CPPLAN02" + b"\0" * 100)
    with pytest.raises(CenterposeHipError):
        cplan.CPlan(str(bad))
    with pytest.raises(CenterposeHipError):
        cplan.CPlan(str(tmp_path / "missing.cpplan"))


@pytest.mark.parametrize("arch", ARCHS)
def test_multistream_capture_is_bit_identical(arch):
    """Branch-parallel hipGraph capture (two streams, event edges from the recorded data dependencies) against the
    single-stream capture and the eager schedule: same bits; every launch respects its RAW / WAR / WAW predecessors."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict(arch)
    x = synth.make_images(2, 128, 128).cuda()
    outs = []
    for ns, graph in ((1, False), (1, True), (2, True)):
        e = engine.Engine(arch, sd, 2, 128, 128, use_graph=graph)
        e.nstreams = ns
        for _ in range(2):
            o = [t.clone() for t in e(x)]
        outs.append(o)
        if ns == 2:
            deps = e.dependencies()
            assert len(deps) == len(e.launches) and all(all(j < i for j in d) for i, d in enumerate(deps))
            assert len(set(e.stream_of_launch)) == 2
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert all(torch.equal(p, q) for p, q in zip(outs[0], o))


@pytest.mark.parametrize("arch", ["dla_34", "hrnet", "shufflenetV2"])
def test_critical_path_schedule(arch, monkeypatch, tmp_path):
    """Engine.schedule: the launch list re-ordered by critical path for two capture streams is a permutation of the emission
    order that respects every data dependency, uses both streams, replays to the same bits as the emission-order capture, and
    travels in the plan file (ops in execution order + stream per op) to `Engine.from_plan`."""
    from centerpose_amd import engine, plan, synth
    sd = synth.make_state_dict(arch)
    x = synth.make_images(2, 128, 128, seed=5).cuda()
    monkeypatch.setenv("CP_SCHED", "0")
    e0 = engine.Engine(arch, sd, 2, 128, 128, use_graph=True)
    for _ in range(2):
        ref = [t.clone() for t in e0(x)]
    assert getattr(e0, "stream_plan", None) is None
    names0 = [n for _, n, _, _ in e0.launches]
    monkeypatch.setenv("CP_SCHED", "1")
    e1 = engine.Engine(arch, sd, 2, 128, 128, use_graph=True)
    for _ in range(3):
        out = [t.clone() for t in e1(x)]
    names1 = [n for _, n, _, _ in e1.launches]
    assert sorted(names0) == sorted(names1) and len(e1.stream_plan) == len(names1) and set(e1.stream_plan) == {0, 1}
    assert e1.stream_of_launch == e1.stream_plan
    deps = e1.dependencies()
    assert all(all(j < i for j in d) for i, d in enumerate(deps))
    assert all(torch.equal(p, q) for p, q in zip(ref, out))
    assert all(torch.equal(p, q) for p, q in zip(ref, (e1.run_eager(), e1.outputs)[1]))      # the new order, eagerly
    path = str(tmp_path / "s.cpplan")
    e1.save_plan(path)
    p = plan.parse(memoryview(np.fromfile(path, dtype=np.uint8)))
    assert [o[5] for o in p["ops"]] == e1.stream_plan and [m["name"] for m in p["meta"]["ops"]] == names1
    e2 = engine.Engine.from_plan(path)
    assert e2.stream_plan == e1.stream_plan
    for _ in range(2):
        out2 = e2(x)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(ref, out2))


@pytest.mark.parametrize("arch,hw", [("dla_34", 128), ("res_50", 160)])
def test_decode_inside_the_schedule(arch, hw, tmp_path):
    """Engine(decode_k=K): multi_pose_decode as the last two launches (cp_decode_topk_f32 / cp_decode_assign_f32) of the schedule.
    One replay gives the six heads AND dets; dets are bit-equal to the stand-alone multi_pose_decode of the same heads (which
    the golden decode tests pin against the reference), eagerly, from the two-stream graph, from a re-loaded plan, and from the
    C plan handle (cp_plan_process hands the in-plan detections over instead of decoding again)."""
    from centerpose_amd import cplan, engine, synth
    from centerpose_amd.decode import multi_pose_decode
    sd = synth.make_state_dict(arch)
    x = synth.make_images(2, hw, hw, seed=9).cuda()
    plain = engine.Engine(arch, sd, 2, hw, hw, use_graph=False)
    heads = [t.clone() for t in plain(x)]
    want = multi_pose_decode(heads[0], heads[1], heads[2], reg=heads[3], hm_hp=heads[4], hp_offset=heads[5], K=100)
    with pytest.raises(Exception):
        plain.process(x)                                      # built without decode_k: says so
    for use_graph in (False, True):
        eng = engine.Engine(arch, sd, 2, hw, hw, use_graph=use_graph, decode_k=100)
        assert [l.fn for _, _, _, l in eng.launches[-2:]] == ["cp_decode_topk_f32", "cp_decode_assign_f32"] or use_graph
        for _ in range(3):
            outs, dets = eng.process(x)
        torch.cuda.synchronize()
        assert all(torch.equal(p, q) for p, q in zip(heads, outs)) and torch.equal(dets, want)
    fns = [l.fn for _, _, _, l in eng.launches]
    assert fns.index("cp_decode_topk_f32") < fns.index("cp_decode_assign_f32") == len(fns) - 1
    path = str(tmp_path / "d.cpplan")
    eng.save_plan(path)
    e2 = engine.Engine.from_plan(path)
    assert e2.decode_k == 100
    for _ in range(2):
        outs, dets = e2.process(x)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(heads, outs)) and torch.equal(dets, want)
    cp = cplan.CPlan(path, use_graph=1)
    for _ in range(3):
        d = cp.process(x, K=100)
    d50 = cp.process(x, K=50)                                   # another K: the plan's own decode does not apply, decoded separately
    torch.cuda.synchronize()
    assert torch.equal(d, want)
    assert torch.equal(d50, multi_pose_decode(heads[0], heads[1], heads[2], reg=heads[3], hm_hp=heads[4], hp_offset=heads[5], K=50))
    cp.close()


def test_detector_process_one_replay_equals_two_stage():
    """MultiPoseDetector.process without stage timing = forward + decode in one graph replay; with return_time (what run() asks
    for) the two-stage form: same outputs, same dets."""
    from centerpose_amd import config, detector, synth
    cfg = config.get_cfg("res_50", TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    x = synth.make_images(1, 128, 128, seed=3).cuda()
    o1, d1 = det.process(x)
    eng = det.model._engines.get((1, 128, 128, cfg.TEST.TOPK))
    assert eng is not None and eng.decode_k == cfg.TEST.TOPK                  # the one-replay path ran
    assert d1.data_ptr() != eng.dets.data_ptr() and torch.equal(d1, eng.dets)  # ... and handed out a private copy
    keep = d1
    o1, d1 = [t.clone() for t in o1], d1.clone()
    det.process(synth.make_images(1, 128, 128, seed=4).cuda())                # the next call must not overwrite what the caller holds
    torch.cuda.synchronize()
    assert torch.equal(keep, d1) and not torch.equal(keep, eng.dets)
    o2, d2, t = det.process(x, return_time=True)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(o1, o2)) and torch.equal(d1, d2) and t > 0
    flip = detector.MultiPoseDetector(config.get_cfg("res_50", TEST__FLIP_TEST=True))
    with pytest.raises(ValueError):                 # the mirrored twin is missing: refuse instead of reading past the batch
        flip.process(x)


@pytest.mark.parametrize("arch", ["res_50", "hrnet"])
def test_process_stream_equals_process_per_batch(arch, monkeypatch):
    """MultiPoseDetector.process_stream on the device (the host logic alone is tests/test_host_logic.py::test_process_many_*): seven
    batches of two shapes -- two full groups of the first shape, a group of MIXED shapes (FIX_RES = false: runs through `process`), a
    full group of the second shape, an odd last batch -- every yielded (outputs, dets) torch.equal to `process(batch)`, in order;
    `dets` are fresh tensors that survive later replays; the generator pulls at most `depth` batches ahead; the plan cache holds ONE
    pipeline per shape and stays inside CP_ENGINE_CACHE; depth 3 as well; FLIP_TEST on: batch by batch through `process`."""
    from centerpose_amd import config, detector, engine, synth
    monkeypatch.setenv("CP_ENGINE_CACHE", "4")
    cfg = config.get_cfg(arch, TEST__FLIP_TEST=False)
    det = detector.MultiPoseDetector(cfg)
    shapes = [(2, 128, 96)] * 4 + [(2, 128, 96), (2, 96, 128)] + [(2, 96, 128)] * 2 + [(2, 128, 96)]
    batches = [synth.make_images(B, H, W, seed=60 + i).cuda() for i, (B, H, W) in enumerate(shapes)]
    want = []
    for x in batches:
        outs, dets = det.process(x)
        want.append([t.clone() for t in outs] + [dets.clone()])
    torch.cuda.synchronize()
    pulled = []

    def source():
        for i, x in enumerate(batches):
            pulled.append(i)
            yield x
    got, kept = [], []
    for i, (outs, dets) in enumerate(det.process_stream(source(), depth=2)):
        assert len(pulled) <= 2 * (i // 2) + 2                                   # never more than one group ahead
        got.append([t.clone() for t in outs] + [dets.clone()])
        kept.append(dets)
    torch.cuda.synchronize()
    assert len(got) == len(batches)
    for i, (g, w) in enumerate(zip(got, want)):
        assert all(torch.equal(a, b) for a, b in zip(g, w)), "batch %d differs from process()" % i
    assert all(torch.equal(k, w[6]) for k, w in zip(kept, want))                # fresh tensors: untouched by the later replays
    pipes = {k: v for k, v in det.model._engines.items() if isinstance(v, engine.EnginePipeline)}
    assert sorted(k[:3] for k in pipes) == [(2, 96, 128), (2, 128, 96)] and len(det.model._engines) <= 4
    assert all(p.capture_mode == "2-stream" and p.engines[0] is det.model._engines.get(k[:4], p.engines[0]) for k, p in pipes.items())
    got3 = [dets.clone() for _, dets in det.process_stream(batches[:4], depth=3)]                 # one joint replay of three + a single
    torch.cuda.synchronize()
    assert all(torch.equal(a, w[6]) for a, w in zip(got3, want))
    flip = detector.MultiPoseDetector(config.get_cfg(arch, TEST__FLIP_TEST=True))
    pair = torch.cat([batches[0][:1], batches[0][:1].flip(3)], 0)
    ref_o, ref_d = flip.process(pair)
    ref_d = ref_d.clone()
    res = list(flip.process_stream([pair, pair], depth=2))
    torch.cuda.synchronize()
    assert len(res) == 2 and all(torch.equal(d, ref_d) and d.shape[0] == 1 for _, d in res)
    assert not any(isinstance(v, engine.EnginePipeline) for v in flip.model._engines.values())


def test_full_size_batch_invariance_and_scaling_property(monkeypatch):
    """Size-independent properties at BASELINE's full 512x512 size (the oracle is too slow there): (1) an image's head
    maps do not depend on which batch it travels in - B=4 in one engine == the same images through a B=2 engine, bit for
    bit (different grids, same per-output arithmetic); (2) one Winograd layer is exactly homogeneous under power-of-two
    scaling of its input (every step is a sum of products)."""
    from centerpose_amd import engine, ops, synth
    sd = synth.make_state_dict("dla_34")
    x = synth.make_images(4).cuda()
    # bit-for-bit needs the same arithmetic per output in both plans: the fused head launch (one 1x1 summation order) switches on
    # with the spatial tile count, i.e. at B = 4 but not at B = 2, and the split-K / split-C factors of the small-map launches
    # depend on the block count (round 3) -> compare like with like, then the default plan within 1e-5
    monkeypatch.setenv("CP_FUSE_HEADS", "0")
    monkeypatch.setenv("CP_DCN_SPLITK", "0")
    monkeypatch.setenv("CP_WINO_SPLITC", "0")
    monkeypatch.setenv("CP_WINO24_RULE", "32,16,1")       # round 4: F(2x4) vs F(2x2) is chosen by block count too -> no block floor
    monkeypatch.setenv("CP_CONV_SPLITK", "0")
    monkeypatch.setenv("CP_SPLIT_BF16_MINBLOCKS", "0")    # round 5: (only when the suite runs with the opt-in CP_SPLIT_BF16=1) f32 vs split kernel by block count
    e4 = engine.Engine("dla_34", sd, 4, 512, 512)
    full = [t.clone() for t in e4(x)]
    del e4
    e2 = engine.Engine("dla_34", sd, 2, 512, 512)
    for half in range(2):
        part = e2(x[2 * half:2 * half + 2])
        torch.cuda.synchronize()
        assert all(torch.equal(f[2 * half:2 * half + 2], p) for f, p in zip(full, part))
    del e2
    monkeypatch.setenv("CP_FUSE_HEADS", "1")
    monkeypatch.setenv("CP_DCN_SPLITK", "1")
    monkeypatch.setenv("CP_WINO_SPLITC", "1")
    monkeypatch.delenv("CP_WINO24_RULE")
    monkeypatch.delenv("CP_CONV_SPLITK")
    monkeypatch.delenv("CP_SPLIT_BF16_MINBLOCKS")
    ef = engine.Engine("dla_34", sd, 4, 512, 512)
    assert any(l.fn == "cp_splitk_reduce_f32" for _, _, _, l in ef.launches)
    assert sum(l.fn == "cp_head3x3_1x1_f32" for _, _, _, l in ef.launches) == 6          # all six branches (round 3: hps / hm_hp too)
    fused = ef(x)
    torch.cuda.synchronize()
    for a, b in zip(fused, full):
        # different summation orders (split partial sums, the fused heads' 1x1) through ~40 layers: ~2e-5 observed, bar 1e-3
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    del ef
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(16, 128, 128, 64, generator=g).cuda()
    w = (torch.randn(256, 64, 3, 3, generator=g) / 24.0).cuda()
    wp = ops.pack_conv_weight(w)
    u = ops.pack_wino_weight(wp, 64, 256)
    sc, sh = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    o1, o2 = torch.empty(16, 128, 128, 256, device="cuda"), torch.empty(16, 128, 128, 256, device="cuda")
    ops.conv2d([xin], wp, sc, sh, o1, kh=3, kw=3, stride=1, pad=1, cout=256, wino=u)
    ops.conv2d([xin * 4.0], wp, sc, sh, o2, kh=3, kw=3, stride=1, pad=1, cout=256, wino=u)
    torch.cuda.synchronize()
    assert torch.equal(o1 * 4.0, o2)


@pytest.mark.parametrize("arch,hw", [("dla_34", (512, 512)), ("hrnet", (256, 384))])
def test_batch_invariant_switch_pins_every_batch_dependent_rule(arch, hw, monkeypatch):
    """CP_BATCH_INVARIANT=1 (ops.BATCH_INVARIANT): ONE switch instead of the five the two tests around this one set -- every
    rule that looks at a launch's block count (F(2x4) vs F(2x2), the head kernels, split-K / split-C factors) is evaluated as if the
    batch were 1, so an image's head maps are the same BITS whether it travels in a batch of 6, 3 or 1; the default plan (rules at
    the real batch) stays within 1e-4 of the head scale of it."""
    from centerpose_amd import engine, ops, synth
    sd = synth.make_state_dict(arch)
    x = synth.make_images(6, hw[0], hw[1], seed=9).cuda()
    monkeypatch.setattr(ops, "BATCH_INVARIANT", True)
    e6 = engine.Engine(arch, sd, 6, hw[0], hw[1])
    full = [t.clone() for t in e6(x)]
    names6 = [(n, l.fn) for _, n, _, l in e6.emission]
    del e6
    for b in (3, 1):
        eb = engine.Engine(arch, sd, b, hw[0], hw[1])
        for i in range(0, 6, b):
            part = eb(x[i:i + b])
            torch.cuda.synchronize()
            assert all(torch.equal(f[i:i + b], p) for f, p in zip(full, part)), "B=%d, images %d.." % (b, i)
        assert [(n, l.fn) for _, n, _, l in eb.emission] == names6                     # the same launch list at every batch
        del eb
    monkeypatch.setattr(ops, "BATCH_INVARIANT", False)
    ed = engine.Engine(arch, sd, 6, hw[0], hw[1])
    dflt = ed(x)
    torch.cuda.synchronize()
    for a, b in zip(dflt, full):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("arch", ["res_50", "hrnet"])
def test_full_size_b8_properties(arch, monkeypatch):
    """BASELINE.json configs[1] (res_50 512x512 batch 8) and configs[4]'s per-GPU shape (hrnet 512x512 batch 8): the oracle is
    too slow at this size, so size-independent properties -- determinism across hipGraph replays, batch invariance (B=8 in
    one plan == the same images through a B=4 plan, bit for bit), and image 0 against the CPU oracle within the 1e-3 bar."""
    from centerpose_amd import engine, synth
    monkeypatch.setenv("CP_WINO_SPLITC", "0")       # the split-C factor of a small-map launch depends on the batch: bit-for-bit batch
    monkeypatch.setenv("CP_WINO24_RULE", "32,16,1") # (and so does the F(2x4) / F(2x2) choice: no block floor)
    monkeypatch.setenv("CP_CONV_SPLITK", "0")       # (and the split-K factor of the generic kernel's small-M launches)
    monkeypatch.setenv("CP_SPLIT_BF16_MINBLOCKS", "0")   # (and, when the suite runs with the opt-in CP_SPLIT_BF16=1, the f32-vs-split choice)
    sd = synth.make_state_dict(arch)                # invariance is a property of plans with the same per-output arithmetic
    x = synth.make_images(8, seed=123)
    e8 = engine.Engine(arch, sd, 8, 512, 512)
    a = [t.clone() for t in e8(x.cuda())]
    b = [t.clone() for t in e8(x.cuda())]
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    del e8
    e4 = engine.Engine(arch, sd, 4, 512, 512)
    for half in range(2):
        part = e4(x[4 * half:4 * half + 4].cuda())
        torch.cuda.synchronize()
        assert all(torch.equal(f[4 * half:4 * half + 4], p) for f, p in zip(a, part))
    refs = nets_torch.forward(arch, sd, x[:1])
    for n, o, r in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset"), a, refs):
        o, r = o[:1].cpu().double(), r.double()
        if n in ("hm", "hm_hp"):
            r = torch.sigmoid(r)            # the engine's default plan sigmoids hm / hm_hp in the head epilogue
        tol = 1e-3 if n in ("hm", "hm_hp", "reg", "hp_offset") else 1e-3 * r.abs().max().item()
        assert (o - r).abs().max().item() <= tol, n


def test_buffer_reuse_is_bit_identical_and_smaller(monkeypatch):
    """engine.BufferPool: reusing dead activations' storage changes neither eager, nor graph, nor two-stream results, and
    the plan allocates a fraction of one-buffer-per-edge."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict("dla_34")
    x = synth.make_images(2, 256, 256, seed=8).cuda()
    monkeypatch.setenv("CP_BUFFER_REUSE", "0")
    e0 = engine.Engine("dla_34", sd, 2, 256, 256, use_graph=False)
    ref = [t.clone() for t in e0(x)]
    big = e0.activation_bytes
    del e0
    monkeypatch.setenv("CP_BUFFER_REUSE", "1")
    for graph in (False, True):
        e1 = engine.Engine("dla_34", sd, 2, 256, 256, use_graph=graph)
        for _ in range(2):
            out = e1(x)
        torch.cuda.synchronize()
        assert all(torch.equal(p, q) for p, q in zip(ref, out))
        assert e1.activation_bytes < 0.8 * big


BENCH_LINE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "bench_line.json")


@pytest.mark.parametrize("arch,B", [("dla_34", 16), ("res_50", 8), ("hrnet", 8)])
def test_timed_configuration_parity(arch, B):
    """The configuration bench.py times, at its own batch: BASELINE.json configs[2] = dla_34 512x512 B=16 (also the
    per-GPU shape of configs[3]); configs[1] = res_50 B=8; configs[4]'s per-GPU shape = hrnet B=8.  The engine comes from
    `bench.make_engine` -- decode inside the schedule, critical-path two-stream hipGraph, fused head launches, the batch-dependent
    tile choices of this size.  Reference side: MultiPoseDetector.process, lib/detectors/multi_pose.py:29-60.
    (a) images 0, B/2-1, B-1 against the CPU oracle network: the north-star 1e-3 bar on every head;
    (b) dets of ALL images bit-equal to the oracle decode of the engine's own head maps;
    (c) the kernel instantiations this engine dispatched to equal -- hard assertion -- the `roofline.kernels` of the bench line that
        tests/test_dist_gpu.py::test_bench_single_gpu_line_contract produced earlier in the SAME pytest session (when that file is
        there and fresh); they are also WRITTEN to gpurun_out/tested_kernels_<arch>.json;
        tools/gpu_check.sh compares that list with `roofline.kernels` of the bench line of the same box session (the tested kernels
        ARE the timed kernels).  Round 3 asserted against the committed profiles/bench_line.json here, which made the driver's test
        run depend on a hand-refreshed file: a stale file now only prints a note;
    (d) a second replay reproduces every bit."""
    import json
    import sys
    from centerpose_amd import synth
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    eng = bench.make_engine(arch, B)
    x = synth.make_images(B, seed=317)
    outs, dets = eng.process(x.cuda())
    torch.cuda.synchronize()
    assert eng.capture_mode == "2-stream", eng.capture_mode
    first = [t.clone() for t in outs] + [dets.clone()]
    outs, dets = eng.process(x.cuda())                                                    # (d)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(first, list(outs) + [dets]))
    o = [t.cpu().numpy() for t in outs]
    want = decode_np.multi_pose_decode(o[0], o[1], o[2], o[3], o[4], o[5], K=100)          # (b)
    assert dets.shape == (B, 100, 56) and np.array_equal(dets.cpu().numpy(), want)
    sd = synth.make_state_dict(arch)
    worst = {}
    for i in sorted({0, B // 2 - 1, B - 1}):                                              # (a)
        refs = nets_torch.forward(arch, sd, x[i:i + 1])
        for n, got, r in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset"), outs, refs):
            got, r = got[i:i + 1].cpu().double(), r.double()
            if n in ("hm", "hm_hp"):
                r = torch.sigmoid(r)                       # the plan sigmoids hm / hm_hp in the head epilogue (multi_pose.py:35-37)
            tol = 1e-3 if n in ("hm", "hm_hp", "reg", "hp_offset") else 1e-3 * r.abs().max().item()
            err = (got - r).abs().max().item()
            worst[n] = max(worst.get(n, 0.0), err / tol)
            assert err <= tol, "image %d %s: max err %.3e > %.3e" % (i, n, err, tol)
    print("%s B=%d: worst error / tolerance per head: %s" % (arch, B, {k: round(v, 4) for k, v in worst.items()}))
    kernels = sorted({l.kernel or l.fn for _, _, _, l in eng.launches})
    print("kernel instantiations:", kernels)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))                      # (c)
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "tested_kernels_%s.json" % arch), "w") as f:
            json.dump({"arch": arch, "batch": B, "kernels": kernels}, f, indent=1)
    except OSError as e:
        print("could not write the tested-kernel list:", e)
    # same-session hard check: tests/test_dist_gpu.py::test_bench_single_gpu_line_contract ran bench.py a few minutes ago
    # in this pytest session and left the kernel list of THAT line; the kernels parity-tested here must be the kernels it timed
    import time
    tk = os.path.join(root, "gpurun_out", "timed_kernels_dla_34.json")
    if arch == "dla_34" and os.path.exists(tk) and time.time() - os.path.getmtime(tk) < 3600:
        timed = json.load(open(tk))["kernels"]
        assert kernels == timed, "bench.py timed other kernels than the ones parity-tested here: %s" % sorted(set(kernels) ^ set(timed))
        print("tested kernels == timed kernels (bench line of this session): %d instantiations" % len(kernels))
    if arch == "dla_34" and os.path.exists(BENCH_LINE):
        timed = sorted(json.load(open(BENCH_LINE))["roofline"]["kernels"])
        if kernels != timed:
            print("note: profiles/bench_line.json lists other kernels than this engine dispatched to (stale file?):",
                  sorted(set(kernels) ^ set(timed)))


def test_timed_configuration_parity_two_steps_in_flight():
    """THE arrangement bench.py's headline times, at its own size and through the PRODUCT entry point:
    `bench.make_detector("dla_34")` -> `MultiPoseDetector.process_stream(batches, depth=2)` at B = 16, 512 x 512 -- two plan
    instances (2 x 1.84 GB of activations), ~190 launches scheduled jointly on the two capture streams, ONE hipGraph per two steps.
    Replaces lib/detectors/multi_pose.py:29-60 over a stream of batches.
    (a) every step's six maps and `dets` are torch.equal to `MultiPoseDetector.process` of the same batch (single plan, one step per
        replay) -- four steps = two joint replays on two DIFFERENT batches, so the second replay reproduces the first;
    (b) the second instance (the one no single-plan test ever ran): images 0 / 7 / 15 within the 1e-3 bar of the CPU oracle network,
        `dets` of all 16 images == the oracle decode of the instance's own maps;
    (c) the joint graph was captured on two streams and runs exactly the single plan's kernel instantiations, twice;
    (d) `dets` are fresh tensors (not the plans' static buffers) and survive the next replay."""
    import sys
    from centerpose_amd import synth
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    B, D = 16, 2
    det = bench.make_detector("dla_34")
    host = bench.make_batches(B, D)
    assert not torch.equal(host[0], host[1])
    batches = [x.cuda() for x in host]
    want = []
    for x in batches:                                                                  # one step per replay: MultiPoseDetector.process
        outs, dets = det.process(x)
        want.append([t.clone() for t in outs] + [dets.clone()])
    torch.cuda.synchronize()
    got, kept = [], []
    for outs, dets in det.process_stream(bench.feed(batches, 2 * D), depth=D):
        got.append([t.clone() for t in outs] + [dets.clone()])
        kept.append(dets)
    torch.cuda.synchronize()
    assert len(got) == 2 * D
    for i, g in enumerate(got):                                                        # (a)
        for n, a, b in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset", "dets"), g, want[i % D]):
            assert torch.equal(a, b), "step %d %s differs from the one-step-per-replay result" % (i, n)
    eng = bench.make_engine("dla_34", B, det=det)
    pipe = det.model.pipeline_for(B, 512, 512, det.cfg.TEST.TOPK, D)
    assert pipe.engines[0] is eng and pipe.capture_mode == "2-stream" and eng.capture_mode == "2-stream"      # (c)
    statics = {e.dets.data_ptr() for e in pipe.engines}
    assert len({k.data_ptr() for k in kept}) == 2 * D and not ({k.data_ptr() for k in kept} & statics)         # (d)
    assert all(torch.equal(k, w[6]) for k, w in zip(kept, want + want))
    kern = lambda launches: sorted(l.kernel or l.fn for _, _, _, l in launches)
    assert kern(pipe.joint.launches) == sorted(kern(eng.launches) * D)
    o = [t.cpu().numpy() for t in got[1][:6]]                                          # (b): step 1 ran on instance 1
    dec = decode_np.multi_pose_decode(o[0], o[1], o[2], o[3], o[4], o[5], K=100)
    assert np.array_equal(got[1][6].cpu().numpy(), dec)
    sd = synth.make_state_dict("dla_34")
    worst = {}
    for i in (0, B // 2 - 1, B - 1):
        refs = nets_torch.forward("dla_34", sd, host[1][i:i + 1])
        for n, t, r in zip(("hm", "wh", "hps", "reg", "hm_hp", "hp_offset"), got[1], refs):
            t, r = t[i:i + 1].cpu().double(), r.double()
            if n in ("hm", "hm_hp"):
                r = torch.sigmoid(r)
            tol = 1e-3 if n in ("hm", "hm_hp", "reg", "hp_offset") else 1e-3 * r.abs().max().item()
            err = (t - r).abs().max().item()
            worst[n] = max(worst.get(n, 0.0), err / tol)
            assert err <= tol, "instance 1 image %d %s: max err %.3e > %.3e" % (i, n, err, tol)
    print("dla_34 B=16 two steps in flight, instance 1: worst error / tolerance per head: %s" % {k: round(v, 4) for k, v in worst.items()})


@pytest.mark.parametrize("arch,depth", [("dla_34", 2), ("hrnet", 3)])
def test_steps_in_flight_same_bits_as_one_after_the_other(arch, depth):
    """engine.EnginePipeline (what bench.py times since round 5: `depth` plan instances whose launch lists are scheduled TOGETHER and
    captured into one hipGraph, so that one step's kernels fill the launch gaps and chain tails of the other): every step's heads
    and detections are BIT-IDENTICAL to the same images through a single engine, one step per replay -- inputs changing every
    replay, the graph replayed several times."""
    from centerpose_amd import engine, synth
    sd = synth.make_state_dict(arch, seed=4)
    B, H, W = 2, 128, 96
    imgs = [synth.make_images(B, H, W, seed=20 + i).cuda() for i in range(2 * depth)]
    one = engine.Engine(arch, sd, B, H, W, use_graph=True, decode_k=100)
    want = []
    for x in imgs:
        outs, dets = one.process(x)
        want.append([t.clone() for t in outs] + [dets.clone()])
    torch.cuda.synchronize()
    pipe = engine.EnginePipeline(arch, sd, B, H, W, depth=depth, use_graph=True, decode_k=100)
    assert len({e.input.data_ptr() for e in pipe.engines}) == depth and len(pipe.joint.launches) == depth * len(one.launches)
    got = []
    for r in range(2):
        res = pipe.process_all(imgs[r * depth:(r + 1) * depth])
        torch.cuda.synchronize()
        got += [[t.clone() for t in outs] + [dets.clone()] for outs, dets in res]
    assert pipe.capture_mode == "2-stream", pipe.capture_mode
    for i, (g, w) in enumerate(zip(got, want)):
        assert all(torch.equal(a, b) for a, b in zip(g, w)), "step %d differs" % i
    # the joint schedule really interleaves the instances: both capture streams carry launches of more than one instance
    owner = {id(l): k for k, e in enumerate(pipe.engines) for _, _, _, l in e.launches}
    per_stream = [{owner[id(l)] for (_, _, _, l), st in zip(pipe.joint.launches, pipe.joint.stream_of_launch) if st == q} for q in (0, 1)]
    assert all(len(x) >= 2 for x in per_stream), per_stream


def test_model_shares_constants_and_schedules_between_shapes():
    """FIX_RES = false compiles a plan per image size (model.BackBoneWithHead.engine_for).  The plans of one model share the
    uploaded / Winograd-transformed weights and, for an identical launch list, the measured two-stream schedule --
    and compute the same bits as a stand-alone engine of that shape; loading new weights drops the shared constants."""
    from centerpose_amd import config, engine, model, synth
    cfg = config.get_cfg("dla_34", TEST__FLIP_TEST=False)
    m = model.create_model(cfg.MODEL.NAME, cfg.MODEL.HEAD_CONV, cfg).to("cuda")
    x1, x2 = synth.make_images(1, 128, 160, seed=1).cuda(), synth.make_images(1, 160, 128, seed=2).cuda()
    o1 = [t.clone() for t in m(x1)]
    n_const, n_sched = len(m._const_cache), len(m._sched_cache)
    assert n_const > 100 and n_sched == 1
    e1 = m._engines[(1, 128, 160)]
    o2 = [t.clone() for t in m(x2)]
    e2 = m._engines[(1, 160, 128)]
    assert len(m._const_cache) == n_const                     # nothing was uploaded or transformed again
    # a schedule is shared only between plans with the same launch list AND the same dependency DAG (buffer reuse depends on sizes)
    same = [n for _, n, _, _ in e1.emission] == [n for _, n, _, _ in e2.emission] and \
        e1.dependencies(e1.emission) == e2.dependencies(e2.emission)
    assert len(m._sched_cache) == (n_sched if same else n_sched + 1)
    if same:
        assert e2.stream_plan == e1.stream_plan and [n for _, n, _, _ in e1.launches] == [n for _, n, _, _ in e2.launches]
    m._engines.pop((1, 128, 160))                               # evicted plan, same shape again: schedule comes from the cache
    o1b = m(x1)
    assert len(m._sched_cache) == (n_sched if same else n_sched + 1) and len(m._const_cache) == n_const
    assert all(torch.equal(a, b) for a, b in zip(o1, o1b))
    u1 = {id(t) for _, _, _, l in e1.launches for t in l.tensors if t is not None}
    assert sum(id(t) in u1 for _, _, _, l in e2.launches for t in l.tensors if t is not None) > 20      # the same Winograd-domain weight tensors
    ref = engine.Engine("dla_34", m.state_dict(), 1, 160, 128, head_conv=cfg.MODEL.HEAD_CONV, sigmoid_heads=("hm", "hm_hp"))(x2)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(o2, ref))
    assert all(torch.equal(a, b) for a, b in zip(o1, m(x1)))
    m.load_state_dict(synth.make_state_dict("dla_34", seed=5))
    assert not m._const_cache and not m._engines
    o3 = m(x1)
    torch.cuda.synchronize()
    assert not torch.equal(o3[0], o1[0])


def test_schedule_cache_never_applies_an_order_of_another_dag():
    """plans of one model with the SAME launch names can have DIFFERENT dependency DAGs -- BufferPool reuse adds
    WAR / WAW edges that depend on buffer sizes, and the split-K / split-C workspaces change size with batch and map size.  A cached
    (order, streams) of one DAG applied to the other ran `conv_offset_mask` before the launches whose workspace it aliases.  The cache
    key now carries the DAG and a hit is validated as a topological order of the emission-order DAG.  Shapes from the advisor's
    simulation: (1, 512, 512) then (8, 64, 64) and (8, 128, 96); every plan == a stand-alone engine of that shape, bit for bit."""
    from centerpose_amd import config, engine, model, synth
    cfg = config.get_cfg("dla_34", TEST__FLIP_TEST=False)
    m = model.create_model(cfg.MODEL.NAME, cfg.MODEL.HEAD_CONV, cfg).to("cuda")
    seen = []
    for (B, H, W) in [(1, 512, 512), (8, 64, 64), (8, 128, 96), (8, 64, 64)]:
        x = synth.make_images(B, H, W, seed=H + W).cuda()
        got = [t.clone() for t in m(x)]
        eng = m._engines[(B, H, W)]
        deps0 = eng.dependencies(eng.emission)
        pos = {id(l): i for i, (_, _, _, l) in enumerate(eng.launches)}
        assert engine.order_is_topological(deps0, sorted(range(len(eng.emission)), key=lambda i: pos[id(eng.emission[i][3])]))
        seen.append(([n for _, n, _, _ in eng.emission], deps0))
        ref = engine.Engine("dla_34", m.state_dict(), B, H, W, head_conv=cfg.MODEL.HEAD_CONV, sigmoid_heads=("hm", "hm_hp"))(x)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), (B, H, W)
    distinct = []
    for names, deps in seen:
        if not any(names == n and deps == d for n, d in distinct):
            distinct.append((names, deps))
    assert len(m._sched_cache) == len(distinct)
    print("shapes with equal names but different DAGs:",
          sum(1 for i, (n, d) in enumerate(seen) for (n2, d2) in seen[:i] if n == n2 and d != d2))
