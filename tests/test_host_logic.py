"""CPU: host-side logic of the drop-in -- config, checkpoint spec, synthetic weights, post-process
affine, soft-NMS (host C++ through the C ABI) -- and oracle cross-checks that need no GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_spec_counts_and_flops():
    from centerpose_amd import nets
    for arch, nkeys, gf in (("dla_34", 410, 80.34), ("res_50", 360, 86.85), ("hrnet", 1776, 85.27), ("mobilenetv3", 471, 15.59),
                            ("shufflenetV2", 399, 136.27), ("resdcn_18", 189, 30.19), ("resdcn_50", 387, 55.95)):
        spec, flops = nets.param_spec(arch)
        assert len(spec) == nkeys                      # 410 / 360 entries (probe of the reference)
        assert abs(flops / 1e9 - gf) < 0.05


def test_synth_is_deterministic_and_complete():
    from centerpose_amd import nets, synth
    a, b = synth.make_state_dict("res_50"), synth.make_state_dict("res_50")
    spec, _ = nets.param_spec("res_50")
    assert set(a) == set(spec)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert all(tuple(a[k].shape) == tuple(spec[k]) for k in a)


@pytest.mark.reference
@pytest.mark.parametrize("arch", ["dla_34", "res_50", "hrnet", "mobilenetv3", "shufflenetV2", "resdcn_18", "resdcn_34", "resdcn_50", "resdcn_101"])
def test_spec_and_oracle_match_imported_reference(arch):
    """key names/shapes == the reference module's state_dict; torch oracle == reference forward."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_nets
    from centerpose_amd import nets, synth
    from oracle import nets_torch
    m = make_golden_nets.build_reference(arch)
    ref_sd = m.state_dict()
    spec, _ = nets.param_spec(arch)
    assert set(spec) == set(ref_sd)
    assert all(tuple(ref_sd[k].shape) == tuple(spec[k]) for k in spec)
    sd = synth.make_state_dict(arch, seed=5)
    m.load_state_dict(sd, strict=True)
    x = synth.make_images(1, 64, 64, seed=1)
    with torch.no_grad():
        ref = m(x)
    out = nets_torch.forward(arch, sd, x)
    for r, o in zip(ref, out):
        assert torch.equal(r, o)


def test_oracle_matches_reference_golden_nets(golden_dir):
    from centerpose_amd import synth
    from oracle import nets_torch
    for arch in ("dla_34", "res_50", "hrnet", "mobilenetv3", "shufflenetV2", "resdcn_18", "resdcn_50"):
        g = np.load(os.path.join(golden_dir, "net_%s_128.npz" % arch))
        out = nets_torch.forward(arch, synth.make_state_dict(arch), synth.make_images(1, 128, 128, seed=7))
        for i, o in enumerate(out):
            assert np.allclose(o.numpy(), g["out%d" % i], rtol=1e-4, atol=1e-4 * np.abs(g["out%d" % i]).max())


def test_config_presets_and_yaml(tmp_path):
    from centerpose_amd import config
    cfg = config.get_cfg("dla_34")
    assert cfg.MODEL.HEAD_CONV == 256 and cfg.TEST.FLIP_TEST is True and cfg.TEST.TOPK == 100
    y = tmp_path / "e.yaml"
    y.write_text("MODEL:\n  NAME: 'res_50'\n  HEAD_CONV: 64\nTEST:\n  TOPK: 50\n")
    config.update_config(cfg, str(y))
    assert cfg.MODEL.NAME == "res_50" and cfg.TEST.TOPK == 50 and cfg.MODEL.INPUT_H == 512


def test_post_process_affine_hand_computed():
    """FIX_RES 512 input, 128 map, image 640x480: c=(320,240), s=640 -> x_img = x_map*5, y_img = y_map*5 - 80."""
    from centerpose_amd.post_process import multi_pose_post_process, transform_preds
    pts = np.array([[0, 0], [64, 64], [128, 128], [10, 100]], np.float32)
    out = transform_preds(pts, np.array([320., 240.], np.float32), 640.0, (128, 128))
    assert np.allclose(out, pts * 5 + np.array([0, -80]), atol=1e-4)
    dets = np.zeros((1, 2, 56), np.float32)
    dets[0, 0, :4] = [10, 20, 30, 40]; dets[0, 0, 4] = 0.9; dets[0, 0, 5:7] = [64, 64]; dets[0, 0, 39:] = 0.5
    ret = multi_pose_post_process(dets, [np.array([320., 240.], np.float32)], [640.0], 128, 128)
    row = np.array(ret[0][1][0])
    assert np.allclose(row[:4], [50, 20, 150, 120], atol=1e-3) and abs(row[4] - 0.9) < 1e-6
    assert np.allclose(row[5:7], [320, 240], atol=1e-3) and np.allclose(row[39:], 0.5)


def _nms_boxes():
    b = np.zeros((4, 56), np.float32)
    b[0, :5] = [0, 0, 9, 9, 0.5]      # lower score, listed first
    b[1, :5] = [0, 0, 9, 9, 0.9]      # identical box, best score
    b[2, :5] = [100, 100, 109, 109, 0.8]
    b[3, :5] = [5, 0, 14, 9, 0.0011]  # half overlap, will fall under the threshold
    for i in range(4):
        b[i, 5:39] = i + 1
        b[i, 39:] = 10 * (i + 1)
    return b


def test_soft_nms_39_hand_worked():
    """nms.pyx:172-275 quirks: swap of cols 0..38 only, gaussian decay exp(-ov^2/sigma), copy+swap on discard."""
    import __graft_entry__ as g
    g.build()
    from centerpose_amd.detector import soft_nms_39
    from oracle import dcn as odcn
    b = _nms_boxes()
    keep = soft_nms_39(b, Nt=0.5, method=2)
    # row 0 now holds the 0.9 box (cols 0..38) but keeps ITS OWN keypoint scores (cols 39..)
    assert b[0, 4] == np.float32(0.9) and np.all(b[0, 5:39] == 2) and np.all(b[0, 39:] == 10)
    # the identical 0.5 box: ov = 1 -> weight exp(-1/0.5) = e^-2
    i05 = int(np.argmin(np.abs(b[:, 4] - 0.5 * np.exp(-2.0))))
    assert abs(b[i05, 4] - np.float32(0.5) * np.float32(np.exp(-2.0))) < 1e-7
    # disjoint box untouched
    assert np.any(np.isclose(b[:, 4], 0.8))
    assert len(keep) == 3            # the 0.0011 box decayed under 0.001 and was dropped from `keep`
    b2 = _nms_boxes()
    keep2 = odcn.soft_nms_39(b2, Nt=0.5, method=2)
    assert np.array_equal(b, b2) and keep == keep2


def test_soft_nms_39_random_vs_oracle():
    from centerpose_amd.detector import soft_nms_39
    from oracle import dcn as odcn
    r = np.random.RandomState(3)
    for method in (0, 1, 2):
        b = r.rand(60, 56).astype(np.float32)
        b[:, 2:4] = b[:, 0:2] + r.rand(60, 2) * 0.8 + 0.05
        b[:, :4] *= 40
        b2 = b.copy()
        k1 = soft_nms_39(b, sigma=0.5, Nt=0.5, threshold=0.05, method=method)
        k2 = odcn.soft_nms_39(b2, sigma=0.5, Nt=0.5, threshold=0.05, method=method)
        assert np.array_equal(b, b2) and k1 == k2


def test_flip_helpers_roundtrip():
    """flip_lr / flip_lr_off are involutions (W-flip + L/R joint swap + x negation)."""
    from oracle import decode_np
    r = np.random.RandomState(0)
    hp = r.randn(1, 17, 8, 8).astype(np.float32)
    hps = r.randn(1, 34, 8, 8).astype(np.float32)
    assert np.array_equal(decode_np.flip_lr(decode_np.flip_lr(hp)), hp)
    assert np.array_equal(decode_np.flip_lr_off(decode_np.flip_lr_off(hps)), hps)


def test_plan_format_serialize_parse_and_schema():
    """Binary plan format on CPU tensors (no pickle): buffer views (offsets), constants, descriptor bytes and integer
    arguments survive serialize -> parse; corrupt files are rejected."""
    import struct
    from centerpose_amd import ops, plan
    inp = torch.zeros(2, 3, 8, 8)
    store = torch.zeros(2 * 8 * 8 * 16 + 64)
    act = store[64:].view(2, 8, 8, 16)                         # view with an offset into a written storage
    w, sc, sh = torch.randn(16, 48), torch.ones(16), torch.zeros(16)
    out = torch.zeros(2, 1, 8, 8)
    l1 = ops.conv2d_launch([inp], w, sc, sh, act, kh=4, kw=4, stride=1, pad=0, cout=16, in_nchw=True, Ho=8, Wo=8)
    l2 = ops.maxpool2d_launch(act, act, 1, 1, 0)
    wp = torch.randn(16, 16)
    l3 = ops.conv2d_launch([act], wp, sc, sh, out, kh=1, kw=1, cout=1, out_nchw=True, act=ops.ACT_SIGMOID)
    blob = plan.serialize([("conv", "a", 10, l1), ("pool", "p", 0, l2), ("conv", "h", 5, l3)],
                          {"arch": "x", "flops_per_image": 3}, inp, [out], 2)
    p = plan.parse(memoryview(blob))
    assert (p["abi"], p["B"], p["H"], p["W"]) == (2, 2, 8, 8) and p["meta"]["arch"] == "x"
    assert p["buffers"] == [inp.numel(), store.numel(), out.numel()]
    assert [o[0] for o in p["ops"]] == ["cp_conv2d_f32", "cp_maxpool2d_nhwc_f32", "cp_conv2d_f32"]
    assert p["ops"][1][3] == l2.ints and p["ops"][0][1] == bytes(l1.desc)
    assert p["ops"][0][2][-1] == (plan.REF_BUF, 1, 64, act.numel())          # out of l1: buffer 1, offset 64 floats
    assert p["ops"][0][2][1] == (plan.REF_NULL, 0, 0, 0)                      # unused source slot
    kinds = [r[0] for r in p["ops"][2][2]]
    assert kinds[4] == plan.REF_CONST and p["outputs"] == [((plan.REF_BUF, 2, 0, out.numel()), (2, 1, 8, 8))]
    # constants come back bit-identical, scale / shift are stored once (shared by both convs)
    n, off = p["consts"][p["ops"][2][2][4][1]]
    assert np.array_equal(np.frombuffer(blob, np.float32, n, off), wp.numpy().reshape(-1))
    assert len(p["consts"]) == 4
    d = ops.ConvDesc.from_buffer_copy(p["ops"][2][1])
    assert (d.kh, d.Cout, d.outNCHW, d.act) == (1, 1, 1, ops.ACT_SIGMOID)
    # execution streams: default all on the main stream; a schedule is stored per op; format 02 files (no schedule) still parse
    assert blob[:8] == b"CPPLAN04" and [o[5] for o in p["ops"]] == [0, 0, 0]
    # CPPLAN04 carries an FNV-1a checksum of everything behind the 48-byte header: one flipped constant byte is caught ...
    flipped = bytearray(blob)
    flipped[-3] ^= 0x10
    with pytest.raises(ValueError, match="checksum"):
        plan.parse(memoryview(bytes(flipped)))
    assert plan.checksum(b"") == 2166136261 and plan.checksum(b"a") == 0xE40C292C          # FNV-1a test vectors
    # ... older files (no checksum) still parse; the structural checks below run on such a file so that they, not the checksum, fire
    blob = b"CPPLAN03" + blob[8:]
    blob2 = plan.serialize([("conv", "a", 10, l1), ("pool", "p", 0, l2), ("conv", "h", 5, l3)],
                           {"arch": "x", "flops_per_image": 3}, inp, [out], 2, streams=[0, 1, 0])
    assert [o[5] for o in plan.parse(memoryview(blob2))["ops"]] == [0, 1, 0] and len(blob2) == len(blob)
    blob2 = b"CPPLAN03" + blob2[8:]
    assert [o[:5] for o in plan.parse(memoryview(b"CPPLAN02" + blob[8:]))["ops"]] == [o[:5] for o in p["ops"]]
    with pytest.raises(ValueError):                             # a third stream does not exist
        bad = bytearray(blob2)
        pos = blob2.index(struct.pack("<IIIIII", ops.FN_IDS["cp_maxpool2d_nhwc_f32"], 0, 2, 9, l2.out_index, 1))
        bad[pos + 20:pos + 24] = struct.pack("<I", 2)
        plan.parse(memoryview(bytes(bad)))
    with pytest.raises(ValueError):
        plan.parse(memoryview(b"NOTAPLAN" + blob[8:]))
    with pytest.raises(ValueError):                             # unknown launch function id
        bad = bytearray(blob)
        pos = blob.index(struct.pack("<IIII", ops.FN_IDS["cp_maxpool2d_nhwc_f32"], 0, 2, 9))
        bad[pos:pos + 4] = struct.pack("<I", 99)
        plan.parse(memoryview(bytes(bad)))
    with pytest.raises((ValueError, struct.error)):             # truncated
        plan.parse(memoryview(blob[:200]))


def test_list_schedule_properties():
    """engine.list_schedule (the critical-path schedule of the launch DAG for the capture streams) on seeded random DAGs: the
    order is a permutation that keeps every dependency, streams are in range, the simulated makespan lies between the critical
    path and the serial sum, one stream = serial, a chain stays on one stream, independent equal tasks split evenly."""
    import random
    from centerpose_amd.engine import list_schedule
    rnd = random.Random(7)
    for trial in range(30):
        n = rnd.randint(1, 60)
        deps = [sorted(rnd.sample(range(i), min(i, rnd.randint(0, 3)))) for i in range(n)]
        dur = [rnd.choice([0.0, 0.01, 0.05, 0.2, 1.0]) for _ in range(n)]
        for ns in (1, 2, 3):
            order, assign, mk = list_schedule(deps, dur, ns)
            assert sorted(order) == list(range(n)) and all(0 <= a < ns for a in assign)
            pos = {i: k for k, i in enumerate(order)}
            assert all(pos[j] < pos[i] for i in range(n) for j in deps[i])
            crit = [0.0] * n
            for i in range(n):
                crit[i] = dur[i] + max([crit[j] for j in deps[i]], default=0.0)
            assert max(crit) - 1e-9 <= mk <= sum(dur) + 1e-9
            if ns == 1:
                assert abs(mk - sum(dur)) < 1e-9
    order, assign, mk = list_schedule([[]] + [[i] for i in range(9)], [1.0] * 10, 2)        # a chain
    assert order == list(range(10)) and len(set(assign)) == 1 and mk == 10.0
    order, assign, mk = list_schedule([[] for _ in range(8)], [1.0] * 8, 2)                   # independent, equal
    assert sorted(assign) == [0] * 4 + [1] * 4 and mk == 4.0
    # the longer branch goes first: 0 -> {1 (short), 2 (long)} -> 3
    order, assign, mk = list_schedule([[], [0], [0], [1, 2]], [1.0, 2.0, 3.0, 1.0], 2)
    assert mk == 5.0 and assign[1] != assign[2]


def test_reconcile_state_dict_and_load_model(tmp_path, capsys):
    """load_model's checkpoint contract (lib/models/model.py:67-101): 'module.' prefix stripped, a shape mismatch keeps the
    model's own tensor, a key the model lacks is dropped, a key the checkpoint lacks is filled from the model -- each with the
    reference's log line."""
    from centerpose_amd import model as cpm

    class Holder:
        def __init__(self, sd):
            self.sd, self.loaded = sd, None

        def state_dict(self):
            return self.sd

        def load_state_dict(self, sd, strict=False):
            self.loaded = sd

    own = {"a.weight": torch.zeros(2, 3), "b.bias": torch.zeros(4), "c.weight": torch.zeros(5)}
    ckpt = {"module.a.weight": torch.ones(2, 3), "module.b.bias": torch.ones(7), "module.extra": torch.ones(1)}
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 12, "state_dict": ckpt}, path)
    h = cpm.load_model(Holder(own), path)
    assert set(h.loaded) == set(own)
    assert torch.equal(h.loaded["a.weight"], torch.ones(2, 3))        # taken from the checkpoint
    assert h.loaded["b.bias"] is own["b.bias"]                        # shape mismatch: the model's tensor stays
    assert h.loaded["c.weight"] is own["c.weight"]                    # absent from the checkpoint
    out = capsys.readouterr().out
    assert "epoch 12" in out and "Skip loading parameter b.bias" in out and "Drop parameter extra." in out and "No param c.weight." in out
    with pytest.raises(NotImplementedError):
        cpm.load_model(Holder(own), path, optimizer=object())


def _check_nms_against_golden(fn, golden_dir):
    sys.path.insert(0, golden_dir)
    import make_golden_nms
    gold = np.load(os.path.join(golden_dir, "soft_nms_39.npz"))
    for name, (boxes, kw) in make_golden_nms.cases().items():
        work = boxes.copy()
        keep = fn(work, **kw)
        exp, exp_keep = gold[name + "__out"], gold[name + "__keep"].tolist()
        assert keep == exp_keep, name
        cols = [c for c in range(56) if c != 4]
        assert np.array_equal(work[:, cols], exp[:, cols]), name            # every move / swap / discard of the reference, bit for bit
        if kw.get("method", 0) == 2:       # Gaussian weight: exp() evaluated in another precision than the generator's: 1 ulp per
            assert np.allclose(work[:, 4], exp[:, 4], rtol=5e-6, atol=0.0), name      # decay, a score takes one per overlapping better box
        else:
            assert np.array_equal(work[:, 4], exp[:, 4]), name


def test_soft_nms_39_matches_reference_source_golden(golden_dir):
    """soft_nms_39 pinned against vectors produced by the reference's own source (lib/external/nms.pyx:172-275
    executed as Python by tests/golden/make_golden_nms.py -- the Cython file does not compile here): the product's host C++
    (csrc/host_nms.cpp) and the oracle's C restatement (oracle/nms_ref.c) reproduce every box move, the 0:39-only swap, the discards
    and `keep` bit for bit for the hard / linear methods, and the Gaussian-decayed scores to the last bit of exp()."""
    import __graft_entry__ as g
    g.build()
    from centerpose_amd.detector import soft_nms_39
    from oracle import dcn as odcn
    _check_nms_against_golden(soft_nms_39, golden_dir)
    _check_nms_against_golden(odcn.soft_nms_39, golden_dir)


@pytest.mark.reference
def test_soft_nms_39_golden_reproduces_from_reference_source(golden_dir):
    """The committed soft_nms_39.npz is what the reference source computes today: regenerate it in memory and compare all bytes."""
    sys.path.insert(0, golden_dir)
    import make_golden_nms
    fresh = make_golden_nms.generate()
    gold = np.load(os.path.join(golden_dir, "soft_nms_39.npz"))
    assert sorted(fresh) == sorted(gold.files)
    for k in fresh:
        assert np.array_equal(fresh[k], gold[k]), k


def test_order_is_topological_and_list_schedule():
    """engine.order_is_topological: the check that guards a schedule-cache hit.  list_schedule always returns a
    topological order of the DAG it was given; the same order is rejected for a DAG with one more (WAR-like) edge against it."""
    from centerpose_amd.engine import list_schedule, order_is_topological
    deps = [[], [0], [0], [1], [2], [3, 4]]
    order, assign, span = list_schedule(deps, [1.0, 2.0, 1.0, 1.0, 3.0, 1.0], 2)
    assert order_is_topological(deps, order) and sorted(order) == list(range(6)) and set(assign) <= {0, 1}
    assert order_is_topological(deps, list(range(6)))
    assert not order_is_topological(deps, [1, 0, 2, 3, 4, 5])          # child before parent
    assert not order_is_topological(deps, [0, 1, 2, 3, 4])             # not a permutation
    assert not order_is_topological(deps, [0, 1, 2, 3, 4, 4])
    # same names, one more edge (buffer reuse): launch 2 must now follow launch 3 -- an order that runs 2 before 3 is refused
    deps2 = [[], [0], [0, 3], [1], [2], [3, 4]]
    bad = [0, 2, 1, 3, 4, 5]
    assert order_is_topological(deps, bad) and not order_is_topological(deps2, bad)


def test_list_schedule_prefers_flagged_light_launch_on_ties():
    """The head section of the step: feature map -> six head launches -> (peak extraction after two of them) -> assignment after all.
    Without `prefer` the peak extraction (short remaining path) is placed behind the last head; flagged, it takes the tie at the moment
    its two heads are done and runs beside the remaining heads.  Either way the order is topological and the simulated span equal."""
    from centerpose_amd.engine import list_schedule, order_is_topological
    #        0 feat, 1 hm_hp, 2 hps, 3 hm, 4 wh, 5 reg, 6 hp_offset, 7 nms_topk (needs 1, 3), 8 assign (needs all)
    deps = [[], [0], [0], [0], [0], [0], [0], [1, 3], [2, 4, 5, 6, 7]]
    dur = [1.0, 0.314, 0.361, 0.268, 0.27, 0.27, 0.27, 0.056, 0.014]
    plain, _, span0 = list_schedule(deps, dur, 2)
    pref, assign, span1 = list_schedule(deps, dur, 2, [i == 7 for i in range(9)])
    assert order_is_topological(deps, plain) and order_is_topological(deps, pref)
    assert abs(span0 - span1) < 1e-9
    assert plain.index(7) == 7 and pref.index(7) < plain.index(7)           # behind every head before; earlier now
    assert sum(1 for i in (4, 5, 6) if pref.index(i) > pref.index(7)) >= 2   # at least two heads still to run beside it
    assert pref[-1] == 8 and set(assign) <= {0, 1}


def test_resdcn_checkpoint_keys_are_the_standalone_models():
    """`resdcn` (resnet_dcn.py) is a stand-alone PoseResNet: its state_dict has no backbone_model. / head_model. prefixes and its
    heads are attributes (hm.0.weight ...).  The spec / synthetic checkpoint use that spelling; the engine maps it onto the graph's
    keys (`nets.internal_key`), also under a DataParallel 'module.' prefix."""
    from centerpose_amd import engine, nets, synth
    spec, _ = nets.param_spec("resdcn_18")
    assert "conv1.weight" in spec and "hm.0.weight" in spec and "deconv_layers.0.conv_offset_mask.weight" in spec
    assert not any(k.startswith(("backbone_model.", "head_model.")) for k in spec)
    assert nets.canonical_arch("resdcn_18") == "resdcn_18" and nets.canonical_arch("ResDCN_50") == "resdcn_50"
    with pytest.raises(ValueError):
        nets.canonical_arch("resdcn_19")
    sd = synth.make_state_dict("resdcn_18")
    assert set(sd) == set(spec)
    inner = engine.normalize_state_dict({"module." + k: v for k, v in sd.items()}, "resdcn_18")
    ispec, _ = nets.param_spec("resdcn_18", internal=True)
    assert set(inner) == set(ispec) and "backbone_model.conv1.weight" in inner and "head_model.hm.0.weight" in inner
    assert nets.internal_key("dla_34", "backbone_model.base.level0.0.weight") == "backbone_model.base.level0.0.weight"


def test_process_many_groups_batches_in_order_and_falls_back(monkeypatch):
    """model.BackBoneWithHead.process_many / detector.MultiPoseDetector.process_stream (the steps-in-flight entry points),
    host logic only: batches are taken `depth` at a time IN ORDER; a full group of equal shapes goes through ONE joint replay
    (`pipeline_for(...).process_all(group)`) and yields one fresh `dets` per batch; a short last group, a group of mixed shapes
    (FIX_RES = false) and depth <= 1 go through `process`, one call per batch; a configuration that needs host logic between forward
    and decode (FLIP_TEST) never reaches the pipeline.  Device code is replaced by recording stand-ins: no GPU here."""
    from centerpose_amd import config, detector, model
    cfg = config.get_cfg("res_50", TEST__FLIP_TEST=False)
    m = model.create_model(cfg.MODEL.NAME, cfg.MODEL.HEAD_CONV, cfg)
    calls = []

    class FakePipe:
        def __init__(self, key):
            self.key = key

        def process_all(self, group):
            calls.append(("joint", self.key, [int(x[0, 0, 0, 0]) for x in group]))
            return [(["outs%d" % int(x[0, 0, 0, 0])], torch.full((1, 2), float(x[0, 0, 0, 0]))) for x in group]

    def fake_process(x, K=100):
        calls.append(("single", tuple(x.shape), int(x[0, 0, 0, 0])))
        return ["outs%d" % int(x[0, 0, 0, 0])], torch.full((1, 2), float(x[0, 0, 0, 0]))

    monkeypatch.setattr(m, "pipeline_for", lambda B, H, W, K, depth: FakePipe((B, H, W, K, depth)))
    monkeypatch.setattr(m, "process", fake_process)
    mk = lambda i, h=8: torch.full((2, 3, h, 8), float(i))

    def consumed():                                   # the generator must pull only `depth` batches ahead of what it has yielded
        for i in range(5):
            pulled.append(i)
            yield mk(i)
    pulled = []
    gen = m.process_many(consumed(), K=7, depth=2)
    first = next(gen)
    assert pulled == [0, 1] and first[0] == ["outs0"] and float(first[1][0, 0]) == 0.0
    rest = list(gen)
    assert [o[0] for o, _ in [first] + rest] == ["outs%d" % i for i in range(5)]
    assert calls == [("joint", (2, 8, 8, 7, 2), [0, 1]), ("joint", (2, 8, 8, 7, 2), [2, 3]), ("single", (2, 3, 8, 8), 4)]
    # mixed shapes inside a group -> singles for that group only; depth 3; depth 1
    calls.clear()
    got = list(m.process_many([mk(0), mk(1, 16), mk(2), mk(3)], depth=2))
    assert [c[0] for c in calls] == ["single", "single", "joint"] and len(got) == 4
    calls.clear()
    list(m.process_many([mk(i) for i in range(7)], depth=3))
    assert [c[0] for c in calls] == ["joint", "joint", "single"] and calls[0][2] == [0, 1, 2]
    calls.clear()
    list(m.process_many([mk(i) for i in range(3)], depth=1))
    assert [c[0] for c in calls] == ["single"] * 3
    assert list(m.process_many([], depth=2)) == []
    # the detector: FLIP_TEST off -> the model's stream; FLIP_TEST on -> process() per batch
    det = object.__new__(detector.MultiPoseDetector)
    det.cfg, det.model = cfg, m
    calls.clear()
    assert len(list(det.process_stream([mk(0), mk(1)], depth=2))) == 2 and calls[0][0] == "joint"
    det.cfg = config.get_cfg("res_50", TEST__FLIP_TEST=True)
    seen = []
    monkeypatch.setattr(detector.MultiPoseDetector, "process", lambda self, images, return_time=False: seen.append(int(images[0, 0, 0, 0])) or ("o", "d"))
    assert list(det.process_stream([mk(5), mk(6)], depth=2)) == [("o", "d")] * 2 and seen == [5, 6]
