"""DCN offset statistics of the synthetic checkpoint (GPU box): per layer max |offset| and share beyond R px."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import engine, synth, ops

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = 4
oms = []
orig = engine.PlanBuilder.emit_dcn
def hook(self, x, name, co):
    n0 = self.bytes_alloc
    out = orig(self, x, name, co)
    return out
eng = engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, use_graph=False)
# om buffers: re-run eagerly and intercept ops.dcn_v2
real = ops.dcn_v2
def spy(x, om, *a, **k):
    oms.append(om)
    return real(x, om, *a, **k)
ops.dcn_v2 = spy
eng.input.copy_(synth.make_images(B).cuda())
eng.run_eager()
torch.cuda.synchronize()
for i, om in enumerate(oms):
    off = om[..., :18].abs()
    print("dcn %2d  HxW %3dx%-3d  max|off| %6.2f  mean %5.2f  >1:%5.1f%% >2:%5.1f%% >3:%5.1f%% >4:%5.1f%%" % (
        i, om.shape[1], om.shape[2], off.max().item(), off.mean().item(), *[100.0 * (off > r).float().mean().item() for r in (1, 2, 3, 4)]))
