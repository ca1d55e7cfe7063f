import torch, torch.nn.functional as F, sys
sys.path.insert(0,'/root/repo')
from centerpose_amd import ops
g=torch.Generator().manual_seed(1)
for cin,cout,H in [(64,256,128),(256,256,32),(512,512,16)]:
    x=torch.randn(2,cin,H,H,generator=g); w=torch.randn(cout,cin,3,3,generator=g)/(cin*9)**.5
    ref=F.conv2d(x.double(),w.double(),None,1,1)
    wp=ops.pack_conv_weight(w.cuda()); u=ops.pack_wino_weight(wp,cin,cout)
    sc=torch.ones(wp.shape[0],device='cuda'); sh=torch.zeros(wp.shape[0],device='cuda')
    xn=x.permute(0,2,3,1).contiguous().cuda()
    o1=torch.empty(2,H,H,cout,device='cuda'); o2=torch.empty_like(o1)
    ops.conv2d([xn],wp,sc,sh,o1,kh=3,kw=3,stride=1,pad=1,cout=cout)
    ops.conv2d([xn],wp,sc,sh,o2,kh=3,kw=3,stride=1,pad=1,cout=cout,wino=u)
    r=ref.permute(0,2,3,1)
    for nm,o in (("direct",o1),("wino",o2)):
        e=(o.cpu().double()-r).abs()
        print(cin,cout,H,nm,"max %.2e rms %.2e (ref max %.2f)"%(e.max(),e.pow(2).mean().sqrt(),r.abs().max()))
