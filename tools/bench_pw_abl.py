"""time the fp16 x 2 pointwise kernel on bench layers -- used with the ablation builds of csrc/p2l_pw.hip
(tools/ab_build.sh p2l_pw -DP2L_PW_ABL=n, P2L_LIB_PATH=...)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev, B = 'cuda', 18
out = []
for H, Cin, Cout in ((64, 256, 512), (64, 512, 256), (32, 1024, 256), (32, 256, 1024), (128, 64, 256), (256, 64, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
    s = (0.5 + torch.rand(B, Cin, generator=g)).to(dev); t = (0.3 * torch.randn(B, Cin, generator=g)).to(dev)
    wp = O.pack_conv_weight((torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev), 1, Cout, Cin, wfmt=3)
    f = lambda: O.conv(x, wp, B, H, H, Cin, Cout, 1, wfmt=3, amax_in=am, pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t, pro_bstride=Cin)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    out.append('%d^2 %d->%d %.3f' % (H, Cin, Cout, e0.elapsed_time(e1) / 20))
print('%-28s' % os.path.basename(os.environ.get('P2L_LIB_PATH', 'product')), ' | '.join(out))
