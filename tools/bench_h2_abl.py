"""time the direct fp16 x 2 kernel (P2L_FORM_NO_WINO) on three bench layers -- used with the ablation builds
of csrc/p2l_h2.hip (tools/ab_build.sh p2l_h2 -DP2L_H2_ABL=n, P2L_LIB_PATH=...)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev, B = 'cuda', 18
out = []
for H, Cin, Cout in ((256, 64, 64), (64, 256, 256), (32, 512, 512)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
    wp = O.pack_conv_weight((torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dev), 9, Cout, Cin, wfmt=2)
    O.DEFAULT_FORM = N.FORM_NO_WINO
    f = lambda: O.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2, amax_in=am)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    out.append('%d^2 %d->%d %.3f ms' % (H, Cin, Cout, e0.elapsed_time(e1) / 20))
print(os.environ.get('P2L_LIB_PATH', 'product'), ' | '.join(out))
