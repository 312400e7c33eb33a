"""time conv_h2r_kernel on 18 x 256^2 and 3 x 1024^2 (64 -> 64) -- with the ablation builds of csrc/p2l_h2r.hip
(tools/ab_build.sh p2l_h2r -DP2L_H2R_ABL=n -mllvm -pragma-unroll-threshold=400000, P2L_LIB_PATH=...)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev = 'cuda'
out = []
for B, H in ((18, 256), (3, 1024), (18, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, 64, generator=g).to(dev)
    am = x.abs().reshape(B, 2048, -1).amax(dim=2).contiguous()
    wp = O.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)).to(dev), 9, 64, 64, wfmt=2)
    f = lambda: O.conv(x, wp, B, H, H, 64, 64, 9, wfmt=2, amax_in=am, form=N.FORM_NO_WINO)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    out.append('%dx%d^2 %.3f ms' % (B, H, e0.elapsed_time(e1) / 20))
print(os.environ.get('P2L_LIB_PATH', 'product'), ' | '.join(out))
