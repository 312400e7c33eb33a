"""the direct fp16 x 2 kernel alone on 64^2 256->256 x 18 (for rocprofv3 --pmc runs of ablation builds)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops as O, _native as N
dev, B, H, Cin, Cout = 'cuda', 18, 64, 256, 256
x = torch.randn(B, H, H, Cin, device=dev)
am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
wp = O.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9), 9, Cout, Cin, wfmt=2)
O.DEFAULT_FORM = N.FORM_NO_WINO
for _ in range(4):
    O.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2, amax_in=am)
torch.cuda.synchronize()
