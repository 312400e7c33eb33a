"""launch a few conv shapes several times (for rocprofv3 --pmc runs)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops, _native as N
dev = torch.device('cuda'); B = 18
SH = [(256, 64, 64, 9), (64, 256, 256, 9), (32, 512, 512, 9)]
# direct bf16x3 | Winograd 8x16 bf16x3 | Winograd 16x16 bf16x3 | Winograd 16x16 fp16x2 (+ its max-|x| pass)
for WF, FORM in ((1, 0), (2, N.FORM_WINO_ANY | N.FORM_WINO_8X16), (2, N.FORM_WINO_ANY | N.FORM_WINO_BF3),
                 (2, N.FORM_WINO_ANY)):
  ops.DEFAULT_FORM = FORM
  for H, Cin, Cout, taps in SH:
      k = 3 if taps == 9 else 1
      x = torch.randn(B, H, H, Cin, device=dev)
      wp = ops.pack_conv_weight(torch.randn(Cout, Cin, k, k, device=dev) / math.sqrt(Cin * k * k), taps, Cout, Cin, wfmt=WF if taps == 9 else 0)
      bias = torch.randn(Cout, device=dev)
      for _ in range(4):
          ops.conv(x, wp, B, H, H, Cin, Cout, taps, bias=bias, wfmt=WF if taps == 9 else 0)
torch.cuda.synchronize()
