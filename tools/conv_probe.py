"""launch a few conv shapes several times (for rocprofv3 --pmc runs, tools/gpu_pmc_conv.sh).
Each (arithmetic, form) block is announced by a marker launch count so that the dispatches can be told
apart in the counter CSV by kernel name: direct bf16x3 (conv_mfma_kernel) | direct fp16x2
(conv_h2_kernel) | Winograd 16x16 bf16x3 / fp16x2 (wino16s_conv_kernel<.., false | true>) | sub-pixel
fp16x2 (conv_h2_kernel<4, ..>)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops, _native as N
dev = torch.device('cuda'); B = 18
SH = [(256, 64, 64), (128, 128, 128), (64, 256, 256), (32, 512, 512)]
for WF, FORM in ((1, 0), (2, N.FORM_NO_WINO), (2, N.FORM_WINO_ANY | N.FORM_WINO_BF3), (2, N.FORM_WINO_ANY)):
    ops.DEFAULT_FORM = FORM
    for H, Cin, Cout in SH:
        x = torch.randn(B, H, H, Cin, device=dev)
        wp = ops.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9), 9, Cout, Cin, wfmt=WF)
        bias = torch.randn(Cout, device=dev)
        for _ in range(4):
            ops.conv(x, wp, B, H, H, Cin, Cout, 9, bias=bias, wfmt=WF)
ops.DEFAULT_FORM = N.FORM_AUTO
for H, Cin, Cout in ((128, 128, 128), (64, 256, 256)):       # sub-pixel up-conv forward (fp16 x 2)
    x = torch.randn(B, H // 2, H // 2, Cin, device=dev)
    wp = ops.pack_conv_weight_subpix(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9), Cout, Cin, wfmt=2)
    for _ in range(4):
        ops.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2, ups=2)
torch.cuda.synchronize()
