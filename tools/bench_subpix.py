"""Sub-pixel forward launches alone (fp16 x 2): two output phases per block (default) vs one (P2L_FORM_NO_SP_PAIR),
BigGAN's nearest-upsample convs and StyleGAN2's transposed convs (ext = 1), ms per launch."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import _native as N, ops as O
dev = torch.device('cuda:0')
CASES = [  # (B, h (low res), Cin, Cout, ext)
    (18, 64, 64, 64, 0), (18, 32, 128, 128, 0), (18, 128, 32, 32, 0), (18, 16, 256, 256, 0),
    (32, 256, 128, 64, 1), (32, 64, 512, 256, 1), (32, 128, 256, 128, 1), (32, 32, 512, 512, 1),
    (3, 512, 64, 32, 1), (3, 256, 128, 64, 1), (3, 64, 512, 256, 1)]
for B, h, ci, co, ext in CASES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, h, h, ci, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).to(dev)
    wp = O.pack_conv_weight_subpix(w, co, ci, mode=1 if ext else 0, wfmt=2)
    am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
    row = []
    for form in (N.FORM_SP_PAIR, N.FORM_NO_SP_PAIR):
        fn = lambda: O.conv(x, wp, B, 2 * h, 2 * h, ci, co, 9, wfmt=2, ups=2, ext=ext, form=form, amax_in=am, splitk=1)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 10)
    print('B %2d  %4d^2 -> %4d^2  %3d -> %3d  ext %d : pair %.3f ms   single %.3f ms   x %.2f' % (
        B, h, 2 * h, ci, co, ext, row[0], row[1], row[1] / row[0]))
