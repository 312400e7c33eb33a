"""3x3 conv launches of the bench workload, one by one: the 16x16 Winograd kernel in fp16 x 2 (the
default where it applies) against the DIRECT kernel in fp16 x 2 (csrc/p2l_h2.hip, P2L_FORM_NO_WINO) and
in bf16 x 3 (P2L_FORM_WINO_BF3) -- VERDICT round 3 #1 "measure a direct fp16 x 2 kernel honestly against
it" -- plus the sub-pixel and small-grid launches that have no Winograd form.  Time per launch, max-|x|
pass included (no maxima handed over here); accuracy against fp64 on two images.
usage: python tools/bench_h2.py [B]"""
import math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N

dev = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 18
# (H, Cin, Cout, ups)
CASES = [(256, 64, 64, 0), (128, 128, 128, 0), (128, 64, 64, 0), (64, 256, 256, 0), (64, 128, 128, 0),
         (32, 512, 512, 0), (32, 256, 256, 0), (16, 512, 512, 0), (16, 256, 256, 0), (8, 512, 512, 0), (4, 512, 512, 0),
         (256, 64, 64, 2), (128, 128, 128, 2), (64, 256, 256, 2), (32, 256, 256, 2),
         (256, 64, 64, 3), (128, 128, 128, 3), (64, 256, 256, 3)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) / n


for H, Cin, Cout, ups in CASES:
    g = torch.Generator().manual_seed(0)
    Hin = H // 2 if ups == 2 else H
    x = torch.randn(B, Hin, Hin, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dev)
    if ups:
        wp = O.pack_conv_weight_subpix(w, Cout, Cin, flip=(ups == 3), wfmt=2)
    else:
        wp = O.pack_conv_weight(w, 9, Cout, Cin, wfmt=2)
    forms = [('direct-h2', N.FORM_NO_WINO), ('direct-bf3', N.FORM_NO_WINO | N.FORM_WINO_BF3)]
    if not ups and H % 16 == 0:
        forms = [('wino-h2', N.FORM_AUTO)] + forms
    res = {}
    for name, form in forms:
        O.DEFAULT_FORM = form
        res[name] = timed(lambda: O.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2, ups=ups)[0])
    O.DEFAULT_FORM = N.FORM_AUTO
    nb = min(B, 2)
    if ups == 0:
        ref = F.conv2d(x[:nb].double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    elif ups == 2:
        ref = F.conv2d(F.interpolate(x[:nb].double().permute(0, 3, 1, 2), scale_factor=2, mode='nearest'),
                       w.double(), padding=1).permute(0, 2, 3, 1)
    else:
        ref = None
    fl = 2.0 * B * H * H * Cin * Cout * 9
    out = '%2d x %3d^2 %3d->%3d ups %d:' % (B, H, Cin, Cout, ups)
    for name, _ in forms:
        y, ms = res[name]
        err = (y[:nb].double() - ref).abs().max().item() / ref.abs().max().item() if ref is not None else float('nan')
        out += '  %s %.3f ms %4.0f TF %.0e' % (name, ms, fl / ms / 1e9, err)
    if 'wino-h2' in res:
        out += '  | wino/direct-h2 %.2fx' % (res['direct-h2'][1] / res['wino-h2'][1])
    out += '  direct h2/bf3 %.2fx' % (res['direct-bf3'][1] / res['direct-h2'][1])
    print(out, flush=True)
