"""1x1 conv: exact-fp32 MFMA kernel vs the bf16x3 pointwise kernel on the bench layers."""
import math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev = 'cuda'
CASES = [(18, 64, 256, 512), (18, 64, 512, 256), (18, 32, 1024, 256), (18, 32, 256, 1024),
         (18, 128, 64, 256), (18, 128, 256, 64), (18, 256, 64, 128), (18, 256, 128, 64), (18, 64, 512, 64),
         (18, 128, 128, 256), (18, 64, 128, 512), (18, 64, 64, 512), (3, 128, 64, 256), (3, 64, 128, 512)]
for B, H, Cin, Cout in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev)
    s = (0.5 + torch.rand(B, Cin, generator=g)).to(dev)
    t = (0.3 * torch.randn(B, Cin, generator=g)).to(dev)
    res = {}
    for name, wf in (('f32', 0), ('pw', 3)):
        wp = O.pack_conv_weight(w, 1, Cout, Cin, wfmt=wf)
        kw = dict(pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t, pro_bstride=Cin, wfmt=wf)
        for _ in range(3):
            y, _ = O.conv(x, wp, B, H, H, Cin, Cout, 1, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y, _ = O.conv(x, wp, B, H, H, Cin, Cout, 1, **kw)
        e1.record(); torch.cuda.synchronize()
        res[name] = (y, e0.elapsed_time(e1) / 10)
    nb = 2
    a = F.relu(x[:nb].double() * s[:nb].double().view(nb, 1, 1, Cin) + t[:nb].double().view(nb, 1, 1, Cin))
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double()).permute(0, 2, 3, 1)
    fl = 2.0 * B * H * H * Cin * Cout
    by = 4.0 * B * H * H * (Cin + Cout)
    out = '%2dx%3d^2 %4d->%4d:' % (B, H, Cin, Cout)
    for name in ('f32', 'pw'):
        y, ms = res[name]
        err = (y[:nb].double() - ref).abs().max().item() / ref.abs().max().item()
        out += '  %s %.3f ms %5.0f TF %5.0f GB/s err %.1e' % (name, ms, fl / ms / 1e9, by / ms / 1e6, err)
    out += '  x%.2f' % (res['f32'][1] / res['pw'][1])
    print(out)
