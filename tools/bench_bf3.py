"""3x3 conv: exact-fp32 MFMA kernel vs the bf16x3 (fp32-equivalent) kernel on the big
layers of the bench workload: accuracy vs an fp64 reference and time per launch."""
import math, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N

dev = 'cuda'
CASES = [(18, 256, 64, 64), (18, 128, 128, 128), (18, 64, 256, 256), (18, 32, 512, 512),
         (18, 16, 512, 512), (9, 64, 256, 256)]
for B, H, Cin, Cout in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dev)
    s = (0.5 + torch.rand(B, Cin, generator=g)).to(dev)
    t = (0.3 * torch.randn(B, Cin, generator=g)).to(dev)
    res = {}
    for name, wf, form in (('f32', 0, 0), ('bf16x3', 1, 0), ('wino', 2, N.FORM_WINO_ANY | N.FORM_WINO_8X16),
                           ('wino16', 2, N.FORM_WINO_ANY)):
        O.DEFAULT_FORM = form
        wp = O.pack_conv_weight(w, 9, Cout, Cin, wfmt=wf)
        for _ in range(3):
            y, _ = O.conv(x, wp, B, H, H, Cin, Cout, 9, pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t,
                          pro_bstride=Cin, wfmt=wf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y, _ = O.conv(x, wp, B, H, H, Cin, Cout, 9, pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t,
                          pro_bstride=Cin, wfmt=wf)
        e1.record(); torch.cuda.synchronize()
        res[name] = (y, e0.elapsed_time(e1) / 10)
    # fp64 reference on a sub-batch (GPU torch fp64)
    nb = min(B, 2)
    a = F.relu(x[:nb].double() * s[:nb].double().view(nb, 1, 1, Cin) + t[:nb].double().view(nb, 1, 1, Cin))
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    fl = 2.0 * B * H * H * Cin * Cout * 9
    out = '%2dx%3d^2 %3d->%3d:' % (B, H, Cin, Cout)
    for name in ('f32', 'bf16x3', 'wino', 'wino16'):
        y, ms = res[name]
        err = (y[:nb].double() - ref).abs().max().item() / ref.abs().max().item()
        out += '  %s %.3f ms %4.0f TF %.0e' % (name, ms, fl / ms / 1e9, err)
    out += '  wino/direct %.2fx  wino16/wino %.2fx' % (res['bf16x3'][1] / res['wino'][1], res['wino'][1] / res['wino16'][1])
    print(out)
