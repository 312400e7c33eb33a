"""full-tile pointwise kernel in the fp16 x 2 arithmetic (pw_h2_kernel<PRO, false>): us per launch of the
1x1 layers of the bench step (>= 32^2), with and without a fused prologue, timed with the launch profiler.
A/B: P2L_LIB_PATH=tools/micro/lib_pw_shallow.so python tools/bench_pw_h2.py"""
import math, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import _native as N, ops as O
dev = torch.device('cuda:0')
lib = N.lib()
B = int(os.environ.get('P2L_POP', 18))
CASES = [(32, 1024, 256), (32, 256, 1024), (64, 512, 256), (64, 256, 512), (64, 512, 128), (64, 128, 512), (64, 512, 64),
         (64, 64, 512), (128, 256, 64), (128, 64, 256), (128, 256, 128), (128, 128, 256), (256, 128, 64), (256, 64, 128)]
REPS = 20
tot = [0.0, 0.0]
print(os.environ.get('P2L_LIB_PATH', 'product'), 'B =', B, ' us per launch: no prologue | affine+relu prologue   (TB/s of tensor traffic)')
for H, Cin, Cout in CASES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev)
    s = (0.5 + torch.rand(B, Cin, generator=g)).to(dev)
    t = (0.3 * torch.randn(B, Cin, generator=g)).to(dev)
    am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
    wp = O.pack_conv_weight(w, 1, Cout, Cin, wfmt=3)
    row = []
    for i, kw in enumerate((dict(), dict(pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t, pro_bstride=Cin))):
        fn = lambda: O.conv(x, wp, B, H, H, Cin, Cout, 1, wfmt=3, amax_in=am, **kw)
        for _ in range(3):
            fn()
        N.check(lib.p2l_prof_begin(4 * REPS), 'p2l_prof_begin')
        lib.p2l_prof_step(0, 1)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        T = N.prof_end()
        assert abs(T.mfma_flops[1] / T.exec_flops[1] - 3.0) < 1e-6, 'not the fp16 x 2 kernel'
        us = 1e3 * T.ms[1] / T.count[1]
        tot[i] += us
        row.append('%7.1f (%.2f)' % (us, 4.0 * B * H * H * (Cin + Cout) / us / 1e6))
    print('%3d^2 %5d -> %-5d  %s | %s' % (H, Cin, Cout, row[0], row[1]))
print('sum %.1f | %.1f us' % (tot[0], tot[1]))
