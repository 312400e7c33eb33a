"""fp16 x 2 Winograd kernel: 16x16-pixel / 8-wave blocks against 8x16-pixel / 4-wave blocks, per
layer shape and local batch (HIP events of the launch profiler around conv + finish kernel)."""
import os, sys, math, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import _native as N, ops as O
dev = torch.device('cuda:0')
lib = N.lib()
REPS = 30
SHAPES = [(32, 512, 512), (32, 256, 256), (32, 512, 256), (64, 256, 256), (64, 128, 128), (16, 512, 512), (128, 128, 128)]


def timed(fn):
    for _ in range(3):
        fn()
    N.check(lib.p2l_prof_begin(4 * REPS), 'p2l_prof_begin')
    lib.p2l_prof_step(0, 1)
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    T = N.prof_end()
    return 1e3 * T.ms[0] / max(T.count[0], 1)


print('us per launch   16x16 blocks | 8x16 blocks | blocks of 16x16 (x slices)')
for H, Cin, Cout in SHAPES:
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    wp = O.pack_conv_weight(w.to(dev), 9, Cout, Cin, wfmt=2)
    for B in [int(b) for b in os.environ.get('BATCHES', '1,2,3,5,9').split(',')]:
        x = torch.randn(B, H, H, Cin, device=dev)
        am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
        d = N.P2LConv()
        d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.wfmt, d.x_ld = B, H, H, Cin, Cout, 9, 2, Cin
        d.n_store = d.y_ld = d.yp_ld = Cout
        sk = lib.p2l_conv_suggest_splitk(C.byref(d))
        row = []
        for form in (N.FORM_WINO_H2_16X16, N.FORM_WINO_H2_8X16):
            O.DEFAULT_FORM = form
            row.append(timed(lambda: O.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2, amax_in=am, splitk=sk)))
        O.DEFAULT_FORM = N.FORM_AUTO
        nb = B * (H // 16) ** 2 * (Cout // 64)
        print('%3dx%-3d %4d -> %-4d B=%d   %7.1f | %7.1f   %4d x %d%s' % (H, H, Cin, Cout, B, row[0], row[1], nb, sk,
                                                                    '   <- 8x16 by default' if nb * sk <= 128 else ''))
