"""Split-K slices of the 4^2 ... 16^2 layers: time per launch (conv + finish kernel, HIP events of the
library's launch profiler) for every slice count, at the local batch sizes of 1 ... 8 GPUs.  Input of
the round-5 decision to make the slice count a function of the LAYER SHAPE only (a candidate's bits
must not depend on who shares its launch): which fixed count costs what, where.

    python tools/splitk_sweep.py > gpurun_out/splitk_sweep.txt
"""
import os, sys, math, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import _native as N, ops as O

dev = torch.device('cuda:0')
lib = N.lib()
# (taps, H, Cin, Cout, ups): the split-K regime of BigGAN-deep-256 (forward shapes; the input-gradient
# convs are the same shapes with the channel counts swapped)
SHAPES = [
    (9, 4, 512, 512, 0), (9, 8, 512, 512, 0), (9, 8, 512, 512, 1), (9, 16, 512, 512, 1), (9, 16, 256, 256, 0),
    (1, 4, 2048, 512, 0), (1, 4, 512, 2048, 0), (1, 8, 2048, 512, 0), (1, 8, 512, 2048, 0),
    (1, 16, 512, 1024, 0), (1, 16, 1024, 512, 0), (1, 16, 1024, 256, 0), (1, 16, 256, 1024, 0),
]
BATCHES = [int(b) for b in os.environ.get('BATCHES', '2,3,5,9,18').split(',')]
REPS = 30


def timed(fn):
    N.check(lib.p2l_prof_begin(4 * REPS), 'p2l_prof_begin')
    lib.p2l_prof_step(0, 1)
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    T = N.prof_end()
    return 1e3 * (T.ms[0] + T.ms[1]) / max(T.count[0] + T.count[1], 1)


print('us per launch (conv + finish); * = what p2l_conv_suggest_splitk gives for this batch')
for taps, H, Cin, Cout, ups in SHAPES:
    wfmt = 2 if taps == 9 else 3
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) / math.sqrt(taps * Cin)
    wp = O.pack_conv_weight(w.to(dev), taps, Cout, Cin, wfmt=wfmt)
    kc = 16 if taps == 9 else 32
    cand = [s for s in (1, 2, 4, 8, 16, 32) if s <= Cin // kc // (2 if taps == 9 else 4) or s == 1]
    print('taps %d  %dx%d  %d -> %d  ups %d' % (taps, H, H, Cin, Cout, ups))
    print('   B   ' + ''.join('S=%-7d' % s for s in cand))
    for B in BATCHES:
        Hi = H // 2 if ups else H
        x = torch.randn(B, Hi, Hi, Cin, device=dev)
        am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
        d = N.P2LConv()
        d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.wfmt, d.x_ld, d.ups = B, H, H, Cin, Cout, taps, wfmt, Cin, ups
        d.n_store = d.y_ld = d.yp_ld = Cout
        sug = lib.p2l_conv_suggest_splitk(C.byref(d))
        row = []
        for s in cand:
            kw = dict(wfmt=wfmt, splitk=s, ups=ups)
            if not ups:
                kw['amax_in'] = am
            for _ in range(3):
                O.conv(x, wp, B, H, H, Cin, Cout, taps, **kw)
            t = timed(lambda: O.conv(x, wp, B, H, H, Cin, Cout, taps, **kw))
            row.append('%6.1f%s ' % (t, '*' if s == sug else ' '))
        print('  %2d   ' % B + ''.join(row))
