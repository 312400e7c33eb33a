"""phase timeline of one block of the 16x16-pixel Winograd kernel (s_memtime stamps)"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops, _native as N
import ctypes as C
dev = torch.device('cuda'); B = 18
H, Cin, Cout = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 256, 256))]
N.check(N.lib().p2l_set_wino_block(2))
x = torch.randn(B, H, H, Cin, device=dev)
wp = ops.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9), 9, Cout, Cin, wfmt=2)
tr = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2)
torch.cuda.synchronize()
N.lib().p2l_wino_set_trace(C.c_void_p(tr.data_ptr()))
ops.conv(x, wp, B, H, H, Cin, Cout, 9, wfmt=2)
torch.cuda.synchronize()
N.lib().p2l_wino_set_trace(C.c_void_p(0))
t = tr.cpu().numpy().reshape(8, 64, 8)
nch = Cin // 16
t0 = t[:, 0, 0].min()
print('chunks', nch, ' total loop ticks', t[:, 0, 5].max() - t0, '(s_memtime ticks)')
print('per chunk: [wave]  top->phase1  phase1->phase2  ->barrierB  ->raw written  | chunk period')
for c in (1, 2, nch // 2, nch - 2):
    for w in (0, 1, 4, 5):
        a = t[w, c]
        per = t[w, c + 1, 0] - a[0] if c + 1 < nch else 0
        print('c%2d w%d  %6d %6d %6d %6d | %6d   start %+d' % (c, w, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], per, a[0] - t0))
per = [(t[:, c + 1, 0] - t[:, c, 0]).mean() for c in range(nch - 1)]
print('mean chunk period', np.mean(per))
