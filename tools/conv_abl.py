import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops, _native as N
dev = torch.device('cuda'); B = 9
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = []
for H, Cin, Cout in [(256, 64, 64), (128, 128, 128)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    wp = ops.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9), 9, Cout, Cin)
    ms = timeit(lambda: ops.conv(x, wp, B, H, H, Cin, Cout, 9))
    out.append('%d/%d: %.3f ms %.1f TF' % (H, Cin, ms, 2.0 * B * H * H * Cin * Cout * 9 / ms / 1e9))
print('ABL=%s FORCE=%s  ' % (os.environ.get('P2L_ABL'), os.environ.get('P2L_CONV_FORCE')) + ' | '.join(out))
