"""Micro-benchmark of the conv / gemm kernels at the BigGAN-256 + VGG16 shapes
(B = 9 candidates per chunk).  Prints achieved fp32 TFLOP/s per shape."""
import math
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import ops, _native as N  # noqa: E402

dev = torch.device('cuda')
B = int(os.environ.get('P2L_B', '9'))

SHAPES = [
    # name, H, Cin, Cout, taps, ups, pro
    ('blk11.conv2 3x3 64->64 @256', 256, 64, 64, 9, False, True),
    ('blk11.conv1 3x3 64->64 @256 ups', 256, 64, 64, 9, True, True),
    ('blk9.conv2 3x3 128->128 @128', 128, 128, 128, 9, False, True),
    ('blk7.conv2 3x3 256->256 @64', 64, 256, 256, 9, False, True),
    ('blk5.conv2 3x3 256->256 @32', 32, 256, 256, 9, False, True),
    ('blk3.conv2 3x3 512->512 @16', 16, 512, 512, 9, False, True),
    ('blk1.conv2 3x3 512->512 @8', 8, 512, 512, 9, False, True),
    ('blk0.conv1 3x3 512->512 @4', 4, 512, 512, 9, False, True),
    ('blk11.conv0 1x1 256->64 @128', 128, 256, 64, 1, False, True),
    ('blk11.conv3 1x1 64->128 @256', 256, 64, 128, 1, False, True),
    ('blk7.conv3 1x1 256->512 @64', 64, 256, 512, 1, False, True),
    ('blk0.conv0 1x1 2048->512 @4', 4, 2048, 512, 1, False, True),
    ('vgg.conv1_2 3x3 64->64 @256', 256, 64, 64, 9, False, False),
    ('vgg.conv3_2 3x3 256->256 @64', 64, 256, 256, 9, False, False),
    ('vgg.conv4_2 3x3 512->512 @32', 32, 512, 512, 9, False, False),
    ('vgg.conv5_2 3x3 512->512 @16', 16, 512, 512, 9, False, False),
    ('rgb 3x3 128->32(3) @256', 256, 128, 32, 9, False, True),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, H, Cin, Cout, taps, ups, pro in SHAPES:
    k = 3 if taps == 9 else 1
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / math.sqrt(Cin * k * k)
    wp = ops.pack_conv_weight(w, taps, Cout, Cin)
    bias = torch.randn(Cout, device=dev)
    s = torch.rand(B, Cin, device=dev) + 0.5
    t = torch.randn(B, Cin, device=dev)
    kw = dict(bias=bias, ups=ups)
    if pro:
        kw.update(pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t, pro_bstride=Cin)
    ms = timeit(lambda: ops.conv(x, wp, B, H, H, Cin, Cout, taps, **kw))
    flops = 2.0 * B * H * H * Cin * Cout * taps
    print('%-36s %8.3f ms  %7.2f TFLOP/s' % (name, ms, flops / ms / 1e9), flush=True)

# attention bmm's
P, C = 4096, 512
th = torch.randn(B, P, C // 8, device=dev)
ph = torch.randn(B, P // 4, C // 8, device=dev)
gp = torch.randn(B, P // 4, C // 2, device=dev)
ms = timeit(lambda: ops.gemm(th, ph, B, P, P // 4, C // 8))
print('%-36s %8.3f ms  %7.2f TFLOP/s' % ('attn QK^T', ms, 2.0 * B * P * (P // 4) * (C // 8) / ms / 1e9))
Pm = torch.rand(B, P, P // 4, device=dev)
ms = timeit(lambda: ops.gemm(Pm, gp, B, P, C // 2, P // 4, b_kmajor=True))
print('%-36s %8.3f ms  %7.2f TFLOP/s' % ('attn PV', ms, 2.0 * B * P * (P // 4) * (C // 2) / ms / 1e9))
dag = torch.randn(B, P, C // 2, device=dev)
ms = timeit(lambda: ops.gemm(Pm, dag, B, P // 4, C // 2, P, a_kmajor=True, b_kmajor=True))
print('%-36s %8.3f ms  %7.2f TFLOP/s' % ('attn dV = P^T dO', ms, 2.0 * B * P * (P // 4) * (C // 2) / ms / 1e9))
