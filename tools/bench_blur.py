"""StyleGAN2 FIR kernels alone: GB/s of p2l_sg2_blur_fwd / _bwd (algorithmic bytes: the frame read once, the output
written once) at the shapes of bench.py's config.extra C4 (cars 512^2 x 32) and C5 (FFHQ 1024^2 x 3)."""
import ctypes as ct, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd import _native as N
lib = N.lib()
dev = torch.device('cuda:0')
SHAPES = [(32, 512, 64), (32, 256, 128), (32, 128, 256), (32, 64, 512), (3, 1024, 32), (3, 512, 64), (3, 256, 128)]
for B, H, C in SHAPES:
    u = torch.randn(B, H + 2, H + 2, C, device=dev)
    d = torch.rand(B, C, device=dev) + 0.5
    noise = torch.randn(B, H * H, device=dev)
    bias = torch.zeros(C, device=dev)
    y = torch.empty(B, H, H, C, device=dev)
    am = torch.zeros(B, 64, device=dev)
    st = N.stream()
    for name, fn, nbytes in (
            ('blur_fwd', lambda: lib.p2l_sg2_blur_fwd(N.ptr(u), N.ptr(d), N.ptr(noise), ct.c_float(0.3), N.ptr(bias), N.ptr(y), B, H, H, C, st), 4.0 * (u.numel() + y.numel())),
            ('blur_fwd+amax', lambda: lib.p2l_sg2_blur_fwd_amax(N.ptr(u), N.ptr(d), N.ptr(noise), ct.c_float(0.3), N.ptr(bias), N.ptr(y), B, H, H, C, N.ptr(d), N.ptr(am), st), 4.0 * (u.numel() + y.numel())),
            ('blur_bwd', lambda: lib.p2l_sg2_blur_bwd(N.ptr(y), N.ptr(u), B, H, H, C, st), 4.0 * (u.numel() + y.numel())),
            ('blur_bwd+amax', lambda: lib.p2l_sg2_blur_bwd_amax(N.ptr(y), N.ptr(u), B, H, H, C, N.ptr(am), st), 4.0 * (u.numel() + y.numel()))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%-14s B %2d  %4d^2 x %3d : %7.3f ms  %6.0f GB/s' % (name, B, H, C, ms, nbytes / ms / 1e6))
