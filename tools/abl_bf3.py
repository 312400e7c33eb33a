"""per-tile fixed cost vs per-chunk cost of the bf16x3 3x3 kernel: time vs Cin at fixed output"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev='cuda'
for (B,H,Cout) in [(18,64,256),(18,256,64)]:
    for Cin in (16,32,64,128,256):
        g=torch.Generator().manual_seed(0)
        x=torch.randn(B,H,H,Cin,generator=g).to(dev)
        w=(torch.randn(Cout,Cin,3,3,generator=g)/math.sqrt(Cin*9)).to(dev)
        wp=O.pack_conv_weight(w,9,Cout,Cin,wfmt=1)
        for _ in range(3): O.conv(x,wp,B,H,H,Cin,Cout,9,wfmt=1,splitk=1)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): O.conv(x,wp,B,H,H,Cin,Cout,9,wfmt=1,splitk=1)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print('%dx%d^2 %3d->%d (%2d chunks): %.3f ms  %.1f TF-equiv' % (B,H,Cin,Cout,Cin//16,ms,2.0*B*H*H*Cin*Cout*9/ms/1e9))
