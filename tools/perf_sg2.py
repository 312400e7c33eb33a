"""StyleGAN2 full-size timing + size-independent checks (run on the GPU box)."""
import sys, time, warnings
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
from pix2latent_amd.model.stylegan2 import StyleGAN2
from pix2latent_amd import _native as N
import ctypes as C

for spec in sys.argv[1:] or ['512:9', '1024:4']:
    size, B = [int(v) for v in spec.split(':')]
    t = time.time()
    m = StyleGAN2(search='z', size=size, device='cuda')
    torch.cuda.synchronize()
    print('size %d: build %.1fs  ws(B=%d) %.2f GB' % (
        size, time.time() - t, B, N.lib().p2l_sg2_ws_bytes(C.byref(m._desc), B) / 2**30))
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, 512, generator=g).cuda().requires_grad_(True)
    noises = [torch.randn(B, 1, s[2], s[3], generator=g).cuda() for s in m.noise_shape]
    outs, grads = [], []
    for _ in range(2):
        z.grad = None
        o = m.forward_z(z, noises=noises)
        (o * o).mean().backward()
        outs.append(o.detach().clone()); grads.append(z.grad.clone())
    print('   finite %s  range [%.3f, %.3f]  deterministic fwd %s bwd %s' % (
        bool(torch.isfinite(outs[0]).all() and torch.isfinite(grads[0]).all()), outs[0].min().item(),
        outs[0].max().item(), torch.equal(outs[0], outs[1]), torch.equal(grads[0], grads[1])))
    # batch independence: sample 0 alone == sample 0 in the batch (bit-exact)
    o1 = m.forward_z(z[:1].detach(), noises=[n[:1] for n in noises])
    print('   batch-independent: max|d| %.2e' % (o1 - outs[0][:1]).abs().max().item())
    for _ in range(2):
        o = m.forward_z(z, noises=noises); (o * o).mean().backward()
    torch.cuda.synchronize()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(5):
        e0.record(); o = m.forward_z(z, noises=noises); l = (o * o).mean(); e1.record()
        l.backward(); e2.record(); torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    print('   B=%d fwd %.2f ms  bwd %.2f ms  -> %.1f evals/s (generator only)' % (
        B, tf / 5, tb / 5, B / ((tf + tb) / 5e3)))
    del m
    torch.cuda.empty_cache()
