"""Print StyleGAN2 native-vs-oracle error magnitudes (diagnostic; run on the GPU box)."""
import sys, time, warnings
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
from pix2latent_amd.utils import synthetic as S
from pix2latent_amd.model.stylegan2 import StyleGAN2
from oracle import stylegan2_ref as R

torch.set_num_threads(32)
def rel(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return ((a - b).norm() / b.norm()).item()

for size in [int(a) for a in sys.argv[1:]] or [64, 256]:
    W = S.stylegan2_weights(size, 0)
    B = 2
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, 512, generator=g)
    noises = [torch.randn(B, 1, s[2], s[3], generator=g) for s in R.noise_shapes(size)]
    probe = torch.randn(B, 3, size, size, generator=g) / size
    m = StyleGAN2(search='w+', weights=W, size=size, device='cuda')
    wplus = torch.randn(B, R.n_latent(size), 512, generator=g) * 0.5
    flat = torch.cat([n.reshape(B, -1) for n in noises], 1)
    wr, nr = wplus.clone().requires_grad_(True), flat.clone().requires_grad_(True)
    t = time.time()
    ref_raw = R.synthesis(W, wr, R.reshape_noise(nr, size), size)
    ref = ref_raw.clamp(-1, 1)
    (ref * probe).sum().backward()
    t_cpu = time.time() - t
    wd, nd = wplus.cuda().requires_grad_(True), flat.cuda().requires_grad_(True)
    out = m(wd, nd)
    (out * probe.cuda()).sum().backward()
    sat = (ref_raw.abs() >= 1).float().mean().item()
    print('size %d: |dpix|max %.3e  pix range [%.3f, %.3f] std %.3f  clamp-saturated %.3f' % (
        size, (out.detach().cpu() - ref.detach()).abs().max().item(), ref_raw.min().item(),
        ref_raw.max().item(), ref_raw.std().item(), sat))
    print('   dW+ rel %.3e (|ref| %.3e)  dnoise rel %.3e (|ref| %.3e)  cpu fwd+bwd %.2fs' % (
        rel(wd.grad, wr.grad), wr.grad.norm().item(), rel(nd.grad, nr.grad), nr.grad.norm().item(), t_cpu))
    per = []
    for l in range(wplus.shape[1]):
        per.append('%.1e' % rel(wd.grad[:, l], wr.grad[:, l]))
    print('   per-latent dW+ rel:', ' '.join(per))
    mz = StyleGAN2(search='z', weights=W, size=size, device='cuda')
    zr = z.clone().requires_grad_(True)
    (R.forward_z(W, zr, noises, size) * probe).sum().backward()
    zd = z.cuda().requires_grad_(True)
    (mz.forward_z(zd, noises=[n.cuda() for n in noises]) * probe.cuda()).sum().backward()
    print('   dz rel %.3e (|ref| %.3e)' % (rel(zd.grad, zr.grad), zr.grad.norm().item()))
    for Bt in (1, 8):
        zz = torch.randn(Bt, 512, device='cuda', requires_grad=True)
        for _ in range(2):
            o = mz.forward_z(zz); o.square().mean().backward()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            o = mz.forward_z(zz); o.square().mean().backward()
        torch.cuda.synchronize()
        print('   B=%d fwd+bwd %.2f ms' % (Bt, (time.time() - t) / 5 * 1e3))
