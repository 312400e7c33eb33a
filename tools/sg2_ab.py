import sys, os, warnings
sys.path.insert(0, os.getcwd())
warnings.simplefilter('ignore')
import torch
from pix2latent_amd.utils import synthetic as S
from pix2latent_amd.model.stylegan2 import StyleGAN2
from pix2latent_amd import _native as N
from oracle import stylegan2_ref as R
SIZE = 64
dev = torch.device('cuda')
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ['%.2e' % v for v in ((a - b).norm(dim=1) / b.norm(dim=1)).tolist()]
for widths, seed in ((None, 3), (None, 4), (None, 5), ({4: 128, 8: 128, 16: 64, 32: 32, 64: 32}, 3), ({4: 128, 8: 128, 16: 64, 32: 32, 64: 32}, 4)):
    W = S.stylegan2_weights(SIZE, 0, channels=widths)
    g = torch.Generator().manual_seed(seed)
    B = 3
    z = torch.randn(B, 512, generator=g)
    noises = [torch.randn(B, 1, s[2], s[3], generator=g) for s in R.noise_shapes(SIZE)]
    probe = torch.randn(B, 3, SIZE, SIZE, generator=g) / SIZE
    zr = z.clone().requires_grad_(True)
    (R.forward_z(W, zr, noises, SIZE) * probe).sum().backward()
    z64 = z.double().requires_grad_(True)
    W64 = {k: v.double() for k, v in W.items()}
    out64 = R.forward_z(W64, z64, [n.double() for n in noises], SIZE)
    (out64 * probe.double()).sum().backward()
    print('widths', 'wide' if widths is None else 'narrow', 'floor', rel(zr.grad, z64.grad))
    for wf, name in ((0, 'f32'), (1, 'bf16x3-direct'), (2, 'default')):
        m = StyleGAN2(model='cars', search='z', weights=W, size=SIZE, device=dev, wfmt=wf)
        zd = z.to(dev).requires_grad_(True)
        out = m.forward_z(zd, noises=[n.to(dev) for n in noises])
        (out * probe.to(dev)).sum().backward()
        print('  %-14s pix %.2e  dz rel %s' % (name, (out.detach().cpu().double() - out64.detach()).abs().max().item(), rel(zd.grad, z64.grad)))
