"""diagnose gradient parity: generator backward alone (linear probe loss), loss
backward alone (oracle image in), error statistics instead of max-norm."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(32)
from pix2latent_amd.utils import synthetic as S
from pix2latent_amd.model.biggan import BigGAN
import pix2latent_amd.loss_functions as LF
from oracle import biggan_ref as R, lpips_ref as L

dev = torch.device('cuda')
W, Wv = S.biggan_weights(0), S.lpips_vgg_weights(1)
model = BigGAN(weights=W)
loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv)
g = torch.Generator().manual_seed(2)
B = 2
z = torch.fmod(torch.randn(B, 128, generator=g), 2.0)
c = (0.05 * torch.randn(1, 128, generator=g)).repeat(B, 1)
target = S.synthetic_target(256, 1).unsqueeze(0).repeat(B, 1, 1, 1)
weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(B, 1, 1, 1)
probe = torch.randn(B, 3, 256, 256, generator=g) / 256.0


def stats(name, got, ref):
    got, ref = got.detach().cpu().double().flatten(), ref.detach().double().flatten()
    d = (got - ref).abs()
    rl2 = (d.norm() / ref.norm()).item()
    mx = (d.max() / ref.abs().max()).item()
    q = np.quantile((d / ref.abs().max()).numpy(), [0.5, 0.9, 0.99, 0.999])
    cos = (got @ ref / got.norm() / ref.norm()).item()
    sign = (torch.sign(got) != torch.sign(ref)).double().mean().item()
    print('%-28s relL2=%.3e max/max=%.3e q50/90/99/99.9=%s cos=%.8f signflip=%.4f' % (name, rl2, mx, np.array2string(q, precision=2), cos, sign), flush=True)


# --- A. generator backward with a linear probe (no ReLU/argmax in the loss)
t0 = time.time()
zr, cr = z.clone().requires_grad_(True), c.clone().requires_grad_(True)
out_r, inter = R.biggan_forward(W, zr, cr, return_intermediates=True)
(out_r * probe).sum().backward()
print('oracle gen fwd+bwd %.1fs' % (time.time() - t0), flush=True)
zd, cd = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
out = model(z=zd, c=cd)
(out * probe.to(dev)).sum().backward()
stats('A gen: out', out, out_r)
stats('A gen: dz', zd.grad, zr.grad)
stats('A gen: dc', cd.grad, cr.grad)

# --- B. loss backward alone on the oracle image
t0 = time.time()
o_r = out_r.detach().clone().requires_grad_(True)
l_r = L.projection_loss(Wv, o_r, target, weight)
l_r.mean().backward()
print('oracle loss fwd+bwd %.1fs' % (time.time() - t0), flush=True)
o_d = out_r.detach().to(dev).requires_grad_(True)
l_d = loss_fn(o_d, target.to(dev), weight.to(dev))
l_d.mean().backward()
stats('B loss: value', l_d, l_r)
stats('B loss: dout', o_d.grad, o_r.grad)
# L1 part only
o_r2 = out_r.detach().clone().requires_grad_(True)
L.reconstruction_loss(o_r2, target, weight).mean().backward()
o_d2 = out_r.detach().to(dev).requires_grad_(True)
LF.ReconstructionLoss()(o_d2, target.to(dev), weight.to(dev)).mean().backward()
stats('B loss: dout (L1 only)', o_d2.grad, o_r2.grad)
lp_r = o_r.grad - o_r2.grad
lp_d = o_d.grad.cpu() - o_d2.grad.cpu()
stats('B loss: dout (LPIPS part)', lp_d, lp_r)

# --- C. full chain
zr2, cr2 = z.clone().requires_grad_(True), c.clone().requires_grad_(True)
L.projection_loss(Wv, R.biggan_forward(W, zr2, cr2), target, weight).mean().backward()
zd2, cd2 = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
loss_fn(model(z=zd2, c=cd2), target.to(dev), weight.to(dev)).mean().backward()
stats('C full: dz', zd2.grad, zr2.grad)
stats('C full: dc', cd2.grad, cr2.grad)
# feed the ORACLE's dout through the native generator backward
zd3, cd3 = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
out3 = model(z=zd3, c=cd3)
out3.backward(o_r.grad.to(dev))
stats('C gen bwd(oracle dout): dz', zd3.grad, zr2.grad)
stats('C gen bwd(oracle dout): dc', cd3.grad, cr2.grad)
print('|dz| quantiles (ref):', np.quantile(zr2.grad.abs().numpy(), [0.1, 0.5, 0.9, 1.0]))
