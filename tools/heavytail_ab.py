"""the heavy-tailed-weights gradient check of tests/test_heavytail_gpu.py for ONE library build
(P2L_LIB_PATH, tools/ab_build.sh): native dz / dc against the fp64 oracle with the native run's
decisions replayed, per candidate.  For bisecting an arithmetic difference by kernel family."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pix2latent_amd.utils import synthetic as S
from pix2latent_amd.model.biggan import BigGAN
import pix2latent_amd.loss_functions as LF
from oracle import biggan_ref as R, lpips_ref as L
from oracle.masks import DecisionTape
from oracle.replay import native_decisions
dev = torch.device('cuda')
W = S.heavy_tailed(S.biggan_weights(0)); Wv = S.heavy_tailed_vgg(S.lpips_vgg_weights(1))
n = int(os.environ.get('N', 2))
g = torch.Generator().manual_seed(int(os.environ.get('SEED', 2)))
z = torch.fmod(torch.randn(n, 128, generator=g), 2.0)
c = (0.05 * torch.randn(1, 128, generator=g)).repeat(n, 1)
target = S.synthetic_target(256, 1).unsqueeze(0).repeat(n, 1, 1, 1)
weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(n, 1, 1, 1)
model = BigGAN(weights=W, device=dev)
loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
zd, cd = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
out = model(z=zd, c=cd)
loss = loss_fn(out, target.to(dev), weight.to(dev))
loss.mean().backward()
torch.cuda.synchronize()
diag = [] if os.environ.get('DIAG') else None
items = native_decisions(model, loss_fn, W, n, dev, out, target.to(dev), diag)
if diag:
    for nm, a, b_, amb in diag:
        if any(a) or any(b_) or any(amb): print('  ambiguous', nm, a, b_, amb)
if os.environ.get('LOOSE'):
    # how loose the hand-over bound max|s| max|x| + max|t| is against max|x*s + t|, per image and layer
    d = model._desc
    s_all, t_all = model.saved_activation(2).view(n, -1), model.saved_activation(3).view(n, -1)
    prev = model.saved_activation(1); bi = 0
    for i, spec in enumerate(R.layer_table()):
        if spec[0] != 'attn':
            for k in range(4):
                x = prev if k == 0 else model.saved_activation(7, 3 * bi + (k - 1))
                cch = x.shape[-1]; off = d.blocks[bi].cbn_off[k]
                sv, tv = s_all[:, off:off + cch].view(n, 1, 1, cch), t_all[:, off:off + cch].view(n, 1, 1, cch)
                act = (x * sv + tv).abs().flatten(1).amax(1)
                med = (x * sv + tv).abs().flatten(1).median(1).values
                bound = sv.abs().flatten(1).amax(1) * x.abs().flatten(1).amax(1) + tv.abs().flatten(1).amax(1)
                print('  layer %2d bn_%d %4dx%-4d bound/actual %s  actual/median %s' % (i, k, x.shape[1], cch,
                      ['%.0f' % v for v in (bound / act).tolist()], ['%.0f' % v for v in (act / med).tolist()]))
            bi += 1
        prev = model.saved_activation(0, i)
dt = torch.float64
zr = z.to(dt).clone().requires_grad_(True); cr = c.to(dt).clone().requires_grad_(True)
tape = DecisionTape(replay=items)
o = R.biggan_forward({k: v.to(dt) for k, v in W.items()}, zr, cr, tape=tape)
l = L.projection_loss({k: v.to(dt) for k, v in Wv.items()}, o, target.to(dt), weight.to(dt), tape=tape)
l.mean().backward()
rel = lambda a, b: ((a - b).norm(dim=1) / b.norm(dim=1)).tolist()
print(os.environ.get('P2L_LIB_PATH', 'default'), 'dz', ['%.2e' % v for v in rel(zd.grad.cpu().double(), zr.grad)],
      'dc', ['%.2e' % v for v in rel(cd.grad.cpu().double(), cr.grad)],
      'dpix %.2e' % (out.detach().cpu().double() - o.detach()).abs().max().item())
