"""The smoke() probe (one candidate) against the fp64 oracle under different kernel switches:
is a large native distance a property of the probe (a ReLU / max-pool mask sitting at a tie) or
of one kernel path?  usage: DIAG_SEED=2 python tools/diag_smoke.py   (oracles cached in /tmp)"""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SEED = int(os.environ.get('DIAG_SEED', '2'))
B = int(os.environ.get('DIAG_B', '1'))
cache = '/tmp/diag_smoke_%d_%d.pt' % (SEED, B)


def inputs():
    from pix2latent_amd.utils import synthetic as S
    g = torch.Generator().manual_seed(SEED)
    z = torch.fmod(torch.randn(B, 128, generator=g), 2.0)
    c = (0.05 * torch.randn(1, 128, generator=g)).repeat(B, 1)
    target = S.synthetic_target(256, 1).unsqueeze(0).repeat(B, 1, 1, 1)
    weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(B, 1, 1, 1)
    return z, c, target, weight


def rel(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return ((a - b).norm() / b.norm()).item()


if len(sys.argv) > 1 and sys.argv[1] == 'native':
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.biggan import BigGAN
    import pix2latent_amd.loss_functions as LF
    import warnings; warnings.simplefilter('ignore')
    z, c, target, weight = inputs()
    dev = torch.device('cuda')
    model = BigGAN(weights=S.biggan_weights(0), device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    zd, cd = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
    loss = loss_fn(model(z=zd, c=cd), target.to(dev), weight.to(dev))
    loss.mean().backward()
    o = torch.load(cache)
    print('%-34s dz vs fp64 %.3e (vs fp32 oracle %.3e)   dc %.3e   [fp32 oracle: dz %.3e dc %.3e]' % (
        sys.argv[2], rel(zd.grad, o['dz64']), rel(zd.grad, o['dz32']), rel(cd.grad, o['dc64']),
        rel(o['dz32'], o['dz64']), rel(o['dc32'], o['dc64'])))
    sys.exit(0)

if not os.path.exists(cache):
    from pix2latent_amd.utils import synthetic as S
    from oracle import biggan_ref as R, lpips_ref as L
    W, Wv = S.biggan_weights(0), S.lpips_vgg_weights(1)
    z, c, target, weight = inputs()
    res = {}
    for dt, tag in ((torch.float64, '64'), (torch.float32, '32')):
        Wd = {k: v.to(dt) for k, v in W.items()}
        Wvd = {k: v.to(dt) for k, v in Wv.items()}
        zr, cr = z.to(dt).clone().requires_grad_(True), c.to(dt).clone().requires_grad_(True)
        L.projection_loss(Wvd, R.biggan_forward(Wd, zr, cr), target.to(dt), weight.to(dt)).mean().backward()
        res['dz' + tag], res['dc' + tag] = zr.grad, cr.grad
    torch.save(res, cache)
for name, env in (('default', {}), ('P2L_CONV_WFMT=f32', {'P2L_CONV_WFMT': 'f32'}), ('P2L_ATTN=0', {'P2L_ATTN': '0'}),
                  ('P2L_PW=0', {'P2L_PW': '0'}), ('P2L_THIN=0', {'P2L_THIN': '0'}), ('direct 3x3 kernels', {'P2L_CONV_WFMT': 'bf16x3-direct'}),
                  ('all bf16x3 forms off', {'P2L_ATTN': '0', 'P2L_PW': '0', 'P2L_THIN': '0', 'P2L_CONV_WFMT': 'bf16x3-direct'})):
    e = dict(os.environ); e.update(env)
    subprocess.call([sys.executable, os.path.abspath(__file__), 'native', name], env=e)
