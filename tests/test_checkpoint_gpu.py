"""SURVEY 8f n3 end to end: checkpoints in the UPSTREAM on-disk formats -> loaders -> HIP models
-> CPU oracle on the loaded weights.

  * HF pytorch_pretrained_biggan `pytorch_model.bin`: spectral-norm keys weight_orig / weight_u /
    weight_v (baked out by the reference, utils/misc.py:150-157), BN running_means / running_vars
    [51, C], 1000 x 128 embedding;
  * torchvision vgg16 `features.N.{weight,bias}` + lpips v0.1 `lin{k}.model.1.weight`
    (reference loss_functions.py:131);
  * rosinality stylegan2-pytorch checkpoint `{'g_ema': state_dict}` (reference
    model/stylegan2.py:84-85).
No real file exists in this environment: the files are written here from seeded tensors with the
upstream key layout and read back through the same environment variables a user would set."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hf_biggan_file(path, seed):
    """synthetic weights re-expressed as a spectral-norm parametrised HF state_dict"""
    from pix2latent_amd.utils import synthetic as S
    W = S.biggan_weights(seed)
    g = torch.Generator().manual_seed(seed + 100)
    sd = {}
    for k, v in W.items():
        if k.endswith('.weight') and v.dim() >= 2 and 'embeddings' not in k:
            u = torch.randn(v.shape[0], generator=g)
            vv = torch.randn(v[0].numel(), generator=g)
            scale = float(1.5 + torch.rand(1, generator=g))
            w_orig = scale * v
            sigma = torch.dot(u, torch.mv(w_orig.reshape(v.shape[0], -1), vv))
            sd[k + '_orig'], sd[k + '_u'], sd[k + '_v'] = w_orig, u * (scale / sigma), vv
        else:
            sd[k] = v
    torch.save(sd, path)
    return sd


def test_hf_biggan_checkpoint_to_pixels(dev, tmp_path, monkeypatch):
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.utils.checkpoint import load_biggan_state_dict
    from oracle import biggan_ref as R
    path = str(tmp_path / 'pytorch_model.bin')
    sd = _hf_biggan_file(path, 4)
    assert any(k.endswith('weight_orig') for k in sd) and sd['generator.bn.running_means'].shape[0] == 51
    monkeypatch.setenv('P2L_BIGGAN_WEIGHTS', path)
    model = BigGAN(device=dev)                       # <- reads the upstream file
    g = torch.Generator().manual_seed(8)
    z = torch.fmod(torch.randn(2, 128, generator=g), 2.0)
    c = 0.05 * torch.randn(2, 128, generator=g)
    with torch.no_grad():
        out = model(z=z.to(dev), c=c.to(dev)).cpu()
        ref = R.biggan_forward(load_biggan_state_dict(sd), z, c)
    assert out.shape == ref.shape == (2, 3, 256, 256)
    assert (out - ref).abs().max().item() < 1e-3     # north_star: per-pixel |delta| < 1e-3


def test_torchvision_and_lpips_files_to_loss(dev, tmp_path, monkeypatch):
    import pix2latent_amd.loss_functions as LF
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.utils.checkpoint import load_lpips_vgg
    from oracle import lpips_ref as L
    Wv = S.lpips_vgg_weights(6)
    idx = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
    vgg = {}
    for i, n in enumerate(idx):
        vgg['features.%d.weight' % n] = Wv['vgg.conv%d.weight' % i]
        vgg['features.%d.bias' % n] = Wv['vgg.conv%d.bias' % i]
    vgg['classifier.0.weight'] = torch.zeros(8, 8)            # (ignored, as in the real file)
    lin = {'lin%d.model.1.weight' % k: Wv['lpips.lin%d.weight' % k] for k in range(5)}
    a, b = str(tmp_path / 'vgg16-397923af.pth'), str(tmp_path / 'vgg.pth')
    torch.save(vgg, a)
    torch.save(lin, b)
    monkeypatch.setenv('P2L_LPIPS_VGG_WEIGHTS', a + ',' + b)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', device=dev)  # <- reads the two upstream files
    g = torch.Generator().manual_seed(2)
    out = torch.tanh(torch.randn(2, 3, 256, 256, generator=g))
    target = S.synthetic_target(256, 1).unsqueeze(0).repeat(2, 1, 1, 1)
    weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(2, 1, 1, 1)
    loss = loss_fn(out.to(dev), target.to(dev), weight.to(dev)).cpu()
    ref = L.projection_loss(load_lpips_vgg(vgg, lin), out, target, weight)
    assert np.allclose(loss.numpy(), ref.numpy(), atol=1e-3, rtol=0)   # north_star: loss parity 1e-3
    assert np.allclose(loss.numpy(), ref.numpy(), rtol=1e-4)


def test_rosinality_checkpoint_to_pixels(dev, tmp_path, monkeypatch):
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    from pix2latent_amd.utils import synthetic as S
    from oracle import stylegan2_ref as R
    size = 64
    W = S.stylegan2_weights(size, 5)
    path = str(tmp_path / 'stylegan2-car-config-f.pt')
    torch.save({'g_ema': W, 'g': {}, 'd': {}, 'latent_avg': torch.zeros(512)}, path)   # rosinality layout
    monkeypatch.setenv('P2L_STYLEGAN2_CARS_WEIGHTS', path)
    model = StyleGAN2(model='cars', search='z', size=size, device=dev)   # <- reads the upstream file
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 512, generator=g)
    noises = [torch.randn(2, 1, s[2], s[3], generator=g) for s in R.noise_shapes(size)]
    with torch.no_grad():
        out = model.forward_z(z.to(dev), noises=[n.to(dev) for n in noises]).cpu()
        ref = R.forward_z(W, z, noises, size)
    assert (out - ref).abs().max().item() < 1e-3
