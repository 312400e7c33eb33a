"""Structural checks of the checkpoint loaders (pix2latent_amd/utils/checkpoint.py)
with state-dicts that use the upstream key layout (no real files exist here)."""
import numpy as np
import pytest
import torch


def test_biggan_loader_bakes_spectral_norm():
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.utils.checkpoint import (load_biggan_state_dict, expected_biggan_shapes,
                                                 bake_spectral_norm)
    W = S.biggan_weights(3)
    assert {k: tuple(v.shape) for k, v in W.items()} == expected_biggan_shapes()
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, v in W.items():
        if k.endswith('.weight') and v.dim() >= 2 and 'embeddings' not in k:
            u = torch.randn(v.shape[0], generator=g)
            vv = torch.randn(v[0].numel(), generator=g)
            w_orig = 3.7 * v                                         # arbitrary overall scale
            sigma = torch.dot(u, torch.mv(w_orig.reshape(v.shape[0], -1), vv))
            # choose u so that weight_orig / sigma == v exactly
            u = u * (3.7 / sigma)
            sd[k + '_orig'], sd[k + '_u'], sd[k + '_v'] = w_orig, u, vv
        else:
            sd[k] = v
    out = load_biggan_state_dict(sd)
    assert set(out) == set(W)
    for k in W:
        assert torch.allclose(out[k], W[k], rtol=1e-4, atol=1e-6), k
    # reference semantics of the bake: weight_orig / (u^T W v)
    w = torch.randn(5, 3, 3, 3, generator=g)
    u, v = torch.randn(5, generator=g), torch.randn(27, generator=g)
    sig = u @ (w.reshape(5, -1) @ v)
    assert torch.allclose(bake_spectral_norm(w, u, v), w / sig)
    bad = dict(sd)
    del bad['generator.gen_z.bias']
    with pytest.raises(KeyError):
        load_biggan_state_dict(bad)


def test_lpips_vgg_loader_key_mapping():
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.utils.checkpoint import load_lpips_vgg
    Wv = S.lpips_vgg_weights(5)
    idx = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
    vgg = {}
    for i, n in enumerate(idx):
        vgg['features.%d.weight' % n] = Wv['vgg.conv%d.weight' % i]
        vgg['features.%d.bias' % n] = Wv['vgg.conv%d.bias' % i]
    lp = {'lin%d.model.1.weight' % k: Wv['lpips.lin%d.weight' % k] for k in range(5)}
    out = load_lpips_vgg(vgg, lp)
    assert set(out) == set(Wv)
    for k in Wv:
        assert torch.equal(out[k], Wv[k])


def test_lpips_alex_loader_key_mapping():
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.utils.checkpoint import load_lpips_alex
    Wa = S.lpips_alex_weights(5)
    alex = {}
    for i, n in enumerate((0, 3, 6, 8, 10)):
        alex['features.%d.weight' % n] = Wa['alex.conv%d.weight' % i]
        alex['features.%d.bias' % n] = Wa['alex.conv%d.bias' % i]
    lp = {'lin%d.model.1.weight' % k: Wa['lpips.lin%d.weight' % k] for k in range(5)}
    out = load_lpips_alex(alex, lp)
    assert set(out) == set(Wa)
    for k in Wa:
        assert torch.equal(out[k], Wa[k])
    bad = dict(alex)
    bad['features.3.weight'] = torch.zeros(192, 64, 3, 3)
    import pytest
    with pytest.raises(ValueError):
        load_lpips_alex(bad, lp)


def test_save_variables_roundtrip(tmp_path):
    """result file layout read by the reference's editor (vars.input.z.data[i])"""
    from pix2latent_amd import VariableManager, save_variables
    from pix2latent_amd.utils.checkpoint import load_result
    vm = VariableManager(device='cpu')
    vm.register('z', (4,), 'input')
    vm.register('target', (3, 2, 2), 'output', requires_grad=False, default=torch.zeros(3, 2, 2))
    v = vm.initialize(3)
    v['loss'] = [[10, {'loss': np.array([0.1, 0.2, 0.3])}]]
    path = str(tmp_path / 'vars.npy')
    save_variables(path, v)
    r = load_result(path)
    assert len(r['input']['z']['data']) == 3
    assert torch.equal(r['input']['z']['data'][1], v.input.z.data[1].detach())
    assert r['loss'][-1][1]['loss'][2] == 0.3


def test_load_lpips_squeeze_maps_torchvision_and_lpips_keys():
    """torchvision squeezenet1_1 + lpips v0.1 'squeeze' files -> the flat dict ProjectionLoss(weights=...) takes
    (the reference downloads both inside lpips.LPIPS(net='squeeze'), loss_functions.py:131)"""
    import torch
    from pix2latent_amd.utils import checkpoint as CK, synthetic as S
    Ws = S.lpips_squeeze_weights(5)
    sd, lin = {}, {}
    sd['features.0.weight'], sd['features.0.bias'] = Ws['squeeze.conv0.weight'], Ws['squeeze.conv0.bias']
    for i, idx in enumerate((3, 4, 6, 7, 9, 10, 11, 12)):
        for part in ('squeeze', 'expand1x1', 'expand3x3'):
            for what in ('weight', 'bias'):
                sd['features.%d.%s.%s' % (idx, part, what)] = Ws['squeeze.fire%d.%s.%s' % (i, part, what)]
    for k in range(7):
        lin['lin%d.model.1.weight' % k] = Ws['lpips.lin%d.weight' % k]
    got = CK.load_lpips_squeeze(sd, lin)
    assert sorted(got) == sorted(Ws)
    assert all(torch.equal(got[k], Ws[k].float()) for k in Ws)
    bad = dict(sd)
    bad['features.6.squeeze.weight'] = torch.zeros(16, 128, 1, 1)
    import pytest
    with pytest.raises(ValueError, match='features.6.squeeze'):
        CK.load_lpips_squeeze(bad, lin)
