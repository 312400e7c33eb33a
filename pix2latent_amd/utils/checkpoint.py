"""Loaders that turn the upstream checkpoints into the flat weight dictionaries
the native engine packs.  No file can be downloaded in this
environment, so these are validated structurally (tests/test_checkpoint.py builds
state-dicts with the upstream key layout from seeded tensors).

* HF `pytorch_pretrained_biggan` `pytorch_model.bin` (reference
  pix2latent/model/biggan.py:26-32): every conv / linear carries spectral-norm
  parametrisation keys `weight_orig`, `weight_u`, `weight_v`; the reference bakes
  them out with `remove_spectral_norm` (utils/misc.py:150-157), i.e.
  W = weight_orig / (u^T . W_mat . v) with the stored u, v (eval mode, no power
  iteration).
* torchvision `vgg16` / `alexnet` `features.N.{weight,bias}` + lpips `lin{k}.model.1.weight`
  (reference pix2latent/loss_functions.py:131).
"""
import torch

from . import synthetic

_VGG_FEATURE_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)


def bake_spectral_norm(weight_orig, u, v):
    """weight of a torch.nn.utils.spectral_norm module in eval mode."""
    w_mat = weight_orig.reshape(weight_orig.shape[0], -1)
    sigma = torch.dot(u.reshape(-1), torch.mv(w_mat, v.reshape(-1)))
    return weight_orig / sigma


def expected_biggan_shapes(ch=synthetic.CH, z_dim=synthetic.Z_DIM, num_classes=1000):
    """{key: shape} of the flat biggan-deep-256 dictionary (spectral norm baked out)."""
    cond = 2 * z_dim
    S = {'embeddings.weight': (z_dim, num_classes),
         'generator.gen_z.weight': (16 * 16 * ch, cond), 'generator.gen_z.bias': (16 * 16 * ch,)}

    def bn(p, c):
        S[p + '.running_means'] = (synthetic.N_STATS, c)
        S[p + '.running_vars'] = (synthetic.N_STATS, c)
        S[p + '.scale.weight'] = (c, cond)
        S[p + '.offset.weight'] = (c, cond)
    for i, spec in enumerate(synthetic.layer_table(ch)):
        p = 'generator.layers.%d' % i
        if spec[0] == 'attn':
            c = spec[1]
            S[p + '.snconv1x1_theta.weight'] = (c // 8, c, 1, 1)
            S[p + '.snconv1x1_phi.weight'] = (c // 8, c, 1, 1)
            S[p + '.snconv1x1_g.weight'] = (c // 2, c, 1, 1)
            S[p + '.snconv1x1_o_conv.weight'] = (c, c // 2, 1, 1)
            S[p + '.gamma'] = (1,)
        else:
            _, up, cin, cout = spec
            mid = cin // 4
            for k, c in enumerate((cin, mid, mid, mid)):
                bn('%s.bn_%d' % (p, k), c)
            for k, (o, i_, ks) in enumerate(((mid, cin, 1), (mid, mid, 3), (mid, mid, 3), (cout, mid, 1))):
                S['%s.conv_%d.weight' % (p, k)] = (o, i_, ks, ks)
                S['%s.conv_%d.bias' % (p, k)] = (o,)
    S['generator.bn.running_means'] = (synthetic.N_STATS, ch)
    S['generator.bn.running_vars'] = (synthetic.N_STATS, ch)
    S['generator.bn.weight'] = (ch,)
    S['generator.bn.bias'] = (ch,)
    S['generator.conv_to_rgb.weight'] = (ch, ch, 3, 3)
    S['generator.conv_to_rgb.bias'] = (ch,)
    return S


def load_biggan_state_dict(sd, strict=True):
    """HF BigGAN state_dict (with or without spectral-norm keys) -> flat dict."""
    out = {}
    for k, v in sd.items():
        if k.endswith('.weight_orig'):
            base = k[:-len('_orig')]
            out[base] = bake_spectral_norm(v.float(), sd[base + '_u'].float(), sd[base + '_v'].float())
        elif k.endswith('.weight_u') or k.endswith('.weight_v'):
            continue
        else:
            out[k] = v.float() if torch.is_floating_point(v) else v
    if strict:
        exp = expected_biggan_shapes()
        missing = [k for k in exp if k not in out]
        if missing:
            raise KeyError('BigGAN checkpoint misses %d keys, e.g. %s' % (len(missing), missing[:3]))
        for k, shp in exp.items():
            if tuple(out[k].shape) != tuple(shp):
                raise ValueError('BigGAN checkpoint: %s has shape %s, expected %s'
                                 % (k, tuple(out[k].shape), shp))
    return out


def load_lpips_vgg(vgg16_sd, lpips_sd):
    """torchvision vgg16 state_dict + lpips v0.1 'vgg' linear layers -> flat dict."""
    out = {}
    for i, idx in enumerate(_VGG_FEATURE_IDX):
        w = vgg16_sd['features.%d.weight' % idx].float()
        b = vgg16_sd['features.%d.bias' % idx].float()
        cin, cout = synthetic.VGG_CONVS[i]
        if tuple(w.shape) != (cout, cin, 3, 3):
            raise ValueError('vgg16 features.%d.weight has shape %s' % (idx, tuple(w.shape)))
        out['vgg.conv%d.weight' % i] = w
        out['vgg.conv%d.bias' % i] = b
    for k, c in enumerate(synthetic.VGG_CHNS):
        key = 'lin%d.model.1.weight' % k
        w = lpips_sd[key].float()
        if tuple(w.shape) != (1, c, 1, 1):
            raise ValueError('lpips %s has shape %s' % (key, tuple(w.shape)))
        out['lpips.lin%d.weight' % k] = w
    return out


_ALEX_FEATURE_IDX = (0, 3, 6, 8, 10)     # torchvision alexnet.features conv positions


def load_lpips_alex(alexnet_sd, lpips_sd):
    """torchvision alexnet state_dict + lpips v0.1 'alex' linear layers -> flat dict
    ('alex.conv{i}.{weight,bias}', 'lpips.lin{k}.weight'), the format
    `ProjectionLoss(weights=...)` / $P2L_LPIPS_ALEX_WEIGHTS take."""
    out = {}
    for i, idx in enumerate(_ALEX_FEATURE_IDX):
        w = alexnet_sd['features.%d.weight' % idx].float()
        b = alexnet_sd['features.%d.bias' % idx].float()
        cin, cout, k = synthetic.ALEX_CONVS[i]
        if tuple(w.shape) != (cout, cin, k, k):
            raise ValueError('alexnet features.%d.weight has shape %s' % (idx, tuple(w.shape)))
        out['alex.conv%d.weight' % i] = w
        out['alex.conv%d.bias' % i] = b
    for k, c in enumerate(synthetic.ALEX_CHNS):
        key = 'lin%d.model.1.weight' % k
        w = lpips_sd[key].float()
        if tuple(w.shape) != (1, c, 1, 1):
            raise ValueError('lpips %s has shape %s' % (key, tuple(w.shape)))
        out['lpips.lin%d.weight' % k] = w
    return out


_SQZ_FIRE_IDX = (3, 4, 6, 7, 9, 10, 11, 12)     # torchvision squeezenet1_1.features Fire positions


def load_lpips_squeeze(squeezenet_sd, lpips_sd):
    """torchvision squeezenet1_1 state_dict + lpips v0.1 'squeeze' linear layers -> flat dict
    ('squeeze.conv0.*', 'squeeze.fire{i}.{squeeze,expand1x1,expand3x3}.*', 'lpips.lin{k}.weight')."""
    out = {}
    w = squeezenet_sd['features.0.weight'].float()
    if tuple(w.shape) != (64, 3, 3, 3):
        raise ValueError('squeezenet1_1 features.0.weight has shape %s' % (tuple(w.shape),))
    out['squeeze.conv0.weight'] = w
    out['squeeze.conv0.bias'] = squeezenet_sd['features.0.bias'].float()
    for i, idx in enumerate(_SQZ_FIRE_IDX):
        cin, sq, ex = synthetic.SQZ_FIRES[i]
        for part, shp in (('squeeze', (sq, cin, 1, 1)), ('expand1x1', (ex, sq, 1, 1)), ('expand3x3', (ex, sq, 3, 3))):
            w = squeezenet_sd['features.%d.%s.weight' % (idx, part)].float()
            if tuple(w.shape) != shp:
                raise ValueError('squeezenet1_1 features.%d.%s.weight has shape %s' % (idx, part, tuple(w.shape)))
            out['squeeze.fire%d.%s.weight' % (i, part)] = w
            out['squeeze.fire%d.%s.bias' % (i, part)] = squeezenet_sd['features.%d.%s.bias' % (idx, part)].float()
    for k, c in enumerate(synthetic.SQZ_CHNS):
        key = 'lin%d.model.1.weight' % k
        w = lpips_sd[key].float()
        if tuple(w.shape) != (1, c, 1, 1):
            raise ValueError('lpips %s has shape %s' % (key, tuple(w.shape)))
        out['lpips.lin%d.weight' % k] = w
    return out


def load_result(path):
    """reads what `save_variables` wrote (reference pix2latent/edit/editor.py:16-22)."""
    import numpy as np
    return np.load(path, allow_pickle=True).item()
