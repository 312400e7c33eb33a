"""Seeded synthetic weights / targets.

There is no network in the build or bench environment, so neither the
HuggingFace `biggan-deep-256` checkpoint (reference pix2latent/model/biggan.py:26)
nor the lpips / torchvision VGG16 weights (pix2latent/loss_functions.py:131) can
be fetched.  These generators produce random-init tensors of exactly those
architectures, keyed like the upstream state_dicts (after the spectral-norm
bake-out of utils/misc.py:150-157) so that a real checkpoint can be dropped in
through the same dictionary.
"""
import math

import torch

CH = 128
Z_DIM = 128
N_STATS = 51
# (up_sample, in_mult, out_mult) of biggan-deep-256
LAYERS = [(False, 16, 16), (True, 16, 16), (False, 16, 16), (True, 16, 8),
          (False, 8, 8), (True, 8, 8), (False, 8, 8), (True, 8, 4),
          (False, 4, 4), (True, 4, 2), (False, 2, 2), (True, 2, 1)]
ATTN_POS = 8

VGG_CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256),
             (256, 256), (256, 512), (512, 512), (512, 512), (512, 512),
             (512, 512), (512, 512)]
VGG_CHNS = (64, 128, 256, 512, 512)


def layer_table(ch=CH, layers=LAYERS, attn_pos=ATTN_POS):
    out = []
    for i, (up, cin, cout) in enumerate(layers):
        if i == attn_pos:
            out.append(('attn', ch * cin))
        out.append(('block', up, ch * cin, ch * cout))
    return out


def biggan_weights(seed=0, ch=CH, layers=LAYERS, attn_pos=ATTN_POS, z_dim=Z_DIM,
                   num_classes=1000):
    g = torch.Generator().manual_seed(seed)
    cond_dim = 2 * z_dim

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    W = {}
    W['embeddings.weight'] = randn(z_dim, num_classes, std=0.05)
    W['generator.gen_z.weight'] = randn(4 * 4 * 16 * ch, cond_dim, std=math.sqrt(1.0 / cond_dim))
    W['generator.gen_z.bias'] = randn(4 * 4 * 16 * ch, std=0.01)

    def bn(prefix, c, conditional=True):
        W[prefix + '.running_means'] = randn(N_STATS, c, std=0.1)
        W[prefix + '.running_vars'] = 0.5 + torch.rand(N_STATS, c, generator=g)
        if conditional:
            W[prefix + '.scale.weight'] = randn(c, cond_dim, std=0.02)
            W[prefix + '.offset.weight'] = randn(c, cond_dim, std=0.02)
        else:
            W[prefix + '.weight'] = 0.5 + 0.5 * torch.rand(c, generator=g)
            W[prefix + '.bias'] = randn(c, std=0.05)

    def conv(prefix, cout, cin, k, bias=True, gain=2.0):
        fan_in = cin * k * k
        W[prefix + '.weight'] = randn(cout, cin, k, k, std=math.sqrt(gain / fan_in))
        if bias:
            W[prefix + '.bias'] = randn(cout, std=0.01)

    for i, spec in enumerate(layer_table(ch, layers, attn_pos)):
        p = 'generator.layers.%d' % i
        if spec[0] == 'attn':
            c = spec[1]
            conv(p + '.snconv1x1_theta', c // 8, c, 1, bias=False, gain=1.0)
            conv(p + '.snconv1x1_phi', c // 8, c, 1, bias=False, gain=1.0)
            conv(p + '.snconv1x1_g', c // 2, c, 1, bias=False, gain=1.0)
            conv(p + '.snconv1x1_o_conv', c, c // 2, 1, bias=False, gain=1.0)
            W[p + '.gamma'] = torch.tensor([0.5])
        else:
            _, up, cin, cout = spec
            mid = cin // 4
            bn(p + '.bn_0', cin)
            conv(p + '.conv_0', mid, cin, 1)
            bn(p + '.bn_1', mid)
            conv(p + '.conv_1', mid, mid, 3)
            bn(p + '.bn_2', mid)
            conv(p + '.conv_2', mid, mid, 3)
            bn(p + '.bn_3', mid)
            conv(p + '.conv_3', cout, mid, 1, gain=0.5)
    bn('generator.bn', ch, conditional=False)
    conv('generator.conv_to_rgb', ch, ch, 3, gain=0.05)
    return W


def heavy_tailed(W, seed=7, frac=0.02, alpha=64.0, spike=1024.0):
    """A heavy-tailed variant of a BigGAN-deep state dict (returns a new dict).

    `biggan_weights` draws every tensor from a well-conditioned Gaussian: activations are near
    Gaussian, max / typical ~ 5.  Trained generators are not like that: BN statistics span orders
    of magnitude and a few channels carry activations 10^2 - 10^3 x the median.  The block-
    floating-point arithmetic of the fp16 x 2 kernels (per-image power-of-two scales, maxima
    handed over as BOUNDS) has to be shown on such data.  The network is
    re-parametrised so that it computes (nearly) the same function -- the activations stay sane
    through 50 layers -- while the tensors the kernels see become heavy-tailed; for every
    conv_k -> bn_{k+1} -> ReLU -> conv_{k+1} chain inside a GenBlock (k = 0, 1, 2):

      * RAW outliers: `frac` of conv_k's output channels x alpha (weight rows, bias), undone by
        bn_{k+1}'s statistics (means x alpha, variances x alpha^2): the stored tensor has outlier
        channels, the fused prologue's scales span alpha;
      * small-variance channels: another `frac` / alpha (variances down to ~1e-3 x the usual);
      * ACTIVATION outliers: another `frac` of bn_{k+1}'s channels get variance / alpha^2 and
        offset x alpha -- the post-ReLU activation is alpha x larger -- undone by conv_{k+1}'s
        input-channel weights / alpha; ONE channel per layer gets `spike` instead of alpha
        ("a few 10^3 outlier activations per image").
    """
    g = torch.Generator().manual_seed(seed)
    W = {k: v.clone() for k, v in W.items()}
    blocks = sorted({k.rsplit('.bn_0.', 1)[0] for k in W if '.bn_0.running_means' in k})
    for p in blocks:
        for k in (0, 1, 2):
            cw, cb = p + '.conv_%d.weight' % k, p + '.conv_%d.bias' % k
            bn = p + '.bn_%d' % (k + 1)
            nw = p + '.conv_%d.weight' % (k + 1)
            C = W[cw].shape[0]
            n = max(1, int(round(frac * C)))
            perm = torch.randperm(C, generator=g)
            raw_hi, raw_lo, act = perm[:n], perm[n:2 * n], perm[2 * n:3 * n]
            for idx, a in ((raw_hi, alpha), (raw_lo, 1.0 / alpha)):
                W[cw][idx] *= a
                W[cb][idx] *= a
                W[bn + '.running_means'][:, idx] *= a
                W[bn + '.running_vars'][:, idx] *= a * a
            a_act = torch.full((len(act),), alpha)
            a_act[0] = spike
            W[bn + '.running_vars'][:, act] /= (a_act * a_act)
            W[bn + '.offset.weight'][act] *= a_act[:, None]
            W[nw][:, act] /= a_act[None, :, None, None]
    return W


def heavy_tailed_vgg(Wv, seed=8, frac=0.02, alpha=32.0):
    """heavy-tailed variant of the LPIPS-VGG16 weights: `frac` of every conv's output channels
    x alpha (trained VGG features have such dominant channels), undone by the next conv's
    input-channel weights where there is one (the LPIPS taps see the outlier channels)."""
    g = torch.Generator().manual_seed(seed)
    Wv = {k: v.clone() for k, v in Wv.items()}
    for i in range(len(VGG_CONVS)):
        C = Wv['vgg.conv%d.weight' % i].shape[0]
        idx = torch.randperm(C, generator=g)[:max(1, int(round(frac * C)))]
        Wv['vgg.conv%d.weight' % i][idx] *= alpha
        Wv['vgg.conv%d.bias' % i][idx] *= alpha
        if i + 1 < len(VGG_CONVS):
            Wv['vgg.conv%d.weight' % (i + 1)][:, idx] /= alpha
    return Wv


def lpips_vgg_weights(seed=1):
    g = torch.Generator().manual_seed(seed)
    Wv = {}
    for i, (cin, cout) in enumerate(VGG_CONVS):
        Wv['vgg.conv%d.weight' % i] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        Wv['vgg.conv%d.bias' % i] = torch.randn(cout, generator=g) * 0.01
    for k, c in enumerate(VGG_CHNS):
        # trained lpips lin weights are non-negative; keep that property
        Wv['lpips.lin%d.weight' % k] = (torch.rand(1, c, 1, 1, generator=g) / c)
    return Wv


ALEX_CONVS = [(3, 64, 11), (64, 192, 5), (192, 384, 3), (384, 256, 3), (256, 256, 3)]
ALEX_CHNS = (64, 192, 384, 256, 256)


def lpips_alex_weights(seed=2):
    """seeded random-init torchvision-AlexNet features + lpips lin layers
    (keys 'alex.conv{0..4}.{weight,bias}', 'lpips.lin{0..4}.weight')."""
    g = torch.Generator().manual_seed(seed)
    Wa = {}
    for i, (cin, cout, k) in enumerate(ALEX_CONVS):
        Wa['alex.conv%d.weight' % i] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cin * k * k))
        Wa['alex.conv%d.bias' % i] = torch.randn(cout, generator=g) * 0.01
    for k, c in enumerate(ALEX_CHNS):
        Wa['lpips.lin%d.weight' % k] = (torch.rand(1, c, 1, 1, generator=g) / c)
    return Wa


# torchvision squeezenet1_1.features as lpips slices it (oracle/lpips_ref.py SQZ_*): (in, squeeze, expand)
SQZ_FIRES = [(64, 16, 64), (128, 16, 64), (128, 32, 128), (256, 32, 128), (256, 48, 192), (384, 48, 192),
             (384, 64, 256), (512, 64, 256)]
SQZ_CHNS = (64, 128, 256, 384, 384, 512, 512)


def lpips_squeeze_weights(seed=3):
    """seeded random-init torchvision-SqueezeNet1.1 features + lpips lin layers (keys 'squeeze.conv0.*',
    'squeeze.fire{0..7}.{squeeze,expand1x1,expand3x3}.{weight,bias}', 'lpips.lin{0..6}.weight')."""
    g = torch.Generator().manual_seed(seed)
    Ws = {}

    def conv(name, cout, cin, k):
        Ws[name + '.weight'] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cin * k * k))
        Ws[name + '.bias'] = torch.randn(cout, generator=g) * 0.01
    conv('squeeze.conv0', 64, 3, 3)
    for i, (cin, sq, ex) in enumerate(SQZ_FIRES):
        conv('squeeze.fire%d.squeeze' % i, sq, cin, 1)
        conv('squeeze.fire%d.expand1x1' % i, ex, sq, 1)
        conv('squeeze.fire%d.expand3x3' % i, ex, sq, 3)
    for k, c in enumerate(SQZ_CHNS):
        Ws['lpips.lin%d.weight' % k] = (torch.rand(1, c, 1, 1, generator=g) / c)
    return Ws


def synthetic_target(size=256, seed=1):
    """smooth image in [-1,1] + a little noise, [3,size,size]."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing='ij')
    img = torch.zeros(3, size, size)
    for c in range(3):
        for _ in range(6):
            a = torch.randn(1, generator=g).item()
            fx, fy = (torch.rand(2, generator=g) * 4).tolist()
            ph = torch.rand(1, generator=g).item() * 2 * math.pi
            img[c] += a * torch.sin(2 * math.pi * (fx * xs + fy * ys) + ph)
    img = torch.tanh(img) + 0.05 * torch.randn(3, size, size, generator=g)
    return img.clamp(-1, 1)


def synthetic_weight_mask(size=256):
    """centred ellipse in {-1,1} -> ((m+1)/2).clamp(0.3,1) as
    examples/invert_biggan_adam.py:49 does with the user mask."""
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing='ij')
    m = ((xs / 0.7) ** 2 + (ys / 0.6) ** 2 <= 1).float() * 2 - 1
    w = ((m + 1) / 2).clamp(0.3, 1.0)
    return w.unsqueeze(0).repeat(3, 1, 1)


# ---------------------------------------------------------------------------
# StyleGAN2 (rosinality g_ema key layout; reference pix2latent/model/stylegan2.py:84-85)
# ---------------------------------------------------------------------------
SG2_STYLE_DIM = 512
SG2_N_MLP = 8
SG2_LR_MLP = 0.01


def sg2_channels(channel_multiplier=2):
    cm = channel_multiplier
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm,
            512: 32 * cm, 1024: 16 * cm}


def stylegan2_weights(size=512, seed=0, channel_multiplier=2, channels=None):
    """seeded random-init `g_ema` state-dict of Generator(size, 512, 8, channel_multiplier).
    `channels` ({resolution: width}) overrides the width table, so that tests can put the
    narrow 64/32-channel layers of the 512^2 / 1024^2 models into a small network."""
    g = torch.Generator().manual_seed(seed)
    ch = dict(sg2_channels(channel_multiplier))
    if channels:
        ch.update(channels)
    W = {}

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    for i in range(1, SG2_N_MLP + 1):
        W['style.%d.weight' % i] = randn(SG2_STYLE_DIM, SG2_STYLE_DIM) / SG2_LR_MLP
        W['style.%d.bias' % i] = randn(SG2_STYLE_DIM, std=0.1) / SG2_LR_MLP * SG2_LR_MLP
    W['input.input'] = randn(1, ch[4], 4, 4)

    def modconv(p, cin, cout, k, wstd=1.0):
        W[p + '.weight'] = randn(1, cout, cin, k, k, std=wstd)
        W[p + '.modulation.weight'] = randn(cin, SG2_STYLE_DIM)
        W[p + '.modulation.bias'] = torch.ones(cin) + randn(cin, std=0.1)

    def styled(p, cin, cout):
        modconv(p + '.conv', cin, cout, 3)
        W[p + '.noise.weight'] = randn(1, std=0.1)
        W[p + '.activate.bias'] = randn(cout, std=0.1)

    def torgb(p, cin):
        modconv(p + '.conv', cin, 3, 1, wstd=0.15)     # keeps the image inside (-1, 1)
        W[p + '.bias'] = randn(1, 3, 1, 1, std=0.05)

    styled('conv1', ch[4], ch[4])
    torgb('to_rgb1', ch[4])
    cin = ch[4]
    log_size = int(math.log2(size))
    for j, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        styled('convs.%d' % (2 * j), cin, cout)
        styled('convs.%d' % (2 * j + 1), cout, cout)
        torgb('to_rgbs.%d' % j, cout)
        cin = cout
    for i in range((log_size - 2) * 2 + 1):
        r = 2 ** ((i + 5) // 2)
        W['noises.noise_%d' % i] = randn(1, 1, r, r)
    return W
