"""ORACLE (test infrastructure, not product code): CPU restatement of the
BigGAN-deep-256 generator that `pix2latent.model.BigGAN.forward` drives.

PARITY UNPINNED for pixels: the arithmetic lives in the un-vendored third-party
package `pytorch_pretrained_biggan>=0.1.1` (reference requirements.txt:7; call
sites pix2latent/model/biggan.py:10,26-28,47,58).  Neither the package nor its
pretrained weights exist in this environment, and the reference holds no test
or golden vector for it, so this file restates the published architecture
(HuggingFace `pytorch_pretrained_biggan/model.py`, config `biggan-deep-256`)
from recall, cf. SURVEY.md Appendix A.  What IS pinned is the call contract of
the wrapper (biggan.py:50-58: `generator(cat(z, c), truncation)`, asserts on the
shapes) and the spectral-norm bake-out (utils/misc.py:150-157, i.e. plain
weights at inference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Plain PyTorch-CPU fp32 functional ops, NCHW, no custom kernels.
"""
import math

import torch
import torch.nn.functional as F

from .masks import relu as _relu, maxpool2 as _maxpool2      # (`tape`: oracle/masks.py; None = plain ops)

# biggan-deep-256 config [3P-recall: config.py BigGANConfig for 'biggan-deep-256']
CH = 128
Z_DIM = 128
N_STATS = 51
BN_EPS = 1e-4
ATTN_POS = 8
# (up_sample, in_mult, out_mult)
LAYERS = [(False, 16, 16), (True, 16, 16), (False, 16, 16), (True, 16, 8),
          (False, 8, 8), (True, 8, 8), (False, 8, 8), (True, 8, 4),
          (False, 4, 4), (True, 4, 2), (False, 2, 2), (True, 2, 1)]


def layer_table(ch=CH, layers=LAYERS, attn_pos=ATTN_POS):
    """ModuleList order of Generator.layers: SelfAttn is inserted BEFORE the
    GenBlock with index `attn_pos` [3P-recall: Generator.__init__]."""
    out = []
    for i, (up, cin, cout) in enumerate(layers):
        if i == attn_pos:
            out.append(('attn', ch * cin))
        out.append(('block', up, ch * cin, ch * cout))
    return out


def _bn_stats(means, vars_, truncation, n_stats=N_STATS):
    """BigGANBatchNorm statistics row selection [3P-recall: BigGANBatchNorm.forward]."""
    step = 1.0 / (n_stats - 1)
    coef, start = math.modf(truncation / step)
    start = int(start)
    if coef != 0.0:
        mean = means[start] * coef + means[start + 1] * (1 - coef)
        var = vars_[start] * coef + vars_[start + 1] * (1 - coef)
    else:
        mean, var = means[start], vars_[start]
    return mean, var


def cbn(W, prefix, x, cond, truncation, taps=None):
    """Conditional BN: (x-mu)/sqrt(var+eps) * (1 + scale(cond)) + offset(cond).
    `taps` (dict): records this layer's per-(sample, channel) gain and bias tensors with
    their gradients retained -- the tests compare d loss / d gain, d loss / d bias of each of
    the 48 layers, which localises a wrong input-gradient conv to the layer it sits in."""
    mean, var = _bn_stats(W[prefix + '.running_means'], W[prefix + '.running_vars'], truncation)
    mean = mean.view(1, -1, 1, 1)
    var = var.view(1, -1, 1, 1)
    weight = 1 + F.linear(cond, W[prefix + '.scale.weight']).unsqueeze(-1).unsqueeze(-1)
    bias = F.linear(cond, W[prefix + '.offset.weight']).unsqueeze(-1).unsqueeze(-1)
    if taps is not None and weight.requires_grad:
        weight.retain_grad()
        bias.retain_grad()
        taps[prefix] = (weight, bias)
    return (x - mean) / torch.sqrt(var + BN_EPS) * weight + bias


def gen_block(W, p, x, cond, truncation, up, cin, cout, taps=None, tape=None):
    """GenBlock.forward [3P-recall]: bottleneck 1x1 -> 3x3 -> 3x3 -> 1x1 + shortcut."""
    x0 = x
    x = _relu(cbn(W, p + '.bn_0', x, cond, truncation, taps), tape, p + '.bn_0')
    x = F.conv2d(x, W[p + '.conv_0.weight'], W[p + '.conv_0.bias'])
    x = _relu(cbn(W, p + '.bn_1', x, cond, truncation, taps), tape, p + '.bn_1')
    if up:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    x = F.conv2d(x, W[p + '.conv_1.weight'], W[p + '.conv_1.bias'], padding=1)
    x = _relu(cbn(W, p + '.bn_2', x, cond, truncation, taps), tape, p + '.bn_2')
    x = F.conv2d(x, W[p + '.conv_2.weight'], W[p + '.conv_2.bias'], padding=1)
    x = _relu(cbn(W, p + '.bn_3', x, cond, truncation, taps), tape, p + '.bn_3')
    x = F.conv2d(x, W[p + '.conv_3.weight'], W[p + '.conv_3.bias'])
    if cin != cout:
        x0 = x0[:, :cin // 2]
    if up:
        x0 = F.interpolate(x0, scale_factor=2, mode='nearest')
    return x + x0


def self_attn(W, p, x, tape=None):
    """SelfAttn.forward [3P-recall]."""
    b, ch, h, w = x.shape
    theta = F.conv2d(x, W[p + '.snconv1x1_theta.weight']).view(b, ch // 8, h * w)
    phi = _maxpool2(F.conv2d(x, W[p + '.snconv1x1_phi.weight']), tape, p + '.phi').view(b, ch // 8, h * w // 4)
    attn = torch.softmax(torch.bmm(theta.permute(0, 2, 1), phi), dim=-1)
    g = _maxpool2(F.conv2d(x, W[p + '.snconv1x1_g.weight']), tape, p + '.g').view(b, ch // 2, h * w // 4)
    attn_g = torch.bmm(g, attn.permute(0, 2, 1)).view(b, ch // 2, h, w)
    attn_g = F.conv2d(attn_g, W[p + '.snconv1x1_o_conv.weight'])
    return x + W[p + '.gamma'] * attn_g


def generator_forward(W, cond, truncation=1.0, ch=CH, layers=LAYERS, attn_pos=ATTN_POS,
                      return_intermediates=False, cbn_taps=None, tape=None):
    """Generator.forward [3P-recall]; `cond` = cat(z, class_embedding) [B, 2*z_dim]."""
    inter = {}
    z = F.linear(cond, W['generator.gen_z.weight'], W['generator.gen_z.bias'])
    z = z.view(-1, 4, 4, 16 * ch).permute(0, 3, 1, 2).contiguous()
    inter['gen_z'] = z
    for i, spec in enumerate(layer_table(ch, layers, attn_pos)):
        p = 'generator.layers.%d' % i
        if spec[0] == 'attn':
            z = self_attn(W, p, z, tape)
        else:
            _, up, cin, cout = spec
            z = gen_block(W, p, z, cond, truncation, up, cin, cout, cbn_taps, tape)
        inter['layer%d' % i] = z
    mean, var = _bn_stats(W['generator.bn.running_means'], W['generator.bn.running_vars'], truncation)
    z = F.batch_norm(z, mean, var, W['generator.bn.weight'], W['generator.bn.bias'],
                     training=False, momentum=0.0, eps=BN_EPS)
    z = _relu(z, tape, 'generator.bn')
    z = F.conv2d(z, W['generator.conv_to_rgb.weight'], W['generator.conv_to_rgb.bias'], padding=1)
    z = z[:, :3]
    out = torch.tanh(z)
    if return_intermediates:
        return out, inter
    return out


def biggan_forward(W, z, c, truncation=1.0, **kw):
    """pix2latent.model.BigGAN.forward (reference pix2latent/model/biggan.py:50-58)."""
    assert 0 < truncation <= 1
    assert z.dim() == 2, 'expected z to be 2D'
    assert c.dim() == 2, 'expected c to be 2D'
    assert c.size(1) == 128, 'expected c to have dim (?, 128) but got {}'.format(c.size())
    return generator_forward(W, torch.cat((z, c), dim=1), truncation, **kw)


def class_embedding(W, cls):
    """BigGAN.get_class_embedding (reference pix2latent/model/biggan.py:37-47):
    one-hot(1000) @ embeddings (Linear(1000,128,bias=False))."""
    with torch.no_grad():
        if isinstance(cls, int):
            c = torch.zeros(1, W['embeddings.weight'].shape[1])
            c[:, cls] = 1
        elif cls.dim() == 2:
            c = cls
        else:
            raise ValueError
        return F.linear(c, W['embeddings.weight'])
